"""GPU diagnostic: where does ccn_curve_fps differ from the oracle?"""
import numpy as np
import torch
from curvecloudnet_amd import ops
from curvecloudnet_amd.synth import make_batch
from oracle import torch_ref as R

for ids, sp in (([6], 0.03), ([1, 2, 3], 0.007)):
    d = make_batch(ids, n_curves=600)
    u = torch.rand(1, generator=torch.Generator().manual_seed(9))
    want = R.curve_fps(d.pos, d.batch, d.curve_idxs, sp, u)
    topo = ops.CurveTopology(d.batch.cuda(), d.curve_idxs.cuda())
    got = ops.curve_fps(d.pos.cuda(), topo, sp, float(u)).cpu()
    w, g = set(want.tolist()), set(got.tolist())
    print("case", ids, sp, "want", len(w), "got", len(g), "only_want", sorted(w - g)[:10], "only_got", sorted(g - w)[:10])
    # CPU intermediates
    glob = R.curve_ids_global(d.curve_idxs, d.batch)
    start = R.curve_start_of_point(glob)
    run = torch.cat([torch.zeros(1), torch.cumsum(R._edge_lengths(d.pos, glob), 0)])
    arclen = run - run[start]
    phase = (start * 117 * u) % sp
    ratio = (arclen + phase) / sp
    for i in sorted((w ^ g))[:6]:
        print("  i", i, "start", int(start[i]), "ratio[i-1], ratio[i]", float(ratio[i - 1]), float(ratio[i]),
              "phase", float(phase[i]), "scaled", float((start * 117 * u)[i]), "arclen", float(arclen[i]))

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from curvecloudnet_amd import ops, steps
from curvecloudnet_amd.nn import MLP
from curvecloudnet_amd.synth import make_batch
DEV = "cuda:0"
d = make_batch([0, 1, 2], n_curves=120)
c = 13
torch.manual_seed(0)
mod = steps.CurveSAModule(None, 0.02, MLP([c + 3 + 3, 64, 32], act="leaky_relu", bias=True), curve_fps_arclen=0.007,
                          use_curve_fps=True, with_xyz=True, aggr_type="max", normalize_radius=True).to(DEV).train()
x = torch.randn(d.pos.size(0), c, generator=torch.Generator().manual_seed(4)).to(DEV)
names = ["out", "dx"] + [n for n, _ in mod.named_parameters()]
runs = []
for gather in (True, True, False, False):
    ops.PN_BWD_GATHER = gather
    for bn in mod.conv.local_nn.norms:
        bn.module.reset_running_stats()
    xi = x.clone().requires_grad_(True)
    torch.manual_seed(3)
    out = mod(xi, d.pos.to(DEV), d.batch.to(DEV), d.curve_idxs.to(DEV))[0]
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    runs.append([out.detach()] + [g.detach().clone() for g in torch.autograd.grad((out * cot).sum(), [xi] + list(mod.parameters()))])
for (a, b, what) in ((0, 1, "gather vs gather"), (2, 3, "atomic vs atomic"), (0, 2, "gather vs atomic")):
    print(what)
    for n, u, v in zip(names, runs[a], runs[b]):
        print("   %-40s equal %s  max|diff| %.3e  scale %.3e" % (n, torch.equal(u, v), float((u - v).abs().max()), float(v.abs().max())))

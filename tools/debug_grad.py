"""Diagnostic: routed gradient parity of the hot-path network, all tensors listed, under A/B switches.
usage: python tools/debug_grad.py <width> <curves> [switch ...]   switches: tn0 (register-staged weight gradients),
dma0 (no LDS-DMA GEMMs), edge (literal edge GEMMs), nocompact"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests.util import build_pair, hotpath_config, routed_parity, maxdiff  # noqa: E402
from curvecloudnet_amd import _lib, steps  # noqa: E402
from curvecloudnet_amd.synth import make_batch  # noqa: E402

width, curves = float(sys.argv[1]), int(sys.argv[2])
sw = set(sys.argv[3:])
if "tn0" in sw:
    _lib.lib().ccn_gemm_tn_use_dma(0)
if "dma0" in sw:
    _lib.lib().ccn_gemm_use_dma(0)
ref, mine = build_pair(hotpath_config(width=width), in_dim=4, n_out=20)
mine = mine.to("cuda:0")
for m in mine.modules():
    if isinstance(m, (steps.SGCNNLayer, steps.PointNetConv2)):
        if "edge" in sw:
            m.force_edge_gemm = True
        if "nocompact" in sw and hasattr(m, "compact_rows"):
            m.compact_rows = False
data = make_batch([0], n_curves=curves)
y = torch.randint(0, 20, (data.pos.size(0),), generator=torch.Generator().manual_seed(3))
ref.train(); mine.train()
res = routed_parity(ref, mine, data, y, "cuda:0")
print("switches", sorted(sw), "points", data.pos.size(0), "logits diff %.2e" % maxdiff(res["out_d"], res["out_r"]),
      "flips", res["flips"], "of", res["entries"], "gap %.2e" % res["max_gap"], "gmax %.3g" % res["grad_scale"])
for (e, n), (_, p) in zip(res["grad_err"], ref.named_parameters()):
    print("  %.2e  |g|max %.2e  %s %s" % (e, float(p.grad.abs().max()), n, tuple(p.shape)))

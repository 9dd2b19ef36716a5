"""16-bit curve convolutions: implicit GEMM on the 16-bit sequence (ConvRowsBNActH) vs the 16-bit shifted-row matrix, and
both vs the CPU emulation, per layer configuration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvecloudnet_amd import ops, steps  # noqa: E402
from curvecloudnet_amd.synth import make_batch  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

dev = "cuda:0"
d = make_batch([2, 3], n_curves=120)
for mode in ("bf16", "fp16"):
    for ver, dims, k in (("v2", [4, 32, 32, 32], 5), ("v2", [67, 32, 32], 5), ("v1", [19, 16, 8, 16], 5), ("v1", [131, 64, 64], 7)):
        torch.manual_seed(0)
        cls_r = R.SymmetricCurve1DConvV2 if ver == "v2" else R.SymmetricCurve1DConvFastV1
        cls_d = steps.SymmetricCurve1DConvV2 if ver == "v2" else steps.SymmetricCurve1DConvFastV1
        ref = cls_r(dims, k, with_xyz=True, with_diff=True).train()
        mine = cls_d(dims, k, with_xyz=True, with_diff=True)
        mine.load_state_dict(ref.state_dict())
        mine = mine.to(dev).train()
        x = torch.randn(d.pos.size(0), dims[0] - 3, generator=torch.Generator().manual_seed(1))
        ops.set_mlp_dtype(mode); R.set_mlp_dtype(mode); R.STORE16 = ops.STORE16
        try:
            outs = {}
            for flag in (True, False):
                ops.CONV_IMPLICIT_H = flag
                outs[flag] = mine(x.to(dev), d.pos.to(dev), d.batch.to(dev), d.curve_idxs.to(dev))[0].detach().cpu()
            ops.CONV_IMPLICIT_H = True
            out_r = ref(x, d.pos, d.batch, d.curve_idxs)[0].detach()
            ops.set_mlp_dtype("fp32")
            out_f = mine(x.to(dev), d.pos.to(dev), d.batch.to(dev), d.curve_idxs.to(dev))[0].detach().cpu()
        finally:
            ops.set_mlp_dtype("fp32"); R.set_mlp_dtype("fp32")
        rel = lambda a, b: float((a - b).norm() / b.norm())
        print("%s %s %-18s k=%d: implicit vs shifted-row %.2e | vs emulation: implicit %.2e, shifted-row %.2e | mode vs fp32 %.2e"
              % (mode, ver, dims, k, rel(outs[True], outs[False]), rel(outs[True], out_r), rel(outs[False], out_r), rel(outs[True], out_f)))

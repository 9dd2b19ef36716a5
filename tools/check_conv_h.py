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
            outs, grads = {}, {}
            cot = torch.randn(d.pos.size(0), dims[-1], generator=torch.Generator().manual_seed(3))
            for flag in (True, False):
                ops.CONV_IMPLICIT_H = flag
                xi = x.to(dev).requires_grad_(True)
                o = mine(xi, d.pos.to(dev), d.batch.to(dev), d.curve_idxs.to(dev))[0]
                grads[flag] = [g.detach().cpu() for g in torch.autograd.grad((o * cot.to(dev)).sum(), [xi] + list(mine.parameters()))]
                outs[flag] = o.detach().cpu()
            ops.CONV_IMPLICIT_H = True
            xr = x.clone().requires_grad_(True)
            o_r = ref(xr, d.pos, d.batch, d.curve_idxs)[0]
            g_r = [g.detach() for g in torch.autograd.grad((o_r * cot).sum(), [xr] + list(ref.parameters()))]
            out_r = o_r.detach()
            names = ["x"] + [n for n, _ in ref.named_parameters()]
            gm = max(float(g.norm()) for g in g_r)
            for n, a, b, r in zip(names, grads[True], grads[False], g_r):
                den = max(float(r.norm()), 1e-3 * gm)
                print("      grad %-28s implicit vs emulation %.2e   shifted-row vs emulation %.2e   implicit vs shifted-row %.2e"
                      % (n, float((a - r).norm()) / den, float((b - r).norm()) / den, float((a - b).norm()) / den))
            ops.set_mlp_dtype("fp32")
            out_f = mine(x.to(dev), d.pos.to(dev), d.batch.to(dev), d.curve_idxs.to(dev))[0].detach().cpu()
        finally:
            ops.set_mlp_dtype("fp32"); R.set_mlp_dtype("fp32")
        rel = lambda a, b: float((a - b).norm() / b.norm())
        print("%s %s %-18s k=%d: implicit vs shifted-row %.2e | vs emulation: implicit %.2e, shifted-row %.2e | mode vs fp32 %.2e"
              % (mode, ver, dims, k, rel(outs[True], outs[False]), rel(outs[True], out_r), rel(outs[False], out_r), rel(outs[True], out_f)))

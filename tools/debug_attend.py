import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvecloudnet_amd import steps, ops
from curvecloudnet_amd.nn import MLP
from curvecloudnet_amd.synth import make_batch
DEV = "cuda"
ops.set_mlp_dtype("bf16")
d = make_batch([2, 3], n_curves=50)
c = 21
torch.manual_seed(0)
mod = steps.SGCNNLayer(MLP([2 * (c + 3), 40, 24], bias=False), 12, r=0.05, with_xyz=True, aggr_type="attend",
                       use_sparse_feat_agg=True, attend_nn=MLP([24, 16, 24], act="leaky_relu", bias=True)).to(DEV).train()
x = torch.randn(d.pos.size(0), c, generator=torch.Generator().manual_seed(4)).to(DEV)
res = []
names = ["x"] + [n for n, _ in mod.named_parameters()]
for direct in (True, False):
    ops.EDGE_OUT16 = direct
    xi = x.clone().requires_grad_(True)
    torch.manual_seed(9)
    out = mod(xi, d.pos.to(DEV), d.batch.to(DEV), d.curve_idxs.to(DEV))[0]
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    res.append([out.detach()] + list(torch.autograd.grad((out * cot).sum(), [xi] + list(mod.parameters()))))
print("out equal", torch.equal(res[0][0], res[1][0]))
for n, a, b in zip(names, res[0][1:], res[1][1:]):
    print("%-32s |a| %.3e |b| %.3e |a-b| %.3e" % (n, float(a.norm()), float(b.norm()), float((a - b).norm())))

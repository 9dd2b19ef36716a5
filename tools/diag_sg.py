import torch
from oracle import torch_ref as R
from curvecloudnet_amd import steps
from curvecloudnet_amd.nn import MLP
from curvecloudnet_amd.synth import make_batch
from tests.util import maxdiff

def run(ids, n_curves, c, K, r, hidden, xyz=True):
    d = make_batch(ids, n_curves=n_curves)
    cin = 2 * (c + (3 if xyz else 0))
    torch.manual_seed(1)
    ref = R.SGCNNLayer(R.MLP([cin] + hidden, bias=False), K, r=r, with_xyz=xyz).train()
    mine = steps.SGCNNLayer(MLP([cin] + hidden, bias=False), K, r=r, with_xyz=xyz)
    mine.load_state_dict(ref.state_dict()); mine = mine.cuda().train()
    x = torch.randn(d.pos.size(0), c, generator=torch.Generator().manual_seed(4))
    xr = x.clone().requires_grad_(True); xd = x.cuda().requires_grad_(True)
    o_r = ref(xr, d.pos, d.batch, d.curve_idxs)[0]
    o_d = mine(xd, d.pos.cuda(), d.batch.cuda(), d.curve_idxs.cuda())[0]
    cot = torch.randn(o_r.shape, generator=torch.Generator().manual_seed(2))
    gr = torch.autograd.grad((o_r * cot).sum(), [xr] + list(ref.parameters()))
    gd = torch.autograd.grad((o_d * cot.cuda()).sum(), [xd] + list(mine.parameters()))
    errs = ["%.1e" % (maxdiff(a, b) / float(b.abs().max())) for a, b in zip(gd, gr)]
    print(ids, n_curves, "c", c, "K", K, "r", r, hidden, "fwd %.1e" % maxdiff(o_d, o_r), "grads", errs)

run([1, 2], 64, 32, 20, 0.08, [32, 32])
run([1], 64, 32, 20, 0.08, [32, 32])
run([1, 2], 64, 32, 8, 0.08, [32, 32])
run([1, 2], 64, 32, 20, 0.02, [32, 32])
run([1, 2], 64, 13, 20, 0.08, [32, 32])
run([1, 2], 64, 32, 20, 0.08, [32])
run([1, 2], 64, 32, 20, 0.08, [64, 32])
run([1, 2, 3], 60, 13, 8, 0.03, [32, 24])
run([1, 2], 64, 32, 20, 0.08, [32, 32], xyz=False)

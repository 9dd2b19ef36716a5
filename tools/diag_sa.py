import torch
from oracle import torch_ref as R
from curvecloudnet_amd import steps
from curvecloudnet_amd.nn import MLP
from curvecloudnet_amd.synth import make_batch
from tests.util import maxdiff
from tests.test_gpu_float import _pair
d = make_batch([5, 6], n_curves=30)
c = 6
for fast, aggr, r in ((False, "attend", 0.2), (False, "max", 0.2), (False, "attend", 0.05), (True, "attend", 0.2)):
    def mk(mod, mlp):
        att = mlp([24, 24, 24], act="leaky_relu", bias=True) if aggr == "attend" else None
        return mod(0.25, r, mlp([c + 3, 32, 24], bias=True), 16, downsample_type="fps", aggr_type=aggr,
                   attend_nn=att, normalize_radius=True, use_fast_knn=fast)
    ref, mine = _pair(lambda: mk(R.SAModule, R.MLP), lambda: mk(steps.SAModule, MLP))
    x = torch.randn(d.pos.size(0), c, generator=torch.Generator().manual_seed(4))
    xr = x.clone().requires_grad_(True); xd = x.cuda().requires_grad_(True)
    torch.manual_seed(3); o_r = ref(xr, d.pos, d.batch, d.curve_idxs)
    torch.manual_seed(3); o_d = mine(xd, d.pos.cuda(), d.batch.cuda(), d.curve_idxs.cuda())
    cot = torch.randn(o_r[0].shape, generator=torch.Generator().manual_seed(2))
    gr = torch.autograd.grad((o_r[0] * cot).sum(), [xr] + list(ref.parameters()))
    gd = torch.autograd.grad((o_d[0] * cot.cuda()).sum(), [xd] + list(mine.parameters()))
    names = ["x"] + [n for n, _ in ref.named_parameters()]
    print(fast, aggr, r, "fwd %.1e" % maxdiff(o_d[0], o_r[0]), "idx eq", torch.equal(o_d[1].cpu(), o_r[1]))
    print("   ", ["%s %.1e" % (n.split('.')[-2][-6:] + '.' + n.split('.')[-1] if '.' in n else n, maxdiff(a, b) / max(1.0, float(b.abs().max()))) for a, b, n in zip(gd, gr, names)])
    # float64 referee
    import copy
    ref64 = copy.deepcopy(ref).double()
    x64 = x.double().requires_grad_(True)
    torch.manual_seed(3); o64 = ref64(x64, d.pos.double(), d.batch, d.curve_idxs)
    g64 = torch.autograd.grad((o64[0] * cot.double()).sum(), [x64])[0]
    print("    x-grad: oracle32 vs 64 %.1e   product vs 64 %.1e" % (maxdiff(gr[0], g64), maxdiff(gd[0], g64)))

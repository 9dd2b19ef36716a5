import torch
from oracle import torch_ref as R
from curvecloudnet_amd.model import segmentation_loss
from curvecloudnet_amd.synth import make_batch
from tests.util import batch_to, build_pair, hotpath_config, maxdiff

ref, mine = build_pair(hotpath_config(width=0.25), in_dim=4, n_out=7)
mine = mine.cuda().train(); ref.train()
data = make_batch([1, 2], n_curves=64)
cap = {}
def pre(tag):
    def f(mod, args, kwargs):
        cap[tag] = [a.detach().clone() if torch.is_tensor(a) else a for a in args]
    return f
ref.steps[5].register_forward_pre_hook(pre("ref"), with_kwargs=True)
mine.steps[5].register_forward_pre_hook(pre("mine"), with_kwargs=True)
torch.manual_seed(5); out_r = ref(data)
torch.manual_seed(5); out_d = mine(batch_to(data, "cuda"))
a, b = cap["ref"], cap["mine"]
print("inputs: x %.2e pos %.2e batch eq %s p2c eq %s" % (maxdiff(b[0], a[0]), maxdiff(b[1], a[1]), torch.equal(b[2].cpu(), a[2]), torch.equal(b[3].cpu(), a[3])))
print("batch counts", torch.bincount(a[2]).tolist())
# isolated run on the captured inputs (oracle inputs fed to both)
xr = a[0].clone().requires_grad_(True); xd = a[0].cuda().requires_grad_(True)
o_r = ref.steps[5](xr, a[1], a[2], a[3])[0]
o_d = mine.steps[5](xd, a[1].cuda(), a[2].cuda(), a[3].cuda())[0]
cot = torch.randn(o_r.shape, generator=torch.Generator().manual_seed(2))
gr = torch.autograd.grad((o_r * cot).sum(), [xr] + list(ref.steps[5].parameters()))
gd = torch.autograd.grad((o_d * cot.cuda()).sum(), [xd] + list(mine.steps[5].parameters()))
print("isolated fwd %.2e" % maxdiff(o_d, o_r), ["%.1e" % (maxdiff(p, q) / float(q.abs().max())) for p, q in zip(gd, gr)])
# neighbour lists
from curvecloudnet_amd import ops
topo = ops.CurveTopology(a[2].cuda(), a[3].cuda())
padded, _ = ops.to_batch_padded(a[1].cuda(), topo)
nbr = ops.fast_knn(padded, padded, topo.lengths, topo.lengths, 20, 0.08)
want, l2, m1 = R.group_fixed_radius(a[1], a[1], a[2], a[2], 20, 0.08, return_dense=True)
print("nbr equal", torch.equal(nbr.cpu(), want), "filled frac", float((want >= 0).float().mean()))
d2 = R.frnn_bruteforce(*R.padded_layout(a[1], a[2])[0:1], R.padded_layout(a[1], a[2])[0], l2, l2, 20, 0.08, return_dists=True)[1]
ties = ((d2[:, :, 1:] == d2[:, :, :-1]) & (d2[:, :, 1:] >= 0)).sum()
print("exact distance ties among kept neighbours:", int(ties))
# --- is it conditioning?  float64 oracle as the referee
import copy
ref64 = copy.deepcopy(ref.steps[5]).double()
x64 = a[0].double().requires_grad_(True)
o64 = ref64(x64, a[1], a[2], a[3])[0]
g64 = torch.autograd.grad((o64 * cot.double()).sum(), [x64] + list(ref64.parameters()))
print("oracle32 vs oracle64:", ["%.1e" % (maxdiff(p, q) / float(q.abs().max())) for p, q in zip(gr, g64)])
print("product  vs oracle64:", ["%.1e" % (maxdiff(p, q) / float(q.abs().max())) for p, q in zip(gd, g64)])

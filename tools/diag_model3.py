import torch, torch.nn.functional as F
from oracle import torch_ref as R
from curvecloudnet_amd import ops
from curvecloudnet_amd.synth import make_batch
from tests.util import batch_to, build_pair, hotpath_config, maxdiff

ref, mine = build_pair(hotpath_config(width=0.25), in_dim=4, n_out=7)
mine = mine.cuda().train(); ref.train()
data = make_batch([1, 2], n_curves=64)
cap = {}
ref.steps[5].register_forward_pre_hook(lambda m, a, k: cap.__setitem__("a", [t.detach().clone() for t in a]), with_kwargs=True)
torch.manual_seed(5); ref(data)
x, pos, batch, p2c = [t.cuda() for t in cap["a"]]
step = mine.steps[5]
topo = ops.CurveTopology(batch, p2c)
xin = torch.cat([x, pos], 1)
padded, _ = ops.to_batch_padded(pos, topo)
nbr = ops.fast_knn(padded, padded, topo.lengths, topo.lengths, 20, 0.08)
B, Nmax, K = nbr.shape
C = xin.size(1)
rel = lambda a, b: maxdiff(a, b) / float(b.abs().max())

# ---- 1. SGGather fwd/bwd vs torch
xa = xin.clone().requires_grad_(True)
feat = ops.SGGather.apply(xa, nbr, topo.cloud_ptr)
xb = xin.clone().requires_grad_(True)
xp, mask1 = ops.to_batch_padded(xb, topo)
me = torch.arange(Nmax, device="cuda").view(1, Nmax, 1).expand(B, Nmax, 1)
full = torch.cat([me, nbr], 2)
g = torch.gather(xp[:, :, None, :].expand(-1, -1, K + 1, -1), 1, full.clamp(min=0)[..., None].expand(-1, -1, -1, C))
g = torch.where((full >= 0)[..., None], g, torch.zeros((), device="cuda"))
feat_t = torch.cat([g, g[:, :, 0:1] - g], -1).reshape(-1, 2 * C)
cot = torch.randn_like(feat_t)
print("gather fwd", maxdiff(feat, feat_t), "bwd", rel(torch.autograd.grad((feat * cot).sum(), xa)[0], torch.autograd.grad((feat_t * cot).sum(), xb)[0]))

# ---- 2. MLP layers vs torch (GPU)
f_in = feat.detach()
lin0, bn0, lin1 = step.nn.lins[0], step.nn.norms[0].module, step.nn.lins[1]
a1 = f_in.clone().requires_grad_(True)
h = ops.linear_bn_act(a1, lin0.weight, None, bn0, True, "relu")
a2 = f_in.clone().requires_grad_(True)
h_t = F.relu(F.batch_norm(F.linear(a2, lin0.weight), None, None, bn0.weight, bn0.bias, True, 0.1, 1e-5))
cot = torch.randn_like(h_t)
ga = torch.autograd.grad((h * cot).sum(), [a1, lin0.weight, bn0.weight, bn0.bias])
gb = torch.autograd.grad((h_t * cot).sum(), [a2, lin0.weight, bn0.weight, bn0.bias])
print("layer0 fwd", maxdiff(h, h_t), "bwd", [("%.1e" % rel(p, q)) for p, q in zip(ga, gb)])
h_in = h.detach()
b1 = h_in.clone().requires_grad_(True)
o = ops.linear_bn_act(b1, lin1.weight, None, None, True, None)
b2 = h_in.clone().requires_grad_(True)
o_t = F.linear(b2, lin1.weight)
cot = torch.randn_like(o_t)
ga = torch.autograd.grad((o * cot).sum(), [b1, lin1.weight]); gb = torch.autograd.grad((o_t * cot).sum(), [b2, lin1.weight])
print("layer1 fwd", maxdiff(o, o_t), "bwd", [("%.1e" % rel(p, q)) for p, q in zip(ga, gb)])

# ---- 3. SGMax vs torch
f = o.detach()
fa = f.clone().requires_grad_(True)
out = ops.SGMax.apply(fa, nbr, topo.cloud_ptr, topo.n)
fb = f.clone().requires_grad_(True)
mask = (full != -1) & mask1[:, :, None]
ft = torch.where(mask[..., None], fb.view(B, Nmax, K + 1, -1), torch.full((), -1e2, device="cuda")).max(dim=2)[0][mask1]
cot = torch.randn_like(ft)
ga = torch.autograd.grad((out * cot).sum(), fa)[0]; gb = torch.autograd.grad((ft * cot).sum(), fb)[0]
print("sgmax fwd", maxdiff(out, ft), "bwd", rel(ga, gb), "nonzero rows differ:", int(((ga != 0) != (gb != 0)).sum()))

# ---- 4. composite on GPU with pure torch ops vs CPU oracle vs product
import copy
cpu_in = [t.cpu() for t in (x, pos, batch, p2c)]
cot = torch.randn(topo.n, 32, generator=torch.Generator().manual_seed(2))
xr = cpu_in[0].clone().requires_grad_(True)
o_r = ref.steps[5](xr, *cpu_in[1:])[0]
g_r = torch.autograd.grad((o_r * cot).sum(), [xr] + list(ref.steps[5].parameters()))
xd = x.clone().requires_grad_(True)
o_d = step(xd, pos, batch, p2c)[0]
g_d = torch.autograd.grad((o_d * cot.cuda()).sum(), [xd] + list(step.parameters()))
xt = x.clone().requires_grad_(True)
xin_t = torch.cat([xt, pos], 1)
xp, _ = ops.to_batch_padded(xin_t, topo)
g = torch.gather(xp[:, :, None, :].expand(-1, -1, K + 1, -1), 1, full.clamp(min=0)[..., None].expand(-1, -1, -1, C))
g = torch.where((full >= 0)[..., None], g, torch.zeros((), device="cuda"))
ft = torch.cat([g, g[:, :, 0:1] - g], -1).reshape(-1, 2 * C)
ht = F.relu(F.batch_norm(F.linear(ft, lin0.weight), None, None, bn0.weight, bn0.bias, True, 0.1, 1e-5))
ot = F.linear(ht, lin1.weight).view(B, Nmax, K + 1, -1)
ot = torch.where(mask[..., None], ot, torch.full((), -1e2, device="cuda"))
val, arg_t = ot.max(dim=2)
o_t = val[mask1]
g_t = torch.autograd.grad((o_t * cot.cuda()).sum(), [xt, lin0.weight, lin1.weight, bn0.weight, bn0.bias])
print("torchGPU vs oracleCPU:", ["%.1e" % rel(p, q) for p, q in zip(g_t, g_r)])
print("product  vs torchGPU :", ["%.1e" % rel(p, q) for p, q in zip(g_d, g_t)])
# CPU argmax vs GPU argmax on the oracle's own f
with torch.no_grad():
    ocpu = ot.detach().cpu()
    a_cpu = ocpu.max(dim=2)[1]
    a_gpu = arg_t.cpu()
    m1 = mask1.cpu()
    diff = (a_cpu != a_gpu)[m1]
    print("argmax differs CPU/GPU on identical values: %d of %d ; slot pairs:" % (int(diff.sum()), diff.numel()),
          torch.stack([a_cpu[m1][diff], a_gpu[m1][diff]], 1)[:8].tolist())

# ---- 5. CPU-computed values vs GPU-computed values: where does the argmax move?
with torch.no_grad():
    f_cpu = ref.steps[5].nn(ft.detach().cpu()).view(B, Nmax, K + 1, -1)
    f_cpu = torch.where(mask.cpu()[..., None], f_cpu, torch.full((), -1e2))
    a_c = f_cpu.max(dim=2)[1][m1]
    a_g = arg_t.cpu()[m1]
    moved = a_c != a_g
    print("argmax moved between CPU values and GPU values: %d of %d" % (int(moved.sum()), moved.numel()))
    pairs = torch.stack([a_c[moved], a_g[moved]], 1)
    import collections
    print(collections.Counter(map(tuple, pairs.tolist())).most_common(8))
    top2 = f_cpu.topk(2, dim=2)[0][m1]
    gap = (top2[:, 0] - top2[:, 1])
    print("gap at moved entries: max %.2e median %.2e ; exact ties on CPU overall: %d" % (float(gap[moved].max()), float(gap[moved].median()), int((gap == 0).sum())))
    # are slot 0 and slot 1 rows bitwise equal on CPU / GPU?
    print("slot0==slot1 bitwise: CPU %.3f GPU %.3f" % (float((f_cpu[:, :, 0] == f_cpu[:, :, 1])[m1].float().mean()), float((ot[:, :, 0] == ot[:, :, 1])[mask1].float().mean())))

import sys, torch
sys.path.insert(0, '.')
from curvecloudnet_amd import configs, ops
from curvecloudnet_amd.model import build_model
from curvecloudnet_amd.synth import make_batch
from tests.util import batch_to
DEV = "cuda:0"
torch.manual_seed(4)
n_clouds, curves = int(sys.argv[1]), int(sys.argv[2])
model = build_model(configs.nuscenes_config(float(sys.argv[3])), in_dim=4, n_out=17).to(DEV).train()
data = batch_to(make_batch(list(range(n_clouds)), n_curves=curves), DEV)
ops.set_mlp_dtype("bf16")
state = {k: v.clone() for k, v in model.state_dict().items()}
outs = []
calls = []
inner = ops.call
def spy(name, *a, **kw):
    calls[-1].append((name, [x for x in a if isinstance(x, int)]))
    return inner(name, *a, **kw)
for rep in range(2):
    feats = []
    hooks = [s.register_forward_hook(lambda m, i, o, feats=feats: feats.append(o[0].detach().float().clone())) for s in model.steps]
    model.load_state_dict(state)
    torch.manual_seed(1)
    calls.append([])
    with torch.no_grad():
        out = model(data).clone()
    for h in hooks: h.remove()
    outs.append((out, feats))
print("logits diff", float((outs[0][0] - outs[1][0]).abs().max()))
for i, (a, b) in enumerate(zip(outs[0][1], outs[1][1])):
    d = float((a - b).abs().max())
    print("step", i, model.step_names[i], tuple(a.shape), "diff", d)
    if d > 0: break
# layer-level: repeat a fused layer alone
x = torch.randn(300000, 256, device=DEV)
w = torch.randn(256, 256, device=DEV) / 16
bn = torch.nn.BatchNorm1d(256).to(DEV)
for act in ("relu", "leaky_relu"):
    r = []
    for rep in range(3):
        with torch.no_grad():
            y = ops.linear_bn_act(x, w, None, bn, True, act, defer=True)
            y2 = ops.linear_bn_act(y, w, None, bn, True, act, defer=False)
        r.append((y.float().clone(), y2.float().clone()))
    print(act, "layer repeat diffs", float((r[0][0] - r[1][0]).abs().max()), float((r[0][1] - r[2][1]).abs().max()), "dtype", y.dtype)

import sys, torch
sys.path.insert(0, '.')
from curvecloudnet_amd import configs, ops
from curvecloudnet_amd.model import build_model, segmentation_loss
from curvecloudnet_amd.synth import make_batch
from tests.util import batch_to
DEV = "cuda:0"
torch.manual_seed(4)
n_clouds, curves = int(sys.argv[1]), int(sys.argv[2])
model = build_model(configs.nuscenes_config(float(sys.argv[3])), in_dim=4, n_out=17).to(DEV).train()
data = batch_to(make_batch(list(range(n_clouds)), n_curves=curves), DEV)
labels = torch.randint(0, 17, (data.pos.size(0),), device=DEV)
ops.set_mlp_dtype("bf16")
state = {k: v.clone() for k, v in model.state_dict().items()}
res = []
for mode in ("grad", "nograd", "grad_bwd_then_clone"):
    feats = []
    hooks = [s.register_forward_hook(lambda m, i, o, feats=feats: feats.append(o[0].detach().float().clone())) for s in model.steps]
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    if mode == "nograd":
        with torch.no_grad():
            out = model(data)
        snap = out.clone()
    else:
        out = model(data)
        snap0 = out.detach().clone()
        loss = segmentation_loss(out, labels)
        if mode == "grad_bwd_then_clone":
            loss.backward()
            print("out changed by backward:", float((out.detach() - snap0).abs().max()))
        snap = out.detach().clone()
    for h in hooks: h.remove()
    res.append((snap, feats))
for j in (1, 2):
    print("mode", j, "logits diff vs grad", float((res[0][0] - res[j][0]).abs().max()))
    for i, (a, b) in enumerate(zip(res[0][1], res[j][1])):
        d = float((a - b).abs().max())
        if d > 0:
            print("  first differing step", i, model.step_names[i], tuple(a.shape), "diff", d); break

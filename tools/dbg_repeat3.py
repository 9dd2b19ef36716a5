import sys, torch
sys.path.insert(0, '.')
from curvecloudnet_amd import configs, ops
from curvecloudnet_amd.model import build_model, segmentation_loss
from curvecloudnet_amd.synth import make_batch
from tests.util import batch_to
DEV = "cuda:0"
cpu = make_batch(list(range(16)), n_curves=1430)
data = batch_to(cpu, DEV)
n = cpu.pos.size(0)
labels = torch.randint(0, 17, (n,), generator=torch.Generator().manual_seed(1)).to(DEV)
torch.manual_seed(0)
model = build_model(configs.nuscenes_config(1.0), in_dim=4, n_out=17).to(DEV).train()
state = {k: v.clone() for k, v in model.state_dict().items()}
skip_fp32 = len(sys.argv) > 1 and sys.argv[1] == "nofp32"
res = {}
for mode in (("bf16",) if skip_fp32 else ("fp32", "bf16")):
    ops.set_mlp_dtype(mode)
    feats = []
    hooks = [s.register_forward_hook(lambda m, i, o, feats=feats: feats.append(o[0].detach().float().clone())) for s in model.steps]
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    out = model(data)
    for h in hooks: h.remove()
    loss = segmentation_loss(out, labels)
    snap0 = out.detach().clone()
    if mode == "bf16":
        loss.backward()
        print("out changed by backward:", float((out.detach() - snap0).abs().max()))
    res[mode] = (out.detach().clone(), feats)
    del out, loss
ops.set_mlp_dtype("bf16")
for rep in range(2):
    feats = []
    hooks = [s.register_forward_hook(lambda m, i, o, feats=feats: feats.append(o[0].detach().float().clone())) for s in model.steps]
    model.load_state_dict(state)
    torch.manual_seed(1)
    with torch.no_grad():
        out2 = model(data)
    for h in hooks: h.remove()
    print("rep", rep, "logits diff vs first bf16 forward", float((out2 - res["bf16"][0]).abs().max()))
    for i, (a, b) in enumerate(zip(res["bf16"][1], feats)):
        d = float((a - b).abs().max())
        if d > 0:
            print("  first differing step", i, model.step_names[i], tuple(a.shape), "diff", d, "rows differing", int(((a - b).abs().amax(1) > 0).sum()))
            break

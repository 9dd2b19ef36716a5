"""Forward time per ModelBase step (events on the main stream) for the KITTI bench input."""
import torch
from curvecloudnet_amd import configs
from curvecloudnet_amd.model import build_model
from curvecloudnet_amd.synth import make_batch, to_device

dev = torch.device("cuda")
model = build_model(configs.kitti_config(), 4, 20).to(dev).train()
data = to_device(make_batch(list(range(8))), dev)
marks = []
for i, step in enumerate(model.steps):
    def pre(mod, args, kwargs, i=i):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("pre", i, e))
    def post(mod, args, kwargs, out, i=i):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("post", i, e))
    step.register_forward_pre_hook(pre, with_kwargs=True)
    step.register_forward_hook(post, with_kwargs=True)
for it in range(3):
    marks.clear()
    torch.manual_seed(7)
    out = model(data)
    out.square().mean().backward()
    torch.cuda.synchronize()
pre = {i: e for k, i, e in marks if k == "pre"}
post = {i: e for k, i, e in marks if k == "post"}
tot = 0.0
for i, name in enumerate(model.step_names):
    ms = pre[i].elapsed_time(post[i])
    tot += ms
    print("%2d %-16s %7.2f ms" % (i, name, ms))
print("forward total over steps %.1f ms" % tot)

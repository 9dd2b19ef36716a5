"""GPU diagnostic: per-step activation and activation-gradient parity between oracle and product."""
import torch
from oracle import torch_ref as R
from curvecloudnet_amd.model import segmentation_loss
from curvecloudnet_amd.synth import make_batch
from tests.util import batch_to, build_pair, hotpath_config, maxdiff

ref, mine = build_pair(hotpath_config(width=0.25), in_dim=4, n_out=7)
mine = mine.cuda().train(); ref.train()
data = make_batch([1, 2], n_curves=64)
y = torch.randint(0, 7, (data.pos.size(0),), generator=torch.Generator().manual_seed(3))
acts = {"ref": [], "mine": []}
def hook(tag):
    def f(mod, inp, out):
        o = out[0]
        o.retain_grad()
        acts[tag].append(o)
    return f
for s in ref.steps: s.register_forward_hook(hook("ref"))
for s in mine.steps: s.register_forward_hook(hook("mine"))
torch.manual_seed(5); R.segmentation_loss(ref(data), y).backward()
torch.manual_seed(5); segmentation_loss(mine(batch_to(data, "cuda")), y.cuda()).backward()
for i, (a, b) in enumerate(zip(acts["ref"], acts["mine"])):
    ga, gb = a.grad, b.grad
    print("step %d %-16s act %.2e  grad rel %.2e  (|g| %.2e) shape %s" % (
        i, ref.step_names[i], maxdiff(b, a), maxdiff(gb, ga) / float(ga.abs().max()), float(ga.abs().max()), tuple(a.shape)))

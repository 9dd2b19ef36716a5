"""Autograd nodes whose output feeds more than one consumer in a bench forward pass (their gradients are summed with an
extra add kernel in backward).   PYTHONPATH=. python tools/grad_fanin.py [--config kitti]"""
import argparse
import collections
import torch
import bench
from curvecloudnet_amd.model import ModelBase, segmentation_loss

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="kitti")
a = ap.parse_args()
args = argparse.Namespace(config=a.config, curves=2048, mixed_lengths=False, clouds_per_gpu=8, width=1.0)
dev = torch.device("cuda", 0)
make_cfg, in_dim, n_classes, _ = bench.networks()[args.config]
cfg = make_cfg(width=1.0)
model = ModelBase(in_dim, n_classes, **{k: v for k, v in cfg.items() if k != "type"}).to(dev).train()
from curvecloudnet_amd.synth import to_device  # noqa: E402
data = to_device(bench.make_input(list(range(8)), in_dim, args), dev)
labels = torch.randint(0, n_classes, (data.pos.size(0),), device=dev)
torch.manual_seed(7)
loss = segmentation_loss(model(data), labels)
uses = collections.Counter()
parents = collections.defaultdict(list)
seen, stack = set(), [loss.grad_fn]
while stack:
    n = stack.pop()
    if n in seen:
        continue
    seen.add(n)
    for child, idx in n.next_functions:
        if child is None:
            continue
        uses[(child, idx)] += 1
        parents[(child, idx)].append(n.name())
        stack.append(child)
print("%d nodes" % len(seen))
for (node, idx), c in sorted(uses.items(), key=lambda kv: -kv[1]):
    if c > 1 and "AccumulateGrad" not in node.name():
        meta = ""
        try:
            meta = str([tuple(m.shape) for m in node._input_metadata][:2])
        except Exception:
            pass
        print("%d consumers of output %d of %-34s <- %s %s" % (c, idx, node.name(), sorted(parents[(node, idx)]), meta))

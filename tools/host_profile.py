"""Host-side profile (cProfile) of ModelBase.prepare -- the sampling / neighbour-search pass of the next batch, where it is the
step's critical path (BASELINE configs[4]):  python tools/host_profile.py [--baseline-config i] [--top n]"""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import bench  # noqa: E402
from curvecloudnet_amd import ops  # noqa: E402
from curvecloudnet_amd.model import ModelBase, segmentation_loss  # noqa: E402
from curvecloudnet_amd.synth import to_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--baseline-config", type=int, default=4)
ap.add_argument("--top", type=int, default=35)
a = ap.parse_args()
preset = bench.BASELINE_PRESETS[a.baseline_config]
bench.torch = torch
args = argparse.Namespace(curves=preset["curves"], mixed_lengths=preset["mixed_lengths"])
ops.set_mlp_dtype(preset["mlp_dtype"])
make_cfg, in_dim, n_classes, _ = bench.networks()[preset["config"]]
cfg = {k: v for k, v in make_cfg(width=1.0).items() if k != "type"}
torch.manual_seed(1234)
dev = torch.device("cuda", 0)
model = ModelBase(in_dim, n_classes, **cfg).to(dev).train()
data = to_device(bench.make_input(list(range(preset["clouds_per_gpu"])), in_dim, args), dev)
labels = torch.randint(0, n_classes, (data.pos.size(0),), generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(3):
    model.zero_grad(set_to_none=True)
    torch.manual_seed(7)
    segmentation_loss(model(data, plan=model.prepare(data)), labels).backward()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    torch.manual_seed(7)
    plan = model.prepare(data)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(a.top)

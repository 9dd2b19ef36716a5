"""Arguments and durations of the farthest-point-sampling launches of one geometry pass of a bench configuration:
   PYTHONPATH=. python tools/fps_args.py --config a2d2 --mixed-lengths"""
import sys
import torch
import bench
from curvecloudnet_amd import _lib, ops
from curvecloudnet_amd.model import ModelBase

ap_args = sys.argv[1:]
sys.argv = ["bench.py"] + ap_args
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="kitti"); ap.add_argument("--curves", type=int, default=2048)
ap.add_argument("--mixed-lengths", action="store_true"); ap.add_argument("--clouds-per-gpu", type=int, default=8)
ap.add_argument("--width", type=float, default=1.0)
args = ap.parse_args(ap_args)
dev = torch.device("cuda", 0)
make_cfg, in_dim, n_classes, _ = bench.NETWORKS[args.config]
cfg = make_cfg(width=args.width)
model = ModelBase(in_dim, n_classes, **{k: v for k, v in cfg.items() if k != "type"}).to(dev).train()
data = bench.to_device(bench.make_input(list(range(args.clouds_per_gpu)), in_dim, args), dev)
torch.manual_seed(7)
model.prepare(data)
torch.cuda.synchronize()
_lib.PROFILE, _lib.PROFILE_ONLY = [], "fps"
torch.manual_seed(7)
model.prepare(data)
torch.cuda.synchronize()
print("points", data.pos.size(0))
for name, ints, beg, end, *_ in _lib.PROFILE:
    print(name, "B=%d max_cloud=%d" % (ints[0], ints[1]), "%.2f ms" % beg.elapsed_time(end))

"""Time every C-ABI call of one training step (HIP events on the launch stream) and print the heaviest (entry, integer
arguments) groups:  python tools/profile_calls.py [--baseline-config i] [--filter name] [--top n]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import bench  # noqa: E402
from curvecloudnet_amd import _lib, ops  # noqa: E402
from curvecloudnet_amd.model import ModelBase, segmentation_loss  # noqa: E402
from curvecloudnet_amd.synth import to_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--baseline-config", type=int, default=1)
ap.add_argument("--filter", default="")
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--cast-callers", action="store_true", help="who asks for fp32 -> 16-bit casts (16-bit storage modes)")
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


def step():
    model.zero_grad(set_to_none=True)
    torch.manual_seed(7)
    segmentation_loss(model(data), labels).backward()


for _ in range(3):
    step()
if a.cast_callers:
    import traceback
    casts = {}
    inner = ops._cast16

    def logged(x, dtype=torch.bfloat16):
        if not ops._is_rows16(x, dtype):
            frames = [f for f in traceback.extract_stack(limit=30)[:-1] if "autograd" not in f.filename]
            where = " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in frames[-12:][::-1] if "module.py" not in f.filename)
            key = (tuple(x.shape), where)
            casts[key] = casts.get(key, 0) + 1
        return inner(x, dtype)

    ops._cast16 = logged
    step()
    ops._cast16 = inner
    for (shape, where), cnt in sorted(casts.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1] * kv[1]):
        print("%3d x %-18s %7.1f MB  %s" % (cnt, shape, cnt * shape[0] * shape[1] * 6e-6, where))
torch.cuda.synchronize()
_lib.PROFILE = []
step()
torch.cuda.synchronize()
rec, _lib.PROFILE = _lib.PROFILE, None
groups = {}
for name, ints, beg, end, nulls, *_ in rec:
    if a.filter and a.filter not in name:
        continue
    g = groups.setdefault((name, ints), [0.0, 0])
    g[0] += beg.elapsed_time(end)
    g[1] += 1
tot = sum(v[0] for v in groups.values())
print("total %.2f ms in %d calls (%s)" % (tot, sum(v[1] for v in groups.values()), a.filter or "all entries"))
for (name, ints), (ms, cnt) in sorted(groups.items(), key=lambda kv: -kv[1][0])[: a.top]:
    print("%8.3f ms %4d x %-24s %s" % (ms, cnt, name, ints))

"""Who copies activation-sized tensors through torch (aten::cat / copy_ / clone / _to_copy / index / slice_backward / add) in one
KITTI training step, by call site of this repo, with element counts:   python tools/copy_callers.py [clouds]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curvecloudnet_amd import configs                                     # noqa: E402
from curvecloudnet_amd.model import build_model, segmentation_loss        # noqa: E402
from curvecloudnet_amd.synth import make_batch, to_device                 # noqa: E402

NAMES = ("aten::cat", "aten::copy_", "aten::clone", "aten::_to_copy", "aten::add", "aten::add_", "aten::slice_backward",
         "aten::index", "aten::index_select", "aten::zeros", "aten::zero_", "aten::fill_", "aten::mul", "aten::sum", "aten::contiguous")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()
        self.el = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.name().split(".")[0]
        if name in NAMES and isinstance(out, torch.Tensor) and out.is_cuda and out.numel() >= 1 << 16:
            frames = [f for f in traceback.extract_stack() if f.filename.startswith(ROOT) and "copy_callers" not in f.filename]
            where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(frames[-3:])) or "(autograd engine)"
            self.n[(name, where)] += 1
            self.el[(name, where)] += out.numel()
        return out


clouds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(configs.kitti_config(1.0), 4, 20).to(dev).train()
data = to_device(make_batch(list(range(clouds))), dev)
labels = torch.randint(0, 20, (data.pos.size(0),), device=dev)
torch.autograd.set_multithreading_enabled(False)
for it in range(2):
    log = Log()
    with log:
        torch.manual_seed(1)
        loss = segmentation_loss(model(data), labels)
        loss.backward()
    model.zero_grad(set_to_none=True)
torch.cuda.synchronize()
tot = sum(log.el.values())
print("torch ops on tensors of >= 64 k elements in one step: %d calls, %.1f M elements" % (sum(log.n.values()), tot / 1e6))
for key, e in sorted(log.el.items(), key=lambda kv: -kv[1])[:45]:
    print("%4d x %-22s %9.1f M elems  %s" % (log.n[key], key[0], e / 1e6, key[1]))

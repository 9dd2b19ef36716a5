"""Shapes of the aten device ops the autograd ENGINE itself issues in one bench step (no frame of this repo on the stack):
gradient sums of tensors with two consumers (aten::add), accumulation into .grad (aten::add_), zeros materialised for
unused outputs of multi-output Functions (aten::zeros).
    CCN_BENCH_ENGINE_OPS=1 python bench.py --no-cpu-baseline --no-kernel-timing"""
import collections
import os
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.name().split(".")[0]
        if name in ("aten::add", "aten::add_", "aten::zeros", "aten::neg", "aten::clone", "aten::flip", "aten::mul"):
            frames = [f for f in traceback.extract_stack() if f.filename.startswith(ROOT) and "aten_engine" not in f.filename]
            if frames and os.path.basename(frames[-1].filename) == "bench.py" and isinstance(out, torch.Tensor) and out.is_cuda:
                self.rows[(name, tuple(out.shape))] += 1
        return out


def table(step):
    torch.autograd.set_multithreading_enabled(False)
    log = Log()
    with log:
        step()
    torch.cuda.synchronize()
    for (name, shape), n in sorted(log.rows.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print("%4d x %-12s %s" % (n, name, shape))

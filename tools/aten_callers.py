"""Which lines of this repo issue torch-side (aten) device ops in one bench step, backward included:
    CCN_BENCH_ATEN_TABLE=1 python bench.py --no-cpu-baseline --no-kernel-timing
(bench.py calls table(step) after its warm-up.)  A TorchDispatchMode sees every aten call of the thread it is entered on, so the
autograd engine is kept on the calling thread for the logged step."""
import collections
import os
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("aten::view", "aten::_unsafe_view", "aten::as_strided", "aten::empty", "aten::detach", "aten::alias", "aten::t",
        "aten::transpose", "aten::slice", "aten::select", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::reshape",
        "aten::new_empty", "aten::empty_like", "aten::empty_strided", "aten::permute", "aten::narrow", "aten::split",
        "aten::unbind", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::is_same_size", "aten::sym_size",
        "aten::stride", "aten::size", "aten::numel", "aten::is_pinned", "aten::record_stream", "aten::set_")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.elems = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.name().split(".")[0] if callable(getattr(func, "name", None)) else str(func)
        if name.startswith(SKIP):
            return out
        tensors = [a for a in args if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in tensors) and not (isinstance(out, torch.Tensor) and out.is_cuda):
            return out
        frames = [f for f in traceback.extract_stack() if f.filename.startswith(ROOT) and "aten_callers" not in f.filename]
        where = " <- ".join("%s:%d" % (os.path.relpath(f.filename, ROOT), f.lineno) for f in reversed(frames[-3:]))
        key = (name, where or "(engine)")
        self.rows[key] += 1
        self.elems[key] += max([t.numel() for t in tensors] + ([out.numel()] if isinstance(out, torch.Tensor) else [0]))
        return out


def table(step):
    torch.autograd.set_multithreading_enabled(False)
    log = Log()
    with log:
        step()
    torch.cuda.synchronize()
    print("aten device ops of one step: %d calls" % sum(log.rows.values()))
    for key, n in sorted(log.rows.items(), key=lambda kv: -log.elems[kv[0]])[:90]:
        print("%5d x %-24s %12d elems  %s" % (n, key[0], log.elems[key], key[1]))

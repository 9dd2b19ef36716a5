"""Who still casts fp32 rows to 16-bit rows (ccn_cast_rows_h) in one training step of a 16-bit mode, with sizes:
    python tools/cast_callers.py [nuscenes|a2d2] [bf16|fp16]"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curvecloudnet_amd import configs, ops                    # noqa: E402
from curvecloudnet_amd.model import build_model, segmentation_loss   # noqa: E402
from curvecloudnet_amd.synth import make_batch, to_device   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "nuscenes"
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg, n_out, clouds, curves, mixed = ((configs.nuscenes_config(1.0), 17, 16, 1430, False) if which == "nuscenes" else
                                     (configs.a2d2_config(1.0), 55, 8, 2048, True))
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(cfg, 4, n_out).to(dev).train()
data = to_device(make_batch(list(range(clouds)), n_curves=curves, mixed_lengths=mixed), dev)
labels = torch.randint(0, n_out, (data.pos.size(0),), device=dev)
ops.set_mlp_dtype(mode)
torch.autograd.set_multithreading_enabled(False)
for it in range(2):
    log = collections.Counter()
    elems = collections.Counter()
    inner = ops.call

    def spy(name, *a, **kw):
        if name in ("cast_rows_h", "add_cast_rows_h", "f16_to_bf16_rows", "transpose_cast_h"):
            ints = [v for v in a if isinstance(v, int)]
            frames = [f for f in traceback.extract_stack() if f.filename.startswith(ROOT) and "cast_callers" not in f.filename]
            where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(frames[-5:-1]))
            log[(name, where)] += 1
            elems[(name, where)] += ints[1] * ints[2]
        return inner(name, *a, **kw)
    ops.call = spy
    try:
        torch.manual_seed(1)
        loss = segmentation_loss(model(data), labels)
        loss.backward()
    finally:
        ops.call = inner
torch.cuda.synchronize()
print("%s %s: 16-bit conversion launches of one step: %d, %.1f M elements" % (which, mode, sum(log.values()), sum(elems.values()) / 1e6))
for key, n in sorted(log.items(), key=lambda kv: -elems[kv[0]])[:40]:
    print("%4d x %-18s %9.1f M elems  %s" % (n, key[0], elems[key] / 1e6, key[1]))

"""Which torch-side ops (not libccn_hip kernels) a bench step launches: one step of the KITTI bench workload under
torch.profiler, aggregated by op and input shapes.   PYTHONPATH=. python tools/torch_ops_profile.py"""
import sys
import torch
from torch.profiler import profile, ProfilerActivity
_mode = sys.argv[1:2]
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-timing", "--steps", "2", "--warmup", "2"]
import bench  # noqa: E402

if __name__ == "__main__":
    import runpy
    mode = _mode[0] if _mode else ""
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=(mode in ("h2d", "stacks"))) as prof:
        runpy.run_module("bench", run_name="__main__")
    if mode == "h2d":      # who issues the host-to-device copies: python stacks of aten::_to_copy
        rows = prof.key_averages(group_by_stack_n=6)
        rows = [e for e in rows if e.key in ("aten::_to_copy", "aten::scalar_tensor", "aten::_local_scalar_dense", "aten::full")]
        for e in sorted(rows, key=lambda e: -e.count)[:25]:
            print("%5d x %-28s %s" % (e.count, e.key, " <- ".join(str(f).split("/")[-1] for f in e.stack[:5])))
        raise SystemExit
    if mode == "stacks":   # device time of torch-side ops by the python frames (inside this repo) that issue them
        rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key.startswith("aten::") and e.self_device_time_total > 0]
        total = sum(e.self_device_time_total for e in rows)
        print("torch-side ops: %.2f ms device time over 6 steps" % (total / 1e3))
        for e in sorted(rows, key=lambda e: -e.self_device_time_total)[:70]:
            frames = [str(f) for f in e.stack if "/repo/" in str(f) or "curvecloudnet_amd" in str(f)]
            frames = [f.split("/")[-1] for f in frames][:4] or ["(no python frame: autograd engine / backward thread)"]
            print("%8.2f ms %5d x %-22s %s" % (e.self_device_time_total / 1e3, e.count, e.key[:22], " <- ".join(frames)))
        raise SystemExit
    rows = prof.key_averages(group_by_input_shape=True)
    rows = [e for e in rows if e.key.startswith("aten::") or "Memcpy" in e.key]
    total = sum(e.self_device_time_total for e in rows)
    print("torch-side ops: %.2f ms device time over the profiled steps (2 warm-up + 2 timed + 2 kernel-table steps)" % (total / 1e3))
    rows = sorted(rows, key=lambda e: -e.self_device_time_total)[:40]
    for e in rows:
        print("%9.2f ms  %5d x  %-38s %s" % (e.self_device_time_total / 1e3, e.count, e.key[:38], str(e.input_shapes)[:110]))

"""CPU-baseline calibration (BUILD CONTAINER ONLY -- needs /root/reference).

bench.py's `cpu_baseline` is `kind: "port"`: the oracle (oracle/torch_ref.py), because the reference cannot travel to
the GPU box.  This script times the oracle's restatement NEXT TO the imported reference on the survey's three
curve-convolution shapes (the only part of the hot path the reference can run here), same inputs, same thread count,
so that the port's speed relative to the reference is on record:  python oracle/calibrate_cpu_baseline.py
Output committed as profiles/archive/r02_cpu_baseline_calibration.txt.
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import ref_import, torch_ref as R  # noqa: E402
from curvecloudnet_amd.synth import make_batch  # noqa: E402

SHAPES = [  # (class name, feat dims, x channels, label)  -- SURVEY.md section 6
    ("SymmetricCurve1DConvV2", [4, 32, 32, 32], 1, "V2 [4->(8)32,32,32] k=5 diff+xyz (KITTI step 0)"),
    ("SymmetricCurve1DConvV2", [131, 32, 32, 32], 128, "V2 [131->(262)32,32,32] k=5 (KITTI step 31)"),
    ("SymmetricCurve1DConvFastV1", [131, 128, 128], 128, "V1 [131->(262)128,(256)128] k=5 (ShapeNet step 2)"),
]


def timed(mod, x, pos, batch, p2c, backward, reps=10, warm=2):
    ts = []
    for it in range(warm + reps):
        xi = x.clone().requires_grad_(True)
        t0 = time.perf_counter()
        y = mod(xi, pos, batch, p2c)[0]
        if backward:
            y.square().mean().backward()
        ts.append(time.perf_counter() - t0)
    ts = ts[warm:]
    return 1e3 * statistics.mean(ts), 1e3 * min(ts)


def main():
    if not ref_import.reference_available():
        raise SystemExit("the reference tree is not mounted here: this script only runs in the build container")
    fc, _, _ = ref_import.load_reference()
    torch.manual_seed(1234)
    d = make_batch([0])                      # 1 cloud, 2048 curves, 49 652 points (seed 1234 + 0)
    n = d.pos.size(0)
    print("torch %s, %d threads, %d points, %d curves" % (torch.__version__, torch.get_num_threads(), n, 2048))
    print("%-52s %10s %10s %10s %10s %8s" % ("shape", "ref fwd", "ref f+b", "port fwd", "port f+b", "port/ref"))
    for cls, dims, cx, label in SHAPES:
        x = torch.rand(n, cx)
        ref = getattr(fc, cls)(dims, 5, with_xyz=True, with_diff=True)
        port = getattr(R, cls)(dims, 5, with_xyz=True, with_diff=True)
        port.load_state_dict(ref.state_dict(), strict=True)
        ref.train(); port.train()
        with torch.no_grad():
            err = float((ref(x, d.pos, d.batch, d.curve_idxs)[0] - port(x, d.pos, d.batch, d.curve_idxs)[0]).abs().max())
        rf, _ = timed(ref, x, d.pos, d.batch, d.curve_idxs, False)
        rb, _ = timed(ref, x, d.pos, d.batch, d.curve_idxs, True)
        pf, _ = timed(port, x, d.pos, d.batch, d.curve_idxs, False)
        pb, _ = timed(port, x, d.pos, d.batch, d.curve_idxs, True)
        print("%-52s %8.1f ms %8.1f ms %8.1f ms %8.1f ms %8.2f   (max |ref - port| %.1e)" % (label, rf, rb, pf, pb, pb / rb, err))


if __name__ == "__main__":
    main()

"""FRNN micro-benchmark: fixed-radius kNN at the level shapes of the KITTI model on the synthetic
benchmark clouds; reports ms, Mqueries/s and achieved GB/s against the algorithmic bytes 12 (P1 + P2) + 8 K P1.
usage: python tools/bench_frnn.py [clouds=8] [curves=2048]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from curvecloudnet_amd import ops  # noqa: E402
from curvecloudnet_amd.synth import make_batch, to_device  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
curves = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
data = to_device(make_batch(list(range(B)), n_curves=curves), dev)
topo0 = ops.CurveTopology(data.batch, data.curve_idxs)
idx1 = ops.curve_fps(data.pos, topo0, 0.007, 0.5)               # level 1: what sa-geo keeps
pos1, batch1 = data.pos[idx1], data.batch[idx1]
z1 = torch.zeros_like(batch1)
topo1 = ops.CurveTopology(batch1, z1, B)
idx2 = ops.voxel_fps(pos1, batch1, 0.025)
pos2, batch2 = pos1[idx2], batch1[idx2]
topo2 = ops.CurveTopology(batch2, torch.zeros_like(batch2), B)
idx3 = ops.voxel_fps(pos2, batch2, 0.07)
pos3, batch3 = pos2[idx3], batch2[idx3]
topo3 = ops.CurveTopology(batch3, torch.zeros_like(batch3), B)


def run(name, pq, tq, ps, ts, K, r, reps=20):
    qp, _ = ops.to_batch_padded(pq, tq)
    sp, _ = ops.to_batch_padded(ps, ts)
    for _ in range(3):
        nbr = ops.fast_knn(qp, sp, tq.lengths, ts.lengths, K, r)
    torch.cuda.synchronize()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(reps):
        nbr = ops.fast_knn(qp, sp, tq.lengths, ts.lengths, K, r)
    end.record()
    torch.cuda.synchronize()
    ms = beg.elapsed_time(end) / reps
    from curvecloudnet_amd import _lib
    _lib.PROFILE = []
    for _ in range(reps):
        ops.fast_knn(qp, sp, tq.lengths, ts.lengths, K, r)
    torch.cuda.synchronize()
    per = {}
    for name, _, b_, e_, *_ in _lib.PROFILE:
        per[name] = per.get(name, 0.0) + b_.elapsed_time(e_) / reps
    _lib.PROFILE = None
    p1, p2 = pq.size(0), ps.size(0)
    found = float((nbr >= 0).sum()) / p1
    alg = 12 * (p1 + p2) + 8 * K * p1
    q_ms = per.get("frnn_query", ms)
    print("%-32s P1=%7d P2=%7d K=%2d r=%.3f  wall %6.3f ms | build %6.3f  query %6.3f ms  %7.1f Mq/s  %6.1f GB/s algorithmic"
          "  found/K=%.2f" % (name, p1, p2, K, r, ms, per.get("frnn_grid_build", 0.0), q_ms, p1 / q_ms / 1e3,
                              alg / q_ms / 1e6, found / K))


MODES = {0: "auto", 1: "thread per query", 2: "team of 32 lanes", 3: "team of 64 lanes"}
print("clouds %d x %d curves: level sizes %d / %d / %d / %d points" % (B, curves, data.pos.size(0), pos1.size(0),
                                                                        pos2.size(0), pos3.size(0)))
from curvecloudnet_amd import _lib as _l  # noqa: E402
for mode in (1, 2, 3, 0):
    _l.lib().ccn_frnn_query_mode(mode)
    print("--- query kernel: %s" % MODES[mode])
    run("sgcnn level 1 (self)", pos1, topo1, pos1, topo1, 20, 0.04)
    run("sa level 1 -> 2 (voxel queries)", pos2, topo2, pos1, topo1, 32, 0.04)
    run("sgcnn level 2 (self)", pos2, topo2, pos2, topo2, 20, 0.08)
    run("sa level 2 -> 3", pos3, topo3, pos2, topo2, 32, 0.1)
    run("sgcnn level 3 (self)", pos3, topo3, pos3, topo3, 20, 0.3)
    run("full cloud (self)", data.pos, topo0, data.pos, topo0, 20, 0.04)

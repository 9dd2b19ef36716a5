#!/usr/bin/env python
"""Headline benchmark: point-clouds/s, forward + backward (+ Adam step), synthetic 2048-curve / ~50k-point
clouds through the CurveCloudNet hot path on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One step = one batch of `--clouds-per-gpu` clouds per GPU: forward, mean-NLL loss, backward, gradient
all-reduce (N > 1, RCCL), Adam update.  Rank 0 prints ONE JSON line (contract in the task prompt), with
  roofline      -- the dominant kernel of the timed region, timed live with HIP events on its stream
  cpu_baseline  -- the CPU oracle (oracle/torch_ref.py, kind "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

torch = _lib = None                # bound in main(), after the self-launch decision

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec


def gemm_label(name, ints, nulls=()):
    """(kernel symbol as rocprofv3 prints it, flops) of a GEMM launch; mirrors the dispatch in csrc/ccn_gemm.hip.  The
    product's shape comes from curvecloudnet_amd.costs (ONE table of the entries' argument orders).  The transforming /
    accumulating / implicit-convolution entries (gemm_nt_xf, gemm_tn_ws_xf, gemm_nt_acc, conv_rows_*) are listed under
    their entry names: they are other instantiations than the plain product the `roofline` object samples."""
    from curvecloudnet_amd import costs
    shape = costs.gemm_shape(name, ints)
    if shape is None:
        return None, 0.0
    m, n, k = shape
    flops = 2.0 * m * n * k
    if name in ("gemm_nt_xf", "gemm_tn_ws_xf", "gemm_nt_acc", "gemm_nt_red", "conv_rows_nt", "conv_rows_tn", "conv_rows_nt_h", "conv_rows_tn_h"):
        return name, flops
    ld_a, ld_b = ints[0], ints[1]
    if name in ("gemm_nt_h", "gemm_nt_h_stats", "gemm_nt_h_bnact"):
        # (16-bit rows: the LDS-DMA kernels of csrc/ccn_gemm_h.hip; <F16, OUT16, DIAG, BN, FUSE> as launch_nt_h / launch_nt_h_fused
        # pick them: 128 x 64 tiles for an fp32 result of width <= 64 or with a last 128-wide tile at most half used; FUSE 1 = the
        # statistics pass, 2 = BatchNorm + activation in the epilogue)
        f16, out16, fuse = ((ints[6], ints[7], 0) if name == "gemm_nt_h" else (ints[5], 0, 1) if name == "gemm_nt_h_stats"
                            else (ints[8], ints[9], 2))
        narrow = not out16 and (n <= 64 or (n % 128 != 0 and n % 128 <= 64))
        return "gemm_h_pair_kernel<%s, %s, false, %d, %d>" % ("true" if f16 else "false", "true" if out16 and not narrow else "false",
                                                              64 if narrow else 128, fuse), flops
    if name in ("gemm_tn_h", "gemm_tn_h_xf16"):
        return "gemm_h_tn_kernel", flops
    if name == "gemm_tn_ws":
        def tile(d):        # (a small remainder over 128 goes to a second, 64-wide launch: the label is the main one's)
            return 64 if d <= 64 else 128
        if ld_a % 4 == 0 and ld_b % 4 == 0 and n > 32 and k > 32 and m >= 1024:
            split = min(max(1, (512 + ((n + tile(n) - 1) // tile(n)) * ((k + tile(k) - 1) // tile(k)) - 1)
                            // (((n + tile(n) - 1) // tile(n)) * ((k + tile(k) - 1) // tile(k)))), max(1, (m + 31) // 32 // 4))
            return "gemm_tn_glds_kernel<%d, %d, %d>" % (tile(n), tile(k), 0 if split > 1 else 2), flops
        name = "gemm_tn"
    if name == "gemm_nt_f16":
        return "gemm_bf16_kernel<128, %d, 4, true>" % (32 if n <= 32 else (64 if n <= 64 else 128)), flops
    if name == "gemm_nt_x3":
        if k % 32 == 0 and k >= 64 and n > 64 and ((m + 127) // 128) * ((n + 127) // 128) >= 128:
            return "gemm_x3_lean_kernel", flops
        if k % 32 == 0 and k >= 64 and ((m + 255) // 256) * ((n + 127) // 128) >= 512:
            return "gemm_x3_persistent_kernel<%d>" % (32 if n <= 32 else (64 if n <= 64 else 128)), flops
        return "gemm_x3_kernel<%s>" % ("32, 4" if n <= 32 else ("64, 2" if n <= 64 else "128, 2")), flops
    if name == "gemm_tn_bf16":
        bm, bn = (64 if n <= 64 else 128), (64 if k <= 64 else 128)
        return "gemm_bf16_tn_kernel<%d, %d, %d>" % (bm, bn, bm // 32), flops
    if name == "gemm_nt_bf16":
        return "gemm_bf16_kernel<128, %d, 4, false>" % (32 if n <= 32 else (64 if n <= 64 else 128)), flops
    aligned = ld_a % 4 == 0 and ld_b % 4 == 0
    if name == "gemm_nt":
        bn = 32 if n <= 32 else (64 if n <= 64 else 128)
        base_ok = aligned and m >= 1024 and k >= 64
        if (base_ok and n > 64 and os.environ.get("CCN_GEMM_DMA") != "4"
                and ((m + 127) // 128) * ((n + 127) // 128) >= 128):
            return "gemm_glds_pair_kernel", flops
        if base_ok and ((m + 255) // 256) * ((n + 127) // 128) >= 512:
            if k % 32 == 0:
                return "gemm_glds_persistent_kernel<%d, 3>" % bn, flops
            return "gemm_glds_kernel<%d>" % bn, flops
        kern = "gemm_fast_kernel" if aligned else "gemm_kernel"
        return "%s<128, %d, 4, 0, 0, 0%s>" % (kern, bn, (", true" if k > 96 else ", false") if aligned else ""), flops
    if name == "gemm_nn":
        bn = 32 if k <= 32 else (64 if k <= 64 else 128)
        return "gemm_fast_kernel<128, %d, 4, 0, 1, 0>" % bn, flops
    tile = "32, 128, 1" if n <= 32 else ("64, 64, 2" if k <= 64 else ("128, 128, 4" if n > 64 and m >= 50000 else "64, 128, 2"))
    return "gemm_fast_kernel<%s, 1, 1, 1, true>" % tile, flops


def gemm_bytes(name, ints):
    """Algorithmic HBM bytes of a 16-bit-storage GEMM launch (operands read once, result written once):
    ccn_gemm_nt_h reads A (2 B) and the weight (2 B), writes Y (4 B, or 2 B for a 16-bit result); ccn_gemm_tn_h reads dY and
    X (2 B each) and adds into dW (4 B read + 4 B written)."""
    if name == "gemm_nt_h":
        _, _, _, m, n, k, _, out16 = ints[:8]
        return m * (2.0 * k + (2.0 if out16 else 4.0) * n) + 2.0 * n * k
    if name == "gemm_nt_h_stats":          # (lda, ldw, M, N, K, f16): operands read, nothing written
        m, n, k = ints[2:5]
        return 2.0 * (m * k + n * k)
    if name == "gemm_nt_h_bnact":          # (lda, ldw, act, ldz, ldt, M, N, K, f16, out16): + the 16-bit pre-activation when ldt
        m, n, k = ints[5:8]
        return m * (2.0 * k + ((2.0 if ints[9] else 4.0) + (2.0 if ints[4] else 0.0)) * n) + 2.0 * n * k
    if name in ("gemm_tn_h", "gemm_tn_h_xf16"):
        _, _, _, m, n, k = ints[:6]
        return 2.0 * m * (n + k) + 8.0 * n * k
    return 0.0


def pmc_traffic(kernel, tag):
    """HBM bytes per launch of `kernel` from the committed PMC passes of the SAME workload (profiles/*_<tag>_pmc*.json, written by
    tools/collect_profiles.sh + tools/pmc_summary.py: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc
    runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; a third pass holds
    SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE).  Only the NEWEST profile set of the workload counts: None when this exact
    instantiation is not in it (an older round's file describes another kernel -- VERDICT r4 #10: the r04 line quoted r03c)."""
    import glob
    if tag is None:
        return None
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pmc*.json" % tag)), reverse=True)
    for path in paths[:1]:
        table = json.load(open(path))
        hits = [(name, t) for name, t in table.items() if kernel + "(" in name or (kernel + "<" in name and "<" not in kernel)]
        for name, t in sorted(hits, key=lambda kv: -kv[1]["launches"])[:1]:      # (template variants: the most launched)
            out = {"bytes_per_launch": t["fetch_bytes_per_launch_corrected"] + t["write_bytes_per_launch"],
                   "source": os.path.relpath(path, ROOT), "launches_profiled": t["launches"]}
            for k in ("mfma_pipe_utilisation", "effective_clock_ghz"):
                if k in t:
                    out[k] = round(t[k], 4)
            return out
    return None


def empty_bracket_ms(n=64):
    """What a pair of timing events measures around NOTHING on the current stream (median of n): the part of a bracketed
    launch's duration that is the events' own latency (a timing event is a system-scope release on this runtime)."""
    pairs = []
    for _ in range(n):
        b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b.record()
        e.record()
        pairs.append((b, e))
    torch.cuda.synchronize()
    return sorted(b.elapsed_time(e) for b, e in pairs)[n // 2]


def _union_ms(spans):
    spans = sorted(spans)
    if not spans:
        return 0.0
    busy, cur_s, cur_e = 0.0, spans[0][0], spans[0][1]
    for s_, e_ in spans[1:]:
        if s_ > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    return busy + cur_e - cur_s


def step_roofline(records, steps, ms_per_step, ref_event, mfma_peak_tflops, bracket_ms=0.0):
    """What the STEP could be (SURVEY.md section 8d "whole model"): every library launch of `steps` fully instrumented steps
    priced by curvecloudnet_amd.costs -- a launch's floor = max(flops / MFMA peak, algorithmic bytes / HBM peak) -- and
    summed per family next to the measured launch durations.  Launches without a byte model count their MEASURED time
    (the floor is then an upper bound of the true floor, never an understatement); position-only work runs on the side
    stream behind the previous backward pass and is listed but not on the critical path.  `unattributed_ms` = step time
    minus the union of the feature-stream launches = torch's own kernels (gradient sums, cat, memsets) + idle gaps."""
    from curvecloudnet_amd import costs
    fams, crit_spans = {}, []
    for name, ints, beg, end, nulls, *rest in records:
        rows = rest[0] if rest else None
        fam, flops, nbytes, modelled = costs.entry_cost(name, ints, rows)
        ms = max(beg.elapsed_time(end) - bracket_ms, 0.0)
        floor = max(flops / (mfma_peak_tflops * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)) * 1e3 if modelled else ms
        f = fams.setdefault(fam, {"measured_ms": 0.0, "floor_ms": 0.0, "tflop": 0.0, "gbytes": 0.0, "launches": 0,
                                  "unmodelled_ms": 0.0})
        f["measured_ms"] += ms / steps
        f["floor_ms"] += floor / steps
        f["tflop"] += flops / 1e12 / steps
        f["gbytes"] += nbytes / 1e9 / steps
        f["launches"] += 1.0 / steps
        if not modelled:
            f["unmodelled_ms"] += ms / steps
        if fam != "geometry":
            crit_spans.append((ref_event.elapsed_time(beg), ref_event.elapsed_time(end)))
    crit = {k: v for k, v in fams.items() if k != "geometry"}
    floor_ms = sum(v["floor_ms"] for v in crit.values())
    busy = _union_ms(crit_spans) / steps
    out = {"floor_ms": floor_ms, "ms_per_step": ms_per_step, "frac_of_floor": floor_ms / ms_per_step,
           "feature_stream_busy_ms": busy, "unattributed_ms": ms_per_step - busy,
           "peaks": {"mfma_tflops": mfma_peak_tflops, "hbm_gbs": PEAK_HBM_GBS},
           "families": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in sorted(fams.items())},
           "note": "floor = sum over the launches of one step of max(flops / MFMA peak, algorithmic bytes / HBM peak), families "
                   "on the feature stream only (geometry runs on the side stream during the previous backward); launches "
                   "without a byte model are counted at their measured time; from %d fully instrumented steps after the timed "
                   "region" % steps}
    return out


def summarise_profile(records, steps, write_shapes=True, bracket_ms=0.0):
    torch.cuda.synchronize()
    table = {}
    for name, ints, beg, end, nulls, *_ in records:
        label, flops = gemm_label(name, ints, nulls)
        key = label or name
        t = table.setdefault(key, {"ms": 0.0, "launches": 0, "flops": 0.0})
        t["ms"] += max(beg.elapsed_time(end) - bracket_ms, 0.0)
        t["launches"] += 1
        t["flops"] += flops
    total = sum(t["ms"] for t in table.values())
    rows = sorted(table.items(), key=lambda kv: -kv[1]["ms"])
    # per-shape view of the GEMMs (which layers carry the time)
    shapes = {}
    from curvecloudnet_amd import costs
    for name, ints, beg, end, *_ in records:
        shape = costs.gemm_shape(name, ints)
        if shape is not None:
            key = (name,) + shape
            t = shapes.setdefault(key, [0.0, 0])
            t[0] += beg.elapsed_time(end)
            t[1] += 1
    if not write_shapes:
        return rows, total
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gemm_shapes.txt"), "w") as f:
        f.write("GEMM time by (entry, M, N, K) over %d steps\n" % steps)
        for (name, m, n, k), (ms, cnt) in sorted(shapes.items(), key=lambda kv: -kv[1][0])[:60]:
            f.write("%8.2f ms %4d x %-8s M=%-9d N=%-5d K=%-5d %6.1f TFLOP/s\n"
                    % (ms, cnt, name, m, n, k, 2.0 * m * n * k * cnt / (ms * 1e-3) / 1e12))
    return rows, total


NETWORK_NAMES = ("kitti", "nuscenes", "a2d2", "shapenet-seg", "kortx", "hotpath")


def networks():
    """name -> (config builder, in_dim, classes, description)"""
    from curvecloudnet_amd import configs as ref_configs
    return {
        "kitti": (ref_configs.kitti_config, 4, 20, "the reference's kitti-curvecloudnet.yaml model section (33 steps, all levels)"),
        "nuscenes": (ref_configs.nuscenes_config, 4, 17, "the reference's nuscenes-curvecloudnet.yaml model section"),
        "a2d2": (ref_configs.a2d2_config, 4, 55, "the reference's audi-curvecloudnet.yaml model section"),
        "shapenet-seg": (ref_configs.shapenet_seg_config, 3, 50, "the reference's shapenet-seg-curvecloudnet.yaml model section"),
        "kortx": (lambda width=1.0: ref_configs.shapenet_seg_config(width, kortx=True), 3, 50,
                  "the reference's kortx-testsplit-curvecloudnet.yaml model section (k=7 curve convolutions, exact kNN K=30)"),
        "hotpath": (ref_configs.hotpath_config, 4, 20,
                    "section-8a hot-path subset (conv1d-fast-v2, sa-geo, mlp, 2x sgcnn, skip, fp-geo, conv)"),
    }


# BASELINE.json `configs` presets (--baseline-config i): what each one fixes on this command line
BASELINE_PRESETS = {
    0: dict(config="shapenet-seg", clouds_per_gpu=1, curves=85, mixed_lengths=False, mlp_dtype="fp32"),    # ~2048 points
    1: dict(config="kitti", clouds_per_gpu=8, curves=2048, mixed_lengths=False, mlp_dtype="fp32"),
    2: dict(config="nuscenes", clouds_per_gpu=16, curves=1430, mixed_lengths=False, mlp_dtype="bf16"),      # ~35k points
    3: dict(config="kitti", clouds_per_gpu=4, curves=4900, mixed_lengths=False, mlp_dtype="fp32"),          # ~120k points
    4: dict(config="a2d2", clouds_per_gpu=8, curves=2048, mixed_lengths=True, mlp_dtype="fp16"),
}


def workload_label(args):
    """Which BASELINE.json configuration (if any) this command line is."""
    a = args
    if a.width != 1.0:
        return "diagnostic (width x%g): not a BASELINE configuration" % a.width
    if a.config == "kitti" and a.curves == 2048 and a.clouds_per_gpu == 8 and not a.mixed_lengths and a.mlp_dtype == "fp32":
        return ("BASELINE metric shape = configs[1] (batch 8 x 2048-curve / ~50k-point clouds on 1 GPU, fp32 curve-conv + "
                "HIP FRNN) through the reference's full KITTI model section")
    if a.config == "kortx" and not a.mixed_lengths and a.mlp_dtype == "fp32":
        return "BASELINE configs[1] network (kortx-testsplit model section, fp32) at %d clouds/GPU x %d curves" % (
            a.clouds_per_gpu, a.curves)
    if a.config == "nuscenes" and a.mlp_dtype == "bf16":
        return "BASELINE configs[2] (nuScenes model section, bf16 MLP MFMA path) at %d clouds/GPU x %d curves" % (
            a.clouds_per_gpu, a.curves)
    if a.config == "kitti" and a.curves >= 4000 and a.mlp_dtype == "fp32":
        return "BASELINE configs[3] per-GPU shape (KITTI model section, %d clouds/GPU x %d curves ~ 120k points)" % (
            a.clouds_per_gpu, a.curves)
    if a.config == "a2d2" and a.mixed_lengths:
        return "BASELINE configs[4] (A2D2 model section, mixed curve lengths, %s products) at %d clouds/GPU" % (
            a.mlp_dtype, a.clouds_per_gpu)
    if a.config == "shapenet-seg":
        return "BASELINE configs[0] network (shapenet-seg model section) at %d clouds/GPU x %d curves" % (
            a.clouds_per_gpu, a.curves)
    return "diagnostic: %s model section at %d clouds/GPU x %d curves%s, %s (not a BASELINE configuration)" % (
        a.config, a.clouds_per_gpu, a.curves, " mixed lengths" if a.mixed_lengths else "", a.mlp_dtype)


def make_input(cloud_ids, in_dim, args, curves=None):
    """Synthetic clouds in the layout the reference's datasets hand to the model: LiDAR sets carry reflectance in x
    (in_dim 4); the object sets have x=None, positions normalised to the unit ball and a category id per cloud."""
    from curvecloudnet_amd.synth import make_batch
    data = make_batch(cloud_ids, n_curves=curves or args.curves, mixed_lengths=args.mixed_lengths)
    if in_dim == 3:
        data.x = None
        data.pos = data.pos / 3.0
        data.labels = torch.arange(len(cloud_ids)) % 16
    return data


def cpu_baseline(cfg, in_dim, n_classes, args, seed, points_per_cloud):
    """The CPU oracle (oracle/torch_ref.py, kind "port") timed on this host's cores on a BOUNDED sample of the same
    workload: ONE cloud of `--cpu-curves` curves (default a quarter of the benchmark cloud), same model section, same
    width, full fp32.  Legs: forward only, and forward + backward + Adam (the metric's step); 1 warm-up + 3 timed
    passes each, min and median reported.  `value` is scaled LINEARLY in the point count to the benchmark's cloud size
    (the oracle's neighbour searches are exhaustive, i.e. super-linear: linear scaling flatters the CPU)."""
    import copy
    import statistics
    from oracle import torch_ref as R
    kw = {k: v for k, v in copy.deepcopy(cfg).items() if k != "type"}
    torch.manual_seed(seed)
    model = R.ModelBase(in_dim, n_classes, **kw).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    data = make_input([0], in_dim, args, curves=args.cpu_curves)
    n = data.pos.size(0)
    labels = torch.randint(0, n_classes, (n,), generator=torch.Generator().manual_seed(1))
    t_all = time.perf_counter()
    fwd, full = [], []
    for it in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        with torch.no_grad():
            model(data)
        fwd.append(time.perf_counter() - t0)
    for it in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = R.segmentation_loss(model(data), labels)
        loss.backward()
        opt.step()
        full.append(time.perf_counter() - t0)
    fwd, full = fwd[1:], full[1:]
    scale = n / float(points_per_cloud)           # fraction of a benchmark-size cloud per sample cloud
    return {"value": scale / min(full), "unit": "clouds/s", "cores": torch.get_num_threads(), "kind": "port",
            "value_median": scale / statistics.median(full),
            "forward_only": {"value": scale / min(fwd), "value_median": scale / statistics.median(fwd), "unit": "clouds/s"},
            "sample_step_s": {"fwd_bwd_adam": [round(t, 3) for t in full], "fwd": [round(t, 3) for t in fwd]},
            "sample": "oracle/torch_ref.py ModelBase (%s, width x%g), 1 cloud of %d curves = %d points per step; legs: "
                      "forward only, forward+backward+Adam; 1 warm-up + %d timed passes each; value = (sample points / "
                      "%d benchmark points per cloud) / min step time, median alongside (%.0f s of CPU work in all)"
                      % (args.config, args.width, args.cpu_curves, n, args.cpu_steps, points_per_cloud,
                         time.perf_counter() - t_all)}


def knn_bit_match(model, plan):
    """The metric's boolean ("kNN idx bit-match", BASELINE.json / SURVEY section 8d), outside the timed region: the FRNN neighbour
    table of the first SGCNN level (ref src/models/utils/point_ops.py:459 fast_knn as dgcnn.py:163 calls it) of ONE cloud of the
    batch the last step prepared, against the exhaustive CPU search (oracle/frnn_bruteforce.c through oracle.torch_ref -- the
    checker; FRNN's own source is absent from the reference tree, so this is a match against its published semantics: strict
    d2 < r^2, ascending (d2, index), -1 padding)."""
    from curvecloudnet_amd.steps import SGCNNLayer
    from oracle import torch_ref as R
    _, tables, _ = plan
    torch.cuda.synchronize()
    for step, st in zip(model.steps, tables):
        if isinstance(step, SGCNNLayer) and st is not None and hasattr(st[0], "nbr"):
            g = st[0]
            n0 = int(g.topo.lengths[0])
            pos0 = g.out[0][:n0].detach().float().cpu().contiguous()
            got = g.nbr[0, :n0].cpu()
            radius = 0.25 if step.r is None else float(step.r)
            ln = torch.tensor([n0], dtype=torch.int64)
            t0 = time.perf_counter()
            want = R.frnn_bruteforce(pos0[None], pos0[None], ln, ln, step.k, radius)[0]
            return {"knn_idx_bit_match": bool(torch.equal(got, want)),
                    "knn_checked": {"P": n0, "K": int(step.k), "r": radius, "cloud": "first cloud of the last prepared batch",
                                    "level": "first sgcnn step (points after the sa-geo sampling)",
                                    "neighbours_found": int((want >= 0).sum()), "cpu_search_s": round(time.perf_counter() - t0, 2),
                                    "against": "oracle/frnn_bruteforce.c (exhaustive; FRNN itself is not vendored in the reference: "
                                               "parity with its published semantics, unpinned)"}}
    return {"knn_idx_bit_match": None, "knn_checked": None}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a torch.distributed launcher around it: start the N rank processes
    here, as FRESH child processes (`python -m torch.distributed.run ... bench.py <same arguments>`), before this process
    has made any GPU call -- it never does: it waits for the children and exits with their return code.  (A process that
    has initialised the GPU must not exec or fork workers on this pool.)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("GPU_MAX_HW_QUEUES", "8")                 # see main(): one hardware queue per stream in use
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)       # (a multiple of 8: every launch site of the dominant kernel is timed equally often)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--event-stride", type=int, default=8,
                    help="inside the timed region every n-th launch of the dominant kernel is bracketed by HIP events")
    ap.add_argument("--baseline-config", type=int, choices=sorted(BASELINE_PRESETS), default=None,
                    help="set --config / --clouds-per-gpu / --curves / --mixed-lengths / --mlp-dtype to BASELINE.json "
                         "configs[i] (explicit flags given alongside still win)")
    ap.add_argument("--config", choices=sorted(NETWORK_NAMES), default=None,
                    help="kitti (default): the reference's full KITTI model (33 steps, 28.8 M parameters); "
                         "hotpath: the section-8a subset without the voxel/FPS levels; the others: the remaining "
                         "shipped model sections")
    ap.add_argument("--clouds-per-gpu", type=int, default=None)
    ap.add_argument("--curves", type=int, default=None, help="curves per cloud (2048 ~ 50k points; 4900 ~ 120k)")
    ap.add_argument("--mixed-lengths", action="store_true", default=None,
                    help="log-normal curve lengths (BASELINE configs[4])")
    ap.add_argument("--width", type=float, default=1.0)
    ap.add_argument("--mlp-dtype", choices=["fp32", "bf16", "fp16", "bf16x3"], default=None,
                    help="bf16 / fp16: forward / data-gradient / weight-gradient products of the MLP and conv layers on "
                         "the 16-bit MFMA path (BASELINE configs 2 and 4); the headline metric is quoted in fp32")
    ap.add_argument("--graph", action="store_true",
                    help="forward-only throughput with the feature pass of a prepared plan captured in a hipGraph "
                         "(BASELINE configs[4]); prints eager and replayed rates")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="compute each step's sampling / neighbour search inside its own forward instead of during the "
                         "previous step's backward pass")
    ap.add_argument("--vary-batch", type=int, default=4,
                    help="number of DIFFERENT batches rotated through the steps (default 4: every step sees other clouds, i.e. "
                         "another point count, other sample / edge counts and other allocation sizes, as a training loop does -- "
                         "ref src/run/kitti_seg.py:30-38); 1 = one batch reused for every step (the rounds 1-5 line)")
    ap.add_argument("--no-knn-check", action="store_true",
                    help="skip the metric's boolean: the first SGCNN level's FRNN table of one cloud against the exhaustive CPU search")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-curves", type=int, default=None,
                    help="curves of the CPU baseline's sample cloud (default: a quarter of --curves)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed passes per CPU-baseline leg")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-second-line", action="store_true",
                    help="skip the bf16x3 second line the default (fp32 headline, 1 GPU) run appends as `second_line`")
    ap.add_argument("--no-third-line", action="store_true",
                    help="skip BASELINE configs[2] (nuScenes section, bf16 storage path) appended as `third_line` by the default run")
    args = ap.parse_args()
    if args.no_second_line:
        args.no_third_line = True           # (the A/B scripts pass --no-second-line to time the headline alone)
    preset = BASELINE_PRESETS.get(args.baseline_config, {})
    defaults = dict(config="kitti", clouds_per_gpu=8, curves=2048, mixed_lengths=False, mlp_dtype="fp32")
    for k, dflt in defaults.items():
        if getattr(args, k) is None:
            setattr(args, k, preset.get(k, dflt))
    if args.cpu_curves is None:
        args.cpu_curves = max(16, args.curves // 4)
    return args


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))

    # Streams in play per rank: the feature stream (default), the geometry stream, RCCL's own (the weight-gradient side stream is
    # retired since round 3: CCN_WGRAD_STREAM=1 brings it back for A/B runs only).  The ROCm runtime maps streams
    # onto GPU_MAX_HW_QUEUES hardware queues (default 4): with a fifth stream two of them share a queue and serialise --
    # measured with a one-rank RCCL group: the geometry stream of the NEXT batch waited behind the whole backward pass
    # (prepare() 18 -> 105 ms of host time, 66.4 -> 61.3 clouds/s).  Must be set before the runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    global torch, _lib                      # (module-level helpers use them; the self-launching parent never imports torch)
    import torch
    from curvecloudnet_amd import _lib, ops
    from curvecloudnet_amd.model import ModelBase, segmentation_loss
    from curvecloudnet_amd.parallel import FlatAdam, GradientAllReduce, init_process_group_from_env
    from curvecloudnet_amd.synth import to_device

    # CCN_DIST_BACKEND=gloo + CCN_FORCE_DEVICE=0 rehearse the N>1 code path with several ranks on ONE GPU
    rank, world, local_rank = init_process_group_from_env(backend=os.environ.get("CCN_DIST_BACKEND"))
    local_rank = int(os.environ.get("CCN_FORCE_DEVICE", local_rank))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if os.environ.get("CCN_BENCH_FAIL_RANK") == str(rank):      # test hook: the launcher must exit non-zero with its rank
        raise SystemExit("rank %d: failing on request (CCN_BENCH_FAIL_RANK)" % rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if os.environ.get("CCN_BENCH_MAIN_PRIORITY"):           # experiment: the whole step on a stream of this priority (torch: lower = first)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["CCN_BENCH_MAIN_PRIORITY"])))
    result = run(args, rank, world, local_rank, dev)
    if result is None:
        return
    # The bf16x3 SECOND LINE (VERDICT r4 #3a): the same workload, the same step, with the forward / data-gradient / weight-
    # gradient products of the MLP stack assembled from 3-way bf16 splits on the bf16 matrix cores (fp32-grade results:
    # tests/test_gpu_gemm_x3.py and the x3 legs of tests/test_gpu_model.py).  Run in THIS process after the headline, so that the
    # driver's record carries it; the headline (`value`, `roofline`) stays the fp32 MFMA line.
    if (world == 1 and args.mlp_dtype == "fp32" and not args.no_second_line and not args.graph
            and workload_label(args).startswith("BASELINE metric shape")):
        import copy
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        args2 = copy.copy(args)
        args2.mlp_dtype, args2.no_cpu_baseline = "bf16x3", True
        second = run(args2, rank, world, local_rank, dev, quiet=True)
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "kernel_time_ms_per_step")
        result["second_line"] = dict({k: second[k] for k in keep if k in second},
                                     workload="the headline's workload and step, --mlp-dtype bf16x3",
                                     loss=second["config"]["loss"],
                                     device_mallocs_in_timed_region=second["config"]["device_mallocs_in_timed_region"])
        ops.set_mlp_dtype(args.mlp_dtype)
        # The THIRD LINE (VERDICT r5 Next 4): BASELINE configs[2] -- the nuScenes model section, 16 x ~35 k-point clouds, bf16 MLP
        # MFMA path (16-bit storage, BatchNorm layers without their fp32 intermediate) -- so that the driver's record carries it.
        if not args.no_third_line:
            gc.collect()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(dev)
            args3 = copy.copy(args)
            for k, v in BASELINE_PRESETS[2].items():
                setattr(args3, k, v)
            args3.no_cpu_baseline, args3.steps, args3.warmup = True, min(args.steps, 12), min(args.warmup, 3)
            third = run(args3, rank, world, local_rank, dev, quiet=True)
            result["third_line"] = dict({k: third[k] for k in keep if k in third},
                                        workload=third["config"]["workload"], loss=third["config"]["loss"],
                                        device_mallocs_in_timed_region=third["config"]["device_mallocs_in_timed_region"])
            ops.set_mlp_dtype(args.mlp_dtype)
    print(json.dumps(result))


def run(args, rank, world, local_rank, dev, quiet=False):
    """One measured line: build the model for `args`, prime, warm up, time `--steps` steps; -> the result object on rank 0
    (None on the other ranks and for the diagnostic modes that print their own output).  quiet: no files under gpurun_out/."""
    from curvecloudnet_amd import ops
    from curvecloudnet_amd.model import ModelBase, segmentation_loss
    from curvecloudnet_amd.parallel import FlatAdam, GradientAllReduce
    from curvecloudnet_amd.synth import to_device

    ops.set_mlp_dtype(args.mlp_dtype)
    make_cfg, in_dim, n_classes, net_desc = networks()[args.config]
    cfg = make_cfg(width=args.width)
    kw = {k: v for k, v in cfg.items() if k != "type"}
    torch.manual_seed(1234)                                  # identical replicas on every rank
    model = ModelBase(in_dim, n_classes, **kw).to(dev).train()
    sync = GradientAllReduce(model)
    opt = FlatAdam(sync, lr=1e-3)                            # torch.optim.Adam arithmetic, one launch per bucket

    b = args.clouds_per_gpu
    # weak scaling: a fixed number of whole clouds per GPU.  --vary-batch V: V different batches (batch v of rank r = clouds
    # (v * world + r) * b ...), all resident in HBM before the timed region, taken in turn by the steps
    n_batches = 1 if args.graph else max(1, args.vary_batch)
    batches = []
    for v in range(n_batches):
        cloud_ids = list(range((v * world + rank) * b, (v * world + rank) * b + b))
        d = to_device(make_input(cloud_ids, in_dim, args), dev)
        lab = torch.randint(0, n_classes, (d.pos.size(0),), generator=torch.Generator().manual_seed(rank + 1000 * v)).to(dev)
        batches.append((d, lab))
    data, labels = batches[0]
    n_points = data.pos.size(0)
    points_by_batch = [d.pos.size(0) for d, _ in batches]

    if args.graph:
        # BASELINE configs[4] leg: forward (inference) over a prepared plan, launched kernel by kernel vs replayed from ONE
        # captured hipGraph.  Secondary line (the headline metric is fwd+bwd); single rank.
        from curvecloudnet_amd.graph import CapturedForward
        model.eval()
        torch.manual_seed(7)
        cap = CapturedForward(model, data)

        def timed(fn):
            for _ in range(args.warmup):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.steps
        t_eager, t_graph = timed(cap.eager), timed(cap.replay)
        same = bool(torch.equal(cap.eager(), cap.replay()))
        # ... and the WHOLE forward -- sampling, neighbour searches, index tables and features -- from one graph, the
        # data-dependent counts kept on the device (graph.CapturedWholeForward; one flag read-back per replay), against the
        # ordinary forward over the same batch (one host read-back per count)
        from curvecloudnet_amd.graph import CapturedWholeForward
        del cap
        torch.cuda.empty_cache()
        torch.manual_seed(7)
        whole = CapturedWholeForward(model, data)
        t_weager, t_wgraph = timed(whole.eager), timed(whole.replay)
        t_wbounded = timed(whole.bounded_eager)        # the captured computation launched kernel by kernel (no read-back either)
        w_gap = float((whole.replay() - whole.reference).abs().max())
        w_scale = max(1.0, float(whole.reference.abs().max()))
        whole_line = {"value": b / t_wgraph, "unit": "clouds/s", "ms_per_step": 1e3 * t_wgraph,
                      "eager": {"value": b / t_weager, "ms_per_step": 1e3 * t_weager}, "graph_speedup": t_weager / t_wgraph,
                      "bounded_eager_ms": 1e3 * t_wbounded,
                      "count_sites": len(whole.caps), "host_readbacks_per_replay": 1, "host_readbacks_per_eager_forward": len(whole.caps),
                      "max_abs_diff_vs_ordinary_forward": w_gap, "logit_scale": w_scale,
                      "note": "geometry + features replayed from one hipGraph; capacities = calibrated counts x 1.0625; the "
                              "batch carries one phantom point (a cloud of its own) that absorbs the slack of every padded list"}
        print(json.dumps({
            "metric": "point-clouds/sec fwd (inference, prepared geometry)", "value": b / t_graph, "unit": "clouds/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_graph, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.mlp_dtype, "data": "synthetic",
            "config": {"workload": "%s; feature pass of a prepared plan replayed from one hipGraph (%d clouds x %d curves, "
                                   "%d points)" % (workload_label(args), b, args.curves, n_points),
                       "network": args.config},
            "eager": {"value": b / t_eager, "ms_per_step": 1e3 * t_eager}, "graph_speedup": t_eager / t_graph,
            "bit_identical_to_eager": same, "whole_forward": whole_line}))
        return

    staged = {"plan": None}
    if os.environ.get("CCN_BENCH_EXTRA_LAUNCH"):        # experiment: what one more tiny kernel per library call costs
        _one = torch.zeros(4, device=dev)
        _idx = torch.zeros(1, dtype=torch.int64, device=dev)
        _out = torch.zeros(4, device=dev)
        _fn = _lib.lib().ccn_gather_rows
        _reps = int(os.environ["CCN_BENCH_EXTRA_LAUNCH"])

        def _extra():
            for _ in range(_reps):
                _fn(_lib.ptr(_one), 4, _lib.ptr(_idx), 1, 1, _lib.ptr(_out), 4, _lib.stream())
        _lib.EXTRA_LAUNCH = _extra

    debug = os.environ.get("CCN_BENCH_DEBUG") == "1"

    site = [0, 0]          # [launch sites of the dominant kernel seen in this step, steps started]
    timing = [False]       # inside the timed region

    threaded = not args.no_pipeline and os.environ.get("CCN_BENCH_PREPARE_THREAD", "1") != "0"

    step_ends = []          # (batch index, event at the end of the step) of the timed steps: per-step spread without a host sync

    def step():
        data, labels = batches[site[1] % n_batches]            # this step's batch ...
        data_next = batches[(site[1] + 1) % n_batches][0]      # ... and the one whose geometry is prepared meanwhile
        site[0], site[1] = 0, site[1] + 1
        marks = [time.perf_counter()]
        sync.zero_grad()
        plan = staged["plan"]
        if plan is None:
            torch.manual_seed(7)                             # fixes the CurveFPS phase draws
        pending = None
        if threaded and plan is not None:
            # the next batch's sampling / neighbour search: on the side stream, driven by a worker thread whose waits for
            # the element counts overlap this thread's queueing of forward and backward (ModelBase.prepare_async) -- what
            # a training loop does with the loader's next batch
            pending = model.prepare_async(data_next, seed=7)
        loss = segmentation_loss(model(data, plan=plan), labels)
        marks.append(time.perf_counter())
        loss.backward()
        marks.append(time.perf_counter())
        if pending is not None:
            staged["plan"] = pending.result()
        elif not args.no_pipeline:
            # (first step, or CCN_BENCH_PREPARE_THREAD=0: queued from this thread while the backward pass, already queued, runs)
            torch.manual_seed(7)
            staged["plan"] = model.prepare(data_next)
        marks.append(time.perf_counter())
        sync.finish()
        opt.step()
        if timing[0]:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_ends.append(((site[1] - 1) % n_batches, ev))
        if debug:
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
            print("rank %d host ms: forward %.1f backward %.1f prepare %.1f finish+opt+sync %.1f"
                  % ((rank,) + tuple(1e3 * (b_ - a_) for a_, b_ in zip(marks[:-1], marks[1:]))), flush=True)
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if torch.distributed.get_backend() == "nccl":
                torch.distributed.barrier(device_ids=[local_rank])      # RCCL: name the device, no guessing
            else:
                torch.distributed.barrier()
        torch.cuda.synchronize()

    # setup, outside warm-up and timing: the first steps grow the caching allocator's pools (hipMalloc is synchronous and
    # slow); two priming steps keep device allocations out of the timed region whatever --warmup is
    for _ in range(max(2, n_batches + 1 - args.warmup)):     # (every batch at least once before the timed region)
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    # Allocator headroom, still outside the timed region (curvecloudnet_amd.memory.reserve_headroom, the call a training
    # loop makes after its own warm-up): a ladder of free blocks sized from the warm-up peak in the
    # pool of every stream the step allocates on, so that no hipMalloc lands in mid-run.  CCN_BENCH_HEADROOM=0 skips it.
    from curvecloudnet_amd import memory as _memory
    headroom = ({} if os.environ.get("CCN_BENCH_HEADROOM", "1") == "0" else _memory.reserve_headroom(dev))
    barrier()
    if os.environ.get("CCN_BENCH_LAZY_LOG") == "1":             # diagnostics: which layers hand over deferred activations
        from curvecloudnet_amd import ops as _ops
        _ops.LAZY_ACT_LOG = []
        step()
        for row in _ops.LAZY_ACT_LOG:
            print("deferred input: rows %8d  N %5d  K %5d  %s" % (row[0], row[1], row[2], "fused" if row[3] else "written out"))
        return
    if os.environ.get("CCN_BENCH_ENGINE_OPS") == "1":           # diagnostics: what the autograd engine itself launches
        from tools.aten_engine_ops import table as engine_table
        engine_table(step)
        return
    if os.environ.get("CCN_BENCH_ATEN_TABLE") == "1":           # diagnostics: who issues torch-side device ops in a step
        from tools.aten_callers import table
        table(step)
        return
    if not args.no_kernel_timing:
        # inside the timed region only the launches of the DOMINANT kernel are bracketed by HIP events (on the stream they
        # run on): an event pair costs ~3 us of GPU time -- over all ~2300 launches of a step 4 % of the step, over every
        # GEMM launch of both streams, or over all ~130 launches of the dominant kernel, still ~1 % (70.8 vs 71.5 clouds/s)
        dominant = {"fp32": "gemm_glds_pair_kernel",
                    # (16-bit storage: every instantiation of the one NT kernel -- plain, statistics pass, fused epilogue, 16-bit result)
                    "bf16": "gemm_h_pair_kernel<false" if ops.STORE16 else "gemm_bf16_kernel<128, 128",
                    "fp16": "gemm_h_pair_kernel<" if ops.STORE16 else "gemm_bf16_kernel<128, 128",
                    "bf16x3": "gemm_x3_lean_kernel"}[args.mlp_dtype]

        def only_dominant(name, cargs):
            # ... and of those every eighth one, the phase moving on by one launch site each step (eight steps visit
            # every site exactly once, the default 16 steps twice): flops and time are summed over the SAME sampled
            # launches.  A timing
            # event is a system-scope release on this runtime (~10 us of GPU time each): every launch of the kernel
            # bracketed costs 1 % of the step, every third 0.8 %, every eighth 0.1 %
            label = gemm_label(name, tuple(a for a in cargs if isinstance(a, int)),
                               tuple(i for i, a in enumerate(cargs) if a is None))[0]
            if label is None or not label.startswith(dominant):
                return False
            ints = [a for a in cargs if isinstance(a, int)]
            if (name == "gemm_nt" and len(ints) >= 6 and ints[4] > 128 and 0 < ints[4] % 128 <= 64 and cargs[-1] is None):
                return False      # this call also launches a 64-wide remainder product (ccn_gemm.hip): not one kernel
            site[0] += 1
            if (site[0] + site[1]) % args.event_stride == 0:
                sampled_sites.append(site[0])       # (records of this kernel are appended in the same order)
                return True
            return False

        sampled_sites = []
        _lib.PROFILE, _lib.PROFILE_ONLY, _lib.PROFILE_FILTER = [], "gemm_", only_dominant
    ref_event = torch.cuda.Event(enable_timing=True)
    ref_event.record()
    mallocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    seg0 = {k: torch.cuda.memory_stats(dev).get("segment.%s_pool.allocated" % k, 0) for k in ("large", "small")}
    timing[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    timing[0] = False
    device_mallocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - mallocs0
    if os.environ.get("CCN_BENCH_DEBUG") == "1":
        print("segments allocated in the timed region:",
              {k: torch.cuda.memory_stats(dev).get("segment.%s_pool.allocated" % k, 0) - v for k, v in seg0.items()},
              "reserved GB", torch.cuda.memory_reserved(dev) / 2 ** 30, flush=True)
    records, _lib.PROFILE, _lib.PROFILE_ONLY, _lib.PROFILE_FILTER = _lib.PROFILE, None, None, None
    full_records = None
    if records is not None and world == 1:
        # every launch, two extra steps OUTSIDE the timed region: the complete per-kernel table
        _lib.PROFILE = []
        for _ in range(2):
            step()
        barrier()
        full_records, _lib.PROFILE = _lib.PROFILE, None
    # what a data-parallel run says about itself: the world size the process group reports, the points
    # each rank processed per step (step time = the slowest rank's: load imbalance = max / mean), the bytes handed to
    # all_reduce per step and the host time finish() waited for the collectives
    dp_stats = dict(sync.stats)
    n_points = sum(points_by_batch) // n_batches           # from here on: the mean over the rotated batches
    points_per_rank = [n_points]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
        gathered = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(world)]
        mine = torch.tensor([n_points, dp_stats["finish_wait_s"], dp_stats["buckets_reduced_in_finish"]],
                            dtype=torch.float64, device=dev)
        torch.distributed.all_gather(gathered, mine)
        points_per_rank = [int(g[0].item()) for g in gathered]
        dp_stats["finish_wait_s_max_rank"] = max(float(g[1].item()) for g in gathered)
        dp_stats["buckets_reduced_in_finish_max_rank"] = max(int(g[2].item()) for g in gathered)
    if rank != 0:
        return
    all_steps = max(dp_stats["steps"], 1)        # (priming + warm-up + timed + instrumented steps all count)
    multi_gpu = {
        "world_size": world, "world_size_observed": (torch.distributed.get_world_size() if world > 1 or
                                                     torch.distributed.is_initialized() else 1),
        "backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
        "collective": "all_reduce(SUM) of the fp32 gradient buckets only" if world > 1 else None,
        "points_per_rank": points_per_rank,
        "load_imbalance_max_over_mean": max(points_per_rank) / (sum(points_per_rank) / len(points_per_rank)),
        "gradient_buckets": len(sync.buckets), "gradient_bytes": sync.num_bytes,
        "allreduce_bytes_per_step": dp_stats["bytes_reduced"] / all_steps,
        "finish_wait_ms_per_step": 1e3 * dp_stats.get("finish_wait_s_max_rank", dp_stats["finish_wait_s"]) / all_steps,
        "buckets_reduced_in_finish_per_step": dp_stats.get("buckets_reduced_in_finish_max_rank",
                                                           dp_stats["buckets_reduced_in_finish"]) / all_steps,
    }

    result = {
        "metric": "point-clouds/sec fwd+bwd @50k pts", "value": world * b * args.steps / elapsed, "unit": "clouds/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32",
                  "bf16": ("bf16 products, f32 accumulate; hidden MLP activations, BatchNorm-backward gradients and cast weights "
                           "stored as bf16 rows" if ops.STORE16 else "bf16 products, f32 accumulate / storage"),
                  "fp16": ("fp16 forward products (activations stored as fp16 rows), bf16 gradient products, f32 accumulate"
                           if ops.STORE16 else "fp16 products, f32 accumulate / storage"),
                  "bf16x3": "f32-grade products assembled from 3-way bf16 splits (6 bf16 MFMAs each; weight gradients "
                            "on the f32 MFMA), f32 accumulate / storage"}[args.mlp_dtype], "data": "synthetic",
        "config": {"workload": "%s; %d clouds/GPU x %d curves (~%dk points each, %d points on rank 0); curve-conv + HIP "
                               "FRNN + MFMA MLP stack; network = %s at width x%g; fwd + mean-NLL + bwd + Adam%s"
                               % (workload_label(args), b, args.curves, round(n_points / b / 1000), n_points, net_desc,
                                  args.width, "" if args.no_pipeline else
                                  "; the next step's sampling / neighbour search runs during this step's backward"),
                   "network": args.config, "parameters": sum(p.numel() for p in model.parameters()),
                   "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                   "reserved_hbm_gb": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1),
                   "device_mallocs_in_timed_region": device_mallocs,
                   "allocator_headroom_gb": round(sum(headroom.values()) / 2 ** 30, 1),
                   "clouds_per_gpu": b, "points_per_cloud": n_points // b, "parallelism": "dp%d" % world,
                   "loss": float(loss.detach())},
        "multi_gpu": multi_gpu,
    }
    # --vary-batch: what the rotation looked like -- points of each batch, and the time between the ends of consecutive timed steps
    # (events on the feature stream: no host synchronisation inside the timed region) by the batch the step ran on
    gaps = {}
    for (_, e0), (v1, e1) in zip(step_ends[:-1], step_ends[1:]):
        gaps.setdefault(v1, []).append(e0.elapsed_time(e1))
    all_gaps = sorted(g for v in gaps.values() for g in v)
    result["config"]["batches"] = {
        "different_batches": n_batches, "points_on_rank0": points_by_batch,
        "step_ms_by_batch": {str(v): round(sum(g) / len(g), 2) for v, g in sorted(gaps.items())},
        "step_ms_min_median_max": ([round(all_gaps[0], 2), round(all_gaps[len(all_gaps) // 2], 2), round(all_gaps[-1], 2)]
                                   if all_gaps else None),
        "note": "steps take the batches in turn; the geometry of the NEXT step's batch is prepared during this step's backward"}
    result["config"]["fps_cluster_fallbacks"] = int(ops.fps_fallbacks(dev).item())
    if not args.no_knn_check and not quiet:
        torch.manual_seed(7)
        result.update(knn_bit_match(model, staged["plan"] or model.prepare(batches[0][0])))
    if records:
        bracket = empty_bracket_ms()
        rows, _ = summarise_profile(records, args.steps, write_shapes=not full_records and not quiet, bracket_ms=bracket)
        name, top = rows[0]
        table_rows, total_ms, table_steps = rows, sum(t["ms"] for _, t in rows), args.steps
        if full_records:
            table_rows, total_ms = summarise_profile(full_records, 2, write_shapes=not quiet)
            table_steps = 2
        full_share = dict(table_rows).get(name, {"ms": 0.0})["ms"] / total_ms if total_ms else None
        if top["flops"] > 0:
            # every launch site (one call site of the step = one shape) weighs the same whatever --steps / --event-stride
            # sampled it how often: mean duration per site first, then flops and time summed over the sites seen
            per_site = {}
            if len(sampled_sites) == len(records):
                for st, (rname, ints, beg, end, nulls, *_) in zip(sampled_sites, records):
                    e = per_site.setdefault(st, [0.0, 0, gemm_label(rname, ints, nulls)[1], gemm_bytes(rname, ints)])
                    e[0] += max(beg.elapsed_time(end) - bracket, 0.0)
                    e[1] += 1
            site_bytes = 0.0
            if per_site:
                site_ms = sum(v[0] / v[1] for v in per_site.values())
                site_flops = sum(v[2] for v in per_site.values())
                site_bytes = sum(v[3] for v in per_site.values())
                achieved = site_flops / (site_ms * 1e-3) / 1e12
                top = dict(top, flops_per_launch=site_flops / len(per_site), avg_launch_ms=site_ms / len(per_site),
                           sites=len(per_site))
            else:
                achieved = top["flops"] / (top["ms"] * 1e-3) / 1e12
            label = workload_label(args)
            tr = pmc_traffic(name, "kitti" if label.startswith("BASELINE metric shape") else
                             "c2" if label.startswith("BASELINE configs[2]") else
                             "c4" if label.startswith("BASELINE configs[4]") and args.mlp_dtype == "fp16" else None)
            # the split product spends six bf16 MFMAs per algorithmic multiply-add
            peak = (PEAK_BF16_MFMA_TFLOPS / 6.0 if "x3" in name else
                    PEAK_BF16_MFMA_TFLOPS if ("bf16" in name or "gemm_h_" in name) else PEAK_F32_MFMA_TFLOPS)
            result["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": peak,
                                  "unit": "TFLOP/s", "frac": achieved / peak,
                                  "traffic": tr["bytes_per_launch"] if tr else None, "traffic_source": tr,
                                  "flops_per_launch": top.get("flops_per_launch", top["flops"] / top["launches"]),
                                  "kernel": (name if not dominant.endswith(("<", "<false")) else
                                             "gemm_h_pair_kernel (every instantiation the step launches: plain, statistics pass, "
                                             "fused BatchNorm epilogue, 16-bit result; the most expensive: %s)" % name),
                                  "avg_launch_ms": top.get("avg_launch_ms", top["ms"] / top["launches"]),
                                  "launches": len(records), "launch_sites": top.get("sites"),
                                  "share_of_kernel_time": full_share,
                                  "empty_bracket_us": round(1e3 * bracket, 2),
                                  "launches_note": "every %d-th launch of this kernel inside the timed region is bracketed by "
                                                   "HIP events (sampling keeps the events' own cost out of `value`), the phase moving by one "
                                                   "launch site per step; durations have an empty bracket (empty_bracket_us, measured after the "
                                                   "run) subtracted and are averaged per launch site first, so every site of the step weighs the "
                                                   "same for any --steps" % args.event_stride}
            if site_bytes > 0:
                # the 16-bit storage products are HBM-bound at the network's widths (128 FLOP per byte at K = N = 256 against
                # 2500 / 8 = 312 at the ridge): the binding roof is HBM; the matrix-core figure stays alongside
                gbs = site_bytes / (site_ms * 1e-3) / 1e9
                result["roofline"].update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                           "frac": gbs / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": site_bytes / len(per_site),
                                           "mfma": {"achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak}})
            # With CCN_WGRAD_STREAM=1 (A/B runs) the GEMM launches run on two streams and a
            # launch's own duration includes the time it shares the chip.  All GEMM launches together: flops over the
            # UNION of their execution intervals = the MFMA throughput the step actually gets out of the chip.
            fam_records, fam_steps = (full_records, 2) if full_records else (records, args.steps)
            from curvecloudnet_amd import costs
            spans = sorted((ref_event.elapsed_time(r[2]), ref_event.elapsed_time(r[3])) for r in fam_records
                           if costs.gemm_shape(r[0], r[1]) is not None)
            busy = _union_ms(spans)
            fam_flops = sum(gemm_label(r[0], r[1], r[4])[1] for r in fam_records)
            fam = fam_flops / (busy * 1e-3) / 1e12
            if world == 1 and os.environ.get("CCN_WGRAD_STREAM", "0") != "0":
                result["roofline"]["note"] = ("CCN_WGRAD_STREAM is on: launch durations include the time this kernel shares the "
                                              "chip with the weight-gradient stream (ops._WgradScope)")
            if full_records and os.environ.get("CCN_BENCH_DUMP_RECORDS"):
                # diagnostics: every launch of the two instrumented steps with its integer arguments, duration and floor
                from curvecloudnet_amd import costs as _costs
                with open(os.environ["CCN_BENCH_DUMP_RECORDS"], "w") as f:
                    for r in full_records:
                        fam, fl, by, ok = _costs.entry_cost(r[0], r[1], r[5] if len(r) > 5 else None)
                        ms = max(r[2].elapsed_time(r[3]) - bracket, 0.0)
                        f.write("%s\t%s\t%.4f\t%.4f\t%.3f\t%.3f\t%s\n" % (
                            r[0], fam, ms, max(fl / (PEAK_F32_MFMA_TFLOPS * 1e12), by / (PEAK_HBM_GBS * 1e9)) * 1e3,
                            fl / 1e9, by / 1e6, ",".join(str(v) for v in r[1])))
            if full_records:
                result["roofline"]["step"] = step_roofline(full_records, 2, 1e3 * elapsed / args.steps, ref_event,
                                                           PEAK_BF16_MFMA_TFLOPS if args.mlp_dtype in ("bf16", "fp16") else
                                                           PEAK_BF16_MFMA_TFLOPS / 6.0 if args.mlp_dtype == "bf16x3" else
                                                           PEAK_F32_MFMA_TFLOPS, bracket)
            result["roofline"]["all_gemm_launches"] = {
                "achieved": fam, "frac": fam / peak, "busy_ms_per_step": busy / fam_steps,
                "note": "flops of every GEMM launch / union of their execution intervals"
                        + ("; from the 2 fully instrumented steps after the timed region" if full_records else "")}
        else:
            result["roofline"] = {"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None,
                                  "traffic": None, "kernel": name, "avg_launch_ms": top["ms"] / top["launches"],
                                  "launches": top["launches"], "share_of_kernel_time": full_share}
        result["kernel_time_ms_per_step"] = total_ms / table_steps
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_kernels%s.txt" % ("_second_line" if quiet else "")), "w") as f:
            f.write("per-kernel time (HIP events on the launch stream), %d steps%s\n"
                    % (table_steps, " after the timed region (inside it only the dominant kernel's launches are timed)" if full_records else ""))
            for k, t in table_rows:
                tf = " %7.1f TFLOP/s" % (t["flops"] / (t["ms"] * 1e-3) / 1e12) if t["flops"] else ""
                f.write("%9.2f ms %5.1f%% %6d launches  %s%s\n" % (t["ms"], 100 * t["ms"] / total_ms, t["launches"], k, tf))
    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(cfg, in_dim, n_classes, args, 1234, n_points // b)
    return result


if __name__ == "__main__":
    main()

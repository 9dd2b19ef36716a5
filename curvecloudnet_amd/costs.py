"""Algorithmic work of a ``libccn_hip.so`` launch, from the integer arguments of the call.

``entry_cost(name, ints, rows)`` -> ``(family, flops, bytes)``: the floating-point operations of the product a GEMM
entry computes, and the bytes every entry HAS to move through HBM when each operand is read once and each result written
once (gathers of rows that fit the L2 count their unique rows; index lists count).  ``bench.py`` sums these over the
launches of a step into the step's floor -- flops / MFMA peak + bytes / HBM peak -- next to the measured time per family
(``roofline.step``), and uses the same table for the per-shape GEMM report, so the argument order of every entry is
written down in ONE place (``include/ccn_hip.h`` is the authority; the tuples below name the integer arguments in its
order).

``ints``: the integer arguments of the call in prototype order (what ``_lib.call`` records).  ``rows``: the row count
of the edge-sized operand for the few entries whose prototype does not carry it (the CSR aggregations: E lives in the
offsets on the device); ``None`` otherwise.

Families: gemm | batchnorm | edge (first layers of the edge MLPs in algebraic form, messages) | aggregation (max /
softmax / mean over neighbours, interpolation) | curve (diff, sequence layout, shifted rows) | geometry (sampling,
searches, index tables: on the side stream) | loss_optim | other.
"""

F32, I64, I32 = 4.0, 8.0, 4.0


def _gemm(m, n, k, a_row_floats=None, el_a=F32, el_w=F32, el_y=F32):
    a_floats = m * (a_row_floats if a_row_floats is not None else k)
    return 2.0 * m * n * k, el_a * a_floats + el_w * n * k + el_y * m * n


# name -> function(ints) -> (M, N, K) of the product in the "rows, output width, contraction" sense of bench.py's tables
GEMM_MNK = {
    "gemm_nt": lambda i: i[3:6], "gemm_nt_acc": lambda i: i[3:6], "gemm_nn": lambda i: i[3:6], "gemm_tn": lambda i: i[3:6],
    "gemm_tn_ws": lambda i: i[3:6], "gemm_nt_bf16": lambda i: i[3:6], "gemm_nt_f16": lambda i: i[3:6],
    "gemm_tn_bf16": lambda i: i[3:6], "gemm_nt_x3": lambda i: i[3:6],
    "gemm_nt_h": lambda i: i[3:6], "gemm_tn_h": lambda i: i[3:6], "gemm_tn_h_xf16": lambda i: i[3:6],
    "gemm_nt_h_stats": lambda i: i[2:5],       # (lda, ldw, M, N, K, f16): the statistics pass, nothing written
    "gemm_nt_h_bnact": lambda i: i[5:8],       # (lda, ldw, act, ldz, ldt, M, N, K, f16, out16): product + BatchNorm + activation
    "gemm_nt_xf": lambda i: i[4:7],            # (lda, a_act, ldw, ldy, M, N, K)
    "gemm_nt_red": lambda i: i[3:6],           # (lda, ldw, ldy, M, N, K, ldyp, act): + the y tile of the previous layer
    "gemm_tn_ws_xf": lambda i: i[4:7],         # (lddy, ldx, x_act, lddw, M, N, K, workspace_bytes)
    "conv_rows_nt": lambda i: i[3:6],          # (lda, ldw, ldy, M, N, K): K = taps x ld over overlapping rows of stride lda
    "conv_rows_tn": lambda i: i[3:6],          # (lddy, ldx, lddw, M, N, K, workspace_bytes)
    "conv_rows_nt_h": lambda i: i[3:6],        # (lda, ldw, ldy, M, N, K, f16, out16): 16-bit sequence rows
    "conv_rows_tn_h": lambda i: i[4:7],        # (lddy, ldx, x_f16, lddw, M, N, K, workspace_bytes)
}


def gemm_shape(name, ints):
    fn = GEMM_MNK.get(name)
    return None if fn is None or len(ints) < 6 else tuple(int(v) for v in fn(ints))


def _gemm_cost(name, ints):
    m, n, k = gemm_shape(name, ints)
    if name == "gemm_nt_h":                    # 16-bit A and W; Y fp32, or 16-bit when out16 (ints[7])
        return _gemm(m, n, k, el_a=2.0, el_w=2.0, el_y=2.0 if len(ints) > 7 and ints[7] else F32)
    if name == "gemm_nt_h_stats":              # A and W read, column sums out
        return 2.0 * m * n * k, 2.0 * (m * k + n * k)
    if name == "gemm_nt_h_bnact":              # A and W read, z written as fp32 or 16-bit rows (out16: ints[9]), + t (16-bit) when ldt
        return _gemm(m, n, k, el_a=2.0, el_w=2.0, el_y=(2.0 if len(ints) > 9 and ints[9] else F32) + (2.0 if ints[4] else 0.0))
    if name in ("gemm_tn_h", "gemm_tn_h_xf16"):  # dW (fp32, read + written) += dY^T X on 16-bit rows: M = rows
        return 2.0 * m * n * k, 2.0 * m * (n + k) + 2 * F32 * n * k
    if name == "conv_rows_nt_h":               # rows overlap: M x lda distinct 16-bit elements of A
        return _gemm(m, n, k, a_row_floats=ints[0], el_a=2.0, el_w=2.0, el_y=2.0 if len(ints) > 7 and ints[7] else F32)
    if name == "conv_rows_tn_h":
        return 2.0 * m * n * k, 2.0 * (m * n + m * ints[1]) + 2 * F32 * n * k
    if name == "conv_rows_nt":                 # rows overlap: M x lda distinct floats of A
        return _gemm(m, n, k, a_row_floats=ints[0])
    if name == "conv_rows_tn":                 # X rows overlap likewise (ldx); dW written
        return 2.0 * m * n * k, F32 * (m * n + m * ints[1] + n * k)
    if name in ("gemm_tn", "gemm_tn_ws", "gemm_tn_ws_xf", "gemm_tn_bf16"):   # dW[N x K] += dY[M x N]^T X[M x K]
        return 2.0 * m * n * k, F32 * (m * n + m * k + 2 * n * k)
    if name == "gemm_nt_red":                  # + y (M x N) read for the fused BatchNorm-backward sums
        f, b = _gemm(m, n, k)
        return f, b + F32 * m * n
    if name == "gemm_nt_acc":                  # Y read and written
        f, b = _gemm(m, n, k)
        return f, b + F32 * m * n
    return _gemm(m, n, k)


def _rc(i, r, c):
    return float(i[r]) * float(i[c])


# name -> (family, function(ints, rows) -> bytes)
_TABLE = {
    # ---- BatchNorm + activation: (…, rows, C, …)
    "bn_act_fwd": ("batchnorm", lambda i, r: 2 * F32 * _rc(i, 1, 2)),                    # (ldy, rows, C, act, ldz): y -> z
    "bn_act_fwd_h": ("batchnorm", lambda i, r: (F32 + 2.0) * _rc(i, 1, 2)),
    "bn_act_bwd_reduce": ("batchnorm", lambda i, r: 2 * F32 * _rc(i, 2, 3)),             # (lddz, ldy, rows, C, act): dz, y
    "bn_act_bwd_reduce_h": ("batchnorm", lambda i, r: (F32 + 2.0) * _rc(i, 2, 3)),
    "bn_act_bwd_reduce_weighted": ("batchnorm", lambda i, r: 2 * F32 * _rc(i, 2, 3)),
    "bn_act_bwd_apply": ("batchnorm", lambda i, r: 3 * F32 * _rc(i, 2, 3)),              # dz, y -> dy
    "bn_act_bwd_apply_ex": ("batchnorm", lambda i, r: 3 * F32 * _rc(i, 2, 3)),
    "bn_act_bwd_apply_count": ("batchnorm", lambda i, r: 3 * F32 * _rc(i, 2, 3)),
    "bn_act_bwd_apply_h": ("batchnorm", lambda i, r: (F32 + 2.0 + 2.0) * _rc(i, 3, 4)),   # (dz16, lddz, ldy, rows, C, …)
    # round 6, from the layer's OUTPUT z (or its 16-bit pre-activation): (dz16, lddz, zt, z_pre, ldz, rows, C, act, …): dz (2 / 4 B),
    # z (2 B, 4 when zt == 3) [-> dy (2 B)]
    "bn_act_bwd_reduce_hz": ("batchnorm", lambda i, r: ((2.0 if i[0] else F32) + (F32 if i[2] == 3 else 2.0)) * _rc(i, 5, 6)),
    "bn_act_bwd_apply_hz": ("batchnorm", lambda i, r: ((2.0 if i[0] else F32) + (F32 if i[2] == 3 else 2.0) + 2.0) * _rc(i, 5, 6)),
    "colsum": ("batchnorm", lambda i, r: F32 * _rc(i, 1, 2)),                            # (ldx, rows, C)
    "colstats_weighted": ("batchnorm", lambda i, r: F32 * _rc(i, 1, 2)),
    "bn_finalize": ("batchnorm", lambda i, r: 16.0 * (i[0] / 128.0 + 1) * i[1]),         # (rows, C): partial rows of doubles
    "bn_finalize_n": ("batchnorm", lambda i, r: 16.0 * i[0] * i[2]),                     # (nparts, rows, C)
    "reduce_partials": ("batchnorm", lambda i, r: 8.0 * _rc(i, 0, 1)),                   # (nparts, width)
    "bn_eval_params": ("batchnorm", lambda i, r: 0.0),
    # ---- SGCNN first layer in algebraic form on compact rows: ps = [a | b] (N x 2 Co), E neighbour rows + Ne self rows
    "cg_edge_stats": ("edge", lambda i, r: 2 * F32 * _rc(i, 1, 4) + I32 * i[2]),         # (ldps, N, E, Ne, Co)
    "cg_edge_apply": ("edge", lambda i, r: 2 * F32 * _rc(i, 1, 4) + I32 * i[2] + F32 * (i[2] + i[3]) * i[4]),   # + z rows
    "cg_edge_apply_h": ("edge", lambda i, r: 2 * F32 * _rc(i, 1, 4) + I32 * i[2] + 2.0 * (i[2] + i[3]) * i[4]),
    "cg_edge_bwd_stats": ("edge", lambda i, r: 2 * F32 * _rc(i, 1, 4) + I32 * i[2] + F32 * (i[2] + i[3]) * i[4]),   # + dz rows
    "cg_edge_bwd_stats_h": ("edge", lambda i, r: 2 * F32 * _rc(i, 1, 4) + I32 * i[2] + 2.0 * (i[2] + i[3]) * i[4]),
    "cg_edge_bwd": ("edge", lambda i, r: 4 * F32 * _rc(i, 1, 3) + I32 * i[2] + F32 * _rc(i, 2, 3)),   # (ldps, N, E, Co, …): ps, dps, dz
    # round 5, atomics-free backward: _sums = (ldps, N, E, Ne, Co, dz16, lddz, act, ldpt): ps (unique rows), dz rows, pt written;
    # _gather = (ldps, N, Co, dz16, lddz, act, ldpp): ps, the dz rows once more (in source order: rows = E, the launch's work_rows),
    # two index lists, pp written; _finish = (ldpt, ldpp, N, E, Co, training, lddps): pt, pp read, dps written
    "cg_edge_bwd_sums": ("edge", lambda i, r: 2 * F32 * _rc(i, 1, 4) + I32 * i[2] + (2.0 if i[5] else F32) * (i[2] + i[3]) * i[4]
                         + 2 * F32 * _rc(i, 1, 4)),
    "cg_edge_bwd_gather": ("edge", lambda i, r: 4 * F32 * _rc(i, 1, 2) + ((2.0 if i[3] else F32) * i[2] + 2 * I32) * (r or 0)),
    "cg_edge_bwd_finish": ("edge", lambda i, r: 6 * F32 * _rc(i, 2, 4)),
    # PointNetConv2's first layer, the same three: _sums = (ldpx, ldwp, E, Co, dz16, lddz, act): px rows (<= E unique), dz, 28 B of
    # geometry per edge; _gather = (ldpx, ldwp, Nsrc, Co, dz16, lddz, act, ldpp), rows = E; _finish = (ldpp, Nsrc, E, Co, training, lddpx)
    "pn_edge_bwd_sums": ("edge", lambda i, r: (F32 + (2.0 if i[4] else F32)) * _rc(i, 2, 3) + 28.0 * i[2]),
    "pn_edge_bwd_gather": ("edge", lambda i, r: 3 * F32 * _rc(i, 2, 3) + ((2.0 if i[4] else F32) * i[3] + 24.0) * (r or 0)),
    "pn_edge_bwd_finish": ("edge", lambda i, r: 3 * F32 * _rc(i, 1, 3)),
    "cg_edge_bwd_h": ("edge", lambda i, r: 4 * F32 * _rc(i, 1, 3) + I32 * i[2] + 2.0 * _rc(i, 2, 3)),
    # ---- PointNetConv2 first layer: px (source rows, L2-resident gathers) -> E x Co rows
    "pn_edge_stats": ("edge", lambda i, r: (I64 * 2 + 2 * 12.0) * i[2]),                 # (ldpx, ldwp, E, Co): indices + positions
    "pn_edge_apply": ("edge", lambda i, r: (I64 * 2 + 24.0) * i[2] + F32 * _rc(i, 2, 3)),
    "pn_edge_apply_h": ("edge", lambda i, r: (I64 * 2 + 24.0) * i[2] + 2.0 * _rc(i, 2, 3)),
    "pn_edge_bwd_stats": ("edge", lambda i, r: (I64 * 2 + 24.0) * i[2] + F32 * _rc(i, 2, 3)),
    "pn_edge_bwd_stats_h": ("edge", lambda i, r: (I64 * 2 + 24.0) * i[2] + 2.0 * _rc(i, 2, 3)),
    "pn_edge_bwd": ("edge", lambda i, r: (I64 * 2 + 24.0) * i[2] + F32 * _rc(i, 2, 3)),
    "pn_edge_bwd_h": ("edge", lambda i, r: (I64 * 2 + 24.0) * i[2] + 2.0 * _rc(i, 2, 3)),
    "edge_feat_fwd": ("edge", lambda i, r: 2 * I64 * i[1] + 2 * F32 * _rc(i, 1, 2)),     # (ldx, E, C, ldm): msg E x 2C
    "edge_feat_fwd_h": ("edge", lambda i, r: 2 * I64 * i[1] + 2 * 2.0 * _rc(i, 1, 2)),
    "edge_feat_bwd": ("edge", lambda i, r: 2 * I64 * i[1] + 2 * F32 * _rc(i, 1, 2)),     # (lddm, E, C, lddx)
    "edge_feat_bwd_csr": ("edge", lambda i, r: I64 * i[4] + 2 * F32 * _rc(i, 4, 5)),     # (dm16, lddm, num_dst, N, E, C, lddx)
    "msg_build_fwd": ("edge", lambda i, r: (2 * I64 + 24.0) * i[1] + F32 * i[1] * (i[2] + 3)),      # (ldx, E, C, ldm): E x (C + 3)
    "msg_build_bwd": ("edge", lambda i, r: (2 * I64) * i[1] + F32 * i[1] * (i[2] + 3)),
    "sg_gather_fwd": ("edge", lambda i, r: (I64 + 2 * F32 * i[4]) * i[1] * i[2] * (i[3] + 1)),       # (ldx, B, Nmax, K, C, ldf)
    "sg_gather_bwd": ("edge", lambda i, r: (I64 + 2 * F32 * i[4]) * i[1] * i[2] * (i[3] + 1)),
    # ---- aggregations over neighbours (rows = edge rows of the E x C operands)
    "cg_max_fwd": ("aggregation", lambda i, r: F32 * ((r or 0) + i[1]) * i[2]),          # (ldf, N, C, ldo): f rows -> N rows
    "cg_max_bwd": ("aggregation", lambda i, r: F32 * (i[1] + i[2]) * i[3]),              # (lddo, N, R, C, lddf)
    "cg_max_bwd_h": ("aggregation", lambda i, r: (F32 * i[1] + 2.0 * i[2]) * i[3]),
    "seg_softmax_agg_fwd": ("aggregation", lambda i, r: F32 * (2 * (r or 0) + i[2]) * i[3]),        # (ldm, lda, M, C, ldo)
    "seg_softmax_agg_bwd": ("aggregation", lambda i, r: F32 * (4 * (r or 0) + i[2]) * i[3]),        # msg, att, dmsg, datt + do
    "seg_softmax_agg_bwd_h": ("aggregation", lambda i, r: (F32 * (3 * (r or 0) + i[2]) + 2.0 * (r or 0)) * i[3]),
    "seg_max_fwd": ("aggregation", lambda i, r: F32 * ((r or 0) + i[1]) * i[2]),         # (ldm, M, C, ldo)
    "seg_max_bwd": ("aggregation", lambda i, r: F32 * ((r or 0) + i[1]) * i[2]),
    "seg_wsum_fwd": ("aggregation", lambda i, r: F32 * (2 * (r or 0) + i[2]) * i[3]),
    "seg_wsum_bwd": ("aggregation", lambda i, r: F32 * (4 * (r or 0) + i[2]) * i[3]),
    "interp_fwd": ("aggregation", lambda i, r: (I64 + F32) * _rc(i, 1, 2) + F32 * _rc(i, 1, 3)),    # (ldx, n, k, C, ldy): y n x C
    "interp_bwd_gather": ("aggregation", lambda i, r: 2 * F32 * _rc(i, 1, 2)),                      # (lddy, M, C, lddx)
    "interp_bwd": ("aggregation", lambda i, r: (I64 + F32) * _rc(i, 1, 2) + 2 * F32 * _rc(i, 1, 3)),
    # ---- curve features and the sequence layout
    "diff_concat_fwd": ("curve", lambda i, r: 3 * F32 * _rc(i, 1, 2)),                   # (ldx, n, C, ldo): x -> [x, diff]
    "diff_concat_bwd": ("curve", lambda i, r: 3 * F32 * _rc(i, 1, 2)),
    "gather_rows": ("curve", lambda i, r: 2 * F32 * _rc(i, 1, 2) + I64 * i[1]),          # (lds, m, C, ldd)
    "scatter_rows": ("curve", lambda i, r: 2 * F32 * _rc(i, 1, 2) + I64 * i[1]),
    "scatter_rows_fill": ("curve", lambda i, r: F32 * _rc(i, 1, 2) + F32 * i[4] * i[3] + I64 * i[1]),   # (lds, m, C, ldd, total_rows, …)
    "shift_add_fwd": ("curve", lambda i, r: F32 * _rc(i, 1, 2) * (i[3] + 1)),            # (ldp, rows, Co, taps, ldy)
    "shift_add_bwd": ("curve", lambda i, r: F32 * _rc(i, 1, 2) * (i[3] + 1)),
    "transpose_pad": ("curve", lambda i, r: 2 * F32 * _rc(i, 1, 2)),                     # (ldw, N, K, ldt)
    "im2col_fwd": ("curve", lambda i, r: F32 * _rc(i, 1, 2) * (1 + i[3])),               # (ldx, rows, C, taps, ldcol)
    "im2col_bwd": ("curve", lambda i, r: F32 * _rc(i, 1, 2) * (1 + i[3])),
    "im2col_fwd_h": ("curve", lambda i, r: _rc(i, 1, 2) * (F32 + 2.0 * i[3])),
    "im2col_bwd_h": ("curve", lambda i, r: _rc(i, 1, 2) * (F32 + 2.0 * i[3])),
    "cast_rows_h": ("curve", lambda i, r: 6.0 * _rc(i, 1, 2)),                           # (ldx, rows, C, ldo, f16)
    "add_cast_rows_h": ("curve", lambda i, r: 8.0 * _rc(i, 2, 3)),                       # (lda, ldb, rows, C, ldy): fp32 + bf16 -> bf16
    "transpose_cast_h": ("curve", lambda i, r: 6.0 * _rc(i, 1, 2)),
    "f16_to_bf16_rows": ("curve", lambda i, r: 4.0 * _rc(i, 1, 2)),
    # ---- FRNN (12 (P1 + P2) + 8 K P1 bytes per cloud and call; padded sizes here)
    "frnn_grid_build": ("geometry", lambda i, r: 12.0 * i[0] * i[1] + float(i[2])),                     # (B, P2, grid_bytes)
    "frnn_query": ("geometry", lambda i, r: 12.0 * i[0] * (i[1] + i[3]) + I64 * i[2] * i[0] * i[1]),    # (B, P1, K, P2)
    # ---- loss / optimiser
    "nll_loss_fwd": ("loss_optim", lambda i, r: F32 * _rc(i, 1, 2) + I64 * i[1]),        # (ld, rows, C, ignore)
    "nll_loss_bwd": ("loss_optim", lambda i, r: 2 * F32 * _rc(i, 1, 2) + I64 * i[1]),
    "adam_step": ("loss_optim", lambda i, r: 7 * F32 * i[0]),                            # p, g, m, v read; p, m, v written
}

_GEOMETRY_PREFIXES = ("frnn_", "fps", "curve_fps", "curve_group", "curve_topology", "curve_split", "segment_ptr", "knn_",
                      "ball_query", "voxel_", "rank_keys", "sort_keys", "key_spread", "dense_to_csr", "cg_count", "cg_fill",
                      "interp_inverse", "exclusive_scan", "scatter_flagged", "inverse_lists", "group_owner")


def family_of(name):
    if name in GEMM_MNK:
        return "gemm"
    if name in _TABLE:
        return _TABLE[name][0]
    if name.startswith(_GEOMETRY_PREFIXES):
        return "geometry"
    if name.startswith(("bn_", "col")):
        return "batchnorm"
    if name.startswith(("cg_", "pn_", "sg_edge", "edge_", "msg_")):
        return "edge"
    if name.startswith(("seg_", "sg_", "interp_")):
        return "aggregation"
    return "other"


def entry_cost(name, ints, rows=None):
    """(family, flops, algorithmic bytes, modelled?) of one launch of ``ccn_<name>``.  Entries without a byte model
    (position-only work on the side stream, rarely used forms) return bytes 0 and modelled False: ``bench.py`` then counts their
    MEASURED time into the floor, so the floor is never understated by a missing formula."""
    if name in GEMM_MNK and gemm_shape(name, ints) is not None:
        flops, nbytes = _gemm_cost(name, ints)
        return "gemm", flops, nbytes, True
    if name in _TABLE:
        fam, fn = _TABLE[name]
        try:
            nbytes = float(fn(ints, rows))
        except (IndexError, TypeError):
            return fam, 0.0, 0.0, False
        return fam, 0.0, nbytes, nbytes > 0.0
    return family_of(name), 0.0, 0.0, False

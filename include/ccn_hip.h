/* libccn_hip.so -- C ABI of the MI355X (gfx950) CurveCloudNet hot-path kernels.
 *
 * This is the drop-in boundary.  The reference reaches its native code
 * through Python extension packages; every entry point below names the reference call site
 * (path:line under the upstream tree) whose arithmetic it replaces.  A maintainer binds these
 * with ctypes exactly as curvecloudnet_amd/_lib.py does (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host;
 *   - the caller (the PyTorch caching allocator) owns inputs, outputs and workspaces; the library
 *     allocates nothing and keeps no device state between calls;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never synchronises
 *     the host, and is therefore hipGraph-capturable;
 *   - return value: 0 on success, negative CCN_ERR_* otherwise; ccn_last_error() gives the text
 *     (thread local).  Nothing throws.
 *   - feature matrices are row-major float32 with an explicit leading dimension (elements);
 *   - index tensors that the reference exposes as int64 stay int64; internal tables are int32.
 */
#ifndef CCN_HIP_H
#define CCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped when the prototype of an EXISTING entry changes (additions do not bump it).
 *   1 -> 2 (round 4, recorded in round 5 / ADVICE r4): ccn_edge_feat_bwd_csr gained `int64_t N` in front of `int64_t E`.
 *   2 -> 3 (round 5): ccn_interp_inverse_workspace_bytes(M) became (n, k, M) -- the lists are made by a stable radix sort of the
 *                     n * k slots, whose buffers live in the workspace.
 *   3 -> 4 (round 6): ccn_fps takes (n, workspace, workspace_bytes, fallbacks) instead of a float scratch of n entries: the
 *                     exchange areas of the clustered form no longer alias the streaming form's running minima, which the
 *                     gated one-workgroup fallback of a cluster that gave up needs (ccn_fps_workspace_bytes). */
#define CCN_ABI_VERSION 4

#define CCN_OK 0
#define CCN_ERR_ARG (-1)
#define CCN_ERR_LAUNCH (-2)
#define CCN_ERR_WORKSPACE (-3)

#define CCN_ACT_NONE 0
#define CCN_ACT_RELU 1
#define CCN_ACT_LEAKY 2

const char* ccn_last_error(void);
int ccn_abi_version(void);

/* generic exclusive prefix sum over int32 (CSR offsets from counts); out has n+1 entries, out[n] = total,
 * which is also written to total64 (device, nullable). */
size_t ccn_exclusive_scan_workspace_bytes(int64_t n);
int ccn_exclusive_scan_i32(const int32_t* counts, int64_t n, int32_t* offsets, int64_t* total64, void* ws,
                           size_t ws_bytes, void* stream);

/* ---- A1: src/models/utils/point_ops.py:47-54 batch2ptr ------------------------------------------
 * ids: sorted int64 (n).  starts: capacity n+1, receives [0, s_1, ..., s_{R-1}, n] (the reference's
 * with_ends form; the interior form is starts[1:R]).  run_of (nullable): dense run number of every
 * element.  meta (device int64[2]): {R, number of descending neighbours (reference asserts == 0)}. */
size_t ccn_segment_ptr_workspace_bytes(int64_t n);
int ccn_segment_ptr(const int64_t* ids, int64_t n, int64_t* starts, int32_t* run_of, int64_t* meta, void* ws,
                    size_t ws_bytes, void* stream);

/* ---- A2: point_ops.py:20-44 curveidx_local2global (+ the CSR tables every curve kernel uses) ----
 * batch, p2c: int64 (n), batch ids must be 0..num_clouds-1 and sorted, p2c sorted inside a cloud.
 * glob: int64 (n) global curve id.  cid: int32 (n) dense curve number.  curve_ptr: int32, capacity n+1,
 * first Q+1 entries valid.  cloud_ptr: int64 (num_clouds+1).
 * meta (device int64[4]): {Q, ordering violations, longest cloud, 0}. */
size_t ccn_curve_topology_workspace_bytes(int64_t n, int64_t num_clouds);
int ccn_curve_topology(const int64_t* batch, const int64_t* p2c, int64_t n, int64_t num_clouds, int64_t* glob,
                       int32_t* cid, int32_t* curve_ptr, int64_t* cloud_ptr, int64_t* meta, void* ws,
                       size_t ws_bytes, void* stream);

/* ---- A3: src/models/modules/fast_conv1d.py:190-205 compute_feature_diffs (fused with the concat
 * of fast_conv1d.py:66 / :133): out[:, 0:C] = x, out[:, C:2C] = |mean in-curve finite difference|. */
int ccn_diff_concat_fwd(const float* x, int64_t ldx, const int32_t* cid, int64_t n, int64_t C, float* out,
                        int64_t ldo, void* stream);
int ccn_diff_concat_bwd(const float* x, int64_t ldx, const int32_t* cid, int64_t n, int64_t C, const float* g,
                        int64_t ldg, float* dx, int64_t lddx, void* stream);

/* ---- A4: fast_conv1d.py:173-184 (F.conv1d 'same', zero pad) as shifted-row matrix ---------------
 * col[i, t*C + c] = x[i + t - taps/2, c] if that row exists and lies in the same segment, else 0.
 * seg == NULL: the whole buffer is one sequence (V2's padded buffer, fast_conv1d.py:67-72). */
int ccn_im2col_fwd(const float* x, int64_t ldx, const int32_t* seg, int64_t rows, int64_t C, int64_t taps,
                   float* col, int64_t ldcol, void* stream);
int ccn_im2col_bwd(const float* dcol, int64_t ldcol, const int32_t* seg, int64_t rows, int64_t C, int64_t taps,
                   float* dx, int64_t lddx, void* stream);
/* ... at the boundary of the 16-bit storage modes (round 3): the shifted-row matrix written as bf16 / fp16 rows for the layer's
 * product (ccn_gemm_nt_h), its gradient read as bf16 rows ((taps * C) % 8 == 0; leading dimensions in 16-bit elements). */
int ccn_im2col_fwd_h(const float* x, int64_t ldx, const int32_t* seg, int64_t rows, int64_t C, int64_t taps, void* col,
                     int64_t ldcol, int f16, void* stream);
int ccn_im2col_bwd_h(const void* dcol, int64_t ldcol, const int32_t* seg, int64_t rows, int64_t C, int64_t taps, float* dx,
                     int64_t lddx, void* stream);

/* row gather / scatter (fast_conv1d.py:136-141 x_padded[valid] = x ; x = x_padded[valid]; x[idx]) */
int ccn_gather_rows(const float* src, int64_t lds, const int64_t* index, int64_t m, int64_t C, float* dst,
                    int64_t ldd, void* stream);
int ccn_scatter_rows(const float* src, int64_t lds, const int64_t* index, int64_t m, int64_t C, float* dst,
                     int64_t ldd, int accumulate, void* stream);
/* ccn_scatter_rows into an all-zero destination in ONE pass, for a strictly ascending index (the zero-separated sequences of
 * fast_conv1d.py:48-61 / :115-126 and the adjoint of their gather, :67-74 / :136-143): writes every one of the total_rows x ldd
 * floats of dst -- row index[i] + row_offset <- src row i with the columns C..ldd-1 zero, every other row zero (row_offset:
 * leading halo rows of the destination buffer). */
int ccn_scatter_rows_fill(const float* src, int64_t lds, const int64_t* index, int64_t m, int64_t C, float* dst, int64_t ldd,
                          int64_t total_rows, int64_t row_offset, void* stream);

/* ---- A7: src/models/modules/fps_ops.py:16-39 CurveFPS -------------------------------------------
 * u is the reference's torch.rand(1) draw.  idx_out: capacity n (sorted point indices), count_out: device int64. */
size_t ccn_curve_fps_workspace_bytes(int64_t n);
int ccn_curve_fps(const float* pos, const int32_t* cid, const int32_t* curve_ptr, int64_t n, float spacing, float u,
                  int64_t* idx_out, int64_t* count_out, void* ws, size_t ws_bytes, void* stream);

/* ---- A8: point_ops.py:143-193 radius_1d_group_subset ---------------------------------------------
 * count: budget (float, Q+1; [Q] = candidate reach), offsets (int32, M+1), total (device int64).
 * fill: row/col int64 (total).  p2c is the LOCAL curve id (quirk Q3 of SURVEY.md is reproduced). */
size_t ccn_curve_group_subset_workspace_bytes(int64_t n, int64_t Q, int64_t M);
int ccn_curve_group_subset_count(const float* pos, const int32_t* cid, const int32_t* curve_ptr, const int64_t* p2c,
                                 int64_t n, int64_t Q, const int64_t* idx, int64_t M, float radius, float* budget,
                                 int32_t* offsets, int64_t* total, void* ws, size_t ws_bytes, void* stream);
int ccn_curve_group_subset_fill(const int32_t* cid, const int32_t* curve_ptr, const int64_t* p2c, int64_t n,
                                int64_t Q, const int64_t* idx, int64_t M, const float* budget,
                                const int32_t* offsets, int64_t* row, int64_t* col, void* stream);
/* round 5 (device-side counts): the same with the capacity of row / col given -- a group that would end past `cap`
 * entries is not written.  For a caller that sizes row / col from a bound instead of reading `total` back (the reference
 * reads it back: boolean flattening at point_ops.py:186-193). */
int ccn_curve_group_subset_fill_cap(const int32_t* cid, const int32_t* curve_ptr, const int64_t* p2c, int64_t n,
                                    int64_t Q, const int64_t* idx, int64_t M, const float* budget,
                                    const int32_t* offsets, int64_t* row, int64_t* col, int64_t cap, void* stream);

/* ---- A9: point_ops.py:196-260 knn_1d_group_superset + :344-355 knn_interpolate_1D ----------------
 * nbr: int64 (n, k) position inside idx of the k nearest sampled points on the same curve, ascending
 * distance, -1 padded; weight: float (n, k) = 1 / max(d^2, 1e-16), 0 where padded. */
size_t ccn_curve_group_superset_workspace_bytes(int64_t n);
int ccn_curve_group_superset(const float* pos, const int32_t* cid, int64_t n, const int64_t* idx, int64_t M,
                             int64_t k, int64_t* nbr, float* weight, void* ws, size_t ws_bytes, void* stream);
int ccn_interp_fwd(const float* x, int64_t ldx, const int64_t* nbr, const float* weight, int64_t n, int64_t k,
                   int64_t C, float* y, int64_t ldy, void* stream);
int ccn_interp_bwd(const float* dy, int64_t lddy, const int64_t* nbr, const float* weight, int64_t n, int64_t k,
                   int64_t C, float* dx, int64_t lddx, void* stream);
/* Backward of the interpolations without atomics (knn_interpolate_1D_pytorch3d / knn_interpolate_pytorch3d,
 * src/models/utils/point_ops.py:344-355, 293-341: the autograd of their scatter_add over (y_idx, x_idx)): ccn_interp_inverse
 * turns the (n, k) neighbour table into per-coarse-row lists (inv_ptr int32[M+1], inv_src int32[n*k], inv_w float[n*k], sorted by
 * fine row; den float[n] = the weight sum of every fine row), ccn_interp_bwd_gather sums dX[m] = sum_e dY[src_e] / den[src_e] *
 * w_e in list order: deterministic.  Workspace: ccn_interp_inverse_workspace_bytes(n, k, M). */
size_t ccn_interp_inverse_workspace_bytes(int64_t n, int64_t k, int64_t M);
int ccn_interp_inverse(const int64_t* nbr, const float* weight, int64_t n, int64_t k, int64_t M, int32_t* inv_ptr,
                       int32_t* inv_src, float* inv_w, float* den, void* ws, size_t ws_bytes, void* stream);

/* Inverse of a flat index list (round 5): rows r = 0 .. n-1 name a source src[r] in [0, M) (int32, or int64 when src_is_i64);
 * inv_ptr (M + 1) / inv_row (n): for every source the rows naming it, ascending -- the lists the atomics-free backward of the first
 * edge layers gathers through (ccn_cg_edge_bwd_gather, ccn_pn_edge_bwd_gather; reference: the autograd of x[src] in
 * src/models/modules/dgcnn.py:172-177 and point_conv.py:60-69, an index_add over the same lists in arbitrary order).
 * Deterministic (stable radix sort by source, payload = row).  Workspace: ccn_inverse_lists_workspace_bytes(n, M).  ccn_group_owner: owner[r] = p for the rows
 * grp_ptr[p] <= r < grp_ptr[p + 1] of a grouped row list (r < E). */
size_t ccn_inverse_lists_workspace_bytes(int64_t n, int64_t M);
int ccn_inverse_lists(const void* src, int src_is_i64, int64_t n, int64_t M, int32_t* inv_ptr, int32_t* inv_row, void* ws,
                      size_t ws_bytes, void* stream);
int ccn_group_owner(const int32_t* grp_ptr, int64_t N, int64_t E, int32_t* owner, void* stream);
int ccn_interp_bwd_gather(const float* dy, int64_t lddy, const int32_t* inv_ptr, const int32_t* inv_src, const float* inv_w,
                          const float* den, int64_t M, int64_t C, float* dx, int64_t lddx, void* stream);

/* Symmetric curve convolution on the zero-separated V2 sequence (fast_conv1d.py:60-73) for layers with many more input
 * than output channels, as "product first, shift-add second": P = X W_all^T (W_all: taps*C_out x C_in, one GEMM) and
 * Y[i][co] = bias[co] + sum_tap P[i + tap - taps/2][tap*C_out + co] (rows outside the sequence contribute 0).  bwd: the
 * gather dP[j][tap*C_out + co] = dY[j - tap + taps/2][co].  Replaces the (rows x taps*C_in) shifted-row matrix. */
int ccn_shift_add_fwd(const float* P, int64_t ldp, const float* bias, int64_t rows, int64_t Co, int64_t taps, float* Y,
                      int64_t ldy, void* stream);
int ccn_shift_add_bwd(const float* dY, int64_t lddy, int64_t rows, int64_t Co, int64_t taps, float* dP, int64_t lddp,
                      void* stream);

/* ---- A11: point_ops.py:459 frnn.frnn_grid_points (third_party/FRNN: grid build + radius query) ---
 * points: (B, P, 3) float32 zero padded, lengths int64 (B), r float32 (B).  idx: (B, P1, K) int64, the
 * <=K nearest points2 with d2 < r*r ascending by (d2, index), -1 padded; rows >= lengths1 are -1.
 * dist2 (nullable) same shape float32, -1 padded.  count (nullable): (B, P1) int32 neighbours found. */
size_t ccn_frnn_grid_bytes(int64_t B, int64_t P2);
int ccn_frnn_grid_build(const float* points2, const int64_t* lengths2, const float* r, int64_t B, int64_t P2,
                        void* grid, size_t grid_bytes, void* stream);
int ccn_frnn_query(const float* points1, const int64_t* lengths1, const float* r, int64_t B, int64_t P1, int64_t K,
                   const void* grid, int64_t P2, int64_t* idx, float* dist2, int32_t* count, void* stream);

/* ---- A12: point_ops.py:98-111 / :287-290 dense (B,P1,K) idx -> flat (row, col) edge list ----------
 * counts: int32 per packed query (cloud_ptr1[b] + i).  fill writes row = packed query, col = packed point. */
int ccn_dense_to_csr_count(const int64_t* idx, const int64_t* cloud_ptr1, int64_t B, int64_t P1, int64_t K,
                           int32_t* counts, void* stream);
int ccn_dense_to_csr_fill(const int64_t* idx, const int64_t* cloud_ptr1, const int64_t* cloud_ptr2, int64_t B,
                          int64_t P1, int64_t K, const int32_t* offsets, int64_t* row, int64_t* col, void* stream);

/* ---- A16: torch_geometric.nn.MLP layers as used at src/models/base.py:32,64,90-125 ---------------
 * fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32).  colstats (nullable): double[(ccn_stats_rows(M)+1) * 2*N];
 * gemm_nt writes one partial row {column sums, column sums of squares} of Y per 128-row tile
 * (deterministic, no atomics); ccn_bn_finalize reduces them (using the last 2*N doubles as scratch)
 * into the BatchNorm batch statistics (torch.nn.BatchNorm1d inside PyG MLP; fast_conv1d.py:30,73). */
int64_t ccn_stats_rows(int64_t rows); /* partial-statistics rows a reduction over `rows` rows produces (= ceil(rows/128)) */
int ccn_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                int64_t M, int64_t N, int64_t K, double* colstats, void* stream);
/* bf16 MFMA form of ccn_gemm_nt (BASELINE configs 3 / 5, "bf16 MLP MFMA path"): same arguments and outputs; A and W are
 * read as fp32 and rounded to bf16 (round to nearest even) inside the kernel, products accumulate in fp32
 * (v_mfma_f32_32x32x16_bf16), bias / Y / colstats stay fp32.  Requires 16-byte aligned A, W and lda, ldw % 4 == 0. */
int ccn_gemm_nt_bf16(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                     int64_t M, int64_t N, int64_t K, double* colstats, void* stream); /* Y = A W^T + b */
/* fp32-grade form of ccn_gemm_nt on the bf16 matrix cores ("bf16x3" MLP mode): every operand is split exactly into three
 * bf16 terms and the product assembled from the six leading partial products, accumulated in fp32 (dropped terms
 * < 2^-23 |a b|).  Same arguments and outputs as ccn_gemm_nt plus caller-owned scratch for the split weight
 * (ccn_gemm_x3_workspace_bytes(N, K) bytes, 16-byte aligned).  Requires 16-byte aligned A and lda % 4 == 0. */
int64_t ccn_gemm_x3_workspace_bytes(int64_t N, int64_t K);
int ccn_gemm_nt_x3(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                   int64_t M, int64_t N, int64_t K, double* colstats, void* wsplit, int64_t wsplit_bytes, void* stream);
int ccn_gemm_nn(const float* dY, int64_t lddy, const float* W, int64_t ldw, float* dX, int64_t lddx, int64_t M,
                int64_t N, int64_t K, void* stream);                              /* dX = dY W      */
int ccn_gemm_tn(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                int64_t N, int64_t K, void* stream);
/* Weight gradient on the LDS-DMA pipeline (csrc/ccn_gemm_tn.hip): same product as ccn_gemm_tn (autograd of F.linear /
 * F.conv1d at src/models/modules/fast_conv1d.py:183 and of the PyG MLP layers, base.py:90-125), with caller-owned scratch
 * for the per-workgroup partial tiles (ccn_gemm_tn_workspace_bytes, 16-byte aligned; may be NULL: the partial tiles are
 * then added to dW with fp32 atomics).  With scratch the result is deterministic: every work item stores its partial
 * tile once and a second launch adds the tiles of each output block to dW in chunk order.  Operands that do not
 * qualify (unaligned, leading dimension not a multiple of 4, N or K <= 32, M < 1024) take ccn_gemm_tn. */
size_t ccn_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ccn_gemm_tn_ws(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                   int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream);
/* Y += A W^T (the accumulators start from Y): lets a data-gradient product ADD into a gradient that another consumer of the
 * same activation has already written -- PointNetConv2's message tensor feeds both attend_nn and the softmax aggregation
 * (src/models/modules/point_conv.py:89-92), autograd would otherwise sum two E x C tensors in a separate pass.  Paired
 * LDS-DMA kernel only: ccn_gemm_nt_acc_ok says whether a shape takes it. */
int ccn_gemm_nt_acc_ok(int64_t lda, int64_t ldw, int64_t M, int64_t N, int64_t K);
int ccn_gemm_nt_acc(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t M, int64_t N,
                    int64_t K, void* stream);
/* Y = act(A * scale + shift) W^T + b: ccn_gemm_nt whose A operand is the PRE-normalisation product of the previous layer,
 * with that layer's BatchNorm + activation (ccn_bn_act_fwd: z = act(a * scale[k] + shift[k]), same bits) applied between LDS and
 * the matrix cores -- the activation tensor of a hidden MLP layer (torch_geometric.nn.MLP: lin -> norm -> act -> lin,
 * src/models/base.py:90-125) is then never written.  Paired LDS-DMA kernel only, K % 32 == 0, K <= 1024: ask _ok. */
/* round 4: the data-gradient product dZ = dY Wt^T of a layer whose input was a DEFERRED activation (ccn_gemm_nt_xf), with the
 * BatchNorm-backward column sums of the layer that produced it taken in the epilogue: sum(g), sum(g * xhat), g = dZ *
 * act'(y scale + shift), xhat = (y - mean) rstd, from the dZ tile in registers and the matching tile of y (y_prev, its
 * pre-normalisation output).  par: that layer's 4 x N table (scale | shift | mean | rstd).  sums: as ccn_bn_act_bwd_reduce
 * writes it (2 N totals + ccn_stats_rows(M) partial rows) -- that pass over (dZ, y) is not needed for the layer any more.
 * Shapes: those ccn_gemm_nt_acc_ok accepts.  Replaces autograd of BatchNorm1d inside PyG MLP (base.py:90-125). */
int ccn_gemm_nt_red(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t M, int64_t N,
                    int64_t K, const float* y_prev, int64_t ldyp, const float* par, int act, float slope, double* sums,
                    void* stream);
int ccn_gemm_nt_xf_ok(int64_t lda, int64_t ldw, int64_t M, int64_t N, int64_t K);
int ccn_gemm_nt_xf(const float* A, int64_t lda, const float* a_scale, const float* a_shift, int a_act, float a_slope,
                   const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                   double* colstats, void* stream);
/* dW += dY^T act(X * scale + shift): ccn_gemm_tn_ws with the same transform on its X operand -- the weight gradient of a
 * layer whose input was consumed through ccn_gemm_nt_xf (the activation was never stored).  LDS-DMA kernel only: ask _ok. */
int ccn_gemm_tn_xf_ok(const float* dY, int64_t lddy, const float* X, int64_t ldx, int64_t M, int64_t N, int64_t K);
int ccn_gemm_tn_ws_xf(const float* dY, int64_t lddy, const float* X, int64_t ldx, const float* x_scale, const float* x_shift,
                      int x_act, float x_slope, float* dW, int64_t lddw, int64_t M, int64_t N, int64_t K, void* workspace,
                      size_t workspace_bytes, void* stream);
/* round 5, "split tails": ccn_gemm_nt / _acc / _xf / _red with caller-owned scratch for the paired LDS-DMA kernel (same
 * products, same callers: F.linear inside PyG MLP, src/models/base.py:90-125).  The kernel walks 128 x 128 output tiles in
 * rounds of 512 workgroups; when the last round is at most half full (10 550 x 1024 -> 1024: 664 tiles = 2 rounds for 1.3
 * rounds of work) its tiles are cut into 2..8 parts along K, one per workgroup, the parts' accumulators go to the scratch
 * and the last part of a tile to arrive adds them IN PART ORDER (deterministic) and runs the tile's epilogue.  workspace:
 * ccn_gemm_nt_split_workspace_bytes() bytes, 16-byte aligned, its first 4 KiB zero on first use (the library leaves them
 * zero); one buffer per stream.  NULL / too small: the product runs unsplit, as the entry without _ws does.
 * ccn_gemm_nt_split_parts: the number of parts a launch of that shape would use (1 = no split). */
size_t ccn_gemm_nt_split_workspace_bytes(void);
int ccn_gemm_nt_split_parts(int64_t M, int64_t N, int64_t K, size_t workspace_bytes);
int ccn_gemm_nt_ws(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                   int64_t M, int64_t N, int64_t K, double* colstats, void* workspace, size_t workspace_bytes, void* stream);
int ccn_gemm_nt_acc_ws(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t M, int64_t N,
                       int64_t K, void* workspace, size_t workspace_bytes, void* stream);
int ccn_gemm_nt_red_ws(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t M, int64_t N,
                       int64_t K, const float* y_prev, int64_t ldyp, const float* par, int act, float slope, double* sums,
                       void* workspace, size_t workspace_bytes, void* stream);
int ccn_gemm_nt_xf_ws(const float* A, int64_t lda, const float* a_scale, const float* a_shift, int a_act, float a_slope,
                      const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                      double* colstats, void* workspace, size_t workspace_bytes, void* stream);
/* ---- A4-A6: the symmetric curve convolution as an IMPLICIT GEMM over the row sequence (no shifted-row matrix) ----
 * Replaces F.conv1d(input(1,C,L), weight, bias, 1, 'same') at src/models/modules/fast_conv1d.py:183 (called from
 * SymmetricCurve1DConvV2 :71 and SymmetricCurve1DConvFastV1 :140) on the reference's own zero-separated sequence
 * (:48-61 / :115-126).  With the sequence stored row-major as (L + 2h) x ld floats -- h = taps / 2 zero halo rows at both
 * ends, channels padded to ld % 4 == 0 with zeros -- row i of the would-be shifted-row matrix IS the contiguous span
 * starting at sequence row i - h: `A` points at the FIRST HALO ROW, `lda` = ld, K = taps * ld, and consecutive rows of
 * the operand simply overlap.  W is the (C_out, taps * ld) [tap][channel] matrix (zero in the padding columns).
 *   ccn_conv_rows_nt : Y[M x N] = A_overlap W^T + b (+ BatchNorm partial statistics, as ccn_gemm_nt)  -- forward, and the
 *                      data gradient (A = dY with halo, W = the tap-flipped transpose)
 *   ccn_conv_rows_tn : dW[N x K] += dY[M x N]^T X_overlap[M x K]                                    -- weight gradient
 * Same kernels, scratch and determinism as ccn_gemm_nt / ccn_gemm_tn_ws; K == lda degenerates to those. */
int ccn_conv_rows_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                     int64_t M, int64_t N, int64_t K, double* colstats, void* stream);
int ccn_conv_rows_tn(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                     int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream);
int ccn_gemm_tn_generic(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                        int64_t N, int64_t K, int overlap, void* stream); /* the split-K register-staged kernels of ccn_gemm_tn; overlap != 0: X rows overlap (per-element bounds) */
/* fp16 MFMA form of ccn_gemm_nt (BASELINE configs[4], "fp16 features"): A and W are read as fp32 and rounded to fp16
 * (round to nearest even) inside the kernel, products accumulate in fp32 (v_mfma_f32_32x32x16_f16); bias / Y / colstats
 * stay fp32.  Same requirements as ccn_gemm_nt_bf16.  The fp16 MLP mode uses it for the FORWARD products; gradients
 * (whose magnitudes fall below fp16's normal range without loss scaling) take the bf16 kernels. */
int ccn_gemm_nt_f16(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                    int64_t M, int64_t N, int64_t K, double* colstats, void* stream);
/* bf16 MFMA form of ccn_gemm_tn (bf16 MLP mode): dW += bf16(dY)^T bf16(X), fp32 accumulation; 16-byte aligned operands. */
int ccn_gemm_tn_bf16(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                     int64_t N, int64_t K, void* stream); /* dW += dY^T X (dW pre-zeroed by caller) */

/* ---- 16-bit STORAGE path of the 16-bit MLP modes (csrc/ccn_gemm_h.hip; BASELINE configs[2] "bf16 MLP MFMA path").
 * Replaces the products of PyG MLP layers (reference base.py:90-125, autograd of F.linear) where the hidden activation, the
 * BatchNorm-backward gradient dY and the cast weight are kept as bf16 / fp16 rows in HBM: operands reach LDS by LDS-DMA with
 * no conversion, v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulation.  16-bit rows: 16-byte aligned, leading dimension (in
 * elements) a multiple of 8, padding columns zero.  f16 != 0 selects fp16 (forward operands of the fp16 mode), else bf16.
 *   ccn_gemm_nt_h: Y = A W^T (+ bias).  out16 == 0: Y fp32 (+ colstats as ccn_gemm_nt); out16 != 0: Y 16-bit rows (ldy in
 *                  elements, % 4 == 0), no bias / statistics -- the data gradient of a 16-bit activation.
 *   ccn_gemm_tn_h: dW (fp32) += dY^T X, both operands bf16 rows; caller-owned scratch for the partial tiles
 *                  (ccn_gemm_tn_h_workspace_bytes), summed in chunk order (deterministic). */
int ccn_gemm_nt_h(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M,
                  int64_t N, int64_t K, double* colstats, int f16, int out16, void* stream);
/* round 6: the forward of a BatchNorm layer of the 16-bit modes WITHOUT its fp32 intermediate (PyG MLP hidden layers, reference
 * base.py:90-125: Linear -> BatchNorm1d -> activation).  The product kernel is HBM-bound at the network's widths, so it runs twice:
 *   ccn_gemm_nt_h_stats: the column statistics of A W^T (+ bias) as ccn_gemm_nt_h(out16 = 0) takes them -- the same fp32 sums --
 *                        and NOTHING written (2 K bytes per row read);
 *   ccn_gemm_nt_h_bnact: Z = act((A W^T) * scale[n] + shift[n]) (a bias is folded into shift by the caller), fp32 rows
 *                        (out16 = 0) or 16-bit rows (out16 != 0): the bits of ccn_gemm_nt_h followed by ccn_bn_act_fwd(_h).
 * 4 K + 2 N bytes per row instead of 2 K + 10 N.  The backward passes of such a layer recover what they need from Z itself:
 *   ccn_bn_act_bwd_reduce_hz / _apply_hz = ccn_bn_act_bwd_reduce(_h) / ccn_bn_act_bwd_apply_h with the layer's OUTPUT in place of
 *   its pre-normalisation product (zt 1 = bf16 rows, 2 = fp16 rows, 3 = fp32 rows; ldz in elements): t = act^-1(z),
 *   xhat = (t - beta) / gamma.  That needs an INVERTIBLE activation (LeakyReLU, none): BatchNorm's backward uses xhat of every row,
 *   also of the rows a ReLU clipped to 0.  For a ReLU layer ccn_gemm_nt_h_bnact therefore writes a second 16-bit result T = the
 *   pre-activation t (nullable; 16-bit form only), which these passes take in place of Z with z_pre != 0 (no inversion).
 *   A column with gamma = 0 has dy = 0 and xhat = 0 (its dgamma is then 0 instead of sum(g xhat): unrecoverable, measure zero in
 *   training).  dZ: bf16 rows (dz16 != 0) or fp32 rows. */
int ccn_gemm_nt_h_stats(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M, int64_t N, int64_t K,
                        double* colstats, int f16, void* stream);
int ccn_gemm_nt_h_bnact(const void* A, int64_t lda, const void* W, int64_t ldw, const float* scale, const float* shift, int act,
                        float slope, void* Z, int64_t ldz, void* T, int64_t ldt, int64_t M, int64_t N, int64_t K, int f16, int out16,
                        void* stream);
int ccn_bn_act_bwd_reduce_hz(const void* dZ, int dz16, int64_t lddz, const void* Z, int zt, int z_pre, int64_t ldz, int64_t rows, int64_t C,
                             const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                             double* sums, void* stream);
int ccn_bn_act_bwd_apply_hz(const void* dZ, int dz16, int64_t lddz, const void* Z, int zt, int z_pre, int64_t ldz, int64_t rows, int64_t C,
                            const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                            const double* sums, float count, int training, int acc_params, void* dY, int64_t lddy, float* dgamma,
                            float* dbeta, void* stream);
/* round 4: the implicit-GEMM curve convolution of ccn_conv_rows_nt / _tn (fast_conv1d.py:173-184 over the zero-separated
 * sequence of :48-61 / :115-126) on 16-BIT row sequences: row i of the shifted-row matrix is the span of K = taps * lda
 * elements starting at A + i * lda, read in place (lda % 8 == 0, taps / 2 zero halo rows at both ends of the allocation).
 * Same outputs as ccn_gemm_nt_h / ccn_gemm_tn_h on the materialised matrix (same products, same order). */
int ccn_conv_rows_nt_h(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M,
                       int64_t N, int64_t K, double* colstats, int f16, int out16, void* stream);
int ccn_conv_rows_tn_h(const void* dY, int64_t lddy, const void* X, int64_t ldx, int x_f16, float* dW, int64_t lddw, int64_t M,
                       int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream);
size_t ccn_gemm_tn_h_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ccn_gemm_tn_h(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw, int64_t M, int64_t N,
                  int64_t K, void* workspace, size_t workspace_bytes, void* stream);
/* ... with X as fp16 rows (fp16 mode): dW += dY^T bf16(fp16 X), the conversion done on the MFMA operand (no ccn_f16_to_bf16_rows pass) */
int ccn_gemm_tn_h_xf16(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw, int64_t M, int64_t N,
                       int64_t K, void* workspace, size_t workspace_bytes, void* stream);
/* fp32 rows -> 16-bit rows (round to nearest even), padding columns [C, ldy) zeroed; W (N x K fp32) -> W^T (K x N 16-bit) */
int ccn_cast_rows_h(const float* X, int64_t ldx, int64_t rows, int64_t C, void* Y, int64_t ldy, int f16, void* stream);
int ccn_transpose_cast_h(const float* W, int64_t ldw, int64_t N, int64_t K, void* Wt, int64_t ldt, int f16, void* stream);
/* Y = bf16(A + B), A fp32 rows, B bf16 rows: the two gradients of a product that left as fp32 rows and as a 16-bit copy */
int ccn_add_cast_rows_h(const float* A, int64_t lda, const void* B, int64_t ldb, int64_t rows, int64_t C, void* Y, int64_t ldy,
                        void* stream);
/* fp16 rows -> bf16 rows (the fp16 mode's weight-gradient operand: bf16(fp16(x))) */
int ccn_f16_to_bf16_rows(const void* X, int64_t ldx, int64_t rows, int64_t C, void* Y, int64_t ldy, void* stream);
/* ccn_bn_act_fwd writing z as 16-bit rows; ccn_bn_act_bwd_reduce reading a bf16 dZ; ccn_bn_act_bwd_apply_ex reading an fp32
 * (dz16 == 0) or bf16 dZ and writing dY as bf16 rows (same expressions, one rounding at the store) */
int ccn_bn_act_fwd_h(const float* Y, int64_t ldy, int64_t rows, int64_t C, const float* scale, const float* shift, int act,
                     float slope, void* Z, int64_t ldz, int f16, void* stream);
int ccn_bn_act_bwd_reduce_h(const void* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                            const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                            double* sums, void* stream);
int ccn_bn_act_bwd_apply_h(const void* dZ, int dz16, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                           const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                           const double* sums, float count, int training, int acc_params, void* dY, int64_t lddy, float* dgamma,
                           float* dbeta, int f16, void* stream);

int ccn_bn_finalize(const double* colstats, int64_t rows, int64_t C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                    float* save_mean, float* save_rstd, void* stream);
int ccn_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int64_t C, float* scale, float* shift, float* save_mean, float* save_rstd,
                       void* stream);
int ccn_bn_act_fwd(const float* Y, int64_t ldy, int64_t rows, int64_t C, const float* scale, const float* shift,
                   int act, float slope, float* Z, int64_t ldz, void* stream);
int ccn_bn_act_bwd_reduce(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                          const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                          float slope, double* sums, void* stream);
int ccn_bn_act_bwd_apply(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                         const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                         float slope, const double* sums, int training, float* dY, int64_t lddy, float* dgamma,
                         float* dbeta, void* stream);
int ccn_colsum(const float* X, int64_t ldx, int64_t rows, int64_t C, double* acc, float* out, void* stream);

/* ---- A15: src/models/modules/dgcnn.py:158-207 StaticEdgeConv.forward_fast ------------------------
 * idx is the FRNN output (B, Nmax, K); the self-loop column of dgcnn.py:166-168 is implicit (slot 0).
 * feat row (b, i, s) = [g, x_i - g], g = x[neighbour] or 0 (frnn_gather, dgcnn.py:172-173).
 * x is PACKED (N, C) with cloud_ptr (B+1); rows i >= len_b are the zero padding rows of quirk Q4. */
int ccn_sg_gather_fwd(const float* x, int64_t ldx, const int64_t* idx, const int64_t* cloud_ptr, int64_t B,
                      int64_t Nmax, int64_t K, int64_t C, float* feat, int64_t ldf, void* stream);
int ccn_sg_gather_bwd(const float* dfeat, int64_t lddf, const int64_t* idx, const int64_t* cloud_ptr, int64_t B,
                      int64_t Nmax, int64_t K, int64_t C, float* dx, int64_t lddx, void* stream);
/* ---- A15, first edge layer in algebraic form:  W [x_j ; x_i - x_j] = (Wa-Wb) x_j + Wb x_i.
 * ps (N, 2*Co): P = X (Wa-Wb)^T in columns [0,Co), S = X Wb^T (+bias) in [Co,2Co).  Dense row (b,i,s) is
 * y = P[neighbour] + S[i] (missing neighbour: S[i]; padding row: pad[c], NULL = 0), exactly the rows of
 * dgcnn.py:173-177 incl. quirk Q4.  stats: one partial row {sum y, sum y^2} per point group
 * (ccn_sg_edge_stats_rows of them) in the layout ccn_bn_finalize_n reduces.  apply: Z = act(y*scale+shift) (scale NULL:
 * identity).  bwd_stats / bwd: BatchNorm+activation backward and the scatter into dps (zero on entry). */
int64_t ccn_sg_edge_stats_rows(int64_t B, int64_t Nmax, int64_t Co);
int ccn_sg_edge_stats(const float* ps, int64_t ldps, const float* pad, const int64_t* idx, const int64_t* cloud_ptr,
                      int64_t B, int64_t Nmax, int64_t K, int64_t Co, double* partial, void* stream);
int ccn_sg_edge_apply(const float* ps, int64_t ldps, const float* pad, const int64_t* idx, const int64_t* cloud_ptr,
                      int64_t B, int64_t Nmax, int64_t K, int64_t Co, const float* scale, const float* shift, int act,
                      float slope, float* Z, int64_t ldz, void* stream);
int ccn_sg_edge_bwd_stats(const float* ps, int64_t ldps, const float* pad, const int64_t* idx,
                          const int64_t* cloud_ptr, int64_t B, int64_t Nmax, int64_t K, int64_t Co, const float* dZ,
                          int64_t lddz, const float* scale, const float* shift, const float* mean, const float* rstd,
                          int act, float slope, double* partial, void* stream);
int ccn_sg_edge_bwd(const float* ps, int64_t ldps, const float* pad, const int64_t* idx, const int64_t* cloud_ptr,
                    int64_t B, int64_t Nmax, int64_t K, int64_t Co, const float* dZ, int64_t lddz, const float* scale,
                    const float* shift, const float* mean, const float* rstd, int act, float slope, const double* sums,
                    int training, float* dps, int64_t lddps, void* stream);
/* reductions of caller-provided partial rows (nparts x 2C doubles, followed by 2C doubles of scratch) */
int ccn_bn_finalize_n(const double* partial, int64_t nparts, int64_t rows, int64_t C, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                      float* scale, float* shift, float* save_mean, float* save_rstd, void* stream);
int ccn_reduce_partials(double* partial, int64_t nparts, int64_t width, double* sums, void* stream);

/* masked max over the K+1 slots (dgcnn.py:187-189, fill -1e2) written to PACKED rows (dgcnn.py:206). */
int ccn_sg_max_fwd(const float* f, int64_t ldf, const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax,
                   int64_t K, int64_t C, float* out, int64_t ldo, int32_t* arg, void* stream);
int ccn_sg_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int64_t* cloud_ptr, int64_t B,
                   int64_t Nmax, int64_t K, int64_t C, float* df, int64_t lddf, void* stream);
/* The remaining aggregation branches of the reference (no shipped config selects them):
 * ccn_seg_wsum_*: PointNetConv2.aggregate over CSR groups (src/models/modules/point_conv.py:82-88) -- mode 0 'mean'
 * (scatter_mean: sum / max(count, 1)), mode 1 'weighted-sum' (scatter_add of msg * sigmoid(att)).
 * ccn_sg_reduce_*: StaticEdgeConv.forward_fast over the dense (b, i, slot) rows (src/models/modules/dgcnn.py:182-203) --
 * mode 0 'mean' (valid slots), 1 'weighted-sum' (sigmoid weights, masked, normalised by clamp(total, 1e-3)),
 * 2 'attend' (softmax over the K+1 slots with -5e2 at the invalid ones).  att / datt may be NULL for mode 0. */
int ccn_seg_wsum_fwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets, int64_t M,
                     int64_t C, int mode, float* out, int64_t ldo, void* stream);
int ccn_seg_wsum_bwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets, int64_t M,
                     int64_t C, int mode, const float* dout, int64_t lddo, float* dmsg, int64_t lddm, float* datt,
                     int64_t ldda, void* stream);
int ccn_sg_reduce_fwd(const float* f, int64_t ldf, const float* att, int64_t lda, const int64_t* idx, const int64_t* cloud_ptr,
                      int64_t B, int64_t Nmax, int64_t K, int64_t C, int mode, float* out, int64_t ldo, void* stream);
int ccn_sg_reduce_bwd(const float* f, int64_t ldf, const float* att, int64_t lda, const int64_t* idx, const int64_t* cloud_ptr,
                      int64_t B, int64_t Nmax, int64_t K, int64_t C, int mode, const float* dout, int64_t lddo, float* df,
                      int64_t lddf, float* datt, int64_t ldda, void* stream);

/* ---- A13: src/models/modules/point_conv.py:60-93 PointNetConv2 message + aggregate ----------------
 * msg[e] = [x_src[src[e]], (pos_src[src[e]] - pos_dst[dst[e]]) / radius]   (radius <= 0: no division). */
int ccn_msg_build_fwd(const float* x_src, int64_t ldx, const float* pos_src, const float* pos_dst,
                      const int64_t* src, const int64_t* dst, int64_t E, int64_t C, float radius, float* msg,
                      int64_t ldm, void* stream);
int ccn_msg_build_bwd(const float* dmsg, int64_t lddm, const int64_t* src, int64_t E, int64_t C, float* dx,
                      int64_t lddx, void* stream);
/* edges grouped by destination: offsets int32 (M+1).  softmax over each group per channel (PyG softmax,
 * +1e-16 in the denominator), weighted sum (point_conv.py:89-93). */
int ccn_seg_softmax_agg_fwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets,
                            int64_t M, int64_t C, float* out, int64_t ldo, void* stream);
int ccn_seg_softmax_agg_bwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets,
                            int64_t M, int64_t C, const float* dout, int64_t lddo, float* dmsg, int64_t lddm,
                            float* datt, int64_t ldda, void* stream);
/* ... datt written as bf16 rows (round 3, 16-bit storage modes): dY of attend_nn's plain last Linear (point_conv.py:89-92). */
int ccn_seg_softmax_agg_bwd_h(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets,
                              int64_t M, int64_t C, const float* dout, int64_t lddo, float* dmsg, int64_t lddm,
                              void* datt, int64_t ldda, void* stream);
/* scatter_max (point_conv.py:81-82): empty groups give 0. */
int ccn_seg_max_fwd(const float* msg, int64_t ldm, const int32_t* offsets, int64_t M, int64_t C, float* out,
                    int64_t ldo, int32_t* arg, void* stream);
int ccn_seg_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* offsets, int64_t M,
                    int64_t C, float* dmsg, int64_t lddm, void* stream);

/* ---- section 8(f) "next" rows ------------------------------------------------------------------------
 * knn_points: pytorch3d.ops.knn_points as used by knn_interpolate_pytorch3d (point_ops.py:91, 331-336): per cloud
 * the K nearest source points of every query, ascending (d2, index); nbr = PACKED source index (-1 if the cloud has
 * fewer than K points), weight = 1 / clamp(|x - y|^2, 1e-16).  q_ptr / s_ptr: int64 (B+1) cloud offsets. */
int ccn_knn_points(const float* q, const int64_t* q_ptr, const float* src, const int64_t* s_ptr, int64_t B,
                   int64_t max_q, int64_t K, int64_t* nbr, float* weight, void* stream);
/* Grid-accelerated form of ccn_knn_points for large clouds (same results): ccn_knn_cloud_radius gives a per-cloud search
 * radius from the source density, ccn_frnn_grid_build / ccn_frnn_query find the K nearest inside it, ccn_knn_from_grid
 * converts the (B,P1,K) cloud-local result into the packed table + weights and flags the queries that found fewer than
 * min(K, sources); those are listed (ccn_exclusive_scan_i32 over the flags + ccn_scatter_flagged) and recomputed
 * exhaustively by ccn_knn_points_list (count: device int64 = number of listed queries, max_count = list capacity). */
int ccn_knn_cloud_radius(const float* src, const int64_t* s_ptr, int64_t B, float scale, float* radius, void* stream);
int ccn_knn_from_grid(const int64_t* idx, const float* q, const int64_t* q_ptr, const float* src, const int64_t* s_ptr,
                      int64_t B, int64_t P1, int64_t K, int64_t* nbr, float* weight, int32_t* flag, void* stream);
int ccn_scatter_flagged(const int32_t* flag, const int32_t* offsets, int64_t n, int64_t* list, void* stream);
int ccn_knn_points_list(const float* q, const int64_t* q_ptr, const float* src, const int64_t* s_ptr, int64_t B,
                        int64_t K, const int64_t* list, const int64_t* count, int64_t max_count, int64_t* nbr,
                        float* weight, void* stream);
/* ball_query: pytorch3d.ops.ball_query as called at point_ops.py:81 (SA with use_fast_knn: False): padded
 * (B,P,3) inputs, idx (B,P1,K) int64 = the first K points2 in index order with d2 < r*r, -1 padded. */
int ccn_ball_query(const float* points1, const int64_t* lengths1, const float* points2, const int64_t* lengths2,
                   int64_t B, int64_t P1, int64_t P2, int64_t K, float radius, int64_t* idx, void* stream);
/* ---- SGCNN dense path on COMPACT rows (dgcnn.py:158-207, same result as the B*Nmax*(K+1)-row computation).
 * A dense slot whose FRNN entry is -1 has the same first-layer value S[i] whatever the slot, and so have all rows of the
 * padding points: they only enter the BatchNorm statistics (quirk Q4).  Layout: rows [0,E) real (point p owns
 * [grp_ptr[p], grp_ptr[p+1]): self, then its neighbours in FRNN order; row_src = packed source point), rows [E,E+Ne) one
 * representative per point with empty slots (rep_row[p] = its row or -1, weight = number of empty slots), row E+Ne = all
 * padding rows (weight set by the caller); row_w holds the Ne+1 weights.  cg_count -> two ccn_exclusive_scan_i32 ->
 * cg_fill build it; the cg_edge_* kernels are the weighted counterparts of ccn_sg_edge_*; ccn_colstats_weighted /
 * ccn_bn_act_bwd_reduce_weighted / ccn_bn_act_bwd_apply_count carry the weights through the following layers
 * (count = B*Nmax*(K+1) = sum of all weights); cg_max is the masked max over the real rows (+ the -1e2 of empty slots). */
int ccn_cg_count(const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax, int64_t K, int32_t* cnt,
                 int32_t* has_rep, void* stream);
int ccn_cg_fill(const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax, int64_t K, const int32_t* grp_ptr,
                const int32_t* rep_off, int64_t E, int32_t* row_src, int32_t* rep_row, float* row_w, void* stream);
int64_t ccn_cg_edge_stats_rows(int64_t N, int64_t Co);
int ccn_cg_edge_stats(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                      const int32_t* rep_row, const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co,
                      double* partial, void* stream);
int ccn_cg_edge_apply(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                      const int32_t* rep_row, int64_t N, int64_t E, int64_t Ne, int64_t Co, const float* scale,
                      const float* shift, int act, float slope, float* Z, int64_t ldz, void* stream);
int ccn_cg_edge_bwd_stats(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                          const int32_t* rep_row, const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co,
                          const float* dZ, int64_t lddz, const float* scale, const float* shift, const float* mean,
                          const float* rstd, int act, float slope, double* partial, void* stream);
int ccn_cg_edge_bwd(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src, const int32_t* rep_row,
                    const float* row_w, int64_t N, int64_t E, int64_t Co, const float* dZ, int64_t lddz,
                    const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                    const double* sums, double count, int training, float* dps, int64_t lddps, void* stream);
int ccn_cg_max_fwd(const float* f, int64_t ldf, const int32_t* grp_ptr, const int32_t* rep_row, int64_t N, int64_t C,
                   float* out, int64_t ldo, int32_t* arg, void* stream);
int ccn_cg_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* grp_ptr, const int32_t* rep_row,
                   int64_t N, int64_t R, int64_t C, float* df, int64_t lddf, void* stream);
/* ... df written as bf16 rows (round 3, 16-bit storage modes): dY of the plain Linear in front of the max (dgcnn.py:172-181). */
int ccn_cg_max_bwd_h(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* grp_ptr, const int32_t* rep_row,
                     int64_t N, int64_t R, int64_t C, void* df, int64_t lddf, void* stream);
/* w == NULL: all weights 1 (plain column sums of x and x^2) */
int ccn_colstats_weighted(const float* X, int64_t ldx, const float* w, int64_t rows, int64_t C, double* acc,
                          void* stream);
int ccn_bn_act_bwd_reduce_weighted(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, const float* w,
                                   int64_t rows, int64_t C, const float* scale, const float* shift, const float* mean,
                                   const float* rstd, int act, float slope, double* sums, void* stream);
int ccn_bn_act_bwd_apply_count(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                               const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                               float slope, const double* sums, double count, int training, float* dY, int64_t lddy,
                               float* dgamma, float* dbeta, void* stream);
/* same, accumulate_params != 0: dgamma / dbeta are ADDED to the given buffers (the parameters' gradient-bucket views) */
int ccn_bn_act_bwd_apply_ex(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                            const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                            float slope, const double* sums, double count, int training, int accumulate_params,
                            float* dY, int64_t lddy, float* dgamma, float* dbeta, void* stream);

/* PointNetConv2 first message layer in algebraic form (point_conv.py:35-93; local_nn.lins[0].weight = [Wx | Wp]):
 *   y[e] = PX[src[e]] + Wp (pos_src[src[e]] - pos_dst[dst[e]]) / radius + bias,   PX = X Wx^T  (N_src x Co, one GEMM over the
 * source points instead of the E edge rows; radius <= 0: no division), followed by BatchNorm over the E edges + activation.
 * stats: partial [ccn_pn_edge_stats_rows(E,Co)][2*Co] doubles (sum y, sum y^2 | sum g, sum g*xhat) for ccn_bn_finalize_n /
 * ccn_reduce_partials; bwd: dPX (zero on entry, atomic per source point) and wpart [ccn_pn_edge_bwd_rows(E)][4*Co] doubles
 * holding per-wave sums of dy*rel_x, dy*rel_y, dy*rel_z, dy (=> dWp columns and dbias after ccn_reduce_partials). */
int64_t ccn_pn_edge_stats_rows(int64_t E, int64_t Co);
int64_t ccn_pn_edge_bwd_rows(int64_t E);
int ccn_pn_edge_stats(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                      const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                      int64_t Co, float radius, double* partial, void* stream);
int ccn_pn_edge_apply(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                      const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                      int64_t Co, float radius, const float* scale, const float* shift, int act, float slope, float* Z,
                      int64_t ldz, void* stream);
int ccn_pn_edge_bwd_stats(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                          const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                          int64_t Co, float radius, const float* dZ, int64_t lddz, const float* scale,
                          const float* shift, const float* mean, const float* rstd, int act, float slope,
                          double* partial, void* stream);
int ccn_pn_edge_bwd(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                    const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                    int64_t Co, float radius, const float* dZ, int64_t lddz, const float* scale, const float* shift,
                    const float* mean, const float* rstd, int act, float slope, const double* sums, int training,
                    float* dpx, int64_t lddpx, double* wpart, void* stream);
/* The same first layers at the boundary of the 16-bit storage modes (round 3; ccn_gemm_nt_h and friends): `_apply_h` writes the
 * activation as bf16 rows (fp16 when f16 != 0; Co % 8 == 0, ldz in 16-bit elements, rows 16-byte aligned) for the next
 * Linear of the MLP (PyG MLP inside dgcnn.py:172-177 / point_conv.py:60-69), `_bwd_stats_h` / `_bwd_h` read the gradient of
 * that activation as bf16 rows.  Same arithmetic as the fp32 entries; one rounding on the way out / in. */
int ccn_cg_edge_apply_h(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                        const int32_t* rep_row, int64_t N, int64_t E, int64_t Ne, int64_t Co, const float* scale,
                        const float* shift, int act, float slope, void* Z, int64_t ldz, int f16, void* stream);
int ccn_cg_edge_bwd_stats_h(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                            const int32_t* rep_row, const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co,
                            const void* dZ, int64_t lddz, const float* scale, const float* shift, const float* mean,
                            const float* rstd, int act, float slope, double* partial, void* stream);
int ccn_cg_edge_bwd_h(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src, const int32_t* rep_row,
                      const float* row_w, int64_t N, int64_t E, int64_t Co, const void* dZ, int64_t lddz,
                      const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                      const double* sums, double count, int training, float* dps, int64_t lddps, void* stream);
/* round 5: the backward of the compact first SGCNN layer (autograd of the gather + first Linear / BatchNorm / activation of
 * src/models/modules/dgcnn.py:172-177) WITHOUT atomics and with one pass over dZ per index order.  dy_r = scale (g_r - m1 -
 * xhat_r m2) is linear in (g_r, 1, xhat_r): _sums takes the BatchNorm-backward column sums (as ccn_cg_edge_bwd_stats: `partial`)
 * and, in the same pass, per point the sums pt = [sum w g | sum w xhat] over ITS rows; _gather takes pp = [sum g | sum xhat] per
 * SOURCE point over the rows that read it, through the inverse of row_src (inv_ptr int32 (N+1), inv_row int32 (E): rows sorted by
 * source, ascending; row_dst int32 (E): the point that owns row r); _finish combines them with the reduced column sums into
 * dps = [dP | dS] (N x 2 Co).  Deterministic; replaces ccn_cg_edge_bwd_stats + ccn_cg_edge_bwd (kept: the round-1..4 form with
 * fp32 atomics).  dz16: dZ as bf16 rows.  Without BatchNorm: scale = 1, shift = mean = rstd = 0, training = 0. */
int ccn_cg_edge_bwd_sums(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src, const int32_t* rep_row,
                         const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co, const void* dZ, int dz16, int64_t lddz,
                         const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                         double* partial, float* pt, int64_t ldpt, void* stream);
int ccn_cg_edge_bwd_gather(const float* ps, int64_t ldps, const int32_t* inv_ptr, const int32_t* inv_row, const int32_t* row_dst,
                           int64_t N, int64_t Co, const void* dZ, int dz16, int64_t lddz, const float* scale, const float* shift,
                           const float* mean, const float* rstd, int act, float slope, float* pp, int64_t ldpp, void* stream);
int ccn_cg_edge_bwd_finish(const float* pt, int64_t ldpt, const float* pp, int64_t ldpp, const int32_t* grp_ptr,
                           const int32_t* rep_row, const float* row_w, const int32_t* inv_ptr, int64_t N, int64_t E, int64_t Co,
                           const float* scale, const double* sums, double count, int training, float* dps, int64_t lddps,
                           void* stream);
/* ... and of PointNetConv2's algebraic first layer (autograd of src/models/modules/point_conv.py:60-69): _sums = the column sums
 * over the E edges in ONE pass over dZ -- partial: [ccn_pn_edge_stats_rows(E, Co)][9 Co + 4] doubles per row [g | g xhat | g rel_0..2 |
 * xhat | xhat rel_0..2 | rel_0..2 -]; _gather = per SOURCE point pp = [sum g | sum xhat] over the edges that read it, through the
 * inverse of `src` (inv_ptr int32 (Nsrc+1), inv_edge int32 (E): edges sorted by source, ascending); _finish = dPX (Nsrc x Co) and
 * dw4 = [dWp[:, 0] | dWp[:, 1] | dWp[:, 2] | dbias] (4 x Co floats) from pp and the reduced sums (9 Co + 4 totals).  Deterministic;
 * replaces ccn_pn_edge_bwd_stats + ccn_pn_edge_bwd (kept). */
int ccn_pn_edge_bwd_sums(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias, const float* pos_src,
                         const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E, int64_t Co, float radius,
                         const void* dZ, int dz16, int64_t lddz, const float* scale, const float* shift, const float* mean,
                         const float* rstd, int act, float slope, double* partial, void* stream);
int ccn_pn_edge_bwd_gather(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias, const float* pos_src,
                           const float* pos_dst, const int64_t* dst, const int32_t* inv_ptr, const int32_t* inv_edge,
                           int64_t Nsrc, int64_t Co, float radius, const void* dZ, int dz16, int64_t lddz, const float* scale,
                           const float* shift, const float* mean, const float* rstd, int act, float slope, float* pp,
                           int64_t ldpp, void* stream);
int ccn_pn_edge_bwd_finish(const float* pp, int64_t ldpp, const int32_t* inv_ptr, int64_t Nsrc, int64_t E, int64_t Co,
                           const float* scale, const double* sums, int training, float* dpx, int64_t lddpx, float* dw4,
                           void* stream);
int ccn_pn_edge_apply_h(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                        const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                        int64_t Co, float radius, const float* scale, const float* shift, int act, float slope, void* Z,
                        int64_t ldz, int f16, void* stream);
int ccn_pn_edge_bwd_stats_h(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                            const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                            int64_t Co, float radius, const void* dZ, int64_t lddz, const float* scale,
                            const float* shift, const float* mean, const float* rstd, int act, float slope,
                            double* partial, void* stream);
int ccn_pn_edge_bwd_h(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                      const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                      int64_t Co, float radius, const void* dZ, int64_t lddz, const float* scale, const float* shift,
                      const float* mean, const float* rstd, int act, float slope, const double* sums, int training,
                      float* dpx, int64_t lddpx, double* wpart, void* stream);
/* ball query between D-dimensional feature vectors (dgcnn.py:114-127 DGCNNLayerRadius: the ball-group search of
 * point_ops.py:81 on features): padded (B,P,ld) rows of D floats, same first-K-in-index-order rule, d2 summed over
 * the D components in order. */
int ccn_ball_query_nd(const float* points1, int64_t ld1, const int64_t* lengths1, const float* points2, int64_t ld2,
                      const int64_t* lengths2, int64_t B, int64_t P1, int64_t P2, int64_t D, int64_t K, float radius,
                      int64_t* idx, void* stream);
/* sparse edge conv message (dgcnn.py:227-228, forward_slow): msg[e] = [x_i, x_j - x_i], i = dst[e], j = src[e];
 * bwd accumulates into dx (zero on entry). */
int ccn_edge_feat_fwd(const float* x, int64_t ldx, const int64_t* src, const int64_t* dst, int64_t E, int64_t C,
                      float* msg, int64_t ldm, void* stream);
int ccn_edge_feat_bwd(const float* dmsg, int64_t lddm, const int64_t* src, const int64_t* dst, int64_t E, int64_t C,
                      float* dx, int64_t lddx, void* stream);
/* round 3: the message rows as 16-bit rows for the 16-bit storage modes ((2 C) % 8 == 0); backward over edges GROUPED by
 * destination (CSR offsets, num_dst + 1 int32): one add per destination and channel instead of one per edge on that side;
 * dmsg as fp32 (dm16 = 0) or bf16 rows (dm16 = 1). */
int ccn_edge_feat_fwd_h(const float* x, int64_t ldx, const int64_t* src, const int64_t* dst, int64_t E, int64_t C, void* msg,
                        int64_t ldm, int f16, void* stream);
int ccn_edge_feat_bwd_csr(const void* dmsg, int dm16, int64_t lddm, const int64_t* src, const int32_t* offsets, int64_t num_dst,
                          int64_t N, int64_t E, int64_t C, float* dx, int64_t lddx, void* stream); /* group i = point i of x: num_dst == N (rows of dx) is required */
/* VoxelFPS (fps_ops.py:42-60): key = (cloud, floor(p/v)) packed in lexicographic order, score = distance to the
 * voxel corner + rnd*v/4; argmin: per voxel the point with the smallest score.  bad: device int64 = #points whose
 * voxel coordinates do not fit 18 bits. */
int ccn_voxel_keys(const float* pos, const int64_t* batch, const float* rnd, int64_t n, float voxel, int64_t* key,
                   float* score, int64_t* bad, void* stream);
int ccn_voxel_argmin(const float* score, const int64_t* voxel_of, int64_t n, int64_t num_voxels, int64_t* scratch,
                     int64_t* idx, void* stream);
/* Dense ranks of non-negative int64 keys: rank[i] = number of DISTINCT keys smaller than key[i] (torch.unique(sorted=True,
 * return_inverse=True) as the reference's VoxelFPS uses it, fps_ops.py:51-60), count[0] = number of distinct keys.  LSD radix
 * sort, 8-bit digits, stable and deterministic; digit_mask bit b = sort on digit b -- ccn_key_spread gives the OR of
 * key[i] ^ key[0], only digits in which the keys differ need a pass.  Caller-owned workspace. */
int ccn_key_spread(const int64_t* key, int64_t n, int64_t* spread, void* stream);
size_t ccn_rank_keys_workspace_bytes(int64_t n);
int ccn_rank_keys(const int64_t* key, int64_t n, int digit_mask, int64_t* rank, int64_t* count, void* workspace,
                  size_t workspace_bytes, void* stream);
/* sorted[] = the keys in ascending order (the final sort of the reference's farthest-point indices, point_ops.py:57-70) */
int ccn_sort_keys(const int64_t* key, int64_t n, int digit_mask, int64_t* sorted, void* workspace, size_t workspace_bytes,
                  void* stream);
/* sample_farthest_points (point_ops.py:57-70): per cloud out_ptr[b+1]-out_ptr[b] samples starting at start[b];
 * out = packed point indices in selection order; n = cloud_ptr[B] (all points); max_cloud: largest cloud size (clouds of up
 * to 16384 points are processed register-resident, 0 = unknown); workspace: ccn_fps_workspace_bytes(n, B), caller-owned.
 * Clouds of 16 k - 64 k points are sampled by a cluster of up to four workgroups that exchange their candidates every round;
 * an ordinary launch cannot promise that the members run at the same time, so a member that hears nothing from a partner for
 * ~0.5 s gives the cloud up and a second, gated launch on the same stream re-samples that cloud with one workgroup: the
 * samples are the same on every path.  fallbacks (nullable, device int32, NOT cleared here): incremented once per cloud that
 * took the second way -- a counter for monitoring, never needed for correctness. */
size_t ccn_fps_workspace_bytes(int64_t n, int64_t B);
int ccn_fps(const float* pos, const int64_t* cloud_ptr, const int64_t* start, const int64_t* out_ptr, int64_t B,
            int64_t max_cloud, int64_t n, void* workspace, size_t workspace_bytes, int32_t* fallbacks, int64_t* out,
            void* stream);

/* Dataset-side curve splitter (src/data/kitti_dataset.py:73-92 with beam == NULL,
 * src/data/nuscenes_dataset.py:101-118 on the beam-sorted sweep): curve_idx[0] = 0, a new curve starts at i where
 * beam[i] != beam[i-1] or the fp64 edge length |p_i - p_{i-1}| exceeds thresh * sqrt(|p_i.xy|) (fp32 right-hand side,
 * as the reference evaluates it).  Bit-exact with the reference's cumsum; num_curves: device int64 = last id + 1. */
size_t ccn_curve_split_workspace_bytes(int64_t n);
int ccn_curve_split(const float* pos, const int64_t* beam, int64_t n, float thresh, int64_t* curve_idx,
                    int64_t* num_curves, void* ws, size_t ws_bytes, void* stream);

/* Harness row H (src/main.py:56 torch.optim.Adam; src/run/kitti_seg.py:19-63 train loop): one Adam update over a
 * flat, 16-byte aligned run of n parameters (param, grad, exp_avg, exp_avg_sq contiguous fp32), same arithmetic as
 * torch.optim.Adam(amsgrad=False, maximize=False): step >= 1 is the 1-based update count. */
/* Mean negative log-likelihood of log_softmax(logits) over the rows whose target != ignore_index (harness counterpart:
 * F.log_softmax + F.nll_loss at src/run/kitti_seg.py:184-192, shapenet_seg.py, nuscenes_seg.py, audi_seg.py).
 * fwd: lse (rows) float, per_point (rows, nullable) float, scratch double[2 * ccn_nll_loss_blocks(rows) + 2] whose last two
 * doubles receive (sum of losses, counted rows), loss = their quotient (device scalar).  bwd: dlogits = (softmax - onehot) *
 * grad_loss / counted rows.  Deterministic (fixed-order double sums). */
int64_t ccn_nll_loss_blocks(int64_t rows);
int ccn_nll_loss_fwd(const float* logits, int64_t ld, const int64_t* target, int64_t rows, int64_t C, int64_t ignore_index,
                     float* lse, float* per_point, double* scratch, float* loss, void* stream);
int ccn_nll_loss_bwd(const float* logits, int64_t ld, const int64_t* target, const float* lse, int64_t rows, int64_t C,
                     int64_t ignore_index, const float* grad_loss, const double* totals, float* dlogits, int64_t ldd,
                     void* stream);
int ccn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, void* stream);

/* ================================================================================================================
 * The entry points SURVEY.md section 8(b) lists as the minimum C-ABI of the hot path (ccn_contract.hip).  Each is a FIXED
 * composition of the kernels declared above, for a host that does not want to sequence them itself; the Python mirror
 * calls the pieces directly because it interleaves them with autograd bookkeeping and a second stream.  Section 8(b)
 * name -> export:
 *   ccn_segment_ptr                       as above (A1/A2)              ccn_curve_fps              as above (A7)
 *   ccn_frnn_grid_build / ccn_frnn_query  as above (A11)                ccn_curve_group_superset   as above (A9)
 *   ccn_curve_group_subset                two phases, the host allocates E edges in between: _count then _fill (A8)
 *   ccn_bn_act_fwd                        as above
 *   ccn_curve_conv_{fwd,bwd_data,bwd_weight}, ccn_bn_act_bwd, ccn_linear_bn_act_{fwd,bwd}, ccn_gather_edge_{fwd,bwd},
 *   ccn_edge_reduce_{max,attend}_{fwd,bwd}                              below
 * ================================================================================================================ */
#define CCN_DTYPE_F32 0   /* v_mfma_f32_32x32x2_f32 products */
#define CCN_DTYPE_BF16 1  /* operands rounded to bf16 inside the kernel, fp32 accumulate (BASELINE configs[2]) */
#define CCN_DTYPE_F16 2   /* forward products in fp16, gradients in bf16 (BASELINE configs[4]) */

/* Wt (K x ldt) = W (N x K, row stride ldw) transposed, columns N..ldt-1 zero: the operand of the data-gradient product
 * dX = dY W written as an NT product (autograd of F.linear in torch_geometric.nn.MLP, src/models/base.py:90-125). */
int ccn_transpose_pad(const float* W, int64_t ldw, int64_t N, int64_t K, float* Wt, int64_t ldt, void* stream);

/* F.conv1d(input (1, C_in, L), weight, bias, stride 1, 'same') at src/models/modules/fast_conv1d.py:183 and its autograd, on the
 * reference's zero-separated row sequence (fast_conv1d.py:48-61 V2, :115-126 V1) stored as (L + 2h) x ld floats, h = taps / 2
 * zero halo rows on both ends, ld % 4 == 0, padding columns zero.  `seq` / `dYseq` point at the FIRST HALO ROW.
 *   fwd        : Y (L x C_out) = conv(seq) + b, W in (C_out, taps, ld) [tap][channel] order (ldw >= taps * ld); colstats (nullable)
 *                receives the BatchNorm partial sums of Y exactly as ccn_gemm_nt does
 *   bwd_data   : dX (L x C_in) from dYseq ((L + 2h) x lddy, halo rows zero) and the SAME forward weight W (row stride taps * ld_in);
 *                workspace holds the tap-reversed transpose
 *   bwd_weight : dW (C_out x taps*ld) += dY^T shifted(seq); dY points at its row 0 (no halo needed) */
int ccn_curve_conv_fwd(const float* seq, int64_t ld, int64_t rows, int64_t taps, const float* W, int64_t ldw, const float* bias,
                       int64_t Cout, float* Y, int64_t ldy, double* colstats, void* stream);
size_t ccn_curve_conv_bwd_data_workspace_bytes(int64_t Cin, int64_t taps, int64_t lddy);
int ccn_curve_conv_bwd_data(const float* dYseq, int64_t lddy, int64_t rows, int64_t taps, const float* W, int64_t ld_in,
                            int64_t Cout, int64_t Cin, float* dX, int64_t lddx, void* workspace, size_t workspace_bytes,
                            void* stream);
size_t ccn_curve_conv_bwd_weight_workspace_bytes(int64_t rows, int64_t Cout, int64_t taps, int64_t ld);
int ccn_curve_conv_bwd_weight(const float* dY, int64_t lddy, const float* seq, int64_t ld, int64_t rows, int64_t taps,
                              int64_t Cout, float* dW, int64_t lddw, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of z = act(BatchNorm(y)) (fast_conv1d.py:72-73, 141-143; the norm + act of torch_geometric.nn.MLP, base.py:90-125):
 * params = the 4 x C table (scale, shift, mean, rstd) the forward produced; training != 0: batch statistics (the two column
 * sums are taken first, then dY, dgamma, dbeta in one pass); dY may alias dZ. */
size_t ccn_bn_act_bwd_workspace_bytes(int64_t rows, int64_t C);
int ccn_bn_act_bwd(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C, const float* params,
                   int act, float slope, int training, float* dY, int64_t lddy, float* dgamma, float* dbeta, void* workspace,
                   size_t workspace_bytes, void* stream);

/* One hidden layer of torch_geometric.nn.MLP as the reference builds it (src/models/base.py:32,64,90-125, mlp.py:13):
 * Z = act(BatchNorm1d(X W^T + b)); gamma == NULL: the plain last Linear (Y only).  fwd keeps Y (pre-normalisation product)
 * and params (4 x N: scale, shift, mean, rstd) for bwd and updates the running statistics when training.  bwd: dY (M x N)
 * receives the gradient of the product; dX / dW / dbias may be NULL; dW is ACCUMULATED INTO (a gradient bucket), dbias,
 * dgamma, dbeta are overwritten.  One workspace size serves both. */
size_t ccn_linear_bn_act_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ccn_linear_bn_act_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, int64_t M, int64_t N, int64_t K, float eps,
                          float momentum, int training, int act, float slope, int dtype, float* Y, int64_t ldy, float* Z,
                          int64_t ldz, float* params, void* workspace, size_t workspace_bytes, void* stream);
int ccn_linear_bn_act_bwd(const float* dZ, int64_t lddz, const float* X, int64_t ldx, const float* W, int64_t ldw, const float* Y,
                          int64_t ldy, const float* params, int64_t M, int64_t N, int64_t K, int training, int act, float slope,
                          int dtype, float* dY, int64_t lddy, float* dX, int64_t lddx, float* dW, int64_t lddw, float* dbias,
                          float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* frnn.frnn_gather(x, idxs, lengths) and its gradient (src/models/modules/dgcnn.py:172), literally: feat[(b, i, s), :] =
 * x[cloud_ptr[b] + idx[b, i, s], :] over the PACKED x (N x C), zero where idx < 0 (which includes the rows past a cloud's
 * length); bwd adds into dx, which must be zero on entry.  (The SGCNN steps do not go through it: ccn_sg_* / ccn_cg_* fold the
 * gather into the first layer.) */
int ccn_gather_edge_fwd(const float* x, int64_t ldx, const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax,
                        int64_t K, int64_t C, float* feat, int64_t ldf, void* stream);
int ccn_gather_edge_bwd(const float* dfeat, int64_t lddf, const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax,
                        int64_t K, int64_t C, float* dx, int64_t lddx, void* stream);
/* Aggregation of the messages of an edge list sorted by destination (offsets int32[M + 1]): max (scatter_max,
 * src/models/modules/point_conv.py:80-81) = ccn_seg_max_*, attend (softmax over the destination's edges times the message,
 * summed: point_conv.py:89-92) = ccn_seg_softmax_agg_*. */
int ccn_edge_reduce_max_fwd(const float* msg, int64_t ldm, const int32_t* offsets, int64_t M, int64_t C, float* out, int64_t ldo,
                            int32_t* arg, void* stream);
int ccn_edge_reduce_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* offsets, int64_t M, int64_t C,
                            float* dmsg, int64_t lddm, void* stream);
int ccn_edge_reduce_attend_fwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets, int64_t M,
                               int64_t C, float* out, int64_t ldo, void* stream);
int ccn_edge_reduce_attend_bwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets, int64_t M,
                               int64_t C, const float* dout, int64_t lddo, float* dmsg, int64_t lddm, float* datt, int64_t ldda,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCN_HIP_H */

// The entry points SURVEY.md section 8(b) names as the minimum C-ABI of the hot path, for hosts that do not want to sequence
// the individual kernels themselves: each one is a FIXED composition of the kernels declared further up in ccn_hip.h (the
// Python mirror calls those pieces directly, because it interleaves them with autograd bookkeeping and a second stream).
// Nothing here allocates, synchronises or keeps state; scratch comes from the caller (..._workspace_bytes).
//
//   ccn_curve_conv_{fwd,bwd_data,bwd_weight}   F.conv1d(1 x C x L, 'same') over the zero-separated row sequence and its
//                                              autograd (src/models/modules/fast_conv1d.py:183; :71 V2, :140 V1)
//   ccn_bn_act_bwd                             BatchNorm1d(batch statistics) + activation, backward (fast_conv1d.py:72-73,
//                                              141-143; torch_geometric MLP norm + act, src/models/base.py:90-125)
//   ccn_linear_bn_act_{fwd,bwd}                one hidden layer of torch_geometric.nn.MLP: Linear -> BatchNorm -> act
//                                              (src/models/base.py:32,64,90-125; src/models/modules/mlp.py:13)
//   ccn_gather_edge_{fwd,bwd}                  frnn.frnn_gather (src/models/modules/dgcnn.py:172), literally: the dense SGCNN
//                                              path itself never materialises it (ccn_sg_* / ccn_cg_*: algebraic first layer)
//   ccn_edge_reduce_max_{fwd,bwd}              scatter_max over the destination (point_conv.py:80-81, dgcnn.py:226-228)
//   ccn_edge_reduce_attend_{fwd,bwd}           softmax(att, dst) * msg -> scatter_add (point_conv.py:89-92)
#include "ccn_common.h"

namespace {

constexpr int TPB = 256;

// W (N x K, row stride ldw) -> Wt (K x ldt), columns N..ldt-1 zero
__global__ void transpose_pad_kernel(const float* __restrict__ W, int64_t ldw, int64_t N, int64_t K, float* __restrict__ Wt,
                                     int64_t ldt) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
  const int64_t n0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8) {
    const int64_t n = n0 + r, k = k0 + tx;
    tile[r][tx] = (n < N && k < K) ? W[n * ldw + k] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t k = k0 + r, n = n0 + tx;
    if (k < K && n < ldt) Wt[k * ldt + n] = tile[tx][r];
  }
}

// W (Cout x taps x ld_in) -> Wf (Cin x taps x ld_out): Wf[ci][t][co] = W[co][taps-1-t][ci], zero for co >= Cout
__global__ void conv_weight_flip_kernel(const float* __restrict__ W, int64_t ld_in, int64_t Cout, int64_t Cin, int taps,
                                        float* __restrict__ Wf, int64_t ld_out) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= Cin * taps * ld_out) return;
  const int64_t co = e % ld_out, t = (e / ld_out) % taps, ci = e / (ld_out * taps);
  Wf[e] = co < Cout ? W[(co * taps + (taps - 1 - t)) * ld_in + ci] : 0.f;
}

// frnn_gather: feat[(b, i, s)] = x[cloud_ptr[b] + idx[b, i, s]], zero where idx < 0.  One (b, i, s) slot per wave.
__global__ __launch_bounds__(256) void gather_edge_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const int64_t* __restrict__ idx,
                                                              const int64_t* __restrict__ cloud_ptr, int64_t slots,
                                                              int64_t per_cloud, int64_t C, float* __restrict__ feat,
                                                              int64_t ldf) {
  const int cx = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= slots) return;
  const int64_t j = idx[e];
  const float* src = j >= 0 ? x + (cloud_ptr[e / per_cloud] + j) * ldx : nullptr;
  for (int64_t c = cx; c < C; c += 64) feat[e * ldf + c] = src ? src[c] : 0.f;
}

// dx must be zero on entry (a point is the neighbour of many queries: atomic adds)
__global__ __launch_bounds__(256) void gather_edge_bwd_kernel(const float* __restrict__ dfeat, int64_t lddf,
                                                              const int64_t* __restrict__ idx,
                                                              const int64_t* __restrict__ cloud_ptr, int64_t slots,
                                                              int64_t per_cloud, int64_t C, float* __restrict__ dx,
                                                              int64_t lddx) {
  const int cx = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= slots) return;
  const int64_t j = idx[e];
  if (j < 0) return;
  float* dst = dx + (cloud_ptr[e / per_cloud] + j) * lddx;
  for (int64_t c = cx; c < C; c += 64) atomicAdd(&dst[c], dfeat[e * lddf + c]);
}

size_t stats_bytes(int64_t rows, int64_t C) { return ccn_align256((size_t)(ccn_stats_rows(rows) + 1) * 2 * C * sizeof(double)); }

int gemm_nt_dtype(int dtype, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                  int64_t M, int64_t N, int64_t K, double* colstats, void* stream) {
  switch (dtype) {
    case CCN_DTYPE_F32: return ccn_gemm_nt(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, stream);
    case CCN_DTYPE_BF16: return ccn_gemm_nt_bf16(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, stream);
    case CCN_DTYPE_F16: return ccn_gemm_nt_f16(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, stream);
  }
  ccn_set_error("linear_bn_act: dtype must be CCN_DTYPE_F32 / _BF16 / _F16");
  return CCN_ERR_ARG;
}

}  // namespace



// ------------------------------------------------------------------ weight transpose for the data-gradient product
int ccn_transpose_pad(const float* W, int64_t ldw, int64_t N, int64_t K, float* Wt, int64_t ldt, void* stream) {
  CCN_REQUIRE(W && Wt && N > 0 && K > 0 && ldw >= K && ldt >= N, "transpose_pad: bad arguments");
  hipLaunchKernelGGL(transpose_pad_kernel, dim3((unsigned)((ldt + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0,
                     (hipStream_t)stream, W, ldw, N, K, Wt, ldt);
  CCN_LAUNCH_OK("transpose_pad");
  return CCN_OK;
}

// ------------------------------------------------------------------ curve convolution
int ccn_curve_conv_fwd(const float* seq, int64_t ld, int64_t rows, int64_t taps, const float* W, int64_t ldw, const float* bias,
                       int64_t Cout, float* Y, int64_t ldy, double* colstats, void* stream) {
  CCN_REQUIRE(seq && W && Y && rows > 0 && taps >= 1 && (taps & 1) && ld % 4 == 0 && ldw >= taps * ld,
              "curve_conv_fwd: bad arguments (odd taps, ld %% 4 == 0, W is C_out x taps*ld)");
  return ccn_conv_rows_nt(seq, ld, W, ldw, bias, Y, ldy, rows, Cout, taps * ld, colstats, stream);
}

size_t ccn_curve_conv_bwd_data_workspace_bytes(int64_t Cin, int64_t taps, int64_t lddy) {
  return ccn_align256((size_t)Cin * taps * lddy * sizeof(float));
}

int ccn_curve_conv_bwd_data(const float* dYseq, int64_t lddy, int64_t rows, int64_t taps, const float* W, int64_t ld_in,
                            int64_t Cout, int64_t Cin, float* dX, int64_t lddx, void* ws, size_t ws_bytes, void* stream) {
  CCN_REQUIRE(dYseq && W && dX && rows > 0 && taps >= 1 && (taps & 1) && lddy % 4 == 0 && lddy >= Cout && ld_in >= Cin &&
                  lddx >= Cin,
              "curve_conv_bwd_data: bad arguments");
  CCN_REQUIRE(ws && ws_bytes >= ccn_curve_conv_bwd_data_workspace_bytes(Cin, taps, lddy), "curve_conv_bwd_data: workspace too small");
  float* Wf = (float*)ws;
  const int64_t total = Cin * taps * lddy;
  hipLaunchKernelGGL(conv_weight_flip_kernel, dim3(ccn_blocks(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, W, ld_in, Cout,
                     Cin, (int)taps, Wf, lddy);
  CCN_LAUNCH_OK("curve_conv_bwd_data");
  // dX[j] = sum_t dY[j + h - t] W_t: the same implicit GEMM over dY with the taps reversed
  return ccn_conv_rows_nt(dYseq, lddy, Wf, taps * lddy, nullptr, dX, lddx, rows, Cin, taps * lddy, nullptr, stream);
}

size_t ccn_curve_conv_bwd_weight_workspace_bytes(int64_t rows, int64_t Cout, int64_t taps, int64_t ld) {
  return ccn_gemm_tn_workspace_bytes(rows, Cout, taps * ld);
}

int ccn_curve_conv_bwd_weight(const float* dY, int64_t lddy, const float* seq, int64_t ld, int64_t rows, int64_t taps,
                              int64_t Cout, float* dW, int64_t lddw, void* ws, size_t ws_bytes, void* stream) {
  CCN_REQUIRE(dY && seq && dW && rows > 0 && taps >= 1 && (taps & 1) && ld % 4 == 0 && lddw >= taps * ld,
              "curve_conv_bwd_weight: bad arguments");
  return ccn_conv_rows_tn(dY, lddy, seq, ld, dW, lddw, rows, Cout, taps * ld, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------ BatchNorm + activation, backward
size_t ccn_bn_act_bwd_workspace_bytes(int64_t rows, int64_t C) { return stats_bytes(rows, C); }

int ccn_bn_act_bwd(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C, const float* params,
                   int act, float slope, int training, float* dY, int64_t lddy, float* dgamma, float* dbeta, void* ws,
                   size_t ws_bytes, void* stream) {
  CCN_REQUIRE(dZ && Y && params && dY && dgamma && dbeta && rows > 0 && C > 0, "bn_act_bwd: bad arguments");
  CCN_REQUIRE(ws && ws_bytes >= stats_bytes(rows, C), "bn_act_bwd: workspace too small");
  const float *scale = params, *shift = params + C, *mean = params + 2 * C, *rstd = params + 3 * C;
  double* sums = (double*)ws;
  int rc = ccn_bn_act_bwd_reduce(dZ, lddz, Y, ldy, rows, C, scale, shift, mean, rstd, act, slope, sums, stream);
  if (rc) return rc;
  return ccn_bn_act_bwd_apply(dZ, lddz, Y, ldy, rows, C, scale, shift, mean, rstd, act, slope, sums, training, dY, lddy, dgamma,
                              dbeta, stream);
}

// ------------------------------------------------------------------ Linear -> BatchNorm -> activation
size_t ccn_linear_bn_act_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const int64_t ldt = (N + 3) / 4 * 4;
  return stats_bytes(M, N) + ccn_align256((size_t)K * ldt * sizeof(float)) + ccn_align256(ccn_gemm_tn_workspace_bytes(M, N, K)) + 256;
}

int ccn_linear_bn_act_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, int64_t M, int64_t N, int64_t K, float eps,
                          float momentum, int training, int act, float slope, int dtype, float* Y, int64_t ldy, float* Z,
                          int64_t ldz, float* params, void* ws, size_t ws_bytes, void* stream) {
  CCN_REQUIRE(X && W && Y && M > 0 && N > 0 && K > 0, "linear_bn_act_fwd: bad arguments");
  if (gamma == nullptr)       // plain Linear (the last layer of an MLP with plain_last)
    return gemm_nt_dtype(dtype, X, ldx, W, ldw, bias, Y, ldy, M, N, K, nullptr, stream);
  CCN_REQUIRE(beta && running_mean && running_var && Z && params, "linear_bn_act_fwd: BatchNorm needs beta, running stats, Z, params");
  float *scale = params, *shift = params + N, *mean = params + 2 * N, *rstd = params + 3 * N;
  int rc;
  if (training) {
    CCN_REQUIRE(M >= 2, "Expected more than 1 value per channel when training");
    CCN_REQUIRE(ws && ws_bytes >= stats_bytes(M, N), "linear_bn_act_fwd: workspace too small");
    double* stats = (double*)ws;
    rc = gemm_nt_dtype(dtype, X, ldx, W, ldw, bias, Y, ldy, M, N, K, stats, stream);
    if (rc) return rc;
    rc = ccn_bn_finalize(stats, M, N, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, rstd, stream);
  } else {
    rc = gemm_nt_dtype(dtype, X, ldx, W, ldw, bias, Y, ldy, M, N, K, nullptr, stream);
    if (rc) return rc;
    rc = ccn_bn_eval_params(gamma, beta, running_mean, running_var, eps, N, scale, shift, mean, rstd, stream);
  }
  if (rc) return rc;
  return ccn_bn_act_fwd(Y, ldy, M, N, scale, shift, act, slope, Z, ldz, stream);
}

int ccn_linear_bn_act_bwd(const float* dZ, int64_t lddz, const float* X, int64_t ldx, const float* W, int64_t ldw, const float* Y,
                          int64_t ldy, const float* params, int64_t M, int64_t N, int64_t K, int training, int act, float slope,
                          int dtype, float* dY, int64_t lddy, float* dX, int64_t lddx, float* dW, int64_t lddw, float* dbias,
                          float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
  CCN_REQUIRE(dZ && X && W && M > 0 && N > 0 && K > 0, "linear_bn_act_bwd: bad arguments");
  CCN_REQUIRE(ws && ws_bytes >= ccn_linear_bn_act_workspace_bytes(M, N, K), "linear_bn_act_bwd: workspace too small");
  CcnArena a(ws, ws_bytes);
  const int64_t ldt = (N + 3) / 4 * 4;
  double* sums = (double*)a.take<char>(stats_bytes(M, N));
  float* Wt = a.take<float>((size_t)K * ldt);
  const size_t tn_bytes = ccn_gemm_tn_workspace_bytes(M, N, K);
  void* tn_ws = a.take<char>(tn_bytes);
  CCN_REQUIRE(a.ok(), "linear_bn_act_bwd: workspace carve failed");
  int rc;
  const float* g = dZ;       // gradient with respect to the product X W^T + b
  int64_t ldg = lddz;
  if (params != nullptr) {
    CCN_REQUIRE(Y && dY && dgamma && dbeta, "linear_bn_act_bwd: BatchNorm needs Y, dY, dgamma, dbeta");
    rc = ccn_bn_act_bwd(dZ, lddz, Y, ldy, M, N, params, act, slope, training, dY, lddy, dgamma, dbeta, sums, stats_bytes(M, N), stream);
    if (rc) return rc;
    g = dY;
    ldg = lddy;
  }
  if (dX != nullptr) {
    // dX = g W as an NT product with W^T (K x N): both operands stream along their contiguous index
    rc = ccn_transpose_pad(W, ldw, N, K, Wt, ldt, stream);
    if (rc) return rc;
    rc = gemm_nt_dtype(dtype == CCN_DTYPE_F16 ? CCN_DTYPE_BF16 : dtype, g, ldg, Wt, ldt, nullptr, dX, lddx, M, K, N, nullptr, stream);
    if (rc) return rc;
  }
  if (dW != nullptr) {       // dW += g^T X
    rc = dtype == CCN_DTYPE_F32 ? ccn_gemm_tn_ws(g, ldg, X, ldx, dW, lddw, M, N, K, tn_ws, tn_bytes, stream)
                                : ccn_gemm_tn_bf16(g, ldg, X, ldx, dW, lddw, M, N, K, stream);
    if (rc) return rc;
  }
  if (dbias != nullptr) {
    rc = ccn_colsum(g, ldg, M, N, sums, dbias, stream);
    if (rc) return rc;
  }
  return CCN_OK;
}

// ------------------------------------------------------------------ edge gather and reductions
int ccn_gather_edge_fwd(const float* x, int64_t ldx, const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax,
                        int64_t K, int64_t C, float* feat, int64_t ldf, void* stream) {
  CCN_REQUIRE(x && idx && cloud_ptr && feat && B > 0 && Nmax > 0 && K > 0 && C > 0 && ldx >= C && ldf >= C,
              "gather_edge_fwd: bad arguments");
  hipLaunchKernelGGL(gather_edge_fwd_kernel, dim3(ccn_blocks(B * Nmax * K, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, idx,
                     cloud_ptr, B * Nmax * K, Nmax * K, C, feat, ldf);
  CCN_LAUNCH_OK("gather_edge_fwd");
  return CCN_OK;
}

int ccn_gather_edge_bwd(const float* dfeat, int64_t lddf, const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax,
                        int64_t K, int64_t C, float* dx, int64_t lddx, void* stream) {
  CCN_REQUIRE(dfeat && idx && cloud_ptr && dx && B > 0 && Nmax > 0 && K > 0 && C > 0 && lddf >= C && lddx >= C,
              "gather_edge_bwd: bad arguments");
  hipLaunchKernelGGL(gather_edge_bwd_kernel, dim3(ccn_blocks(B * Nmax * K, 4)), dim3(256), 0, (hipStream_t)stream, dfeat, lddf,
                     idx, cloud_ptr, B * Nmax * K, Nmax * K, C, dx, lddx);
  CCN_LAUNCH_OK("gather_edge_bwd");
  return CCN_OK;
}

int ccn_edge_reduce_max_fwd(const float* msg, int64_t ldm, const int32_t* offsets, int64_t M, int64_t C, float* out, int64_t ldo,
                            int32_t* arg, void* stream) {
  return ccn_seg_max_fwd(msg, ldm, offsets, M, C, out, ldo, arg, stream);
}

int ccn_edge_reduce_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* offsets, int64_t M, int64_t C,
                            float* dmsg, int64_t lddm, void* stream) {
  return ccn_seg_max_bwd(dout, lddo, arg, offsets, M, C, dmsg, lddm, stream);
}

int ccn_edge_reduce_attend_fwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets, int64_t M,
                               int64_t C, float* out, int64_t ldo, void* stream) {
  return ccn_seg_softmax_agg_fwd(msg, ldm, att, lda, offsets, M, C, out, ldo, stream);
}

int ccn_edge_reduce_attend_bwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets, int64_t M,
                               int64_t C, const float* dout, int64_t lddo, float* dmsg, int64_t lddm, float* datt, int64_t ldda,
                               void* stream) {
  return ccn_seg_softmax_agg_bwd(msg, ldm, att, lda, offsets, M, C, dout, lddo, dmsg, lddm, datt, ldda, stream);
}



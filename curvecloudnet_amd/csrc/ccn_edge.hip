// Edge-feature construction and neighbourhood aggregation.
// All kernels are HBM-bound gathers / segmented reductions with one thread per (row, channel) and the
// channel index fastest, so every wave touches whole contiguous feature rows.
#include "ccn_common.h"

namespace {

constexpr int TPB = 256;

// ------------------------------------------------------------------ A15: dense SGCNN path
// dense row e = (b*Nmax + i)*(K+1) + s ; slot 0 is the self loop, slot s>0 is FRNN neighbour s-1
__device__ __forceinline__ int64_t sg_neighbour(const int64_t* __restrict__ idx, int64_t b, int64_t i, int s,
                                                int64_t Nmax, int64_t K, int64_t len) {
  if (i >= len) return -1;
  if (s == 0) return i;
  return idx[(b * Nmax + i) * K + (s - 1)];
}

__global__ void sg_gather_fwd_kernel(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ idx,
                                     const int64_t* __restrict__ cloud_ptr, int64_t B, int64_t Nmax, int64_t K,
                                     int64_t C, float* __restrict__ feat) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t K1 = K + 1;
  if (t >= B * Nmax * K1 * C) return;
  const int64_t e = t / C, c = t - e * C;
  const int64_t bi = e / K1;
  const int s = (int)(e - bi * K1);
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  const int64_t j = sg_neighbour(idx, b, i, s, Nmax, K, len);
  const float self = i < len ? x[(base + i) * ldx + c] : 0.f;
  const float g = j >= 0 ? x[(base + j) * ldx + c] : 0.f;
  feat[e * 2 * C + c] = g;
  feat[e * 2 * C + C + c] = self - g;
}

// dx[n] = sum over the K+1 slots of d(second half) + slot-0 terms (plain stores: initialises dx)
__global__ void sg_gather_bwd_self_kernel(const float* __restrict__ dfeat, const int64_t* __restrict__ cloud_ptr,
                                          int64_t B, int64_t Nmax, int64_t K, int64_t C, float* __restrict__ dx,
                                          int64_t lddx) {
  const int64_t b = blockIdx.y;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= len * C) return;
  const int64_t i = t / C, c = t - i * C;
  const int64_t K1 = K + 1;
  const float* row = dfeat + ((b * Nmax + i) * K1) * 2 * C;
  float acc = row[c] - row[C + c];  // slot 0 gathers x_i itself: d/dx_i of [x_i, x_i - x_i]
  for (int64_t s = 0; s < K1; ++s) acc += row[s * 2 * C + C + c];
  dx[(base + i) * lddx + c] = acc;
}

__global__ void sg_gather_bwd_nbr_kernel(const float* __restrict__ dfeat, const int64_t* __restrict__ idx,
                                         const int64_t* __restrict__ cloud_ptr, int64_t B, int64_t Nmax, int64_t K,
                                         int64_t C, float* __restrict__ dx, int64_t lddx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * Nmax * K * C) return;
  const int64_t e = t / C, c = t - e * C;  // e over (b, i, s-1)
  const int64_t bi = e / K, s1 = e - bi * K;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  if (i >= len) return;
  const int64_t j = idx[bi * K + s1];
  if (j < 0) return;
  const float* row = dfeat + (bi * (K + 1) + s1 + 1) * 2 * C;
  atomicAdd(&dx[(base + j) * lddx + c], row[c] - row[C + c]);
}

__global__ void sg_max_fwd_kernel(const float* __restrict__ f, const int64_t* __restrict__ idx,
                                  const int64_t* __restrict__ cloud_ptr, int64_t B, int64_t Nmax, int64_t K, int64_t C,
                                  float* __restrict__ out, int64_t ldo, int32_t* __restrict__ arg) {
  const int64_t b = blockIdx.y;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= len * C) return;
  const int64_t i = t / C, c = t - i * C;
  const int64_t K1 = K + 1;
  const float* row = f + ((b * Nmax + i) * K1) * C;
  float best = row[c];  // slot 0 (self) is always valid
  int at = 0;
  for (int64_t s = 1; s < K1; ++s) {
    const bool ok = idx[(b * Nmax + i) * K + (s - 1)] != -1;
    const float v = ok ? row[s * C + c] : -1e2f;
    if (v > best) {
      best = v;
      at = ok ? (int)s : -1;
    }
  }
  out[(base + i) * ldo + c] = best;
  arg[(base + i) * C + c] = at;
}

__global__ void sg_max_bwd_kernel(const float* __restrict__ dout, int64_t lddo, const int32_t* __restrict__ arg,
                                  const int64_t* __restrict__ cloud_ptr, int64_t B, int64_t Nmax, int64_t K, int64_t C,
                                  float* __restrict__ df) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t K1 = K + 1;
  if (t >= B * Nmax * K1 * C) return;
  const int64_t e = t / C, c = t - e * C;
  const int64_t bi = e / K1;
  const int s = (int)(e - bi * K1);
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  float v = 0.f;
  if (i < len && arg[(base + i) * C + c] == s) v = dout[(base + i) * lddo + c];
  df[t] = v;
}

// ------------------------------------------------------------------ A13: PointNetConv2 message
__global__ void msg_build_fwd_kernel(const float* __restrict__ x_src, int64_t ldx, const float* __restrict__ pos_src,
                                     const float* __restrict__ pos_dst, const int64_t* __restrict__ src,
                                     const int64_t* __restrict__ dst, int64_t E, int64_t C, float radius,
                                     float* __restrict__ msg) {
  const int64_t W = C + 3;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= E * W) return;
  const int64_t e = t / W, c = t - e * W;
  const int64_t j = src[e];
  float v;
  if (c < C) {
    v = x_src[j * ldx + c];
  } else {
    const int d = (int)(c - C);
    v = pos_src[3 * j + d] - pos_dst[3 * dst[e] + d];
    if (radius > 0.f) v = __fdiv_rn(v, radius);
  }
  msg[t] = v;
}

__global__ void msg_build_bwd_kernel(const float* __restrict__ dmsg, const int64_t* __restrict__ src, int64_t E,
                                     int64_t C, float* __restrict__ dx, int64_t lddx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= E * C) return;
  const int64_t e = t / C, c = t - e * C;
  atomicAdd(&dx[src[e] * lddx + c], dmsg[e * (C + 3) + c]);
}

// ------------------------------------------------------------------ grouped (CSR) aggregation
__global__ void seg_softmax_agg_fwd_kernel(const float* __restrict__ msg, const float* __restrict__ att,
                                           const int32_t* __restrict__ offsets, int64_t M, int64_t C,
                                           float* __restrict__ out, int64_t ldo) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int64_t i = t / C, c = t - i * C;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  float top = -__builtin_inff();
  for (int32_t e = lo; e < hi; ++e) top = fmaxf(top, att[(int64_t)e * C + c]);
  float tot = 0.f;
  for (int32_t e = lo; e < hi; ++e) tot += __expf(att[(int64_t)e * C + c] - top);
  const float inv = 1.0f / (tot + 1e-16f);
  float acc = 0.f;
  for (int32_t e = lo; e < hi; ++e)
    acc += msg[(int64_t)e * C + c] * (__expf(att[(int64_t)e * C + c] - top) * inv);
  out[i * ldo + c] = acc;
}

__global__ void seg_softmax_agg_bwd_kernel(const float* __restrict__ msg, const float* __restrict__ att,
                                           const int32_t* __restrict__ offsets, int64_t M, int64_t C,
                                           const float* __restrict__ dout, int64_t lddo, float* __restrict__ dmsg,
                                           float* __restrict__ datt) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int64_t i = t / C, c = t - i * C;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  const float g = dout[i * lddo + c];
  float top = -__builtin_inff();
  for (int32_t e = lo; e < hi; ++e) top = fmaxf(top, att[(int64_t)e * C + c]);
  float tot = 0.f;
  for (int32_t e = lo; e < hi; ++e) tot += __expf(att[(int64_t)e * C + c] - top);
  const float inv = 1.0f / (tot + 1e-16f);
  float dot = 0.f;  // sum_e w_e * dL/dw_e
  for (int32_t e = lo; e < hi; ++e) {
    const float w = __expf(att[(int64_t)e * C + c] - top) * inv;
    dot += w * msg[(int64_t)e * C + c] * g;
  }
  for (int32_t e = lo; e < hi; ++e) {
    const float w = __expf(att[(int64_t)e * C + c] - top) * inv;
    dmsg[(int64_t)e * C + c] = w * g;
    datt[(int64_t)e * C + c] = w * (msg[(int64_t)e * C + c] * g - dot);
  }
}

__global__ void seg_max_fwd_kernel(const float* __restrict__ msg, const int32_t* __restrict__ offsets, int64_t M,
                                   int64_t C, float* __restrict__ out, int64_t ldo, int32_t* __restrict__ arg) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int64_t i = t / C, c = t - i * C;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  float best = 0.f;  // scatter_max: empty groups give 0
  int at = -1;
  for (int32_t e = lo; e < hi; ++e) {
    const float v = msg[(int64_t)e * C + c];
    if (at < 0 || v > best) {
      best = v;
      at = e - lo;
    }
  }
  out[i * ldo + c] = best;
  arg[i * C + c] = at;
}

__global__ void seg_max_bwd_kernel(const float* __restrict__ dout, int64_t lddo, const int32_t* __restrict__ arg,
                                   const int32_t* __restrict__ offsets, int64_t M, int64_t C,
                                   float* __restrict__ dmsg) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int64_t i = t / C, c = t - i * C;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  const int at = arg[i * C + c];
  const float g = dout[i * lddo + c];
  for (int32_t e = lo; e < hi; ++e) dmsg[(int64_t)e * C + c] = (e - lo == at) ? g : 0.f;
}

}  // namespace

extern "C" {

int ccn_sg_gather_fwd(const float* x, int64_t ldx, const int64_t* idx, const int64_t* cloud_ptr, int64_t B,
                      int64_t Nmax, int64_t K, int64_t C, float* feat, void* stream) {
  CCN_REQUIRE(x && idx && cloud_ptr && feat && B > 0 && Nmax > 0 && K > 0 && C > 0 && ldx >= C,
              "sg_gather_fwd: bad arguments");
  const int64_t total = B * Nmax * (K + 1) * C;
  hipLaunchKernelGGL(sg_gather_fwd_kernel, dim3(ccn_blocks(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, x, ldx,
                     idx, cloud_ptr, B, Nmax, K, C, feat);
  CCN_LAUNCH_OK("sg_gather_fwd");
  return CCN_OK;
}

int ccn_sg_gather_bwd(const float* dfeat, const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax,
                      int64_t K, int64_t C, float* dx, int64_t lddx, void* stream) {
  CCN_REQUIRE(dfeat && idx && cloud_ptr && dx && B > 0 && B < 65536 && Nmax > 0 && K > 0 && C > 0 && lddx >= C,
              "sg_gather_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sg_gather_bwd_self_kernel, dim3(ccn_blocks(Nmax * C, TPB), (unsigned)B), dim3(TPB), 0, s, dfeat,
                     cloud_ptr, B, Nmax, K, C, dx, lddx);
  hipLaunchKernelGGL(sg_gather_bwd_nbr_kernel, dim3(ccn_blocks(B * Nmax * K * C, TPB)), dim3(TPB), 0, s, dfeat, idx,
                     cloud_ptr, B, Nmax, K, C, dx, lddx);
  CCN_LAUNCH_OK("sg_gather_bwd");
  return CCN_OK;
}

int ccn_sg_max_fwd(const float* f, const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax, int64_t K,
                   int64_t C, float* out, int64_t ldo, int32_t* arg, void* stream) {
  CCN_REQUIRE(f && idx && cloud_ptr && out && arg && B > 0 && B < 65536 && Nmax > 0 && K > 0 && C > 0 && ldo >= C,
              "sg_max_fwd: bad arguments");
  hipLaunchKernelGGL(sg_max_fwd_kernel, dim3(ccn_blocks(Nmax * C, TPB), (unsigned)B), dim3(TPB), 0,
                     (hipStream_t)stream, f, idx, cloud_ptr, B, Nmax, K, C, out, ldo, arg);
  CCN_LAUNCH_OK("sg_max_fwd");
  return CCN_OK;
}

int ccn_sg_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int64_t* cloud_ptr, int64_t B,
                   int64_t Nmax, int64_t K, int64_t C, float* df, void* stream) {
  CCN_REQUIRE(dout && arg && cloud_ptr && df && B > 0 && Nmax > 0 && K > 0 && C > 0 && lddo >= C,
              "sg_max_bwd: bad arguments");
  const int64_t total = B * Nmax * (K + 1) * C;
  hipLaunchKernelGGL(sg_max_bwd_kernel, dim3(ccn_blocks(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, dout, lddo,
                     arg, cloud_ptr, B, Nmax, K, C, df);
  CCN_LAUNCH_OK("sg_max_bwd");
  return CCN_OK;
}

int ccn_msg_build_fwd(const float* x_src, int64_t ldx, const float* pos_src, const float* pos_dst, const int64_t* src,
                      const int64_t* dst, int64_t E, int64_t C, float radius, float* msg, void* stream) {
  CCN_REQUIRE(pos_src && pos_dst && src && dst && msg && C >= 0 && (C == 0 || (x_src && ldx >= C)),
              "msg_build_fwd: bad arguments");
  if (E == 0) return CCN_OK;
  hipLaunchKernelGGL(msg_build_fwd_kernel, dim3(ccn_blocks(E * (C + 3), TPB)), dim3(TPB), 0, (hipStream_t)stream,
                     x_src, ldx, pos_src, pos_dst, src, dst, E, C, radius, msg);
  CCN_LAUNCH_OK("msg_build_fwd");
  return CCN_OK;
}

int ccn_msg_build_bwd(const float* dmsg, const int64_t* src, int64_t E, int64_t C, float* dx, int64_t lddx,
                      void* stream) {
  CCN_REQUIRE(dmsg && src && dx && C > 0 && lddx >= C, "msg_build_bwd: bad arguments");
  if (E == 0) return CCN_OK;
  hipLaunchKernelGGL(msg_build_bwd_kernel, dim3(ccn_blocks(E * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, dmsg, src,
                     E, C, dx, lddx);
  CCN_LAUNCH_OK("msg_build_bwd");
  return CCN_OK;
}

int ccn_seg_softmax_agg_fwd(const float* msg, const float* att, const int32_t* offsets, int64_t M, int64_t C,
                            float* out, int64_t ldo, void* stream) {
  CCN_REQUIRE(msg && att && offsets && out && C > 0 && ldo >= C, "seg_softmax_agg_fwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_softmax_agg_fwd_kernel, dim3(ccn_blocks(M * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, msg,
                     att, offsets, M, C, out, ldo);
  CCN_LAUNCH_OK("seg_softmax_agg_fwd");
  return CCN_OK;
}

int ccn_seg_softmax_agg_bwd(const float* msg, const float* att, const int32_t* offsets, int64_t M, int64_t C,
                            const float* dout, int64_t lddo, float* dmsg, float* datt, void* stream) {
  CCN_REQUIRE(msg && att && offsets && dout && dmsg && datt && C > 0 && lddo >= C, "seg_softmax_agg_bwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_softmax_agg_bwd_kernel, dim3(ccn_blocks(M * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, msg,
                     att, offsets, M, C, dout, lddo, dmsg, datt);
  CCN_LAUNCH_OK("seg_softmax_agg_bwd");
  return CCN_OK;
}

int ccn_seg_max_fwd(const float* msg, const int32_t* offsets, int64_t M, int64_t C, float* out, int64_t ldo,
                    int32_t* arg, void* stream) {
  CCN_REQUIRE(msg && offsets && out && arg && C > 0 && ldo >= C, "seg_max_fwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_max_fwd_kernel, dim3(ccn_blocks(M * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, msg, offsets,
                     M, C, out, ldo, arg);
  CCN_LAUNCH_OK("seg_max_fwd");
  return CCN_OK;
}

int ccn_seg_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* offsets, int64_t M, int64_t C,
                    float* dmsg, void* stream) {
  CCN_REQUIRE(dout && arg && offsets && dmsg && C > 0 && lddo >= C, "seg_max_bwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_max_bwd_kernel, dim3(ccn_blocks(M * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, dout, lddo,
                     arg, offsets, M, C, dmsg);
  CCN_LAUNCH_OK("seg_max_bwd");
  return CCN_OK;
}

}  // extern "C"

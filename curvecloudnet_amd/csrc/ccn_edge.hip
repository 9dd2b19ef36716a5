// Edge-feature construction and neighbourhood aggregation.
// All kernels are HBM-bound gathers / segmented reductions.  Thread layout: a 256-thread workgroup is
// 4 waves; wave `ry` owns whole rows (points / edges / slots), its 64 lanes `cx` stride over the
// channels, so every memory instruction touches 256 contiguous bytes of one feature row and the
// per-row index arithmetic is wave-uniform (no per-element integer division).
#include "ccn_common.h"

namespace {

constexpr int TPB = 256;
constexpr int ROWS_PER_WG = 4;  // one row per wave at a time

#define CCN_LANES const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6

// Neighbour list of one dense row held across the wave: lane s keeps the neighbour of slot s+1 (one coalesced load per
// point instead of a dependent scalar load in front of every gathered row), read back with a wave-uniform lane
// index.  K > 64 falls back to direct loads.
struct SgNbrs {
  const int64_t* p;
  int64_t v;
  int K;
};
__device__ __forceinline__ SgNbrs sg_nbrs(const int64_t* __restrict__ idx, int64_t bi, int K, int lane, bool live) {
  SgNbrs n;
  n.p = idx + bi * K;
  n.K = K;
  n.v = (live && K <= 64 && lane < K) ? n.p[lane] : -1;
  return n;
}
// slot s >= 1
__device__ __forceinline__ int64_t sg_nbr(const SgNbrs& n, int s) {
  if (n.K <= 64) {
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)n.v & 0xffffffffu), s - 1);
    const int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)n.v >> 32), s - 1);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
  }
  return n.p[s - 1];
}
constexpr int SG_UNROLL = 4;  // gathered rows in flight per lane

// ------------------------------------------------------------------ A15: dense SGCNN path
// dense row e = (b*Nmax + i)*(K+1) + s ; slot 0 is the self loop, slot s>0 is FRNN neighbour s-1.
// One wave per (b, i): loops over the K+1 slots; the self row is read once per channel.
__global__ __launch_bounds__(TPB) void sg_gather_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const int64_t* __restrict__ idx,
                                                            const int64_t* __restrict__ cloud_ptr, int64_t B,
                                                            int64_t Nmax, int K, int C, float* __restrict__ feat,
                                                            int64_t ldf) {
  CCN_LANES;
  const int64_t bi = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (bi >= B * Nmax) return;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  const bool live = i < len;
  const float* xs = x + (base + (live ? i : 0)) * ldx;
  float* out = feat + bi * (K + 1) * ldf;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cx;
    const bool in = c < C;
    const float self = (live && in) ? xs[c] : 0.f;
    for (int s = 0; s <= K; ++s) {
      int64_t j = -1;
      if (live) j = s == 0 ? i : idx[bi * K + (s - 1)];
      if (in) {
        const float g = j >= 0 ? x[(base + j) * ldx + c] : 0.f;
        out[s * ldf + c] = g;
        out[s * ldf + C + c] = self - g;
      }
    }
  }
}

// dx must be zero on entry: both the self term and the neighbour terms are added atomically.
__global__ __launch_bounds__(TPB) void sg_gather_bwd_kernel(const float* __restrict__ dfeat, int64_t lddf,
                                                            const int64_t* __restrict__ idx,
                                                            const int64_t* __restrict__ cloud_ptr, int64_t B,
                                                            int64_t Nmax, int K, int C, float* __restrict__ dx,
                                                            int64_t lddx) {
  CCN_LANES;
  const int64_t bi = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (bi >= B * Nmax) return;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  if (i >= len) return;
  const float* d = dfeat + bi * (K + 1) * lddf;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cx;
    if (c >= C) continue;
    float self = d[c] - d[C + c];  // slot 0 gathers x_i itself
    for (int s = 0; s <= K; ++s) {
      const float second = d[s * lddf + C + c];
      self += second;
      if (s > 0) {
        const int64_t j = idx[bi * K + (s - 1)];
        if (j >= 0) atomicAdd(&dx[(base + j) * lddx + c], d[s * lddf + c] - second);
      }
    }
    atomicAdd(&dx[(base + i) * lddx + c], self);
  }
}

__global__ __launch_bounds__(TPB) void sg_max_fwd_kernel(const float* __restrict__ f, int64_t ldf,
                                                         const int64_t* __restrict__ idx,
                                                         const int64_t* __restrict__ cloud_ptr, int64_t B,
                                                         int64_t Nmax, int K, int C, float* __restrict__ out,
                                                         int64_t ldo, int32_t* __restrict__ arg) {
  CCN_LANES;
  const int64_t bi = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (bi >= B * Nmax) return;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  if (i >= len) return;
  const SgNbrs nb = sg_nbrs(idx, bi, K, cx, true);
  const float* row = f + bi * (K + 1) * ldf;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cx;
    const int cc = c < C ? c : C - 1;
    float best = row[cc];  // slot 0 (self) is always valid
    int at = 0;
    for (int s0 = 1; s0 <= K; s0 += SG_UNROLL) {
      float v[SG_UNROLL];
      bool ok[SG_UNROLL];
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        const int s = s0 + u;
        ok[u] = s <= K && sg_nbr(nb, s <= K ? s : K) != -1;
        v[u] = ok[u] ? row[s * ldf + cc] : -1e2f;
      }
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        if (s0 + u <= K && v[u] > best) {
          best = v[u];
          at = ok[u] ? s0 + u : -1;
        }
      }
    }
    if (c < C) {
      out[(base + i) * ldo + c] = best;
      arg[(base + i) * C + c] = at;
    }
  }
}

__global__ __launch_bounds__(TPB) void sg_max_bwd_kernel(const float* __restrict__ dout, int64_t lddo,
                                                         const int32_t* __restrict__ arg,
                                                         const int64_t* __restrict__ cloud_ptr, int64_t B,
                                                         int64_t Nmax, int K, int C, float* __restrict__ df,
                                                         int64_t lddf) {
  CCN_LANES;
  const int64_t bi = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (bi >= B * Nmax) return;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  const bool live = i < len;
  float* row = df + bi * (K + 1) * lddf;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cx;
    if (c >= C) continue;
    const int at = live ? arg[(base + i) * C + c] : -2;
    const float g = live ? dout[(base + i) * lddo + c] : 0.f;
    for (int s = 0; s <= K; ++s) row[s * lddf + c] = (s == at) ? g : 0.f;
  }
}

// ------------------------------------------------------------------ A15, algebraic form of the first edge layer
// W [x_j ; x_i - x_j] = (Wa - Wb) x_j + Wb x_i  =>  y(b,i,s) = P[j] + S[i] with the per-point products
// PS = X [Wa-Wb ; Wb]^T (N x 2*Co: P = columns [0,Co), S = columns [Co,2Co)).  The 21x larger edge GEMM
// and the 2C-wide edge tensor disappear; what is left per dense row is a gather-add.
// Quirk Q4 is kept: statistics run over all B*Nmax*(K+1) rows (missing neighbours contribute S[i],
// padding rows contribute `pad` = the bias, i.e. what a zero input row gives).
constexpr int SG_PTS_MAX = 32;  // points per wave (fewer on small levels so that the grid still fills the chip)

__device__ __forceinline__ float edge_act(float z, int act, float slope) {
  if (act == CCN_ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == CCN_ACT_LEAKY) return z > 0.f ? z : z * slope;
  return z;
}
__device__ __forceinline__ float edge_act_grad(float z, int act, float slope) {
  if (act == CCN_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == CCN_ACT_LEAKY) return z > 0.f ? 1.f : slope;
  return 1.f;
}

// MODE 0: column sums of y and y^2          (forward batch statistics)
// MODE 1: column sums of g and g*xhat       (backward, g = dZ * act'(y*scale+shift))
template <int MODE>
__global__ __launch_bounds__(TPB) void sg_edge_stats_kernel(
    const float* __restrict__ ps, int64_t ldps, const float* __restrict__ pad, const int64_t* __restrict__ idx,
    const int64_t* __restrict__ cloud_ptr, int64_t B, int64_t Nmax, int K, int Co, const float* __restrict__ dZ,
    int64_t lddz, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int act, float slope, int pts, double* __restrict__ partial) {
  __shared__ double red[4][64][2];
  CCN_LANES;
  const int64_t first = ((int64_t)ccn_xcd_block() * 4 + ry) * pts;
  {
    const int c = blockIdx.y * 64 + cx;  // one 64-channel chunk per workgroup
    double s1 = 0.0, s2 = 0.0;
    // all 64 lanes walk the points (the neighbour list lives across the wave); lanes past Co only skip the sums
    const int cc = c < Co ? c : Co - 1;
    {
      const float padv = pad ? pad[cc] : 0.f;
      float sc = 0.f, sh = 0.f, mu = 0.f, rs = 0.f;
      if (MODE == 1) {
        sc = scale[cc];
        sh = shift[cc];
        mu = mean[cc];
        rs = rstd[cc];
      }
      for (int t = 0; t < pts; ++t) {
        const int64_t bi = first + t;
        if (bi >= B * Nmax) break;
        const int64_t b = bi / Nmax, i = bi - b * Nmax;
        const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
        const bool live = i < len;
        const SgNbrs nb = sg_nbrs(idx, bi, K, cx, live);
        const float si = live ? ps[(base + i) * ldps + Co + cc] : padv;
        for (int s0 = 0; s0 <= K; s0 += SG_UNROLL) {
          float pv[SG_UNROLL], dz[SG_UNROLL];
#pragma unroll
          for (int u = 0; u < SG_UNROLL; ++u) {
            const int s = s0 + u;
            int64_t j = -1;
            if (live && s <= K) j = s == 0 ? i : sg_nbr(nb, s);
            pv[u] = j >= 0 ? ps[(base + j) * ldps + cc] : 0.f;
            if (MODE == 1) dz[u] = s <= K ? dZ[(bi * (K + 1) + s) * lddz + cc] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < SG_UNROLL; ++u) {
            if (s0 + u > K) continue;
            const float y = pv[u] + si;
            if (MODE == 0) {
              s1 += (double)y;
              s2 += (double)y * (double)y;
            } else {
              const float g = dz[u] * edge_act_grad(y * sc + sh, act, slope);
              s1 += (double)g;
              s2 += (double)(g * ((y - mu) * rs));
            }
          }
        }
      }
    }
    red[ry][cx][0] = s1;
    red[ry][cx][1] = s2;
    __syncthreads();
    if (ry == 0 && c < Co) {
      double a = 0.0, b2 = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += red[w][cx][0];
        b2 += red[w][cx][1];
      }
      partial[(int64_t)ccn_xcd_block() * 2 * Co + c] = a;
      partial[(int64_t)ccn_xcd_block() * 2 * Co + Co + c] = b2;
    }
  }
}

__global__ __launch_bounds__(TPB) void sg_edge_apply_kernel(const float* __restrict__ ps, int64_t ldps,
                                                            const float* __restrict__ pad,
                                                            const int64_t* __restrict__ idx,
                                                            const int64_t* __restrict__ cloud_ptr, int64_t B,
                                                            int64_t Nmax, int K, int Co,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int act, float slope,
                                                            float* __restrict__ Z, int64_t ldz) {
  CCN_LANES;
  const int64_t bi = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (bi >= B * Nmax) return;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  const bool live = i < len;
  float* out = Z + bi * (K + 1) * ldz;
  const SgNbrs nb = sg_nbrs(idx, bi, K, cx, live);
  for (int c0 = 0; c0 < Co; c0 += 64) {
    const int c = c0 + cx, cc = c < Co ? c : Co - 1;
    const float si = live ? ps[(base + i) * ldps + Co + cc] : (pad ? pad[cc] : 0.f);
    const float sc = scale ? scale[cc] : 1.f, sh = shift ? shift[cc] : 0.f;
    for (int s0 = 0; s0 <= K; s0 += SG_UNROLL) {
      float pv[SG_UNROLL];
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        const int s = s0 + u;
        int64_t j = -1;
        if (live && s <= K) j = s == 0 ? i : sg_nbr(nb, s);
        pv[u] = j >= 0 ? ps[(base + j) * ldps + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u)
        if (s0 + u <= K && c < Co) out[(s0 + u) * ldz + c] = edge_act((pv[u] + si) * sc + sh, act, slope);
    }
  }
}

// dPS must be zero on entry.  dS[i] is owned by point i (plain store), dP[j] is accumulated atomically.
__global__ __launch_bounds__(TPB) void sg_edge_bwd_kernel(
    const float* __restrict__ ps, int64_t ldps, const float* __restrict__ pad, const int64_t* __restrict__ idx,
    const int64_t* __restrict__ cloud_ptr, int64_t B, int64_t Nmax, int K, int Co, const float* __restrict__ dZ,
    int64_t lddz, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int act, float slope, const double* __restrict__ sums, int64_t rows_total,
    int training, float* __restrict__ dps, int64_t lddps) {
  CCN_LANES;
  const int64_t bi = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (bi >= B * Nmax) return;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  if (i >= len) return;  // padding rows feed nothing upstream
  const float inv_n = 1.0f / (float)rows_total;
  const SgNbrs nb = sg_nbrs(idx, bi, K, cx, true);
  for (int c0 = 0; c0 < Co; c0 += 64) {
    const int c = c0 + cx, cc = c < Co ? c : Co - 1;
    const float si = ps[(base + i) * ldps + Co + cc];
    const float sc = scale ? scale[cc] : 1.f, sh = shift ? shift[cc] : 0.f;
    const float mu = mean ? mean[cc] : 0.f, rs = rstd ? rstd[cc] : 0.f;
    const float m1 = (training && sums) ? (float)sums[cc] * inv_n : 0.f;
    const float m2 = (training && sums) ? (float)sums[Co + cc] * inv_n : 0.f;
    float ds = 0.f;
    for (int s0 = 0; s0 <= K; s0 += SG_UNROLL) {
      float pv[SG_UNROLL], dz[SG_UNROLL];
      int64_t jj[SG_UNROLL];
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        const int s = s0 + u;
        jj[u] = s > K ? -1 : (s == 0 ? i : sg_nbr(nb, s));
        pv[u] = jj[u] >= 0 ? ps[(base + jj[u]) * ldps + cc] : 0.f;
        dz[u] = s <= K ? dZ[(bi * (K + 1) + s) * lddz + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        if (s0 + u > K) continue;
        const float y = pv[u] + si;
        const float g = dz[u] * edge_act_grad(y * sc + sh, act, slope);
        const float dy = training ? sc * (g - m1 - (y - mu) * rs * m2) : sc * g;
        ds += dy;
        if (jj[u] >= 0 && c < Co) atomicAdd(&dps[(base + jj[u]) * lddps + c], dy);
      }
    }
    if (c < Co) dps[(base + i) * lddps + Co + c] = ds;
  }
}

// ------------------------------------------------------------------ A15 on COMPACT rows
// In the dense layout every point owns K+1 rows, but a slot whose FRNN entry is -1 carries no neighbour: its first-layer
// value is S[i] whatever the slot, so all empty slots of a point are the SAME row through the whole per-row MLP, and so
// are all rows of the padding points (value `0`).  They matter only through the BatchNorm statistics (quirk Q4), which
// are sums.  The compact layout keeps the real rows (self + found neighbours, grouped by point), ONE representative
// row per point that has empty slots (weight = number of empty slots) and ONE row for all padding rows (weight = their
// count); every reduction over rows takes the weights into account, so the result equals the dense computation
// while the GEMMs run over 23-85 % of the rows (KITTI bench levels).
//   rows [0, E): real, point p owns [grp_ptr[p], grp_ptr[p+1]) (self first, then the neighbours in FRNN order)
//   rows [E, E+Ne): representatives, rep_row[p] = row of point p or -1;  row E+Ne: padding row;  row_w: weights of [E, R)
__global__ void cg_count_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ cloud_ptr, int64_t Nmax,
                                int K, int32_t* __restrict__ cnt, int32_t* __restrict__ has_rep) {
  const int64_t b = blockIdx.y, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  if (i >= len) return;
  if (i >= Nmax) {            // (a cloud longer than the padded table -- bounded counts, capacity exceeded: no row to read)
    cnt[base + i] = 1;
    has_rep[base + i] = 1;
    return;
  }
  const int64_t* row = idx + (b * Nmax + i) * K;
  int found = 0;
  for (int s = 0; s < K; ++s) found += row[s] >= 0;
  cnt[base + i] = 1 + found;
  has_rep[base + i] = found < K;
}

__global__ void cg_fill_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ cloud_ptr, int64_t Nmax, int K,
                               const int32_t* __restrict__ grp_ptr, const int32_t* __restrict__ rep_off, int64_t E,
                               int32_t* __restrict__ row_src, int32_t* __restrict__ rep_row, float* __restrict__ row_w) {
  const int64_t b = blockIdx.y, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  if (i >= len) return;
  const int64_t p = base + i;
  const int64_t* row = idx + (b * Nmax + (i < Nmax ? i : 0)) * K;
  int32_t at = grp_ptr[p];
  // (at < E: E is the true number of real rows, or -- with bounded counts -- the capacity of row_src: never write past it)
  if (at < E) row_src[at] = (int32_t)p;
  ++at;
  int found = 0;
  for (int s = 0; s < K && i < Nmax; ++s) {
    const int64_t j = row[s];
    if (j >= 0) {
      if (at < E) row_src[at] = (int32_t)(base + j);
      ++at;
      ++found;
    }
  }
  if (found < K) {
    const int32_t r = rep_off[p];
    rep_row[p] = (int32_t)E + r;
    row_w[r] = (float)(K - found);
  } else {
    rep_row[p] = -1;
  }
}

// the sources of one point's real rows across the wave (lane s = row s of the group, groups have <= 64 rows)
__device__ __forceinline__ int cg_src(int v, int s) { return __builtin_amdgcn_readlane(v, s); }

// MODE 0: weighted column sums of y and y^2;  MODE 1: of g and g*xhat (g = dZ * act'(y*scale+shift)).
// "Point" N is the padding pseudo-point: no real rows, representative row E+Ne with y = 0.
template <int MODE, int DT = 0>
__global__ __launch_bounds__(TPB) void cg_edge_stats_kernel(
    const float* __restrict__ ps, int64_t ldps, const int32_t* __restrict__ grp_ptr, const int32_t* __restrict__ row_src,
    const int32_t* __restrict__ rep_row, const float* __restrict__ row_w, int64_t N, int64_t E, int64_t Ne, int Co,
    const void* __restrict__ dZ, int64_t lddz, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, int act, float slope, int pts,
    double* __restrict__ partial, float* __restrict__ pt = nullptr, int64_t ldpt = 0) {
  // pt (MODE 1, nullable; round 5): per point p the weighted sums over ITS rows, pt[p][c] = sum w g, pt[p][Co + c] = sum w xhat
  // -- what dS[p] is made of once the column sums are known (cg_edge_finish_kernel), so that no second pass over dZ is needed
  __shared__ double red[4][64][2];
  CCN_LANES;
  const int64_t first = ((int64_t)ccn_xcd_block() * 4 + ry) * pts;
  const int c = blockIdx.y * 64 + cx, cc = c < Co ? c : Co - 1;
  double s1 = 0.0, s2 = 0.0;
  float sc = 0.f, sh = 0.f, mu = 0.f, rs = 0.f;
  if (MODE == 1) {
    sc = scale[cc];
    sh = shift[cc];
    mu = mean[cc];
    rs = rstd[cc];
  }
  for (int t = 0; t < pts; ++t) {
    const int64_t p = first + t;
    if (p > N) break;
    const bool pad = p == N;
    const int32_t g0 = pad ? 0 : grp_ptr[p], cnt = pad ? 0 : grp_ptr[p + 1] - g0;
    const int32_t rrow = pad ? (int32_t)(E + Ne) : rep_row[p];
    const int mysrc = cx < cnt ? row_src[g0 + cx] : 0;
    const float si = pad ? 0.f : ps[p * ldps + Co + cc];
    float pg = 0.f, px = 0.f;         // this point's sums (MODE 1 with pt)
    for (int s0 = 0; s0 < cnt; s0 += SG_UNROLL) {
      float pv[SG_UNROLL], dz[SG_UNROLL];
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        const int sidx = s0 + u;
        const bool ok = sidx < cnt;
        const int j = cg_src(mysrc, sidx & 63);
        pv[u] = ok ? ps[(int64_t)j * ldps + cc] : 0.f;
        if (MODE == 1) {   // (row clamped, load unconditional, then select: a predicated 16-bit load became a branch + full wait each)
          const float v = ld_el<DT>(dZ, (int64_t)(g0 + (ok ? sidx : 0)) * lddz + cc);
          dz[u] = ok ? v : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        if (s0 + u >= cnt) continue;
        const float y = pv[u] + si;
        if (MODE == 0) {
          s1 += (double)y;
          s2 += (double)y * (double)y;
        } else {
          const float g = dz[u] * edge_act_grad(y * sc + sh, act, slope);
          const float xh = (y - mu) * rs;
          s1 += (double)g;
          s2 += (double)(g * xh);
          pg += g;
          px += xh;
        }
      }
    }
    if (rrow >= 0) {
      const double w = (double)row_w[rrow - E];
      const float y = si;
      if (MODE == 0) {
        s1 += w * (double)y;
        s2 += w * (double)y * (double)y;
      } else {
        const float g = ld_el<DT>(dZ, (int64_t)rrow * lddz + cc) * edge_act_grad(y * sc + sh, act, slope);
        const float xh = (y - mu) * rs;
        s1 += w * (double)g;
        s2 += w * (double)(g * xh);
        pg += (float)w * g;
        px += (float)w * xh;
      }
    }
    if (MODE == 1 && pt != nullptr && !pad && c < Co) {
      pt[p * ldpt + c] = pg;
      pt[p * ldpt + Co + c] = px;
    }
  }
  red[ry][cx][0] = s1;
  red[ry][cx][1] = s2;
  __syncthreads();
  if (ry == 0 && c < Co) {
    double a = 0.0, b2 = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[w][cx][0];
      b2 += red[w][cx][1];
    }
    partial[(int64_t)ccn_xcd_block() * 2 * Co + c] = a;
    partial[(int64_t)ccn_xcd_block() * 2 * Co + Co + c] = b2;
  }
}

template <int ZT>
__global__ __launch_bounds__(TPB) void cg_edge_apply_kernel(
    const float* __restrict__ ps, int64_t ldps, const int32_t* __restrict__ grp_ptr, const int32_t* __restrict__ row_src,
    const int32_t* __restrict__ rep_row, int64_t N, int64_t E, int64_t Ne, int Co, const float* __restrict__ scale,
    const float* __restrict__ shift, int act, float slope, void* __restrict__ Z, int64_t ldz) {
  CCN_LANES;
  const int64_t p = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (p > N) return;
  const bool pad = p == N;
  const int32_t g0 = pad ? 0 : grp_ptr[p], cnt = pad ? 0 : grp_ptr[p + 1] - g0;
  const int32_t rrow = pad ? (int32_t)(E + Ne) : rep_row[p];
  const int mysrc = cx < cnt ? row_src[g0 + cx] : 0;
  for (int c0 = 0; c0 < Co; c0 += 64) {
    const int c = c0 + cx, cc = c < Co ? c : Co - 1;
    const float si = pad ? 0.f : ps[p * ldps + Co + cc];
    const float sc = scale ? scale[cc] : 1.f, sh = shift ? shift[cc] : 0.f;
    for (int s0 = 0; s0 < cnt; s0 += SG_UNROLL) {
      float pv[SG_UNROLL];
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        const int j = cg_src(mysrc, (s0 + u) & 63);
        pv[u] = s0 + u < cnt ? ps[(int64_t)j * ldps + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u)
        if (s0 + u < cnt && c < Co) st_el<ZT>(Z, (int64_t)(g0 + s0 + u) * ldz + c, edge_act((pv[u] + si) * sc + sh, act, slope));
    }
    if (rrow >= 0 && c < Co) st_el<ZT>(Z, (int64_t)rrow * ldz + c, edge_act(si * sc + sh, act, slope));
  }
}

// dS[p] is owned by point p (plain store), dP[src] is accumulated atomically (the entry point zeroes that half).
template <int DT>
__global__ __launch_bounds__(TPB) void cg_edge_bwd_kernel(
    const float* __restrict__ ps, int64_t ldps, const int32_t* __restrict__ grp_ptr, const int32_t* __restrict__ row_src,
    const int32_t* __restrict__ rep_row, const float* __restrict__ row_w, int64_t N, int64_t E, int Co,
    const void* __restrict__ dZ, int64_t lddz, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, int act, float slope,
    const double* __restrict__ sums, double count, int training, float* __restrict__ dps, int64_t lddps) {
  CCN_LANES;
  const int64_t p = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (p >= N) return;  // the padding row feeds nothing upstream
  const int32_t g0 = grp_ptr[p], cnt = grp_ptr[p + 1] - g0;
  const int32_t rrow = rep_row[p];
  const int mysrc = cx < cnt ? row_src[g0 + cx] : 0;
  const float inv_n = (float)(1.0 / count);
  for (int c0 = 0; c0 < Co; c0 += 64) {
    const int c = c0 + cx, cc = c < Co ? c : Co - 1;
    const float si = ps[p * ldps + Co + cc];
    const float sc = scale ? scale[cc] : 1.f, sh = shift ? shift[cc] : 0.f;
    const float mu = mean ? mean[cc] : 0.f, rs = rstd ? rstd[cc] : 0.f;
    const float m1 = (training && sums) ? (float)sums[cc] * inv_n : 0.f;
    const float m2 = (training && sums) ? (float)sums[Co + cc] * inv_n : 0.f;
    float ds = 0.f;
    for (int s0 = 0; s0 < cnt; s0 += SG_UNROLL) {
      float pv[SG_UNROLL], dz[SG_UNROLL];
      int jj[SG_UNROLL];
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        const bool ok = s0 + u < cnt;
        jj[u] = cg_src(mysrc, (s0 + u) & 63);
        pv[u] = ok ? ps[(int64_t)jj[u] * ldps + cc] : 0.f;
        const float dzv = ld_el<DT>(dZ, (int64_t)(g0 + (ok ? s0 + u : 0)) * lddz + cc);
        dz[u] = ok ? dzv : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) {
        if (s0 + u >= cnt) continue;
        const float y = pv[u] + si;
        const float g = dz[u] * edge_act_grad(y * sc + sh, act, slope);
        const float dy = (training && sums) ? sc * (g - m1 - (y - mu) * rs * m2) : sc * g;
        ds += dy;
        if (c < Co) atomicAdd(&dps[(int64_t)jj[u] * lddps + c], dy);
      }
    }
    if (rrow >= 0) {
      const float y = si;
      const float g = ld_el<DT>(dZ, (int64_t)rrow * lddz + cc) * edge_act_grad(y * sc + sh, act, slope);
      const float dy = (training && sums) ? sc * (g - m1 - (y - mu) * rs * m2) : sc * g;
      ds += row_w[rrow - E] * dy;
    }
    if (c < Co) dps[p * lddps + Co + c] = ds;
  }
}

// ---- round 5: the same backward WITHOUT atomics and with ONE pass over dZ per index order.
// dy_r = sc (g_r - m1 - xhat_r m2) is linear in (g_r, 1, xhat_r), so the sums over a point's rows commute with the BatchNorm
// correction:   dS[p] = sc (sum_w g - W_p m1 - m2 sum_w xhat)  over the rows OF p (cg_edge_stats_kernel<1> writes the two sums
// next to the column sums it takes anyway),   dP[j] = sc (sum g - n_j m1 - m2 sum xhat)  over the rows whose SOURCE is j --
// gathered here through the inverse of row_src (built with the geometry: rows sorted by source, ascending row numbers, so the
// summation order is fixed).  The round-1..4 form read dZ a second time and added dy into dP[src] with fp32 atomics: 1.3 TB/s of
// added bytes at best (MI355X_MICROARCH.md), 39 % of HBM measured (profiles/r04_kitti_hbm_table.md), summation order free.
template <int DT>
__global__ __launch_bounds__(TPB) void cg_edge_gather_kernel(
    const float* __restrict__ ps, int64_t ldps, const int32_t* __restrict__ inv_ptr, const int32_t* __restrict__ inv_row,
    const int32_t* __restrict__ row_dst, int64_t N, int Co, const void* __restrict__ dZ, int64_t lddz,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int act, float slope, float* __restrict__ pp, int64_t ldpp) {
  CCN_LANES;
  const int64_t j = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (j >= N) return;
  const int32_t t0 = inv_ptr[j], t1 = inv_ptr[j + 1];
  for (int c0 = 0; c0 < Co; c0 += 64) {
    const int c = c0 + cx, cc = c < Co ? c : Co - 1;
    const float pj = ps[j * ldps + cc];
    const float sc = scale[cc], sh = shift[cc], mu = mean[cc], rs = rstd[cc];
    float pg = 0.f, px = 0.f;
    for (int32_t tb = t0; tb < t1; tb += 64) {      // 64 list entries at a time: lane s holds entry s (row and its destination)
      const int nb = t1 - tb < 64 ? t1 - tb : 64;
      const int myrow = cx < nb ? inv_row[tb + cx] : 0;
      const int mydst = cx < nb ? row_dst[myrow] : 0;
      for (int s0 = 0; s0 < nb; s0 += SG_UNROLL) {
        float sv[SG_UNROLL], dz[SG_UNROLL];
#pragma unroll
        for (int u = 0; u < SG_UNROLL; ++u) {
          const bool ok = s0 + u < nb;
          const int r = cg_src(myrow, (s0 + u) & 63), d = cg_src(mydst, (s0 + u) & 63);
          const float a = ps[(int64_t)d * ldps + Co + cc];
          const float v = ld_el<DT>(dZ, (int64_t)r * lddz + cc);
          sv[u] = ok ? a : 0.f;
          dz[u] = ok ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SG_UNROLL; ++u) {
          if (s0 + u >= nb) continue;
          const float y = pj + sv[u];
          pg += dz[u] * edge_act_grad(y * sc + sh, act, slope);
          px += (y - mu) * rs;
        }
      }
    }
    if (c < Co) {
      pp[j * ldpp + c] = pg;
      pp[j * ldpp + Co + c] = px;
    }
  }
}

// dps[j] = [dP[j] | dS[j]] from the per-point sums (pp: over the rows whose source is j, pt: over the rows of j) and the column sums
__global__ __launch_bounds__(TPB) void cg_edge_finish_kernel(
    const float* __restrict__ pt, int64_t ldpt, const float* __restrict__ pp, int64_t ldpp, const int32_t* __restrict__ grp_ptr,
    const int32_t* __restrict__ rep_row, const float* __restrict__ row_w, const int32_t* __restrict__ inv_ptr, int64_t N,
    int64_t E, int Co, const float* __restrict__ scale, const double* __restrict__ sums, double count, int training,
    float* __restrict__ dps, int64_t lddps) {
  CCN_LANES;
  const int64_t j = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (j >= N) return;
  const float inv_n = (float)(1.0 / count);
  const int32_t rrow = rep_row[j];
  const float wsum = (float)(grp_ptr[j + 1] - grp_ptr[j]) + (rrow >= 0 ? row_w[rrow - E] : 0.f);
  const float nsrc = (float)(inv_ptr[j + 1] - inv_ptr[j]);
  for (int c = cx; c < Co; c += 64) {
    const float sc = scale[c];
    const float m1 = training ? (float)sums[c] * inv_n : 0.f, m2 = training ? (float)sums[Co + c] * inv_n : 0.f;
    dps[j * lddps + c] = sc * (pp[j * ldpp + c] - nsrc * m1 - m2 * pp[j * ldpp + Co + c]);
    dps[j * lddps + Co + c] = sc * (pt[j * ldpt + c] - wsum * m1 - m2 * pt[j * ldpt + Co + c]);
  }
}

// masked max over a point's real rows; the empty slots of the dense layout enter as the constant -1e2 (dgcnn.py:187-189)
__global__ __launch_bounds__(TPB) void cg_max_fwd_kernel(const float* __restrict__ f, int64_t ldf,
                                                         const int32_t* __restrict__ grp_ptr,
                                                         const int32_t* __restrict__ rep_row, int64_t N, int C,
                                                         float* __restrict__ out, int64_t ldo, int32_t* __restrict__ arg) {
  CCN_LANES;
  const int64_t p = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (p >= N) return;
  const int32_t g0 = grp_ptr[p], cnt = grp_ptr[p + 1] - g0;
  const bool has_empty = rep_row[p] >= 0;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cx, cc = c < C ? c : C - 1;
    float best = f[(int64_t)g0 * ldf + cc];  // the self row always exists
    int at = 0;
    for (int s0 = 1; s0 < cnt; s0 += SG_UNROLL) {
      float v[SG_UNROLL];
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u) v[u] = s0 + u < cnt ? f[(int64_t)(g0 + s0 + u) * ldf + cc] : -__builtin_inff();
#pragma unroll
      for (int u = 0; u < SG_UNROLL; ++u)
        if (s0 + u < cnt && v[u] > best) {
          best = v[u];
          at = s0 + u;
        }
    }
    if (has_empty && -1e2f > best) {
      best = -1e2f;
      at = -1;
    }
    if (c < C) {
      out[p * ldo + c] = best;
      arg[p * C + c] = at;
    }
  }
}

// df rows of the point: the argmax row gets the gradient, the others (and the representative) zero; wave N clears
// the padding row
template <int T>     // T = 1: df as bf16 rows (the 16-bit storage modes: dY of the plain Linear in front of the max)
__global__ __launch_bounds__(TPB) void cg_max_bwd_kernel(const float* __restrict__ dout, int64_t lddo,
                                                         const int32_t* __restrict__ arg,
                                                         const int32_t* __restrict__ grp_ptr,
                                                         const int32_t* __restrict__ rep_row, int64_t N, int64_t R, int C,
                                                         void* __restrict__ df, int64_t lddf) {
  CCN_LANES;
  const int64_t p = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (p > N) return;
  if (p == N) {
    for (int c = cx; c < C; c += 64) st_el<T>(df, (R - 1) * lddf + c, 0.f);
    return;
  }
  const int32_t g0 = grp_ptr[p], cnt = grp_ptr[p + 1] - g0;
  const int32_t rrow = rep_row[p];
  for (int c = cx; c < C; c += 64) {
    const int at = arg[p * C + c];
    const float g = dout[p * lddo + c];
    for (int s = 0; s < cnt; ++s) st_el<T>(df, (int64_t)(g0 + s) * lddf + c, s == at ? g : 0.f);
    if (rrow >= 0) st_el<T>(df, (int64_t)rrow * lddf + c, 0.f);
  }
}

// ------------------------------------------------------------------ A13, algebraic form of the first message layer
// W [x_j ; (p_j - p_i)/r] + b = PX[j] + Wp (p_j - p_i)/r + b with the per-point product PX = X Wx^T (N_src rows
// instead of E edge rows); the 3-column position part is evaluated per edge from the relative position itself
// (no cancellation between large coordinates).  BatchNorm statistics run over the E real edges.
constexpr int PN_EDGES_MAX = 32;  // edges per wave in the statistics kernels

struct PnEdge {
  int64_t j;
  float r0, r1, r2;
};
// Geometry of 64 consecutive edges, one per lane (source index and relative position), handed to the whole wave
// edge by edge with readlane: the index -> position -> feature-row chain is paid once per 64 edges, and the
// feature-row gathers of 4 edges are in flight together.
__device__ __forceinline__ PnEdge pn_edge_lane(const float* __restrict__ pos_src, const float* __restrict__ pos_dst,
                                               const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                               int64_t e, int64_t e_end, float radius) {
  PnEdge o = {-1, 0.f, 0.f, 0.f};
  if (e < e_end) {
    o.j = src[e];
    const int64_t q = dst[e];
    o.r0 = pos_src[3 * o.j] - pos_dst[3 * q];
    o.r1 = pos_src[3 * o.j + 1] - pos_dst[3 * q + 1];
    o.r2 = pos_src[3 * o.j + 2] - pos_dst[3 * q + 2];
    if (radius > 0.f) {
      o.r0 = __fdiv_rn(o.r0, radius);
      o.r1 = __fdiv_rn(o.r1, radius);
      o.r2 = __fdiv_rn(o.r2, radius);
    }
  }
  return o;
}
__device__ __forceinline__ PnEdge pn_edge_bcast(const PnEdge& g, int t) {
  PnEdge o;
  const int lo = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)g.j & 0xffffffffu), t);
  const int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)g.j >> 32), t);
  o.j = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
  o.r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.r0), t));
  o.r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.r1), t));
  o.r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.r2), t));
  return o;
}

// MODE 0: column sums of y and y^2;  MODE 1: column sums of g and g*xhat, g = dZ * act'(y*scale+shift)
template <int MODE, int DT = 0>
__global__ __launch_bounds__(TPB) void pn_edge_stats_kernel(
    const float* __restrict__ px, int64_t ldpx, const float* __restrict__ wp, int64_t ldwp,
    const float* __restrict__ bias, const float* __restrict__ pos_src, const float* __restrict__ pos_dst,
    const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int Co, float radius,
    const void* __restrict__ dZ, int64_t lddz, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, int act, float slope, int per_wave,
    double* __restrict__ partial) {
  __shared__ double red[4][64][2];
  CCN_LANES;
  const int64_t first = ((int64_t)ccn_xcd_block() * 4 + ry) * per_wave;  // per_wave <= 64
  const int64_t last = first + per_wave < E ? first + per_wave : E;
  const int c = blockIdx.y * 64 + cx, cc = c < Co ? c : Co - 1;
  double s1 = 0.0, s2 = 0.0;
  const float w0 = wp[cc * ldwp], w1 = wp[cc * ldwp + 1], w2 = wp[cc * ldwp + 2];
  const float bv = bias ? bias[cc] : 0.f;
  float sc = 0.f, sh = 0.f, mu = 0.f, rs = 0.f;
  if (MODE == 1) {
    sc = scale[cc];
    sh = shift[cc];
    mu = mean[cc];
    rs = rstd[cc];
  }
  const PnEdge mine = pn_edge_lane(pos_src, pos_dst, src, dst, first + cx, last, radius);
  const int cnt = first < last ? (int)(last - first) : 0;
  for (int t0 = 0; t0 < cnt; t0 += 4) {
    PnEdge ed[4];
    float pv[4], dz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ed[u] = pn_edge_bcast(mine, (t0 + u) & 63);
      const bool ok = t0 + u < cnt;
      pv[u] = ok ? px[ed[u].j * ldpx + cc] : 0.f;
      if (MODE == 1) {
        const float v = ld_el<DT>(dZ, (first + (ok ? t0 + u : 0)) * lddz + cc);
        dz[u] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (t0 + u >= cnt) continue;
      const float y = pv[u] + (w0 * ed[u].r0 + w1 * ed[u].r1 + w2 * ed[u].r2) + bv;
      if (MODE == 0) {
        s1 += (double)y;
        s2 += (double)y * (double)y;
      } else {
        const float g = dz[u] * edge_act_grad(y * sc + sh, act, slope);
        s1 += (double)g;
        s2 += (double)(g * ((y - mu) * rs));
      }
    }
  }
  red[ry][cx][0] = s1;
  red[ry][cx][1] = s2;
  __syncthreads();
  if (ry == 0 && c < Co) {
    double a = 0.0, b2 = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[w][cx][0];
      b2 += red[w][cx][1];
    }
    partial[(int64_t)ccn_xcd_block() * 2 * Co + c] = a;
    partial[(int64_t)ccn_xcd_block() * 2 * Co + Co + c] = b2;
  }
}

constexpr int PN_APPLY_EDGES = 16;  // edges per wave in the apply kernel
template <int ZT>
__global__ __launch_bounds__(TPB) void pn_edge_apply_kernel(
    const float* __restrict__ px, int64_t ldpx, const float* __restrict__ wp, int64_t ldwp,
    const float* __restrict__ bias, const float* __restrict__ pos_src, const float* __restrict__ pos_dst,
    const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int Co, float radius,
    const float* __restrict__ scale, const float* __restrict__ shift, int act, float slope, void* __restrict__ Z,
    int64_t ldz) {
  CCN_LANES;
  const int64_t first = ((int64_t)ccn_xcd_block() * 4 + ry) * PN_APPLY_EDGES;
  if (first >= E) return;
  const int64_t last = first + PN_APPLY_EDGES < E ? first + PN_APPLY_EDGES : E;
  const int cnt = (int)(last - first);
  const PnEdge mine = pn_edge_lane(pos_src, pos_dst, src, dst, first + cx, last, radius);
  for (int c0 = 0; c0 < Co; c0 += 64) {
    const int c = c0 + cx, cc = c < Co ? c : Co - 1;
    const float w0 = wp[cc * ldwp], w1 = wp[cc * ldwp + 1], w2 = wp[cc * ldwp + 2];
    const float bv = bias ? bias[cc] : 0.f;
    const float sc = scale ? scale[cc] : 1.f, sh = shift ? shift[cc] : 0.f;
    for (int t0 = 0; t0 < cnt; t0 += 4) {
      PnEdge ed[4];
      float pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ed[u] = pn_edge_bcast(mine, (t0 + u) & 63);
        pv[u] = t0 + u < cnt ? px[ed[u].j * ldpx + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (t0 + u >= cnt || c >= Co) continue;
        const float y = pv[u] + (w0 * ed[u].r0 + w1 * ed[u].r1 + w2 * ed[u].r2) + bv;
        st_el<ZT>(Z, (first + t0 + u) * ldz + c, edge_act(y * sc + sh, act, slope));
      }
    }
  }
}

// dPX must be zero on entry (atomic accumulation per source point).  Each wave walks `per_wave` consecutive edges
// and keeps the sums dWp[c][0..2] = sum dy*rel, dbias[c] = sum dy in registers; wpart: [gridDim.x*4][4][Co] doubles,
// every row written (no initialisation needed).
template <int DT>
__global__ __launch_bounds__(TPB) void pn_edge_bwd_kernel(
    const float* __restrict__ px, int64_t ldpx, const float* __restrict__ wp, int64_t ldwp,
    const float* __restrict__ bias, const float* __restrict__ pos_src, const float* __restrict__ pos_dst,
    const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int Co, float radius,
    const void* __restrict__ dZ, int64_t lddz, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, int act, float slope,
    const double* __restrict__ sums, int training, int per_wave, float* __restrict__ dpx, int64_t lddpx,
    double* __restrict__ wpart) {
  CCN_LANES;
  const int64_t wave_id = (int64_t)ccn_xcd_block() * 4 + ry;
  const int64_t first = wave_id * per_wave;
  if (first >= E) {  // a wave without edges still owns a row of partial sums
    for (int c = cx; c < 4 * Co; c += 64) wpart[wave_id * 4 * Co + c] = 0.0;
    return;
  }
  const int64_t last = first + per_wave < E ? first + per_wave : E;
  const float inv_n = 1.0f / (float)E;
  for (int c0 = 0; c0 < Co; c0 += 64) {
    const int c = c0 + cx, cc = c < Co ? c : Co - 1;
    const float w0 = wp[cc * ldwp], w1 = wp[cc * ldwp + 1], w2 = wp[cc * ldwp + 2];
    const float bv = bias ? bias[cc] : 0.f;
    const float sc = scale ? scale[cc] : 1.f, sh = shift ? shift[cc] : 0.f;
    const float mu = mean ? mean[cc] : 0.f, rs = rstd ? rstd[cc] : 0.f;
    const float m1 = (training && sums) ? (float)sums[cc] * inv_n : 0.f;
    const float m2 = (training && sums) ? (float)sums[Co + cc] * inv_n : 0.f;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, ab = 0.0;
    for (int64_t e0 = first; e0 < last; e0 += 64) {
      const int64_t e1 = e0 + 64 < last ? e0 + 64 : last;
      const int cnt = (int)(e1 - e0);
      const PnEdge mine = pn_edge_lane(pos_src, pos_dst, src, dst, e0 + cx, e1, radius);
      for (int t0 = 0; t0 < cnt; t0 += 4) {
        PnEdge ed[4];
        float pv[4], dz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ed[u] = pn_edge_bcast(mine, (t0 + u) & 63);
          const bool ok = t0 + u < cnt;
          pv[u] = ok ? px[ed[u].j * ldpx + cc] : 0.f;
          const float dzv = ld_el<DT>(dZ, (e0 + (ok ? t0 + u : 0)) * lddz + cc);
          dz[u] = ok ? dzv : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (t0 + u >= cnt) continue;
          const float y = pv[u] + (w0 * ed[u].r0 + w1 * ed[u].r1 + w2 * ed[u].r2) + bv;
          const float g = dz[u] * edge_act_grad(y * sc + sh, act, slope);
          const float dy = (training && sums) ? sc * (g - m1 - (y - mu) * rs * m2) : sc * g;
          if (c < Co) atomicAdd(&dpx[ed[u].j * lddpx + c], dy);
          a0 += (double)(dy * ed[u].r0);
          a1 += (double)(dy * ed[u].r1);
          a2 += (double)(dy * ed[u].r2);
          ab += (double)dy;
        }
      }
    }
    if (c < Co) {
      double* o = wpart + wave_id * 4 * Co;
      o[c] = a0;
      o[Co + c] = a1;
      o[2 * Co + c] = a2;
      o[3 * Co + c] = ab;
    }
  }
}

// ---- round 5: the same backward WITHOUT atomics (as cg_edge_gather_kernel for the SGCNN layer).  dy_e = sc (g_e - m1 - xhat_e m2)
// is linear in (g_e, 1, xhat_e), so everything the backward needs is a SUM that can be taken before m1, m2 are known:
//   column sums (one pass over dZ in edge order): g, g xhat (BatchNorm backward), g rel_k, xhat, xhat rel_k, and rel_k itself
//   -> dWp[c][k] = sc (sum g rel_k - m1 sum rel_k - m2 sum xhat rel_k),  dbias[c] = sc (sum g - E m1 - m2 sum xhat);
//   per SOURCE point j the sums of g and xhat over the edges that read it, gathered through the inverse of `src`
//   -> dPX[j] = sc (sum g - n_j m1 - m2 sum xhat).
// partial: [gridDim.x][9 Co + 4] doubles: rows of [g | g xhat | g r0 | g r1 | g r2 | xhat | xhat r0 | xhat r1 | xhat r2 | r0 r1 r2 -].
template <int DT>
__global__ __launch_bounds__(TPB) void pn_edge_sums_kernel(
    const float* __restrict__ px, int64_t ldpx, const float* __restrict__ wp, int64_t ldwp,
    const float* __restrict__ bias, const float* __restrict__ pos_src, const float* __restrict__ pos_dst,
    const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int Co, float radius,
    const void* __restrict__ dZ, int64_t lddz, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, int act, float slope, int per_wave,
    double* __restrict__ partial) {
  __shared__ double red[4][64][9];
  __shared__ double redr[4][3];
  CCN_LANES;
  const int64_t first = ((int64_t)ccn_xcd_block() * 4 + ry) * per_wave;  // per_wave <= 64
  const int64_t last = first + per_wave < E ? first + per_wave : E;
  const int c = blockIdx.y * 64 + cx, cc = c < Co ? c : Co - 1;
  double a[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, rr[3] = {0.0, 0.0, 0.0};
  const float w0 = wp[cc * ldwp], w1 = wp[cc * ldwp + 1], w2 = wp[cc * ldwp + 2];
  const float bv = bias ? bias[cc] : 0.f;
  const float sc = scale[cc], sh = shift[cc], mu = mean[cc], rs = rstd[cc];
  const PnEdge mine = pn_edge_lane(pos_src, pos_dst, src, dst, first + cx, last, radius);
  const int cnt = first < last ? (int)(last - first) : 0;
  for (int t0 = 0; t0 < cnt; t0 += 4) {
    PnEdge ed[4];
    float pv[4], dz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ed[u] = pn_edge_bcast(mine, (t0 + u) & 63);
      const bool ok = t0 + u < cnt;
      pv[u] = ok ? px[ed[u].j * ldpx + cc] : 0.f;
      const float v = ld_el<DT>(dZ, (first + (ok ? t0 + u : 0)) * lddz + cc);
      dz[u] = ok ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (t0 + u >= cnt) continue;
      const float y = pv[u] + (w0 * ed[u].r0 + w1 * ed[u].r1 + w2 * ed[u].r2) + bv;
      const float g = dz[u] * edge_act_grad(y * sc + sh, act, slope);
      const float xh = (y - mu) * rs;
      a[0] += (double)g;
      a[1] += (double)(g * xh);
      a[2] += (double)(g * ed[u].r0);
      a[3] += (double)(g * ed[u].r1);
      a[4] += (double)(g * ed[u].r2);
      a[5] += (double)xh;
      a[6] += (double)(xh * ed[u].r0);
      a[7] += (double)(xh * ed[u].r1);
      a[8] += (double)(xh * ed[u].r2);
      rr[0] += (double)ed[u].r0;
      rr[1] += (double)ed[u].r1;
      rr[2] += (double)ed[u].r2;
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) red[ry][cx][k] = a[k];
  if (cx == 0) {
    redr[ry][0] = rr[0];
    redr[ry][1] = rr[1];
    redr[ry][2] = rr[2];
  }
  __syncthreads();
  double* const o = partial + (int64_t)ccn_xcd_block() * (9 * Co + 4);
  if (ry == 0 && c < Co) {
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k * Co + c] = red[0][cx][k] + red[1][cx][k] + red[2][cx][k] + red[3][cx][k];
  }
  if (blockIdx.y == 0 && threadIdx.x < 4)
    o[9 * Co + threadIdx.x] = threadIdx.x < 3 ? redr[0][threadIdx.x] + redr[1][threadIdx.x] + redr[2][threadIdx.x] + redr[3][threadIdx.x] : 0.0;
}

template <int DT>
__global__ __launch_bounds__(TPB) void pn_edge_gather_kernel(
    const float* __restrict__ px, int64_t ldpx, const float* __restrict__ wp, int64_t ldwp, const float* __restrict__ bias,
    const float* __restrict__ pos_src, const float* __restrict__ pos_dst, const int64_t* __restrict__ dst,
    const int32_t* __restrict__ inv_ptr, const int32_t* __restrict__ inv_edge, int64_t Nsrc, int Co, float radius,
    const void* __restrict__ dZ, int64_t lddz, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, int act, float slope, float* __restrict__ pp,
    int64_t ldpp) {
  CCN_LANES;
  const int64_t j = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (j >= Nsrc) return;
  const int32_t t0 = inv_ptr[j], t1 = inv_ptr[j + 1];
  const float sx = pos_src[3 * j], sy = pos_src[3 * j + 1], sz = pos_src[3 * j + 2];
  for (int c0 = 0; c0 < Co; c0 += 64) {
    const int c = c0 + cx, cc = c < Co ? c : Co - 1;
    const float pj = px[j * ldpx + cc] + (bias ? bias[cc] : 0.f);
    const float w0 = wp[cc * ldwp], w1 = wp[cc * ldwp + 1], w2 = wp[cc * ldwp + 2];
    const float sc = scale[cc], sh = shift[cc], mu = mean[cc], rs = rstd[cc];
    float pg = 0.f, pxh = 0.f;
    for (int32_t tb = t0; tb < t1; tb += 64) {        // 64 list entries at a time: lane s holds edge s and its relative position
      const int nb = t1 - tb < 64 ? t1 - tb : 64;
      int myedge = 0;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
      if (cx < nb) {
        myedge = inv_edge[tb + cx];
        const int64_t q = dst[myedge];
        r0 = sx - pos_dst[3 * q];
        r1 = sy - pos_dst[3 * q + 1];
        r2 = sz - pos_dst[3 * q + 2];
        if (radius > 0.f) {
          r0 = __fdiv_rn(r0, radius);
          r1 = __fdiv_rn(r1, radius);
          r2 = __fdiv_rn(r2, radius);
        }
      }
      for (int s0 = 0; s0 < nb; s0 += SG_UNROLL) {
        float dz[SG_UNROLL], rel[SG_UNROLL];
#pragma unroll
        for (int u = 0; u < SG_UNROLL; ++u) {
          const int l = (s0 + u) & 63;
          const bool ok = s0 + u < nb;
          const int e = cg_src(myedge, l);
          const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0), l));
          const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1), l));
          const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r2), l));
          const float v = ld_el<DT>(dZ, (int64_t)e * lddz + cc);
          dz[u] = ok ? v : 0.f;
          rel[u] = w0 * a0 + w1 * a1 + w2 * a2;
        }
#pragma unroll
        for (int u = 0; u < SG_UNROLL; ++u) {
          if (s0 + u >= nb) continue;
          const float y = (px[j * ldpx + cc] + rel[u]) + (bias ? bias[cc] : 0.f);     // (the forward's order of the three terms)
          pg += dz[u] * edge_act_grad(y * sc + sh, act, slope);
          pxh += (y - mu) * rs;
        }
      }
    }
    (void)pj;
    if (c < Co) {
      pp[j * ldpp + c] = pg;
      pp[j * ldpp + Co + c] = pxh;
    }
  }
}

// dpx[j] from the per-source sums; block 0 also turns the column sums into dw4 = [dWp[:, 0] | dWp[:, 1] | dWp[:, 2] | dbias] (4 x Co)
__global__ __launch_bounds__(TPB) void pn_edge_finish_kernel(const float* __restrict__ pp, int64_t ldpp,
                                                             const int32_t* __restrict__ inv_ptr, int64_t Nsrc, int64_t E,
                                                             int Co, const float* __restrict__ scale,
                                                             const double* __restrict__ sums, int training,
                                                             float* __restrict__ dpx, int64_t lddpx, float* __restrict__ dw4) {
  CCN_LANES;
  const double inv_e = 1.0 / (double)E;
  if (ccn_xcd_block() == 0) {
    for (int c = threadIdx.x; c < Co; c += TPB) {
      const double sc = (double)scale[c];
      const double m1 = training ? sums[c] * inv_e : 0.0, m2 = training ? sums[Co + c] * inv_e : 0.0;
      for (int k = 0; k < 3; ++k)
        dw4[k * Co + c] = (float)(sc * (sums[(2 + k) * Co + c] - m1 * sums[9 * Co + k] - m2 * sums[(6 + k) * Co + c]));
      dw4[3 * Co + c] = (float)(sc * (sums[c] - (double)E * m1 - m2 * sums[5 * Co + c]));
    }
  }
  const int64_t j = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (j >= Nsrc) return;
  const float nj = (float)(inv_ptr[j + 1] - inv_ptr[j]);
  for (int c = cx; c < Co; c += 64) {
    const float sc = scale[c];
    const float m1 = training ? (float)(sums[c] * inv_e) : 0.f, m2 = training ? (float)(sums[Co + c] * inv_e) : 0.f;
    dpx[j * lddpx + c] = sc * (pp[j * ldpp + c] - nj * m1 - m2 * pp[j * ldpp + Co + c]);
  }
}

// ------------------------------------------------------------------ A13: PointNetConv2 message
__global__ __launch_bounds__(TPB) void msg_build_fwd_kernel(const float* __restrict__ x_src, int64_t ldx,
                                                            const float* __restrict__ pos_src,
                                                            const float* __restrict__ pos_dst,
                                                            const int64_t* __restrict__ src,
                                                            const int64_t* __restrict__ dst, int64_t E, int C,
                                                            float radius, float* __restrict__ msg, int64_t ldm) {
  CCN_LANES;
  const int64_t e = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (e >= E) return;
  const int64_t j = src[e], q = dst[e];
  const int W = C + 3;
  for (int c0 = 0; c0 < W; c0 += 64) {
    const int c = c0 + cx;
    if (c >= W) continue;
    float v;
    if (c < C) {
      v = x_src[j * ldx + c];
    } else {
      const int d = c - C;
      v = pos_src[3 * j + d] - pos_dst[3 * q + d];
      if (radius > 0.f) v = __fdiv_rn(v, radius);
    }
    msg[e * ldm + c] = v;
  }
}

__global__ __launch_bounds__(TPB) void msg_build_bwd_kernel(const float* __restrict__ dmsg, int64_t lddm,
                                                            const int64_t* __restrict__ src, int64_t E, int C,
                                                            float* __restrict__ dx, int64_t lddx) {
  CCN_LANES;
  const int64_t e = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (e >= E) return;
  const int64_t j = src[e];
  for (int c = cx; c < C; c += 64) atomicAdd(&dx[j * lddx + c], dmsg[e * lddm + c]);
}

// ------------------------------------------------------------------ sparse edge conv message (dgcnn.py:227-228)
// msg[e] = [x_i, x_j - x_i], i = dst[e] (the query), j = src[e] (its neighbour)
template <int ZT>     // ZT: the message rows as fp32 (0), bf16 (1) or fp16 (2) (ccn_common.h: st_el)
__global__ __launch_bounds__(TPB) void edge_feat_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const int64_t* __restrict__ src,
                                                            const int64_t* __restrict__ dst, int64_t E, int C,
                                                            void* __restrict__ msg, int64_t ldm) {
  CCN_LANES;
  const int64_t e = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (e >= E) return;
  const int64_t j = src[e], i = dst[e];
  for (int c = cx; c < C; c += 64) {
    const float xi = x[i * ldx + c];
    st_el<ZT>(msg, e * ldm + c, xi);
    st_el<ZT>(msg, e * ldm + C + c, x[j * ldx + c] - xi);
  }
}

// Backward over edges GROUPED by destination (CSR offsets: the layout the sparse path builds anyway): a wave owns one
// destination i, sums (a - b) over its edges in registers and adds it to dx[i] once; only the neighbour side keeps one atomic
// per edge and channel (round 3: half of the per-edge form's atomics, 2.5 ms per step of BASELINE configs[4]).  DT = 1: dmsg as
// bf16 rows (the gradient of 16-bit message rows).
template <int DT>
__global__ __launch_bounds__(TPB) void edge_feat_bwd_csr_kernel(const void* __restrict__ dmsg, int64_t lddm,
                                                                const int64_t* __restrict__ src,
                                                                const int32_t* __restrict__ offsets, int64_t num_dst, int C,
                                                                float* __restrict__ dx, int64_t lddx) {
  CCN_LANES;
  const int64_t i = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (i >= num_dst) return;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  if (lo >= hi) return;
  for (int c = cx; c < C; c += 64) {
    float own = 0.f;
    int32_t e = lo;
    for (; e + 4 <= hi; e += 4) {
      float a[4], b[4];
      int64_t j[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        j[u] = src[e + u];
        a[u] = ld_el<DT>(dmsg, (int64_t)(e + u) * lddm + c);
        b[u] = ld_el<DT>(dmsg, (int64_t)(e + u) * lddm + C + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        own += a[u] - b[u];
        atomicAdd(&dx[j[u] * lddx + c], b[u]);
      }
    }
    for (; e < hi; ++e) {
      const float a = ld_el<DT>(dmsg, (int64_t)e * lddm + c), b = ld_el<DT>(dmsg, (int64_t)e * lddm + C + c);
      own += a - b;
      atomicAdd(&dx[src[e] * lddx + c], b);
    }
    atomicAdd(&dx[i * lddx + c], own);
  }
}

__global__ __launch_bounds__(TPB) void edge_feat_bwd_kernel(const float* __restrict__ dmsg, int64_t lddm,
                                                            const int64_t* __restrict__ src,
                                                            const int64_t* __restrict__ dst, int64_t E, int C,
                                                            float* __restrict__ dx, int64_t lddx) {
  CCN_LANES;
  const int64_t e = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (e >= E) return;
  const int64_t j = src[e], i = dst[e];
  for (int c = cx; c < C; c += 64) {
    const float a = dmsg[e * lddm + c], b = dmsg[e * lddm + C + c];
    atomicAdd(&dx[i * lddx + c], a - b);
    atomicAdd(&dx[j * lddx + c], b);
  }
}

// ------------------------------------------------------------------ grouped (CSR) aggregation
// One streaming pass with a running maximum (the sums are rescaled when the maximum moves), 4 edge rows in flight
// per lane: the group's att / msg rows are read once in forward and twice in backward.
struct SoftAcc {
  float top, tot, acc;
};
__device__ __forceinline__ void soft_push(SoftAcc& s, float a, float v) {
  if (a > s.top) {
    const float k = __expf(s.top - a);  // exp(-inf) = 0 on the first element
    s.tot *= k;
    s.acc *= k;
    s.top = a;
  }
  const float p = __expf(a - s.top);
  s.tot += p;
  s.acc += p * v;
}

__global__ __launch_bounds__(TPB) void seg_softmax_agg_fwd_kernel(const float* __restrict__ msg, int64_t ldm,
                                                                  const float* __restrict__ att, int64_t lda,
                                                                  const int32_t* __restrict__ offsets, int64_t M,
                                                                  int C, float* __restrict__ out, int64_t ldo) {
  CCN_LANES;
  const int64_t i = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (i >= M) return;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  for (int c = cx; c < C; c += 64) {
    SoftAcc s = {-__builtin_inff(), 0.f, 0.f};
    int32_t e = lo;
    for (; e + 4 <= hi; e += 4) {
      float a[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = att[(int64_t)(e + u) * lda + c];
        v[u] = msg[(int64_t)(e + u) * ldm + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) soft_push(s, a[u], v[u]);
    }
    for (; e < hi; ++e) soft_push(s, att[(int64_t)e * lda + c], msg[(int64_t)e * ldm + c]);
    out[i * ldo + c] = s.acc * (1.0f / (s.tot + 1e-16f));
  }
}

template <int T>     // T = 1: datt as bf16 rows (16-bit storage modes: dY of the plain Linear that produced att)
__global__ __launch_bounds__(TPB) void seg_softmax_agg_bwd_kernel(const float* __restrict__ msg, int64_t ldm,
                                                                  const float* __restrict__ att, int64_t lda,
                                                                  const int32_t* __restrict__ offsets, int64_t M,
                                                                  int C, const float* __restrict__ dout,
                                                                  int64_t lddo, float* __restrict__ dmsg,
                                                                  int64_t lddm, void* __restrict__ datt,
                                                                  int64_t ldda) {
  CCN_LANES;
  const int64_t i = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (i >= M) return;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  for (int c = cx; c < C; c += 64) {
    const float g = dout[i * lddo + c];
    SoftAcc s = {-__builtin_inff(), 0.f, 0.f};
    int32_t e = lo;
    for (; e + 4 <= hi; e += 4) {
      float a[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = att[(int64_t)(e + u) * lda + c];
        v[u] = msg[(int64_t)(e + u) * ldm + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) soft_push(s, a[u], v[u]);
    }
    for (; e < hi; ++e) soft_push(s, att[(int64_t)e * lda + c], msg[(int64_t)e * ldm + c]);
    const float inv = 1.0f / (s.tot + 1e-16f);
    const float dot = s.acc * inv * g;  // sum_e w_e * msg_e * g
    e = lo;
    for (; e + 4 <= hi; e += 4) {
      float a[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = att[(int64_t)(e + u) * lda + c];
        v[u] = msg[(int64_t)(e + u) * ldm + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float w = __expf(a[u] - s.top) * inv;
        dmsg[(int64_t)(e + u) * lddm + c] = w * g;
        st_el<T>(datt, (int64_t)(e + u) * ldda + c, w * (v[u] * g - dot));
      }
    }
    for (; e < hi; ++e) {
      const float w = __expf(att[(int64_t)e * lda + c] - s.top) * inv;
      dmsg[(int64_t)e * lddm + c] = w * g;
      st_el<T>(datt, (int64_t)e * ldda + c, w * (msg[(int64_t)e * ldm + c] * g - dot));
    }
  }
}

__global__ __launch_bounds__(TPB) void seg_max_fwd_kernel(const float* __restrict__ msg, int64_t ldm,
                                                          const int32_t* __restrict__ offsets, int64_t M, int C,
                                                          float* __restrict__ out, int64_t ldo,
                                                          int32_t* __restrict__ arg) {
  CCN_LANES;
  const int64_t i = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (i >= M) return;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  for (int c = cx; c < C; c += 64) {
    float best = 0.f;  // scatter_max: empty groups give 0
    int at = -1;
    for (int32_t e = lo; e < hi; ++e) {
      const float v = msg[(int64_t)e * ldm + c];
      if (at < 0 || v > best) {
        best = v;
        at = e - lo;
      }
    }
    out[i * ldo + c] = best;
    arg[i * C + c] = at;
  }
}

__global__ __launch_bounds__(TPB) void seg_max_bwd_kernel(const float* __restrict__ dout, int64_t lddo,
                                                          const int32_t* __restrict__ arg,
                                                          const int32_t* __restrict__ offsets, int64_t M, int C,
                                                          float* __restrict__ dmsg, int64_t lddm) {
  CCN_LANES;
  const int64_t i = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (i >= M) return;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  for (int c = cx; c < C; c += 64) {
    const int at = arg[i * C + c];
    const float g = dout[i * lddo + c];
    for (int32_t e = lo; e < hi; ++e) dmsg[(int64_t)e * lddm + c] = (e - lo == at) ? g : 0.f;
  }
}

// ------------------------------------------------------------------ the remaining aggregation branches
// (no shipped config selects them; built so that every branch of the reference's two aggregate() functions computes)
//
// CSR groups, point_conv.py:82-88:
//   mode 0 'mean'         out = sum_e msg / max(count, 1)                  (torch_scatter scatter_mean)
//   mode 1 'weighted-sum' out = sum_e msg * sigmoid(att)                   (scatter_add of inputs * F.sigmoid(attend_nn))
__device__ __forceinline__ float sigmoidf_(float a) { return 1.0f / (1.0f + __expf(-a)); }

__global__ __launch_bounds__(TPB) void seg_wsum_fwd_kernel(const float* __restrict__ msg, int64_t ldm,
                                                           const float* __restrict__ att, int64_t lda,
                                                           const int32_t* __restrict__ offsets, int64_t M, int C, int mode,
                                                           float* __restrict__ out, int64_t ldo) {
  CCN_LANES;
  const int64_t i = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (i >= M) return;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  const float inv = 1.0f / (float)(hi - lo > 1 ? hi - lo : 1);
  for (int c = cx; c < C; c += 64) {
    float acc = 0.f;
    for (int32_t e = lo; e < hi; ++e) {
      const float v = msg[(int64_t)e * ldm + c];
      acc += mode == 0 ? v : v * sigmoidf_(att[(int64_t)e * lda + c]);
    }
    out[i * ldo + c] = mode == 0 ? acc * inv : acc;
  }
}

__global__ __launch_bounds__(TPB) void seg_wsum_bwd_kernel(const float* __restrict__ msg, int64_t ldm,
                                                           const float* __restrict__ att, int64_t lda,
                                                           const int32_t* __restrict__ offsets, int64_t M, int C, int mode,
                                                           const float* __restrict__ dout, int64_t lddo,
                                                           float* __restrict__ dmsg, int64_t lddm, float* __restrict__ datt,
                                                           int64_t ldda) {
  CCN_LANES;
  const int64_t i = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (i >= M) return;
  const int32_t lo = offsets[i], hi = offsets[i + 1];
  const float inv = 1.0f / (float)(hi - lo > 1 ? hi - lo : 1);
  for (int c = cx; c < C; c += 64) {
    const float g = dout[i * lddo + c];
    for (int32_t e = lo; e < hi; ++e) {
      if (mode == 0) {
        dmsg[(int64_t)e * lddm + c] = g * inv;
      } else {
        const float sg = sigmoidf_(att[(int64_t)e * lda + c]);
        dmsg[(int64_t)e * lddm + c] = g * sg;
        datt[(int64_t)e * ldda + c] = g * msg[(int64_t)e * ldm + c] * sg * (1.0f - sg);
      }
    }
  }
}

// Dense SGCNN rows (b, i, slot), dgcnn.py:182-203; slot 0 (self) is always valid, slot s > 0 iff its FRNN entry >= 0:
//   mode 0 'mean'         sum of the valid f / number of valid slots
//   mode 1 'weighted-sum' w = sigmoid(a) on the valid slots, 0 elsewhere; out = sum f w / clamp(sum w, min 1e-3)
//   mode 2 'attend'       a = -5e2 on the invalid slots, softmax over ALL K+1 slots, out = sum f softmax
__global__ __launch_bounds__(TPB) void sg_reduce_fwd_kernel(const float* __restrict__ f, int64_t ldf,
                                                            const float* __restrict__ att, int64_t lda,
                                                            const int64_t* __restrict__ idx,
                                                            const int64_t* __restrict__ cloud_ptr, int64_t B, int64_t Nmax,
                                                            int K, int C, int mode, float* __restrict__ out, int64_t ldo) {
  CCN_LANES;
  const int64_t bi = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (bi >= B * Nmax) return;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  if (i >= len) return;
  const SgNbrs nb = sg_nbrs(idx, bi, K, cx, true);
  const float* frow = f + bi * (K + 1) * ldf;
  const float* arow = att ? att + bi * (K + 1) * lda : nullptr;
  for (int c = cx; c < C; c += 64) {
    if (mode == 0) {
      float acc = 0.f;
      int cnt = 0;
      for (int s = 0; s <= K; ++s) {
        const bool ok = s == 0 || sg_nbr(nb, s) != -1;
        if (ok) {
          acc += frow[s * ldf + c];
          ++cnt;
        }
      }
      out[(base + i) * ldo + c] = acc / (float)cnt;
    } else if (mode == 1) {
      float acc = 0.f, tot = 0.f;
      for (int s = 0; s <= K; ++s) {
        const bool ok = s == 0 || sg_nbr(nb, s) != -1;
        if (ok) {
          const float w = sigmoidf_(arow[s * lda + c]);
          acc += frow[s * ldf + c] * w;
          tot += w;
        }
      }
      out[(base + i) * ldo + c] = acc / (tot > 1e-3f ? tot : 1e-3f);
    } else {
      SoftAcc sm = {-__builtin_inff(), 0.f, 0.f};
      for (int s = 0; s <= K; ++s) {
        const bool ok = s == 0 || sg_nbr(nb, s) != -1;
        soft_push(sm, ok ? arow[s * lda + c] : -5e2f, frow[s * ldf + c]);
      }
      out[(base + i) * ldo + c] = sm.acc / sm.tot;
    }
  }
}

__global__ __launch_bounds__(TPB) void sg_reduce_bwd_kernel(const float* __restrict__ f, int64_t ldf,
                                                            const float* __restrict__ att, int64_t lda,
                                                            const int64_t* __restrict__ idx,
                                                            const int64_t* __restrict__ cloud_ptr, int64_t B, int64_t Nmax,
                                                            int K, int C, int mode, const float* __restrict__ dout,
                                                            int64_t lddo, float* __restrict__ df, int64_t lddf,
                                                            float* __restrict__ datt, int64_t ldda) {
  CCN_LANES;
  const int64_t bi = (int64_t)ccn_xcd_block() * ROWS_PER_WG + ry;
  if (bi >= B * Nmax) return;
  const int64_t b = bi / Nmax, i = bi - b * Nmax;
  const int64_t base = cloud_ptr[b], len = cloud_ptr[b + 1] - base;
  const bool live = i < len;
  const SgNbrs nb = sg_nbrs(idx, bi, K, cx, live);
  const float* frow = f + bi * (K + 1) * ldf;
  const float* arow = att ? att + bi * (K + 1) * lda : nullptr;
  float* dfrow = df + bi * (K + 1) * lddf;
  float* darow = datt ? datt + bi * (K + 1) * ldda : nullptr;
  for (int c = cx; c < C; c += 64) {
    if (!live) {  // padding rows of the dense layout receive no gradient
      for (int s = 0; s <= K; ++s) {
        dfrow[s * lddf + c] = 0.f;
        if (darow) darow[s * ldda + c] = 0.f;
      }
      continue;
    }
    const float g = dout[(base + i) * lddo + c];
    if (mode == 0) {
      int cnt = 0;
      for (int s = 0; s <= K; ++s) cnt += (s == 0 || sg_nbr(nb, s) != -1) ? 1 : 0;
      const float w = g / (float)cnt;
      for (int s = 0; s <= K; ++s) dfrow[s * lddf + c] = (s == 0 || sg_nbr(nb, s) != -1) ? w : 0.f;
    } else if (mode == 1) {
      float acc = 0.f, tot = 0.f;
      for (int s = 0; s <= K; ++s)
        if (s == 0 || sg_nbr(nb, s) != -1) {
          const float w = sigmoidf_(arow[s * lda + c]);
          acc += frow[s * ldf + c] * w;
          tot += w;
        }
      const bool clamped = !(tot > 1e-3f);
      const float den = clamped ? 1e-3f : tot;
      const float o = acc / den;
      for (int s = 0; s <= K; ++s) {
        const bool ok = s == 0 || sg_nbr(nb, s) != -1;
        float dfv = 0.f, dav = 0.f;
        if (ok) {
          const float w = sigmoidf_(arow[s * lda + c]);
          dfv = g * w / den;
          // d out / d w_s = (f_s - out) / den (the clamp passes no gradient to the total)
          const float dw = g * (frow[s * ldf + c] - (clamped ? 0.f : o)) / den;
          dav = dw * w * (1.0f - w);
        }
        dfrow[s * lddf + c] = dfv;
        darow[s * ldda + c] = dav;
      }
    } else {
      SoftAcc sm = {-__builtin_inff(), 0.f, 0.f};
      for (int s = 0; s <= K; ++s) {
        const bool ok = s == 0 || sg_nbr(nb, s) != -1;
        soft_push(sm, ok ? arow[s * lda + c] : -5e2f, frow[s * ldf + c]);
      }
      const float inv = 1.0f / sm.tot;
      const float o = sm.acc * inv;
      for (int s = 0; s <= K; ++s) {
        const bool ok = s == 0 || sg_nbr(nb, s) != -1;
        const float w = __expf((ok ? arow[s * lda + c] : -5e2f) - sm.top) * inv;
        dfrow[s * lddf + c] = g * w;                      // (the reference's invalid rows receive w = exp(-500 - max) = 0)
        darow[s * ldda + c] = ok ? g * w * (frow[s * ldf + c] - o) : 0.f;   // masked entries were overwritten: no gradient
      }
    }
  }
}

inline unsigned row_blocks(int64_t rows) { return (unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG); }

}  // namespace

#define CCN_SMALL_INT(v) ((v) > 0 && (v) < (1 << 30))

extern "C" {

static int sg_pts(int64_t B, int64_t Nmax, int64_t Co) {
  // aim at >= 2048 workgroups: groups * channel chunks
  const int64_t chunks = (Co + 63) / 64;
  int64_t pts = B * Nmax * chunks / (4 * 2048);
  if (pts > SG_PTS_MAX) pts = SG_PTS_MAX;
  if (pts < 1) pts = 1;
  return (int)pts;
}

int64_t ccn_sg_edge_stats_rows(int64_t B, int64_t Nmax, int64_t Co) {
  const int pts = sg_pts(B, Nmax, Co);
  return (B * Nmax + 4 * pts - 1) / (4 * pts);
}

int ccn_sg_edge_stats(const float* ps, int64_t ldps, const float* pad, const int64_t* idx, const int64_t* cloud_ptr,
                      int64_t B, int64_t Nmax, int64_t K, int64_t Co, double* partial, void* stream) {
  CCN_REQUIRE(ps && idx && cloud_ptr && partial && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(Co) &&
                  ldps >= 2 * Co,
              "sg_edge_stats: bad arguments");
  hipLaunchKernelGGL(sg_edge_stats_kernel<0>,
                     dim3((unsigned)ccn_sg_edge_stats_rows(B, Nmax, Co), (unsigned)((Co + 63) / 64)), dim3(TPB), 0,
                     (hipStream_t)stream, ps, ldps, pad, idx, cloud_ptr, B, Nmax, (int)K, (int)Co,
                     (const float*)nullptr, (int64_t)0, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, 0, 0.f, sg_pts(B, Nmax, Co), partial);
  CCN_LAUNCH_OK("sg_edge_stats");
  return CCN_OK;
}

int ccn_sg_edge_apply(const float* ps, int64_t ldps, const float* pad, const int64_t* idx, const int64_t* cloud_ptr,
                      int64_t B, int64_t Nmax, int64_t K, int64_t Co, const float* scale, const float* shift, int act,
                      float slope, float* Z, int64_t ldz, void* stream) {
  CCN_REQUIRE(ps && idx && cloud_ptr && Z && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(Co) &&
                  ldps >= 2 * Co && ldz >= Co,
              "sg_edge_apply: bad arguments");
  hipLaunchKernelGGL(sg_edge_apply_kernel, dim3(row_blocks(B * Nmax)), dim3(TPB), 0, (hipStream_t)stream, ps, ldps, pad,
                     idx, cloud_ptr, B, Nmax, (int)K, (int)Co, scale, shift, act, slope, Z, ldz);
  CCN_LAUNCH_OK("sg_edge_apply");
  return CCN_OK;
}

int ccn_sg_edge_bwd_stats(const float* ps, int64_t ldps, const float* pad, const int64_t* idx,
                          const int64_t* cloud_ptr, int64_t B, int64_t Nmax, int64_t K, int64_t Co, const float* dZ,
                          int64_t lddz, const float* scale, const float* shift, const float* mean, const float* rstd,
                          int act, float slope, double* partial, void* stream) {
  CCN_REQUIRE(ps && idx && cloud_ptr && dZ && scale && shift && mean && rstd && partial && B > 0 && Nmax > 0 &&
                  CCN_SMALL_INT(K) && CCN_SMALL_INT(Co) && ldps >= 2 * Co && lddz >= Co,
              "sg_edge_bwd_stats: bad arguments");
  hipLaunchKernelGGL(sg_edge_stats_kernel<1>,
                     dim3((unsigned)ccn_sg_edge_stats_rows(B, Nmax, Co), (unsigned)((Co + 63) / 64)), dim3(TPB), 0,
                     (hipStream_t)stream, ps, ldps, pad, idx, cloud_ptr, B, Nmax, (int)K, (int)Co, dZ, lddz, scale,
                     shift, mean, rstd, act, slope, sg_pts(B, Nmax, Co), partial);
  CCN_LAUNCH_OK("sg_edge_bwd_stats");
  return CCN_OK;
}

int ccn_sg_edge_bwd(const float* ps, int64_t ldps, const float* pad, const int64_t* idx, const int64_t* cloud_ptr,
                    int64_t B, int64_t Nmax, int64_t K, int64_t Co, const float* dZ, int64_t lddz, const float* scale,
                    const float* shift, const float* mean, const float* rstd, int act, float slope, const double* sums,
                    int training, float* dps, int64_t lddps, void* stream) {
  CCN_REQUIRE(ps && idx && cloud_ptr && dZ && dps && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(Co) &&
                  ldps >= 2 * Co && lddz >= Co && lddps >= 2 * Co,
              "sg_edge_bwd: bad arguments");
  hipLaunchKernelGGL(sg_edge_bwd_kernel, dim3(row_blocks(B * Nmax)), dim3(TPB), 0, (hipStream_t)stream, ps, ldps, pad,
                     idx, cloud_ptr, B, Nmax, (int)K, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, sums,
                     B * Nmax * (K + 1), training, dps, lddps);
  CCN_LAUNCH_OK("sg_edge_bwd");
  return CCN_OK;
}

int ccn_sg_gather_fwd(const float* x, int64_t ldx, const int64_t* idx, const int64_t* cloud_ptr, int64_t B,
                      int64_t Nmax, int64_t K, int64_t C, float* feat, int64_t ldf, void* stream) {
  CCN_REQUIRE(x && idx && cloud_ptr && feat && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(C) &&
                  ldx >= C && ldf >= 2 * C,
              "sg_gather_fwd: bad arguments");
  hipLaunchKernelGGL(sg_gather_fwd_kernel, dim3(row_blocks(B * Nmax)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, idx,
                     cloud_ptr, B, Nmax, (int)K, (int)C, feat, ldf);
  CCN_LAUNCH_OK("sg_gather_fwd");
  return CCN_OK;
}

int ccn_sg_gather_bwd(const float* dfeat, int64_t lddf, const int64_t* idx, const int64_t* cloud_ptr, int64_t B,
                      int64_t Nmax, int64_t K, int64_t C, float* dx, int64_t lddx, void* stream) {
  // dx (packed N x C) must be zero on entry
  CCN_REQUIRE(dfeat && idx && cloud_ptr && dx && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(C) &&
                  lddx >= C && lddf >= 2 * C,
              "sg_gather_bwd: bad arguments");
  hipLaunchKernelGGL(sg_gather_bwd_kernel, dim3(row_blocks(B * Nmax)), dim3(TPB), 0, (hipStream_t)stream, dfeat, lddf,
                     idx, cloud_ptr, B, Nmax, (int)K, (int)C, dx, lddx);
  CCN_LAUNCH_OK("sg_gather_bwd");
  return CCN_OK;
}

int ccn_sg_max_fwd(const float* f, int64_t ldf, const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax,
                   int64_t K, int64_t C, float* out, int64_t ldo, int32_t* arg, void* stream) {
  CCN_REQUIRE(f && idx && cloud_ptr && out && arg && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(C) &&
                  ldo >= C && ldf >= C,
              "sg_max_fwd: bad arguments");
  hipLaunchKernelGGL(sg_max_fwd_kernel, dim3(row_blocks(B * Nmax)), dim3(TPB), 0, (hipStream_t)stream, f, ldf, idx,
                     cloud_ptr, B, Nmax, (int)K, (int)C, out, ldo, arg);
  CCN_LAUNCH_OK("sg_max_fwd");
  return CCN_OK;
}

int ccn_sg_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int64_t* cloud_ptr, int64_t B,
                   int64_t Nmax, int64_t K, int64_t C, float* df, int64_t lddf, void* stream) {
  CCN_REQUIRE(dout && arg && cloud_ptr && df && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(C) &&
                  lddo >= C && lddf >= C,
              "sg_max_bwd: bad arguments");
  hipLaunchKernelGGL(sg_max_bwd_kernel, dim3(row_blocks(B * Nmax)), dim3(TPB), 0, (hipStream_t)stream, dout, lddo, arg,
                     cloud_ptr, B, Nmax, (int)K, (int)C, df, lddf);
  CCN_LAUNCH_OK("sg_max_bwd");
  return CCN_OK;
}

int ccn_seg_wsum_fwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets, int64_t M,
                     int64_t C, int mode, float* out, int64_t ldo, void* stream) {
  CCN_REQUIRE(msg && offsets && out && M >= 0 && CCN_SMALL_INT(C) && ldm >= C && ldo >= C && (mode == 0 || (mode == 1 && att)),
              "seg_wsum_fwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_wsum_fwd_kernel, dim3(row_blocks(M)), dim3(TPB), 0, (hipStream_t)stream, msg, ldm, att, lda, offsets,
                     M, (int)C, mode, out, ldo);
  CCN_LAUNCH_OK("seg_wsum_fwd");
  return CCN_OK;
}

int ccn_seg_wsum_bwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets, int64_t M,
                     int64_t C, int mode, const float* dout, int64_t lddo, float* dmsg, int64_t lddm, float* datt,
                     int64_t ldda, void* stream) {
  CCN_REQUIRE(msg && offsets && dout && dmsg && M >= 0 && CCN_SMALL_INT(C) && (mode == 0 || (mode == 1 && att && datt)),
              "seg_wsum_bwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_wsum_bwd_kernel, dim3(row_blocks(M)), dim3(TPB), 0, (hipStream_t)stream, msg, ldm, att, lda, offsets,
                     M, (int)C, mode, dout, lddo, dmsg, lddm, datt, ldda);
  CCN_LAUNCH_OK("seg_wsum_bwd");
  return CCN_OK;
}

int ccn_sg_reduce_fwd(const float* f, int64_t ldf, const float* att, int64_t lda, const int64_t* idx, const int64_t* cloud_ptr,
                      int64_t B, int64_t Nmax, int64_t K, int64_t C, int mode, float* out, int64_t ldo, void* stream) {
  CCN_REQUIRE(f && idx && cloud_ptr && out && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(C) && ldo >= C && ldf >= C &&
                  mode >= 0 && mode <= 2 && (mode == 0 || att),
              "sg_reduce_fwd: bad arguments");
  hipLaunchKernelGGL(sg_reduce_fwd_kernel, dim3(row_blocks(B * Nmax)), dim3(TPB), 0, (hipStream_t)stream, f, ldf, att, lda, idx,
                     cloud_ptr, B, Nmax, (int)K, (int)C, mode, out, ldo);
  CCN_LAUNCH_OK("sg_reduce_fwd");
  return CCN_OK;
}

int ccn_sg_reduce_bwd(const float* f, int64_t ldf, const float* att, int64_t lda, const int64_t* idx, const int64_t* cloud_ptr,
                      int64_t B, int64_t Nmax, int64_t K, int64_t C, int mode, const float* dout, int64_t lddo, float* df,
                      int64_t lddf, float* datt, int64_t ldda, void* stream) {
  CCN_REQUIRE(f && idx && cloud_ptr && dout && df && B > 0 && Nmax > 0 && CCN_SMALL_INT(K) && CCN_SMALL_INT(C) && mode >= 0 &&
                  mode <= 2 && (mode == 0 || (att && datt)),
              "sg_reduce_bwd: bad arguments");
  hipLaunchKernelGGL(sg_reduce_bwd_kernel, dim3(row_blocks(B * Nmax)), dim3(TPB), 0, (hipStream_t)stream, f, ldf, att, lda, idx,
                     cloud_ptr, B, Nmax, (int)K, (int)C, mode, dout, lddo, df, lddf, datt, ldda);
  CCN_LAUNCH_OK("sg_reduce_bwd");
  return CCN_OK;
}

// ---- compact SGCNN rows
int ccn_cg_count(const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax, int64_t K, int32_t* cnt,
                 int32_t* has_rep, void* stream) {
  CCN_REQUIRE(idx && cloud_ptr && cnt && has_rep && B > 0 && B < 65536 && Nmax > 0 && K >= 1 && K <= 63,
              "cg_count: bad arguments (K <= 63)");
  hipLaunchKernelGGL(cg_count_kernel, dim3(ccn_blocks(Nmax, 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, idx,
                     cloud_ptr, Nmax, (int)K, cnt, has_rep);
  CCN_LAUNCH_OK("cg_count");
  return CCN_OK;
}

int ccn_cg_fill(const int64_t* idx, const int64_t* cloud_ptr, int64_t B, int64_t Nmax, int64_t K, const int32_t* grp_ptr,
                const int32_t* rep_off, int64_t E, int32_t* row_src, int32_t* rep_row, float* row_w, void* stream) {
  CCN_REQUIRE(idx && cloud_ptr && grp_ptr && rep_off && row_src && rep_row && row_w && B > 0 && B < 65536 && Nmax > 0 &&
                  K >= 1 && K <= 63 && E > 0 && E < 2147483647LL,
              "cg_fill: bad arguments");
  hipLaunchKernelGGL(cg_fill_kernel, dim3(ccn_blocks(Nmax, 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, idx,
                     cloud_ptr, Nmax, (int)K, grp_ptr, rep_off, E, row_src, rep_row, row_w);
  CCN_LAUNCH_OK("cg_fill");
  return CCN_OK;
}

static int cg_pts(int64_t N, int64_t Co) {
  const int64_t chunks = (Co + 63) / 64;
  int64_t pts = (N + 1) * chunks / (4 * 2048);
  if (pts > SG_PTS_MAX) pts = SG_PTS_MAX;
  if (pts < 1) pts = 1;
  return (int)pts;
}

int64_t ccn_cg_edge_stats_rows(int64_t N, int64_t Co) {
  const int pts = cg_pts(N, Co);
  return (N + 1 + 4 * pts - 1) / (4 * pts);
}

int ccn_cg_edge_stats(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                      const int32_t* rep_row, const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co,
                      double* partial, void* stream) {
  CCN_REQUIRE(ps && grp_ptr && row_src && rep_row && row_w && partial && N > 0 && CCN_SMALL_INT(Co) && ldps >= 2 * Co,
              "cg_edge_stats: bad arguments");
  hipLaunchKernelGGL((cg_edge_stats_kernel<0, 0>), dim3((unsigned)ccn_cg_edge_stats_rows(N, Co), (unsigned)((Co + 63) / 64)),
                     dim3(TPB), 0, (hipStream_t)stream, ps, ldps, grp_ptr, row_src, rep_row, row_w, N, E, Ne, (int)Co,
                     (const float*)nullptr, (int64_t)0, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, 0, 0.f, cg_pts(N, Co), partial);
  CCN_LAUNCH_OK("cg_edge_stats");
  return CCN_OK;
}

static int cg_edge_apply_impl(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                              const int32_t* rep_row, int64_t N, int64_t E, int64_t Ne, int64_t Co, const float* scale,
                              const float* shift, int act, float slope, void* Z, int64_t ldz, int zt, void* stream) {
  CCN_REQUIRE(ps && grp_ptr && row_src && rep_row && Z && N > 0 && CCN_SMALL_INT(Co) && ldps >= 2 * Co && ldz >= Co,
              "cg_edge_apply: bad arguments");
#define CCN_CG_APPLY(ZT_)                                                                                                  \
  hipLaunchKernelGGL(cg_edge_apply_kernel<ZT_>, dim3(row_blocks(N + 1)), dim3(TPB), 0, (hipStream_t)stream, ps, ldps,      \
                     grp_ptr, row_src, rep_row, N, E, Ne, (int)Co, scale, shift, act, slope, Z, ldz)
  if (zt == 0) CCN_CG_APPLY(0);
  else if (zt == 1) CCN_CG_APPLY(1);
  else CCN_CG_APPLY(2);
#undef CCN_CG_APPLY
  CCN_LAUNCH_OK("cg_edge_apply");
  return CCN_OK;
}

int ccn_cg_edge_apply(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                      const int32_t* rep_row, int64_t N, int64_t E, int64_t Ne, int64_t Co, const float* scale,
                      const float* shift, int act, float slope, float* Z, int64_t ldz, void* stream) {
  return cg_edge_apply_impl(ps, ldps, grp_ptr, row_src, rep_row, N, E, Ne, Co, scale, shift, act, slope, Z, ldz, 0, stream);
}

// ... the activation written as 16-bit rows (bf16, or fp16 when f16 != 0): every row's columns [0, Co) are written, Co % 8 == 0
// (no padding columns to clear), ldz in 16-bit elements
int ccn_cg_edge_apply_h(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                        const int32_t* rep_row, int64_t N, int64_t E, int64_t Ne, int64_t Co, const float* scale,
                        const float* shift, int act, float slope, void* Z, int64_t ldz, int f16, void* stream) {
  CCN_REQUIRE(Co % 8 == 0 && ldz % 8 == 0 && ((uintptr_t)Z & 15) == 0, "cg_edge_apply_h: rows of Co % 8 == 0 elements, 16-byte aligned");
  return cg_edge_apply_impl(ps, ldps, grp_ptr, row_src, rep_row, N, E, Ne, Co, scale, shift, act, slope, Z, ldz, f16 ? 2 : 1,
                            stream);
}

static int cg_edge_bwd_stats_impl(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                                  const int32_t* rep_row, const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co,
                                  const void* dZ, int dz16, int64_t lddz, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, int act, float slope, double* partial,
                                  void* stream) {
  CCN_REQUIRE(ps && grp_ptr && row_src && rep_row && row_w && dZ && scale && shift && mean && rstd && partial && N > 0 &&
                  CCN_SMALL_INT(Co) && ldps >= 2 * Co && lddz >= Co,
              "cg_edge_bwd_stats: bad arguments");
  const dim3 grid((unsigned)ccn_cg_edge_stats_rows(N, Co), (unsigned)((Co + 63) / 64));
  if (dz16)
    hipLaunchKernelGGL((cg_edge_stats_kernel<1, 1>), grid, dim3(TPB), 0, (hipStream_t)stream, ps, ldps, grp_ptr, row_src,
                       rep_row, row_w, N, E, Ne, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, cg_pts(N, Co), partial);
  else
    hipLaunchKernelGGL((cg_edge_stats_kernel<1, 0>), grid, dim3(TPB), 0, (hipStream_t)stream, ps, ldps, grp_ptr, row_src,
                       rep_row, row_w, N, E, Ne, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, cg_pts(N, Co), partial);
  CCN_LAUNCH_OK("cg_edge_bwd_stats");
  return CCN_OK;
}

int ccn_cg_edge_bwd_stats(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                          const int32_t* rep_row, const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co,
                          const float* dZ, int64_t lddz, const float* scale, const float* shift, const float* mean,
                          const float* rstd, int act, float slope, double* partial, void* stream) {
  return cg_edge_bwd_stats_impl(ps, ldps, grp_ptr, row_src, rep_row, row_w, N, E, Ne, Co, dZ, 0, lddz, scale, shift, mean, rstd,
                                act, slope, partial, stream);
}

// ... with dZ as bf16 rows (the gradient of a 16-bit activation; lddz in 16-bit elements)
int ccn_cg_edge_bwd_stats_h(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                            const int32_t* rep_row, const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co,
                            const void* dZ, int64_t lddz, const float* scale, const float* shift, const float* mean,
                            const float* rstd, int act, float slope, double* partial, void* stream) {
  return cg_edge_bwd_stats_impl(ps, ldps, grp_ptr, row_src, rep_row, row_w, N, E, Ne, Co, dZ, 1, lddz, scale, shift, mean, rstd,
                                act, slope, partial, stream);
}

static int cg_edge_bwd_impl(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src,
                            const int32_t* rep_row, const float* row_w, int64_t N, int64_t E, int64_t Co, const void* dZ,
                            int dz16, int64_t lddz, const float* scale, const float* shift, const float* mean,
                            const float* rstd, int act, float slope, const double* sums, double count, int training,
                            float* dps, int64_t lddps, void* stream) {
  CCN_REQUIRE(ps && grp_ptr && row_src && rep_row && row_w && dZ && dps && N > 0 && CCN_SMALL_INT(Co) && ldps >= 2 * Co &&
                  lddz >= Co && lddps >= 2 * Co && count > 0,
              "cg_edge_bwd: bad arguments");
  // dP (the left half of every row) is accumulated atomically and must start from zero; dS (right half) is stored by its
  // owner.  One linear memset of the whole table: the 2-D fill of the left half alone (hipMemset2DAsync) measured 73 us
  // against ~45 us for twice the bytes in one run.
  CCN_HIP(hipMemsetAsync(dps, 0, (size_t)N * (size_t)lddps * sizeof(float), (hipStream_t)stream), "cg_edge_bwd");
  if (dz16)
    hipLaunchKernelGGL(cg_edge_bwd_kernel<1>, dim3(row_blocks(N)), dim3(TPB), 0, (hipStream_t)stream, ps, ldps, grp_ptr,
                       row_src, rep_row, row_w, N, E, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, sums, count,
                       training, dps, lddps);
  else
    hipLaunchKernelGGL(cg_edge_bwd_kernel<0>, dim3(row_blocks(N)), dim3(TPB), 0, (hipStream_t)stream, ps, ldps, grp_ptr,
                       row_src, rep_row, row_w, N, E, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, sums, count,
                       training, dps, lddps);
  CCN_LAUNCH_OK("cg_edge_bwd");
  return CCN_OK;
}

int ccn_cg_edge_bwd(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src, const int32_t* rep_row,
                    const float* row_w, int64_t N, int64_t E, int64_t Co, const float* dZ, int64_t lddz,
                    const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                    const double* sums, double count, int training, float* dps, int64_t lddps, void* stream) {
  return cg_edge_bwd_impl(ps, ldps, grp_ptr, row_src, rep_row, row_w, N, E, Co, dZ, 0, lddz, scale, shift, mean, rstd, act, slope,
                          sums, count, training, dps, lddps, stream);
}

int ccn_cg_edge_bwd_h(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src, const int32_t* rep_row,
                      const float* row_w, int64_t N, int64_t E, int64_t Co, const void* dZ, int64_t lddz,
                      const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                      const double* sums, double count, int training, float* dps, int64_t lddps, void* stream) {
  return cg_edge_bwd_impl(ps, ldps, grp_ptr, row_src, rep_row, row_w, N, E, Co, dZ, 1, lddz, scale, shift, mean, rstd, act, slope,
                          sums, count, training, dps, lddps, stream);
}

// ---- round 5: atomics-free backward of the compact first SGCNN layer (see cg_edge_gather_kernel).  Three launches:
//   ccn_cg_edge_bwd_sums    = ccn_cg_edge_bwd_stats (column sums of g, g xhat) + the per-point sums pt (N x 2 Co) -- ONE pass over dZ
//   ccn_cg_edge_bwd_gather  per SOURCE point the sums pp (N x 2 Co) over the rows that read it, through the inverse row list
//   ccn_cg_edge_bwd_finish  dps = [dP | dS] from pt, pp and the reduced column sums (`sums`: 2 Co totals; ignored unless training)
// scale / shift / mean / rstd: the layer's BatchNorm table; without BatchNorm pass scale = 1, shift = mean = rstd = 0, training = 0.
int ccn_cg_edge_bwd_sums(const float* ps, int64_t ldps, const int32_t* grp_ptr, const int32_t* row_src, const int32_t* rep_row,
                         const float* row_w, int64_t N, int64_t E, int64_t Ne, int64_t Co, const void* dZ, int dz16, int64_t lddz,
                         const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                         double* partial, float* pt, int64_t ldpt, void* stream) {
  CCN_REQUIRE(ps && grp_ptr && row_src && rep_row && row_w && dZ && scale && shift && mean && rstd && partial && pt && N > 0 &&
                  CCN_SMALL_INT(Co) && ldps >= 2 * Co && lddz >= Co && ldpt >= 2 * Co,
              "cg_edge_bwd_sums: bad arguments");
  const dim3 grid((unsigned)ccn_cg_edge_stats_rows(N, Co), (unsigned)((Co + 63) / 64));
  if (dz16)
    hipLaunchKernelGGL((cg_edge_stats_kernel<1, 1>), grid, dim3(TPB), 0, (hipStream_t)stream, ps, ldps, grp_ptr, row_src,
                       rep_row, row_w, N, E, Ne, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, cg_pts(N, Co), partial,
                       pt, ldpt);
  else
    hipLaunchKernelGGL((cg_edge_stats_kernel<1, 0>), grid, dim3(TPB), 0, (hipStream_t)stream, ps, ldps, grp_ptr, row_src,
                       rep_row, row_w, N, E, Ne, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, cg_pts(N, Co), partial,
                       pt, ldpt);
  CCN_LAUNCH_OK("cg_edge_bwd_sums");
  return CCN_OK;
}

int ccn_cg_edge_bwd_gather(const float* ps, int64_t ldps, const int32_t* inv_ptr, const int32_t* inv_row, const int32_t* row_dst,
                           int64_t N, int64_t Co, const void* dZ, int dz16, int64_t lddz, const float* scale, const float* shift,
                           const float* mean, const float* rstd, int act, float slope, float* pp, int64_t ldpp, void* stream) {
  CCN_REQUIRE(ps && inv_ptr && inv_row && row_dst && dZ && scale && shift && mean && rstd && pp && N > 0 && CCN_SMALL_INT(Co) &&
                  ldps >= 2 * Co && lddz >= Co && ldpp >= 2 * Co,
              "cg_edge_bwd_gather: bad arguments");
  if (dz16)
    hipLaunchKernelGGL(cg_edge_gather_kernel<1>, dim3(row_blocks(N)), dim3(TPB), 0, (hipStream_t)stream, ps, ldps, inv_ptr,
                       inv_row, row_dst, N, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, pp, ldpp);
  else
    hipLaunchKernelGGL(cg_edge_gather_kernel<0>, dim3(row_blocks(N)), dim3(TPB), 0, (hipStream_t)stream, ps, ldps, inv_ptr,
                       inv_row, row_dst, N, (int)Co, dZ, lddz, scale, shift, mean, rstd, act, slope, pp, ldpp);
  CCN_LAUNCH_OK("cg_edge_bwd_gather");
  return CCN_OK;
}

int ccn_cg_edge_bwd_finish(const float* pt, int64_t ldpt, const float* pp, int64_t ldpp, const int32_t* grp_ptr,
                           const int32_t* rep_row, const float* row_w, const int32_t* inv_ptr, int64_t N, int64_t E, int64_t Co,
                           const float* scale, const double* sums, double count, int training, float* dps, int64_t lddps,
                           void* stream) {
  CCN_REQUIRE(pt && pp && grp_ptr && rep_row && row_w && inv_ptr && scale && dps && N > 0 && CCN_SMALL_INT(Co) && ldpt >= 2 * Co &&
                  ldpp >= 2 * Co && lddps >= 2 * Co && count > 0 && (sums || !training),
              "cg_edge_bwd_finish: bad arguments");
  hipLaunchKernelGGL(cg_edge_finish_kernel, dim3(row_blocks(N)), dim3(TPB), 0, (hipStream_t)stream, pt, ldpt, pp, ldpp, grp_ptr,
                     rep_row, row_w, inv_ptr, N, E, (int)Co, scale, sums, count, training, dps, lddps);
  CCN_LAUNCH_OK("cg_edge_bwd_finish");
  return CCN_OK;
}

int ccn_cg_max_fwd(const float* f, int64_t ldf, const int32_t* grp_ptr, const int32_t* rep_row, int64_t N, int64_t C,
                   float* out, int64_t ldo, int32_t* arg, void* stream) {
  CCN_REQUIRE(f && grp_ptr && rep_row && out && arg && N > 0 && CCN_SMALL_INT(C) && ldf >= C && ldo >= C,
              "cg_max_fwd: bad arguments");
  hipLaunchKernelGGL(cg_max_fwd_kernel, dim3(row_blocks(N)), dim3(TPB), 0, (hipStream_t)stream, f, ldf, grp_ptr, rep_row, N,
                     (int)C, out, ldo, arg);
  CCN_LAUNCH_OK("cg_max_fwd");
  return CCN_OK;
}

int ccn_cg_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* grp_ptr, const int32_t* rep_row,
                   int64_t N, int64_t R, int64_t C, float* df, int64_t lddf, void* stream) {
  CCN_REQUIRE(dout && arg && grp_ptr && rep_row && df && N > 0 && R > N && CCN_SMALL_INT(C) && lddo >= C && lddf >= C,
              "cg_max_bwd: bad arguments");
  hipLaunchKernelGGL(cg_max_bwd_kernel<0>, dim3(row_blocks(N + 1)), dim3(TPB), 0, (hipStream_t)stream, dout, lddo, arg,
                     grp_ptr, rep_row, N, R, (int)C, df, lddf);
  CCN_LAUNCH_OK("cg_max_bwd");
  return CCN_OK;
}

// ... df as bf16 rows (C % 8 == 0, lddf in 16-bit elements): the dY operand of the data- / weight-gradient products of the
// plain Linear in front of the max (ccn_gemm_nt_h / ccn_gemm_tn_h)
int ccn_cg_max_bwd_h(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* grp_ptr, const int32_t* rep_row,
                     int64_t N, int64_t R, int64_t C, void* df, int64_t lddf, void* stream) {
  CCN_REQUIRE(dout && arg && grp_ptr && rep_row && df && N > 0 && R > N && CCN_SMALL_INT(C) && lddo >= C && lddf >= C &&
                  C % 8 == 0 && lddf % 8 == 0 && ((uintptr_t)df & 15) == 0,
              "cg_max_bwd_h: bad arguments");
  hipLaunchKernelGGL(cg_max_bwd_kernel<1>, dim3(row_blocks(N + 1)), dim3(TPB), 0, (hipStream_t)stream, dout, lddo, arg,
                     grp_ptr, rep_row, N, R, (int)C, df, lddf);
  CCN_LAUNCH_OK("cg_max_bwd_h");
  return CCN_OK;
}

// ---- PointNetConv2 first layer, algebraic form (point_conv.py:35-93 with local_nn.lins[0] split as [Wx | Wp])
static int pn_per_wave(int64_t E, int64_t Co) {
  const int64_t chunks = (Co + 63) / 64;
  int64_t per = E * chunks / (4 * 2048);
  if (per > PN_EDGES_MAX) per = PN_EDGES_MAX;
  if (per < 1) per = 1;
  return (int)per;
}

int64_t ccn_pn_edge_stats_rows(int64_t E, int64_t Co) {
  const int per = pn_per_wave(E, Co);
  return (E + 4 * per - 1) / (4 * per);
}

int64_t ccn_pn_edge_bwd_rows(int64_t E) {
  // waves of the backward kernel (each owns one row of 4*Co partial sums): ~2048 workgroups of 4 waves
  int64_t per = (E + 8191) / 8192;
  if (per < 1) per = 1;
  return ((E + per - 1) / per + 3) / 4 * 4;
}

int ccn_pn_edge_stats(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                      const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                      int64_t Co, float radius, double* partial, void* stream) {
  CCN_REQUIRE(px && wp && pos_src && pos_dst && src && dst && partial && E > 0 && CCN_SMALL_INT(Co) && ldpx >= Co &&
                  ldwp >= 3,
              "pn_edge_stats: bad arguments");
  hipLaunchKernelGGL((pn_edge_stats_kernel<0, 0>), dim3((unsigned)ccn_pn_edge_stats_rows(E, Co), (unsigned)((Co + 63) / 64)),
                     dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp, ldwp, bias, pos_src, pos_dst, src, dst, E, (int)Co,
                     radius, (const float*)nullptr, (int64_t)0, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, 0, 0.f, pn_per_wave(E, Co), partial);
  CCN_LAUNCH_OK("pn_edge_stats");
  return CCN_OK;
}

static int pn_edge_apply_impl(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                              const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                              int64_t Co, float radius, const float* scale, const float* shift, int act, float slope,
                              void* Z, int64_t ldz, int zt, void* stream) {
  CCN_REQUIRE(px && wp && pos_src && pos_dst && src && dst && Z && E > 0 && CCN_SMALL_INT(Co) && ldpx >= Co &&
                  ldwp >= 3 && ldz >= Co,
              "pn_edge_apply: bad arguments");
  const dim3 grid((unsigned)((E + 4 * PN_APPLY_EDGES - 1) / (4 * PN_APPLY_EDGES)));
#define CCN_PN_APPLY(ZT_)                                                                                                 \
  hipLaunchKernelGGL(pn_edge_apply_kernel<ZT_>, grid, dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp, ldwp, bias, pos_src, \
                     pos_dst, src, dst, E, (int)Co, radius, scale, shift, act, slope, Z, ldz)
  if (zt == 0) CCN_PN_APPLY(0);
  else if (zt == 1) CCN_PN_APPLY(1);
  else CCN_PN_APPLY(2);
#undef CCN_PN_APPLY
  CCN_LAUNCH_OK("pn_edge_apply");
  return CCN_OK;
}

int ccn_pn_edge_apply(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                      const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                      int64_t Co, float radius, const float* scale, const float* shift, int act, float slope, float* Z,
                      int64_t ldz, void* stream) {
  return pn_edge_apply_impl(px, ldpx, wp, ldwp, bias, pos_src, pos_dst, src, dst, E, Co, radius, scale, shift, act, slope, Z, ldz,
                            0, stream);
}

// ... the activation written as 16-bit rows (as ccn_cg_edge_apply_h)
int ccn_pn_edge_apply_h(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                        const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                        int64_t Co, float radius, const float* scale, const float* shift, int act, float slope, void* Z,
                        int64_t ldz, int f16, void* stream) {
  CCN_REQUIRE(Co % 8 == 0 && ldz % 8 == 0 && ((uintptr_t)Z & 15) == 0, "pn_edge_apply_h: rows of Co % 8 == 0 elements, 16-byte aligned");
  return pn_edge_apply_impl(px, ldpx, wp, ldwp, bias, pos_src, pos_dst, src, dst, E, Co, radius, scale, shift, act, slope, Z, ldz,
                            f16 ? 2 : 1, stream);
}

static int pn_edge_bwd_stats_impl(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                                  const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst,
                                  int64_t E, int64_t Co, float radius, const void* dZ, int dz16, int64_t lddz,
                                  const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                                  float slope, double* partial, void* stream) {
  CCN_REQUIRE(px && wp && pos_src && pos_dst && src && dst && dZ && scale && shift && mean && rstd && partial && E > 0 &&
                  CCN_SMALL_INT(Co) && ldpx >= Co && ldwp >= 3 && lddz >= Co,
              "pn_edge_bwd_stats: bad arguments");
  const dim3 grid((unsigned)ccn_pn_edge_stats_rows(E, Co), (unsigned)((Co + 63) / 64));
  if (dz16)
    hipLaunchKernelGGL((pn_edge_stats_kernel<1, 1>), grid, dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp, ldwp, bias, pos_src,
                       pos_dst, src, dst, E, (int)Co, radius, dZ, lddz, scale, shift, mean, rstd, act, slope, pn_per_wave(E, Co),
                       partial);
  else
    hipLaunchKernelGGL((pn_edge_stats_kernel<1, 0>), grid, dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp, ldwp, bias, pos_src,
                       pos_dst, src, dst, E, (int)Co, radius, dZ, lddz, scale, shift, mean, rstd, act, slope, pn_per_wave(E, Co),
                       partial);
  CCN_LAUNCH_OK("pn_edge_bwd_stats");
  return CCN_OK;
}

int ccn_pn_edge_bwd_stats(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                          const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                          int64_t Co, float radius, const float* dZ, int64_t lddz, const float* scale,
                          const float* shift, const float* mean, const float* rstd, int act, float slope,
                          double* partial, void* stream) {
  return pn_edge_bwd_stats_impl(px, ldpx, wp, ldwp, bias, pos_src, pos_dst, src, dst, E, Co, radius, dZ, 0, lddz, scale, shift,
                                mean, rstd, act, slope, partial, stream);
}

int ccn_pn_edge_bwd_stats_h(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                            const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                            int64_t Co, float radius, const void* dZ, int64_t lddz, const float* scale,
                            const float* shift, const float* mean, const float* rstd, int act, float slope,
                            double* partial, void* stream) {
  return pn_edge_bwd_stats_impl(px, ldpx, wp, ldwp, bias, pos_src, pos_dst, src, dst, E, Co, radius, dZ, 1, lddz, scale, shift,
                                mean, rstd, act, slope, partial, stream);
}

static int pn_edge_bwd_impl(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                            const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                            int64_t Co, float radius, const void* dZ, int dz16, int64_t lddz, const float* scale,
                            const float* shift, const float* mean, const float* rstd, int act, float slope,
                            const double* sums, int training, float* dpx, int64_t lddpx, double* wpart, void* stream) {
  CCN_REQUIRE(px && wp && pos_src && pos_dst && src && dst && dZ && dpx && wpart && E > 0 && CCN_SMALL_INT(Co) &&
                  ldpx >= Co && ldwp >= 3 && lddz >= Co && lddpx >= Co,
              "pn_edge_bwd: bad arguments");
  const int64_t waves = ccn_pn_edge_bwd_rows(E);
  int64_t per = (E + 8191) / 8192;
  if (per < 1) per = 1;
  if (dz16)
    hipLaunchKernelGGL(pn_edge_bwd_kernel<1>, dim3((unsigned)(waves / 4)), dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp,
                       ldwp, bias, pos_src, pos_dst, src, dst, E, (int)Co, radius, dZ, lddz, scale, shift, mean, rstd, act,
                       slope, sums, training, (int)per, dpx, lddpx, wpart);
  else
    hipLaunchKernelGGL(pn_edge_bwd_kernel<0>, dim3((unsigned)(waves / 4)), dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp,
                       ldwp, bias, pos_src, pos_dst, src, dst, E, (int)Co, radius, dZ, lddz, scale, shift, mean, rstd, act,
                       slope, sums, training, (int)per, dpx, lddpx, wpart);
  CCN_LAUNCH_OK("pn_edge_bwd");
  return CCN_OK;
}

int ccn_pn_edge_bwd(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                    const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                    int64_t Co, float radius, const float* dZ, int64_t lddz, const float* scale, const float* shift,
                    const float* mean, const float* rstd, int act, float slope, const double* sums, int training,
                    float* dpx, int64_t lddpx, double* wpart, void* stream) {
  return pn_edge_bwd_impl(px, ldpx, wp, ldwp, bias, pos_src, pos_dst, src, dst, E, Co, radius, dZ, 0, lddz, scale, shift, mean,
                          rstd, act, slope, sums, training, dpx, lddpx, wpart, stream);
}

int ccn_pn_edge_bwd_h(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias,
                      const float* pos_src, const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E,
                      int64_t Co, float radius, const void* dZ, int64_t lddz, const float* scale, const float* shift,
                      const float* mean, const float* rstd, int act, float slope, const double* sums, int training,
                      float* dpx, int64_t lddpx, double* wpart, void* stream) {
  return pn_edge_bwd_impl(px, ldpx, wp, ldwp, bias, pos_src, pos_dst, src, dst, E, Co, radius, dZ, 1, lddz, scale, shift, mean,
                          rstd, act, slope, sums, training, dpx, lddpx, wpart, stream);
}

// ---- round 5: atomics-free backward of PointNetConv2's algebraic first layer (see pn_edge_sums_kernel)
//   ccn_pn_edge_bwd_sums    column sums: partial = [ccn_pn_edge_stats_rows(E, Co)][9 Co + 4] doubles -- ONE pass over dZ
//   ccn_pn_edge_bwd_gather  per SOURCE point the sums pp (Nsrc x 2 Co) over the edges that read it (inverse of src: inv_ptr, inv_edge)
//   ccn_pn_edge_bwd_finish  dpx (Nsrc x Co) and dw4 = [dWp[:, 0] | dWp[:, 1] | dWp[:, 2] | dbias] (4 x Co) from pp and the reduced sums
int ccn_pn_edge_bwd_sums(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias, const float* pos_src,
                         const float* pos_dst, const int64_t* src, const int64_t* dst, int64_t E, int64_t Co, float radius,
                         const void* dZ, int dz16, int64_t lddz, const float* scale, const float* shift, const float* mean,
                         const float* rstd, int act, float slope, double* partial, void* stream) {
  CCN_REQUIRE(px && wp && pos_src && pos_dst && src && dst && dZ && scale && shift && mean && rstd && partial && E > 0 &&
                  CCN_SMALL_INT(Co) && ldpx >= Co && ldwp >= 3 && lddz >= Co,
              "pn_edge_bwd_sums: bad arguments");
  const dim3 grid((unsigned)ccn_pn_edge_stats_rows(E, Co), (unsigned)((Co + 63) / 64));
  if (dz16)
    hipLaunchKernelGGL(pn_edge_sums_kernel<1>, grid, dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp, ldwp, bias, pos_src, pos_dst,
                       src, dst, E, (int)Co, radius, dZ, lddz, scale, shift, mean, rstd, act, slope, pn_per_wave(E, Co), partial);
  else
    hipLaunchKernelGGL(pn_edge_sums_kernel<0>, grid, dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp, ldwp, bias, pos_src, pos_dst,
                       src, dst, E, (int)Co, radius, dZ, lddz, scale, shift, mean, rstd, act, slope, pn_per_wave(E, Co), partial);
  CCN_LAUNCH_OK("pn_edge_bwd_sums");
  return CCN_OK;
}

int ccn_pn_edge_bwd_gather(const float* px, int64_t ldpx, const float* wp, int64_t ldwp, const float* bias, const float* pos_src,
                           const float* pos_dst, const int64_t* dst, const int32_t* inv_ptr, const int32_t* inv_edge,
                           int64_t Nsrc, int64_t Co, float radius, const void* dZ, int dz16, int64_t lddz, const float* scale,
                           const float* shift, const float* mean, const float* rstd, int act, float slope, float* pp,
                           int64_t ldpp, void* stream) {
  CCN_REQUIRE(px && wp && pos_src && pos_dst && dst && inv_ptr && inv_edge && dZ && scale && shift && mean && rstd && pp &&
                  Nsrc > 0 && CCN_SMALL_INT(Co) && ldpx >= Co && ldwp >= 3 && lddz >= Co && ldpp >= 2 * Co,
              "pn_edge_bwd_gather: bad arguments");
  if (dz16)
    hipLaunchKernelGGL(pn_edge_gather_kernel<1>, dim3(row_blocks(Nsrc)), dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp, ldwp,
                       bias, pos_src, pos_dst, dst, inv_ptr, inv_edge, Nsrc, (int)Co, radius, dZ, lddz, scale, shift, mean, rstd,
                       act, slope, pp, ldpp);
  else
    hipLaunchKernelGGL(pn_edge_gather_kernel<0>, dim3(row_blocks(Nsrc)), dim3(TPB), 0, (hipStream_t)stream, px, ldpx, wp, ldwp,
                       bias, pos_src, pos_dst, dst, inv_ptr, inv_edge, Nsrc, (int)Co, radius, dZ, lddz, scale, shift, mean, rstd,
                       act, slope, pp, ldpp);
  CCN_LAUNCH_OK("pn_edge_bwd_gather");
  return CCN_OK;
}

int ccn_pn_edge_bwd_finish(const float* pp, int64_t ldpp, const int32_t* inv_ptr, int64_t Nsrc, int64_t E, int64_t Co,
                           const float* scale, const double* sums, int training, float* dpx, int64_t lddpx, float* dw4,
                           void* stream) {
  CCN_REQUIRE(pp && inv_ptr && scale && sums && dpx && dw4 && Nsrc > 0 && E > 0 && CCN_SMALL_INT(Co) && ldpp >= 2 * Co &&
                  lddpx >= Co,
              "pn_edge_bwd_finish: bad arguments");
  hipLaunchKernelGGL(pn_edge_finish_kernel, dim3(row_blocks(Nsrc)), dim3(TPB), 0, (hipStream_t)stream, pp, ldpp, inv_ptr, Nsrc, E,
                     (int)Co, scale, sums, training, dpx, lddpx, dw4);
  CCN_LAUNCH_OK("pn_edge_bwd_finish");
  return CCN_OK;
}

int ccn_msg_build_fwd(const float* x_src, int64_t ldx, const float* pos_src, const float* pos_dst, const int64_t* src,
                      const int64_t* dst, int64_t E, int64_t C, float radius, float* msg, int64_t ldm, void* stream) {
  CCN_REQUIRE(pos_src && pos_dst && src && dst && msg && C >= 0 && C < (1 << 30) && (C == 0 || (x_src && ldx >= C)) &&
                  ldm >= C + 3,
              "msg_build_fwd: bad arguments");
  if (E == 0) return CCN_OK;
  hipLaunchKernelGGL(msg_build_fwd_kernel, dim3(row_blocks(E)), dim3(TPB), 0, (hipStream_t)stream, x_src, ldx, pos_src,
                     pos_dst, src, dst, E, (int)C, radius, msg, ldm);
  CCN_LAUNCH_OK("msg_build_fwd");
  return CCN_OK;
}

int ccn_msg_build_bwd(const float* dmsg, int64_t lddm, const int64_t* src, int64_t E, int64_t C, float* dx,
                      int64_t lddx, void* stream) {
  CCN_REQUIRE(dmsg && src && dx && CCN_SMALL_INT(C) && lddx >= C && lddm >= C + 3, "msg_build_bwd: bad arguments");
  if (E == 0) return CCN_OK;
  hipLaunchKernelGGL(msg_build_bwd_kernel, dim3(row_blocks(E)), dim3(TPB), 0, (hipStream_t)stream, dmsg, lddm, src, E,
                     (int)C, dx, lddx);
  CCN_LAUNCH_OK("msg_build_bwd");
  return CCN_OK;
}

int ccn_edge_feat_fwd(const float* x, int64_t ldx, const int64_t* src, const int64_t* dst, int64_t E, int64_t C,
                      float* msg, int64_t ldm, void* stream) {
  CCN_REQUIRE(x && src && dst && msg && CCN_SMALL_INT(C) && ldx >= C && ldm >= 2 * C, "edge_feat_fwd: bad arguments");
  if (E == 0) return CCN_OK;
  hipLaunchKernelGGL(edge_feat_fwd_kernel<0>, dim3(row_blocks(E)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, src, dst, E,
                     (int)C, msg, ldm);
  CCN_LAUNCH_OK("edge_feat_fwd");
  return CCN_OK;
}

// ... the message rows written as 16-bit rows (bf16, fp16 when f16 != 0; (2 C) % 8 == 0, ldm in 16-bit elements)
int ccn_edge_feat_fwd_h(const float* x, int64_t ldx, const int64_t* src, const int64_t* dst, int64_t E, int64_t C, void* msg,
                        int64_t ldm, int f16, void* stream) {
  CCN_REQUIRE(x && src && dst && msg && CCN_SMALL_INT(C) && ldx >= C && ldm >= 2 * C && (2 * C) % 8 == 0 && ldm % 8 == 0 &&
                  ((uintptr_t)msg & 15) == 0,
              "edge_feat_fwd_h: bad arguments");
  if (E == 0) return CCN_OK;
  if (f16)
    hipLaunchKernelGGL(edge_feat_fwd_kernel<2>, dim3(row_blocks(E)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, src, dst, E,
                       (int)C, msg, ldm);
  else
    hipLaunchKernelGGL(edge_feat_fwd_kernel<1>, dim3(row_blocks(E)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, src, dst, E,
                       (int)C, msg, ldm);
  CCN_LAUNCH_OK("edge_feat_fwd_h");
  return CCN_OK;
}

// backward over edges grouped by destination (offsets: num_dst + 1 int32, edge e of destination i in [offsets[i], offsets[i+1]));
// dmsg fp32 rows (dm16 = 0) or bf16 rows (dm16 = 1, lddm in 16-bit elements); accumulates into dx (N rows, zero on entry).
// Group i of the CSR list is POINT i of x (the destination side of the message is x[i]): num_dst must equal N.
int ccn_edge_feat_bwd_csr(const void* dmsg, int dm16, int64_t lddm, const int64_t* src, const int32_t* offsets, int64_t num_dst,
                          int64_t N, int64_t E, int64_t C, float* dx, int64_t lddx, void* stream) {
  CCN_REQUIRE(dmsg && src && offsets && dx && num_dst >= 0 && CCN_SMALL_INT(C) && lddx >= C && lddm >= 2 * C,
              "edge_feat_bwd_csr: bad arguments");
  CCN_REQUIRE(num_dst == N, "edge_feat_bwd_csr: %lld destination groups for %lld points (the grouped form needs group i = point i; "
              "use ccn_edge_feat_bwd for an edge list over a subset)", (long long)num_dst, (long long)N);
  if (E == 0 || num_dst == 0) return CCN_OK;
  if (dm16)
    hipLaunchKernelGGL(edge_feat_bwd_csr_kernel<1>, dim3(row_blocks(num_dst)), dim3(TPB), 0, (hipStream_t)stream, dmsg, lddm, src,
                       offsets, num_dst, (int)C, dx, lddx);
  else
    hipLaunchKernelGGL(edge_feat_bwd_csr_kernel<0>, dim3(row_blocks(num_dst)), dim3(TPB), 0, (hipStream_t)stream, dmsg, lddm, src,
                       offsets, num_dst, (int)C, dx, lddx);
  CCN_LAUNCH_OK("edge_feat_bwd_csr");
  return CCN_OK;
}

int ccn_edge_feat_bwd(const float* dmsg, int64_t lddm, const int64_t* src, const int64_t* dst, int64_t E, int64_t C,
                      float* dx, int64_t lddx, void* stream) {
  CCN_REQUIRE(dmsg && src && dst && dx && CCN_SMALL_INT(C) && lddx >= C && lddm >= 2 * C,
              "edge_feat_bwd: bad arguments");
  if (E == 0) return CCN_OK;
  hipLaunchKernelGGL(edge_feat_bwd_kernel, dim3(row_blocks(E)), dim3(TPB), 0, (hipStream_t)stream, dmsg, lddm, src, dst,
                     E, (int)C, dx, lddx);
  CCN_LAUNCH_OK("edge_feat_bwd");
  return CCN_OK;
}

int ccn_seg_softmax_agg_fwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets,
                            int64_t M, int64_t C, float* out, int64_t ldo, void* stream) {
  CCN_REQUIRE(msg && att && offsets && out && CCN_SMALL_INT(C) && ldo >= C && ldm >= C && lda >= C,
              "seg_softmax_agg_fwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_softmax_agg_fwd_kernel, dim3(row_blocks(M)), dim3(TPB), 0, (hipStream_t)stream, msg, ldm, att,
                     lda, offsets, M, (int)C, out, ldo);
  CCN_LAUNCH_OK("seg_softmax_agg_fwd");
  return CCN_OK;
}

int ccn_seg_softmax_agg_bwd(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets,
                            int64_t M, int64_t C, const float* dout, int64_t lddo, float* dmsg, int64_t lddm,
                            float* datt, int64_t ldda, void* stream) {
  CCN_REQUIRE(msg && att && offsets && dout && dmsg && datt && CCN_SMALL_INT(C) && lddo >= C && ldm >= C &&
                  lda >= C && lddm >= C && ldda >= C,
              "seg_softmax_agg_bwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_softmax_agg_bwd_kernel<0>, dim3(row_blocks(M)), dim3(TPB), 0, (hipStream_t)stream, msg, ldm, att,
                     lda, offsets, M, (int)C, dout, lddo, dmsg, lddm, datt, ldda);
  CCN_LAUNCH_OK("seg_softmax_agg_bwd");
  return CCN_OK;
}

// ... datt as bf16 rows (C % 8 == 0, ldda in 16-bit elements): the dY operand of the backward products of the plain Linear
// that produced att (attend_nn's last layer)
int ccn_seg_softmax_agg_bwd_h(const float* msg, int64_t ldm, const float* att, int64_t lda, const int32_t* offsets,
                              int64_t M, int64_t C, const float* dout, int64_t lddo, float* dmsg, int64_t lddm,
                              void* datt, int64_t ldda, void* stream) {
  CCN_REQUIRE(msg && att && offsets && dout && dmsg && datt && CCN_SMALL_INT(C) && lddo >= C && ldm >= C &&
                  lda >= C && lddm >= C && ldda >= C && C % 8 == 0 && ldda % 8 == 0 && ((uintptr_t)datt & 15) == 0,
              "seg_softmax_agg_bwd_h: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_softmax_agg_bwd_kernel<1>, dim3(row_blocks(M)), dim3(TPB), 0, (hipStream_t)stream, msg, ldm, att,
                     lda, offsets, M, (int)C, dout, lddo, dmsg, lddm, datt, ldda);
  CCN_LAUNCH_OK("seg_softmax_agg_bwd_h");
  return CCN_OK;
}

int ccn_seg_max_fwd(const float* msg, int64_t ldm, const int32_t* offsets, int64_t M, int64_t C, float* out,
                    int64_t ldo, int32_t* arg, void* stream) {
  CCN_REQUIRE(msg && offsets && out && arg && CCN_SMALL_INT(C) && ldo >= C && ldm >= C, "seg_max_fwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_max_fwd_kernel, dim3(row_blocks(M)), dim3(TPB), 0, (hipStream_t)stream, msg, ldm, offsets, M,
                     (int)C, out, ldo, arg);
  CCN_LAUNCH_OK("seg_max_fwd");
  return CCN_OK;
}

int ccn_seg_max_bwd(const float* dout, int64_t lddo, const int32_t* arg, const int32_t* offsets, int64_t M, int64_t C,
                    float* dmsg, int64_t lddm, void* stream) {
  CCN_REQUIRE(dout && arg && offsets && dmsg && CCN_SMALL_INT(C) && lddo >= C && lddm >= C,
              "seg_max_bwd: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(seg_max_bwd_kernel, dim3(row_blocks(M)), dim3(TPB), 0, (hipStream_t)stream, dout, lddo, arg,
                     offsets, M, (int)C, dmsg, lddm);
  CCN_LAUNCH_OK("seg_max_bwd");
  return CCN_OK;
}

}  // extern "C"

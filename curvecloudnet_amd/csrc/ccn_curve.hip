// Curve index algebra and the curve-local float kernels.
// Built with -ffp-contract=off: every float expression below is evaluated exactly as written so
// that integer results (sample indices, edge lists) are bit-identical to the CPU oracle.
#include "ccn_common.h"
#include <type_traits>

namespace {

constexpr int TPB = 256;

// ------------------------------------------------------------------ A1: segment pointers
__global__ void run_flags_kernel(const int64_t* __restrict__ ids, int64_t n, int32_t* __restrict__ flag,
                                 unsigned long long* __restrict__ meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int f = 1;
  if (i > 0) {
    const int64_t a = ids[i - 1], b = ids[i];
    f = b != a;
    if (b < a) atomicAdd(&meta[1], 1ULL);
  }
  flag[i] = f;
}

// rank = inclusive scan of flags; element i starts run rank[i]-1 when flag[i]
__global__ void run_starts_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ rank, int64_t n,
                                  int64_t* __restrict__ starts, int32_t* __restrict__ run_of,
                                  int64_t* __restrict__ meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rank[i] - 1;
  if (flag[i]) starts[r] = i;
  if (run_of) run_of[i] = r;
  if (i == n - 1) {
    starts[r + 1] = n;
    meta[0] = r + 1;
  }
}

// ------------------------------------------------------------------ A2: topology
__global__ void cloud_ptr_kernel(const int64_t* __restrict__ batch, int64_t n, int64_t num_clouds,
                                 int64_t* __restrict__ cloud_ptr, unsigned long long* __restrict__ meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t b = batch[i];
  if (b < 0 || b >= num_clouds) {
    atomicAdd(&meta[1], 1ULL);
    return;
  }
  if (i == 0) {
    if (b != 0) atomicAdd(&meta[1], 1ULL);
    cloud_ptr[0] = 0;
  } else {
    const int64_t a = batch[i - 1];
    if (b < a || b > a + 1) atomicAdd(&meta[1], 1ULL);  // unsorted, or a cloud id without points
    if (b != a) cloud_ptr[b] = i;
  }
  if (i == n - 1) {
    if (b != num_clouds - 1) atomicAdd(&meta[1], 1ULL);
    cloud_ptr[num_clouds] = n;
  }
}

// one thread: per-cloud curve-id offsets (exclusive running sum of "last local id + 1") and the longest cloud
__global__ void cloud_offsets_kernel(const int64_t* __restrict__ p2c, const int64_t* __restrict__ cloud_ptr,
                                     int64_t num_clouds, int64_t* __restrict__ curve_off, int64_t* __restrict__ meta) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int64_t run = 0, longest = 0;
  for (int64_t b = 0; b < num_clouds; ++b) {
    curve_off[b] = run;
    const int64_t lo = cloud_ptr[b], hi = cloud_ptr[b + 1];
    if (hi > lo) run += p2c[hi - 1] + 1;
    if (hi - lo > longest) longest = hi - lo;
  }
  meta[2] = longest;
  meta[3] = 0;
}

__global__ void glob_flags_kernel(const int64_t* __restrict__ batch, const int64_t* __restrict__ p2c,
                                  const int64_t* __restrict__ curve_off, int64_t n, int64_t num_clouds,
                                  int64_t* __restrict__ glob, int32_t* __restrict__ flag,
                                  unsigned long long* __restrict__ meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t b = batch[i];
  b = b < 0 ? 0 : (b >= num_clouds ? num_clouds - 1 : b);
  const int64_t g = p2c[i] + curve_off[b];
  glob[i] = g;
  int f = 1;
  if (i > 0) {
    int64_t pb = batch[i - 1];
    pb = pb < 0 ? 0 : (pb >= num_clouds ? num_clouds - 1 : pb);
    const int64_t pg = p2c[i - 1] + curve_off[pb];
    f = g != pg;
    if (g < pg) atomicAdd(&meta[1], 1ULL);
  }
  flag[i] = f;
}

__global__ void curve_ptr_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ rank, int64_t n,
                                 int32_t* __restrict__ cid, int32_t* __restrict__ curve_ptr,
                                 int64_t* __restrict__ meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rank[i] - 1;
  cid[i] = r;
  if (flag[i]) curve_ptr[r] = (int32_t)i;
  if (i == n - 1) {
    curve_ptr[r + 1] = (int32_t)n;
    meta[0] = r + 1;
  }
}

// ------------------------------------------------------------------ A3: feature differences
// v_i = (a_i + b_i) / max(1, nlinks), a_i = x[i+1]-x[i] if linked, b_i = x[i]-x[i-1] if linked
__device__ __forceinline__ float diff_raw(const float* __restrict__ x, int64_t ldx, int64_t i, int64_t c, bool lp,
                                          bool ln) {
  const float xi = x[i * ldx + c];
  const float a = ln ? x[(i + 1) * ldx + c] - xi : 0.0f;
  const float b = lp ? xi - x[(i - 1) * ldx + c] : 0.0f;
  const float cnt = (float)((int)lp + (int)ln);
  return (a + b) / (cnt < 1.0f ? 1.0f : cnt);
}

__global__ void diff_concat_fwd_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ cid,
                                       int64_t n, int64_t C, float* __restrict__ out, int64_t ldo) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * C) return;
  // (32-bit division whenever the element count allows it: a 64-bit one costs more than the pass itself -- 400 070 x 131:
  // 0.35 -> see profiles/r04)
  const int64_t i = n * C < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)C) : t / C;
  const int64_t c = t - i * C;
  const int32_t me = cid[i];
  const bool lp = i > 0 && cid[i - 1] == me, ln = i + 1 < n && cid[i + 1] == me;
  out[i * ldo + c] = x[i * ldx + c];
  out[i * ldo + C + c] = fabsf(diff_raw(x, ldx, i, c, lp, ln));
}

__device__ __forceinline__ float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

__global__ void diff_concat_bwd_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ cid,
                                       int64_t n, int64_t C, const float* __restrict__ g, int64_t ldg,
                                       float* __restrict__ dx, int64_t lddx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * C) return;
  const int64_t j = n * C < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)C) : t / C;
  const int64_t c = t - j * C;
  const int32_t me = cid[j];
  const bool l_m2 = j > 1 && cid[j - 2] == me && cid[j - 1] == me;  // link (j-2, j-1)
  const bool l_m1 = j > 0 && cid[j - 1] == me;                      // link (j-1, j)
  const bool l_p0 = j + 1 < n && cid[j + 1] == me;                  // link (j, j+1)
  const bool l_p1 = j + 2 < n && cid[j + 2] == me && l_p0;          // link (j+1, j+2)
  float acc = g[j * ldg + c];
  // s_i = sign(v_i) * gd_i / cnt_i
  if (l_m1) {  // point j-1: ahead edge touches x[j] with +1
    const float cnt = (float)((int)l_m2 + 1);
    acc += sgn(diff_raw(x, ldx, j - 1, c, l_m2, true)) * g[(j - 1) * ldg + C + c] / cnt;
  }
  {
    const int k = (int)l_m1 + (int)l_p0;
    if (k > 0) {
      const float coef = (float)((int)l_m1 - (int)l_p0);
      if (coef != 0.0f) acc += coef * sgn(diff_raw(x, ldx, j, c, l_m1, l_p0)) * g[j * ldg + C + c] / (float)k;
    }
  }
  if (l_p0) {  // point j+1: behind edge touches x[j] with -1
    const float cnt = (float)((int)l_p1 + 1);
    acc -= sgn(diff_raw(x, ldx, j + 1, c, true, l_p1)) * g[(j + 1) * ldg + C + c] / cnt;
  }
  dx[j * lddx + c] = acc;
}

// ------------------------------------------------------------------ A4: shifted rows
// Element-parallel: thread -> (row i, q = tap*C + c).  Row i of the shifted-row matrix is the contiguous span
// x[(i - taps/2)*C ...] with the taps that leave the curve zeroed, so reads and writes are both coalesced whatever C
// is (the narrow first layers, C = 8 / 32, would leave most of a wave idle with one row per wave).
template <int ZT>     // ZT: the shifted-row matrix as fp32 (0), bf16 (1) or fp16 (2) rows (ccn_common.h: st_el)
__global__ __launch_bounds__(256) void im2col_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const int32_t* __restrict__ seg, int64_t rows, int C,
                                                         int taps, void* __restrict__ col, int64_t ldcol) {
  const int W = taps * C;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * W) return;
  // 32-bit division whenever the element count allows it (a 64-bit one costs more than the copy itself)
  const int64_t i = rows * W < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)W) : t / W;
  const int q = (int)(t - i * W);
  const int tap = (int)((uint32_t)q / (uint32_t)C), c = q - tap * C;
  const int64_t j = i + tap - taps / 2;
  const bool ok = j >= 0 && j < rows && (seg == nullptr || seg[j] == seg[i]);
  st_el<ZT>(col, i * ldcol + q, ok ? x[j * ldx + c] : 0.0f);
}

// wide rows: one wave per output row, lanes stride over the channels of each tap (256 contiguous bytes per access)
template <int ZT>
__global__ __launch_bounds__(256) void im2col_fwd_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const int32_t* __restrict__ seg, int64_t rows, int C,
                                                              int taps, void* __restrict__ col, int64_t ldcol) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + ry;
  if (i >= rows) return;
  const int32_t me = seg ? seg[i] : 0;
  for (int tap = 0; tap < taps; ++tap) {
    const int64_t j = i + tap - taps / 2;
    const bool ok = j >= 0 && j < rows && (seg == nullptr || seg[j] == me);
    for (int c = cx; c < C; c += 64) st_el<ZT>(col, i * ldcol + tap * C + c, ok ? x[j * ldx + c] : 0.0f);
  }
}

template <int DT>     // DT = 1: dcol as bf16 rows (the gradient of a 16-bit shifted-row matrix)
__global__ __launch_bounds__(256) void im2col_bwd_kernel(const void* __restrict__ dcol, int64_t ldcol,
                                                         const int32_t* __restrict__ seg, int64_t rows, int C,
                                                         int taps, float* __restrict__ dx, int64_t lddx) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * C) return;
  const int64_t j = rows * C < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)C) : t / C;
  const int c = (int)(t - j * C);
  const int32_t me = seg ? seg[j] : 0;
  float acc = 0.0f;
  for (int tap = 0; tap < taps; ++tap) {
    const int64_t i = j - tap + taps / 2;  // output row whose tap `tap` read x[j]
    if (i >= 0 && i < rows && (seg == nullptr || seg[i] == me)) acc += ld_el<DT>(dcol, i * ldcol + tap * C + c);
  }
  dx[j * lddx + c] = acc;
}

// Convolution as "product first, shift-add second" (layers with C_in >> C_out on the zero-separated V2 sequence):
// P = X W_all^T holds, per row, the contribution of that row to each tap of each output channel (taps*C_out columns
// instead of the taps*C_in columns of the shifted-row matrix); Y[i][co] = b[co] + sum_tap P[i + tap - h][tap*C_out + co].
__global__ __launch_bounds__(256) void shift_add_fwd_kernel(const float* __restrict__ P, int64_t ldp,
                                                            const float* __restrict__ bias, int64_t rows, int Co, int taps,
                                                            float* __restrict__ Y, int64_t ldy) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * Co) return;
  const int64_t i = rows * Co < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)Co) : t / Co;
  const int co = (int)(t - i * Co);
  float acc = bias ? bias[co] : 0.f;
  for (int tap = 0; tap < taps; ++tap) {
    const int64_t j = i + tap - taps / 2;
    if (j >= 0 && j < rows) acc += P[j * ldp + tap * Co + co];
  }
  Y[i * ldy + co] = acc;
}

__global__ __launch_bounds__(256) void shift_add_bwd_kernel(const float* __restrict__ dY, int64_t lddy, int64_t rows, int Co,
                                                            int taps, float* __restrict__ dP, int64_t lddp) {
  const int W = taps * Co;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * W) return;
  const int64_t j = rows * W < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)W) : t / W;
  const int q = (int)(t - j * W);
  const int tap = (int)((uint32_t)q / (uint32_t)Co), co = q - tap * Co;
  const int64_t i = j - tap + taps / 2;  // the output row that read P[j] through tap `tap`
  dP[j * lddp + q] = (i >= 0 && i < rows) ? dY[i * lddy + co] : 0.f;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t lds_, const int64_t* __restrict__ index,
                                   int64_t m, int64_t C, float* __restrict__ dst, int64_t ldd) {
  const int64_t t = (int64_t)ccn_xcd_block() * blockDim.x + threadIdx.x;
  if (t >= m * C) return;
  const int64_t r = m * C < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)C) : t / C, c = t - r * C;
  dst[r * ldd + c] = src[index[r] * lds_ + c];
}

__global__ void scatter_rows_kernel(const float* __restrict__ src, int64_t lds_, const int64_t* __restrict__ index,
                                    int64_t m, int64_t C, float* __restrict__ dst, int64_t ldd, int accumulate) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * C) return;
  const int64_t r = m * C < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)C) : t / C, c = t - r * C;
  const float v = src[r * lds_ + c];
  float* p = dst + index[r] * ldd + c;
  if (accumulate)
    atomicAdd(p, v);
  else
    *p = v;
}

// scatter of the rows of a matrix into a LONGER zero-separated sequence whose target rows are strictly ascending (the curve
// convolutions' layout): one pass writes EVERY row of the destination -- row index[i] <- src row i (columns >= C zero), the
// rows between index[i-1] and index[i] (before index[0], after index[m-1]) zero -- instead of a memset of the whole
// sequence followed by the scatter.  One source row per wave.
__global__ __launch_bounds__(256) void scatter_rows_fill_kernel(const float* __restrict__ src, int64_t lds_,
                                                                const int64_t* __restrict__ index, int64_t m, int64_t C,
                                                                float* __restrict__ dst, int64_t ldd, int64_t total,
                                                                int64_t row_offset) {
  const int cx = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= m) return;
  const int64_t at = index[i] + row_offset;
  const int64_t lo = i == 0 ? 0 : index[i - 1] + row_offset + 1;
  const int64_t hi = i == m - 1 ? total : at + 1;
  for (int64_t c = cx; c < ldd; c += 64) {
    for (int64_t r = lo; r < at; ++r) dst[r * ldd + c] = 0.f;
    dst[at * ldd + c] = c < C ? src[i * lds_ + c] : 0.f;
    for (int64_t r = at + 1; r < hi; ++r) dst[r * ldd + c] = 0.f;
  }
}

// ------------------------------------------------------------------ shared curve helpers
// NB: sqrtf() is correctly rounded on gfx950 (hipcc default); the __fsqrt_rn intrinsic is NOT (measured: 15% of
// inputs differ from IEEE by one ulp), which flips arclength buckets.
__device__ __forceinline__ float edge_len(const float* __restrict__ pos, int64_t i) {  // |pos[i+1] - pos[i]|
  const float dx = pos[3 * (i + 1)] - pos[3 * i];
  const float dy = pos[3 * (i + 1) + 1] - pos[3 * i + 1];
  const float dz = pos[3 * (i + 1) + 2] - pos[3 * i + 2];
  return sqrtf(ccn_sqdist3(dx, dy, dz));
}

// ------------------------------------------------------------------ A7: CurveFPS
// step[i] = length of the in-curve edge ending at point i (0 for curve starts), as float64
__global__ void fps_steps_kernel(const float* __restrict__ pos, const int32_t* __restrict__ cid, int64_t n,
                                 double* __restrict__ step) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  if (i > 0 && cid[i - 1] == cid[i]) v = (double)edge_len(pos, i - 1);
  step[i] = v;
}

__device__ __forceinline__ float fps_bucket(const double* __restrict__ run, int64_t i, int32_t start, float spacing,
                                            float u) {
  // torch: cumsum accumulates in float64 and rounds each output to float32
  float arclen = (float)run[i] - (float)run[start];
  const float scaled = __ll2float_rn((long long)start * 117LL) * u;
  float phase = fmodf(scaled, spacing);
  if (phase != 0.0f && ((spacing < 0.0f) != (phase < 0.0f))) phase += spacing;  // python-style remainder
  arclen = arclen + phase;
  return rintf(__fdiv_rn(arclen, spacing));
}

__global__ void fps_keep_kernel(const double* __restrict__ run, const int32_t* __restrict__ cid,
                                const int32_t* __restrict__ curve_ptr, int64_t n, float spacing, float u,
                                int32_t* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t start = curve_ptr[cid[i]];
  int k = 1;
  if (i > 0 && i != start) {
    const float b1 = fps_bucket(run, i, start, spacing, u);
    const float b0 = fps_bucket(run, i - 1, curve_ptr[cid[i - 1]], spacing, u);
    k = (b1 - b0) != 0.0f;
  }
  keep[i] = k;
}

__global__ void compact_kernel(const int32_t* __restrict__ keep, const int32_t* __restrict__ rank_excl, int64_t n,
                               int64_t* __restrict__ out, int64_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keep[i]) out[rank_excl[i]] = i;
  if (i == n - 1) *count = (int64_t)rank_excl[i] + keep[i];
}

// ------------------------------------------------------------------ A8: radius group along curves
__global__ void curve_budget_kernel(const float* __restrict__ pos, const int32_t* __restrict__ curve_ptr, int64_t Q,
                                    float radius, float* __restrict__ budget, unsigned int* __restrict__ maxbits) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Q) return;
  const int32_t lo = curve_ptr[c], hi = curve_ptr[c + 1];
  float len = 0.0f;  // sequential float32 sum in point order (index_add_ semantics)
  for (int32_t i = lo; i + 1 < hi; ++i) len += edge_len(pos, i);
  const float cnt = (float)(hi - lo);
  const float mean_edge = __fdiv_rn(len, cnt);
  // the reference writes `radius / tensor`, which torch evaluates as reciprocal(tensor) * radius
  float b = ceilf(__frcp_rn(mean_edge) * radius);
  if (isinf(b)) b = 1.0f;
  budget[c] = b;
  // positive floats order like their bit patterns
  atomicMax(&maxbits[0], __float_as_uint(b > 0.0f ? b : 0.0f));
  atomicMax(&maxbits[1], __float_as_uint(cnt));
}

__global__ void curve_reach_kernel(const unsigned int* __restrict__ maxbits, float* __restrict__ budget, int64_t Q) {
  if (blockIdx.x == 0 && threadIdx.x == 0) budget[Q] = fminf(__uint_as_float(maxbits[0]), __uint_as_float(maxbits[1]));
}

// walk 0,-1,+1,-2,+2,... inside [lo,hi) up to `reach`; accept while fewer than `allow` accepted.
template <bool FILL>
__device__ __forceinline__ int group_walk(int64_t centre, int32_t lo, int32_t hi, int reach, float allow, int64_t q,
                                          int64_t* __restrict__ row, int64_t* __restrict__ col) {
  int taken = 0;
  if ((float)(taken + 1) <= allow) {
    if (FILL) { row[taken] = q; col[taken] = centre; }
    ++taken;
  } else {
    return 0;
  }
  for (int m = 1; m <= reach; ++m) {
    const int64_t left = centre - m, right = centre + m;
    if (left >= lo) {
      if ((float)(taken + 1) <= allow) {
        if (FILL) { row[taken] = q; col[taken] = left; }
        ++taken;
      } else {
        break;
      }
    }
    if (right < hi) {
      if ((float)(taken + 1) <= allow) {
        if (FILL) { row[taken] = q; col[taken] = right; }
        ++taken;
      } else {
        break;
      }
    }
    if (left < lo && right >= hi) break;
  }
  return taken;
}

template <bool FILL>
__global__ void group_subset_kernel(const int32_t* __restrict__ cid, const int32_t* __restrict__ curve_ptr,
                                    const int64_t* __restrict__ p2c, int64_t Q, const int64_t* __restrict__ idx,
                                    int64_t M, const float* __restrict__ budget, int32_t* __restrict__ counts,
                                    const int32_t* __restrict__ offsets, int64_t* __restrict__ row,
                                    int64_t* __restrict__ col, int64_t cap) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= M) return;
  // (cap: entries row / col hold; a group that would end past it is not written -- only with bounded counts whose
  // capacity was exceeded, the caller's overflow flag is up then)
  if (FILL && (int64_t)offsets[q + 1] > cap) return;
  const int64_t centre = idx[q];
  const int32_t c = cid[centre];
  int64_t local = p2c[centre];  // quirk Q3: LOCAL id indexes the global table
  local = local < 0 ? 0 : (local >= Q ? Q - 1 : local);
  const float allow = budget[local];
  const int reach = (int)budget[Q];
  if (FILL) {
    const int32_t o = offsets[q];
    group_walk<true>(centre, curve_ptr[c], curve_ptr[c + 1], reach, allow, q, row + o, col + o);
  } else {
    counts[q] = group_walk<false>(centre, curve_ptr[c], curve_ptr[c + 1], reach, allow, q, nullptr, nullptr);
  }
}

// ------------------------------------------------------------------ A9: k nearest sampled points on the curve
__global__ void mark_samples_kernel(const int64_t* __restrict__ idx, int64_t M, int32_t* __restrict__ taken) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m < M) taken[idx[m]] = 1;
}

constexpr int SUP_MAXK = 8;  // supports k <= 8 (2k+3 <= 19 candidates)

__global__ void group_superset_kernel(const float* __restrict__ pos, const int32_t* __restrict__ cid, int64_t n,
                                      const int64_t* __restrict__ idx, int64_t M, int k,
                                      const int32_t* __restrict__ upto, int64_t* __restrict__ nbr,
                                      float* __restrict__ weight) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
  const int32_t me = cid[i];
  const int64_t base = upto[i];
  float bd[SUP_MAXK];
  int64_t bi[SUP_MAXK];
#pragma unroll
  for (int t = 0; t < SUP_MAXK; ++t) {
    bd[t] = __builtin_inff();
    bi[t] = -1;
  }
  const int ncand = 2 * k + 3;
  for (int s = 0; s < ncand; ++s) {
    // candidate order 0,-1,+1,-2,+2,...; equal distances keep this order (stable insertion)
    const int mag = (s + 1) >> 1;
    const int64_t cand = base + ((s & 1) ? -mag : mag);
    if (cand < 0 || cand >= M) continue;
    const int64_t p = idx[cand];
    if (cid[p] != me) continue;
    float cd = sqrtf(ccn_sqdist3(pos[3 * p] - px, pos[3 * p + 1] - py, pos[3 * p + 2] - pz));
    int64_t ci = cand;
#pragma unroll
    for (int t = 0; t < SUP_MAXK; ++t) {  // static indices only: the lists stay in registers
      if (t < k) {
        const bool sw = cd < bd[t];
        const float td = bd[t];
        const int64_t ti = bi[t];
        bd[t] = sw ? cd : td;
        bi[t] = sw ? ci : ti;
        cd = sw ? td : cd;
        ci = sw ? ti : ci;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < SUP_MAXK; ++t) {
    if (t < k) {
      const bool ok = bi[t] >= 0;
      nbr[i * k + t] = bi[t];
      float w = 0.0f;
      if (ok) {
        const float d2 = bd[t] * bd[t];
        w = __frcp_rn(d2 < 1e-16f ? 1e-16f : d2);
      }
      weight[i * k + t] = w;
    }
  }
}

__global__ void interp_fwd_kernel(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ nbr,
                                  const float* __restrict__ weight, int64_t n, int k, int64_t C,
                                  float* __restrict__ y, int64_t ldy) {
  const int64_t t = (int64_t)ccn_xcd_block() * blockDim.x + threadIdx.x;
  if (t >= n * C) return;
  const int64_t i = n * C < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)C) : t / C, c = t - i * C;
  float num = 0.0f, den = 0.0f;
  for (int s = 0; s < k; ++s) {
    const int64_t m = nbr[i * k + s];
    if (m < 0) break;
    const float w = weight[i * k + s];
    num += x[m * ldx + c] * w;
    den += w;
  }
  y[i * ldy + c] = num / den;
}

__global__ void interp_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const int64_t* __restrict__ nbr,
                                  const float* __restrict__ weight, int64_t n, int k, int64_t C,
                                  float* __restrict__ dx, int64_t lddx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * C) return;
  const int64_t i = n * C < 0xffffffffLL ? (int64_t)((uint32_t)t / (uint32_t)C) : t / C, c = t - i * C;
  float den = 0.0f;
  for (int s = 0; s < k; ++s) {
    if (nbr[i * k + s] < 0) break;
    den += weight[i * k + s];
  }
  const float g = dy[i * lddy + c] / den;
  for (int s = 0; s < k; ++s) {
    const int64_t m = nbr[i * k + s];
    if (m < 0) break;
    atomicAdd(&dx[m * lddx + c], g * weight[i * k + s]);
  }
}

// ---- interpolation backward without atomics: the inverse neighbour lists (built once per forward, on the geometry stream)
// Every coarse row m gets the list of (fine row, weight) pairs that interpolate from it, sorted by fine row, so the
// backward pass is a gather with a fixed summation order: deterministic, and 4 C (k + 1) bytes per fine row read at the
// gather rate instead of k row-wide fp32 atomic adds per fine row (r02b: interp_bwd 424 us per launch at 0.85 TB/s).
__global__ void interp_inv_pack_kernel(const int64_t* __restrict__ nbr, const float* __restrict__ weight, int64_t n, int k,
                                       int64_t M, int64_t* __restrict__ key, int32_t* __restrict__ counts,
                                       float* __restrict__ den) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = 0.0f;
  bool live = true;
  for (int s = 0; s < k; ++s) {
    const int64_t m = nbr[i * k + s];
    live = live && m >= 0 && m < M;             // (the slots behind the first -1 are unused, as in the forward pass)
    key[i * k + s] = live ? m : M;
    if (live) {
      d += weight[i * k + s];
      atomicAdd(&counts[m], 1);
    }
  }
  den[i] = d;
}

// entry j of the sorted order = flat slot e = i * k + s: the fine row and its weight (slots of one coarse row arrive in ascending e,
// i.e. ascending fine row: the stable sort's order)
__global__ void interp_inv_fill_kernel(const int32_t* __restrict__ order, const float* __restrict__ weight,
                                       const int32_t* __restrict__ inv_ptr, int64_t M, int k, int64_t slots,
                                       int32_t* __restrict__ inv_src, float* __restrict__ inv_w) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= slots || j >= inv_ptr[M]) return;
  const int32_t e = order[j];
  inv_src[j] = (int32_t)((uint32_t)e / (uint32_t)k);
  inv_w[j] = weight[e];
}

// owner of every row of a grouped row list: owner[r] = p for grp_ptr[p] <= r < grp_ptr[p + 1] (groups of at most a few dozen rows)
__global__ void group_owner_kernel(const int32_t* __restrict__ grp_ptr, int64_t N, int64_t E, int32_t* __restrict__ owner) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const int32_t lo = grp_ptr[p], hi = grp_ptr[p + 1];
  for (int32_t r = lo; r < hi && r < E; ++r) owner[r] = (int32_t)p;
}

// one coarse row per wave.  The list entries (fine row, weight, weight sum) are loaded once, one per lane, and handed round
// with v_readlane, so the dY row reads of successive entries do not wait on the list; VEC: 4 channels per lane.
template <bool VEC>
__global__ __launch_bounds__(256) void interp_bwd_gather_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                const int32_t* __restrict__ inv_ptr,
                                                                const int32_t* __restrict__ inv_src,
                                                                const float* __restrict__ inv_w,
                                                                const float* __restrict__ den, int64_t M, int64_t C,
                                                                float* __restrict__ dx, int64_t lddx) {
  constexpr int W = VEC ? 4 : 1;
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t m = (int64_t)ccn_xcd_block() * 4 + ry;
  if (m >= M) return;
  const int32_t lo = inv_ptr[m], hi = inv_ptr[m + 1];
  for (int64_t c0 = 0; c0 < C; c0 += 64 * W) {
    const int64_t c = c0 + (int64_t)cx * W;
    float acc[W];
#pragma unroll
    for (int q = 0; q < W; ++q) acc[q] = 0.0f;
    for (int32_t base = lo; base < hi; base += 64) {
      const int cnt = hi - base < 64 ? hi - base : 64;
      int my_src = 0;
      float my_w = 0.0f, my_den = 1.0f;
      if (cx < cnt) {
        my_src = inv_src[base + cx];
        my_w = inv_w[base + cx];
        my_den = den[my_src];
      }
      using vec_t = typename std::conditional<VEC, float4, float>::type;
      auto fetch = [&](int e) -> vec_t {
        const int64_t i = __builtin_amdgcn_readlane(my_src, e);
        return c < C ? *reinterpret_cast<const vec_t*>(dy + i * lddy + c) : vec_t{};
      };
      auto add = [&](int e, const vec_t& v) {
        const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), e));
        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_den), e));
        const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
        for (int q = 0; q < W; ++q) acc[q] += (f[q] / d) * w;
      };
      int e = 0;
      for (; e + 4 <= cnt; e += 4) {          // four rows in flight
        const vec_t v0 = fetch(e), v1 = fetch(e + 1), v2 = fetch(e + 2), v3 = fetch(e + 3);
        add(e, v0);
        add(e + 1, v1);
        add(e + 2, v2);
        add(e + 3, v3);
      }
      for (; e < cnt; ++e) add(e, fetch(e));
    }
    if (c < C) {
      if (VEC) *reinterpret_cast<float4*>(dx + m * lddx + c) = make_float4(acc[0], acc[1 % W], acc[2 % W], acc[3 % W]);
      else dx[m * lddx + c] = acc[0];
    }
  }
}

// ---- dataset-side curve splitter (kitti_dataset.py:73-92, nuscenes_dataset.py:101-118) ----
// split[i] (i >= 1): beam change, or fp64 |p_i - p_{i-1}| > (double)(thresh * sqrtf(|p_i.xy|)) with the right-hand
// side in fp32.  torch's CPU norms are fma chains: sqrt(fma(z,z,fma(y,y,x*x))) -- reproduced literally (this file is
// compiled with -ffp-contract=off, so only the explicit fma calls fuse).
__global__ void curve_split_flags_kernel(const float* __restrict__ pos, const int64_t* __restrict__ beam, int64_t n,
                                         float thresh, int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0) {
    flag[0] = 0;
    return;
  }
  const float x1 = pos[3 * i], y1 = pos[3 * i + 1], z1 = pos[3 * i + 2];
  const double dx = (double)x1 - (double)pos[3 * i - 3], dy = (double)y1 - (double)pos[3 * i - 2],
               dz = (double)z1 - (double)pos[3 * i - 1];
  const double edge = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
  const float radius = sqrtf(fmaf(y1, y1, x1 * x1));
  const float rhs = thresh * sqrtf(radius);
  int f = edge > (double)rhs;
  if (beam != nullptr) f |= (beam[i] != beam[i - 1]);
  flag[i] = f;
}

__global__ void curve_split_widen_kernel(const int32_t* __restrict__ run, int64_t n, int64_t* __restrict__ curve,
                                         int64_t* __restrict__ num_curves) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  curve[i] = run[i];
  if (i == n - 1) num_curves[0] = (int64_t)run[i] + 1;
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

size_t ccn_exclusive_scan_workspace_bytes(int64_t n) { return ccn_scan_scratch_bytes(n + 1) + 256; }

namespace {
__global__ void widen_total_kernel(const int32_t* __restrict__ offsets, int64_t n, int64_t* __restrict__ total64) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *total64 = offsets[n];
}
__global__ void zero_tail_kernel(int32_t* p) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *p = 0;
}
}  // namespace

int ccn_exclusive_scan_i32(const int32_t* counts, int64_t n, int32_t* offsets, int64_t* total64, void* ws,
                           size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(n >= 0 && offsets != nullptr, "exclusive_scan: bad arguments");
  CCN_REQUIRE(ws_bytes >= ccn_exclusive_scan_workspace_bytes(n), "exclusive_scan: workspace too small");
  // offsets[n] = total: scan n elements exclusively, total lands in offsets[n]
  if (n == 0) {
    hipLaunchKernelGGL(zero_tail_kernel, dim3(1), dim3(64), 0, s, offsets);
  } else {
    int rc = ccn_scan_i32(counts, offsets, n, false, offsets + n, ws, s);
    if (rc) return rc;
  }
  if (total64) hipLaunchKernelGGL(widen_total_kernel, dim3(1), dim3(64), 0, s, offsets, n, total64);
  CCN_LAUNCH_OK("exclusive_scan");
  return CCN_OK;
}

size_t ccn_segment_ptr_workspace_bytes(int64_t n) {
  return 2 * ccn_align256((size_t)(n + 1) * 4) + ccn_scan_scratch_bytes(n) + 256;
}

int ccn_segment_ptr(const int64_t* ids, int64_t n, int64_t* starts, int32_t* run_of, int64_t* meta, void* ws,
                    size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(n >= 0 && starts && meta, "segment_ptr: bad arguments");
  CCN_REQUIRE(ws_bytes >= ccn_segment_ptr_workspace_bytes(n), "segment_ptr: workspace too small");
  CCN_HIP(hipMemsetAsync(meta, 0, 2 * sizeof(int64_t), s), "segment_ptr");
  CCN_HIP(hipMemsetAsync(starts, 0, sizeof(int64_t), s), "segment_ptr");
  if (n == 0) return CCN_OK;
  CcnArena a(ws, ws_bytes);
  int32_t* flag = a.take<int32_t>(n + 1);
  int32_t* rank = a.take<int32_t>(n + 1);
  void* scratch = a.take<char>(ccn_scan_scratch_bytes(n));
  CCN_REQUIRE(a.ok(), "segment_ptr: workspace carve failed");
  const int nb = ccn_blocks(n, TPB);
  hipLaunchKernelGGL(run_flags_kernel, dim3(nb), dim3(TPB), 0, s, ids, n, flag, (unsigned long long*)meta);
  int rc = ccn_scan_i32(flag, rank, n, true, nullptr, scratch, s);
  if (rc) return rc;
  hipLaunchKernelGGL(run_starts_kernel, dim3(nb), dim3(TPB), 0, s, flag, rank, n, starts, run_of, meta);
  CCN_LAUNCH_OK("segment_ptr");
  return CCN_OK;
}

size_t ccn_curve_topology_workspace_bytes(int64_t n, int64_t num_clouds) {
  return 2 * ccn_align256((size_t)(n + 1) * 4) + ccn_align256((size_t)(num_clouds + 1) * 8) +
         ccn_scan_scratch_bytes(n) + 512;
}

int ccn_curve_topology(const int64_t* batch, const int64_t* p2c, int64_t n, int64_t num_clouds, int64_t* glob,
                       int32_t* cid, int32_t* curve_ptr, int64_t* cloud_ptr, int64_t* meta, void* ws,
                       size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(n > 0 && num_clouds > 0, "curve_topology: empty input (n=%lld, clouds=%lld)", (long long)n,
              (long long)num_clouds);
  CCN_REQUIRE(batch && p2c && glob && cid && curve_ptr && cloud_ptr && meta, "curve_topology: null pointer");
  CCN_REQUIRE(n < 2147483647LL, "curve_topology: more than 2^31-1 points");
  CCN_REQUIRE(ws_bytes >= ccn_curve_topology_workspace_bytes(n, num_clouds), "curve_topology: workspace too small");
  CcnArena a(ws, ws_bytes);
  int32_t* flag = a.take<int32_t>(n + 1);
  int32_t* rank = a.take<int32_t>(n + 1);
  int64_t* curve_off = a.take<int64_t>(num_clouds + 1);
  void* scratch = a.take<char>(ccn_scan_scratch_bytes(n));
  CCN_REQUIRE(a.ok(), "curve_topology: workspace carve failed");
  CCN_HIP(hipMemsetAsync(meta, 0, 4 * sizeof(int64_t), s), "curve_topology");
  CCN_HIP(hipMemsetAsync(cloud_ptr, 0, (num_clouds + 1) * sizeof(int64_t), s), "curve_topology");
  const int nb = ccn_blocks(n, TPB);
  hipLaunchKernelGGL(cloud_ptr_kernel, dim3(nb), dim3(TPB), 0, s, batch, n, num_clouds, cloud_ptr,
                     (unsigned long long*)meta);
  hipLaunchKernelGGL(cloud_offsets_kernel, dim3(1), dim3(64), 0, s, p2c, cloud_ptr, num_clouds, curve_off, meta);
  hipLaunchKernelGGL(glob_flags_kernel, dim3(nb), dim3(TPB), 0, s, batch, p2c, curve_off, n, num_clouds, glob, flag,
                     (unsigned long long*)meta);
  int rc = ccn_scan_i32(flag, rank, n, true, nullptr, scratch, s);
  if (rc) return rc;
  hipLaunchKernelGGL(curve_ptr_kernel, dim3(nb), dim3(TPB), 0, s, flag, rank, n, cid, curve_ptr, meta);
  CCN_LAUNCH_OK("curve_topology");
  return CCN_OK;
}

int ccn_diff_concat_fwd(const float* x, int64_t ldx, const int32_t* cid, int64_t n, int64_t C, float* out,
                        int64_t ldo, void* stream) {
  CCN_REQUIRE(x && cid && out && ldx >= C && ldo >= 2 * C, "diff_concat_fwd: bad arguments");
  if (n * C == 0) return CCN_OK;
  hipLaunchKernelGGL(diff_concat_fwd_kernel, dim3(ccn_blocks(n * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, x, ldx,
                     cid, n, C, out, ldo);
  CCN_LAUNCH_OK("diff_concat_fwd");
  return CCN_OK;
}

int ccn_diff_concat_bwd(const float* x, int64_t ldx, const int32_t* cid, int64_t n, int64_t C, const float* g,
                        int64_t ldg, float* dx, int64_t lddx, void* stream) {
  CCN_REQUIRE(x && cid && g && dx && ldx >= C && ldg >= 2 * C && lddx >= C, "diff_concat_bwd: bad arguments");
  if (n * C == 0) return CCN_OK;
  hipLaunchKernelGGL(diff_concat_bwd_kernel, dim3(ccn_blocks(n * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, x, ldx,
                     cid, n, C, g, ldg, dx, lddx);
  CCN_LAUNCH_OK("diff_concat_bwd");
  return CCN_OK;
}

static int im2col_fwd_impl(const float* x, int64_t ldx, const int32_t* seg, int64_t rows, int64_t C, int64_t taps, void* col,
                           int64_t ldcol, int zt, void* stream) {
  CCN_REQUIRE(x && col && ldx >= C && taps >= 1 && (taps & 1) && taps < 64 && C > 0 && C < (1 << 24) &&
                  ldcol >= taps * C,
              "im2col_fwd: bad arguments");
  if (rows == 0) return CCN_OK;
#define CCN_IM2COL(ZT_)                                                                                                   \
  do {                                                                                                                    \
    if (C >= 64)                                                                                                          \
      hipLaunchKernelGGL(im2col_fwd_rows_kernel<ZT_>, dim3(ccn_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, \
                         seg, rows, (int)C, (int)taps, col, ldcol);                                                       \
    else                                                                                                                  \
      hipLaunchKernelGGL(im2col_fwd_kernel<ZT_>, dim3(ccn_blocks(rows * taps * C, 256)), dim3(256), 0, (hipStream_t)stream, \
                         x, ldx, seg, rows, (int)C, (int)taps, col, ldcol);                                               \
  } while (0)
  if (zt == 0) CCN_IM2COL(0);
  else if (zt == 1) CCN_IM2COL(1);
  else CCN_IM2COL(2);
#undef CCN_IM2COL
  CCN_LAUNCH_OK("im2col_fwd");
  return CCN_OK;
}

int ccn_im2col_fwd(const float* x, int64_t ldx, const int32_t* seg, int64_t rows, int64_t C, int64_t taps, float* col,
                   int64_t ldcol, void* stream) {
  return im2col_fwd_impl(x, ldx, seg, rows, C, taps, col, ldcol, 0, stream);
}

// ... the shifted-row matrix as 16-bit rows (bf16, fp16 when f16 != 0; (taps * C) % 8 == 0, ldcol in 16-bit elements)
int ccn_im2col_fwd_h(const float* x, int64_t ldx, const int32_t* seg, int64_t rows, int64_t C, int64_t taps, void* col,
                     int64_t ldcol, int f16, void* stream) {
  CCN_REQUIRE((taps * C) % 8 == 0 && ldcol % 8 == 0 && ((uintptr_t)col & 15) == 0,
              "im2col_fwd_h: rows of (taps * C) % 8 == 0 elements, 16-byte aligned");
  return im2col_fwd_impl(x, ldx, seg, rows, C, taps, col, ldcol, f16 ? 2 : 1, stream);
}

int ccn_im2col_bwd(const float* dcol, int64_t ldcol, const int32_t* seg, int64_t rows, int64_t C, int64_t taps,
                   float* dx, int64_t lddx, void* stream) {
  CCN_REQUIRE(dcol && dx && lddx >= C && taps >= 1 && (taps & 1) && taps < 64 && C > 0 && C < (1 << 24) &&
                  ldcol >= taps * C,
              "im2col_bwd: bad arguments");
  if (rows == 0) return CCN_OK;
  hipLaunchKernelGGL(im2col_bwd_kernel<0>, dim3(ccn_blocks(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, dcol, ldcol, seg,
                     rows, (int)C, (int)taps, dx, lddx);
  CCN_LAUNCH_OK("im2col_bwd");
  return CCN_OK;
}

// ... dcol as bf16 rows (ldcol in 16-bit elements)
int ccn_im2col_bwd_h(const void* dcol, int64_t ldcol, const int32_t* seg, int64_t rows, int64_t C, int64_t taps, float* dx,
                     int64_t lddx, void* stream) {
  CCN_REQUIRE(dcol && dx && lddx >= C && taps >= 1 && (taps & 1) && taps < 64 && C > 0 && C < (1 << 24) &&
                  ldcol >= taps * C,
              "im2col_bwd_h: bad arguments");
  if (rows == 0) return CCN_OK;
  hipLaunchKernelGGL(im2col_bwd_kernel<1>, dim3(ccn_blocks(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, dcol, ldcol, seg,
                     rows, (int)C, (int)taps, dx, lddx);
  CCN_LAUNCH_OK("im2col_bwd_h");
  return CCN_OK;
}

int ccn_shift_add_fwd(const float* P, int64_t ldp, const float* bias, int64_t rows, int64_t Co, int64_t taps, float* Y,
                      int64_t ldy, void* stream) {
  CCN_REQUIRE(P && Y && rows >= 0 && Co > 0 && Co < (1 << 20) && taps >= 1 && (taps & 1) && taps < 64 &&
                  ldp >= taps * Co && ldy >= Co,
              "shift_add_fwd: bad arguments");
  if (rows == 0) return CCN_OK;
  hipLaunchKernelGGL(shift_add_fwd_kernel, dim3(ccn_blocks(rows * Co, 256)), dim3(256), 0, (hipStream_t)stream, P, ldp,
                     bias, rows, (int)Co, (int)taps, Y, ldy);
  CCN_LAUNCH_OK("shift_add_fwd");
  return CCN_OK;
}

int ccn_shift_add_bwd(const float* dY, int64_t lddy, int64_t rows, int64_t Co, int64_t taps, float* dP, int64_t lddp,
                      void* stream) {
  CCN_REQUIRE(dY && dP && rows >= 0 && Co > 0 && Co < (1 << 20) && taps >= 1 && (taps & 1) && taps < 64 &&
                  lddp >= taps * Co && lddy >= Co,
              "shift_add_bwd: bad arguments");
  if (rows == 0) return CCN_OK;
  hipLaunchKernelGGL(shift_add_bwd_kernel, dim3(ccn_blocks(rows * taps * Co, 256)), dim3(256), 0, (hipStream_t)stream, dY,
                     lddy, rows, (int)Co, (int)taps, dP, lddp);
  CCN_LAUNCH_OK("shift_add_bwd");
  return CCN_OK;
}

int ccn_gather_rows(const float* src, int64_t lds_, const int64_t* index, int64_t m, int64_t C, float* dst,
                    int64_t ldd, void* stream) {
  CCN_REQUIRE(src && index && dst && lds_ >= C && ldd >= C, "gather_rows: bad arguments");
  if (m * C == 0) return CCN_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ccn_blocks(m * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, src, lds_,
                     index, m, C, dst, ldd);
  CCN_LAUNCH_OK("gather_rows");
  return CCN_OK;
}

int ccn_scatter_rows(const float* src, int64_t lds_, const int64_t* index, int64_t m, int64_t C, float* dst,
                     int64_t ldd, int accumulate, void* stream) {
  CCN_REQUIRE(src && index && dst && lds_ >= C && ldd >= C, "scatter_rows: bad arguments");
  if (m * C == 0) return CCN_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(ccn_blocks(m * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, src, lds_,
                     index, m, C, dst, ldd, accumulate);
  CCN_LAUNCH_OK("scatter_rows");
  return CCN_OK;
}

int ccn_scatter_rows_fill(const float* src, int64_t lds_, const int64_t* index, int64_t m, int64_t C, float* dst, int64_t ldd,
                          int64_t total_rows, int64_t row_offset, void* stream) {
  CCN_REQUIRE(src && index && dst && lds_ >= C && ldd >= C && m > 0 && total_rows >= m + row_offset && row_offset >= 0,
              "scatter_rows_fill: bad arguments");
  hipLaunchKernelGGL(scatter_rows_fill_kernel, dim3(ccn_blocks(m, 4)), dim3(256), 0, (hipStream_t)stream, src, lds_, index, m, C,
                     dst, ldd, total_rows, row_offset);
  CCN_LAUNCH_OK("scatter_rows_fill");
  return CCN_OK;
}

size_t ccn_curve_fps_workspace_bytes(int64_t n) {
  return 2 * ccn_align256((size_t)(n + 1) * 8) + 2 * ccn_align256((size_t)(n + 1) * 4) + ccn_scan_scratch_bytes(n) +
         512;
}

int ccn_curve_fps(const float* pos, const int32_t* cid, const int32_t* curve_ptr, int64_t n, float spacing, float u,
                  int64_t* idx_out, int64_t* count_out, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(pos && cid && curve_ptr && idx_out && count_out && n > 0, "curve_fps: bad arguments");
  CCN_REQUIRE(spacing > 0.0f, "curve_fps: arclength spacing must be positive");
  CCN_REQUIRE(ws_bytes >= ccn_curve_fps_workspace_bytes(n), "curve_fps: workspace too small");
  CcnArena a(ws, ws_bytes);
  double* step = a.take<double>(n + 1);
  double* run = a.take<double>(n + 1);
  int32_t* keep = a.take<int32_t>(n + 1);
  int32_t* rank = a.take<int32_t>(n + 1);
  void* scratch = a.take<char>(ccn_scan_scratch_bytes(n));
  CCN_REQUIRE(a.ok(), "curve_fps: workspace carve failed");
  const int nb = ccn_blocks(n, TPB);
  hipLaunchKernelGGL(fps_steps_kernel, dim3(nb), dim3(TPB), 0, s, pos, cid, n, step);
  int rc = ccn_scan_f64(step, run, n, true, scratch, s);
  if (rc) return rc;
  hipLaunchKernelGGL(fps_keep_kernel, dim3(nb), dim3(TPB), 0, s, run, cid, curve_ptr, n, spacing, u, keep);
  rc = ccn_scan_i32(keep, rank, n, false, nullptr, scratch, s);
  if (rc) return rc;
  hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(TPB), 0, s, keep, rank, n, idx_out, count_out);
  CCN_LAUNCH_OK("curve_fps");
  return CCN_OK;
}

size_t ccn_curve_group_subset_workspace_bytes(int64_t n, int64_t Q, int64_t M) {
  (void)n;
  (void)Q;
  return ccn_align256((size_t)(M + 1) * 4) + ccn_scan_scratch_bytes(M + 1) + 1024;
}

int ccn_curve_group_subset_count(const float* pos, const int32_t* cid, const int32_t* curve_ptr, const int64_t* p2c,
                                 int64_t n, int64_t Q, const int64_t* idx, int64_t M, float radius, float* budget,
                                 int32_t* offsets, int64_t* total, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(pos && cid && curve_ptr && p2c && idx && budget && offsets && total, "group_subset_count: null pointer");
  CCN_REQUIRE(n > 0 && Q > 0 && M > 0, "group_subset_count: empty input");
  CCN_REQUIRE(ws_bytes >= ccn_curve_group_subset_workspace_bytes(n, Q, M), "group_subset_count: workspace too small");
  CcnArena a(ws, ws_bytes);
  int32_t* counts = a.take<int32_t>(M + 1);
  unsigned int* maxbits = a.take<unsigned int>(2);
  void* scratch = a.take<char>(ccn_scan_scratch_bytes(M + 1));
  CCN_REQUIRE(a.ok(), "group_subset_count: workspace carve failed");
  CCN_HIP(hipMemsetAsync(maxbits, 0, 2 * sizeof(unsigned int), s), "group_subset_count");
  hipLaunchKernelGGL(curve_budget_kernel, dim3(ccn_blocks(Q, TPB)), dim3(TPB), 0, s, pos, curve_ptr, Q, radius, budget,
                     maxbits);
  hipLaunchKernelGGL(curve_reach_kernel, dim3(1), dim3(64), 0, s, maxbits, budget, Q);
  hipLaunchKernelGGL(group_subset_kernel<false>, dim3(ccn_blocks(M, TPB)), dim3(TPB), 0, s, cid, curve_ptr, p2c, Q, idx,
                     M, budget, counts, (const int32_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0);
  int rc = ccn_scan_i32(counts, offsets, M, false, offsets + M, scratch, s);
  if (rc) return rc;
  hipLaunchKernelGGL(widen_total_kernel, dim3(1), dim3(64), 0, s, offsets, M, total);
  CCN_LAUNCH_OK("group_subset_count");
  return CCN_OK;
}

int ccn_curve_group_subset_fill(const int32_t* cid, const int32_t* curve_ptr, const int64_t* p2c, int64_t n,
                                int64_t Q, const int64_t* idx, int64_t M, const float* budget,
                                const int32_t* offsets, int64_t* row, int64_t* col, void* stream) {
  return ccn_curve_group_subset_fill_cap(cid, curve_ptr, p2c, n, Q, idx, M, budget, offsets, row, col, INT64_MAX, stream);
}

int ccn_curve_group_subset_fill_cap(const int32_t* cid, const int32_t* curve_ptr, const int64_t* p2c, int64_t n,
                                    int64_t Q, const int64_t* idx, int64_t M, const float* budget,
                                    const int32_t* offsets, int64_t* row, int64_t* col, int64_t cap, void* stream) {
  (void)n;
  CCN_REQUIRE(cid && curve_ptr && p2c && idx && budget && offsets && row && col && cap >= 0, "group_subset_fill: bad arguments");
  if (M == 0) return CCN_OK;
  hipLaunchKernelGGL(group_subset_kernel<true>, dim3(ccn_blocks(M, TPB)), dim3(TPB), 0, (hipStream_t)stream, cid,
                     curve_ptr, p2c, Q, idx, M, budget, (int32_t*)nullptr, offsets, row, col, cap);
  CCN_LAUNCH_OK("group_subset_fill");
  return CCN_OK;
}

size_t ccn_curve_group_superset_workspace_bytes(int64_t n) {
  return 2 * ccn_align256((size_t)(n + 1) * 4) + ccn_scan_scratch_bytes(n) + 512;
}

int ccn_curve_group_superset(const float* pos, const int32_t* cid, int64_t n, const int64_t* idx, int64_t M, int64_t k,
                             int64_t* nbr, float* weight, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(pos && cid && idx && nbr && weight && n > 0 && M > 0, "group_superset: bad arguments");
  CCN_REQUIRE(k >= 1 && k <= SUP_MAXK, "group_superset: k must be in [1, %d]", SUP_MAXK);
  CCN_REQUIRE(ws_bytes >= ccn_curve_group_superset_workspace_bytes(n), "group_superset: workspace too small");
  CcnArena a(ws, ws_bytes);
  int32_t* taken = a.take<int32_t>(n + 1);
  int32_t* upto = a.take<int32_t>(n + 1);
  void* scratch = a.take<char>(ccn_scan_scratch_bytes(n));
  CCN_REQUIRE(a.ok(), "group_superset: workspace carve failed");
  CCN_HIP(hipMemsetAsync(taken, 0, (size_t)n * 4, s), "group_superset");
  hipLaunchKernelGGL(mark_samples_kernel, dim3(ccn_blocks(M, TPB)), dim3(TPB), 0, s, idx, M, taken);
  int rc = ccn_scan_i32(taken, upto, n, true, nullptr, scratch, s);
  if (rc) return rc;
  hipLaunchKernelGGL(group_superset_kernel, dim3(ccn_blocks(n, TPB)), dim3(TPB), 0, s, pos, cid, n, idx, M, (int)k,
                     upto, nbr, weight);
  CCN_LAUNCH_OK("group_superset");
  return CCN_OK;
}

int ccn_interp_fwd(const float* x, int64_t ldx, const int64_t* nbr, const float* weight, int64_t n, int64_t k,
                   int64_t C, float* y, int64_t ldy, void* stream) {
  CCN_REQUIRE(x && nbr && weight && y && ldx >= C && ldy >= C && k >= 1, "interp_fwd: bad arguments");
  if (n * C == 0) return CCN_OK;
  hipLaunchKernelGGL(interp_fwd_kernel, dim3(ccn_blocks(n * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, nbr,
                     weight, n, (int)k, C, y, ldy);
  CCN_LAUNCH_OK("interp_fwd");
  return CCN_OK;
}

int ccn_interp_bwd(const float* dy, int64_t lddy, const int64_t* nbr, const float* weight, int64_t n, int64_t k,
                   int64_t C, float* dx, int64_t lddx, void* stream) {
  CCN_REQUIRE(dy && nbr && weight && dx && lddy >= C && lddx >= C && k >= 1, "interp_bwd: bad arguments");
  if (n * C == 0) return CCN_OK;
  hipLaunchKernelGGL(interp_bwd_kernel, dim3(ccn_blocks(n * C, TPB)), dim3(TPB), 0, (hipStream_t)stream, dy, lddy, nbr,
                     weight, n, (int)k, C, dx, lddx);
  CCN_LAUNCH_OK("interp_bwd");
  return CCN_OK;
}

size_t ccn_interp_inverse_workspace_bytes(int64_t n, int64_t k, int64_t M) {
  const int64_t slots = n * k > 0 ? n * k : 1;
  return ccn_align256((size_t)slots * 8) + ccn_align256((size_t)(M + 1) * 4) + ccn_align256(ccn_scan_scratch_bytes(M + 1)) +
         ccn_rank_keys_workspace_bytes(slots) + 1024;
}

int ccn_interp_inverse(const int64_t* nbr, const float* weight, int64_t n, int64_t k, int64_t M, int32_t* inv_ptr,
                       int32_t* inv_src, float* inv_w, float* den, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(nbr && weight && inv_ptr && inv_src && inv_w && den && n >= 0 && k >= 1 && M > 0, "interp_inverse: bad arguments");
  CCN_REQUIRE(n * k < (int64_t)1 << 31 && M < (int64_t)1 << 31, "interp_inverse: more than 2^31 entries");
  CCN_REQUIRE(ws_bytes >= ccn_interp_inverse_workspace_bytes(n, k, M), "interp_inverse: workspace too small");
  const int64_t slots = n * k;
  CcnArena a(ws, ws_bytes);
  int64_t* key = a.take<int64_t>(slots > 0 ? slots : 1);
  int32_t* counts = a.take<int32_t>(M + 1);
  void* scratch = a.take<char>(ccn_scan_scratch_bytes(M + 1));
  const size_t sort_bytes = ccn_rank_keys_workspace_bytes(slots > 0 ? slots : 1);
  void* sort_ws = a.take<char>(sort_bytes);
  CCN_REQUIRE(a.ok(), "interp_inverse: workspace carve failed");
  CCN_HIP(hipMemsetAsync(counts, 0, (size_t)(M + 1) * 4, s), "interp_inverse");
  if (n > 0)
    hipLaunchKernelGGL(interp_inv_pack_kernel, dim3(ccn_blocks(n, TPB)), dim3(TPB), 0, s, nbr, weight, n, (int)k, M, key, counts, den);
  int rc = ccn_scan_i32(counts, inv_ptr, M + 1, false, nullptr, scratch, s);       // inv_ptr[M] = total
  if (rc) return rc;
  if (slots > 0) {
    // the lists by a stable sort of the slots by coarse row (round 5; rounds 2-4: fill at an atomic cursor + one thread's
    // insertion sort per list -- quadratic in the list length, minutes for a coarse row that thousands of fine rows read)
    int mask = 0;
    for (int b = 0; b < 4; ++b)
      if (b == 0 || ((uint64_t)M >> (8 * b)) != 0) mask |= 1 << b;
    const int32_t* order;
    rc = ccn_sort_payload(key, slots, mask, sort_ws, sort_bytes, s, &order);
    if (rc) return rc;
    hipLaunchKernelGGL(interp_inv_fill_kernel, dim3(ccn_blocks(slots, TPB)), dim3(TPB), 0, s, order, weight, inv_ptr, M, (int)k, slots,
                       inv_src, inv_w);
  }
  CCN_LAUNCH_OK("interp_inverse");
  return CCN_OK;
}

int ccn_group_owner(const int32_t* grp_ptr, int64_t N, int64_t E, int32_t* owner, void* stream) {
  CCN_REQUIRE(grp_ptr && owner && N > 0 && E >= 0 && E < (int64_t)1 << 31, "group_owner: bad arguments");
  hipLaunchKernelGGL(group_owner_kernel, dim3(ccn_blocks(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, grp_ptr, N, E, owner);
  CCN_LAUNCH_OK("group_owner");
  return CCN_OK;
}

int ccn_interp_bwd_gather(const float* dy, int64_t lddy, const int32_t* inv_ptr, const int32_t* inv_src, const float* inv_w,
                          const float* den, int64_t M, int64_t C, float* dx, int64_t lddx, void* stream) {
  CCN_REQUIRE(dy && inv_ptr && inv_src && inv_w && den && dx && lddy >= C && lddx >= C && M >= 0, "interp_bwd_gather: bad arguments");
  if (M * C == 0) return CCN_OK;
  const bool vec = C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(interp_bwd_gather_kernel<true>, dim3(ccn_blocks(M, 4)), dim3(256), 0, (hipStream_t)stream, dy, lddy,
                       inv_ptr, inv_src, inv_w, den, M, C, dx, lddx);
  else
    hipLaunchKernelGGL(interp_bwd_gather_kernel<false>, dim3(ccn_blocks(M, 4)), dim3(256), 0, (hipStream_t)stream, dy, lddy,
                       inv_ptr, inv_src, inv_w, den, M, C, dx, lddx);
  CCN_LAUNCH_OK("interp_bwd_gather");
  return CCN_OK;
}

size_t ccn_curve_split_workspace_bytes(int64_t n) {
  return 2 * ccn_align256((size_t)(n + 1) * 4) + ccn_scan_scratch_bytes(n) + 512;
}

int ccn_curve_split(const float* pos, const int64_t* beam, int64_t n, float thresh, int64_t* curve_idx,
                    int64_t* num_curves, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(pos && curve_idx && num_curves && n > 0, "curve_split: bad arguments");
  CCN_REQUIRE(n < (int64_t)1 << 31, "curve_split: more than 2^31 points");
  CCN_REQUIRE(ws_bytes >= ccn_curve_split_workspace_bytes(n), "curve_split: workspace too small");
  CcnArena a(ws, ws_bytes);
  int32_t* flag = a.take<int32_t>(n + 1);
  int32_t* run = a.take<int32_t>(n + 1);
  void* scratch = a.take<char>(ccn_scan_scratch_bytes(n));
  CCN_REQUIRE(a.ok(), "curve_split: workspace carve failed");
  const int nb = ccn_blocks(n, TPB);
  hipLaunchKernelGGL(curve_split_flags_kernel, dim3(nb), dim3(TPB), 0, s, pos, beam, n, thresh, flag);
  int rc = ccn_scan_i32(flag, run, n, true, nullptr, scratch, s);
  if (rc) return rc;
  hipLaunchKernelGGL(curve_split_widen_kernel, dim3(nb), dim3(TPB), 0, s, run, n, curve_idx, num_curves);
  CCN_LAUNCH_OK("curve_split");
  return CCN_OK;
}

}  // extern "C"

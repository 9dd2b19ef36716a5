// Y = A W^T in fp32-grade arithmetic on the bf16 matrix cores ("bf16x3" MLP mode).
//
// Every fp32 operand is split exactly into three bf16 terms, x = h + m + l (h = rn(x), m = rn(x - h), l = x - h - m:
// 3 x 8 significand bits cover the 24 of an fp32, and bf16 has fp32's exponent range), and the product is assembled from
// the six partial products whose weight is at least 2^-16 of the leading one,
//     a b  ~=  a_l b_h + a_h b_l + a_m b_m + a_m b_h + a_h b_m + a_h b_h          (dropped: a_m b_l, a_l b_m, a_l b_l),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  The dropped terms are below 2^-23 |a b| per
// product -- the size of ONE fp32 rounding, and smaller than the rounding error the fp32 accumulation itself makes over a
// contraction -- so results agree with the fp32 MFMA kernels to fp32 accuracy (tests/test_gpu_gemm_x3.py measures both
// against an fp64 product).  Cost: six bf16 MFMAs (32 cycles each) per 16-deep step of a 32 x 32 tile instead of eight
// fp32 MFMAs (64 cycles each): 2.7x fewer matrix-core cycles, at a fraction of the energy per flop.
// Not IEEE in the corners: an infinite or NaN operand gives NaN (inf - inf in the split), and |x| > 3.39e38 overflows.
//
// Data path.  The weight operand (N x K, KBs..MBs) is split ONCE per call by a small kernel into three zero-padded bf16
// images in caller-owned scratch; the row operand A (M x K fp32 in HBM) is read once per N tile, split in registers on its
// way into LDS.  Workgroup = 4 waves on a 128 x BN tile, K slices of 32; LDS holds one slice of the three A and three
// B images (80-byte rows: a lane's MFMA fragment is ONE ds_read_b128, conflict-free) = 60 KB at BN = 128, so two
// workgroups share a CU and one computes while the other converts; the next slice's global loads are in flight during the
// MFMAs.  Workgroup ids are mapped so that the N tiles of one row block run on the SAME XCD (ids w and w + 8), i.e. A is
// read from HBM once and from that XCD's L2 afterwards.
#include "ccn_common.h"

#include <atomic>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

constexpr int XK = 32;        // K slice
constexpr int XLD = XK + 8;   // bf16 elements per LDS row (80 bytes)
constexpr int X_TPB = 256;
constexpr int X_BM = 128;
constexpr int X_NPAD = 128;   // weight images are padded to a multiple of this many rows (every BN divides it)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// W (N x K fp32, row stride ldw) -> img[s][Np][Kp] bf16, s = 0 (h), 1 (m), 2 (l); rows >= N and columns >= K are zero
__global__ __launch_bounds__(256) void x3_split_weights_kernel(const float* __restrict__ W, int64_t ldw, int64_t N,
                                                               int64_t K, int64_t Np, int64_t Kp,
                                                               __bf16* __restrict__ img) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= Np * Kp) return;
  const int64_t n = e / Kp, k = e - n * Kp;
  const float x = (n < N && k < K) ? W[n * ldw + k] : 0.f;
  __bf16 h, m, l;
  split3(x, h, m, l);
  img[e] = h;
  img[Np * Kp + e] = m;
  img[2 * Np * Kp + e] = l;
}

template <int BN, int WMW>
__global__ __launch_bounds__(X_TPB, 2) void gemm_x3_kernel(const float* __restrict__ A, int64_t lda,
                                                           const __bf16* __restrict__ Wimg, int64_t Np, int64_t Kp,
                                                           const float* __restrict__ bias, float* __restrict__ C,
                                                           int64_t ldc, int64_t M, int64_t N, int64_t K, int64_t gm,
                                                           int gn, double* __restrict__ colstats) {
  constexpr int WNW = 4 / WMW;               // waves along N
  constexpr int AB = X_BM / (32 * WMW);      // 32-row blocks per wave
  constexpr int NT = BN / (32 * WNW);        // 32-column blocks per wave
  constexpr int A_IMG = X_BM * XLD, B_IMG = BN * XLD;   // bf16 elements per LDS image
  constexpr int PT_A = X_BM * XK / 4 / X_TPB;           // float4 per thread and slice (= 4)
  constexpr int B_CHUNKS = 3 * BN * 4;                  // 16-byte chunks of the three B images per slice
  constexpr int PT_B = (B_CHUNKS + X_TPB - 1) / X_TPB;
  constexpr bool B_EXACT = B_CHUNKS % X_TPB == 0;       // every thread has a chunk in every round
  static_assert(AB >= 1 && NT >= 1 && AB * WMW * 32 == X_BM && NT * WNW * 32 == BN, "tile shape");
  static_assert(3 * (A_IMG + B_IMG) * 2 >= WMW * BN * 2 * 8, "statistics scratch fits");
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * (A_IMG + B_IMG)];
  __bf16* const Al = lds;
  __bf16* const Bl = lds + 3 * A_IMG;

  // XCD-aware tile order: ids w, w + 8, ... (same XCD, dispatched back to back) walk the N tiles of one row block
  const int64_t w = blockIdx.x;
  const int64_t group = w / (8 * gn);
  const int within = (int)(w - group * (8 * gn));
  const int64_t mt = group * 8 + (within & 7);
  const int nt = within >> 3;
  if (mt >= gm) return;
  const int64_t m0 = mt * X_BM, n0 = (int64_t)nt * BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % WMW, wn = wave / WMW;
  const int i = lane & 31, h = lane >> 5;

  f32x16 acc[AB][NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t n = n0 + (wn * NT + t) * 32 + i;
    const float b = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int ab = 0; ab < AB; ++ab)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ab][t][r] = b;
  }

  // ---- loaders: A rows clamped into the matrix (rows >= M only feed outputs that are never stored)
  const float* pa[PT_A];
  int ka[PT_A];
#pragma unroll
  for (int it = 0; it < PT_A; ++it) {
    const int slot = threadIdx.x + it * X_TPB;
    const int r = slot >> 3, kq = slot & 7;
    int64_t row = m0 + r;
    row = row < M ? row : M - 1;
    pa[it] = A + row * lda + kq * 4;
    ka[it] = kq * 4;
  }
  const __bf16* pb[PT_B];
  int sb[PT_B];   // LDS element offset of the chunk (negative: no chunk for this thread)
#pragma unroll
  for (int it = 0; it < PT_B; ++it) {
    const int c = threadIdx.x + it * X_TPB;
    if (c < B_CHUNKS) {
      const int img = c / (BN * 4), rem = c - img * (BN * 4);
      const int r = rem >> 2, q = rem & 3;
      pb[it] = Wimg + ((int64_t)img * Np + n0 + r) * Kp + q * 8;
      sb[it] = img * B_IMG + r * XLD + q * 8;
    } else {
      pb[it] = Wimg;
      sb[it] = -1;
    }
  }
  float4 ra[PT_A];
  uint4 rb[PT_B];
#pragma unroll
  for (int it = 0; it < PT_B; ++it) rb[it] = make_uint4(0u, 0u, 0u, 0u);
  auto load_slice = [&](int64_t k0) {
    if (k0 + XK <= K) {
#pragma unroll
      for (int it = 0; it < PT_A; ++it) ra[it] = *reinterpret_cast<const float4*>(pa[it] + k0);
    } else {   // K tail: clamp the address into the row, zero what lies at k >= K
#pragma unroll
      for (int it = 0; it < PT_A; ++it) {
        const int64_t k = k0 + ka[it];
        const int64_t kc = k <= lda - 4 ? k : lda - 4;
        float4 v = *reinterpret_cast<const float4*>(pa[it] + (kc - ka[it]));
        v.x = k + 0 < K ? v.x : 0.f;
        v.y = k + 1 < K ? v.y : 0.f;
        v.z = k + 2 < K ? v.z : 0.f;
        v.w = k + 3 < K ? v.w : 0.f;
        ra[it] = v;
      }
    }
#pragma unroll
    for (int it = 0; it < PT_B; ++it)
      if (B_EXACT || sb[it] >= 0) rb[it] = *reinterpret_cast<const uint4*>(pb[it] + k0);
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int it = 0; it < PT_A; ++it) {
      const int slot = threadIdx.x + it * X_TPB;
      const int r = slot >> 3, kq = slot & 7;
      bf16x4 vh, vm, vl;
      const float xs[4] = {ra[it].x, ra[it].y, ra[it].z, ra[it].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 eh, em, el;
        split3(xs[e], eh, em, el);
        vh[e] = eh;
        vm[e] = em;
        vl[e] = el;
      }
      __bf16* dst = Al + r * XLD + kq * 4;
      *reinterpret_cast<bf16x4*>(dst) = vh;
      *reinterpret_cast<bf16x4*>(dst + A_IMG) = vm;
      *reinterpret_cast<bf16x4*>(dst + 2 * A_IMG) = vl;
    }
#pragma unroll
    for (int it = 0; it < PT_B; ++it)
      if (B_EXACT || sb[it] >= 0) *reinterpret_cast<uint4*>(Bl + sb[it]) = rb[it];
  };

  load_slice(0);
  store_slice();
  __syncthreads();
  for (int64_t k0 = 0; k0 < K; k0 += XK) {
    const bool has_next = k0 + XK < K;
    if (has_next) load_slice(k0 + XK);   // in flight during the MFMAs below
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bf16x8 a[AB][3], b[NT][3];
#pragma unroll
      for (int ab = 0; ab < AB; ++ab)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          a[ab][s] = *reinterpret_cast<const bf16x8*>(Al + s * A_IMG + ((wm * AB + ab) * 32 + i) * XLD + (2 * st + h) * 8);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          b[t][s] = *reinterpret_cast<const bf16x8*>(Bl + s * B_IMG + ((wn * NT + t) * 32 + i) * XLD + (2 * st + h) * 8);
      // smallest partial products first; consecutive MFMAs go to different accumulators
      constexpr int SA[6] = {2, 0, 1, 1, 0, 0};
      constexpr int SB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int ab = 0; ab < AB; ++ab)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[ab][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ab][SA[p]], b[t][SB[p]], acc[ab][t], 0, 0, 0);
    }
    if (has_next) {
      __syncthreads();   // every wave is done reading this slice
      store_slice();
      __syncthreads();
    }
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  double* stat_lds = reinterpret_cast<double*>(lds);   // [WMW][BN][2]
  if (colstats != nullptr) __syncthreads();            // LDS is reused
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col_local = (wn * NT + t) * 32 + i;
    const int64_t n = n0 + col_local;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int ab = 0; ab < AB; ++ab) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + (wm * AB + ab) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M && n < N) {
          const float v = acc[ab][t][r];
          C[m * ldc + n] = v;
          if (colstats != nullptr) {
            s1 += (double)v;
            s2 += (double)v * (double)v;
          }
        }
      }
    }
    if (colstats != nullptr) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        stat_lds[(wm * BN + col_local) * 2] = s1;
        stat_lds[(wm * BN + col_local) * 2 + 1] = s2;
      }
    }
  }
  if (colstats != nullptr) {
    __syncthreads();
    for (int c = threadIdx.x; c < BN; c += X_TPB) {
      const int64_t n = n0 + c;
      if (n < N) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int ww = 0; ww < WMW; ++ww) {
          s1 += stat_lds[(ww * BN + c) * 2];
          s2 += stat_lds[(ww * BN + c) * 2 + 1];
        }
        double* dst = colstats + mt * 2 * N;   // one partial row per 128-row block (deterministic)
        dst[n] = s1;
        dst[N + n] = s2;
      }
    }
  }
}


// ------------------------------------------------------------------ persistent LDS-DMA form (K % 32 == 0, many tiles)
// One 8-wave workgroup per CU walks 256 x BN output tiles; wave w owns the 32-row band w.  The K slices of consecutive
// tiles form one stream:
//   A  fp32, 256 rows x 128 B per slice, copied by global_load_lds into a 3-stage ring two slices ahead (HBM latency);
//      LDS chunk c (16 B) of row r holds global chunk c ^ ((r >> 1) & 7), so a lane's eight consecutive k (two 16-byte
//      reads at chunks 4 st + 2 h + {0, 1}) are conflict-free.  The split into three bf16 terms happens in registers,
//      between the MFMAs -- no bf16 image of A is ever written to LDS, and no second barrier per slice is needed;
//   W  the three pre-split bf16 images, BN rows x 64 B each per slice, copied one slice ahead (they come from L2) into a
//      2-stage ring; LDS chunk c of row r holds global chunk c ^ ((r >> 2) & 3): one ds_read_b128 per fragment,
//      conflict-free.
// vmcnt retires in issue order, so each iteration issues W(g+1) BEFORE A(g+2): waiting for slice g then leaves exactly
// the four A copies of slice g+1 in flight.  The epilogue (stores issued and forgotten, BatchNorm partial sums parked in
// an LDS table that is read out after the next tile's first barrier) is the one of gemm_glds_persistent_kernel.
static std::atomic<bool> g_use_persistent{true};   // A/B hook (ccn_gemm_x3_use_persistent)
constexpr int P_TPB = 512;
constexpr int P_BM = 256;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void split8(const f32x4& lo, const f32x4& hi, bf16x8& vh, bf16x8& vm, bf16x8& vl) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = e < 4 ? lo[e] : hi[e - 4];
    __bf16 eh, em, el;
    split3(x, eh, em, el);
    vh[e] = eh;
    vm[e] = em;
    vl[e] = el;
  }
}

template <int BN>
__global__ __launch_bounds__(P_TPB) void gemm_x3_persistent_kernel(const float* __restrict__ A, int64_t lda,
                                                                   const __bf16* __restrict__ Wimg, int64_t Np,
                                                                   int64_t Kp, const float* __restrict__ bias,
                                                                   float* __restrict__ C, int64_t ldc, int64_t M,
                                                                   int64_t N, int64_t K, int64_t gm, int gn, int xcd_map,
                                                                   double* __restrict__ colstats) {
  constexpr int NT = BN / 32;
  constexpr int A_ST = P_BM * XK;        // floats per A stage (32 KiB)
  constexpr int B_ST = 3 * BN * 64;      // bytes per W stage
  constexpr int NA = 4;                  // A copies per wave and slice (8 rows x 128 B each)
  constexpr int B_GROUPS = 3 * BN / 16;  // 1-KiB copies (16 rows x 64 B) per W slice
  constexpr int NB = (B_GROUPS + 7) / 8;
  __shared__ __attribute__((aligned(16))) float ldsA[3 * A_ST];
  __shared__ __attribute__((aligned(16))) unsigned char ldsB[2 * B_ST];
  __shared__ float stat_part[8 * BN * 2];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const uint32_t ldsA_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)ldsA;
  const uint32_t ldsB_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)ldsB;
  const uint32_t a_row = (uint32_t)((wave * 32 + i) * 128);
  const uint32_t a_swz = (uint32_t)((i >> 1) & 7);
  const uint32_t b_row = (uint32_t)(i * 64);
  const uint32_t b_swz = (uint32_t)((i >> 2) & 3);
  const int T = (int)(K / XK);

  // tile of step j for this workgroup.  xcd_map: the gn column tiles of one row block go to workgroups w, w + 8, ...
  // (the same XCD, the same step), so A is read from HBM once and from that XCD's L2 by the others
  const int rows_per_step = xcd_map ? (int)(gridDim.x / gn) : 0;
  auto tile_at = [&](int64_t j, int64_t& m, int& n) -> bool {
    if (xcd_map) {
      const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
      n = slot % gn;
      m = j * rows_per_step + (slot / gn) * 8 + xcd;
    } else {
      const int64_t tile = j * gridDim.x + blockIdx.x;
      m = tile / gn;
      n = (int)(tile - m * gn);
    }
    return m < gm;
  };

  // ---- issue cursors: A runs two slices ahead of the compute cursor, W one
  const int lr8 = lane >> 3, lc8 = lane & 7;   // A copy: lane -> (row lane/8 of the 8-row group, 16-byte chunk lane%8)
  const int lr4 = lane >> 2, lc4 = lane & 3;   // W copy: lane -> (row lane/4 of the 16-row group, chunk lane%4)
  const float* a_src[NA];
  const __bf16* b_src[NB];
  int64_t ca_j = 0, ca_m = 0, cb_j = 0, cb_m = 0;
  int ca_n = 0, cb_n = 0, ca_u = 0, cb_u = 0;
  bool ca_live = tile_at(0, ca_m, ca_n), cb_live = tile_at(0, cb_m, cb_n);
  int64_t issuedA = 0, issuedB = 0;
  auto issue_a = [&]() {
    if (!ca_live) return;
    if (ca_u == 0) {
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int r = 8 * (wave * NA + q) + lr8;
        int64_t row = ca_m * P_BM + r;
        row = row < M ? row : M - 1;
        a_src[q] = A + row * lda + 4 * (lc8 ^ ((r >> 1) & 7));
      }
    }
    float* st = ldsA + (issuedA % 3) * A_ST;
    const int64_t k0 = (int64_t)ca_u * XK;
#pragma unroll
    for (int q = 0; q < NA; ++q) glds16(a_src[q] + k0, st + (8 * (wave * NA + q)) * XK);
    ++issuedA;
    if (++ca_u == T) {
      ca_u = 0;
      ++ca_j;
      ca_live = tile_at(ca_j, ca_m, ca_n);
    }
  };
  auto issue_b = [&]() {
    if (!cb_live) return;
    if (cb_u == 0) {
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int gidx = (wave * NB + q) % B_GROUPS;   // (surplus copies repeat a group: same bytes, same place)
        const int img = gidx / (BN / 16), rb = gidx - img * (BN / 16);
        const int r = rb * 16 + lr4;
        b_src[q] = Wimg + ((int64_t)img * Np + (int64_t)cb_n * BN + r) * Kp + 8 * (lc4 ^ ((r >> 2) & 3));
      }
    }
    unsigned char* st = ldsB + (issuedB % 2) * B_ST;
    const int64_t k0 = (int64_t)cb_u * XK;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int gidx = (wave * NB + q) % B_GROUPS;
      glds16(b_src[q] + k0, st + gidx * 1024);
    }
    ++issuedB;
    if (++cb_u == T) {
      cb_u = 0;
      ++cb_j;
      cb_live = tile_at(cb_j, cb_m, cb_n);
    }
  };
  issue_b();
  issue_a();
  issue_a();

  int64_t stat_m = -1;   // row block / column tile whose statistics wait in stat_part
  int stat_n = 0;
  auto stats_readout = [&]() {
    for (int e = threadIdx.x; e < 2 * BN; e += P_TPB) {
      const int half = e / BN, c = e - half * BN;
      const int64_t n = (int64_t)stat_n * BN + c;
      const int64_t prow = stat_m * 2 + half;
      if (n < N && prow * 128 < M) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s1 += (double)stat_part[((half * 4 + w) * BN + c) * 2];
          s2 += (double)stat_part[((half * 4 + w) * BN + c) * 2 + 1];
        }
        double* dst = colstats + prow * 2 * N;
        dst[n] = s1;
        dst[N + n] = s2;
      }
    }
    stat_m = -1;
  };

  // ---- compute.  Waves 0-3 (group 0) and 4-7 (group 1; wave w + 4 shares its SIMD with wave w) run half a slice apart:
  // per slice, group 0 does  load/split(st 0) . MFMA(st 0) . load/split(st 1) . MFMA(st 1),  group 1 starts with the MFMAs
  // of the fragments it loaded at the end of the PREVIOUS slice and ends with the loads of st 1 -- so while one wave of a
  // SIMD reads LDS and splits its A fragment (vector work), the other one feeds the matrix core, with the one barrier per
  // slice the ring needs anyway.  (Both groups in step left the matrix core idle for half the time: SQ_VALU_MFMA_BUSY 47 %.)
  const bool late = wave >= 4;
  int64_t g = 0, landedA = 0, landedB = 0;
  f32x16 acc[NT];
  f32x4 fb[NT][3];
  bf16x8 ah, am, al;
  auto acc_init = [&](int64_t n0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t n = n0 + t * 32 + i;
      const float b = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = b;
    }
  };
  // fragments of step st of the slice in ring position (sa, sb): A raw -> three bf16 terms, W images as they are
  auto load_frags = [&](int sa, int sb, int st) {
    const uint32_t aB = ldsA_base + (uint32_t)(sa * A_ST * 4) + a_row;
    const uint32_t bB = ldsB_base + (uint32_t)(sb * B_ST) + b_row + 16u * ((uint32_t)(2 * st + h) ^ b_swz);
    const uint32_t c0 = 16u * ((uint32_t)(4 * st + 2 * h) ^ a_swz), c1 = 16u * ((uint32_t)(4 * st + 2 * h + 1) ^ a_swz);
    f32x4 a0, a1;
    asm volatile("ds_read_b128 %0, %1" : "=v"(a0) : "v"(aB + c0) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(a1) : "v"(aB + c1) : "memory");
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int sp = 0; sp < 3; ++sp)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[t][sp]) : "v"(bB), "n"(sp * BN * 64 + t * 32 * 64) : "memory");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * NT) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    split8(a0, a1, ah, am, al);       // under the latency of the W reads
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto products = [&]() {   // the 6 NT MFMAs of one step, smallest partial products first
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, fb[t][0]), bm = __builtin_bit_cast(bf16x8, fb[t][1]),
                   bl = __builtin_bit_cast(bf16x8, fb[t][2]);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // tile epilogue.  Everything issued so far (slices of the next tile) is waited for first: vmcnt retires in order, and
  // the stores below would otherwise sit in front of younger copies in every later counted wait.
  auto epilogue = [&](int64_t em, int en) {
    const int64_t m0 = em * P_BM, n0 = (int64_t)en * BN;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landedA = issuedA;
    landedB = issuedB;
    const bool interior = m0 + P_BM <= M && n0 + BN <= N;
    float* const crow = C + (m0 + wave * 32 + 4 * h) * ldc + n0 + i;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ncol = t * 32 + i;
      const int64_t n = n0 + ncol;
      float s1 = 0.f, s2 = 0.f;
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r];
          crow[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc + t * 32] = v;
          s1 += v;
          s2 += v * v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < M && n < N) {
            const float v = acc[t][r];
            C[m * ldc + n] = v;
            s1 += v;
            s2 += v * v;
          }
        }
      }
      if (colstats != nullptr) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) {
          stat_part[(wave * BN + ncol) * 2] = s1;
          stat_part[(wave * BN + ncol) * 2 + 1] = s2;
        }
      }
    }
    if (colstats != nullptr) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // table written before the next barrier
  };

  // One flat loop over this workgroup's slices plus a final drain pass; the epilogue of a tile runs at the start of the
  // NEXT tile's first pass (after group 1 has issued the tile's last MFMAs), so there is one copy of it in the code.
  int64_t tm = 0, pm = -1;   // current tile, finished tile whose epilogue is due
  int tn = 0, pn = 0, u = 0;
  int64_t j = 0;
  bool have = tile_at(0, tm, tn);
  bool pending = false;      // group 1: fragments of (previous slice, step 1) loaded, MFMAs not issued yet
  if (!have) return;         // (the whole workgroup: no barrier has been executed yet)
  for (;;) {
    if (have && (g >= landedB || g >= landedA)) {
      if (issuedA > g + 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (u == 1 && stat_m >= 0) stats_readout();   // parked by every wave before this barrier
    if (late && pending) {
      products();
      pending = false;
    }
    if (u == 0 && pm >= 0) {
      epilogue(pm, pn);
      if (colstats != nullptr) {
        stat_m = pm;
        stat_n = pn;
      }
      pm = -1;
    }
    if (!have) break;
    issue_b();   // (one at a time between the MFMA groups instead of in a burst here: tried, 5 % slower)
    issue_a();
    if (u == 0) acc_init((int64_t)tn * BN);
    const int sa = (int)(g % 3), sb = (int)(g & 1);
    load_frags(sa, sb, 0);
    products();
    load_frags(sa, sb, 1);
    if (!late)
      products();
    else
      pending = true;
    ++g;
    if (++u == T) {
      u = 0;
      pm = tm;
      pn = tn;
      ++j;
      have = tile_at(j, tm, tn);
    }
  }
  if (stat_m >= 0) {   // statistics of the last tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stats_readout();
  }
}

template <int BN>
void launch_x3_persistent(const float* A, int64_t lda, const __bf16* img, int64_t Np, int64_t Kp, const float* bias,
                          float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, double* colstats, hipStream_t s) {
  const int64_t gm = (M + P_BM - 1) / P_BM;
  const int gn = (int)((N + BN - 1) / BN);
  const int xcd_map = (gn <= 32 && 32 % gn == 0) ? 1 : 0;
  hipLaunchKernelGGL((gemm_x3_persistent_kernel<BN>), dim3(256), dim3(P_TPB), 0, s, A, lda, img, Np, Kp, bias, Y, ldy, M, N,
                     K, gm, gn, xcd_map, colstats);
}



// ------------------------------------------------------------------ two independent workgroups per CU (N > 64), software-pipelined
// The split product leaves ~1.6 us of matrix-core work per 32-deep slice of a 256 x 128 tile, which no longer hides the
// per-tile costs of the 8-wave kernel above (knock-out runs at K = N = 256: LDS-DMA traffic 0.33 ms and store epilogue
// 0.25 ms exposed out of 1.24 ms; matrix pipe 45 % busy).  As for the fp32 product (gemm_glds_pair_kernel), the cure is
// two 4-wave workgroups per CU on 128 x 128 tiles that share nothing: while one waits for its copies, stores a tile or sits
// at its barrier, the other one issues MFMAs.  The first form of that kernel (rounds 1-3: wave w = rows 32 w ..., a step =
// "14 LDS reads . wait . 41 vector instructions of the split . 24 MFMAs", 32-deep slices in a two-stage ring) reached
// 0.36-0.39 of the split product's matrix-core ceiling: a wave's own MFMAs never hid its own vector work, and the partner
// wave of the SIMD cannot hide it either -- while one wave streams MFMAs the other one's vector instructions issue at about
// one per MFMA (profiles/r03_pair_kernel_anatomy.txt); and a copy had one slice (~1.3 us at the full matrix rate, 2.7x less
// than in the fp32 kernel) to arrive from HBM.  Round 4 keeps the tile and changes the schedule (same bits as the first
// form on 13 shapes, profiles/r04_gemm_x3_anatomy.txt):
//   * one step = 16 contraction elements = 24 MFMAs per wave; the LDS reads and the three-way split of step s + 1 are
//     placed BETWEEN the MFMAs of step s (2 reads or 3 single-issue vector instructions per MFMA gap: an MFMA holds the
//     SIMD's issue port for 8 of its 32 cycles) into a second set of fragment registers;
//   * A (fp32 rows, from HBM): a 4-stage ring of 16-deep pieces PRIVATE to the wave that consumes them (32 rows x 64 B per
//     stage and wave, two LDS-DMA copies of 16 rows x 64 B; source chunk c of row r lands in slot 4 r + (c ^ (r >> 2 & 3)),
//     which makes a lane's two ds_read_b128 conflict-free) -- no barrier guards it, only the wave's own vmcnt, and a copy
//     has three steps to arrive;
//   * W: the three bf16 images are laid out by the split kernel in FRAGMENT order -- per (column tile, step) 12 pieces of
//     1 KiB = [image][32-column block][k half][column] x 16 B -- so a step's W operand is one contiguous 12 KiB, copied
//     by three fully coalesced LDS-DMA instructions per wave into a 3-stage ring and read back with one conflict-free
//     ds_read_b128 per fragment at immediate offsets; one barrier per step orders the ring;
//   * vmcnt is counted (loads, LDS-DMA and stores retire in issue order): a step waits for the copies of step s + 1 and
//     leaves the 7 younger ones in flight; behind a tile's 64 stores the count saturates at 63, which is still enough
//     (the copies waited for are older than the stores);
//   * lean issue streams as in the fp32 kernel: wave index as a scalar, copy sources as scalar base + 32-bit lane offset,
//     stores as scalar row base + lane offset, tile arithmetic in 32 bits.
// LDS: A 4 waves x 4 stages x 2 KiB + W 3 x 12 KiB + 4 KiB for the BatchNorm partial sums = 72 KiB per workgroup.
constexpr int L_TPB = 256;
constexpr int L_BM = 128, L_BN = 128;
constexpr int L_A_RING = 4, L_B_RING = 3;
constexpr int L_A_STAGE = 32 * 64;                      // bytes per wave and stage: 32 rows x 16 fp32
constexpr int L_B_STAGE = 12 * 1024;                    // bytes per stage: 3 images x 4 column blocks x 1 KiB
constexpr int L_LDS_A = 4 * L_A_RING * L_A_STAGE;       // 32 KiB
constexpr int L_LDS_B = L_B_RING * L_B_STAGE;           // 36 KiB
constexpr int L_LDS_STAT = 4 * L_BN * 2 * 4;            // 4 KiB
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// W (N x K fp32) -> the three bf16 images in fragment order: element (n, k) of image s sits at bf16 index
//   (((n / 128) T16 + k / 16) 3 + s) 2048 + ((n / 32) % 4) 512 + ((k / 8) % 2) 256 + (n % 32) 8 + k % 8;   rows >= N, columns >= K: 0
__global__ __launch_bounds__(256) void x3_split_weights_frag_kernel(const float* __restrict__ W, int64_t ldw, int64_t N,
                                                                    int64_t K, int64_t Np, int64_t T16,
                                                                    __bf16* __restrict__ img) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t Kp = T16 * 16;
  if (e >= Np * Kp) return;
  const int64_t n = e / Kp, k = e - n * Kp;
  const float x = (n < N && k < K) ? W[n * ldw + k] : 0.f;
  __bf16 h, m, l;
  split3(x, h, m, l);
  const int64_t stage = (n >> 7) * T16 + (k >> 4);
  const int64_t in = ((n >> 5) & 3) * 512 + ((k >> 3) & 1) * 256 + (n & 31) * 8 + (k & 7);
  img[(stage * 3 + 0) * 2048 + in] = h;
  img[(stage * 3 + 1) * 2048 + in] = m;
  img[(stage * 3 + 2) * 2048 + in] = l;
}

// diagnostic build (ccn_gemm_x3_debug): per wave, where its cycles go (s_memtime stamps), 16 words per wave
__device__ unsigned long long* g_x3_dbg_dev = nullptr;
#define X3_STAMP(t)                                                              \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)

__device__ __forceinline__ uint32_t x3_cvt_pk(float a, float b) {   // (bf16(a), bf16(b)), round to nearest even
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (inline asm: the compiler otherwise pairs the subtractions into v_pk_add_f32, which costs an MFMA gap far more than two
// plain ones -- MI355X_MICROARCH.md, per-instruction cycle constants)
__device__ __forceinline__ float x3_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct X3Frag {
  u32x4 a[3];      // this lane's 8 contraction elements of its A row: high / middle / low terms
  f32x4 b[4][3];   // [32-column block][image]
};

template <bool STAMP>
__global__ __launch_bounds__(L_TPB, 2) void gemm_x3_lean_kernel(const float* __restrict__ A, int64_t lda,
                                                                const unsigned char* __restrict__ Wfrag,
                                                                const float* __restrict__ bias, float* __restrict__ C,
                                                                int64_t ldc, int64_t M, int64_t N, int64_t K, int64_t tiles,
                                                                int64_t gn, int xcd_order, double* __restrict__ colstats) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[L_LDS_A + L_LDS_B + L_LDS_STAT];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 31, h = lane >> 5;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const uint32_t ldsA_w = lds_base + (uint32_t)(wave * L_A_RING * L_A_STAGE);
  const uint32_t ldsB = lds_base + L_LDS_A;
  float* const stat_part = reinterpret_cast<float*>(lds + L_LDS_A + L_LDS_B);   // [4 waves][128 columns][sum, sum of squares]
  const int T16 = (int)(K / 16);

  // fragment reads: A row i, source chunks 2 h and 2 h + 1 (slot 4 r + (c ^ (r >> 2 & 3))); W column i, k half h
  const uint32_t a_sw = (uint32_t)((i >> 2) & 3);
  const uint32_t a_rd0 = (uint32_t)(4 * i + ((2 * h) ^ a_sw)) * 16u, a_rd1 = (uint32_t)(4 * i + ((2 * h + 1) ^ a_sw)) * 16u;
  const uint32_t b_rd = (uint32_t)(h * 512 + i * 16);

  // ---- tiles of this workgroup (as gemm_glds_pair_kernel: 64 workgroup slots per XCD walk the column tiles of a row block)
  const uint32_t gnu = (uint32_t)gn;
  const uint32_t gm_tiles = (uint32_t)tiles / gnu;
  const bool xcd_map = xcd_order && gridDim.x == 512 && gnu <= 64 && 64 % gnu == 0;
  const uint32_t slot = blockIdx.x >> 3;
  const uint32_t rows_per_step = 512u / gnu, slot_row = (slot / gnu) * 8 + (blockIdx.x & 7), slot_col = slot % gnu;
  auto tile_of = [&](int64_t j) -> int64_t {
    if (xcd_map) {
      const uint32_t m = (uint32_t)j * rows_per_step + slot_row;
      return m < gm_tiles ? (int64_t)(m * gnu + slot_col) : tiles;
    }
    return j * gridDim.x + blockIdx.x;
  };
  auto tile_row = [&](int64_t t) -> int64_t { return (int64_t)((uint32_t)t / gnu); };
  auto tile_col = [&](int64_t t) -> int64_t { return (int64_t)((uint32_t)t % gnu); };

  // ---- copy cursors: each walks the steps of this workgroup's tiles in order (A runs one step ahead of W)
  // An exhausted cursor keeps re-copying its last step (valid memory, into stages nobody reads any more): every step then
  // issues the same five copies and the counted waits below hold to the end.
  int64_t bj = 0, b_tile = tile_of(0);
  int b_u = 0, b_stage = 0;
  const unsigned char* b_ptr = Wfrag;
  unsigned char* b_dst = lds;
  const uint32_t b_lane = (uint32_t)(lane * 16);
  auto b_begin = [&]() {
    if (b_u == 0 && b_tile < tiles) b_ptr = Wfrag + ((tile_col(b_tile) * T16) * 12 + 3 * wave) * 1024;
    b_dst = lds + L_LDS_A + b_stage * L_B_STAGE + 3 * wave * 1024;
  };
  auto b_piece = [&](int p) {
    const unsigned char* const piece = b_ptr + p * 1024;   // (scalar base + 32-bit lane offset: no 64-bit vector arithmetic)
    glds16(piece + b_lane, b_dst + p * 1024);
  };
  auto b_end = [&]() {
    b_stage = b_stage == L_B_RING - 1 ? 0 : b_stage + 1;
    if (b_tile < tiles) {
      if (++b_u == T16) {
        b_u = 0;
        b_tile = tile_of(++bj);
        if (b_tile >= tiles) b_u = 1;   // exhausted: b_ptr stays on the last step
      } else {
        b_ptr += 12 * 1024;
      }
    }
  };
  int64_t aj = 0, a_tile = tile_of(0);
  int a_u = 0, a_stage = 0;
  const char* a_ptr = reinterpret_cast<const char*>(A);
  unsigned char* a_dst = lds;
  uint32_t a_off32[2] = {0u, 0u};
  auto a_begin = [&]() {
    if (a_u == 0 && a_tile < tiles) {
      const int64_t m0 = tile_row(a_tile) * L_BM;
      a_ptr = reinterpret_cast<const char*>(A + m0 * lda);
      const int64_t rows_left = M - m0;   // >= 1; rows beyond the matrix re-read its last row (they feed outputs never stored)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = 16 * q + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        int64_t rr = 32 * wave + r;
        rr = rr < rows_left ? rr : rows_left - 1;
        a_off32[q] = (uint32_t)((rr * lda + 4 * c) * 4);
      }
    }
    a_dst = lds + (wave * L_A_RING + a_stage) * L_A_STAGE;
  };
  auto a_piece = [&](int q) { glds16(a_ptr + a_off32[q], a_dst + q * 1024); };
  auto a_end = [&]() {
    a_stage = (a_stage + 1) & (L_A_RING - 1);
    if (a_tile < tiles) {
      if (++a_u == T16) {
        a_u = 0;
        a_tile = tile_of(++aj);
        if (a_tile >= tiles) a_u = 1;   // exhausted: a_ptr stays on the last step
      } else {
        a_ptr += 64;
      }
    }
  };
  auto issue_b = [&]() { b_begin(); b_piece(0); b_piece(1); b_piece(2); b_end(); };
  auto issue_a = [&]() { a_begin(); a_piece(0); a_piece(1); a_end(); };

  X3Frag F0, F1;
  f32x16 acc[4];
  int rdA = 1, rdB = 1;   // ring stages the NEXT step's operands sit in

  // ---- BatchNorm partial sums of a finished tile: parked in stat_part by its epilogue, written out behind the next barrier
  int64_t stat_tile = -1;
  auto stats_readout = [&]() {
    const int64_t pm = tile_row(stat_tile), pn0 = tile_col(stat_tile) * L_BN;
    for (int c = threadIdx.x; c < L_BN; c += L_TPB) {
      const int64_t n = pn0 + c;
      if (n < N) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s1 += (double)stat_part[(w * L_BN + c) * 2];
          s2 += (double)stat_part[(w * L_BN + c) * 2 + 1];
        }
        double* dst = colstats + pm * 2 * N;   // one partial row per 128-row block (ccn_stats_rows)
        dst[n] = s1;
        dst[N + n] = s2;
      }
    }
    stat_tile = -1;
  };

  unsigned long long st_t0 = 0, st_a = 0, st_b = 0, st_r0 = 0;
  uint32_t sum_w = 0, sum_b = 0, sum_i = 0, sum_c = 0, sum_e = 0, n_st = 0;

  // ---- prologue: W(0) A(0) W(1) A(1) W(2) A(2) A(3) issued, the fragments of step 0 in F0
  issue_b(); issue_a();
  issue_b(); issue_a();
  issue_b(); issue_a();
  issue_a();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    f32x4 r0, r1;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(ldsA_w + a_rd0) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(r1) : "v"(ldsA_w + a_rd1) : "memory");
#pragma unroll
    for (int q = 0; q < 12; ++q)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F0.b[q & 3][q >> 2]) : "v"(ldsB + b_rd), "n"(q * 1024) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(r0), "+v"(r1), "+v"(F0.b[0][0]), "+v"(F0.b[1][0]), "+v"(F0.b[2][0]), "+v"(F0.b[3][0]), "+v"(F0.b[0][1]),
                   "+v"(F0.b[1][1]), "+v"(F0.b[2][1]), "+v"(F0.b[3][1]), "+v"(F0.b[0][2]), "+v"(F0.b[1][2]), "+v"(F0.b[2][2]),
                   "+v"(F0.b[3][2])
                 :
                 : "memory");
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const float x0 = pr < 2 ? r0[2 * pr] : r1[2 * pr - 4], x1 = pr < 2 ? r0[2 * pr + 1] : r1[2 * pr - 3];
      const uint32_t ph = x3_cvt_pk(x0, x1);
      const float q0 = x3_sub(x0, __uint_as_float(ph << 16)), q1 = x3_sub(x1, __uint_as_float(ph & 0xffff0000u));
      const uint32_t pm = x3_cvt_pk(q0, q1);
      const float s0 = x3_sub(q0, __uint_as_float(pm << 16)), s1 = x3_sub(q1, __uint_as_float(pm & 0xffff0000u));
      F0.a[0][pr] = ph;
      F0.a[1][pr] = pm;
      F0.a[2][pr] = x3_cvt_pk(s0, s1);
    }
  }

  if (STAMP) {
    X3_STAMP(st_t0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_r0) : : "memory");
  }

  int epi_age = 8;          // step tops since the last full-tile epilogue (its 64 stores sit in the vmcnt queue for two of them)
  bool drain_next = false;  // an edge tile's epilogue (fewer stores than 63): drain the queue once
  // One step: [wait for the copies of step s + 1] [barrier] [issue W(s + 3), A(s + 4)] then the 24 MFMAs of step s on the
  // fragments in `cur`, with the LDS reads and the split of step s + 1 into `nxt` between them.
  auto step = [&](X3Frag& cur, X3Frag& nxt) {
    if (STAMP) X3_STAMP(st_a);
    if (epi_age < 2)
      asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else if (drain_next)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    drain_next = false;
    ++epi_age;
    if (STAMP) { X3_STAMP(st_b); sum_w += (uint32_t)(st_b - st_a); }
    __builtin_amdgcn_s_barrier();
    if (STAMP) { X3_STAMP(st_a); sum_b += (uint32_t)(st_a - st_b); }
    if (stat_tile >= 0) stats_readout();
    if (STAMP) { X3_STAMP(st_b); sum_i += (uint32_t)(st_b - st_a); }

    const uint32_t at = ldsA_w + (uint32_t)(rdA * L_A_STAGE);
    const uint32_t bt = ldsB + (uint32_t)(rdB * L_B_STAGE) + b_rd;
    rdA = (rdA + 1) & (L_A_RING - 1);
    rdB = rdB == L_B_RING - 1 ? 0 : rdB + 1;
    f32x4 r0, r1;
    uint32_t ph = 0, pm = 0;
    float x0 = 0.f, x1 = 0.f, q0 = 0.f, q1 = 0.f, f0 = 0.f, f1 = 0.f, s0 = 0.f;
    constexpr int SA[6] = {2, 0, 1, 1, 0, 0};   // smallest partial products first
    constexpr int SB[6] = {0, 2, 1, 0, 1, 0};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int p = m >> 2, t = m & 3;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[SA[p]]),
                                                       __builtin_bit_cast(bf16x8, cur.b[t][SB[p]]), acc[t], 0, 0, 0);
      if (m == 0) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(at + a_rd0) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(r1) : "v"(at + a_rd1) : "memory");
      } else if (m <= 6) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = 2 * (m - 1) + qq;   // piece = image * 4 + column block
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nxt.b[q & 3][q >> 2]) : "v"(bt), "n"(q * 1024) : "memory");
        }
      } else if (m <= 22) {
        const int pr = (m - 7) >> 2, c = (m - 7) & 3;   // pair of contraction elements, quarter of its split
        if (m == 7) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(r0), "+v"(r1) : : "memory");   // the two A reads have landed
        if (c == 0) {
          x0 = pr < 2 ? r0[2 * pr] : r1[2 * pr - 4];
          x1 = pr < 2 ? r0[2 * pr + 1] : r1[2 * pr - 3];
          ph = x3_cvt_pk(x0, x1);
          f0 = __uint_as_float(ph << 16);
          f1 = __uint_as_float(ph & 0xffff0000u);
          nxt.a[0][pr] = ph;
        } else if (c == 1) {
          q0 = x3_sub(x0, f0);
          q1 = x3_sub(x1, f1);
          pm = x3_cvt_pk(q0, q1);
          nxt.a[1][pr] = pm;
        } else if (c == 2) {
          f0 = __uint_as_float(pm << 16);
          f1 = __uint_as_float(pm & 0xffff0000u);
          s0 = x3_sub(q0, f0);
        } else {
          nxt.a[2][pr] = x3_cvt_pk(s0, x3_sub(q1, f1));
        }
      }
      // the copies of W(s + 3) and A(s + 4), one per third gap of the split (the stages they go to were read a step ago:
      // this workgroup's barrier above is behind those reads)
      if (m == 9) { b_begin(); b_piece(0); }
      if (m == 12) b_piece(1);
      if (m == 15) { b_piece(2); b_end(); }
      if (m == 18) { a_begin(); a_piece(0); }
      if (m == 21) { a_piece(1); a_end(); }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(nxt.b[0][0]), "+v"(nxt.b[1][0]), "+v"(nxt.b[2][0]), "+v"(nxt.b[3][0]), "+v"(nxt.b[0][1]), "+v"(nxt.b[1][1]),
                   "+v"(nxt.b[2][1]), "+v"(nxt.b[3][1]), "+v"(nxt.b[0][2]), "+v"(nxt.b[1][2]), "+v"(nxt.b[2][2]), "+v"(nxt.b[3][2])
                 :
                 : "memory");
    if (STAMP) { X3_STAMP(st_a); sum_c += (uint32_t)(st_a - st_b); ++n_st; }
  };

  for (int64_t j = 0, tile; (tile = tile_of(j)) < tiles; ++j) {
    const int64_t m0 = tile_row(tile) * L_BM, n0 = tile_col(tile) * L_BN;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t n = n0 + t * 32 + i;
      const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = bv;
    }
    for (int u = 0; u < T16; u += 2) {
      step(F0, F1);
      step(F1, F0);
    }

    // ---- tile epilogue (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
    if (STAMP) X3_STAMP(st_a);
    const bool interior = m0 + L_BM <= M && n0 + L_BN <= N;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ncol = t * 32 + i;
      const int64_t n = n0 + ncol;
      float s1 = 0.f, s2 = 0.f;
      if (interior) {
        // store address = a scalar row base (tile, wave band, register's row) + this lane's 32-bit offset (its 4 h rows, its column)
        float* const cbase = C + (m0 + wave * 32) * ldc + n0 + t * 32;
        const uint32_t lane_off = (uint32_t)((4 * h * ldc + i) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* const rowp = cbase + (int64_t)((r & 3) + 8 * (r >> 2)) * ldc;
          asm volatile("global_store_dword %0, %1, %2" : : "v"(lane_off), "v"(acc[t][r]), "s"(rowp) : "memory");
        }
        if (colstats != nullptr) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[t][r];
            s1 += v;
            s2 = __builtin_fmaf(v, v, s2);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < M && n < N) {
            const float v = acc[t][r];
            C[m * ldc + n] = v;
            s1 += v;
            s2 = __builtin_fmaf(v, v, s2);
          }
        }
      }
      if (colstats != nullptr) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) {
          stat_part[(wave * L_BN + ncol) * 2] = s1;
          stat_part[(wave * L_BN + ncol) * 2 + 1] = s2;
        }
      }
    }
    if (colstats != nullptr) {
      stat_tile = tile;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // table written before this wave reaches the next barrier
    }
    if (interior)
      epi_age = 0;
    else
      drain_next = true;
    if (STAMP) { X3_STAMP(st_b); sum_e += (uint32_t)(st_b - st_a); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA copy of this wave may land after the workgroup has gone
  if (STAMP && g_x3_dbg_dev != nullptr) {
    unsigned long long t1, r1;
    X3_STAMP(t1);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) : : "memory");
    if (lane == 0) {
      unsigned long long* d = g_x3_dbg_dev + ((int64_t)blockIdx.x * 4 + wave) * 16;
      d[0] = t1 - st_t0; d[1] = r1 - st_r0; d[2] = sum_w; d[3] = sum_b; d[4] = sum_i; d[5] = sum_c; d[6] = sum_e; d[7] = n_st;
      d[8] = st_t0; d[9] = t1; d[10] = st_r0; d[11] = r1;
    }
  }
  if (stat_tile >= 0) {   // statistics of the last tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stats_readout();
  }
}

static std::atomic<void*> g_x3_dbg_host{nullptr};
static std::atomic<bool> g_use_pair{true};   // A/B hook (ccn_gemm_x3_use_persistent(2) = no paired kernel)

void launch_x3_lean(const float* A, int64_t lda, const __bf16* img, const float* bias, float* Y, int64_t ldy, int64_t M,
                    int64_t N, int64_t K, double* colstats, hipStream_t s) {
  const int64_t gm = (M + L_BM - 1) / L_BM, gn = (N + L_BN - 1) / L_BN;
  const int64_t tiles = gm * gn;
  const int64_t grid = tiles < 512 ? tiles : 512;   // two workgroups per CU
  const unsigned char* wf = reinterpret_cast<const unsigned char*>(img);
  if (g_x3_dbg_host != nullptr)
    hipLaunchKernelGGL((gemm_x3_lean_kernel<true>), dim3((unsigned)grid), dim3(L_TPB), 0, s, A, lda, wf, bias, Y, ldy, M, N, K,
                       tiles, gn, 1, colstats);
  else
    hipLaunchKernelGGL((gemm_x3_lean_kernel<false>), dim3((unsigned)grid), dim3(L_TPB), 0, s, A, lda, wf, bias, Y, ldy, M, N, K,
                       tiles, gn, 1, colstats);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int BN, int WMW>
void launch_x3(const float* A, int64_t lda, const __bf16* img, int64_t Np, int64_t Kp, const float* bias, float* Y,
               int64_t ldy, int64_t M, int64_t N, int64_t K, double* colstats, hipStream_t s) {
  const int64_t gm = (M + X_BM - 1) / X_BM;
  const int gn = (int)((N + BN - 1) / BN);
  const int64_t grid = (gm + 7) / 8 * 8 * gn;
  hipLaunchKernelGGL((gemm_x3_kernel<BN, WMW>), dim3((unsigned)grid), dim3(X_TPB), 0, s, A, lda, img, Np, Kp, bias, Y, ldy,
                     M, N, K, gm, gn, colstats);
}

}  // namespace

extern "C" {

int ccn_gemm_x3_use_persistent(int on) {
  g_use_persistent = on != 0;
  g_use_pair = on == 1;
  return CCN_OK;
}

int ccn_gemm_x3_debug(void* buf) {   // diagnostic: 512 x 4 x 16 uint64 words (see STAMP in gemm_x3_lean_kernel); nullptr = off
  g_x3_dbg_host = buf;
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_x3_dbg_dev), &p, sizeof(p)) == hipSuccess ? CCN_OK : CCN_ERR_ARG;
}

int64_t ccn_gemm_x3_workspace_bytes(int64_t N, int64_t K) {
  if (N <= 0 || K <= 0) return 0;
  const int64_t Np = (N + X_NPAD - 1) / X_NPAD * X_NPAD, Kp = (K + XK - 1) / XK * XK;
  return 3 * Np * Kp * 2;
}

int ccn_gemm_nt_x3(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                   int64_t M, int64_t N, int64_t K, double* colstats, void* wsplit, int64_t wsplit_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(A && W && Y, "gemm_nt_x3: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N, "gemm_nt_x3: bad sizes M=%lld N=%lld K=%lld",
              (long long)M, (long long)N, (long long)K);
  CCN_REQUIRE(aligned16(A) && lda % 4 == 0 && lda >= 4,
              "gemm_nt_x3: the row operand must be 16-byte aligned with a leading dimension that is a multiple of 4");
  CCN_REQUIRE(wsplit && aligned16(wsplit) && wsplit_bytes >= ccn_gemm_x3_workspace_bytes(N, K),
              "gemm_nt_x3: weight-split scratch too small (need %lld bytes)", (long long)ccn_gemm_x3_workspace_bytes(N, K));
  if (M == 0) return CCN_OK;
  const int64_t Np = (N + X_NPAD - 1) / X_NPAD * X_NPAD, Kp = (K + XK - 1) / XK * XK;
  const int64_t gm = (M + X_BM - 1) / X_BM;
  CCN_REQUIRE((gm + 7) / 8 * 8 * ((N + 31) / 32) <= 2147483647LL, "gemm_nt_x3: grid too large");
  __bf16* img = reinterpret_cast<__bf16*>(wsplit);
  // the software-pipelined paired kernel: N > 64, whole 32-deep slices, at least 128 tiles of 128 x 128 (< 2^31 of them)
  if (g_use_persistent && g_use_pair && N > 64 && K % XK == 0 && K >= 64 &&
      ((M + L_BM - 1) / L_BM) * ((N + L_BN - 1) / L_BN) >= 128 && ((M + L_BM - 1) / L_BM) * ((N + L_BN - 1) / L_BN) < (1LL << 31) &&
      (L_BM - 1) * lda < (1LL << 29)) {
    const int64_t T16 = K / 16;
    hipLaunchKernelGGL(x3_split_weights_frag_kernel, dim3((unsigned)((Np * K + 255) / 256)), dim3(256), 0, s, W, ldw, N, K, Np,
                       T16, img);
    launch_x3_lean(A, lda, img, bias, Y, ldy, M, N, K, colstats, s);
    CCN_LAUNCH_OK("gemm_nt_x3");
    return CCN_OK;
  }
  hipLaunchKernelGGL(x3_split_weights_kernel, dim3((unsigned)((Np * Kp + 255) / 256)), dim3(256), 0, s, W, ldw, N, K, Np,
                     Kp, img);
  // many 256-row tiles and whole K slices: the persistent LDS-DMA kernel (one workgroup per CU, 256 CUs)
  const bool persistent = g_use_persistent && K % XK == 0 && K >= 64 &&
                          ((M + P_BM - 1) / P_BM) * ((N + 127) / 128) >= 512;
  if (persistent) {
    if (N <= 32)
      launch_x3_persistent<32>(A, lda, img, Np, Kp, bias, Y, ldy, M, N, K, colstats, s);
    else if (N <= 64)
      launch_x3_persistent<64>(A, lda, img, Np, Kp, bias, Y, ldy, M, N, K, colstats, s);
    else
      launch_x3_persistent<128>(A, lda, img, Np, Kp, bias, Y, ldy, M, N, K, colstats, s);
    CCN_LAUNCH_OK("gemm_nt_x3");
    return CCN_OK;
  }
  if (N <= 32)
    launch_x3<32, 4>(A, lda, img, Np, Kp, bias, Y, ldy, M, N, K, colstats, s);
  else if (N <= 64)
    launch_x3<64, 2>(A, lda, img, Np, Kp, bias, Y, ldy, M, N, K, colstats, s);
  else
    launch_x3<128, 2>(A, lda, img, Np, Kp, bias, Y, ldy, M, N, K, colstats, s);
  CCN_LAUNCH_OK("gemm_nt_x3");
  return CCN_OK;
}

}  // extern "C"

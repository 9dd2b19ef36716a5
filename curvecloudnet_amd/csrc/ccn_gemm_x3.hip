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
                                                           int gn, double* __restrict__ colstats, int knock) {
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
    if (has_next && !(knock & 1)) load_slice(k0 + XK);   // in flight during the MFMAs below
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
      if (knock & 2) continue;
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
      if (!(knock & 4)) store_slice();
      __syncthreads();
    }
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  double* stat_lds = reinterpret_cast<double*>(lds);   // [WMW][BN][2]
  if (knock & 8) return;
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

static int g_knock = 0;
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int BN, int WMW>
void launch_x3(const float* A, int64_t lda, const __bf16* img, int64_t Np, int64_t Kp, const float* bias, float* Y,
               int64_t ldy, int64_t M, int64_t N, int64_t K, double* colstats, hipStream_t s) {
  const int64_t gm = (M + X_BM - 1) / X_BM;
  const int gn = (int)((N + BN - 1) / BN);
  const int64_t grid = (gm + 7) / 8 * 8 * gn;
  hipLaunchKernelGGL((gemm_x3_kernel<BN, WMW>), dim3((unsigned)grid), dim3(X_TPB), 0, s, A, lda, img, Np, Kp, bias, Y, ldy,
                     M, N, K, gm, gn, colstats, g_knock);
}

}  // namespace

extern "C" {

int ccn_gemm_x3_knock(int k) { g_knock = k; return 0; }
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
  hipLaunchKernelGGL(x3_split_weights_kernel, dim3((unsigned)((Np * Kp + 255) / 256)), dim3(256), 0, s, W, ldw, N, K, Np,
                     Kp, img);
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

// fp32 MFMA GEMMs for the per-point / per-edge MLP stack and the
// BatchNorm / activation kernels around them.
//
// One kernel template serves forward (Y = A W^T), data-gradient (dX = dY W) and weight-gradient
// (dW += dY^T X, split over the row dimension with fp32 atomics): the operands differ only in which
// index is contiguous in memory.
//   KC operand: element (row r, k) at p[r*ld + k]   -> LDS tile [rows][BK+4], fragments by ds_read_b128
//   MC operand: element (row r, k) at p[k*ld + r]   -> LDS tile [BK][rows+4], fragments by ds_read_b32
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  Lane l = (i = l&31, h = l>>5) supplies A[i][k] and
// B[k][i] for k = 8q + 4h + s of the current 32-wide K slice (q = 0..3 picks a float4, s its element):
// the K index is permuted identically for A and B, which leaves the contraction unchanged and lets a
// lane fetch four MFMA steps with one 16-byte LDS read.
// Workgroup = 4 waves; WM waves along M (32 rows each), 4/WM along N.
#include "ccn_common.h"

#include <atomic>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;
constexpr int KC_LD = BK + 4;  // 144-byte rows: 16-byte aligned, conflict-free ds_read_b128
constexpr int GEMM_TPB = 256;

enum Layout { KC = 0, MC = 1 };
enum Epilogue { EPI_STORE = 0, EPI_ATOMIC = 1 };

template <int ROWS, int LAY>
struct Tile {
  static constexpr int FLOATS = LAY == KC ? ROWS * KC_LD : BK * (ROWS + 4);
  static constexpr int VEC4 = ROWS * BK / 4;           // float4 slots in the tile
  static constexpr int PER_THREAD = VEC4 / GEMM_TPB;   // >= 1 for ROWS >= 32
};

// global -> registers for one K slice of one operand
template <int ROWS, int LAY>
__device__ __forceinline__ void tile_load(const float* __restrict__ p, int64_t ld, int64_t row0, int64_t nrows,
                                          int64_t k0, int64_t kend, bool vec_ok,
                                          float4 (&regs)[Tile<ROWS, LAY>::PER_THREAD]) {
#pragma unroll
  for (int it = 0; it < Tile<ROWS, LAY>::PER_THREAD; ++it) {
    const int slot = threadIdx.x + it * GEMM_TPB;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LAY == KC) {
      const int r = slot >> 3, kq = slot & 7;
      const int64_t row = row0 + r, k = k0 + kq * 4;
      if (row < nrows && k < kend) {
        const float* src = p + row * ld + k;
        if (vec_ok && k + 4 <= kend) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (k + 1 < kend) v.y = src[1];
          if (k + 2 < kend) v.z = src[2];
          if (k + 3 < kend) v.w = src[3];
        }
      }
    } else {
      constexpr int Q = ROWS / 4;
      const int kk = slot / Q, rq = slot - kk * Q;
      const int64_t k = k0 + kk, row = row0 + rq * 4;
      if (k < kend && row < nrows) {
        const float* src = p + k * ld + row;
        if (vec_ok && row + 4 <= nrows) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (row + 1 < nrows) v.y = src[1];
          if (row + 2 < nrows) v.z = src[2];
          if (row + 3 < nrows) v.w = src[3];
        }
      }
    }
    regs[it] = v;
  }
}

template <int ROWS, int LAY>
__device__ __forceinline__ void tile_store(float* __restrict__ lds,
                                           const float4 (&regs)[Tile<ROWS, LAY>::PER_THREAD]) {
#pragma unroll
  for (int it = 0; it < Tile<ROWS, LAY>::PER_THREAD; ++it) {
    const int slot = threadIdx.x + it * GEMM_TPB;
    if (LAY == KC) {
      const int r = slot >> 3, kq = slot & 7;
      *reinterpret_cast<float4*>(lds + r * KC_LD + kq * 4) = regs[it];
    } else {
      constexpr int Q = ROWS / 4;
      const int kk = slot / Q, rq = slot - kk * Q;
      *reinterpret_cast<float4*>(lds + kk * (ROWS + 4) + rq * 4) = regs[it];
    }
  }
}

// four MFMA steps' worth of one operand for lane (i, h), K group q
template <int ROWS, int LAY>
__device__ __forceinline__ float4 frag_read(const float* __restrict__ lds, int r, int q, int h) {
  if (LAY == KC) {
    return *reinterpret_cast<const float4*>(lds + r * KC_LD + 8 * q + 4 * h);
  } else {
    const float* base = lds + (8 * q + 4 * h) * (ROWS + 4) + r;
    return make_float4(base[0], base[ROWS + 4], base[2 * (ROWS + 4)], base[3 * (ROWS + 4)]);
  }
}

// The bias is the INITIAL value of the accumulators (column n = lane&31 of every register of a tile): loading it
// in the epilogue would put a global load in front of every conditional store, and since vmcnt retires loads and
// stores in order hipcc then waits vmcnt(0) before each of the 16*NT stores (measured: ~14 us per 256x128 tile).
template <int NT>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[NT], const float* __restrict__ bias, int64_t n_first, int64_t N,
                                         int i) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t n = n_first + t * 32 + i;
    const float b = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = b;
  }
}

// accumulator tiles -> C (+BatchNorm partial statistics) or atomic accumulation.
// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
template <int BM, int BN, int WM, int EPI, int NT>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[NT], float* __restrict__ As, const float* __restrict__ bias,
                                              float* __restrict__ C, int64_t ldc, int64_t M, int64_t N, int64_t m0,
                                              int64_t n0, int wm, int wn, int i, int h,
                                              double* __restrict__ colstats) {
  constexpr int WN = 4 / WM;
  constexpr int WCOLS = BN / WN;
  double* stat_lds = reinterpret_cast<double*>(As);  // [WM][BN][2] doubles, reused after the last barrier
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ncol_local = wn * WCOLS + t * 32 + i;
    const int64_t n = n0 + ncol_local;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < M && n < N) {
        const float v = acc[t][r];
        if (EPI == EPI_STORE) {
          C[m * ldc + n] = v;
          if (colstats != nullptr) {
            s1 += (double)v;
            s2 += (double)v * (double)v;
          }
        } else {
          atomicAdd(&C[m * ldc + n], v);
        }
      }
    }
    if (EPI == EPI_STORE && colstats != nullptr) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        stat_lds[(wm * BN + ncol_local) * 2] = s1;
        stat_lds[(wm * BN + ncol_local) * 2 + 1] = s2;
      }
    }
  }
  if (EPI == EPI_STORE && colstats != nullptr) {
    __syncthreads();
    for (int c = threadIdx.x; c < BN; c += GEMM_TPB) {
      const int64_t n = n0 + c;
      if (n < N) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          s1 += stat_lds[(w * BN + c) * 2];
          s2 += stat_lds[(w * BN + c) * 2 + 1];
        }
        double* dst = colstats + (int64_t)blockIdx.x * 2 * N;  // one partial row per M tile (deterministic)
        dst[n] = s1;
        dst[N + n] = s2;
      }
    }
  }
}

// C[M x N] (+)= A[M x K] * B[K x N]; K range split over blockIdx.z
template <int BM, int BN, int WM, int ALAY, int BLAY, int EPI>
__global__ __launch_bounds__(GEMM_TPB) void gemm_kernel(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb,
                                                        const float* __restrict__ bias, float* __restrict__ C,
                                                        int64_t ldc, int64_t M, int64_t N, int64_t K, int64_t kchunk,
                                                        int a_vec, int b_vec, double* __restrict__ colstats) {
  constexpr int WN = 4 / WM;
  constexpr int WCOLS = BN / WN;  // columns per wave
  constexpr int NT = WCOLS / 32;  // 32x32 MFMA tiles per wave along N
  static_assert(BM == 32 * WM && NT >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) float As[Tile<BM, ALAY>::FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[Tile<BN, BLAY>::FLOATS];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int i = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;

  f32x16 acc[NT];
  acc_init<NT>(acc, (EPI == EPI_STORE && blockIdx.z == 0) ? bias : nullptr, n0 + wn * WCOLS, N, i);

  float4 ra[Tile<BM, ALAY>::PER_THREAD], rb[Tile<BN, BLAY>::PER_THREAD];
  if (kbeg < kend) {
    tile_load<BM, ALAY>(A, lda, m0, M, kbeg, kend, a_vec != 0, ra);
    tile_load<BN, BLAY>(B, ldb, n0, N, kbeg, kend, b_vec != 0, rb);
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    tile_store<BM, ALAY>(As, ra);
    tile_store<BN, BLAY>(Bs, rb);
    __syncthreads();
    if (k0 + BK < kend) {  // next slice in flight while this one is multiplied
      tile_load<BM, ALAY>(A, lda, m0, M, k0 + BK, kend, a_vec != 0, ra);
      tile_load<BN, BLAY>(B, ldb, n0, N, k0 + BK, kend, b_vec != 0, rb);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a4 = frag_read<BM, ALAY>(As, wm * 32 + i, q, h);
      float4 b4[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b4[t] = frag_read<BN, BLAY>(Bs, wn * WCOLS + t * 32 + i, q, h);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[t].w, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  gemm_epilogue<BM, BN, WM, EPI, NT>(acc, As, bias, C, ldc, M, N, m0, n0, wm, wn, i, h, colstats);
}


// ------------------------------------------------------------------ fast path: 16-byte aligned operands
// Branch-free tile loads (addresses clamped into the allocation; rows/columns outside the problem only
// ever feed output elements that are never stored, the K tail is zeroed with selects), two LDS buffers
// and ONE barrier per 32-deep K slice: the global loads of slice s+1 are issued before the MFMAs of
// slice s and written to the other buffer after them.
template <int ROWS, int LAY>
struct FastLoader {
  static constexpr int PT = Tile<ROWS, LAY>::PER_THREAD;
  const float* ptr[PT];   // KC: &p[row*ld + kq*4]   MC: &p[rq*4] (row index contiguous)
  int64_t ld;
  int kk[PT];             // KC: kq*4   MC: k row inside the slice

  __device__ __forceinline__ void init(const float* __restrict__ p, int64_t ld_, int64_t row0, int64_t nrows) {
    ld = ld_;
#pragma unroll
    for (int it = 0; it < PT; ++it) {
      const int slot = threadIdx.x + it * GEMM_TPB;
      if (LAY == KC) {
        const int r = slot >> 3, kq = slot & 7;
        int64_t row = row0 + r;
        row = row < nrows ? row : nrows - 1;
        ptr[it] = p + row * ld + kq * 4;
        kk[it] = kq * 4;
      } else {
        constexpr int Q = ROWS / 4;
        const int k = slot / Q, rq = slot - k * Q;
        int64_t row = row0 + rq * 4;
        row = row <= ld - 4 ? row : ld - 4;
        ptr[it] = p + row;
        kk[it] = k;
      }
    }
  }
  // full slice: every k in [k0, k0+32) is inside the problem
  __device__ __forceinline__ void load_full(int64_t k0, float4 (&regs)[PT]) const {
#pragma unroll
    for (int it = 0; it < PT; ++it) {
      if (LAY == KC)
        regs[it] = *reinterpret_cast<const float4*>(ptr[it] + k0);
      else
        regs[it] = *reinterpret_cast<const float4*>(ptr[it] + (k0 + kk[it]) * ld);
    }
  }
  // last, partial slice: clamp the address, zero what lies at k >= kend
  __device__ __forceinline__ void load_tail(int64_t k0, int64_t kend, float4 (&regs)[PT]) const {
#pragma unroll
    for (int it = 0; it < PT; ++it) {
      const int64_t k = k0 + kk[it];
      float4 v;
      if (LAY == KC) {
        const int64_t kc = k <= ld - 4 ? k : ld - 4;
        v = *reinterpret_cast<const float4*>(ptr[it] + (kc - kk[it]));
        v.x = k + 0 < kend ? v.x : 0.f;
        v.y = k + 1 < kend ? v.y : 0.f;
        v.z = k + 2 < kend ? v.z : 0.f;
        v.w = k + 3 < kend ? v.w : 0.f;
      } else {
        const int64_t kc = k < kend ? k : kend - 1;
        v = *reinterpret_cast<const float4*>(ptr[it] + kc * ld);
        if (k >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      regs[it] = v;
    }
  }
};

template <int BM, int BN, int WM, int ALAY, int BLAY, int EPI, bool DBUF>
__global__ __launch_bounds__(GEMM_TPB) void gemm_fast_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ B, int64_t ldb,
                                                             const float* __restrict__ bias, float* __restrict__ C,
                                                             int64_t ldc, int64_t M, int64_t N, int64_t K,
                                                             int64_t kchunk, double* __restrict__ colstats,
                                                             int xcd_order) {
  constexpr int WN = 4 / WM;
  constexpr int WCOLS = BN / WN;
  constexpr int NT = WCOLS / 32;
  constexpr int AF = Tile<BM, ALAY>::FLOATS, BF = Tile<BN, BLAY>::FLOATS;
  static_assert(BM == 32 * WM && NT >= 1, "tile shape");
  // DBUF: two LDS buffers, one barrier per slice (deep K).  !DBUF: one buffer, two barriers, twice the
  // workgroups per CU (short K, HBM-bound layers).
  __shared__ __attribute__((aligned(16))) float lds[(DBUF ? 2 : 1) * (AF + BF)];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int i = lane & 31, h = lane >> 5;
  int64_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (EPI == EPI_ATOMIC && xcd_order) {
    // one-dimensional launch of the split-K form: workgroup ids w, w + 8, w + 16, ... (one XCD, dispatched back to back)
    // walk the output tiles of ONE K chunk, so the chunk's operand rows are read from HBM once and from that XCD's L2 by
    // the other tiles (round-robin ids spread the tiles of a chunk over all eight XCDs, each fetching the rows again)
    const int64_t tiles_m = (M + BM - 1) / BM, tpc = tiles_m * ((N + BN - 1) / BN);
    const int64_t seq = bx >> 3;
    bz = (seq / tpc) * 8 + (bx & 7);
    const int64_t t = seq % tpc;
    by = t / tiles_m;
    bx = t - by * tiles_m;
    if (bz * kchunk >= K) return;
  }
  const int64_t m0 = bx * BM, n0 = by * BN;
  const int64_t kbeg = bz * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;

  f32x16 acc[NT];
  acc_init<NT>(acc, (EPI == EPI_STORE && bz == 0) ? bias : nullptr, n0 + wn * WCOLS, N, i);

  FastLoader<BM, ALAY> la;
  FastLoader<BN, BLAY> lb;
  la.init(A, lda, m0, M);
  lb.init(B, ldb, n0, N);
  float4 ra[Tile<BM, ALAY>::PER_THREAD], rb[Tile<BN, BLAY>::PER_THREAD];

  if (kbeg < kend) {
    if (kbeg + BK <= kend) {
      la.load_full(kbeg, ra);
      lb.load_full(kbeg, rb);
    } else {
      la.load_tail(kbeg, kend, ra);
      lb.load_tail(kbeg, kend, rb);
    }
    tile_store<BM, ALAY>(lds, ra);
    tile_store<BN, BLAY>(lds + AF, rb);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const float* As = lds + buf * (AF + BF);
    const float* Bs = As + AF;
    const int64_t kn = k0 + BK;
    if (kn < kend) {
      if (kn + BK <= kend) {
        la.load_full(kn, ra);
        lb.load_full(kn, rb);
      } else {
        la.load_tail(kn, kend, ra);
        lb.load_tail(kn, kend, rb);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a4 = frag_read<BM, ALAY>(As, wm * 32 + i, q, h);
      float4 b4[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b4[t] = frag_read<BN, BLAY>(Bs, wn * WCOLS + t * 32 + i, q, h);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[t].w, acc[t], 0, 0, 0);
      }
    }
    if (kn < kend) {
      if (!DBUF) __syncthreads();
      float* An = lds + (DBUF ? (buf ^ 1) : 0) * (AF + BF);
      tile_store<BM, ALAY>(An, ra);
      tile_store<BN, BLAY>(An + AF, rb);
    }
    __syncthreads();
    if (DBUF) buf ^= 1;
  }
  gemm_epilogue<BM, BN, WM, EPI, NT>(acc, lds, bias, C, ldc, M, N, m0, n0, wm, wn, i, h, colstats);
}


// ------------------------------------------------------------------ bf16 MFMA variant of Y = A W^T (BASELINE configs 3 / 5)
// Operands stay fp32 in HBM; a K slice is staged through registers exactly like gemm_fast_kernel, rounded to bf16
// (v_cvt_pk_bf16_f32, round-to-nearest-even) ONCE per element on its way into LDS, and multiplied with
// v_mfma_f32_32x32x16_bf16 into fp32 accumulators (bias = initial accumulator value, BatchNorm statistics of the
// fp32 result in the epilogue, as in the fp32 kernels).  LDS rows hold 32 bf16 + 8 of padding (80 bytes): a lane's
// fragment for k-step s is ONE 16-byte read at chunk 2s+h (k = 16s + 8h .. +7, the instruction's own operand
// layout -- no K permutation needed), conflict-free across 16 lanes at the 20-bank row stride.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
constexpr int BF_LD = BK + 8;  // bf16 elements per LDS row

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

// F16 = false: bf16 image, true: fp16 image (same 2-byte layout; only the rounding and the MFMA opcode differ)
template <int ROWS, bool F16 = false>
__device__ __forceinline__ void tile_store_bf16(__bf16* __restrict__ lds, const float4 (&regs)[Tile<ROWS, KC>::PER_THREAD]) {
#pragma unroll
  for (int it = 0; it < Tile<ROWS, KC>::PER_THREAD; ++it) {
    const int slot = threadIdx.x + it * GEMM_TPB;
    const int r = slot >> 3, kq = slot & 7;
    if (F16) {
      f16x4 v;
      v[0] = (_Float16)regs[it].x;
      v[1] = (_Float16)regs[it].y;
      v[2] = (_Float16)regs[it].z;
      v[3] = (_Float16)regs[it].w;
      *reinterpret_cast<f16x4*>(lds + r * BF_LD + kq * 4) = v;
    } else {
      bf16x4 v;
      v[0] = (__bf16)regs[it].x;
      v[1] = (__bf16)regs[it].y;
      v[2] = (__bf16)regs[it].z;
      v[3] = (__bf16)regs[it].w;
      *reinterpret_cast<bf16x4*>(lds + r * BF_LD + kq * 4) = v;
    }
  }
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const __bf16* a, const __bf16* b, f32x16 c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(a), *reinterpret_cast<const f16x8*>(b), c, 0,
                                                  0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(a), *reinterpret_cast<const bf16x8*>(b), c,
                                                 0, 0, 0);
}

template <int BM, int BN, int WM, bool F16 = false>
__global__ __launch_bounds__(GEMM_TPB) void gemm_bf16_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ B, int64_t ldb,
                                                             const float* __restrict__ bias, float* __restrict__ C,
                                                             int64_t ldc, int64_t M, int64_t N, int64_t K,
                                                             double* __restrict__ colstats) {
  constexpr int WN = 4 / WM;
  constexpr int WCOLS = BN / WN;
  constexpr int NT = WCOLS / 32;
  constexpr int AE = BM * BF_LD, BE = BN * BF_LD;  // bf16 elements per tile
  static_assert(BM == 32 * WM && NT >= 1, "tile shape");
  static_assert((AE + BE) * 2 * 2 >= WM * BN * 2 * 8, "epilogue scratch fits");
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * (AE + BE)];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int i = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;

  f32x16 acc[NT];
  acc_init<NT>(acc, bias, n0 + wn * WCOLS, N, i);

  FastLoader<BM, KC> la;
  FastLoader<BN, KC> lb;
  la.init(A, lda, m0, M);
  lb.init(B, ldb, n0, N);
  float4 ra[Tile<BM, KC>::PER_THREAD], rb[Tile<BN, KC>::PER_THREAD];
  if (BK <= K) {
    la.load_full(0, ra);
    lb.load_full(0, rb);
  } else {
    la.load_tail(0, K, ra);
    lb.load_tail(0, K, rb);
  }
  tile_store_bf16<BM, F16>(lds, ra);
  tile_store_bf16<BN, F16>(lds + AE, rb);
  __syncthreads();
  int buf = 0;
  for (int64_t k0 = 0; k0 < K; k0 += BK) {
    const __bf16* As = lds + buf * (AE + BE);
    const __bf16* Bs = As + AE;
    const int64_t kn = k0 + BK;
    if (kn < K) {  // next slice in flight while this one is multiplied
      if (kn + BK <= K) {
        la.load_full(kn, ra);
        lb.load_full(kn, rb);
      } else {
        la.load_tail(kn, K, ra);
        lb.load_tail(kn, K, rb);
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(As + (wm * 32 + i) * BF_LD + (2 * st + h) * 8);
      bf16x8 b8[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        b8[t] = *reinterpret_cast<const bf16x8*>(Bs + (wn * WCOLS + t * 32 + i) * BF_LD + (2 * st + h) * 8);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = mfma16<F16>(reinterpret_cast<const __bf16*>(&a8), reinterpret_cast<const __bf16*>(&b8[t]), acc[t]);
    }
    if (kn < K) {
      __bf16* An = lds + (buf ^ 1) * (AE + BE);
      tile_store_bf16<BM, F16>(An, ra);
      tile_store_bf16<BN, F16>(An + AE, rb);
    }
    __syncthreads();
    buf ^= 1;
  }
  gemm_epilogue<BM, BN, WM, EPI_STORE, NT>(acc, reinterpret_cast<float*>(lds), bias, C, ldc, M, N, m0, n0, wm, wn, i, h,
                                           colstats);
}

// bf16 weight gradient: C[M x N] += A^T B with both operands stored contraction-major (A: (k, r) at p[k*lda + r], i.e.
// dW += dY^T X with the sample index as k).  A thread's float4 holds 4 consecutive rows r at ONE k, so the rounded
// values are scattered into the same [row][k] bf16 LDS image the NT kernel uses (4 two-byte writes per float4); the
// multiply loop and the fragment reads are then identical.  The reduction over k is split across blockIdx.z and
// accumulated with fp32 atomics like the fp32 kernel.
template <int ROWS>
__device__ __forceinline__ void tile_store_bf16_mc(__bf16* __restrict__ lds,
                                                   const float4 (&regs)[Tile<ROWS, MC>::PER_THREAD]) {
  constexpr int Q = ROWS / 4;
#pragma unroll
  for (int it = 0; it < Tile<ROWS, MC>::PER_THREAD; ++it) {
    const int slot = threadIdx.x + it * GEMM_TPB;
    const int kk = slot / Q, rq = slot - kk * Q;
    __bf16* dst = lds + (4 * rq) * BF_LD + kk;
    dst[0] = (__bf16)regs[it].x;
    dst[BF_LD] = (__bf16)regs[it].y;
    dst[2 * BF_LD] = (__bf16)regs[it].z;
    dst[3 * BF_LD] = (__bf16)regs[it].w;
  }
}

template <int BM, int BN, int WM>
__global__ __launch_bounds__(GEMM_TPB) void gemm_bf16_tn_kernel(const float* __restrict__ A, int64_t lda,
                                                                const float* __restrict__ B, int64_t ldb,
                                                                float* __restrict__ C, int64_t ldc, int64_t M, int64_t N,
                                                                int64_t K, int64_t kchunk) {
  constexpr int WN = 4 / WM;
  constexpr int WCOLS = BN / WN;
  constexpr int NT = WCOLS / 32;
  constexpr int AE = BM * BF_LD, BE = BN * BF_LD;
  static_assert(BM == 32 * WM && NT >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * (AE + BE)];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int i = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;

  f32x16 acc[NT];
  acc_init<NT>(acc, nullptr, n0 + wn * WCOLS, N, i);

  FastLoader<BM, MC> la;
  FastLoader<BN, MC> lb;
  la.init(A, lda, m0, M);
  lb.init(B, ldb, n0, N);
  float4 ra[Tile<BM, MC>::PER_THREAD], rb[Tile<BN, MC>::PER_THREAD];
  if (kbeg < kend) {
    if (kbeg + BK <= kend) {
      la.load_full(kbeg, ra);
      lb.load_full(kbeg, rb);
    } else {
      la.load_tail(kbeg, kend, ra);
      lb.load_tail(kbeg, kend, rb);
    }
    tile_store_bf16_mc<BM>(lds, ra);
    tile_store_bf16_mc<BN>(lds + AE, rb);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const __bf16* As = lds + buf * (AE + BE);
    const __bf16* Bs = As + AE;
    const int64_t kn = k0 + BK;
    if (kn < kend) {
      if (kn + BK <= kend) {
        la.load_full(kn, ra);
        lb.load_full(kn, rb);
      } else {
        la.load_tail(kn, kend, ra);
        lb.load_tail(kn, kend, rb);
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(As + (wm * 32 + i) * BF_LD + (2 * st + h) * 8);
      bf16x8 b8[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        b8[t] = *reinterpret_cast<const bf16x8*>(Bs + (wn * WCOLS + t * 32 + i) * BF_LD + (2 * st + h) * 8);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8[t], acc[t], 0, 0, 0);
    }
    if (kn < kend) {
      __bf16* An = lds + (buf ^ 1) * (AE + BE);
      tile_store_bf16_mc<BM>(An, ra);
      tile_store_bf16_mc<BN>(An + AE, rb);
    }
    __syncthreads();
    buf ^= 1;
  }
  gemm_epilogue<BM, BN, WM, EPI_ATOMIC, NT>(acc, reinterpret_cast<float*>(lds), nullptr, C, ldc, M, N, m0, n0, wm, wn, i, h,
                                            nullptr);
}

// ------------------------------------------------------------------ LDS-DMA pipeline (Y = A W^T, both operands KC)
// 256 x BN tile, 8 waves (one 32-row band each), BK = 32.  Tiles reach LDS with global_load_lds_dwordx4
// (no VGPR staging): one wave instruction copies 8 rows x 128 B into a linear 1 KiB span, so the tile rows are
// NOT padded; bank conflicts are avoided by swizzling on the SOURCE side instead -- LDS chunk c (16 B) of row r
// holds global chunk c ^ ((r >> 1) & 7), and a fragment read of global chunk g = 2q + h goes to LDS chunk
// g ^ ((r >> 1) & 7): each 16-lane group of ds_read_b128 then covers all 16 distinct 16-B slots.
// Three stages: the DMA of slice u+2 is issued right after the barrier of iteration u and has two full
// compute phases to land; waits are counted (vmcnt(NA+NB) leaves the youngest slice in flight) and the barrier
// is a raw s_barrier, so nothing drains the queue.  A K remainder (< 32) is loaded once through registers
// (zero-filled) into stage 0 before the pipeline starts; summation order over K slices is immaterial.
constexpr int GL_TPB = 512;
constexpr int GL_BM = 256;

__device__ __forceinline__ void glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// one K group (q) of fragments for lane (i, h): A band row and NT B rows, 16 bytes each, swizzled chunk
template <int NT>
__device__ __forceinline__ void lds_read_frag(uint32_t a_row_b, uint32_t b_row_b, int q, int h, int swz, f32x4& fa,
                                              f32x4 (&fb)[NT]) {
  const uint32_t ch = 16u * (uint32_t)((2 * q + h) ^ swz);
  asm volatile("ds_read_b128 %0, %1" : "=v"(fa) : "v"(a_row_b + ch) : "memory");
#pragma unroll
  for (int t = 0; t < NT; ++t)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[t]) : "v"(b_row_b + ch), "n"(t * 32 * BK * 4) : "memory");
}

template <int BN>
__global__ __launch_bounds__(GL_TPB) void gemm_glds_kernel(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb,
                                                           const float* __restrict__ bias, float* __restrict__ C,
                                                           int64_t ldc, int64_t M, int64_t N, int64_t K,
                                                           double* __restrict__ colstats) {
  constexpr int NT = BN / 32;
  constexpr int AF = GL_BM * BK, BF = BN * BK, STAGE = AF + BF;
  constexpr int NA = GL_BM / 8 / 8;                 // LDS-DMA instructions per wave per slice for A (4)
  constexpr int NB = BN >= 64 ? BN / 64 : 1;        // ... and for B (waves 4-7 repeat rows when BN = 32)
  __shared__ __attribute__((aligned(16))) float lds[3 * STAGE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int swz = (i >> 1) & 7;
  const int64_t m0 = (int64_t)blockIdx.x * GL_BM, n0 = (int64_t)blockIdx.y * BN;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const uint32_t a_off = (uint32_t)((wave * 32 + i) * BK * 4);        // this lane's A row inside a stage (bytes)
  const uint32_t b_off = (uint32_t)((AF + i * BK) * 4);               // B row i; tile t adds an immediate

  f32x16 acc[NT];
  acc_init<NT>(acc, bias, n0, N, i);

  // ---- per-lane DMA sources: row group g covers tile rows 8g .. 8g+7, lane -> (row 8g + lane/8, chunk lane%8)
  const int lr = lane >> 3, lc = lane & 7;
  const float* a_src[NA];
  const float* b_src[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int r = 8 * (wave * NA + j) + lr;
    int64_t row = m0 + r;
    row = row < M ? row : M - 1;
    a_src[j] = A + row * lda + 4 * (lc ^ ((r >> 1) & 7));
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int g = BN >= 64 ? wave * NB + j : (wave & 3);
    const int r = 8 * g + lr;
    int64_t row = n0 + r;
    row = row < N ? row : N - 1;
    b_src[j] = B + row * ldb + 4 * (lc ^ ((r >> 1) & 7));
  }
  const int has_tail = (K % BK) != 0;
  const int64_t nfull = K / BK;
  const int T = (int)nfull + has_tail;

  auto issue = [&](int u) {  // LDS-DMA of full slice u (ring position u % 3)
    float* st = lds + (u % 3) * STAGE;
    const int64_t k0 = (int64_t)(u - has_tail) * BK;
#pragma unroll
    for (int j = 0; j < NA; ++j) glds16(a_src[j] + k0, st + (8 * (wave * NA + j)) * BK);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int g = BN >= 64 ? wave * NB + j : (wave & 3);
      glds16(b_src[j] + k0, st + AF + (8 * g) * BK);
    }
  };

  if (has_tail) {  // K remainder through registers, zero filled, same swizzled image, stage 0
    const int64_t kt = nfull * BK;
#pragma unroll
    for (int it = 0; it < ((GL_BM + BN) * 8 + GL_TPB - 1) / GL_TPB; ++it) {
      const int slot = threadIdx.x + it * GL_TPB;
      if (slot >= (GL_BM + BN) * 8) break;
      const bool isA = slot < GL_BM * 8;
      const int sl = isA ? slot : slot - GL_BM * 8;
      const int r = sl >> 3, kq = sl & 7;
      const float* p = isA ? A : B;
      const int64_t ld = isA ? lda : ldb;
      int64_t row = (isA ? m0 : n0) + r;
      const int64_t lim = isA ? M : N;
      row = row < lim ? row : lim - 1;
      const int64_t k = kt + kq * 4;
      const int64_t kc = k <= ld - 4 ? k : ld - 4;
      float4 v = *reinterpret_cast<const float4*>(p + row * ld + kc);
      v.x = k + 0 < K ? v.x : 0.f;
      v.y = k + 1 < K ? v.y : 0.f;
      v.z = k + 2 < K ? v.z : 0.f;
      v.w = k + 3 < K ? v.w : 0.f;
      float* dst = lds + (isA ? 0 : AF) + r * BK + 4 * (kq ^ ((r >> 1) & 7));
      *reinterpret_cast<float4*>(dst) = v;
    }
  }
  int next = has_tail;
  for (; next < T && next < 2; ++next) issue(next);

  for (int u = 0; u < T; ++u) {
    if (u >= has_tail) {
      if (u + 1 < next)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the register-path ds_writes of the tail slice
    }
    __builtin_amdgcn_s_barrier();
    if (next < T) {
      issue(next);
      ++next;
    }
    // Fragment reads are inline asm on purpose: hipcc orders every compiler-visible ds_read behind ALL pending
    // LDS-DMA (s_waitcnt vmcnt(0)), which would drain the two slices in flight.  The data dependence that matters
    // (slice u landed) is the counted vmcnt + barrier above.  Reads of K group q+1 are issued before the MFMAs of
    // group q and waited for with a counted lgkmcnt.
    const uint32_t stage_b = lds_base + (uint32_t)((u % 3) * STAGE * 4);
    f32x4 fa[2], fb[2][NT];
    lds_read_frag<NT>(stage_b + a_off, stage_b + b_off, 0, h, swz, fa[0], fb[0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < 3) {
        lds_read_frag<NT>(stage_b + a_off, stage_b + b_off, q + 1, h, swz, fa[(q + 1) & 1], fb[(q + 1) & 1]);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT + 1) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1].x, fb[q & 1][t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1].y, fb[q & 1][t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1].z, fb[q & 1][t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1].w, fb[q & 1][t].w, acc[t], 0, 0, 0);
      }
    }
  }
  __syncthreads();

  // ---- epilogue (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); the statistics of the
  // two 128-row halves go to two partial rows so that the ccn_stats_rows(M) convention holds
  double* stat_lds = reinterpret_cast<double*>(lds);  // [8][BN][2]
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ncol = t * 32 + i;
    const int64_t n = n0 + ncol;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < M && n < N) {
        const float v = acc[t][r];
        C[m * ldc + n] = v;
        if (colstats != nullptr) {
          s1 += (double)v;
          s2 += (double)v * (double)v;
        }
      }
    }
    if (colstats != nullptr) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        stat_lds[(wave * BN + ncol) * 2] = s1;
        stat_lds[(wave * BN + ncol) * 2 + 1] = s2;
      }
    }
  }
  if (colstats != nullptr) {
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * BN; e += GL_TPB) {
      const int half = e / BN, c = e - half * BN;
      const int64_t n = n0 + c;
      const int64_t prow = (int64_t)blockIdx.x * 2 + half;
      if (n < N && prow * 128 < M) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s1 += stat_lds[((half * 4 + w) * BN + c) * 2];
          s2 += stat_lds[((half * 4 + w) * BN + c) * 2 + 1];
        }
        double* dst = colstats + prow * 2 * N;
        dst[n] = s1;
        dst[N + n] = s2;
      }
    }
  }
}

template <int BN>
int launch_glds(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                int64_t M, int64_t N, int64_t K, double* colstats, hipStream_t s) {
  const int64_t gm = (M + GL_BM - 1) / GL_BM, gn = (N + BN - 1) / BN;
  if (gm > 2147483647LL || gn > 65535) {
    ccn_set_error("gemm: grid too large");
    return CCN_ERR_ARG;
  }
  hipLaunchKernelGGL((gemm_glds_kernel<BN>), dim3((unsigned)gm, (unsigned)gn), dim3(GL_TPB), 0, s, A, lda, W, ldw,
                     bias, Y, ldy, M, N, K, colstats);
  return CCN_OK;
}


// ------------------------------------------------------------------ persistent form of the LDS-DMA kernel
// One workgroup per CU walks the output tiles (K % 32 == 0).  The K slices of consecutive tiles form ONE
// stream through the 3-stage ring, so the DMA of the next tile's first two slices is already in flight while the
// current tile is finished, and the epilogue's global stores (issued and forgotten) drain under the next tile's
// MFMAs: with a single 8-wave workgroup per CU nothing else could hide the pipeline fill or the 128 KiB of
// stores per tile (measured on the one-tile-per-workgroup kernel: 75 TFLOP/s at K=128 vs 132 at K=4096).
// vmcnt retires loads, stores and LDS-DMA in issue order, so the counted waits are arranged as follows: before
// the stores every DMA issued so far is waited for (it is at least one compute phase old) and remembered as
// landed; the next two slices then need no wait, and later counted waits see the stores as the oldest entries.
// WAVES = 8: one workgroup per CU on 256-row tiles (above).  WAVES = 4 (round 3, widths <= 64): 128-row tiles, 74 KB of LDS, TWO
// independent workgroups per CU -- at N = K = 64 the 8-wave form is bound neither by its stores (+9 % without them) nor by
// the copies (+1 % without waiting for them) nor by HBM (48 %) or the matrix pipe (43 %), but by eight waves marching in step.
template <int BN, int STAGES, int WAVES = 8>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 1 : 2) void gemm_glds_persistent_kernel(const float* __restrict__ A, int64_t lda,
                                                                      const float* __restrict__ B, int64_t ldb,
                                                                      const float* __restrict__ bias,
                                                                      float* __restrict__ C, int64_t ldc, int64_t M,
                                                                      int64_t N, int64_t K, int64_t tiles, int64_t gn,
                                                                      int xcd_order, double* __restrict__ colstats) {
  constexpr int NT = BN / 32;
  constexpr int BM = WAVES * 32, TPB_ = WAVES * 64;
  constexpr int AF = BM * BK, BF = BN * BK, STAGE = AF + BF;
  constexpr int NA = BM / 8 / WAVES;                                   // copies per wave and slice: A (4)
  constexpr int NB = BN / 8 >= WAVES ? BN / 8 / WAVES : 1;             // ... B (fewer copies than waves: some waves duplicate)
  __shared__ __attribute__((aligned(16))) float lds[STAGES * STAGE];
  // per-wave BatchNorm partial sums of the finished tile ([8 waves][BN][2] fp32, sums over the wave's 32 rows).  A
  // table of its own instead of a recycled stage: no barrier is then needed before it is written, and it is read out
  // (fp64 across the 4 waves of each 128-row half) after the FIRST slice barrier of the next tile, so the statistics
  // add no workgroup barrier at all to the tile loop (two extra barriers per tile cost 13 % at K = 256).
  __shared__ float stat_part[WAVES * BN * 2];

  // (wave index as a scalar, copies addressed as scalar tile base + 32-bit lane offset: see the paired kernel below)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 31, h = lane >> 5;
  const int swz = (i >> 1) & 7;
  const int lr = lane >> 3, lc = lane & 7;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const uint32_t a_off = (uint32_t)((wave * 32 + i) * BK * 4);
  const uint32_t b_off = (uint32_t)((AF + i * BK) * 4);
  const int T = (int)(K / BK);

  // ---- issue cursor (runs two slices ahead of the compute cursor, across tile boundaries)
  uint32_t a_off32[NA], b_off32[NB];
  const char* a_tile = reinterpret_cast<const char*>(A);
  const char* b_tile = reinterpret_cast<const char*>(B);
  // Tile order.  With all 256 workgroups launched and gn dividing 32, the gn column tiles of one 256-row block go to
  // workgroups w, w + 8, ... of the SAME step: those ids sit on one XCD, so the row block is read from HBM once and from
  // that XCD's L2 by the other column tiles (measured on the split-bf16 twin of this kernel: 707 -> 379 MB read per
  // launch at K = N = 256).  Otherwise tiles are dealt round-robin.
  const int64_t gm_tiles = tiles / gn;
  constexpr int GRID = WAVES == 8 ? 256 : 512;       // workgroups of a full launch (GRID / 8 slots per XCD)
  const bool xcd_map = xcd_order && gridDim.x == GRID && gn <= GRID / 8 && (GRID / 8) % gn == 0;
  auto tile_of = [&](int64_t j) -> int64_t {   // (>= tiles: this workgroup has no tile in step j)
    if (xcd_map) {
      const int64_t slot = blockIdx.x >> 3;
      const int64_t m = j * (GRID / gn) + (slot / gn) * 8 + (blockIdx.x & 7);
      return m < gm_tiles ? m * gn + slot % gn : tiles;
    }
    return j * gridDim.x + blockIdx.x;
  };
  int64_t it_j = 0;
  int64_t it_tile = tile_of(0);
  int it_u = 0;
  int64_t gi = 0;  // slices issued so far
  auto issue_next = [&]() {
    if (it_tile >= tiles) return;
    if (it_u == 0) {
      const int64_t im0 = (it_tile / gn) * BM, in0 = (it_tile % gn) * BN;
      a_tile = reinterpret_cast<const char*>(A + im0 * lda);
      b_tile = reinterpret_cast<const char*>(B + in0 * ldb);
      const int64_t a_rows = M - im0, b_rows = N - in0;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int r = 8 * (wave * NA + j) + lr;
        const int64_t ra = r < a_rows ? r : a_rows - 1;
        a_off32[j] = (uint32_t)((ra * lda + 4 * (lc ^ ((r >> 1) & 7))) * 4);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int g = BN / 8 >= WAVES ? wave * NB + j : (wave % (BN / 8));
        const int r = 8 * g + lr;
        const int64_t rb = r < b_rows ? r : b_rows - 1;
        b_off32[j] = (uint32_t)((rb * ldb + 4 * (lc ^ ((r >> 1) & 7))) * 4);
      }
    }
    float* st = lds + (gi % STAGES) * STAGE;
    const int64_t k0 = (int64_t)it_u * BK;
    const char* const a_sl = a_tile + k0 * 4;
    const char* const b_sl = b_tile + k0 * 4;
#pragma unroll
    for (int j = 0; j < NA; ++j) glds16(reinterpret_cast<const float*>(a_sl + a_off32[j]), st + (8 * (wave * NA + j)) * BK);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int g = BN / 8 >= WAVES ? wave * NB + j : (wave % (BN / 8));
      glds16(reinterpret_cast<const float*>(b_sl + b_off32[j]), st + AF + (8 * g) * BK);
    }
    ++gi;
    if (++it_u == T) {
      it_u = 0;
      it_tile = tile_of(++it_j);
    }
  };
  for (int pre = 0; pre < STAGES - 1; ++pre) issue_next();  // the DMA runs STAGES-1 slices ahead of the compute

  int64_t stat_tile = -1;  // tile whose statistics wait in stat_part
  auto stats_readout = [&]() {
    constexpr int HALVES = WAVES / 4;      // 128-row partial rows per tile (ccn_stats_rows): four waves each
    const int64_t pm = stat_tile / gn, pn0 = (stat_tile % gn) * BN;
    for (int e = threadIdx.x; e < HALVES * BN; e += TPB_) {
      const int half = e / BN, c = e - half * BN;
      const int64_t n = pn0 + c;
      const int64_t prow = pm * HALVES + half;
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
    stat_tile = -1;
  };

  int64_t g = 0;       // slices computed so far
  int64_t landed = 0;  // slices [0, landed) are known to be in LDS for this wave
  for (int64_t j = 0, tile; (tile = tile_of(j)) < tiles; ++j) {
    const int64_t m0 = (tile / gn) * BM, n0 = (tile % gn) * BN;
    f32x16 acc[NT];
    acc_init<NT>(acc, bias, n0, N, i);  // bias (or 0) as the initial accumulator value

    for (int u = 0; u < T; ++u, ++g) {
      if (g >= landed) {
        if (g + 1 < gi)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      issue_next();
      if (u == 0 && stat_tile >= 0) stats_readout();
      const uint32_t stage_b = lds_base + (uint32_t)((g % STAGES) * STAGE * 4);
      f32x4 fa[2], fb[2][NT];
      lds_read_frag<NT>(stage_b + a_off, stage_b + b_off, 0, h, swz, fa[0], fb[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) {
          lds_read_frag<NT>(stage_b + a_off, stage_b + b_off, q + 1, h, swz, fa[(q + 1) & 1], fb[(q + 1) & 1]);
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT + 1) : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1].x, fb[q & 1][t].x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1].y, fb[q & 1][t].y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1].z, fb[q & 1][t].z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1].w, fb[q & 1][t].w, acc[t], 0, 0, 0);
        }
      }
    }

    // ---- tile epilogue.  Everything issued so far (<= 2 slices of the next tile) is waited for first.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed = gi;
    // (16-byte stores after a DPP quad transpose of the accumulators -- 16 instead of 64 store instructions per lane -- were
    // tried and measured 2-5 % SLOWER: the epilogue is bound by the write path, not by the number of store instructions.)
    // Interior tiles (all but the last tile row / column) store without per-element bounds tests: 16*NT stores per lane
    // with a compare, a branch and a 64-bit multiply-add each made the epilogue ~8.5k cycles per tile (measured 4.2 us
    // per tile whatever K), a quarter of a K = 128 tile.
    const bool interior = m0 + BM <= M && n0 + BN <= N;
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
    if (colstats != nullptr) {
      stat_tile = tile;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // table written before this wave reaches the next barrier
    }
  }
  if (stat_tile >= 0) {  // statistics of the last tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stats_readout();
  }
}

static std::atomic<bool> g_xcd_map{true};  // A-B hook (ccn_gemm_use_dma(3) = persistent kernel with round-robin tiles)

static std::atomic<int> g_pair_opt{0};      // A-B hook (ccn_gemm_pair_opt)

__device__ __forceinline__ float act_fwd(float z, int act, float slope) {
  if (act == CCN_ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == CCN_ACT_LEAKY) return z > 0.f ? z : z * slope;
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
  if (act == CCN_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == CCN_ACT_LEAKY) return z > 0.f ? 1.f : slope;
  return 1.f;
}
// BatchNorm + activation of one value, the expression of bn_act_fwd (ONE definition: the fused form below must give the same bits)
__device__ __forceinline__ float bn_act_value(float y, float sc, float sh, int act, float slope) {
  return act_fwd(y * sc + sh, act, slope);
}

// ------------------------------------------------------------------ two independent workgroups per CU (N > 64)
// The 8-wave persistent kernel above keeps ONE workgroup on a CU, so nothing runs while its waves store a finished tile
// (3.65 us of a 36 us tile at K = 256, 19 % at K = 128).  Here a CU holds TWO 4-wave workgroups on 128 x 128 tiles (each
// wave a 64 x 64 quadrant: one A and one B fragment read per four MFMAs instead of five per sixteen): they share nothing,
// drift apart, and one's epilogue, pipeline refill and barrier waits run under the other's MFMAs.  64 KB of LDS each:
// a two-stage LDS-DMA ring, the copy of slice g + 1 issued right after the barrier of slice g -- one slice of lead is
// enough because the partner workgroup's compute phase lies in between as well.  Every iteration waits vmcnt(0): the
// epilogue's stores are simply part of what is waited for (the partner computes meanwhile).
constexpr int PR_TPB = 256;
constexpr int PR_BM = 128, PR_BN = 128;

// ACC: Y += A W^T -- the accumulators start from the Y tile itself (64 loads per lane straight into the accumulator
// registers at the top of a tile, covered by the first slice's wait) instead of from the bias (ccn_gemm_nt_acc).
// XF: the A operand is the PRE-normalisation output of the previous layer and z = act(a * scale[k] + shift[k]) -- its
// BatchNorm + activation (ccn_bn_act_fwd) -- is applied to the fragments on their way from LDS to the MFMAs, so that the
// activation tensor is never written or read (ccn_gemm_nt_xf).  The per-channel table (K <= 1024, K % 32 == 0) sits in LDS;
// the transform of K group q + 1 runs on the VALU while the matrix pipe works on group q.
constexpr int XF_MAX_K = 1024;

// TAG: 1 = the 128-wide part of a product that is split into this launch + a 64-wide remainder (gemm_nt_impl): the same
// code under its own name, so that profiles list the single-launch products (what bench.py samples) separately.
// STAMP: diagnostic build (ccn_gemm_pair_debug): per workgroup, every wave adds up where its cycles go (s_memtime stamps)
// and writes 16 words to the debug buffer.  Never launched unless a debug buffer is set.
__device__ unsigned long long* g_pair_dbg_dev = nullptr;
#define PR_STAMP(t)                                                                         \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)

// RED (round 4): this product is the DATA GRADIENT dZ = dY W of a layer whose input was the deferred (pre-normalisation)
// output y of the previous layer: the column sums that layer's BatchNorm backward needs -- sum(g) and sum(g * xhat) with
// g = dZ * act'(y * scale + shift), xhat = (y - mean) * rstd -- are taken in THIS epilogue, from the dZ tile in the
// accumulators and the matching y tile (16 loads per 32 x 32 block, issued in front of the block's stores and waited for
// behind them), into the same partial-row table the forward statistics use.  ccn_bn_act_bwd_reduce's pass over (dZ, y)
// -- 8 bytes per element, 2.3 ms per KITTI step -- then does not run for that layer (ccn_gemm_nt_red).
// SPLIT (round 5): the instantiation that also walks a split tail round (see full_tiles / split_s below).  The launches without
// a tail round take the instantiation without it: carried along as run-time state the tail bookkeeping cost the plain product
// 2 % (117.5 against 120.5 TFLOP/s in the KITTI step, one box, profiles/r05_split_tails.txt).
template <bool ACC, bool XF, int TAG = 0, bool STAMP = false, bool RED = false, bool SPLIT = false>
__global__ __launch_bounds__(PR_TPB, 2) void gemm_glds_pair_kernel(const float* __restrict__ A, int64_t lda,
                                                                   const float* __restrict__ B, int64_t ldb,
                                                                   const float* __restrict__ bias, float* __restrict__ C,
                                                                   int64_t ldc, int64_t M, int64_t N, int64_t K,
                                                                   int64_t tiles, int64_t gn, int xcd_order,
                                                                   double* __restrict__ colstats, int64_t a_extent,
                                                                   const float* __restrict__ xf_scale,
                                                                   const float* __restrict__ xf_shift, int xf_act,
                                                                   float xf_slope, int64_t red_ldy, int64_t full_tiles,
                                                                   int split_s, float* __restrict__ split_ws) {
  // full_tiles / split_s / split_ws (round 5, "split tails"): tiles [0, full_tiles) are walked whole by the persistent loop, in
  // rounds of gridDim.x; the tiles [full_tiles, tiles) of the last, partly filled round are cut into split_s parts along K, one
  // part per workgroup, so that the round takes 1 / split_s of a tile time instead of a whole one with most CUs idle
  // (10 550 x 1024 -> 1024: 664 tiles over 512 slots = 2 rounds for 1.3 rounds of work).  A part writes its accumulators to
  // split_ws; the LAST part of a tile to arrive (a counter per tile in front of the partials, left at zero again) adds the
  // parts in part order -- its own included, so the sum does not depend on who came last -- and runs the tile's epilogue.
  // split_s <= 1: no tail round (full_tiles == tiles).
  // a_extent: floats readable from the start of an A row (= lda, or K when rows overlap: ccn_conv_rows_nt)
  // RED: xf_scale = the previous layer's 4 x N table (scale | shift | mean | rstd, rows N floats apart), xf_shift = its
  // pre-normalisation output y (leading dimension red_ldy), xf_act / xf_slope = its activation
  const float red_neg = xf_act == CCN_ACT_RELU ? 0.f : (xf_act == CCN_ACT_LEAKY ? xf_slope : 1.f);   // act'(z <= 0)
  constexpr int AF = PR_BM * BK, BF = PR_BN * BK, STAGE = AF + BF;
  constexpr int NC = 4;  // LDS-DMA copies (8 rows x 128 B) per wave, slice and operand
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
  __shared__ float stat_part[2 * PR_BN * 2];  // [wm][column][sum, sum of squares] of the finished tile
  __shared__ __attribute__((aligned(16))) float xf_tab[XF ? 2 * XF_MAX_K : 4];   // [scale | shift] of the A channels
  if (XF) {
    for (int k = threadIdx.x; k < K; k += PR_TPB) {
      xf_tab[k] = xf_scale[k];
      xf_tab[XF_MAX_K + k] = xf_shift[k];
    }
    __syncthreads();
  }
  const uint32_t xf_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)xf_tab;
  const float xf_neg = xf_act == CCN_ACT_RELU ? 0.f : (xf_act == CCN_ACT_LEAKY ? xf_slope : 1.f);   // multiplier of v <= 0

  // (the wave index as a SCALAR: LDS-DMA destinations and tile bases then stay in SGPRs -- three vector instructions fewer per copy)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave & 1, wn = wave >> 1;
  const int i = lane & 31, h = lane >> 5;
  const int swz = (i >> 1) & 7;
  const int lr = lane >> 3, lc = lane & 7;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const uint32_t a_off = (uint32_t)((wm * 64 + i) * BK * 4);            // A fragment row of block ab = 0 (ab adds 32 rows)
  const uint32_t b_off = (uint32_t)((AF + (wn * 64 + i) * BK) * 4);     // B fragment row of block t = 0
  const int T = (int)(K / BK);                      // full slices (LDS-DMA)
  const int has_tail = (K % BK) != 0;               // K remainder: one more slice per tile, staged through registers
  const int TT = T + has_tail;

  // (tile arithmetic in 32 bits -- a 64-bit division is ~130 dependent scalar instructions and there were eight per tile;
  // measured neutral on speed, 12 VGPRs and 14 spilled SGPRs fewer; the launcher checks tiles < 2^31)
  const uint32_t gnu = (uint32_t)gn;
  const uint32_t gm_tiles = (uint32_t)(SPLIT ? full_tiles : tiles) / gnu;     // (a tail round starts at a whole tile row: 512 % gnu == 0 there)
  const bool xcd_map = xcd_order && gridDim.x == 512 && gnu <= 64 && 64 % gnu == 0;
  const uint32_t slot = blockIdx.x >> 3;
  const uint32_t rows_per_step = 512u / gnu, slot_row = (slot / gnu) * 8 + (blockIdx.x & 7), slot_col = slot % gnu;
  auto tile_of = [&](int64_t j) -> int64_t {   // as in the 8-wave kernel, with 64 workgroup slots per XCD (>= tiles: none)
    if (xcd_map) {
      const uint32_t m = (uint32_t)j * rows_per_step + slot_row;
      return m < gm_tiles ? (int64_t)(m * gnu + slot_col) : tiles;
    }
    const int64_t t = j * gridDim.x + blockIdx.x;
    if (!SPLIT) return t;
    return t < full_tiles ? t : tiles;
  };
  // this workgroup's part of the tail round: tile full_tiles + (id % rem), slices [tail_u0, tail_u1)
  const int64_t tail_rem = tiles - full_tiles;
  const bool has_tail_item = SPLIT && split_s > 1 && (int64_t)blockIdx.x < tail_rem * split_s;
  const int tail_idx = has_tail_item ? (int)((int64_t)blockIdx.x % tail_rem) : 0;
  const int tail_part = has_tail_item ? (int)((int64_t)blockIdx.x / tail_rem) : 0;
  const int tail_u0 = tail_part * TT / (split_s > 1 ? split_s : 1), tail_u1 = (tail_part + 1) * TT / (split_s > 1 ? split_s : 1);
  __shared__ int split_flag;
  auto tile_row = [&](int64_t t) -> int64_t { return (int64_t)((uint32_t)t / gnu); };
  auto tile_col = [&](int64_t t) -> int64_t { return (int64_t)((uint32_t)t % gnu); };

  // ---- issue cursor: one slice ahead of the compute cursor, across tile boundaries
  // a copy's source = tile base (scalar, 64 bit) + this lane's byte offset inside the tile (32 bit: row * ld + swizzled chunk;
  // a tile spans 128 rows, so the offset fits whatever the matrix size)
  uint32_t a_off32[NC], b_off32[NC];
  const char* a_tile = reinterpret_cast<const char*>(A);
  const char* b_tile = reinterpret_cast<const char*>(B);
  int64_t it_j = 0, it_tile = tile_of(0), gi = 0;
  int64_t im0 = 0, in0 = 0;
  int it_u = 0, it_uend = TT;
  bool it_first = true, it_tail_left = has_tail_item;
  if (SPLIT && it_tile >= tiles && it_tail_left) {     // no whole tile for this workgroup: the tail part is its first item
    it_tail_left = false;
    it_tile = full_tiles + tail_idx;
    it_u = tail_u0;
    it_uend = tail_u1;
  }
  auto issue_next = [&]() {
    if (it_tile >= tiles) return;
    if (SPLIT ? it_first : it_u == 0) {
      it_first = false;
      im0 = tile_row(it_tile) * PR_BM;
      in0 = tile_col(it_tile) * PR_BN;
      a_tile = reinterpret_cast<const char*>(A + im0 * lda);
      b_tile = reinterpret_cast<const char*>(B + in0 * ldb);
      const int64_t a_rows = M - im0, b_rows = N - in0;       // rows of the matrix from the tile's first row on (>= 1)
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const int r = 8 * (wave * NC + q) + lr;
        const int64_t ra = r < a_rows ? r : a_rows - 1, rb = r < b_rows ? r : b_rows - 1;
        a_off32[q] = (uint32_t)((ra * lda + 4 * (lc ^ ((r >> 1) & 7))) * 4);
        b_off32[q] = (uint32_t)((rb * ldb + 4 * (lc ^ ((r >> 1) & 7))) * 4);
      }
    }
    float* st = lds + (gi & 1) * STAGE;
    const int64_t k0 = (int64_t)it_u * BK;
    if (it_u < T) {
      const char* const a_sl = a_tile + k0 * 4;
      const char* const b_sl = b_tile + k0 * 4;
#pragma unroll
      for (int q = 0; q < NC; ++q)
        glds16(reinterpret_cast<const float*>(a_sl + a_off32[q]), st + (8 * (wave * NC + q)) * BK);
#pragma unroll
      for (int q = 0; q < NC; ++q)
        glds16(reinterpret_cast<const float*>(b_sl + b_off32[q]), st + AF + (8 * (wave * NC + q)) * BK);
    } else {
      // K remainder (< 32 columns, e.g. the 3 xyz columns of a 259-wide concat): both operands through registers, zero
      // filled beyond K, into the same swizzled image.  (Nothing else is in flight here: every iteration waits vmcnt(0).)
#pragma unroll
      for (int it = 0; it < 2 * PR_BM * 8 / PR_TPB; ++it) {
        const int slot = threadIdx.x + it * PR_TPB;
        const bool isA = slot < PR_BM * 8;
        const int sl = isA ? slot : slot - PR_BM * 8;
        const int r = sl >> 3, kq = sl & 7;
        const float* p = isA ? A : B;
        const int64_t ld = isA ? lda : ldb;
        int64_t row = (isA ? im0 : in0) + r;
        const int64_t lim = isA ? M : N;
        row = row < lim ? row : lim - 1;
        const int64_t k = k0 + kq * 4;
        const int64_t ext = isA ? a_extent : ldb;
        const int64_t kc = k <= ext - 4 ? k : ext - 4;    // (k > extent - 4 implies k >= K: everything is zeroed below)
        float4 v = *reinterpret_cast<const float4*>(p + row * ld + kc);
        v.x = k + 0 < K ? v.x : 0.f;
        v.y = k + 1 < K ? v.y : 0.f;
        v.z = k + 2 < K ? v.z : 0.f;
        v.w = k + 3 < K ? v.w : 0.f;
        *reinterpret_cast<float4*>(st + (isA ? 0 : AF) + r * BK + 4 * (kq ^ ((r >> 1) & 7))) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // written before this wave reaches the next barrier
    }
    ++gi;
    if (++it_u == (SPLIT ? it_uend : TT)) {
      it_u = 0;
      it_uend = TT;
      it_first = true;
      it_tile = tile_of(++it_j);
      if (SPLIT && it_tile >= tiles && it_tail_left) {
        it_tail_left = false;
        it_tile = full_tiles + tail_idx;
        it_u = tail_u0;
        it_uend = tail_u1;
      }
    }
  };
  // (in-kernel experiments of rounds 2-3 -- a start stagger / instruction priorities for the second workgroup of a CU, a
  // counted wait behind the stores, s_setprio around the MFMAs -- are measured in profiles/archive/r02_pair_kernel_experiments.txt and
  // r03_pair_kernel_anatomy.txt; none paid, none is compiled any more)
  issue_next();

  int64_t stat_tile = -1;
  auto stats_readout = [&]() {
    const int64_t pm = tile_row(stat_tile), pn0 = tile_col(stat_tile) * PR_BN;
    for (int c = threadIdx.x; c < PR_BN; c += PR_TPB) {
      const int64_t n = pn0 + c;
      if (n < N) {
        double* dst = colstats + pm * 2 * N;   // one partial row per 128-row block (ccn_stats_rows)
        dst[n] = (double)stat_part[c * 2] + (double)stat_part[(PR_BN + c) * 2];
        dst[N + n] = (double)stat_part[c * 2 + 1] + (double)stat_part[(PR_BN + c) * 2 + 1];
      }
    }
    stat_tile = -1;
  };

  unsigned long long st_t0 = 0, st_a = 0, st_b = 0, st_r0 = 0;
  uint32_t sum_w = 0, sum_b = 0, sum_i = 0, sum_c = 0, sum_e = 0, n_sl = 0;
  if (STAMP) {
    PR_STAMP(st_t0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_r0) : : "memory");
  }
  int64_t g = 0;
  bool tail_left = has_tail_item;
  for (int64_t j = 0;; ++j) {
    int64_t tile = tile_of(j);
    int u_beg = 0, u_end = TT, part = -1;     // part >= 0: this item is part `part` of a split tile
    if (tile >= tiles) {
      if (!SPLIT || !tail_left) break;
      tail_left = false;
      tile = full_tiles + tail_idx;
      u_beg = tail_u0;
      u_end = tail_u1;
      part = tail_part;
    }
    const int64_t m0 = tile_row(tile) * PR_BM, n0 = tile_col(tile) * PR_BN;
    f32x16 acc[2][2];
    if (SPLIT && part > 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ab][t][r] = 0.f;
    } else
    if (ACC) {
      const bool inside = m0 + PR_BM <= M && n0 + PR_BN <= N;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int64_t n = n0 + wn * 64 + t * 32 + i;
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
          if (inside) {
            // as the epilogue's stores: scalar row base + one 32-bit lane offset, straight into the accumulator registers
            // (the 64-bit vector address per element of the generic form below cost this variant 256 VGPRs + 296 B of scratch)
            const float* const cbase = C + (m0 + wm * 64 + ab * 32) * ldc + n0 + wn * 64 + t * 32;
            const uint32_t lane_off = (uint32_t)((4 * h * ldc + i) * 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float* const rowp = cbase + (int64_t)((r & 3) + 8 * (r >> 2)) * ldc;
              asm volatile("global_load_dword %0, %1, %2" : "=v"(acc[ab][t][r]) : "v"(lane_off), "s"(rowp) : "memory");
            }
            continue;
          }
          const float* const crow = C + (m0 + wm * 64 + ab * 32 + 4 * h) * ldc + n;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            acc[ab][t][r] = (m < M && n < N) ? crow[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] : 0.f;
          }
        }
      }
      // The inline-asm loads above are invisible to the compiler's own wait insertion: this wait names the accumulators as
      // operands, so no copy, spill or AGPR move of them can be scheduled in front of it.  It is the wait the first slice
      // would take anyway (vmcnt(0): nothing is issued in between), moved up by a few scalar instructions.
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) : : "memory");
    } else {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t n = n0 + wn * 64 + t * 32 + i;
      const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ab][t][r] = bv;
    }
    }

    for (int u = SPLIT ? u_beg : 0; u < (SPLIT ? u_end : TT); ++u, ++g) {
      if (STAMP) PR_STAMP(st_a);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slice g has landed (and the previous tile's stores retired)
      if (STAMP) { PR_STAMP(st_b); sum_w += (uint32_t)(st_b - st_a); }
      __builtin_amdgcn_s_barrier();
      if (STAMP) { PR_STAMP(st_a); sum_b += (uint32_t)(st_a - st_b); }
      issue_next();
      if (u == (SPLIT ? u_beg : 0) && stat_tile >= 0) stats_readout();
      if (STAMP) { PR_STAMP(st_b); sum_i += (uint32_t)(st_b - st_a); }
      const uint32_t stage_b = lds_base + (uint32_t)((g & 1) * STAGE * 4);
      f32x4 fa[2][2], fb[2][2];   // [K group parity][block]
      auto read_group = [&](int q, f32x4 (&da)[2], f32x4 (&db)[2]) {
        const uint32_t ch = 16u * (uint32_t)((2 * q + h) ^ swz);
        asm volatile("ds_read_b128 %0, %1" : "=v"(da[0]) : "v"(stage_b + a_off + ch) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(da[1]) : "v"(stage_b + a_off + ch), "n"(32 * BK * 4) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(db[0]) : "v"(stage_b + b_off + ch) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(db[1]) : "v"(stage_b + b_off + ch), "n"(32 * BK * 4) : "memory");
      };
      auto mfma_group = [&](int par) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc[ab][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][ab].x, fb[par][t].x, acc[ab][t], 0, 0, 0);
            acc[ab][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][ab].y, fb[par][t].y, acc[ab][t], 0, 0, 0);
            acc[ab][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][ab].z, fb[par][t].z, acc[ab][t], 0, 0, 0);
            acc[ab][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][ab].w, fb[par][t].w, acc[ab][t], 0, 0, 0);
          }
      };
      if (XF) {
        f32x4 xs[2], xh[2];   // scale / shift of the four channels of a K group, [parity]
        auto xf_read = [&](int q, f32x4& sc, f32x4& sh) {
          const uint32_t at = xf_base + (uint32_t)((u * BK + 4 * (2 * q + h)) * 4);
          asm volatile("ds_read_b128 %0, %1" : "=v"(sc) : "v"(at) : "memory");
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sh) : "v"(at), "n"(XF_MAX_K * 4) : "memory");
        };
        // One K component (x / y / z / w of the fragment quads) at a time: four MFMAs on four different accumulators, then
        // a handful of VALU instructions of the NEXT group's transform.  A wave issues in order and the matrix pipe takes
        // one MFMA per 64 cycles, so VALU work placed between MFMAs is free, VALU work behind all sixteen is not (first
        // form: transform after the group's MFMAs, 10-25 % slower than the plain product).
        // (round 5: TWO K components per call in packed form -- v_pk_fma_f32, v_pk_mul_f32 and two v_max_f32 per block instead
        // of fma / mul / cmp / cndmask + the VCC hazard's wait states per element: 8 vector instructions per call where the
        // round-2 form issued 16 + 8 s_nop.  max(v, v * neg) == (v > 0 ? v : v * neg) for 0 <= neg <= 1, the entry point checks it.)
        const f32x2 neg2 = {xf_neg, xf_neg};
        auto xform_pair = [&](int par, int c2) {
          __builtin_amdgcn_sched_barrier(0);
          const f32x2 sc = {xs[par][2 * c2], xs[par][2 * c2 + 1]}, sh = {xh[par][2 * c2], xh[par][2 * c2 + 1]};
#pragma unroll
          for (int ab = 0; ab < 2; ++ab) {
            // branch-free form of bn_act_value: same bits, except that ReLU gives -0.0 instead of +0.0 for a negative
            // input -- equal under comparison and in every sum it enters
            const f32x2 a = {fa[par][ab][2 * c2], fa[par][ab][2 * c2 + 1]};
            const f32x2 v = __builtin_elementwise_fma(a, sc, sh);
            const f32x2 w = v * neg2;
            fa[par][ab][2 * c2] = __builtin_fmaxf(v.x, w.x);
            fa[par][ab][2 * c2 + 1] = __builtin_fmaxf(v.y, w.y);
          }
        };
        auto mfma_comp = [&](int par, int comp) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              acc[ab][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][ab][comp], fb[par][t][comp], acc[ab][t], 0, 0, 0);
        };
        // (the waits in front of a transform name its registers as operands: a plain VALU use of an inline-asm ds_read's
        // result is otherwise free to be scheduled in front of the wait)
        read_group(0, fa[0], fb[0]);
        xf_read(0, xs[0], xh[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(xs[0]), "+v"(xh[0]) : : "memory");
        xform_pair(0, 0);
        xform_pair(0, 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int par = q & 1, nxt = (q + 1) & 1;
          if (q < 3) {
            read_group(q + 1, fa[nxt], fb[nxt]);
            xf_read(q + 1, xs[nxt], xh[nxt]);
          }
          mfma_comp(par, 0);
          if (q < 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[nxt][0]), "+v"(fa[nxt][1]), "+v"(xs[nxt]), "+v"(xh[nxt]) : : "memory");   // (issued before four MFMAs: long landed)
            xform_pair(nxt, 0);
          }
          mfma_comp(par, 1);
          if (q < 3) xform_pair(nxt, 1);
          mfma_comp(par, 2);
          mfma_comp(par, 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      read_group(0, fa[0], fb[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) {
          read_group(q + 1, fa[(q + 1) & 1], fb[(q + 1) & 1]);
          asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        mfma_group(q & 1);
      }
      if (STAMP) { PR_STAMP(st_a); sum_c += (uint32_t)(st_a - st_b); ++n_sl; }
    }
    if (STAMP) PR_STAMP(st_a);

    if (SPLIT && part >= 0) {
      // ---- part of a split tile: accumulators -> split_ws in register order (every store instruction of a wave writes 256
      // contiguous bytes), count; the last part to arrive re-reads ALL parts in part order and goes on to the epilogue, the
      // others are done.  The parts of a tile may run on different XCDs, whose L2s are not coherent for plain accesses: the
      // partials are stored and loaded with AGENT scope (`sc1`: what a relaxed agent-scope atomic store / load is on
      // gfx942 / gfx950 -- written through to, and fetched from, the memory side), ordered against the counter by
      // s_waitcnt vmcnt(0) + the workgroup barrier.  (First form: plain accesses between two __threadfence() = buffer_wbl2 +
      // buffer_inv of the XCD's WHOLE L2 per part: every part flushed the finished C tiles of its neighbours and evicted
      // their operand tiles -- 10 550 x 1024 x 1024 took 0.261 ms split against 0.220 unsplit.)
      int* const counters = reinterpret_cast<int*>(split_ws);
      float* const parts = split_ws + 1024 + (int64_t)tail_idx * split_s * (PR_BM * PR_BN);
      float* const mine = parts + (int64_t)part * (PR_BM * PR_BN);
      // (scalar base per register + one 32-bit lane offset, as the epilogue's stores: 64 different 64-bit vector addresses --
      // the offsets exceed the instruction's immediate -- cost 128 registers and spilled the accumulators.  s_nop 4 in front
      // of every access: with this many row pointers live the compiler parks them in VGPR lanes and fetches each with
      // v_readlane right in front of the inline asm -- a VALU write of an SGPR needs five wait states before a VMEM
      // instruction may use it as an address, and the hazard recogniser does not look inside inline asm: without the nops
      // a load took the PREVIOUS register's row, every now and then)
      const uint32_t lane_off = (uint32_t)threadIdx.x * 4u;
#pragma unroll
      for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* const rowp = mine + ((ab * 2 + t) * 16 + r) * PR_TPB;
            asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 sc1" : : "v"(lane_off), "v"(acc[ab][t][r]), "s"(rowp) : "memory");
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's partial is written through
      __syncthreads();
      if (threadIdx.x == 0) {
        const int old = atomicAdd(&counters[tail_idx], 1);
        split_flag = old;
        if (old == split_s - 1) counters[tail_idx] = 0;     // (every part has counted: left at zero for the next launch)
      }
      __syncthreads();
      if (split_flag != split_s - 1) break;                  // (the tail part is a workgroup's last item)
#pragma unroll
      for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float* const rowp = parts + ((ab * 2 + t) * 16 + r) * PR_TPB;
            asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1" : "=v"(acc[ab][t][r]) : "v"(lane_off), "s"(rowp) : "memory");
          }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) : : "memory");
      for (int pp = 1; pp < split_s; ++pp) {
        const float* const pq = parts + (int64_t)pp * (PR_BM * PR_BN);
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            // (one 32 x 32 block = 16 loads at a time: all 64 in flight would need 64 registers beside the accumulators)
            // (sixteen scalars, each named by the wait, as the y tile of the RED epilogue: an element of a vector as the
            // output of an inline-asm load may be copied into the vector's registers BEFORE the wait the compiler cannot see)
            float tmp[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float* const rowp = pq + ((ab * 2 + t) * 16 + r) * PR_TPB;
              asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1" : "=v"(tmp[r]) : "v"(lane_off), "s"(rowp) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(tmp[0]), "+v"(tmp[1]), "+v"(tmp[2]), "+v"(tmp[3]), "+v"(tmp[4]), "+v"(tmp[5]), "+v"(tmp[6]),
                           "+v"(tmp[7]), "+v"(tmp[8]), "+v"(tmp[9]), "+v"(tmp[10]), "+v"(tmp[11]), "+v"(tmp[12]), "+v"(tmp[13]),
                           "+v"(tmp[14]), "+v"(tmp[15])
                         :
                         : "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ab][t][r] += tmp[r];
          }
      }
    }

    // ---- tile epilogue (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
    const bool interior = m0 + PR_BM <= M && n0 + PR_BN <= N;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ncol = wn * 64 + t * 32 + i;
      const int64_t n = n0 + ncol;
      float s1 = 0.f, s2 = 0.f;
      float r_sc = 0.f, r_sh = 0.f, r_mu = 0.f, r_rs = 0.f;     // RED: this lane's column of the previous layer's table
      if (RED && n < N) {
        r_sc = xf_scale[n];
        r_sh = xf_scale[N + n];
        r_mu = xf_scale[2 * N + n];
        r_rs = xf_scale[3 * N + n];
      }
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {
        if (interior) {
          // store address = a scalar row base (tile, wave quadrant, register's row) + this lane's 32-bit offset (its 4 h rows
          // and its column): no vector address arithmetic per store
          // (inline asm: hipcc folds a uniform base + lane offset back into 64-bit vector address arithmetic per store)
          float* const cbase = C + (m0 + wm * 64 + ab * 32) * ldc + n0 + wn * 64 + t * 32;
          const uint32_t lane_off = (uint32_t)((4 * h * ldc + i) * 4);
          float yv[16];
          if (RED) {
            const float* const ybase = xf_shift + (m0 + wm * 64 + ab * 32) * red_ldy + n0 + wn * 64 + t * 32;
            const uint32_t lane_off_y = (uint32_t)((4 * h * red_ldy + i) * 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float* const rowp = ybase + (int64_t)((r & 3) + 8 * (r >> 2)) * red_ldy;
              asm volatile("global_load_dword %0, %1, %2" : "=v"(yv[r]) : "v"(lane_off_y), "s"(rowp) : "memory");
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* const rowp = cbase + (int64_t)((r & 3) + 8 * (r >> 2)) * ldc;
            asm volatile("global_store_dword %0, %1, %2" : : "v"(lane_off), "v"(acc[ab][t][r]), "s"(rowp) : "memory");
          }
          if (RED) {
            // the 16 loads were issued in front of the 16 stores and vmcnt retires in order: all but the newest 16 done =
            // the loads have landed.  The wait names the registers, so nothing reads them in front of it.
            asm volatile("s_waitcnt vmcnt(16)"
                         : "+v"(yv[0]), "+v"(yv[1]), "+v"(yv[2]), "+v"(yv[3]), "+v"(yv[4]), "+v"(yv[5]), "+v"(yv[6]), "+v"(yv[7]),
                           "+v"(yv[8]), "+v"(yv[9]), "+v"(yv[10]), "+v"(yv[11]), "+v"(yv[12]), "+v"(yv[13]), "+v"(yv[14]),
                           "+v"(yv[15])
                         :
                         : "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float yy = yv[r], v = acc[ab][t][r];
              const float g = (yy * r_sc + r_sh) > 0.f ? v : v * red_neg;
              s1 += g;
              s2 += g * ((yy - r_mu) * r_rs);
            }
          } else
          if (colstats != nullptr) {
            // column statistics only where a BatchNorm follows (a data-gradient launch skips 128 vector instructions per tile
            // and wave), two accumulator registers per packed instruction (v_pk_add_f32 / v_pk_fma_f32)
            f32x2 p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const f32x2 vv = {acc[ab][t][r], acc[ab][t][r + 1]};
              p1 += vv;
              p2 = __builtin_elementwise_fma(vv, vv, p2);
            }
            s1 += p1.x + p1.y;
            s2 += p2.x + p2.y;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M && n < N) {
              const float v = acc[ab][t][r];
              C[m * ldc + n] = v;
              if (RED) {
                const float yy = xf_shift[m * red_ldy + n];
                const float g = (yy * r_sc + r_sh) > 0.f ? v : v * red_neg;
                s1 += g;
                s2 += g * ((yy - r_mu) * r_rs);
              } else {
                s1 += v;
                s2 += v * v;
              }
            }
          }
        }
      }
      if (colstats != nullptr) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) {
          stat_part[(wm * PR_BN + ncol) * 2] = s1;
          stat_part[(wm * PR_BN + ncol) * 2 + 1] = s2;
        }
      }
    }
    if (colstats != nullptr) {
      stat_tile = tile;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // table written before this wave reaches the next barrier
    }
    if (STAMP) { PR_STAMP(st_b); sum_e += (uint32_t)(st_b - st_a); }
  }
  if (STAMP && g_pair_dbg_dev != nullptr) {
    unsigned long long t1, r1;
    PR_STAMP(t1);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) : : "memory");
    if (lane == 0) {
      unsigned long long* d = g_pair_dbg_dev + ((int64_t)blockIdx.x * 4 + wave) * 16;
      d[0] = t1 - st_t0; d[1] = r1 - st_r0; d[2] = sum_w; d[3] = sum_b; d[4] = sum_i; d[5] = sum_c; d[6] = sum_e; d[7] = n_sl;
      d[8] = st_t0; d[9] = t1; d[10] = st_r0; d[11] = r1;
      uint32_t xcc, hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      d[12] = xcc; d[13] = hwid;
    }
  }
  if (stat_tile >= 0) {  // statistics of the last tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stats_readout();
  }
}

static std::atomic<void*> g_pair_dbg_host{nullptr};
static std::atomic<bool> g_use_pair{true};  // A-B hook (ccn_gemm_use_dma(4) = the 8-wave persistent kernel for N > 64 as well)
// from this many 128 x 128 tiles on (measured at 128 / 256 / 512 / 1024: 3168 x 2048 -> 1024 (200 tiles) 87 vs 80 TFLOP/s on
// the register-staged kernel, 10550 x 1024 -> 1024 (664 tiles) 99 vs 84, 35151 x 512 -> 512 107 vs 88 on the 8-wave kernel)
constexpr int64_t PAIR_MIN_TILES = 128;

// Tail split (see the kernel): parts per tile of the last, partly filled round.  1 = none.  Needs caller-owned scratch
// (ccn_gemm_nt_split_workspace_bytes: 4 KiB of counters, zero on first use and left zero, + one 64 KiB partial per part).
constexpr int64_t PAIR_SLOTS = 512;                // two workgroups per CU
constexpr size_t SPLIT_WS_HEAD = 4096;             // counters
constexpr size_t SPLIT_WS_BYTES = SPLIT_WS_HEAD + (size_t)PAIR_SLOTS * PR_BM * PR_BN * 4;
static int pair_split_parts(int64_t tiles, int64_t K, const void* ws, size_t ws_bytes) {
  if (ws == nullptr || (g_pair_opt & (4 | 512)) || ((uintptr_t)ws & 15)) return 1;      // (512: A-B hook, no split)
  const int64_t rem = tiles % PAIR_SLOTS, TT = K / BK + (K % BK != 0);
  // (K >= 512 only: at K = 256 the fix-up -- a 64 KiB partial written through, the count, up to eight partials read back --
  // costs what the split saves: 78 136 x 256 x 256 0.096 -> 0.104 ms.  A lone workgroup on a CU runs ~1.85x as fast as one of a
  // pair, so an unsplit tail round costs ~0.54 of a tile time, not a whole one: the split's ceiling is ~13 % of such a launch.)
  if (rem == 0 || rem > PAIR_SLOTS / 2 || TT < 16) return 1;
  int64_t sp = PAIR_SLOTS / rem;
  if (sp > 8) sp = 8;
  if (sp > TT / 4) sp = TT / 4;                    // a part is at least four slices long
  while (sp >= 2 && SPLIT_WS_HEAD + (size_t)(rem * sp) * PR_BM * PR_BN * 4 > ws_bytes) --sp;
  return sp >= 2 ? (int)sp : 1;
}

int launch_glds_pair(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                     int64_t M, int64_t N, int64_t K, double* colstats, hipStream_t s, int64_t a_extent, bool accumulate = false,
                     const float* xf_scale = nullptr, const float* xf_shift = nullptr, int xf_act = 0, float xf_slope = 0.f,
                     bool split_part = false, int64_t red_ldy = 0, void* ws = nullptr, size_t ws_bytes = 0) {
  const int64_t gm = (M + PR_BM - 1) / PR_BM, gn = (N + PR_BN - 1) / PR_BN;
  const int64_t tiles = gm * gn;
  if (tiles >= ((int64_t)1 << 31)) {
    ccn_set_error("gemm_nt: more than 2^31 output tiles");
    return CCN_ERR_ARG;
  }
  const int64_t slots = (g_pair_opt & 4) ? 256 : PAIR_SLOTS;
  const int sp = pair_split_parts(tiles, K, ws, ws_bytes);
  const int64_t rem = tiles % PAIR_SLOTS;
  const int64_t full_tiles = sp > 1 ? tiles - rem : tiles;
  const int64_t grid = sp > 1 ? (full_tiles > 0 ? PAIR_SLOTS : rem * sp) : (tiles < slots ? tiles : slots);
  float* const sws = sp > 1 ? (float*)ws : nullptr;
#define CCN_PAIR_LAUNCH(ACC_, XF_, TAG_, STAMP_, RED_)                                                                           \
  do {                                                                                                                          \
    if (sp > 1)                                                                                                                 \
      hipLaunchKernelGGL((gemm_glds_pair_kernel<ACC_, XF_, TAG_, STAMP_, RED_, true>), dim3((unsigned)grid), dim3(PR_TPB), 0, s, A, \
                         lda, W, ldw, bias, Y, ldy, M, N, K, tiles, gn, g_xcd_map ? 1 : 0, colstats, a_extent, xf_scale,       \
                         xf_shift, xf_act, xf_slope, red_ldy, full_tiles, sp, sws);                                             \
    else                                                                                                                        \
      hipLaunchKernelGGL((gemm_glds_pair_kernel<ACC_, XF_, TAG_, STAMP_, RED_, false>), dim3((unsigned)grid), dim3(PR_TPB), 0, s, A, \
                         lda, W, ldw, bias, Y, ldy, M, N, K, tiles, gn, g_xcd_map ? 1 : 0, colstats, a_extent, xf_scale,       \
                         xf_shift, xf_act, xf_slope, red_ldy, full_tiles, sp, sws);                                             \
  } while (0)
  if (g_pair_dbg_host != nullptr && xf_scale == nullptr && !accumulate && !split_part)
    hipLaunchKernelGGL((gemm_glds_pair_kernel<false, false, 0, true, false, false>), dim3((unsigned)(tiles < slots ? tiles : slots)),
                       dim3(PR_TPB), 0, s, A, lda, W, ldw, bias, Y, ldy, M, N, K, tiles, gn, g_xcd_map ? 1 : 0, colstats, a_extent,
                       xf_scale, xf_shift, xf_act, xf_slope, red_ldy, tiles, 1, (float*)nullptr);
  else if (red_ldy > 0)      // (xf_scale = the previous layer's table, xf_shift = its output y: see RED)
    CCN_PAIR_LAUNCH(false, false, 0, false, true);
  else if (xf_scale != nullptr)
    CCN_PAIR_LAUNCH(false, true, 0, false, false);
  else if (accumulate)
    CCN_PAIR_LAUNCH(true, false, 0, false, false);
  else if (split_part)
    CCN_PAIR_LAUNCH(false, false, 1, false, false);
  else
    CCN_PAIR_LAUNCH(false, false, 0, false, false);
#undef CCN_PAIR_LAUNCH
  return CCN_OK;
}


template <int BN>
int launch_glds_persistent(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y,
                           int64_t ldy, int64_t M, int64_t N, int64_t K, double* colstats, hipStream_t s) {
  if constexpr (BN <= 64) {
    // two 4-wave workgroups per CU on 128-row tiles (measured stand-alone, 2.1 M x 64 x 64: 67.4 -> 73.2 TFLOP/s, 618 k x 64 x 64:
    // 61.8 -> 68.1; A/B hook: ccn_gemm_pair_opt bit 8 = the 8-wave form)
    if (!(g_pair_opt & 256)) {
      const int64_t gm4 = (M + 127) / 128, gn4 = (N + BN - 1) / BN;
      const int64_t tiles4 = gm4 * gn4;
      const int64_t grid4 = tiles4 < 512 ? tiles4 : 512;
      hipLaunchKernelGGL((gemm_glds_persistent_kernel<BN, 3, 4>), dim3((unsigned)grid4), dim3(256), 0, s, A, lda, W, ldw, bias,
                         Y, ldy, M, N, K, tiles4, gn4, g_xcd_map ? 1 : 0, colstats);
      return CCN_OK;
    }
  }
  const int64_t gm = (M + GL_BM - 1) / GL_BM, gn = (N + BN - 1) / BN;
  const int64_t tiles = gm * gn;
  const int64_t grid = tiles < 256 ? tiles : 256;  // one workgroup per CU (147 KB of LDS each)
  // (a 256 x 256 tile with a 2-stage ring was tried: +2 % at N = 512, -6 % where it leaves few tiles, 12 spilled VGPRs)
  hipLaunchKernelGGL((gemm_glds_persistent_kernel<BN, 3>), dim3((unsigned)grid), dim3(GL_TPB), 0, s, A, lda, W, ldw, bias,
                     Y, ldy, M, N, K, tiles, gn, g_xcd_map ? 1 : 0, colstats);
  return CCN_OK;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static std::atomic<bool> g_force_generic{false};  // test hook (ccn_gemm_force_generic)
static std::atomic<bool> g_use_glds{true};        // test / A-B hook (ccn_gemm_use_dma)
static std::atomic<bool> g_use_persistent{true};  // A-B hook (ccn_gemm_use_dma(2) = DMA without the persistent tile loop)
static std::atomic<int64_t> g_dma_min_k{64};      // DMA kernels from this K on (A-B hook: ccn_gemm_use_dma(k >= 32) sets it)

template <int BM, int BN, int WM, int ALAY, int BLAY, int EPI>
int launch_gemm(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, int64_t ksplit, double* colstats, hipStream_t s, bool generic = false) {
  const int64_t gm = (M + BM - 1) / BM, gn = (N + BN - 1) / BN;
  int64_t kchunk = (K + ksplit - 1) / ksplit;
  kchunk = (kchunk + BK - 1) / BK * BK;
  const int64_t gz = (K + kchunk - 1) / kchunk;
  const int a_vec = aligned16(A) && (lda % 4 == 0);
  const int b_vec = aligned16(B) && (ldb % 4 == 0);
  if (gm > 2147483647LL || gn > 65535 || gz > 65535) {
    ccn_set_error("gemm: grid too large");
    return CCN_ERR_ARG;
  }
  // fast path needs 16-byte aligned rows and at least one 16-byte group per row to clamp into
  // (generic: an operand whose rows overlap -- ccn_conv_rows_* -- has no leading dimension to clamp tail loads against)
  const bool fast = a_vec && b_vec && lda >= 4 && ldb >= 4 && !g_force_generic && !generic;
  // split-K (weight-gradient) products with several tiles per chunk: XCD-ordered one-dimensional grid (see the kernel)
  const bool xcd_order = EPI == EPI_ATOMIC && g_xcd_map && gz >= 8 && gm * gn > 1 &&
                         gm * gn * ((gz + 7) / 8 * 8) <= 2147483647LL;
  const dim3 grid = xcd_order ? dim3((unsigned)(gm * gn * ((gz + 7) / 8 * 8))) : dim3((unsigned)gm, (unsigned)gn, (unsigned)gz);
  if (fast && kchunk > 96)
    hipLaunchKernelGGL((gemm_fast_kernel<BM, BN, WM, ALAY, BLAY, EPI, true>), grid, dim3(GEMM_TPB), 0, s, A, lda, B, ldb,
                       bias, C, ldc, M, N, K, kchunk, colstats, xcd_order ? 1 : 0);
  else if (fast)
    hipLaunchKernelGGL((gemm_fast_kernel<BM, BN, WM, ALAY, BLAY, EPI, false>), grid, dim3(GEMM_TPB), 0, s, A, lda, B, ldb,
                       bias, C, ldc, M, N, K, kchunk, colstats, xcd_order ? 1 : 0);
  else
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, ALAY, BLAY, EPI>), dim3((unsigned)gm, (unsigned)gn, (unsigned)gz),
                       dim3(GEMM_TPB), 0, s, A, lda, B, ldb, bias, C, ldc, M, N, K, kchunk, a_vec, b_vec, colstats);
  return CCN_OK;
}

// ------------------------------------------------------------------ column reductions (BatchNorm)
constexpr int RED_ROWS = 128;  // rows per workgroup == GEMM BM, so both produce ceil(rows/128) partial rows


// MODE 0: (sum x, 0)   MODE 1: (sum g, sum g*xhat) with g = dZ * act'(y*scale+shift)
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ X, int64_t ldx,
                                                          const float* __restrict__ Y, int64_t ldy, int64_t rows,
                                                          int64_t C, const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, int act, float slope,
                                                          double* __restrict__ partial) {
  __shared__ double red[4][64][2];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * RED_ROWS;
  const int64_t r1 = r0 + RED_ROWS < rows ? r0 + RED_ROWS : rows;
  for (int64_t c0 = 0; c0 < C; c0 += 64) {
    const int64_t c = c0 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
      float sc = 0.f, sh = 0.f, mu = 0.f, rs = 0.f;
      if (MODE == 1) {
        sc = scale[c];
        sh = shift[c];
        mu = mean[c];
        rs = rstd[c];
      }
      for (int64_t r = r0 + ry; r < r1; r += 4) {
        const float xv = X[r * ldx + c];
        if (MODE == 0) {
          s1 += (double)xv;
        } else {
          const float y = Y[r * ldy + c];
          const float g = xv * act_grad(y * sc + sh, act, slope);
          s1 += (double)g;
          s2 += (double)(g * ((y - mu) * rs));
        }
      }
    }
    red[ry][cx][0] = s1;
    red[ry][cx][1] = s2;
    __syncthreads();
    if (ry == 0 && c < C) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += red[w][cx][0];
        b += red[w][cx][1];
      }
      partial[(int64_t)blockIdx.x * 2 * C + c] = a;
      partial[(int64_t)blockIdx.x * 2 * C + C + c] = b;
    }
    __syncthreads();
  }
}

// Row-weighted form (compact SGCNN rows: a representative row stands for `w` identical rows of the dense layout).
// MODE 0: (sum w*x, sum w*x^2)   MODE 1: (sum w*g, sum w*g*xhat)
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_w_kernel(const float* __restrict__ X, int64_t ldx,
                                                            const float* __restrict__ Y, int64_t ldy,
                                                            const float* __restrict__ w, int64_t rows, int64_t C,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int act, float slope,
                                                            double* __restrict__ partial) {
  __shared__ double red[4][64][2];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * RED_ROWS;
  const int64_t r1 = r0 + RED_ROWS < rows ? r0 + RED_ROWS : rows;
  for (int64_t c0 = 0; c0 < C; c0 += 64) {
    const int64_t c = c0 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
      float sc = 0.f, sh = 0.f, mu = 0.f, rs = 0.f;
      if (MODE == 1) {
        sc = scale[c];
        sh = shift[c];
        mu = mean[c];
        rs = rstd[c];
      }
      for (int64_t r = r0 + ry; r < r1; r += 4) {
        const double wr = w ? (double)w[r] : 1.0;
        const float xv = X[r * ldx + c];
        if (MODE == 0) {
          s1 += wr * (double)xv;
          s2 += wr * (double)xv * (double)xv;
        } else {
          const float y = Y[r * ldy + c];
          const float g = xv * act_grad(y * sc + sh, act, slope);
          s1 += wr * (double)g;
          s2 += wr * (double)(g * ((y - mu) * rs));
        }
      }
    }
    red[ry][cx][0] = s1;
    red[ry][cx][1] = s2;
    __syncthreads();
    if (ry == 0 && c < C) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a += red[k][cx][0];
        b += red[k][cx][1];
      }
      partial[(int64_t)blockIdx.x * 2 * C + c] = a;
      partial[(int64_t)blockIdx.x * 2 * C + C + c] = b;
    }
    __syncthreads();
  }
}

// float4 form of the above for 16-byte aligned operands with C % 4 == 0: a wave instruction covers 4 rows x 64
// columns (lane = 16*row + column quad) and two such groups are in flight per lane, which is what it takes to keep
// HBM busy from 128-row workgroups (the scalar form reaches 3.5 TB/s, this one streams at the BN-apply rate).
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_vec_kernel(const float* __restrict__ X, int64_t ldx,
                                                              const float* __restrict__ Y, int64_t ldy, int64_t rows,
                                                              int64_t C, const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, int act, float slope,
                                                              double* __restrict__ partial, int64_t ccols) {
  // gridDim.y slices of ccols columns (a multiple of 64): a layer with few rows and many channels (10 550 x 1024, 3 168 x
  // 2048: 83 / 25 row blocks) otherwise runs 16-32 column passes per workgroup on a tenth of the CUs
  __shared__ double red[4][64][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & 15, rsub = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * RED_ROWS;
  const int64_t r1 = r0 + RED_ROWS < rows ? r0 + RED_ROWS : rows;
  const int64_t cbeg = (int64_t)blockIdx.y * ccols, cend = cbeg + ccols < C ? cbeg + ccols : C;
  for (int64_t c0 = cbeg; c0 < cend; c0 += 64) {
    const int64_t c = c0 + 4 * cq;
    const bool in = c < C;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    float4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc, mu = sc, rs = sc;
    if (MODE == 1 && in) {
      sc = *reinterpret_cast<const float4*>(scale + c);
      sh = *reinterpret_cast<const float4*>(shift + c);
      mu = *reinterpret_cast<const float4*>(mean + c);
      rs = *reinterpret_cast<const float4*>(rstd + c);
    }
    if (in) {
      for (int64_t r = r0 + wave * 4 + rsub; r < r1; r += 32) {
        const int64_t rb = r + 16;
        const bool two = rb < r1;
        float4 xa = *reinterpret_cast<const float4*>(X + r * ldx + c), xb = {0.f, 0.f, 0.f, 0.f};
        float4 ya = {0.f, 0.f, 0.f, 0.f}, yb = ya;
        if (two) xb = *reinterpret_cast<const float4*>(X + rb * ldx + c);
        if (MODE == 1) {
          ya = *reinterpret_cast<const float4*>(Y + r * ldy + c);
          if (two) yb = *reinterpret_cast<const float4*>(Y + rb * ldy + c);
        }
        const float* xs[2] = {reinterpret_cast<const float*>(&xa), reinterpret_cast<const float*>(&xb)};
        const float* ys[2] = {reinterpret_cast<const float*>(&ya), reinterpret_cast<const float*>(&yb)};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (t == 1 && !two) break;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (MODE == 0) {
              s1[k] += (double)xs[t][k];
            } else {
              const float y = ys[t][k];
              const float g = xs[t][k] * act_grad(y * (&sc.x)[k] + (&sh.x)[k], act, slope);
              s1[k] += (double)g;
              s2[k] += (double)(g * ((y - (&mu.x)[k]) * (&rs.x)[k]));
            }
          }
        }
      }
    }
    // fold the 4 row sub-groups of the wave (lanes l, l^16, l^32, l^48), then the 4 waves through LDS
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s1[k] += __shfl_xor(s1[k], 16, 64);
      s1[k] += __shfl_xor(s1[k], 32, 64);
      s2[k] += __shfl_xor(s2[k], 16, 64);
      s2[k] += __shfl_xor(s2[k], 32, 64);
    }
    if (rsub == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        red[wave][4 * cq + k][0] = s1[k];
        red[wave][4 * cq + k][1] = s2[k];
      }
    }
    __syncthreads();
    if (wave == 0 && c0 + lane < C) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += red[w][lane][0];
        b += red[w][lane][1];
      }
      partial[(int64_t)blockIdx.x * 2 * C + c0 + lane] = a;
      partial[(int64_t)blockIdx.x * 2 * C + C + c0 + lane] = b;
    }
    __syncthreads();
  }
}

static inline bool col_vec_ok(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t C) {
  return C % 4 == 0 && lda % 4 == 0 && ((uintptr_t)a % 16) == 0 && (b == nullptr || (ldb % 4 == 0 && ((uintptr_t)b % 16) == 0));
}

// Two-level deterministic reduction of the partial rows: sums[w] = sum_p partial[p][w], w < width.
// Level 1: grid (width/64, slices): every workgroup folds `per_slice` consecutive partial rows into the
// first row of its slice (in place).  Level 2: one workgroup per 64 columns adds the slice heads.
__global__ __launch_bounds__(256) void col_slice_kernel(double* __restrict__ partial, int64_t nparts, int64_t width,
                                                        int64_t per_slice) {
  __shared__ double red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t w = (int64_t)blockIdx.x * 64 + cx;
  const int64_t p0 = (int64_t)blockIdx.y * per_slice;
  const int64_t p1 = p0 + per_slice < nparts ? p0 + per_slice : nparts;
  double s = 0.0;
  if (w < width)
    for (int64_t p = p0 + ry; p < p1; p += 4) s += partial[p * width + w];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && w < width && p0 < nparts) partial[p0 * width + w] = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
}

// sum of the slice heads of one column: 4 waves x strided heads, 8 loads in flight per lane (the heads were written by other
// XCDs: every load is an L2 miss), combined in a fixed order
__device__ __forceinline__ double col_heads_sum(const double* __restrict__ partial, int64_t nslices, int64_t per_slice,
                                                int64_t width, int64_t w, bool live, double (*red)[64]) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (live)
    for (int64_t p0 = ry; p0 < nslices; p0 += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t p = p0 + 4 * u;
        acc[u] += p < nslices ? partial[p * per_slice * width + w] : 0.0;
      }
    }
  red[ry][cx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  const double total = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(256) void col_final_kernel(const double* __restrict__ partial, int64_t nslices,
                                                        int64_t per_slice, int64_t width, double* __restrict__ sums) {
  __shared__ double red[4][64];
  const int64_t w = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const double total = col_heads_sum(partial, nslices, per_slice, width, w, w < width, red);
  if ((threadIdx.x >> 6) == 0 && w < width) sums[w] = total;
}

// the same for a table of [sum | sum of squares] (width = 2 C) with the BatchNorm finalisation of ccn_bn_finalize fused in:
// one workgroup per 64 channels reduces both halves and writes scale / shift / mean / rstd and the running statistics
__global__ __launch_bounds__(256) void col_final_bn_kernel(const double* __restrict__ partial, int64_t nslices,
                                                           int64_t per_slice, int64_t rows, int64_t C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float momentum, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ scale,
                                                           float* __restrict__ shift, float* __restrict__ save_mean,
                                                           float* __restrict__ save_rstd, double* __restrict__ sums) {
  __shared__ double red[4][64];
  const int64_t c = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const double s1 = col_heads_sum(partial, nslices, per_slice, 2 * C, c, c < C, red);
  const double s2 = col_heads_sum(partial, nslices, per_slice, 2 * C, C + c, c < C, red);
  if ((threadIdx.x >> 6) != 0 || c >= C) return;
  sums[c] = s1;
  sums[C + c] = s2;
  const double n = (double)rows;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * rstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  save_mean[c] = (float)mean;
  save_rstd[c] = rstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

struct ColSlices {
  int64_t nslices, per_slice;
};

// level 1 of the reduction (in place, into the first row of every slice); at most 64 slices, so that level 2 is short
ColSlices launch_col_slices(double* partial, int64_t nparts, int64_t width, hipStream_t s) {
  int64_t nslices = (nparts + 7) / 8;
  if (nslices > 64) nslices = 64;
  if (nslices < 1) nslices = 1;
  const int64_t per_slice = (nparts + nslices - 1) / nslices;
  nslices = (nparts + per_slice - 1) / per_slice;
  if (per_slice > 1)
    hipLaunchKernelGGL(col_slice_kernel, dim3((unsigned)ccn_blocks(width, 64), (unsigned)nslices), dim3(256), 0, s, partial,
                       nparts, width, per_slice);
  return ColSlices{nslices, per_slice};
}

void launch_col_reduce(double* partial, int64_t nparts, int64_t width, double* sums, hipStream_t s) {
  const ColSlices cs = launch_col_slices(partial, nparts, width, s);
  hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)ccn_blocks(width, 64)), dim3(256), 0, s, partial, cs.nslices,
                     cs.per_slice, width, sums);
}

void launch_bn_finalize(double* partial, int64_t nparts, int64_t rows, int64_t C, const float* gamma, const float* beta, float eps,
                        float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* save_mean,
                        float* save_rstd, hipStream_t s) {
  const ColSlices cs = launch_col_slices(partial, nparts, 2 * C, s);
  hipLaunchKernelGGL(col_final_bn_kernel, dim3((unsigned)ccn_blocks(C, 64)), dim3(256), 0, s, partial, cs.nslices, cs.per_slice,
                     rows, C, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_rstd,
                     partial + nparts * 2 * C);
}

__global__ void bn_eval_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float eps, int64_t C, float* __restrict__ scale, float* __restrict__ shift,
                                      float* __restrict__ save_mean, float* __restrict__ save_rstd) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.0f / sqrtf(running_var[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * rstd;
  shift[c] = b - running_mean[c] * g * rstd;
  save_mean[c] = running_mean[c];
  save_rstd[c] = rstd;
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ Y, int64_t ldy, int64_t rows,
                                                         int C, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, float slope,
                                                         float* __restrict__ Z, int64_t ldz) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  for (int c = cx; c < C; c += 64) {
    const float sc = scale[c], sh = shift[c];
    for (int64_t r = r0 + ry; r < r0 + 32 && r < rows; r += 4) Z[r * ldz + c] = bn_act_value(Y[r * ldy + c], sc, sh, act, slope);
  }
}

__global__ void bn_act_fwd_vec_kernel(const float4* __restrict__ Y, int64_t ldy4, int64_t rows, int64_t C4,
                                      const float4* __restrict__ scale, const float4* __restrict__ shift, int act,
                                      float slope, float4* __restrict__ Z, int64_t ldz4) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * C4) return;
  const int64_t r = t / C4, c = t - r * C4;
  const float4 y = Y[r * ldy4 + c], sc = scale[c], sh = shift[c];
  float4 z;
  z.x = bn_act_value(y.x, sc.x, sh.x, act, slope);
  z.y = bn_act_value(y.y, sc.y, sh.y, act, slope);
  z.z = bn_act_value(y.z, sc.z, sh.z, act, slope);
  z.w = bn_act_value(y.w, sc.w, sh.w, act, slope);
  Z[r * ldz4 + c] = z;
}

__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(
    const float* __restrict__ dZ, int64_t lddz, const float* __restrict__ Y, int64_t ldy, int64_t rows, int C,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int act, float slope, const double* __restrict__ sums, int training,
    float* __restrict__ dY, int64_t lddy, float* __restrict__ dgamma, float* __restrict__ dbeta, float inv_n,
    int acc_params) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  for (int c = cx; c < C; c += 64) {
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    const float m1 = (float)sums[c] * inv_n, m2 = (float)sums[C + c] * inv_n;
    for (int64_t r = r0 + ry; r < r0 + 32 && r < rows; r += 4) {
      const float y = Y[r * ldy + c];
      const float g = dZ[r * lddz + c] * act_grad(y * sc + sh, act, slope);
      dY[r * lddy + c] = training ? sc * (g - m1 - (y - mu) * rs * m2) : sc * g;
    }
    if (blockIdx.x == 0 && ry == 0) {  // acc_params: add into the caller's gradient buffers (gradient-bucket views)
      if (dgamma) dgamma[c] = (acc_params ? dgamma[c] : 0.f) + (float)sums[C + c];
      if (dbeta) dbeta[c] = (acc_params ? dbeta[c] : 0.f) + (float)sums[c];
    }
  }
}

__global__ void colsum_out_kernel(const double* __restrict__ sums, int64_t C, float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) out[c] = (float)sums[c];
}

}  // namespace

template <int BM, int BN, int WM>
static int launch_bf16_tn(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                          int64_t N, int64_t K, hipStream_t s) {
  // output N x K, contraction over the M rows split across ~2048 workgroups
  const int64_t gm = (N + BM - 1) / BM, gn = (K + BN - 1) / BN;
  int64_t ksplit = (2048 + gm * gn - 1) / (gm * gn);
  const int64_t max_split = (M + 4 * BK - 1) / (4 * BK);
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1) ksplit = 1;
  int64_t kchunk = (M + ksplit - 1) / ksplit;
  kchunk = (kchunk + BK - 1) / BK * BK;
  const int64_t gz = (M + kchunk - 1) / kchunk;
  if (gm > 2147483647LL || gn > 65535 || gz > 65535) {
    ccn_set_error("gemm_tn_bf16: grid too large");
    return CCN_ERR_ARG;
  }
  hipLaunchKernelGGL((gemm_bf16_tn_kernel<BM, BN, WM>), dim3((unsigned)gm, (unsigned)gn, (unsigned)gz), dim3(GEMM_TPB), 0, s,
                     dY, lddy, X, ldx, dW, lddw, N, K, M, kchunk);
  return CCN_OK;
}

extern "C" {

int64_t ccn_stats_rows(int64_t rows) { return (rows + RED_ROWS - 1) / RED_ROWS; }

int ccn_gemm_use_dma(int on) {
  if (on >= 32) {  // threshold experiment: DMA kernels from K >= on
    g_dma_min_k = on;
    g_use_glds = g_use_persistent = true;
    return CCN_OK;
  }
  g_dma_min_k = 64;
  g_use_glds = on != 0;
  g_use_persistent = on == 1 || on == 3 || on == 4;
  g_xcd_map = on != 3;
  g_use_pair = on != 4;
  return CCN_OK;
}

int ccn_gemm_pair_debug(void* buf) {   // diagnostic: 512 x 4 x 16 uint64 words (see STAMP above); nullptr = off
  g_pair_dbg_host = buf;
  unsigned long long* p = (unsigned long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pair_dbg_dev), &p, sizeof(p)) == hipSuccess ? CCN_OK : CCN_ERR_ARG;
}

int ccn_gemm_pair_opt(int bits) {
  g_pair_opt = bits;
  return CCN_OK;
}

int ccn_gemm_force_generic(int on) {
  g_force_generic = on != 0;
  return CCN_OK;
}

int ccn_gemm_nt_bf16(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                     int64_t M, int64_t N, int64_t K, double* colstats, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(A && W && Y, "gemm_nt_bf16: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N, "gemm_nt_bf16: bad sizes M=%lld N=%lld K=%lld",
              (long long)M, (long long)N, (long long)K);
  CCN_REQUIRE(aligned16(A) && aligned16(W) && lda % 4 == 0 && ldw % 4 == 0 && lda >= 4 && ldw >= 4,
              "gemm_nt_bf16: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  if (M == 0) return CCN_OK;
  const int64_t gm = (M + 127) / 128;
  CCN_REQUIRE(gm <= 2147483647LL, "gemm_nt_bf16: grid too large");
  if (N <= 32)
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 32, 4>), dim3((unsigned)gm, (unsigned)((N + 31) / 32)), dim3(GEMM_TPB), 0, s,
                       A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats);
  else if (N <= 64)
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 64, 4>), dim3((unsigned)gm, (unsigned)((N + 63) / 64)), dim3(GEMM_TPB), 0, s,
                       A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 4>), dim3((unsigned)gm, (unsigned)((N + 127) / 128)), dim3(GEMM_TPB), 0,
                       s, A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats);
  CCN_LAUNCH_OK("gemm_nt_bf16");
  return CCN_OK;
}

int ccn_gemm_nt_f16(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                    int64_t M, int64_t N, int64_t K, double* colstats, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(A && W && Y, "gemm_nt_f16: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N, "gemm_nt_f16: bad sizes M=%lld N=%lld K=%lld",
              (long long)M, (long long)N, (long long)K);
  CCN_REQUIRE(aligned16(A) && aligned16(W) && lda % 4 == 0 && ldw % 4 == 0 && lda >= 4 && ldw >= 4,
              "gemm_nt_f16: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  if (M == 0) return CCN_OK;
  const int64_t gm = (M + 127) / 128;
  CCN_REQUIRE(gm <= 2147483647LL, "gemm_nt_f16: grid too large");
  if (N <= 32)
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 32, 4, true>), dim3((unsigned)gm, (unsigned)((N + 31) / 32)), dim3(GEMM_TPB), 0,
                       s, A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats);
  else if (N <= 64)
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 64, 4, true>), dim3((unsigned)gm, (unsigned)((N + 63) / 64)), dim3(GEMM_TPB), 0,
                       s, A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 4, true>), dim3((unsigned)gm, (unsigned)((N + 127) / 128)), dim3(GEMM_TPB),
                       0, s, A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats);
  CCN_LAUNCH_OK("gemm_nt_f16");
  return CCN_OK;
}

static int gemm_nt_impl(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                        int64_t M, int64_t N, int64_t K, double* colstats, void* stream, bool overlap, void* ws = nullptr,
                        size_t ws_bytes = 0);

int ccn_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                int64_t M, int64_t N, int64_t K, double* colstats, void* stream) {
  CCN_REQUIRE(A && W && Y, "gemm_nt: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N, "gemm_nt: bad sizes M=%lld N=%lld K=%lld",
              (long long)M, (long long)N, (long long)K);
  return gemm_nt_impl(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, stream, false);
}

// ---- the same products with caller-owned scratch for the tail split of the paired kernel (round 5; launch_glds_pair).  The
// scratch is used only when the split applies; its first 4 KiB must be zero on first use and are left zero.  One buffer per
// stream (two launches that may overlap must not share it).
size_t ccn_gemm_nt_split_workspace_bytes(void) { return SPLIT_WS_BYTES; }

int ccn_gemm_nt_split_parts(int64_t M, int64_t N, int64_t K, size_t workspace_bytes) {   // what the launch would do (diagnostic)
  static const char dummy[16] __attribute__((aligned(16))) = {0};
  return pair_split_parts(((M + PR_BM - 1) / PR_BM) * ((N + PR_BN - 1) / PR_BN), K, dummy, workspace_bytes);
}

int ccn_gemm_nt_ws(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                   int64_t M, int64_t N, int64_t K, double* colstats, void* workspace, size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(A && W && Y, "gemm_nt_ws: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N, "gemm_nt_ws: bad sizes M=%lld N=%lld K=%lld",
              (long long)M, (long long)N, (long long)K);
  return gemm_nt_impl(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, stream, false, workspace, workspace_bytes);
}

int ccn_gemm_nt_acc_ok(int64_t lda, int64_t ldw, int64_t M, int64_t N, int64_t K) {
  return lda % 4 == 0 && ldw % 4 == 0 && lda >= 4 && ldw >= 4 && !g_force_generic && g_use_glds && g_use_persistent &&
         g_use_pair && M >= 1024 && K >= g_dma_min_k && N > 64 &&
         ((M + PR_BM - 1) / PR_BM) * ((N + PR_BN - 1) / PR_BN) >= PAIR_MIN_TILES;
}

int ccn_gemm_nt_acc_ws(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t M, int64_t N,
                       int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(A && W && Y, "gemm_nt_acc: null pointer");
  CCN_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N, "gemm_nt_acc: bad sizes");
  CCN_REQUIRE(ccn_gemm_nt_acc_ok(lda, ldw, M, N, K) && aligned16(A) && aligned16(W),
              "gemm_nt_acc: shape / alignment outside the paired LDS-DMA kernel (ask ccn_gemm_nt_acc_ok first)");
  int rc = launch_glds_pair(A, lda, W, ldw, nullptr, Y, ldy, M, N, K, nullptr, (hipStream_t)stream, lda, true, nullptr, nullptr, 0,
                            0.f, false, 0, workspace, workspace_bytes);
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_nt_acc");
  return CCN_OK;
}

int ccn_gemm_nt_acc(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t M, int64_t N,
                    int64_t K, void* stream) {
  return ccn_gemm_nt_acc_ws(A, lda, W, ldw, Y, ldy, M, N, K, nullptr, 0, stream);
}

// dZ = dY W^T-form product (as ccn_gemm_nt, no bias) + the BatchNorm-backward column sums of the layer that produced the
// deferred input y (see RED above).  par: that layer's 4 x N table (scale | shift | mean | rstd), rows N floats apart.
// sums: 2 N totals followed by [ccn_stats_rows(M)][2 N] doubles of scratch -- what ccn_bn_act_bwd_reduce writes.
int ccn_gemm_nt_red(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t M, int64_t N,
                    int64_t K, const float* y_prev, int64_t ldyp, const float* par, int act, float slope, double* sums,
                    void* stream) {
  return ccn_gemm_nt_red_ws(A, lda, W, ldw, Y, ldy, M, N, K, y_prev, ldyp, par, act, slope, sums, nullptr, 0, stream);
}

int ccn_gemm_nt_red_ws(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t M, int64_t N,
                       int64_t K, const float* y_prev, int64_t ldyp, const float* par, int act, float slope, double* sums,
                       void* workspace, size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(A && W && Y && y_prev && par && sums, "gemm_nt_red: null pointer");
  CCN_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N && ldyp >= N, "gemm_nt_red: bad sizes");
  CCN_REQUIRE(ccn_gemm_nt_acc_ok(lda, ldw, M, N, K) && aligned16(A) && aligned16(W) && ldyp < ((int64_t)1 << 27),
              "gemm_nt_red: shape / alignment outside the paired LDS-DMA kernel (ask ccn_gemm_nt_acc_ok first)");
  hipStream_t s = (hipStream_t)stream;
  double* partial = sums + 2 * N;
  int rc = launch_glds_pair(A, lda, W, ldw, nullptr, Y, ldy, M, N, K, partial, s, lda, false, par, y_prev, act, slope, false, ldyp,
                            workspace, workspace_bytes);
  if (rc) return rc;
  launch_col_reduce(partial, ccn_stats_rows(M), 2 * N, sums, s);
  CCN_LAUNCH_OK("gemm_nt_red");
  return CCN_OK;
}

int ccn_gemm_nt_xf_ok(int64_t lda, int64_t ldw, int64_t M, int64_t N, int64_t K) {
  return ccn_gemm_nt_acc_ok(lda, ldw, M, N, K) && K % BK == 0 && K <= XF_MAX_K;
}

int ccn_gemm_nt_xf(const float* A, int64_t lda, const float* a_scale, const float* a_shift, int a_act, float a_slope,
                   const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                   double* colstats, void* stream) {
  return ccn_gemm_nt_xf_ws(A, lda, a_scale, a_shift, a_act, a_slope, W, ldw, bias, Y, ldy, M, N, K, colstats, nullptr, 0, stream);
}

int ccn_gemm_nt_xf_ws(const float* A, int64_t lda, const float* a_scale, const float* a_shift, int a_act, float a_slope,
                      const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                      double* colstats, void* workspace, size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(A && a_scale && a_shift && W && Y, "gemm_nt_xf: null pointer");
  CCN_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N, "gemm_nt_xf: bad sizes");
  CCN_REQUIRE(a_act != CCN_ACT_LEAKY || (a_slope >= 0.f && a_slope <= 1.f), "gemm_nt_xf: LeakyReLU slope outside [0, 1]");
  CCN_REQUIRE(ccn_gemm_nt_xf_ok(lda, ldw, M, N, K) && aligned16(A) && aligned16(W),
              "gemm_nt_xf: shape / alignment outside the paired LDS-DMA kernel (ask ccn_gemm_nt_xf_ok first)");
  int rc = launch_glds_pair(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, (hipStream_t)stream, lda, false, a_scale, a_shift,
                            a_act, a_slope, false, 0, workspace, workspace_bytes);
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_nt_xf");
  return CCN_OK;
}

int ccn_conv_rows_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                     int64_t M, int64_t N, int64_t K, double* colstats, void* stream) {
  CCN_REQUIRE(A && W && Y, "conv_rows_nt: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lda > 0 && K % lda == 0 && ldw >= K && ldy >= N,
              "conv_rows_nt: bad sizes M=%lld N=%lld K=%lld lda=%lld (K must be taps * lda)", (long long)M, (long long)N,
              (long long)K, (long long)lda);
  return gemm_nt_impl(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, stream, K > lda);
}

static int gemm_nt_impl(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                        int64_t M, int64_t N, int64_t K, double* colstats, void* stream, bool overlap, void* ws,
                        size_t ws_bytes) {
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) return CCN_OK;
  int rc;
  // K < 64, or fewer than two rounds of 256-row tiles over the 256 CUs: the register-staged kernels (128-row tiles,
  // several workgroups per CU) win -- measured 95 / 101 vs 84 / 89 TFLOP/s at M = 10550 / 35151; from K = 64 on the
  // persistent DMA kernel is ahead (K = 64: 70 vs 62, K = 96: 91 vs 82 TFLOP/s over 1.3 M rows)
  const bool base_ok = aligned16(A) && aligned16(W) && lda % 4 == 0 && ldw % 4 == 0 && lda >= 4 && ldw >= 4 &&
                       !g_force_generic && g_use_glds && M >= 1024 && K >= g_dma_min_k;
  const bool dma_ok = base_ok && ((M + GL_BM - 1) / GL_BM) * ((N + 127) / 128) >= 512;
  // N > 64: two 4-wave workgroups per CU on 128 x 128 tiles
  // (K % 32 != 0 -- the widths made by the +3 xyz concat: 259, 262, 515, 1027, 2051 -- is handled inside the kernel)
  if (base_ok && N > 64 && g_use_persistent && g_use_pair &&
      ((M + PR_BM - 1) / PR_BM) * ((N + PR_BN - 1) / PR_BN) >= PAIR_MIN_TILES) {
    // The last 128-wide column tile at most half used (N = 192, 259, 262: a quarter to a third of the launch's MFMA work
    // would be on padding): the first N - N % 128 columns here, the remainder as a second product on the 64-wide kernels
    // (another pass over A, ~0.3 ms at 1.3 M x 256).  Not with BatchNorm statistics (their partial rows are laid out by N).
    const int64_t n_main = N / PR_BN * PR_BN;
    if ((g_pair_opt & 64) == 0 && colstats == nullptr && N > PR_BN && N % PR_BN != 0 && N % PR_BN <= 64 &&
        ((M + PR_BM - 1) / PR_BM) * (n_main / PR_BN) >= PAIR_MIN_TILES) {
      rc = launch_glds_pair(A, lda, W, ldw, bias, Y, ldy, M, n_main, K, nullptr, s, overlap ? K : lda, false, nullptr, nullptr, 0,
                            0.f, true, 0, ws, ws_bytes);
      if (rc) return rc;
      return gemm_nt_impl(A, lda, W + n_main * ldw, ldw, bias ? bias + n_main : nullptr, Y + n_main, ldy, M, N - n_main, K,
                          nullptr, stream, overlap);
    }
    rc = launch_glds_pair(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, s, overlap ? K : lda, false, nullptr, nullptr, 0, 0.f,
                          false, 0, ws, ws_bytes);
    if (rc) return rc;
    CCN_LAUNCH_OK("gemm_nt");
    return CCN_OK;
  }
  if (dma_ok && K % BK == 0 && g_use_persistent) {
    if (N <= 32)
      rc = launch_glds_persistent<32>(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, s);
    else if (N <= 64)
      rc = launch_glds_persistent<64>(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, s);
    else
      rc = launch_glds_persistent<128>(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, s);
    if (rc) return rc;
    CCN_LAUNCH_OK("gemm_nt");
    return CCN_OK;
  }
  if (dma_ok && !overlap) {
    if (N <= 32)
      rc = launch_glds<32>(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, s);
    else if (N <= 64)
      rc = launch_glds<64>(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, s);
    else
      rc = launch_glds<128>(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, s);
    if (rc) return rc;
    CCN_LAUNCH_OK("gemm_nt");
    return CCN_OK;
  }
  // overlapping rows with a K tail: the fast loaders clamp tail loads against the leading dimension -> generic kernel
  const bool gen = overlap && K % BK != 0;
  if (N <= 32)
    rc = launch_gemm<128, 32, 4, KC, KC, EPI_STORE>(A, lda, W, ldw, bias, Y, ldy, M, N, K, 1, colstats, s, gen);
  else if (N <= 64)
    rc = launch_gemm<128, 64, 4, KC, KC, EPI_STORE>(A, lda, W, ldw, bias, Y, ldy, M, N, K, 1, colstats, s, gen);
  else
    rc = launch_gemm<128, 128, 4, KC, KC, EPI_STORE>(A, lda, W, ldw, bias, Y, ldy, M, N, K, 1, colstats, s, gen);
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_nt");
  return CCN_OK;
}

int ccn_gemm_nn(const float* dY, int64_t lddy, const float* W, int64_t ldw, float* dX, int64_t lddx, int64_t M,
                int64_t N, int64_t K, void* stream) {
  // dX[M x K] = dY[M x N] * W[N x K]: contraction over N (the layer's output channels)
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(dY && W && dX, "gemm_nn: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldw >= K && lddx >= K, "gemm_nn: bad sizes");
  if (M == 0) return CCN_OK;
  int rc;
  if (K <= 32)
    rc = launch_gemm<128, 32, 4, KC, MC, EPI_STORE>(dY, lddy, W, ldw, nullptr, dX, lddx, M, K, N, 1, nullptr, s);
  else if (K <= 64)
    rc = launch_gemm<128, 64, 4, KC, MC, EPI_STORE>(dY, lddy, W, ldw, nullptr, dX, lddx, M, K, N, 1, nullptr, s);
  else
    rc = launch_gemm<128, 128, 4, KC, MC, EPI_STORE>(dY, lddy, W, ldw, nullptr, dX, lddx, M, K, N, 1, nullptr, s);
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_nn");
  return CCN_OK;
}

int ccn_gemm_tn_generic(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                        int64_t N, int64_t K, int overlap, void* stream);

int ccn_gemm_tn(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                int64_t N, int64_t K, void* stream) {
  CCN_REQUIRE(dY && X && dW, "gemm_tn: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K, "gemm_tn: bad sizes");
  return ccn_gemm_tn_generic(dY, lddy, X, ldx, dW, lddw, M, N, K, 0, stream);
}

int ccn_gemm_tn_generic(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                        int64_t N, int64_t K, int overlap, void* stream) {
  // dW[N x K] += dY[M x N]^T * X[M x K]: contraction over the M rows, split across workgroups.
  // overlap: the rows of X overlap (ldx < K, ccn_conv_rows_tn) -> the kernels with per-element bounds
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) return CCN_OK;
  const bool gen = overlap != 0;
  const int64_t tiles = ((N + 63) / 64) * ((K + 127) / 128);
  int64_t ksplit = (2048 + tiles - 1) / tiles;  // aim at ~2048 workgroups
  const int64_t max_split = (M + 4 * BK - 1) / (4 * BK);
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1) ksplit = 1;
  int rc;
  if (N <= 32) {
    rc = launch_gemm<32, 128, 1, MC, MC, EPI_ATOMIC>(dY, lddy, X, ldx, nullptr, dW, lddw, N, K, M, ksplit, nullptr, s, gen);
  } else {
    if (K <= 64)
      rc = launch_gemm<64, 64, 2, MC, MC, EPI_ATOMIC>(dY, lddy, X, ldx, nullptr, dW, lddw, N, K, M, ksplit, nullptr, s, gen);
    else if (N > 64 && M >= 50000)  // 128 x 128 tiles once the row split alone fills the chip: 112 -> 121 TFLOP/s at
                                    // M = 1.3 M, K = N = 256 (but 112 -> 105 at M = 10 k, where the 64-row tile stays)
      rc = launch_gemm<128, 128, 4, MC, MC, EPI_ATOMIC>(dY, lddy, X, ldx, nullptr, dW, lddw, N, K, M, ksplit, nullptr, s, gen);
    else
      rc = launch_gemm<64, 128, 2, MC, MC, EPI_ATOMIC>(dY, lddy, X, ldx, nullptr, dW, lddw, N, K, M, ksplit, nullptr, s, gen);
  }
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_tn");
  return CCN_OK;
}

int ccn_gemm_tn_bf16(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                     int64_t N, int64_t K, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(dY && X && dW, "gemm_tn_bf16: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K, "gemm_tn_bf16: bad sizes");
  CCN_REQUIRE(aligned16(dY) && aligned16(X) && lddy % 4 == 0 && ldx % 4 == 0 && lddy >= 4 && ldx >= 4,
              "gemm_tn_bf16: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  if (M == 0) return CCN_OK;
  int rc;
  if (N <= 64)
    rc = K <= 64 ? launch_bf16_tn<64, 64, 2>(dY, lddy, X, ldx, dW, lddw, M, N, K, s)
                 : launch_bf16_tn<64, 128, 2>(dY, lddy, X, ldx, dW, lddw, M, N, K, s);
  else
    rc = K <= 64 ? launch_bf16_tn<128, 64, 4>(dY, lddy, X, ldx, dW, lddw, M, N, K, s)
                 : launch_bf16_tn<128, 128, 4>(dY, lddy, X, ldx, dW, lddw, M, N, K, s);
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_tn_bf16");
  return CCN_OK;
}

int ccn_bn_finalize(const double* colstats, int64_t rows, int64_t C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                    float* save_mean, float* save_rstd, void* stream) {
  // colstats: [ccn_stats_rows(rows)][2*C] partial rows followed by 2*C doubles of scratch for the totals
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(colstats && scale && shift && save_mean && save_rstd && rows > 0 && C > 0, "bn_finalize: bad arguments");
  const int64_t nparts = ccn_stats_rows(rows);
  launch_bn_finalize(const_cast<double*>(colstats), nparts, rows, C, gamma, beta, eps, momentum, running_mean, running_var,
                     scale, shift, save_mean, save_rstd, s);      // (the totals land behind the partial rows, as before)
  CCN_LAUNCH_OK("bn_finalize");
  return CCN_OK;
}

int ccn_bn_finalize_n(const double* partial, int64_t nparts, int64_t rows, int64_t C, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                      float* scale, float* shift, float* save_mean, float* save_rstd, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(partial && scale && shift && save_mean && save_rstd && rows > 0 && C > 0 && nparts > 0,
              "bn_finalize_n: bad arguments");
  launch_bn_finalize(const_cast<double*>(partial), nparts, rows, C, gamma, beta, eps, momentum, running_mean, running_var,
                     scale, shift, save_mean, save_rstd, s);
  CCN_LAUNCH_OK("bn_finalize_n");
  return CCN_OK;
}

int ccn_reduce_partials(double* partial, int64_t nparts, int64_t width, double* sums, void* stream) {
  CCN_REQUIRE(partial && sums && nparts > 0 && width > 0, "reduce_partials: bad arguments");
  launch_col_reduce(partial, nparts, width, sums, (hipStream_t)stream);
  CCN_LAUNCH_OK("reduce_partials");
  return CCN_OK;
}

int ccn_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int64_t C, float* scale, float* shift, float* save_mean, float* save_rstd,
                       void* stream) {
  CCN_REQUIRE(running_mean && running_var && scale && shift && save_mean && save_rstd && C > 0,
              "bn_eval_params: bad arguments");
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3(ccn_blocks(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift, save_mean, save_rstd);
  CCN_LAUNCH_OK("bn_eval_params");
  return CCN_OK;
}

int ccn_bn_act_fwd(const float* Y, int64_t ldy, int64_t rows, int64_t C, const float* scale, const float* shift,
                   int act, float slope, float* Z, int64_t ldz, void* stream) {
  CCN_REQUIRE(Y && Z && scale && shift && ldy >= C && ldz >= C, "bn_act_fwd: bad arguments");
  if (rows * C == 0) return CCN_OK;
  const bool vec = (C % 4 == 0) && (ldy % 4 == 0) && (ldz % 4 == 0) && aligned16(Y) && aligned16(Z) &&
                   aligned16(scale) && aligned16(shift);
  if (vec)
    hipLaunchKernelGGL(bn_act_fwd_vec_kernel, dim3(ccn_blocks(rows * C / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)Y, ldy / 4, rows, C / 4, (const float4*)scale, (const float4*)shift, act, slope,
                       (float4*)Z, ldz / 4);
  else
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(ccn_blocks(rows, 32)), dim3(256), 0, (hipStream_t)stream, Y, ldy, rows,
                       (int)C, scale, shift, act, slope, Z, ldz);
  CCN_LAUNCH_OK("bn_act_fwd");
  return CCN_OK;
}

// columns per gridDim.y slice of col_partial_vec_kernel: enough slices for ~1024 workgroups, whole 64-column groups
static inline int64_t col_slice_cols(int64_t nparts, int64_t C, unsigned* gy) {
  const int64_t groups = (C + 63) / 64;
  int64_t want = nparts >= 1024 ? 1 : (1024 + nparts - 1) / nparts;
  if (want > groups) want = groups;
  const int64_t per = (groups + want - 1) / want;       // 64-column groups per slice
  *gy = (unsigned)((groups + per - 1) / per);
  return per * 64;
}

int ccn_bn_act_bwd_reduce(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                          const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                          float slope, double* sums, void* stream) {
  // sums: 2*C totals followed by [ccn_stats_rows(rows)][2*C] doubles of scratch
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(dZ && Y && scale && shift && mean && rstd && sums && rows > 0 && C > 0, "bn_act_bwd_reduce: bad arguments");
  const int64_t nparts = ccn_stats_rows(rows);
  double* partial = sums + 2 * C;
  if (col_vec_ok(dZ, lddz, Y, ldy, C) && ((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)rstd) % 16 == 0) {
    unsigned gy;
    const int64_t ccols = col_slice_cols(nparts, C, &gy);
    hipLaunchKernelGGL(col_partial_vec_kernel<1>, dim3((unsigned)nparts, gy), dim3(256), 0, s, dZ, lddz, Y, ldy, rows, C,
                       scale, shift, mean, rstd, act, slope, partial, ccols);
  } else
    hipLaunchKernelGGL(col_partial_kernel<1>, dim3((unsigned)nparts), dim3(256), 0, s, dZ, lddz, Y, ldy, rows, C, scale,
                       shift, mean, rstd, act, slope, partial);
  launch_col_reduce(partial, nparts, 2 * C, sums, s);
  CCN_LAUNCH_OK("bn_act_bwd_reduce");
  return CCN_OK;
}

int ccn_bn_act_bwd_apply(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                         const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                         float slope, const double* sums, int training, float* dY, int64_t lddy, float* dgamma,
                         float* dbeta, void* stream) {
  CCN_REQUIRE(dZ && Y && scale && shift && mean && rstd && sums && dY && rows > 0 && C > 0,
              "bn_act_bwd_apply: bad arguments");
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ccn_blocks(rows, 32)), dim3(256), 0, (hipStream_t)stream, dZ, lddz,
                     Y, ldy, rows, (int)C, scale, shift, mean, rstd, act, slope, sums, training, dY, lddy, dgamma,
                     dbeta, 1.0f / (float)rows, 0);
  CCN_LAUNCH_OK("bn_act_bwd_apply");
  return CCN_OK;
}

// ---- row-weighted BatchNorm pieces (compact SGCNN rows; `count` = sum of all row weights = rows of the dense layout)
int ccn_colstats_weighted(const float* X, int64_t ldx, const float* w, int64_t rows, int64_t C, double* acc,
                          void* stream) {
  // acc: 2*C totals (sum w*x, sum w*x^2) followed by [ccn_stats_rows(rows)][2*C] doubles of scratch
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(X && acc && rows > 0 && C > 0 && ldx >= C, "colstats_weighted: bad arguments");   // w == NULL: weights 1
  const int64_t nparts = ccn_stats_rows(rows);
  double* partial = acc + 2 * C;
  hipLaunchKernelGGL(col_partial_w_kernel<0>, dim3((unsigned)nparts), dim3(256), 0, s, X, ldx, (const float*)nullptr,
                     (int64_t)0, w, rows, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, 0, 0.f, partial);
  launch_col_reduce(partial, nparts, 2 * C, acc, s);
  CCN_LAUNCH_OK("colstats_weighted");
  return CCN_OK;
}

int ccn_bn_act_bwd_reduce_weighted(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, const float* w,
                                   int64_t rows, int64_t C, const float* scale, const float* shift, const float* mean,
                                   const float* rstd, int act, float slope, double* sums, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(dZ && Y && w && scale && shift && mean && rstd && sums && rows > 0 && C > 0,
              "bn_act_bwd_reduce_weighted: bad arguments");
  const int64_t nparts = ccn_stats_rows(rows);
  double* partial = sums + 2 * C;
  hipLaunchKernelGGL(col_partial_w_kernel<1>, dim3((unsigned)nparts), dim3(256), 0, s, dZ, lddz, Y, ldy, w, rows, C,
                     scale, shift, mean, rstd, act, slope, partial);
  launch_col_reduce(partial, nparts, 2 * C, sums, s);
  CCN_LAUNCH_OK("bn_act_bwd_reduce_weighted");
  return CCN_OK;
}

int ccn_bn_act_bwd_apply_count(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                               const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                               float slope, const double* sums, double count, int training, float* dY, int64_t lddy,
                               float* dgamma, float* dbeta, void* stream) {
  return ccn_bn_act_bwd_apply_ex(dZ, lddz, Y, ldy, rows, C, scale, shift, mean, rstd, act, slope, sums, count, training, 0,
                                 dY, lddy, dgamma, dbeta, stream);
}

int ccn_bn_act_bwd_apply_ex(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                            const float* scale, const float* shift, const float* mean, const float* rstd, int act,
                            float slope, const double* sums, double count, int training, int accumulate_params,
                            float* dY, int64_t lddy, float* dgamma, float* dbeta, void* stream) {
  CCN_REQUIRE(dZ && Y && scale && shift && mean && rstd && sums && dY && rows > 0 && C > 0 && count > 0,
              "bn_act_bwd_apply_ex: bad arguments");
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ccn_blocks(rows, 32)), dim3(256), 0, (hipStream_t)stream, dZ, lddz,
                     Y, ldy, rows, (int)C, scale, shift, mean, rstd, act, slope, sums, training, dY, lddy, dgamma,
                     dbeta, (float)(1.0 / count), accumulate_params);
  CCN_LAUNCH_OK("bn_act_bwd_apply_ex");
  return CCN_OK;
}

int ccn_colsum(const float* X, int64_t ldx, int64_t rows, int64_t C, double* acc, float* out, void* stream) {
  // acc: 2*C totals followed by [ccn_stats_rows(rows)][2*C] doubles of scratch
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(X && acc && out && rows > 0 && C > 0 && ldx >= C, "colsum: bad arguments");
  const int64_t nparts = ccn_stats_rows(rows);
  double* partial = acc + 2 * C;
  if (col_vec_ok(X, ldx, nullptr, 0, C)) {
    unsigned gy;
    const int64_t ccols = col_slice_cols(nparts, C, &gy);
    hipLaunchKernelGGL(col_partial_vec_kernel<0>, dim3((unsigned)nparts, gy), dim3(256), 0, s, X, ldx, (const float*)nullptr,
                       (int64_t)0, rows, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, 0, 0.f, partial, ccols);
  } else
    hipLaunchKernelGGL(col_partial_kernel<0>, dim3((unsigned)nparts), dim3(256), 0, s, X, ldx, (const float*)nullptr,
                       (int64_t)0, rows, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, 0, 0.f, partial);
  launch_col_reduce(partial, nparts, 2 * C, acc, s);
  hipLaunchKernelGGL(colsum_out_kernel, dim3(ccn_blocks(C, 128)), dim3(128), 0, s, acc, C, out);
  CCN_LAUNCH_OK("colsum");
  return CCN_OK;
}

}  // extern "C"

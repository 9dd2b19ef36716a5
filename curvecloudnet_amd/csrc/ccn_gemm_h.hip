// 16-bit operand GEMMs on the LDS-DMA pipeline (BASELINE configs[2] "bf16 MLP MFMA path", configs[4] "fp16 features").
//
// Replaces, for the 16-bit MLP modes, the products of PyG MLP layers as the reference uses them (base.py:90-125; autograd of
// F.linear): the hidden activations of an MLP, the BatchNorm-backward gradient dY and the (cast) weights are STORED as
// bf16 / fp16 in HBM, so that these products stream half the operand bytes and need no conversion on the way to the
// matrix cores.  At the widths of the shipped networks (K, N <= 1024, mostly 64..256) the products are HBM-bound on
// v_mfma_f32_32x32x16_{bf16,f16} (16x the fp32 MFMA rate): the kernels are built around bytes in flight, not MFMA issue.
//
//   ccn_gemm_nt_h   Y[M x N] = A[M x K] W[N x K]^T (+ bias), fp32 accumulate; Y fp32 (+ BatchNorm column statistics of
//                   the fp32 result) or 16-bit (the data gradient handed to a 16-bit activation)
//   ccn_gemm_tn_h   dW[N x K] += dY[M x N]^T X[M x K]   (fp32 accumulation target)
//
// NT kernel = the fp32 paired kernel's structure (ccn_gemm.hip: two independent 4-wave workgroups per CU on 128 x 128
// tiles, two-stage LDS-DMA ring, XOR swizzle on the DMA source + ds_read_b128, persistent XCD-aware tile loop) with 64
// 16-bit elements per slice and row instead of 32 floats: the same 128-byte rows, the same LDS image, the same swizzle; a
// lane's fragment of k-step s (16 contraction elements per 32x32x16 MFMA, k = 16 s + 8 h .. + 7) is the 16-byte chunk
// 2 s + h of its row -- one ds_read_b128, no permutation.
#include "ccn_common.h"

#include <atomic>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u16 = uint16_t;

constexpr int HB_TPB = 256;
constexpr int HB_BM = 128, HB_BN = 128;
constexpr int HB_BK = 64;      // 16-bit elements per slice and row (128 bytes)

__device__ __forceinline__ void glds16h(const u16* src, u16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma_h(const f32x4& a, const f32x4& b, f32x16 c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ u16 to_h(float v) {
  if (F16) {
    const _Float16 x = (_Float16)v;
    return __builtin_bit_cast(u16, x);
  }
  const __bf16 x = (__bf16)v;          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(u16, x);
}

template <bool F16>
__device__ __forceinline__ float from_h(u16 v) {
  if (F16) return (float)__builtin_bit_cast(_Float16, v);
  return __builtin_bit_cast(float, (uint32_t)v << 16);
}

__device__ __forceinline__ float act_fwd_h(float z, int act, float slope) {
  if (act == CCN_ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == CCN_ACT_LEAKY) return z > 0.f ? z : z * slope;
  return z;
}
__device__ __forceinline__ float act_grad_h(float z, int act, float slope) {
  if (act == CCN_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == CCN_ACT_LEAKY) return z > 0.f ? 1.f : slope;
  return 1.f;
}

// OUT16: the result is written as 16-bit (a data gradient): the MFMA operands are SWAPPED, so that a lane holds one output
// ROW and four consecutive COLUMNS per register quad -- 16 eight-byte stores per lane and tile instead of 64 two-byte ones
// (the sums per output element are the same fma chains in the same k order: a*b commutes).  No statistics in that form.
// DIAG: the timing-only ablation build behind ccn_gemm_h_opt (include/ccn_hip_debug.h); in the shipping instantiations
// (DIAG = false) ``opt`` is the constant 0 and every branch on it is compiled out.
// BN: columns of a tile.  128: two workgroups per CU (64 KB of LDS each).  64 (round 4): 128 x 64 tiles, 48 KB of LDS and
// ~120 VGPRs per workgroup, THREE workgroups per CU -- the kernel is HBM-bound at the network's widths and a workgroup
// cannot have copies land past its own pending stores (vmcnt retires in order), so the store burst of one tile only
// overlaps OTHER workgroups' loads: more, smaller workgroups per CU keep more of both in flight.  The A tile is then read
// by N / 64 column tiles instead of N / 128: they sit on one XCD in the same step (tile_of) and share it through that L2.
// FUSE (round 6: the forward of a BatchNorm layer WITHOUT its fp32 intermediate -- VERDICT r3-r5).  The kernel is HBM-bound at the
// network's widths (MFMA pipes 17 % busy, 63 % of its bytes the fp32 result the BatchNorm pass re-reads), so the product is
// computed twice instead of stored once:  FUSE 1 = statistics only -- the fp32-result form with its stores compiled out (2 K bytes
// per row read, nothing written);  FUSE 2 = z = act(acc * ep_scale[n] + ep_shift[n]) in the epilogue (the expression of
// ccn_bn_act_fwd_h on the same fp32 sums: the same bits as product + BatchNorm pass), written as fp32 rows (!OUT16: a lane holds
// a column, its two constants are registers) or as 16-bit rows (OUT16: a lane holds a row, the constants of the tile's columns
// sit in LDS -- the statistics table, unused in this form -- and are read four at a time, one broadcast ds_read_b128 each).
// Per row: 4 K + 2 N bytes instead of 2 K + 10 N.
template <bool F16, bool OUT16, bool DIAG = false, int BN = HB_BN, int FUSE = 0>
__global__ __launch_bounds__(HB_TPB, BN == 128 ? 2 : 3) void gemm_h_pair_kernel(const u16* __restrict__ A, int64_t lda,
                                                                const u16* __restrict__ B, int64_t ldb,
                                                                const float* __restrict__ bias, void* __restrict__ Cv,
                                                                int64_t ldc, int64_t M, int64_t N, int64_t K, int64_t tiles,
                                                                int64_t gn, int xcd_order, double* __restrict__ colstats,
                                                                int opt_rt, int64_t a_extent,
                                                                const float* __restrict__ ep_scale,
                                                                const float* __restrict__ ep_shift, int ep_act, float ep_slope,
                                                                void* __restrict__ Tv, int64_t ldt) {
  // Tv (FUSE 3 = FUSE 2 with two results, OUT16 only): a second 16-bit result, the PRE-activation t = acc * scale + shift.  A ReLU layer needs it for its
  // backward pass -- xhat of a clipped element is not recoverable from z = 0, and BatchNorm's backward needs xhat of EVERY row --
  // a LeakyReLU layer does not (z is invertible).  2 N more bytes per row, still 6 N - 2 K fewer than the three-kernel form.
  static_assert(FUSE == 0 || !DIAG, "the fused forms have no diagnostics build");
  static_assert(FUSE != 1 || !OUT16, "statistics come from the fp32-result layout");
  static_assert(FUSE != 3 || OUT16, "the pre-activation output exists in the 16-bit form only");
  // a_extent: 16-bit elements readable from the start of an A row (= lda, or K when rows overlap: ccn_conv_rows_nt_h)
  // opt (diagnostics, ccn_gemm_h_opt; results WRONG when set): bit 0 = no epilogue stores, bit 1 = no wait for the LDS-DMA
  const int opt = DIAG ? opt_rt : 0;
  static_assert(BN == 128 || (BN == 64 && !OUT16), "tile widths: 128, or 64 for the fp32-result form");
  constexpr int AF = HB_BM * HB_BK, BF = BN * HB_BK, STAGE = AF + BF;   // 16-bit elements
  constexpr int NC = 4;         // LDS-DMA copies (8 rows x 128 B) per wave and slice of A
  constexpr int NCB = BN / 32;  // ... of B
  constexpr int NTW = BN / 64;  // 32-column blocks per wave (waves: 2 x 2 quadrants of 64 x BN/2)
  __shared__ __attribute__((aligned(16))) u16 lds[2 * STAGE];
  __shared__ float stat_part[2 * BN * 2];   // [wm][column][sum, sum of squares] of the finished tile

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: SGPR addressing)
  const int wm = wave & 1, wn = wave >> 1;
  const int i = lane & 31, h = lane >> 5;
  const int swz = (i >> 1) & 7;
  const int lr = lane >> 3, lc = lane & 7;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const uint32_t a_off = (uint32_t)((wm * 64 + i) * HB_BK * 2);            // bytes: A fragment row of block ab = 0
  const uint32_t b_off = (uint32_t)((AF + (wn * (BN / 2) + i) * HB_BK) * 2);   // bytes: B fragment row of block t = 0
  const int T = (int)(K / HB_BK);
  const int has_tail = (K % HB_BK) != 0;
  const int TT = T + has_tail;

  const uint32_t gnu = (uint32_t)gn;
  const uint32_t gm_tiles = (uint32_t)tiles / gnu;
  // (workgroups go to the 8 XCDs round-robin: the gridDim.x / 8 slots of an XCD take whole rows of tiles, gnu at a time)
  const bool xcd_map = xcd_order && gridDim.x >= 512 && (gridDim.x & 7) == 0 && (gridDim.x >> 3) % gnu == 0;
  const uint32_t slot = blockIdx.x >> 3;
  const uint32_t rows_per_step = gridDim.x / gnu, slot_row = (slot / gnu) * 8 + (blockIdx.x & 7), slot_col = slot % gnu;
  auto tile_of = [&](int64_t j) -> int64_t {
    if (xcd_map) {
      const uint32_t m = (uint32_t)j * rows_per_step + slot_row;
      return m < gm_tiles ? (int64_t)(m * gnu + slot_col) : tiles;
    }
    return j * gridDim.x + blockIdx.x;
  };
  auto tile_row = [&](int64_t t) -> int64_t { return (int64_t)((uint32_t)t / gnu); };
  auto tile_col = [&](int64_t t) -> int64_t { return (int64_t)((uint32_t)t % gnu); };

  // ---- issue cursor: one slice ahead of the compute cursor, across tile boundaries
  // (a copy's source = scalar tile base + 32-bit lane offset: no vector address arithmetic per copy, as in ccn_gemm.hip)
  uint32_t a_off32[NC], b_off32[NCB];
  const char* a_tile = reinterpret_cast<const char*>(A);
  const char* b_tile = reinterpret_cast<const char*>(B);
  int64_t it_j = 0, it_tile = tile_of(0), gi = 0;
  int64_t im0 = 0, in0 = 0;
  int it_u = 0;
  auto issue_next = [&]() {
    if (it_tile >= tiles) return;
    if (it_u == 0) {
      im0 = tile_row(it_tile) * HB_BM;
      in0 = tile_col(it_tile) * BN;
      a_tile = reinterpret_cast<const char*>(A + im0 * lda);
      b_tile = reinterpret_cast<const char*>(B + in0 * ldb);
      const int64_t a_rows = M - im0, b_rows = N - in0;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const int r = 8 * (wave * NC + q) + lr;
        const int64_t ra = r < a_rows ? r : a_rows - 1;
        a_off32[q] = (uint32_t)((ra * lda + 8 * (lc ^ ((r >> 1) & 7))) * 2);
      }
#pragma unroll
      for (int q = 0; q < NCB; ++q) {
        const int r = 8 * (wave * NCB + q) + lr;
        const int64_t rb = r < b_rows ? r : b_rows - 1;
        b_off32[q] = (uint32_t)((rb * ldb + 8 * (lc ^ ((r >> 1) & 7))) * 2);
      }
    }
    u16* st = lds + (gi & 1) * STAGE;
    const int64_t k0 = (int64_t)it_u * HB_BK;
    if (it_u < T) {
      const char* const a_sl = a_tile + k0 * 2;
      const char* const b_sl = b_tile + k0 * 2;
#pragma unroll
      for (int q = 0; q < NC; ++q) glds16h(reinterpret_cast<const u16*>(a_sl + a_off32[q]), st + (8 * (wave * NC + q)) * HB_BK);
#pragma unroll
      for (int q = 0; q < NCB; ++q) glds16h(reinterpret_cast<const u16*>(b_sl + b_off32[q]), st + AF + (8 * (wave * NCB + q)) * HB_BK);
    } else {
      // K remainder (< 64 elements): both operands through registers, zero filled beyond K, into the same swizzled image
      // (nothing else is in flight here: every iteration waits vmcnt(0))
#pragma unroll
      for (int it = 0; it < (HB_BM + BN) * 8 / HB_TPB; ++it) {
        const int sl0 = threadIdx.x + it * HB_TPB;
        const bool isA = sl0 < HB_BM * 8;
        const int sl = isA ? sl0 : sl0 - HB_BM * 8;
        const int r = sl >> 3, kq = sl & 7;
        const u16* p = isA ? A : B;
        const int64_t ld = isA ? lda : ldb;
        int64_t row = (isA ? im0 : in0) + r;
        const int64_t lim = isA ? M : N;
        row = row < lim ? row : lim - 1;
        const int64_t k = k0 + kq * 8;
        const int64_t ext = isA ? a_extent : ldb;
        const int64_t kc = k <= ext - 8 ? k : ext - 8;   // (k > extent - 8 implies k >= K: everything is zeroed below)
        uint4 v = *reinterpret_cast<const uint4*>(p + row * ld + kc);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t lo = (k + 2 * e < K) ? (w[e] & 0xffffu) : 0u;
          const uint32_t hi = (k + 2 * e + 1 < K) ? (w[e] & 0xffff0000u) : 0u;
          w[e] = lo | hi;
        }
        *reinterpret_cast<uint4*>(st + (isA ? 0 : AF) + r * HB_BK + 8 * (kq ^ ((r >> 1) & 7))) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // written before this wave reaches the next barrier
    }
    ++gi;
    if (++it_u == TT) {
      it_u = 0;
      it_tile = tile_of(++it_j);
    }
  };
  if ((opt & 4) && blockIdx.x >= 256)          // (experiment: the second workgroup of a CU starts (opt >> 8) x 3.4 us late)
    for (int r = 0; r < (opt >> 8); ++r) __builtin_amdgcn_s_sleep(127);
  issue_next();

  int64_t stat_tile = -1;
  auto stats_readout = [&]() {
    const int64_t pm = tile_row(stat_tile), pn0 = tile_col(stat_tile) * BN;
    for (int c = threadIdx.x; c < BN; c += HB_TPB) {
      const int64_t n = pn0 + c;
      if (n < N) {
        double* dst = colstats + pm * 2 * N;   // one partial row per 128-row block (ccn_stats_rows)
        dst[n] = (double)stat_part[c * 2] + (double)stat_part[(BN + c) * 2];
        dst[N + n] = (double)stat_part[c * 2 + 1] + (double)stat_part[(BN + c) * 2 + 1];
      }
    }
    stat_tile = -1;
  };

  int64_t g = 0;
  for (int64_t j = 0, tile; (tile = tile_of(j)) < tiles; ++j) {
    const int64_t m0 = tile_row(tile) * HB_BM, n0 = tile_col(tile) * BN;
    f32x16 acc[2][NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      float bv = 0.f;
      if (!OUT16) {
        const int64_t n = n0 + wn * (BN / 2) + t * 32 + i;
        bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
      }
#pragma unroll
      for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ab][t][r] = bv;
    }

    for (int u = 0; u < TT; ++u, ++g) {
      if (!(opt & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slice g has landed (and the previous tile's stores retired)
      __builtin_amdgcn_s_barrier();
      issue_next();
      if (u == 0 && stat_tile >= 0) stats_readout();
      const uint32_t stage_b = lds_base + (uint32_t)((g & 1) * STAGE * 2);
      f32x4 fa[2][2], fb[2][NTW];   // [k-step parity][block]
      auto read_group = [&](int s, f32x4 (&da)[2], f32x4 (&db)[NTW]) {
        const uint32_t ch = 16u * (uint32_t)((2 * s + h) ^ swz);
        asm volatile("ds_read_b128 %0, %1" : "=v"(da[0]) : "v"(stage_b + a_off + ch) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(da[1]) : "v"(stage_b + a_off + ch), "n"(32 * HB_BK * 2) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(db[0]) : "v"(stage_b + b_off + ch) : "memory");
        if (NTW == 2)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(db[NTW - 1]) : "v"(stage_b + b_off + ch), "n"(32 * HB_BK * 2) : "memory");
      };
      read_group(0, fa[0], fb[0]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {
          read_group(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
          asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(2 + NTW) : "memory");   // the group issued last stays in flight
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            if (OUT16) acc[ab][t] = mfma_h<F16>(fb[s & 1][t], fa[s & 1][ab], acc[ab][t]);
            else acc[ab][t] = mfma_h<F16>(fa[s & 1][ab], fb[s & 1][t], acc[ab][t]);
          }
      }
    }

    // ---- tile epilogue
    if (opt & 1) {
      if (acc[0][0][0] == 1.2345e30f) reinterpret_cast<float*>(Cv)[0] = acc[1][NTW - 1][3] + acc[0][NTW - 1][5] + acc[1][0][7];   // (keeps the MFMAs alive)
      continue;
    }
    if (FUSE >= 2 && OUT16) {
      // the tile's BatchNorm constants: [scale | shift][BN] in the (otherwise unused) statistics table; the previous tile's readers
      // are at least one slice barrier behind, this tile's readers wait at the barrier below
      for (int c = threadIdx.x; c < BN; c += HB_TPB) {
        const int64_t n = n0 + c;
        stat_part[c] = n < N ? ep_scale[n] : 0.f;
        stat_part[BN + c] = n < N ? ep_shift[n] : 0.f;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int col = wn * 64 + t * 32 + 8 * q4 + 4 * h;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(&stat_part[col]);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(&stat_part[BN + col]);
#pragma unroll
          for (int ab = 0; ab < 2; ++ab)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ab][t][4 * q4 + e] = acc[ab][t][4 * q4 + e] * sc[e] + sh[e];
        }
    }
    // (FUSE 2 with a pre-activation output: the store code below runs twice -- t into Tv, then z = act(t) into Cv)
#pragma unroll
    for (int pass = (FUSE == 3) ? 0 : 1; pass < 2; ++pass) {
    void* const Ov = pass == 0 ? Tv : Cv;
    const int64_t ldo = pass == 0 ? ldt : ldc;
    if (FUSE >= 2 && OUT16 && pass == 1) {
#pragma unroll
      for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ab][t][r] = act_fwd_h(acc[ab][t][r], ep_act, ep_slope);
    }
    if (OUT16 && (N & 7) == 0 && (ldo & 7) == 0 && ((uintptr_t)Ov & 15) == 0) {
      // Through LDS, so that the tile leaves in whole 256-byte rows (the direct form below writes 8 bytes into each of 32
      // rows per instruction: the store phase ran at 3.3 TB/s against 5.7 for full lines, tools/bench_gemm_h_opt.py).  The
      // stage the last slice was read from is free until the next iteration's copies (issued behind that iteration's
      // barrier): [128 rows][256 bytes], 16-byte chunk c of row r at chunk c ^ (r & 15).
      u16* const C = reinterpret_cast<u16*>(Ov);
      const uint32_t tbase = lds_base + (uint32_t)(((g - 1) & 1) * STAGE * 2);
      __builtin_amdgcn_s_barrier();            // every wave has read its last fragments of that stage
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {
        const int r = wm * 64 + ab * 32 + i;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int col = wn * 64 + t * 32 + 8 * q4 + 4 * h;         // 4 consecutive columns = 8 bytes
            const uint32_t w0 = (uint32_t)to_h<F16>(acc[ab][t][4 * q4 + 0]) | ((uint32_t)to_h<F16>(acc[ab][t][4 * q4 + 1]) << 16);
            const uint32_t w1 = (uint32_t)to_h<F16>(acc[ab][t][4 * q4 + 2]) | ((uint32_t)to_h<F16>(acc[ab][t][4 * q4 + 3]) << 16);
            const uint32_t at = tbase + (uint32_t)(r * 256 + (((col >> 3) ^ (r & 15)) << 4) + ((col & 7) << 1));
            asm volatile("ds_write_b64 %0, %1" : : "v"(at), "v"(make_uint2(w0, w1)) : "memory");
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // 16 lanes x 16 bytes = one row; a wave instruction stores 4 rows, the workgroup 16: 8 passes
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int r = ps * 16 + (int)(threadIdx.x >> 4), c = (int)(threadIdx.x & 15);
        uint4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(tbase + (uint32_t)(r * 256 + ((c ^ (r & 15)) << 4))) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t m = m0 + r, n = n0 + 8 * c;
        if (m < M && n < N) *reinterpret_cast<uint4*>(C + m * ldo + n) = v;
      }
      continue;      // (the next iteration's barrier sits between these reads and the copies that reuse the stage)
    }
    if (OUT16) {
      // D[p][q] of mfma(B-fragment, A-fragment): p = column n (registers: (r&3) + 8*(r>>2) + 4*h), q = row m (lane i)
      u16* const C = reinterpret_cast<u16*>(Ov);
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {
        const int64_t m = m0 + wm * 64 + ab * 32 + i;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int64_t n = n0 + wn * 64 + t * 32 + 8 * q4 + 4 * h;
            if (m < M) {
              u16* const dst = C + m * ldo + n;
              const u16 v0 = to_h<F16>(acc[ab][t][4 * q4 + 0]), v1 = to_h<F16>(acc[ab][t][4 * q4 + 1]);
              const u16 v2 = to_h<F16>(acc[ab][t][4 * q4 + 2]), v3 = to_h<F16>(acc[ab][t][4 * q4 + 3]);
              if (n + 3 < N) {
                *reinterpret_cast<uint2*>(dst) = make_uint2((uint32_t)v0 | ((uint32_t)v1 << 16), (uint32_t)v2 | ((uint32_t)v3 << 16));
              } else {
                if (n + 0 < N) dst[0] = v0;
                if (n + 1 < N) dst[1] = v1;
                if (n + 2 < N) dst[2] = v2;
              }
            }
          }
      }
      continue;
    }
    break;
    }      // (passes)
    if (OUT16) continue;
    float* const C = reinterpret_cast<float*>(Cv);
    const bool interior = m0 + HB_BM <= M && n0 + BN <= N;
    // The statistics table is read out by waves 0-1 behind the FIRST slice barrier of the next tile and rewritten in that tile's
    // epilogue: with one slice per tile (K <= 64) nothing but the readers' own speed lies between the two.  The storing forms are
    // thousands of cycles of store issue away from the rewrite; the statistics-only form is not (found as a forward that was not
    // repeatable on the 64 -> 64 head of the nuScenes section): it takes a barrier here.
    if (TT == 1 && colstats != nullptr) __builtin_amdgcn_s_barrier();      // (every form: a barrier per tile at K <= 64 only)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ncol = wn * (BN / 2) + t * 32 + i;
      const int64_t n = n0 + ncol;
      float s1 = 0.f, s2 = 0.f;
      float esc = 0.f, esh = 0.f;
      if (FUSE == 2) {
        esc = n < N ? ep_scale[n] : 0.f;
        esh = n < N ? ep_shift[n] : 0.f;
      }
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {
        float* const crow = C + (m0 + wm * 64 + ab * 32 + 4 * h) * ldc + n;
        if (interior) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[ab][t][r];
            if (FUSE == 2) v = act_fwd_h(v * esc + esh, ep_act, ep_slope);
            if (FUSE != 1) crow[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = v;
            s1 += v;
            s2 = __builtin_fmaf(v, v, s2);      // (explicitly fused: the storing and the statistics-only build must agree to the bit)
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M && n < N) {
              float v = acc[ab][t][r];
              if (FUSE == 2) v = act_fwd_h(v * esc + esh, ep_act, ep_slope);
              if (FUSE != 1) C[m * ldc + n] = v;
              s1 += v;
              s2 = __builtin_fmaf(v, v, s2);      // (explicitly fused: the storing and the statistics-only build must agree to the bit)
            }
          }
        }
      }
      if (colstats != nullptr) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) {
          stat_part[(wm * BN + ncol) * 2] = s1;
          stat_part[(wm * BN + ncol) * 2 + 1] = s2;
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

// ---- element-wise companions: casts and the BatchNorm passes that read or write 16-bit rows ----------------------------
// Common shape: a thread owns ONE 8-column chunk (16 bytes of 16-bit data, 32 bytes of fp32) and walks rows; the per-column
// BatchNorm constants live in registers for the whole walk (a first form that re-read six of them per element was
// instruction-bound: 18 ms per step on BASELINE configs[2] against 5 ms for the fp32 pass it replaced).  A workgroup of
// 256 threads covers CPB = min(chunks per row, 256) chunks x RPP = 256 / CPB rows per pass; gridDim.y covers wider rows.
constexpr int EW_TPB = 256;
constexpr int EW_ROWS = 128;      // rows per workgroup (= one partial row of the statistics passes: ccn_stats_rows)

struct EwGeom {
  int cpb, rpp;            // chunks per workgroup row, rows per pass
  unsigned gy;             // workgroups along the columns
};
inline EwGeom ew_geom(int64_t ld16) {
  const int64_t cpr = ld16 / 8;
  EwGeom g;
  g.cpb = (int)(cpr < EW_TPB ? cpr : EW_TPB);
  g.rpp = EW_TPB / g.cpb;
  g.gy = (unsigned)((cpr + g.cpb - 1) / g.cpb);
  return g;
}

__device__ __forceinline__ void load8(const float* __restrict__ p, bool vec, int64_t c0, int64_t C, float (&v)[8]) {
  if (vec) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? p[e] : 0.f;
  }
}
__device__ __forceinline__ void load8h(const u16* __restrict__ p, bool vec, int64_t c0, int64_t C, float (&v)[8]) {   // bf16
  if (vec) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = __builtin_bit_cast(float, qq[e] << 16);
      v[2 * e + 1] = __builtin_bit_cast(float, qq[e] & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? from_h<false>(p[e]) : 0.f;
  }
}
__device__ __forceinline__ void load8f16(const u16* __restrict__ p, bool vec, int64_t c0, int64_t C, float (&v)[8]) {   // fp16
  if (vec) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = from_h<true>((u16)(qq[e] & 0xffffu));
      v[2 * e + 1] = from_h<true>((u16)(qq[e] >> 16));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? from_h<true>(p[e]) : 0.f;
  }
}
// The second operand of the BatchNorm-backward passes.  YT 0: the fp32 PRE-normalisation product y (rounds 3-5).  YT 1 / 2 / 3
// (round 6, layers whose forward ran ccn_gemm_nt_h_bnact and never wrote y): the layer's OUTPUT z = act(t), t = y scale + shift,
// as bf16 / fp16 / fp32 rows -- t = act^-1(z) (ReLU: z where z > 0, and act'(t) = 0 elsewhere, so nothing else is needed;
// LeakyReLU: z / slope where z <= 0), xhat = (t - beta) / gamma = t * xa + xb with xa = rstd / scale, xb = -(shift / scale + mean) rstd
// (gamma = 0 makes xhat unrecoverable AND dy = 0: xa = xb = 0 there).  ld in elements of the operand's own type.
template <int YT>
__device__ __forceinline__ void load_y8(const void* __restrict__ Yv, int64_t row_off, bool vec, int64_t c0, int64_t C, float (&v)[8]) {
  if (YT == 0 || YT == 3) load8(reinterpret_cast<const float*>(Yv) + row_off, vec, c0, C, v);
  else if (YT == 1) load8h(reinterpret_cast<const u16*>(Yv) + row_off, vec, c0, C, v);
  else load8f16(reinterpret_cast<const u16*>(Yv) + row_off, vec, c0, C, v);
}
template <int YT>
__device__ __forceinline__ bool y8_vec_ok(const void* Yv, int64_t ldy, bool full) {
  return full && ((YT == 0 || YT == 3) ? (ldy & 3) == 0 : (ldy & 7) == 0) && ((uintptr_t)Yv & 15) == 0;
}
__device__ __forceinline__ float act_inv_h(float z, int act, float inv_slope) {
  if (act == CCN_ACT_LEAKY) return z > 0.f ? z : z * inv_slope;      // (a multiplication: a division per element made the pass instruction-bound)
  return z;          // (ReLU is NOT invertible: such layers hand over their pre-activation, z_pre)
}

template <bool F16>
__device__ __forceinline__ void store8h(u16* __restrict__ p, const float (&v)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) w[e] = (uint32_t)to_h<F16>(v[2 * e]) | ((uint32_t)to_h<F16>(v[2 * e + 1]) << 16);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

// fp32 rows -> 16-bit rows; columns [C, ld16) are zeroed
template <bool F16>
__global__ __launch_bounds__(EW_TPB) void cast_rows_h_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int64_t C,
                                                             u16* __restrict__ Y, int64_t ldy, int cpb, int rpp) {
  const int ch = threadIdx.x % cpb, rr = threadIdx.x / cpb;
  const int64_t c0 = ((int64_t)blockIdx.y * cpb + ch) * 8;
  if (rr >= rpp || c0 >= ldy) return;
  const bool vec = c0 + 8 <= C && (ldx & 3) == 0 && ((uintptr_t)X & 15) == 0;
  const int64_t r0 = (int64_t)blockIdx.x * EW_ROWS;
  const int64_t r1 = r0 + EW_ROWS < rows ? r0 + EW_ROWS : rows;
  for (int64_t r = r0 + rr; r < r1; r += rpp) {
    float v[8];
    load8(X + r * ldx + c0, vec, c0, C, v);
    store8h<F16>(Y + r * ldy + c0, v);
  }
}

// bf16(A + B): the gradient of a product that left as fp32 rows AND as a 16-bit copy (LinearBNActH, dual output) -- the fp32
// gradient of the one and the bf16 gradient of the other become the dY rows of the backward products in one pass
__global__ __launch_bounds__(EW_TPB) void add_cast_rows_h_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const u16* __restrict__ B, int64_t ldb, int64_t rows, int64_t C,
                                                                 u16* __restrict__ Y, int64_t ldy, int cpb, int rpp) {
  const int ch = threadIdx.x % cpb, rr = threadIdx.x / cpb;
  const int64_t c0 = ((int64_t)blockIdx.y * cpb + ch) * 8;
  if (rr >= rpp || c0 >= ldy) return;
  const bool full = c0 + 8 <= C;
  const bool avec = full && (lda & 3) == 0 && ((uintptr_t)A & 15) == 0;
  const bool bvec = full && (ldb & 7) == 0 && ((uintptr_t)B & 15) == 0;
  const int64_t r0 = (int64_t)blockIdx.x * EW_ROWS;
  const int64_t r1 = r0 + EW_ROWS < rows ? r0 + EW_ROWS : rows;
  for (int64_t r = r0 + rr; r < r1; r += rpp) {
    float a[8], b[8];
    load8(A + r * lda + c0, avec, c0, C, a);
    load8h(B + r * ldb + c0, bvec, c0, C, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = c0 + e < C ? a[e] + b[e] : 0.f;
    store8h<false>(Y + r * ldy + c0, a);
  }
}

// fp16 rows -> bf16 rows (the fp16 mode keeps its activations as fp16 rows for the forward products; the weight-gradient
// product, a bf16 product, takes bf16(fp16(x))): 8 elements per thread and row, same walk as the casts above
__global__ __launch_bounds__(EW_TPB) void f16_to_bf16_rows_kernel(const u16* __restrict__ X, int64_t ldx, int64_t rows, int64_t C,
                                                                  u16* __restrict__ Y, int64_t ldy, int cpb, int rpp) {
  const int ch = threadIdx.x % cpb, rr = threadIdx.x / cpb;
  const int64_t c0 = ((int64_t)blockIdx.y * cpb + ch) * 8;
  if (rr >= rpp || c0 >= ldy) return;
  const bool vec = c0 + 8 <= ldx && (ldx & 7) == 0 && ((uintptr_t)X & 15) == 0;
  const int64_t r0 = (int64_t)blockIdx.x * EW_ROWS;
  const int64_t r1 = r0 + EW_ROWS < rows ? r0 + EW_ROWS : rows;
  for (int64_t r = r0 + rr; r < r1; r += rpp) {
    float v[8];
    const u16* src = X + r * ldx + c0;
    if (vec) {
      const uint4 q = *reinterpret_cast<const uint4*>(src);
      const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = from_h<true>((u16)(qq[e] & 0xffffu));
        v[2 * e + 1] = from_h<true>((u16)(qq[e] >> 16));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? from_h<true>(src[e]) : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? v[e] : 0.f;
    store8h<false>(Y + r * ldy + c0, v);
  }
}

// W[n][k] fp32 -> Wt[k][n] 16-bit (the data-gradient product's "weight"), LDS-tiled 32 x 32; columns [N, ldt) zeroed
template <bool F16>
__global__ __launch_bounds__(256) void transpose_cast_h_kernel(const float* __restrict__ W, int64_t ldw, int64_t N, int64_t K,
                                                               u16* __restrict__ Wt, int64_t ldt) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  const int64_t n0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t n = n0 + ty + 8 * rr, k = k0 + tx;
    tile[ty + 8 * rr][tx] = (n < N && k < K) ? W[n * ldw + k] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t k = k0 + ty + 8 * rr, n = n0 + tx;
    if (k < K && n < ldt) Wt[k * ldt + n] = to_h<F16>(tile[tx][ty + 8 * rr]);
  }
}


// z = act(y * scale + shift) written as 16-bit rows (the expression of ccn_bn_act_fwd, then ONE rounding); padding columns
// [C, ldz) zeroed
template <bool F16>
__global__ __launch_bounds__(EW_TPB) void bn_act_fwd_h_kernel(const float* __restrict__ Y, int64_t ldy, int64_t rows, int64_t C,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              int act, float slope, u16* __restrict__ Z, int64_t ldz, int cpb,
                                                              int rpp) {
  const int ch = threadIdx.x % cpb, rr = threadIdx.x / cpb;
  const int64_t c0 = ((int64_t)blockIdx.y * cpb + ch) * 8;
  if (rr >= rpp || c0 >= ldz) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = c0 + e < C ? scale[c0 + e] : 0.f;
    sh[e] = c0 + e < C ? shift[c0 + e] : 0.f;
  }
  const bool vec = c0 + 8 <= C && (ldy & 3) == 0 && ((uintptr_t)Y & 15) == 0;
  const int64_t r0 = (int64_t)blockIdx.x * EW_ROWS;
  const int64_t r1 = r0 + EW_ROWS < rows ? r0 + EW_ROWS : rows;
  for (int64_t r = r0 + rr; r < r1; r += rpp) {
    float v[8];
    load8(Y + r * ldy + c0, vec, c0, C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? act_fwd_h(v[e] * sc[e] + sh[e], act, slope) : 0.f;
    store8h<F16>(Z + r * ldz + c0, v);
  }
}

// BatchNorm + activation backward, second pass: dY (bf16 rows) from dZ (fp32 or bf16), y and the column sums of the first
// pass -- the expression of bn_act_bwd_apply_kernel, then ONE rounding (both consumers of dY, the data- and the weight-
// gradient product, round it to bf16 anyway in the 16-bit modes).  DZ16: dZ is bf16 (a 16-bit activation's gradient).
template <bool F16, bool DZ16, int YT = 0>
__global__ __launch_bounds__(EW_TPB) void bn_act_bwd_apply_h_kernel(
    const void* __restrict__ dZv, int64_t lddz, const void* __restrict__ Y, int64_t ldy, int64_t rows, int64_t C,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int act, float slope, const double* __restrict__ sums, int training, float inv_n,
    u16* __restrict__ dY, int64_t lddy, float* __restrict__ dgamma, float* __restrict__ dbeta, int acc_params, int cpb, int rpp,
    int pre = 0) {
  const int ch = threadIdx.x % cpb, rr = threadIdx.x / cpb;
  const int64_t c0 = ((int64_t)blockIdx.y * cpb + ch) * 8;
  if (rr >= rpp || c0 >= lddy) return;
  float sc[8], sh[8], mu[8], rs[8], m1[8], m2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t c = c0 + e < C ? c0 + e : C - 1;
    sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; rs[e] = rstd[c];
    if (YT != 0) {      // (sh, mu reused as xa, xb: xhat = t * xa + xb)
      const float isc = sc[e] != 0.f ? 1.f / sc[e] : 0.f;
      const float xa = rs[e] * isc, xb = sc[e] != 0.f ? -(sh[e] * isc + mu[e]) * rs[e] : 0.f;
      sh[e] = xa;
      mu[e] = xb;
    }
    m1[e] = (float)sums[c] * inv_n;
    m2[e] = (float)sums[C + c] * inv_n;
    if (blockIdx.x == 0 && rr == 0 && c0 + e < C) {   // parameter gradients (acc_params: add into gradient-bucket views)
      if (dgamma) dgamma[c] = (acc_params ? dgamma[c] : 0.f) + (float)sums[C + c];
      if (dbeta) dbeta[c] = (acc_params ? dbeta[c] : 0.f) + (float)sums[c];
    }
  }
  const bool full = c0 + 8 <= C;
  const float inv_slope = slope != 0.f ? 1.f / slope : 0.f;
  const bool yvec = y8_vec_ok<YT>(Y, ldy, full);
  const bool gvec = full && (DZ16 ? ((lddz & 7) == 0) : ((lddz & 3) == 0)) && ((uintptr_t)dZv & 15) == 0;
  const int64_t r0 = (int64_t)blockIdx.x * EW_ROWS;
  const int64_t r1 = r0 + EW_ROWS < rows ? r0 + EW_ROWS : rows;
  for (int64_t r = r0 + rr; r < r1; r += rpp) {
    float y[8], gz[8], o[8];
    load_y8<YT>(Y, r * ldy + c0, yvec, c0, C, y);
    if (DZ16) load8h(reinterpret_cast<const u16*>(dZv) + r * lddz + c0, gvec, c0, C, gz);
    else load8(reinterpret_cast<const float*>(dZv) + r * lddz + c0, gvec, c0, C, gz);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float g, xhat;
      if (YT == 0) {
        g = gz[e] * act_grad_h(y[e] * sc[e] + sh[e], act, slope);
        xhat = (y[e] - mu[e]) * rs[e];
      } else {
        const float t = pre ? y[e] : act_inv_h(y[e], act, inv_slope);
        g = gz[e] * act_grad_h(t, act, slope);
        xhat = t * sh[e] + mu[e];
      }
      const float v = training ? sc[e] * (g - m1[e] - xhat * m2[e]) : sc[e] * g;
      o[e] = c0 + e < C ? v : 0.f;
    }
    store8h<F16>(dY + r * lddy + c0, o);
  }
}

// first pass with a bf16 dZ: column sums of g = dZ * act'(u) and of g * xhat; one fp64 partial row per 128-row workgroup
template <bool DZ16 = true, int YT = 0>
__global__ __launch_bounds__(EW_TPB) void bn_act_bwd_reduce_h_kernel(const void* __restrict__ dZ, int64_t lddz,
                                                                     const void* __restrict__ Y, int64_t ldy, int64_t rows,
                                                                     int64_t C, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, int act, float slope,
                                                                     double* __restrict__ partial, int cpb, int rpp, int pre = 0) {
  __shared__ float red[2][EW_TPB * 8];          // [sum | sum * xhat][row group rr][chunk ch][8]
  const int ch = threadIdx.x % cpb, rr = threadIdx.x / cpb;
  const int64_t c0 = ((int64_t)blockIdx.y * cpb + ch) * 8;
  const bool live = rr < rpp && c0 < C;
  float a1[8], a2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a1[e] = a2[e] = 0.f;
  if (live) {
    float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t c = c0 + e < C ? c0 + e : C - 1;
      sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; rs[e] = rstd[c];
      if (YT != 0) {      // (sh, mu reused as xa, xb: xhat = t * xa + xb, see load_y8)
        const float isc = sc[e] != 0.f ? 1.f / sc[e] : 0.f;
        const float xa = rs[e] * isc, xb = sc[e] != 0.f ? -(sh[e] * isc + mu[e]) * rs[e] : 0.f;
        sh[e] = xa;
        mu[e] = xb;
      }
    }
    const bool full = c0 + 8 <= C;
    const float inv_slope = slope != 0.f ? 1.f / slope : 0.f;
    const bool yvec = y8_vec_ok<YT>(Y, ldy, full);
    const bool gvec = full && (DZ16 ? (lddz & 7) == 0 : (lddz & 3) == 0) && ((uintptr_t)dZ & 15) == 0;
    const int64_t r0 = (int64_t)blockIdx.x * EW_ROWS;
    const int64_t r1 = r0 + EW_ROWS < rows ? r0 + EW_ROWS : rows;
    for (int64_t r = r0 + rr; r < r1; r += rpp) {
      float y[8], gz[8];
      load_y8<YT>(Y, r * ldy + c0, yvec, c0, C, y);
      if (DZ16) load8h(reinterpret_cast<const u16*>(dZ) + r * lddz + c0, gvec, c0, C, gz);
      else load8(reinterpret_cast<const float*>(dZ) + r * lddz + c0, gvec, c0, C, gz);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (YT == 0) {
          const float g = gz[e] * act_grad_h(y[e] * sc[e] + sh[e], act, slope);
          a1[e] += g;
          a2[e] += g * ((y[e] - mu[e]) * rs[e]);
        } else {
          const float t = pre ? y[e] : act_inv_h(y[e], act, inv_slope);
          const float g = gz[e] * act_grad_h(t, act, slope);
          a1[e] += g;
          a2[e] += g * (t * sh[e] + mu[e]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][threadIdx.x * 8 + e] = a1[e];
    red[1][threadIdx.x * 8 + e] = a2[e];
  }
  __syncthreads();
  if (rr == 0 && c0 < C) {
    double* dst = partial + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (c0 + e >= C) break;
      double s1 = 0.0, s2 = 0.0;
      for (int q = 0; q < rpp; ++q) {
        s1 += (double)red[0][(q * cpb + ch) * 8 + e];
        s2 += (double)red[1][(q * cpb + ch) * 8 + e];
      }
      dst[c0 + e] = s1;
      dst[C + c0 + e] = s2;
    }
  }
}

// ------------------------------------------------------------------ weight gradient: dW[N x K] += dY[M x N]^T X[M x K]
// Both operands are CONTRACTION-major in HBM (the contraction index m is their row), the 32x32x16 MFMA wants 8 consecutive
// contraction elements per lane: the slice images [64 rows m][128 columns] (256-byte rows, filled by LDS-DMA, 4 rows per
// wave instruction) are read with ds_read_b64_tr_b16 -- per 16-lane group a block of 4 rows x 16 columns, delivered
// column-major: lane i of the group receives column i of the 4 rows.  Two such reads (rows 8h + 0..3 and 8h + 4..7 of the
// k-step) are a lane's operand of one MFMA.  16-byte chunk c of image row R sits at chunk c ^ f(R), f(R) = ((R & 3) << 2) |
// ((R >> 2) & 3) (applied to the DMA source column and to the read address): the four rows of a block, and the two blocks
// of a 32-lane half, then fall on different banks.
// Work decomposition as the fp32 kernel (ccn_gemm_tn.hip): a workgroup owns a (128 x 128 tile of dW, chunk of rows) item,
// partial tiles go to caller-owned slabs, ccn_tn_reduce-style summation in chunk order (deterministic).
constexpr int HT_TPB = 256;
constexpr int HT_T = 128;     // tile edge (columns of both slice images)
constexpr int HT_SL = 64;     // contraction rows per slice (4 MFMA k-steps)

// (transposed-read results are held as 64-bit scalars: an ext_vector(2) inline-asm output came back with its second dword
// treated as a copy of the first by hipcc 7.2 -- v_mov v89, v88 in front of the MFMAs)
using u64 = unsigned long long;

// two fp16 values -> their bf16 roundings (the conversion of f16_to_bf16_rows_kernel, on an MFMA operand dword)
__device__ __forceinline__ uint32_t f16x2_to_bf16x2(uint32_t w) {
  const float lo = from_h<true>((u16)(w & 0xffffu)), hi = from_h<true>((u16)(w >> 16));
  return (uint32_t)to_h<false>(lo) | ((uint32_t)to_h<false>(hi) << 16);
}

// XF16 (fp16 mode): X holds fp16 rows and is converted to bf16 on the way into the MFMA -- the product of bf16(fp16(x)) that
// ccn_f16_to_bf16_rows + this kernel give, without the conversion pass (78 per step of BASELINE configs[4], 5 ms) and without
// keeping a bf16 copy alive; the matrix pipe is 17 % busy here, the 24 extra VALU per k-step are free
template <int EPI, bool XF16>   // EPI 0: slab store, 2: plain add into dW (one chunk)
__global__ __launch_bounds__(HT_TPB, 2) void gemm_h_tn_kernel(const u16* __restrict__ A, int64_t lda,
                                                              const u16* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                              int64_t ldc, int64_t M, int64_t N, int64_t K, int tiles_k,
                                                              int tiles, int split, int64_t slices_per_chunk, int64_t n_ids,
                                                              int xcd_order, float* __restrict__ slabs, int64_t b_extent) {
  constexpr int AF = HT_SL * HT_T, STAGE = 2 * AF;      // 16-bit elements: 16 KB per operand image, 32 KB per stage
  __shared__ __attribute__((aligned(16))) u16 lds[2 * STAGE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: SGPR addressing)
  const int i = lane & 31, h = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  // transposed-read geometry: lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p + 3 of the group's block
  const int L = lane & 15, q = L >> 2, p = L & 3, gsel = (lane >> 4) & 1;
  uint32_t a_addr[2][2], b_addr[2][2];     // [column block ta / tb][row half]: bytes inside a stage, k-step 0
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int R = 8 * h + 4 * half + q;                        // (+ 16 s per k-step: leaves f(R) unchanged)
    const int f = ((R & 3) << 2) | ((R >> 2) & 3);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int cha = 8 * wn + 4 * t + 2 * gsel + (p >> 1), chb = 8 * wk + 4 * t + 2 * gsel + (p >> 1);
      a_addr[t][half] = (uint32_t)(256 * R + 16 * (cha ^ f) + 8 * (p & 1));
      b_addr[t][half] = (uint32_t)(AF * 2 + 256 * R + 16 * (chb ^ f) + 8 * (p & 1));
    }
  }
  const int64_t total_slices = (M + HT_SL - 1) / HT_SL;

  for (int64_t id = blockIdx.x; id < n_ids; id += gridDim.x) {
    int64_t chunk, tile;
    if (xcd_order) {
      const int64_t slot = id >> 3;
      chunk = (slot / tiles) * 8 + (id & 7);
      tile = slot % tiles;
    } else {
      chunk = id / tiles;
      tile = id % tiles;
    }
    if (chunk >= split) continue;
    const int64_t s_beg = chunk * slices_per_chunk;
    int64_t s_end = s_beg + slices_per_chunk;
    if (s_end > total_slices) s_end = total_slices;
    const int64_t n0 = (tile / tiles_k) * HT_T, k0 = (tile % tiles_k) * HT_T;

    f32x16 acc[2][2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ta][tb][r] = 0.f;

    if (s_beg < s_end) {
      // ---- DMA sources: instruction g = 4 wave + qn copies image rows 4 g .. 4 g + 3 (lane: row lane >> 4, LDS chunk lane & 15)
      int64_t a_col[4], b_col[4];
#pragma unroll
      for (int qn = 0; qn < 4; ++qn) {
        const int f = ((lane >> 4) << 2) | qn;                 // f(R) of image row R = 4 (4 wave + qn) + (lane >> 4)
        const int64_t ca = n0 + 8 * ((lane & 15) ^ f), cb = k0 + 8 * ((lane & 15) ^ f);
        a_col[qn] = ca <= lda - 8 ? ca : lda - 8;              // (clamped columns only feed elements that are never stored)
        b_col[qn] = cb <= b_extent - 8 ? cb : b_extent - 8;    // (b_extent = ldb, or K when the X rows overlap: ccn_conv_rows_tn_h)
      }
      uint32_t a_off32[4], b_off32[4];      // lane offsets inside a slice: (row * ld + column) bytes
#pragma unroll
      for (int qn = 0; qn < 4; ++qn) {
        const int64_t r = 4 * (4 * wave + qn) + (lane >> 4);
        a_off32[qn] = (uint32_t)((r * lda + a_col[qn]) * 2);
        b_off32[qn] = (uint32_t)((r * ldb + b_col[qn]) * 2);
      }
      auto issue = [&](int64_t sl, int stage) {
        u16* st = lds + stage * STAGE;
        const int64_t mrow0 = sl * HT_SL;
        if (mrow0 + HT_SL <= M) {      // scalar slice base + 32-bit lane offsets; only the matrix's last slice clamps rows per lane
          const char* const a_sl = reinterpret_cast<const char*>(A + mrow0 * lda);
          const char* const b_sl = reinterpret_cast<const char*>(B + mrow0 * ldb);
#pragma unroll
          for (int qn = 0; qn < 4; ++qn) glds16h(reinterpret_cast<const u16*>(a_sl + a_off32[qn]), st + (4 * wave + qn) * 512);
#pragma unroll
          for (int qn = 0; qn < 4; ++qn) glds16h(reinterpret_cast<const u16*>(b_sl + b_off32[qn]), st + AF + (4 * wave + qn) * 512);
          return;
        }
#pragma unroll
        for (int qn = 0; qn < 4; ++qn) {
          int64_t row = mrow0 + 4 * (4 * wave + qn) + (lane >> 4);
          row = row < M ? row : M - 1;
          glds16h(A + row * lda + a_col[qn], st + (4 * wave + qn) * 512);
        }
#pragma unroll
        for (int qn = 0; qn < 4; ++qn) {
          int64_t row = mrow0 + 4 * (4 * wave + qn) + (lane >> 4);
          row = row < M ? row : M - 1;
          glds16h(B + row * ldb + b_col[qn], st + AF + (4 * wave + qn) * 512);
        }
      };
      __builtin_amdgcn_s_barrier();          // every wave is done with the previous item's LDS reads
      issue(s_beg, 0);
      int stage = 0;
      for (int64_t sl = s_beg; sl < s_end; ++sl, stage ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slice sl has landed (this wave's share)
        __builtin_amdgcn_s_barrier();                      // ... and everybody else's; the other stage is free again
        if (sl + 1 < s_end) issue(sl + 1, stage ^ 1);
        const uint32_t sb = lds_base + (uint32_t)(stage * STAGE * 2);
        const int64_t rl64 = M - sl * HT_SL;
        const int rows_left = rl64 < HT_SL ? (int)rl64 : HT_SL;
        u64 fa[2][2][2], fb[2][2][2];      // [k-step parity][column block][row half]
        // (immediate offsets must be literals: the four k-steps are written out)
#define HT_READ(S, PAR)                                                                                                         \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int half = 0; half < 2; ++half) {                        \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[PAR][t][half]) : "v"(sb + a_addr[t][half]), "n"((S) * 4096) : "memory"); \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[PAR][t][half]) : "v"(sb + b_addr[t][half]), "n"((S) * 4096) : "memory"); \
  }
#define HT_MFMA(S, PAR)                                                                                                  \
  {                                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    f32x4 oa[2], ob[2];                                                                                                  \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                      \
      uint32_t l0 = (uint32_t)fa[PAR][t][0], l1 = (uint32_t)(fa[PAR][t][0] >> 32);                                      \
      uint32_t h0 = (uint32_t)fa[PAR][t][1], h1 = (uint32_t)(fa[PAR][t][1] >> 32);                                      \
      if (rows_left < HT_SL) {      /* rows beyond M (clamped source rows): their dY elements are zeroed */              \
        const int m = 16 * (S) + 8 * h;                                                                                  \
        l0 = (m + 0 < rows_left ? l0 & 0xffffu : 0u) | (m + 1 < rows_left ? l0 & 0xffff0000u : 0u);                     \
        l1 = (m + 2 < rows_left ? l1 & 0xffffu : 0u) | (m + 3 < rows_left ? l1 & 0xffff0000u : 0u);                     \
        h0 = (m + 4 < rows_left ? h0 & 0xffffu : 0u) | (m + 5 < rows_left ? h0 & 0xffff0000u : 0u);                     \
        h1 = (m + 6 < rows_left ? h1 & 0xffffu : 0u) | (m + 7 < rows_left ? h1 & 0xffff0000u : 0u);                     \
      }                                                                                                                  \
      oa[t] = f32x4{__builtin_bit_cast(float, l0), __builtin_bit_cast(float, l1), __builtin_bit_cast(float, h0),        \
                    __builtin_bit_cast(float, h1)};                                                                      \
      uint32_t x0 = (uint32_t)fb[PAR][t][0], x1 = (uint32_t)(fb[PAR][t][0] >> 32);                                      \
      uint32_t x2 = (uint32_t)fb[PAR][t][1], x3 = (uint32_t)(fb[PAR][t][1] >> 32);                                      \
      if (XF16) {                                                                                                        \
        x0 = f16x2_to_bf16x2(x0); x1 = f16x2_to_bf16x2(x1); x2 = f16x2_to_bf16x2(x2); x3 = f16x2_to_bf16x2(x3);          \
      }                                                                                                                  \
      ob[t] = f32x4{__builtin_bit_cast(float, x0), __builtin_bit_cast(float, x1), __builtin_bit_cast(float, x2),        \
                    __builtin_bit_cast(float, x3)};                                                                      \
    }                                                                                                                    \
    _Pragma("unroll") for (int ta = 0; ta < 2; ++ta) _Pragma("unroll") for (int tb = 0; tb < 2; ++tb)                    \
        acc[ta][tb] = mfma_h<false>(oa[ta], ob[tb], acc[ta][tb]);                                                        \
  }
        HT_READ(0, 0)
        HT_READ(1, 1)
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]),
                     "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]) : : "memory");
        HT_MFMA(0, 0)
        HT_READ(2, 0)
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]),
                     "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]) : : "memory");
        HT_MFMA(1, 1)
        HT_READ(3, 1)
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]),
                     "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]) : : "memory");
        HT_MFMA(2, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]),
                     "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]) : : "memory");
        HT_MFMA(3, 1)
#undef HT_READ
#undef HT_MFMA
      }
    }

    // ---- epilogue.  C/D layout: col = lane & 31 (k), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (n)
    if (EPI == 0) {
      float* slab = slabs + ((int64_t)tile * split + chunk) * (HT_T * HT_T);
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int nl = wn * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            slab[nl * HT_T + wk * 64 + tb * 32 + i] = acc[ta][tb][r];
          }
    } else {
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t n = n0 + wn * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t k = k0 + wk * 64 + tb * 32 + i;
            if (n < N && k < K) C[n * ldc + k] += acc[ta][tb][r];
          }
    }
  }
}

// dW[n][k] += sum over the chunks of the tile's slabs, in chunk order (the reduction of ccn_gemm_tn.hip, 128 x 128 tiles)
constexpr int HRED_WAVES = 16;
__global__ __launch_bounds__(64 * HRED_WAVES) void tn_h_reduce_kernel(const float* __restrict__ slabs, int split, int tiles_k,
                                                                      int tiles, int64_t N, int64_t K, float* __restrict__ C,
                                                                      int64_t ldc) {
  constexpr int Q = HT_T * HT_T / 4;                  // float4 per slab
  __shared__ float4 red[HRED_WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  const bool live = e < (int64_t)tiles * Q;
  const int64_t tile = live ? e / Q : 0;
  const int within = live ? (int)(e - tile * Q) : 0;
  const float* src = slabs + tile * split * (int64_t)(HT_T * HT_T) + (int64_t)within * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    for (int c = w; c < split; c += 4 * HRED_WAVES) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = c + u * HRED_WAVES;
        v[u] = cc < split ? *reinterpret_cast<const float4*>(src + (int64_t)cc * (HT_T * HT_T)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s.x += v[u].x;
        s.y += v[u].y;
        s.z += v[u].z;
        s.w += v[u].w;
      }
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w != 0 || !live) return;
#pragma unroll
  for (int u = 1; u < HRED_WAVES; ++u) {
    const float4 t = red[u][lane];
    s.x += t.x;
    s.y += t.y;
    s.z += t.z;
    s.w += t.w;
  }
  const int64_t n = (tile / tiles_k) * HT_T + (within * 4) / HT_T;
  const int64_t k = (tile % tiles_k) * HT_T + (within * 4) % HT_T;
  if (n >= N || k >= K) return;
  float* dst = C + n * ldc + k;
  dst[0] += s.x;
  if (k + 1 < K) dst[1] += s.y;
  if (k + 2 < K) dst[2] += s.z;
  if (k + 3 < K) dst[3] += s.w;
}

struct HtPlan {
  int tiles_n, tiles_k, tiles, split;
  int64_t slices_per_chunk, n_ids, slab_floats;
  bool xcd;
};

inline HtPlan ht_plan(int64_t M, int64_t N, int64_t K) {
  HtPlan p;
  p.tiles_n = (int)((N + HT_T - 1) / HT_T);
  p.tiles_k = (int)((K + HT_T - 1) / HT_T);
  p.tiles = p.tiles_n * p.tiles_k;
  const int64_t slices = (M + HT_SL - 1) / HT_SL;
  // ~1024 work items (two workgroups per CU, two rounds: the products are HBM-bound, tails are cheap) of >= 4 slices each
  int64_t split = (1024 + p.tiles - 1) / p.tiles;
  if (split > slices / 4) split = slices / 4;
  if (split < 1) split = 1;
  p.slices_per_chunk = (slices + split - 1) / split;
  split = (slices + p.slices_per_chunk - 1) / p.slices_per_chunk;
  p.split = (int)split;
  p.xcd = split >= 8 && p.tiles > 1;
  p.n_ids = p.xcd ? (int64_t)p.tiles * ((split + 7) / 8 * 8) : (int64_t)p.tiles * split;
  p.slab_floats = split > 1 ? (int64_t)p.tiles * split * HT_T * HT_T : 0;
  return p;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static std::atomic<int> g_h_opt{0};      // diagnostics (ccn_gemm_h_opt)

template <bool F16, bool OUT16>
int launch_nt_h(const u16* A, int64_t lda, const u16* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M,
                int64_t N, int64_t K, double* colstats, hipStream_t s, int64_t a_extent) {
  if constexpr (!OUT16) {
    // fp32 result, N <= 64 or a last 128-wide tile at most half used (192, 320): 128 x 64 tiles, three workgroups per CU.
    // Measured stand-alone (tools/bench_gemm_h_tiles.py, profiles/r04_gemm_h_tiles.txt): 557 k x 64 x 64 1.59x, 1.87 M x 192 x
    // 128 1.08x; at N = 256 ... 1024 the 128-wide form wins (0.85-0.97x: the A tile is read by twice as many column tiles),
    // so the "more, smaller workgroups" lever of DESIGN r3 section 7 does not exist at the network's main widths.
    // A/B hooks (ccn_gemm_h_opt): bit 4 = never, bit 5 = always.
    const bool narrow = N <= 64 || (N % HB_BN != 0 && N % HB_BN <= 64);
    if (((narrow && g_h_opt == 0) || g_h_opt == 32)) {
      const int64_t gm6 = (M + HB_BM - 1) / HB_BM, gn6 = (N + 63) / 64;
      const int64_t tiles6 = gm6 * gn6;
      if (tiles6 < ((int64_t)1 << 31)) {
        const int64_t grid6 = tiles6 < 768 ? tiles6 : 768;
        hipLaunchKernelGGL((gemm_h_pair_kernel<F16, false, false, 64>), dim3((unsigned)grid6), dim3(HB_TPB), 0, s, A, lda, W, ldw,
                           bias, Y, ldy, M, N, K, tiles6, gn6, 1, colstats, 0, a_extent, (const float*)nullptr, (const float*)nullptr, 0,
                           0.f, (void*)nullptr, (int64_t)0);
        return CCN_OK;
      }
    }
  }
  const int64_t gm = (M + HB_BM - 1) / HB_BM, gn = (N + HB_BN - 1) / HB_BN;
  const int64_t tiles = gm * gn;
  if (tiles >= ((int64_t)1 << 31)) {
    ccn_set_error("gemm_nt_h: more than 2^31 output tiles");
    return CCN_ERR_ARG;
  }
  const int64_t grid = tiles < 512 ? tiles : 512;  // two workgroups per CU
  if (g_h_opt != 0)
    hipLaunchKernelGGL((gemm_h_pair_kernel<F16, OUT16, true>), dim3((unsigned)grid), dim3(HB_TPB), 0, s, A, lda, W, ldw, bias, Y,
                       ldy, M, N, K, tiles, gn, 1, colstats, g_h_opt, a_extent, (const float*)nullptr, (const float*)nullptr, 0, 0.f, (void*)nullptr, (int64_t)0);
  else
    hipLaunchKernelGGL((gemm_h_pair_kernel<F16, OUT16>), dim3((unsigned)grid), dim3(HB_TPB), 0, s, A, lda, W, ldw, bias, Y, ldy,
                       M, N, K, tiles, gn, 1, colstats, 0, a_extent, (const float*)nullptr, (const float*)nullptr, 0, 0.f, (void*)nullptr, (int64_t)0);
  return CCN_OK;
}

// the fused forms (FUSE 1: statistics only, FUSE 2: BatchNorm + activation in the epilogue); tile choice as launch_nt_h
template <bool F16, bool OUT16, int FUSE>
int launch_nt_h_fused(const u16* A, int64_t lda, const u16* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M,
                      int64_t N, int64_t K, double* colstats, hipStream_t s, const float* scale, const float* shift, int act,
                      float slope, void* T = nullptr, int64_t ldt = 0) {
  if constexpr (!OUT16) {
    const bool narrow = N <= 64 || (N % HB_BN != 0 && N % HB_BN <= 64);
    if (narrow) {
      const int64_t gm6 = (M + HB_BM - 1) / HB_BM, gn6 = (N + 63) / 64;
      const int64_t tiles6 = gm6 * gn6;
      if (tiles6 < ((int64_t)1 << 31)) {
        const int64_t grid6 = tiles6 < 768 ? tiles6 : 768;
        hipLaunchKernelGGL((gemm_h_pair_kernel<F16, false, false, 64, FUSE>), dim3((unsigned)grid6), dim3(HB_TPB), 0, s, A, lda, W,
                           ldw, bias, Y, ldy, M, N, K, tiles6, gn6, 1, colstats, 0, lda, scale, shift, act, slope, (void*)nullptr, (int64_t)0);
        return CCN_OK;
      }
    }
  }
  const int64_t gm = (M + HB_BM - 1) / HB_BM, gn = (N + HB_BN - 1) / HB_BN;
  const int64_t tiles = gm * gn;
  if (tiles >= ((int64_t)1 << 31)) {
    ccn_set_error("gemm_nt_h: more than 2^31 output tiles");
    return CCN_ERR_ARG;
  }
  const int64_t grid = tiles < 512 ? tiles : 512;
  if constexpr (OUT16 && FUSE == 2) {
    if (T != nullptr) {
      hipLaunchKernelGGL((gemm_h_pair_kernel<F16, true, false, HB_BN, 3>), dim3((unsigned)grid), dim3(HB_TPB), 0, s, A, lda, W, ldw,
                         bias, Y, ldy, M, N, K, tiles, gn, 1, colstats, 0, lda, scale, shift, act, slope, T, ldt);
      return CCN_OK;
    }
  }
  hipLaunchKernelGGL((gemm_h_pair_kernel<F16, OUT16, false, HB_BN, FUSE>), dim3((unsigned)grid), dim3(HB_TPB), 0, s, A, lda, W, ldw,
                     bias, Y, ldy, M, N, K, tiles, gn, 1, colstats, 0, lda, scale, shift, act, slope, (void*)nullptr, (int64_t)0);
  return CCN_OK;
}

}  // namespace

extern "C" {

int ccn_gemm_h_opt(int opt) {
  g_h_opt = opt;
  return CCN_OK;
}

static int gemm_nt_h_impl(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M,
                          int64_t N, int64_t K, double* colstats, int f16, int out16, void* stream, int64_t a_extent) {
  CCN_REQUIRE(A && W && Y, "gemm_nt_h: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && a_extent >= K && lda >= 8 && ldw >= K && ldy >= N, "gemm_nt_h: bad sizes");
  CCN_REQUIRE(aligned16(A) && aligned16(W) && lda % 8 == 0 && ldw % 8 == 0,
              "gemm_nt_h: 16-bit operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  CCN_REQUIRE(!out16 || (colstats == nullptr && bias == nullptr && ((uintptr_t)Y & 7) == 0 && ldy % 4 == 0),
              "gemm_nt_h: a 16-bit result takes no bias / statistics and needs 8-byte aligned rows");
  if (M == 0) return CCN_OK;
  hipStream_t s = (hipStream_t)stream;
  const u16* a = (const u16*)A;
  const u16* w = (const u16*)W;
  int rc;
  if (f16) rc = out16 ? launch_nt_h<true, true>(a, lda, w, ldw, bias, Y, ldy, M, N, K, colstats, s, a_extent)
                      : launch_nt_h<true, false>(a, lda, w, ldw, bias, Y, ldy, M, N, K, colstats, s, a_extent);
  else rc = out16 ? launch_nt_h<false, true>(a, lda, w, ldw, bias, Y, ldy, M, N, K, colstats, s, a_extent)
                  : launch_nt_h<false, false>(a, lda, w, ldw, bias, Y, ldy, M, N, K, colstats, s, a_extent);
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_nt_h");
  return CCN_OK;
}

int ccn_gemm_nt_h(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M,
                  int64_t N, int64_t K, double* colstats, int f16, int out16, void* stream) {
  CCN_REQUIRE(lda >= K, "gemm_nt_h: bad sizes (lda < K)");
  return gemm_nt_h_impl(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, f16, out16, stream, lda);
}

int ccn_gemm_nt_h_stats(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M, int64_t N, int64_t K,
                        double* colstats, int f16, void* stream) {
  CCN_REQUIRE(A && W && colstats, "gemm_nt_h_stats: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lda >= K && lda >= 8 && ldw >= K, "gemm_nt_h_stats: bad sizes");
  CCN_REQUIRE(aligned16(A) && aligned16(W) && lda % 8 == 0 && ldw % 8 == 0,
              "gemm_nt_h_stats: 16-bit operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  if (M == 0) return CCN_OK;
  const u16* a = (const u16*)A;
  const u16* w = (const u16*)W;
  const int rc = f16 ? launch_nt_h_fused<true, false, 1>(a, lda, w, ldw, bias, nullptr, N, M, N, K, colstats, (hipStream_t)stream, nullptr, nullptr, 0, 0.f)
                     : launch_nt_h_fused<false, false, 1>(a, lda, w, ldw, bias, nullptr, N, M, N, K, colstats, (hipStream_t)stream, nullptr, nullptr, 0, 0.f);
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_nt_h_stats");
  return CCN_OK;
}

int ccn_gemm_nt_h_bnact(const void* A, int64_t lda, const void* W, int64_t ldw, const float* scale, const float* shift, int act,
                        float slope, void* Z, int64_t ldz, void* T, int64_t ldt, int64_t M, int64_t N, int64_t K, int f16, int out16,
                        void* stream) {
  CCN_REQUIRE(A && W && Z && scale && shift, "gemm_nt_h_bnact: null pointer");
  CCN_REQUIRE(T == nullptr || (out16 && ldt >= N && ldt % 4 == 0 && ((uintptr_t)T & 7) == 0),
              "gemm_nt_h_bnact: the pre-activation output exists in the 16-bit form only (8-byte aligned rows)");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lda >= K && lda >= 8 && ldw >= K && ldz >= N, "gemm_nt_h_bnact: bad sizes");
  CCN_REQUIRE(aligned16(A) && aligned16(W) && lda % 8 == 0 && ldw % 8 == 0,
              "gemm_nt_h_bnact: 16-bit operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  CCN_REQUIRE(!out16 || (((uintptr_t)Z & 7) == 0 && ldz % 4 == 0), "gemm_nt_h_bnact: a 16-bit result needs 8-byte aligned rows");
  CCN_REQUIRE(act == CCN_ACT_NONE || act == CCN_ACT_RELU || act == CCN_ACT_LEAKY, "gemm_nt_h_bnact: unknown activation");
  if (M == 0) return CCN_OK;
  hipStream_t s = (hipStream_t)stream;
  const u16* a = (const u16*)A;
  const u16* w = (const u16*)W;
  int rc;
  if (f16) rc = out16 ? launch_nt_h_fused<true, true, 2>(a, lda, w, ldw, nullptr, Z, ldz, M, N, K, nullptr, s, scale, shift, act, slope, T, ldt)
                      : launch_nt_h_fused<true, false, 2>(a, lda, w, ldw, nullptr, Z, ldz, M, N, K, nullptr, s, scale, shift, act, slope);
  else rc = out16 ? launch_nt_h_fused<false, true, 2>(a, lda, w, ldw, nullptr, Z, ldz, M, N, K, nullptr, s, scale, shift, act, slope, T, ldt)
                  : launch_nt_h_fused<false, false, 2>(a, lda, w, ldw, nullptr, Z, ldz, M, N, K, nullptr, s, scale, shift, act, slope);
  if (rc) return rc;
  CCN_LAUNCH_OK("gemm_nt_h_bnact");
  return CCN_OK;
}

// Implicit-GEMM curve convolution on a 16-bit row sequence (the 16-bit form of ccn_conv_rows_nt): row i of the shifted-row
// matrix is the contiguous span of K = taps * lda elements starting at A + i * lda -- overlapping operand rows, read in place
// from a sequence with taps / 2 zero halo rows at both ends of the same allocation.
int ccn_conv_rows_nt_h(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t M,
                       int64_t N, int64_t K, double* colstats, int f16, int out16, void* stream) {
  CCN_REQUIRE(lda > 0 && K % lda == 0, "conv_rows_nt_h: K must be a whole number of sequence rows");
  return gemm_nt_h_impl(A, lda, W, ldw, bias, Y, ldy, M, N, K, colstats, f16, out16, stream, K);
}

size_t ccn_gemm_tn_h_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (size_t)ht_plan(M, N, K).slab_floats * sizeof(float);
}

static int gemm_tn_h_impl(const void* dY, int64_t lddy, const void* X, int x_f16, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                          int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream, int64_t x_extent = 0) {
  // x_extent: 16-bit elements readable from the start of an X row (0 = ldx; K when the X rows overlap: ccn_conv_rows_tn_h)
  if (x_extent == 0) x_extent = ldx;
  CCN_REQUIRE(dY && X && dW, "gemm_tn_h: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && x_extent >= K && lddw >= K, "gemm_tn_h: bad sizes");
  CCN_REQUIRE(aligned16(dY) && aligned16(X) && lddy % 8 == 0 && ldx % 8 == 0 && lddy >= 8 && ldx >= 8,
              "gemm_tn_h: bf16 operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  if (M == 0) return CCN_OK;
  hipStream_t s = (hipStream_t)stream;
  const HtPlan p = ht_plan(M, N, K);
  const int64_t grid = p.n_ids < 512 ? p.n_ids : 512;
  const u16* a = (const u16*)dY;
  const u16* b = (const u16*)X;
#define CCN_TN_H(EPI_, SLABS_)                                                                                                \
  do {                                                                                                                        \
    if (x_f16)                                                                                                                \
      hipLaunchKernelGGL((gemm_h_tn_kernel<EPI_, true>), dim3((unsigned)grid), dim3(HT_TPB), 0, s, a, lddy, b, ldx, dW, lddw,  \
                         M, N, K, p.tiles_k, p.tiles, p.split, p.slices_per_chunk, p.n_ids, p.xcd ? 1 : 0, SLABS_, x_extent); \
    else                                                                                                                      \
      hipLaunchKernelGGL((gemm_h_tn_kernel<EPI_, false>), dim3((unsigned)grid), dim3(HT_TPB), 0, s, a, lddy, b, ldx, dW, lddw, \
                         M, N, K, p.tiles_k, p.tiles, p.split, p.slices_per_chunk, p.n_ids, p.xcd ? 1 : 0, SLABS_, x_extent); \
  } while (0)
  if (p.split == 1) {
    CCN_TN_H(2, (float*)nullptr);
  } else {
    CCN_REQUIRE(workspace != nullptr && workspace_bytes >= (size_t)p.slab_floats * sizeof(float) && aligned16(workspace),
                "gemm_tn_h: workspace too small (%zu < %zu bytes) or unaligned", workspace_bytes,
                (size_t)p.slab_floats * sizeof(float));
    float* slabs = (float*)workspace;
    CCN_TN_H(0, slabs);
    const int64_t work = (int64_t)p.tiles * (HT_T * HT_T / 4);
    hipLaunchKernelGGL(tn_h_reduce_kernel, dim3((unsigned)((work + 63) / 64)), dim3(64 * HRED_WAVES), 0, s, slabs, p.split,
                       p.tiles_k, p.tiles, N, K, dW, lddw);
  }
#undef CCN_TN_H
  CCN_LAUNCH_OK("gemm_tn_h");
  return CCN_OK;
}

int ccn_gemm_tn_h(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw, int64_t M, int64_t N,
                  int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_tn_h_impl(dY, lddy, X, 0, ldx, dW, lddw, M, N, K, workspace, workspace_bytes, stream);
}

// Weight gradient of the implicit-GEMM curve convolution on 16-bit row sequences (the 16-bit form of ccn_conv_rows_tn): X row i
// is the span of K = taps * ldx elements starting at X + i * ldx (overlapping rows); x_f16 != 0: X holds fp16 rows.
int ccn_conv_rows_tn_h(const void* dY, int64_t lddy, const void* X, int64_t ldx, int x_f16, float* dW, int64_t lddw, int64_t M,
                       int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(ldx > 0 && K % ldx == 0, "conv_rows_tn_h: K must be a whole number of sequence rows");
  return gemm_tn_h_impl(dY, lddy, X, x_f16 ? 1 : 0, ldx, dW, lddw, M, N, K, workspace, workspace_bytes, stream, K);
}

// ... with X as fp16 rows (the fp16 mode's forward operand): dW += dY^T bf16(X), converted inside the kernel
int ccn_gemm_tn_h_xf16(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw, int64_t M, int64_t N,
                       int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_tn_h_impl(dY, lddy, X, 1, ldx, dW, lddw, M, N, K, workspace, workspace_bytes, stream);
}

int ccn_cast_rows_h(const float* X, int64_t ldx, int64_t rows, int64_t C, void* Y, int64_t ldy, int f16, void* stream) {
  CCN_REQUIRE(X && Y && rows >= 0 && C > 0 && ldx >= C && ldy >= C && ldy % 8 == 0 && aligned16(Y), "cast_rows_h: bad arguments");
  if (rows == 0) return CCN_OK;
  const EwGeom g = ew_geom(ldy);
  const dim3 grid((unsigned)ccn_blocks(rows, EW_ROWS), g.gy);
  if (f16) hipLaunchKernelGGL(cast_rows_h_kernel<true>, grid, dim3(EW_TPB), 0, (hipStream_t)stream, X, ldx, rows, C, (u16*)Y, ldy, g.cpb, g.rpp);
  else hipLaunchKernelGGL(cast_rows_h_kernel<false>, grid, dim3(EW_TPB), 0, (hipStream_t)stream, X, ldx, rows, C, (u16*)Y, ldy, g.cpb, g.rpp);
  CCN_LAUNCH_OK("cast_rows_h");
  return CCN_OK;
}

int ccn_add_cast_rows_h(const float* A, int64_t lda, const void* B, int64_t ldb, int64_t rows, int64_t C, void* Y, int64_t ldy,
                        void* stream) {
  CCN_REQUIRE(A && B && Y && rows >= 0 && C > 0 && lda >= C && ldb >= C && ldy >= C && ldy % 8 == 0 && aligned16(Y),
              "add_cast_rows_h: bad arguments");
  if (rows == 0) return CCN_OK;
  const EwGeom g = ew_geom(ldy);
  hipLaunchKernelGGL(add_cast_rows_h_kernel, dim3((unsigned)ccn_blocks(rows, EW_ROWS), g.gy), dim3(EW_TPB), 0, (hipStream_t)stream,
                     A, lda, (const u16*)B, ldb, rows, C, (u16*)Y, ldy, g.cpb, g.rpp);
  CCN_LAUNCH_OK("add_cast_rows_h");
  return CCN_OK;
}

int ccn_f16_to_bf16_rows(const void* X, int64_t ldx, int64_t rows, int64_t C, void* Y, int64_t ldy, void* stream) {
  CCN_REQUIRE(X && Y && rows >= 0 && C > 0 && ldx >= C && ldy >= C && ldy % 8 == 0 && aligned16(Y), "f16_to_bf16_rows: bad arguments");
  if (rows == 0) return CCN_OK;
  const EwGeom g = ew_geom(ldy);
  hipLaunchKernelGGL(f16_to_bf16_rows_kernel, dim3((unsigned)ccn_blocks(rows, EW_ROWS), g.gy), dim3(EW_TPB), 0, (hipStream_t)stream,
                     (const u16*)X, ldx, rows, C, (u16*)Y, ldy, g.cpb, g.rpp);
  CCN_LAUNCH_OK("f16_to_bf16_rows");
  return CCN_OK;
}

int ccn_transpose_cast_h(const float* W, int64_t ldw, int64_t N, int64_t K, void* Wt, int64_t ldt, int f16, void* stream) {
  CCN_REQUIRE(W && Wt && N > 0 && K > 0 && ldw >= K && ldt >= N && ldt % 8 == 0 && aligned16(Wt), "transpose_cast_h: bad arguments");
  const dim3 grid((unsigned)((ldt + 31) / 32), (unsigned)((K + 31) / 32));
  if (f16) hipLaunchKernelGGL(transpose_cast_h_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, (u16*)Wt, ldt);
  else hipLaunchKernelGGL(transpose_cast_h_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, (u16*)Wt, ldt);
  CCN_LAUNCH_OK("transpose_cast_h");
  return CCN_OK;
}

int ccn_bn_act_fwd_h(const float* Y, int64_t ldy, int64_t rows, int64_t C, const float* scale, const float* shift, int act,
                     float slope, void* Z, int64_t ldz, int f16, void* stream) {
  CCN_REQUIRE(Y && Z && scale && shift && rows >= 0 && C > 0 && ldy >= C && ldz >= C && ldz % 8 == 0 && aligned16(Z),
              "bn_act_fwd_h: bad arguments");
  if (rows == 0) return CCN_OK;
  const EwGeom g = ew_geom(ldz);
  const dim3 grid((unsigned)ccn_blocks(rows, EW_ROWS), g.gy);
  if (f16) hipLaunchKernelGGL(bn_act_fwd_h_kernel<true>, grid, dim3(EW_TPB), 0, (hipStream_t)stream, Y, ldy, rows, C, scale, shift, act, slope, (u16*)Z, ldz, g.cpb, g.rpp);
  else hipLaunchKernelGGL(bn_act_fwd_h_kernel<false>, grid, dim3(EW_TPB), 0, (hipStream_t)stream, Y, ldy, rows, C, scale, shift, act, slope, (u16*)Z, ldz, g.cpb, g.rpp);
  CCN_LAUNCH_OK("bn_act_fwd_h");
  return CCN_OK;
}

int ccn_bn_act_bwd_reduce_h(const void* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                            const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                            double* sums, void* stream) {
  // dZ: bf16 rows.  sums: 2*C totals followed by [ccn_stats_rows(rows)][2*C] doubles of scratch (as ccn_bn_act_bwd_reduce)
  CCN_REQUIRE(dZ && Y && sums && rows > 0 && C > 0 && lddz >= C && ldy >= C, "bn_act_bwd_reduce_h: bad arguments");
  const int64_t nparts = ccn_stats_rows(rows);
  double* partial = sums + 2 * C;
  const EwGeom g = ew_geom((C + 7) / 8 * 8);
  hipLaunchKernelGGL((bn_act_bwd_reduce_h_kernel<true, 0>), dim3((unsigned)nparts, g.gy), dim3(EW_TPB), 0, (hipStream_t)stream,
                     dZ, lddz, (const void*)Y, ldy, rows, C, scale, shift, mean, rstd, act, slope, partial, g.cpb, g.rpp);
  CCN_LAUNCH_OK("bn_act_bwd_reduce_h");
  return ccn_reduce_partials(partial, nparts, 2 * C, sums, stream);
}

int ccn_bn_act_bwd_apply_h(const void* dZ, int dz16, int64_t lddz, const float* Y, int64_t ldy, int64_t rows, int64_t C,
                           const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                           const double* sums, float count, int training, int acc_params, void* dY, int64_t lddy, float* dgamma,
                           float* dbeta, int f16, void* stream) {
  CCN_REQUIRE(dZ && Y && dY && sums && rows > 0 && C > 0 && lddz >= C && ldy >= C && lddy >= C && lddy % 8 == 0 && aligned16(dY),
              "bn_act_bwd_apply_h: bad arguments");
  const EwGeom g = ew_geom(lddy);
  const dim3 grid((unsigned)ccn_blocks(rows, EW_ROWS), g.gy);
  const float inv_n = 1.f / count;
#define CCN_APPLY_H(F16_, DZ16_)                                                                                              \
  hipLaunchKernelGGL((bn_act_bwd_apply_h_kernel<F16_, DZ16_>), grid, dim3(EW_TPB), 0, (hipStream_t)stream,                    \
                     dZ, lddz, (const void*)Y, ldy, rows, C, scale, shift, mean, rstd, act, slope, sums, training, inv_n, (u16*)dY, lddy,  \
                     dgamma, dbeta, acc_params, g.cpb, g.rpp)
  if (f16) { if (dz16) CCN_APPLY_H(true, true); else CCN_APPLY_H(true, false); }
  else { if (dz16) CCN_APPLY_H(false, true); else CCN_APPLY_H(false, false); }
#undef CCN_APPLY_H
  CCN_LAUNCH_OK("bn_act_bwd_apply_h");
  return CCN_OK;
}

// ---- the same two passes for a layer that kept only its OUTPUT z (ccn_gemm_nt_h_bnact): zt 1 = bf16 rows, 2 = fp16 rows, 3 = fp32 rows
static bool hz_args_ok(const void* Z, int zt, int64_t ldz, int64_t C) {
  return Z != nullptr && zt >= 1 && zt <= 3 && ldz >= C;
}

int ccn_bn_act_bwd_reduce_hz(const void* dZ, int dz16, int64_t lddz, const void* Z, int zt, int z_pre, int64_t ldz, int64_t rows, int64_t C,
                             const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                             double* sums, void* stream) {
  CCN_REQUIRE(dZ && sums && rows > 0 && C > 0 && lddz >= C && hz_args_ok(Z, zt, ldz, C), "bn_act_bwd_reduce_hz: bad arguments");
  const int64_t nparts = ccn_stats_rows(rows);
  double* partial = sums + 2 * C;
  const EwGeom g = ew_geom((C + 7) / 8 * 8);
#define CCN_REDUCE_HZ(DZ16_, YT_)                                                                                                \
  hipLaunchKernelGGL((bn_act_bwd_reduce_h_kernel<DZ16_, YT_>), dim3((unsigned)nparts, g.gy), dim3(EW_TPB), 0, (hipStream_t)stream, \
                     dZ, lddz, Z, ldz, rows, C, scale, shift, mean, rstd, act, slope, partial, g.cpb, g.rpp, z_pre)
  if (dz16) { if (zt == 1) CCN_REDUCE_HZ(true, 1); else if (zt == 2) CCN_REDUCE_HZ(true, 2); else CCN_REDUCE_HZ(true, 3); }
  else { if (zt == 1) CCN_REDUCE_HZ(false, 1); else if (zt == 2) CCN_REDUCE_HZ(false, 2); else CCN_REDUCE_HZ(false, 3); }
#undef CCN_REDUCE_HZ
  CCN_LAUNCH_OK("bn_act_bwd_reduce_hz");
  return ccn_reduce_partials(partial, nparts, 2 * C, sums, stream);
}

int ccn_bn_act_bwd_apply_hz(const void* dZ, int dz16, int64_t lddz, const void* Z, int zt, int z_pre, int64_t ldz, int64_t rows, int64_t C,
                            const float* scale, const float* shift, const float* mean, const float* rstd, int act, float slope,
                            const double* sums, float count, int training, int acc_params, void* dY, int64_t lddy, float* dgamma,
                            float* dbeta, void* stream) {
  CCN_REQUIRE(dZ && dY && sums && rows > 0 && C > 0 && lddz >= C && lddy >= C && lddy % 8 == 0 && aligned16(dY) &&
                  hz_args_ok(Z, zt, ldz, C),
              "bn_act_bwd_apply_hz: bad arguments");
  const EwGeom g = ew_geom(lddy);
  const dim3 grid((unsigned)ccn_blocks(rows, EW_ROWS), g.gy);
  const float inv_n = 1.f / count;
#define CCN_APPLY_HZ(DZ16_, YT_)                                                                                              \
  hipLaunchKernelGGL((bn_act_bwd_apply_h_kernel<false, DZ16_, YT_>), grid, dim3(EW_TPB), 0, (hipStream_t)stream,               \
                     dZ, lddz, Z, ldz, rows, C, scale, shift, mean, rstd, act, slope, sums, training, inv_n, (u16*)dY, lddy,   \
                     dgamma, dbeta, acc_params, g.cpb, g.rpp, z_pre)
  if (dz16) { if (zt == 1) CCN_APPLY_HZ(true, 1); else if (zt == 2) CCN_APPLY_HZ(true, 2); else CCN_APPLY_HZ(true, 3); }
  else { if (zt == 1) CCN_APPLY_HZ(false, 1); else if (zt == 2) CCN_APPLY_HZ(false, 2); else CCN_APPLY_HZ(false, 3); }
#undef CCN_APPLY_HZ
  CCN_LAUNCH_OK("bn_act_bwd_apply_hz");
  return CCN_OK;
}

}  // extern "C"

// Weight-gradient product on the LDS-DMA pipeline:  dW[N x K] += dY[M x N]^T  X[M x K]   (fp32 MFMA, exact fma chains)
//
// Replaces autograd's weight gradient of F.linear / F.conv1d (reference fast_conv1d.py:183; PyG MLP as used at
// base.py:90-125).  Both operands are CONTRACTION-MAJOR: the contraction index m is the row of dY and of X, so a 32-row
// slice of either operand is 32 contiguous row pieces.  One workgroup (4 waves) owns one (output tile, row chunk) work item:
//   * tile TN x TK of dW (TN, TK in {64, 128}); the M rows are split into `split` contiguous chunks of whole 32-row slices
//   * a slice of both operands reaches LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging) as a [32][TN] and a
//     [32][TK] fp32 image: one wave instruction copies 64 lanes x 16 B = (256 / TN) rows of TN floats, linear in LDS,
//     the per-lane global address supplies row and column; two-stage ring, slice u+1 in flight while slice u is multiplied
//     (counted waits + raw s_barrier; two workgroups per CU, so one's waits sit under the other's MFMAs)
//   * v_mfma_f32_32x32x2_f32: lane (i, h) supplies A[i][k = h] and B[k = h][i] for one step of TWO contraction rows.  With
//     the images contraction-major, lane i reads 8 bytes at row 2s + h: the values of output rows n = 2i, 2i + 1 (and of
//     output columns k = 2i, 2i + 1) -- i.e. the fragments of TWO MFMA tiles per operand in one conflict-free ds_read_b64.
//     The wave's 64 x 64 quadrant is therefore an interleave of four 32 x 32 tiles (even/odd n x even/odd k); the epilogue
//     undoes the interleave with 8-byte stores.
//   * waves: (TN/64) x (TK/64) quadrants; with fewer than four quadrants the waves split the slice's 32 contraction rows
//     between them (WC = 4 / quadrants) and their accumulators are summed through LDS once per work item
//   * the partial tile of a work item goes to a caller-owned slab (plain 8-byte stores); ccn_tn_reduce adds the slabs of a
//     tile to dW in chunk order: deterministic, and 4 * N * K * split bytes written once instead of one fp32 atomic per
//     accumulator element (r01p: 61 MB of atomics per launch at 1.3 M x 256 x 256).  Without scratch (or split == 1)
//     the accumulators are added to dW directly (atomics when split > 1).
//   * XCD-aware work order: the tiles of one row chunk run on workgroup ids w, w + 8, ... (one XCD, dispatched back to
//     back), so the chunk's rows are fetched from HBM once and re-read from that XCD's L2 by the other tiles.
// Rows beyond M in the last slice: source rows are clamped (in bounds) and the dY fragment is zeroed for them.
// Columns beyond N / K: source columns are clamped into the row; they only feed accumulator elements that are never stored.
#include <atomic>
#include <type_traits>

#include "ccn_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int TN_TPB = 256;
constexpr int TN_SLICE = 32;   // contraction rows per slice

__device__ __forceinline__ void glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// EPI 0: slab store, 1: atomic add into dW, 2: plain add into dW (single owner)
// XF: the X operand is the PRE-normalisation product of the previous layer; x = act(b * scale[k] + shift[k]) (that layer's
// BatchNorm + activation, the expression of ccn_bn_act_fwd) is applied to the fragments between LDS and the MFMAs.  A lane's
// two X columns are fixed for a work item, so scale / shift are four registers; the transform of step t + 1 sits between the
// MFMAs of step t (ccn_gemm_tn_ws_xf; counterpart of ccn_gemm_nt_xf).
template <int TN, int TK, int EPI, bool XF>
__global__ __launch_bounds__(TN_TPB, 2) void gemm_tn_glds_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ B, int64_t ldb,
                                                                 float* __restrict__ C, int64_t ldc, int64_t M, int64_t N,
                                                                 int64_t K, int tiles_k, int tiles, int split,
                                                                 int64_t slices_per_chunk, int64_t n_ids, int xcd_order,
                                                                 float* __restrict__ slabs, int64_t a_extent,
                                                                 int64_t b_extent, const float* __restrict__ xf_scale,
                                                                 const float* __restrict__ xf_shift, float xf_neg) {
  // a_extent / b_extent: floats readable from the start of an A / B row as the kernel sees it -- the leading dimension minus the
  // column offset of this launch's block (tn_ws_impl splits N and K into a 128-multiple part and a remainder block whose
  // operand pointers are shifted by n0 / k0), or K when the B rows overlap (ccn_conv_rows_tn)
  constexpr int QN = TN / 64, QK = TK / 64, WC = 4 / (QN * QK);   // quadrants, waves sharing a quadrant
  constexpr int STEPS = TN_SLICE / 2 / WC;                          // 2-row MFMA steps per wave and slice
  constexpr int AF = TN_SLICE * TN, BF = TN_SLICE * TK, STAGE = AF + BF;
  constexpr int RPI_A = 256 / TN, RPI_B = 256 / TK;                 // rows per DMA instruction
  constexpr int NIA = TN_SLICE / RPI_A / 4, NIB = TN_SLICE / RPI_B / 4;   // DMA instructions per wave, slice, operand
  // one LDS array (a second __shared__ object beside a DMA ring makes hipcc drain the queue before every ds_read)
  constexpr int RED = QN * QK * (WC - 1) * 64 * 64;                 // cross-wave accumulator reduction (reuses the ring)
  constexpr int LDS_FLOATS = 2 * STAGE > RED ? 2 * STAGE : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  // (the wave index as a scalar: LDS-DMA destinations and slice bases stay in SGPRs)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 31, h = lane >> 5;
  const int quad = wave / WC, wc = wave % WC;
  const int wn = quad / QK, wk = quad % QK;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  // this lane's fragment address inside a stage: row (wc * 2 * STEPS + h), columns wn*64 + 2i (A) / wk*64 + 2i (B)
  const uint32_t a_off = (uint32_t)(((wc * 2 * STEPS + h) * TN + wn * 64 + 2 * i) * 4);
  const uint32_t b_off = (uint32_t)((AF + (wc * 2 * STEPS + h) * TK + wk * 64 + 2 * i) * 4);
  const int64_t total_slices = (M + TN_SLICE - 1) / TN_SLICE;

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
    const int64_t n0 = (tile / tiles_k) * TN, k0 = (tile % tiles_k) * TK;

    f32x16 acc[2][2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ta][tb][r] = 0.f;

    float xsc[2] = {1.f, 1.f}, xsh[2] = {0.f, 0.f};
    if (XF) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        int64_t k = k0 + wk * 64 + 2 * i + c;
        k = k < K ? k : K - 1;
        xsc[c] = xf_scale[k];
        xsh[c] = xf_shift[k];
      }
    }
    // (branch-free; bits of act(b * scale + shift), ReLU's zero may be -0.0.  Round 5: packed -- v_pk_fma_f32, v_pk_mul_f32 and
    // two v_max_f32 instead of two each of fma / mul / cmp / cndmask + the VCC hazard's wait states; max(v, v * neg) equals
    // (v > 0 ? v : v * neg) for 0 <= neg <= 1, which the entry point checks.)
    const f32x2 xsc2 = {xsc[0], xsc[1]}, xsh2 = {xsh[0], xsh[1]}, neg2 = {xf_neg, xf_neg};
    auto xform = [&](f32x2& b) {
      const f32x2 v = __builtin_elementwise_fma(b, xsc2, xsh2);
      const f32x2 w = v * neg2;
      b.x = __builtin_fmaxf(v.x, w.x);
      b.y = __builtin_fmaxf(v.y, w.y);
    };
    if (s_beg < s_end) {
      // ---- per-lane DMA source columns (clamped into the row: lda, ldb are multiples of 4 and >= 4)
      int64_t a_col[NIA], b_col[NIB];
      int a_row[NIA], b_row[NIB];
#pragma unroll
      for (int q = 0; q < NIA; ++q) {
        const int g = wave * NIA + q;
        a_row[q] = g * RPI_A + lane / (TN / 4);
        int64_t c = n0 + 4 * (lane % (TN / 4));
        a_col[q] = c <= a_extent - 4 ? c : a_extent - 4;
      }
#pragma unroll
      for (int q = 0; q < NIB; ++q) {
        const int g = wave * NIB + q;
        b_row[q] = g * RPI_B + lane / (TK / 4);
        int64_t c = k0 + 4 * (lane % (TK / 4));
        b_col[q] = c <= b_extent - 4 ? c : b_extent - 4;
      }
      // a copy's source = slice base (scalar) + this lane's 32-bit byte offset inside the slice (row * ld + column): no vector
      // address arithmetic per copy.  Only the last slice of the matrix (rows beyond M) takes the per-lane clamped form.
      uint32_t a_off32[NIA], b_off32[NIB];
#pragma unroll
      for (int q = 0; q < NIA; ++q) a_off32[q] = (uint32_t)((a_row[q] * lda + a_col[q]) * 4);
#pragma unroll
      for (int q = 0; q < NIB; ++q) b_off32[q] = (uint32_t)((b_row[q] * ldb + b_col[q]) * 4);
      auto issue = [&](int64_t s, int stage) {
        float* st = lds + stage * STAGE;
        const int64_t m0 = s * TN_SLICE;
        if (m0 + TN_SLICE <= M) {
          const char* const a_sl = reinterpret_cast<const char*>(A + m0 * lda);
          const char* const b_sl = reinterpret_cast<const char*>(B + m0 * ldb);
#pragma unroll
          for (int q = 0; q < NIA; ++q) glds16(reinterpret_cast<const float*>(a_sl + a_off32[q]), st + (wave * NIA + q) * 256);
#pragma unroll
          for (int q = 0; q < NIB; ++q) glds16(reinterpret_cast<const float*>(b_sl + b_off32[q]), st + AF + (wave * NIB + q) * 256);
          return;
        }
#pragma unroll
        for (int q = 0; q < NIA; ++q) {
          int64_t row = m0 + a_row[q];
          row = row < M ? row : M - 1;
          glds16(A + row * lda + a_col[q], st + (wave * NIA + q) * 256);
        }
#pragma unroll
        for (int q = 0; q < NIB; ++q) {
          int64_t row = m0 + b_row[q];
          row = row < M ? row : M - 1;
          glds16(B + row * ldb + b_col[q], st + AF + (wave * NIB + q) * 256);
        }
      };
      __builtin_amdgcn_s_barrier();          // every wave is done with the previous item's LDS reads
      issue(s_beg, 0);
      int stage = 0;
      for (int64_t s = s_beg; s < s_end; ++s, stage ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slice s has landed (this wave's share)
        __builtin_amdgcn_s_barrier();                      // ... and everybody else's; stage^1 is free again
        if (s + 1 < s_end) issue(s + 1, stage ^ 1);
        const uint32_t sb = lds_base + (uint32_t)(stage * STAGE * 4);
        const int64_t rl64 = M - s * TN_SLICE;              // >= 32 except in the last slice of all
        const int rows_left = rl64 < TN_SLICE ? (int)rl64 : TN_SLICE;
        // The slice's MFMA steps, in two instantiations: only the LAST slice of the matrix can hold rows beyond M, and the test
        // "is this lane's row beyond M" if-converts into a compare and four selects PER STEP -- five vector instructions per four
        // MFMAs on every slice of every launch when written inline (round 2).  Full slices take the form without it.
        auto compute = [&](auto partial_tag) {
          constexpr bool partial = decltype(partial_tag)::value;
          f32x2 fa[2], fb[2];
          // fragment reads are inline asm (hipcc would otherwise wait for every pending LDS-DMA before a visible ds_read);
          // step t+1 is read before the MFMAs of step t and waited for with a counted lgkmcnt
          asm volatile("ds_read_b64 %0, %1" : "=v"(fa[0]) : "v"(sb + a_off) : "memory");
          asm volatile("ds_read_b64 %0, %1" : "=v"(fb[0]) : "v"(sb + b_off) : "memory");
          if (XF) {
            // (the waits name the fragment registers as operands: a plain VALU use of an inline-asm ds_read's result is otherwise
            // free to be scheduled in front of the wait)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0]) : : "memory");
            xform(fb[0]);
#pragma unroll
            for (int t = 0; t < STEPS; ++t) {
              if (t + 1 < STEPS) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[(t + 1) & 1]) : "v"(sb + a_off), "n"((t + 1) * 2 * TN * 4) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[(t + 1) & 1]) : "v"(sb + b_off), "n"((t + 1) * 2 * TK * 4) : "memory");
              }
              __builtin_amdgcn_sched_barrier(0);
              f32x2 a = fa[t & 1];
              const f32x2 b = fb[t & 1];
              if (partial && wc * 2 * STEPS + 2 * t + h >= rows_left) a = f32x2{0.f, 0.f};   // rows beyond M (last slice only)
              acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
              acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
              acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
              acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (t + 1 < STEPS) {     // the next step's transform behind this step's four MFMAs: the wave gets here when the
                // fourth has ISSUED (in-order issue, one MFMA per 64 cycles), i.e. with 64 cycles of matrix work still ahead
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[(t + 1) & 1]), "+v"(fa[(t + 1) & 1]) : : "memory");
                xform(fb[(t + 1) & 1]);
              }
            }
            return;
          }
#pragma unroll
          for (int t = 0; t < STEPS; ++t) {
            if (t + 1 < STEPS) {
              asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[(t + 1) & 1]) : "v"(sb + a_off), "n"((t + 1) * 2 * TN * 4) : "memory");
              asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[(t + 1) & 1]) : "v"(sb + b_off), "n"((t + 1) * 2 * TK * 4) : "memory");
              asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            } else {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x2 a = fa[t & 1];
            const f32x2 b = fb[t & 1];
            if (partial && wc * 2 * STEPS + 2 * t + h >= rows_left) a = f32x2{0.f, 0.f};   // rows beyond M (last slice only)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
          }
        };
        if (rows_left < TN_SLICE) compute(std::true_type{});
        else compute(std::false_type{});
      }
    }

    // ---- waves that shared a quadrant: sum their accumulators into the wc == 0 wave through LDS
    if (WC > 1) {
      __builtin_amdgcn_s_barrier();   // all fragment reads of the last slice are done: the ring is free
      float* red = lds;
      if (wc > 0) {
        float* dst = red + ((quad * (WC - 1) + (wc - 1)) * 64 * 64);
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((ta * 2 + tb) * 16 + r) * 64 + lane] = acc[ta][tb][r];
      }
      __syncthreads();
      if (wc == 0) {
#pragma unroll
        for (int w = 0; w < WC - 1; ++w) {
          const float* src = red + ((quad * (WC - 1) + w) * 64 * 64);
#pragma unroll
          for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[ta][tb][r] += src[((ta * 2 + tb) * 16 + r) * 64 + lane];
        }
      }
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    // tile (ta, tb) holds output rows n = 2 * row + ta and columns k = 2 * col + tb of the quadrant.
    if (wc == 0) {
      if (EPI == 0) {
        float* slab = slabs + ((int64_t)tile * split + chunk) * (TN * TK);
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int nl = wn * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + ta;
            *reinterpret_cast<f32x2*>(slab + nl * TK + wk * 64 + 2 * i) = f32x2{acc[ta][0][r], acc[ta][1][r]};
          }
      } else {
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t n = n0 + wn * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + ta;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
              const int64_t k = k0 + wk * 64 + 2 * i + tb;
              if (n < N && k < K) {
                if (EPI == 1)
                  atomicAdd(&C[n * ldc + k], acc[ta][tb][r]);
                else
                  C[n * ldc + k] += acc[ta][tb][r];
              }
            }
          }
      }
    }
    if (WC > 1) __syncthreads();      // the reduction scratch is the ring: finish reading before the next item's DMA
  }
}

// dW[n][k] += sum over the chunks of the tile's slabs (fixed order: deterministic).  A workgroup owns 64 consecutive
// float4 of slab space (one coalesced KiB per slab and wave); its 16 waves take the chunks c = w, w + 16, ... with four
// loads in flight each, and wave 0 adds the 16 partial sums in wave order.  (First form, r02a: one thread per float4
// walking all `split` slabs serially -- 128..512 dependent 64 KB-strided loads: 145-350 us per launch, 7.5 % of GPU time.)
constexpr int RED_WAVES = 16;

template <int TN, int TK>
__global__ __launch_bounds__(64 * RED_WAVES) void tn_reduce_kernel(const float* __restrict__ slabs, int split, int tiles_k,
                                                                   int tiles, int64_t N, int64_t K, float* __restrict__ C,
                                                                   int64_t ldc) {
  constexpr int Q = TN * TK / 4;                      // float4 per slab
  __shared__ float4 red[RED_WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;  // float4 index in (tile, row, column) slab space
  const bool live = e < (int64_t)tiles * Q;
  const int64_t tile = live ? e / Q : 0;
  const int within = live ? (int)(e - tile * Q) : 0;
  const float* src = slabs + tile * split * (int64_t)(TN * TK) + (int64_t)within * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    for (int c = w; c < split; c += 4 * RED_WAVES) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = c + u * RED_WAVES;
        v[u] = cc < split ? *reinterpret_cast<const float4*>(src + (int64_t)cc * (TN * TK)) : make_float4(0.f, 0.f, 0.f, 0.f);
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
  for (int u = 1; u < RED_WAVES; ++u) {
    const float4 t = red[u][lane];
    s.x += t.x;
    s.y += t.y;
    s.z += t.z;
    s.w += t.w;
  }
  const int64_t n = (tile / tiles_k) * TN + (within * 4) / TK;
  const int64_t k = (tile % tiles_k) * TK + (within * 4) % TK;
  if (n >= N || k >= K) return;
  float* dst = C + n * ldc + k;
  dst[0] += s.x;
  if (k + 1 < K) dst[1] += s.y;
  if (k + 2 < K) dst[2] += s.z;
  if (k + 3 < K) dst[3] += s.w;
}

struct TnPlan {
  int tn, tk, tiles_n, tiles_k, tiles, split;
  int64_t slices_per_chunk, n_ids, slab_floats;
  bool xcd;
};

// Tile choice (A/B hook ccn_gemm_tn_use_dma(2 + mode)).  mode 2 (default): an output dimension d > 128 whose last 128-wide
// tile would be at most half used is SPLIT into d - d % 128 columns on 128-wide tiles and a remainder launch on 64-wide tiles
// (192 = 128 + 64, 259 = 256 + 3): the 128 x 128 tile runs at ~116 TFLOP/s, anything with a 64-wide side at 60-98, and a
// remainder launch only costs another pass over the operands (measured, M = 1.34 M: 256 x 192 67 -> 99 TFLOP/s, 192 x 128
// 69 -> 90; 208 k x 256 x 259: 58 -> 94).  mode 0: 64-wide tiles over the whole dimension in that case (the r2a rule);
// mode 1: 128-wide tiles whenever d > 64.
static std::atomic<int> g_tn_pick{2};

inline int pick_tile(int64_t d) {
  if (d <= 64) return 64;
  if (g_tn_pick == 1) return 128;
  const int64_t rem = d % 128;
  return (rem == 0 || rem > 64) ? 128 : 64;
}

// [0, main) on the plan's tiles, [main, d) as a second launch (0: none)
inline int64_t tn_main_part(int64_t d) {
  return (g_tn_pick == 2 && d > 128 && d % 128 != 0 && d % 128 <= 64) ? d / 128 * 128 : d;
}

inline TnPlan tn_plan(int64_t M, int64_t N, int64_t K) {
  TnPlan p;
  p.tn = pick_tile(N);
  p.tk = pick_tile(K);
  p.tiles_n = (int)((N + p.tn - 1) / p.tn);
  p.tiles_k = (int)((K + p.tk - 1) / p.tk);
  p.tiles = p.tiles_n * p.tiles_k;
  const int64_t slices = (M + TN_SLICE - 1) / TN_SLICE;
  // ~512 work items of at least 4 slices each (two workgroups per CU); more tiles than that: one chunk per tile
  int64_t split = (512 + p.tiles - 1) / p.tiles;
  if (split > slices / 4) split = slices / 4;
  if (split < 1) split = 1;
  p.slices_per_chunk = (slices + split - 1) / split;
  split = (slices + p.slices_per_chunk - 1) / p.slices_per_chunk;
  p.split = (int)split;
  p.xcd = split >= 8 && p.tiles > 1;
  p.n_ids = p.xcd ? (int64_t)p.tiles * ((split + 7) / 8 * 8) : (int64_t)p.tiles * split;
  p.slab_floats = split > 1 ? (int64_t)p.tiles * split * p.tn * p.tk : 0;
  return p;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

struct TnXf {
  const float* scale;
  const float* shift;
  float neg;      // multiplier of non-positive values: 0 (ReLU), slope (LeakyReLU), 1 (none)
};

// Background mode (round 6 experiment, ccn_gemm_tn_background): a weight-gradient product has no consumer before the optimiser, so
// it can run on a side stream BESIDE the backward pass.  Two of these workgroups fill a CU's register file (216 + 216 VGPRs of the
// 512 per SIMD lane) and, next to the paired NT kernel, leave no room for anybody: a streaming kernel of the other stream then
// crawls at one wave per SIMD.  With `g_tn_bg_lds` > 0 every workgroup claims that many bytes of LDS in all (static + unused
// dynamic): 84 KiB -> ONE of them per CU (2 x 84 > 160 KiB) beside ONE workgroup of the paired NT kernel (84 + 66 / 74 KiB fit)
// or 4 waves per SIMD of a streaming kernel.  Same arithmetic, same results.
static std::atomic<int> g_tn_bg_lds{0};

template <int TN, int TK>
constexpr int tn_static_lds_bytes() {
  constexpr int QN = TN / 64, QK = TK / 64, WC = 4 / (QN * QK);
  constexpr int STAGE = TN_SLICE * TN + TN_SLICE * TK, RED = QN * QK * (WC - 1) * 64 * 64;
  return 4 * (2 * STAGE > RED ? 2 * STAGE : RED);
}

template <int TN, int TK, int EPI, bool XF>
static size_t tn_bg_dyn_lds() {
  const int want = g_tn_bg_lds.load(std::memory_order_relaxed);
  if (want <= tn_static_lds_bytes<TN, TK>()) return 0;
  const int dyn = want - tn_static_lds_bytes<TN, TK>();
  static std::atomic<int> raised_on[CCN_MAX_DEVICES];   // (per instantiation and device: the dynamic bytes allowed so far)
  std::atomic<int>* const raised = ccn_device_slot(raised_on);
  if (raised == nullptr) return 0;
  if (raised->load(std::memory_order_relaxed) < dyn) {
    if (hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<TN, TK, EPI, XF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            dyn) != hipSuccess) {
      (void)hipGetLastError();                  // (not this launch's error)
      return 0;
    }
    raised->store(dyn, std::memory_order_relaxed);
  }
  return (size_t)dyn;
}

template <int TN, int TK>
int launch_tn(const TnPlan& p, const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw,
              int64_t M, int64_t N, int64_t K, float* slabs, hipStream_t s, int64_t a_extent, int64_t b_extent,
              const TnXf* xf) {
  const int64_t grid = p.n_ids < 512 ? p.n_ids : 512;
  const float* xs = xf ? xf->scale : nullptr;
  const float* xh = xf ? xf->shift : nullptr;
  const float xn = xf ? xf->neg : 1.f;
#define CCN_TN_LAUNCH(EPI_, XF_, SLABS_)                                                                                       \
  hipLaunchKernelGGL((gemm_tn_glds_kernel<TN, TK, EPI_, XF_>), dim3((unsigned)grid), dim3(TN_TPB),                               \
                     (tn_bg_dyn_lds<TN, TK, EPI_, XF_>()), s, dY, lddy, X, ldx, dW,                                              \
                     lddw, M, N, K, p.tiles_k, p.tiles, p.split, p.slices_per_chunk, p.n_ids, p.xcd ? 1 : 0, SLABS_, a_extent, \
                     b_extent, xs, xh, xn)
  if (p.split == 1) {
    if (xf) CCN_TN_LAUNCH(2, true, (float*)nullptr);
    else CCN_TN_LAUNCH(2, false, (float*)nullptr);
  } else if (slabs == nullptr) {
    if (xf) CCN_TN_LAUNCH(1, true, (float*)nullptr);
    else CCN_TN_LAUNCH(1, false, (float*)nullptr);
  } else {
    if (xf) CCN_TN_LAUNCH(0, true, slabs);
    else CCN_TN_LAUNCH(0, false, slabs);
    const int64_t work = (int64_t)p.tiles * (TN * TK / 4);
    hipLaunchKernelGGL((tn_reduce_kernel<TN, TK>), dim3((unsigned)((work + 63) / 64)), dim3(64 * RED_WAVES), 0, s, slabs,
                       p.split, p.tiles_k, p.tiles, N, K, dW, lddw);
  }
#undef CCN_TN_LAUNCH
  return CCN_OK;
}

static std::atomic<bool> g_tn_dma{true};   // A/B hook (ccn_gemm_tn_use_dma)

}  // namespace

extern "C" {

int ccn_gemm_tn_background(int lds_bytes) {   // 0 = off; e.g. 86016: one weight-gradient workgroup per CU (see g_tn_bg_lds)
  g_tn_bg_lds.store(lds_bytes < 0 ? 0 : (lds_bytes > 160 * 1024 ? 160 * 1024 : lds_bytes), std::memory_order_relaxed);
  return CCN_OK;
}

int ccn_gemm_tn_use_dma(int on) {   // 0 / 1: the register-staged / LDS-DMA kernels; 2 + mode: tile choice A/B (pick_tile)
  if (on >= 2) g_tn_pick = on - 2;
  else g_tn_dma = on != 0;
  return CCN_OK;
}

// The LDS-DMA kernel takes 16-byte aligned operands whose leading dimensions are multiples of 4, from N >= 32 and K >= 32
// on (narrower outputs leave most of a 64-wide tile empty: the register-staged kernels keep those) and M >= 1024.
static bool tn_dma_ok(const float* dY, int64_t lddy, const float* X, int64_t ldx, int64_t M, int64_t N, int64_t K) {
  return g_tn_dma && aligned16(dY) && aligned16(X) && lddy % 4 == 0 && ldx % 4 == 0 && lddy >= 4 && ldx >= 4 && N >= 32 &&
         K >= 32 && M >= 1024;
}

size_t ccn_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N < 32 || K < 32 || M < 1024 || !g_tn_dma) return 0;
  const int64_t nm = tn_main_part(N), km = tn_main_part(K);
  int64_t most = 0;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int64_t n = a ? N - nm : nm, k = b ? K - km : km;
      if (n <= 0 || k <= 0) continue;
      const int64_t f = tn_plan(M, n, k).slab_floats;
      most = f > most ? f : most;
    }
  return (size_t)most * sizeof(float);
}

int ccn_gemm_tn_generic(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                        int64_t N, int64_t K, int overlap, void* stream);

// one launch (+ its slab reduction) of the LDS-DMA kernel over an N x K block of the output
static int tn_launch_block(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                           int64_t N, int64_t K, void* workspace, size_t workspace_bytes, hipStream_t s, int64_t aext,
                           int64_t ext, const TnXf* xf) {
  const TnPlan p = tn_plan(M, N, K);
  float* slabs = nullptr;
  if (p.slab_floats > 0 && workspace != nullptr) {
    CCN_REQUIRE(workspace_bytes >= (size_t)p.slab_floats * sizeof(float) && aligned16(workspace),
                "gemm_tn_ws: workspace too small (%zu < %zu bytes) or unaligned", workspace_bytes,
                (size_t)p.slab_floats * sizeof(float));
    slabs = (float*)workspace;
  }
  if (p.tn == 128 && p.tk == 128) return launch_tn<128, 128>(p, dY, lddy, X, ldx, dW, lddw, M, N, K, slabs, s, aext, ext, xf);
  if (p.tn == 128) return launch_tn<128, 64>(p, dY, lddy, X, ldx, dW, lddw, M, N, K, slabs, s, aext, ext, xf);
  if (p.tk == 128) return launch_tn<64, 128>(p, dY, lddy, X, ldx, dW, lddw, M, N, K, slabs, s, aext, ext, xf);
  return launch_tn<64, 64>(p, dY, lddy, X, ldx, dW, lddw, M, N, K, slabs, s, aext, ext, xf);
}

static int tn_ws_impl(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M,
                      int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream, bool overlap,
                      const TnXf* xf = nullptr) {
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) return CCN_OK;
  if (!tn_dma_ok(dY, lddy, X, ldx, M, N, K)) {
    CCN_REQUIRE(xf == nullptr, "gemm_tn_ws_xf: shape / alignment outside the LDS-DMA kernel (ask ccn_gemm_tn_xf_ok first)");
    return ccn_gemm_tn_generic(dY, lddy, X, ldx, dW, lddw, M, N, K, overlap ? 1 : 0, stream);
  }
  const int64_t ext = overlap ? K : ldx;
  // up to 2 x 2 blocks, launched back to back on the stream (they share the slab scratch in stream order): multiples of
  // 128 columns keep every block's operand pointers 16-byte aligned
  const int64_t nm = tn_main_part(N), km = tn_main_part(K);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int64_t n0 = a ? nm : 0, n = a ? N - nm : nm, k0 = b ? km : 0, k = b ? K - km : km;
      if (n <= 0 || k <= 0) continue;
      TnXf part;
      if (xf) part = TnXf{xf->scale + k0, xf->shift + k0, xf->neg};
      const int rc = tn_launch_block(dY + n0, lddy, X + k0, ldx, dW + n0 * lddw + k0, lddw, M, n, k, workspace,
                                     workspace_bytes, s, lddy - n0, ext - k0, xf ? &part : nullptr);
      if (rc) return rc;
    }
  CCN_LAUNCH_OK("gemm_tn_ws");
  return CCN_OK;
}

int ccn_gemm_tn_ws(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M, int64_t N,
                   int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(dY && X && dW, "gemm_tn_ws: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K, "gemm_tn_ws: bad sizes");
  return tn_ws_impl(dY, lddy, X, ldx, dW, lddw, M, N, K, workspace, workspace_bytes, stream, false);
}

int ccn_gemm_tn_xf_ok(const float* dY, int64_t lddy, const float* X, int64_t ldx, int64_t M, int64_t N, int64_t K) {
  return tn_dma_ok(dY, lddy, X, ldx, M, N, K) ? 1 : 0;
}

int ccn_gemm_tn_ws_xf(const float* dY, int64_t lddy, const float* X, int64_t ldx, const float* x_scale, const float* x_shift,
                      int x_act, float x_slope, float* dW, int64_t lddw, int64_t M, int64_t N, int64_t K, void* workspace,
                      size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(dY && X && dW && x_scale && x_shift, "gemm_tn_ws_xf: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K, "gemm_tn_ws_xf: bad sizes");
  CCN_REQUIRE(x_act != CCN_ACT_LEAKY || (x_slope >= 0.f && x_slope <= 1.f), "gemm_tn_ws_xf: LeakyReLU slope outside [0, 1]");
  const TnXf xf{x_scale, x_shift, x_act == CCN_ACT_RELU ? 0.f : (x_act == CCN_ACT_LEAKY ? x_slope : 1.f)};
  return tn_ws_impl(dY, lddy, X, ldx, dW, lddw, M, N, K, workspace, workspace_bytes, stream, false, &xf);
}

int ccn_conv_rows_tn(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw, int64_t M, int64_t N,
                     int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(dY && X && dW, "conv_rows_tn: null pointer");
  CCN_REQUIRE(M >= 0 && N > 0 && K > 0 && lddy >= N && ldx > 0 && K % ldx == 0 && lddw >= K,
              "conv_rows_tn: bad sizes (K must be taps * ldx)");
  return tn_ws_impl(dY, lddy, X, ldx, dW, lddw, M, N, K, workspace, workspace_bytes, stream, K > ldx);
}

}  // extern "C"

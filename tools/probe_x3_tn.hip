// Upper bound of a split-operand ("bf16x3") WEIGHT-GRADIENT kernel: dW += dY^T X with BOTH operands fp32 activations, i.e. no
// pre-split weight image -- every fragment value of both operands is split into three bf16 terms in registers on its way to
// the matrix cores (VERDICT r4 #3c: "write the split-operand TN kernel or show why it cannot beat 130 TFLOP/s").
//
// The probe keeps only what such a kernel cannot avoid per 16-deep contraction step of a wave's 64 x 64 quadrant:
//   16 ds_read_b64 (two fragments of two 32-wide tiles per operand, contraction-major LDS images as gemm_tn_glds_kernel's),
//   V single-issue vector instructions of the three-way split (the forward kernel's measured form: 5.5 per value;
//     32 values per lane and step -> V = 176; the forward kernel splits ONE operand of half the size: V = 44),
//   24 v_mfma_f32_32x32x16_bf16 (six partial products for each of the four 32 x 32 tiles),
// software-pipelined like gemm_x3_lean_kernel (the vector work of step s + 1 between the MFMAs of step s), two 4-wave
// workgroups per CU, no global traffic, no barriers.  It prints the TFLOP/s-equivalent (2 * 64 * 64 * 16 flop per wave
// and step) for V = 0, 44, 88, 176: what the real kernel would reach if its copies and barriers were free.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fsub(float a, float b) {
  float r;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// split one PAIR of values (two consecutive contraction elements) into its three packed bf16 dwords: 11 vector instructions
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(x0, x1);
  const float f0 = __uint_as_float(h << 16), f1 = __uint_as_float(h & 0xffff0000u);
  const float q0 = fsub(x0, f0), q1 = fsub(x1, f1);
  m = cvt_pk(q0, q1);
  const float g0 = __uint_as_float(m << 16), g1 = __uint_as_float(m & 0xffff0000u);
  l = cvt_pk(fsub(q0, g0), fsub(q1, g1));
}

template <int PAIRS>   // pairs of values split per step and lane (16 = both operands of a 64 x 64 quadrant: 32 values)
__global__ __launch_bounds__(256, 2) void probe(float* out, int steps) {
  __shared__ float lds[2 * 32 * 128];      // two contraction-major [32][128] fp32 images (dY slice, X slice)
  for (int i = threadIdx.x; i < 2 * 32 * 128; i += 256) lds[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  struct Frag { u32x4 f[4][3]; };   // [fragment: A even, A odd, B even, B odd][image h, m, l]: 4 dwords = 8 bf16 each
  Frag F0, F1;
  for (int f = 0; f < 4; ++f)
    for (int s = 0; s < 3; ++s) F0.f[f][s] = F1.f[f][s] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};
  auto step = [&](Frag& cur, Frag& nxt) {     // (static register indices only: the two parities are two calls)
    f32x2 v[16];
    // 16 ds_read_b64: rows 8h + j of a 16-row step, columns 2i, 2i + 1 of each operand
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(base + (unsigned)(((8 * h + j) * 128 + 2 * i) * 4)) : "memory");
      asm volatile("ds_read_b64 %0, %1" : "=v"(v[8 + j]) : "v"(base + (unsigned)((32 * 128 + (8 * h + j) * 128 + 2 * i) * 4)) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mI = 0; mI < 24; ++mI) {
      const int pp = mI >> 2, t = mI & 3;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.f[t >> 1][SA[pp]]),
                                                       __builtin_bit_cast(bf16x8, cur.f[2 + (t & 1)][SB[pp]]), acc[t], 0, 0, 0);
      // the split of the next step's fragments, spread over the 24 gaps: PAIRS pairs in all (pair k in gap k * 24 / PAIRS)
#pragma unroll
      for (int k = 0; k < PAIRS; ++k)
        if (k * 24 / (PAIRS > 0 ? PAIRS : 1) == mI) {
          const int f = (k >> 2) & 3, d = k & 3, j = (2 * d) & 7;
          unsigned hh, mm, ll;
          split_pair(v[(f >> 1) * 8 + j][f & 1], v[(f >> 1) * 8 + j + 1][f & 1], hh, mm, ll);
          nxt.f[f][0][d] = hh;
          nxt.f[f][1][d] = mm;
          nxt.f[f][2][d] = ll;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int st = 0; st < steps; st += 2) {
    step(F0, F1);
    step(F1, F0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

template <int PAIRS>
static void run(const char* what, int vec) {
  float* out;
  hipMalloc(&out, 64);
  const int steps = 4096, grid = 512;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<PAIRS>, dim3(grid), dim3(256), 0, 0, out, 64);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<PAIRS>, dim3(grid), dim3(256), 0, 0, out, steps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flops = 2.0 * 64 * 64 * 16 * (double)steps * 4 * grid;
  printf("%-52s %3d vector instructions / step: %7.3f ms  %6.1f TFLOP/s-equivalent (matrix ceiling 417 at 2.4 GHz)\n", what, vec,
         ms, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<0>("24 MFMAs + 16 LDS reads, no split", 0);
  run<4>("+ the forward kernel's split (one operand, 8 values)", 44);
  run<8>("+ 16 values", 88);
  run<16>("+ 32 values: BOTH operands of a TN product", 176);
  return 0;
}

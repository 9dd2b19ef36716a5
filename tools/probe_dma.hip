// Probe of global_load_lds_dwordx4 placement: 64 lanes copy 16 B each from per-lane global addresses; where does lane L land?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void probe(const uint16_t* src, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2048];
  for (int e = threadIdx.x; e < 2048; e += 64) lds[e] = 0xffff;
  __syncthreads();
  const int lane = threadIdx.x;
  // lane -> row lane>>4 of a 4 x 128 matrix, chunk (lane & 15): source element offset
  const uint16_t* g = src + (lane >> 4) * 128 + 8 * (lane & 15);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(lds + 512), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = threadIdx.x; e < 2048; e += 64) out[e] = lds[e];
}
int main() {
  uint16_t h_src[512], h_out[2048];
  for (int i = 0; i < 512; ++i) h_src[i] = (uint16_t)i;
  uint16_t *d_src, *d_out;
  (void)hipMalloc(&d_src, sizeof(h_src));
  (void)hipMalloc(&d_out, sizeof(h_out));
  (void)hipMemcpy(d_src, h_src, sizeof(h_src), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_src, d_out);
  (void)hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int e = 0; e < 2048; ++e) {
    const int want = (e >= 512 && e < 1024) ? e - 512 : 0xffff;
    if (h_out[e] != want && bad < 20) { printf("lds[%d] = %d, expected %d\n", e, h_out[e], want); ++bad; }
  }
  printf("%s\n", bad ? "MISMATCH" : "LDS-DMA: lane L lands at base + 16 L bytes (linear)");
  return 0;
}

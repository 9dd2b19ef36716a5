// Probe of ds_read_b64_tr_b16: LDS holds u16 element e at byte 2e (value = e); every lane supplies an address; prints
// which 4 elements each lane receives.  Usage: probe_tr   (address pattern: lane L of a 16-lane group: row q = L>>2,
// 8-byte piece p = L&3 of a 4 x 16 block with 256-byte rows; group g uses columns 16 g .. 16 g + 15)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint32_t* out, const uint32_t* addr_in) {
  __shared__ uint16_t lds[4096];
  for (int e = threadIdx.x; e < 4096; e += 64) lds[e] = (uint16_t)e;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr_in[threadIdx.x]) : "memory");
  out[threadIdx.x * 2] = v.x;
  out[threadIdx.x * 2 + 1] = v.y;
}
int main(int argc, char**) {
  uint32_t h_addr[64], h_out[128];
  for (int lane = 0; lane < 64; ++lane) {
    const int g = lane >> 4, L = lane & 15, q = L >> 2, p = L & 3;
    h_addr[lane] = 256 * q + 32 * g + 8 * p;     // row q (128 elements per row), columns 16 g + 4 p .. + 3
    if (argc > 1) {   // the swizzled addresses of gemm_h_tn_kernel (wn = 0, t = 0, half = 0)
      const int h = lane >> 5, gsel = (lane >> 4) & 1, R = 8 * h + q, f = ((R & 3) << 2) | ((R >> 2) & 3);
      const int cha = 2 * gsel + (p >> 1);
      h_addr[lane] = 256 * R + 16 * (cha ^ f) + 8 * (p & 1);
    }
  }
  uint32_t *d_addr, *d_out;
  hipMalloc(&d_addr, sizeof(h_addr));
  hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out, d_addr);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; ++lane) {
    const uint32_t a = h_out[2 * lane], b = h_out[2 * lane + 1];
    const int e[4] = {(int)(a & 0xffff), (int)(a >> 16), (int)(b & 0xffff), (int)(b >> 16)};
    printf("lane %2d (addr %4u = row %u col %3u): ", lane, h_addr[lane], h_addr[lane] / 256, (h_addr[lane] % 256) / 2);
    for (int j = 0; j < 4; ++j) printf(" [r%d c%3d]", e[j] / 128, e[j] % 128);
    if (argc > 1) {   // which lane's address + element explains each value
      printf("   <-");
      for (int j = 0; j < 4; ++j) {
        int src = -1, el = -1;
        for (int l2 = 0; l2 < 64; ++l2) for (int k = 0; k < 4; ++k) if ((int)h_addr[l2] / 2 + k == e[j]) { src = l2; el = k; }
        printf(" (lane %d el %d)", src, el);
      }
    }
    printf("\n");
  }
  return 0;
}

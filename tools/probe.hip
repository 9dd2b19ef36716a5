// GPU probe: are the float primitives used by the index kernels correctly rounded on gfx950?
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const float* x, const float* y, float* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o[i] = fmodf(x[i], y[i]);
  o[n + i] = __fdiv_rn(x[i], y[i]);
  o[2 * n + i] = sqrtf(x[i]);
  o[3 * n + i] = (float)sqrt((double)x[i]);
  o[4 * n + i] = rintf(x[i] * 1e-5f);
}
int main() {
  const int n = 1 << 20;
  float *hx = (float*)malloc(n * 4), *hy = (float*)malloc(n * 4), *ho = (float*)malloc(5 * n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    hx[i] = (float)(rand() % 6000000) * (0.3f + (rand() % 1000) * 0.0007f);
    hy[i] = (i & 1) ? 0.03f : 0.007f;
    if (i % 7 == 0) hy[i] = 0.001f + (rand() % 1000) * 1e-4f;
  }
  hx[0] = 837242.6875f; hy[0] = 0.03f;
  float *dx, *dy, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, 5 * n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, hy, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
  hipMemcpy(ho, dout, 5 * n * 4, hipMemcpyDeviceToHost);
  int bad[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    float r0 = fmodf(hx[i], hy[i]), r1 = hx[i] / hy[i], r2 = sqrtf(hx[i]), r3 = sqrtf(hx[i]), r4 = rintf(hx[i] * 1e-5f);
    if (r0 != ho[i]) { if (bad[0] < 5) printf("fmod x=%a y=%a cpu=%a gpu=%a\n", hx[i], hy[i], r0, ho[i]); bad[0]++; }
    if (r1 != ho[n + i]) bad[1]++;
    if (r2 != ho[2 * n + i]) bad[2]++;
    if (r3 != ho[3 * n + i]) bad[3]++;
    if (r4 != ho[4 * n + i]) bad[4]++;
  }
  printf("mismatches of %d: fmod %d div %d sqrt %d rcp %d rint %d ; fmod(837242.6875,0.03)=%.9g cpu %.9g\n", n, bad[0], bad[1],
         bad[2], bad[3], bad[4], ho[0], fmodf(hx[0], hy[0]));
  return 0;
}

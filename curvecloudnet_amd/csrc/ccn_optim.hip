// Harness-side optimiser step (ref src/main.py:56 builds torch.optim.Adam over all
// parameters).  One launch per flat gradient bucket instead of a few hundred small per-tensor updates: the
// parameters, gradients and both moments of a bucket are contiguous (parallel.FlatAdam), so the update is
// a single HBM-bound stream over 4 arrays (16 B read + 12 B written per parameter).
#include "ccn_common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float beta1,
                                                   float beta2, float step_size, float inv_bc2_sqrt, float eps,
                                                   float weight_decay) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      float4 pp = *(const float4*)(p + i), gg = *(const float4*)(g + i);
      float4 mm = *(const float4*)(m + i), vv = *(const float4*)(v + i);
      float* pa = (float*)&pp; float* ga = (float*)&gg; float* ma = (float*)&mm; float* va = (float*)&vv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float grad = ga[j] + weight_decay * pa[j];
        ma[j] = ma[j] + (grad - ma[j]) * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
        va[j] = va[j] * beta2 + (1.0f - beta2) * grad * grad;            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        const float denom = sqrtf(va[j]) * inv_bc2_sqrt + eps;
        pa[j] = pa[j] - step_size * (ma[j] / denom);
      }
      *(float4*)(p + i) = pp; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv;
    } else {
      for (int64_t k = i; k < n; ++k) {
        const float grad = g[k] + weight_decay * p[k];
        const float mk = m[k] + (grad - m[k]) * (1.0f - beta1);
        const float vk = v[k] * beta2 + (1.0f - beta2) * grad * grad;
        m[k] = mk; v[k] = vk;
        p[k] = p[k] - step_size * (mk / (sqrtf(vk) * inv_bc2_sqrt + eps));
      }
    }
  }
}

}  // namespace

extern "C" int ccn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
  CCN_REQUIRE(n >= 0 && step >= 1, "adam_step: n=%lld step=%lld", (long long)n, (long long)step);
  CCN_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
              "adam_step: buffers must be 16-byte aligned");
  if (n == 0) return CCN_OK;
  // bias corrections in double on the host, exactly as torch.optim.Adam computes them from the Python step count
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     n, beta1, beta2, step_size, inv_bc2_sqrt, eps, weight_decay);
  CCN_LAUNCH_OK("adam_step");
  return CCN_OK;
}

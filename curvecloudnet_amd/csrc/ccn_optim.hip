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

// ---- mean negative log-likelihood of log_softmax(logits) (harness row H: F.log_softmax + F.nll_loss, ref
// src/run/kitti_seg.py:184-192 and the other runners): one pass over the logits forward (row max, log-sum-exp, per-row
// loss; workgroup partial sums in double), one backward (softmax - onehot, scaled).  torch's two nll_loss kernels alone
// took 0.8 ms per step on 400 k x 20 logits.
constexpr int NLL_TPB = 256;

// STAGED: the workgroup's 256 rows go through LDS first (coalesced loads, row pitch C + 1 floats) and every thread then walks
// ITS row there -- same per-row arithmetic in the same order; a thread walking its row in global memory touches 64 different
// lines per wave instruction: 4.1 GB read for 86 MB of logits at C = 55 (A2D2), 0.57 ms instead of 0.1
template <bool STAGED>
__global__ __launch_bounds__(NLL_TPB) void nll_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const int64_t* __restrict__ target, int64_t rows, int C,
                                                          int64_t ignore, float* __restrict__ lse_out,
                                                          float* __restrict__ per_point, double* __restrict__ partial) {
  extern __shared__ float nll_tile[];
  __shared__ double red[2][NLL_TPB / 64];
  const int64_t i = (int64_t)blockIdx.x * NLL_TPB + threadIdx.x;
  if (STAGED) {
    const int64_t i0 = (int64_t)blockIdx.x * NLL_TPB;
    const int64_t live = rows - i0 < NLL_TPB ? rows - i0 : NLL_TPB;
    for (int idx = threadIdx.x; idx < (int)live * C; idx += NLL_TPB) {
      const int r = idx / C, c = idx - r * C;
      nll_tile[r * (C + 1) + c] = x[(i0 + r) * ldx + c];
    }
    __syncthreads();
  }
  double loss = 0.0, cnt = 0.0;
  if (i < rows) {
    const float* row = STAGED ? nll_tile + threadIdx.x * (C + 1) : x + i * ldx;
    float m = row[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(row[c] - m);
    const float lse = m + logf(sum);
    lse_out[i] = lse;
    const int64_t t = target[i];
    float li = 0.f;
    if (t != ignore && t >= 0 && t < C) {
      li = lse - row[t];
      loss = (double)li;
      cnt = 1.0;
    }
    if (per_point) per_point[i] = li;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    loss += __shfl_xor(loss, d, 64);
    cnt += __shfl_xor(cnt, d, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][w] = loss;
    red[1][w] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < NLL_TPB / 64; ++k) {
      a += red[0][k];
      b += red[1][k];
    }
    partial[(int64_t)blockIdx.x * 2] = a;
    partial[(int64_t)blockIdx.x * 2 + 1] = b;
  }
}

// totals[0] = sum of the per-row losses, totals[1] = number of counted rows; loss = totals[0] / totals[1]
__global__ __launch_bounds__(256) void nll_final_kernel(const double* __restrict__ partial, int64_t nblocks,
                                                        double* __restrict__ totals, float* __restrict__ loss) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int64_t k = threadIdx.x; k < nblocks; k += 256) {   // fixed assignment and order: deterministic
    a += partial[2 * k];
    b += partial[2 * k + 1];
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (threadIdx.x < d) {
      red[0][threadIdx.x] += red[0][threadIdx.x + d];
      red[1][threadIdx.x] += red[1][threadIdx.x + d];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    totals[0] = red[0][0];
    totals[1] = red[1][0];
    *loss = (float)(red[0][0] / red[1][0]);        // 0 / 0 = nan for an all-ignored batch, as torch
  }
}

__global__ __launch_bounds__(NLL_TPB) void nll_bwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const int64_t* __restrict__ target,
                                                          const float* __restrict__ lse, int64_t rows, int C,
                                                          int64_t ignore, const float* __restrict__ gout,
                                                          const double* __restrict__ totals, float* __restrict__ dx,
                                                          int64_t lddx) {
  const int64_t e = (int64_t)blockIdx.x * NLL_TPB + threadIdx.x;   // one logit per thread: coalesced
  if (e >= rows * C) return;
  const int64_t i = e / C;
  const int c = (int)(e - i * C);
  const int64_t t = target[i];
  float g = 0.f;
  if (t != ignore && t >= 0 && t < C) {
    const float scale = gout[0] / (float)totals[1];
    g = (expf(x[i * ldx + c] - lse[i]) - (c == t ? 1.f : 0.f)) * scale;
  }
  dx[i * lddx + c] = g;
}

}  // namespace

extern "C" int64_t ccn_nll_loss_blocks(int64_t rows) { return (rows + NLL_TPB - 1) / NLL_TPB; }

extern "C" int ccn_nll_loss_fwd(const float* logits, int64_t ld, const int64_t* target, int64_t rows, int64_t C,
                                int64_t ignore_index, float* lse, float* per_point, double* scratch, float* loss,
                                void* stream) {
  // scratch: double[2 * ccn_nll_loss_blocks(rows) + 2]; its LAST two doubles receive (sum of losses, counted rows)
  CCN_REQUIRE(logits && target && lse && scratch && loss && rows > 0 && C > 0 && C < (1 << 20) && ld >= C,
              "nll_loss_fwd: bad arguments");
  const int64_t nb = ccn_nll_loss_blocks(rows);
  if (C <= 63) {     // 256 rows x (C + 1) floats of dynamic LDS: 64 KB at C = 63, plus the static reduction scratch
    const size_t dyn = (size_t)NLL_TPB * (C + 1) * sizeof(float);
    if (dyn > 48 * 1024) {   // above the default dynamic-LDS limit: opt in explicitly (as ccn_fps does), not by runtime leniency
      static std::atomic<int> raised_on[CCN_MAX_DEVICES];
      std::atomic<int>* const raised = ccn_device_slot(raised_on);
      CCN_REQUIRE(raised != nullptr, "nll_loss_fwd: cannot name the current device");
      if (!raised->load(std::memory_order_relaxed)) {
        CCN_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(nll_fwd_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, NLL_TPB * 64 * (int)sizeof(float)) == hipSuccess,
                    "nll_loss_fwd: cannot raise the dynamic LDS limit");
        raised->store(1, std::memory_order_relaxed);
      }
    }
    hipLaunchKernelGGL(nll_fwd_kernel<true>, dim3((unsigned)nb), dim3(NLL_TPB), (size_t)NLL_TPB * (C + 1) * sizeof(float),
                       (hipStream_t)stream, logits, ld, target, rows, (int)C, ignore_index, lse, per_point, scratch);
  } else
    hipLaunchKernelGGL(nll_fwd_kernel<false>, dim3((unsigned)nb), dim3(NLL_TPB), 0, (hipStream_t)stream, logits, ld, target, rows,
                       (int)C, ignore_index, lse, per_point, scratch);
  hipLaunchKernelGGL(nll_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, nb, scratch + 2 * nb, loss);
  CCN_LAUNCH_OK("nll_loss_fwd");
  return CCN_OK;
}

extern "C" int ccn_nll_loss_bwd(const float* logits, int64_t ld, const int64_t* target, const float* lse, int64_t rows,
                                int64_t C, int64_t ignore_index, const float* grad_loss, const double* totals,
                                float* dlogits, int64_t ldd, void* stream) {
  CCN_REQUIRE(logits && target && lse && grad_loss && totals && dlogits && rows > 0 && C > 0 && ld >= C && ldd >= C,
              "nll_loss_bwd: bad arguments");
  const int64_t n = rows * C;
  hipLaunchKernelGGL(nll_bwd_kernel, dim3((unsigned)((n + NLL_TPB - 1) / NLL_TPB)), dim3(NLL_TPB), 0, (hipStream_t)stream,
                     logits, ld, target, lse, rows, (int)C, ignore_index, grad_loss, totals, dlogits, ldd);
  CCN_LAUNCH_OK("nll_loss_bwd");
  return CCN_OK;
}

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

// "Next" rows of SURVEY.md section 8(f): exact k-nearest neighbours for FPModule
// (pytorch3d.ops.knn_points, src/models/utils/point_ops.py:91,331), voxel sampling
// (src/models/modules/fps_ops.py:42-60) and farthest point sampling
// (pytorch3d.ops.sample_farthest_points, point_ops.py:57-70).  Built with -ffp-contract=off.
#include "ccn_common.h"

namespace {

constexpr int KNN_TPB = 256;
constexpr int KNN_TILE = 1024;
constexpr int KNN_MAXK = 8;

// one query per thread, source points streamed through LDS in tiles; top-K kept in registers
// (static indices only).  Order: ascending (d2, source index) -- sources are visited in index order and
// the insertion is strict, exactly like oracle/frnn_bruteforce.c::ccn_oracle_knn.
__global__ __launch_bounds__(KNN_TPB) void knn_points_kernel(const float* __restrict__ q,
                                                             const int64_t* __restrict__ q_ptr,
                                                             const float* __restrict__ src,
                                                             const int64_t* __restrict__ s_ptr, int K,
                                                             int64_t* __restrict__ nbr, float* __restrict__ weight) {
  __shared__ float tile[KNN_TILE * 3];
  const int64_t b = blockIdx.y;
  const int64_t q0 = q_ptr[b], nq = q_ptr[b + 1] - q0;
  const int64_t s0 = s_ptr[b], ns = s_ptr[b + 1] - s0;
  const int64_t iq = (int64_t)blockIdx.x * KNN_TPB + threadIdx.x;
  if ((int64_t)blockIdx.x * KNN_TPB >= nq) return;
  const bool live = iq < nq;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (live) {
    qx = q[3 * (q0 + iq)];
    qy = q[3 * (q0 + iq) + 1];
    qz = q[3 * (q0 + iq) + 2];
  }
  float bd[KNN_MAXK];
  int64_t bi[KNN_MAXK];
  float worst = __builtin_inff();  // current K-th best distance
#pragma unroll
  for (int t = 0; t < KNN_MAXK; ++t) {
    bd[t] = __builtin_inff();
    bi[t] = -1;
  }
  for (int64_t t0 = 0; t0 < ns; t0 += KNN_TILE) {
    const int64_t cnt = ns - t0 < KNN_TILE ? ns - t0 : KNN_TILE;
    __syncthreads();
    for (int64_t e = threadIdx.x; e < cnt * 3; e += KNN_TPB) tile[e] = src[3 * (s0 + t0) + e];
    __syncthreads();
    if (live) {
      for (int64_t j = 0; j < cnt; ++j) {
        float cd = ccn_sqdist3(tile[3 * j] - qx, tile[3 * j + 1] - qy, tile[3 * j + 2] - qz);
        if (!(cd < worst)) continue;
        int64_t ci = s0 + t0 + j;
#pragma unroll
        for (int t = 0; t < KNN_MAXK; ++t) {
          if (t < K) {
            const bool sw = cd < bd[t];
            const float td = bd[t];
            const int64_t ti = bi[t];
            bd[t] = sw ? cd : td;
            bi[t] = sw ? ci : ti;
            cd = sw ? td : cd;
            ci = sw ? ti : ci;
            if (t == K - 1) worst = bd[t];
          }
        }
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int t = 0; t < KNN_MAXK; ++t) {
    if (t < K) {
      nbr[(q0 + iq) * K + t] = bi[t];
      float w = 0.f;
      if (bi[t] >= 0) {
        // the reference recomputes the weight as 1 / clamp(sum((x - y)^2), 1e-16)  (point_ops.py:334-336)
        const float dx = src[3 * bi[t]] - qx, dy = src[3 * bi[t] + 1] - qy, dz = src[3 * bi[t] + 2] - qz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        w = __frcp_rn(d2 < 1e-16f ? 1e-16f : d2);
      }
      weight[(q0 + iq) * K + t] = w;
    }
  }
}

// ---------------------------------------------------------------- ball query (pytorch3d.ops.ball_query)
// first K points (index order) with d2 < r*r, -1 padded; one query per thread, sources through LDS tiles
__global__ __launch_bounds__(KNN_TPB) void ball_query_kernel(const float* __restrict__ q,
                                                             const int64_t* __restrict__ len1,
                                                             const float* __restrict__ src,
                                                             const int64_t* __restrict__ len2, int64_t P1, int64_t P2,
                                                             int K, float r2, int64_t* __restrict__ idx) {
  __shared__ float tile[KNN_TILE * 3];
  const int64_t b = blockIdx.y;
  const int64_t nq = len1[b], ns = len2[b];
  const int64_t iq = (int64_t)blockIdx.x * KNN_TPB + threadIdx.x;
  const bool in_range = iq < P1;
  const bool live = iq < nq;
  int64_t* out = idx + (b * P1 + (in_range ? iq : 0)) * K;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (live) {
    qx = q[3 * (b * P1 + iq)];
    qy = q[3 * (b * P1 + iq) + 1];
    qz = q[3 * (b * P1 + iq) + 2];
  }
  int have = 0;
  for (int64_t t0 = 0; t0 < ns; t0 += KNN_TILE) {
    const int64_t cnt = ns - t0 < KNN_TILE ? ns - t0 : KNN_TILE;
    __syncthreads();
    for (int64_t e = threadIdx.x; e < cnt * 3; e += KNN_TPB) tile[e] = src[3 * (b * P2 + t0) + e];
    __syncthreads();
    if (live) {
      for (int64_t j = 0; j < cnt && have < K; ++j) {
        const float d2 = ccn_sqdist3(tile[3 * j] - qx, tile[3 * j + 1] - qy, tile[3 * j + 2] - qz);
        if (d2 < r2) out[have++] = t0 + j;
      }
    }
  }
  if (in_range)
    for (int s = have; s < K; ++s) out[s] = -1;
}

// ball query in a D-dimensional feature space (dgcnn.py:114-127 DGCNNLayerRadius -> point_ops.py:81 with feature
// vectors as points): same rule as above, distance = sum_d (a_d - b_d)^2 accumulated in index order.  One query per
// thread; the candidate row address is wave-uniform (one broadcast load per element).
__global__ __launch_bounds__(KNN_TPB) void ball_query_nd_kernel(const float* __restrict__ q, int64_t ldq,
                                                                const int64_t* __restrict__ len1,
                                                                const float* __restrict__ src, int64_t lds_,
                                                                const int64_t* __restrict__ len2, int64_t P1, int64_t P2,
                                                                int D, int K, float r2, int64_t* __restrict__ idx) {
  const int64_t b = blockIdx.y;
  const int64_t nq = len1[b], ns = len2[b];
  const int64_t iq = (int64_t)blockIdx.x * KNN_TPB + threadIdx.x;
  if (iq >= P1) return;
  int64_t* out = idx + (b * P1 + iq) * K;
  int have = 0;
  if (iq < nq) {
    const float* qr = q + (b * P1 + iq) * ldq;
    for (int64_t j = 0; j < ns && have < K; ++j) {
      const float* sr = src + (b * P2 + j) * lds_;
      float d2 = 0.f;
      for (int d = 0; d < D; ++d) {
        const float diff = qr[d] - sr[d];
        d2 += diff * diff;
      }
      if (d2 < r2) out[have++] = j;
    }
  }
  for (int s = have; s < K; ++s) out[s] = -1;
}

// ---------------------------------------------------------------- voxel sampling
// key = (cloud, floor(x/v), floor(y/v), floor(z/v)) packed so that integer order == lexicographic order
// (what torch.unique(dim=0) sorts by); score = |voxel corner - p/v| + rand * v / 4  (fps_ops.py:52-56).
__global__ void voxel_keys_kernel(const float* __restrict__ pos, const int64_t* __restrict__ batch,
                                  const float* __restrict__ rnd, int64_t n, float voxel, int64_t* __restrict__ key,
                                  float* __restrict__ score, unsigned long long* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d[3];
  int64_t v[3];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float t = __fdiv_rn(pos[3 * i + a], voxel);
    const float f = floorf(t);
    v[a] = (int64_t)f;
    d[a] = (float)v[a] - t;
    ok = ok && v[a] > -(1 << 17) && v[a] < (1 << 17);
  }
  ok = ok && batch[i] >= 0 && batch[i] < 512;
  if (!ok) atomicAdd(bad, 1ULL);
  const int64_t off = 1 << 17;  // 18 bits per axis, 9 bits of cloud id
  key[i] = (batch[i] << 54) | ((v[0] + off) << 36) | ((v[1] + off) << 18) | (v[2] + off);
  const float dist = sqrtf(ccn_sqdist3(d[0], d[1], d[2]));
  score[i] = dist + __fdiv_rn(rnd[i] * voxel, 4.0f);
}

// per voxel the point with the smallest score (ties: smallest index), as one 64-bit atomicMin on
// (score bits, index): scores are >= 0 so their bit patterns order like the values
__global__ void voxel_argmin_kernel(const float* __restrict__ score, const int64_t* __restrict__ voxel_of, int64_t n,
                                    unsigned long long* __restrict__ best) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long packed = ((unsigned long long)__float_as_uint(score[i]) << 32) | (unsigned long long)i;
  atomicMin(&best[voxel_of[i]], packed);
}

__global__ void voxel_unpack_kernel(const unsigned long long* __restrict__ best, int64_t m, int64_t* __restrict__ idx) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < m) idx[v] = (int64_t)(best[v] & 0xffffffffULL);
}

// ---------------------------------------------------------------- farthest point sampling
// one workgroup per cloud; every iteration updates the distance-to-chosen-set of all points and takes the
// arg max (ties: smallest index).  d2 = (dx*dx + dy*dy) + dz*dz, matching the oracle.
constexpr int FPS_TPB = 1024;

__global__ __launch_bounds__(FPS_TPB) void fps_kernel(const float* __restrict__ pos,
                                                      const int64_t* __restrict__ cloud_ptr,
                                                      const int64_t* __restrict__ start,
                                                      const int64_t* __restrict__ out_ptr, float* __restrict__ mind,
                                                      int64_t* __restrict__ out) {
  __shared__ float red_v[FPS_TPB / 64];
  __shared__ int red_i[FPS_TPB / 64];
  __shared__ int chosen;
  const int64_t b = blockIdx.x;
  const int64_t p0 = cloud_ptr[b];
  const int n = (int)(cloud_ptr[b + 1] - p0);
  const int64_t o0 = out_ptr[b];
  const int keep = (int)(out_ptr[b + 1] - o0);
  if (n <= 0 || keep <= 0) return;
  const float* p = pos + 3 * p0;
  float* md = mind + p0;
  for (int i = threadIdx.x; i < n; i += FPS_TPB) md[i] = __builtin_inff();
  int cur = (int)start[b];
  cur = cur < 0 ? 0 : (cur >= n ? n - 1 : cur);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < keep; ++it) {
    if (threadIdx.x == 0) out[o0 + it] = p0 + cur;
    const float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += FPS_TPB) {
      const float dx = p[3 * i] - cx, dy = p[3 * i + 1] - cy, dz = p[3 * i + 2] - cz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      const float m = fminf(md[i], d2);
      md[i] = m;
      if (m > bv) {  // i increases within a thread: strict > keeps the smallest index
        bv = m;
        bi = i;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      red_v[wave] = bv;
      red_i[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = red_v[0];
      int ix = red_i[0];
      for (int w = 1; w < FPS_TPB / 64; ++w)
        if (red_v[w] > v || (red_v[w] == v && red_i[w] < ix)) {
          v = red_v[w];
          ix = red_i[w];
        }
      chosen = ix;
    }
    __syncthreads();
    cur = chosen;
  }
}

}  // namespace

extern "C" {

int ccn_knn_points(const float* q, const int64_t* q_ptr, const float* src, const int64_t* s_ptr, int64_t B,
                   int64_t max_q, int64_t K, int64_t* nbr, float* weight, void* stream) {
  CCN_REQUIRE(q && q_ptr && src && s_ptr && nbr && weight && B > 0 && B < 65536 && max_q > 0,
              "knn_points: bad arguments");
  CCN_REQUIRE(K >= 1 && K <= KNN_MAXK, "knn_points: K must be in [1, %d]", KNN_MAXK);
  hipLaunchKernelGGL(knn_points_kernel, dim3(ccn_blocks(max_q, KNN_TPB), (unsigned)B), dim3(KNN_TPB), 0,
                     (hipStream_t)stream, q, q_ptr, src, s_ptr, (int)K, nbr, weight);
  CCN_LAUNCH_OK("knn_points");
  return CCN_OK;
}

int ccn_ball_query(const float* points1, const int64_t* lengths1, const float* points2, const int64_t* lengths2,
                   int64_t B, int64_t P1, int64_t P2, int64_t K, float radius, int64_t* idx, void* stream) {
  CCN_REQUIRE(points1 && lengths1 && points2 && lengths2 && idx && B > 0 && B < 65536 && P1 > 0 && P2 > 0 && K > 0 &&
                  K < (1 << 20),
              "ball_query: bad arguments");
  hipLaunchKernelGGL(ball_query_kernel, dim3(ccn_blocks(P1, KNN_TPB), (unsigned)B), dim3(KNN_TPB), 0,
                     (hipStream_t)stream, points1, lengths1, points2, lengths2, P1, P2, (int)K, radius * radius, idx);
  CCN_LAUNCH_OK("ball_query");
  return CCN_OK;
}

int ccn_ball_query_nd(const float* points1, int64_t ld1, const int64_t* lengths1, const float* points2, int64_t ld2,
                      const int64_t* lengths2, int64_t B, int64_t P1, int64_t P2, int64_t D, int64_t K, float radius,
                      int64_t* idx, void* stream) {
  CCN_REQUIRE(points1 && lengths1 && points2 && lengths2 && idx && B > 0 && B < 65536 && P1 > 0 && P2 > 0 && K > 0 &&
                  K < (1 << 20) && D > 0 && D < (1 << 20) && ld1 >= D && ld2 >= D,
              "ball_query_nd: bad arguments");
  hipLaunchKernelGGL(ball_query_nd_kernel, dim3(ccn_blocks(P1, KNN_TPB), (unsigned)B), dim3(KNN_TPB), 0,
                     (hipStream_t)stream, points1, ld1, lengths1, points2, ld2, lengths2, P1, P2, (int)D, (int)K,
                     radius * radius, idx);
  CCN_LAUNCH_OK("ball_query_nd");
  return CCN_OK;
}

int ccn_voxel_keys(const float* pos, const int64_t* batch, const float* rnd, int64_t n, float voxel, int64_t* key,
                   float* score, int64_t* bad, void* stream) {
  CCN_REQUIRE(pos && batch && rnd && key && score && bad && n > 0 && voxel > 0.f, "voxel_keys: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  CCN_HIP(hipMemsetAsync(bad, 0, sizeof(int64_t), s), "voxel_keys");
  hipLaunchKernelGGL(voxel_keys_kernel, dim3(ccn_blocks(n, 256)), dim3(256), 0, s, pos, batch, rnd, n, voxel, key, score,
                     (unsigned long long*)bad);
  CCN_LAUNCH_OK("voxel_keys");
  return CCN_OK;
}

int ccn_voxel_argmin(const float* score, const int64_t* voxel_of, int64_t n, int64_t num_voxels, int64_t* scratch,
                     int64_t* idx, void* stream) {
  CCN_REQUIRE(score && voxel_of && scratch && idx && n > 0 && num_voxels > 0, "voxel_argmin: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  CCN_HIP(hipMemsetAsync(scratch, 0xff, (size_t)num_voxels * 8, s), "voxel_argmin");
  hipLaunchKernelGGL(voxel_argmin_kernel, dim3(ccn_blocks(n, 256)), dim3(256), 0, s, score, voxel_of, n,
                     (unsigned long long*)scratch);
  hipLaunchKernelGGL(voxel_unpack_kernel, dim3(ccn_blocks(num_voxels, 256)), dim3(256), 0, s,
                     (const unsigned long long*)scratch, num_voxels, idx);
  CCN_LAUNCH_OK("voxel_argmin");
  return CCN_OK;
}

int ccn_fps(const float* pos, const int64_t* cloud_ptr, const int64_t* start, const int64_t* out_ptr, int64_t B,
            float* mind, int64_t* out, void* stream) {
  CCN_REQUIRE(pos && cloud_ptr && start && out_ptr && mind && out && B > 0, "fps: bad arguments");
  hipLaunchKernelGGL(fps_kernel, dim3((unsigned)B), dim3(FPS_TPB), 0, (hipStream_t)stream, pos, cloud_ptr, start, out_ptr,
                     mind, out);
  CCN_LAUNCH_OK("fps");
  return CCN_OK;
}

}  // extern "C"

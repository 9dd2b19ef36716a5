// "Next" rows of SURVEY.md section 8(f): exact k-nearest neighbours for FPModule
// (pytorch3d.ops.knn_points, src/models/utils/point_ops.py:91,331), voxel sampling
// (src/models/modules/fps_ops.py:42-60) and farthest point sampling
// (pytorch3d.ops.sample_farthest_points, point_ops.py:57-70).  Built with -ffp-contract=off.
#include "ccn_common.h"

#include <atomic>
#include <mutex>

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

// ---------------------------------------------------------------- grid-accelerated exact kNN (large clouds)
// The exhaustive kernel above costs P1*P2 distance evaluations per cloud (1e10 per call at the benchmark size).
// For large clouds the K nearest neighbours are searched with the FRNN hash grid at a per-cloud radius derived from
// the point density; a query whose K-th neighbour was not found inside that radius is recomputed exhaustively, so
// the result is the exact kNN for every query (same distance arithmetic and (d2, index) order as above).
__device__ __forceinline__ float knn_weight(const float* __restrict__ src, int64_t j, float qx, float qy, float qz) {
  // the reference recomputes the weight as 1 / clamp(sum((x - y)^2), 1e-16)  (point_ops.py:334-336)
  const float dx = src[3 * j] - qx, dy = src[3 * j + 1] - qy, dz = src[3 * j + 2] - qz;
  const float d2 = (dx * dx + dy * dy) + dz * dz;
  return __frcp_rn(d2 < 1e-16f ? 1e-16f : d2);
}

// r[b] = scale * cbrt(bounding-box volume / points) of the source cloud (extents floored at 1 % of the largest)
__global__ __launch_bounds__(256) void knn_cloud_radius_kernel(const float* __restrict__ src,
                                                               const int64_t* __restrict__ s_ptr, float scale,
                                                               float* __restrict__ radius) {
  __shared__ float red[6][256];
  const int64_t b = blockIdx.x, s0 = s_ptr[b], ns = s_ptr[b + 1] - s0;
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-lo[0], -lo[0], -lo[0]};
  for (int64_t j = threadIdx.x; j < ns; j += 256)
    for (int d = 0; d < 3; ++d) {
      const float v = src[3 * (s0 + j) + d];
      lo[d] = fminf(lo[d], v);
      hi[d] = fmaxf(hi[d], v);
    }
  for (int d = 0; d < 3; ++d) {
    red[d][threadIdx.x] = lo[d];
    red[3 + d][threadIdx.x] = hi[d];
  }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int d = 0; d < 3; ++d) {
        red[d][threadIdx.x] = fminf(red[d][threadIdx.x], red[d][threadIdx.x + w]);
        red[3 + d][threadIdx.x] = fmaxf(red[3 + d][threadIdx.x], red[3 + d][threadIdx.x + w]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float e[3], big = 0.f;
    for (int d = 0; d < 3; ++d) {
      e[d] = ns > 0 ? red[3 + d][0] - red[d][0] : 0.f;
      big = fmaxf(big, e[d]);
    }
    float vol = 1.f;
    for (int d = 0; d < 3; ++d) vol *= fmaxf(e[d], 0.01f * big);
    const float r = scale * cbrtf(vol / (float)(ns > 0 ? ns : 1));
    radius[b] = (r > 0.f && r < 1e30f) ? r : 1.0f;
  }
}

// FRNN result (B, P1, K) of cloud-local indices -> packed neighbour table + weights; flag = 1 where the search radius
// did not hold min(K, source points) neighbours (those rows are rewritten by the exhaustive kernel below).
__global__ void knn_from_grid_kernel(const int64_t* __restrict__ idx, const float* __restrict__ q,
                                     const int64_t* __restrict__ q_ptr, const float* __restrict__ src,
                                     const int64_t* __restrict__ s_ptr, int64_t P1, int K, int64_t* __restrict__ nbr,
                                     float* __restrict__ weight, int32_t* __restrict__ flag) {
  const int64_t b = blockIdx.y, iq = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t q0 = q_ptr[b], nq = q_ptr[b + 1] - q0;
  if (iq >= nq) return;
  const int64_t s0 = s_ptr[b], ns = s_ptr[b + 1] - s0;
  const float qx = q[3 * (q0 + iq)], qy = q[3 * (q0 + iq) + 1], qz = q[3 * (q0 + iq) + 2];
  const int64_t* row = idx + (b * P1 + iq) * K;
  int found = 0;
  for (int t = 0; t < K; ++t) {
    const int64_t j = row[t];
    if (j >= 0) ++found;
    nbr[(q0 + iq) * K + t] = j >= 0 ? s0 + j : -1;
    weight[(q0 + iq) * K + t] = j >= 0 ? knn_weight(src, s0 + j, qx, qy, qz) : 0.f;
  }
  flag[q0 + iq] = found < (ns < K ? (int)ns : K) ? 1 : 0;
}

__global__ void scatter_flagged_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ offsets, int64_t n,
                                       int64_t* __restrict__ list) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) list[offsets[i]] = i;
}

// exhaustive search for the listed queries only: one WAVE per query (the list is short, so a thread per query would
// leave the chip idle and pay the full latency of a 25k-point scan per thread).  Lane l visits sources l, l+64, ...
// in index order and keeps its own ascending top-K; the K results are then drawn by K wave-wide minima over the
// lanes' list heads, ordered by (d2, index) -- the same order the sequential scan produces.
__global__ __launch_bounds__(KNN_TPB) void knn_points_list_kernel(const float* __restrict__ q,
                                                                  const int64_t* __restrict__ q_ptr,
                                                                  const float* __restrict__ src,
                                                                  const int64_t* __restrict__ s_ptr, int B, int K,
                                                                  const int64_t* __restrict__ list,
                                                                  const int64_t* __restrict__ count,
                                                                  int64_t* __restrict__ nbr, float* __restrict__ weight) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (KNN_TPB / 64) + (threadIdx.x >> 6);
  if (t >= count[0]) return;
  const int64_t iq = list[t];
  int b = 0;
  while (b + 1 < B && q_ptr[b + 1] <= iq) ++b;
  const int64_t s0 = s_ptr[b], ns = s_ptr[b + 1] - s0;
  const float qx = q[3 * iq], qy = q[3 * iq + 1], qz = q[3 * iq + 2];
  float bd[KNN_MAXK];
  int64_t bi[KNN_MAXK];
  float worst = __builtin_inff();
#pragma unroll
  for (int u = 0; u < KNN_MAXK; ++u) {
    bd[u] = __builtin_inff();
    bi[u] = -1;
  }
  for (int64_t j = lane; j < ns; j += 64) {
    float cd = ccn_sqdist3(src[3 * (s0 + j)] - qx, src[3 * (s0 + j) + 1] - qy, src[3 * (s0 + j) + 2] - qz);
    if (!(cd < worst)) continue;
    int64_t ci = s0 + j;
#pragma unroll
    for (int u = 0; u < KNN_MAXK; ++u) {
      if (u < K) {
        const bool sw = cd < bd[u];
        const float td = bd[u];
        const int64_t ti = bi[u];
        bd[u] = sw ? cd : td;
        bi[u] = sw ? ci : ti;
        cd = sw ? td : cd;
        ci = sw ? ti : ci;
        if (u == K - 1) worst = bd[u];
      }
    }
  }
  for (int r = 0; r < K; ++r) {
    // wave-wide minimum of the list heads by (d2, index); exhausted lists offer (inf, -1) and lose to any real entry
    float md = bd[0];
    int64_t mi = bi[0];
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_xor(md, off, 64);
      const int64_t oi = __shfl_xor(mi, off, 64);
      const bool take = oi >= 0 && (mi < 0 || od < md || (od == md && oi < mi));
      md = take ? od : md;
      mi = take ? oi : mi;
    }
    if (mi >= 0 && bi[0] == mi) {  // the winning lane pops its head
#pragma unroll
      for (int u = 0; u + 1 < KNN_MAXK; ++u) {
        bd[u] = bd[u + 1];
        bi[u] = bi[u + 1];
      }
      bd[KNN_MAXK - 1] = __builtin_inff();
      bi[KNN_MAXK - 1] = -1;
    }
    if (lane == 0) {
      nbr[iq * K + r] = mi;
      weight[iq * K + r] = mi >= 0 ? knn_weight(src, mi, qx, qy, qz) : 0.f;
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
                                    int64_t num_voxels, unsigned long long* __restrict__ best) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = voxel_of[i];
  // (num_voxels may be a CAPACITY -- a caller that keeps the counts on the device: a voxel past it has no slot and is dropped;
  // round 6: this was an unguarded write, found by replaying a deliberately overflowing batch)
  if (v < 0 || v >= num_voxels) return;
  const unsigned long long packed = ((unsigned long long)__float_as_uint(score[i]) << 32) | (unsigned long long)i;
  atomicMin(&best[v], packed);
}

__global__ void voxel_unpack_kernel(const unsigned long long* __restrict__ best, int64_t m, int64_t* __restrict__ idx) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < m) idx[v] = (int64_t)(best[v] & 0xffffffffULL);
}

// ---------------------------------------------------------------- farthest point sampling
// one workgroup per cloud; every iteration updates the distance-to-chosen-set of all points and takes the
// arg max (ties: smallest index).  d2 = (dx*dx + dy*dy) + dz*dz, matching the oracle.
constexpr int FPS_TPB = 1024;
constexpr int FPS_CLAIM_MAX = 96 * 1024;   // dynamic LDS a sampling workgroup claims to keep its CU to itself (see ccn_fps)
static std::atomic<int> g_fps_claim{FPS_CLAIM_MAX};    // A/B hook: ccn_fps_set_lds_claim
static std::atomic<int> g_fps_fault{0};                // test hook: ccn_fps_debug_fault
static std::atomic<int> g_fps_cluster{1};              // A/B hook: ccn_fps_use_cluster (0 = one workgroup per cloud whatever its size, 2 = the cluster with agent-scope stores whatever the placement)

// The cluster form (fps_cluster_kernel, below) gives a cloud up when a member never hears from a partner: it raises the cloud's
// abort word.  The launch behind it on the same stream -- this kernel or fps_hybrid_kernel with `gate` = the exchange areas --
// re-samples exactly those clouds with one workgroup each (same samples: every form is bit-identical), so the result of ccn_fps
// never depends on the co-residency of a cluster's members; `fallbacks` (optional) counts the clouds that took this way.
constexpr int FPS_CL_AREA = 512;                 // bytes of exchange area per cloud: 2 parities x 4 workgroups x 5 granules x 8 B, abort word, 4 XCD ids
constexpr int FPS_CL_ABORT_AT = 2 * 4 * 5 * 8;   // byte offset of the abort word in a cloud's area
__device__ __forceinline__ bool fps_gate_open(const char* gate, int64_t b, int32_t* fallbacks) {
  const uint32_t ab = *reinterpret_cast<const uint32_t*>(gate + b * FPS_CL_AREA + FPS_CL_ABORT_AT);
  if (ab && fallbacks && threadIdx.x == 0) atomicAdd(fallbacks, 1);
  return ab != 0;
}

__global__ __launch_bounds__(FPS_TPB) void fps_kernel(const float* __restrict__ pos,
                                                      const int64_t* __restrict__ cloud_ptr,
                                                      const int64_t* __restrict__ start,
                                                      const int64_t* __restrict__ out_ptr, float* __restrict__ mind,
                                                      int64_t* __restrict__ out, const char* __restrict__ gate,
                                                      int32_t* __restrict__ fallbacks) {
  __shared__ float red_v[FPS_TPB / 64];
  __shared__ int red_i[FPS_TPB / 64];
  __shared__ int chosen;
  const int64_t b = blockIdx.x;
  if (gate && !fps_gate_open(gate, b, fallbacks)) return;    // (the cluster's fallback: only the clouds whose cluster gave up)
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

// Register-resident form for clouds of at most PPT*1024 points: each thread owns PPT points (coordinates and running
// minimum distance in registers), so an iteration touches no global memory; the arg-max reduction carries the
// winner's coordinates with it (wave shuffles, then the 16 per-wave entries through a double-buffered LDS table),
// which leaves ONE barrier per selected sample.  Same arithmetic and tie rule as fps_kernel.
struct FpsBest {
  float v;
  int i;
  float x, y, z;
};
__device__ __forceinline__ void fps_take(FpsBest& a, const FpsBest& o) {
  if (o.v > a.v || (o.v == a.v && o.i < a.i)) a = o;
}
__device__ __forceinline__ FpsBest fps_shfl(const FpsBest& a, int off) {
  FpsBest o;
  o.v = __shfl_xor(a.v, off, 64);
  o.i = __shfl_xor(a.i, off, 64);
  o.x = __shfl_xor(a.x, off, 64);
  o.y = __shfl_xor(a.y, off, 64);
  o.z = __shfl_xor(a.z, off, 64);
  return o;
}

// The arg-max of a wave (round 5).  Rounds 1-4 ran a 6-step butterfly of five __shfl_xor each (ds_bpermute_b32: ~30 dependent
// trips through the LDS crossbar per round, and 20 more for the 16 per-wave entries) -- a third of a round of the
// register-resident forms.  Here: the maximum by four DPP steps inside each row of 16 lanes (quad permutes, row_half_mirror,
// row_mirror: VALU speed) and three scalar maxima over the rows; the smallest index among the lanes that hold it the same way;
// the winner's coordinates by v_readlane from the one lane that has that index.  Same rule (larger value, then smaller index).
template <int CTRL>
__device__ __forceinline__ float fps_dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int fps_dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ float fps_wave_max(float v) {
  v = fmaxf(v, fps_dpp_f<0xB1>(v));     // quad_perm [1,0,3,2]
  v = fmaxf(v, fps_dpp_f<0x4E>(v));     // quad_perm [2,3,0,1]
  v = fmaxf(v, fps_dpp_f<0x141>(v));    // row_half_mirror
  v = fmaxf(v, fps_dpp_f<0x140>(v));    // row_mirror: every lane of a row holds the row's maximum
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int fps_wave_min(int v) {
  int o;
  o = fps_dpp_i<0xB1>(v);  v = o < v ? o : v;
  o = fps_dpp_i<0x4E>(v);  v = o < v ? o : v;
  o = fps_dpp_i<0x141>(v); v = o < v ? o : v;
  o = fps_dpp_i<0x140>(v); v = o < v ? o : v;
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const int ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}
__device__ __forceinline__ FpsBest fps_wave_best(const FpsBest& mine) {
  FpsBest out;
  out.v = fps_wave_max(mine.v);
  out.i = fps_wave_min(mine.v == out.v ? mine.i : 0x7fffffff);
  const unsigned long long who = __ballot(mine.v == out.v && mine.i == out.i);
  const int src = who ? __builtin_ctzll(who) : 0;
  out.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.x), src));
  out.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.y), src));
  out.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.z), src));
  return out;
}

template <int PPT>
__global__ __launch_bounds__(FPS_TPB) void fps_reg_kernel(const float* __restrict__ pos,
                                                          const int64_t* __restrict__ cloud_ptr,
                                                          const int64_t* __restrict__ start,
                                                          const int64_t* __restrict__ out_ptr,
                                                          int64_t* __restrict__ out) {
  __shared__ FpsBest red[2][FPS_TPB / 64];
  const int64_t b = blockIdx.x;
  const int64_t p0 = cloud_ptr[b];
  const int n = (int)(cloud_ptr[b + 1] - p0);
  const int64_t o0 = out_ptr[b];
  const int keep = (int)(out_ptr[b + 1] - o0);
  if (n <= 0 || keep <= 0) return;
  const float* p = pos + 3 * p0;
  float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int i = threadIdx.x + k * FPS_TPB;
    const bool in = i < n;
    px[k] = in ? p[3 * i] : 0.f;
    py[k] = in ? p[3 * i + 1] : 0.f;
    pz[k] = in ? p[3 * i + 2] : 0.f;
    md[k] = in ? __builtin_inff() : -2.f;  // never the maximum
  }
  int cur = (int)start[b];
  cur = cur < 0 ? 0 : (cur >= n ? n - 1 : cur);
  float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < keep; ++it) {
    if (threadIdx.x == 0) out[o0 + it] = p0 + cur;
    FpsBest best = {-1.f, 0x7fffffff, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      const float m = md[k] < 0.f ? md[k] : fminf(md[k], d2);
      md[k] = m;
      if (m > best.v) {  // k (hence the point index) increases: strict > keeps the smallest index
        best.v = m;
        best.i = threadIdx.x + k * FPS_TPB;
        best.x = px[k];
        best.y = py[k];
        best.z = pz[k];
      }
    }
    best = fps_wave_best(best);
    FpsBest* tab = red[it & 1];
    if (lane == 0) tab[wave] = best;
    __syncthreads();
    FpsBest all = fps_wave_best(tab[lane & (FPS_TPB / 64 - 1)]);     // (the 16 entries, four times over: maxima and minima do not mind)
    cur = all.i;
    cx = all.x;
    cy = all.y;
    cz = all.z;
  }
}

// Clouds beyond the register-resident form's 16 k points (round 3: the exact FPS over A2D2's ~49 k-point clouds was 37.8 ms per
// step in fps_kernel, the critical path of BASELINE configs[4] -- one CU streams 20 bytes per point and round through its L1).
// Hybrid: the first PR points of every thread live in registers as above; for the rest only the coordinates are re-read
// from global memory (12 bytes per point and round), their running minima live in LDS (up to FPS_HYB_LDS floats), and the
// winner's coordinates travel with the reduction: one barrier per sample, no dependent global load.  Same arithmetic, same
// visiting order per thread (indices increase), same tie rule as fps_kernel.
constexpr int FPS_HYB_LDS = 36864;      // 144 KB of running minima
template <int PR>
__global__ __launch_bounds__(FPS_TPB) void fps_hybrid_kernel(const float* __restrict__ pos,
                                                             const int64_t* __restrict__ cloud_ptr,
                                                             const int64_t* __restrict__ start,
                                                             const int64_t* __restrict__ out_ptr,
                                                             int64_t* __restrict__ out, const char* __restrict__ gate,
                                                             int32_t* __restrict__ fallbacks) {
  extern __shared__ float md_l[];                       // [n - PR * FPS_TPB] running minima of the points beyond the registers
  __shared__ FpsBest red[2][FPS_TPB / 64];
  const int64_t b = blockIdx.x;
  if (gate && !fps_gate_open(gate, b, fallbacks)) return;    // (the cluster's fallback, see fps_gate_open)
  const int64_t p0 = cloud_ptr[b];
  const int n = (int)(cloud_ptr[b + 1] - p0);
  const int64_t o0 = out_ptr[b];
  const int keep = (int)(out_ptr[b + 1] - o0);
  if (n <= 0 || keep <= 0) return;
  const float* p = pos + 3 * p0;
  float px[PR], py[PR], pz[PR], md[PR];
#pragma unroll
  for (int k = 0; k < PR; ++k) {
    const int i = threadIdx.x + k * FPS_TPB;
    const bool in = i < n;
    px[k] = in ? p[3 * i] : 0.f;
    py[k] = in ? p[3 * i + 1] : 0.f;
    pz[k] = in ? p[3 * i + 2] : 0.f;
    md[k] = in ? __builtin_inff() : -2.f;  // never the maximum
  }
  const int rest0 = PR * FPS_TPB;
  for (int i = rest0 + threadIdx.x; i < n; i += FPS_TPB) md_l[i - rest0] = __builtin_inff();
  int cur = (int)start[b];
  cur = cur < 0 ? 0 : (cur >= n ? n - 1 : cur);
  float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < keep; ++it) {
    if (threadIdx.x == 0) out[o0 + it] = p0 + cur;
    FpsBest best = {-1.f, 0x7fffffff, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PR; ++k) {
      const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      const float m = md[k] < 0.f ? md[k] : fminf(md[k], d2);
      md[k] = m;
      if (m > best.v) {  // k (hence the point index) increases: strict > keeps the smallest index
        best.v = m;
        best.i = threadIdx.x + k * FPS_TPB;
        best.x = px[k];
        best.y = py[k];
        best.z = pz[k];
      }
    }
    for (int i0 = rest0 + threadIdx.x; i0 < n; i0 += 4 * FPS_TPB) {
      float x[4], y[4], z[4], m0[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {      // (four points' loads in flight together)
        const int i = i0 + u * FPS_TPB;
        const int ic = i < n ? i : i0;
        x[u] = p[3 * ic];
        y[u] = p[3 * ic + 1];
        z[u] = p[3 * ic + 2];
        m0[u] = md_l[ic - rest0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * FPS_TPB;
        if (i >= n) break;
        const float dx = x[u] - cx, dy = y[u] - cy, dz = z[u] - cz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        const float m = fminf(m0[u], d2);
        md_l[i - rest0] = m;
        if (m > best.v) {
          best.v = m;
          best.i = i;
          best.x = x[u];
          best.y = y[u];
          best.z = z[u];
        }
      }
    }
    best = fps_wave_best(best);
    FpsBest* tab = red[it & 1];
    if (lane == 0) tab[wave] = best;
    __syncthreads();
    FpsBest all = fps_wave_best(tab[lane & (FPS_TPB / 64 - 1)]);     // (the 16 entries, four times over: maxima and minima do not mind)
    cur = all.i;
    cx = all.x;
    cy = all.y;
    cz = all.z;
  }
}


// ---- exact FPS over a cloud of more than 16 k points by a CLUSTER of G workgroups (round 5).
// The hybrid form above is bound by what one CU can pull through its L1 per round (the coordinates of the points beyond its
// registers: 400 KB per round for a 49 k-point cloud, 2.4 us per round, 12 k dependent rounds).  Here G <= 4 workgroups share the
// cloud, every point lives in registers (16 per thread), and a round exchanges the G candidates through global memory
// (measured: 28.7 -> 23.6 ms for the 49 k -> 12 k level of configs[4]; whole forward 54.7 -> 52.2 ms):
//   * every workgroup finds its own best point as fps_reg_kernel does (one barrier), then lanes 0..4 of its wave 0 publish it as
//     five 8-byte granules {field, round tag} (stored with agent scope, `sc1`, when a partner sits on another XCD, whose L2 is
//     not coherent with this one -- MI355X_MICROARCH.md, inter-workgroup visibility) into the slot of the round's parity;
//   * EVERY wave of every workgroup then polls the other workgroups' granules itself (agent-scope loads, tag == round + 1) and
//     reduces the G candidates redundantly: no second barrier, nothing to broadcast inside the workgroup.
// Two slots by round parity suffice: a workgroup publishes round r + 2 only after it has the others' round r + 1, which they publish
// behind a barrier that all their waves reach after reading round r.  Same arithmetic, same tie rule (larger value, then smaller
// index) as every other form: bit-identical samples.  The cluster's workgroup ids are congruent modulo 8 (one XCD under the
// observed round-robin placement: speed only, the protocol does not depend on it).  Co-residency is NOT guaranteed by an ordinary
// launch (G x B is held to half of the device's CUs, each workgroup a whole CU, but a busy device may still start one member long
// after the other): a poll that sees nothing for ~2^18 tries (~0.5 s; a round is 2 us) raises the cloud's abort word, every wave of
// every member leaves, and the gated one-workgroup launch behind this kernel re-samples that cloud (fps_gate_open): the samples
// ccn_fps returns never depend on the placement.  `fault` (test hook ccn_fps_debug_fault): 1 = member 1 silently leaves before
// round 1 (its partners run into the timeout), 2 = member 1 raises the abort word itself and leaves (the fast way out).
constexpr int FPS_CL_MAXG = 4;
static_assert(FPS_CL_ABORT_AT == 2 * FPS_CL_MAXG * 5 * 8, "abort word sits behind the two parities' granules");
struct FpsGranule {
  uint32_t value, tag;
};
__device__ __forceinline__ void fps_store_sc1(FpsGranule* at, uint32_t value, uint32_t tag) {
  const uint64_t both = ((uint64_t)tag << 32) | value;          // (value in the low dword, tag in the high one: ONE 8-byte store)
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(at), "v"(both) : "memory");
}
__device__ __forceinline__ void fps_store_plain(FpsGranule* at, uint32_t value, uint32_t tag) {
  const uint64_t both = ((uint64_t)tag << 32) | value;
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(at), "v"(both) : "memory");
}
__device__ __forceinline__ uint2 fps_load_sc1(const FpsGranule* at) {
  uint64_t both;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(both) : "v"(at) : "memory");
  return make_uint2((uint32_t)both, (uint32_t)(both >> 32));
}

template <int PPT>
__global__ __launch_bounds__(FPS_TPB) void fps_cluster_kernel(const float* __restrict__ pos,
                                                              const int64_t* __restrict__ cloud_ptr,
                                                              const int64_t* __restrict__ start,
                                                              const int64_t* __restrict__ out_ptr, int64_t B, int G, int force_sc1,
                                                              int fault, char* __restrict__ xch_all, int64_t* __restrict__ out) {
  __shared__ FpsBest red[2][FPS_TPB / 64];
  // workgroup ids x, x + 8, x + 16, ... (x = id % 8) form the clusters x, x, ... in turn: cluster c = x + 8 * (t / G), member t % G
  const int64_t t = blockIdx.x >> 3;
  const int64_t b = (blockIdx.x & 7) + 8 * (t / G);
  const int g = (int)(t % G);
  if (b >= B) return;
  const int64_t p0 = cloud_ptr[b];
  const int n = (int)(cloud_ptr[b + 1] - p0);
  const int64_t o0 = out_ptr[b];
  const int keep = (int)(out_ptr[b + 1] - o0);
  if (n <= 0 || keep <= 0) return;
  const float* p = pos + 3 * p0;
  const int S = (n + G - 1) / G;                   // points per member (<= PPT * FPS_TPB: checked by the host against max_cloud)
  const int lo = g * S, hi = lo + S < n ? lo + S : n;
  float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int i = lo + threadIdx.x + k * FPS_TPB;
    const bool in = i < hi;
    px[k] = in ? p[3 * i] : 0.f;
    py[k] = in ? p[3 * i + 1] : 0.f;
    pz[k] = in ? p[3 * i + 2] : 0.f;
    md[k] = in ? __builtin_inff() : -2.f;  // never the maximum
  }
  int cur = (int)start[b];
  cur = cur < 0 ? 0 : (cur >= n ? n - 1 : cur);
  float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  FpsGranule* const xch = reinterpret_cast<FpsGranule*>(xch_all + b * FPS_CL_AREA);
  uint32_t* const abort_word = reinterpret_cast<uint32_t*>(xch_all + b * FPS_CL_AREA + FPS_CL_ABORT_AT);
  const int h_of = lane / 5, f_of = lane - 5 * h_of;      // lane = 5 * member + field (lanes 0 .. 5 G - 1 poll)
  // Where do the members sit?  Every member publishes the id of its XCD (hardware register XCC_ID) once, with agent scope, and reads
  // the others'.  All on one XCD (the usual case: ids congruent modulo 8): the granules of a round are written with PLAIN stores,
  // which keep the line in that XCD's L2, where the partners' L1-bypassing loads find it a few hundred ns later.  Otherwise they are
  // written through with `sc1` (the line leaves the L2; readers on any XCD see it, at the cross-XCD price: measured on the
  // 49 k-point clouds of BASELINE configs[4], 12 k rounds -- one workgroup (hybrid form) 2.4 us per round, this cluster with `sc1`
  // stores 2.4, with plain stores 1.9: a round is now the store's way into the L2 plus one L2 round trip of the poll).  Every member
  // sees the same ids and takes the same decision; correctness never depends on the placement.
  bool same_xcd = true;
  {
    FpsGranule* const where = xch + 2 * FPS_CL_MAXG * 5 + 1;          // (behind the abort word's granule)
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (wave == 0 && lane == 0) fps_store_sc1(where + g, xcc, 0xC0DE0001u);
    bool done = lane >= G;
    uint32_t theirs = xcc;
    for (int tries = 0;; ++tries) {
      if (!done) {
        const uint2 r = fps_load_sc1(where + lane);
        if (r.y == 0xC0DE0001u) {
          theirs = r.x;
          done = true;
        }
      }
      if (__ballot(!done) == 0ull) break;
      if (tries >= (1 << 19)) {
        if (lane == 0) atomicExch(abort_word, 1u);
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    same_xcd = !force_sc1 && __ballot(theirs != xcc) == 0ull;
  }
  for (int it = 0; it < keep; ++it) {
    if (fault && g == 1 && it == 1) {                    // test hook: this member is gone (scalar condition, never taken in production)
      if (fault == 2 && threadIdx.x == 0) atomicExch(abort_word, 1u);
      return;
    }
    if (g == 0 && threadIdx.x == 0) out[o0 + it] = p0 + cur;
    FpsBest best = {-1.f, 0x7fffffff, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      const float m = md[k] < 0.f ? md[k] : fminf(md[k], d2);
      md[k] = m;
      if (m > best.v) {  // k (hence the point index) increases: strict > keeps the smallest index
        best.v = m;
        best.i = lo + threadIdx.x + k * FPS_TPB;
        best.x = px[k];
        best.y = py[k];
        best.z = pz[k];
      }
    }
    best = fps_wave_best(best);
    FpsBest* tab = red[it & 1];
    if (lane == 0) tab[wave] = best;
    __syncthreads();
    FpsBest all = fps_wave_best(tab[lane & (FPS_TPB / 64 - 1)]);     // (the 16 entries, four times over: maxima and minima do not mind)
    // ---- the cluster's exchange
    FpsGranule* const slot = xch + (it & 1) * (FPS_CL_MAXG * 5);
    const uint32_t tag = (uint32_t)it + 1u;
    if (wave == 0 && lane < 5) {
      const uint32_t field = lane == 0 ? __float_as_uint(all.v) : lane == 1 ? (uint32_t)all.i : lane == 2 ? __float_as_uint(all.x)
                             : lane == 3 ? __float_as_uint(all.y) : __float_as_uint(all.z);
      if (same_xcd)
        fps_store_plain(slot + g * 5 + lane, field, tag);
      else
        fps_store_sc1(slot + g * 5 + lane, field, tag);
    }
    uint32_t got = 0;
    const bool polls = lane < 5 * G && h_of != g;
    bool done = !polls;
    // Four agent-scope loads of the granule in flight, re-issued as each returns (they return in order): a poll leaves every
    // quarter of a round trip, so the partner's store is seen one round trip after it lands instead of up to two.
    const unsigned long long* const at = reinterpret_cast<const unsigned long long*>(slot + (polls ? h_of * 5 + f_of : g * 5));
    unsigned long long r0 = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long r1 = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long r2 = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long r3 = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int tries = 0;; ++tries) {
#define CCN_FPS_POLL(R_)                                                                   \
      if (!done && (uint32_t)(R_ >> 32) == tag) {                                          \
        got = (uint32_t)R_;                                                                \
        done = true;                                                                       \
      }                                                                                    \
      if (__ballot(!done) == 0ull) break;                                                  \
      R_ = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      CCN_FPS_POLL(r0)
      CCN_FPS_POLL(r1)
      CCN_FPS_POLL(r2)
      CCN_FPS_POLL(r3)
#undef CCN_FPS_POLL
      if ((tries & 255) == 255) {                        // a partner that never arrives: leave instead of hanging the queue
        uint32_t ab = 0;
        if (lane == 0) ab = fps_load_sc1(reinterpret_cast<const FpsGranule*>(abort_word)).x;
        ab = __builtin_amdgcn_readfirstlane(ab);
        if (tries >= (1 << 18)) {
          if (lane == 0) atomicExch(abort_word, 1u);
          ab = 1;
        }
        if (ab) return;
      }
    }
#pragma unroll
    for (int h = 0; h < FPS_CL_MAXG; ++h) {
      if (h >= G || h == g) continue;
      FpsBest o;
      o.v = __uint_as_float(__builtin_amdgcn_readlane(got, 5 * h));
      o.i = (int)__builtin_amdgcn_readlane(got, 5 * h + 1);
      o.x = __uint_as_float(__builtin_amdgcn_readlane(got, 5 * h + 2));
      o.y = __uint_as_float(__builtin_amdgcn_readlane(got, 5 * h + 3));
      o.z = __uint_as_float(__builtin_amdgcn_readlane(got, 5 * h + 4));
      fps_take(all, o);
    }
    cur = all.i;
    cx = all.x;
    cy = all.y;
    cz = all.z;
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

int ccn_knn_cloud_radius(const float* src, const int64_t* s_ptr, int64_t B, float scale, float* radius, void* stream) {
  CCN_REQUIRE(src && s_ptr && radius && B > 0 && B < 65536 && scale > 0.f, "knn_cloud_radius: bad arguments");
  hipLaunchKernelGGL(knn_cloud_radius_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, src, s_ptr, scale,
                     radius);
  CCN_LAUNCH_OK("knn_cloud_radius");
  return CCN_OK;
}

int ccn_knn_from_grid(const int64_t* idx, const float* q, const int64_t* q_ptr, const float* src, const int64_t* s_ptr,
                      int64_t B, int64_t P1, int64_t K, int64_t* nbr, float* weight, int32_t* flag, void* stream) {
  CCN_REQUIRE(idx && q && q_ptr && src && s_ptr && nbr && weight && flag && B > 0 && B < 65536 && P1 > 0 && K >= 1 &&
                  K <= KNN_MAXK,
              "knn_from_grid: bad arguments");
  hipLaunchKernelGGL(knn_from_grid_kernel, dim3(ccn_blocks(P1, 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, idx,
                     q, q_ptr, src, s_ptr, P1, (int)K, nbr, weight, flag);
  CCN_LAUNCH_OK("knn_from_grid");
  return CCN_OK;
}

int ccn_scatter_flagged(const int32_t* flag, const int32_t* offsets, int64_t n, int64_t* list, void* stream) {
  CCN_REQUIRE(flag && offsets && list && n >= 0, "scatter_flagged: bad arguments");
  if (n == 0) return CCN_OK;
  hipLaunchKernelGGL(scatter_flagged_kernel, dim3(ccn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, flag, offsets,
                     n, list);
  CCN_LAUNCH_OK("scatter_flagged");
  return CCN_OK;
}

int ccn_knn_points_list(const float* q, const int64_t* q_ptr, const float* src, const int64_t* s_ptr, int64_t B,
                        int64_t K, const int64_t* list, const int64_t* count, int64_t max_count, int64_t* nbr,
                        float* weight, void* stream) {
  CCN_REQUIRE(q && q_ptr && src && s_ptr && list && count && nbr && weight && B > 0 && B < 65536 && K >= 1 &&
                  K <= KNN_MAXK && max_count >= 0,
              "knn_points_list: bad arguments");
  if (max_count == 0) return CCN_OK;
  hipLaunchKernelGGL(knn_points_list_kernel, dim3(ccn_blocks(max_count, KNN_TPB / 64)), dim3(KNN_TPB), 0, (hipStream_t)stream,
                     q, q_ptr, src, s_ptr, (int)B, (int)K, list, count, nbr, weight);
  CCN_LAUNCH_OK("knn_points_list");
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
  hipLaunchKernelGGL(voxel_argmin_kernel, dim3(ccn_blocks(n, 256)), dim3(256), 0, s, score, voxel_of, n, num_voxels,
                     (unsigned long long*)scratch);
  hipLaunchKernelGGL(voxel_unpack_kernel, dim3(ccn_blocks(num_voxels, 256)), dim3(256), 0, s,
                     (const unsigned long long*)scratch, num_voxels, idx);
  CCN_LAUNCH_OK("voxel_argmin");
  return CCN_OK;
}

int ccn_fps_use_cluster(int on) {
  g_fps_cluster.store(on < 0 ? 0 : (on > 2 ? 2 : on), std::memory_order_relaxed);
  return CCN_OK;
}

int ccn_fps_set_lds_claim(int bytes) {
  g_fps_claim.store(bytes < 0 ? 0 : (bytes > FPS_CLAIM_MAX ? FPS_CLAIM_MAX : bytes), std::memory_order_relaxed);
  return CCN_OK;
}

int ccn_fps_debug_fault(int mode) {
  g_fps_fault.store(mode < 0 ? 0 : (mode > 2 ? 2 : mode), std::memory_order_relaxed);
  return CCN_OK;
}

size_t ccn_fps_workspace_bytes(int64_t n, int64_t B) {
  if (n < 0 || B < 0) return 0;
  return ccn_align256((size_t)n * 4) + (size_t)B * FPS_CL_AREA;
}

// Per device, once: the kernels' dynamic-LDS limits and the number of CUs (a cluster's members spin on one another, so a launch
// of clusters is held to half of them).  Returns the CU count, 0 on failure.
static int fps_device_setup() {
  constexpr int MAX_DEV = 64;
  static std::mutex mu;
  static int cus[MAX_DEV] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 0;
  std::lock_guard<std::mutex> lock(mu);
  if (cus[dev] == 0) {
    bool ok = hipFuncSetAttribute((const void*)fps_reg_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FPS_CLAIM_MAX) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)fps_reg_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, FPS_CLAIM_MAX) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)fps_reg_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, FPS_CLAIM_MAX) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)fps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FPS_CLAIM_MAX) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)fps_cluster_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, FPS_CLAIM_MAX) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)fps_hybrid_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   FPS_HYB_LDS * 4) == hipSuccess;
    int n = 0;
    ok = ok && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0;
    if (!ok) return 0;
    cus[dev] = n;
  }
  return cus[dev];
}

int ccn_fps(const float* pos, const int64_t* cloud_ptr, const int64_t* start, const int64_t* out_ptr, int64_t B,
            int64_t max_cloud, int64_t n, void* workspace, size_t workspace_bytes, int32_t* fallbacks, int64_t* out,
            void* stream) {
  CCN_REQUIRE(pos && cloud_ptr && start && out_ptr && workspace && out && B > 0 && n >= 0, "fps: bad arguments");
  CCN_REQUIRE(workspace_bytes >= ccn_fps_workspace_bytes(n, B), "fps: workspace of %zu bytes, %zu needed", workspace_bytes,
              ccn_fps_workspace_bytes(n, B));
  hipStream_t s = (hipStream_t)stream;
  float* const mind = reinterpret_cast<float*>(workspace);                                    // running minima of the streaming form: n floats
  char* const xch = reinterpret_cast<char*>(workspace) + ccn_align256((size_t)n * 4);         // the clusters' exchange areas: 512 B per cloud
  // One workgroup per cloud runs K dependent rounds: its speed is its latency.  It asks for 96 KB of (unused) dynamic LDS
  // so that no GEMM workgroup of the feature stream is co-scheduled on its CU -- next to two 4-wave GEMM workgroups a
  // 40 ms sampling pass of the A2D2 model took 60 % longer and the whole step followed it (the geometry of the next batch
  // is the critical path there); a handful of the 256 CUs is all it takes.
  const int cus = fps_device_setup();
  CCN_REQUIRE(cus > 0, "fps: cannot raise the dynamic LDS limit / read the device's CU count");
  const int claim = g_fps_claim.load(std::memory_order_relaxed), cluster = g_fps_cluster.load(std::memory_order_relaxed);
  const int64_t G = (max_cloud + 16 * FPS_TPB - 1) / (16 * FPS_TPB);
  if (max_cloud > 0 && max_cloud <= 4 * FPS_TPB)
    hipLaunchKernelGGL(fps_reg_kernel<4>, dim3((unsigned)B), dim3(FPS_TPB), (size_t)claim, s, pos, cloud_ptr, start, out_ptr, out);
  else if (max_cloud > 0 && max_cloud <= 8 * FPS_TPB)
    hipLaunchKernelGGL(fps_reg_kernel<8>, dim3((unsigned)B), dim3(FPS_TPB), (size_t)claim, s, pos, cloud_ptr, start, out_ptr, out);
  else if (max_cloud > 0 && max_cloud <= 16 * FPS_TPB)
    hipLaunchKernelGGL(fps_reg_kernel<16>, dim3((unsigned)B), dim3(FPS_TPB), (size_t)claim, s, pos, cloud_ptr, start, out_ptr, out);
  else if (cluster && claim > 0 && max_cloud > 16 * FPS_TPB && max_cloud <= FPS_CL_MAXG * 16 * FPS_TPB &&
           8 * G * ((B + 7) / 8) <= cus / 2) {   // (at most half of the CUs spin on one another)
    // more than 16 k points: G workgroups per cloud, every point in registers, one exchange per round (fps_cluster_kernel);
    // then the gated one-workgroup form for the clouds whose cluster gave up (none, normally: B workgroups that leave at once).
    CCN_HIP(hipMemsetAsync(xch, 0, (size_t)B * FPS_CL_AREA, s), "fps");
    hipLaunchKernelGGL(fps_cluster_kernel<16>, dim3((unsigned)(8 * G * ((B + 7) / 8))), dim3(FPS_TPB), (size_t)claim, s, pos,
                       cloud_ptr, start, out_ptr, B, (int)G, cluster == 2 ? 1 : 0, g_fps_fault.load(std::memory_order_relaxed), xch,
                       out);
    if (max_cloud <= 16 * FPS_TPB + FPS_HYB_LDS)
      hipLaunchKernelGGL(fps_hybrid_kernel<16>, dim3((unsigned)B), dim3(FPS_TPB), (size_t)FPS_HYB_LDS * 4, s, pos, cloud_ptr, start,
                         out_ptr, out, xch, fallbacks);
    else
      hipLaunchKernelGGL(fps_kernel, dim3((unsigned)B), dim3(FPS_TPB), (size_t)claim, s, pos, cloud_ptr, start, out_ptr, mind, out,
                         xch, fallbacks);
  } else if (max_cloud > 0 && max_cloud <= 16 * FPS_TPB + FPS_HYB_LDS && claim > 0)      // (claim 0 = A/B: the streaming form)
    hipLaunchKernelGGL(fps_hybrid_kernel<16>, dim3((unsigned)B), dim3(FPS_TPB), (size_t)FPS_HYB_LDS * 4, s, pos, cloud_ptr, start,
                       out_ptr, out, (const char*)nullptr, (int32_t*)nullptr);
  else
    hipLaunchKernelGGL(fps_kernel, dim3((unsigned)B), dim3(FPS_TPB), (size_t)claim, s, pos, cloud_ptr, start, out_ptr, mind, out,
                       (const char*)nullptr, (int32_t*)nullptr);
  CCN_LAUNCH_OK("fps");
  return CCN_OK;
}

}  // extern "C"

// Fixed-radius k-nearest-neighbour search on a hashed uniform grid:
// replaces third_party/FRNN (frnn.frnn_grid_points, called at src/models/utils/point_ops.py:459).
//
// Layout in HBM (all inside the caller-owned `grid` buffer, see ccn_frnn_grid_bytes):
//   cell_start  int32 [B*T + 1]   exclusive prefix of bucket populations, T = pow2 >= 2*P2 buckets per cloud
//   cell_fill   int32 [B*T]       bucket populations (build) / scatter cursors
//   sorted_pts  float4[B*P2]      (x, y, z, original index) grouped by bucket -> coalesced candidate reads
// A query reads the 27 buckets of the cells around it (pairwise different by construction of the hash, see bucket_of);
// points of other cells that hash into the same buckets are rejected by the distance test, so no per-point cell table
// is needed and no candidate is seen twice.
// The grid only prunes: a candidate is accepted iff d2 = fma(dz,dz,fma(dy,dy,dx*dx)) < r*r, and the K best
// are kept ordered by (d2, original index), exactly the exhaustive oracle (oracle/frnn_bruteforce.c).
// Cell edge = 1.001 r, so every point within r of a query lies in the 27 cells around the query's cell
// (cell coordinates are evaluated in double: valid for |coordinate| / r < 2e9, see cell_axis).
// Built with -ffp-contract=off.
#include "ccn_common.h"

#include <atomic>

namespace {

constexpr int BUILD_TPB = 256;
constexpr int QUERY_TPB = 128;

struct GridView {
  int32_t* cell_start;
  int32_t* cell_fill;
  float4* sorted_pts;
  void* scan_scratch;
  int64_t T;
};

__host__ int64_t table_size(int64_t P2) {
  int64_t t = 256;
  while (t < 2 * P2) t <<= 1;
  return t;
}

__host__ size_t grid_bytes(int64_t B, int64_t P2) {
  const int64_t T = table_size(P2);
  return ccn_align256((size_t)(B * T + 1) * 4) + ccn_align256((size_t)(B * T) * 4) +
         ccn_align256((size_t)(B * P2) * 16) + ccn_scan_scratch_bytes(B * T + 1) + 1024;
}

__host__ bool carve(void* grid, size_t bytes, int64_t B, int64_t P2, GridView* g) {
  CcnArena a(grid, bytes);
  g->T = table_size(P2);
  g->cell_start = a.take<int32_t>(B * g->T + 1);
  g->cell_fill = a.take<int32_t>(B * g->T);
  g->sorted_pts = a.take<float4>(B * P2);
  g->scan_scratch = a.take<char>(ccn_scan_scratch_bytes(B * g->T + 1));
  return a.ok();
}

// Cell coordinate in DOUBLE: x * inv_cell is then exact to ~1e-16 relative, so a point within r of a query always lies in
// one of the 27 cells around the query's cell (the 0.1 % margin of the cell edge dwarfs the rounding) for any coordinate
// whose cell index fits an int -- |coordinate| / r < 2e9 -- instead of the < 1e4 that fp32 cell arithmetic allowed
// (outside that range the fp32 form silently dropped neighbours).  Beyond the int range the index
// saturates: far-away points share a cell, which costs speed, never correctness (the grid only prunes).
__device__ __forceinline__ int cell_axis(float x, double inv_cell) {
  double c = floor((double)x * inv_cell);
  c = c < -2147483000.0 ? -2147483000.0 : (c > 2147483000.0 ? 2147483000.0 : c);
  return (int)c;
}
__device__ __forceinline__ int3 cell_of(float x, float y, float z, float inv_cell) {
  const double ic = (double)inv_cell;
  return make_int3(cell_axis(x, ic), cell_axis(y, ic), cell_axis(z, ic));
}

// Bucket of a cell: the low 6 bits are (cx mod 4, cy mod 4, cz mod 4), the rest a hash of the 4 x 4 x 4 block the cell lies
// in.  Three consecutive integers are distinct mod 4, so the 27 cells around any cell ALWAYS fall into 27 different
// buckets (T >= 256 keeps the low 6 bits): no bucket is visited twice by one query, and the cells of a block sit next to
// each other in the table (their range loads share cache lines).
__device__ __forceinline__ uint32_t bucket_of(int cx, int cy, int cz, uint32_t mask) {
  const uint32_t low = (uint32_t)(cx & 3) | ((uint32_t)(cy & 3) << 2) | ((uint32_t)(cz & 3) << 4);
  const uint32_t blk = ((uint32_t)(cx >> 2) * 73856093u) ^ ((uint32_t)(cy >> 2) * 19349663u) ^ ((uint32_t)(cz >> 2) * 83492791u);
  return ((blk << 6) | low) & mask;
}

__device__ __forceinline__ float inv_cell_of(float r) { return __frcp_rn(r * 1.001f); }

template <bool SCATTER>
__global__ __launch_bounds__(BUILD_TPB) void grid_insert_kernel(const float* __restrict__ pts,
                                                                const int64_t* __restrict__ lengths,
                                                                const float* __restrict__ radius, int64_t P2,
                                                                int64_t T, const int32_t* __restrict__ cell_start,
                                                                int32_t* __restrict__ cell_fill,
                                                                float4* __restrict__ sorted_pts) {
  const int64_t b = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= lengths[b] || j >= P2) return;     // (a length beyond the padded row: never read past it)
  const float* p = pts + (b * P2 + j) * 3;
  const float x = p[0], y = p[1], z = p[2];
  const int3 c = cell_of(x, y, z, inv_cell_of(radius[b]));
  const int64_t slot = b * T + bucket_of(c.x, c.y, c.z, (uint32_t)(T - 1));
  if (!SCATTER) {
    atomicAdd(&cell_fill[slot], 1);
  } else {
    // (the counters of the first pass are counted DOWN: every point of a bucket gets a distinct slot cnt-1 .. 0 without a
    // second clearing of the table between the passes; the order inside a bucket is arbitrary either way)
    const int32_t at = cell_start[slot] + atomicSub(&cell_fill[slot], 1) - 1;
    sorted_pts[at] = make_float4(x, y, z, __int_as_float((int)j));
  }
}

// one thread per query; its running top-K lives in LDS as column `threadIdx.x` of [K][QUERY_TPB] arrays,
// so every access is bank-conflict free whatever slot each lane is working on.
__global__ __launch_bounds__(QUERY_TPB) void grid_query_kernel(
    const float* __restrict__ q_pts, const int64_t* __restrict__ lengths1, const float* __restrict__ radius,
    int64_t P1, int K, int64_t T, const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted_pts,
    int64_t* __restrict__ idx_out, float* __restrict__ dist_out,
    int32_t* __restrict__ count_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* best_d = (float*)lds_raw;                       // [K][QUERY_TPB]
  int* best_i = (int*)(lds_raw + (size_t)K * QUERY_TPB * 4);  // [K][QUERY_TPB]
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + tid;
  int have = 0;
  if (i < P1 && i < lengths1[b]) {
    const float* q = q_pts + (b * P1 + i) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float r = radius[b];
    const float r2 = r * r;
    const int3 cq = cell_of(qx, qy, qz, inv_cell_of(r));
    const uint32_t mask = (uint32_t)(T - 1);
    const int32_t* starts = cell_start + b * T;
    // Phase 1: the 27 bucket ranges, all loads in flight together.  Every point within r of the query lies in one of
    // the 27 cells, hence in one of these (distinct) buckets; points of colliding far-away cells fail the distance
    // test -- so the exact-cell table need not be read and the accepted set is exactly {d2 < r*r}.
    int32_t lo[27], hi[27];
    uint32_t hs[27];
#pragma unroll
    for (int c = 0; c < 27; ++c) {
      const int dx = c % 3 - 1, dy = (c / 3) % 3 - 1, dz = c / 9 - 1;
      hs[c] = bucket_of(cq.x + dx, cq.y + dy, cq.z + dz, mask);
      lo[c] = starts[hs[c]];
      hi[c] = starts[hs[c] + 1];
    }
    // Phase 2: candidates, two point records in flight
#pragma unroll 1
    for (int c = 0; c < 27; ++c) {
      for (int32_t s = lo[c]; s < hi[c]; s += 2) {
        const float4 p0 = sorted_pts[s];
        const bool two = s + 1 < hi[c];
        const float4 p1 = two ? sorted_pts[s + 1] : p0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !two) break;
          const float4 p = u == 0 ? p0 : p1;
          const float d2 = ccn_sqdist3(p.x - qx, p.y - qy, p.z - qz);
          if (!(d2 < r2)) continue;
          const int j = __float_as_int(p.w);
          if (have == K) {
            const float wd = best_d[(K - 1) * QUERY_TPB + tid];
            if (!(d2 < wd || (d2 == wd && j < best_i[(K - 1) * QUERY_TPB + tid]))) continue;
          }
          int slot = have < K ? have : K - 1;
          while (slot > 0) {
            const float pd = best_d[(slot - 1) * QUERY_TPB + tid];
            const int pi = best_i[(slot - 1) * QUERY_TPB + tid];
            if (!(pd > d2 || (pd == d2 && pi > j))) break;
            best_d[slot * QUERY_TPB + tid] = pd;
            best_i[slot * QUERY_TPB + tid] = pi;
            --slot;
          }
          best_d[slot * QUERY_TPB + tid] = d2;
          best_i[slot * QUERY_TPB + tid] = j;
          if (have < K) ++have;
        }
      }
    }
  }
  // The block's results form one contiguous (queries x K) span of the outputs: written cooperatively so that
  // consecutive lanes store consecutive elements (a row per thread would scatter 8-byte stores K*8 bytes apart).
  __shared__ int have_s[QUERY_TPB];
  have_s[tid] = have;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x;
  const int64_t nq = P1 - i0 < QUERY_TPB ? P1 - i0 : QUERY_TPB;
  int64_t* out_i = idx_out + (b * P1 + i0) * K;
  float* out_d = dist_out ? dist_out + (b * P1 + i0) * K : nullptr;
  for (int64_t e = tid; e < nq * K; e += QUERY_TPB) {
    const int qq = (int)(e / K), sl = (int)(e - (int64_t)qq * K);
    const bool ok = sl < have_s[qq];
    out_i[e] = ok ? (int64_t)best_i[sl * QUERY_TPB + qq] : -1;
    if (out_d) out_d[e] = ok ? best_d[sl * QUERY_TPB + qq] : -1.0f;
  }
  if (count_out && i < P1) count_out[b * P1 + i] = have;
}

// ------------------------------------------------------------------ team form: TEAM lanes per query
// The thread-per-query kernel above walks every query's candidates serially in one lane: with few queries and many
// candidates per query (the coarse levels: 35 k queries at r = 0.3 see ~200 candidates each and keep ~30) the chip
// holds two waves per CU and each lane spends its time shifting an insertion-sorted list in LDS.  Here a TEAM of 32 or
// 64 lanes serves ONE query:
//   * lanes 0..26 hash one neighbour cell each and fetch its bucket range (one load instruction for all 27 ranges)
//   * the non-empty buckets are visited one after the other, their points read TEAM at a time (coalesced float4 records);
//     accepted candidates (d2 < r*r) are appended to the team's list in LDS as 64-bit keys (d2 bits << 32 | index:
//     unsigned order of the keys == (d2, index) order, d2 >= +0) at positions given by a ballot / prefix count
//   * ranking instead of sorting: lane l counts the keys smaller than its own (every key is read from LDS as a
//     broadcast), which IS its output position; ranks >= K are dropped.  A full list is pruned to its K best the same way.
// Output identical to the thread form (same acceptance test, same (d2, index) order).
constexpr int TEAM_TPB = 256;
constexpr int TEAM_CAP = 128;      // accepted keys held per team before a prune (K <= 128 - 64)
constexpr int TEAM_SLOTS = 192;    // candidates a team handles in flat form (more: bucket by bucket)

template <int TEAM>
__global__ __launch_bounds__(TEAM_TPB) void grid_query_team_kernel(
    const float* __restrict__ q_pts, const int64_t* __restrict__ lengths1, const float* __restrict__ radius, int64_t P1,
    int K, int64_t T, const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted_pts,
    int64_t* __restrict__ idx_out, float* __restrict__ dist_out, int32_t* __restrict__ count_out) {
  constexpr int TEAMS = TEAM_TPB / TEAM;
  __shared__ unsigned long long keys[TEAMS][TEAM_CAP];
  __shared__ int slots[TEAMS][TEAM_SLOTS];
  const int tl = threadIdx.x % TEAM;                 // lane inside the team
  const int team = threadIdx.x / TEAM;
  const int half = (threadIdx.x & 63) / TEAM;        // team inside the wave (TEAM = 32: 0 / 1)
  const int64_t b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * TEAMS + team;
  if (i >= P1) return;                               // (whole teams leave together)
  unsigned long long* mine = keys[team];
  const bool live = i < lengths1[b];
  int A = 0;                                         // accepted keys in the list (team-uniform)
  int total = 0;                                     // accepted candidates seen (for count_out)
  // rank the list's first n keys; keep the K smallest, in order, at the front.  Returns the new length.
  auto prune = [&](int n) -> int {
    unsigned long long kk[TEAM_CAP / TEAM];           // this lane's share of the list (2 or 4 keys)
    int rk[TEAM_CAP / TEAM];
#pragma unroll
    for (int u = 0; u < TEAM_CAP / TEAM; ++u) {
      kk[u] = tl + u * TEAM < n ? mine[tl + u * TEAM] : ~0ull;
      rk[u] = 0;
    }
    for (int j = 0; j < n; ++j) {
      const unsigned long long kj = mine[j];          // same address for the whole team: LDS broadcast
#pragma unroll
      for (int u = 0; u < TEAM_CAP / TEAM; ++u) rk[u] += kj < kk[u];
    }
    __builtin_amdgcn_wave_barrier();                  // every lane has read the old list
#pragma unroll
    for (int u = 0; u < TEAM_CAP / TEAM; ++u)
      if (tl + u * TEAM < n && rk[u] < K) mine[rk[u]] = kk[u];
    __builtin_amdgcn_wave_barrier();
    return n < K ? n : K;
  };

  if (live) {
    const float* q = q_pts + (b * P1 + i) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float r = radius[b];
    const float r2 = r * r;
    const int3 cq = cell_of(qx, qy, qz, inv_cell_of(r));
    const uint32_t mask = (uint32_t)(T - 1);
    const int32_t* starts = cell_start + b * T;
    // ---- the 27 bucket ranges: one cell per lane
    uint32_t h = 0xffffffffu;
    int32_t lo = 0, cnt = 0;
    if (tl < 27) {
      const int dx = tl % 3 - 1, dy = (tl / 3) % 3 - 1, dz = tl / 9 - 1;
      h = bucket_of(cq.x + dx, cq.y + dy, cq.z + dz, mask);
      lo = starts[h];
      cnt = starts[h + 1] - lo;
    }
    // (the 27 buckets are pairwise different by construction of bucket_of: nothing is visited twice)
    // ---- flat form (few candidates, the common case at the fine levels): the candidates of all 27 buckets are numbered
    // 0 .. C-1 (prefix sum of the populations across the team), every cell-lane writes the sorted-array position of its
    // candidates into the team's slot table, and the team then reads the candidates TEAM at a time -- one dependent memory
    // round trip for all of them instead of one per non-empty bucket (8 per query at r = 0.04).
    int pre = cnt;                                   // inclusive prefix over the team
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int v = __shfl_up(pre, d, TEAM);
      if (tl >= d) pre += v;
    }
    const int C = __shfl(pre, 26, TEAM);             // lanes >= 27 hold cnt = 0: lane 26 has the total
    pre -= cnt;
    if (C <= TEAM_SLOTS) {
      int* slot = slots[team];
      for (int u = 0; u < cnt; ++u) slot[pre + u] = lo + u;
      __builtin_amdgcn_wave_barrier();
      for (int t0 = 0; t0 < C; t0 += 2 * TEAM) {
        const int ta = t0 + tl, tb = t0 + TEAM + tl;
        const bool ina = ta < C, inb = tb < C;
        const float4 pa = sorted_pts[slot[ina ? ta : 0]];
        const float4 pb = sorted_pts[slot[inb ? tb : 0]];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float4 p = u == 0 ? pa : pb;
          const bool in = u == 0 ? ina : inb;
          const float d2 = ccn_sqdist3(p.x - qx, p.y - qy, p.z - qz);
          const bool ok = in && d2 < r2;
          const unsigned long long bal = __ballot(ok);
          const unsigned long long mybits = TEAM == 64 ? bal : ((bal >> (32 * half)) & 0xffffffffull);
          const int before = __popcll(mybits & ((1ull << tl) - 1ull));
          if (ok) mine[A + before] = ((unsigned long long)__float_as_uint(d2) << 32) | (uint32_t)__float_as_int(p.w);
          const int add = __popcll(mybits);
          A += add;
          total += add;
          __builtin_amdgcn_wave_barrier();
          if (A > TEAM_CAP - TEAM) A = prune(A);
        }
      }
      cnt = 0;                                       // nothing left for the bucket loop below
    }
    unsigned long long todo = __ballot(cnt > 0);
    uint32_t work = TEAM == 64 ? 0u : (uint32_t)(todo >> (32 * half));    // 32-lane teams: this team's half of the ballot
    unsigned long long work64 = todo;
    while (TEAM == 64 ? work64 != 0ull : work != 0u) {
      int c;
      if (TEAM == 64) {
        c = __builtin_ctzll(work64);
        work64 &= work64 - 1;
      } else {
        c = __builtin_ctz(work);
        work &= work - 1;
      }
      const int32_t lo_c = __shfl(lo, c, TEAM), cnt_c = __shfl(cnt, c, TEAM);
      for (int32_t off = 0; off < cnt_c; off += TEAM) {
        const int32_t t = off + tl;
        const bool in = t < cnt_c;
        const float4 p = sorted_pts[lo_c + (in ? t : 0)];
        const float d2 = ccn_sqdist3(p.x - qx, p.y - qy, p.z - qz);
        const bool ok = in && d2 < r2;
        const unsigned long long bal = __ballot(ok);
        const unsigned long long mybits = TEAM == 64 ? bal : ((bal >> (32 * half)) & 0xffffffffull);
        const int before = __popcll(mybits & ((1ull << tl) - 1ull));
        if (ok) mine[A + before] = ((unsigned long long)__float_as_uint(d2) << 32) | (uint32_t)__float_as_int(p.w);
        const int add = __popcll(mybits);
        A += add;
        total += add;
        __builtin_amdgcn_wave_barrier();
        if (A > TEAM_CAP - TEAM) A = prune(A);          // room for one more round of TEAM candidates
      }
    }
    A = prune(A);                                       // final order
  }
  // ---- output row: K entries, -1 padded
  int64_t* out_i = idx_out + (b * P1 + i) * K;
  float* out_d = dist_out ? dist_out + (b * P1 + i) * K : nullptr;
  for (int sl = tl; sl < K; sl += TEAM) {
    const bool ok = sl < A;
    const unsigned long long key = ok ? mine[sl] : 0ull;
    out_i[sl] = ok ? (int64_t)(uint32_t)(key & 0xffffffffull) : -1;
    if (out_d) out_d[sl] = ok ? __uint_as_float((uint32_t)(key >> 32)) : -1.0f;
  }
  if (count_out && tl == 0) count_out[b * P1 + i] = total < K ? total : K;
}

static std::atomic<int> g_query_mode{0};   // A/B hook (ccn_frnn_query_mode): 0 auto, 1 thread per query, 2 / 3 teams of 32 / 64 lanes

// ------------------------------------------------------------------ dense idx -> CSR edge list
__global__ void dense_count_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ cloud_ptr1, int64_t P1,
                                   int64_t K, int32_t* __restrict__ counts) {
  const int64_t b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t len = cloud_ptr1[b + 1] - cloud_ptr1[b];
  if (i >= len || i >= P1) return;
  const int64_t* row = idx + (b * P1 + i) * K;
  int c = 0;
  for (int64_t s = 0; s < K; ++s) c += row[s] != -1;
  counts[cloud_ptr1[b] + i] = c;
}

__global__ void dense_fill_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ cloud_ptr1,
                                  const int64_t* __restrict__ cloud_ptr2, int64_t P1, int64_t K,
                                  const int32_t* __restrict__ offsets, int64_t* __restrict__ row_out,
                                  int64_t* __restrict__ col_out) {
  const int64_t b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t len = cloud_ptr1[b + 1] - cloud_ptr1[b];
  if (i >= len || i >= P1) return;
  const int64_t* row = idx + (b * P1 + i) * K;
  const int64_t q = cloud_ptr1[b] + i;
  int32_t at = offsets[q];
  const int32_t end = offsets[q + 1];       // (== at + this query's count; smaller only when the caller clamped the offsets to
  const int64_t base2 = cloud_ptr2[b];      //  the capacity of row_out / col_out: bounded counts, capacity exceeded)
  for (int64_t s = 0; s < K; ++s) {
    const int64_t j = row[s];
    if (j != -1 && at < end) {
      row_out[at] = q;
      col_out[at] = base2 + j;
      ++at;
    }
  }
}

}  // namespace

extern "C" {

size_t ccn_frnn_grid_bytes(int64_t B, int64_t P2) { return grid_bytes(B < 1 ? 1 : B, P2 < 1 ? 1 : P2); }

int ccn_frnn_grid_build(const float* points2, const int64_t* lengths2, const float* r, int64_t B, int64_t P2,
                        void* grid, size_t bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(points2 && lengths2 && r && grid, "frnn_grid_build: null pointer");
  CCN_REQUIRE(B > 0 && P2 > 0 && B < 65536, "frnn_grid_build: bad batch/points (B=%lld, P2=%lld)", (long long)B,
              (long long)P2);
  CCN_REQUIRE(B * P2 < 2147483647LL, "frnn_grid_build: B*P2 exceeds int32");
  CCN_REQUIRE(bytes >= grid_bytes(B, P2), "frnn_grid_build: grid buffer too small");
  GridView g;
  CCN_REQUIRE(carve(grid, bytes, B, P2, &g), "frnn_grid_build: carve failed");
  const int64_t cells = B * g.T;
  CCN_HIP(hipMemsetAsync(g.cell_fill, 0, (size_t)cells * 4, s), "frnn_grid_build");
  dim3 gridDim_(ccn_blocks(P2, BUILD_TPB), (unsigned)B);
  hipLaunchKernelGGL(grid_insert_kernel<false>, gridDim_, dim3(BUILD_TPB), 0, s, points2, lengths2, r, P2, g.T,
                     (const int32_t*)nullptr, g.cell_fill, (float4*)nullptr);
  int rc = ccn_scan_i32(g.cell_fill, g.cell_start, cells, false, g.cell_start + cells, g.scan_scratch, s);
  if (rc) return rc;
  hipLaunchKernelGGL(grid_insert_kernel<true>, gridDim_, dim3(BUILD_TPB), 0, s, points2, lengths2, r, P2, g.T,
                     g.cell_start, g.cell_fill, g.sorted_pts);
  CCN_LAUNCH_OK("frnn_grid_build");
  return CCN_OK;
}

int ccn_frnn_query(const float* points1, const int64_t* lengths1, const float* r, int64_t B, int64_t P1, int64_t K,
                   const void* grid, int64_t P2, int64_t* idx, float* dist2, int32_t* count, void* stream) {
  CCN_REQUIRE(points1 && lengths1 && r && grid && idx, "frnn_query: null pointer");
  CCN_REQUIRE(B > 0 && P1 > 0 && P2 > 0 && B < 65536, "frnn_query: bad sizes");
  CCN_REQUIRE(K >= 1 && K <= 128, "frnn_query: K must be in [1, 128] (got %lld)", (long long)K);
  GridView g;
  CCN_REQUIRE(carve(const_cast<void*>(grid), grid_bytes(B, P2), B, P2, &g), "frnn_query: carve failed");
  // teams need K <= TEAM_CAP - 64 list slots free for a round of candidates
  int mode = g_query_mode;
  // measured (tools/bench_frnn.py, profiles/archive/r02_frnn_microbench.txt): 32-lane teams win at every level of the KITTI
  // model -- 2.5x on the full 8 x 50 k cloud, 4-10x on the coarse levels -- over both the thread form and 64-lane teams
  if (mode == 0) mode = 2;
  if (K > TEAM_CAP - 64) mode = 1;
  if (mode == 1) {
    const size_t lds = (size_t)K * QUERY_TPB * 8;
    dim3 gridDim_(ccn_blocks(P1, QUERY_TPB), (unsigned)B);
    hipLaunchKernelGGL(grid_query_kernel, gridDim_, dim3(QUERY_TPB), lds, (hipStream_t)stream, points1, lengths1, r, P1,
                       (int)K, g.T, g.cell_start, g.sorted_pts, idx, dist2, count);
  } else if (mode == 2) {
    dim3 gridDim_(ccn_blocks(P1, TEAM_TPB / 32), (unsigned)B);
    hipLaunchKernelGGL(grid_query_team_kernel<32>, gridDim_, dim3(TEAM_TPB), 0, (hipStream_t)stream, points1, lengths1, r,
                       P1, (int)K, g.T, g.cell_start, g.sorted_pts, idx, dist2, count);
  } else {
    dim3 gridDim_(ccn_blocks(P1, TEAM_TPB / 64), (unsigned)B);
    hipLaunchKernelGGL(grid_query_team_kernel<64>, gridDim_, dim3(TEAM_TPB), 0, (hipStream_t)stream, points1, lengths1, r,
                       P1, (int)K, g.T, g.cell_start, g.sorted_pts, idx, dist2, count);
  }
  CCN_LAUNCH_OK("frnn_query");
  return CCN_OK;
}

int ccn_frnn_query_mode(int mode) {
  g_query_mode = (mode >= 0 && mode <= 3) ? mode : 0;
  return CCN_OK;
}

int ccn_dense_to_csr_count(const int64_t* idx, const int64_t* cloud_ptr1, int64_t B, int64_t P1, int64_t K,
                           int32_t* counts, void* stream) {
  CCN_REQUIRE(idx && cloud_ptr1 && counts && B > 0 && P1 > 0 && K > 0 && B < 65536, "dense_to_csr_count: bad arguments");
  dim3 gridDim_(ccn_blocks(P1, 256), (unsigned)B);
  hipLaunchKernelGGL(dense_count_kernel, gridDim_, dim3(256), 0, (hipStream_t)stream, idx, cloud_ptr1, P1, K, counts);
  CCN_LAUNCH_OK("dense_to_csr_count");
  return CCN_OK;
}

int ccn_dense_to_csr_fill(const int64_t* idx, const int64_t* cloud_ptr1, const int64_t* cloud_ptr2, int64_t B,
                          int64_t P1, int64_t K, const int32_t* offsets, int64_t* row, int64_t* col, void* stream) {
  CCN_REQUIRE(idx && cloud_ptr1 && cloud_ptr2 && offsets && row && col && B > 0 && P1 > 0 && K > 0 && B < 65536,
              "dense_to_csr_fill: bad arguments");
  dim3 gridDim_(ccn_blocks(P1, 256), (unsigned)B);
  hipLaunchKernelGGL(dense_fill_kernel, gridDim_, dim3(256), 0, (hipStream_t)stream, idx, cloud_ptr1, cloud_ptr2, P1,
                     K, offsets, row, col);
  CCN_LAUNCH_OK("dense_to_csr_fill");
  return CCN_OK;
}

}  // extern "C"

// Dense ranks of 64-bit keys: rank[i] = number of DISTINCT keys smaller than key[i]  (= the `return_inverse` of
// torch.unique(sorted=True), which the reference's VoxelFPS takes over its (cloud, voxel) rows: fps_ops.py:51-60; the sorted
// order of the unique keys IS the order of its output).  Replaces the torch.unique / rocprim merge sort the voxel levels
// still went through in round 2.
//
// LSD radix sort of (key, original index) pairs, 8-bit digits, one wave per 1024-key tile:
//   * ccn_key_spread: OR over all keys of (key XOR key[0]) -- the host runs a pass only for digits in which the keys differ
//     at all (voxel keys of a scene at one voxel size differ in 4-5 of the 8 digits)
//   * per pass: tile histograms (LDS atomics) -> exclusive scan over [digit][tile] (ccn_scan_i32) -> stable scatter: a wave
//     walks its tile 64 keys at a time, a lane's rank among the lanes holding the same digit comes from eight ballots
//     (match mask) and a popcount, the digit's running base lives in LDS -- no sorting network, no atomics on the output
//   * ranks: flags key[i] != key[i-1] over the sorted keys -> inclusive scan -> scattered back through the index payload.
// Stable, deterministic, HBM traffic 2 x 12 bytes per key and pass.
#include "ccn_common.h"

namespace {

constexpr int RS_TILE = 1024;   // keys per workgroup (one wave)
constexpr int RS_WAVE = 64;

__global__ __launch_bounds__(256) void key_spread_kernel(const uint64_t* __restrict__ key, int64_t n,
                                                         unsigned long long* __restrict__ spread) {
  const uint64_t k0 = key[0];
  uint64_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc |= key[i] ^ k0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc |= __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicOr(spread, (unsigned long long)acc);
}

__global__ __launch_bounds__(RS_WAVE) void radix_hist_kernel(const uint64_t* __restrict__ key, int64_t n, int shift,
                                                             int64_t tiles, int32_t* __restrict__ hist) {
  __shared__ int32_t bins[256];
  const int lane = threadIdx.x;
  for (int b = lane; b < 256; b += RS_WAVE) bins[b] = 0;
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_TILE / RS_WAVE; ++r) {
    const int64_t i = t0 + r * RS_WAVE + lane;
    if (i < n) atomicAdd(&bins[(int)((key[i] >> shift) & 255)], 1);
  }
  __syncthreads();
  for (int b = lane; b < 256; b += RS_WAVE) hist[(int64_t)b * tiles + blockIdx.x] = bins[b];   // [digit][tile]
}

template <bool FIRST>
__global__ __launch_bounds__(RS_WAVE) void radix_scatter_kernel(const uint64_t* __restrict__ key_in,
                                                                const int32_t* __restrict__ val_in, int64_t n, int shift,
                                                                int64_t tiles, const int32_t* __restrict__ offs,
                                                                uint64_t* __restrict__ key_out, int32_t* __restrict__ val_out) {
  __shared__ int32_t base[256];
  const int lane = threadIdx.x;
  for (int b = lane; b < 256; b += RS_WAVE) base[b] = offs[(int64_t)b * tiles + blockIdx.x];
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * RS_TILE;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < RS_TILE / RS_WAVE; ++r) {
    const int64_t i = t0 + r * RS_WAVE + lane;
    const bool live = i < n;
    const uint64_t k = live ? key_in[i] : 0;
    const int32_t v = live ? (FIRST ? (int32_t)i : val_in[i]) : 0;
    const int d = (int)((k >> shift) & 255);
    uint64_t same = __ballot(live);                       // lanes holding the same digit
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t m = __ballot((d >> b) & 1);
      same &= ((d >> b) & 1) ? m : ~m;
    }
    const int before = __popcll(same & lt);
    int32_t dst = 0;
    if (live) dst = base[d] + before;
    __syncthreads();                                      // everyone has read the digit bases of this round
    if (live && before == 0) base[d] += __popcll(same);   // the digit group's first lane advances its base
    __syncthreads();
    if (live) {
      key_out[dst] = k;
      val_out[dst] = v;
    }
  }
}

__global__ __launch_bounds__(256) void key_flags_kernel(const uint64_t* __restrict__ key, int64_t n, int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i > 0 && key[i] != key[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void rank_scatter_kernel(const int32_t* __restrict__ incl, const int32_t* __restrict__ val,
                                                           int64_t n, int64_t* __restrict__ rank, int64_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    rank[val[i]] = incl[i];
    if (i == n - 1) count[0] = (int64_t)incl[i] + 1;
  }
}

inline int64_t rs_tiles(int64_t n) { return (n + RS_TILE - 1) / RS_TILE; }

}  // namespace

namespace {
// key = the source number (M for a row that names none: behind every list); counts = rows per source
template <typename IT>
__global__ void inv_pack_kernel(const IT* __restrict__ src, int64_t n, int64_t M, int64_t* __restrict__ key,
                                int32_t* __restrict__ counts) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t m = (int64_t)src[r];
  const bool ok = m >= 0 && m < M;
  key[r] = ok ? m : M;
  if (ok) atomicAdd(&counts[m], 1);
}
}  // namespace

extern "C" {

// spread: one device uint64 (zeroed here): OR over (key[i] ^ key[0])
int ccn_key_spread(const int64_t* key, int64_t n, int64_t* spread, void* stream) {
  CCN_REQUIRE(key && spread && n >= 0, "key_spread: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  CCN_HIP(hipMemsetAsync(spread, 0, 8, s), "key_spread");
  if (n == 0) return CCN_OK;
  const int blocks = (int)((n + 4095) / 4096 < 1024 ? (n + 4095) / 4096 : 1024);
  hipLaunchKernelGGL(key_spread_kernel, dim3(blocks), dim3(256), 0, s, (const uint64_t*)key, n, (unsigned long long*)spread);
  CCN_LAUNCH_OK("key_spread");
  return CCN_OK;
}

size_t ccn_rank_keys_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  const int64_t tiles = rs_tiles(n);
  return ccn_align256((size_t)n * 8) * 2 + ccn_align256((size_t)n * 4) * 3 + ccn_align256((size_t)256 * tiles * 4) * 2 +
         ccn_align256(ccn_scan_scratch_bytes(256 * tiles > n ? 256 * tiles : n)) + 1024;
}

// the LSD passes: on return *kout / *vout point at the sorted keys and their original indices (inside the workspace)
static int rs_sort(const int64_t* key, int64_t n, int digit_mask, void* workspace, size_t workspace_bytes, hipStream_t s,
                   const uint64_t** kout, const int32_t** vout, int32_t** flag_out, int32_t** idle_out, void** scan_out) {
  const int64_t tiles = rs_tiles(n);
  CcnArena ar(workspace, workspace_bytes);
  uint64_t* kbuf[2] = {ar.take<uint64_t>(n), ar.take<uint64_t>(n)};
  int32_t* vbuf[2] = {ar.take<int32_t>(n), ar.take<int32_t>(n)};
  int32_t* flag = ar.take<int32_t>(n);
  int32_t* hist = ar.take<int32_t>(256 * tiles);
  int32_t* offs = ar.take<int32_t>(256 * tiles);
  void* scan_ws = ar.take<char>(ccn_scan_scratch_bytes(256 * tiles > n ? 256 * tiles : n));
  CCN_REQUIRE(ar.ok(), "rank_keys / sort_keys: workspace too small (%zu bytes given, %zu needed)", workspace_bytes,
              ccn_rank_keys_workspace_bytes(n));
  const uint64_t* kin = (const uint64_t*)key;
  const int32_t* vin = nullptr;
  int cur = 0;
  bool first = true;
  if ((digit_mask & 0xff) == 0) digit_mask = 1;   // all keys equal: one pass on digit 0 gives the identity payload
  for (int b = 0; b < 8; ++b) {
    if (!((digit_mask >> b) & 1)) continue;
    const int shift = 8 * b;
    hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)tiles), dim3(RS_WAVE), 0, s, kin, n, shift, tiles, hist);
    int rc = ccn_scan_i32(hist, offs, 256 * tiles, false, nullptr, scan_ws, s);
    if (rc) return rc;
    if (first)
      hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3((unsigned)tiles), dim3(RS_WAVE), 0, s, kin, vin, n, shift, tiles, offs,
                         kbuf[cur], vbuf[cur]);
    else
      hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3((unsigned)tiles), dim3(RS_WAVE), 0, s, kin, vin, n, shift, tiles, offs,
                         kbuf[cur], vbuf[cur]);
    kin = kbuf[cur];
    vin = vbuf[cur];
    cur ^= 1;
    first = false;
  }
  *kout = kin;
  *vout = vin;
  *flag_out = flag;
  *idle_out = (vin == vbuf[0]) ? vbuf[1] : vbuf[0];      // (the idle half of the payload ping-pong)
  *scan_out = scan_ws;
  return CCN_OK;
}

// rank[i] = dense rank of key[i] (non-negative int64) among the distinct keys; count[0] = number of distinct keys.
// digit_mask: bit b set = sort on the 8-bit digit b (0 = least significant); pass the digits in which the keys differ
// (ccn_key_spread), or 0xff for all eight.
int ccn_rank_keys(const int64_t* key, int64_t n, int digit_mask, int64_t* rank, int64_t* count, void* workspace,
                  size_t workspace_bytes, void* stream) {
  CCN_REQUIRE(key && rank && count && n >= 0 && n < ((int64_t)1 << 31), "rank_keys: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    CCN_HIP(hipMemsetAsync(count, 0, 8, s), "rank_keys");
    return CCN_OK;
  }
  const uint64_t* kin;
  const int32_t* vin;
  int32_t *flag, *incl;
  void* scan_ws;
  int rc = rs_sort(key, n, digit_mask, workspace, workspace_bytes, s, &kin, &vin, &flag, &incl, &scan_ws);
  if (rc) return rc;
  hipLaunchKernelGGL(key_flags_kernel, dim3(ccn_blocks(n, 256)), dim3(256), 0, s, kin, n, flag);
  rc = ccn_scan_i32(flag, incl, n, true, nullptr, scan_ws, s);
  if (rc) return rc;
  hipLaunchKernelGGL(rank_scatter_kernel, dim3(ccn_blocks(n, 256)), dim3(256), 0, s, incl, vin, n, rank, count);
  CCN_LAUNCH_OK("rank_keys");
  return CCN_OK;
}

// sorted[] = the keys in ascending order (the final torch.sort of the reference's farthest-point indices, point_ops.py:57-70;
// same workspace as ccn_rank_keys)
int ccn_sort_keys(const int64_t* key, int64_t n, int digit_mask, int64_t* sorted, void* workspace, size_t workspace_bytes,
                  void* stream) {
  CCN_REQUIRE(key && sorted && n >= 0 && n < ((int64_t)1 << 31), "sort_keys: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return CCN_OK;
  const uint64_t* kin;
  const int32_t* vin;
  int32_t *flag, *incl;
  void* scan_ws;
  int rc = rs_sort(key, n, digit_mask, workspace, workspace_bytes, s, &kin, &vin, &flag, &incl, &scan_ws);
  if (rc) return rc;
  CCN_HIP(hipMemcpyAsync(sorted, kin, (size_t)n * 8, hipMemcpyDeviceToDevice, s), "sort_keys");
  return CCN_OK;
}


// ---- inverse of a flat index list (round 5): rows r = 0 .. n-1 name a source src[r] in [0, M); for every source the rows that
// name it, ascending (what the atomics-free backward of the first edge layers gathers through: ccn_cg_edge_bwd_gather,
// ccn_pn_edge_bwd_gather).  The stable LSD sort above over the source numbers, payload = row number: rows of one source keep
// their ascending order, whatever the list lengths (a first form -- fill at an atomic cursor, insertion sort per list -- took
// minutes on five lists of 60 000 rows).  Replaces torch.sort + bincount + cumsum on the geometry stream (rocprim merge-sort
// and look-back-scan launches inside the step).  Rows naming no source (src < 0 or >= M) sort behind the last list.
size_t ccn_inverse_lists_workspace_bytes(int64_t n, int64_t M) {
  return ccn_align256((size_t)(n > 0 ? n : 1) * 8) + ccn_align256((size_t)(M + 1) * 4) + ccn_align256(ccn_scan_scratch_bytes(M + 1)) +
         ccn_rank_keys_workspace_bytes(n) + 1024;
}

int ccn_inverse_lists(const void* src, int src_is_i64, int64_t n, int64_t M, int32_t* inv_ptr, int32_t* inv_row, void* ws,
                      size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  CCN_REQUIRE(src && inv_ptr && inv_row && n >= 0 && M > 0, "inverse_lists: bad arguments");
  CCN_REQUIRE(n < (int64_t)1 << 31 && M < (int64_t)1 << 31, "inverse_lists: more than 2^31 entries");
  CCN_REQUIRE(ws && ws_bytes >= ccn_inverse_lists_workspace_bytes(n, M), "inverse_lists: workspace too small");
  CcnArena a(ws, ws_bytes);
  int64_t* key = a.take<int64_t>(n > 0 ? n : 1);
  int32_t* counts = a.take<int32_t>(M + 1);
  void* scratch = a.take<char>(ccn_scan_scratch_bytes(M + 1));
  const size_t sort_bytes = ccn_rank_keys_workspace_bytes(n);
  void* sort_ws = a.take<char>(sort_bytes);
  CCN_REQUIRE(a.ok(), "inverse_lists: workspace carve failed");
  CCN_HIP(hipMemsetAsync(counts, 0, (size_t)(M + 1) * 4, s), "inverse_lists");
  if (n > 0) {
    if (src_is_i64)
      hipLaunchKernelGGL(inv_pack_kernel<int64_t>, dim3(ccn_blocks(n, 256)), dim3(256), 0, s, (const int64_t*)src, n, M, key, counts);
    else
      hipLaunchKernelGGL(inv_pack_kernel<int32_t>, dim3(ccn_blocks(n, 256)), dim3(256), 0, s, (const int32_t*)src, n, M, key, counts);
  }
  int rc = ccn_scan_i32(counts, inv_ptr, M + 1, false, nullptr, scratch, s);       // inv_ptr[M] = number of rows listed
  if (rc) return rc;
  if (n > 0) {
    int mask = 0;
    for (int b = 0; b < 4; ++b)
      if (b == 0 || ((uint64_t)M >> (8 * b)) != 0) mask |= 1 << b;                  // the digits in which keys 0 .. M can differ
    const uint64_t* kin;
    const int32_t* vin;
    int32_t *flag, *incl;
    void* scan_ws;
    rc = rs_sort(key, n, mask, sort_ws, sort_bytes, s, &kin, &vin, &flag, &incl, &scan_ws);
    if (rc) return rc;
    CCN_HIP(hipMemcpyAsync(inv_row, vin, (size_t)n * 4, hipMemcpyDeviceToDevice, s), "inverse_lists");
  }
  CCN_LAUNCH_OK("inverse_lists");
  return CCN_OK;
}

}  // extern "C"

int ccn_sort_payload(const int64_t* key, int64_t n, int digit_mask, void* ws, size_t ws_bytes, hipStream_t s,
                     const int32_t** vout) {
  const uint64_t* kin;
  int32_t *flag, *incl;
  void* scan_ws;
  return rs_sort(key, n, digit_mask, ws, ws_bytes, s, &kin, vout, &flag, &incl, &scan_ws);
}


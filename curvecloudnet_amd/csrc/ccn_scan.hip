// Device-wide prefix sums (reduce-then-scan, three launches) + error plumbing.
// Used for: run/segment ranking (A1/A2), CSR edge offsets (A8), stream compaction (A7),
// the float64 arclength running sum (A7) and the hash-grid cell offsets (A11).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "ccn_common.h"

static thread_local char g_err[512] = "";

void ccn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ccn_last_error(void) { return g_err; }
extern "C" int ccn_abi_version(void) { return CCN_ABI_VERSION; }

// ONE door for every diagnostics / A-B / test hook (include/ccn_hip_debug.h): serialised by a mutex, so that a hook flipped from
// one thread never interleaves with another flip; the state behind the hooks is std::atomic, so the launch paths of other threads
// (the geometry worker of ModelBase.prepare_async beside the main thread) read it without a data race.
extern "C" int ccn_debug_set(const char* key, int64_t value) {
  static std::mutex mu;
  struct Hook {
    const char* key;
    int (*fn)(int);
  };
  static const Hook hooks[] = {
      {"gemm_use_dma", ccn_gemm_use_dma},         {"gemm_pair_opt", ccn_gemm_pair_opt},
      {"gemm_force_generic", ccn_gemm_force_generic}, {"gemm_h_opt", ccn_gemm_h_opt},
      {"frnn_query_mode", ccn_frnn_query_mode},   {"gemm_x3_use_persistent", ccn_gemm_x3_use_persistent},
      {"gemm_tn_use_dma", ccn_gemm_tn_use_dma},   {"gemm_tn_background", ccn_gemm_tn_background},
      {"fps_set_lds_claim", ccn_fps_set_lds_claim}, {"fps_use_cluster", ccn_fps_use_cluster},
      {"fps_debug_fault", ccn_fps_debug_fault},
  };
  CCN_REQUIRE(key != nullptr, "debug_set: null key");
  CCN_REQUIRE(value >= INT32_MIN && value <= INT32_MAX, "debug_set: %s: value out of range", key);
  std::lock_guard<std::mutex> lock(mu);
  for (const Hook& h : hooks)
    if (strcmp(h.key, key) == 0) return h.fn((int)value);
  ccn_set_error("debug_set: unknown key '%s'", key);
  return CCN_ERR_ARG;
}

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;  // elements per workgroup

template <typename T>
__device__ __forceinline__ T wave_inclusive(T v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// inclusive scan of one value per thread across the 256-thread workgroup; returns the
// inclusive value, *block_total = sum of all.
template <typename T>
__device__ __forceinline__ T block_inclusive(T v, T* lds /*4*/, T* block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T inc = wave_inclusive(v, lane);
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  T add = 0;
  T tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    T x = lds[w];
    if (w < wave) add += x;
    tot += x;
  }
  __syncthreads();
  *block_total = tot;
  return inc + add;
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_chunk_sums(const T* __restrict__ in, T* __restrict__ partial,
                                                                int64_t n) {
  __shared__ T lds[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  T s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j)
    if (base + j < n) s += in[base + j];
  T tot;
  block_inclusive(s, lds, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the chunk sums in place; partial[nb] = grand total
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_partials(T* __restrict__ partial, int64_t nb) {
  __shared__ T lds[4];
  __shared__ T carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += SCAN_THREADS) {
    const int64_t i = base + threadIdx.x;
    T v = i < nb ? partial[i] : (T)0;
    T tot;
    T inc = block_inclusive(v, lds, &tot);
    T carry = carry_s;
    if (i < nb) partial[i] = carry + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[nb] = carry_s;
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const T* in, T* out,
                                                           const T* __restrict__ partial, int64_t n, int inclusive,
                                                           T* __restrict__ total_out, int64_t nb) {
  __shared__ T lds[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  T v[SCAN_ITEMS];
  T s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = base + j < n ? in[base + j] : (T)0;
    s += v[j];
  }
  T tot;
  T inc = block_inclusive(s, lds, &tot);
  T run = partial[blockIdx.x] + inc - s;  // exclusive prefix of this thread's first item
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) out[base + j] = inclusive ? run + v[j] : run;
    run += v[j];
  }
  if (total_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *total_out = partial[nb];
}

// Integer scans with at most SCAN_INLINE_BLOCKS chunks skip the single-workgroup pass over the chunk sums: every
// workgroup adds up the (raw) sums of the chunks in front of it itself -- two launches instead of three for the many
// small scans of a forward pass (exact for integers; the fp64 scan keeps its fixed summation order).
constexpr int SCAN_INLINE_BLOCKS = 4096;

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_inline(const T* in, T* out, const T* __restrict__ chunk_sums,
                                                                  int64_t n, int inclusive, T* __restrict__ total_out,
                                                                  int64_t nb) {
  __shared__ T lds[4];
  __shared__ T offset_s, total_s;
  {
    T before = 0, all = 0;
    for (int64_t j = threadIdx.x; j < nb; j += SCAN_THREADS) {
      const T c = chunk_sums[j];
      all += c;
      if (j < (int64_t)blockIdx.x) before += c;
    }
    T tot_b, tot_a;
    block_inclusive(before, lds, &tot_b);
    block_inclusive(all, lds, &tot_a);
    if (threadIdx.x == 0) {
      offset_s = tot_b;
      total_s = tot_a;
    }
    __syncthreads();
  }
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  T v[SCAN_ITEMS];
  T sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = base + j < n ? in[base + j] : (T)0;
    sum += v[j];
  }
  T tot;
  T inc = block_inclusive(sum, lds, &tot);
  T run = offset_s + inc - sum;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) out[base + j] = inclusive ? run + v[j] : run;
    run += v[j];
  }
  if (total_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *total_out = total_s;
}

template <typename T>
struct ScanIsInteger {
  static constexpr bool value = false;
};
template <>
struct ScanIsInteger<int32_t> {
  static constexpr bool value = true;
};

template <typename T>
int scan_impl(const T* in, T* out, int64_t n, bool inclusive, T* total_out, void* scratch, hipStream_t s) {
  if (n <= 0) {
    if (total_out) CCN_HIP(hipMemsetAsync(total_out, 0, sizeof(T), s), "scan");
    return CCN_OK;
  }
  const int64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  T* partial = (T*)scratch;
  hipLaunchKernelGGL(scan_chunk_sums<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, partial, n);
  if (ScanIsInteger<T>::value && nb <= SCAN_INLINE_BLOCKS) {
    hipLaunchKernelGGL(scan_apply_inline<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, partial, n,
                       inclusive ? 1 : 0, total_out, nb);
    CCN_LAUNCH_OK("scan");
    return CCN_OK;
  }
  hipLaunchKernelGGL(scan_partials<T>, dim3(1), dim3(SCAN_THREADS), 0, s, partial, nb);
  hipLaunchKernelGGL(scan_apply<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, partial, n,
                     inclusive ? 1 : 0, total_out, nb);
  CCN_LAUNCH_OK("scan");
  return CCN_OK;
}

}  // namespace

size_t ccn_scan_scratch_bytes(int64_t n) {
  const int64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  return ccn_align256((size_t)(nb + 2) * sizeof(double));
}

int ccn_scan_i32(const int32_t* in, int32_t* out, int64_t n, bool inclusive, int32_t* total_out, void* scratch,
                 hipStream_t s) {
  return scan_impl<int32_t>(in, out, n, inclusive, total_out, scratch, s);
}

int ccn_scan_f64(const double* in, double* out, int64_t n, bool inclusive, void* scratch, hipStream_t s) {
  return scan_impl<double>(in, out, n, inclusive, (double*)nullptr, scratch, s);
}

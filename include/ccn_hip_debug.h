/* libccn_hip.so -- diagnostics, A/B hooks and test hooks.  NOT part of the drop-in boundary (include/ccn_hip.h): nothing a
 * host integrating the library needs.  Used by tests/ (kernel-variant coverage) and tools/ (measurements in profiles/).
 * Each sets process-global dispatch state; none changes results unless its comment says so. */
#ifndef CCN_HIP_DEBUG_H
#define CCN_HIP_DEBUG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ONE door for the integer hooks below: ccn_debug_set("<name without ccn_>", value), e.g. ccn_debug_set("fps_use_cluster", 0).
 * Serialised by a mutex; the state behind every hook is std::atomic (the geometry worker thread and the main thread both launch
 * kernels).  Unknown key: CCN_ERR_ARG.  The individual functions stay for the tests and tools that call them directly. */
int ccn_debug_set(const char* key, int64_t value);

int ccn_gemm_use_dma(int on);       /* A/B hook: 0 = register-staged kernels only, 2 = LDS-DMA without the persistent tile loop, 3 = persistent with round-robin tiles, 4 = the 8-wave persistent kernel for every N (no paired 4-wave workgroups), 1 = default */
int ccn_gemm_pair_debug(void* buf);  /* diagnostic: a device buffer of 512 x 4 x 16 uint64 words switches ccn_gemm_nt's paired kernel to a build that stamps (s_memtime) where every wave's cycles go; NULL = off (tools/pair_stamps.py) */
int ccn_gemm_pair_opt(int bits);    /* A/B hook of the paired kernel's LAUNCHER: bit 2 = one workgroup per CU, bit 6 = no split of a wide product into a 128-wide and a 64-wide launch, bit 8 = the 8-wave kernel for N <= 64, bit 9 = no tail split even with scratch (the in-kernel experiments of rounds 2-3 are no longer compiled) */
int ccn_gemm_force_generic(int on); /* test hook: route every GEMM through the unaligned-operand kernel */
/* diagnostics of ccn_gemm_nt_h (timing only, results wrong when set; a separate instantiation of the kernel): bit 0 = no epilogue stores, bit 1 = no wait for the copies, bit 2 = start stagger of the second workgroup of a CU */
int ccn_gemm_h_opt(int opt);
int ccn_frnn_query_mode(int mode); /* A/B hook: 0 = automatic, 1 = one thread per query, 2 / 3 = a team of 32 / 64 lanes per query */
int ccn_gemm_x3_use_persistent(int on); /* A/B hook: 0 = the register-staged kernel for every shape, 2 = no paired 4-wave workgroups (8-wave persistent kernel for every N), 1 = default */
int ccn_gemm_x3_debug(void* buf);       /* diagnostic: a device buffer of 512 x 4 x 16 uint64 words switches ccn_gemm_nt_x3's software-pipelined kernel to a build that stamps (s_memtime) where every wave's cycles go; NULL = off (tools/x3_stamps.py) */
int ccn_gemm_tn_use_dma(int on);    /* A/B hook: 0 = always the register-staged split-K kernel of ccn_gemm_tn */
int ccn_gemm_tn_background(int lds_bytes); /* experiment (round 6): total LDS a weight-gradient workgroup claims (static + unused dynamic); 86016 = ONE per CU, leaving half of the register file and 74 KiB of LDS to the other stream's kernels; 0 = off (default). Same results. */
int ccn_fps_set_lds_claim(int bytes); /* A/B hook: dynamic LDS a sampling workgroup claims to keep its CU free of GEMM workgroups (default and maximum 98304, 0 = none) */
int ccn_fps_use_cluster(int on);      /* A/B hook (round 5): 0 = exact FPS with ONE workgroup per cloud whatever its size (the hybrid / streaming forms); default 1 = clouds of more than 16384 points by a cluster of up to four workgroups; 2 = the cluster with agent-scope (sc1) stores even when its members share an XCD */
int ccn_fps_debug_fault(int mode);    /* test hook (round 6): 1 = member 1 of every cluster of ccn_fps silently leaves before round 1 (its partners run into the poll timeout, ~0.5 s), 2 = it raises the abort word itself and leaves; either way the gated one-workgroup launch re-samples the cloud; 0 = off */

#ifdef __cplusplus
}
#endif
#endif

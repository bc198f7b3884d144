/* libpggan_hip.so — diagnostic exports (NOT part of the drop-in boundary).
 *
 * Tuning and attribution aids used by bench.py and tools/: they read / set THREAD-LOCAL state of the calling thread
 * (the last launched kernel symbol, a forced tile configuration), which the product path never touches.  Kept apart
 * from pggan_hip.h so that the product header states its re-entrancy contract without exceptions. */
#ifndef PGGAN_HIP_DEBUG_H
#define PGGAN_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Profiling aid: symbol (as rocprofv3 prints it, e.g. "conv_igemm_kernel<3, 4, 2, 2, 4>") of the kernel instantiation most
 * recently launched by the calling thread through pg_conv2d_*_nhwc / pg_conv2d_wgrad_*_nhwc, pg_conv2d_wino_nhwc,
 * pg_conv2d_wgrad_wino_nhwc ("" before the first launch): lets bench.py attribute its HIP-event timings to the exact
 * symbol that rocprofv3 --kernel-trace --stats reports. */
const char* pg_debug_last_conv_kernel(void);
const char* pg_debug_last_wino_kernel(void);
const char* pg_debug_last_wino_wgrad_kernel(void);

/* Tuning aids (tools/sweeps/microbench_conv.py, tools/sweep_*.py): force a configuration for the calling thread's next launches.
 * pg_debug_set_tuning: key 0 conv tile candidate, key 1 weight-gradient configuration, key 2 conv split-K factor (further
 * keys: see csrc/conv_igemm.hip); value -1 restores the built-in choice.  pg_debug_set_wino: K-chunk of 4*vec channels. */
int pg_debug_set_tuning(int key, int value);
int pg_debug_set_wino(int vec);
/* K slices per (tile block, cout block) of the second-generation Winograd conv: -1 built-in choice, 0 / 1 never split, n: n slices
 * wherever a scratch is registered (pg_set_workspace) and the layer has that many 8-channel chunks. */
int pg_debug_set_wino_ksplit(int n);
/* 0: the general epilogue for every launch of the second-generation Winograd conv (A/B against the specialised ones); -1: built-in choice. */
int pg_debug_set_wino_epi(int mode);

#ifdef __cplusplus
}
#endif
#endif /* PGGAN_HIP_DEBUG_H */

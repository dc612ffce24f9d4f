/* Builder-side profiling hooks of libcapreolus_amd.so.  NOT part of the scoring interface (include/capreolus_amd.h does not declare
 * them): bench.py's `roofline` object and the scripts/ probes bind them through capreolus_amd._lib.profiling(). */
#ifndef CAPAMD_PROFILING_H
#define CAPAMD_PROFILING_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* (off by default)
 * capamd_debug_set_gemm_stamps: when non-NULL, capamd_bert_gemm blocks write up to 32 s_memtime stamps each into
 * stamps[block][32] (uint64, device memory).  Pass NULL to switch it off.
 * capamd_debug_ffn1_timing (used by bench.py's `roofline` object): while enabled, capamd_bert_maxp_forward brackets
 * every launch of its dominant kernel (the FFN1 GEMM, bias + GELU) with HIP events on the caller's stream;
 * capamd_debug_ffn1_timing_read synchronises those events, returns the summed duration in milliseconds, the number of
 * launches and the summed GEMM rows since the last read, and clears the list. */
void capamd_debug_set_gemm_stamps(void* stamps);
/* capamd_debug_lists_timing (bench.py's headline leg): while enabled, the whole-list entries (capamd_*_forward_lists) record a HIP event
 * on the caller's stream after each of their passes - memset, mark, query, sims, pool - of every launch group;
 * capamd_debug_lists_timing_read synchronises them, adds the five durations in milliseconds to ms[0..4], returns the number of launch
 * groups since the last read and clears the list. */
void capamd_debug_lists_timing(int enable);
int capamd_debug_lists_timing_read(double* ms);
void capamd_debug_ffn1_timing(int enable);
int capamd_debug_ffn1_timing_read(double* total_ms, int64_t* launches, int64_t* rows);

#ifdef __cplusplus
}
#endif
#endif

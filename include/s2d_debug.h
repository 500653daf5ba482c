/*
 * s2d_debug.h - debugging / tuning entries of libs2d_hip.so.  NOT part of the drop-in boundary (include/s2d.h): nothing in the reference
 * binds them; they exist for tools/side_stress.py and tools/spconv_kernel_bench.py and may change without notice.
 */
#ifndef S2D_DEBUG_H
#define S2D_DEBUG_H

#include "s2d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* debugging aid (no reference counterpart): `blocks` workgroups fill 64 KB of LDS each with `value`, `spin` + 1 times; sink may be NULL.
 * tools/side_stress.py uses it to look for kernels that read LDS they did not write. */
int s2d_debug_lds_fill(float value, int blocks, int spin, float *sink, s2d_stream_t stream);

/* tuning aid (tools/spconv_kernel_bench.py --trace): device buffer int64[grid][64] that the ablation build of the register-gather
 * kernel (csrc/spconv_rg.hip) fills with per-step s_memtime stamps when S2D_RG_DEBUG has bit 32 set; NULL switches it off */
void s2d_debug_rg_trace(void *buf);

#ifdef __cplusplus
}
#endif

#endif /* S2D_DEBUG_H */

/* somhip_test.h -- entry points of libsomhip.so that exist for the test suite and the measurement tools only.
 * Not part of the drop-in boundary (include/somhip.h): nothing a caller of the hot path needs, nothing whose behaviour is
 * promised from one build to the next. */
#ifndef SOMHIP_TEST_H
#define SOMHIP_TEST_H

#include "somhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the canary's TEST HOOK (som_set_verify, include/somhip.h): zeroes the operand images the kernels read (bit 0: the 16-bit
 * image, bit 1: the float32 image) without marking them stale, as a lost staging copy would (tests/test_gpu_verify.py). */
int som_debug_corrupt_operands(som_handle* h, int32_t which);

/* measurement hook: ONE v_mfma_f32_16x16x32 (_f16 when is_f16, else _bf16) on the caller's operands -- a [16][32] and
 * b [32][16] as 16-bit patterns, c and d [16][16] float32, row-major.  tests/test_gpu_exact.py uses it to measure the
 * rounding error the exact mode's bound charges per MFMA (the hardware's internal summation is not documented). */
int som_debug_mfma16(som_handle* h, const uint16_t* a_host, const uint16_t* b_host, const float* c_host, float* d_host,
                     int32_t is_f16);

/* diagnostic builds only (-DSOM_STAMPS, tools/stamps.py builds one on demand): out_host == NULL attaches a buffer of n_pairs
 * (shader-clock ticks, 100 MHz ticks) pairs, one per workgroup of the next BMU launches (n_pairs == 0 detaches);
 * out_host != NULL reads n_pairs pairs back.  The product build refuses both. */
int som_debug_stamps(som_handle* h, int64_t n_pairs, uint64_t* out_host);

/* The exact mode's POLICY (csrc/exact_policy.hpp: commit a scouted plan, level 2, did a sort pay, is a plan idle, does the
 * scout go on, is a row set worth a scout) on caller-supplied numbers -- pure host arithmetic, NO device needed: the seam the
 * CPU suite tests the decisions through (tests/test_policy_cpu.py).
 *   costs[9] = {full_total, full_screen, plan_total, plan_over, plan_over_scout, blk_ms, l2_ms_group, l2_ratio, sort_ms}
 *              (ms per row; blk_ms per 16-unit block; l2_ms_group per kept (tile, group) pair; 0 = not measured yet)
 *   which: 0 commit_scouted_plan(share, blocks_per_row)   1 level2_from_sample(share, share1, blocks_per_row)
 *          2 level2_pays(share, share1)                   3 sort_paid(share_stale, share_fresh, blocks_per_row, epochs_served)
 *          4 plan_idle(share)                             5 scout_continues(win_share, share_last, blocks_per_row)
 *          6 rows_worth_a_scout(rows, units, features)
 *   *out = 0 / 1; returns non-zero for an unknown `which` or a NULL argument. */
int som_policy_eval(int32_t which, const double* costs, const double* args, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif

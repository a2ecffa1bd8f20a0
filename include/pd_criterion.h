/*
 * pd_criterion.h — C-ABI of the matcher / criterion kernels of libpd_hip.so.
 *
 * These replace, on the device, work the reference does through PyTorch +
 * SciPy on the host side of the training step:
 *   pd_lsa_batched   scipy.optimize.linear_sum_assignment + the cost-ordered
 *                    pair sort, reference part_distillation/modeling/matcher.py:159-163
 *                    (there: C.cpu() -> SciPy -> topk; one D2H sync per image
 *                    per decoder layer).
 * All pointers are device pointers; `stream` is a hipStream_t.  Return 0 or a
 * negative PD_ERR_* (pd_msda.h); message via pd_last_error().
 */
#ifndef PD_CRITERION_H
#define PD_CRITERION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Solve `nbatch` independent rectangular assignment problems (minimise).
 *   cost      float32 [nbatch, nrows, ncols_max] row-major; problem b uses columns [0, ncols[b])
 *   ncols     int32   [nbatch]   (0 <= ncols[b] <= ncols_max)
 *   out_rows  int64   [nbatch, ncols_max]  selected row  (query)  of pair k, pairs sorted by ascending cost
 *   out_cols  int64   [nbatch, ncols_max]  selected col  (target) of pair k; entries k >= min(nrows, ncols[b]) are -1
 * The solver is the float64 shortest-augmenting-path algorithm SciPy uses
 * (Crouse 2016), including its tie-breaking scan order, so that the result is
 * SciPy's for the same matrix.  Limits: nrows <= 4096, ncols_max <= 64.
 */
int pd_lsa_batched(const float *cost, const int32_t *ncols, int64_t *out_rows, int64_t *out_cols,
                   int nbatch, int nrows, int ncols_max, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_CRITERION_H */

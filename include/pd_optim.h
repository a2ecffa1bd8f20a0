/*
 * pd_optim.h — C-ABI of the optimizer kernels of libpd_hip.so.
 *
 * Replaces the many-tensor loops of the reference's optimizer step,
 *   base_trainer.py:118-133  FullModelGradientClippingOptimizer.step():
 *       torch.nn.utils.clip_grad_norm_(all_params, CLIP_VALUE)   (global L2 norm)
 *       torch.optim.AdamW.step()
 * by two bandwidth-bound passes over FLAT parameter / gradient / state buffers.
 * All pointers are device pointers; dtype is PD_F32 or PD_F64 (pd_msda.h).
 */
#ifndef PD_OPTIM_H
#define PD_OPTIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* *accum (float64, device) += sum_i x[i]^2.  The caller zeroes *accum. */
int pd_sumsq_accumulate(const void *x, int64_t n, int dtype, double *accum, void *stream);

/*
 * One AdamW update (decoupled weight decay, torch.optim.AdamW arithmetic) of n
 * elements with global-norm clipping folded in: every gradient is first scaled
 * by coef = min(1, max_norm / (sqrt(*grad_sumsq) + 1e-6)) exactly as
 * torch.nn.utils.clip_grad_norm_ does; max_norm <= 0 disables clipping
 * (grad_sumsq may then be NULL).  `step` is the 1-based step count.
 * `dyn` (nullable, device, float[3] = {lr, 1-beta1^step, sqrt(1-beta2^step)}) overrides lr / step when given, so a
 * launch captured in a hipGraph keeps following the LR schedule and the bias corrections on replay.
 */
int pd_adamw_clipped(void *param, const void *grad, void *exp_avg, void *exp_avg_sq, int64_t n, int dtype,
                     double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                     const double *grad_sumsq, double max_norm, const float *dyn, void *stream);

/* fp32 variant that also writes a bf16 (round-to-nearest-even) copy of the updated parameters to `shadow_bf16`:
 * the autocast (bf16) modules read that copy directly instead of re-casting every weight every step.
 * n % 4 == 0, 16-byte aligned buffers (8-byte for the shadow). */
int pd_adamw_clipped_shadow(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, void *shadow_bf16,
                            int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                            const double *grad_sumsq, double max_norm, const float *dyn, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_OPTIM_H */

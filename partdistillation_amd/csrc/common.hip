// Error plumbing + version of libpd_hip.so (C-ABI declared in include/pd_msda.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pd_common.h"
#include "pd_msda.h"

namespace {
thread_local char g_err[512] = {0};
}

int pd_set_error(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int pd_check_launch(const char *what)
{
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return PD_OK;
  return pd_set_error(PD_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

extern "C" const char *pd_last_error(void) { return g_err; }
extern "C" int pd_abi_version(void) { return 39; }

// experiment knobs (not part of the public ABI contract; used by tools/ only)
extern int g_pd_dbg_atomic_scope;
extern int g_pd_dbg_force_generic;
extern int g_pd_dbg_ablate;
extern int g_pd_dbg_x3_narrow;
extern int g_pd_dbg_f16x2;
extern int g_pd_dbg_wgrad_wide;
extern "C" void pd_dbg_set_wgrad_xcd(int v);
extern "C" void pd_dbg_set_wgrad_fast(int v);
extern int g_pd_dbg_bwd_variant;
extern int g_pd_dbg_msda_gate_pct;
extern int g_pd_dbg_wgrad_wgs;
extern int g_pd_dbg_bwd_threads;
extern int g_pd_dbg_attn_scalar;
extern int g_pd_dbg_wattn;
extern int g_pd_dbg_x3;
extern int g_pd_dbg_kmeans;
extern int g_pd_dbg_conv_group_rows;
extern int g_pd_dbg_conv_xcd_major;
extern int g_pd_dbg_sgemm_deep;
extern int g_ig_bn, g_ig_nst, g_ig_splits, g_ig_patch, g_ig_pcls;
extern int g_mx_bn, g_mx_nst;
extern int g_swin_ln_abl;
extern int g_ln_bwd_cap;
extern int g_crit_abl;
extern int g_attn_bwd_kc;
extern int g_wg_nst, g_wg_splits, g_wg_mode;
extern "C" int pd_debug_set(const char *key, int value)
{
  if (!key) return PD_ERR_INVALID_ARG;
  if (!strcmp(key, "msda_bwd_atomic_scope")) { g_pd_dbg_atomic_scope = value; return PD_OK; }
  if (!strcmp(key, "msda_ablate")) { g_pd_dbg_ablate = value; return PD_OK; }
  if (!strcmp(key, "msda_bwd_variant")) { g_pd_dbg_bwd_variant = value; return PD_OK; }
  if (!strcmp(key, "msda_gate_pct")) { g_pd_dbg_msda_gate_pct = value; return PD_OK; }
  if (!strcmp(key, "attn_scalar")) { g_pd_dbg_attn_scalar = value; return PD_OK; }
  if (!strcmp(key, "msda_bwd_threads")) { g_pd_dbg_bwd_threads = value; return PD_OK; }
  if (!strcmp(key, "wgrad_wgs")) { g_pd_dbg_wgrad_wgs = value; return PD_OK; }
  if (!strcmp(key, "kmeans_ablate")) { g_pd_dbg_kmeans = value; return PD_OK; }
  if (!strcmp(key, "wg_nst")) { g_wg_nst = value; return PD_OK; }
  if (!strcmp(key, "wg_splits")) { g_wg_splits = value; return PD_OK; }
  if (!strcmp(key, "wg_mode")) { g_wg_mode = value; return PD_OK; }
  if (!strcmp(key, "ig_bn")) { g_ig_bn = value; return PD_OK; }
  if (!strcmp(key, "ig_patch")) { g_ig_patch = value; return PD_OK; }
  if (!strcmp(key, "ig_pcls")) { g_ig_pcls = value; return PD_OK; }
  if (!strcmp(key, "ig_nst")) { g_ig_nst = value; return PD_OK; }
  if (!strcmp(key, "mx_bn")) { g_mx_bn = value; return PD_OK; }
  if (!strcmp(key, "attn_bwd_kc")) { g_attn_bwd_kc = value; return PD_OK; }
  if (!strcmp(key, "crit_abl")) { g_crit_abl = value; return PD_OK; }
  if (!strcmp(key, "ln_bwd_cap")) { g_ln_bwd_cap = value; return PD_OK; }
  if (!strcmp(key, "swin_ln_abl")) { g_swin_ln_abl = value; return PD_OK; }
  if (!strcmp(key, "mx_nst")) { g_mx_nst = value; return PD_OK; }
  if (!strcmp(key, "ig_splits")) { g_ig_splits = value; return PD_OK; }
  if (!strcmp(key, "sgemm_deep")) { g_pd_dbg_sgemm_deep = value; return PD_OK; }
  if (!strcmp(key, "conv_group_rows")) { g_pd_dbg_conv_group_rows = value; return PD_OK; }
  if (!strcmp(key, "conv_xcd_major")) { g_pd_dbg_conv_xcd_major = value; return PD_OK; }
  if (!strcmp(key, "x3_ablate")) { g_pd_dbg_x3 = value; return PD_OK; }
  if (!strcmp(key, "f16x2_tile")) { g_pd_dbg_f16x2 = value; return PD_OK; }
  if (!strcmp(key, "wgrad_wide")) { g_pd_dbg_wgrad_wide = value; return PD_OK; }
  if (!strcmp(key, "wgrad_xcd")) { pd_dbg_set_wgrad_xcd(value); return PD_OK; }
  if (!strcmp(key, "wgrad_fast")) { pd_dbg_set_wgrad_fast(value); return PD_OK; }
  if (!strcmp(key, "x3_narrow")) { g_pd_dbg_x3_narrow = value; return PD_OK; }
  if (!strcmp(key, "wattn_ablate")) { g_pd_dbg_wattn = value; return PD_OK; }
  if (!strcmp(key, "msda_force_generic")) { g_pd_dbg_force_generic = value; return PD_OK; }
  return pd_set_error(PD_ERR_INVALID_ARG, "pd_debug_set: unknown key %s", key);
}

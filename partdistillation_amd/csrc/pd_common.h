// Internal helpers shared by the HIP translation units of libpd_hip.so.
#ifndef PD_COMMON_H
#define PD_COMMON_H
#include <hip/hip_runtime.h>

// Records a printf-style message for pd_last_error() and returns `code`.
int pd_set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// hipGetLastError() -> PD_OK / PD_ERR_LAUNCH (message recorded).
int pd_check_launch(const char *what);


// A pointer READ FROM MEMORY (a problem table of a grouped launch) has no address space the compiler could know: every access through it is a FLAT
// instruction, and flat loads count in lgkmcnt as well as vmcnt — each s_waitcnt lgkmcnt(0) in front of an LDS read or a barrier then also waits for
// every global load in flight, i.e. the operand prefetch of a GEMM loop is drained at every step.  pd_as_global(p) says "global memory" for a
// WAVE-UNIFORM pointer (a table field): cast to address space 1, pinned in scalar registers by an empty asm (without it the cast pair folds away
// and the accesses stay flat; with it InferAddressSpaces sees an address-space-1 origin), and back.  The accesses become global_load / global_store
// with a scalar base.  Kernel ARGUMENTS (and the fields of by-value argument structs) do not need it.
#ifdef __HIPCC__
template <typename T>
__device__ __forceinline__ T *pd_as_global(T *p)
{
  typedef __attribute__((address_space(1))) T *G;
  G g = (G)p;
  asm("" : "+s"(g));
  return (T *)g;
}
#endif
#endif

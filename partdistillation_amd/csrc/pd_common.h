// Internal helpers shared by the HIP translation units of libpd_hip.so.
#ifndef PD_COMMON_H
#define PD_COMMON_H
#include <hip/hip_runtime.h>

// Records a printf-style message for pd_last_error() and returns `code`.
int pd_set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// hipGetLastError() -> PD_OK / PD_ERR_LAUNCH (message recorded).
int pd_check_launch(const char *what);

#endif

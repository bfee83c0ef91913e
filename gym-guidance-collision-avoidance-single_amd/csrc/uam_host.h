/*
 * uam_host.h — what the host-side translation units of libuamenv_hip.so share (not part of the boundary; hidden
 * from the library's export table): the thread-local error text behind uam_last_error() and the exact squared
 * distance thresholds.
 */
#ifndef UAM_HOST_H
#define UAM_HOST_H

#include <hip/hip_runtime.h>
#include "../../include/uamenv.h"

#define UAM_HIDDEN __attribute__((visibility("hidden")))

/* records the message for uam_last_error() and returns `code` */
UAM_HIDDEN int uam_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

/* min{v >= 0 : sqrt(v) >= thr}: `sqrt(d2) < thr`  <=>  `d2 < T2` exactly */
UAM_HIDDEN double uam_sq_threshold_f64(double thr);

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return uam_fail(UAM_EIO, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#endif /* UAM_HOST_H */

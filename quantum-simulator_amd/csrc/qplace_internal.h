// Shared by the translation units of libqplace.so; not part of the public ABI.
#pragma once
#include "../../include/qplace.h"

namespace qp {
// Records a printf-style message for qp_last_error() and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Compute units of the current device (256 when the runtime cannot say).  Device translation units only.
int device_cus();
}  // namespace qp

// For functions that return a QP_* code: a failing HIP call becomes QP_EHIP with the call's text.
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return qp::fail(QP_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

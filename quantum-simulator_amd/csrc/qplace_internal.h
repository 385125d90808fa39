// Shared by the translation units of libqplace.so; not part of the public ABI.
#pragma once
#include "../../include/qplace.h"

namespace qp {
// Records a printf-style message for qp_last_error() and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Compute units of the current device (256 when the runtime cannot say).  Device translation units only.
int device_cus();

// The assignment solver behind qp_lsap_f64 with caller-held scratch, for loops that solve many problems on
// one thread: cost has row pitch ld; QP_OK, QP_EINVAL (NaN / -inf entries) or QP_EINFEASIBLE.  Sets no message.
struct LsapWork;
LsapWork* lsap_work_new();
void lsap_work_free(LsapWork*);
int lsap(int32_t nr, int32_t nc, const double* cost, int32_t ld, int32_t* row_out, int32_t* col_out, LsapWork* work);

// The make-space step of qp_make_space on one assignment row and its occupancy counts.
void make_space_row(uint8_t* a, int64_t* occ, int32_t Q, int32_t C, int32_t cap, const double* D);
}  // namespace qp

// For functions that return a QP_* code: a failing HIP call becomes QP_EHIP with the call's text.
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return qp::fail(QP_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

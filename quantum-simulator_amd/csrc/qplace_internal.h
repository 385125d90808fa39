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

// qp_cost_matrix_multi (HQA / TRHQA) for callers that hold the live rows themselves: instead of op ids and rows,
// resolved[B][max_ops][8] holds per op the cores of its up to six look-ahead partners (-1 = none, in the order of
// QpCircuit.lookahead) and of its own two qubits.  Every pointer but `tables` may be device-visible host memory:
// the kernel then needs one round of loads from it, not a chain.  Device translation units only.
int cost_matrix_resolved(int32_t kind, const double* tables, int32_t C, const int32_t* tab_index, int64_t B,
                         const int16_t* resolved, const int32_t* n_ops, int32_t max_ops, const uint8_t* avail,
                         const int32_t* n_avail, double* cost_out, void* stream);

// qp_core_activity through the pair counters of the dense cost kernels (rows of 513 to 1024 bytes with the per-slice
// gate list, shorter rows with the grouped one when a partition is 32 KB or more): adds the counts to out[B][C], which the caller has zeroed.  QP_OK, a negative code, or 1 when
// this form does not apply.  Device translation units only.
int core_activity_pairs(const QpCircuit* circ, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, int32_t C,
                        uint64_t* out, void* stream);

// qp_comm_matrix through the same pair counters, walking the partitions row by row (rows of 513 to 1024 bytes, a circuit
// with visit_mask / visit_prev that visits most of every row): adds the counts to out[B][C][C] and sets status[b] where
// the reference raises; both zeroed by the caller.  QP_OK, a negative code, or 1 when this form does not apply.
int comm_matrix_pairs(const QpCircuit* circ, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, int32_t C, int32_t* out,
                      int32_t* status, void* stream);

// The make-space step of qp_make_space on one assignment row and its occupancy counts.
void make_space_row(uint8_t* a, int64_t* occ, int32_t Q, int32_t C, int32_t cap, const double* D);
}  // namespace qp

// For functions that return a QP_* code: a failing HIP call becomes QP_EHIP with the call's text.
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return qp::fail(QP_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

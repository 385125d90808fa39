// Shared by the translation units of libqplace.so; not part of the public ABI.
#pragma once
#include "../../include/qplace.h"

#include <atomic>
#include <exception>
#include <new>
#include <thread>
#include <vector>

namespace qp {
// Records a printf-style message for qp_last_error() and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// fn(0) .. fn(n - 1), each on a host thread of its own (fn(0) on the caller's).  Nothing escapes: a worker
// that cannot be started (thread limit of the process) is run on the calling thread after the others, and an
// exception out of fn (a failed allocation in its own scratch vectors) ends that worker only and turns into
// QP_ENOMEM for the call.  Returns QP_OK or QP_ENOMEM (message set).
template <class F>
int run_workers(int32_t n, F&& fn) {
    std::atomic<int> bad{0};
    auto guarded = [&](int32_t w) {
        try {
            fn(w);
        } catch (const std::bad_alloc&) {
            bad.store(1);
        } catch (...) {
            bad.store(2);
        }
    };
    std::vector<std::thread> pool;
    std::vector<int32_t> here;  // workers that run on the calling thread
    try {
        pool.reserve(n > 1 ? (size_t)(n - 1) : 0);
        here.reserve(n > 0 ? (size_t)n : 0);
    } catch (...) {
        for (int32_t w = 0; w < n; ++w) guarded(w);
        pool.clear();
        return bad.load() ? fail(QP_ENOMEM, "a host worker ran out of memory or threw") : QP_OK;
    }
    here.push_back(0);
    for (int32_t w = 1; w < n; ++w) {
        try {
            pool.emplace_back(guarded, w);
        } catch (...) {
            here.push_back(w);
        }
    }
    for (int32_t w : here)
        if (w < n) guarded(w);
    for (auto& th : pool) th.join();
    return bad.load() ? fail(QP_ENOMEM, "a host worker ran out of memory or threw") : QP_OK;
}

// Host threads a call of this process may start: min(limit, budget), at least 1 (see qp_host_threads).
int32_t host_threads(int32_t limit);

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

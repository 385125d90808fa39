// Analysis outputs that ride on the same gathers as the cost sums: per-core inter-core activity of dense
// partitions (HQA_placer.py:288-321) and the core x core communication-count matrix of the gate-ordered
// walk over sparse partitions (Qubo_placer.py:208-245).  Counts are integers, so the order in which they
// are added is free: workgroups count into LDS and add their totals to the caller's zeroed output.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "qplace_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 64;

// grid = (row chunks, B).  Each workgroup counts rows [r0, r1) of one partition.
__global__ __launch_bounds__(kThreads) void k_core_activity(const uint8_t* A, const int32_t* sep, const QpEdge* edges,
                                                            int64_t Tp, int32_t T, int32_t Q, int32_t ldq, int32_t C,
                                                            unsigned long long* out) {
    __shared__ uint32_t hist[256];
    const int64_t b = blockIdx.y;
    const int r0 = blockIdx.x * kRowsPerBlock, r1 = min(T, r0 + kRowsPerBlock);
    if (r0 >= T) return;
    for (int c = threadIdx.x; c < 256; c += kThreads) hist[c] = 0;
    __syncthreads();
    const uint8_t* base = A + (size_t)b * Tp * ldq;
    // qubits that changed core between rows t-1 and t, t = 1 .. T-1
    for (int t = max(r0, 1); t < r1; ++t) {
        const uint8_t* cur = base + (size_t)t * ldq;
        const uint8_t* prv = cur - ldq;
        for (int q = threadIdx.x; q < Q; q += kThreads) {
            const uint32_t a = prv[q], c = cur[q];
            if (a != c) {
                atomicAdd(&hist[a], 1u);
                atomicAdd(&hist[c], 1u);
            }
        }
    }
    // counted gates whose qubits sit on different cores
    for (int e = sep[r0] + threadIdx.x; e < sep[r1]; e += kThreads) {
        const QpEdge ed = edges[e];
        const uint8_t* row = base + (size_t)ed.t * ldq;
        const uint32_t a = row[ed.u], c = row[ed.v];
        if (a != c) {
            atomicAdd(&hist[a], 1u);
            atomicAdd(&hist[c], 1u);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += kThreads)
        if (hist[c]) atomicAdd(&out[(size_t)b * C + c], (unsigned long long)hist[c]);
}

// grid = (visit chunks, B); one thread per (gate, qubit) visit.
__global__ __launch_bounds__(kThreads) void k_comm_matrix(const uint8_t* A, const QpVisit* visits, int32_t nV, int64_t Tp,
                                                          int32_t ldq, int32_t C, int32_t* out, int32_t* status) {
    const int64_t b = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nV) return;
    const uint8_t* base = A + (size_t)b * Tp * ldq;
    int32_t* M = out + (size_t)b * C * C;
    const QpVisit v = visits[i];
    const uint32_t cur = base[(size_t)v.t * ldq + v.q];
    // a core index outside the architecture (other than the "absent" mark) is reported like a missing
    // qubit and never used as an index
    bool bad = cur != QP_ABSENT && cur >= (uint32_t)C;
    if (bad) {
        status[b] = 1;
        return;
    }
    if (v.partner != 0xFFFFu) {
        const uint32_t oth = base[(size_t)v.t * ldq + v.partner];
        if (cur == QP_ABSENT || oth == QP_ABSENT || oth >= (uint32_t)C) {
            bad = true;  // KeyError in the reference
        } else if (cur != oth) {
            atomicAdd(&M[cur * C + oth], 1);
            atomicAdd(&M[oth * C + cur], 1);
        }
    }
    if (v.prev_t >= 0) {
        const uint32_t last = base[(size_t)v.prev_t * ldq + v.q];
        if (last != QP_ABSENT && cur != last) {
            if (cur == QP_ABSENT || last >= (uint32_t)C) {
                bad = true;  // list.index(None): ValueError in the reference
            } else {
                atomicAdd(&M[cur * C + last], 1);
                atomicAdd(&M[last * C + cur], 1);
            }
        }
    }
    if (bad) status[b] = 1;
}

// The same with the matrix in LDS (up to 128 cores): grid = (chunks of visits, B); a workgroup walks its chunk of the
// partition's visits, counts into its own copy and adds what is not zero to the caller's matrix at the end (stores it
// when the partition is one chunk).  The global atomics of k_comm_matrix meet on a few hundred addresses per partition.
__global__ __launch_bounds__(kThreads) void k_comm_matrix_lds(const uint8_t* A, const QpVisit* visits, int32_t nV, int32_t per_chunk,
                                                              int64_t Tp, int32_t ldq, int32_t C, int32_t* out, int32_t* status) {
    extern __shared__ int32_t lm[];
    const int64_t b = blockIdx.y;
    const int v0 = blockIdx.x * per_chunk, v1 = min(nV, v0 + per_chunk);
    for (int k = threadIdx.x; k < C * C; k += kThreads) lm[k] = 0;
    __syncthreads();
    const uint8_t* base = A + (size_t)b * Tp * ldq;
    bool bad = false;
    for (int i = v0 + threadIdx.x; i < v1; i += kThreads) {
        const QpVisit v = visits[i];
        const uint32_t cur = base[(size_t)v.t * ldq + v.q];
        const uint32_t oth = base[(size_t)v.t * ldq + (v.partner != 0xFFFFu ? v.partner : v.q)];
        const uint32_t last = base[(size_t)(v.prev_t >= 0 ? v.prev_t : v.t) * ldq + v.q];
        if (cur != QP_ABSENT && cur >= (uint32_t)C) {  // as in k_comm_matrix: reported, never used as an index
            bad = true;
            continue;
        }
        if (v.partner != 0xFFFFu) {
            if (cur == QP_ABSENT || oth == QP_ABSENT || oth >= (uint32_t)C) {
                bad = true;
            } else if (cur != oth) {
                atomicAdd(&lm[cur * C + oth], 1);
                atomicAdd(&lm[oth * C + cur], 1);
            }
        }
        if (v.prev_t >= 0 && last != QP_ABSENT && cur != last) {
            if (cur == QP_ABSENT || last >= (uint32_t)C) {
                bad = true;
            } else {
                atomicAdd(&lm[cur * C + last], 1);
                atomicAdd(&lm[last * C + cur], 1);
            }
        }
    }
    if (bad) status[b] = 1;
    __syncthreads();
    int32_t* M = out + (size_t)b * C * C;
    const bool only_writer = gridDim.x == 1;
    for (int k = threadIdx.x; k < C * C; k += kThreads) {
        const int32_t n = lm[k];
        if (n) {
            if (only_writer) M[k] = n;
            else atomicAdd(&M[k], n);
        }
    }
}

}  // namespace

extern "C" int qp_core_activity(const QpCircuit* circ, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, int32_t C,
                                uint64_t* out, void* stream_) {
    // (a circuit without slices has partitions without rows: nothing to read, every count is zero)
    if (!circ || (!A && circ->T > 0) || !out) return qp::fail(QP_EINVAL, "qp_core_activity: null argument");
    if (B <= 0 || B > 65535) return qp::fail(QP_EINVAL, "qp_core_activity: B must be 1..65535 per call");
    if (C <= 0 || C > 256) return qp::fail(QP_EINVAL, "qp_core_activity: bad core count");
    if (circ->Q <= 0 || ldq < circ->Q) return qp::fail(QP_EINVAL, "qp_core_activity: bad row pitch");
    if (Tp < circ->T) return qp::fail(QP_EINVAL, "qp_core_activity: partition has %d rows, circuit has %d slices", Tp, circ->T);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)B * C * sizeof(uint64_t), stream));
    if (circ->T == 0) return QP_OK;
    if (!circ->slice_edge_ptr || (circ->nE > 0 && !circ->edges)) return qp::fail(QP_EINVAL, "qp_core_activity: circuit edge tables missing");
    // rows of up to 1024 bytes (partitions of 32 KB or more when shorter than 513): the events are the row and column sums of the core-pair counts the dense cost
    // kernel forms (one LDS atomic per moved qubit where the kernel below needs two on 256 counters)
    if (const int rc = qp::core_activity_pairs(circ, A, B, Tp, ldq, C, out, stream_); rc <= 0) return rc;
    const dim3 grid((unsigned)((circ->T + kRowsPerBlock - 1) / kRowsPerBlock), (unsigned)B);
    k_core_activity<<<grid, kThreads, 0, stream>>>(A, circ->slice_edge_ptr, circ->edges, Tp, circ->T, circ->Q, ldq, C,
                                                   reinterpret_cast<unsigned long long*>(out));
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

extern "C" int qp_comm_matrix(const QpCircuit* circ, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, int32_t C,
                              int32_t* out, int32_t* status, void* stream_) {
    if (!circ || (!A && circ->T > 0) || !out || !status) return qp::fail(QP_EINVAL, "qp_comm_matrix: null argument");
    if (B <= 0 || B > 65535) return qp::fail(QP_EINVAL, "qp_comm_matrix: B must be 1..65535 per call");
    if (C <= 0 || C > 255) return qp::fail(QP_EINVAL, "qp_comm_matrix: bad core count");
    if (circ->Q <= 0 || ldq < circ->Q) return qp::fail(QP_EINVAL, "qp_comm_matrix: bad row pitch");
    if (Tp < circ->T) return qp::fail(QP_EINVAL, "qp_comm_matrix: partition has %d rows, circuit has %d slices", Tp, circ->T);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)B * C * C * sizeof(int32_t), stream));
    HIP_TRY(hipMemsetAsync(status, 0, (size_t)B * sizeof(int32_t), stream));
    if (circ->nV == 0) return QP_OK;
    // large batches on a circuit that visits most of every row: the pair counters of the row walk are the matrix
    if (const int rc = qp::comm_matrix_pairs(circ, A, B, Tp, ldq, C, out, status, stream_); rc <= 0) return rc;
    if (!circ->visits) return qp::fail(QP_EINVAL, "qp_comm_matrix: circuit visit table missing");
    if (C <= 128) {
        // chunks of 4096 visits or more, whole partitions once those alone give every CU a few workgroups
        const int cus = qp::device_cus();
        const int64_t want_chunks = std::max<int64_t>(1, ((int64_t)cus * 4 + B - 1) / B);
        const int64_t per = std::max<int64_t>(4096, (circ->nV + want_chunks - 1) / want_chunks);
        const int32_t per_chunk = (int32_t)std::min<int64_t>(circ->nV, (per + kThreads - 1) / kThreads * kThreads);
        const dim3 lgrid((unsigned)((circ->nV + per_chunk - 1) / per_chunk), (unsigned)B);
        const size_t lds = (size_t)C * C * sizeof(int32_t);
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_comm_matrix_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_comm_matrix_lds<<<lgrid, kThreads, lds, stream>>>(A, circ->visits, circ->nV, per_chunk, Tp, ldq, C, out, status);
        HIP_TRY(hipGetLastError());
        return QP_OK;
    }
    const dim3 grid((unsigned)((circ->nV + kThreads - 1) / kThreads), (unsigned)B);
    k_comm_matrix<<<grid, kThreads, 0, stream>>>(A, circ->visits, circ->nV, Tp, ldq, C, out, status);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

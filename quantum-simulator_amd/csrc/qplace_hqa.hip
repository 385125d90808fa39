// HQA / TRHQA placement batches: the reference's per-slice control loop (HQA_placer.py:56-176,
// TRHQA_placer.py:62-228) as native host code around the device cost matrices.
//
// The loop is a recurrence in the timeslice: slice t's matrix depends on where slice t-1 left the qubits,
// and the assignment solve in between stays on the host.  So one instance is a chain of T dependent
// (launch, wait, solve, apply) steps, and throughput comes from running many chains at once: the batch is
// split over `lanes` host threads, each with its own HIP stream and its own slab of pinned host memory.
// A lane writes the live rows, op lists and core lists of its instances straight into the slab, launches
// the cost-matrix kernel on them (the kernel reads its inputs from, and writes its matrices to, that
// device-visible host memory: no copies are enqueued), polls the stream, solves its instances'
// assignments and applies them.  While one lane solves, the others' kernels run.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "qplace_internal.h"

namespace {

inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }

// Bytes of one instance's share of a lane's pinned slab: resolved ops (16 bytes each), core list, the two counts,
// the table index, the matrix.
size_t instance_bytes(int32_t ldq, int32_t C, int32_t max_ops) {
    (void)ldq;
    return (size_t)max_ops * 16 + up16((size_t)C) + 16 + up16((size_t)max_ops * C * 8);  // 16: two counts and the table index
}

struct Batch {
    int32_t kind, T, Q, ldq, C, cap, max_ops, S;
    const QpCircuit* circ;
    const double* tables;  // dev [S][C*C]
    const int32_t* tab;    // host [B] table of every instance, or null: all use table 0
    const int32_t* sep;
    const QpEdge* edges;
    const double* D;       // host [S][C*C]
    const int16_t* lookahead;  // host copy of QpCircuit.lookahead: [nE][6] partner qubits or -1
    const uint8_t* initial;
    uint8_t* out;
    int32_t rows_out;
    int device;
};

struct Lane {
    int64_t b0 = 0, b1 = 0;
    hipStream_t stream = nullptr;
    uint8_t* slab = nullptr;
    int rc = QP_OK;
    char msg[256] = "";
};

void lane_fail(Lane& ln, int rc, const char* what, const char* detail) {
    ln.rc = rc;
    snprintf(ln.msg, sizeof(ln.msg), "qp_hqa_place_batch: %s: %s", what, detail);
}

void run_lane(const Batch& bx, Lane& ln) {
    if (hipSetDevice(bx.device) != hipSuccess) return lane_fail(ln, QP_EHIP, "hipSetDevice", "failed on a worker thread");
    const int64_t nb = ln.b1 - ln.b0;
    const int32_t Q = bx.Q, C = bx.C, cap = bx.cap, T = bx.T;
    // slab layout: what the cost-matrix kernel reads and writes, sized for all of the lane's instances
    uint8_t* p = ln.slab;
    int16_t* pinfo = reinterpret_cast<int16_t*>(p);
    p += (size_t)nb * bx.max_ops * 16;
    uint8_t* pavail = p;
    p += up16((size_t)nb * C);
    int32_t* pnops = reinterpret_cast<int32_t*>(p);
    p += up16((size_t)nb * 4);
    int32_t* pnav = reinterpret_cast<int32_t*>(p);
    p += up16((size_t)nb * 4);
    int32_t* ptab = reinterpret_cast<int32_t*>(p);
    p += up16((size_t)nb * 4);
    double* pcost = reinterpret_cast<double*>(p);

    std::vector<uint8_t> a((size_t)nb * Q);
    std::vector<int64_t> occ((size_t)nb * C);
    std::vector<uint8_t> rem((size_t)nb * bx.max_ops);
    std::vector<int32_t> ops((size_t)nb * bx.max_ops);  // per launched instance: the slice-local gates of its matrix rows
    std::vector<int32_t> nrem(nb), budget(nb), spent(nb), go, rr(std::max(bx.max_ops, C)), cc(std::max(bx.max_ops, C));
    struct WorkHolder {  // freed on every way out of the lane
        qp::LsapWork* w = qp::lsap_work_new();
        ~WorkHolder() { qp::lsap_work_free(w); }
    } holder;
    qp::LsapWork* work = holder.w;
    for (int64_t k = 0; k < nb; ++k) {
        const uint8_t* src = bx.initial + (size_t)(ln.b0 + k) * Q;
        memcpy(a.data() + (size_t)k * Q, src, Q);
        memcpy(bx.out + ((size_t)(ln.b0 + k) * bx.rows_out) * Q, src, Q);
    }
    for (int32_t t = 1; t < T && ln.rc == QP_OK; ++t) {
        const int32_t e0 = bx.sep[t], E = bx.sep[t + 1] - e0;
        const QpEdge* ed = bx.edges + e0;
        bool any = false;
        for (int64_t k = 0; k < nb && E > 0; ++k) {
            const uint8_t* ak = a.data() + (size_t)k * Q;
            uint8_t* rk = rem.data() + (size_t)k * bx.max_ops;
            int32_t n = 0;
            for (int32_t e = 0; e < E; ++e) {
                rk[e] = ak[ed[e].u] != ak[ed[e].v];
                n += rk[e];
            }
            nrem[k] = n;
            budget[k] = n + 1;
            spent[k] = 0;
            if (n) {
                any = true;
                int64_t* ok = occ.data() + (size_t)k * C;
                std::fill(ok, ok + C, 0);
                for (int32_t q = 0; q < Q; ++q) ++ok[ak[q]];
            }
        }
        while (any) {
            any = false;
            go.clear();
            for (int64_t k = 0; k < nb; ++k) {
                if (nrem[k] == 0 || spent[k] >= budget[k]) continue;
                any = true;
                ++spent[k];
                const int64_t* ok = occ.data() + (size_t)k * C;
                bool room = false;
                for (int32_t c = 0; c < C && !room; ++c) room = ok[c] <= cap - 2;
                if (room) go.push_back((int32_t)k);
                else qp::make_space_row(a.data() + (size_t)k * Q, occ.data() + (size_t)k * C, Q, C, cap,
                                        bx.D + (size_t)(bx.tab ? bx.tab[ln.b0 + k] : 0) * C * C);
            }
            if (go.empty()) continue;
            const int32_t ng = (int32_t)go.size();
            for (int32_t g = 0; g < ng; ++g) {
                const int64_t k = go[g];
                const uint8_t* ak = a.data() + (size_t)k * Q;
                const uint8_t* rk = rem.data() + (size_t)k * bx.max_ops;
                int32_t* og = ops.data() + (size_t)g * E;
                int16_t* ig = pinfo + (size_t)g * E * 8;
                int32_t n = 0;
                for (int32_t e = 0; e < E; ++e) {
                    if (!rk[e]) continue;
                    // the op's row of the matrix: the cores of its look-ahead partners and of its own qubits, resolved here
                    // so that the kernel reads nothing that hangs on another load from this (host) memory
                    const int16_t* la = bx.lookahead + (size_t)(e0 + e) * 6;
                    int16_t* rec = ig + (size_t)n * 8;
                    for (int s = 0; s < 6; ++s) rec[s] = la[s] >= 0 ? (int16_t)ak[la[s]] : (int16_t)-1;
                    rec[6] = ak[ed[e].u];
                    rec[7] = ak[ed[e].v];
                    og[n++] = e;
                }
                pnops[g] = n;
                const int64_t* ok = occ.data() + (size_t)k * C;
                uint8_t* ag = pavail + (size_t)g * C;
                int32_t m = 0;
                for (int32_t c = 0; c < C; ++c)
                    if (ok[c] <= cap - 2) ag[m++] = (uint8_t)c;
                pnav[g] = m;
                ptab[g] = bx.tab ? bx.tab[ln.b0 + k] : 0;
            }
            const int rc = qp::cost_matrix_resolved(bx.kind, bx.tables, C, ptab, ng, pinfo, pnops, E, pavail, pnav, pcost, ln.stream);
            if (rc != QP_OK) return lane_fail(ln, rc, "cost matrix launch", qp_last_error());
            // wait for the matrices: poll the stream (an interrupt-driven wait costs more than the kernel)
            hipError_t st;
            unsigned spins = 0;
            while ((st = hipStreamQuery(ln.stream)) == hipErrorNotReady)
                if (++spins % 64 == 0) std::this_thread::yield();
            if (st != hipSuccess) return lane_fail(ln, QP_EHIP, "hipStreamQuery", hipGetErrorString(st));
            for (int32_t g = 0; g < ng; ++g) {
                const int64_t k = go[g];
                const double* M = pcost + (size_t)g * E * C;
                const int32_t n_ops = pnops[g], n_av = pnav[g];
                const int lrc = qp::lsap(n_ops, n_av, M, C, rr.data(), cc.data(), work);
                if (lrc != QP_OK) {  // the reference swallows the solver's exception and leaves the slice as it stands
                    spent[k] = budget[k];
                    continue;
                }
                uint8_t* ak = a.data() + (size_t)k * Q;
                int64_t* ok = occ.data() + (size_t)k * C;
                uint8_t* rk = rem.data() + (size_t)k * bx.max_ops;
                const int32_t* og = ops.data() + (size_t)g * E;
                const uint8_t* ag = pavail + (size_t)g * C;
                const int32_t pairs = std::min(n_ops, n_av);
                for (int32_t i = 0; i < pairs; ++i) {
                    if (M[(size_t)rr[i] * C + cc[i]] == std::numeric_limits<double>::infinity()) continue;
                    const int32_t e = og[rr[i]];
                    const uint8_t core = ag[cc[i]];
                    const int32_t u = ed[e].u, v = ed[e].v;
                    --ok[ak[u]];
                    --ok[ak[v]];
                    ak[u] = core;
                    ak[v] = core;
                    ok[core] += 2;
                    rk[e] = 0;
                    --nrem[k];
                }
            }
        }
        for (int64_t k = 0; k < nb; ++k)
            memcpy(bx.out + ((size_t)(ln.b0 + k) * bx.rows_out + t) * Q, a.data() + (size_t)k * Q, Q);
    }
}

}  // namespace

extern "C" size_t qp_hqa_place_pinned_bytes(int32_t ldq, int32_t C, int32_t max_slice_edges, int64_t B, int32_t lanes) {
    if (ldq <= 0 || C <= 0 || max_slice_edges < 0 || B <= 0 || lanes <= 0) return 0;
    // every array of a lane is rounded up on its own: 5 arrays of up to 15 bytes of padding per lane
    return (size_t)B * instance_bytes(ldq, C, std::max(max_slice_edges, 1)) + (size_t)lanes * 128;
}

extern "C" int qp_hqa_place_batch_multi(int32_t kind, const QpCircuit* circ, const double* tables, int32_t S, int32_t C,
                                        const int32_t* tab_index, const int32_t* slice_edge_ptr, const QpEdge* edges,
                                        const double* D, int32_t cap, const uint8_t* initial, int64_t B, uint8_t* rows_out,
                                        void* pinned, size_t pinned_bytes, int32_t lanes, void* const* streams) try {
    const char* fn = "qp_hqa_place_batch";
    if (kind != QP_CM_HQA && kind != QP_CM_TRHQA) return qp::fail(QP_EINVAL, "%s: kind must be QP_CM_HQA or QP_CM_TRHQA", fn);
    if (!circ || !tables || !slice_edge_ptr || !D || !initial || !rows_out || !pinned || !streams)
        return qp::fail(QP_EINVAL, "%s: null argument", fn);
    if (B <= 0 || lanes <= 0 || cap < 0 || S <= 0) return qp::fail(QP_EINVAL, "%s: B, S and lanes must be positive", fn);
    if (circ->Q <= 0 || circ->T < 0 || (circ->nE > 0 && !edges)) return qp::fail(QP_EINVAL, "%s: bad circuit", fn);
    if (C <= 0 || C > 256) return qp::fail(QP_EINVAL, "%s: 1..256 cores", fn);
    if (tab_index)
        for (int64_t b = 0; b < B; ++b)
            if (tab_index[b] < 0 || tab_index[b] >= S) return qp::fail(QP_EINVAL, "%s: table index of instance %lld out of range", fn, (long long)b);
    Batch bx{};
    bx.kind = kind;
    bx.T = circ->T;
    bx.Q = circ->Q;
    bx.ldq = (circ->Q + 15) / 16 * 16;
    bx.C = C;
    bx.S = S;
    bx.cap = cap;
    bx.circ = circ;
    bx.tables = tables;
    bx.tab = tab_index;
    bx.sep = slice_edge_ptr;
    bx.edges = edges;
    bx.D = D;
    bx.initial = initial;
    bx.out = rows_out;
    bx.rows_out = std::max(circ->T, 1);
    int32_t most = 1;
    for (int32_t t = 0; t < circ->T; ++t) most = std::max(most, slice_edge_ptr[t + 1] - slice_edge_ptr[t]);
    bx.max_ops = most;
    for (int64_t b = 0; b < B; ++b)
        for (int32_t q = 0; q < bx.Q; ++q)
            if (initial[(size_t)b * bx.Q + q] >= bx.C) return qp::fail(QP_EINVAL, "%s: initial holds a core index outside the table", fn);
    HIP_TRY(hipGetDevice(&bx.device));
    // the look-ahead partners of every gate, for the lanes to resolve cores on the host
    std::vector<int16_t> lookahead((size_t)std::max(circ->nE, 1) * 6, (int16_t)-1);
    if (circ->nE > 0) {
        if (!circ->lookahead) return qp::fail(QP_EINVAL, "%s: circuit look-ahead table missing", fn);
        HIP_TRY(hipMemcpy(lookahead.data(), circ->lookahead, (size_t)circ->nE * 6 * sizeof(int16_t), hipMemcpyDeviceToHost));
    }
    bx.lookahead = lookahead.data();
    lanes = (int32_t)std::min<int64_t>(lanes, B);
    const size_t per = instance_bytes(bx.ldq, bx.C, bx.max_ops);
    if (pinned_bytes < (size_t)B * per + (size_t)lanes * 128)
        return qp::fail(QP_ENOSPC, "%s: pinned buffer needs %zu bytes", fn, (size_t)B * per + (size_t)lanes * 128);
    if (reinterpret_cast<uintptr_t>(pinned) % 16 != 0) return qp::fail(QP_EINVAL, "%s: pinned buffer must be 16-byte aligned", fn);
    if (bx.T == 0) {
        memcpy(rows_out, initial, (size_t)B * bx.Q);
        return QP_OK;
    }
    std::vector<Lane> ln(lanes);
    uint8_t* slab = static_cast<uint8_t*>(pinned);
    for (int32_t l = 0; l < lanes; ++l) {
        ln[l].b0 = B * l / lanes;
        ln[l].b1 = B * (l + 1) / lanes;
        ln[l].stream = static_cast<hipStream_t>(streams[l]);
        ln[l].slab = slab;
        slab += (size_t)(ln[l].b1 - ln[l].b0) * per + 128;
    }
    // a lane whose thread cannot be started runs on this thread after the others; a lane that throws (its
    // scratch vectors) fails the call with QP_ENOMEM
    if (int rc = qp::run_workers(lanes, [&](int32_t l) { run_lane(bx, ln[l]); })) return rc;
    for (int32_t l = 0; l < lanes; ++l)
        if (ln[l].rc != QP_OK) return qp::fail(ln[l].rc, "%s", ln[l].msg);
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_hqa_place_batch: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_hqa_place_batch: unknown exception");
}

extern "C" int qp_hqa_place_batch(int32_t kind, const QpCircuit* circ, const QpDTable* dt, const int32_t* slice_edge_ptr,
                                  const QpEdge* edges, const double* D, int32_t cap, const uint8_t* initial, int64_t B,
                                  uint8_t* rows_out, void* pinned, size_t pinned_bytes, int32_t lanes,
                                  void* const* streams) {
    if (!dt) return qp::fail(QP_EINVAL, "qp_hqa_place_batch: null argument");
    if (dt->kind != QP_DT_F64 || !dt->table) return qp::fail(QP_EINVAL, "qp_hqa_place_batch: needs a QP_DT_F64 table");
    return qp_hqa_place_batch_multi(kind, circ, static_cast<const double*>(dt->table), 1, dt->C, nullptr, slice_edge_ptr, edges,
                                    D, cap, initial, B, rows_out, pinned, pinned_bytes, lanes, streams);
}

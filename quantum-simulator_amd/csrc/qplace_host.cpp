// Host-side pieces of libqplace.so: error plumbing, all-pairs distances, hop counts, the rectangular
// assignment solver and the distance-table packer.  No device code here.
#include "qplace_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <queue>
#include <sched.h>
#include <thread>
#include <utility>
#include <vector>

namespace qp {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace qp

namespace qp {

static std::atomic<int32_t> g_thread_budget{0};  // 0: derive from the affinity mask and LOCAL_WORLD_SIZE

int32_t host_threads(int32_t limit) {
    int32_t budget = g_thread_budget.load();
    if (budget <= 0) {
        cpu_set_t set;
        long cpus = 0;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        if (cpus <= 0) cpus = (long)std::max(1u, std::thread::hardware_concurrency());
        long ranks = 1;
        if (const char* env = std::getenv("LOCAL_WORLD_SIZE")) {
            char* end = nullptr;
            const long v = std::strtol(env, &end, 10);
            if (end != env && v > 0) ranks = v;
        }
        budget = (int32_t)std::max(1L, std::min(cpus / ranks, 1L << 20));
    }
    return std::max(1, std::min(limit, budget));
}

}  // namespace qp

extern "C" int32_t qp_host_threads(int32_t limit) { return qp::host_threads(limit); }
extern "C" int qp_set_host_threads(int32_t n) {
    if (n < 0) return qp::fail(QP_EINVAL, "qp_set_host_threads: n must be >= 0");
    qp::g_thread_budget.store(n);
    return QP_OK;
}

extern "C" int qp_version(void) { return QP_VERSION; }
extern "C" const char* qp_last_error(void) { return qp::g_err; }

// ------------------------------------------------------------------------------------------------
// Distances
// ------------------------------------------------------------------------------------------------

namespace {

struct Adj {
    std::vector<int32_t> ptr, nbr;
    std::vector<double> w;
};

// Neighbour lists in edge-insertion order (u gets v, v gets u, edge by edge).
int build_adj(int32_t C, int32_t nE, const int32_t* eu, const int32_t* ev, const double* w, Adj& a) {
    a.ptr.assign(C + 1, 0);
    for (int32_t e = 0; e < nE; ++e) {
        if (eu[e] < 0 || eu[e] >= C || ev[e] < 0 || ev[e] >= C)
            return qp::fail(QP_EINVAL, "edge %d endpoint out of range", e);
        a.ptr[eu[e] + 1]++;
        a.ptr[ev[e] + 1]++;
    }
    for (int32_t i = 0; i < C; ++i) a.ptr[i + 1] += a.ptr[i];
    a.nbr.assign(a.ptr[C], 0);
    a.w.assign(a.ptr[C], 0.0);
    std::vector<int32_t> fill(a.ptr.begin(), a.ptr.end() - 1);
    for (int32_t e = 0; e < nE; ++e) {
        double we = w ? w[e] : 1.0;
        a.nbr[fill[eu[e]]] = ev[e];
        a.w[fill[eu[e]]++] = we;
        a.nbr[fill[ev[e]]] = eu[e];
        a.w[fill[ev[e]]++] = we;
    }
    return QP_OK;
}

void bfs_hops(const Adj& a, int32_t C, int32_t src, std::vector<int32_t>& hop) {
    hop.assign(C, -1);
    std::vector<int32_t> frontier{src}, next;
    hop[src] = 0;
    while (!frontier.empty()) {
        next.clear();
        for (int32_t u : frontier)
            for (int32_t k = a.ptr[u]; k < a.ptr[u + 1]; ++k) {
                int32_t v = a.nbr[k];
                if (hop[v] < 0) {
                    hop[v] = hop[u] + 1;
                    next.push_back(v);
                }
            }
        frontier.swap(next);
    }
}

}  // namespace

extern "C" int qp_apsp_f64(int32_t C, int32_t nE, const int32_t* eu, const int32_t* ev, const double* w,
                           double* D) try {
    if (C <= 0 || nE < 0 || !D || (nE > 0 && (!eu || !ev || !w))) return qp::fail(QP_EINVAL, "qp_apsp_f64: bad argument");
    Adj a;
    if (int rc = build_adj(C, nE, eu, ev, w, a)) return rc;
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> best(C);
    std::vector<char> reached(C), done(C);
    std::vector<int32_t> hop;
    using Item = std::pair<double, int32_t>;
    for (int32_t s = 0; s < C; ++s) {
        // Each tentative distance is the rounded sum (distance of the settled node) + (edge weight), the
        // settled value of a node is the smallest such sum: independent of how ties are popped.
        std::fill(reached.begin(), reached.end(), 0);
        std::fill(done.begin(), done.end(), 0);
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
        best[s] = 0.0;
        reached[s] = 1;
        pq.push({0.0, s});
        while (!pq.empty()) {
            auto [d, u] = pq.top();
            pq.pop();
            if (done[u]) continue;
            done[u] = 1;
            for (int32_t k = a.ptr[u]; k < a.ptr[u + 1]; ++k) {
                int32_t v = a.nbr[k];
                if (done[v]) continue;
                double nd = d + a.w[k];
                if (!reached[v] || nd < best[v]) {
                    reached[v] = 1;
                    best[v] = nd;
                    pq.push({nd, v});
                }
            }
        }
        bool need_hops = false;
        for (int32_t j = 0; j < C; ++j) {
            if (j == s) {
                D[(size_t)s * C + j] = 0.0;
            } else if (done[j]) {
                double v = best[j];
                D[(size_t)s * C + j] = std::isnan(v) ? INF : v;
            } else {
                need_hops = true;
            }
        }
        if (need_hops) {
            bfs_hops(a, C, s, hop);
            for (int32_t j = 0; j < C; ++j)
                if (j != s && !done[j]) D[(size_t)s * C + j] = hop[j] < 0 ? INF : (double)hop[j];
        }
    }
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_apsp_f64: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_apsp_f64: unknown exception");
}

extern "C" int qp_apsp_f64_batch(int32_t C, int32_t nE, const int32_t* eu, const int32_t* ev, const double* w,
                                 int32_t S, double* D, int32_t threads) try {
    if (S < 0 || C <= 0 || nE < 0 || (S > 0 && !D) || (S > 0 && nE > 0 && (!eu || !ev || !w)))
        return qp::fail(QP_EINVAL, "qp_apsp_f64_batch: bad argument");
    for (int32_t e = 0; e < nE; ++e)
        if (eu[e] < 0 || eu[e] >= C || ev[e] < 0 || ev[e] >= C)
            return qp::fail(QP_EINVAL, "qp_apsp_f64_batch: edge %d endpoint out of range", e);
    threads = std::max(1, std::min(threads, S));
    std::vector<int> rcs(threads, QP_OK);
    auto work = [&](int32_t tid) {
        for (int32_t k = tid; k < S; k += threads) {
            const int rc = qp_apsp_f64(C, nE, eu, ev, w + (size_t)k * nE, D + (size_t)k * C * C);
            if (rc != QP_OK) rcs[tid] = rc;
        }
    };
    if (int rc = qp::run_workers(threads, work)) return rc;
    for (int rc : rcs)
        if (rc != QP_OK) return qp::fail(rc, "qp_apsp_f64_batch: a graph of the batch was rejected");
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_apsp_f64_batch: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_apsp_f64_batch: unknown exception");
}

extern "C" int qp_hops_f64(int32_t C, int32_t nE, const int32_t* eu, const int32_t* ev, double* H) try {
    if (C <= 0 || nE < 0 || !H || (nE > 0 && (!eu || !ev))) return qp::fail(QP_EINVAL, "qp_hops_f64: bad argument");
    Adj a;
    if (int rc = build_adj(C, nE, eu, ev, nullptr, a)) return rc;
    std::vector<int32_t> hop;
    for (int32_t s = 0; s < C; ++s) {
        bfs_hops(a, C, s, hop);
        for (int32_t j = 0; j < C; ++j)
            H[(size_t)s * C + j] = hop[j] < 0 ? std::numeric_limits<double>::infinity() : (double)hop[j];
    }
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_hops_f64: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_hops_f64: unknown exception");
}

// ------------------------------------------------------------------------------------------------
// Rectangular linear sum assignment (shortest augmenting paths, Crouse 2016), with the tie handling
// that makes the result equal to scipy.optimize.linear_sum_assignment's.
// ------------------------------------------------------------------------------------------------

namespace {

struct LsapScratch {
    std::vector<double> u, v, spc, tcost;
    std::vector<int32_t> path, col4row, row4col, remaining, order;
    std::vector<char> SR, SC;
};

int lsap_solve(int32_t nr, int32_t nc, const double* cost, int32_t ld, int32_t* row_out, int32_t* col_out,
               LsapScratch& s) {
    if (nr == 0 || nc == 0) return QP_OK;
    const bool transpose = nc < nr;
    const double* c = cost;
    int32_t cld = ld;
    if (transpose) {
        s.tcost.resize((size_t)nr * nc);
        for (int32_t i = 0; i < nr; ++i)
            for (int32_t j = 0; j < nc; ++j) s.tcost[(size_t)j * nr + i] = cost[(size_t)i * ld + j];
        c = s.tcost.data();
        cld = nr;
        std::swap(nr, nc);
    }
    for (int32_t i = 0; i < nr; ++i)
        for (int32_t j = 0; j < nc; ++j) {
            double x = c[(size_t)i * cld + j];
            if (std::isnan(x) || x == -std::numeric_limits<double>::infinity()) return QP_EINVAL;
        }
    const double INF = std::numeric_limits<double>::infinity();
    s.u.assign(nr, 0.0);
    s.v.assign(nc, 0.0);
    s.spc.resize(nc);
    s.path.assign(nc, -1);
    s.col4row.assign(nr, -1);
    s.row4col.assign(nc, -1);
    s.remaining.resize(nc);
    s.SR.resize(nr);
    s.SC.resize(nc);
    for (int32_t cur = 0; cur < nr; ++cur) {
        double minval = 0.0;
        int32_t i = cur, nrem = nc, sink = -1;
        for (int32_t it = 0; it < nc; ++it) s.remaining[it] = nc - it - 1;
        std::fill(s.SR.begin(), s.SR.end(), 0);
        std::fill(s.SC.begin(), s.SC.end(), 0);
        std::fill(s.spc.begin(), s.spc.end(), INF);
        while (sink == -1) {
            int32_t index = -1;
            double lowest = INF;
            s.SR[i] = 1;
            const double* ci = c + (size_t)i * cld;
            const double ui = s.u[i];
            for (int32_t it = 0; it < nrem; ++it) {
                int32_t j = s.remaining[it];
                double r = minval + ci[j] - ui - s.v[j];
                if (r < s.spc[j]) {
                    s.path[j] = i;
                    s.spc[j] = r;
                }
                // among equal tentative distances prefer a column that ends the search
                if (s.spc[j] < lowest || (s.spc[j] == lowest && s.row4col[j] == -1)) {
                    lowest = s.spc[j];
                    index = it;
                }
            }
            minval = lowest;
            if (minval == INF) return QP_EINFEASIBLE;
            int32_t j = s.remaining[index];
            if (s.row4col[j] == -1) sink = j;
            else i = s.row4col[j];
            s.SC[j] = 1;
            s.remaining[index] = s.remaining[--nrem];
        }
        s.u[cur] += minval;
        for (int32_t r = 0; r < nr; ++r)
            if (s.SR[r] && r != cur) s.u[r] += minval - s.spc[s.col4row[r]];
        for (int32_t j = 0; j < nc; ++j)
            if (s.SC[j]) s.v[j] -= minval - s.spc[j];
        int32_t j = sink;
        while (true) {
            int32_t r = s.path[j];
            s.row4col[j] = r;
            std::swap(s.col4row[r], j);
            if (r == cur) break;
        }
    }
    if (transpose) {
        s.order.resize(nr);
        for (int32_t k = 0; k < nr; ++k) s.order[k] = k;
        std::stable_sort(s.order.begin(), s.order.end(),
                         [&](int32_t a, int32_t b) { return s.col4row[a] < s.col4row[b]; });
        for (int32_t k = 0; k < nr; ++k) {
            row_out[k] = s.col4row[s.order[k]];
            col_out[k] = s.order[k];
        }
    } else {
        for (int32_t k = 0; k < nr; ++k) {
            row_out[k] = k;
            col_out[k] = s.col4row[k];
        }
    }
    return QP_OK;
}

}  // namespace

struct qp::LsapWork {
    LsapScratch s;
};
qp::LsapWork* qp::lsap_work_new() { return new qp::LsapWork(); }
void qp::lsap_work_free(qp::LsapWork* w) { delete w; }
int qp::lsap(int32_t nr, int32_t nc, const double* cost, int32_t ld, int32_t* row_out, int32_t* col_out, qp::LsapWork* work) {
    return lsap_solve(nr, nc, cost, ld, row_out, col_out, work->s);
}

extern "C" int qp_lsap_f64(int32_t nr, int32_t nc, const double* cost, int32_t* row_out, int32_t* col_out) try {
    if (nr < 0 || nc < 0 || ((nr > 0 && nc > 0) && (!cost || !row_out || !col_out)))
        return qp::fail(QP_EINVAL, "qp_lsap_f64: bad argument");
    LsapScratch s;
    int rc = lsap_solve(nr, nc, cost, nc, row_out, col_out, s);
    if (rc == QP_EINVAL) return qp::fail(rc, "qp_lsap_f64: matrix contains invalid numeric entries");
    if (rc == QP_EINFEASIBLE) return qp::fail(rc, "qp_lsap_f64: cost matrix is infeasible");
    return rc;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_lsap_f64: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_lsap_f64: unknown exception");
}

extern "C" int qp_lsap_f64_batch(int32_t n, const int32_t* nr, const int32_t* nc, const double* cost,
                                 int64_t stride, int32_t ld, int32_t* row_out, int32_t* col_out,
                                 int32_t out_stride, int32_t* status, int32_t threads) try {
    if (n < 0 || (n > 0 && (!nr || !nc || !cost || !row_out || !col_out || !status)))
        return qp::fail(QP_EINVAL, "qp_lsap_f64_batch: bad argument");
    for (int32_t k = 0; k < n; ++k)
        if (nr[k] < 0 || nc[k] < 0 || nc[k] > ld || std::min(nr[k], nc[k]) > out_stride)
            return qp::fail(QP_EINVAL, "qp_lsap_f64_batch: problem %d does not fit ld/out_stride", k);
    threads = std::max(1, std::min(threads, n));
    auto work = [&](int32_t tid) {
        LsapScratch s;
        for (int32_t k = tid; k < n; k += threads)
            status[k] = lsap_solve(nr[k], nc[k], cost + (size_t)k * stride, ld, row_out + (size_t)k * out_stride,
                                   col_out + (size_t)k * out_stride, s);
    };
    return qp::run_workers(threads, work);
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_lsap_f64_batch: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_lsap_f64_batch: unknown exception");
}

// ------------------------------------------------------------------------------------------------
// Distance-table packing
// ------------------------------------------------------------------------------------------------

extern "C" int qp_dtable_pack(const double* D, int32_t C, void* staging, size_t staging_bytes, int32_t* kind_out,
                              size_t* bytes_out) try {
    if (!D || C <= 0 || C > 256 || !staging || !kind_out || !bytes_out)
        return qp::fail(QP_EINVAL, "qp_dtable_pack: bad argument (C must be 1..256)");
    const size_t n = (size_t)C * C;
    bool integral = true;
    for (size_t k = 0; k < n && integral; ++k) {
        if (k / C == k % C) continue;
        double x = D[k];
        integral = std::isfinite(x) && x >= 0.0 && x <= 65534.0 && x == std::floor(x);
    }
    if (integral) {
        if (staging_bytes < n * 2) return qp::fail(QP_ENOSPC, "qp_dtable_pack: staging too small");
        uint16_t* o = static_cast<uint16_t*>(staging);
        for (size_t k = 0; k < n; ++k) o[k] = (k / C == k % C) ? 0 : (uint16_t)D[k];
        *kind_out = QP_DT_U16;
        *bytes_out = n * 2;
    } else {
        if (staging_bytes < n * 8) return qp::fail(QP_ENOSPC, "qp_dtable_pack: staging too small");
        double* o = static_cast<double*>(staging);
        for (size_t k = 0; k < n; ++k) o[k] = (k / C == k % C) ? 0.0 : D[k];
        *kind_out = QP_DT_F64;
        *bytes_out = n * 8;
    }
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_dtable_pack: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_dtable_pack: unknown exception");
}

extern "C" int qp_make_space(uint8_t* a, int64_t* occ, const int64_t* rows, int64_t n_rows, int32_t Q, int32_t lda,
                             int32_t C, int32_t cap, const double* D) try {
    if (!a || !occ || (n_rows > 0 && !rows) || !D || n_rows < 0 || Q <= 0 || lda < Q || C <= 0 || C > 256)
        return qp::fail(QP_EINVAL, "qp_make_space: bad arguments");
    for (int64_t k = 0; k < n_rows; ++k) {
        if (rows[k] < 0) return qp::fail(QP_EINVAL, "qp_make_space: negative row index");
        qp::make_space_row(a + (size_t)rows[k] * lda, occ + (size_t)rows[k] * C, Q, C, cap, D);
    }
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_make_space: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_make_space: unknown exception");
}

void qp::make_space_row(uint8_t* ar, int64_t* oc, int32_t Q, int32_t C, int32_t cap, const double* D) {
    const int64_t lim = (int64_t)cap - 2;
    for (int32_t core = 0; core < C; ++core) {
        if (oc[core] <= lim) continue;
        // the core's first two qubits as the row stands now (earlier cores may have sent it some)
        int32_t inside[2], n_in = 0;
        for (int32_t q = 0; q < Q && n_in < 2; ++q)
            if (ar[q] == core) inside[n_in++] = q;
        const double* drow = D + (size_t)core * C;
        for (int32_t j = 0; j < n_in; ++j) {
            int32_t best = -1;
            for (int32_t tc = 0; tc < C; ++tc) {
                if (tc == core || oc[tc] >= (int64_t)cap - 1) continue;
                // strict comparisons keep the first of equal keys, as the reference's stable sort does
                if (best < 0 || drow[tc] < drow[best] || (drow[tc] == drow[best] && oc[tc] < oc[best])) best = tc;
            }
            if (best >= 0) {
                ar[inside[j]] = (uint8_t)best;
                --oc[core];
                ++oc[best];
            }
            if (oc[core] <= lim) break;
        }
    }
}

extern "C" int qp_dtable_scaled_grid_metric(const double* D, int32_t C, double rel_tol, QpGridMetric* out, double* scale_out) try {
    if (!D || !out || !scale_out || C <= 0 || C > 256 || !(rel_tol >= 0.0))
        return qp::fail(QP_EINVAL, "qp_dtable_scaled_grid_metric: bad argument");
    *out = QpGridMetric{};
    *scale_out = 0.0;
    if (C == 1) return QP_OK;
    const size_t n = (size_t)C * C;
    double s = 0.0;
    bool integral = true;
    for (size_t k = 0; k < n; ++k) {
        if (k / C == k % C) continue;
        const double x = D[k];
        if (!(std::isfinite(x) && x > 0.0)) return QP_OK;
        if (x != std::floor(x)) integral = false;
        if (s == 0.0 || x < s) s = x;
    }
    if (integral || s <= 0.0) return QP_OK;
    std::vector<double> M(n, 0.0);
    for (size_t k = 0; k < n; ++k) {
        if (k / C == k % C) continue;
        const double m = std::nearbyint(D[k] / s);
        if (!(m >= 1.0 && m <= 65534.0) || std::fabs(D[k] - m * s) > rel_tol * D[k]) return QP_OK;
        M[k] = m;
    }
    QpGridMetric g{};
    if (int rc = qp_dtable_grid_metric(M.data(), C, &g)) return rc;
    if (g.valid) {
        *out = g;
        *scale_out = s;
    }
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_dtable_scaled_grid_metric: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_dtable_scaled_grid_metric: unknown exception");
}

extern "C" int qp_dtable_symmetric(const double* D, int32_t C, double rel_tol) try {
    if (!D || C <= 0 || !(rel_tol >= 0.0)) return qp::fail(QP_EINVAL, "qp_dtable_symmetric: bad arguments");
    for (int32_t a = 0; a < C; ++a) {
        for (int32_t b = a + 1; b < C; ++b) {
            const double x = D[(size_t)a * C + b], y = D[(size_t)b * C + a];
            if (std::isnan(x) || std::isnan(y)) return 0;
            if (std::isinf(x) || std::isinf(y)) {
                if (x != y) return 0;
                continue;
            }
            if (std::fabs(x - y) > rel_tol * std::max(std::fabs(x), std::fabs(y))) return 0;
        }
    }
    return 1;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_dtable_symmetric: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_dtable_symmetric: unknown exception");
}

extern "C" int qp_dtable_grid_metric(const double* D, int32_t C, QpGridMetric* out) try {
    if (!D || !out || C <= 0 || C > 256) return qp::fail(QP_EINVAL, "qp_dtable_grid_metric: bad argument");
    *out = QpGridMetric{};
    const size_t n = (size_t)C * C;
    for (size_t k = 0; k < n; ++k) {
        if (k / C == k % C) continue;
        const double x = D[k];
        if (!(std::isfinite(x) && x >= 0.0 && x <= 65534.0 && x == std::floor(x))) return QP_OK;
    }
    if (C == 1) return QP_OK;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    for (int xbits = 0; xbits <= 8; ++xbits) {
        const int W = 1 << xbits;
        if (xbits > 0 && (W >> 1) >= C) break;  // x already covers every core id at the previous width
        const int H = (C + W - 1) / W;
        const double wx = W > 1 ? D[1] : 0.0;                   // D[0][1]
        const double wy = C > W ? D[(size_t)W] : 0.0;           // D[0][W]
        for (int wrap = 0; wrap < 4; ++wrap) {
            const bool wrx = wrap & 1, wry = wrap & 2;
            if (wrx && !(W >= 4 && W <= 128 && C >= W)) continue;
            if (wry && !(H >= 4 && H <= 128 && pow2(H) && C == W * H)) continue;
            bool ok = true;
            for (int a = 0; a < C && ok; ++a)
                for (int b = 0; b < C; ++b) {
                    if (a == b) continue;
                    int dx = std::abs((a & (W - 1)) - (b & (W - 1))), dy = std::abs((a >> xbits) - (b >> xbits));
                    if (wrx) dx = std::min(dx, W - dx);
                    if (wry) dy = std::min(dy, H - dy);
                    if (D[(size_t)a * C + b] != wx * dx + wy * dy) {
                        ok = false;
                        break;
                    }
                }
            if (ok) {
                out->valid = 1;
                out->xbits = xbits;
                out->wx = (uint32_t)wx;
                out->wy = (uint32_t)wy;
                out->wrap_x = wrx ? W : 0;
                out->wrap_y = wry ? H : 0;
                return QP_OK;
            }
        }
    }
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_dtable_grid_metric: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_dtable_grid_metric: unknown exception");
}

// ------------------------------------------------------------------------------------------------
// Timeslicing (ASAP layering with the reference's within-slice order) and gate scheduling
// ------------------------------------------------------------------------------------------------

namespace {

// Iteration order of a CPython set that received the small non-negative ints a then b (a != b):
// 8-slot table, slot = value & 7, on collision i = (5 i + 1 + (perturb >>= 5)) & 7 with perturb = value;
// iteration walks the slots upwards.  Returns true when b comes out before a.
inline bool second_first(int64_t a, int64_t b) {
    const unsigned sa = (unsigned)(a & 7);
    unsigned i = (unsigned)(b & 7);
    uint64_t perturb = (uint64_t)b;
    while (i == sa) {
        perturb >>= 5;
        i = (unsigned)((i * 5u + 1u + perturb) & 7u);
    }
    return i < sa;
}

}  // namespace

extern "C" int qp_timeslices(int32_t n_gates, const int32_t* gq0, const int32_t* gq1, int32_t Q, int32_t* order_out,
                             int32_t* slice_ptr_out, int32_t* T_out) try {
    if (n_gates < 0 || Q <= 0 || !T_out || (n_gates > 0 && (!gq0 || !gq1 || !order_out || !slice_ptr_out)))
        return qp::fail(QP_EINVAL, "qp_timeslices: bad argument");
    std::vector<int32_t> last(Q, -1);
    std::vector<int32_t> pred((size_t)n_gates * 2, -1), succ((size_t)n_gates * 2, -1);
    for (int32_t g = 0; g < n_gates; ++g) {
        const int32_t qs[2] = {gq0[g], gq1[g]};
        if (qs[0] < 0 || qs[0] >= Q || qs[1] >= Q || qs[1] == qs[0]) return qp::fail(QP_EINVAL, "qp_timeslices: gate %d has bad qubits", g);
        for (int k = 0; k < 2; ++k) {
            const int32_t q = qs[k];
            if (q < 0) continue;
            const int32_t pg = last[q];
            if (pg >= 0) {
                int32_t* pp = &pred[(size_t)g * 2];
                if (pp[0] != pg && pp[1] != pg) pp[pp[0] < 0 ? 0 : 1] = pg;  // predecessors[g].add(pg)
                int32_t* sp = &succ[(size_t)pg * 2];
                if (sp[0] != g && sp[1] != g) sp[sp[0] < 0 ? 0 : 1] = g;      // successors[pg].add(g)
            }
            last[q] = g;
        }
    }
    std::vector<int32_t> ready, next, busy(Q, -1);
    std::vector<char> done(n_gates, 0);
    for (int32_t g = 0; g < n_gates; ++g)
        if (pred[(size_t)g * 2] < 0) ready.push_back(g);
    int32_t T = 0, n_out = 0;
    slice_ptr_out[0] = 0;
    while (!ready.empty()) {
        const int32_t begin = n_out;
        for (int32_t g : ready) {
            const int32_t a = gq0[g], b = gq1[g];
            if (busy[a] == T || (b >= 0 && busy[b] == T)) continue;  // a gate queued twice: second copy
            busy[a] = T;
            if (b >= 0) busy[b] = T;
            done[g] = 1;
            order_out[n_out++] = g;
        }
        next.clear();
        for (int32_t k = begin; k < n_out; ++k) {
            const int32_t g = order_out[k];
            int32_t s0 = succ[(size_t)g * 2], s1 = succ[(size_t)g * 2 + 1];
            if (s1 >= 0 && second_first(s0, s1)) std::swap(s0, s1);
            for (int32_t sgate : {s0, s1}) {
                if (sgate < 0 || done[sgate]) continue;
                const int32_t p0 = pred[(size_t)sgate * 2], p1 = pred[(size_t)sgate * 2 + 1];
                if (done[p0] && (p1 < 0 || done[p1])) next.push_back(sgate);
            }
        }
        ++T;
        slice_ptr_out[T] = n_out;
        ready.swap(next);
    }
    *T_out = T;
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_timeslices: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_timeslices: unknown exception");
}

namespace {

// Writes the (first, second) endpoint pairs of one bucket of gates (a slice, or a group of slices) as words
// first | second << 16, ordered so that runs of 32 consecutive gates (the lanes that share LDS cycles) have their
// endpoints in different 4-byte LDS banks where possible.  Returns the number of words written (= pairs.size()).
int32_t spread_over_banks(const std::vector<std::pair<uint16_t, uint16_t>>& pairs, uint32_t* out) {
    std::vector<int32_t> pool(pairs.size()), rest, group;
    for (size_t k = 0; k < pairs.size(); ++k) pool[k] = (int32_t)k;
    int32_t w = 0;
    while (!pool.empty()) {
        uint32_t used_u = 0, used_v = 0;
        group.clear();
        rest.clear();
        for (int32_t e : pool) {  // gates whose two endpoints fall into LDS banks nobody in the group uses
            const uint32_t mu = 1u << ((pairs[e].first >> 2) & 31), mv = 1u << ((pairs[e].second >> 2) & 31);
            if (group.size() < 32 && !(used_u & mu) && !(used_v & mv)) {
                group.push_back(e);
                used_u |= mu;
                used_v |= mv;
            } else {
                rest.push_back(e);
            }
        }
        if (group.size() < 32 && !rest.empty()) {  // then gates that clash in one endpoint only, then any
            pool.clear();
            for (int32_t e : rest) {
                const uint32_t mu = 1u << ((pairs[e].first >> 2) & 31);
                if (group.size() < 32 && !(used_u & mu)) {
                    group.push_back(e);
                    used_u |= mu;
                } else {
                    pool.push_back(e);
                }
            }
            rest.clear();
            for (int32_t e : pool) {
                if (group.size() < 32) group.push_back(e);
                else rest.push_back(e);
            }
        }
        for (int32_t e : group) out[w++] = (uint32_t)pairs[e].first | ((uint32_t)pairs[e].second << 16);
        pool.swap(rest);
    }
    return w;
}

}  // namespace

extern "C" int qp_schedule_gates(int32_t T, const int32_t* slice_edge_ptr, const uint16_t* eu, const uint16_t* ev,
                                 int32_t* sched_ptr_out, uint32_t* sched_out, int64_t sched_capacity) try {
    if (T < 0 || !slice_edge_ptr || !sched_ptr_out || !sched_out || (slice_edge_ptr[T] > 0 && (!eu || !ev)))
        return qp::fail(QP_EINVAL, "qp_schedule_gates: bad argument");
    int64_t need = QP_SCHED_TAIL;
    for (int32_t t = 0; t < T; ++t)
        need += (int64_t)(slice_edge_ptr[t + 1] - slice_edge_ptr[t] + QP_SCHED_CHUNK - 1) / QP_SCHED_CHUNK * QP_SCHED_CHUNK;
    if (sched_capacity < need || need > std::numeric_limits<int32_t>::max())
        return qp::fail(QP_ENOSPC, "qp_schedule_gates: needs room for %lld words", (long long)need);
    std::fill(sched_out, sched_out + need, 0u);
    sched_ptr_out[0] = 0;
    std::vector<std::pair<uint16_t, uint16_t>> pairs;
    for (int32_t t = 0; t < T; ++t) {
        const int32_t a = slice_edge_ptr[t], b = slice_edge_ptr[t + 1];
        pairs.clear();
        for (int32_t e = a; e < b; ++e) pairs.emplace_back(eu[e], ev[e]);
        const int32_t w = sched_ptr_out[t];
        sched_ptr_out[t + 1] = w + (b - a + QP_SCHED_CHUNK - 1) / QP_SCHED_CHUNK * QP_SCHED_CHUNK;
        spread_over_banks(pairs, sched_out + w);
    }
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_schedule_gates: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_schedule_gates: unknown exception");
}

extern "C" int32_t qp_group_rows(int32_t ldq) {
    if (ldq < 16 || ldq > 512 || ldq % 16 != 0) return 0;
    return 64 / (ldq / 16);
}

extern "C" int qp_schedule_gate_groups(int32_t T, const int32_t* slice_edge_ptr, const uint16_t* eu, const uint16_t* ev,
                                       int32_t ldq, int32_t* group_ptr_out, uint32_t* sched_out, int64_t sched_capacity) try {
    const int32_t G = qp_group_rows(ldq);
    if (T < 0 || G == 0 || !slice_edge_ptr || !group_ptr_out || !sched_out || (slice_edge_ptr[T] > 0 && (!eu || !ev)))
        return qp::fail(QP_EINVAL, "qp_schedule_gate_groups: bad argument (rows of 16 to 512 bytes, a multiple of 16)");
    const int32_t NG = (T + G - 1) / G;
    int64_t need = QP_SCHED_TAIL;
    for (int32_t j = 0; j < NG; ++j) {
        const int32_t a = slice_edge_ptr[j * G], b = slice_edge_ptr[std::min(T, (j + 1) * G)];
        need += (int64_t)(b - a + QP_SCHED_CHUNK - 1) / QP_SCHED_CHUNK * QP_SCHED_CHUNK;
    }
    if (sched_capacity < need || need > std::numeric_limits<int32_t>::max())
        return qp::fail(QP_ENOSPC, "qp_schedule_gate_groups: needs room for %lld words", (long long)need);
    std::fill(sched_out, sched_out + need, 0u);
    group_ptr_out[0] = 0;
    std::vector<std::pair<uint16_t, uint16_t>> pairs;
    for (int32_t j = 0; j < NG; ++j) {
        const int32_t t0 = j * G, t1 = std::min(T, t0 + G);
        pairs.clear();
        for (int32_t t = t0; t < t1; ++t) {
            const int32_t off = (t - t0) * ldq;  // the slice's row inside the group's strip
            for (int32_t e = slice_edge_ptr[t]; e < slice_edge_ptr[t + 1]; ++e) {
                if (eu[e] >= ldq || ev[e] >= ldq)
                    return qp::fail(QP_EINVAL, "qp_schedule_gate_groups: gate %d names a qubit beyond the row pitch %d", e, ldq);
                pairs.emplace_back((uint16_t)(off + eu[e]), (uint16_t)(off + ev[e]));
            }
        }
        const int32_t w = group_ptr_out[j];
        group_ptr_out[j + 1] = w + ((int32_t)pairs.size() + QP_SCHED_CHUNK - 1) / QP_SCHED_CHUNK * QP_SCHED_CHUNK;
        spread_over_banks(pairs, sched_out + w);
    }
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_schedule_gate_groups: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_schedule_gate_groups: unknown exception");
}

// ------------------------------------------------------------------------------------------------
// OpenQASM 2.0: runs of plain gate statements
// ------------------------------------------------------------------------------------------------

namespace {

inline bool ident_start(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
inline bool ident_char(char c) { return ident_start(c) || (c >= '0' && c <= '9'); }

// skips white space and // comments
inline int64_t skip_blank(const char* t, int64_t n, int64_t i) {
    for (;;) {
        while (i < n && (t[i] == ' ' || t[i] == '\t' || t[i] == '\n' || t[i] == '\r' || t[i] == '\f' || t[i] == '\v')) ++i;
        if (i + 1 < n && t[i] == '/' && t[i + 1] == '/') {
            while (i < n && t[i] != '\n') ++i;
            continue;
        }
        return i;
    }
}

inline bool is_keyword(const char* s, int64_t len) {
    static const char* const kw[] = {"OPENQASM", "include", "qreg", "creg", "gate", "opaque", "measure", "reset", "barrier", "if"};
    for (const char* k : kw)
        if ((int64_t)std::strlen(k) == len && std::memcmp(k, s, (size_t)len) == 0) return true;
    return false;
}

}  // namespace

extern "C" int qp_qasm_scan(const char* text, int64_t n, int64_t start, int32_t n_regs, const char* reg_names,
                            const int32_t* reg_size, const int32_t* reg_base, int64_t cap, uint64_t* name_hash,
                            int32_t* name_pos, int32_t* param_pos, int32_t* qidx, int64_t* n_out, int64_t* stop_out) try {
    if (!text || n < 0 || start < 0 || start > n || cap < 0 || !n_out || !stop_out || n > 0x7FFFFFFF ||
        (cap > 0 && (!name_hash || !name_pos || !param_pos || !qidx)) || (n_regs > 0 && (!reg_names || !reg_size || !reg_base)))
        return qp::fail(QP_EINVAL, "qp_qasm_scan: bad argument");
    std::vector<std::pair<const char*, size_t>> regs;
    {
        const char* r = reg_names;
        for (int32_t k = 0; k < n_regs; ++k) {
            const size_t len = std::strlen(r);
            regs.emplace_back(r, len);
            r += len + 1;
        }
    }
    int64_t i = start, count = 0;
    while (count < cap) {
        const int64_t st = skip_blank(text, n, i);
        int64_t j = st;
        if (j >= n || !ident_start(text[j])) {
            i = st;
            break;
        }
        while (j < n && ident_char(text[j])) ++j;
        const int64_t nb = st, nl = j - st;
        if (is_keyword(text + nb, nl)) {
            i = st;
            break;
        }
        j = skip_blank(text, n, j);
        int64_t pb = -1, pl = -1;
        if (j < n && text[j] == '(') {
            int depth = 0;
            int64_t k = j;
            for (; k < n; ++k) {
                if (text[k] == '(') ++depth;
                else if (text[k] == ')' && --depth == 0) break;
                else if (text[k] == ';') break;
            }
            if (k >= n || text[k] != ')') {
                i = st;
                break;
            }
            pb = j + 1;
            pl = k - j - 1;
            j = skip_blank(text, n, k + 1);
        }
        int32_t q[3] = {-1, -1, -1};
        int nq = 0;
        bool ok = true;
        for (;;) {
            if (j >= n || !ident_start(text[j]) || nq == 3) {
                ok = false;
                break;
            }
            int64_t k = j;
            while (k < n && ident_char(text[k])) ++k;
            int32_t reg = -1;
            for (int32_t r = 0; r < n_regs; ++r)
                if ((int64_t)regs[r].second == k - j && std::memcmp(regs[r].first, text + j, (size_t)(k - j)) == 0) reg = r;
            k = skip_blank(text, n, k);
            if (reg < 0 || k >= n || text[k] != '[') {
                ok = false;
                break;
            }
            k = skip_blank(text, n, k + 1);
            int64_t v = 0, digits = 0;
            while (k < n && text[k] >= '0' && text[k] <= '9' && digits < 9) {
                v = v * 10 + (text[k] - '0');
                ++k;
                ++digits;
            }
            k = skip_blank(text, n, k);
            if (digits == 0 || k >= n || text[k] != ']' || v >= reg_size[reg]) {
                ok = false;
                break;
            }
            q[nq++] = reg_base[reg] + (int32_t)v;
            k = skip_blank(text, n, k + 1);
            if (k < n && text[k] == ',') {
                j = skip_blank(text, n, k + 1);
                continue;
            }
            if (k < n && text[k] == ';') {
                j = k + 1;
                break;
            }
            ok = false;
            break;
        }
        if (!ok || nq == 0) {
            i = st;
            break;
        }
        uint64_t h = 1469598103934665603ull;  // FNV-1a of the gate name
        for (int64_t k = 0; k < nl; ++k) h = (h ^ (unsigned char)text[nb + k]) * 1099511628211ull;
        name_hash[count] = h;
        name_pos[2 * count] = (int32_t)nb;
        name_pos[2 * count + 1] = (int32_t)nl;
        param_pos[2 * count] = (int32_t)pb;
        param_pos[2 * count + 1] = (int32_t)pl;
        qidx[3 * count] = q[0];
        qidx[3 * count + 1] = q[1];
        qidx[3 * count + 2] = q[2];
        ++count;
        i = j;
    }
    *n_out = count;
    *stop_out = i;
    return QP_OK;
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_qasm_scan: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_qasm_scan: unknown exception");
}

// ------------------------------------------------------------------------------------------------
// RANDOM and BFS placements (Random_placer.py:25-59, BFS_placer.py:101-200): sequential per slice and
// driven by Python's global random stream.  The stream is a Mersenne Twister whose 624 words + position
// the caller hands over (random.getstate()) and takes back (random.setstate()); random.choice(seq) is
// seq[randbelow(len(seq))] with randbelow(n) = the top bit_length(n) bits of the next 32-bit output,
// redrawn until below n.
// ------------------------------------------------------------------------------------------------
namespace {

struct PyRandom {
    uint32_t* s;  // [625]: words, then the position
    explicit PyRandom(uint32_t* state) : s(state) {}
    uint32_t next32() {
        constexpr int N = 624, M = 397;
        uint32_t pos = s[N];
        if (pos >= (uint32_t)N) {
            auto mix = [](uint32_t u, uint32_t v) { return (u & 0x80000000u) | (v & 0x7fffffffu); };
            int kk = 0;
            for (; kk < N - M; ++kk) {
                const uint32_t y = mix(s[kk], s[kk + 1]);
                s[kk] = s[kk + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; kk < N - 1; ++kk) {
                const uint32_t y = mix(s[kk], s[kk + 1]);
                s[kk] = s[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            const uint32_t y = mix(s[N - 1], s[0]);
            s[N - 1] = s[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            pos = 0;
        }
        uint32_t y = s[pos++];
        s[N] = pos;
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    uint32_t below(uint32_t n) {  // n >= 1
        int k = 0;
        for (uint32_t v = n; v; v >>= 1) ++k;
        uint32_t r = next32() >> (32 - k);
        while (r >= n) r = next32() >> (32 - k);
        return r;
    }
};

}  // namespace

extern "C" int qp_place_sparse(int32_t which, int32_t T, int32_t Q, int32_t ldq, int32_t C, int32_t cap,
                               const int32_t* member_ptr, const uint16_t* members, const int32_t* cmd_ptr,
                               const int32_t* cmd_q0, const int32_t* cmd_q1, const int32_t* nb_ptr, const int32_t* nb,
                               const double* D, uint32_t* mt_states, int32_t n_states, int32_t n_rep, uint8_t* A_out) try {
    if (which != QP_PLACE_RANDOM && which != QP_PLACE_BFS) return qp::fail(QP_EINVAL, "qp_place_sparse: unknown placer %d", which);
    if (T < 0 || Q <= 0 || ldq < Q || C <= 0 || C > 255 || cap < 0 || n_rep < 0 || !member_ptr || !mt_states ||
        (n_states != 1 && n_states != n_rep) || (n_rep > 0 && !A_out) || (member_ptr[T] > 0 && !members))
        return qp::fail(QP_EINVAL, "qp_place_sparse: bad argument");
    if (which == QP_PLACE_BFS && (!cmd_ptr || !nb_ptr || !D || (cmd_ptr[T] > 0 && (!cmd_q0 || !cmd_q1)) || (nb_ptr[C] > 0 && !nb)))
        return qp::fail(QP_EINVAL, "qp_place_sparse: BFS needs the command lists, the neighbour lists and the distance matrix");
    for (int32_t k = 0; k < member_ptr[T]; ++k)
        if (members[k] >= Q) return qp::fail(QP_EINVAL, "qp_place_sparse: qubit index out of range");
    if (which == QP_PLACE_BFS) {
        for (int32_t k = 0; k < cmd_ptr[T]; ++k)
            if (cmd_q0[k] >= Q || cmd_q1[k] >= Q) return qp::fail(QP_EINVAL, "qp_place_sparse: qubit index out of range");
        for (int32_t k = 0; k < nb_ptr[C]; ++k)
            if (nb[k] < 0 || nb[k] >= C) return qp::fail(QP_EINVAL, "qp_place_sparse: core index out of range");
    }
    // Repeats that draw from one stream run one after another, as the calls they stand for would; repeats with a
    // state each are independent and are dealt to host threads.
    auto run_reps = [&](int32_t rep_begin, int32_t rep_end) {
    std::vector<int32_t> room((size_t)C), open_nodes, queue, start;
    std::vector<uint8_t> done((size_t)Q), seen((size_t)C);
    for (int32_t rep = rep_begin; rep < rep_end; ++rep) {
        PyRandom rng(mt_states + (size_t)(n_states == 1 ? 0 : rep) * 625);
        uint8_t* Ar = A_out + (size_t)rep * T * ldq;
        for (int32_t t = 0; t < T; ++t) {
            uint8_t* row = Ar + (size_t)t * ldq;
            std::memset(row, QP_ABSENT, (size_t)Q);
            std::memset(row + Q, 0, (size_t)(ldq - Q));
            std::fill(room.begin(), room.end(), cap);
            auto rebuild_open = [&]() {
                open_nodes.clear();
                for (int32_t n = 0; n < C; ++n)
                    if (room[n] > 0) open_nodes.push_back(n);
            };
            const uint16_t* mem = members + member_ptr[t];
            const int32_t n_mem = member_ptr[t + 1] - member_ptr[t];
            if (which == QP_PLACE_RANDOM) {
                rebuild_open();
                for (int32_t k = 0; k < n_mem && !open_nodes.empty(); ++k) {
                    const int32_t n = open_nodes[rng.below((uint32_t)open_nodes.size())];
                    row[mem[k]] = (uint8_t)n;
                    if (--room[n] == 0) rebuild_open();
                }
                continue;
            }
            for (int32_t k = 0; k < n_mem; ++k) done[mem[k]] = 0;
            auto place = [&](int32_t q, int32_t n) {
                row[q] = (uint8_t)n;
                --room[n];
                done[q] = 1;
            };
            for (int32_t c = cmd_ptr[t]; c < cmd_ptr[t + 1]; ++c) {
                const int32_t a = cmd_q0[c], b = cmd_q1[c];
                if (a < 0) continue;  // neither one nor two qubits: only a member of the slice
                if (b >= 0) {
                    if (done[a] || done[b]) continue;
                    int32_t home = -1;
                    for (int32_t n = 0; n < C && home < 0; ++n)
                        if (room[n] >= 2) home = n;
                    if (home >= 0) {
                        place(a, home);
                        place(b, home);
                        continue;
                    }
                    // the closest pair of neighbouring cores with a free slot each; of equal distances the first
                    // in (core, nearest-first neighbour) order, as a stable sort by distance leaves it
                    int32_t b1 = -1, b2 = -1;
                    double best = 0.0;
                    for (int32_t n1 = 0; n1 < C; ++n1) {
                        if (room[n1] < 1) continue;
                        for (int32_t j = nb_ptr[n1]; j < nb_ptr[n1 + 1]; ++j) {
                            const int32_t n2 = nb[j];
                            if (room[n2] < 1) continue;
                            const double d = D[(size_t)n1 * C + n2];
                            if (b1 < 0 || d < best) {
                                b1 = n1;
                                b2 = n2;
                                best = d;
                            }
                        }
                    }
                    if (b1 >= 0) {
                        place(a, b1);
                        place(b, b2);
                    }
                } else if (!done[a]) {
                    rebuild_open();
                    if (!open_nodes.empty()) place(a, open_nodes[rng.below((uint32_t)open_nodes.size())]);
                }
            }
            // the rest, breadth-first from the emptiest cores
            start.clear();
            for (int32_t n = 0; n < C; ++n)
                if (room[n] > 0) start.push_back(n);
            std::stable_sort(start.begin(), start.end(), [&](int32_t x, int32_t y) { return room[x] > room[y]; });
            queue.assign(start.begin(), start.end());
            std::fill(seen.begin(), seen.end(), 0);
            for (int32_t n : start) seen[n] = 1;
            size_t head = 0;
            for (int32_t k = 0; k < n_mem && head < queue.size(); ++k) {
                const int32_t q = mem[k];
                if (done[q]) continue;
                const int32_t n = queue[head++];
                place(q, n);
                for (int32_t j = nb_ptr[n]; j < nb_ptr[n + 1]; ++j) {
                    const int32_t m = nb[j];
                    if (!seen[m] && room[m] > 0) {
                        seen[m] = 1;
                        queue.push_back(m);
                    }
                }
            }
        }
    }
    };
    const int32_t workers = (n_states == n_rep && n_rep > 1) ? std::min(n_rep, qp::host_threads(16)) : 1;
    return qp::run_workers(workers, [&](int32_t w) {
        run_reps((int32_t)((int64_t)n_rep * w / workers), (int32_t)((int64_t)n_rep * (w + 1) / workers));
    });
} catch (const std::exception& e) {
    return qp::fail(QP_ENOMEM, "qp_place_sparse: %s", e.what());
} catch (...) {
    return qp::fail(QP_ENOMEM, "qp_place_sparse: unknown exception");
}

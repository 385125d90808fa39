// GTRHQA placement on the device: one workgroup owns one instance and walks all of its timeslices,
// so a batch of placements needs no host round trip per slice.  Follows GTRHQA_placer.py:62-265:
// slice t starts from row t-1 (row 0 for t = 0) and its result becomes row t+1 (two interleaved
// chains); the ops of the slice are its counted gates whose qubits sit on different cores; a core is
// available when it holds at most cap-2 qubits at the start of the slice; the op x core cost is
// GTRHQA_placer.py:99-112 (moves away from core 0 are free); assignment is the bidding of :128-254.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "qplace_internal.h"

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return qp::fail(QP_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

constexpr int kThreads = 256;
constexpr int kWave = 64;
constexpr int kWaves = kThreads / kWave;
constexpr double kBig = 1e12;  // what a non-finite entry counts as in an argmin (GTRHQA_placer.py:139)

struct GpArgs {
    const int32_t* sep;
    const QpEdge* edges;
    const double* D;
    const uint8_t* initial;
    uint8_t* out;
    int64_t B;
    int32_t T, Q, ldq, C, cap, max_ops;
    // byte offsets into dynamic LDS
    uint32_t off_rows, off_occ, off_best, off_win, off_bid, off_target, off_uv, off_a12, off_flags, off_cores;
};

struct OpFlags {
    uint8_t left, need, fin, pad;
};

__device__ inline unsigned long long dbits(double x) { return (unsigned long long)__double_as_longlong(x); }

// GTRHQA_placer.py:99-112: sum over the op's two qubits of D[core of the qubit][c], skipping a qubit
// that is on core c already or on core 0 (label 0 is falsy in the reference's test)
__device__ inline double entry(const double* D, int C, uint32_t a1, uint32_t a2, uint32_t c) {
    double cost = 0.0;
    if (a1 != 0 && a1 != c) cost += D[a1 * C + c];
    if (a2 != 0 && a2 != c) cost += D[a2 * C + c];
    return cost;
}

__global__ __launch_bounds__(kThreads) void k_gtrhqa_place(GpArgs p) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_nops, s_nav, s_nleft, s_nfree, s_flag, s_dfinite, s_wave_cnt[kWaves], s_base;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int C = p.C, Q = p.Q, ldq = p.ldq;
    uint8_t* rows = smem + p.off_rows;                                 // [2][ldq]
    uint32_t* occ = reinterpret_cast<uint32_t*>(smem + p.off_occ);     // [C]
    unsigned long long* best = reinterpret_cast<unsigned long long*>(smem + p.off_best);  // [C] by column
    uint32_t* win = reinterpret_cast<uint32_t*>(smem + p.off_win);     // [C] by column
    double* bid = reinterpret_cast<double*>(smem + p.off_bid);         // [max_ops]
    int32_t* target = reinterpret_cast<int32_t*>(smem + p.off_target); // [max_ops] column
    uint32_t* uv = reinterpret_cast<uint32_t*>(smem + p.off_uv);       // [max_ops] u | v << 16
    uint16_t* a12 = reinterpret_cast<uint16_t*>(smem + p.off_a12);     // [max_ops] a1 | a2 << 8
    OpFlags* flags = reinterpret_cast<OpFlags*>(smem + p.off_flags);   // [max_ops]
    uint8_t* avail = smem + p.off_cores;                               // [C] core id of column j
    uint8_t* cfree = avail + C;                                        // [C] by column

    // are all distances finite?  (then every entry is, and the "nothing finite left" test never fires)
    if (tid == 0) s_dfinite = 1;
    __syncthreads();
    {
        bool ok = true;
        for (int k = tid; k < C * C; k += kThreads) ok = ok && isfinite(p.D[k]);
        if (!ok) s_dfinite = 0;
    }
    __syncthreads();
    const bool dfinite = s_dfinite != 0;

    for (int64_t b = blockIdx.x; b < p.B; b += gridDim.x) {
        uint8_t* outb = p.out + (size_t)b * (p.T + 1) * ldq;
        for (int k = tid * 16; k < ldq; k += kThreads * 16) {
            uint4 v = *reinterpret_cast<const uint4*>(p.initial + (size_t)b * ldq + k);
            *reinterpret_cast<uint4*>(rows + k) = v;
            *reinterpret_cast<uint4*>(outb + k) = v;
        }
        __syncthreads();

        for (int t = 0; t < p.T; ++t) {
            uint8_t* a = rows + (size_t)((t == 0) ? 1 : ((t - 1) & 1)) * ldq;
            if (t == 0) {
                for (int k = tid * 16; k < ldq; k += kThreads * 16)
                    *reinterpret_cast<uint4*>(a + k) = *reinterpret_cast<const uint4*>(rows + k);
            }
            for (int c = tid; c < C; c += kThreads) occ[c] = 0;
            if (tid == 0) {
                s_nops = 0;
                s_nav = 0;
                s_base = 0;
            }
            __syncthreads();
            for (int q = tid; q < Q; q += kThreads) atomicAdd(&occ[a[q]], 1u);
            __syncthreads();

            // columns: cores with room for a whole op, ascending (GTRHQA_placer.py:81-86)
            {
                const bool has = tid < C && (int)occ[tid] <= p.cap - 2;
                const unsigned long long m = __ballot(has);
                if (lane == 0) s_wave_cnt[wave] = __popcll(m);
                __syncthreads();
                int base = 0;
                for (int w = 0; w < wave; ++w) base += s_wave_cnt[w];
                if (has) {
                    const int j = base + __popcll(m & ((1ull << lane) - 1ull));
                    avail[j] = (uint8_t)tid;
                    cfree[j] = 1;
                }
                if (tid == 0) {
                    int n = 0;
                    for (int w = 0; w < kWaves; ++w) n += s_wave_cnt[w];
                    s_nav = n;
                }
                __syncthreads();
            }
            // rows: the slice's counted gates whose qubits sit on different cores, slice order (:72-79)
            const int e0 = p.sep[t], e1 = p.sep[t + 1];
            for (int eb = e0; eb < e1; eb += kThreads) {
                const int e = eb + tid;
                bool has = false;
                uint32_t u = 0, v = 0, au = 0, av = 0;
                if (e < e1) {
                    const QpEdge ed = p.edges[e];
                    u = ed.u;
                    v = ed.v;
                    au = a[u];
                    av = a[v];
                    has = au != av;
                }
                const unsigned long long m = __ballot(has);
                if (lane == 0) s_wave_cnt[wave] = __popcll(m);
                __syncthreads();
                int base = s_base;
                for (int w = 0; w < wave; ++w) base += s_wave_cnt[w];
                if (has) {
                    const int k = base + __popcll(m & ((1ull << lane) - 1ull));
                    uv[k] = u | (v << 16);
                    a12[k] = (uint16_t)(au | (av << 8));
                    flags[k] = OpFlags{1, 1, 0, 0};
                }
                __syncthreads();
                if (tid == 0) {
                    int n = s_base;
                    for (int w = 0; w < kWaves; ++w) n += s_wave_cnt[w];
                    s_base = n;
                    s_nops = n;
                }
                __syncthreads();
            }
            const int nops = s_nops, nav = s_nav;
            if (nops > 0 && nav > 0) {
                if (tid == 0) {
                    s_nleft = nops;
                    s_nfree = nav;
                }
                __syncthreads();
                const int max_rounds = min(nav, nops) + 2;
                int rounds = 0;
                while (s_nleft > 0 && s_nfree > 0 && rounds < max_rounds) {
                    ++rounds;
                    if (!dfinite) {  // :135-137: stop when no open op has a finite entry on a free core
                        if (tid == 0) s_flag = 0;
                        __syncthreads();
                        for (int k = tid; k < nops; k += kThreads) {
                            if (!flags[k].left) continue;
                            const uint32_t a1 = a12[k] & 0xFF, a2 = a12[k] >> 8;
                            bool any = false;
                            for (int j = 0; j < nav && !any; ++j)
                                any = cfree[j] && isfinite(entry(p.D, C, a1, a2, avail[j]));
                            if (any) s_flag = 1;
                        }
                        __syncthreads();
                        const bool go = s_flag != 0;
                        __syncthreads();
                        if (!go) break;
                    }
                    // every open op without a standing bid looks for its cheapest free core (first minimum)
                    for (int k = wave; k < nops; k += kWaves) {
                        const OpFlags f = flags[k];
                        if (!f.left || !f.need) continue;
                        const uint32_t a1 = a12[k] & 0xFF, a2 = a12[k] >> 8;
                        double bv = 0.0;
                        int bj = 0x7FFFFFFF;
                        bool bfin = false;
                        for (int j = lane; j < nav; j += kWave) {
                            if (!cfree[j]) continue;
                            const double val = entry(p.D, C, a1, a2, avail[j]);
                            const bool fin = isfinite(val);
                            const double vm = fin ? val : kBig;
                            if (bj == 0x7FFFFFFF || vm < bv) {
                                bv = vm;
                                bj = j;
                                bfin = fin;
                            }
                        }
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) {
                            const double ov = __shfl_xor(bv, off, kWave);
                            const int oj = __shfl_xor(bj, off, kWave);
                            const int ofin = __shfl_xor((int)bfin, off, kWave);
                            if (oj != 0x7FFFFFFF && (bj == 0x7FFFFFFF || ov < bv || (ov == bv && oj < bj))) {
                                bv = ov;
                                bj = oj;
                                bfin = ofin != 0;
                            }
                        }
                        if (lane == 0) {
                            bid[k] = bv;
                            target[k] = bj;
                            flags[k] = OpFlags{1, 0, (uint8_t)bfin, 0};
                        }
                    }
                    for (int j = tid; j < nav; j += kThreads) {
                        best[j] = ~0ull;
                        win[j] = ~0u;
                    }
                    __syncthreads();
                    // a core takes its cheapest bidder, the first one in op order among equals (:150-176)
                    for (int k = tid; k < nops; k += kThreads)
                        if (flags[k].left) atomicMin(&best[target[k]], dbits(bid[k]));
                    __syncthreads();
                    for (int k = tid; k < nops; k += kThreads)
                        if (flags[k].left && dbits(bid[k]) == best[target[k]]) atomicMin(&win[target[k]], (uint32_t)k);
                    __syncthreads();
                    for (int j = tid; j < nav; j += kThreads) {
                        if (!cfree[j] || win[j] == ~0u) continue;
                        const int k = (int)win[j];
                        if (!flags[k].fin) continue;  // :178-179 a non-finite winner leaves both open
                        flags[k].left = 0;
                        cfree[j] = 0;
                        const uint32_t w = uv[k];
                        a[w & 0xFFFF] = avail[j];
                        a[w >> 16] = avail[j];
                        atomicSub(&s_nleft, 1);
                        atomicSub(&s_nfree, 1);
                    }
                    __syncthreads();
                    for (int k = tid; k < nops; k += kThreads)
                        if (flags[k].left && !cfree[target[k]]) flags[k].need = 1;
                    __syncthreads();
                }
                // :222-254 what is still open takes its cheapest remaining core, op by op
                if (s_nleft > 0 && s_nfree > 0) {
                    for (int k = 0; k < nops; ++k) {
                        if (!flags[k].left) continue;  // same value in every thread (read after a barrier)
                        if (s_nfree == 0) break;
                        if (tid == 0) {
                            s_flag = 0;
                            best[0] = ~0ull;
                            win[0] = ~0u;
                        }
                        __syncthreads();
                        const uint32_t a1 = a12[k] & 0xFF, a2 = a12[k] >> 8;
                        double vm = 0.0;
                        bool mine = false;
                        if (tid < nav && cfree[tid]) {
                            const double val = entry(p.D, C, a1, a2, avail[tid]);
                            const bool fin = isfinite(val);
                            if (fin) s_flag = 1;
                            vm = fin ? val : kBig;
                            mine = true;
                            atomicMin(&best[0], dbits(vm));
                        }
                        __syncthreads();
                        if (mine && dbits(vm) == best[0]) atomicMin(&win[0], (uint32_t)tid);
                        __syncthreads();
                        const bool any = s_flag != 0;
                        const int j = (int)win[0];
                        __syncthreads();
                        if (any && tid == 0) {
                            flags[k].left = 0;
                            cfree[j] = 0;
                            const uint32_t w = uv[k];
                            a[w & 0xFFFF] = avail[j];
                            a[w >> 16] = avail[j];
                            s_nfree -= 1;
                            s_nleft -= 1;
                        }
                        __syncthreads();
                    }
                }
            }
            __syncthreads();
            uint8_t* orow = outb + (size_t)(t + 1) * ldq;
            for (int k = tid * 16; k < ldq; k += kThreads * 16)
                *reinterpret_cast<uint4*>(orow + k) = *reinterpret_cast<const uint4*>(a + k);
            __syncthreads();
        }
    }
}

}  // namespace

extern "C" int qp_gtrhqa_place(const QpCircuit* circ, const QpDTable* dt, const uint8_t* initial, int64_t B, int32_t ldq,
                               int32_t cap, uint8_t* out, void* stream_) {
    if (!circ || !dt || !initial || !out) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: null argument");
    if (B <= 0) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: B must be positive");
    if (circ->Q <= 0 || circ->Q > 65535 || circ->T < 0) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: bad circuit sizes");
    if (ldq < circ->Q || ldq % 16 != 0) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: ldq must be >= Q and a multiple of 16");
    if (dt->kind != QP_DT_F64 || !dt->table) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: needs a QP_DT_F64 distance table");
    if (dt->C <= 0 || dt->C > 256) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: bad core count");
    if (cap < 2) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: capacity must be at least 2");
    if (circ->T > 0 && (!circ->slice_edge_ptr || (circ->nE > 0 && !circ->edges)))
        return qp::fail(QP_EINVAL, "qp_gtrhqa_place: circuit edge tables missing");
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    // the most counted gates any slice has: sizes the per-op arrays.  slice_edge_ptr lives on the device;
    // at most Q/2 gates fit a slice (a qubit is in one gate per slice)
    const int32_t max_ops = std::max(1, circ->Q / 2);
    GpArgs p{};
    p.sep = circ->slice_edge_ptr;
    p.edges = circ->edges;
    p.D = static_cast<const double*>(dt->table);
    p.initial = initial;
    p.out = out;
    p.B = B;
    p.T = circ->T;
    p.Q = circ->Q;
    p.ldq = ldq;
    p.C = dt->C;
    p.cap = cap;
    p.max_ops = max_ops;
    uint32_t off = 0;
    auto take = [&](size_t bytes) {
        const uint32_t at = off;
        off = (uint32_t)((off + bytes + 15) & ~(size_t)15);
        return at;
    };
    p.off_rows = take((size_t)2 * ldq);
    p.off_occ = take((size_t)4 * dt->C);
    p.off_best = take((size_t)8 * dt->C);
    p.off_win = take((size_t)4 * dt->C);
    p.off_bid = take((size_t)8 * max_ops);
    p.off_target = take((size_t)4 * max_ops);
    p.off_uv = take((size_t)4 * max_ops);
    p.off_a12 = take((size_t)2 * max_ops);
    p.off_flags = take((size_t)4 * max_ops);
    p.off_cores = take((size_t)2 * dt->C);
    if (off > 160u * 1024u) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: circuit too wide for the on-chip state (%u bytes)", off);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gtrhqa_place), hipFuncAttributeMaxDynamicSharedMemorySize, (int)off));
    int dev = 0, cus = 256, per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_gtrhqa_place), kThreads, off));
    const unsigned grid = (unsigned)std::min<int64_t>(B, (int64_t)std::max(cus, 1) * std::max(per_cu, 1));
    k_gtrhqa_place<<<grid, kThreads, off, stream>>>(p);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

// Device side of libqplace.so: gfx950 kernels for placement-cost evaluation and op x core cost
// matrices, plus their C-ABI launchers.  Built with -ffp-contract=off: the cost-matrix entries must
// round exactly where the reference's Python floats round.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <limits>
#include <type_traits>

#include "qplace_internal.h"

#ifndef QP_GROUPS4_AHEAD
#define QP_GROUPS4_AHEAD 1
#endif

namespace {

constexpr int kWave = 64;
constexpr int kLdsBytes = 160 * 1024;  // LDS per CU on gfx950; one workgroup may take all of it

template <int KIND>
struct Tab;
template <>
struct Tab<QP_DT_F64> {
    using elem = double;
    using acc = double;
};
template <>
struct Tab<QP_DT_U16> {
    using elem = uint16_t;
    using acc = uint32_t;  // per work item: < 2^16 terms of < 2^16 each
};

__device__ inline double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    return x;
}
__device__ inline double wave_sum(uint32_t x) {
    unsigned long long v = x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return (double)v;
}

// 16 bytes of a partition row.  Rows are read exactly once, so the load is marked non-temporal: it does
// not displace the gate lists and tables that every wave re-reads from the caches.
__device__ inline uint4 load_row16(const uint8_t* ptr) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Copies the distance table (16-byte aligned, exactly `bytes` long) into LDS, 16 bytes per thread per
// step and the last few bytes one by one, and returns the size rounded up to 16.
__device__ inline uint32_t stage_table(uint8_t* smem, const void* table, uint32_t bytes) {
    const uint32_t padded = (bytes + 15u) & ~15u;
    const uint4* src = static_cast<const uint4*>(table);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (uint32_t k = threadIdx.x; k < bytes / 16; k += blockDim.x) dst[k] = src[k];
    const uint32_t done = bytes & ~15u;
    if (threadIdx.x < padded - done)
        smem[done + threadIdx.x] = done + threadIdx.x < bytes ? static_cast<const uint8_t*>(table)[done + threadIdx.x] : uint8_t(0);
    __syncthreads();
    return padded;
}

// Same, but every row is placed at a pitch of 256 entries so that (from << 8 | to) indexes the table.
template <typename E>
__device__ inline uint32_t stage_table_pitch256(uint8_t* smem, const void* table, int C) {
    const E* src = static_cast<const E*>(table);
    E* dst = reinterpret_cast<E*>(smem);
    for (int k = threadIdx.x; k < C * 256; k += blockDim.x) {
        const int r = k >> 8, c = k & 255;
        dst[k] = c < C ? src[r * C + c] : E(0);
    }
    __syncthreads();
    return (uint32_t)(C * 256 * sizeof(E));
}

// The same at a pitch of 128 entries (up to 128 cores): rows 0 .. C - 1 only.
template <typename E>
__device__ inline uint32_t stage_table_pitch128(uint8_t* smem, const void* table, int C) {
    const E* src = static_cast<const E*>(table);
    E* dst = reinterpret_cast<E*>(smem);
    for (int k = threadIdx.x; k < C * 128; k += blockDim.x) {
        const int r = k >> 7, c = k & 127;
        dst[k] = c < C ? src[r * C + c] : E(0);
    }
    __syncthreads();
    return (uint32_t)(C * 128 * sizeof(E));
}

// Small tables, every entry stored 2^rep_shift times side by side (16-bit entries widened to 32 bits): a lane
// reads copy (lane & (copies - 1)), so the lanes of an instruction that use different copies can never meet in a
// bank, whatever entries they want.  With 32 copies (up to 16 cores) no look-up ever waits for another; with 8 or 4
// copies (up to 64 cores) only the lanes that share a copy can collide.  Index = from << BITS | to.
template <typename E, int BITS>
__device__ inline uint32_t stage_table_replicated(uint8_t* smem, const void* table, int C, int rep_shift) {
    using S = typename std::conditional<sizeof(E) == 2, uint32_t, E>::type;  // what a copy is stored as
    const E* src = static_cast<const E*>(table);
    S* dst = reinterpret_cast<S*>(smem);
    const int n = (1 << (2 * BITS)) << rep_shift;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const int idx = k >> rep_shift, r = idx >> BITS, c = idx & ((1 << BITS) - 1);
        dst[k] = (r < C && c < C) ? (S)src[r * C + c] : S(0);
    }
    __syncthreads();
    return (uint32_t)((size_t)n * sizeof(S));
}

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// The two table entries (2- or 8-byte elements, table at LDS address `base`) whose indices are the 16-bit halves
// of x: one v_mad_u32_u16 per entry forms the address from a half without unpacking it first.
template <typename E>
__device__ inline void entries_at_halves(uint32_t x, uint32_t base, E& e0, E& e1) {
    typedef __attribute__((address_space(3))) const E lds_e;
    uint32_t a0, a1;
    if constexpr (sizeof(E) == 2) {
        asm("v_mad_u32_u16 %0, %1, 2, %2 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(base));
        asm("v_mad_u32_u16 %0, %1, 2, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(base));
    } else {
        static_assert(sizeof(E) == 8, "2- or 8-byte table entries");
        asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(base));
        asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(base));
    }
    e0 = *(lds_e*)(uintptr_t)a0;
    e1 = *(lds_e*)(uintptr_t)a1;
}

// Where the distance table lives while a cost kernel runs.
constexpr int kTabLds256 = 0;  // LDS, row pitch 256 entries: index = from << 8 | to
constexpr int kTabLdsC = 1;    // LDS, row pitch C
constexpr int kTabGlobal = 2;  // device memory (L2-resident), row pitch C
constexpr int kTabLdsBand = 3; // LDS holds rows [band_lo, band_lo + band_rows) at pitch 256 plus one zero entry;
                               // pairs whose source core is outside the band read the zero entry, and the
                               // launcher runs one pass per band (for fp64 tables too big for LDS)
constexpr int kTabGrid = 4;    // no table: the distance is a grid metric computed from the two core ids
constexpr int kTabLdsRep4 = 6;  // LDS, up to 16 cores: index = from << 4 | to, every entry stored R times side by side
constexpr int kTabLdsRep6 = 7;  // LDS, up to 64 cores: index = from << 6 | to, likewise (k_cost_v3_groups)
constexpr int kTabLdsP128 = 8;  // LDS, up to 128 cores: index = from << 7 | to, one copy of rows 0 .. C - 1 (k_cost_v3_groups4, fp64)
constexpr int kTabPairs = 5;   // no look-ups: unordered core pairs are counted in LDS and multiplied by the symmetrised
                               // table once per item (fp64 tables too large for LDS; k_cost_v3_pairs, k_cost_v3_wide)

// ------------------------------------------------------------------------------------------------
// Cost V3: dense partitions, HQA-family definition.
//
// k_cost_v3 is the form for rows of at most 512 bytes when the caller gives no grouped gate list (with one,
// k_cost_v3_groups below does the same work about three times as fast).
// Work item = (timeslice chunk, partition); item id = chunk * B + b so that the workgroups running
// at one time read the same part of the circuit's edge list (it stays in L2) while each streams its
// own contiguous rows of A.  One wave owns an item: it walks the rows in groups of G = 64*16/ldq rows
// (one 16-byte load per lane), keeps the group plus the row before it in a private LDS strip
// (movement compares a 16-byte piece with the piece one row pitch earlier; edge endpoints are byte
// gathers from the strip), and looks every (core, core) pair up in the LDS-resident distance table.
// The diagonal of the table is zero, so "add only if the cores differ" needs no compare.  Waves never
// synchronise after the table is staged.
// ------------------------------------------------------------------------------------------------

constexpr int kV3Threads = 1024;
constexpr int kV3Waves = kV3Threads / kWave;
constexpr int kV3Ahead = 4;  // row groups in flight per wave beyond the one being processed
constexpr int kRowEdgeChunks = 4;  // 64-gate chunks of a row (or group of rows) that are loaded together with it

struct V3Args {
    const uint8_t* A;
    const int32_t* sep;
    const QpEdge* edges;
    const uint32_t* edges_sched;
    const int32_t* sptr;
    const uint32_t* gsched;  // grouped gate list (rows of at most 512 bytes) and the start of every group in it
    const int32_t* gptr;
    const void* table;
    double* partial;
    int64_t B, nitems;
    int32_t T, Tp, ldq, Q, C, rows_per_item, G, lpr;
    uint32_t table_bytes;
    int32_t band_lo, band_rows;
    int32_t rep_shift;  // kTabLdsRep*: log2 of the copies of every table entry (copy = lane & (copies - 1))
    unsigned long long* activity;  // k_cost_v3_pairs<., true>: per-core event counts [B][C] instead of a cost
    const uint8_t* vmask;          // k_cost_v2_pairs: [T][ldq] 0xFF where a slice visits the qubit
    const int16_t* vprev;          // k_cost_v2_pairs: [T][ldq] last slice before t that visits the qubit, -1 if none
    int32_t* comm;                 // k_comm_matrix_pairs: [B][C][C] event counts, zeroed by the caller
    int32_t* status;               // k_comm_matrix_pairs: [B], set to 1 where the reference raises
    int32_t gate_errors;           // k_cost_v2_pairs: non-zero = a counted gate with an operand on no core is an error
                                   // (V2, V4); zero = it adds nothing (V1)
    int32_t parts_per_item;  // k_cost_v3_groups: consecutive partitions a wave takes as one stream of steps (> 1 only
    int64_t nbgroups;        // when an item is a whole partition), and how many such groups of partitions there are
    // grid metric (kTabGrid): byte-replicated masks of the x and y fields of a core id, the shift that
    // brings y down, half the ring length per byte for a dimension that wraps (0 = does not wrap), weights
    uint32_t gxm, gym, gxb, ghalfx, ghalfy, gwx, gwy;
    uint32_t gyi;  // gym << gxb: the y field where it sits in a core id
};

// Sums of per-byte coordinate distances of four core pairs (from, to packed four to a dword) under a
// grid metric, accumulated with v_sad_u8.  A dimension that wraps has ring length W = 2 * half, a power
// of two <= 128: d = (from - to) mod W per byte (bit 7 guards the borrow), ring distance = half - |d - half|;
// the caller adds 4 * half per call and subtracts what accumulates here.
struct GridAcc {
    uint32_t x = 0, y = 0, calls = 0;
    __device__ inline void add4(const V3Args& p, uint32_t from, uint32_t to) {
        const uint32_t fx = from & p.gxm, tx = to & p.gxm;
        const uint32_t fy = (from >> p.gxb) & p.gym, ty = (to >> p.gxb) & p.gym;
        if (p.ghalfx) x = __builtin_amdgcn_sad_u8(((fx | 0x80808080u) - tx) & p.gxm, p.ghalfx, x);
        else x = __builtin_amdgcn_sad_u8(fx, tx, x);
        if (p.ghalfy) y = __builtin_amdgcn_sad_u8(((fy | 0x80808080u) - ty) & p.gym, p.ghalfy, y);
        else y = __builtin_amdgcn_sad_u8(fy, ty, y);
        ++calls;
    }
    // weighted total over the wave (exact: everything is an integer far below 2^53)
    __device__ inline double total(const V3Args& p) const;
};

__device__ inline double GridAcc::total(const V3Args& p) const {
    const double n = wave_sum(calls), sx = wave_sum(x), sy = wave_sum(y);
    const double dx = p.ghalfx ? n * (double)(4u * (p.ghalfx & 0xFFu)) - sx : sx;
    const double dy = p.ghalfy ? n * (double)(4u * (p.ghalfy & 0xFFu)) - sy : sy;
    return (double)p.gwx * dx + (double)p.gwy * dy;
}

template <int KIND, int TABMODE>
__global__ __launch_bounds__(kV3Threads) void k_cost_v3(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = typename Tab<KIND>::elem;
    using Acc = typename Tab<KIND>::acc;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform by construction
    uint32_t tab_off = 0;
    const E* tab;
    if constexpr (TABMODE == kTabLds256) {
        tab_off = stage_table_pitch256<E>(smem, p.table, p.C);
        tab = reinterpret_cast<const E*>(smem);
    } else if constexpr (TABMODE == kTabLdsC) {
        tab_off = stage_table(smem, p.table, p.table_bytes);
        tab = reinterpret_cast<const E*>(smem);
    } else {
        tab = static_cast<const E*>(p.table);
    }
    const int ldq = p.ldq, G = p.G, lpr = p.lpr;
    const uint32_t C = (uint32_t)p.C;
    // a table in device memory is never read past its end, whatever core ids the caller put in A (ids >= C
    // are the caller's error; LDS tables need no guard: out-of-range LDS reads return nothing harmful)
    auto in_table = [&](uint32_t idx) -> uint32_t {
        if constexpr (TABMODE == kTabGlobal) return min(idx, C * C - 1u);
        else return idx;
    };
    uint8_t* strip = smem + tab_off + (size_t)wave * (G + 1) * ldq;  // [row before][G rows]
    const int nact = G * lpr;
    const bool lane_on = lane < nact;
    const int lrow = lane / lpr;
    const int lq0 = (lane - lrow * lpr) * 16;
    const int nvalid_row = min(16, max(0, p.Q - lq0));  // qubits of this lane's piece that exist
    // byte masks of the piece's four dwords; all ones except in the last piece of a ragged row
    uint32_t keep[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int nb = min(4, max(0, nvalid_row - 4 * w));
        keep[w] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    }

    // distance of one (from, to) pair
    auto pair_at = [&](uint32_t from, uint32_t to) -> Acc {
        if constexpr (TABMODE == kTabLds256) return tab[(from << 8) | to];
        else return tab[in_table(__umul24(from, C) + to)];
    };
    // sum over the four byte lanes of (from dword, to dword)
    auto four = [&](uint32_t pf, uint32_t cf) -> Acc {
        if constexpr (TABMODE == kTabLds256) {
            const uint32_t lo = __builtin_amdgcn_perm(pf, cf, 0x05010400u);  // (p0<<8|c0) | (p1<<8|c1) << 16
            const uint32_t hi = __builtin_amdgcn_perm(pf, cf, 0x07030602u);
            return (Acc)tab[lo & 0xFFFFu] + (Acc)tab[lo >> 16] + (Acc)tab[hi & 0xFFFFu] + (Acc)tab[hi >> 16];
        } else {
            Acc s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (Acc)tab[in_table(__umul24((pf >> (8 * j)) & 0xFFu, C) + ((cf >> (8 * j)) & 0xFFu))];
            return s;
        }
    };

    for (int64_t it = (int64_t)blockIdx.x * kV3Waves + wave; it < p.nitems; it += (int64_t)gridDim.x * kV3Waves) {
        const int chunk = (int)(it / p.B);
        const int64_t b = it - (int64_t)chunk * p.B;
        const int r0 = chunk * p.rows_per_item;
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const uint8_t* base = p.A + (size_t)b * p.Tp * ldq;
        const int ngroups = (r1 - r0 + G - 1) / G;
        Acc acc = 0;
        GridAcc gacc;

        auto fetch = [&](int g) -> uint4 {
            const int rg = r0 + g * G;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (g < ngroups && lane_on && rg + lrow < r1)
                v = load_row16(base + (size_t)rg * ldq + lane * 16);
            return v;
        };

        if (r0 >= 1 && lane < lpr)
            *reinterpret_cast<uint4*>(strip + lane * 16) =
                *reinterpret_cast<const uint4*>(base + (size_t)(r0 - 1) * ldq + lane * 16);

        uint4 ahead[kV3Ahead];
#pragma unroll
        for (int k = 0; k < kV3Ahead; ++k) ahead[k] = fetch(k);

        for (int g0 = 0; g0 < ngroups; g0 += kV3Ahead) {
            uint4 now[kV3Ahead];
#pragma unroll
            for (int k = 0; k < kV3Ahead; ++k) now[k] = ahead[k];
#pragma unroll
            for (int k = 0; k < kV3Ahead; ++k) ahead[k] = fetch(g0 + kV3Ahead + k);
#pragma unroll
            for (int k = 0; k < kV3Ahead; ++k) {
                const int g = g0 + k;
                if (g >= ngroups) break;
                const int rg = r0 + g * G;
                const uint4 cur = now[k];
                uint4 prv = make_uint4(0, 0, 0, 0);
                if (lane_on) {
                    *reinterpret_cast<uint4*>(strip + ldq + lane * 16) = cur;
                    prv = *reinterpret_cast<const uint4*>(strip + lane * 16);
                }
                // movement: rows 2..T-1, previous row -> this row.  Lanes that take no part look up
                // (0, 0), which is on the zero diagonal like every qubit that stayed where it was.
                const int row = rg + lrow;
                const bool mv = lane_on && row >= 2 && row < r1;
                const uint32_t c0 = mv ? cur.x & keep[0] : 0u, p0 = mv ? prv.x & keep[0] : 0u;
                const uint32_t c1 = mv ? cur.y & keep[1] : 0u, p1 = mv ? prv.y & keep[1] : 0u;
                const uint32_t c2 = mv ? cur.z & keep[2] : 0u, p2 = mv ? prv.z & keep[2] : 0u;
                const uint32_t c3 = mv ? cur.w & keep[3] : 0u, p3 = mv ? prv.w & keep[3] : 0u;
                if constexpr (TABMODE == kTabGrid) {
                    gacc.add4(p, p0, c0);
                    gacc.add4(p, p1, c1);
                    gacc.add4(p, p2, c2);
                    gacc.add4(p, p3, c3);
                } else {
                    const bool moved = ((c0 ^ p0) | (c1 ^ p1) | (c2 ^ p2) | (c3 ^ p3)) != 0;
                    if (__builtin_amdgcn_ballot_w64(moved) != 0)
                        acc += (four(p0, c0) + four(p1, c1)) + (four(p2, c2) + four(p3, c3));
                }
                // counted two-qubit gates of these slices
                const int e0 = __builtin_amdgcn_readfirstlane(p.sep[rg]);
                const int e1 = __builtin_amdgcn_readfirstlane(p.sep[min(rg + G, r1)]);
                for (int e = e0 + lane; e < e1; e += kWave) {
                    const QpEdge ed = p.edges[e];
                    const uint8_t* rowp = strip + ldq + (ed.t - rg) * ldq;
                    if constexpr (TABMODE == kTabGrid) gacc.add4(p, rowp[ed.u], rowp[ed.v]);
                    else acc += pair_at(rowp[ed.u], rowp[ed.v]);
                }
                // the last row of the group becomes "the row before" of the next group
                if (lane_on && lrow == G - 1) *reinterpret_cast<uint4*>(strip + lq0) = cur;
            }
        }
        double s;
        if constexpr (TABMODE == kTabGrid) s = gacc.total(p);
        else s = wave_sum(acc);
        if (lane == 0) p.partial[it] = s;
    }
}

// Rows of at most 512 bytes with the grouped gate list (QpCircuit.group_sched).  As in k_cost_v3 a wave owns an
// item and takes G = 64 / lpr rows per step, one 16-byte piece per lane, parked in a private LDS strip
// [row before][G rows]; what differs is that nothing a step consumes is loaded inside the step.  Items are runs of
// whole groups of G slices (group j = slices j*G .. j*G + G - 1 whatever the item), so the gates of a step are the
// words of one group of the list: endpoints as byte offsets into the strip's G rows, padded to whole 64-gate
// chunks with zero words (one byte with itself: distance zero).  The rows of step s + AHEAD and the first
// kRowEdgeChunks chunks of its group are requested while step s is worked on, from addresses that depend on no
// loaded value, so the waits the compiler counts leave the younger loads in flight.  Items are long (up to 256
// steps; a whole partition when it is shorter) so that the start of an item, where the first rows are waited for
// with nothing else to do, is a small part of it.  TABMODE: kTabLds256, kTabLdsC, kTabGlobal, or kTabGrid (no table;
// WRAP = some dimension of the grid is a ring).
template <int KIND, int TABMODE, bool WRAP, int THREADS = kV3Threads, int AHEAD = kV3Ahead>
__global__ __launch_bounds__(THREADS, THREADS == 512 ? 6 : 4) void k_cost_v3_groups(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = typename Tab<KIND>::elem;
    using Acc = typename Tab<KIND>::acc;
    static_assert(kRowEdgeChunks == 4, "four gate chunks fill one dword of endpoints");
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    uint32_t tab_off = 0;
    const E* tab;
    if constexpr (TABMODE == kTabLds256) {
        tab_off = stage_table_pitch256<E>(smem, p.table, p.C);
        tab = reinterpret_cast<const E*>(smem);
    } else if constexpr (TABMODE == kTabLdsC) {
        tab_off = stage_table(smem, p.table, p.table_bytes);
        tab = reinterpret_cast<const E*>(smem);
    } else if constexpr (TABMODE == kTabLdsRep4) {
        tab_off = stage_table_replicated<E, 4>(smem, p.table, p.C, p.rep_shift);
        tab = reinterpret_cast<const E*>(smem);
    } else if constexpr (TABMODE == kTabLdsRep6) {
        tab_off = stage_table_replicated<E, 6>(smem, p.table, p.C, p.rep_shift);
        tab = reinterpret_cast<const E*>(smem);
    } else {
        tab = static_cast<const E*>(p.table);
    }
    constexpr bool kRep = TABMODE == kTabLdsRep4 || TABMODE == kTabLdsRep6;
    constexpr int kRepBits = TABMODE == kTabLdsRep4 ? 4 : 6;
    using RepE = typename std::conditional<sizeof(E) == 2, uint32_t, E>::type;  // a stored copy of an entry
    typedef __attribute__((address_space(3))) const RepE lds_rep;
    const int ldq = p.ldq, G = p.G, lpr = p.lpr;
    const uint32_t C = (uint32_t)p.C;
    const uint32_t tbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem);
    // replicated table: bytes between consecutive entries, and this lane's copy
    const uint32_t rep_pitch = (uint32_t)sizeof(RepE) << p.rep_shift;
    const uint32_t rep_base = tbase + (uint32_t)(lane & ((1 << p.rep_shift) - 1)) * (uint32_t)sizeof(RepE);
    auto rep_at_halves = [&](uint32_t x, RepE& e0, RepE& e1) {  // the entries whose indices are the halves of x
        uint32_t a0, a1;
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(rep_pitch), "v"(rep_base));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(rep_pitch), "v"(rep_base));
        e0 = *(lds_rep*)(uintptr_t)a0;
        e1 = *(lds_rep*)(uintptr_t)a1;
    };
    auto in_table = [&](uint32_t idx) -> uint32_t {
        if constexpr (TABMODE == kTabGlobal) return min(idx, C * C - 1u);
        else return idx;
    };
    uint8_t* strip = smem + tab_off + (size_t)wave * (G + 1) * ldq;  // [row before][G rows]
    uint8_t* rows = strip + ldq;
    const bool lane_on = lane < G * lpr;
    const int lrow = lane_on ? lane / lpr : 0;
    const int lq0 = lane_on ? (lane - lrow * lpr) * 16 : 0;
    const int lane_off = lane_on ? lane * 16 : 0;  // = lrow * ldq + lq0: the lane's piece inside a group
    const bool last_row = lane_on && lrow == G - 1;
    const int nvalid_row = lane_on ? min(16, max(0, p.Q - lq0)) : 0;
    uint32_t keep[4], kx[4], ky[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int nb = min(4, max(0, nvalid_row - 4 * w));
        keep[w] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
        kx[w] = keep[w] & p.gxm;
        ky[w] = keep[w] & p.gyi;
    }
    auto pair_at = [&](uint32_t from, uint32_t to) -> Acc {
        if constexpr (TABMODE == kTabLds256) {
            return tab[(from << 8) | to];
        } else if constexpr (kRep) {
            constexpr uint32_t m = (1u << kRepBits) - 1u;
            const uint32_t a = __umul24(((from & m) << kRepBits) | (to & m), rep_pitch) + rep_base;
            return (Acc) * (lds_rep*)(uintptr_t)a;
        } else {
            return tab[in_table(__umul24(from, C) + to)];
        }
    };
    auto four = [&](uint32_t pf, uint32_t cf) -> Acc {
        if constexpr (TABMODE == kTabLds256) {
            E e0, e1, e2, e3;
            entries_at_halves<E>(__builtin_amdgcn_perm(pf, cf, 0x05010400u), tbase, e0, e1);
            entries_at_halves<E>(__builtin_amdgcn_perm(pf, cf, 0x07030602u), tbase, e2, e3);
            return ((Acc)e0 + (Acc)e1) + ((Acc)e2 + (Acc)e3);
        } else if constexpr (TABMODE == kTabLdsRep4) {
            // core ids below 16: from << 4 | to of the four pairs is one byte each
            const uint32_t k = ((pf & 0x0F0F0F0Fu) << 4) | (cf & 0x0F0F0F0Fu);
            RepE e0, e1, e2, e3;
            rep_at_halves(__builtin_amdgcn_perm(0u, k, 0x0c010c00u), e0, e1);
            rep_at_halves(__builtin_amdgcn_perm(0u, k, 0x0c030c02u), e2, e3);
            return ((Acc)e0 + (Acc)e1) + ((Acc)e2 + (Acc)e3);
        } else if constexpr (TABMODE == kTabLdsRep6) {
            // halves from << 8 | to -> from << 6 | to (ids below 64)
            auto squeeze = [](uint32_t y) { return ((y >> 2) & 0x0FC00FC0u) | (y & 0x003F003Fu); };
            RepE e0, e1, e2, e3;
            rep_at_halves(squeeze(__builtin_amdgcn_perm(pf, cf, 0x05010400u)), e0, e1);
            rep_at_halves(squeeze(__builtin_amdgcn_perm(pf, cf, 0x07030602u)), e2, e3);
            return ((Acc)e0 + (Acc)e1) + ((Acc)e2 + (Acc)e3);
        } else {
            Acc s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (Acc)tab[in_table(__umul24((pf >> (8 * j)) & 0xFFu, C) + ((cf >> (8 * j)) & 0xFFu))];
            return s;
        }
    };
    const uint32_t* gsched_lane = p.gsched + lane;

    const size_t pstride = (size_t)p.Tp * ldq;
    for (int64_t it = (int64_t)blockIdx.x * (THREADS / kWave) + wave; it < p.nitems; it += (int64_t)gridDim.x * (THREADS / kWave)) {
        // an item: a run of whole groups of slices (rows r0 .. r1 - 1) of np consecutive partitions; np > 1 only
        // when the run is the whole circuit, and the wave then walks the partitions as one stream of steps, so
        // that the loads of a partition's first steps are in flight while the one before it is finished
        const int chunk = (int)(it / p.nbgroups);
        const int64_t b0 = (it - (int64_t)chunk * p.nbgroups) * p.parts_per_item;
        const int np = (int)min((int64_t)p.parts_per_item, p.B - b0);
        const int r0 = chunk * p.rows_per_item;  // a multiple of G
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const int nsteps = (r1 - r0 + G - 1) / G;
        const int j0 = r0 / G;
        const int N = np * nsteps;
        const uint8_t* base0 = p.A + (size_t)b0 * pstride;
        Acc acc = 0;
        uint32_t ax = 0, ay = 0;
        GridAcc gacc;

        // the next step of the stream (partition fp, step fs): the lane's piece of its rows (rows past the run's end
        // repeat its last row, never counted) and the first gate chunks of its group; past the end of the stream
        // the last step is requested again
        int fp = 0, fs = 0;
        auto fetch_next = [&](uint4& rslot, uint32_t (&gslot)[kRowEdgeChunks]) {
            const uint8_t* base = base0 + (size_t)fp * pstride;  // wave-uniform
            const int rowl = min(r0 + fs * G + lrow, r1 - 1);
            rslot = load_row16(base + ((uint32_t)rowl * (uint32_t)ldq + (uint32_t)lq0));
            const uint32_t* gp = gsched_lane + p.gptr[j0 + fs];  // wave-uniform + lane
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; ++c) gslot[c] = gp[c * kWave];
            if (fs + 1 < nsteps) {
                ++fs;
            } else if (fp + 1 < np) {
                fs = 0;
                ++fp;
            }
        };

        if (r0 >= 1 && lane < lpr)  // (np == 1 here)
            *reinterpret_cast<uint4*>(strip + lane * 16) =
                *reinterpret_cast<const uint4*>(base0 + (size_t)(r0 - 1) * ldq + lane * 16);
        uint4 ring[AHEAD];
        uint32_t gring[AHEAD][kRowEdgeChunks];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) fetch_next(ring[k], gring[k]);

        int cp = 0, s = 0;  // the step being worked on: partition cp of the item, step s
        for (int n0 = 0; n0 < N; n0 += AHEAD) {
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                if (n0 + k >= N) break;
                const uint4 cur = ring[k];
                uint32_t gates[kRowEdgeChunks];
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gates[c] = gring[k][c];
                fetch_next(ring[k], gring[k]);

                uint4 prv = make_uint4(0, 0, 0, 0);
                if (lane_on) {
                    *reinterpret_cast<uint4*>(rows + lane_off) = cur;
                    prv = *reinterpret_cast<const uint4*>(strip + lane_off);  // the same piece one row earlier
                }
                const int w0 = p.gptr[j0 + s];
                const int nch = (p.gptr[j0 + s + 1] - w0) / kWave;
                uint32_t au[kRowEdgeChunks], av[kRowEdgeChunks];
                if (nch >= kRowEdgeChunks) {
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {
                        au[c] = rows[gates[c] & 0xFFFFu];
                        av[c] = rows[gates[c] >> 16];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {  // a chunk the group lacks: one byte twice
                        const uint32_t g = c < nch ? gates[c] : 0u;
                        au[c] = rows[g & 0xFFFFu];
                        av[c] = rows[g >> 16];
                    }
                }
                // movement: rows 2 .. r1 - 1, previous row -> this row; everything else compares a piece with itself
                const int row = r0 + s * G + lrow;
                const bool mv = row >= 2 && row < r1;
                const uint32_t cw[4] = {cur.x, cur.y, cur.z, cur.w};
                const uint32_t pw[4] = {mv ? prv.x : cur.x, mv ? prv.y : cur.y, mv ? prv.z : cur.z, mv ? prv.w : cur.w};
                if constexpr (TABMODE == kTabGrid) {
                    if constexpr (WRAP) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) gacc.add4(p, pw[w] & keep[w], cw[w] & keep[w]);
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            ax = __builtin_amdgcn_sad_u8(pw[w] & kx[w], cw[w] & kx[w], ax);
                            ay = __builtin_amdgcn_sad_u8(pw[w] & ky[w], cw[w] & ky[w], ay);
                        }
                    }
                    const uint32_t U = au[0] | au[1] << 8 | au[2] << 16 | au[3] << 24;
                    const uint32_t V = av[0] | av[1] << 8 | av[2] << 16 | av[3] << 24;
                    if constexpr (WRAP) {
                        gacc.add4(p, U, V);
                    } else {
                        ax = __builtin_amdgcn_sad_u8(U & p.gxm, V & p.gxm, ax);
                        ay = __builtin_amdgcn_sad_u8(U & p.gyi, V & p.gyi, ay);
                    }
                } else {
                    if constexpr (TABMODE == kTabLdsRep4) {
                        // look-ups that never collide: all sixteen go out together (no wait per dword) unless
                        // nothing moved in the whole step
                        uint32_t pm[4], cm[4];
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            pm[w] = pw[w] & keep[w];
                            cm[w] = cw[w] & keep[w];
                        }
                        const bool moved = ((pm[0] ^ cm[0]) | (pm[1] ^ cm[1]) | (pm[2] ^ cm[2]) | (pm[3] ^ cm[3])) != 0;
                        if (__builtin_amdgcn_ballot_w64(moved) != 0)
                            acc += (four(pm[0], cm[0]) + four(pm[1], cm[1])) + (four(pm[2], cm[2]) + four(pm[3], cm[3]));
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {  // a lane skips four qubits that all stayed where they were
                            const uint32_t pm = pw[w] & keep[w], cm = cw[w] & keep[w];
                            if (pm != cm) acc += four(pm, cm);
                        }
                    }
                    // a lane whose four gates all have both qubits on one core (what a placer works for) looks
                    // nothing up
                    if (((au[0] ^ av[0]) | (au[1] ^ av[1]) | (au[2] ^ av[2]) | (au[3] ^ av[3])) != 0) {
                        Acc sg = 0;
#pragma unroll
                        for (int c = 0; c < kRowEdgeChunks; ++c) sg += pair_at(au[c], av[c]);
                        acc += sg;
                    }
                }
                for (int c = kRowEdgeChunks; c < nch; ++c) {
                    const uint32_t g = gsched_lane[w0 + c * kWave];
                    const uint32_t U = rows[g & 0xFFFFu], V = rows[g >> 16];
                    if constexpr (TABMODE != kTabGrid) {
                        acc += pair_at(U, V);
                    } else if constexpr (WRAP) {
                        gacc.add4(p, U, V);
                    } else {
                        ax = __builtin_amdgcn_sad_u8(U & p.gxm, V & p.gxm, ax);
                        ay = __builtin_amdgcn_sad_u8(U & p.gyi, V & p.gyi, ay);
                    }
                }
                // the group's last row is "the row before" of the next step
                if (last_row) *reinterpret_cast<uint4*>(strip + lq0) = cur;
                if (s + 1 < nsteps) {
                    ++s;
                } else {  // the partition's run is done: its sum, and a fresh start for the next partition
                    double sum;
                    if constexpr (TABMODE != kTabGrid) sum = wave_sum(acc);
                    else if constexpr (WRAP) sum = gacc.total(p);
                    else sum = (double)p.gwx * wave_sum(ax) + (double)p.gwy * (wave_sum(ay) / (double)(1u << p.gxb));
                    if (lane == 0) p.partial[(size_t)chunk * p.B + b0 + cp] = sum;
                    acc = 0;
                    ax = ay = 0;
                    gacc = GridAcc();
                    s = 0;
                    ++cp;
                }
            }
        }
    }
}

// The table forms of k_cost_v3_groups, four partitions per wave step.  The gates of a group of slices are the same for
// every partition, so a wave takes the group's rows of four consecutive partitions in one step (a lane: its
// 16-byte piece of each) and parks them interleaved: strip word (row, q) = the cores of qubit q in the four
// partitions.  One ds_read_b32 then fetches a gate endpoint for all four (two gathers per step and partition
// instead of eight), and a pair of words, (row before, row) of a qubit or (first, second endpoint) of a gate, is
// four look-ups, one partition per byte lane, each into that partition's own sum.  A qubit that moved in none of
// the four partitions, and a gate whose qubits share a core in all four, look nothing up.
// Strip of a wave: [row before][G rows] of words; a lane's sixteen words go out as four 16-byte writes whose
// quarters are permuted between lane pairs (word index i sits at i ^ (((i >> 5) & 3) << 2), indices counted from
// the first of the G rows, negative in the row before), so that the eight lanes of an LDS pass hit all 32 banks.
// Work: the steps of all quads form one stream, cut into runs the waves take in turn (see the launcher); a run that
// starts inside a partition loads the row before its first, one that ends inside writes a partial sum, and
// k_sum_parts adds a quad's parts in run order.  A last quad that is not full repeats the batch's last partition and
// drops the copies.  Sixteen waves per CU (two workgroups, 128 registers): fewer waves and deeper prefetch both lose.
constexpr int kGroups4Threads = 512;
constexpr int kGroups4Ahead = QP_GROUPS4_AHEAD;

template <int KIND, int TABMODE, int AHEAD, bool ONE>
__global__ __launch_bounds__(kGroups4Threads, 4) void k_cost_v3_groups4(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = typename Tab<KIND>::elem;
    using Acc = typename Tab<KIND>::acc;
    static_assert(kRowEdgeChunks == 4, "four gate chunks per step are prefetched");
    static_assert(TABMODE == kTabLds256 || TABMODE == kTabLdsRep4 || TABMODE == kTabLdsRep6 || TABMODE == kTabLdsP128,
                  "LDS tables indexed by packed core pairs");
    static_assert(TABMODE != kTabLdsP128 || KIND == QP_DT_F64, "a 16-bit table of up to 128 cores is no larger at a pitch of 256 entries");
    constexpr int kWaves = kGroups4Threads / kWave;
    constexpr uint32_t kIdBits = TABMODE == kTabLdsRep4 ? 0x0F0F0F0Fu : TABMODE == kTabLdsRep6 ? 0x3F3F3F3Fu : TABMODE == kTabLdsP128 ? 0x7F7F7F7Fu : 0xFFFFFFFFu;
    constexpr int kIdShift = TABMODE == kTabLdsRep6 ? 2 : TABMODE == kTabLdsP128 ? 1 : 0;  // ids are stored times 1 << kIdShift
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    uint32_t tab_off;
    if constexpr (TABMODE == kTabLds256) tab_off = stage_table_pitch256<E>(smem, p.table, p.C);
    else if constexpr (TABMODE == kTabLdsRep4) tab_off = stage_table_replicated<E, 4>(smem, p.table, p.C, p.rep_shift);
    else if constexpr (TABMODE == kTabLdsP128) tab_off = stage_table_pitch128<E>(smem, p.table, p.C);
    else tab_off = stage_table_replicated<E, 6>(smem, p.table, p.C, p.rep_shift);
    using RepE = typename std::conditional<sizeof(E) == 2, uint32_t, E>::type;  // a stored copy of an entry
    typedef __attribute__((address_space(3))) const RepE lds_rep;
    const int ldq = p.ldq, G = p.G, lpr = p.lpr;
    const uint32_t tbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem);
    const uint32_t rep_pitch = ((uint32_t)sizeof(RepE) << p.rep_shift) >> kIdShift;  // bytes per unit of a half
    const uint32_t rep_base = tbase + (uint32_t)(lane & ((1 << p.rep_shift) - 1)) * (uint32_t)sizeof(RepE);
    auto rep_at_halves = [&](uint32_t x, RepE& e0, RepE& e1) {  // the entries whose indices are the halves of x
        uint32_t a0, a1;
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(rep_pitch), "v"(rep_base));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(rep_pitch), "v"(rep_base));
        e0 = *(lds_rep*)(uintptr_t)a0;
        e1 = *(lds_rep*)(uintptr_t)a1;
    };
    // ONE (rows of 513 to 1024 bytes, one row per step): a lane's piece of the row before is what the lane itself held a
    // step earlier, so it stays in registers and the strip is the one row
    const uint32_t before_words = ONE ? 0u : (uint32_t)ldq;
    const uint32_t strip_words = (uint32_t)G * (uint32_t)ldq + before_words;
    uint32_t* rows = reinterpret_cast<uint32_t*>(smem + tab_off) + (size_t)wave * strip_words + before_words;  // word (row, q) of the step
    const uint32_t rbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem + tab_off + ((uint32_t)wave * strip_words + before_words) * 4u);
    const bool lane_on = lane < G * lpr;
    const int lrow = lane_on ? lane / lpr : 0;
    const int lq0 = lane_on ? (lane - lrow * lpr) * 16 : 0;
    const int lane_off = lane_on ? lane * 16 : 0;  // = lrow * ldq + lq0: word index of the lane's piece
    const bool last_row = lane_on && lrow == G - 1;
    const int nvalid = lane_on ? min(16, max(0, p.Q - lq0)) : 0;  // qubits of the piece that exist
    // Core ids as the strip holds them: bytes past the row's last qubit cleared (they then never differ between rows
    // and look up the zero diagonal), ids cut to the table's range, and for tables of up to 64 cores ids times four:
    // the halves (from / 4) << 8 | to of a packed pair are then (from << 6 | to) << 2, a multiple of the entry index, and
    // one v_mad_u32_u16 turns a half into an address (up to 128 cores: ids times two, (from << 7 | to) << 1).
    uint32_t keep[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int nb = min(4, max(0, nvalid - 4 * w));
        keep[w] = ((nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u)) & kIdBits) << kIdShift;
    }
    auto ids = [&](const uint4& v) { return make_uint4((v.x << kIdShift) & keep[0], (v.y << kIdShift) & keep[1], (v.z << kIdShift) & keep[2], (v.w << kIdShift) & keep[3]); };
    auto quarter = [](int piece, int w) { return piece + 4 * (w ^ ((piece >> 5) & 3)); };  // where words 4w .. 4w + 3 of a piece sit

    const uint32_t* gsched_lane = p.gsched + lane;
    const size_t pstride = (size_t)p.Tp * ldq;
    // The stream of steps (quad 0 step 0, quad 0 step 1, ..., quad 1 step 0, ...) is cut into p.nbgroups runs of equal
    // length (to within one unit of `gran` steps); the waves of the launch take the runs in turn.
    const int ns = (p.T + G - 1) / G;  // steps of a partition
    const int gran = p.parts_per_item;  // 1, or ns: runs are then whole quads
    const int64_t R = p.nbgroups;
    for (int64_t run = (int64_t)blockIdx.x * kWaves + wave; run < R; run += (int64_t)gridDim.x * kWaves) {
    const int64_t n_begin = (run * p.nitems / R) * gran, n_end = ((run + 1) * p.nitems / R) * gran;
    if (n_begin >= n_end) continue;
    // the part of quad q this run sums is number (run - first run with a step of q) of the quad's parts
    auto part_of = [&](int64_t q) -> int64_t {
        if (gran != 1) return 0;
        return run - (((q * ns + 1) * R - 1) / p.nitems);
    };
    const int64_t N = n_end - n_begin;
    int64_t cq = n_begin / ns;            // the step being worked on: quad cq, step s
    int s = (int)(n_begin - cq * ns);
    Acc acc[4] = {0, 0, 0, 0};
    double tot[4] = {0.0, 0.0, 0.0, 0.0};  // 16-bit tables: what acc held when it was last emptied (every 256 steps)
    int since = 0;

    // one look-up per partition: the byte lanes of (from, to) are the partitions
    auto look4 = [&](uint32_t from, uint32_t to) {
        if constexpr (TABMODE == kTabLds256) {
            E e0, e1, e2, e3;
            entries_at_halves<E>(__builtin_amdgcn_perm(from, to, 0x05010400u), tbase, e0, e1);
            entries_at_halves<E>(__builtin_amdgcn_perm(from, to, 0x07030602u), tbase, e2, e3);
            acc[0] += (Acc)e0;
            acc[1] += (Acc)e1;
            acc[2] += (Acc)e2;
            acc[3] += (Acc)e3;
        } else {
            RepE e0, e1, e2, e3;
            if constexpr (TABMODE == kTabLdsRep4) {
                const uint32_t k = (from << 4) | to;  // from << 4 | to, a byte each
                rep_at_halves(__builtin_amdgcn_perm(0u, k, 0x0c010c00u), e0, e1);
                rep_at_halves(__builtin_amdgcn_perm(0u, k, 0x0c030c02u), e2, e3);
            } else {  // ids times 2^kIdShift: with the source ids shifted back a half is that many times the entry index
                const uint32_t f = from >> kIdShift;
                rep_at_halves(__builtin_amdgcn_perm(f, to, 0x05010400u), e0, e1);
                rep_at_halves(__builtin_amdgcn_perm(f, to, 0x07030602u), e2, e3);
            }
            acc[0] += (Acc)e0;
            acc[1] += (Acc)e1;
            acc[2] += (Acc)e2;
            acc[3] += (Acc)e3;
        }
    };
    // word w of the four partitions' pieces -> the words of qubits 4w .. 4w + 3 (4 x 4 byte transposes)
    auto interleave = [&](const uint4 (&x)[4], uint32_t (&o)[16]) {
        const uint32_t xw[4][4] = {{x[0].x, x[0].y, x[0].z, x[0].w}, {x[1].x, x[1].y, x[1].z, x[1].w},
                                   {x[2].x, x[2].y, x[2].z, x[2].w}, {x[3].x, x[3].y, x[3].z, x[3].w}};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t t01a = __builtin_amdgcn_perm(xw[1][w], xw[0][w], 0x05010400u);  // a0 b0 a1 b1
            const uint32_t t01b = __builtin_amdgcn_perm(xw[1][w], xw[0][w], 0x07030602u);  // a2 b2 a3 b3
            const uint32_t t23a = __builtin_amdgcn_perm(xw[3][w], xw[2][w], 0x05010400u);  // c0 d0 c1 d1
            const uint32_t t23b = __builtin_amdgcn_perm(xw[3][w], xw[2][w], 0x07030602u);
            o[4 * w + 0] = __builtin_amdgcn_perm(t23a, t01a, 0x05040100u);  // a0 b0 c0 d0
            o[4 * w + 1] = __builtin_amdgcn_perm(t23a, t01a, 0x07060302u);
            o[4 * w + 2] = __builtin_amdgcn_perm(t23b, t01b, 0x05040100u);
            o[4 * w + 3] = __builtin_amdgcn_perm(t23b, t01b, 0x07060302u);
        }
    };
    auto park = [&](const uint32_t (&o)[16], int piece) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
            *reinterpret_cast<uint4*>(rows + quarter(piece, w)) = make_uint4(o[4 * w], o[4 * w + 1], o[4 * w + 2], o[4 * w + 3]);
    };
    // both endpoints of a packed gate word, for the four partitions
    auto endpoints = [&](uint32_t g, uint32_t& u4, uint32_t& v4) {
        const uint32_t x = g ^ ((g >> 3) & 0x000C000Cu);
        uint32_t a0, a1;
        asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(rbase));
        asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(rbase));
        u4 = *(lds_u32*)(uintptr_t)a0;
        v4 = *(lds_u32*)(uintptr_t)a1;
    };
    auto part_base = [&](int64_t q, int g) { return p.A + (size_t)min(4 * q + g, p.B - 1) * pstride; };  // wave-uniform

    // the next step of the stream (quad fq, step fs): the lane's piece of its rows in the four partitions (rows
    // past the circuit's end repeat its last row, never counted) and the first gate chunks of its group; past the
    // end of this wave's part the last step is requested again
    int64_t fq = cq, fleft = N;
    int fs = s;
    auto fetch_next = [&](uint4 (&rslot)[4], uint32_t (&gslot)[kRowEdgeChunks]) {
        const int rowl = min(fs * G + lrow, p.T - 1);
        const uint32_t off = (uint32_t)rowl * (uint32_t)ldq + (uint32_t)lq0;
#pragma unroll
        for (int g = 0; g < 4; ++g) rslot[g] = load_row16(part_base(fq, g) + off);  // (raw: ids() is applied when the step is worked on)
        const uint32_t* gp = gsched_lane + p.gptr[fs];
#pragma unroll
        for (int c = 0; c < kRowEdgeChunks; ++c) gslot[c] = gp[c * kWave];
        if (fleft > 1) {
            --fleft;
            if (++fs == ns) {
                fs = 0;
                ++fq;
            }
        }
    };

    uint32_t last[ONE ? 16 : 1] = {};  // ONE: the lane's piece of the row before, as the strip would hold it
    if (s >= 1 && (ONE ? lane_on : lane < lpr)) {  // the part starts inside a partition: the row before its first
        uint4 x[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            x[g] = ids(*reinterpret_cast<const uint4*>(part_base(cq, g) + (size_t)(s * G - 1) * ldq + lane * 16));
        uint32_t o[16];
        interleave(x, o);
        if constexpr (ONE) {
#pragma unroll
            for (int j = 0; j < 16; ++j) last[j] = o[j];
        } else {
            park(o, lane * 16 - ldq);
        }
    }
    uint4 ring[AHEAD][4];
    uint32_t gring[AHEAD][kRowEdgeChunks];
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) fetch_next(ring[k], gring[k]);

    for (int64_t n0 = 0; n0 < N; n0 += AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            if (n0 + k >= N) break;
            uint32_t cur[16], gates[kRowEdgeChunks];
            {
                const uint4 x[4] = {ids(ring[k][0]), ids(ring[k][1]), ids(ring[k][2]), ids(ring[k][3])};
                interleave(x, cur);
            }
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; ++c) gates[c] = gring[k][c];
            fetch_next(ring[k], gring[k]);

            if (lane_on) park(cur, lane_off);
            const int w0 = p.gptr[s];
            const int nch = (p.gptr[s + 1] - w0) / kWave;
            uint32_t U[kRowEdgeChunks], V[kRowEdgeChunks];
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; ++c) endpoints(c < nch ? gates[c] : 0u, U[c], V[c]);  // a chunk the group lacks: one word twice
            // movement: rows 2 .. T - 1, row before -> this row, two qubits at a time (the look-ups of a pair
            // are in flight together; a qubit that stayed looks up the zero diagonal)
            const int row = s * G + lrow;
            const bool mv = lane_on && row >= 2 && row < p.T;
            uint32_t prv[16];
            if constexpr (ONE) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    prv[j] = last[j];
                    last[j] = cur[j];
                }
            } else {
                const int before = lane_off - ldq;  // the same piece one row earlier
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint4 v = *reinterpret_cast<const uint4*>(rows + quarter(before, w));
                    prv[4 * w] = v.x;
                    prv[4 * w + 1] = v.y;
                    prv[4 * w + 2] = v.z;
                    prv[4 * w + 3] = v.w;
                }
                // the group's last row is "the row before" of the next step (after every lane has read its own)
                if (last_row) park(cur, lq0 - ldq);
            }
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                if (mv && ((prv[j] ^ cur[j]) | (prv[j + 1] ^ cur[j + 1])) != 0) {
                    look4(prv[j], cur[j]);
                    look4(prv[j + 1], cur[j + 1]);
                }
            }
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; c += 2) {
                if (((U[c] ^ V[c]) | (U[c + 1] ^ V[c + 1])) != 0) {
                    look4(U[c], V[c]);
                    look4(U[c + 1], V[c + 1]);
                }
            }
            for (int c = kRowEdgeChunks; c < nch; ++c) {
                uint32_t u4, v4;
                endpoints(gsched_lane[w0 + c * kWave], u4, v4);
                if (u4 != v4) look4(u4, v4);
            }
            if constexpr (KIND == QP_DT_U16) {  // a lane's 32-bit sum holds 256 steps of 16-bit entries
                if (++since == 256) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        tot[g] += (double)acc[g];
                        acc[g] = 0;
                    }
                    since = 0;
                }
            }
            const bool part_done = n0 + k + 1 == N;
            if (s + 1 < ns && !part_done) {
                ++s;
            } else {  // the quad (or this wave's part of it) is done: its four sums, and a fresh start
                const int64_t slot = part_of(cq);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    double sum;
                    if constexpr (KIND == QP_DT_U16) sum = wave_sum(tot[g] + (double)acc[g]);
                    else sum = wave_sum(acc[g]);
                    const int64_t b = 4 * cq + g;
                    if (lane == 0 && b < p.B) p.partial[(size_t)slot * p.B + b] = sum;
                    acc[g] = 0;
                    tot[g] = 0.0;
                }
                since = 0;
                s = 0;
                ++cq;
            }
        }
    }
    }  // runs
}

// out[b] = the sum of the parts k_cost_v3_groups4 wrote for partition b's quad, in the order of the waves that wrote
// them: the quad's steps n .. n + ns - 1 of the stream lie in runs first .. last (W runs, `units` steps).
__global__ void k_sum_parts(const double* partial, int64_t B, int64_t W, int64_t units, int32_t ns, int32_t gran, double* out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t nparts = 1;
    if (gran == 1) {
        const int64_t q = b / 4;
        const int64_t first = ((q * ns + 1) * W - 1) / units, last = (((q + 1) * ns) * W - 1) / units;
        nparts = last - first + 1;
    }
    double s = 0.0;
    for (int64_t c = 0; c < nparts; ++c) s += partial[(size_t)c * B + b];
    out[b] = s;
}

// Rows longer than 512 bytes (one row per wave step).  Differences from k_cost_v3: the previous row
// stays in registers (no LDS round trip), the LDS strip holds only the current row for the edge
// endpoint gathers, and each row's first kRowEdgeChunks*64 gates are loaded together with the row,
// kV3Ahead rows ahead, so that vector-memory results are consumed in the order they were requested.
// Gates come from the scheduled list (u | v << 16, every slice padded to whole 64-gate chunks with
// zero words, readable a few chunks past its end), so chunks are read and used without lane masks.

template <int KIND, int TABMODE, int AHEAD = kV3Ahead>
__global__ __launch_bounds__(kV3Threads) void k_cost_v3_row(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = typename Tab<KIND>::elem;
    using Acc = typename Tab<KIND>::acc;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform by construction
    uint32_t tab_off = 0;
    const E* tab;
    if constexpr (TABMODE == kTabLds256) {
        tab_off = stage_table_pitch256<E>(smem, p.table, p.C);
        tab = reinterpret_cast<const E*>(smem);
    } else if constexpr (TABMODE == kTabLdsC) {
        tab_off = stage_table(smem, p.table, p.table_bytes);
        tab = reinterpret_cast<const E*>(smem);
    } else if constexpr (TABMODE == kTabLdsBand) {
        const E* src = static_cast<const E*>(p.table) + (size_t)p.band_lo * p.C;
        E* dst = reinterpret_cast<E*>(smem);
        for (int k = threadIdx.x; k <= p.band_rows * 256; k += blockDim.x) {
            const int r = k >> 8, c = k & 255;
            dst[k] = (r < p.band_rows && c < p.C) ? src[r * p.C + c] : E(0);
        }
        __syncthreads();
        tab_off = (uint32_t)(((size_t)(p.band_rows * 256 + 1) * sizeof(E) + 15) & ~(size_t)15);
        tab = reinterpret_cast<const E*>(smem);
    } else {
        tab = static_cast<const E*>(p.table);
    }
    const int ldq = p.ldq;
    const uint32_t C = (uint32_t)p.C;
    const uint32_t tbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem);  // LDS address of the table
    // a table in device memory is never read past its end, whatever core ids the caller put in A (ids >= C
    // are the caller's error; LDS tables need no guard: out-of-range LDS reads return nothing harmful)
    auto in_table = [&](uint32_t idx) -> uint32_t {
        if constexpr (TABMODE == kTabGlobal) return min(idx, C * C - 1u);
        else return idx;
    };
    // Band mode: the band starts at a multiple of its (power of two) number of rows, so XOR-ing the
    // source core with band_lo maps the band onto 0..band_rows-1 and everything else above it; the byte
    // offset of (from ^ band_lo) << 8 | to is then clamped onto the zero entry that follows the band.
    const uint32_t band_x4 = (uint32_t)p.band_lo * 0x01010101u;
    const uint32_t band_slots = (uint32_t)p.band_rows << 8;  // <= 2^14: fits the 16-bit packed minimum
    const uint32_t band_end = band_slots * (uint32_t)sizeof(E);
    auto band_at = [&](uint32_t byte_off) -> Acc {
        return *reinterpret_cast<const E*>(smem + min(byte_off, band_end));
    };
    uint8_t* strip = smem + tab_off + (size_t)wave * ldq;
    const bool lane_on = lane * 16 < ldq;
    const int nvalid_row = min(16, max(0, p.Q - lane * 16));
    uint32_t keep[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int nb = min(4, max(0, nvalid_row - 4 * w));
        keep[w] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    }
    auto pair_at = [&](uint32_t from, uint32_t to) -> Acc {
        if constexpr (TABMODE == kTabLds256) return tab[(from << 8) | to];
        else if constexpr (TABMODE == kTabLdsBand) return band_at(((((from ^ (uint32_t)p.band_lo) << 8) | to)) * (uint32_t)sizeof(E));
        else return tab[in_table(__umul24(from, C) + to)];
    };
    auto four = [&](uint32_t pf, uint32_t cf) -> Acc {
        if constexpr (TABMODE == kTabLds256) {
            E e0, e1, e2, e3;
            entries_at_halves<E>(__builtin_amdgcn_perm(pf, cf, 0x05010400u), tbase, e0, e1);
            entries_at_halves<E>(__builtin_amdgcn_perm(pf, cf, 0x07030602u), tbase, e2, e3);
            return ((Acc)e0 + (Acc)e1) + ((Acc)e2 + (Acc)e3);
        } else if constexpr (TABMODE == kTabLdsBand) {
            // two 16-bit slots per dword, both clamped onto the zero entry by one packed minimum
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            const uint32_t px = pf ^ band_x4;
            const u16x2 cap = {(unsigned short)band_slots, (unsigned short)band_slots};
            const u16x2 lo = __builtin_elementwise_min(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(px, cf, 0x05010400u)), cap);
            const u16x2 hi = __builtin_elementwise_min(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(px, cf, 0x07030602u)), cap);
            return ((Acc)tab[lo.x] + (Acc)tab[lo.y]) + ((Acc)tab[hi.x] + (Acc)tab[hi.y]);
        } else {
            Acc s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (Acc)tab[in_table(__umul24((pf >> (8 * j)) & 0xFFu, C) + ((cf >> (8 * j)) & 0xFFu))];
            return s;
        }
    };

    for (int64_t it = (int64_t)blockIdx.x * kV3Waves + wave; it < p.nitems; it += (int64_t)gridDim.x * kV3Waves) {
        const int chunk = (int)(it / p.B);
        const int64_t b = it - (int64_t)chunk * p.B;
        const int r0 = chunk * p.rows_per_item;
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const uint8_t* base = p.A + (size_t)b * p.Tp * ldq;
        Acc acc = 0;

        // Loads are unconditional (addresses clamped into the item / the gate list) so that the row
        // loop has no divergent control flow; what they return for a row past the end is never used.
        // Bases are wave-uniform, the per-lane part of every address is fixed.
        const int lane16 = lane_on ? lane * 16 : 0;
        const uint32_t* sched_lane = p.edges_sched + lane;
        auto fetch_row = [&](int row) -> uint4 {
            const uint8_t* rp = base + (size_t)min(row, r1 - 1) * ldq;
            return load_row16(rp + lane16);
        };
        // gate `lane` of chunk c of the row (a later slice's gates or padding if the row has fewer
        // chunks: the consumer skips those, so nothing here waits on a loaded value)
        auto fetch_gate = [&](int row, int c) -> uint32_t {
            const uint32_t* gp = sched_lane + p.sptr[min(row, r1 - 1)];
            return gp[c * kWave];
        };

        // previous row with the bytes of qubits that do not exist zeroed (keep is zero for unused lanes)
        uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        if (r0 >= 1) {
            const uint4 prv = fetch_row(r0 - 1);
            p0 = prv.x & keep[0];
            p1 = prv.y & keep[1];
            p2 = prv.z & keep[2];
            p3 = prv.w & keep[3];
        }
        uint4 ring[AHEAD];
        uint32_t gring[AHEAD][kRowEdgeChunks];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            ring[k] = fetch_row(r0 + k);
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(r0 + k, c);
        }

        for (int g0 = r0; g0 < r1; g0 += AHEAD) {
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                const int row = g0 + k;
                if (row >= r1) break;
                const uint4 cur = ring[k];
                uint32_t gates[kRowEdgeChunks];
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gates[c] = gring[k][c];
                ring[k] = fetch_row(row + AHEAD);
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(row + AHEAD, c);

                if (lane_on) *reinterpret_cast<uint4*>(strip + lane * 16) = cur;
                // endpoint gathers of the row's gates first, so the movement look-ups overlap them; a
                // chunk the row does not have uses (0, 0): one qubit twice, distance zero
                const int s0 = p.sptr[row];
                const int nch = (p.sptr[row + 1] - s0) / kWave;
                uint32_t au[kRowEdgeChunks], av[kRowEdgeChunks];
                if (nch >= kRowEdgeChunks) {
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {
                        au[c] = strip[gates[c] & 0xFFFFu];
                        av[c] = strip[gates[c] >> 16];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {
                        const uint32_t g = c < nch ? gates[c] : 0u;
                        au[c] = strip[g & 0xFFFFu];
                        av[c] = strip[g >> 16];
                    }
                }
                const uint32_t c0 = cur.x & keep[0], c1 = cur.y & keep[1], c2 = cur.z & keep[2], c3 = cur.w & keep[3];
                if (row >= 2) {  // movement is charged from row 2 on
                    if constexpr (TABMODE == kTabLdsBand) {
                        // instruction-bound: skip only when no lane of the wave saw a qubit move
                        const bool moved = ((c0 ^ p0) | (c1 ^ p1) | (c2 ^ p2) | (c3 ^ p3)) != 0;
                        if (__builtin_amdgcn_ballot_w64(moved) != 0)
                            acc += (four(p0, c0) + four(p1, c1)) + (four(p2, c2) + four(p3, c3));
                    } else {
                        // LDS-bound: a lane skips a group of four qubits that all stayed where they were; fewer
                        // active lanes per look-up instruction means fewer bank conflicts (placers that move
                        // few qubits per slice)
                        if (c0 != p0) acc += four(p0, c0);
                        if (c1 != p1) acc += four(p1, c1);
                        if (c2 != p2) acc += four(p2, c2);
                        if (c3 != p3) acc += four(p3, c3);
                    }
                }
                // a lane whose four gates all have both qubits on one core (what a placer works for) looks nothing up
                if (((au[0] ^ av[0]) | (au[1] ^ av[1]) | (au[2] ^ av[2]) | (au[3] ^ av[3])) != 0) {
                    Acc s = 0;
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) s += pair_at(au[c], av[c]);
                    acc += s;
                }
                for (int c = kRowEdgeChunks; c < nch; ++c) {
                    const uint32_t g = sched_lane[s0 + c * kWave];
                    acc += pair_at(strip[g & 0xFFFFu], strip[g >> 16]);
                }
                p0 = c0;
                p1 = c1;
                p2 = c2;
                p3 = c3;
            }
        }
        const double s = wave_sum(acc);
        if (lane == 0) p.partial[it] = s;
    }
}

// Rows of 513..1024 bytes with the table in LDS at pitch 256, four partitions at a time.  The gates of slice t
// are the same for every partition, so a wave takes row t of four consecutive partitions in one step and parks
// them interleaved: strip word q = the cores of qubit q in the four partitions.  One ds_read_b32 then fetches a
// gate endpoint for all four (two gathers per row and partition instead of eight), and a (word of first
// endpoints, word of second endpoints) pair feeds the same packed look-ups as a (previous, current) pair of
// movement words, one partition per byte lane.  The LDS instruction count per row drops from 29 to 23, and
// that count is what bounds the table kernels.  Eight waves per workgroup: 8 x 4 KB of strips beside a 128 KB
// table fill the CU's LDS exactly.
constexpr int kRow4Waves = 8;
constexpr int kRow4Threads = kRow4Waves * kWave;
constexpr int kRow4Strip = 4096;

template <int KIND, int AHEAD>
__global__ __launch_bounds__(kRow4Threads) void k_cost_v3_row4(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = typename Tab<KIND>::elem;
    using Acc = typename Tab<KIND>::acc;
    static_assert(kRowEdgeChunks == 4, "four gate chunks per row are prefetched");
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t tab_off = stage_table_pitch256<E>(smem, p.table, p.C);
    const uint32_t tbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem);  // the table's LDS address
    uint32_t* strip = reinterpret_cast<uint32_t*>(smem + tab_off + (size_t)wave * kRow4Strip);  // [qubit] -> 4 cores
    const int ldq = p.ldq;
    const bool lane_on = lane * 16 < ldq;
    const int lane16 = lane_on ? lane * 16 : 0;
    const int nvalid_row = min(16, max(0, p.Q - lane * 16));
    uint32_t keep[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int nb = min(4, max(0, nvalid_row - 4 * w));
        keep[w] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    }
    const uint32_t* sched_lane = p.edges_sched + lane;
    const int64_t ngroups = (p.B + 3) / 4;
    // A lane parks its 16 qubits as four 16-byte writes into its own 64 bytes of the strip.  Eight lanes make one
    // LDS pass; with every lane writing the same quarter they would meet in 8 of the 32 banks, so lane pairs
    // permute their quarters: qubit q sits at word q ^ (((q >> 5) & 3) << 2).  Both qubits of a packed gate word
    // are mapped at once, and one v_mad_u32_u16 per endpoint turns a half into the LDS address of its strip word.
    const int quarter_x = (lane >> 1) & 3;
    const uint32_t sbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem + tab_off + (uint32_t)wave * kRow4Strip);
    auto endpoints = [&](uint32_t g, uint32_t& u4, uint32_t& v4) {
        const uint32_t x = g ^ ((g >> 3) & 0x000C000Cu);
        uint32_t a0, a1;
        asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(sbase));
        asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(sbase));
        u4 = *(lds_u32*)(uintptr_t)a0;
        v4 = *(lds_u32*)(uintptr_t)a1;
    };
    for (int64_t it = (int64_t)blockIdx.x * kRow4Waves + wave; it < p.nitems; it += (int64_t)gridDim.x * kRow4Waves) {  // items = chunks x groups
        const int chunk = (int)(it / ngroups);
        const int64_t b0 = (it - (int64_t)chunk * ngroups) * 4;
        const int r0 = chunk * p.rows_per_item;
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const uint8_t* base[4];  // a group that runs past B repeats the last partition and drops the copies
#pragma unroll
        for (int g = 0; g < 4; ++g) base[g] = p.A + (size_t)min(b0 + g, p.B - 1) * p.Tp * ldq + lane16;
        Acc acc[4] = {0, 0, 0, 0};

        auto fetch_row = [&](int row, int g) -> uint4 { return load_row16(base[g] + (size_t)min(row, r1 - 1) * ldq); };
        auto fetch_gate = [&](int row, int c) -> uint32_t { return sched_lane[p.sptr[min(row, r1 - 1)] + c * kWave]; };
        // one look-up per partition: the byte lanes of (from, to) index the table as from << 8 | to
        auto look4 = [&](uint32_t from, uint32_t to) {
            E e0, e1, e2, e3;
            entries_at_halves<E>(__builtin_amdgcn_perm(from, to, 0x05010400u), tbase, e0, e1);
            entries_at_halves<E>(__builtin_amdgcn_perm(from, to, 0x07030602u), tbase, e2, e3);
            acc[0] += (Acc)e0;
            acc[1] += (Acc)e1;
            acc[2] += (Acc)e2;
            acc[3] += (Acc)e3;
        };
        // the four look-ups of one partition's (previous, current) word
        auto four = [&](int g, uint32_t pf, uint32_t cf) {
            E e0, e1, e2, e3;
            entries_at_halves<E>(__builtin_amdgcn_perm(pf, cf, 0x05010400u), tbase, e0, e1);
            entries_at_halves<E>(__builtin_amdgcn_perm(pf, cf, 0x07030602u), tbase, e2, e3);
            acc[g] += ((Acc)e0 + (Acc)e1) + ((Acc)e2 + (Acc)e3);
        };

        uint32_t prev[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r0 >= 1) v = fetch_row(r0 - 1, g);
            prev[g][0] = v.x & keep[0];
            prev[g][1] = v.y & keep[1];
            prev[g][2] = v.z & keep[2];
            prev[g][3] = v.w & keep[3];
        }
        uint4 ring[AHEAD][4];
        uint32_t gring[AHEAD][kRowEdgeChunks];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ring[k][g] = fetch_row(r0 + k, g);
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(r0 + k, c);
        }
        // interleave: word w of the four partitions -> four strip words (one per qubit), 4 x 4 byte transposes
        auto park = [&](const uint32_t (&x)[4][4]) {
            if (lane_on) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t t01a = __builtin_amdgcn_perm(x[1][w], x[0][w], 0x05010400u);  // a0 b0 a1 b1
                    const uint32_t t01b = __builtin_amdgcn_perm(x[1][w], x[0][w], 0x07030602u);  // a2 b2 a3 b3
                    const uint32_t t23a = __builtin_amdgcn_perm(x[3][w], x[2][w], 0x05010400u);  // c0 d0 c1 d1
                    const uint32_t t23b = __builtin_amdgcn_perm(x[3][w], x[2][w], 0x07030602u);
                    uint4 o;
                    o.x = __builtin_amdgcn_perm(t23a, t01a, 0x05040100u);  // a0 b0 c0 d0
                    o.y = __builtin_amdgcn_perm(t23a, t01a, 0x07060302u);  // a1 b1 c1 d1
                    o.z = __builtin_amdgcn_perm(t23b, t01b, 0x05040100u);
                    o.w = __builtin_amdgcn_perm(t23b, t01b, 0x07060302u);
                    *reinterpret_cast<uint4*>(strip + lane * 16 + 4 * (w ^ quarter_x)) = o;
                }
            }
        };
        // endpoint words of the first chunks of a row's gates from the strip that holds that row (a chunk the row
        // lacks: qubit 0 with itself, distance zero)
        auto gather = [&](int row, const uint32_t (&gates)[kRowEdgeChunks], uint32_t (&U)[kRowEdgeChunks], uint32_t (&V)[kRowEdgeChunks]) {
            const int rr = min(row, r1 - 1);
            const int nch = (p.sptr[rr + 1] - p.sptr[rr]) / kWave;
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; ++c) {
                endpoints(c < nch ? gates[c] : 0u, U[c], V[c]);
            }
        };
        auto masked = [&](const uint4 (&r)[4], uint32_t (&x)[4][4]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                x[g][0] = r[g].x & keep[0];
                x[g][1] = r[g].y & keep[1];
                x[g][2] = r[g].z & keep[2];
                x[g][3] = r[g].w & keep[3];
            }
        };
        // The strip write and the gathers of row t + 1 are issued before the table look-ups of row t, whose
        // addresses were gathered one step earlier: one LDS round trip per step in the dependency chain, not two.
        uint32_t cur[4][4], U[kRowEdgeChunks], V[kRowEdgeChunks];
        masked(ring[0], cur);
        park(cur);
        gather(r0, gring[0], U, V);
        for (int g0 = r0; g0 < r1; g0 += AHEAD) {
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                const int row = g0 + k;
                if (row >= r1) break;
                const int kn = (k + 1) % AHEAD;  // slot of row + 1 (slot 0 was refilled earlier in this pass)
#pragma unroll
                for (int g = 0; g < 4; ++g) ring[k][g] = fetch_row(row + AHEAD, g);
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(row + AHEAD, c);
                // gates beyond the prefetched chunks (rare) while the strip still holds this row
                const int s0 = p.sptr[row];
                const int nch = (p.sptr[row + 1] - s0) / kWave;
                for (int c = kRowEdgeChunks; c < nch; ++c) {
                    const uint32_t g = sched_lane[s0 + c * kWave];
                    uint32_t u4, v4;
                    endpoints(g, u4, v4);
                    look4(u4, v4);
                }
                uint32_t nxt[4][4], Un[kRowEdgeChunks], Vn[kRowEdgeChunks];
                masked(ring[kn], nxt);
                park(nxt);
                gather(row + 1, gring[kn], Un, Vn);
                if (row >= 2) {  // movement is charged from row 2 on
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int w = 0; w < 4; ++w)
                            if (cur[g][w] != prev[g][w]) four(g, prev[g][w], cur[g][w]);
                }
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) look4(U[c], V[c]);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        prev[g][w] = cur[g][w];
                        cur[g][w] = nxt[g][w];
                    }
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) {
                    U[c] = Un[c];
                    V[c] = Vn[c];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const double sum = wave_sum(acc[g]);
            if (lane == 0 && b0 + g < p.B) p.partial[(size_t)chunk * p.B + b0 + g] = sum;
        }
    }
}

// fp64 table too large for LDS (more than about 134 cores), symmetric up to rounding: instead of looking
// distances up, count how often every unordered pair of cores occurs and multiply by the table once per
// item.  The counters (one uint32 per unordered pair) fit in LDS where the table does not.  One workgroup =
// one item (rows r0..r1 of one partition); its waves take consecutive runs of those rows, feed the shared
// counters with LDS atomics, and after a barrier the whole workgroup forms sum(count * Dsym) over all slots
// in a fixed order, zeroing the counters as it goes.  Dsym is the mean of D[a][b] and D[b][a] (written by
// k_sym_fold), so what is lost against summing the directed entries is
// |count(a->b) - count(b->a)| * |D[a][b] - D[b][a]| / 2 per pair.
//
// Slot of the pair hi >= lo: the triangle (diagonal included) folded into a 128 x 256 rectangle,
//   key = hi << 8 | lo;   slot = key & 0x7FFF when hi >= 128,  ~key & 0x7FFF when hi < 128
// (row hi - 128 holds columns 0..hi, row 127 - hi the columns 255 - hi..255; the two diagonal entries that
// share a slot are never summed: the table is zero there).  Two pairs are handled per register: the keys
// p << 8 | c and c << 8 | p of both come out of one v_perm_b32 each, their packed 16-bit maximum is
// hi << 8 | lo, an arithmetic shift by 15 and an exclusive-nor fold it (bit 15 of the result is always set,
// which the address constant absorbs), and one v_mad_u32_u16 per pair turns a half into an LDS address.
constexpr int kPairSlots = 128 * 256;

__global__ void k_sym_fold(const double* D, int32_t C, double* fold) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kPairSlots) return;
    const int r = idx >> 8, col = idx & 255;
    int hi, lo;
    if (col <= 128 + r) {
        hi = 128 + r;
        lo = col;
    } else {
        hi = 127 - r;
        lo = 255 - col;
    }
    fold[idx] = (hi < C && lo < hi) ? (D[(size_t)lo * C + hi] + D[(size_t)hi * C + lo]) * 0.5 : 0.0;
}

// hi << 8 | lo of two (a, b) pairs, folded: k = a << 8 | b and s = b << 8 | a in each 16-bit half
__device__ inline uint32_t fold_keys(uint32_t k, uint32_t s) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    const u16x2 m = __builtin_elementwise_max(__builtin_bit_cast(u16x2, k), __builtin_bit_cast(u16x2, s));
    const i16x2 t = __builtin_bit_cast(i16x2, m) >> 15;
    return ~(__builtin_bit_cast(uint32_t, m) ^ __builtin_bit_cast(uint32_t, t));
}

// One count for each of the two pairs whose folded keys are the halves of x.  base = LDS address of the
// counters minus 0x20000 (bit 15 of a folded key is set: 0x8000 * 4).
__device__ inline void count_keys(uint32_t x, uint32_t base) {
    uint32_t a0, a1;
    asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(base));
    asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(base));
    __hip_atomic_fetch_add((lds_u32*)(uintptr_t)a0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add((lds_u32*)(uintptr_t)a1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the four (previous, current) pairs of a dword; a qubit that stayed counts into a diagonal slot
__device__ inline void count_moves4(uint32_t pf, uint32_t cf, uint32_t base) {
    count_keys(fold_keys(__builtin_amdgcn_perm(pf, cf, 0x05010400u), __builtin_amdgcn_perm(cf, pf, 0x05010400u)), base);
    count_keys(fold_keys(__builtin_amdgcn_perm(pf, cf, 0x07030602u), __builtin_amdgcn_perm(cf, pf, 0x07030602u)), base);
}
// two gates: the cores of their first qubits in the 16-bit halves of u2, of their second qubits in v2
__device__ inline void count_gate_pair2(uint32_t u2, uint32_t v2, uint32_t base) {
    if (u2 != v2) count_keys(fold_keys(u2 << 8 | v2, v2 << 8 | u2), base);
}
// sum over all slots of count * symmetrised entry, in a fixed order, zeroing the counters on the way; table
// entries are loaded whether or not the pair occurred (coalesced, cache-resident), sixteen at a time, so the loop
// is not a chain of load latencies; an unused entry may be infinite: select, not multiply
template <int THREADS>
__device__ inline double apply_pair_counts(uint32_t* cnt, const double* fold) {
    double acc = 0.0;
    constexpr int kBatch = 16;  // slots per thread whose table entries are in flight together
    static_assert(kPairSlots % (kBatch * THREADS) == 0, "whole batches of slots per thread");
    for (int k0 = threadIdx.x; k0 < kPairSlots; k0 += kBatch * THREADS) {
        double d[kBatch];
        uint32_t n[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            d[j] = fold[k0 + j * THREADS];
            n[j] = cnt[k0 + j * THREADS];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            if (n[j]) {
                cnt[k0 + j * THREADS] = 0;
                acc += (double)n[j] * d[j];
            }
        }
    }
    return acc;
}

// The same pass for the per-core activity (HQA_placer.py:288-321: every move and every gate across two cores is one
// event for each of the two cores): a pair's count goes to both of its cores in hist[256], diagonal slots (a qubit
// that stayed, a gate inside one core) to nobody.  Slot k of the folded rectangle is row k >> 8, column k & 255:
// pair (128 + row, column) up to the diagonal, pair (127 - row, 255 - column) beyond it (see k_sym_fold).
template <int THREADS>
__device__ inline void apply_pair_activity(uint32_t* cnt, uint32_t* hist) {
    for (int k = threadIdx.x; k < kPairSlots; k += THREADS) {
        const uint32_t n = cnt[k];
        if (n) {
            cnt[k] = 0;
            const int r = k >> 8, col = k & 255;
            const bool upper = col <= 128 + r;
            const int hi = upper ? 128 + r : 127 - r, lo = upper ? col : 255 - col;
            if (hi != lo) {
                atomicAdd(&hist[hi], n);
                atomicAdd(&hist[lo], n);
            }
        }
    }
}

// Helpers of the row walk of the gate-ordered cost (k_cost_v2_walk, k_cost_v2_pairs).
// non-zero iff some byte of y is 0xFF
__device__ inline uint32_t bytes_ff(uint32_t y) { return (~y - 0x01010101u) & y & 0x80808080u; }
// a visited byte (mask m) that is 0xFF in the row (c) but not among the last cores (l): a qubit that had a core and
// now has none, which the reference cannot price
__device__ inline bool walk_lost(uint32_t m, uint32_t l, uint32_t c) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        n += (((m >> (8 * j)) & 0xFFu) != 0 && ((c >> (8 * j)) & 0xFFu) == 0xFFu && ((l >> (8 * j)) & 0xFFu) != 0xFFu) ? 1 : 0;
    return n != 0;
}
// non-zero iff some byte of y is >= n (1 <= n <= 255; n is the same for the whole wave)
__device__ inline uint32_t bytes_ge(uint32_t y, int n) {
    const uint32_t low = y & 0x7F7F7F7Fu;
    if (n <= 128) return ((low + (uint32_t)(128 - n) * 0x01010101u) | y) & 0x80808080u;
    return (low + (uint32_t)(256 - n) * 0x01010101u) & y & 0x80808080u;
}
// a visited byte (mask m) of the row (c) that names a core outside 0 .. C - 1 (0xFF, no core, is not one)
__device__ inline bool walk_stray(uint32_t m, uint32_t c, int C) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t v = (c >> (8 * j)) & 0xFFu;
        n += (((m >> (8 * j)) & 0xFFu) != 0 && v != 0xFFu && v >= (uint32_t)C) ? 1 : 0;
    }
    return n != 0;
}
// The cores the lane's 16 qubits (row bytes lane16 .. lane16 + 15) had at their last visit before slice r0, from
// vprev[r0][q] = that slice or -1; 0xFF where there was none.  Sixteen byte gathers per lane, all in flight together.
__device__ inline void walk_state_at(const uint8_t* base, const int16_t* vprev, int r0, int ldq, int lane16, uint32_t (&st)[4]) {
    st[0] = st[1] = st[2] = st[3] = ~0u;
    if (r0 == 0) return;
    const uint4* pr = reinterpret_cast<const uint4*>(vprev + (size_t)r0 * ldq + lane16);
    const uint4 a = pr[0], b = pr[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t got[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = (int)(int16_t)(w[j / 2] >> (16 * (j & 1)));
        got[j] = base[(size_t)max(r, 0) * ldq + lane16 + j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = (int)(int16_t)(w[j / 2] >> (16 * (j & 1)));
        const uint32_t v = r >= 0 ? got[j] : 0xFFu;
        st[j / 4] = (st[j / 4] & ~(0xFFu << (8 * (j & 3)))) | (v << (8 * (j & 3)));
    }
}

// The same pass for the core x core communication counts of the gate-ordered walk (Qubo_placer.py:208-245: every event
// between two different cores adds one to M[a][b] and to M[b][a]); several items of a partition add into one matrix,
// a partition that is one item stores into its (zeroed) matrix.
template <int THREADS>
__device__ inline void apply_pair_comm(uint32_t* cnt, int32_t* M, int C, bool only_writer) {
    for (int k = threadIdx.x; k < kPairSlots; k += THREADS) {
        const uint32_t n = cnt[k];
        if (n) {
            cnt[k] = 0;
            const int r = k >> 8, col = k & 255;
            const bool upper = col <= 128 + r;
            const int hi = upper ? 128 + r : 127 - r, lo = upper ? col : 255 - col;
            if (hi != lo && hi < C) {
                if (only_writer) {
                    M[hi * C + lo] = (int)n;
                    M[lo * C + hi] = (int)n;
                } else {
                    atomicAdd(&M[hi * C + lo], (int)n);
                    atomicAdd(&M[lo * C + hi], (int)n);
                }
            }
        }
    }
}

// WALK: the gate-ordered cost (V1 / V2 / V4; V2 and V4 differ only in the table the caller passes once it is symmetrised)
// instead of the dense one: movement of the qubits a slice visits against their cores at their previous visit (see
// k_cost_v2_walk), from the first row on; a wave's run starts from the state walk_state_at gathers.
// COMM (with WALK): the counts themselves are the result (apply_pair_comm); a visited qubit on a core outside 0 .. C - 1
// is an error like an absent operand.
template <int AHEAD, bool ACTIVITY, bool WALK, bool COMM = false>
__device__ __forceinline__ void pairs_by_row(const V3Args& p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int ldq = p.ldq;
    // LDS: the counters, the waves' strips, the workgroup reduction.  Every byte a core id can take has a
    // slot, so ids >= C (the caller's error) count into slots whose table entry is zero.
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);
    uint8_t* strip = smem + (size_t)kPairSlots * 4 + (size_t)wave * ldq;
    double* red = reinterpret_cast<double*>(smem + (size_t)kPairSlots * 4 + (size_t)kV3Waves * ldq);
    const uint32_t cbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem - 0x20000u);
    const double* fold = static_cast<const double*>(p.table);
    uint32_t* hist = reinterpret_cast<uint32_t*>(red + kV3Waves);  // ACTIVITY: events per core
    for (int k = threadIdx.x; k < kPairSlots; k += kV3Threads) cnt[k] = 0;
    if (ACTIVITY && threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();

    const bool lane_on = lane * 16 < ldq;
    const uint32_t lane_mask = lane_on ? ~0u : 0u;
    const int nvalid_row = min(16, max(0, p.Q - lane * 16));
    uint32_t keep[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int nb = min(4, max(0, nvalid_row - 4 * w));
        keep[w] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    }
    auto count4 = [&](uint32_t pf, uint32_t cf) { count_moves4(pf, cf, cbase); };
    auto count_gates2 = [&](uint32_t u2, uint32_t v2) { count_gate_pair2(u2, v2, cbase); };

    for (int64_t it = blockIdx.x; it < p.nitems; it += gridDim.x) {
        const int chunk = (int)(it / p.B);
        const int64_t b = it - (int64_t)chunk * p.B;
        const int r0 = chunk * p.rows_per_item;
        const int r1 = min(p.T, r0 + p.rows_per_item);
        // runs that differ by at most one row: every wave reaches the barrier at about the same time
        const int w0 = r0 + (r1 - r0) * wave / kV3Waves, w1 = r0 + (r1 - r0) * (wave + 1) / kV3Waves;
        const uint8_t* base = p.A + (size_t)b * p.Tp * ldq;
        bool bad = false;  // WALK: a case the reference fails on
        if (w0 < w1) {
            const int lane16 = lane_on ? lane * 16 : 0;
            const uint32_t* sched_lane = p.edges_sched + lane;
            auto fetch_row = [&](int row) -> uint4 { return load_row16(base + (size_t)min(row, w1 - 1) * ldq + lane16); };
            auto fetch_mask = [&](int row) -> uint4 {  // WALK: the bytes of the qubits the slice visits
                const uint4 m = *reinterpret_cast<const uint4*>(p.vmask + (size_t)min(row, w1 - 1) * ldq + lane16);
                return make_uint4(m.x & lane_mask, m.y & lane_mask, m.z & lane_mask, m.w & lane_mask);
            };
            // Slice starts in the gate list.  The barriers of this kernel keep the compiler from reading them
            // through the scalar cache, and as wave-uniform vector loads it would move each result to a
            // scalar register at once, draining every prefetch issued before it.  So they are loaded through
            // an index the compiler cannot prove uniform, stay in vector registers, and are made scalar
            // where they are used: the start of slice row + AHEAD + 1 is requested a row before it is
            // needed, ahead of that row's other loads; start and end travel with the ring.
            int opaque_zero = 0;
            asm volatile("" : "+v"(opaque_zero));
            auto slice_start = [&](int row) -> int { return p.sptr[min(row, w1) + opaque_zero]; };
            auto fetch_gate = [&](int start, int c) -> uint32_t { return sched_lane[start + c * kWave]; };
            uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            if constexpr (WALK) {
                uint32_t st[4];
                walk_state_at(base, p.vprev, w0, ldq, lane16, st);
                p0 = st[0];
                p1 = st[1];
                p2 = st[2];
                p3 = st[3];
            } else if (w0 >= 1) {
                const uint4 prv = fetch_row(w0 - 1);
                p0 = prv.x & keep[0];
                p1 = prv.y & keep[1];
                p2 = prv.z & keep[2];
                p3 = prv.w & keep[3];
            }
            uint4 ring[AHEAD], mring[WALK ? AHEAD : 1];
            uint32_t gring[AHEAD][kRowEdgeChunks];
            int sring[AHEAD], ering[AHEAD];
            int sp_next = slice_start(w0);
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                sring[k] = sp_next;
                sp_next = slice_start(w0 + k + 1);
                ering[k] = sp_next;
                ring[k] = fetch_row(w0 + k);
                if constexpr (WALK) mring[k] = fetch_mask(w0 + k);
                const int start = __builtin_amdgcn_readfirstlane(sring[k]);
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(start, c);
            }
            // one row: consume slot k of the ring, refill it with row + AHEAD (REFILL), count the row's pairs
            auto do_row = [&](int row, uint4& rslot, uint4& mslot, uint32_t (&gslot)[kRowEdgeChunks], int& sslot, int& eslot, bool refill) {
                const uint4 cur = rslot;
                uint4 vm = make_uint4(0, 0, 0, 0);
                if constexpr (WALK) vm = mslot;
                const int s0 = __builtin_amdgcn_readfirstlane(sslot);
                const int nch = (__builtin_amdgcn_readfirstlane(eslot) - s0) / kWave;
                uint32_t gates[kRowEdgeChunks];
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gates[c] = gslot[c];
                if (refill) {
                    const int sp_here = sp_next;             // start of slice row + AHEAD, asked for a row ago
                    sp_next = slice_start(row + AHEAD + 1);  // used a row from now
                    sslot = sp_here;
                    eslot = sp_next;
                    rslot = fetch_row(row + AHEAD);
                    if constexpr (WALK) mslot = fetch_mask(row + AHEAD);
                    const int start = __builtin_amdgcn_readfirstlane(sp_here);
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) gslot[c] = fetch_gate(start, c);
                }
                if (lane_on) *reinterpret_cast<uint4*>(strip + lane * 16) = cur;
                uint32_t au[kRowEdgeChunks], av[kRowEdgeChunks];
                if (nch >= kRowEdgeChunks) {  // the usual case, without a select per chunk
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {
                        au[c] = strip[gates[c] & 0xFFFFu];
                        av[c] = strip[gates[c] >> 16];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {  // a chunk the row lacks: one qubit twice
                        const uint32_t g = c < nch ? gates[c] : 0u;
                        au[c] = strip[g & 0xFFFFu];
                        av[c] = strip[g >> 16];
                    }
                }
                const uint32_t c0 = cur.x & keep[0], c1 = cur.y & keep[1], c2 = cur.z & keep[2], c3 = cur.w & keep[3];
                // the gates first: their counts hang on the gathers just issued, and LDS operations complete in
                // order, so whatever is issued between a gather and its use is waited for as well
                static_assert(kRowEdgeChunks == 4, "the first four gate chunks of a row are counted two to a register");
                // byte 0 of two gathered values into the 16-bit halves of one register
                auto halves = [](uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x0c040c00u); };
                count_gates2(halves(au[0], au[1]), halves(av[0], av[1]));
                count_gates2(halves(au[2], au[3]), halves(av[2], av[3]));
                for (int c = kRowEdgeChunks; c < nch; ++c) {
                    const uint32_t g = sched_lane[s0 + c * kWave];
                    const uint32_t u = strip[g & 0xFFFFu], v = strip[g >> 16];
                    count_gates2(u, v);  // the upper halves are the pair (0, 0): a diagonal slot
                    if constexpr (COMM) bad = bad || (g != 0u && max(u, v) >= (uint32_t)p.C);
                    else if constexpr (WALK) bad = bad || (p.gate_errors && g != 0u && max(u, v) == 0xFFu);
                }
                if constexpr (WALK) {
                    // a counted gate (padding words are zero) with an operand on no core
                    uint32_t top = 0;
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) top = max(top, max(au[c], av[c]));
                    if (COMM ? top >= (uint32_t)p.C : (top == 0xFFu && p.gate_errors)) {
#pragma unroll
                        for (int c = 0; c < kRowEdgeChunks; ++c)
                            bad = bad || ((c < nch ? gates[c] : 0u) != 0u && max(au[c], av[c]) >= (COMM ? (uint32_t)p.C : 0xFFu));
                    }
                    // visited qubits against their cores at the previous visit; the others against themselves
                    const uint32_t f0 = (vm.x & p0) | (~vm.x & cur.x), f1 = (vm.y & p1) | (~vm.y & cur.y);
                    const uint32_t f2 = (vm.z & p2) | (~vm.z & cur.z), f3 = (vm.w & p3) | (~vm.w & cur.w);
                    if (f0 != cur.x) count4(f0, cur.x);
                    if (f1 != cur.y) count4(f1, cur.y);
                    if (f2 != cur.z) count4(f2, cur.z);
                    if (f3 != cur.w) count4(f3, cur.w);
                    if constexpr (COMM) {
                        // a visited byte >= C (0xFF included): then the exact tests
                        if ((bytes_ge(cur.x & vm.x, p.C) | bytes_ge(cur.y & vm.y, p.C) | bytes_ge(cur.z & vm.z, p.C) |
                             bytes_ge(cur.w & vm.w, p.C)) != 0)
                            bad = bad || walk_lost(vm.x, p0, cur.x) || walk_lost(vm.y, p1, cur.y) || walk_lost(vm.z, p2, cur.z) ||
                                  walk_lost(vm.w, p3, cur.w) || walk_stray(vm.x, cur.x, p.C) || walk_stray(vm.y, cur.y, p.C) ||
                                  walk_stray(vm.z, cur.z, p.C) || walk_stray(vm.w, cur.w, p.C);
                    } else if ((bytes_ff(cur.x & vm.x) | bytes_ff(cur.y & vm.y) | bytes_ff(cur.z & vm.z) | bytes_ff(cur.w & vm.w)) != 0)
                        bad = bad || walk_lost(vm.x, p0, cur.x) || walk_lost(vm.y, p1, cur.y) || walk_lost(vm.z, p2, cur.z) ||
                              walk_lost(vm.w, p3, cur.w);
                    p0 = (vm.x & cur.x) | (~vm.x & p0);
                    p1 = (vm.y & cur.y) | (~vm.y & p1);
                    p2 = (vm.z & cur.z) | (~vm.z & p2);
                    p3 = (vm.w & cur.w) | (~vm.w & p3);
                    return;
                }
                if (row >= (ACTIVITY ? 1 : 2)) {  // movement is charged from row 2 on (events are counted from row 1 on); a dword of qubits that all stayed is skipped
                    if (c0 != p0) count4(p0, c0);
                    if (c1 != p1) count4(p1, c1);
                    if (c2 != p2) count4(p2, c2);
                    if (c3 != p3) count4(p3, c3);
                }
                p0 = c0;
                p1 = c1;
                p2 = c2;
                p3 = c3;
            };
            // whole groups of AHEAD rows without an exit inside the group (the ring then stays in fixed
            // registers), then the up to AHEAD - 1 rows that are left, already in the ring
            int g0 = w0;
            for (; g0 + AHEAD <= w1; g0 += AHEAD) {
#pragma unroll
                for (int k = 0; k < AHEAD; ++k) do_row(g0 + k, ring[k], mring[WALK ? k : 0], gring[k], sring[k], ering[k], true);
            }
#pragma unroll
            for (int k = 0; k < AHEAD - 1; ++k)
                if (g0 + k < w1) do_row(g0 + k, ring[k], mring[WALK ? k : 0], gring[k], sring[k], ering[k], false);
        }
        __syncthreads();
        if constexpr (COMM) {
            if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) p.status[b] = 1;
            apply_pair_comm<kV3Threads>(cnt, p.comm + (size_t)b * p.C * p.C, p.C, p.rows_per_item >= p.T);
            __syncthreads();  // every counter is read and zero again before the next item counts
            continue;
        }
        if constexpr (ACTIVITY) {
            apply_pair_activity<kV3Threads>(cnt, hist);
            __syncthreads();
            if (threadIdx.x < 256) {
                const uint32_t n = hist[threadIdx.x];
                hist[threadIdx.x] = 0;
                if (n && (int)threadIdx.x < p.C) atomicAdd(&p.activity[(size_t)b * p.C + threadIdx.x], (unsigned long long)n);
            }
            __syncthreads();  // hist is zero again before the next item's pass adds to it
            continue;
        }
        double acc = apply_pair_counts<kV3Threads>(cnt, fold);
        acc = wave_sum(acc);
        if constexpr (WALK) {
            if (__builtin_amdgcn_ballot_w64(bad) != 0) acc = std::numeric_limits<double>::infinity();
        }
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int w = 0; w < kV3Waves; ++w) s += red[w];
            p.partial[it] = s;
        }
        // the next item's counting cannot pass the barrier after it before thread 0 has read red[]; its
        // counter updates touch entries this pass has already zeroed or left at zero
    }
}

template <int AHEAD = kV3Ahead, bool ACTIVITY = false>
__global__ __launch_bounds__(kV3Threads) void k_cost_v3_pairs(V3Args p) {
    pairs_by_row<AHEAD, ACTIVITY, false>(p);
}

// The communication-count matrix of the gate-ordered walk from the same counts.
template <int AHEAD>
__global__ __launch_bounds__(kV3Threads) void k_comm_matrix_pairs(V3Args p) {
    pairs_by_row<AHEAD, false, true, true>(p);
}

// Cost V1 / V2 / V4 with an fp64 table that is symmetric to rounding, by core-pair counts.
template <int AHEAD>
__global__ __launch_bounds__(kV3Threads) void k_cost_v2_pairs(V3Args p) {
    pairs_by_row<AHEAD, false, true>(p);
}

// The same for rows of at most 512 bytes (grouped gate list, G rows per wave step as in k_cost_v3_groups): the
// item's steps are dealt to the workgroup's waves in runs, every wave parks its steps in its own strip
// [row before][G rows] and counts into the shared counters.
template <int AHEAD = kV3Ahead, bool ACTIVITY = false>
__global__ __launch_bounds__(kV3Threads) void k_cost_v3_pairs_groups(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int ldq = p.ldq, G = p.G, lpr = p.lpr;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);
    uint8_t* strip = smem + (size_t)kPairSlots * 4 + (size_t)wave * (G + 1) * ldq;  // [row before][G rows]
    uint8_t* rows = strip + ldq;
    double* red = reinterpret_cast<double*>(smem + (size_t)kPairSlots * 4 + (size_t)kV3Waves * (G + 1) * ldq);
    const uint32_t cbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem - 0x20000u);
    const double* fold = static_cast<const double*>(p.table);
    uint32_t* hist = reinterpret_cast<uint32_t*>(red + kV3Waves);  // ACTIVITY: events per core
    for (int k = threadIdx.x; k < kPairSlots; k += kV3Threads) cnt[k] = 0;
    if (ACTIVITY && threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();

    const bool lane_on = lane < G * lpr;
    const int lrow = lane_on ? lane / lpr : 0;
    const int lq0 = lane_on ? (lane - lrow * lpr) * 16 : 0;
    const int lane_off = lane_on ? lane * 16 : 0;
    const bool last_row = lane_on && lrow == G - 1;
    const int nvalid_row = lane_on ? min(16, max(0, p.Q - lq0)) : 0;
    uint32_t keep[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int nb = min(4, max(0, nvalid_row - 4 * w));
        keep[w] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    }
    const uint32_t* gsched_lane = p.gsched + lane;
    // byte 0 of two gathered values into the 16-bit halves of one register
    auto halves = [](uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x0c040c00u); };

    for (int64_t it = blockIdx.x; it < p.nitems; it += gridDim.x) {
        const int chunk = (int)(it / p.B);
        const int64_t b = it - (int64_t)chunk * p.B;
        const int r0 = chunk * p.rows_per_item;  // a multiple of G
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const int isteps = (r1 - r0 + G - 1) / G;
        // this wave's run of the item's steps (runs differ by at most one step)
        const int sb = isteps * wave / kV3Waves, se = isteps * (wave + 1) / kV3Waves;
        const uint8_t* base = p.A + (size_t)b * p.Tp * ldq;
        if (sb < se) {
            const int rw = r0 + sb * G;        // first row of the run
            const int nsteps = se - sb;
            const int j0 = rw / G, jlast = j0 + nsteps - 1;
            auto fetch_rows = [&](int s) -> uint4 {
                const int rowl = min(rw + s * G + lrow, r1 - 1);
                return load_row16(base + ((uint32_t)rowl * (uint32_t)ldq + (uint32_t)lq0));
            };
            auto fetch_gate = [&](int s, int c) -> uint32_t {
                const uint32_t* gp = gsched_lane + p.gptr[min(j0 + s, jlast)];
                return gp[c * kWave];
            };
            if (rw >= 1 && lane < lpr)
                *reinterpret_cast<uint4*>(strip + lane * 16) =
                    *reinterpret_cast<const uint4*>(base + (size_t)(rw - 1) * ldq + lane * 16);
            uint4 ring[AHEAD];
            uint32_t gring[AHEAD][kRowEdgeChunks];
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                ring[k] = fetch_rows(k);
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(k, c);
            }
            for (int s0 = 0; s0 < nsteps; s0 += AHEAD) {
#pragma unroll
                for (int k = 0; k < AHEAD; ++k) {
                    const int s = s0 + k;
                    if (s >= nsteps) break;
                    const uint4 cur = ring[k];
                    uint32_t gates[kRowEdgeChunks];
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) gates[c] = gring[k][c];
                    ring[k] = fetch_rows(s + AHEAD);
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(s + AHEAD, c);
                    uint4 prv = make_uint4(0, 0, 0, 0);
                    if (lane_on) {
                        *reinterpret_cast<uint4*>(rows + lane_off) = cur;
                        prv = *reinterpret_cast<const uint4*>(strip + lane_off);
                    }
                    const int w0 = p.gptr[j0 + s];
                    const int nch = (p.gptr[j0 + s + 1] - w0) / kWave;
                    uint32_t au[kRowEdgeChunks], av[kRowEdgeChunks];
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {  // a chunk the group lacks: one byte twice
                        const uint32_t g = c < nch ? gates[c] : 0u;
                        au[c] = rows[g & 0xFFFFu];
                        av[c] = rows[g >> 16];
                    }
                    count_gate_pair2(halves(au[0], au[1]), halves(av[0], av[1]), cbase);
                    count_gate_pair2(halves(au[2], au[3]), halves(av[2], av[3]), cbase);
                    for (int c = kRowEdgeChunks; c < nch; ++c) {
                        const uint32_t g = gsched_lane[w0 + c * kWave];
                        const uint32_t u = rows[g & 0xFFFFu], v = rows[g >> 16];
                        count_gate_pair2(u, v, cbase);  // the upper halves are the pair (0, 0): a diagonal slot
                    }
                    // movement: rows 2 .. r1 - 1; everything else compares a piece with itself and is skipped
                    const int row = rw + s * G + lrow;
                    const bool mv = row >= (ACTIVITY ? 1 : 2) && row < r1;  // (events are counted from row 1 on)
                    const uint32_t cw[4] = {cur.x, cur.y, cur.z, cur.w};
                    const uint32_t pw[4] = {mv ? prv.x : cur.x, mv ? prv.y : cur.y, mv ? prv.z : cur.z, mv ? prv.w : cur.w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t pm = pw[w] & keep[w], cm = cw[w] & keep[w];
                        if (pm != cm) count_moves4(pm, cm, cbase);
                    }
                    if (last_row) *reinterpret_cast<uint4*>(strip + lq0) = cur;
                }
            }
        }
        __syncthreads();
        if constexpr (ACTIVITY) {
            apply_pair_activity<kV3Threads>(cnt, hist);
            __syncthreads();
            if (threadIdx.x < 256) {
                const uint32_t n = hist[threadIdx.x];
                hist[threadIdx.x] = 0;
                if (n && (int)threadIdx.x < p.C) atomicAdd(&p.activity[(size_t)b * p.C + threadIdx.x], (unsigned long long)n);
            }
            __syncthreads();
            continue;
        }
        double acc = wave_sum(apply_pair_counts<kV3Threads>(cnt, fold));
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int w = 0; w < kV3Waves; ++w) s += red[w];
            p.partial[it] = s;
        }
        // the next item's counting cannot pass the barrier after it before thread 0 has read red[]; its
        // counter updates touch entries this pass has already zeroed or left at zero
    }
}

// Rows longer than 512 bytes under a grid metric (no table).  Written for a low vector-instruction
// count, which is what bounds it once nothing is looked up: row and gate loads use wave-uniform base
// pointers plus a fixed per-lane offset; strips sit at a pitch of 1024 bytes; the previous row is
// carried as its masked x and y fields, with the y field left where it sits in the byte (the sum comes
// out scaled by 2^xbits and is scaled back once per item); the "qubit exists" byte masks of a ragged
// row are folded into the field masks.  WRAP (some dimension is a ring) takes the general per-byte path.
constexpr int kGridStrip = 1024;

template <bool WRAP, int AHEAD>
__global__ __launch_bounds__(kV3Threads) void k_cost_v3_grid_row(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    static_assert(kRowEdgeChunks == 4, "four gate chunks fill one dword of endpoints");
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int ldq = p.ldq;
    uint8_t* strip = smem + wave * kGridStrip;
    const bool lane_on = lane * 16 < ldq;
    const int lane16 = lane_on ? lane * 16 : 0;
    const int nvalid_row = min(16, max(0, p.Q - lane * 16));
    uint32_t keep[4], kx[4], ky[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int nb = min(4, max(0, nvalid_row - 4 * w));
        keep[w] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
        kx[w] = keep[w] & p.gxm;
        ky[w] = keep[w] & p.gyi;
    }
    const uint32_t* sched_lane = p.edges_sched + lane;

    for (int64_t it = (int64_t)blockIdx.x * kV3Waves + wave; it < p.nitems; it += (int64_t)gridDim.x * kV3Waves) {
        const int chunk = (int)(it / p.B);
        const int64_t b = it - (int64_t)chunk * p.B;
        const int r0 = chunk * p.rows_per_item;
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const uint8_t* base = p.A + (size_t)b * p.Tp * ldq;
        uint32_t ax = 0, ay = 0;
        GridAcc gacc;

        auto fetch_row = [&](int row) -> uint4 {
            const uint8_t* rp = base + (size_t)min(row, r1 - 1) * ldq;  // wave-uniform
            return load_row16(rp + lane16);
        };
        auto fetch_gate = [&](int row, int c) -> uint32_t {
            const uint32_t* gp = sched_lane + p.sptr[min(row, r1 - 1)];  // wave-uniform + lane
            return gp[c * kWave];
        };

        uint4 prv = make_uint4(0, 0, 0, 0);
        if (r0 >= 1) prv = fetch_row(r0 - 1);
        // non-WRAP: masked x and y fields of the previous row; WRAP: px is the whole masked row
        const uint32_t* mx = WRAP ? keep : kx;
        uint32_t px[4] = {prv.x & mx[0], prv.y & mx[1], prv.z & mx[2], prv.w & mx[3]};
        uint32_t py[4] = {prv.x & ky[0], prv.y & ky[1], prv.z & ky[2], prv.w & ky[3]};
        uint4 ring[AHEAD];
        uint32_t gring[AHEAD][kRowEdgeChunks];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            ring[k] = fetch_row(r0 + k);
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(r0 + k, c);
        }

        for (int g0 = r0; g0 < r1; g0 += AHEAD) {
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                const int row = g0 + k;
                if (row >= r1) break;
                const uint4 cur = ring[k];
                uint32_t gates[kRowEdgeChunks];
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gates[c] = gring[k][c];
                ring[k] = fetch_row(row + AHEAD);
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(row + AHEAD, c);

                if (lane_on) *reinterpret_cast<uint4*>(strip + lane * 16) = cur;
                const int s0 = p.sptr[row];
                const int nch = (p.sptr[row + 1] - s0) / kWave;
                // endpoint gathers of the row's gates; a chunk the row does not have stays (0, 0).  Rows
                // with all four chunks take the branch-free path: eight gathers in flight, one wait.
                uint32_t au[kRowEdgeChunks], av[kRowEdgeChunks];
                if (nch >= kRowEdgeChunks) {
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {
                        au[c] = strip[gates[c] & 0xFFFFu];
                        av[c] = strip[gates[c] >> 16];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) {
                        au[c] = 0;
                        av[c] = 0;
                        if (c < nch) {
                            au[c] = strip[gates[c] & 0xFFFFu];
                            av[c] = strip[gates[c] >> 16];
                        }
                    }
                }
                const uint32_t cw[4] = {cur.x, cur.y, cur.z, cur.w};
                if constexpr (WRAP) {
                    if (row >= 2) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) gacc.add4(p, px[w], cw[w] & keep[w]);
                    }
#pragma unroll
                    for (int w = 0; w < 4; ++w) px[w] = cw[w] & keep[w];  // WRAP carries the whole masked row in px
                } else {
                    uint32_t cx[4], cy[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        cx[w] = cw[w] & kx[w];
                        cy[w] = cw[w] & ky[w];
                    }
                    if (row >= 2) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            ax = __builtin_amdgcn_sad_u8(px[w], cx[w], ax);
                            ay = __builtin_amdgcn_sad_u8(py[w], cy[w], ay);
                        }
                    }
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        px[w] = cx[w];
                        py[w] = cy[w];
                    }
                }
                {
                    const uint32_t U = au[0] | au[1] << 8 | au[2] << 16 | au[3] << 24;
                    const uint32_t V = av[0] | av[1] << 8 | av[2] << 16 | av[3] << 24;
                    if constexpr (WRAP) {
                        gacc.add4(p, U, V);
                    } else {
                        ax = __builtin_amdgcn_sad_u8(U & p.gxm, V & p.gxm, ax);
                        ay = __builtin_amdgcn_sad_u8(U & p.gyi, V & p.gyi, ay);
                    }
                }
                for (int c = kRowEdgeChunks; c < nch; ++c) {
                    const uint32_t g = sched_lane[s0 + c * kWave];
                    const uint32_t U = strip[g & 0xFFFFu], V = strip[g >> 16];
                    if constexpr (WRAP) {
                        gacc.add4(p, U, V);
                    } else {
                        ax = __builtin_amdgcn_sad_u8(U & p.gxm, V & p.gxm, ax);
                        ay = __builtin_amdgcn_sad_u8(U & p.gyi, V & p.gyi, ay);
                    }
                }
            }
        }
        double s;
        if constexpr (WRAP) s = gacc.total(p);
        else s = (double)p.gwx * wave_sum(ax) + (double)p.gwy * (wave_sum(ay) / (double)(1u << p.gxb));
        if (lane == 0) p.partial[it] = s;
    }
}

// Closed-form distances on rows of 1040 to 2048 bytes (NP = 2): k_cost_v3_grid_row with NP 1 KB pieces of the row per wave (a
// lane holds piece (i * 64 + lane) of 16 bytes for i < NP, so every load instruction reads 1 KB in a row), the whole
// row parked in the wave's strip for the gate gathers, the row before in registers: no barrier, no workgroup-wide
// step.  Only the row's last piece can be ragged or missing for some lanes; which i that is, is the same for the
// whole wave.  The first 8 gate chunks of a row are loaded together with it (the gate list is readable 8 chunks past
// its end), two rows ahead.
constexpr int kGridNThreads = 512;

template <bool WRAP, int NP>
__global__ __launch_bounds__(kGridNThreads) void k_cost_v3_grid_rown(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int AHEAD = 2;
    constexpr int kWavesN = kGridNThreads / kWave;
    constexpr int NG = 8;  // gate chunks that travel with a row
    static_assert(NG * QP_SCHED_CHUNK <= QP_SCHED_TAIL, "the prefetched chunks of the last slice stay inside the list's tail");
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int ldq = p.ldq;
    const int npieces = ldq / 16;
    const int i_last = (npieces - 1) / kWave, l_last = (npieces - 1) % kWave;  // the row's last piece
    uint8_t* strip = smem + (size_t)wave * NP * 1024;
    const bool on_last = lane <= l_last;  // the lane has a piece i_last
    uint32_t keep[4];                     // byte masks of piece i_last: all ones except in the row's last piece
    {
        const int nvalid = lane == l_last ? p.Q - (npieces - 1) * 16 : 16;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int nb = min(4, max(0, nvalid - 4 * w));
            keep[w] = !on_last ? 0u : nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
        }
    }
    const uint32_t* sched_lane = p.edges_sched + lane;

    for (int64_t it = (int64_t)blockIdx.x * kWavesN + wave; it < p.nitems; it += (int64_t)gridDim.x * kWavesN) {
        const int chunk = (int)(it / p.B);
        const int64_t b = it - (int64_t)chunk * p.B;
        const int r0 = chunk * p.rows_per_item;
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const uint8_t* base = p.A + (size_t)b * p.Tp * ldq;
        uint32_t ax = 0, ay = 0;
        GridAcc gacc;

        // piece i of a row (a piece the lane does not have: the row's first bytes, masked away by keep)
        auto fetch_piece = [&](int row, int i) -> uint4 {
            const uint8_t* rp = base + (size_t)min(row, r1 - 1) * ldq;  // wave-uniform
            const int off = (i < i_last || on_last) ? (i * kWave + lane) * 16 : 0;
            return load_row16(rp + off);
        };
        auto fetch_gate = [&](int row, int c) -> uint32_t {
            const uint32_t* gp = sched_lane + p.sptr[min(row, r1 - 1)];  // wave-uniform + lane
            return gp[c * kWave];
        };
        // a row's words as they are compared: piece i_last masked, pieces past it absent
        auto masked = [&](const uint4& v, int i, uint32_t (&o)[4]) {
            o[0] = v.x;
            o[1] = v.y;
            o[2] = v.z;
            o[3] = v.w;
            if (i == i_last) {
#pragma unroll
                for (int w = 0; w < 4; ++w) o[w] &= keep[w];
            }
        };

        uint32_t px[NP][4], py[WRAP ? 1 : NP][4];  // non-WRAP: x and y fields of the row before; WRAP: its masked words in px
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            uint32_t o[4] = {0, 0, 0, 0};
            if (r0 >= 1 && i <= i_last) masked(fetch_piece(r0 - 1, i), i, o);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                px[i][w] = WRAP ? o[w] : o[w] & p.gxm;
                if constexpr (!WRAP) py[i][w] = o[w] & p.gyi;
            }
        }
        uint4 ring[AHEAD][NP];
        uint32_t gring[AHEAD][NG];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
#pragma unroll
            for (int i = 0; i < NP; ++i) ring[k][i] = fetch_piece(r0 + k, min(i, i_last));
#pragma unroll
            for (int c = 0; c < NG; ++c) gring[k][c] = fetch_gate(r0 + k, c);
        }

        for (int g0 = r0; g0 < r1; g0 += AHEAD) {
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                const int row = g0 + k;
                if (row >= r1) break;
                uint4 cur[NP];
                uint32_t gates[NG];
#pragma unroll
                for (int i = 0; i < NP; ++i) cur[i] = ring[k][i];
#pragma unroll
                for (int c = 0; c < NG; ++c) gates[c] = gring[k][c];
#pragma unroll
                for (int i = 0; i < NP; ++i) ring[k][i] = fetch_piece(row + AHEAD, min(i, i_last));
#pragma unroll
                for (int c = 0; c < NG; ++c) gring[k][c] = fetch_gate(row + AHEAD, c);

#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if (i < i_last || (i == i_last && on_last)) *reinterpret_cast<uint4*>(strip + (i * kWave + lane) * 16) = cur[i];
                const int s0 = p.sptr[row];
                const int nch = (p.sptr[row + 1] - s0) / kWave;
                // the row's gates, four chunks to a packed word of first and of second endpoints; a chunk the row does
                // not have is (qubit 0, qubit 0)
#pragma unroll
                for (int c0 = 0; c0 < NG; c0 += 4) {
                    if (c0 >= nch) break;
                    uint32_t au[4], av[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t g = c0 + c < nch ? gates[c0 + c] : 0u;
                        au[c] = strip[g & 0xFFFFu];
                        av[c] = strip[g >> 16];
                    }
                    const uint32_t U = au[0] | au[1] << 8 | au[2] << 16 | au[3] << 24;
                    const uint32_t V = av[0] | av[1] << 8 | av[2] << 16 | av[3] << 24;
                    if constexpr (WRAP) {
                        gacc.add4(p, U, V);
                    } else {
                        ax = __builtin_amdgcn_sad_u8(U & p.gxm, V & p.gxm, ax);
                        ay = __builtin_amdgcn_sad_u8(U & p.gyi, V & p.gyi, ay);
                    }
                }
                for (int c = NG; c < nch; ++c) {
                    const uint32_t g = sched_lane[s0 + c * kWave];
                    const uint32_t U = strip[g & 0xFFFFu], V = strip[g >> 16];
                    if constexpr (WRAP) {
                        gacc.add4(p, U, V);
                    } else {
                        ax = __builtin_amdgcn_sad_u8(U & p.gxm, V & p.gxm, ax);
                        ay = __builtin_amdgcn_sad_u8(U & p.gyi, V & p.gyi, ay);
                    }
                }
                // movement: rows 2 .. r1 - 1, row before -> this row
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    if (i > i_last) break;
                    uint32_t cw[4];
                    masked(cur[i], i, cw);
                    if constexpr (WRAP) {
                        if (row >= 2) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) gacc.add4(p, px[i][w], cw[w]);
                        }
#pragma unroll
                        for (int w = 0; w < 4; ++w) px[i][w] = cw[w];
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const uint32_t cx = cw[w] & p.gxm, cy = cw[w] & p.gyi;
                            if (row >= 2) {
                                ax = __builtin_amdgcn_sad_u8(px[i][w], cx, ax);
                                ay = __builtin_amdgcn_sad_u8(py[i][w], cy, ay);
                            }
                            px[i][w] = cx;
                            py[i][w] = cy;
                        }
                    }
                }
            }
        }
        double s;
        if constexpr (WRAP) s = gacc.total(p);
        else s = (double)p.gwx * wave_sum(ax) + (double)p.gwy * (wave_sum(ay) / (double)(1u << p.gxb));
        if (lane == 0) p.partial[it] = s;
    }
}

// Rows longer than 1024 bytes.  One workgroup of 16 waves owns an item.  Its threads form 1024 / P groups
// (P = pieces of 16 bytes per row, rounded up to a power of two, at most 1024); a group takes a run of the
// item's rows, thread j of it owns pieces j, j + 1024, ... (NP of them: 1 for rows up to 16 KB, kWidePieces
// for rows up to 64 KB) and carries the previous row's pieces in registers, so rows are read once, 16 bytes
// per lane, with the loads of the next AHEAD rows in flight.  Only qubits that moved are looked up; under a
// grid metric nothing is (packed coordinate differences, as in the 1 KB kernel).  The groups step through
// their runs together: each parks its row in an LDS strip, the workgroup meets at a barrier, and the group's
// threads gather the endpoints of the row's gates from the strip (byte gathers from device memory run at
// one address per clock per CU and were 4x slower than the row stream).  NBUF = 2 strips per group alternate,
// one barrier per step; NBUF = 1 needs a second barrier.  Every load a step consumes was issued at least one
// step earlier and nothing is loaded under a branch, so the waits the compiler counts leave the younger
// loads in flight: rows AHEAD steps ahead together with their slice bounds, the first 8 gate words of a
// thread (two 16-byte loads from the padded gate list) AHEAD / 2 steps ahead.  Without the padded gate list
// the gates come from the edge records, loaded where they are used (slower, same sums up to their order).
// TABMODE: kTabLdsC (table staged in LDS at pitch C), kTabGlobal, kTabGrid.  THREADS: 1024, or 512 for rows of up
// to 8 KB when two workgroups then fit a CU (one alone has nothing to overlap the start and the end of an
// item with); wherever the text above says 1024, read THREADS.
constexpr int kWidePieces = 4;
constexpr int kWideGateLoads = 2;  // 16-byte loads of gate words per thread and row that are prefetched

template <int KIND, int TABMODE, int NP, int NBUF, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 512 ? 6 : 4) void k_cost_v3_wide(V3Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = typename Tab<KIND>::elem;
    using Acc = typename Tab<KIND>::acc;
    constexpr int AHEAD = NP == 1 ? 4 : 2;
    constexpr int GA = AHEAD / 2;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t tab_off = 0;
    const E* tab = static_cast<const E*>(p.table);
    if constexpr (TABMODE == kTabLdsC) {
        tab_off = stage_table(smem, p.table, p.table_bytes);
        tab = reinterpret_cast<const E*>(smem);
    }
    // kTabPairs: the counters of the unordered core pairs come first (p.table is the symmetrised, folded table)
    const uint32_t cbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem - 0x20000u);
    if constexpr (TABMODE == kTabPairs) {
        static_assert(KIND == QP_DT_F64 && NP == 1, "pair counts stand in for an fp64 table, rows of up to 16 KB");
        tab_off = (uint32_t)(kPairSlots * sizeof(uint32_t));
        for (int k = threadIdx.x; k < kPairSlots; k += THREADS) reinterpret_cast<uint32_t*>(smem)[k] = 0;
        __syncthreads();
    }
    uint8_t* strips = smem + tab_off;
    double* red = reinterpret_cast<double*>(strips);  // the per-wave sums of an item reuse the strips' first bytes
    const uint32_t C = (uint32_t)p.C, last = C * C - 1u;  // a core id >= C (caller's error) never reads past a table
    auto pair_at = [&](uint32_t from, uint32_t to) -> Acc { return (Acc)tab[min(from * C + to, last)]; };
    const int npieces = p.ldq / 16;
    int P = 128;  // rows here have more than 64 pieces, so a group is made of whole waves
    while (P < npieces && P < THREADS) P *= 2;
    const int groups = THREADS / P, tin = threadIdx.x % P;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / P);
    const int strip_bytes = P * 16 * NP;
    // the padded gate list, or (without one) any readable 16 bytes whose contents are ignored
    const bool sched = p.edges_sched != nullptr && p.sptr != nullptr;
    const uint32_t* gsrc = sched ? p.edges_sched : reinterpret_cast<const uint32_t*>(p.A);
    const int32_t* bsrc = sched ? p.sptr : p.sep;
    const int sched_end = sched ? p.sptr[p.T] : 0;  // zero words follow the last slice
    // bytes of qubits that do not exist are zeroed in every piece (keep[k][w]: dword w of piece k); a thread
    // without a k-th piece re-reads piece 0, masks all of it and writes nothing to the strip
    uint32_t keep[NP][4];
    int off[NP];
    bool have[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int piece = tin + k * THREADS;
        have[k] = piece < npieces;
        off[k] = have[k] ? piece * 16 : 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int nb = have[k] ? min(4, max(0, p.Q - (piece * 16 + 4 * w))) : 0;
            keep[k][w] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
        }
    }
    for (int64_t it = blockIdx.x; it < p.nitems; it += gridDim.x) {
        const int chunk = (int)(it / p.B);
        const int64_t b = it - (int64_t)chunk * p.B;
        const int r0 = chunk * p.rows_per_item;
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const uint8_t* base = p.A + (size_t)b * p.Tp * p.ldq;
        Acc acc = 0;
        GridAcc gacc;
        // this group's run of rows; runs differ by at most one row, every group takes nsteps steps
        const int g0 = r0 + (int)((int64_t)(r1 - r0) * grp / groups), g1 = r0 + (int)((int64_t)(r1 - r0) * (grp + 1) / groups);
        const int nsteps = (r1 - r0 + groups - 1) / groups;
        auto fetch = [&](int row, int k) -> uint4 { return load_row16(base + (size_t)min(row, r1 - 1) * p.ldq + off[k]); };
        auto fetch_bounds = [&](int row) -> int2 {
            const int32_t* bp = bsrc + min(row, r1 - 1);
            return make_int2(bp[0], bp[1]);
        };
        auto fetch_gates = [&](int slice_start, int j) -> uint4 {  // words 4 * (tin + j * P) .. + 3 of a slice
            const int w = sched ? min(slice_start + 4 * (tin + j * P), sched_end) : 0;
            return *reinterpret_cast<const uint4*>(gsrc + w);
        };
        auto gate_cost = [&](uint32_t u, uint32_t v) {
            if constexpr (TABMODE == kTabGrid) gacc.add4(p, u, v);
            else if constexpr (TABMODE == kTabPairs) count_gate_pair2(u, v, cbase);  // upper halves: (0, 0), a diagonal slot
            else if (u != v) acc += pair_at(u, v);
        };
        uint32_t prev[NP][4];
        uint4 ring[AHEAD][NP];
        int2 bring[AHEAD];
        uint4 gring[GA][kWideGateLoads];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const uint4 v = g0 >= 1 ? fetch(g0 - 1, k) : make_uint4(0, 0, 0, 0);
            prev[k][0] = v.x & keep[k][0];
            prev[k][1] = v.y & keep[k][1];
            prev[k][2] = v.z & keep[k][2];
            prev[k][3] = v.w & keep[k][3];
        }
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) bring[a] = fetch_bounds(g0 + a);
#pragma unroll
        for (int a = 0; a < AHEAD; ++a)
#pragma unroll
            for (int k = 0; k < NP; ++k) ring[a][k] = fetch(g0 + a, k);
#pragma unroll
        for (int a = 0; a < GA; ++a)
#pragma unroll
            for (int j = 0; j < kWideGateLoads; ++j) gring[a][j] = fetch_gates(bring[a].x, j);
        for (int s0 = 0; s0 < nsteps; s0 += AHEAD) {
#pragma unroll
            for (int a = 0; a < AHEAD; ++a) {
                // no early exit from the unrolled steps (the register copies where such paths meet would wait
                // for loads just issued): a step past the last one loads clamped addresses and does nothing else
                const int step = s0 + a;
                const int row = g0 + step;
                const bool active = row < g1;
                uint8_t* strip = strips + ((step & (NBUF - 1)) * groups + grp) * strip_bytes;
                const int sb = bring[a].x, se = bring[a].y;
                const int nw = active && sched ? se - sb : 0;  // gate words of this row
                uint32_t cur[NP][4];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    cur[k][0] = ring[a][k].x & keep[k][0];
                    cur[k][1] = ring[a][k].y & keep[k][1];
                    cur[k][2] = ring[a][k].z & keep[k][2];
                    cur[k][3] = ring[a][k].w & keep[k][3];
                }
                uint4 gw[kWideGateLoads];
#pragma unroll
                for (int j = 0; j < kWideGateLoads; ++j) {
                    gw[j] = gring[a % GA][j];
                    gring[a % GA][j] = fetch_gates(bring[(a + GA) % AHEAD].x, j);  // gates of row + GA
                }
#pragma unroll
                for (int k = 0; k < NP; ++k) ring[a][k] = fetch(row + AHEAD, k);
                bring[a] = fetch_bounds(row + AHEAD);
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (active && have[k]) *reinterpret_cast<uint4*>(strip + off[k]) = make_uint4(cur[k][0], cur[k][1], cur[k][2], cur[k][3]);
                if (active && row >= 2) {  // movement is charged from row 2 on
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const uint32_t pf = prev[k][w], cf = cur[k][w];
                            if constexpr (TABMODE == kTabGrid) {
                                gacc.add4(p, pf, cf);
                            } else if constexpr (TABMODE == kTabPairs) {
                                if (pf != cf) count_moves4(pf, cf, cbase);
                            } else if (pf != cf) {
                                if constexpr (TABMODE == kTabGlobal) {  // a look-up is an L2 gather: only for qubits that moved
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const uint32_t f = (pf >> (8 * j)) & 0xFFu, c = (cf >> (8 * j)) & 0xFFu;
                                        if (f != c) acc += pair_at(f, c);
                                    }
                                } else {  // the table's diagonal is zero: the four look-ups of a dword go out together
                                    Acc d4[4];
#pragma unroll
                                    for (int j = 0; j < 4; ++j) d4[j] = pair_at((pf >> (8 * j)) & 0xFFu, (cf >> (8 * j)) & 0xFFu);
#pragma unroll
                                    for (int j = 0; j < 4; ++j) acc += d4[j];
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < NP; ++k)
#pragma unroll
                    for (int w = 0; w < 4; ++w) prev[k][w] = cur[k][w];
                __syncthreads();  // the rows of this step are in their strips
                if (4 * tin < nw) {
#pragma unroll
                    for (int j = 0; j < kWideGateLoads; ++j) {
                        if (j > 0 && 4 * (tin + j * P) >= nw) continue;  // most rows have fewer than 4 * P gates
                        const uint32_t g4[4] = {gw[j].x, gw[j].y, gw[j].z, gw[j].w};
                        uint32_t au[4], av[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            // a word past the slice's end belongs to a later slice: qubit 0 with itself instead
                            const uint32_t g = 4 * (tin + j * P) + c < nw ? g4[c] : 0u;
                            au[c] = strip[g & 0xFFFFu];
                            av[c] = strip[g >> 16];
                        }
                        if constexpr (TABMODE == kTabGrid) {
                            gacc.add4(p, au[0] | au[1] << 8 | au[2] << 16 | au[3] << 24, av[0] | av[1] << 8 | av[2] << 16 | av[3] << 24);
                        } else if constexpr (TABMODE == kTabPairs) {
                            count_gate_pair2(au[0] | au[1] << 16, av[0] | av[1] << 16, cbase);
                            count_gate_pair2(au[2] | au[3] << 16, av[2] | av[3] << 16, cbase);
                        } else {
                            Acc d4[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) d4[c] = pair_at(au[c], av[c]);
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc += d4[c];
                        }
                    }
                    for (int w = 4 * kWideGateLoads * P + tin; w < nw; w += P) {  // rows with more than 8 * P gates
                        const uint32_t g = gsrc[sb + w];
                        gate_cost(strip[g & 0xFFFFu], strip[g >> 16]);
                    }
                }
                if (!sched && active) {
                    for (int e = sb + tin; e < se; e += P) {
                        const QpEdge ed = p.edges[e];
                        gate_cost(strip[ed.u], strip[ed.v]);
                    }
                }
                if constexpr (NBUF == 1) __syncthreads();  // the one strip is free again
            }
        }
        double s;
        if constexpr (TABMODE == kTabGrid) {
            s = gacc.total(p);
        } else if constexpr (TABMODE == kTabPairs) {
            __syncthreads();  // every count of the item is in
            s = wave_sum(apply_pair_counts<THREADS>(reinterpret_cast<uint32_t*>(smem), static_cast<const double*>(p.table)));
        } else {
            s = wave_sum(acc);
        }
        __syncthreads();  // nobody reads a strip any more
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int w = 0; w < (THREADS / kWave); ++w) tot += red[w];
            p.partial[it] = tot;
        }
        __syncthreads();  // the sums are read before the next item's rows overwrite them
    }
}

// out[b] = partial[0*B + b] + partial[1*B + b] + ... in chunk order.
// scale: 1, or the real number a scaled grid metric multiplies its hop counts by (the partial sums are then integers)
__global__ void k_sum_chunks(const double* partial, int64_t B, int32_t nchunks, double* out, double scale = 1.0) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int32_t c = 0; c < nchunks; ++c) s += partial[(size_t)c * B + b];
    out[b] = scale == 1.0 ? s : s * scale;
}

// ------------------------------------------------------------------------------------------------
// Cost V1 / V2 / V4: sparse partitions walked visit by visit.
// ------------------------------------------------------------------------------------------------

struct V2Args {
    const uint8_t* A;
    const int32_t* svp;
    const QpVisit* visits;
    const void* table;
    double* partial;
    int64_t B, nitems;
    int32_t T, Tp, ldq, C, rows_per_item, mode;
    uint32_t table_bytes;
};

constexpr int kV2Unroll = 2;  // visits per lane in flight
constexpr int kV2Group = 4;   // partitions that share one pass over the visit records

// One wave = (chunk of slices, kV2Group consecutive partitions).  The visit records are the same for every
// partition, so each is loaded once and applied to the whole group; with kV2Unroll visits per lane that
// puts 3 * kV2Unroll * kV2Group independent placement bytes in flight before the first table look-up.
template <int KIND, bool TAB_LDS, int THREADS>
__global__ __launch_bounds__(THREADS) void k_cost_v2(V2Args p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = typename Tab<KIND>::elem;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    constexpr int nwaves = THREADS / kWave;
    const E* tab;
    if constexpr (TAB_LDS) {
        stage_table(smem, p.table, p.table_bytes);
        tab = reinterpret_cast<const E*>(smem);
    } else {
        tab = static_cast<const E*>(p.table);
    }
    const int C = p.C;
    const uint32_t last = (uint32_t)C * (uint32_t)C - 1u;  // core ids >= C (caller's error) never index past the table
    const int64_t ngroups = (p.B + kV2Group - 1) / kV2Group;
    const size_t pstride = (size_t)p.Tp * p.ldq;
    for (int64_t it = (int64_t)blockIdx.x * nwaves + wave; it < p.nitems; it += (int64_t)gridDim.x * nwaves) {
        const int chunk = (int)(it / ngroups);
        const int64_t b0 = (it - (int64_t)chunk * ngroups) * kV2Group;
        const int r0 = chunk * p.rows_per_item;
        const int r1 = min(p.T, r0 + p.rows_per_item);
        const uint8_t* base[kV2Group];  // a group that runs past B repeats the last partition and drops the copies
#pragma unroll
        for (int g = 0; g < kV2Group; ++g) base[g] = p.A + (size_t)min(b0 + g, p.B - 1) * pstride;
        double acc[kV2Group];
        bool bad[kV2Group];
#pragma unroll
        for (int g = 0; g < kV2Group; ++g) {
            acc[g] = 0.0;
            bad[g] = false;
        }
        const int vend = p.svp[r1];
        for (int i0 = p.svp[r0]; i0 < vend; i0 += kV2Unroll * kWave) {
            QpVisit v[kV2Unroll];
#pragma unroll
            for (int u = 0; u < kV2Unroll; ++u) {
                const int i = i0 + u * kWave + lane;
                if (i < vend) {
                    v[u] = p.visits[i];
                } else {  // reads the item's first row, which exists, and adds nothing
                    v[u].q = 0;
                    v[u].partner = 0xFFFFu;
                    v[u].t = r0;
                    v[u].prev_t = -1;
                }
            }
            uint32_t cur[kV2Unroll][kV2Group], oth[kV2Unroll][kV2Group], prv[kV2Unroll][kV2Group];
#pragma unroll
            for (int u = 0; u < kV2Unroll; ++u) {
                const size_t row = (size_t)v[u].t * p.ldq;
                const size_t o_cur = row + v[u].q;
                const size_t o_oth = row + (v[u].partner != 0xFFFFu ? v[u].partner : v[u].q);
                const size_t o_prv = (size_t)(v[u].prev_t >= 0 ? v[u].prev_t : v[u].t) * p.ldq + v[u].q;
#pragma unroll
                for (int g = 0; g < kV2Group; ++g) {
                    cur[u][g] = base[g][o_cur];
                    oth[u][g] = base[g][o_oth];
                    prv[u][g] = base[g][o_prv];
                }
            }
#pragma unroll
            for (int u = 0; u < kV2Unroll; ++u) {
#pragma unroll
                for (int g = 0; g < kV2Group; ++g) {
                    const uint32_t c = cur[u][g], o = oth[u][g], q = prv[u][g];
                    if (v[u].partner != 0xFFFFu) {
                        if (c == QP_ABSENT || o == QP_ABSENT) {
                            if (p.mode != QP_V1) bad[g] = true;  // KeyError in the reference
                        } else {
                            acc[g] += (double)tab[min(c * C + o, last)];
                        }
                    }
                    if (v[u].prev_t >= 0 && q != QP_ABSENT && c != q) {
                        if (c == QP_ABSENT) bad[g] = true;  // distance to None: TypeError in the reference
                        else acc[g] += (double)(p.mode == QP_V4 ? tab[min(c * C + q, last)] : tab[min(q * C + c, last)]);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < kV2Group; ++g) {
            const double s = wave_sum(acc[g]);
            const bool any_bad = __builtin_amdgcn_ballot_w64(bad[g]) != 0;
            if (lane == 0 && b0 + g < p.B)
                p.partial[(size_t)chunk * p.B + b0 + g] = any_bad ? std::numeric_limits<double>::infinity() : s;
        }
    }
}

// Cost V1 / V2 / V4 as a row walk, for circuits that visit most qubits in most slices (rows of 513 to 1024 bytes, 16-bit
// table).  One wave owns a partition and reads its rows once, 16 bytes per lane; the core every qubit had at its last
// visit stays in four registers per lane (0xFF = not visited yet, or on no core then).  vmask holds, per slice, 0xFF in
// the bytes of the qubits the slice visits, so
//     from = (vmask & last) | (~vmask & row)    equals the row wherever nothing is visited,
//     last = (vmask & row)  | (~vmask & last)   afterwards,
// and the movement term of the whole row is the packed look-up of (from, row) pairs the dense kernels use: pairs of
// equal cores read the zero diagonal.  The table is staged at pitch 256 with row and column 255 zero, so a qubit
// without a previous core adds nothing; the cases the reference fails on (a visited qubit that has a previous core
// but no current one, a counted gate with an operand on no core: inf in the results) are found by byte tests and
// end in inf.  Gates come from the per-slice list of the dense kernels, (first operand | second << 16), padding words
// zero.  Sums are integers: flushed into 64 bits every few rows.
struct V2WalkArgs {
    const uint8_t* A;
    const uint8_t* vmask;
    const uint32_t* sched;
    const int32_t* sptr;
    const void* table;
    const int16_t* vprev;  // needed when a partition is split (rows_per_item < T)
    double* partial;       // [chunk][B]
    int64_t B, nitems;
    int32_t T, Tp, ldq, C, rows_per_item;
    int32_t gate_errors;  // non-zero: a counted gate with an operand on no core is an error (V2, V4); zero: it adds nothing (V1)
    // k_cost_v2_walk_parts: the gate records (first dword of each = first operand | second << 16) and where every
    // slice's start; 16-byte pieces per row, partitions per wave step, groups of that many partitions
    const QpEdge* edges;
    const int32_t* sep;
    int32_t nE, lpr, G;
    int64_t ngroups;
    int32_t id_bits, rep_shift;  // k_cost_v2_walk_parts: table index = from << id_bits | to; 2^rep_shift copies of every entry
};

template <bool V4, int AHEAD>
__global__ __launch_bounds__(kV3Threads) void k_cost_v2_walk(V2WalkArgs p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = uint16_t;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    {
        const E* src = static_cast<const E*>(p.table);
        E* dst = reinterpret_cast<E*>(smem);
        for (int k = threadIdx.x; k < 256 * 256; k += blockDim.x) {
            const int r = k >> 8, c = k & 255;
            dst[k] = (r < p.C && c < p.C) ? src[r * p.C + c] : E(0);
        }
        __syncthreads();
    }
    const E* tab = reinterpret_cast<const E*>(smem);
    const uint32_t tbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem);
    const int ldq = p.ldq;
    uint8_t* strip = smem + 256 * 256 * sizeof(E) + (size_t)wave * ldq;
    const bool lane_on = lane * 16 < ldq;
    const int lane16 = lane_on ? lane * 16 : 0;
    const uint32_t lane_mask = lane_on ? ~0u : 0u;
    auto four = [&](uint32_t from, uint32_t to) -> uint32_t {
        E e0, e1, e2, e3;
        entries_at_halves<E>(__builtin_amdgcn_perm(from, to, 0x05010400u), tbase, e0, e1);
        entries_at_halves<E>(__builtin_amdgcn_perm(from, to, 0x07030602u), tbase, e2, e3);
        return ((uint32_t)e0 + e1) + ((uint32_t)e2 + e3);
    };
    const uint32_t* sched_lane = p.sched + lane;

    for (int64_t it = (int64_t)blockIdx.x * kV3Waves + wave; it < p.nitems; it += (int64_t)gridDim.x * kV3Waves) {
        const int chunk = (int)(it / p.B);
        const int64_t b = it - (int64_t)chunk * p.B;
        const int r0 = chunk * p.rows_per_item;
        const int T = min(p.T, r0 + p.rows_per_item);  // the item's rows are r0 .. T - 1
        const uint8_t* base = p.A + (size_t)b * p.Tp * ldq;
        auto fetch_row = [&](int row) -> uint4 { return load_row16(base + (size_t)min(row, T - 1) * ldq + lane16); };
        auto fetch_mask = [&](int row) -> uint4 {  // lanes past the end of the row visit nothing
            const uint4 m = *reinterpret_cast<const uint4*>(p.vmask + (size_t)min(row, T - 1) * ldq + lane16);
            return make_uint4(m.x & lane_mask, m.y & lane_mask, m.z & lane_mask, m.w & lane_mask);
        };
        // slice starts in the gate list: vector loads through an index the compiler cannot prove uniform, carried
        // with the ring and made scalar where they are used (as scalar loads each would be waited for together with
        // every LDS operation in flight); the start of slice row + AHEAD + 1 is requested a row before it is needed
        int opaque_zero = 0;
        asm volatile("" : "+v"(opaque_zero));
        auto slice_start = [&](int row) -> int { return p.sptr[min(row, T) + opaque_zero]; };
        auto fetch_gate = [&](int start, int c) -> uint32_t { return sched_lane[start + c * kWave]; };
        uint32_t st[4];
        walk_state_at(base, p.vprev, r0, ldq, lane16, st);
        uint32_t l0 = st[0], l1 = st[1], l2 = st[2], l3 = st[3];
        unsigned long long total = 0;
        bool bad = false;
        uint4 ring[AHEAD], mring[AHEAD];
        uint32_t gring[AHEAD][kRowEdgeChunks];
        int sring[AHEAD], ering[AHEAD];
        int sp_next = slice_start(r0);
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            sring[k] = sp_next;
            sp_next = slice_start(r0 + k + 1);
            ering[k] = sp_next;
            ring[k] = fetch_row(r0 + k);
            mring[k] = fetch_mask(r0 + k);
            const int start = __builtin_amdgcn_readfirstlane(sring[k]);
#pragma unroll
            for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(start, c);
        }
        for (int g0 = r0; g0 < T; g0 += AHEAD) {
            uint32_t acc = 0;  // at most AHEAD * (16 + gates of a slice / 64 + 1) terms below 2^16 each
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                const int row = g0 + k;
                if (row >= T) break;
                const uint4 cur = ring[k], vm = mring[k];
                uint32_t gates[kRowEdgeChunks];
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) gates[c] = gring[k][c];
                const int s0 = __builtin_amdgcn_readfirstlane(sring[k]);
                const int nch = (__builtin_amdgcn_readfirstlane(ering[k]) - s0) / kWave;
                {
                    const int sp_here = sp_next;             // start of slice row + AHEAD, asked for a row ago
                    sp_next = slice_start(row + AHEAD + 1);  // used a row from now
                    sring[k] = sp_here;
                    ering[k] = sp_next;
                    ring[k] = fetch_row(row + AHEAD);
                    mring[k] = fetch_mask(row + AHEAD);
                    const int start = __builtin_amdgcn_readfirstlane(sp_here);
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) gring[k][c] = fetch_gate(start, c);
                }

                if (lane_on) *reinterpret_cast<uint4*>(strip + lane * 16) = cur;
                uint32_t au[kRowEdgeChunks], av[kRowEdgeChunks];
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) {
                    if (c >= nch) gates[c] = 0u;  // (qubit 0, qubit 0): one core twice, distance zero
                    au[c] = strip[gates[c] & 0xFFFFu];
                    av[c] = strip[gates[c] >> 16];
                }
                // movement of the visited qubits
                const uint32_t f0 = (vm.x & l0) | (~vm.x & cur.x), f1 = (vm.y & l1) | (~vm.y & cur.y);
                const uint32_t f2 = (vm.z & l2) | (~vm.z & cur.z), f3 = (vm.w & l3) | (~vm.w & cur.w);
                if (f0 != cur.x) acc += V4 ? four(cur.x, f0) : four(f0, cur.x);
                if (f1 != cur.y) acc += V4 ? four(cur.y, f1) : four(f1, cur.y);
                if (f2 != cur.z) acc += V4 ? four(cur.z, f2) : four(f2, cur.z);
                if (f3 != cur.w) acc += V4 ? four(cur.w, f3) : four(f3, cur.w);
                if ((bytes_ff(cur.x & vm.x) | bytes_ff(cur.y & vm.y) | bytes_ff(cur.z & vm.z) | bytes_ff(cur.w & vm.w)) != 0)
                    bad = bad || walk_lost(vm.x, l0, cur.x) || walk_lost(vm.y, l1, cur.y) || walk_lost(vm.z, l2, cur.z) ||
                          walk_lost(vm.w, l3, cur.w);
                l0 = (vm.x & cur.x) | (~vm.x & l0);
                l1 = (vm.y & cur.y) | (~vm.y & l1);
                l2 = (vm.z & cur.z) | (~vm.z & l2);
                l3 = (vm.w & cur.w) | (~vm.w & l3);
                // the slice's gates
                uint32_t top = 0;
#pragma unroll
                for (int c = 0; c < kRowEdgeChunks; ++c) top = max(top, max(au[c], av[c]));
                if (top == 0xFFu && p.gate_errors) {
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) bad = bad || (gates[c] != 0u && max(au[c], av[c]) == 0xFFu);
                }
                if (((au[0] ^ av[0]) | (au[1] ^ av[1]) | (au[2] ^ av[2]) | (au[3] ^ av[3])) != 0) {
                    uint32_t s = 0;
#pragma unroll
                    for (int c = 0; c < kRowEdgeChunks; ++c) s += tab[(au[c] << 8) | av[c]];
                    acc += s;
                }
                for (int c = kRowEdgeChunks; c < nch; ++c) {
                    const uint32_t g = sched_lane[s0 + c * kWave];
                    const uint32_t cu = strip[g & 0xFFFFu], cv = strip[g >> 16];
                    bad = bad || (p.gate_errors && g != 0u && max(cu, cv) == 0xFFu);
                    acc += tab[(cu << 8) | cv];
                }
            }
            total += acc;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, kWave);
        const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) p.partial[it] = any_bad ? std::numeric_limits<double>::infinity() : (double)total;
    }
}

// The same walk for rows of at most 512 bytes.  A wave step is row t of G = 64 / (ldq / 16) consecutive partitions (not G
// rows of one partition: the last cores then stay in the registers of the lane that owns the (partition, 16-byte piece),
// with no scan over rows).  The strip holds the G rows side by side (lane * 16 is both where a lane parks its piece and
// partition * ldq + piece * 16); slice t's gates are the same for all G partitions, so lane (partition g, slot k) takes
// gates k, k + ldq / 16, ... of the slice against its own partition's row: every gate instruction serves 64 (gate,
// partition) pairs.  Gates come from the edge records (slice order, no padding), the first kWalkGateChunks chunks of a
// slice loaded together with the row.  Table in LDS: index = from << k | to on the low k bits of the core ids, 2^k > C, so
// 0xFF (no core) lands on row / column 2^k - 1, which is zero like the diagonal; every entry is stored 2^r times side
// by side (16-bit entries widened to dwords, except at k = 8 where there is no room) and a lane reads copy
// lane & (2^r - 1): with 32 dword copies or 16 fp64 copies (up to 31 cores) the look-ups of an instruction can never
// meet in a bank, with fewer only the lanes that share a copy can.  An fp64 table fits up to 127 cores (k = 7, 128 KB).
constexpr int kWalkGateChunks = 6;
constexpr int kWalkPartsLds = 1024 + 768;  // per wave: the strip, then 64 sums and 64 error flags

template <int KIND, bool V4, int AHEAD, bool NATIVE16 = false>
__global__ __launch_bounds__(kV3Threads) void k_cost_v2_walk_parts(V2WalkArgs p) {
    extern __shared__ __align__(16) uint8_t smem[];
    using E = typename Tab<KIND>::elem;
    using Acc = typename Tab<KIND>::acc;
    using S = typename std::conditional<KIND == QP_DT_F64, double, typename std::conditional<NATIVE16, uint16_t, uint32_t>::type>::type;
    typedef __attribute__((address_space(3))) const S lds_s;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int kb = p.id_bits, rs = p.rep_shift;
    const int nslots = (1 << (2 * kb)) << rs;
    {
        const E* src = static_cast<const E*>(p.table);
        S* dst = reinterpret_cast<S*>(smem);
        for (int k = threadIdx.x; k < nslots; k += blockDim.x) {
            const int idx = k >> rs, r = idx >> kb, c = idx & ((1 << kb) - 1);
            dst[k] = (r < p.C && c < p.C) ? (S)src[r * p.C + c] : S(0);
        }
        __syncthreads();
    }
    const size_t tab_bytes = (size_t)nslots * sizeof(S);
    const uint32_t low = (1u << kb) - 1u;
    const uint32_t mask_to = low * 0x00010001u, mask_from = (low << kb) * 0x00010001u;
    const int step_log = rs + (sizeof(S) == 8 ? 3 : sizeof(S) == 4 ? 2 : 1);
    const uint32_t step = 1u << step_log;  // bytes from an entry's copies to the next entry's
    // LDS address of this lane's copy of entry 0
    const uint32_t tlane = (uint32_t)(uintptr_t)(lds_u8*)smem + (uint32_t)(lane & ((1 << rs) - 1)) * (uint32_t)sizeof(S);
    const int ldq = p.ldq, lpr = p.lpr, G = p.G;
    uint8_t* strip = smem + tab_bytes + (size_t)wave * kWalkPartsLds;
    const bool lane_on = lane < G * lpr;
    const int g = lane_on ? lane / lpr : 0, piece = lane_on ? lane - g * lpr : 0;
    const uint32_t lane_mask = lane_on ? ~0u : 0u;
    const uint32_t goff = (uint32_t)(g * ldq);
    auto at = [&](uint32_t from, uint32_t to) -> Acc {
        const uint32_t idx = ((from & low) << kb) | (to & low);
        return (Acc) * (lds_s*)(uintptr_t)(tlane + (idx << step_log));
    };
    // the two entries whose indices are the 16-bit halves of x
    auto two = [&](uint32_t x, S& e0, S& e1) {
        uint32_t a0, a1;
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(step), "v"(tlane));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(step), "v"(tlane));
        e0 = *(lds_s*)(uintptr_t)a0;
        e1 = *(lds_s*)(uintptr_t)a1;
    };
    // four (from, to) pairs of a dword: from << 8 | to in 16-bit halves, squeezed to from << k | to on the ids' low k bits
    auto four = [&](uint32_t from, uint32_t to) -> Acc {
        const uint32_t k01 = __builtin_amdgcn_perm(from, to, 0x05010400u), k23 = __builtin_amdgcn_perm(from, to, 0x07030602u);
        const uint32_t x01 = ((k01 >> (8 - kb)) & mask_from) | (k01 & mask_to);
        const uint32_t x23 = ((k23 >> (8 - kb)) & mask_from) | (k23 & mask_to);
        S e0, e1, e2, e3;
        two(x01, e0, e1);
        two(x23, e2, e3);
        return ((Acc)e0 + (Acc)e1) + ((Acc)e2 + (Acc)e3);
    };
    const uint32_t* words = reinterpret_cast<const uint32_t*>(p.edges);  // records of 8 bytes: the pair is dword 0
    const int last_edge = max(p.nE - 1, 0);

    for (int64_t it = (int64_t)blockIdx.x * kV3Waves + wave; it < p.nitems; it += (int64_t)gridDim.x * kV3Waves) {
        const int chunk = (int)(it / p.ngroups);
        const int64_t b0 = (it - (int64_t)chunk * p.ngroups) * G;
        const int64_t b = min(b0 + g, p.B - 1);  // a group that runs past B repeats the last partition and drops the copies
        const int r0 = chunk * p.rows_per_item;
        const int T = min(p.T, r0 + p.rows_per_item);
        const uint8_t* part = p.A + (size_t)b * p.Tp * ldq;
        const uint8_t* base = part + piece * 16;
        auto fetch_row = [&](int row) -> uint4 { return load_row16(base + (size_t)min(row, T - 1) * ldq); };
        auto fetch_mask = [&](int row) -> uint4 {
            const uint4 m = *reinterpret_cast<const uint4*>(p.vmask + (size_t)min(row, T - 1) * ldq + piece * 16);
            return make_uint4(m.x & lane_mask, m.y & lane_mask, m.z & lane_mask, m.w & lane_mask);
        };
        // Slice starts in the edge records.  As scalar loads each would be waited for together with every LDS
        // operation in flight; so they are loaded through an index the compiler cannot prove uniform, stay in vector
        // registers, travel with the ring and are made scalar where they are used: the start of slice
        // row + AHEAD + 1 is requested a row before it is needed.
        int opaque_zero = 0;
        asm volatile("" : "+v"(opaque_zero));
        auto slice_start = [&](int row) -> int { return p.sep[min(row, T) + opaque_zero]; };
        // gate `piece + c * lpr` of the slice that starts at `start`, or a later slice's (the consumer checks the count)
        auto fetch_gate = [&](int start, int c) -> uint32_t {
            const uint32_t off = min((uint32_t)(start + c * lpr + piece), (uint32_t)last_edge) * 8u;  // 32-bit offset from a uniform base
            return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(words) + off);
        };
        uint32_t st[4];
        walk_state_at(part, p.vprev, r0, ldq, piece * 16, st);
        uint32_t l0 = st[0], l1 = st[1], l2 = st[2], l3 = st[3];
        typename std::conditional<KIND == QP_DT_U16, unsigned long long, double>::type total = 0;
        bool bad = false;
        uint4 ring[AHEAD], mring[AHEAD];
        uint32_t gring[AHEAD][kWalkGateChunks];
        int sring[AHEAD], ering[AHEAD];
        int sp_next = slice_start(r0);
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            sring[k] = sp_next;
            sp_next = slice_start(r0 + k + 1);
            ering[k] = sp_next;
            ring[k] = fetch_row(r0 + k);
            mring[k] = fetch_mask(r0 + k);
            const int start = __builtin_amdgcn_readfirstlane(sring[k]);
#pragma unroll
            for (int c = 0; c < kWalkGateChunks; ++c) gring[k][c] = fetch_gate(start, c);
        }
        for (int g0 = r0; g0 < T; g0 += AHEAD) {
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) {
                const int row = g0 + k;
                if (row >= T) break;
                const uint4 cur = ring[k], vm = mring[k];
                uint32_t gates[kWalkGateChunks];
#pragma unroll
                for (int c = 0; c < kWalkGateChunks; ++c) gates[c] = gring[k][c];
                const int e0 = __builtin_amdgcn_readfirstlane(sring[k]);
                const int slice_gates = __builtin_amdgcn_readfirstlane(ering[k]) - e0;
                {
                    const int sp_here = sp_next;             // start of slice row + AHEAD, asked for a row ago
                    sp_next = slice_start(row + AHEAD + 1);  // used a row from now
                    sring[k] = sp_here;
                    ering[k] = sp_next;
                    ring[k] = fetch_row(row + AHEAD);
                    mring[k] = fetch_mask(row + AHEAD);
                    const int start = __builtin_amdgcn_readfirstlane(sp_here);
#pragma unroll
                    for (int c = 0; c < kWalkGateChunks; ++c) gring[k][c] = fetch_gate(start, c);
                }

                if (lane_on) *reinterpret_cast<uint4*>(strip + lane * 16) = cur;
                const int mine = slice_gates - piece;  // gates c * lpr < mine are this lane's
                uint32_t au[kWalkGateChunks], av[kWalkGateChunks];
#pragma unroll
                for (int c = 0; c < kWalkGateChunks; ++c) {
                    if (!(lane_on && c * lpr < mine)) gates[c] = 0u;  // (qubit 0, qubit 0): one core twice, distance zero
                    au[c] = av[c] = 0u;
                    if (c * lpr < slice_gates) {  // the same for the whole wave
                        au[c] = strip[goff + (gates[c] & 0xFFFFu)];
                        av[c] = strip[goff + (gates[c] >> 16)];
                    }
                }
                Acc acc = 0;
                const uint32_t f0 = (vm.x & l0) | (~vm.x & cur.x), f1 = (vm.y & l1) | (~vm.y & cur.y);
                const uint32_t f2 = (vm.z & l2) | (~vm.z & cur.z), f3 = (vm.w & l3) | (~vm.w & cur.w);
                if (f0 != cur.x) acc += V4 ? four(cur.x, f0) : four(f0, cur.x);
                if (f1 != cur.y) acc += V4 ? four(cur.y, f1) : four(f1, cur.y);
                if (f2 != cur.z) acc += V4 ? four(cur.z, f2) : four(f2, cur.z);
                if (f3 != cur.w) acc += V4 ? four(cur.w, f3) : four(f3, cur.w);
                // a visited byte with its top bit set may be 0xFF: only then the byte tests
                if (((cur.x & vm.x) | (cur.y & vm.y) | (cur.z & vm.z) | (cur.w & vm.w)) & 0x80808080u) {
                    if ((bytes_ff(cur.x & vm.x) | bytes_ff(cur.y & vm.y) | bytes_ff(cur.z & vm.z) | bytes_ff(cur.w & vm.w)) != 0)
                        bad = bad || walk_lost(vm.x, l0, cur.x) || walk_lost(vm.y, l1, cur.y) || walk_lost(vm.z, l2, cur.z) ||
                              walk_lost(vm.w, l3, cur.w);
                }
                l0 = (vm.x & cur.x) | (~vm.x & l0);
                l1 = (vm.y & cur.y) | (~vm.y & l1);
                l2 = (vm.z & cur.z) | (~vm.z & l2);
                l3 = (vm.w & cur.w) | (~vm.w & l3);
                uint32_t top = 0;
#pragma unroll
                for (int c = 0; c < kWalkGateChunks; ++c) top = max(top, max(au[c], av[c]));
                if (top == 0xFFu && p.gate_errors) {
#pragma unroll
                    for (int c = 0; c < kWalkGateChunks; ++c) bad = bad || (gates[c] != 0u && max(au[c], av[c]) == 0xFFu);
                }
#pragma unroll
                for (int c = 0; c < kWalkGateChunks; ++c) {
                    if (c * lpr < slice_gates) acc += at(au[c], av[c]);  // the same for the whole wave
                }
                for (int c = kWalkGateChunks; c * lpr < mine && lane_on; ++c) {
                    const uint32_t w = words[2 * (size_t)(e0 + c * lpr + piece)];
                    const uint32_t cu = strip[goff + (w & 0xFFFFu)], cv = strip[goff + (w >> 16)];
                    bad = bad || (p.gate_errors && max(cu, cv) == 0xFFu);
                    acc += at(cu, cv);
                }
                total += acc;  // 16-bit table: a lane adds 16 + gates / lpr terms below 2^16 per row into 32 bits here
            }
        }
        // per partition: the sums and error flags of its lpr lanes, through LDS (a wave's LDS operations complete in order)
        volatile double* sums = reinterpret_cast<volatile double*>(strip + 1024);
        volatile uint32_t* flags = reinterpret_cast<volatile uint32_t*>(strip + 1024 + 512);
        sums[lane] = (double)total;
        flags[lane] = bad ? 1u : 0u;
        if (lane_on && piece == 0 && b0 + g < p.B) {
            double s = 0.0;
            uint32_t any = 0;
            for (int j = 0; j < lpr; ++j) {
                s += sums[lane + j];
                any |= flags[lane + j];
            }
            p.partial[(size_t)chunk * p.B + b] = any ? std::numeric_limits<double>::infinity() : s;
        }
    }
}

__global__ void k_fill(double* out, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

// ------------------------------------------------------------------------------------------------
// Cost matrices
// ------------------------------------------------------------------------------------------------

struct CmArgs {
    const QpEdge* edges;
    const int16_t* lookahead;
    const double* D;
    const int32_t* t;
    const uint8_t* a;
    const int32_t* ops;
    const int32_t* n_ops;
    const uint8_t* avail;
    const int32_t* n_avail;
    double* out;
    int32_t C, ldq, max_ops, kind;
    const int32_t* tab;  // optional: instance b uses the table D + tab[b] * C * C
    // optional: [B][max_ops][8] the cores of every op resolved by the caller (partners 0..5 or -1, then the cores of
    // the op's own two qubits): then ops, a and the circuit tables are not read
    const int16_t* resolved;
};

__device__ inline double pymax(double a, double b) { return b > a ? b : a; }  // Python's max(a, b)

// One workgroup = one instance x kCmOps consecutive ops; thread j owns available-core column j.
// What depends on the op only (the cores of its two qubits and of its up to six look-ahead partners)
// hangs on a chain of dependent loads (op -> gate -> qubit -> core).  kCmOps * 8 threads walk one chain
// each, side by side, and leave the eight cores in LDS.  The look-ahead attraction of an op is the same
// sum for every core that holds none of its partners (all terms take the 0.3 branch), so it is formed
// once per op, in the reference's order; a 256-bit mask per op marks the few cores that do hold a
// partner, and only those entries walk the six slots.  After that an entry costs two coalesced distance
// loads, three broadcast LDS reads, the reference's clamp arithmetic and one coalesced 8-byte store.
constexpr int kCmOps = 32;

__device__ inline double cm_attraction(const int32_t* pc, int32_t c) {
    double att = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (pc[k] >= 0) {
            const double w = k < 2 ? 0.5 : (k < 4 ? 0.25 : 0.125);
            att += (pc[k] == c) ? w * 0.9 : w * 0.3;
        }
    }
    return att;
}

template <int KINDC>
__global__ __launch_bounds__(256) void k_cost_matrix(CmArgs p) {
    __shared__ __align__(16) int32_t opinfo[kCmOps][8];  // [0..5] partner cores (-1 = none), [6] a1, [7] a2
    __shared__ double att_far[kCmOps];                   // attraction seen from a core that holds no partner
    __shared__ uint32_t near_mask[kCmOps][8];            // bit c: core c holds a look-ahead partner
    __shared__ double att_near[kCmOps][6];               // attraction seen from the core of partner k
    const int64_t b = blockIdx.y;
    const int i0 = blockIdx.x * kCmOps;
    const int C = p.C;
    // With resolved ops every input of the workgroup is requested here, before anything depends on a loaded value:
    // the caller's arrays may live in host memory, where each dependent load is a round trip over the bus.
    uint4 rec = make_uint4(0, 0, 0, 0);
    if (p.resolved && threadIdx.x < kCmOps)
        rec = *reinterpret_cast<const uint4*>(p.resolved + ((size_t)b * p.max_ops + min(i0 + (int)threadIdx.x, p.max_ops - 1)) * 8);
    const int32_t c_first = p.avail[(size_t)b * C + min((int)threadIdx.x, C - 1)];  // the core of this thread's column
    const size_t tab_off = p.tab ? (size_t)p.tab[b] * C * C : (size_t)0;
    const int n_ops = p.n_ops[b], n_av = p.n_avail[b];
    if (i0 >= n_ops) return;
    const int i1 = min(n_ops, i0 + kCmOps);
    const uint8_t* a = p.a + (size_t)b * p.ldq;
    if (p.resolved) {
        if (threadIdx.x < kCmOps) {
            const uint32_t w[4] = {rec.x, rec.y, rec.z, rec.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int v = (int)(int16_t)(w[k >> 1] >> (16 * (k & 1)));
                opinfo[threadIdx.x][k] = (i0 + (int)threadIdx.x < i1) ? min(v, C - 1) : -1;
            }
        }
    } else if (threadIdx.x < kCmOps * 8) {
        const int i = i0 + (threadIdx.x >> 3), k = threadIdx.x & 7;
        int32_t v = -1;
        if (i < i1) {
            const int eid = p.ops[(size_t)b * p.max_ops + i];
            int q;
            if (k < 6) q = KINDC == QP_CM_GTRHQA ? -1 : p.lookahead[(size_t)eid * 6 + k];
            else q = k == 6 ? p.edges[eid].u : p.edges[eid].v;
            if (q >= 0) v = min((int)a[q], C - 1);  // a core id >= C is the caller's error; never index with it
        }
        opinfo[threadIdx.x >> 3][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kCmOps * 8 && (threadIdx.x & 7) < 6) {  // once per op and partner, not once per column
        const int32_t* pc = opinfo[threadIdx.x >> 3];
        const int32_t c = pc[threadIdx.x & 7];
        att_near[threadIdx.x >> 3][threadIdx.x & 7] = c >= 0 ? cm_attraction(pc, c) : 0.0;
    }
    if (threadIdx.x < kCmOps) {
        const int32_t* pc = opinfo[threadIdx.x];
        att_far[threadIdx.x] = cm_attraction(pc, -2);  // -2 equals no core
        uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (pc[k] >= 0) {
#pragma unroll
                for (int w = 0; w < 8; ++w) m[w] |= (pc[k] >> 5) == w ? 1u << (pc[k] & 31) : 0u;
            }
#pragma unroll
        for (int w = 0; w < 8; ++w) near_mask[threadIdx.x][w] = m[w];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_av; j += blockDim.x) {
        const int32_t c = min(j == (int)threadIdx.x ? c_first : (int32_t)p.avail[(size_t)b * C + j], C - 1);
        const double* Dc = p.D + tab_off + c;  // column c of the instance's table
        double* outp = p.out + ((size_t)b * p.max_ops + i0) * C + j;
        // four ops at a time: their eight distance loads are issued before any of the arithmetic
        const int nloc = i1 - i0;
        for (int ib = 0; ib < nloc; ib += 4) {
            int32_t a1s[4], a2s[4];
            double d1s[4], d2s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(ib + u, nloc - 1);
                a1s[u] = opinfo[i][6];
                a2s[u] = opinfo[i][7];
                d1s[u] = Dc[a1s[u] * C];
                d2s[u] = Dc[a2s[u] * C];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
            const int i = ib + u;
            if (i >= nloc) break;
            const int32_t a1 = a1s[u], a2 = a2s[u];
            const double d1 = d1s[u], d2 = d2s[u];
            double r;
            if constexpr (KINDC == QP_CM_GTRHQA) {
                // a core whose label is 0 is "falsy" in the reference: leaving core 0 costs nothing
                double cost = 0.0;
                if (a1 != 0 && a1 != c) cost += d1;
                if (a2 != 0 && a2 != c) cost += d2;
                r = cost;
            } else {
                double att = att_far[i];
                if ((near_mask[i][c >> 5] >> (c & 31)) & 1u) {
                    // the sum depends on the core only: take it from any partner slot that sits on this core
                    int slot = 0;
#pragma unroll
                    for (int k = 1; k < 6; ++k) slot = opinfo[i][k] == c ? k : slot;
                    att = att_near[i][slot];
                }
                if constexpr (KINDC == QP_CM_HQA) {
                    const double m = (double)((a1 != c) + (a2 != c));
                    r = pymax(0.1, pymax(0.1, m - att));
                } else {
                    double mv = 0.0;
                    if (a1 != c) mv += d1;
                    if (a2 != c) mv += d2;
                    const double T_ = pymax(0.1, mv - att);
                    double here = 0.0;
                    if (a1 != 0 && a1 != c) here += d1;
                    if (a2 != 0 && a2 != c) here += d2;
                    const double attraction = here - T_;
                    r = pymax(0.1, here - attraction);
                }
            }
            __builtin_nontemporal_store(r, &outp[(size_t)i * C]);  // written once, read by the host or a later kernel
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------

// Byte-replicated masks and weights of a grid metric, as the closed-form kernels take them.
// A closed form stands for the table: an integer grid metric, or a real number times one.
bool has_grid_form(const QpDTable* dt) {
    return dt->grid.valid && (dt->kind == QP_DT_U16 || (dt->kind == QP_DT_F64 && dt->grid_scale > 0.0));
}
double grid_scale_of(const QpDTable* dt) { return dt->kind == QP_DT_F64 ? dt->grid_scale : 1.0; }

int fill_grid_args(V3Args& p, const QpDTable* dt) {
    const QpGridMetric& g = dt->grid;
    if (g.xbits < 0 || g.xbits > 8 || (g.wrap_x && g.wrap_x != (1 << g.xbits)) || g.wrap_x > 128 ||
        g.wrap_y > 128 || (g.wrap_y & (g.wrap_y - 1)) || (g.wrap_y && (g.wrap_y << g.xbits) != dt->C))
        return qp::fail(QP_EINVAL, "qp_slice_cost_v3: inconsistent grid metric");
    const uint32_t rep = 0x01010101u;
    p.gxb = (uint32_t)g.xbits;
    p.gxm = ((1u << g.xbits) - 1u) * rep;
    p.gym = (g.wrap_y ? (uint32_t)(g.wrap_y - 1) : (0xFFu >> g.xbits)) * rep;
    p.ghalfx = (uint32_t)(g.wrap_x / 2) * rep;
    p.ghalfy = (uint32_t)(g.wrap_y / 2) * rep;
    p.gwx = g.wx;
    p.gwy = g.wy;
    p.gyi = p.gym << p.gxb;
    return QP_OK;
}

int rows_per_item_for(int32_t T, int32_t G, int64_t B, int cus) {
    if (G == 1) {
        // one row per step: long items amortise the pipeline fill at an item's start (the first rows are
        // waited for with nothing else in flight); keep about 4 items per resident wave for balance
        const int64_t want = (int64_t)T * B / ((int64_t)cus * kV3Waves * 4);
        return (int)std::min<int64_t>(256, std::max<int64_t>(32, want));
    }
    // about 32 rows per item, a multiple of the group size, so the one extra row read per item is ~3%
    int r = std::max(G, (32 / G) * G);
    if (r < kV3Ahead * G && T > kV3Ahead * G) r = kV3Ahead * G;
    return r;
}

int check_common(const char* fn, const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B,
                 int32_t Tp, int32_t ldq, const double* out) {
    if (!circ || !dt || !A || !out) return qp::fail(QP_EINVAL, "%s: null argument", fn);
    if (B <= 0) return qp::fail(QP_EINVAL, "%s: B must be positive", fn);
    if (circ->Q <= 0 || circ->T < 0 || circ->Q > 65535) return qp::fail(QP_EINVAL, "%s: bad circuit sizes", fn);
    if (ldq < circ->Q || ldq % 16 != 0) return qp::fail(QP_EINVAL, "%s: ldq must be >= Q and a multiple of 16", fn);
    if (dt->C <= 0 || dt->C > 256 || !dt->table) return qp::fail(QP_EINVAL, "%s: bad distance table", fn);
    if (dt->kind != QP_DT_F64 && dt->kind != QP_DT_U16) return qp::fail(QP_EINVAL, "%s: unknown table kind", fn);
    if (reinterpret_cast<uintptr_t>(dt->table) % 16 != 0) return qp::fail(QP_EINVAL, "%s: the distance table must be 16-byte aligned", fn);
    if (reinterpret_cast<uintptr_t>(A) % 16 != 0) return qp::fail(QP_EINVAL, "%s: the partitions must be 16-byte aligned", fn);
    if (Tp <= 0) return qp::fail(QP_EINVAL, "%s: Tp must be positive", fn);
    return QP_OK;
}

}  // namespace

extern "C" size_t qp_slice_cost_workspace(int32_t T, int64_t B) {
    if (T <= 0 || B <= 0) return 16;
    // one double per (chunk, partition); chunks are never shorter than 16 rows unless T is
    const int64_t max_chunks = (T + 15) / 16 + 1;
    // up to four table-band passes; room for the symmetrised, folded triangle of a 256-core table
    return (size_t)(max_chunks * B) * sizeof(double) * 4 + 256 + 128 * 256 * sizeof(double) + 256;
}

namespace {

// The dense cost path behind qp_slice_cost_v3 and qp_slice_cost_v3_static: every launch of it, with the
// partial sums (and the symmetrised triangle of a large fp64 table) in `workspace`.
// Which kernel runs depends on the circuit, the table and the batch only, never on how much workspace there is: a
// workspace too small for the chosen kernel is QP_ENOSPC (so both entry points give the same bits or an error).
int slice_cost_v3_impl(const char* fn, const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B, int32_t Tp,
                       int32_t ldq, double* out, void* workspace, size_t workspace_bytes, void* stream_) {
    if (int rc = check_common(fn, circ, dt, A, B, Tp, ldq, out)) return rc;
    auto need_workspace = [&](size_t need) -> int {
        if (workspace && workspace_bytes >= need) return QP_OK;
        return qp::fail(QP_ENOSPC, "%s: this batch needs %zu bytes of scratch, %zu given", fn, need, workspace ? workspace_bytes : (size_t)0);
    };
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int32_t T = circ->T;
    if (Tp < T) return qp::fail(QP_EINVAL, "%s: partition has %d rows, circuit has %d slices", fn, Tp, T);
    if (T == 0) {
        k_fill<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(out, B, 0.0);
        HIP_TRY(hipGetLastError());
        return QP_OK;
    }
    if (!circ->slice_edge_ptr || (circ->nE > 0 && !circ->edges))
        return qp::fail(QP_EINVAL, "%s: circuit edge tables missing", fn);

    V3Args p{};
    p.A = A;
    p.sep = circ->slice_edge_ptr;
    p.edges = circ->edges;
    p.edges_sched = circ->edges_sched;
    p.sptr = circ->sched_ptr;
    p.gsched = circ->group_sched;
    p.gptr = circ->group_ptr;
    p.table = dt->table;
    p.B = B;
    p.T = T;
    p.Tp = Tp;
    p.ldq = ldq;
    p.Q = circ->Q;
    p.C = dt->C;
    const size_t esz = dt->kind == QP_DT_F64 ? 8 : 2;
    p.table_bytes = (uint32_t)((size_t)dt->C * dt->C * esz);
    const size_t tab_padded = (p.table_bytes + 15u) & ~(size_t)15;
    const bool narrow = ldq <= 1024;
    if (narrow) {
        p.lpr = ldq / 16;
        p.G = kWave / p.lpr;
    } else {
        p.lpr = 0;
        p.G = 1;
    }
    const int cus = qp::device_cus();
    p.rows_per_item = rows_per_item_for(T, p.G, B, cus);
    // rows longer than 1 KB: workgroup size, LDS contents and rows per item of k_cost_v3_wide
    int wide_threads = kV3Threads;
    size_t wide_tab = 0, wide_strips = 0;
    bool wide_two = false, wide_pairs = false;
    double* wide_fold = nullptr;
    if (!narrow) {
        if (ldq / 16 > kWidePieces * kV3Threads) return qp::fail(QP_EINVAL, "%s: rows longer than 65536 bytes", fn);
        const bool gridpath = has_grid_form(dt);
        // the table (if it fits beside one set of strips) and one or two strips of a row per group
        wide_strips = (size_t)kV3Threads * 16 * (ldq / 16 <= kV3Threads ? 1 : kWidePieces);
        if (!gridpath && tab_padded + wide_strips <= (size_t)kLdsBytes) wide_tab = tab_padded;
        // half-size workgroups when two of them (table and two strips each) fit a CU
        if (ldq / 16 <= kV3Threads / 2 && 2 * (wide_tab + wide_strips) <= (size_t)kLdsBytes) {
            wide_threads = kV3Threads / 2;
            wide_strips /= 2;
        }
        wide_two = wide_tab + 2 * wide_strips <= (size_t)kLdsBytes;
        // an item's rows are split over threads / P thread groups (P as the kernel forms it): runs of at least
        // 32 rows per group, so a run's start-up (one extra row, the first loads waited for) is small
        int P = 128;
        while (P < ldq / 16 && P < wide_threads) P *= 2;
        p.rows_per_item = std::min<int>(T, std::max(p.rows_per_item, 32 * (wide_threads / P)));
        // fp64 table that does not fit beside the strips but is symmetric to rounding, rows of up to 16 KB: pair
        // counts instead of gathers from device memory.  One 1024-thread workgroup per CU (the counters take
        // 128 KB); long items, about 8 per workgroup, since every item ends with a pass over all counters
        if (dt->kind == QP_DT_F64 && dt->symmetric && dt->C >= 2 && wide_tab == 0 && !gridpath && ldq / 16 <= kV3Threads) {
            int P1 = 128;
            while (P1 < ldq / 16) P1 *= 2;
            const int64_t want = ((int64_t)T * B + (int64_t)cus * 8 - 1) / ((int64_t)cus * 8);
            const int rows = (int)std::min<int64_t>(T, std::max<int64_t>(32 * (kV3Threads / P1), want));
            const int64_t items = (int64_t)((T + rows - 1) / rows) * B;
            const size_t part_bytes = ((size_t)items * sizeof(double) + 255) & ~(size_t)255;
            if (int rc = need_workspace(part_bytes + (size_t)kPairSlots * sizeof(double))) return rc;
            {
                wide_pairs = true;
                wide_fold = reinterpret_cast<double*>(static_cast<uint8_t*>(workspace) + part_bytes);
                wide_threads = kV3Threads;
                wide_strips = (size_t)kV3Threads * 16;
                wide_two = (size_t)kPairSlots * sizeof(uint32_t) + 2 * wide_strips <= (size_t)kLdsBytes;
                p.rows_per_item = rows;
            }
        }
    }
    const int32_t nchunks = (T + p.rows_per_item - 1) / p.rows_per_item;
    p.nitems = (int64_t)nchunks * B;
    p.partial = static_cast<double*>(workspace);  // every path checks its own need before it launches

    // Rows of 513 to 1024 bytes and a 16-bit table of at most 16 cores without a closed form: one slice per step, so
    // the per-slice gate list is the grouped list, and the kernel for short rows brings its table form whose
    // look-ups never collide (2.85 -> 3.7 TB/s; an fp64 table of that size gains nothing over the four-partition kernel).
    const bool one_row_groups = narrow && ldq > 512 && dt->C <= 16 && dt->kind == QP_DT_U16 && !dt->grid.valid &&
                                circ->edges_sched && circ->sched_ptr && circ->nE > 0;
    // The same rows, any table of at most 64 cores without a closed form, 64 partitions or more: the four-partition kernel
    // of the short rows with one row per step (its small table forms leave room for sixteen waves per CU where the
    // table at a pitch of 256 entries leaves room for eight).  16-bit tables of up to 96 cores too: at a pitch of 256
    // they take at most 48 KB, two workgroups still fit (81 cores: 1.58 -> 1.33 ms on the benchmark's rows); beyond
    // that one workgroup fits either way and k_cost_v3_row4 is the faster of the two (256 cores: 1.20 against 1.25 ms).
    const bool one_row_quads = narrow && ldq > 512 && (dt->C <= 64 || (dt->kind == QP_DT_U16 ? dt->C <= 96 : dt->C <= 128)) && !has_grid_form(dt) &&
                               B >= 64 && circ->edges_sched && circ->sched_ptr && circ->nE > 0;
    if (one_row_groups || one_row_quads ||
        (narrow && ldq <= 512 && circ->group_sched && circ->group_ptr && circ->group_ldq == ldq && circ->nE > 0)) {
        // rows of at most 512 bytes with the grouped gate list: whole groups of G slices per item, long items
        V3Args q = p;
        if (one_row_groups || one_row_quads) {
            q.gsched = circ->edges_sched;
            q.gptr = circ->sched_ptr;
        }
        const int G = q.G;
        const int64_t steps_total = (T + G - 1) / G;
        const int64_t want = steps_total * B / ((int64_t)cus * kV3Waves * 4);
        const int64_t spi = std::min<int64_t>(steps_total, std::min<int64_t>(256, std::max<int64_t>(8, want)));
        q.rows_per_item = (int)(spi * G);
        const int32_t gchunks = (T + q.rows_per_item - 1) / q.rows_per_item;
        // a circuit of few steps: several consecutive partitions per item (up to 256 steps), walked as one stream,
        // as long as that leaves a few items for every wave the GPU holds
        q.parts_per_item = 1;
        if (gchunks == 1)
            q.parts_per_item = (int)std::max<int64_t>(1, std::min<int64_t>({256 / steps_total, 64, B / ((int64_t)cus * 24 * 2)}));
        q.nbgroups = (B + q.parts_per_item - 1) / q.parts_per_item;
        q.nitems = (int64_t)gchunks * q.nbgroups;
        const size_t gstrips = (size_t)kV3Waves * (G + 1) * ldq;
        if (dt->kind == QP_DT_F64 && dt->symmetric && dt->C >= 2 && tab_padded + gstrips > (size_t)kLdsBytes &&
            !has_grid_form(dt) && (int64_t)T * ldq >= 24576) {
            // fp64 table too large for LDS, symmetric to rounding, partitions of 24 KB or more: pair counts, a
            // workgroup per item (about 8 items per workgroup: every item ends with a pass over all counters, which
            // costs as much as counting 30 KB of rows; measured on 256 cores: 31 KB partitions 0.69 TB/s against 0.55
            // through gathers from device memory, 16 KB partitions 0.44 against 0.54)
            V3Args c = p;
            const int64_t want = (steps_total * B + (int64_t)cus * 8 - 1) / ((int64_t)cus * 8);
            const int64_t cspi = std::min<int64_t>(steps_total, std::max<int64_t>(2 * kV3Waves, want));
            c.rows_per_item = (int)(cspi * G);
            const int32_t cchunks = (T + c.rows_per_item - 1) / c.rows_per_item;
            c.nitems = (int64_t)cchunks * B;
            const size_t part_bytes = ((size_t)c.nitems * sizeof(double) + 255) & ~(size_t)255;
            const size_t clds = (size_t)kPairSlots * sizeof(uint32_t) + gstrips + kV3Waves * sizeof(double);
            if (clds <= (size_t)kLdsBytes) {
                if (int rc = need_workspace(part_bytes + (size_t)kPairSlots * sizeof(double))) return rc;
                double* fold = reinterpret_cast<double*>(static_cast<uint8_t*>(workspace) + part_bytes);
                k_sym_fold<<<kPairSlots / 256, 256, 0, stream>>>(static_cast<const double*>(dt->table), dt->C, fold);
                HIP_TRY(hipGetLastError());
                c.table = fold;
                void (*kern)(V3Args) = k_cost_v3_pairs_groups<kV3Ahead, false>;
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
                kern<<<(unsigned)std::min<int64_t>(cus, c.nitems), kV3Threads, clds, stream>>>(c);
                HIP_TRY(hipGetLastError());
                k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(c.partial, B, cchunks, out);
                HIP_TRY(hipGetLastError());
                return QP_OK;
            }
        }
        {
            if (int rc = need_workspace((size_t)gchunks * B * sizeof(double))) return rc;
            const bool gridpath = has_grid_form(dt);
            const size_t strips = gstrips;
            const size_t tab256 = (size_t)dt->C * 256 * esz;
            // what the table takes in LDS and how it is addressed
            int mode = kTabGlobal;
            size_t tab_lds = 0;
            if (gridpath) {
                if (int rc = fill_grid_args(q, dt)) return rc;
                mode = kTabGrid;
            } else if (dt->C <= 16) {
                // up to 16 cores: a 16-bit table 32 times, one copy per LDS bank: no look-up waits for another
                // (fully random partitions: 1.5 -> 3.5 TB/s); an fp64 table 16 times (two lanes share a copy, but
                // three 8-wave workgroups then fit a CU: 3.0 TB/s against 2.8 with 32 copies and one workgroup)
                mode = kTabLdsRep4;
                q.rep_shift = dt->kind == QP_DT_F64 ? 4 : 5;
                tab_lds = ((size_t)256 << q.rep_shift) * (dt->kind == QP_DT_F64 ? 8 : 4);
            } else if (dt->C <= 64) {
                // up to 64 cores: the table at a pitch of 64 entries is 32 KB (fp64, or 16-bit twice), so three
                // 8-wave workgroups fit a CU; more copies (up to 128 KB, one workgroup) measured no faster
                mode = kTabLdsRep6;
                q.rep_shift = dt->kind == QP_DT_F64 ? 0 : 1;
                tab_lds = ((size_t)4096 << q.rep_shift) * (dt->kind == QP_DT_F64 ? 8 : 4);
            } else if (tab256 + strips <= (size_t)kLdsBytes) {
                mode = kTabLds256;
                tab_lds = tab256;
            } else if (tab_padded + strips <= (size_t)kLdsBytes) {
                mode = kTabLdsC;
                tab_lds = tab_padded;
            }
            // LDS tables indexed by packed core pairs, 64 partitions or more: four partitions per wave step (a last
            // quad that is not full repeats a partition, which is at most 5 % more work from 64 partitions on)
            const size_t strips4 = (size_t)(kGroups4Threads / kWave) * (G + (one_row_quads ? 0 : 1)) * ldq * 4;
            int mode4 = mode;  // the table form of the four-partition kernel
            size_t tab4 = tab_lds;
            if (!gridpath && dt->kind == QP_DT_F64 && dt->C > 64 && dt->C <= 128) {
                // fp64 tables of 65 to 128 cores: at a pitch of 128 entries (a form the one-partition kernel does not have)
                mode4 = kTabLdsP128;
                tab4 = (size_t)dt->C * 128 * sizeof(double);
            }
            if ((one_row_quads || !one_row_groups) && B >= 64 &&
                (mode4 == kTabLdsRep4 || mode4 == kTabLdsRep6 || mode4 == kTabLds256 || mode4 == kTabLdsP128) &&
                tab4 + strips4 <= (size_t)kLdsBytes) {
                if (mode4 == kTabLdsP128) q.rep_shift = 0;
                // The stream of (quad, step) is cut into runs: one per wave of the launch (all resident, every wave the
                // same number of steps to within one) when a partition has more than 16 steps -- a quad is then summed in
                // parts -- else runs of a few whole quads.
                const int64_t nquads = (B + 3) / 4;
                const bool whole = steps_total <= 16;
                q.parts_per_item = whole ? (int)steps_total : 1;
                q.nitems = whole ? nquads : nquads * steps_total;
                void (*kern4)(V3Args) = nullptr;
#define QP_GROUPS4(MODE, ONE) (dt->kind == QP_DT_F64 ? k_cost_v3_groups4<QP_DT_F64, MODE, kGroups4Ahead, ONE> : k_cost_v3_groups4<QP_DT_U16, MODE, kGroups4Ahead, ONE>)
                if (mode4 == kTabLdsP128) kern4 = one_row_quads ? k_cost_v3_groups4<QP_DT_F64, kTabLdsP128, kGroups4Ahead, true> : k_cost_v3_groups4<QP_DT_F64, kTabLdsP128, kGroups4Ahead, false>;
                else if (one_row_quads) kern4 = mode4 == kTabLdsRep4 ? QP_GROUPS4(kTabLdsRep4, true) : mode4 == kTabLdsRep6 ? QP_GROUPS4(kTabLdsRep6, true) : QP_GROUPS4(kTabLds256, true);
                else kern4 = mode4 == kTabLdsRep4 ? QP_GROUPS4(kTabLdsRep4, false) : mode4 == kTabLdsRep6 ? QP_GROUPS4(kTabLdsRep6, false) : QP_GROUPS4(kTabLds256, false);
#undef QP_GROUPS4
                const size_t lds4 = tab4 + strips4;
                const int waves4 = kGroups4Threads / kWave;
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
                int per_cu4 = 1;
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, reinterpret_cast<const void*>(kern4), kGroups4Threads, lds4));
                const int64_t grid4 = std::min<int64_t>((int64_t)cus * std::max(per_cu4, 1), (q.nitems + waves4 - 1) / waves4);
                int64_t nparts = 1;  // a quad's steps lie in at most this many runs
                if (whole) {
                    // runs of a few quads, taken in turn (measured: faster than one run per wave; 3 quads for 120 000
                    // partitions of 5 steps)
                    const int64_t ppi = std::max<int64_t>(1, std::min<int64_t>({256 / steps_total, 16, nquads / ((int64_t)cus * 16 * 2)}));
                    q.nbgroups = (nquads + ppi - 1) / ppi;
                } else {
                    // R runs over nquads * steps_total steps: fewer than R / nquads + 1 run boundaries inside a quad.  As many
                    // parts as a run per wave needs, within what qp_slice_cost_workspace promises for every batch.
                    const int64_t waves_all = grid4 * waves4;
                    nparts = std::min<int64_t>(4 * ((T + 15) / 16 + 1), std::max<int64_t>(3, (waves_all + nquads - 1) / nquads + 2));
                    // (never more runs than steps: a run without a step would leave its part of a quad unwritten)
                    q.nbgroups = std::max<int64_t>(1, std::min<int64_t>({waves_all, nquads * (nparts - 2), q.nitems}));
                }
                if (int rc = need_workspace((size_t)nparts * B * sizeof(double))) return rc;
                kern4<<<(unsigned)grid4, kGroups4Threads, lds4, stream>>>(q);
                HIP_TRY(hipGetLastError());
                k_sum_parts<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(q.partial, B, q.nbgroups, q.nitems, (int32_t)steps_total, q.parts_per_item, out);
                HIP_TRY(hipGetLastError());
                return QP_OK;
            }
            // workgroups of 8 waves when three or more of them fit a CU (24 waves or more in flight where one
            // workgroup of 16 waves would be alone: the registers of this kernel allow 5 to 7 waves per SIMD)
            const bool half = 3 * (tab_lds + strips / 2) <= (size_t)kLdsBytes;
            const int threads = half ? kV3Threads / 2 : kV3Threads;
            const size_t lds = tab_lds + (half ? strips / 2 : strips);
            void (*kern)(V3Args) = nullptr;
#define QP_GROUPS(KIND, MODE, WRAP) (half ? k_cost_v3_groups<KIND, MODE, WRAP, kV3Threads / 2> : k_cost_v3_groups<KIND, MODE, WRAP, kV3Threads>)
#define QP_GROUPS_KIND(MODE) (dt->kind == QP_DT_F64 ? QP_GROUPS(QP_DT_F64, MODE, false) : QP_GROUPS(QP_DT_U16, MODE, false))
            switch (mode) {
                case kTabGrid:
                    kern = (dt->grid.wrap_x || dt->grid.wrap_y) ? QP_GROUPS(QP_DT_U16, kTabGrid, true) : QP_GROUPS(QP_DT_U16, kTabGrid, false);
                    break;
                case kTabLdsRep4: kern = QP_GROUPS_KIND(kTabLdsRep4); break;
                case kTabLdsRep6: kern = QP_GROUPS_KIND(kTabLdsRep6); break;
                case kTabLds256: kern = QP_GROUPS_KIND(kTabLds256); break;
                case kTabLdsC: kern = QP_GROUPS_KIND(kTabLdsC); break;
                default: kern = QP_GROUPS_KIND(kTabGlobal); break;
            }
#undef QP_GROUPS_KIND
#undef QP_GROUPS
            const int waves = threads / kWave;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int per_cu = 1;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), threads, lds));
            const unsigned ggrid = (unsigned)std::min<int64_t>((int64_t)cus * std::max(per_cu, 1), (q.nitems + waves - 1) / waves);
            kern<<<ggrid, threads, lds, stream>>>(q);
            HIP_TRY(hipGetLastError());
            k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(q.partial, B, gchunks, out, gridpath ? grid_scale_of(dt) : 1.0);
            HIP_TRY(hipGetLastError());
            return QP_OK;
        }
    }
    if (narrow) {
        const bool rowpath = ldq > 512 && circ->edges_sched != nullptr && circ->sched_ptr != nullptr && circ->nE > 0;
        const bool gridpath = has_grid_form(dt);
        const size_t strips = rowpath ? (size_t)kV3Waves * (gridpath ? kGridStrip : ldq) : (size_t)kV3Waves * (p.G + 1) * ldq;
        const size_t tab256 = (size_t)dt->C * 256 * esz;
        int mode = kTabGlobal;
        size_t lds = strips;
        if (tab256 + strips <= (size_t)kLdsBytes) {
            mode = kTabLds256;
            lds += tab256;
        } else if (tab_padded + strips <= (size_t)kLdsBytes) {
            mode = kTabLdsC;
            lds += tab_padded;
        }
        const unsigned grid = (unsigned)std::min<int64_t>(cus, (p.nitems + kV3Waves - 1) / kV3Waves);
        void (*kern)(V3Args) = nullptr;
        if (gridpath) {
            // closed-form distances: no table in LDS, so more than one workgroup fits a CU
            if (int rc = need_workspace((size_t)p.nitems * sizeof(double))) return rc;
            if (int rc = fill_grid_args(p, dt)) return rc;
            const QpGridMetric& g = dt->grid;
            const bool wrap = g.wrap_x || g.wrap_y;
            if (!rowpath) kern = k_cost_v3<QP_DT_U16, kTabGrid>;
            else kern = wrap ? k_cost_v3_grid_row<true, kV3Ahead> : k_cost_v3_grid_row<false, kV3Ahead>;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)strips));
            int per_cu = 1;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kV3Threads, strips));
            const unsigned ggrid = (unsigned)std::min<int64_t>((int64_t)cus * std::max(per_cu, 1), (p.nitems + kV3Waves - 1) / kV3Waves);
            kern<<<ggrid, kV3Threads, strips, stream>>>(p);
            HIP_TRY(hipGetLastError());
            k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(p.partial, B, nchunks, out, grid_scale_of(dt));
            HIP_TRY(hipGetLastError());
            return QP_OK;
        }
        if (rowpath && mode == kTabGlobal && dt->kind == QP_DT_F64 && dt->symmetric && dt->C >= 2) {
            // fp64 table too big for LDS but symmetric up to rounding: count core pairs in LDS, one pass
            const size_t fold_bytes = (size_t)kPairSlots * sizeof(double);
            const size_t pairs_lds = (size_t)kPairSlots * sizeof(uint32_t) + (size_t)kV3Waves * ldq + kV3Waves * sizeof(double);
            // whole workgroups take an item; rows in runs of at least 4 per wave, about 8 items per workgroup
            V3Args q = p;
            const int64_t want = ((int64_t)T * B + (int64_t)cus * 8 - 1) / ((int64_t)cus * 8);
            q.rows_per_item = (int)std::min<int64_t>(T, std::max<int64_t>(4 * kV3Waves, (want + kV3Waves - 1) / kV3Waves * kV3Waves));
            const int32_t pchunks = (T + q.rows_per_item - 1) / q.rows_per_item;
            q.nitems = (int64_t)pchunks * B;
            const size_t part_bytes = ((size_t)q.nitems * sizeof(double) + 255) & ~(size_t)255;
            if (pairs_lds <= (size_t)kLdsBytes) {
                if (int rc = need_workspace(part_bytes + fold_bytes)) return rc;
                double* fold = reinterpret_cast<double*>(static_cast<uint8_t*>(workspace) + part_bytes);
                k_sym_fold<<<kPairSlots / 256, 256, 0, stream>>>(static_cast<const double*>(dt->table), dt->C, fold);
                HIP_TRY(hipGetLastError());
                q.table = fold;
                kern = k_cost_v3_pairs<kV3Ahead, false>;
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pairs_lds));
                const unsigned pgrid = (unsigned)std::min<int64_t>(cus, q.nitems);
                kern<<<pgrid, kV3Threads, pairs_lds, stream>>>(q);
                HIP_TRY(hipGetLastError());
                k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(q.partial, B, pchunks, out);
                HIP_TRY(hipGetLastError());
                return QP_OK;
            }
        }
        if (rowpath && mode == kTabGlobal && dt->kind == QP_DT_F64) {
            // fp64 table too big for LDS: one pass per band of source cores, each with its band in LDS
            int band_rows = 1;  // largest power of two whose band fits beside the strips
            while ((size_t)band_rows * 2 * 256 * esz + 16 + strips <= (size_t)kLdsBytes) band_rows *= 2;
            const int passes = (dt->C + band_rows - 1) / band_rows;
            if (int rc = need_workspace((size_t)passes * p.nitems * sizeof(double))) return rc;
            kern = k_cost_v3_row<QP_DT_F64, kTabLdsBand>;
            const size_t band_lds = strips + (((size_t)band_rows * 256 + 1) * esz + 15) / 16 * 16;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)band_lds));
            for (int k = 0; k < passes; ++k) {
                V3Args q = p;
                q.band_lo = k * band_rows;
                q.band_rows = std::min(band_rows, dt->C - q.band_lo);
                q.partial = p.partial + (size_t)k * p.nitems;
                kern<<<grid, kV3Threads, band_lds, stream>>>(q);
            }
            HIP_TRY(hipGetLastError());
            k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(p.partial, B, nchunks * passes, out);
            HIP_TRY(hipGetLastError());
            return QP_OK;
        }
        if (rowpath && tab256 + (size_t)kRow4Waves * kRow4Strip <= (size_t)kLdsBytes && B >= 64) {
            // table at pitch 256 beside eight interleaved strips: four partitions per wave step (a last group that
            // is not full repeats a partition, which is at most 5 % more work from 64 partitions on)
            V3Args q = p;
            const int64_t ngroups = (B + 3) / 4;
            const int64_t want = (int64_t)T * ngroups / ((int64_t)cus * kRow4Waves * 4);
            q.rows_per_item = (int)std::min<int64_t>(T, std::min<int64_t>(256, std::max<int64_t>(32, want)));
            const int32_t qchunks = (T + q.rows_per_item - 1) / q.rows_per_item;
            q.nitems = (int64_t)qchunks * ngroups;
            {
                if (int rc = need_workspace((size_t)qchunks * B * sizeof(double))) return rc;
                kern = dt->kind == QP_DT_F64 ? k_cost_v3_row4<QP_DT_F64, kV3Ahead> : k_cost_v3_row4<QP_DT_U16, kV3Ahead>;
                const size_t lds4 = tab256 + (size_t)kRow4Waves * kRow4Strip;
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
                const unsigned grid4 = (unsigned)std::min<int64_t>(cus, (q.nitems + kRow4Waves - 1) / kRow4Waves);
                kern<<<grid4, kRow4Threads, lds4, stream>>>(q);
                HIP_TRY(hipGetLastError());
                k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(q.partial, B, qchunks, out);
                HIP_TRY(hipGetLastError());
                return QP_OK;
            }
        }
        if (rowpath) {
            if (dt->kind == QP_DT_F64)
                kern = mode == kTabLds256 ? k_cost_v3_row<QP_DT_F64, kTabLds256>
                     : mode == kTabLdsC   ? k_cost_v3_row<QP_DT_F64, kTabLdsC> : k_cost_v3_row<QP_DT_F64, kTabGlobal>;
            else
                kern = mode == kTabLds256 ? k_cost_v3_row<QP_DT_U16, kTabLds256>
                     : mode == kTabLdsC   ? k_cost_v3_row<QP_DT_U16, kTabLdsC> : k_cost_v3_row<QP_DT_U16, kTabGlobal>;
        } else if (dt->kind == QP_DT_F64) {
            kern = mode == kTabLds256 ? k_cost_v3<QP_DT_F64, kTabLds256>
                 : mode == kTabLdsC   ? k_cost_v3<QP_DT_F64, kTabLdsC> : k_cost_v3<QP_DT_F64, kTabGlobal>;
        } else {
            kern = mode == kTabLds256 ? k_cost_v3<QP_DT_U16, kTabLds256>
                 : mode == kTabLdsC   ? k_cost_v3<QP_DT_U16, kTabLdsC> : k_cost_v3<QP_DT_U16, kTabGlobal>;
        }
        if (int rc = need_workspace((size_t)p.nitems * sizeof(double))) return rc;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, kV3Threads, lds, stream>>>(p);
    } else {
        const bool gridpath = has_grid_form(dt);
        if (gridpath && ldq <= 2048 && circ->edges_sched && circ->sched_ptr && circ->nE > 0) {
            // closed form on rows of up to 2 KB: a wave per row, no workgroup-wide step (3.3 -> 4.7 TB/s, the rate of the
            // 1 KB rows; four pieces per lane for rows of up to 4 KB need 154 registers and were slower than the
            // workgroup-wide kernel below, 2.7 against 3.25 TB/s)
            V3Args q = p;
            if (int rc = fill_grid_args(q, dt)) return rc;
            q.rows_per_item = rows_per_item_for(T, 1, B, cus);
            const int32_t chunks = (T + q.rows_per_item - 1) / q.rows_per_item;
            q.nitems = (int64_t)chunks * B;
            if (int rc = need_workspace((size_t)q.nitems * sizeof(double))) return rc;
            const bool wrap = dt->grid.wrap_x || dt->grid.wrap_y;
            const int np = 2;
            void (*kern)(V3Args) = wrap ? k_cost_v3_grid_rown<true, 2> : k_cost_v3_grid_rown<false, 2>;
            const int waves = kGridNThreads / kWave;
            const size_t lds = (size_t)waves * np * 1024;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int per_cu = 1;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kGridNThreads, lds));
            const unsigned grid = (unsigned)std::min<int64_t>((int64_t)cus * std::max(per_cu, 1), (q.nitems + waves - 1) / waves);
            kern<<<grid, kGridNThreads, lds, stream>>>(q);
            HIP_TRY(hipGetLastError());
            k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(q.partial, B, chunks, out, grid_scale_of(dt));
            HIP_TRY(hipGetLastError());
            return QP_OK;
        }
        const bool one = ldq / 16 <= kV3Threads;  // one piece per thread
        const bool half = wide_threads < kV3Threads, two = wide_two, tab_lds = wide_tab > 0;
        const int threads = wide_threads;
        void (*kern)(V3Args) = nullptr;
#define QP_WIDE(KIND, MODE) \
    (half ? k_cost_v3_wide<KIND, MODE, 1, 2, kV3Threads / 2> \
     : one ? (two ? k_cost_v3_wide<KIND, MODE, 1, 2, kV3Threads> : k_cost_v3_wide<KIND, MODE, 1, 1, kV3Threads>) \
           : (two ? k_cost_v3_wide<KIND, MODE, kWidePieces, 2, kV3Threads> : k_cost_v3_wide<KIND, MODE, kWidePieces, 1, kV3Threads>))
        if (wide_pairs) {
            k_sym_fold<<<kPairSlots / 256, 256, 0, stream>>>(static_cast<const double*>(dt->table), dt->C, wide_fold);
            HIP_TRY(hipGetLastError());
            p.table = wide_fold;
            kern = two ? k_cost_v3_wide<QP_DT_F64, kTabPairs, 1, 2, kV3Threads> : k_cost_v3_wide<QP_DT_F64, kTabPairs, 1, 1, kV3Threads>;
        } else if (gridpath) {
            if (int rc = fill_grid_args(p, dt)) return rc;
            kern = QP_WIDE(QP_DT_U16, kTabGrid);
        } else if (dt->kind == QP_DT_F64) {
            kern = tab_lds ? QP_WIDE(QP_DT_F64, kTabLdsC) : QP_WIDE(QP_DT_F64, kTabGlobal);
        } else {
            kern = tab_lds ? QP_WIDE(QP_DT_U16, kTabLdsC) : QP_WIDE(QP_DT_U16, kTabGlobal);
        }
#undef QP_WIDE
        if (int rc = need_workspace((size_t)p.nitems * sizeof(double))) return rc;
        const size_t lds = (wide_pairs ? (size_t)kPairSlots * sizeof(uint32_t) : wide_tab) + (two ? 2 : 1) * wide_strips;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 1;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), threads, lds));
        const unsigned grid = (unsigned)std::min<int64_t>((int64_t)cus * std::max(per_cu, 1), p.nitems);
        kern<<<grid, threads, lds, stream>>>(p);
    }
    HIP_TRY(hipGetLastError());
    k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(p.partial, B, nchunks, out,
                                                                   (!narrow && !wide_pairs && has_grid_form(dt)) ? grid_scale_of(dt) : 1.0);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

// Scratch of qp_slice_cost_v3_static: lives in the code object, one copy per GPU the library runs on, for
// as long as the process does.  The name below is a host-side handle, not a device address: host code
// only ever uses the per-device address hipGetSymbolAddress returns for it.
constexpr size_t kStaticScratchBytes = (size_t)QP_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
std::atomic<void*> g_scratch_ptr[kMaxDevices];  // zero-initialised; set once per device

// Device address of this GPU's copy of the scratch array (looked up on the first call on a device).
int static_scratch(void** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_static: device %d out of range", dev);
    void* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return qp::fail(QP_EHIP, "qp_slice_cost_v3_static: no address for the scratch array on device %d", dev);
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);  // every thread gets the same address for a device
    }
    *out = ptr;
    return QP_OK;
}

}  // namespace

// Per-core activity counts through the pair counters (rows of 513 to 1024 bytes with the per-slice gate list, shorter
// rows with the grouped one and partitions of 32 KB or more):
// QP_OK when the launch was made, 1 when this form does not apply and the caller should use its own kernel.
int qp::core_activity_pairs(const QpCircuit* circ, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, int32_t C,
                            uint64_t* out, void* stream_) {
    const int32_t T = circ->T;
    if (ldq < 16 || ldq > 1024 || ldq % 16 || circ->nE <= 0 || T <= 0 || reinterpret_cast<uintptr_t>(A) % 16) return 1;
    const bool rows_1k = ldq > 512;
    if (rows_1k ? (!circ->edges_sched || !circ->sched_ptr)
                : (!circ->group_sched || !circ->group_ptr || circ->group_ldq != ldq || (int64_t)T * ldq < 32768))
        return 1;  // (short partitions: a pass over all counters per item costs more than it saves)
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    V3Args q{};
    q.A = A;
    q.edges_sched = circ->edges_sched;
    q.sptr = circ->sched_ptr;
    q.gsched = circ->group_sched;
    q.gptr = circ->group_ptr;
    q.lpr = ldq / 16;
    q.G = rows_1k ? 1 : kWave / q.lpr;
    q.B = B;
    q.T = T;
    q.Tp = Tp;
    q.ldq = ldq;
    q.Q = circ->Q;
    q.C = C;
    q.activity = reinterpret_cast<unsigned long long*>(out);
    const int cus = qp::device_cus();
    size_t strips;
    void (*kern)(V3Args);
    if (rows_1k) {
        const int64_t want = ((int64_t)T * B + (int64_t)cus * 8 - 1) / ((int64_t)cus * 8);
        q.rows_per_item = (int)std::min<int64_t>(T, std::max<int64_t>(4 * kV3Waves, (want + kV3Waves - 1) / kV3Waves * kV3Waves));
        strips = (size_t)kV3Waves * ldq;
        kern = k_cost_v3_pairs<kV3Ahead, true>;
    } else {  // whole groups of G slices per item, as the cost launch forms them
        const int64_t steps_total = (T + q.G - 1) / q.G;
        const int64_t want = (steps_total * B + (int64_t)cus * 8 - 1) / ((int64_t)cus * 8);
        q.rows_per_item = (int)(std::min<int64_t>(steps_total, std::max<int64_t>(2 * kV3Waves, want)) * q.G);
        strips = (size_t)kV3Waves * (q.G + 1) * ldq;
        kern = k_cost_v3_pairs_groups<kV3Ahead, true>;
    }
    const int32_t chunks = (T + q.rows_per_item - 1) / q.rows_per_item;
    q.nitems = (int64_t)chunks * B;
    const size_t lds = (size_t)kPairSlots * sizeof(uint32_t) + strips + kV3Waves * sizeof(double) + 256 * sizeof(uint32_t);
    if (lds > (size_t)kLdsBytes) return 1;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<(unsigned)std::min<int64_t>(cus, q.nitems), kV3Threads, lds, stream>>>(q);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

int qp::comm_matrix_pairs(const QpCircuit* circ, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, int32_t C,
                          int32_t* out, int32_t* status, void* stream_) {
    const int32_t T = circ->T;
    // rows of 513 to 1024 bytes, a circuit that visits over 128 qubits per slice, enough rows for the GPU
    if (ldq <= 512 || ldq > 1024 || ldq % 16 || T <= 0 || T > 32767 || reinterpret_cast<uintptr_t>(A) % 16 || !circ->visit_mask ||
        circ->visit_mask_ldq != ldq || !circ->visit_prev || !circ->edges_sched || !circ->sched_ptr ||
        (int64_t)circ->nV < (int64_t)128 * T || (int64_t)B * T < 16384)
        return 1;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    V3Args q{};
    q.A = A;
    q.edges_sched = circ->edges_sched;
    q.sptr = circ->sched_ptr;
    q.vmask = circ->visit_mask;
    q.vprev = circ->visit_prev;
    q.B = B;
    q.T = T;
    q.Tp = Tp;
    q.ldq = ldq;
    q.Q = circ->Q;
    q.C = C;
    q.gate_errors = 1;
    q.comm = out;
    q.status = status;
    const int cus = qp::device_cus();
    // an item ends with a pass over all counters and up to two global updates per core pair that occurred: whole
    // partitions (every matrix then has one writer, which stores instead of adding) unless that leaves most of the GPU
    // idle; then about one item per workgroup
    const int64_t want = ((int64_t)T * B + cus - 1) / cus;
    q.rows_per_item = B * 4 >= cus ? T : (int)std::min<int64_t>(T, std::max<int64_t>(4 * kV3Waves, (want + kV3Waves - 1) / kV3Waves * kV3Waves));
    q.nitems = (int64_t)((T + q.rows_per_item - 1) / q.rows_per_item) * B;
    const size_t lds = (size_t)kPairSlots * sizeof(uint32_t) + (size_t)kV3Waves * ldq + kV3Waves * sizeof(double);
    void (*kern)(V3Args) = k_comm_matrix_pairs<kV3Ahead>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<(unsigned)std::min<int64_t>(cus, q.nitems), kV3Threads, lds, stream>>>(q);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

extern "C" int qp_slice_cost_v3(const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B, int32_t Tp,
                                int32_t ldq, double* out, void* workspace, size_t workspace_bytes, void* stream) {
    return slice_cost_v3_impl("qp_slice_cost_v3", circ, dt, A, B, Tp, ldq, out, workspace, workspace_bytes, stream);
}

extern "C" size_t qp_static_scratch_bytes(void) { return kStaticScratchBytes; }

extern "C" int qp_slice_cost_v3_static(const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B,
                                       int32_t Tp, int32_t ldq, double* out, void* stream) {
    void* ws = nullptr;
    if (int rc = static_scratch(&ws)) return rc;
    const int rc = slice_cost_v3_impl("qp_slice_cost_v3_static", circ, dt, A, B, Tp, ldq, out, ws, kStaticScratchBytes, stream);
    if (rc == QP_ENOSPC)
        return qp::fail(QP_ENOSPC,
                        "qp_slice_cost_v3_static: %lld partitions x %d slices need more than the %zu bytes of static "
                        "scratch (one double per partition and chunk of slices); split the batch or use qp_slice_cost_v3",
                        (long long)B, circ->T, kStaticScratchBytes);
    return rc;
}

extern "C" int qp_slice_cost_v2(const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B, int32_t Tp,
                                int32_t ldq, int32_t flags, double* out, void* workspace, size_t workspace_bytes,
                                void* stream_) {
    if (int rc = check_common("qp_slice_cost_v2", circ, dt, A, B, Tp, ldq, out)) return rc;
    if (flags != QP_V1 && flags != QP_V2 && flags != QP_V4) return qp::fail(QP_EINVAL, "qp_slice_cost_v2: bad flags");
    if (dt->C > 255) return qp::fail(QP_EINVAL, "qp_slice_cost_v2: at most 255 cores (0xFF marks an absent qubit)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int32_t T = circ->T;
    if (Tp < T) {
        if (flags == QP_V1) {
            T = Tp;  // Placer.py:87-88
        } else {      // partition[i] past the end: IndexError in the reference, inf in its results
            k_fill<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(out, B, std::numeric_limits<double>::infinity());
            HIP_TRY(hipGetLastError());
            return QP_OK;
        }
    }
    if (T == 0 || circ->nV == 0) {
        k_fill<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(out, B, 0.0);
        HIP_TRY(hipGetLastError());
        return QP_OK;
    }
    // A circuit that visits most of every row (over 128 qubits a slice, where reading whole rows beats gathering the
    // visited bytes), enough rows to fill the GPU: the row walk.  16-bit tables are looked up (integer sums); fp64 tables
    // that are symmetric to rounding are applied to core-pair counts; other fp64 tables stay with the visits.
    // (visit_prev holds slice numbers in 16 bits)
    const int16_t* const vprev_ok = circ->T <= 32767 ? circ->visit_prev : nullptr;
    const bool walkable = ldq > 512 && ldq <= 1024 && circ->visit_mask && circ->visit_mask_ldq == ldq &&
                          circ->edges_sched && circ->sched_ptr && (int64_t)circ->nV >= (int64_t)128 * T &&
                          (vprev_ok ? (int64_t)B * T >= 65536 : B >= 1024);
    const int wcus = qp::device_cus();
    // rows of at most 512 bytes: the same walk with G partitions per wave step
    // (and longer rows on up to 16 cores, one partition per step: this kernel's table form has a copy of every entry
    // per LDS bank; 2.07 -> 1.79 ms per 4 GB on a 16-bit table, 2.56 -> 1.89 ms on an fp64 one, against the two below)
    const int32_t wlpr = ldq / 16, wG = wlpr > 0 && (wlpr <= 32 || (wlpr <= 64 && dt->C <= 16)) ? kWave / wlpr : 0;
    if (wG >= 1 && circ->visit_mask && circ->visit_mask_ldq == ldq && vprev_ok && circ->edges && circ->slice_edge_ptr &&
        circ->nE > 0 && (int64_t)circ->nV * 8 >= (int64_t)T * ldq && B >= wG && ((int64_t)B * T * ldq >= ((int64_t)1 << 26) || (ldq > 512 && (int64_t)B * T >= 65536)) &&
        (dt->kind == QP_DT_U16 || dt->C <= 127)) {
        V2WalkArgs w{};
        w.A = A;
        w.vmask = circ->visit_mask;
        w.vprev = vprev_ok;
        w.edges = circ->edges;
        w.sep = circ->slice_edge_ptr;
        w.nE = circ->nE;
        w.table = dt->table;
        w.B = B;
        w.T = T;
        w.Tp = Tp;
        w.ldq = ldq;
        w.C = dt->C;
        w.lpr = wlpr;
        w.G = wG;
        w.ngroups = (B + wG - 1) / wG;
        w.gate_errors = flags != QP_V1;
        int32_t wchunks = (int32_t)std::max<int64_t>(1, std::min<int64_t>(T / 64, ((int64_t)wcus * kV3Waves * 2 + w.ngroups - 1) / w.ngroups));
        w.rows_per_item = (T + wchunks - 1) / wchunks;
        wchunks = (T + w.rows_per_item - 1) / w.rows_per_item;
        w.nitems = (int64_t)wchunks * w.ngroups;
        if (wchunks == 1) {
            w.partial = out;
        } else {
            if (!workspace || workspace_bytes < (size_t)wchunks * B * sizeof(double))
                return qp::fail(QP_ENOSPC, "qp_slice_cost_v2: workspace needs %zu bytes", (size_t)wchunks * B * sizeof(double));
            w.partial = static_cast<double*>(workspace);
        }
        const bool v4 = flags == QP_V4;
        // ids squeezed to k bits, 2^k > C (0xFF becomes 2^k - 1); as many copies of every entry as fit 128 KB, up to
        // one per bank (32 dwords, 16 doubles); a 16-bit table of 128 cores or more stays 16 bits wide, one copy
        int kb = 1;
        while ((1 << kb) <= dt->C) ++kb;
        const bool f64 = dt->kind == QP_DT_F64, native16 = !f64 && kb == 8;
        const int esz_log = f64 ? 3 : native16 ? 1 : 2;
        w.id_bits = kb;
        w.rep_shift = std::max(0, std::min(f64 ? 4 : native16 ? 0 : 5, 17 - 2 * kb - esz_log));
        void (*walk)(V2WalkArgs) =
            f64 ? (v4 ? k_cost_v2_walk_parts<QP_DT_F64, true, 3> : k_cost_v2_walk_parts<QP_DT_F64, false, 3>)
            : native16 ? (v4 ? k_cost_v2_walk_parts<QP_DT_U16, true, 3, true> : k_cost_v2_walk_parts<QP_DT_U16, false, 3, true>)
                       : (v4 ? k_cost_v2_walk_parts<QP_DT_U16, true, 3> : k_cost_v2_walk_parts<QP_DT_U16, false, 3>);
        const size_t lds = ((((size_t)1 << (2 * kb)) << w.rep_shift) << esz_log) + (size_t)kV3Waves * kWalkPartsLds;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(walk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const unsigned grid = (unsigned)std::min<int64_t>(wcus, (w.nitems + kV3Waves - 1) / kV3Waves);
        walk<<<grid, kV3Threads, lds, stream>>>(w);
        HIP_TRY(hipGetLastError());
        if (wchunks > 1) {
            k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(w.partial, B, wchunks, out);
            HIP_TRY(hipGetLastError());
        }
        return QP_OK;
    }
    if (walkable && dt->kind == QP_DT_U16) {
        V2WalkArgs w{};
        w.A = A;
        w.vmask = circ->visit_mask;
        w.vprev = vprev_ok;
        w.sched = circ->edges_sched;
        w.sptr = circ->sched_ptr;
        w.table = dt->table;
        w.B = B;
        w.T = T;
        w.Tp = Tp;
        w.ldq = ldq;
        w.C = dt->C;
        w.gate_errors = flags != QP_V1;
        // whole partitions when there are two or more per wave the GPU holds, else chunks of 64 slices or more
        int32_t wchunks = 1;
        if (vprev_ok)
            wchunks = (int32_t)std::max<int64_t>(1, std::min<int64_t>(T / 64, ((int64_t)wcus * kV3Waves * 2 + B - 1) / B));
        w.rows_per_item = (T + wchunks - 1) / wchunks;
        wchunks = (T + w.rows_per_item - 1) / w.rows_per_item;
        w.nitems = (int64_t)wchunks * B;
        if (wchunks == 1) {
            w.partial = out;
        } else {
            if (!workspace || workspace_bytes < (size_t)w.nitems * sizeof(double))
                return qp::fail(QP_ENOSPC, "qp_slice_cost_v2: workspace needs %zu bytes", (size_t)w.nitems * sizeof(double));
            w.partial = static_cast<double*>(workspace);
        }
        void (*walk)(V2WalkArgs) = flags == QP_V4 ? k_cost_v2_walk<true, kV3Ahead> : k_cost_v2_walk<false, kV3Ahead>;
        const size_t lds = (size_t)256 * 256 * sizeof(uint16_t) + (size_t)kV3Waves * ldq;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(walk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const unsigned grid = (unsigned)std::min<int64_t>(wcus, (w.nitems + kV3Waves - 1) / kV3Waves);
        walk<<<grid, kV3Threads, lds, stream>>>(w);
        HIP_TRY(hipGetLastError());
        if (wchunks > 1) {
            k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(w.partial, B, wchunks, out);
            HIP_TRY(hipGetLastError());
        }
        return QP_OK;
    }
    if (walkable && dt->kind == QP_DT_F64 && dt->symmetric && dt->C >= 2 && vprev_ok) {
        // as the dense cost does it: whole workgroups take an item, about 8 items per workgroup
        V3Args q{};
        q.A = A;
        q.edges_sched = circ->edges_sched;
        q.sptr = circ->sched_ptr;
        q.vmask = circ->visit_mask;
        q.vprev = vprev_ok;
        q.B = B;
        q.T = T;
        q.Tp = Tp;
        q.ldq = ldq;
        q.Q = circ->Q;
        q.C = dt->C;
        q.gate_errors = flags != QP_V1;
        const int64_t want = ((int64_t)T * B + (int64_t)wcus * 8 - 1) / ((int64_t)wcus * 8);
        q.rows_per_item = (int)std::min<int64_t>(T, std::max<int64_t>(4 * kV3Waves, (want + kV3Waves - 1) / kV3Waves * kV3Waves));
        const int32_t pchunks = (T + q.rows_per_item - 1) / q.rows_per_item;
        q.nitems = (int64_t)pchunks * B;
        const size_t part_bytes = ((size_t)q.nitems * sizeof(double) + 255) & ~(size_t)255;
        const size_t fold_bytes = (size_t)kPairSlots * sizeof(double);
        const size_t pairs_lds = (size_t)kPairSlots * sizeof(uint32_t) + (size_t)kV3Waves * ldq + kV3Waves * sizeof(double);
        if (workspace && workspace_bytes >= part_bytes + fold_bytes) {
            q.partial = static_cast<double*>(workspace);
            double* fold = reinterpret_cast<double*>(static_cast<uint8_t*>(workspace) + part_bytes);
            k_sym_fold<<<kPairSlots / 256, 256, 0, stream>>>(static_cast<const double*>(dt->table), dt->C, fold);
            HIP_TRY(hipGetLastError());
            q.table = fold;
            void (*pk)(V3Args) = k_cost_v2_pairs<kV3Ahead>;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pairs_lds));
            pk<<<(unsigned)std::min<int64_t>(wcus, q.nitems), kV3Threads, pairs_lds, stream>>>(q);
            HIP_TRY(hipGetLastError());
            k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(q.partial, B, pchunks, out);
            HIP_TRY(hipGetLastError());
            return QP_OK;
        }
    }
    if (!circ->slice_visit_ptr || !circ->visits) return qp::fail(QP_EINVAL, "qp_slice_cost_v2: circuit visit tables missing");
    V2Args p{};
    p.A = A;
    p.svp = circ->slice_visit_ptr;
    p.visits = circ->visits;
    p.table = dt->table;
    p.B = B;
    p.T = T;
    p.Tp = Tp;
    p.ldq = ldq;
    p.C = dt->C;
    p.mode = flags;
    p.rows_per_item = 32;
    const int32_t nchunks = (T + p.rows_per_item - 1) / p.rows_per_item;
    const size_t need = (size_t)nchunks * B * sizeof(double);
    p.nitems = (int64_t)nchunks * ((B + kV2Group - 1) / kV2Group);
    if (!workspace || workspace_bytes < need) return qp::fail(QP_ENOSPC, "qp_slice_cost_v2: workspace needs %zu bytes", need);
    p.partial = static_cast<double*>(workspace);
    const size_t esz = dt->kind == QP_DT_F64 ? 8 : 2;
    p.table_bytes = (uint32_t)((size_t)dt->C * dt->C * esz);
    const size_t tab_padded = (p.table_bytes + 15u) & ~(size_t)15;
    // A table of up to 64 KB shares the CU's LDS between several 256-thread workgroups; a larger one that
    // still fits (every 16-bit table does: 255 cores are 127 KB) gets one 1024-thread workgroup per CU.
    // Random look-ups in a table left in global memory touch one cache line per lane and bound the kernel.
    const bool tab_small = tab_padded <= 64 * 1024;
    const bool tab_big = !tab_small && tab_padded <= 144 * 1024;
    const size_t lds = (tab_small || tab_big) ? tab_padded : 0;
    const int threads = tab_big ? 1024 : 256;
    const int cus = qp::device_cus();
    void (*kern)(V2Args) = nullptr;
    if (dt->kind == QP_DT_F64)
        kern = tab_small ? k_cost_v2<QP_DT_F64, true, 256> : tab_big ? k_cost_v2<QP_DT_F64, true, 1024> : k_cost_v2<QP_DT_F64, false, 256>;
    else
        kern = tab_small ? k_cost_v2<QP_DT_U16, true, 256> : tab_big ? k_cost_v2<QP_DT_U16, true, 1024> : k_cost_v2<QP_DT_U16, false, 256>;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 1;  // the loop hides gather latency with resident waves: as many workgroups as fit
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), threads, lds));
    const int wpb = threads / kWave;
    const unsigned grid = (unsigned)std::min<int64_t>((int64_t)cus * std::max(per_cu, 1), (p.nitems + wpb - 1) / wpb);
    kern<<<grid, threads, lds, stream>>>(p);
    HIP_TRY(hipGetLastError());
    k_sum_chunks<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(p.partial, B, nchunks, out);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

namespace {

int cost_matrix_impl(const char* fn, int32_t kind, const QpCircuit* circ, const double* D, int32_t C, const int32_t* tab,
                     int64_t B, const uint8_t* a, int32_t ldq, const int32_t* ops, const int32_t* n_ops, int32_t max_ops,
                     const uint8_t* avail, const int32_t* n_avail, double* cost_out, void* stream_,
                     const int16_t* resolved = nullptr) {
    if (kind != QP_CM_HQA && kind != QP_CM_TRHQA && kind != QP_CM_GTRHQA) return qp::fail(QP_EINVAL, "%s: bad kind", fn);
    if (!D || !n_ops || !avail || !n_avail || !cost_out) return qp::fail(QP_EINVAL, "%s: null argument", fn);
    if (!resolved && (!circ || !a || !ops)) return qp::fail(QP_EINVAL, "%s: null argument", fn);
    if (B <= 0 || B > 65535) return qp::fail(QP_EINVAL, "%s: B must be 1..65535 per call", fn);
    if (max_ops <= 0) return qp::fail(QP_EINVAL, "%s: max_ops must be positive", fn);
    if (C <= 0 || C > 256) return qp::fail(QP_EINVAL, "%s: bad core count", fn);
    if (!resolved) {
        if (!circ->edges || (kind != QP_CM_GTRHQA && !circ->lookahead)) return qp::fail(QP_EINVAL, "%s: circuit tables missing", fn);
        if (ldq < circ->Q) return qp::fail(QP_EINVAL, "%s: ldq < Q", fn);
    } else if (reinterpret_cast<uintptr_t>(resolved) % 16 != 0) {
        return qp::fail(QP_EINVAL, "%s: resolved ops must be 16-byte aligned", fn);
    }
    CmArgs p{};
    p.resolved = resolved;
    p.edges = circ ? circ->edges : nullptr;
    p.lookahead = circ ? circ->lookahead : nullptr;
    p.D = D;
    p.tab = tab;
    p.t = nullptr;
    p.a = a;
    p.ops = ops;
    p.n_ops = n_ops;
    p.avail = avail;
    p.n_avail = n_avail;
    p.out = cost_out;
    p.C = C;
    p.ldq = ldq;
    p.max_ops = max_ops;
    p.kind = kind;
    const dim3 grid((unsigned)((max_ops + kCmOps - 1) / kCmOps), (unsigned)B);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (kind == QP_CM_HQA) k_cost_matrix<QP_CM_HQA><<<grid, 256, 0, stream>>>(p);
    else if (kind == QP_CM_TRHQA) k_cost_matrix<QP_CM_TRHQA><<<grid, 256, 0, stream>>>(p);
    else k_cost_matrix<QP_CM_GTRHQA><<<grid, 256, 0, stream>>>(p);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

}  // namespace

int qp::cost_matrix_resolved(int32_t kind, const double* tables, int32_t C, const int32_t* tab_index, int64_t B,
                             const int16_t* resolved, const int32_t* n_ops, int32_t max_ops, const uint8_t* avail,
                             const int32_t* n_avail, double* cost_out, void* stream) {
    return cost_matrix_impl("cost_matrix_resolved", kind, nullptr, tables, C, tab_index, B, nullptr, 0, nullptr, n_ops, max_ops,
                            avail, n_avail, cost_out, stream, resolved);
}

extern "C" int qp_cost_matrix(int32_t kind, const QpCircuit* circ, const QpDTable* dt, int64_t B, const int32_t* t,
                              const uint8_t* a, int32_t ldq, const int32_t* ops, const int32_t* n_ops, int32_t max_ops,
                              const uint8_t* avail, const int32_t* n_avail, double* cost_out, void* stream) {
    (void)t;
    if (!dt) return qp::fail(QP_EINVAL, "qp_cost_matrix: null argument");
    if (dt->kind != QP_DT_F64 || !dt->table) return qp::fail(QP_EINVAL, "qp_cost_matrix: needs a QP_DT_F64 distance table");
    return cost_matrix_impl("qp_cost_matrix", kind, circ, static_cast<const double*>(dt->table), dt->C, nullptr, B, a, ldq, ops,
                            n_ops, max_ops, avail, n_avail, cost_out, stream);
}

extern "C" int qp_cost_matrix_multi(int32_t kind, const QpCircuit* circ, const double* tables, int32_t S, int32_t C,
                                    const int32_t* tab_index, int64_t B, const uint8_t* a, int32_t ldq, const int32_t* ops,
                                    const int32_t* n_ops, int32_t max_ops, const uint8_t* avail, const int32_t* n_avail,
                                    double* cost_out, void* stream) {
    if (!tables || !tab_index || S <= 0) return qp::fail(QP_EINVAL, "qp_cost_matrix_multi: null argument");
    return cost_matrix_impl("qp_cost_matrix_multi", kind, circ, tables, C, tab_index, B, a, ldq, ops, n_ops, max_ops, avail,
                            n_avail, cost_out, stream);
}

// ------------------------------------------------------------------------------------------------
// Cost V3 slice by slice: out[b][t - t_begin] = the terms timeslice t adds to the sum of HQA_placer.py:259-287
// (its crossing gates; from slice 2 on the qubits that changed core since slice t - 1).  One wave per
// (slice, partition); the partition rows may be a window of the circuit's slices that starts at slice row0, so
// a long circuit can be costed in contiguous pieces (one per GPU) that only share the row before each piece.
// ------------------------------------------------------------------------------------------------

namespace {

struct V3SliceArgs {
    const uint8_t* A;
    const int32_t* sep;
    const QpEdge* edges;
    const void* table;
    double* out;
    int64_t B;
    int32_t Tp, ldq, Q, C, row0, t_begin, t_end;
};

template <int KIND>
__global__ __launch_bounds__(256) void k_cost_v3_slices(V3SliceArgs p) {
    using E = typename Tab<KIND>::elem;
    const E* tab = static_cast<const E*>(p.table);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int nt = p.t_end - p.t_begin;
    const int64_t items = (int64_t)nt * p.B;
    const int C = p.C;
    for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < items; it += (int64_t)gridDim.x * 4) {
        const int t = p.t_begin + (int)(it / p.B);  // consecutive waves share the slice's gates
        const int64_t b = it % p.B;
        const uint8_t* cur = p.A + ((size_t)b * p.Tp + (t - p.row0)) * p.ldq;
        double acc = 0.0;
        for (int e = p.sep[t] + lane; e < p.sep[t + 1]; e += kWave) {
            const QpEdge ed = p.edges[e];
            const int au = min((int)cur[ed.u], C - 1), av = min((int)cur[ed.v], C - 1);
            if (au != av) acc += (double)tab[au * C + av];
        }
        if (t >= 2) {  // movement is charged from slice 2 on
            const uint8_t* prv = cur - p.ldq;
            for (int k = lane * 16; k < p.ldq; k += kWave * 16) {
                const uint4 c4 = *reinterpret_cast<const uint4*>(cur + k), p4 = *reinterpret_cast<const uint4*>(prv + k);
                const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (cw[w] == pw[w]) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int from = min((int)((pw[w] >> (8 * j)) & 0xFFu), C - 1), to = min((int)((cw[w] >> (8 * j)) & 0xFFu), C - 1);
                        if (k + 4 * w + j < p.Q && from != to) acc += (double)tab[from * C + to];
                    }
                }
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) p.out[(size_t)b * nt + (t - p.t_begin)] = acc;
    }
}

}  // namespace

extern "C" int qp_slice_cost_v3_slices(const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B, int32_t Tp,
                                       int32_t ldq, int32_t row0, int32_t t_begin, int32_t t_end, double* out, void* stream_) {
    if (int rc = check_common("qp_slice_cost_v3_slices", circ, dt, A, B, Tp, ldq, out)) return rc;
    const int32_t T = circ->T;
    if (t_begin < 0 || t_end > T || t_begin > t_end) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_slices: slices %d..%d of %d", t_begin, t_end, T);
    if (t_begin == t_end) return QP_OK;
    // the window must hold every row the slices read: rows t_begin .. t_end - 1, and row t_begin - 1 from slice 2 on
    const int32_t first = t_begin >= 2 ? t_begin - 1 : t_begin;
    if (row0 < 0 || row0 > first || (int64_t)row0 + Tp < t_end)
        return qp::fail(QP_EINVAL, "qp_slice_cost_v3_slices: rows %d..%d do not cover slices %d..%d", row0, row0 + Tp, first, t_end);
    if (!circ->slice_edge_ptr || (circ->nE > 0 && !circ->edges)) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_slices: circuit edge tables missing");
    V3SliceArgs p{};
    p.A = A;
    p.sep = circ->slice_edge_ptr;
    p.edges = circ->edges;
    p.table = dt->table;
    p.out = out;
    p.B = B;
    p.Tp = Tp;
    p.ldq = ldq;
    p.Q = circ->Q;
    p.C = dt->C;
    p.row0 = row0;
    p.t_begin = t_begin;
    p.t_end = t_end;
    const int64_t items = (int64_t)(t_end - t_begin) * B;
    const unsigned grid = (unsigned)std::min<int64_t>((int64_t)qp::device_cus() * 8, (items + 3) / 4);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (dt->kind == QP_DT_F64) k_cost_v3_slices<QP_DT_F64><<<grid, 256, 0, stream>>>(p);
    else k_cost_v3_slices<QP_DT_U16><<<grid, 256, 0, stream>>>(p);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

// ------------------------------------------------------------------------------------------------
// Cost V3 with one distance table per instance (many damaged copies of one network)
// ------------------------------------------------------------------------------------------------

namespace {

struct V3MultiArgs {
    const uint8_t* A;
    const int32_t* sep;
    const QpEdge* edges;
    const double* tables;
    const int32_t* tab;
    double* out;
    int64_t B;
    int32_t T, Tp, ldq, Q, C, S;
};

constexpr int kMultiThreads = 256;

// One workgroup per instance at a time: it stages the instance's table in LDS (TAB_LDS) or reads it where it
// lies, its waves take every fourth row (moves into the row from the row before, then the row's gates), and
// the instance's cost is the fixed-order sum of the lanes' and waves' partial sums.
template <bool TAB_LDS>
__global__ __launch_bounds__(kMultiThreads) void k_cost_v3_multi(V3MultiArgs p) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ double red[kMultiThreads / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t C = (uint32_t)p.C, last = C * C - 1u;
    for (int64_t b = blockIdx.x; b < p.B; b += gridDim.x) {
        const int32_t ti = min(max(p.tab[b], 0), p.S - 1);
        const double* tab = p.tables + (size_t)ti * C * C;
        __syncthreads();  // the previous instance's look-ups are done and thread 0 has read its partial sums
        if constexpr (TAB_LDS) {
            double* dst = reinterpret_cast<double*>(smem);
            for (uint32_t k = threadIdx.x; k <= last; k += kMultiThreads) dst[k] = tab[k];
            __syncthreads();
            tab = dst;
        }
        const uint8_t* base = p.A + (size_t)b * p.Tp * p.ldq;
        double acc = 0.0;
        for (int t = wave; t < p.T; t += kMultiThreads / kWave) {
            const uint8_t* cur = base + (size_t)t * p.ldq;
            if (t >= 2) {  // movement is charged from row 2 on
                const uint8_t* prv = cur - p.ldq;
                for (int q = lane; q < p.Q; q += kWave) {
                    const uint32_t from = prv[q], to = cur[q];
                    if (from != to) acc += tab[min(from * C + to, last)];
                }
            }
            for (int e = p.sep[t] + lane; e < p.sep[t + 1]; e += kWave) {
                const QpEdge ed = p.edges[e];
                const uint32_t u = cur[ed.u], v = cur[ed.v];
                if (u != v) acc += tab[min(u * C + v, last)];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int w = 0; w < kMultiThreads / kWave; ++w) s += red[w];
            p.out[b] = s;
        }
    }
}

}  // namespace

extern "C" int qp_slice_cost_v3_multi(const QpCircuit* circ, const double* tables, int32_t S, int32_t C,
                                      const int32_t* tab_index, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq,
                                      double* out, void* stream_) {
    if (!circ || !tables || !tab_index || !A || !out) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_multi: null argument");
    if (B <= 0 || S <= 0) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_multi: B and S must be positive");
    if (C <= 0 || C > 256) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_multi: 1..256 cores");
    if (circ->Q <= 0 || circ->T < 0 || ldq < circ->Q) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_multi: bad circuit sizes or ldq < Q");
    if (Tp < circ->T) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_multi: partition has %d rows, circuit has %d slices", Tp, circ->T);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (circ->T == 0) {
        k_fill<<<(unsigned)((B + 255) / 256), 256, 0, stream>>>(out, B, 0.0);
        HIP_TRY(hipGetLastError());
        return QP_OK;
    }
    if (!circ->slice_edge_ptr || (circ->nE > 0 && !circ->edges)) return qp::fail(QP_EINVAL, "qp_slice_cost_v3_multi: circuit edge tables missing");
    V3MultiArgs p{};
    p.A = A;
    p.sep = circ->slice_edge_ptr;
    p.edges = circ->edges;
    p.tables = tables;
    p.tab = tab_index;
    p.out = out;
    p.B = B;
    p.T = circ->T;
    p.Tp = Tp;
    p.ldq = ldq;
    p.Q = circ->Q;
    p.C = C;
    p.S = S;
    const size_t tab_bytes = (size_t)C * C * sizeof(double);
    const int cus = qp::device_cus();
    if (tab_bytes <= 128 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cost_v3_multi<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes));
        int per_cu = 1;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_cost_v3_multi<true>), kMultiThreads, tab_bytes));
        const unsigned grid = (unsigned)std::min<int64_t>((int64_t)cus * std::max(per_cu, 1), B);
        k_cost_v3_multi<true><<<grid, kMultiThreads, tab_bytes, stream>>>(p);
    } else {
        const unsigned grid = (unsigned)std::min<int64_t>((int64_t)cus * 8, B);
        k_cost_v3_multi<false><<<grid, kMultiThreads, 0, stream>>>(p);
    }
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

int qp::device_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

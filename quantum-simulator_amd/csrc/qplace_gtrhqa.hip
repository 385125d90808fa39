// GTRHQA placement on the device: one workgroup owns one instance and walks all of its timeslices,
// so a batch of placements needs no host round trip per slice.  Follows GTRHQA_placer.py:62-265:
// slice t starts from row t-1 (row 0 for t = 0) and its result becomes row t+1 (two interleaved
// chains); the ops of the slice are its counted gates whose qubits sit on different cores; a core is
// available when it holds at most cap-2 qubits at the start of the slice; the op x core cost is
// GTRHQA_placer.py:99-112 (moves away from core 0 are free); assignment is the bidding of :128-254.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "qplace_internal.h"

// -DQP_GTRHQA_TIMING (development): thread 0 accumulates clock64() per phase and instance 0 prints them
#ifdef QP_GTRHQA_TIMING
#define TIC(slot) do { const long long now_ = clock64(); tacc[slot] += now_ - tlast; tlast = now_; } while (0)
#else
#define TIC(slot) do { } while (0)
#endif

// waves per SIMD the narrow workgroup is compiled for (register budget 512 / QP_GT_WAVES): 6 measured best
#ifndef QP_GT_WAVES
#define QP_GT_WAVES 6
#endif
#ifndef QP_GT_WAVES_INT
#define QP_GT_WAVES_INT 7
#endif

namespace {

constexpr int kWave = 64;
constexpr double kBig = 1e12;  // what a non-finite entry counts as in an argmin (GTRHQA_placer.py:139)

struct GpArgs {
    const int32_t* sep;
    const QpEdge* edges;
    const double* D;            // prepared copy (row 0 and diagonal zero)
    const uint16_t* D16;        // the same as 16-bit integers; only meaningful when flags[1] is set
    const int32_t* all_finite;  // [0] 1 when every distance is finite, [1] 1 when every distance is an integer <= 32767
    const uint8_t* shortlist;   // [C][C][kShort]: the cheapest cores of every pair of source cores, see k_gtrhqa_shortlist
    const uint8_t* initial;
    uint8_t* out;
    int64_t B;
    int32_t T, Q, ldq, C, cap, max_ops;
    // byte offsets into dynamic LDS
    uint32_t off_rows, off_occ, off_best, off_win, off_bid, off_target, off_uv, off_a12, off_flags, off_cores, off_need, off_scan;
};

struct OpFlags {
    uint8_t left, need, fin, pad;
};

__device__ inline unsigned long long dbits(double x) { return (unsigned long long)__double_as_longlong(x); }

// Lexicographic minimum of (key, idx) over the wave with DPP moves (no LDS traffic); every lane gets it.
template <int CTRL, int ROW_MASK>
__device__ inline void lexmin_step(unsigned long long& key, uint32_t& idx) {
    const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xF, false);
    const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xF, false);
    const uint32_t oidx = (uint32_t)__builtin_amdgcn_update_dpp((int)idx, (int)idx, CTRL, ROW_MASK, 0xF, false);
    const unsigned long long okey = ((unsigned long long)ohi << 32) | olo;
    if (okey < key || (okey == key && oidx < idx)) {
        key = okey;
        idx = oidx;
    }
}
// Same over each 16-lane row: every lane of a row ends up with its row's minimum.
__device__ inline void lexmin_row(unsigned long long& key, uint32_t& idx) {
    lexmin_step<0xB1, 0xF>(key, idx);   // quad_perm [1,0,3,2]
    lexmin_step<0x4E, 0xF>(key, idx);   // quad_perm [2,3,0,1]
    lexmin_step<0x141, 0xF>(key, idx);  // row_half_mirror
    lexmin_step<0x140, 0xF>(key, idx);  // row_mirror
}
__device__ inline void lexmin_wave(unsigned long long& key, uint32_t& idx) {
    lexmin_step<0xB1, 0xF>(key, idx);   // quad_perm [1,0,3,2]
    lexmin_step<0x4E, 0xF>(key, idx);   // quad_perm [2,3,0,1]
    lexmin_step<0x141, 0xF>(key, idx);  // row_half_mirror
    lexmin_step<0x140, 0xF>(key, idx);  // row_mirror: every lane of a 16-lane row holds the row's minimum
    lexmin_step<0x142, 0xA>(key, idx);  // row_bcast:15 into rows 1 and 3
    lexmin_step<0x143, 0xC>(key, idx);  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's minimum
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)key, 63);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), 63);
    key = ((unsigned long long)hi << 32) | lo;
    idx = (uint32_t)__builtin_amdgcn_readlane((int)idx, 63);
}

// Minimum over each 16-lane row, every lane of the row gets it.  32-bit: the DPP operand folds into v_min_u32.
template <int CTRL>
__device__ inline uint32_t dpp_u32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, false);
}
__device__ inline uint32_t rowmin_u32(uint32_t x) {
    x = min(x, dpp_u32<0xB1>(x));   // quad_perm [1,0,3,2]
    x = min(x, dpp_u32<0x4E>(x));   // quad_perm [2,3,0,1]
    x = min(x, dpp_u32<0x141>(x));  // row_half_mirror
    x = min(x, dpp_u32<0x140>(x));  // row_mirror
    return x;
}
template <int CTRL>
__device__ inline unsigned long long min_step_u64(unsigned long long k) {
    const unsigned long long o = ((unsigned long long)dpp_u32<CTRL>((uint32_t)(k >> 32)) << 32) | dpp_u32<CTRL>((uint32_t)k);
    return o < k ? o : k;
}
__device__ inline unsigned long long rowmin_u64(unsigned long long k) {
    k = min_step_u64<0xB1>(k);
    k = min_step_u64<0x4E>(k);
    k = min_step_u64<0x141>(k);
    k = min_step_u64<0x140>(k);
    return k;
}

constexpr int kPitch = 256;  // entries per row of the prepared tables, whatever C is: a row is 2 KB (fp64) / 512 B (16-bit)

// GTRHQA_placer.py:99-112: sum over the op's two qubits of D[core of the qubit][c], skipping a qubit
// that is on core c already or on core 0 (label 0 is falsy in the reference's test).  The skips are
// folded into a copy of the table whose row 0 and diagonal are zero: 0.0 + x + 0.0 == x exactly, so the
// sum has the bits of the reference's conditional additions.
__device__ inline double entry(const double* Dz, uint32_t a1, uint32_t a2, uint32_t c) {
    return Dz[a1 * kPitch + c] + Dz[a2 * kPitch + c];
}

// Dz = D with row 0 and the diagonal zeroed, at a row pitch of 256 entries (columns from C on are zero: no core
// has them, they are never free), Dz16 the same as 16-bit integers; flags[0] = 1 when every entry
// of D is finite, flags[1] = 1 when every entry is an integer <= 32767 (integer edge weights): the sum of two
// then fits 16 bits, is exact in either arithmetic, and the bids can be formed from the small table.
__global__ void k_gtrhqa_prepare(const double* D, int C, double* Dz, uint16_t* Dz16, int32_t* flags) {
    __shared__ int bad, frac;
    if (threadIdx.x == 0) bad = frac = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < C * kPitch; k += blockDim.x) {
        const int r = k / kPitch, c = k - r * kPitch;
        if (c >= C) {
            Dz[k] = 0.0;
            Dz16[k] = 0;
            continue;
        }
        const double v = D[r * C + c];
        if (!isfinite(v)) bad = 1;
        const bool small_int = v >= 0.0 && v <= 32767.0 && v == floor(v);
        if (!small_int) frac = 1;
        const bool zero = r == 0 || r == c;
        Dz[k] = zero ? 0.0 : v;
        Dz16[k] = zero || !small_int ? (uint16_t)0 : (uint16_t)v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        flags[0] = bad ? 0 : 1;
        flags[1] = (bad || frac) ? 0 : 1;
    }
}

// The cheapest cores of an op, before anything is known about a slice: for every pair (a1, a2) of source cores the
// kShort cores c with the smallest entry Dz[a1][c] + Dz[a2][c], ascending by (entry, core) -- the order in which an op
// whose qubits sit on a1 and a2 would bid for cores if all of them were free.  The first core of that list that is free
// in a slice is therefore exactly the op's first minimum over the free cores, whenever the list holds a free core at
// all; only an op whose kShort best cores are all gone has to scan the table.  One thread per pair, an insertion into
// a sorted array per core; built once per call (the table is fixed for all slices of all instances).  0xFF = no core
// (fewer than kShort cores).
#ifndef QP_GT_SHORT
#define QP_GT_SHORT 16
#endif
constexpr int kShort = QP_GT_SHORT;  // a multiple of 16

__global__ __launch_bounds__(256) void k_gtrhqa_shortlist(const double* Dz, int C, uint8_t* out) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= C * C) return;
    const int a1 = pair / C, a2 = pair - a1 * C;
    unsigned long long key[kShort];
    uint32_t core[kShort];
#pragma unroll
    for (int j = 0; j < kShort; ++j) {
        key[j] = ~0ull;
        core[j] = 0xFFu;
    }
    const double* r1 = Dz + (size_t)a1 * kPitch;
    const double* r2 = Dz + (size_t)a2 * kPitch;
    for (int c = 0; c < C; ++c) {
        unsigned long long k = dbits(r1[c] + r2[c]);
        uint32_t id = (uint32_t)c;
        if (k >= key[kShort - 1]) continue;  // cores come in ascending order: an equal entry stays behind
        bool moved = false;  // from the slot the core goes into, every later slot takes its predecessor
#pragma unroll
        for (int j = 0; j < kShort; ++j) {
            const bool lt = moved || k < key[j];
            moved = lt;
            const unsigned long long kk = lt ? key[j] : k;
            const uint32_t ii = lt ? core[j] : id;
            key[j] = lt ? k : key[j];
            core[j] = lt ? id : core[j];
            k = kk;
            id = ii;
        }
    }
    uint32_t w[kShort / 4];
#pragma unroll
    for (int j = 0; j < kShort / 4; ++j)
        w[j] = core[4 * j] | (core[4 * j + 1] << 8) | (core[4 * j + 2] << 16) | (core[4 * j + 3] << 24);
#pragma unroll
    for (int j = 0; j < kShort / 16; ++j)
        *reinterpret_cast<uint4*>(out + (size_t)pair * kShort + 16 * j) = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}

// kThreads = 1024: 16 waves share an instance's bidding (shortest time for one placement);
// kThreads = 256: more instances resident per CU (highest throughput for large batches).
// SMALLINT: the instantiation for tables of small integers (bids from the 16-bit copy).  Which kind a table
// is, is known on the device only (k_gtrhqa_prepare's flag), so both instantiations are launched and the one
// that does not apply returns at once; each is compiled for the registers its own bid loop needs.
//
// Everything is indexed by core: the reference's columns are the available cores in ascending order, so "first
// minimum by column" is "first minimum by core", and a core that is not available is simply never free.
template <int kThreads, bool SMALLINT>
__global__ __launch_bounds__(kThreads, kThreads == 256 ? (SMALLINT ? QP_GT_WAVES_INT : QP_GT_WAVES) : 4) void k_gtrhqa_place(GpArgs p) {
    constexpr int kWaves = kThreads / kWave;
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_nops, s_nav, s_nleft, s_nfree, s_nneed, s_nscan, s_flag, s_wave_cnt[kWaves], s_base;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int C = p.C, Q = p.Q, ldq = p.ldq;
    uint8_t* rows = smem + p.off_rows;                                 // [2][ldq]
    uint32_t* occ = reinterpret_cast<uint32_t*>(smem + p.off_occ);     // [C]
    unsigned long long* best = reinterpret_cast<unsigned long long*>(smem + p.off_best);  // [C] by core
    uint32_t* win = reinterpret_cast<uint32_t*>(smem + p.off_win);     // [C] by core
    double* bid = reinterpret_cast<double*>(smem + p.off_bid);         // [max_ops]
    int32_t* target = reinterpret_cast<int32_t*>(smem + p.off_target); // [max_ops] core
    uint32_t* uv = reinterpret_cast<uint32_t*>(smem + p.off_uv);       // [max_ops] u | v << 16
    uint16_t* a12 = reinterpret_cast<uint16_t*>(smem + p.off_a12);     // [max_ops] a1 | a2 << 8
    OpFlags* flags = reinterpret_cast<OpFlags*>(smem + p.off_flags);   // [max_ops]
    uint8_t* cfree = smem + p.off_cores;                               // [256] by core: 1 = a column of the slice and not taken yet
    uint16_t* need_list = reinterpret_cast<uint16_t*>(smem + p.off_need);  // [max_ops] ops that must bid again
    uint16_t* scan_list = reinterpret_cast<uint16_t*>(smem + p.off_scan);  // [max_ops] those of them that must scan the table

    // all distances finite: then every entry is, and the "nothing finite left" test never fires
    const bool dfinite = p.all_finite[0] != 0;
    if ((p.all_finite[1] != 0) != SMALLINT) return;  // uniform: the other instantiation does this table

#ifdef QP_GTRHQA_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
    long long n_rounds = 0, n_need = 0, n_left = 0, n_scan = 0;
#endif
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
            for (int q = tid; q < Q; q += kThreads) atomicAdd(&occ[min((int)a[q], C - 1)], 1u);  // (a core id >= C is the caller's error; it must not index outside the table)
            __syncthreads();

            // columns: cores with room for a whole op (GTRHQA_placer.py:81-86)
            {
                int mine = 0;
                for (int c = tid; c < kPitch; c += kThreads) {
                    const bool has = c < C && (int)occ[c] <= p.cap - 2;
                    cfree[c] = has ? 1 : 0;
                    mine += has ? 1 : 0;
                }
                if (mine) atomicAdd(&s_nav, mine);
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
                    au = min((int)a[u], C - 1);
                    av = min((int)a[v], C - 1);
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
            TIC(0);
            if (nops > 0 && nav > 0) {
                // A 16-lane row of a wave serves one bidding op; lane l of the row looks at cores 32 i + 2 l and
                // 32 i + 2 l + 1, i = 0 .. 7: 16 bytes of a table row per load, 256 contiguous bytes per row of lanes.
                const int l16 = lane & 15, rowi = lane >> 4;
                const int nshort = min(C, kShort);  // entries of a short list that name a core
                if (tid == 0) {
                    s_nleft = nops;
                    s_nfree = nav;
                    s_nneed = nops;
                }
                for (int k = tid; k < nops; k += kThreads) need_list[k] = (uint16_t)k;
                __syncthreads();
                const int max_rounds = min(nav, nops) + 2;
                int rounds = 0;
                bool use_short = dfinite;  // bids from the short lists (tables with an infinite entry always scan)
                while (s_nleft > 0 && s_nfree > 0 && rounds < max_rounds) {
                    ++rounds;
                    if (!dfinite) {  // :135-137: stop when no open op has a finite entry on a free core
                        if (tid == 0) s_flag = 0;
                        __syncthreads();
                        for (int k = tid; k < nops; k += kThreads) {
                            if (!flags[k].left) continue;
                            const uint32_t a1 = a12[k] & 0xFF, a2 = a12[k] >> 8;
                            bool any = false;
                            for (int c = 0; c < C && !any; ++c)
                                any = cfree[c] && isfinite(entry(p.D, a1, a2, c));
                            if (any) s_flag = 1;
                        }
                        __syncthreads();
                        const bool go = s_flag != 0;
                        __syncthreads();
                        if (!go) break;
                    }
                    // the ops without a standing bid: all of them in the first round, afterwards the ones the
                    // last round's settlement put on the list (in no particular order: bids are independent)
                    const int nneed = s_nneed;
                    TIC(1);
#ifdef QP_GTRHQA_TIMING
                    n_rounds += 1;
                    n_need += nneed;
#endif
                    // First the short way: lane l of an op's row looks at entry l of the op's list of cheapest cores;
                    // the first of them that is free is the op's bid.  Ops whose whole list is taken go on scan_list.
                    const uint16_t* todo = need_list;
                    int ntodo = nneed;
                    if (use_short) {
                        if (tid == 0) s_nscan = 0;
                        __syncthreads();
                        // a thread per op: its list is one 16-byte load, the sixteen "is this core free" bytes are read
                        // together, the first free entry wins (walked from the back so that the earliest one is kept)
                        static_assert(kShort % 16 == 0, "a short list is read in 16-byte loads");
                        for (int e0 = 0; e0 < nneed; e0 += kThreads) {
                            const int e = e0 + tid;
                            const bool act = e < nneed;
                            const int k = need_list[act ? e : 0];
                            const uint32_t a1 = a12[k] & 0xFF, a2 = a12[k] >> 8;
                            uint32_t w[kShort / 4];
#pragma unroll
                            for (int h = 0; h < kShort / 16; ++h) {
                                const uint4 lst = *reinterpret_cast<const uint4*>(p.shortlist + ((size_t)a1 * C + a2) * kShort + 16 * h);
                                w[4 * h] = lst.x;
                                w[4 * h + 1] = lst.y;
                                w[4 * h + 2] = lst.z;
                                w[4 * h + 3] = lst.w;
                            }
                            uint32_t fr[kShort];
#pragma unroll
                            for (int j = 0; j < kShort; ++j) fr[j] = cfree[(w[j >> 2] >> (8 * (j & 3))) & 0xFFu];  // (0xFF past the list's end: inside cfree)
                            uint32_t core = 0;
                            bool found = false;
#pragma unroll
                            for (int j = kShort - 1; j >= 0; --j) {
                                if (j < nshort && fr[j] != 0) {
                                    core = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                                    found = true;
                                }
                            }
                            if (act && found) {
                                bid[k] = entry(p.D, a1, a2, core);
                                target[k] = (int)core;
                                flags[k] = OpFlags{1, 0, 1, 0};
                            }
                            const bool scan = act && !found;
                            const unsigned long long m = __ballot(scan);
                            if (m) {  // one list-counter update per wave
                                const int leader = __ffsll((long long)m) - 1;
                                int base = 0;
                                if (lane == leader) base = atomicAdd(&s_nscan, __popcll(m));
                                base = __builtin_amdgcn_readlane(base, leader);
                                if (scan) scan_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)k;
                            }
                        }
                        __syncthreads();
                        todo = scan_list;
                        ntodo = s_nscan;
                        // a slice whose cheapest cores are contested (equal entries everywhere, as integer weights
                        // give them) exhausts the lists: once most of a round's look-ups fail, the rest of the slice scans
                        if (2 * ntodo > nneed) use_short = false;
                    }
#ifdef QP_GTRHQA_TIMING
                    n_scan += ntodo;
#endif
                    TIC(6);
                    // the others look for their cheapest free core (first minimum) in the table; four ops per wave step.
                    // taken: bit 2 i + s set = this lane's core 32 i + 2 l16 + s is not a free column
                    uint32_t taken = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t two = *reinterpret_cast<const uint16_t*>(cfree + 32 * i + 2 * l16);
                        taken |= (((two & 1u) | ((two >> 7) & 2u)) ^ 3u) << (2 * i);
                    }
                    bool isfree[16];  // the same per position, held as lane masks for the compares below
#pragma unroll
                    for (int j = 0; j < 16; ++j) isfree[j] = ((taken >> j) & 1u) == 0;
                    for (int nb = wave * 4; nb < ntodo; nb += kWaves * 4) {
                        const int e = nb + rowi;
                        const bool act = e < ntodo;
                        const int k = todo[act ? e : nb];  // a row without an op of its own repeats the first
                        const uint32_t a1 = a12[k] & 0xFF, a2 = a12[k] >> 8;
                        // key = bits of the entry (non-negative doubles order like their bits); a lane looks at its
                        // cores in ascending order, so "<" keeps its first minimum; among the lanes of the row the
                        // smallest core with the row's minimum wins.  A core that is not free gets all ones in the
                        // key's upper half, which no entry reaches.  The 16 loads of a lane are issued together, with
                        // no branch between them: a step costs one memory latency.
                        uint32_t sel = 0;  // position 2 i + s of the lane's minimum
                        unsigned long long row_key;
                        uint32_t fin = 1;
                        if constexpr (SMALLINT) {  // integer table: two 16-bit entries per load, packed 16-bit sums
                            const char* hbase = reinterpret_cast<const char*>(p.D16);
                            const uint32_t h1 = a1 * (uint32_t)(kPitch * 2) + (uint32_t)l16 * 4u, h2 = a2 * (uint32_t)(kPitch * 2) + (uint32_t)l16 * 4u;
                            uint32_t s1[8], s2[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                s1[i] = *reinterpret_cast<const uint32_t*>(hbase + (h1 + i * 64));
                                s2[i] = *reinterpret_cast<const uint32_t*>(hbase + (h2 + i * 64));
                            }
                            uint32_t key32 = ~0u;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const uint32_t sum = s1[i] + s2[i];  // each half < 65536: no carry between the halves
#pragma unroll
                                for (int sft = 0; sft < 2; ++sft) {
                                    const uint32_t kk = sft ? sum >> 16 : sum & 0xFFFFu;
                                    if (isfree[2 * i + sft] && kk < key32) {
                                        key32 = kk;
                                        sel = (uint32_t)(2 * i + sft);
                                    }
                                }
                            }
                            const uint32_t m32 = rowmin_u32(key32);
                            const uint32_t cand = key32 == m32 ? (sel >> 1) * 32u + (sel & 1u) + 2u * (uint32_t)l16 : 0xFFFFu;
                            sel = rowmin_u32(cand);
                            row_key = dbits((double)m32);  // the same bits the fp64 sum of the two (integer) entries has
                        } else if (dfinite) {
                            const char* dbase = reinterpret_cast<const char*>(p.D);
                            const uint32_t r1 = a1 * (uint32_t)(kPitch * 8) + (uint32_t)l16 * 16u, r2 = a2 * (uint32_t)(kPitch * 8) + (uint32_t)l16 * 16u;
                            unsigned long long key = ~0ull;
                            // two batches of eight loads: the registers of one batch leave room for six waves per SIMD,
                            // whose loads overlap (all sixteen at once cost more in occupancy than they saved in waiting)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                double2 d1[4], d2[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    d1[i] = *reinterpret_cast<const double2*>(dbase + (r1 + (4 * h + i) * 256));
                                    d2[i] = *reinterpret_cast<const double2*>(dbase + (r2 + (4 * h + i) * 256));
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
#pragma unroll
                                    for (int sft = 0; sft < 2; ++sft) {
                                        const double val = sft ? d1[i].y + d2[i].y : d1[i].x + d2[i].x;
                                        const unsigned long long kk = dbits(val);
                                        if (isfree[8 * h + 2 * i + sft] && kk < key) {
                                            key = kk;
                                            sel = (uint32_t)(8 * h + 2 * i + sft);
                                        }
                                    }
                                }
                            }
                            row_key = rowmin_u64(key);
                            const uint32_t cand = key == row_key ? (sel >> 1) * 32u + (sel & 1u) + 2u * (uint32_t)l16 : 0xFFFFu;
                            sel = rowmin_u32(cand);
                        } else {  // some distance is infinite: such an entry bids 1e12 and cannot win a core (:139, :178)
                            unsigned long long key = ~0ull;
                            uint32_t idx = 0x7FFFFFFFu;
#pragma unroll 4
                            for (int i = 0; i < 16; ++i) {
                                const uint32_t c = (uint32_t)((i >> 1) * 32 + (i & 1) + 2 * l16);
                                const double val = entry(p.D, a1, a2, c);
                                const bool f = isfinite(val);
                                const unsigned long long kk = dbits(f ? val : kBig);
                                if (!((taken >> i) & 1u) && kk < key) {
                                    key = kk;
                                    idx = c | (f ? 0x10000u : 0u);
                                }
                            }
                            lexmin_row(key, idx);
                            row_key = key;
                            sel = idx & 0xFFFFu;
                            fin = (idx >> 16) & 1u;
                        }
                        if (act && l16 == 0) {
                            bid[k] = __longlong_as_double((long long)row_key);
                            target[k] = (int)sel;
                            flags[k] = OpFlags{1, 0, (uint8_t)fin, 0};
                        }
                    }
                    for (int c = tid; c < C; c += kThreads) {
                        best[c] = ~0ull;
                        win[c] = ~0u;
                    }
                    __syncthreads();         // every thread has read s_nneed and is done with the list
                    if (tid == 0) s_nneed = 0;  // (ordered before the appends below by the barriers in between)
                    TIC(2);
                    // a core takes its cheapest bidder, the first one in op order among equals (:150-176)
                    for (int k = tid; k < nops; k += kThreads)
                        if (flags[k].left) atomicMin(&best[target[k]], dbits(bid[k]));
                    __syncthreads();
                    for (int k = tid; k < nops; k += kThreads)
                        if (flags[k].left && dbits(bid[k]) == best[target[k]]) atomicMin(&win[target[k]], (uint32_t)k);
                    __syncthreads();
                    for (int c = tid; c < C; c += kThreads) {
                        if (!cfree[c] || win[c] == ~0u) continue;
                        const int k = (int)win[c];
                        if (!flags[k].fin) continue;  // :178-179 a non-finite winner leaves both open
                        flags[k].left = 0;
                        cfree[c] = 0;
                        const uint32_t w = uv[k];
                        a[w & 0xFFFF] = (uint8_t)c;
                        a[w >> 16] = (uint8_t)c;
                        atomicSub(&s_nleft, 1);
                        atomicSub(&s_nfree, 1);
                    }
                    __syncthreads();
                    for (int kb0 = 0; kb0 < nops; kb0 += kThreads) {
                        const int k = kb0 + tid;
                        // its core went to someone else: bid again (one list-counter update per wave)
                        const bool again = k < nops && flags[k].left && !cfree[target[k]];
                        const unsigned long long m = __ballot(again);
                        if (m) {
                            const int leader = __ffsll((long long)m) - 1;
                            int base = 0;
                            if (lane == leader) base = atomicAdd(&s_nneed, __popcll(m));
                            base = __builtin_amdgcn_readlane(base, leader);
                            if (again) {
                                flags[k].need = 1;
                                need_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)k;
                            }
                        }
                    }
                    __syncthreads();
                    TIC(3);
                }
                // :222-254 what is still open takes its cheapest remaining core, op by op
#ifdef QP_GTRHQA_TIMING
                n_left += s_nleft;
#endif
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
                        for (int c = tid; c < C; c += kThreads) {
                            if (!cfree[c]) continue;
                            const double val = entry(p.D, a1, a2, c);
                            const bool f = isfinite(val);
                            if (f) s_flag = 1;
                            atomicMin(&best[0], dbits(f ? val : kBig));
                        }
                        __syncthreads();
                        for (int c = tid; c < C; c += kThreads) {
                            if (!cfree[c]) continue;
                            const double val = entry(p.D, a1, a2, c);
                            if (dbits(isfinite(val) ? val : kBig) == best[0]) atomicMin(&win[0], (uint32_t)c);
                        }
                        __syncthreads();
                        const bool any = s_flag != 0;
                        const int c = (int)win[0];
                        __syncthreads();
                        if (any && tid == 0) {
                            flags[k].left = 0;
                            cfree[c] = 0;
                            const uint32_t w = uv[k];
                            a[w & 0xFFFF] = (uint8_t)c;
                            a[w >> 16] = (uint8_t)c;
                            s_nfree -= 1;
                            s_nleft -= 1;
                        }
                        __syncthreads();
                    }
                }
            }
            __syncthreads();
            TIC(4);
            uint8_t* orow = outb + (size_t)(t + 1) * ldq;
            for (int k = tid * 16; k < ldq; k += kThreads * 16) {  // written once: keep it out of the caches
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(a + k), reinterpret_cast<u32x4*>(orow + k));
            }
            __syncthreads();
            TIC(5);
        }
#ifdef QP_GTRHQA_TIMING
        if (tid == 0 && b == 0)
            printf("gtrhqa phases (cycles, instance 0, %d threads): setup %lld  needlist %lld  short lists %lld  scans %lld  resolve %lld  leftover %lld  store %lld | rounds %lld  bids %lld  of them scans %lld  open after the rounds %lld\n",
                   kThreads, tacc[0], tacc[1], tacc[6], tacc[2], tacc[3], tacc[4], tacc[5], n_rounds, n_need, n_scan, n_left);
#endif
    }
}

}  // namespace

extern "C" size_t qp_gtrhqa_workspace(int32_t C) {
    // fp64 copy at a row pitch of 256 entries, two flags, 16-bit copy at the same pitch, the short lists
    return C > 0 ? (size_t)C * kPitch * (sizeof(double) + sizeof(uint16_t)) + 16 + (size_t)C * C * kShort : 16;
}

extern "C" int qp_gtrhqa_place(const QpCircuit* circ, const QpDTable* dt, const uint8_t* initial, int64_t B, int32_t ldq,
                               int32_t cap, uint8_t* out, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!circ || !dt || !initial || !out) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: null argument");
    if (B <= 0) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: B must be positive");
    if (circ->Q <= 0 || circ->Q > 65535 || circ->T < 0) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: bad circuit sizes");
    if (ldq < circ->Q || ldq % 16 != 0) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: ldq must be >= Q and a multiple of 16");
    if (dt->kind != QP_DT_F64 || !dt->table) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: needs a QP_DT_F64 distance table");
    if (dt->C <= 0 || dt->C > 256) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: bad core count");
    if (cap < 2) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: capacity must be at least 2");
    if (circ->T > 0 && (!circ->slice_edge_ptr || (circ->nE > 0 && !circ->edges)))
        return qp::fail(QP_EINVAL, "qp_gtrhqa_place: circuit edge tables missing");
    if (!workspace || workspace_bytes < qp_gtrhqa_workspace(dt->C) || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return qp::fail(QP_ENOSPC, "qp_gtrhqa_place: workspace needs %zu bytes, 16-byte aligned", qp_gtrhqa_workspace(dt->C));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* Dz = static_cast<double*>(workspace);
    int32_t* all_finite = reinterpret_cast<int32_t*>(Dz + (size_t)dt->C * kPitch);
    uint16_t* Dz16 = reinterpret_cast<uint16_t*>(all_finite + 4);
    uint8_t* shortlist = reinterpret_cast<uint8_t*>(Dz16 + (size_t)dt->C * kPitch);  // 16-byte aligned: everything before it is
    k_gtrhqa_prepare<<<1, 1024, 0, stream>>>(static_cast<const double*>(dt->table), dt->C, Dz, Dz16, all_finite);
    HIP_TRY(hipGetLastError());
    k_gtrhqa_shortlist<<<(unsigned)((dt->C * dt->C + 255) / 256), 256, 0, stream>>>(Dz, dt->C, shortlist);
    HIP_TRY(hipGetLastError());

    // the most counted gates any slice has sizes the per-op arrays: the caller's figure if it gave one
    // (slice_edge_ptr lives on the device), else Q/2 (a qubit is in at most one gate per slice)
    const int32_t max_ops = std::max(1, circ->max_slice_edges > 0 ? std::min(circ->max_slice_edges, circ->Q / 2) : circ->Q / 2);
    GpArgs p{};
    p.sep = circ->slice_edge_ptr;
    p.edges = circ->edges;
    p.D = Dz;
    p.D16 = Dz16;
    p.all_finite = all_finite;
    p.shortlist = shortlist;
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
    p.off_cores = take((size_t)kPitch);
    p.off_need = take((size_t)2 * max_ops);
    p.off_scan = take((size_t)2 * max_ops);
    if (off > 160u * 1024u) return qp::fail(QP_EINVAL, "qp_gtrhqa_place: circuit too wide for the on-chip state (%u bytes)", off);
    const int cus = qp::device_cus();
    int per_cu = 1;
    // few instances: give each one a whole 16-wave workgroup; many: small workgroups, more of them per CU
    const bool wide = B <= (int64_t)2 * cus;
    const int threads = wide ? 1024 : 256;
    void (*kerns[2])(GpArgs) = {wide ? k_gtrhqa_place<1024, false> : k_gtrhqa_place<256, false>,
                                wide ? k_gtrhqa_place<1024, true> : k_gtrhqa_place<256, true>};
    for (auto kern : kerns) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)off));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), threads, off));
        const unsigned grid = (unsigned)std::min<int64_t>(B, (int64_t)cus * std::max(per_cu, 1));
        kern<<<grid, threads, off, stream>>>(p);
        HIP_TRY(hipGetLastError());
    }
    return QP_OK;
}

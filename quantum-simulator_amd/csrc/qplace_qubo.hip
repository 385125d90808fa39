// QUBO placer on gfx950: exact integer energies / local fields of R replica states, and replica-parallel
// simulated annealing that works from the model's structure with one wave per replica.
//
// The coefficient matrix of a timeslice is a Kronecker sum (variables v = j * P + k: core j, k-th variable
// of the core):  I_C (x) same  +  (J - I)_C (x) diag(8 on the qubit rows),  and the coupling to the next slice
// is 10 * hops (x) (matching of the qubits both slices have).  Only `same` (W x W, the same for every core)
// is a dense matrix: it goes through an int8 MFMA contraction, one W x W x R product per core.  The other two
// terms only ever couple the C bits "qubit k on core j": they are evaluated from one C-bit mask per (slice,
// qubit, replica), which a small kernel extracts from the states first.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "qplace_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// F_same = same . X_j  (and  sum_v x_v (2 a_v + F_same_v))  with v_mfma_i32_32x32x32_i8
//
// Workgroup = 4 waves (2 x 2), tile = 128 rows of `same` x 64 WN replicas for one core of one slice.  Both
// operands are contiguous along the contraction axis (`same` row-major, states stored replica-major), so a
// lane's A or B fragment is 16 consecutive bytes.  Tiles go through LDS with a 144-byte row pitch: 16
// consecutive rows then start in 16 different 16-byte slots of the 256-byte bank row, which is what
// ds_read_b128 needs to run conflict-free.
// ------------------------------------------------------------------------------------------------

constexpr int kTile = 128;
constexpr int kBK = 128;
constexpr int kPitch = 144;  // 128 + 16: consecutive rows start 9 sixteen-byte slots apart (odd => conflict-free)

struct KronArgs {
    QpQuboKron m;
    const int8_t* Xt;  // [R][Npad]
    int32_t* out;      // fields: [Npad][R]; energy: [R] (accumulated with integer atomics)
    uint32_t* masks;   // [T][nmax][R]: bit j = qubit on core j
    int32_t R;
};

// What a workgroup does with its accumulators: store the fields, or fold them into the replicas' energies.
// rows: the tile's first row inside the core's block (krow0), valid while < P; v0 = state index of row krow0.
template <bool ENERGY, int WN>
__device__ inline void kron_epilogue(const KronArgs& p, v16i (&acc)[2][WN], int64_t v0, int krow0, int P, int r0, int wm,
                                     int wn, int lane, const uint16_t* sbits) {
    // accumulator element k of lane l: row (k & 3) + 8 * (k >> 2) + 4 * (l >> 5), column l & 31
    const int h = lane >> 5;
    if constexpr (!ENERGY) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                const int col = r0 + wn * 32 * WN + b * 32 + (lane & 31);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int i = wm * 64 + a * 32 + (k & 3) + 8 * (k >> 2) + 4 * h;
                    if (krow0 + i < P) p.out[(size_t)(v0 + i) * p.R + col] = acc[a][b][k];
                }
            }
    } else {
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            const int col = r0 + wn * 32 * WN + b * 32 + (lane & 31);
            int s = 0;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = wm * 64 + a * 32 + 8 * g + 4 * h;  // four consecutive rows; P is a multiple of 32
                    if (krow0 + i < P) {
                        // the states of rows i..i+3 of this replica: four bits of the tile's own piece of the B operand
                        const uint32_t x4 = (uint32_t)sbits[(col - r0) * (kTile / 16) + (i >> 4)] >> (i & 15);
                        const int4 lin = *reinterpret_cast<const int4*>(p.m.linear + v0 + i);
                        const int l4[4] = {lin.x, lin.y, lin.z, lin.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((x4 >> j) & 1u) s += 2 * l4[j] + acc[a][b][4 * g + j];
                    }
                }
            s += __shfl_xor(s, 32, 64);
            if (h == 0) atomicAdd(&p.out[col], s);
        }
    }
}

// bit k of the result = lowest bit of byte k of the 16 bytes
__device__ inline uint16_t low_bits16(const uint4& v) {
    // byte i's low bit lands on bit 24 + i (8 i + 24 - 7 i); the other partial products stay below bit 24
    auto nib = [](uint32_t x) -> uint32_t { return ((x & 0x01010101u) * 0x01020408u) >> 24; };
    return (uint16_t)(nib(v.x) | nib(v.y) << 4 | nib(v.z) << 8 | nib(v.w) << 12);
}

// WN = MFMA tiles per wave along the replica axis: 2 (128 replicas per workgroup) or 4 (256).  With 4, a
// wave feeds 8 MFMAs from 6 fragment reads instead of 4 from 4, which is what the LDS read rate allows.
template <bool ENERGY, int WN>
__global__ __launch_bounds__(256, 2) void k_qubo_kron(KronArgs p) {
    constexpr int kCols = 64 * WN;  // replicas per workgroup
    __shared__ __align__(16) int8_t sA[kTile * kPitch];
    __shared__ __align__(16) int8_t sB[kCols * kPitch];
    // energy: bit k of sbits[replica][piece] = state of the tile's row piece * 16 + k (the rows of the tile are a
    // 128-byte piece of the B operand: it is caught on its way through the staging registers)
    __shared__ uint16_t sbits[ENERGY ? kCols * (kTile / 16) : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.y, coltile = blockIdx.x;  // the replica tiles of one row tile run side by side
    const int t = p.m.tile_slice[tile];
    const int P = p.m.pitch[t];
    const int Pp = (P + kTile - 1) / kTile * kTile;  // rows and row pitch of the slice's `same` matrix
    const int MT = Pp / kTile;
    const int local = tile - p.m.tile_first[t];
    const int j = local / MT, mt = local - j * MT;
    const int krow0 = mt * kTile;
    const int64_t vcore = p.m.off[t] + (int64_t)j * P;  // first state byte of this core's variables
    const int r0 = coltile * kCols;

    v16i acc[2][WN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[a][b][k] = 0;

    // staging: thread -> (row = tid / 8 + 32 i, 16-byte piece = tid % 8) of a (rows x 128)-byte tile; the
    // next tile's global loads are issued before the current tile's MFMAs so they overlap.  `same` is padded
    // with zero columns up to Pp; the states are not read past the core's P bytes (zeros instead).
    const int srow = tid >> 3, spiece = tid & 7;
    const int8_t* Ab = p.m.same + p.m.same_off[t] + (size_t)krow0 * Pp;
    const int8_t* Bb = p.Xt + (size_t)r0 * p.m.Npad + vcore;
    // (named registers and a macro: staging arrays, or a lambda capturing them, end up in scratch memory)
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
#define QP_LOAD_TILE(k0)                                                                                       \
    do {                                                                                                      \
        const int8_t* pa_ = Ab + (size_t)srow * Pp + (k0) + spiece * 16;                                      \
        const int8_t* pb_ = Bb + (size_t)srow * p.m.Npad + (k0) + spiece * 16;                                \
        const bool in_ = (k0) + spiece * 16 < P;                                                              \
        ra0 = *reinterpret_cast<const uint4*>(pa_);                                                           \
        ra1 = *reinterpret_cast<const uint4*>(pa_ + (size_t)32 * Pp);                                         \
        ra2 = *reinterpret_cast<const uint4*>(pa_ + (size_t)64 * Pp);                                         \
        ra3 = *reinterpret_cast<const uint4*>(pa_ + (size_t)96 * Pp);                                         \
        rb0 = in_ ? *reinterpret_cast<const uint4*>(pb_) : zero4;                                             \
        rb1 = in_ ? *reinterpret_cast<const uint4*>(pb_ + (size_t)32 * p.m.Npad) : zero4;                     \
        rb2 = in_ ? *reinterpret_cast<const uint4*>(pb_ + (size_t)64 * p.m.Npad) : zero4;                     \
        rb3 = in_ ? *reinterpret_cast<const uint4*>(pb_ + (size_t)96 * p.m.Npad) : zero4;                     \
        if constexpr (WN == 4) {                                                                              \
            rb4 = in_ ? *reinterpret_cast<const uint4*>(pb_ + (size_t)128 * p.m.Npad) : zero4;                \
            rb5 = in_ ? *reinterpret_cast<const uint4*>(pb_ + (size_t)160 * p.m.Npad) : zero4;                \
            rb6 = in_ ? *reinterpret_cast<const uint4*>(pb_ + (size_t)192 * p.m.Npad) : zero4;                \
            rb7 = in_ ? *reinterpret_cast<const uint4*>(pb_ + (size_t)224 * p.m.Npad) : zero4;                \
        }                                                                                                     \
    } while (0)
    QP_LOAD_TILE(0);
    for (int k0 = 0; k0 < P; k0 += kBK) {
        __syncthreads();
        {
            int8_t* wa = sA + srow * kPitch + spiece * 16;
            int8_t* wb = sB + srow * kPitch + spiece * 16;
            *reinterpret_cast<uint4*>(wa) = ra0;
            *reinterpret_cast<uint4*>(wa + 32 * kPitch) = ra1;
            *reinterpret_cast<uint4*>(wa + 64 * kPitch) = ra2;
            *reinterpret_cast<uint4*>(wa + 96 * kPitch) = ra3;
            *reinterpret_cast<uint4*>(wb) = rb0;
            *reinterpret_cast<uint4*>(wb + 32 * kPitch) = rb1;
            *reinterpret_cast<uint4*>(wb + 64 * kPitch) = rb2;
            *reinterpret_cast<uint4*>(wb + 96 * kPitch) = rb3;
            if constexpr (WN == 4) {
                *reinterpret_cast<uint4*>(wb + 128 * kPitch) = rb4;
                *reinterpret_cast<uint4*>(wb + 160 * kPitch) = rb5;
                *reinterpret_cast<uint4*>(wb + 192 * kPitch) = rb6;
                *reinterpret_cast<uint4*>(wb + 224 * kPitch) = rb7;
            }
            if constexpr (ENERGY) {
                if (k0 == krow0) {  // this K tile holds the states of the tile's own rows
                    uint16_t* sb = sbits + srow * (kTile / 16) + spiece;
                    sb[0] = low_bits16(rb0);
                    sb[32 * (kTile / 16)] = low_bits16(rb1);
                    sb[64 * (kTile / 16)] = low_bits16(rb2);
                    sb[96 * (kTile / 16)] = low_bits16(rb3);
                    if constexpr (WN == 4) {
                        sb[128 * (kTile / 16)] = low_bits16(rb4);
                        sb[160 * (kTile / 16)] = low_bits16(rb5);
                        sb[192 * (kTile / 16)] = low_bits16(rb6);
                        sb[224 * (kTile / 16)] = low_bits16(rb7);
                    }
                }
            }
        }
        __syncthreads();
        if (k0 + kBK < P) QP_LOAD_TILE(k0 + kBK);
#pragma unroll
        for (int kk = 0; kk < kBK / 32; ++kk) {
            if (k0 + kk * 32 >= P) break;  // P is a multiple of 32: whole MFMA steps
            v4i fa[2], fb[WN];
            const int koff = kk * 32 + (lane >> 5) * 16;
#pragma unroll
            for (int a = 0; a < 2; ++a)
                fa[a] = *reinterpret_cast<const v4i*>(sA + (wm * 64 + a * 32 + (lane & 31)) * kPitch + koff);
#pragma unroll
            for (int b = 0; b < WN; ++b)
                fb[b] = *reinterpret_cast<const v4i*>(sB + (wn * 32 * WN + b * 32 + (lane & 31)) * kPitch + koff);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }
#undef QP_LOAD_TILE
    kron_epilogue<ENERGY, WN>(p, acc, vcore + krow0, krow0, P, r0, wm, wn, lane, sbits);
}

// The same contraction with `same` in registers: workgroup = (slice, 64 replicas), 8 waves.  Wave w owns the
// 32-row tile w of `same[t]` and one 32-replica half of a tile from 8 on, and keeps their MFMA A fragments (up to
// 2 x 12 x 16 bytes per lane) in registers for the whole kernel; the workgroup then walks the C cores of the slice: the 64 x P bytes of states
// of a core are the B operand, staged through two LDS buffers from two register sets, so that a tile's loads
// are issued two tiles before it is used and one barrier per tile suffices.  No operand is re-read from device
// memory (the states once, `same` once per workgroup from cache), no A fragment is read from LDS, two waves
// per SIMD overlap one's epilogue with the other's MFMAs, and the energy leaves as one atomic per replica and
// wave.  For pitches up to 384 (12 tiles of 32).
constexpr int kRegK = 12;                                          // 32-wide steps along the contraction axis: pitch <= 384
constexpr int kRegCols = 64;                                       // replicas per workgroup
constexpr int kRegThreads = 512;
constexpr int kRegPieces = kRegCols * kRegK * 2 / kRegThreads;     // 16-byte pieces of a states tile per thread

template <bool ENERGY>
__global__ __launch_bounds__(kRegThreads) void k_qubo_kron_areg(KronArgs p) {
    extern __shared__ __align__(16) int8_t smem_areg[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int t = blockIdx.y, r0 = blockIdx.x * kRegCols;
    const int P = p.m.pitch[t], C = p.m.C;
    const int KT = P / 32;  // K steps, and 32-row tiles of `same`
    const int Pp = (P + kTile - 1) / kTile * kTile;
    const int bpitch = P + 16;  // an odd number of 16-byte slots: conflict-free ds_read_b128
    int8_t* sB0 = smem_areg;
    int8_t* sB1 = smem_areg + kRegCols * bpitch;
    int32_t* sLin0 = reinterpret_cast<int32_t*>(smem_areg + 2 * kRegCols * bpitch);
    int32_t* sLin1 = sLin0 + P;

    // this thread's pieces of a states tile: (replica row, 16-byte piece) -> offsets in device memory and in LDS
    const int ppr = P / 16;
    // (offsets of buffer 1 are relative to sB1; a spare slot lies behind both buffers and the linear terms)
    const int spare = 2 * kRegCols * bpitch + 2 * P * (int)sizeof(int32_t) + tid * 16;
    int64_t goff[kRegPieces];
    int loff0[kRegPieces], loff1[kRegPieces];
#pragma unroll
    for (int i = 0; i < kRegPieces; ++i) {
        const int q = tid + kRegThreads * i;
        const bool on = q < kRegCols * ppr;
        const int row = on ? q / ppr : 0, pc = on ? q - row * ppr : 0;
        goff[i] = (int64_t)row * p.m.Npad + pc * 16;
        loff0[i] = on ? row * bpitch + pc * 16 : spare;
        loff1[i] = on ? row * bpitch + pc * 16 : spare - kRegCols * bpitch;
    }
    // Core masks, built on the way: the state bytes of core j (0 or 1) are OR-ed, shifted left by j & 7, into a
    // packed copy of the tile per group of 8 cores (byte k of group g = bits 8g..8g+7 of qubit k's mask), each thread
    // into its own pieces; after the last core the workgroup unpacks them into masks[t][k][r].
    const int mask_base = 2 * kRegCols * bpitch + 2 * P * (int)sizeof(int32_t) + kRegThreads * 16;
    const int NG = (C + 7) / 8;
    int moff[kRegPieces], mstep[kRegPieces];
#pragma unroll
    for (int i = 0; i < kRegPieces; ++i) {
        const int q = tid + kRegThreads * i;
        const bool on = q < kRegCols * ppr;
        moff[i] = on ? mask_base + q * 16 : spare;  // dense: row * P + piece * 16 = q * 16
        mstep[i] = on ? kRegCols * P : 0;           // a piece that does not exist keeps hitting the spare slot
        for (int g = 0; g < NG; ++g)
            if (on) *reinterpret_cast<uint4*>(smem_areg + moff[i] + g * mstep[i]) = make_uint4(0, 0, 0, 0);
    }
    // the linear terms of a tile: P / 4 threads load and store 16 bytes each, the others their spare slot
    const int slin0 = tid < P / 4 ? tid * 4 : (spare - 2 * kRegCols * bpitch) / 4;  // in int32 units from sLin0
    const int slin1 = tid < P / 4 ? tid * 4 : slin0 - P;
    const int lin4 = tid < P / 4 ? tid * 4 : 0;
    const int8_t* Xbase = p.Xt + (size_t)r0 * p.m.Npad + p.m.off[t];
    const int32_t* Lbase = p.m.linear + p.m.off[t];
    // (named registers and macros: as arrays the two register sets end up in scratch memory)
    static_assert(kRegPieces == 3, "three named staging registers per set");
    uint4 rs0_0, rs0_1, rs0_2, rs1_0, rs1_1, rs1_2;
    int4 rl0 = make_int4(0, 0, 0, 0), rl1 = make_int4(0, 0, 0, 0);
    rs0_0 = rs0_1 = rs0_2 = rs1_0 = rs1_1 = rs1_2 = make_uint4(0, 0, 0, 0);
// every load is issued whatever the thread and the tile (a piece that does not exist reads the tile's first bytes,
// a tile past the last core reads the last one again): with loads under branches the compiler cannot count how
// many newer loads are in flight and waits for all of them before a set is stored, prefetch included
#define QP_AREG_LOAD1(j_, set_, i_) \
    rs##set_##_##i_ = *reinterpret_cast<const uint4*>(Xbase + (int64_t)min((j_), C - 1) * P + goff[i_])
#define QP_AREG_LOAD(j_, set_)                                                                                     \
    do {                                                                                                          \
        QP_AREG_LOAD1(j_, set_, 0);                                                                               \
        QP_AREG_LOAD1(j_, set_, 1);                                                                               \
        QP_AREG_LOAD1(j_, set_, 2);                                                                               \
        if constexpr (ENERGY) rl##set_ = *reinterpret_cast<const int4*>(Lbase + (int64_t)min((j_), C - 1) * P + lin4); \
    } while (0)
// stores are unconditional too (a piece that does not exist goes to a spare slot of the thread's own)
#define QP_AREG_STORE1(set_, buf_, i_, jt_)                                                        \
    do {                                                                                          \
        *reinterpret_cast<uint4*>(sB##buf_ + loff##buf_[i_]) = rs##set_##_##i_;                   \
        uint4* mp_ = reinterpret_cast<uint4*>(smem_areg + moff[i_] + ((jt_) >> 3) * mstep[i_]);  \
        uint4 mv_ = *mp_;                                                                         \
        mv_.x |= rs##set_##_##i_.x << ((jt_) & 7);                                                \
        mv_.y |= rs##set_##_##i_.y << ((jt_) & 7);                                                \
        mv_.z |= rs##set_##_##i_.z << ((jt_) & 7);                                                \
        mv_.w |= rs##set_##_##i_.w << ((jt_) & 7);                                                \
        *mp_ = mv_;                                                                               \
    } while (0)
#define QP_AREG_STORE(set_, buf_, j_)                                                   \
    do {                                                                               \
        const int jt_ = min((j_), C - 1); /* a tile stored twice ORs the same bits */   \
        QP_AREG_STORE1(set_, buf_, 0, jt_);                                            \
        QP_AREG_STORE1(set_, buf_, 1, jt_);                                            \
        QP_AREG_STORE1(set_, buf_, 2, jt_);                                            \
        if constexpr (ENERGY) *reinterpret_cast<int4*>(sLin##buf_ + slin##buf_) = rl##set_;  \
    } while (0)
    QP_AREG_LOAD(0, 0);
    QP_AREG_LOAD(1, 1);

    // Work of a wave: the whole 32-row tile `wave` (both 32-replica halves), and, of the tiles 8.. that are left
    // when the slice has more than 8, one half: tile 8 + wave / 2, replica half wave & 1.
    const int mx = 8 + (wave >> 1), bx = wave & 1;
    const bool extra = mx < KT;
    int esum[3] = {0, 0, 0};  // replica halves 0 and 1 of the own tile, the extra half

    // One tile of states: core j against this wave's rows.  The number of K steps and whether the wave has an extra
    // half tile are compile-time here, so the body is straight-line code: fragment reads run ahead of the MFMAs and
    // the epilogue of one accumulator overlaps the MFMAs of the next.
    auto tile = [&](auto ktc, auto extrac, const auto& fa, int j, const int8_t* B, const int32_t* lin) {
        constexpr int KT_ = decltype(ktc)::value;
        constexpr bool EXTRA = decltype(extrac)::value;
        constexpr int NU = EXTRA ? 3 : 2;  // units of 32 rows x 32 replicas
        v16i acc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[u][k] = 0;
        const int8_t* Bu[3] = {B + l31 * bpitch + h * 16, B + (32 + l31) * bpitch + h * 16, B + (bx * 32 + l31) * bpitch + h * 16};
#pragma unroll
        for (int kk = 0; kk < KT_; ++kk) {
            v4i fb[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) fb[u] = *reinterpret_cast<const v4i*>(Bu[u] + kk * 32);
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u >> 1][kk], fb[u], acc[u], 0, 0, 0);
        }
        // accumulator element k of lane l: row (k & 3) + 8 * (k >> 2) + 4 * (l >> 5) of the tile, column l & 31
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int mt = u < 2 ? wave : mx;
            const int col = (u < 2 ? u : bx) * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = mt * 32 + 8 * g + 4 * h;  // four consecutive rows
                if constexpr (ENERGY) {
                    const int4 l4 = *reinterpret_cast<const int4*>(lin + i);
                    const int l2[4] = {2 * l4.x, 2 * l4.y, 2 * l4.z, 2 * l4.w};
                    // state bytes are 0 or 1 (the MFMA multiplies by the byte as it is, so does this)
                    const uint32_t x4 = *reinterpret_cast<const uint32_t*>(B + col * bpitch + i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) esum[u] += __mul24(acc[u][4 * g + q] + l2[q], (int)((x4 >> (8 * q)) & 0xFFu));
                } else {
                    const int64_t v = p.m.off[t] + (int64_t)j * P + i;
#pragma unroll
                    for (int q = 0; q < 4; ++q) p.out[(size_t)(v + q) * p.R + r0 + col] = acc[u][4 * g + q];
                }
            }
        }
    };
    // the walk over the cores, once per (K steps, extra half tile or not)
    auto walk = [&](auto ktc, auto twoc) {
        constexpr int KT_ = decltype(ktc)::value;
        const bool rows = wave < KT_;  // a wave without rows only stages
        // A fragments of this wave's tile (and of the extra one): lane (row l31, 16-byte half h of a 32-byte step)
        v4i fa[decltype(twoc)::value ? 2 : 1][KT_];
        {
            const int8_t* Abase = p.m.same + p.m.same_off[t];
#pragma unroll
            for (int a = 0; a < (decltype(twoc)::value ? 2 : 1); ++a) {
                const int mt = min(a == 0 ? wave : mx, KT_ - 1);  // (a wave without rows loads the last tile and never uses it)
#pragma unroll
                for (int kk = 0; kk < KT_; ++kk)
                    fa[a][kk] = *reinterpret_cast<const v4i*>(Abase + (size_t)(mt * 32 + l31) * Pp + kk * 32 + h * 16);
            }
        }
        QP_AREG_STORE(0, 0, 0);
        QP_AREG_LOAD(2, 0);
        int j = 0;
        for (; j + 1 < C; j += 2) {  // no branch around a load or a store in here: the waits the compiler places stay exact
            __syncthreads();  // tile j is in buffer 0, and every wave is done with buffer 1
            QP_AREG_STORE(1, 1, j + 1);
            QP_AREG_LOAD(j + 3, 1);
            if (rows) tile(ktc, twoc, fa, j, sB0, sLin0);
            __syncthreads();
            QP_AREG_STORE(0, 0, j + 2);  // (the last tile once more when there is none: never used)
            QP_AREG_LOAD(j + 4, 0);
            if (rows) tile(ktc, twoc, fa, j + 1, sB1, sLin1);
        }
        if (j < C) {
            __syncthreads();
            if (rows) tile(ktc, twoc, fa, j, sB0, sLin0);
        }
    };
#define QP_AREG_CASE(k_)                                                                         \
    case k_:                                                                                    \
        if (k_ > 8 && extra) walk(std::integral_constant<int, k_>{}, std::true_type{});         \
        else walk(std::integral_constant<int, k_>{}, std::false_type{});                        \
        break
    switch (KT) {
        QP_AREG_CASE(1);
        QP_AREG_CASE(2);
        QP_AREG_CASE(3);
        QP_AREG_CASE(4);
        QP_AREG_CASE(5);
        QP_AREG_CASE(6);
        QP_AREG_CASE(7);
        QP_AREG_CASE(8);
        QP_AREG_CASE(9);
        QP_AREG_CASE(10);
        QP_AREG_CASE(11);
        QP_AREG_CASE(12);
        default: break;
    }
#undef QP_AREG_CASE
#undef QP_AREG_LOAD
#undef QP_AREG_LOAD1
#undef QP_AREG_STORE
#undef QP_AREG_STORE1
    // the masks of the slice's qubits for this workgroup's replicas: lane = replica, 256 contiguous bytes per qubit
    __syncthreads();
    {
        const int n = p.m.n[t];
        const uint32_t* packed = reinterpret_cast<const uint32_t*>(smem_areg + mask_base) + lane * (P / 4);
        uint32_t* mout = p.masks + (size_t)t * p.m.nmax * p.R + r0 + lane;
        for (int kd = wave; kd * 4 < n; kd += kRegThreads / 64) {
            uint32_t m4[4] = {0, 0, 0, 0};
            for (int g = 0; g < NG; ++g) {
                const uint32_t w = packed[g * (kRegCols * P / 4) + kd];
#pragma unroll
                for (int b = 0; b < 4; ++b) m4[b] |= ((w >> (8 * b)) & 0xFFu) << (8 * g);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (kd * 4 + b < n) mout[(size_t)(kd * 4 + b) * p.R] = m4[b];
        }
    }
    if constexpr (ENERGY) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int s = esum[u] + __shfl_xor(esum[u], 32, 64);
            const int col = (u < 2 ? u : bx) * 32 + l31;
            if (h == 0 && wave < KT && (u < 2 || extra)) atomicAdd(&p.out[r0 + col], s);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Core masks: masks[t][k][r] bit j = state byte of (slice t, core j, qubit k) of replica r.
// Workgroup = (64 replicas, slice, 64 qubits): the 64-byte runs of every (replica, core) go through LDS
// (pitch per replica C * 64 + 4: the lanes of a wave, one replica each, hit different banks) and every
// thread assembles 16 masks of its replica; the masks of one qubit leave as 256 contiguous bytes.
// ------------------------------------------------------------------------------------------------

constexpr int kBitsK = 64;
constexpr int kBitsR = 64;

__global__ __launch_bounds__(256) void k_qubo_bits(KronArgs p) {
    extern __shared__ __align__(16) uint8_t lds[];
    const int t = blockIdx.y, k0 = blockIdx.z * kBitsK, r0 = blockIdx.x * kBitsR;
    const int n = p.m.n[t], P = p.m.pitch[t], C = p.m.C;
    if (k0 >= n) return;
    const int rpitch = C * kBitsK + 4;
    const int64_t vslice = p.m.off[t];
    const int pieces = kBitsR * C * (kBitsK / 16);
    for (int q = threadIdx.x; q < pieces; q += 256) {
        const int piece = q & 3, j = (q >> 2) % C, r = (q >> 2) / C;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k0 + piece * 16 < P)
            v = *reinterpret_cast<const uint4*>(p.Xt + (size_t)(r0 + r) * p.m.Npad + vslice + (int64_t)j * P + k0 + piece * 16);
        uint32_t* d = reinterpret_cast<uint32_t*>(lds + r * rpitch + j * kBitsK + piece * 16);
        d[0] = v.x;
        d[1] = v.y;
        d[2] = v.z;
        d[3] = v.w;
    }
    __syncthreads();
    const int r = threadIdx.x & 63, kq = threadIdx.x >> 6;  // 16 qubits per thread
    uint32_t mk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mk[i] = 0;
    for (int j = 0; j < C; ++j) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(lds + r * rpitch + j * kBitsK + kq * 16);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t x = src[w];
#pragma unroll
            for (int i = 0; i < 4; ++i) mk[4 * w + i] |= ((x >> (8 * i)) & 1u) << j;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = k0 + kq * 16 + i;
        if (k < n) p.masks[((size_t)t * p.m.nmax + k) * p.R + r0 + r] = mk[i];
    }
}

// sum over the set bits j of a, l of b of h[j * C + l]
__device__ inline int hop_pairs(uint32_t a, uint32_t b, const int32_t* h, int C) {
    int s = 0;
    while (a) {
        const int j = __builtin_ctz(a);
        a &= a - 1;
        uint32_t bb = b;
        while (bb) {
            s += h[j * C + __builtin_ctz(bb)];
            bb &= bb - 1;
        }
    }
    return s;
}

// The terms of sum_v x_v F_v that the `same` matrix does not hold, from the masks: a qubit on p cores adds
// 8 p (p - 1) (same qubit, two cores), a qubit present in the next slice as well adds twice (once from each
// end) 10 * hops over all pairs (core now, core then).  Thread = replica; block y = slice.
constexpr int kExtraK = 32;  // qubits per block

__global__ __launch_bounds__(256) void k_qubo_extra_energy(KronArgs p) {
    __shared__ int32_t h[32 * 32];
    const int t = blockIdx.y, C = p.m.C;
    const int n = p.m.n[t];
    const int kb = blockIdx.z * kExtraK;
    if (kb >= n) return;
    for (int i = threadIdx.x; i < C * C; i += 256) h[i] = p.m.hops10[i];
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= p.R) return;
    const uint32_t* mk = p.masks + (size_t)t * p.m.nmax * p.R + r;
    const uint32_t* mk_next = mk + (size_t)p.m.nmax * p.R;
    const int32_t* nxt = p.m.next + (size_t)t * p.m.nmax;
    const bool has_next = t + 1 < p.m.T;
    int e = 0;
    for (int k0 = kb; k0 < min(n, kb + kExtraK); k0 += 8) {
        uint32_t m[8], m2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // eight qubits' masks (and their next-slice masks) in flight
            const int k = min(k0 + i, n - 1);
            const int k2 = has_next ? nxt[k] : -1;
            m[i] = k0 + i < n ? mk[(size_t)k * p.R] : 0u;
            m2[i] = k2 >= 0 ? mk_next[(size_t)k2 * p.R] : 0u;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pc = __popc(m[i]);
            e += 8 * pc * (pc - 1);
            if (m[i] && m2[i]) e += 2 * hop_pairs(m[i], m2[i], h, C);
        }
    }
    atomicAdd(&p.out[r], e);
}

// The same terms of the local fields: F[(t, j, k)] += 8 (p - bit_j) + sum_l hops10[j][l] bit_l(next) +
// sum_l hops10[l][j] bit_l(previous).  Thread = replica; block y = slice, z = 16 qubits.
__global__ __launch_bounds__(256) void k_qubo_extra_fields(KronArgs p) {
    __shared__ int32_t h[32 * 32];
    const int t = blockIdx.y, C = p.m.C;
    for (int i = threadIdx.x; i < C * C; i += 256) h[i] = p.m.hops10[i];
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int n = p.m.n[t], P = p.m.pitch[t];
    if (r >= p.R) return;
    const uint32_t* mk = p.masks + (size_t)t * p.m.nmax * p.R + r;
    for (int k = blockIdx.z * 16; k < min(n, (int)blockIdx.z * 16 + 16); ++k) {
        const uint32_t m = mk[(size_t)k * p.R];
        const int pc = __popc(m);
        const int k2 = t + 1 < p.m.T ? p.m.next[(size_t)t * p.m.nmax + k] : -1;
        const int k0 = t > 0 ? p.m.prev[(size_t)t * p.m.nmax + k] : -1;
        const uint32_t mn = k2 >= 0 ? mk[((size_t)p.m.nmax + k2) * p.R] : 0u;
        const uint32_t mp = k0 >= 0 ? *(mk - (size_t)p.m.nmax * p.R + (size_t)k0 * p.R) : 0u;
        for (int j = 0; j < C; ++j) {
            int v = 8 * (pc - (int)((m >> j) & 1u));
            for (uint32_t b = mn; b; b &= b - 1) v += h[j * C + __builtin_ctz(b)];
            for (uint32_t b = mp; b; b &= b - 1) v += h[__builtin_ctz(b) * C + j];
            p.out[(size_t)(p.m.off[t] + (int64_t)j * P + k) * p.R + r] += v;
        }
    }
}

// energy2[r] -> constant + energy2[r] / 2
__global__ void k_qubo_finish(int32_t* e, int32_t R, int32_t constant) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) e[r] = constant + e[r] / 2;
}

// ------------------------------------------------------------------------------------------------
// Simulated annealing, one wave per replica.
//
// Lane (s, j) owns core j of one timeslice; 64 / C slices of equal parity are updated at once
// (slices two apart never share a term).  Inside a slice the lanes walk a Latin square: in step k lane
// j proposes qubit i = m*C + (k - j mod C), so the proposals of one step touch distinct qubits and
// distinct cores and therefore share no term either — simultaneous Metropolis updates of them equal
// sequential ones.  Per replica the wave keeps, in LDS, one core bit-mask per (slice, active qubit);
// core loads and slack bits live in registers while their lane owns the core.
// ------------------------------------------------------------------------------------------------

constexpr int kHopPlanes = 5;  // hop counts up to 31

struct SaArgs {
    QpQuboStruct m;
    int8_t* Xt;
    int32_t* dE;
    int32_t R, sweeps, flags;
    float beta_hot, beta_ratio_log;  // beta(s) = hot * exp(ratio_log * s / (sweeps - 1))
    uint64_t seed;
};

__device__ inline uint32_t rng_next(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}

__device__ inline uint32_t mix32(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z ^ (z >> 32)) | 1u;
}

// beta2 = -beta * log2(e): accept with probability exp(-beta dE) = 2^(beta2 dE)
__device__ inline bool metropolis(int dE, float beta2, uint32_t& rng) {
    if (dE <= 0) return true;
    const float u = (float)(rng_next(rng) >> 8) * (1.0f / 16777216.0f);
    return u < __builtin_amdgcn_exp2f(beta2 * (float)dE);
}

__global__ __launch_bounds__(256) void k_qubo_anneal(SaArgs p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const QpQuboStruct& m = p.m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * (blockDim.x >> 6) + wave;
    const int T = m.T, C = m.C, cap = m.cap, nmax = m.nmax;
    // per-wave LDS: mask[T][nmax] u32 + one word that stays zero ("no such qubit"), load[T][C] i32,
    // slack[T][C] u32
    const size_t per_wave = ((size_t)T * nmax + 1 + 2 * (size_t)T * C) * 4;
    const size_t hop_bytes = ((size_t)C * C + 1 + 15) & ~(size_t)15;
    uint8_t* hop = smem;  // hop counts, shared by the block's replicas; hop[C*C] = 0 for "no bit set"
    for (int k = threadIdx.x; k <= C * C; k += blockDim.x) hop[k] = k < C * C ? m.hops[k] : 0;
    __syncthreads();
    uint32_t* mask = reinterpret_cast<uint32_t*>(smem + hop_bytes + wave * per_wave);
    int32_t* load = reinterpret_cast<int32_t*>(mask + (size_t)T * nmax + 1);
    uint32_t* slack = reinterpret_cast<uint32_t*>(load + (size_t)T * C);
    if (r >= p.R) return;  // after the block-wide barrier above
    int8_t* X = p.Xt + (size_t)r * m.Npad;
    uint32_t rng = mix32(p.seed * 0x100000001B3ull + ((uint64_t)r << 8) + lane);
    const uint32_t capmask = cap >= 32 ? 0xFFFFFFFFu : ((1u << cap) - 1u);
    const uint32_t coremask = C >= 32 ? 0xFFFFFFFFu : ((1u << C) - 1u);

    // ---- initial state
    if (lane == 0) mask[(size_t)T * nmax] = 0;
    for (int t = 0; t < T; ++t) {
        const int n = m.n[t];
        const int W = m.pitch[t];  // state bytes per core: the n qubit bits, the cap slack bits, padding
        const int8_t* Xs = X + m.off[t];
        for (int i = lane; i < nmax; i += 64) {
            uint32_t bits = 0;
            if (i < n) {
                if (p.flags & QP_QUBO_INIT_GIVEN) {
                    for (int j = 0; j < C; ++j) bits |= (uint32_t)(Xs[(size_t)j * W + i] & 1) << j;
                } else {
                    bits = rng_next(rng) & coremask;
                }
            }
            mask[t * nmax + i] = bits;
        }
        for (int j = lane; j < C; j += 64) {
            uint32_t y = 0;
            if (p.flags & QP_QUBO_INIT_GIVEN) {
                for (int q = 0; q < cap; ++q) y |= (uint32_t)(Xs[(size_t)j * W + n + q] & 1) << q;
            } else {
                y = rng_next(rng) & capmask;
            }
            slack[t * C + j] = y;
        }
    }
    for (int t = 0; t < T; ++t) {
        const int n = m.n[t];
        for (int j = lane; j < C; j += 64) {
            int L = -__popc(slack[t * C + j]);
            for (int i = 0; i < n; ++i) L += (mask[t * nmax + i] >> j) & 1u;
            load[t * C + j] = L;
        }
    }

    // ---- sweeps
    const int S = C <= 64 ? max(1, 64 / C) : 1;  // slices updated side by side
    const int ls = lane / C, lj = lane - ls * C;
    const bool lane_used = ls < S;
    // bit planes of the lane's row and column of the hop table: bit l of prow[k] = bit k of hop[lj][l], of pcol[k] = bit k
    // of hop[l][lj]; hop counts between at most 32 connected cores are at most 31
    uint32_t prow[kHopPlanes], pcol[kHopPlanes];
    int nplanes = 1;
    {
        uint32_t top = 0;
#pragma unroll
        for (int kpl = 0; kpl < kHopPlanes; ++kpl) prow[kpl] = pcol[kpl] = 0;
        for (int l = 0; l < C; ++l) {
            const uint32_t hr = lane_used ? hop[lj * C + l] : 0u, hc = lane_used ? hop[l * C + lj] : 0u;
            top |= hr | hc;
#pragma unroll
            for (int kpl = 0; kpl < kHopPlanes; ++kpl) {
                prow[kpl] |= ((hr >> kpl) & 1u) << l;
                pcol[kpl] |= ((hc >> kpl) & 1u) << l;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) top |= __shfl_xor(top, o, 64);
        while (nplanes < kHopPlanes && (top >> nplanes)) ++nplanes;
        nplanes = __builtin_amdgcn_readfirstlane(nplanes);
    }
    int dEsum = 0;
    for (int sweep = 0; sweep < p.sweeps; ++sweep) {
        const float beta = -1.44269504088896f * p.beta_hot *
                           __expf(p.beta_ratio_log * (p.sweeps > 1 ? (float)sweep / (float)(p.sweeps - 1) : 0.0f));  // times -log2(e)
        for (int par = 0; par < 2; ++par) {
            const int nsl = (T - par + 1) / 2;  // slices of this parity
            for (int b0 = 0; b0 < nsl; b0 += S) {
                const int k = b0 + ls;
                const int t = par + 2 * k;
                const bool on = lane_used && k < nsl;
                const int n = on ? m.n[t] : 0;
                int nb = n;  // widest slice of the batch
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) nb = max(nb, __shfl_xor(nb, o, 64));
                int L = on ? load[t * C + lj] : 0;
                uint32_t y = on ? slack[t * C + lj] : 0u;
                uint32_t* mk = mask + (size_t)(on ? t : 0) * nmax;
                const size_t tb = (size_t)(on ? t : 0) * nmax;
                const uint64_t* infp = m.info + tb;
                const int blocks = (nb + C - 1) / C;
                // LDS address of word x of this replica's masks = mbase + 4 x: one v_mad_u32_u16 from either half of a
                // packed pair of word indices
                typedef __attribute__((address_space(3))) uint32_t lds_word;
                const uint32_t mbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)mask;
                auto word_lo = [&](uint32_t packed) {
                    uint32_t a;
                    asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[0,0,0,0]" : "=v"(a) : "v"(packed), "v"(mbase));
                    return *(lds_word*)(uintptr_t)a;
                };
                auto word_hi = [&](uint32_t packed) {
                    uint32_t a;
                    asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a) : "v"(packed), "v"(mbase));
                    return *(lds_word*)(uintptr_t)a;
                };
                for (int mb = 0; mb < blocks; ++mb) {
                    // the lane walks the block's qubits in turn, starting lj short of its end: i = mb C + (step - lj mod C)
                    const int lo = mb * C, hi = lo + C;
                    int i = lj == 0 ? lo : hi - lj;
                    for (int step = 0; step < C; ++step) {
                        if (on && i < n) {
                            const uint32_t w = mk[i];
                            // packed per-qubit record: degree | word of the first partner | word of the same
                            // qubit in slice t+1 | in t-1 (words index this replica's mask array; the word
                            // after the last slice is always zero and stands for "none")
                            const uint64_t inf = infp[i];
                            const uint32_t ilo = (uint32_t)inf, ihi = (uint32_t)(inf >> 32);
                            const int dg = (int)(ilo & 0xFFu);
                            // the three words this proposal looks at besides its own: the first partner's, the same
                            // qubit's in the two neighbouring slices ("none" is a word that stays zero): read together
                            const uint32_t pw = word_hi(ilo);
                            const uint32_t o1 = word_lo(ihi), o2 = word_hi(ihi);
                            int lap;
                            if (dg <= 1) {
                                lap = dg - 2 * (int)((uint32_t)dg & (pw >> lj));
                            } else {
                                lap = dg;
                                for (int e = m.nbr_ptr[tb + i]; e < m.nbr_ptr[tb + i + 1]; ++e)
                                    lap -= 2 * (int)((mk[m.nbr[e]] >> lj) & 1u);
                            }
                            // hops to wherever the same qubit sits in the neighbouring slices: sum over the set bits l of
                            // hop[lj][l] (next slice) and hop[l][lj] (previous), from the bit planes of the lane's row and
                            // column of the hop table: bit k of the sum's terms is a population count
                            int mv = 0;
#pragma unroll
                            for (int kpl = 0; kpl < kHopPlanes; ++kpl)
                                if (kpl < nplanes)
                                    mv += (__popc(o1 & prow[kpl]) + __popc(o2 & pcol[kpl])) << kpl;
                            const int neg = __builtin_amdgcn_sbfe((int)w, (uint32_t)lj, 1u);  // -1 if the bit is set (the flip clears it), else 0
                            const int Ai = __popc(w);
                            // 4 d (2 (Ai - 1) + d) + 4 d (2 L + d) + 10 d (lap + mv) with d = +-1 (d d = 1):
                            // 8 + d (8 (Ai - 1 + L) + 10 (lap + mv)); the negation as xor and subtract
                            const int sl = ((Ai - 1 + L) << 3) + __mul24(10, lap + mv);
                            const int dE = ((sl ^ neg) - neg) + 8;
                            if (metropolis(dE, beta, rng)) {
                                mk[i] = w ^ (1u << lj);
                                L += 1 + 2 * neg;
                                dEsum += dE;
                            }
                        }
                        ++i;
                        if (i == hi) i = lo;
                    }
                }
                if (on) {
                    for (int q = 0; q < cap; ++q) {
                        const int bit = (y >> q) & 1u;
                        const int d = 1 - 2 * bit;
                        const int dE = 4 - (bit ? -8 * L : 8 * L);  // 4 d (-2 L + d)
                        if (metropolis(dE, beta, rng)) {
                            y ^= 1u << q;
                            L -= d;
                            dEsum += dE;
                        }
                    }
                    load[t * C + lj] = L;
                    slack[t * C + lj] = y;
                }
            }
        }
    }

    // ---- write the state back (padding bytes are zeroed) and the summed energy change
    for (int t = 0; t < T; ++t) {
        const int n = m.n[t];
        const int W = m.pitch[t];
        int8_t* Xs = X + m.off[t];
        const int64_t total = m.off[t + 1] - m.off[t];
        for (int64_t v = lane; v < total; v += 64) {
            int8_t val = 0;
            if (v < (int64_t)C * W) {
                const int j = (int)(v / W), kx = (int)(v - (int64_t)j * W);
                if (kx < n) val = (int8_t)((mask[t * nmax + kx] >> j) & 1u);
                else if (kx < n + cap) val = (int8_t)((slack[t * C + j] >> (kx - n)) & 1u);
            }
            Xs[v] = val;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dEsum += __shfl_xor(dEsum, o, 64);
    if (lane == 0) p.dE[r] = dEsum;
}

int check_kron(const char* fn, const QpQuboKron* m, const int8_t* Xt, int32_t R, const void* out, const void* ws,
               size_t ws_bytes) {
    if (!m || !Xt || !out) return qp::fail(QP_EINVAL, "%s: null argument", fn);
    if (m->T <= 0 || m->Npad <= 0 || m->Npad % kTile) return qp::fail(QP_EINVAL, "%s: Npad must be a positive multiple of 128", fn);
    if (m->C <= 0 || m->C > 32 || m->cap <= 0 || m->cap > 32 || m->nmax <= 0) return qp::fail(QP_EINVAL, "%s: needs 1..32 cores and capacity 1..32", fn);
    if (R <= 0 || R % kTile) return qp::fail(QP_EINVAL, "%s: R must be a positive multiple of 128", fn);
    if (!m->n || !m->pitch || !m->off || !m->same || !m->same_off || !m->tile_slice || !m->tile_first || !m->linear ||
        !m->next || !m->prev || !m->hops10)
        return qp::fail(QP_EINVAL, "%s: model tables missing", fn);
    const size_t need = (size_t)m->T * m->nmax * R * sizeof(uint32_t);
    if (!ws || ws_bytes < need) return qp::fail(QP_ENOSPC, "%s: workspace needs %zu bytes", fn, need);
    if (reinterpret_cast<uintptr_t>(ws) % 16 || reinterpret_cast<uintptr_t>(Xt) % 16) return qp::fail(QP_EINVAL, "%s: states and workspace must be 16-byte aligned", fn);
    return QP_OK;
}

KronArgs kron_args(const QpQuboKron* m, const int8_t* Xt, int32_t R, int32_t* out, void* ws) {
    KronArgs a{};
    a.m = *m;
    a.Xt = Xt;
    a.out = out;
    a.masks = static_cast<uint32_t*>(ws);
    a.R = R;
    return a;
}

int launch_bits(const KronArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)kBitsR * (a.m.C * kBitsK + 4);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qubo_bits), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_qubo_bits<<<dim3(a.R / kBitsR, a.m.T, (a.m.nmax + kBitsK - 1) / kBitsK), 256, lds, stream>>>(a);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

// The contraction: `same` in registers while the widest slice's pitch allows it (nmax + cap rounded up to 32, at
// most 384), else a workgroup per 128-row tile with both operands staged through LDS.
template <bool ENERGY>
int launch_kron(const KronArgs& a, hipStream_t stream) {
    const int P = (a.m.nmax + a.m.cap + 31) / 32 * 32;
    if (P <= 32 * kRegK) {  // also writes the core masks
        auto kern = k_qubo_kron_areg<ENERGY>;
        const size_t lds = (size_t)2 * kRegCols * (P + 16) + (size_t)2 * P * sizeof(int32_t) + (size_t)kRegThreads * 16 +
                           (size_t)((a.m.C + 7) / 8) * kRegCols * P;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<dim3(a.R / kRegCols, a.m.T), kRegThreads, lds, stream>>>(a);
    } else {
        if (int rc = launch_bits(a, stream)) return rc;
        if (a.R % 256 == 0) k_qubo_kron<ENERGY, 4><<<dim3(a.R / 256, a.m.n_tiles), 256, 0, stream>>>(a);
        else k_qubo_kron<ENERGY, 2><<<dim3(a.R / 128, a.m.n_tiles), 256, 0, stream>>>(a);
    }
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

}  // namespace

extern "C" size_t qp_qubo_workspace(const QpQuboKron* m, int32_t R) {
    if (!m || R <= 0 || m->T <= 0 || m->nmax <= 0) return 16;
    return (size_t)m->T * m->nmax * R * sizeof(uint32_t);
}

extern "C" int qp_qubo_fields_i32(const QpQuboKron* m, const int8_t* Xt, int32_t R, int32_t* F, void* ws, size_t ws_bytes,
                                  void* stream_) {
    if (int rc = check_kron("qp_qubo_fields_i32", m, Xt, R, F, ws, ws_bytes)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const KronArgs a = kron_args(m, Xt, R, F, ws);
    HIP_TRY(hipMemsetAsync(F, 0, (size_t)m->Npad * R * sizeof(int32_t), stream));  // padding rows are never written
    if (int rc = launch_kron<false>(a, stream)) return rc;  // (and the core masks)
    k_qubo_extra_fields<<<dim3((R + 255) / 256, m->T, (m->nmax + 15) / 16), 256, 0, stream>>>(a);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

extern "C" int qp_qubo_energy_i32(const QpQuboKron* m, const int8_t* Xt, int32_t R, int32_t* energy, void* ws,
                                  size_t ws_bytes, void* stream_) {
    if (int rc = check_kron("qp_qubo_energy_i32", m, Xt, R, energy, ws, ws_bytes)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const KronArgs a = kron_args(m, Xt, R, energy, ws);
    HIP_TRY(hipMemsetAsync(energy, 0, (size_t)R * sizeof(int32_t), stream));
    if (int rc = launch_kron<true>(a, stream)) return rc;  // (and the core masks)
    k_qubo_extra_energy<<<dim3((R + 255) / 256, m->T, (m->nmax + kExtraK - 1) / kExtraK), 256, 0, stream>>>(a);
    HIP_TRY(hipGetLastError());
    k_qubo_finish<<<(R + 255) / 256, 256, 0, stream>>>(energy, R, m->constant);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

extern "C" int qp_qubo_anneal(const QpQuboStruct* m, int32_t R, int32_t sweeps, double beta_hot, double beta_cold,
                              uint64_t seed, int8_t* Xt, int32_t* dE_out, int32_t flags, void* stream_) {
    if (!m || !Xt || !dE_out) return qp::fail(QP_EINVAL, "qp_qubo_anneal: null argument");
    if (m->T <= 0 || m->C <= 0 || m->C > 32 || m->cap <= 0 || m->cap > 32 || m->nmax <= 0)
        return qp::fail(QP_EINVAL, "qp_qubo_anneal: needs 1..32 cores and capacity 1..32");
    if (R <= 0 || sweeps < 0 || !(beta_hot > 0) || !(beta_cold > 0)) return qp::fail(QP_EINVAL, "qp_qubo_anneal: bad R/sweeps/beta");
    if (!m->n || !m->off || !m->info || !m->nbr_ptr || !m->hops || !m->pitch)
        return qp::fail(QP_EINVAL, "qp_qubo_anneal: model tables missing");
    if ((int64_t)m->T * m->nmax > 65534)
        return qp::fail(QP_EINVAL, "qp_qubo_anneal: T * nmax must stay below 65535 (16-bit word indices in info)");
    const size_t per_wave = ((size_t)m->T * m->nmax + 1 + 2 * (size_t)m->T * m->C) * 4;
    int waves = 4;
    while (waves > 1 && per_wave * waves > 64 * 1024) waves >>= 1;
    if (per_wave * waves > 160 * 1024) return qp::fail(QP_EINVAL, "qp_qubo_anneal: model too large for LDS (%zu bytes per replica)", per_wave);
    SaArgs a{};
    a.m = *m;
    a.Xt = Xt;
    a.dE = dE_out;
    a.R = R;
    a.sweeps = sweeps;
    a.flags = flags;
    a.beta_hot = (float)beta_hot;
    a.beta_ratio_log = (float)std::log(beta_cold / beta_hot);
    a.seed = seed;
    const size_t lds = per_wave * waves + (((size_t)m->C * m->C + 1 + 15) & ~(size_t)15);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qubo_anneal), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_qubo_anneal<<<(R + waves - 1) / waves, waves * 64, lds, static_cast<hipStream_t>(stream_)>>>(a);
    HIP_TRY(hipGetLastError());
    return QP_OK;
}

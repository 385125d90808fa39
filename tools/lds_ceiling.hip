// What the LDS charges for the counter updates of the pair-count cost kernel (k_cost_v3_pairs), measured on its own:
// the same launch shape (one 1024-thread workgroup per CU, 128 KB of uint32 counters + a 1 KB strip per wave,
// rows of 1 KB streamed 16 bytes per lane with four rows in flight) doing nothing but a chosen mix of LDS
// operations per row on addresses that come from the streamed bytes.
//
//   mode 0  stream only (rows are read and folded into a checksum)
//   mode 1  16 ds_add_u32 per row: the movement counts of 1024 qubits (two pairs per register, real key arithmetic)
//   mode 2  20 ds_add_u32 per row: movement + the counts of 256 gates
//   mode 3  mode 2 + the strip write (ds_write_b128) + 8 ds_read_u8 gathers at random strip bytes
//   mode 4  20 ds_add_u32 per row whose 32 lanes of a group each hit their own bank (no conflicts possible)
//   mode 5  20 ds_read_b32 per row at the same random addresses (reads instead of read-modify-writes)
//   mode 6  20 ds_add_u32 per row, all lanes of the wave on one address
//   mode 7  mode 2 + the strip written by four ds_write_addtid_b32 + 5 gathers whose 32 lanes hit 32 banks
//
// Build:  hipcc -O3 --offload-arch=gfx950 tools/lds_ceiling.hip -o tools/bin/lds_ceiling
// Run:    tools/bin/lds_ceiling [rows (default 4096000)] [launches per mode (default 10)]
// Prints one line per mode: ms per launch, rows/s, LDS cycles per row and CU at 2.4 GHz.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int kWave = 64, kThreads = 1024, kWaves = kThreads / kWave, kAhead = 4;
constexpr int kSlots = 128 * 256;
constexpr int kLdq = 1024;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ inline uint4 load_row16(const uint8_t* ptr) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ inline uint32_t fold_keys(uint32_t k, uint32_t s) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    const u16x2 m = __builtin_elementwise_max(__builtin_bit_cast(u16x2, k), __builtin_bit_cast(u16x2, s));
    const i16x2 t = __builtin_bit_cast(i16x2, m) >> 15;
    return ~(__builtin_bit_cast(uint32_t, m) ^ __builtin_bit_cast(uint32_t, t));
}

template <int MODE>
__device__ inline void touch(uint32_t x, uint32_t base, int lane, uint32_t& sink) {
    uint32_t a0, a1;
    if constexpr (MODE == 4) {  // bank = lane & 31 whatever the key
        x = (x & 0xFFE0FFE0u) | (uint32_t)(lane & 31) * 0x00010001u;
    }
    if constexpr (MODE == 6) x = __builtin_amdgcn_readfirstlane(x);
    asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[0,0,0,0]" : "=v"(a0) : "v"(x), "s"(base));
    asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(x), "s"(base));
    if constexpr (MODE == 5) {
        sink += *(lds_u32*)(uintptr_t)a0;
        sink += *(lds_u32*)(uintptr_t)a1;
    } else {
        __hip_atomic_fetch_add((lds_u32*)(uintptr_t)a0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add((lds_u32*)(uintptr_t)a1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <int MODE>
__device__ inline void moves4(uint32_t pf, uint32_t cf, uint32_t base, int lane, uint32_t& sink) {
    touch<MODE>(fold_keys(__builtin_amdgcn_perm(pf, cf, 0x05010400u), __builtin_amdgcn_perm(cf, pf, 0x05010400u)), base, lane, sink);
    touch<MODE>(fold_keys(__builtin_amdgcn_perm(pf, cf, 0x07030602u), __builtin_amdgcn_perm(cf, pf, 0x07030602u)), base, lane, sink);
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_ceiling(const uint8_t* A, int64_t rows, int rows_per_item, uint32_t* out) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);
    uint8_t* strip = smem + (size_t)kSlots * 4 + (size_t)wave * kLdq;
    const uint32_t cbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)smem - 0x20000u);
    const uint32_t sbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8*)strip);
    for (int k = threadIdx.x; k < kSlots; k += kThreads) cnt[k] = 0;
    __syncthreads();
    uint32_t sink = 0;
    const int64_t nitems = (rows + rows_per_item - 1) / rows_per_item;
    for (int64_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int64_t r0 = it * rows_per_item, r1 = min(rows, r0 + rows_per_item);
        const int64_t w0 = r0 + (r1 - r0) * wave / kWaves, w1 = r0 + (r1 - r0) * (wave + 1) / kWaves;
        if (w0 < w1) {
            auto fetch = [&](int64_t row) { return load_row16(A + (size_t)min(row, w1 - 1) * kLdq + lane * 16); };
            uint4 ring[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) ring[k] = fetch(w0 + k);
            uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            auto do_row = [&](int64_t row, uint4& slot, bool refill) {
                const uint4 cur = slot;
                if (refill) slot = fetch(row + kAhead);
                if constexpr (MODE == 0) {
                    sink += cur.x ^ cur.y ^ cur.z ^ cur.w;
                    return;
                }
                uint32_t g[8];
                if constexpr (MODE == 3) {
                    *reinterpret_cast<uint4*>(strip + lane * 16) = cur;
                    // eight gathers at bytes the row itself names (10 bits each)
                    const uint32_t idx[8] = {cur.x & 1023u, (cur.x >> 10) & 1023u, cur.y & 1023u, (cur.y >> 10) & 1023u,
                                             cur.z & 1023u, (cur.z >> 10) & 1023u, cur.w & 1023u, (cur.w >> 10) & 1023u};
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] = strip[idx[j]];
                }
                if constexpr (MODE == 7) {
                    // the strip as four planes of 64 dwords: plane j, dword lane; M0 holds the plane's LDS address
                    const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t m0v = sbase + j * 256;
                        asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1" ::"s"(m0v), "v"(w[j]) : "m0", "memory");
                    }
                    // five gathers, bank = lane & 31 (plane and byte from the data)
                    const uint32_t sel[5] = {cur.x, cur.x >> 8, cur.y, cur.z, cur.w};
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const uint32_t a = ((sel[j] >> 2) & 3u) * 256u + ((sel[j] >> 4) & 1u) * 128u + (uint32_t)(lane & 31) * 4u + (sel[j] & 3u);
                        g[j] = strip[a];
                    }
                    g[5] = g[6] = g[7] = 0;
                }
                if constexpr (MODE == 1 || MODE >= 2) {
                    moves4<MODE>(p0, cur.x, cbase, lane, sink);
                    moves4<MODE>(p1, cur.y, cbase, lane, sink);
                    moves4<MODE>(p2, cur.z, cbase, lane, sink);
                    moves4<MODE>(p3, cur.w, cbase, lane, sink);
                }
                if constexpr (MODE >= 2) {
                    uint32_t u2 = __builtin_amdgcn_perm(cur.y, cur.x, 0x0c060c01u), v2 = __builtin_amdgcn_perm(cur.w, cur.z, 0x0c050c02u);
                    uint32_t u3 = __builtin_amdgcn_perm(cur.z, cur.x, 0x0c070c03u), v3 = __builtin_amdgcn_perm(cur.w, cur.y, 0x0c040c00u);
                    if constexpr (MODE == 3 || MODE == 7) {  // make the gate counts wait for the gathers, as in the kernel
                        u2 ^= (g[0] ^ g[1] ^ g[2] ^ g[3]) & 1u;
                        u3 ^= (g[4] ^ g[5] ^ g[6] ^ g[7]) & 1u;
                    }
                    touch<MODE>(fold_keys(u2 << 8 | v2, v2 << 8 | u2), cbase, lane, sink);
                    touch<MODE>(fold_keys(u3 << 8 | v3, v3 << 8 | u3), cbase, lane, sink);
                }
                p0 = cur.x;
                p1 = cur.y;
                p2 = cur.z;
                p3 = cur.w;
            };
            int64_t g0 = w0;
            for (; g0 + kAhead <= w1; g0 += kAhead) {
#pragma unroll
                for (int k = 0; k < kAhead; ++k) do_row(g0 + k, ring[k], true);
            }
#pragma unroll
            for (int k = 0; k < kAhead - 1; ++k)
                if (g0 + k < w1) do_row(g0 + k, ring[k], false);
        }
        __syncthreads();
    }
    // keep everything alive
    uint32_t s = sink;
    for (int k = threadIdx.x; k < kSlots; k += kThreads) s += cnt[k];
    if (s == 0x12345678u) out[blockIdx.x] = s;
}

template <int MODE>
float run(const uint8_t* A, int64_t rows, int reps, uint32_t* out, int cus) {
    const size_t lds = (size_t)kSlots * 4 + (size_t)kWaves * kLdq + 256;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ceiling<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_ceiling<MODE>, dim3(cus), dim3(kThreads), lds, 0, A, rows, 256, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_ceiling<MODE>, dim3(cus), dim3(kThreads), lds, 0, A, rows, 256, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 4096000;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t bytes = (size_t)rows * kLdq;
    uint8_t* A;
    uint32_t* out;
    CHECK(hipMalloc(&A, bytes));
    CHECK(hipMalloc(&out, 4096 * 4));
    {  // random bytes: a xorshift stream per 1 MB block, generated on the host in pieces
        std::vector<uint64_t> buf(1 << 17);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (size_t off = 0; off < bytes; off += buf.size() * 8) {
            for (auto& w : buf) {
                s ^= s << 13;
                s ^= s >> 7;
                s ^= s << 17;
                w = s;
            }
            const size_t n = bytes - off < buf.size() * 8 ? bytes - off : buf.size() * 8;
            CHECK(hipMemcpy(A + off, buf.data(), n, hipMemcpyHostToDevice));
        }
    }
    const char* names[8] = {"stream only", "16 ds_add_u32 (movement)", "20 ds_add_u32 (movement + gates)",
                            "20 ds_add_u32 + ds_write_b128 + 8 random ds_read_u8", "20 ds_add_u32, one bank per lane (conflict-free)",
                            "20 ds_read_b32 at the same random addresses", "20 ds_add_u32, one address per wave",
                            "20 ds_add_u32 + 4 ds_write_addtid_b32 + 5 conflict-free ds_read_u8"};
    float ms[8];
    ms[0] = run<0>(A, rows, reps, out, cus);
    ms[1] = run<1>(A, rows, reps, out, cus);
    ms[2] = run<2>(A, rows, reps, out, cus);
    ms[3] = run<3>(A, rows, reps, out, cus);
    ms[4] = run<4>(A, rows, reps, out, cus);
    ms[5] = run<5>(A, rows, reps, out, cus);
    ms[6] = run<6>(A, rows, reps, out, cus);
    ms[7] = run<7>(A, rows, reps, out, cus);
    printf("%s, %d CUs, %lld rows of %d bytes (%.2f GB), %d launches per mode\n", prop.name, cus, (long long)rows, kLdq, bytes / 1e9, reps);
    for (int m = 0; m < 8; ++m) {
        const double sec = ms[m] * 1e-3;
        printf("mode %d  %-68s %8.3f ms  %6.2f G rows/s  %5.2f TB/s  %6.1f cycles per row and CU at 2.4 GHz  frac of 8 TB/s %.3f\n", m, names[m],
               ms[m], rows / sec / 1e9, bytes / sec / 1e12, sec * 2.4e9 * cus / rows, bytes / sec / 8e12);
    }
    return 0;
}

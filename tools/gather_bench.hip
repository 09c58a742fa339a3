// gather_bench.hip -- what the MI355X memory system gives to the access pattern of an Occ lookup.
//
// Standalone microbenchmark (not part of the library).  A DPP quad (4 lanes) fetches random,
// aligned records from a large HBM buffer with nothing else to do, so the rates printed here are
// the ceiling for any kernel that makes the same accesses:
//   line128   one random 128-B line per quad per load pair      (the block fetch of an Occ lookup)
//   line64    one random 64-B half line per quad                (smaller blocks)
//   dep       8-B entry from a second table, then the 128-B line it selects
//             (directory entry -> block: the two dependent accesses of today's Occ lookup)
// Usage: gather_bench [buffer_GiB=32] [dir_GiB=6] [iters=64] [blocks_per_cu=8]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int UNROLL>
__global__ void __launch_bounds__(256)
line128_kernel(const uint4 *__restrict__ buf, uint64_t nlines, int iters, uint32_t *__restrict__ sink) {
    const uint64_t quad = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint32_t t = threadIdx.x & 3;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 a[UNROLL], b[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t line = mix64(quad * 1315423911ull + (uint64_t)(it * UNROLL + u)) % nlines;
            const uint4 *p = buf + line * 8 + t * 2;
            a[u] = p[0];
            b[u] = p[1];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= a[u].x ^ a[u].w ^ b[u].y ^ b[u].z;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int UNROLL>
__global__ void __launch_bounds__(256)
line64_kernel(const uint4 *__restrict__ buf, uint64_t nhalf, int iters, uint32_t *__restrict__ sink) {
    const uint64_t quad = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint32_t t = threadIdx.x & 3;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 a[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t h = mix64(quad * 1315423911ull + (uint64_t)(it * UNROLL + u)) % nhalf;
            a[u] = buf[h * 4 + t];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= a[u].x ^ a[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// LANES lanes share one random 128-B line; each lane reads 128/LANES bytes as 16-B loads.
template <int LANES>
__global__ void __launch_bounds__(256)
lane_kernel(const uint4 *__restrict__ buf, uint64_t nlines, int iters, uint32_t *__restrict__ sink) {
    constexpr int PIECES = 8 / LANES;
    const uint64_t grp = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint32_t t = threadIdx.x % LANES;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint64_t line = mix64(grp * 1315423911ull + (uint64_t)it) % nlines;
        const uint4 *p = buf + line * 8 + t * PIECES;
        uint4 a[PIECES];
#pragma unroll
        for (int u = 0; u < PIECES; ++u) a[u] = p[u];
#pragma unroll
        for (int u = 0; u < PIECES; ++u) acc ^= a[u].x ^ a[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// dependent pair: dir[random] (8 B) -> line index -> 128-B line; the next address depends on the
// line just read (as the next LF step depends on the rank just computed).
template <int CHAINS>
__global__ void __launch_bounds__(256)
dep_kernel(const uint4 *__restrict__ buf, uint64_t nlines, const uint2 *__restrict__ dir,
           uint64_t ndir, int iters, uint32_t *__restrict__ sink) {
    const uint64_t quad = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint32_t t = threadIdx.x & 3;
    uint64_t state[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) state[c] = mix64(quad * CHAINS + c);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint2 e[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) e[c] = dir[state[c] % ndir];
        uint4 a[CHAINS], b[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            const uint64_t line = (mix64(state[c]) + e[c].x) % nlines;
            const uint4 *p = buf + line * 8 + t * 2;
            a[c] = p[0];
            b[c] = p[1];
        }
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            uint32_t v = a[c].x ^ b[c].w;
            v ^= __shfl_xor(v, 1);
            v ^= __shfl_xor(v, 2);
            state[c] = mix64(state[c] + v);
            acc ^= v;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// The wave-cooperative pattern of search_wave_kernel: every lane wants its own random line; in
// four rounds quad q fetches the line of lane 16r+q (ids by ds_bpermute), parks it in LDS, then
// each lane reads its line back and derives the next address from it (a dependent chain per lane).
template <int SETS>
__global__ void __launch_bounds__(256)
coop_kernel(const uint4 *__restrict__ buf, uint64_t nlines, int iters, uint32_t *__restrict__ sink) {
    __shared__ uint4 stage[4][SETS][64 * 8];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, quad = lane >> 2, t = lane & 3;
    uint64_t state[SETS];
    uint32_t acc = 0;
#pragma unroll
    for (int s = 0; s < SETS; ++s) state[s] = mix64(((uint64_t)blockIdx.x * 256 + threadIdx.x) * SETS + s);
    for (int it = 0; it < iters; ++it) {
        uint4 a[SETS][4], c[SETS][4];
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
            const uint32_t want = (uint32_t)(state[s] % nlines);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t tb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((16 * r + quad) << 2), (int)want);
                const uint4 *bp = buf + (uint64_t)tb * 8 + t * 2;
                a[s][r] = bp[0];
                c[s][r] = bp[1];
            }
        }
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t T = 16 * r + quad, sw = (T ^ (T >> 3)) & 7;
                uint4 *dst = &stage[wave][s][T * 8];
                dst[(2 * t) ^ sw] = a[s][r];
                dst[(2 * t + 1) ^ sw] = c[s][r];
            }
            asm volatile("" ::: "memory");
            const uint32_t sw = (lane ^ (lane >> 3)) & 7;
            const uint4 v = stage[wave][s][lane * 8 + (3 ^ sw)];
            const uint4 w = stage[wave][s][lane * 8 + (6 ^ sw)];
            state[s] = mix64(state[s] + v.x + w.w);
            acc ^= v.y;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void fill_kernel(uint4 *buf, uint64_t n16) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nt = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gid; i < n16; i += nt) {
        const uint32_t h = (uint32_t)mix64(i);
        buf[i] = make_uint4(h, h ^ 1, h ^ 2, h ^ 3);
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

// The search kernel's fetch as it is now: 64 random lines per wave and pass, a full 128-B line per
// octet of lanes, by LDS-DMA (global_load_lds_dwordx4) into an 8 KB stage, all 64 waited for, then
// (as a stand-in for the rank) one LDS read per lane.  WGS workgroups of 4 waves per CU.
typedef __attribute__((address_space(3))) void *gb_lds_ptr;
typedef const __attribute__((address_space(1))) void *gb_glb_ptr;
__global__ void __launch_bounds__(256)
glds_octet_kernel(const uint4 *__restrict__ buf, uint64_t nlines, int iters, uint32_t *__restrict__ sink) {
    __shared__ uint4 stage[4][512];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(gb_lds_ptr)stage[wave]);
    const uint64_t me = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const char *base = reinterpret_cast<const char *>(buf);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t mine = (uint32_t)(mix64(me * 1315423911ull + (uint64_t)it) % nlines);  // the line this lane wants
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t line = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & ~7u) + k) << 2), (int)mine);
            const char *src = base + (uint64_t)line * 128u + (((lane & 7u) ^ (uint32_t)k) << 4);
            __builtin_amdgcn_global_load_lds((gb_glb_ptr)src, (gb_lds_ptr)(uintptr_t)(lds0 + k * 1024u), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        acc ^= reinterpret_cast<const volatile uint32_t *>(stage[wave])[(lane & 7u) * 256u + (lane >> 3) * 32u + 1u];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const double buf_gib = argc > 1 ? atof(argv[1]) : 32.0;
    const double dir_gib = argc > 2 ? atof(argv[2]) : 6.0;
    const int iters = argc > 3 ? atoi(argv[3]) : 64;
    const int bpc = argc > 4 ? atoi(argv[4]) : 8;
    const uint64_t nlines = (uint64_t)(buf_gib * (1ull << 30)) / 128;
    uint64_t ndir = (uint64_t)(dir_gib * (1ull << 30)) / 8;
    if (ndir < 1024) ndir = 1024;  // never an empty second table: the dependent pattern indexes it modulo its size
    uint4 *buf;
    uint2 *dir;
    uint32_t *sink;
    CK(hipMalloc(&buf, nlines * 128));
    CK(hipMalloc(&dir, ndir * 8));
    CK(hipMalloc(&sink, 4));
    fill_kernel<<<4096, 256>>>(buf, nlines * 8);
    fill_kernel<<<4096, 256>>>((uint4 *)dir, ndir / 2);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * bpc;
    const uint64_t quads = (uint64_t)grid * 64;
    printf("# buffer %.1f GiB (%llu lines), dir %.1f GiB, %d CUs x %d blocks of 256, iters %d\n", buf_gib,
           (unsigned long long)nlines, dir_gib, prop.multiProcessorCount, bpc, iters);
    printf("%-22s %12s %12s %12s\n", "pattern", "ms", "Gacc/s", "GB/s");
#define RUN128(U)                                                                                   \
    {                                                                                               \
        double ms = time_ms([&] { line128_kernel<U><<<grid, 256>>>(buf, nlines, iters, sink); }, 3); \
        double acc = (double)quads * iters * U;                                                     \
        printf("line128 x%-2d inflight   %12.3f %12.2f %12.1f\n", U, ms, acc / ms / 1e6, acc * 128 / ms / 1e6); \
    }
    RUN128(1) RUN128(2) RUN128(4) RUN128(8)
#define RUN64(U)                                                                                    \
    {                                                                                               \
        double ms = time_ms([&] { line64_kernel<U><<<grid, 256>>>(buf, nlines * 2, iters, sink); }, 3); \
        double acc = (double)quads * iters * U;                                                     \
        printf("line64  x%-2d inflight   %12.3f %12.2f %12.1f\n", U, ms, acc / ms / 1e6, acc * 64 / ms / 1e6); \
    }
    RUN64(1) RUN64(2) RUN64(4) RUN64(8)
#define RUNDEP(CH)                                                                                  \
    {                                                                                               \
        double ms = time_ms([&] { dep_kernel<CH><<<grid, 256>>>(buf, nlines, dir, ndir, iters, sink); }, 3); \
        double acc = (double)quads * iters * CH;                                                    \
        printf("dep dir->line x%-2d      %12.3f %12.2f %12.1f\n", CH, ms, acc / ms / 1e6, acc * 128 / ms / 1e6); \
    }
    RUNDEP(1) RUNDEP(2) RUNDEP(4)
#define RUNLANE(LN)                                                                                 \
    {                                                                                               \
        double ms = time_ms([&] { lane_kernel<LN><<<grid, 256>>>(buf, nlines, iters, sink); }, 3);  \
        double acc = (double)grid * 256 / LN * iters;                                               \
        printf("%d lane(s) per line      %12.3f %12.2f %12.1f\n", LN, ms, acc / ms / 1e6, acc * 128 / ms / 1e6); \
    }
    RUNLANE(1) RUNLANE(2) RUNLANE(4) RUNLANE(8)
#define RUNCOOP(SETS, WGS)                                                                          \
    {                                                                                               \
        const int g = prop.multiProcessorCount * WGS;                                               \
        double ms = time_ms([&] { coop_kernel<SETS><<<g, 256>>>(buf, nlines, iters, sink); }, 3);    \
        double acc = (double)g * 256 * SETS * iters;                                                \
        printf("coop x%d sets, %d WG/CU   %12.3f %12.2f %12.1f\n", SETS, WGS, ms, acc / ms / 1e6, acc * 128 / ms / 1e6); \
    }
    RUNCOOP(1, 2) RUNCOOP(1, 3) RUNCOOP(1, 4) RUNCOOP(1, 5) RUNCOOP(2, 1) RUNCOOP(2, 2) RUNCOOP(2, 3)
#define RUNGLDS(WGS)                                                                                \
    {                                                                                               \
        const int g = prop.multiProcessorCount * WGS;                                               \
        double ms = time_ms([&] { glds_octet_kernel<<<g, 256>>>(buf, nlines, iters, sink); }, 3);    \
        double acc = (double)g * 256 * iters;                                                       \
        printf("glds octet, %d WG/CU     %12.3f %12.2f %12.1f\n", WGS, ms, acc / ms / 1e6, acc * 128 / ms / 1e6); \
    }
    RUNGLDS(2) RUNGLDS(3) RUNGLDS(4) RUNGLDS(5)
    return 0;
}

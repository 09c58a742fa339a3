// index.hip -- builds the HBM index (blocks + directory, block_format.h) from run bytes in HBM.
//
// Role of RLEBWT::initialiseFMIndex (src/bwt/rlebwt.cpp:34-148) in the reference: one pass over
// the runs producing cumulative checkpoints and C[].  Here it is three streaming kernels over
// R run bytes (read twice, 4/3 R + directory written once), HBM-bandwidth bound:
//   1. chunk_totals   per 256-block chunk: symbols and A/C/G/T counts
//   2. scan_chunks    exclusive scan of the chunk totals (one workgroup), totals -> C[]
//   3. write_blocks   per block: absolute checkpoints + the 96 run bytes, quad-interleaved
//   4. fill_dir       per block: directory ids for the windows it starts in, and its own
//                     start offset into the window it interrupts
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "block_format.h"
#include "kernels.h"

namespace rsb {

constexpr int CHUNK = 256;  // blocks per chunk = threads per workgroup

struct blk_tot {
    uint32_t v[5];  // span, A, C, G, T
};

// 96 run bytes of block j into 24 dwords (zero beyond the end of the stream).
__device__ __forceinline__ void load_block_runs(const uint8_t *__restrict__ runs, uint64_t R,
                                                uint64_t j, bool aligned16, uint32_t w[24]) {
    const uint64_t base = j * RSBWT_BLOCK_RUNS;
    if (aligned16 && base + RSBWT_BLOCK_RUNS <= R) {
        const uint4 *p = reinterpret_cast<const uint4 *>(runs + base);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint4 x = p[i];
            w[4 * i] = x.x; w[4 * i + 1] = x.y; w[4 * i + 2] = x.z; w[4 * i + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t a = base + (uint64_t)(4 * i + k);
                if (a < R) x |= (uint32_t)runs[a] << (8 * k);
            }
            w[i] = x;
        }
    }
}

__device__ __forceinline__ blk_tot totals_of(const uint32_t w[24]) {
    blk_tot t = {{0, 0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 24; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t len = __builtin_amdgcn_ubfe(w[i], 8 * k, 5);
            const uint32_t sym = __builtin_amdgcn_ubfe(w[i], 8 * k + 5, 3);
            t.v[0] += len;
            t.v[1] += sym == 1u ? len : 0u;
            t.v[2] += sym == 2u ? len : 0u;
            t.v[3] += sym == 3u ? len : 0u;
            t.v[4] += sym == 4u ? len : 0u;
        }
    }
    return t;
}

__global__ void __launch_bounds__(CHUNK)
chunk_totals_kernel(const uint8_t *__restrict__ runs, uint64_t R, uint64_t nblocks, bool aligned16,
                    uint64_t *__restrict__ chunk_tot) {
    __shared__ uint32_t acc[5];
    if (threadIdx.x < 5) acc[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t j = (uint64_t)blockIdx.x * CHUNK + threadIdx.x;
    blk_tot t = {{0, 0, 0, 0, 0}};
    if (j < nblocks) {
        uint32_t w[24];
        load_block_runs(runs, R, j, aligned16, w);
        t = totals_of(w);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        uint32_t v = t.v[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&acc[i], v);
    }
    __syncthreads();
    if (threadIdx.x < 5) chunk_tot[(uint64_t)blockIdx.x * 5 + threadIdx.x] = acc[threadIdx.x];
}

// In-place exclusive scan of nchunks x 5 u64 by one 1024-thread workgroup; totals -> tot[5].
__global__ void __launch_bounds__(1024)
scan_chunks_kernel(uint64_t *__restrict__ chunk_tot, uint64_t nchunks, uint64_t *__restrict__ tot) {
    __shared__ uint64_t part[1024][5];
    const uint64_t per = (nchunks + 1023) / 1024;
    const uint64_t b = (uint64_t)threadIdx.x * per;
    const uint64_t e = b + per < nchunks ? b + per : nchunks;
    uint64_t s[5] = {0, 0, 0, 0, 0};
    for (uint64_t c = b; c < e; ++c)
        for (int i = 0; i < 5; ++i) s[i] += chunk_tot[c * 5 + i];
    for (int i = 0; i < 5; ++i) part[threadIdx.x][i] = s[i];
    __syncthreads();
    if (threadIdx.x < 5) {  // 5 serial scans of 1024 partials
        uint64_t run = 0;
        for (int k = 0; k < 1024; ++k) {
            const uint64_t v = part[k][threadIdx.x];
            part[k][threadIdx.x] = run;
            run += v;
        }
        tot[threadIdx.x] = run;
    }
    __syncthreads();
    for (int i = 0; i < 5; ++i) s[i] = part[threadIdx.x][i];
    for (uint64_t c = b; c < e; ++c) {
        for (int i = 0; i < 5; ++i) {
            const uint64_t v = chunk_tot[c * 5 + i];
            chunk_tot[c * 5 + i] = s[i];
            s[i] += v;
        }
    }
}

__global__ void __launch_bounds__(CHUNK)
write_blocks_kernel(const uint8_t *__restrict__ runs, uint64_t R, uint64_t nblocks, bool aligned16,
                    const uint64_t *__restrict__ chunk_base, uint4 *__restrict__ blocks,
                    uint64_t *__restrict__ P0arr) {
    __shared__ uint32_t sc[2][CHUNK][5];
    const uint64_t j = (uint64_t)blockIdx.x * CHUNK + threadIdx.x;
    uint32_t w[24];
    blk_tot t = {{0, 0, 0, 0, 0}};
    if (j < nblocks) {
        load_block_runs(runs, R, j, aligned16, w);
        t = totals_of(w);
    }
    // inclusive Hillis-Steele scan of the 256 block totals (u32 is enough inside a chunk)
    int cur = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[0][threadIdx.x][i] = t.v[i];
    __syncthreads();
    for (int off = 1; off < CHUNK; off <<= 1) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            uint32_t v = sc[cur][threadIdx.x][i];
            if ((int)threadIdx.x >= off) v += sc[cur][threadIdx.x - off][i];
            sc[cur ^ 1][threadIdx.x][i] = v;
        }
        cur ^= 1;
        __syncthreads();
    }
    if (j >= nblocks) return;
    uint64_t before[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
        before[i] = chunk_base[(uint64_t)blockIdx.x * 5 + i] + (sc[cur][threadIdx.x][i] - t.v[i]);
    const uint64_t P0 = before[0];
    uint64_t used = R - j * RSBWT_BLOCK_RUNS;
    if (used > RSBWT_BLOCK_RUNS) used = RSBWT_BLOCK_RUNS;
    uint32_t start[4] = {0, 0, 0, 0};  // symbols held by the lanes below lane q
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) sum = __builtin_amdgcn_sad_u8(w[6 * (q - 1) + i] & 0x1F1F1F1Fu, 0u, sum);
        start[q] = start[q - 1] + sum;
    }
    const uint32_t meta[4] = {(uint32_t)(P0 & 0xFFFFFFu),
                              (uint32_t)((P0 >> 24) & 0xFFFFu) | ((uint32_t)used << 16),
                              t.v[0] | (start[1] << 12), start[2] | (start[3] << 12)};
    uint4 *dst = blocks + j * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t word = (before[q + 1] & RSBWT_COUNT_MASK) | ((uint64_t)meta[q] << 40);
        dst[2 * q] = make_uint4((uint32_t)word, (uint32_t)(word >> 32), w[6 * q], w[6 * q + 1]);
        dst[2 * q + 1] = make_uint4(w[6 * q + 2], w[6 * q + 3], w[6 * q + 4], w[6 * q + 5]);
    }
    P0arr[j] = P0;
}

__global__ void __launch_bounds__(256)
fill_dir_kernel(const uint64_t *__restrict__ P0arr, uint64_t nblocks, uint64_t n, uint32_t s,
                uint32_t fields, uint2 *__restrict__ dir) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nblocks) return;
    const uint64_t mask = (1ull << s) - 1;
    const uint64_t P0 = P0arr[j];
    const uint64_t P1 = (j + 1 < nblocks) ? P0arr[j + 1] : (n > P0 ? n : P0 + 1);
    // windows that start inside this block
    for (uint64_t w = (P0 + mask) >> s; (w << s) < P1; ++w) dir[w].x = (uint32_t)j;
    // this block's own start, if it falls strictly inside a window
    const uint32_t off = (uint32_t)(P0 & mask);
    if (j > 0 && off != 0) {
        const uint64_t w = P0 >> s;
        const uint64_t ws = w << s;
        uint64_t jj = j - 1;
        while (P0arr[jj] > ws) --jj;  // P0arr[0] == 0 <= ws
        const uint64_t kth = j - jj;  // this is the kth block start inside window w
        if (kth <= fields) atomicOr(&dir[w].y, off << ((uint32_t)(kth - 1) * s));
    }
}

#define HIP_TRY(x)                     \
    do {                               \
        hipError_t _e = (x);           \
        if (_e != hipSuccess) {        \
            err = _e;                  \
            goto fail;                 \
        }                              \
    } while (0)

hipError_t build_device_index(const void *d_runs, uint64_t num_runs, uint32_t dir_shift,
                              hipStream_t stream, build_result *out, int *range_error) {
    hipError_t err = hipSuccess;
    *range_error = 0;
    const uint8_t *runs = (const uint8_t *)d_runs;
    const uint64_t R = num_runs;
    const uint64_t nblocks = R ? (R + RSBWT_BLOCK_RUNS - 1) / RSBWT_BLOCK_RUNS : 1;
    const uint64_t nchunks = (nblocks + CHUNK - 1) / CHUNK;
    const bool aligned16 = ((uintptr_t)runs & 15u) == 0;
    uint64_t *d_chunk = nullptr, *d_tot = nullptr, *d_P0 = nullptr;
    uint4 *d_blocks = nullptr;
    uint2 *d_dir = nullptr;
    uint64_t tot[5] = {0, 0, 0, 0, 0};
    rsbwt_view v = {};

    if (nblocks >= (1ull << 32) || nchunks >= (1ull << 31)) { *range_error = 1; return hipSuccess; }

    HIP_TRY(hipMalloc(&d_chunk, nchunks * 5 * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&d_tot, 5 * sizeof(uint64_t)));
    hipLaunchKernelGGL(chunk_totals_kernel, dim3((unsigned)nchunks), dim3(CHUNK), 0, stream, runs, R,
                       nblocks, aligned16, d_chunk);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(1), dim3(1024), 0, stream, d_chunk, nchunks, d_tot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));

    {
        const uint64_t n = tot[0];
        if (n >= RSBWT_MAX_SYMBOLS) { *range_error = 1; goto fail; }
        uint32_t s = dir_shift;
        if (s == 0) {
            // about 32 mean-length runs per window: directory ~ 1/4 of the run bytes
            const double L = R ? (double)n / (double)R : 1.0;
            s = RSBWT_MIN_DIR_SHIFT;
            while (s < 12 && (double)(1u << s) * 1.4142 < 32.0 * L) ++s;
        }
        if (s < RSBWT_MIN_DIR_SHIFT) s = RSBWT_MIN_DIR_SHIFT;
        if (s > RSBWT_MAX_DIR_SHIFT) s = RSBWT_MAX_DIR_SHIFT;
        const uint64_t nwin = (n >> s) + 1;

        HIP_TRY(hipMalloc(&d_blocks, nblocks * RSBWT_BLOCK_BYTES));
        HIP_TRY(hipMalloc(&d_dir, nwin * sizeof(uint2)));
        HIP_TRY(hipMalloc(&d_P0, nblocks * sizeof(uint64_t)));
        HIP_TRY(hipMemsetAsync(d_dir, 0, nwin * sizeof(uint2), stream));
        hipLaunchKernelGGL(write_blocks_kernel, dim3((unsigned)nchunks), dim3(CHUNK), 0, stream, runs,
                           R, nblocks, aligned16, d_chunk, d_blocks, d_P0);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(fill_dir_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0,
                           stream, d_P0, nblocks, n, s, 32u / s, d_dir);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));

        v.blocks = d_blocks;
        v.dir = d_dir;
        v.n = n;
        v.nblocks = nblocks;
        v.nwin = nwin;
        v.dir_shift = s;
        v.dir_fields = 32u / s;
        v.ktab = nullptr;
        v.ktab_depth = 0;
        // tot = {n, A, C, G, T}; '$' = n - (A+C+G+T).  C[] as rlebwt.cpp:129-147.
        v.total[0] = n - (tot[1] + tot[2] + tot[3] + tot[4]);
        for (int c = 1; c < 5; ++c) v.total[c] = tot[c];
        v.C[0] = 0;
        for (int c = 1; c < 5; ++c) v.C[c] = v.C[c - 1] + v.total[c - 1];
        out->view = v;
        out->num_runs = R;
        out->hbm_bytes = nblocks * RSBWT_BLOCK_BYTES + nwin * sizeof(uint2);
    }
    (void)hipFree(d_chunk);
    (void)hipFree(d_tot);
    (void)hipFree(d_P0);
    return hipSuccess;

fail:
    if (d_chunk) (void)hipFree(d_chunk);
    if (d_tot) (void)hipFree(d_tot);
    if (d_P0) (void)hipFree(d_P0);
    if (d_blocks) (void)hipFree(d_blocks);
    if (d_dir) (void)hipFree(d_dir);
    return err;
}

}  // namespace rsb

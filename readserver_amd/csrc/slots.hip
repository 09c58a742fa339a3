// slots.hip -- the single-request search layout ("slots") and its builder.
//
// The classic layout (block_format.h) costs two dependent HBM requests per Occ lookup: directory
// entry, then block.  MI355X serves ~46e9 random requests/s whatever their size
// (tools/gather_bench.hip), so requests, not bytes, bound the search.  A SLOT is a block whose
// position is computable: slot i holds the run pieces of BWT[i*S, (i+1)*S) (runs are split at slot
// borders) with absolute A/C/G/T counts at i*S -- address = slots + 128 * (p / S), one request.
// S = m << a (a = 7 or 8, m with an exact 32-bit reciprocal) is chosen from the mean run length so
// that the 64-run payload is ~3/4 full and shrunk while more than 1 window in 1000 needs more
// pieces; those chain to overflow blocks stored behind the slots (`next`), 64 pieces each.
//
// Slot / overflow block (128 B) = 4 quarters of 32 B, each { word0, word1, 16 run bytes }:
//   word0 of quarter t = count of symbol t+1 before the block (40 bits) | meta_t << 40
//       meta_0  span | start_1 << 12     span = symbols held by this block; start_t = symbols held
//       meta_1  start_2 | start_3 << 12  by quarters 0..t-1
//       meta_2  ostart | chain << 12     the window is split over several blocks (chain = 1):
//                                        ostart = symbols of the window before this block
//   word1 of quarter t < 3 = the symbols A,C,G,T held by quarters 0..t (4 x 11 bits): the rank of
//       a position in quarter t+1 starts from them, so it scans at most the 16 runs of one quarter
//       and never re-adds whole quarters
//   word1 of quarter 3 = next: index (into the same array) of the block that continues the
//       window, 0 = none
// Half of a block is counters: a lookup costs one 128-B request whatever the block holds, the
// search kernel is short of VALU issue and request slots, not of HBM capacity, and the number of
// distinct blocks a batch touches hardly depends on S (8.67e7 at S = 640, 8.79e7 at S = 384 on the
// bench batch) -- so the block is laid out for the cheapest rank, not for density.
// Memory is ~ n/S * 128 B (about the size of the classic index), so slots are built only on
// request / when HBM allows, NEXT TO the classic index, which the other kernels keep using.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "bwt_device.h"
#include "kernels.h"

namespace rsb {

// ---- classic index, one thread: block holding position p, and its header fields
__device__ __forceinline__ uint64_t thread_block_of(const rsbwt_view &ix, uint64_t p) {
    const uint2 e = ix.dir[p >> ix.dir_shift];
    uint64_t j = dir_decode<false>(ix, e, p);
    const uint64_t *w = (const uint64_t *)ix.blocks;
    for (;;) {
        const uint64_t w0 = w[16 * j], w1 = w[16 * j + 4], w2 = w[16 * j + 8];
        const uint64_t P0 = (w0 >> 40) | (((w1 >> 40) & 0xFFFFull) << 24);
        const uint32_t span = (uint32_t)(w2 >> 40) & 0xFFFu;
        if (p < P0 + span || j + 1 >= ix.nblocks) return j;
        ++j;
    }
}

__device__ __forceinline__ uint8_t classic_run(const rsbwt_view &ix, uint64_t j, uint32_t r) {
    const uint8_t *base = (const uint8_t *)ix.blocks + j * RSBWT_BLOCK_BYTES;
    return base[32u * (r / RSBWT_LANE_RUNS) + 8u + (r % RSBWT_LANE_RUNS)];
}

// Walks the run stream from a symbol position on.
struct run_walker {
    uint64_t j;      // classic block
    uint32_t r;      // run inside it
    uint32_t left;   // symbols of that run not yet consumed
    uint32_t sym;
    uint64_t cnt[4];  // A,C,G,T before the current position

    __device__ void seek(const rsbwt_view &ix, uint64_t p) {
        j = thread_block_of(ix, p);
        const uint64_t *w = (const uint64_t *)ix.blocks + 16 * j;
        const uint64_t P0 = (w[0] >> 40) | (((w[4] >> 40) & 0xFFFFull) << 24);
        for (int c = 0; c < 4; ++c) cnt[c] = w[4 * c] & RSBWT_COUNT_MASK;
        uint64_t at = P0;
        r = 0;
        for (;;) {
            const uint8_t u = classic_run(ix, j, r);
            const uint32_t len = u & 31u;
            sym = u >> 5;
            if (at + len > p || (r + 1 >= RSBWT_BLOCK_RUNS && j + 1 >= ix.nblocks)) {
                const uint32_t skip = (uint32_t)(p - at);
                if (sym >= 1u) cnt[sym - 1u] += skip;
                left = len > skip ? len - skip : 0u;
                return;
            }
            if (sym >= 1u) cnt[sym - 1u] += len;
            at += len;
            if (++r == RSBWT_BLOCK_RUNS) { r = 0; ++j; }
        }
    }
    // next piece of at most `want` symbols; returns its length (0 at the end of the stream)
    __device__ uint32_t take(const rsbwt_view &ix, uint32_t want, uint32_t &piece_sym) {
        while (left == 0u) {
            if (++r == RSBWT_BLOCK_RUNS) { r = 0; ++j; }
            if (j >= ix.nblocks) return 0u;
            const uint8_t u = classic_run(ix, j, r);
            left = u & 31u;
            sym = u >> 5;
            if (left == 0u && j + 1 >= ix.nblocks && r + 1 >= RSBWT_BLOCK_RUNS) return 0u;
        }
        const uint32_t len = left < want ? left : want;
        left -= len;
        piece_sym = sym;
        if (sym >= 1u) cnt[sym - 1u] += len;
        return len;
    }
};

constexpr uint32_t SLOT_RUNS = 64;  // run pieces per slot / overflow block (4 quarters of 16)

// pass 1: overflow blocks each window needs
__global__ void __launch_bounds__(256)
slot_count_kernel(const rsbwt_view ix, uint32_t S, uint64_t nslots, uint32_t *__restrict__ novf) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    run_walker w;
    w.seek(ix, i * S);
    uint32_t remaining = (uint32_t)((ix.n - i * S) < S ? (ix.n - i * S) : S);
    uint32_t pieces = 0, ps;
    while (remaining) {
        const uint32_t len = w.take(ix, remaining, ps);
        if (!len) break;
        remaining -= len;
        ++pieces;
    }
    novf[i] = pieces > SLOT_RUNS ? (pieces - 1u) / SLOT_RUNS : 0u;
}

__device__ void flush_slot_block(uint4 *dst, const uint8_t *buf, uint32_t used, const uint64_t cnt0[4],
                                 uint32_t next, uint32_t ostart) {
    uint32_t wds[16];
    for (int d = 0; d < 16; ++d) {
        uint32_t x = 0;
        for (int k = 0; k < 4; ++k) {
            const uint32_t idx = 4u * d + k;
            x |= (uint32_t)(idx < used ? buf[idx] : 0u) << (8 * k);
        }
        wds[d] = x;
    }
    uint32_t start[4] = {0, 0, 0, 0}, span = 0;
    uint32_t held[4] = {0, 0, 0, 0};  // A,C,G,T held so far
    uint64_t word1[4] = {0, 0, 0, 0};
    for (int q = 0; q < 4; ++q) {
        start[q] = span;
        for (int d = 0; d < 4; ++d) {
            const uint32_t w = wds[4 * q + d];
            for (int k = 0; k < 4; ++k) {
                const uint32_t len = (w >> (8 * k)) & 31u, sym = (w >> (8 * k + 5)) & 7u;
                span += len;
                if (sym >= 1u && sym <= 4u) held[sym - 1u] += len;
            }
        }
        if (q < 3)
            word1[q] = (uint64_t)held[0] | ((uint64_t)held[1] << 11) | ((uint64_t)held[2] << 22) | ((uint64_t)held[3] << 33);
    }
    word1[3] = next;
    const bool chain = next != 0u || ostart != 0u;
    const uint32_t meta[4] = {span | (start[1] << 12), start[2] | (start[3] << 12),
                              ostart | (chain ? 1u << 12 : 0u), 0u};
    for (int q = 0; q < 4; ++q) {
        const uint64_t word0 = (cnt0[q] & RSBWT_COUNT_MASK) | ((uint64_t)meta[q] << 40);
        dst[2 * q] = make_uint4((uint32_t)word0, (uint32_t)(word0 >> 32), (uint32_t)word1[q], (uint32_t)(word1[q] >> 32));
        dst[2 * q + 1] = make_uint4(wds[4 * q], wds[4 * q + 1], wds[4 * q + 2], wds[4 * q + 3]);
    }
}

// pass 2: write slot i and its overflow chain (blocks nslots + ovf_base[i] ...)
__global__ void __launch_bounds__(256)
slot_write_kernel(const rsbwt_view ix, uint32_t S, uint64_t nslots, const uint64_t *__restrict__ ovf_base,
                  const uint32_t *__restrict__ novf, uint4 *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    run_walker w;
    w.seek(ix, i * S);
    uint32_t remaining = (uint32_t)((ix.n - i * S) < S ? (ix.n - i * S) : S);
    uint8_t buf[SLOT_RUNS];
    uint64_t cnt0[4] = {w.cnt[0], w.cnt[1], w.cnt[2], w.cnt[3]};
    uint32_t used = 0, ostart = 0, emitted = 0, chain = 0;
    const uint32_t nchain = novf[i];
    uint64_t dst = i;  // block being filled
    for (;;) {
        uint32_t ps = 0;
        const uint32_t len = remaining ? w.take(ix, remaining, ps) : 0u;
        if (len && used == SLOT_RUNS) {
            // block full and the window goes on: chain to the next overflow block
            const uint64_t nxt = nslots + ovf_base[i] + chain;
            flush_slot_block(out + dst * 8, buf, used, cnt0, (uint32_t)nxt, ostart);
            ++chain;
            dst = nxt;
            ostart = emitted;
            used = 0;
            // counts at the start of the new block = counts before this piece
            for (int c = 0; c < 4; ++c) cnt0[c] = w.cnt[c];
            if (ps >= 1u) cnt0[ps - 1u] -= len;
        }
        if (!len) break;
        buf[used++] = (uint8_t)((ps << 5) | len);
        emitted += len;
        remaining -= len;
    }
    flush_slot_block(out + dst * 8, buf, used, cnt0, 0u, ostart);
    (void)nchain;
}

// ---- exclusive scan u32 -> u64 (three small kernels)
__global__ void __launch_bounds__(1024)
scan_sums_kernel(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ sums) {
    __shared__ uint64_t part[16];
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    uint64_t v = i < n ? in[i] : 0;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t s = 0;
        for (int k = 0; k < 16; ++k) s += part[k];
        sums[blockIdx.x] = s;
    }
}

__global__ void __launch_bounds__(1024)
scan_top_kernel(uint64_t *__restrict__ sums, uint64_t nchunks, uint64_t *__restrict__ total) {
    __shared__ uint64_t part[1024];
    const uint64_t per = (nchunks + 1023) / 1024;
    const uint64_t b = (uint64_t)threadIdx.x * per, e = b + per < nchunks ? b + per : nchunks;
    uint64_t s = 0;
    for (uint64_t c = b; c < e; ++c) s += sums[c];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int k = 0; k < 1024; ++k) { const uint64_t v = part[k]; part[k] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    s = part[threadIdx.x];
    for (uint64_t c = b; c < e; ++c) { const uint64_t v = sums[c]; sums[c] = s; s += v; }
}

__global__ void __launch_bounds__(1024)
scan_final_kernel(const uint32_t *__restrict__ in, uint64_t n, const uint64_t *__restrict__ sums,
                  uint64_t *__restrict__ out) {
    __shared__ uint64_t sc[2][1024];
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const uint64_t v = i < n ? in[i] : 0;
    int cur = 0;
    sc[0][threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint64_t x = sc[cur][threadIdx.x];
        if ((int)threadIdx.x >= off) x += sc[cur][threadIdx.x - off];
        sc[cur ^ 1][threadIdx.x] = x;
        cur ^= 1;
        __syncthreads();
    }
    if (i < n) out[i] = sums[blockIdx.x] + sc[cur][threadIdx.x] - v;
}

// ---- choice of S and its reciprocal
static bool reciprocal32(uint32_t m, uint32_t *magic, uint32_t *shift) {
    // q = mulhi(x, magic) >> shift == x / m for every 32-bit x (round-up reciprocal: exact when
    // magic * m - 2^(32+shift) <= 2^shift)
    if (m < 2) return false;
    if ((m & (m - 1)) == 0) {  // power of two: mulhi(x, 2^31) >> (log2 m - 1)
        uint32_t s = 0;
        while ((1u << s) < m) ++s;
        *magic = 0x80000000u;
        *shift = s - 1;
        return true;
    }
    for (uint32_t s = 0; s < 6; ++s) {
        const unsigned __int128 two = (unsigned __int128)1 << (32 + s);
        const unsigned __int128 M = (two + m - 1) / m;
        if (M >> 32) continue;
        if (M * m - two <= ((unsigned __int128)1 << s)) {
            *magic = (uint32_t)M;
            *shift = s;
            return true;
        }
    }
    return false;
}

// The largest usable span <= cap (S = m << a, m with an exact 32-bit reciprocal; at least the smallest)
static bool span_at_most(uint64_t n, double cap, slot_params *sp) {
    const uint32_t a = n <= (1ull << 39) ? 7u : 8u;  // p >> a must fit 32 bits
    uint32_t best_m = 0, best_magic = 0, best_shift = 0;
    for (uint32_t m = 2; m <= 32; ++m) {
        uint32_t mg, sh;
        if (!reciprocal32(m, &mg, &sh)) continue;
        const double S = (double)(m << a);
        if (S > 4095.0) break;  // span and ostart are 12-bit fields
        if (S <= cap || best_m == 0) { best_m = m; best_magic = mg; best_shift = sh; }
    }
    if (!best_m) return false;
    sp->S = best_m << a;
    sp->a = a;
    sp->magic = best_magic;
    sp->shift = best_shift;
    sp->nslots = (n + sp->S - 1) / sp->S;
    if (sp->nslots == 0) sp->nslots = 1;
    return true;
}

// The span asked for, or the starting point of build_slots' choice: windows of ~3/4 of the 64-run
// payload (mean run length x 49).  build_slots then shrinks it while more than 1 window in 1000
// needs an overflow block.
bool choose_slot_span(uint64_t n, uint64_t num_runs, uint32_t want_S, slot_params *sp) {
    const double L = num_runs ? (double)n / (double)num_runs : 1.0;
    return span_at_most(n, (want_S ? (double)want_S : 49.0 * L) * 1.04, sp);
}

#define HIP_TRY(x)              \
    do {                        \
        hipError_t _e = (x);    \
        if (_e != hipSuccess) { \
            err = _e;           \
            goto fail;          \
        }                       \
    } while (0)

hipError_t build_slots(const rsbwt_view &ix, uint64_t num_runs, uint32_t want_S, hipStream_t stream,
                       slot_view *out, uint64_t *bytes, int *range_error) {
    hipError_t err = hipSuccess;
    *range_error = 0;
    slot_params sp;
    if (!choose_slot_span(ix.n, num_runs, want_S, &sp)) { *range_error = 1; return hipSuccess; }
    uint64_t ns = 0, nchunks = 0;
    uint32_t *d_novf = nullptr;
    uint64_t *d_sums = nullptr, *d_base = nullptr, *d_total = nullptr;
    uint4 *d_out = nullptr;
    uint64_t total_ovf = 0;
    // Count the overflow blocks of the candidate span; unless the caller fixed it, shrink it while
    // more than 1 window in 1000 overflows.  A search pass makes 64 lookups: at 1.5 % overflowing
    // windows most passes pay an extra dependent fetch round (measured: 4 % slower than at 0.03 %),
    // and how full the windows may be for that depends on the spread of the run lengths, which
    // only the data tells.  One streaming pass over the blocks per candidate.
    for (;;) {
        ns = sp.nslots;
        nchunks = (ns + 1023) / 1024;
        if (ns >= (1ull << 32)) { *range_error = 1; goto fail; }
        HIP_TRY(hipMalloc(&d_novf, ns * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&d_sums, nchunks * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&d_base, ns * sizeof(uint64_t)));
        if (!d_total) HIP_TRY(hipMalloc(&d_total, sizeof(uint64_t)));
        hipLaunchKernelGGL(slot_count_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, ix, sp.S, ns, d_novf);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)nchunks), dim3(1024), 0, stream, d_novf, ns, d_sums);
        hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, stream, d_sums, nchunks, d_total);
        hipLaunchKernelGGL(scan_final_kernel, dim3((unsigned)nchunks), dim3(1024), 0, stream, d_novf, ns, d_sums, d_base);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&total_ovf, d_total, sizeof total_ovf, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        slot_params smaller;
        if (want_S || total_ovf * 1000 <= ns || !span_at_most(ix.n, (double)sp.S - 1.0, &smaller) || smaller.S >= sp.S) break;
        sp = smaller;
        (void)hipFree(d_novf); (void)hipFree(d_sums); (void)hipFree(d_base);
        d_novf = nullptr; d_sums = nullptr; d_base = nullptr;
    }
    if (ns + total_ovf >= (1ull << 32)) { *range_error = 1; goto fail; }
    HIP_TRY(hipMalloc(&d_out, (ns + total_ovf) * RSBWT_BLOCK_BYTES));
    hipLaunchKernelGGL(slot_write_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, ix, sp.S, ns,
                       d_base, d_novf, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    out->slots = d_out;
    out->p = sp;
    out->noverflow = total_ovf;
    *bytes = (ns + total_ovf) * RSBWT_BLOCK_BYTES;
    (void)hipFree(d_novf); (void)hipFree(d_sums); (void)hipFree(d_base); (void)hipFree(d_total);
    return hipSuccess;
fail:
    if (d_novf) (void)hipFree(d_novf);
    if (d_sums) (void)hipFree(d_sums);
    if (d_base) (void)hipFree(d_base);
    if (d_total) (void)hipFree(d_total);
    if (d_out) (void)hipFree(d_out);
    return err;
}

}  // namespace rsb

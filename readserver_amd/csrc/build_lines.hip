// build_lines.hip -- builds the HBM index (window lines, line_format.h) from run bytes in HBM.
//
// Role of RLEBWT::initialiseFMIndex (src/bwt/rlebwt.cpp:34-148) in the reference: one pass over
// the runs producing cumulative checkpoints and C[].  Here:
//   1. tile_totals    per 256-run tile: symbols and A/C/G/T counts, scanned inside its 256-tile
//                     chunk (one workgroup); chunk totals
//   2. scan_chunks    exclusive scan of the chunk totals (one workgroup), totals -> n, C[]
//   3. count_groups   one thread per group of 16 windows: seeks its first symbol through the tile
//                     prefix, walks its runs and decides -- exactly as the write pass will -- which
//                     windows spill and how many far lines the group needs (line_format.h,
//                     build_group<false>); also the statistics the choice of S is made from
//   4. scan           far lines before each group
//   5. write_groups   the same walk, writing the group's 17 lines and its far lines
// Everything is streaming over R run bytes (read three times) and ~1.55 R bytes written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "kernels.h"
#include "line_format.h"

namespace rsb {

// Workgroups of 256 threads per CU the two group kernels are compiled for (tuning knob, tools/build_variant.sh): a
// thread lays out a group of 16 windows, byte after byte -- a chain of dependent loads -- so what the kernels need is
// waves in flight, not registers: left to itself hipcc keeps the group's piece buffer in 256 VGPRs, one wave per SIMD
// (2.44 s per 20 GB shard); 8 = 60 VGPRs, 8 waves per SIMD: 1.26 s (2: 1.58 s; 4: 12.9 s -- 816 bytes of scratch per lane).
#ifndef RSB_BUILD_MIN_WGS
#define RSB_BUILD_MIN_WGS 8
#endif
constexpr int TILE_RUNS = 256;  // run bytes per tile = per thread
constexpr int CHUNK_TILES = 256;  // tiles per chunk = threads per workgroup

struct seek_index {
    const uint8_t *runs;
    uint64_t R;
    const uint64_t *chunk_pre;  // nchunks x 5 (symbols, A, C, G, T before the chunk)
    const uint32_t *tile_rel;   // ntiles x 5 (the same before the tile, relative to its chunk)
    uint64_t nchunks, ntiles;
};

__global__ void __launch_bounds__(CHUNK_TILES)
tile_totals_kernel(const uint8_t *__restrict__ runs, uint64_t R, uint64_t ntiles, bool aligned16,
                   uint32_t *__restrict__ tile_rel, uint64_t *__restrict__ chunk_tot, uint32_t *__restrict__ bad) {
    __shared__ uint32_t sc[2][CHUNK_TILES][5];
    const uint64_t tile = (uint64_t)blockIdx.x * CHUNK_TILES + threadIdx.x;
    uint32_t t[5] = {0, 0, 0, 0, 0};
    bool invalid = false;
    if (tile < ntiles) {
        const uint64_t base = tile * TILE_RUNS;
        auto add = [&](uint32_t w) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t len = (w >> (8 * k)) & 31u, sym = (w >> (8 * k + 5)) & 7u;
                t[0] += len;
                t[1] += sym == 1u ? len : 0u;
                t[2] += sym == 2u ? len : 0u;
                t[3] += sym == 3u ? len : 0u;
                t[4] += sym == 4u ? len : 0u;
                invalid |= sym > 4u;
            }
        };
        if (aligned16 && base + TILE_RUNS <= R) {
            const uint4 *p = reinterpret_cast<const uint4 *>(runs + base);
            for (int i = 0; i < TILE_RUNS / 16; ++i) {
                const uint4 x = p[i];
                add(x.x); add(x.y); add(x.z); add(x.w);
            }
        } else {
            for (int i = 0; i < TILE_RUNS / 4; ++i) {
                uint32_t w = 0;
                for (int k = 0; k < 4; ++k) {
                    const uint64_t a = base + (uint64_t)(4 * i + k);
                    if (a < R) w |= (uint32_t)runs[a] << (8 * k);
                }
                add(w);
            }
        }
    }
    if (invalid) atomicOr(bad, 1u);
    // inclusive Hillis-Steele scan of the 256 tile totals (u32 is enough inside a chunk)
    int cur = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[0][threadIdx.x][i] = t[i];
    __syncthreads();
    for (int off = 1; off < CHUNK_TILES; off <<= 1) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            uint32_t v = sc[cur][threadIdx.x][i];
            if ((int)threadIdx.x >= off) v += sc[cur][threadIdx.x - off][i];
            sc[cur ^ 1][threadIdx.x][i] = v;
        }
        cur ^= 1;
        __syncthreads();
    }
    if (tile < ntiles) {
#pragma unroll
        for (int i = 0; i < 5; ++i) tile_rel[tile * 5 + i] = sc[cur][threadIdx.x][i] - t[i];
    }
    if (threadIdx.x == CHUNK_TILES - 1) {
#pragma unroll
        for (int i = 0; i < 5; ++i) chunk_tot[(uint64_t)blockIdx.x * 5 + i] = sc[cur][threadIdx.x][i];
    }
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

// A reader standing at symbol position P (P < n).
__device__ void seek_reader(const seek_index &sx, uint64_t P, run_reader &rd) {
    uint64_t lo = 0, hi = sx.nchunks - 1;  // largest chunk with symbols-before <= P
    while (hi > lo) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (sx.chunk_pre[mid * 5] > P) hi = mid - 1;
        else lo = mid;
    }
    const uint64_t c = lo;
    const uint64_t cpos = sx.chunk_pre[c * 5];
    const uint32_t want = (uint32_t)(P - cpos);
    uint64_t tl = c * CHUNK_TILES, th = tl + CHUNK_TILES - 1;
    if (th >= sx.ntiles) th = sx.ntiles - 1;
    while (th > tl) {  // largest tile of the chunk with symbols-before <= P
        const uint64_t mid = tl + (th - tl + 1) / 2;
        if (sx.tile_rel[mid * 5] > want) th = mid - 1;
        else tl = mid;
    }
    uint64_t cnt[4];
    for (int i = 0; i < 4; ++i) cnt[i] = sx.chunk_pre[c * 5 + 1 + i] + sx.tile_rel[tl * 5 + 1 + i];
    rd.start(sx.runs, sx.R, tl * TILE_RUNS, cnt);
    rd.skip_symbols(P - (cpos + sx.tile_rel[tl * 5]));
}

__global__ void __launch_bounds__(256, RSB_BUILD_MIN_WGS)
count_groups_kernel(const seek_index sx, const span_params sp, uint64_t n, uint64_t nwin, uint64_t ngroups,
                    uint32_t *__restrict__ far_lines, unsigned long long *__restrict__ stats, uint64_t every, bool room) {
    // every > 1: a SAMPLE of the groups (every `every`-th one), statistics only -- what the choice of S is tried on
    // before the one full pass (a full pass reads all the run bytes: 1.2 s for a 20 GB shard)
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * every;
    group_stats st = {0, 0, 0, 0};
    if (g < ngroups) {
        run_reader rd;
        seek_reader(sx, g * GROUP * (uint64_t)sp.S, rd);
        st = build_group<false>(sp, n, nwin, g, rd, nullptr, 0, room);
        if (far_lines) far_lines[g] = st.far_lines;
    }
    // statistics: one atomic set per wave
    unsigned long long v[4] = {st.far_lines, st.chunk_windows, st.far_windows, st.spilled_symbols};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off);
        if ((threadIdx.x & 63) == 0 && v[i]) atomicAdd(&stats[i], v[i]);
    }
}

__global__ void __launch_bounds__(256, RSB_BUILD_MIN_WGS)
write_groups_kernel(const seek_index sx, const span_params sp, uint64_t n, uint64_t nwin, uint64_t ngroups,
                    const uint64_t *__restrict__ far_before, uint64_t first_far, uint32_t *__restrict__ lines, bool room) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    run_reader rd;
    seek_reader(sx, g * GROUP * (uint64_t)sp.S, rd);
    build_group<true>(sp, n, nwin, g, rd, lines, first_far + far_before[g], room);
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

#define HIP_TRY(x)              \
    do {                        \
        hipError_t _e = (x);    \
        if (_e != hipSuccess) { \
            err = _e;           \
            goto fail;          \
        }                       \
    } while (0)

hipError_t build_lines(const void *d_runs, uint64_t num_runs, uint32_t want_span, bool hint_room, hipStream_t stream,
                       build_result *out, int *build_error) {
    hipError_t err = hipSuccess;
    *build_error = 0;
    const uint8_t *runs = (const uint8_t *)d_runs;
    const uint64_t R = num_runs;
    const uint64_t ntiles = R ? (R + TILE_RUNS - 1) / TILE_RUNS : 1;
    const uint64_t nchunks = (ntiles + CHUNK_TILES - 1) / CHUNK_TILES;
    const bool aligned16 = ((uintptr_t)runs & 15u) == 0;
    uint32_t *d_tile = nullptr, *d_far = nullptr, *d_lines = nullptr, *d_bad = nullptr;
    uint64_t *d_chunk = nullptr, *d_tot = nullptr, *d_sums = nullptr, *d_base = nullptr, *d_total = nullptr;
    unsigned long long *d_stats = nullptr;
    uint64_t tot[5] = {0, 0, 0, 0, 0};
    unsigned long long stats[4] = {0, 0, 0, 0};
    uint32_t bad = 0;
    shard_view v;
    memset(&v, 0, sizeof v);
    memset(out, 0, sizeof *out);
    v.sel_shift = hint_room ? SEL_SHIFT_SPARSE : SEL_SHIFT_DENSE;
    v.hint_room = hint_room ? 1u : 0u;

    if (nchunks >= (1ull << 31)) { *build_error = BUILD_ERANGE; return hipSuccess; }
    HIP_TRY(hipMalloc(&d_tile, ntiles * 5 * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d_chunk, nchunks * 5 * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&d_tot, 5 * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&d_bad, sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d_stats, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&d_total, sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(tile_totals_kernel, dim3((unsigned)nchunks), dim3(CHUNK_TILES), 0, stream, runs, R, ntiles,
                       aligned16, d_tile, d_chunk, d_bad);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(1), dim3(1024), 0, stream, d_chunk, nchunks, d_tot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (bad) { *build_error = BUILD_EFORMAT; goto fail; }
    {
        const uint64_t n = tot[0];
        if (n >= MAX_SYMBOLS) { *build_error = BUILD_ERANGE; goto fail; }
        v.n = n;
        // tot = {n, A, C, G, T}; '$' = n - (A+C+G+T).  C[] as rlebwt.cpp:129-147.
        v.total[0] = n - (tot[1] + tot[2] + tot[3] + tot[4]);
        for (int c = 1; c < 5; ++c) v.total[c] = tot[c];
        v.C[0] = 0;
        for (int c = 1; c < 5; ++c) v.C[c] = v.C[c - 1] + v.total[c - 1];
        if (n == 0) {  // nothing to lay out
            v.sp = make_span(MAX_SPAN);
            out->view = v;
            out->num_runs = R;
            goto done;
        }
        // Window span: ~88 pieces per window (1.55 bytes per run byte with ~1.5 % of the positions
        // one request further away on the bench stream; tools/ sweep in DESIGN.md); shrunk while more
        // than 2.5 % of the positions spill or more than 1.5 % of the windows need far lines -- how
        // full a window may be depends on the spread of the run lengths, which only the data tells.
        // (a shard laid out with room for a psi hint in every window line keeps 88 of a line's 96 piece bytes: the
        // same fill of what is left)
        const double L = (double)n / (double)(R ? R : 1);
        const double target = hint_room ? 88.0 * (double)HINT_PIECES / (double)LINE_PIECES : 88.0;
        span_params sp = make_span(want_span ? want_span : (uint32_t)(target * L + 0.5));
        seek_index sx = {runs, R, d_chunk, d_tile, nchunks, ntiles};
        uint64_t nwin = 0, ngroups = 0, nsum = 0;
        // S is first tried on a sample of the groups (one in 64, spread over the whole shard): the spans that
        // would clearly spill too much are passed over without a full pass each.  The full pass below still
        // decides: a span the sample let through is shrunk further if the whole shard says so.
        // (the layout with hint room is a tenth larger, and eight 20 GB shards of it plus the run bytes of the one being
        // built are all an MI355X holds: its span comes down in steps of 1.25 %, not 5 %, to the first that passes)
        const double shrink = hint_room ? 0.9875 : 0.95;
        const int max_attempts = hint_room ? 16 : 4;
        constexpr uint64_t SAMPLE_EVERY = 64;
        if (!want_span && (n + sp.S - 1) / sp.S / GROUP >= 64 * SAMPLE_EVERY) {
            for (int attempt = 0; attempt < max_attempts && sp.S > 8u; ++attempt) {
                const uint64_t nw = (n + sp.S - 1) / sp.S, ng = (nw + GROUP - 1) / GROUP, ns = (ng + SAMPLE_EVERY - 1) / SAMPLE_EVERY;
                HIP_TRY(hipMemsetAsync(d_stats, 0, 4 * sizeof(unsigned long long), stream));
                hipLaunchKernelGGL(count_groups_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, sx, sp, n, nw, ng,
                                   (uint32_t *)nullptr, d_stats, SAMPLE_EVERY, hint_room);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(stats, d_stats, sizeof stats, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                // (held to the limits themselves: a sample within 5 % of them is left to the full pass)
                const bool ok = stats[3] * SAMPLE_EVERY * 40 <= n + n / 20 && stats[2] * SAMPLE_EVERY * 200 <= (nw + nw / 20) * 3;
                if (ok) break;
                const span_params smaller = make_span((uint32_t)((double)sp.S * shrink));
                if (smaller.S >= sp.S) break;
                sp = smaller;
            }
        }
        for (int attempt = 0;; ++attempt) {
            nwin = (n + sp.S - 1) / sp.S;
            ngroups = (nwin + GROUP - 1) / GROUP;
            nsum = (ngroups + 1023) / 1024;
            if (ngroups * (GROUP + 1) >= 0xFFFFFFF0ull) { *build_error = BUILD_ERANGE; goto fail; }
            HIP_TRY(hipMalloc(&d_far, ngroups * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(d_stats, 0, 4 * sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(count_groups_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, stream, sx, sp,
                               n, nwin, ngroups, d_far, d_stats, (uint64_t)1, hint_room);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(stats, d_stats, sizeof stats, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            const bool ok = stats[3] * 40 <= n && stats[2] * 200 <= nwin * 3;
            if (want_span || ok || attempt >= max_attempts || sp.S <= 8u) break;
            const span_params smaller = make_span((uint32_t)((double)sp.S * shrink));
            if (smaller.S >= sp.S) break;
            sp = smaller;
            (void)hipFree(d_far);
            d_far = nullptr;
        }
        const uint64_t first_far = ngroups * (GROUP + 1), nlines = first_far + stats[0];
        if (nlines >= 0xFFFFFFF0ull) { *build_error = BUILD_ERANGE; goto fail; }
        HIP_TRY(hipMalloc(&d_sums, nsum * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&d_base, ngroups * sizeof(uint64_t)));
        hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)nsum), dim3(1024), 0, stream, d_far, ngroups, d_sums);
        hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, stream, d_sums, nsum, d_total);
        hipLaunchKernelGGL(scan_final_kernel, dim3((unsigned)nsum), dim3(1024), 0, stream, d_far, ngroups, d_sums, d_base);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMalloc(&d_lines, nlines * LINE_BYTES));
        HIP_TRY(hipMemsetAsync(d_lines, 0, nlines * LINE_BYTES, stream));
        hipLaunchKernelGGL(write_groups_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, stream, sx, sp, n,
                           nwin, ngroups, d_base, first_far, d_lines, hint_room);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        v.lines = d_lines;
        v.nwin = nwin;
        v.nlines = nlines;
        v.first_far = first_far;
        v.sp = sp;
        out->view = v;
        out->num_runs = R;
        out->hbm_bytes = nlines * LINE_BYTES;
        out->far_lines = stats[0];
        out->chunk_windows = stats[1];
        out->far_windows = stats[2];
        out->spilled_symbols = stats[3];
    }
done:
    (void)hipFree(d_tile); (void)hipFree(d_chunk); (void)hipFree(d_tot); (void)hipFree(d_bad);
    (void)hipFree(d_stats); (void)hipFree(d_total);
    if (d_far) (void)hipFree(d_far);
    if (d_sums) (void)hipFree(d_sums);
    if (d_base) (void)hipFree(d_base);
    return hipSuccess;

fail:
    if (d_tile) (void)hipFree(d_tile);
    if (d_chunk) (void)hipFree(d_chunk);
    if (d_tot) (void)hipFree(d_tot);
    if (d_bad) (void)hipFree(d_bad);
    if (d_stats) (void)hipFree(d_stats);
    if (d_total) (void)hipFree(d_total);
    if (d_far) (void)hipFree(d_far);
    if (d_sums) (void)hipFree(d_sums);
    if (d_base) (void)hipFree(d_base);
    if (d_lines) (void)hipFree(d_lines);
    return err;
}

}  // namespace rsb

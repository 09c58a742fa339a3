// kernels.hip -- query kernels of the popBWT engine (gfx950).
//
// Hot path: search_kernel = batched findInterval (src/bwt/query.cpp:24-41).  A query is owned
// by an octet of lanes = two DPP quads: quad L resolves Occ(b, lower-1), quad U resolves
// Occ(b, upper) (updateInterval, query.cpp:11-15), in the same instructions.  A wavefront
// carries 8 queries; each octet walks its own list of queries and refills as soon as one ends,
// so short (early-terminating) queries do not idle the wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include <algorithm>

#include "bwt_device.h"
#include "kernels.h"
#include "synth_runs.h"

namespace rsb {

// ------------------------------------------------------------------------------------------
// ASCII -> 2-bit packing.  One thread per (k-mer, word).
// ------------------------------------------------------------------------------------------
__global__ void pack_kernel(const uint8_t *__restrict__ kmers, size_t Q, uint32_t k, size_t stride,
                            uint32_t wpq, uint64_t *__restrict__ packed,
                            uint8_t *__restrict__ valid) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    for (size_t q = gid; q < Q; q += nthreads) {
        const uint8_t *s = kmers + q * stride;
        bool ok = k > 0;
        for (uint32_t w = 0; w < wpq; ++w) {
            uint64_t word = 0;
            const uint32_t base = w * 32u;
            const uint32_t m = (k - base) < 32u ? (k - base) : 32u;
            for (uint32_t i = 0; i < m; ++i) {
                const uint8_t ch = s[base + i];
                uint32_t code;
                switch (ch) {
                case 'A': code = 0; break;
                case 'C': code = 1; break;
                case 'G': code = 2; break;
                case 'T': code = 3; break;
                default: code = 0; ok = false; break;
                }
                word |= (uint64_t)code << (2u * i);
            }
            packed[q * wpq + w] = word;
        }
        valid[q] = ok ? 1 : 0;
    }
}

// Dense input (stride == k): a workgroup's 256 k-mers are one contiguous byte range; it is
// read with aligned 16-byte loads into LDS and packed from there.
__global__ void __launch_bounds__(256)
pack_dense_kernel(const uint8_t *__restrict__ kmers, size_t Q, uint32_t k, uint32_t wpq,
                  uint64_t *__restrict__ packed, uint8_t *__restrict__ valid) {
    extern __shared__ uint4 s_bytes[];  // 256 * k + 32 bytes
    const size_t q0 = (size_t)blockIdx.x * 256;
    const size_t nq = (Q - q0) < 256 ? (Q - q0) : 256;
    const size_t begin = q0 * k, end = begin + nq * k;
    const uintptr_t base = (uintptr_t)kmers;
    const size_t a0 = (base + begin) & ~(size_t)15;          // aligned address of the first chunk
    const size_t skew = (base + begin) - a0;
    const size_t nchunks = (skew + (end - begin) + 15) / 16;
    const size_t last = (base + Q * (size_t)k + 15) & ~(size_t)15;  // do not read past the array's last chunk
    for (size_t c = threadIdx.x; c < nchunks; c += 256) {
        const uintptr_t addr = a0 + 16 * c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (addr + 16 <= last) v = *reinterpret_cast<const uint4 *>(addr);
        s_bytes[c] = v;
    }
    __syncthreads();
    if (threadIdx.x >= nq) return;
    // four symbols per step: the k-mer's bytes are pulled out of LDS as dwords (v_alignbyte over
    // the misaligned start), coded ((c >> 1) ^ (c >> 2)) & 3 = A 0, C 1, G 2, T 3 in all four bytes
    // at once, and checked by looking the codes up again ("ACGT"[code] must give the byte back)
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(s_bytes);
    const uint32_t o = (uint32_t)skew + threadIdx.x * k, sh = o & 3u;
    const uint32_t nd = (k + 3u) / 4u;
    const size_t q = q0 + threadIdx.x;
    bool ok = k > 0;
    uint64_t word = 0;
    uint32_t lo = sw[o >> 2];
    for (uint32_t i = 0; i < nd; ++i) {
        const uint32_t hi = sw[(o >> 2) + i + 1u];  // at most the 16 spare bytes behind the range
        uint32_t x = __builtin_amdgcn_alignbyte(hi, lo, sh);
        lo = hi;
        const uint32_t left = k - 4u * i;
        if (left < 4u) {  // the k-mer ends inside this dword: 'A' beyond it
            const uint32_t keep = (1u << (8u * left)) - 1u;
            x = (x & keep) | (0x41414141u & ~keep);
        }
        const uint32_t code = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
        ok = ok && __builtin_amdgcn_perm(0u, 0x54474341u, code) == x;
        const uint32_t c8 = (code | (code >> 6) | (code >> 12) | (code >> 18)) & 0xFFu;
        word |= (uint64_t)c8 << (8u * (i & 7u));
        if ((i & 7u) == 7u || i + 1u == nd) {
            packed[q * wpq + (i >> 3)] = word;
            word = 0;
        }
    }
    valid[q] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// Batched backward search.
// ------------------------------------------------------------------------------------------
template <bool COUNT_WORK, bool COUNTS_ONLY, bool KTAB, bool EXACT8>
__global__ void __launch_bounds__(256)
search_kernel(const rsbwt_view ix, const uint64_t *__restrict__ packed,
              const uint8_t *__restrict__ valid, size_t Q, uint32_t k, uint32_t wpq,
              uint64_t *__restrict__ out_lower, uint64_t *__restrict__ out_upper,
              unsigned long long *__restrict__ work) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t = lane & 3u;          // lane in quad
    const uint32_t role = (lane >> 2) & 1u;  // 0: lower-1 side, 1: upper side
    const size_t octet = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const size_t noctets = ((size_t)gridDim.x * blockDim.x) >> 3;
    const uint4 *lane_base = ix.blocks + 2u * t;

    // C[] and the symbol totals, indexed by symbol rank, in LDS: a dynamically indexed kernel
    // argument would otherwise become a dependent global load at the end of every step.
    __shared__ uint64_t s_C[8], s_total[8];
    if (threadIdx.x < 5) {
        s_C[threadIdx.x] = ix.C[threadIdx.x];
        s_total[threadIdx.x] = ix.total[threadIdx.x];
    }
    __syncthreads();

    size_t q = octet;
    bool fresh = true;   // the next iteration starts query q
    int j = 0;           // index of the next symbol to prepend
    uint64_t word = 0;   // packed word holding symbol j
    uint64_t lo = 0, hi = 0;
    unsigned long long w_steps = 0, w_occ = 0, w_blocks = 0, w_ktab = 0;

    while (q < Q) {
        bool done = false;
        if (fresh) {
            fresh = false;
            j = (int)k - 1;
            const uint64_t *pq = packed + q * wpq;
            const uint8_t okb = valid[q];  // both loads issue together
            word = pq[(uint32_t)j >> 5];
            if (okb == 0) {
                if ((lane & 7u) == 0u) {
                    if (COUNTS_ONLY) out_lower[q] = 0;
                    else { out_lower[q] = 1; out_upper[q] = 0; }
                }
                q += noctets;
                fresh = true;
                continue;
            }
            bool from_table = false;
            if (KTAB) {
                // the last T symbols of the k-mer select a precomputed interval
                const uint32_t T = ix.ktab_depth;
                const uint32_t off = 2u * (k - T);      // bit offset of symbol k-T in the packing
                const uint32_t w0 = off >> 6, sh = off & 63u;
                uint64_t bits = (w0 == ((uint32_t)j >> 5) ? word : pq[w0]) >> sh;
                if (sh + 2u * T > 64u) bits |= word << (64u - sh);  // spills into the last word
                const uint64_t code = bits & ((1ull << (2u * T)) - 1ull);
                const uint64_t e = ix.ktab[code];
                const uint32_t width = (uint32_t)(e >> RSBWT_COUNT_BITS);
                if (COUNT_WORK) w_ktab += 1;
                if (width != RSBWT_KTAB_WIDE) {
                    from_table = true;
                    lo = e & RSBWT_COUNT_MASK;
                    hi = lo + width - 1ull;
                    j = (int)(k - T) - 1;
                    // an already-empty tabulated suffix ends the search (query.cpp:35-37); the
                    // unsigned compare keeps the reference's (0, 2^64-1) corner going, as it does
                    done = (lo > hi) || (j < 0);
                    if (!done && ((uint32_t)j >> 5) != ((k - 1u) >> 5)) word = pq[(uint32_t)j >> 5];
                }
            }
            if (!from_table) {
                const uint32_t b = (uint32_t)((word >> (2u * ((uint32_t)j & 31u))) & 3u) + 1u;
                // initInterval (query.cpp:18-21): Occ(b, n-1) is the symbol's total.
                lo = s_C[b];
                hi = lo + s_total[b] - 1ull;
                --j;
                done = j < 0;
            }
        }
        if (!done) {
            if ((j & 31) == 31) word = packed[q * wpq + ((uint32_t)j >> 5)];
            const uint32_t b = (uint32_t)((word >> (2u * ((uint32_t)j & 31u))) & 3u) + 1u;
            // updateInterval (query.cpp:11-15)
            const uint64_t pb = s_C[b];
            // Occ(b, -1) = 0: lower - 1 at lower == 0, and upper itself after a step that found no b
            // at the top of the BWT (upper = 0 + 0 - 1 wraps; the reference carries on the same way
            // and reports the empty interval one step later: query.cpp:11-15,35, rlebwt.cpp:269)
            const uint64_t p_raw = role ? hi : lo - 1ull;
            const bool skip = p_raw == ~0ull;
            const uint64_t p = skip ? 0ull : p_raw;
            lane_block lb;
            block_meta bm;
            uint32_t off;
            const uint64_t blk = quad_fetch<EXACT8>(ix, lane_base, p, t, lb, bm, off);
            uint64_t occ = quad_rank(lb, bm, t, b, off);
            occ = skip ? 0ull : occ;
            const uint64_t other = dpp_mov64<DPP_ROW_HALF_MIRROR>(occ);
            const uint64_t occL = role ? other : occ;
            const uint64_t occU = role ? occ : other;
            if (COUNT_WORK) {
                const uint64_t oblk = dpp_mov64<DPP_ROW_HALF_MIRROR>(blk);
                if ((lane & 7u) == 0u) {  // role 0, so `skip` is the L side's
                    w_steps += 1;
                    w_occ += skip ? 1 : 2;
                    w_blocks += (skip || oblk == blk) ? 1 : 2;  // (an upper-side skip counts as a read of block 0)
                }
            }
            lo = pb + occL;
            hi = pb + occU - 1ull;
            --j;
            done = (lo > hi) || (j < 0);  // query.cpp:35-37
        }
        if (done) {
            if ((lane & 7u) == 0u) {
                if (COUNTS_ONLY) {
                    out_lower[q] = hi >= lo ? hi - lo + 1ull : 0ull;  // service.cpp:304
                } else {
                    out_lower[q] = lo;
                    out_upper[q] = hi;
                }
            }
            q += noctets;
            fresh = true;
        }
    }
    if (COUNT_WORK) {
        if ((lane & 7u) == 0u && (w_steps || w_ktab)) {
            atomicAdd(&work[0], w_steps);
            atomicAdd(&work[1], w_occ);
            atomicAdd(&work[2], w_blocks);
            atomicAdd(&work[3], w_ktab);
        }
    }
}

// k-mer table build: codes -> packed queries, and (lower, upper) -> 8-byte entries
__global__ void ktab_codes_kernel(uint64_t base, size_t m, uint64_t *__restrict__ packed,
                                  uint8_t *__restrict__ valid) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        packed[i] = base + i;
        valid[i] = 1;
    }
}

__global__ void ktab_encode_kernel(const uint64_t *__restrict__ lower, const uint64_t *__restrict__ upper,
                                   size_t m, uint64_t *__restrict__ entries) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        const uint64_t lo = lower[i], up = upper[i];
        uint64_t width = up + 1ull - lo;  // 0 when empty: an empty result always has upper = lower - 1
        if (lo > up && up + 1ull != lo) width = RSBWT_KTAB_WIDE;  // never produced; stay safe
        if (width >= RSBWT_KTAB_WIDE) width = RSBWT_KTAB_WIDE;
        entries[i] = (lo & RSBWT_COUNT_MASK) | (width << RSBWT_COUNT_BITS);
    }
}

// ------------------------------------------------------------------------------------------
// class BWT mirrors, batched: one quad per item.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ascii_rank(uint8_t ch) {
    return ch == 'A' ? 1u : ch == 'C' ? 2u : ch == 'G' ? 3u : ch == 'T' ? 4u : 0u;
}

// Occ for any symbol rank 0..4; `$` comes from P0 minus the four stored counts.
__device__ __forceinline__ uint64_t quad_occ_any(const rsbwt_view &ix, uint32_t b, uint64_t p,
                                                 uint32_t t) {
    lane_block lb;
    block_meta bm;
    uint32_t off;
    quad_fetch<false>(ix, ix.blocks + 2u * t, p, t, lb, bm, off);
    uint64_t r = quad_rank(lb, bm, t, b, off);  // for b == 0: just the in-block '$' symbols
    if (b == 0u) {
        const uint64_t cnt = ((uint64_t)(lb.hdr_hi & 0xFFu) << 32) | lb.hdr_lo;
        r += (p - off) - quad_sum64(cnt);  // P0 = p - off
    }
    return r;
}

__global__ void occ_batch_kernel(const rsbwt_view ix, const uint8_t *__restrict__ syms,
                                 const uint64_t *__restrict__ index, size_t n,
                                 uint64_t *__restrict__ out) {
    const size_t quad = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const size_t nquads = ((size_t)gridDim.x * blockDim.x) >> 2;
    const uint32_t t = threadIdx.x & 3u;
    for (size_t i = quad; i < n; i += nquads) {
        const uint32_t b = ascii_rank(syms[i]);
        uint64_t p = index[i];
        uint64_t r = 0;
        if (p != ~0ull && ix.n != 0) {  // getOcc(b, -1) = 0
            if (p >= ix.n) p = ix.n - 1;
            r = quad_occ_any(ix, b, p, t);
        }
        if (t == 0u) out[i] = r;
    }
}

__global__ void char_batch_kernel(const rsbwt_view ix, const uint64_t *__restrict__ index, size_t n,
                                  uint8_t *__restrict__ out) {
    const size_t quad = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const size_t nquads = ((size_t)gridDim.x * blockDim.x) >> 2;
    const uint32_t t = threadIdx.x & 3u;
    for (size_t i = quad; i < n; i += nquads) {
        uint64_t p = index[i];
        if (p >= ix.n) p = ix.n - 1;
        lane_block lb;
        block_meta bm;
        uint32_t off;
        quad_fetch<false>(ix, ix.blocks + 2u * t, p, t, lb, bm, off);
        const uint32_t c = quad_char(lb, bm, t, off);
        if (t == 0u) out[i] = (uint8_t)("$ACGT"[c]);
    }
}

// Count of symbol b (1..4) before block j; `$` (0) from P0.  One thread, two or four loads.
__device__ __forceinline__ uint64_t block_count_before(const rsbwt_view &ix, uint64_t j, uint32_t b) {
    const uint64_t *w = (const uint64_t *)ix.blocks;  // header word t at u64 index 16*j + 4*t
    if (b != 0u) return w[16 * j + 4 * (b - 1u)] & RSBWT_COUNT_MASK;
    const uint64_t w0 = w[16 * j], w1 = w[16 * j + 4], w2 = w[16 * j + 8], w3 = w[16 * j + 12];
    const uint64_t P0 = (w0 >> 40) | (((w1 >> 40) & 0xFFFFull) << 24);
    return P0 - ((w0 & RSBWT_COUNT_MASK) + (w1 & RSBWT_COUNT_MASK) + (w2 & RSBWT_COUNT_MASK) +
                 (w3 & RSBWT_COUNT_MASK));
}

// Offset inside block j of its `offset`-th b (offset >= 1, at most the block's count of b):
// RLEBWT::getOccAt's scan (src/bwt/rlebwt.cpp:245-263), a quarter (32 B = two 16-byte loads) at a
// time -- quarters before the one holding the occurrence are only added up, 4 runs per dot4.
__device__ uint64_t thread_select_in_block(const rsbwt_view &ix, uint64_t j, uint32_t b, uint64_t offset) {
    const uint4 *blk = ix.blocks + 8 * j;
    const uint32_t bb = b * 0x01010101u;
    uint64_t index = 0;
    for (uint32_t t = 0; t < 4u; ++t) {
        const uint4 a = blk[2 * t], c = blk[2 * t + 1];
        const uint32_t r[6] = {a.z, a.w, c.x, c.y, c.z, c.w};
        uint32_t matched = 0, total = 0;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            matched = dword_matched(r[d], bb, matched);
            total = __builtin_amdgcn_udot4(r[d] & 0x1F1F1F1Fu, 0x01010101u, total, false);
        }
        if (offset > matched && t < 3u) {
            offset -= matched;
            index += total;
            continue;
        }
#pragma unroll
        for (int d = 0; d < 6; ++d) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t u = (r[d] >> (8 * k)) & 0xFFu, len = u & 31u;
                if ((u >> 5) != b) { index += len; continue; }
                if (offset <= len) return index + offset - 1;
                offset -= len;
                index += len;
            }
        }
    }
    return index;
}

// getOccAt(b, bc): position of the bc-th b (bc >= 1).  Floor search over the block headers
// (BPTree::select's role, include/bwt/BPTree.h:50-67), then RLEBWT::getOccAt's scan
// (src/bwt/rlebwt.cpp:245-263) over the block's 96 runs.  One thread per item.
__device__ uint64_t thread_occ_at(const rsbwt_view &ix, uint32_t b, uint64_t bc) {
    uint64_t lo = 0, hi = ix.nblocks - 1;  // largest j with count_before(j) < bc
    while (hi > lo) {
        const uint64_t mid = (lo + hi + 1) >> 1;
        if (block_count_before(ix, mid, b) >= bc) hi = mid - 1;
        else lo = mid;
    }
    const uint64_t j = lo;
    const uint64_t *w = (const uint64_t *)ix.blocks + 16 * j;
    const uint64_t P0 = (w[0] >> 40) | (((w[4] >> 40) & 0xFFFFull) << 24);
    return P0 + thread_select_in_block(ix, j, b, bc - block_count_before(ix, j, b));
}

__global__ void occ_at_batch_kernel(const rsbwt_view ix, const uint8_t *__restrict__ syms,
                                    const uint64_t *__restrict__ bc, size_t n,
                                    uint64_t *__restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    for (size_t i = gid; i < n; i += nthreads) {
        const uint32_t b = ascii_rank(syms[i]);
        uint64_t c = bc[i];
        const uint64_t tot = ix.total[b];
        uint64_t r = ix.n;  // out of range -> n
        if (c >= 1 && c <= tot) r = thread_occ_at(ix, b, c);
        out[i] = r;
    }
}

// ------------------------------------------------------------------------------------------
// Synthetic inputs.
// ------------------------------------------------------------------------------------------
__global__ void synth_runs_kernel(uint8_t *__restrict__ runs, uint64_t num_runs, uint64_t seed) {
    // 16 run bytes per thread, stored as one uint4
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nvec = num_runs / 16;
    for (uint64_t v = gid; v < nvec; v += nthreads) {
        uint32_t wds[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                x |= (uint32_t)synth_run_byte(seed, v * 16 + (uint64_t)(d * 4 + k)) << (8 * k);
            wds[d] = x;
        }
        reinterpret_cast<uint4 *>(runs)[v] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
    }
    if (gid < (num_runs & 15)) runs[nvec * 16 + gid] = synth_run_byte(seed, nvec * 16 + gid);
}

// K-mers that occur in the index: start at a random row r, emit F(r) as the last symbol, then
// repeatedly prepend BWT[r] and move r <- LF(r).  Every suffix of the k-mer then has a non-empty
// interval.  A walk that meets '$' restarts from another row.  One quad per k-mer.
__global__ void sample_present_kernel(const rsbwt_view ix, size_t Q, uint32_t k, size_t stride,
                                      uint64_t seed, uint8_t *__restrict__ kmers) {
    const size_t quad = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const size_t nquads = ((size_t)gridDim.x * blockDim.x) >> 2;
    const uint32_t t = threadIdx.x & 3u;
    const uint64_t nonterm = ix.n - ix.C[1];  // rows whose first symbol is not '$'
    for (size_t q = quad; q < Q; q += nquads) {
        uint8_t *out = kmers + q * stride;
        bool ok = false;
        for (uint32_t attempt = 0; attempt < 64 && !ok && nonterm > 0; ++attempt) {
            uint64_t r = ix.C[1] + synth_mix64(seed ^ synth_mix64(q * 64 + attempt)) % nonterm;
            uint32_t f = 1;
            while (f < 4 && ix.C[f + 1] <= r) ++f;
            if (t == 0u) out[k - 1] = (uint8_t)("$ACGT"[f]);
            ok = true;
            for (int i = (int)k - 2; i >= 0; --i) {
                lane_block lb;
                block_meta bm;
                uint32_t off;
                quad_fetch<false>(ix, ix.blocks + 2u * t, r, t, lb, bm, off);
                const uint32_t c = quad_char(lb, bm, t, off);
                if (c == 0u) { ok = false; break; }
                r = ix.C[c] + quad_rank(lb, bm, t, c, off) - 1ull;  // LF(r)
                if (t == 0u) out[i] = (uint8_t)("$ACGT"[c]);
            }
        }
        if (!ok && t == 0u)
            for (uint32_t i = 0; i < k; ++i) out[i] = 'A';
    }
}

// ------------------------------------------------------------------------------------------
// 1-mismatch search by composition: every k-mer expands to itself plus its 3k
// single-substitution variants, in canonical order (variant 0 = the k-mer; 1 + 3i + d = position
// i carries the d-th base of ACGT \ {original}); each variant is then an exact findInterval.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
variants_kernel(const uint64_t *__restrict__ packed, const uint8_t *__restrict__ valid, size_t Q,
                uint32_t k, uint32_t wpq, uint64_t *__restrict__ vpacked, uint8_t *__restrict__ vvalid) {
    const uint32_t V = 3u * k + 1u;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q * V) return;
    const size_t q = i / V;
    const uint32_t v = (uint32_t)(i % V);
    vvalid[i] = valid[q];
    for (uint32_t w = 0; w < wpq; ++w) vpacked[i * wpq + w] = packed[q * wpq + w];
    if (v == 0u) return;
    const uint32_t pos = (v - 1u) / 3u, d = (v - 1u) % 3u;
    const uint32_t w = pos >> 5, sh = 2u * (pos & 31u);
    uint64_t word = packed[q * wpq + w];
    const uint32_t orig = (uint32_t)(word >> sh) & 3u;
    const uint32_t repl = d < orig ? d : d + 1u;
    word = (word & ~(3ull << sh)) | ((uint64_t)repl << sh);
    vpacked[i * wpq + w] = word;
}

// Compaction of a slice's [m][V] variant intervals into the list of those that occur (SURVEY 8
// f3's output): a count per k-mer, an exclusive scan on the host (m is a few 10^4), a scatter.
__global__ void __launch_bounds__(256)
hits1mm_count_kernel(const uint64_t *__restrict__ lower, const uint64_t *__restrict__ upper, size_t m,
                     uint32_t V, uint32_t *__restrict__ counts) {
    // one wave per k-mer: lanes stride over its V variants
    const size_t q = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (q >= m) return;
    uint32_t c = 0;
    for (uint32_t v = lane; v < V; v += 64u) c += lower[q * V + v] <= upper[q * V + v] ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0u) counts[q] = c;
}

struct hit_1mm_rec {  // = rsbwt_hit_1mm (include/rsbwt.h)
    uint64_t lower, upper;
    uint32_t query;
    int16_t pos;
    char base, reserved;
};

__global__ void __launch_bounds__(256)
hits1mm_write_kernel(const uint64_t *__restrict__ lower, const uint64_t *__restrict__ upper,
                     const uint64_t *__restrict__ packed, size_t m, uint32_t V, uint32_t wpq,
                     const uint64_t *__restrict__ offsets, uint32_t query0, hit_1mm_rec *__restrict__ hits) {
    const size_t q = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (q >= m) return;
    uint64_t at = offsets[q];
    for (uint32_t v0 = 0; v0 < V; v0 += 64u) {  // V in canonical (pos, base) order: keep it
        const uint32_t v = v0 + lane;
        const bool hit = v < V && lower[q * V + v] <= upper[q * V + v];
        const uint64_t mask = __builtin_amdgcn_ballot_w64(hit);
        if (hit) {
            hit_1mm_rec r;
            r.lower = lower[q * V + v];
            r.upper = upper[q * V + v];
            r.query = query0 + (uint32_t)q;
            r.reserved = 0;
            if (v == 0u) {
                r.pos = -1;
                r.base = 0;
            } else {
                const uint32_t pos = (v - 1u) / 3u, d = (v - 1u) % 3u;
                const uint32_t orig = (uint32_t)(packed[q * wpq + (pos >> 5)] >> (2u * (pos & 31u))) & 3u;
                r.pos = (int16_t)pos;
                r.base = "ACGT"[d < orig ? d : d + 1u];
            }
            hits[at + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = r;
        }
        at += __builtin_popcountll(mask);
    }
}

// ------------------------------------------------------------------------------------------
// Read extraction (query.cpp:43-85): the read whose suffix is SA row `row`.
// ------------------------------------------------------------------------------------------
// Sampled select: sel[c-1][m] = block holding the (m << SEL_SHIFT) + 1 -th occurrence of symbol c.
constexpr uint32_t SEL_SHIFT = 8;  // one sample per 256 occurrences: the header search spans 1-3 blocks

__global__ void __launch_bounds__(256)
select_sample_kernel(const rsbwt_view ix, uint32_t *__restrict__ sel, uint64_t stride_m) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ix.nblocks) return;
    for (uint32_t c = 1; c <= 4; ++c) {
        const uint64_t cb = block_count_before(ix, j, c);
        const uint64_t ce = (j + 1 < ix.nblocks) ? block_count_before(ix, j + 1, c) : ix.total[c];
        if (ce == cb) continue;
        // occurrences cb+1 .. ce live here; sample m is occurrence (m << SEL_SHIFT) + 1
        for (uint64_t m = (cb + (1ull << SEL_SHIFT) - 1) >> SEL_SHIFT; (m << SEL_SHIFT) < ce; ++m)
            sel[(c - 1) * stride_m + m] = (uint32_t)j;
    }
}

// getOccAt with the sample table bounding the header search (BPTree::select's role)
__device__ uint64_t thread_occ_at_sampled(const rsbwt_view &ix, const uint32_t *__restrict__ sel,
                                          uint64_t stride_m, uint32_t b, uint64_t bc) {
    const uint64_t m = (bc - 1) >> SEL_SHIFT;
    uint64_t lo = sel[(b - 1) * stride_m + m];
    uint64_t hi = ((m + 1) << SEL_SHIFT) < ix.total[b] ? sel[(b - 1) * stride_m + m + 1] : ix.nblocks - 1;
    while (hi > lo) {  // largest j in [lo, hi] with count_before(j) < bc
        const uint64_t mid = (lo + hi + 1) >> 1;
        if (block_count_before(ix, mid, b) >= bc) hi = mid - 1;
        else lo = mid;
    }
    const uint64_t j = lo;
    const uint64_t *w = (const uint64_t *)ix.blocks + 16 * j;
    const uint64_t P0 = (w[0] >> 40) | (((w[4] >> 40) & 0xFFFFull) << 24);
    return P0 + thread_select_in_block(ix, j, b, bc - block_count_before(ix, j, b));
}

// extractPrefix (query.cpp:43-63): LF walk left until '$'.  One quad per row; the characters are
// produced right to left, so they are written downwards from the end of the row's buffer.
__global__ void __launch_bounds__(256)
extract_prefix_kernel(const rsbwt_view ix, const uint64_t *__restrict__ rows, size_t n,
                      uint8_t *__restrict__ out, uint32_t stride, uint32_t *__restrict__ plen) {
    const size_t quad = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const size_t nquads = ((size_t)gridDim.x * blockDim.x) >> 2;
    const uint32_t t = threadIdx.x & 3u;
    const uint4 *lane_base = ix.blocks + 2u * t;
    for (size_t i = quad; i < n; i += nquads) {
        uint64_t idx = rows[i];
        uint8_t *buf = out + i * (size_t)stride;
        uint32_t len = 0;
        bool fits = idx < ix.n;
        while (fits) {
            lane_block lb;
            block_meta bm;
            uint32_t off;
            quad_fetch<false>(ix, lane_base, idx, t, lb, bm, off);
            const uint32_t c = quad_char(lb, bm, t, off);
            if (c == 0u) break;
            if (len == stride) { fits = false; break; }  // the reference would spin (query.cpp:48)
            idx = ix.C[c] + quad_rank(lb, bm, t, c, off) - 1ull;  // C[b] + Occ(b, idx-1)
            if (t == 0u) buf[stride - 1u - len] = (uint8_t)("$ACGT"[c]);
            ++len;
        }
        if (t == 0u) plen[i] = fits ? len : 0xFFFFFFFFu;
    }
}

// extractPostfix (query.cpp:65-85): F / select walk right until '$', appended after the prefix.
__global__ void __launch_bounds__(256)
extract_postfix_kernel(const rsbwt_view ix, const uint32_t *__restrict__ sel, uint64_t stride_m,
                       const uint64_t *__restrict__ rows, size_t n, uint8_t *__restrict__ out,
                       uint32_t stride, const uint32_t *__restrict__ plen, uint32_t *__restrict__ tlen) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t *buf = out + i * (size_t)stride;
    const uint32_t pl = plen[i];
    if (pl == 0xFFFFFFFFu || rows[i] >= ix.n) { tlen[i] = 0xFFFFFFFFu; return; }
    for (uint32_t k = 0; k < pl; ++k) buf[k] = buf[stride - pl + k];  // prefix into place (src >= dst)
    uint32_t len = pl;
    uint64_t idx = rows[i];
    for (;;) {
        uint32_t f = 0;  // getF, rlebwt.cpp:307-314
        while (f < 4u && ix.C[f + 1] <= idx) ++f;
        if (f == 0u) break;
        if (len == stride) { len = 0xFFFFFFFFu; break; }
        idx = thread_occ_at_sampled(ix, sel, stride_m, f, idx - ix.C[f] + 1ull);
        buf[len++] = (uint8_t)("$ACGT"[f]);
    }
    tlen[i] = len;
}

// ------------------------------------------------------------------------------------------
// Host launchers
// ------------------------------------------------------------------------------------------
static inline int grid_for(size_t items_per_block, size_t items, int max_blocks) {
    size_t g = (items + items_per_block - 1) / items_per_block;
    if (g < 1) g = 1;
    if (g > (size_t)max_blocks) g = (size_t)max_blocks;
    return (int)g;
}

hipError_t launch_pack(const void *d_kmers, size_t Q, uint32_t k, size_t stride, void *d_packed,
                       void *d_valid, hipStream_t stream) {
    if (Q == 0) return hipSuccess;
    const uint32_t wpq = (k + 31u) / 32u ? (k + 31u) / 32u : 1u;
    const size_t lds = 256 * (size_t)k + 32;
    if (stride == k && k > 0 && lds <= 48 * 1024 && (Q + 255) / 256 < (1ull << 31))
        hipLaunchKernelGGL(pack_dense_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), lds, stream,
                           (const uint8_t *)d_kmers, Q, k, wpq, (uint64_t *)d_packed, (uint8_t *)d_valid);
    else
        hipLaunchKernelGGL(pack_kernel, dim3(grid_for(256, Q, 8192)), dim3(256), 0, stream,
                           (const uint8_t *)d_kmers, Q, k, stride, wpq, (uint64_t *)d_packed,
                           (uint8_t *)d_valid);
    return hipGetLastError();
}

template <bool CW, bool CO, bool KT>
static void launch_search_e(bool exact8, int grid, hipStream_t stream, const rsbwt_view &ix,
                            const uint64_t *pk, const uint8_t *vd, size_t Q, uint32_t k, uint32_t wpq,
                            uint64_t *lo, uint64_t *up, unsigned long long *work) {
    if (exact8)
        hipLaunchKernelGGL((search_kernel<CW, CO, KT, true>), dim3(grid), dim3(256), 0, stream, ix, pk, vd,
                           Q, k, wpq, lo, up, work);
    else
        hipLaunchKernelGGL((search_kernel<CW, CO, KT, false>), dim3(grid), dim3(256), 0, stream, ix, pk,
                           vd, Q, k, wpq, lo, up, work);
}

template <bool CW, bool CO>
static void launch_search_t(bool ktab, int grid, hipStream_t stream, const rsbwt_view &ix,
                            const uint64_t *pk, const uint8_t *vd, size_t Q, uint32_t k, uint32_t wpq,
                            uint64_t *lo, uint64_t *up, unsigned long long *work) {
    const bool exact8 = ix.dir_shift == 8;
    if (ktab) launch_search_e<CW, CO, true>(exact8, grid, stream, ix, pk, vd, Q, k, wpq, lo, up, work);
    else launch_search_e<CW, CO, false>(exact8, grid, stream, ix, pk, vd, Q, k, wpq, lo, up, work);
}

// RSBWT_SEARCH_KERNEL=octet|wave picks the kernel form (default: wave where the index allows it)
static bool prefer_wave_kernel() {
    static const int v = [] {
        const char *e = getenv("RSBWT_SEARCH_KERNEL");
        return (e && e[0] == 'o') ? 0 : 1;
    }();
    return v != 0;
}

bool search_uses_wave_kernel(const rsbwt_view &ix, const slot_view *sv) {
    const bool have_slots = sv && sv->slots;
    return (have_slots || ix.dir_shift == 8) && prefer_wave_kernel();
}

hipError_t launch_search(const rsbwt_view &ix, const slot_view *sv, const void *d_packed, const void *d_valid,
                         size_t Q, uint32_t k, void *d_lower, void *d_upper, bool counts_only,
                         unsigned long long *d_work, int num_cus, hipStream_t stream, hipEvent_t ev0,
                         hipEvent_t ev1, const wave_search_extra *extra) {
    if (Q == 0) return hipSuccess;
    const bool have_slots = sv && sv->slots;
    if ((have_slots || ix.dir_shift == 8) && prefer_wave_kernel())
        return launch_search_wave(ix, have_slots ? sv : nullptr, d_packed, d_valid, Q, k, d_lower, d_upper,
                                  counts_only, d_work, num_cus, stream, ev0, ev1, extra);
    if (extra) return hipErrorInvalidValue;  // traced / resumed searches exist in the wave kernel only
    if (ev0) (void)hipEventRecord(ev0, stream);
    const uint32_t wpq = (k + 31u) / 32u ? (k + 31u) / 32u : 1u;
    // 32 queries per 256-thread workgroup; 8 workgroups per CU fill the 32 wave slots.
    const int grid = grid_for(32, Q, num_cus * 8);
    const uint64_t *pk = (const uint64_t *)d_packed;
    const uint8_t *vd = (const uint8_t *)d_valid;
    uint64_t *lo = (uint64_t *)d_lower, *up = (uint64_t *)d_upper;
    const bool ktab = ix.ktab != nullptr && ix.ktab_depth >= 2 && k >= ix.ktab_depth;
    if (d_work) {
        if (counts_only) launch_search_t<true, true>(ktab, grid, stream, ix, pk, vd, Q, k, wpq, lo, up, d_work);
        else launch_search_t<true, false>(ktab, grid, stream, ix, pk, vd, Q, k, wpq, lo, up, d_work);
    } else {
        if (counts_only) launch_search_t<false, true>(ktab, grid, stream, ix, pk, vd, Q, k, wpq, lo, up, d_work);
        else launch_search_t<false, false>(ktab, grid, stream, ix, pk, vd, Q, k, wpq, lo, up, d_work);
    }
    const hipError_t le = hipGetLastError();
    if (ev1) (void)hipEventRecord(ev1, stream);
    return le;
}

// Fills ix.ktab-to-be `d_entries` (4^T entries) by searching every T-mer with the table-less
// kernel, in slices that bound the temporary memory.  `ix` must not have a table yet.
hipError_t build_ktable(const rsbwt_view &ix, const slot_view *sv, uint32_t T, uint64_t *d_entries,
                        int num_cus, hipStream_t stream) {
    const uint64_t total = 1ull << (2u * T);
    const size_t SL = (size_t)std::min<uint64_t>(total, 1ull << 24);
    uint64_t *d_pk = nullptr, *d_lo = nullptr, *d_up = nullptr;
    uint8_t *d_ok = nullptr;
    hipError_t e;
    if ((e = hipMalloc(&d_pk, SL * 8)) != hipSuccess) return e;
    if ((e = hipMalloc(&d_lo, SL * 8)) != hipSuccess) { (void)hipFree(d_pk); return e; }
    if ((e = hipMalloc(&d_up, SL * 8)) != hipSuccess) { (void)hipFree(d_pk); (void)hipFree(d_lo); return e; }
    if ((e = hipMalloc(&d_ok, SL)) != hipSuccess) { (void)hipFree(d_pk); (void)hipFree(d_lo); (void)hipFree(d_up); return e; }
    rsbwt_view plain = ix;
    plain.ktab = nullptr;
    plain.ktab_depth = 0;
    // a deep table is built on top of a shallow one: with the (T/2)-mer table in place, each of
    // the 4^T searches starts from its last T/2 symbols' entry and takes half the LF steps
    uint64_t *d_half = nullptr;
    if (T >= 10u) {
        const uint32_t T0 = T / 2u;
        if ((e = hipMalloc(&d_half, 8ull << (2u * T0))) == hipSuccess) e = build_ktable(ix, sv, T0, d_half, num_cus, stream);
        if (e != hipSuccess) {
            (void)hipFree(d_pk); (void)hipFree(d_lo); (void)hipFree(d_up); (void)hipFree(d_ok);
            if (d_half) (void)hipFree(d_half);
            return e;
        }
        plain.ktab = d_half;
        plain.ktab_depth = T0;
    }
    for (uint64_t base = 0; base < total && e == hipSuccess; base += SL) {
        const size_t m = (size_t)std::min<uint64_t>(SL, total - base);
        const int g = (int)((m + 255) / 256);
        hipLaunchKernelGGL(ktab_codes_kernel, dim3(g), dim3(256), 0, stream, base, m, d_pk, d_ok);
        e = launch_search(plain, sv, d_pk, d_ok, m, T, d_lo, d_up, false, nullptr, num_cus, stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(ktab_encode_kernel, dim3(g), dim3(256), 0, stream, d_lo, d_up, m, d_entries + base);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_pk); (void)hipFree(d_lo); (void)hipFree(d_up); (void)hipFree(d_ok);
    if (d_half) (void)hipFree(d_half);
    return e;
}

hipError_t launch_occ_batch(const rsbwt_view &ix, const void *d_syms, const void *d_index, size_t n,
                            void *d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(occ_batch_kernel, dim3(grid_for(64, n, 8192)), dim3(256), 0, stream, ix,
                       (const uint8_t *)d_syms, (const uint64_t *)d_index, n, (uint64_t *)d_out);
    return hipGetLastError();
}

hipError_t launch_char_batch(const rsbwt_view &ix, const void *d_index, size_t n, void *d_out,
                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(char_batch_kernel, dim3(grid_for(64, n, 8192)), dim3(256), 0, stream, ix,
                       (const uint64_t *)d_index, n, (uint8_t *)d_out);
    return hipGetLastError();
}

hipError_t launch_occ_at_batch(const rsbwt_view &ix, const void *d_syms, const void *d_bc, size_t n,
                               void *d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(occ_at_batch_kernel, dim3(grid_for(256, n, 8192)), dim3(256), 0, stream, ix,
                       (const uint8_t *)d_syms, (const uint64_t *)d_bc, n, (uint64_t *)d_out);
    return hipGetLastError();
}

hipError_t launch_hits1mm_count(const void *d_lower, const void *d_upper, size_t m, uint32_t V, void *d_counts,
                                hipStream_t stream) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(hits1mm_count_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, stream,
                       (const uint64_t *)d_lower, (const uint64_t *)d_upper, m, V, (uint32_t *)d_counts);
    return hipGetLastError();
}

hipError_t launch_hits1mm_write(const void *d_lower, const void *d_upper, const void *d_packed, size_t m, uint32_t V,
                                uint32_t k, const void *d_offsets, uint32_t query0, void *d_hits, hipStream_t stream) {
    if (m == 0) return hipSuccess;
    const uint32_t wpq = (k + 31u) / 32u ? (k + 31u) / 32u : 1u;
    hipLaunchKernelGGL(hits1mm_write_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, stream,
                       (const uint64_t *)d_lower, (const uint64_t *)d_upper, (const uint64_t *)d_packed, m, V, wpq,
                       (const uint64_t *)d_offsets, query0, (hit_1mm_rec *)d_hits);
    return hipGetLastError();
}

hipError_t launch_variants(const void *d_packed, const void *d_valid, size_t Q, uint32_t k, void *d_vpacked,
                           void *d_vvalid, hipStream_t stream) {
    if (Q == 0) return hipSuccess;
    const uint32_t wpq = (k + 31u) / 32u ? (k + 31u) / 32u : 1u;
    const size_t total = Q * (3 * (size_t)k + 1);
    hipLaunchKernelGGL(variants_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const uint64_t *)d_packed, (const uint8_t *)d_valid, Q, k, wpq, (uint64_t *)d_vpacked,
                       (uint8_t *)d_vvalid);
    return hipGetLastError();
}

uint64_t select_sample_stride(const rsbwt_view &ix) {
    uint64_t mx = 0;
    for (int c = 1; c <= 4; ++c) mx = ix.total[c] > mx ? ix.total[c] : mx;
    return (mx >> SEL_SHIFT) + 2;
}

hipError_t launch_select_samples(const rsbwt_view &ix, uint32_t *d_sel, hipStream_t stream) {
    hipLaunchKernelGGL(select_sample_kernel, dim3((unsigned)((ix.nblocks + 255) / 256)), dim3(256), 0, stream, ix,
                       d_sel, select_sample_stride(ix));
    return hipGetLastError();
}

hipError_t launch_extract(const rsbwt_view &ix, const uint32_t *d_sel, const void *d_rows, size_t n,
                          void *d_out, uint32_t stride, void *d_plen, void *d_len, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(extract_prefix_kernel, dim3(grid_for(64, n, 8192)), dim3(256), 0, stream, ix,
                       (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride, (uint32_t *)d_plen);
    hipLaunchKernelGGL(extract_postfix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ix, d_sel,
                       select_sample_stride(ix), (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride,
                       (const uint32_t *)d_plen, (uint32_t *)d_len);
    return hipGetLastError();
}

hipError_t launch_synth_runs(void *d_runs, uint64_t num_runs, uint64_t seed, hipStream_t stream) {
    if (num_runs == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_runs_kernel, dim3(grid_for(256 * 16, num_runs, 16384)), dim3(256), 0,
                       stream, (uint8_t *)d_runs, num_runs, seed);
    return hipGetLastError();
}

hipError_t launch_sample_present(const rsbwt_view &ix, size_t Q, uint32_t k, size_t stride,
                                 uint64_t seed, void *d_kmers, hipStream_t stream) {
    if (Q == 0 || k == 0) return hipSuccess;
    hipLaunchKernelGGL(sample_present_kernel, dim3(grid_for(64, Q, 8192)), dim3(256), 0, stream, ix,
                       Q, k, stride, seed, (uint8_t *)d_kmers);
    return hipGetLastError();
}

}  // namespace rsb

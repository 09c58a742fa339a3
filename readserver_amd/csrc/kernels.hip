// kernels.hip -- the kernels around the search (gfx950): k-mer packing, the k-mer table build,
// the class BWT mirrors, read extraction, 1-mismatch variants and hit lists, synthetic inputs.
// The batched search itself is search_lines.hip; the layout is line_format.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include <algorithm>

#include "kernels.h"
#include "line_format.h"
#include "rank_device.h"
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

// Queries of lengths of their own, one after the other in `text` (query q = bytes off[q] .. off[q+1]): packed from word
// q * wpq on (wpq words hold the batch's longest), len[q] = its length, valid[q] = 0 for an empty one, one with a symbol
// outside ACGT (service.cpp:299: the callers' find_first_not_of("ACGT")) or one longer than 65,535.  A thread per query.
__global__ void pack_var_kernel(const uint8_t *__restrict__ text, const uint64_t *__restrict__ off, size_t Q, uint32_t wpq,
                                uint64_t *__restrict__ packed, uint8_t *__restrict__ valid, uint32_t *__restrict__ len) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    for (size_t q = gid; q < Q; q += nthreads) {
        const uint8_t *s = text + off[q];
        const uint64_t n64 = off[q + 1] - off[q];
        const bool fits = n64 != 0ull && n64 <= 65535ull && n64 <= 32ull * wpq;
        const uint32_t n = fits ? (uint32_t)n64 : 0u;
        bool ok = fits;
        for (uint32_t w = 0; w < wpq; ++w) {
            uint64_t word = 0;
            const uint32_t base = w * 32u;
            const uint32_t m = base >= n ? 0u : ((n - base) < 32u ? (n - base) : 32u);
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
        len[q] = n;
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

// k-mer table build: codes -> packed queries, and (lower, upper) -> 8-byte entries
__global__ void ktab_codes_kernel(uint64_t base, size_t m, uint64_t *__restrict__ packed,
                                  uint8_t *__restrict__ valid) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        packed[i] = base + i;
        valid[i] = 1;
    }
}

// grouped table: slice entry i = sibling i & 3 of group gbase + (i >> 2) (its last symbol in the code's high bits)
__global__ void ktab_group_codes_kernel(uint64_t gbase, uint32_t gbits, size_t m, uint64_t *__restrict__ packed,
                                        uint8_t *__restrict__ valid) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        packed[i] = ((uint64_t)(i & 3u) << gbits) | (gbase + (i >> 2));
        valid[i] = 1;
    }
}

__global__ void ktab_group_encode_kernel(const uint64_t *__restrict__ lower, const uint64_t *__restrict__ upper, size_t groups,
                                         uint32_t *__restrict__ records, uint32_t stride, unsigned long long *__restrict__ untabulated) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    uint64_t lo[4], up[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = lower[4 * g + i];
        up[i] = upper[4 * g + i];
    }
    uint32_t rec[3];
    ktab_group_encode(lo, up, rec);
    uint32_t *r = records + g * stride * 3u;
    r[0] = rec[0];
    r[1] = rec[1];
    r[2] = rec[2];
    uint32_t left = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) left += (ktab_group_entry(rec[0], rec[1], rec[2], i) >> COUNT_BITS) == KTAB_WIDE ? 1u : 0u;
    if (left) atomicAdd(untabulated, (unsigned long long)left);
}

__global__ void ktab_encode_kernel(const uint64_t *__restrict__ lower, const uint64_t *__restrict__ upper,
                                   size_t m, uint64_t *__restrict__ entries, uint32_t stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        const uint64_t lo = lower[i], up = upper[i];
        uint64_t width = up + 1ull - lo;  // 0 when empty: an empty result always has upper = lower - 1
        if (lo > up && up + 1ull != lo) width = KTAB_WIDE;  // never produced; stay safe
        if (width >= KTAB_WIDE) width = KTAB_WIDE;
        entries[i * stride] = (lo & COUNT_MASK) | (width << COUNT_BITS);
    }
}

// ------------------------------------------------------------------------------------------
// One LF step's worth of a window line, thread-private: the symbol at position p and the number
// of its occurrences up to p (getChar + getOcc of that symbol, query.cpp:49-57).  The header
// names the quarter holding the position, so 24 pieces are walked instead of 96; the ~1.5 % of
// positions past their window's own pieces take the general walk (line_format.h).
// ------------------------------------------------------------------------------------------
__device__ uint32_t thread_char_occ(const shard_view &v, uint64_t p, uint64_t *occ_of_char) {
    if (p >= v.n) p = v.n - 1ull;  // never for a walk over a sound index: keeps the line address inside it
    uint32_t pin;
    const uint32_t w = fast_window(p, v.sp.S, v.sp.inv, pin);
    const uint32_t o = pin + 1u;
    const uint32_t *L = v.lines + (uint64_t)(w + (w >> GROUP_SHIFT)) * LINE_DWORDS;
    const uint4 h0 = *reinterpret_cast<const uint4 *>(L), h1 = *reinterpret_cast<const uint4 *>(L + 4);
    const uint32_t m0 = h0.y >> 8, m1 = h0.w >> 8;
    const uint32_t s1 = m0 & 0x3FFu, s2 = (m0 >> 10) & 0x7FFu;
    const uint32_t s3 = s2 + (m1 & 0x3FFu), span = s3 + ((m1 >> 10) & 0x3FFu);
    if (o > span) return view_char_occ(v, p, occ_of_char);
    const uint32_t cq = (o > s1 ? 1u : 0u) + (o > s2 ? 1u : 0u) + (o > s3 ? 1u : 0u);
    const uint32_t start = cq == 0u ? 0u : cq == 1u ? s1 : cq == 2u ? s2 : s3;
    const uint2 *Q = reinterpret_cast<const uint2 *>(L + HDR_DWORDS + 6u * cq);
    const uint2 a = Q[0], b2 = Q[1], c2 = Q[2];
    const uint32_t r[6] = {a.x, a.y, b2.x, b2.y, c2.x, c2.y};
    // the piece holding the position
    uint32_t rem = o - start, c = 0, upto = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const uint32_t u = (r[i >> 2] >> (8 * (i & 3))) & 0xFFu, len = u & 31u;
        const bool here = rem != 0u && rem <= len;
        c = here ? (u >> 5) : c;
        upto = here ? (uint32_t)i : upto;
        rem = rem > len ? rem - len : 0u;
    }
    if (c == 0u) {  // '$': callers stop here (query.cpp:52); its count comes from the general walk
        return view_char_occ(v, p, occ_of_char);
    }
    // occurrences of c among the quarter's pieces up to the position
    uint32_t in = 0;
    rem = o - start;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const uint32_t u = (r[i >> 2] >> (8 * (i & 3))) & 0xFFu, len = u & 31u;
        const uint32_t take = len < rem ? len : rem;
        in += ((u >> 5) == c) ? take : 0u;
        rem -= take;
    }
    (void)upto;
    uint32_t before = 0;
    if (cq >= 2u) {
        const uint32_t hm = ((c <= 2u) ? h1.y : h1.w) >> 8;
        before = (hm >> (11u * ((c - 1u) & 1u))) & 0x7FFu;
    }
    if (cq & 1u) {
        const uint2 *P = reinterpret_cast<const uint2 *>(L + HDR_DWORDS + 6u * (cq - 1u));
        const uint2 x0 = P[0], x1 = P[1], x2 = P[2];
        const uint32_t bb = c * 0x01010101u;
        uint32_t m = dword_matched(x0.x, bb, 0u);
        m = dword_matched(x0.y, bb, m);
        m = dword_matched(x1.x, bb, m);
        m = dword_matched(x1.y, bb, m);
        m = dword_matched(x2.x, bb, m);
        m = dword_matched(x2.y, bb, m);
        before += m;
    }
    const uint32_t wlo = c == 1u ? h0.x : c == 2u ? h0.z : c == 3u ? h1.x : h1.z;
    const uint32_t whi = c == 1u ? h0.y : c == 2u ? h0.w : c == 3u ? h1.y : h1.w;
    *occ_of_char = (((uint64_t)(whi & 0xFFu) << 32) | wlo) + before + in;
    return c;
}

// ------------------------------------------------------------------------------------------
// class BWT mirrors, batched: one thread per item (line_format.h's scalar readers).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ascii_rank(uint8_t ch) {
    return ch == 'A' ? 1u : ch == 'C' ? 2u : ch == 'G' ? 3u : ch == 'T' ? 4u : 0u;
}

__global__ void __launch_bounds__(256)
occ_batch_kernel(const shard_view ix, const uint8_t *__restrict__ syms, const uint64_t *__restrict__ index, size_t n,
                 uint64_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = ascii_rank(syms[i]);
    uint64_t p = index[i];
    uint64_t r = 0;
    if (p != ~0ull && ix.n != 0) {  // getOcc(b, -1) = 0
        if (p >= ix.n) p = ix.n - 1;
        r = view_occ(ix, b, p);
    }
    out[i] = r;
}

__global__ void __launch_bounds__(256)
char_batch_kernel(const shard_view ix, const uint64_t *__restrict__ index, size_t n, uint8_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t p = index[i];
    if (p >= ix.n) p = ix.n - 1;
    out[i] = (uint8_t)("$ACGT"[view_char(ix, p)]);
}

// Sampled select: sel[c][m] describes the block of 2^sel_shift occurrences of symbol c that starts with occurrence
// (m << sel_shift) + 1: bits 0..31 = the window w0 holding that first occurrence, then four bytes k0..k3 with
// k_j + 1 = how many of the block's occurrences lie in windows <= w0 + j (capped at 256).  The window of the
// block's r-th occurrence is then w0 + [r > k0] + [r > k1] + [r > k2] + [r > k3] -- EXACT while r <= k3, a lower
// bound beyond (line_format.h, sample_window).  Round 2 kept the bare window (4 bytes) and interpolated between
// two samples: 31 % of first guesses were wrong and cost a second line fetch and a second pass
// (profiles/r02d_extract_profile.json).

__global__ void __launch_bounds__(256)
select_sample_kernel(const shard_view ix, uint64_t *__restrict__ sel, uint64_t stride_m) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ix.nwin) return;
    for (uint32_t c = 0; c <= 4; ++c)
        window_samples(ix, w, c, [&](uint64_t m, uint64_t word) { sel[c * stride_m + m] = word; });
}

// psi hints (line_format.h): one thread per window.  Needs the select samples (the window of an occurrence).
__global__ void __launch_bounds__(256)
psi_hint_kernel(const shard_view ix, const uint64_t *__restrict__ sel, uint64_t stride_m, uint32_t *__restrict__ lines,
                unsigned long long *__restrict__ made) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ix.nwin) return;
    uint32_t w0, kk;
    if (!window_psi_hint(ix, sel, stride_m, w, &w0, &kk)) return;
    uint32_t *Ln = lines + line_of_window(w) * LINE_DWORDS;
    const uint32_t hd = hint_dword((Ln[3] >> 28) & 3u);  // (kind: bits 20, 21 of meta_1)
    Ln[hd + 1u] = kk;
    Ln[hd] = w0;
    __threadfence();
    Ln[1] |= 1u << (8u + HINT_META0_BIT);  // the flag last: a reader that sees it sees the hint
    if (made) atomicAdd(made, 1ull);
}

// getOccAt with the sample table naming the window (BPTree::select's role)
__device__ uint64_t thread_occ_at_sampled(const shard_view &ix, const uint64_t *__restrict__ sel,
                                          uint64_t stride_m, uint32_t b, uint64_t bc) {
    const uint64_t w = select_window(ix, sel, stride_m, b, bc);
    return view_occ_at(ix, b, bc, w, w);
}

__global__ void __launch_bounds__(256)
occ_at_batch_kernel(const shard_view ix, const uint64_t *__restrict__ sel, uint64_t stride_m,
                    const uint8_t *__restrict__ syms, const uint64_t *__restrict__ bc, size_t n,
                    uint64_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = ascii_rank(syms[i]);
    const uint64_t c = bc[i];
    uint64_t r = ix.n;  // out of range -> n
    if (c >= 1 && c <= ix.total[b]) r = thread_occ_at_sampled(ix, sel, stride_m, b, c);
    out[i] = r;
}

// ------------------------------------------------------------------------------------------
// Synthetic inputs.
// ------------------------------------------------------------------------------------------
__global__ void synth_runs_kernel(uint8_t *__restrict__ runs, uint64_t num_runs, uint64_t seed, uint64_t first) {
    // 16 run bytes per thread, stored as one uint4; runs[i] = byte first + i of the stream
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
                x |= (uint32_t)synth_run_byte(seed, first + v * 16 + (uint64_t)(d * 4 + k)) << (8 * k);
            wds[d] = x;
        }
        reinterpret_cast<uint4 *>(runs)[v] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
    }
    if (gid < (num_runs & 15)) runs[nvec * 16 + gid] = synth_run_byte(seed, first + nvec * 16 + gid);
}

// K-mers that occur in the index: start at a random row r, emit F(r) as the last symbol, then
// repeatedly prepend BWT[r] and move r <- LF(r).  Every suffix of the k-mer then has a non-empty
// interval.  A walk that meets '$' restarts from another row.  One thread per k-mer.
__global__ void __launch_bounds__(256)
sample_present_kernel(const shard_view ix, size_t Q, uint32_t k, size_t stride, uint64_t seed,
                      uint8_t *__restrict__ kmers) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const uint64_t nonterm = ix.n - ix.C[1];  // rows whose first symbol is not '$'
    uint8_t *out = kmers + q * stride;
    bool ok = false;
    for (uint32_t attempt = 0; attempt < 64 && !ok && nonterm > 0; ++attempt) {
        uint64_t r = ix.C[1] + synth_mix64(seed ^ synth_mix64(q * 64 + attempt)) % nonterm;
        uint32_t f = 1;
        while (f < 4 && ix.C[f + 1] <= r) ++f;
        out[k - 1] = (uint8_t)("$ACGT"[f]);
        ok = true;
        for (int i = (int)k - 2; i >= 0; --i) {
            uint64_t occ = 0;
            const uint32_t c = thread_char_occ(ix, r, &occ);
            if (c == 0u) { ok = false; break; }
            r = ix.C[c] + occ - 1ull;  // LF(r)
            out[i] = (uint8_t)("$ACGT"[c]);
        }
    }
    if (!ok)
        for (uint32_t i = 0; i < k; ++i) out[i] = 'A';
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
    const size_t QV = Q * V;
    if (i >= QV) return;
    size_t q;
    uint32_t v;
    if (QV <= 0xFFFFFFFFull) {  // (a 64-bit divide costs this kernel more than its memory traffic)
        const uint32_t q32 = (uint32_t)i / V;
        q = q32;
        v = (uint32_t)i - q32 * V;
    } else {
        q = i / V;
        v = (uint32_t)(i - q * V);
    }
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

// ---- 1-mismatch hit list: the set bits of the search kernels' hit map, in order -------------------
// Three small launches over n/64 words (the map of 3.8*10^7 variants is 4.7 MB): set bits per block of
// 256 words, an exclusive scan of the block sums by one workgroup, the scatter.  blockIdx.y = the segment (the shard
// of a launch over several): maps hit_map_words(n) words apart, sparse results n records apart, lists cap records
// apart, block sums nblocks + 2 words apart.
constexpr uint32_t HITS_BLOCK_WORDS = 256;

__global__ void __launch_bounds__(256)
hit_block_counts_kernel(const uint64_t *__restrict__ bits, size_t nwords, size_t map_words, size_t nblocks,
                        unsigned long long *__restrict__ block_counts) {
    bits += (size_t)blockIdx.y * map_words;
    block_counts += (size_t)blockIdx.y * (nblocks + 2);
    const size_t w = (size_t)blockIdx.x * HITS_BLOCK_WORDS + threadIdx.x;
    uint32_t c = w < nwords ? (uint32_t)__builtin_popcountll(bits[w]) : 0u;
    __shared__ uint32_t part[4];
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0u) block_counts[blockIdx.x] = (unsigned long long)part[0] + part[1] + part[2] + part[3];
}

// exclusive scan in place; block_counts[nblocks] and *total receive the sum
__global__ void __launch_bounds__(256)
hit_block_scan_kernel(unsigned long long *__restrict__ block_counts, size_t nblocks, unsigned long long *__restrict__ total) {
    block_counts += (size_t)blockIdx.x * (nblocks + 2);  // one workgroup per segment
    total += blockIdx.x;
    __shared__ unsigned long long carry, sums[256];
    if (threadIdx.x == 0u) carry = 0;
    __syncthreads();
    for (size_t base = 0; base < nblocks; base += 256) {
        const size_t i = base + threadIdx.x;
        const unsigned long long v = i < nblocks ? block_counts[i] : 0ull;
        sums[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 256u; off <<= 1) {  // Hillis-Steele, inclusive
            const unsigned long long t = threadIdx.x >= off ? sums[threadIdx.x - off] : 0ull;
            __syncthreads();
            sums[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) block_counts[i] = carry + sums[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255u) carry += sums[255];
        __syncthreads();
    }
    if (threadIdx.x == 0u) {
        block_counts[nblocks] = carry;
        *total = carry;
    }
}

__global__ void __launch_bounds__(256)
hit_scatter_kernel(const uint64_t *__restrict__ bits, const ulonglong2 *__restrict__ sparse, size_t nwords, size_t map_words,
                   size_t n_searches, size_t nblocks, const unsigned long long *__restrict__ block_offsets,
                   ulonglong2 *__restrict__ hits, size_t cap) {
    bits += (size_t)blockIdx.y * map_words;
    sparse += (size_t)blockIdx.y * n_searches;
    block_offsets += (size_t)blockIdx.y * (nblocks + 2);
    hits += (size_t)blockIdx.y * cap * 2;
    const size_t w = (size_t)blockIdx.x * HITS_BLOCK_WORDS + threadIdx.x;
    uint64_t word = w < nwords ? bits[w] : 0ull;
    const uint32_t c = (uint32_t)__builtin_popcountll(word);
    // exclusive prefix of c over the block: within the wave by shuffles, across the four waves through LDS
    uint32_t incl = c;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __shared__ uint32_t wave_tot[4];
    if (lane == 63u) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (uint32_t v = 0; v < (threadIdx.x >> 6); ++v) before += wave_tot[v];
    unsigned long long at = block_offsets[blockIdx.x] + before;
    while (word) {
        const uint32_t b = (uint32_t)__builtin_ctzll(word);
        word &= word - 1ull;
        const unsigned long long index = (unsigned long long)w * 64ull + b;
        if (at < cap) {
            hits[2 * at] = sparse[index];
            hits[2 * at + 1] = make_ulonglong2(index, 0ull);
        }
        ++at;
    }
}

// ---- interval pairs for the wire: 10 bytes instead of 16 ----------------------------------------
// Every interval findInterval leaves -- empty ones, the (1, 0) of an invalid k-mer and the reference's
// (0, 2^64 - 1) corner included -- satisfies upper = lower + width - 1 (mod 2^64) with lower < 2^40 and
// 0 <= width < 2^40 (Occ is monotone, so an empty interval is always (lower, lower - 1)).  {lower:40,
// width:40} therefore carries a pair exactly; the gather of a batch's pairs to the root GPU moves 5/8
// of the bytes (xGMI links run ~77 GB/s per direction: 1.28 GB of pairs per peer and batch would take
// longer than the search that produced them).  Four pairs (64 B) <-> ten dwords per thread.
__global__ void __launch_bounds__(256)
pack_pairs10_kernel(const ulonglong2 *__restrict__ pairs, size_t n, uint32_t *__restrict__ out, uint32_t *__restrict__ unfit) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = 4 * t;
    uint64_t f[8];  // lo0, w0, lo1, w1, ...  (a thread past the end packs zeros and writes nothing)
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ulonglong2 p = make_ulonglong2(0, 0);
        if (i0 + j < n) p = pairs[i0 + j];
        const uint64_t w = p.y - p.x + 1ull;
        bad |= i0 + j < n && ((p.x >> 40) != 0ull || (w >> 40) != 0ull);
        f[2 * j] = p.x & 0xFFFFFFFFFFull;
        f[2 * j + 1] = w & 0xFFFFFFFFFFull;
    }
    if (bad && unfit) atomicAdd(unfit, 1u);
    // eight 40-bit fields = 320 bits = ten dwords, little endian, field k at bit 40 k
    uint32_t d[10];
#pragma unroll
    for (int w = 0; w < 10; ++w) {
        const int bit = 32 * w, k0 = bit / 40, off = bit - 40 * k0;  // dword w starts inside field k0
        uint64_t v = f[k0] >> off;
        if (off > 8 && k0 + 1 < 8) v |= f[k0 + 1] << (40 - off);
        d[w] = (uint32_t)v;
    }
    // through LDS, so that the block's 2,560 dwords leave as ten runs of 256 consecutive ones instead of
    // 256 interleaved 40-byte pieces
    __shared__ uint32_t turn[2560];
#pragma unroll
    for (int w = 0; w < 10; ++w) turn[10 * threadIdx.x + w] = d[w];
    __syncthreads();
    const size_t block0 = (size_t)blockIdx.x * 2560;               // first dword of this block's records
    const size_t total = (n * 10 + 3) / 4;                          // dwords the whole buffer holds
#pragma unroll
    for (int w = 0; w < 10; ++w) {
        const size_t at = block0 + 256u * w + threadIdx.x;
        if (at < total) out[at] = turn[256 * w + threadIdx.x];
    }
}

__global__ void __launch_bounds__(256)
unpack_pairs10_kernel(const uint32_t *__restrict__ in, size_t n, ulonglong2 *__restrict__ pairs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = 4 * t;
    __shared__ uint32_t turn[2560];
    const size_t block0 = (size_t)blockIdx.x * 2560, total = (n * 10 + 3) / 4;
#pragma unroll
    for (int w = 0; w < 10; ++w) {  // the block's dwords in ten coalesced runs
        const size_t at = block0 + 256u * w + threadIdx.x;
        turn[256 * w + threadIdx.x] = at < total ? in[at] : 0u;
    }
    __syncthreads();
    uint32_t d[10];
#pragma unroll
    for (int w = 0; w < 10; ++w) d[w] = turn[10 * threadIdx.x + w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint64_t f[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int bit = 40 * (2 * j + h), w0 = bit / 32, off = bit - 32 * w0;  // off is 0, 8, 16 or 24
            const uint64_t lo64 = (uint64_t)d[w0] | ((uint64_t)d[w0 + 1 < 10 ? w0 + 1 : 9] << 32);
            f[h] = (lo64 >> off) & 0xFFFFFFFFFFull;
        }
        if (i0 + j < n) pairs[i0 + j] = make_ulonglong2(f[0], f[0] + f[1] - 1ull);
    }
}

// query / query_exactmatch (query.cpp:87-120) over the extracted rows of a batch of k-mers: row i
// belongs to k-mer owner[i]; flags[i] = 1 when the read equals the k-mer (exact match: the whole read
// is the query, query.cpp:112-116).
__global__ void __launch_bounds__(256)
match_reads_kernel(const uint8_t *__restrict__ reads, const uint32_t *__restrict__ len, size_t n, uint32_t stride,
                   const uint32_t *__restrict__ owner, const uint8_t *__restrict__ kmers, uint32_t k, size_t kstride,
                   uint8_t *__restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool eq = len[i] == k;
    const uint8_t *r = reads + i * (size_t)stride, *w = kmers + (size_t)owner[i] * kstride;
    for (uint32_t t = 0; eq && t < k; ++t) eq = r[t] == w[t];
    flags[i] = eq ? 1 : 0;
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
static inline unsigned blocks256(size_t n) { return (unsigned)((n + 255) / 256); }

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

hipError_t launch_pack_var(const void *d_text, const void *d_off, size_t Q, uint32_t wpq, void *d_packed, void *d_valid, void *d_len,
                           hipStream_t stream) {
    if (Q == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_var_kernel, dim3(grid_for(64, Q, 16384)), dim3(64), 0, stream, (const uint8_t *)d_text, (const uint64_t *)d_off, Q,
                       wpq, (uint64_t *)d_packed, (uint8_t *)d_valid, (uint32_t *)d_len);
    return hipGetLastError();
}

// ---- scratch_cache (kernels.h) ------------------------------------------------------------------------
hipError_t scratch_cache::take(size_t bytes, hipStream_t stream, lease *out) {
    // sizes step by an eighth of a power of two: batches of varying size settle on a few buffers
    size_t unit = 4096;
    while (unit * 8 < bytes) unit <<= 1;
    const size_t want = (bytes + unit - 1) / unit * unit;
    std::lock_guard<std::mutex> lock(mu_);
    auto idle = [&](slot_t &s) {  // no launch uses it any more, as far as `stream` can tell
        if (s.busy) return false;
        if (!s.recorded || s.last == stream) return true;
        return hipEventQuery(s.done) == hipSuccess;
    };
    int pick = -1, grow = -1, fresh = -1, wait = -1;
    for (int i = 0; i < SLOTS; ++i) {
        slot_t &s = slots_[i];
        if (!s.p) { if (fresh < 0 && !s.busy) fresh = i; continue; }
        if (s.busy) continue;
        if (idle(s)) {
            if (s.bytes >= want) { if (pick < 0 || s.bytes < slots_[pick].bytes) pick = i; }
            else if (grow < 0) grow = i;
        } else if (s.bytes >= want && wait < 0) {
            wait = i;
        }
    }
    if (pick < 0 && fresh >= 0) {
        slot_t &s = slots_[fresh];
        hipError_t e = hipSuccess;
        if (!s.done) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(&s.p, want);
        if (e != hipSuccess) { s.p = nullptr; return e; }
        s.bytes = want;
        s.recorded = false;
        pick = fresh;
    }
    if (pick < 0 && grow >= 0) {  // an idle buffer that is too small is replaced
        slot_t &s = slots_[grow];
        // by one at least twice its size (hipFree waits for the whole device: batches that creep up in
        // size, as a service's windows do, should not pay that at every step)
        const size_t bigger = want > 2 * s.bytes ? want : 2 * s.bytes;
        (void)hipFree(s.p);
        s.p = nullptr;
        s.bytes = 0;
        hipError_t e = hipMalloc(&s.p, bigger);
        if (e == hipSuccess) {
            s.bytes = bigger;
        } else {
            (void)hipGetLastError();
            s.p = nullptr;
            e = hipMalloc(&s.p, want);
            if (e != hipSuccess) { s.p = nullptr; return e; }
            s.bytes = want;
        }
        s.recorded = false;
        pick = grow;
    }
    if (pick < 0 && wait >= 0) {  // every buffer is in some stream's queue: line up behind one
        const hipError_t e = hipStreamWaitEvent(stream, slots_[wait].done, 0);
        if (e != hipSuccess) return e;
        pick = wait;
    }
    if (pick < 0) return hipErrorOutOfMemory;  // 16 launch sequences being enqueued at once, none fitting
    slots_[pick].busy = true;
    out->p = slots_[pick].p;
    out->slot = pick;
    return hipSuccess;
}

void scratch_cache::give(const lease &l, hipStream_t stream) {
    if (l.slot < 0) return;
    std::lock_guard<std::mutex> lock(mu_);
    slot_t &s = slots_[l.slot];
    s.recorded = hipEventRecord(s.done, stream) == hipSuccess;
    if (!s.recorded) (void)hipStreamSynchronize(stream);  // no event: the buffer is idle only once the stream is
    s.last = stream;
    s.busy = false;
}

void scratch_cache::destroy() {
    std::lock_guard<std::mutex> lock(mu_);
    for (slot_t &s : slots_) {
        if (s.p) (void)hipFree(s.p);
        if (s.done) (void)hipEventDestroy(s.done);
        s = slot_t();
    }
}

size_t scratch_cache::held_bytes() {
    std::lock_guard<std::mutex> lock(mu_);
    size_t b = 0;
    for (slot_t &s : slots_) b += s.bytes;
    return b;
}

// Fills `d_entries` (4^T entries, `stride` apart) by searching every T-mer, in slices that bound the
// temporary memory.  `view` must not have a table yet.
hipError_t build_ktable(const shard_view &view, uint32_t T, uint64_t *d_entries, uint32_t stride, int num_cus,
                        hipStream_t stream, uint32_t fmt, uint64_t *untabulated) {
    const uint64_t total = 1ull << (2u * T);
    const size_t SL = (size_t)std::min<uint64_t>(total, 1ull << 24);
    uint64_t *d_pk = nullptr, *d_lo = nullptr, *d_up = nullptr, *d_half = nullptr;
    unsigned long long *d_left = nullptr;
    if (untabulated) *untabulated = 0;
    if (fmt == KTAB_GROUPED && T < 2u) return hipErrorInvalidValue;
    uint8_t *d_ok = nullptr;
    shard_view *d_view = nullptr;
    shard_view plain = view;
    plain.ktab = nullptr;
    plain.ktab_depth = 0;
    plain.ktab_stride = 1;
    plain.ktab_fmt = KTAB_PLAIN;
    hipError_t e = hipSuccess;
    scratch_cache scratch;
    auto cleanup = [&] {
        scratch.destroy();
        if (d_pk) (void)hipFree(d_pk);
        if (d_lo) (void)hipFree(d_lo);
        if (d_up) (void)hipFree(d_up);
        if (d_ok) (void)hipFree(d_ok);
        if (d_half) (void)hipFree(d_half);
        if (d_view) (void)hipFree(d_view);
        if (d_left) (void)hipFree(d_left);
    };
    if ((e = hipMalloc(&d_pk, SL * 8)) != hipSuccess || (e = hipMalloc(&d_lo, SL * 8)) != hipSuccess ||
        (e = hipMalloc(&d_up, SL * 8)) != hipSuccess || (e = hipMalloc(&d_ok, SL)) != hipSuccess ||
        (e = hipMalloc(&d_view, sizeof(shard_view))) != hipSuccess || (e = hipMalloc(&d_left, 8)) != hipSuccess ||
        (e = hipMemsetAsync(d_left, 0, 8, stream)) != hipSuccess) {
        cleanup();
        return e;
    }
    // a deep table is built on top of a shallow one: with the (T/2)-mer table in place, each of
    // the 4^T searches starts from its last T/2 symbols' entry and takes half the LF steps
    if (T >= 10u) {
        const uint32_t T0 = T / 2u;
        if ((e = hipMalloc(&d_half, 8ull << (2u * T0))) == hipSuccess) e = build_ktable(view, T0, d_half, 1, num_cus, stream);
        if (e != hipSuccess) {
            cleanup();
            return e;
        }
        plain.ktab = d_half;
        plain.ktab_depth = T0;
        plain.ktab_fmt = KTAB_PLAIN;
    }
    if ((e = hipMemcpy(d_view, &plain, sizeof plain, hipMemcpyHostToDevice)) != hipSuccess) {
        cleanup();
        return e;
    }
    for (uint64_t base = 0; base < total && e == hipSuccess; base += SL) {
        const size_t m = (size_t)std::min<uint64_t>(SL, total - base);
        if (fmt == KTAB_GROUPED)  // (a slice = whole groups: SL is a multiple of 4, base / 4 the slice's first group)
            hipLaunchKernelGGL(ktab_group_codes_kernel, dim3(blocks256(m)), dim3(256), 0, stream, base >> 2, 2u * (T - 1u), m, d_pk, d_ok);
        else
            hipLaunchKernelGGL(ktab_codes_kernel, dim3(blocks256(m)), dim3(256), 0, stream, base, m, d_pk, d_ok);
        search_extra role;
        role.table_build = true;
        e = launch_search(scratch, d_view, 1, d_pk, d_ok, m, T, d_lo, d_up, false, nullptr, num_cus, stream, nullptr, nullptr, &role);
        if (e != hipSuccess) break;
        if (fmt == KTAB_GROUPED)
            hipLaunchKernelGGL(ktab_group_encode_kernel, dim3(blocks256(m >> 2)), dim3(256), 0, stream, d_lo, d_up, m >> 2,
                               reinterpret_cast<uint32_t *>(d_entries) + (base >> 2) * stride * 3u, stride, d_left);
        else
            hipLaunchKernelGGL(ktab_encode_kernel, dim3(blocks256(m)), dim3(256), 0, stream, d_lo, d_up, m,
                               d_entries + base * stride, stride);
        e = hipGetLastError();
    }
    if (e == hipSuccess && untabulated) e = hipMemcpyAsync(untabulated, d_left, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    cleanup();
    return e;
}

hipError_t launch_occ_batch(const shard_view &ix, const void *d_syms, const void *d_index, size_t n,
                            void *d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(occ_batch_kernel, dim3(blocks256(n)), dim3(256), 0, stream, ix,
                       (const uint8_t *)d_syms, (const uint64_t *)d_index, n, (uint64_t *)d_out);
    return hipGetLastError();
}

hipError_t launch_char_batch(const shard_view &ix, const void *d_index, size_t n, void *d_out,
                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(char_batch_kernel, dim3(blocks256(n)), dim3(256), 0, stream, ix,
                       (const uint64_t *)d_index, n, (uint8_t *)d_out);
    return hipGetLastError();
}

hipError_t launch_occ_at_batch(const shard_view &ix, const void *d_syms, const void *d_bc,
                               size_t n, void *d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(occ_at_batch_kernel, dim3(blocks256(n)), dim3(256), 0, stream, ix, ix.sel, ix.sel_stride,
                       (const uint8_t *)d_syms, (const uint64_t *)d_bc, n, (uint64_t *)d_out);
    return hipGetLastError();
}

size_t compact_hits_block_words(size_t n_searches) {
    const size_t nwords = (n_searches + 63) / 64;
    return (nwords + HITS_BLOCK_WORDS - 1) / HITS_BLOCK_WORDS + 2;
}

hipError_t launch_compact_hits(const void *d_bits, const void *d_sparse, size_t n_searches, void *d_hits, size_t cap,
                               void *d_total, void *d_block_counts, hipStream_t stream, uint32_t nseg) {
    const size_t nwords = (n_searches + 63) / 64, nblocks = (nwords + HITS_BLOCK_WORDS - 1) / HITS_BLOCK_WORDS;
    if (nseg == 0) return hipSuccess;
    if (nwords == 0) return hipMemsetAsync(d_total, 0, 8 * (size_t)nseg, stream);
    const size_t map_words = hit_map_words(n_searches);
    hipLaunchKernelGGL(hit_block_counts_kernel, dim3((unsigned)nblocks, nseg), dim3(256), 0, stream, (const uint64_t *)d_bits, nwords,
                       map_words, nblocks, (unsigned long long *)d_block_counts);
    hipLaunchKernelGGL(hit_block_scan_kernel, dim3(nseg), dim3(256), 0, stream, (unsigned long long *)d_block_counts, nblocks,
                       (unsigned long long *)d_total);
    hipLaunchKernelGGL(hit_scatter_kernel, dim3((unsigned)nblocks, nseg), dim3(256), 0, stream, (const uint64_t *)d_bits,
                       (const ulonglong2 *)d_sparse, nwords, map_words, n_searches, nblocks,
                       (const unsigned long long *)d_block_counts, (ulonglong2 *)d_hits, cap);
    return hipGetLastError();
}

// Extracted reads for the wire: [n][stride] ASCII bytes + lengths -> [n][stride / 4] bytes, 2 bits per base (A, C, G, T
// = 0..3, base i at bits 2 * (i % 4) of byte i / 4), zeros past a read's end (and for a read marked as not fitting).
// One thread per output dword = 16 bases; stride is a multiple of 16.
__global__ void __launch_bounds__(256)
pack_reads2_kernel(const uint8_t *__restrict__ reads, const uint32_t *__restrict__ len, size_t n, uint32_t stride,
                   uint32_t *__restrict__ out) {
    const uint32_t per = stride / 16u;  // dwords per read
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * per) return;
    const size_t r = i / per;
    const uint32_t d = (uint32_t)(i - r * per);
    uint32_t ln = len[r];
    if (ln == 0xFFFFFFFFu) ln = 0;
    const uint4 v = *reinterpret_cast<const uint4 *>(reads + r * (size_t)stride + 16u * d);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t code = ((w[q] >> 1) ^ (w[q] >> 2)) & 0x03030303u;  // A 0, C 1, G 2, T 3 in every byte
        const uint32_t c8 = (code | (code >> 6) | (code >> 12) | (code >> 18)) & 0xFFu;
        o |= c8 << (8 * q);
    }
    const uint32_t base0 = 16u * d;  // first base this dword holds
    const uint32_t keep = ln <= base0 ? 0u : ln - base0 >= 16u ? 32u : 2u * (ln - base0);  // bits of it inside the read
    out[i] = keep >= 32u ? o : (o & ((1u << keep) - 1u));
}

__global__ void __launch_bounds__(256)
unpack_reads2_kernel(const uint32_t *__restrict__ in, const uint32_t *__restrict__ len, size_t n, uint32_t stride,
                     uint8_t *__restrict__ reads) {
    const uint32_t per = stride / 16u;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * per) return;
    const size_t r = i / per;
    const uint32_t d = (uint32_t)(i - r * per);
    uint32_t ln = len[r];
    if (ln == 0xFFFFFFFFu) ln = 0;
    const uint32_t x = in[i];
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t o = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t pos = 16u * d + 4u * q + b;
            const uint32_t c = (x >> (8 * q + 2 * b)) & 3u;
            const uint32_t ch = pos < ln ? ((0x54474341u >> (8u * c)) & 0xFFu) : 0u;  // "ACGT"[c], NUL past the end
            o |= ch << (8 * b);
        }
        w[q] = o;
    }
    *reinterpret_cast<uint4 *>(reads + r * (size_t)stride + 16u * d) = make_uint4(w[0], w[1], w[2], w[3]);
}

hipError_t launch_pack_reads2(const void *d_reads, const void *d_len, size_t n, uint32_t stride, void *d_packed, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_reads2_kernel, dim3(blocks256(n * (stride / 16u))), dim3(256), 0, stream, (const uint8_t *)d_reads,
                       (const uint32_t *)d_len, n, stride, (uint32_t *)d_packed);
    return hipGetLastError();
}
hipError_t launch_unpack_reads2(const void *d_packed, const void *d_len, size_t n, uint32_t stride, void *d_reads, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_reads2_kernel, dim3(blocks256(n * (stride / 16u))), dim3(256), 0, stream, (const uint32_t *)d_packed,
                       (const uint32_t *)d_len, n, stride, (uint8_t *)d_reads);
    return hipGetLastError();
}

hipError_t launch_pack_pairs10(const void *d_pairs, size_t n, void *d_packed, void *d_unfit, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_pairs10_kernel, dim3(blocks256((n + 3) / 4)), dim3(256), 0, stream, (const ulonglong2 *)d_pairs, n,
                       (uint32_t *)d_packed, (uint32_t *)d_unfit);
    return hipGetLastError();
}

hipError_t launch_unpack_pairs10(const void *d_packed, size_t n, void *d_pairs, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_pairs10_kernel, dim3(blocks256((n + 3) / 4)), dim3(256), 0, stream, (const uint32_t *)d_packed, n,
                       (ulonglong2 *)d_pairs);
    return hipGetLastError();
}

hipError_t launch_variants(const void *d_packed, const void *d_valid, size_t Q, uint32_t k, void *d_vpacked,
                           void *d_vvalid, hipStream_t stream) {
    if (Q == 0) return hipSuccess;
    const uint32_t wpq = (k + 31u) / 32u ? (k + 31u) / 32u : 1u;
    const size_t total = Q * (3 * (size_t)k + 1);
    hipLaunchKernelGGL(variants_kernel, dim3(blocks256(total)), dim3(256), 0, stream,
                       (const uint64_t *)d_packed, (const uint8_t *)d_valid, Q, k, wpq, (uint64_t *)d_vpacked,
                       (uint8_t *)d_vvalid);
    return hipGetLastError();
}

// Test hook: the device's position -> window division (rank_device.h, fast_window) on caller-supplied
// positions, so that a test can hold it to integer division over the whole 40-bit range for every span.
__global__ void __launch_bounds__(256)
debug_fast_window_kernel(const uint64_t *__restrict__ p, size_t n, span_params sp, uint32_t *__restrict__ w,
                         uint32_t *__restrict__ r) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t pin;
    w[i] = fast_window(p[i], sp.S, sp.inv, pin);
    r[i] = pin;
}
hipError_t launch_debug_fast_window(const void *d_p, size_t n, uint32_t S, void *d_w, void *d_r, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(debug_fast_window_kernel, dim3(blocks256(n)), dim3(256), 0, stream, (const uint64_t *)d_p, n,
                       make_span(S), (uint32_t *)d_w, (uint32_t *)d_r);
    return hipGetLastError();
}

uint64_t select_sample_stride(const shard_view &ix) { return select_stride(ix); }

hipError_t launch_select_samples(const shard_view &ix, uint64_t *d_sel, hipStream_t stream) {
    if (ix.nwin == 0) return hipSuccess;
    hipLaunchKernelGGL(select_sample_kernel, dim3(blocks256(ix.nwin)), dim3(256), 0, stream, ix, d_sel,
                       select_sample_stride(ix));
    return hipGetLastError();
}

hipError_t launch_psi_hints(const shard_view &ix, unsigned long long *d_made, hipStream_t stream) {
    if (ix.nwin == 0 || !ix.sel) return hipSuccess;
    hipLaunchKernelGGL(psi_hint_kernel, dim3(blocks256(ix.nwin)), dim3(256), 0, stream, ix, ix.sel, ix.sel_stride,
                       const_cast<uint32_t *>(ix.lines), d_made);
    return hipGetLastError();
}

hipError_t launch_match_reads(const void *d_reads, const void *d_len, size_t n, uint32_t stride, const void *d_owner,
                              const void *d_kmers, uint32_t k, size_t kstride, void *d_flags, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(match_reads_kernel, dim3(blocks256(n)), dim3(256), 0, stream, (const uint8_t *)d_reads,
                       (const uint32_t *)d_len, n, stride, (const uint32_t *)d_owner, (const uint8_t *)d_kmers, k, kstride,
                       (uint8_t *)d_flags);
    return hipGetLastError();
}

hipError_t launch_synth_runs(void *d_runs, uint64_t num_runs, uint64_t seed, hipStream_t stream, uint64_t first) {
    if (num_runs == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_runs_kernel, dim3(grid_for(256 * 16, num_runs, 16384)), dim3(256), 0,
                       stream, (uint8_t *)d_runs, num_runs, seed, first);
    return hipGetLastError();
}

hipError_t launch_sample_present(const shard_view &ix, size_t Q, uint32_t k, size_t stride,
                                 uint64_t seed, void *d_kmers, hipStream_t stream) {
    if (Q == 0 || k == 0) return hipSuccess;
    hipLaunchKernelGGL(sample_present_kernel, dim3(blocks256(Q)), dim3(256), 0, stream, ix,
                       Q, k, stride, seed, (uint8_t *)d_kmers);
    return hipGetLastError();
}

}  // namespace rsb

// wave_lines.h -- what the wave-cooperative kernels share (search_lines.hip, extract_lines.hip): the
// wave's LDS stage, the direct-to-LDS fetch of one window line per lane, and accessors into a lane's
// staged line.
#ifndef RSBWT_WAVE_LINES_H
#define RSBWT_WAVE_LINES_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "line_format.h"
#include "rank_device.h"

namespace rsb {

// Cache policy of the line fetch: nt (aux = 2).  A fetched line is parsed once and not asked for
// again, and with the default policy the stream of random lines churns the vector L1 and the L2: same
// box, same batch, 8 x 20 GB shards: 5.33-5.40 ms per launch with nt against 5.67-5.78 ms with the
// default policy (4 shards; sc1 and sc0 variants in between; nt on the result stores is slower).
#ifndef RSB_LINE_LOAD_AUX  // tuning knob (tools/build_variant.sh): 0 default, 1 sc0, 2 nt, 16 sc1
#define RSB_LINE_LOAD_AUX 2
#endif

// LDS stage: 128 B per lane, 8 KB per wave, 32 KB per 4-wave workgroup, so 5 workgroups
// (20 waves) would fit a CU's 160 KB; 4 are launched.
constexpr int SLOT_U4 = 8;
constexpr int WG_WAVES = 4;  // waves per workgroup (one-wave groups would pack 17 per CU but measured 1.4x slower)
#ifndef RSB_MIN_WGS_PER_CU  // tuning knob (tools/build_variant.sh): register budget = 512 / this many waves per SIMD
#define RSB_MIN_WGS_PER_CU 4
#endif

// The stage is written by LDS-DMA and parsed as dwords / 8- / 16-byte pieces: the read types may
// alias anything, or type-based alias analysis lets hipcc reuse values read before a re-fetch.
typedef uint32_t __attribute__((may_alias)) lds_u32;
typedef uint2 __attribute__((may_alias)) lds_u2;
typedef uint4 __attribute__((may_alias)) lds_u4;
typedef __attribute__((address_space(3))) void *lds_void_ptr;
typedef const __attribute__((address_space(1))) void *global_void_ptr;

// Fetch of up to 64 lines into the wave's LDS stage, direct to LDS (global_load_lds_dwordx4,
// gfx950): no register round trip, no ds_write pass.  One instruction writes 1 KB of LDS in lane
// order, so the work is split the way that makes this the stage layout itself: instruction k
// (0..7) serves lanes T = 8o + k, the eight lanes of octet o each bringing 16 B of the line
// lane T wants (a full 128-B line per octet: the request shape tools/gather_bench.hip measures
// fastest).  Lane T's line then sits at k * 1 KB + o * 128 B, chunk c at position c ^ k -- the
// swizzle is applied on the SOURCE side (lane l of the octet loads chunk (l & 7) ^ k).
// want == ~0u: that lane needs nothing (its octet's lanes are masked off for that instruction and
// the row keeps what it held).
__device__ __forceinline__ void glds_fetch(const char *lines, uint32_t want, uint32_t lane, uint32_t stage_lds) {
    uint32_t tb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        tb[k] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & ~7u) + k) << 2), (int)want);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (tb[k] != ~0u) {
            const char *src = lines + (uint64_t)tb[k] * 128u + (((lane & 7u) ^ (uint32_t)k) << 4);
            __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)(uintptr_t)(stage_lds + k * 1024u), 16, 0, RSB_LINE_LOAD_AUX);
        }
    }
}
// the lines are in LDS once every outstanding load has returned
// (the builtin, not inline asm: hipcc's wait-count bookkeeping then knows nothing is outstanding and
// does not add its own vmcnt(0) at the head of the next pass, in front of that pass's loads)
__device__ __forceinline__ void glds_wait() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt left alone (gfx9 encoding)
    asm volatile("" ::: "memory");
}


// A lane's staged line: dword d of the line lane T asked for sits at chunk (d >> 2) ^ (T & 7) of the
// row (T & 7) * 1 KB + (T >> 3) * 128 B of its wave's stage (glds_fetch); lanes l and l + 32 share
// the swizzle, so the upper side of a pair can read the lower side's row (row - 512 B).
struct staged_line {
    const lds_u32 *row;
    uint32_t swz;
    __device__ __forceinline__ const lds_u32 *at(uint32_t d) const { return row + ((((d) >> 2) ^ swz) << 2) + ((d) & 3u); }
    __device__ __forceinline__ uint32_t dword(uint32_t d) const { return *at(d); }
    __device__ __forceinline__ uint2 u2(uint32_t d) const { return *reinterpret_cast<const lds_u2 *>(at(d)); }  // d even
    __device__ __forceinline__ uint4 u4(uint32_t d) const { return *reinterpret_cast<const lds_u4 *>(at(d)); }  // d % 4 == 0
};

__device__ __forceinline__ const lds_u32 *own_stage_row(const uint4 *stage, uint32_t lane) {
    return reinterpret_cast<const lds_u32 *>(stage + (lane & 7u) * 64u + (lane >> 3) * SLOT_U4);
}

// what the 24 pieces at dwords qd .. qd+5 hold of the table's symbol (rank_device.h, sym_tab: the 0/1 match mask of a
// dword is one v_perm_b32), 4 runs per v_dot4_u32_u8
__device__ __forceinline__ uint32_t matched24(const staged_line &L, uint32_t qd, const sym_tab &t) {
    const uint2 x0 = L.u2(qd & 31u), x1 = L.u2((qd + 2u) & 31u), x2 = L.u2((qd + 4u) & 31u);
    const uint32_t e[6] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y};
    return matched24_tab(e, t);
}
__device__ __forceinline__ uint32_t matched24(const staged_line &L, uint32_t qd, uint32_t b) {
    return matched24(L, qd, make_sym_tab(b));
}

// the 24 pieces at dword dw into registers
__device__ __forceinline__ void load24(const staged_line &L, uint32_t dw, uint32_t r6[6]) {
    const uint2 y0 = L.u2(dw & 31u), y1 = L.u2((dw + 2u) & 31u), y2 = L.u2((dw + 4u) & 31u);
    r6[0] = y0.x; r6[1] = y0.y; r6[2] = y1.x; r6[3] = y1.y; r6[4] = y2.x; r6[5] = y2.y;
}

// the header fields a lane-private parse needs
struct line_head {
    uint32_t s1, s2, s3, span, kind;
};
__device__ __forceinline__ line_head read_head(const staged_line &L) {
    const uint4 h0 = L.u4(0);
    const uint32_t m0 = h0.y >> 8, m1 = h0.w >> 8;
    line_head h;
    h.s1 = m0 & 0x3FFu;
    h.s2 = (m0 >> 10) & 0x7FFu;
    h.s3 = h.s2 + (m1 & 0x3FFu);
    h.span = h.s3 + ((m1 >> 10) & 0x3FFu);
    h.kind = (m1 >> 20) & 3u;
    return h;
}
__device__ __forceinline__ uint64_t read_count(const staged_line &L, uint32_t b) {  // b = 1..4
    const uint2 cw = L.u2(2u * (b - 1u));
    return ((uint64_t)(cw.y & 0xFFu) << 32) | cw.x;
}
__device__ __forceinline__ uint32_t read_half(const staged_line &L, uint32_t b) {
    const uint32_t hm = L.dword(5u + 2u * ((b - 1u) >> 1)) >> 8;
    return (hm >> (11u * ((b - 1u) & 1u))) & 0x7FFu;
}
__device__ __forceinline__ uint32_t read_chunk_dword(const staged_line &L) {
    const uint32_t m2 = L.dword(5) >> 8, m3 = L.dword(7) >> 8;
    return 2u * (((m2 >> 22) & 3u) | (((m3 >> 22) & 3u) << 2));
}

// ---- k-mer table lookups shared by the start-record kernels, the one-lane kernel and the worklist pre-pass
__device__ __forceinline__ bool view_uses_ktab(const shard_view &ix, uint32_t k) {
    return ix.ktab != nullptr && ix.ktab_depth >= 2u && k >= ix.ktab_depth;
}

// Entry of T-mer `code` in the plain table's 8-byte form, whatever the table's format (line_format.h)
__device__ __forceinline__ uint64_t ktab_entry(const uint64_t *__restrict__ ktab, uint32_t fmt, uint32_t T, uint32_t stride, uint64_t code) {
    if (fmt == KTAB_GROUPED) {
        const uint32_t gbits = 2u * (T - 1u);
        const uint64_t g = code & ((1ull << gbits) - 1ull);
        const uint32_t *r = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(ktab) + g * stride * (uint64_t)KTAB_GROUP_BYTES);
        return ktab_group_entry(r[0], r[1], r[2], (uint32_t)(code >> gbits) & 3u);
    }
    return ktab[code * stride];
}

// ---- the 1-mismatch worklists' shared pieces (mm1_worklist.hip, search_solo.h)
// Worklist record (32 B): x = lower (40 bits) | next symbol j << 40 (16 bits) | WL_DEAD << 63;  y = upper;
//                         z = canonical search index q * (3k+1) + v;  w = the variant's packed word (k <= 32).
__device__ __forceinline__ void wl_store(ulonglong2 *wl, size_t slot, uint64_t lo, uint32_t j, uint64_t hi, uint64_t canon, uint64_t word) {
    wl[2u * slot] = make_ulonglong2((lo & COUNT_MASK) | ((uint64_t)(j & 0xFFFFu) << COUNT_BITS), hi);
    wl[2u * slot + 1u] = make_ulonglong2(canon, word);
}


// Occ of the THREE bases other than `orig` (0..3 = A..T) up to offset o (1-based, within the line's own pieces:
// o <= span) of a staged line: out[d] for the d-th base of ACGT without the original one.  One look at the quarter's
// 24 pieces for all of them (rank_device.h, char_rank24, taken apart): what does not depend on the symbol -- the
// pieces' lengths and symbols, the dword and the piece holding the position, the lengths that lie BEFORE it (every
// other length masked to zero: a piece of no length counts for no symbol) -- is computed once; a base then costs
// four instructions per dword: its match mask with the matching bytes at 0x80, and a v_dot4 against the masked
// lengths, the sum shifted down by 7 once.  (The first version ran the whole of char_rank24's count per base and
// position, three times over in the kernel: 2,195 VALU instructions per pass, which bound the launch.)
__device__ __forceinline__ void staged_occ_alts(const staged_line &L, const line_head &h, uint32_t o, uint32_t orig, uint64_t out[3]) {
    const uint32_t cq = (o > h.s1 ? 1u : 0u) + (o > h.s2 ? 1u : 0u) + (o > h.s3 ? 1u : 0u);
    const uint32_t start = cq == 0u ? 0u : cq == 1u ? h.s1 : cq == 2u ? h.s2 : h.s3;
    const uint32_t rem = o - start;  // >= 1
    // the earlier quarter of the position's half, added whole when cq is odd -- first, so that its registers are free again
    uint32_t me[3];
    {
        uint32_t e[6], le[6], se[6];
        load24(L, HDR_DWORDS + 6u * (cq & 2u), e);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            le[i] = e[i] & 0x1F1F1F1Fu;
            se[i] = (e[i] >> 5) & 0x07070707u;
        }
#pragma unroll
        for (uint32_t d = 0; d < 3u; ++d) {
            const uint32_t bb = splat_byte((d < orig ? d : d + 1u) + 1u);
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) m = __builtin_amdgcn_udot4(le[i], (0x80808080u - (se[i] ^ bb)) & 0x80808080u, m, false);
            me[d] = (cq & 1u) ? (m >> 7) : 0u;
        }
    }
    uint32_t r[6];  // the quarter holding the position
    load24(L, HDR_DWORDS + 6u * cq, r);
    uint32_t lr[6], sr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        lr[i] = r[i] & 0x1F1F1F1Fu;
        sr[i] = (r[i] >> 5) & 0x07070707u;
    }
    uint32_t cum[6];
    cum[0] = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) cum[i + 1] = __builtin_amdgcn_udot4(lr[i], 0x01010101u, cum[i], false);
    uint32_t x = r[0], base = 0, di = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        const bool past = rem > cum[i];
        x = past ? r[i] : x;
        base = past ? cum[i] : base;
        di = past ? (uint32_t)i : di;
    }
    const uint32_t rd = rem - base;
    const uint32_t ps = (x & 0x1F1F1F1Fu) * 0x01010101u;
    const uint32_t jj = (rd > (ps & 0xFFu) ? 1u : 0u) + (rd > ((ps >> 8) & 0xFFu) ? 1u : 0u) + (rd > ((ps >> 16) & 0xFFu) ? 1u : 0u);
    const uint32_t here = (x >> (8u * jj + 5u)) & 7u;
    const uint32_t pj = jj ? __builtin_amdgcn_ubfe(ps, 8u * jj - 8u, 8u) : 0u;
    const uint32_t reach = rd <= (ps >> 24) ? rd - pj : (ps >> 24) - pj;
    // the lengths before the position: whole dwords before dword di, dword di's pieces before piece jj, nothing after
    const uint32_t inner = (1u << (8u * jj)) - 1u;
#pragma unroll
    for (uint32_t i = 0; i < 6u; ++i) lr[i] = i < di ? lr[i] : (i == di ? lr[i] & inner : 0u);
#pragma unroll
    for (uint32_t d = 0; d < 3u; ++d) {
        const uint32_t b = (d < orig ? d : d + 1u) + 1u, bb = splat_byte(b);
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc = __builtin_amdgcn_udot4(lr[i], (0x80808080u - (sr[i] ^ bb)) & 0x80808080u, acc, false);
        out[d] = read_count(L, b) + (cq >= 2u ? read_half(L, b) : 0u) + me[d] + (acc >> 7) + (here == b ? reach : 0u);
    }
}

}  // namespace rsb
#endif

// rank_device.h -- gfx950 device primitives shared by the search kernel: the run-by-run scan of
// RLEBWT::getOcc (src/bwt/rlebwt.cpp:281-298) on SDWA operands, whole-dword matched sums on
// v_dot4_u32_u8, and the position -> window division.
#ifndef RSBWT_RANK_DEVICE_H
#define RSBWT_RANK_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsb {

// matched symbols in one dword of 4 runs: acc + sum of len over the bytes whose symbol == b
// (bb = b in every byte), 4 runs per v_dot4_u32_u8
__device__ __forceinline__ uint32_t dword_matched(uint32_t x, uint32_t bb, uint32_t acc) {
    const uint32_t z = ((x >> 5) & 0x07070707u) ^ bb;             // 0 where the symbol matches
    const uint32_t m01 = ((0x80808080u - z) >> 7) & 0x01010101u;  // 1 where it matches
    return __builtin_amdgcn_udot4(x & 0x1F1F1F1Fu, m01, acc, false);
}

// Sum over ND dwords of runs of min(len, what is left of `rem` symbols), counting only runs of
// symbol b.  RLEBWT::getOcc's bucket scan (src/bwt/rlebwt.cpp:281-298), 4.5 VALU per run byte:
// SDWA operands pick the byte out of the pre-masked dwords.
template <int ND>
__device__ __forceinline__ uint32_t runs_scan(const uint32_t *r, uint32_t b, uint32_t rem) {
    const uint32_t b5 = b << 5;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const uint32_t l = r[i] & 0x1F1F1F1Fu;   // lengths
        const uint32_t sy = r[i] & 0xE0E0E0E0u;  // symbols << 5
        uint32_t t0, t1;
        asm("v_cmp_eq_u32_sdwa vcc, %[sy], %[b5] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
            "v_min_u32_sdwa %[t0], %[rem], %[l] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
            "v_sub_u32 %[rem], %[rem], %[t0]\n\t"
            "v_cndmask_b32 %[t0], 0, %[t0], vcc\n\t"
            "v_cmp_eq_u32_sdwa vcc, %[sy], %[b5] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_min_u32_sdwa %[t1], %[rem], %[l] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
            "v_sub_u32 %[rem], %[rem], %[t1]\n\t"
            "v_cndmask_b32 %[t1], 0, %[t1], vcc\n\t"
            "v_add3_u32 %[acc], %[acc], %[t0], %[t1]\n\t"
            "v_cmp_eq_u32_sdwa vcc, %[sy], %[b5] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
            "v_min_u32_sdwa %[t0], %[rem], %[l] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
            "v_sub_u32 %[rem], %[rem], %[t0]\n\t"
            "v_cndmask_b32 %[t0], 0, %[t0], vcc\n\t"
            "v_cmp_eq_u32_sdwa vcc, %[sy], %[b5] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
            "v_min_u32_sdwa %[t1], %[rem], %[l] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
            "v_sub_u32 %[rem], %[rem], %[t1]\n\t"
            "v_cndmask_b32 %[t1], 0, %[t1], vcc\n\t"
            "v_add3_u32 %[acc], %[acc], %[t0], %[t1]"
            : [acc] "+v"(acc), [rem] "+v"(rem), [t0] "=&v"(t0), [t1] "=&v"(t1)
            : [l] "v"(l), [sy] "v"(sy), [b5] "v"(b5)
            : "vcc");
    }
    return acc;
}

// ---- the search kernels' rank since round 5: dword by dword, not run by run.
// The search launches are bound by the instructions a SIMD issues (a wave of the headline launch
// issued ~600 instructions per pass, 400 of them VALU, at 4 waves per SIMD; + 12.6 % VALU cost + 5.7 % time), and the
// run-by-run scan above was 120 of them per lookup.  Here the 24 pieces' dword totals -- all symbols, and symbol b's --
// come from two v_dot4 each; the 0/1 match mask of a dword is ONE v_perm_b32: the pieces' symbols (0..4, three bits)
// are the byte selectors into an 8-byte table that holds 1 at byte b, so no compare, subtract or shift of a mask is
// needed (3 instructions per dword where dword_matched's mask takes 5); the dword holding the position is picked by
// five compare-and-selects on the running totals, and only ITS four pieces are scanned run by run: 76 VALU
// instructions per lookup, 30 for the whole-quarter sum an odd quarter needs.

// v_perm_b32 table of symbol b (1..4): byte s of {hi, lo} is 1 where s == b, 0 elsewhere (s = 0..7 selects it)
struct sym_tab {
    uint32_t lo, hi;
};
__device__ __forceinline__ sym_tab make_sym_tab(uint32_t b) {
    const uint64_t t = 1ull << (8u * b);
    sym_tab o;
    o.lo = (uint32_t)t;
    o.hi = (uint32_t)(t >> 32);
    return o;
}
// 0/1 per piece of the dword: its symbol is the table's
__device__ __forceinline__ uint32_t match01(uint32_t x, const sym_tab &t) {
    return __builtin_amdgcn_perm(t.hi, t.lo, (x >> 5) & 0x07070707u);
}
// what 24 pieces hold of the table's symbol
__device__ __forceinline__ uint32_t matched24_tab(const uint32_t e[6], const sym_tab &t) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) m = __builtin_amdgcn_udot4(e[i] & 0x1F1F1F1Fu, match01(e[i], t), m, false);
    return m;
}
// RLEBWT::getOcc's bucket scan (src/bwt/rlebwt.cpp:281-298) over 24 pieces: how many of their first `rem` symbols are
// b (t = make_sym_tab(b)); rem = 0 gives 0, rem beyond what the pieces hold gives all they hold of b
__device__ __forceinline__ uint32_t rank24(const uint32_t r[6], const sym_tab &t, uint32_t b, uint32_t rem) {
    uint32_t cum = 0, mat = 0, x = r[0], base = 0, mb = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (i) {  // the position lies past dwords 0..i-1: dword i is the candidate
            const bool past = rem > cum;
            x = past ? r[i] : x;
            base = past ? cum : base;
            mb = past ? mat : mb;
        }
        if (i < 5) {
            const uint32_t l = r[i] & 0x1F1F1F1Fu;
            cum = __builtin_amdgcn_udot4(l, 0x01010101u, cum, false);
            mat = __builtin_amdgcn_udot4(l, match01(r[i], t), mat, false);
        }
    }
    return mb + runs_scan<1>(&x, b, rem - base);
}

// ---- whole-dword forms for the walk kernels (extract_lines.hip), where a lane has to find out WHICH
// symbol sits at a position before it can rank it.  24 pieces = 6 dwords: dword totals come from
// v_dot4, the dword holding the position from five compares, and only that dword's four pieces are
// looked at one by one (prefix sums of its bytes = one multiply: 4 x 31 < 256).

// b in every byte
__device__ __forceinline__ uint32_t splat_byte(uint32_t b) { return __builtin_amdgcn_perm(0u, b, 0u); }

// RLEBWT::getChar + RLEBWT::getOcc of that symbol (rlebwt.cpp:202-227,268-301) within 24 pieces: c =
// rank (0..4) of the symbol of the piece holding the rem-th symbol (rem >= 1; c = 0 when the pieces
// hold fewer, or rem == 0), occ = how many of the first rem symbols are c.  want != 0: the symbol is
// known already (c = want), only its count is wanted.
struct char_rank {
    uint32_t c, occ;
    sym_tab tab;  // make_sym_tab(c): the caller's whole-quarter sums of c use it too
};
__device__ __forceinline__ char_rank char_rank24(const uint32_t r[6], uint32_t rem, uint32_t want) {
    uint32_t cum[7];
    cum[0] = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) cum[i + 1] = __builtin_amdgcn_udot4(r[i] & 0x1F1F1F1Fu, 0x01010101u, cum[i], false);
    uint32_t x = r[0], base = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        const bool past = rem > cum[i];
        x = past ? r[i] : x;
        base = past ? cum[i] : base;
    }
    const uint32_t rd = rem - base;
    const uint32_t ps = (x & 0x1F1F1F1Fu) * 0x01010101u;  // bytes: symbols up to and including piece 0, 1, 2, 3
    const uint32_t j = (rd > (ps & 0xFFu) ? 1u : 0u) + (rd > ((ps >> 8) & 0xFFu) ? 1u : 0u) + (rd > ((ps >> 16) & 0xFFu) ? 1u : 0u);
    const bool found = rem != 0u && rd <= (ps >> 24);
    const uint32_t here = (x >> (8u * j + 5u)) & 7u;
    char_rank o;
    o.c = want ? want : (found ? here : 0u);
    const uint32_t pj = j ? __builtin_amdgcn_ubfe(ps, 8u * j - 8u, 8u) : 0u;  // symbols before piece j of the dword
    o.tab = make_sym_tab(o.c);
    uint32_t mcum = 0, before = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        mcum = __builtin_amdgcn_udot4(r[i] & 0x1F1F1F1Fu, match01(r[i], o.tab), mcum, false);
        before = rem > cum[i + 1] ? mcum : before;
    }
    // the dword's pieces before piece j: the other pieces' lengths are masked off, whatever their symbols
    const uint32_t lm = x & 0x1F1F1F1Fu & ((1u << (8u * j)) - 1u);
    const uint32_t inner = __builtin_amdgcn_udot4(lm, match01(x, o.tab), 0u, false);
    // piece j counts up to the position (all of it when the position lies past the pieces) if it is
    // a run of c -- always, unless c was given and differs
    const uint32_t reach = found ? rd - pj : (rem ? (ps >> 24) - pj : 0u);
    o.occ = before + inner + (here == o.c ? reach : 0u);
    return o;
}

// BPTree::select's leaf step (BPTree.h:131-187) within 24 pieces: position (symbols from the first
// piece, 0-based) of the t-th b (t >= 1); *left = what remains of t when the pieces hold fewer
// (0: found; t == 0 gives position 0, left 0).
__device__ __forceinline__ uint32_t select_in24(const uint32_t r[6], uint32_t b, uint32_t t, uint32_t *left) {
    const sym_tab tab = make_sym_tab(b);
    uint32_t cum[7], mat[7];
    cum[0] = mat[0] = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t l = r[i] & 0x1F1F1F1Fu;
        cum[i + 1] = __builtin_amdgcn_udot4(l, 0x01010101u, cum[i], false);
        mat[i + 1] = __builtin_amdgcn_udot4(l, match01(r[i], tab), mat[i], false);
    }
    uint32_t x = r[0], tb = 0, mb = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        const bool past = t > mat[i];
        x = past ? r[i] : x;
        tb = past ? cum[i] : tb;
        mb = past ? mat[i] : mb;
    }
    *left = t > mat[6] ? t - mat[6] : 0u;
    const uint32_t td = t - mb;
    const uint32_t lx = x & 0x1F1F1F1Fu;
    const uint32_t m01 = match01(x, tab);
    const uint32_t ps = lx * 0x01010101u;                   // symbols up to and including piece 0..3
    const uint32_t qs = (lx & (m01 * 0xFFu)) * 0x01010101u;  // b's up to and including piece 0..3
    const uint32_t j = (td > (qs & 0xFFu) ? 1u : 0u) + (td > ((qs >> 8) & 0xFFu) ? 1u : 0u) + (td > ((qs >> 16) & 0xFFu) ? 1u : 0u);
    const uint32_t pj = j ? __builtin_amdgcn_ubfe(ps, 8u * j - 8u, 8u) : 0u;
    const uint32_t qj = j ? __builtin_amdgcn_ubfe(qs, 8u * j - 8u, 8u) : 0u;
    return t ? tb + pj + (td - qj) - 1u : 0u;
}

// w = p / S and p mod S for p < 2^40 (line_format.h, span_params): the f64 product's floor is w or
// w - 1, one compare puts it right.  Any S works, so the window span follows the data instead of
// the few divisors that have an exact 32-bit reciprocal.
__device__ __forceinline__ uint32_t fast_window(uint64_t p, uint32_t S, double inv, uint32_t &pin) {
    uint32_t w = (uint32_t)((double)p * inv);
    uint32_t r = (uint32_t)p - w * S;
    if (r >= S) {
        w += 1u;
        r -= S;
    }
    pin = r;
    return w;
}

}  // namespace rsb
#endif

// bwt_device.h -- gfx950 device primitives over the block/directory layout (block_format.h).
//
// One Occ lookup is served by a DPP quad: 4 adjacent lanes read one 128-B block as 4 x 32 B
// (two global_load_dwordx4 each) and rank it with quad_perm moves only -- no LDS traffic for the
// data, no barriers.  A 64-lane wavefront therefore resolves 16 Occ lookups per pass.
#ifndef RSBWT_BWT_DEVICE_H
#define RSBWT_BWT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_format.h"

namespace rsb {

// ---- DPP moves inside a row of 16 lanes (all lanes of a quad / octet are always active
// ---- together in the kernels below, so every source lane is valid).
//
// The result is made opaque (empty asm) so that hipcc's DPP combiner cannot fold the move into
// the consuming VALU op: ROCm 7.2 folds `x - dpp(y)` into `v_subrev_u32_dpp`, and on gfx950 that
// instruction permutes the OTHER operand (it computes dpp(x) - y with x, y swapped: measured with
// tools/dpp_subrev_test.hip), which silently broke the hop loop below.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
    asm("" : "+v"(r));
    return r;
}
// quad_perm encodings: sel0 | sel1<<2 | sel2<<4 | sel3<<6
constexpr int DPP_QUAD_BCAST0 = 0x00;
constexpr int DPP_QUAD_BCAST1 = 0x55;
constexpr int DPP_QUAD_BCAST2 = 0xAA;
constexpr int DPP_QUAD_BCAST3 = 0xFF;
constexpr int DPP_QUAD_XOR1 = 0xB1;     // [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i <-> 7-i inside each 8 lanes

template <int CTRL>
__device__ __forceinline__ uint64_t dpp_mov64(uint64_t v) {
    const uint32_t lo = dpp_mov<CTRL>((uint32_t)v);
    const uint32_t hi = dpp_mov<CTRL>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t quad_sum64(uint64_t v) {
    v += dpp_mov64<DPP_QUAD_XOR1>(v);
    v += dpp_mov64<DPP_QUAD_XOR2>(v);
    return v;
}

// ---- directory: symbol position -> block (block_format.h, DIRECTORY)
// EXACT8: s == 8, four 8-bit fields, always exact.
template <bool EXACT8>
__device__ __forceinline__ uint32_t dir_decode(const rsbwt_view &ix, uint2 e, uint64_t p) {
    uint32_t j = e.x;
    if (EXACT8) {
        const uint32_t pin = (uint32_t)p & 255u;
        // field k counts when 1 <= f_k <= pin, i.e. (f_k - 1) < pin as unsigned
        j += (__builtin_amdgcn_ubfe(e.y, 0, 8) - 1u < pin) ? 1u : 0u;
        j += (__builtin_amdgcn_ubfe(e.y, 8, 8) - 1u < pin) ? 1u : 0u;
        j += (__builtin_amdgcn_ubfe(e.y, 16, 8) - 1u < pin) ? 1u : 0u;
        j += ((e.y >> 24) - 1u < pin) ? 1u : 0u;
    } else {
        const uint32_t s = ix.dir_shift;
        const uint32_t mask = (1u << s) - 1u;
        const uint32_t pin = (uint32_t)p & mask;
        uint32_t f = e.y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < (int)ix.dir_fields) {
                j += ((f & mask) - 1u < pin) ? 1u : 0u;
                f >>= s;
            }
        }
    }
    return j;
}

template <bool EXACT8>
__device__ __forceinline__ const uint2 *dir_entry_ptr(const rsbwt_view &ix, uint64_t p) {
    return ix.dir + (EXACT8 ? (p >> 8) : (p >> ix.dir_shift));
}

// One quad lane's 32 bytes of a block.
struct lane_block {
    uint32_t hdr_lo, hdr_hi;  // header word t
    uint32_t r[6];            // run bytes 24t .. 24t+23
};

// lane_base = ix.blocks + 2 * t (the lane's 32-byte slice of block 0)
__device__ __forceinline__ lane_block load_lane_block(const uint4 *lane_base, uint64_t blk) {
    const uint4 *bp = lane_base + blk * 8u;
    uint4 a = bp[0];
    uint4 c = bp[1];
    // Pin both 16-B loads here: left alone, hipcc fetches only the meta dword first (for the hop
    // test) and the rest after it -- two dependent HBM round trips instead of one.
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
    lane_block lb;
    lb.hdr_lo = a.x;
    lb.hdr_hi = a.y;
    lb.r[0] = a.z;
    lb.r[1] = a.w;
    lb.r[2] = c.x;
    lb.r[3] = c.y;
    lb.r[4] = c.z;
    lb.r[5] = c.w;
    return lb;
}

// What the quad needs from the four meta fields (block_format.h).
struct block_meta {
    uint32_t P0_lo24;  // P0 bits 0..23
    uint32_t span;     // symbols in the block
    uint32_t start;    // symbols of the block held by the lanes below this one
};

__device__ __forceinline__ block_meta quad_block_meta(const lane_block &lb, uint32_t t) {
    const uint32_t meta = lb.hdr_hi >> 8;
    block_meta bm;
    bm.P0_lo24 = dpp_mov<DPP_QUAD_BCAST0>(meta);
    const uint32_t m2 = dpp_mov<DPP_QUAD_BCAST2>(meta);
    const uint32_t m3 = dpp_mov<DPP_QUAD_BCAST3>(meta);
    bm.span = m2 & 0xFFFu;
    const uint32_t s12 = (t & 1u) ? (m2 >> 12) : (m3 & 0xFFFu);  // t=1: start_1, t=2: start_2
    bm.start = (t == 0u) ? 0u : (t == 3u) ? (m3 >> 12) : s12;
    return bm;
}

// Offset of position p inside the block (0-based), from 24-bit modular arithmetic: the block the
// directory names starts at most a few blocks (<< 2^24 symbols) before p.
__device__ __forceinline__ uint32_t offset_in_block(const block_meta &bm, uint64_t p) {
    return ((uint32_t)p - bm.P0_lo24) & 0xFFFFFFu;
}

// matched symbols in one dword of 4 runs: acc + sum of len over the bytes whose symbol == b
// (bb = b in every byte), 4 runs per v_dot4_u32_u8
__device__ __forceinline__ uint32_t dword_matched(uint32_t x, uint32_t bb, uint32_t acc) {
    const uint32_t z = ((x >> 5) & 0x07070707u) ^ bb;             // 0 where the symbol matches
    const uint32_t m01 = ((0x80808080u - z) >> 7) & 0x01010101u;  // 1 where it matches
    return __builtin_amdgcn_udot4(x & 0x1F1F1F1Fu, m01, acc, false);
}

// Sum over this lane's 24 runs of min(len, what is left of `rem` symbols), counting only runs of
// symbol b.  RLEBWT::getOcc's bucket scan (src/bwt/rlebwt.cpp:281-298), 5 VALU per run byte:
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

__device__ __forceinline__ uint32_t lane_scan(const lane_block &lb, uint32_t b, uint32_t rem) {
    return runs_scan<6>(lb.r, b, rem);
}

// Occ(b, p) for the block that holds position p: # of symbol b (rank 1..4) in BWT[0..p].
// The 96 runs are scanned 24 per lane and summed across the quad together with the header count.
__device__ __forceinline__ uint64_t quad_rank(const lane_block &lb, const block_meta &bm,
                                              uint32_t t, uint32_t b, uint32_t off) {
    const uint32_t o = off + 1u;  // symbols of this block to count
    const uint32_t rem = o > bm.start ? o - bm.start : 0u;
    const uint32_t acc = lane_scan(lb, b, rem);
    const uint64_t cnt = ((uint64_t)(lb.hdr_hi & 0xFFu) << 32) | lb.hdr_lo;
    uint64_t mine = (t + 1u == b) ? cnt : 0ull;
    mine += acc;
    return quad_sum64(mine);
}

// Rank (0..4) of the symbol at offset `off` of the block
// (RLEBWT::getChar's bucket scan, src/bwt/rlebwt.cpp:213-224).
__device__ __forceinline__ uint32_t quad_char(const lane_block &lb, const block_meta &bm,
                                              uint32_t t, uint32_t off) {
    int rem = (int)(off + 1u) - (int)bm.start;
    uint32_t found = 0;  // sym + 1 of the run holding the offset, if it is in this lane
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t x = lb.r[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int len = (int)__builtin_amdgcn_ubfe(x, 8 * k, 5);
            const uint32_t sym = __builtin_amdgcn_ubfe(x, 8 * k + 5, 3);
            found = (rem >= 1 && rem <= len) ? sym + 1u : found;
            rem -= len;
        }
    }
    found |= dpp_mov<DPP_QUAD_XOR1>(found);
    found |= dpp_mov<DPP_QUAD_XOR2>(found);
    return found - 1u;
}

// Fetch the block holding position p: directory entry, then the block it names; when the entry
// is not exact (windows with more block starts than fields) advance while p is beyond the block.
// Returns the block id; `off` receives p's offset inside it.
template <bool EXACT8>
__device__ __forceinline__ uint64_t quad_fetch(const rsbwt_view &ix, const uint4 *lane_base,
                                               uint64_t p, uint32_t t, lane_block &lb,
                                               block_meta &bm, uint32_t &off) {
    const uint2 e = *dir_entry_ptr<EXACT8>(ix, p);
    uint64_t blk = dir_decode<EXACT8>(ix, e, p);
    lb = load_lane_block(lane_base, blk);
    bm = quad_block_meta(lb, t);
    off = offset_in_block(bm, p);
    if (!EXACT8) {
        // Rare: the window holds more block starts than the entry has fields, so the entry named
        // an earlier block.  Every p < n lies in some block at or after it, so advancing while p
        // is beyond the block ends.  The loop is wave-uniform (ballot) with a predicated body: all
        // lanes of every quad stay active for the DPP moves whatever their neighbours need.
        bool need = off >= bm.span;
        while (__builtin_amdgcn_ballot_w64(need) != 0ull) {
            blk += need ? 1u : 0u;
            lb = load_lane_block(lane_base, blk);
            bm = quad_block_meta(lb, t);
            off = offset_in_block(bm, p);
            need = off >= bm.span && blk + 1 < ix.nblocks;
        }
    }
    return blk;
}

}  // namespace rsb
#endif

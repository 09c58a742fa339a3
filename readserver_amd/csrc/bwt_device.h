// bwt_device.h -- gfx950 device primitives over the block/directory layout (block_format.h).
//
// One Occ lookup is served by a DPP quad: 4 adjacent lanes read one 128-B block as 4 x 32 B
// (two global_load_dwordx4 each) and rank it with quad_perm moves only -- no LDS, no barriers.
// A 64-lane wavefront therefore resolves 16 Occ lookups per pass.
#ifndef RSBWT_BWT_DEVICE_H
#define RSBWT_BWT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "block_format.h"

namespace rsb {

// ---- DPP moves inside a row of 16 lanes (all lanes of a quad / octet are always active
// ---- together in the kernels below, so every source lane is valid).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
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
__device__ __forceinline__ uint64_t dir_lookup(const rsbwt_view &ix, uint64_t p) {
    const uint32_t s = ix.dir_shift;
    const uint32_t mask = (1u << s) - 1u;
    const uint2 e = ix.dir[p >> s];
    const uint32_t pin = (uint32_t)p & mask;
    uint32_t j = e.x;
    uint32_t f = e.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < (int)ix.dir_fields) {
            const uint32_t fk = f & mask;
            j += (fk != 0u && pin >= fk) ? 1u : 0u;
            f >>= s;
        }
    }
    return j;
}

// One quad lane's 32 bytes of a block.
struct lane_block {
    uint32_t hdr_lo, hdr_hi;  // header word t
    uint32_t r[6];            // run bytes 24t .. 24t+23
};

__device__ __forceinline__ lane_block load_lane_block(const rsbwt_view &ix, uint64_t blk,
                                                      uint32_t t) {
    const uint4 *bp = ix.blocks + blk * 8u + t * 2u;
    uint4 a = bp[0];
    uint4 c = bp[1];
    // Pin both 16-B loads here: left alone, hipcc fetches only the meta dword first (for the hop
    // test below) and the rest after it -- two dependent HBM round trips instead of one.
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

struct block_meta {
    uint64_t P0;    // symbols before the block
    uint32_t span;  // symbols in the block
};

__device__ __forceinline__ block_meta quad_block_meta(const lane_block &lb) {
    const uint32_t meta = lb.hdr_hi >> 8;
    const uint32_t m0 = dpp_mov<DPP_QUAD_BCAST0>(meta);
    const uint32_t m1 = dpp_mov<DPP_QUAD_BCAST1>(meta);
    block_meta bm;
    bm.span = dpp_mov<DPP_QUAD_BCAST2>(meta);
    bm.P0 = (uint64_t)m0 | ((uint64_t)m1 << 24);
    return bm;
}

// Symbols held by this lane's 24 run bytes, and the symbols held by the quad's lower lanes.
__device__ __forceinline__ uint32_t lane_symbols(const lane_block &lb) {
    uint32_t tot = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) tot = __builtin_amdgcn_sad_u8(lb.r[i] & 0x1F1F1F1Fu, 0u, tot);
    return tot;
}

__device__ __forceinline__ uint32_t quad_exclusive_start(uint32_t tot, uint32_t t) {
    const uint32_t t0 = dpp_mov<DPP_QUAD_BCAST0>(tot);
    const uint32_t t1 = dpp_mov<DPP_QUAD_BCAST1>(tot);
    const uint32_t t2 = dpp_mov<DPP_QUAD_BCAST2>(tot);
    return (t > 0u ? t0 : 0u) + (t > 1u ? t1 : 0u) + (t > 2u ? t2 : 0u);
}

// Occ(b, p) for the block that holds position p: # of symbol b (rank 1..4) in BWT[0..p].
// Follows RLEBWT::getOcc's bucket scan (src/bwt/rlebwt.cpp:281-298) with the marker replaced by
// the block header; the 96 runs are scanned 24 per lane and summed across the quad.
__device__ __forceinline__ uint64_t quad_rank(const lane_block &lb, const block_meta &bm,
                                              uint32_t t, uint32_t b, uint64_t p) {
    const uint32_t o = (uint32_t)(p - bm.P0) + 1u;  // symbols of this block to count
    const uint32_t tot = lane_symbols(lb);
    int rem = (int)o - (int)quad_exclusive_start(tot, t);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t x = lb.r[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int len = (int)__builtin_amdgcn_ubfe(x, 8 * k, 5);
            const uint32_t sym = __builtin_amdgcn_ubfe(x, 8 * k + 5, 3);
            const int take = min(max(rem, 0), len);  // v_med3_i32
            acc += (sym == b) ? (uint32_t)take : 0u;
            rem -= len;
        }
    }
    const uint64_t cnt = ((uint64_t)(lb.hdr_hi & 0xFFu) << 32) | lb.hdr_lo;
    uint64_t mine = (t + 1u == b) ? cnt : 0ull;
    mine += acc;
    return quad_sum64(mine);
}

// Rank (0..4) of the symbol at position p of the block that holds it
// (RLEBWT::getChar's bucket scan, src/bwt/rlebwt.cpp:213-224).
__device__ __forceinline__ uint32_t quad_char(const lane_block &lb, const block_meta &bm,
                                              uint32_t t, uint64_t p) {
    const uint32_t o = (uint32_t)(p - bm.P0) + 1u;
    const uint32_t tot = lane_symbols(lb);
    int rem = (int)o - (int)quad_exclusive_start(tot, t);
    uint32_t found = 0;  // sym + 1 of the run holding offset o, if it is in this lane
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

// Fetch the block holding position p (directory estimate, then forward hops in the rare
// windows that hold more block starts than the entry has fields).
__device__ __forceinline__ uint64_t quad_fetch(const rsbwt_view &ix, uint64_t p, uint32_t t,
                                               lane_block &lb, block_meta &bm) {
    uint64_t blk = dir_lookup(ix, p);
    lb = load_lane_block(ix, blk, t);
    bm = quad_block_meta(lb);
    if (__builtin_expect(p >= bm.P0 + bm.span, 0)) {
        while (p >= bm.P0 + bm.span && blk + 1 < ix.nblocks) {
            ++blk;
            lb = load_lane_block(ix, blk, t);
            bm = quad_block_meta(lb);
        }
    }
    return blk;
}

__device__ __forceinline__ uint64_t select_C(const rsbwt_view &ix, uint32_t b) {
    uint64_t c = ix.C[1];
    c = (b == 2u) ? ix.C[2] : c;
    c = (b == 3u) ? ix.C[3] : c;
    c = (b == 4u) ? ix.C[4] : c;
    c = (b == 0u) ? ix.C[0] : c;
    return c;
}

__device__ __forceinline__ uint64_t select_total(const rsbwt_view &ix, uint32_t b) {
    uint64_t c = ix.total[1];
    c = (b == 2u) ? ix.total[2] : c;
    c = (b == 3u) ? ix.total[3] : c;
    c = (b == 4u) ? ix.total[4] : c;
    c = (b == 0u) ? ix.total[0] : c;
    return c;
}

}  // namespace rsb
#endif

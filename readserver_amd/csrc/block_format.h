// block_format.h -- the HBM layout of one popBWT shard ("Data layout in HBM").
//
// The run bytes are ReadServer's RLUnit bytes verbatim (include/bwt/rlunit.h:8-11:
// rank(sym) << 5 | len, len 1..31, rank $=0 A=1 C=2 G=3 T=4; alphabet.h:8-9).
//
// BLOCK (128 B, 128-B aligned, one L2 line) = 96 consecutive run bytes plus absolute
// checkpoints, cut so that a DPP quad (4 lanes) reads it as 4 x 32 B:
//
//   lane t (t = 0..3), bytes [32t, 32t+8)   header word t (u64, little endian)
//        bits  0..39   # of symbol (t+1) in BWT[0, P0)        (A, C, G, T)
//        bits 40..63   meta_t : t=0  P0 bits 0..23
//                               t=1  P0 bits 24..39 | run bytes used (0..96) << 16
//                               t=2  span (symbols in this block, 0..2976) | start_1 << 12
//                               t=3  start_2 | start_3 << 12
//                      start_t = symbols of the block held by lanes 0..t-1 (start_0 = 0)
//                 bytes [32t+8, 32t+32)     run bytes 24t .. 24t+23 of the block (unused = 0)
//
//   P0 = number of BWT symbols before the block.  #$ before the block = P0 - (A+C+G+T).
//
// DIRECTORY (8 B per 2^s symbols): entry w = { u32 id, u32 offs }.
//   id    = block holding symbol position w << s.
//   offs  = K = 32 / s fields of s bits: the offsets (1 .. 2^s-1, ascending, 0 = none) inside
//           window w at which blocks id+1, id+2, ... start.
//   block(p) = id + #{k : offs_k != 0 && (p & (2^s-1)) >= offs_k}, exact whenever window w holds
//   at most K block starts (always for s == 8, as a full block spans >= 96 symbols); otherwise
//   a lower bound that the reader advances while p >= P0 + span.
//
// K-MER TABLE (8 B per T-mer, 4^T entries): the interval findInterval returns for every string of
// T symbols over ACGT, so a search of k >= T symbols starts from one lookup on its last T symbols
// instead of T-1 LF steps.  Code of a T-mer = its 2-bit packing (symbol i at bits 2i).
#ifndef RSBWT_BLOCK_FORMAT_H
#define RSBWT_BLOCK_FORMAT_H

#include <stdint.h>

#define RSBWT_BLOCK_BYTES 128
#define RSBWT_BLOCK_RUNS 96
#define RSBWT_LANE_RUNS 24
#define RSBWT_COUNT_BITS 40
#define RSBWT_COUNT_MASK ((1ull << RSBWT_COUNT_BITS) - 1)
#define RSBWT_MAX_SYMBOLS (1ull << 40) /* per shard; counts are 40-bit */
#define RSBWT_MIN_DIR_SHIFT 8
#define RSBWT_MAX_DIR_SHIFT 16

// What a kernel needs to search one shard.  Plain pointers into HBM, passed by value.
struct rsbwt_view {
    const uint4 *blocks;  // nblocks * 8 uint4
    const uint2 *dir;     // nwin entries
    uint64_t n;           // symbols (getBWLen)
    uint64_t nblocks;
    uint64_t nwin;
    uint64_t C[5];        // C[c] = # symbols with rank < c   (getPC)
    uint64_t total[5];    // occurrences of each symbol in the whole BWT
    uint32_t dir_shift;   // s
    uint32_t dir_fields;  // K = 32 / s
    // k-mer table (optional): entry c = interval of the T-mer whose 2-bit code is c, i.e. what
    // findInterval returns for it (early exit included): bits 0..39 lower, bits 40..63 width =
    // upper - lower + 1 (0 = empty, upper = lower - 1; RSBWT_KTAB_WIDE = not tabulated).
    const uint64_t *ktab;
    uint32_t ktab_depth;  // T (0 = no table)
};

#define RSBWT_KTAB_WIDE 0xFFFFFFu

// Single-request search layout (slots.hip): slot(p) = mulhi(p >> a, magic) >> shift = p / S.
struct slot_params {
    uint32_t S;      // symbols per slot = m << a
    uint32_t a;      // 7 or 8: p >> a fits 32 bits
    uint32_t magic;  // exact 32-bit reciprocal of m
    uint32_t shift;
    uint64_t nslots;
};

struct slot_view {
    const uint4 *slots;  // nslots slots followed by the overflow blocks; nullptr = not built
    slot_params p;
    uint64_t noverflow;
};

#endif

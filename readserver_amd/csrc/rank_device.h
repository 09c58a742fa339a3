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

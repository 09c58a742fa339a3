// valu_rate.hip -- what one SIMD of an MI355X CU issues per cycle, by instruction kind and by the number of
// waves it holds.  The question it settles: is a wave64 VALU instruction 4 cycles of its
// SIMD whoever else is resident (then N waves x I instructions x 4 cycles is the SIMD's busy time and the search
// kernel's 400 VALU per pass at 4 waves per SIMD is 83 % of its pass), or do two waves' instructions overlap
// (2 cycles each at >= 2 waves per SIMD: 42 %)?
//
// Every wave runs ITERS x 64 instructions of one kind on 8 independent registers (no dependent chain shorter than
// 8 instructions) between two s_memtime stamps; printed: cycles per instruction AS SEEN BY ONE WAVE and the
// instructions per cycle and SIMD that follow, at 1, 2, 4 and 8 waves per SIMD (256-thread workgroups, one wave per
// SIMD each, k workgroups per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

#define REP8(INS)                                                                                              \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) \
                     INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2)   \
                         INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6)       \
                             INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2)   \
                                 INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5)       \
                                     INS(6) INS(7)                                                                  \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(c), "v"(d)                                                                                   \
                 : "vcc")

#define I_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define I_LSHR(i) "v_lshrrev_b32 %" #i ", 5, %" #i "\n\t"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_DOT4(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n\t"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_CNDM(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define I_CMP(i) "v_cmp_gt_u32 vcc, %" #i ", %8\n\t"
#define I_SDWA(i) "v_min_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define I_CMPS(i) "v_cmp_eq_u32_sdwa vcc, %" #i ", %8 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n\t"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n\t"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 5, 3\n\t"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 8, %" #i "\n\t"
#define I_CVTF64(i) "v_cvt_f32_u32 %" #i ", %" #i "\n\t"
#define I_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n\t"
#define I_MSAD(i) "v_msad_u8 %" #i ", %" #i ", %8, %9\n\t"
#define I_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_PK16(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n\t"
#define I_PKSUB(i) "v_pk_sub_u16 %" #i ", %" #i ", %8\n\t"

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t *sink, unsigned long long *cycles, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    uint32_t c = threadIdx.x | 0x01010101u, d = (threadIdx.x << 3) | 0x1F;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) REP8(I_ADD);
        if (KIND == 1) REP8(I_AND);
        if (KIND == 2) REP8(I_LSHR);
        if (KIND == 3) REP8(I_ADD3);
        if (KIND == 4) REP8(I_DOT4);
        if (KIND == 5) REP8(I_PERM);
        if (KIND == 6) REP8(I_CNDM);
        if (KIND == 7) REP8(I_CMP);
        if (KIND == 8) REP8(I_SDWA);
        if (KIND == 9) REP8(I_CMPS);
        if (KIND == 10) REP8(I_MULLO);
        if (KIND == 11) REP8(I_MUL24);
        if (KIND == 12) REP8(I_BFE);
        if (KIND == 13) REP8(I_LSHLADD);
        if (KIND == 14) REP8(I_SAD);
        if (KIND == 15) REP8(I_MSAD);
        if (KIND == 16) REP8(I_MOVDPP);
        if (KIND == 17) REP8(I_PK16);
        if (KIND == 18) REP8(I_ANDOR);
        if (KIND == 19) REP8(I_MAD24);
        if (KIND == 20) REP8(I_XOR);
        if (KIND == 21) REP8(I_PKSUB);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (s == 0x12345678u) sink[0] = s;  // (keeps the registers alive)
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int cus, uint32_t *sink, unsigned long long *d_cyc) {
    const int iters = 2000;
    printf("%-22s", name);
    for (int k : {1, 2, 4, 8}) {
        const int blocks = cus * k;
        CHECK(hipMemset(d_cyc, 0, sizeof(unsigned long long) * 4 * cus * 8));
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, sink, d_cyc, 10);  // warm
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, sink, d_cyc, iters);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(4 * blocks);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double per = (double)h[h.size() / 2] / ((double)iters * 64.0);  // cycles per instruction seen by the median wave
        printf("  %dw: %5.2f c/i = %4.2f i/c/SIMD", k, per, (double)k / per);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; median wave's cycles per instruction (s_memtime ticks) at k waves per SIMD, and instructions per cycle and SIMD\n", prop.name, cus);
    uint32_t *sink;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 4 * cus * 8));
    run<0>("v_add_u32", cus, sink, d_cyc);
    run<1>("v_and_b32", cus, sink, d_cyc);
    run<20>("v_xor_b32", cus, sink, d_cyc);
    run<2>("v_lshrrev_b32", cus, sink, d_cyc);
    run<3>("v_add3_u32", cus, sink, d_cyc);
    run<18>("v_and_or_b32", cus, sink, d_cyc);
    run<13>("v_lshl_add_u32", cus, sink, d_cyc);
    run<12>("v_bfe_u32", cus, sink, d_cyc);
    run<4>("v_dot4_u32_u8", cus, sink, d_cyc);
    run<5>("v_perm_b32", cus, sink, d_cyc);
    run<6>("v_cndmask_b32", cus, sink, d_cyc);
    run<7>("v_cmp_gt_u32", cus, sink, d_cyc);
    run<8>("v_min_u32_sdwa", cus, sink, d_cyc);
    run<9>("v_cmp_eq_u32_sdwa", cus, sink, d_cyc);
    run<10>("v_mul_lo_u32", cus, sink, d_cyc);
    run<11>("v_mul_u32_u24", cus, sink, d_cyc);
    run<19>("v_mad_u32_u24", cus, sink, d_cyc);
    run<14>("v_sad_u8", cus, sink, d_cyc);
    run<15>("v_msad_u8", cus, sink, d_cyc);
    run<16>("v_mov_b32_dpp", cus, sink, d_cyc);
    run<17>("v_pk_min_u16", cus, sink, d_cyc);
    run<21>("v_pk_sub_u16", cus, sink, d_cyc);
    return 0;
}

// Which operand does DPP permute, and what is computed, for VOP2 integer sub/subrev/add with DPP on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define T(name, insn)                                                                              \
    {                                                                                              \
        unsigned r;                                                                                \
        asm volatile("s_nop 4\n\t" insn " %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" \
                     : "=&v"(r) : "v"(x), "v"(y));                                                \
        out[idx * 64 + lane] = r;                                                                  \
        ++idx;                                                                                     \
    }
__global__ void k(unsigned *out) {
    const unsigned lane = threadIdx.x;
    unsigned x = lane * 100u + 7u;   // "meta"
    unsigned y = 1000000u + lane;    // "p"
    int idx = 0;
    T("sub", "v_sub_u32_dpp")
    T("subrev", "v_subrev_u32_dpp")
    T("add", "v_add_u32_dpp")
    T("xor", "v_xor_b32_dpp")
    T("max", "v_max_u32_dpp")
    T("and", "v_and_b32_dpp")
}
int main() {
    unsigned *d, h[6 * 64];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[6] = {"v_sub_u32_dpp d,x,y", "v_subrev_u32_dpp d,x,y", "v_add_u32_dpp d,x,y", "v_xor_b32_dpp", "v_max_u32_dpp", "v_and_b32_dpp"};
    for (int t = 0; t < 6; ++t) {
        const int l = 1;  // lane 1 reads lane 0 through DPP
        unsigned x0 = 7, x1 = 107, y0 = 1000000, y1 = 1000001;
        printf("%-24s lane1 = %u   | dpp(x)-y=%u  y-dpp(x)=%u  x-dpp(y)=%u  dpp(y)-x=%u  dpp(x)+y=%u x+dpp(y)=%u dpp(x)&y=%u x&dpp(y)=%u\n", names[t], h[t * 64 + l],
               x0 - y1, y1 - x0, x1 - y0, y0 - x1, x0 + y1, x1 + y0, x0 & y1, x1 & y0);
    }
    return 0;
}

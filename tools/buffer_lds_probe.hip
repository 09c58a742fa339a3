// buffer_lds_probe.hip -- what `buffer_load_dwordx4 ... idxen offen lds` does on gfx950 with a STRUCTURED descriptor
// (stride 128 = one window line per record) over an array larger than 4 GiB, before the search kernels' line fetch
// is built on it (csrc/wave_lines.h, glds_fetch):
//   (1) record index x stride is a 64-bit product in the address unit: lines past 4 GiB are reached;
//   (2) num_records counts RECORDS for such a descriptor: index >= num_records is out of range;
//   (3) an out-of-range lane makes no memory request and writes ZEROS into its LDS slot (or leaves it: printed).
// Why it matters: with the index in a VGPR the hardware forms the address -- no 64-bit shift and add per fetch
// instruction -- and a lane that wants no line can say so with an out-of-range index instead of an EXEC mask, a
// compare and a branch around each of the eight fetch instructions of a pass.
//   hipcc -O3 --offload-arch=gfx950 tools/buffer_lds_probe.hip -o tools/bin/buffer_lds_probe && tools/bin/buffer_lds_probe [GiB=6]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x)                                                            \
    do {                                                                    \
        hipError_t e_ = (x);                                                \
        if (e_ != hipSuccess) {                                             \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
            return 1;                                                       \
        }                                                                   \
    } while (0)

typedef __attribute__((address_space(3))) void *lds_void_ptr;

__host__ __device__ inline uint32_t pattern(uint64_t line, uint32_t dword) {
    uint64_t x = line * 32u + dword;
    x ^= x >> 29;
    x *= 0x9E3779B97F4A7C15ull;
    x ^= x >> 32;
    return (uint32_t)x | 1u;  // never 0: a zero read back is a zero written by the load, not data
}

__global__ void fill(uint32_t *p, uint64_t ndwords) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndwords; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = pattern(i >> 5, (uint32_t)(i & 31u));
}

// 64 lanes: lane t fetches 16 bytes (chunk t & 7) of record idx[t >> 3 ... ] -- here simply idx[t], chunk t & 7
__global__ void probe(const char *base, uint32_t nrec, const uint32_t *idx, uint4 *out) {
    __shared__ uint4 stage[64];
    stage[threadIdx.x] = make_uint4(0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu);
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)128, (int)nrec, 0x00020000);
    __builtin_amdgcn_struct_ptr_buffer_load_lds(r, (lds_void_ptr)stage, 16, idx[threadIdx.x], (threadIdx.x & 7u) << 4, 0, 0, 2);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    out[threadIdx.x] = stage[threadIdx.x];
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 6.0;
    const uint64_t nlines = (uint64_t)(gib * (1ull << 30)) / 128u;
    uint32_t *d = nullptr;
    CHECK(hipMalloc(&d, nlines * 128u));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d, nlines * 32u);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> idx(64);
    const uint32_t picks[8] = {0u, 12345u, (1u << 25) - 1u, 1u << 25, (1u << 25) + 777u, (uint32_t)nlines - 1u, (uint32_t)nlines, 0xFFFFFFFFu};
    for (int t = 0; t < 64; ++t) idx[t] = picks[t >> 3];
    uint32_t *d_idx;
    uint4 *d_out;
    CHECK(hipMalloc(&d_idx, 256));
    CHECK(hipMalloc(&d_out, 1024));
    CHECK(hipMemcpy(d_idx, idx.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, (const char *)d, (uint32_t)nlines, d_idx, d_out);
    CHECK(hipDeviceSynchronize());
    std::vector<uint4> out(64);
    CHECK(hipMemcpy(out.data(), d_out, 1024, hipMemcpyDeviceToHost));
    printf("array of %llu lines (%.1f GiB), num_records = %llu\n", (unsigned long long)nlines, gib, (unsigned long long)nlines);
    int bad = 0;
    for (int g = 0; g < 8; ++g) {
        const uint32_t i = picks[g];
        const bool in_range = (uint64_t)i < nlines;
        int ok = 0, zero = 0, kept = 0;
        for (int l = 0; l < 8; ++l) {
            const uint4 v = out[g * 8 + l];
            const uint32_t c = (uint32_t)l;
            if (in_range && v.x == pattern(i, 4 * c) && v.y == pattern(i, 4 * c + 1) && v.z == pattern(i, 4 * c + 2) && v.w == pattern(i, 4 * c + 3)) ++ok;
            if (v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0) ++zero;
            if (v.x == 0xDEADBEEFu && v.w == 0xDEADBEEFu) ++kept;
        }
        printf("  index %10u (%s): %d of 8 chunks are the line's bytes, %d zeros, %d left as they were\n", i, in_range ? "in range" : "OUT of range", ok, zero, kept);
        if (in_range ? ok != 8 : (zero != 8 && kept != 8)) ++bad;
    }
    printf("%s\n", bad ? "UNEXPECTED: do not build the fetch on this" : "as expected: 64-bit record addressing, record-count range check");
    return bad ? 2 : 0;
}

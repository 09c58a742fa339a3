// synth_runs.h -- the direct run-stream synthesiser: run byte i of stream `seed` is
// a pure function of (seed, i), so the GPU fill, the host fill and any slice agree bit for bit.
// Lengths: 20 % full units (31, i.e. pieces of long runs), 20 % 6..21, 60 % 1..4 (mean ~10.4
// symbols per byte, a population-BWT-like mix); symbols uniform over ACGT with ~1.2 % '$'.
// Seeds with bit 63 set select the LONG-RUN stream (a deep population BWT, whose units are mostly
// the 31-symbol pieces of long runs): blocks of 8 consecutive units, 80 % of them one run of
// 7 x 31 + (1..31) symbols, the rest drawn as above -- mean ~25 symbols per byte.
// Seeds with bit 62 set select the POPULATION stream: the unit-length histogram MEASURED on a valid
// population BWT of 1.1e9 symbols (64 haplotypes, 64 suffix shards at 28x depth each, 1 % base errors:
// tools/popbwt_gpu.py, profiles/r03_popbwt_calibration.json) -- 51 % units of one symbol, 10 % of two,
// a tail over 3..30 and 3 % full units: mean 5.8 symbols per byte; 2 % of the units are '$', so that an LF walk
// plus a psi walk from a random row cover ~100 symbols, the reference's read length.
#ifndef RSBWT_SYNTH_RUNS_H
#define RSBWT_SYNTH_RUNS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RSBWT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RSBWT_HD static inline
#endif

namespace rsb {

RSBWT_HD uint64_t synth_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

RSBWT_HD uint8_t synth_run_byte(uint64_t seed, uint64_t i) {
    if (seed >> 63) {
        const uint64_t hb = synth_mix64(seed * 0xD1342543DE82EF95ull + 0x5BD1E9955BD1E995ull + (i >> 3));
        if ((uint32_t)(hb % 10u) < 8u) {
            const uint32_t sym = 1u + (uint32_t)((hb >> 8) & 3u);
            const uint32_t len = (i & 7u) == 7u ? 1u + (uint32_t)((hb >> 16) % 31u) : 31u;
            return (uint8_t)((sym << 5) | len);
        }
    }
    const uint64_t h = synth_mix64(seed * 0xD1342543DE82EF95ull + i);
    if ((seed >> 62) & 1u) {
        const uint32_t sym = (uint32_t)(h & 0x3FF) < 20u ? 0u : 1u + (uint32_t)((h >> 10) & 3u);
        const uint32_t u = (uint32_t)((h >> 16) % 1000u);
        const uint32_t v = (uint32_t)(h >> 32);
        uint32_t len;
        if (u < 509u) len = 1u;
        else if (u < 613u) len = 2u;
        else if (u < 646u) len = 3u;
        else if (u < 667u) len = 4u;
        else if (u < 687u) len = 5u;
        else if (u < 917u) len = 6u + v % 12u;   // 6..17, ~1.9 % each
        else if (u < 968u) len = 18u + v % 13u;  // 18..30
        else len = 31u;
        return (uint8_t)((sym << 5) | len);
    }
    const uint32_t a = (uint32_t)(h & 0xFF);
    const uint32_t sym = a < 3u ? 0u : 1u + (uint32_t)((h >> 8) & 3u);
    const uint32_t u = (uint32_t)((h >> 16) % 10u);
    const uint32_t v = (uint32_t)(h >> 32);
    uint32_t len;
    if (u < 2u) len = 31u;
    else if (u < 4u) len = 6u + (v & 15u);
    else len = 1u + (v & 3u);
    return (uint8_t)((sym << 5) | len);
}

}  // namespace rsb
#endif

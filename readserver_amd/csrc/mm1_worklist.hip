// mm1_worklist.hip -- the 1-mismatch search of a shard set as a WORKLIST of live searches (gfx950).
//
// BASELINE configs[3]; not in the reference, defined by composition: the result is the exact findInterval
// (src/bwt/query.cpp:24-41) of every k-mer and of each of its 3k single-substitution variants.  Rounds 2-3 ran every
// variant as a search of its own, resumed from its k-mer's traced interval where the substitution lies left of the
// k-mer table's reach.  Measured on 8 x 20 GB shards (T = 14, k = 31): of a k-mer's 94 variant searches per shard the
// 51 resumed ones live 1.05 steps -- 19 in 20 die on the substituted symbol itself -- and the three substitutions of
// one position look up the SAME two positions (the traced interval's ends) for three different symbols.  So:
//
//   1. the k-mers are searched once, traced (search_lines.hip): trace[s][q][j] = the interval of suffix j+1.. of k-mer q;
//   2. wl_branch_kernel: one lane per (shard, k-mer, position j < tn) takes the step of ALL THREE substituted symbols
//      off one fetch of the interval's line(s) -- one line where the old launch read three -- and only the variants
//      that SURVIVE the step (1 in 20) become searches: a worklist record with their interval after the step;
//   3. the variants substituted inside the k-mer table's reach start from their own table entry (findInterval's
//      answer for their last T symbols): they need no record -- search_solo_kernel<WL> spells them out of the k-mer's
//      packed word and reads the entry itself, two passes ahead of taking the search up (the first version wrote a
//      32-byte record per variant with a kernel of its own: 1.6 ms and a tenth of the search launch's requests);
//   4. search_solo_kernel<WL> (search_solo.h) runs both: the implicit items, then the appended records -- a take-up
//      is one 32-byte read, no start-record launch, no variants spelled out in memory; results go to the variant's
//      canonical index (sparse results + hit map), so the ordered hit lists come out of the same compaction as before;
//   5. wl_own_kernel: the k-mers' own intervals (the traced launch's results) are variant 0's hits.
//
// Since the second half of round 4, steps 1, 2 and 5 are ONE walk of the k-mers (search_solo.h, WALK; sets.hip runs it unless
// RSBWT_SET_1MM_NO_WALK is set): a lane that searches a k-mer has the line of every traced position in LDS already, ranks the
// three other bases off it (staged_occ_alts, wave_lines.h) and appends the survivors itself -- no trace written and read back,
// no line fetched twice: 3.8 -> 2.1 ms per 4e5 31-mers x 8 shards.  The kernels below are the A/B path and what a set whose
// walk does not apply falls back to.
//
// Worklist record (32 B): x = lower (40 bits) | next symbol j << 40 (16 bits) | WL_DEAD << 63;  y = upper;
//                         z = canonical search index q * (3k+1) + v;  w = the variant's packed word (k <= 32).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "line_format.h"
#include "rank_device.h"
#include "wave_lines.h"

namespace rsb {

namespace {

#define COUNT_WORK_BRANCH(w) ((w) != nullptr)  // (wave-uniform: the counters cost a branch when they are off)

__global__ void wl_init_counts_kernel(unsigned long long *__restrict__ counts, uint32_t nshards, unsigned long long first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nshards) counts[(size_t)i * WL_COUNT_STRIDE] = first;
}

// ---- 3'. (round 5, optional: RSBWT_SET_1MM_TABLE_PREPASS) the table entries of the variants substituted INSIDE the
// tables' reach, read ahead for ALL the shards of the device.  Item i = variant r = i % 3T of k-mer q = i / 3T, as
// search_solo_kernel<WL> numbers its implicit items.  The S shards' records of one variant lie side by side in the
// interleaved tables -- one stretch of 12 S (8 S) bytes -- but the search kernel reads them from S different waves on S
// different XCDs at S different times: the same line crosses the fabric up to S times (PMC: 86.7 GB of traffic against
// 74.5 algorithmic per 4e5 31-mers x 8 shards, profiles/r04d_1mm_pmc.json).  Here adjacent lanes = the shards of one
// item read the stretch ONCE; the block's V x S entries are turned round in LDS and leave as plain 8-byte entries
// pre[s][i], which the search kernel then reads in item order (coalesced) instead of the table.
__global__ void __launch_bounds__(256)
wl_table_entries_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ packed, size_t items,
                        uint32_t k, uint32_t tn, uint32_t V, uint64_t *__restrict__ pre) {
    __shared__ uint64_t tile[256];
    const size_t i0 = (size_t)blockIdx.x * V;
    const uint32_t t = threadIdx.x;
    {
        const uint32_t v = t / nshards, s = t - v * nshards;
        if (v < V && i0 + v < items) {
            const shard_view &ix = shards[s];
            const uint32_t T = ix.ktab_depth, per = 3u * (k - tn);
            const uint32_t i32 = (uint32_t)(i0 + v), q = i32 / per, r = i32 - q * per;
            const uint32_t pr = r / 3u, d = r - 3u * pr, p = tn + pr;
            const uint64_t word = packed[q];
            const uint32_t orig = (uint32_t)((word >> (2u * p)) & 3u);
            const uint32_t alt = d < orig ? d : d + 1u;
            const uint64_t vword = word ^ ((uint64_t)(orig ^ alt) << (2u * p));
            const uint64_t code = (vword >> (2u * tn)) & ((1ull << (2u * T)) - 1ull);
            tile[v * nshards + s] = ktab_entry(ix.ktab, ix.ktab_fmt, T, ix.ktab_stride, code);
        }
    }
    __syncthreads();
    {
        const uint32_t s = t / V, v = t - s * V;
        if (s < nshards && i0 + v < items) pre[(size_t)s * items + i0 + v] = tile[v * nshards + s];
    }
}

// ---- 5. variant 0 = the k-mer itself: its traced search's result
__global__ void __launch_bounds__(256)
wl_own_kernel(const ulonglong2 *__restrict__ own, const uint8_t *__restrict__ valid, uint32_t nshards, size_t m, uint32_t k,
              ulonglong2 *__restrict__ sparse, unsigned long long *__restrict__ hit_bits, size_t mv) {
    const size_t t_ = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t_ >= m * nshards) return;
    const size_t s = t_ / m, q = t_ - s * m;
    const ulonglong2 o = own[t_];
    if (valid[q] == 0 || o.x > o.y) return;
    const size_t canon = q * (3ull * k + 1ull);
    sparse[s * mv + canon] = o;
    atomicOr(hit_bits + s * hit_map_words(mv) + (canon >> 6), 1ull << (canon & 63u));
}

// ---- 2. the step of the three substituted symbols at every traced position
// One lane per (k-mer, position) item of the wave's current shard; items are dealt statically (every item costs the
// same: one or two line fetches).  Pass A fetches the line of lower - 1 (of upper when lower = 0); pass B the line of
// upper when that is another one.  A position past its line's own pieces (a spill chunk / far line, ~1.5 % of the
// lookups) is not chased here: the item's three variants go to the worklist UNSTEPPED and the search kernel takes
// the step itself, continuation included.
__global__ void __launch_bounds__(64 * WG_WAVES, RSB_MIN_WGS_PER_CU)
wl_branch_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ packed,
                 const uint8_t *__restrict__ valid, size_t m, uint32_t k, uint32_t tn, const ulonglong2 *__restrict__ trace,
                 ulonglong2 *__restrict__ wl, size_t wl_cap, unsigned long long *__restrict__ wl_counts,
                 ulonglong2 *__restrict__ sparse, unsigned long long *__restrict__ hit_bits, size_t mv,
                 unsigned long long *__restrict__ work) {
    __shared__ uint4 s_stage[WG_WAVES][64 * SLOT_U4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *stage = s_stage[wave];
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)stage);
    const staged_line L = {own_stage_row(stage, lane), lane & 7u};
    const uint64_t V = 3ull * k + 1ull;
    const size_t items = m * tn;  // per shard (< 2^32: the caller's condition)
    const uint32_t waves_total = gridDim.x * WG_WAVES, wave_id = blockIdx.x * WG_WAVES + wave;
    unsigned long long w_lines = 0, w_items = 0, w_surv = 0, w_unstepped = 0;
    // (workgroups start on different shards: the lists' counters are one address per shard, and every wave appending
    // to the same one at the same time serialises on it -- 30 ms for 5e7 items before this and the single atomic below)
    uint32_t sid = blockIdx.x % nshards;
    for (uint32_t visited = 0; visited < nshards; ++visited, sid = (sid + 1u == nshards) ? 0u : sid + 1u) {
        const shard_view *sv = shards + sid;
        const char *lines_bytes = reinterpret_cast<const char *>(sv->lines);
        const uint32_t S = sv->sp.S, nlines = (uint32_t)sv->nlines;
        const double inv = sv->sp.inv;
        const uint64_t c1 = sv->C[1], c2 = sv->C[2], c3 = sv->C[3], c4 = sv->C[4];
        const ulonglong2 *trace_s = trace + (size_t)sid * items;
        ulonglong2 *wl_s = wl + (size_t)sid * wl_cap * 2u;
        ulonglong2 *sparse_s = sparse + (size_t)sid * mv;
        unsigned long long *bits_s = hit_bits + (size_t)sid * hit_map_words(mv);
        unsigned long long *count = wl_counts + (size_t)sid * WL_COUNT_STRIDE;
        for (size_t t0 = (size_t)wave_id * 64u; t0 < items; t0 += (size_t)waves_total * 64u) {
            const size_t t = t0 + lane;
            bool have = t < items;
            uint32_t q = 0, j = 0;
            uint64_t lo = 0, hi = 0, word = 0;
            if (have) {
                q = (uint32_t)(t / tn);
                j = (uint32_t)(t - (size_t)q * tn);
                const ulonglong2 tr = trace_s[t];
                lo = tr.x;
                hi = tr.y;
                // an invalid k-mer's trace was never written; an absent suffix has no variant that occurs.  (The
                // reference's (0, 2^64 - 1) carry -- query.cpp:35, rlebwt.cpp:269 -- is a live interval by its unsigned
                // compare: Occ(b, -1) = 0 on both sides, no line is needed)
                have = valid[q] != 0 && lo <= hi;
                if (have) word = packed[q];
            }
            if (COUNT_WORK_BRANCH(work)) w_items += __builtin_popcountll(__builtin_amdgcn_ballot_w64(have));
            // ---- the two positions and their windows
            uint32_t wL = 0, oL = 0, wU = 0, oU = 0;
            {
                uint32_t pin;
                if (have && lo != 0ull) {
                    wL = fast_window(lo - 1ull, S, inv, pin);
                    oL = pin + 1u;
                }
                if (have && hi != ~0ull) {
                    wU = fast_window(hi, S, inv, pin);
                    oU = pin + 1u;
                }
            }
            const bool needL = have && lo != 0ull;
            const bool needU = have && hi != ~0ull;  // (Occ(b, 2^64 - 1) = Occ(b, -1) = 0: rlebwt.cpp:269)
            const uint32_t orig = (uint32_t)((word >> (2u * j)) & 3u);  // (its own step is the traced search's: not taken again)
            uint64_t occL[3] = {0, 0, 0}, occU[3] = {0, 0, 0};
            bool spill = false;
            // ---- two passes over ONE copy of the code: A fetches the line of lower - 1 (at lower = 0: of upper) and
            // ranks lower - 1 in it; B fetches the line of upper where it is another one (a wave-uniform skip when no
            // lane needs it; the row of a lane that fetches nothing keeps pass A's line) and ranks upper in the lane's row
#pragma unroll 1
            for (uint32_t pass = 0; pass < 2u; ++pass) {
                const bool isU = pass != 0u;
                const bool fetch = isU ? (needL && needU && wU != wL && !spill) : (needL || needU);
                const uint32_t wsel = (isU || !needL) ? wU : wL;
                uint32_t line = wsel + (wsel >> GROUP_SHIFT);
                if (line >= nlines) line = 0;
                const uint64_t fmask = __builtin_amdgcn_ballot_w64(fetch);
                if (fmask != 0ull) {
                    if (COUNT_WORK_BRANCH(work)) w_lines += __builtin_popcountll(fmask);
                    glds_fetch(lines_bytes, fetch ? line : ~0u, lane, stage_lds);
                    glds_wait();
                }
                const line_head h = read_head(L);
                const bool need = isU ? needU : needL;
                const uint32_t o = isU ? oU : oL;
                if (need && o > h.span) spill = true;
                uint64_t t3[3];
                staged_occ_alts(L, h, need ? o : 1u, orig, t3);  // (every lane: a lane without the lookup drops the result)
                if (need && !spill) {
                    if (isU) { occU[0] = t3[0]; occU[1] = t3[1]; occU[2] = t3[2]; }
                    else { occL[0] = t3[0]; occL[1] = t3[1]; occL[2] = t3[2]; }
                }
            }
            // ---- the three substitutions of position j: updateInterval (query.cpp:11-15) with the substituted symbol
            uint64_t nlo[3], nhi[3];
            bool enq[3];
            uint64_t masks[3];
#pragma unroll
            for (uint32_t d = 0; d < 3u; ++d) {
                const uint32_t alt = d < orig ? d : d + 1u;  // the d-th base of ACGT without the original one
                const uint64_t cb = alt == 0u ? c1 : alt == 1u ? c2 : alt == 2u ? c3 : c4;
                nlo[d] = cb + occL[d];
                nhi[d] = cb + occU[d] - 1ull;
                const bool stepped_live = have && !spill && nlo[d] <= nhi[d];
                const bool final_hit = stepped_live && j == 0u;
                enq[d] = (stepped_live && j != 0u) || (have && spill);
                if (final_hit) {
                    const uint64_t canon = (uint64_t)q * V + 1ull + 3ull * j + d;
                    sparse_s[canon] = make_ulonglong2(nlo[d], nhi[d]);
                    atomicOr(bits_s + (canon >> 6), 1ull << (canon & 63u));
                }
                masks[d] = __builtin_amdgcn_ballot_w64(enq[d]);
                if (COUNT_WORK_BRANCH(work)) {
                    w_surv += __builtin_popcountll(__builtin_amdgcn_ballot_w64((enq[d] && !spill) || final_hit));
                    w_unstepped += __builtin_popcountll(__builtin_amdgcn_ballot_w64(enq[d] && spill));
                }
            }
            // ONE append per wave and pass for the three rounds together
            const uint32_t n0 = (uint32_t)__builtin_popcountll(masks[0]), n1 = (uint32_t)__builtin_popcountll(masks[1]);
            const uint32_t total = n0 + n1 + (uint32_t)__builtin_popcountll(masks[2]);
            if (total != 0u) {
                unsigned long long base = 0;
                if (lane == 0u) base = atomicAdd(count, (unsigned long long)total);
                base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
                const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
                for (uint32_t d = 0; d < 3u; ++d) {
                    if (!enq[d]) continue;
                    const size_t slot = (size_t)base + (d >= 1u ? n0 : 0u) + (d >= 2u ? n1 : 0u) + (size_t)__builtin_popcountll(masks[d] & lt);
                    if (slot >= wl_cap) continue;  // (room for every variant: the caller's sizing)
                    const uint32_t alt = d < orig ? d : d + 1u;
                    const uint64_t vword = word ^ ((uint64_t)(orig ^ alt) << (2u * j));
                    const uint64_t canon = (uint64_t)q * V + 1ull + 3ull * j + d;
                    if (spill) wl_store(wl_s, slot, lo, j, hi, canon, vword);              // the step is the search kernel's
                    else wl_store(wl_s, slot, nlo[d], j - 1u, nhi[d], canon, vword);       // already taken
                }
            }
        }
    }
    // (into the search launches' counters, search_lines.hip WORK_*: an item is the step of three variants off two lookups)
    if (work != nullptr && lane == 0u) {
        atomicAdd(&work[0], 3ull * w_items);   // WORK_STEPS
        atomicAdd(&work[1], 2ull * w_items);   // WORK_OCC
        atomicAdd(&work[2], w_lines);          // WORK_LINES
        atomicAdd(&work[13], w_surv);          // variants alive after the step
        atomicAdd(&work[14], w_unstepped);     // variants passed on unstepped (a position past its line's own pieces)
    }
}

}  // namespace

hipError_t launch_wl_table_entries(const shard_view *d_shards, uint32_t nshards, const void *d_packed, size_t m, uint32_t k, uint32_t tn,
                                   void *d_pre, hipStream_t stream) {
    const size_t items = m * 3u * (size_t)(k - tn);
    if (items == 0 || nshards == 0) return hipSuccess;
    if (nshards > 256u || k > 32u || tn >= k || items > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint32_t V = 256u / nshards;
    hipLaunchKernelGGL(wl_table_entries_kernel, dim3((unsigned)((items + V - 1) / V)), dim3(256), 0, stream, d_shards, nshards,
                       (const uint64_t *)d_packed, items, k, tn, V, (uint64_t *)d_pre);
    return hipGetLastError();
}

hipError_t launch_mm1_worklists(const shard_view *d_shards, uint32_t nshards, const void *d_packed, const void *d_valid, size_t m,
                                uint32_t k, uint32_t tn, const void *d_trace, const void *d_own, void *d_worklists, size_t wl_cap,
                                void *d_counts, void *d_sparse, void *d_hit_bits, int num_cus, hipStream_t stream,
                                unsigned long long *d_branch_work) {
    if (m == 0 || nshards == 0) return hipSuccess;
    if (tn == 0 || tn >= k || k > 32u) return hipErrorInvalidValue;
    const size_t mv = m * (3 * (size_t)k + 1);
    hipLaunchKernelGGL(wl_init_counts_kernel, dim3((nshards + 255u) / 256u), dim3(256), 0, stream, (unsigned long long *)d_counts, nshards, 0ull);
    const size_t no = m * nshards;
    hipLaunchKernelGGL(wl_own_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, stream, (const ulonglong2 *)d_own,
                       (const uint8_t *)d_valid, nshards, m, k, (ulonglong2 *)d_sparse, (unsigned long long *)d_hit_bits, mv);
    const size_t items = m * (size_t)tn;
    size_t g = (items + 64 * WG_WAVES - 1) / (64 * WG_WAVES);
    const size_t cap = (size_t)num_cus * RSB_MIN_WGS_PER_CU;
    if (g > cap) g = cap;
    hipLaunchKernelGGL(wl_branch_kernel, dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, d_shards, nshards, (const uint64_t *)d_packed,
                       (const uint8_t *)d_valid, m, k, tn, (const ulonglong2 *)d_trace, (ulonglong2 *)d_worklists, wl_cap,
                       (unsigned long long *)d_counts, (ulonglong2 *)d_sparse, (unsigned long long *)d_hit_bits, mv, d_branch_work);
    return hipGetLastError();
}

}  // namespace rsb

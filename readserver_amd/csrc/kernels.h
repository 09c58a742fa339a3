// kernels.h -- host-side launchers of the query kernels (kernels.hip, search_lines.hip) and of the
// index builder (build_lines.hip).
#ifndef RSBWT_KERNELS_H
#define RSBWT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "line_format.h"

namespace rsb {

// Scratch in HBM for the duration of one launch sequence on a stream (the search's start records,
// a walk's row counters).  A few buffers are kept and handed out again: to the stream that used one
// last (stream order makes that safe at once), or to any stream once the event recorded at the last
// give() has completed.  Nothing here calls the runtime's stream-ordered allocator: on this stack a
// hipMallocAsync / hipFreeAsync pair per call stalls for seconds once in a few thousand calls
// (profiles/r02d_latency.md).
class scratch_cache {
  public:
    struct lease {
        void *p = nullptr;
        int slot = -1;
    };
    hipError_t take(size_t bytes, hipStream_t stream, lease *out);
    void give(const lease &l, hipStream_t stream);  // after the last launch that uses l.p was enqueued
    void destroy();                                 // with no launch in flight
    size_t held_bytes();

  private:
    static constexpr int SLOTS = 16;
    struct slot_t {
        void *p = nullptr;
        size_t bytes = 0;
        hipStream_t last = nullptr;
        hipEvent_t done = nullptr;
        bool recorded = false, busy = false;
    };
    slot_t slots_[SLOTS];
    std::mutex mu_;
};

hipError_t launch_pack(const void *d_kmers, size_t Q, uint32_t k, size_t stride, void *d_packed,
                       void *d_valid, hipStream_t stream);

// search_lines.hip: batched findInterval of Q packed k-mers in each of the nshards shards whose
// views are the device array d_shards (all on the current device).  d_lower/d_upper: [nshards][Q]
// (d_lower alone receives counts with counts_only).  ev0/ev1 (optional) are recorded on `stream`
// immediately around the search kernel itself.
// words of the hit map of Q searches (one bit per search, a whole number of 16-byte units: the maps of a launch's
// shards lie back to back)
__host__ __device__ inline size_t hit_map_words(size_t Q) { return ((Q + 127) / 128) * 2; }

struct search_extra {
    // 1-mismatch search.  A traced search records, per (shard, k-mer), the interval
    // it holds when about to take each of its first trace_n symbols ([nshards][Q][trace_n] x {lower, upper});
    // the search of the k-mers' variants (`variants` per k-mer, variants_kernel's order) then starts
    // every variant whose substituted position is < trace_n from that interval (the shards of one launch
    // share trace_n: their k-mer tables have one depth).
    void *d_trace_out = nullptr;
    const void *d_trace_in = nullptr;
    uint32_t trace_n = 0, variants = 0;
    bool table_build = false;  // a k-mer table's own searches: same kernel under another name (profiles)
    bool pairs = false;        // results as {lower, upper}[nshards][Q] at d_lower (one 16-byte store per search)
    // sparse results: nothing is written for a search that ends empty; the others store
    // {lower, upper} at d_lower[s][search index] (16 B each, as with `pairs`) and set their bit in shard s's map
    // d_hit_bits[s][hit_map_words(Q)] (one bit per search, zeroed by the caller); launch_compact_hits turns the
    // two into a list per shard, ordered by search index
    void *d_hit_bits = nullptr;
    // the k-mer table leaves intervals well inside a window (n / 4^T << S): most steps of a search find
    // both positions in one line, which is what the one-lane-per-search kernel is for (search_solo.h)
    bool narrow = false;
    // start records computed ahead (launch_search_init, on any stream the caller orders before this launch):
    // [nshards][Q] x 16 B in the caller's HBM; the launch then takes no scratch and runs no start-record kernel
    const void *d_init = nullptr;
};
// the start-record kernel alone: d_init[s * Q + q] for every (query, shard) search of a batch
hipError_t launch_search_init(const shard_view *d_shards, uint32_t nshards, const void *d_packed, const void *d_valid, size_t Q,
                              uint32_t k, void *d_init, hipStream_t stream);
// queries of lengths of their own (search_lines.hip, search_init_var_kernel): the packing of `text` cut at off[0..Q] into wpq
// words per query + validity + lengths, and the start records [nshards][Q] a search launch then takes as search_extra::d_init
hipError_t launch_pack_var(const void *d_text, const void *d_off, size_t Q, uint32_t wpq, void *d_packed, void *d_valid, void *d_len,
                           hipStream_t stream);
hipError_t launch_search_init_var(const shard_view *d_shards, uint32_t nshards, const void *d_packed, const void *d_valid,
                                  const void *d_len, size_t Q, uint32_t wpq, void *d_init, hipStream_t stream);
hipError_t launch_search(scratch_cache &scratch, const shard_view *d_shards, uint32_t nshards, const void *d_packed,
                         const void *d_valid, size_t Q, uint32_t k, void *d_lower, void *d_upper, bool counts_only,
                         unsigned long long *d_work, int num_cus, hipStream_t stream, hipEvent_t ev0 = nullptr,
                         hipEvent_t ev1 = nullptr, const search_extra *extra = nullptr);
uint32_t trace_entries(const shard_view &ix, uint32_t k);
// The 1-mismatch search of a set by worklist (mm1_worklist.hip; k <= 32, 0 < tn < k, one table depth k - tn for all
// shards).  launch_mm1_worklists takes the step of the three substitutions of every traced position (d_trace
// [nshards][m][tn], d_own [nshards][m] pairs: the traced search's output) and appends the variants that survive it to
// d_worklists [nshards][wl_cap] x 32 B, wl_cap >= m * 3 * tn; d_counts u64[nshards * WL_COUNT_STRIDE] their lengths (entry s
// at s * WL_COUNT_STRIDE); hits that need no further step go straight to d_sparse [nshards][mv] / d_hit_bits.  d_branch_work
// (optional): the search launches' counter words (WORK_*): steps += 3 per item, lookups += 2, lines fetched; word 13 =
// variants alive after the step, 14 = variants passed on unstepped.
// launch_search_worklist then runs, per shard, the m * 3 (k - tn) variants substituted inside the tables' reach -- no
// records: the kernel spells them out and reads their table entries itself (search_solo.h, WL) -- and the appended
// records: results at the variants' canonical indices.
// (the lists' lengths sit WL_COUNT_STRIDE u64 apart: appended to by every wave of the branch kernel, they must not share
// a cache line -- eight counters in one line serialised the kernel at one atomic at a time: 10 ms instead of 3)
constexpr uint32_t WL_COUNT_STRIDE = 32;
// the search kernels' query pools (one counter per shard, drawn from by every wave of the launch) sit POOL_STRIDE u64
// apart too: eight adjacent counters are one line of one L2 channel, and a small batch -- 4,096 waves probing eight
// drained pools each -- then waits on that line longer than it searches (0.58 against 0.32 ms for 4e4 31-mers x 8 shards
// with a quarter of the waves)
constexpr uint32_t POOL_STRIDE = 32;
hipError_t launch_mm1_worklists(const shard_view *d_shards, uint32_t nshards, const void *d_packed, const void *d_valid, size_t m,
                                uint32_t k, uint32_t tn, const void *d_trace, const void *d_own, void *d_worklists, size_t wl_cap,
                                void *d_counts, void *d_sparse, void *d_hit_bits, int num_cus, hipStream_t stream,
                                unsigned long long *d_branch_work = nullptr);
// (d_pre, optional: the implicit items' table entries read ahead by launch_wl_table_entries, u64 [nshards][m * 3 (k - tn)])
hipError_t launch_search_worklist(scratch_cache &scratch, const shard_view *d_shards, uint32_t nshards, const void *d_packed,
                                  const void *d_valid, size_t m, uint32_t tn, const void *d_worklists, const void *d_counts, size_t wl_cap,
                                  uint32_t k, void *d_sparse, void *d_hit_bits, unsigned long long *d_work, int num_cus,
                                  hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const void *d_pre = nullptr);
hipError_t launch_wl_table_entries(const shard_view *d_shards, uint32_t nshards, const void *d_packed, size_t m, uint32_t k, uint32_t tn,
                                   void *d_pre, hipStream_t stream);
// launch_search_walk replaces the traced launch and launch_mm1_worklists' branch kernel by ONE walk of the k-mers
// (search_solo.h, WALK): the k-mers' own intervals to d_sparse / d_hit_bits at their canonical indices, the variants
// that survive the step of their position appended to d_worklists (d_counts zeroed by the caller), no trace.
hipError_t launch_search_walk(scratch_cache &scratch, const shard_view *d_shards, uint32_t nshards, const void *d_packed,
                              const void *d_valid, size_t m, uint32_t tn, void *d_worklists, void *d_counts, size_t wl_cap, uint32_t k,
                              void *d_sparse, void *d_hit_bits, unsigned long long *d_work, int num_cus, hipStream_t stream,
                              hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
constexpr int WORK_WORDS = 16;  // counters of a counting launch (search_lines.hip, WORK_*)

// k-mer table: fills d_entries[c * stride], c < 4^T, by searching every T-mer (fmt = KTAB_GROUPED: the 12-byte
// records at d_entries + 12 * g * stride bytes, g < 4^(T-1); *untabulated = T-mers left to the search itself).
// `view` is the host copy of the shard's view (no table yet).
hipError_t build_ktable(const shard_view &view, uint32_t T, uint64_t *d_entries, uint32_t stride, int num_cus,
                        hipStream_t stream, uint32_t fmt = KTAB_PLAIN, uint64_t *untabulated = nullptr);

// class BWT mirrors, batched
hipError_t launch_occ_batch(const shard_view &ix, const void *d_syms, const void *d_index, size_t n,
                            void *d_out, hipStream_t stream);
hipError_t launch_char_batch(const shard_view &ix, const void *d_index, size_t n, void *d_out,
                             hipStream_t stream);
// (needs the sampled select table: shard_view::sel, launch_select_samples)
hipError_t launch_occ_at_batch(const shard_view &ix, const void *d_syms, const void *d_bc, size_t n, void *d_out,
                               hipStream_t stream);
// The list of the set bits of `bits` (n_searches bits), in order: record i = {lower, upper, search index, 0}
// (32 B) from sparse[index]; at most `cap` records are written, *d_total receives how many there are.
// d_block_counts: compact_hits_block_words(n_searches) u64 of scratch.
// nseg > 1: the same for nseg maps at once (the shards of one search launch): maps hit_map_words(n_searches) words
// apart, sparse results n_searches records apart, lists cap records apart, totals one u64 each, block scratch
// compact_hits_block_words(n_searches) apart.
size_t compact_hits_block_words(size_t n_searches);
hipError_t launch_compact_hits(const void *d_bits, const void *d_sparse, size_t n_searches, void *d_hits, size_t cap,
                               void *d_total, void *d_block_counts, hipStream_t stream, uint32_t nseg = 1);
// {lower, upper} pairs <-> 10-byte {lower:40, width:40} records (kernels.hip); d_unfit: optional u32 counter of
// pairs that do not fit the record (none does for an interval findInterval produced)
hipError_t launch_pack_pairs10(const void *d_pairs, size_t n, void *d_packed, void *d_unfit, hipStream_t stream);
hipError_t launch_unpack_pairs10(const void *d_packed, size_t n, void *d_pairs, hipStream_t stream);
// extracted reads <-> 2 bits per base ([n][stride] ASCII + lengths <-> [n][stride / 4] bytes; stride % 16 == 0)
hipError_t launch_pack_reads2(const void *d_reads, const void *d_len, size_t n, uint32_t stride, void *d_packed, hipStream_t stream);
hipError_t launch_unpack_reads2(const void *d_packed, const void *d_len, size_t n, uint32_t stride, void *d_reads, hipStream_t stream);
hipError_t launch_variants(const void *d_packed, const void *d_valid, size_t Q, uint32_t k, void *d_vpacked,
                           void *d_vvalid, hipStream_t stream);
// read extraction: sampled select table (5 x stride u64: for every 256th occurrence of each symbol its
// window and how the 256 occurrences from it on spread over the next windows -- kernels.hip) and the walk kernel
hipError_t launch_debug_fast_window(const void *d_p, size_t n, uint32_t S, void *d_w, void *d_r, hipStream_t stream);
uint64_t select_sample_stride(const shard_view &ix);
hipError_t launch_select_samples(const shard_view &ix, uint64_t *d_sel, hipStream_t stream);
// psi hints inside the window lines that have room for one (line_format.h); after the samples.  *d_made (optional,
// zeroed by the caller) counts the lines that got one
hipError_t launch_psi_hints(const shard_view &ix, unsigned long long *d_made, hipStream_t stream);
// extract_lines.hip: extractPrefix + extractPostfix of n rows of EACH of the nshards shards whose views (with their
// select samples: shard_view::sel) are the device array d_shards, wave-cooperative, one launch sequence for all of
// them: d_rows [nshards][n], d_out [nshards][n][stride], d_plen / d_len [nshards][n]
hipError_t launch_extract_wave(scratch_cache &scratch, const shard_view *d_shards, uint32_t nshards, const void *d_rows,
                               size_t n, void *d_out, uint32_t stride, void *d_plen, void *d_len, int num_cus,
                               hipStream_t stream, unsigned long long *d_work = nullptr);
// d_work (counting mode): WORK_WORDS counters, zeroed by the caller: words 0-7 the prefix walk, 8-15 the
// postfix walk (extract_lines.hip, XW_*)
// (SEL_SHIFT, sample_window, window_samples, window_psi_hint: line_format.h -- shared with the host-side layout test)
// query / query_exactmatch (query.cpp:87-120) over extracted reads
hipError_t launch_match_reads(const void *d_reads, const void *d_len, size_t n, uint32_t stride, const void *d_owner,
                              const void *d_kmers, uint32_t k, size_t kstride, void *d_flags, hipStream_t stream);
hipError_t launch_synth_runs(void *d_runs, uint64_t num_runs, uint64_t seed, hipStream_t stream, uint64_t first = 0);
hipError_t launch_sample_present(const shard_view &ix, size_t Q, uint32_t k, size_t stride,
                                 uint64_t seed, void *d_kmers, hipStream_t stream);

// build_lines.hip: builds the window lines in HBM from run bytes in HBM.  On success fills `view`
// (lines are hipMalloc'ed and owned by the caller).  want_span 0 = choose S from the data.
// Synchronises `stream`.  build_error: 0 ok, BUILD_ERANGE, BUILD_EFORMAT.
enum { BUILD_OK = 0, BUILD_ERANGE = 1, BUILD_EFORMAT = 2 };
struct build_result {
    shard_view view;
    uint64_t num_runs;
    uint64_t hbm_bytes;
    uint64_t far_lines, chunk_windows, far_windows, spilled_symbols;
};
// hint_room: every window line keeps room for a psi hint (line_format.h; RSBWT_OPEN_READS).
hipError_t build_lines(const void *d_runs, uint64_t num_runs, uint32_t want_span, bool hint_room, hipStream_t stream,
                       build_result *out, int *build_error);

}  // namespace rsb
#endif

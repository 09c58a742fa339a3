// capi_internal.h -- what the C-ABI translation units (capi.hip, sets.hip) share: the handle
// structs behind include/rsbwt.h's opaque types, per-call contexts, and the search launcher.
#ifndef RSBWT_CAPI_INTERNAL_H
#define RSBWT_CAPI_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/rsbwt.h"
#include "ctx_pool.h"
#include "kernels.h"
#include "line_format.h"

namespace rsb {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int fail_hip(hipError_t e, const char *what);
int use_device(int device);
int resolve_device(int logical, int *physical);  // the device number of an rsbwt_open* call -> the GPU it names (capi.hip)

// A stream pair and a staging buffer for the duration of one host-buffer call.
struct call_ctx {
    hipStream_t st[2] = {nullptr, nullptr};
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    int stage(size_t bytes);  // grows the staging buffer; RSBWT_OK or RSBWT_ENOMEM
    // a small page-locked host buffer: the k-mers and answers of a SMALL call travel through it (a copy to
    // or from pageable memory costs the runtime a staging step and a wait of its own, three of them are
    // half of a lone request's latency); nullptr when it could not be had -- the call then goes the
    // ordinary way
    static constexpr size_t PIN_BYTES = 256u << 10;
    void *h_pin = nullptr;
};

// (the waiting logic: ctx_pool.h; 8 = the threads of the reference's query pool, service.cpp:88)
struct ctx_pool : bounded_pool<call_ctx, 8> {
    static constexpr int MAX_CTX = 8;
    call_ctx *acquire();  // blocks while MAX_CTX calls are in flight; nullptr = no stream could be made
    void destroy();       // with no call in flight
};

// HIP-event pairs of the most recent search launches + the counting mode's counters
struct search_meter {
    static constexpr int RING = 64;
    hipEvent_t ev_start[RING] = {}, ev_stop[RING] = {};
    uint64_t launches = 0;  // search launches so far; launch i uses pair i % RING
    bool counting = false;
    unsigned long long *d_work = nullptr;  // WORK_WORDS counters (search_lines.hip)
    scratch_cache scratch;                 // start records / row counters of the launches in flight
    std::mutex mu;
};

int search_launch(search_meter &m, const shard_view *d_views, uint32_t nshards, int num_cus, const void *d_packed,
                  const void *d_valid, size_t Q, uint32_t k, void *d_lower, void *d_upper, bool counts_only,
                  hipStream_t stream, const search_extra *extra);
int search_host_views(search_meter &m, ctx_pool &pool, const shard_view *d_views, uint32_t nshards, int num_cus,
                      const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *lower, uint64_t *upper,
                      bool counts_only, bool narrow = false);
int search_host_views_var(search_meter &m, ctx_pool &pool, const shard_view *d_views, uint32_t nshards, int num_cus,
                          const char *text, const uint64_t *off, size_t Q, uint64_t *lower, uint64_t *upper, bool counts_only,
                          bool narrow = false);
// search_extra::narrow for a launch on this shard: a T-mer's interval is ~ n / 4^T rows wide;
// a quarter of a window or less = the steps after the table find both positions in one line
inline bool view_is_narrow(const shard_view &v, uint32_t k) {
    return v.ktab && v.ktab_depth >= 2u && k >= v.ktab_depth && ((v.n >> (2u * v.ktab_depth)) << 2) <= v.sp.S;
}
// the worklist launch of a set's 1-mismatch search (kernels.h), metered like search_launch; the counters ACCUMULATE
// onto what the traced launch and the branch kernel left
int search_launch_worklist(search_meter &m, const shard_view *d_views, uint32_t nshards, int num_cus, const void *d_packed,
                           const void *d_valid, size_t nkmers, uint32_t tn, const void *d_worklists, const void *d_counts, size_t wl_cap,
                           uint32_t k, void *d_sparse, void *d_hit_bits, hipStream_t stream, const void *d_pre = nullptr);
// the walk that makes those worklists (kernels.h, launch_search_walk), metered like search_launch: a counting launch
// ZEROES the counters first (it is the first launch of the sequence)
int search_launch_walk(search_meter &m, const shard_view *d_views, uint32_t nshards, int num_cus, const void *d_packed,
                       const void *d_valid, size_t nkmers, uint32_t tn, void *d_worklists, void *d_counts, size_t wl_cap, uint32_t k,
                       void *d_sparse, void *d_hit_bits, hipStream_t stream);
int meter_history_ms(search_meter &m, float *ms, size_t cap, size_t *count);
// 1-mismatch hit list of one shard from variants expanded once for the whole batch (sets.hip: every shard of a set
// searches the same variants)
size_t variants_bytes(size_t m, uint32_t k);
int variants_of_batch_dev(const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_variants, hipStream_t stream);
int meter_work(search_meter &m, uint64_t *words, size_t nwords);

}  // namespace rsb

struct rsbwt : rsb::search_meter {
    int device = 0;          // the GPU (hipSetDevice)
    int logical_device = 0;  // the number the caller opened it with: what a shard set groups by (= device outside tests)
    int num_cus = 256;
    rsb::shard_view view;              // host copy
    rsb::shard_view *d_view = nullptr; // the same in HBM, for kernels that take views from memory
    uint64_t num_runs = 0, num_strings = 0, hbm_bytes = 0;
    uint64_t far_lines = 0, chunk_windows = 0, far_windows = 0, spilled_symbols = 0;
    rsb::ctx_pool pool;
    // What read extraction and getOccAt read: the view PLUS the sampled select table.  `view` / `d_view` -- what every
    // search reads -- are never written after the handle has been handed out (rsbwt_attach_ktab* apart: an explicit
    // step of the owner's).  A shard opened with RSBWT_OPEN_READS has its samples and psi hints before that, and
    // xview == view, d_xview == d_view.  Any other shard builds the samples on its first extraction INTO A SIDE TABLE
    // and publishes them here (x_ready, release / acquire): its lines are not touched, searches that run meanwhile
    // read what they always read.  psi hints go into such a shard's lines only by rsbwt_prepare_extraction, which the
    // owner calls before it shares the handle.
    rsb::shard_view xview;
    rsb::shard_view *d_xview = nullptr;
    std::atomic<bool> x_ready{false};
    uint64_t *d_sel = nullptr;  // sampled select table
    uint64_t psi_hint_lines = 0;  // window lines that carry a psi hint
    bool ktab_owned = true;     // false: view.ktab points into a shard set's interleaved table
    uint64_t ktab_untabulated = 0;  // grouped table: T-mers whose record leaves them to the search (empty, or a group too wide)
};

namespace rsb {
int hits_1mm_dev_shared(rsbwt *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_hits, size_t cap,
                        void *d_total, void *d_scratch, void *stream, const void *d_variants);
// Builds h's k-mer table of depth T into d_table[c * stride] (memory owned by the caller).
// (fmt: rsb::KTAB_PLAIN / KTAB_GROUPED; d_table = this shard's first entry / record inside an interleaved allocation)
int attach_ktab_into(rsbwt *h, uint32_t T, uint64_t *d_table, uint32_t stride, uint32_t fmt = 0);
// RSBWT_KTAB_FORMAT_* -> the format a table of depth T over n symbols gets (auto: the caller's plain / grouped choice)
uint32_t auto_ktab_depth_for(uint64_t budget, uint64_t n, uint32_t *fmt);
constexpr uint32_t KTAB_MAX_DEPTH_PLAIN = 16, KTAB_MAX_DEPTH_GROUPED = 17;
// Where the grouped records pay: a record holds a group of < 16383 rows (so: four siblings expected to hold an eighth of
// that at most), and a T-mer that does not occur is left to the search, T steps instead of none -- rare only while a
// T-mer's interval is still many runs wide (measured on the bench's 1.17e11-symbol run stream: 5.5e-5 of the 15-mers at
// 109 rows each, 0.46 of the 17-mers at 7 rows each: a context that narrow mostly has ONE preceding symbol).
inline bool ktab_grouped_sensible(uint64_t n, uint32_t T) {
    return T >= 2u && T <= 31u && (n >> (2u * (T - 1u))) <= 2048ull && (n >> (2u * T)) >= 64ull;
}
int detach_ktab(rsbwt *h);  // forgets a table it does not own
int ensure_samples(rsbwt *h, hipStream_t stream);  // the select samples of a shard (built once, into a side table): h->xview / d_xview then name them
}  // namespace rsb

#endif

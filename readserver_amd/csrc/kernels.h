// kernels.h -- host-side launchers of the query kernels (kernels.hip) and index builder (index.hip).
#ifndef RSBWT_KERNELS_H
#define RSBWT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "block_format.h"

namespace rsb {

hipError_t launch_pack(const void *d_kmers, size_t Q, uint32_t k, size_t stride, void *d_packed,
                       void *d_valid, hipStream_t stream);
// `sv` may be null or hold no slots: the search then runs on the classic blocks + directory
// ev0/ev1 (optional) are recorded on `stream` immediately around the search kernel itself.
hipError_t launch_search(const rsbwt_view &ix, const slot_view *sv, const void *d_packed, const void *d_valid,
                         size_t Q, uint32_t k, void *d_lower, void *d_upper, bool counts_only,
                         unsigned long long *d_work, int num_cus, hipStream_t stream,
                         hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                         const struct wave_search_extra *extra = nullptr);  // extra: wave kernel only
// search_wave.hip: the wave-cooperative form of the same search (needs slots or dir_shift == 8)
struct wave_search_extra {
    // 1-mismatch search.  A traced search records, per k-mer, the interval it holds
    // when about to take each of its first trace_n symbols ([Q][trace_n] x {lower, upper}); the
    // search of the k-mers' variants (`variants` per k-mer, variants_kernel's order) then starts
    // every variant whose substituted position is < trace_n from that interval.
    void *d_trace_out = nullptr;
    const void *d_trace_in = nullptr;
    uint32_t trace_n = 0, variants = 0;
};
hipError_t launch_search_wave(const rsbwt_view &ix, const slot_view *sv, const void *d_packed,
                              const void *d_valid, size_t Q, uint32_t k, void *d_lower, void *d_upper,
                              bool counts_only, unsigned long long *d_work, int num_cus, hipStream_t stream,
                              hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                              const wave_search_extra *extra = nullptr);
uint32_t wave_trace_entries(const rsbwt_view &ix, uint32_t k);
// does launch_search run the wave kernel on this index?
bool search_uses_wave_kernel(const rsbwt_view &ix, const slot_view *sv);
// slots.hip
bool choose_slot_span(uint64_t n, uint64_t num_runs, uint32_t want_S, slot_params *sp);
hipError_t build_slots(const rsbwt_view &ix, uint64_t num_runs, uint32_t want_S, hipStream_t stream,
                       slot_view *out, uint64_t *bytes, int *range_error);
hipError_t build_ktable(const rsbwt_view &ix, const slot_view *sv, uint32_t T, uint64_t *d_entries,
                        int num_cus, hipStream_t stream);
hipError_t launch_occ_batch(const rsbwt_view &ix, const void *d_syms, const void *d_index, size_t n,
                            void *d_out, hipStream_t stream);
hipError_t launch_char_batch(const rsbwt_view &ix, const void *d_index, size_t n, void *d_out,
                             hipStream_t stream);
hipError_t launch_occ_at_batch(const rsbwt_view &ix, const void *d_syms, const void *d_bc, size_t n,
                               void *d_out, hipStream_t stream);
hipError_t launch_hits1mm_count(const void *d_lower, const void *d_upper, size_t m, uint32_t V, void *d_counts,
                                hipStream_t stream);
hipError_t launch_hits1mm_write(const void *d_lower, const void *d_upper, const void *d_packed, size_t m, uint32_t V,
                                uint32_t k, const void *d_offsets, uint32_t query0, void *d_hits, hipStream_t stream);
hipError_t launch_variants(const void *d_packed, const void *d_valid, size_t Q, uint32_t k, void *d_vpacked,
                           void *d_vvalid, hipStream_t stream);
// read extraction: sampled select table (4 x stride u32) and the two walk kernels
uint64_t select_sample_stride(const rsbwt_view &ix);
hipError_t launch_select_samples(const rsbwt_view &ix, uint32_t *d_sel, hipStream_t stream);
hipError_t launch_extract(const rsbwt_view &ix, const uint32_t *d_sel, const void *d_rows, size_t n,
                          void *d_out, uint32_t stride, void *d_plen, void *d_len, hipStream_t stream);
hipError_t launch_synth_runs(void *d_runs, uint64_t num_runs, uint64_t seed, hipStream_t stream);
hipError_t launch_sample_present(const rsbwt_view &ix, size_t Q, uint32_t k, size_t stride,
                                 uint64_t seed, void *d_kmers, hipStream_t stream);

// index.hip: build blocks + directory in HBM from run bytes in HBM.  On success fills `view`
// (blocks/dir are hipMalloc'ed and owned by the caller).  dir_shift 0 = choose from the mean
// run length.  Synchronises `stream`.
struct build_result {
    rsbwt_view view;
    uint64_t num_runs;
    uint64_t hbm_bytes;
};
hipError_t build_device_index(const void *d_runs, uint64_t num_runs, uint32_t dir_shift,
                              hipStream_t stream, build_result *out, int *range_error);

}  // namespace rsb
#endif

/*
 * rsbwt.h -- C-ABI of the MI355X population-BWT engine (librsbwt.so).
 *
 * Drop-in boundary for ReadServer's src/bwt backward-search path.  Every entry point
 * names the reference interface it replaces (paths relative to the ReadServer tree).
 * Plain pointers and sizes only; no C++ or torch types.  All functions returning int
 * give RSBWT_OK (0) or a negative RSBWT_E* code and never call exit() (the reference
 * exits on a bad file: src/bwt/rlebwt_reader.cpp:31-34).
 *
 * Conventions kept from the reference:
 *   - symbols are ASCII '$','A','C','G','T' (include/bwt/alphabet.h:8-9);
 *   - intervals are inclusive SA-row ranges [lower, upper], empty <=> lower > upper
 *     (include/bwt/query.h:8-11, src/bwt/query.cpp:35);
 *   - a k-mer holding a symbol outside ACGT, or k == 0, yields lower = 1, upper = 0 and
 *     count 0 (callers filter these in the reference: src/service/service.cpp:299-301);
 *   - count = upper >= lower ? upper - lower + 1 : 0 (src/service/service.cpp:304).
 *
 * The engine is the HIP path only: there is no CPU fallback.  Every call that needs the
 * GPU fails with RSBWT_ENODEV when none is usable.
 *
 * Thread safety: a handle may be queried concurrently from several host threads (the
 * reference shares one BWT* across its pool threads, src/service/service.cpp:1513).
 * Host-buffer calls serialise on the handle's internal stream and staging buffers;
 * *_dev calls only enqueue work on the caller's stream.
 */
#ifndef RSBWT_H
#define RSBWT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSBWT_OK 0
#define RSBWT_EINVAL (-1)  /* bad argument                                   */
#define RSBWT_EIO (-2)     /* file missing / short / unreadable              */
#define RSBWT_EFORMAT (-3) /* not an SGA RLE .bwt (magic 0xCACA) or corrupt  */
#define RSBWT_ENOMEM (-4)  /* host or HBM allocation failed                  */
#define RSBWT_ENODEV (-5)  /* no usable HIP device                           */
#define RSBWT_EHIP (-6)    /* a HIP runtime call failed (see rsbwt_last_error) */
#define RSBWT_ERANGE (-7)  /* shard exceeds format limits (2^40 symbols) / buffer too small */
#define RSBWT_ESYS (-8)    /* the host runtime failed (a thread could not be started, ...)   */

/* open flags */
/* bit 0: RSBWT_OPEN_READS -- the shard will serve read extraction (extractPrefix / extractPostfix, query(),
 * src/bwt/query.cpp:43-100: what the reference's find_reads does with every interval): every window line is laid
 * out with room for a psi hint (88 of its 96 piece bytes hold pieces, so ~9 % more lines), the hints and a sparse
 * select-sample table (n / 512 bytes) are built as part of the open, and the index is immutable from then on.
 * Extraction's select then finds its window in the line the walk already stands in: one HBM request per step.
 * Without the flag the layout is the denser one; the first extraction / getOccAt then builds a dense sample
 * table (n / 32 bytes) and writes hints into the lines that happen to have room (about 6 in 10) -- into the
 * resident lines, where no search reads them, but an index a search-only deployment never pays for.
 * bit 1: RSBWT_OPEN_KTAB_GROUPED -- the k-mer table in its grouped format (below, rsbwt_attach_ktab_format): 3 B per
 * T-mer instead of 8, so the same HBM holds a table one level deeper; with depth 0 the deepest grouped table that fits
 * the budget and is deeper than the plain one would be (else the plain one).
 * bits 2..4: reserved (0).
 * bits 5..9: depth T of the k-mer table (4^T entries of 8 B holding findInterval's answer for
 * every T-mer; searches of k >= T symbols start from one lookup).  0 = auto (the deepest table
 * no larger than the index itself nor than a quarter of the free HBM, with 4^T <= n; rsbwt_set_open
 * sizes the tables of one GPU's shards together, out of three quarters of what is free once they are all resident), 31 = no table, else T = 2..16 (T = 16: 34 GB). */
#define RSBWT_OPEN_READS 1u
#define RSBWT_OPEN_KTAB_GROUPED 2u
#define RSBWT_KTAB_SHIFT 5
#define RSBWT_KTAB_MASK (0x1Fu << RSBWT_KTAB_SHIFT)
#define RSBWT_KTAB_NONE (31u << RSBWT_KTAB_SHIFT)
#define RSBWT_KTAB_DEPTH(t) ((uint32_t)(t) << RSBWT_KTAB_SHIFT)
/* bits 12..23: symbols per window of the HBM layout (one 128-byte line per window, so that an Occ
 * lookup is a single request); 0 = chosen from the data: ~88 run pieces per window, shrunk while
 * more than 2.5 % of the positions would lie past their window's line.  2..2944. */
#define RSBWT_SPAN_SHIFT 12
#define RSBWT_SPAN_MASK (0xFFFu << RSBWT_SPAN_SHIFT)
#define RSBWT_SPAN(s) ((uint32_t)(s) << RSBWT_SPAN_SHIFT)

typedef struct rsbwt rsbwt_t;         /* one BWT shard resident in one GPU's HBM */
typedef struct rsbwt_set rsbwt_set_t; /* several shards on this process's GPU(s) */

/* Library / environment ------------------------------------------------------------ */
const char *rsbwt_version(void);
int rsbwt_device_count(void);           /* number of visible HIP devices, 0 if none */
const char *rsbwt_last_error(void);     /* thread-local message of the last failure */
const char *rsbwt_strerror(int code);

/* Index lifetime ---------------------------------------------------------------------
 * rsbwt_open replaces  RLEBWT::RLEBWT(const std::string& filename, int smallSampleRate)
 * (include/bwt/rlebwt.h:17, src/bwt/rlebwt.cpp:11-32): reads the SGA .bwt file
 * (src/bwt/rlebwt_reader.cpp:27-48), uploads the run bytes and builds the device index in
 * HBM.  A "<file>.bpi2" next to it is ignored: the device index is derived from the runs. */
int rsbwt_open(const char *bwt_path, int device, uint32_t flags, rsbwt_t **out);
/* Same from run bytes in host memory (RLUnit bytes, include/bwt/rlunit.h:8-11). */
int rsbwt_open_runs(const uint8_t *runs, uint64_t num_runs, uint64_t num_strings,
                    int device, uint32_t flags, rsbwt_t **out);
/* Same from run bytes already in HBM on `device` (not retained after the call). */
int rsbwt_open_device_runs(const void *d_runs, uint64_t num_runs, uint64_t num_strings,
                           int device, uint32_t flags, rsbwt_t **out);
/* Replaces the owner's unique_ptr<BWT> release (src/service/service.cpp:1513). */
void rsbwt_close(rsbwt_t *h);

/* class BWT mirrors (include/bwt/bwt.h:6-15), one value per call ---------------------- */
uint64_t rsbwt_bwlen(const rsbwt_t *h);               /* BWT::getBWLen, rlebwt.cpp:303-305 */
uint64_t rsbwt_pc(const rsbwt_t *h, char b);          /* BWT::getPC,    rlebwt.cpp:229-231 */
char rsbwt_f(const rsbwt_t *h, uint64_t index);       /* BWT::getF,     rlebwt.cpp:307-314 */
int rsbwt_occ(rsbwt_t *h, char b, uint64_t index, uint64_t *occ);   /* BWT::getOcc,  rlebwt.cpp:268-301 */
int rsbwt_char(rsbwt_t *h, uint64_t index, char *c);                /* BWT::getChar, rlebwt.cpp:202-227 */
int rsbwt_occ_at(rsbwt_t *h, char b, uint64_t bc, uint64_t *index); /* BWT::getOccAt, rlebwt.cpp:233-266 */

/* batched forms of the same (host buffers) */
int rsbwt_occ_batch(rsbwt_t *h, const char *b, const uint64_t *index, size_t n, uint64_t *occ);
int rsbwt_char_batch(rsbwt_t *h, const uint64_t *index, size_t n, char *c);
int rsbwt_occ_at_batch(rsbwt_t *h, const char *b, const uint64_t *bc, size_t n, uint64_t *index);

/* Shape of the resident index */
uint64_t rsbwt_num_runs(const rsbwt_t *h);
uint64_t rsbwt_num_strings(const rsbwt_t *h);
uint64_t rsbwt_num_lines(const rsbwt_t *h);       /* 128-byte lines of the index in HBM */
uint32_t rsbwt_ktab_depth(const rsbwt_t *h);      /* 0 = no k-mer table */
uint32_t rsbwt_window_span(const rsbwt_t *h);     /* symbols per window */
uint64_t rsbwt_far_lines(const rsbwt_t *h);       /* lines that continue windows of > 120 pieces */
uint64_t rsbwt_spilled_symbols(const rsbwt_t *h); /* positions one request past their window's line */
uint64_t rsbwt_hbm_bytes(const rsbwt_t *h);       /* lines + tables */
uint64_t rsbwt_psi_hint_lines(const rsbwt_t *h);  /* window lines carrying a psi hint (without RSBWT_OPEN_READS: 0 until rsbwt_prepare_extraction) */
int rsbwt_opened_for_reads(const rsbwt_t *h);     /* 1: laid out with RSBWT_OPEN_READS */
/* Read extraction and getOccAt need a sampled select table.  A shard opened with RSBWT_OPEN_READS has it (and a psi
 * hint in every window line) when rsbwt_open* returns.  Any other shard builds it on its first extraction / getOccAt
 * call INTO A SIDE TABLE: nothing a search reads -- the lines, the handle's view -- is written after the handle has
 * been handed out, so that first call may run beside searches on other threads (until round 5 it rewrote the
 * handle's view and wrote psi hints into the resident lines).  Such a shard then extracts WITHOUT hints (about 1.9
 * requests per psi step instead of 1.4).  rsbwt_prepare_extraction gives it the samples and the hints its lines have
 * room for (about two thirds of them) in one go; it WRITES INTO THE RESIDENT LINES, so the owner calls it before it
 * shares the handle with other threads, like rsbwt_attach_ktab.  No-op on a shard opened for reads. */
int rsbwt_prepare_extraction(rsbwt_t *h);
/* Builds the k-mer table of depth T (2..16) of an open handle that has none. */
int rsbwt_attach_ktab(rsbwt_t *h, uint32_t T);
/* The same with the table's format named.  The table holds what findInterval (src/bwt/query.cpp:24-41) returns for
 * every T-mer, so that a search of k >= T symbols starts T steps in (the reference has no such table: it takes every
 * step, query.cpp:33-38).  PLAIN: 8 bytes per T-mer {lower:40, width:24}.  GROUPED: the four T-mers that differ in
 * their LAST symbol only are neighbours in the BWT's rows; one 12-byte record holds the first row of the four and their
 * running widths (14 bits each) = 3 bytes per T-mer, T up to 17.  A T-mer the record cannot describe -- one that does
 * not occur (the reference's empty interval depends on the step the search died at), or a group of 16383 rows or more
 * -- is searched from initInterval like an untabulated one: same answers, T more steps; rsbwt_ktab_info counts them.
 * AUTO: grouped where 64 <= n / 4^T and n / 4^(T-1) <= 2048 -- the groups fit their records and a T-mer's interval is
 * still many runs wide, so nearly every T-mer occurs (5.5e-5 of the 15-mers of a 1.17e11-symbol shard do not; at 7 rows
 * per 17-mer 46 % do not, and a plain table answers those in no step at all) -- else plain. */
#define RSBWT_KTAB_FORMAT_PLAIN 0u
#define RSBWT_KTAB_FORMAT_GROUPED 1u
#define RSBWT_KTAB_FORMAT_AUTO 2u
int rsbwt_attach_ktab_format(rsbwt_t *h, uint32_t T, uint32_t format);
/* format (RSBWT_KTAB_FORMAT_PLAIN / _GROUPED), bytes in HBM and -- grouped -- the T-mers left to the search; any may be NULL */
int rsbwt_ktab_info(const rsbwt_t *h, uint32_t *format, uint64_t *bytes, uint64_t *untabulated);
/* Test hook (RSBWT_ENABLE_TEST_HOOKS): n bytes of the resident index (region 0: the lines, 1: an owned k-mer table)
 * copied out, the reading twin of rsbwt_debug_poke. */
int rsbwt_debug_peek(rsbwt_t *h, int region, uint64_t offset, void *bytes, size_t n);
int rsbwt_device(const rsbwt_t *h);          /* the GPU the shard is resident on */
/* The device number the shard was opened with.  Equal to rsbwt_device() except under the TEST HOOK
 * RSBWT_TEST_DEVICE_ALIASES=N (honoured only while RSBWT_ENABLE_TEST_HOOKS is set; read at every rsbwt_open*):
 * device numbers 0..N-1 then name N logical devices dealt round-robin over the physical ones, and a shard set groups
 * its shards by LOGICAL device -- so the several-device host code of a set (a thread, a context pool and a fused
 * launch per device group; the merge of the groups' counts, lists and reads) runs on a box with one GPU.  RCCL is
 * not used between groups that share a GPU (it refuses two ranks on one device): such a set takes the paths of a
 * box without librccl. */
int rsbwt_logical_device(const rsbwt_t *h);

/* query.h mirrors, batched (include/bwt/query.h:18-32) ---------------------------------
 * rsbwt_find_intervals replaces  BWTInterval findInterval(const BWT*, const std::string& w)
 * (src/bwt/query.cpp:24-41) for Q k-mers at once: k-mer q is the k ASCII bytes at
 * kmers + q*stride (stride >= k).  lower/upper receive Q values each. */
int rsbwt_find_intervals(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                         uint64_t *lower, uint64_t *upper);
/* Replaces the count in count_reads (src/service/service.cpp:303-304). */
int rsbwt_count(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                uint64_t *counts);

/* 1-mismatch search (BASELINE configs[3]; not in the reference, defined by composition): for each
 * k-mer, the exact findInterval of the k-mer itself and of each of its 3k single-substitution
 * variants.  Outputs are [Q][3k+1] in canonical order: column 0 = the k-mer, column 1 + 3i + d =
 * position i replaced by the d-th base of ACGT without the original one.  Empty variants have
 * lower > upper, exactly as findInterval leaves them; a k-mer holding a symbol outside ACGT is
 * invalid as a whole (every column lower = 1, upper = 0). */
int rsbwt_find_intervals_1mm(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                             uint64_t *lower, uint64_t *upper);

/* (device-resident searches need 1 <= k <= 65535) */

/* The same search with the output SURVEY 8 f3 defines: only the variants that occur, as a list
 * sorted by (query, pos, base).  The dense [Q][3k+1] matrices above are mostly empty intervals
 * (1.5 KB per 31-mer over PCIe); the list is compacted on the device.  `hits` receives at most
 * `cap` records, *nhits the number found; RSBWT_ERANGE (nothing written) when cap is too small. */
typedef struct rsbwt_hit_1mm {
    uint64_t lower, upper; /* lower <= upper */
    uint32_t query;        /* index of the k-mer in the batch */
    int16_t pos;           /* -1: the k-mer itself, else the substituted position (0 = leftmost) */
    char base;             /* the base put there ('\0' for the k-mer itself) */
    char reserved;
} rsbwt_hit_1mm;
int rsbwt_hits_1mm(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                   rsbwt_hit_1mm *hits, size_t cap, size_t *nhits);

/* Batched read extraction: replaces  extractPrefix(pBWT, row) + extractPostfix(pBWT, row)
 * (src/bwt/query.cpp:43-85; joined as query() does, :94-96) for n SA rows.  Row i's read is written
 * to out + i*stride (no NUL), its length to len[i] and the length of its prefix part to
 * prefix_len[i] (either may be NULL).  A read that does not fit `stride` bytes, or a row >= BWLen,
 * gets len = UINT32_MAX (the reference would loop forever / read out of bounds). */
int rsbwt_extract(rsbwt_t *h, const uint64_t *rows, size_t n, char *out, uint32_t stride,
                  uint32_t *len, uint32_t *prefix_len);

/* query.cpp:102-120  bool query_exactmatch(const BWT*, const string& w), batched: found[q] = 1 when
 * k-mer q is itself one of the indexed reads (w == extractPrefix(i) + extractPostfix(i) for a row i
 * of its interval), else 0 -- also for a string holding a symbol outside ACGT (query.cpp:103-105). */
int rsbwt_query_exactmatch(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                           uint8_t *found);
/* query.cpp:87-100  vector<string> query(const BWT*, const string& w), batched: every read that
 * contains k-mer q, in SA-row order (extractPrefix(i) + extractPostfix(i) for i = lower..upper).
 * first[Q+1] receives the offsets of each k-mer's reads in the output (first[Q] = their number, also
 * stored in *nreads); read r is the read_len[r] bytes at reads + r*read_stride (UINT32_MAX: longer
 * than read_stride).  RSBWT_ERANGE with *nreads set and nothing extracted when cap_reads is too
 * small: call once with cap_reads = 0 to size the buffers. */
int rsbwt_query(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *first,
                char *reads, uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads);

/* Device-resident forms: all pointers are HBM addresses on the handle's device, `stream` is a
 * hipStream_t (NULL = the null stream).  Nothing is synchronised. ------------------------ */
/* ASCII k-mers -> 2-bit packed words (A,C,G,T = 0..3, symbol i at bits 2*(i%32) of word i/32,
 * ceil(k/32) words per k-mer) + one validity byte per k-mer (0 = holds a non-ACGT symbol). */
int rsbwt_pack_kmers_dev(const void *d_kmers, size_t Q, uint32_t k, size_t stride,
                         void *d_packed, void *d_valid, int device, void *stream);
int rsbwt_find_intervals_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q,
                             uint32_t k, void *d_lower, void *d_upper, void *stream);
int rsbwt_count_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                    void *d_counts, void *stream);
/* The same intervals as BWTInterval pairs (include/bwt/query.h:8-11): d_pairs[q] = {lower, upper}, 16
 * bytes per k-mer, written by one store per search (the separate arrays cost two scattered stores). */
int rsbwt_find_interval_pairs_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                                  void *d_pairs, void *stream);
/* Interval pairs for the wire.  Every interval findInterval leaves (empty ones, the (1, 0) of an invalid
 * k-mer and the reference's (0, 2^64-1) corner included) is {lower < 2^40, upper = lower + width - 1 mod 2^64,
 * width < 2^40}, so {lower:40, width:40} carries it exactly: n pairs (16 n bytes, as rsbwt_*_interval_pairs_dev
 * writes them) <-> rsbwt_packed_pairs_bytes(n) = 10 n bytes rounded up to 4.  What a rank sends to the root GPU
 * of a multi-GPU job (5/8 of the bytes over xGMI).  d_unfit (optional u32 in HBM, zeroed by the caller) counts
 * pairs that do not fit the record -- none does for an interval this library produced. */
size_t rsbwt_packed_pairs_bytes(size_t n);
int rsbwt_pack_interval_pairs_dev(const void *d_pairs, size_t n, void *d_packed, void *d_unfit, int device, void *stream);
int rsbwt_unpack_interval_pairs_dev(const void *d_packed, size_t n, void *d_pairs, int device, void *stream);
/* Extracted reads for the wire: [n][stride] ASCII bytes (rsbwt_extract_dev's d_out) + their lengths <-> [n][stride / 4]
 * bytes, 2 bits per base (A, C, G, T = 0..3, base i at bits 2 (i % 4) of byte i / 4; zeros past a read's end and for a
 * read marked UINT32_MAX).  stride % 16 == 0.  What a rank sends of an extraction batch at N > 1: a quarter of the
 * bytes over xGMI; the lengths travel beside it.  Unpacking restores the bytes up to each read's length (NUL beyond). */
int rsbwt_pack_reads_dev(const void *d_reads, const void *d_len, size_t n, uint32_t stride, void *d_packed, int device, void *stream);
int rsbwt_unpack_reads_dev(const void *d_packed, const void *d_len, size_t n, uint32_t stride, void *d_reads, int device, void *stream);
/* 1-mismatch search of m packed k-mers: d_lower/d_upper [m][3k+1] (rsbwt_find_intervals_1mm's layout);
 * d_scratch: rsbwt_1mm_scratch_bytes(h, m, k) bytes. */
size_t rsbwt_1mm_scratch_bytes(const rsbwt_t *h, size_t m, uint32_t k);
int rsbwt_find_intervals_1mm_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k,
                                 void *d_lower, void *d_upper, void *d_scratch, void *stream);
/* The same search leaving only the variants that occur, as a list ordered by search index: record i =
 * {u64 lower, u64 upper, u64 index, u64 0} (32 B), index = q * (3k+1) + v (v = 0: the k-mer itself, else
 * 1 + 3*pos + the rank of the substituted base among the three alternatives) -- the (k-mer, position, base)
 * order of rsbwt_hits_1mm.  At most `cap` records are written to d_hits; *d_total (a u64 in HBM) receives how
 * many there are.  Nothing is written for the variants that end empty -- no [m][3k+1] matrices.
 * d_scratch: rsbwt_hits_1mm_scratch_bytes(h, m, k) bytes.  Nothing is synchronised. */
size_t rsbwt_hits_1mm_scratch_bytes(const rsbwt_t *h, size_t m, uint32_t k);
int rsbwt_hits_1mm_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_hits,
                       size_t cap, void *d_total, void *d_scratch, void *stream);
/* Read extraction of n rows (d_rows: u64): d_out [n][stride] bytes, d_len and d_prefix_len [n] u32. */
int rsbwt_extract_dev(rsbwt_t *h, const void *d_rows, size_t n, void *d_out, uint32_t stride, void *d_len,
                      void *d_prefix_len, void *stream);

/* Measurement hooks (bench.py): wall time of the last search kernel launched through this
 * handle, from HIP events recorded on the launch stream; synchronises on the stop event. */
int rsbwt_last_search_ms(rsbwt_t *h, float *ms);
/* The same for up to `cap` of the most recent launches (the handle keeps 64 event pairs), oldest
 * first; *count receives how many were written. */
int rsbwt_search_history_ms(rsbwt_t *h, float *ms, size_t cap, size_t *count);
/* Exact work counters of the last rsbwt_find_intervals_dev launch when the handle was put
 * in counting mode with rsbwt_set_counting(h, 1): LF steps taken, Occ lookups made, distinct
 * window lines those lookups read (one 128-byte request each; a lookup whose position lies past
 * its window's line costs one more request, counted apart: word 11 of the counters below).
 * Counting mode costs atomics; leave it off when timing. */
int rsbwt_set_counting(rsbwt_t *h, int on);
int rsbwt_last_search_work(rsbwt_t *h, uint64_t *lf_steps, uint64_t *occ_lookups,
                           uint64_t *line_reads);
/* all 16 counter words: 0 LF steps, 1 Occ lookups, 2 window lines read, 3 k-mer-table starts,
 * 4..9 phase cycles (lane-pair kernel only), 10 passes, 11 continuation (spill / far) lines read,
 * 12 = 1 when the launch ran one lane per search (a full batch on a single shard behind a deep k-mer
 * table; csrc/search_solo.h), 0 on lane pairs.
 * After an rsbwt_extract_dev in counting mode the words are the walk kernels' instead: 0..7 the LF
 * (prefix) walk, 8..15 the select (postfix) walk, each {passes, lanes holding a row over those
 * passes, steps completed, lanes on a continuation line, lines fetched, cycles, cycles from issuing
 * a pass's fetches until they landed, count-word probes}, summed over the waves. */
int rsbwt_last_search_counters(rsbwt_t *h, uint64_t *words16);
/* ... and the number of k-mer-table lookups of that launch. */
int rsbwt_last_search_ktab_lookups(rsbwt_t *h, uint64_t *lookups);
/* ... and, for the wave kernel, shader cycles (s_memtime, summed over waves) spent in the six
 * phases of a pass -- set-up, load issue, wait + LDS park, overflow hops, rank, update -- plus the
 * number of passes.  Shares only: the stamps fence the pipeline. */
int rsbwt_last_search_phases(rsbwt_t *h, uint64_t *cycles6, uint64_t *passes);

/* Synthetic data (bench / tests) ----------------------------------------------- */
/* Fill d_runs (HBM) with num_runs pseudo-random RLUnit bytes: the direct run-stream
 * synthesiser for throughput runs.  Same bytes as rsbwt_synth_runs_host for the same seed.
 * Seeds with bit 63 set (RSBWT_SYNTH_LONG_RUNS | seed) give the long-run stream: mostly 31-symbol
 * units of long runs, mean ~25 symbols per unit, as in a deep population BWT. */
#define RSBWT_SYNTH_LONG_RUNS (1ull << 63)
int rsbwt_synth_runs_dev(void *d_runs, uint64_t num_runs, uint64_t seed, int device, void *stream);
int rsbwt_synth_runs_host(uint8_t *runs, uint64_t num_runs, uint64_t seed);
/* A slice of the same stream: d_runs[i] = byte first + i (a 20 GB stream piece by piece, for a checker that has no
 * room for it in HBM). */
int rsbwt_synth_runs_dev_at(void *d_runs, uint64_t first, uint64_t num_runs, uint64_t seed, int device, void *stream);
/* Draw Q k-mers that are present in the index (LF walks from random rows, so every one of the
 * k-1 updateInterval steps keeps a non-empty interval) into d_kmers (ASCII, stride bytes). */
int rsbwt_sample_present_kmers_dev(rsbwt_t *h, size_t Q, uint32_t k, size_t stride, uint64_t seed,
                                   void *d_kmers, void *stream);
/* Build a valid multi-string BWT of a synthetic read population (a base genome, haplotypes with
 * SNPs, fixed-length reads, reverse-lexicographic sort + dedup) and write it as an SGA .bwt;
 * optionally also the sorted reads, one per line.  Host only. */
int rsbwt_synth_popbwt(const char *bwt_path, const char *reads_path, uint64_t seed,
                       uint64_t genome_len, uint32_t haplotypes, double snp_rate,
                       uint32_t read_len, double coverage, int shard, int num_shards);

/* The reference's FM-index file "<bwt>.bpi2"  -------------------------------------
 * Replaces src/util/index_rlebwt.cpp:19-22 (RLEBWT(path) + serialiseFMIndex(path + ".bpi2"),
 * src/bwt/rlebwt.cpp:150-161) for deployments that keep the CPU reference next to this engine:
 * the file written is byte-identical to the reference's for the same .bwt.  Host only; the
 * engine itself never reads a .bpi2 (its device index is derived from the run bytes). */
int rsbwt_bpi2_write(const char *bwt_path, const char *bpi2_path);
/* Validates a prebuilt .bpi2 (what deserialiseFMIndex would load, rlebwt.cpp:163-200) against the
 * resident index: level shapes, C[], vSum, and the absolute counts at the start of up to
 * max_samples evenly spaced 64-run buckets (0 = all) against Occ on the GPU.  *mismatches == 0
 * means the file describes this BWT; rsbwt_last_error() names the first difference. */
int rsbwt_bpi2_check(rsbwt_t *h, const char *bpi2_path, uint64_t max_samples, uint64_t *checked,
                     uint64_t *mismatches);
/* Host only: does the file parse as a .bpi2 (what deserialiseFMIndex reads, rlebwt.cpp:163-200)?
 * RSBWT_OK, RSBWT_EIO or RSBWT_EFORMAT (truncated, trailing bytes, sizes the file cannot back). */
int rsbwt_bpi2_validate_file(const char *bpi2_path);

/* Shard sets: the shards one process holds on its GPU(s), searched with one call ------
 * Every query goes to every shard (src/service/server.cpp:124,578).  The shards of one device are
 * searched by ONE fused launch -- the batch is uploaded and packed once per device -- and devices
 * are driven concurrently.  device_map[i] = HIP device of shard i (NULL: all on device 0); the
 * deployment SURVEY 8e describes is shard s -> GPU s / 8. */
int rsbwt_set_open(const char *const *bwt_paths, size_t num_shards, const int *device_map,
                   uint32_t flags, rsbwt_set_t **out);
int rsbwt_set_from_handles(rsbwt_t *const *handles, size_t num_shards, rsbwt_set_t **out);
void rsbwt_set_close(rsbwt_set_t *s); /* closes the shards it opened itself */
size_t rsbwt_set_size(const rsbwt_set_t *s);
size_t rsbwt_set_devices(const rsbwt_set_t *s);
rsbwt_t *rsbwt_set_shard(rsbwt_set_t *s, size_t i);
/* k-mer tables for the shards that have none; depth 0 = sized per device from its free HBM: the deepest tables that
 * fit three quarters of what is free with the shards resident, leave 8 GiB, and are no larger than 5/4 of a shard's
 * lines (the depth that gives, over all devices: rsbwt_set_auto_ktab_depth) */
int rsbwt_set_attach_ktabs(rsbwt_set_t *s, uint32_t depth);
uint32_t rsbwt_set_auto_ktab_depth(rsbwt_set_t *s);
/* THE sizing rule, with the format: the depth and format (RSBWT_KTAB_FORMAT_PLAIN / _GROUPED) the set's shards that
 * have no table yet would get -- one pair for the whole set, the shallowest over its devices -- when format_in
 * (PLAIN / GROUPED / AUTO) is asked for and keep_free_bytes of each device's free HBM are to stay free (0: the set's
 * own rule, a quarter of what is free and at least 8 GiB).  What rsbwt_set_open and rsbwt_set_attach_ktabs*(depth 0)
 * build, and what bench.py / onehost.py ask instead of restating the rule.  *depth 0 = no table. */
int rsbwt_set_auto_ktab(rsbwt_set_t *s, uint32_t format_in, uint64_t keep_free_bytes, uint32_t *depth, uint32_t *format_out);
/* The same rule as plain arithmetic (host only): the deepest table of at most budget_bytes over a shard of n_symbols --
 * plain: 8 B x 4^T <= budget, 4^T <= n, T <= 16; grouped (3 B x 4^T, T <= 17) only where it is deeper than the plain
 * one, a T-mer still has 64 rows and four siblings fit a record, else plain. */
int rsbwt_auto_ktab_for_budget(uint64_t budget_bytes, uint64_t n_symbols, uint32_t format_in, uint32_t *depth, uint32_t *format_out);
/* the same with the format named (rsbwt_attach_ktab_format; rsbwt_set_attach_ktabs = PLAIN); grouped tables are
 * interleaved like plain ones: a query's records for the S shards of a device are one stretch of 12 * S bytes */
int rsbwt_set_attach_ktabs_format(rsbwt_set_t *s, uint32_t depth, uint32_t format);
/* lower/upper: [num_shards][Q]; counts: [Q] summed over shards, the way the front-end sums
 * per-partition replies (src/service/server.cpp:184-197).  With several devices the per-device sums
 * are reduced onto the first device over RCCL (ncclReduce) and cross PCIe once. */
int rsbwt_set_find_intervals(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k,
                             size_t stride, uint64_t *lower, uint64_t *upper);
int rsbwt_set_count(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride,
                    uint64_t *counts);
/* Device-resident forms for a set on ONE device: one fused launch on `stream`, nothing synchronised.
 * d_lower/d_upper/d_counts: [num_shards][Q]. */
int rsbwt_set_find_intervals_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t Q,
                                 uint32_t k, void *d_lower, void *d_upper, void *stream);
int rsbwt_set_count_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                        void *d_counts, void *stream);
/* 1-mismatch search (SURVEY 8 f3 / BASELINE configs[3]) of m packed k-mers in every shard of a one-device
 * set: d_lower/d_upper [num_shards][m][3k+1]; d_scratch: rsbwt_set_1mm_scratch_bytes(s, m, k) bytes, shared
 * by the shards' searches, which run one after the other on `stream`. */
/* The pair search in two halves, for a pipelined caller: the start records of a batch ([num_shards][Q] x 16 B =
 * rsbwt_set_records_bytes) depend on its k-mers and the k-mer tables only, so batch i + 1's can be computed on a
 * second stream (rsbwt_set_prepare_dev) while batch i is searched; the caller orders the two (an event) and
 * rsbwt_set_find_interval_pairs_prepared_dev then runs the search kernel alone.  Same answers as
 * rsbwt_set_find_interval_pairs_dev (initInterval / the k-mer table: src/bwt/query.cpp:18-21). */
size_t rsbwt_set_records_bytes(const rsbwt_set_t *s, size_t Q);
int rsbwt_set_prepare_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t Q, uint32_t k, void *d_records,
                          void *stream);
int rsbwt_set_find_interval_pairs_prepared_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, const void *d_records,
                                               size_t Q, uint32_t k, void *d_pairs, void *stream);
size_t rsbwt_set_1mm_scratch_bytes(const rsbwt_set_t *s, size_t m, uint32_t k);
int rsbwt_set_find_intervals_1mm_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t m, uint32_t k,
                                     void *d_lower, void *d_upper, void *d_scratch, void *stream);
/* d_pairs: [num_shards][Q] x {lower, upper} */
int rsbwt_set_find_interval_pairs_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t Q,
                                      uint32_t k, void *d_pairs, void *stream);
/* Gathers per-device result blocks onto the set's first device over RCCL / xGMI (ncclSend/ncclRecv in
 * one group): d_blocks[g], bytes[g], streams[g] belong to device g of the set; d_root (first device)
 * receives the blocks back to back.  For GPU-resident consumers of all shards' intervals. */
int rsbwt_set_gather_intervals_dev(rsbwt_set_t *s, const void *const *d_blocks, const size_t *bytes,
                                   void *d_root, void *const *streams);
/* BASELINE configs[3] / configs[4] over a set that may span devices.  The reference's front-end sends every
 * request to every partition and CONCATENATES the per-partition read lists (src/service/server.cpp:124,199-261):
 * the set-level forms are every shard's own result, side by side.  Devices work concurrently, a device's shards
 * take turns; each device's lists / reads cross its own PCIe link straight into the caller's buffers.
 *
 * rsbwt_set_hits_1mm: every shard's rsbwt_hits_1mm list: hits[first[i] .. first[i+1]) = shard i's (first has
 * num_shards + 1 entries; *nhits = first[num_shards]); RSBWT_ERANGE with *nhits and first[] set when cap is short. */
int rsbwt_set_hits_1mm(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride, rsbwt_hit_1mm *hits,
                       size_t cap, uint64_t *first, size_t *nhits);
/* rsbwt_extract for rows of several shards: row i = SA row rows[i] of shard shard_of[i] (the ExtractTask chunks of
 * src/service/service.cpp:729-740 of all partitions in one call). */
int rsbwt_set_extract(rsbwt_set_t *s, const uint32_t *shard_of, const uint64_t *rows, size_t n, char *out, uint32_t stride,
                      uint32_t *len, uint32_t *prefix_len);
/* rsbwt_query over every shard = find_reads of a short query (service.cpp:714-743) in every partition, the lists
 * concatenated as the front-end does: k-mer q's reads are first[q] .. first[q+1], shard 0's first, each shard's in
 * SA-row order; read_shard[r] (optional) names the shard.  cap_reads = 0 sizes the buffers (RSBWT_ERANGE, *nreads set). */
int rsbwt_set_query(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *first,
                    uint32_t *read_shard, char *reads, uint32_t read_stride, uint32_t *read_len, size_t cap_reads,
                    size_t *nreads);
/* Queries of LENGTHS OF THEIR OWN in one call: query q = text[off[q] .. off[q+1]) (off has Q + 1 entries).  What a window of
 * the service loop holds -- the reference answers each request with its own findInterval (src/service/service.cpp:303,
 * src/bwt/query.cpp:24-41), whatever its length -- without a launch sequence per distinct length.  Results as
 * rsbwt_set_find_intervals / rsbwt_set_count / rsbwt_set_query give them for the queries of one length; an empty query,
 * one with a symbol outside ACGT or one longer than 65,535 symbols ends as the empty interval (1, 0) / count 0 / no reads. */
int rsbwt_set_find_intervals_var(rsbwt_set_t *s, const char *text, const uint64_t *off, size_t Q, uint64_t *lower, uint64_t *upper);
int rsbwt_set_count_var(rsbwt_set_t *s, const char *text, const uint64_t *off, size_t Q, uint64_t *counts);
int rsbwt_set_query_var(rsbwt_set_t *s, const char *text, const uint64_t *off, size_t Q, uint64_t *first, uint32_t *read_shard,
                        char *reads, uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads);
/* Device-resident forms, for a set on ONE device (one process per GPU: bench.py --mode 1mm|extract).
 * d_hits [num_shards][cap_per_shard] x 32-byte records (rsbwt_hits_1mm_dev's), d_totals u64[num_shards];
 * d_rows [num_shards][n] (row numbers are per shard), d_out [num_shards][n][stride], d_len / d_prefix_len [num_shards][n]. */
/* (below 2^26 variant searches per shard the shards work side by side on streams of the set, forked from and joined to `stream`, each with a scratch of its
 * own -- rsbwt_set_hits_1mm_scratch_bytes accounts for that) */
size_t rsbwt_set_hits_1mm_scratch_bytes(const rsbwt_set_t *s, size_t m, uint32_t k);
/* 1 when a call of this size searches ALL the set's shards by two launches (one device, k-mer tables of one depth,
 * below the same 2^26 variant searches per shard) instead of shard by shard -- for k <= 32 a walk of the k-mers that
 * also takes the step of the three substitutions at every position left of the tables' reach, then one search of the
 * variants that survived it and of those inside the tables' reach (csrc/search_solo.h WALK / WL, csrc/mm1_worklist.hip);
 * else a traced and a resumed launch: the launches are
 * then metered by the set (rsbwt_set_search_history_ms, rsbwt_set_last_search_counters), not by the shards' handles */
int rsbwt_set_hits_1mm_is_fused(const rsbwt_set_t *s, size_t m, uint32_t k);
int rsbwt_set_hits_1mm_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_hits,
                           size_t cap_per_shard, void *d_totals, void *d_scratch, void *stream);
int rsbwt_set_extract_dev(rsbwt_set_t *s, const void *d_rows, size_t n, void *d_out, uint32_t stride, void *d_len,
                          void *d_prefix_len, void *stream);
int rsbwt_rccl_available(void); /* 1 when librccl could be bound at run time */
/* measurement hooks of a set's fused launches (first device): as the per-handle ones */
int rsbwt_set_set_counting(rsbwt_set_t *s, int on);
int rsbwt_set_search_history_ms(rsbwt_set_t *s, float *ms, size_t cap, size_t *count);
int rsbwt_set_last_search_counters(rsbwt_set_t *s, uint64_t *words16);

/* Service slice: the CountReads / ExactMatch-Count path of the query service --------
 * rsbwt_service_counts replaces, for a batch of serialised `Request` messages
 * (src/service/readserver.proto:3-14; message i = requests[req_off[i] .. req_off[i+1]), offsets
 * ascending and inside requests_len), what the recv loop does per message
 * (src/service/service.cpp:1549-1554,1567-1570 -> count_reads :279-315): two serialised `Reply`
 * messages per request, forward strand then reverse complement, each carrying the original query and
 * an int32 count summed over the set's shards (the front-end only adds partition counts:
 * src/service/server.cpp:184-197; the narrowing to int32, readserver.proto:31-33, applies to the sum:
 * the per-partition form is what rsbwt_service_create(per_partition = 1) sends).  Reply j of request i
 * is replies[rep_off[2i+j] .. rep_off[2i+j+1]); requests of any other type get two empty replies and
 * stay with the caller.  *needed receives the bytes required; RSBWT_ERANGE if cap is too small. */
int rsbwt_service_counts(rsbwt_set_t *set, const uint8_t *requests, size_t requests_len, const uint64_t *req_off,
                         size_t n, uint8_t *replies, size_t cap, uint64_t *rep_off, size_t *needed);
/* The codec on its own (host only). */
int rsbwt_proto_decode_request(const uint8_t *msg, size_t len, int *t, int *rt, const char **q, size_t *qlen);
size_t rsbwt_proto_encode_count_reply(uint8_t *out, size_t cap, int request_type, const char *q, size_t qlen,
                                      int revcomp, int32_t c);
/* Reply{rt = request_type, t = ReplyReads, q, r = ReplyReads{forward_matches | revcomp_matches = ResultReads{r}}*}
 * (src/service/readserver.proto:35-37,39-49,61-64): what QueryTask::run sends for a Request whose return type is
 * Reads (src/service/service.cpp:1260-1291).  `r` is present even with no read (mutable_r(), :1278).  Returns the
 * bytes needed (written when out != NULL and they fit cap); 0 = bad arguments. */
size_t rsbwt_proto_encode_reads_reply(uint8_t *out, size_t cap, int request_type, const char *q, size_t qlen, int revcomp,
                                      const char *const *reads, const size_t *read_len, size_t nreads);

/* The service's configuration file: the libconfig subset the reference's service.cfg uses
 * (`key = "value";`, `key = [ "a", ... ];`, comments; demo/TEMPLATE.service.cfg).  Loading fails with
 * RSBWT_EFORMAT when a setting the reference looks up unconditionally is missing (prefix, suffix,
 * hashfile, pull, push, push_count, rocksdb_path, rocksdb_ext, rocksdb: service.cpp:1425-1442). */
typedef struct rsbwt_service_config rsbwt_service_config_t;
int rsbwt_service_config_load(const char *path, rsbwt_service_config_t **out);
void rsbwt_service_config_free(rsbwt_service_config_t *cfg);
const char *rsbwt_service_config_get(const rsbwt_service_config_t *cfg, const char *key); /* NULL: absent */
size_t rsbwt_service_config_array_len(const rsbwt_service_config_t *cfg, const char *key);
const char *rsbwt_service_config_array_item(const rsbwt_service_config_t *cfg, const char *key, size_t i);

/* Transports: what the loop needs of ZeroMQ -- a SUB socket connected to `pull` and subscribed to
 * everything, PUSH sockets connected to `push` (channel 0) and `push_count` (channel 1)
 * (service.cpp:1493-1502).  In-process: a queue pair for tests and embedding. */
typedef struct rsbwt_transport rsbwt_transport_t;
int rsbwt_transport_inproc(rsbwt_transport_t **out);
/* ZeroMQ (src/service/service.cpp:1493-1502: SUB connect(pull) + subscribe-all, PUSH connect(push), PUSH
 * connect(push_count)).  libzmq is bound at run time (dlopen of libzmq.so.5; RSBWT_LIBZMQ names another):
 * RSBWT_ENODEV on a box without it, no rebuild on one with it. */
int rsbwt_transport_zmq(const char *pull, const char *push, const char *push_count, rsbwt_transport_t **out);
int rsbwt_zmq_available(void); /* 1 when libzmq could be bound */
void rsbwt_transport_free(rsbwt_transport_t *t);
int rsbwt_transport_push_request(rsbwt_transport_t *t, const uint8_t *msg, size_t n); /* in-process only */
int rsbwt_transport_pop_reply(rsbwt_transport_t *t, int channel, uint8_t *buf, size_t cap, size_t *n, int64_t timeout_us);
/* The same in bulk (in-process only; one lock per call instead of one per message): `count` Requests, message j =
 * base[off[j] .. off[j + 1]); up to max_msgs Replies of a channel back to back into buf (message j at off[j] ..
 * off[j + 1], off has max_msgs + 1 entries; *count = how many, 0 after timeout_us without one; a Reply that does not
 * fit `cap` any more stays queued). */
int rsbwt_transport_push_requests(rsbwt_transport_t *t, const uint8_t *base, const uint64_t *off, size_t count);
int rsbwt_transport_pop_replies(rsbwt_transport_t *t, int channel, uint8_t *buf, size_t cap, uint64_t *off, size_t max_msgs,
                                size_t *count, int64_t timeout_us);
void rsbwt_transport_close(rsbwt_transport_t *t); /* the loop ends once what was pushed is answered */

/* The recv loop (service.cpp:1521-1577) with a micro-batch window: the first Request opens a window
 * that closes after window_us or at max_batch messages; all CountReads / ExactMatch-Count requests of
 * the window are answered by one batched search per query length over the set; replies go out in
 * arrival order, forward strand then reverse complement, CountReads on push_count and ExactMatch on
 * push.  per_partition = 1: two replies per request PER SHARD, each what a reference service holding
 * that partition sends (front-end `workers` = 2 x shards); 0: two replies with the counts summed
 * (`workers` = 2).  ExactMatch requests that ask for Reads are answered too (rsbwt_service_set_reads).  Requests of
 * other types go to the handler (may be NULL), which is called on the loop's SENDER thread, not on the thread that
 * called rsbwt_service_run; an exception it throws is caught there and becomes the run's error. */
typedef struct rsbwt_service rsbwt_service_t;
typedef void (*rsbwt_service_other_fn)(void *arg, const uint8_t *request, size_t len);
int rsbwt_service_create(rsbwt_set_t *set, rsbwt_transport_t *t, int64_t window_us, size_t max_batch,
                         int per_partition, rsbwt_service_t **out);
void rsbwt_service_set_other_handler(rsbwt_service_t *s, rsbwt_service_other_fn fn, void *arg);
/* ExactMatch requests whose return type is Reads (`GET /get?output=reads`: the front-end's count pre-flight is
 * followed by this request, and it waits for `workers` replies without a timeout, src/service/server.cpp:469,565-601).
 * The loop answers them itself (default on): find_reads (src/service/service.cpp:714-797) batched over the set --
 * a query shorter than min_read_length: the reads of every row of its interval (in find_reads' order: the tail of an
 * interval wider than 4,097 rows first, then its 2,048-row chunks, :724-751); up to max_read_length: the
 * min_read_length-long tiles of the query that are themselves reads (query_exactmatch, src/bwt/query.cpp:102-120), then
 * every read that contains it (query, :87-100); longer: its max_read_length-long tiles that are reads, then the
 * min_read_length-long ones -- and sends, per partition (or once, summed mode) and strand,
 * Reply{rt = ExactMatch, t = ReplyReads, q, r} on `push` (QueryTask::run, service.cpp:1260-1291), forward strand
 * first.  min / max_read_length: service.cfg's keys (service.cpp:1417-1420; 0 = keep: 73 / 100, :56-57).
 * Tiles are visited in the order of the reference's container (std::unordered_set<std::string>, filled the same
 * way): the same order on the same C++ standard library.  A query holding a symbol outside ACGT matches nothing
 * (find_reads' short-query branch would search it as it stands; count_reads, the pre-flight, answers 0 for it).
 * The return type All needs the RocksDB shards: still the `other` handler's.  enable = 0: Reads requests go to the
 * `other` handler too, as before round 5. */
void rsbwt_service_set_reads(rsbwt_service_t *s, int enable, uint32_t min_read_length, uint32_t max_read_length);
/* service.cfg's `suffix` of every shard of the set (demo/TEMPLATE.service.cfg:16-17), n = rsbwt_set_size (0: none):
 * a tile is looked up only in the partitions whose suffix it ends with (is_suffix_of, service.cpp:228-230,759). */
int rsbwt_service_set_suffixes(rsbwt_service_t *s, const char *const *suffix, size_t n);
uint64_t rsbwt_service_read_requests(const rsbwt_service_t *s); /* Reads requests answered so far */
/* The loop is a pipeline: the thread that runs it receives and cuts the windows, `workers` threads answer a whole
 * window each (decode, one batched search per query length, Reply bytes -- the set's entry points are re-entrant),
 * a sender thread sends the windows' Replies in window order, so replies still leave in arrival order.  Default 8,
 * the threads of the reference's query pool (src/service/service.cpp:88,1505); before rsbwt_service_run / _start. */
void rsbwt_service_set_workers(rsbwt_service_t *s, int workers);
int rsbwt_service_run(rsbwt_service_t *s);   /* on the calling thread, until the transport closes */
int rsbwt_service_start(rsbwt_service_t *s); /* on a thread of its own */
int rsbwt_service_stop(rsbwt_service_t *s);  /* at once; after rsbwt_transport_close: once all that was pushed is answered */
void rsbwt_service_free(rsbwt_service_t *s);
/* {requests, count requests, windows, replies sent, malformed messages, largest window} */
void rsbwt_service_stats(const rsbwt_service_t *s, uint64_t *stats6);

/* Test hook (host only, answers no query): lays `runs` out as window lines with the code the GPU
 * builder runs and holds the layout's scalar readers to naive ranks at every position.  stats6 =
 * {S, lines, far lines, chunk windows, far windows, spilled symbols}; *first_bad = first position
 * that disagrees (RSBWT_EFORMAT) or UINT64_MAX.  Bit 31 of window_span (both hooks): the RSBWT_OPEN_READS
 * layout. */
int rsbwt_layout_selftest_host(const uint8_t *runs, uint64_t num_runs, uint32_t window_span,
                               uint64_t *stats6, uint64_t *first_bad);

/* Test hook (host only, answers no query): the select samples and psi hints of read extraction -- the code the
 * builder kernels and the walk kernels share with the host (csrc/line_format.h) -- built over a host-side layout of
 * `runs` and held to the naive select at EVERY occurrence and EVERY row, then every scalar reader again over the
 * lines that now carry hints.  stats4 = {sample words, occurrences whose sample is only a bound, lines with a
 * hint, rows a hint settles (the others it bounds)}; *first_bad as above. */
int rsbwt_layout_selftest_psi_host(const uint8_t *runs, uint64_t num_runs, uint32_t window_span, uint64_t *stats4,
                                   uint64_t *first_bad);

/* Test hook (host only, answers no query): the grouped k-mer table's record code (rsbwt_attach_ktab_format) -- `groups`
 * x 4 sibling intervals in, the 4 entries each 12-byte record gives back out as {lower:40 | width:24} words, width
 * 0xFFFFFF = left to the search. */
int rsbwt_ktab_group_selftest_host(const uint64_t *lower, const uint64_t *upper, size_t groups, uint64_t *entries);

/* Test hook (answers no query): overwrites n bytes of the index in HBM -- region 0: the window lines,
 * 1: the handle's own k-mer table -- so that tests can hold the kernels to what they do with a DAMAGED
 * index: no read outside the index, every wave drains, a table entry that is not an interval of this BWT
 * is not believed.  RSBWT_ERANGE outside the region.  Refused (RSBWT_EINVAL) unless the environment has
 * RSBWT_ENABLE_TEST_HOOKS set: nothing in a deployment may write into a published index. */
int rsbwt_debug_poke(rsbwt_t *h, int region, uint64_t offset, const void *bytes, size_t n);

/* Test hook (answers no query): the kernels' position -> window division (an f64 multiply and one fix-up step
 * instead of a 64-bit divide) on n host positions: w[i] = p[i] / S, r[i] = p[i] % S, so that a test can hold
 * it to integer division up to the 2^40 symbols a shard may have, for every span S in 2..2944. */
int rsbwt_debug_fast_window(const uint64_t *p, size_t n, uint32_t S, uint32_t *w, uint32_t *r, int device);

#ifdef __cplusplus
}
#endif
#endif

/*
 * rlebwt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11) of ReadServer's src/bwt backward-search path,
 * used as the parity checker for the HIP engine and as the timed "port" CPU
 * baseline in bench.py.  Nothing under readserver_amd/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Parity pinning: this restatement is checked against the reference itself
 * (compiled from /root/reference/src/bwt by oracle/Makefile into
 * oracle/_ref/, build container only) on a fixture in the reference's sound
 * regime; the resulting vectors are committed under tests/golden/ and
 * re-checked on every CPU test run (tests/test_oracle_golden.py).
 *
 * Documented divergences from the reference:
 *   D1  BPTree::rank / BPTree::access skip the top-level search in the last
 *       65,536-symbol window (include/bwt/BPTree.h:84-94,150-157) and return
 *       wrong ranks there on most small indexes.  The oracle always searches.
 *   D2  vSum lacks its final entry when the last run crosses a 65,536-symbol
 *       threshold (src/bwt/rlebwt.cpp:103-106), an out-of-bounds read in the
 *       reference.  The oracle always emits that entry.
 * In both cases the oracle returns the true rank of the BWT it was given.
 */
#ifndef RLEBWT_ORACLE_H
#define RLEBWT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rso_index rso_index;

/* Load an SGA run-length BWT file (src/bwt/rlebwt_reader.cpp:27-48) and build
 * the hierarchical rank index (src/bwt/rlebwt.cpp:34-148).  NULL on error. */
rso_index *rso_load(const char *path);

/* Same, from run bytes already in memory.  If `borrow` is non-zero the oracle
 * keeps the caller's pointer (it must outlive the index) instead of copying. */
rso_index *rso_from_runs(const uint8_t *runs, uint64_t num_runs,
                         uint64_t num_strings, int borrow);

void rso_free(rso_index *ix);

uint64_t rso_num_runs(const rso_index *ix);
uint64_t rso_num_strings(const rso_index *ix);
uint64_t rso_index_bytes(const rso_index *ix); /* markers + vSum, without runs */

/* class BWT mirrors (include/bwt/bwt.h:6-15); b is an ASCII symbol. */
uint64_t rso_bwlen(const rso_index *ix);                       /* getBWLen */
uint64_t rso_pc(const rso_index *ix, char b);                  /* getPC    */
uint64_t rso_occ(const rso_index *ix, char b, uint64_t index); /* getOcc   */
uint64_t rso_occ_at(const rso_index *ix, char b, uint64_t bc); /* getOccAt */
char rso_char(const rso_index *ix, uint64_t index);            /* getChar  */
char rso_f(const rso_index *ix, uint64_t index);               /* getF     */

/* query.h mirrors (src/bwt/query.cpp:11-85). */
void rso_find_interval(const rso_index *ix, const char *w, size_t len,
                       uint64_t *lower, uint64_t *upper);
/* Both return the string length written (no NUL), or (size_t)-1 if `cap`
 * symbols were produced without meeting '$' (the reference would spin). */
size_t rso_extract_prefix(const rso_index *ix, uint64_t index, char *out,
                          size_t cap);
size_t rso_extract_postfix(const rso_index *ix, uint64_t index, char *out,
                           size_t cap);

/* Batched findInterval over Q k-mers laid out at `stride` bytes, split over
 * `nthreads` POSIX threads (strided split, shared read-only index: the way
 * ReadServer's pool threads share one BWT*, src/service/service.cpp:1513).
 * If `steps` is non-NULL it receives the number of updateInterval calls made
 * per query.  K-mers with a symbol outside ACGT get lower=1, upper=0 (callers
 * in the reference never pass them on: src/service/service.cpp:299-301). */
void rso_find_intervals(const rso_index *ix, const char *kmers, size_t Q,
                        uint32_t k, size_t stride, uint64_t *lower,
                        uint64_t *upper, uint8_t *steps, int nthreads);

/* extractPrefix + extractPostfix of n rows over `nthreads` POSIX threads: read i at out + i * stride, len[i]
 * (UINT32_MAX: does not fit stride), prefix_len[i] (may be NULL). */
void rso_extract_batch(const rso_index *ix, const uint64_t *rows, size_t n, char *out, size_t stride, uint32_t *len,
                       uint32_t *prefix_len, int nthreads);

#ifdef __cplusplus
}
#endif
#endif

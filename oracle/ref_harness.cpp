// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C-callable wrapper around the *real* ReadServer src/bwt code, compiled by
// oracle/Makefile from the sources where they lie under /root/reference into
// oracle/_ref/libref_bwt.so (git-ignored; build container only).  It is used
// (1) by tests/golden/make_golden.py to produce the committed golden vectors
// that pin oracle/rlebwt_oracle.c, and (2) optionally as the "reference" CPU
// baseline.  No reference source text lives in this file: it only calls the
// reference's public interface (include/bwt/bwt.h:6-15, query.h:18-32,
// rlebwt.h:17-20).
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "query.h"
#include "rlebwt.h"

extern "C" {

void *ref_open(const char *path) { return new RLEBWT(std::string(path)); }

void ref_close(void *h) { delete static_cast<RLEBWT *>(h); }

// writes "<path>.bpi2"-style index the way src/util/index_rlebwt.cpp:19-22 does
void ref_serialise(void *h, const char *path) {
    static_cast<RLEBWT *>(h)->serialiseFMIndex(std::string(path));
}

uint64_t ref_bwlen(void *h) { return static_cast<BWT *>(static_cast<RLEBWT *>(h))->getBWLen(); }
uint64_t ref_pc(void *h, char b) { return static_cast<RLEBWT *>(h)->getPC(b); }
uint64_t ref_occ(void *h, char b, uint64_t i) { return static_cast<RLEBWT *>(h)->getOcc(b, i); }
uint64_t ref_occ_at(void *h, char b, uint64_t bc) { return static_cast<RLEBWT *>(h)->getOccAt(b, bc); }
char ref_char(void *h, uint64_t i) { return static_cast<RLEBWT *>(h)->getChar(i); }
char ref_f(void *h, uint64_t i) { return static_cast<RLEBWT *>(h)->getF(i); }

void ref_occ_table(void *h, char b, const uint64_t *idx, size_t n, uint64_t *out) {
    const RLEBWT *p = static_cast<RLEBWT *>(h);
    for (size_t i = 0; i < n; ++i) out[i] = p->getOcc(b, idx[i]);
}

void ref_find_interval(void *h, const char *w, size_t len, uint64_t *lo, uint64_t *up) {
    const BWTInterval itv = findInterval(static_cast<RLEBWT *>(h), std::string(w, len));
    *lo = itv.lower;
    *up = itv.upper;
}

void ref_find_intervals(void *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                        uint64_t *lo, uint64_t *up, int nthreads) {
    const BWT *p = static_cast<RLEBWT *>(h);
    auto work = [=](int tid) {
        for (size_t q = tid; q < Q; q += nthreads) {
            const BWTInterval itv = findInterval(p, std::string(kmers + q * stride, k));
            lo[q] = itv.lower;
            up[q] = itv.upper;
        }
    };
    if (nthreads <= 1) { nthreads = 1; work(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto &t : th) t.join();
}

// extractPrefix(row) + extractPostfix(row), as src/bwt/query.cpp:94-96 joins them.
// Only call on an index verified sound (the walk can spin otherwise).
size_t ref_extract(void *h, uint64_t row, char *out, size_t cap, size_t *prefix_len) {
    const BWT *p = static_cast<RLEBWT *>(h);
    const std::string pre = extractPrefix(p, row);
    const std::string post = extractPostfix(p, row);
    if (prefix_len) *prefix_len = pre.size();
    const std::string s = pre + post;
    const size_t n = s.size() < cap ? s.size() : cap;
    memcpy(out, s.data(), n);
    return s.size();
}

// query(pBWT, w) (src/bwt/query.cpp:87-100): the reads containing w, in the order the reference
// returns them, '\n'-terminated into out; returns the bytes needed, *count = number of reads.
size_t ref_query(void *h, const char *w, size_t len, char *out, size_t cap, size_t *count) {
    const std::vector<std::string> seqs = query(static_cast<RLEBWT *>(h), std::string(w, len));
    size_t need = 0;
    for (const std::string &s : seqs) {
        if (need + s.size() + 1 <= cap) {
            memcpy(out + need, s.data(), s.size());
            out[need + s.size()] = '\n';
        }
        need += s.size() + 1;
    }
    if (count) *count = seqs.size();
    return need;
}

// The order in which a std::unordered_set<std::string> filled with the kmer-long substrings of w, position by position
// (what get_tiles does, src/service/service.cpp:232-246 with skip = 0), is ITERATED on this C++ standard library: the
// order find_reads visits its tiles in (service.cpp:758-764).  No reference code: the container is the library's.
// Tiles '\n'-terminated into out; returns the bytes needed, *count = number of distinct tiles.
size_t ref_tiles_order(const char *w, size_t len, size_t kmer, char *out, size_t cap, size_t *count) {
    std::unordered_set<std::string> vs;
    const std::string s(w, len);
    if (kmer != 0 && len >= kmer)
        for (size_t i = 0; i <= len - kmer; ++i) vs.insert(s.substr(i, kmer));
    size_t need = 0;
    for (const std::string &t : vs) {
        if (need + t.size() + 1 <= cap) {
            memcpy(out + need, t.data(), t.size());
            out[need + t.size()] = '\n';
        }
        need += t.size() + 1;
    }
    if (count) *count = vs.size();
    return need;
}

// query_exactmatch(pBWT, w) (src/bwt/query.cpp:102-120)
int ref_query_exactmatch(void *h, const char *w, size_t len) {
    return query_exactmatch(static_cast<RLEBWT *>(h), std::string(w, len)) ? 1 : 0;
}

}  // extern "C"

// service_slice.cpp -- the CountReads / ExactMatch-Count slice of ReadServer's query service
// (SURVEY 8 f1), host side: proto2 wire codec for the two messages involved and the batched
// count_reads (src/service/service.cpp:279-315) over a shard set; and, since round 5, the ExactMatch requests
// whose return type is Reads: find_reads (service.cpp:714-797) batched over the set, the replies of
// QueryTask::run (:1260-1291) -- what `GET /get?output=reads` waits for after its count pre-flight
// (server.cpp:565-601); BWT only (the `all` return type needs the RocksDB shards and stays with the caller).
//
// Wire schema followed: src/service/readserver.proto:3-14 (Request), :31-33 (ResultCount),
// :39-49 (Reply), :56-59 (ReplyCount).  protobuf is not in this image, so the codec is written
// against the proto2 encoding itself (varint keys, length-delimited strings/messages, fields
// emitted in field-number order as protobuf's C++ serialiser does); tests/test_service_slice.py
// compares it byte for byte with the Python protobuf runtime on a re-typed schema.
// Transport (ZeroMQ SUB/PUSH, service.cpp:1493-1502) stays with the caller: INTEGRATION.md.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <thread>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/rsbwt.h"
#include "service.h"
#include "capi_guard.h"

namespace {

struct request_view {
    int t = 0, rt = 0;      // Request.RequestType / ReturnType (readserver.proto:4-5)
    const char *q = nullptr;
    size_t qlen = 0;
    bool has_t = false, has_rt = false, has_q = false;
};

bool get_varint(const uint8_t *&p, const uint8_t *end, uint64_t &v) {
    v = 0;
    for (int shift = 0; p < end && shift < 70; shift += 7) {
        const uint8_t b = *p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

bool decode_request(const uint8_t *p, size_t n, request_view &r) {
    const uint8_t *end = p + n;
    while (p < end) {
        uint64_t key;
        if (!get_varint(p, end, key)) return false;
        const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
        if (wt == 0) {
            uint64_t v;
            if (!get_varint(p, end, v)) return false;
            if (field == 1) { r.t = (int)v; r.has_t = true; }
            else if (field == 2) { r.rt = (int)v; r.has_rt = true; }
        } else if (wt == 2) {
            uint64_t len;
            if (!get_varint(p, end, len) || len > (uint64_t)(end - p)) return false;
            if (field == 3) { r.q = (const char *)p; r.qlen = (size_t)len; r.has_q = true; }
            p += len;
        } else if (wt == 1) {
            if (end - p < 8) return false;
            p += 8;
        } else if (wt == 5) {
            if (end - p < 4) return false;
            p += 4;
        } else {
            return false;
        }
    }
    return r.has_t && r.has_rt && r.has_q;  // all three are `required`
}

inline size_t varint_len(uint64_t v) {
    size_t n = 1;
    while (v >= 0x80) { v >>= 7; ++n; }
    return n;
}

inline uint8_t *put_varint(uint8_t *p, uint64_t v) {
    while (v >= 0x80) { *p++ = (uint8_t)(v | 0x80); v >>= 7; }
    *p++ = (uint8_t)v;
    return p;
}

// Reply{rt, t = ReplyCount, q, c = ReplyCount{forward_matches | revcomp_matches = ResultCount{c}}}
size_t count_reply_len(int request_type, size_t qlen, int32_t c) {
    const size_t result_count = 1 + varint_len((uint64_t)(int64_t)c);  // int32: negative values are sign-extended to 10 bytes
    const size_t reply_count = 1 + varint_len(result_count) + result_count;
    return 1 + varint_len((uint64_t)request_type) + 2 + 1 + varint_len(qlen) + qlen + 1 + varint_len(reply_count) + reply_count;
}

uint8_t *encode_count_reply(uint8_t *p, int request_type, const char *q, size_t qlen, bool revcomp, int32_t c) {
    const size_t result_count = 1 + varint_len((uint64_t)(int64_t)c);
    const size_t reply_count = 1 + varint_len(result_count) + result_count;
    *p++ = 0x08; p = put_varint(p, (uint64_t)request_type);  // rt
    *p++ = 0x10; *p++ = 1;                                    // t = ReplyCount
    *p++ = 0x1A; p = put_varint(p, qlen);
    if (qlen) memcpy(p, q, qlen);
    p += qlen;
    *p++ = 0x22; p = put_varint(p, reply_count);
    *p++ = revcomp ? 0x12 : 0x0A; p = put_varint(p, result_count);
    *p++ = 0x08; p = put_varint(p, (uint64_t)(int64_t)c);
    return p;
}

std::string rev_comp(const char *q, size_t n) {  // service.cpp:251-276
    std::string s(n, 'N');
    for (size_t i = 0; i < n; ++i) {
        const char ch = q[n - 1 - i];
        s[i] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch;
    }
    return s;
}

}  // namespace

extern "C" {

int rsbwt_proto_decode_request(const uint8_t *msg, size_t len, int *t, int *rt, const char **q, size_t *qlen) {
    request_view r;
    if (!msg || !decode_request(msg, len, r)) return RSBWT_EFORMAT;
    if (t) *t = r.t;
    if (rt) *rt = r.rt;
    if (q) *q = r.q;
    if (qlen) *qlen = r.qlen;
    return RSBWT_OK;
}

size_t rsbwt_proto_encode_count_reply(uint8_t *out, size_t cap, int request_type, const char *q, size_t qlen,
                                      int revcomp, int32_t c) {
    if (!q && qlen) return 0;
    const size_t len = count_reply_len(request_type, qlen, c);
    if (out && len <= cap) encode_count_reply(out, request_type, q, qlen, revcomp != 0, c);
    return len;
}

}  // extern "C"

namespace rsb {

// count_reads (src/service/service.cpp:279-315) for a batch of decoded requests over a shard set.
// Every request of type CountReads, or ExactMatch with return type Count, gets its replies; the
// others are left to the caller (handled[i] = 0).  per_partition: one forward and one reverse-
// complement Reply PER SHARD, each carrying that shard's count narrowed to int32 exactly as a
// reference service process holding that partition would send it (readserver.proto:31-33,
// service.cpp:304) -- the front-end adds them up (server.cpp:184-197).  Otherwise two replies per
// request with the counts summed over the shards first (set the front-end's `workers` to 2): the
// narrowing then applies to the sum, which differs from the above only beyond 2^31 matches.
int service_count_batch(rsbwt_set_t *set, const std::vector<service_request> &rq, bool per_partition,
                        reply_arena *replies, std::vector<char> *handled) {
    const size_t n = rq.size();
    const size_t S = rsbwt_set_size(set);
    handled->assign(n, 0);
    replies->bytes.clear();
    replies->off.assign(1, 0);
    replies->first.assign(n + 1, 0);
    std::map<size_t, std::vector<size_t>> by_len;  // one batched search per query length
    for (size_t i = 0; i < n; ++i) {
        const bool count_reads = rq[i].t == 1;                   // Request::CountReads
        const bool exact_count = rq[i].t == 2 && rq[i].rt == 1;  // ExactMatch + Count
        if (!count_reads && !exact_count) continue;               // other paths stay with the caller
        (*handled)[i] = 1;
        by_len[rq[i].q.size()].push_back(i);
    }
    // per request: [strand][shard] counts (one row when summed)
    const size_t rows = per_partition ? S : 1;
    std::vector<std::vector<uint64_t>> cnt(n);
    // A window whose requests have lengths of their own (what a front-end sends: any length up to its limit): ONE search
    // of all of them, both strands -- forward queries, then their reverse complements (rsbwt_set_find_intervals_var /
    // rsbwt_set_count_var: a start record per search says where it goes on).  A window of one length keeps the
    // fixed-length call (its summed form reduces over RCCL on a set of several devices).
    if (by_len.size() > 1) {
        std::vector<size_t> who;
        for (size_t i = 0; i < n; ++i)
            if ((*handled)[i]) {
                cnt[i].assign(2 * rows, 0);
                if (!rq[i].q.empty()) who.push_back(i);  // (empty query: count 0, findInterval on "" is undefined in the reference)
            }
        const size_t m = who.size();
        std::string text;
        std::vector<uint64_t> off(2 * m + 1, 0);
        for (size_t j = 0; j < m; ++j) {
            text += rq[who[j]].q;
            off[j + 1] = text.size();
        }
        for (size_t j = 0; j < m; ++j) {
            text += rev_comp(rq[who[j]].q.data(), rq[who[j]].q.size());
            off[m + j + 1] = text.size();
        }
        if (m != 0 && per_partition) {
            std::vector<uint64_t> lo(S * 2 * m), up(S * 2 * m);
            const int rc = rsbwt_set_find_intervals_var(set, text.data(), off.data(), 2 * m, lo.data(), up.data());
            if (rc != RSBWT_OK) return rc;
            for (size_t s = 0; s < S; ++s)
                for (size_t j = 0; j < 2 * m; ++j) {
                    const uint64_t l = lo[s * 2 * m + j], u = up[s * 2 * m + j];
                    cnt[who[j % m]][(j / m) * rows + s] = u >= l ? u - l + 1 : 0;  // service.cpp:304
                }
        } else if (m != 0) {
            std::vector<uint64_t> c(2 * m);
            const int rc = rsbwt_set_count_var(set, text.data(), off.data(), 2 * m, c.data());
            if (rc != RSBWT_OK) return rc;
            for (size_t j = 0; j < 2 * m; ++j) cnt[who[j % m]][j / m] = c[j];
        }
        by_len.clear();
    }
    for (auto &g : by_len) {
        const size_t k = g.first, m = g.second.size();
        for (size_t j = 0; j < m; ++j) cnt[g.second[j]].assign(2 * rows, 0);
        if (k == 0) continue;  // empty query: count 0 (findInterval on "" is undefined in the reference)
        // both strands of the group in one batch: forward k-mers, then their reverse complements
        std::string flat(2 * m * k, 'N');
        for (size_t j = 0; j < m; ++j) {
            const std::string &q = rq[g.second[j]].q;
            memcpy(&flat[j * k], q.data(), k);
            const std::string rc = rev_comp(q.data(), k);
            memcpy(&flat[(m + j) * k], rc.data(), k);
        }
        int rc;
        if (per_partition) {
            std::vector<uint64_t> lo(S * 2 * m), up(S * 2 * m);
            rc = rsbwt_set_find_intervals(set, flat.data(), 2 * m, (uint32_t)k, k, lo.data(), up.data());
            if (rc != RSBWT_OK) return rc;
            for (size_t s = 0; s < S; ++s)
                for (size_t j = 0; j < 2 * m; ++j) {
                    const uint64_t l = lo[s * 2 * m + j], u = up[s * 2 * m + j];
                    const uint64_t c = u >= l ? u - l + 1 : 0;  // service.cpp:304
                    cnt[g.second[j % m]][(j / m) * rows + s] = c;
                }
        } else {
            std::vector<uint64_t> c(2 * m);
            rc = rsbwt_set_count(set, flat.data(), 2 * m, (uint32_t)k, k, c.data());
            if (rc != RSBWT_OK) return rc;
            for (size_t j = 0; j < 2 * m; ++j) cnt[g.second[j % m]][j / m] = c[j];
        }
    }
    // resultc->set_c(...) narrows the 64-bit count to int32 (readserver.proto:31-33)
    auto narrowed = [&](size_t i, size_t r, int strand) { return (int32_t)(uint32_t)cnt[i][strand * rows + r]; };
    size_t total = 0, messages = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!(*handled)[i]) continue;
        for (size_t r = 0; r < rows; ++r)
            for (int strand = 0; strand < 2; ++strand) total += count_reply_len(rq[i].t, rq[i].q.size(), narrowed(i, r, strand));
        messages += 2 * rows;
    }
    replies->bytes.resize(total);
    replies->off.reserve(messages + 1);
    uint8_t *p = replies->bytes.data();
    for (size_t i = 0; i < n; ++i) {
        replies->first[i] = replies->off.size() - 1;
        if (!(*handled)[i]) continue;
        for (size_t r = 0; r < rows; ++r)
            for (int strand = 0; strand < 2; ++strand) {
                p = encode_count_reply(p, rq[i].t, rq[i].q.data(), rq[i].q.size(), strand == 1, narrowed(i, r, strand));
                replies->off.push_back((size_t)(p - replies->bytes.data()));
            }
    }
    replies->first[n] = replies->off.size() - 1;
    return RSBWT_OK;
}

// ---- ExactMatch + Reads: find_reads (service.cpp:714-797), batched ---------------------------------------------
namespace {

// Reply{rt, t = ReplyReads (2), q, r = ReplyReads{forward_matches | revcomp_matches = ResultReads{r}}*}
// (readserver.proto:35-37,39-49,61-64); `r` is present even when no read matched (QueryTask::run calls
// r.mutable_r() before the loop, service.cpp:1278: an empty sub-message, 2 bytes on the wire)
size_t reads_body_len(const std::vector<const std::string *> &reads) {
    size_t body = 0;
    for (const std::string *s : reads) {
        const size_t rr = 1 + varint_len(s->size()) + s->size();  // ResultReads{r}
        body += 1 + varint_len(rr) + rr;
    }
    return body;
}
size_t reads_reply_len(int request_type, size_t qlen, size_t body) {
    return 1 + varint_len((uint64_t)request_type) + 2 + 1 + varint_len(qlen) + qlen + 1 + varint_len(body) + body;
}
uint8_t *encode_reads_reply(uint8_t *p, int request_type, const std::string &q, bool revcomp, const std::vector<const std::string *> &reads, size_t body) {
    *p++ = 0x08; p = put_varint(p, (uint64_t)request_type);  // rt
    *p++ = 0x10; *p++ = 2;                                    // t = ReplyReads
    *p++ = 0x1A; p = put_varint(p, q.size());
    if (!q.empty()) memcpy(p, q.data(), q.size());
    p += q.size();
    *p++ = 0x2A; p = put_varint(p, body);                     // r
    for (const std::string *s : reads) {
        const size_t rr = 1 + varint_len(s->size()) + s->size();
        *p++ = revcomp ? 0x12 : 0x0A; p = put_varint(p, rr);
        *p++ = 0x0A; p = put_varint(p, s->size());
        if (!s->empty()) memcpy(p, s->data(), s->size());
        p += s->size();
    }
    return p;
}

// if w ends with s (service.cpp:228-230)
inline bool is_suffix_of(const std::string &s, const std::string &w) {
    return s.empty() || (w.size() >= s.size() && memcmp(s.data(), w.data() + (w.size() - s.size()), s.size()) == 0);
}

// get_tiles(w, kmer) (service.cpp:232-250, skip = 0): every kmer-long substring once -- into the SAME container the
// reference uses, filled in the same order, so that iterating it visits the tiles in the order the reference's
// loop does on the same standard library (the order of an unordered_set is the library's, not the standard's)
std::unordered_set<std::string> get_tiles(const std::string &w, size_t kmer) {
    std::unordered_set<std::string> vs;
    if (kmer == 0 || w.size() < kmer) return vs;
    for (size_t i = 0; i <= w.size() - kmer; ++i) vs.insert(w.substr(i, kmer));
    return vs;
}

// The order find_reads leaves the rows of a wide interval in (service.cpp:724-751): while more than 2 x 2,048 rows
// are left a chunk of 2,048 goes to the extraction pool; the rows that remain are extracted on the spot and come
// FIRST, the chunks' reads are appended after them in chunk order.  Returns the number of leading rows that move
// behind the tail (0 for intervals of up to 4,097 rows).
inline size_t chunked_head(size_t n) {
    const size_t large = 2048;  // service.cpp:86
    size_t start = 0;
    while (n != 0 && (n - 1) - start > 2 * large) start += large;
    return start;
}

}  // namespace

void service_reads_empty(const std::vector<service_request> &rq, size_t rows, reply_arena *replies, std::vector<char> *handled) {
    const size_t n = rq.size();
    handled->assign(n, 0);
    replies->bytes.clear();
    replies->off.assign(1, 0);
    replies->first.assign(n + 1, 0);
    const std::vector<const std::string *> none;
    for (size_t i = 0; i < n; ++i) {
        replies->first[i] = replies->off.size() - 1;
        if (!service_is_reads_request(rq[i])) continue;
        (*handled)[i] = 1;
        for (size_t r = 0; r < rows; ++r)
            for (int strand = 0; strand < 2; ++strand) {
                const size_t at = replies->bytes.size(), len = reads_reply_len(rq[i].t, rq[i].q.size(), 0);
                replies->bytes.resize(at + len);
                encode_reads_reply(replies->bytes.data() + at, rq[i].t, rq[i].q, strand == 1, none, 0);
                replies->off.push_back(replies->bytes.size());
            }
    }
    replies->first[n] = replies->off.size() - 1;
}

// fn(i) for i in [0, n), on up to 8 threads (the library's entry points are re-entrant: a stream per calling thread):
// a window's ExactMatch / Reads requests make one engine call per distinct query length -- dozens of small launch
// sequences, each waiting for its own copy back -- and one per (partition, tile length).  The first error wins.
template <class F>
static int for_each_parallel(size_t n, F &&fn) {
    if (n == 0) return RSBWT_OK;
    const size_t T = std::min<size_t>({n, (size_t)8, (size_t)std::max(1u, std::thread::hardware_concurrency())});
    if (T <= 1) {
        for (size_t i = 0; i < n; ++i) {
            const int rc = fn(i);
            if (rc != RSBWT_OK) return rc;
        }
        return RSBWT_OK;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> err{RSBWT_OK};
    std::vector<std::string> msg(T);
    auto work = [&](size_t t) {
        for (size_t i = next.fetch_add(1); i < n && err.load(std::memory_order_relaxed) == RSBWT_OK; i = next.fetch_add(1)) {
            int rc;
            try {
                rc = fn(i);
            } catch (...) {
                rc = RSBWT_ENOMEM;
            }
            if (rc != RSBWT_OK) {
                int none = RSBWT_OK;
                if (err.compare_exchange_strong(none, rc)) msg[t] = rsbwt_last_error();  // (thread-local: carried over below)
            }
        }
    };
    std::vector<std::thread> th;
    try {
        for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
    } catch (...) {  // no more threads to be had: the ones that started and this one do the work
    }
    work(0);
    for (auto &t : th) t.join();
    const int rc = err.load();
    if (rc != RSBWT_OK)
        for (const std::string &m : msg)
            if (!m.empty()) return fail(rc, "%s", m.c_str());
    return rc;
}

int service_reads_batch(rsbwt_set_t *set, const std::vector<service_request> &rq, bool per_partition, const reads_config &cfg,
                        reply_arena *replies, std::vector<char> *handled) {
    const size_t n = rq.size(), S = rsbwt_set_size(set);
    const size_t MINL = cfg.min_read_length, MAXL = cfg.max_read_length;
    handled->assign(n, 0);
    replies->bytes.clear();
    replies->off.assign(1, 0);
    replies->first.assign(n + 1, 0);
    auto suffix_of = [&](size_t p) -> const std::string & {
        static const std::string none;
        return p < cfg.suffix.size() ? cfg.suffix[p] : none;
    };
    // a (request, strand) = one call of find_reads per partition; its result per partition: the tile matches, then the
    // reads of query() / of the interval
    struct job_t {
        size_t req;
        int strand;
        std::string w;
        std::vector<std::vector<std::string>> tiles;  // [shard]: tiles that are reads of that partition, in find_reads' order
        std::vector<std::vector<std::string>> reads;  // [shard]: query(w) / the interval's rows
    };
    std::vector<job_t> jobs;
    for (size_t i = 0; i < n; ++i) {
        if (!service_is_reads_request(rq[i])) continue;
        (*handled)[i] = 1;
        for (int strand = 0; strand < 2; ++strand) {
            job_t j;
            j.req = i;
            j.strand = strand;
            j.w = strand ? rev_comp(rq[i].q.data(), rq[i].q.size()) : rq[i].q;  // QueryTask::run, service.cpp:1268-1272
            j.tiles.resize(S);
            j.reads.resize(S);
            jobs.push_back(std::move(j));
        }
    }
    if (jobs.empty()) {
        replies->first.assign(n + 1, 0);
        return RSBWT_OK;
    }
    // RSBWT_SERVICE_TIMING: a line per window on stderr with what its phases took (diagnosis, tools/README.md)
    static const bool timing = getenv("RSBWT_SERVICE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = timing ? now() : 0.0;
    double t_tiles = 0, t_exact = 0, t_query = 0;
    // ---- the tiles (service.cpp:755-764,773-794): MAX-long ones of a query of MAX or more, then MIN-long ones (of any
    // query longer than MIN; not again when MIN == MAX).  A tile is looked up in the partitions whose suffix it ends
    // with; query_exactmatch (query.cpp:102-120) says whether it is a read there.  One batched call per (partition, length).
    struct cand_t {
        size_t job, seq;  // seq: the tile's place in find_reads' order within the job
        std::string tile;
    };
    std::map<std::pair<size_t, size_t>, std::vector<cand_t>> by_shard_len;  // (shard, tile length) -> candidates
    std::vector<size_t> nseq(jobs.size(), 0);
    for (size_t ji = 0; ji < jobs.size(); ++ji) {
        const std::string &w = jobs[ji].w;
        const size_t sz = w.size();
        std::vector<size_t> lens;
        if (sz >= MAXL) {
            lens.push_back(MAXL);
            if (MINL != MAXL) lens.push_back(MINL);
        } else if (sz >= MINL && sz != MINL) {
            lens.push_back(MINL);
        }
        for (size_t T : lens) {
            const std::unordered_set<std::string> vs = get_tiles(w, T);
            for (const std::string &tile : vs) {
                const size_t seq = nseq[ji]++;
                for (size_t p = 0; p < S; ++p)
                    if (is_suffix_of(suffix_of(p), tile)) by_shard_len[{p, T}].push_back(cand_t{ji, seq, tile});
            }
        }
    }
    if (timing) t_tiles = now();
    std::vector<std::vector<std::pair<size_t, std::pair<size_t, std::string>>>> hits(jobs.size());  // job -> (seq, (shard, tile))
    {
        // the (partition, tile length) groups side by side; what each found is merged afterwards (a job's tiles may
        // sit in several groups)
        std::vector<std::pair<const std::pair<size_t, size_t>, std::vector<cand_t>> *> groups;
        for (auto &g : by_shard_len) groups.push_back(&g);
        std::vector<std::vector<uint8_t>> found(groups.size());
        const int rc = for_each_parallel(groups.size(), [&](size_t gi) -> int {
            const size_t p = groups[gi]->first.first, T = groups[gi]->first.second, m = groups[gi]->second.size();
            if (T == 0 || m == 0) return RSBWT_OK;
            std::string flat(m * T, 'N');
            for (size_t j = 0; j < m; ++j) memcpy(&flat[j * T], groups[gi]->second[j].tile.data(), T);
            found[gi].assign(m, 0);
            return rsbwt_query_exactmatch(rsbwt_set_shard(set, p), flat.data(), m, (uint32_t)T, T, found[gi].data());
        });
        if (rc != RSBWT_OK) return rc;
        for (size_t gi = 0; gi < groups.size(); ++gi)
            for (size_t j = 0; j < found[gi].size(); ++j)
                if (found[gi][j]) hits[groups[gi]->second[j].job].push_back({groups[gi]->second[j].seq, {groups[gi]->first.first, groups[gi]->second[j].tile}});
    }
    for (size_t ji = 0; ji < jobs.size(); ++ji) {
        std::sort(hits[ji].begin(), hits[ji].end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        for (auto &h : hits[ji]) jobs[ji].tiles[h.second.first].push_back(std::move(h.second.second));
    }
    if (timing) t_exact = now();
    // ---- the reads that CONTAIN w: query() for MIN <= |w| < MAX (service.cpp:767; query.cpp:87-100), the interval's
    // rows for |w| < MIN (service.cpp:718-753).
    // The jobs' queries have lengths of their own: rsbwt_set_query_var answers a slice of them in ONE search + ONE
    // extraction (until round 5: a call per distinct length, 75 of them in a window of 4,096 requests).  Slices of 2,048
    // jobs, side by side on the worker's threads.
    std::vector<std::vector<size_t>> slices;
    for (size_t ji = 0; ji < jobs.size(); ++ji) {
        const size_t sz = jobs[ji].w.size();
        if (sz == 0 || sz >= MAXL) continue;
        if (slices.empty() || slices.back().size() >= 2048) slices.emplace_back();
        slices.back().push_back(ji);
    }
    // (the slices write into jobs of their own)
    const int rc_len = for_each_parallel(slices.size(), [&](size_t gi) -> int {
        const std::vector<size_t> &members = slices[gi];
        const size_t m = members.size();
        std::string flat;
        std::vector<uint64_t> qoff(m + 1, 0);
        for (size_t j = 0; j < m; ++j) {
            flat += jobs[members[j]].w;
            qoff[j + 1] = flat.size();
        }
        std::vector<uint64_t> first(m + 1, 0);
        size_t nreads = 0;
        uint32_t stride = (uint32_t)std::max<size_t>(256, (2 * MAXL + 63) & ~(size_t)15);
        std::vector<char> reads;
        std::vector<uint32_t> rlen, rshard;
        size_t room = 16 * m + 256;  // reads the buffers hold: one call answers when they fit, else it says how many there are
        for (int attempt = 0;; ++attempt) {
            int rc = RSBWT_ERANGE;
            for (int sized = 0; sized < 2 && rc == RSBWT_ERANGE; ++sized) {
                reads.assign(room * (size_t)stride, 0);
                rlen.assign(room, 0);
                rshard.assign(room, 0);
                rc = rsbwt_set_query_var(set, flat.data(), qoff.data(), m, first.data(), rshard.data(), reads.data(), stride, rlen.data(), room, &nreads);
                if (rc == RSBWT_ERANGE && nreads > room) room = nreads;
            }
            if (rc != RSBWT_OK) return rc;
            if (nreads == 0) break;
            bool over = false;
            for (size_t r = 0; r < nreads && !over; ++r) over = rlen[r] == 0xFFFFFFFFu;
            if (!over || attempt == 2) break;
            stride = attempt == 0 ? 4096u : 65536u;  // a read longer than the buffer (none in a collection built for these lengths): once more, wider
        }
        for (size_t j = 0; j < m && nreads; ++j) {
            job_t &jb = jobs[members[j]];
            const size_t k = jb.w.size();
            // this query's reads arrive shard by shard (shard 0's first), each shard's in SA-row order
            std::vector<size_t> begin_of(S + 1, 0);
            for (uint64_t r = first[j]; r < first[j + 1]; ++r) begin_of[rshard[r] + 1]++;
            for (size_t p = 0; p < S; ++p) begin_of[p + 1] += begin_of[p];
            for (size_t p = 0; p < S; ++p) {
                const size_t cnt = begin_of[p + 1] - begin_of[p], base = (size_t)first[j] + begin_of[p];
                const size_t head = k < MINL ? chunked_head(cnt) : 0;  // (find_reads' own extraction reorders wide intervals; query() does not)
                std::vector<std::string> &out = jb.reads[p];
                out.reserve(cnt);
                for (size_t t = 0; t < cnt; ++t) {
                    const size_t r = base + (t + head < cnt ? t + head : t + head - cnt);
                    if (rlen[r] == 0xFFFFFFFFu) continue;  // (longer than 64 KB: not a read of this service)
                    out.emplace_back(reads.data() + r * (size_t)stride, rlen[r]);
                }
            }
        }
        return RSBWT_OK;
    });
    if (rc_len != RSBWT_OK) return rc_len;
    if (timing) t_query = now();
    // ---- Reply bytes: request by request, partition by partition (or all partitions' lists joined, shard 0's first),
    // forward then reverse complement
    const size_t rows = per_partition ? S : 1;
    std::vector<std::vector<const std::string *>> lists(jobs.size() * rows);
    for (size_t ji = 0; ji < jobs.size(); ++ji)
        for (size_t p = 0; p < S; ++p) {
            std::vector<const std::string *> &l = lists[ji * rows + (per_partition ? p : 0)];
            // (a partition's list: its tile matches first, then the reads that contain w: service.cpp:757-769)
            for (const std::string &t : jobs[ji].tiles[p]) l.push_back(&t);
            for (const std::string &t : jobs[ji].reads[p]) l.push_back(&t);
        }
    if (!per_partition) {
        // joined lists keep find_reads' order WITHIN a partition; across partitions the front-end concatenates in
        // arrival order (server.cpp:199-261): here shard order, tiles and reads of shard 0 first
    }
    std::vector<size_t> body(lists.size());
    size_t total = 0, messages = 0;
    for (size_t ji = 0; ji < jobs.size(); ++ji)
        for (size_t r = 0; r < rows; ++r) {
            body[ji * rows + r] = reads_body_len(lists[ji * rows + r]);
            total += reads_reply_len(rq[jobs[ji].req].t, rq[jobs[ji].req].q.size(), body[ji * rows + r]);
            ++messages;
        }
    replies->bytes.resize(total);
    replies->off.reserve(messages + 1);
    uint8_t *p = replies->bytes.data();
    size_t ji = 0;
    for (size_t i = 0; i < n; ++i) {
        replies->first[i] = replies->off.size() - 1;
        if (!(*handled)[i]) continue;
        // jobs[ji] = forward, jobs[ji + 1] = reverse complement of request i; per partition: forward, then reverse
        // complement (the count path's order)
        for (size_t r = 0; r < rows; ++r)
            for (int strand = 0; strand < 2; ++strand) {
                const size_t li = (ji + strand) * rows + r;
                p = encode_reads_reply(p, rq[i].t, rq[i].q, strand == 1, lists[li], body[li]);
                replies->off.push_back((size_t)(p - replies->bytes.data()));
            }
        ji += 2;
    }
    replies->first[n] = replies->off.size() - 1;
    if (timing)
        fprintf(stderr, "rsbwt reads window: %zu requests, tiles %.1f ms, exact-match calls %.1f (%zu groups), query calls %.1f (%zu slices), replies %.1f ms (%zu bytes)\n",
                jobs.size() / 2, t_tiles - t_begin, t_exact - t_tiles, by_shard_len.size(), t_query - t_exact, slices.size(), now() - t_query, total);
    return RSBWT_OK;
}

bool service_decode(const uint8_t *msg, size_t len, service_request *out) {
    request_view r;
    if (!decode_request(msg, len, r)) return false;
    out->t = r.t;
    out->rt = r.rt;
    out->q.assign(r.q ? r.q : "", r.qlen);
    return true;
}

}  // namespace rsb

extern "C" {

static int rsbwt_service_counts_body(rsbwt_set_t *set, const uint8_t *requests, size_t requests_len, const uint64_t *req_off,
                         size_t n, uint8_t *replies, size_t cap, uint64_t *rep_off, size_t *needed) {
    if (!set || (!requests && n) || !req_off || !rep_off) return rsb::fail(RSBWT_EINVAL, "null argument");
    for (size_t i = 0; i < n; ++i)
        if (req_off[i] > req_off[i + 1] || req_off[i + 1] > requests_len)
            return rsb::fail(RSBWT_EINVAL, "request %zu: offsets %llu..%llu are not an ascending range inside the %zu-byte buffer",
                             i, (unsigned long long)req_off[i], (unsigned long long)req_off[i + 1], requests_len);
    try {
        std::vector<rsb::service_request> rq(n);
        std::vector<char> parsed(n, 0);
        for (size_t i = 0; i < n; ++i)
            parsed[i] = rsb::service_decode(requests + req_off[i], (size_t)(req_off[i + 1] - req_off[i]), &rq[i]) ? 1 : 0;
        for (size_t i = 0; i < n; ++i)
            if (!parsed[i]) rq[i].t = 0;  // not a Request: no reply
        rsb::reply_arena rep;
        std::vector<char> handled;
        const int rc = rsb::service_count_batch(set, rq, false, &rep, &handled);
        if (rc != RSBWT_OK) return rc;
        const size_t total = rep.bytes.size();
        rep_off[0] = 0;
        for (size_t i = 0; i < n; ++i)
            for (int strand = 0; strand < 2; ++strand)
                rep_off[2 * i + strand + 1] = handled[i] ? rep.off[rep.first[i] + strand + 1] : rep.off[rep.first[i]];
        if (replies && total <= cap && total) memcpy(replies, rep.bytes.data(), total);
        if (needed) *needed = total;
        if ((replies && total <= cap) || total == 0) return RSBWT_OK;
        return rsb::fail(RSBWT_ERANGE, "%zu reply bytes, room for %zu", total, cap);
    } catch (const std::bad_alloc &) {
        return rsb::fail(RSBWT_ENOMEM, "host allocation failed");
    }
}
int rsbwt_service_counts(rsbwt_set_t *set, const uint8_t *requests, size_t requests_len, const uint64_t *req_off,
                         size_t n, uint8_t *replies, size_t cap, uint64_t *rep_off, size_t *needed) {
    return rsb::guarded("rsbwt_service_counts", [&]() -> int { return rsbwt_service_counts_body(set, requests, requests_len, req_off, n, replies, cap, rep_off, needed); });
}


size_t rsbwt_proto_encode_reads_reply(uint8_t *out, size_t cap, int request_type, const char *q, size_t qlen, int revcomp,
                                      const char *const *reads, const size_t *read_len, size_t nreads) {
    if ((!q && qlen) || ((!reads || !read_len) && nreads)) return 0;
    try {
        std::vector<std::string> own(nreads);
        std::vector<const std::string *> l(nreads);
        for (size_t i = 0; i < nreads; ++i) {
            own[i].assign(reads[i] ? reads[i] : "", read_len[i]);
            l[i] = &own[i];
        }
        const std::string qs(q ? q : "", qlen);
        const size_t body = rsb::reads_body_len(l), len = rsb::reads_reply_len(request_type, qlen, body);
        if (out && len <= cap) rsb::encode_reads_reply(out, request_type, qs, revcomp != 0, l, body);
        return len;
    } catch (const std::bad_alloc &) {
        return 0;
    }
}


}  // extern "C"

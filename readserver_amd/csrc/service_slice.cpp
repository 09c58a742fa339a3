// service_slice.cpp -- the CountReads / ExactMatch-Count slice of ReadServer's query service
// (SURVEY 8 f1), host side: proto2 wire codec for the two messages involved and the batched
// count_reads (src/service/service.cpp:279-315) over a shard set.
//
// Wire schema followed: src/service/readserver.proto:3-14 (Request), :31-33 (ResultCount),
// :39-49 (Reply), :56-59 (ReplyCount).  protobuf is not in this image, so the codec is written
// against the proto2 encoding itself (varint keys, length-delimited strings/messages, fields
// emitted in field-number order as protobuf's C++ serialiser does); tests/test_service_slice.py
// compares it byte for byte with the Python protobuf runtime on a re-typed schema.
// Transport (ZeroMQ SUB/PUSH, service.cpp:1493-1502) stays with the caller: INTEGRATION.md.
#include <stdint.h>
#include <string.h>

#include <map>
#include <new>
#include <string>
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


}  // extern "C"

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
#include <string>
#include <vector>

#include "../../include/rsbwt.h"

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

void put_varint(std::vector<uint8_t> &o, uint64_t v) {
    while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; }
    o.push_back((uint8_t)v);
}

// Reply{rt, t = ReplyCount, q, c = ReplyCount{forward_matches | revcomp_matches = ResultCount{c}}}
void encode_count_reply(std::vector<uint8_t> &o, int request_type, const char *q, size_t qlen, bool revcomp,
                        int32_t c) {
    std::vector<uint8_t> rc;  // ResultCount
    rc.push_back(0x08);
    put_varint(rc, (uint64_t)(int64_t)c);  // int32: negative values are sign-extended to 10 bytes
    std::vector<uint8_t> rcount;  // ReplyCount
    rcount.push_back(revcomp ? 0x12 : 0x0A);
    put_varint(rcount, rc.size());
    rcount.insert(rcount.end(), rc.begin(), rc.end());
    o.push_back(0x08); put_varint(o, (uint64_t)request_type);  // rt
    o.push_back(0x10); put_varint(o, 1);                        // t = ReplyCount
    o.push_back(0x1A); put_varint(o, qlen); o.insert(o.end(), q, q + qlen);
    o.push_back(0x22); put_varint(o, rcount.size()); o.insert(o.end(), rcount.begin(), rcount.end());
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
    std::vector<uint8_t> o;
    encode_count_reply(o, request_type, q, qlen, revcomp != 0, c);
    if (out && o.size() <= cap) memcpy(out, o.data(), o.size());
    return o.size();
}

int rsbwt_service_counts(rsbwt_set_t *set, const uint8_t *requests, const uint64_t *req_off, size_t n,
                         uint8_t *replies, size_t cap, uint64_t *rep_off, size_t *needed) {
    if (!set || (!requests && n) || !req_off || !rep_off) return RSBWT_EINVAL;
    std::vector<request_view> rq(n);
    std::vector<char> handled(n, 0);
    // group the count requests by query length: one batched call per length and strand
    std::map<size_t, std::vector<size_t>> by_len;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *m = requests + req_off[i];
        if (!decode_request(m, (size_t)(req_off[i + 1] - req_off[i]), rq[i])) continue;
        const bool count_reads = rq[i].t == 1;                      // Request::CountReads
        const bool exact_count = rq[i].t == 2 && rq[i].rt == 1;     // ExactMatch + Count
        if (!count_reads && !exact_count) continue;                  // other paths stay with the caller
        handled[i] = 1;
        by_len[rq[i].qlen].push_back(i);
    }
    std::vector<uint64_t> fwd(n, 0), rev(n, 0);
    for (auto &g : by_len) {
        const size_t k = g.first, m = g.second.size();
        if (k == 0) continue;  // empty query: count 0 (findInterval on "" is undefined in the reference)
        std::string flat_f(m * k, 'N'), flat_r(m * k, 'N');
        for (size_t j = 0; j < m; ++j) {
            const request_view &r = rq[g.second[j]];
            memcpy(&flat_f[j * k], r.q, k);
            const std::string rc = rev_comp(r.q, k);
            memcpy(&flat_r[j * k], rc.data(), k);
        }
        std::vector<uint64_t> cf(m), cr(m);
        int rc = rsbwt_set_count(set, flat_f.data(), m, (uint32_t)k, k, cf.data());
        if (rc == RSBWT_OK) rc = rsbwt_set_count(set, flat_r.data(), m, (uint32_t)k, k, cr.data());
        if (rc != RSBWT_OK) return rc;
        for (size_t j = 0; j < m; ++j) { fwd[g.second[j]] = cf[j]; rev[g.second[j]] = cr[j]; }
    }
    std::vector<uint8_t> o;
    size_t total = 0;
    rep_off[0] = 0;
    for (size_t i = 0; i < n; ++i) {
        for (int strand = 0; strand < 2; ++strand) {
            if (handled[i]) {
                o.clear();
                // resultc->set_c(...) narrows the 64-bit count to int32 (readserver.proto:31-33)
                const int32_t c = (int32_t)(uint32_t)(strand ? rev[i] : fwd[i]);
                encode_count_reply(o, rq[i].t, rq[i].q, rq[i].qlen, strand == 1, c);
                if (replies && total + o.size() <= cap) memcpy(replies + total, o.data(), o.size());
                total += o.size();
            }
            rep_off[2 * i + strand + 1] = total;
        }
    }
    if (needed) *needed = total;
    return (replies && total <= cap) || total == 0 ? RSBWT_OK : RSBWT_ERANGE;
}

}  // extern "C"

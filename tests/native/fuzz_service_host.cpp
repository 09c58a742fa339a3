// fuzz_service_host.cpp -- AddressSanitizer / UBSan run of the host-side parsers of the service slice
// (csrc/service_slice.cpp: proto2 Request decoder, Reply encoder; csrc/service_loop.cpp: service.cfg
// reader) on random and mutated inputs, and of find_reads' host side (service_reads_batch: tiles, suffix
// filter, chunked order, reply arena) over a STUB engine that hands back made-up reads -- some empty, some
// longer than the buffer (the retry with a wider one).  CPU only: the library calls those files make into the
// GPU engine are stubbed here.  Built and run by tests/test_service_slice.py.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "rsbwt.h"
#include "../../readserver_amd/csrc/service.h"

namespace rsb {
int fail(int code, const char *, ...) { return code; }
}  // namespace rsb
static size_t g_shards = 1;
static uint64_t mix(uint64_t x) {
    x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29;
    return x;
}
extern "C" {
size_t rsbwt_set_size(const rsbwt_set_t *) { return g_shards; }
rsbwt_t *rsbwt_set_shard(rsbwt_set_t *, size_t i) { return (rsbwt_t *)(uintptr_t)(i + 1); }
// stub engine: a tile "is a read" by a hash of its bytes and the shard
int rsbwt_query_exactmatch(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride, uint8_t *found) {
    for (size_t q = 0; q < Q; ++q) {
        uint64_t x = (uint64_t)(uintptr_t)h;
        for (uint32_t i = 0; i < k; ++i) x = mix(x + (uint8_t)kmers[q * stride + i]);
        found[q] = (x & 3) == 0;
    }
    return RSBWT_OK;
}
// stub engine: k-mer q has (hash % 6) reads in every shard; one in 50 is reported longer than a 512-byte buffer; an
// interval of 5,000 rows now and then (the chunked order)
int rsbwt_set_query(rsbwt_set_t *, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *first, uint32_t *read_shard,
                    char *reads, uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads) {
    size_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        size_t at = 0;
        for (size_t q = 0; q < Q; ++q) {
            uint64_t x = 7;
            for (uint32_t i = 0; i < k; ++i) x = mix(x + (uint8_t)kmers[q * stride + i]);
            first[q] = at;
            for (size_t s = 0; s < g_shards; ++s) {
                const size_t cnt = (x % 97 == 0) ? 5000 : (size_t)(mix(x + s) % 6);
                for (size_t r = 0; r < cnt; ++r, ++at) {
                    if (pass == 0 || cap_reads == 0) continue;
                    const uint64_t y = mix(x + 131 * s + r);
                    uint32_t len = (uint32_t)(y % 120);
                    if (y % 50 == 0 && read_stride < 4096) len = 0xFFFFFFFFu;
                    read_len[at] = len;
                    if (read_shard) read_shard[at] = (uint32_t)s;
                    if (len != 0xFFFFFFFFu)
                        for (uint32_t i = 0; i < len; ++i) reads[at * (size_t)read_stride + i] = "ACGT"[(y >> (i % 60)) & 3];
                }
            }
        }
        first[Q] = at;
        total = at;
        if (cap_reads == 0 || total > cap_reads) break;
    }
    *nreads = total;
    return (cap_reads == 0 || total > cap_reads) && total ? RSBWT_ERANGE : RSBWT_OK;
}
// (the mixed-length forms: the same made-up reads, keyed on each query's own bytes)
int rsbwt_set_query_var(rsbwt_set_t *s, const char *text, const uint64_t *off, size_t Q, uint64_t *first, uint32_t *read_shard, char *reads,
                        uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads) {
    size_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        size_t at = 0;
        for (size_t q = 0; q < Q; ++q) {
            uint64_t x = 7;
            for (uint64_t i = off[q]; i < off[q + 1]; ++i) x = mix(x + (uint8_t)text[i]);
            first[q] = at;
            for (size_t sh = 0; sh < g_shards; ++sh) {
                const size_t cnt = (x % 97 == 0) ? 5000 : (size_t)(mix(x + sh) % 6);
                for (size_t r = 0; r < cnt; ++r, ++at) {
                    if (pass == 0 || cap_reads == 0) continue;
                    const uint64_t y = mix(x + 131 * sh + r);
                    uint32_t len = (uint32_t)(y % 120);
                    if (y % 50 == 0 && read_stride < 4096) len = 0xFFFFFFFFu;
                    read_len[at] = len;
                    if (read_shard) read_shard[at] = (uint32_t)sh;
                    if (len != 0xFFFFFFFFu)
                        for (uint32_t i = 0; i < len; ++i) reads[at * (size_t)read_stride + i] = "ACGT"[(y >> (i % 60)) & 3];
                }
            }
        }
        first[Q] = at;
        total = at;
        if (cap_reads == 0 || total > cap_reads) break;
    }
    (void)s;
    *nreads = total;
    return (cap_reads == 0 || total > cap_reads) && total ? RSBWT_ERANGE : RSBWT_OK;
}
int rsbwt_set_find_intervals_var(rsbwt_set_t *, const char *, const uint64_t *, size_t, uint64_t *, uint64_t *) { return RSBWT_ENODEV; }
int rsbwt_set_count_var(rsbwt_set_t *, const char *, const uint64_t *, size_t, uint64_t *) { return RSBWT_ENODEV; }
int rsbwt_set_find_intervals(rsbwt_set_t *, const char *, size_t, uint32_t, size_t, uint64_t *, uint64_t *) { return RSBWT_ENODEV; }
int rsbwt_set_count(rsbwt_set_t *, const char *, size_t, uint32_t, size_t, uint64_t *) { return RSBWT_ENODEV; }
const char *rsbwt_last_error(void) { return ""; }
}

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return s;
}

int main(int argc, char **argv) {
    const size_t iters = argc > 1 ? (size_t)atof(argv[1]) : 300000;
    const char *tmp = argc > 2 ? argv[2] : "/tmp/fuzz_service.cfg";
    const uint8_t good[] = {0x08, 0x01, 0x10, 0x01, 0x1A, 0x05, 'A', 'C', 'G', 'T', 'A'};
    size_t accepted = 0;
    for (size_t i = 0; i < iters; ++i) {
        std::vector<uint8_t> b;
        if (i & 1) {
            b.resize(rnd() % 48);
            for (auto &x : b) x = (uint8_t)rnd();
        } else {
            b.assign(good, good + sizeof good);
            for (int m = 1 + (int)(rnd() % 3); m > 0; --m) b[rnd() % b.size()] = (uint8_t)rnd();
            b.resize(rnd() % (b.size() + 1));
        }
        // exact-size heap copy: any read past the end is an ASan report
        uint8_t *p = (uint8_t *)malloc(b.size() ? b.size() : 1);
        if (!b.empty()) memcpy(p, b.data(), b.size());
        int t = 0, rt = 0;
        const char *q = nullptr;
        size_t ql = 0;
        if (rsbwt_proto_decode_request(p, b.size(), &t, &rt, &q, &ql) == RSBWT_OK) {
            ++accepted;
            if (q < (const char *)p || q + ql > (const char *)p + b.size()) { fprintf(stderr, "query view outside the message\n"); return 1; }
            // what the service would answer with: encode into an exact-size buffer
            const int32_t c = (int32_t)rnd();
            const size_t need = rsbwt_proto_encode_count_reply(nullptr, 0, t, q, ql, (int)(i & 1), c);
            uint8_t *o = (uint8_t *)malloc(need ? need : 1);
            if (rsbwt_proto_encode_count_reply(o, need, t, q, ql, (int)(i & 1), c) != need) { fprintf(stderr, "encoder length changed\n"); return 1; }
            free(o);
        }
        free(p);
    }
    // find_reads' host side over the stub engine: random Reads requests, read lengths and suffixes; every arena must be
    // well-formed (offsets ascending, 2 x rows messages per request, each message a parsable Reply that says ExactMatch /
    // ReplyReads and carries the request's query)
    size_t read_msgs = 0;
    for (size_t round = 0; round < iters / 3000 + 3; ++round) {
        g_shards = 1 + rnd() % 4;
        rsb::reads_config cfg;
        cfg.min_read_length = 3 + rnd() % 60;
        cfg.max_read_length = cfg.min_read_length + rnd() % 40;
        for (size_t sidx = 0; sidx < g_shards && (round & 1); ++sidx) cfg.suffix.push_back(std::string(rnd() % 3, "ACGT"[rnd() % 4]));
        std::vector<rsb::service_request> rq(1 + rnd() % 40);
        for (auto &r : rq) {
            r.t = (rnd() & 1) ? 2 : 1 + (int)(rnd() % 3);
            r.rt = (rnd() & 3) ? 2 : 1 + (int)(rnd() % 3);
            r.q.resize(rnd() % 180);
            for (auto &c : r.q) c = "ACGTACGTACGTN"[rnd() % 13];
        }
        for (int per = 0; per < 2; ++per) {
            rsb::reply_arena rep;
            std::vector<char> handled;
            if (rsb::service_reads_batch((rsbwt_set_t *)0x1, rq, per != 0, cfg, &rep, &handled) != RSBWT_OK) { fprintf(stderr, "reads batch failed\n"); return 1; }
            const size_t rows = per ? g_shards : 1;
            if (rep.first.size() != rq.size() + 1 || rep.off.empty() || rep.off[0] != 0 || rep.off.back() != rep.bytes.size()) { fprintf(stderr, "arena shape\n"); return 1; }
            for (size_t i = 0; i < rq.size(); ++i) {
                const bool is_reads = rq[i].t == 2 && rq[i].rt == 2;
                if ((handled[i] != 0) != is_reads || rep.first[i + 1] - rep.first[i] != (is_reads ? 2 * rows : 0)) { fprintf(stderr, "messages per request\n"); return 1; }
                for (size_t j = rep.first[i]; j < rep.first[i + 1]; ++j) {
                    if (rep.off[j + 1] < rep.off[j]) { fprintf(stderr, "offsets\n"); return 1; }
                    // exact-size heap copy of the message, then a field walk: rt = 2, t = 2, q, r
                    const size_t n = rep.off[j + 1] - rep.off[j];
                    uint8_t *m = (uint8_t *)malloc(n ? n : 1);
                    memcpy(m, rep.bytes.data() + rep.off[j], n);
                    size_t at = 0;
                    auto varint = [&](uint64_t *v) { *v = 0; for (int sh = 0; at < n; sh += 7) { const uint8_t b = m[at++]; *v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) return true; } return false; };
                    uint64_t v = 0;
                    bool ok = n >= 6 && m[at++] == 0x08 && varint(&v) && v == 2 && m[at++] == 0x10 && varint(&v) && v == 2 && m[at++] == 0x1A && varint(&v) &&
                              v == rq[i].q.size() && at + v <= n && memcmp(m + at, rq[i].q.data(), v) == 0;
                    at += ok ? v : 0;
                    ok = ok && at < n && m[at++] == 0x2A && varint(&v) && at + v == n;
                    while (ok && at < n) {  // ReplyReads: repeated ResultReads, all on one strand's field
                        const uint8_t tag = m[at++];
                        uint64_t l1 = 0, l2 = 0;
                        ok = (tag == 0x0A || tag == 0x12) && varint(&l1) && at + l1 <= n && m[at++] == 0x0A && varint(&l2) && at + l2 <= n;
                        at += ok ? l2 : 0;
                    }
                    free(m);
                    if (!ok) { fprintf(stderr, "a Reply does not parse\n"); return 1; }
                    ++read_msgs;
                }
            }
        }
    }
    g_shards = 1;
    printf("%zu read replies checked\n", read_msgs);
    const std::string base =
        "prefix = \"p\"; suffix = \"s\"; hashfile = \"h\"; pull = \"a\"; push = \"b\"; push_count = \"c\";\n"
        "rocksdb_path = \"r\"; rocksdb_ext = \".db\"; rocksdb = [ \"x\", \"y\" ]; // c\n/* d */ # e\n";
    size_t loaded = 0;
    for (size_t i = 0; i < iters / 100; ++i) {
        std::string t = base;
        for (int m = (int)(rnd() % 5); m > 0; --m) t[rnd() % t.size()] = (char)(1 + rnd() % 255);
        t.resize(1 + rnd() % t.size());
        FILE *f = fopen(tmp, "wb");
        if (!f) return 2;
        fwrite(t.data(), 1, t.size(), f);
        fclose(f);
        rsbwt_service_config_t *cfg = nullptr;
        if (rsbwt_service_config_load(tmp, &cfg) == RSBWT_OK) {
            ++loaded;
            (void)rsbwt_service_config_get(cfg, "pull");
            for (size_t k = 0; k <= rsbwt_service_config_array_len(cfg, "rocksdb"); ++k) (void)rsbwt_service_config_array_item(cfg, "rocksdb", k);
            rsbwt_service_config_free(cfg);
        } else if (cfg) {
            fprintf(stderr, "a handle came back with an error\n");
            return 1;
        }
    }
    remove(tmp);
    printf("%zu messages, %zu accepted; %zu config files, %zu loaded\n", iters, accepted, iters / 100, loaded);
    return 0;
}

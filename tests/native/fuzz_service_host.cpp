// fuzz_service_host.cpp -- AddressSanitizer / UBSan run of the host-side parsers of the service slice
// (csrc/service_slice.cpp: proto2 Request decoder, Reply encoder; csrc/service_loop.cpp: service.cfg
// reader) on random and mutated inputs.  CPU only: the library calls those files make into the GPU
// engine are stubbed here (never reached by the parsers).  Built and run by tests/test_service_slice.py.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "rsbwt.h"

namespace rsb {
int fail(int code, const char *, ...) { return code; }
}  // namespace rsb
extern "C" {
size_t rsbwt_set_size(const rsbwt_set_t *) { return 1; }
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

// tsan_service_loop.cpp -- ThreadSanitizer run of the service loop (csrc/service_loop.cpp) and the
// in-process transport: a producer thread pushes Requests, the loop thread batches and answers them, a
// consumer thread pops Replies, a fourth thread polls the statistics; then the transport closes and the
// loop drains.  The GPU engine behind the loop is stubbed (every count 0, two made-up reads per k-mer and partition):
// what is checked is the loop's own locking, and that exactly 2 x partitions Replies leave per count or Reads
// Request.  CPU only.
// Built and run by tests/test_service_slice.py.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "rsbwt.h"

namespace rsb {
int fail(int code, const char *, ...) { return code; }
}  // namespace rsb
static const size_t PARTS = 3;
extern "C" {
size_t rsbwt_set_size(const rsbwt_set_t *) { return PARTS; }
int rsbwt_set_find_intervals(rsbwt_set_t *, const char *, size_t Q, uint32_t, size_t, uint64_t *lo, uint64_t *up) {
    for (size_t i = 0; i < PARTS * Q; ++i) { lo[i] = 1; up[i] = 0; }
    return RSBWT_OK;
}
int rsbwt_set_count(rsbwt_set_t *, const char *, size_t Q, uint32_t, size_t, uint64_t *c) {
    for (size_t i = 0; i < Q; ++i) c[i] = 0;
    return RSBWT_OK;
}
int rsbwt_set_find_intervals_var(rsbwt_set_t *, const char *, const uint64_t *, size_t Q, uint64_t *lo, uint64_t *up) {
    for (size_t i = 0; i < PARTS * Q; ++i) { lo[i] = 1; up[i] = 0; }
    return RSBWT_OK;
}
int rsbwt_set_count_var(rsbwt_set_t *, const char *, const uint64_t *, size_t Q, uint64_t *c) {
    for (size_t i = 0; i < Q; ++i) c[i] = 0;
    return RSBWT_OK;
}
const char *rsbwt_last_error(void) { return ""; }
rsbwt_t *rsbwt_set_shard(rsbwt_set_t *, size_t i) { return (rsbwt_t *)(uintptr_t)(i + 1); }
int rsbwt_query_exactmatch(rsbwt_t *, const char *, size_t Q, uint32_t, size_t, uint8_t *found) {
    for (size_t q = 0; q < Q; ++q) found[q] = 0;
    return RSBWT_OK;
}
// two 6-base reads per k-mer and partition
int rsbwt_set_query(rsbwt_set_t *, const char *, size_t Q, uint32_t, size_t, uint64_t *first, uint32_t *read_shard, char *reads,
                    uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads) {
    const size_t total = Q * PARTS * 2;
    for (size_t q = 0; q <= Q; ++q) first[q] = q * PARTS * 2;
    *nreads = total;
    if (cap_reads < total) return RSBWT_ERANGE;
    for (size_t r = 0; r < total; ++r) {
        memcpy(reads + r * (size_t)read_stride, "ACGTAC", 6);
        read_len[r] = 6;
        if (read_shard) read_shard[r] = (uint32_t)((r / 2) % PARTS);
    }
    return RSBWT_OK;
}
int rsbwt_set_query_var(rsbwt_set_t *, const char *, const uint64_t *, size_t Q, uint64_t *first, uint32_t *read_shard, char *reads,
                        uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads) {
    const size_t total = Q * PARTS * 2;
    for (size_t q = 0; q <= Q; ++q) first[q] = q * PARTS * 2;
    *nreads = total;
    if (cap_reads < total) return RSBWT_ERANGE;
    for (size_t r = 0; r < total; ++r) {
        memcpy(reads + r * (size_t)read_stride, "ACGTAC", 6);
        read_len[r] = 6;
        if (read_shard) read_shard[r] = (uint32_t)((r / 2) % PARTS);
    }
    return RSBWT_OK;
}
}

int main(int argc, char **argv) {
    const size_t N = argc > 1 ? (size_t)atof(argv[1]) : 20000;
    rsbwt_transport_t *tr = nullptr;
    rsbwt_service_t *svc = nullptr;
    if (rsbwt_transport_inproc(&tr) || rsbwt_service_create((rsbwt_set_t *)0x1, tr, 100, 512, 1, &svc) || rsbwt_service_start(svc)) return 2;
    std::atomic<bool> done{false};
    std::thread producer([&] {
        for (size_t i = 0; i < N; ++i) {
            std::string m("\x08\x01\x10\x01\x1A\x05", 6);
            if (i % 7 == 3) m[1] = 0x02;  // ExactMatch + Count: answered on the other socket
            if (i % 11 == 5) { m[1] = 0x02; m[3] = 0x02; }  // ExactMatch + Reads: find_reads' host side on a worker thread, woven into the window
            m += "ACGTA";
            m[6 + i % 5] = "ACGT"[i % 4];
            rsbwt_transport_push_request(tr, (const uint8_t *)m.data(), m.size());
            if (i % 1000 == 999) std::this_thread::yield();
        }
    });
    size_t got[2] = {0, 0};
    std::thread consumer([&] {
        uint8_t buf[256];
        size_t n = 0;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(300);
        while (got[0] + got[1] < 2 * PARTS * N && std::chrono::steady_clock::now() < deadline) {
            // drain what is there on either socket; wait (briefly) only when both are empty
            while (rsbwt_transport_pop_reply(tr, 1, buf, sizeof buf, &n, 0) == RSBWT_OK) ++got[1];
            while (rsbwt_transport_pop_reply(tr, 0, buf, sizeof buf, &n, 0) == RSBWT_OK) ++got[0];
            if (got[0] + got[1] < 2 * PARTS * N && rsbwt_transport_pop_reply(tr, 1, buf, sizeof buf, &n, 500) == RSBWT_OK) ++got[1];
        }
    });
    std::thread watcher([&] {
        uint64_t st[6];
        while (!done.load()) {
            rsbwt_service_stats(svc, st);
            std::this_thread::yield();
        }
    });
    producer.join();
    consumer.join();
    rsbwt_transport_close(tr);
    const int rc = rsbwt_service_stop(svc);
    done.store(true);
    watcher.join();
    uint64_t st[6];
    rsbwt_service_stats(svc, st);
    rsbwt_service_free(svc);
    rsbwt_transport_free(tr);
    const size_t exact = (N + 3) / 7;  // i % 7 == 3
    printf("%zu requests, %zu + %zu replies, %llu windows\n", N, got[0], got[1], (unsigned long long)st[2]);
    if (rc != RSBWT_OK || got[0] + got[1] != 2 * PARTS * N || st[0] != N || st[3] != 2 * PARTS * N) return 1;
    (void)exact;
    return 0;
}

// service_bench.cpp -- throughput of the f1 service loop (csrc/service_loop.cpp) end to end inside one
// process: a producer thread pushes serialised CountReads Requests into the in-process transport, the
// loop batches them (window / max batch), searches all shards of the set on the GPU and sends 2 x P
// Replies per request, a consumer thread pops them.  What a ZeroMQ deployment adds is the sockets.
//   tools/bin/service_bench [requests=200000] [shards=1] [run_bytes=2e8] [window_us=200] [max_batch=4096] [closed=0] [workers=8] [zmq=0]
// zmq=1: the same open loop over REAL ZeroMQ sockets on tcp loopback -- this program plays the front-end (binds a PUB
// and two PULL sockets, src/service/server.cpp:118-124), the service connects SUB / PUSH / PUSH (libzmq bound at run
// time on both sides).
// closed=1: one request in flight at a time (push, wait for its 2 x P replies): the latency of a lone
// request, printed beside the latency of the library calls under it (rsbwt_set_find_intervals and
// rsbwt_find_intervals with one k-mer).
// build: g++ -O2 -std=c++17 -Iinclude tools/service_bench.cpp -Lreadserver_amd/lib -lrsbwt -lpthread
//            -Wl,-rpath,'$ORIGIN/../../readserver_amd/lib' -Wl,-rpath,/opt/rocm/lib -o tools/bin/service_bench
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "rsbwt.h"

static uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    const size_t N = argc > 1 ? (size_t)atof(argv[1]) : 200000;
    const size_t P = argc > 2 ? (size_t)atoi(argv[2]) : 1;
    const uint64_t R = argc > 3 ? (uint64_t)atof(argv[3]) : 200000000ull;
    const int64_t window = argc > 4 ? atoll(argv[4]) : 200;
    const size_t max_batch = argc > 5 ? (size_t)atoll(argv[5]) : 4096;
    const bool closed_loop = argc > 6 && atoi(argv[6]) != 0;
    const int workers = argc > 7 ? atoi(argv[7]) : 8;
    const bool over_zmq = argc > 8 && atoi(argv[8]) != 0;
    const uint32_t k = 31;
    std::vector<rsbwt_t *> shards;
    {
        std::vector<uint8_t> runs(R);
        for (size_t s = 0; s < P; ++s) {
            rsbwt_synth_runs_host(runs.data(), R, 4242 + s);
            rsbwt_t *h = nullptr;
            if (rsbwt_open_runs(runs.data(), R, 0, 0, 0, &h) != RSBWT_OK) {
                fprintf(stderr, "%s\n", rsbwt_last_error());
                return 1;
            }
            shards.push_back(h);
        }
    }
    rsbwt_set_t *set = nullptr;
    rsbwt_transport_t *tr = nullptr;
    rsbwt_service_t *svc = nullptr;
    // zmq=1: the front-end's sockets, bound here on tcp loopback
    struct {
        void *lib = nullptr, *ctx = nullptr, *pub = nullptr, *pull = nullptr, *pull_count = nullptr;
        void *(*ctx_new)() = nullptr;
        void *(*socket)(void *, int) = nullptr;
        int (*bind)(void *, const char *) = nullptr;
        int (*getsockopt)(void *, int, void *, size_t *) = nullptr;
        int (*setsockopt)(void *, int, const void *, size_t) = nullptr;
        int (*send)(void *, const void *, size_t, int) = nullptr;
        int (*recv)(void *, void *, size_t, int) = nullptr;
    } z;
    char ep[3][256] = {{0}, {0}, {0}};
    if (over_zmq) {
        for (const char *name : {(const char *)getenv("RSBWT_LIBZMQ"), "libzmq.so.5", "libzmq.so", "/usr/local/lib/libzmq.so.5", "/opt/conda/lib/libzmq.so.5"})
            if (name && *name && !z.lib) z.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!z.lib || !rsbwt_zmq_available()) { fprintf(stderr, "no libzmq here\n"); return 3; }
#define ZS(f, n) z.f = reinterpret_cast<decltype(z.f)>(dlsym(z.lib, n))
        ZS(ctx_new, "zmq_ctx_new"); ZS(socket, "zmq_socket"); ZS(bind, "zmq_bind"); ZS(getsockopt, "zmq_getsockopt");
        ZS(setsockopt, "zmq_setsockopt"); ZS(send, "zmq_send"); ZS(recv, "zmq_recv");
#undef ZS
        z.ctx = z.ctx_new();
        void **socks[3] = {&z.pub, &z.pull, &z.pull_count};
        const int types[3] = {1 /* ZMQ_PUB */, 7 /* ZMQ_PULL */, 7};
        for (int i = 0; i < 3; ++i) {
            *socks[i] = z.socket(z.ctx, types[i]);
            const int zero = 0, tmo = 20000, hwm = 0;
            z.setsockopt(*socks[i], 17 /* ZMQ_LINGER */, &zero, sizeof zero);
            z.setsockopt(*socks[i], 27 /* ZMQ_RCVTIMEO */, &tmo, sizeof tmo);
            z.setsockopt(*socks[i], 23 /* ZMQ_SNDHWM */, &hwm, sizeof hwm);  // (an open loop: nothing may be dropped at the publisher)
            z.setsockopt(*socks[i], 24 /* ZMQ_RCVHWM */, &hwm, sizeof hwm);
            size_t n = sizeof ep[i];
            if (z.bind(*socks[i], "tcp://127.0.0.1:*") != 0 || z.getsockopt(*socks[i], 32 /* ZMQ_LAST_ENDPOINT */, ep[i], &n) != 0) {
                fprintf(stderr, "cannot bind the front-end's sockets\n");
                return 3;
            }
        }
    }
    if (rsbwt_set_from_handles(shards.data(), P, &set) || (over_zmq ? rsbwt_transport_zmq(ep[0], ep[1], ep[2], &tr) : rsbwt_transport_inproc(&tr)) ||
        rsbwt_service_create(set, tr, window, max_batch, 1, &svc)) {
        fprintf(stderr, "%s\n", rsbwt_last_error());
        return 1;
    }
    rsbwt_service_set_workers(svc, workers);
    if (rsbwt_service_start(svc)) {
        fprintf(stderr, "%s\n", rsbwt_last_error());
        return 1;
    }
    // Request{t = CountReads, rt = Count, q = 31-mer}: 08 01 10 01 1A 1F <31 bytes>
    std::vector<std::string> msgs(N);
    for (size_t i = 0; i < N; ++i) {
        std::string m("\x08\x01\x10\x01\x1A\x1F", 6);
        uint64_t h = mix(i);
        for (uint32_t j = 0; j < k; ++j) {
            if ((j & 31) == 31) h = mix(h);
            m.push_back("ACGT"[(h >> (2 * (j & 31))) & 3]);
        }
        msgs[i] = m;
    }
    if (closed_loop) {
        auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        auto summary = [](std::vector<double> &v, double *mean, double *p50, double *p99) {
            size_t slow = 0, worst = 0;
            for (size_t i = 0; i < v.size(); ++i) {
                slow += v[i] > 1000.0;
                if (v[i] > v[worst]) worst = i;
            }
            fprintf(stderr, "  %zu samples, %zu over 1 ms, worst %.0f us at sample %zu\n", v.size(), slow, v[worst], worst);
            std::sort(v.begin(), v.end());
            double s = 0;
            for (double x : v) s += x;
            *mean = s / v.size();
            *p50 = v[v.size() / 2];
            *p99 = v[std::min(v.size() - 1, v.size() * 99 / 100)];
        };
        // a clock watcher: if it sees the same gap as a slow call, the process was not running at all
        std::atomic<bool> watch_stop{false};
        double watch_worst = 0;
        std::thread watcher([&] {
            double last = now_us();
            while (!watch_stop.load()) {
                std::this_thread::sleep_for(std::chrono::microseconds(200));
                const double t = now_us();
                if (t - last > watch_worst) watch_worst = t - last;
                last = t;
            }
        });
        std::vector<double> svc_us, set_us, one_us;
        uint8_t buf[512];
        size_t n = 0;
        for (size_t i = 0; i < N; ++i) {
            const double t = now_us();
            rsbwt_transport_push_request(tr, (const uint8_t *)msgs[i].data(), msgs[i].size());
            for (size_t r = 0; r < 2 * P; ++r)
                if (rsbwt_transport_pop_reply(tr, 1, buf, sizeof buf, &n, 30000000) != RSBWT_OK) return 2;
            if (i >= 16) svc_us.push_back(now_us() - t);
        }
        std::vector<uint64_t> lo(2 * P), up(2 * P);
        for (size_t i = 0; i < N; ++i) {  // each entry by itself, so that neither sees the other's stream
            const char *q = msgs[i].data() + 6;
            const double t = now_us();
            if (rsbwt_find_intervals(shards[0], q, 1, k, k, lo.data(), up.data()) != RSBWT_OK) return 2;
            if (i >= 16) one_us.push_back(now_us() - t);
        }
        for (size_t i = 0; i < N; ++i) {
            const char *q = msgs[i].data() + 6;
            const double t = now_us();
            if (rsbwt_set_find_intervals(set, q, 1, k, k, lo.data(), up.data()) != RSBWT_OK) return 2;
            if (i >= 16) set_us.push_back(now_us() - t);
        }
        watch_stop.store(true);
        watcher.join();
        fprintf(stderr, "  clock watcher's longest gap between 200 us sleeps: %.0f us\n", watch_worst);
        double a[3], b[3], c[3];
        summary(svc_us, &a[0], &a[1], &a[2]);
        summary(set_us, &b[0], &b[1], &b[2]);
        summary(one_us, &c[0], &c[1], &c[2]);
        printf("{\"closed_loop_requests\": %zu, \"partitions\": %zu, \"window_us\": %lld, \"max_batch\": %zu, "
               "\"service_us\": {\"mean\": %.1f, \"p50\": %.1f, \"p99\": %.1f}, "
               "\"set_find_intervals_1_us\": {\"mean\": %.1f, \"p50\": %.1f, \"p99\": %.1f}, "
               "\"find_intervals_1_us\": {\"mean\": %.1f, \"p50\": %.1f, \"p99\": %.1f}}\n",
               N, P, (long long)window, max_batch, a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2]);
        rsbwt_transport_close(tr);
        rsbwt_service_stop(svc);
        rsbwt_service_free(svc);
        rsbwt_transport_free(tr);
        rsbwt_set_close(set);
        for (rsbwt_t *h : shards) rsbwt_close(h);
        return 0;
    }
    if (over_zmq) {
        // PUB/SUB drops what is published before the subscription has arrived: probe until one is answered, then drain
        uint8_t buf[512];
        bool up = false;
        const int short_tmo = 200, long_tmo = 20000;
        z.setsockopt(z.pull_count, 27, &short_tmo, sizeof short_tmo);
        for (int tries = 0; tries < 100 && !up; ++tries) {
            z.send(z.pub, msgs[0].data(), msgs[0].size(), 0);
            up = z.recv(z.pull_count, buf, sizeof buf, 0) >= 0;
        }
        if (!up) { fprintf(stderr, "the service never subscribed\n"); return 3; }
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        while (z.recv(z.pull_count, buf, sizeof buf, 0) >= 0) {}
        z.setsockopt(z.pull_count, 27, &long_tmo, sizeof long_tmo);
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::thread producer([&] {
        if (over_zmq) {
            for (size_t i = 0; i < N; ++i) z.send(z.pub, msgs[i].data(), msgs[i].size(), 0);
            return;
        }
        // the in-process transport in bulk: a few hundred Requests per call
        const size_t CH = 512;
        std::vector<uint8_t> flat;
        std::vector<uint64_t> off;
        for (size_t i0 = 0; i0 < N; i0 += CH) {
            const size_t m = std::min(CH, N - i0);
            flat.clear();
            off.assign(1, 0);
            for (size_t i = i0; i < i0 + m; ++i) {
                flat.insert(flat.end(), msgs[i].begin(), msgs[i].end());
                off.push_back(flat.size());
            }
            rsbwt_transport_push_requests(tr, flat.data(), off.data(), m);
        }
    });
    size_t got = 0, bytes = 0;
    std::thread consumer([&] {
        if (over_zmq) {
            uint8_t buf[512];
            while (got < 2 * P * N) {
                const int n = z.recv(z.pull_count, buf, sizeof buf, 0);
                if (n < 0) break;
                ++got;
                bytes += (size_t)n;
            }
            return;
        }
        std::vector<uint8_t> buf(1 << 20);
        std::vector<uint64_t> off(8193);
        size_t n = 0;
        while (got < 2 * P * N) {
            if (rsbwt_transport_pop_replies(tr, 1, buf.data(), buf.size(), off.data(), 8192, &n, 30000000) != RSBWT_OK || n == 0) break;
            got += n;
            bytes += off[n];
        }
    });
    producer.join();
    consumer.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    rsbwt_transport_close(tr);
    rsbwt_service_stop(svc);
    uint64_t st[6];
    rsbwt_service_stats(svc, st);
    printf("{\"requests\": %zu, \"partitions\": %zu, \"replies\": %zu, \"reply_bytes\": %zu, \"seconds\": %.4f, "
           "\"requests_per_s\": %.1f, \"searches_per_s\": %.1f, \"windows\": %llu, \"mean_requests_per_window\": %.1f, "
           "\"largest_window\": %llu, \"window_us\": %lld, \"max_batch\": %zu, \"run_bytes_per_shard\": %llu, \"workers\": %d, "
           "\"transport\": \"%s\"}\n",
           N, P, got, bytes, dt, N / dt, 2.0 * P * N / dt, (unsigned long long)st[2], (double)st[0] / (double)(st[2] ? st[2] : 1),
           (unsigned long long)st[5], (long long)window, max_batch, (unsigned long long)R, workers,
           over_zmq ? "ZeroMQ PUB/SUB + PUSH/PULL over tcp loopback (this program = the front-end)" : "in-process queue pair, bulk push / pop");
    rsbwt_service_free(svc);
    rsbwt_transport_free(tr);
    rsbwt_set_close(set);
    for (rsbwt_t *h : shards) rsbwt_close(h);
    return got == 2 * P * N ? 0 : 2;
}

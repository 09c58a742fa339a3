// service_loop.cpp -- the recv loop of ReadServer's query service for the CountReads /
// ExactMatch-Count slice, with the batching the GPU engine needs.
//
// Reference: `service` main (src/service/service.cpp:1366-1583): reads its libconfig file
// (:1379-1442), connects a SUB socket to `pull` (subscribing to everything) and PUSH sockets to
// `push` and `push_count` (:1493-1502), then loops forever (:1521-1577): recv one Request, and for
// CountReads answer inline on push_count, forward then reverse complement (:1567-1570); for
// ExactMatch with return type Count queue two CountTasks that answer on push (:1549-1554).  Exactly
// two replies per request per partition: the front-end waits for `workers` of them
// (server.cpp:403,469) and has no timeout.
//
// Here the loop gathers requests for a MICRO-BATCH WINDOW (the first message opens it; it closes
// after window_us or at max_batch messages, whichever comes first), answers all count requests of the
// window with one batched search per query length over the shard set (service_slice.cpp), and sends
// the replies in arrival order.  Requests of other types are handed to the caller's handler (the
// RocksDB / alignment paths stay with the reference's code).  The transport is an interface: an
// in-process queue pair for tests and embedding, and ZeroMQ -- libzmq is BOUND AT RUN TIME (dlopen, as
// sets.hip binds RCCL): the same librsbwt.so serves a box with libzmq.so.5 and reports RSBWT_ENODEV on
// one without; no zmq.h is needed to build (the handful of libzmq 4.x entry points and constants used
// are declared below).
#include <dlfcn.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rsbwt.h"
#include "service.h"

#include "capi_guard.h"

using namespace rsb;

// ---- service.cfg: the subset of libconfig the reference's files use -------------------------------
// `key = "value";` and `key = [ "a", "b", ... ];` (the `;` optional), `//`, `#` and /* */ comments;
// every value is a string (service.cpp:1407-1442 converts with atoi where it needs a number).
struct rsbwt_service_config {
    std::map<std::string, std::string> scalars;
    std::map<std::string, std::vector<std::string>> arrays;
};

namespace {

struct cfg_lexer {
    const std::string &s;
    size_t i = 0;
    int line = 1;
    explicit cfg_lexer(const std::string &text) : s(text) {}
    void skip() {
        for (;;) {
            while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\n' || s[i] == ';' || s[i] == ',')) {
                if (s[i] == '\n') ++line;
                ++i;
            }
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') ++i; continue; }
            if (i < s.size() && s[i] == '#') { while (i < s.size() && s[i] != '\n') ++i; continue; }
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') {
                i += 2;
                while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) { if (s[i] == '\n') ++line; ++i; }
                i = i + 2 <= s.size() ? i + 2 : s.size();
                continue;
            }
            return;
        }
    }
    bool name(std::string *out) {
        const size_t a = i;
        while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-' || s[i] == '*')) ++i;
        *out = s.substr(a, i - a);
        return i > a;
    }
    bool string_value(std::string *out) {  // adjacent "a" "b" concatenate, as in libconfig
        out->clear();
        bool any = false;
        for (;;) {
            skip_ws_only();
            if (i >= s.size() || s[i] != '"') return any;
            ++i;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) {
                    ++i;
                    const char c = s[i];
                    out->push_back(c == 'n' ? '\n' : c == 't' ? '\t' : c == 'r' ? '\r' : c);
                } else {
                    if (s[i] == '\n') ++line;
                    out->push_back(s[i]);
                }
                ++i;
            }
            if (i >= s.size()) return false;
            ++i;
            any = true;
        }
    }
    void skip_ws_only() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\n')) {
            if (s[i] == '\n') ++line;
            ++i;
        }
    }
};

int parse_config(const std::string &text, rsbwt_service_config *cfg, std::string *err) {
    cfg_lexer lx(text);
    for (;;) {
        lx.skip();
        if (lx.i >= text.size()) return RSBWT_OK;
        std::string key;
        if (!lx.name(&key)) { *err = "line " + std::to_string(lx.line) + ": a setting name was expected"; return RSBWT_EFORMAT; }
        lx.skip_ws_only();
        if (lx.i >= text.size() || (text[lx.i] != '=' && text[lx.i] != ':')) {
            *err = "line " + std::to_string(lx.line) + ": '=' expected after " + key;
            return RSBWT_EFORMAT;
        }
        ++lx.i;
        lx.skip_ws_only();
        if (lx.i < text.size() && (text[lx.i] == '[' || text[lx.i] == '(')) {
            const char close = text[lx.i] == '[' ? ']' : ')';
            ++lx.i;
            std::vector<std::string> items;
            for (;;) {
                lx.skip();
                if (lx.i < text.size() && text[lx.i] == close) { ++lx.i; break; }
                std::string v;
                if (!lx.string_value(&v)) { *err = "line " + std::to_string(lx.line) + ": a quoted string was expected in " + key; return RSBWT_EFORMAT; }
                items.push_back(v);
            }
            cfg->arrays[key] = items;
        } else {
            std::string v;
            if (!lx.string_value(&v)) { *err = "line " + std::to_string(lx.line) + ": a quoted string was expected for " + key; return RSBWT_EFORMAT; }
            cfg->scalars[key] = v;
        }
    }
}

// ---- in-process transport ---------------------------------------------------------------------------
class inproc_transport : public transport {
  public:
    bool recv(std::vector<uint8_t> *msg, int64_t timeout_us) override {
        std::unique_lock<std::mutex> lock(mu_);
        auto ready = [&] { return !in_.empty() || closed_; };
        if (timeout_us < 0) cv_.wait(lock, ready);
        else if (!cv_.wait_for(lock, std::chrono::microseconds(timeout_us), ready)) return false;
        if (in_.empty()) return false;
        *msg = std::move(in_.front());
        in_.pop_front();
        return true;
    }
    size_t recv_many(std::vector<std::vector<uint8_t>> *out, size_t max, int64_t timeout_us) override {
        std::unique_lock<std::mutex> lock(mu_);
        auto ready = [&] { return !in_.empty() || closed_; };
        if (timeout_us < 0) cv_.wait(lock, ready);
        else if (!cv_.wait_for(lock, std::chrono::microseconds(timeout_us), ready)) return 0;
        size_t n = 0;
        while (n < max && !in_.empty()) {  // (one lock for the lot: a lock and a wake-up per message bound the loop at ~1e6 Requests/s)
            out->push_back(std::move(in_.front()));
            in_.pop_front();
            ++n;
        }
        return n;
    }
    bool closed() override {
        std::lock_guard<std::mutex> lock(mu_);
        return closed_ && in_.empty();
    }
    bool closing() override {
        std::lock_guard<std::mutex> lock(mu_);
        return closed_;
    }
    void send(channel c, const uint8_t *data, size_t n) override {
        {
            std::lock_guard<std::mutex> lock(mu_out_);
            out_[c].emplace_back(data, data + n);
        }
        cv_out_.notify_all();
    }
    void send_many(channel c, const uint8_t *base, const size_t *off, size_t count) override {
        {
            std::lock_guard<std::mutex> lock(mu_out_);
            for (size_t j = 0; j < count; ++j) out_[c].emplace_back(base + off[j], base + off[j + 1]);
        }
        cv_out_.notify_all();
    }
    void push_request(const uint8_t *data, size_t n) {
        {
            std::lock_guard<std::mutex> lock(mu_);
            in_.emplace_back(data, data + n);
        }
        cv_.notify_all();
    }
    void push_requests(const uint8_t *base, const uint64_t *off, size_t count) {
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (size_t j = 0; j < count; ++j) in_.emplace_back(base + off[j], base + off[j + 1]);
        }
        cv_.notify_all();
    }
    // up to max_msgs replies of channel c, back to back into buf (message j at off[j] .. off[j + 1]); a reply that no
    // longer fits stays queued.  The number taken; 0 after timeout_us without one.
    size_t pop_replies(int c, uint8_t *buf, size_t cap, uint64_t *off, size_t max_msgs, int64_t timeout_us) {
        std::unique_lock<std::mutex> lock(mu_out_);
        if (!cv_out_.wait_for(lock, std::chrono::microseconds(timeout_us), [&] { return !out_[c].empty(); })) return 0;
        size_t n = 0, at = 0;
        off[0] = 0;
        while (n < max_msgs && !out_[c].empty() && at + out_[c].front().size() <= cap) {
            const std::vector<uint8_t> &m = out_[c].front();
            if (!m.empty()) memcpy(buf + at, m.data(), m.size());
            at += m.size();
            off[++n] = at;
            out_[c].pop_front();
        }
        return n;
    }
    bool pop_reply(int c, std::vector<uint8_t> *msg, int64_t timeout_us) {
        std::unique_lock<std::mutex> lock(mu_out_);
        if (!cv_out_.wait_for(lock, std::chrono::microseconds(timeout_us), [&] { return !out_[c].empty(); })) return false;
        *msg = std::move(out_[c].front());
        out_[c].pop_front();
        return true;
    }
    void close() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            closed_ = true;
        }
        cv_.notify_all();
    }

  private:
    std::mutex mu_, mu_out_;  // requests in / replies out
    std::condition_variable cv_, cv_out_;
    std::deque<std::vector<uint8_t>> in_, out_[2];
    bool closed_ = false;
};

// ---- libzmq 4.x, bound at run time.  The ABI used (zmq.h of 4.0 .. 4.3): socket types, option and
// flag numbers are part of the wire / C contract and have not changed since 3.x.
constexpr int ZMQ_SUB_ = 2, ZMQ_PUSH_ = 8, ZMQ_SUBSCRIBE_ = 6, ZMQ_LINGER_ = 17, ZMQ_POLLIN_ = 1;
struct zmq_msg_ {  // zmq_msg_t: 64 opaque bytes, pointer-aligned
    alignas(8) unsigned char _[64];
};
struct zmq_pollitem_ {  // zmq_pollitem_t on POSIX
    void *socket;
    int fd;
    short events, revents;
};
struct zmq_api {
    void *lib = nullptr;
    void *(*ctx_new)() = nullptr;
    int (*ctx_term)(void *) = nullptr;
    void *(*socket)(void *, int) = nullptr;
    int (*close)(void *) = nullptr;
    int (*connect)(void *, const char *) = nullptr;
    int (*setsockopt)(void *, int, const void *, size_t) = nullptr;
    int (*poll)(zmq_pollitem_ *, int, long) = nullptr;
    int (*msg_init)(zmq_msg_ *) = nullptr;
    int (*msg_recv)(zmq_msg_ *, void *, int) = nullptr;
    void *(*msg_data)(zmq_msg_ *) = nullptr;
    int (*msg_close)(zmq_msg_ *) = nullptr;
    int (*send)(void *, const void *, size_t, int) = nullptr;
    int (*zerrno)() = nullptr;
    const char *(*strerror)(int) = nullptr;
    bool ok = false;
};

zmq_api &zmq() {
    static zmq_api api = [] {
        zmq_api a;
        // RSBWT_LIBZMQ names the library outright; otherwise the loader's search path, then two prefixes
        // where distributions that are not on it keep theirs
        const char *env = getenv("RSBWT_LIBZMQ");
        const char *names[] = {env, "libzmq.so.5", "libzmq.so", "/usr/local/lib/libzmq.so.5", "/opt/conda/lib/libzmq.so.5"};
        for (const char *name : names) {
            if (!name || !*name) continue;
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
#define ZMQ_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, sym))
        ZMQ_SYM(ctx_new, "zmq_ctx_new");
        ZMQ_SYM(ctx_term, "zmq_ctx_term");
        ZMQ_SYM(socket, "zmq_socket");
        ZMQ_SYM(close, "zmq_close");
        ZMQ_SYM(connect, "zmq_connect");
        ZMQ_SYM(setsockopt, "zmq_setsockopt");
        ZMQ_SYM(poll, "zmq_poll");
        ZMQ_SYM(msg_init, "zmq_msg_init");
        ZMQ_SYM(msg_recv, "zmq_msg_recv");
        ZMQ_SYM(msg_data, "zmq_msg_data");
        ZMQ_SYM(msg_close, "zmq_msg_close");
        ZMQ_SYM(send, "zmq_send");
        ZMQ_SYM(zerrno, "zmq_errno");
        ZMQ_SYM(strerror, "zmq_strerror");
#undef ZMQ_SYM
        a.ok = a.ctx_new && a.ctx_term && a.socket && a.close && a.connect && a.setsockopt && a.poll && a.msg_init &&
               a.msg_recv && a.msg_data && a.msg_close && a.send && a.zerrno && a.strerror;
        return a;
    }();
    return api;
}

// ZeroMQ: SUB connect(pull) + subscribe-all, PUSH connect(push), PUSH connect(push_count)
// (service.cpp:1493-1502).
class zmq_transport : public transport {
  public:
    zmq_transport(const std::string &pull, const std::string &push, const std::string &push_count) {
        zmq_api &z = zmq();
        ctx_ = z.ctx_new();
        if (!ctx_) return;
        sub_ = z.socket(ctx_, ZMQ_SUB_);
        out_[PUSH] = z.socket(ctx_, ZMQ_PUSH_);
        out_[PUSH_COUNT] = z.socket(ctx_, ZMQ_PUSH_);
        if (!sub_ || !out_[PUSH] || !out_[PUSH_COUNT]) return;
        // replies not yet delivered when the service stops are dropped after a second instead of holding
        // zmq_ctx_term for ever (the reference's process simply exits)
        const int linger_ms = 1000;
        for (void *so : {sub_, out_[PUSH], out_[PUSH_COUNT]}) (void)z.setsockopt(so, ZMQ_LINGER_, &linger_ms, sizeof linger_ms);
        ok_ = z.connect(sub_, pull.c_str()) == 0 && z.setsockopt(sub_, ZMQ_SUBSCRIBE_, "", 0) == 0 &&
              z.connect(out_[PUSH], push.c_str()) == 0 && z.connect(out_[PUSH_COUNT], push_count.c_str()) == 0;
        if (!ok_) err_ = z.strerror(z.zerrno());
    }
    ~zmq_transport() override {
        zmq_api &z = zmq();
        if (sub_) z.close(sub_);
        if (out_[0]) z.close(out_[0]);
        if (out_[1]) z.close(out_[1]);
        if (ctx_) z.ctx_term(ctx_);
    }
    bool ok() const { return ok_; }
    const std::string &error() const { return err_; }
    bool recv(std::vector<uint8_t> *msg, int64_t timeout_us) override {
        zmq_api &z = zmq();
        zmq_pollitem_ it = {sub_, 0, (short)ZMQ_POLLIN_, 0};
        // a stop request must be seen: never sleep longer than 50 ms in one poll
        const long ms = timeout_us < 0 ? 50 : (long)std::min<int64_t>((timeout_us + 999) / 1000, 50);
        if (z.poll(&it, 1, ms) <= 0) return false;  // nothing yet, or EINTR: the loop asks again
        zmq_msg_ m;
        z.msg_init(&m);
        const int n = z.msg_recv(&m, sub_, 0);
        if (n >= 0) msg->assign((const uint8_t *)z.msg_data(&m), (const uint8_t *)z.msg_data(&m) + n);
        z.msg_close(&m);
        return n >= 0;
    }
    bool closed() override { return stop_.load(); }
    bool closing() override { return stop_.load(); }
    void send(channel c, const uint8_t *data, size_t n) override {
        zmq_api &z = zmq();
        std::lock_guard<std::mutex> lock(mu_);  // the reference guards sender->send the same way (service.cpp:311-313)
        // the front-end waits for every reply and has no timeout (server.cpp:403,469): an interrupted send is
        // repeated; any other failure is counted and logged, never silently dropped
        for (;;) {
            if (z.send(out_[c], data, n, 0) >= 0) return;
            const int e = z.zerrno();
            if (e == EINTR) continue;
            const uint64_t k = send_failures_.fetch_add(1) + 1;
            if (k <= 10 || (k & (k - 1)) == 0)
                fprintf(stderr, "rsbwt service: zmq_send of a %zu-byte reply failed (%s); %llu replies lost so far\n", n,
                        z.strerror(e), (unsigned long long)k);
            return;
        }
    }
    uint64_t send_failures() const { return send_failures_.load(); }
    void stop() { stop_.store(true); }

  private:
    void *ctx_ = nullptr, *sub_ = nullptr, *out_[2] = {nullptr, nullptr};
    bool ok_ = false;
    std::string err_;
    std::atomic<bool> stop_{false};
    std::atomic<uint64_t> send_failures_{0};
    std::mutex mu_;
};

}  // namespace

struct rsbwt_transport {
    transport *t = nullptr;
    inproc_transport *inproc = nullptr;  // same object when in-process
    zmq_transport *zmq = nullptr;        // same object when ZeroMQ
};

// The loop as a pipeline (the reference answers from a pool of 8 threads on one shared index:
// src/service/service.cpp:88,1505,1532-1561):
//   receiver (the thread that called run): gathers Requests into micro-batch windows, numbers them;
//   `workers` threads: a whole window each -- decode, ONE batched search per query length over the shard set
//       (the set's entry points are re-entrant: concurrent windows run on contexts / streams of their own), Reply
//       bytes;
//   sender: the windows' Replies in window order -- so replies leave in arrival order, as with one thread.
// With one thread doing all of it (rounds 2-3) nobody received while the GPU answered a window and nobody used the GPU
// while Replies were encoded: 1.0e6 Requests/s against a kernel that resolves 4e8 queries/s.
struct window_job {
    uint64_t seq = 0;
    std::vector<std::vector<uint8_t>> msgs;
    std::vector<service_request> rq;
    std::vector<char> parsed, handled;
    reply_arena rep;
    bool done = false;
};

struct rsbwt_service {
    rsbwt_set_t *set = nullptr;
    transport *tr = nullptr;
    int64_t window_us = 200;
    size_t max_batch = 4096;
    bool per_partition = true;
    int workers = 8;  // service.cpp:88
    rsbwt_service_other_fn other = nullptr;
    void *other_arg = nullptr;
    // ExactMatch requests that ask for Reads (QueryTask, service.cpp:1556-1561 -> find_reads :714-797): answered here
    // (BWT only); off = left to `other`, as until round 5
    bool serve_reads = true;
    reads_config reads_cfg;
    std::atomic<uint64_t> read_requests{0};
    std::atomic<bool> stop{false};
    std::thread worker;
    // statistics
    std::atomic<uint64_t> requests{0}, count_requests{0}, batches{0}, replies{0}, malformed{0}, max_batch_seen{0};
    std::mutex err_mu;
    int last_rc = RSBWT_OK;
    std::string last_err;

    // the pipeline's queues: windows waiting for a worker, windows in flight (in window order) for the sender
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_room;
    std::deque<window_job *> todo;
    std::deque<window_job *> inflight;
    bool no_more = false;

    // receiver: one window.  false = the transport closed with nothing pending
    // A window fills WHILE WINDOWS ARE IN FLIGHT: with every earlier window answered and sent, waiting out the timer
    // would only add its length to the latency of the requests that are here (a lone request: 335 us through the
    // loop, 200 of them this wait) -- so an idle pipeline takes what has arrived and goes; under load the windows
    // in flight are what the next one batches behind, up to window_us / max_batch as before.
    bool gather(std::vector<std::vector<uint8_t>> *msgs) {
        if (tr->recv_many(msgs, max_batch, 50000) == 0) return !tr->closed() && !stop.load();
        const auto t0 = std::chrono::steady_clock::now();
        while (msgs->size() < max_batch) {
            const int64_t spent = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (spent >= window_us) break;
            bool idle;
            {
                std::lock_guard<std::mutex> lock(mu);
                idle = inflight.empty();
            }
            if (tr->recv_many(msgs, max_batch - msgs->size(), idle ? 0 : window_us - spent) == 0) break;
        }
        return true;
    }

    // worker: decode, search, Reply bytes
    void answer(window_job &job) {
        const size_t n = job.msgs.size();
        job.rq.resize(n);
        job.parsed.assign(n, 0);
        bool any_reads = false;
        for (size_t i = 0; i < n; ++i) {
            job.parsed[i] = service_decode(job.msgs[i].data(), job.msgs[i].size(), &job.rq[i]) ? 1 : 0;
            if (!job.parsed[i]) { job.rq[i].t = 0; malformed++; }
            else if (serve_reads && service_is_reads_request(job.rq[i])) any_reads = true;
        }
        const size_t rows = per_partition ? rsbwt_set_size(set) : 1;
        int rc = service_count_batch(set, job.rq, per_partition, &job.rep, &job.handled);
        if (rc != RSBWT_OK) {
            // The front-end has no timeout (server.cpp:403,469,480): a request must not go unanswered.
            // A failed batch is answered with zero counts and the error kept for the operator.
            note_failure(rc, n, "count");
            job.handled.assign(n, 0);
            job.rep.bytes.clear();
            job.rep.off.assign(1, 0);
            job.rep.first.assign(n + 1, 0);
            for (size_t i = 0; i < n; ++i) {
                job.rep.first[i] = job.rep.off.size() - 1;
                const bool is_count = job.rq[i].t == 1 || (job.rq[i].t == 2 && job.rq[i].rt == 1);
                if (!is_count) continue;
                job.handled[i] = 1;
                for (size_t r = 0; r < rows; ++r)
                    for (int strand = 0; strand < 2; ++strand) {
                        const size_t len = rsbwt_proto_encode_count_reply(nullptr, 0, job.rq[i].t, job.rq[i].q.data(), job.rq[i].q.size(), strand, 0);
                        const size_t at = job.rep.bytes.size();
                        job.rep.bytes.resize(at + len);
                        rsbwt_proto_encode_count_reply(job.rep.bytes.data() + at, len, job.rq[i].t, job.rq[i].q.data(), job.rq[i].q.size(), strand, 0);
                        job.rep.off.push_back(job.rep.bytes.size());
                    }
            }
            job.rep.first[n] = job.rep.off.size() - 1;
        }
        if (!any_reads) return;
        // the window's ExactMatch-Reads requests (find_reads, service.cpp:714-797): their Replies are woven into the
        // window's in arrival order.  A failed batch is answered with empty read lists, for the same reason as above.
        reply_arena rr;
        std::vector<char> handled_r;
        rc = service_reads_batch(set, job.rq, per_partition, reads_cfg, &rr, &handled_r);
        if (rc != RSBWT_OK) {
            note_failure(rc, n, "read");
            service_reads_empty(job.rq, rows, &rr, &handled_r);
        }
        reply_arena all;
        all.off.assign(1, 0);
        all.first.assign(n + 1, 0);
        all.bytes.reserve(job.rep.bytes.size() + rr.bytes.size());
        for (size_t i = 0; i < n; ++i) {
            all.first[i] = all.off.size() - 1;
            const reply_arena *src = job.handled[i] ? &job.rep : handled_r[i] ? &rr : nullptr;
            if (!src) continue;
            for (size_t j = src->first[i]; j < src->first[i + 1]; ++j) {
                all.bytes.insert(all.bytes.end(), src->bytes.begin() + src->off[j], src->bytes.begin() + src->off[j + 1]);
                all.off.push_back(all.bytes.size());
            }
            if (handled_r[i]) { job.handled[i] = 2; read_requests++; }
        }
        all.first[n] = all.off.size() - 1;
        job.rep = std::move(all);
    }

    void note_failure(int rc, size_t n, const char *what) {
        std::lock_guard<std::mutex> lock(err_mu);
        last_rc = rc;
        last_err = rsbwt_last_error();
        fprintf(stderr, "rsbwt service: the %s requests of a window of %zu failed: %s\n", what, n, last_err.c_str());
    }

    // sender: in arrival order; consecutive messages for the same socket go out in one call.
    // CountReads answers on push_count, ExactMatch-Count on push (service.cpp:1549-1554,1567-1570)
    void emit(window_job &job) {
        const size_t n = job.msgs.size();
        const reply_arena &rep = job.rep;
        size_t run0 = 0, run1 = 0;
        transport::channel run_ch = transport::PUSH_COUNT;
        auto flush = [&] {
            if (run1 > run0) tr->send_many(run_ch, rep.bytes.data(), rep.off.data() + run0, run1 - run0);
            replies += run1 - run0;
            run0 = run1;
        };
        for (size_t i = 0; i < n; ++i) {
            if (job.handled[i]) {
                if (job.handled[i] == 1) count_requests++;
                const transport::channel ch = job.rq[i].t == 1 ? transport::PUSH_COUNT : transport::PUSH;
                if (ch != run_ch) { flush(); run_ch = ch; }
                run0 = run1 > run0 ? run0 : rep.first[i];
                run1 = rep.first[i + 1];
            } else if (job.parsed[i] && other) {
                flush();
                other(other_arg, job.msgs[i].data(), job.msgs[i].size());  // KmerMatch, SiteMatch, ExactMatch with All / Samples: the caller's
            }
        }
        flush();
        requests += n;
        batches++;
        uint64_t seen = max_batch_seen.load();
        while (n > seen && !max_batch_seen.compare_exchange_weak(seen, n)) {}
    }

    void worker_loop() {
        for (;;) {
            window_job *job = nullptr;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv_work.wait(lock, [&] { return !todo.empty() || no_more; });
                if (todo.empty()) return;
                job = todo.front();
                todo.pop_front();
            }
            // (an exception must not leave this thread -- std::terminate -- nor the window unanswered: the front-end
            // has no timeout.  What answer() could not finish is answered with zero counts / empty lists.)
            try {
                answer(*job);
            } catch (...) {
                fail_window(*job, "answering");
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                job->done = true;
            }
            cv_done.notify_all();
        }
    }

    // A window whose answer threw (std::bad_alloc in the reply arena, ...): every count / read request of it gets its
    // zero / empty Replies (built from the decoded requests alone; if even that fails the window is dropped and the
    // loss logged), the error is kept for rsbwt_service_run's return code.
    void fail_window(window_job &job, const char *doing) noexcept {
        try {
            {
                std::lock_guard<std::mutex> lock(err_mu);
                last_rc = RSBWT_ENOMEM;
                last_err = std::string("exception while ") + doing + " a window";
                fprintf(stderr, "rsbwt service: %s (%zu requests): answered with empty results\n", last_err.c_str(), job.msgs.size());
            }
            const size_t n = job.msgs.size(), rows = per_partition ? rsbwt_set_size(set) : 1;
            if (job.rq.size() != n) {
                job.rq.assign(n, service_request());
                job.parsed.assign(n, 0);
                for (size_t i = 0; i < n; ++i) job.parsed[i] = service_decode(job.msgs[i].data(), job.msgs[i].size(), &job.rq[i]) ? 1 : 0;
            }
            reply_arena all, rr;
            std::vector<char> handled_r;
            service_reads_empty(job.rq, rows, &rr, &handled_r);
            all.off.assign(1, 0);
            all.first.assign(n + 1, 0);
            job.handled.assign(n, 0);
            for (size_t i = 0; i < n; ++i) {
                all.first[i] = all.off.size() - 1;
                const bool is_count = job.parsed[i] && (job.rq[i].t == 1 || (job.rq[i].t == 2 && job.rq[i].rt == 1));
                if (is_count) {
                    job.handled[i] = 1;
                    for (size_t r = 0; r < rows; ++r)
                        for (int strand = 0; strand < 2; ++strand) {
                            const size_t len = rsbwt_proto_encode_count_reply(nullptr, 0, job.rq[i].t, job.rq[i].q.data(), job.rq[i].q.size(), strand, 0);
                            const size_t at = all.bytes.size();
                            all.bytes.resize(at + len);
                            rsbwt_proto_encode_count_reply(all.bytes.data() + at, len, job.rq[i].t, job.rq[i].q.data(), job.rq[i].q.size(), strand, 0);
                            all.off.push_back(all.bytes.size());
                        }
                } else if (serve_reads && job.parsed[i] && handled_r[i]) {
                    job.handled[i] = 2;
                    for (size_t j = rr.first[i]; j < rr.first[i + 1]; ++j) {
                        all.bytes.insert(all.bytes.end(), rr.bytes.begin() + rr.off[j], rr.bytes.begin() + rr.off[j + 1]);
                        all.off.push_back(all.bytes.size());
                    }
                }
            }
            all.first[n] = all.off.size() - 1;
            job.rep = std::move(all);
        } catch (...) {
            fprintf(stderr, "rsbwt service: a window of %zu requests could not be answered at all\n", job.msgs.size());
            job.handled.assign(job.msgs.size(), 0);
            job.parsed.assign(job.msgs.size(), 0);
            job.rep.bytes.clear();
            job.rep.off.assign(1, 0);
            job.rep.first.assign(job.msgs.size() + 1, 0);
        }
    }

    void sender_loop() {
        for (;;) {
            window_job *job = nullptr;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv_done.wait(lock, [&] { return (!inflight.empty() && inflight.front()->done) || (no_more && inflight.empty()); });
                if (inflight.empty()) return;
                job = inflight.front();
                inflight.pop_front();
            }
            cv_room.notify_all();
            try {
                emit(*job);  // (calls the embedder's `other` handler: on THIS thread, the sender's)
            } catch (...) {
                std::lock_guard<std::mutex> lock(err_mu);
                last_rc = RSBWT_ESYS;
                last_err = "exception while sending a window's replies (the `other` handler threw?)";
                fprintf(stderr, "rsbwt service: %s\n", last_err.c_str());
            }
            delete job;
        }
    }

    void run() {
        const int nw = workers < 1 ? 1 : workers > 64 ? 64 : workers;
        {
            std::lock_guard<std::mutex> lock(mu);
            no_more = false;
        }
        std::vector<std::thread> pool;
        std::thread sender;
        // (a thread that cannot be started -- std::system_error -- must not unwind past the joinable ones already
        // running: they are told there is no more work and joined, and the error is the run's)
        try {
            for (int i = 0; i < nw; ++i) pool.emplace_back([this] { worker_loop(); });
            sender = std::thread([this] { sender_loop(); });
        } catch (...) {
            {
                std::lock_guard<std::mutex> lock(mu);
                no_more = true;
            }
            cv_work.notify_all();
            cv_done.notify_all();
            for (std::thread &t : pool) t.join();
            if (sender.joinable()) sender.join();
            std::lock_guard<std::mutex> lock(err_mu);
            last_rc = RSBWT_ESYS;
            last_err = "cannot start the service's threads";
            return;
        }
        uint64_t seq = 0;
        while (!stop.load()) {
            window_job *job = new (std::nothrow) window_job();
            if (!job) break;
            if (!gather(&job->msgs)) { delete job; break; }
            if (job->msgs.empty()) { delete job; continue; }  // nothing arrived in time: ask again
            job->seq = seq++;
            {
                std::unique_lock<std::mutex> lock(mu);
                // at most two windows per worker in flight: a sender that cannot keep up holds the receiver back
                cv_room.wait(lock, [&] { return inflight.size() < 2 * (size_t)nw; });
                todo.push_back(job);
                inflight.push_back(job);
            }
            cv_work.notify_one();
        }
        {
            std::lock_guard<std::mutex> lock(mu);
            no_more = true;
        }
        cv_work.notify_all();
        cv_done.notify_all();
        for (std::thread &t : pool) t.join();
        cv_done.notify_all();
        sender.join();
    }
};

extern "C" {

// ---- configuration ---------------------------------------------------------------------------------
static int rsbwt_service_config_load_body(const char *path, rsbwt_service_config_t **out) {
    if (!path || !out) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(RSBWT_EIO, "cannot open %s", path);
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    rsbwt_service_config *cfg = new (std::nothrow) rsbwt_service_config();
    if (!cfg) return fail(RSBWT_ENOMEM, "host allocation failed");
    std::string err;
    const int rc = parse_config(text, cfg, &err);
    if (rc != RSBWT_OK) {
        delete cfg;
        return fail(rc, "%s: %s", path, err.c_str());
    }
    // the settings the reference requires (cfg.lookup throws SettingNotFoundException: service.cpp:1425-1442)
    for (const char *k : {"prefix", "suffix", "hashfile", "pull", "push", "push_count", "rocksdb_path", "rocksdb_ext"})
        if (!cfg->scalars.count(k)) {
            delete cfg;
            return fail(RSBWT_EFORMAT, "%s: setting '%s' not found", path, k);
        }
    if (!cfg->arrays.count("rocksdb")) {
        delete cfg;
        return fail(RSBWT_EFORMAT, "%s: setting 'rocksdb' not found", path);
    }
    *out = cfg;
    return RSBWT_OK;
}
int rsbwt_service_config_load(const char *path, rsbwt_service_config_t **out) {
    return guarded("rsbwt_service_config_load", [&]() -> int { return rsbwt_service_config_load_body(path, out); });
}


void rsbwt_service_config_free(rsbwt_service_config_t *cfg) { delete cfg; }

const char *rsbwt_service_config_get(const rsbwt_service_config_t *cfg, const char *key) {
    if (!cfg || !key) return nullptr;
    auto it = cfg->scalars.find(key);
    return it == cfg->scalars.end() ? nullptr : it->second.c_str();
}

size_t rsbwt_service_config_array_len(const rsbwt_service_config_t *cfg, const char *key) {
    if (!cfg || !key) return 0;
    auto it = cfg->arrays.find(key);
    return it == cfg->arrays.end() ? 0 : it->second.size();
}

const char *rsbwt_service_config_array_item(const rsbwt_service_config_t *cfg, const char *key, size_t i) {
    if (!cfg || !key) return nullptr;
    auto it = cfg->arrays.find(key);
    return (it == cfg->arrays.end() || i >= it->second.size()) ? nullptr : it->second[i].c_str();
}

// ---- transports ------------------------------------------------------------------------------------
static int rsbwt_transport_inproc_body(rsbwt_transport_t **out) {
    if (!out) return fail(RSBWT_EINVAL, "null argument");
    rsbwt_transport *t = new (std::nothrow) rsbwt_transport();
    inproc_transport *ip = new (std::nothrow) inproc_transport();
    if (!t || !ip) { delete t; delete ip; return fail(RSBWT_ENOMEM, "host allocation failed"); }
    t->t = ip;
    t->inproc = ip;
    *out = t;
    return RSBWT_OK;
}
int rsbwt_transport_inproc(rsbwt_transport_t **out) {
    return guarded("rsbwt_transport_inproc", [&]() -> int { return rsbwt_transport_inproc_body(out); });
}


int rsbwt_transport_zmq(const char *pull, const char *push, const char *push_count, rsbwt_transport_t **out) {
    if (!pull || !push || !push_count || !out) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    return guarded("rsbwt_transport_zmq", [&]() -> int {
        if (!zmq().ok)
            return fail(RSBWT_ENODEV, "libzmq could not be bound at run time (libzmq.so.5 not on the loader's path; RSBWT_LIBZMQ names one)");
        rsbwt_transport *t = new (std::nothrow) rsbwt_transport();
        zmq_transport *z = new (std::nothrow) zmq_transport(pull, push, push_count);
        if (!t || !z || !z->ok()) {
            const std::string why = z ? z->error() : std::string("host allocation failed");
            delete t;
            delete z;
            return fail(RSBWT_EIO, "cannot connect the ZeroMQ sockets (%s, %s, %s): %s", pull, push, push_count, why.c_str());
        }
        t->t = z;
        t->zmq = z;
        *out = t;
        return RSBWT_OK;
    });
}

int rsbwt_zmq_available(void) { return zmq().ok ? 1 : 0; }

void rsbwt_transport_free(rsbwt_transport_t *t) {
    if (!t) return;
    delete t->t;
    delete t;
}

static int rsbwt_transport_push_request_body(rsbwt_transport_t *t, const uint8_t *msg, size_t n) {
    if (!t || !t->inproc || (!msg && n)) return fail(RSBWT_EINVAL, "not an in-process transport");
    t->inproc->push_request(msg, n);
    return RSBWT_OK;
}
int rsbwt_transport_push_request(rsbwt_transport_t *t, const uint8_t *msg, size_t n) {
    return guarded("rsbwt_transport_push_request", [&]() -> int { return rsbwt_transport_push_request_body(t, msg, n); });
}


int rsbwt_transport_push_requests(rsbwt_transport_t *t, const uint8_t *base, const uint64_t *off, size_t count) {
    return guarded("rsbwt_transport_push_requests", [&]() -> int {
        if (!t || !t->inproc || ((!base || !off) && count)) return fail(RSBWT_EINVAL, "not an in-process transport");
        for (size_t j = 0; j < count; ++j)
            if (off[j + 1] < off[j]) return fail(RSBWT_EINVAL, "request offsets must ascend");
        t->inproc->push_requests(base, off, count);
        return RSBWT_OK;
    });
}

int rsbwt_transport_pop_replies(rsbwt_transport_t *t, int channel, uint8_t *buf, size_t cap, uint64_t *off, size_t max_msgs,
                                size_t *count, int64_t timeout_us) {
    return guarded("rsbwt_transport_pop_replies", [&]() -> int {
        if (!t || !t->inproc || !buf || !off || !count || channel < 0 || channel > 1) return fail(RSBWT_EINVAL, "not an in-process transport");
        *count = t->inproc->pop_replies(channel, buf, cap, off, max_msgs, timeout_us);
        return RSBWT_OK;
    });
}

static int rsbwt_transport_pop_reply_body(rsbwt_transport_t *t, int channel, uint8_t *buf, size_t cap, size_t *n, int64_t timeout_us) {
    if (!t || !t->inproc || !n || channel < 0 || channel > 1) return fail(RSBWT_EINVAL, "not an in-process transport");
    std::vector<uint8_t> m;
    if (!t->inproc->pop_reply(channel, &m, timeout_us)) return fail(RSBWT_EIO, "no reply within %lld us", (long long)timeout_us);
    *n = m.size();
    if (m.size() > cap) return fail(RSBWT_ERANGE, "%zu reply bytes, room for %zu", m.size(), cap);
    if (!m.empty()) memcpy(buf, m.data(), m.size());
    return RSBWT_OK;
}
int rsbwt_transport_pop_reply(rsbwt_transport_t *t, int channel, uint8_t *buf, size_t cap, size_t *n, int64_t timeout_us) {
    return guarded("rsbwt_transport_pop_reply", [&]() -> int { return rsbwt_transport_pop_reply_body(t, channel, buf, cap, n, timeout_us); });
}


void rsbwt_transport_close(rsbwt_transport_t *t) {
    if (t && t->inproc) t->inproc->close();
    else if (t && t->zmq) t->zmq->stop();
}

// ---- the service ------------------------------------------------------------------------------------
static int rsbwt_service_create_body(rsbwt_set_t *set, rsbwt_transport_t *t, int64_t window_us, size_t max_batch, int per_partition,
                         rsbwt_service_t **out) {
    if (!set || !t || !out) return fail(RSBWT_EINVAL, "null argument");
    rsbwt_service *s = new (std::nothrow) rsbwt_service();
    if (!s) return fail(RSBWT_ENOMEM, "host allocation failed");
    s->set = set;
    s->tr = t->t;
    s->window_us = window_us < 0 ? 0 : window_us;
    s->max_batch = max_batch ? max_batch : 1;
    s->per_partition = per_partition != 0;
    *out = s;
    return RSBWT_OK;
}
int rsbwt_service_create(rsbwt_set_t *set, rsbwt_transport_t *t, int64_t window_us, size_t max_batch, int per_partition,
                         rsbwt_service_t **out) {
    return guarded("rsbwt_service_create", [&]() -> int { return rsbwt_service_create_body(set, t, window_us, max_batch, per_partition, out); });
}


void rsbwt_service_set_workers(rsbwt_service_t *s, int workers) {
    if (s && workers >= 1) s->workers = workers > 64 ? 64 : workers;
}

void rsbwt_service_set_reads(rsbwt_service_t *s, int enable, uint32_t min_read_length, uint32_t max_read_length) {
    if (!s) return;
    s->serve_reads = enable != 0;
    if (min_read_length) s->reads_cfg.min_read_length = min_read_length;
    if (max_read_length) s->reads_cfg.max_read_length = max_read_length;
}

int rsbwt_service_set_suffixes(rsbwt_service_t *s, const char *const *suffix, size_t n) {
    return guarded("rsbwt_service_set_suffixes", [&]() -> int {
        if (!s || (!suffix && n)) return fail(RSBWT_EINVAL, "null argument");
        if (n != 0 && n != rsbwt_set_size(s->set)) return fail(RSBWT_EINVAL, "%zu suffixes for a set of %zu shards", n, rsbwt_set_size(s->set));
        s->reads_cfg.suffix.clear();
        for (size_t i = 0; i < n; ++i) s->reads_cfg.suffix.emplace_back(suffix[i] ? suffix[i] : "");
        return RSBWT_OK;
    });
}

void rsbwt_service_set_other_handler(rsbwt_service_t *s, rsbwt_service_other_fn fn, void *arg) {
    if (!s) return;
    s->other = fn;
    s->other_arg = arg;
}

static int rsbwt_service_run_body(rsbwt_service_t *s) {
    if (!s) return fail(RSBWT_EINVAL, "null service");
    s->run();
    return s->last_rc == RSBWT_OK ? RSBWT_OK : fail(s->last_rc, "%s", s->last_err.c_str());
}
int rsbwt_service_run(rsbwt_service_t *s) {
    return guarded("rsbwt_service_run", [&]() -> int { return rsbwt_service_run_body(s); });
}


static int rsbwt_service_start_body(rsbwt_service_t *s) {
    if (!s || s->worker.joinable()) return fail(RSBWT_EINVAL, "null or already running service");
    s->stop.store(false);
    s->worker = std::thread([s] { s->run(); });
    return RSBWT_OK;
}
int rsbwt_service_start(rsbwt_service_t *s) {
    return guarded("rsbwt_service_start", [&]() -> int { return rsbwt_service_start_body(s); });
}


static int rsbwt_service_stop_body(rsbwt_service_t *s) {
    if (!s) return fail(RSBWT_EINVAL, "null service");
    // a transport that is closing drains: the loop ends by itself once what was received is answered
    if (!s->tr->closing()) s->stop.store(true);
    if (s->worker.joinable()) s->worker.join();
    return s->last_rc == RSBWT_OK ? RSBWT_OK : fail(s->last_rc, "%s", s->last_err.c_str());
}
int rsbwt_service_stop(rsbwt_service_t *s) {
    return guarded("rsbwt_service_stop", [&]() -> int { return rsbwt_service_stop_body(s); });
}


void rsbwt_service_free(rsbwt_service_t *s) {
    if (!s) return;
    s->stop.store(true);
    if (s->worker.joinable()) s->worker.join();
    delete s;
}

void rsbwt_service_stats(const rsbwt_service_t *s, uint64_t *stats6) {
    if (!s || !stats6) return;
    stats6[0] = s->requests.load();
    stats6[1] = s->count_requests.load();
    stats6[2] = s->batches.load();
    stats6[3] = s->replies.load();
    stats6[4] = s->malformed.load();
    stats6[5] = s->max_batch_seen.load();
}

uint64_t rsbwt_service_read_requests(const rsbwt_service_t *s) { return s ? s->read_requests.load() : 0; }

}  // extern "C"

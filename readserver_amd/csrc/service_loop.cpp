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
// in-process queue pair for tests and embedding, ZeroMQ where libzmq exists (-DRSBWT_WITH_ZMQ).
#include <stdint.h>
#include <stdio.h>
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

#ifdef RSBWT_WITH_ZMQ
#include <zmq.h>
#endif

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

#ifdef RSBWT_WITH_ZMQ
// ZeroMQ: SUB connect(pull) + subscribe-all, PUSH connect(push), PUSH connect(push_count)
// (service.cpp:1493-1502).  Compiled only where libzmq's zmq.h exists.
class zmq_transport : public transport {
  public:
    zmq_transport(const std::string &pull, const std::string &push, const std::string &push_count) {
        ctx_ = zmq_ctx_new();
        sub_ = zmq_socket(ctx_, ZMQ_SUB);
        ok_ = zmq_connect(sub_, pull.c_str()) == 0 && zmq_setsockopt(sub_, ZMQ_SUBSCRIBE, "", 0) == 0;
        out_[PUSH] = zmq_socket(ctx_, ZMQ_PUSH);
        out_[PUSH_COUNT] = zmq_socket(ctx_, ZMQ_PUSH);
        ok_ = ok_ && zmq_connect(out_[PUSH], push.c_str()) == 0 && zmq_connect(out_[PUSH_COUNT], push_count.c_str()) == 0;
    }
    ~zmq_transport() override {
        zmq_close(sub_);
        zmq_close(out_[0]);
        zmq_close(out_[1]);
        zmq_ctx_term(ctx_);
    }
    bool ok() const { return ok_; }
    bool recv(std::vector<uint8_t> *msg, int64_t timeout_us) override {
        zmq_pollitem_t it = {sub_, 0, ZMQ_POLLIN, 0};
        const long ms = timeout_us < 0 ? -1 : (long)((timeout_us + 999) / 1000);
        if (zmq_poll(&it, 1, ms) <= 0) return false;
        zmq_msg_t m;
        zmq_msg_init(&m);
        const int n = zmq_msg_recv(&m, sub_, 0);
        if (n >= 0) msg->assign((const uint8_t *)zmq_msg_data(&m), (const uint8_t *)zmq_msg_data(&m) + n);
        zmq_msg_close(&m);
        return n >= 0;
    }
    bool closed() override { return stop_.load(); }
    bool closing() override { return stop_.load(); }
    void send(channel c, const uint8_t *data, size_t n) override {
        std::lock_guard<std::mutex> lock(mu_);  // the reference guards sender->send the same way (service.cpp:311-313)
        zmq_send(out_[c], data, n, 0);
    }
    void stop() { stop_.store(true); }

  private:
    void *ctx_ = nullptr, *sub_ = nullptr, *out_[2] = {nullptr, nullptr};
    bool ok_ = false;
    std::atomic<bool> stop_{false};
    std::mutex mu_;
};
#endif

}  // namespace

struct rsbwt_transport {
    transport *t = nullptr;
    inproc_transport *inproc = nullptr;  // same object when in-process
};

struct rsbwt_service {
    rsbwt_set_t *set = nullptr;
    transport *tr = nullptr;
    int64_t window_us = 200;
    size_t max_batch = 4096;
    bool per_partition = true;
    rsbwt_service_other_fn other = nullptr;
    void *other_arg = nullptr;
    std::atomic<bool> stop{false};
    std::thread worker;
    // statistics
    std::atomic<uint64_t> requests{0}, count_requests{0}, batches{0}, replies{0}, malformed{0}, max_batch_seen{0};
    int last_rc = RSBWT_OK;
    std::string last_err;

    // one window: gather, answer, send.  false = the transport closed with nothing pending
    bool window() {
        std::vector<std::vector<uint8_t>> msgs;
        std::vector<uint8_t> m;
        if (!tr->recv(&m, 50000)) return !tr->closed() && !stop.load();
        msgs.push_back(std::move(m));
        const auto t0 = std::chrono::steady_clock::now();
        while (msgs.size() < max_batch) {
            const int64_t spent = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (spent >= window_us) break;
            if (!tr->recv(&m, window_us - spent)) break;
            msgs.push_back(std::move(m));
        }
        const size_t n = msgs.size();
        std::vector<service_request> rq(n);
        std::vector<char> parsed(n, 0);
        for (size_t i = 0; i < n; ++i) {
            parsed[i] = service_decode(msgs[i].data(), msgs[i].size(), &rq[i]) ? 1 : 0;
            if (!parsed[i]) { rq[i].t = 0; malformed++; }
        }
        reply_arena rep;
        std::vector<char> handled;
        const int rc = service_count_batch(set, rq, per_partition, &rep, &handled);
        if (rc != RSBWT_OK) {
            // The front-end has no timeout (server.cpp:403,469,480): a request must not go unanswered.
            // A failed batch is answered with zero counts and the error kept for the operator.
            last_rc = rc;
            last_err = rsbwt_last_error();
            fprintf(stderr, "rsbwt service: batch of %zu requests failed: %s\n", n, last_err.c_str());
            const size_t rows = per_partition ? rsbwt_set_size(set) : 1;
            for (size_t i = 0; i < n; ++i) {
                const bool is_count = rq[i].t == 1 || (rq[i].t == 2 && rq[i].rt == 1);
                if (!is_count) continue;
                for (size_t r = 0; r < rows; ++r)
                    for (int strand = 0; strand < 2; ++strand) {
                        uint8_t buf[1024];
                        std::vector<uint8_t> big;
                        size_t len = rsbwt_proto_encode_count_reply(buf, sizeof buf, rq[i].t, rq[i].q.data(), rq[i].q.size(), strand, 0);
                        const uint8_t *p = buf;
                        if (len > sizeof buf) {
                            big.resize(len);
                            rsbwt_proto_encode_count_reply(big.data(), len, rq[i].t, rq[i].q.data(), rq[i].q.size(), strand, 0);
                            p = big.data();
                        }
                        tr->send(rq[i].t == 1 ? transport::PUSH_COUNT : transport::PUSH, p, len);
                        replies++;
                    }
            }
        } else {
            // in arrival order; consecutive messages for the same socket go out in one call.
            // CountReads answers on push_count, ExactMatch-Count on push (service.cpp:1549-1554,1567-1570)
            size_t run0 = 0, run1 = 0;
            transport::channel run_ch = transport::PUSH_COUNT;
            auto flush = [&] {
                if (run1 > run0) tr->send_many(run_ch, rep.bytes.data(), rep.off.data() + run0, run1 - run0);
                replies += run1 - run0;
                run0 = run1;
            };
            for (size_t i = 0; i < n; ++i) {
                if (handled[i]) {
                    count_requests++;
                    const transport::channel ch = rq[i].t == 1 ? transport::PUSH_COUNT : transport::PUSH;
                    if (ch != run_ch) { flush(); run_ch = ch; }
                    run0 = run1 > run0 ? run0 : rep.first[i];
                    run1 = rep.first[i + 1];
                } else if (parsed[i] && other) {
                    flush();
                    other(other_arg, msgs[i].data(), msgs[i].size());  // KmerMatch, SiteMatch, ExactMatch with reads: the caller's
                }
            }
            flush();
        }
        requests += n;
        batches++;
        uint64_t seen = max_batch_seen.load();
        while (n > seen && !max_batch_seen.compare_exchange_weak(seen, n)) {}
        return true;
    }

    void run() {
        while (!stop.load() && window()) {}
    }
};

extern "C" {

// ---- configuration ---------------------------------------------------------------------------------
int rsbwt_service_config_load(const char *path, rsbwt_service_config_t **out) {
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
int rsbwt_transport_inproc(rsbwt_transport_t **out) {
    if (!out) return fail(RSBWT_EINVAL, "null argument");
    rsbwt_transport *t = new (std::nothrow) rsbwt_transport();
    inproc_transport *ip = new (std::nothrow) inproc_transport();
    if (!t || !ip) { delete t; delete ip; return fail(RSBWT_ENOMEM, "host allocation failed"); }
    t->t = ip;
    t->inproc = ip;
    *out = t;
    return RSBWT_OK;
}

int rsbwt_transport_zmq(const char *pull, const char *push, const char *push_count, rsbwt_transport_t **out) {
    if (!pull || !push || !push_count || !out) return fail(RSBWT_EINVAL, "null argument");
#ifdef RSBWT_WITH_ZMQ
    rsbwt_transport *t = new (std::nothrow) rsbwt_transport();
    zmq_transport *z = new (std::nothrow) zmq_transport(pull, push, push_count);
    if (!t || !z || !z->ok()) { delete t; delete z; return fail(RSBWT_EIO, "cannot connect the ZeroMQ sockets (%s, %s, %s)", pull, push, push_count); }
    t->t = z;
    *out = t;
    return RSBWT_OK;
#else
    *out = nullptr;
    return fail(RSBWT_ENODEV, "this build has no libzmq (compile service_loop.cpp with -DRSBWT_WITH_ZMQ -lzmq)");
#endif
}

void rsbwt_transport_free(rsbwt_transport_t *t) {
    if (!t) return;
    delete t->t;
    delete t;
}

int rsbwt_transport_push_request(rsbwt_transport_t *t, const uint8_t *msg, size_t n) {
    if (!t || !t->inproc || (!msg && n)) return fail(RSBWT_EINVAL, "not an in-process transport");
    t->inproc->push_request(msg, n);
    return RSBWT_OK;
}

int rsbwt_transport_pop_reply(rsbwt_transport_t *t, int channel, uint8_t *buf, size_t cap, size_t *n, int64_t timeout_us) {
    if (!t || !t->inproc || !n || channel < 0 || channel > 1) return fail(RSBWT_EINVAL, "not an in-process transport");
    std::vector<uint8_t> m;
    if (!t->inproc->pop_reply(channel, &m, timeout_us)) return fail(RSBWT_EIO, "no reply within %lld us", (long long)timeout_us);
    *n = m.size();
    if (m.size() > cap) return fail(RSBWT_ERANGE, "%zu reply bytes, room for %zu", m.size(), cap);
    if (!m.empty()) memcpy(buf, m.data(), m.size());
    return RSBWT_OK;
}

void rsbwt_transport_close(rsbwt_transport_t *t) {
    if (t && t->inproc) t->inproc->close();
#ifdef RSBWT_WITH_ZMQ
    else if (t) static_cast<zmq_transport *>(t->t)->stop();
#endif
}

// ---- the service ------------------------------------------------------------------------------------
int rsbwt_service_create(rsbwt_set_t *set, rsbwt_transport_t *t, int64_t window_us, size_t max_batch, int per_partition,
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

void rsbwt_service_set_other_handler(rsbwt_service_t *s, rsbwt_service_other_fn fn, void *arg) {
    if (!s) return;
    s->other = fn;
    s->other_arg = arg;
}

int rsbwt_service_run(rsbwt_service_t *s) {
    if (!s) return fail(RSBWT_EINVAL, "null service");
    s->run();
    return s->last_rc == RSBWT_OK ? RSBWT_OK : fail(s->last_rc, "%s", s->last_err.c_str());
}

int rsbwt_service_start(rsbwt_service_t *s) {
    if (!s || s->worker.joinable()) return fail(RSBWT_EINVAL, "null or already running service");
    s->stop.store(false);
    s->worker = std::thread([s] { s->run(); });
    return RSBWT_OK;
}

int rsbwt_service_stop(rsbwt_service_t *s) {
    if (!s) return fail(RSBWT_EINVAL, "null service");
    // a transport that is closing drains: the loop ends by itself once what was received is answered
    if (!s->tr->closing()) s->stop.store(true);
    if (s->worker.joinable()) s->worker.join();
    return s->last_rc == RSBWT_OK ? RSBWT_OK : fail(s->last_rc, "%s", s->last_err.c_str());
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

}  // extern "C"

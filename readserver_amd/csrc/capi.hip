// capi.hip -- the C-ABI of librsbwt.so (include/rsbwt.h) over the HIP engine.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rsbwt.h"
#include "bpi2.h"
#include "bwt_file.h"
#include "capi_internal.h"
#include "kernels.h"
#include "line_format.h"

using namespace rsb;

namespace rsb {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int fail_hip(hipError_t e, const char *what) {
    if (e == hipErrorOutOfMemory) return fail(RSBWT_ENOMEM, "%s: out of HBM", what);
    return fail(RSBWT_EHIP, "%s: %s", what, hipGetErrorString(e));
}

int use_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RSBWT_ENODEV, "no HIP device is visible: the popBWT engine has no CPU fallback");
    if (device < 0 || device >= n) return fail(RSBWT_ENODEV, "device %d out of range (0..%d)", device, n - 1);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    return RSBWT_OK;
}

// The device number an rsbwt_open* caller gives is a LOGICAL device.  In a deployment logical = physical.  Test hook
// (honoured only while RSBWT_ENABLE_TEST_HOOKS is set, read at every open): RSBWT_TEST_DEVICE_ALIASES=N makes the
// numbers 0..N-1 name N logical devices dealt round-robin over the physical ones -- on a one-GPU box all N are
// GPU 0.  A shard set groups its shards by LOGICAL device (sets.hip, make_groups), so the host code of a set that
// spans several devices -- one thread, one context pool, one fused launch per group; the merge of the groups'
// counts, lists and reads -- runs where only one GPU exists.  RCCL refuses two ranks on one physical device: such a
// set takes the paths a box without librccl takes (host-side sums, peer copies for the gather).
int resolve_device(int logical, int *physical) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RSBWT_ENODEV, "no HIP device is visible: the popBWT engine has no CPU fallback");
    const char *hooks = getenv("RSBWT_ENABLE_TEST_HOOKS"), *al = getenv("RSBWT_TEST_DEVICE_ALIASES");
    const int aliases = (hooks && al) ? atoi(al) : 0;
    const int limit = aliases > 0 ? aliases : n;
    if (logical < 0 || logical >= limit) return fail(RSBWT_ENODEV, "device %d out of range (0..%d)", logical, limit - 1);
    *physical = aliases > 0 ? logical % n : logical;
    return RSBWT_OK;
}

// ---- per-call contexts: a stream pair + staging buffer, so that concurrent host callers of one
// handle (the reference shares one BWT* across its pool threads, service.cpp:1513,1532-1569) run
// side by side instead of queueing on one lock.  At most MAX_CTX per handle; further callers wait.
call_ctx *ctx_pool::acquire() {
    return bounded_pool<call_ctx, 8>::acquire([]() -> call_ctx * {
        call_ctx *c = new (std::nothrow) call_ctx();
        if (!c) return nullptr;
        if (hipStreamCreateWithFlags(&c->st[0], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&c->st[1], hipStreamNonBlocking) != hipSuccess) {
            if (c->st[0]) (void)hipStreamDestroy(c->st[0]);
            delete c;
            return nullptr;
        }
        if (hipHostMalloc(&c->h_pin, call_ctx::PIN_BYTES, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->h_pin = nullptr;  // small calls go the ordinary way
        }
        return c;
    });
}

void ctx_pool::destroy() {
    std::lock_guard<std::mutex> lock(mu);
    for (call_ctx *c : free_) {
        (void)hipStreamSynchronize(c->st[0]);
        (void)hipStreamSynchronize(c->st[1]);
        if (c->d_stage) (void)hipFree(c->d_stage);
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        (void)hipStreamDestroy(c->st[0]);
        (void)hipStreamDestroy(c->st[1]);
        delete c;
    }
    free_.clear();
    created = 0;
}

int call_ctx::stage(size_t bytes) {
    if (bytes <= stage_bytes) return RSBWT_OK;
    if (d_stage) {
        (void)hipStreamSynchronize(st[0]);
        (void)hipStreamSynchronize(st[1]);
        (void)hipFree(d_stage);
    }
    d_stage = nullptr;
    stage_bytes = 0;
    hipError_t e = hipMalloc(&d_stage, bytes);
    if (e != hipSuccess) return fail(RSBWT_ENOMEM, "hipMalloc(%zu) for staging: %s", bytes, hipGetErrorString(e));
    stage_bytes = bytes;
    return RSBWT_OK;
}

}  // namespace rsb

namespace {

#ifdef RSB_TIME_HIP_CALLS  // diagnostic build: names any runtime call that takes longer than 20 ms
#include <chrono>
struct slow_call_timer {
    const char *what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit slow_call_timer(const char *w) : what(w) {}
    ~slow_call_timer() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 20.0) fprintf(stderr, "rsbwt: %.1f ms in %s\n", ms, what);
    }
};
#define RSB_TIMED(x) slow_call_timer _timer(#x)
#else
#define RSB_TIMED(x) (void)0
#endif

#define HIP_OK(x)                                              \
    do {                                                       \
        RSB_TIMED(x);                                          \
        hipError_t _e = (x);                                   \
        if (_e != hipSuccess) return fail_hip(_e, #x);         \
    } while (0)

inline uint32_t words_per_kmer(uint32_t k) { return k ? (k + 31u) / 32u : 1u; }

struct ctx_guard {
    ctx_pool &pool;
    call_ctx *c;
    explicit ctx_guard(ctx_pool &p) : pool(p), c(p.acquire()) {}
    ~ctx_guard() { pool.release(c); }
};

// (re)publishes the handle's view for kernels that take it from device memory
int upload_view(rsbwt_t *h) {
    if (!h->d_view) HIP_OK(hipMalloc(&h->d_view, sizeof(shard_view)));
    HIP_OK(hipMemcpy(h->d_view, &h->view, sizeof(shard_view), hipMemcpyHostToDevice));
    return RSBWT_OK;
}

}  // namespace

extern "C" {

const char *rsbwt_version(void) { return "rsbwt 0.2 (gfx950, window lines)"; }

int rsbwt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *rsbwt_last_error(void) { return g_err; }

const char *rsbwt_strerror(int code) {
    switch (code) {
    case RSBWT_OK: return "ok";
    case RSBWT_EINVAL: return "invalid argument";
    case RSBWT_EIO: return "i/o error";
    case RSBWT_EFORMAT: return "not an SGA run-length BWT file";
    case RSBWT_ENOMEM: return "out of memory";
    case RSBWT_ENODEV: return "no usable HIP device";
    case RSBWT_EHIP: return "HIP runtime error";
    case RSBWT_ERANGE: return "shard exceeds format limits";
    case RSBWT_ESYS: return "host runtime error";
    default: return "unknown error";
    }
}

// ---- lifetime -------------------------------------------------------------------------------

}  // extern "C"

namespace rsb {

// The deepest table of *fmt that fits `budget` bytes and whose T-mers are still expected to occur (4^T <= n; grouped:
// 64 times over, a T-mer that does not occur being left to the search: ktab_grouped_sensible).  A grouped table that would not be deeper than the plain
// one, or whose groups would overflow their records, is not worth its escapes: *fmt becomes KTAB_PLAIN.  0 = none.
uint32_t auto_ktab_depth_for(uint64_t budget, uint64_t n, uint32_t *fmt) {
    uint32_t Tp = 1;
    while (Tp < KTAB_MAX_DEPTH_PLAIN && ktab_bytes(KTAB_PLAIN, Tp + 1u) <= budget && (1ull << (2u * (Tp + 1u))) <= n) ++Tp;
    if (*fmt == KTAB_GROUPED) {
        uint32_t Tg = 1;
        while (Tg < KTAB_MAX_DEPTH_GROUPED && ktab_bytes(KTAB_GROUPED, Tg + 1u) <= budget && (64ull << (2u * (Tg + 1u))) <= n) ++Tg;
        if (Tg > Tp && ktab_grouped_sensible(n, Tg)) return Tg;
        *fmt = KTAB_PLAIN;
    }
    return Tp < 2u ? 0u : Tp;
}

int attach_ktab_into(rsbwt *h, uint32_t T, uint64_t *d_table, uint32_t stride, uint32_t fmt) {
    int rc = use_device(h->device);
    if (rc) return rc;
    ctx_guard g(h->pool);
    if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
    uint64_t left = 0;
    hipError_t e = build_ktable(h->view, T, d_table, stride, h->num_cus, g.c->st[0], fmt, &left);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail_hip(e, "building the k-mer table");
    }
    h->view.ktab = d_table;
    h->view.ktab_depth = T;
    h->view.ktab_stride = stride;
    h->view.ktab_fmt = fmt;
    h->ktab_untabulated = left;
    h->hbm_bytes += ktab_bytes(fmt, T);
    if (h->d_xview == h->d_view) h->xview = h->view;  // (a shard opened for reads: one view)
    return upload_view(h);
}

int detach_ktab(rsbwt *h) {
    if (h->ktab_owned || !h->view.ktab) return RSBWT_OK;
    h->hbm_bytes -= ktab_bytes(h->view.ktab_fmt, h->view.ktab_depth);
    h->view.ktab = nullptr;
    h->view.ktab_depth = 0;
    h->view.ktab_stride = 1;
    h->view.ktab_fmt = KTAB_PLAIN;
    h->ktab_untabulated = 0;
    h->ktab_owned = true;
    if (h->d_xview == h->d_view) h->xview = h->view;
    const int rc = use_device(h->device);
    return rc ? rc : upload_view(h);
}

}  // namespace rsb

extern "C" {

// Builds the k-mer table of depth T (2..16; grouped: ..17) for an open handle that has none.
int rsbwt_attach_ktab_format(rsbwt_t *h, uint32_t T, uint32_t format) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (format > RSBWT_KTAB_FORMAT_AUTO) return fail(RSBWT_EINVAL, "k-mer table format %u", format);
    if (h->view.ktab || h->view.n == 0) return RSBWT_OK;
    if (T < 2u) T = 2;
    // auto: the grouped records where they can say what four siblings hold, the plain entries elsewhere
    uint32_t fmt = format == RSBWT_KTAB_FORMAT_AUTO ? (ktab_grouped_sensible(h->view.n, T) ? KTAB_GROUPED : KTAB_PLAIN) : format;
    const uint32_t Tmax = fmt == KTAB_GROUPED ? KTAB_MAX_DEPTH_GROUPED : KTAB_MAX_DEPTH_PLAIN;
    if (T > Tmax) T = Tmax;  // 8 B * 4^16 = 34 GB; 3 B * 4^17 = 52 GB
    int rc = use_device(h->device);
    if (rc) return rc;
    uint64_t *d_tab = nullptr;
    hipError_t e = hipMalloc(&d_tab, ktab_bytes(fmt, T));
    if (e != hipSuccess) return fail_hip(e, "allocating the k-mer table");
    rc = attach_ktab_into(h, T, d_tab, 1, fmt);
    if (rc) (void)hipFree(d_tab);
    else h->ktab_owned = true;
    return rc;
}
int rsbwt_attach_ktab(rsbwt_t *h, uint32_t T) { return rsbwt_attach_ktab_format(h, T, RSBWT_KTAB_FORMAT_PLAIN); }

// The sizing rule as plain arithmetic (host only, no GPU): the deepest table of at most budget_bytes per shard over a
// shard of n_symbols, in the format asked for (AUTO / GROUPED: grouped only where that is deeper and sensible).
int rsbwt_auto_ktab_for_budget(uint64_t budget_bytes, uint64_t n_symbols, uint32_t format_in, uint32_t *depth, uint32_t *format_out) {
    if (!depth || !format_out) return fail(RSBWT_EINVAL, "null argument");
    if (format_in > RSBWT_KTAB_FORMAT_AUTO) return fail(RSBWT_EINVAL, "k-mer table format %u", format_in);
    uint32_t fmt = format_in == RSBWT_KTAB_FORMAT_PLAIN ? KTAB_PLAIN : KTAB_GROUPED;
    *depth = auto_ktab_depth_for(budget_bytes, n_symbols, &fmt);
    *format_out = fmt == KTAB_GROUPED ? RSBWT_KTAB_FORMAT_GROUPED : RSBWT_KTAB_FORMAT_PLAIN;
    return RSBWT_OK;
}

int rsbwt_ktab_info(const rsbwt_t *h, uint32_t *format, uint64_t *bytes, uint64_t *untabulated) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    const bool has = h->view.ktab != nullptr && h->view.ktab_depth != 0u;
    if (format) *format = has ? h->view.ktab_fmt : 0u;
    if (bytes) *bytes = has ? ktab_bytes(h->view.ktab_fmt, h->view.ktab_depth) : 0ull;
    if (untabulated) *untabulated = has ? h->ktab_untabulated : 0ull;
    return RSBWT_OK;
}

static int ensure_select_samples(rsbwt_t *h, hipStream_t stream, bool with_hints = false);

static int finish_open(const void *d_runs, uint64_t num_runs, uint64_t num_strings, int device, int logical,
                       uint32_t flags, rsbwt_t **out) {
    rsbwt_t *h = new (std::nothrow) rsbwt();
    if (!h) return fail(RSBWT_ENOMEM, "host allocation failed");
    h->device = device;
    h->logical_device = logical;
    h->num_strings = num_strings;
    memset(&h->view, 0, sizeof h->view);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        h->num_cus = prop.multiProcessorCount;
    hipError_t e;
    if ((e = hipMalloc(&h->d_work, WORK_WORDS * sizeof(unsigned long long))) != hipSuccess) {
        rsbwt_close(h);
        return fail_hip(e, "allocating counters");
    }
    for (int i = 0; i < rsbwt::RING; ++i) {
        if ((e = hipEventCreate(&h->ev_start[i])) != hipSuccess ||
            (e = hipEventCreate(&h->ev_stop[i])) != hipSuccess) {
            rsbwt_close(h);
            return fail_hip(e, "hipEventCreate");
        }
    }
    build_result br;
    int berr = 0;
    {
        ctx_guard g(h->pool);
        if (!g.c) { rsbwt_close(h); return fail(RSBWT_EHIP, "cannot create a HIP stream"); }
        const uint32_t want_span = (flags & RSBWT_SPAN_MASK) >> RSBWT_SPAN_SHIFT;
        e = build_lines(d_runs, num_runs, want_span, (flags & RSBWT_OPEN_READS) != 0u, g.c->st[0], &br, &berr);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rsbwt_close(h);
        return fail_hip(e, "building the index");
    }
    if (berr == BUILD_ERANGE) {
        rsbwt_close(h);
        return fail(RSBWT_ERANGE, "shard too large: at most 2^40 symbols and 2^32 lines");
    }
    if (berr == BUILD_EFORMAT) {
        rsbwt_close(h);
        return fail(RSBWT_EFORMAT, "run byte with a symbol code above 4 ($ACGT = 0..4, include/bwt/alphabet.h:8-9)");
    }
    h->view = br.view;
    h->view.ktab_stride = 1;
    h->num_runs = br.num_runs;
    h->hbm_bytes = br.hbm_bytes;
    h->far_lines = br.far_lines;
    h->chunk_windows = br.chunk_windows;
    h->far_windows = br.far_windows;
    h->spilled_symbols = br.spilled_symbols;
    int rc = upload_view(h);
    if (rc) { rsbwt_close(h); return rc; }
    // a shard opened for reads gets its select samples and psi hints now: the index is complete, and immutable,
    // before the handle is handed out
    if ((flags & RSBWT_OPEN_READS) != 0u && h->view.n != 0) {
        ctx_guard g(h->pool);
        if (!g.c) { rsbwt_close(h); return fail(RSBWT_EHIP, "cannot create a HIP stream"); }
        rc = ensure_select_samples(h, g.c->st[0], true);
        if (rc) { rsbwt_close(h); return rc; }
        // (nobody has the handle yet: its one view names the samples)
        (void)hipFree(h->d_xview);
        h->view = h->xview;
        h->d_xview = h->d_view;
        if ((rc = upload_view(h)) != RSBWT_OK) { rsbwt_close(h); return rc; }
    }
    // k-mer table: explicit depth, none, or auto = the deepest whose 8-byte entries take no more
    // HBM than 5/4 of the index itself and no more than a quarter of what is still free (HBM is there
    // to be used: every level replaces one LF step, two Occ lookups, of each query by the same
    // single 8-byte read), and whose T-mers still have ~1 expected occurrence (4^T <= n).
    // 8 B * 4^16 = 34 GB is the ceiling.  (rsbwt_set_open sizes the tables of its shards together.)
    uint32_t T = (flags & RSBWT_KTAB_MASK) >> RSBWT_KTAB_SHIFT;
    uint32_t fmt = (flags & RSBWT_OPEN_KTAB_GROUPED) ? KTAB_GROUPED : KTAB_PLAIN;
    if (T == 31u || h->view.n == 0) T = 0;
    else if (T == 0u) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const uint64_t budget = std::min<uint64_t>(h->hbm_bytes + h->hbm_bytes / 4, free_b / 4);
        T = auto_ktab_depth_for(budget, h->view.n, &fmt);
    }
    if (T) {
        rc = rsbwt_attach_ktab_format(h, T, fmt);
        if (rc) { rsbwt_close(h); return rc; }
    }
    *out = h;
    return RSBWT_OK;
}

int rsbwt_open_device_runs(const void *d_runs, uint64_t num_runs, uint64_t num_strings, int device,
                           uint32_t flags, rsbwt_t **out) {
    if (!out || (!d_runs && num_runs)) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    const int logical = device;
    int rc = resolve_device(logical, &device);
    if (rc == RSBWT_OK) rc = use_device(device);
    if (rc) return rc;
    return finish_open(d_runs, num_runs, num_strings, device, logical, flags, out);
}

int rsbwt_open_runs(const uint8_t *runs, uint64_t num_runs, uint64_t num_strings, int device,
                    uint32_t flags, rsbwt_t **out) {
    if (!out || (!runs && num_runs)) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    const int logical = device;
    int rc = resolve_device(logical, &device);
    if (rc == RSBWT_OK) rc = use_device(device);
    if (rc) return rc;
    void *d_runs = nullptr;
    hipError_t e = hipMalloc(&d_runs, num_runs ? num_runs : 16);
    if (e != hipSuccess) return fail(RSBWT_ENOMEM, "hipMalloc(%llu) for run bytes: %s", (unsigned long long)num_runs, hipGetErrorString(e));
    e = hipMemcpy(d_runs, runs, num_runs, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d_runs); return fail_hip(e, "hipMemcpy(runs)"); }
    rc = finish_open(d_runs, num_runs, num_strings, device, logical, flags, out);
    (void)hipFree(d_runs);
    return rc;
}

int rsbwt_open(const char *bwt_path, int device, uint32_t flags, rsbwt_t **out) {
    if (!out || !bwt_path) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    FILE *f = nullptr;
    bwt_header hdr;
    int rc = bwt_open_read(bwt_path, &f, &hdr);
    if (rc == RSBWT_EIO) return fail(rc, "cannot open %s", bwt_path);
    if (rc) return fail(rc, "%s is not an SGA run-length BWT (magic 0xCACA) or is truncated", bwt_path);
    const int logical = device;
    rc = resolve_device(logical, &device);
    if (rc == RSBWT_OK) rc = use_device(device);
    if (rc) { fclose(f); return rc; }
    // stream the file through two pinned buffers into HBM
    const size_t CH = 64u << 20;
    void *d_runs = nullptr, *pin[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipError_t e = hipMalloc(&d_runs, hdr.num_runs ? hdr.num_runs : 16);
    if (e != hipSuccess) { fclose(f); return fail(RSBWT_ENOMEM, "hipMalloc(%llu) for run bytes: %s", (unsigned long long)hdr.num_runs, hipGetErrorString(e)); }
    rc = RSBWT_OK;
    if ((e = hipStreamCreate(&st)) != hipSuccess) rc = fail_hip(e, "hipStreamCreate");
    for (int i = 0; i < 2 && rc == RSBWT_OK; ++i) {
        if ((e = hipHostMalloc(&pin[i], CH, hipHostMallocDefault)) != hipSuccess) rc = fail(RSBWT_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(e));
        else if ((e = hipEventCreate(&done[i])) != hipSuccess) rc = fail_hip(e, "hipEventCreate");
    }
    uint64_t off = 0;
    for (int i = 0; rc == RSBWT_OK && off < hdr.num_runs; i ^= 1) {
        const size_t m = (size_t)std::min<uint64_t>(CH, hdr.num_runs - off);
        if ((e = hipEventSynchronize(done[i])) != hipSuccess) { rc = fail_hip(e, "hipEventSynchronize"); break; }
        if (fread(pin[i], 1, m, f) != m) { rc = fail(RSBWT_EIO, "short read from %s", bwt_path); break; }
        if ((e = hipMemcpyAsync((uint8_t *)d_runs + off, pin[i], m, hipMemcpyHostToDevice, st)) != hipSuccess ||
            (e = hipEventRecord(done[i], st)) != hipSuccess) { rc = fail_hip(e, "hipMemcpyAsync(runs)"); break; }
        off += m;
    }
    if (st) (void)hipStreamSynchronize(st);
    fclose(f);
    if (rc == RSBWT_OK) {
        rc = finish_open(d_runs, hdr.num_runs, hdr.num_strings, device, logical, flags, out);
        if (rc == RSBWT_OK && (*out)->view.n != hdr.num_symbols) {
            rsbwt_close(*out);
            *out = nullptr;
            rc = fail(RSBWT_EFORMAT, "%s: header says %llu symbols, runs hold a different number", bwt_path, (unsigned long long)hdr.num_symbols);
        }
    }
    for (int i = 0; i < 2; ++i) {
        if (done[i]) (void)hipEventDestroy(done[i]);
        if (pin[i]) (void)hipHostFree(pin[i]);
    }
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(d_runs);
    return rc;
}

void rsbwt_close(rsbwt_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->pool.destroy();
    if (h->view.lines) (void)hipFree((void *)h->view.lines);
    if (h->view.ktab && h->ktab_owned) (void)hipFree((void *)h->view.ktab);
    if (h->d_xview && h->d_xview != h->d_view) (void)hipFree(h->d_xview);
    if (h->d_view) (void)hipFree(h->d_view);
    if (h->d_sel) (void)hipFree(h->d_sel);
    if (h->d_work) (void)hipFree(h->d_work);
    h->scratch.destroy();
    for (int i = 0; i < rsbwt::RING; ++i) {
        if (h->ev_start[i]) (void)hipEventDestroy(h->ev_start[i]);
        if (h->ev_stop[i]) (void)hipEventDestroy(h->ev_stop[i]);
    }
    delete h;
}

// ---- shape ----------------------------------------------------------------------------------

static inline int rank_of(char b) { return b == 'A' ? 1 : b == 'C' ? 2 : b == 'G' ? 3 : b == 'T' ? 4 : 0; }

uint64_t rsbwt_bwlen(const rsbwt_t *h) { return h->view.n; }
uint64_t rsbwt_pc(const rsbwt_t *h, char b) { return h->view.C[rank_of(b)]; }
char rsbwt_f(const rsbwt_t *h, uint64_t index) {
    int ci = 0;
    while (ci < 5 && h->view.C[ci] <= index) ci++;
    return "$ACGT"[ci > 0 ? ci - 1 : 0];
}
uint64_t rsbwt_num_runs(const rsbwt_t *h) { return h->num_runs; }
uint64_t rsbwt_num_strings(const rsbwt_t *h) { return h->num_strings; }
uint64_t rsbwt_num_lines(const rsbwt_t *h) { return h->view.nlines; }
uint32_t rsbwt_ktab_depth(const rsbwt_t *h) { return h->view.ktab_depth; }
uint32_t rsbwt_window_span(const rsbwt_t *h) { return h->view.n ? h->view.sp.S : 0u; }
uint64_t rsbwt_far_lines(const rsbwt_t *h) { return h->far_lines; }
uint64_t rsbwt_spilled_symbols(const rsbwt_t *h) { return h->spilled_symbols; }
uint64_t rsbwt_hbm_bytes(const rsbwt_t *h) { return h->hbm_bytes; }
uint64_t rsbwt_psi_hint_lines(const rsbwt_t *h) { return h->psi_hint_lines; }
int rsbwt_opened_for_reads(const rsbwt_t *h) { return h->view.hint_room ? 1 : 0; }
int rsbwt_device(const rsbwt_t *h) { return h->device; }
int rsbwt_logical_device(const rsbwt_t *h) { return h->logical_device; }

// Test hook: w[i] = p[i] / S, r[i] = p[i] % S as the KERNELS compute them (fast_window); host buffers.
int rsbwt_debug_fast_window(const uint64_t *p, size_t n, uint32_t S, uint32_t *w, uint32_t *r, int device) {
    if ((!p || !w || !r) && n) return fail(RSBWT_EINVAL, "null argument");
    if (S < 2u || S > MAX_SPAN) return fail(RSBWT_ERANGE, "span %u outside 2..%u", S, MAX_SPAN);
    if (n == 0) return RSBWT_OK;
    int rc = use_device(device);
    if (rc) return rc;
    uint8_t *d = nullptr;
    HIP_OK(hipMalloc(&d, n * 16));
    hipError_t e = hipMemcpy(d, p, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_debug_fast_window(d, n, S, d + n * 8, d + n * 12, nullptr);
    if (e == hipSuccess) e = hipMemcpy(w, d + n * 8, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(r, d + n * 12, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "rsbwt_debug_fast_window");
}

// Test hook: overwrites n bytes of the index in HBM (region 0: the lines, 1: the k-mer table) with `bytes`,
// so that a test can hold the kernels to what they do with a DAMAGED index (never read outside it, every
// wave drains); answers no query.
int rsbwt_debug_poke(rsbwt_t *h, int region, uint64_t offset, const void *bytes, size_t n) {
    if (!h || (!bytes && n)) return fail(RSBWT_EINVAL, "null argument");
    // (a write into a published index: refused unless the process asked for the test hooks)
    if (getenv("RSBWT_ENABLE_TEST_HOOKS") == nullptr) return fail(RSBWT_EINVAL, "rsbwt_debug_poke is a test hook: set RSBWT_ENABLE_TEST_HOOKS=1");
    const uint64_t size = region == 0 ? h->view.nlines * (uint64_t)LINE_BYTES
                          : region == 1 && h->view.ktab && h->ktab_owned ? ktab_bytes(h->view.ktab_fmt, h->view.ktab_depth) : 0ull;
    if (offset > size || n > size - offset) return fail(RSBWT_ERANGE, "poke outside the region (%llu bytes)", (unsigned long long)size);
    if (n == 0) return RSBWT_OK;
    int rc = use_device(h->device);
    if (rc != RSBWT_OK) return rc;
    char *base = region == 0 ? (char *)h->view.lines : (char *)h->view.ktab;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(base + offset, bytes, n, hipMemcpyHostToDevice);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "rsbwt_debug_poke");
}

// Test hook, the reading twin of rsbwt_debug_poke: n bytes of the index in HBM (region 0: the lines, 1: the k-mer
// table) copied to `bytes` -- so that a test can hold "nothing a search reads is written after the handle was handed
// out" to the bytes themselves.  Answers no query.
int rsbwt_debug_peek(rsbwt_t *h, int region, uint64_t offset, void *bytes, size_t n) {
    if (!h || (!bytes && n)) return fail(RSBWT_EINVAL, "null argument");
    if (getenv("RSBWT_ENABLE_TEST_HOOKS") == nullptr) return fail(RSBWT_EINVAL, "rsbwt_debug_peek is a test hook: set RSBWT_ENABLE_TEST_HOOKS=1");
    const uint64_t size = region == 0 ? h->view.nlines * (uint64_t)LINE_BYTES
                          : region == 1 && h->view.ktab && h->ktab_owned ? ktab_bytes(h->view.ktab_fmt, h->view.ktab_depth) : 0ull;
    if (offset > size || n > size - offset) return fail(RSBWT_ERANGE, "peek outside the region (%llu bytes)", (unsigned long long)size);
    if (n == 0) return RSBWT_OK;
    int rc = use_device(h->device);
    if (rc != RSBWT_OK) return rc;
    const char *base = region == 0 ? (const char *)h->view.lines : (const char *)h->view.ktab;
    const hipError_t e = hipMemcpy(bytes, base + offset, n, hipMemcpyDeviceToHost);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "rsbwt_debug_peek");
}

// ---- class BWT mirrors ------------------------------------------------------------------------

// The sampled select table (getOccAt, read extraction): built once -- at open for a shard opened with
// RSBWT_OPEN_READS (with the psi hints, into lines nobody reads yet), else on first use INTO A SIDE TABLE: nothing
// the searches read (h->view, h->d_view, the lines) is written.  with_hints: also write psi hints into the window
// lines that have room for one -- only before the handle is shared (finish_open, rsbwt_prepare_extraction).
static int ensure_select_samples(rsbwt_t *h, hipStream_t stream, bool with_hints) {
    if (h->x_ready.load(std::memory_order_acquire) && !with_hints) return RSBWT_OK;
    std::lock_guard<std::mutex> lock(h->mu);
    const bool have = h->x_ready.load(std::memory_order_relaxed);
    if (have && (!with_hints || h->psi_hint_lines != 0 || getenv("RSBWT_NO_PSI_HINTS") != nullptr)) return RSBWT_OK;
    const uint64_t stride_m = select_sample_stride(h->view);
    const uint64_t words = 5 * stride_m;
    uint64_t *d = h->d_sel;
    hipError_t e = hipSuccess;
    if (!have) {
        HIP_OK(hipMalloc(&d, words * sizeof(uint64_t)));
        e = hipMemsetAsync(d, 0, words * sizeof(uint64_t), stream);
        if (e == hipSuccess) e = launch_select_samples(h->view, d, stream);
    }
    shard_view with = h->view;
    with.sel = d;
    with.sel_stride = stride_m;
    // the psi hints, into the window lines that have room for one (line_format.h): extraction's select then needs no
    // sample for the rows of those windows.  A hint sits where a line holds no piece of its own, behind a header bit
    // no search reads -- but it is a write into the index: only while the owner alone holds the handle.
    unsigned long long *d_made = nullptr, made = 0;
    if (e == hipSuccess && with_hints && getenv("RSBWT_NO_PSI_HINTS") == nullptr) {
        e = hipMalloc(&d_made, sizeof made);
        if (e == hipSuccess) e = hipMemsetAsync(d_made, 0, sizeof made, stream);
        if (e == hipSuccess) e = launch_psi_hints(with, d_made, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&made, d_made, sizeof made, hipMemcpyDeviceToHost, stream);
    }
    shard_view *dx = h->d_xview;
    if (e == hipSuccess && !have) {
        e = hipMalloc(&dx, sizeof(shard_view));
        if (e == hipSuccess) e = hipMemcpyAsync(dx, &with, sizeof(shard_view), hipMemcpyHostToDevice, stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (d_made) (void)hipFree(d_made);
    if (e != hipSuccess) {
        if (!have) {
            (void)hipFree(d);
            if (dx) (void)hipFree(dx);
        }
        return fail_hip(e, "select sample kernel");
    }
    if (with_hints) h->psi_hint_lines = made;
    if (!have) {
        h->hbm_bytes += words * sizeof(uint64_t);
        h->d_sel = d;
        h->xview = with;
        h->d_xview = dx;
        h->x_ready.store(true, std::memory_order_release);
    }
    return RSBWT_OK;
}

// Open-time step for a shard of the PLAIN layout that will serve reads all the same (a shard opened with
// RSBWT_OPEN_READS needs none): its select samples now, and a psi hint in every window line that has room for one
// (about two thirds of them at the span the builder picks).  It writes into the resident lines: the owner calls it
// BEFORE it shares the handle, like rsbwt_attach_ktab.  Without it such a shard builds its samples on its first
// extraction -- into a side table, safely beside any other call -- and extracts without hints (more requests per step).
int rsbwt_prepare_extraction(rsbwt_t *h) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (h->view.n == 0 || h->view.hint_room) return RSBWT_OK;
    int rc = use_device(h->device);
    if (rc) return rc;
    ctx_guard g(h->pool);
    if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
    return ensure_select_samples(h, g.c->st[0], true);
}

}  // extern "C"
namespace rsb {
int ensure_samples(rsbwt *h, hipStream_t stream) { return ensure_select_samples(h, stream); }
}  // namespace rsb
extern "C" {

// kind 0: occ(syms, vals)  1: char(vals)  2: occ_at(syms, vals)
static int mirror_batch(rsbwt_t *h, int kind, const char *syms, const uint64_t *vals, size_t n, void *out) {
    if (!h || (!vals && n) || (!out && n) || (kind != 1 && !syms && n)) return fail(RSBWT_EINVAL, "null argument");
    if (n == 0) return RSBWT_OK;
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    ctx_guard g(h->pool);
    if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
    hipStream_t st = g.c->st[0];
    if (kind == 2 && (rc = ensure_select_samples(h, st)) != RSBWT_OK) return rc;
    const size_t out_bytes = (kind == 1) ? n : n * 8;
    const size_t need = n * 8 + n + out_bytes + 64;
    if ((rc = g.c->stage(need)) != RSBWT_OK) return rc;
    uint8_t *base = (uint8_t *)g.c->d_stage;
    uint64_t *d_vals = (uint64_t *)base;
    uint8_t *d_out = base + n * 8;              // 8-aligned
    uint8_t *d_syms = d_out + ((out_bytes + 7) & ~(size_t)7);
    HIP_OK(hipMemcpyAsync(d_vals, vals, n * 8, hipMemcpyHostToDevice, st));
    if (kind != 1) HIP_OK(hipMemcpyAsync(d_syms, syms, n, hipMemcpyHostToDevice, st));
    hipError_t e = hipSuccess;
    if (kind == 0) e = launch_occ_batch(h->view, d_syms, d_vals, n, d_out, st);
    else if (kind == 1) e = launch_char_batch(h->view, d_vals, n, d_out, st);
    else e = launch_occ_at_batch(h->xview, d_syms, d_vals, n, d_out, st);
    if (e != hipSuccess) return fail_hip(e, "mirror kernel launch");
    HIP_OK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return RSBWT_OK;
}

int rsbwt_occ_batch(rsbwt_t *h, const char *b, const uint64_t *index, size_t n, uint64_t *occ) {
    return mirror_batch(h, 0, b, index, n, occ);
}
int rsbwt_char_batch(rsbwt_t *h, const uint64_t *index, size_t n, char *c) {
    return mirror_batch(h, 1, nullptr, index, n, c);
}
int rsbwt_occ_at_batch(rsbwt_t *h, const char *b, const uint64_t *bc, size_t n, uint64_t *index) {
    return mirror_batch(h, 2, b, bc, n, index);
}
int rsbwt_occ(rsbwt_t *h, char b, uint64_t index, uint64_t *occ) { return mirror_batch(h, 0, &b, &index, 1, occ); }
int rsbwt_char(rsbwt_t *h, uint64_t index, char *c) { return mirror_batch(h, 1, nullptr, &index, 1, c); }
int rsbwt_occ_at(rsbwt_t *h, char b, uint64_t bc, uint64_t *index) { return mirror_batch(h, 2, &b, &bc, 1, index); }

// ---- .bpi2, the reference's FM-index file -----------------------------------------

int rsbwt_bpi2_write(const char *bwt_path, const char *bpi2_path) {
    if (!bwt_path || !bpi2_path) return fail(RSBWT_EINVAL, "null argument");
    try {
        bpi2_index ix;
        std::string err;
        int rc = bpi2_from_bwt(bwt_path, &ix, &err);
        if (rc == RSBWT_OK) rc = bpi2_save(ix, bpi2_path, &err);
        if (rc != RSBWT_OK) return fail(rc, "%s", err.c_str());
    } catch (const std::bad_alloc &) {
        return fail(RSBWT_ENOMEM, "host allocation failed while building the .bpi2 index");
    }
    return RSBWT_OK;
}

int rsbwt_bpi2_validate_file(const char *bpi2_path) {
    if (!bpi2_path) return fail(RSBWT_EINVAL, "null argument");
    try {
        bpi2_index f;
        std::string err;
        const int rc = bpi2_load(bpi2_path, &f, &err);
        if (rc != RSBWT_OK) return fail(rc, "%s", err.c_str());
    } catch (const std::bad_alloc &) {
        return fail(RSBWT_ENOMEM, "host allocation failed while reading %s", bpi2_path);
    }
    return RSBWT_OK;
}

static int bpi2_check_impl(rsbwt_t *h, const char *bpi2_path, uint64_t max_samples, uint64_t *checked,
                           uint64_t *mismatches) {
    bpi2_index f;
    std::string err;
    int rc = bpi2_load(bpi2_path, &f, &err);
    if (rc != RSBWT_OK) return fail(rc, "%s", err.c_str());
    uint64_t bad = 0;
    std::string first;
    auto note = [&](const std::string &m) {
        if (bad++ == 0) first = m;
    };
    // shape: the levels RLEBWT::initialiseFMIndex plans for this many runs (rlebwt.cpp:46-78)
    const bpi2_builder plan(h->num_runs);
    if (f.levels.size() != plan.ix.levels.size()) {
        fail(RSBWT_OK, "%s: %zu counter levels, %zu expected for %llu runs", bpi2_path, f.levels.size(),
             plan.ix.levels.size(), (unsigned long long)h->num_runs);
        *mismatches = 1;
        return RSBWT_OK;
    }
    for (size_t k = 0; k < f.levels.size(); ++k) {
        const bpi2_level &a = f.levels[k], &b = plan.ix.levels[k];
        const uint64_t len = h->num_runs ? (h->num_runs - 1) / b.bucket + 1 : 0;
        if (a.width != b.width || a.block != b.block || a.bucket != b.bucket || a.length != len)
            note("level " + std::to_string(k) + ": width/block/bucket/length differ from the plan for this run count");
    }
    // C[]
    static const char SYM[5] = {'$', 'A', 'C', 'G', 'T'};
    for (int c = 0; c < 5; ++c)
        if (f.pc[c] != h->view.C[c]) note(std::string("C[") + SYM[c] + "] differs");
    if (bad) {
        fail(RSBWT_OK, "%s: %s", bpi2_path, first.c_str());
        *mismatches = bad;
        return RSBWT_OK;
    }
    // counters: absolute counts at the start of sampled 64-run buckets = sum over the levels of
    // the entries covering the bucket; they must equal Occ(c, position - 1) of the resident index
    const bpi2_level &bot = f.levels.back();
    const uint64_t nb = bot.length;
    uint64_t S = max_samples ? std::min<uint64_t>(max_samples, nb) : nb;
    std::vector<uint64_t> pos, want;
    std::vector<char> sym;
    std::vector<uint64_t> idx;
    for (uint64_t s = 0; s < S; ++s) {
        const uint64_t p = (S == nb) ? s : (s + 1 == S ? nb - 1 : (uint64_t)((unsigned __int128)s * nb / S));
        uint64_t abs[5] = {0, 0, 0, 0, 0};
        for (size_t k = 0; k < f.levels.size(); ++k) {
            const bpi2_level &l = f.levels[k];
            const uint64_t e = p * bot.bucket / l.bucket;
            const uint64_t mask = l.width == 8 ? ~0ull : (1ull << (8 * l.width)) - 1ull;
            uint64_t sum = 0;
            for (int c = 0; c < 5; ++c) {
                abs[c] += l.counts[e * 5 + c];
                sum += l.counts[e * 5 + c];
            }
            if ((sum & mask) != l.sums[e]) note("level " + std::to_string(k) + " entry " + std::to_string(e) + ": sum field differs from its counts");
        }
        const uint64_t at = abs[0] + abs[1] + abs[2] + abs[3] + abs[4];
        for (int c = 0; c < 5; ++c) {
            if (at == 0) {
                if (abs[c] != 0) note("bucket 0 does not start at zero");
                continue;
            }
            sym.push_back(SYM[c]);
            idx.push_back(at - 1);
            want.push_back(abs[c]);
            pos.push_back(p);
        }
    }
    std::vector<uint64_t> got(idx.size());
    if (!idx.empty()) {
        rc = rsbwt_occ_batch(h, sym.data(), idx.data(), idx.size(), got.data());
        if (rc != RSBWT_OK) return rc;
        for (size_t i = 0; i < idx.size(); ++i)
            if (got[i] != want[i] || idx[i] >= h->view.n)
                note("bucket " + std::to_string(pos[i]) + ": count of '" + sym[i] + "' is " + std::to_string(want[i]) +
                     ", the index has " + std::to_string(got[i]));
    }
    // vSum[t]: the bucket of the run that reaches symbol t * 65,536 (rlebwt.cpp:103-106)
    auto before = [&](uint64_t p) {
        uint64_t at = 0;
        for (const bpi2_level &l : f.levels) at += l.sums[p * bot.bucket / l.bucket];
        return at;
    };
    for (size_t t = 0; t < f.vsum.size(); ++t) {
        const uint64_t b = f.vsum[t], mark = (uint64_t)t << 16;
        if (b >= nb) { note("vSum entry beyond the last bucket"); continue; }
        if (t == 0) { if (b != 0) note("vSum[0] is not 0"); continue; }
        if (!(before(b) < mark) || (b + 1 < nb && before(b + 1) < mark)) note("vSum[" + std::to_string(t) + "] names the wrong bucket");
    }
    *checked = S;
    *mismatches = bad;
    if (bad) fail(RSBWT_OK, "%s: %s", bpi2_path, first.c_str());
    return RSBWT_OK;
}

int rsbwt_bpi2_check(rsbwt_t *h, const char *bpi2_path, uint64_t max_samples, uint64_t *checked,
                     uint64_t *mismatches) {
    if (!h || !bpi2_path || !checked || !mismatches) return fail(RSBWT_EINVAL, "null argument");
    *checked = *mismatches = 0;
    try {
        return bpi2_check_impl(h, bpi2_path, max_samples, checked, mismatches);
    } catch (const std::bad_alloc &) {
        return fail(RSBWT_ENOMEM, "host allocation failed while checking %s", bpi2_path);
    }
}

// ---- batched search ---------------------------------------------------------------------------

int rsbwt_pack_kmers_dev(const void *d_kmers, size_t Q, uint32_t k, size_t stride, void *d_packed,
                         void *d_valid, int device, void *stream) {
    if ((!d_kmers || !d_packed || !d_valid) && Q) return fail(RSBWT_EINVAL, "null argument");
    if (stride < k) return fail(RSBWT_EINVAL, "stride %zu < k %u", stride, k);
    int rc = use_device(device);
    if (rc) return rc;
    hipError_t e = launch_pack(d_kmers, Q, k, stride, d_packed, d_valid, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "pack kernel launch");
    return RSBWT_OK;
}

}  // extern "C"

namespace rsb {

// One fused search launch over the shards whose device views are d_views[0..nshards) on the current
// device, timed and counted through `m`.  k >= 1 (k == 0 is answered on the host by the callers that
// allow it); k <= 65535 (the resume position of a traced search is a 16-bit field).
int search_launch(search_meter &m, const shard_view *d_views, uint32_t nshards, int num_cus, const void *d_packed,
                  const void *d_valid, size_t Q, uint32_t k, void *d_lower, void *d_upper, bool counts_only,
                  hipStream_t stream, const search_extra *extra) {
    const bool pairs = extra && (extra->pairs || extra->d_hit_bits);
    if (Q && (!d_packed || !d_valid || !d_lower || (!counts_only && !pairs && !d_upper))) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0) return fail(RSBWT_EINVAL, "k must be at least 1 for device-resident searches");
    if (k > 65535u) return fail(RSBWT_ERANGE, "k %u: at most 65535 symbols per k-mer", k);
    std::lock_guard<std::mutex> lock(m.mu);
    unsigned long long *work = nullptr;
    if (m.counting) {
        work = m.d_work;
        HIP_OK(hipMemsetAsync(work, 0, WORK_WORDS * sizeof(unsigned long long), stream));
    }
    const int slot = (int)(m.launches % search_meter::RING);
    hipError_t e = launch_search(m.scratch, d_views, nshards, d_packed, d_valid, Q, k, d_lower, d_upper, counts_only, work, num_cus,
                                 stream, m.ev_start[slot], m.ev_stop[slot], extra);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail_hip(e, "search kernel launch");
    }
    m.launches++;
    return RSBWT_OK;
}

int search_launch_walk(search_meter &m, const shard_view *d_views, uint32_t nshards, int num_cus, const void *d_packed,
                       const void *d_valid, size_t nkmers, uint32_t tn, void *d_worklists, void *d_counts, size_t wl_cap, uint32_t k,
                       void *d_sparse, void *d_hit_bits, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(m.mu);
    unsigned long long *work = nullptr;
    if (m.counting) {
        work = m.d_work;
        HIP_OK(hipMemsetAsync(work, 0, WORK_WORDS * sizeof(unsigned long long), stream));
    }
    const int slot = (int)(m.launches % search_meter::RING);
    hipError_t e = launch_search_walk(m.scratch, d_views, nshards, d_packed, d_valid, nkmers, tn, d_worklists, d_counts, wl_cap, k, d_sparse,
                                      d_hit_bits, work, num_cus, stream, m.ev_start[slot], m.ev_stop[slot]);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail_hip(e, "1-mismatch walk kernel launch");
    }
    m.launches++;
    return RSBWT_OK;
}

int search_launch_worklist(search_meter &m, const shard_view *d_views, uint32_t nshards, int num_cus, const void *d_packed,
                           const void *d_valid, size_t nkmers, uint32_t tn, const void *d_worklists, const void *d_counts, size_t wl_cap,
                           uint32_t k, void *d_sparse, void *d_hit_bits, hipStream_t stream, const void *d_pre) {
    std::lock_guard<std::mutex> lock(m.mu);
    const int slot = (int)(m.launches % search_meter::RING);
    hipError_t e = launch_search_worklist(m.scratch, d_views, nshards, d_packed, d_valid, nkmers, tn, d_worklists, d_counts, wl_cap, k, d_sparse,
                                          d_hit_bits, m.counting ? m.d_work : nullptr, num_cus, stream, m.ev_start[slot], m.ev_stop[slot], d_pre);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail_hip(e, "worklist search kernel launch");
    }
    m.launches++;
    return RSBWT_OK;
}

}  // namespace rsb

static int search_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                      void *d_lower, void *d_upper, bool counts_only, hipStream_t stream,
                      const search_extra *extra = nullptr) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    int rc = use_device(h->device);
    if (rc) return rc;
    if (h->view.n == 0 && Q) return fail(RSBWT_EINVAL, "empty index");
    search_extra ex = extra ? *extra : search_extra();
    ex.narrow = view_is_narrow(h->view, k);
    return search_launch(*h, h->d_view, 1, h->num_cus, d_packed, d_valid, Q, k, d_lower, d_upper, counts_only, stream, &ex);
}

// Host buffers: slices of at most 2M k-mers alternate between the context's two streams and two
// halves of its staging buffer: while the host sits in slice i's copy back, the GPU already
// searches slice i + 1 (its k-mers went up before that copy was issued).  `nshards` result rows of
// Q values each come back per slice (lower[s * Q + q]).
namespace rsb {
int search_host_views(search_meter &m, ctx_pool &pool, const shard_view *d_views, uint32_t nshards, int num_cus,
                      const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *lower, uint64_t *upper,
                      bool counts_only, bool narrow) {
    const uint32_t wpq = words_per_kmer(k);
    search_extra ex;
    ex.narrow = narrow;
    ctx_guard g(pool);
    if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
    const size_t SLICE = std::max<size_t>(1u << 16, (2u << 20) / nshards);
    const size_t m_max = std::min(SLICE, Q);
    const size_t a_ascii = (((m_max - 1) * stride + k) + 15) & ~(size_t)15;
    const size_t a_packed = m_max * wpq * 8;
    const size_t a_valid = (m_max + 15) & ~(size_t)15;
    const size_t a_res = (size_t)nshards * m_max * 8;
    const size_t half = a_ascii + a_packed + a_valid + 2 * a_res;
    int rc;
    if ((rc = g.c->stage(Q > SLICE ? 2 * half : half)) != RSBWT_OK) return rc;
    // A small call (everything fits the context's page-locked buffer): k-mers and answers travel through
    // that buffer, one upload and one download, both truly asynchronous.
    {
        const size_t ascii_bytes = (Q - 1) * stride + k, res_bytes = (size_t)nshards * Q * 8;
        const size_t h_res = (ascii_bytes + 15) & ~(size_t)15;
        if (g.c->h_pin && Q <= SLICE && h_res + 2 * res_bytes <= call_ctx::PIN_BYTES) {
            uint8_t *hp = (uint8_t *)g.c->h_pin;
            uint8_t *d_ascii = (uint8_t *)g.c->d_stage, *d_packed = d_ascii + a_ascii, *d_valid = d_packed + a_packed;
            uint8_t *d_lo = d_valid + a_valid;  // d_up follows at d_lo + a_res = d_lo + res_bytes (m_max = Q)
            hipStream_t st = g.c->st[0];
            memcpy(hp, kmers, ascii_bytes);
            HIP_OK(hipMemcpyAsync(d_ascii, hp, ascii_bytes, hipMemcpyHostToDevice, st));
            hipError_t e = launch_pack(d_ascii, Q, k, stride, d_packed, d_valid, st);
            if (e != hipSuccess) return fail_hip(e, "pack kernel launch");
            rc = search_launch(m, d_views, nshards, num_cus, d_packed, d_valid, Q, k, d_lo, d_lo + a_res, counts_only, st, &ex);
            if (rc) return rc;
            HIP_OK(hipMemcpyAsync(hp + h_res, d_lo, (counts_only ? 1 : 2) * res_bytes, hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            memcpy(lower, hp + h_res, res_bytes);
            if (!counts_only) memcpy(upper, hp + h_res + res_bytes, res_bytes);
            return RSBWT_OK;
        }
    }
    struct slice_t {
        size_t q0 = 0, m = 0;
        uint8_t *d_lo = nullptr, *d_up = nullptr;
        hipStream_t st = nullptr;
    } prev;
    auto collect = [&](const slice_t &sl) -> int {  // results of a slice whose search is under way
        if (sl.m == Q) {  // the only slice: [nshards][Q] in HBM is the caller's layout, one copy per array
            HIP_OK(hipMemcpyAsync(lower, sl.d_lo, (size_t)nshards * Q * 8, hipMemcpyDeviceToHost, sl.st));
            if (!counts_only) HIP_OK(hipMemcpyAsync(upper, sl.d_up, (size_t)nshards * Q * 8, hipMemcpyDeviceToHost, sl.st));
            HIP_OK(hipStreamSynchronize(sl.st));
            return RSBWT_OK;
        }
        for (uint32_t s = 0; s < nshards; ++s) {
            HIP_OK(hipMemcpyAsync(lower + s * Q + sl.q0, sl.d_lo + (size_t)s * sl.m * 8, sl.m * 8, hipMemcpyDeviceToHost, sl.st));
            if (!counts_only)
                HIP_OK(hipMemcpyAsync(upper + s * Q + sl.q0, sl.d_up + (size_t)s * sl.m * 8, sl.m * 8, hipMemcpyDeviceToHost, sl.st));
        }
        HIP_OK(hipStreamSynchronize(sl.st));
        return RSBWT_OK;
    };
    size_t i = 0;
    for (size_t q0 = 0; q0 < Q; q0 += SLICE, ++i) {
        slice_t cur;
        cur.q0 = q0;
        cur.m = std::min(SLICE, Q - q0);
        cur.st = g.c->st[i & 1];
        uint8_t *base = (uint8_t *)g.c->d_stage + (i & 1) * half;
        uint8_t *d_ascii = base, *d_packed = d_ascii + a_ascii, *d_valid = d_packed + a_packed;
        cur.d_lo = d_valid + a_valid;
        cur.d_up = cur.d_lo + a_res;
        const size_t ascii_bytes = (cur.m - 1) * stride + k;  // the last k-mer needs only k bytes
        HIP_OK(hipMemcpyAsync(d_ascii, kmers + q0 * stride, ascii_bytes, hipMemcpyHostToDevice, cur.st));
        {
            RSB_TIMED(launch_pack);
            hipError_t e = launch_pack(d_ascii, cur.m, k, stride, d_packed, d_valid, cur.st);
            if (e != hipSuccess) return fail_hip(e, "pack kernel launch");
        }
        {
            RSB_TIMED(search_launch);
            rc = search_launch(m, d_views, nshards, num_cus, d_packed, d_valid, cur.m, k, cur.d_lo, cur.d_up, counts_only, cur.st, &ex);
        }
        if (rc) return rc;
        if (prev.m && (rc = collect(prev)) != RSBWT_OK) return rc;
        prev = cur;
    }
    if (prev.m && (rc = collect(prev)) != RSBWT_OK) return rc;
    return RSBWT_OK;
}

// The same for queries of LENGTHS OF THEIR OWN: query q = text[off[q] .. off[q+1]) (search_lines.hip, search_init_var_kernel:
// a start record says where its search goes on).  lower / upper [nshards][Q]; an empty query, one with a symbol outside
// ACGT or one longer than 65,535 symbols ends as (1, 0) / count 0.
int search_host_views_var(search_meter &m, ctx_pool &pool, const shard_view *d_views, uint32_t nshards, int num_cus,
                          const char *text, const uint64_t *off, size_t Q, uint64_t *lower, uint64_t *upper, bool counts_only,
                          bool narrow) {
    ctx_guard g(pool);
    if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
    hipStream_t st = g.c->st[0];
    const size_t SLICE = 1u << 16;
    std::vector<uint64_t> rel;
    for (size_t q0 = 0; q0 < Q; q0 += SLICE) {
        const size_t mq = std::min(SLICE, Q - q0);
        uint64_t kmax = 0;
        for (size_t i = 0; i < mq; ++i) {
            if (off[q0 + i + 1] < off[q0 + i]) return fail(RSBWT_EINVAL, "query %zu: its end lies before its start", q0 + i);
            const uint64_t n = off[q0 + i + 1] - off[q0 + i];
            if (n <= 65535ull) kmax = std::max(kmax, n);  // (a longer one is refused by the packing: it does not size the batch)
        }
        if (kmax == 0) {  // nothing to search in this slice
            for (uint32_t s = 0; s < nshards; ++s)
                for (size_t i = 0; i < mq; ++i) {
                    lower[s * Q + q0 + i] = counts_only ? 0 : 1;
                    if (!counts_only) upper[s * Q + q0 + i] = 0;
                }
            continue;
        }
        const uint32_t k = (uint32_t)kmax, wpq = words_per_kmer(k);
        const size_t tb = (size_t)(off[q0 + mq] - off[q0]);
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t a_text = al(tb + 16), a_off = al((mq + 1) * 8), a_pk = al(mq * wpq * 8), a_ok = al(mq), a_len = al(mq * 4),
                     a_rec = al((size_t)nshards * mq * 16), a_res = al((size_t)nshards * mq * 8);
        int rc;
        if ((rc = g.c->stage(a_text + a_off + a_pk + a_ok + a_len + a_rec + 2 * a_res)) != RSBWT_OK) return rc;
        uint8_t *d_text = (uint8_t *)g.c->d_stage, *d_off = d_text + a_text, *d_pk = d_off + a_off, *d_ok = d_pk + a_pk, *d_len = d_ok + a_ok,
                *d_rec = d_len + a_len, *d_lo = d_rec + a_rec, *d_up = d_lo + a_res;
        rel.resize(mq + 1);
        for (size_t i = 0; i <= mq; ++i) rel[i] = off[q0 + i] - off[q0];
        if (tb) HIP_OK(hipMemcpyAsync(d_text, text + off[q0], tb, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_off, rel.data(), (mq + 1) * 8, hipMemcpyHostToDevice, st));
        hipError_t e = launch_pack_var(d_text, d_off, mq, wpq, d_pk, d_ok, d_len, st);
        if (e != hipSuccess) return fail_hip(e, "pack kernel launch");
        e = launch_search_init_var(d_views, nshards, d_pk, d_ok, d_len, mq, wpq, d_rec, st);
        if (e != hipSuccess) return fail_hip(e, "start-record kernel launch");
        search_extra ex;
        ex.narrow = narrow;
        ex.d_init = d_rec;
        if ((rc = search_launch(m, d_views, nshards, num_cus, d_pk, d_ok, mq, k, d_lo, d_up, counts_only, st, &ex)) != RSBWT_OK) return rc;
        for (uint32_t s = 0; s < nshards; ++s) {
            HIP_OK(hipMemcpyAsync(lower + s * Q + q0, d_lo + (size_t)s * mq * 8, mq * 8, hipMemcpyDeviceToHost, st));
            if (!counts_only) HIP_OK(hipMemcpyAsync(upper + s * Q + q0, d_up + (size_t)s * mq * 8, mq * 8, hipMemcpyDeviceToHost, st));
        }
        HIP_OK(hipStreamSynchronize(st));  // (rel and the staging buffer are used again by the next slice)
    }
    return RSBWT_OK;
}
}  // namespace rsb

static int search_host(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                       uint64_t *lower, uint64_t *upper, bool counts_only) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (Q == 0) return RSBWT_OK;
    if (!kmers || !lower || (!counts_only && !upper)) return fail(RSBWT_EINVAL, "null argument");
    if (stride < k) return fail(RSBWT_EINVAL, "stride %zu < k %u", stride, k);
    int rc = use_device(h->device);
    if (rc) return rc;
    if (k == 0) {  // empty k-mer: empty interval
        for (size_t q = 0; q < Q; ++q) {
            if (counts_only) lower[q] = 0;
            else { lower[q] = 1; upper[q] = 0; }
        }
        return RSBWT_OK;
    }
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    return search_host_views(*h, h->pool, h->d_view, 1, h->num_cus, kmers, Q, k, stride, lower, upper, counts_only, view_is_narrow(h->view, k));
}

extern "C" {

int rsbwt_find_intervals_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                             void *d_lower, void *d_upper, void *stream) {
    return search_dev(h, d_packed, d_valid, Q, k, d_lower, d_upper, false, (hipStream_t)stream);
}

int rsbwt_count_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                    void *d_counts, void *stream) {
    return search_dev(h, d_packed, d_valid, Q, k, d_counts, nullptr, true, (hipStream_t)stream);
}

int rsbwt_find_interval_pairs_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                                  void *d_pairs, void *stream) {
    search_extra ex;
    ex.pairs = true;
    return search_dev(h, d_packed, d_valid, Q, k, d_pairs, nullptr, false, (hipStream_t)stream, &ex);
}

size_t rsbwt_packed_pairs_bytes(size_t n) { return (n * 10 + 3) / 4 * 4; }

int rsbwt_pack_interval_pairs_dev(const void *d_pairs, size_t n, void *d_packed, void *d_unfit, int device, void *stream) {
    if (n == 0) return RSBWT_OK;
    if (!d_pairs || !d_packed) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(device);
    if (rc) return rc;
    const hipError_t e = launch_pack_pairs10(d_pairs, n, d_packed, d_unfit, (hipStream_t)stream);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "pack kernel launch");
}

int rsbwt_unpack_interval_pairs_dev(const void *d_packed, size_t n, void *d_pairs, int device, void *stream) {
    if (n == 0) return RSBWT_OK;
    if (!d_pairs || !d_packed) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(device);
    if (rc) return rc;
    const hipError_t e = launch_unpack_pairs10(d_packed, n, d_pairs, (hipStream_t)stream);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "unpack kernel launch");
}

int rsbwt_pack_reads_dev(const void *d_reads, const void *d_len, size_t n, uint32_t stride, void *d_packed, int device, void *stream) {
    if (n == 0) return RSBWT_OK;
    if (!d_reads || !d_len || !d_packed) return fail(RSBWT_EINVAL, "null argument");
    if (stride == 0 || stride % 16u) return fail(RSBWT_EINVAL, "stride %u: a multiple of 16 bytes", stride);
    int rc = use_device(device);
    if (rc) return rc;
    const hipError_t e = launch_pack_reads2(d_reads, d_len, n, stride, d_packed, (hipStream_t)stream);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "read packing kernel launch");
}

int rsbwt_unpack_reads_dev(const void *d_packed, const void *d_len, size_t n, uint32_t stride, void *d_reads, int device, void *stream) {
    if (n == 0) return RSBWT_OK;
    if (!d_reads || !d_len || !d_packed) return fail(RSBWT_EINVAL, "null argument");
    if (stride == 0 || stride % 16u) return fail(RSBWT_EINVAL, "stride %u: a multiple of 16 bytes", stride);
    int rc = use_device(device);
    if (rc) return rc;
    const hipError_t e = launch_unpack_reads2(d_packed, d_len, n, stride, d_reads, (hipStream_t)stream);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "read unpacking kernel launch");
}

int rsbwt_find_intervals(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                         uint64_t *lower, uint64_t *upper) {
    return search_host(h, kmers, Q, k, stride, lower, upper, false);
}

int rsbwt_count(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *counts) {
    return search_host(h, kmers, Q, k, stride, counts, nullptr, true);
}

// ---- 1-mismatch search ------------------------------------------------------------------------

// The [m][3k+1] variant intervals of m packed k-mers (variants_kernel's order) into d_lo/d_up.
// With a k-mer table that does not cover the whole k-mer, the k-mers themselves are searched first
// with a trace, and every variant whose substituted position lies left of the table's reach resumes
// from its k-mer's interval at that position instead of being searched from scratch (it shares that
// whole suffix).  `scratch` holds the trace and the k-mers' own results: variants_scratch_bytes().
static size_t variants_scratch_bytes(const rsbwt_t *h, size_t m, uint32_t k) {
    const uint32_t tn = trace_entries(h->view, k);
    return tn ? m * (size_t)tn * 16 + 2 * m * 8 : 0;
}

// d_hit_bits != nullptr: instead of the [m][3k+1] matrices, only the variants that occur are written:
// {lower, upper} at d_lo[q * (3k+1) + v] and their bit in the map (search_extra::d_hit_bits).
static int search_variants(rsbwt_t *h, const void *d_pk, const void *d_ok, size_t m, uint32_t k, const void *d_vpk,
                           const void *d_vok, void *d_lo, void *d_up, uint8_t *scratch, hipStream_t stream,
                           void *d_hit_bits = nullptr) {
    const size_t V = 3 * (size_t)k + 1;
    const uint32_t tn = trace_entries(h->view, k);
    if (tn == 0) {
        search_extra plain;
        plain.d_hit_bits = d_hit_bits;
        return search_dev(h, d_vpk, d_vok, m * V, k, d_lo, d_up, false, stream, &plain);
    }
    uint8_t *d_trace = scratch, *d_olo = d_trace + m * (size_t)tn * 16, *d_oup = d_olo + m * 8;
    search_extra traced;
    traced.d_trace_out = d_trace;
    traced.trace_n = tn;
    int rc = search_dev(h, d_pk, d_ok, m, k, d_olo, d_oup, false, stream, &traced);
    if (rc) return rc;
    search_extra resumed;
    resumed.d_trace_in = d_trace;
    resumed.trace_n = tn;
    resumed.variants = (uint32_t)V;
    resumed.d_hit_bits = d_hit_bits;
    return search_dev(h, d_vpk, d_vok, m * V, k, d_lo, d_up, false, stream, &resumed);
}

int rsbwt_find_intervals_1mm(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                             uint64_t *lower, uint64_t *upper) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (Q == 0) return RSBWT_OK;
    if (!kmers || !lower || !upper) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0 || stride < k) return fail(RSBWT_EINVAL, "bad k/stride");
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    const uint32_t wpq = words_per_kmer(k);
    const size_t V = 3 * (size_t)k + 1;
    const size_t SLICE = std::max<size_t>(1, (16u << 20) / V);  // ~16M variants per pass: below a few million a launch does not fill the GPU
    ctx_guard g(h->pool);
    if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
    hipStream_t st = g.c->st[0];
    for (size_t q0 = 0; q0 < Q; q0 += SLICE) {
        const size_t m = std::min(SLICE, Q - q0), mv = m * V;
        const size_t ascii_bytes = (m - 1) * stride + k;
        const size_t a_ascii = (ascii_bytes + 15) & ~(size_t)15, a_pk = (m * wpq * 8 + 15) & ~(size_t)15, a_ok = (m + 15) & ~(size_t)15;
        const size_t a_vpk = (mv * wpq * 8 + 15) & ~(size_t)15, a_vok = (mv + 15) & ~(size_t)15;
        const size_t a_scr = (variants_scratch_bytes(h, m, k) + 15) & ~(size_t)15;
        if ((rc = g.c->stage(a_ascii + a_pk + a_ok + a_vpk + a_vok + 2 * mv * 8 + a_scr)) != RSBWT_OK) return rc;
        uint8_t *d_ascii = (uint8_t *)g.c->d_stage, *d_pk = d_ascii + a_ascii, *d_ok = d_pk + a_pk;
        uint8_t *d_vpk = d_ok + a_ok, *d_vok = d_vpk + a_vpk, *d_lo = d_vok + a_vok, *d_up = d_lo + mv * 8;
        uint8_t *d_scr = d_up + mv * 8;
        HIP_OK(hipMemcpyAsync(d_ascii, kmers + q0 * stride, ascii_bytes, hipMemcpyHostToDevice, st));
        hipError_t e = launch_pack(d_ascii, m, k, stride, d_pk, d_ok, st);
        if (e == hipSuccess) e = launch_variants(d_pk, d_ok, m, k, d_vpk, d_vok, st);
        if (e != hipSuccess) return fail_hip(e, "variant kernel launch");
        rc = search_variants(h, d_pk, d_ok, m, k, d_vpk, d_vok, d_lo, d_up, d_scr, st);
        if (rc) return rc;
        HIP_OK(hipMemcpyAsync(lower + q0 * V, d_lo, mv * 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(upper + q0 * V, d_up, mv * 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    return RSBWT_OK;
}

// Device-resident form: m packed k-mers -> d_lower/d_upper [m][3k+1]; d_scratch holds the variants and
// the trace (rsbwt_1mm_scratch_bytes).  Nothing is synchronised.
size_t rsbwt_1mm_scratch_bytes(const rsbwt_t *h, size_t m, uint32_t k) {
    if (!h || k == 0) return 0;
    const size_t V = 3 * (size_t)k + 1, mv = m * V;
    // every part starts on a 16-byte boundary: the trace and the sparse results are read and written as 16-byte words
    return ((mv * words_per_kmer(k) * 8 + 15) & ~(size_t)15) + ((mv + 15) & ~(size_t)15) + ((variants_scratch_bytes(h, m, k) + 15) & ~(size_t)15);
}

int rsbwt_find_intervals_1mm_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k,
                                 void *d_lower, void *d_upper, void *d_scratch, void *stream) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (m == 0) return RSBWT_OK;
    if (!d_packed || !d_valid || !d_lower || !d_upper || !d_scratch) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0) return fail(RSBWT_EINVAL, "k must be at least 1");
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    const size_t V = 3 * (size_t)k + 1, mv = m * V;
    uint8_t *d_vpk = (uint8_t *)d_scratch, *d_vok = d_vpk + ((mv * words_per_kmer(k) * 8 + 15) & ~(size_t)15);
    uint8_t *d_scr = d_vok + ((mv + 15) & ~(size_t)15);
    hipError_t e = launch_variants(d_packed, d_valid, m, k, d_vpk, d_vok, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "variant kernel launch");
    return search_variants(h, d_packed, d_valid, m, k, d_vpk, d_vok, d_lower, d_upper, d_scr, (hipStream_t)stream);
}

// Scratch of a hit-list search of m k-mers: the variants and the trace (rsbwt_1mm_scratch_bytes), the
// sparse {lower, upper} array (16 B per variant, written only where a variant occurs), the hit map (one
// bit per variant) and the block sums of its compaction.
struct hits_layout {
    size_t variants, sparse, bits, blocks, total;
};
static hits_layout hits_scratch_layout(const rsbwt_t *h, size_t m, uint32_t k) {
    const size_t mv = m * (3 * (size_t)k + 1), nwords = (mv + 63) / 64;
    hits_layout L;
    L.variants = (rsbwt_1mm_scratch_bytes(h, m, k) + 15) & ~(size_t)15;
    L.sparse = mv * 16;
    L.bits = (nwords * 8 + 15) & ~(size_t)15;
    L.blocks = ((nwords + 255) / 256 + 2) * 8;
    L.total = L.variants + L.sparse + L.bits + ((L.blocks + 15) & ~(size_t)15);
    return L;
}

size_t rsbwt_hits_1mm_scratch_bytes(const rsbwt_t *h, size_t m, uint32_t k) {
    if (!h || k == 0) return 0;
    return hits_scratch_layout(h, m, k).total;
}

// the search half: variants -> sparse results + hit map (zeroed here)
// d_variants != nullptr: the variants of this batch were expanded already (variants_of_batch_dev, by a caller that
// searches them in several shards): they are read from there instead of being made again in this shard's scratch
static int hits_search(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k, uint8_t *scratch,
                       hipStream_t stream, const void *d_variants = nullptr) {
    const hits_layout L = hits_scratch_layout(h, m, k);
    const size_t V = 3 * (size_t)k + 1, mv = m * V;
    const size_t a_vpk = (mv * words_per_kmer(k) * 8 + 15) & ~(size_t)15;
    uint8_t *d_vpk = scratch, *d_vok = d_vpk + a_vpk, *d_scr = d_vok + ((mv + 15) & ~(size_t)15);
    uint8_t *d_sparse = scratch + L.variants, *d_bits = d_sparse + L.sparse;
    HIP_OK(hipMemsetAsync(d_bits, 0, L.bits, stream));
    if (d_variants) {
        d_vpk = (uint8_t *)d_variants;
        d_vok = d_vpk + a_vpk;
    } else {
        hipError_t e = launch_variants(d_packed, d_valid, m, k, d_vpk, d_vok, stream);
        if (e != hipSuccess) return fail_hip(e, "variant kernel launch");
    }
    return search_variants(h, d_packed, d_valid, m, k, d_vpk, d_vok, d_sparse, nullptr, d_scr, stream, d_bits);
}

}  // extern "C"

namespace rsb {
// The 3k+1 variants of m packed k-mers (packed words, then validity bytes), for hits_1mm_dev_shared: they depend on
// the batch alone, so a caller that searches several shards expands them once.
size_t variants_bytes(size_t m, uint32_t k) {
    const size_t mv = m * (3 * (size_t)k + 1);
    return ((mv * words_per_kmer(k) * 8 + 15) & ~(size_t)15) + ((mv + 15) & ~(size_t)15);
}
int variants_of_batch_dev(const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_variants, hipStream_t stream) {
    const size_t mv = m * (3 * (size_t)k + 1);
    const hipError_t e = launch_variants(d_packed, d_valid, m, k, d_variants, (uint8_t *)d_variants + ((mv * words_per_kmer(k) * 8 + 15) & ~(size_t)15), stream);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "variant kernel launch");
}
static int hits_1mm_dev_impl(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_hits, size_t cap,
                             void *d_total, void *d_scratch, void *stream, const void *d_variants);
int hits_1mm_dev_shared(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_hits, size_t cap,
                        void *d_total, void *d_scratch, void *stream, const void *d_variants) {
    return hits_1mm_dev_impl(h, d_packed, d_valid, m, k, d_hits, cap, d_total, d_scratch, stream, d_variants);
}
}  // namespace rsb

extern "C" {

int rsbwt_hits_1mm_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_hits,
                       size_t cap, void *d_total, void *d_scratch, void *stream) {
    return hits_1mm_dev_impl(h, d_packed, d_valid, m, k, d_hits, cap, d_total, d_scratch, stream, nullptr);
}

}  // extern "C"

static int rsb::hits_1mm_dev_impl(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_hits, size_t cap,
                                  void *d_total, void *d_scratch, void *stream, const void *d_variants) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (!d_total || (!d_hits && cap)) return fail(RSBWT_EINVAL, "null argument");
    if (m == 0) {
        HIP_OK(hipMemsetAsync(d_total, 0, 8, (hipStream_t)stream));
        return RSBWT_OK;
    }
    if (!d_packed || !d_valid || !d_scratch) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0) return fail(RSBWT_EINVAL, "k must be at least 1");
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    if ((rc = hits_search(h, d_packed, d_valid, m, k, (uint8_t *)d_scratch, (hipStream_t)stream, d_variants)) != RSBWT_OK) return rc;
    const hits_layout L = hits_scratch_layout(h, m, k);
    uint8_t *d_sparse = (uint8_t *)d_scratch + L.variants, *d_bits = d_sparse + L.sparse, *d_blocks = d_bits + L.bits;
    const hipError_t e = launch_compact_hits(d_bits, d_sparse, m * (3 * (size_t)k + 1), d_hits, cap, d_total, d_blocks, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "hit list kernels");
    return RSBWT_OK;
}

extern "C" {

int rsbwt_hits_1mm(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride, rsbwt_hit_1mm *hits,
                   size_t cap, size_t *nhits) {
    if (!h || !nhits) return fail(RSBWT_EINVAL, "null argument");
    *nhits = 0;
    if (Q == 0) return RSBWT_OK;
    if (!kmers || (!hits && cap)) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0 || stride < k) return fail(RSBWT_EINVAL, "bad k/stride");
    if (k > 32767u) return fail(RSBWT_ERANGE, "k %u: positions are reported as int16", k);
    if (Q > 0xFFFFFFFFull) return fail(RSBWT_ERANGE, "at most 2^32 - 1 k-mers per call");
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    const uint32_t wpq = words_per_kmer(k);
    const size_t V = 3 * (size_t)k + 1;
    const size_t SLICE = std::max<size_t>(1, (16u << 20) / V);  // ~16M variants per pass: below a few million a launch does not fill the GPU
    ctx_guard g(h->pool);
    if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
    hipStream_t st = g.c->st[0];
    // The variants that occur leave the search kernel as sparse stores + a hit map and are put in order on
    // the device (launch_compact_hits): nothing is written or read back for the ~99 % that end empty.
    struct hit_rec { uint64_t lower, upper, index, zero; };
    std::vector<hit_rec> got;
    size_t total = 0;
    bool overflow = false;
    try {
        for (size_t q0 = 0; q0 < Q; q0 += SLICE) {
            const size_t m = std::min(SLICE, Q - q0), mv = m * V;
            const size_t ascii_bytes = (m - 1) * stride + k;
            const size_t a_ascii = (ascii_bytes + 15) & ~(size_t)15, a_pk = (m * wpq * 8 + 15) & ~(size_t)15, a_ok = (m + 15) & ~(size_t)15;
            const hits_layout L = hits_scratch_layout(h, m, k);
            size_t room = std::min(mv, std::max<size_t>(4 * m, 1u << 14));  // records this slice may leave; more on demand
            if ((rc = g.c->stage(a_ascii + a_pk + a_ok + L.total + 16 + room * sizeof(hit_rec))) != RSBWT_OK) return rc;
            uint8_t *d_ascii = (uint8_t *)g.c->d_stage, *d_pk = d_ascii + a_ascii, *d_ok = d_pk + a_pk, *d_scr = d_ok + a_ok;
            uint8_t *d_total = d_scr + L.total, *d_list = d_total + 16;
            uint8_t *d_sparse = d_scr + L.variants, *d_bits = d_sparse + L.sparse, *d_blocks = d_bits + L.bits;
            HIP_OK(hipMemcpyAsync(d_ascii, kmers + q0 * stride, ascii_bytes, hipMemcpyHostToDevice, st));
            hipError_t e = launch_pack(d_ascii, m, k, stride, d_pk, d_ok, st);
            if (e != hipSuccess) return fail_hip(e, "pack kernel launch");
            if ((rc = hits_search(h, d_pk, d_ok, m, k, d_scr, st)) != RSBWT_OK) return rc;
            uint64_t count = 0;
            for (;;) {
                e = launch_compact_hits(d_bits, d_sparse, mv, d_list, room, d_total, d_blocks, st);
                if (e != hipSuccess) return fail_hip(e, "hit list kernels");
                HIP_OK(hipMemcpyAsync(&count, d_total, 8, hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
                if (count <= room || overflow || total + count > cap) break;
                // more hits than the list had room for: the search's results are still in place, only the
                // list is made again, in a buffer of its own (growing the staging buffer would move them)
                room = (size_t)count;
                void *bigger = nullptr;
                HIP_OK(hipMalloc(&bigger, room * sizeof(hit_rec)));
                e = launch_compact_hits(d_bits, d_sparse, mv, bigger, room, d_total, d_blocks, st);
                hipError_t e2 = hipSuccess;
                if (e == hipSuccess) {
                    got.resize((size_t)count);
                    e2 = hipMemcpyAsync(got.data(), bigger, (size_t)count * sizeof(hit_rec), hipMemcpyDeviceToHost, st);
                    if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
                }
                (void)hipFree(bigger);
                if (e != hipSuccess) return fail_hip(e, "hit list kernels");
                if (e2 != hipSuccess) return fail_hip(e2, "hit list copy");
                d_list = nullptr;  // `got` is filled
                break;
            }
            if (!overflow && total + count <= cap && count) {
                if (d_list) {
                    got.resize((size_t)count);
                    HIP_OK(hipMemcpyAsync(got.data(), d_list, (size_t)count * sizeof(hit_rec), hipMemcpyDeviceToHost, st));
                    HIP_OK(hipStreamSynchronize(st));
                }
                for (size_t i = 0; i < got.size(); ++i) {  // already in (k-mer, position, base) order
                    const size_t q = (size_t)(got[i].index / V);
                    const uint32_t v = (uint32_t)(got[i].index % V);
                    rsbwt_hit_1mm &r = hits[total + i];
                    r.lower = got[i].lower;
                    r.upper = got[i].upper;
                    r.query = (uint32_t)(q0 + q);
                    r.reserved = 0;
                    if (v == 0u) {
                        r.pos = -1;
                        r.base = 0;
                    } else {
                        const uint32_t pos = (v - 1u) / 3u, d = (v - 1u) % 3u;
                        const int orig = rank_of(kmers[(q0 + q) * stride + pos]) - 1;  // 0..3: a hit's k-mer is all ACGT
                        r.pos = (int16_t)pos;
                        r.base = "ACGT"[(int)d < orig ? d : d + 1u];
                    }
                }
            } else if (total + count > cap) {
                overflow = true;  // keep counting so that the caller learns the size it needs
            }
            total += (size_t)count;
        }
    } catch (const std::bad_alloc &) {
        return fail(RSBWT_ENOMEM, "host allocation failed");
    }
    *nhits = total;
    if (overflow) return fail(RSBWT_ERANGE, "%zu hits, room for %zu", total, cap);
    return RSBWT_OK;
}

// ---- read extraction --------------------------------------------------------------------------

}  // extern "C"

// rows in host memory -> reads in HBM (d_out: n x stride, d_len / d_plen: n x u32), slice by slice
// through `fn(i0, m, d_out, d_len, d_plen)`, which consumes a slice before the next one overwrites it
template <class F>
static int extract_slices(rsbwt_t *h, call_ctx *c, const uint64_t *rows, size_t n, uint32_t stride, size_t extra_bytes,
                          F &&fn) {
    hipStream_t st = c->st[0];
    int rc = ensure_select_samples(h, st);
    if (rc) return rc;
    const size_t SLICE = 1u << 20;
    for (size_t i0 = 0; i0 < n; i0 += SLICE) {
        const size_t m = std::min(SLICE, n - i0);
        const size_t a_rows = m * 8, a_out = (m * (size_t)stride + 15) & ~(size_t)15, a_len = (m * 4 + 15) & ~(size_t)15;
        if ((rc = c->stage(a_rows + a_out + 2 * a_len + extra_bytes)) != RSBWT_OK) return rc;
        uint8_t *base = (uint8_t *)c->d_stage;
        uint8_t *d_rows = base, *d_out = base + a_rows, *d_pl = d_out + a_out, *d_len = d_pl + a_len;
        HIP_OK(hipMemcpyAsync(d_rows, rows + i0, a_rows, hipMemcpyHostToDevice, st));
        hipError_t e = launch_extract_wave(h->scratch, h->d_xview, 1, d_rows, m, d_out, stride, d_pl, d_len, h->num_cus, st);
        if (e != hipSuccess) return fail_hip(e, "extract kernel launch");
        if ((rc = fn(i0, m, d_out, d_len, d_pl, d_len + a_len)) != RSBWT_OK) return rc;
    }
    return RSBWT_OK;
}

extern "C" {

int rsbwt_extract(rsbwt_t *h, const uint64_t *rows, size_t n, char *out, uint32_t stride, uint32_t *len,
                  uint32_t *prefix_len) {
    if (!h || (!rows && n) || (!out && n)) return fail(RSBWT_EINVAL, "null argument");
    if (n == 0) return RSBWT_OK;
    if (stride == 0) return fail(RSBWT_EINVAL, "stride must be positive");
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    ctx_guard g(h->pool);
    if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
    hipStream_t st = g.c->st[0];
    return extract_slices(h, g.c, rows, n, stride, 0,
                          [&](size_t i0, size_t m, uint8_t *d_out, uint8_t *d_len, uint8_t *d_pl, uint8_t *) -> int {
                              HIP_OK(hipMemcpyAsync(out + i0 * (size_t)stride, d_out, m * (size_t)stride, hipMemcpyDeviceToHost, st));
                              if (len) HIP_OK(hipMemcpyAsync(len + i0, d_len, m * 4, hipMemcpyDeviceToHost, st));
                              if (prefix_len) HIP_OK(hipMemcpyAsync(prefix_len + i0, d_pl, m * 4, hipMemcpyDeviceToHost, st));
                              HIP_OK(hipStreamSynchronize(st));
                              return RSBWT_OK;
                          });
}

// Device-resident form: d_rows [n] u64 -> d_out [n][stride] bytes, d_len / d_prefix_len [n] u32 (both
// required), on `stream`; nothing is synchronised (without RSBWT_OPEN_READS the select sample table is built on
// first use).
int rsbwt_extract_dev(rsbwt_t *h, const void *d_rows, size_t n, void *d_out, uint32_t stride, void *d_len,
                      void *d_prefix_len, void *stream) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (n == 0) return RSBWT_OK;
    if (!d_rows || !d_out || !d_len || !d_prefix_len) return fail(RSBWT_EINVAL, "null argument");
    if (stride == 0) return fail(RSBWT_EINVAL, "stride must be positive");
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    if ((rc = ensure_select_samples(h, (hipStream_t)stream)) != RSBWT_OK) return rc;
    unsigned long long *work = nullptr;
    if (h->counting) {  // rsbwt_set_counting: the walk kernels' counters, read with rsbwt_last_search_counters
        work = h->d_work;
        HIP_OK(hipMemsetAsync(work, 0, WORK_WORDS * sizeof(unsigned long long), (hipStream_t)stream));
    }
    hipError_t e = launch_extract_wave(h->scratch, h->d_xview, 1, d_rows, n, d_out, stride, d_prefix_len, d_len, h->num_cus,
                                       (hipStream_t)stream, work);
    if (e != hipSuccess) return fail_hip(e, "extract kernel launch");
    return RSBWT_OK;
}

// ---- query / query_exactmatch (src/bwt/query.cpp:87-120) ------------------------------------------

// rows of the intervals of a batch: row list + the k-mer each belongs to.  An interval that is not a
// proper row range (lower > upper, or the reference's (0, 2^64-1) corner, which it would walk off
// the BWT with) contributes no row.
static void interval_rows(const uint64_t *lo, const uint64_t *up, size_t Q, uint64_t n, std::vector<uint64_t> *rows,
                          std::vector<uint32_t> *owner, uint64_t *first) {
    uint64_t total = 0;
    for (size_t q = 0; q < Q; ++q) {
        first[q] = total;
        if (lo[q] <= up[q] && up[q] < n) total += up[q] - lo[q] + 1;
    }
    first[Q] = total;
    if (!rows) return;
    rows->resize(total);
    owner->resize(total);
    for (size_t q = 0; q < Q; ++q) {
        const uint64_t c = first[q + 1] - first[q];
        for (uint64_t i = 0; i < c; ++i) {
            (*rows)[first[q] + i] = lo[q] + i;
            (*owner)[first[q] + i] = (uint32_t)q;
        }
    }
}

int rsbwt_query_exactmatch(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride, uint8_t *found) {
    if (!h || (!kmers && Q) || (!found && Q)) return fail(RSBWT_EINVAL, "null argument");
    if (Q == 0) return RSBWT_OK;
    if (Q > 0xFFFFFFFFull) return fail(RSBWT_ERANGE, "at most 2^32 - 1 strings per call");
    memset(found, 0, Q);
    if (k == 0) return RSBWT_OK;
    try {
        std::vector<uint64_t> lo(Q), up(Q), first(Q + 1), rows;
        std::vector<uint32_t> owner;
        int rc = rsbwt_find_intervals(h, kmers, Q, k, stride, lo.data(), up.data());
        if (rc) return rc;
        interval_rows(lo.data(), up.data(), Q, h->view.n, &rows, &owner, first.data());
        if (rows.empty()) return RSBWT_OK;
        ctx_guard g(h->pool);
        if (!g.c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
        hipStream_t st = g.c->st[0];
        // the k-mers once, next to the staging buffer (they are compared with every extracted read)
        void *d_km = nullptr;
        const size_t km_bytes = (Q - 1) * stride + k;
        HIP_OK(hipMalloc(&d_km, km_bytes));
        hipError_t e = hipMemcpyAsync(d_km, kmers, km_bytes, hipMemcpyHostToDevice, st);
        std::vector<uint8_t> flags(std::min<size_t>(rows.size(), 1u << 20));
        const uint32_t rstride = ((k + 1u) + 15u) & ~15u;  // a read longer than k cannot equal the query
        if (e == hipSuccess)
            rc = extract_slices(h, g.c, rows.data(), rows.size(), rstride, (1u << 20) * 5,
                                [&](size_t i0, size_t m, uint8_t *d_out, uint8_t *d_len, uint8_t *, uint8_t *d_extra) -> int {
                                    uint8_t *d_owner = d_extra, *d_flags = d_extra + (1u << 20) * 4;
                                    HIP_OK(hipMemcpyAsync(d_owner, owner.data() + i0, m * 4, hipMemcpyHostToDevice, st));
                                    hipError_t e2 = launch_match_reads(d_out, d_len, m, rstride, d_owner, d_km, k, stride, d_flags, st);
                                    if (e2 != hipSuccess) return fail_hip(e2, "match kernel launch");
                                    HIP_OK(hipMemcpyAsync(flags.data(), d_flags, m, hipMemcpyDeviceToHost, st));
                                    HIP_OK(hipStreamSynchronize(st));
                                    for (size_t i = 0; i < m; ++i)
                                        if (flags[i]) found[owner[i0 + i]] = 1;
                                    return RSBWT_OK;
                                });
        else rc = fail_hip(e, "hipMemcpyAsync(kmers)");
        (void)hipStreamSynchronize(st);
        (void)hipFree(d_km);
        return rc;
    } catch (const std::bad_alloc &) {
        return fail(RSBWT_ENOMEM, "host allocation failed");
    }
}

int rsbwt_query(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *first, char *reads,
                uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads) {
    if (!h || (!kmers && Q) || !first || !nreads) return fail(RSBWT_EINVAL, "null argument");
    *nreads = 0;
    first[0] = 0;
    if (Q == 0) return RSBWT_OK;
    if (read_stride == 0) return fail(RSBWT_EINVAL, "read_stride must be positive");
    try {
        std::vector<uint64_t> lo(Q), up(Q), rows;
        std::vector<uint32_t> owner;
        if (k == 0) {
            for (size_t q = 0; q <= Q; ++q) first[q] = 0;
            return RSBWT_OK;
        }
        int rc = rsbwt_find_intervals(h, kmers, Q, k, stride, lo.data(), up.data());
        if (rc) return rc;
        interval_rows(lo.data(), up.data(), Q, h->view.n, nullptr, nullptr, first);
        *nreads = (size_t)first[Q];
        if (first[Q] > cap_reads) return fail(RSBWT_ERANGE, "%llu reads, room for %zu", (unsigned long long)first[Q], cap_reads);
        if (first[Q] == 0) return RSBWT_OK;
        if (!reads || !read_len) return fail(RSBWT_EINVAL, "null argument");
        interval_rows(lo.data(), up.data(), Q, h->view.n, &rows, &owner, first);
        return rsbwt_extract(h, rows.data(), rows.size(), reads, read_stride, read_len, nullptr);
    } catch (const std::bad_alloc &) {
        return fail(RSBWT_ENOMEM, "host allocation failed");
    }
}

// ---- measurement ------------------------------------------------------------------------------

}  // extern "C"

namespace rsb {

int meter_history_ms(search_meter &m, float *ms, size_t cap, size_t *count) {
    std::lock_guard<std::mutex> lock(m.mu);
    size_t n = (size_t)std::min<uint64_t>(m.launches, search_meter::RING);
    if (n > cap) n = cap;
    for (size_t i = 0; i < n; ++i) {  // oldest of the n first
        const int slot = (int)((m.launches - n + i) % search_meter::RING);
        HIP_OK(hipEventSynchronize(m.ev_stop[slot]));
        HIP_OK(hipEventElapsedTime(&ms[i], m.ev_start[slot], m.ev_stop[slot]));
    }
    *count = n;
    return RSBWT_OK;
}

int meter_work(search_meter &m, uint64_t *words, size_t nwords) {
    std::lock_guard<std::mutex> lock(m.mu);
    // (the walk kernels of an extraction leave their counters here too, without a search launch: the copy
    // below runs on the null stream, which waits for whatever the caller's streams have enqueued)
    if (m.launches) HIP_OK(hipEventSynchronize(m.ev_stop[(m.launches - 1) % search_meter::RING]));
    else HIP_OK(hipDeviceSynchronize());
    unsigned long long w[WORK_WORDS];
    HIP_OK(hipMemcpy(w, m.d_work, sizeof w, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nwords && i < (size_t)WORK_WORDS; ++i) words[i] = w[i];
    return RSBWT_OK;
}

}  // namespace rsb

extern "C" {

int rsbwt_last_search_ms(rsbwt_t *h, float *ms) {
    if (!h || !ms) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    size_t cnt = 0;
    rc = meter_history_ms(*h, ms, 1, &cnt);
    if (rc == RSBWT_OK && cnt == 0) return fail(RSBWT_EINVAL, "no search has been launched through this handle");
    return rc;
}

int rsbwt_search_history_ms(rsbwt_t *h, float *ms, size_t cap, size_t *count) {
    if (!h || (!ms && cap) || !count) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    return meter_history_ms(*h, ms, cap, count);
}

int rsbwt_set_counting(rsbwt_t *h, int on) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    std::lock_guard<std::mutex> lock(h->mu);
    h->counting = on != 0;
    return RSBWT_OK;
}

int rsbwt_last_search_work(rsbwt_t *h, uint64_t *lf_steps, uint64_t *occ_lookups, uint64_t *line_reads) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    int rc = use_device(h->device);
    if (rc) return rc;
    uint64_t w[WORK_WORDS];
    if ((rc = meter_work(*h, w, WORK_WORDS)) != RSBWT_OK) return rc;
    if (lf_steps) *lf_steps = w[0];
    if (occ_lookups) *occ_lookups = w[1];
    if (line_reads) *line_reads = w[2];
    return RSBWT_OK;
}

int rsbwt_last_search_counters(rsbwt_t *h, uint64_t *words16) {
    if (!h || !words16) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    return meter_work(*h, words16, WORK_WORDS);
}

int rsbwt_last_search_phases(rsbwt_t *h, uint64_t *cycles, uint64_t *passes) {
    if (!h || !cycles || !passes) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    uint64_t w[WORK_WORDS];
    if ((rc = meter_work(*h, w, WORK_WORDS)) != RSBWT_OK) return rc;
    for (int i = 0; i < 6; ++i) cycles[i] = w[4 + i];
    *passes = w[10];
    return RSBWT_OK;
}

int rsbwt_last_search_ktab_lookups(rsbwt_t *h, uint64_t *lookups) {
    if (!h || !lookups) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    uint64_t w[WORK_WORDS];
    if ((rc = meter_work(*h, w, WORK_WORDS)) != RSBWT_OK) return rc;
    *lookups = w[3];
    return RSBWT_OK;
}

// ---- synthetic data ---------------------------------------------------------------------------

int rsbwt_synth_runs_dev(void *d_runs, uint64_t num_runs, uint64_t seed, int device, void *stream) {
    if (!d_runs && num_runs) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(device);
    if (rc) return rc;
    hipError_t e = launch_synth_runs(d_runs, num_runs, seed, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "synth kernel launch");
    return RSBWT_OK;
}

int rsbwt_synth_runs_dev_at(void *d_runs, uint64_t first, uint64_t num_runs, uint64_t seed, int device, void *stream) {
    if (!d_runs && num_runs) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(device);
    if (rc) return rc;
    hipError_t e = launch_synth_runs(d_runs, num_runs, seed, (hipStream_t)stream, first);
    if (e != hipSuccess) return fail_hip(e, "synth kernel launch");
    return RSBWT_OK;
}

int rsbwt_sample_present_kmers_dev(rsbwt_t *h, size_t Q, uint32_t k, size_t stride, uint64_t seed,
                                   void *d_kmers, void *stream) {
    if (!h || (!d_kmers && Q)) return fail(RSBWT_EINVAL, "null argument");
    if (stride < k || k == 0) return fail(RSBWT_EINVAL, "bad k/stride");
    int rc = use_device(h->device);
    if (rc) return rc;
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    hipError_t e = launch_sample_present(h->view, Q, k, stride, seed, d_kmers, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "sample kernel launch");
    return RSBWT_OK;
}

}  // extern "C"

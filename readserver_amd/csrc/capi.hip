// capi.hip -- the C-ABI of librsbwt.so (include/rsbwt.h) over the HIP engine.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/rsbwt.h"
#include "block_format.h"
#include "bpi2.h"
#include "bwt_file.h"
#include "kernels.h"

using namespace rsb;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int fail_hip(hipError_t e, const char *what) {
    return fail(RSBWT_EHIP, "%s: %s", what, hipGetErrorString(e));
}

#define HIP_OK(x)                                              \
    do {                                                       \
        hipError_t _e = (x);                                   \
        if (_e != hipSuccess) return fail_hip(_e, #x);         \
    } while (0)

int use_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RSBWT_ENODEV, "no HIP device is visible: the popBWT engine has no CPU fallback");
    if (device < 0 || device >= n) return fail(RSBWT_ENODEV, "device %d out of range (0..%d)", device, n - 1);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    return RSBWT_OK;
}

inline uint32_t words_per_kmer(uint32_t k) { return k ? (k + 31u) / 32u : 1u; }

}  // namespace

struct rsbwt {
    int device = 0;
    int num_cus = 256;
    rsbwt_view view;
    slot_view slots = {};
    uint64_t num_runs = 0, num_strings = 0, hbm_bytes = 0;
    hipStream_t stream = nullptr;  // host-buffer calls run here
    hipStream_t stream2 = nullptr; // ... and here: consecutive slices of a big host batch alternate
    static constexpr int RING = 64;  // HIP-event pairs of the most recent search launches
    hipEvent_t ev_start[RING] = {}, ev_stop[RING] = {};
    uint64_t launches = 0;  // search launches so far; launch i uses pair i % RING
    bool counting = false;
    unsigned long long *d_work = nullptr;  // 4 counters: LF steps, Occ lookups, block reads, k-table lookups
    std::recursive_mutex mu;
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    uint32_t *d_sel = nullptr;  // sampled select table, built on the first extraction

    int stage(size_t bytes) {
        if (bytes <= stage_bytes) return RSBWT_OK;
        if (d_stage) (void)hipFree(d_stage);
        d_stage = nullptr;
        stage_bytes = 0;
        hipError_t e = hipMalloc(&d_stage, bytes);
        if (e != hipSuccess) return fail(RSBWT_ENOMEM, "hipMalloc(%zu) for staging: %s", bytes, hipGetErrorString(e));
        stage_bytes = bytes;
        return RSBWT_OK;
    }
};

struct rsbwt_set {
    std::vector<rsbwt_t *> shards;
    bool owns = false;
};

extern "C" {

const char *rsbwt_version(void) { return "rsbwt 0.1 (gfx950)"; }

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
    default: return "unknown error";
    }
}

// ---- lifetime -------------------------------------------------------------------------------

static int finish_open(const void *d_runs, uint64_t num_runs, uint64_t num_strings, int device,
                       uint32_t flags, rsbwt_t **out) {
    rsbwt_t *h = new (std::nothrow) rsbwt();
    if (!h) return fail(RSBWT_ENOMEM, "host allocation failed");
    h->device = device;
    h->num_strings = num_strings;
    memset(&h->view, 0, sizeof h->view);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        h->num_cus = prop.multiProcessorCount;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc(&h->d_work, 16 * sizeof(unsigned long long))) != hipSuccess) {
        rsbwt_close(h);
        return fail_hip(e, "creating stream/events");
    }
    for (int i = 0; i < rsbwt::RING; ++i) {
        if ((e = hipEventCreate(&h->ev_start[i])) != hipSuccess ||
            (e = hipEventCreate(&h->ev_stop[i])) != hipSuccess) {
            rsbwt_close(h);
            return fail_hip(e, "hipEventCreate");
        }
    }
    build_result br;
    int range_error = 0;
    e = build_device_index(d_runs, num_runs, flags & RSBWT_DIR_SHIFT_MASK, h->stream, &br, &range_error);
    if (e != hipSuccess) {
        rsbwt_close(h);
        if (e == hipErrorOutOfMemory) return fail(RSBWT_ENOMEM, "HBM allocation failed while building the index");
        return fail_hip(e, "build_device_index");
    }
    if (range_error) {
        rsbwt_close(h);
        return fail(RSBWT_ERANGE, "shard too large: at most 2^40 symbols and 2^32 blocks");
    }
    h->view = br.view;
    h->num_runs = br.num_runs;
    h->hbm_bytes = br.hbm_bytes;
    // single-request search layout: on request, or (auto) when it is affordable and fits
    {
        const uint32_t mode = flags & RSBWT_SLOTS_MASK;
        const uint32_t want_S = (flags & RSBWT_SLOT_SPAN_MASK) >> RSBWT_SLOT_SPAN_SHIFT;
        bool build = mode == RSBWT_SLOTS_ON;
        slot_params sp;
        if (mode == RSBWT_SLOTS_AUTO && h->view.n > 0 && choose_slot_span(h->view.n, h->num_runs, want_S, &sp)) {
            size_t free_b = 0, total_b = 0;
            // the builder may shrink the span by a step or two (slots.hip): allow for 1.4x the
            // starting estimate.  Index + slots within 45 % of the device leaves room for the k-mer
            // table and the batch buffers with one shard per GPU.
            const uint64_t est = sp.nslots * RSBWT_BLOCK_BYTES * 7 / 5;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                build = (h->hbm_bytes + est) <= total_b * 45 / 100 && est <= free_b / 2;
        }
        if (build && h->view.n > 0) {
            uint64_t bytes = 0;
            int rerr = 0;
            e = build_slots(h->view, h->num_runs, want_S, h->stream, &h->slots, &bytes, &rerr);
            if (e != hipSuccess || rerr) {
                h->slots = slot_view{};
                (void)hipGetLastError();
                if (mode == RSBWT_SLOTS_ON) {
                    rsbwt_close(h);
                    if (rerr) return fail(RSBWT_ERANGE, "slot layout does not fit this shard (2^32 blocks, span < 4096)");
                    if (e == hipErrorOutOfMemory) return fail(RSBWT_ENOMEM, "HBM allocation failed while building the slot layout");
                    return fail_hip(e, "build_slots");
                }
            } else {
                h->hbm_bytes += bytes;
            }
        }
    }
    // k-mer table: explicit depth, none, or auto = the deepest whose 8-byte entries take no more
    // HBM than the index itself and no more than a quarter of what is still free (HBM is there
    // to be used: every level replaces one LF step, two Occ lookups, of each query by the same
    // single 8-byte read), and whose T-mers still have ~1 expected occurrence (4^T <= n).
    // 8 B * 4^16 = 34 GB is the ceiling.
    uint32_t T = (flags & RSBWT_KTAB_MASK) >> RSBWT_KTAB_SHIFT;
    if (T == 31u || h->view.n == 0) T = 0;
    else if (T == 0u) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const uint64_t budget = std::min<uint64_t>(h->hbm_bytes, free_b / 4);
        T = 1;
        while (T < 16u && (8ull << (2u * (T + 1u))) <= budget && (1ull << (2u * (T + 1u))) <= h->view.n) ++T;
        if (T < 2u) T = 0;
    } else if (T < 2u) T = 2;
    if (T > 16u) T = 16;  // 8 B * 4^16 = 34 GB
    if (T) {
        uint64_t *d_tab = nullptr;
        const uint64_t bytes = 8ull << (2u * T);
        e = hipMalloc(&d_tab, bytes);
        if (e == hipSuccess) {
            e = build_ktable(h->view, &h->slots, T, d_tab, h->num_cus, h->stream);
            if (e != hipSuccess) (void)hipFree(d_tab);
        }
        if (e != hipSuccess) {
            rsbwt_close(h);
            if (e == hipErrorOutOfMemory) return fail(RSBWT_ENOMEM, "HBM allocation failed while building the k-mer table");
            return fail_hip(e, "build_ktable");
        }
        h->view.ktab = d_tab;
        h->view.ktab_depth = T;
        h->hbm_bytes += bytes;
    }
    *out = h;
    return RSBWT_OK;
}

int rsbwt_open_device_runs(const void *d_runs, uint64_t num_runs, uint64_t num_strings, int device,
                           uint32_t flags, rsbwt_t **out) {
    if (!out || (!d_runs && num_runs)) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    int rc = use_device(device);
    if (rc) return rc;
    return finish_open(d_runs, num_runs, num_strings, device, flags, out);
}

int rsbwt_open_runs(const uint8_t *runs, uint64_t num_runs, uint64_t num_strings, int device,
                    uint32_t flags, rsbwt_t **out) {
    if (!out || (!runs && num_runs)) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    int rc = use_device(device);
    if (rc) return rc;
    void *d_runs = nullptr;
    hipError_t e = hipMalloc(&d_runs, num_runs ? num_runs : 16);
    if (e != hipSuccess) return fail(RSBWT_ENOMEM, "hipMalloc(%llu) for run bytes: %s", (unsigned long long)num_runs, hipGetErrorString(e));
    e = hipMemcpy(d_runs, runs, num_runs, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d_runs); return fail_hip(e, "hipMemcpy(runs)"); }
    rc = finish_open(d_runs, num_runs, num_strings, device, flags, out);
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
    rc = use_device(device);
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
        rc = finish_open(d_runs, hdr.num_runs, hdr.num_strings, device, flags, out);
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
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->view.blocks) (void)hipFree((void *)h->view.blocks);
    if (h->view.dir) (void)hipFree((void *)h->view.dir);
    if (h->view.ktab) (void)hipFree((void *)h->view.ktab);
    if (h->slots.slots) (void)hipFree((void *)h->slots.slots);
    if (h->d_sel) (void)hipFree(h->d_sel);
    if (h->d_stage) (void)hipFree(h->d_stage);
    if (h->d_work) (void)hipFree(h->d_work);
    for (int i = 0; i < rsbwt::RING; ++i) {
        if (h->ev_start[i]) (void)hipEventDestroy(h->ev_start[i]);
        if (h->ev_stop[i]) (void)hipEventDestroy(h->ev_stop[i]);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
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
uint64_t rsbwt_num_blocks(const rsbwt_t *h) { return h->view.nblocks; }
uint32_t rsbwt_dir_shift(const rsbwt_t *h) { return h->view.dir_shift; }
uint32_t rsbwt_ktab_depth(const rsbwt_t *h) { return h->view.ktab_depth; }
uint32_t rsbwt_slot_span(const rsbwt_t *h) { return h->slots.slots ? h->slots.p.S : 0u; }
uint64_t rsbwt_slot_overflow_blocks(const rsbwt_t *h) { return h->slots.slots ? h->slots.noverflow : 0; }
uint64_t rsbwt_hbm_bytes(const rsbwt_t *h) { return h->hbm_bytes; }
int rsbwt_device(const rsbwt_t *h) { return h->device; }

// ---- class BWT mirrors ------------------------------------------------------------------------

// kind 0: occ(syms, vals)  1: char(vals)  2: occ_at(syms, vals)
static int mirror_batch(rsbwt_t *h, int kind, const char *syms, const uint64_t *vals, size_t n, void *out) {
    if (!h || (!vals && n) || (!out && n) || (kind != 1 && !syms && n)) return fail(RSBWT_EINVAL, "null argument");
    if (n == 0) return RSBWT_OK;
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    const size_t out_bytes = (kind == 1) ? n : n * 8;
    const size_t need = n * 8 + n + out_bytes + 64;
    if ((rc = h->stage(need)) != RSBWT_OK) return rc;
    uint8_t *base = (uint8_t *)h->d_stage;
    uint64_t *d_vals = (uint64_t *)base;
    uint8_t *d_out = base + n * 8;              // 8-aligned
    uint8_t *d_syms = d_out + ((out_bytes + 7) & ~(size_t)7);
    HIP_OK(hipMemcpyAsync(d_vals, vals, n * 8, hipMemcpyHostToDevice, h->stream));
    if (kind != 1) HIP_OK(hipMemcpyAsync(d_syms, syms, n, hipMemcpyHostToDevice, h->stream));
    hipError_t e = hipSuccess;
    if (kind == 0) e = launch_occ_batch(h->view, d_syms, d_vals, n, d_out, h->stream);
    else if (kind == 1) e = launch_char_batch(h->view, d_vals, n, d_out, h->stream);
    else e = launch_occ_at_batch(h->view, d_syms, d_vals, n, d_out, h->stream);
    if (e != hipSuccess) return fail_hip(e, "mirror kernel launch");
    HIP_OK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
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
    bpi2_index ix;
    std::string err;
    int rc = bpi2_from_bwt(bwt_path, &ix, &err);
    if (rc == RSBWT_OK) rc = bpi2_save(ix, bpi2_path, &err);
    if (rc != RSBWT_OK) return fail(rc, "%s", err.c_str());
    return RSBWT_OK;
}

int rsbwt_bpi2_check(rsbwt_t *h, const char *bpi2_path, uint64_t max_samples, uint64_t *checked,
                     uint64_t *mismatches) {
    if (!h || !bpi2_path || !checked || !mismatches) return fail(RSBWT_EINVAL, "null argument");
    *checked = *mismatches = 0;
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

static int search_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                      void *d_lower, void *d_upper, bool counts_only, hipStream_t stream,
                      const wave_search_extra *extra = nullptr) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (Q && (!d_packed || !d_valid || !d_lower || (!counts_only && !d_upper))) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    if (h->view.n == 0 && Q) return fail(RSBWT_EINVAL, "empty index");
    unsigned long long *work = nullptr;
    {
        std::lock_guard<std::recursive_mutex> lock(h->mu);
        if (h->counting) {
            work = h->d_work;
            HIP_OK(hipMemsetAsync(work, 0, 16 * sizeof(unsigned long long), stream));
        }
        const int slot = (int)(h->launches % rsbwt::RING);
        hipError_t e = launch_search(h->view, &h->slots, d_packed, d_valid, Q, k, d_lower, d_upper, counts_only, work, h->num_cus, stream, h->ev_start[slot], h->ev_stop[slot], extra);
        if (e != hipSuccess) return fail_hip(e, "search kernel launch");
        h->launches++;
    }
    return RSBWT_OK;
}

int rsbwt_find_intervals_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                             void *d_lower, void *d_upper, void *stream) {
    return search_dev(h, d_packed, d_valid, Q, k, d_lower, d_upper, false, (hipStream_t)stream);
}

int rsbwt_count_dev(rsbwt_t *h, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                    void *d_counts, void *stream) {
    return search_dev(h, d_packed, d_valid, Q, k, d_counts, nullptr, true, (hipStream_t)stream);
}

// host buffers: stage through HBM in slices of at most 4M k-mers
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
    const uint32_t wpq = words_per_kmer(k);
    // Slices of at most 2M k-mers alternate between two streams and two halves of the staging
    // buffer: while the host sits in slice i's copy back, the GPU already searches slice i + 1
    // (its k-mers went up before that copy was issued).
    const size_t SLICE = 2u << 20;
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    if (!h->stream2 && Q > SLICE) {
        hipError_t e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking);
        if (e != hipSuccess) return fail_hip(e, "hipStreamCreate");
    }
    const size_t m_max = std::min(SLICE, Q);
    const size_t a_ascii = (((m_max - 1) * stride + k) + 15) & ~(size_t)15;
    const size_t a_packed = m_max * wpq * 8;
    const size_t a_valid = (m_max + 15) & ~(size_t)15;
    const size_t half = a_ascii + a_packed + a_valid + 2 * m_max * 8;
    if ((rc = h->stage(Q > SLICE ? 2 * half : half)) != RSBWT_OK) return rc;
    struct slice_t {
        size_t q0 = 0, m = 0;
        uint8_t *d_lo = nullptr, *d_up = nullptr;
        hipStream_t st = nullptr;
    } prev;
    auto collect = [&](const slice_t &sl) -> int {  // results of a slice whose search is under way
        HIP_OK(hipMemcpyAsync(lower + sl.q0, sl.d_lo, sl.m * 8, hipMemcpyDeviceToHost, sl.st));
        if (!counts_only) HIP_OK(hipMemcpyAsync(upper + sl.q0, sl.d_up, sl.m * 8, hipMemcpyDeviceToHost, sl.st));
        HIP_OK(hipStreamSynchronize(sl.st));
        return RSBWT_OK;
    };
    size_t i = 0;
    for (size_t q0 = 0; q0 < Q; q0 += SLICE, ++i) {
        slice_t cur;
        cur.q0 = q0;
        cur.m = std::min(SLICE, Q - q0);
        cur.st = (i & 1) ? h->stream2 : h->stream;
        uint8_t *base = (uint8_t *)h->d_stage + (i & 1) * half;
        uint8_t *d_ascii = base, *d_packed = d_ascii + a_ascii, *d_valid = d_packed + a_packed;
        cur.d_lo = d_valid + a_valid;
        cur.d_up = cur.d_lo + m_max * 8;
        const size_t ascii_bytes = (cur.m - 1) * stride + k;  // the last k-mer needs only k bytes
        HIP_OK(hipMemcpyAsync(d_ascii, kmers + q0 * stride, ascii_bytes, hipMemcpyHostToDevice, cur.st));
        hipError_t e = launch_pack(d_ascii, cur.m, k, stride, d_packed, d_valid, cur.st);
        if (e != hipSuccess) return fail_hip(e, "pack kernel launch");
        rc = search_dev(h, d_packed, d_valid, cur.m, k, cur.d_lo, cur.d_up, counts_only, cur.st);
        if (rc) return rc;
        if (prev.m && (rc = collect(prev)) != RSBWT_OK) return rc;
        prev = cur;
    }
    if (prev.m && (rc = collect(prev)) != RSBWT_OK) return rc;
    return RSBWT_OK;
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
// With the wave kernel and a k-mer table that does not cover the whole k-mer, the k-mers
// themselves are searched first with a trace, and every variant whose substituted position lies
// left of the table's reach resumes from its k-mer's interval at that position instead of being
// searched from scratch (it shares that whole suffix).  `scratch` holds the trace and the k-mers'
// own results: variants_scratch_bytes().
static size_t variants_scratch_bytes(const rsbwt_t *h, size_t m, uint32_t k) {
    if (!search_uses_wave_kernel(h->view, &h->slots)) return 0;
    const uint32_t tn = wave_trace_entries(h->view, k);
    return tn ? m * (size_t)tn * 16 + 2 * m * 8 : 0;
}

static int search_variants(rsbwt_t *h, const void *d_pk, const void *d_ok, size_t m, uint32_t k, const void *d_vpk,
                           const void *d_vok, void *d_lo, void *d_up, uint8_t *scratch, hipStream_t stream) {
    const size_t V = 3 * (size_t)k + 1;
    const uint32_t tn = search_uses_wave_kernel(h->view, &h->slots) ? wave_trace_entries(h->view, k) : 0u;
    if (tn == 0) return search_dev(h, d_vpk, d_vok, m * V, k, d_lo, d_up, false, stream);
    uint8_t *d_trace = scratch, *d_olo = d_trace + m * (size_t)tn * 16, *d_oup = d_olo + m * 8;
    wave_search_extra traced;
    traced.d_trace_out = d_trace;
    traced.trace_n = tn;
    int rc = search_dev(h, d_pk, d_ok, m, k, d_olo, d_oup, false, stream, &traced);
    if (rc) return rc;
    wave_search_extra resumed;
    resumed.d_trace_in = d_trace;
    resumed.trace_n = tn;
    resumed.variants = (uint32_t)V;
    return search_dev(h, d_vpk, d_vok, m * V, k, d_lo, d_up, false, stream, &resumed);
}

int rsbwt_find_intervals_1mm(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride,
                             uint64_t *lower, uint64_t *upper) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    if (Q == 0) return RSBWT_OK;
    if (!kmers || !lower || !upper) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0 || stride < k) return fail(RSBWT_EINVAL, "bad k/stride");
    int rc = use_device(h->device);
    if (rc) return rc;
    const uint32_t wpq = words_per_kmer(k);
    const size_t V = 3 * (size_t)k + 1;
    const size_t SLICE = std::max<size_t>(1, (4u << 20) / V);  // ~4M variants per pass
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    for (size_t q0 = 0; q0 < Q; q0 += SLICE) {
        const size_t m = std::min(SLICE, Q - q0), mv = m * V;
        const size_t ascii_bytes = (m - 1) * stride + k;
        const size_t a_ascii = (ascii_bytes + 15) & ~(size_t)15, a_pk = m * wpq * 8, a_ok = (m + 15) & ~(size_t)15;
        const size_t a_vpk = mv * wpq * 8, a_vok = (mv + 15) & ~(size_t)15;
        const size_t a_scr = (variants_scratch_bytes(h, m, k) + 15) & ~(size_t)15;
        if ((rc = h->stage(a_ascii + a_pk + a_ok + a_vpk + a_vok + 2 * mv * 8 + a_scr)) != RSBWT_OK) return rc;
        uint8_t *d_ascii = (uint8_t *)h->d_stage, *d_pk = d_ascii + a_ascii, *d_ok = d_pk + a_pk;
        uint8_t *d_vpk = d_ok + a_ok, *d_vok = d_vpk + a_vpk, *d_lo = d_vok + a_vok, *d_up = d_lo + mv * 8;
        uint8_t *d_scr = d_up + mv * 8;
        HIP_OK(hipMemcpyAsync(d_ascii, kmers + q0 * stride, ascii_bytes, hipMemcpyHostToDevice, h->stream));
        hipError_t e = launch_pack(d_ascii, m, k, stride, d_pk, d_ok, h->stream);
        if (e == hipSuccess) e = launch_variants(d_pk, d_ok, m, k, d_vpk, d_vok, h->stream);
        if (e != hipSuccess) return fail_hip(e, "variant kernel launch");
        rc = search_variants(h, d_pk, d_ok, m, k, d_vpk, d_vok, d_lo, d_up, d_scr, h->stream);
        if (rc) return rc;
        HIP_OK(hipMemcpyAsync(lower + q0 * V, d_lo, mv * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipMemcpyAsync(upper + q0 * V, d_up, mv * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
    }
    return RSBWT_OK;
}

int rsbwt_hits_1mm(rsbwt_t *h, const char *kmers, size_t Q, uint32_t k, size_t stride, rsbwt_hit_1mm *hits,
                   size_t cap, size_t *nhits) {
    if (!h || !nhits) return fail(RSBWT_EINVAL, "null argument");
    *nhits = 0;
    if (Q == 0) return RSBWT_OK;
    if (!kmers || (!hits && cap)) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0 || stride < k) return fail(RSBWT_EINVAL, "bad k/stride");
    if (k > 32767u) return fail(RSBWT_ERANGE, "k %u: positions are reported as int16", k);
    if (Q > 0xFFFFFFFFull) return fail(RSBWT_ERANGE, "at most 2^32 - 1 k-mers per call");
    int rc = use_device(h->device);
    if (rc) return rc;
    const uint32_t wpq = words_per_kmer(k);
    const size_t V = 3 * (size_t)k + 1;
    const size_t SLICE = std::max<size_t>(1, (4u << 20) / V);  // ~4M variants per pass
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    std::vector<uint32_t> counts;
    std::vector<uint64_t> offsets;
    size_t total = 0;
    bool overflow = false;
    for (size_t q0 = 0; q0 < Q; q0 += SLICE) {
        const size_t m = std::min(SLICE, Q - q0), mv = m * V;
        const size_t ascii_bytes = (m - 1) * stride + k;
        const size_t a_ascii = (ascii_bytes + 15) & ~(size_t)15, a_pk = m * wpq * 8, a_ok = (m + 15) & ~(size_t)15;
        const size_t a_vpk = mv * wpq * 8, a_vok = (mv + 15) & ~(size_t)15;
        const size_t a_cnt = (m * 4 + 15) & ~(size_t)15, a_off = m * 8, a_hits = mv * sizeof(rsbwt_hit_1mm);
        const size_t a_scr = (variants_scratch_bytes(h, m, k) + 15) & ~(size_t)15;
        if ((rc = h->stage(a_ascii + a_pk + a_ok + a_vpk + a_vok + 2 * mv * 8 + a_cnt + a_off + a_hits + a_scr)) != RSBWT_OK) return rc;
        uint8_t *d_ascii = (uint8_t *)h->d_stage, *d_pk = d_ascii + a_ascii, *d_ok = d_pk + a_pk;
        uint8_t *d_vpk = d_ok + a_ok, *d_vok = d_vpk + a_vpk, *d_lo = d_vok + a_vok, *d_up = d_lo + mv * 8;
        uint8_t *d_cnt = d_up + mv * 8, *d_off = d_cnt + a_cnt, *d_hits = d_off + a_off, *d_scr = d_hits + a_hits;
        HIP_OK(hipMemcpyAsync(d_ascii, kmers + q0 * stride, ascii_bytes, hipMemcpyHostToDevice, h->stream));
        hipError_t e = launch_pack(d_ascii, m, k, stride, d_pk, d_ok, h->stream);
        if (e == hipSuccess) e = launch_variants(d_pk, d_ok, m, k, d_vpk, d_vok, h->stream);
        if (e != hipSuccess) return fail_hip(e, "variant kernel launch");
        rc = search_variants(h, d_pk, d_ok, m, k, d_vpk, d_vok, d_lo, d_up, d_scr, h->stream);
        if (rc) return rc;
        e = launch_hits1mm_count(d_lo, d_up, m, (uint32_t)V, d_cnt, h->stream);
        if (e != hipSuccess) return fail_hip(e, "hit count kernel launch");
        counts.resize(m);
        offsets.resize(m);
        HIP_OK(hipMemcpyAsync(counts.data(), d_cnt, m * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        uint64_t run = 0;
        for (size_t i = 0; i < m; ++i) {
            offsets[i] = run;
            run += counts[i];
        }
        if (!overflow && total + run <= cap && run) {
            HIP_OK(hipMemcpyAsync(d_off, offsets.data(), m * 8, hipMemcpyHostToDevice, h->stream));
            e = launch_hits1mm_write(d_lo, d_up, d_pk, m, (uint32_t)V, k, d_off, (uint32_t)q0, d_hits, h->stream);
            if (e != hipSuccess) return fail_hip(e, "hit write kernel launch");
            HIP_OK(hipMemcpyAsync(hits + total, d_hits, run * sizeof(rsbwt_hit_1mm), hipMemcpyDeviceToHost, h->stream));
            HIP_OK(hipStreamSynchronize(h->stream));
        } else if (total + run > cap) {
            overflow = true;  // keep counting so that the caller learns the size it needs
        }
        total += run;
    }
    *nhits = total;
    if (overflow) return fail(RSBWT_ERANGE, "%zu hits, room for %zu", total, cap);
    return RSBWT_OK;
}

// ---- read extraction --------------------------------------------------------------------------

int rsbwt_extract(rsbwt_t *h, const uint64_t *rows, size_t n, char *out, uint32_t stride, uint32_t *len,
                  uint32_t *prefix_len) {
    if (!h || (!rows && n) || (!out && n)) return fail(RSBWT_EINVAL, "null argument");
    if (n == 0) return RSBWT_OK;
    if (stride == 0) return fail(RSBWT_EINVAL, "stride must be positive");
    if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
    int rc = use_device(h->device);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    if (!h->d_sel) {
        const uint64_t words = 4 * select_sample_stride(h->view);
        HIP_OK(hipMalloc(&h->d_sel, words * sizeof(uint32_t)));
        HIP_OK(hipMemsetAsync(h->d_sel, 0, words * sizeof(uint32_t), h->stream));
        hipError_t e = launch_select_samples(h->view, h->d_sel, h->stream);
        if (e != hipSuccess) return fail_hip(e, "select sample kernel launch");
        h->hbm_bytes += words * sizeof(uint32_t);
    }
    const size_t SLICE = 1u << 20;
    for (size_t i0 = 0; i0 < n; i0 += SLICE) {
        const size_t m = std::min(SLICE, n - i0);
        const size_t a_rows = m * 8, a_out = (m * (size_t)stride + 15) & ~(size_t)15, a_len = m * 4;
        if ((rc = h->stage(a_rows + a_out + 2 * a_len)) != RSBWT_OK) return rc;
        uint8_t *base = (uint8_t *)h->d_stage;
        uint8_t *d_rows = base, *d_out = base + a_rows, *d_pl = d_out + a_out, *d_len = d_pl + a_len;
        HIP_OK(hipMemcpyAsync(d_rows, rows + i0, a_rows, hipMemcpyHostToDevice, h->stream));
        hipError_t e = launch_extract(h->view, h->d_sel, d_rows, m, d_out, stride, d_pl, d_len, h->stream);
        if (e != hipSuccess) return fail_hip(e, "extract kernel launch");
        HIP_OK(hipMemcpyAsync(out + i0 * (size_t)stride, d_out, m * (size_t)stride, hipMemcpyDeviceToHost, h->stream));
        if (len) HIP_OK(hipMemcpyAsync(len + i0, d_len, a_len, hipMemcpyDeviceToHost, h->stream));
        if (prefix_len) HIP_OK(hipMemcpyAsync(prefix_len + i0, d_pl, a_len, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
    }
    return RSBWT_OK;
}

// ---- measurement ------------------------------------------------------------------------------

int rsbwt_last_search_ms(rsbwt_t *h, float *ms) {
    if (!h || !ms) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    if (!h->launches) return fail(RSBWT_EINVAL, "no search has been launched through this handle");
    const int slot = (int)((h->launches - 1) % rsbwt::RING);
    HIP_OK(hipEventSynchronize(h->ev_stop[slot]));
    HIP_OK(hipEventElapsedTime(ms, h->ev_start[slot], h->ev_stop[slot]));
    return RSBWT_OK;
}

int rsbwt_search_history_ms(rsbwt_t *h, float *ms, size_t cap, size_t *count) {
    if (!h || (!ms && cap) || !count) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    size_t n = (size_t)std::min<uint64_t>(h->launches, rsbwt::RING);
    if (n > cap) n = cap;
    for (size_t i = 0; i < n; ++i) {  // oldest of the n first
        const int slot = (int)((h->launches - n + i) % rsbwt::RING);
        HIP_OK(hipEventSynchronize(h->ev_stop[slot]));
        HIP_OK(hipEventElapsedTime(&ms[i], h->ev_start[slot], h->ev_stop[slot]));
    }
    *count = n;
    return RSBWT_OK;
}

int rsbwt_set_counting(rsbwt_t *h, int on) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    h->counting = on != 0;
    return RSBWT_OK;
}

int rsbwt_last_search_work(rsbwt_t *h, uint64_t *lf_steps, uint64_t *occ_lookups, uint64_t *block_reads) {
    if (!h) return fail(RSBWT_EINVAL, "null handle");
    int rc = use_device(h->device);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    if (!h->launches) return fail(RSBWT_EINVAL, "no search has been launched through this handle");
    HIP_OK(hipEventSynchronize(h->ev_stop[(h->launches - 1) % rsbwt::RING]));
    unsigned long long w[4];
    HIP_OK(hipMemcpy(w, h->d_work, sizeof w, hipMemcpyDeviceToHost));
    if (lf_steps) *lf_steps = w[0];
    if (occ_lookups) *occ_lookups = w[1];
    if (block_reads) *block_reads = w[2];
    return RSBWT_OK;
}

int rsbwt_last_search_phases(rsbwt_t *h, uint64_t *cycles, uint64_t *passes) {
    if (!h || !cycles || !passes) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    if (!h->launches) return fail(RSBWT_EINVAL, "no search has been launched through this handle");
    HIP_OK(hipEventSynchronize(h->ev_stop[(h->launches - 1) % rsbwt::RING]));
    unsigned long long w[16];
    HIP_OK(hipMemcpy(w, h->d_work, sizeof w, hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) cycles[i] = w[4 + i];
    *passes = w[10];
    return RSBWT_OK;
}

int rsbwt_last_search_ktab_lookups(rsbwt_t *h, uint64_t *lookups) {
    if (!h || !lookups) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lock(h->mu);
    if (!h->launches) return fail(RSBWT_EINVAL, "no search has been launched through this handle");
    HIP_OK(hipEventSynchronize(h->ev_stop[(h->launches - 1) % rsbwt::RING]));
    unsigned long long w[4];
    HIP_OK(hipMemcpy(w, h->d_work, sizeof w, hipMemcpyDeviceToHost));
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

// ---- shard sets -------------------------------------------------------------------------------

int rsbwt_set_open(const char *const *bwt_paths, size_t num_shards, const int *device_map,
                   uint32_t flags, rsbwt_set_t **out) {
    if (!out || (!bwt_paths && num_shards)) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    rsbwt_set_t *s = new (std::nothrow) rsbwt_set();
    if (!s) return fail(RSBWT_ENOMEM, "host allocation failed");
    s->owns = true;
    for (size_t i = 0; i < num_shards; ++i) {
        rsbwt_t *h = nullptr;
        // many shards per GPU: the slot layout (about as large as the index) only on explicit request
        const uint32_t f = (num_shards > 1 && (flags & RSBWT_SLOTS_MASK) == RSBWT_SLOTS_AUTO) ? (flags | RSBWT_SLOTS_OFF) : flags;
        int rc = rsbwt_open(bwt_paths[i], device_map ? device_map[i] : 0, f, &h);
        if (rc) { rsbwt_set_close(s); return rc; }
        s->shards.push_back(h);
    }
    *out = s;
    return RSBWT_OK;
}

int rsbwt_set_from_handles(rsbwt_t *const *handles, size_t num_shards, rsbwt_set_t **out) {
    if (!out || (!handles && num_shards)) return fail(RSBWT_EINVAL, "null argument");
    rsbwt_set_t *s = new (std::nothrow) rsbwt_set();
    if (!s) return fail(RSBWT_ENOMEM, "host allocation failed");
    s->owns = false;
    s->shards.assign(handles, handles + num_shards);
    *out = s;
    return RSBWT_OK;
}

void rsbwt_set_close(rsbwt_set_t *s) {
    if (!s) return;
    if (s->owns)
        for (rsbwt_t *h : s->shards) rsbwt_close(h);
    delete s;
}

size_t rsbwt_set_size(const rsbwt_set_t *s) { return s ? s->shards.size() : 0; }
rsbwt_t *rsbwt_set_shard(rsbwt_set_t *s, size_t i) { return (s && i < s->shards.size()) ? s->shards[i] : nullptr; }

int rsbwt_set_find_intervals(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride,
                             uint64_t *lower, uint64_t *upper) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    for (size_t i = 0; i < s->shards.size(); ++i) {
        int rc = rsbwt_find_intervals(s->shards[i], kmers, Q, k, stride, lower + i * Q, upper + i * Q);
        if (rc) return rc;
    }
    return RSBWT_OK;
}

int rsbwt_set_count(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *counts) {
    if (!s || (!counts && Q)) return fail(RSBWT_EINVAL, "null argument");
    std::vector<uint64_t> tmp(Q);
    for (size_t q = 0; q < Q; ++q) counts[q] = 0;
    for (size_t i = 0; i < s->shards.size(); ++i) {
        int rc = rsbwt_count(s->shards[i], kmers, Q, k, stride, tmp.data());
        if (rc) return rc;
        for (size_t q = 0; q < Q; ++q) counts[q] += tmp[q];
    }
    return RSBWT_OK;
}

}  // extern "C"

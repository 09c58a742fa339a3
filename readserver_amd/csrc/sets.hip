// sets.hip -- shard sets: the shards one process holds on its GPU(s), searched by one
// call.  Every query goes to every shard (src/service/server.cpp:124,578); per-shard results are only
// concatenated (intervals) or summed (counts) (server.cpp:184-197,404-410).
//
// The shards of one device form a GROUP and are searched by ONE fused launch (search_lines.hip:
// (query, shard) pairs drawn from per-shard pools): the batch is uploaded and packed once per device,
// not once per shard.  Devices are driven concurrently, one host thread each.  With more than one
// device the per-device count sums are reduced onto the first device over RCCL (xGMI) and cross PCIe
// once; interval gathers onto one GPU are offered for GPU-resident consumers
// (rsbwt_set_gather_intervals_dev).  RCCL is bound at run time (dlopen): a single-GPU deployment
// needs no librccl.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <thread>
#include <vector>

#include "../../include/rsbwt.h"
#include "capi_guard.h"
#include "capi_internal.h"

using namespace rsb;

#ifndef RSB_1MM_TABLE_PREPASS_DEFAULT
#define RSB_1MM_TABLE_PREPASS_DEFAULT true
#endif

namespace {

#define HIP_OK(x)                                              \
    do {                                                       \
        hipError_t _e = (x);                                   \
        if (_e != hipSuccess) return fail_hip(_e, #x);         \
    } while (0)

struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

rccl_api &rccl() {
    static rccl_api api = [] {
        rccl_api a;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
#define RCCL_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, sym))
        RCCL_SYM(CommInitAll, "ncclCommInitAll");
        RCCL_SYM(CommDestroy, "ncclCommDestroy");
        RCCL_SYM(GroupStart, "ncclGroupStart");
        RCCL_SYM(GroupEnd, "ncclGroupEnd");
        RCCL_SYM(Send, "ncclSend");
        RCCL_SYM(Recv, "ncclRecv");
        RCCL_SYM(Reduce, "ncclReduce");
        RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
        a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv && a.Reduce && a.GetErrorString;
        return a;
    }();
    return api;
}

// counts[S][m] -> sum[m]
__global__ void __launch_bounds__(256)
sum_rows_kernel(const uint64_t *__restrict__ rows, uint32_t S, size_t m, uint64_t *__restrict__ sum) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint64_t s = 0;
    for (uint32_t r = 0; r < S; ++r) s += rows[(size_t)r * m + i];
    sum[i] = s;
}

}  // namespace

// the shards of one device
struct dev_group : search_meter {
    int device = 0;   // the GPU
    int logical = 0;  // the device number its shards were opened with: what groups them (= device outside tests, capi.hip resolve_device)
    int num_cus = 256;
    std::vector<size_t> idx;         // positions in the set, ascending
    shard_view *d_views = nullptr;   // [idx.size()] in HBM: what the searches read (written by make_groups and when tables are attached, never else)
    shard_view *d_xviews = nullptr;  // the same WITH the shards' select samples, for the fused extraction: a second array, made once
    std::atomic<bool> xviews_ready{false};
    uint64_t *d_ktab = nullptr;      // the interleaved k-mer tables of the shards this set gave one
    ctx_pool pool;
    ncclComm_t comm = nullptr;
    // the gather without RCCL (peer copies issued on the root's stream): "the block is ready" on this group's stream,
    // "the block has been read" back onto it
    hipEvent_t peer_ready = nullptr, peer_done = nullptr;
    // side streams for calls that let the group's shards work side by side (small 1-mismatch batches: one shard's
    // launch does not fill the GPU), forked from and joined to the caller's stream with events; made on first use
    static constexpr int FORK = 8;
    hipStream_t fork_st[FORK] = {};
    hipEvent_t fork_ev = nullptr, join_ev[FORK] = {};
    std::mutex fork_mu;
    int ensure_fork() {
        if (fork_ev) return RSBWT_OK;  // (set last: everything below exists)
        // each stream / event is made once, whatever a previous, partly failed attempt left behind
        for (int i = 0; i < FORK; ++i) {
            if ((!fork_st[i] && hipStreamCreateWithFlags(&fork_st[i], hipStreamNonBlocking) != hipSuccess) ||
                (!join_ev[i] && hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming) != hipSuccess)) {
                (void)hipGetLastError();
                return fail(RSBWT_EHIP, "cannot create the side streams of a shard set");
            }
        }
        if (hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming) != hipSuccess) {
            fork_ev = nullptr;
            return fail(RSBWT_EHIP, "cannot create an event");
        }
        return RSBWT_OK;
    }
};

struct rsbwt_set {
    std::vector<rsbwt_t *> shards;
    bool owns = false;
    std::vector<dev_group *> groups;
    bool comms_tried = false, comms_ok = false;
    std::mutex mu;
    // Collectives on the set's communicators are enqueued by one thread at a time: two callers
    // interleaving their group calls could reach the communicators in different orders.
    std::mutex comm_mu;
};

namespace {

int publish_views(rsbwt_set_t *s) {
    for (dev_group *g : s->groups) {
        int rc = use_device(g->device);
        if (rc) return rc;
        std::vector<shard_view> v;
        for (size_t i : g->idx) v.push_back(s->shards[i]->view);
        HIP_OK(hipMemcpy(g->d_views, v.data(), v.size() * sizeof(shard_view), hipMemcpyHostToDevice));
    }
    return RSBWT_OK;
}

int make_groups(rsbwt_set_t *s) {
    for (size_t i = 0; i < s->shards.size(); ++i) {
        rsbwt_t *h = s->shards[i];
        if (!h) return fail(RSBWT_EINVAL, "null shard handle");
        dev_group *g = nullptr;
        for (dev_group *x : s->groups)
            if (x->logical == h->logical_device) g = x;
        if (!g) {
            g = new (std::nothrow) dev_group();
            if (!g) return fail(RSBWT_ENOMEM, "host allocation failed");
            g->device = h->device;
            g->logical = h->logical_device;
            g->num_cus = h->num_cus;
            s->groups.push_back(g);
        }
        g->idx.push_back(i);
    }
    for (dev_group *g : s->groups) {
        int rc = use_device(g->device);
        if (rc) return rc;
        HIP_OK(hipMalloc(&g->d_views, g->idx.size() * sizeof(shard_view)));
        HIP_OK(hipMalloc(&g->d_work, WORK_WORDS * sizeof(unsigned long long)));
        for (int i = 0; i < search_meter::RING; ++i) {
            HIP_OK(hipEventCreate(&g->ev_start[i]));
            HIP_OK(hipEventCreate(&g->ev_stop[i]));
        }
    }
    return publish_views(s);
}

// one communicator per device group, all in this process (ncclCommInitAll); false = no RCCL here
bool ensure_comms(rsbwt_set_t *s) {
    std::lock_guard<std::mutex> lock(s->mu);
    if (s->comms_tried) return s->comms_ok;
    s->comms_tried = true;
    if (s->groups.size() < 2 || !rccl().ok) return false;
    std::vector<int> devs;
    for (dev_group *g : s->groups) {
        // two groups on one GPU (logical devices of the test hook, capi.hip resolve_device): RCCL refuses two ranks
        // on one device, and the set goes the way a box without librccl goes
        if (std::find(devs.begin(), devs.end(), g->device) != devs.end()) return false;
        devs.push_back(g->device);
    }
    std::vector<ncclComm_t> comms(devs.size());
    if (rccl().CommInitAll(comms.data(), (int)devs.size(), devs.data()) != ncclSuccess) return false;
    for (size_t i = 0; i < devs.size(); ++i) s->groups[i]->comm = comms[i];
    s->comms_ok = true;
    return true;
}

// search_extra::narrow for a fused launch over a group's shards: every one of them behind a k-mer table deep enough
// that what is left of a search are steps inside one window (capi_internal.h, view_is_narrow) -- the launch then runs
// one lane per search (search_solo.h)
bool group_is_narrow(const rsbwt_set_t *s, const dev_group *g, uint32_t k) {
    for (size_t i : g->idx)
        if (!view_is_narrow(s->shards[i]->view, k)) return false;
    return !g->idx.empty();
}

// runs fn(group index) for every device group, concurrently when there are several
template <class F>
int for_each_group(rsbwt_set_t *s, F &&fn) {
    const size_t G = s->groups.size();
    if (G == 1) return fn(0);
    std::vector<int> rcs(G, RSBWT_OK);
    std::vector<std::string> errs(G);
    std::vector<std::thread> th;
    for (size_t g = 0; g < G; ++g)
        th.emplace_back([&, g] {
            rcs[g] = fn(g);
            if (rcs[g]) errs[g] = rsbwt_last_error();  // the message is thread-local: carry it over
        });
    for (auto &t : th) t.join();
    for (size_t g = 0; g < G; ++g)
        if (rcs[g]) return fail(rcs[g], "%s", errs[g].c_str());
    return RSBWT_OK;
}

}  // namespace

extern "C" {

void rsbwt_set_close(rsbwt_set_t *s) {
    if (!s) return;
    for (dev_group *g : s->groups) {
        (void)hipSetDevice(g->device);
        g->pool.destroy();
        if (g->comm && rccl().ok) (void)rccl().CommDestroy(g->comm);
        if (g->d_ktab) {  // shards that outlive the set lose the table that lived in it
            for (size_t i : g->idx)
                if (!s->owns && s->shards[i]) (void)detach_ktab(s->shards[i]);
            (void)hipFree(g->d_ktab);
        }
        for (int i = 0; i < dev_group::FORK; ++i) {
            if (g->fork_st[i]) {
                (void)hipStreamSynchronize(g->fork_st[i]);
                (void)hipStreamDestroy(g->fork_st[i]);
            }
            if (g->join_ev[i]) (void)hipEventDestroy(g->join_ev[i]);
        }
        if (g->fork_ev) (void)hipEventDestroy(g->fork_ev);
        if (g->peer_ready) (void)hipEventDestroy(g->peer_ready);
        if (g->peer_done) (void)hipEventDestroy(g->peer_done);
        if (g->d_views) (void)hipFree(g->d_views);
        if (g->d_xviews) (void)hipFree(g->d_xviews);
        if (g->d_work) (void)hipFree(g->d_work);
        g->scratch.destroy();
        for (int i = 0; i < search_meter::RING; ++i) {
            if (g->ev_start[i]) (void)hipEventDestroy(g->ev_start[i]);
            if (g->ev_stop[i]) (void)hipEventDestroy(g->ev_stop[i]);
        }
        delete g;
    }
    if (s->owns)
        for (rsbwt_t *h : s->shards) rsbwt_close(h);
    delete s;
}

static int rsbwt_set_from_handles_body(rsbwt_t *const *handles, size_t num_shards, rsbwt_set_t **out) {
    if (!out || (!handles && num_shards)) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    if (num_shards == 0) return fail(RSBWT_EINVAL, "a shard set needs at least one shard");
    if (num_shards > (1u << 16)) return fail(RSBWT_ERANGE, "%zu shards: a set holds at most 65536", num_shards);
    rsbwt_set_t *s = new (std::nothrow) rsbwt_set();
    if (!s) return fail(RSBWT_ENOMEM, "host allocation failed");
    s->owns = false;
    try {
        s->shards.assign(handles, handles + num_shards);
    } catch (...) {
        delete s;
        throw;  // guarded() turns it into RSBWT_ENOMEM
    }
    int rc = make_groups(s);
    if (rc) { rsbwt_set_close(s); return rc; }
    *out = s;
    return RSBWT_OK;
}
int rsbwt_set_from_handles(rsbwt_t *const *handles, size_t num_shards, rsbwt_set_t **out) {
    return guarded("rsbwt_set_from_handles", [&]() -> int { return rsbwt_set_from_handles_body(handles, num_shards, out); });
}


static int rsbwt_set_attach_ktabs_body(rsbwt_set_t *s, uint32_t depth, uint32_t format);
static int rsbwt_set_open_body(const char *const *bwt_paths, size_t num_shards, const int *device_map,
                   uint32_t flags, rsbwt_set_t **out) {
    if (!out || (!bwt_paths && num_shards)) return fail(RSBWT_EINVAL, "null argument");
    *out = nullptr;
    if (num_shards == 0) return fail(RSBWT_EINVAL, "a shard set needs at least one shard");
    if (num_shards > (1u << 16)) return fail(RSBWT_ERANGE, "%zu shards: a set holds at most 65536", num_shards);
    rsbwt_set_t *s = new (std::nothrow) rsbwt_set();
    if (!s) return fail(RSBWT_ENOMEM, "host allocation failed");
    s->owns = true;
    // The k-mer tables of the shards of one GPU share what HBM the lines leave, so with depth "auto"
    // the shards are opened without tables and the depth is chosen per device afterwards.
    const uint32_t T_req = (flags & RSBWT_KTAB_MASK) >> RSBWT_KTAB_SHIFT;
    const uint32_t open_flags = T_req == 0u ? ((flags & ~RSBWT_KTAB_MASK) | RSBWT_KTAB_NONE) : flags;
    for (size_t i = 0; i < num_shards; ++i) {
        rsbwt_t *h = nullptr;
        int rc = rsbwt_open(bwt_paths[i], device_map ? device_map[i] : 0, open_flags, &h);
        if (rc) { rsbwt_set_close(s); return rc; }
        s->shards.push_back(h);
    }
    int rc = make_groups(s);
    // (RSBWT_OPEN_KTAB_GROUPED: the grouped format where it is a level deeper and nearly every T-mer occurs, else plain)
    if (rc == RSBWT_OK && T_req == 0u)
        rc = rsbwt_set_attach_ktabs_body(s, 0, (flags & RSBWT_OPEN_KTAB_GROUPED) ? RSBWT_KTAB_FORMAT_AUTO : RSBWT_KTAB_FORMAT_PLAIN);
    if (rc) { rsbwt_set_close(s); return rc; }
    *out = s;
    return RSBWT_OK;
}
int rsbwt_set_open(const char *const *bwt_paths, size_t num_shards, const int *device_map,
                   uint32_t flags, rsbwt_set_t **out) {
    return guarded("rsbwt_set_open", [&]() -> int { return rsbwt_set_open_body(bwt_paths, num_shards, device_map, flags, out); });
}


// The depth rsbwt_set_attach_ktabs(s, 0) would give the shards of device group g: the deepest T whose
// tables (one per shard of that device without one) fit three quarters of the device's free HBM and leave 8 GiB of it
// (the shards are resident when this is asked: what is free is what the tables and the caller's batch buffers share --
// a third of it, the rule until round 4, left 8 x 20 GB shards at 13-mer tables where 15-mer ones fit), none larger
// than 5/4 of its shard's lines, with 4^T <= the smallest shard's length; at most 16 (grouped: 17); 0 = none.
// *fmt: in = the format asked for (KTAB_PLAIN / KTAB_GROUPED), out = the one to build (auto_ktab_depth_for).
static uint32_t auto_ktab_depth(rsbwt_set_t *s, dev_group *g, uint32_t *fmt, uint64_t keep_free = 0) {
    if (use_device(g->device)) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    uint64_t min_n = ~0ull, min_bytes = ~0ull, need = 0;
    for (size_t i : g->idx) {
        const rsbwt_t *h = s->shards[i];
        if (h->view.ktab || h->view.n == 0) continue;
        ++need;
        min_n = std::min(min_n, h->view.n);
        min_bytes = std::min(min_bytes, h->hbm_bytes);
    }
    if (!need) return 0;
    // (keep_free: what the caller wants left of the free HBM; 0 = the set's own rule, a quarter of it and at least 8 GiB)
    const uint64_t keep = keep_free ? keep_free : std::max<uint64_t>(8ull << 30, free_b / 4);
    const uint64_t budget = std::min<uint64_t>(min_bytes + min_bytes / 4, (free_b > keep ? free_b - keep : 0) / need);
    return auto_ktab_depth_for(budget, min_n, fmt);
}

// THE table-sizing rule, for every caller (rsbwt_set_open, rsbwt_set_attach_ktabs*(depth 0), bench.py, onehost.py --
// until round 5 bench.py restated it in Python and rsbwt_set_auto_ktab_depth answered for the plain format only):
// the depth and the format the shards of the set that have no table yet would get, one pair for the whole set (the
// shallowest over its devices; plain if any device must).  format_in: RSBWT_KTAB_FORMAT_PLAIN / _GROUPED / _AUTO
// (grouped where that is deeper and its records can say what four siblings hold).
int rsbwt_set_auto_ktab(rsbwt_set_t *s, uint32_t format_in, uint64_t keep_free_bytes, uint32_t *depth, uint32_t *format_out) {
    if (!s || !depth || !format_out) return fail(RSBWT_EINVAL, "null argument");
    if (format_in > RSBWT_KTAB_FORMAT_AUTO) return fail(RSBWT_EINVAL, "k-mer table format %u", format_in);
    uint32_t T = ~0u, F = KTAB_GROUPED;
    for (dev_group *g : s->groups) {
        uint32_t fmt = format_in == RSBWT_KTAB_FORMAT_PLAIN ? KTAB_PLAIN : KTAB_GROUPED;
        const uint32_t t = auto_ktab_depth(s, g, &fmt, keep_free_bytes);
        if (t < T || (t == T && fmt == KTAB_PLAIN)) {
            if (t < T) F = fmt;
            else F = KTAB_PLAIN;
            T = t;
        }
    }
    if (T == ~0u) T = 0;
    // one format for the set: a device that must take the plain format at the common depth decides it
    if (F == KTAB_GROUPED)
        for (dev_group *g : s->groups) {
            uint64_t min_n = ~0ull;
            for (size_t i : g->idx)
                if (!s->shards[i]->view.ktab && s->shards[i]->view.n) min_n = std::min(min_n, s->shards[i]->view.n);
            if (min_n != ~0ull && !ktab_grouped_sensible(min_n, T)) F = KTAB_PLAIN;
        }
    *depth = T;
    *format_out = T >= 2u ? (F == KTAB_GROUPED ? RSBWT_KTAB_FORMAT_GROUPED : RSBWT_KTAB_FORMAT_PLAIN) : RSBWT_KTAB_FORMAT_PLAIN;
    return RSBWT_OK;
}

uint32_t rsbwt_set_auto_ktab_depth(rsbwt_set_t *s) {  // (the plain format's depth: rsbwt_set_auto_ktab names the format too)
    uint32_t T = 0, F = 0;
    return rsbwt_set_auto_ktab(s, RSBWT_KTAB_FORMAT_PLAIN, 0, &T, &F) == RSBWT_OK ? T : 0u;
}

// Builds the k-mer tables of the shards that have none.  depth 0 = per device, auto_ktab_depth.  The
// tables of one device are interleaved in one allocation of the set (line_format.h): the start
// records of a query for all shards come out of one stretch of 8 x shards (grouped: 12 x shards) bytes.
// format: RSBWT_KTAB_FORMAT_PLAIN / _GROUPED / _AUTO (grouped where the smallest shard's groups fit their records).
static int rsbwt_set_attach_ktabs_body(rsbwt_set_t *s, uint32_t depth, uint32_t format) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (format > RSBWT_KTAB_FORMAT_AUTO) return fail(RSBWT_EINVAL, "k-mer table format %u", format);
    for (dev_group *g : s->groups) {
        if (g->d_ktab) continue;
        std::vector<size_t> need;
        uint64_t min_n = ~0ull;
        for (size_t i : g->idx)
            if (!s->shards[i]->view.ktab && s->shards[i]->view.n) {
                need.push_back(i);
                min_n = std::min(min_n, s->shards[i]->view.n);
            }
        if (need.empty()) continue;
        uint32_t fmt = format == RSBWT_KTAB_FORMAT_PLAIN ? KTAB_PLAIN : KTAB_GROUPED;
        uint32_t T = depth;
        if (T == 0u) T = auto_ktab_depth(s, g, &fmt);
        else if (format == RSBWT_KTAB_FORMAT_AUTO && !ktab_grouped_sensible(min_n, T)) fmt = KTAB_PLAIN;
        if (T < 2u) continue;
        const uint32_t Tmax = fmt == KTAB_GROUPED ? KTAB_MAX_DEPTH_GROUPED : KTAB_MAX_DEPTH_PLAIN;
        if (T > Tmax) T = Tmax;
        int rc = use_device(g->device);
        if (rc) return rc;
        const uint64_t per = fmt == KTAB_GROUPED ? KTAB_GROUP_BYTES : 8u;  // bytes of one shard's entry / record in a stretch
        hipError_t e = hipMalloc(&g->d_ktab, ktab_bytes(fmt, T) * need.size());
        if (e != hipSuccess) return fail_hip(e, "allocating the k-mer tables");
        for (size_t j = 0; j < need.size(); ++j) {
            rsbwt_t *h = s->shards[need[j]];
            rc = attach_ktab_into(h, T, reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(g->d_ktab) + j * per), (uint32_t)need.size(), fmt);
            if (rc) return rc;
            h->ktab_owned = false;
        }
    }
    return publish_views(s);
}
int rsbwt_set_attach_ktabs(rsbwt_set_t *s, uint32_t depth) {
    return guarded("rsbwt_set_attach_ktabs", [&]() -> int { return rsbwt_set_attach_ktabs_body(s, depth, RSBWT_KTAB_FORMAT_PLAIN); });
}
int rsbwt_set_attach_ktabs_format(rsbwt_set_t *s, uint32_t depth, uint32_t format) {
    return guarded("rsbwt_set_attach_ktabs_format", [&]() -> int { return rsbwt_set_attach_ktabs_body(s, depth, format); });
}


size_t rsbwt_set_size(const rsbwt_set_t *s) { return s ? s->shards.size() : 0; }
rsbwt_t *rsbwt_set_shard(rsbwt_set_t *s, size_t i) { return (s && i < s->shards.size()) ? s->shards[i] : nullptr; }
size_t rsbwt_set_devices(const rsbwt_set_t *s) { return s ? s->groups.size() : 0; }

// Host buffers.  lower/upper: [num_shards][Q] in the set's shard order.
static int rsbwt_set_find_intervals_body(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride,
                             uint64_t *lower, uint64_t *upper) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (Q == 0) return RSBWT_OK;
    if (!kmers || !lower || !upper) return fail(RSBWT_EINVAL, "null argument");
    if (stride < k) return fail(RSBWT_EINVAL, "stride %zu < k %u", stride, k);
    const size_t S = s->shards.size();
    if (k == 0) {
        for (size_t i = 0; i < S * Q; ++i) { lower[i] = 1; upper[i] = 0; }
        return RSBWT_OK;
    }
    for (rsbwt_t *h : s->shards)
        if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index in the set");
    return for_each_group(s, [&](size_t gi) -> int {
        dev_group *g = s->groups[gi];
        int rc = use_device(g->device);
        if (rc) return rc;
        // a group's shards are searched into a [S_g][Q] block; when they sit next to each other in
        // the set (the usual map: shard s -> GPU s / 8) that block IS the output
        const size_t Sg = g->idx.size();
        const bool contiguous = g->idx.back() - g->idx.front() + 1 == Sg;
        if (contiguous)
            return search_host_views(*g, g->pool, g->d_views, (uint32_t)Sg, g->num_cus, kmers, Q, k, stride,
                                     lower + g->idx.front() * Q, upper + g->idx.front() * Q, false, group_is_narrow(s, g, k));
        std::vector<uint64_t> lo(Sg * Q), up(Sg * Q);
        rc = search_host_views(*g, g->pool, g->d_views, (uint32_t)Sg, g->num_cus, kmers, Q, k, stride, lo.data(), up.data(), false,
                               group_is_narrow(s, g, k));
        if (rc) return rc;
        for (size_t j = 0; j < Sg; ++j) {
            memcpy(lower + g->idx[j] * Q, lo.data() + j * Q, Q * 8);
            memcpy(upper + g->idx[j] * Q, up.data() + j * Q, Q * 8);
        }
        return RSBWT_OK;
    });
}
int rsbwt_set_find_intervals(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride,
                             uint64_t *lower, uint64_t *upper) {
    return guarded("rsbwt_set_find_intervals", [&]() -> int { return rsbwt_set_find_intervals_body(s, kmers, Q, k, stride, lower, upper); });
}

// Queries of lengths of their own: query q = text[off[q] .. off[q+1]) -- a window of the service loop in ONE search
// (search_lines.hip, search_init_var_kernel).  lower / upper: [num_shards][Q]; an empty query, one with a symbol
// outside ACGT or one longer than 65,535 symbols ends as (1, 0).
static int rsbwt_set_find_intervals_var_body(rsbwt_set_t *s, const char *text, const uint64_t *off, size_t Q, uint64_t *lower,
                                             uint64_t *upper, bool counts_only) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (Q == 0) return RSBWT_OK;
    if (!off || !lower || (!upper && !counts_only) || (!text && off[Q] != off[0])) return fail(RSBWT_EINVAL, "null argument");
    for (rsbwt_t *h : s->shards)
        if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index in the set");
    uint64_t kmax = 0;
    for (size_t q = 0; q < Q; ++q)
        if (off[q + 1] >= off[q] && off[q + 1] - off[q] <= 65535ull) kmax = std::max(kmax, off[q + 1] - off[q]);
    return for_each_group(s, [&](size_t gi) -> int {
        dev_group *g = s->groups[gi];
        int rc = use_device(g->device);
        if (rc) return rc;
        const size_t Sg = g->idx.size();
        const bool narrow = kmax != 0 && group_is_narrow(s, g, (uint32_t)kmax);
        const bool contiguous = g->idx.back() - g->idx.front() + 1 == Sg;
        if (contiguous)
            return search_host_views_var(*g, g->pool, g->d_views, (uint32_t)Sg, g->num_cus, text, off, Q, lower + g->idx.front() * Q,
                                         counts_only ? nullptr : upper + g->idx.front() * Q, counts_only, narrow);
        std::vector<uint64_t> lo(Sg * Q), up(counts_only ? 0 : Sg * Q);
        rc = search_host_views_var(*g, g->pool, g->d_views, (uint32_t)Sg, g->num_cus, text, off, Q, lo.data(), counts_only ? nullptr : up.data(),
                                   counts_only, narrow);
        if (rc) return rc;
        for (size_t j = 0; j < Sg; ++j) {
            memcpy(lower + g->idx[j] * Q, lo.data() + j * Q, Q * 8);
            if (!counts_only) memcpy(upper + g->idx[j] * Q, up.data() + j * Q, Q * 8);
        }
        return RSBWT_OK;
    });
}
int rsbwt_set_find_intervals_var(rsbwt_set_t *s, const char *text, const uint64_t *off, size_t Q, uint64_t *lower, uint64_t *upper) {
    return guarded("rsbwt_set_find_intervals_var", [&]() -> int { return rsbwt_set_find_intervals_var_body(s, text, off, Q, lower, upper, false); });
}
// counts[Q] summed over the set's shards (what rsbwt_set_count gives for one length); the sum is made on the host
int rsbwt_set_count_var(rsbwt_set_t *s, const char *text, const uint64_t *off, size_t Q, uint64_t *counts) {
    return guarded("rsbwt_set_count_var", [&]() -> int {
        if (!s || (!counts && Q)) return fail(RSBWT_EINVAL, "null argument");
        if (Q == 0) return RSBWT_OK;
        const size_t S = s->shards.size();
        std::vector<uint64_t> c(S * Q);
        const int rc = rsbwt_set_find_intervals_var_body(s, text, off, Q, c.data(), nullptr, true);
        if (rc) return rc;
        for (size_t q = 0; q < Q; ++q) {
            uint64_t t = 0;
            for (size_t i = 0; i < S; ++i) t += c[i * Q + q];
            counts[q] = t;
        }
        return RSBWT_OK;
    });
}


// counts[Q] summed over the set's shards, the way the front-end sums per-partition replies
// (src/service/server.cpp:184-197): per device one fused search + a row sum; the per-device sums
// are reduced onto the first device over RCCL when there are several, and cross PCIe once.
static int rsbwt_set_count_body(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *counts) {
    if (!s || (!counts && Q)) return fail(RSBWT_EINVAL, "null argument");
    if (Q == 0) return RSBWT_OK;
    if (!kmers) return fail(RSBWT_EINVAL, "null argument");
    if (stride < k) return fail(RSBWT_EINVAL, "stride %zu < k %u", stride, k);
    if (k == 0) {
        for (size_t q = 0; q < Q; ++q) counts[q] = 0;
        return RSBWT_OK;
    }
    if (k > 65535u) return fail(RSBWT_ERANGE, "k %u: at most 65535 symbols per k-mer", k);
    for (rsbwt_t *h : s->shards)
        if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index in the set");
    const size_t G = s->groups.size();
    const bool use_rccl = G > 1 && ensure_comms(s);
    const uint32_t wpq = k ? (k + 31u) / 32u : 1u;
    const size_t SLICE = 1u << 20;
    std::vector<std::vector<uint64_t>> part(use_rccl ? 0 : G);
    for (size_t q0 = 0; q0 < Q; q0 += SLICE) {
        const size_t m = std::min(SLICE, Q - q0);
        std::vector<call_ctx *> ctx(G, nullptr);
        std::vector<uint64_t *> d_sum(G, nullptr);
        // phase 1, per device concurrently: upload, pack, fused count search, row sum
        int rc = for_each_group(s, [&](size_t gi) -> int {
            dev_group *g = s->groups[gi];
            int r = use_device(g->device);
            if (r) return r;
            call_ctx *c = g->pool.acquire();
            if (!c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
            ctx[gi] = c;
            const size_t Sg = g->idx.size();
            const size_t ascii_bytes = (m - 1) * stride + k;
            const size_t a_ascii = (ascii_bytes + 15) & ~(size_t)15, a_pk = (m * wpq * 8 + 15) & ~(size_t)15, a_ok = (m + 15) & ~(size_t)15;
            if ((r = c->stage(a_ascii + a_pk + a_ok + (Sg + 2) * m * 8)) != RSBWT_OK) return r;
            uint8_t *d_ascii = (uint8_t *)c->d_stage, *d_pk = d_ascii + a_ascii, *d_ok = d_pk + a_pk;
            uint64_t *d_cnt = (uint64_t *)(d_ok + a_ok);
            d_sum[gi] = d_cnt + Sg * m;  // followed by m more words: the reduced total on the root
            hipStream_t st = c->st[0];
            HIP_OK(hipMemcpyAsync(d_ascii, kmers + q0 * stride, ascii_bytes, hipMemcpyHostToDevice, st));
            hipError_t e = launch_pack(d_ascii, m, k, stride, d_pk, d_ok, st);
            if (e != hipSuccess) return fail_hip(e, "pack kernel launch");
            search_extra ex;
            ex.narrow = group_is_narrow(s, g, k);
            r = search_launch(*g, g->d_views, (uint32_t)Sg, g->num_cus, d_pk, d_ok, m, k, d_cnt, nullptr, true, st, &ex);
            if (r) return r;
            hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_cnt, (uint32_t)Sg, m, d_sum[gi]);
            HIP_OK(hipGetLastError());
            if (!use_rccl) {
                part[gi].resize(m);
                HIP_OK(hipMemcpyAsync(part[gi].data(), d_sum[gi], m * 8, hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
            }
            return RSBWT_OK;
        });
        // phase 2: one sum over the devices
        if (rc == RSBWT_OK && use_rccl) {
            std::lock_guard<std::mutex> comm_lock(s->comm_mu);
            ncclResult_t nr = rccl().GroupStart();
            for (size_t gi = 0; gi < G && nr == ncclSuccess; ++gi) {
                (void)hipSetDevice(s->groups[gi]->device);
                nr = rccl().Reduce(d_sum[gi], d_sum[gi] + m, m, ncclUint64, ncclSum, 0, s->groups[gi]->comm, ctx[gi]->st[0]);
            }
            const ncclResult_t ne = rccl().GroupEnd();
            if (nr == ncclSuccess) nr = ne;
            if (nr != ncclSuccess) rc = fail(RSBWT_EHIP, "ncclReduce: %s", rccl().GetErrorString(nr));
            if (rc == RSBWT_OK) {
                (void)hipSetDevice(s->groups[0]->device);
                hipError_t e = hipMemcpyAsync(counts + q0, d_sum[0] + m, m * 8, hipMemcpyDeviceToHost, ctx[0]->st[0]);
                for (size_t gi = 0; gi < G && e == hipSuccess; ++gi) {
                    (void)hipSetDevice(s->groups[gi]->device);
                    e = hipStreamSynchronize(ctx[gi]->st[0]);
                }
                if (e != hipSuccess) rc = fail_hip(e, "count reduction");
            }
        } else if (rc == RSBWT_OK) {
            for (size_t i = 0; i < m; ++i) {
                uint64_t t = 0;
                for (size_t gi = 0; gi < G; ++gi) t += part[gi][i];
                counts[q0 + i] = t;
            }
        }
        for (size_t gi = 0; gi < G; ++gi)
            if (ctx[gi]) {
                (void)hipSetDevice(s->groups[gi]->device);
                (void)hipStreamSynchronize(ctx[gi]->st[0]);
                s->groups[gi]->pool.release(ctx[gi]);
            }
        if (rc) return rc;
    }
    return RSBWT_OK;
}
int rsbwt_set_count(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *counts) {
    return guarded("rsbwt_set_count", [&]() -> int { return rsbwt_set_count_body(s, kmers, Q, k, stride, counts); });
}


// Device-resident, for a set whose shards all sit on one device: one fused launch on `stream`;
// d_lower/d_upper: [num_shards][Q].  Nothing is synchronised.
int rsbwt_set_find_intervals_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                                 void *d_lower, void *d_upper, void *stream) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (s->groups.size() != 1) return fail(RSBWT_EINVAL, "the set spans %zu devices: device-resident calls need one", s->groups.size());
    dev_group *g = s->groups[0];
    int rc = use_device(g->device);
    if (rc) return rc;
    search_extra ex;
    ex.narrow = group_is_narrow(s, g, k);
    return search_launch(*g, g->d_views, (uint32_t)g->idx.size(), g->num_cus, d_packed, d_valid, Q, k, d_lower, d_upper,
                         false, (hipStream_t)stream, &ex);
}

int rsbwt_set_find_interval_pairs_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                                      void *d_pairs, void *stream) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (s->groups.size() != 1) return fail(RSBWT_EINVAL, "the set spans %zu devices: device-resident calls need one", s->groups.size());
    dev_group *g = s->groups[0];
    int rc = use_device(g->device);
    if (rc) return rc;
    search_extra ex;
    ex.pairs = true;
    ex.narrow = group_is_narrow(s, g, k);
    return search_launch(*g, g->d_views, (uint32_t)g->idx.size(), g->num_cus, d_packed, d_valid, Q, k, d_pairs, nullptr,
                         false, (hipStream_t)stream, &ex);
}

// The same search in two halves, so that a pipelined caller can compute the start records of batch i + 1 on a
// second stream while batch i is searched (they depend on the batch's k-mers and the k-mer tables only):
// rsbwt_set_prepare_dev fills d_records ([num_shards][Q] x 16 B) on `stream`; the caller orders it before
// rsbwt_set_find_interval_pairs_prepared_dev (an event, or the same stream), which runs the search kernel alone.
size_t rsbwt_set_records_bytes(const rsbwt_set_t *s, size_t Q) { return s ? s->shards.size() * Q * 16 : 0; }

int rsbwt_set_prepare_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t Q, uint32_t k, void *d_records,
                          void *stream) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (s->groups.size() != 1) return fail(RSBWT_EINVAL, "the set spans %zu devices: device-resident calls need one", s->groups.size());
    if (Q == 0) return RSBWT_OK;
    if (!d_packed || !d_valid || !d_records) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0 || k > 65535u) return fail(RSBWT_ERANGE, "k %u: 1..65535 symbols per k-mer", k);
    dev_group *g = s->groups[0];
    int rc = use_device(g->device);
    if (rc) return rc;
    const hipError_t e = launch_search_init(g->d_views, (uint32_t)g->idx.size(), d_packed, d_valid, Q, k, d_records, (hipStream_t)stream);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "start-record kernel launch");
}

int rsbwt_set_find_interval_pairs_prepared_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, const void *d_records,
                                               size_t Q, uint32_t k, void *d_pairs, void *stream) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (s->groups.size() != 1) return fail(RSBWT_EINVAL, "the set spans %zu devices: device-resident calls need one", s->groups.size());
    if (Q && !d_records) return fail(RSBWT_EINVAL, "null argument");
    dev_group *g = s->groups[0];
    int rc = use_device(g->device);
    if (rc) return rc;
    search_extra ex;
    ex.pairs = true;
    ex.d_init = d_records;
    ex.narrow = group_is_narrow(s, g, k);
    return search_launch(*g, g->d_views, (uint32_t)g->idx.size(), g->num_cus, d_packed, d_valid, Q, k, d_pairs, nullptr,
                         false, (hipStream_t)stream, &ex);
}

int rsbwt_set_count_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t Q, uint32_t k,
                        void *d_counts, void *stream) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (s->groups.size() != 1) return fail(RSBWT_EINVAL, "the set spans %zu devices: device-resident calls need one", s->groups.size());
    dev_group *g = s->groups[0];
    int rc = use_device(g->device);
    if (rc) return rc;
    search_extra ex;
    ex.narrow = group_is_narrow(s, g, k);
    return search_launch(*g, g->d_views, (uint32_t)g->idx.size(), g->num_cus, d_packed, d_valid, Q, k, d_counts, nullptr,
                         true, (hipStream_t)stream, &ex);
}

// Device-resident forms for a set on ONE device (one process per GPU drives its shards this way: bench.py).
// Three ways, best first: (1) all the shards in ONE traced and ONE resumed launch (fused_1mm_applies, below: tables of
// one depth); (2) else side by side; (3) at or above 2^26 variant searches per shard the shards take turns on the
// stream: each turn is a batch large enough to fill the GPU by itself.
// Below this many variant searches per shard (2^26 = 7e5 31-mers; a scratch per shard is what bounds it) the shards of
// the set work SIDE BY SIDE, each on a stream of its own with a scratch of its own (forked from and joined to the
// caller's stream with events), instead of taking turns: a shard's own sequence -- variants, start records, the traced
// and the resumed search, three small compaction launches -- leaves gaps and tails another shard's kernels fill
// (8 resident 20 GB shards, per batch: 4e4 31-mers 5.7 -> 4.3 ms, 1e5 9.4 -> 7.8, 4e5 30.7 -> 27.9:
// profiles/r03_set_side_by_side.json).
static size_t side_by_side_below() {
    static const size_t v = [] {
        const char *e = getenv("RSBWT_SET_1MM_SIDE_LOG2");  // A/B knob (tools/README.md)
        const int b = e ? atoi(e) : 0;
        return (size_t)1 << (b >= 10 && b <= 40 ? b : 26);
    }();
    return v;
}
#define SIDE_BY_SIDE_BELOW side_by_side_below()

// ONE launch sequence for all the shards of the device (below the same size): the k-mers traced in every shard by one
// fused launch (traces [S][m][tn]), their variants resumed in every shard by another (sparse results [S][mv], hit maps
// [S][hit_map_words(mv)]), one three-launch compaction with a segment per shard.  A (31-mer x shard) search is then
// one of S * m * (3k+1) in a launch that fills the GPU where a shard's own m * (3k+1) do not, and the set's tails
// are one tail.  Needs what the fused exact search needs -- one device -- and one trace length for all shards (k-mer
// tables of one depth: rsbwt_set_attach_ktabs gives every shard of a device the same).
// The table entries of the variants inside the tables' reach are read ahead for all shards by a kernel of their own
// (mm1_worklist.hip, wl_table_entries_kernel) instead of by the search kernel's lanes: the S shards' records of a variant
// are ONE stretch of the interleaved tables, which S waves on S XCDs otherwise fetch S times (round 5: 14.07 -> 13.60 ms
// per 4e5 31-mers x 8 shards, 1.74 -> 1.69 at 4e4; round 4 measured the same idea 2 % SLOWER -- the search kernel it
// relieved was then bound by instruction issue, not by requests).  RSBWT_SET_1MM_TABLE_PREPASS=0: A/B knob (tools/README.md)
static bool table_prepass() {
    static const bool on = [] {
        const char *e = getenv("RSBWT_SET_1MM_TABLE_PREPASS");
        return e ? atoi(e) != 0 : RSB_1MM_TABLE_PREPASS_DEFAULT;
    }();
    return on;
}
struct fused_1mm_layout {
    uint32_t tn;
    size_t trace, own, sparse, bits, blocks, total;  // byte sizes of the parts behind the variants, each 256-aligned
    // the hit lists by WORKLIST (mm1_worklist.hip): k <= 32 with something left of the tables' reach -- the variants'
    // searches are records of live searches (a record per variant at most), no variants spelled out, no start records
    bool worklist;
    size_t wl_cap, wl, counts;
    size_t pre;  // the table-part variants' entries read ahead for all shards (RSBWT_SET_1MM_TABLE_PREPASS): u64 [S][m * 3 (k - tn)], or 0
};
static bool fused_1mm_applies(const rsbwt_set_t *s, size_t m, uint32_t k, fused_1mm_layout *L, bool dense = false) {
    static const bool off = getenv("RSBWT_SET_1MM_UNFUSED") != nullptr;  // A/B knob (tools/README.md)
    const size_t S = s->shards.size(), mv = m * (3 * (size_t)k + 1);
    if (off || S < 2 || S > 1024 || s->groups.size() != 1 || mv >= SIDE_BY_SIDE_BELOW || k > 32767u) return false;  // (S: a grid row per shard)
    const uint32_t tn = trace_entries(s->shards[0]->view, k);
    for (const rsbwt_t *h : s->shards)
        if (trace_entries(h->view, k) != tn || h->view.n == 0) return false;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    L->tn = tn;
    L->sparse = al(S * mv * 16);
    L->bits = al(S * hit_map_words(mv) * 8);
    L->blocks = al(S * compact_hits_block_words(mv) * 8);
    static const bool no_worklist = getenv("RSBWT_SET_1MM_NO_WORKLIST") != nullptr;  // A/B knob (tools/README.md): round 3's launches
    static const bool no_walk = getenv("RSBWT_SET_1MM_NO_WALK") != nullptr;          // A/B knob: the traced launch + the branch kernel
    L->worklist = !no_worklist && tn > 0 && tn < k && k <= 32u && m * (size_t)tn < 0xFFFFFFFFull;
    // the trace ([S][m][tn] intervals) and the k-mers' own intervals are what the TRACED launch leaves: the default
    // worklist path walks the k-mers instead (search_solo.h, WALK) and touches neither (0.87 GB of scratch at 4e5
    // 31-mers x 8 shards that the tables can have: ADVICE r04)
    // (dense: rsbwt_set_find_intervals_1mm_dev's [S][m][3k+1] matrices -- always the traced and the resumed launch)
    const bool traced = dense || !L->worklist || no_walk;
    L->trace = traced ? al(S * m * (size_t)tn * 16) : 0;
    L->own = traced && tn ? al(S * m * 16) : 0;  // the k-mers' own intervals ({lower, upper}[S][m]: nobody reads them)
    L->wl_cap = L->worklist ? m * 3u * (size_t)tn : 0;  // (a record per variant substituted left of the tables' reach, at most)
    L->wl = L->worklist ? al(S * L->wl_cap * 32) : 0;
    L->counts = L->worklist ? al(S * 8 * (size_t)WL_COUNT_STRIDE) : 0;
    L->pre = (L->worklist && table_prepass() && S <= 256) ? al(S * m * 3u * (size_t)(k - tn) * 8) : 0;
    L->total = L->trace + L->own + L->sparse + L->bits + L->blocks + L->wl + L->counts + L->pre;
    return true;
}

// The two fused launches: the k-mers traced in every shard (when the tables leave something to trace), then their
// variants (expanded at d_var: variants_of_batch_dev's layout) resumed in every shard.  d_bits != nullptr: sparse
// results at d_lower + the hit maps; else dense [S][m][3k+1] lower and upper.
static int fused_1mm_launches(rsbwt_set_t *s, dev_group *g, const fused_1mm_layout &L, const void *d_packed, const void *d_valid,
                              size_t m, uint32_t k, const uint8_t *d_var, uint8_t *d_trace, uint8_t *d_own, void *d_lower,
                              void *d_upper, void *d_bits, hipStream_t st) {
    const uint32_t S = (uint32_t)s->shards.size();
    const size_t V = 3 * (size_t)k + 1, mv = m * V;
    const uint8_t *d_vok = d_var + ((mv * ((k + 31u) / 32u) * 8 + 15) & ~(size_t)15);
    search_extra resumed;
    if (L.tn) {
        search_extra traced;
        traced.d_trace_out = d_trace;
        traced.trace_n = L.tn;
        traced.pairs = true;
        const int rc = search_launch(*g, g->d_views, S, g->num_cus, d_packed, d_valid, m, k, d_own, nullptr, false, st, &traced);
        if (rc) return rc;
        resumed.d_trace_in = d_trace;
        resumed.trace_n = L.tn;
        resumed.variants = (uint32_t)V;
    }
    resumed.d_hit_bits = d_bits;
    return search_launch(*g, g->d_views, S, g->num_cus, d_var, d_vok, mv, k, d_lower, d_upper, false, st, &resumed);
}

// 1-mismatch search over the shards of a one-device set: the fused launches above where they apply (dense
// results: the resumed launch writes [S][m][3k+1] lower and upper itself), else the shards take turns on the stream,
// each turn m x (3k+1) searches of one shard.
size_t rsbwt_set_1mm_scratch_bytes(const rsbwt_set_t *s, size_t m, uint32_t k) {
    size_t need = 0;
    if (!s) return 0;
    for (rsbwt_t *h : s->shards) need = std::max(need, rsbwt_1mm_scratch_bytes(h, m, k));
    fused_1mm_layout L;
    if (fused_1mm_applies(s, m, k, &L, true)) need = std::max(need, ((variants_bytes(m, k) + 255) & ~(size_t)255) + L.trace + L.own);
    return need;
}

static int rsbwt_set_find_intervals_1mm_dev_body(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t m, uint32_t k,
                                     void *d_lower, void *d_upper, void *d_scratch, void *stream) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (s->groups.size() != 1) return fail(RSBWT_EINVAL, "the set spans %zu devices: device-resident calls need one", s->groups.size());
    if (m == 0) return RSBWT_OK;
    if (!d_lower || !d_upper) return fail(RSBWT_EINVAL, "null argument");
    const size_t row = m * (3 * (size_t)k + 1) * 8;
    fused_1mm_layout L;
    if (fused_1mm_applies(s, m, k, &L, true)) {
        if (!d_packed || !d_valid || !d_scratch) return fail(RSBWT_EINVAL, "null argument");
        dev_group *g = s->groups[0];
        int rc = use_device(g->device);
        if (rc) return rc;
        uint8_t *d_var = (uint8_t *)d_scratch, *d_trace = d_var + ((variants_bytes(m, k) + 255) & ~(size_t)255), *d_own = d_trace + L.trace;
        if ((rc = variants_of_batch_dev(d_packed, d_valid, m, k, d_var, (hipStream_t)stream)) != RSBWT_OK) return rc;
        return fused_1mm_launches(s, g, L, d_packed, d_valid, m, k, d_var, d_trace, d_own, d_lower, d_upper, nullptr, (hipStream_t)stream);
    }
    for (size_t i = 0; i < s->shards.size(); ++i) {
        const int rc = rsbwt_find_intervals_1mm_dev(s->shards[i], d_packed, d_valid, m, k, (uint8_t *)d_lower + i * row,
                                                    (uint8_t *)d_upper + i * row, d_scratch, stream);
        if (rc) return rc;
    }
    return RSBWT_OK;
}
int rsbwt_set_find_intervals_1mm_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t m, uint32_t k,
                                     void *d_lower, void *d_upper, void *d_scratch, void *stream) {
    return guarded("rsbwt_set_find_intervals_1mm_dev", [&]() -> int { return rsbwt_set_find_intervals_1mm_dev_body(s, d_packed, d_valid, m, k, d_lower, d_upper, d_scratch, stream); });
}


// Gathers per-device interval blocks onto the first device of the set over RCCL (xGMI): d_blocks[g]
// = device g's [S_g][Q] x {lower, upper} block of `bytes[g]` bytes on that device; d_root on device 0
// receives them back to back in device order.  One call per batch; nothing is synchronised beyond
// the streams given (streams[g] on device g).  With one device it is a device-to-device copy.
static int rsbwt_set_gather_intervals_dev_body(rsbwt_set_t *s, const void *const *d_blocks, const size_t *bytes, void *d_root,
                                   void *const *streams) {
    if (!s || !d_blocks || !bytes || !d_root || !streams) return fail(RSBWT_EINVAL, "null argument");
    const size_t G = s->groups.size();
    (void)hipSetDevice(s->groups[0]->device);
    HIP_OK(hipMemcpyAsync(d_root, d_blocks[0], bytes[0], hipMemcpyDeviceToDevice, (hipStream_t)streams[0]));
    if (G == 1) return RSBWT_OK;
    size_t off = bytes[0];
    std::lock_guard<std::mutex> comm_lock(s->comm_mu);
    if (!ensure_comms(s)) {
        // No RCCL (no librccl on the box, or groups that share a GPU): the root PULLS every block with a peer copy
        // on its own stream, behind an event the block's stream records ("ready"), and tells that stream when the
        // block has been read ("done") -- the stream order ncclSend / ncclRecv would have given.  xGMI peer copies
        // by the copy engines; slower to start than one grouped RCCL call, the same bytes over the same links.
        static const bool quiet = getenv("RSBWT_QUIET") != nullptr;
        static bool said = false;
        if (!said && !quiet) {
            said = true;
            fprintf(stderr, "rsbwt: interval gather over peer copies (%s)\n", rccl().ok ? "device groups share a GPU" : "librccl not found");
        }
        for (size_t g = 1; g < G; ++g) {
            dev_group *grp = s->groups[g];
            (void)hipSetDevice(grp->device);
            if (!grp->peer_ready) HIP_OK(hipEventCreateWithFlags(&grp->peer_ready, hipEventDisableTiming));
            if (!grp->peer_done) HIP_OK(hipEventCreateWithFlags(&grp->peer_done, hipEventDisableTiming));
            HIP_OK(hipEventRecord(grp->peer_ready, (hipStream_t)streams[g]));
            (void)hipSetDevice(s->groups[0]->device);
            HIP_OK(hipStreamWaitEvent((hipStream_t)streams[0], grp->peer_ready, 0));
            HIP_OK(hipMemcpyPeerAsync((uint8_t *)d_root + off, s->groups[0]->device, d_blocks[g], grp->device, bytes[g], (hipStream_t)streams[0]));
            HIP_OK(hipEventRecord(grp->peer_done, (hipStream_t)streams[0]));
            (void)hipSetDevice(grp->device);
            HIP_OK(hipStreamWaitEvent((hipStream_t)streams[g], grp->peer_done, 0));
            off += bytes[g];
        }
        return RSBWT_OK;
    }
    ncclResult_t nr = rccl().GroupStart();
    for (size_t g = 1; g < G && nr == ncclSuccess; ++g) {
        (void)hipSetDevice(s->groups[g]->device);
        nr = rccl().Send(d_blocks[g], bytes[g], ncclUint8, 0, s->groups[g]->comm, (hipStream_t)streams[g]);
        if (nr != ncclSuccess) break;
        (void)hipSetDevice(s->groups[0]->device);
        nr = rccl().Recv((uint8_t *)d_root + off, bytes[g], ncclUint8, (int)g, s->groups[0]->comm, (hipStream_t)streams[0]);
        off += bytes[g];
    }
    const ncclResult_t ne = rccl().GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return fail(RSBWT_EHIP, "RCCL gather: %s", rccl().GetErrorString(nr));
    return RSBWT_OK;
}
int rsbwt_set_gather_intervals_dev(rsbwt_set_t *s, const void *const *d_blocks, const size_t *bytes, void *d_root,
                                   void *const *streams) {
    return guarded("rsbwt_set_gather_intervals_dev", [&]() -> int { return rsbwt_set_gather_intervals_dev_body(s, d_blocks, bytes, d_root, streams); });
}


// ---- configs[3] / configs[4] over a set that may span devices -------------------------------------------
// The reference's front-end sends every request to every partition and CONCATENATES the per-partition
// read lists (src/service/server.cpp:124,199-261); a partition answers with what ITS BWT holds.  So the
// set-level forms of the 1-mismatch search and of locate + extract are per-shard results laid side by
// side: the devices work concurrently (one host thread each), a device's shards take turns on it, and
// what crosses PCIe is each device's own lists / reads, straight to the caller's host buffers.

// every shard's rsbwt_hits_1mm list, in shard order: hits[first[i] .. first[i+1]) are shard i's
static int rsbwt_set_hits_1mm_body(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride,
                                   rsbwt_hit_1mm *hits, size_t cap, uint64_t *first, size_t *nhits) {
    if (!s || !nhits || !first) return fail(RSBWT_EINVAL, "null argument");
    const size_t S = s->shards.size();
    *nhits = 0;
    for (size_t i = 0; i <= S; ++i) first[i] = 0;
    if (Q == 0) return RSBWT_OK;
    if (!kmers || (!hits && cap)) return fail(RSBWT_EINVAL, "null argument");
    std::vector<std::vector<rsbwt_hit_1mm>> part(S);
    int rc = for_each_group(s, [&](size_t gi) -> int {
        dev_group *g = s->groups[gi];
        for (size_t i : g->idx) {
            std::vector<rsbwt_hit_1mm> &v = part[i];
            v.resize(std::max<size_t>(4 * Q, 1024));
            size_t n = 0;
            int r = rsbwt_hits_1mm(s->shards[i], kmers, Q, k, stride, v.data(), v.size(), &n);
            if (r == RSBWT_ERANGE && n > v.size()) {  // (k out of range also says ERANGE: n stays 0)
                v.resize(n);
                r = rsbwt_hits_1mm(s->shards[i], kmers, Q, k, stride, v.data(), v.size(), &n);
            }
            if (r) return r;
            v.resize(n);
        }
        return RSBWT_OK;
    });
    if (rc) return rc;
    size_t total = 0;
    for (size_t i = 0; i < S; ++i) {
        first[i] = total;
        total += part[i].size();
    }
    first[S] = total;
    *nhits = total;
    if (total > cap) return fail(RSBWT_ERANGE, "%zu hits over the set, room for %zu", total, cap);
    for (size_t i = 0; i < S; ++i)
        if (!part[i].empty()) memcpy(hits + first[i], part[i].data(), part[i].size() * sizeof(rsbwt_hit_1mm));
    return RSBWT_OK;
}
int rsbwt_set_hits_1mm(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride, rsbwt_hit_1mm *hits,
                       size_t cap, uint64_t *first, size_t *nhits) {
    return guarded("rsbwt_set_hits_1mm", [&]() -> int { return rsbwt_set_hits_1mm_body(s, kmers, Q, k, stride, hits, cap, first, nhits); });
}

// The group's shard views WITH the shards' select samples, for the fused extraction (extract_lines.hip: ONE launch sequence
// walks the rows of every shard of the group): an array of its own (d_xviews), made on the first call from the shards'
// extraction views -- the array the searches read (d_views) is not touched, so an extraction may start beside searches
// on the same set.
static int ensure_group_xviews(rsbwt_set_t *s, dev_group *g, hipStream_t stream) {
    if (g->xviews_ready.load(std::memory_order_acquire)) return RSBWT_OK;
    std::lock_guard<std::mutex> lock(s->mu);
    if (g->xviews_ready.load(std::memory_order_relaxed)) return RSBWT_OK;
    std::vector<shard_view> v;
    for (size_t i : g->idx) {
        rsbwt_t *h = s->shards[i];
        if (h->view.n == 0) return fail(RSBWT_EINVAL, "empty index");
        const int rc = ensure_samples(h, stream);
        if (rc != RSBWT_OK) return rc;
        v.push_back(h->xview);
    }
    shard_view *dx = nullptr;
    HIP_OK(hipMalloc(&dx, v.size() * sizeof(shard_view)));
    const hipError_t e = hipMemcpy(dx, v.data(), v.size() * sizeof(shard_view), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(dx);
        return fail_hip(e, "publishing the extraction views");
    }
    g->d_xviews = dx;
    g->xviews_ready.store(true, std::memory_order_release);
    return RSBWT_OK;
}

// row i = SA row rows[i] of shard shard_of[i]
static int rsbwt_set_extract_body(rsbwt_set_t *s, const uint32_t *shard_of, const uint64_t *rows, size_t n, char *out,
                                  uint32_t stride, uint32_t *len, uint32_t *prefix_len) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (n == 0) return RSBWT_OK;
    if (!shard_of || !rows || !out) return fail(RSBWT_EINVAL, "null argument");
    const size_t S = s->shards.size();
    std::vector<std::vector<size_t>> where(S);
    for (size_t i = 0; i < n; ++i) {
        if (shard_of[i] >= S) return fail(RSBWT_EINVAL, "row %zu names shard %u of %zu", i, shard_of[i], S);
        where[shard_of[i]].push_back(i);
    }
    static const bool turns_only = getenv("RSBWT_SET_EXTRACT_TURNS") != nullptr;  // A/B knob (tools/README.md): a launch sequence per shard
    return for_each_group(s, [&](size_t gi) -> int {
        dev_group *g = s->groups[gi];
        // The rows of all the group's shards in ONE launch sequence (the fused extraction bench.py --mode extract times):
        // [S_g][nmax] rows, a shard with fewer padded with rows past any index (they end at once).  What a window of the
        // service loop asks for -- a few hundred rows spread over the partitions -- then costs one walk's latency, not
        // one per partition (round 5; until then a launch sequence and a copy back per shard).
        const size_t Sg = g->idx.size();
        size_t nmax = 0, ng = 0;
        for (size_t si : g->idx) {
            nmax = std::max(nmax, where[si].size());
            ng += where[si].size();
        }
        if (ng == 0) return RSBWT_OK;
        if (!turns_only && Sg > 1 && nmax < (1ull << 31) && Sg * nmax <= 4 * ng + 4096) {
            int rc = use_device(g->device);
            if (rc) return rc;
            call_ctx *c = g->pool.acquire();
            if (!c) return fail(RSBWT_EHIP, "cannot create a HIP stream");
            struct release_t {
                dev_group *g;
                call_ctx *c;
                ~release_t() { g->pool.release(c); }
            } release{g, c};
            hipStream_t st = c->st[0];
            if ((rc = ensure_group_xviews(s, g, st)) != RSBWT_OK) return rc;
            const size_t cells = Sg * nmax;
            const size_t a_rows = (cells * 8 + 255) & ~(size_t)255, a_out = (cells * (size_t)stride + 255) & ~(size_t)255, a_u32 = (cells * 4 + 255) & ~(size_t)255;
            if ((rc = c->stage(a_rows + a_out + 2 * a_u32)) != RSBWT_OK) return rc;
            uint8_t *d_rows = (uint8_t *)c->d_stage, *d_out = d_rows + a_rows, *d_len = d_out + a_out, *d_pl = d_len + a_u32;
            std::vector<uint64_t> hr(cells, ~0ull);
            for (size_t j = 0; j < Sg; ++j) {
                const std::vector<size_t> &w = where[g->idx[j]];
                for (size_t t = 0; t < w.size(); ++t) hr[j * nmax + t] = rows[w[t]];
            }
            HIP_OK(hipMemcpyAsync(d_rows, hr.data(), cells * 8, hipMemcpyHostToDevice, st));
            const hipError_t e = launch_extract_wave(g->scratch, g->d_xviews, (uint32_t)Sg, d_rows, nmax, d_out, stride, d_pl, d_len, g->num_cus, st, nullptr);
            if (e != hipSuccess) return fail_hip(e, "extract kernel launch");
            std::vector<char> ho(cells * (size_t)stride);
            std::vector<uint32_t> hl(cells), hp(cells);
            HIP_OK(hipMemcpyAsync(ho.data(), d_out, cells * (size_t)stride, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(hl.data(), d_len, cells * 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(hp.data(), d_pl, cells * 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            for (size_t j = 0; j < Sg; ++j) {
                const std::vector<size_t> &w = where[g->idx[j]];
                for (size_t t = 0; t < w.size(); ++t) {
                    const size_t cell = j * nmax + t;
                    const uint32_t keep = hl[cell] == 0xFFFFFFFFu ? 0u : hl[cell];
                    if (keep) memcpy(out + w[t] * (size_t)stride, ho.data() + cell * (size_t)stride, keep);
                    if (len) len[w[t]] = hl[cell];
                    if (prefix_len) prefix_len[w[t]] = hp[cell];
                }
            }
            return RSBWT_OK;
        }
        for (size_t si : g->idx) {
            const std::vector<size_t> &w = where[si];
            if (w.empty()) continue;
            const size_t m = w.size();
            std::vector<uint64_t> r(m);
            std::vector<char> o(m * (size_t)stride);
            std::vector<uint32_t> ln(m), pl(m);
            for (size_t j = 0; j < m; ++j) r[j] = rows[w[j]];
            const int rc = rsbwt_extract(s->shards[si], r.data(), m, o.data(), stride, ln.data(), pl.data());
            if (rc) return rc;
            for (size_t j = 0; j < m; ++j) {
                const uint32_t keep = ln[j] == 0xFFFFFFFFu ? 0u : ln[j];
                if (keep) memcpy(out + w[j] * (size_t)stride, o.data() + j * (size_t)stride, keep);
                if (len) len[w[j]] = ln[j];
                if (prefix_len) prefix_len[w[j]] = pl[j];
            }
        }
        return RSBWT_OK;
    });
}
int rsbwt_set_extract(rsbwt_set_t *s, const uint32_t *shard_of, const uint64_t *rows, size_t n, char *out, uint32_t stride,
                      uint32_t *len, uint32_t *prefix_len) {
    return guarded("rsbwt_set_extract", [&]() -> int { return rsbwt_set_extract_body(s, shard_of, rows, n, out, stride, len, prefix_len); });
}

// what rsbwt_set_query makes of the batch's intervals lo / up [S][Q]: first[], and the rows' reads in the caller's order
static int set_query_rows(rsbwt_set_t *s, size_t Q, const std::vector<uint64_t> &lo, const std::vector<uint64_t> &up, uint64_t *first,
                          uint32_t *read_shard, char *reads, uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads) {
    const size_t S = s->shards.size();
    // rows of shard i for k-mer q: its interval, if it is one of rows of that shard (rsbwt_query's rule: capi.hip, interval_rows)
    auto width = [&](size_t i, size_t q) -> uint64_t {
        const uint64_t l = lo[i * Q + q], u = up[i * Q + q];
        return (l <= u && u < s->shards[i]->view.n) ? u - l + 1 : 0;
    };
    size_t total = 0;
    for (size_t q = 0; q < Q; ++q) {
        first[q] = total;
        for (size_t i = 0; i < S; ++i) total += (size_t)width(i, q);
    }
    first[Q] = total;
    *nreads = total;
    if (total > cap_reads) return fail(RSBWT_ERANGE, "%zu reads over the set, room for %zu", total, cap_reads);
    if (total == 0) return RSBWT_OK;
    if (!reads || !read_len) return fail(RSBWT_EINVAL, "null argument");
    std::vector<uint32_t> shard_of(total);
    std::vector<uint64_t> rows(total);
    size_t at = 0;
    for (size_t q = 0; q < Q; ++q)
        for (size_t i = 0; i < S; ++i) {
            const uint64_t w = width(i, q), l = lo[i * Q + q];
            for (uint64_t r = 0; r < w; ++r, ++at) {
                shard_of[at] = (uint32_t)i;
                rows[at] = l + r;
            }
        }
    if (read_shard) memcpy(read_shard, shard_of.data(), total * sizeof(uint32_t));
    return rsbwt_set_extract_body(s, shard_of.data(), rows.data(), total, reads, read_stride, read_len, nullptr);
}

// query() of every shard (query.cpp:87-100), k-mer by k-mer: k-mer q's reads are first[q] .. first[q+1],
// shard 0's first (each shard's in SA-row order), read_shard[r] naming the shard read r came from
static int rsbwt_set_query_body(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *first,
                                uint32_t *read_shard, char *reads, uint32_t read_stride, uint32_t *read_len, size_t cap_reads,
                                size_t *nreads) {
    if (!s || !nreads || (!first && Q)) return fail(RSBWT_EINVAL, "null argument");
    *nreads = 0;
    if (Q == 0) return RSBWT_OK;
    if (!kmers) return fail(RSBWT_EINVAL, "null argument");
    if (read_stride == 0) return fail(RSBWT_EINVAL, "read_stride must be positive");
    const size_t S = s->shards.size();
    if (k == 0) {  // (rsbwt_query: no rows)
        for (size_t q = 0; q <= Q; ++q) first[q] = 0;
        return RSBWT_OK;
    }
    // ONE search of the batch over all the shards (a fused launch per device), then ONE extraction of the intervals' rows
    // addressed as (shard, row) in the order the caller gets them.  (Until round 5 every shard ran rsbwt_query by itself,
    // twice -- to size, then to fetch: 2 S searches and S extractions, each a launch sequence with a copy back of its
    // own; a window of the service loop made one such call per distinct query length.)
    std::vector<uint64_t> lo(S * Q), up(S * Q);
    const int rc = rsbwt_set_find_intervals_body(s, kmers, Q, k, stride, lo.data(), up.data());
    if (rc) return rc;
    return set_query_rows(s, Q, lo, up, first, read_shard, reads, read_stride, read_len, cap_reads, nreads);
}
// the same for queries of lengths of their own (rsbwt_set_find_intervals_var): one call answers a window of the service loop
int rsbwt_set_query_var(rsbwt_set_t *s, const char *text, const uint64_t *off, size_t Q, uint64_t *first, uint32_t *read_shard,
                        char *reads, uint32_t read_stride, uint32_t *read_len, size_t cap_reads, size_t *nreads) {
    return guarded("rsbwt_set_query_var", [&]() -> int {
        if (!s || !nreads || (!first && Q)) return fail(RSBWT_EINVAL, "null argument");
        *nreads = 0;
        if (Q == 0) return RSBWT_OK;
        if (read_stride == 0) return fail(RSBWT_EINVAL, "read_stride must be positive");
        const size_t S = s->shards.size();
        std::vector<uint64_t> lo(S * Q), up(S * Q);
        const int rc = rsbwt_set_find_intervals_var_body(s, text, off, Q, lo.data(), up.data(), false);
        if (rc) return rc;
        return set_query_rows(s, Q, lo, up, first, read_shard, reads, read_stride, read_len, cap_reads, nreads);
    });
}
int rsbwt_set_query(rsbwt_set_t *s, const char *kmers, size_t Q, uint32_t k, size_t stride, uint64_t *first,
                    uint32_t *read_shard, char *reads, uint32_t read_stride, uint32_t *read_len, size_t cap_reads,
                    size_t *nreads) {
    return guarded("rsbwt_set_query", [&]() -> int {
        return rsbwt_set_query_body(s, kmers, Q, k, stride, first, read_shard, reads, read_stride, read_len, cap_reads, nreads);
    });
}

static size_t hits_1mm_scratch_one(const rsbwt_set_t *s, size_t m, uint32_t k) {
    size_t need = 0;
    for (rsbwt_t *h : s->shards) need = std::max(need, rsbwt_hits_1mm_scratch_bytes(h, m, k));
    return (need + 255) & ~(size_t)255;
}

// scratch of a set's hit-list search: the batch's variants, expanded ONCE for all shards (they depend on the k-mers
// alone), then a slot per shard that works at the same time (or the fused launches' parts)
size_t rsbwt_set_hits_1mm_scratch_bytes(const rsbwt_set_t *s, size_t m, uint32_t k) {
    if (!s) return 0;
    const size_t one = hits_1mm_scratch_one(s, m, k);
    const bool side = s->shards.size() > 1 && m * (3 * (size_t)k + 1) < SIDE_BY_SIDE_BELOW;
    const size_t slots = side ? std::min<size_t>(s->shards.size(), dev_group::FORK) : 1;
    fused_1mm_layout L;
    const size_t fused = fused_1mm_applies(s, m, k, &L) ? L.total : 0;
    return ((variants_bytes(m, k) + 255) & ~(size_t)255) + std::max(one * slots, fused);
}

// 1: rsbwt_set_hits_1mm_dev of m k-mers runs as the fused launches (for a caller that prices them: bench.py)
int rsbwt_set_hits_1mm_is_fused(const rsbwt_set_t *s, size_t m, uint32_t k) {
    fused_1mm_layout L;
    static const bool turns_only = getenv("RSBWT_SET_1MM_TURNS") != nullptr;
    return s && m && k && !turns_only && fused_1mm_applies(s, m, k, &L) ? 1 : 0;
}

static int set_hits_1mm_fused(rsbwt_set_t *s, dev_group *g, const fused_1mm_layout &L, const void *d_packed, const void *d_valid,
                              size_t m, uint32_t k, void *d_hits, size_t cap_per_shard, void *d_totals, const uint8_t *d_var,
                              uint8_t *d_parts, hipStream_t st) {
    const uint32_t S = (uint32_t)s->shards.size();
    const size_t mv = m * (3 * (size_t)k + 1);
    uint8_t *d_trace = d_parts, *d_own = d_trace + L.trace, *d_sparse = d_own + L.own, *d_bits = d_sparse + L.sparse;
    uint8_t *d_blocks = d_bits + L.bits;
    HIP_OK(hipMemsetAsync(d_bits, 0, (size_t)S * hit_map_words(mv) * 8, st));
    int rc;
    if (L.worklist) {
        // the k-mers traced; the step of the three substitutions of every traced position off one fetch, the variants
        // inside the tables' reach from their table entries: worklists of live searches; then ONE launch runs them
        // (d_pre: the table entries of the variants inside the tables' reach, read ahead for all shards by the worklist
        // launch's first kernel: mm1_worklist.hip, wl_table_entries_kernel)
        uint8_t *d_wl = d_blocks + L.blocks, *d_counts = d_wl + L.wl, *d_pre = L.pre ? d_counts + L.counts : nullptr;
        static const bool no_walk = getenv("RSBWT_SET_1MM_NO_WALK") != nullptr;  // A/B knob (tools/README.md): the traced launch + the branch kernel
        if (!no_walk) {
            // ONE walk of the k-mers: their own searches, and at every position left of the tables' reach the step of the
            // three substitutions off the same fetch -- the survivors appended to the worklists (search_solo.h, WALK)
            HIP_OK(hipMemsetAsync(d_counts, 0, L.counts, st));
            if ((rc = search_launch_walk(*g, g->d_views, S, g->num_cus, d_packed, d_valid, m, L.tn, d_wl, d_counts, L.wl_cap, k, d_sparse, d_bits, st)) != RSBWT_OK) return rc;
        } else {
            search_extra traced;
            traced.d_trace_out = d_trace;
            traced.trace_n = L.tn;
            traced.pairs = true;
            if ((rc = search_launch(*g, g->d_views, S, g->num_cus, d_packed, d_valid, m, k, d_own, nullptr, false, st, &traced)) != RSBWT_OK) return rc;
            const hipError_t ew = launch_mm1_worklists(g->d_views, S, d_packed, d_valid, m, k, L.tn, d_trace, d_own, d_wl, L.wl_cap, d_counts,
                                                       d_sparse, d_bits, g->num_cus, st, g->counting ? g->d_work : nullptr);
            if (ew != hipSuccess) return fail_hip(ew, "worklist kernels");
        }
        rc = search_launch_worklist(*g, g->d_views, S, g->num_cus, d_packed, d_valid, m, L.tn, d_wl, d_counts, L.wl_cap, k, d_sparse, d_bits, st, d_pre);
    } else {
        rc = fused_1mm_launches(s, g, L, d_packed, d_valid, m, k, d_var, d_trace, d_own, d_sparse, nullptr, d_bits, st);
    }
    if (rc) return rc;
    const hipError_t e = launch_compact_hits(d_bits, d_sparse, mv, d_hits, cap_per_shard, d_totals, d_blocks, st, S);
    return e == hipSuccess ? RSBWT_OK : fail_hip(e, "hit list kernels");
}

// d_hits: [num_shards][cap_per_shard] records of 32 B (rsbwt_hits_1mm_dev's), d_totals: u64[num_shards]
int rsbwt_set_hits_1mm_dev(rsbwt_set_t *s, const void *d_packed, const void *d_valid, size_t m, uint32_t k, void *d_hits,
                           size_t cap_per_shard, void *d_totals, void *d_scratch, void *stream) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (s->groups.size() != 1) return fail(RSBWT_EINVAL, "the set spans %zu devices: device-resident calls need one", s->groups.size());
    if (!d_totals || (!d_hits && cap_per_shard)) return fail(RSBWT_EINVAL, "null argument");
    dev_group *g = s->groups[0];
    const size_t S = s->shards.size();
    int rc = use_device(g->device);
    if (rc) return rc;
    if (m == 0) {
        HIP_OK(hipMemsetAsync(d_totals, 0, 8 * S, (hipStream_t)stream));
        return RSBWT_OK;
    }
    if (!d_packed || !d_valid || !d_scratch) return fail(RSBWT_EINVAL, "null argument");
    if (k == 0) return fail(RSBWT_EINVAL, "k must be at least 1");
    static const bool turns_only = getenv("RSBWT_SET_1MM_TURNS") != nullptr;  // A/B knob (tools/README.md)
    const bool side = !turns_only && S > 1 && m * (3 * (size_t)k + 1) < SIDE_BY_SIDE_BELOW;
    const size_t one = hits_1mm_scratch_one(s, m, k);
    uint8_t *d_var = (uint8_t *)d_scratch, *d_slots = d_var + ((variants_bytes(m, k) + 255) & ~(size_t)255);
    fused_1mm_layout FL;
    const bool fused = !turns_only && fused_1mm_applies(s, m, k, &FL);
    // (the worklist form spells no variants out)
    if (!(fused && FL.worklist) && (rc = variants_of_batch_dev(d_packed, d_valid, m, k, d_var, (hipStream_t)stream)) != RSBWT_OK) return rc;
    if (fused)
        return set_hits_1mm_fused(s, g, FL, d_packed, d_valid, m, k, d_hits, cap_per_shard, d_totals, d_var, d_slots, (hipStream_t)stream);
    auto one_shard = [&](size_t i, uint8_t *slot, hipStream_t st) {
        return hits_1mm_dev_shared(s->shards[i], d_packed, d_valid, m, k, (uint8_t *)d_hits + i * cap_per_shard * 32, cap_per_shard,
                                   (uint8_t *)d_totals + i * 8, slot, st, d_var);
    };
    if (!side) {
        for (size_t i = 0; i < S; ++i)
            if ((rc = one_shard(i, d_slots, (hipStream_t)stream)) != RSBWT_OK) return rc;
        return RSBWT_OK;
    }
    std::lock_guard<std::mutex> lock(g->fork_mu);  // one fork / join sequence at a time uses the side streams' events
    if ((rc = g->ensure_fork()) != RSBWT_OK) return rc;
    HIP_OK(hipEventRecord(g->fork_ev, (hipStream_t)stream));  // (behind the variants)
    // (a failure inside the loop leaves it through `rc`, never by returning: what was already enqueued on the side
    // streams reads and writes the caller's buffers and is joined below before the caller's stream goes on)
    auto hip_rc = [](hipError_t e, const char *what) { return e == hipSuccess ? RSBWT_OK : fail_hip(e, what); };
    for (size_t i = 0; i < S && rc == RSBWT_OK; ++i) {
        hipStream_t st = g->fork_st[i % dev_group::FORK];
        if ((rc = hip_rc(hipStreamWaitEvent(st, g->fork_ev, 0), "hipStreamWaitEvent")) != RSBWT_OK) break;
        if ((rc = one_shard(i, d_slots + (i % dev_group::FORK) * one, st)) != RSBWT_OK) break;
        if (i + dev_group::FORK >= S) {  // the last shard of each side stream: its event joins the caller's stream
            if ((rc = hip_rc(hipEventRecord(g->join_ev[i % dev_group::FORK], st), "hipEventRecord")) != RSBWT_OK) break;
            rc = hip_rc(hipStreamWaitEvent((hipStream_t)stream, g->join_ev[i % dev_group::FORK], 0), "hipStreamWaitEvent");
        }
    }
    if (rc) {  // whatever was enqueued still finishes before the caller's stream goes on
        for (int i = 0; i < dev_group::FORK; ++i) {
            if (hipEventRecord(g->join_ev[i], g->fork_st[i]) == hipSuccess) (void)hipStreamWaitEvent((hipStream_t)stream, g->join_ev[i], 0);
        }
    }
    return rc;
}

// d_rows: [num_shards][n] SA rows (row numbers are per shard); d_out [num_shards][n][stride], d_len / d_prefix_len [num_shards][n].
// ONE launch sequence walks the rows of all the shards (extract_lines.hip): a walk kernel ends in a tail as long as its
// longest walk, and a launch per shard -- round 3: side by side on streams of the set -- has a tail per shard and a
// fraction of the rows per lane.
int rsbwt_set_extract_dev(rsbwt_set_t *s, const void *d_rows, size_t n, void *d_out, uint32_t stride, void *d_len,
                          void *d_prefix_len, void *stream) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    if (s->groups.size() != 1) return fail(RSBWT_EINVAL, "the set spans %zu devices: device-resident calls need one", s->groups.size());
    if (n == 0) return RSBWT_OK;
    if (!d_rows || !d_out || !d_len || !d_prefix_len) return fail(RSBWT_EINVAL, "null argument");
    if (stride == 0) return fail(RSBWT_EINVAL, "stride must be positive");
    dev_group *g = s->groups[0];
    const size_t S = s->shards.size();
    static const bool turns_only = getenv("RSBWT_SET_EXTRACT_TURNS") != nullptr;  // A/B knob (tools/README.md): a launch sequence per shard
    if (turns_only || n > (1ull << 31)) {
        for (size_t i = 0; i < S; ++i) {
            const int rc = rsbwt_extract_dev(s->shards[i], (const uint8_t *)d_rows + i * n * 8, n, (uint8_t *)d_out + i * n * (size_t)stride,
                                             stride, (uint8_t *)d_len + i * n * 4, (uint8_t *)d_prefix_len + i * n * 4, stream);
            if (rc) return rc;
        }
        return RSBWT_OK;
    }
    int rc = use_device(g->device);
    if (rc) return rc;
    if ((rc = ensure_group_xviews(s, g, (hipStream_t)stream)) != RSBWT_OK) return rc;
    unsigned long long *work = nullptr;
    if (g->counting) {  // rsbwt_set_set_counting: the walk kernels' counters over all shards, read with rsbwt_set_last_search_counters
        work = g->d_work;
        HIP_OK(hipMemsetAsync(work, 0, WORK_WORDS * sizeof(unsigned long long), (hipStream_t)stream));
    }
    const hipError_t e = launch_extract_wave(g->scratch, g->d_xviews, (uint32_t)S, d_rows, n, d_out, stride, d_prefix_len, d_len, g->num_cus,
                                             (hipStream_t)stream, work);
    if (e != hipSuccess) return fail_hip(e, "extract kernel launch");
    return RSBWT_OK;
}

int rsbwt_rccl_available(void) { return rccl().ok ? 1 : 0; }

// measurement hooks of the set's first device group (bench.py)
int rsbwt_set_set_counting(rsbwt_set_t *s, int on) {
    if (!s) return fail(RSBWT_EINVAL, "null set");
    for (dev_group *g : s->groups) {
        std::lock_guard<std::mutex> lock(g->mu);
        g->counting = on != 0;
    }
    return RSBWT_OK;
}

int rsbwt_set_search_history_ms(rsbwt_set_t *s, float *ms, size_t cap, size_t *count) {
    if (!s || (!ms && cap) || !count) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(s->groups[0]->device);
    if (rc) return rc;
    return meter_history_ms(*s->groups[0], ms, cap, count);
}

int rsbwt_set_last_search_counters(rsbwt_set_t *s, uint64_t *words16) {
    if (!s || !words16) return fail(RSBWT_EINVAL, "null argument");
    int rc = use_device(s->groups[0]->device);
    if (rc) return rc;
    return meter_work(*s->groups[0], words16, WORK_WORDS);
}

}  // extern "C"

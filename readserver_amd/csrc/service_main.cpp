// service_main.cpp -- `rsbwt_service <service.cfg>`: the GPU twin of ReadServer's `service` process for
// the BWT-only paths (src/service/service.cpp:1366-1583).  Reads the same configuration file, loads the
// BWT(s) into HBM, connects the same three sockets and answers CountReads, ExactMatch-Count and ExactMatch-Reads
// requests in micro-batches (`GET /get?output=count|reads`); requests of other types are left unanswered (they
// belong to the RocksDB-backed paths of the reference's service, which can run beside this process on `push`).
//
// One process may hold many partitions: besides the reference's `prefix` (one BWT), the engine reads
//   shards  = [ "<prefix of shard 0>", ... ];     one .bwt per suffix partition (64)
//   devices = [ "0", "0", ..., "7" ];             HIP device of each shard (default: shard s -> GPU s * ndev / nshards)
//   batch_window_us = "200";  batch_max = "4096";  replies = "per_partition" | "summed";
//   query_threads = "8";                          windows answered at once (the reference's query pool: service.cpp:88)
//   suffixes = [ "<suffix of shard 0>", ... ];    the partitions' `suffix` values (default: `suffix` for a single shard)
//   reads = "on" | "off";                         off: ExactMatch-Reads requests are not answered here
// and then sends 2 x shards replies per request (front-end `workers` = 2 x shards) or 2 (`summed`).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rsbwt.h"

static const char *get(const rsbwt_service_config_t *c, const char *k, const char *dflt) {
    const char *v = rsbwt_service_config_get(c, k);
    return v ? v : dflt;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "Require more arguments to run the programme.\nusage: %s <service.cfg>\n", argv[0]);
        return EXIT_FAILURE;
    }
    rsbwt_service_config_t *cfg = nullptr;
    if (rsbwt_service_config_load(argv[1], &cfg) != RSBWT_OK) {
        fprintf(stderr, "%s\n", rsbwt_last_error());
        return EXIT_FAILURE;
    }
    printf("starting server for %s\n", get(cfg, "suffix", ""));
    // the sockets first: a box without libzmq should say so before minutes are spent loading shards into HBM
    // (connecting is asynchronous in ZeroMQ: nothing is received until the loop polls)
    rsbwt_transport_t *tr = nullptr;
    if (rsbwt_transport_zmq(get(cfg, "pull", ""), get(cfg, "push", ""), get(cfg, "push_count", ""), &tr) != RSBWT_OK) {
        fprintf(stderr, "%s\n", rsbwt_last_error());
        return EXIT_FAILURE;
    }
    std::vector<std::string> paths;
    const size_t ns = rsbwt_service_config_array_len(cfg, "shards");
    for (size_t i = 0; i < ns; ++i) paths.push_back(std::string(rsbwt_service_config_array_item(cfg, "shards", i)) + ".bwt");
    if (paths.empty()) paths.push_back(std::string(get(cfg, "prefix", "")) + ".bwt");
    const int ndev = rsbwt_device_count();
    if (ndev <= 0) {
        fprintf(stderr, "no HIP device is visible: the popBWT engine has no CPU fallback\n");
        return EXIT_FAILURE;
    }
    std::vector<int> devs(paths.size());
    const size_t nd = rsbwt_service_config_array_len(cfg, "devices");
    for (size_t i = 0; i < paths.size(); ++i)
        devs[i] = i < nd ? atoi(rsbwt_service_config_array_item(cfg, "devices", i)) : (int)(i * (size_t)ndev / paths.size());
    std::vector<const char *> cpaths;
    for (const std::string &p : paths) cpaths.push_back(p.c_str());
    rsbwt_set_t *set = nullptr;
    // (shards that serve reads are laid out with a psi hint in every window line and their select samples, built at open:
    // include/rsbwt.h, RSBWT_OPEN_READS)
    const bool serve_reads = strcmp(get(cfg, "reads", "on"), "off") != 0;
    if (rsbwt_set_open(cpaths.data(), cpaths.size(), devs.data(), RSBWT_OPEN_KTAB_GROUPED | (serve_reads ? RSBWT_OPEN_READS : 0u), &set) != RSBWT_OK) {
        fprintf(stderr, "%s\n", rsbwt_last_error());
        return EXIT_FAILURE;
    }
    printf("loaded %zu bwt shard(s) on %zu GPU(s).\n", rsbwt_set_size(set), rsbwt_set_devices(set));
    rsbwt_service_t *svc = nullptr;
    const bool summed = strcmp(get(cfg, "replies", "per_partition"), "summed") == 0;
    if (rsbwt_service_create(set, tr, atoll(get(cfg, "batch_window_us", "200")), (size_t)atoll(get(cfg, "batch_max", "4096")),
                             summed ? 0 : 1, &svc) != RSBWT_OK) {
        fprintf(stderr, "%s\n", rsbwt_last_error());
        return EXIT_FAILURE;
    }
    rsbwt_service_set_workers(svc, atoi(get(cfg, "query_threads", "8")));
    // min_read_length / max_read_length: service.cpp:1417-1420 (defaults 73 / 100, :56-57)
    rsbwt_service_set_reads(svc, serve_reads ? 1 : 0, (uint32_t)atoi(get(cfg, "min_read_length", "0")), (uint32_t)atoi(get(cfg, "max_read_length", "0")));
    {
        std::vector<std::string> suf;
        const size_t nsuf = rsbwt_service_config_array_len(cfg, "suffixes");
        for (size_t i = 0; i < paths.size(); ++i)
            suf.push_back(i < nsuf ? rsbwt_service_config_array_item(cfg, "suffixes", i) : (paths.size() == 1 ? get(cfg, "suffix", "") : ""));
        std::vector<const char *> csuf;
        for (const std::string &x : suf) csuf.push_back(x.c_str());
        if (rsbwt_service_set_suffixes(svc, csuf.data(), csuf.size()) != RSBWT_OK) {
            fprintf(stderr, "%s\n", rsbwt_last_error());
            return EXIT_FAILURE;
        }
    }
    printf("ready to serve from %s\n", get(cfg, "suffix", ""));
    fflush(stdout);
    const int rc = rsbwt_service_run(svc);  // forever (service.cpp:1521)
    rsbwt_service_free(svc);
    rsbwt_transport_free(tr);
    rsbwt_set_close(set);
    rsbwt_service_config_free(cfg);
    return rc == RSBWT_OK ? 0 : EXIT_FAILURE;
}

// service_main.cpp -- `rsbwt_service <service.cfg>`: the GPU twin of ReadServer's `service` process for
// the count path (src/service/service.cpp:1366-1583).  Reads the same configuration file, loads the
// BWT(s) into HBM, connects the same three sockets and answers CountReads / ExactMatch-Count requests
// in micro-batches; requests of other types are left unanswered (they belong to the RocksDB-backed
// paths of the reference's service, which can run beside this process on `push`).
//
// One process may hold many partitions: besides the reference's `prefix` (one BWT), the engine reads
//   shards  = [ "<prefix of shard 0>", ... ];     one .bwt per suffix partition (64)
//   devices = [ "0", "0", ..., "7" ];             HIP device of each shard (default: shard s -> GPU s * ndev / nshards)
//   batch_window_us = "200";  batch_max = "4096";  replies = "per_partition" | "summed";
//   query_threads = "8";                          windows answered at once (the reference's query pool: service.cpp:88)
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
    if (rsbwt_set_open(cpaths.data(), cpaths.size(), devs.data(), RSBWT_OPEN_KTAB_GROUPED, &set) != RSBWT_OK) {
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
    printf("ready to serve from %s\n", get(cfg, "suffix", ""));
    fflush(stdout);
    const int rc = rsbwt_service_run(svc);  // forever (service.cpp:1521)
    rsbwt_service_free(svc);
    rsbwt_transport_free(tr);
    rsbwt_set_close(set);
    rsbwt_service_config_free(cfg);
    return rc == RSBWT_OK ? 0 : EXIT_FAILURE;
}

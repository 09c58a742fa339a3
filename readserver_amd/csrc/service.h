// service.h -- the query service's CountReads / ExactMatch-Count slice, host side:
// what service_slice.cpp (codec + batched count_reads) and service_loop.cpp (the recv loop with its
// micro-batch window, the service.cfg reader, the transport interface) share.
#ifndef RSBWT_SERVICE_H
#define RSBWT_SERVICE_H

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/rsbwt.h"

namespace rsb {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

struct service_request {
    int t = 0, rt = 0;  // Request.RequestType / ReturnType (readserver.proto:4-5)
    std::string q;
};

bool service_decode(const uint8_t *msg, size_t len, service_request *out);

// The serialised Reply messages of one batch, back to back: message j is bytes[off[j] .. off[j+1]);
// request i's messages are first[i] .. first[i+1], in sending order (forward, reverse complement; per
// shard first when per_partition).
struct reply_arena {
    std::vector<uint8_t> bytes;
    std::vector<size_t> off, first;
    size_t messages() const { return off.empty() ? 0 : off.size() - 1; }
};
// handled[i] = 0: not a count request, left to the caller (no messages)
int service_count_batch(rsbwt_set_t *set, const std::vector<service_request> &rq, bool per_partition,
                        reply_arena *replies, std::vector<char> *handled);

// find_reads (src/service/service.cpp:714-797) for the ExactMatch requests of a batch whose return type is Reads
// (QueryTask::run, :1260-1291): per request, partition and strand one Reply{rt = ExactMatch, t = ReplyReads, q, r =
// ReplyReads{forward_matches | revcomp_matches}} carrying the reads that partition holds for the query.
struct reads_config {
    size_t min_read_length = 73, max_read_length = 100;  // service.cpp:56-57; service.cfg min_read_length / max_read_length (:1417-1420)
    std::vector<std::string> suffix;                      // service.cfg `suffix` of each shard of the set ("" where absent): a tile is looked
                                                          // up only in the partitions whose suffix it ends with (is_suffix_of, :228-230)
};
int service_reads_batch(rsbwt_set_t *set, const std::vector<service_request> &rq, bool per_partition, const reads_config &cfg,
                        reply_arena *replies, std::vector<char> *handled);
// the same requests answered with EMPTY read lists (a failed batch: the front-end has no timeout, server.cpp:469)
void service_reads_empty(const std::vector<service_request> &rq, size_t rows, reply_arena *replies, std::vector<char> *handled);
inline bool service_is_reads_request(const service_request &r) { return r.t == 2 && r.rt == 2; }  // ExactMatch + Reads

// What the loop needs of ZeroMQ: the SUB socket it receives Requests on (service.cpp:1495-1497) and
// the two PUSH sockets it answers on (push for ExactMatch, push_count for CountReads: :1499-1502,1568).
class transport {
  public:
    virtual ~transport() {}
    // next Request message; waits at most timeout_us (< 0: until one arrives or the transport
    // closes).  false = nothing arrived in time, or closed.
    virtual bool recv(std::vector<uint8_t> *msg, int64_t timeout_us) = 0;
    // up to `max` messages appended to *out: waits at most timeout_us for the first, takes whatever else is queued
    // already; the number taken (0: nothing arrived in time, or closed)
    virtual size_t recv_many(std::vector<std::vector<uint8_t>> *out, size_t max, int64_t timeout_us) {
        std::vector<uint8_t> m;
        if (max == 0 || !recv(&m, timeout_us)) return 0;
        out->push_back(std::move(m));
        return 1;
    }
    virtual bool closed() = 0;   // closed AND nothing left to receive
    virtual bool closing() = 0;  // close was asked for: what is queued is still answered
    enum channel { PUSH = 0, PUSH_COUNT = 1 };
    virtual void send(channel c, const uint8_t *data, size_t n) = 0;
    // messages [off[0], off[1]), ..., [off[count-1], off[count]) of `base`, in order
    virtual void send_many(channel c, const uint8_t *base, const size_t *off, size_t count) {
        for (size_t j = 0; j < count; ++j) send(c, base + off[j], off[j + 1] - off[j]);
    }
};

}  // namespace rsb
#endif

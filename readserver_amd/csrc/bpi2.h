// bpi2.h -- the reference's on-disk FM-index next to a .bwt ("<bwt>.bpi2"), host side.
// Written by RLEBWT::serialiseFMIndex (src/bwt/rlebwt.cpp:150-161: BPTree::serialise
// include/bwt/BPTree.h:189-199, BPNodes::serialise include/bwt/BPNodes.h:198-214, AlphaCount
// include/bwt/alphabet.h:89-91), read back by deserialiseFMIndex (rlebwt.cpp:163-200):
//   u64 depth
//   per level, top first:  u64 sizeof(counter) (2|4|8)  u64 length  u64 block  u64 bucket
//                          length x 5 counters ($ACGT)   length x 1 counter (their sum)
//   u64 nSum, nSum x u32 vSum
//   5 x u64 C[] ($ACGT)
// The engine never needs this file (its device index is built from the run bytes in seconds); it
// can write one for deployments that still run the CPU reference, and check one against the
// resident index.
#ifndef RSBWT_BPI2_H
#define RSBWT_BPI2_H

#include <stdint.h>

#include <string>
#include <vector>

namespace rsb {

struct bpi2_level {
    uint64_t width = 0, length = 0, block = 0, bucket = 0;  // bucket: runs per entry; block: entries per parent entry
    std::vector<uint64_t> counts;                           // length x 5, widened
    std::vector<uint64_t> sums;                             // length
};

struct bpi2_index {
    std::vector<bpi2_level> levels;  // top first; the last one has 64-run buckets
    std::vector<uint32_t> vsum;
    uint64_t pc[5] = {0, 0, 0, 0, 0};
    uint64_t num_runs = 0, num_symbols = 0;
};

// The index the reference would build for these runs (RLEBWT::initialiseFMIndex,
// rlebwt.cpp:34-148), fed run bytes in any number of pieces.
class bpi2_builder {
  public:
    explicit bpi2_builder(uint64_t num_runs);
    void add(const uint8_t *runs, size_t n);
    void finish();
    bool invalid() const { return invalid_; }  // a run byte with symbol rank > 4 was fed (and skipped)
    bpi2_index ix;

  private:
    std::vector<uint64_t> last_;  // depth x 5: BPNodes::m_last
    std::vector<uint64_t> next_;  // nextBuckets
    uint64_t i_ = 0, total_ = 0, next_sum_ = 0;
    bool invalid_ = false;
    void append(size_t level);
};

// 0 or an RSBWT_E* code; *err describes a failure
int bpi2_save(const bpi2_index &ix, const char *path, std::string *err);
int bpi2_load(const char *path, bpi2_index *ix, std::string *err);
int bpi2_from_bwt(const char *bwt_path, bpi2_index *ix, std::string *err);

}  // namespace rsb
#endif

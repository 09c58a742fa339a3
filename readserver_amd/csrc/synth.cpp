// synth.cpp -- deterministic synthetic population BWT (host only).
//
// Stands in for the part of demo/build_bwt.sh this image cannot run (SRA download, BFC, `sga
// index -a ropebwt --no-reverse`, demo/build_bwt.sh:923): a base genome, haplotypes carrying
// shared SNPs, fixed-length reads from both strands, reverse-lexicographic sort + dedup (what
// src/util/rlosort_seq_and_convert_sample_names.cpp:16-20,42-63 does), optional partition by the
// reversed last three bases (src/util/load_data_into_rocksdb.cpp:45, demo/permutations-3.txt),
// then the multi-string BWT ($ < A < C < G < T, $_i ordered by read index) written as an SGA
// .bwt (bwt_file.h).  The BWT is built by sorting all suffixes with a 21-symbol radix key and a
// tie-breaking comparison; sizes up to ~1e8 symbols are practical.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/rsbwt.h"
#include "bwt_file.h"
#include "synth_runs.h"

namespace rsb {

struct rng_t {  // xoshiro256**
    uint64_t s[4];
    explicit rng_t(uint64_t seed) {
        for (int i = 0; i < 4; ++i) s[i] = synth_mix64(seed + 0x1234567ull * (uint64_t)(i + 1));
    }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

static const char BASES[4] = {'A', 'C', 'G', 'T'};

static inline uint8_t sym3(char c) {  // $ACGT -> 0..4
    return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : c == 'T' ? 4 : 0;
}

static inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

struct suffix_t {
    uint64_t key;   // first 21 symbols, 3 bits each, left aligned; '$' and beyond = 0
    uint32_t read;
    uint32_t off;
};

static void build_bwt_runs(const std::vector<std::string> &reads, std::vector<uint8_t> &runs,
                           uint64_t &num_symbols) {
    std::vector<suffix_t> sa;
    uint64_t N = 0;
    for (const auto &r : reads) N += r.size() + 1;
    sa.reserve(N);
    for (uint32_t i = 0; i < reads.size(); ++i) {
        const std::string &r = reads[i];
        const uint32_t len = (uint32_t)r.size();
        // rolling key: key(off) from key(off+1)
        uint64_t key = 0;
        std::vector<uint64_t> keys(len + 1);
        keys[len] = 0;
        for (int o = (int)len - 1; o >= 0; --o) {
            key = (key >> 3) | ((uint64_t)sym3(r[o]) << 60);
            keys[o] = key;
        }
        for (uint32_t o = 0; o <= len; ++o) sa.push_back({keys[o], i, o});
    }
    auto less = [&reads](const suffix_t &a, const suffix_t &b) {
        if (a.key != b.key) return a.key < b.key;
        const std::string &ra = reads[a.read], &rb = reads[b.read];
        const size_t la = ra.size() - a.off, lb = rb.size() - b.off;  // symbols before '$'
        if (la > 21 && lb > 21) {
            const size_t m = std::min(la, lb) - 21;
            const int c = memcmp(ra.data() + a.off + 21, rb.data() + b.off + 21, m);
            if (c != 0) return c < 0;
        }
        if (la != lb) return la < lb;  // the shorter one meets its '$' first
        return a.read < b.read;        // equal strings: $_i < $_j for i < j
    };
    std::sort(sa.begin(), sa.end(), less);

    runs.clear();
    num_symbols = N;
    uint8_t cur_sym = 0xFF;
    uint32_t cur_len = 0;
    auto flush = [&]() {
        if (cur_len) runs.push_back((uint8_t)((cur_sym << 5) | cur_len));
        cur_len = 0;
    };
    for (const suffix_t &s : sa) {
        const uint8_t c = s.off ? sym3(reads[s.read][s.off - 1]) : 0;
        if (c != cur_sym || cur_len == 31) {
            flush();
            cur_sym = c;
        }
        ++cur_len;
    }
    flush();
}

}  // namespace rsb

using namespace rsb;

extern "C" int rsbwt_synth_popbwt(const char *bwt_path, const char *reads_path, uint64_t seed,
                                  uint64_t genome_len, uint32_t haplotypes, double snp_rate,
                                  uint32_t read_len, double coverage, int shard, int num_shards) {
    if (!bwt_path || genome_len < read_len || read_len < 3 || haplotypes == 0) return RSBWT_EINVAL;
    if (num_shards > 1 && (64 % num_shards != 0 || shard < 0 || shard >= num_shards))
        return RSBWT_EINVAL;
    rng_t rng(seed);
    std::string base(genome_len, 'A');
    for (auto &c : base) c = BASES[rng.below(4)];

    // shared variant pool: site, alt base, allele frequency
    struct variant { uint64_t pos; char alt; double freq; };
    std::vector<variant> pool((size_t)((double)genome_len * snp_rate * 4.0));
    for (auto &v : pool) {
        v.pos = rng.below(genome_len);
        do { v.alt = BASES[rng.below(4)]; } while (v.alt == base[v.pos]);
        v.freq = 0.05 + 0.5 * rng.unit();
    }
    std::vector<std::string> haps(haplotypes, base);
    for (auto &h : haps)
        for (const auto &v : pool)
            if (rng.unit() < v.freq) h[v.pos] = v.alt;

    const uint64_t nreads = (uint64_t)(coverage * (double)genome_len * haplotypes / read_len);
    std::vector<std::string> reads;
    reads.reserve(nreads);
    for (uint64_t i = 0; i < nreads; ++i) {
        const std::string &h = haps[rng.below(haplotypes)];
        const uint64_t st = rng.below(genome_len - read_len + 1);
        std::string r = h.substr(st, read_len);
        if (rng.next() & 1) {
            std::reverse(r.begin(), r.end());
            for (auto &c : r) c = comp(c);
        }
        reads.push_back(std::move(r));
    }
    // reverse-lexicographic order + dedup
    auto rlo_less = [](const std::string &a, const std::string &b) {
        return std::lexicographical_compare(a.rbegin(), a.rend(), b.rbegin(), b.rend());
    };
    std::sort(reads.begin(), reads.end(), rlo_less);
    reads.erase(std::unique(reads.begin(), reads.end()), reads.end());
    if (num_shards > 1) {
        std::vector<std::string> mine;
        for (auto &r : reads) {
            const size_t L = r.size();
            const int key = (sym3(r[L - 1]) - 1) * 16 + (sym3(r[L - 2]) - 1) * 4 + (sym3(r[L - 3]) - 1);
            if (key * num_shards / 64 == shard) mine.push_back(std::move(r));
        }
        reads.swap(mine);
    }

    std::vector<uint8_t> runs;
    uint64_t num_symbols = 0;
    build_bwt_runs(reads, runs, num_symbols);
    bwt_header hdr{(uint64_t)reads.size(), num_symbols, (uint64_t)runs.size(), 0};
    int rc = bwt_write(bwt_path, hdr, runs.data());
    if (rc != RSBWT_OK) return rc;
    if (reads_path && reads_path[0]) {
        FILE *f = fopen(reads_path, "w");
        if (!f) return RSBWT_EIO;
        for (const auto &r : reads) {
            fwrite(r.data(), 1, r.size(), f);
            fputc('\n', f);
        }
        if (fclose(f) != 0) return RSBWT_EIO;
    }
    return RSBWT_OK;
}

extern "C" int rsbwt_synth_runs_host(uint8_t *runs, uint64_t num_runs, uint64_t seed) {
    if (!runs && num_runs) return RSBWT_EINVAL;
    for (uint64_t i = 0; i < num_runs; ++i) runs[i] = synth_run_byte(seed, i);
    return RSBWT_OK;
}

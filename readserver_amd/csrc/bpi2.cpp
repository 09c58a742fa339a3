// bpi2.cpp -- see bpi2.h.  Host code only (file formats are the caller's side of the boundary).
#include "bpi2.h"

#include <errno.h>
#include <string.h>

#include "../../include/rsbwt.h"
#include "bwt_file.h"

namespace rsb {

namespace {
constexpr uint64_t LARGE = 1024, SMALL = 64, SYMBOL_RATE = 65536, MAX_RUN = 31;  // rlebwt.h:11-13, BPTree.h:9
constexpr uint64_t LARGE_SHIFT = 10;

int err_set(std::string *err, int code, const std::string &m) {
    if (err) *err = m;
    return code;
}
}  // namespace

// rlebwt.cpp:46-78: one upper level per factor of 1024 in the run count (inserted at the front, so
// the coarsest ends up first), counter width from bucket * block * 31 in size_t arithmetic, then
// the 64-run level with 16 entries per parent entry.
bpi2_builder::bpi2_builder(uint64_t num_runs) {
    ix.num_runs = num_runs;
    std::vector<bpi2_level> upper;
    uint64_t nb = num_runs >> LARGE_SHIFT, per_bucket = 1;
    while (nb > 0) {
        per_bucket *= LARGE;
        uint64_t max_count = per_bucket * LARGE * MAX_RUN;
        bpi2_level l;
        max_count >>= 16;
        if (max_count < 1) l.width = 2;
        else {
            max_count >>= 16;
            l.width = max_count < 1 ? 4 : 8;
        }
        l.bucket = per_bucket;
        l.block = LARGE;
        upper.push_back(l);
        nb >>= LARGE_SHIFT;
    }
    for (size_t k = upper.size(); k-- > 0;) ix.levels.push_back(upper[k]);
    bpi2_level bottom;
    bottom.width = 2;
    bottom.bucket = SMALL;
    bottom.block = LARGE / SMALL;
    ix.levels.push_back(bottom);
    for (auto &l : ix.levels) {
        const uint64_t expect = num_runs / l.bucket + 1;
        l.counts.reserve(expect * 5);
        l.sums.reserve(expect);
    }
    last_.assign(ix.levels.size() * 5, 0);
    next_.assign(ix.levels.size(), 0);
    ix.vsum.push_back(0);  // rlebwt.cpp:86
    next_sum_ = SYMBOL_RATE;
}

// BPNodes::appendLast (BPNodes.h:86-90): the running counters become the entry of the bucket
// that opens, truncated to the level's counter type
void bpi2_builder::append(size_t k) {
    bpi2_level &l = ix.levels[k];
    const uint64_t mask = l.width == 8 ? ~0ull : (1ull << (8 * l.width)) - 1ull;
    uint64_t sum = 0;
    for (int c = 0; c < 5; ++c) {
        l.counts.push_back(last_[k * 5 + c] & mask);
        sum += last_[k * 5 + c] & mask;
    }
    l.sums.push_back(sum & mask);
    ++l.length;
}

// the loop of rlebwt.cpp:99-127
void bpi2_builder::add(const uint8_t *runs, size_t n) {
    const size_t depth = ix.levels.size(), bottom = depth - 1;
    for (size_t r = 0; r < n; ++r, ++i_) {
        const uint8_t u = runs[r];
        if (total_ >= next_sum_) {  // the bucket of the run that reached the next 65,536-symbol mark
            ix.vsum.push_back((uint32_t)(ix.levels[bottom].length - 1));
            next_sum_ += SYMBOL_RATE;
        }
        if (i_ == next_[bottom]) {
            for (size_t j = 0; j < depth; ++j) {
                if (i_ != next_[j]) continue;
                for (size_t k = bottom; k > j; --k) {  // close the levels below: roll up, restart
                    for (int c = 0; c < 5; ++c) {
                        last_[(k - 1) * 5 + c] += last_[k * 5 + c];
                        last_[k * 5 + c] = 0;
                    }
                    append(k);
                    next_[k] = ix.levels[k].bucket * ix.levels[k].length;
                }
                append(j);
                next_[j] = ix.levels[j].bucket * ix.levels[j].length;
                break;
            }
        }
        if ((u >> 5) > 4u) {  // not a run of $ACGT: the reference would index past its AlphaCount
            invalid_ = true;
            continue;
        }
        last_[bottom * 5 + (u >> 5)] += u & 31u;
        total_ += u & 31u;
    }
}

// rlebwt.cpp:129-147: C[] from the running counters of all levels
void bpi2_builder::finish() {
    uint64_t tot[5] = {0, 0, 0, 0, 0};
    for (size_t k = 0; k < ix.levels.size(); ++k)
        for (int c = 0; c < 5; ++c) tot[c] += last_[k * 5 + c];
    ix.pc[0] = 0;
    for (int c = 1; c < 5; ++c) ix.pc[c] = ix.pc[c - 1] + tot[c - 1];
    ix.num_symbols = total_;
}

namespace {
bool put(FILE *f, const void *p, size_t n) { return fwrite(p, 1, n, f) == n; }
bool put_narrow(FILE *f, const std::vector<uint64_t> &v, uint64_t width) {
    std::vector<uint8_t> buf;
    const size_t CH = 1u << 16;
    for (size_t a = 0; a < v.size(); a += CH) {
        const size_t m = v.size() - a < CH ? v.size() - a : CH;
        buf.resize(m * width);
        for (size_t i = 0; i < m; ++i) memcpy(&buf[i * width], &v[a + i], width);  // little endian
        if (!put(f, buf.data(), buf.size())) return false;
    }
    return true;
}
bool get(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }
// `left` = bytes the file still holds: nothing is allocated for a count the file cannot back
bool get_wide(FILE *f, std::vector<uint64_t> *v, uint64_t count, uint64_t width, uint64_t *left) {
    if (count > *left / width) return false;
    *left -= count * width;
    v->assign(count, 0);
    std::vector<uint8_t> buf;
    const size_t CH = 1u << 16;
    for (uint64_t a = 0; a < count; a += CH) {
        const size_t m = count - a < CH ? (size_t)(count - a) : CH;
        buf.resize(m * width);
        if (!get(f, buf.data(), buf.size())) return false;
        for (size_t i = 0; i < m; ++i) memcpy(&(*v)[a + i], &buf[i * width], width);
    }
    return true;
}
}  // namespace

int bpi2_save(const bpi2_index &ix, const char *path, std::string *err) {
    FILE *f = fopen(path, "wb");
    if (!f) return err_set(err, RSBWT_EIO, std::string("cannot create ") + path + ": " + strerror(errno));
    bool ok = true;
    const uint64_t depth = ix.levels.size();
    ok = ok && put(f, &depth, 8);
    for (const auto &l : ix.levels) {
        ok = ok && put(f, &l.width, 8) && put(f, &l.length, 8) && put(f, &l.block, 8) && put(f, &l.bucket, 8);
        ok = ok && put_narrow(f, l.counts, l.width) && put_narrow(f, l.sums, l.width);
    }
    const uint64_t ns = ix.vsum.size();
    ok = ok && put(f, &ns, 8) && put(f, ix.vsum.data(), ns * 4) && put(f, ix.pc, 40);
    if (fclose(f) != 0) ok = false;
    if (!ok) return err_set(err, RSBWT_EIO, std::string("short write to ") + path);
    return RSBWT_OK;
}

int bpi2_load(const char *path, bpi2_index *ix, std::string *err) {
    FILE *f = fopen(path, "rb");
    if (!f) return err_set(err, RSBWT_EIO, std::string("cannot open ") + path + ": " + strerror(errno));
    *ix = bpi2_index();
    // every size field is held against what the file still has to give before anything is
    // allocated for it: a truncated or malformed file is reported, never a std::bad_alloc
    uint64_t left = 0;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long end = ftell(f);
        if (end > 0) left = (uint64_t)end;
    }
    rewind(f);
    uint64_t depth = 0;
    bool ok = left >= 8 && get(f, &depth, 8) && depth >= 1 && depth <= 8;
    left -= ok ? 8 : 0;
    for (uint64_t k = 0; ok && k < depth; ++k) {
        bpi2_level l;
        ok = left >= 32 && get(f, &l.width, 8) && get(f, &l.length, 8) && get(f, &l.block, 8) && get(f, &l.bucket, 8);
        if (ok) left -= 32;
        ok = ok && (l.width == 2 || l.width == 4 || l.width == 8) && l.length < (1ull << 40) && l.bucket > 0;
        ok = ok && get_wide(f, &l.counts, l.length * 5, l.width, &left) && get_wide(f, &l.sums, l.length, l.width, &left);
        if (ok) ix->levels.push_back(std::move(l));
    }
    uint64_t ns = 0;
    ok = ok && left >= 8 && get(f, &ns, 8) && ns < (1ull << 32) && ns * 4 + 40 <= left - 8;
    if (ok) {
        ix->vsum.resize(ns);
        ok = get(f, ix->vsum.data(), ns * 4) && get(f, ix->pc, 40);
    }
    uint8_t extra;
    if (ok && fread(&extra, 1, 1, f) != 0) ok = false;  // trailing bytes
    fclose(f);
    if (!ok) return err_set(err, RSBWT_EFORMAT, std::string(path) + " is not a .bpi2 index (truncated or malformed)");
    return RSBWT_OK;
}

int bpi2_from_bwt(const char *bwt_path, bpi2_index *ix, std::string *err) {
    FILE *f = nullptr;
    bwt_header hdr;
    const int rc = bwt_open_read(bwt_path, &f, &hdr);
    if (rc) return err_set(err, rc, std::string("cannot read ") + bwt_path + " as an SGA run-length BWT");
    bpi2_builder b(hdr.num_runs);
    std::vector<uint8_t> buf(1u << 22);
    uint64_t left = hdr.num_runs;
    while (left) {
        const size_t m = left < buf.size() ? (size_t)left : buf.size();
        if (fread(buf.data(), 1, m, f) != m) {
            fclose(f);
            return err_set(err, RSBWT_EFORMAT, std::string(bwt_path) + ": fewer run bytes than the header announces");
        }
        b.add(buf.data(), m);
        left -= m;
    }
    fclose(f);
    if (b.invalid()) return err_set(err, RSBWT_EFORMAT, std::string(bwt_path) + ": a run byte names a symbol rank above 4");
    b.finish();
    *ix = std::move(b.ix);
    return RSBWT_OK;
}

}  // namespace rsb

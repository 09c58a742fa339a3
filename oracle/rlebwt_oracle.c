/*
 * rlebwt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rlebwt_oracle.h).
 *
 * Clean-room C restatement of ReadServer's RLE-BWT FM-index and backward
 * search.  Each function cites the reference lines whose behaviour it follows
 * (paths relative to /root/reference).  The data structure is the reference's:
 * one byte per run (3-bit symbol rank | 5-bit length), a hierarchy of relative
 * per-symbol counters over 64-run buckets grouped 16 / 1024 / 1024 ... to a
 * parent, and a 65,536-symbol position sample ("vSum") that narrows the
 * per-level binary searches.
 */
#define _POSIX_C_SOURCE 200809L
#include "rlebwt_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
    SMALL_RATE = 64,        /* runs per bottom bucket: include/bwt/rlebwt.h:11 */
    SMALL_SHIFT = 6,
    LARGE_SHIFT = 10,       /* 1024 buckets per upper-level block: rlebwt.h:12 */
    SYMBOL_SHIFT = 16,      /* vSum sample rate 65,536: include/bwt/BPTree.h:9-10 */
    MAX_LEVELS = 8,
    NSYM = 5
};

/* "$ACGT" -> 0..4, anything else -> 0 (include/bwt/alphabet.h:8-36) */
static inline unsigned sym_rank(char b) {
    switch (b) {
    case 'A': return 1;
    case 'C': return 2;
    case 'G': return 3;
    case 'T': return 4;
    default: return 0;
    }
}
static const char RANK_ALPHABET[NSYM] = {'$', 'A', 'C', 'G', 'T'};

/* RLUnit: include/bwt/rlunit.h:8-11,64-83 */
static inline unsigned run_len(uint8_t u) { return u & 0x1Fu; }
static inline unsigned run_sym(uint8_t u) { return u >> 5; }

/* One level of relative counters = BPNodes<IntType> (include/bwt/BPNodes.h:35-236).
 * Entry p holds, for bucket p, the symbols (`sums`) and per-symbol counts
 * (`counts[5p..5p+4]`) between the start of the enclosing parent bucket and the
 * start of bucket p.  The top level is relative to the start of the BWT. */
typedef struct {
    int width;       /* bytes per counter: 2, 4 or 8 (BPNodes16/32/64) */
    uint64_t length; /* m_length      */
    uint64_t block;  /* m_block_size  : buckets per parent bucket */
    uint64_t bucket; /* m_bucket_size : runs per bucket           */
    int shift;       /* bottom-bucket id >> shift = bucket id on this level */
    void *counts;
    void *sums;
} level_t;

struct rso_index {
    const uint8_t *runs;
    uint8_t *owned;
    uint64_t num_runs, num_strings, num_symbols;
    int depth;
    level_t lv[MAX_LEVELS]; /* lv[0] = top, lv[depth-1] = 64-run buckets */
    uint32_t *vsum;         /* bottom bucket holding symbol number t*65536 */
    uint64_t nvsum;
    uint64_t pc[NSYM];      /* m_predCount */
};

static inline uint64_t lv_sum(const level_t *l, uint64_t p) {
    switch (l->width) {
    case 2: return ((const uint16_t *)l->sums)[p];
    case 4: return ((const uint32_t *)l->sums)[p];
    default: return ((const uint64_t *)l->sums)[p];
    }
}
static inline uint64_t lv_cnt(const level_t *l, unsigned c, uint64_t p) {
    switch (l->width) {
    case 2: return ((const uint16_t *)l->counts)[p * NSYM + c];
    case 4: return ((const uint32_t *)l->counts)[p * NSYM + c];
    default: return ((const uint64_t *)l->counts)[p * NSYM + c];
    }
}
static inline void lv_put(level_t *l, uint64_t p, const uint64_t rel[NSYM]) {
    uint64_t s = 0;
    for (int c = 0; c < NSYM; ++c) {
        s += rel[c];
        switch (l->width) {
        case 2: ((uint16_t *)l->counts)[p * NSYM + c] = (uint16_t)rel[c]; break;
        case 4: ((uint32_t *)l->counts)[p * NSYM + c] = (uint32_t)rel[c]; break;
        default: ((uint64_t *)l->counts)[p * NSYM + c] = rel[c]; break;
        }
    }
    switch (l->width) {
    case 2: ((uint16_t *)l->sums)[p] = (uint16_t)s; break;
    case 4: ((uint32_t *)l->sums)[p] = (uint32_t)s; break;
    default: ((uint64_t *)l->sums)[p] = s; break;
    }
}

/* Level sizing follows RLEBWT::initialiseFMIndex, src/bwt/rlebwt.cpp:46-78:
 * one upper level per factor of 1024 in the run count, counter width chosen
 * from bucket*block*31, then the 64-run bottom level with 16 buckets a block. */
static int plan_levels(rso_index *ix) {
    level_t up[MAX_LEVELS];
    int nup = 0;
    uint64_t nb = ix->num_runs >> LARGE_SHIFT;
    uint64_t per_bucket = 1;
    while (nb > 0) {
        if (nup == MAX_LEVELS - 1) return -1;
        per_bucket <<= LARGE_SHIFT;
        /* bucket*block*31 is a 128-bit product from the fourth level on; the
         * width rule only asks whether it fits 16 / 32 bits */
        unsigned __int128 max_count =
            (unsigned __int128)per_bucket * 1024u * 31u;
        level_t *l = &up[nup++];
        memset(l, 0, sizeof *l);
        l->width = (max_count >> 16) == 0 ? 2 : (max_count >> 32) == 0 ? 4 : 8;
        l->bucket = per_bucket;
        l->block = 1024;
        nb >>= LARGE_SHIFT;
    }
    ix->depth = nup + 1;
    for (int i = 0; i < nup; ++i) ix->lv[i] = up[nup - 1 - i]; /* insertAtFront */
    level_t *bot = &ix->lv[nup];
    memset(bot, 0, sizeof *bot);
    bot->width = 2;
    bot->bucket = SMALL_RATE;
    bot->block = 1u << (LARGE_SHIFT - SMALL_SHIFT);
    for (int i = 0; i < ix->depth; ++i) {
        level_t *l = &ix->lv[i];
        int below = ix->depth - 1 - i; /* levels beneath this one */
        /* BPTree::rank's sv: include/bwt/BPTree.h:98 */
        l->shift = (below > 0 ? (LARGE_SHIFT - SMALL_SHIFT) : 0) +
                   (below > 1 ? LARGE_SHIFT * (below - 1) : 0);
        l->length = (ix->num_runs + l->bucket - 1) / l->bucket;
        if (l->length == 0) l->length = 1;
        l->counts = calloc(l->length * NSYM, (size_t)l->width);
        l->sums = calloc(l->length, (size_t)l->width);
        if (!l->counts || !l->sums) return -1;
    }
    return 0;
}

/* One pass over the runs = RLEBWT::initialiseFMIndex, src/bwt/rlebwt.cpp:80-147.
 * The reference rolls bucket totals upward as buckets close; here the same
 * entries are produced from running absolute counts and the absolute counts
 * at the start of each level's current parent bucket. */
static int build_index(rso_index *ix) {
    if (plan_levels(ix) != 0) return -1;
    const int depth = ix->depth;
    uint64_t abs_cnt[NSYM] = {0, 0, 0, 0, 0};
    uint64_t base[MAX_LEVELS][NSYM];
    memset(base, 0, sizeof base);

    /* pass 0: number of symbols, to size vSum */
    uint64_t n = 0;
    for (uint64_t i = 0; i < ix->num_runs; ++i) n += run_len(ix->runs[i]);
    ix->num_symbols = n;
    ix->nvsum = (n >> SYMBOL_SHIFT) + 1;
    ix->vsum = (uint32_t *)calloc(ix->nvsum, sizeof(uint32_t));
    if (!ix->vsum) return -1;

    uint64_t total = 0;
    uint64_t next_sample = 1; /* vsum[0] = 0 already: rlebwt.cpp:91 */
    for (uint64_t i = 0; i < ix->num_runs; ++i) {
        if ((i & (SMALL_RATE - 1)) == 0) {
            for (int k = 0; k < depth; ++k) {
                level_t *l = &ix->lv[k];
                if (i % l->bucket != 0) continue;
                /* a bucket opening on level k-1 restarts level k's counters
                 * (clearLast/appendLast, rlebwt.cpp:112-117) */
                if (k > 0 && i % ix->lv[k - 1].bucket == 0)
                    memcpy(base[k], abs_cnt, sizeof abs_cnt);
                uint64_t rel[NSYM];
                for (int c = 0; c < NSYM; ++c) rel[c] = abs_cnt[c] - base[k][c];
                lv_put(l, i / l->bucket, rel);
            }
        }
        const uint8_t u = ix->runs[i];
        abs_cnt[run_sym(u)] += run_len(u);
        total += run_len(u);
        /* vSum[t] = bottom bucket of the run holding symbol number t*65536
         * (rlebwt.cpp:103-106; emitted here as soon as the run is consumed,
         * which also covers the last run: divergence D2) */
        while (next_sample < ix->nvsum &&
               total >= (next_sample << SYMBOL_SHIFT)) {
            ix->vsum[next_sample++] = (uint32_t)(i >> SMALL_SHIFT);
        }
    }
    /* C[]: rlebwt.cpp:129-147 */
    ix->pc[0] = 0;
    for (int c = 1; c < NSYM; ++c) ix->pc[c] = ix->pc[c - 1] + abs_cnt[c - 1];
    return 0;
}

rso_index *rso_from_runs(const uint8_t *runs, uint64_t num_runs,
                         uint64_t num_strings, int borrow) {
    rso_index *ix = (rso_index *)calloc(1, sizeof *ix);
    if (!ix) return NULL;
    ix->num_runs = num_runs;
    ix->num_strings = num_strings;
    if (borrow) {
        ix->runs = runs;
    } else {
        ix->owned = (uint8_t *)malloc(num_runs ? num_runs : 1);
        if (!ix->owned) { free(ix); return NULL; }
        memcpy(ix->owned, runs, num_runs);
        ix->runs = ix->owned;
    }
    if (build_index(ix) != 0) { rso_free(ix); return NULL; }
    return ix;
}

/* BWTReaderRLE::readHeader / readRuns, src/bwt/rlebwt_reader.cpp:27-48:
 * u16 magic 0xCACA, u64 strings, u64 symbols, u64 runs, 4-byte flag, runs. */
rso_index *rso_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    uint16_t magic = 0;
    uint64_t nstr = 0, nsym = 0, nruns = 0;
    uint32_t flag = 0;
    rso_index *ix = NULL;
    uint8_t *buf = NULL;
    if (fread(&magic, 2, 1, f) != 1 || magic != 0xCACA) goto out;
    if (fread(&nstr, 8, 1, f) != 1 || fread(&nsym, 8, 1, f) != 1 ||
        fread(&nruns, 8, 1, f) != 1 || fread(&flag, 4, 1, f) != 1)
        goto out;
    buf = (uint8_t *)malloc(nruns ? nruns : 1);
    if (!buf) goto out;
    if (fread(buf, 1, nruns, f) != nruns) goto out;
    ix = rso_from_runs(buf, nruns, nstr, 1);
    if (ix) {
        ix->owned = buf;
        buf = NULL;
        if (ix->num_symbols != nsym) { /* header disagrees with the runs */
            rso_free(ix);
            ix = NULL;
        }
    }
out:
    free(buf);
    fclose(f);
    return ix;
}

void rso_free(rso_index *ix) {
    if (!ix) return;
    for (int i = 0; i < ix->depth; ++i) {
        free(ix->lv[i].counts);
        free(ix->lv[i].sums);
    }
    free(ix->vsum);
    free(ix->owned);
    free(ix);
}

uint64_t rso_num_runs(const rso_index *ix) { return ix->num_runs; }
uint64_t rso_num_strings(const rso_index *ix) { return ix->num_strings; }
uint64_t rso_index_bytes(const rso_index *ix) {
    uint64_t b = ix->nvsum * 4;
    for (int i = 0; i < ix->depth; ++i)
        b += ix->lv[i].length * (uint64_t)ix->lv[i].width * (NSYM + 1);
    return b;
}

/* BPNodes::get_access, include/bwt/BPNodes.h:133-162: the largest p in
 * [lower, upper] whose sums[p] < cnt (lower if there is none). */
static inline uint64_t floor_by_sum(const level_t *l, uint64_t cnt,
                                    uint64_t lower, uint64_t upper) {
    uint64_t begin = lower, end = upper;
    if (end > l->length - 1) end = l->length - 1;
    while (end > begin) {
        const uint64_t p = (begin + end + 1) >> 1;
        if (lv_sum(l, p) >= cnt) {
            if (end == p) break;
            end = p;
        } else {
            begin = p;
        }
    }
    return begin;
}

/* BPNodes::select, include/bwt/BPNodes.h:164-196: the same search over the
 * count of symbol c inside parent bucket `offset`. */
static inline uint64_t floor_by_count(const level_t *l, unsigned c,
                                      uint64_t cnt, uint64_t offset) {
    uint64_t begin = l->block * offset;
    uint64_t end = begin + l->block - 1;
    if (cnt == 0) return begin;
    if (end > l->length - 1) end = l->length - 1;
    while (end > begin) {
        const uint64_t p = (begin + end + 1) >> 1;
        if (lv_cnt(l, c, p) >= cnt) {
            if (end == p) break;
            end = p;
        } else {
            begin = p;
        }
    }
    return begin;
}

typedef struct {
    uint64_t index; /* bottom bucket           */
    uint64_t sum;   /* symbols before it       */
    uint64_t count; /* symbols c before it     */
} marker_t;

/* BPTree::rank, include/bwt/BPTree.h:69-129, general branch (:96-118): vSum
 * gives the bottom buckets of the samples either side of `idx`; on each level
 * the search range is that pair scaled to the level, clamped to the parent
 * bucket chosen one level up.  In the last window the upper sample is the
 * last bottom bucket (divergence D1). */
static inline marker_t tree_rank(const rso_index *ix, unsigned c, uint64_t idx,
                                 int want_count) {
    marker_t mk = {0, 0, 0};
    const uint64_t lb = idx >> SYMBOL_SHIFT;
    const uint64_t lo_b = ix->vsum[lb];
    const uint64_t hi_b = (lb + 1 < ix->nvsum)
                              ? ix->vsum[lb + 1]
                              : ix->lv[ix->depth - 1].length - 1;
    uint64_t offset = 0;
    for (int i = 0; i < ix->depth; ++i) {
        const level_t *l = &ix->lv[i];
        uint64_t lower = lo_b >> l->shift;
        uint64_t upper = hi_b >> l->shift;
        if (i > 0) {
            const uint64_t lower_bound = offset * l->block;
            const uint64_t upper_bound = lower_bound + l->block - 1;
            if (lower_bound > lower) lower = lower_bound;
            if (upper_bound < upper) upper = upper_bound;
        }
        offset = (lower == upper) ? lower
                                  : floor_by_sum(l, idx - mk.sum, lower, upper);
        mk.sum += lv_sum(l, offset);
        if (want_count) mk.count += lv_cnt(l, c, offset);
    }
    mk.index = offset;
    return mk;
}

/* BPTree::select, include/bwt/BPTree.h:50-67 */
static inline marker_t tree_select(const rso_index *ix, unsigned c,
                                   uint64_t bc) {
    marker_t mk = {0, 0, 0};
    uint64_t offset = 0;
    for (int i = 0; i < ix->depth; ++i) {
        const level_t *l = &ix->lv[i];
        offset = floor_by_count(l, c, bc - mk.count, offset);
        mk.sum += lv_sum(l, offset);
        mk.count += lv_cnt(l, c, offset);
    }
    mk.index = offset;
    return mk;
}

uint64_t rso_bwlen(const rso_index *ix) { return ix->num_symbols; } /* rlebwt.cpp:303-305 */

uint64_t rso_pc(const rso_index *ix, char b) { return ix->pc[sym_rank(b)]; } /* :229-231 */

/* RLEBWT::getOcc, src/bwt/rlebwt.cpp:268-301: symbols b in BWT[0..index].
 * index == (uint64_t)-1 (updateInterval's lower-1 at lower == 0) gives 0. */
static inline uint64_t occ_rank(const rso_index *ix, unsigned c,
                                uint64_t index) {
    const uint64_t idx = index + 1;
    if (idx == 0) return 0;
    const marker_t mk = tree_rank(ix, c, idx, 1);
    uint64_t begin = mk.index * SMALL_RATE;
    uint64_t end = begin + SMALL_RATE - 1;
    uint64_t occ = mk.count;
    uint64_t offset = idx - mk.sum;
    if (end >= ix->num_runs) end = ix->num_runs - 1;
    for (uint64_t i = begin; i <= end; ++i) {
        const uint8_t u = ix->runs[i];
        const uint64_t count = run_len(u);
        if (offset <= count) {
            if (run_sym(u) == c) occ += offset;
            break;
        }
        offset -= count;
        if (run_sym(u) == c) occ += count;
    }
    return occ;
}

uint64_t rso_occ(const rso_index *ix, char b, uint64_t index) {
    if (index != (uint64_t)-1 && index >= ix->num_symbols)
        index = ix->num_symbols - 1;
    return occ_rank(ix, sym_rank(b), index);
}

/* RLEBWT::getChar, src/bwt/rlebwt.cpp:202-227 */
static inline unsigned char_rank(const rso_index *ix, uint64_t index) {
    const uint64_t idx = index + 1;
    const marker_t mk = tree_rank(ix, 0, idx, 0);
    uint64_t begin = mk.index * SMALL_RATE;
    uint64_t end = begin + SMALL_RATE - 1;
    uint64_t offset = idx - mk.sum;
    if (end >= ix->num_runs) end = ix->num_runs - 1;
    for (uint64_t i = begin; i <= end; ++i) {
        const uint64_t count = run_len(ix->runs[i]);
        if (offset <= count) return run_sym(ix->runs[i]);
        offset -= count;
    }
    return run_sym(ix->runs[end]);
}

char rso_char(const rso_index *ix, uint64_t index) {
    return RANK_ALPHABET[char_rank(ix, index)];
}

/* RLEBWT::getOccAt, src/bwt/rlebwt.cpp:233-266: position of the bc-th b. */
static inline uint64_t occ_at_rank(const rso_index *ix, unsigned c,
                                   uint64_t bc) {
    const marker_t mk = tree_select(ix, c, bc);
    uint64_t begin = mk.index * SMALL_RATE;
    uint64_t end = begin + SMALL_RATE - 1;
    uint64_t offset = bc - mk.count;
    uint64_t index = mk.sum;
    if (end >= ix->num_runs) end = ix->num_runs - 1;
    for (uint64_t i = begin; i <= end; ++i) {
        const uint8_t u = ix->runs[i];
        const uint64_t count = run_len(u);
        if (run_sym(u) != c) {
            index += count;
            continue;
        }
        if (offset <= count) {
            index += offset - 1;
            break;
        }
        offset -= count;
        index += count;
    }
    return index;
}

uint64_t rso_occ_at(const rso_index *ix, char b, uint64_t bc) {
    return occ_at_rank(ix, sym_rank(b), bc);
}

/* RLEBWT::getF, src/bwt/rlebwt.cpp:307-314 */
static inline unsigned f_rank(const rso_index *ix, uint64_t idx) {
    unsigned ci = 0;
    while (ci < NSYM && ix->pc[ci] <= idx) ci++;
    return ci - 1;
}

char rso_f(const rso_index *ix, uint64_t index) {
    return RANK_ALPHABET[f_rank(ix, index)];
}

/* findInterval / initInterval / updateInterval, src/bwt/query.cpp:11-41.
 * Returns the number of updateInterval calls made. */
static inline unsigned find_interval(const rso_index *ix, const char *w,
                                     size_t len, uint64_t *lower_out,
                                     uint64_t *upper_out) {
    unsigned steps = 0;
    if (len == 0) { *lower_out = 1; *upper_out = 0; return 0; }
    for (size_t i = 0; i < len; ++i) {
        if (sym_rank(w[i]) == 0) { *lower_out = 1; *upper_out = 0; return 0; }
    }
    long j = (long)len - 1;
    unsigned c = sym_rank(w[j]);
    /* initInterval: query.cpp:18-21 */
    uint64_t lower = ix->pc[c];
    uint64_t upper = lower + occ_rank(ix, c, ix->num_symbols - 1) - 1;
    for (--j; j >= 0; --j) {
        c = sym_rank(w[j]);
        /* updateInterval: query.cpp:11-15 */
        const uint64_t pb = ix->pc[c];
        lower = pb + occ_rank(ix, c, lower - 1);
        upper = pb + occ_rank(ix, c, upper) - 1;
        ++steps;
        if (lower > upper) break; /* query.cpp:35-37 */
    }
    *lower_out = lower;
    *upper_out = upper;
    return steps;
}

void rso_find_interval(const rso_index *ix, const char *w, size_t len,
                       uint64_t *lower, uint64_t *upper) {
    (void)find_interval(ix, w, len, lower, upper);
}

/* extractPrefix, src/bwt/query.cpp:43-63 (LF walk until '$', then reverse) */
size_t rso_extract_prefix(const rso_index *ix, uint64_t index, char *out,
                          size_t cap) {
    size_t n = 0;
    uint64_t idx = index;
    for (;;) {
        const unsigned c = char_rank(ix, idx);
        if (c == 0) break;
        if (n == cap) return (size_t)-1;
        idx = ix->pc[c] + occ_rank(ix, c, idx - 1);
        out[n++] = RANK_ALPHABET[c];
    }
    for (size_t i = 0; i < n / 2; ++i) {
        const char t = out[i];
        out[i] = out[n - 1 - i];
        out[n - 1 - i] = t;
    }
    return n;
}

/* extractPostfix, src/bwt/query.cpp:65-85 (F / select walk until '$') */
size_t rso_extract_postfix(const rso_index *ix, uint64_t index, char *out,
                           size_t cap) {
    size_t n = 0;
    uint64_t idx = index;
    for (;;) {
        const unsigned f = f_rank(ix, idx);
        if (f == 0) break;
        if (n == cap) return (size_t)-1;
        const uint64_t fc = idx - ix->pc[f] + 1;
        idx = occ_at_rank(ix, f, fc);
        out[n++] = RANK_ALPHABET[f];
    }
    return n;
}

typedef struct {
    const rso_index *ix;
    const char *kmers;
    size_t Q, stride;
    uint32_t k;
    uint64_t *lower, *upper;
    uint8_t *steps;
    int tid, nthreads;
} job_t;

static void *job_main(void *p) {
    const job_t *j = (const job_t *)p;
    for (size_t q = (size_t)j->tid; q < j->Q; q += (size_t)j->nthreads) {
        const unsigned s = find_interval(j->ix, j->kmers + q * j->stride, j->k,
                                         &j->lower[q], &j->upper[q]);
        if (j->steps) j->steps[q] = (uint8_t)(s > 255 ? 255 : s);
    }
    return NULL;
}

void rso_find_intervals(const rso_index *ix, const char *kmers, size_t Q,
                        uint32_t k, size_t stride, uint64_t *lower,
                        uint64_t *upper, uint8_t *steps, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t jobs[256];
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (job_t){ix, kmers, Q, stride, k, lower, upper, steps, t, nthreads};
    }
    if (nthreads == 1) { job_main(&jobs[0]); return; }
    for (int t = 0; t < nthreads; ++t)
        pthread_create(&th[t], NULL, job_main, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* extractPrefix + extractPostfix (src/bwt/query.cpp:43-85, joined as query() does, :94-96) of n rows, split over
 * `nthreads` POSIX threads sharing the index: row i's read to out + i * stride (no NUL), its length to len[i] and the
 * length of its prefix part to prefix_len[i] (may be NULL); a read that does not fit `stride` bytes gets
 * len = UINT32_MAX.  The CPU baseline of bench.py --mode extract on the host's cores, and the checker of its reads. */
typedef struct {
    const rso_index *ix;
    const uint64_t *rows;
    size_t n, stride;
    char *out;
    uint32_t *len, *plen;
    int tid, nthreads;
} xjob_t;

static void *xjob_main(void *p) {
    const xjob_t *j = (const xjob_t *)p;
    for (size_t i = (size_t)j->tid; i < j->n; i += (size_t)j->nthreads) {
        char *o = j->out + i * j->stride;
        const size_t a = rso_extract_prefix(j->ix, j->rows[i], o, j->stride);
        size_t b = (size_t)-1;
        if (a != (size_t)-1) b = rso_extract_postfix(j->ix, j->rows[i], o + a, j->stride - a);
        if (a == (size_t)-1 || b == (size_t)-1) {
            j->len[i] = 0xFFFFFFFFu;
            if (j->plen) j->plen[i] = 0;
        } else {
            j->len[i] = (uint32_t)(a + b);
            if (j->plen) j->plen[i] = (uint32_t)a;
        }
    }
    return NULL;
}

void rso_extract_batch(const rso_index *ix, const uint64_t *rows, size_t n, char *out, size_t stride, uint32_t *len,
                       uint32_t *prefix_len, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    xjob_t jobs[256];
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) jobs[t] = (xjob_t){ix, rows, n, stride, out, len, prefix_len, t, nthreads};
    if (nthreads == 1) { xjob_main(&jobs[0]); return; }
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, xjob_main, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

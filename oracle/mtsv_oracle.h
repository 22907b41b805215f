/*
 * mtsv_oracle.h -- CPU restatement of the mtsv-binner hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there only
 * as the checker / the reported CPU baseline.  The product (mtsv_tools_amd/) never links, imports
 * or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to the upstream
 * FofanovLab/mtsv_tools checkout).  FM-index arithmetic lives in the un-vendored crate
 * `bio 3.0.0` (Cargo.lock:63-66) and the file codec in `bincode 1.3.3` (Cargo.lock:54-57); their
 * published algorithms are restated here and anchored on the reference's call sites
 * (index.rs:305, :347, :560-574; io.rs:121,131).
 *
 * Pinning status:
 *   - min_edit_distance: pinned by the 9 known-answer tests of src/align.rs:100-170.
 *   - window arithmetic / merge rules: pinned by src/index.rs:721-857.
 *   - result line format: pinned by src/binner.rs:440-472.
 *   - SW prefilter score: pinned against the reference's own ssw/src/ssw.c compiled into
 *     oracle/_ref/libssw_ref.so (tests/test_oracle_ssw.py) and by committed golden vectors.
 *   - FM search / locate / MG-index bytes: the reference holds no vector at the bio/bincode
 *     boundary -> "parity unpinned" there; mitigated by brute-force substring search, which the
 *     exact-match semantics of the path make an independent oracle.
 */
#ifndef MTSV_ORACLE_H
#define MTSV_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/index.rs:45-54 */
typedef struct {
    uint32_t gi;
    uint32_t tax_id;
    uint64_t start;
    uint64_t end;
} orc_bin;

#define ORC_OCC_SYMS 117 /* max symbol of n_alphabet() 't'=116, +1  (index.rs:560,571) */
#define ORC_LESS_LEN 118

/* src/index.rs:60-68 + bio SampledSuffixArray<BWT,Less,Occ> */
typedef struct {
    uint64_t n;         /* length of sequences incl. trailing '$' (index.rs:555) */
    uint8_t *text;      /* sequences */
    uint64_t nbins;
    orc_bin *bins;
    uint8_t *bwt;       /* one ASCII symbol per byte */
    uint64_t less_len;
    uint64_t *less;
    uint64_t occ_outer; /* 117 */
    uint64_t *occ[ORC_OCC_SYMS];
    uint64_t occ_len[ORC_OCC_SYMS];
    uint32_t k;
    uint64_t nsample;
    uint64_t *sample;
    uint64_t s;
    uint64_t n_extra;
    uint64_t *extra_key;
    uint64_t *extra_val;
    uint8_t sentinel;
} orc_index;

/* argument list of MGIndex::matching_tax_ids, src/index.rs:258-269 */
typedef struct {
    double edit_rate;
    uint64_t seed_size;
    uint64_t seed_gap;
    double min_seed;
    uint64_t max_hits;
    uint64_t tune_max_hits;
    int64_t max_candidates;  /* -1 = None */
    int64_t max_assignments; /* -1 = None */
} orc_params;

/* src/index.rs:30-40, plus which strand call produced it */
typedef struct {
    uint64_t read;
    uint32_t tax_id;
    uint32_t gi;
    uint32_t edit;
    uint32_t strand; /* 0 = forward call (binner.rs:102), 1 = revcomp call (binner.rs:116) */
    uint64_t offset;
} orc_hit;

/* instrumentation for SURVEY.md 8(d): bytes = 64*(2X+S) + 8H + W + L + 24R */
typedef struct {
    uint64_t X;      /* backward-search extension steps executed */
    uint64_t S;      /* LF steps executed in locate */
    uint64_t H;      /* located seed hits */
    uint64_t W;      /* sum of window lengths handed to the SW prefilter */
    uint64_t R;      /* hits returned */
    uint64_t Lsum;   /* sum of read lengths (once per read) */
    uint64_t n_sw;   /* SW calls */
    uint64_t n_edit; /* edit-distance calls */
    uint64_t n_cand; /* candidates after coalescing */
    uint64_t n_seed; /* seeds searched */
} orc_counters;

void orc_default_params(orc_params *p);

/* ---- index construction / codec ---- */
/* MGIndex::new (index.rs:491-582).  seqs given in BTreeMap iteration order by the caller is NOT
 * assumed: entries are stably sorted by tax_id here. */
orc_index *orc_index_build(uint64_t nseq, const uint32_t *tax, const uint32_t *gi,
                           const uint8_t *const *seq, const uint64_t *seq_len,
                           uint32_t occ_k, uint64_t sa_s);
int orc_index_write(const orc_index *ix, const char *path); /* io.rs:125-133 */
orc_index *orc_index_read(const char *path);                /* io.rs:115-123 */
void orc_index_free(orc_index *ix);
const char *orc_last_error(void);

/* ---- FM primitives (bio) ---- */
uint64_t orc_occ_get(const orc_index *ix, uint64_t r, uint8_t a);
/* returns 1 if Complete, writes [lower, upper) */
int orc_backward_search(const orc_index *ix, const uint8_t *pat, uint64_t len, uint64_t *lower,
                        uint64_t *upper, orc_counters *c);
uint64_t orc_sa_get(const orc_index *ix, uint64_t row, orc_counters *c);

/* ---- verification kernels ---- */
uint32_t orc_min_edit_distance(const uint8_t *p, uint64_t m, const uint8_t *t, uint64_t n);
/* exact local alignment score, scores +1/-1 (N==N matches), gap open go / extend ge in ssw's
 * convention (first gap base costs go, each further one ge) */
uint32_t orc_sw_exact(const uint8_t *read, uint64_t m, const uint8_t *ref, uint64_t n, int go,
                      int ge);
/* literal lane-by-lane emulation of ssw.c:123-328 / :354-530 / :762-852 as called from
 * ssw/src/lib.rs:36-84; returns score1 */
uint32_t orc_ssw_score(const uint8_t *read, uint64_t m, const uint8_t *ref, uint64_t n);
uint32_t orc_ssw_byte(const uint8_t *read, uint64_t m, const uint8_t *ref, uint64_t n);
uint32_t orc_ssw_word(const uint8_t *read, uint64_t m, const uint8_t *ref, uint64_t n);

/* SeedHit::candidate_indices, index.rs:118-153.  returns 1 = Some */
int orc_candidate_indices(uint64_t site, uint64_t qoff, const orc_bin *bin, uint64_t read_len,
                          uint64_t edit_distance, uint64_t *start, uint64_t *end);

/* ---- the path ---- */
/* MGIndex::matching_tax_ids (index.rs:258-432) on one already-normalised strand.
 * Appends to hits (capacity cap); returns number of hits, or -1 on capacity overflow. */
int64_t orc_matching_tax_ids(const orc_index *ix, const uint8_t *seq, uint64_t len,
                             const orc_params *p, orc_hit *hits, uint64_t cap, orc_counters *c);

/* worker closure of binner.rs:77-131 for a batch of reads.  bases = concatenated raw read bytes,
 * read_off[n_reads+1].  Hits ordered by (read, strand, rank order).  *hits is malloc'd. */
int orc_bin_batch(const orc_index *ix, const uint8_t *bases, const uint64_t *read_off,
                  uint64_t n_reads, const orc_params *p, int n_threads, orc_hit **hits,
                  uint64_t *n_hits, orc_counters *c);
void orc_free(void *p);

/* write_assignments, binner.rs:310-379.  Formats one read's line into buf (cap bytes) and
 * returns its length (0 when the read has no hits), -1 if buf is too small. */
int64_t orc_format_line(const char *read_id, const orc_hit *hits, uint64_t n_hits, int long_format,
                        char *buf, uint64_t cap);

/* brute force exact-match positions of pat in text[0..n-1) (independent check of search+locate) */
uint64_t orc_brute_find(const orc_index *ix, const uint8_t *pat, uint64_t len, uint64_t *out,
                        uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif

/*
 * mtsv_oracle.c -- CPU restatement of the mtsv-binner hot path (see mtsv_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked, imported or called by the product.
 *
 * Plain C99 (+OpenMP for the batch driver).  Deliberately keeps the reference's data layout:
 * byte-per-symbol BWT, u64 Occ checkpoints every k, row-sampled SA every s with LF-walk locate,
 * full-matrix edit distance, lane-by-lane emulation of the SSE2 striped Smith-Waterman.
 */
#define _GNU_SOURCE
#include "mtsv_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];
const char *orc_last_error(void) { return g_err; }
#define FAIL(...)                                  \
    do {                                           \
        snprintf(g_err, sizeof g_err, __VA_ARGS__); \
    } while (0)

void orc_free(void *p) { free(p); }

/* defaults: src/bin/mtsv-binner.rs:63-94 */
void orc_default_params(orc_params *p) {
    p->edit_rate = 0.13;
    p->seed_size = 18;
    p->seed_gap = 15;
    p->min_seed = 0.015;
    p->max_hits = 2000;
    p->tune_max_hits = 200;
    p->max_candidates = -1;
    p->max_assignments = -1;
}

/* ------------------------------------------------------------------------------------------ */
/* Index construction: MGIndex::new, src/index.rs:491-582                                      */
/* ------------------------------------------------------------------------------------------ */

static const uint8_t *g_sort_text;
static uint64_t g_sort_n;
static int suffix_cmp(const void *a, const void *b) {
    uint64_t i = *(const uint64_t *)a, j = *(const uint64_t *)b;
    if (i == j) return 0;
    uint64_t li = g_sort_n - i, lj = g_sort_n - j;
    uint64_t l = li < lj ? li : lj;
    int c = memcmp(g_sort_text + i, g_sort_text + j, l);
    if (c) return c;
    /* unreachable with a unique smallest sentinel, kept for totality */
    return li < lj ? -1 : 1;
}

/* symbols that get Occ arrays: n_alphabet() = "ACGTNacgtn" plus the sentinel '$'
 * (bio Occ::new; call site index.rs:560,571) */
static const uint8_t OCC_ALPHA[11] = {'A', 'C', 'G', 'T', 'N', 'a', 'c', 'g', 't', 'n', '$'};

typedef struct {
    uint32_t tax, gi;
    uint64_t idx;
} ent_t;
static int ent_cmp(const void *a, const void *b) {
    const ent_t *x = a, *y = b;
    if (x->tax != y->tax) return x->tax < y->tax ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

orc_index *orc_index_build(uint64_t nseq, const uint32_t *tax, const uint32_t *gi,
                           const uint8_t *const *seq, const uint64_t *seq_len, uint32_t occ_k,
                           uint64_t sa_s) {
    if (occ_k == 0 || sa_s == 0) {
        FAIL("sampling intervals must be > 0");
        return NULL;
    }
    orc_index *ix = calloc(1, sizeof *ix);
    /* BTreeMap<TaxId, Vec<(Gi,Seq)>> iteration: ascending TaxId, insertion order inside
     * (index.rs:497-510) */
    ent_t *ent = malloc((nseq ? nseq : 1) * sizeof *ent);
    uint64_t total = 0;
    for (uint64_t i = 0; i < nseq; i++) {
        ent[i].tax = tax[i];
        ent[i].gi = gi[i];
        ent[i].idx = i;
        total += seq_len[i];
    }
    qsort(ent, nseq, sizeof *ent, ent_cmp);
    uint64_t n = total + 1;
    ix->n = n;
    ix->text = malloc(n);
    ix->nbins = nseq;
    ix->bins = malloc((nseq ? nseq : 1) * sizeof(orc_bin));
    uint64_t pos = 0;
    for (uint64_t e = 0; e < nseq; e++) {
        uint64_t i = ent[e].idx;
        ix->bins[e].gi = ent[e].gi;
        ix->bins[e].tax_id = ent[e].tax;
        ix->bins[e].start = pos;
        ix->bins[e].end = pos + seq_len[i];
        memcpy(ix->text + pos, seq[i], seq_len[i]);
        pos += seq_len[i];
    }
    free(ent);
    /* DNA5 normalisation, index.rs:543-553 */
    for (uint64_t i = 0; i < total; i++) {
        uint8_t b = ix->text[i];
        switch (b) {
        case 'A': case 'C': case 'G': case 'T': case 'N': break;
        case 'a': b = 'A'; break;
        case 'c': b = 'C'; break;
        case 'g': b = 'G'; break;
        case 't': b = 'T'; break;
        default: b = 'N';
        }
        ix->text[i] = b;
    }
    ix->text[total] = '$'; /* index.rs:555 */
    ix->sentinel = '$';

    /* suffix_array(&seq), index.rs:563 -- any correct SA is the same SA */
    uint64_t *sa = malloc(n * sizeof *sa);
    for (uint64_t i = 0; i < n; i++) sa[i] = i;
    g_sort_text = ix->text;
    g_sort_n = n;
    qsort(sa, n, sizeof *sa, suffix_cmp);

    /* bwt(&seq,&sa), index.rs:567: bwt[i] = text[sa[i]-1], or the last symbol when sa[i]==0 */
    ix->bwt = malloc(n);
    for (uint64_t i = 0; i < n; i++) ix->bwt[i] = sa[i] ? ix->text[sa[i] - 1] : ix->text[n - 1];

    /* less(&bwt,&alphabet), index.rs:570: len = max_symbol + 2; less[c] = #symbols < c */
    ix->less_len = ORC_LESS_LEN;
    ix->less = calloc(ORC_LESS_LEN, sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) ix->less[ix->bwt[i]]++;
    for (int i = 1; i < ORC_LESS_LEN; i++) ix->less[i] += ix->less[i - 1];
    for (int i = ORC_LESS_LEN - 1; i >= 1; i--) ix->less[i] = ix->less[i - 1];
    ix->less[0] = 0;

    /* Occ::new(&bwt,k,&alphabet), index.rs:571: entry j of symbol a = count of a in bwt[0..=j*k] */
    ix->k = occ_k;
    ix->occ_outer = ORC_OCC_SYMS;
    uint64_t nchk = (n - 1) / occ_k + 1;
    for (int a = 0; a < 11; a++) {
        ix->occ[OCC_ALPHA[a]] = malloc(nchk * sizeof(uint64_t));
        ix->occ_len[OCC_ALPHA[a]] = 0;
    }
    uint64_t curr[256];
    memset(curr, 0, sizeof curr);
    for (uint64_t i = 0; i < n; i++) {
        curr[ix->bwt[i]]++;
        if (i % occ_k == 0)
            for (int a = 0; a < 11; a++) {
                uint8_t s = OCC_ALPHA[a];
                ix->occ[s][ix->occ_len[s]++] = curr[s];
            }
    }

    /* sa.sample(&seq,bwt,less,occ,s), index.rs:574: rows i%s==0 keep SA[i]; a non-sampled row
     * whose BWT symbol is the sentinel goes to extra_rows */
    ix->s = sa_s;
    ix->nsample = (n + sa_s - 1) / sa_s;
    ix->sample = malloc(ix->nsample * sizeof(uint64_t));
    ix->extra_key = malloc(sizeof(uint64_t));
    ix->extra_val = malloc(sizeof(uint64_t));
    ix->n_extra = 0;
    uint64_t ns = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (i % sa_s == 0)
            ix->sample[ns++] = sa[i];
        else if (ix->bwt[i] == ix->sentinel) {
            ix->extra_key[0] = i;
            ix->extra_val[0] = sa[i];
            ix->n_extra = 1;
        }
    }
    free(sa);
    return ix;
}

void orc_index_free(orc_index *ix) {
    if (!ix) return;
    free(ix->text);
    free(ix->bins);
    free(ix->bwt);
    free(ix->less);
    for (int i = 0; i < ORC_OCC_SYMS; i++) free(ix->occ[i]);
    free(ix->sample);
    free(ix->extra_key);
    free(ix->extra_val);
    free(ix);
}

/* ------------------------------------------------------------------------------------------ */
/* bincode 1.3.3 default options (io.rs:121,131): little-endian fixed ints, usize->u64,        */
/* Vec = u64 len + items, struct = fields in order, map = u64 len + (k,v)                      */
/* ------------------------------------------------------------------------------------------ */
static int w64(FILE *f, uint64_t v) { return fwrite(&v, 8, 1, f) == 1; }
static int w32(FILE *f, uint32_t v) { return fwrite(&v, 4, 1, f) == 1; }

int orc_index_write(const orc_index *ix, const char *path) {
    FILE *f = fopen(path, "wb");
    if (!f) {
        FAIL("cannot open %s for writing", path);
        return -1;
    }
    int ok = 1;
    ok &= w64(f, ix->n);
    ok &= fwrite(ix->text, 1, ix->n, f) == ix->n;
    ok &= w64(f, ix->nbins);
    for (uint64_t i = 0; i < ix->nbins; i++) {
        ok &= w32(f, ix->bins[i].gi); /* field order index.rs:45-54 */
        ok &= w32(f, ix->bins[i].tax_id);
        ok &= w64(f, ix->bins[i].start);
        ok &= w64(f, ix->bins[i].end);
    }
    /* SampledSuffixArray { bwt, less, occ{occ,k}, sample, s, extra_rows, sentinel } */
    ok &= w64(f, ix->n);
    ok &= fwrite(ix->bwt, 1, ix->n, f) == ix->n;
    ok &= w64(f, ix->less_len);
    ok &= fwrite(ix->less, 8, ix->less_len, f) == ix->less_len;
    ok &= w64(f, ix->occ_outer);
    for (uint64_t a = 0; a < ix->occ_outer; a++) {
        ok &= w64(f, ix->occ_len[a]);
        if (ix->occ_len[a]) ok &= fwrite(ix->occ[a], 8, ix->occ_len[a], f) == ix->occ_len[a];
    }
    ok &= w32(f, ix->k);
    ok &= w64(f, ix->nsample);
    ok &= fwrite(ix->sample, 8, ix->nsample, f) == ix->nsample;
    ok &= w64(f, ix->s);
    ok &= w64(f, ix->n_extra);
    for (uint64_t i = 0; i < ix->n_extra; i++) {
        ok &= w64(f, ix->extra_key[i]);
        ok &= w64(f, ix->extra_val[i]);
    }
    ok &= fwrite(&ix->sentinel, 1, 1, f) == 1;
    if (fclose(f) != 0) ok = 0;
    if (!ok) {
        FAIL("short write to %s", path);
        return -1;
    }
    return 0;
}

static int r64(FILE *f, uint64_t *v) { return fread(v, 8, 1, f) == 1; }
static int r32(FILE *f, uint32_t *v) { return fread(v, 4, 1, f) == 1; }

orc_index *orc_index_read(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        FAIL("cannot open %s", path);
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    uint64_t fsize = (uint64_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    orc_index *ix = calloc(1, sizeof *ix);
#define NEED(cond, msg)            \
    if (!(cond)) {                 \
        FAIL("%s: %s", path, msg); \
        goto bad;                  \
    }
    NEED(r64(f, &ix->n) && ix->n >= 1 && ix->n <= fsize, "bad sequences length");
    ix->text = malloc(ix->n);
    NEED(fread(ix->text, 1, ix->n, f) == ix->n, "truncated sequences");
    NEED(r64(f, &ix->nbins) && ix->nbins <= fsize / 24, "bad bins length");
    ix->bins = malloc((ix->nbins ? ix->nbins : 1) * sizeof(orc_bin));
    for (uint64_t i = 0; i < ix->nbins; i++) {
        NEED(r32(f, &ix->bins[i].gi) && r32(f, &ix->bins[i].tax_id) && r64(f, &ix->bins[i].start) &&
                 r64(f, &ix->bins[i].end),
             "truncated bins");
    }
    uint64_t nb;
    NEED(r64(f, &nb) && nb == ix->n, "bwt length != sequences length");
    ix->bwt = malloc(nb);
    NEED(fread(ix->bwt, 1, nb, f) == nb, "truncated bwt");
    NEED(r64(f, &ix->less_len) && ix->less_len == ORC_LESS_LEN, "less length != 118");
    ix->less = malloc(ix->less_len * 8);
    NEED(fread(ix->less, 8, ix->less_len, f) == ix->less_len, "truncated less");
    NEED(r64(f, &ix->occ_outer) && ix->occ_outer == ORC_OCC_SYMS, "occ outer length != 117");
    for (uint64_t a = 0; a < ix->occ_outer; a++) {
        NEED(r64(f, &ix->occ_len[a]) && ix->occ_len[a] <= fsize / 8, "bad occ length");
        if (ix->occ_len[a]) {
            ix->occ[a] = malloc(ix->occ_len[a] * 8);
            NEED(fread(ix->occ[a], 8, ix->occ_len[a], f) == ix->occ_len[a], "truncated occ");
        }
    }
    NEED(r32(f, &ix->k) && ix->k > 0, "bad occ k");
    NEED(r64(f, &ix->nsample) && ix->nsample <= fsize / 8, "bad sample length");
    ix->sample = malloc((ix->nsample ? ix->nsample : 1) * 8);
    NEED(fread(ix->sample, 8, ix->nsample, f) == ix->nsample, "truncated sample");
    NEED(r64(f, &ix->s) && ix->s > 0, "bad s");
    NEED(ix->nsample == (ix->n + ix->s - 1) / ix->s, "sample length != ceil(n/s)");
    NEED(r64(f, &ix->n_extra) && ix->n_extra <= fsize / 16, "bad extra_rows length");
    ix->extra_key = malloc((ix->n_extra ? ix->n_extra : 1) * 8);
    ix->extra_val = malloc((ix->n_extra ? ix->n_extra : 1) * 8);
    for (uint64_t i = 0; i < ix->n_extra; i++)
        NEED(r64(f, &ix->extra_key[i]) && r64(f, &ix->extra_val[i]), "truncated extra_rows");
    NEED(fread(&ix->sentinel, 1, 1, f) == 1, "missing sentinel");
    NEED((uint64_t)ftell(f) == fsize, "trailing bytes");
    NEED(ix->text[ix->n - 1] == ix->sentinel, "sequences do not end with the sentinel");
    {
        uint64_t nchk = (ix->n - 1) / ix->k + 1;
        for (int a = 0; a < 11; a++)
            NEED(ix->occ_len[OCC_ALPHA[a]] == nchk, "occ checkpoint count != floor((n-1)/k)+1");
    }
#undef NEED
    fclose(f);
    return ix;
bad:
    fclose(f);
    orc_index_free(ix);
    return NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* FM primitives (bio)                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* Occ::get(bwt, r, a): occ[a][r/k] + count(bwt[(r/k)*k+1 ..= r] == a)  -- inclusive rank */
uint64_t orc_occ_get(const orc_index *ix, uint64_t r, uint8_t a) {
    uint64_t lo = r / ix->k;
    uint64_t cnt = ix->occ[a][lo];
    for (uint64_t p = lo * ix->k + 1; p <= r; p++) cnt += (ix->bwt[p] == a);
    return cnt;
}

/* FMIndex::backward_search (call site index.rs:305): (l,r) = (0,n-1); for a in pattern.rev():
 * l = less[a] + (l>0 ? occ(l-1,a) : 0); r = less[a] + occ(r,a) - 1; if l == r+1 break.
 * All symbols matched with a non-empty interval => Complete{lower:l, upper:r+1}. */
int orc_backward_search(const orc_index *ix, const uint8_t *pat, uint64_t len, uint64_t *lower,
                        uint64_t *upper, orc_counters *c) {
    uint64_t l = 0, r = ix->n - 1;
    uint64_t matched = 0;
    for (uint64_t i = len; i-- > 0;) {
        uint8_t a = pat[i];
        uint64_t less = ix->less[a];
        l = less + (l > 0 ? orc_occ_get(ix, l - 1, a) : 0);
        r = less + orc_occ_get(ix, r, a) - 1;
        if (c) c->X++;
        if (l == r + 1) break;
        matched++;
    }
    if (matched == len && len > 0 && l <= r) {
        *lower = l;
        *upper = r + 1;
        return 1;
    }
    *lower = *upper = 0; /* Partial / Absent leave interval_upper = interval_lower = 0, index.rs:310-331 */
    return 0;
}

/* SampledSuffixArray::get (call site index.rs:347 via Interval::occ): walk LF until a sampled row
 * (pos % s == 0) or the row whose BWT symbol is the sentinel (extra_rows) */
uint64_t orc_sa_get(const orc_index *ix, uint64_t row, orc_counters *c) {
    uint64_t pos = row, off = 0;
    for (;;) {
        if (pos % ix->s == 0) return ix->sample[pos / ix->s] + off;
        uint8_t ch = ix->bwt[pos];
        if (ch == ix->sentinel) {
            for (uint64_t i = 0; i < ix->n_extra; i++)
                if (ix->extra_key[i] == pos) return ix->extra_val[i] + off;
            return off; /* row with BWT '$' is the suffix at text position 0 */
        }
        pos = ix->less[ch] + orc_occ_get(ix, pos - 1, ch);
        off++;
        if (c) c->S++;
    }
}

uint64_t orc_brute_find(const orc_index *ix, const uint8_t *pat, uint64_t len, uint64_t *out,
                        uint64_t cap) {
    uint64_t cnt = 0;
    if (len == 0 || ix->n < len) return 0;
    for (uint64_t i = 0; i + len <= ix->n; i++)
        if (memcmp(ix->text + i, pat, len) == 0) {
            if (cnt < cap) out[cnt] = i;
            cnt++;
        }
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* Aligner::min_edit_distance, src/align.rs:28-85                                              */
/* ------------------------------------------------------------------------------------------ */
uint32_t orc_min_edit_distance(const uint8_t *p, uint64_t m, const uint8_t *t, uint64_t n) {
    uint64_t row_mult = n + 1;
    uint32_t *d = calloc((m + 1) * row_mult, sizeof(uint32_t));
    for (uint64_t i = 0; i < row_mult; i++) d[i] = 0;                 /* align.rs:37-41 */
    for (uint64_t row = 1; row <= m; row++) d[row * row_mult] = (uint32_t)row; /* :44-48 */
    for (uint64_t row = 1; row <= m; row++)
        for (uint64_t col = 1; col <= n; col++) {
            uint32_t delta = p[row - 1] != t[col - 1];
            uint32_t diag = d[(row - 1) * row_mult + col - 1] + delta;
            uint32_t up = d[(row - 1) * row_mult + col] + 1;
            uint32_t left = d[row * row_mult + col - 1] + 1;
            uint32_t v = up < left ? up : left;
            d[row * row_mult + col] = diag < v ? diag : v;
        }
    uint32_t best = d[m * row_mult];
    for (uint64_t col = 1; col <= n; col++)
        if (d[m * row_mult + col] < best) best = d[m * row_mult + col];
    free(d);
    return best;
}

/* ------------------------------------------------------------------------------------------ */
/* SW prefilter                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* Profile::sequence_to_numeric, ssw/src/lib.rs:88-105 */
static inline int8_t to_num(uint8_t b) {
    switch (b) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 4;
    }
}
/* IDENT_W_PENALTY_NO_N_MATCH, ssw/src/lib.rs:11-16: +1 on the diagonal (N/N included), -1 off */
static inline int mat_score(int a, int b) { return a == b ? 1 : -1; }

/* Textbook Gotoh local alignment with ssw's gap convention.  Independent of the striped layout;
 * used to show the byte kernel equals the exact score (tests) */
uint32_t orc_sw_exact(const uint8_t *read, uint64_t m, const uint8_t *ref, uint64_t n, int go,
                      int ge) {
    int32_t *H = calloc(m + 1, sizeof(int32_t));
    int32_t *E = calloc(m + 1, sizeof(int32_t)); /* gap along the reference direction */
    int32_t best = 0;
    for (uint64_t j = 0; j < n; j++) {
        int rc = to_num(ref[j]);
        int32_t diag = 0, F = 0;
        for (uint64_t i = 1; i <= m; i++) {
            int32_t h = diag + mat_score(to_num(read[i - 1]), rc);
            if (h < E[i]) h = E[i];
            if (h < F) h = F;
            if (h < 0) h = 0;
            diag = H[i];
            H[i] = h;
            if (h > best) best = h;
            int32_t e = E[i] - ge, ho = h - go;
            E[i] = e > ho ? e : ho;
            if (E[i] < 0) E[i] = 0;
            int32_t f = F - ge;
            F = f > ho ? f : ho;
            if (F < 0) F = 0;
        }
    }
    free(H);
    free(E);
    return (uint32_t)best;
}

/* --- literal emulation of sw_sse2_byte, ssw/src/ssw.c:123-328 (score only) --- */
typedef struct { uint8_t b[16]; } v16;
static inline v16 v16_zero(void) { v16 r; memset(&r, 0, sizeof r); return r; }
static inline v16 v16_set1(uint8_t x) { v16 r; memset(&r, x, sizeof r); return r; }
static inline v16 v16_adds(v16 a, v16 b) { v16 r; for (int i = 0; i < 16; i++) { int s = a.b[i] + b.b[i]; r.b[i] = s > 255 ? 255 : s; } return r; }
static inline v16 v16_subs(v16 a, v16 b) { v16 r; for (int i = 0; i < 16; i++) { int s = a.b[i] - b.b[i]; r.b[i] = s < 0 ? 0 : s; } return r; }
static inline v16 v16_max(v16 a, v16 b) { v16 r; for (int i = 0; i < 16; i++) r.b[i] = a.b[i] > b.b[i] ? a.b[i] : b.b[i]; return r; }
static inline v16 v16_shl1(v16 a) { v16 r; r.b[0] = 0; for (int i = 1; i < 16; i++) r.b[i] = a.b[i - 1]; return r; }
static inline int v16_all_zero(v16 a) { for (int i = 0; i < 16; i++) if (a.b[i]) return 0; return 1; }
static inline int v16_eq(v16 a, v16 b) { return memcmp(&a, &b, 16) == 0; }
static inline uint8_t v16_hmax(v16 a) { uint8_t m = 0; for (int i = 0; i < 16; i++) if (a.b[i] > m) m = a.b[i]; return m; }

uint32_t orc_ssw_byte(const uint8_t *read, uint64_t m, const uint8_t *ref, uint64_t n) {
    const uint8_t bias = 1; /* ssw.c:739-744: |min(mat)| */
    const uint8_t go = 1, ge = 1;
    int32_t segLen = (int32_t)((m + 15) / 16);
    /* qP_byte, ssw.c:89-114 */
    v16 *prof = malloc(5 * (size_t)segLen * sizeof(v16));
    for (int nt = 0; nt < 5; nt++)
        for (int i = 0; i < segLen; i++) {
            int64_t j = i;
            for (int seg = 0; seg < 16; seg++) {
                prof[nt * segLen + i].b[seg] =
                    (uint64_t)j >= m ? bias : (uint8_t)(mat_score(nt, to_num(read[j])) + bias);
                j += segLen;
            }
        }
    v16 *pvHStore = calloc(segLen, sizeof(v16)), *pvHLoad = calloc(segLen, sizeof(v16));
    v16 *pvE = calloc(segLen, sizeof(v16));
    v16 vGapO = v16_set1(go), vGapE = v16_set1(ge), vBias = v16_set1(bias);
    v16 vMaxScore = v16_zero(), vMaxMark = v16_zero();
    uint8_t max = 0;
    int overflow = 0;
    for (uint64_t i = 0; i < n; i++) {
        v16 e, vF = v16_zero(), vMaxColumn = v16_zero();
        v16 vH = v16_shl1(pvHStore[segLen - 1]);
        const v16 *vP = prof + to_num(ref[i]) * segLen;
        v16 *pv = pvHLoad; pvHLoad = pvHStore; pvHStore = pv;
        for (int j = 0; j < segLen; j++) {
            vH = v16_adds(vH, vP[j]);
            vH = v16_subs(vH, vBias);
            e = pvE[j];
            vH = v16_max(vH, e);
            vH = v16_max(vH, vF);
            vMaxColumn = v16_max(vMaxColumn, vH);
            pvHStore[j] = vH;
            vH = v16_subs(vH, vGapO);
            e = v16_subs(e, vGapE);
            e = v16_max(e, vH);
            pvE[j] = e;
            vF = v16_subs(vF, vGapE);
            vF = v16_max(vF, vH);
            vH = pvHLoad[j];
        }
        /* Lazy_F loop, ssw.c:227-258 (E is deliberately not updated) */
        int j = 0;
        vH = pvHStore[j];
        vF = v16_shl1(vF);
        v16 vTemp = v16_subs(vH, vGapO);
        vTemp = v16_subs(vF, vTemp);
        while (!v16_all_zero(vTemp)) {
            vH = v16_max(vH, vF);
            vMaxColumn = v16_max(vMaxColumn, vH);
            pvHStore[j] = vH;
            vF = v16_subs(vF, vGapE);
            j++;
            if (j >= segLen) { j = 0; vF = v16_shl1(vF); }
            vH = pvHStore[j];
            vTemp = v16_subs(vH, vGapO);
            vTemp = v16_subs(vF, vTemp);
        }
        vMaxScore = v16_max(vMaxScore, vMaxColumn);
        if (!v16_eq(vMaxMark, vMaxScore)) {
            vMaxMark = vMaxScore;
            uint8_t temp = v16_hmax(vMaxScore);
            if (temp > max) {
                max = temp;
                if (max + bias >= 255) { overflow = 1; break; } /* ssw.c:271 */
            }
        }
        /* maxColumn[i] == terminate(255) cannot precede the overflow break */
    }
    free(prof); free(pvHStore); free(pvHLoad); free(pvE);
    return (overflow || max + bias >= 255) ? 255u : max; /* ssw.c:302 */
}

/* --- literal emulation of sw_sse2_word, ssw/src/ssw.c:354-530 (score only) --- */
typedef struct { int16_t w[8]; } v8;
static inline v8 v8_zero(void) { v8 r; memset(&r, 0, sizeof r); return r; }
static inline v8 v8_set1(int16_t x) { v8 r; for (int i = 0; i < 8; i++) r.w[i] = x; return r; }
static inline v8 v8_adds(v8 a, v8 b) { v8 r; for (int i = 0; i < 8; i++) { int s = a.w[i] + b.w[i]; r.w[i] = s > 32767 ? 32767 : (s < -32768 ? -32768 : s); } return r; }
static inline v8 v8_subs_epu(v8 a, v8 b) { v8 r; for (int i = 0; i < 8; i++) { int s = (uint16_t)a.w[i] - (uint16_t)b.w[i]; r.w[i] = (int16_t)(uint16_t)(s < 0 ? 0 : s); } return r; }
static inline v8 v8_max(v8 a, v8 b) { v8 r; for (int i = 0; i < 8; i++) r.w[i] = a.w[i] > b.w[i] ? a.w[i] : b.w[i]; return r; }
static inline v8 v8_shl1(v8 a) { v8 r; r.w[0] = 0; for (int i = 1; i < 8; i++) r.w[i] = a.w[i - 1]; return r; }
static inline int v8_any_gt(v8 a, v8 b) { for (int i = 0; i < 8; i++) if (a.w[i] > b.w[i]) return 1; return 0; }

uint32_t orc_ssw_word(const uint8_t *read, uint64_t m, const uint8_t *ref, uint64_t n) {
    const int16_t go = 1, ge = 1;
    int32_t segLen = (int32_t)((m + 7) / 8);
    /* qP_word, ssw.c:330-352 */
    v8 *prof = malloc(5 * (size_t)segLen * sizeof(v8));
    for (int nt = 0; nt < 5; nt++)
        for (int i = 0; i < segLen; i++) {
            int64_t j = i;
            for (int seg = 0; seg < 8; seg++) {
                prof[nt * segLen + i].w[seg] =
                    (uint64_t)j >= m ? 0 : (int16_t)mat_score(nt, to_num(read[j]));
                j += segLen;
            }
        }
    v8 *pvHStore = calloc(segLen, sizeof(v8)), *pvHLoad = calloc(segLen, sizeof(v8));
    v8 *pvE = calloc(segLen, sizeof(v8));
    v8 vGapO = v8_set1(go), vGapE = v8_set1(ge);
    v8 vMaxScore = v8_zero();
    for (uint64_t i = 0; i < n; i++) {
        v8 e, vF = v8_zero();
        v8 vH = v8_shl1(pvHStore[segLen - 1]);
        v8 *pv = pvHLoad;
        v8 vMaxColumn = v8_zero();
        const v8 *vP = prof + to_num(ref[i]) * segLen;
        pvHLoad = pvHStore; pvHStore = pv;
        for (int j = 0; j < segLen; j++) {
            vH = v8_adds(vH, vP[j]);
            e = pvE[j];
            vH = v8_max(vH, e);
            vH = v8_max(vH, vF);
            vMaxColumn = v8_max(vMaxColumn, vH);
            pvHStore[j] = vH;
            vH = v8_subs_epu(vH, vGapO);
            e = v8_subs_epu(e, vGapE);
            e = v8_max(e, vH);
            pvE[j] = e;
            vF = v8_subs_epu(vF, vGapE);
            vF = v8_max(vF, vH);
            vH = pvHLoad[j];
        }
        /* Lazy_F loop, ssw.c:452-463: at most 8 passes, early-out when F can no longer win */
        for (int k = 0; k < 8; k++) {
            vF = v8_shl1(vF);
            for (int j = 0; j < segLen; j++) {
                vH = pvHStore[j];
                vH = v8_max(vH, vF);
                vMaxColumn = v8_max(vMaxColumn, vH);
                pvHStore[j] = vH;
                vH = v8_subs_epu(vH, vGapO);
                vF = v8_subs_epu(vF, vGapE);
                if (!v8_any_gt(vF, vH)) goto end;
            }
        }
    end:
        vMaxScore = v8_max(vMaxScore, vMaxColumn);
    }
    int16_t max = 0;
    for (int i = 0; i < 8; i++) if (vMaxScore.w[i] > max) max = vMaxScore.w[i];
    free(prof); free(pvHStore); free(pvHLoad); free(pvE);
    return (uint32_t)(uint16_t)max;
}

/* ssw_align with flag 0 as called from ssw/src/lib.rs:61-84: byte kernel, and the word kernel
 * iff the byte kernel reports 255 (ssw.c:787-792); score1 only */
uint32_t orc_ssw_score(const uint8_t *read, uint64_t m, const uint8_t *ref, uint64_t n) {
    uint32_t s = orc_ssw_byte(read, m, ref, n);
    if (s == 255) s = orc_ssw_word(read, m, ref, n);
    return s;
}

/* ------------------------------------------------------------------------------------------ */
/* SeedHit::candidate_indices, src/index.rs:118-153 (release-build wrapping arithmetic)        */
/* ------------------------------------------------------------------------------------------ */
int orc_candidate_indices(uint64_t site, uint64_t qoff, const orc_bin *bin, uint64_t read_len,
                          uint64_t edit_distance, uint64_t *start, uint64_t *end) {
    uint64_t start_offset = qoff + edit_distance;
    uint64_t cand_start =
        ((uint64_t)(site - start_offset) < bin->start || start_offset > site) ? bin->start
                                                                              : site - start_offset;
    uint64_t cand_end = site + (read_len - qoff) + edit_distance;
    if (cand_end > bin->end) cand_end = bin->end;
    if (cand_start > cand_end || cand_start < bin->start || cand_end > bin->end ||
        cand_end - cand_start < read_len - edit_distance)
        return 0;
    *start = cand_start;
    *end = cand_end;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* MGIndex::matching_tax_ids, src/index.rs:258-432                                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t ref, q; } seed_hit;     /* index.rs:108-113, Ord = (ref, q) */
typedef struct { uint64_t s, e, bin, nseeds; } cand_t; /* index.rs:159-165 */

static int seed_hit_cmp(const void *a, const void *b) {
    const seed_hit *x = a, *y = b;
    if (x->ref != y->ref) return x->ref < y->ref ? -1 : 1;
    if (x->q != y->q) return x->q < y->q ? -1 : 1;
    return 0;
}

/* stable merge sort by nseeds descending (refs.sort_by(|a,b| b.num_seeds.cmp(&a.num_seeds)),
 * index.rs:369; slice::sort_by is stable) */
static void cand_sort(cand_t *a, cand_t *tmp, uint64_t n) {
    if (n < 2) return;
    uint64_t h = n / 2;
    cand_sort(a, tmp, h);
    cand_sort(a + h, tmp, n - h);
    uint64_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].nseeds > a[i].nseeds) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof *a);
}

int64_t orc_matching_tax_ids(const orc_index *ix, const uint8_t *seq, uint64_t len,
                             const orc_params *p, orc_hit *hits, uint64_t cap, orc_counters *c) {
    orc_counters local;
    if (!c) { memset(&local, 0, sizeof local); c = &local; }
    /* reference behaviour for len == 0 is an assert (ssw/src/lib.rs:37), for len+1 < seed_size a
     * wrapped range and a slice panic (index.rs:284-286): trapped here as "no hits" */
    if (len == 0 || len + 1 < p->seed_size || p->seed_gap == 0 || p->seed_size == 0) return 0;

    /* index.rs:272-279 */
    uint8_t *seq_no_n = malloc(len);
    for (uint64_t i = 0; i < len; i++) seq_no_n[i] = seq[i] == 'N' ? '.' : seq[i];
    /* index.rs:281-282: IEEE double product then ceil */
    uint64_t edit_distance = (uint64_t)ceil((double)len * p->edit_rate);

    uint64_t hcap = 64, nh = 0;
    seed_hit *H = malloc(hcap * sizeof *H);
    double n_seeds = 0.0;
    uint64_t next_offset = 0, seed_interval = p->seed_gap;
    uint64_t range_end = len + 1 - p->seed_size; /* index.rs:284 */
    for (uint64_t offset = 0; offset < range_end; offset += p->seed_gap) {
        if (offset < next_offset) continue; /* index.rs:300-302 */
        uint64_t lo, hi;
        c->n_seed++;
        orc_backward_search(ix, seq + offset, p->seed_size, &lo, &hi, c); /* index.rs:305 */
        if (hi == 0 && lo == 0) continue;                                 /* index.rs:330-332 */
        uint64_t n_hits = hi - lo;
        if (n_hits > p->max_hits) continue; /* index.rs:335-337 */
        if (n_hits > p->tune_max_hits) {    /* index.rs:338-344 */
            seed_interval *= 2;
            next_offset = offset + seed_interval;
        }
        for (uint64_t r = lo; r < hi; r++) { /* index.rs:347-352 */
            if (nh == hcap) { hcap *= 2; H = realloc(H, hcap * sizeof *H); }
            H[nh].ref = orc_sa_get(ix, r, c);
            H[nh].q = offset;
            nh++;
            c->H++;
        }
        n_seeds += 1.0;
    }
    /* index.rs:358 */
    double ms = floor(n_seeds * p->min_seed);
    if (ms < 1.0) ms = 1.0;
    uint64_t min_seeds = (uint64_t)ms;

    /* coalesce_seed_sites, index.rs:435-487 */
    qsort(H, nh, sizeof *H, seed_hit_cmp);
    cand_t *C = malloc((nh ? nh : 1) * sizeof *C);
    uint64_t nc = 0;
    int have = 0;
    cand_t cur = {0, 0, 0, 0};
    uint64_t b = 0;
    for (uint64_t i = 0; i < nh; i++) {
        while (ix->bins[b].end <= H[i].ref) b++; /* index.rs:455-458 */
        uint64_t ws, we;
        int ok = orc_candidate_indices(H[i].ref, H[i].q, &ix->bins[b], len, edit_distance, &ws, &we);
        if (have) {
            /* add_seed_hit, index.rs:201-235 */
            if (ok && b == cur.bin && ((cur.s <= ws && ws < cur.e) || (cur.s < we && we <= cur.e))) {
                if (ws < cur.s) cur.s = ws;
                if (we > cur.e) cur.e = we;
                cur.nseeds++;
            } else {
                if (cur.nseeds >= min_seeds) C[nc++] = cur; /* index.rs:467-469 */
                have = ok;                                   /* index.rs:472 */
                if (ok) { cur.s = ws; cur.e = we; cur.bin = b; cur.nseeds = 1; }
            }
        } else {
            have = ok; /* index.rs:475 */
            if (ok) { cur.s = ws; cur.e = we; cur.bin = b; cur.nseeds = 1; }
        }
    }
    if (have && cur.nseeds >= min_seeds) C[nc++] = cur; /* index.rs:481-485 */
    c->n_cand += nc;
    cand_t *tmp = malloc((nc ? nc : 1) * sizeof *tmp);
    cand_sort(C, tmp, nc); /* index.rs:369 */
    free(tmp);

    /* index.rs:375-431 */
    uint64_t thr = len - 2 * edit_distance; /* usize arithmetic, wraps in a release build */
    uint64_t nm = 0, nout = 0, checked = 0;
    uint32_t *matched = malloc((nc ? nc : 1) * sizeof *matched);
    int64_t ret = 0;
    for (uint64_t i = 0; i < nc; i++) {
        if (p->max_candidates >= 0 && checked >= (uint64_t)p->max_candidates) break;
        checked++;
        const orc_bin *bin = &ix->bins[C[i].bin];
        int dup = 0;
        for (uint64_t k = 0; k < nm; k++)
            if (matched[k] == bin->tax_id) { dup = 1; break; }
        if (dup) continue;
        const uint8_t *w = ix->text + C[i].s;
        uint64_t wl = C[i].e - C[i].s;
        c->W += wl;
        c->n_sw++;
        uint64_t score = wl ? orc_ssw_score(seq, len, w, wl) : 0; /* index.rs:401-402 */
        if (score >= thr) {                                       /* index.rs:406 */
            c->n_edit++;
            uint32_t edits = orc_min_edit_distance(seq_no_n, len, w, wl); /* index.rs:409 */
            if (edits <= edit_distance) {
                matched[nm++] = bin->tax_id;
                if (nout >= cap) { ret = -1; break; }
                hits[nout].read = 0;
                hits[nout].strand = 0;
                hits[nout].tax_id = bin->tax_id;
                hits[nout].gi = bin->gi;
                hits[nout].offset = C[i].s >= bin->start ? C[i].s - bin->start : 0; /* :416 */
                hits[nout].edit = edits;
                nout++;
                c->R++;
                if (p->max_assignments >= 0 && nout >= (uint64_t)p->max_assignments) break;
            }
        }
    }
    free(matched);
    free(C);
    free(H);
    free(seq_no_n);
    return ret < 0 ? ret : (int64_t)nout;
}

/* ------------------------------------------------------------------------------------------ */
/* worker closure, src/binner.rs:77-131                                                        */
/* ------------------------------------------------------------------------------------------ */
static inline uint8_t norm_base(uint8_t b) { /* binner.rs:88-100 */
    switch (b) {
    case 'A': case 'a': return 'A';
    case 'C': case 'c': return 'C';
    case 'G': case 'g': return 'G';
    case 'T': case 't': return 'T';
    default: return 'N';
    }
}
static inline uint8_t comp_base(uint8_t b) { /* bio::alphabets::dna::revcomp on ACGTN */
    switch (b) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}

int orc_bin_batch(const orc_index *ix, const uint8_t *bases, const uint64_t *read_off,
                  uint64_t n_reads, const orc_params *p, int n_threads, orc_hit **hits_out,
                  uint64_t *n_hits_out, orc_counters *ctr) {
    orc_hit **per_read = calloc(n_reads ? n_reads : 1, sizeof *per_read);
    uint32_t *per_cnt = calloc(n_reads ? n_reads : 1, sizeof *per_cnt);
    orc_counters total;
    memset(&total, 0, sizeof total);
    int failed = 0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        orc_counters c;
        memset(&c, 0, sizeof c);
        uint64_t cap = ix->nbins ? ix->nbins : 1;
        orc_hit *buf = malloc(2 * cap * sizeof *buf);
        uint8_t *fwd = NULL, *rev = NULL;
        uint64_t bcap = 0;
#pragma omp for schedule(dynamic, 256)
        for (int64_t r = 0; r < (int64_t)n_reads; r++) {
            uint64_t len = read_off[r + 1] - read_off[r];
            if (len > bcap) {
                bcap = len * 2;
                fwd = realloc(fwd, bcap);
                rev = realloc(rev, bcap);
            }
            const uint8_t *src = bases + read_off[r];
            for (uint64_t i = 0; i < len; i++) fwd[i] = norm_base(src[i]);
            for (uint64_t i = 0; i < len; i++) rev[i] = comp_base(fwd[len - 1 - i]); /* :115 */
            c.Lsum += len;
            int64_t nf = orc_matching_tax_ids(ix, fwd, len, p, buf, cap, &c);
            if (nf < 0) { failed = 1; continue; }
            int64_t nr = orc_matching_tax_ids(ix, rev, len, p, buf + nf, cap, &c);
            if (nr < 0) { failed = 1; continue; }
            for (int64_t i = 0; i < nf + nr; i++) {
                buf[i].read = (uint64_t)r;
                buf[i].strand = i >= nf;
            }
            if (nf + nr) {
                per_read[r] = malloc((nf + nr) * sizeof(orc_hit));
                memcpy(per_read[r], buf, (nf + nr) * sizeof(orc_hit)); /* fwd ++ rev, :128 */
                per_cnt[r] = (uint32_t)(nf + nr);
            }
        }
        free(buf);
        free(fwd);
        free(rev);
#pragma omp critical
        {
            total.X += c.X; total.S += c.S; total.H += c.H; total.W += c.W; total.R += c.R;
            total.Lsum += c.Lsum; total.n_sw += c.n_sw; total.n_edit += c.n_edit;
            total.n_cand += c.n_cand; total.n_seed += c.n_seed;
        }
    }
    uint64_t tot = 0;
    for (uint64_t r = 0; r < n_reads; r++) tot += per_cnt[r];
    orc_hit *out = malloc((tot ? tot : 1) * sizeof *out);
    uint64_t k = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        if (per_cnt[r]) memcpy(out + k, per_read[r], per_cnt[r] * sizeof(orc_hit));
        k += per_cnt[r];
        free(per_read[r]);
    }
    free(per_read);
    free(per_cnt);
    if (ctr) *ctr = total;
    *hits_out = out;
    *n_hits_out = tot;
    if (failed) { FAIL("hit buffer overflow"); return -1; }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* write_assignments, src/binner.rs:310-379                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t tax, gi; uint64_t off; uint32_t edit; } fmt_item;
static int fmt_cmp_default(const void *a, const void *b) {
    const fmt_item *x = a, *y = b;
    if (x->tax != y->tax) return x->tax < y->tax ? -1 : 1;
    return x->edit < y->edit ? -1 : (x->edit > y->edit);
}
static int fmt_cmp_long(const void *a, const void *b) {
    const fmt_item *x = a, *y = b;
    if (x->tax != y->tax) return x->tax < y->tax ? -1 : 1;
    if (x->gi != y->gi) return x->gi < y->gi ? -1 : 1;
    if (x->off != y->off) return x->off < y->off ? -1 : 1;
    return x->edit < y->edit ? -1 : (x->edit > y->edit);
}

int64_t orc_format_line(const char *read_id, const orc_hit *hits, uint64_t n_hits, int long_format,
                        char *buf, uint64_t cap) {
    if (n_hits == 0) return 0; /* binner.rs:316-318 */
    fmt_item *it = malloc(n_hits * sizeof *it);
    uint64_t ni = 0;
    for (uint64_t i = 0; i < n_hits; i++) {
        uint64_t k;
        for (k = 0; k < ni; k++) {
            int same = long_format ? (it[k].tax == hits[i].tax_id && it[k].gi == hits[i].gi &&
                                      it[k].off == hits[i].offset)
                                   : (it[k].tax == hits[i].tax_id);
            if (same) break;
        }
        if (k == ni) {
            it[ni].tax = hits[i].tax_id;
            it[ni].gi = hits[i].gi;
            it[ni].off = hits[i].offset;
            it[ni].edit = hits[i].edit;
            ni++;
        } else if (hits[i].edit < it[k].edit)
            it[k].edit = hits[i].edit;
    }
    qsort(it, ni, sizeof *it, long_format ? fmt_cmp_long : fmt_cmp_default);
    uint64_t pos = 0;
    int n = snprintf(buf, cap, "%s:", read_id);
    if (n < 0 || (uint64_t)n >= cap) { free(it); return -1; }
    pos = (uint64_t)n;
    for (uint64_t k = 0; k < ni; k++) {
        if (long_format)
            n = snprintf(buf + pos, cap - pos, "%s%u-%u-%llu=%u", k ? "," : "", it[k].tax, it[k].gi,
                         (unsigned long long)it[k].off, it[k].edit);
        else
            n = snprintf(buf + pos, cap - pos, "%s%u=%u", k ? "," : "", it[k].tax, it[k].edit);
        if (n < 0 || (uint64_t)n >= cap - pos) { free(it); return -1; }
        pos += (uint64_t)n;
    }
    free(it);
    if (pos + 2 > cap) return -1;
    buf[pos++] = '\n';
    buf[pos] = 0;
    return (int64_t)pos;
}

/*
 * mtsv_amd.h -- C ABI of the MI355X-native mtsv-binner hot path (libmtsv_amd.so).
 *
 * Plain C: opaque handles, pointers and sizes only.  A Rust host binds it the way the reference
 * already binds its only native dependency (extern "C" block + #[repr(C)] mirrors + RAII Drop:
 * ssw/src/lib.rs:108-154, :26-32); INTEGRATION.md shows the stub.
 *
 * Every int-returning function returns 0 on success and a negative MTSV_E_* code otherwise;
 * mtsv_last_error() returns a thread-local message.  There is NO CPU fallback: a call that needs
 * the GPU fails with MTSV_E_DEVICE when no gfx950 device / HIP runtime is usable.
 *
 * What each entry point replaces in the reference (paths relative to FofanovLab/mtsv_tools):
 *   mtsv_index_load        io::from_file::<MGIndex>                      src/io.rs:115-123, src/binner.rs:63
 *   mtsv_index_to_device   FMIndex::new(bwt, less, occ) (borrow -> HBM)  src/binner.rs:64-67
 *   mtsv_bin_batch         the worker closure: normalise, matching_tax_ids(fwd), revcomp,
 *                          matching_tax_ids(rev), chain                   src/binner.rs:77-131,
 *                          MGIndex::matching_tax_ids                      src/index.rs:258-432
 *   mtsv_hit               Hit {tax_id, gi, offset, edit}                 src/index.rs:30-40
 *   mtsv_params            matching_tax_ids' scalar arguments             src/index.rs:258-269
 *                          (defaults: src/bin/mtsv-binner.rs:63-94)
 *   mtsv_format_results    write_assignments                              src/binner.rs:310-379
 *   mtsv_index_build*, mtsv_index_write
 *                          MGIndex::new + io::write_to_file               src/index.rs:491-582, src/io.rs:125-133
 *   mtsv_batch_*           the same path as mtsv_bin_batch, split so a host can keep read
 *                          batches resident in HBM and overlap upload / run / download
 *                          (replaces the bounded queue of vendor/cue/src/lib.rs:45-105)
 */
#ifndef MTSV_AMD_H
#define MTSV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTSV_OK 0
#define MTSV_E_ARG (-1)      /* invalid argument */
#define MTSV_E_IO (-2)       /* file cannot be opened / read / written */
#define MTSV_E_FORMAT (-3)   /* MG-index bytes violate the bincode layout or its invariants */
#define MTSV_E_DEVICE (-4)   /* no usable HIP device, HIP error, or device-side capacity */
#define MTSV_E_LIMIT (-5)    /* input exceeds a documented limit of the device layout */
#define MTSV_E_NOMEM (-6)

typedef struct mtsv_index mtsv_index; /* host copy of the MG-index + per-device HBM layout */
typedef struct mtsv_batch mtsv_batch; /* device workspace for one batch of reads on one GPU */

/* scalar arguments of MGIndex::matching_tax_ids (src/index.rs:258-269) */
typedef struct {
    double edit_rate;        /* -e / --edit-rate      0.13  */
    uint32_t seed_size;      /* --seed-size           18    */
    uint32_t seed_interval;  /* --seed-interval       15    */
    double min_seed;         /* --min-seed            0.015 */
    uint64_t max_hits;       /* --max-hits            2000  */
    uint64_t tune_max_hits;  /* --tune-max-hits       200   */
    int64_t max_assignments; /* --max-assignments     -1 = None */
    int64_t max_candidates;  /* --max-candidates      -1 = None */
} mtsv_params;

/* one Hit (src/index.rs:30-40) tagged with the read it belongs to and the strand call
 * (0 = forward, binner.rs:102; 1 = reverse complement, binner.rs:116) that produced it */
typedef struct {
    uint64_t read;   /* index of the read inside the batch */
    uint32_t tax_id;
    uint32_t gi;
    uint32_t edit;
    uint8_t strand;
    uint8_t _pad[3];
    uint64_t offset; /* candidate start - bin start (index.rs:416) */
} mtsv_hit;

/* shape of a loaded index */
typedef struct {
    uint64_t n;        /* symbols in `sequences`, including the trailing '$' */
    uint64_t n_bins;   /* reference sequences (GI records) */
    uint32_t occ_k;    /* Occ sampling interval of the file */
    uint64_t sa_s;     /* suffix-array sampling interval of the file */
    uint64_t file_bytes;
    uint64_t device_bytes; /* HBM held on the device it was last uploaded to (0 if none) */
    uint32_t kmer_k;       /* symbols of the resident k-mer interval table on that device (0: none) */
    uint32_t sa_full;      /* 1 when the full suffix array is resident there */
} mtsv_index_info_t;

/* per-stage device time of the last mtsv_batch_run, measured with HIP events on the batch's
 * own stream, plus the work counters the roofline accounting needs */
#define MTSV_N_STAGES 8
typedef struct {
    float stage_ms[MTSV_N_STAGES]; /* search, thin+scan, expand, locate, coalesce, verify, gather, total */
    uint64_t n_reads;
    uint64_t n_seed_slots;  /* seeds searched (both strands) */
    uint64_t n_seed_hits;   /* located seed hits */
    uint64_t lf_steps;      /* LF steps walked by k_locate (0 when the full SA is resident) */
    uint64_t n_candidates;  /* candidates after coalescing + min_seeds filter */
    uint64_t n_verified;    /* candidates whose window went through the SW prefilter */
    uint64_t window_bytes;  /* summed window length of those */
    uint64_t n_hits;        /* hits returned */
    uint64_t n_passes;      /* >1 when the batch had to be split to fit the hit workspace */
    uint64_t n_rounds;      /* verification rounds of the last pass: 1 + rounds that took the successors of
                             * candidates that passed the SW prefilter and failed the edit distance */
    uint64_t n_lanes;       /* concurrent parts the batch ran as (MTSV_LANES, default 3 for >= 98304 reads): with
                             * more than one, stage_ms[0..6] are device times summed over the overlapping parts
                             * and stage_ms[7] is the wall time of the run */
    uint64_t sw_cell_pairs; /* k_sw_pairs: DP cell pairs its sweeps computed, per 16-lane group (one packed
                             * 7-instruction recurrence each; 4 groups share a wave instruction) */
    float sw_prefilter_ms;  /* device time of the SW prefilter's kernels in the first round of every pass (HIP events on
                             * the lane's stream), summed over lanes like stage_ms; part of stage_ms[5] */
    float sw_sweep_ms;      /* of that, the DP sweeps (k_sw_pairs); the rest of sw_prefilter_ms is sw_diag_ms + sw_bound_ms */
    uint64_t n_sw_passed;   /* candidates whose SW score reached the threshold of index.rs:406, i.e. the edit
                             * distances the reference computes (:407-409).  Counted where the prefilter is a
                             * kernel of its own (reference order, reads up to 253 bases); 0 otherwise */
    float sw_diag_ms;       /* k_sw_diag: the lower bounds on the seed diagonal */
    float sw_bound_ms;      /* k_edit_myers in bound mode: the two-sided bound by the unit-cost edit distance */
    float edit_ms;          /* k_edit_myers on the candidates that passed the prefilter (first round of every pass) */
    uint64_t myers_columns; /* window columns the bit-vector recurrences of k_edit_myers advanced (bound + edit distance) */
    uint64_t n_sw_bound_refuted; /* candidates the edit-distance bound refuted without a sweep */
} mtsv_batch_stats;

const char *mtsv_last_error(void);
const char *mtsv_version(void);
void mtsv_params_default(mtsv_params *p);

/* number of visible HIP devices (0 when none / no runtime); never fails */
int mtsv_device_count(void);

/* ---- MG-index: load / build / write ----------------------------------------------------- */
int mtsv_index_load(const char *path, mtsv_index **out);
/* MGIndex::new: entries in database insertion order; sorted by tax_id (stable) like the BTreeMap */
int mtsv_index_build(uint64_t n_seqs, const uint32_t *tax_ids, const uint32_t *gis,
                     const uint8_t *const *seqs, const uint64_t *seq_lens, uint32_t occ_k,
                     uint64_t sa_s, int n_threads, mtsv_index **out);
/* mtsv-build: FASTA with `SEQID-TAXID` headers (src/util.rs:26-56, src/io.rs:135-150) */
int mtsv_index_build_fasta(const char *fasta_path, uint32_t occ_k, uint64_t sa_s, int n_threads,
                           mtsv_index **out);
/* Where later mtsv_index_build* / mtsv_synth_index calls run the suffix sort: a HIP device ordinal (prefix
 * doubling in HBM, ~32 B per symbol) or -1 = host threads (default).  The index bytes are the same. */
int mtsv_set_build_device(int hip_device);
int mtsv_index_write(const mtsv_index *ix, const char *path);
int mtsv_index_info(const mtsv_index *ix, mtsv_index_info_t *info);
void mtsv_index_free(mtsv_index *ix);

/* Re-pack the FM-index into the HBM layout of DESIGN.md and upload it to `hip_device`.
 * flags: MTSV_DEV_* below.  Idempotent per device. */
#define MTSV_DEV_DEFAULT 0u
#define MTSV_DEV_SAMPLED_SA_ONLY 1u /* keep the file's row-sampled SA only (LF-walk locate) */
#define MTSV_DEV_NO_KMER_TABLE 2u   /* no seed-suffix interval table */
int mtsv_index_to_device(mtsv_index *ix, int hip_device, uint32_t flags);

/* ---- the hot path ----------------------------------------------------------------------- */
/* bases: concatenated raw read bytes (any case, any byte), read_off[n_reads+1].
 * *hits is ordered by (read, strand, rank order of the reference's candidate loop) and owned by
 * the caller (mtsv_hits_free); the array is page-locked host memory from a pool the library
 * recycles (the hits of each slice are copied there while later slices still compute). */
int mtsv_bin_batch(mtsv_index *ix, int hip_device, const uint8_t *bases, const uint64_t *read_off,
                   uint64_t n_reads, const mtsv_params *params, mtsv_hit **hits, uint64_t *n_hits);
void mtsv_hits_free(mtsv_hit *hits);
/* Page-locked host memory for read buffers (the reference fills plain Vec<u8> buffers, src/binner.rs:21-66; a host
 * that keeps its parsed reads in memory from here spares every call a copy).  mtsv_bin_batch*, mtsv_batch_upload and
 * mtsv_batch_run_host recognise bases that lie in such memory and let the GPU's copy engine read them where they
 * are; any other memory is staged through the library's own page-locked buffers first (a host memcpy per slice).
 * mtsv_host_register page-locks memory the caller already owns (page-aligned ranges are best; it must be
 * unregistered before it is freed).  Both work without an index or a batch; NULL / an error code when no HIP
 * device is usable. */
void *mtsv_host_alloc(size_t bytes);
void mtsv_host_free(void *p);
int mtsv_host_register(void *p, size_t bytes);
int mtsv_host_unregister(void *p);
/* ---- several GPUs of one node (no collective: reads are independent) --------
 * Mode A -- the reference's single-index workflow (README.md:69-73) on several GPUs: the index is replicated
 * on every listed device, the reads are cut into n_devices contiguous blocks, one host thread and workspace per
 * entry, hits concatenated in read order (hit.read indexes the whole batch).  A device may be listed more than
 * once (two workspaces on it).  Output identical to mtsv_bin_batch. */
int mtsv_bin_batch_multi(mtsv_index *ix, const int *devices, int n_devices, const uint8_t *bases,
                         const uint64_t *read_off, uint64_t n_reads, const mtsv_params *params,
                         mtsv_hit **hits, uint64_t *n_hits);
/* Mode B -- the chunked-database workflow (mtsv-chunk, then one binner run per chunk, then mtsv-collapse:
 * README.md:189, src/collapse.rs:597-625): chunk k of the database is resident on devices[k], every chunk sees
 * every read, and the per-chunk hit lists are merged per read (chunk order within a read).
 * mtsv_format_results on the merged list writes the line mtsv-collapse would produce from the per-chunk
 * result files: smallest edit per (read, TaxId). */
int mtsv_bin_batch_chunks(mtsv_index *const *chunks, const int *devices, int n_chunks,
                          const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads,
                          const mtsv_params *params, mtsv_hit **hits, uint64_t *n_hits);
/* reads per device workspace mtsv_bin_batch creates (and keeps) for a host batch of n_reads reads; larger
 * batches stream through it in slices */
uint64_t mtsv_bin_batch_workspace_reads(uint64_t n_reads);

/* The same path with the batch resident in HBM (what bench.py times).  The index must already be
 * on `hip_device`.  max_hits_ws = seed-hit workspace entries (0 = default). */
int mtsv_batch_create(mtsv_index *ix, int hip_device, uint64_t max_reads, uint64_t max_bases,
                      uint64_t max_hits_ws, mtsv_batch **out);
/* The same with the number of LANES chosen: a host batch (mtsv_batch_run_host*) is worked off in ranges, `lanes` of them
 * at a time, each on a stream and a set of work arrays of its own (0: the default, three).  A host that keeps several
 * calls in flight on one device -- a workspace per worker thread, as mtsv-binner does -- asks for one lane each: its
 * calls are what overlap.  (No counterpart in the reference: src/binner.rs:57-76 is a pool of CPU threads.) */
int mtsv_batch_create_lanes(mtsv_index *ix, int hip_device, uint64_t max_reads, uint64_t max_bases,
                            uint64_t max_hits_ws, int lanes, mtsv_batch **out);
/* Make the workspace ready for host batches of up to n_reads reads / n_bases bases: what mtsv_batch_run_host* would
 * otherwise size on its first calls (device arenas, the page-locked offset table, a result array) is created now, and
 * with warm_read_len > 0 a small batch of reads of that length sampled from the index runs through every kernel (the
 * first launch of a kernel loads its code object).  Part of the device set-up, like mtsv_index_to_device. */
int mtsv_batch_reserve_host(mtsv_batch *b, uint64_t n_reads, uint64_t n_bases, uint32_t warm_read_len);
/* Order in which the two acceptance predicates of index.rs:406,410 are evaluated (results identical):
 *   MTSV_VERIFY_REFERENCE   SW prefilter score and edit distance for every verified candidate (default)
 *   MTSV_VERIFY_EDIT_FIRST  edit distance first; for reads <= 253 bases edits <= ED implies the SW
 *                           threshold, so the SW sweep is not needed (longer reads use REFERENCE)
 * The environment variable MTSV_VERIFY=reference|edit_first sets the default of new batches. */
#define MTSV_VERIFY_REFERENCE 0
#define MTSV_VERIFY_EDIT_FIRST 1
int mtsv_batch_set_verify_mode(mtsv_batch *b, int mode);
/* process-wide default of workspaces created from now on (also those mtsv_bin_batch* keep); the environment
 * variable, when set, wins */
int mtsv_set_default_verify_mode(int mode);
int mtsv_batch_upload(mtsv_batch *b, const uint8_t *bases, const uint64_t *read_off,
                      uint64_t n_reads);
int mtsv_batch_run(mtsv_batch *b, const mtsv_params *params); /* synchronous: returns when done */
/* upload + run in one call for host buffers of any size: the reads are cut into slices of at most
 * the workspace's max_reads / max_bases, and slice k+1 is copied to the device while slice k runs
 * (two input buffers, a copy stream), so the PCIe transfer hides behind the kernels.  Hit `read`
 * fields index the whole host batch.  Stats cover all slices. */
int mtsv_batch_run_host(mtsv_batch *b, const uint8_t *bases, const uint64_t *read_off,
                        uint64_t n_reads, const mtsv_params *params);
/* The same for a batch that lies in n_parts pieces (a host that parses its input in blocks hands several blocks to one
 * call -- larger passes on the device -- without putting them together first): part k holds n_reads[k] reads, bases[k]
 * with read_off[k][0 .. n_reads[k]]; the reads are numbered through the parts in order, and mtsv_batch_download returns
 * one hit list over all of them.  (The reference has no such call: its worker closure takes one read,
 * src/binner.rs:77-131.) */
int mtsv_batch_run_host_parts(mtsv_batch *b, int n_parts, const uint8_t *const *bases, const uint64_t *const *read_off,
                              const uint64_t *n_reads, const mtsv_params *params);
int mtsv_batch_stats_get(const mtsv_batch *b, mtsv_batch_stats *st);
int mtsv_batch_download(mtsv_batch *b, mtsv_hit **hits, uint64_t *n_hits);
void mtsv_batch_free(mtsv_batch *b);

/* ---- result lines (host) ---------------------------------------------------------------- */
/* write_assignments for a whole batch: hits ordered by read; ids = NUL-separated read ids,
 * id_off[n_reads+1].  Lines are appended to a malloc'd buffer (*out, *out_len), one per read with
 * >= 1 hit, in read order.  long_format = --output-format long. */
int mtsv_format_results(const mtsv_hit *hits, uint64_t n_hits, const char *ids,
                        const uint64_t *id_off, uint64_t n_reads, int long_format, char **out,
                        uint64_t *out_len);
void mtsv_free(void *p);

/* ---- the transfer format of mtsv_batch_run_host*, exposed for tests (not part of the drop-in) ---- */
/* The bases of a host batch cross PCIe as 4-bit codes, two per byte (src/binner.rs:88-100 applied on the host: A/a C/c
 * G/g T/t -> 0..3, any other byte -> 4): src[0, n) are the bases at offsets [first_offset, first_offset + n) of a segment,
 * dst receives bytes [first_offset / 2, (first_offset + n + 1) / 2) of its packed image (base i in nibble i & 1 of byte
 * i / 2); prev_code is the code of the base before the first (it shares the first byte when first_offset is odd).
 * Returns the code of the last base. */
uint8_t mtsv_pack_bases(uint8_t *dst, const uint8_t *src, uint64_t first_offset, uint64_t n, uint8_t prev_code);
/* Host threads that pack the bases of a host batch in this process (the CPUs it may use less four, ten at most;
 * MTSV_PACK_THREADS overrides); 0: the bytes go as they are (MTSV_H2D_PLAIN=1, or fewer than nine threads). */
int mtsv_host_pack_threads(void);

/* ---- synthetic workloads for bench.py / tests (not part of the drop-in) ---- */
/* i.i.d. ACGT reference of n_taxa x gis_per_taxon sequences of seq_len, 5% of each overwritten by
 * a 1%-diverged copy from another taxon, 0.1% of positions in N runs; built straight into an index */
int mtsv_synth_index(uint64_t seed, uint32_t n_taxa, uint32_t gis_per_taxon, uint64_t seq_len,
                     uint32_t occ_k, uint64_t sa_s, int n_threads, mtsv_index **out);
/* n_reads reads of read_len sampled from the index text (90%: sub 1%, ins 0.1%, del 0.1%, N 0.2%,
 * half reverse-complemented; 10% random).  bases must hold n_reads*read_len bytes. */
int mtsv_synth_reads(const mtsv_index *ix, uint64_t seed, uint64_t n_reads, uint32_t read_len,
                     uint8_t *bases, uint64_t *read_off);

#ifdef __cplusplus
}
#endif
#endif /* MTSV_AMD_H */

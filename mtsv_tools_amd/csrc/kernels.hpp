// kernels.hpp -- launch interface of the hot-path kernels.
//
// The mtsv-binner hot path as gfx950 kernels (k_seed.hip, k_coalesce.hip, k_verify.hip).
//
// One batch of reads flows through staged kernels with worklists in HBM:
//
//   k_search    lane per (read, strand, seed): FMIndex::backward_search           index.rs:305
//   k_thin      lane per strand: adaptive seed thinning / max_hits filter         index.rs:293-344,354
//   scan        exclusive scan of per-strand seed-hit counts
//   k_expand    lane per kept seed: its SA rows / text positions (Interval::occ)  index.rs:347-352
//   k_locate    lane per seed hit with wavefront refill: SampledSuffixArray::get  index.rs:347
//   k_coalesce  wavefront per strand: sort, coalesce_seed_sites, min_seeds, rank  index.rs:358-369,435-487
//   k_sw_pairs  16-lane group per two candidates: the SW prefilter                index.rs:401-406, ssw.c:123-328
//   k_edit_myers  lane per candidate that passed it: bit-vector edit distance     index.rs:407-410, align.rs:28-85
//               (rounds follow the same-TaxId chains of the ordered loop)
//   k_evaluate  longer reads (tiled beyond 256 bases): both in one sweep + sw_sse2_word  ssw.c:354-530
//   k_resolve   lane per strand: cut-offs and rank order of the selection loop    index.rs:384-428
//   scan + k_gather  compact per-strand hits into (read, strand, rank) order      binner.rs:128
//
// All arithmetic is integer; positions are u32 (n < 2^32).  No MFMA: the path is rank queries and
// small dynamic programs.
#pragma once
#include <algorithm>

#include "dev_layout.hpp"

namespace mtsv {

struct DevHit {  // byte-identical to mtsv_hit (include/mtsv_amd.h)
    uint64_t read;
    uint32_t tax_id;
    uint32_t gi;
    uint32_t edit;
    uint8_t strand;
    uint8_t pad[3];
    uint64_t offset;
};
static_assert(sizeof(DevHit) == 32, "DevHit must match mtsv_hit");

// slots of a lane's counter block (batch.hip: d_counters) that kernels address by number
constexpr uint32_t kCtrVerified = 4, kCtrWindowBytes = 5, kCtrSwCursor = 8, kCtrPassCount = 9, kCtrSwCellPairs = 14;

struct EvalArgs {
    const uint8_t* bases;
    const uint32_t* read_off;
    uint32_t r0;
    double edit_rate;
    int64_t max_candidates;
    const uint32_t* strand_off;
    const uint4* cand;            // (start, end, bin, strand) in rank order per strand
    const uint32_t* cand_next;    // rank of the next candidate with the same TaxId, 0xffffffff = none
    uint32_t* cand_status;        // 0 not verified, 1 failed, 2 passed
    const uint32_t* worklist;     // candidate indices of this round
    const uint32_t* wl_count;
    uint32_t* wl_cursor;         // dynamic scheduling: next unclaimed worklist position
    uint32_t wl_reverse = 0;     // k_sw_pairs: consume the worklist from its end
    // k_sw_pairs: the lane's counter block (kCtr* slots below) and the slot whose low word holds its worklist length
    uint64_t* counters = nullptr;
    uint32_t wl_count_slot = 0;
    uint4* out;                   // out[candidate] = (tax_id, gi, offset, edit) when it passed
    unsigned long long* n_verified;
    unsigned long long* window_bytes;
    unsigned long long* sw_columns = nullptr;  // k_sw_pairs: packed DP cell pairs swept (group columns x rows per lane)
    // reference order, reads <= 253 bases: k_sw_pairs appends candidates that pass the prefilter to
    // pass_list; k_edit_myers (list mode) verifies them and appends the successors of those that fail
    // the edit distance to next_list, the worklist of the next round
    uint32_t* pass_list = nullptr;
    uint32_t* pass_count = nullptr;
    uint32_t* next_list = nullptr;
    uint32_t* next_count = nullptr;
    unsigned long long* myers_ctr = nullptr;  // k_edit_myers: [0] += columns its recurrences advanced, [1] += candidates its bound refuted,
                                              // [2] += successors the list mode's own bound passed (they count as sent on to the edit distance)
    uint32_t* und_list = nullptr;  // k_sw_pairs TOP / k_edit_myers bound mode: candidates they do not decide
    uint32_t und_slot = 0;
    // tiled long-read kernel: one strip of strip_len window columns per 16-lane group (bottom row of a band)
    uint2* strip = nullptr;
    uint32_t strip_len = 0;
    // filled in by the launchers
    uint32_t maxc = 0xffffffffu;  // max_candidates as a bound on candidate ranks
    uint32_t claim_shift = 0;     // k_sw_pairs: work items per claim = clamp(n_work >> claim_shift, 4, 32)
};

constexpr uint32_t kMaxRegisterReadLen = 256;  // 16 lanes x 16 read rows per lane: k_evaluate with the whole matrix band in registers
constexpr uint32_t kMaxReadLen = 32767;        // the tiled kernel's packed 16-bit cells (edit distance <= read length)

// n <= 64 counters from HBM into mapped page-locked host memory, by a kernel on stream s (no copy engine involved)
void launch_publish(hipStream_t s, const uint64_t* src, uint64_t* dst_host, uint32_t n);
// base normalisation of bytes [begin, end) of a read buffer, src -> dst (may be equal): every other kernel expects codes
void launch_normalise(hipStream_t s, const uint8_t* src, uint8_t* dst, uint64_t begin, uint64_t end);
// run_host's transfer format (4-bit codes, host_pack.hpp) into byte codes: bytes [lo, hi) of dst from packed[lo / 2 ...]
void launch_unpack(hipStream_t s, const uint8_t* packed, uint8_t* dst, uint64_t lo, uint64_t hi);
void launch_search(hipStream_t s, const DevIndexView& ix, const uint8_t* bases, const uint32_t* read_off, uint32_t r0,
                   uint32_t n_reads, uint32_t max_ns, uint32_t K, uint32_t G, uint32_t* seed_lo, uint32_t* seed_cnt,
                   uint32_t* slow_list, uint32_t* slow_count, uint32_t listed_cap);
// slow_list: room for every seed slot; *slow_count: a counter of the lane; listed_cap: list entries the second kernel's grid covers
// (the caller compares *slow_count with it afterwards)
// strand_nseeds receives one word per strand for the coalescing kernels: min_seeds (index.rs:358) | edit tolerance << 16 |
// a flag for the strands no candidate of which can be accepted (more N in the read than the edit tolerance, or the usize
// wrap of index.rs:406)
void launch_thin(hipStream_t s, const uint8_t* bases, const uint32_t* read_off, uint32_t r0, uint32_t n_reads, double edit_rate,
                 double min_seed, uint32_t max_ns, uint32_t K, uint32_t G, uint64_t max_hits, uint64_t tune, uint32_t* seed_cnt, uint32_t* seed_pre,
                 uint32_t* strand_hits, uint32_t* strand_nseeds);
// out has n+1 entries (out[n] = total); tile_sums needs scan_tiles(n) entries
void launch_scan(hipStream_t s, const uint32_t* in, uint32_t n, uint64_t* tile_sums, uint64_t* total, uint32_t* out);
uint32_t scan_tiles(uint32_t n);
void launch_expand(hipStream_t s, const DevIndexView& ix, uint32_t n_strands, uint32_t max_ns, uint32_t G,
                   const uint32_t* seed_lo, const uint32_t* seed_cnt, const uint32_t* seed_pre, const uint32_t* strand_off,
                   uint32_t* hit_row, uint32_t* hit_ref, uint32_t* hit_q);
void launch_locate(hipStream_t s, const DevIndexView& ix, uint32_t total_hits_host, const uint32_t* total_hits_dev,
                   const uint32_t* hit_row, uint32_t* hit_ref, unsigned long long* lf_steps);
void launch_coalesce(hipStream_t s, const DevIndexView& ix, const uint32_t* read_off, uint32_t r0, uint32_t n_strands,
                     int64_t max_candidates, const uint32_t* strand_off,
                     const uint32_t* strand_nseeds, const uint32_t* hit_ref, const uint32_t* hit_q, uint64_t* hit_key,
                     uint64_t* cand_tmp, uint4* cand, uint32_t* cand_next, uint32_t* cand_status,
                     uint32_t* strand_ncand, uint32_t* worklist, uint32_t* heavy_list, uint64_t* counters);
void launch_evaluate(hipStream_t s, const DevIndexView& ix, const EvalArgs& a, uint64_t max_items, uint32_t max_len);
// reads of kMaxRegisterReadLen + 1 .. kMaxReadLen bases: the same sweep in bands of 256 rows (a.strip / a.strip_len set:
// tiled_groups(max_items, strip_len) strips of strip_len uint2 each, strip_len >= the pass's longest window)
void launch_evaluate_tiled(hipStream_t s, const DevIndexView& ix, const EvalArgs& a, uint64_t max_items);
uint32_t tiled_groups(uint64_t max_items, uint32_t strip_len);
void launch_max_window(hipStream_t s, uint32_t n_strands, const uint32_t* strand_off, const uint32_t* strand_ncand,
                       const uint4* cand, unsigned long long* out);
// Myers bit-vector edit distance, lane per candidate (reads up to 253 bases).
// mode 0: edit-first order over the same-TaxId chains; 1: the candidates of a.worklist that passed the SW prefilter
// (reference order); 2: the recurrence as a two-sided bound on the prefilter's predicate itself -- what it proves to
// pass goes to a.pass_list, what it refutes is marked failed (and the TaxId's next candidate bounded), the rest goes
// to a.und_list (count in the low word of counter slot a.und_slot), flagged, for launch_sw_pairs; a.counters set
void launch_edit_myers(hipStream_t s, const DevIndexView& ix, const EvalArgs& a, uint64_t max_items, uint32_t max_len,
                       int mode = 0);
// reference order for reads <= 253 bases: SW prefilter alone, two candidates per 16-lane group
// diag = false: without the lower bounds on the seed diagonal (every candidate that is not hopeless is swept)
// top = true: the sweep on the top half of the read rows (k_sw_pairs<R/2, false, TOP>): refutes or passes what those rows
// decide, appends the rest to a.und_list (count in the low word of counter slot a.und_slot), flagged, for a second launch
// sparse = true: the work list holds a small fraction of max_items (what the edit-distance bound left): a small grid
void launch_sw_pairs(hipStream_t s, const DevIndexView& ix, const EvalArgs& a, uint64_t max_items, uint32_t max_len,
                     bool diag = true, bool top = false, bool sparse = false);
// the prefilter's lower bounds alone, a lane per work item: decided candidates go to a.pass_list (count in kCtrPassCount),
// the others to sweep_list (count in the low word of counter slot sweep_slot), flagged, for launch_sw_pairs
void launch_sw_diag(hipStream_t s, const DevIndexView& ix, const EvalArgs& a, uint64_t max_items, uint32_t max_len,
                    uint32_t* sweep_list, uint32_t sweep_slot);
void launch_resolve(hipStream_t s, uint32_t n_strands, int64_t max_candidates, int64_t max_assignments,
                    const uint32_t* strand_off, const uint32_t* strand_ncand, const uint32_t* cand_status, uint4* out,
                    uint32_t* strand_nout);
void launch_gather(hipStream_t s, uint32_t n_strands, uint64_t r0, const uint32_t* strand_off, const uint32_t* strand_nout,
                   const uint32_t* out_off, const uint4* out, DevHit* hits, uint64_t hits_base);

}  // namespace mtsv

// batch.hpp -- device workspace of one read batch (see batch.hip).
#pragma once
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/mtsv_amd.h"
#include "host_pack.hpp"
#include "dev_index.hpp"
#include "kernels.hpp"

namespace mtsv {

struct Batch {
    mtsv_index* ix;
    DeviceIndex* di;
    hipStream_t stream = nullptr, copy_stream2 = nullptr;
    mtsv_hit* h_hits_stage = nullptr;  // pinned (pool): run_host copies every slice's hits here while later slices run
    uint64_t h_hits_cap = 0, staged_hits = 0, last_total_hits = 0;
    bool staged_valid = false;
    // run_host: the input arenas in HBM (the whole host batch, or segments of it taking turns), filled in read order by one
    // feeder thread on copy_stream; page-locked staging chunks for bases that lie in ordinary memory
    struct Arena {
        uint8_t* d_bases = nullptr;
        uint8_t* d_packed = nullptr;  // the same bases as they arrive: 4-bit codes, two per byte (host_pack.hpp)
        uint32_t* d_off = nullptr;
        uint64_t cap_bases = 0, cap_reads = 0;
    };
    static constexpr uint64_t kArenaBases = 3ull << 30;   // offsets inside a segment are u32
    static constexpr uint64_t kArenaReads = 1ull << 30;
    Arena arena[2];
    uint32_t* h_off_all = nullptr;  // pinned: the batch's offsets, narrowed to segment-relative u32
    uint64_t h_off_cap = 0;
    static constexpr int kStage = 4;
    uint8_t* h_stage[kStage] = {};
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> chunk_ev;  // one per copy chunk, kept between runs
    std::mutex* commit_mu = nullptr;  // set while run_host's threads may touch the lanes' result arrays
    uint64_t max_reads, max_bases, hit_cap, hits_cap = 0;
    uint64_t bytes = 0;
    // Lanes: the workspace is cut into n_lanes equal parts, each with its own stream; a resident range of
    // reads is split across them and the parts run concurrently (one host thread each), so that the
    // HBM-latency-bound stages of one part overlap the VALU-bound verification of another.  The owner
    // is lane 0 and holds the input buffers; the other lanes borrow them.  MTSV_LANES=1 disables it.
    Batch* parent = nullptr;
    std::vector<std::unique_ptr<Batch>> extra;
    int n_lanes = 1;
    uint64_t ws_reads = 0;  // reads one lane's workspace is sized for
    struct Segment {
        Batch* lane;
        uint64_t offset, count;
    };
    std::vector<Segment> segments;  // where the hits of the last run sit, in read order
    uint64_t total_hits = 0;
    double run_t0 = 0;
    uint64_t lanes_used = 1;

    uint8_t* d_codes = nullptr;  // run(): normalised copy of d_bases
    uint8_t* d_bases = nullptr;  // resident batch of upload() / run()
    uint32_t* d_read_off = nullptr;
    uint32_t *d_seed_lo = nullptr, *d_seed_cnt = nullptr, *d_seed_pre = nullptr;
    uint64_t seed_cap = 0;
    uint32_t *d_strand_hits = nullptr, *d_strand_nseeds = nullptr, *d_strand_off = nullptr, *d_strand_ncand = nullptr,
             *d_worklist = nullptr, *d_strand_nout = nullptr, *d_out_off = nullptr;
    uint64_t* d_tile_sums = nullptr;
    uint64_t* d_counters = nullptr;
    uint32_t *d_hit_row = nullptr, *d_hit_ref = nullptr, *d_hit_q = nullptr;
    uint64_t *d_hit_key = nullptr, *d_cand_tmp = nullptr;
    uint4 *d_cand = nullptr, *d_out = nullptr;
    uint32_t *d_cand_next = nullptr, *d_cand_status = nullptr, *d_heavy_list = nullptr;
    DevHit* d_hits = nullptr;
    uint2* d_strip = nullptr;  // tiled long-read kernel: band hand-over strips, allocated when a pass first needs them
    uint64_t strip_cap = 0;
    uint64_t* h_counters = nullptr;  // pinned
    hipEvent_t ev[15];  // [0..7] stage boundaries of a pass, [8..9] the lane's run, [10..11] around the prefilter kernels, [12] after k_sw_diag,
                        // [13] after the edit-distance bound, [14] after the first round's edit distances
    float sw_ms_acc = 0, sweep_ms_acc = 0, diag_ms_acc = 0, bound_ms_acc = 0, edit_ms_acc = 0;
    uint64_t sw_passed_acc = 0;   // candidates k_sw_pairs sent on to the edit distance (all rounds and passes of the run)

    std::vector<uint32_t> h_read_off;
    uint32_t max_len = 0;
    static constexpr int kCounters = 24;
    double listed_share = 0.125;  // seed slots that took the general search code in the last pass, with a margin (k_search_listed's grid)
    bool sw_diag = true;   // k_sw_pairs tries the ungapped diagonal as a lower bound first (MTSV_SW_DIAG=0: off)
    bool sw_top = true;      // ... and what they leave is swept on the top half of the read rows first (MTSV_SW_TOP=0: off)
    bool sw_bound = true;    // ... first by the edit-distance bound on the score (k_edit_myers in bound mode; MTSV_SW_BOUND=0: off)
    bool sw_prepass = true;  // the first round's bounds run as a kernel of their own, k_sw_diag (MTSV_SW_PREPASS=0: inside k_sw_pairs)
    bool sw_pairs = true;  // reference order for reads <= 253 bases: k_sw_pairs + k_edit_myers (MTSV_SW=packed: k_evaluate)
    int verify_mode = 0;  // 0 = reference order (SW + edit per candidate), 1 = edit first (MTSV_VERIFY_EDIT_FIRST)
    uint64_t n_reads = 0;
    uint64_t n_hits_total = 0;
    mtsv_batch_stats stats{};
    float stage_acc[MTSV_N_STAGES] = {0};

    // lanes: ranges of a host batch that run through the kernels at once, each on a stream and workspace of its own (0: three,
    // MTSV_LANES)
    Batch(mtsv_index* ix, DeviceIndex* di, uint64_t max_reads, uint64_t max_bases, uint64_t hit_cap, Batch* parent = nullptr, int lanes = 0);
    ~Batch();
    Batch(const Batch&) = delete;
    Batch& operator=(const Batch&) = delete;

    void upload(const uint8_t* bases, const uint64_t* read_off, uint64_t n);
    void run(const mtsv_params& p);
    // read_base: added to the `read` field of every hit (a caller that shards one host batch over devices)
    void run_host(const uint8_t* bases, const uint64_t* read_off, uint64_t n, const mtsv_params& p, uint64_t read_base = 0);
    struct HostPart {
        const uint8_t* bases;
        const uint64_t* read_off;  // n + 1 offsets into bases
        uint64_t n;
    };
    void run_host_parts(const HostPart* parts, int n_parts, const mtsv_params& p, uint64_t read_base = 0);
    // everything run_host sizes by the batch it is given (device arenas, offset table, result array), for batches of up to
    // n reads / n_bases bases, now
    void reserve_host(uint64_t n, uint64_t n_bases);
    void download(mtsv_hit** hits, uint64_t* n);
    // the hits of the last run, still in HBM (keep_on_device was set), into caller memory that holds n == total_hits entries
    void download_into(mtsv_hit* dst, uint64_t n);
    bool keep_on_device = false;  // run_host: leave the hits in the lanes' result arrays (no copy to the host while it runs)

   private:
    void begin_run(const mtsv_params& p);
    void reset_lane();
    void stage_reserve(uint64_t n_hits_needed);
    void host_room(uint64_t n, uint64_t total_bases, bool trace);
    void finish_lane();
    void alloc_hit_workspace();
    void grow_hit_workspace(uint64_t need);
    void run_range(const mtsv_params& p, const uint8_t* raw, uint8_t* sb, const uint32_t* so, const uint32_t* h_off, uint64_t n,
                   uint32_t range_max_len, uint64_t read_base);
    void run_slice(const mtsv_params& p, const uint8_t* sb, const uint32_t* so, const uint32_t* h_off, uint64_t n_slice,
                   uint32_t slice_max_len, uint64_t read_base);
    void end_run();
};

extern int g_default_verify_mode;  // mtsv_set_default_verify_mode

// result arrays in pinned host memory, recycled through a pool (mtsv_hits_free returns them)
mtsv_hit* pinned_hits_alloc(uint64_t n_hits, uint64_t* cap_hits);
bool pinned_hits_release(void* p);  // false: p is not a pool array
// page-locked host memory for the callers' read buffers (mtsv_host_alloc & co.)
void* host_pinned_alloc(uint64_t bytes);
void host_pinned_free(void* p);
bool host_pinned_register(void* p, uint64_t bytes);
bool host_pinned_unregister(void* p);
bool host_pinned(const void* p, uint64_t bytes);

}  // namespace mtsv

struct mtsv_batch {
    mtsv::Batch impl;
    template <class... A>
    explicit mtsv_batch(A&&... a) : impl(std::forward<A>(a)...) {}
};

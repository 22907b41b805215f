// batch.hip -- device workspace for one batch of reads and the stage-by-stage driver of the hot
// path.  Replaces the producer -> workers -> joiner queue of vendor/cue/src/lib.rs:45-105 with
// large HBM-resident batches: reads stay on the device between stages, every stage is one launch
// over the whole batch, and the only host round-trips are two 8-byte totals per pass.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/mtsv_amd.h"
#include "batch.hpp"
#include "kernels.hpp"

namespace mtsv {

namespace {
// memcpy / first-touch split over a few threads for large host buffers
template <class F>
void parallel_ranges(uint64_t n, F f) {
    static const int max_nt = [] {
        const char* e = getenv("MTSV_COPY_THREADS");
        return e ? std::max(1, std::min(16, atoi(e))) : 4;
    }();
    const int nt = std::min(max_nt, n >= (32ull << 20) ? 4 : n >= (4ull << 20) ? 2 : 1);
    if (nt == 1) return f(0, n);
    std::vector<std::thread> th;
    const uint64_t chunk = ((n + nt - 1) / nt + 4095) & ~4095ull;
    for (int k = 0; k < nt; k++) {
        uint64_t a = std::min(n, chunk * k), b = std::min(n, chunk * (k + 1));
        if (a < b) th.emplace_back([=] { f(a, b); });
    }
    for (auto& t : th) t.join();
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void parallel_copy(uint8_t* dst, const uint8_t* src, uint64_t n) {
    parallel_ranges(n, [=](uint64_t a, uint64_t b) { memcpy(dst + a, src + a, b - a); });
}

template <class T>
void dev_alloc(T** p, uint64_t count, uint64_t* bytes) {
    uint64_t b = std::max<uint64_t>(count, 1) * sizeof(T);
    HIP_CHECK(hipMalloc((void**)p, b));
    *bytes += b;
}
}  // namespace

// true when [p, p + bytes) is page-locked host memory the runtime knows (hipHostMalloc / hipHostRegister)
bool host_pinned(const void* p, uint64_t bytes) {
    if (!p || !bytes) return false;
    for (const uint8_t* q : {(const uint8_t*)p, (const uint8_t*)p + bytes - 1}) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();  // ordinary memory: not an error of ours
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

void* host_pinned_alloc(uint64_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void host_pinned_free(void* p) {
    if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}
bool host_pinned_register(void* p, uint64_t bytes) {
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}
bool host_pinned_unregister(void* p) {
    if (hipHostUnregister(p) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}


int g_default_verify_mode = MTSV_VERIFY_REFERENCE;

constexpr uint64_t kLaneMinReads = 32768;  // a lane below this many reads does not fill the device
constexpr uint64_t kChunkMaxReads = 4ull << 20;  // lanes take a range in chunks of at most this many reads (smaller chunks measured slower: per-pass launches and host round trips)

Batch::Batch(mtsv_index* ix_, DeviceIndex* di_, uint64_t max_reads_, uint64_t max_bases_, uint64_t hit_cap_, Batch* parent_, int lanes_)
    : ix(ix_), di(di_), max_reads(max_reads_), max_bases(max_bases_), hit_cap(hit_cap_), parent(parent_) {
    if (max_reads == 0) max_reads = 1;
    if (max_reads > 0x7fffffffull) throw std::runtime_error("limit: more than 2^31 reads in one batch");
    if (max_bases >= 0xffffffffull) throw std::runtime_error("limit: 4 GiB of bases or more in one batch");
    const uint64_t hit_cap_user = hit_cap;
    if (!parent) {
        n_lanes = 3;
        if (lanes_ > 0) n_lanes = std::min(8, lanes_);
        else if (const char* e = getenv("MTSV_LANES")) n_lanes = std::max(1, std::min(8, atoi(e)));
        if (max_reads < (uint64_t)n_lanes * kLaneMinReads) n_lanes = 1;
    }
    ws_reads = (max_reads + n_lanes - 1) / n_lanes;
    if (!parent && n_lanes > 1) {
        uint64_t cmax = kChunkMaxReads;
        if (const char* e = getenv("MTSV_CHUNK")) cmax = std::max<uint64_t>(kLaneMinReads, strtoull(e, nullptr, 10));
        ws_reads = std::min(ws_reads, cmax);
    }
    if (hit_cap == 0) hit_cap = std::max<uint64_t>(1ull << 20, 32 * ws_reads);
    if (hit_cap > 0xfffffff0ull) hit_cap = 0xfffffff0ull;
    hits_cap = std::max<uint64_t>(1ull << 20, 16 * ws_reads);
    if (const char* e = getenv("MTSV_HITS_CAP")) hits_cap = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));  // (tests: initial size)
    HIP_CHECK(hipSetDevice(di->device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    const uint64_t ns = 2 * ws_reads;
    if (!parent) {
        dev_alloc(&d_bases, max_bases + 64, &bytes);  // k_search reads up to 36 bytes past a seed start with dword loads
        dev_alloc(&d_read_off, max_reads + 1, &bytes);
        dev_alloc(&d_codes, max_bases + 64, &bytes);  // normalised copy of d_bases, rewritten by every run()
    }
    dev_alloc(&d_strand_hits, ns + 1, &bytes);
    dev_alloc(&d_strand_nseeds, ns, &bytes);
    dev_alloc(&d_strand_off, ns + 1, &bytes);
    dev_alloc(&d_strand_ncand, ns, &bytes);
    dev_alloc(&d_strand_nout, ns + 1, &bytes);
    dev_alloc(&d_heavy_list, 2 * ns, &bytes);  // [0, ns): the two strand lists of the coalescing kernels (front / back), [ns, 2 ns): the 13..16-hit strands
    dev_alloc(&d_out_off, ns + 1, &bytes);
    dev_alloc(&d_tile_sums, (uint64_t)scan_tiles((uint32_t)ns) + 1, &bytes);
    dev_alloc(&d_counters, kCounters, &bytes);
    alloc_hit_workspace();
    dev_alloc(&d_hits, hits_cap, &bytes);
    // page-locked and mapped: kernels store the counters the host waits for straight into it (launch_publish) -- a
    // hipMemcpyAsync of 8 bytes can queue on a copy engine behind tens of milliseconds of the reads' own transfer
    HIP_CHECK(hipHostMalloc((void**)&h_counters, kCounters * sizeof(uint64_t), hipHostMallocMapped));
    if (const char* e = getenv("MTSV_SW")) sw_pairs = strcmp(e, "packed") != 0;
    if (const char* e = getenv("MTSV_SW_DIAG")) sw_diag = atoi(e) != 0;
    if (const char* e = getenv("MTSV_SW_PREPASS")) sw_prepass = atoi(e) != 0;
    if (const char* e = getenv("MTSV_SW_TOP")) sw_top = atoi(e) != 0;
    if (const char* e = getenv("MTSV_SW_BOUND")) sw_bound = atoi(e) != 0;
    verify_mode = g_default_verify_mode;
    if (const char* e = getenv("MTSV_VERIFY")) verify_mode = !strcmp(e, "edit_first") ? 1 : 0;
    for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
    for (int k = 1; k < n_lanes; k++) {
        extra.emplace_back(new Batch(ix, di, ws_reads, 0, hit_cap_user, this));
        bytes += extra.back()->bytes;
    }
}

Batch::~Batch() {
    extra.clear();
    (void)hipSetDevice(di->device);
    (void)hipStreamSynchronize(stream);
    for (void* p : {(void*)d_bases, (void*)d_read_off, (void*)d_seed_lo, (void*)d_seed_cnt, (void*)d_seed_pre, (void*)d_strand_hits,
                    (void*)d_strand_nseeds, (void*)d_strand_off, (void*)d_strand_ncand, (void*)d_worklist,
                    (void*)d_strand_nout, (void*)d_out_off, (void*)d_tile_sums, (void*)d_counters, (void*)d_hit_row,
                    (void*)d_hit_ref, (void*)d_hit_q, (void*)d_hit_key, (void*)d_cand_tmp, (void*)d_cand, (void*)d_out,
                    (void*)d_hits, (void*)d_cand_next, (void*)d_cand_status, (void*)d_heavy_list})
        (void)hipFree(p);
    (void)hipFree(d_codes);
    (void)hipFree(d_strip);
    if (h_hits_stage) pinned_hits_release(h_hits_stage);
    if (copy_stream2) (void)hipStreamDestroy(copy_stream2);
    if (copy_stream) {
        (void)hipStreamSynchronize(copy_stream);
        (void)hipStreamDestroy(copy_stream);
    }
    for (auto& ar : arena) {
        (void)hipFree(ar.d_bases);
        (void)hipFree(ar.d_packed);
        (void)hipFree(ar.d_off);
    }
    if (h_off_all) (void)hipHostFree(h_off_all);
    for (auto& hs : h_stage)
        if (hs) (void)hipHostFree(hs);
    for (auto& e : chunk_ev) (void)hipEventDestroy(e);
    (void)hipHostFree(h_counters);
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(stream);
}

// the arrays indexed by seed hit / candidate; nothing of a pass lives in them before its locate stage
void Batch::alloc_hit_workspace() {
    dev_alloc(&d_hit_row, hit_cap, &bytes);
    dev_alloc(&d_hit_ref, hit_cap, &bytes);
    dev_alloc(&d_hit_q, hit_cap, &bytes);
    dev_alloc(&d_hit_key, hit_cap, &bytes);
    dev_alloc(&d_cand_tmp, 2 * hit_cap, &bytes);
    dev_alloc(&d_cand, hit_cap, &bytes);
    dev_alloc(&d_out, hit_cap, &bytes);
    dev_alloc(&d_cand_next, hit_cap, &bytes);
    dev_alloc(&d_cand_status, hit_cap, &bytes);
    dev_alloc(&d_worklist, hit_cap, &bytes);
}

void Batch::grow_hit_workspace(uint64_t need) {
    if (getenv("MTSV_TRACE")) fprintf(stderr, "[workspace] seed-hit arrays grow %llu -> %llu entries\n", (unsigned long long)hit_cap, (unsigned long long)need);
    HIP_CHECK(hipStreamSynchronize(stream));
    for (void* p : {(void*)d_hit_row, (void*)d_hit_ref, (void*)d_hit_q, (void*)d_hit_key, (void*)d_cand_tmp, (void*)d_cand,
                    (void*)d_out, (void*)d_cand_next, (void*)d_cand_status, (void*)d_worklist})
        (void)hipFree(p);
    d_hit_row = d_hit_ref = d_hit_q = d_cand_next = d_cand_status = d_worklist = nullptr;
    d_hit_key = d_cand_tmp = nullptr;
    d_cand = d_out = nullptr;
    hit_cap = need;
    alloc_hit_workspace();
}

void Batch::upload(const uint8_t* bases, const uint64_t* read_off, uint64_t n) {
    if (n > max_reads) throw std::runtime_error("arg: batch holds more reads than the workspace was created for");
    const uint64_t first = n ? read_off[0] : 0;
    const uint64_t nb = n ? read_off[n] - first : 0;
    if (nb > max_bases) throw std::runtime_error("arg: batch holds more bases than the workspace was created for");
    h_read_off.resize(n + 1);
    max_len = 0;
    for (uint64_t i = 0; i <= n; i++) {
        if (i && read_off[i] < read_off[i - 1]) throw std::runtime_error("arg: read_off is not ascending");
        h_read_off[i] = (uint32_t)(read_off[i] - first);
        if (i) max_len = std::max(max_len, h_read_off[i] - h_read_off[i - 1]);
    }
    if (n == 0) h_read_off[0] = 0;
    HIP_CHECK(hipSetDevice(di->device));
    if (nb) HIP_CHECK(hipMemcpyAsync(d_bases, bases + first, nb, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_read_off, h_read_off.data(), (n + 1) * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    n_reads = n;
    n_hits_total = 0;
    total_hits = 0;
    segments.clear();
}

// counters in d_counters: [0] scan total (u64), [1] wl_count (u32), [2] lf_steps, [3] n_cand,
// [4] n_verified, [5] window_bytes, [6] second scan total
void Batch::reset_lane() {
    HIP_CHECK(hipSetDevice(di->device));
    memset(&stats, 0, sizeof stats);
    memset(stage_acc, 0, sizeof stage_acc);
    sw_ms_acc = 0;
    sweep_ms_acc = diag_ms_acc = bound_ms_acc = edit_ms_acc = 0;
    sw_passed_acc = 0;
    n_hits_total = 0;
    HIP_CHECK(hipMemsetAsync(d_counters, 0, kCounters * sizeof(uint64_t), stream));
    HIP_CHECK(hipEventRecord(ev[8], stream));
}

void Batch::finish_lane() {
    HIP_CHECK(hipEventRecord(ev[9], stream));
    launch_publish(stream, d_counters, h_counters, kCounters);
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipEventElapsedTime(&stage_acc[7], ev[8], ev[9]));
    if (getenv("MTSV_TRACE")) fprintf(stderr, "[lane] last pass: %llu strands of 13..64 seed hits (or more than 4 candidates) to k_coalesce_mid, %llu heavier ones\n",
                                      (unsigned long long)(h_counters[15] & 0xffffffffull), (unsigned long long)(h_counters[1] >> 32));
    stats.sw_cell_pairs = h_counters[14];
    stats.sw_prefilter_ms = sw_ms_acc;
    stats.sw_sweep_ms = sweep_ms_acc;
    stats.sw_diag_ms = diag_ms_acc;
    stats.sw_bound_ms = bound_ms_acc;
    stats.edit_ms = edit_ms_acc;
    stats.myers_columns = h_counters[19];
    stats.n_sw_bound_refuted = h_counters[20];
    stats.n_sw_passed = sw_passed_acc + h_counters[21];  // (+ the successors k_edit_myers' list mode passed by its own bound)
    stats.lf_steps = h_counters[2];
    stats.n_candidates = h_counters[3];
    stats.n_verified = h_counters[4];
    stats.window_bytes = h_counters[5];
}

void Batch::begin_run(const mtsv_params& p) {
    if (!(p.edit_rate >= 0.0 && p.edit_rate <= 1.0)) throw std::runtime_error("arg: edit_rate must be within [0, 1]");
    if (!(p.min_seed >= 0.0) || !std::isfinite(p.min_seed)) throw std::runtime_error("arg: min_seed must be finite and >= 0");
    if (p.seed_size == 0 || p.seed_interval == 0) throw std::runtime_error("arg: seed_size and seed_interval must be > 0");
    segments.clear();
    total_hits = 0;
    lanes_used = 1;
    staged_valid = false;
    run_t0 = now_s();
    reset_lane();
    for (auto& l : extra) {
        l->verify_mode = verify_mode;
        l->sw_pairs = sw_pairs;
        l->sw_diag = sw_diag;
        l->sw_prepass = sw_prepass;
        l->sw_top = sw_top;
        l->sw_bound = sw_bound;
        l->reset_lane();
    }
}

void Batch::end_run() {
    finish_lane();
    for (auto& l : extra) l->finish_lane();
    const double wall_ms = (now_s() - run_t0) * 1e3;
    // stage times are summed over the lanes (device time per stage); the total is the wall time of the
    // run when lanes overlapped, the stream's own event span otherwise
    bool overlapped = false;
    for (auto& l : extra) {
        if (l->stats.n_passes == 0) continue;
        overlapped = true;
        for (int s = 0; s < 7; s++) stage_acc[s] += l->stage_acc[s];
        stats.n_seed_slots += l->stats.n_seed_slots;
        stats.n_seed_hits += l->stats.n_seed_hits;
        stats.n_passes += l->stats.n_passes;
        stats.n_rounds = std::max(stats.n_rounds, l->stats.n_rounds);
        stats.lf_steps += l->stats.lf_steps;
        stats.n_candidates += l->stats.n_candidates;
        stats.n_verified += l->stats.n_verified;
        stats.window_bytes += l->stats.window_bytes;
        stats.sw_cell_pairs += l->stats.sw_cell_pairs;
        stats.sw_prefilter_ms += l->stats.sw_prefilter_ms;
        stats.sw_sweep_ms += l->stats.sw_sweep_ms;
        stats.sw_diag_ms += l->stats.sw_diag_ms;
        stats.sw_bound_ms += l->stats.sw_bound_ms;
        stats.edit_ms += l->stats.edit_ms;
        stats.myers_columns += l->stats.myers_columns;
        stats.n_sw_bound_refuted += l->stats.n_sw_bound_refuted;
        stats.n_sw_passed += l->stats.n_sw_passed;
    }
    if (overlapped) stage_acc[7] = (float)wall_ms;
    for (int s = 0; s < MTSV_N_STAGES; s++) stats.stage_ms[s] = stage_acc[s];
    stats.n_reads = n_reads;
    stats.n_lanes = lanes_used;
    total_hits = 0;
    for (auto& sg : segments) total_hits += sg.count;
    stats.n_hits = total_hits;
}

void Batch::run(const mtsv_params& p) {
    begin_run(p);
    run_range(p, d_bases, d_codes, d_read_off, h_read_off.data(), n_reads, max_len, 0);
    end_run();
}

// A resident range of reads, split over the lanes; the hits stay in the lanes, `segments` records them in
// read order.  raw != nullptr: the range is still ASCII in `raw` and each chunk normalises its bytes into
// `sb` first (h_off = host copy of the range's n + 1 offsets); raw == nullptr: `sb` already holds codes.
void Batch::run_range(const mtsv_params& p, const uint8_t* raw, uint8_t* sb, const uint32_t* so, const uint32_t* h_off, uint64_t n,
                      uint32_t range_max_len, uint64_t read_base) {
    std::vector<Batch*> ls{this};
    for (auto& l : extra) ls.push_back(l.get());
    // Chunks of at most ws_reads reads, handed to the lanes from a shared counter.  The lanes must not march in
    // step (three index lookups at once, then three prefilters at once overlap nothing): every lane gets several
    // chunks, and the first chunk of lane i is (i + 1) / lanes of a full one, so the lanes stay a fraction of a
    // chunk apart and one lane's index lookups (gather-bound) run under another lane's prefilter (VALU-bound).
    uint64_t k = n >= ls.size() * kLaneMinReads ? ls.size() : 1;
    uint64_t per_lane = 2;  // measured on config2: 1: 52.2 ms, 2: 51.9, 3: 53.0, 4: 54.7, 6: 57.0 per 10 M reads
    if (const char* e = getenv("MTSV_CHUNKS_PER_LANE")) per_lane = std::max(1, atoi(e));
    uint64_t n_chunks = std::max<uint64_t>(k, (n + ws_reads - 1) / ws_reads);
    if (k > 1 && n >= k * per_lane * kLaneMinReads) n_chunks = std::max(n_chunks, k * per_lane);
    std::vector<uint64_t> bound(n_chunks + 1, n);
    bound[0] = 0;
    // (the short first chunks make the full ones longer: more chunks until a full one fits the workspace)
    while (k > 1 && n_chunks >= 2 * k && (double)n / ((double)n_chunks - (double)(k - 1) / 2.0) > (double)ws_reads) n_chunks++;
    bound.assign(n_chunks + 1, n);
    bound[0] = 0;
    if (k > 1 && n_chunks >= 2 * k) {
        // sizes: per * 1/k, per * 2/k, .., per, then equal chunks of the rest
        const double per0 = (double)n / ((double)n_chunks - (double)(k - 1) / 2.0);  // full chunk size with the short ones counted
        uint64_t at = 0;
        for (uint64_t c = 0; c < k; c++) {
            at += (uint64_t)(per0 * (double)(c + 1) / (double)k);
            bound[c + 1] = std::min(at, n);
        }
        const uint64_t rest = n - bound[k], m = n_chunks - k;
        for (uint64_t c = 0; c < m; c++) bound[k + c + 1] = bound[k] + rest * (c + 1) / m;
    } else {
        for (uint64_t c = 0; c < n_chunks; c++) bound[c + 1] = n * (c + 1) / n_chunks;
    }
    for (uint64_t c = 0; c < n_chunks; c++)
        if (bound[c + 1] - bound[c] > ws_reads) throw std::runtime_error("arg: range holds more reads than the workspace was created for");
    k = std::min(k, n_chunks);
    lanes_used = std::max<uint64_t>(lanes_used, k);
    std::vector<Segment> segs(n_chunks);
    std::atomic<uint64_t> next{0};
    auto work = [&](Batch* lane) {
        for (;;) {
            const uint64_t c = next.fetch_add(1);
            if (c >= n_chunks) return;
            const uint64_t a = bound[c], b = bound[c + 1];
            const uint64_t before = lane->n_hits_total;
            // base normalisation (binner.rs:88-100) of this chunk's bytes: raw -> codes, on the lane's stream
            if (raw) launch_normalise(lane->stream, raw, sb, h_off[a], h_off[b]);
            lane->run_slice(p, sb, so + a, h_off ? h_off + a : nullptr, b - a, range_max_len, read_base + a);
            segs[c] = Segment{lane, before, lane->n_hits_total - before};
        }
    };
    if (k == 1) {
        work(this);
    } else {
        std::vector<std::exception_ptr> errs(k);
        std::vector<std::thread> th;
        for (uint64_t i = 1; i < k; i++)
            th.emplace_back([&, i] {
                try {
                    HIP_CHECK(hipSetDevice(di->device));
                    work(ls[i]);
                } catch (...) {
                    errs[i] = std::current_exception();
                    next.store(n_chunks);  // stop the others
                }
            });
        try {
            work(this);
        } catch (...) {
            errs[0] = std::current_exception();
            next.store(n_chunks);
        }
        for (auto& t : th) t.join();
        for (auto& e : errs)
            if (e) std::rethrow_exception(e);
    }
    for (auto& sg : segs) segments.push_back(sg);
}

// One slice of reads already in HBM (bases `sb`, offsets `so`, `n_slice` reads whose first read is
// number `read_base` of the caller's batch): passes over the slice, hits appended to d_hits.
// h_off: host copy of the slice's n_slice + 1 offsets (needed only when the slice holds reads beyond
// kMaxRegisterReadLen bases).
void Batch::run_slice(const mtsv_params& p, const uint8_t* sb, const uint32_t* so, const uint32_t* h_off, uint64_t n_slice,
                      uint32_t slice_max_len, uint64_t read_base) {
    if (slice_max_len > kMaxReadLen)
        throw std::runtime_error("limit: read of " + std::to_string(slice_max_len) + " bases; this build verifies reads up to " +
                                 std::to_string(kMaxReadLen));
    const DevIndexView& v = di->view;
    const uint32_t K = p.seed_size, G = p.seed_interval;
    float* stage_ms = stage_acc;
    // Long reads (beyond the register-resident kernels) run in passes of their own, through the tiled kernel:
    // a pass is a contiguous range of reads, so cutting the slice at them keeps the hits in read order, and
    // the dense per-strand seed slots of the other passes stay sized for ordinary reads.
    const bool mixed = slice_max_len > kMaxRegisterReadLen;
    if (mixed && !h_off) throw std::runtime_error("internal: a slice with long reads needs its host offsets");
    constexpr uint64_t kLongPassSlots = 16ull << 20;  // seed slots of one long-read pass, at most

    uint64_t pass_reads = n_slice;
    uint64_t r0 = 0;
    while (r0 < n_slice) {
        uint32_t nr = (uint32_t)std::min<uint64_t>(pass_reads, n_slice - r0);
        uint32_t pass_max_len = slice_max_len;
        bool tiled = false;
        if (mixed) {
            auto len_of = [&](uint64_t i) { return h_off[i + 1] - h_off[i]; };
            tiled = len_of(r0) > kMaxRegisterReadLen;
            uint32_t cnt = 0, ml = 0;
            while (cnt < nr && (len_of(r0 + cnt) > kMaxRegisterReadLen) == tiled) {
                const uint32_t ml2 = std::max(ml, len_of(r0 + cnt));
                if (tiled && cnt && 2ull * (cnt + 1) * (ml2 >= K ? (ml2 - K) / G + 1 : 0) > kLongPassSlots) break;
                ml = ml2;
                cnt++;
            }
            nr = cnt;
            pass_max_len = ml;
        }
        const uint32_t max_ns = pass_max_len >= K ? (pass_max_len - K) / G + 1 : 0;
        const uint32_t nstr = nr * 2;
        const uint64_t slots = (uint64_t)nstr * max_ns;
        if (slots > seed_cap) {
            (void)hipFree(d_seed_lo);
            (void)hipFree(d_seed_cnt);
            (void)hipFree(d_seed_pre);
            d_seed_lo = d_seed_cnt = d_seed_pre = nullptr;
            // room for a full workspace of reads like these, not just this pass: slices of a host batch grow
            // towards the workspace size, and every hipFree / hipMalloc stalls the whole device
            const uint64_t want = tiled ? slots : std::max<uint64_t>(slots, 2 * ws_reads * (uint64_t)max_ns);
            uint64_t dummy = 0;
            if (getenv("MTSV_TRACE")) fprintf(stderr, "[workspace] seed slot arrays grow %llu -> %llu entries\n", (unsigned long long)seed_cap, (unsigned long long)want);
            dev_alloc(&d_seed_lo, want, &dummy);
            dev_alloc(&d_seed_cnt, want, &dummy);
            dev_alloc(&d_seed_pre, want, &dummy);
            seed_cap = want;
        }
        // ---- seeds ----
        HIP_CHECK(hipEventRecord(ev[0], stream));
        // (d_seed_pre is k_thin's output: until then it holds the list of the slots that take the general code -- an N in
        //  the seed's table part; counter slot 22.  The second kernel's grid covers the share of such slots the passes before
        //  had, half as much again: the count is checked at the pass's first round trip below.)
        const uint32_t listed_cap = (uint32_t)std::min<uint64_t>(slots, std::max<uint64_t>(4096, (uint64_t)((double)slots * listed_share)));
        launch_search(stream, v, sb, so, (uint32_t)r0, nr, max_ns, K, G, d_seed_lo, d_seed_cnt, d_seed_pre, (uint32_t*)(d_counters + 22), listed_cap);
        HIP_CHECK(hipEventRecord(ev[1], stream));
        if (max_ns)
            launch_thin(stream, sb, so, (uint32_t)r0, nr, p.edit_rate, p.min_seed, max_ns, K, G, p.max_hits, p.tune_max_hits, d_seed_cnt,
                        d_seed_pre, d_strand_hits, d_strand_nseeds);
        else {
            HIP_CHECK(hipMemsetAsync(d_strand_hits, 0, (uint64_t)nstr * 4, stream));
            HIP_CHECK(hipMemsetAsync(d_strand_nseeds, 0, (uint64_t)nstr * 4, stream));
        }
        launch_scan(stream, d_strand_hits, nstr, d_tile_sums, d_counters + 0, d_strand_off);
        launch_publish(stream, d_counters, h_counters, 1);
        launch_publish(stream, d_counters + 22, h_counters + 22, 1);
        HIP_CHECK(hipEventRecord(ev[2], stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        {
            const uint64_t n_listed = h_counters[22] & 0xffffffffull;
            const double share = slots ? (double)n_listed / (double)slots : 0.0;
            if (n_listed > listed_cap) {  // more slots on the list than the grid covered: the pass again, with a grid for them
                listed_share = std::min(1.0, share * 1.25 + 0.01);
                if (getenv("MTSV_TRACE")) fprintf(stderr, "[lane] %llu listed seed slots, grid for %u: pass again\n", (unsigned long long)n_listed, listed_cap);
                continue;
            }
            listed_share = std::min(1.0, std::max(0.02, share * 1.5 + 0.005));
        }
        const uint64_t total_hits = h_counters[0];
        if (total_hits > hit_cap) {
            if (nr > 1) {
                pass_reads = std::max<uint64_t>(1, nr / 2);
                continue;  // redo this pass with fewer reads
            }
            // one read with more seed hits than the workspace (a long read in a repeat): grow the workspace
            if (total_hits > 0xfffffff0ull)
                throw std::runtime_error("device: one read has " + std::to_string(total_hits) + " seed hits, more than 2^32");
            grow_hit_workspace(total_hits);
        }
        stats.n_passes++;
        stats.n_seed_slots += slots;
        stats.n_seed_hits += total_hits;
        // ---- locate ----
        launch_expand(stream, v, nstr, max_ns, G, d_seed_lo, d_seed_cnt, d_seed_pre, d_strand_off, d_hit_row, d_hit_ref,
                      d_hit_q);
        HIP_CHECK(hipEventRecord(ev[3], stream));
        if (!v.sa_full)
            launch_locate(stream, v, (uint32_t)total_hits, d_strand_off + nstr, d_hit_row, d_hit_ref,
                          (unsigned long long*)(d_counters + 2));
        HIP_CHECK(hipEventRecord(ev[4], stream));
        // ---- candidates ----
        // counters: [1] lo = round-0 worklist count, [7] lo/hi = ping-pong counts of later rounds
        HIP_CHECK(hipMemsetAsync(d_counters + 1, 0, 8, stream));
        HIP_CHECK(hipMemsetAsync(d_counters + 7, 0, 8, stream));
        HIP_CHECK(hipMemsetAsync(d_counters + 15, 0, 8, stream));  // [15] lo = strands for k_coalesce_mid, hi = strands of 13..16 seed hits
        launch_coalesce(stream, v, so, (uint32_t)r0, nstr, p.max_candidates, d_strand_off,
                        d_strand_nseeds, d_hit_ref, d_hit_q, d_hit_key, d_cand_tmp, d_cand, d_cand_next,
                        d_cand_status, d_strand_ncand, d_worklist, d_heavy_list, d_counters);
        HIP_CHECK(hipEventRecord(ev[5], stream));
        // ---- verify: rounds over the same-TaxId chains ----
        {
            EvalArgs a;
            a.bases = sb;
            a.read_off = so;
            a.r0 = (uint32_t)r0;
            a.edit_rate = p.edit_rate;
            a.max_candidates = p.max_candidates;
            a.strand_off = d_strand_off;
            a.cand = d_cand;
            a.cand_next = d_cand_next;
            a.cand_status = d_cand_status;
            a.out = d_out;
            a.n_verified = (unsigned long long*)(d_counters + 4);
            a.window_bytes = (unsigned long long*)(d_counters + 5);
            a.sw_columns = (unsigned long long*)(d_counters + 14);
            // one launch: a group whose candidate fails walks on to the next candidate of the same TaxId
            a.worklist = d_worklist;
            a.wl_count = (const uint32_t*)(d_counters + 1);
            a.wl_cursor = (uint32_t*)(d_counters + 7);
            stats.n_rounds = 1;
            if (tiled) {
                // strips sized from the longest window of the pass (one small round trip; long reads are rare)
                HIP_CHECK(hipMemsetAsync(d_counters + 13, 0, sizeof(uint64_t), stream));
                launch_max_window(stream, nstr, d_strand_off, d_strand_ncand, d_cand, (unsigned long long*)(d_counters + 13));
                launch_publish(stream, d_counters + 13, h_counters + 13, 1);
                HIP_CHECK(hipStreamSynchronize(stream));
                const uint32_t strip_len = (uint32_t)std::max<uint64_t>(h_counters[13], 1);
                const uint64_t need = (uint64_t)tiled_groups(total_hits, strip_len) * strip_len;
                if (need > strip_cap) {
                    (void)hipFree(d_strip);
                    d_strip = nullptr;
                    uint64_t dummy = 0;
                    dev_alloc(&d_strip, need, &dummy);
                    strip_cap = need;
                }
                a.strip = d_strip;
                a.strip_len = strip_len;
                launch_evaluate_tiled(stream, v, a, total_hits);
            } else if (verify_mode == 1 && pass_max_len <= 253) {
                launch_edit_myers(stream, v, a, total_hits, pass_max_len);
            } else if (sw_pairs && pass_max_len <= 253 && hit_cap < 0x80000000ull) {  // (bit 31 of a work item is a flag)
                // Reference order, split by predicate: k_sw_pairs runs the prefilter of index.rs:406 two
                // candidates per group, k_edit_myers the edit distance of :407-410 on those that passed.
                // A candidate that passes the first and fails the second sends its TaxId's next
                // candidate to another round (rare); rounds end when nothing is left.
                // counters: [8] SW cursor, [9] pass count, [10] Myers cursor, [11]/[12] next-round counts, [16] sweep list
                // Round 0 starts with k_sw_diag: the lower bounds on the seed diagonal, a lane per work item; what
                // they decide goes straight to pass_list, the rest to sweep_list for k_sw_pairs.
                uint32_t* pass_list = (uint32_t*)d_hit_key;                       // coalesce scratch is free now
                uint32_t* sweep_list = (uint32_t*)d_hit_key + hit_cap;            // (8 bytes per seed hit)
                uint32_t* next_lists[2] = {(uint32_t*)d_cand_tmp, (uint32_t*)d_cand_tmp + hit_cap};
                const uint32_t* wl = d_worklist;
                const uint32_t* wl_count = (const uint32_t*)(d_counters + 1);
                uint64_t items = total_hits;
                for (uint32_t round = 0;; round++) {
                    uint64_t* next_slot = d_counters + 11 + (round & 1);
                    HIP_CHECK(hipMemsetAsync(d_counters + 8, 0, 3 * sizeof(uint64_t), stream));
                    HIP_CHECK(hipMemsetAsync(next_slot, 0, sizeof(uint64_t), stream));
                    EvalArgs sw = a;
                    sw.worklist = wl;
                    sw.wl_count = wl_count;
                    sw.wl_cursor = (uint32_t*)(d_counters + 8);
                    sw.wl_reverse = round == 0;
                    sw.counters = d_counters;
                    sw.wl_count_slot = round == 0 ? 1u : 11u + ((round - 1) & 1u);
                    sw.pass_list = pass_list;
                    sw.pass_count = (uint32_t*)(d_counters + 9);
#ifdef MTSV_SW_HIST
                    if (!d_strip) { uint64_t dummy = 0; dev_alloc(&d_strip, 4096, &dummy); strip_cap = 4096; }
                    HIP_CHECK(hipMemsetAsync(d_strip, 0, 768 * 4, stream));
                    sw.strip = d_strip;
#endif
                    if (round == 0) HIP_CHECK(hipEventRecord(ev[10], stream));
                    if (round == 0 && sw_diag && sw_prepass) {
                        HIP_CHECK(hipMemsetAsync(d_counters + 16, 0, sizeof(uint64_t), stream));
                        launch_sw_diag(stream, v, sw, items, pass_max_len, sweep_list, 16);
                        HIP_CHECK(hipEventRecord(ev[12], stream));
                        sw.worklist = sweep_list;
                        sw.wl_count_slot = 16;
                        sw.wl_reverse = 0;  // k_sw_diag read the worklist from its end
                    }
                    if (round == 0 && sw_bound) {
                        // What the lower bounds leave are mostly chance seed hits.  The unit-cost edit distance under the SW
                        // matrix's matches bounds the score from both sides (k_edit_myers in bound mode) at a third of a
                        // sweep's instructions: it refutes those, passes most of the rest, and only what lies between its
                        // two thresholds ([17] counts it; the seed-hit rows of k_expand are free by now) is swept below.
                        HIP_CHECK(hipMemsetAsync(d_counters + 17, 0, 2 * sizeof(uint64_t), stream));  // [18]: its claim cursor
                        EvalArgs bd = sw;
                        bd.wl_count = (const uint32_t*)(d_counters + sw.wl_count_slot);
                        bd.wl_cursor = (uint32_t*)(d_counters + 18);
                        bd.und_list = (uint32_t*)d_hit_row;
                        bd.und_slot = 17;
                        bd.myers_ctr = (unsigned long long*)(d_counters + 19);
                        launch_edit_myers(stream, v, bd, items, pass_max_len, 2);
                        HIP_CHECK(hipEventRecord(ev[13], stream));
                        sw.worklist = bd.und_list;
                        sw.wl_count_slot = 17;
                        sw.wl_reverse = 0;
                    } else if (round == 0 && sw_diag && sw_prepass && sw_top) {
                        // (MTSV_SW_BOUND=0) most of these are refuted on the top half of the read rows; the full-height launch
                        // below takes what is left ([17] counts it)
                        HIP_CHECK(hipMemsetAsync(d_counters + 17, 0, sizeof(uint64_t), stream));
                        sw.und_list = (uint32_t*)d_hit_row;
                        sw.und_slot = 17;
                        launch_sw_pairs(stream, v, sw, items, pass_max_len, false, true);
                        HIP_CHECK(hipMemsetAsync(d_counters + 8, 0, sizeof(uint64_t), stream));  // the claim cursor
                        sw.worklist = sw.und_list;
                        sw.wl_count_slot = 17;
                    }
                    launch_sw_pairs(stream, v, sw, items, pass_max_len, sw_diag, false, round == 0 && sw_bound);
#ifdef MTSV_SW_HIST
                    {
                        static uint32_t hh[768];
                        HIP_CHECK(hipMemcpyAsync(hh, d_strip, sizeof hh, hipMemcpyDeviceToHost, stream));
                        HIP_CHECK(hipStreamSynchronize(stream));
                        fprintf(stderr, "SWHIST round %u steps/4:", round);
                        for (int i = 0; i < 128; i++) if (hh[i]) fprintf(stderr, " %d:%u", i * 4, hh[i]);
                        fprintf(stderr, "\nSWHIST halves sweeping 0/1/2: %u %u %u\nSWHIST fail best:", hh[128], hh[129], hh[130]);
                        for (int i = 0; i < 256; i++) if (hh[256 + i]) fprintf(stderr, " %d:%u", i, hh[256 + i]);
                        fprintf(stderr, "\nSWHIST pass best:");
                        for (int i = 0; i < 256; i++) if (hh[512 + i]) fprintf(stderr, " %d:%u", i, hh[512 + i]);
                        fprintf(stderr, "\n");
                    }
#endif
                    if (round == 0) HIP_CHECK(hipEventRecord(ev[11], stream));
                    EvalArgs my = a;
                    my.worklist = pass_list;
                    my.wl_count = (const uint32_t*)(d_counters + 9);
                    my.wl_cursor = (uint32_t*)(d_counters + 10);
                    my.next_list = next_lists[round & 1];
                    my.next_count = (uint32_t*)next_slot;
                    my.myers_ctr = (unsigned long long*)(d_counters + 19);
                    launch_edit_myers(stream, v, my, items, pass_max_len, 1);
                    if (round == 0) HIP_CHECK(hipEventRecord(ev[14], stream));
                    launch_publish(stream, next_slot, h_counters + 11, 1);
                    launch_publish(stream, d_counters + 9, h_counters + 9, 1);
                    HIP_CHECK(hipStreamSynchronize(stream));
                    sw_passed_acc += h_counters[9] & 0xffffffffull;
                    if (round == 0) {
                        float ms = 0;
                        HIP_CHECK(hipEventElapsedTime(&ms, ev[10], ev[11]));
                        sw_ms_acc += ms;
                        int from = 10;  // the sweeps start after whichever bound kernels ran
                        if (sw_diag && sw_prepass) {
                            HIP_CHECK(hipEventElapsedTime(&ms, ev[10], ev[12]));
                            diag_ms_acc += ms;
                            from = 12;
                        }
                        if (sw_bound) {
                            HIP_CHECK(hipEventElapsedTime(&ms, ev[from], ev[13]));
                            bound_ms_acc += ms;
                            from = 13;
                        }
                        HIP_CHECK(hipEventElapsedTime(&ms, ev[from], ev[11]));
                        sweep_ms_acc += ms;
                        HIP_CHECK(hipEventElapsedTime(&ms, ev[11], ev[14]));
                        edit_ms_acc += ms;
                    }
                    const uint64_t n_next = h_counters[11] & 0xffffffffull;
                    if (n_next == 0) break;
                    stats.n_rounds++;
                    wl = next_lists[round & 1];
                    wl_count = (const uint32_t*)next_slot;
                    items = n_next;
                }
            } else {
                launch_evaluate(stream, v, a, total_hits, pass_max_len);
            }
            launch_resolve(stream, nstr, p.max_candidates, p.max_assignments, d_strand_off, d_strand_ncand, d_cand_status,
                           d_out, d_strand_nout);
        }
        HIP_CHECK(hipEventRecord(ev[6], stream));
        // ---- gather ----
        launch_scan(stream, d_strand_nout, nstr, d_tile_sums, d_counters + 6, d_out_off);
        launch_publish(stream, d_counters, h_counters, 8);
        HIP_CHECK(hipStreamSynchronize(stream));
        const uint64_t total_out = h_counters[6];
        if (n_hits_total + total_out > hits_cap) {
            // grow the result array, keeping what earlier passes produced
            uint64_t ncap = std::max(hits_cap * 2, n_hits_total + total_out);
            DevHit* nh = nullptr;
            uint64_t dummy = 0;
            dev_alloc(&nh, ncap, &dummy);
            if (n_hits_total) HIP_CHECK(hipMemcpyAsync(nh, d_hits, n_hits_total * sizeof(DevHit), hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            // run_host may be copying an earlier slice's hits out of the old array on the owner's copy stream (and
            // another lane's thread may be issuing such a copy right now: the commit mutex covers the pointer swap)
            Batch* root = parent ? parent : this;
            std::unique_lock<std::mutex> commit_lk;
            if (root->commit_mu) commit_lk = std::unique_lock<std::mutex>(*root->commit_mu);
            if (root->copy_stream2) HIP_CHECK(hipStreamSynchronize(root->copy_stream2));
            (void)hipFree(d_hits);
            d_hits = nh;
            hits_cap = ncap;
        }
        launch_gather(stream, nstr, read_base + r0, d_strand_off, d_strand_nout, d_out_off, d_out, d_hits, n_hits_total);
        HIP_CHECK(hipEventRecord(ev[7], stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipGetLastError());
        n_hits_total += total_out;
        for (int s = 0; s < 7; s++) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, ev[s], ev[s + 1]));
            stage_ms[s] += ms;
        }
        r0 += nr;
    }
}

// ---------------------------------------------------------------------------------------------
// Pinned result arrays.  The hits of a run travel to page-locked host memory slice by slice while later
// slices still compute, and that array itself is what the caller receives (mtsv_hits_free hands it back
// to this pool), so no copy into fresh malloc memory -- page faults at ~12 GB/s -- sits behind the kernels.
// ---------------------------------------------------------------------------------------------
namespace {
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<void*, uint64_t>> free_list;  // (pointer, bytes)
    std::vector<std::pair<void*, uint64_t>> live;
    static constexpr size_t kKeep = 6;  // idle arrays kept for reuse
    void* get(uint64_t bytes, uint64_t* cap) {
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = free_list.size();
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].second >= bytes && (best == free_list.size() || free_list[i].second < free_list[best].second)) best = i;
            if (best != free_list.size()) {
                auto e = free_list[best];
                free_list.erase(free_list.begin() + best);
                live.push_back(e);
                *cap = e.second;
                return e.first;
            }
        }
        void* p = nullptr;
        const uint64_t b = std::max<uint64_t>((bytes + (bytes >> 3) + 4095) & ~4095ull, 1ull << 16);
        HIP_CHECK(hipHostMalloc(&p, b, hipHostMallocPortable));
        std::lock_guard<std::mutex> lk(mu);
        live.emplace_back(p, b);
        *cap = b;
        return p;
    }
    bool put(void* p) {  // false: not one of ours
        std::pair<void*, uint64_t> drop{nullptr, 0};
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t i = 0;
            while (i < live.size() && live[i].first != p) i++;
            if (i == live.size()) return false;
            auto e = live[i];
            live.erase(live.begin() + i);
            free_list.push_back(e);
            if (free_list.size() > kKeep) {  // drop the smallest
                size_t m = 0;
                for (size_t k = 1; k < free_list.size(); k++)
                    if (free_list[k].second < free_list[m].second) m = k;
                drop = free_list[m];
                free_list.erase(free_list.begin() + m);
            }
        }
        if (drop.first) (void)hipHostFree(drop.first);
        return true;
    }
};
PinnedPool& pool() {
    static PinnedPool* p = new PinnedPool;  // never destroyed: arrays may outlive static destruction order
    return *p;
}
}  // namespace

mtsv_hit* pinned_hits_alloc(uint64_t n_hits, uint64_t* cap_hits) {
    uint64_t cap = 0;
    void* p = pool().get(std::max<uint64_t>(n_hits, 1) * sizeof(mtsv_hit), &cap);
    *cap_hits = cap / sizeof(mtsv_hit);
    return (mtsv_hit*)p;
}
bool pinned_hits_release(void* p) { return p && pool().put(p); }

// Host buffers in, hits in pinned host memory out.
//
// The PCIe rate of the reads (~50 GB/s = 330 k reads of 150 bases per ms) and the rate of the kernels are about
// equal, so the copy must run flat out from the first microsecond and the kernels must follow right behind it:
//   * one FEEDER thread copies the batch, strictly in read order, in chunks of a few MB (growing to 32 MB) on one
//     copy stream into an ARENA in HBM that holds the whole batch (or 3 GiB segments of it, two arenas taking
//     turns), an event after every chunk.  Bases in page-locked memory are copied from where they lie, others are
//     staged through a few page-locked chunk buffers; the u64 offsets are narrowed to arena-relative u32 on the way;
//   * the LANES -- a host thread and a stream each -- take the reads that HAVE ARRIVED and nobody has taken yet:
//     a pass is a range of reads of the arena, as large as the ramp allows and the copy has delivered, never
//     tied to how the copy was cut.  An idle device takes what is there; while other lanes keep it busy a lane
//     waits for a worthwhile range (the ramp grows with the reads already taken and falls towards the end, so
//     the lanes finish together);
//   * a finished range's hits leave for the pinned result array at once, in read order, on a third stream.
// ---------------------------------------------------------------------------------------------
void Batch::run_host(const uint8_t* bases, const uint64_t* read_off, uint64_t n, const mtsv_params& p, uint64_t read_base) {
    const HostPart one{bases, read_off, n};
    run_host_parts(&one, 1, p, read_base);
}

// The same for a batch that lies in several pieces (mtsv_batch_run_host_parts: a host that parses its input in blocks hands
// several blocks to one call -- larger passes on the device -- without putting them together first).  The reads are
// numbered through the parts in order.
// What run_host sizes by its batch: the copy streams, the device arenas, the page-locked offset table.
void Batch::host_room(uint64_t n, uint64_t total_bases, bool trace) {
    if (!copy_stream) {
        // The copy streams get a priority of their own.  The runtime maps streams onto a few hardware queues (four by
        // default, GPU_MAX_HW_QUEUES) round robin, per priority; an asynchronous copy holds its queue with a barrier
        // packet until the copy engine is done, and with it every kernel of any stream that shares the queue -- a lane
        // that shared one with the reads' transfer did not get a kernel in for 20 ms at a time.
        int prio_low = 0, prio_high = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        HIP_CHECK(hipStreamCreateWithPriority(&copy_stream, hipStreamNonBlocking, prio_high));
        HIP_CHECK(hipStreamCreateWithPriority(&copy_stream2, hipStreamNonBlocking, prio_high));
    }
    uint64_t arena_bases = kArenaBases, arena_reads = kArenaReads;
    if (const char* e = getenv("MTSV_ARENA_BASES")) arena_bases = std::max<uint64_t>(1 << 16, strtoull(e, nullptr, 10));  // (tests)
    const bool one_segment = total_bases <= arena_bases && n <= arena_reads;
    {
        const uint64_t want_b = one_segment ? total_bases : arena_bases, want_r = one_segment ? n : std::min(n, arena_reads);
        for (int k = 0; k < (one_segment ? 1 : 2); k++) {
            Arena& ar = arena[k];
            if (ar.cap_bases < want_b || ar.cap_reads < want_r) {
                if (trace) fprintf(stderr, "[run_host] arena %d: %.1f MB of bases, %llu reads\n", k, want_b / 1e6, (unsigned long long)want_r);
                (void)hipFree(ar.d_bases);
                (void)hipFree(ar.d_packed);
                (void)hipFree(ar.d_off);
                ar.d_bases = nullptr;
                ar.d_packed = nullptr;
                ar.d_off = nullptr;
                ar.cap_bases = ar.cap_reads = 0;
                const uint64_t cb = want_b + want_b / 16, cr = want_r + want_r / 16;  // a little room: the next batch is rarely the same size
                dev_alloc(&ar.d_bases, std::min(cb, arena_bases) + 64, &bytes);  // k_search reads up to 36 bytes past a seed start
                dev_alloc(&ar.d_packed, std::min(cb, arena_bases) / 2 + 64, &bytes);
                dev_alloc(&ar.d_off, std::min(cr, arena_reads) + 1, &bytes);
                ar.cap_bases = std::min(cb, arena_bases);
                ar.cap_reads = std::min(cr, arena_reads);
            }
        }
    }
    // the whole batch's narrowed offsets, every segment with a closing entry of its own (a segment is closed when the next
    // read does not fit: it is more than half full); run_slice looks at read lengths on the host when a range holds long reads
    const uint64_t off_need = n + 2 + (one_segment ? 0 : 2 * (total_bases / arena_bases + n / arena_reads) + 8);
    if (h_off_cap < off_need) {
        if (h_off_all) (void)hipHostFree(h_off_all);
        h_off_all = nullptr;
        h_off_cap = off_need + n / 16;
        HIP_CHECK(hipHostMalloc((void**)&h_off_all, h_off_cap * sizeof(uint32_t)));
    }
}

void Batch::reserve_host(uint64_t n, uint64_t n_bases) {
    HIP_CHECK(hipSetDevice(di->device));
    host_room(n, n_bases, getenv("MTSV_TRACE") != nullptr);
    if (!keep_on_device) {  // a result array of the size the first call will ask for, parked in the pool
        uint64_t cap = 0;
        mtsv_hit* h = pinned_hits_alloc(n + n / 8, &cap);
        pinned_hits_release(h);
    }
}

void Batch::run_host_parts(const HostPart* parts, int n_parts, const mtsv_params& p, uint64_t read_base) {
    const double t_entry = now_s();
    HIP_CHECK(hipSetDevice(di->device));
    std::vector<Batch*> ls{this};
    for (auto& l : extra) ls.push_back(l.get());
    const bool trace = getenv("MTSV_TRACE") != nullptr;
    // the parts as one batch: reads numbered through them, a virtual byte offset that ascends through them
    struct PartView {
        const uint8_t* bases;
        const uint64_t* off;
        uint64_t n, first, virt;
        bool pinned;
    };
    std::vector<PartView> pv;
    uint64_t n = 0, total_bases = 0;
    for (int k = 0; k < n_parts; k++) {
        const HostPart& hp = parts[k];
        if (!hp.n) continue;
        if (!hp.read_off || hp.read_off[hp.n] < hp.read_off[0]) throw std::runtime_error("arg: read_off is not ascending");
        const uint64_t nb = hp.read_off[hp.n] - hp.read_off[0];
        if (nb && !hp.bases) throw std::runtime_error("arg: null bases");
        // bases in page-locked memory (mtsv_host_alloc / mtsv_host_register) go to the GPU from where they lie
        const bool pin = nb && host_pinned(hp.bases + hp.read_off[0], nb) && !getenv("MTSV_STAGE_ALWAYS");
        pv.push_back(PartView{hp.bases, hp.read_off, hp.n, n, total_bases, pin});
        n += hp.n;
        total_bases += nb;
    }
    const uint64_t first_base = 0;
    auto part_of = [&](uint64_t r) {  // the part read r lies in (r == n: the last part)
        size_t k = pv.size() - 1;
        while (k > 0 && pv[k].first > r) k--;
        return k;
    };
    auto RO = [&](uint64_t r) {  // virtual offset of read r's first base (r == n: the end of the batch)
        if (pv.empty()) return (uint64_t)0;
        const PartView& v = pv[part_of(r)];
        return v.virt + (v.off[r - v.first] - v.off[0]);
    };
    bool any_staged = false, any_direct = false;
    for (auto& v : pv) (v.pinned ? any_direct : any_staged) = true;
    const bool direct = !any_staged;
    if (trace) fprintf(stderr, "[run_host] input %s%s\n", direct ? "page-locked: copied from the caller's buffer" : "pageable: staged",
                       any_staged && any_direct ? " (some parts page-locked)" : "");

    // ---- arenas: segments of at most kArenaBases bases / kArenaReads reads (u32 offsets inside a segment) ----
    host_room(n, total_bases, trace);
    constexpr uint64_t kChunkMax = 32ull << 20, kChunkMin = 4ull << 20;
    // The bases cross PCIe as 4-bit codes, packed on the host chunk by chunk into page-locked staging buffers while the
    // chunk before is on its way -- when the process has the CPUs for it (host_pack.hpp).  Otherwise, and with
    // MTSV_H2D_PLAIN=1: the bytes as they are (from page-locked memory in place) and k_normalise on the device.
    // (one host batch at a time has the packer: a second one that runs beside it in this process -- another device of
    //  mtsv_bin_batch_multi, another worker -- would wait for the pool chunk by chunk, and sends its bytes plain)
    static std::atomic<int> host_runs{0};
    struct HostRun {
        std::atomic<int>& c;
        const int mine;
        explicit HostRun(std::atomic<int>& c_) : c(c_), mine(++c_) {}
        ~HostRun() { c--; }
    } host_run{host_runs};
    const bool packed = !getenv("MTSV_H2D_PLAIN") && pack_threads() >= kPackWorthwhile && host_run.mine == 1;
    if ((packed || !direct) && !h_stage[0])
        for (auto& hs : h_stage) HIP_CHECK(hipHostMalloc((void**)&hs, kChunkMax + 64));

    struct Chunk {
        uint64_t begin = 0, end = 0;  // reads
        uint32_t max_len = 0, seg = 0;
        hipEvent_t ev = nullptr;
        bool complete = false;
    };
    struct Range {
        uint64_t begin = 0, end = 0;
        bool done = false;
        Batch* lane = nullptr;
        uint64_t hit_off = 0, hit_cnt = 0;
    };
    std::mutex mu;  // chunks, ranges, cursors, commit state
    std::condition_variable cv;
    std::vector<Chunk> chunks;            // issued copies, in read order
    std::vector<uint64_t> seg_begin{0};   // first read of every segment (feeder appends)
    uint64_t n_complete = 0;              // chunks [0, n_complete) have arrived
    bool issued_all = n == 0;
    std::vector<Range> ranges;
    uint64_t next_read = 0;    // first read no lane has taken
    uint64_t next_commit = 0;  // next range whose hits go to the host
    uint64_t done_prefix = 0;  // reads [0, done_prefix) are through the kernels (arena reuse)
    int busy_lanes = 0;
    bool abort = false;
    std::exception_ptr first_err;
    auto fail = [&](std::exception_ptr e) {
        std::lock_guard<std::mutex> lk(mu);
        if (!first_err) first_err = e;
        abort = true;
        cv.notify_all();
    };
    auto event_for = [&](size_t k) {
        while (chunk_ev.size() <= k) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            chunk_ev.push_back(e);
        }
        return chunk_ev[k];
    };

    n_reads = n;
    max_len = 0;
    if (trace) fprintf(stderr, "[run_host] entered; begin_run at %.2f ms\n", (now_s() - t_entry) * 1e3);
    begin_run(p);
    staged_hits = 0;
    if (!h_hits_stage && !keep_on_device) {
        // expect about as many hits as the last run produced (first run: one per read)
        // (a fresh page-locked array costs ~60 us per MB to create and is slow on its first copy: ask for little more than
        //  the last batch needed, so that the array that batch returned to the pool fits again)
        h_hits_stage = pinned_hits_alloc(last_total_hits ? last_total_hits + last_total_hits / 64 : n + n / 8, &h_hits_cap);
    }
    if (trace) fprintf(stderr, "[run_host] result array of %llu hits ready at %.2f ms\n", (unsigned long long)h_hits_cap, (now_s() - t_entry) * 1e3);
    const uint64_t n_lanes_used = n >= ls.size() * kLaneMinReads ? ls.size() : 1;
    lanes_used = n_lanes_used;

    // ---- feeder ----
    // Only the WATCHER thread looks at the copies' events, by polling, and publishes what has arrived (lanes that
    // blocked in hipEventSynchronize while a third lane sat in hipStreamSynchronize kept that lane from returning
    // for 10 ms at a time).
    auto watcher = [&]() {
        try {
            HIP_CHECK(hipSetDevice(di->device));
            size_t c = 0;
            for (;;) {
                size_t upto;
                bool last;
                std::vector<hipEvent_t> evs;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (abort) return;
                    upto = chunks.size();
                    last = issued_all;
                    for (size_t i = c; i < upto; i++) evs.push_back(chunks[i].ev);
                }
                const size_t c0 = c;
                while (c < upto) {
                    const hipError_t q = hipEventQuery(evs[c - c0]);
                    if (q == hipSuccess) c++;
                    else if (q == hipErrorNotReady) break;
                    else throw_hip(q, "hipEventQuery(chunk)", __FILE__, __LINE__);
                }
                if (c != c0) {
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        n_complete = c;
                    }
                    cv.notify_all();
                }
                if (last && c == upto) return;
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
        } catch (...) {
            fail(std::current_exception());
        }
    };
    auto feeder = [&]() {
        try {
            HIP_CHECK(hipSetDevice(di->device));
            uint64_t r = 0, seg = 0, seg_first_base = first_base, seg_first_read = 0;
            uint64_t chunk_bytes = kChunkMin;
            size_t k = 0;
            uint8_t prev_code = 0;  // packed transfer: the code of the segment's last base so far (it may share a byte with the next)
            while (r < n) {
                // the chunk: whole reads, about chunk_bytes of bases, inside the current segment
                uint64_t e = r;
                {
                    // largest e with read_off[e] - read_off[r] <= chunk_bytes (at least one read)
                    uint64_t lo = r + 1, hi = n;
                    hi = pv[part_of(r)].first + pv[part_of(r)].n;  // a chunk is copied from one part
                    const uint64_t lim = RO(r) + chunk_bytes;
                    if (RO(hi) > lim) {
                        while (lo < hi) {
                            const uint64_t mid = (lo + hi + 1) / 2;
                            if (RO(mid) <= lim) lo = mid;
                            else hi = mid - 1;
                        }
                        hi = lo;
                    }
                    e = hi;
                }
                Arena* ar = &arena[seg & 1];
                if (RO(e) - seg_first_base > ar->cap_bases || e - seg_first_read > ar->cap_reads) {
                    // does not fit the segment any more: shorten to what fits, or open the next segment
                    uint64_t lo = r, hi = e;
                    while (lo < hi) {
                        const uint64_t mid = (lo + hi + 1) / 2;
                        if (RO(mid) - seg_first_base <= ar->cap_bases && mid - seg_first_read <= ar->cap_reads) lo = mid;
                        else hi = mid - 1;
                    }
                    if (lo == r) {
                        if (r == seg_first_read) throw std::runtime_error("limit: one read holds more bases than an input segment (" + std::to_string(ar->cap_bases) + ")");
                        seg++;
                        seg_first_base = RO(r);
                        seg_first_read = r;
                        ar = &arena[seg & 1];
                        {
                            std::lock_guard<std::mutex> lk(mu);
                            seg_begin.push_back(r);
                        }
                        cv.notify_all();  // lanes waiting for more reads of the closed segment take what it has
                        if (seg >= 2) {  // the arena's previous tenant (segment seg - 2) must be through the kernels
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return abort || done_prefix >= seg_begin[seg - 1]; });
                            if (abort) return;
                        }
                        continue;  // cut the chunk again inside the new segment
                    }
                    e = lo;
                }
                const PartView& src_part = pv[part_of(r)];
                const uint64_t* poff = src_part.off + (r - src_part.first);  // this chunk's offsets inside its part
                const uint64_t cnt = e - r, b0 = RO(r), nb = RO(e) - b0;
                const uint64_t dst_b = b0 - seg_first_base, dst_r = r - seg_first_read;
                // offsets relative to the segment's first base; h_off_all[seg ... ] holds them at index (read + seg) so that
                // every segment has its own closing entry
                uint32_t* ho = h_off_all + r + seg;
                if (r + seg + cnt + 1 > h_off_cap) throw std::runtime_error("internal: more input segments than the offset array was sized for");
                uint32_t ml = 0;
                if (r == seg_first_read) ho[0] = 0;  // (else the previous chunk wrote this entry; a lane may be reading it)
                {
                    // (on the packer's threads when it runs: a quarter of a million reads per chunk were 0.15 ms of the
                    //  feeder's 0.56 per chunk)
                    std::atomic<uint32_t> ml_all{0};
                    std::atomic<bool> bad{false};
                    auto narrow = [&](uint64_t i0, uint64_t i1) {  // entries i0 + 1 .. i1
                        uint32_t m = 0;
                        for (uint64_t i = i0 + 1; i <= i1; i++) {
                            if (poff[i] < poff[i - 1]) bad.store(true);
                            const uint32_t v = (uint32_t)(b0 + (poff[i] - poff[0]) - seg_first_base);
                            m = std::max(m, (uint32_t)(poff[i] - poff[i - 1]));
                            ho[i] = v;
                        }
                        uint32_t cur = ml_all.load();
                        while (m > cur && !ml_all.compare_exchange_weak(cur, m)) {}
                    };
                    if (packed && cnt >= (1u << 16)) pack_pool_for(cnt, 1u << 14, narrow);
                    else narrow(0, cnt);
                    if (bad.load()) throw std::runtime_error("arg: read_off is not ascending");
                    ml = ml_all.load();
                }
                const double t0 = now_s();
                hipEvent_t ev = event_for(k);
                if (nb) {
                    const uint8_t* src = src_part.bases + poff[0];
                    if (packed || !src_part.pinned) {
                        uint8_t* hs = h_stage[k % kStage];
                        if (k >= kStage) {  // its last copy must have left (polled, like the watcher)
                            for (;;) {
                                const hipError_t q = hipEventQuery(chunk_ev[k - kStage]);
                                if (q == hipSuccess) break;
                                if (q != hipErrorNotReady) throw_hip(q, "hipEventQuery(stage)", __FILE__, __LINE__);
                                std::this_thread::sleep_for(std::chrono::microseconds(20));
                            }
                        }
                        if (packed) {
                            if (dst_b == 0) prev_code = 0;  // a new segment
                            prev_code = pack_chunk(hs, src, dst_b, nb, prev_code);
                        } else {
                            parallel_copy(hs, src, nb);
                        }
                        src = hs;
                    }
                    if (packed)
                        HIP_CHECK(hipMemcpyAsync(ar->d_packed + (dst_b >> 1), src, ((dst_b + nb + 1) >> 1) - (dst_b >> 1), hipMemcpyHostToDevice, copy_stream));
                    else
                        HIP_CHECK(hipMemcpyAsync(ar->d_bases + dst_b, src, nb, hipMemcpyHostToDevice, copy_stream));
                }
                HIP_CHECK(hipMemcpyAsync(ar->d_off + dst_r, ho, (cnt + 1) * 4, hipMemcpyHostToDevice, copy_stream));
                HIP_CHECK(hipEventRecord(ev, copy_stream));
                if (trace) fprintf(stderr, "[run_host] chunk %zu: reads %llu..%llu (%.1f MB, segment %llu) issued at %.2f ms (host side %.2f ms)\n", k, (unsigned long long)r, (unsigned long long)e, nb / 1e6, (unsigned long long)seg, (now_s() - run_t0) * 1e3, (now_s() - t0) * 1e3);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    Chunk c;
                    c.begin = r;
                    c.end = e;
                    c.max_len = ml;
                    c.seg = (uint32_t)seg;
                    c.ev = ev;
                    chunks.push_back(c);
                    if (e == n) issued_all = true;
                }
                cv.notify_all();
                r = e;
                k++;
                chunk_bytes = std::min(kChunkMax, chunk_bytes * 2);
            }
        } catch (...) {
            fail(std::current_exception());
        }
    };

    // ---- lanes ----
    auto commit_ready = [&]() {  // mu held: hits of finished ranges leave for the host in read order
        while (next_commit < ranges.size() && ranges[next_commit].done) {
            Range& rg = ranges[next_commit];
            if (rg.hit_cnt && !keep_on_device) {
                stage_reserve(staged_hits + rg.hit_cnt);
                HIP_CHECK(hipMemcpyAsync(h_hits_stage + staged_hits, rg.lane->d_hits + rg.hit_off, rg.hit_cnt * sizeof(mtsv_hit),
                                         hipMemcpyDeviceToHost, copy_stream2));
                staged_hits += rg.hit_cnt;
            }
            segments.push_back(Segment{rg.lane, rg.hit_off, rg.hit_cnt});
            done_prefix = rg.end;
            next_commit++;
        }
    };
    // (the first ranges: with the bases packed the reads arrive faster than the device takes them, and fewer, larger ranges
    //  are worth more than an early start -- same-box medians 31.1-31.3 ms per step against 31.8-32.4 with 256 Ki; with the
    //  plain transfer the order is the other way round, 36.4 against 37.0)
    uint64_t ramp_floor = packed ? 768 << 10 : 256 << 10;
    if (const char* e = getenv("MTSV_RAMP_FLOOR")) ramp_floor = std::max<uint64_t>(4096, strtoull(e, nullptr, 10));
    uint64_t idle_take = 64 << 10;  // an idle device starts on this little
    if (const char* e = getenv("MTSV_IDLE_TAKE")) idle_take = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));
    auto lane_main = [&](Batch* lane) {
        try {
            HIP_CHECK(hipSetDevice(di->device));
            uint64_t last_done = ~0ull;  // the last range this lane finished
            for (;;) {
                uint64_t k = 0, rb = 0, re = 0, seg_first = 0;
                uint32_t seg = 0, ml = 0;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    // The lane's result array only has to hold hits until they have left for the host: once it is half
                    // full and everything in it has been committed, start again at its beginning (a 100 M-read host
                    // batch would otherwise pile a third of its hits up on the device).
                    if (!keep_on_device && lane->n_hits_total > lane->hits_cap / 2 && (last_done == ~0ull || last_done < next_commit)) {
                        HIP_CHECK(hipStreamSynchronize(copy_stream2));
                        lane->n_hits_total = 0;
                    }
                    for (;;) {
                        if (abort || next_read >= n) return;  // failed elsewhere / every read taken
                        const uint64_t arrived = n_complete ? chunks[n_complete - 1].end : 0;
                        const bool all_arrived = issued_all && n_complete == chunks.size();
                        // the segment next_read lies in: a range stays inside one
                        size_t sg = seg_begin.size() - 1;
                        while (seg_begin[sg] > next_read) sg--;
                        const uint64_t seg_end = sg + 1 < seg_begin.size() ? seg_begin[sg + 1] : n;
                        const uint64_t avail = std::min(arrived, seg_end) > next_read ? std::min(arrived, seg_end) - next_read : 0;
                        // the ramp: grows with the reads already taken, falls with the reads left
                        uint64_t want = ws_reads;
                        if (n > 2 * ws_reads) {
                            const uint64_t floor_reads = std::min<uint64_t>(ws_reads, ramp_floor);
                            const uint64_t up = floor_reads + next_read / n_lanes_used;
                            const uint64_t down = std::max(floor_reads, (n - next_read) / (n_lanes_used + 1));
                            want = std::min(ws_reads, std::min(up, down));
                        }
                        uint64_t take = 0;
                        if (avail >= want) take = want;
                        else if (avail && (all_arrived || arrived >= seg_end)) take = avail;            // nothing more will come for it
                        else if (busy_lanes == 0 && avail >= std::min(idle_take, want)) take = avail;  // do not leave the device idle
                        if (take) {
                            rb = next_read;
                            re = next_read + take;
                            next_read = re;
                            seg = (uint32_t)sg;
                            seg_first = seg_begin[sg];
                            for (size_t c = 0; c < n_complete; c++)  // (a few dozen chunks)
                                if (chunks[c].end > rb && chunks[c].begin < re) ml = std::max(ml, chunks[c].max_len);
                            k = ranges.size();
                            ranges.emplace_back();
                            ranges[k].begin = rb;
                            ranges[k].end = re;
                            busy_lanes++;
                            break;
                        }
                        cv.wait(lk);  // for the next copy to arrive (the feeder watches them), or for a lane to finish (the idle rule)
                    }
                }
                cv.notify_all();
                const Arena& ar = arena[seg & 1];
                const uint32_t* ho = h_off_all + rb + seg;  // offsets of reads rb .. re inside segment seg
                const uint32_t* dso = ar.d_off + (rb - seg_first);
                const uint64_t before = lane->n_hits_total;
                const double t0 = now_s();
                // base normalisation (binner.rs:88-100) in place, on the lane's own stream: a kernel on the copy
                // stream would queue behind the persistent verification kernels of the other lanes
                if (packed) launch_unpack(lane->stream, ar.d_packed, ar.d_bases, ho[0], ho[re - rb]);
                else launch_normalise(lane->stream, ar.d_bases, ar.d_bases, ho[0], ho[re - rb]);
                lane->run_slice(p, ar.d_bases, dso, ho, re - rb, ml, read_base + rb);
                if (trace) fprintf(stderr, "[run_host] range %llu (reads %llu..%llu): kernels %.1f ms, done at %.1f ms\n", (unsigned long long)k, (unsigned long long)rb, (unsigned long long)re, (now_s() - t0) * 1e3, (now_s() - run_t0) * 1e3);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ranges[k].lane = lane;
                    ranges[k].hit_off = before;
                    ranges[k].hit_cnt = lane->n_hits_total - before;
                    ranges[k].done = true;
                    last_done = k;
                    max_len = std::max(max_len, ml);
                    busy_lanes--;
                    commit_ready();
                }
                cv.notify_all();
            }
        } catch (...) {
            fail(std::current_exception());
        }
    };

    commit_mu = &mu;
    std::vector<std::thread> th;
    th.emplace_back(feeder);
    th.emplace_back(watcher);
    for (uint64_t i = 1; i < n_lanes_used; i++) th.emplace_back(lane_main, ls[i]);
    lane_main(this);
    for (auto& t : th) t.join();
    commit_mu = nullptr;
    if (trace) fprintf(stderr, "[run_host] threads joined at %.1f ms\n", (now_s() - run_t0) * 1e3);
    if (first_err) {
        (void)hipStreamSynchronize(copy_stream);
        (void)hipStreamSynchronize(copy_stream2);
        std::rethrow_exception(first_err);
    }
    end_run();
    HIP_CHECK(hipStreamSynchronize(copy_stream2));
    staged_valid = !keep_on_device && staged_hits == total_hits;
    if (trace) fprintf(stderr, "[run_host] hits on the host at %.1f ms (%llu ranges, %llu chunks); %.1f ms since the call began\n", (now_s() - run_t0) * 1e3, (unsigned long long)ranges.size(), (unsigned long long)chunks.size(), (now_s() - t_entry) * 1e3);
    if (!staged_valid && !keep_on_device) throw std::runtime_error("internal: run_host staged " + std::to_string(staged_hits) + " of " + std::to_string(total_hits) + " hits");
}

// pinned result array of run_host: grown geometrically through the pool (mu of run_host held)
void Batch::stage_reserve(uint64_t n_hits_needed) {
    if (n_hits_needed <= h_hits_cap) return;
    if (getenv("MTSV_TRACE")) fprintf(stderr, "[run_host] result array grows %llu -> %llu hits\n", (unsigned long long)h_hits_cap, (unsigned long long)n_hits_needed);
    HIP_CHECK(hipStreamSynchronize(copy_stream2));
    uint64_t ncap = 0;
    mtsv_hit* nh = pinned_hits_alloc(std::max<uint64_t>(2 * h_hits_cap, n_hits_needed), &ncap);
    if (staged_hits) memcpy(nh, h_hits_stage, staged_hits * sizeof(mtsv_hit));
    if (h_hits_stage) pinned_hits_release(h_hits_stage);
    h_hits_stage = nh;
    h_hits_cap = ncap;
}

// The result array is pinned host memory from the pool; the caller owns it until mtsv_hits_free.
void Batch::download(mtsv_hit** hits, uint64_t* n) {
    HIP_CHECK(hipSetDevice(di->device));
    last_total_hits = total_hits;
    if (staged_valid) {  // run_host already brought them over
        *hits = h_hits_stage;
        *n = total_hits;
        h_hits_stage = nullptr;
        h_hits_cap = 0;
        staged_valid = false;
        return;
    }
    uint64_t cap = 0;
    mtsv_hit* h = pinned_hits_alloc(total_hits, &cap);
    uint64_t at = 0;
    for (auto& sg : segments) {
        if (!sg.count) continue;
        hipError_t e = hipMemcpy(h + at, sg.lane->d_hits + sg.offset, sg.count * sizeof(mtsv_hit), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            pinned_hits_release(h);
            throw_hip(e, "hipMemcpy(hits)", __FILE__, __LINE__);
        }
        at += sg.count;
    }
    *hits = h;
    *n = total_hits;
}

void Batch::download_into(mtsv_hit* dst, uint64_t n) {
    HIP_CHECK(hipSetDevice(di->device));
    if (n != total_hits) throw std::runtime_error("internal: download_into of " + std::to_string(n) + " hits, the run produced " + std::to_string(total_hits));
    last_total_hits = total_hits;
    hipStream_t cs = copy_stream2 ? copy_stream2 : stream;
    uint64_t at = 0;
    for (auto& sg : segments) {
        if (!sg.count) continue;
        HIP_CHECK(hipMemcpyAsync(dst + at, sg.lane->d_hits + sg.offset, sg.count * sizeof(mtsv_hit), hipMemcpyDeviceToHost, cs));
        at += sg.count;
    }
    HIP_CHECK(hipStreamSynchronize(cs));
}

}  // namespace mtsv

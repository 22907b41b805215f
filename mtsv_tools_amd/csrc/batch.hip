// batch.hip -- device workspace for one batch of reads and the stage-by-stage driver of the hot
// path.  Replaces the producer -> workers -> joiner queue of vendor/cue/src/lib.rs:45-105 with
// large HBM-resident batches: reads stay on the device between stages, every stage is one launch
// over the whole batch, and the only host round-trips are two 8-byte totals per pass.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/mtsv_amd.h"
#include "batch.hpp"
#include "kernels.hpp"

namespace mtsv {

namespace {
template <class T>
void dev_alloc(T** p, uint64_t count, uint64_t* bytes) {
    uint64_t b = std::max<uint64_t>(count, 1) * sizeof(T);
    HIP_CHECK(hipMalloc((void**)p, b));
    *bytes += b;
}
}  // namespace

Batch::Batch(mtsv_index* ix_, DeviceIndex* di_, uint64_t max_reads_, uint64_t max_bases_, uint64_t hit_cap_)
    : ix(ix_), di(di_), max_reads(max_reads_), max_bases(max_bases_), hit_cap(hit_cap_) {
    if (max_reads == 0) max_reads = 1;
    if (max_reads > 0x7fffffffull) throw std::runtime_error("limit: more than 2^31 reads in one batch");
    if (max_bases >= 0xffffffffull) throw std::runtime_error("limit: 4 GiB of bases or more in one batch");
    if (hit_cap == 0) hit_cap = std::max<uint64_t>(1ull << 20, 32 * max_reads);
    if (hit_cap > 0xfffffff0ull) hit_cap = 0xfffffff0ull;
    hits_cap = std::max<uint64_t>(1ull << 20, 16 * max_reads);
    HIP_CHECK(hipSetDevice(di->device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    const uint64_t ns = 2 * max_reads;
    dev_alloc(&d_bases, max_bases + 64, &bytes);  // k_search reads up to 36 bytes past a seed start with dword loads
    dev_alloc(&d_read_off, max_reads + 1, &bytes);
    dev_alloc(&d_strand_hits, ns + 1, &bytes);
    dev_alloc(&d_strand_nseeds, ns, &bytes);
    dev_alloc(&d_strand_off, ns + 1, &bytes);
    dev_alloc(&d_strand_ncand, ns, &bytes);
    dev_alloc(&d_strand_nout, ns + 1, &bytes);
    dev_alloc(&d_heavy_list, ns, &bytes);
    dev_alloc(&d_out_off, ns + 1, &bytes);
    dev_alloc(&d_tile_sums, (uint64_t)scan_tiles((uint32_t)ns) + 1, &bytes);
    dev_alloc(&d_counters, 8, &bytes);
    dev_alloc(&d_hit_row, hit_cap, &bytes);
    dev_alloc(&d_hit_ref, hit_cap, &bytes);
    dev_alloc(&d_hit_q, hit_cap, &bytes);
    dev_alloc(&d_hit_key, hit_cap, &bytes);
    dev_alloc(&d_cand_tmp, 2 * hit_cap, &bytes);
    dev_alloc(&d_cand, hit_cap, &bytes);
    dev_alloc(&d_out, hit_cap, &bytes);
    dev_alloc(&d_cand_next, hit_cap, &bytes);
    dev_alloc(&d_cand_rs, hit_cap, &bytes);
    dev_alloc(&d_cand_status, hit_cap, &bytes);
    dev_alloc(&d_worklist, hit_cap, &bytes);
    dev_alloc(&d_hits, hits_cap, &bytes);
    HIP_CHECK(hipHostMalloc((void**)&h_counters, 8 * sizeof(uint64_t)));
    if (const char* e = getenv("MTSV_VERIFY")) verify_mode = !strcmp(e, "edit_first") ? 1 : 0;
    for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
}

Batch::~Batch() {
    (void)hipSetDevice(di->device);
    (void)hipStreamSynchronize(stream);
    for (void* p : {(void*)d_bases, (void*)d_read_off, (void*)d_seed_lo, (void*)d_seed_cnt, (void*)d_seed_pre, (void*)d_strand_hits,
                    (void*)d_strand_nseeds, (void*)d_strand_off, (void*)d_strand_ncand, (void*)d_worklist,
                    (void*)d_strand_nout, (void*)d_out_off, (void*)d_tile_sums, (void*)d_counters, (void*)d_hit_row,
                    (void*)d_hit_ref, (void*)d_hit_q, (void*)d_hit_key, (void*)d_cand_tmp, (void*)d_cand, (void*)d_out,
                    (void*)d_hits, (void*)d_cand_next, (void*)d_cand_rs, (void*)d_cand_status, (void*)d_heavy_list})
        (void)hipFree(p);
    (void)hipHostFree(h_counters);
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(stream);
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
}

// counters in d_counters: [0] scan total (u64), [1] wl_count (u32), [2] lf_steps, [3] n_cand,
// [4] n_verified, [5] window_bytes, [6] second scan total
void Batch::run(const mtsv_params& p) {
    if (!(p.edit_rate >= 0.0 && p.edit_rate <= 1.0)) throw std::runtime_error("arg: edit_rate must be within [0, 1]");
    if (!(p.min_seed >= 0.0) || !std::isfinite(p.min_seed)) throw std::runtime_error("arg: min_seed must be finite and >= 0");
    if (p.seed_size == 0 || p.seed_interval == 0) throw std::runtime_error("arg: seed_size and seed_interval must be > 0");
    if (max_len > kMaxReadLen)
        throw std::runtime_error("limit: read of " + std::to_string(max_len) + " bases; this build verifies reads up to " +
                                 std::to_string(kMaxReadLen));
    HIP_CHECK(hipSetDevice(di->device));
    const DevIndexView& v = di->view;
    const uint32_t K = p.seed_size, G = p.seed_interval;
    const uint32_t max_ns = max_len >= K ? (max_len - K) / G + 1 : 0;
    memset(&stats, 0, sizeof stats);
    stats.n_reads = n_reads;
    n_hits_total = 0;
    float stage_ms[MTSV_N_STAGES] = {0};
    HIP_CHECK(hipMemsetAsync(d_counters, 0, 8 * sizeof(uint64_t), stream));

    uint64_t pass_reads = n_reads;
    uint64_t r0 = 0;
    hipEvent_t t_begin = ev[8], t_end = ev[9];
    HIP_CHECK(hipEventRecord(t_begin, stream));
    while (r0 < n_reads) {
        const uint32_t nr = (uint32_t)std::min<uint64_t>(pass_reads, n_reads - r0);
        const uint32_t nstr = nr * 2;
        const uint64_t slots = (uint64_t)nstr * max_ns;
        if (slots > seed_cap) {
            (void)hipFree(d_seed_lo);
            (void)hipFree(d_seed_cnt);
            (void)hipFree(d_seed_pre);
            d_seed_lo = d_seed_cnt = d_seed_pre = nullptr;
            uint64_t dummy = 0;
            dev_alloc(&d_seed_lo, slots, &dummy);
            dev_alloc(&d_seed_cnt, slots, &dummy);
            dev_alloc(&d_seed_pre, slots, &dummy);
            seed_cap = slots;
        }
        // ---- seeds ----
        HIP_CHECK(hipEventRecord(ev[0], stream));
        launch_search(stream, v, d_bases, d_read_off, (uint32_t)r0, nr, max_ns, K, G, d_seed_lo, d_seed_cnt);
        HIP_CHECK(hipEventRecord(ev[1], stream));
        if (max_ns)
            launch_thin(stream, d_read_off, (uint32_t)r0, nr, max_ns, K, G, p.max_hits, p.tune_max_hits, d_seed_cnt,
                        d_seed_pre, d_strand_hits, d_strand_nseeds);
        else {
            HIP_CHECK(hipMemsetAsync(d_strand_hits, 0, (uint64_t)nstr * 4, stream));
            HIP_CHECK(hipMemsetAsync(d_strand_nseeds, 0, (uint64_t)nstr * 4, stream));
        }
        launch_scan(stream, d_strand_hits, nstr, d_tile_sums, d_counters + 0, d_strand_off);
        HIP_CHECK(hipMemcpyAsync(h_counters, d_counters, 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipEventRecord(ev[2], stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        const uint64_t total_hits = h_counters[0];
        if (total_hits > hit_cap) {
            if (nr == 1)
                throw std::runtime_error("device: one read has " + std::to_string(total_hits) +
                                         " seed hits, more than the hit workspace (" + std::to_string(hit_cap) + ")");
            pass_reads = std::max<uint64_t>(1, nr / 2);
            continue;  // redo this pass with fewer reads
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
        launch_coalesce(stream, v, d_read_off, (uint32_t)r0, nstr, p.edit_rate, p.min_seed, p.max_candidates, d_strand_off,
                        d_strand_nseeds, d_hit_ref, d_hit_q, d_hit_key, d_cand_tmp, d_cand, d_cand_next, d_cand_rs,
                        d_cand_status, d_strand_ncand, d_worklist, (uint32_t*)(d_counters + 1), d_heavy_list,
                        (uint32_t*)(d_counters + 1) + 1, (unsigned long long*)(d_counters + 3));
        HIP_CHECK(hipEventRecord(ev[5], stream));
        // ---- verify: rounds over the same-TaxId chains ----
        {
            EvalArgs a;
            a.bases = d_bases;
            a.read_off = d_read_off;
            a.r0 = (uint32_t)r0;
            a.edit_rate = p.edit_rate;
            a.max_candidates = p.max_candidates;
            a.strand_off = d_strand_off;
            a.cand = d_cand;
            a.cand_next = d_cand_next;
            a.cand_rs = d_cand_rs;
            a.cand_status = d_cand_status;
            a.out = d_out;
            a.n_verified = (unsigned long long*)(d_counters + 4);
            a.window_bytes = (unsigned long long*)(d_counters + 5);
            // one launch: a group whose candidate fails walks on to the next candidate of the same TaxId
            a.worklist = d_worklist;
            a.wl_count = (const uint32_t*)(d_counters + 1);
            a.wl_cursor = (uint32_t*)(d_counters + 7);
            if (verify_mode == 1 && max_len <= 253)
                launch_edit_myers(stream, v, a, total_hits, max_len);
            else
                launch_evaluate(stream, v, a, total_hits, max_len);
            stats.n_rounds = 1;
            launch_resolve(stream, nstr, p.max_candidates, p.max_assignments, d_strand_off, d_strand_ncand, d_cand_status,
                           d_out, d_strand_nout);
        }
        HIP_CHECK(hipEventRecord(ev[6], stream));
        // ---- gather ----
        launch_scan(stream, d_strand_nout, nstr, d_tile_sums, d_counters + 6, d_out_off);
        HIP_CHECK(hipMemcpyAsync(h_counters, d_counters, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
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
            (void)hipFree(d_hits);
            d_hits = nh;
            hits_cap = ncap;
        }
        launch_gather(stream, nstr, (uint32_t)r0, d_strand_off, d_strand_nout, d_out_off, d_out, d_hits, n_hits_total);
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
    HIP_CHECK(hipEventRecord(t_end, stream));
    HIP_CHECK(hipMemcpyAsync(h_counters, d_counters, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipEventElapsedTime(&stage_ms[7], t_begin, t_end));
    for (int s = 0; s < MTSV_N_STAGES; s++) stats.stage_ms[s] = stage_ms[s];
    stats.lf_steps = h_counters[2];
    stats.n_candidates = h_counters[3];
    stats.n_verified = h_counters[4];
    stats.window_bytes = h_counters[5];
    stats.n_hits = n_hits_total;
}

void Batch::download(mtsv_hit** hits, uint64_t* n) {
    HIP_CHECK(hipSetDevice(di->device));
    mtsv_hit* h = (mtsv_hit*)malloc(std::max<uint64_t>(n_hits_total, 1) * sizeof(mtsv_hit));
    if (!h) throw std::runtime_error("nomem: result array");
    if (n_hits_total) {
        hipError_t e = hipMemcpy(h, d_hits, n_hits_total * sizeof(mtsv_hit), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            free(h);
            throw_hip(e, "hipMemcpy(hits)", __FILE__, __LINE__);
        }
    }
    *hits = h;
    *n = n_hits_total;
}

}  // namespace mtsv

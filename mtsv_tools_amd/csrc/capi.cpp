// capi.cpp -- the extern "C" boundary declared in include/mtsv_amd.h.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mtsv_amd.h"
#include "batch.hpp"
#include "dev_index.hpp"
#include "mgindex.hpp"

using namespace mtsv;

static thread_local std::string g_err;
static int g_build_device = -1;  // mtsv_set_build_device

static int fail(const std::exception& e) {
    g_err = e.what();
    auto starts = [&](const char* p) { return g_err.rfind(p, 0) == 0; };
    if (starts("io:")) return MTSV_E_IO;
    if (starts("format:")) return MTSV_E_FORMAT;
    if (starts("device:")) return MTSV_E_DEVICE;
    if (starts("limit:")) return MTSV_E_LIMIT;
    if (starts("nomem:")) return MTSV_E_NOMEM;
    if (starts("arg:")) return MTSV_E_ARG;
    return MTSV_E_DEVICE;
}
static int fail_arg(const char* msg) {
    g_err = std::string("arg: ") + msg;
    return MTSV_E_ARG;
}
#define GUARD(...)                           \
    try {                                    \
        __VA_ARGS__;                         \
        return MTSV_OK;                      \
    } catch (const std::bad_alloc&) {        \
        g_err = "nomem: host allocation";    \
        return MTSV_E_NOMEM;                 \
    } catch (const std::exception& e) {      \
        return fail(e);                      \
    }

namespace {
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint64_t below(uint64_t n) { return n ? (uint64_t)(((unsigned __int128)next() * n) >> 64) : 0; }
    bool chance(double p) { return (next() >> 11) * (1.0 / 9007199254740992.0) < p; }
};
inline uint64_t mix(uint64_t a, uint64_t b) { return SplitMix64(a ^ (b * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull)).next(); }

template <class F>
void par_for(uint64_t n, int threads, F fn) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    uint64_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        uint64_t lo = std::min(n, (uint64_t)t * chunk), hi = std::min(n, lo + chunk);
        if (lo < hi) pool.emplace_back(fn, lo, hi);
    }
    for (auto& th : pool) th.join();
}
}  // namespace

namespace {
// run fn(k) for k in [0, n) on one thread each; the first failure (by k) is rethrown in the caller
template <class F>
void fork_join(int n, F fn) {
    std::vector<std::string> err(n);
    std::vector<char> bad(n, 0);
    std::vector<std::thread> th;
    auto body = [&](int k) {
        try {
            fn(k);
        } catch (const std::exception& e) {
            err[k] = e.what();
            bad[k] = 1;
        }
    };
    for (int k = 1; k < n; k++) th.emplace_back(body, k);
    body(0);
    for (auto& t : th) t.join();
    for (int k = 0; k < n; k++)
        if (bad[k]) throw std::runtime_error(err[k]);
}
// slot of entry k: how many earlier entries name the same device
int slot_of(const int* devices, int k) {
    int s = 0;
    for (int j = 0; j < k; j++) s += devices[j] == devices[k];
    return s;
}
void release_parts(std::vector<mtsv_hit*>& parts) {
    for (auto*& p : parts) {
        if (p && !mtsv::pinned_hits_release(p)) free(p);
        p = nullptr;
    }
}
}  // namespace

extern "C" {

const char* mtsv_last_error(void) { return g_err.c_str(); }
const char* mtsv_version(void) { return "mtsv_tools_amd 0.1.0 (gfx950)"; }

void mtsv_params_default(mtsv_params* p) {  // src/bin/mtsv-binner.rs:63-94
    p->edit_rate = 0.13;
    p->seed_size = 18;
    p->seed_interval = 15;
    p->min_seed = 0.015;
    p->max_hits = 2000;
    p->tune_max_hits = 200;
    p->max_assignments = -1;
    p->max_candidates = -1;
}

int mtsv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mtsv_index_load(const char* path, mtsv_index** out) {
    if (!path || !out) return fail_arg("null argument");
    GUARD({
        auto ix = std::make_unique<mtsv_index>();
        load_index(path, ix->host);
        *out = ix.release();
    })
}

int mtsv_index_build(uint64_t n_seqs, const uint32_t* tax_ids, const uint32_t* gis, const uint8_t* const* seqs,
                     const uint64_t* seq_lens, uint32_t occ_k, uint64_t sa_s, int n_threads, mtsv_index** out) {
    if (!out || (n_seqs && (!tax_ids || !gis || !seqs || !seq_lens))) return fail_arg("null argument");
    GUARD({
        std::vector<SeqEntry> e(n_seqs);
        for (uint64_t i = 0; i < n_seqs; i++) e[i] = SeqEntry{tax_ids[i], gis[i], seqs[i], seq_lens[i]};
        auto ix = std::make_unique<mtsv_index>();
        build_index(std::move(e), occ_k, sa_s, n_threads, ix->host, g_build_device);
        *out = ix.release();
    })
}

int mtsv_set_build_device(int hip_device) {
    g_build_device = hip_device;
    return MTSV_OK;
}

int mtsv_index_build_fasta(const char* fasta_path, uint32_t occ_k, uint64_t sa_s, int n_threads, mtsv_index** out) {
    if (!fasta_path || !out) return fail_arg("null argument");
    GUARD({
        auto ix = std::make_unique<mtsv_index>();
        build_index_from_fasta(fasta_path, occ_k, sa_s, n_threads, ix->host, g_build_device);
        *out = ix.release();
    })
}

int mtsv_index_write(const mtsv_index* ix, const char* path) {
    if (!ix || !path) return fail_arg("null argument");
    GUARD({
        write_index(ix->host, path);
        const_cast<mtsv_index*>(ix)->host.file_bytes = 0;
    })
}

int mtsv_index_info(const mtsv_index* ix, mtsv_index_info_t* info) {
    if (!ix || !info) return fail_arg("null argument");
    info->n = ix->host.n();
    info->n_bins = ix->host.bins.size();
    info->occ_k = ix->host.k;
    info->sa_s = ix->host.s;
    info->file_bytes = ix->host.file_bytes;
    info->device_bytes = 0;
    info->kmer_k = 0;
    info->sa_full = 0;
    for (auto& kv : ix->dev) {
        info->device_bytes = kv.second->bytes;
        info->kmer_k = kv.second->view.kmer_tab ? kv.second->view.kmer_k : 0;
        info->sa_full = kv.second->view.sa_full ? 1 : 0;
    }
    return MTSV_OK;
}

void mtsv_index_free(mtsv_index* ix) {
    if (!ix) return;
    for (auto& kv : ix->cached_batch) delete kv.second->b;
    delete ix;
}

int mtsv_index_to_device(mtsv_index* ix, int hip_device, uint32_t flags) {
    if (!ix) return fail_arg("null index");
    GUARD({
        std::lock_guard<std::mutex> lk(ix->mu);
        auto it = ix->dev.find(hip_device);
        if (it == ix->dev.end() || it->second->flags != flags) {
            for (auto cb = ix->cached_batch.begin(); cb != ix->cached_batch.end();) {
                if (cb->first.first == hip_device) {
                    {
                        std::lock_guard<std::mutex> wl(cb->second->mu);  // no call may be running on it
                        delete cb->second->b;
                        cb->second->b = nullptr;
                    }
                    cb = ix->cached_batch.erase(cb);
                } else
                    ++cb;
            }
            if (it != ix->dev.end()) ix->dev.erase(it);
            ix->dev[hip_device] = upload_index(ix->host, hip_device, flags);
        }
    })
}

static DeviceIndex* device_index(mtsv_index* ix, int dev) {
    auto it = ix->dev.find(dev);
    if (it == ix->dev.end()) throw std::runtime_error("arg: index is not resident on device " + std::to_string(dev) + " (call mtsv_index_to_device)");
    return it->second.get();
}

int mtsv_batch_create(mtsv_index* ix, int hip_device, uint64_t max_reads, uint64_t max_bases, uint64_t max_hits_ws,
                      mtsv_batch** out) {
    if (!ix || !out) return fail_arg("null argument");
    GUARD({
        std::lock_guard<std::mutex> lk(ix->mu);
        *out = new mtsv_batch(ix, device_index(ix, hip_device), max_reads, max_bases, max_hits_ws);
    })
}

int mtsv_batch_create_lanes(mtsv_index* ix, int hip_device, uint64_t max_reads, uint64_t max_bases, uint64_t max_hits_ws, int lanes,
                            mtsv_batch** out) {
    if (!ix || !out || lanes < 0) return fail_arg("null argument");
    GUARD({
        std::lock_guard<std::mutex> lk(ix->mu);
        *out = new mtsv_batch(ix, device_index(ix, hip_device), max_reads, max_bases, max_hits_ws, nullptr, lanes);
    })
}

int mtsv_batch_reserve_host(mtsv_batch* b, uint64_t n_reads, uint64_t n_bases, uint32_t warm_read_len) {
    if (!b) return fail_arg("null argument");
    GUARD({
        b->impl.reserve_host(n_reads, n_bases);
        const std::vector<uint8_t>& text = b->impl.ix->host.text;
        if (warm_read_len && text.size() >= (uint64_t)warm_read_len * 2 + 16) {
            // reads sampled from the index through every kernel once: the first launch of a kernel loads its code object,
            // the first pass of a lane sizes its seed arrays
            const uint64_t n = std::min<uint64_t>(std::max<uint64_t>(n_reads, 1), std::max<uint64_t>(4096, (uint64_t)b->impl.n_lanes * 32768));
            std::vector<uint8_t> bases(n * warm_read_len);
            std::vector<uint64_t> off(n + 1);
            if (mtsv_synth_reads(b->impl.ix, 0x7761726d, n, warm_read_len, bases.data(), off.data()) != MTSV_OK)
                throw std::runtime_error(std::string("internal: ") + mtsv_last_error());
            mtsv_params p;
            mtsv_params_default(&p);
            b->impl.run_host(bases.data(), off.data(), n, p);
            mtsv_hit* h = nullptr;
            uint64_t nh = 0;
            b->impl.download(&h, &nh);
            mtsv_hits_free(h);
        }
    })
}

int mtsv_batch_upload(mtsv_batch* b, const uint8_t* bases, const uint64_t* read_off, uint64_t n_reads) {
    if (!b || !read_off || (!bases && n_reads && read_off[n_reads] != read_off[0])) return fail_arg("null argument");
    GUARD(b->impl.upload(bases, read_off, n_reads))
}

int mtsv_batch_run(mtsv_batch* b, const mtsv_params* params) {
    if (!b || !params) return fail_arg("null argument");
    GUARD(b->impl.run(*params))
}

int mtsv_batch_run_host(mtsv_batch* b, const uint8_t* bases, const uint64_t* read_off, uint64_t n_reads,
                        const mtsv_params* params) {
    if (!b || !read_off || !params || (!bases && n_reads && read_off[n_reads] > read_off[0])) return fail_arg("null argument");
    GUARD(b->impl.run_host(bases, read_off, n_reads, *params))
}

int mtsv_batch_run_host_parts(mtsv_batch* b, int n_parts, const uint8_t* const* bases, const uint64_t* const* read_off,
                              const uint64_t* n_reads, const mtsv_params* params) {
    if (!b || n_parts < 0 || (n_parts && (!bases || !read_off || !n_reads)) || !params) return fail_arg("null argument");
    GUARD({
        std::vector<mtsv::Batch::HostPart> parts((size_t)n_parts);
        for (int k = 0; k < n_parts; k++) {
            if (n_reads[k] && !read_off[k]) throw std::runtime_error("arg: null read_off");
            parts[(size_t)k] = mtsv::Batch::HostPart{bases[k], read_off[k], n_reads[k]};
        }
        b->impl.run_host_parts(parts.data(), n_parts, *params);
    })
}

int mtsv_batch_stats_get(const mtsv_batch* b, mtsv_batch_stats* st) {
    if (!b || !st) return fail_arg("null argument");
    *st = b->impl.stats;
    return MTSV_OK;
}

int mtsv_batch_download(mtsv_batch* b, mtsv_hit** hits, uint64_t* n_hits) {
    if (!b || !hits || !n_hits) return fail_arg("null argument");
    GUARD(b->impl.download(hits, n_hits))
}

int mtsv_batch_set_verify_mode(mtsv_batch* b, int mode) {
    if (!b || (mode != MTSV_VERIFY_REFERENCE && mode != MTSV_VERIFY_EDIT_FIRST)) return fail_arg("bad verify mode");
    b->impl.verify_mode = mode;
    return MTSV_OK;
}

int mtsv_set_default_verify_mode(int mode) {
    if (mode != MTSV_VERIFY_REFERENCE && mode != MTSV_VERIFY_EDIT_FIRST) return fail_arg("bad verify mode");
    mtsv::g_default_verify_mode = mode;
    return MTSV_OK;
}

void mtsv_batch_free(mtsv_batch* b) { delete b; }

// Three lanes of up to 1 Mi reads each (MTSV_WORKSPACE_READS overrides): ~2.7 KB of HBM per workspace read.
uint64_t mtsv_bin_batch_workspace_reads(uint64_t n_reads) {
    uint64_t cap = 3ull << 20;
    if (const char* e = getenv("MTSV_WORKSPACE_READS")) cap = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));
    return std::min<uint64_t>(std::max<uint64_t>(n_reads, 1024), cap);
}

// One host batch (or a contiguous part of one) on one device, through the workspace kept for (device, slot).
// hold == nullptr: the hits come back in a pooled page-locked array (*hits, *n_hits).
// hold != nullptr: the hits stay in HBM, *n_hits says how many, and `hold` keeps the workspace locked until the
// caller has fetched them with impl.download_into (mtsv_bin_batch_multi: every part straight into its extent of
// one result array, once all the counts are known).
struct HeldWorkspace {
    std::unique_lock<std::mutex> lk;
    mtsv_batch* b = nullptr;
};
static void bin_batch_on(mtsv_index* ix, int hip_device, int slot, const uint8_t* bases, const uint64_t* read_off, uint64_t n_reads,
                         uint64_t read_base, const mtsv_params& params, mtsv_hit** hits, uint64_t* n_hits, HeldWorkspace* hold = nullptr) {
    if (hip_device < 0) throw std::runtime_error("arg: hip_device must name a GPU: this library has no CPU path");
    uint32_t flags = MTSV_DEV_DEFAULT;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        if (ix->dev.count(hip_device)) flags = ix->dev[hip_device]->flags;
    }
    if (int rc = mtsv_index_to_device(ix, hip_device, flags); rc != MTSV_OK) throw std::runtime_error(g_err);
    mtsv_cached_workspace* ws;
    DeviceIndex* di;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        auto& slot_ws = ix->cached_batch[{hip_device, slot}];
        if (!slot_ws) slot_ws = std::make_unique<mtsv_cached_workspace>();
        ws = slot_ws.get();
        di = device_index(ix, hip_device);
    }
    std::unique_lock<std::mutex> wl(ws->mu);
    // Workspace for ranges of a bounded number of reads: larger host batches stream through it (run_host),
    // which bounds the workspace (~2.7 KB of HBM per read) and overlaps the copies with compute.
    const uint64_t nb = n_reads ? read_off[n_reads] - read_off[0] : 0;
    const uint64_t slice_reads = mtsv_bin_batch_workspace_reads(n_reads);
    const uint64_t slice_bases = n_reads > slice_reads
                                     ? std::max<uint64_t>(1 << 16, (uint64_t)((double)nb / (double)n_reads * (double)slice_reads * 1.25) + 4096)
                                     : std::max<uint64_t>(nb, 1 << 16);
    if (!ws->b || ws->b->impl.max_reads < slice_reads || ws->b->impl.max_bases < slice_bases) {
        delete ws->b;
        ws->b = nullptr;
        ws->b = new mtsv_batch(ix, di, slice_reads, slice_bases, 0);
    }
    ws->b->impl.keep_on_device = hold != nullptr;
    try {
        ws->b->impl.run_host(bases, read_off, n_reads, params, read_base);
    } catch (...) {
        ws->b->impl.keep_on_device = false;
        throw;
    }
    ws->b->impl.keep_on_device = false;
    if (hold) {
        *n_hits = ws->b->impl.total_hits;
        hold->b = ws->b;
        hold->lk = std::move(wl);
    } else {
        ws->b->impl.download(hits, n_hits);
    }
}

int mtsv_bin_batch(mtsv_index* ix, int hip_device, const uint8_t* bases, const uint64_t* read_off, uint64_t n_reads,
                   const mtsv_params* params, mtsv_hit** hits, uint64_t* n_hits) {
    if (!ix || !read_off || !params || !hits || !n_hits) return fail_arg("null argument");
    if (hip_device < 0) return fail_arg("hip_device must name a GPU: this library has no CPU path");
    GUARD(bin_batch_on(ix, hip_device, 0, bases, read_off, n_reads, 0, *params, hits, n_hits))
}


// Mode A of SURVEY 8(e): index replicated, reads in contiguous blocks, host concat.
int mtsv_bin_batch_multi(mtsv_index* ix, const int* devices, int n_devices, const uint8_t* bases, const uint64_t* read_off,
                         uint64_t n_reads, const mtsv_params* params, mtsv_hit** hits, uint64_t* n_hits) {
    if (!ix || !devices || n_devices < 1 || !read_off || !params || !hits || !n_hits) return fail_arg("null argument");
    if (n_devices == 1) {
        GUARD(bin_batch_on(ix, devices[0], 0, bases, read_off, n_reads, 0, *params, hits, n_hits))
    }
    // Two phases.  Every part runs on its device and leaves its hits in HBM; once all the counts are known, every part
    // copies its hits straight into its extent of ONE pooled page-locked array (no host-side concatenation: at 100 M
    // reads that was 3 GB of memcpy behind the slowest device).  The call returns when the slowest device has: a caller
    // that wants no such barrier runs mtsv_bin_batch per device from its own threads (what mtsv-binner --devices does).
    std::vector<HeldWorkspace> held(n_devices);
    std::vector<uint64_t> cnt(n_devices, 0);
    mtsv_hit* out = nullptr;
    try {
        fork_join(n_devices, [&](int k) {
            const uint64_t a = n_reads * (uint64_t)k / n_devices, b = n_reads * (uint64_t)(k + 1) / n_devices;
            bin_batch_on(ix, devices[k], slot_of(devices, k), bases, read_off + a, b - a, a, *params, nullptr, &cnt[k], &held[k]);
        });
        uint64_t total = 0, cap = 0;
        for (auto c : cnt) total += c;
        out = pinned_hits_alloc(total, &cap);
        std::vector<uint64_t> at(n_devices + 1, 0);
        for (int k = 0; k < n_devices; k++) at[k + 1] = at[k] + cnt[k];
        fork_join(n_devices, [&](int k) {
            held[k].b->impl.download_into(out + at[k], cnt[k]);
            held[k].lk.unlock();
        });
        *hits = out;
        *n_hits = total;
        return MTSV_OK;
    } catch (const std::exception& e) {
        if (out) pinned_hits_release(out);
        return fail(e);
    }
}

// Mode B of SURVEY 8(e): one chunk of the database per entry, every chunk sees every read, the hit lists are
// merged per read (mtsv-collapse then keeps the smallest edit per (read, TaxId): collapse.rs:597-602 -- and so
// does mtsv_format_results on the merged list).
int mtsv_bin_batch_chunks(mtsv_index* const* chunks, const int* devices, int n_chunks, const uint8_t* bases,
                          const uint64_t* read_off, uint64_t n_reads, const mtsv_params* params, mtsv_hit** hits, uint64_t* n_hits) {
    if (!chunks || !devices || n_chunks < 1 || !read_off || !params || !hits || !n_hits) return fail_arg("null argument");
    for (int k = 0; k < n_chunks; k++)
        if (!chunks[k]) return fail_arg("null chunk index");
    std::vector<mtsv_hit*> parts(n_chunks, nullptr);
    std::vector<uint64_t> cnt(n_chunks, 0);
    try {
        fork_join(n_chunks, [&](int k) {
            // distinct index handles own distinct workspaces: slot 0 each
            bin_batch_on(chunks[k], devices[k], 0, bases, read_off, n_reads, 0, *params, &parts[k], &cnt[k]);
        });
        if (n_chunks == 1) {
            *hits = parts[0];
            *n_hits = cnt[0];
            return MTSV_OK;
        }
        uint64_t total = 0, cap = 0;
        for (auto c : cnt) total += c;
        mtsv_hit* out = pinned_hits_alloc(total, &cap);
        // k-way merge by read, chunk order within a read; read ranges in parallel
        const int T = (int)std::min<uint64_t>(8, std::max<uint64_t>(1, total >> 16));
        auto lower = [&](int k, uint64_t r) {  // first hit of part k with read >= r
            return (uint64_t)(std::lower_bound(parts[k], parts[k] + cnt[k], r, [](const mtsv_hit& h, uint64_t v) { return h.read < v; }) - parts[k]);
        };
        fork_join(T, [&](int t) {
            const uint64_t r0 = n_reads * (uint64_t)t / T, r1 = n_reads * (uint64_t)(t + 1) / T;
            std::vector<uint64_t> pos(n_chunks), end(n_chunks);
            uint64_t o = 0;
            for (int k = 0; k < n_chunks; k++) {
                pos[k] = lower(k, r0);
                end[k] = t + 1 == T ? cnt[k] : lower(k, r1);
                o += pos[k];
            }
            for (;;) {
                uint64_t r = ~0ull;
                for (int k = 0; k < n_chunks; k++)
                    if (pos[k] < end[k]) r = std::min(r, parts[k][pos[k]].read);
                if (r == ~0ull) break;
                for (int k = 0; k < n_chunks; k++)
                    while (pos[k] < end[k] && parts[k][pos[k]].read == r) out[o++] = parts[k][pos[k]++];
            }
        });
        release_parts(parts);
        *hits = out;
        *n_hits = total;
        return MTSV_OK;
    } catch (const std::exception& e) {
        release_parts(parts);
        return fail(e);
    }
}

void mtsv_hits_free(mtsv_hit* hits) {
    if (hits && !mtsv::pinned_hits_release(hits)) free(hits);
}
void mtsv_free(void* p) { free(p); }

void* mtsv_host_alloc(size_t bytes) {
    void* p = mtsv::host_pinned_alloc(bytes);
    if (!p) g_err = "device: page-locked allocation of " + std::to_string(bytes) + " bytes failed (no HIP device?)";
    return p;
}
void mtsv_host_free(void* p) { mtsv::host_pinned_free(p); }
int mtsv_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return fail_arg("null argument");
    if (!mtsv::host_pinned_register(p, bytes)) {
        g_err = "device: the memory could not be page-locked (no HIP device, or the range is already registered)";
        return MTSV_E_DEVICE;
    }
    return MTSV_OK;
}
int mtsv_host_unregister(void* p) {
    if (!p) return fail_arg("null argument");
    if (!mtsv::host_pinned_unregister(p)) {
        g_err = "device: not a registered range";
        return MTSV_E_DEVICE;
    }
    return MTSV_OK;
}

// ---- write_assignments (src/binner.rs:310-379) ------------------------------------------------
int mtsv_format_results(const mtsv_hit* hits, uint64_t n_hits, const char* ids, const uint64_t* id_off, uint64_t n_reads,
                        int long_format, char** out, uint64_t* out_len) {
    if ((!hits && n_hits) || !ids || !id_off || !out || !out_len) return fail_arg("null argument");
    GUARD({
        // the text grows in a malloc'd buffer that is handed to the caller as it is; numbers are written digit by
        // digit (snprintf per TaxID cost more than everything else on the result side of mtsv-binner)
        struct Text {
            char* p = nullptr;
            uint64_t n = 0, cap = 0;
            ~Text() { free(p); }
            void room(uint64_t more) {
                if (n + more <= cap) return;
                const uint64_t ncap = std::max<uint64_t>(n + more, cap + cap / 2 + (1 << 16));
                char* q = (char*)realloc(p, ncap);
                if (!q) throw std::bad_alloc();
                p = q;
                cap = ncap;
            }
            void put(uint64_t v) {  // decimal, as %u / %llu
                char tmp[20];
                int k = 0;
                do {
                    tmp[k++] = (char)('0' + v % 10);
                    v /= 10;
                } while (v);
                while (k) p[n++] = tmp[--k];
            }
        } buf;
        struct Item {
            uint32_t tax, gi;
            uint64_t off;
            uint32_t edit;
        };
        std::vector<Item> items;
        uint64_t i = 0;
        while (i < n_hits) {
            uint64_t r = hits[i].read;
            if (r >= n_reads) throw std::runtime_error("arg: hit refers to a read outside the batch");
            uint64_t j = i;
            items.clear();
            for (; j < n_hits && hits[j].read == r; j++) {
                const mtsv_hit& h = hits[j];
                bool found = false;
                for (auto& it : items) {
                    bool same = long_format ? (it.tax == h.tax_id && it.gi == h.gi && it.off == h.offset) : it.tax == h.tax_id;
                    if (same) {
                        if (h.edit < it.edit) it.edit = h.edit;  // smallest edit per key (binner.rs:326-331,358-361)
                        found = true;
                        break;
                    }
                }
                if (!found) items.push_back(Item{h.tax_id, h.gi, h.offset, h.edit});
            }
            if (j < n_hits && hits[j].read < r) throw std::runtime_error("arg: hits are not ordered by read");
            if (long_format)
                std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) {
                    if (a.tax != b.tax) return a.tax < b.tax;
                    if (a.gi != b.gi) return a.gi < b.gi;
                    if (a.off != b.off) return a.off < b.off;
                    return a.edit < b.edit;
                });
            else
                std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) {
                    if (a.tax != b.tax) return a.tax < b.tax;
                    return a.edit < b.edit;
                });
            const uint64_t id_len = strnlen(ids + id_off[r], id_off[r + 1] - id_off[r]);
            buf.room(id_len + 2 + items.size() * 56 + 1);  // "tax-gi-offset=edit," is at most 10 + 1 + 10 + 1 + 20 + 1 + 10 + 1 characters
            memcpy(buf.p + buf.n, ids + id_off[r], id_len);
            buf.n += id_len;
            buf.p[buf.n++] = ':';
            bool first = true;
            for (auto& it : items) {
                if (!first) buf.p[buf.n++] = ',';
                first = false;
                buf.put(it.tax);
                if (long_format) {  // "%u-%u-%llu=%u"
                    buf.p[buf.n++] = '-';
                    buf.put(it.gi);
                    buf.p[buf.n++] = '-';
                    buf.put(it.off);
                }
                buf.p[buf.n++] = '=';
                buf.put(it.edit);
            }
            buf.p[buf.n++] = '\n';
            i = j;
        }
        buf.room(1);
        buf.p[buf.n] = 0;
        *out = buf.p;
        *out_len = buf.n;
        buf.p = nullptr;  // the caller's now (mtsv_free)
    })
}

// ---- synthetic workloads -------------------------------------------------------

int mtsv_synth_index(uint64_t seed, uint32_t n_taxa, uint32_t gis_per_taxon, uint64_t seq_len, uint32_t occ_k,
                     uint64_t sa_s, int n_threads, mtsv_index** out) {
    if (!out || !n_taxa || !gis_per_taxon || seq_len < 64) return fail_arg("bad synthetic database shape");
    GUARD({
        const uint64_t nseq = (uint64_t)n_taxa * gis_per_taxon;
        std::vector<std::string> seqs(nseq);
        static const char ACGT[4] = {'A', 'C', 'G', 'T'};
        par_for(nseq, n_threads, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; i++) {
                SplitMix64 rng(mix(seed, i));
                std::string& s = seqs[i];
                s.resize(seq_len);
                for (uint64_t p = 0; p < seq_len;) {
                    uint64_t w = rng.next();
                    for (int k = 0; k < 32 && p < seq_len; k++, p++, w >>= 2) s[p] = ACGT[w & 3];
                }
            }
        });
        // 5% of every GI: a 1%-diverged copy of a segment of a GI of another taxon
        SplitMix64 rng(mix(seed, 0xC0FFEE));
        const uint64_t seg = std::max<uint64_t>(1, seq_len / 20);
        if (n_taxa > 1)
            for (uint64_t i = 0; i < nseq; i++) {
                uint64_t my_tax = i / gis_per_taxon;
                uint64_t src_tax = rng.below(n_taxa - 1);
                if (src_tax >= my_tax) src_tax++;
                uint64_t src = src_tax * gis_per_taxon + rng.below(gis_per_taxon);
                uint64_t sp = rng.below(seq_len - seg + 1), dp = rng.below(seq_len - seg + 1);
                for (uint64_t k = 0; k < seg; k++) {
                    char c = seqs[src][sp + k];
                    if (rng.chance(0.01)) c = ACGT[rng.below(4)];
                    seqs[i][dp + k] = c;
                }
            }
        // 0.1% of positions in runs of N of length 10..100
        for (uint64_t i = 0; i < nseq; i++) {
            uint64_t target = seq_len / 1000, placed = 0;
            while (placed < target) {
                uint64_t len = 10 + rng.below(91);
                if (len >= seq_len) break;
                uint64_t at = rng.below(seq_len - len);
                for (uint64_t k = 0; k < len; k++) seqs[i][at + k] = 'N';
                placed += len;
            }
        }
        // distinct TaxIDs / GIs
        std::vector<uint32_t> tax(nseq), gi(nseq);
        std::vector<uint32_t> taxids;
        while (taxids.size() < n_taxa) {
            uint32_t t = (uint32_t)rng.next();
            if (std::find(taxids.begin(), taxids.end(), t) == taxids.end()) taxids.push_back(t);
        }
        for (uint64_t i = 0; i < nseq; i++) {
            tax[i] = taxids[i / gis_per_taxon];
            gi[i] = (uint32_t)(mix(seed ^ 0x6D747376ull, i) | 1u) + (uint32_t)i;  // distinct enough; uniqueness not required
        }
        std::vector<SeqEntry> e(nseq);
        for (uint64_t i = 0; i < nseq; i++) e[i] = SeqEntry{tax[i], gi[i], (const uint8_t*)seqs[i].data(), seqs[i].size()};
        auto ix = std::make_unique<mtsv_index>();
        build_index(std::move(e), occ_k, sa_s, n_threads, ix->host, g_build_device);
        *out = ix.release();
    })
}

int mtsv_host_pack_threads(void) {
    if (getenv("MTSV_H2D_PLAIN")) return 0;
    const int n = mtsv::pack_threads();
    return n >= mtsv::kPackWorthwhile ? n : 0;
}

uint8_t mtsv_pack_bases(uint8_t* dst, const uint8_t* src, uint64_t first_offset, uint64_t n, uint8_t prev_code) {
    return mtsv::pack_chunk(dst, src, first_offset, n, prev_code);
}

int mtsv_synth_reads(const mtsv_index* ix, uint64_t seed, uint64_t n_reads, uint32_t read_len, uint8_t* bases,
                     uint64_t* read_off) {
    if (!ix || !bases || !read_off || read_len == 0) return fail_arg("null argument");
    const std::vector<uint8_t>& text = ix->host.text;
    if (text.size() < (uint64_t)read_len * 2 + 16) return fail_arg("index too small for this read length");
    GUARD({
        static const uint8_t ACGT[4] = {'A', 'C', 'G', 'T'};
        int threads = (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        const uint64_t span = text.size() - 1 - ((uint64_t)read_len + read_len / 4 + 8);
        for (uint64_t r = 0; r <= n_reads; r++) read_off[r] = r * read_len;
        par_for(n_reads, threads, [&](uint64_t lo, uint64_t hi) {
            std::vector<uint8_t> tmp(read_len);
            for (uint64_t r = lo; r < hi; r++) {
                SplitMix64 rng(mix(seed, r));
                uint8_t* dst = bases + r * read_len;
                if (rng.chance(0.10)) {  // no origin
                    for (uint32_t i = 0; i < read_len; i++) dst[i] = ACGT[rng.below(4)];
                    continue;
                }
                uint64_t p = rng.below(span);
                uint32_t i = 0;
                while (i < read_len) {
                    if (rng.chance(0.001)) {  // deletion from the read
                        p++;
                        continue;
                    }
                    if (rng.chance(0.001)) {  // insertion
                        tmp[i++] = ACGT[rng.below(4)];
                        continue;
                    }
                    uint8_t c = text[p++];
                    if (rng.chance(0.01)) c = ACGT[rng.below(4)];
                    if (rng.chance(0.002)) c = 'N';
                    tmp[i++] = c;
                }
                if (rng.chance(0.5)) {
                    for (uint32_t k = 0; k < read_len; k++) {
                        uint8_t c = tmp[read_len - 1 - k];
                        dst[k] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
                    }
                } else {
                    memcpy(dst, tmp.data(), read_len);
                }
            }
        });
    })
}

}  // extern "C"

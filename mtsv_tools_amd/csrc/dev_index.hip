// dev_index.hip -- re-pack the file's FM-index into the HBM layout (dev_layout.hpp), upload it,
// and build the two HBM-only acceleration structures the 288 GB of an MI355X make affordable:
//   * the full suffix array, reconstructed from the file's row-sampled SA by LF-walking on the
//     GPU, so that SampledSuffixArray::get (call site src/index.rs:347; ~s dependent rank queries
//     per seed hit) becomes one 4-byte gather;
//   * the SA interval of every ACGT k-mer, so that the first k of the seed_size backward-search
//     steps of FMIndex::backward_search (call site src/index.rs:305) become one 8-byte gather.
// Both hold exactly the values the reference's primitives would compute; results are unchanged.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include "dev_index.hpp"

namespace mtsv {

void throw_hip(hipError_t e, const char* what, const char* file, int line) {
    throw std::runtime_error(std::string("device: ") + hipGetErrorString(e) + " at " + what + " (" + file + ":" +
                             std::to_string(line) + ")");
}

DeviceIndex::~DeviceIndex() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    (void)hipFree(d_blocks);
    (void)hipFree(d_sa_sample);
    (void)hipFree(d_sa_full);
    (void)hipFree(d_text);
    (void)hipFree(d_bin_end);
    (void)hipFree(d_bins);
    (void)hipFree(d_bin_lut);
    (void)hipFree(d_kmer);
}

namespace {

inline uint32_t sym_code(uint8_t c) {
    switch (c) {
    case 'A': return kCodeA;
    case 'C': return kCodeC;
    case 'G': return kCodeG;
    case 'T': return kCodeT;
    case 'N': return kCodeN;
    case '$': return kCodeSentinel;
    default: return 7;
    }
}

// ---- acceleration-structure kernels -------------------------------------------------------

// One lane per sampled row j*s: walk LF (towards smaller text positions) writing SA values into
// every non-sampled row met, until the next sampled row or the sentinel row.  LF is one cycle over
// all rows, so every row is written exactly once.
__global__ void k_expand_sa(DevIndexView ix, uint32_t n_samples, uint32_t* __restrict__ sa_full) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_samples) return;
    uint32_t row = j * ix.sa_s;
    uint32_t v = ix.sa_sample[j];
    sa_full[row] = v;
    for (uint32_t guard = 0; guard <= ix.n; guard++) {
        uint32_t blk = row >> kBlockShift, off = row & (kBlockRows - 1);
        LoadedBlock b = load_block(ix.blocks, blk);
        uint32_t c = block_code(b, off);
        if (c == kCodeSentinel) break;  // this row is the suffix at text position 0
        row = less_of(ix, c) + block_rank(b, c, blk, off, ix.sentinel_row);
        v -= 1;
        if (row % ix.sa_s == 0) break;  // owned by another lane
        sa_full[row] = v;
    }
}

// level 1: intervals of the four single symbols
__global__ void k_kmer_level1(DevIndexView ix, uint2* __restrict__ tab) {
    uint32_t a = threadIdx.x;
    if (a >= 4) return;
    uint32_t lo = less_of(ix, a) + dev_rank(ix, a, 0);
    uint32_t hi = less_of(ix, a) + dev_rank(ix, a, ix.n);
    tab[a] = make_uint2(lo, hi);
}

// level j from level j-1: interval(aP) = LF-step(a, interval(P)); index(aP) = a*4^(j-1)+index(P)
__global__ void k_kmer_level(DevIndexView ix, const uint2* __restrict__ prev, uint2* __restrict__ cur,
                             uint32_t prev_shift) {
    // grid-stride: the last level of a 16-mer table has 2^32 entries (of a 17-mer table 2^34), more than one launch may have threads;
    // the level below has 4^(j-1) = 2^prev_shift entries
    const uint64_t total = 4ull << prev_shift, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        uint32_t a = (uint32_t)(t >> prev_shift);
        uint64_t p = t & ((1ull << prev_shift) - 1);
        uint2 iv = prev[p];
        uint32_t lo = 0, hi = 0;
        if (iv.x < iv.y) {
            uint32_t b0 = iv.x >> kBlockShift, b1 = iv.y >> kBlockShift;
            LoadedBlock B0 = load_block(ix.blocks, b0);
            lo = less_of(ix, a) + block_rank(B0, a, b0, iv.x & (kBlockRows - 1), ix.sentinel_row);
            if (b1 == b0) {
                hi = less_of(ix, a) + block_rank(B0, a, b1, iv.y & (kBlockRows - 1), ix.sentinel_row);
            } else {
                LoadedBlock B1 = load_block(ix.blocks, b1);
                hi = less_of(ix, a) + block_rank(B1, a, b1, iv.y & (kBlockRows - 1), ix.sentinel_row);
            }
        }
        cur[t] = make_uint2(lo, hi);
    }
}

template <class F>
void parallel_ranges(uint64_t n, int threads, F fn) {
    std::vector<std::thread> pool;
    uint64_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        uint64_t lo = std::min(n, (uint64_t)t * chunk), hi = std::min(n, lo + chunk);
        pool.emplace_back(fn, lo, hi, t);
    }
    for (auto& th : pool) th.join();
}

}  // namespace

std::unique_ptr<DeviceIndex> upload_index(const HostIndex& hx, int device, uint32_t flags) {
    const uint64_t n64 = hx.n();
    if (n64 >= 0xFFFFFF00ull)
        throw std::runtime_error("limit: index of 2^32 symbols or more does not fit the u32 device layout; chunk the database (mtsv-chunk)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw std::runtime_error("device: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) throw std::runtime_error("device: invalid device ordinal " + std::to_string(device));
    HIP_CHECK(hipSetDevice(device));

    const uint32_t n = (uint32_t)n64;
    const uint32_t n_blocks = (n >> kBlockShift) + 1;
    int threads = (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (n_blocks < 4096) threads = 1;

    // ---- pack rank blocks (two passes: per-range symbol counts, then blocks) ----------------
    std::vector<RankBlock> blocks(n_blocks);
    std::vector<std::array<uint64_t, 6>> range_cnt(threads);
    // ranges are block-aligned
    uint64_t blocks_per = ((uint64_t)n_blocks + threads - 1) / threads;
    parallel_ranges(n_blocks, threads, [&](uint64_t, uint64_t, int t) {
        uint64_t b0 = std::min<uint64_t>(n_blocks, (uint64_t)t * blocks_per), b1 = std::min<uint64_t>(n_blocks, b0 + blocks_per);
        std::array<uint64_t, 6> c{};
        uint64_t lo = b0 << kBlockShift, hi = std::min<uint64_t>(n, b1 << kBlockShift);
        for (uint64_t i = lo; i < hi; i++) {
            uint32_t code = sym_code(hx.bwt[i]);
            if (code < 6) c[code]++;
        }
        range_cnt[t] = c;
    });
    std::vector<std::array<uint64_t, 6>> range_base(threads);
    {
        std::array<uint64_t, 6> acc{};
        for (int t = 0; t < threads; t++) {
            range_base[t] = acc;
            for (int a = 0; a < 6; a++) acc[a] += range_cnt[t][a];
        }
    }
    std::vector<std::string> errs(threads);
    std::vector<uint64_t> sentinel_rows(threads, UINT64_MAX);
    static const uint8_t syms[6] = {'A', 'C', 'G', 'T', 'N', '$'};
    parallel_ranges(n_blocks, threads, [&](uint64_t, uint64_t, int t) {
        uint64_t b0 = std::min<uint64_t>(n_blocks, (uint64_t)t * blocks_per), b1 = std::min<uint64_t>(n_blocks, b0 + blocks_per);
        std::array<uint64_t, 6> c = range_base[t];
        for (uint64_t b = b0; b < b1; b++) {
            RankBlock rb;
            for (int a = 0; a < 4; a++) rb.cnt[a] = (uint32_t)c[a];
            uint64_t p0[2] = {0, 0}, p1[2] = {0, 0}, p2[2] = {0, 0};
            for (uint32_t o = 0; o < kBlockRows; o++) {
                uint64_t i = (b << kBlockShift) + o;
                uint32_t code = 7;
                if (i < n) {
                    code = sym_code(hx.bwt[i]);
                    if (code >= 6) {
                        errs[t] = "format: bwt holds a symbol outside ACGTN$";
                        return;
                    }
                    c[code]++;
                    if (code == kCodeSentinel) sentinel_rows[t] = i;
                    // cross-check the file's Occ table: checkpoint j = inclusive counts in bwt[0..=j*k]
                    if (i % hx.k == 0) {
                        uint64_t j = i / hx.k;
                        for (int a = 0; a < 6; a++)
                            if (hx.occ[syms[a]][j] != c[a]) {
                                errs[t] = "format: Occ checkpoint " + std::to_string(j) + " disagrees with the bwt";
                                return;
                            }
                    }
                }
                p0[o >> 6] |= (uint64_t)(code & 1) << (o & 63);
                p1[o >> 6] |= (uint64_t)((code >> 1) & 1) << (o & 63);
                p2[o >> 6] |= (uint64_t)((code >> 2) & 1) << (o & 63);
            }
            rb.p0[0] = p0[0]; rb.p0[1] = p0[1];
            rb.p1[0] = p1[0]; rb.p1[1] = p1[1];
            rb.p2[0] = p2[0]; rb.p2[1] = p2[1];
            blocks[b] = rb;
        }
    });
    for (auto& e : errs)
        if (!e.empty()) throw std::runtime_error(e);
    uint64_t sentinel_row = UINT64_MAX;
    for (uint64_t s : sentinel_rows)
        if (s != UINT64_MAX) sentinel_row = s;
    if (sentinel_row == UINT64_MAX) throw std::runtime_error("format: bwt holds no sentinel");

    // ---- text codes, SA samples, bins ---------------------------------------------------------
    std::vector<uint8_t> codes(((uint64_t)n + 15) / 16 * 16 + 32, (uint8_t)7);  // padded for load16(): two aligned 16-byte loads from any position
    parallel_ranges(n, threads, [&](uint64_t lo, uint64_t hi, int) {
        for (uint64_t i = lo; i < hi; i++) codes[i] = (uint8_t)sym_code(hx.text[i]);
    });
    std::vector<uint32_t> samp(hx.sample.size());
    for (size_t i = 0; i < samp.size(); i++) samp[i] = (uint32_t)hx.sample[i];
    std::vector<DevBin> bins(hx.bins.size());
    std::vector<uint32_t> bin_end(hx.bins.size());
    for (size_t i = 0; i < bins.size(); i++) {
        bins[i] = DevBin{(uint32_t)hx.bins[i].start, (uint32_t)hx.bins[i].end, hx.bins[i].tax_id, hx.bins[i].gi};
        bin_end[i] = (uint32_t)hx.bins[i].end;
    }

    // coarse bin lookup: bucket k covers text positions [k << shift, (k+1) << shift)
    uint32_t lut_shift = 0;
    while (((uint64_t)n >> lut_shift) > 65536) lut_shift++;
    std::vector<uint32_t> bin_lut(((uint64_t)n >> lut_shift) + 2);
    {
        uint32_t b = 0;
        for (size_t k = 0; k < bin_lut.size(); k++) {
            uint64_t p = (uint64_t)k << lut_shift;
            while (b + 1 < bins.size() && bin_end[b] <= p) b++;
            bin_lut[k] = b;
        }
    }

    auto di = std::make_unique<DeviceIndex>();
    di->device = device;
    di->flags = flags;
    auto up = [&](auto** dptr, const void* src, uint64_t bytes) {
        HIP_CHECK(hipMalloc((void**)dptr, bytes ? bytes : 16));
        if (bytes) HIP_CHECK(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
        di->bytes += bytes;
    };
    up(&di->d_blocks, blocks.data(), (uint64_t)n_blocks * sizeof(RankBlock));
    up(&di->d_sa_sample, samp.data(), samp.size() * 4);
    up(&di->d_text, codes.data(), codes.size());
    up(&di->d_bin_end, bin_end.data(), bin_end.size() * 4);
    up(&di->d_bins, bins.data(), bins.size() * sizeof(DevBin));
    up(&di->d_bin_lut, bin_lut.data(), bin_lut.size() * 4);

    DevIndexView& v = di->view;
    v.blocks = di->d_blocks;
    v.n = n;
    v.n_blocks = n_blocks;
    v.C[0] = (uint32_t)hx.less['A'];
    v.C[1] = (uint32_t)hx.less['C'];
    v.C[2] = (uint32_t)hx.less['G'];
    v.C[3] = (uint32_t)hx.less['T'];
    v.C[4] = (uint32_t)hx.less['N'];
    v.sentinel_row = (uint32_t)sentinel_row;
    v.sa_sample = di->d_sa_sample;
    v.sa_s = (uint32_t)hx.s;
    v.sa_pow2_shift = 0xffffffffu;
    if (hx.s > 0xFFFFFFFFull) throw std::runtime_error("limit: suffix sampling interval does not fit u32");
    for (uint32_t sh = 0; sh < 32; sh++)
        if ((1ull << sh) == hx.s) v.sa_pow2_shift = sh;
    v.sa_full = nullptr;
    v.text = di->d_text;
    v.bin_end = di->d_bin_end;
    v.bins = di->d_bins;
    v.n_bins = (uint32_t)bins.size();
    v.bin_lut = di->d_bin_lut;
    v.bin_lut_shift = lut_shift;
    v.kmer_tab = nullptr;
    v.kmer_k = 0;

    // ---- acceleration structures ---------------------------------------------------------------
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, 0));
    if (!(flags & 1u /* MTSV_DEV_SAMPLED_SA_ONLY */)) {
        HIP_CHECK(hipMalloc((void**)&di->d_sa_full, (uint64_t)n * 4));
        di->bytes += (uint64_t)n * 4;
        uint32_t ns = (uint32_t)samp.size();
        hipLaunchKernelGGL(k_expand_sa, dim3((ns + 255) / 256), dim3(256), 0, 0, v, ns, di->d_sa_full);
        HIP_CHECK(hipGetLastError());
        v.sa_full = di->d_sa_full;
    }
    if (!(flags & 2u /* MTSV_DEV_NO_KMER_TABLE */)) {
        // table size ~ index size: 4^k entries of 8 B for 4^k <= 2n up to k = 16 (32 GiB: every symbol of the
        // table saves a dependent pair of rank-block gathers per seed), and only if it fits the free HBM twice over.
        // Indexes of 2^31 symbols or more get a 17th symbol (128 GiB) when the device has the room: that is what
        // 288 GB of HBM are for -- the 10 GB index then costs 153 GB and a default seed of 18 symbols one rank step.
        uint32_t k = 1;
        while (k < 16 && (1ull << (2 * (k + 1))) <= 2 * (uint64_t)n) k++;
        if (k == 16 && n >= (1u << 31)) k = 17;
        if (const char* e = getenv("MTSV_KMER_K")) {
            int kk = atoi(e);
            if (kk >= 1 && kk <= 17) k = (uint32_t)kk;
        }
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        // 17: the table, the level below it while building, and 48 GiB left for batch workspaces
        if (k == 17 && (10ull << 34) + (48ull << 30) > (uint64_t)free_b) k = 16;
        while (k > 1 && k <= 16 && (10ull << (2 * k)) * 2 > (uint64_t)free_b) k--;  // table + the previous level while building
        uint64_t entries = 1ull << (2 * k);
        uint2 *ta = nullptr, *tb = nullptr;
        if (k == 17) {  // 128 + 32 GiB in two pieces: free memory by the count is not always free memory in one piece
            if (hipMalloc((void**)&ta, entries * 8) != hipSuccess || hipMalloc((void**)&tb, entries / 4 * 8) != hipSuccess) {
                (void)hipGetLastError();
                if (ta) (void)hipFree(ta);
                ta = tb = nullptr;
                k = 16;
                entries = 1ull << 32;
            }
        }
        if (!ta) {
            HIP_CHECK(hipMalloc((void**)&ta, entries * 8));
            HIP_CHECK(hipMalloc((void**)&tb, std::max<uint64_t>(entries / 4, 4) * 8));
        }
        // ping-pong so the last level lands in `ta`
        uint2* cur = (k % 2 == 1) ? ta : tb;
        uint2* oth = (k % 2 == 1) ? tb : ta;
        hipLaunchKernelGGL(k_kmer_level1, dim3(1), dim3(64), 0, 0, v, cur);
        for (uint32_t lvl = 2; lvl <= k; lvl++) {
            const uint64_t total = 1ull << (2 * lvl);
            hipLaunchKernelGGL(k_kmer_level, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 1u << 22)), dim3(256), 0, 0, v, cur, oth,
                               2 * (lvl - 1));
            std::swap(cur, oth);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipFree(tb));
        di->d_kmer = ta;  // cur == ta by construction
        di->bytes += entries * 8;
        v.kmer_tab = di->d_kmer;
        v.kmer_k = k;
    }
    HIP_CHECK(hipEventRecord(e1, 0));
    HIP_CHECK(hipEventSynchronize(e1));
    HIP_CHECK(hipEventElapsedTime(&di->accel_build_ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return di;
}

}  // namespace mtsv

// gpu_builder.hip -- suffix array, BWT and SA samples of MGIndex::new (src/index.rs:560-575) on
// the GPU: prefix doubling (Manber-Myers) over the whole text in HBM.  Round 0 sorts the suffixes
// by their first 21 symbols (3-bit ranks of "$ACGNT" packed into one u64), every later round sorts
// by (rank[i], rank[i+h]) and doubles h, until every suffix has a unique rank.  Sorting uses
// rocPRIM's device radix sort (a library primitive; this is one-time preprocessing, not the hot
// path); keys, ranks, BWT and sampling are hand-written kernels.  ~32 bytes of HBM per symbol:
// 88 GB for the 10 GB MG-index -- what a 288 GB device is for.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdexcept>
#include <string>
#include <vector>

#include "dev_index.hpp"
#include "mgindex.hpp"

namespace mtsv {

namespace {

__device__ inline uint32_t sym_rank_dev(uint8_t c) {  // $ < A < C < G < N < T
    return c == '$' ? 0u : c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 3u : c == 'N' ? 4u : 5u;
}

__global__ void k_key0(const uint8_t* __restrict__ text, uint32_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = 0;
    for (uint32_t j = 0; j < 21; j++) {
        uint64_t p = i + j;
        k = (k << 3) | (p < n ? sym_rank_dev(text[p]) : 0u);
    }
    keys[i] = k;
    vals[i] = (uint32_t)i;
}

// head[j] = j if the sorted key at j starts a new group, else 0 (inclusive max-scan gives the group start)
__global__ void k_heads(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ head,
                        unsigned long long* __restrict__ n_groups) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool is_head = false;
    if (j < n) {
        is_head = j == 0 || keys[j] != keys[j - 1];
        head[j] = is_head ? (uint32_t)j : 0u;
    }
    unsigned long long m = __ballot(is_head);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_groups, (unsigned long long)__popcll(m));
}

__global__ void k_scatter_rank(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ group_start, uint32_t n,
                               uint32_t* __restrict__ rank) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) rank[sa[j]] = group_start[j];
}

// key of suffix i for the next round: (rank[i], rank[i+h] + 1), 0 in the low half past the end
__global__ void k_key_h(const uint32_t* __restrict__ rank, uint32_t n, uint32_t h, uint64_t* __restrict__ keys,
                        uint32_t* __restrict__ vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t lo = i + h < n ? (uint64_t)rank[i + h] + 1 : 0;
    keys[i] = ((uint64_t)rank[i] << 32) | lo;
    vals[i] = (uint32_t)i;
}

__global__ void k_bwt(const uint8_t* __restrict__ text, const uint32_t* __restrict__ sa, uint32_t n, uint8_t* __restrict__ bwt,
                      uint32_t* __restrict__ sentinel_row) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t p = sa[j];
    bwt[j] = p ? text[p - 1] : text[n - 1];  // index.rs:567
    if (p == 0) *sentinel_row = (uint32_t)j;
}

__global__ void k_sample(const uint32_t* __restrict__ sa, uint32_t n, uint32_t s, uint32_t* __restrict__ sample) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t row = j * s;
    if (row < n) sample[j] = sa[row];
}

struct MaxOp {
    __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

inline uint32_t grid_for(uint64_t n) { return (uint32_t)((n + 255) / 256); }

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(uint64_t bytes) { HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16)); }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    template <class T>
    T* as() const { return (T*)p; }
};

}  // namespace

// text: normalised "ACGTN...$", n symbols.  Fills bwt (n bytes), sample (ceil(n/s) entries) and the
// row whose BWT symbol is '$'.  Throws "device: ..." on any HIP failure.
void gpu_suffix_sort(const uint8_t* text, uint32_t n, uint64_t s, int device, std::vector<uint8_t>& bwt,
                     std::vector<uint64_t>& sample, uint64_t* sentinel_row) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        throw std::runtime_error("device: no HIP device " + std::to_string(device) + " for the index build");
    HIP_CHECK(hipSetDevice(device));
    DevBuf d_text(n), d_keys_a((uint64_t)n * 8), d_keys_b((uint64_t)n * 8), d_vals_a((uint64_t)n * 4), d_vals_b((uint64_t)n * 4),
        d_rank((uint64_t)n * 4), d_head((uint64_t)n * 4), d_cnt(16);
    HIP_CHECK(hipMemcpy(d_text.p, text, n, hipMemcpyHostToDevice));
    size_t tmp_sort = 0, tmp_scan = 0;
    rocprim::double_buffer<uint64_t> kb(d_keys_a.as<uint64_t>(), d_keys_b.as<uint64_t>());
    rocprim::double_buffer<uint32_t> vb(d_vals_a.as<uint32_t>(), d_vals_b.as<uint32_t>());
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_sort, kb, vb, (size_t)n, 0, 64));
    HIP_CHECK(rocprim::inclusive_scan(nullptr, tmp_scan, d_head.as<uint32_t>(), d_vals_b.as<uint32_t>(), (size_t)n, MaxOp()));
    DevBuf d_tmp(std::max(tmp_sort, tmp_scan));

    hipLaunchKernelGGL(k_key0, dim3(grid_for(n)), dim3(256), 0, 0, d_text.as<uint8_t>(), n, kb.current(), vb.current());
    uint32_t h = 21;
    for (int round = 0;; round++) {
        size_t t1 = tmp_sort;
        // round 0 keys use 63 bits; later rounds 64
        HIP_CHECK(rocprim::radix_sort_pairs(d_tmp.p, t1, kb, vb, (size_t)n, 0, round == 0 ? 63 : 64));
        HIP_CHECK(hipMemset(d_cnt.p, 0, 8));
        hipLaunchKernelGGL(k_heads, dim3(grid_for(n)), dim3(256), 0, 0, kb.current(), n, d_head.as<uint32_t>(),
                           d_cnt.as<unsigned long long>());
        unsigned long long groups = 0;
        HIP_CHECK(hipMemcpy(&groups, d_cnt.p, 8, hipMemcpyDeviceToHost));
        if (groups == n) break;  // every suffix distinguished: vb.current() is the suffix array
        if (round > 40) throw std::runtime_error("device: suffix sort did not converge");
        size_t t2 = tmp_scan;
        uint32_t* group_start = vb.alternate();  // free after the sort
        HIP_CHECK(rocprim::inclusive_scan(d_tmp.p, t2, d_head.as<uint32_t>(), group_start, (size_t)n, MaxOp()));
        hipLaunchKernelGGL(k_scatter_rank, dim3(grid_for(n)), dim3(256), 0, 0, vb.current(), group_start, n, d_rank.as<uint32_t>());
        hipLaunchKernelGGL(k_key_h, dim3(grid_for(n)), dim3(256), 0, 0, d_rank.as<uint32_t>(), n, h, kb.current(), vb.current());
        h = h > 0x40000000u ? 0x80000000u : h * 2;
    }
    HIP_CHECK(hipGetLastError());
    const uint32_t* sa = vb.current();
    // BWT into the (now free) alternate key buffer, samples into the head buffer
    uint8_t* d_bwt = (uint8_t*)kb.alternate();
    uint32_t* d_sample = d_head.as<uint32_t>();
    HIP_CHECK(hipMemset(d_cnt.p, 0xff, 4));
    hipLaunchKernelGGL(k_bwt, dim3(grid_for(n)), dim3(256), 0, 0, d_text.as<uint8_t>(), sa, n, d_bwt, d_cnt.as<uint32_t>());
    const uint64_t ns = ((uint64_t)n + s - 1) / s;
    if (s <= 0xffffffffull)
        hipLaunchKernelGGL(k_sample, dim3(grid_for(ns)), dim3(256), 0, 0, sa, n, (uint32_t)s, d_sample);
    HIP_CHECK(hipGetLastError());
    bwt.resize(n);
    HIP_CHECK(hipMemcpy(bwt.data(), d_bwt, n, hipMemcpyDeviceToHost));
    std::vector<uint32_t> samp32(ns);
    if (s <= 0xffffffffull) {
        HIP_CHECK(hipMemcpy(samp32.data(), d_sample, ns * 4, hipMemcpyDeviceToHost));
    } else {
        HIP_CHECK(hipMemcpy(samp32.data(), sa, 4, hipMemcpyDeviceToHost));  // only row 0 is sampled
    }
    sample.assign(samp32.begin(), samp32.end());
    uint32_t srow = 0;
    HIP_CHECK(hipMemcpy(&srow, d_cnt.p, 4, hipMemcpyDeviceToHost));
    *sentinel_row = srow;
}

}  // namespace mtsv

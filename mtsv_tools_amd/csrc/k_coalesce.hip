// k_coalesce.hip -- coalesce_seed_sites, min_seeds, stable rank, same-TaxId chains (index.rs:358-369,435-487)
// (one of the three kernel files of the hot path; the stage map is in kernels.hpp / DESIGN.md section 3)
#include "kernels_common.hpp"

namespace mtsv {

namespace {

// ---------------------------------------------------------------------------------------------
// K3: coalesce
// ---------------------------------------------------------------------------------------------
__device__ inline uint64_t wave_bitonic_sort(uint64_t key) {
    const uint32_t lane = lane_id();
    for (uint32_t k = 2; k <= kWave; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            uint64_t other = __shfl_xor(key, j);
            bool up = (lane & k) == 0;
            bool lower = (lane & j) == 0;
            key = (lower == up) ? min(key, other) : max(key, other);
        }
    return key;
}

// first bin whose end > site (the forward-only cursor of index.rs:455-458 on sorted hits): a coarse
// table gives the first bin that can hold the site's bucket, then a short forward scan
__device__ inline uint32_t find_bin(const DevIndexView& ix, uint32_t site) {
    const uint32_t k = site >> ix.bin_lut_shift;
    uint32_t lo = ix.bin_lut[k], hi = ix.bin_lut[k + 1];  // the answer lies in [lo, hi]; usually lo == hi
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (ix.bin_end[mid] <= site)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// SeedHit::candidate_indices (index.rs:118-153); returns false for None
__device__ inline bool candidate_window(uint32_t site, uint32_t q, const DevBin& bin, uint32_t L, uint32_t ED,
                                        uint32_t* ws, uint32_t* we) {
    uint32_t start_offset = q + ED;
    uint32_t s = (start_offset > site || site - start_offset < bin.start) ? bin.start : site - start_offset;
    uint64_t e64 = (uint64_t)site + (L - q) + ED;
    uint32_t e = e64 > bin.end ? bin.end : (uint32_t)e64;
    *ws = s;
    *we = e;
    return !(s > e || e - s < L - ED);
}

struct StrandGeom {
    uint32_t L, ED, min_seeds;
};


// geo = k_thin's word for the strand: min_seeds (16 bits) | ED << 16 | hopeless flag
__device__ inline StrandGeom strand_geom(const uint32_t* read_off, uint32_t r, uint32_t geo) {
    StrandGeom g;
    g.L = read_off[r + 1] - read_off[r];
    g.ED = (geo >> 16) & 0x7fffu;
    g.min_seeds = geo & 0xffffu;
    return g;
}

// running state of the coalescing walk (index.rs:445-485)
struct Walk {
    bool have;
    uint32_t s, e, b, n;
};

__device__ inline uint64_t gload(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void gstore(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void wave_mem_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// ---------------------------------------------------------------------------------------------
// Strands with more than 64 seed hits (repeats, N-run flanks: up to seeds x max_hits).  One
// workgroup per strand; the three sorts (hits by (ref, q); candidates by (num_seeds desc, order);
// candidates by (TaxId, rank)) run as a block-wide bitonic network on keys held in LDS.  Segments
// that do not fit the LDS array fall back to one wavefront sorting in L2-resident scratch.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kHeavyKeys = 8192;  // 64 KiB of LDS: the workgroup-wide sorts of the largest strands
constexpr uint32_t kHeavyKeysSmall = 2048;  // 16 KiB: most heavy strands (a few seeds of several hundred hits each) fit, and
                                            // four times as many workgroups are resident per CU

template <bool BLK>
struct KeyMem {
    uint64_t* p;
    __device__ uint64_t ld(uint32_t i) const { return BLK ? p[i] : gload(p + i); }
    __device__ void st(uint32_t i, uint64_t v) const {
        if (BLK) p[i] = v;
        else gstore(p + i, v);
    }
    __device__ void sync() const {
        if (BLK) __syncthreads();
        else wave_mem_sync();
    }
    __device__ uint32_t tid() const { return BLK ? threadIdx.x : lane_id(); }
    __device__ uint32_t nthreads() const { return BLK ? blockDim.x : (uint32_t)kWave; }
};

// flip/disperse bitonic network (every comparator ascending, so virtual +inf padding above n is inert)
template <bool BLK>
__device__ void big_sort(KeyMem<BLK> km, uint32_t n) {
    const uint32_t tid = km.tid(), nt = km.nthreads();
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    const uint32_t half = np2 >> 1;
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        for (uint32_t i = tid; i < half; i += nt) {
            uint32_t hk = k >> 1, blk = i / hk, pos = i % hk;
            uint32_t a = blk * k + pos, b = blk * k + k - 1 - pos;
            if (b < n) {
                uint64_t x = km.ld(a), y = km.ld(b);
                if (x > y) {
                    km.st(a, y);
                    km.st(b, x);
                }
            }
        }
        km.sync();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < half; i += nt) {
                uint32_t blk = i / j, pos = i % j;
                uint32_t a = blk * 2 * j + pos, b = a + j;
                if (b < n) {
                    uint64_t x = km.ld(a), y = km.ld(b);
                    if (x > y) {
                        km.st(a, y);
                        km.st(b, x);
                    }
                }
            }
            km.sync();
        }
    }
}

struct HeavyArgs {
    const uint32_t* read_off;
    uint32_t r0;
    uint32_t maxc;
    const uint32_t* strand_off;
    const uint32_t* strand_nseeds;
    const uint32_t* hit_ref;
    const uint32_t* hit_q;
    uint64_t* hit_key;
    uint64_t* cand_tmp;
    uint4* cand;
    uint32_t* cand_next;
    uint32_t* cand_status;
    uint32_t* strand_ncand;
    uint32_t* worklist;
    uint32_t* wl_count;
    unsigned long long* n_cand_total;
};

// BLK: called by every thread of the workgroup (contains barriers); !BLK: by one wavefront
template <bool BLK>
__device__ void coalesce_big(const DevIndexView& ix, const HeavyArgs& a, uint32_t rs, KeyMem<BLK> km, uint32_t* sh_nc) {
    const uint32_t lane = lane_id();
    const uint32_t tid = km.tid(), nt = km.nthreads();
    const bool walker = !BLK || threadIdx.x < kWave;  // the sequential walk runs on one wavefront
    const uint32_t o = a.strand_off[rs];
    const uint32_t nh = a.strand_off[rs + 1] - o;
    const uint32_t ns_raw = a.strand_nseeds[rs];
    const bool hopeless = (ns_raw & kHopeless) != 0;  // see k_coalesce
    const StrandGeom g = strand_geom(a.read_off, a.r0 + (rs >> 1), ns_raw);
    for (uint32_t i = tid; i < nh; i += nt) km.st(i, ((uint64_t)a.hit_ref[o + i] << 32) | a.hit_q[o + i]);
    km.sync();
    big_sort(km, nh);  // seed_hits.sort(), index.rs:443
    uint32_t nc = 0;
    uint64_t* ct = a.cand_tmp + 2ull * o;
    if (walker) {
        Walk w{false, 0, 0, 0, 0};
        for (uint32_t base = 0; base < nh; base += kWave) {
            uint32_t cntv = min((uint32_t)kWave, nh - base);
            uint32_t b = 0, ws = 0, we = 0;
            bool ok = false;
            if (lane < cntv) {
                uint64_t key = km.ld(base + lane);
                uint32_t site = (uint32_t)(key >> 32), q = (uint32_t)key;
                b = min(find_bin(ix, site), ix.n_bins - 1);
                DevBin bin = ix.bins[b];
                ok = candidate_window(site, q, bin, g.L, g.ED, &ws, &we);
            }
            for (uint32_t i = 0; i < cntv; i++) {
                uint32_t wsi = __builtin_amdgcn_readlane(ws, i), wei = __builtin_amdgcn_readlane(we, i);
                uint32_t bi = __builtin_amdgcn_readlane(b, i);
                bool oki = __builtin_amdgcn_readlane((uint32_t)ok, i) != 0;
                bool merge = w.have && oki && bi == w.b && ((w.s <= wsi && wsi < w.e) || (w.s < wei && wei <= w.e));
                if (merge) {  // add_seed_hit, index.rs:216-229
                    w.s = min(w.s, wsi);
                    w.e = max(w.e, wei);
                    w.n++;
                } else {
                    if (w.have && w.n >= g.min_seeds) {  // index.rs:467-469
                        if (lane == 0) {
                            gstore(ct + 2ull * nc, ((uint64_t)w.e << 32) | w.s);
                            gstore(ct + 2ull * nc + 1, ((uint64_t)w.n << 32) | w.b);
                        }
                        nc++;
                    }
                    w.have = oki;
                    w.s = wsi; w.e = wei; w.b = bi; w.n = 1;
                }
            }
        }
        if (w.have && w.n >= g.min_seeds) {  // index.rs:481-485
            if (lane == 0) {
                gstore(ct + 2ull * nc, ((uint64_t)w.e << 32) | w.s);
                gstore(ct + 2ull * nc + 1, ((uint64_t)w.n << 32) | w.b);
            }
            nc++;
        }
        if (BLK && threadIdx.x == 0) *sh_nc = nc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    km.sync();
    if (BLK) nc = *sh_nc;
    if (hopeless && a.maxc >= nc) {
        // every candidate is prefiltered by the reference and none can pass: account the work, skip the ranking
        unsigned long long wsum = 0;
        for (uint32_t i = tid; i < nc; i += nt) {
            const uint64_t se = gload(ct + 2ull * i);
            wsum += (uint32_t)(se >> 32) - (uint32_t)se;
        }
        for (int d = 32; d > 0; d >>= 1) wsum += __shfl_down(wsum, d);
        if (lane == 0 && wsum) atomicAdd(a.n_cand_total + 2, wsum);
        if (tid == 0) {
            a.strand_ncand[rs] = 0;
            if (nc) {
                atomicAdd(a.n_cand_total, (unsigned long long)nc);
                atomicAdd(a.n_cand_total + 1, (unsigned long long)nc);
            }
        }
        km.sync();  // the key array is free for the next strand
        return;
    }
    // rank: sort (num_seeds descending, walk order ascending) -- the stable sort of index.rs:369
    for (uint32_t i = tid; i < nc; i += nt) {
        uint32_t n_i = (uint32_t)(gload(ct + 2ull * i + 1) >> 32);
        km.st(i, ((uint64_t)(0xffffffffu - n_i) << 32) | i);
    }
    km.sync();
    big_sort(km, nc);
    for (uint32_t base = 0; base < nc; base += nt) {  // every thread runs the same trip count (barrier below)
        uint32_t i = base + tid;
        uint64_t taxkey = 0;
        if (i < nc) {
            uint32_t src = (uint32_t)km.ld(i);
            uint64_t se = gload(ct + 2ull * src), bn = gload(ct + 2ull * src + 1);
            a.cand[o + i] = make_uint4((uint32_t)se, (uint32_t)(se >> 32), (uint32_t)bn, rs);
            a.cand_status[o + i] = 0;
            taxkey = ((uint64_t)ix.bins[(uint32_t)bn].tax_id << 32) | i;  // (TaxId, rank)
            if (hopeless && i < a.maxc) {  // max_candidates < nc: only the first ranks are prefiltered
                atomicAdd(a.n_cand_total + 1, 1ull);
                atomicAdd(a.n_cand_total + 2, (unsigned long long)((uint32_t)(se >> 32) - (uint32_t)se));
            }
        }
        km.sync();  // all reads of this tile's rank keys are done before they are overwritten
        if (i < nc) km.st(i, taxkey);
    }
    km.sync();
    big_sort(km, nc);
    for (uint32_t base = (BLK ? (threadIdx.x / kWave) * kWave : 0); base < nc; base += nt) {
        uint32_t p = base + lane;
        bool emit = false;
        uint32_t rk = 0;
        if (p < nc) {
            uint64_t k0 = km.ld(p);
            uint64_t kn = p + 1 < nc ? km.ld(p + 1) : ~0ull;
            uint64_t kp = p > 0 ? km.ld(p - 1) : ~0ull;
            rk = (uint32_t)k0;
            a.cand_next[o + rk] = (kn >> 32) == (k0 >> 32) ? (uint32_t)kn : 0xffffffffu;
            emit = (p == 0 || (kp >> 32) != (k0 >> 32)) && rk < a.maxc && !hopeless;
        }
        unsigned long long em = __ballot(emit);
        if (em) {
            uint32_t b2 = 0;
            if (lane == 0) b2 = atomicAdd(a.wl_count, (uint32_t)__popcll(em));
            b2 = __builtin_amdgcn_readfirstlane(b2);
            if (emit) a.worklist[b2 + __popcll(em & ((1ull << lane) - 1))] = o + rk;
        }
    }
    if (tid == 0) {
        a.strand_ncand[rs] = hopeless ? 0 : nc;
        if (nc) atomicAdd(a.n_cand_total, (unsigned long long)nc);
    }
    km.sync();  // the key array is free for the next strand
}

// The heavy strands sit at the END of the strand list (entry list_len - 1 - h), the 17..64-hit ones at its front.
// KEYS = kHeavyKeysSmall takes the strands of up to that many hits, KEYS = kHeavyKeys all longer ones.
template <uint32_t KEYS>
__global__ __launch_bounds__(256) void k_coalesce_heavy(DevIndexView ix, HeavyArgs a, const uint32_t* __restrict__ heavy_list,
                                                        uint32_t list_len, const uint32_t* __restrict__ heavy_count) {
    __shared__ uint64_t lk[KEYS];
    __shared__ uint32_t sh_nc;
    const uint32_t n_heavy = *heavy_count;
    for (uint32_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        const uint32_t rs = heavy_list[list_len - 1 - h];
        const uint32_t nh = a.strand_off[rs + 1] - a.strand_off[rs];
        if (KEYS == kHeavyKeysSmall ? nh > kHeavyKeysSmall : nh <= kHeavyKeysSmall) continue;  // the other instantiation's
        if (nh <= KEYS) {
            coalesce_big<true>(ix, a, rs, KeyMem<true>{lk}, &sh_nc);
        } else {
            if (threadIdx.x < kWave) coalesce_big<false>(ix, a, rs, KeyMem<false>{a.hit_key + a.strand_off[rs]}, nullptr);
            __syncthreads();
        }
    }
}


// Strands with at most 16 seed hits, four per wavefront (one 16-lane group each); longer ones go to the list
// k_coalesce_mid (17..64 hits, a wavefront each) and k_coalesce_heavy (more) work off.
struct CoalesceArgs {
    const uint32_t* read_off;
    uint32_t r0, n_strands;
    uint32_t maxc;
    const uint32_t* strand_off;
    const uint32_t* strand_nseeds;
    const uint32_t* hit_ref;
    const uint32_t* hit_q;
    uint4* cand;
    uint32_t* cand_next;
    uint32_t* cand_status;
    uint32_t* strand_ncand;
    uint32_t* worklist;
    uint32_t* heavy_list;   // n_strands entries: 17..64-hit strands from the front, longer ones from the back
    uint64_t* counters;     // the lane's counter block: one pointer instead of four (this kernel sits at the SGPR limit)
};
// slots of the counter block (batch.hip's d_counters) the coalescing kernels use
__device__ inline uint32_t* co_wl_count(const CoalesceArgs& a) { return reinterpret_cast<uint32_t*>(a.counters + 1); }
__device__ inline uint32_t* co_heavy_count(const CoalesceArgs& a) { return reinterpret_cast<uint32_t*>(a.counters + 1) + 1; }
__device__ inline uint32_t* co_lane16_count(const CoalesceArgs& a) { return reinterpret_cast<uint32_t*>(a.counters + 15) + 1; }
__device__ inline uint32_t* co_mid_count(const CoalesceArgs& a) { return reinterpret_cast<uint32_t*>(a.counters + 15); }
__device__ inline unsigned long long* co_n_cand(const CoalesceArgs& a) { return reinterpret_cast<unsigned long long*>(a.counters + 3); }

// A strand per LANE.  Most strands hold no seed hit at all (the other strand of a read, reads without an origin), a
// chance hit or two, or the seven to ten hits of the read's own origin -- and the walk of index.rs:445-485 is sequential,
// so sixteen lanes per strand (the previous arrangement) repeated one walk sixteen times: 1.8e9 vector and 1.4e9
// scalar instructions for 10 M non-empty strands.  Here a lane sorts its strand's up to kLaneHits (reference, query)
// keys in registers with a fixed comparator network, walks them (the bin of a hit is looked up only when the sorted
// sites leave the current bin), keeps up to kLaneCands coalesced candidates in registers, ranks them by counting
// (stable: num_seeds descending, index.rs:369) and links the same-TaxId chains.  Strands with more hits go to
// k_coalesce_mid (up to 64) and k_coalesce_heavy; a strand whose hits coalesce into more than kLaneCands candidates is
// rare (every candidate beyond the first is a separate chance window) and goes to k_coalesce_mid as well.
// Everything a wavefront appends to a shared list -- work items, the strands it passes on -- is buffered in LDS and
// appended with one atomic per few hundred entries: a returning atomic on one address completes at ~11 ns, and one per
// 64 strands (312 k of them for 10 M reads) had been a floor of 3.4 ms under this kernel whatever its instructions.
constexpr int kLaneHitsFirst = 12, kLaneHitsSecond = 16, kLaneCands = 4;
constexpr uint32_t kPendCap = 768, kPendFlush = 512;  // work items a wavefront buffers in LDS; flushed with one atomic beyond kPendFlush
constexpr uint32_t kMidCap = 256, kMidFlush = 192;    // strands for k_coalesce_mid, likewise
constexpr uint32_t kHeavyCap = 128, kHeavyFlush = 64;  // ... and for k_coalesce_heavy

// ascending comparator network for 12 keys (Batcher's odd-even merge sort of 16 with the comparators of the four padding
// positions removed): 41 comparators
#define MTSV_SORT12(CX)                                                                                                  \
    CX(0, 1) CX(2, 3) CX(0, 2) CX(1, 3) CX(1, 2) CX(4, 5) CX(6, 7) CX(4, 6) CX(5, 7) CX(5, 6) CX(0, 4) CX(2, 6) CX(2, 4)  \
    CX(1, 5) CX(3, 7) CX(3, 5) CX(1, 2) CX(3, 4) CX(5, 6) CX(8, 9) CX(10, 11) CX(8, 10) CX(9, 11) CX(9, 10) CX(0, 8)      \
    CX(4, 8) CX(2, 10) CX(6, 10) CX(2, 4) CX(6, 8) CX(1, 9) CX(5, 9) CX(3, 11) CX(7, 11) CX(3, 5) CX(7, 9) CX(1, 2)       \
    CX(3, 4) CX(5, 6) CX(7, 8) CX(9, 10)

// the whole network for 16 keys: 63 comparators
#define MTSV_SORT16(CX) \
    CX(0, 1) CX(2, 3) CX(0, 2) CX(1, 3) CX(1, 2) CX(4, 5) CX(6, 7) CX(4, 6) CX(5, 7) CX(5, 6) CX(0, 4) CX(2, 6) \
    CX(2, 4) CX(1, 5) CX(3, 7) CX(3, 5) CX(1, 2) CX(3, 4) CX(5, 6) CX(8, 9) CX(10, 11) CX(8, 10) CX(9, 11) CX(9, 10) \
    CX(12, 13) CX(14, 15) CX(12, 14) CX(13, 15) CX(13, 14) CX(8, 12) CX(10, 14) CX(10, 12) CX(9, 13) CX(11, 15) \
    CX(11, 13) CX(9, 10) CX(11, 12) CX(13, 14) CX(0, 8) CX(4, 12) CX(4, 8) CX(2, 10) CX(6, 14) CX(6, 10) CX(2, 4) \
    CX(6, 8) CX(10, 12) CX(1, 9) CX(5, 13) CX(5, 9) CX(3, 11) CX(7, 15) CX(7, 11) CX(3, 5) CX(7, 9) CX(11, 13) CX(1, \
    2) CX(3, 4) CX(5, 6) CX(7, 8) CX(9, 10) CX(11, 12) CX(13, 14)

// HITS = 12, LISTED = false: the pass over all strands.  HITS = 16, LISTED = true: the strands of 13..16 seed hits the first
// pass put on a list of their own (a read's origin plus a few chance hits, or the edge of a second origin): a lane each
// as well, instead of a wavefront each in k_coalesce_mid.
template <int HITS, bool LISTED>
__global__ __launch_bounds__(256) void k_coalesce(DevIndexView ix, CoalesceArgs a) {
    __shared__ uint32_t pend_all[256 / kWave][kPendCap];
    __shared__ uint32_t mid_all[256 / kWave][kMidCap];
    __shared__ uint32_t second_all[LISTED ? 1 : 256 / kWave][LISTED ? 1 : kMidCap];
    __shared__ uint64_t walk_keys[LISTED ? HITS : 1][LISTED ? 256 : 1];
    __shared__ uint32_t heavy_all[256 / kWave][kHeavyCap];
    uint32_t* pend_buf = pend_all[threadIdx.x / kWave];
    uint32_t* mid_buf = mid_all[threadIdx.x / kWave];
    uint32_t* second_buf = second_all[LISTED ? 0 : threadIdx.x / kWave];
    uint32_t n_second = 0;
    uint32_t* heavy_buf = heavy_all[threadIdx.x / kWave];
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint32_t n_waves = gridDim.x * blockDim.x / kWave;
    uint32_t pend = 0, n_mid = 0, n_heavy = 0;  // entries buffered in LDS (wave-uniform)
    unsigned long long cand_sum = 0, ver_sum = 0, win_sum = 0;
    // (uniform values that are needed rarely live in vector registers: the kernel sits at the scalar register file's limit)
    uint32_t maxc = a.maxc, r0 = a.r0;
    asm volatile("" : "+v"(maxc), "+v"(r0));
    auto flush_pend = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(co_wl_count(a), pend);
        base = __builtin_amdgcn_readfirstlane(base);
        wave_mem_sync();
        for (uint32_t i = lane; i < pend; i += kWave) a.worklist[base + i] = pend_buf[i];
        wave_mem_sync();
        pend = 0;
    };
    auto flush_second = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(co_lane16_count(a), n_second);
        base = __builtin_amdgcn_readfirstlane(base);
        wave_mem_sync();
        for (uint32_t i = lane; i < n_second; i += kWave) a.heavy_list[a.n_strands + base + i] = second_buf[i];
        wave_mem_sync();
        n_second = 0;
    };
    auto flush_mid = [&]() {  // front of the strand list: k_coalesce_mid
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(co_mid_count(a), n_mid);
        base = __builtin_amdgcn_readfirstlane(base);
        wave_mem_sync();
        for (uint32_t i = lane; i < n_mid; i += kWave) a.heavy_list[base + i] = mid_buf[i];
        wave_mem_sync();
        n_mid = 0;
    };
    auto flush_heavy = [&]() {  // back of the strand list: k_coalesce_heavy
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(co_heavy_count(a), n_heavy);
        base = __builtin_amdgcn_readfirstlane(base);
        wave_mem_sync();
        for (uint32_t i = lane; i < n_heavy; i += kWave) a.heavy_list[a.n_strands - 1 - (base + i)] = heavy_buf[i];
        wave_mem_sync();
        n_heavy = 0;
    };
    const uint32_t n_items = LISTED ? *co_lane16_count(a) : a.n_strands;
    for (uint32_t base64 = wave * kWave; base64 < n_items; base64 += n_waves * kWave) {
        const bool valid = base64 + lane < n_items;
        const uint32_t rs = LISTED ? (valid ? a.heavy_list[a.n_strands + base64 + lane] : 0u) : base64 + lane;
        const uint32_t o = valid ? a.strand_off[rs] : 0;
        const uint32_t nh = valid ? a.strand_off[rs + 1] - o : 0;
        if (!LISTED && valid && nh == 0) a.strand_ncand[rs] = 0;
        const bool to_second = !LISTED && nh > (uint32_t)HITS && nh <= (uint32_t)kLaneHitsSecond;
        bool to_mid = nh > (uint32_t)(LISTED ? HITS : kLaneHitsSecond) && nh <= (uint32_t)kWave;
        const bool heavy = nh > (uint32_t)kWave;
        uint32_t gnc = 0;
        uint32_t cs[kLaneCands], ce[kLaneCands], cb[kLaneCands], cn[kLaneCands], ct[kLaneCands];
#pragma unroll
        for (int k = 0; k < kLaneCands; k++) cs[k] = ce[k] = cb[k] = cn[k] = ct[k] = 0;
        bool hopeless = false;
        if (nh >= 1 && nh <= (uint32_t)HITS) {
            const uint32_t ns_raw = a.strand_nseeds[rs];
            hopeless = (ns_raw & kHopeless) != 0;  // k_thin's flag (see kHopeless)
            const StrandGeom gg = strand_geom(a.read_off, r0 + (rs >> 1), ns_raw);
            uint64_t key[HITS];
#pragma unroll
            for (int i = 0; i < HITS; i++)
                key[i] = (uint32_t)i < nh ? ((uint64_t)a.hit_ref[o + i] << 32) | a.hit_q[o + i] : ~0ull;
            // seed_hits.sort(): (reference_offset, query_offset), index.rs:443
#define MTSV_CX(I, J)                      \
    {                                      \
        const uint64_t lo_ = key[I] < key[J] ? key[I] : key[J], hi_ = key[I] < key[J] ? key[J] : key[I]; \
        key[I] = lo_;                      \
        key[J] = hi_;                      \
    }
            if (HITS == 12) {
                MTSV_SORT12(MTSV_CX)
            } else {
                MTSV_SORT16(MTSV_CX)
            }
#undef MTSV_CX
            // the walk of index.rs:445-485
            bool have = false;
            uint32_t w_s = 0, w_e = 0, w_b = 0, w_n = 0, w_t = 0;
            uint32_t cur_b = 0xffffffffu;
            DevBin cur{0, 0, 0, 0};
            auto flush = [&]() {  // index.rs:467-469: the running candidate is kept if it has enough seeds
                if (have && w_n >= gg.min_seeds) {
#pragma unroll
                    for (int k = 0; k < kLaneCands; k++)
                        if (gnc == (uint32_t)k) {
                            cs[k] = w_s;
                            ce[k] = w_e;
                            cb[k] = w_b;
                            cn[k] = w_n;
                            ct[k] = w_t;
                        }
                    gnc++;
                }
            };
            // (the second pass walks its sixteen keys in a rolled loop out of LDS: unrolled sixteen times the walk does not fit
            //  the scalar register file)
            if (LISTED) {
#pragma unroll
                for (int i = 0; i < HITS; i++) walk_keys[LISTED ? i : 0][LISTED ? threadIdx.x : 0] = key[i];
            }
            auto walk_step = [&](uint64_t key_i) {
                    const uint32_t site = (uint32_t)(key_i >> 32), q = (uint32_t)key_i;
                    // first bin whose end > site (index.rs:455-458): the sites ascend, so it only changes when they pass its end
                    if (cur_b == 0xffffffffu || site >= cur.end) {
                        cur_b = min(find_bin(ix, site), ix.n_bins - 1);
                        cur = ix.bins[cur_b];
                    }
                    uint32_t ws, we;
                    const bool ok = candidate_window(site, q, cur, gg.L, gg.ED, &ws, &we);
                    const bool merge = have && ok && cur_b == w_b && ((w_s <= ws && ws < w_e) || (w_s < we && we <= w_e));
                    if (merge) {  // add_seed_hit, index.rs:216-229
                        w_s = min(w_s, ws);
                        w_e = max(w_e, we);
                        w_n++;
                    } else {
                        flush();
                        have = ok;  // ReferenceCandidate::new, index.rs:472,475
                        w_s = ws;
                        w_e = we;
                        w_b = cur_b;
                        w_n = 1;
                        w_t = cur.tax_id;
                    }
            };
            if (LISTED) {
#pragma unroll 1
                for (uint32_t i = 0; i < nh; i++) walk_step(walk_keys[LISTED ? i : 0][LISTED ? threadIdx.x : 0]);
            } else {
#pragma unroll
                for (int i = 0; i < HITS; i++)
                    if ((uint32_t)i < nh) walk_step(key[i]);
            }
            flush();  // index.rs:481-485
            if (gnc > (uint32_t)kLaneCands) {  // more candidates than the registers hold: k_coalesce_mid does this strand
                to_mid = true;
                gnc = 0;
            }
        }
        {   // the longer strands: 13..64 hits (and the rare overflow above) for k_coalesce_mid, > 64 for k_coalesce_heavy
            const unsigned long long bm = __ballot(to_mid), bh = __ballot(heavy);
            if (!LISTED) {  // 13..16 hits: the list of the second lane pass (behind the n_strands entries of the other two lists)
                const unsigned long long bs = __ballot(to_second);
                if (bs) {
                    if (to_second) second_buf[n_second + __popcll(bs & ((1ull << lane) - 1))] = rs;
                    n_second += (uint32_t)__popcll(bs);
                    if (n_second > kMidFlush) flush_second();
                }
            }
            if (bm) {
                if (to_mid) mid_buf[n_mid + __popcll(bm & ((1ull << lane) - 1))] = rs;
                n_mid += (uint32_t)__popcll(bm);
                if (n_mid > kMidFlush) flush_mid();
            }
            if (bh) {
                if (heavy) heavy_buf[n_heavy + __popcll(bh & ((1ull << lane) - 1))] = rs;
                n_heavy += (uint32_t)__popcll(bh);
                if (n_heavy > kHeavyFlush) flush_heavy();
            }
        }
        const bool mine = nh >= 1 && nh <= (uint32_t)HITS && !to_mid;  // this lane finishes its strand here
        // stable sort by num_seeds descending (index.rs:369) as a rank; the same-TaxId chain in rank order: a candidate is
        // verified only after every earlier candidate of its TaxId has failed (index.rs:393), so only the first of each
        // TaxId starts as work
        uint32_t rank[kLaneCands], nxt[kLaneCands];
        bool first[kLaneCands];
#pragma unroll
        for (int k = 0; k < kLaneCands; k++) {
            rank[k] = 0;
#pragma unroll
            for (int j = 0; j < kLaneCands; j++)
                if (j != k && (uint32_t)j < gnc) rank[k] += (cn[j] > cn[k]) || (cn[j] == cn[k] && j < k);
        }
#pragma unroll
        for (int k = 0; k < kLaneCands; k++) {
            nxt[k] = 0xffffffffu;
            first[k] = (uint32_t)k < gnc;
#pragma unroll
            for (int j = 0; j < kLaneCands; j++)
                if (j != k && (uint32_t)j < gnc && ct[j] == ct[k]) {
                    if (rank[j] > rank[k] && rank[j] < nxt[k]) nxt[k] = rank[j];
                    if (rank[j] < rank[k]) first[k] = false;
                }
        }
        uint32_t n_emit = 0;
        if (mine) {
            cand_sum += gnc;
            if (hopeless) {
                // the reference prefilters each candidate of such a strand (up to max_candidates) and none can pass: that
                // work is accounted here and the strand leaves the pipeline (no candidates to resolve)
                a.strand_ncand[rs] = 0;
                ver_sum += min(gnc, maxc);
#pragma unroll
                for (int k = 0; k < kLaneCands; k++)
                    if ((uint32_t)k < gnc && rank[k] < maxc) win_sum += ce[k] - cs[k];
            } else {
                a.strand_ncand[rs] = gnc;
#pragma unroll
                for (int k = 0; k < kLaneCands; k++)
                    if ((uint32_t)k < gnc) {
                        a.cand[o + rank[k]] = make_uint4(cs[k], ce[k], cb[k], rs);
                        a.cand_next[o + rank[k]] = nxt[k];
                        a.cand_status[o + rank[k]] = 0;
                        n_emit += first[k] && rank[k] < maxc;
                    }
            }
        }
        {   // the strands' first candidates of every TaxId become work items: lane-major into the wavefront's buffer
            uint32_t incl = n_emit;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= (uint32_t)d) incl += up;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
            uint32_t at = pend + incl - n_emit;
            if (n_emit) {
#pragma unroll
                for (int k = 0; k < kLaneCands; k++)
                    if ((uint32_t)k < gnc && first[k] && rank[k] < maxc) pend_buf[at++] = o + rank[k];
            }
            pend += total;
            if (pend > kPendFlush) flush_pend();
        }
    }
    if (pend) flush_pend();
    if (n_mid) flush_mid();
    if (!LISTED && n_second) flush_second();
    if (n_heavy) flush_heavy();
    for (int d = 32; d > 0; d >>= 1) {
        cand_sum += __shfl_down(cand_sum, d);
        ver_sum += __shfl_down(ver_sum, d);
        win_sum += __shfl_down(win_sum, d);
    }
    if (lane == 0 && cand_sum) atomicAdd(co_n_cand(a), cand_sum);
    if (lane == 0 && ver_sum) {  // the counters sit side by side: n_candidates, n_verified, window_bytes
        atomicAdd(co_n_cand(a) + 1, ver_sum);
        atomicAdd(co_n_cand(a) + 2, win_sum);
    }
}

// Strands of the list with up to 64 seed hits (those k_coalesce passes on): one wavefront each, everything in registers.
// one strand of 17..64 seed hits on one wavefront; returns its candidate count.  Not inlined on purpose: the
// caller's loop state and the argument block would otherwise all be live across this body (SGPR spills).
__device__ __attribute__((noinline)) uint32_t coalesce_mid_strand(const DevIndexView& ix, const CoalesceArgs& a, uint32_t rs, uint32_t o,
                                                                    uint32_t nh, uint32_t* pend_buf, uint32_t& pend) {
    const uint32_t lane = lane_id();
    const uint32_t ns_raw = a.strand_nseeds[rs];
    const bool hopeless = (ns_raw & kHopeless) != 0;  // see k_coalesce
    const StrandGeom g = strand_geom(a.read_off, a.r0 + (rs >> 1), ns_raw);
    uint32_t nc = 0;
    // ---- registers only ----
    uint64_t key = lane < nh ? ((uint64_t)a.hit_ref[o + lane] << 32) | a.hit_q[o + lane] : ~0ull;
    key = wave_bitonic_sort(key);  // seed_hits.sort(): (reference_offset, query_offset), index.rs:443
    uint32_t site = (uint32_t)(key >> 32), q = (uint32_t)key;
    uint32_t b = 0, ws = 0, we = 0;
    bool ok = false;
    if (lane < nh) {
        b = min(find_bin(ix, site), ix.n_bins - 1);
        DevBin bin = ix.bins[b];
        ok = candidate_window(site, q, bin, g.L, g.ED, &ws, &we);
    }
    Walk w{false, 0, 0, 0, 0};
    uint32_t ms = 0, me = 0, mb = 0, mn = 0;
    for (uint32_t i = 0; i < nh; i++) {
        uint32_t wsi = (uint32_t)__shfl((int)ws, (int)i), wei = (uint32_t)__shfl((int)we, (int)i);
        uint32_t bi = (uint32_t)__shfl((int)b, (int)i);
        bool oki = (uint32_t)__shfl((int)ok, (int)i) != 0;
        bool merge = w.have && oki && bi == w.b && ((w.s <= wsi && wsi < w.e) || (w.s < wei && wei <= w.e));
        if (merge) {  // add_seed_hit, index.rs:216-229
            w.s = min(w.s, wsi);
            w.e = max(w.e, wei);
            w.n++;
        } else {
            if (w.have && w.n >= g.min_seeds) {  // index.rs:467-469
                if (lane == nc) { ms = w.s; me = w.e; mb = w.b; mn = w.n; }
                nc++;
            }
            w.have = oki;  // ReferenceCandidate::new, index.rs:472,475
            w.s = wsi; w.e = wei; w.b = bi; w.n = 1;
        }
    }
    if (w.have && w.n >= g.min_seeds) {  // index.rs:481-485
        if (lane == nc) { ms = w.s; me = w.e; mb = w.b; mn = w.n; }
        nc++;
    }
    // stable sort by num_seeds descending (index.rs:369) as a rank computation
    uint32_t rank = 0;
    for (uint32_t j = 0; j < nc; j++) {
        uint32_t nj = (uint32_t)__shfl((int)mn, (int)j);
        rank += (nj > mn) || (nj == mn && j < lane);
    }
    // same-TaxID chain in rank order: a candidate is verified only after every earlier candidate
    // of its TaxID has failed (index.rs:393), so only the first of each TaxID starts as work
    uint32_t tax = lane < nc ? ix.bins[mb].tax_id : 0;
    uint32_t nxt = 0xffffffffu;
    bool first = lane < nc;
    for (uint32_t j = 0; j < nc; j++) {
        uint32_t tj = (uint32_t)__shfl((int)tax, (int)j), rj = (uint32_t)__shfl((int)rank, (int)j);
        if (tj == tax && rj > rank && rj < nxt) nxt = rj;
        if (tj == tax && rj < rank) first = false;
    }
    if (hopeless) {
        unsigned long long wsum = (lane < nc && rank < a.maxc) ? me - ms : 0;
        for (int d = 32; d > 0; d >>= 1) wsum += __shfl_down(wsum, d);
        if (lane == 0) {
            a.strand_ncand[rs] = 0;
            atomicAdd(co_n_cand(a) + 1, (unsigned long long)min(nc, a.maxc));
            atomicAdd(co_n_cand(a) + 2, wsum);
        }
        return nc;
    }
    if (lane < nc) {
        a.cand[o + rank] = make_uint4(ms, me, mb, rs);
        a.cand_next[o + rank] = nxt;
        a.cand_status[o + rank] = 0;
    }
    if (lane == 0) a.strand_ncand[rs] = nc;
    {
        bool emit = first && rank < a.maxc;
        unsigned long long em = __ballot(emit);
        uint32_t m = __popcll(em);
        if (pend + m > kWave) {  // flush the buffered items with one atomic
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(co_wl_count(a), pend);
            base = __builtin_amdgcn_readfirstlane(base);
            if (lane < pend) a.worklist[base + lane] = pend_buf[lane];
            pend = 0;
        }
        if (emit) pend_buf[pend + __popcll(em & ((1ull << lane) - 1))] = o + rank;
        pend += m;
    }
    return nc;
}

__global__ __launch_bounds__(256) void k_coalesce_mid(DevIndexView ix, CoalesceArgs a) {
    __shared__ uint32_t pend_all[256 / kWave][kWave];
    uint32_t* pend_buf = pend_all[threadIdx.x / kWave];
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint32_t n_waves = gridDim.x * blockDim.x / kWave;
    const uint32_t n_list = *co_mid_count(a);
    uint32_t pend = 0;
    unsigned long long cand_sum = 0;
    for (uint32_t h = wave; h < n_list; h += n_waves) {
        const uint32_t rs = a.heavy_list[h];
        const uint32_t o = a.strand_off[rs];
        const uint32_t nh = __builtin_amdgcn_readfirstlane(a.strand_off[rs + 1] - o);
        const uint32_t nc = coalesce_mid_strand(ix, a, rs, o, nh, pend_buf, pend);  // writes strand_ncand
        if (lane == 0) cand_sum += nc;
    }
    if (pend) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(co_wl_count(a), pend);
        base = __builtin_amdgcn_readfirstlane(base);
        wave_mem_sync();
        if (lane < pend) a.worklist[base + lane] = pend_buf[lane];
    }
    for (int d = 32; d > 0; d >>= 1) cand_sum += __shfl_down(cand_sum, d);
    if (lane == 0 && cand_sum) atomicAdd(co_n_cand(a), cand_sum);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch_coalesce(hipStream_t s, const DevIndexView& ix, const uint32_t* read_off, uint32_t r0, uint32_t n_strands,
                     int64_t max_candidates, const uint32_t* strand_off,
                     const uint32_t* strand_nseeds, const uint32_t* hit_ref, const uint32_t* hit_q, uint64_t* hit_key,
                     uint64_t* cand_tmp, uint4* cand, uint32_t* cand_next, uint32_t* cand_status,
                     uint32_t* strand_ncand, uint32_t* worklist, uint32_t* heavy_list, uint64_t* counters) {
    uint32_t* wl_count = reinterpret_cast<uint32_t*>(counters + 1);
    uint32_t* heavy_count = wl_count + 1;
    unsigned long long* n_cand_total = reinterpret_cast<unsigned long long*>(counters + 3);
    // Grids follow the work on passes of a few hundred thousand reads: a workgroup costs ~0.1 us to dispatch whether it
    // finds work or not, and the kernels of the other lanes wait for the same dispatcher (a host batch of 10 M reads in
    // ranges of 0.3 - 1 M: 39.7 -> 38.8 ms).  Every kernel here strides over its list; a wavefront takes 64 strands per
    // trip, at least four trips before the grid grows, but never fewer than 256 workgroups (short lists are latency-bound).
    uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(cdiv(n_strands, 16), std::max<uint32_t>(cdiv(n_strands, 4 * 256), 256)), 256 * 8));
    CoalesceArgs c;
    c.read_off = read_off;
    c.r0 = r0;
    c.n_strands = n_strands;
    c.maxc = rank_bound(max_candidates);
    c.strand_off = strand_off;
    c.strand_nseeds = strand_nseeds;
    c.hit_ref = hit_ref;
    c.hit_q = hit_q;
    c.cand = cand;
    c.cand_next = cand_next;
    c.cand_status = cand_status;
    c.strand_ncand = strand_ncand;
    c.worklist = worklist;
    c.heavy_list = heavy_list;
    c.counters = counters;
    hipLaunchKernelGGL((k_coalesce<kLaneHitsFirst, false>), dim3(blocks), dim3(256), 0, s, ix, c);
    // the 13..16-hit strands it listed (a few percent of the strands: a modest grid strides over the device-side count)
    hipLaunchKernelGGL((k_coalesce<kLaneHitsSecond, true>), dim3(std::min<uint32_t>(std::max<uint32_t>(n_strands / 8192, 8), 512)), dim3(256), 0, s, ix, c);
    // the list of longer strands is short (device-side count): a modest grid of wavefronts walks it
    hipLaunchKernelGGL(k_coalesce_mid, dim3(std::min<uint32_t>(std::max<uint32_t>(n_strands / 2048, std::min<uint32_t>(std::max<uint32_t>(n_strands / 256, 1), 128)), 1024)), dim3(256), 0, s, ix, c);
    HeavyArgs a;
    a.read_off = read_off;
    a.r0 = r0;
    a.maxc = rank_bound(max_candidates);
    a.strand_off = strand_off;
    a.strand_nseeds = strand_nseeds;
    a.hit_ref = hit_ref;
    a.hit_q = hit_q;
    a.hit_key = hit_key;
    a.cand_tmp = cand_tmp;
    a.cand = cand;
    a.cand_next = cand_next;
    a.cand_status = cand_status;
    a.strand_ncand = strand_ncand;
    a.worklist = worklist;
    a.wl_count = wl_count;
    a.n_cand_total = n_cand_total;
    // persistent workgroups over the (device-side) list of heavy strands; usually a few thousand at most
    const dim3 hgrid(std::min<uint32_t>(std::max<uint32_t>(n_strands / 64, 1), 2048));
    hipLaunchKernelGGL(k_coalesce_heavy<kHeavyKeysSmall>, hgrid, dim3(256), 0, s, ix, a, heavy_list, n_strands, heavy_count);
    hipLaunchKernelGGL(k_coalesce_heavy<kHeavyKeys>, dim3(std::min<uint32_t>(hgrid.x, 512)), dim3(256), 0, s, ix, a, heavy_list, n_strands, heavy_count);
}

// longest candidate window of a pass (the tiled kernel's strips are sized from it)
namespace {
__global__ __launch_bounds__(256) void k_max_window(uint32_t n_strands, const uint32_t* __restrict__ strand_off,
                                                    const uint32_t* __restrict__ strand_ncand, const uint4* __restrict__ cand,
                                                    unsigned long long* __restrict__ out) {
    const uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t m = 0;
    if (rs < n_strands) {
        const uint32_t o = strand_off[rs], nc = strand_ncand[rs];
        for (uint32_t i = 0; i < nc; i++) {
            const uint4 c = cand[o + i];
            m = max(m, c.y - c.x);
        }
    }
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, d));
    if (lane_id() == 0 && m) atomicMax(out, (unsigned long long)m);
}
}  // namespace

void launch_max_window(hipStream_t s, uint32_t n_strands, const uint32_t* strand_off, const uint32_t* strand_ncand,
                       const uint4* cand, unsigned long long* out) {
    hipLaunchKernelGGL(k_max_window, dim3(cdiv(n_strands, 256)), dim3(256), 0, s, n_strands, strand_off, strand_ncand, cand, out);
}

}  // namespace mtsv

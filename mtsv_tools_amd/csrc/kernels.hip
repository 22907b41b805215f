// kernels.hip -- the mtsv-binner hot path as gfx950 kernels.
//
// One batch of reads flows through staged kernels with worklists in HBM:
//
//   k_search    lane per (read, strand, seed): FMIndex::backward_search           index.rs:305
//   k_thin      lane per strand: adaptive seed thinning / max_hits filter         index.rs:293-344,354
//   scan        exclusive scan of per-strand seed-hit counts
//   k_expand    lane per strand: SA rows of every kept seed (Interval::occ)       index.rs:347-352
//   k_locate    lane per seed hit with wavefront refill: SampledSuffixArray::get  index.rs:347
//   k_coalesce  wavefront per strand: sort, coalesce_seed_sites, min_seeds, rank  index.rs:358-369,435-487
//   k_verify    wavefront per strand with candidates: SW prefilter + edit
//               distance + the ordered selection loop                             index.rs:375-431,
//                                                                                 ssw.c:123-328, align.rs:28-85
//   scan + k_gather  compact per-strand hits into (read, strand, rank) order      binner.rs:128
//
// All arithmetic is integer; positions are u32 (n < 2^32).  No MFMA: the path is rank queries and
// small dynamic programs.
#include "kernels.hpp"

namespace mtsv {

namespace {

constexpr int kWave = 64;

__device__ inline uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

// ---------------------------------------------------------------------------------------------
// strand access: symbol code of position p of strand `strand` of a read (binner.rs:88-100,115)
// ---------------------------------------------------------------------------------------------
__device__ inline uint32_t strand_code(const uint8_t* __restrict__ read, uint32_t L, uint32_t strand, uint32_t p) {
    return strand ? comp_code(base_code(read[L - 1 - p])) : base_code(read[p]);
}

__device__ inline uint32_t n_seeds_of(uint32_t L, uint32_t K, uint32_t G) {
    // offsets 0, G, 2G, ... < L + 1 - K  (index.rs:284-286); L + 1 < K is trapped as "no seeds"
    return (L >= K) ? (L - K) / G + 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// K1: backward search, one lane per seed slot
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_search(DevIndexView ix, const uint8_t* __restrict__ bases,
                                                const uint32_t* __restrict__ read_off, uint32_t r0, uint32_t n_reads,
                                                uint32_t max_ns, uint32_t K, uint32_t G,
                                                uint32_t* __restrict__ seed_lo, uint32_t* __restrict__ seed_cnt) {
    uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)n_reads * 2 * max_ns;
    if (slot >= total) return;
    uint32_t j = (uint32_t)(slot % max_ns);
    uint64_t rs = slot / max_ns;
    uint32_t strand = (uint32_t)(rs & 1);
    uint32_t r = r0 + (uint32_t)(rs >> 1);
    uint32_t b0 = read_off[r], L = read_off[r + 1] - b0;
    uint32_t ns = n_seeds_of(L, K, G);
    uint32_t lo = 0, hi = 0;
    if (j < ns) {
        const uint8_t* read = bases + b0;
        uint32_t off = j * G;
        lo = 0;
        hi = ix.n;
        int i = (int)K - 1;
        // the seed's last kmer_k symbols in one gather when none of them is N
        if (ix.kmer_tab && K >= ix.kmer_k) {
            uint32_t idx = 0;
            bool acgt = true;
            for (uint32_t t = 0; t < ix.kmer_k; t++) {
                uint32_t a = strand_code(read, L, strand, off + K - ix.kmer_k + t);
                acgt &= a < 4;
                idx = (idx << 2) | (a & 3);
            }
            if (acgt) {
                uint2 iv = ix.kmer_tab[idx];
                lo = iv.x;
                hi = iv.y;
                i = (int)K - 1 - (int)ix.kmer_k;
            }
        }
        for (; i >= 0 && lo < hi; i--) {
            uint32_t a = strand_code(read, L, strand, off + (uint32_t)i);
            uint32_t bl = lo >> kBlockShift, bh = hi >> kBlockShift;
            LoadedBlock B0 = load_block(ix.blocks, bl);
            uint32_t nlo = ix.C[a] + block_rank(B0, a, bl, lo & (kBlockRows - 1), ix.sentinel_row);
            uint32_t nhi;
            if (bh == bl) {
                nhi = ix.C[a] + block_rank(B0, a, bh, hi & (kBlockRows - 1), ix.sentinel_row);
            } else {
                LoadedBlock B1 = load_block(ix.blocks, bh);
                nhi = ix.C[a] + block_rank(B1, a, bh, hi & (kBlockRows - 1), ix.sentinel_row);
            }
            lo = nlo;
            hi = nhi;
        }
        if (lo >= hi) lo = hi = 0;  // Partial / Absent: only Complete intervals count (index.rs:312-332)
    }
    seed_lo[slot] = lo;
    seed_cnt[slot] = hi - lo;
}

// ---------------------------------------------------------------------------------------------
// K1b: sequential seed policy of one strand (index.rs:293-344,354)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_thin(const uint32_t* __restrict__ read_off, uint32_t r0, uint32_t n_reads,
                                              uint32_t max_ns, uint32_t K, uint32_t G, uint64_t max_hits,
                                              uint64_t tune_max_hits, uint32_t* __restrict__ seed_cnt,
                                              uint32_t* __restrict__ strand_hits, uint32_t* __restrict__ strand_nseeds) {
    uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rs >= n_reads * 2) return;
    uint32_t r = r0 + (rs >> 1);
    uint32_t L = read_off[r + 1] - read_off[r];
    uint32_t ns = n_seeds_of(L, K, G);
    uint64_t next_offset = 0, seed_interval = G;
    uint32_t total = 0, nseeds = 0;
    uint32_t* cnt = seed_cnt + (uint64_t)rs * max_ns;
    for (uint32_t j = 0; j < ns; j++) {
        uint64_t offset = (uint64_t)j * G;
        uint32_t c = cnt[j];
        if (offset < next_offset) {  // index.rs:300-302
            if (c) cnt[j] = 0;
            continue;
        }
        if (c == 0) continue;        // index.rs:330-332
        if ((uint64_t)c > max_hits) {  // index.rs:335-337
            cnt[j] = 0;
            continue;
        }
        if ((uint64_t)c > tune_max_hits) {  // index.rs:338-344
            seed_interval *= 2;
            next_offset = offset + seed_interval;
        }
        total += c;
        nseeds++;
    }
    strand_hits[rs] = total;
    strand_nseeds[rs] = nseeds;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan of u32 counts (block sums in u64 so the host can detect > 2^32 totals)
// ---------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    for (int d = 1; d < kWave; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const uint32_t* __restrict__ in, uint32_t n,
                                                                 uint64_t* __restrict__ tile_sums) {
    __shared__ uint32_t ws[kScanThreads / kWave];
    uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t s = 0;
    for (int i = 0; i < kScanItems; i++)
        if (base + i < n) s += in[base + i];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if (lane_id() == 0) ws[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < kScanThreads / kWave; w++) t += ws[w];
        tile_sums[blockIdx.x] = t;
    }
}

// single block: exclusive scan of tile sums in place; writes the grand total to *total
__global__ __launch_bounds__(1024) void k_scan_sums(uint64_t* __restrict__ tile_sums, uint32_t n_tiles,
                                                    uint64_t* __restrict__ total) {
    __shared__ uint64_t buf[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < n_tiles ? tile_sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            uint64_t o = threadIdx.x >= (uint32_t)d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += o;
            __syncthreads();
        }
        uint64_t incl = buf[threadIdx.x];
        uint64_t c = carry;
        if (i < n_tiles) tile_sums[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// out[i] = exclusive prefix, out[n] = total
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t n,
                                                             const uint64_t* __restrict__ tile_sums,
                                                             const uint64_t* __restrict__ total,
                                                             uint32_t* __restrict__ out) {
    __shared__ uint32_t ws[kScanThreads / kWave];
    uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t s = 0;
    for (int i = 0; i < kScanItems; i++) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    uint32_t incl = wave_incl_scan(s);
    if (lane_id() == kWave - 1) ws[threadIdx.x / kWave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < threadIdx.x / kWave; w++) wbase += ws[w];
    uint32_t run = (uint32_t)tile_sums[blockIdx.x] + wbase + incl - s;
    for (int i = 0; i < kScanItems; i++) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (uint32_t)*total;
}

// ---------------------------------------------------------------------------------------------
// K1c: expand kept seeds into SA rows (or straight into text positions with the full SA)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_expand(DevIndexView ix, uint32_t n_strands, uint32_t max_ns, uint32_t G,
                                                const uint32_t* __restrict__ seed_lo,
                                                const uint32_t* __restrict__ seed_cnt,
                                                const uint32_t* __restrict__ strand_off,
                                                uint32_t* __restrict__ hit_row, uint32_t* __restrict__ hit_ref,
                                                uint32_t* __restrict__ hit_q) {
    uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rs >= n_strands) return;
    uint32_t o = strand_off[rs], end = strand_off[rs + 1];
    if (o == end) return;
    const uint32_t* lo = seed_lo + (uint64_t)rs * max_ns;
    const uint32_t* cnt = seed_cnt + (uint64_t)rs * max_ns;
    for (uint32_t j = 0; o < end; j++) {
        uint32_t c = cnt[j], l = lo[j];
        for (uint32_t i = 0; i < c; i++, o++) {
            if (ix.sa_full)
                hit_ref[o] = ix.sa_full[l + i];
            else
                hit_row[o] = l + i;
            hit_q[o] = j * G;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: locate by LF-walk over the row-sampled SA; finished lanes pull the next hit of the
// wavefront's chunk (ballot + prefix count) so the geometric walk lengths do not idle the wave
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kLocateChunk = 512;

__global__ __launch_bounds__(256) void k_locate(DevIndexView ix, const uint32_t* __restrict__ total_hits,
                                                const uint32_t* __restrict__ hit_row,
                                                uint32_t* __restrict__ hit_ref, unsigned long long* __restrict__ lf_steps) {
    const uint32_t total = *total_hits;
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint32_t n_waves = gridDim.x * blockDim.x / kWave;
    unsigned long long my_steps = 0;
    for (uint64_t cbase = (uint64_t)wave * kLocateChunk; cbase < total; cbase += (uint64_t)n_waves * kLocateChunk) {
        uint32_t next = (uint32_t)cbase;
        const uint32_t end = (uint32_t)min((uint64_t)total, cbase + kLocateChunk);
        bool active = false;
        uint32_t idx = 0, row = 0, steps = 0;
        for (;;) {
            unsigned long long need = __ballot(!active);
            uint32_t take = next + __popcll(need & ((1ull << lane) - 1));
            if (!active && take < end) {
                idx = take;
                row = hit_row[idx];
                steps = 0;
                active = true;
            }
            next += __popcll(need);
            if (!__any(active)) break;
            if (active) {
                bool sampled = ix.sa_pow2_shift != 0xffffffffu ? (row & (ix.sa_s - 1)) == 0 : (row % ix.sa_s) == 0;
                if (sampled) {
                    uint32_t j = ix.sa_pow2_shift != 0xffffffffu ? row >> ix.sa_pow2_shift : row / ix.sa_s;
                    hit_ref[idx] = ix.sa_sample[j] + steps;
                    active = false;
                } else {
                    uint32_t blk = row >> kBlockShift, off = row & (kBlockRows - 1);
                    LoadedBlock b = load_block(ix.blocks, blk);
                    uint32_t c = block_code(b, off);
                    if (c == kCodeSentinel) {  // extra_rows: this row is the suffix at text position 0
                        hit_ref[idx] = steps;
                        active = false;
                    } else {
                        row = ix.C[c] + block_rank(b, c, blk, off, ix.sentinel_row);
                        steps++;
                        my_steps++;
                        if (steps > ix.n) {  // only a corrupt index can cycle without a sampled row
                            hit_ref[idx] = 0;
                            active = false;
                        }
                    }
                }
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) my_steps += __shfl_down(my_steps, d);
    if (lane == 0 && my_steps) atomicAdd(lf_steps, my_steps);
}

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

// first bin whose end > site (the forward-only cursor of index.rs:455-458 on sorted hits)
__device__ inline uint32_t find_bin(const DevIndexView& ix, uint32_t site) {
    uint32_t lo = 0, hi = ix.n_bins;
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

__device__ inline StrandGeom strand_geom(const uint32_t* read_off, uint32_t r, double edit_rate, double min_seed,
                                         uint32_t nseeds) {
    StrandGeom g;
    g.L = read_off[r + 1] - read_off[r];
    g.ED = (uint32_t)ceil((double)g.L * edit_rate);           // index.rs:281-282
    double ms = floor((double)nseeds * min_seed);             // index.rs:358
    g.min_seeds = ms < 1.0 ? 1u : (ms > 4294967295.0 ? 0xffffffffu : (uint32_t)ms);
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

// bitonic network in its flip/disperse form: every comparator puts the minimum at the lower index,
// so virtual +inf padding above n never moves and any n works.  Keys live in L2 (agent-scope
// relaxed accesses bypass the per-CU L1); one wavefront owns the segment.
__device__ void global_bitonic_sort(uint64_t* keys, uint32_t n) {
    const uint32_t lane = lane_id();
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    const uint32_t half = np2 >> 1;
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        for (uint32_t i = lane; i < half; i += kWave) {  // flip
            uint32_t hk = k >> 1;
            uint32_t blk = i / hk, pos = i % hk;
            uint32_t a = blk * k + pos, b = blk * k + k - 1 - pos;
            if (b < n) {
                uint64_t x = gload(keys + a), y = gload(keys + b);
                if (x > y) {
                    gstore(keys + a, y);
                    gstore(keys + b, x);
                }
            }
        }
        wave_mem_sync();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {  // disperse
            for (uint32_t i = lane; i < half; i += kWave) {
                uint32_t blk = i / j, pos = i % j;
                uint32_t a = blk * 2 * j + pos, b = a + j;
                if (b < n) {
                    uint64_t x = gload(keys + a), y = gload(keys + b);
                    if (x > y) {
                        gstore(keys + a, y);
                        gstore(keys + b, x);
                    }
                }
            }
            wave_mem_sync();
        }
    }
}

__global__ __launch_bounds__(256) void k_coalesce(DevIndexView ix, const uint32_t* __restrict__ read_off, uint32_t r0,
                                                  uint32_t n_strands, double edit_rate, double min_seed,
                                                  const uint32_t* __restrict__ strand_off,
                                                  const uint32_t* __restrict__ strand_nseeds,
                                                  const uint32_t* __restrict__ hit_ref,
                                                  const uint32_t* __restrict__ hit_q, uint64_t* __restrict__ hit_key,
                                                  uint64_t* __restrict__ cand_tmp, uint4* __restrict__ cand,
                                                  uint32_t* __restrict__ strand_ncand,
                                                  uint32_t* __restrict__ worklist, uint32_t* __restrict__ wl_count,
                                                  unsigned long long* __restrict__ n_cand_total) {
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint32_t n_waves = gridDim.x * blockDim.x / kWave;
    for (uint32_t rs = wave; rs < n_strands; rs += n_waves) {
        const uint32_t o = strand_off[rs];
        const uint32_t nh = __builtin_amdgcn_readfirstlane(strand_off[rs + 1] - o);
        if (nh == 0) {
            if (lane == 0) strand_ncand[rs] = 0;
            continue;
        }
        const StrandGeom g = strand_geom(read_off, r0 + (rs >> 1), edit_rate, min_seed, strand_nseeds[rs]);
        uint32_t nc = 0;
        if (nh <= kWave) {
            // ---- registers only ----
            uint64_t key = lane < nh ? ((uint64_t)hit_ref[o + lane] << 32) | hit_q[o + lane] : ~0ull;
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
                uint32_t nj = __builtin_amdgcn_readlane(mn, j);
                rank += (nj > mn) || (nj == mn && j < lane);
            }
            if (lane < nc) cand[o + rank] = make_uint4(ms, me, mb, mn);
        } else {
            // ---- segments longer than a wavefront: sort in L2-resident scratch ----
            uint64_t* keys = hit_key + o;
            for (uint32_t i = lane; i < nh; i += kWave) gstore(keys + i, ((uint64_t)hit_ref[o + i] << 32) | hit_q[o + i]);
            wave_mem_sync();
            global_bitonic_sort(keys, nh);
            Walk w{false, 0, 0, 0, 0};
            uint64_t* ct = cand_tmp + 2ull * o;
            for (uint32_t base = 0; base < nh; base += kWave) {
                uint32_t cntv = min((uint32_t)kWave, nh - base);
                uint32_t b = 0, ws = 0, we = 0;
                bool ok = false;
                if (lane < cntv) {
                    uint64_t key = gload(keys + base + lane);
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
                    if (merge) {
                        w.s = min(w.s, wsi);
                        w.e = max(w.e, wei);
                        w.n++;
                    } else {
                        if (w.have && w.n >= g.min_seeds) {
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
            if (w.have && w.n >= g.min_seeds) {
                if (lane == 0) {
                    gstore(ct + 2ull * nc, ((uint64_t)w.e << 32) | w.s);
                    gstore(ct + 2ull * nc + 1, ((uint64_t)w.n << 32) | w.b);
                }
                nc++;
            }
            wave_mem_sync();
            // rank: sort (num_seeds descending, walk order ascending)
            for (uint32_t i = lane; i < nc; i += kWave) {
                uint32_t n_i = (uint32_t)(gload(ct + 2ull * i + 1) >> 32);
                gstore(keys + i, ((uint64_t)(0xffffffffu - n_i) << 32) | i);
            }
            wave_mem_sync();
            global_bitonic_sort(keys, nc);
            for (uint32_t i = lane; i < nc; i += kWave) {
                uint32_t src = (uint32_t)gload(keys + i);
                uint64_t se = gload(ct + 2ull * src), bn = gload(ct + 2ull * src + 1);
                cand[o + i] = make_uint4((uint32_t)se, (uint32_t)(se >> 32), (uint32_t)bn, (uint32_t)(bn >> 32));
            }
        }
        if (lane == 0) {
            strand_ncand[rs] = nc;
            if (nc) {
                worklist[atomicAdd(wl_count, 1u)] = rs;
                atomicAdd(n_cand_total, (unsigned long long)nc);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4: verification.  One wavefront per strand walks its ranked candidates in order; each
// candidate's window is swept once by an anti-diagonal pipeline over the lanes (lane l owns read
// rows [l*R, l*R+R)), computing in packed 16-bit halves of one register
//   low  half: the Smith-Waterman local score   H = max(0, diag+s, up-1, left-1)   (ssw.c:123-328
//              with the matrix of ssw/src/lib.rs:11-16 and gap 1/1; equals the striped byte
//              kernel while the score stays below 254, i.e. for every read up to 253 bases)
//   high half: the negated semi-global edit distance -D, so that
//              -D = max(diag-delta, up-1, left-1) shares the same max/add instructions
//              (align.rs:28-85: first row 0, first column i, answer = min of the last row).
// ---------------------------------------------------------------------------------------------
typedef short pk16 __attribute__((ext_vector_type(2)));

__device__ inline pk16 pk(int lo, int hi) {
    pk16 r;
    r.x = (short)lo;
    r.y = (short)hi;
    return r;
}
__device__ inline pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
__device__ inline uint32_t pk_bits(pk16 a) { return __builtin_bit_cast(uint32_t, a); }
__device__ inline pk16 pk_from_bits(uint32_t u) { return __builtin_bit_cast(pk16, u); }

constexpr uint32_t kRingChunk = 256;             // window bytes staged per refill
constexpr uint32_t kRingBytes = 2 * kRingChunk;  // two chunks: lanes lag the head by < 64 columns

template <int R>
__device__ void dp_sweep(const uint8_t* __restrict__ text, uint32_t text_n, uint32_t wstart, uint32_t Wn,
                         const uint32_t (&qc)[R], uint32_t L, uint8_t* ring, uint32_t* sw_out, uint32_t* ed_out) {
    const uint32_t lane = lane_id();
    const uint32_t nl = (L + R - 1) / R;  // lanes that own at least one read row
    const uint32_t row0 = lane * R;
    pk16 h[R], mv[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        h[r] = pk(0, -(int)(row0 + r + 1));            // column-0 boundary: H = 0, D[i][0] = i
        mv[r] = qc[r] == kCodeN ? pk(1, -1) : pk(1, 0);  // N/N scores +1 in SW, never matches in edit distance
    }
    pk16 up_prev = pk(0, -(int)row0);
    pk16 best = pk(0, -32768);
    pk16 last = pk(0, -(int)L);  // min over the last row starts at D[L][0] = L
    const int rstar = (int)((L - 1) % R);
    const pk16 one = pk(1, 1), miss = pk(-1, -1), clamp = pk(0, -32768);
    const uint32_t steps = Wn + nl - 1;
    for (uint32_t t = 0; t < steps; t++) {
        if ((t & (kRingChunk - 1)) == 0 && t < Wn) {
            // stage window columns [t, t+256) into the ring half t/256 & 1
            uint32_t half = (t / kRingChunk) & 1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t col = t + lane * 4 + k;
                uint32_t pos = wstart + col;
                uint8_t c = (col < Wn && pos < text_n) ? text[pos] : (uint8_t)7;
                ring[half * kRingChunk + lane * 4 + k] = c;
            }
        }
        uint32_t in_bits = __shfl_up(pk_bits(h[R - 1]), 1);
        pk16 in = lane == 0 ? pk(0, 0) : pk_from_bits(in_bits);  // top boundary: H = 0, D[0][j] = 0
        int j = (int)t - (int)lane;
        if (j >= 0 && j < (int)Wn && lane < nl) {
            uint32_t c = ring[(uint32_t)j & (kRingBytes - 1)];
            pk16 diag = up_prev, up = in;
#pragma unroll
            for (int r = 0; r < R; r++) {
                pk16 sv = qc[r] == c ? mv[r] : miss;
                pk16 x = diag + sv;
                pk16 y = pk_max(up, h[r]) - one;
                pk16 v = pk_max(pk_max(x, y), clamp);
                diag = h[r];
                h[r] = v;
                up = v;
                best = pk_max(best, v);
                if (r == rstar) last = pk_max(last, v);
            }
            up_prev = in;
        }
    }
    int sw = best.x;
    for (int d = 32; d > 0; d >>= 1) sw = max(sw, __shfl_xor(sw, d));
    int lastv = __shfl((int)last.y, (int)((L - 1) / R));
    *sw_out = (uint32_t)sw;
    *ed_out = (uint32_t)(-lastv);
}

template <int R>
__device__ void verify_strand(const DevIndexView& ix, const VerifyArgs& a, uint32_t rs, uint8_t* ring,
                              unsigned long long* wave_verified, unsigned long long* wave_wbytes) {
    const uint32_t lane = lane_id();
    const uint32_t r = a.r0 + (rs >> 1), strand = rs & 1;
    const uint32_t o = a.strand_off[rs];
    const uint32_t nc = __builtin_amdgcn_readfirstlane(a.strand_ncand[rs]);
    const uint32_t b0 = a.read_off[r], L = a.read_off[r + 1] - b0;
    const uint32_t ED = (uint32_t)ceil((double)L * a.edit_rate);
    const uint64_t thr = (uint64_t)L - 2ull * ED;  // usize arithmetic, wraps like the release build (index.rs:406)
    const uint8_t* read = a.bases + b0;
    uint32_t qc[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        uint32_t row = lane * R + k;
        qc[k] = row < L ? strand_code(read, L, strand, row) : 6u;
    }
    uint32_t my_tax = 0;  // lane m remembers the m-th matched TaxId (m < 64)
    uint32_t nout = 0, checked = 0;
    uint4* out = a.out + o;
    for (uint32_t i = 0; i < nc; i++) {
        if (a.max_candidates >= 0 && (uint64_t)checked >= (uint64_t)a.max_candidates) break;  // index.rs:385-389
        checked++;
        uint4 c = a.cand[o + i];
        DevBin bin = ix.bins[c.z];
        // matches.iter().find(...), index.rs:393
        bool dup = __any(lane < min(nout, (uint32_t)kWave) && my_tax == bin.tax_id);
        for (uint32_t m = kWave + lane; !dup && m - lane < nout; m += kWave) {
            uint32_t t = m < nout ? (uint32_t)__hip_atomic_load(&out[m].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : bin.tax_id + 1;
            dup = __any(m < nout && t == bin.tax_id);
        }
        if (dup) continue;
        uint32_t Wn = c.y - c.x;
        uint32_t sw = 0, ed = 0;
        dp_sweep<R>(ix.text, ix.n, c.x, Wn, qc, L, ring, &sw, &ed);
        (*wave_verified)++;
        (*wave_wbytes) += Wn;
        if ((uint64_t)sw >= thr && ed <= ED) {  // index.rs:406,410
            if (lane == (nout & (kWave - 1)) && nout < kWave) my_tax = bin.tax_id;
            if (lane == 0) out[nout] = make_uint4(bin.tax_id, bin.gi, c.x >= bin.start ? c.x - bin.start : 0, ed);
            wave_mem_sync();
            nout++;
            if (a.max_assignments >= 0 && (uint64_t)nout >= (uint64_t)a.max_assignments) break;  // index.rs:421-425
        }
    }
    if (lane == 0) a.strand_nout[rs] = nout;
}

__global__ __launch_bounds__(256) void k_verify(DevIndexView ix, VerifyArgs a) {
    __shared__ uint8_t ring_all[4][kRingBytes];
    uint8_t* ring = ring_all[threadIdx.x / kWave];
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const uint32_t n_waves = gridDim.x * blockDim.x / kWave;
    const uint32_t n_work = *a.wl_count;
    unsigned long long verified = 0, wbytes = 0;
    for (uint32_t w = wave; w < n_work; w += n_waves) {
        uint32_t rs = a.worklist[w];
        uint32_t r = a.r0 + (rs >> 1);
        uint32_t L = a.read_off[r + 1] - a.read_off[r];
        uint32_t R = (L + kWave - 1) / kWave;
        switch (R) {
        case 1: verify_strand<1>(ix, a, rs, ring, &verified, &wbytes); break;
        case 2: verify_strand<2>(ix, a, rs, ring, &verified, &wbytes); break;
        case 3: verify_strand<3>(ix, a, rs, ring, &verified, &wbytes); break;
        case 4: verify_strand<4>(ix, a, rs, ring, &verified, &wbytes); break;
        default: break;  // host rejects longer reads (MTSV_E_LIMIT)
        }
    }
    if (lane == 0 && verified) {
        atomicAdd(a.n_verified, verified);
        atomicAdd(a.window_bytes, wbytes);
    }
}

// strands without candidates never reach k_verify: zero their output count
__global__ __launch_bounds__(256) void k_zero_u32(uint32_t* p, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// K5: gather per-strand hits into the final (read, strand, rank) order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(uint32_t n_strands, uint32_t r0, const uint32_t* __restrict__ strand_off,
                                                const uint32_t* __restrict__ strand_nout,
                                                const uint32_t* __restrict__ out_off, const uint4* __restrict__ out,
                                                DevHit* __restrict__ hits, uint64_t hits_base) {
    uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rs >= n_strands) return;
    uint32_t n = strand_nout[rs];
    if (!n) return;
    uint32_t src = strand_off[rs];
    uint64_t dst = hits_base + out_off[rs];
    for (uint32_t i = 0; i < n; i++) {
        uint4 v = out[src + i];
        DevHit h;
        h.read = r0 + (rs >> 1);
        h.tax_id = v.x;
        h.gi = v.y;
        h.edit = v.w;
        h.strand = rs & 1;
        h.offset = v.z;
        hits[dst + i] = h;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_search(hipStream_t s, const DevIndexView& ix, const uint8_t* bases, const uint32_t* read_off, uint32_t r0,
                   uint32_t n_reads, uint32_t max_ns, uint32_t K, uint32_t G, uint32_t* seed_lo, uint32_t* seed_cnt) {
    uint64_t total = (uint64_t)n_reads * 2 * max_ns;
    if (!total) return;
    hipLaunchKernelGGL(k_search, dim3(cdiv(total, 256)), dim3(256), 0, s, ix, bases, read_off, r0, n_reads, max_ns, K, G,
                       seed_lo, seed_cnt);
}

void launch_thin(hipStream_t s, const uint32_t* read_off, uint32_t r0, uint32_t n_reads, uint32_t max_ns, uint32_t K,
                 uint32_t G, uint64_t max_hits, uint64_t tune, uint32_t* seed_cnt, uint32_t* strand_hits,
                 uint32_t* strand_nseeds) {
    hipLaunchKernelGGL(k_thin, dim3(cdiv((uint64_t)n_reads * 2, 256)), dim3(256), 0, s, read_off, r0, n_reads, max_ns, K,
                       G, max_hits, tune, seed_cnt, strand_hits, strand_nseeds);
}

void launch_scan(hipStream_t s, const uint32_t* in, uint32_t n, uint64_t* tile_sums, uint64_t* total, uint32_t* out) {
    uint32_t tiles = cdiv(n ? n : 1, kScanTile);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(kScanThreads), 0, s, in, n, tile_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, tile_sums, tiles, total);
    hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(kScanThreads), 0, s, in, n, tile_sums, total, out);
}
uint32_t scan_tiles(uint32_t n) { return cdiv(n ? n : 1, kScanTile); }

void launch_expand(hipStream_t s, const DevIndexView& ix, uint32_t n_strands, uint32_t max_ns, uint32_t G,
                   const uint32_t* seed_lo, const uint32_t* seed_cnt, const uint32_t* strand_off, uint32_t* hit_row,
                   uint32_t* hit_ref, uint32_t* hit_q) {
    hipLaunchKernelGGL(k_expand, dim3(cdiv(n_strands, 256)), dim3(256), 0, s, ix, n_strands, max_ns, G, seed_lo, seed_cnt,
                       strand_off, hit_row, hit_ref, hit_q);
}

void launch_locate(hipStream_t s, const DevIndexView& ix, uint32_t total_hits_host, const uint32_t* total_hits_dev,
                   const uint32_t* hit_row, uint32_t* hit_ref, unsigned long long* lf_steps) {
    if (!total_hits_host) return;
    uint32_t chunks = cdiv(total_hits_host, kLocateChunk);
    uint32_t blocks = std::min<uint32_t>(cdiv(chunks, 4), 256 * 8);
    hipLaunchKernelGGL(k_locate, dim3(blocks), dim3(256), 0, s, ix, total_hits_dev, hit_row, hit_ref, lf_steps);
}

void launch_coalesce(hipStream_t s, const DevIndexView& ix, const uint32_t* read_off, uint32_t r0, uint32_t n_strands,
                     double edit_rate, double min_seed, const uint32_t* strand_off, const uint32_t* strand_nseeds,
                     const uint32_t* hit_ref, const uint32_t* hit_q, uint64_t* hit_key, uint64_t* cand_tmp, uint4* cand,
                     uint32_t* strand_ncand, uint32_t* worklist, uint32_t* wl_count, unsigned long long* n_cand_total) {
    uint32_t blocks = std::min<uint32_t>(cdiv(n_strands, 4), 256 * 8);
    hipLaunchKernelGGL(k_coalesce, dim3(blocks), dim3(256), 0, s, ix, read_off, r0, n_strands, edit_rate, min_seed,
                       strand_off, strand_nseeds, hit_ref, hit_q, hit_key, cand_tmp, cand, strand_ncand, worklist, wl_count,
                       n_cand_total);
}

void launch_verify(hipStream_t s, const DevIndexView& ix, const VerifyArgs& a, uint32_t n_strands) {
    hipLaunchKernelGGL(k_zero_u32, dim3(cdiv(n_strands, 256)), dim3(256), 0, s, a.strand_nout, n_strands);
    uint32_t blocks = std::min<uint32_t>(cdiv(n_strands, 4), 256 * 8);
    hipLaunchKernelGGL(k_verify, dim3(blocks), dim3(256), 0, s, ix, a);
}

void launch_gather(hipStream_t s, uint32_t n_strands, uint32_t r0, const uint32_t* strand_off, const uint32_t* strand_nout,
                   const uint32_t* out_off, const uint4* out, DevHit* hits, uint64_t hits_base) {
    hipLaunchKernelGGL(k_gather, dim3(cdiv(n_strands, 256)), dim3(256), 0, s, n_strands, r0, strand_off, strand_nout,
                       out_off, out, hits, hits_base);
}

}  // namespace mtsv

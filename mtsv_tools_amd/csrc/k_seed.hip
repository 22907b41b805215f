// k_seed.hip -- base normalisation, backward search, seed policy, scans, expand / locate (index.rs:284-352)
// (one of the three kernel files of the hot path; the stage map is in kernels.hpp / DESIGN.md section 3)
#include "kernels_common.hpp"

namespace mtsv {

namespace {

// ---------------------------------------------------------------------------------------------
// K0: base normalisation of the worker closure (binner.rs:88-100), src -> dst (may be the same buffer):
// A/a C/c G/g T/t -> codes 0..3, every other byte -> N (4).  16 bytes per lane.  Everything
// downstream reads codes; the reverse complement (binner.rs:115) is applied where a strand is read.
// ---------------------------------------------------------------------------------------------
__device__ inline uint32_t fast_code(uint32_t ch);
__global__ __launch_bounds__(256) void k_normalise(const uint8_t* src, uint8_t* dst, uint64_t lo, uint64_t hi) {
    // bytes [lo, hi) of the buffers; the 16-byte groups at the two edges may be shared with a neighbouring range that
    // another stream normalises in place at the same time, so only the range's own bytes of them are touched
    const uint64_t i = (lo & ~15ull) + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= hi) return;
    if (i < lo || i + 16 > hi) {
        const uint64_t a = i < lo ? lo : i, b = i + 16 > hi ? hi : i + 16;
        for (uint64_t q = a; q < b; q++) dst[q] = (uint8_t)fast_code(src[q]);
        return;
    }
    uint4 v = *reinterpret_cast<const uint4*>(src + i);  // both buffers are 16-byte aligned
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t o = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) o |= fast_code((w[k] >> (8 * q)) & 0xffu) << (8 * q);
        w[k] = o;
    }
    *reinterpret_cast<uint4*>(dst + i) = make_uint4(w[0], w[1], w[2], w[3]);
}

// the same for reads that came over PCIe as 4-bit codes (host_pack.cpp): base i of the segment is nibble (i & 1) of
// packed[i / 2]; bytes [lo, hi) of dst are written, sixteen per thread, the edge groups byte by byte as above
__global__ __launch_bounds__(256) void k_unpack(const uint8_t* __restrict__ packed, uint8_t* __restrict__ dst, uint64_t lo, uint64_t hi) {
    const uint64_t i = (lo & ~15ull) + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= hi) return;
    if (i < lo || i + 16 > hi) {
        const uint64_t a = i < lo ? lo : i, b = i + 16 > hi ? hi : i + 16;
        for (uint64_t q = a; q < b; q++) dst[q] = (uint8_t)((packed[q >> 1] >> (4 * (q & 1))) & 0xfu);
        return;
    }
    const uint2 v = *reinterpret_cast<const uint2*>(packed + (i >> 1));  // (i is a multiple of 16: 8-byte aligned)
    const uint32_t in[2] = {v.x, v.y};
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t h = (in[k >> 1] >> (16 * (k & 1))) & 0xffffu;  // four nibbles
        const uint32_t t = (h | (h << 8)) & 0x00ff00ffu;              // bytes 0 and 2: two nibbles each
        w[k] = (t | (t << 4)) & 0x0f0f0f0fu;
    }
    *reinterpret_cast<uint4*>(dst + i) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---------------------------------------------------------------------------------------------
// K1: backward search, one lane per seed slot
// ---------------------------------------------------------------------------------------------
// symbol code of an ASCII byte without a table: A/a C/c G/g T/t -> 0..3, anything else -> N (4)
__device__ inline uint32_t fast_code(uint32_t ch) {
    uint32_t uc = ch & 0xDFu;
    uint32_t x = (uc >> 1) & 3u;
    uint32_t c = x ^ (x >> 1);  // A0 C1 G2 T3
    bool acgt = uc == 'A' || uc == 'C' || uc == 'G' || uc == 'T';
    return acgt ? c : kCodeN;
}

// Seeds of up to 32 symbols are fetched with aligned dword loads (the read buffer is padded) and
// packed as 3-bit codes in strand order, so the search loop itself issues only rank-block loads.
constexpr uint32_t kMaxPackedSeed = 32;

// one seed slot with the general code: any seed size, with or without the k-mer table (also the path of the
// seeds whose table part holds an N, which k_search_fast hands over)
__device__ inline void search_slot(const DevIndexView& ix, const uint8_t* __restrict__ bases, const uint32_t* __restrict__ read_off,
                                   uint32_t r0, uint32_t max_ns, uint32_t K, uint32_t G, uint64_t slot,
                                   uint32_t* __restrict__ seed_lo, uint32_t* __restrict__ seed_cnt) {

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
        // packed codes: symbol i of the seed (strand order) at bits [3i, 3i+3) of c_lo (i < 21) / c_hi
        uint64_t c_lo = 0, c_hi = 0;
        const bool packed = K <= kMaxPackedSeed;
        // forward strand: bytes [off, off+K); reverse strand: bytes [L-off-K, L-off) reversed + complemented
        const uint32_t s0 = b0 + (strand ? L - off - K : off);
        const uint32_t* b32 = reinterpret_cast<const uint32_t*>(bases);
        const uint32_t w0 = s0 >> 2, sh = s0 & 3;
        if (K <= 21) {
            // up to 21 symbols fit one 64-bit word; four codes of a dword are squeezed to 12 bits with shifts
            // (codes < 8), the reverse strand is complemented bytewise and turned round by one 64-bit bit
            // reversal (after swapping bit 0 and bit 2 of every code, which the reversal swaps back)
            uint32_t d[7];
#pragma unroll
            for (int k = 0; k < 7; k++) d[k] = (uint32_t)(4 * k) < K + 4 ? b32[w0 + k] : 0u;
            uint64_t c = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                uint32_t w = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
                if (strand) {
                    const uint32_t m = (~w >> 2) & 0x01010101u;  // codes below 4: x -> 3 - x = x ^ 3
                    w ^= m | (m << 1);
                    const uint32_t t = (w ^ (w >> 2)) & 0x01010101u;
                    w ^= t | (t << 2);
                }
                const uint32_t x = (w | (w >> 5)) & 0x003f003fu;
                const uint32_t p12 = (x | (x >> 10)) & 0xfffu;
                if (12 * k < 64) c |= (uint64_t)p12 << (12 * k);
            }
            c &= (K < 21 ? (1ull << (3 * K)) : (1ull << 63)) - 1;
            if (strand) {
                const uint64_t rv = ((uint64_t)__builtin_bitreverse32((uint32_t)c) << 32) | __builtin_bitreverse32((uint32_t)(c >> 32));
                c = rv >> (64 - 3 * K);
            }
            c_lo = c;
        } else if (packed) {
            uint32_t d[9];
            const uint32_t nd = (K + 3) / 4;  // dwords that hold seed bytes (wave-uniform)
#pragma unroll
            for (int k = 0; k < 9; k++) d[k] = (uint32_t)k <= nd ? b32[w0 + k] : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if ((uint32_t)k >= nd) break;
                uint32_t w = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t bi = k * 4 + q;  // byte index inside the fetched span
                    uint32_t code = (w >> (8 * q)) & 0xffu;  // already a code (k_normalise)
                    if (strand) code = comp_code(code);
                    const uint32_t pos = strand ? (K - 1 - bi) : bi;  // strand-order position of this byte
                    if (bi < K) {
                        if (pos < 21) c_lo |= (uint64_t)code << (3 * pos);
                        else c_hi |= (uint64_t)code << (3 * (pos - 21));
                    }
                }
            }
        }
        auto sym = [&](uint32_t i) -> uint32_t {
            if (packed) return (uint32_t)((i < 21 ? c_lo >> (3 * i) : c_hi >> (3 * (i - 21))) & 7u);
            return strand_code(read, L, strand, off + i);
        };
        lo = 0;
        hi = ix.n;
        int i = (int)K - 1;
        // the seed's last kmer_k symbols in one gather when none of them is N
        if (ix.kmer_tab && K >= ix.kmer_k) {
            uint64_t idx = 0;  // (a table of 17 symbols has 2^34 entries)
            bool acgt = true;
            if (K <= 21 && ix.kmer_k <= 16) {
                const uint32_t kk = ix.kmer_k;  // <= 16
                const uint64_t sub = c_lo >> (3 * (K - kk));  // the last kk symbols, first of them lowest
                const uint32_t s_lo = (uint32_t)sub & 0x3fffffffu, s_hi = (uint32_t)(sub >> 30) & 0x3ffffu;
                acgt = ((s_lo & 0x24924924u) | (s_hi & 0x24924u)) == 0;  // no code has bit 2 set: all of A C G T
                uint32_t idx16 = 0;  // symbol t at bits [2(15-t), 2(15-t)+2): the table's order, first symbol highest
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const uint32_t a = t < 10 ? (s_lo >> (3 * t)) & 3u : (s_hi >> (3 * (t - 10))) & 3u;
                    idx16 |= a << (2 * (15 - t));
                }
                idx = kk >= 16 ? idx16 : idx16 >> (2 * (16 - kk));
            } else {
                for (uint32_t t = 0; t < ix.kmer_k; t++) {
                    uint32_t a = sym(K - ix.kmer_k + t);
                    acgt &= a < 4;
                    idx = (idx << 2) | (a & 3);
                }
            }
            if (acgt) {
                uint2 iv = ix.kmer_tab[idx];
                lo = iv.x;
                hi = iv.y;
                i = (int)K - 1 - (int)ix.kmer_k;
            }
        }
        for (; i >= 0 && lo < hi; i--) {
            uint32_t a = sym((uint32_t)i);
            uint32_t bl = lo >> kBlockShift, bh = hi >> kBlockShift;
            LoadedBlock B0 = load_block(ix.blocks, bl);
            uint32_t nlo = less_of(ix, a) + block_rank(B0, a, bl, lo & (kBlockRows - 1), ix.sentinel_row);
            uint32_t nhi;
            if (bh == bl) {
                nhi = less_of(ix, a) + block_rank(B0, a, bh, hi & (kBlockRows - 1), ix.sentinel_row);
            } else {
                LoadedBlock B1 = load_block(ix.blocks, bh);
                nhi = less_of(ix, a) + block_rank(B1, a, bh, hi & (kBlockRows - 1), ix.sentinel_row);
            }
            lo = nlo;
            hi = nhi;
        }
        if (lo >= hi) lo = hi = 0;  // Partial / Absent: only Complete intervals count (index.rs:312-332)
    }
    seed_lo[slot] = lo;
    seed_cnt[slot] = hi - lo;
}

__global__ __launch_bounds__(256) void k_search(DevIndexView ix, const uint8_t* __restrict__ bases,
                                                const uint32_t* __restrict__ read_off, uint32_t r0, uint32_t n_reads,
                                                uint32_t max_ns, uint32_t K, uint32_t G,
                                                uint32_t* __restrict__ seed_lo, uint32_t* __restrict__ seed_cnt) {
    uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)n_reads * 2 * max_ns;
    if (slot >= total) return;
    search_slot(ix, bases, read_off, r0, max_ns, K, G, slot, seed_lo, seed_cnt);
}

// The common case -- k-mer table of KK symbols resident, 16 <= K <= 24, at most 8 symbols left for the FM
// steps, fewer than 2^32 slots -- without any data-dependent trip count: the table index comes straight from
// the seed's code bytes (four codes of a dword squeezed to 8 bits; complementing is a bitwise NOT and the
// reverse strand's byte order already is the table's), then exactly K - KK rank steps.  Seeds with an N in
// the table part (about one in thirty; they walk up to K steps) are held back in LDS, per wavefront, and run through the
// general code 64 at a time: the wavefronts stride over the slots (a grid of resident workgroups), so a wavefront
// collects a full batch every ~1900 slots.  (One workgroup per 256 slots with its own queue walked ~9 such seeds on
// one wavefront per workgroup -- a quarter of the kernel's time, measured by leaving them out; a global list fed by one
// atomic per wavefront was atomic-bound: ~9 ns per single-address atomic, 2.5 M wavefronts with such a seed on config2.)
// KK = 17: the 16 packed symbols plus the one in front of them as bits 32-33 of the table index (a table of 2^34 entries).
// one seed slot on the table path: stores its interval, or returns true when the slot needs the general code (an N in the
// table part)
template <int KK>
__device__ inline bool fast_slot(const DevIndexView& ix, const uint8_t* __restrict__ bases, const uint32_t* __restrict__ read_off,
                                 uint32_t r0, uint32_t total, uint32_t max_ns, uint32_t K, uint32_t G, uint32_t slot,
                                 uint32_t* __restrict__ seed_lo, uint32_t* __restrict__ seed_cnt) {
    const bool in_range = slot < total;
    const uint32_t rs = in_range ? slot / max_ns : 0, j = in_range ? slot - rs * max_ns : 0;
    const uint32_t strand = rs & 1, r = r0 + (rs >> 1);
    const uint32_t b0 = read_off[r], L = read_off[r + 1] - b0;
    const bool live = in_range && j < n_seeds_of(L, K, G);
    uint32_t lo = 0, hi = 0;
    bool slow = false;
    if (live) {
        const uint32_t off = j * G;
        const uint32_t s0 = b0 + (strand ? L - off - K : off);  // first byte of the seed's span in the read buffer
        const uint32_t* b32 = reinterpret_cast<const uint32_t*>(bases);
        // 16 code bytes that hold the table part: the span's last 16 (forward) or first 16 (reverse strand)
        const uint32_t tp = strand ? s0 : s0 + K - 16;
        const uint32_t tw = tp >> 2, tsh = tp & 3;
        uint32_t d[5];
#pragma unroll
        for (int k = 0; k < 5; k++) d[k] = b32[tw + k];
        uint32_t packed = 0, nbits = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t w = __builtin_amdgcn_alignbyte(d[k + 1], d[k], tsh);
            nbits |= (w & 0x04040404u) >> 2 << k;  // byte q of dword k -> bit 8q + k: which of the 16 bytes is an N
            w &= 0x03030303u;                      // (an N outside the table part must not spill into its neighbour's field)
            const uint32_t t = (w | (w >> 6)) & 0x000f000fu;
            packed |= ((t | (t >> 12)) & 0xffu) << (8 * k);  // byte i of the 16 at bits [2i, 2i+2)
        }
        // the T bytes that count: the last T of the 16 (forward), the first T (reverse)
        constexpr int T = KK > 16 ? 16 : KK;
        uint32_t used = 0;  // bit 8q + k set when byte 4k + q is one of them
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool in = i >= 16 - T;  // forward strand; mirrored below for the reverse strand
            if (in) used |= 1u << (8 * (i & 3) + (i >> 2));
        }
        uint32_t used_rev = 0;
#pragma unroll
        for (int i = 0; i < T; i++) used_rev |= 1u << (8 * (i & 3) + (i >> 2));
        slow = (nbits & (strand ? used_rev : used)) != 0;
        if (!slow) {
            uint32_t idx;
            if (strand) {
                idx = ~packed;  // complement; byte 0 is the seed's last symbol: lowest bits, as the table wants
            } else {
                const uint32_t rv = __builtin_bitreverse32(packed);  // field order reversed, bits inside a field swapped
                idx = ((rv >> 1) & 0x55555555u) | ((rv & 0x55555555u) << 1);
            }
            if (T < 16) idx &= (1u << (2 * T)) - 1u;
            // the K - T symbols in front of the packed part, eight bytes at most
            const uint32_t fp = strand ? s0 + T : s0;
            const uint32_t fw = fp >> 2, fsh = fp & 3;
            const uint32_t f0 = b32[fw], f1 = b32[fw + 1], f2 = b32[fw + 2];
            const uint64_t fm = ((uint64_t)__builtin_amdgcn_alignbyte(f2, f1, fsh) << 32) | __builtin_amdgcn_alignbyte(f1, f0, fsh);
            // strand-order symbol i: span byte i (forward) or the complement of span byte K-1-i (reverse)
            auto front = [&](int i) {
                uint32_t a = (uint32_t)(fm >> (8 * (strand ? (int)K - 1 - i - T : i))) & 0xffu;
                return strand ? comp_code(a) : a;
            };
            int i = (int)K - T - 1;
            uint64_t tix = idx;
            if (KK > 16) {  // the table's 17th symbol
                const uint32_t a = front(i);
                slow = a > 3;  // N: the general code
                tix |= (uint64_t)(a & 3u) << 32;
                i--;
            }
            const uint2 iv = ix.kmer_tab[slow ? 0 : tix];
            lo = iv.x;
            hi = iv.y;
            for (; i >= 0; i--) {  // wave-uniform trip count
                const uint32_t a = front(i);
                if (lo < hi) {
                    const uint32_t bl = lo >> kBlockShift, bh = hi >> kBlockShift;
                    const LoadedBlock B0 = load_block(ix.blocks, bl);
                    const uint32_t nlo = less_of(ix, a) + block_rank(B0, a, bl, lo & (kBlockRows - 1), ix.sentinel_row);
                    uint32_t nhi;
                    if (bh == bl) {
                        nhi = less_of(ix, a) + block_rank(B0, a, bh, hi & (kBlockRows - 1), ix.sentinel_row);
                    } else {
                        const LoadedBlock B1 = load_block(ix.blocks, bh);
                        nhi = less_of(ix, a) + block_rank(B1, a, bh, hi & (kBlockRows - 1), ix.sentinel_row);
                    }
                    lo = nlo;
                    hi = nhi;
                }
            }
            if (lo >= hi) lo = hi = 0;  // Partial / Absent: only Complete intervals count (index.rs:312-332)
        }
    }
    if (in_range && !slow) {
        seed_lo[slot] = lo;
        seed_cnt[slot] = hi - lo;
    }
    return slow;
}

constexpr uint32_t kSlowCap = 128;  // slots a wavefront holds back in LDS before it appends 64 of them to the list

template <int KK>
__global__ __launch_bounds__(256) void k_search_fast(DevIndexView ix, const uint8_t* __restrict__ bases,
                                                     const uint32_t* __restrict__ read_off, uint32_t r0, uint32_t total,
                                                     uint32_t max_ns, uint32_t K, uint32_t G,
                                                     uint32_t* __restrict__ seed_lo, uint32_t* __restrict__ seed_cnt,
                                                     uint32_t* __restrict__ slow_list, uint32_t* __restrict__ slow_count) {
    __shared__ uint32_t slow_all[256 / kWave][kSlowCap];
    __shared__ uint32_t tail_q[256];
    __shared__ uint32_t tail_n;
    if (threadIdx.x == 0) tail_n = 0;
    const uint32_t lane = lane_id(), wave = threadIdx.x / kWave;
    uint32_t* slow_q = slow_all[wave];
    uint32_t n_slow = 0;  // (wave-uniform)
    const uint32_t n_tiles = (total + kWave - 1) / kWave;
    for (uint32_t tile = blockIdx.x * (256 / kWave) + wave; tile < n_tiles; tile += gridDim.x * (256 / kWave)) {
        const uint32_t slot = tile * kWave + lane;
        const bool slow = fast_slot<KK>(ix, bases, read_off, r0, total, max_ns, K, G, slot, seed_lo, seed_cnt);
        const unsigned long long sb = __ballot(slow);
        if (sb) {
            if (slow) slow_q[n_slow + __popcll(sb & ((1ull << lane) - 1))] = slot;
            n_slow += (uint32_t)__popcll(sb);
            if (n_slow >= kWave) {  // 64 of them leave for the list with one atomic
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
                n_slow -= kWave;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(slow_count, kWave);
                base = __builtin_amdgcn_readfirstlane(base);
                slow_list[base + lane] = slow_q[n_slow + lane];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // what the four wavefronts have left (fewer than 64 each) leaves with one atomic per workgroup
    __syncthreads();
    uint32_t at = 0;
    if (lane == 0 && n_slow) at = atomicAdd(&tail_n, n_slow);
    at = __builtin_amdgcn_readfirstlane(at);
    if (lane < n_slow) tail_q[at + lane] = slow_q[lane];
    __syncthreads();
    const uint32_t nt = tail_n;
    if (nt) {
        __shared__ uint32_t tail_base;
        if (threadIdx.x == 0) tail_base = atomicAdd(slow_count, nt);
        __syncthreads();
        if (threadIdx.x < nt) slow_list[tail_base + threadIdx.x] = tail_q[threadIdx.x];
    }
}

// the listed slots through the general code, every lane of a wavefront busy.  One entry per thread, no loop (a grid that
// strides over the device-side count keeps every field of the index view live across the loop: scalar-register spills):
// the host sizes the grid from the share of such slots in the passes before (batch.hip: it runs the pass again when the
// list turned out longer than the grid).
__global__ __launch_bounds__(256) void k_search_listed(DevIndexView ix, const uint8_t* __restrict__ bases,
                                                       const uint32_t* __restrict__ read_off, uint32_t r0, uint32_t max_ns, uint32_t K,
                                                       uint32_t G, const uint32_t* __restrict__ slow_list,
                                                       const uint32_t* __restrict__ slow_count, uint32_t* __restrict__ seed_lo,
                                                       uint32_t* __restrict__ seed_cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *slow_count) return;
    search_slot(ix, bases, read_off, r0, max_ns, K, G, slow_list[i], seed_lo, seed_cnt);
}

// ---------------------------------------------------------------------------------------------
// K1b: sequential seed policy of one strand (index.rs:293-344,354)
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_thin(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ read_off,
                                              uint32_t r0, uint32_t n_reads, double edit_rate, double min_seed,
                                              uint32_t max_ns, uint32_t K, uint32_t G, uint64_t max_hits,
                                              uint64_t tune_max_hits, uint32_t* __restrict__ seed_cnt,
                                              uint32_t* __restrict__ seed_pre, uint32_t* __restrict__ strand_hits,
                                              uint32_t* __restrict__ strand_nseeds) {
    uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rs >= n_reads * 2) return;
    uint32_t r = r0 + (rs >> 1);
    const uint32_t b0 = read_off[r];
    uint32_t L = read_off[r + 1] - b0;
    uint32_t ns = n_seeds_of(L, K, G);
    uint64_t next_offset = 0, seed_interval = G;
    uint32_t total = 0, nseeds = 0;
    uint32_t* cnt = seed_cnt + (uint64_t)rs * max_ns;
    uint32_t* pre = seed_pre + (uint64_t)rs * max_ns;
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
        pre[j] = total;  // hits of the strand's earlier kept seeds: k_expand's output offset (it reads the entries of kept seeds only)
        total += c;
        nseeds++;
    }
    {
        // the two strands of a read sit in neighbouring lanes (rs even / odd): the pair counts the read's N once, half of
        // the read each, and only when one of the strands has seed hits (only such strands can have candidates)
        const bool need = total != 0;
        const int other_need = __shfl_xor((int)need, 1);  // (unconditionally: both lanes of the pair take part in the exchange)
        const bool pair_need = need || other_need != 0;
        const uint32_t ED = (uint32_t)ceil((double)L * edit_rate);  // index.rs:281-282
        bool hopeless = 2ull * ED > (uint64_t)L;
        uint32_t nn = 0;
        if (pair_need && !hopeless) {
            // codes are 0..4, N = 4: bit 2 of every byte; aligned 16-byte groups of the code buffer (padded past its end).
            // The two lanes of the pair share the read's groups (even lane: even groups), and only the first and the last
            // group hold bytes of the neighbouring reads.
            const uint4* b128 = reinterpret_cast<const uint4*>(bases);
            const uint32_t q0 = b0 >> 4, q1 = (b0 + L + 15) >> 4;  // groups [q0, q1)
            auto masked = [&](uint32_t qi) {
                const uint4 v = b128[qi];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                uint32_t cnt = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t byte0 = qi * 16 + 4 * k;  // buffer position of this dword's first byte
                    uint32_t m = (w[k] >> 2) & 0x01010101u;
                    const int lo = (int)b0 - (int)byte0, hi = (int)(b0 + L) - (int)byte0;  // valid bytes [lo, hi) of this dword
                    if (lo > 0) m = lo >= 4 ? 0u : (m & (0xffffffffu << (8 * lo)));
                    if (hi < 4) m = hi <= 0 ? 0u : (m & ((1u << (8 * hi)) - 1u));
                    cnt += __popc(m);
                }
                return cnt;
            };
            const uint32_t half = rs & 1, ng = q1 - q0;  // group g of the read goes to the lane with (g & 1) == half
            if (ng >= 1 && half == 0) nn += masked(q0);
            if (ng >= 2 && half == ((ng - 1) & 1)) nn += masked(q1 - 1);
            for (uint32_t g = half ? 1 : 2; g + 1 < ng; g += 2) {  // the groups between the two: all bytes are the read's
                const uint4 v = b128[q0 + g];
                nn += __popc((v.x >> 2) & 0x01010101u) + __popc((v.y >> 2) & 0x01010101u) + __popc((v.z >> 2) & 0x01010101u) +
                      __popc((v.w >> 2) & 0x01010101u);
            }
        }
        nn += (uint32_t)__shfl_xor((int)nn, 1);  // the pair's two halves (every lane takes part in the exchange)
        hopeless = hopeless || nn > ED;
        // what the coalescing kernels need of this strand, in one word: min_seeds = max(1, floor(n_seeds * pct))
        // (index.rs:358; saturated at 16 bits, above any candidate's seed count), the edit tolerance, the flag
        const double ms = floor((double)nseeds * min_seed);
        const uint32_t min_seeds = ms < 1.0 ? 1u : (ms > 65535.0 ? 65535u : (uint32_t)ms);
        nseeds = min_seeds | (ED << 16) | ((need && hopeless) ? kHopeless : 0u);
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
__global__ __launch_bounds__(256) void k_expand(DevIndexView ix, uint64_t n_slots, uint32_t max_ns, uint32_t G,
                                                const uint32_t* __restrict__ seed_lo,
                                                const uint32_t* __restrict__ seed_cnt,
                                                const uint32_t* __restrict__ seed_pre,
                                                const uint32_t* __restrict__ strand_off,
                                                uint32_t* __restrict__ hit_row, uint32_t* __restrict__ hit_ref,
                                                uint32_t* __restrict__ hit_q) {
    uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = slot < n_slots ? seed_cnt[slot] : 0;
    uint32_t o = 0, l = 0, q = 0;
    if (c) {
        const uint32_t rs = (uint32_t)(slot / max_ns), j = (uint32_t)(slot % max_ns);
        o = strand_off[rs] + seed_pre[slot];
        l = seed_lo[slot];
        q = j * G;
    }
    // seeds with few hits: the lane writes them itself; repeats (up to max_hits per seed) are spread over the
    // wavefront, one seed after the other, so that no single lane walks thousands of entries
    constexpr uint32_t kOwn = 16;
    if (c && c <= kOwn) {
        for (uint32_t i = 0; i < c; i++) {
            if (ix.sa_full) hit_ref[o + i] = ix.sa_full[l + i];
            else hit_row[o + i] = l + i;
            hit_q[o + i] = q;
        }
    }
    unsigned long long big = __ballot(c > kOwn);
    const uint32_t lane = lane_id();
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t cs = __builtin_amdgcn_readlane(c, src), os = __builtin_amdgcn_readlane(o, src);
        const uint32_t ls = __builtin_amdgcn_readlane(l, src), qs = __builtin_amdgcn_readlane(q, src);
        for (uint32_t i = lane; i < cs; i += kWave) {
            if (ix.sa_full) hit_ref[os + i] = ix.sa_full[ls + i];
            else hit_row[os + i] = ls + i;
            hit_q[os + i] = qs;
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
                        row = less_of(ix, c) + block_rank(b, c, blk, off, ix.sentinel_row);
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

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
namespace {
// a few counters from HBM into mapped page-locked host memory (visible to the host once the stream is synchronised)
__global__ void k_publish(const uint64_t* __restrict__ src, uint64_t* dst, uint32_t n) {
    if (threadIdx.x < n) __hip_atomic_store(dst + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

void launch_publish(hipStream_t s, const uint64_t* src, uint64_t* dst_host, uint32_t n) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, src, dst_host, n);
}

void launch_normalise(hipStream_t s, const uint8_t* src, uint8_t* dst, uint64_t begin, uint64_t end) {
    if (end <= begin) return;
    const uint64_t n = end - (begin & ~15ull);  // 16-byte groups, the first one aligned
    hipLaunchKernelGGL(k_normalise, dim3(cdiv((n + 15) / 16, 256)), dim3(256), 0, s, src, dst, begin, end);
}
void launch_unpack(hipStream_t s, const uint8_t* packed, uint8_t* dst, uint64_t lo, uint64_t hi) {
    if (hi <= lo) return;
    const uint64_t groups = ((hi + 15) >> 4) - (lo >> 4);
    hipLaunchKernelGGL(k_unpack, dim3((uint32_t)cdiv(groups, 256)), dim3(256), 0, s, packed, dst, lo, hi);
}

void launch_search(hipStream_t s, const DevIndexView& ix, const uint8_t* bases, const uint32_t* read_off, uint32_t r0,
                   uint32_t n_reads, uint32_t max_ns, uint32_t K, uint32_t G, uint32_t* seed_lo, uint32_t* seed_cnt,
                   uint32_t* slow_list, uint32_t* slow_count, uint32_t listed_cap) {
    (void)hipMemsetAsync(slow_count, 0, sizeof(uint32_t), s);  // (whichever path is taken: the caller reads it after every pass)
    uint64_t total = (uint64_t)n_reads * 2 * max_ns;
    if (!total) return;
    const bool fast = ix.kmer_tab && total < 0xffffffffull && K >= 16 && K <= 24 && ix.kmer_k >= 12 && ix.kmer_k <= 17 &&
                      K >= ix.kmer_k && K - std::min<uint32_t>(ix.kmer_k, 16) <= 8 && !getenv("MTSV_SEARCH_GENERIC");
    if (!fast) {
        hipLaunchKernelGGL(k_search, dim3(cdiv(total, 256)), dim3(256), 0, s, ix, bases, read_off, r0, n_reads, max_ns, K, G,
                           seed_lo, seed_cnt);
        return;
    }
    // resident wavefronts that stride over the slots (a wavefront fills a batch of 64 held-back slots every ~1900)
    // (two generations of resident workgroups: 0.81 ms per pass of 1 Mi reads, 0.84 with one, 0.82 with half of one)
    const dim3 grid(std::min<uint32_t>(cdiv(total, 256), 256 * 16));
#define FAST_CASE(KKV)                                                                                                  \
    hipLaunchKernelGGL((k_search_fast<KKV>), grid, dim3(256), 0, s, ix, bases, read_off, r0, (uint32_t)total, max_ns, K, G, seed_lo, \
                       seed_cnt, slow_list, slow_count)
    switch (ix.kmer_k) {
    case 12: FAST_CASE(12); break;
    case 13: FAST_CASE(13); break;
    case 14: FAST_CASE(14); break;
    case 15: FAST_CASE(15); break;
    case 17: FAST_CASE(17); break;
    default: FAST_CASE(16); break;
    }
#undef FAST_CASE
    // (about one slot in thirty with the synthetic reads' 0.2 % of N)
    hipLaunchKernelGGL(k_search_listed, dim3(cdiv(std::max<uint32_t>(listed_cap, 1), 256)), dim3(256), 0, s, ix, bases, read_off, r0, max_ns, K, G,
                       slow_list, slow_count, seed_lo, seed_cnt);
}

void launch_thin(hipStream_t s, const uint8_t* bases, const uint32_t* read_off, uint32_t r0, uint32_t n_reads, double edit_rate,
                 double min_seed, uint32_t max_ns, uint32_t K, uint32_t G, uint64_t max_hits, uint64_t tune, uint32_t* seed_cnt, uint32_t* seed_pre,
                 uint32_t* strand_hits, uint32_t* strand_nseeds) {
    hipLaunchKernelGGL(k_thin, dim3(cdiv((uint64_t)n_reads * 2, 256)), dim3(256), 0, s, bases, read_off, r0, n_reads, edit_rate,
                       min_seed, max_ns, K, G, max_hits, tune, seed_cnt, seed_pre, strand_hits, strand_nseeds);
}

void launch_scan(hipStream_t s, const uint32_t* in, uint32_t n, uint64_t* tile_sums, uint64_t* total, uint32_t* out) {
    uint32_t tiles = cdiv(n ? n : 1, kScanTile);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(kScanThreads), 0, s, in, n, tile_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, tile_sums, tiles, total);
    hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(kScanThreads), 0, s, in, n, tile_sums, total, out);
}
uint32_t scan_tiles(uint32_t n) { return cdiv(n ? n : 1, kScanTile); }

void launch_expand(hipStream_t s, const DevIndexView& ix, uint32_t n_strands, uint32_t max_ns, uint32_t G,
                   const uint32_t* seed_lo, const uint32_t* seed_cnt, const uint32_t* seed_pre, const uint32_t* strand_off,
                   uint32_t* hit_row, uint32_t* hit_ref, uint32_t* hit_q) {
    uint64_t n_slots = (uint64_t)n_strands * max_ns;
    if (!n_slots) return;
    hipLaunchKernelGGL(k_expand, dim3(cdiv(n_slots, 256)), dim3(256), 0, s, ix, n_slots, max_ns, G, seed_lo, seed_cnt, seed_pre,
                       strand_off, hit_row, hit_ref, hit_q);
}

void launch_locate(hipStream_t s, const DevIndexView& ix, uint32_t total_hits_host, const uint32_t* total_hits_dev,
                   const uint32_t* hit_row, uint32_t* hit_ref, unsigned long long* lf_steps) {
    if (!total_hits_host) return;
    uint32_t chunks = cdiv(total_hits_host, kLocateChunk);
    uint32_t blocks = std::min<uint32_t>(cdiv(chunks, 4), 256 * 8);
    hipLaunchKernelGGL(k_locate, dim3(blocks), dim3(256), 0, s, ix, total_hits_dev, hit_row, hit_ref, lf_steps);
}

}  // namespace mtsv

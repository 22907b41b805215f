// dev_layout.hpp -- HBM layout of the MG-index and the device-side rank primitive.
//
// The file's FM-index (byte-per-symbol BWT + eleven u64 Occ arrays sampled every k,
// src/index.rs:560-575) is re-packed at upload into 64-byte rank blocks covering 128 BWT rows:
//
//     uint32 cnt[4]   occurrences of A,C,G,T in rows [0, 128*blk)            16 B
//     uint64 p0[2]    bit 0 of the 3-bit symbol code of each of the 128 rows 16 B
//     uint64 p1[2]    bit 1                                                  16 B
//     uint64 p2[2]    bit 2                                                  16 B
//
// codes: A=0 C=1 G=2 T=3 N=4 $=5 (7 = padding past row n-1, matches nothing).  The N count is not
// stored: rows before the block minus A+C+G+T minus the sentinel if it lies before the block.
// One rank query = one aligned 64 B load + 2 popcounts; 0.5 byte per symbol instead of the
// reference's 1 + 11*8/k.  Positions are u32 on the device (n < 2^32).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mtsv {

constexpr uint32_t kCodeA = 0, kCodeC = 1, kCodeG = 2, kCodeT = 3, kCodeN = 4, kCodeSentinel = 5;
constexpr uint32_t kBlockRows = 128;
constexpr uint32_t kBlockShift = 7;

struct alignas(64) RankBlock {
    uint32_t cnt[4];
    uint64_t p0[2];
    uint64_t p1[2];
    uint64_t p2[2];
};
static_assert(sizeof(RankBlock) == 64, "rank block must be one 64-byte line");

struct DevBin {  // src/index.rs:45-54 narrowed to u32 positions
    uint32_t start, end, tax_id, gi;
};

// Passed by value to every kernel.
struct DevIndexView {
    const RankBlock* blocks;
    uint32_t n;             // BWT rows (text symbols incl. '$')
    uint32_t n_blocks;      // (n >> 7) + 1: rank(a, n) must be addressable
    uint32_t C[5];          // less[] of A,C,G,T,N
    uint32_t sentinel_row;  // row whose BWT symbol is '$'
    const uint32_t* sa_sample;  // sample[j] = SA[j*s]  (the file's row-sampled SA)
    uint32_t sa_s;
    uint32_t sa_pow2_shift;     // log2(s) if s is a power of two, else 0xffffffff
    const uint32_t* sa_full;    // full SA reconstructed in HBM at upload (nullptr if disabled)
    const uint8_t* text;        // symbol codes of `sequences`, one per byte
    const uint32_t* bin_end;    // bins[i].end, ascending (binary-search key)
    const DevBin* bins;
    uint32_t n_bins;
    const uint32_t* bin_lut;    // bin_lut[p >> bin_lut_shift] = first bin whose end > (p >> shift) << shift
    uint32_t bin_lut_shift;
    const uint2* kmer_tab;      // [4^kmer_k] SA interval (lo, hi) of every ACGT k-mer, or nullptr
    uint32_t kmer_k;
};

// binner.rs:88-100 as symbol codes: A/a C/c G/g T/t, everything else (incl. N/n) -> N
__host__ __device__ inline uint32_t base_code(uint8_t b) {
    switch (b) {
    case 'A': case 'a': return kCodeA;
    case 'C': case 'c': return kCodeC;
    case 'G': case 'g': return kCodeG;
    case 'T': case 't': return kCodeT;
    default: return kCodeN;
    }
}
// bio::alphabets::dna::revcomp on the normalised alphabet: A<->T, C<->G, N->N
__host__ __device__ inline uint32_t comp_code(uint32_t c) { return c < 4 ? 3 - c : c; }

#ifdef __HIPCC__
struct LoadedBlock {
    uint4 h, q0, q1, q2;
};

__device__ inline LoadedBlock load_block(const RankBlock* blocks, uint32_t blk) {
    const uint4* p = reinterpret_cast<const uint4*>(blocks + blk);
    LoadedBlock b;
    b.h = p[0];
    b.q0 = p[1];
    b.q1 = p[2];
    b.q2 = p[3];
    return b;
}

__device__ inline uint64_t u64_of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// symbol code stored at in-block offset off (0..127)
__device__ inline uint32_t block_code(const LoadedBlock& b, uint32_t off) {
    // 64-bit select + shift: keeps the block in registers (indexing its words would go to scratch)
    bool hi = off & 64;
    uint32_t sh = off & 63;
    uint64_t w0 = hi ? u64_of(b.q0.z, b.q0.w) : u64_of(b.q0.x, b.q0.y);
    uint64_t w1 = hi ? u64_of(b.q1.z, b.q1.w) : u64_of(b.q1.x, b.q1.y);
    uint64_t w2 = hi ? u64_of(b.q2.z, b.q2.w) : u64_of(b.q2.x, b.q2.y);
    return (uint32_t)((w0 >> sh) & 1) | ((uint32_t)((w1 >> sh) & 1) << 1) | ((uint32_t)((w2 >> sh) & 1) << 2);
}

// #rows p < pos with bwt[p] == a, for pos = blk*128 + off, a in 0..4 (exclusive rank; bio's
// inclusive Occ::get(r, a) is rank(a, r+1))
__device__ inline uint32_t block_rank(const LoadedBlock& b, uint32_t a, uint32_t blk, uint32_t off,
                                      uint32_t sentinel_row) {
    uint64_t x0 = (a & 1) ? 0ull : ~0ull, x1 = (a & 2) ? 0ull : ~0ull, x2 = (a & 4) ? 0ull : ~0ull;
    uint64_t ma = (u64_of(b.q0.x, b.q0.y) ^ x0) & (u64_of(b.q1.x, b.q1.y) ^ x1) &
                  (u64_of(b.q2.x, b.q2.y) ^ x2);
    uint64_t mb = (u64_of(b.q0.z, b.q0.w) ^ x0) & (u64_of(b.q1.z, b.q1.w) ^ x1) &
                  (u64_of(b.q2.z, b.q2.w) ^ x2);
    uint64_t maskA = off >= 64 ? ~0ull : ((1ull << off) - 1);
    uint64_t maskB = off > 64 ? ((1ull << (off - 64)) - 1) : 0ull;
    uint32_t base;
    if (a < 4) {
        base = a == 0 ? b.h.x : a == 1 ? b.h.y : a == 2 ? b.h.z : b.h.w;
    } else {
        uint32_t before = blk << kBlockShift;
        base = before - (b.h.x + b.h.y + b.h.z + b.h.w) - (before > sentinel_row ? 1u : 0u);
    }
    return base + __popcll(ma & maskA) + __popcll(mb & maskB);
}

// less[] of symbol code a without dynamic indexing of the by-value view (that would spill C[] to scratch)
__device__ inline uint32_t less_of(const DevIndexView& ix, uint32_t a) {
    return a == 0 ? ix.C[0] : a == 1 ? ix.C[1] : a == 2 ? ix.C[2] : a == 3 ? ix.C[3] : ix.C[4];
}

__device__ inline uint32_t dev_rank(const DevIndexView& ix, uint32_t a, uint32_t pos) {
    uint32_t blk = pos >> kBlockShift;
    LoadedBlock b = load_block(ix.blocks, blk);
    return block_rank(b, a, blk, pos & (kBlockRows - 1), ix.sentinel_row);
}
#endif  // __HIPCC__

}  // namespace mtsv

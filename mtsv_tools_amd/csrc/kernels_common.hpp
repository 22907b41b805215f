// kernels_common.hpp -- device helpers shared by the kernel files of the hot path (k_seed.hip: normalise, search,
// thin, expand, locate; k_coalesce.hip; k_verify.hip: SW prefilter, edit distance, selection, gather).
#pragma once
#include <cstdlib>
#include <stdexcept>

#include "kernels.hpp"

namespace mtsv {
namespace {

constexpr int kWave = 64;

__device__ inline uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

// ---------------------------------------------------------------------------------------------
// strand access: symbol code of position p of strand `strand` of a read (binner.rs:88-100,115)
// ---------------------------------------------------------------------------------------------
// (the read buffer holds symbol codes: k_normalise has run over it)
__device__ inline uint32_t strand_code(const uint8_t* __restrict__ read, uint32_t L, uint32_t strand, uint32_t p) {
    return strand ? comp_code(read[L - 1 - p]) : (uint32_t)read[p];
}

__device__ inline uint32_t n_seeds_of(uint32_t L, uint32_t K, uint32_t G) {
    // offsets 0, G, 2G, ... < L + 1 - K  (index.rs:284-286); L + 1 < K is trapped as "no seeds"
    return (L >= K) ? (L - K) / G + 1 : 0;
}

// A strand is HOPELESS when no candidate of it can be accepted whatever its window holds: the usize wrap of
// index.rs:406 (2*ED > L: the threshold L - 2*ED wraps and no score reaches it), or more N in the read than
// the edit tolerance (a read N never matches in the edit-distance recurrence, index.rs:272-279, so
// edits >= #N > ED fails :410).  The reference still runs its prefilter on every candidate of such a strand;
// here the coalescing kernels account that work (counters) and emit no work items for it.  The flag
// travels in the top bit of strand_nseeds.
constexpr uint32_t kHopeless = 0x80000000u;

// sum over the 16 lanes of a DPP row, result in every lane
__device__ inline int row_sum16(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    return v;
}

}  // namespace

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
// --max-candidates as a bound on candidate ranks (index.rs:385-389): None = no bound
static inline uint32_t rank_bound(int64_t max_candidates) {
    return max_candidates < 0 ? 0xffffffffu : (max_candidates > 0xfffffffeLL ? 0xfffffffeu : (uint32_t)max_candidates);
}

}  // namespace mtsv

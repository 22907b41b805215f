// Host side of run_host's transfer format (host_pack.cpp; plain C++, no HIP types).
#pragma once
#include <cstdint>
#include <functional>

namespace mtsv {
// bases src[0, n) that lie at segment offsets [a, a + n), as 4-bit codes (binner.rs:88-100: A/a C/c G/g T/t -> 0..3, anything
// else -> 4, what k_normalise does on the device for the resident path) into dst = bytes [a / 2, (a + n + 1) / 2) of the
// segment's packed image: base i in nibble (i & 1) of byte i / 2.  prev_code: the code of the base at offset a - 1 (it shares
// the first byte when a is odd).  Returns the code of the last base.  Split over pack_threads() threads.
uint8_t pack_chunk(uint8_t* dst, const uint8_t* src, uint64_t a, uint64_t n, uint8_t prev_code);
// threads pack_chunk uses, the caller included (MTSV_PACK_THREADS; else from the CPUs the process may use).  Fewer than
// kPackWorthwhile of them are slower than the copy engine on the plain bytes: run_host then sends those.
int pack_threads();
// fn(p0, p1) over [0, n) in pieces of `piece`, on the packer's threads and the caller's; returns when all are done
void pack_pool_for(uint64_t n, uint64_t piece, const std::function<void(uint64_t, uint64_t)>& fn);
constexpr int kPackWorthwhile = 9;
}  // namespace mtsv

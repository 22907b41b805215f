// mgindex.hpp -- host-side MG-index: the data model of src/index.rs:20-68 as it sits on disk.
//
// On-disk codec = bincode 1.3.3 default options (src/io.rs:121,131): little-endian fixed-width
// integers, usize -> u64, Vec<T> -> u64 length + elements, struct -> fields in declaration
// order, HashMap -> u64 length + (key, value) pairs.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace mtsv {

struct Bin {  // src/index.rs:45-54, field order as serialised
    uint32_t gi;
    uint32_t tax_id;
    uint64_t start;
    uint64_t end;
};

constexpr uint64_t kOccOuter = 117;  // n_alphabet() max symbol 't' (116) + 1
constexpr uint64_t kLessLen = 118;

// MGIndex { sequences, bins, suffix_array: SampledSuffixArray { bwt, less, occ{occ,k}, sample, s,
// extra_rows, sentinel } }
struct HostIndex {
    std::vector<uint8_t> text;  // sequences: A C G T N ... '$'
    std::vector<Bin> bins;
    std::vector<uint8_t> bwt;  // one ASCII symbol per byte
    std::vector<uint64_t> less;
    std::vector<std::vector<uint64_t>> occ;  // [symbol][checkpoint]
    uint32_t k = 0;
    std::vector<uint64_t> sample;  // sample[j] = SA[j*s]
    uint64_t s = 0;
    std::vector<std::pair<uint64_t, uint64_t>> extra_rows;
    uint8_t sentinel = '$';
    uint64_t file_bytes = 0;

    uint64_t n() const { return text.size(); }
};

// Throws std::runtime_error (message prefixed "io:" or "format:") on failure.
void load_index(const std::string& path, HostIndex& ix);
void write_index(const HostIndex& ix, const std::string& path);
// Structural invariants every mtsv-build output satisfies; throws "format: ..." otherwise.
void validate_index(const HostIndex& ix);

struct SeqEntry {
    uint32_t tax_id;
    uint32_t gi;
    const uint8_t* seq;
    uint64_t len;
};
// MGIndex::new (src/index.rs:491-582)
// gpu_device >= 0: suffix array / BWT / samples are built on that HIP device (gpu_builder.hip)
void build_index(std::vector<SeqEntry> entries, uint32_t occ_k, uint64_t sa_s, int n_threads,
                 HostIndex& out, int gpu_device = -1);
// parse_fasta_db (src/io.rs:135-150) + parse_read_header (src/util.rs:26-56) + build_index
void build_index_from_fasta(const std::string& fasta, uint32_t occ_k, uint64_t sa_s, int n_threads,
                            HostIndex& out, int gpu_device = -1);

}  // namespace mtsv

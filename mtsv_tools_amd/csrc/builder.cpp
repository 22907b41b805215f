// builder.cpp -- MGIndex::new (src/index.rs:491-582) and build_and_write_index
// (src/builder.rs:14-36): concatenate sequences in ascending TaxId order, normalise to DNA5,
// append '$', suffix array -> BWT -> less -> Occ(k) -> row-sampled SA(s).
//
// The suffix array is built by a multi-threaded two-level bucket sort (prefix bucket by the first
// P symbols, then a 21-symbol packed key, then direct comparison), which is what the host can do
// quickly for the near-random texts of benchmark databases.  Any correct suffix array yields the
// same index bytes.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <array>
#include <thread>

#include "mgindex.hpp"

namespace mtsv {

namespace {

// lexicographic rank of the six index symbols: $ < A < C < G < N < T
inline uint8_t sym_rank(uint8_t c) {
    switch (c) {
    case '$': return 0;
    case 'A': return 1;
    case 'C': return 2;
    case 'G': return 3;
    case 'N': return 4;
    default: return 5;  // 'T'
    }
}

template <class F>
void parallel_for(uint64_t n, int threads, F fn) {
    if (threads <= 1 || n < 2) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    uint64_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        uint64_t lo = std::min(n, (uint64_t)t * chunk), hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        pool.emplace_back(fn, lo, hi, t);
    }
    for (auto& th : pool) th.join();
}

struct SuffixSorter {
    const uint8_t* text;
    uint64_t n;
    int threads;

    // compare suffixes i and j from depth d on (both known equal before d)
    bool less_from(uint32_t i, uint32_t j, uint64_t d) const {
        uint64_t li = n - i, lj = n - j;
        if (d >= li || d >= lj) return li < lj;  // unreachable with a unique sentinel
        uint64_t l = std::min(li, lj) - d;
        int c = memcmp(text + i + d, text + j + d, l);
        // byte order of "$ACGNT" equals the rank order, so memcmp is the suffix order
        if (c) return c < 0;
        return li < lj;
    }

    // 21 symbols starting at text[p], 3 bits each, most significant first; past the end -> 0
    uint64_t key21(uint64_t p) const {
        uint64_t k = 0;
        uint64_t end = std::min(n, p + 21);
        uint64_t i = p;
        for (; i < end; i++) k = (k << 3) | sym_rank(text[i]);
        k <<= 3 * (p + 21 - i);
        return k;
    }

    void sort(std::vector<uint32_t>& sa) {
        // prefix length P: ~n/64 populated buckets, 6^P table entries
        int P = 2;
        uint64_t pop = 16;  // 4^P
        while (P < 10 && pop * 64 < n) {
            P++;
            pop *= 4;
        }
        uint64_t nb = 1;
        for (int i = 0; i < P; i++) nb *= 6;
        std::vector<std::atomic<uint32_t>> cursor(nb + 1);
        for (auto& c : cursor) c.store(0, std::memory_order_relaxed);

        auto key_at = [&](uint64_t i) {
            uint64_t k = 0;
            for (int j = 0; j < P; j++) k = k * 6 + (i + j < n ? sym_rank(text[i + j]) : 0);
            return k;
        };
        // histogram
        parallel_for(n, threads, [&](uint64_t lo, uint64_t hi, int) {
            for (uint64_t i = lo; i < hi; i++) cursor[key_at(i)].fetch_add(1, std::memory_order_relaxed);
        });
        std::vector<uint64_t> start(nb + 1);
        uint64_t acc = 0;
        for (uint64_t b = 0; b < nb; b++) {
            start[b] = acc;
            acc += cursor[b].load(std::memory_order_relaxed);
            cursor[b].store((uint32_t)start[b], std::memory_order_relaxed);
        }
        start[nb] = acc;
        sa.resize(n);
        // scatter (order inside a bucket is fixed by the sort below)
        parallel_for(n, threads, [&](uint64_t lo, uint64_t hi, int) {
            for (uint64_t i = lo; i < hi; i++) {
                uint32_t at = cursor[key_at(i)].fetch_add(1, std::memory_order_relaxed);
                sa[at] = (uint32_t)i;
            }
        });
        // sort every bucket; buckets handed out dynamically in blocks
        std::atomic<uint64_t> next(0);
        const uint64_t grain = 256;
        auto worker = [&]() {
            std::vector<std::pair<uint64_t, uint32_t>> tmp;
            for (;;) {
                uint64_t b0 = next.fetch_add(grain);
                if (b0 >= nb) break;
                uint64_t b1 = std::min(nb, b0 + grain);
                for (uint64_t b = b0; b < b1; b++) {
                    uint64_t lo = start[b], hi = start[b + 1];
                    if (hi - lo < 2) continue;
                    tmp.resize(hi - lo);
                    for (uint64_t i = lo; i < hi; i++) tmp[i - lo] = {key21((uint64_t)sa[i] + P), sa[i]};
                    std::sort(tmp.begin(), tmp.end(), [&](const auto& x, const auto& y) {
                        if (x.first != y.first) return x.first < y.first;
                        return less_from(x.second, y.second, (uint64_t)P + 21);
                    });
                    for (uint64_t i = lo; i < hi; i++) sa[i] = tmp[i - lo].second;
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }
};

}  // namespace

// gpu_builder.hip
void gpu_suffix_sort(const uint8_t* text, uint32_t n, uint64_t s, int device, std::vector<uint8_t>& bwt,
                     std::vector<uint64_t>& sample, uint64_t* sentinel_row);

void build_index(std::vector<SeqEntry> entries, uint32_t occ_k, uint64_t sa_s, int n_threads,
                 HostIndex& ix, int gpu_device) {
    if (occ_k == 0 || sa_s == 0) throw std::runtime_error("arg: sampling intervals must be > 0");
    if (n_threads < 1) n_threads = 1;
    ix = HostIndex();
    // BTreeMap<TaxId, Vec<(Gi, Sequence)>>: ascending TaxId, insertion order inside (index.rs:497)
    std::stable_sort(entries.begin(), entries.end(),
                     [](const SeqEntry& a, const SeqEntry& b) { return a.tax_id < b.tax_id; });
    uint64_t total = 0;
    for (auto& e : entries) total += e.len;
    if (total + 1 >= (1ull << 32)) throw std::runtime_error("limit: text of 2^32 symbols or more; split the database into chunks");
    ix.text.resize(total + 1);
    ix.bins.reserve(entries.size());
    uint64_t pos = 0;
    for (auto& e : entries) {
        ix.bins.push_back(Bin{e.gi, e.tax_id, pos, pos + e.len});
        if (e.len) memcpy(ix.text.data() + pos, e.seq, e.len);
        pos += e.len;
    }
    // DNA5 normalisation (index.rs:543-553)
    parallel_for(total, n_threads, [&](uint64_t lo, uint64_t hi, int) {
        for (uint64_t i = lo; i < hi; i++) {
            uint8_t b = ix.text[i];
            switch (b) {
            case 'A': case 'C': case 'G': case 'T': case 'N': break;
            case 'a': b = 'A'; break;
            case 'c': b = 'C'; break;
            case 'g': b = 'G'; break;
            case 't': b = 'T'; break;
            default: b = 'N';
            }
            ix.text[i] = b;
        }
    });
    ix.text[total] = '$';  // index.rs:555
    ix.sentinel = '$';
    const uint64_t n = total + 1;

    ix.s = sa_s;
    uint64_t sentinel_row = 0;
    if (gpu_device >= 0) {
        // suffix array, BWT and samples on the GPU (gpu_builder.hip); only bwt + samples come back
        gpu_suffix_sort(ix.text.data(), (uint32_t)n, sa_s, gpu_device, ix.bwt, ix.sample, &sentinel_row);
    } else {
        std::vector<uint32_t> sa;
        SuffixSorter{ix.text.data(), n, n_threads}.sort(sa);
        // bwt (index.rs:567)
        ix.bwt.resize(n);
        parallel_for(n, n_threads, [&](uint64_t lo, uint64_t hi, int) {
            for (uint64_t i = lo; i < hi; i++) ix.bwt[i] = sa[i] ? ix.text[sa[i] - 1] : ix.text[n - 1];
        });
        // sample (index.rs:574)
        ix.sample.resize((n + sa_s - 1) / sa_s);
        for (uint64_t i = 0, j = 0; i < n; i += sa_s, j++) ix.sample[j] = sa[i];
        for (uint64_t i = 0; i < n; i++)
            if (sa[i] == 0) {
                sentinel_row = i;
                break;
            }
    }
    // a non-sampled row whose BWT symbol is the sentinel goes to extra_rows; it is the suffix at 0
    if (sentinel_row % sa_s != 0) ix.extra_rows.push_back({sentinel_row, 0});

    // less (index.rs:570) and Occ::new (index.rs:571): checkpoint j = inclusive counts in bwt[0..=j*k];
    // two passes over thread ranges aligned to k
    ix.k = occ_k;
    ix.occ.assign(kOccOuter, {});
    const uint64_t nchk = (n - 1) / occ_k + 1;
    static const uint8_t alpha[11] = {'A', 'C', 'G', 'T', 'N', 'a', 'c', 'g', 't', 'n', '$'};
    for (uint8_t a : alpha) ix.occ[a].assign(nchk, 0);
    const int T = n_threads;
    const uint64_t chk_per = (nchk + T - 1) / T;  // checkpoints per thread; thread t owns rows (c0*k - k, c1*k - k] shifted below
    std::vector<std::array<uint64_t, 256>> part(T);
    auto row_lo = [&](int t) { uint64_t c0 = std::min(nchk, (uint64_t)t * chk_per); return c0 == 0 ? 0 : (c0 - 1) * occ_k + 1; };
    auto row_hi = [&](int t) { uint64_t c1 = std::min(nchk, (uint64_t)(t + 1) * chk_per); return t == T - 1 ? n : (c1 == 0 ? 0 : std::min<uint64_t>(n, (c1 - 1) * occ_k + 1)); };
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; t++)
            pool.emplace_back([&, t] {
                std::array<uint64_t, 256> c{};
                for (uint64_t i = row_lo(t); i < row_hi(t); i++) c[ix.bwt[i]]++;
                part[t] = c;
            });
        for (auto& th : pool) th.join();
    }
    std::vector<std::array<uint64_t, 256>> base(T);
    {
        std::array<uint64_t, 256> acc{};
        for (int t = 0; t < T; t++) {
            base[t] = acc;
            for (int c = 0; c < 256; c++) acc[c] += part[t][c];
        }
        uint64_t cum = 0;
        ix.less.assign(kLessLen, 0);
        for (uint64_t c = 0; c < kLessLen; c++) {
            ix.less[c] = cum;
            cum += acc[c];
        }
    }
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; t++)
            pool.emplace_back([&, t] {
                std::array<uint64_t, 256> cur = base[t];
                for (uint64_t i = row_lo(t); i < row_hi(t); i++) {
                    cur[ix.bwt[i]]++;
                    if (i % occ_k == 0) {
                        uint64_t j = i / occ_k;
                        for (uint8_t a : {'A', 'C', 'G', 'T', 'N', '$'}) ix.occ[a][j] = cur[a];
                    }
                }
            });
        for (auto& th : pool) th.join();
    }
}

void build_index_from_fasta(const std::string& fasta, uint32_t occ_k, uint64_t sa_s, int n_threads,
                            HostIndex& out, int gpu_device) {
    std::ifstream in(fasta, std::ios::binary);
    if (!in) throw std::runtime_error("io: cannot open " + fasta);
    struct Rec {
        uint32_t tax, gi;
        std::string seq;
    };
    std::vector<Rec> recs;
    std::string line;
    bool have = false;
    auto parse_u32 = [&](const std::string& t) -> uint32_t {
        if (t.empty()) throw std::runtime_error("format: invalid integer in FASTA header");
        uint64_t v = 0;
        for (char c : t) {
            if (c < '0' || c > '9') throw std::runtime_error("format: invalid integer '" + t + "' in FASTA header");
            v = v * 10 + (c - '0');
            if (v > 0xFFFFFFFFull) throw std::runtime_error("format: integer out of range in FASTA header");
        }
        return (uint32_t)v;
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            // record.id() = first whitespace-delimited token; header grammar GI-TAXID (util.rs:26-56)
            size_t e = line.find_first_of(" \t", 1);
            std::string id = line.substr(1, e == std::string::npos ? std::string::npos : e - 1);
            size_t d = id.find('-');
            if (d == std::string::npos || id.find('-', d + 1) != std::string::npos)
                throw std::runtime_error("format: invalid FASTA header '" + id + "' (want SEQID-TAXID)");
            recs.push_back(Rec{parse_u32(id.substr(d + 1)), parse_u32(id.substr(0, d)), {}});
            have = true;
        } else {
            if (!have) throw std::runtime_error("format: FASTA does not start with '>'");
            recs.back().seq += line;
        }
    }
    std::vector<SeqEntry> entries;
    for (auto& r : recs) entries.push_back(SeqEntry{r.tax, r.gi, (const uint8_t*)r.seq.data(), r.seq.size()});
    build_index(std::move(entries), occ_k, sa_s, n_threads, out, gpu_device);
}

}  // namespace mtsv

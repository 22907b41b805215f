// mgindex.cpp -- MG-index file codec (bincode 1.3.3 default options) and invariant checks.
// Replaces io::from_file::<MGIndex> / io::write_to_file (src/io.rs:115-133) for the layout of
// MGIndex (src/index.rs:60-68) and bio's SampledSuffixArray<BWT, Less, Occ>.
#include "mgindex.hpp"

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace mtsv {

namespace {

struct Reader {
    FILE* f;
    uint64_t size;
    uint64_t pos = 0;
    std::string path;

    [[noreturn]] void bad(const char* what) const {
        throw std::runtime_error("format: " + path + ": " + what);
    }
    void raw(void* dst, uint64_t bytes, const char* what) {
        if (bytes > size - pos) bad(what);
        if (bytes && fread(dst, 1, bytes, f) != bytes) throw std::runtime_error("io: short read on " + path);
        pos += bytes;
    }
    uint64_t u64(const char* what) {
        uint64_t v;
        raw(&v, 8, what);
        return v;
    }
    uint32_t u32(const char* what) {
        uint32_t v;
        raw(&v, 4, what);
        return v;
    }
    // Vec<T> header: u64 element count, bounded by what is left in the file
    uint64_t vec_len(uint64_t elem_bytes, const char* what) {
        uint64_t n = u64(what);
        if (n > (size - pos) / elem_bytes) bad(what);
        return n;
    }
};

struct Writer {
    FILE* f;
    std::string path;
    void raw(const void* src, uint64_t bytes) {
        if (bytes && fwrite(src, 1, bytes, f) != bytes) throw std::runtime_error("io: short write on " + path);
    }
    void u64(uint64_t v) { raw(&v, 8); }
    void u32(uint32_t v) { raw(&v, 4); }
};

}  // namespace

void load_index(const std::string& path, HostIndex& ix) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("io: cannot open " + path);
    struct Closer {
        FILE* f;
        ~Closer() { fclose(f); }
    } closer{f};
    if (fseeko(f, 0, SEEK_END) != 0) throw std::runtime_error("io: cannot seek " + path);
    Reader r{f, (uint64_t)ftello(f), 0, path};
    fseeko(f, 0, SEEK_SET);
    ix = HostIndex();
    ix.file_bytes = r.size;

    // sequences: Vec<u8>
    uint64_t n = r.vec_len(1, "sequences length exceeds file");
    if (n == 0) r.bad("empty sequences (no sentinel)");
    ix.text.resize(n);
    r.raw(ix.text.data(), n, "truncated sequences");
    // bins: Vec<Bin>, Bin = {gi u32, tax_id u32, start u64, end u64}
    uint64_t nb = r.vec_len(24, "bins length exceeds file");
    ix.bins.resize(nb);
    static_assert(sizeof(Bin) == 24, "Bin must match the serialised layout");
    r.raw(ix.bins.data(), nb * 24, "truncated bins");
    // suffix_array.bwt: Vec<u8>
    uint64_t nbwt = r.vec_len(1, "bwt length exceeds file");
    if (nbwt != n) r.bad("bwt length != sequences length");
    ix.bwt.resize(n);
    r.raw(ix.bwt.data(), n, "truncated bwt");
    // suffix_array.less: Vec<usize>
    uint64_t nl = r.vec_len(8, "less length exceeds file");
    if (nl != kLessLen) r.bad("less length != 118 (alphabet is not bio's n_alphabet)");
    ix.less.resize(nl);
    r.raw(ix.less.data(), nl * 8, "truncated less");
    // suffix_array.occ: Occ { occ: Vec<Vec<usize>>, k: u32 }
    uint64_t no = r.vec_len(8, "occ outer length exceeds file");
    if (no != kOccOuter) r.bad("occ outer length != 117");
    ix.occ.resize(no);
    for (uint64_t a = 0; a < no; a++) {
        uint64_t len = r.vec_len(8, "occ inner length exceeds file");
        ix.occ[a].resize(len);
        r.raw(ix.occ[a].data(), len * 8, "truncated occ");
    }
    ix.k = r.u32("missing occ k");
    // suffix_array.sample: Vec<usize>, s: usize
    uint64_t ns = r.vec_len(8, "sample length exceeds file");
    ix.sample.resize(ns);
    r.raw(ix.sample.data(), ns * 8, "truncated sample");
    ix.s = r.u64("missing s");
    // suffix_array.extra_rows: HashMap<usize, usize>
    uint64_t ne = r.vec_len(16, "extra_rows length exceeds file");
    ix.extra_rows.resize(ne);
    for (uint64_t i = 0; i < ne; i++) {
        ix.extra_rows[i].first = r.u64("truncated extra_rows");
        ix.extra_rows[i].second = r.u64("truncated extra_rows");
    }
    // suffix_array.sentinel: u8
    r.raw(&ix.sentinel, 1, "missing sentinel");
    if (r.pos != r.size) r.bad("trailing bytes after MGIndex");
    validate_index(ix);
}

void validate_index(const HostIndex& ix) {
    auto bad = [](const std::string& what) { throw std::runtime_error("format: " + what); };
    const uint64_t n = ix.n();
    if (n == 0) bad("empty sequences");
    if (ix.bwt.size() != n) bad("bwt length != sequences length");
    if (ix.less.size() != kLessLen) bad("less length != 118");
    if (ix.occ.size() != kOccOuter) bad("occ outer length != 117");
    if (ix.k == 0) bad("occ sampling interval k == 0");
    if (ix.s == 0) bad("suffix sampling interval s == 0");
    if (ix.sentinel != '$') bad("sentinel is not '$'");
    if (ix.text[n - 1] != ix.sentinel) bad("sequences do not end with the sentinel");
    if (ix.sample.size() != (n + ix.s - 1) / ix.s) bad("sample length != ceil(n/s)");
    const uint64_t nchk = (n - 1) / ix.k + 1;
    for (uint8_t a : {'A', 'C', 'G', 'T', 'N', '$'})
        if (ix.occ[a].size() != nchk) bad("occ checkpoint count != floor((n-1)/k)+1");
    if (ix.extra_rows.size() > 1) bad("more than one extra row");
    // bins: contiguous, ascending, covering [0, n-1)  (index.rs:497-510)
    uint64_t pos = 0;
    for (size_t i = 0; i < ix.bins.size(); i++) {
        if (ix.bins[i].start != pos || ix.bins[i].end < ix.bins[i].start) bad("bins are not contiguous");
        pos = ix.bins[i].end;
    }
    if (pos != n - 1) bad("bins do not cover the sequences");
    // less must be the cumulative symbol histogram of the bwt over $ < A < C < G < N < T
    uint64_t hist[256] = {0};
    for (uint64_t i = 0; i < n; i++) hist[ix.bwt[i]]++;
    uint64_t cum = 0;
    for (uint64_t c = 0; c < kLessLen; c++) {
        if (ix.less[c] != cum) bad("less is not the prefix sum of the bwt histogram");
        if (c < 256) cum += hist[c];
    }
    for (int c = 0; c < 256; c++)
        if (hist[c] && c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N' && c != '$')
            bad("bwt holds a symbol outside ACGTN$");
    if (hist['$'] != 1) bad("bwt must hold exactly one sentinel");
    for (uint64_t v : ix.sample)
        if (v >= n) bad("suffix sample out of range");
}

void write_index(const HostIndex& ix, const std::string& path) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("io: cannot create " + path);
    Writer w{f, path};
    try {
        w.u64(ix.text.size());
        w.raw(ix.text.data(), ix.text.size());
        w.u64(ix.bins.size());
        w.raw(ix.bins.data(), ix.bins.size() * sizeof(Bin));
        w.u64(ix.bwt.size());
        w.raw(ix.bwt.data(), ix.bwt.size());
        w.u64(ix.less.size());
        w.raw(ix.less.data(), ix.less.size() * 8);
        w.u64(ix.occ.size());
        for (auto& v : ix.occ) {
            w.u64(v.size());
            w.raw(v.data(), v.size() * 8);
        }
        w.u32(ix.k);
        w.u64(ix.sample.size());
        w.raw(ix.sample.data(), ix.sample.size() * 8);
        w.u64(ix.s);
        w.u64(ix.extra_rows.size());
        for (auto& kv : ix.extra_rows) {
            w.u64(kv.first);
            w.u64(kv.second);
        }
        w.raw(&ix.sentinel, 1);
    } catch (...) {
        fclose(f);
        throw;
    }
    if (fclose(f) != 0) throw std::runtime_error("io: close failed on " + path);
}

}  // namespace mtsv

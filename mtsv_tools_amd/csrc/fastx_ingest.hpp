// fastx_ingest.hpp -- block-parallel FASTA/FASTQ ingest for mtsv-binner (the producer side of
// get_fastx_and_write_matching_bin_ids, src/binner.rs:149-217; bio::io::{fasta,fastq} readers).
//
// Plain (not gzip-compressed) regular files are mapped and cut into blocks at record boundaries;
// a pool of threads parses the blocks, the consumer takes them back in input order.  The parse of a
// block is strict (FASTQ: exactly header / sequence / '+' / quality of equal length; FASTA: '>'
// records) and must end exactly on the next block's boundary.  The first block that does not fit
// (wrapped FASTQ, empty reads, truncated files, anything the serial reader would report) is handed
// back as "irregular at byte offset X": every block before it is, by induction from offset 0,
// exactly what the serial reader would have produced, and the caller continues with the serial
// reader from X, which also produces the reference's error for broken input.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <new>
#include <deque>
#include <functional>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pgzip.hpp"

namespace mtsv_ingest {

// Where the bases of a block live.  mtsv-binner points these at mtsv_host_alloc / mtsv_host_free, so that a parsed block
// is page-locked memory the GPU's copy engine reads in place (no staging copy); elsewhere they stay malloc / free.
struct ByteAlloc {
    void* (*alloc)(size_t) = nullptr;
    void (*release)(void*) = nullptr;
};
inline ByteAlloc& byte_alloc() {
    static ByteAlloc a;
    return a;
}

// the part of std::vector<uint8_t> the parsers use, on memory from byte_alloc() (malloc when that is unset or fails)
class ByteBuf {
   public:
    ByteBuf() = default;
    ByteBuf(const ByteBuf&) = delete;
    ByteBuf& operator=(const ByteBuf&) = delete;
    ByteBuf(ByteBuf&& o) noexcept { swap(o); }
    ByteBuf& operator=(ByteBuf&& o) noexcept {
        swap(o);
        return *this;
    }
    ~ByteBuf() { drop(); }
    void swap(ByteBuf& o) noexcept {
        std::swap(p_, o.p_);
        std::swap(n_, o.n_);
        std::swap(cap_, o.cap_);
        std::swap(pinned_, o.pinned_);
    }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    uint64_t size() const { return n_; }
    const uint8_t* begin() const { return p_; }
    const uint8_t* end() const { return p_ + n_; }
    void clear() { n_ = 0; }
    void reserve(uint64_t c) {
        if (c <= cap_) return;
        c = std::max<uint64_t>(c, cap_ + cap_ / 2);
        uint8_t* q = nullptr;
        bool pin = false;
        if (byte_alloc().alloc) {
            q = (uint8_t*)byte_alloc().alloc(c);
            pin = q != nullptr;
        }
        if (!q) q = (uint8_t*)malloc(c);
        if (!q) throw std::bad_alloc();
        if (n_) memcpy(q, p_, n_);
        const uint64_t keep = n_;
        drop();
        p_ = q;
        n_ = keep;
        cap_ = c;
        pinned_ = pin;
    }
    // append [a, b) (the only insert position the parsers use is end())
    void insert(const uint8_t* /*at_end*/, const uint8_t* a, const uint8_t* b) {
        const uint64_t len = (uint64_t)(b - a);
        if (n_ + len > cap_) reserve(n_ + len);
        memcpy(p_ + n_, a, len);
        n_ += len;
    }
    void assign(const uint8_t* a, const uint8_t* b) {
        n_ = 0;
        insert(nullptr, a, b);
    }

   private:
    void drop() {
        if (p_) {
            if (pinned_) byte_alloc().release(p_);
            else free(p_);
        }
        p_ = nullptr;
        n_ = cap_ = 0;
        pinned_ = false;
    }
    uint8_t* p_ = nullptr;
    uint64_t n_ = 0, cap_ = 0;
    bool pinned_ = false;
};

struct ReadBlock {
    ByteBuf bases;
    std::vector<uint64_t> off{0};     // n + 1 offsets into bases
    std::string ids;                  // NUL-terminated ids, back to back
    std::vector<uint64_t> id_off{0};  // n + 1 offsets into ids
    uint64_t n() const { return off.size() - 1; }
    void clear() {
        bases.clear();
        off.assign(1, 0);
        ids.clear();
        id_off.assign(1, 0);
    }
    // append records [from, b.n()) of b
    void append(const ReadBlock& b, uint64_t from = 0) {
        const uint64_t nb = b.n();
        if (from >= nb) return;
        const uint64_t b0 = b.off[from], i0 = b.id_off[from];
        const uint64_t base_b = bases.size(), base_i = ids.size();
        bases.insert(bases.end(), b.bases.begin() + (ptrdiff_t)b0, b.bases.end());
        ids.append(b.ids, i0, std::string::npos);
        off.reserve(off.size() + (nb - from));
        id_off.reserve(id_off.size() + (nb - from));
        for (uint64_t r = from + 1; r <= nb; r++) {
            off.push_back(base_b + (b.off[r] - b0));
            id_off.push_back(base_i + (b.id_off[r] - i0));
        }
    }
};

class ParallelFastx {
   public:
    // called on every block object the parser threads create themselves (the others arrive through next(), in exchange for
    // the parsed ones): the caller may swap in a block whose buffers are already allocated
    std::function<void(ReadBlock&)> prepare;

    // CORRUPT: the gzip data decoded but a member's CRC-32 / length did not match after records of it had been handed out
    enum Result { BLOCK = 0, END = 1, IRREGULAR = 2, CORRUPT = 3 };

    ~ParallelFastx() { close(); }

    // false: not a plain regular file (gzip magic, pipe, empty, mmap failure) -> use the serial reader
    bool open(const std::string& path, bool fastq, unsigned threads, uint64_t block_bytes = 16ull << 20) {
        fastq_ = fastq;
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) {
            ::close(fd);
            return false;
        }
        size_ = (uint64_t)st.st_size;
        void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            ::close(fd);
            return false;
        }
        fd_ = fd;
        data_ = (const uint8_t*)m;
        if (data_[0] == 0x1f && data_[1] == 0x8b) {  // gzip magic (binner.rs:21-33)
            close();
            return false;
        }
        // block boundaries
        // (blocks of more than 32 MiB start with an eighth, an eighth, a quarter and a half of a block: the consumer has
        //  its first records after an eighth of the time a whole block takes to parse)
        bounds_.push_back(0);
        {
            const uint64_t ramp[4] = {block_bytes / 8, block_bytes / 8, block_bytes / 4, block_bytes / 2};
            uint64_t t = 0;
            for (size_t k = 0;; k++) {
                t += (block_bytes > (32ull << 20) && k < 4) ? std::max<uint64_t>(ramp[k], 1) : block_bytes;
                if (t >= size_) break;
                uint64_t b = find_boundary(std::max(t, bounds_.back() + 1));
                if (b != UINT64_MAX && b > bounds_.back() && b < size_) {
                    bounds_.push_back(b);
                    t = std::max(t, b);
                }
            }
        }
        bounds_.push_back(size_);
        n_blocks_ = bounds_.size() - 1;
        window_ = 2 * std::max(1u, threads) + 2;
        for (unsigned k = 0; k < std::max(1u, threads); k++) pool_.emplace_back([this] { worker(); });
        return true;
    }

    // blocks come back in input order
    Result next(ReadBlock& out, uint64_t* irregular_offset) {
        if (consumed_ >= n_blocks_) return END;
        std::unique_ptr<Parsed> p;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return done_.count(consumed_) != 0; });
            p = std::move(done_[consumed_]);
            done_.erase(consumed_);
            consumed_++;
        }
        cv_.notify_all();
        if (!p->ok) {
            *irregular_offset = bounds_[consumed_ - 1];
            stop();
            return IRREGULAR;
        }
        std::swap(out, p->block);  // the caller's old buffers go back to the workers: no fresh pages per block
        p->block.clear();
        {
            std::lock_guard<std::mutex> lk(mu_);
            free_.push_back(std::move(p));
        }
        return BLOCK;
    }

    void close() {
        stop();
        if (data_) munmap((void*)data_, size_);
        data_ = nullptr;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }

    // ---- shared with GzFastx: strict block parsers and the record-boundary heuristic on a memory range ----
    static bool parse_block(const uint8_t* d, uint64_t len, bool fastq, bool first, bool last, ReadBlock& b) {
        return fastq ? parse_fastq(d, len, b) : parse_fasta(d, len, first, last, b);
    }
    // first record start at or after t inside d[0, size) (a hint only: blocks are validated by the strict parse)
    static uint64_t find_boundary_in(const uint8_t* d, uint64_t size, bool fastq, uint64_t t) {
        auto lend = [&](uint64_t s, uint64_t limit) {
            const void* nl = memchr(d + s, '\n', limit - s);
            return nl ? (uint64_t)((const uint8_t*)nl - d) : limit;
        };
        if (t == 0 || t >= size) return t >= size ? UINT64_MAX : 0;
        const uint64_t scan_limit = std::min<uint64_t>(size, t + (4ull << 20));
        uint64_t s = lend(t - 1, scan_limit) + 1;
        if (!fastq) {
            while (s < scan_limit) {
                if (d[s] == '>') return s;
                s = lend(s, scan_limit) + 1;
            }
            return UINT64_MAX;
        }
        while (s < scan_limit) {
            if (d[s] == '@') {
                uint64_t l1 = lend(s, size) + 1;
                uint64_t l2 = l1 < size ? lend(l1, size) + 1 : size;
                if (l2 < size && d[l2] == '+') {
                    uint64_t l3 = lend(l2, size) + 1;
                    uint64_t l4 = l3 < size ? lend(l3, size) + 1 : size;
                    if (l4 < size && d[l4] == '@') return s;  // (a record cut off by the end of the range is no proof)
                }
            }
            s = lend(s, scan_limit) + 1;
        }
        return UINT64_MAX;
    }

   private:
    struct Parsed {
        ReadBlock block;
        bool ok = true;
    };

    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : pool_)
            if (t.joinable()) t.join();
        pool_.clear();
    }

    // A block is parsed where the file is mapped: one madvise(MADV_POPULATE_READ) maps its pages, and the parser reads the
    // page cache's own pages.  (Until the end of r03 every block was copied out with pread first -- 5 GB/s per thread,
    // more than half of a parser thread's time; page faults in one address space are no longer what they were when that
    // was chosen: per-VMA locks.)  MTSV_INGEST_PREAD=1 keeps the copy.
    void worker() {
        std::vector<uint8_t> buf;
        for (;;) {
            uint64_t k = next_block_.fetch_add(1);
            if (k >= n_blocks_) return;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || k < consumed_ + window_; });
                if (stop_) return;
            }
            std::unique_ptr<Parsed> p;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (!free_.empty()) {
                    p = std::move(free_.back());
                    free_.pop_back();
                }
            }
            if (!p) {
                p = std::make_unique<Parsed>();
                if (prepare) prepare(p->block);  // (a block with its storage already there, from the caller's stock)
            }
            const uint64_t s = bounds_[k], len = bounds_[k + 1] - s;
            const uint8_t* src;
            uint64_t got = 0;
            if (from_map_) {  // parse where the page cache's pages are mapped (one madvise call maps the block's pages)
                src = data_ + s;
                const uint64_t a = s & ~4095ull;
#ifdef MADV_POPULATE_READ
                (void)madvise(const_cast<uint8_t*>(data_) + a, s + len - a, MADV_POPULATE_READ);
#endif
                got = len;
            } else {
                if (buf.size() < len) buf.resize(len);
                while (got < len) {
                    ssize_t r = pread(fd_, buf.data() + got, len - got, (off_t)(s + got));
                    if (r <= 0) break;
                    got += (uint64_t)r;
                }
                src = buf.data();
            }
            p->ok = got == len && (fastq_ ? parse_fastq(src, len, p->block) : parse_fasta(src, len, k == 0, k + 1 == n_blocks_, p->block));
            if (from_map_) {  // the block's pages leave the address space again: a 300 GB input must not become 300 GB of mapped pages
                const uint64_t a = (s + 4095) & ~4095ull, e = (s + len) & ~4095ull;
                if (e > a) (void)madvise(const_cast<uint8_t*>(data_) + a, e - a, MADV_DONTNEED);
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                done_[k] = std::move(p);
            }
            cv_.notify_all();
        }
    }

    // [s, e) of the line starting at s, e = position of '\n' or end; next line starts at e + 1
    uint64_t line_end(uint64_t s, uint64_t limit) const {
        const void* nl = memchr(data_ + s, '\n', limit - s);
        return nl ? (uint64_t)((const uint8_t*)nl - data_) : limit;
    }

    // first record start at or after t (a hint only: blocks are validated by the strict parse)
    uint64_t find_boundary(uint64_t t) const {
        const uint64_t scan_limit = std::min<uint64_t>(size_, t + (4ull << 20));
        uint64_t s = line_end(t - 1, scan_limit) + 1;  // first line start after t-1
        if (!fastq_) {
            while (s < scan_limit) {
                if (data_[s] == '>') return s;
                s = line_end(s, scan_limit) + 1;
            }
            return UINT64_MAX;
        }
        while (s < scan_limit) {
            if (data_[s] == '@') {
                uint64_t l1 = line_end(s, size_) + 1;
                uint64_t l2 = l1 < size_ ? line_end(l1, size_) + 1 : size_;
                if (l2 < size_ && data_[l2] == '+') {
                    uint64_t l3 = line_end(l2, size_) + 1;
                    uint64_t l4 = l3 < size_ ? line_end(l3, size_) + 1 : size_;
                    if (l4 >= size_ || data_[l4] == '@') return s;
                }
            }
            s = line_end(s, scan_limit) + 1;
        }
        return UINT64_MAX;
    }

    static void push_id(ReadBlock& b, const uint8_t* h, uint64_t len) {  // first token after the marker
        uint64_t e = 1;
        while (e < len && h[e] != ' ' && h[e] != '\t') e++;
        if (len > 1) b.ids.append((const char*)h + 1, e - 1);
        b.ids.push_back('\0');
        b.id_off.push_back(b.ids.size());
    }

    static uint64_t eol(const uint8_t* d, uint64_t s, uint64_t limit) {
        const void* nl = memchr(d + s, '\n', limit - s);
        return nl ? (uint64_t)((const uint8_t*)nl - d) : limit;
    }

    static bool parse_fastq(const uint8_t* d, uint64_t e, ReadBlock& b) {
        b.clear();
        b.bases.reserve(e / 2);
        uint64_t p = 0;
        while (p < e) {
            uint64_t e0 = eol(d, p, e);
            if (e0 >= e) return false;  // header without a sequence line inside the block
            uint64_t p1 = e0 + 1, e1 = eol(d, p1, e);
            if (e1 >= e) return false;
            uint64_t p2 = e1 + 1;
            uint64_t e2 = p2 + 1 < e && d[p2 + 1] == '\n' ? p2 + 1 : eol(d, p2, e);  // ("+" alone on its line, as a rule)
            if (e2 >= e) return false;
            uint64_t p3 = e2 + 1, e3 = eol(d, p3, e);  // the last line may end at e without '\n'
            uint64_t h_len = e0 - p, s_len = e1 - p1, q_len = e3 - p3;
            if (h_len && d[e0 - 1] == '\r') h_len--;
            if (s_len && d[e1 - 1] == '\r') s_len--;
            if (q_len && d[e3 - 1] == '\r') q_len--;
            if (h_len == 0 || d[p] != '@' || s_len == 0 || d[p1] == '+' || p2 >= e || d[p2] != '+' || q_len != s_len) return false;
            push_id(b, d + p, h_len);
            b.bases.insert(b.bases.end(), d + p1, d + p1 + s_len);
            b.off.push_back(b.bases.size());
            p = e3 + 1;
        }
        return p == e || p == e + 1;  // ended on the boundary (or on a final line without '\n')
    }

    static bool parse_fasta(const uint8_t* d, uint64_t e, bool first, bool last, ReadBlock& b) {
        b.clear();
        b.bases.reserve(e);
        uint64_t p = 0;
        if (first) {  // the serial reader skips blank lines before the first header
            while (p < e) {
                uint64_t le = eol(d, p, e);
                uint64_t len = le - p;
                if (len && d[le - 1] == '\r') len--;
                if (len) break;
                p = le + 1;
            }
            if (p >= e) return last;
        }
        if (d[p] != '>') return false;
        while (p < e) {
            uint64_t le = eol(d, p, e);
            uint64_t len = le - p;
            if (len && d[le - 1] == '\r') len--;
            if (len && d[p] == '>') {
                if (b.id_off.size() > 1) b.off.push_back(b.bases.size());  // close the previous record
                push_id(b, d + p, len);
            } else if (len) {
                b.bases.insert(b.bases.end(), d + p, d + p + len);
            }
            p = le + 1;
        }
        b.off.push_back(b.bases.size());
        return b.off.size() == b.id_off.size();
    }

    const uint8_t* data_ = nullptr;
    bool from_map_ = getenv("MTSV_INGEST_PREAD") == nullptr;  // parse where the file is mapped (MTSV_INGEST_PREAD=1: copy the block out with pread first)
    int fd_ = -1;
    uint64_t size_ = 0;
    bool fastq_ = true;
    std::vector<uint64_t> bounds_;
    uint64_t n_blocks_ = 0, window_ = 4;
    std::atomic<uint64_t> next_block_{0};
    std::mutex mu_;
    std::condition_variable cv_;
    std::map<uint64_t, std::unique_ptr<Parsed>> done_;
    std::vector<std::unique_ptr<Parsed>> free_;
    uint64_t consumed_ = 0;
    bool stop_ = false;
    std::vector<std::thread> pool_;
};

// gzip input: parallel decompression (pgzip.hpp) feeding the same strict block parsers.  A producer thread
// inflates the file round by round, cuts the text at record boundaries into blocks and hands them to a pool of
// parser threads; next() returns the blocks in input order.  IRREGULAR carries the offset in the DECOMPRESSED
// stream from which the caller's serial reader (zlib: gzseek) has to continue -- for a block the strict parse
// rejects, and also when the parallel inflater gives up (the serial reader then reports what zlib thinks).
class GzFastx {
   public:
    using Result = ParallelFastx::Result;
    ~GzFastx() { close(); }

    bool open(const std::string& path, bool fastq, unsigned threads, uint64_t block_bytes = 16ull << 20) {
        fastq_ = fastq;
        block_ = std::max<uint64_t>(block_bytes, 1);
        threads_ = std::max(1u, threads);
        uint64_t chunk = 2ull << 20;  // compressed bytes per inflater chunk (two chunks per thread and round)
        if (const char* e = getenv("MTSV_PGZIP_CHUNK")) chunk = strtoull(e, nullptr, 10);
        if (!gz_.open(path, threads_, chunk)) return false;
        window_ = 2 * threads_ + 2;
        producer_ = std::thread([this] { produce(); });
        for (unsigned k = 0; k < threads_; k++) pool_.emplace_back([this] { worker(); });
        return true;
    }

    Result next(ReadBlock& out, uint64_t* irregular_offset) {
        std::unique_ptr<Task> t;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return done_.count(consumed_) != 0 || (produced_all_ && consumed_ >= n_tasks_); });
            if (!done_.count(consumed_)) {
                if (failed_at_ != UINT64_MAX) {  // the inflater gave up: everything before failed_at_ has been delivered
                    *irregular_offset = failed_at_;
                    const bool corrupt = corrupt_;
                    lk.unlock();
                    stop();
                    return corrupt ? ParallelFastx::CORRUPT : ParallelFastx::IRREGULAR;
                }
                return ParallelFastx::END;
            }
            t = std::move(done_[consumed_]);
            done_.erase(consumed_);
            consumed_++;
        }
        cv_.notify_all();
        if (!t->ok) {
            *irregular_offset = t->offset;
            stop();
            return ParallelFastx::IRREGULAR;
        }
        std::swap(out, t->block);
        return ParallelFastx::BLOCK;
    }

    void close() {
        stop();
        gz_.close();
    }

   private:
    struct Task {
        std::shared_ptr<std::vector<uint8_t>> keep;  // the buffer ptr points into (a whole inflated piece, or a small copy)
        const uint8_t* ptr = nullptr;
        uint64_t len = 0;
        uint64_t offset = 0;  // of ptr[0] in the decompressed stream
        bool first = false, last = false, ok = true;
        ReadBlock block;
    };

    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (producer_.joinable()) producer_.join();
        for (auto& t : pool_)
            if (t.joinable()) t.join();
        pool_.clear();
    }

    // hand a block to the parsers (bounded number in flight); false when stopping
    bool submit(std::shared_ptr<std::vector<uint8_t>> keep, const uint8_t* ptr, uint64_t len, uint64_t offset, bool first, bool last) {
        auto t = std::make_unique<Task>();
        t->keep = std::move(keep);
        t->ptr = ptr;
        t->len = len;
        t->offset = offset;
        t->first = first;
        t->last = last;
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || n_tasks_ < consumed_ + window_; });
        if (stop_) return false;
        queue_.emplace_back(n_tasks_++, std::move(t));
        lk.unlock();
        cv_.notify_all();
        return true;
    }

    // The inflated pieces are parsed in place: blocks are (pointer, length) views into a piece, cut at record
    // boundaries; only the record that straddles two pieces is copied (tail of one piece + head of the next).
    void produce() {
        using Buf = std::shared_ptr<std::vector<uint8_t>>;
        std::vector<uint8_t> carry;  // unparsed tail of the previous piece (less than a block, usually less than a record)
        uint64_t carry_off = 0;      // decompressed offset of carry[0]
        bool first = true;
        std::vector<std::vector<uint8_t>> pieces;
        auto finish = [&](uint64_t failed_at) {
            std::lock_guard<std::mutex> lk(mu_);
            failed_at_ = failed_at;
            produced_all_ = true;
            cv_.notify_all();
        };
        while (!gz_.at_end()) {
            bool stopping;
            {
                std::lock_guard<std::mutex> lk(mu_);
                stopping = stop_;
            }
            if (stopping) return finish(UINT64_MAX);  // (finish takes the mutex itself)
            if (!gz_.next_round(pieces)) {
                // "Cannot decode in parallel" (no entry point, a header the decoder does not take): the serial reader takes
                // over at carry_off -- nothing at or after it has been handed out.  A CRC-32 / length mismatch is
                // another matter: a member is checked at its end, after its earlier rounds went to the GPU, so records
                // already delivered are suspect and the run must fail (the reference's reader fails such a file too).
                if (gz_.crc_mismatch() && !first) {
                    std::lock_guard<std::mutex> lk(mu_);
                    corrupt_ = true;
                }
                return finish(carry_off);
            }
            for (auto& piece : pieces) {
                if (piece.empty()) continue;
                Buf pb = std::make_shared<std::vector<uint8_t>>(std::move(piece));
                const uint8_t* d = pb->data();
                const uint64_t n = pb->size();
                uint64_t pos = 0;  // bytes of this piece already handed out (or moved into the straddling block)
                if (!carry.empty()) {
                    // first record start inside the piece: the carry plus everything before it is one block
                    uint64_t b0 = ParallelFastx::find_boundary_in(d, n, fastq_, 1);
                    if (b0 == UINT64_MAX) {  // no boundary in the whole piece: keep collecting
                        carry.insert(carry.end(), d, d + n);
                        // a gigabyte without a record start (one FASTA record of that size, or not FASTA / FASTQ at all):
                        // the serial reader continues at carry_off and gives the reference's result or error
                        if (carry.size() > (1ull << 30)) return finish(carry_off);
                        continue;
                    }
                    Buf sb = std::make_shared<std::vector<uint8_t>>(std::move(carry));
                    sb->insert(sb->end(), d, d + b0);
                    carry.clear();
                    if (!submit(sb, sb->data(), sb->size(), carry_off, first, false)) return finish(UINT64_MAX);
                    first = false;
                    carry_off += sb->size();
                    pos = b0;
                }
                while (n - pos > block_) {
                    const uint64_t b = ParallelFastx::find_boundary_in(d, n, fastq_, pos + block_);
                    if (b == UINT64_MAX || b <= pos) break;
                    if (!submit(pb, d + pos, b - pos, carry_off, first, false)) return finish(UINT64_MAX);
                    first = false;
                    carry_off += b - pos;
                    pos = b;
                }
                // what is left may end inside a record: everything up to the last record start that can be verified
                // goes out as a view, the rest waits for the next piece
                uint64_t lastb = pos;
                if (n - pos > 0) {
                    // search backwards from the end for a record start with some lookahead left behind it
                    const uint64_t tail = std::min<uint64_t>(n - pos, 1u << 16);
                    uint64_t t = n - tail;
                    for (;;) {
                        const uint64_t b = ParallelFastx::find_boundary_in(d, n, fastq_, std::max<uint64_t>(t, pos + 1));
                        if (b == UINT64_MAX || b <= lastb) break;
                        lastb = b;
                        t = b + 1;
                    }
                }
                if (lastb > pos) {
                    if (!submit(pb, d + pos, lastb - pos, carry_off, first, false)) return finish(UINT64_MAX);
                    first = false;
                    carry_off += lastb - pos;
                    pos = lastb;
                }
                carry.assign(d + pos, d + n);
            }
        }
        if (!carry.empty() || first) {
            Buf cb = std::make_shared<std::vector<uint8_t>>(std::move(carry));
            if (!submit(cb, cb->data(), cb->size(), carry_off, first, true)) return finish(UINT64_MAX);
        }
        finish(UINT64_MAX);
    }

    void worker() {
        for (;;) {
            uint64_t idx;
            std::unique_ptr<Task> t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !queue_.empty() || produced_all_; });
                if (stop_) return;
                if (queue_.empty()) {
                    if (produced_all_) return;
                    continue;
                }
                idx = queue_.front().first;
                t = std::move(queue_.front().second);
                queue_.pop_front();
            }
            t->ok = ParallelFastx::parse_block(t->ptr, t->len, fastq_, t->first, t->last, t->block);
            t->keep.reset();
            {
                std::lock_guard<std::mutex> lk(mu_);
                done_[idx] = std::move(t);
            }
            cv_.notify_all();
        }
    }

    mtsv_pgzip::ParallelGunzip gz_;
    bool fastq_ = true;
    uint64_t block_ = 16 << 20, window_ = 4;
    unsigned threads_ = 1;
    std::thread producer_;
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::pair<uint64_t, std::unique_ptr<Task>>> queue_;
    std::map<uint64_t, std::unique_ptr<Task>> done_;
    uint64_t n_tasks_ = 0, consumed_ = 0, failed_at_ = UINT64_MAX;
    bool produced_all_ = false, stop_ = false, corrupt_ = false;
};

}  // namespace mtsv_ingest

// pgzip.hpp -- parallel decompression of an ordinary (single-stream) gzip file for mtsv-binner's ingest.
//
// The reference opens .gz input through one inflate stream (src/binner.rs:21-33, flate2); at the rate the GPU
// path consumes reads that stream is the bottleneck of the whole binary.  A deflate stream cannot be entered in
// the middle in general -- a block may refer to the 32 KiB of output before it -- but it can be decoded from a
// block boundary with that window left SYMBOLIC and resolved once the preceding part is known (the two-pass
// scheme of Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed files and random access to DNA
// sequences", 2019).  Here:
//
//   1. the compressed bytes of a round are cut into chunks; for every chunk but the first a thread searches the
//      first bit position at which a dynamic-Huffman block header parses, its code sets are complete, and it and
//      its successors decode to plausible text (printable ASCII / newlines) for a few tens of KiB;
//   2. every chunk is inflated by its own thread from its entry point up to the next chunk's entry point, into
//      16-bit symbols: a literal, or 256 + i for "byte i of the 32 KiB window before this chunk";
//   3. the windows at the chunk ends are resolved one after the other (32 KiB each), then all chunks are resolved
//      to bytes in parallel, with their CRC-32, which is combined and checked against every member's trailer.
//
// A chunk whose inflation does not arrive exactly at the next entry point has proven that entry point false and
// simply keeps going over it.  Multi-member files, stored and fixed-Huffman blocks are handled by the inflater
// (entry points are only ever dynamic blocks).  Anything this code does not like -- header flags it cannot
// parse, a reference outside the window, a bad code set, a CRC mismatch -- is reported as an error; the caller
// then falls back to zlib from the start of the file or, when records have already been handed out, fails the
// run like a corrupt file would.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace mtsv_pgzip {

constexpr uint32_t kWindow = 32768;

// LSB-first bit reader over a memory range; position in bits since `base`
struct BitReader {
    const uint8_t* base;
    const uint8_t* end;
    const uint8_t* p;
    uint64_t buf = 0;
    int cnt = 0;
    BitReader(const uint8_t* b, const uint8_t* e, uint64_t bitpos) : base(b), end(e), p(b + (bitpos >> 3)) {
        refill();
        consume((int)(bitpos & 7));
    }
    void refill() {
        if (p + 8 <= end) {  // eight bytes at once: keeps 56..63 valid bits
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
            return;
        }
        while (cnt <= 56) {
            if (p < end) buf |= (uint64_t)*p++ << cnt;
            else if (p < end + 16) p++;  // zero bits past the end; overrun() tells when they have been consumed
            else break;
            cnt += 8;
        }
    }
    uint64_t bitpos() const { return (uint64_t)(p - base) * 8 - (uint64_t)cnt; }
    bool overrun_now() const { return bitpos() > (uint64_t)(end - base) * 8; }
    uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    void consume(int n) {
        buf >>= n;
        cnt -= n;
    }
    uint32_t get(int n) {
        if (cnt < n) refill();
        uint32_t v = peek(n);
        consume(n);
        return v;
    }
    void align_byte() { consume(cnt & 7); }
};

// canonical Huffman decoding table, indexed by the next `bits` input bits; entry = symbol << 4 | length, 0 = invalid
struct Huff {
    std::vector<uint16_t> tab;
    int bits = 0;
    // returns false when the code set is over-subscribed, or incomplete (unless allow_single: exactly one code)
    bool build(const uint8_t* len, int n, bool allow_single) {
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        int maxl = 15;
        while (maxl > 0 && count[maxl] == 0) maxl--;
        if (maxl == 0) {
            bits = 1;
            tab.assign(2, 0);
            return allow_single;  // no codes at all: legal only for an unused distance tree
        }
        long left = 1;
        for (int l = 1; l <= maxl; l++) {
            left = (left << 1) - count[l];
            if (left < 0) return false;
        }
        int used = 0;
        for (int l = 1; l <= 15; l++) used += count[l];
        if (left > 0 && !(allow_single && used == 1 && maxl == 1)) return false;  // zlib: an incomplete set only as one 1-bit distance code
        bits = maxl;
        tab.assign((size_t)1 << bits, 0);
        int next[16], code = 0;
        for (int l = 1; l <= maxl; l++) {
            code = (code + count[l - 1]) << 1;
            next[l] = code;
        }
        for (int s = 0; s < n; s++) {
            const int l = len[s];
            if (!l) continue;
            uint32_t c = (uint32_t)next[l]++, r = 0;
            for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);  // codes are sent MSB first
            for (uint32_t i = r; i < tab.size(); i += 1u << l) tab[i] = (uint16_t)((s << 4) | l);
        }
        return true;
    }
};

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// Two-level decoding table for the literal/length and distance codes: a root of ROOT bits (codes up to that length
// decode with one look-up in a table that stays in the L1 cache) and sub-tables for the longer ones.  An entry says
// everything the inner loop needs: bits 0-4 the code's length, bits 5-7 its kind, bits 8-11 the extra bits of a
// length / distance (or the index width of a sub-table), bits 16-31 the literal, the base value, or the sub-table's
// offset.  Building one costs ~1 K entries, not 2^15 as a flat table of the longest code would.
struct FastHuff {
    enum : uint32_t { INVALID = 0, LITERAL = 1, BASE = 2, END = 3, SUB = 4 };
    static constexpr int kMaxEntries = 1024 + 2048;
    uint32_t tab[kMaxEntries];
    int root = 0;
    static uint32_t make(uint32_t len, uint32_t kind, uint32_t extra, uint32_t value) { return len | (kind << 5) | (extra << 8) | (value << 16); }
    static uint32_t len_of(uint32_t e) { return e & 31u; }
    static uint32_t kind_of(uint32_t e) { return (e >> 5) & 7u; }
    static uint32_t extra_of(uint32_t e) { return (e >> 8) & 15u; }
    static uint32_t value_of(uint32_t e) { return e >> 16; }
    // lit_len: symbols 0..255 literals, 256 end of block, 257..285 lengths; else distance symbols 0..29
    bool build(const uint8_t* len, int n, bool allow_single, int root_bits, bool lit_len) {
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        int maxl = 15;
        while (maxl > 0 && count[maxl] == 0) maxl--;
        root = root_bits;
        if (maxl == 0) {
            for (int i = 0; i < (1 << root); i++) tab[i] = 0;
            return allow_single;  // no codes at all: legal only for an unused distance tree
        }
        long left = 1;
        for (int l = 1; l <= maxl; l++) {
            left = (left << 1) - count[l];
            if (left < 0) return false;
        }
        int used = 0;
        for (int l = 1; l <= 15; l++) used += count[l];
        if (left > 0 && !(allow_single && used == 1 && maxl == 1)) return false;  // zlib: an incomplete set only as one 1-bit distance code
        if (root > maxl) root = maxl;
        const uint32_t root_size = 1u << root;
        for (uint32_t i = 0; i < root_size; i++) tab[i] = 0;
        int next[16], code = 0;
        for (int l = 1; l <= maxl; l++) {
            code = (code + count[l - 1]) << 1;
            next[l] = code;
        }
        // first pass over the long codes: the longest code under every root prefix gives its sub-table's width
        uint8_t sub_bits[1 << 11] = {0};  // root <= 11
        uint32_t rev_of[288];
        for (int sy = 0; sy < n; sy++) {
            const int l = len[sy];
            if (!l) continue;
            const uint32_t c = (uint32_t)next[l]++;
            uint32_t r = 0;
            for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);  // codes are sent MSB first
            rev_of[sy] = r;
            if (l > root) {
                const uint32_t pre = r & (root_size - 1);
                if ((int)sub_bits[pre] < l - root) sub_bits[pre] = (uint8_t)(l - root);
            }
        }
        uint32_t used_entries = root_size;
        if (maxl > root)
            for (uint32_t pre = 0; pre < root_size; pre++)
                if (sub_bits[pre]) {
                    if (used_entries + (1u << sub_bits[pre]) > (uint32_t)kMaxEntries) return false;
                    tab[pre] = make((uint32_t)root, SUB, sub_bits[pre], used_entries);
                    for (uint32_t i = 0; i < (1u << sub_bits[pre]); i++) tab[used_entries + i] = 0;
                    used_entries += 1u << sub_bits[pre];
                }
        for (int sy = 0; sy < n; sy++) {
            const int l = len[sy];
            if (!l) continue;
            uint32_t e;
            if (lit_len) {
                if (sy < 256) e = make((uint32_t)l, LITERAL, 0, (uint32_t)sy);
                else if (sy == 256) e = make((uint32_t)l, END, 0, 0);
                else if (sy <= 285) e = make((uint32_t)l, BASE, kLenExtra[sy - 257], kLenBase[sy - 257]);
                else e = 0;  // 286, 287: in the fixed code's alphabet, never valid
            } else {
                e = sy <= 29 ? make((uint32_t)l, BASE, kDistExtra[sy], kDistBase[sy]) : 0;
            }
            const uint32_t r = rev_of[sy];
            if (l <= root) {
                for (uint32_t i = r; i < root_size; i += 1u << l) tab[i] = e;
            } else {
                const uint32_t pre = r & (root_size - 1), sb = sub_bits[pre], ofs = value_of(tab[pre]);
                for (uint32_t i = r >> root; i < (1u << sb); i += 1u << (l - root)) tab[ofs + i] = e;
            }
        }
        return true;
    }
};

// a vector that does not fill what resize() adds: the inflater writes through raw pointers into room it has resized
// into, and resizes back to what it wrote
template <class T>
struct NoInit : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = NoInit<U>;
    };
    template <class U>
    void construct(U* p) {
        ::new ((void*)p) U;
    }
    template <class U, class A0, class... A>
    void construct(U* p, A0&& a0, A&&... a) {
        ::new ((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...);
    }
};
using SymVec = std::vector<uint16_t, NoInit<uint16_t>>;

// output of one chunk: symbols < 256 are bytes, 256 + i = byte i of the window before the chunk
struct Symbols {
    SymVec s;
    uint64_t member_start = 0;  // symbols before this index belong to an earlier gzip member
    uint64_t marker_end = 0;    // no symbol at or after this index is a window reference (an upper bound of the last one)
};

struct Member {  // a member that ended inside a chunk
    uint64_t end_symbol;  // symbols [.., end_symbol) of the chunk belong to it
    uint32_t crc, isize;
};

enum Stop { AT_TARGET, AT_LIMIT, AT_EOF, FAILED };

struct Inflater {
    BitReader br;
    Symbols* out;
    std::vector<Member>* members;
    bool text_only;  // entry-point validation: stop with an error at the first byte that is not text
    uint64_t stop_symbols = ~0ull;  // validation: leave a block once this many symbols are out
    std::string err;
    FastHuff lit, dist;
    static constexpr int kLitRoot = 10, kDistRoot = 8;
    Inflater(const uint8_t* b, const uint8_t* e, uint64_t bitpos, Symbols* o, std::vector<Member>* m, bool text)
        : br(b, e, bitpos), out(o), members(m), text_only(text) {}

    bool fail(const char* m) {
        if (err.empty()) err = m;
        return false;
    }
    static bool is_text(uint32_t c) { return c == '\n' || c == '\r' || c == '\t' || (c >= 32 && c < 127); }

    bool read_dynamic() {
        const uint32_t hlit = br.get(5) + 257, hdist = br.get(5) + 1, hclen = br.get(4) + 4;
        if (hlit > 286 || hdist > 30) return fail("bad block header");
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (uint32_t i = 0; i < hclen; i++) cl[order[i]] = (uint8_t)br.get(3);
        Huff clh;
        if (!clh.build(cl, 19, false)) return fail("bad code-length code");
        uint8_t lens[286 + 30] = {0};
        uint32_t i = 0;
        while (i < hlit + hdist) {
            br.refill();
            const uint16_t e = clh.tab[br.peek(clh.bits)];
            if (!e) return fail("bad code-length symbol");
            br.consume(e & 15);
            const uint32_t sym = e >> 4;
            if (sym < 16) {
                lens[i++] = (uint8_t)sym;
            } else {
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return fail("repeat without a previous length");
                    val = lens[i - 1];
                    rep = 3 + br.get(2);
                } else if (sym == 17) {
                    rep = 3 + br.get(3);
                } else {
                    rep = 11 + br.get(7);
                }
                if (i + rep > hlit + hdist) return fail("length repeat overruns");
                while (rep--) lens[i++] = (uint8_t)val;
            }
        }
        if (lens[256] == 0) return fail("no end-of-block code");
        if (!lit.build(lens, (int)hlit, false, kLitRoot, true)) return fail("bad literal/length code");
        if (!dist.build(lens + hlit, (int)hdist, true, kDistRoot, false)) return fail("bad distance code");
        return !br.overrun_now();
    }
    void set_fixed() {
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        lit.build(l, 288, false, kLitRoot, true);
        uint8_t d[32];
        for (int i = 0; i < 32; i++) d[i] = 5;
        dist.build(d, 32, true, kDistRoot, false);  // 32 five-bit codes; 30 and 31 never occur in valid data (their entries are invalid)
    }
    // While a block body runs, out->s.size() is the ROOM (resized into without initialisation) and n_ the count; the
    // body leaves with size() == count again.
    uint64_t n_ = 0;
    void open_room(uint64_t want) {
        if (out->s.size() < n_ + want) out->s.resize(std::max<uint64_t>(n_ + want, out->s.size() + out->s.size() / 2 + (1 << 16)));
    }
    uint32_t decode(const FastHuff& h) {
        uint32_t e = h.tab[(uint32_t)br.buf & ((1u << h.root) - 1u)];
        if (FastHuff::kind_of(e) == FastHuff::SUB)
            e = h.tab[FastHuff::value_of(e) + ((uint32_t)(br.buf >> h.root) & ((1u << FastHuff::extra_of(e)) - 1u))];
        br.consume((int)FastHuff::len_of(e));  // (an invalid entry has length 0)
        return e;
    }
    template <bool VALIDATE>
    bool copy(uint32_t len, uint32_t d) {
        const uint64_t have = n_ - out->member_start;  // symbols of the current member inside this chunk
        if (d > kWindow) return fail("distance beyond the window");
        if (d > have) {
            // reaches before the chunk: only legal while the current member started before the chunk
            if (out->member_start != 0) return fail("reference before the start of its member");
        }
        uint16_t* s0 = out->s.data();
        const uint64_t n0 = n_;
        uint16_t* o = s0 + n0;
        if (d <= n0) {
            const uint16_t* src = o - d;
            const bool may_mark = n0 - d < out->marker_end;  // the source range can hold window references
            if (d >= 8) {  // eight symbols at a time; the room reaches 320 symbols past the count (overlap: d < len repeats)
                for (uint32_t j = 0; j < len; j += 8) memcpy(o + j, src + j, 16);
            } else {
                for (uint32_t j = 0; j < len; j++) o[j] = src[j];
            }
            // (an upper bound, not a scan: the resolve pass maps everything below marker_end through one table, which
            //  leaves plain bytes alone; in FASTQ the references live on to the end of a chunk anyway)
            if (may_mark) out->marker_end = n0 + len;
            if (VALIDATE && text_only)
                for (uint32_t j = 0; j < len; j++)
                    if (o[j] < 256 && !is_text(o[j])) return fail("binary data");
            n_ = n0 + len;
            return true;
        }
        for (uint32_t j = 0; j < len; j++) {
            const uint64_t n = n0 + j;
            uint16_t v;
            if (d <= n) {
                v = s0[n - d];
            } else {
                const uint64_t back = d - n;  // bytes before the chunk: window index kWindow - back
                v = (uint16_t)(256 + (kWindow - back));
            }
            if (v >= 256) out->marker_end = n + 1;
            if (VALIDATE && text_only && v < 256 && !is_text(v)) return fail("binary data");
            o[j] = v;
        }
        n_ = n0 + len;
        return true;
    }
    // VALIDATE: entry-point validation (text only, leave after stop_symbols)
    template <bool VALIDATE>
    bool block_body() {
        n_ = out->s.size();
        struct Close {
            Inflater* f;
            ~Close() { f->out->s.resize(f->n_); }
        } close{this};
        open_room(1 << 16);
        for (;;) {
            // past the end of the input the reader supplies zero bits: a code set in which those decode to a literal
            // would otherwise never stop (entry-point validation runs on arbitrary bit positions)
            if (br.p > br.end + 8) return fail("truncated");
            if (VALIDATE && n_ >= stop_symbols) return true;  // validation: enough seen (run() stops at its limit next)
            if (out->s.size() < n_ + 320) open_room(1 << 16);
            uint16_t* o = out->s.data();
            br.refill();  // 56 bits or more: a symbol takes at most 15, a length 5 more; the distance refills again
            uint32_t e = decode(lit);
            bool again = false;
            while (FastHuff::kind_of(e) == FastHuff::LITERAL) {
                const uint32_t c = FastHuff::value_of(e);
                if (VALIDATE && text_only && !is_text(c)) return fail("binary data");
                o[n_++] = (uint16_t)c;
                if (br.cnt < 20) {  // fewer than a symbol and a length's extra bits left: back to the refill
                    again = true;
                    break;
                }
                e = decode(lit);
            }
            if (again) continue;
            const uint32_t k = FastHuff::kind_of(e);
            if (k == FastHuff::END) return !br.overrun_now() || fail("truncated");
            if (k != FastHuff::BASE) return fail("bad literal/length symbol");
            // (at least 5 bits are left for the length's extra bits: the symbol was decoded from 20 or more)
            const uint32_t lx = FastHuff::extra_of(e);
            const uint32_t len = FastHuff::value_of(e) + br.peek((int)lx);
            br.consume((int)lx);
            if (br.cnt < 28) br.refill();  // a distance code and its extra bits: 15 + 13
            const uint32_t de = decode(dist);
            if (FastHuff::kind_of(de) != FastHuff::BASE) return fail("bad distance symbol");
            const uint32_t dx = FastHuff::extra_of(de);
            const uint32_t d = FastHuff::value_of(de) + br.peek((int)dx);
            br.consume((int)dx);
            // the common case inline: the source lies inside the chunk and its member, eight or more symbols back
            if (!VALIDATE && d >= 8 && d <= n_ - out->member_start) {
                uint16_t* q = o + n_;
                const uint16_t* src = q - d;
                for (uint32_t j = 0; j < len; j += 8) memcpy(q + j, src + j, 16);
                if (n_ - d < out->marker_end) out->marker_end = n_ + len;  // (an upper bound: see copy())
                n_ += len;
                continue;
            }
            if (!copy<VALIDATE>(len, d)) return false;
            if (VALIDATE && br.overrun_now()) return fail("truncated");
        }
    }
    bool inflate_block_body() { return (text_only || stop_symbols != ~0ull) ? block_body<true>() : block_body<false>(); }
    // gzip member header at the (byte-aligned) reader position; false on anything unexpected
    bool skip_member_header() {
        br.align_byte();
        auto byte = [&]() { return br.get(8); };
        if (byte() != 0x1f || byte() != 0x8b || byte() != 8) return fail("not a gzip member");
        const uint32_t flg = byte();
        for (int i = 0; i < 6; i++) byte();
        if (flg & 0xe0) return fail("reserved gzip flags");
        if (flg & 4) {
            uint32_t xlen = byte();
            xlen |= byte() << 8;
            while (xlen--) byte();
        }
        if (flg & 8)
            while (byte() != 0 && !br.overrun_now()) {
            }
        if (flg & 16)
            while (byte() != 0 && !br.overrun_now()) {
            }
        if (flg & 2) {
            byte();
            byte();
        }
        return !br.overrun_now() || fail("truncated header");
    }

    // Inflate block after block from the current position (a block boundary).  Stops
    //   AT_TARGET  exactly at bit position `target` (before that block's header),
    //   AT_LIMIT   at the first block boundary at or after `limit_bits`, or once max_symbols are out (validation),
    //   AT_EOF     after the last member of the file.
    // *next_bit = position of the next unread block header (or of the end of the file).
    Stop run(uint64_t target, uint64_t limit_bits, uint64_t max_symbols, uint64_t* next_bit) {
        stop_symbols = max_symbols;
        if (max_symbols == ~0ull && limit_bits != ~0ull) {  // a real chunk: room for ~5x its compressed size up front
            const uint64_t at = br.bitpos();
            if (limit_bits > at) out->s.reserve(out->s.size() + (limit_bits - at) / 8 * 5 + (1 << 16));
        }
        for (;;) {
            const uint64_t at = br.bitpos();
            *next_bit = at;
            if (at == target) return AT_TARGET;
            if (at >= limit_bits || out->s.size() >= max_symbols) return AT_LIMIT;
            const uint32_t bfinal = br.get(1), btype = br.get(2);
            if (btype == 0) {
                br.align_byte();
                const uint32_t len = br.get(16), nlen = br.get(16);
                if ((len ^ nlen) != 0xffffu) return fail("bad stored block"), FAILED;
                for (uint32_t i = 0; i < len; i++) {
                    if (br.p > br.end + 8) return fail("truncated"), FAILED;
                    const uint32_t c = br.get(8);
                    if (text_only && !is_text(c)) return fail("binary data"), FAILED;
                    out->s.push_back((uint16_t)c);
                }
                if (br.overrun_now()) return fail("truncated"), FAILED;
            } else if (btype == 1) {
                set_fixed();
                if (!inflate_block_body()) return FAILED;
            } else if (btype == 2) {
                if (!read_dynamic() || !inflate_block_body()) return FAILED;
            } else {
                return fail("bad block type"), FAILED;
            }
            if (out->s.size() >= max_symbols) {  // validation stopped inside the block
                *next_bit = br.bitpos();
                return AT_LIMIT;
            }
            if (bfinal) {
                br.align_byte();
                Member m;
                m.end_symbol = out->s.size();
                m.crc = br.get(16);
                m.crc |= br.get(16) << 16;
                m.isize = br.get(16);
                m.isize |= br.get(16) << 16;
                if (br.overrun_now()) return fail("truncated trailer"), FAILED;
                if (members) members->push_back(m);
                // another member, trailing zero padding, or the end of the file
                const uint64_t total_bits = (uint64_t)(br.end - br.base) * 8;
                for (;;) {
                    if (br.bitpos() >= total_bits) {
                        *next_bit = total_bits;
                        return AT_EOF;
                    }
                    br.refill();
                    if (br.peek(8) != 0) break;
                    br.consume(8);
                }
                if (!skip_member_header()) return FAILED;
                out->member_start = out->s.size();
            }
        }
    }
};

// does a dynamic block start at `bitpos`, followed by enough plausible text?
inline bool entry_point_ok(const uint8_t* base, const uint8_t* end, uint64_t bitpos) {
    {  // cheap rejections first: BFINAL = 0, BTYPE = 2, sane counts
        BitReader br(base, end, bitpos);
        if (br.get(1) != 0 || br.get(2) != 2) return false;
        const uint32_t hlit = br.get(5) + 257, hdist = br.get(5) + 1;
        if (hlit > 286 || hdist > 30) return false;
    }
    Symbols s;
    Inflater inf(base, end, bitpos, &s, nullptr, true);
    uint64_t next = 0;
    const Stop st = inf.run(~0ull, ~0ull, 64 * 1024, &next);
    return st == AT_LIMIT && s.s.size() >= 64 * 1024;
}

inline uint64_t find_entry_point(const uint8_t* base, const uint8_t* end, uint64_t from_bit, uint64_t to_bit) {
    for (uint64_t b = from_bit; b < to_bit; b++)
        if (entry_point_ok(base, end, b)) return b;
    return ~0ull;
}

// One gzip file, decompressed round by round on `threads` threads.
class ParallelGunzip {
   public:
    ~ParallelGunzip() { close(); }
    bool open(const std::string& path, unsigned threads, uint64_t chunk_bytes = 2ull << 20) {
        threads_ = std::max(1u, threads);
        chunk_ = std::max<uint64_t>(chunk_bytes, 64 * 1024);
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return false;
        struct stat st;
        if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) return close(), false;
        size_ = (uint64_t)st.st_size;
        void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) return close(), false;
        data_ = (const uint8_t*)m;
        Symbols dummy;
        Inflater inf(data_, data_ + size_, 0, &dummy, nullptr, false);
        if (!inf.skip_member_header()) return close(), false;
        pos_bit_ = inf.br.bitpos();
        window_.clear();
        crc_ = crc32(0L, Z_NULL, 0);
        member_len_ = 0;
        return true;
    }
    void close() {
        if (data_) munmap((void*)data_, size_);
        data_ = nullptr;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }
    const std::string& error() const { return err_; }
    bool crc_mismatch() const { return crc_mismatch_; }  // the last failure: data decoded, a member's CRC-32 / ISIZE did not match
    bool at_end() const { return done_; }

    // Decompress the next round; the pieces come back in file order.  false + error() on failure.
    bool next_round(std::vector<std::vector<uint8_t>>& pieces) {
        pieces.clear();
        if (done_) return true;
        const uint8_t* end = data_ + size_;
        const uint64_t first_byte = pos_bit_ >> 3;
        // twice as many chunks as threads, handed out from a counter: the chunks of a round take unequal time, and a
        // thread per chunk left the fast ones waiting at every one of the round's three barriers
        const unsigned T = (unsigned)std::min<uint64_t>(2ull * threads_, std::max<uint64_t>(1, (size_ - first_byte + chunk_ - 1) / chunk_));
        auto for_each = [&](size_t n, size_t first, const std::function<void(size_t)>& fn) {
            std::atomic<size_t> next{first};
            auto work = [&] {
                for (size_t k; (k = next.fetch_add(1)) < n;) fn(k);
            };
            std::vector<std::thread> th;
            const size_t nt = std::min<size_t>(threads_, n > first ? n - first : 0);
            for (size_t t = 1; t < nt; t++) th.emplace_back(work);
            if (nt) work();
            for (auto& t : th) t.join();
        };
        const bool trace = getenv("MTSV_PGZIP_TRACE") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        // 1. entry points (chunk 0 starts at pos_bit_)
        std::vector<uint64_t> entry(T, ~0ull);
        entry[0] = pos_bit_;
        for_each(T, 1, [&](size_t i) {
            const uint64_t from = (first_byte + i * chunk_) * 8;
            entry[i] = find_entry_point(data_, end, from, std::min<uint64_t>(from + chunk_ * 8, size_ * 8));
        });
        const double t1 = now();
        const uint64_t limit_bits = std::min<uint64_t>((first_byte + (uint64_t)T * chunk_) * 8, size_ * 8);
        // 2. inflate every chunk up to the next valid entry point (symbol buffers are kept from round to round:
        // fresh pages for ~100 MB of symbols per round cost more than the inflation itself)
        if (ch_.size() < T) ch_.resize(T);
        std::vector<Chunk>& ch = ch_;
        for (unsigned i = 0; i < T; i++) {
            ch[i].sym.s.clear();
            ch[i].sym.member_start = 0;
            ch[i].sym.marker_end = 0;
            ch[i].members.clear();
            ch[i].next_bit = 0;
            ch[i].stop = FAILED;
            ch[i].err.clear();
            ch[i].used = false;
        }
        auto inflate_chunk = [&](unsigned i, unsigned target_chunk) {
            Chunk& c = ch[i];
            Inflater inf(data_, end, c.used ? c.next_bit : entry[i], &c.sym, &c.members, false);
            if (c.used) inf.out->member_start = c.sym.member_start;
            c.used = true;
            const uint64_t target = target_chunk < T ? entry[target_chunk] : ~0ull;
            c.stop = inf.run(target, limit_bits, ~0ull, &c.next_bit);
            c.err = inf.err;
        };
        std::vector<unsigned> order;  // chunks that take part, in file order
        for (unsigned i = 0; i < T; i++)
            if (entry[i] != ~0ull) order.push_back(i);
        for_each(order.size(), 0, [&](size_t k) { inflate_chunk(order[k], k + 1 < order.size() ? order[k + 1] : T); });
        // chains: chunk order[k] must have stopped AT_TARGET on order[k+1]; if it ran past it (a false entry point),
        // it carries on to the one after (serially: rare)
        const double t2 = now();
        // Chunk order[k] is good once its predecessor ARRIVED on its entry point (AT_TARGET).  A chunk that did not
        // stop there has run over a false entry point -- run() only returns at its target, at the round limit, at the
        // end of the file or on an error -- and then holds everything up to the limit itself: the chunks behind it
        // are dropped.
        std::vector<unsigned> kept;
        for (size_t k = 0; k < order.size(); k++) {
            const unsigned i = order[k];
            kept.push_back(i);
            if (ch[i].stop == FAILED) return fail("chunk " + std::to_string(i) + ": " + ch[i].err);
            if (ch[i].stop != AT_TARGET) break;
        }
        const Chunk& last = ch[kept.back()];
        if (last.stop == AT_TARGET) return fail("internal: the last chunk of a round stopped on an entry point");
        // 3. resolve: windows first (serially), then the bodies (in parallel)
        std::vector<std::vector<uint8_t>> win(kept.size() + 1);
        win[0] = window_;
        for (size_t k = 0; k < kept.size(); k++) {
            const Symbols& s = ch[kept[k]].sym;
            std::vector<uint8_t>& w = win[k + 1];
            const std::vector<uint8_t>& pw = win[k];
            const uint64_t n = s.s.size();
            // only bytes of the member that is open at the end of the chunk can be referred to later
            const uint64_t take = std::min<uint64_t>(n - s.member_start, kWindow);
            if (take < kWindow && s.member_start == 0) {  // the tail of the previous window stays in view
                const uint64_t keep = std::min<uint64_t>(kWindow - take, pw.size());
                w.assign(pw.end() - (ptrdiff_t)keep, pw.end());
            }
            for (uint64_t j = n - take; j < n; j++) {
                uint16_t v = s.s[j];
                if (v >= 256) {
                    const uint64_t idx = v - 256;  // position in a full 32 KiB window that ends where the chunk starts
                    if (idx + pw.size() < kWindow) return fail("reference before the start of the stream");
                    v = pw[idx - (kWindow - pw.size())];
                }
                w.push_back((uint8_t)v);
            }
        }
        const double t3 = now();
        pieces.resize(kept.size());
        std::vector<std::vector<std::pair<uint32_t, uint64_t>>> crcs(kept.size());  // per chunk: (crc, length) of every member part
        std::vector<std::string> errs(kept.size());
        for_each(kept.size(), 0, [&](size_t k) {
                    const Chunk& c = ch[kept[k]];
                    const std::vector<uint8_t>& pw = win[k];
                    std::vector<uint8_t>& o = pieces[k];
                    const uint64_t n = c.sym.s.size();
                    o.resize(n);
                    const uint16_t* sy = c.sym.s.data();
                    const uint64_t me = std::min<uint64_t>(c.sym.marker_end, n);
                    if (pw.size() == kWindow && me > 4096) {
                        // a full window before the chunk (every chunk but the stream's first): one table maps bytes to
                        // themselves and window references to the window's bytes -- no branch per symbol (in FASTQ the
                        // references live on through most of a chunk)
                        std::vector<uint8_t> lut(256 + kWindow);
                        for (uint32_t v = 0; v < 256; v++) lut[v] = (uint8_t)v;
                        memcpy(lut.data() + 256, pw.data(), kWindow);
                        uint8_t* od = o.data();
                        const uint8_t* lt = lut.data();
                        for (uint64_t j = 0; j < me; j++) od[j] = lt[sy[j]];
                    } else {
                        for (uint64_t j = 0; j < me; j++) {
                            uint16_t v = sy[j];
                            if (v >= 256) {
                                const uint64_t idx = v - 256;
                                if (idx + pw.size() < kWindow) {
                                    errs[k] = "reference before the start of the stream";
                                    return;
                                }
                                v = pw[idx - (kWindow - pw.size())];
                            }
                            o[j] = (uint8_t)v;
                        }
                    }
                    uint8_t* ob = o.data();
                    for (uint64_t j = me; j < n; j++) ob[j] = (uint8_t)sy[j];  // plain narrowing: vectorises
                    uint64_t from = 0;
                    for (const Member& m : c.members) {
                        crcs[k].emplace_back(crc_of(o.data() + from, m.end_symbol - from), m.end_symbol - from);
                        from = m.end_symbol;
                    }
                    crcs[k].emplace_back(crc_of(o.data() + from, n - from), n - from);
        });
        for (auto& e : errs)
            if (!e.empty()) return fail(e);
        // member CRCs across chunks
        for (size_t k = 0; k < kept.size(); k++) {
            const Chunk& c = ch[kept[k]];
            for (size_t m = 0; m <= c.members.size(); m++) {
                crc_ = (uint32_t)crc32_combine(crc_, crcs[k][m].first, (z_off_t)crcs[k][m].second);
                member_len_ += crcs[k][m].second;
                if (m < c.members.size()) {
                    if (crc_ != c.members[m].crc || (uint32_t)member_len_ != c.members[m].isize) {
                        crc_mismatch_ = true;
                        return fail("CRC mismatch: corrupt gzip data");
                    }
                    crc_ = (uint32_t)crc32(0L, Z_NULL, 0);
                    member_len_ = 0;
                }
            }
        }
        if (trace) {
            uint64_t nsym = 0;
            for (unsigned i : kept) nsym += ch[i].sym.s.size();
            fprintf(stderr, "[pgzip] round of %u chunks (%zu kept): entry %.1f ms, inflate %.1f ms, windows %.1f ms, resolve+crc %.1f ms, %.1f MB out\n", T,
                    kept.size(), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (now() - t3) * 1e3, nsym / 1e6);
        }
        window_ = win[kept.size()];
        if (last.stop == AT_EOF) {
            done_ = true;
            window_.clear();
        } else {
            pos_bit_ = last.next_bit;
        }
        return true;
    }

   private:
    struct Chunk {
        Symbols sym;
        std::vector<Member> members;
        uint64_t next_bit = 0;
        Stop stop = FAILED;
        std::string err;
        bool used = false;
    };
    std::vector<Chunk> ch_;
    static uint32_t crc_zlib(const uint8_t* p, uint64_t n) {
        uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
        while (n) {
            const uInt step = (uInt)std::min<uint64_t>(n, 1u << 30);
            c = (uint32_t)crc32(c, p, step);
            p += step;
            n -= step;
        }
        return c;
    }
#if defined(__x86_64__)
    // CRC-32 (the gzip polynomial, reflected) by carry-less multiplication: four 128-bit lanes folded 512 bits at a
    // time, then 128 -> 64 -> 32 bits and a Barrett reduction (the scheme of Intel's "Fast CRC Computation Using
    // PCLMULQDQ" paper).  n >= 64 and a multiple of 16; `crc` and the result are the raw register (not inverted).
    // Used only after it has reproduced zlib's crc32 on a test pattern in this process (crc_of).
    __attribute__((target("pclmul,sse4.1"))) static uint32_t crc_clmul(const uint8_t* buf, uint64_t len, uint32_t crc) {
        const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
        const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
        const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124ll);
        const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
        __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
        x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
        x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
        x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
        x0 = k1k2;
        buf += 64;
        len -= 64;
        while (len >= 64) {
            x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
            x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
            x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
            x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
            x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
            x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
            x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
            x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
            y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
            y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
            y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
            y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
            x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
            x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
            x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
            x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
            buf += 64;
            len -= 64;
        }
        x0 = k3k4;
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
        while (len >= 16) {
            x2 = _mm_loadu_si128((const __m128i*)buf);
            x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
            x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
            x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
            buf += 16;
            len -= 16;
        }
        x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
        x3 = _mm_setr_epi32(~0, 0, ~0, 0);
        x1 = _mm_srli_si128(x1, 8);
        x1 = _mm_xor_si128(x1, x2);
        x0 = k5k0;
        x2 = _mm_srli_si128(x1, 4);
        x1 = _mm_and_si128(x1, x3);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_xor_si128(x1, x2);
        x0 = poly;
        x2 = _mm_and_si128(x1, x3);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
        x2 = _mm_and_si128(x2, x3);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x1 = _mm_xor_si128(x1, x2);
        return (uint32_t)_mm_extract_epi32(x1, 1);
    }
    static bool clmul_ok() {  // the CPU has the instruction and the code above agrees with zlib
        static const bool ok = [] {
            if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
            std::vector<uint8_t> t(4096 + 7);
            uint32_t v = 12345;
            for (auto& b : t) {
                v = v * 1664525u + 1013904223u;
                b = (uint8_t)(v >> 24);
            }
            for (uint64_t n : {64ull, 80ull, 1024ull, 4096ull})
                for (uint32_t seed : {0u, 0xdeadbeefu}) {
                    const uint32_t want = (uint32_t)crc32(seed, t.data() + 3, (uInt)n);
                    if ((crc_clmul(t.data() + 3, n, ~seed) ^ 0xffffffffu) != want) return false;
                }
            return true;
        }();
        return ok;
    }
#endif
    static uint32_t crc_of(const uint8_t* p, uint64_t n) {
#if defined(__x86_64__)
        if (n >= 256 && clmul_ok()) {
            const uint64_t body = n & ~15ull;
            uint32_t c = crc_clmul(p, body, 0xffffffffu) ^ 0xffffffffu;
            if (n > body) c = (uint32_t)crc32(c, p + body, (uInt)(n - body));
            return c;
        }
#endif
        return crc_zlib(p, n);
    }
    bool fail(const std::string& m) {
        err_ = m;
        return false;
    }
    int fd_ = -1;
    const uint8_t* data_ = nullptr;
    uint64_t size_ = 0, chunk_ = 2 << 20, pos_bit_ = 0, member_len_ = 0;
    unsigned threads_ = 1;
    std::vector<uint8_t> window_;
    uint32_t crc_ = 0;
    bool done_ = false;
    std::string err_;
    bool crc_mismatch_ = false;
};

}  // namespace mtsv_pgzip

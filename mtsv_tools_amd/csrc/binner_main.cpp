// mtsv-binner -- drop-in command line of the reference binary (src/bin/mtsv-binner.rs) over
// libmtsv_amd: same flags and defaults (:26-113), same validation (:140-262), same exit codes
// (0 ok, 2 query error, 3 no results path, 4 resume error, 11 write error, 12 read error;
// invalid numbers abort like the reference's panics, exit 101), same results grammar
// (src/binner.rs:310-379), same resume rule (:347-411).  Extras: --device / --devices, --batch-reads,
// --parse-only (ingest check: prints record / base counts and checksums, needs no index or GPU).
// The reads of a batch are processed on the GPU; result lines are written in input order (the
// reference's order is unspecified: vendor/cue/src/lib.rs:67-74).
//
// Several GPUs of one node:
//   --devices 0,1,..  with one --index: the index is loaded once and made resident on every listed device;
//                     worker threads with a workspace each (three per entry, MTSV_CLI_WORKERS) pull groups of read
//                     batches (README.md:69-73 workflow)
//   --index a,b,..    the chunks of a database cut by mtsv-chunk, chunk k on the k-th listed device (round
//                     robin): every chunk sees every batch and the hits are merged per read, so the one results
//                     file equals what mtsv-collapse makes of the per-chunk files (README.md:189,
//                     collapse.rs:597-625: smallest edit per read and TaxId)
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/mtsv_amd.h"
#include "fastx_ingest.hpp"

namespace {

bool g_verbose = false;
void logmsg(const char* level, const std::string& msg) {
    if (!g_verbose && !strcmp(level, "DEBUG")) return;
    char ts[32];
    time_t t = time(nullptr);
    strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", localtime(&t));
    printf("[%s %s mtsv_binner] %s\n", level, ts, msg.c_str());  // util.rs:10-24: stdout
    fflush(stdout);
}
[[noreturn]] void panic(const std::string& msg) {  // the reference's expect()/panic!() paths
    fprintf(stderr, "thread 'main' panicked: %s\n", msg.c_str());
    exit(101);
}

// open_maybe_gz (binner.rs:21-33): gzip magic sniff; zlib reads both transparently
struct Input {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, len = 0;
    bool eof = false;
    bool open(const std::string& path) {
        f = gzopen(path.c_str(), "rb");
        if (!f) return false;
        gzbuffer(f, 1 << 20);
        buf.resize(1 << 20);
        return true;
    }
    ~Input() {
        if (f) gzclose(f);
    }
    // returns false at EOF with no data; sets *err on a read error
    bool getline(std::string& out, bool* err) {
        out.clear();
        for (;;) {
            if (pos == len) {
                if (eof) return !out.empty();
                int n = gzread(f, buf.data(), (unsigned)buf.size());
                if (n < 0) {
                    *err = true;
                    return false;
                }
                if (n == 0) {
                    eof = true;
                    return !out.empty();
                }
                pos = 0;
                len = (size_t)n;
            }
            char* nl = (char*)memchr(buf.data() + pos, '\n', len - pos);
            if (nl) {
                out.append(buf.data() + pos, nl - (buf.data() + pos));
                pos = nl - buf.data() + 1;
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            out.append(buf.data() + pos, len - pos);
            pos = len;
        }
    }
};

struct Record {
    std::string id, seq;
};

// bio::io::{fasta,fastq} readers as used at binner.rs:169-199; id = first token of the header
struct FastxReader {
    Input in;
    bool fastq;
    std::string pending;  // a header line already consumed (FASTA)
    bool have_pending = false;
    bool error = false;
    std::string err_msg;

    static std::string first_token(const std::string& h) {
        size_t e = h.find_first_of(" \t", 1);
        return h.substr(1, e == std::string::npos ? std::string::npos : e - 1);
    }
    bool next(Record& r) {
        std::string line;
        bool ioerr = false;
        if (fastq) {
            do {
                if (!in.getline(line, &ioerr)) {
                    if (ioerr) fail("read error");
                    return false;
                }
            } while (line.empty());
            if (line[0] != '@') return fail("Expected @ at record start.");
            r.id = first_token(line);
            r.seq.clear();
            for (;;) {
                if (!in.getline(line, &ioerr)) return fail("Incomplete record.");
                if (!line.empty() && line[0] == '+') break;
                r.seq += line;
            }
            size_t q = 0;
            while (q < r.seq.size()) {
                if (!in.getline(line, &ioerr)) return fail("Incomplete record.");
                q += line.size();
            }
            if (r.seq.empty()) in.getline(line, &ioerr);  // empty quality line of an empty read
            if (q != r.seq.size() && !r.seq.empty()) return fail("Unequal length of sequence an qualities.");
            return true;
        }
        if (!have_pending) {
            do {
                if (!in.getline(line, &ioerr)) {
                    if (ioerr) fail("read error");
                    return false;
                }
            } while (line.empty());
            pending = line;
        }
        have_pending = false;
        if (pending.empty() || pending[0] != '>') return fail("Expected > at record start.");
        r.id = first_token(pending);
        r.seq.clear();
        while (in.getline(line, &ioerr)) {
            if (!line.empty() && line[0] == '>') {
                pending = line;
                have_pending = true;
                break;
            }
            r.seq += line;
        }
        if (ioerr) return fail("read error");
        return true;
    }
    bool fail(const char* m) {
        error = true;
        err_msg = m;
        return false;
    }
};

// resume_offset_from_results (mtsv-binner.rs:366-396): index of the last input record whose id
// appears in the results file, plus one
int resume_offset(const std::string& results, const std::string& input, bool fastq, uint64_t* off) {
    FILE* f = fopen(results.c_str(), "rb");
    if (!f) return -1;
    std::unordered_set<std::string> ids;
    std::string line;
    char buf[1 << 16];
    auto flush_line = [&](std::string& l) -> bool {
        size_t a = l.find_first_not_of(" \t\r\n");
        if (a != std::string::npos) {
            size_t c = l.rfind(':');
            if (c == std::string::npos || c == 0) return false;  // "Missing read id"
            ids.insert(l.substr(0, c));
        }
        l.clear();
        return true;
    };
    while (fgets(buf, sizeof buf, f)) {
        line += buf;
        if (!line.empty() && line.back() == '\n') {
            line.pop_back();
            if (!flush_line(line)) {
                fclose(f);
                return -1;
            }
        }
    }
    if (!line.empty() && !flush_line(line)) {
        fclose(f);
        return -1;
    }
    fclose(f);
    FastxReader rd;
    rd.fastq = fastq;
    if (!rd.in.open(input)) return -1;
    Record r;
    uint64_t idx = 0, last = 0;
    bool any = false;
    while (rd.next(r)) {
        if (ids.count(r.id)) {
            last = idx;
            any = true;
        }
        idx++;
    }
    if (rd.error) return -1;
    *off = any ? last + 1 : 0;
    return 0;
}

struct Args {
    std::string fasta, fastq, index, results, output_format = "default";
    std::string threads = "4", edit = "0.13", seed_size = "18", seed_interval = "15", min_seed = "0.015",
                max_hits = "2000", tune_max_hits = "200", max_assign, max_cand, read_offset = "0";
    bool verbose = false, force = false, parse_only = false;
    std::vector<int> devices{0};
    uint64_t batch_reads = 1u << 17;  // parser blocks of ~40 MB; the GPU workers take up to 1 Mi reads of them per library call. End to end on 32 M reads: 64 Ki .., 128 Ki 70 M reads/s, 256 Ki 54, 512 Ki 37
};

[[noreturn]] void usage_error(const std::string& m) {
    fprintf(stderr, "error: %s\n\nUSAGE:\n    mtsv-binner [FLAGS] [OPTIONS] --index <INDEX> <--fasta <FASTA>|--fastq <FASTQ>>\n", m.c_str());
    exit(1);  // clap usage errors
}

uint64_t parse_usize(const std::string& s, const char* what) {
    if (s.empty()) panic(what);
    char* e = nullptr;
    errno = 0;
    if (s[0] == '-') panic(what);
    unsigned long long v = strtoull(s.c_str(), &e, 10);
    if (*e || errno) panic(what);
    return v;
}
double parse_f64(const std::string& s, const char* what) {
    char* e = nullptr;
    if (s.empty()) panic(what);
    double v = strtod(s.c_str(), &e);
    if (*e) panic(what);
    return v;
}

}  // namespace

// MTSV_CLI_MARKS=1: the set-up phases on stderr as they end, in ms since the process began
static void setup_mark(const char* what) {
    static const bool on = getenv("MTSV_CLI_MARKS") != nullptr;
    static struct timespec t0 = [] {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return t;
    }();
    if (!on) return;
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    fprintf(stderr, "[cli set-up] %9.3f ms  %s\n", ((t.tv_sec - t0.tv_sec) + (t.tv_nsec - t0.tv_nsec) * 1e-9) * 1e3, what);
}

int main(int argc, char** argv) {
    setup_mark("main");
    Args a;
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i];
        auto val = [&]() -> std::string {
            size_t eq = k.find('=');
            if (k.rfind("--", 0) == 0 && eq != std::string::npos) {
                std::string v = k.substr(eq + 1);
                k = k.substr(0, eq);
                return v;
            }
            if (i + 1 >= argc) usage_error("The argument '" + k + "' requires a value but none was supplied");
            return argv[++i];
        };
        std::string key = k.substr(0, k.find('='));
        if (key == "--fasta") a.fasta = val();
        else if (key == "--fastq") a.fastq = val();
        else if (key == "-i" || key == "--index") a.index = val();
        else if (key == "-m" || key == "--results") a.results = val();
        else if (key == "-t" || key == "--threads") a.threads = val();
        else if (key == "-e" || key == "--edit-rate") a.edit = val();
        else if (key == "--seed-size") a.seed_size = val();
        else if (key == "--seed-interval") a.seed_interval = val();
        else if (key == "--min-seed") a.min_seed = val();
        else if (key == "--max-hits") a.max_hits = val();
        else if (key == "--tune-max-hits") a.tune_max_hits = val();
        else if (key == "--max-assignments") a.max_assign = val();
        else if (key == "--max-candidates") a.max_cand = val();
        else if (key == "--read-offset") a.read_offset = val();
        else if (key == "--output-format") a.output_format = val();
        else if (key == "--force-overwrite") a.force = true;
        else if (key == "-v") a.verbose = true;
        else if (key == "--device" || key == "--devices") {
            a.devices.clear();
            const std::string v = val();
            size_t at = 0;
            while (at <= v.size()) {
                size_t c = v.find(',', at);
                if (c == std::string::npos) c = v.size();
                const std::string tok = v.substr(at, c - at);
                char* e = nullptr;
                const long d = strtol(tok.c_str(), &e, 10);
                if (tok.empty() || *e || d < 0 || d > 1023) usage_error("Invalid value for '" + key + "': a comma-separated list of GPU ordinals is expected");
                a.devices.push_back((int)d);
                at = c + 1;
            }
        } else if (key == "--batch-reads") {
            const std::string v = val();
            char* e = nullptr;
            errno = 0;
            const unsigned long long n = strtoull(v.c_str(), &e, 10);
            if (v.empty() || v[0] == '-' || *e || errno || n < 1 || n > 0x7fffffffull)
                usage_error("Invalid value for '--batch-reads <N>': a number of reads between 1 and 2147483647 is expected");
            a.batch_reads = n;
        }
        else if (key == "--parse-only") a.parse_only = true;
        else if (key == "-h" || key == "--help") {
            printf("mtsv-binner (MI355X) -- flags as the reference: --fasta|--fastq, -i/--index, -m/--results, -t/--threads,\n"
                   "-e/--edit-rate, --seed-size, --seed-interval, --min-seed, --max-hits, --tune-max-hits, --max-assignments,\n"
                   "--max-candidates, --read-offset, --output-format default|long, --force-overwrite, -v;\n"
                   "extras: --devices 0,1,.. (index replicated, reads shared out), --index a,b,.. (database chunks, one per GPU, hits merged),\n"
                   "--batch-reads N, --parse-only\n");
            return 0;
        } else if (key == "-V" || key == "--version") {
            printf("mtsv 2.1.0 (%s)\n", mtsv_version());
            return 0;
        } else
            usage_error("Found argument '" + k + "' which wasn't expected, or isn't valid in this context");
    }
    if (a.fasta.empty() == a.fastq.empty())
        usage_error(a.fasta.empty() ? "The following required arguments were not provided: --fasta <FASTA> | --fastq <FASTQ>"
                                    : "The argument '--fasta <FASTA>' cannot be used with '--fastq <FASTQ>'");
    if (a.index.empty() && !a.parse_only) usage_error("The following required arguments were not provided: --index <INDEX>");
    if (a.output_format != "default" && a.output_format != "long")
        usage_error("'" + a.output_format + "' isn't a valid value for '--output-format <OUTPUT_FORMAT>'");
    g_verbose = a.verbose;

    const bool fastq = a.fasta.empty();
    const std::string input = fastq ? a.fastq : a.fasta;
    const uint64_t t_flag = parse_usize(a.threads, "Invalid number entered for number of threads!");
    // -t sized the reference's worker pool; here it sizes the host helper threads (parse, format)
    unsigned host_threads = (unsigned)std::min<uint64_t>(std::max<uint64_t>(t_flag, 8), std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("MTSV_HOST_THREADS")) host_threads = (unsigned)std::max(1, atoi(e));
    mtsv_params p;
    mtsv_params_default(&p);
    p.edit_rate = parse_f64(a.edit, "Invalid edit proportion entered!");
    logmsg("INFO", "Max Edit Tolerance Proportion: " + a.edit);
    if (p.edit_rate < 0.0 || p.edit_rate > 1.0) panic("Edit tolerance proportion must be between 0 and 1, inclusive");
    uint64_t seed_size = parse_usize(a.seed_size, "Invalid seed size entered!");
    if (seed_size < 16) logmsg("WARN", "Seed size may be small enough that it causes performance issues.");
    else if (seed_size > 24) logmsg("WARN", "Seed size may be large enough that significant results are ignored.");
    uint64_t seed_gap = parse_usize(a.seed_interval, "Invalid seed interval entered!");
    if (seed_gap < 2) logmsg("WARN", "Seed interval may be small enough that it causes performance issues.");
    else if (seed_gap > 10) logmsg("WARN", "Seed interval may be large enough that significant results are ignored.");
    p.min_seed = parse_f64(a.min_seed, "Invalid min seeds entered!");
    if (p.min_seed <= 0.0 || p.min_seed > 1.0) panic("Min seed percent must be between 0 and 1");
    p.max_hits = parse_usize(a.max_hits, "Invalid cutoff for max hits!");
    p.tune_max_hits = parse_usize(a.tune_max_hits, "Invalid cutoff for max hits!");
    p.max_assignments = a.max_assign.empty() ? -1 : (int64_t)parse_usize(a.max_assign, "Invalid number entered for max assignments!");
    p.max_candidates = a.max_cand.empty() ? -1 : (int64_t)parse_usize(a.max_cand, "Invalid number entered for max candidates!");
    uint64_t read_offset = parse_usize(a.read_offset, "Invalid read offset entered!");
    if (seed_size == 0 || seed_size > 0xffffffffull || seed_gap == 0 || seed_gap > 0xffffffffull)
        panic("seed size / interval out of range");  // the reference panics on a zero step (itertools)
    p.seed_size = (uint32_t)seed_size;
    p.seed_interval = (uint32_t)seed_gap;
    const bool long_fmt = a.output_format == "long";

    if (a.results.empty() && !a.parse_only) {
        logmsg("ERROR", "No results path provided!");
        return 3;
    }
    FILE* probe = fopen(a.results.c_str(), "rb");
    const bool exists = probe != nullptr;
    if (probe) fclose(probe);
    const bool append = !a.force && exists;
    uint64_t resume = 0;
    if (a.force) {
        logmsg("INFO", "Forcing overwrite of " + a.results);
    } else if (exists && !a.parse_only) {
        logmsg("INFO", "Existing results detected at " + a.results + "; resuming previous run.");
        if (resume_offset(a.results, input, fastq, &resume) != 0) {
            logmsg("ERROR", "Error computing resume offset");
            return 4;
        }
        logmsg("INFO", "Resuming after read offset " + std::to_string(resume) + " from " + a.results);
    }
    read_offset += resume;

    // get_fastx_and_write_matching_bin_ids (binner.rs:149-217)
    // Producer: block-parallel ingest for plain files (fastx_ingest.hpp), the serial reader for gzip
    // input and from the first irregular block on.  `emit` gets records in input order.
    using mtsv_ingest::ReadBlock;
    FastxReader rd;
    rd.fastq = fastq;
    if (!rd.in.open(input)) {
        logmsg("ERROR", "Error running query: cannot open " + input);
        return 2;
    }
    // batches are recycled (writer -> producer) so that steady state touches no fresh pages
    struct BlockPool {
        std::mutex mu;
        std::vector<std::unique_ptr<ReadBlock>> free;
        std::unique_ptr<ReadBlock> get() {
            std::lock_guard<std::mutex> lk(mu);
            if (free.empty()) return std::make_unique<ReadBlock>();
            auto b = std::move(free.back());
            free.pop_back();
            b->clear();
            return b;
        }
        size_t keep = 8;
        void put(std::unique_ptr<ReadBlock> b) {
            std::lock_guard<std::mutex> lk(mu);
            if (free.size() < keep) free.push_back(std::move(b));
        }
        // a stock of blocks whose (page-locked) base buffers exist before the first query: creating one costs
        // milliseconds per block, which the first second of a run would otherwise spend in its parser threads
        void stock(size_t n, uint64_t base_bytes) {
            keep = std::max(keep, n);
            std::vector<std::thread> th;
            std::vector<std::unique_ptr<ReadBlock>> made(n);
            for (size_t k = 0; k < n; k++)
                th.emplace_back([&made, k, base_bytes] {
                    made[k] = std::make_unique<ReadBlock>();
                    made[k]->bases.reserve(base_bytes);
                });
            for (auto& t : th) t.join();
            std::lock_guard<std::mutex> lk(mu);
            for (auto& b : made) free.push_back(std::move(b));
        }
    } pool;
    uint64_t plain_block_bytes = 0;  // set by block_bytes_for(): the ingest block size of plain input
    std::function<void()> all_emitted;  // called once the last batch has been handed to emit, before the input is closed
    auto produce = [&](uint64_t batch_reads, const std::function<bool(std::unique_ptr<ReadBlock>)>& emit) -> bool {
        auto w = pool.get();
        uint64_t skipped = 0;
        auto full = [&] { return w->n() >= batch_reads || w->bases.size() >= (1ull << 30); };
        mtsv_ingest::ParallelFastx par;
        mtsv_ingest::GzFastx gzpar;  // gzip input: parallel inflate + the same block parsers (MTSV_SERIAL_GZIP=1: zlib's one stream)
        bool use_gz = false;
        // inflating is the expensive part of gzip input: more helpers than plain text needs (measured: 8 threads 3.3 M
        // reads/s, 16 5.5, 32 5.7 against 1.3 on zlib's one stream)
        // (-t / --threads bounds it like every other host-side pool; MTSV_GZ_THREADS=16 is what the 9.0 M reads/s were measured with)
        unsigned gz_threads = host_threads;
        if (const char* e = getenv("MTSV_GZ_THREADS")) gz_threads = (unsigned)std::max(1, atoi(e));
        gz_threads = std::min(gz_threads, std::max(1u, std::thread::hardware_concurrency()));
        // Plain input is cut into blocks of about one batch of reads each (bytes per record estimated from the file's
        // head): a parsed block then IS the batch -- its bases lie in page-locked memory (byte_alloc) that the copy
        // engine reads in place, and the producer thread hands it on without touching the bases again.
        uint64_t ingest_block = 16ull << 20;
        if (getenv("MTSV_INGEST_BLOCK")) ingest_block = strtoull(getenv("MTSV_INGEST_BLOCK"), nullptr, 10);
        else if (plain_block_bytes) ingest_block = plain_block_bytes;
        else if (FILE* hf = fopen(input.c_str(), "rb")) {
            std::vector<char> head(256 << 10);
            const size_t got = fread(head.data(), 1, head.size(), hf);
            fclose(hf);
            uint64_t recs = 0;
            if (got >= 2 && !((uint8_t)head[0] == 0x1f && (uint8_t)head[1] == 0x8b)) {
                if (fastq) {
                    uint64_t lines = 0;
                    for (size_t i = 0; i < got; i++) lines += head[i] == '\n';
                    recs = lines / 4;
                } else {
                    for (size_t i = 0; i < got; i++) recs += head[i] == '>' && (i == 0 || head[i - 1] == '\n');
                }
            }
            // (capped at 96 MiB: a parser thread is slower per byte on larger blocks -- 1 Mi-read blocks of 335 MB parsed at a
            //  third of the rate -- and batches beyond that are put together from several blocks as before)
            if (recs >= 8) ingest_block = std::min<uint64_t>(std::max<uint64_t>((uint64_t)((double)got / (double)recs * (double)batch_reads), 4ull << 20), 96ull << 20);
        }
        plain_block_bytes = ingest_block;
        par.prepare = [&pool](ReadBlock& b) {
            auto stocked = pool.get();
            std::swap(b, *stocked);
        };
        bool serial_from_start = getenv("MTSV_SERIAL_INGEST") != nullptr || !par.open(input, fastq, host_threads, ingest_block);
        if (serial_from_start && !getenv("MTSV_SERIAL_INGEST") && !getenv("MTSV_SERIAL_GZIP") && gzpar.open(input, fastq, gz_threads, ingest_block)) {
            use_gz = true;
            serial_from_start = false;
        }
        if (!serial_from_start) {
            auto blk_owner = pool.get();
            uint64_t irregular = 0, direct_emitted = 0;
            for (;;) {
                ReadBlock& blk = *blk_owner;
                auto r = use_gz ? gzpar.next(blk, &irregular) : par.next(blk, &irregular);
                if (r == mtsv_ingest::ParallelFastx::END) break;
                if (r == mtsv_ingest::ParallelFastx::CORRUPT) {  // records of a member whose CRC-32 then failed have been handed out
                    rd.fail("corrupt gzip data (CRC mismatch)");
                    return false;
                }
                if (r == mtsv_ingest::ParallelFastx::IRREGULAR) {
                    logmsg("DEBUG", std::string(use_gz ? "gzip input" : "input") + " is not plain 4-line FASTQ / FASTA (or not decodable in parallel) at byte " + std::to_string(irregular) + "; continuing with the serial reader");
                    if (gzseek(rd.in.f, (z_off_t)irregular, SEEK_SET) < 0) {
                        rd.fail("read error");
                        return false;
                    }
                    serial_from_start = true;  // the serial loop below continues from here
                    break;
                }
                uint64_t from = 0;
                if (skipped < read_offset) {
                    from = std::min<uint64_t>(read_offset - skipped, blk.n());
                    skipped += from;
                }
                // a block of about a batch, nothing pending: the block is the batch
                // (smaller blocks are collected into a batch -- except the first ones of the file, which the parser cuts
                //  short on purpose so that the GPU has work early)
                if (from == 0 && w->n() == 0 && blk.n() && blk.n() <= batch_reads + batch_reads / 2 && blk.bases.size() < (1ull << 30) &&
                    (2 * blk.n() >= batch_reads || direct_emitted < 4)) {
                    direct_emitted++;
                    if (!emit(std::move(blk_owner))) return true;
                    blk_owner = pool.get();
                    continue;
                }
                // cut the block at batch boundaries
                while (from < blk.n()) {
                    const uint64_t room = batch_reads > w->n() ? batch_reads - w->n() : 0;
                    const uint64_t take = std::min<uint64_t>(room, blk.n() - from);
                    if (take == blk.n() - from) {
                        w->append(blk, from);
                        from = blk.n();
                    } else {
                        ReadBlock part;  // records [from, from + take)
                        part.bases.assign(blk.bases.begin() + (ptrdiff_t)blk.off[from], blk.bases.begin() + (ptrdiff_t)blk.off[from + take]);
                        part.ids.assign(blk.ids, blk.id_off[from], blk.id_off[from + take] - blk.id_off[from]);
                        for (uint64_t r = from + 1; r <= from + take; r++) {
                            part.off.push_back(blk.off[r] - blk.off[from]);
                            part.id_off.push_back(blk.id_off[r] - blk.id_off[from]);
                        }
                        w->append(part);
                        from += take;
                    }
                    if (full()) {
                        if (!emit(std::move(w))) return true;
                        w = pool.get();
                    }
                }
            }
            if (!serial_from_start) {
                if (w->n() && !emit(std::move(w))) return true;
                if (all_emitted) all_emitted();  // before the teardown: unmapping a 10 GB input took 45 ms
                par.close();
                gzpar.close();
                return true;
            }
            par.close();
            gzpar.close();
        }
        Record r;
        while (rd.next(r)) {
            if (skipped < read_offset) {
                skipped++;
                continue;
            }
            w->bases.insert(w->bases.end(), (const uint8_t*)r.seq.data(), (const uint8_t*)r.seq.data() + r.seq.size());
            w->off.push_back(w->bases.size());
            w->ids += r.id;
            w->ids.push_back('\0');
            w->id_off.push_back(w->ids.size());
            if (full()) {
                if (!emit(std::move(w))) return true;
                w = pool.get();
            }
        }
        if (rd.error) return false;
        if (w->n()) emit(std::move(w));
        if (all_emitted) all_emitted();
        return true;
    };

    if (a.parse_only) {  // ingest self-check: counts and FNV-1a checksums of everything the binner would see
        uint64_t n = 0, nb = 0, hb = 1469598103934665603ull, hi = 1469598103934665603ull, hl = 1469598103934665603ull;
        auto fnv = [](uint64_t& h, const uint8_t* p, uint64_t len) {
            for (uint64_t i = 0; i < len; i++) h = (h ^ p[i]) * 1099511628211ull;
        };
        bool ok = produce(a.batch_reads, [&](std::unique_ptr<ReadBlock> w) {
            n += w->n();
            nb += w->bases.size();
            if (getenv("MTSV_PARSE_NOHASH")) {
                pool.put(std::move(w));
                return true;
            }
            fnv(hb, w->bases.data(), w->bases.size());
            fnv(hi, (const uint8_t*)w->ids.data(), w->ids.size());
            for (uint64_t r = 0; r < w->n(); r++) {
                uint64_t len = w->off[r + 1] - w->off[r];
                fnv(hl, (const uint8_t*)&len, 8);
            }
            return true;
        });
        if (!ok) {
            logmsg("ERROR", "Unable to read from input file: " + rd.err_msg);
            return 12;
        }
        printf("records=%llu bases=%llu bases_fnv=%016llx ids_fnv=%016llx lens_fnv=%016llx\n", (unsigned long long)n,
               (unsigned long long)nb, (unsigned long long)hb, (unsigned long long)hi, (unsigned long long)hl);
        return 0;
    }

    // positional writes from several threads: the result file is written at page-cache speed per thread
    const int out_fd = ::open(a.results.c_str(), O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0644);
    off_t out_pos = out_fd >= 0 ? lseek(out_fd, 0, SEEK_END) : 0;
    if (out_fd < 0) {
        logmsg("ERROR", "Error running query: cannot open results file " + a.results);
        return 2;
    }
    logmsg("INFO", "Deserializing candidate filter ...");
    std::vector<std::string> index_paths;
    for (size_t at = 0; at <= a.index.size();) {
        size_t c = a.index.find(',', at);
        if (c == std::string::npos) c = a.index.size();
        if (c > at) index_paths.push_back(a.index.substr(at, c - at));
        at = c + 1;
    }
    if (index_paths.empty()) usage_error("The following required arguments were not provided: --index <INDEX>");
    const bool chunked = index_paths.size() > 1;  // Mode B
    // One library call takes every batch that is waiting, up to kGroupReads reads (mtsv_batch_run_host_parts): the device
    // is several times faster on passes of a million reads than on a quarter of that (a pass costs ~2.5 ms before it does
    // any work), while the parser is fastest on blocks of ~80 MB.
    uint64_t kGroupReads = std::max<uint64_t>(a.batch_reads, 1ull << 20);
    if (const char* e = getenv("MTSV_CLI_GROUP_READS")) kGroupReads = std::max<uint64_t>(a.batch_reads, strtoull(e, nullptr, 10));
    const size_t group_max = (size_t)std::min<uint64_t>(32, std::max<uint64_t>(1, kGroupReads / std::max<uint64_t>(a.batch_reads, 1)));
    // Several workers per --devices entry, a workspace of ONE lane each: a call is copy in -> kernels -> hits out, and
    // what overlaps on the device are the calls of different workers (tools/call_stream.py: one worker with the
    // default three lanes 165 M reads/s on megaread calls, three workers of one lane 229 M).
    // (three workers on groups of 1 Mi reads: 0.210 s for 32 M reads twice over; two on 512 Ki: 0.225-0.245 -- since the
    //  parser stopped copying its blocks the workers are what a run waits for)
    size_t workers_per_device = 3;
    if (const char* e = getenv("MTSV_CLI_WORKERS")) workers_per_device = (size_t)std::max(1, std::min(8, atoi(e)));
    // (a small input is through before the extra workspaces have paid for themselves)
    uint64_t input_bytes = 0;
    {
        struct stat st;
        if (stat(input.c_str(), &st) == 0) input_bytes = (uint64_t)st.st_size;
    }
    const bool small_input = input_bytes < (256ull << 20) && !getenv("MTSV_CLI_WORKERS");
    if (small_input) workers_per_device = 1;
    const size_t n_workers = chunked ? 2 : a.devices.size() * workers_per_device;
    // the length of the input's first read (plain text; 150 otherwise): the workspaces are warmed with reads like it
    uint32_t warm_len = 150;
    if (FILE* hf = fopen(input.c_str(), "rb")) {
        std::vector<char> head(64 << 10);
        const size_t got = fread(head.data(), 1, head.size(), hf);
        fclose(hf);
        if (got >= 2 && !((uint8_t)head[0] == 0x1f && (uint8_t)head[1] == 0x8b)) {
            const char* nl = (const char*)memchr(head.data(), '\n', got);
            if (nl) {
                size_t len = 0;
                for (const char* q = nl + 1; q < head.data() + got && *q != '>' && *q != '+'; q++) len += *q != '\n' && *q != '\r';
                if (len >= 32) warm_len = (uint32_t)std::min<size_t>(len, 1000);
            }
        }
    }
    // (batches too large for one parser block are put together from several blocks by appending: those stay in ordinary
    //  memory -- growing a page-locked buffer means allocating another one -- and are staged by the library)
    std::thread stock_thread;
    struct Joiner {  // (an early return must not leave the thread running)
        std::thread& t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } stock_joiner{stock_thread};
    if (!getenv("MTSV_CLI_PAGEABLE") && (uint64_t)a.batch_reads * 320 <= (128ull << 20)) {
        mtsv_ingest::byte_alloc().alloc = [](size_t n) { return mtsv_host_alloc(n); };
        mtsv_ingest::byte_alloc().release = [](void* q) { mtsv_host_free(q); };
        // stock: the parser's window of blocks plus what sits in the queues and with the workers
        if (const char* e = getenv("MTSV_INGEST_BLOCK")) plain_block_bytes = strtoull(e, nullptr, 10);
        const uint64_t est = plain_block_bytes ? plain_block_bytes : (uint64_t)a.batch_reads * 320;
        const uint64_t per_call = std::min<uint64_t>(8, std::max<uint64_t>(1, std::max<uint64_t>(a.batch_reads, 1ull << 20) / std::max<uint64_t>(a.batch_reads, 1)));
        // (on a thread of its own: ~2 GB of page-locked memory take 0.1 s to create, the index is loaded meanwhile)
        const size_t n_stock = small_input ? 8 : std::min<size_t>(2 * host_threads + 2 + (size_t)(2 * per_call + 1) * (n_workers + 1), 128);
        const uint64_t stock_bytes = std::min<uint64_t>(est / 2 + (1 << 20), 512ull << 20);
        stock_thread = std::thread([&pool, n_stock, stock_bytes] { pool.stock(n_stock, stock_bytes); });
    }
    // The library would pack the bases to 4-bit codes on the host before they cross PCIe (host_pack.hpp: a dozen threads);
    // here the parser's threads need the CPUs and the link is not what bounds a run: the blocks go as they are
    // (MTSV_CLI_PACKED=1: packed).
    if (!getenv("MTSV_CLI_PACKED")) setenv("MTSV_H2D_PLAIN", "1", 0);
    // the two acceptance predicates are evaluated edit distance first (identical hits, about twice the device
    // rate for reads up to 253 bases); MTSV_VERIFY=reference keeps the reference's order
    if (!getenv("MTSV_VERIFY")) mtsv_set_default_verify_mode(MTSV_VERIFY_EDIT_FIRST);
    std::vector<mtsv_index*> idx(index_paths.size(), nullptr);
    std::vector<int> chunk_dev(index_paths.size(), 0);
    for (size_t c = 0; c < index_paths.size(); c++) {
        chunk_dev[c] = a.devices[c % a.devices.size()];
        if (mtsv_index_load(index_paths[c].c_str(), &idx[c]) != MTSV_OK) {
            logmsg("ERROR", std::string("Error running query: ") + mtsv_last_error());
            return 2;
        }
        // one index: resident on every listed device; chunks: chunk c on its device
        for (size_t d = 0; d < (chunked ? 1 : a.devices.size()); d++)
            if (mtsv_index_to_device(idx[c], chunked ? chunk_dev[c] : a.devices[d], MTSV_DEV_DEFAULT) != MTSV_OK) {
                logmsg("ERROR", std::string("Error running query: ") + mtsv_last_error());
                return 2;
            }
    }
    setup_mark("index loaded and resident");
    // the workers' workspaces (one index): part of the device set-up, like making the index resident -- created, sized for
    // the calls to come and run once on reads sampled from the index (mtsv_batch_reserve_host)
    std::vector<mtsv_batch*> ws_ready(chunked ? 0 : n_workers, nullptr);
    {
        std::vector<int> ws_rc(ws_ready.size(), MTSV_OK);
        std::vector<std::string> ws_msg(ws_ready.size());
        auto make_ws = [&](size_t wk) {
            const uint64_t call_reads = kGroupReads + a.batch_reads + a.batch_reads / 2;
            const int dev = a.devices[wk % a.devices.size()];
            int rc = workers_per_device > 1 ? mtsv_batch_create_lanes(idx[0], dev, call_reads, 1 << 22, 0, 1, &ws_ready[wk])
                                            : mtsv_batch_create(idx[0], dev, mtsv_bin_batch_workspace_reads(call_reads), 1 << 22, 0, &ws_ready[wk]);
            if (rc == MTSV_OK && !small_input && !getenv("MTSV_CLI_COLD"))
                rc = mtsv_batch_reserve_host(ws_ready[wk], call_reads, call_reads * (uint64_t)(warm_len + warm_len / 8), warm_len);
            ws_rc[wk] = rc;
            if (rc != MTSV_OK) ws_msg[wk] = mtsv_last_error();  // (thread-local)
        };
        std::vector<std::thread> th;
        for (size_t wk = 1; wk < ws_ready.size(); wk++) th.emplace_back(make_ws, wk);
        if (!ws_ready.empty()) make_ws(0);
        for (auto& t : th) t.join();
        for (size_t wk = 0; wk < ws_ready.size(); wk++)
            if (ws_rc[wk] != MTSV_OK) {
                logmsg("ERROR", "Error running query: " + ws_msg[wk]);
                return 2;
            }
    }
    setup_mark("workspaces ready");
    // parsed blocks land in page-locked memory from here on: the GPU copies them from where the parser put them
    if (stock_thread.joinable()) stock_thread.join();
    setup_mark("stock of page-locked blocks ready");
    logmsg("INFO", "Beginning queries.");
    struct timespec w0;
    clock_gettime(CLOCK_MONOTONIC, &w0);
    // MTSV_CLI_TIMING=1: where the stages of the command line spend their time (seconds, summed per stage)
    const bool cli_timing = getenv("MTSV_CLI_TIMING") != nullptr;
    auto now = [] {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return t.tv_sec + t.tv_nsec * 1e-9;
    };
    std::atomic<uint64_t> t_ingest_wait{0}, t_push_wait{0}, t_gpu{0}, t_gpu_wait{0}, t_fmt{0}, t_done_wait{0}, t_write_wait{0};
    auto acc = [](std::atomic<uint64_t>& a, double s) { a.fetch_add((uint64_t)(s * 1e6)); };
    const double t_begin = now();
    std::mutex marks_mu;
    std::vector<std::pair<std::string, double>> marks;  // (what, seconds since the queries began), printed with the timing
    auto mark = [&](const std::string& what) {
        if (!cli_timing) return;
        std::lock_guard<std::mutex> lk(marks_mu);
        marks.emplace_back(what, now() - t_begin);
    };

    // Overlapped stages (the reference overlaps producer / workers / joiner the same way,
    // vendor/cue/src/lib.rs:45-105): the producer parses FASTX into numbered batches, GPU workers (one per
    // --devices entry; two dispatchers over all chunks in chunk mode) take batches as they come, a writer
    // thread formats and writes the result lines in input order.
    struct Work {
        std::unique_ptr<ReadBlock> rb;
        uint64_t seq = 0;
        mtsv_hit* hits = nullptr;  // this batch's hits: a slice of the array hits_owner holds (read numbers: the call's, read_first + the batch's)
        uint64_t n_hits = 0;
        uint64_t read_first = 0;  // the batch's first read in the numbering of its call (the formatter subtracts it)
        std::shared_ptr<void> hits_owner;  // the result array of the library call the batch was part of
    };
    struct Queue {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::unique_ptr<Work>> q;
        bool closed = false;
        size_t cap = 2;
        size_t n_takers = 1;  // threads that call pop_group
        void push(std::unique_ptr<Work> w) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return q.size() < cap || closed; });
            q.push_back(std::move(w));
            cv.notify_all();
        }
        // the next batches in order, up to max_reads reads / max_n batches: a full group, or -- when the input has ended or
        // `idle` says that the device has nothing else to do -- whatever is waiting
        std::vector<std::unique_ptr<Work>> pop_group(uint64_t max_reads, size_t max_n, const std::function<bool()>& idle) {
            std::unique_lock<std::mutex> lk(mu);
            auto waiting = [&] {
                uint64_t r = 0;
                for (auto& w : q) r += w->rb->n();
                return r;
            };
            for (;;) {
                if (!q.empty() && (closed || q.size() >= max_n || waiting() >= max_reads || idle())) break;
                if (q.empty() && closed) return {};
                cv.wait_for(lk, std::chrono::microseconds(200));  // (idle() changes without a notification)
            }
            // (the input has ended: what is left is shared out, so that the workers finish together)
            if (closed && n_takers > 1) max_n = std::min(max_n, (q.size() + n_takers - 1) / n_takers);
            std::vector<std::unique_ptr<Work>> g;
            uint64_t r = 0;
            while (!q.empty() && g.size() < max_n && r < max_reads) {
                r += q.front()->rb->n();
                g.push_back(std::move(q.front()));
                q.pop_front();
            }
            cv.notify_all();
            return g;
        }
        std::unique_ptr<Work> try_pop() {  // nullptr when nothing is waiting
            std::lock_guard<std::mutex> lk(mu);
            if (q.empty()) return nullptr;
            auto w = std::move(q.front());
            q.pop_front();
            cv.notify_all();
            return w;
        }
        std::unique_ptr<Work> pop() {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !q.empty() || closed; });
            if (q.empty()) return nullptr;
            auto w = std::move(q.front());
            q.pop_front();
            cv.notify_all();
            return w;
        }
        void close() {
            std::lock_guard<std::mutex> lk(mu);
            closed = true;
            cv.notify_all();
        }
    };
    Queue parsed, done;
    parsed.cap = (n_workers + 1) * group_max + 1;
    parsed.n_takers = n_workers;
    done.cap = n_workers * group_max + 1;
    std::atomic<int> calls_in_flight{0};
    std::mutex err_mu;
    int exit_code = 0;
    auto set_code = [&](int c) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (!exit_code) exit_code = c;
    };
    auto failed = [&] {
        std::lock_guard<std::mutex> lk(err_mu);
        return exit_code != 0;
    };

    uint64_t n_batches = 0;
    all_emitted = [&] {
        mark("last block parsed");
        parsed.close();
    };
    std::thread reader([&] {
        double t_last = now();
        bool ok = produce(a.batch_reads, [&](std::unique_ptr<ReadBlock> rb) {
            if (failed()) return false;
            auto w = std::make_unique<Work>();
            w->rb = std::move(rb);
            w->seq = n_batches++;
            const double t_a = now();
            if (w->seq == 0) mark("first block parsed");
            acc(t_ingest_wait, t_a - t_last);  // producing this batch (mostly: waiting for the parser threads)
            parsed.push(std::move(w));
            t_last = now();
            acc(t_push_wait, t_last - t_a);    // waiting for room in the queue to the GPU workers
            return true;
        });
        if (!ok) {
            logmsg("ERROR", "Unable to read from input file: " + rd.err_msg);
            set_code(12);  // binner.rs:81-84
        }
        parsed.close();
        mark("reader done");
    });

    // helper threads of the result side: the formatting of a batch's hits runs on them in parallel, and the
    // finished text of batch k is written (positional writes at offsets fixed in batch order) while batch k+1
    // is being formatted
    struct Helpers {
        std::mutex mu;
        std::condition_variable cv, idle;
        std::deque<std::function<void()>> q;
        std::vector<std::thread> th;
        size_t running = 0;
        bool stop = false;
        explicit Helpers(unsigned n) {
            for (unsigned i = 0; i < n; i++)
                th.emplace_back([this] {
                    for (;;) {
                        std::function<void()> f;
                        {
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return stop || !q.empty(); });
                            if (q.empty()) return;
                            f = std::move(q.front());
                            q.pop_front();
                            running++;
                        }
                        f();
                        {
                            std::lock_guard<std::mutex> lk(mu);
                            running--;
                        }
                        idle.notify_all();
                    }
                });
        }
        void submit(std::function<void()> f) {
            {
                std::lock_guard<std::mutex> lk(mu);
                q.push_back(std::move(f));
            }
            cv.notify_one();
        }
        void wait_all() {
            std::unique_lock<std::mutex> lk(mu);
            idle.wait(lk, [&] { return q.empty() && running == 0; });
        }
        void wait_below(size_t n) {  // until fewer than n jobs are queued or running
            std::unique_lock<std::mutex> lk(mu);
            idle.wait(lk, [&] { return q.size() + running < n; });
        }
        ~Helpers() {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            for (auto& t : th) t.join();
        }
    };
    Helpers fmt_pool(host_threads), write_pool(std::max(2u, host_threads / 2));
    std::atomic<bool> write_failed{false};
    std::thread writer([&] {
        uint64_t total = 0, next_seq = 0;
        std::vector<std::unique_ptr<Work>> held;  // batches that finished ahead of their turn
        for (;;) {
            std::unique_ptr<Work> w;
            for (auto& h : held)
                if (h && h->seq == next_seq) {
                    w = std::move(h);
                    h = std::move(held.back());
                    held.pop_back();
                    break;
                }
            if (!w) {
                const double t_a = now();
                w = done.pop();
                acc(t_done_wait, now() - t_a);
                if (!w) break;
                if (w->seq != next_seq) {
                    held.push_back(std::move(w));
                    continue;
                }
            }
            next_seq++;
            const double t_f0 = now();
            // write_assignments over slices of the batch's hits (cut between reads), one thread each
            const uint64_t n_reads = w->rb->n();
            const unsigned parts = w->n_hits >= (1u << 16) ? host_threads : 1;
            std::vector<uint64_t> cut(parts + 1, w->n_hits);
            cut[0] = 0;
            for (unsigned k = 1; k < parts; k++) {
                uint64_t c = std::max(cut[k - 1], w->n_hits * k / parts);
                while (c > cut[k - 1] && c < w->n_hits && w->hits[c].read == w->hits[c - 1].read) c++;
                cut[k] = c;
            }
            std::vector<char*> text(parts, nullptr);
            std::vector<uint64_t> len(parts, 0);
            std::vector<int> rc(parts, MTSV_OK);
            std::vector<std::string> msg(parts);
            auto fmt = [&](unsigned k) {
                if (w->read_first)
                    for (uint64_t i = cut[k]; i < cut[k + 1]; i++) w->hits[i].read -= w->read_first;
                rc[k] = mtsv_format_results(w->hits + cut[k], cut[k + 1] - cut[k], w->rb->ids.data(), w->rb->id_off.data(), n_reads,
                                            long_fmt, &text[k], &len[k]);
                if (rc[k] != MTSV_OK) msg[k] = mtsv_last_error();  // thread-local
            };
            if (parts > 1) {
                for (unsigned k = 1; k < parts; k++) fmt_pool.submit([&fmt, k] { fmt(k); });
                fmt(0);
                fmt_pool.wait_all();
            } else {
                fmt(0);
            }
            w->hits_owner.reset();  // (the array goes back to the library's pool with the last batch of its call)
            acc(t_fmt, now() - t_f0);
            bool ok = true;
            for (unsigned k = 0; k < parts; k++)
                if (rc[k] != MTSV_OK) {
                    if (ok) logmsg("ERROR", "Error running query: " + msg[k]);
                    set_code(2);
                    ok = false;
                }
            if (ok && !failed() && !write_failed.load()) {
                for (unsigned k = 0; k < parts; k++) {
                    const off_t at = out_pos;
                    out_pos += (off_t)len[k];
                    char* tx = text[k];
                    const uint64_t ln = len[k];
                    text[k] = nullptr;  // the write job owns it now
                    write_pool.submit([tx, ln, at, out_fd, &write_failed, &set_code] {
                        uint64_t done = 0;
                        while (done < ln) {
                            ssize_t r = pwrite(out_fd, tx + done, ln - done, at + (off_t)done);
                            if (r <= 0) {
                                write_failed.store(true);
                                set_code(11);  // binner.rs:136-139: the producer and the workers stop at once
                                break;
                            }
                            done += (uint64_t)r;
                        }
                        mtsv_free(tx);
                    });
                }
            }
            for (unsigned k = 0; k < parts; k++) mtsv_free(text[k]);
            // at most about two batches of text wait for the disk: the file stays a prefix of the results up to the
            // writes in flight (resume reads its last line), and formatted text does not pile up behind a slow disk
            {
                const double t_a = now();
                write_pool.wait_below(2 * (size_t)host_threads + 1);
                acc(t_write_wait, now() - t_a);
            }
            pool.put(std::move(w->rb));
            if (!ok) continue;
            total += n_reads;
            logmsg("DEBUG", "taxonomic binning: " + std::to_string(total) + " reads done");
        }
        write_pool.wait_all();
        if (write_failed.load()) {
            logmsg("ERROR", "Error writing to result file");
            set_code(11);  // binner.rs:136-139
        }
    });

    auto gpu_worker = [&](size_t wk) {
        mtsv_batch* ws = chunked ? nullptr : ws_ready[wk];  // one index: this worker's own workspace on its device
        for (;;) {
            const double t_p = now();
            std::vector<std::unique_ptr<Work>> group;
            if (chunked) {
                group.push_back(parsed.pop());
                if (!group[0]) break;
            } else {
                // a full group, or what there is when no call is running on any device (the start of the input, a slow parser)
                group = parsed.pop_group(kGroupReads, group_max, [&] { return calls_in_flight.load() == 0; });
                if (group.empty()) break;
            }
            acc(t_gpu_wait, now() - t_p);
            if (failed()) continue;  // drain
            uint64_t group_reads = 0;
            for (auto& w : group) group_reads += w->rb->n();
            calls_in_flight++;
            struct InFlight {
                std::atomic<int>& c;
                ~InFlight() { c--; }
            } in_flight{calls_in_flight};
            const double t_g = now();
            int rc;
            mtsv_hit* hits = nullptr;
            uint64_t n_hits = 0;
            if (chunked) {
                auto& w = group[0];
                rc = mtsv_bin_batch_chunks(idx.data(), chunk_dev.data(), (int)idx.size(), w->rb->bases.data(), w->rb->off.data(), w->rb->n(), &p,
                                           &hits, &n_hits);
            } else {
                std::vector<const uint8_t*> pb;
                std::vector<const uint64_t*> po;
                std::vector<uint64_t> pn;
                for (auto& w : group) {
                    pb.push_back(w->rb->bases.data());
                    po.push_back(w->rb->off.data());
                    pn.push_back(w->rb->n());
                }
                rc = mtsv_batch_run_host_parts(ws, (int)group.size(), pb.data(), po.data(), pn.data(), &p);
                if (rc == MTSV_OK) rc = mtsv_batch_download(ws, &hits, &n_hits);
            }
            if (rc != MTSV_OK) {
                logmsg("ERROR", std::string("Error running query: ") + mtsv_last_error());
                set_code(2);
                continue;
            }
            acc(t_gpu, now() - t_g);
            if (cli_timing) {
                char what[96];
                snprintf(what, sizeof what, "worker %zu: call on %llu reads in %zu blocks took %.2f ms, ended", wk, (unsigned long long)group_reads,
                         group.size(), (now() - t_g) * 1e3);
                mark(what);
            }
            // every batch gets its slice of the hits, read numbers relative to the batch
            // (the boundaries by bisection, the renumbering on the formatting threads: a pass over a million hits between two
            //  calls was 1.7 ms of every 8 the worker spent per megaread)
            std::shared_ptr<void> owner(hits, [](void* q) { mtsv_hits_free((mtsv_hit*)q); });
            uint64_t first = 0, at = 0;
            for (auto& w : group) {
                const uint64_t nr = w->rb->n();
                const uint64_t end = (uint64_t)(std::partition_point(hits + at, hits + n_hits, [&](const mtsv_hit& h) { return h.read < first + nr; }) - hits);
                w->hits = hits + at;
                w->n_hits = end - at;
                w->read_first = first;
                w->hits_owner = owner;
                at = end;
                first += nr;
            }
            for (auto& w : group) done.push(std::move(w));
        }
        mark("worker out of batches");
    };
    {
        std::vector<std::thread> workers;
        for (size_t wk = 1; wk < n_workers; wk++) workers.emplace_back(gpu_worker, wk);
        gpu_worker(0);
        for (auto& t : workers) t.join();
    }
    done.close();
    writer.join();
    mark("writer done");
    // (the reader has handed on its last batch, or failed and said so, before the workers and the writer can end; what it
    //  may still be doing is closing its input -- unmapping 10 GB takes 45 ms -- and that is not part of the queries)
    struct ReaderJoin {
        std::thread& t;
        ~ReaderJoin() { t.join(); }
    } reader_join{reader};
    if (exit_code) return exit_code;
    if (::close(out_fd) != 0) {
        logmsg("ERROR", "Error writing to result file");
        return 11;
    }
    mark("results file closed");
    struct timespec w1;
    clock_gettime(CLOCK_MONOTONIC, &w1);
    char msg[160];
    snprintf(msg, sizeof msg, "All worker and result consumer threads terminated. Took %.3f seconds.",
             (w1.tv_sec - w0.tv_sec) + (w1.tv_nsec - w0.tv_nsec) * 1e-9);
    logmsg("INFO", msg);
    if (cli_timing)
        fprintf(stderr, "[cli timing] batches %llu; reader: producing %.3f s, queue full %.3f s; gpu workers: in the library %.3f s, waiting for batches %.3f s; "
                        "writer: formatting %.3f s, waiting for hits %.3f s, waiting for the disk %.3f s\n",
                (unsigned long long)n_batches, t_ingest_wait.load() * 1e-6, t_push_wait.load() * 1e-6, t_gpu.load() * 1e-6, t_gpu_wait.load() * 1e-6,
                t_fmt.load() * 1e-6, t_done_wait.load() * 1e-6, t_write_wait.load() * 1e-6);
    if (cli_timing && getenv("MTSV_CLI_MARKS"))
        for (auto& m : marks) fprintf(stderr, "[cli timing] %9.3f ms  %s\n", m.second * 1e3, m.first.c_str());
    setup_mark("queries done");
    // The results are on their way to the disk (close() has returned) and the log line is out: leave.  Handing 2 GB of
    // page-locked blocks, the workspaces and the resident index back piece by piece takes 0.3 s that the kernel's own
    // teardown of the process does not need (MTSV_CLI_CLEAN_EXIT=1: free everything, for leak checkers).
    if (!getenv("MTSV_CLI_CLEAN_EXIT")) {
        fflush(nullptr);  // (the reader thread may still be unmapping its input: it ends with the process)
        _exit(0);
    }
    for (auto* ws : ws_ready) mtsv_batch_free(ws);  // (30 ms per workspace: after the queries' clock, like the index)
    setup_mark("workspaces freed");
    for (auto* ix : idx) mtsv_index_free(ix);
    setup_mark("index freed");
    return 0;
}

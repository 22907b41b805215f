// Host side of run_host's transfer format: the reads' bases as 4-bit codes, two per byte (host only: x86 intrinsics).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdio>
#include <cstring>
#if defined(__linux__)
#include <sched.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "host_pack.hpp"

namespace mtsv {
int pack_threads();
namespace {
// binner.rs:88-100 on the host: A/a C/c G/g T/t -> 0..3, anything else -> N (4) -- what k_normalise does on the device
// for the resident path.  Two codes per byte, base i of a segment in nibble (i & 1) of byte i / 2: half the bytes over
// PCIe, which is what bounds the host path.
inline uint8_t host_code(uint8_t ch) {
    const uint8_t uc = ch & 0xDFu, x = (uc >> 1) & 3u;
    return (uc == 'A' || uc == 'C' || uc == 'G' || uc == 'T') ? (uint8_t)(x ^ (x >> 1)) : (uint8_t)4;
}
void pack_pairs_scalar(uint8_t* dst, const uint8_t* src, uint64_t n_pairs) {
    for (uint64_t i = 0; i < n_pairs; i++) dst[i] = (uint8_t)(host_code(src[2 * i]) | (host_code(src[2 * i + 1]) << 4));
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) void pack_pairs_avx2(uint8_t* dst, const uint8_t* src, uint64_t n_pairs) {
    const __m256i up = _mm256_set1_epi8((char)0xDF), three = _mm256_set1_epi8(3), four = _mm256_set1_epi8(4);
    const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
    const __m256i mul = _mm256_set1_epi16(0x1001);  // low byte x 1 + high byte x 16
    uint64_t i = 0;
    const bool stream = (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && n_pairs >= (1u << 14);
    for (; i + 16 <= n_pairs; i += 16) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 2 * i));
        const __m256i uc = _mm256_and_si256(v, up);
        const __m256i x = _mm256_and_si256(_mm256_srli_epi16(uc, 1), three);           // (bits that cross bytes are masked off)
        const __m256i c = _mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), three));
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(uc, cA), _mm256_cmpeq_epi8(uc, cC)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(uc, cG), _mm256_cmpeq_epi8(uc, cT)));
        const __m256i code = _mm256_blendv_epi8(four, c, ok);
        const __m256i w = _mm256_maddubs_epi16(code, mul);                             // 16 words: even code | odd code << 4
        const __m256i b = _mm256_permute4x64_epi64(_mm256_packus_epi16(w, w), 0xD8);   // the 16 low bytes
        // (the staging buffer is read next by the copy engine, not by a CPU: past the caches when it is aligned)
        if (stream) _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), _mm256_castsi256_si128(b));
        else _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), _mm256_castsi256_si128(b));
    }
    if (stream) _mm_sfence();
    pack_pairs_scalar(dst + i, src + 2 * i, n_pairs - i);
}
#endif
void pack_pairs(uint8_t* dst, const uint8_t* src, uint64_t n_pairs) {
#if defined(__x86_64__)
    // (a 512-bit variant -- one vpermi2b table look-up per 64 bases -- packs a third faster on one core and made the whole
    //  step erratic on the GPU box's EPYC 9575F: 31.9 to 38.5 ms where this one holds 31.8-32.0)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return pack_pairs_avx2(dst, src, n_pairs);
#endif
    pack_pairs_scalar(dst, src, n_pairs);
}
// bases src[0, n) that lie at segment offsets [a, a + n) into packed bytes dst[0 ...) = bytes [a / 2, (a + n + 1) / 2) of the
// segment; prev_code: the code of the base at offset a - 1 (when a is odd, it shares the first byte).  Returns the code of
// the last base.
// a few threads that live as long as the library: a job is cut into pieces, the caller takes pieces too.  Everything a
// thread needs of a job lies in the job's own record, and run() does not return while a thread still holds it (jobs of
// different piece sizes follow each other: a thread that came late to one must not mix its numbers with the next one's).
class PackPool {
   public:
    explicit PackPool(int n) {
        for (int k = 0; k < n; k++) th_.emplace_back([this] { work(); });
    }
    ~PackPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // fn(p0, p1) over [0, n) in pieces; returns when all of them are done (one job at a time)
    void run(uint64_t n, uint64_t piece, const std::function<void(uint64_t, uint64_t)>& fn) {
        std::lock_guard<std::mutex> one(job_mu_);
        Job j;
        j.fn = &fn;
        j.n = n;
        j.piece = piece;
        j.left = (n + piece - 1) / piece;
        j.holders = 1;  // the caller
        {
            std::lock_guard<std::mutex> lk(mu_);
            cur_ = &j;
            gen_++;
        }
        cv_.notify_all();
        take(j);
        std::unique_lock<std::mutex> lk(mu_);
        j.holders--;
        done_.wait(lk, [&] { return j.left == 0 && j.holders == 0; });
        cur_ = nullptr;
    }

   private:
    struct Job {
        const std::function<void(uint64_t, uint64_t)>* fn = nullptr;
        uint64_t n = 0, piece = 1;
        std::atomic<uint64_t> next{0};
        uint64_t left = 0;  // pieces not finished (mu_)
        int holders = 0;    // threads that may still touch the record (mu_)
    };
    void take(Job& j) {
        for (;;) {
            const uint64_t p0 = j.next.fetch_add(j.piece);
            if (p0 >= j.n) return;
            (*j.fn)(p0, std::min(j.n, p0 + j.piece));
            std::lock_guard<std::mutex> lk(mu_);
            j.left--;
        }
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (cur_ && gen_ != seen); });
                if (stop_) return;
                seen = gen_;
                j = cur_;
                j->holders++;
            }
            take(*j);
            {
                std::lock_guard<std::mutex> lk(mu_);
                j->holders--;
                if (j->left == 0 && j->holders == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_;
    bool stop_ = false;
    uint64_t gen_ = 0;
    Job* cur_ = nullptr;
};
// CPUs this process may keep busy: its affinity mask, capped by the cgroup's CPU quota (v2 cpu.max, v1 cfs_quota_us)
int usable_cpus() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    auto quota = [](const char* path, bool v2) -> double {
        FILE* f = fopen(path, "r");
        if (!f) return 0.0;
        char a[64] = {0}, b[64] = {0};
        double q = 0.0;
        if (v2) {
            if (fscanf(f, "%63s %63s", a, b) == 2 && strcmp(a, "max") != 0 && atof(b) > 0) q = atof(a) / atof(b);
        } else if (fscanf(f, "%63s", a) == 1 && atof(a) > 0) {
            q = atof(a) / 100000.0;  // (the default period)
        }
        fclose(f);
        return q;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", true);
    if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
    if (q > 0) n = std::max(1, std::min(n, (int)(q + 0.5)));
#endif
    return n;
}
PackPool& pool() {
    static PackPool p(pack_threads() - 1);  // (the caller packs too)
    return p;
}
}  // namespace

// Threads that pack a chunk, the caller included: MTSV_PACK_THREADS, else what the process may use less the four threads
// a host batch keeps busy besides (three lanes waiting on their streams, the feeder), ten at most (measured on a
// 16-CPU grant, medians of ten steps: 8 threads 36.1 ms per 10 M reads, 10 31.9, 12 31.8, 16 32.6, 24 35.3; the plain
// transfer 36.3).
int pack_threads() {
    static const int n = [] {
        if (const char* e = getenv("MTSV_PACK_THREADS")) return std::max(1, std::min(64, atoi(e)));
        return std::max(1, std::min(10, usable_cpus() - 4));
    }();
    return n;
}

uint8_t pack_chunk(uint8_t* dst, const uint8_t* src, uint64_t a, uint64_t n, uint8_t prev_code) {
    if (!n) return prev_code;
    uint64_t s = 0;  // bases consumed
    uint8_t* d = dst;
    if (a & 1) {
        *d++ = (uint8_t)(prev_code | (host_code(src[0]) << 4));
        s = 1;
    }
    const uint64_t n_pairs = (n - s) / 2;
    if (n_pairs < (1u << 18)) pack_pairs(d, src + s, n_pairs);  // (half a MiB of bases: not worth waking anybody)
    else {
        const uint8_t* sp = src + s;
        pool().run(n_pairs, 1u << 17, [d, sp](uint64_t p0, uint64_t p1) { pack_pairs(d + p0, sp + 2 * p0, p1 - p0); });  // pieces of 256 KiB of bases
    }
    if ((n - s) & 1) d[n_pairs] = host_code(src[n - 1]);  // the last base alone in its byte (the next chunk completes it)
    return host_code(src[n - 1]);
}


void pack_pool_for(uint64_t n, uint64_t piece, const std::function<void(uint64_t, uint64_t)>& fn) {
    if (n <= piece) fn(0, n);
    else pool().run(n, piece, fn);
}

}  // namespace mtsv

// k_verify.hip -- SW prefilter, edit distance, selection loop, hit gather (index.rs:384-428, ssw.c, align.rs)
// (one of the three kernel files of the hot path; the stage map is in kernels.hpp / DESIGN.md section 3)
#include "kernels_common.hpp"

namespace mtsv {

namespace {

// ---------------------------------------------------------------------------------------------
// K4: verification.  A wavefront is four 16-lane groups (one DPP row each); every group walks the
// ranked candidates of one strand in the reference's order (index.rs:384-428).  A candidate's
// window is swept once by an anti-diagonal pipeline over the group's lanes: lane g owns R
// consecutive read rows, gets the row above from lane g-1 through a DPP row shift, and the window
// symbols from an LDS ring staged once per candidate.  Each cell carries, in the two 16-bit halves
// of one register,
//   low  half: the Smith-Waterman local score   H = max(0, diag+s, up-1, left-1)   (ssw.c:123-328
//              with the matrix of ssw/src/lib.rs:11-16 and gap 1/1; equals the striped byte
//              kernel while the score stays below 254, i.e. for every read up to 253 bases)
//   high half: the negated semi-global edit distance -D, so that
//              -D = max(diag-delta, up-1, left-1) shares the same packed max/add instructions
//              (align.rs:28-85: first row 0, first column i, answer = min of the last row).
// Reads shorter than 16*R rows are padded at the top with rows that behave like the zero
// boundary of both recurrences, so the last read row is always row R-1 of lane 15.
// ---------------------------------------------------------------------------------------------
typedef short pk16 __attribute__((ext_vector_type(2)));

__device__ inline pk16 pk(int lo, int hi) {
    pk16 r;
    r.x = (short)lo;
    r.y = (short)hi;
    return r;
}
__device__ inline pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
__device__ inline int pk_bits(pk16 a) { return __builtin_bit_cast(int, a); }
__device__ inline pk16 pk_from_bits(int u) { return __builtin_bit_cast(pk16, u); }

constexpr uint32_t kGroup = 16;
constexpr uint32_t kRing = 1024;  // window bytes resident per group

// value of lane g-1 of the same 16-lane row (row_shr:1); lane 0 of a row gets 0
__device__ inline int row_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }

__device__ inline uint32_t group_ballot(bool p, uint32_t group) { return (uint32_t)(__ballot(p) >> (group * kGroup)) & 0xffffu; }

// 16 window symbols starting at text[pos] (any alignment) as four little-endian dwords; bytes past the text read as 7
// (the text allocation is padded with 7 to a 16-byte multiple plus 32: dev_index.hip).  Two aligned 16-byte loads and a
// select instead of five dword loads: a lane's window lies anywhere in a multi-GB text, so every load instruction of a
// wavefront touches 64 different lines, and it is the number of such instructions that the texture path pays for.
__device__ inline uint4 load16(const uint8_t* __restrict__ text, uint32_t n, uint32_t pos) {
    (void)n;
    const uint4* t128 = reinterpret_cast<const uint4*>(text);
    const uint32_t w = pos >> 4, dsel = (pos >> 2) & 3u, sh = pos & 3u;
    const uint4 a = t128[w], b = t128[w + 1];
    const uint32_t e0 = dsel == 0 ? a.x : dsel == 1 ? a.y : dsel == 2 ? a.z : a.w;
    const uint32_t e1 = dsel == 0 ? a.y : dsel == 1 ? a.z : dsel == 2 ? a.w : b.x;
    const uint32_t e2 = dsel == 0 ? a.z : dsel == 1 ? a.w : dsel == 2 ? b.x : b.y;
    const uint32_t e3 = dsel == 0 ? a.w : dsel == 1 ? b.x : dsel == 2 ? b.y : b.z;
    const uint32_t e4 = dsel == 0 ? b.x : dsel == 1 ? b.y : dsel == 2 ? b.z : b.w;
    uint4 r;
    r.x = __builtin_amdgcn_alignbyte(e1, e0, sh);
    r.y = __builtin_amdgcn_alignbyte(e2, e1, sh);
    r.z = __builtin_amdgcn_alignbyte(e3, e2, sh);
    r.w = __builtin_amdgcn_alignbyte(e4, e3, sh);
    return r;
}

enum : uint32_t { PH_FETCH = 0, PH_SWEEP = 1, PH_DONE = 2, PH_CHAIN = 3, PH_WAIT = 4 };

// One work item = the first candidate of a TaxId of a strand (index into cand[]).  When it fails, the
// group moves on to the next candidate of that TaxId in rank order (same read rows, new window), which
// is exactly when the reference's loop would verify it (index.rs:393).  Items are claimed four at a
// time from an atomic cursor, so chains of failing candidates do not unbalance the groups.
// WORD: reads of 254+ bases can push the byte kernel of ssw.c to its overflow value 255 (score >= 254,
// ssw.c:271,302), after which ssw_align reruns sw_sse2_word (ssw.c:789-792).  With gap open == gap
// extend that kernel's lazy-F loop (ssw.c:452-463) always leaves after its first cell, so a vertical
// gap crosses a stripe boundary (read rows that are multiples of segLen = ceil(L/8)) by one cell only
// and E never sees the lazy correction.  Row-wise that is: hb = max(0, diag+s, left_pre-1);
// h = max(hb, up_pre-1); pre = boundary row ? hb : h -- computed here next to the exact score.
// TILED: reads of any length (up to 32767 bases, the range of the packed 16-bit cells).  The GS*R rows the
// group's registers hold are one band of the matrix; the bands are swept top to bottom, each over all
// columns of the window, and the bottom row of a band (both recurrences and the word-kernel pair) travels
// to the next band through a per-group strip in HBM: lane GS-1 writes column j after computing it, lane 0
// of the next band reads it as its row above (it runs GS-1 columns ahead of the writer, so one strip
// serves both directions).  Padding rows sit above the first band.
template <int R, bool WORD, int GS, bool TILED = false>
__global__ __launch_bounds__(256, (WORD || R * GS > 160) ? 1 : (GS == 8 ? 3 : 5)) void k_evaluate(DevIndexView ix, EvalArgs a) {
    static_assert(!TILED || WORD, "the tiled kernel is the long-read path: it carries the word kernel");
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[256 / GS][kRing];
    const uint32_t lane = lane_id();
    const uint32_t gl = lane & (GS - 1);  // lane inside the group
    uint8_t* ring = ring_all[threadIdx.x / GS];
    uint2* strip = TILED ? a.strip + (uint64_t)(blockIdx.x * (256 / GS) + threadIdx.x / GS) * a.strip_len : nullptr;
    uint32_t band = 0, n_bands = 1;  // TILED: current band and bands of the strand's read
    uint2 bnd_next = make_uint2(0, 0);
    uint32_t b0_read = 0, strand_read = 0;
    const uint32_t n_work = *a.wl_count;
    const pk16 one = pk(1, 1), miss = pk(-1, -1);
    const uint32_t maxc = a.maxc;  // max_candidates as a rank bound, clamped on the host

    uint32_t phase = PH_FETCH;
    uint32_t loc = 0, loc_end = 0;  // this group's slice of the worklist, refilled 4 items at a time
    uint32_t g = 0, o = 0, ED = 0, thr = 0, L = 0;
    bool thr_wrapped = false, hopeless = false;
    uint32_t qc[R];
    pk16 clampRow[R];
    uint32_t t = 0, steps = 0, Wn = 0, wstart = 0, c_next = 7;
    uint32_t c_tax = 0, c_gi = 0, c_off = 0;
    pk16 h[R], up_prev = pk(0, 0), best = pk(0, -32768), last = pk(0, -32768);
    // word-kernel emulation: post-lazy H, pre-lazy H, boundary-row mask, running max
    int hw[WORD ? R : 1], hwp[WORD ? R : 1];
    int upw_prev = 0, bestw = 0;
    uint32_t bmask = 0;
    unsigned long long verified = 0, wbytes = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        qc[r] = 6;
        clampRow[r] = pk(0, 0);
        h[r] = pk(0, 0);
    }

    for (;;) {
        if (phase == PH_FETCH || phase == PH_CHAIN) {
            bool have = true;
            if (phase == PH_FETCH) {
                if (loc == loc_end) {  // dynamic scheduling: chains of failing candidates make items uneven
                    uint32_t base = 0;
                    if (gl == 0) base = atomicAdd(a.wl_cursor, 16u);  // see k_sw_pairs: claims of 4 are atomic-bound
                    base = (uint32_t)__shfl((int)base, (int)(lane & ~(uint32_t)(GS - 1)));
                    loc = min(base, n_work);
                    loc_end = min(base + 16, n_work);
                }
                if (loc < loc_end) {
                    g = a.worklist[loc++];
                    const uint32_t rs = a.cand[g].w;
                    const uint32_t r_ = a.r0 + (rs >> 1), strand = rs & 1;
                    o = a.strand_off[rs];
                    const uint32_t b0 = a.read_off[r_];
                    L = a.read_off[r_ + 1] - b0;
                    ED = (uint32_t)ceil((double)L * a.edit_rate);  // index.rs:281-282
                    thr_wrapped = 2ull * ED > (uint64_t)L;         // usize wrap of index.rs:406: nothing can pass
                    thr = L - 2 * ED;
                    const uint8_t* read = a.bases + b0;
                    if (TILED) {
                        n_bands = (L + GS * R - 1) / (GS * R);
                        b0_read = b0;
                        strand_read = strand;
                        // an N of the read never matches in the edit-distance recurrence: edits >= #N
                        int nn = 0;
                        for (uint32_t p = gl; p < L; p += GS) nn += strand_code(read, L, strand, p) == kCodeN;
#pragma unroll
                        for (int d = 1; d < GS; d <<= 1) nn += __shfl_xor(nn, d);
                        hopeless = thr_wrapped || (uint32_t)nn > ED;
                    } else {
                    const int pad = (int)(GS * R) - (int)L;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        int p = (int)(gl * R + r) - pad;  // read position of this row, < 0 for padding
                        qc[r] = p >= 0 ? strand_code(read, L, strand, (uint32_t)p) : 6u;
                        clampRow[r] = p >= 0 ? pk(0, -32768) : pk(0, 0);
                    }
                    if (WORD) {
                        const int seg8 = (int)((L + 7) / 8);  // qP_word, ssw.c:336
                        bmask = 0;
#pragma unroll
                        for (int r = 0; r < (WORD ? R : 1); r++) {
                            int p = (int)(gl * R + r) - pad;
                            if (p >= 0 && p % seg8 == 0) bmask |= 1u << r;
                        }
                    }
                    // an N of the read never matches in the edit-distance recurrence (index.rs:272-279):
                    // edits >= #N.  More N than the tolerance: no candidate of this strand can pass :410
                    int nn = 0;
#pragma unroll
                    for (int r = 0; r < R; r++) nn += qc[r] == kCodeN;
#pragma unroll
                    for (int d = 1; d < GS; d <<= 1) nn += __shfl_xor(nn, d);
                    hopeless = thr_wrapped || (uint32_t)nn > ED;
                    }
                } else {
                    phase = PH_DONE;
                    have = false;
                }
            }
            if (have && hopeless) {  // rejected without a sweep; the reference still ran its prefilter on it
                const uint4 c = a.cand[g];
                verified++;
                wbytes += c.y - c.x;
                if (gl == 0) a.cand_status[g] = 1;
                const uint32_t nxt = a.cand_next[g];
                if (nxt != 0xffffffffu && nxt < maxc) {
                    g = o + nxt;
                    phase = PH_CHAIN;
                } else {
                    phase = PH_FETCH;
                }
                have = false;
            }
            if (have) {  // (re)start a sweep: candidate g of the strand whose rows are already loaded
                const uint4 c = a.cand[g];
                const DevBin bin = ix.bins[c.z];
                wstart = c.x;
                Wn = c.y - c.x;
                c_tax = bin.tax_id;
                c_gi = bin.gi;
                c_off = c.x >= bin.start ? c.x - bin.start : 0;  // index.rs:416
                verified++;
                wbytes += Wn;
                band = 0;
                best = pk(0, -32768);
                if (WORD) bestw = 0;
                phase = PH_SWEEP;
                t = steps = 0;  // the band set-up below starts the sweep
            }
        }
        if (__all(phase == PH_DONE)) break;
        if (phase == PH_SWEEP && t >= steps) {  // set up band `band` of candidate g (the only band unless TILED)
            {
                const int pad = (int)(n_bands * GS * R) - (int)L;
                const int row0 = (int)(band * GS * R + gl * R) - pad;  // read position of this lane's first row
                if (TILED) {  // this band's rows of the read
                    const uint8_t* read = a.bases + b0_read;
                    const int seg8 = (int)((L + 7) / 8);  // qP_word, ssw.c:336
                    bmask = 0;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int p = row0 + r;
                        qc[r] = p >= 0 ? strand_code(read, L, strand_read, (uint32_t)p) : 6u;
                        clampRow[r] = p >= 0 ? pk(0, -32768) : pk(0, 0);
                        if (p >= 0 && p % seg8 == 0) bmask |= 1u << r;
                    }
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int p = row0 + r;
                    h[r] = pk(0, p >= 0 ? -(p + 1) : 0);  // column 0: H = 0, D[i][0] = i
                }
                up_prev = pk(0, row0 > 0 ? -row0 : 0);  // row above this lane's first row, at column 0
                if (WORD) {
                    upw_prev = 0;
#pragma unroll
                    for (int r = 0; r < (WORD ? R : 1); r++) {
                        hw[r] = 0;
                        hwp[r] = 0;
                    }
                }
                if (TILED && band > 0) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the strip lane GS-1 wrote during the band above
                    if (gl == 0) bnd_next = strip[0];
                }
                const uint32_t lim = min(Wn, kRing);
                for (uint32_t base = 0; base < lim; base += 16 * GS) {  // 16 symbols per lane per pass
                    const uint32_t col = base + gl * 16;
                    if (col < lim) *reinterpret_cast<uint4*>(ring + col) = load16(ix.text, ix.n, wstart + col);
                }
                last = h[R - 1];  // min over the last row starts at D[L][0] (only lane 15 of the last band is read)
                t = 0;
                steps = Wn + GS - 1;
                c_next = ring[(0u - gl) & (kRing - 1)];
            }
        }
        if (phase == PH_SWEEP) {
            // windows longer than the ring: once every lane has crossed a half boundary b (lane 15 lags
            // by 15 columns), overwrite columns [b-512, b) with [b+512, b+1024)
            if (Wn > kRing && t >= kRing / 2 + GS && ((t - GS) & (kRing / 2 - 1)) == 0) {
                const uint32_t from = (t - GS) + kRing / 2, to = min(Wn, from + kRing / 2);
                for (uint32_t col = from + gl; col < to; col += GS) {
                    uint32_t pos = wstart + col;
                    ring[col & (kRing - 1)] = pos < ix.n ? ix.text[pos] : (uint8_t)7;
                }
            }
            // Exact early rejection, every 32 steps.  The lanes' current cells form a staircase frontier
            // (row i of lane g is at column t-1-g; a path can leave the swept region one column left of
            // it, hence the +-1 slack).  Any alignment touching unswept cells scores at most
            //   max_i H_i + 2 + min(rows left, columns left)   or   columns left (if it starts there),
            // and needs at least  min_i D_i - 2 + max(0, rows left - columns left)  edits.  If neither
            // the swept part nor that bound can reach the SW threshold, or both exceed the edit
            // tolerance, the candidate fails index.rs:406 or :410 whatever the remaining cells hold.
            if (!WORD && (t & 31) == 0 && t >= 32 && t < Wn) {
                const int jl = (int)t - 1 - (int)gl;                       // last column this lane has swept (-1: none)
                const int jc = min(jl, (int)Wn - 1);
                const int cols_left = (int)Wn - 1 - jc;
                const int L_ = (int)L;
                const int pad = (int)(GS * R) - L_;
                int sw_ub = 0, ed_lb = 0x7fff;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int p = (int)(gl * R + r) - pad;
                    int rows_left = L_ - 1 - p;
                    int hsw = h[r].x, hd = -(int)h[r].y;
                    int s1 = hsw + 2 + min(rows_left, cols_left);
                    int e1 = hd - 2 + max(0, rows_left - cols_left);
                    sw_ub = max(sw_ub, p >= 0 ? s1 : 0);
                    ed_lb = min(ed_lb, p >= 0 ? e1 : 0x7fff);
                }
                if (gl == 0) ed_lb = min(ed_lb, max(0, L_ - cols_left));  // paths still in the free first row
                sw_ub = max(sw_ub, (int)best.x);
                if (gl == GS - 1) ed_lb = min(ed_lb, -(int)last.y);     // best of the last row so far
#pragma unroll
                for (int d = 1; d < GS; d <<= 1) {
                    sw_ub = max(sw_ub, __shfl_xor(sw_ub, d));
                    ed_lb = min(ed_lb, __shfl_xor(ed_lb, d));
                }
                sw_ub = max(sw_ub, (int)Wn - (int)t + GS);          // alignments that start in unswept columns
                if (thr_wrapped || sw_ub < (int)thr || ed_lb > (int)ED) {
                    if (gl == 0) a.cand_status[g] = 1;
                    const uint32_t nxt = a.cand_next[g];  // next candidate of this TaxId in rank order (index.rs:393)
                    if (nxt != 0xffffffffu && nxt < maxc) {
                        g = o + nxt;
                        phase = PH_CHAIN;
                    } else {
                        phase = PH_FETCH;
                    }
                }
            }
        }
        if (phase == PH_SWEEP) {
#pragma unroll
          for (int rep = 0; rep < 2; rep++) {  // two columns per trip of the phase loop
            const uint32_t c = c_next;
            const int j = (int)t - (int)gl;
            c_next = ring[(uint32_t)(j + 1) & (kRing - 1)];
            int in_bits = row_shr1(pk_bits(h[R - 1]));
            if (GS < 16 && gl == 0) in_bits = 0;  // a DPP row holds two groups: cut the shift at the group edge
            int inw = 0;
            if (WORD) inw = row_shr1((hw[(WORD ? R : 1) - 1] << 16) | hwp[(WORD ? R : 1) - 1]);  // (post, pre) of the row above
            if (WORD && GS < 16 && gl == 0) inw = 0;
            if (TILED && band > 0 && gl == 0 && j >= 0 && j < (int)Wn) {  // bottom row of the band above, column j
                in_bits = (int)bnd_next.x;
                inw = (int)bnd_next.y;
                if (j + 1 < (int)Wn) bnd_next = strip[j + 1];
            }
            const pk16 in = pk_from_bits(in_bits);  // lane 0 of the first band reads 0: H = 0, D[0][j] = 0
            if (j >= 0 && j < (int)Wn) {
                const pk16 mvc = c == kCodeN ? pk(1, -1) : pk(1, 0);  // N/N: +1 in SW, never a match in edit distance
                pk16 diag = up_prev, up = in;
                bool eq[R];
#pragma unroll
                for (int r = 0; r < R; r++) eq[r] = qc[r] == c;  // all compares first: no VCC hazard stalls
#pragma unroll
                for (int r = 0; r < R; r++) {
                    pk16 sv = eq[r] ? mvc : miss;
                    pk16 x = diag + sv;
                    pk16 y = pk_max(up, h[r]) - one;
                    pk16 v = pk_max(pk_max(x, y), clampRow[r]);
                    diag = h[r];
                    h[r] = v;
                    up = v;
                    best = pk_max(best, v);
                }
                last = pk_max(last, h[R - 1]);
                up_prev = in;
                if (WORD) {
                    int diagw = upw_prev, uppre = inw & 0xffff;
#pragma unroll
                    for (int r = 0; r < (WORD ? R : 1); r++) {
                        int sc = qc[r] == c ? 1 : -1;
                        int hb = max(max(diagw + sc, hwp[r] - 1), 0);
                        int hh = max(hb, uppre - 1);
                        int pre = ((bmask >> r) & 1u) ? hb : hh;
                        diagw = hw[r];
                        hw[r] = hh;
                        hwp[r] = pre;
                        uppre = pre;
                        bestw = max(bestw, hh);
                    }
                    upw_prev = inw >> 16;
                }
                if (TILED && gl == GS - 1 && band + 1 < n_bands)
                    strip[j] = make_uint2((uint32_t)pk_bits(h[R - 1]), ((uint32_t)hw[(WORD ? R : 1) - 1] << 16) | (uint32_t)hwp[(WORD ? R : 1) - 1]);
            }
            t++;
            if (t >= steps) break;
          }
            if (TILED && t >= steps && band + 1 < n_bands) {
                band++;  // next band of the same candidate: set up at the top of the loop (t >= steps still holds)
            } else if (t >= steps) {
                int sw = best.x;
#pragma unroll
                for (int d = 1; d < GS; d <<= 1) sw = max(sw, __shfl_xor(sw, d));
                const int lastv = __shfl((int)last.y, (int)(lane | (GS - 1)));
                const uint32_t ed = (uint32_t)(-lastv);
                if (WORD) {
                    int w2 = bestw;
#pragma unroll
                    for (int d = 1; d < GS; d <<= 1) w2 = max(w2, __shfl_xor(w2, d));
                    if (sw >= 254) sw = w2;  // byte kernel overflowed -> sw_sse2_word's score (ssw.c:789-792)
                }
                const bool pass = !thr_wrapped && (uint32_t)sw >= thr && ed <= ED;  // index.rs:406,410
                phase = PH_FETCH;
                if (pass) {
                    if (gl == 0) {
                        a.out[g] = make_uint4(c_tax, c_gi, c_off, ed);
                        a.cand_status[g] = 2;
                    }
                } else {
                    if (gl == 0) a.cand_status[g] = 1;
                    const uint32_t nxt = a.cand_next[g];  // next candidate of this TaxId in rank order (index.rs:393)
                    if (nxt != 0xffffffffu && nxt < maxc) {
                        g = o + nxt;
                        phase = PH_CHAIN;
                    }
                }
            }
        }
    }
    if (gl != 0) {
        verified = 0;
        wbytes = 0;
    }
    for (int d = 32; d > 0; d >>= 1) {
        verified += __shfl_down(verified, d);
        wbytes += __shfl_down(wbytes, d);
    }
    if (lane == 0 && verified) {
        atomicAdd(a.n_verified, verified);
        atomicAdd(a.window_bytes, wbytes);
    }
}

// ---------------------------------------------------------------------------------------------
// K4b: the SW prefilter alone, two candidates per sweep (reads up to 253 bases, where the score of
// ssw.c's byte kernel is the exact local score -- see k_evaluate).  A 16-lane group owns two
// candidates at a time, A in the low and B in the high 16-bit half of every register; the two
// recurrences share all packed instructions:
//     H + 1 = max(1, diag + 2*[match], up + 1 - 1, left + 1 - 1)            (ssw.c:123-328 with the
//     matrix of ssw/src/lib.rs:11-16 and gap 1/1; codes are kept pre-shifted so that one saturating
//     subtraction of the XOR of read and window code yields 2*[match])
// The reference consumes only the predicate score >= L - 2*ED (index.rs:406), so a half is decided as
// soon as its running maximum reaches the threshold, or an exact bound shows it cannot any more;
// the sweep of a pair ends when both halves are decided.  A candidate that passes goes to pass_list
// for the edit distance (k_edit_myers in list mode, index.rs:407-410); one that fails hands the half
// to the next candidate of its TaxId (index.rs:393), as in k_evaluate.
// ---------------------------------------------------------------------------------------------
typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
__device__ inline upk16 as_u2(uint32_t v) { return __builtin_bit_cast(upk16, v); }
__device__ inline uint32_t as_bits(upk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ inline uint32_t pku_max(uint32_t a, uint32_t b) { return as_bits(__builtin_elementwise_max(as_u2(a), as_u2(b))); }
__device__ inline uint32_t pku_min(uint32_t a, uint32_t b) { return as_bits(__builtin_elementwise_min(as_u2(a), as_u2(b))); }
__device__ inline uint32_t pku_add(uint32_t a, uint32_t b) { return as_bits(as_u2(a) + as_u2(b)); }
__device__ inline uint32_t pku_sub(uint32_t a, uint32_t b) { return as_bits(as_u2(a) - as_u2(b)); }
__device__ inline uint32_t pku_satsub(uint32_t a, uint32_t b) { return as_bits(__builtin_elementwise_sub_sat(as_u2(a), as_u2(b))); }

enum : uint32_t { HF_EMPTY = 0, HF_SWEEP = 1, HF_PASS = 2, HF_FAIL = 3, HF_UNDEC = 4 };
constexpr uint32_t kSweepFlag = 0x80000000u;  // worklist entry from k_sw_diag: verified/window_bytes counted, bounds tried

// maximum over the 16 lanes of a DPP row, result in every lane: quad swaps, then the two mirrors
__device__ inline int row_max16(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}

#ifndef MTSV_SW_OCC
#define MTSV_SW_OCC 3
#endif
constexpr int kSwOcc = MTSV_SW_OCC;  // resident workgroups of k_sw_pairs per CU
constexpr int kSwTopOcc = 4;         // ... of its top-half instantiations (half the rows: 4 x 40 KiB of LDS fill the CU; 10.2 -> 10.0 ms)
constexpr uint32_t kRingP = 512;  // window bytes resident per candidate in k_sw_pairs
#ifndef MTSV_SW_DECIDE
#define MTSV_SW_DECIDE 8
#endif
#ifndef MTSV_SW_SCHED
#define MTSV_SW_SCHED 1
#endif
#ifndef MTSV_SW_SLACK
#define MTSV_SW_SLACK 4
#endif
constexpr uint32_t kDecide = MTSV_SW_DECIDE;  // columns between decision points after the two scheduled ones

constexpr uint32_t kPend = 64;   // passed candidates a group of k_sw_pairs buffers before one atomic
constexpr uint32_t kClaim = 32;  // work items a group of k_sw_pairs claims with one atomic, at most

// Intra-wavefront LDS hand-off: data one lane wrote is read by other lanes of the same wavefront next.
// LDS operations of a wavefront execute in order, so no instruction is needed; the fences keep the
// compiler from moving the accesses across the hand-off (without them the accesses are a data race
// in the language's memory model).
__device__ inline void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DIAG: try the lower bounds on the seed diagonal before a sweep (default; MTSV_SW_DIAG=0 launches the
// instantiation without them, which tests use to check that bounds and sweeps decide alike).
// k_sw_pairs reaches its counters through one base pointer (slots of batch.hip's d_counters): six separate
// pointers cost twelve scalar registers of a kernel that sits at the SGPR file's limit.
__device__ inline uint32_t* sw_cursor(const EvalArgs& a) { return reinterpret_cast<uint32_t*>(a.counters + kCtrSwCursor); }
__device__ inline uint32_t* sw_pass_count(const EvalArgs& a) { return reinterpret_cast<uint32_t*>(a.counters + kCtrPassCount); }

// TOP: the group's 16*R rows are the TOP of a longer read (the launcher halves R).  A candidate without a real
// alignment -- nearly everything that reaches a sweep once k_sw_diag has run -- is refuted on those rows alone
// at half the instructions: an alignment that reaches the threshold L - 2*ED starts within the first 2*ED read
// rows, so it runs through the covered rows and leaves them through the frontier of the sweep (bounded as in
// the full sweep, with the rows of the whole read below) or through the bottom covered row, where H never
// exceeded `bot`: bot + (L - 16*R) < threshold refutes the latter.  What the covered rows cannot decide goes to
// und_list (flagged like k_sw_diag's output) for the full-height launch.  Reads that fit the rows are decided
// exactly as without TOP.
template <int R, bool DIAG, bool TOP = false>
__global__ __launch_bounds__(256, R > 10 ? 3 : (TOP ? kSwTopOcc : kSwOcc)) void k_sw_pairs(DevIndexView ix, EvalArgs a) {
    static_assert(!(TOP && DIAG), "the bounds on the seed diagonal need the whole read");
    constexpr int GS = 16;
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[256 / GS][2][kRingP];
    __shared__ uint32_t pend_all[256 / GS][kPend];  // a group's passed candidates, flushed with one atomic per kPend
    __shared__ uint32_t meta_all[256 / GS][kClaim][8];    // the group's claimed work items, staged by its lanes
    __shared__ uint32_t und_all[TOP ? 256 / GS : 1][TOP ? kPend : 1];  // TOP: a group's undecided candidates
    const uint32_t lane = lane_id();
    const uint32_t gl = lane & (GS - 1);
    uint32_t* pend_buf = pend_all[threadIdx.x / GS];
    uint32_t pend = 0;  // group-uniform
    uint32_t* und_buf = und_all[TOP ? threadIdx.x / GS : 0];
    uint32_t und = 0;   // group-uniform
    auto und_push = [&](uint32_t gq) {
        if (gl == 0) und_buf[und] = gq | kSweepFlag;
        und++;
        if (und == kPend) {
            uint32_t base = 0;
            if (gl == 0) base = atomicAdd(reinterpret_cast<uint32_t*>(a.counters + a.und_slot), kPend);
            base = (uint32_t)__shfl((int)base, (int)(lane & ~(uint32_t)(GS - 1)));
            wave_lds_handoff();
            for (uint32_t i = gl; i < kPend; i += GS) a.und_list[base + i] = und_buf[i];
            wave_lds_handoff();
            und = 0;
        }
    };
    // append candidate gq to the pass list (all lanes of the group call it together)
    auto pass_push = [&](uint32_t gq) {
        if (gl == 0) pend_buf[pend] = gq;
        pend++;
        if (pend == kPend) {
            uint32_t base = 0;
            if (gl == 0) base = atomicAdd(sw_pass_count(a), kPend);
            base = (uint32_t)__shfl((int)base, (int)(lane & ~(uint32_t)(GS - 1)));
            wave_lds_handoff();  // lane 0's buffered entries are read by the whole group
            for (uint32_t i = gl; i < kPend; i += GS) a.pass_list[base + i] = pend_buf[i];
            wave_lds_handoff();  // ... before lane 0 overwrites them
            pend = 0;
        }
    };
    const uint32_t n_work = *reinterpret_cast<const uint32_t*>(a.counters + a.wl_count_slot);
    const uint32_t maxc = a.maxc;  // max_candidates as a rank bound, clamped on the host
    const uint32_t ONE = 0x00010001u, TWO = 0x00020002u;
    // items per claim: large enough that the claims do not bound the kernel, small enough that the last
    // claims of the launch stay balanced (two or more claims per group: claim_shift = log2(groups * 2); small
    // worklists -- slices of a host batch -- must not fall back to tiny claims, ~11 ns of atomic each)
    const uint32_t chunk = min(max(n_work >> a.claim_shift, 4u), kClaim);

    uint32_t phase = PH_FETCH;
    uint32_t loc = 0, loc_end = 0;  // position inside the claimed slice
    // per-half state; index 0 = low half
    uint32_t st[2] = {HF_EMPTY, HF_EMPTY};
    uint32_t g[2] = {0, 0}, o[2] = {0, 0}, Wn[2] = {0, 0};
    uint32_t geo = 0;  // read length (<= 253) and SW threshold of both halves, one byte each: L0 L1 thr0 thr1
    auto Lh = [&](int hf) { return (geo >> (8 * hf)) & 0xffu; };
    auto thrh = [&](int hf) { return (geo >> (16 + 8 * hf)) & 0xffu; };
    // group-uniform flags kept as bits of one VGPR (as separate bools they live in SGPR pairs, and this
    // kernel is already at the SGPR limit): bit hf = chain pending, 2 + hf = hopeless read, 4 = worklist drained
    uint32_t gf = 0;
    constexpr uint32_t F_CHAIN = 1u, F_HOPELESS = 4u, F_DRAINED = 16u, F_LONG = 32u;  // F_LONG: a window of this pair exceeds the ring
    uint32_t qc2[R];  // read codes << 1, both halves
    uint32_t h[R], up_prev = 0, best = 0;
    uint32_t bot = 0;  // TOP: largest H of the lane's last row over the swept columns (lane 15: the bottom covered row)
    uint32_t t = 0, steps = 0;
    uint32_t checks = 0;  // next decision point (low 16 bits, saturated) and the scheduled second one (high 16 bits)
    uint32_t verified = 0, wbytes = 0;  // per group: far below 2^32 in one launch
    unsigned long long swept = 0;       // columns this group's sweeps advanced (a column = R rows of both halves)
#pragma unroll
    for (int r = 0; r < R; r++) {
        qc2[r] = 0x000c000cu;  // code 6 << 1 in both halves: matches nothing
        h[r] = 0;
    }

    for (;;) {
        // The four groups of a wavefront fetch together: a group whose pair is decided waits until its
        // siblings' sweeps are over.  A fetch executed for one group costs the wavefront as many
        // instructions as one executed for all four, and fetching is ~40 % of this kernel's instructions.
        {
            const bool sweeping = __any(phase == PH_SWEEP);
            if (phase == PH_WAIT && !sweeping) phase = PH_FETCH;
        }
        if (__builtin_expect(phase == PH_FETCH, 0)) {
            uint8_t* rings = &ring_all[threadIdx.x / GS][0][0];
            uint32_t* meta = &meta_all[threadIdx.x / GS][0][0];
            // Two passes so that the global loads of both halves are in flight together: pass 1 picks the
            // candidates and issues the loads (window text, read bytes), pass 2 consumes them.  Repeats
            // only when a freshly loaded read turns out to hold more N than the edit tolerance.
            for (;;) {
                uint32_t wf = 0;  // bit hf = this half takes a candidate in this pass, 2 + hf = with a new read
                uint32_t cx[2] = {0, 0}, cy[2] = {0, 0}, b0s[2] = {0, 0}, strands[2] = {0, 0};
                uint4 txt[2];
                constexpr int RW = (R + 3) / 4;  // dwords that hold a lane's R read bytes
                uint32_t raw[2][RW];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    txt[hf] = make_uint4(0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u);
                    if (st[hf] != HF_EMPTY) continue;
                    if (gf & (F_CHAIN << hf)) {  // next candidate of the same TaxId: same read rows, new window
                        const uint4 c = a.cand[g[hf]];
                        cx[hf] = c.x;
                        cy[hf] = c.y;
                        gf &= ~(F_CHAIN << hf);
                        wf |= 1u << hf;
                    } else if (!(gf & F_DRAINED)) {
                        if (loc == loc_end) {
                            // One atomic per `chunk` items: atomics on one address complete at ~11 ns each on
                            // this chip, so 4 items per claim put a floor of 55 ms under 20 M work items.
                            uint32_t base = 0;
                            if (gl == 0) base = atomicAdd(sw_cursor(a), chunk);
                            base = (uint32_t)__shfl((int)base, (int)(lane & ~(uint32_t)(GS - 1)));
                            const uint32_t first = min(base, n_work);
                            loc = 0;
                            loc_end = min(base + chunk, n_work) - first;
                            // the lanes of the group walk the dependent loads of one item each, side by side
                            for (uint32_t i = gl; i < loc_end; i += GS) {
                                // wl_reverse: last entries first -- the coalescing kernels append the strands with the most
                                // seed hits (long same-TaxId chains, long merged windows) last, and the longest items
                                // should start first, not finish the launch alone
                                const uint32_t ge = a.worklist[a.wl_reverse ? n_work - 1 - (first + i) : first + i];
                                const uint32_t gi = ge & ~kSweepFlag;  // flagged: k_sw_diag counted it and its bounds do not decide it
                                const uint4 c = a.cand[gi];
                                const uint32_t rs = c.w;
                                const uint32_t r_ = a.r0 + (rs >> 1);
                                const uint32_t b0 = a.read_off[r_];
                                uint32_t* m = meta + i * 8;
                                m[0] = gi;
                                m[1] = c.x;
                                m[2] = c.y;
                                m[3] = a.strand_off[rs];
                                m[4] = b0;
                                const uint32_t Li = a.read_off[r_ + 1] - b0;
                                const uint32_t EDi = (uint32_t)ceil((double)Li * a.edit_rate);  // index.rs:281-282
                                m[5] = Li;
                                // strand, usize wrap of index.rs:406 (2*ED > L: nothing can pass), ED, threshold L - 2*ED
                                m[6] = (rs & 1) | (2ull * EDi > (uint64_t)Li ? 2u : 0u) | ((ge >> 31) << 2) | ((EDi & 0xffu) << 8) | (((Li - 2 * EDi) & 0xffu) << 16);
                            }
                            wave_lds_handoff();  // every lane of the group reads the staged items
                        }
                        if (loc == loc_end) {
                            gf |= F_DRAINED;
                        } else {
                            const uint32_t* m = meta + loc * 8;
                            loc++;
                            g[hf] = m[0];
                            cx[hf] = m[1];
                            cy[hf] = m[2];
                            o[hf] = m[3];
                            b0s[hf] = m[4];
                            geo = (geo & ~(0xffu << (8 * hf))) | ((m[5] & 0xffu) << (8 * hf));
                            strands[hf] = m[6];  // bit 0 strand, bit 1 wrapped, bit 2 counted and bounds tried (k_sw_diag), bits 8-15 ED, 16-23 threshold
                            geo = (geo & ~(0xffu << (16 + 8 * hf))) | (((m[6] >> 16) & 0xffu) << (16 + 8 * hf));
                            wf |= 5u << hf;
                        }
                    }
                    if (wf & (1u << hf)) {
                        if (gl * 16 < cy[hf] - cx[hf]) txt[hf] = load16(ix.text, ix.n, cx[hf] + gl * 16);
                        if (wf & (4u << hf)) {
                            // the lane's rows are R consecutive read bytes (descending for the reverse strand):
                            // aligned dword loads + v_alignbyte; byte k of raw[] is read byte s0 + k
                            const int pad = TOP ? max((int)(GS * R) - (int)Lh(hf), 0) : (int)(GS * R) - (int)Lh(hf);
                            const int p0 = (int)(gl * R) - pad;  // read position of row 0, < 0 for padding rows
                            const int s0 = (strands[hf] & 1u) ? (int)Lh(hf) - 1 - p0 - (R - 1) : p0;
                            const long long byte0 = (long long)b0s[hf] + s0;
                            const long long w0 = byte0 >> 2;
                            const uint32_t sh = (uint32_t)(byte0 & 3);
                            const uint32_t* b32 = reinterpret_cast<const uint32_t*>(a.bases);
                            uint32_t d[RW + 1];
#pragma unroll
                            for (int k = 0; k <= RW; k++) d[k] = w0 + k >= 0 ? b32[w0 + k] : 0u;
#pragma unroll
                            for (int k = 0; k < RW; k++) raw[hf][k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
                        }
                    }
                }
                bool again = false;
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    if (!(wf & (1u << hf))) continue;
                    if (wf & (4u << hf)) {
                        const uint32_t ED = (strands[hf] >> 8) & 0xffu;
                        const bool wrapped = (strands[hf] & 2u) != 0;
                        const int pad = TOP ? max((int)(GS * R) - (int)Lh(hf), 0) : (int)(GS * R) - (int)Lh(hf);
                        int nn = 0;
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const int p = (int)(gl * R + r) - pad;
                            const uint32_t kf = (uint32_t)r, kr = (uint32_t)(R - 1 - r);  // byte of this row in raw[]
                            const uint32_t bf = (raw[hf][kf >> 2] >> (8 * (kf & 3))) & 0xffu;
                            const uint32_t br = (raw[hf][kr >> 2] >> (8 * (kr & 3))) & 0xffu;
                            uint32_t code = (strands[hf] & 1u) ? br : bf;  // already a code (k_normalise)
                            if (strands[hf] & 1u) code = comp_code(code);
                            if (p < 0) code = 6u;
                            nn += code == kCodeN;
                            qc2[r] = hf ? ((qc2[r] & 0x0000ffffu) | (code << 17)) : ((qc2[r] & 0xffff0000u) | (code << 1));
                        }
                        nn = row_sum16(nn);
                        // more N in the read than the edit tolerance: index.rs:410 fails whatever the prefilter says
                        gf = (wrapped || (uint32_t)nn > ED) ? (gf | (F_HOPELESS << hf)) : (gf & ~(F_HOPELESS << hf));
                    }
                    if (!(strands[hf] & 4u)) {
                        verified++;
                        wbytes += cy[hf] - cx[hf];
                    }
                    if (gf & (F_HOPELESS << hf)) {  // rejected without a sweep; the reference still ran its prefilter on it
                        if (gl == 0) a.cand_status[g[hf]] = 1;
                        const uint32_t nxt = a.cand_next[g[hf]];
                        if (nxt != 0xffffffffu && nxt < maxc) {
                            g[hf] = o[hf] + nxt;
                            gf |= F_CHAIN << hf;
                        }
                        again = again || (gf & (F_CHAIN << hf)) || !(gf & F_DRAINED);
                    } else {
                        Wn[hf] = cy[hf] - cx[hf];
                        const uint32_t lim = min(Wn[hf], kRingP);
                        if (gl * 16 < lim) *reinterpret_cast<uint4*>(rings + hf * kRingP + gl * 16) = txt[hf];
                        static_assert(kRingP == 2 * 16 * GS, "the ring holds two passes of 16 symbols per lane");
                        {  // windows beyond 256 symbols: the second half of the ring
                            const uint32_t col = 16 * GS + gl * 16;
                            if (col < lim) *reinterpret_cast<uint4*>(rings + hf * kRingP + col) = load16(ix.text, ix.n, cx[hf] + col);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // other lanes' ring bytes are read next
                        // A lower bound decides most true candidates without the sweep: the ungapped alignment
                        // on the diagonal the window was cut around (window start = seed site - (q + ED),
                        // index.rs:128-133, so read position 0 faces window column ED) scores L - 2*mismatches,
                        // and the local score is at least that.  mismatches <= ED  =>  score >= L - 2*ED.
                        const uint32_t Lq = Lh(hf), EDq = (Lq - thrh(hf)) / 2;
                        bool decided = false;
                        if (DIAG && !(strands[hf] & 4u) && EDq + Lq <= lim) {
                            const int pad = (int)(GS * R) - (int)Lq;
                            // the lane's R window symbols on that diagonal (and its two neighbours): aligned dwords +
                            // v_alignbyte; sy[d] faces read position p with window column ED + p + d - 1
                            const int base = (int)EDq + (int)(gl * R) - pad - 1;  // negative for padding rows (ignored below)
                            const uint32_t* ring32 = reinterpret_cast<const uint32_t*>(rings + hf * kRingP);
                            constexpr int RD = (R + 2 + 3) / 4 + 1;
                            uint32_t dd[RD + 1];
#pragma unroll
                            for (int k = 0; k <= RD; k++) dd[k] = ring32[(uint32_t)((base >> 2) + k) & (kRingP / 4 - 1)];
                            int mm[3] = {0, 0, 0};
#pragma unroll
                            for (int d = 0; d < 3; d++) {
                                const int bd = base + d;
                                const int skip = (bd >> 2) - (base >> 2);  // 0 or 1 dwords further into dd[]
                                uint32_t sy[(R + 3) / 4];
#pragma unroll
                                for (int k = 0; k < (R + 3) / 4; k++) {
                                    const uint32_t lo_ = skip ? dd[k + 1] : dd[k], hi_ = skip ? dd[k + 2] : dd[k + 1];
                                    sy[k] = __builtin_amdgcn_alignbyte(hi_, lo_, (uint32_t)bd & 3u);
                                }
#pragma unroll
                                for (int r = 0; r < R; r++) {
                                    const int p = (int)(gl * R + r) - pad;
                                    const uint32_t sym = (sy[r >> 2] >> (8 * (r & 3))) & 0xffu;
                                    const uint32_t code = (qc2[r] >> (16 * hf + 1)) & 7u;
                                    mm[d] += (p >= 0 && code != sym) ? 1 : 0;
                                }
                            }
                            const int mm0 = row_sum16(mm[1]);
                            decided = (uint32_t)mm0 <= EDq;
                            // One gap: read rows of the lanes below s on the diagonal, the rest on a neighbouring one
                            // (a base missing from the read, or an extra one), joined by a gap of one.  That alignment
                            // scores at least L - 2 - 2*mismatches, so mismatches <= ED - 1 still proves the
                            // threshold; the split s is tried at every lane boundary with two row scans.
                            // The window starts at the smallest start any of its seed hits asks for, so the
                            // read's first rows sit on the middle diagonal when the window's later part is
                            // shifted right (a base missing from the read) and on the right-hand one when it is
                            // shifted left (an extra base in the read): both orders of every adjacent pair.
                            if (!decided && EDq >= 1 && EDq + Lq + 1 <= lim) {
                                int pre[3], suf[3];
#pragma unroll
                                for (int d = 0; d < 3; d++) {
                                    int v = mm[d];  // inclusive prefix sum over the lanes of the row
                                    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
                                    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
                                    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
                                    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
                                    pre[d] = v - mm[d];  // lanes below this one
                                    int w = mm[d];  // inclusive suffix sum
                                    w += __builtin_amdgcn_update_dpp(0, w, 0x101, 0xf, 0xf, true);  // row_shl:1
                                    w += __builtin_amdgcn_update_dpp(0, w, 0x102, 0xf, 0xf, true);  // row_shl:2
                                    w += __builtin_amdgcn_update_dpp(0, w, 0x104, 0xf, 0xf, true);  // row_shl:4
                                    w += __builtin_amdgcn_update_dpp(0, w, 0x108, 0xf, 0xf, true);  // row_shl:8
                                    suf[d] = w;
                                }
                                // split before this lane: lanes < gl on one diagonal, this lane and above on its neighbour
                                int best_mm = min(min(pre[1] + suf[0], pre[1] + suf[2]), min(pre[0] + suf[1], pre[2] + suf[1]));
                                best_mm = -row_max16(-best_mm);
                                decided = best_mm <= (int)EDq - 1;
                            }
                        }
                        if (decided) {
                            pass_push(g[hf]);  // its edit distance is k_edit_myers' business (index.rs:407-410)
                            again = again || !(gf & F_DRAINED);
                        } else if (TOP && Lq > (uint32_t)(GS * R) && Lq - (uint32_t)(GS * R) >= thrh(hf)) {
                            // an alignment below the covered rows alone could reach the threshold: nothing to refute here
                            und_push(g[hf]);
                            again = again || !(gf & F_DRAINED);
                        } else {
                            st[hf] = HF_SWEEP;
                        }
                    }
                }
                if (!again) break;
            }
            if (st[0] == HF_SWEEP || st[1] == HF_SWEEP) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++)
                    if (st[hf] != HF_SWEEP) Wn[hf] = 0;
#pragma unroll
                for (int r = 0; r < R; r++) h[r] = 0;
                up_prev = 0;
                best = 0;
                bot = 0;
                t = 0;
                steps = max(Wn[0], Wn[1]) + GS - 1;
                gf = max(Wn[0], Wn[1]) > kRingP ? (gf | F_LONG) : (gf & ~F_LONG);
                // Decision points: the first where a window without a real alignment typically runs out of
                // columns (Wn - thr columns swept, plus the lane skew and the score such a window reaches
                // by chance), the second where a true alignment typically reaches the threshold (its
                // start offset ~(Wn - L)/2, thr rows further down, in lane thr/R), then every kDecide.
                {
                    uint32_t tj = 0, tt = 0;
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) {
                        if (st[hf] != HF_SWEEP) continue;
                        const uint32_t slack = Wn[hf] > Lh(hf) ? (Wn[hf] - Lh(hf)) / 2 : 0;
                        tj = max(tj, (Wn[hf] > thrh(hf) ? Wn[hf] - thrh(hf) : 0) + GS + MTSV_SW_SLACK);
                        tt = max(tt, slack + thrh(hf) + thrh(hf) / R + 10);
                    }
                    checks = min(max(32u, (tj + 3) & ~3u), 0xfffcu) | (min((tt + 3) & ~3u, 0xfffcu) << 16);
                }
                phase = PH_SWEEP;
            } else {
                phase = PH_DONE;
            }
        }
        if (__all(phase == PH_DONE)) break;
        bool finish = false;
        // A sweeping group stays in this loop until its pair is decided (groups that are done wait at its end for
        // their siblings, as they would in the outer loop, which has nothing for a sweeping group to do between trips).
        if (phase == PH_SWEEP) do {
            uint8_t* rings = &ring_all[threadIdx.x / GS][0][0];
            // windows longer than the ring: see k_evaluate
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                if (__builtin_expect((gf & F_LONG) != 0, 0))
                if (Wn[hf] > kRingP && t >= kRingP / 2 + GS && ((t - GS) & (kRingP / 2 - 1)) == 0) {
                    const uint32_t from = (t - GS) + kRingP / 2, to = min(Wn[hf], from + kRingP / 2);
                    const uint32_t wstart = a.cand[g[hf]].x;  // long windows only: not worth a register
                    for (uint32_t col = from + gl; col < to; col += GS) {
                        uint32_t pos = wstart + col;
                        rings[hf * kRingP + (col & (kRingP - 1))] = pos < ix.n ? ix.text[pos] : (uint8_t)7;
                    }
                }
            }
            // Decision point (bounds as in k_evaluate): a half passes as soon as its
            // maximum reaches the threshold and fails as soon as no alignment through unswept cells can.
            if (__builtin_expect(t >= (checks & 0xffffu), 0)) {
                int bmax[2], ub[2];
                // A path that leaves the swept cells through row r of this lane (now at column jc) can add
                // at most min(rows below r, columns right of jc) to H[r]; the rows are bottom-aligned, so
                // row r of lane gl has (GS - gl) * R - 1 - r read rows below it whatever the read length.
                // The lane's last row is also left diagonally from its previous column, where H is at most
                // one larger: + 2.  Padding rows (H = 0) yield the column count, which the bound on
                // alignments that start in unswept columns covers anyway.  Both halves in packed u16.
                uint32_t hb;
                {
                    const uint32_t cl0 = (uint32_t)max((int)Wn[0] - (int)t + (int)gl, 0), cl1 = (uint32_t)max((int)Wn[1] - (int)t + (int)gl, 0);
                    const uint32_t cl = min(cl0, 0xffffu) | (min(cl1, 0xffffu) << 16);
                    uint32_t rl = ((GS - gl) * R - 1) * ONE;
                    if (TOP) {  // the rows of the whole read below row 0 of this lane: L - 1 - (gl * R - pad), pad = max(16 * R - L, 0)
                        const int r0_ = (int)Lh(0) - 1 + max((int)(GS * R) - (int)Lh(0), 0) - (int)(gl * R);
                        const int r1_ = (int)Lh(1) - 1 + max((int)(GS * R) - (int)Lh(1), 0) - (int)(gl * R);
                        rl = (uint32_t)max(r0_, 0) | ((uint32_t)max(r1_, 0) << 16);
                    }
                    hb = 0;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        uint32_t v = pku_add(h[r], pku_min(rl, cl));
                        if (r == R - 1) v = pku_add(v, TWO);
                        hb = pku_max(hb, v);
                        rl = pku_sub(rl, ONE);
                    }
                }
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    ub[hf] = (int)((hb >> (16 * hf)) & 0xffffu);
                    bmax[hf] = (int)((best >> (16 * hf)) & 0xffffu) - 1;  // best tracks H + 1
                }
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    bmax[hf] = row_max16(bmax[hf]);
                    ub[hf] = row_max16(ub[hf]);
                }
                int botv[2] = {0, 0};
                if (TOP) {
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) botv[hf] = row_max16(gl == GS - 1 ? (int)((bot >> (16 * hf)) & 0xffffu) : 0);
                }
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    if (st[hf] == HF_SWEEP) {
                        const int u = max(max(ub[hf], bmax[hf]), (int)Wn[hf] - (int)t + GS);  // or starts in unswept columns
                        // TOP: what left the covered rows through their last row cannot be refuted any more once H there got too large
                        const bool open_below = TOP && (int)Lh(hf) > GS * R && botv[hf] + ((int)Lh(hf) - GS * R) >= (int)thrh(hf);
                        if (bmax[hf] >= (int)thrh(hf)) st[hf] = HF_PASS;
                        else if (open_below) st[hf] = HF_UNDEC;
                        else if (u < (int)thrh(hf) || t >= Wn[hf] + GS - 1) st[hf] = HF_FAIL;
                    }
                }
                finish = st[0] != HF_SWEEP && st[1] != HF_SWEEP;
#if MTSV_SW_SCHED == 0
                checks = (checks >> 16) > t + 8 ? (checks >> 16) : min(t + kDecide, 0xfffcu);  // past 65 k columns: every trip
#else
                {
                    const uint32_t second = checks >> 16;
                    const uint32_t nx = min(t + kDecide, 0xfffcu);
                    checks = (second > t + 8 ? min(second, nx) : nx) | (second << 16);
                }
#endif
            }
            if (!finish) {
                // four columns per trip: this lane's columns j0 .. j0+3 of both windows come from the rings as two
                // aligned dwords + v_alignbyte each; columns outside [0, Wn) read as 7 (matches nothing)
                const int j0 = (int)t - (int)gl;
                // every lane's four columns inside both windows (a half that is not sweeping has Wn = 0 and is ignored):
                // most trips of a sweep, and then no edge masks are needed
                const bool interior = t >= (uint32_t)(GS - 1) && min(Wn[0] - 1u, Wn[1] - 1u) >= t + 3u;
                uint32_t four2[2];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const uint32_t* ring32 = reinterpret_cast<const uint32_t*>(rings + hf * kRingP);
                    const int w = j0 >> 2;  // arithmetic: floor for the negative columns of the pipeline fill
                    const uint32_t d0 = ring32[(uint32_t)w & (kRingP / 4 - 1)], d1 = ring32[(uint32_t)(w + 1) & (kRingP / 4 - 1)];
                    const uint32_t four = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)j0 & 3u);
                    if (interior) {
                        four2[hf] = four << 1;
                        continue;
                    }
                    const int lo = min(max(-j0, 0), 4), hi = min(max((int)Wn[hf] - j0, 0), 4);  // valid bytes: [lo, hi)
                    const uint32_t mlo = lo >= 4 ? 0xffffffffu : (1u << (8 * lo)) - 1u;
                    const uint32_t mhi = hi >= 4 ? 0xffffffffu : (1u << (8 * hi)) - 1u;
                    const uint32_t mask = mhi & ~mlo;
                    four2[hf] = ((four & mask) | (0x07070707u & ~mask)) << 1;  // codes << 1, no carry between bytes
                }
#pragma unroll
                for (int rep = 0; rep < 4; rep++) {
                    // byte `rep` of window A into the low half, of window B into the high half
                    const uint32_t cp2 = __builtin_amdgcn_perm(four2[1], four2[0], 0x0c040c00u + (uint32_t)rep * 0x00010001u);
                    const uint32_t in = (uint32_t)row_shr1((int)h[R - 1]);  // lane 0 of the group reads 0: H = 0
                    // off the dependent chain: T[r] = max(diag + 2*[match], left) from the previous column
                    uint32_t T[R];
                    {
                        uint32_t diag = up_prev;
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const uint32_t e2 = pku_satsub(TWO, qc2[r] ^ cp2);  // 2 where the codes are equal, else 0
                            T[r] = pku_max(pku_add(diag, e2), h[r]);
                            diag = h[r];
                        }
                    }
                    // the chain down the rows: H + 1 = max(T, up, 1), two dependent operations per row
                    uint32_t up = in;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const uint32_t w = pku_max(T[r], up);
                        best = pku_max(best, w);
                        h[r] = pku_satsub(w, ONE);  // max(w, 1) - 1
                        up = h[r];
                    }
                    if (TOP) bot = pku_max(bot, up);
                    up_prev = in;
                    t++;
                }
                finish = t >= steps;
            }
        } while (!finish);
        if (__builtin_expect(phase == PH_SWEEP && finish, 0)) {
            int bm[2];
            swept += t;
#ifdef MTSV_SW_HIST
            if (gl == 0) {
                uint32_t* hist = reinterpret_cast<uint32_t*>(a.strip);
                atomicAdd(&hist[min(t / 4, 127u)], 1u);
                atomicAdd(&hist[128 + (Wn[0] != 0) + (Wn[1] != 0)], 1u);
            }
#endif

#pragma unroll
            for (int hf = 0; hf < 2; hf++) bm[hf] = (int)((best >> (16 * hf)) & 0xffffu) - 1;
            bm[0] = row_max16(bm[0]);
            bm[1] = row_max16(bm[1]);
            int bv[2] = {0, 0};
            if (TOP) {
#pragma unroll
                for (int hf = 0; hf < 2; hf++) bv[hf] = row_max16(gl == GS - 1 ? (int)((bot >> (16 * hf)) & 0xffffu) : 0);
            }
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                if (st[hf] == HF_SWEEP) {
                    const bool open_below = TOP && (int)Lh(hf) > GS * R && bv[hf] + ((int)Lh(hf) - GS * R) >= (int)thrh(hf);
                    st[hf] = bm[hf] >= (int)thrh(hf) ? HF_PASS : (open_below ? HF_UNDEC : HF_FAIL);
                }
                if (TOP && st[hf] == HF_UNDEC) und_push(g[hf]);
#ifdef MTSV_SW_HIST
                if (gl == 0 && Wn[hf] != 0) atomicAdd(&reinterpret_cast<uint32_t*>(a.strip)[256 + (st[hf] == HF_PASS ? 256 : 0) + min(max(bm[hf], 0), 255)], 1u);
#endif
                if (st[hf] == HF_PASS) {
                    pass_push(g[hf]);
                } else if (st[hf] == HF_FAIL) {
                    if (gl == 0) a.cand_status[g[hf]] = 1;
                    const uint32_t nxt = a.cand_next[g[hf]];  // next candidate of this TaxId in rank order (index.rs:393)
                    if (nxt != 0xffffffffu && nxt < maxc) {
                        g[hf] = o[hf] + nxt;
                        gf |= F_CHAIN << hf;
                    }
                }
                st[hf] = HF_EMPTY;
            }
            phase = PH_WAIT;
        }
    }
    if (pend) {
        uint32_t base = 0;
        if (gl == 0) base = atomicAdd(sw_pass_count(a), pend);
        base = (uint32_t)__shfl((int)base, (int)(lane & ~(uint32_t)(GS - 1)));
        wave_lds_handoff();
        for (uint32_t i = gl; i < pend; i += GS) a.pass_list[base + i] = pend_buf[i];
    }
    if (TOP && und) {
        uint32_t base = 0;
        if (gl == 0) base = atomicAdd(reinterpret_cast<uint32_t*>(a.counters + a.und_slot), und);
        base = (uint32_t)__shfl((int)base, (int)(lane & ~(uint32_t)(GS - 1)));
        wave_lds_handoff();
        for (uint32_t i = gl; i < und; i += GS) a.und_list[base + i] = und_buf[i];
    }
    unsigned long long v64 = gl == 0 ? verified : 0, w64 = gl == 0 ? wbytes : 0, s64 = gl == 0 ? swept : 0;
    for (int d = 32; d > 0; d >>= 1) {
        v64 += __shfl_down(v64, d);
        w64 += __shfl_down(w64, d);
        s64 += __shfl_down(s64, d);
    }
    if (lane == 0 && v64) {
        atomicAdd((unsigned long long*)a.counters + kCtrVerified, v64);
        atomicAdd((unsigned long long*)a.counters + kCtrWindowBytes, w64);
    }
    // cell pairs swept (one packed 7-instruction recurrence each); also when every candidate came counted from k_sw_diag
    if (lane == 0 && s64) atomicAdd((unsigned long long*)a.counters + kCtrSwCellPairs, s64 * R);
}


// ---------------------------------------------------------------------------------------------
// K4a: the lower bounds of the prefilter on their own (first round of a pass), a 16-lane group per work item.
// Most candidates that will be accepted are the read's own origin: the read lies on the diagonal the
// window was cut around (read position 0 faces window column ED, index.rs:128-133) with substitutions
// only, or with one base missing or extra.  The ungapped alignment on that diagonal scores
// L - 2*mismatches, one that changes to a neighbouring diagonal once at least L - 2 - 2*mismatches, and
// the local score ssw.c computes is at least either: mismatches <= ED (ED - 1 with the gap) proves
// score >= L - 2*ED, the predicate of index.rs:406, without the sweep.  Such a candidate goes to
// pass_list (k_edit_myers computes its edit distance, :407-410); every other one to sweep_list with
// bit 31 set, for k_sw_pairs: "counted here, and the bounds do not decide it".
// Lane gl compares 8, 12 or 16 read positions (16 lanes cover the longest read of the pass) with the three diagonals, four codes per
// instruction (mismatch = the codes differ, as in the sweep: N faces N as a match,
// ssw/src/lib.rs:11-16); the change of diagonal is tried after every fourth read position, both
// orders of every adjacent pair (prefix counts of the lanes below + the best split inside the lane +
// suffix counts of the lanes above).  Decisions are buffered per wavefront: one atomic per ~120 entries.
// ---------------------------------------------------------------------------------------------
__device__ inline uint32_t nz_bytes(uint32_t x) { return (x | (x >> 1) | (x >> 2)) & 0x01010101u; }  // codes < 8
__device__ inline uint32_t comp_codes4(uint32_t x) {  // 3 - c for the bases, N stays (four codes 0..4)
    const uint32_t n = (x >> 2) & 0x01010101u;
    return x ^ ((n ^ 0x01010101u) * 3u);
}
__device__ inline int row_prefix_incl16(int v) {  // inclusive prefix sum over the 16 lanes of a DPP row
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    return v;
}
__device__ inline int row_suffix_incl16(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true);  // row_shl:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x102, 0xf, 0xf, true);  // row_shl:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0xf, true);  // row_shl:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x108, 0xf, 0xf, true);  // row_shl:8
    return v;
}

constexpr uint32_t kDiagBuf = 512;  // decisions a wavefront of k_sw_diag buffers per list (128: one returning atomic per ~120 entries = 147 k per 10 M reads on each counter, ~11 ns each)

// NW: words of four read positions per lane (16 * 4 * NW >= the longest read of the pass)
#ifndef MTSV_DIAG_OCC
#define MTSV_DIAG_OCC 6
#endif
template <int NW>
__global__ __launch_bounds__(256, MTSV_DIAG_OCC) void k_sw_diag(DevIndexView ix, EvalArgs a, uint32_t* __restrict__ sweep_list, uint32_t sweep_slot) {
    __shared__ uint32_t buf_all[256 / kWave][2][kDiagBuf];
    uint32_t* pbuf = buf_all[threadIdx.x / kWave][0];
    uint32_t* sbuf = buf_all[threadIdx.x / kWave][1];
    uint32_t np = 0, nsw = 0;  // wave-uniform fill of the two buffers
    const uint32_t n_work = *reinterpret_cast<const uint32_t*>(a.counters + a.wl_count_slot);
    const uint32_t* t32 = reinterpret_cast<const uint32_t*>(ix.text);
    const uint32_t* b32 = reinterpret_cast<const uint32_t*>(a.bases);
    const uint32_t lastw = (ix.n - 1) >> 2;  // text allocations are padded to a dword multiple
    const uint32_t lane = lane_id(), gl = lane & 15u;
    unsigned long long verified = 0, wbytes = 0;
    auto flush = [&](uint32_t* buf, uint32_t& n, uint32_t* count, uint32_t* list, uint32_t flag) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(count, n);
        base = (uint32_t)__shfl((int)base, 0);
        wave_lds_handoff();
        for (uint32_t j = lane; j < n; j += kWave) list[base + j] = buf[j] | flag;
        wave_lds_handoff();
        n = 0;
    };
    // The work item's description hangs on three dependent loads (worklist -> candidate -> read offsets); they run
    // three, two and one items ahead of the comparison, so that an item waits for its bases and text only.
    // Items in runs of 32 per wavefront (eight trips of four), the runs of all wavefronts side by side in a tile:
    // neighbours in the worklist are reads next to each other, and they stay neighbours in pass_list (k_edit_myers
    // works a lane per entry) while the order of the tiles -- from the end of the worklist, where the coalescing
    // kernels put the strands with the most seed hits -- carries over to sweep_list.
    auto item_at = [&](uint32_t i) { return a.worklist[a.wl_reverse ? n_work - 1 - i : i]; };
    const uint32_t n_waves = gridDim.x * (256 / kWave), wave_g = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    auto index_of = [&](uint32_t it) {  // 64-bit: past the last tile the index only has to compare as >= n_work
        return ((uint64_t)(it >> 3) * n_waves + wave_g) * 32 + (it & 7u) * 4 + (lane >> 4);
    };
    const uint32_t trips = (n_work + n_waves * 32 - 1) / (n_waves * 32) * 8;
    uint32_t gi_2 = 0, gi_1 = 0, gi_0 = 0;      // items two / one ahead, the current one
    uint4 c_1 = make_uint4(0, 0, 0, 0), c_0 = c_1;
    uint32_t b0_0 = 0, L_0 = 0;
    if (index_of(2) < n_work) gi_2 = item_at((uint32_t)index_of(2));
    if (index_of(1) < n_work) {
        gi_1 = item_at((uint32_t)index_of(1));
        c_1 = a.cand[gi_1];
    }
    if (index_of(0) < n_work) {
        gi_0 = item_at((uint32_t)index_of(0));
        c_0 = a.cand[gi_0];
        const uint32_t r_ = a.r0 + (c_0.w >> 1);
        b0_0 = a.read_off[r_];
        L_0 = a.read_off[r_ + 1] - b0_0;
    }
    for (uint32_t it = 0; it < trips; it++) {  // the same trip count for every wavefront
        const bool have = index_of(it) < n_work;  // group-uniform, like every branch below that holds a row operation
        // issue the loads of the items ahead
        uint32_t gi_3 = 0, b0_1 = 0, L_1 = 0;
        uint4 c_2 = make_uint4(0, 0, 0, 0);
        if (index_of(it + 3) < n_work) gi_3 = item_at((uint32_t)index_of(it + 3));
        if (index_of(it + 2) < n_work) c_2 = a.cand[gi_2];
        if (index_of(it + 1) < n_work) {
            const uint32_t r1 = a.r0 + (c_1.w >> 1);
            b0_1 = a.read_off[r1];
            L_1 = a.read_off[r1 + 1] - b0_1;
        }
        bool decided = false;
        const uint32_t gi = gi_0;
        if (have) {
            const uint4 c = c_0;
            const uint32_t rs = c.w;
            const uint32_t b0 = b0_0, L = L_0;
            const uint32_t ED = (uint32_t)ceil((double)L * a.edit_rate);  // index.rs:281-282
            const uint32_t W = c.y - c.x;
            if (gl == 0) {
                verified++;
                wbytes += W;
            }
            if (2ull * ED <= (uint64_t)L && ED + L <= W) {
                const uint32_t p0 = gl * (4 * NW);  // this lane's first read position
                // read codes of positions p0 .. p0+4*NW-1 (reverse strand: the complement of bytes b0+L-1-p0 downwards)
                uint32_t rd[NW];
                {
                    const bool rev = (rs & 1u) != 0;
                    // (lanes past the read compare nothing: they load the read's first bytes instead of running off the buffer;
                    //  word indices below 0 -- the tail of a reverse strand at the buffer's start -- are clamped, those bytes are masked)
                    const uint32_t pq = p0 < L ? p0 : 0u;
                    const long long ba = rev ? (long long)b0 + L - 4 * NW - pq : (long long)b0 + pq;
                    const long long w = ba >> 2;
                    const uint32_t sh = (uint32_t)(ba & 3);
                    uint32_t d[NW + 1];
#pragma unroll
                    for (int k = 0; k <= NW; k++) d[k] = b32[max(w + k, 0ll)];  // (the buffer is padded past its end)
                    uint32_t x[NW];
#pragma unroll
                    for (int k = 0; k < NW; k++) x[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
#pragma unroll
                    for (int k = 0; k < NW; k++) rd[k] = rev ? comp_codes4(__builtin_bswap32(x[NW - 1 - k])) : x[k];
                }
                // window bytes pc - 1 .. pc + 4*NW around the middle diagonal (byte pc + j faces read position p0 + j)
                const uint32_t pc = c.x + ED + p0;
                const uint32_t wi = pc >> 2, sc = pc & 3u;
                uint32_t wv[NW + 2];  // words wi - 1 .. wi + NW
                // (clamped, not guarded: every byte a valid position compares lies inside the window, hence inside the text;
                //  the byte before position 0 is only wrong when c.x + ED = 0, where the gap bound -- ED >= 1 -- is not tried)
                wv[0] = t32[wi >= 1 ? min(wi - 1, lastw) : 0u];
#pragma unroll
                for (int k = 0; k <= NW; k++) wv[k + 1] = t32[min(wi + k, lastw)];
                int ca[3][NW];  // mismatches per diagonal and word of four positions
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    const uint32_t lo = wv[k + 1], hi = wv[k + 2], bl = wv[k];
                    const uint32_t s0 = sc == 0 ? __builtin_amdgcn_alignbyte(lo, bl, 3u) : __builtin_amdgcn_alignbyte(hi, lo, sc - 1u);
                    const uint32_t s1 = __builtin_amdgcn_alignbyte(hi, lo, sc);
                    const uint32_t s2 = sc == 3 ? hi : __builtin_amdgcn_alignbyte(hi, lo, sc + 1u);
                    const uint32_t pk_ = p0 + 4 * k;
                    const uint32_t nv = pk_ < L ? min(L - pk_, 4u) : 0u;
                    const uint32_t valid = nv >= 4 ? 0x01010101u : ((1u << (8 * nv)) - 1u) & 0x01010101u;
                    ca[0][k] = __popc(nz_bytes(rd[k] ^ s0) & valid);
                    ca[1][k] = __popc(nz_bytes(rd[k] ^ s1) & valid);
                    ca[2][k] = __popc(nz_bytes(rd[k] ^ s2) & valid);
                }
                int mm[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    mm[d] = 0;
#pragma unroll
                    for (int k = 0; k < NW; k++) mm[d] += ca[d][k];
                }
                decided = (uint32_t)row_sum16(mm[1]) <= ED;
                if (!decided && ED >= 1 && ED + L + 1 <= W) {
                    int below[3], above[3];  // counts of the lanes below / above this one
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        below[d] = row_prefix_incl16(mm[d]) - mm[d];
                        above[d] = row_suffix_incl16(mm[d]) - mm[d];
                    }
                    // first positions on diagonal da, the rest on db, the change inside this lane (after 0 .. NW words)
                    auto split = [&](int da, int db) {
                        int pa = 0, sb = mm[db], best = mm[db];
#pragma unroll
                        for (int k = 0; k < NW; k++) {
                            pa += ca[da][k];
                            sb -= ca[db][k];
                            best = min(best, pa + sb);
                        }
                        return below[da] + best + above[db];
                    };
                    int best = min(min(split(1, 0), split(1, 2)), min(split(0, 1), split(2, 1)));
                    best = -row_max16(-best);
                    decided = best <= (int)ED - 1;
                }
            }
        }
        // one entry per group, from its first lane
        const unsigned long long pm = __ballot(have && gl == 0 && decided), sm = __ballot(have && gl == 0 && !decided);
        const unsigned long long lower = (1ull << lane) - 1ull;
        if (have && gl == 0 && decided) pbuf[np + (uint32_t)__popcll(pm & lower)] = gi;
        if (have && gl == 0 && !decided) sbuf[nsw + (uint32_t)__popcll(sm & lower)] = gi;
        np += (uint32_t)__popcll(pm);
        nsw += (uint32_t)__popcll(sm);
        if (np + 4 > kDiagBuf) flush(pbuf, np, sw_pass_count(a), a.pass_list, 0u);
        if (nsw + 4 > kDiagBuf) flush(sbuf, nsw, reinterpret_cast<uint32_t*>(a.counters + sweep_slot), sweep_list, kSweepFlag);
        gi_0 = gi_1;
        gi_1 = gi_2;
        gi_2 = gi_3;
        c_0 = c_1;
        c_1 = c_2;
        b0_0 = b0_1;
        L_0 = L_1;
    }
    if (np) flush(pbuf, np, sw_pass_count(a), a.pass_list, 0u);
    if (nsw) flush(sbuf, nsw, reinterpret_cast<uint32_t*>(a.counters + sweep_slot), sweep_list, kSweepFlag);
    for (int d = 32; d > 0; d >>= 1) {
        verified += __shfl_down(verified, d);
        wbytes += __shfl_down(wbytes, d);
    }
    if (lane == 0 && verified) {
        atomicAdd((unsigned long long*)a.counters + kCtrVerified, verified);
        atomicAdd((unsigned long long*)a.counters + kCtrWindowBytes, wbytes);
    }
}

// 16 window columns of a candidate in the order the sweep consumes them, as codes clamped to 0..4
// (4 = matches nothing).  Forward strand: text[start+j0 ..]; reverse strand: text[end-1-j0], downwards
// (complementing is folded into the match table).  Columns past the window are don't-cares.
__device__ inline uint32_t clamp_codes4(uint32_t x) {
    const uint32_t t = x & 0x04040404u;
    return t | (x & 0x03030303u & ~((t >> 1) | (t >> 2)));
}
// the same under the SW matrix's matches (ssw/src/lib.rs:11-16: a window N matches a read N): N becomes 5, the row of the
// match table that holds the read's N positions; everything above ('$', padding) 4 = matches nothing, as in clamp_codes4
__device__ inline uint32_t clamp_codes_sw(uint32_t x) {
    const uint32_t t = x & 0x04040404u, lo = x & 0x03030303u;
    const uint32_t nz = (lo | (lo >> 1)) & 0x01010101u, m = t >> 2;
    return t | (lo & ~(m * 3u)) | ((nz ^ 0x01010101u) & m);
}
// sw_matches: clamp for the SW matrix's matches (k_edit_myers bounding the prefilter) instead of the edit distance's
__device__ inline uint4 fetch_cols(const DevIndexView& ix, uint32_t start, uint32_t end, uint32_t strand, uint32_t j0, bool sw_matches) {
    uint4 r;
    if (!strand) {
        r = load16(ix.text, ix.n, start + j0);
    } else if (end >= j0 + 16) {
        const uint4 v = load16(ix.text, ix.n, end - j0 - 16);
        r.x = __builtin_amdgcn_perm(0, v.w, 0x00010203u);
        r.y = __builtin_amdgcn_perm(0, v.z, 0x00010203u);
        r.z = __builtin_amdgcn_perm(0, v.y, 0x00010203u);
        r.w = __builtin_amdgcn_perm(0, v.x, 0x00010203u);
    } else {  // fewer than 16 symbols left above text position 0 (only the first bins of an index)
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t code = end >= j0 + 1 + i ? ix.text[end - j0 - 1 - i] : 7u;
            w[i >> 2] |= code << (8 * (i & 3));
        }
        r = make_uint4(w[0], w[1], w[2], w[3]);
    }
    r.x = sw_matches ? clamp_codes_sw(r.x) : clamp_codes4(r.x);
    r.y = sw_matches ? clamp_codes_sw(r.y) : clamp_codes4(r.y);
    r.z = sw_matches ? clamp_codes_sw(r.z) : clamp_codes4(r.z);
    r.w = sw_matches ? clamp_codes_sw(r.w) : clamp_codes4(r.w);
    return r;
}

// ---------------------------------------------------------------------------------------------
// Alternative verification order ("edit first"): acceptance is the conjunction of two pure
// predicates, SW >= L - 2*ED and edits <= ED (index.rs:406,410), and for reads up to 253 bases the
// SW score is the exact local score, for which edits <= ED implies SW >= L - 2*ED (an alignment with
// e edits scores >= L - 2e; tests/test_oracle.py).  So for such reads edits <= ED alone decides.
// k_edit_myers computes Aligner::min_edit_distance (align.rs:28-85: first row 0, answer = min of
// the last row) with Myers' bit-vector recurrence, one lane per candidate, W 32-bit words per
// column, match masks in LDS.  Read 'N' matches nothing (index.rs:272-279), reference 'N' matches nothing either.
// ---------------------------------------------------------------------------------------------
// MY_LIST: the items are single candidates that already passed the SW prefilter (k_sw_pairs); one that
// fails index.rs:410 hands its TaxId's next candidate to the next round instead of walking on.
//
// MY_BOUND: the same recurrence as a two-sided bound on the SW prefilter's predicate (index.rs:406), for the candidates
// k_sw_diag's lower bounds left undecided -- nearly all of them chance seed hits that the DP sweep of k_sw_pairs
// refutes at several times the cost.  Matches are the SW matrix's (ssw/src/lib.rs:11-16: equal codes, a read N facing
// a window N included); D = the unit-cost semi-global distance of the read to the window under those matches.
//   (1) D <= ED  =>  score >= L - 2*ED.  The alignment behind D has m mismatches, i read bases and d window bases
//       against gaps, m + i + d = D, and L - m - i matches: as a local alignment it scores
//       (L - m - i) - m - i - d >= L - 2*D  (+1 match, -1 mismatch, a gap of g costs g).
//   (2) score >= L - 2*ED  =>  D <= 2*ED.  The best local alignment leaves u read rows uncovered and has M matches,
//       m mismatches, i and d gap bases: L = u + M + m + i and score = M - m - i - d = L - u - 2m - 2i - d.
//       Extended by the u uncovered read bases as insertions it is a semi-global alignment of the whole read with
//       u + m + i + d <= u + 2m + 2i + d = L - score <= 2*ED unit edits.
// So D <= ED passes the candidate on to the edit distance, D > 2*ED refutes it (its TaxId's next candidate is then
// bounded by the same lane, index.rs:393), and the few in between go to und_list for the sweep.  A chance candidate
// sits near D = 0.4*L, far above 2*ED = 0.26*L.  A lane leaves the columns early once D <= ED is reached, or once the
// last row can no longer come down to 2*ED in the columns that are left (neighbouring cells differ by at most 1).
enum : int { MY_CHAIN = 0, MY_LIST = 1, MY_BOUND = 2 };
template <int W, int MODE>  // W 32-bit words per column: reads of up to 32*W bases
__global__ __launch_bounds__(256) void k_edit_myers(DevIndexView ix, EvalArgs a) {
    // match masks of the lane's read, one row per window symbol (A C G T, other), word-major so that a
    // lane's reads are conflict-free whatever row it picks.  The read sits at the TOP of the 32*W rows
    // (its last base is bit 31 of word W-1); the rows below it are wildcard rows whose vertical deltas
    // start at 0: they stay 0 in every column, i.e. they reproduce the all-zero first row of
    // align.rs:28-85 right under the read's first base, and the running score is simply the carry out
    // of the last word.
    constexpr bool LIST = MODE == MY_LIST, BOUND = MODE == MY_BOUND;
    // rows 0-3: the bases; 4: matches nothing; 5 (bound and list mode): the read's N positions, which a window N matches in
    // the SW matrix
    constexpr int NROW = (BOUND || LIST) ? 6 : 5;
    __shared__ uint32_t eq_tab[NROW][W][256];
    const uint32_t tid = threadIdx.x;
    const uint32_t n_work = *a.wl_count;
    const uint32_t maxc = a.maxc;  // max_candidates as a rank bound, clamped on the host
    const uint32_t lane = lane_id();
    const uint32_t* bases32 = reinterpret_cast<const uint32_t*>(a.bases);
    unsigned long long verified = 0, wbytes = 0;
    uint32_t cols = 0, refuted = 0;  // columns this lane's recurrences advanced; bound mode: candidates it refuted
    // wave-uniform slice of the worklist, claimed 256 items at a time; a lane whose candidate passed
    // (or whose TaxId chain ended) takes the next item, a lane whose candidate failed keeps its read
    // and moves to the next candidate of the same TaxId (index.rs:393)
    uint32_t bnext = 0, bend = 0;
    bool active = false, exhausted = false;
    bool counted = false;  // bound mode: k_sw_diag has counted this candidate (verified, window bytes)
    uint32_t g = 0, o = 0, L = 0, ED = 0, strand = 0;
    bool thr_wrapped = false, hopeless = false;
    uint32_t pv0[W];  // vertical +1 deltas of column 0: the read's rows only
    // list mode: what the lane is doing with candidate g.  0: the listed candidate's edit distance.  A candidate that fails it
    // hands on to its TaxId's next candidate (index.rs:393), which the same lane then decides: 1: that candidate's prefilter
    // by the edit-distance bound (as in bound mode), 2: its edit distance once the bound has passed it (reads with N only:
    // without N the two distances are the same number).  Only a successor the bound leaves undecided goes to another round.
    uint32_t ph = 0, read_n = 0, extra_passed = 0;
    for (;;) {
        unsigned long long need = __ballot(!active);
        if (need) {
            if (bnext == bend && !exhausted) {
                // few, large claims (~11 ns per atomic on one address), but not larger than an even share of a short
                // worklist: a wavefront works through its claim 64 candidates at a time
                const uint32_t n_waves = gridDim.x * (256 / kWave);
                const uint32_t claim = min(256u, max(64u, (n_work / n_waves + 63u) & ~63u));
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.wl_cursor, claim);
                base = __builtin_amdgcn_readfirstlane(base);
                bnext = min(base, n_work);
                bend = min(base + claim, n_work);
                exhausted = bnext == bend;
            }
            const uint32_t take = bnext + __popcll(need & ((1ull << lane) - 1));
            if (!active && take < bend) {
                g = a.worklist[take];
                if (BOUND) {
                    counted = (g & kSweepFlag) != 0;
                    g &= ~kSweepFlag;
                }
                const uint32_t rs = a.cand[g].w;
                const uint32_t r_ = a.r0 + (rs >> 1);
                strand = rs & 1;
                o = a.strand_off[rs];
                const uint32_t b0 = a.read_off[r_];
                L = a.read_off[r_ + 1] - b0;
                ED = (uint32_t)ceil((double)L * a.edit_rate);
                thr_wrapped = 2ull * ED > (uint64_t)L;
                // masks of the FORWARD read; the reverse strand walks the window backwards with
                // complemented symbols instead (edit distance is invariant under reversing both strings)
                const int pad = 32 * W - (int)L;
                uint32_t matchable = 0;
#pragma unroll
                for (int k = 0; k < W; k++) {
                    // rows 32k .. 32k+31 hold read positions q0 .. q0+31 (negative: wildcard rows)
                    const int q0 = 32 * k - pad;
                    uint32_t mA = 0, mC = 0, mG = 0, mT = 0, wild = 0;
                    if (q0 + 31 >= 0) {
                        const long long byte0 = (long long)b0 + q0;  // may point before this read (or the buffer)
                        const uint32_t sh = (uint32_t)(byte0 & 3), dsel = (uint32_t)(byte0 >> 2) & 3u;
                        // the word's 33..36 bytes as three aligned 16-byte loads and a select (see load16): nine dword loads
                        // per word made the set-up two thirds of the kernel's load instructions
                        const long long q16 = byte0 >> 4;
                        const uint4* bases128 = reinterpret_cast<const uint4*>(bases32);
                        const uint4 zero4 = make_uint4(0, 0, 0, 0);
                        const uint4 l0 = q16 >= 0 ? bases128[q16] : zero4, l1 = q16 + 1 >= 0 ? bases128[q16 + 1] : zero4,
                                    l2 = q16 + 2 >= 0 ? bases128[q16 + 2] : zero4;
                        const uint32_t dd[12] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w};
                        uint32_t d[9];
#pragma unroll
                        for (int j = 0; j < 9; j++) d[j] = dsel == 0 ? dd[j] : dsel == 1 ? dd[j + 1] : dsel == 2 ? dd[j + 2] : dd[j + 3];
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const uint32_t four = __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh);
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const int bit_i = 4 * j + q;
                                const uint32_t code = (four >> (8 * q)) & 0xffu;  // already a code (k_normalise)
                                const uint32_t bit = 1u << bit_i;
                                const bool real = q0 + bit_i >= 0;
                                mA |= (real && code == 0) ? bit : 0u;
                                mC |= (real && code == 1) ? bit : 0u;
                                mG |= (real && code == 2) ? bit : 0u;
                                mT |= (real && code == 3) ? bit : 0u;
                                wild |= real ? 0u : bit;
                            }
                        }
                    } else {
                        wild = 0xffffffffu;
                    }
                    matchable += __popc(mA | mC | mG | mT);
                    // reverse strand: the window is walked backwards and row c answers for the complement of c
                    eq_tab[0][k][tid] = (strand ? mT : mA) | wild;
                    eq_tab[1][k][tid] = (strand ? mG : mC) | wild;
                    eq_tab[2][k][tid] = (strand ? mC : mG) | wild;
                    eq_tab[3][k][tid] = (strand ? mA : mT) | wild;
                    eq_tab[4][k][tid] = wild;
                    // every read position that is no base is an N (k_normalise): it matches a window N in the SW matrix only
                    if (NROW == 6) eq_tab[5][k][tid] = ~(mA | mC | mG | mT);
                    pv0[k] = ~wild;
                }
                // edits >= number of read positions that match nothing (N): see k_evaluate
                hopeless = thr_wrapped || (!BOUND && L - matchable > ED);
                read_n = L - matchable;
                ph = 0;
                active = true;
            }
            bnext = min(bnext + (uint32_t)__popcll(need), bend);
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }
        uint32_t verdict = 0, vg = 0;  // bound mode: 1 = passes the prefilter, 2 = undecided (candidate vg)
        if (!LIST && active && hopeless) {  // no sweep needed: every candidate of this strand fails
            const uint4 c = a.cand[g];
            if (!BOUND || !counted) {
                verified++;
                wbytes += c.y - c.x;
            }
            counted = false;
            a.cand_status[g] = 1;
            const uint32_t nxt = a.cand_next[g];
            if (nxt == 0xffffffffu || nxt >= maxc) active = false;
            else g = o + nxt;
        } else if (active) {
            const uint4 c = a.cand[g];
            uint32_t Pv[W], Mv[W];
#pragma unroll
            for (int k = 0; k < W; k++) {
                Pv[k] = pv0[k];
                Mv[k] = 0;
            }
            const uint32_t Wn = c.y - c.x;
            int score = (int)L, best = (int)L;  // D[L][0] = L
            // The window's columns arrive 16 at a time, one fetch ahead of the recurrence (three ahead measured the same and
            // cost eight registers: a wavefront of occupancy here)
            const bool swm = BOUND || (LIST && ph == 1);  // matches of the SW matrix, early exits of the bound
            uint4 nxt4 = Wn ? fetch_cols(ix, c.x, c.y, strand, 0, swm) : make_uint4(0, 0, 0, 0);
            const uint32_t* lane_tab = &eq_tab[0][0][tid];
            uint32_t j0 = 0;
            for (; j0 < Wn; j0 += 16) {
                if (swm) {  // decided already: passes, or the last row cannot come down to 2*ED any more
                    if (BOUND && best <= (int)ED) break;  // (list mode goes on: its minimum is the edit distance of a read without N)
                    if (best > 2 * (int)ED && score - (int)(Wn - j0) > 2 * (int)ED) break;
                }
                const uint4 cur = nxt4;
                if (j0 + 16 < Wn) nxt4 = fetch_cols(ix, c.x, c.y, strand, j0 + 16, swm);  // prefetch the next 16 columns
                const uint32_t lim = min(16u, Wn - j0);
#pragma unroll
                for (int jj = 0; jj < 16; jj++) {
                    if ((uint32_t)jj < lim) {  // predicated, not a break: keeps the 16 columns unrolled
                        const uint32_t word = jj < 4 ? cur.x : jj < 8 ? cur.y : jj < 12 ? cur.z : cur.w;
                        const uint32_t tc = (word >> ((jj & 3) * 8)) & 0xffu;
                        const uint32_t* row = lane_tab + tc * (W * 256);
                        // One word step in 12 vector instructions: the horizontal deltas enter a word as bit 31 of the word
                        // below (v_alignbit_b32 shifts them in), and the delta entering word 0 is 0 (first row all zeros).
                        uint32_t ph_below = 0, mh_below = 0;
#pragma unroll
                        for (int k = 0; k < W; k++) {
                            const uint32_t Eq = row[k * 256];
                            const uint32_t pv = Pv[k], mv = Mv[k];
                            const uint32_t Xv = Eq | mv;
                            const uint32_t Eqh = k ? (Eq | (mh_below >> 31)) : Eq;
                            const uint32_t Xh = __builtin_amdgcn_bitop3_b32((Eqh & pv) + pv, pv, Eqh, 0xBE);  // (sum ^ pv) | Eqh
                            const uint32_t Ph = mv | ~(Xh | pv);
                            const uint32_t Mh = pv & Xh;
                            const uint32_t Phs = k ? __builtin_amdgcn_alignbit(Ph, ph_below, 31) : (Ph << 1);  // (Ph << 1) | carry
                            const uint32_t Mhs = k ? __builtin_amdgcn_alignbit(Mh, mh_below, 31) : (Mh << 1);
                            Pv[k] = Mhs | ~(Xv | Phs);
                            Mv[k] = Phs & Xv;
                            ph_below = Ph;
                            mh_below = Mh;
                        }
                        const uint32_t hp = ph_below >> 31, hm = mh_below >> 31;
                        score += (int)hp - (int)hm;  // the read's last base is bit 31 of the last word
                        best = min(best, score);
                    }
                }
            }
            cols += min(j0, Wn);
            if (LIST ? ph == 1 : (!BOUND || !counted)) {  // (list mode: a successor is counted where its prefilter runs)
                verified++;
                wbytes += Wn;
            }
            counted = false;  // a TaxId's next candidate has not been counted by anyone
            const uint32_t ed = (uint32_t)best;
            if (BOUND) {
                if (!thr_wrapped && ed <= ED) {
                    verdict = 1;
                    vg = g;
                    active = false;
                } else if (!thr_wrapped && ed <= 2 * ED) {
                    verdict = 2;
                    vg = g;
                    active = false;
                } else {  // refuted: index.rs:406 fails
                    refuted++;
                    a.cand_status[g] = 1;
                    const uint32_t nxt = a.cand_next[g];
                    if (nxt == 0xffffffffu || nxt >= maxc) active = false;
                    else g = o + nxt;
                }
            } else if (LIST && ph == 1) {  // a successor's prefilter, decided by the bound
                if (ed > 2 * ED) {  // refuted (index.rs:406)
                    a.cand_status[g] = 1;
                    const uint32_t nxt = a.cand_next[g];
                    if (nxt == 0xffffffffu || nxt >= maxc) active = false;
                    else g = o + nxt;
                } else if (ed > ED) {  // neither bound decides: the sweep of the next round does (counted here: flagged)
                    a.next_list[atomicAdd(a.next_count, 1u)] = g | kSweepFlag;
                    active = false;
                } else {  // passes the prefilter; without N in the read this number is its edit distance too
                    extra_passed++;
                    if (read_n == 0) {
                        const DevBin bin = ix.bins[c.z];
                        a.out[g] = make_uint4(bin.tax_id, bin.gi, c.x >= bin.start ? c.x - bin.start : 0, ed);
                        a.cand_status[g] = 2;
                        active = false;
                    } else {
                        ph = 2;
                    }
                }
            } else {
            const bool pass = !thr_wrapped && ed <= ED;
            if (pass) {
                const DevBin bin = ix.bins[c.z];
                a.out[g] = make_uint4(bin.tax_id, bin.gi, c.x >= bin.start ? c.x - bin.start : 0, ed);
                a.cand_status[g] = 2;
                active = false;
            } else {
                a.cand_status[g] = 1;
                const uint32_t nxt = a.cand_next[g];
                if (nxt == 0xffffffffu || nxt >= maxc) active = false;
                else {
                    g = o + nxt;
                    if (LIST) ph = 1;  // passed the prefilter, failed the edit distance (rare): the TaxId's next candidate, here
                }
            }
            }
        }
        if (BOUND) {  // the wavefront's decisions of this trip: one atomic per list
            const unsigned long long pm = __ballot(verdict == 1), um = __ballot(verdict == 2);
            const unsigned long long lower = (1ull << lane) - 1ull;
            if (pm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(sw_pass_count(a), (uint32_t)__popcll(pm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (verdict == 1) a.pass_list[base + (uint32_t)__popcll(pm & lower)] = vg;
            }
            if (um) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(reinterpret_cast<uint32_t*>(a.counters + a.und_slot), (uint32_t)__popcll(um));
                base = __builtin_amdgcn_readfirstlane(base);
                if (verdict == 2) a.und_list[base + (uint32_t)__popcll(um & lower)] = vg | kSweepFlag;  // counted, bounds tried
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) {
        verified += __shfl_down(verified, d);
        wbytes += __shfl_down(wbytes, d);
    }
    if (lane == 0 && verified) {
        atomicAdd(a.n_verified, verified);
        atomicAdd(a.window_bytes, wbytes);
    }
    if (a.myers_ctr) {  // [0] columns (of W words each) advanced, [1] candidates refuted by the bound
        unsigned long long c64 = cols, r64 = refuted;
        for (int d = 32; d > 0; d >>= 1) {
            c64 += __shfl_down(c64, d);
            r64 += __shfl_down(r64, d);
        }
        unsigned long long e64 = extra_passed;
        for (int d = 32; d > 0; d >>= 1) e64 += __shfl_down(e64, d);
        if (lane == 0) {
            atomicAdd(a.myers_ctr, c64);
            if (BOUND && r64) atomicAdd(a.myers_ctr + 1, r64);
            if (LIST && e64) atomicAdd(a.myers_ctr + 2, e64);
        }
    }
}

// The ordered selection loop of index.rs:384-428 over the verified statuses: a candidate is in
// status "pass" only if every earlier candidate of its TaxId failed, so the duplicate-TaxId skip
// has already been applied; what remains are the two cut-offs and the rank order of the hits.
__global__ __launch_bounds__(256) void k_resolve(uint32_t n_strands, int64_t max_candidates, int64_t max_assignments,
                                                 const uint32_t* __restrict__ strand_off,
                                                 const uint32_t* __restrict__ strand_ncand,
                                                 const uint32_t* __restrict__ cand_status, uint4* __restrict__ out,
                                                 uint32_t* __restrict__ strand_nout) {
    const uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = rs < n_strands;
    uint32_t nc = valid ? strand_ncand[rs] : 0;
    uint32_t nout = 0;
    const uint32_t o = nc ? strand_off[rs] : 0;
    if (max_candidates >= 0 && (uint64_t)nc > (uint64_t)max_candidates) nc = (uint32_t)max_candidates;  // index.rs:385-389
    // a strand of many candidates (repeats: hundreds) is compacted by its whole wavefront, 64 candidates a step: one lane
    // walking it alone was the tail of the kernel, ~0.1 ms in every pass however small
    constexpr uint32_t kCoop = 32;
    const bool big = nc > kCoop;
    if (nc && !big) {
        for (uint32_t i = 0; i < nc; i++) {
            if (cand_status[o + i] == 2) {
                if (nout != i) out[o + nout] = out[o + i];
                nout++;
                if (max_assignments >= 0 && (uint64_t)nout >= (uint64_t)max_assignments) break;  // index.rs:421-425
            }
        }
    }
    const uint32_t lane = lane_id();
    // (the reference checks the cut-off after it has pushed a hit: at least one is kept)
    const uint64_t cap = max_assignments < 0 ? ~0ull : (uint64_t)(max_assignments > 1 ? max_assignments : 1);
    for (unsigned long long m = __ballot(big); m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        const uint32_t o_l = __builtin_amdgcn_readlane(o, l), nc_l = __builtin_amdgcn_readlane(nc, l);
        uint32_t nout_l = 0;
        for (uint32_t base = 0; base < nc_l; base += kWave) {
            const uint32_t i = base + lane;
            const bool pass = i < nc_l && cand_status[o_l + i] == 2;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (pass) v = out[o_l + i];
            const unsigned long long pm = __ballot(pass);
            const uint32_t pos = nout_l + (uint32_t)__popcll(pm & ((1ull << lane) - 1));
            // (pos <= i, and every entry of this step has been read before one is written: the loads above complete first)
            if (pass && (uint64_t)pos < cap) out[o_l + pos] = v;
            nout_l += (uint32_t)__popcll(pm);
            if ((uint64_t)nout_l >= cap) {  // index.rs:421-425
                nout_l = (uint32_t)cap;
                break;
            }
        }
        if ((int)lane == l) nout = nout_l;
    }
    if (!valid) return;
    strand_nout[rs] = nout;
}

// ---------------------------------------------------------------------------------------------
// K5: gather per-strand hits into the final (read, strand, rank) order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(uint32_t n_strands, uint64_t r0, const uint32_t* __restrict__ strand_off,
                                                const uint32_t* __restrict__ strand_nout,
                                                const uint32_t* __restrict__ out_off, const uint4* __restrict__ out,
                                                DevHit* __restrict__ hits, uint64_t hits_base) {
    uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rs >= n_strands) return;
    uint32_t n = strand_nout[rs];
    if (!n) return;
    uint32_t src = strand_off[rs];
    uint64_t dst = hits_base + out_off[rs];
    for (uint32_t i = 0; i < n; i++) {
        uint4 v = out[src + i];
        DevHit h;
        h.read = r0 + (rs >> 1);
        h.tax_id = v.x;
        h.gi = v.y;
        h.edit = v.w;
        h.strand = rs & 1;
        h.offset = v.z;
        hits[dst + i] = h;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch_evaluate(hipStream_t s, const DevIndexView& ix, const EvalArgs& a_, uint64_t max_items, uint32_t max_len) {
    EvalArgs a = a_;
    a.maxc = rank_bound(a.max_candidates);
    // persistent 16-lane groups claim candidates from the worklist
#define EVAL_CASE(RR, WW, GG) hipLaunchKernelGGL((k_evaluate<RR, WW, GG>), dim3(std::max<uint32_t>(1, std::min<uint32_t>(cdiv(max_items, 256 / GG), 256 * 8))), dim3(256), 0, s, ix, a)
    // (8-lane groups, 19 rows per lane, were measured slower for 150-base reads: 21.2 vs 19.2 ms per 1 M reads;
    //  fewer instructions but only 3 wavefronts per SIMD and twice the fetch stalls per wavefront)
    if (max_len <= 32) EVAL_CASE(2, false, 16);
    else if (max_len <= 64) EVAL_CASE(4, false, 16);
    else if (max_len <= 80) EVAL_CASE(5, false, 16);
    else if (max_len <= 112) EVAL_CASE(7, false, 16);
    else if (max_len <= 128) EVAL_CASE(8, false, 16);
    else if (max_len <= 160) EVAL_CASE(10, false, 16);
    else if (max_len <= 208) EVAL_CASE(13, false, 16);
    else if (max_len < 254) EVAL_CASE(16, false, 16);
    else if (max_len <= kMaxRegisterReadLen) EVAL_CASE(16, true, 16);
    else throw std::runtime_error("internal: launch_evaluate called for a read beyond the register-resident kernels");
#undef EVAL_CASE
}

uint32_t tiled_groups(uint64_t max_items, uint32_t strip_len) {
    // at most 1 GiB of strips (8 bytes per window column per group)
    const uint64_t by_mem = std::max<uint64_t>(1, (1ull << 30) / (16ull * 8 * std::max<uint32_t>(strip_len, 1)));
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(cdiv(max_items, 16), 256 * 2), by_mem));
    return blocks * 16;
}

void launch_evaluate_tiled(hipStream_t s, const DevIndexView& ix, const EvalArgs& a_, uint64_t max_items) {
    EvalArgs a = a_;
    a.maxc = rank_bound(a.max_candidates);
    const uint32_t blocks = tiled_groups(max_items, a.strip_len) / 16;
    hipLaunchKernelGGL((k_evaluate<16, true, 16, true>), dim3(blocks), dim3(256), 0, s, ix, a);
}

void launch_sw_pairs(hipStream_t s, const DevIndexView& ix, const EvalArgs& a_, uint64_t max_items_, uint32_t max_len, bool diag, bool top, bool sparse) {
    // sparse: the work list is what the edit-distance bound left undecided -- a fraction of a percent of the seed hits
    // max_items_ counts; a small grid (workgroups cost ~0.1 us each to dispatch, and these hold 40 KB of LDS)
    const uint64_t max_items = sparse ? std::max<uint64_t>(max_items_ / 64, 1024) : max_items_;
    // persistent 16-lane groups, two candidates per group: one resident generation of blocks pulls the worklist
    EvalArgs a = a_;
    a.maxc = rank_bound(a.max_candidates);
    if (top) {  // the top half of the rows (a.und_list / a.und_slot take what they cannot decide)
#define SW_TOP(RR)                                                                                                       \
    do {                                                                                                                 \
        const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(cdiv(max_items, 32), 256 * kSwTopOcc));           \
        a.claim_shift = 0;                                                                                               \
        while ((1ull << a.claim_shift) < (uint64_t)grid * 16 * 2) a.claim_shift++;                                       \
        hipLaunchKernelGGL((k_sw_pairs<RR, false, true>), dim3(grid), dim3(256), 0, s, ix, a);                           \
    } while (0)
        if (max_len <= 64) SW_TOP(2);
        else if (max_len <= 96) SW_TOP(3);
        else if (max_len <= 128) SW_TOP(4);
        else if (max_len <= 160) SW_TOP(5);
        else if (max_len <= 208) SW_TOP(7);
        else SW_TOP(8);
#undef SW_TOP
        return;
    }
#define SW_CASE(RR)                                                                                                      \
    do {                                                                                                                 \
        const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(cdiv(max_items, 32), 256 * ((RR) > 10 ? 3 : kSwOcc))); \
        a.claim_shift = 0;                                                                                               \
        while ((1ull << a.claim_shift) < (uint64_t)grid * 16 * 2) a.claim_shift++;                                       \
        if (diag) hipLaunchKernelGGL((k_sw_pairs<RR, true>), dim3(grid), dim3(256), 0, s, ix, a);                        \
        else hipLaunchKernelGGL((k_sw_pairs<RR, false>), dim3(grid), dim3(256), 0, s, ix, a);                            \
    } while (0)
    if (max_len <= 64) SW_CASE(4);
    else if (max_len <= 96) SW_CASE(6);
    else if (max_len <= 128) SW_CASE(8);
    else if (max_len <= 160) SW_CASE(10);
    else if (max_len <= 208) SW_CASE(13);
    else SW_CASE(16);
#undef SW_CASE
}

void launch_sw_diag(hipStream_t s, const DevIndexView& ix, const EvalArgs& a, uint64_t max_items, uint32_t max_len,
                    uint32_t* sweep_list, uint32_t sweep_slot) {
    // 16 groups per workgroup, a candidate per group and trip; max_items counts seed hits, about four per candidate
    const uint32_t grid = std::max<uint32_t>(16, std::min<uint32_t>(cdiv(max_items, 16 * 32), 256 * 6));
    if (max_len <= 128) hipLaunchKernelGGL(k_sw_diag<2>, dim3(grid), dim3(256), 0, s, ix, a, sweep_list, sweep_slot);
    else if (max_len <= 192) hipLaunchKernelGGL(k_sw_diag<3>, dim3(grid), dim3(256), 0, s, ix, a, sweep_list, sweep_slot);
    else hipLaunchKernelGGL(k_sw_diag<4>, dim3(grid), dim3(256), 0, s, ix, a, sweep_list, sweep_slot);
}

void launch_edit_myers(hipStream_t s, const DevIndexView& ix, const EvalArgs& a_, uint64_t max_items, uint32_t max_len,
                       int mode) {
    EvalArgs a = a_;
    a.maxc = rank_bound(a.max_candidates);
    // max_items bounds the work list from far above (seed hits, not candidates); a grid of one resident generation at
    // most, and smaller when the list is short: workgroups cost ~0.1 us each to dispatch, with work or without, and
    // the lists are claimed 64 candidates at a time whatever the grid
    // (4096, 1024 or 512 hits per workgroup: the same within the noise on passes of 84 k to 1 M reads)
    uint32_t blocks = std::max<uint32_t>(16, std::min<uint32_t>(cdiv(max_items, 4096), 256 * 5));
    uint32_t W = (max_len + 31) / 32;
#define MYERS_CASE(WW)                                                                                   \
    do {                                                                                                 \
        if (mode == MY_LIST) hipLaunchKernelGGL((k_edit_myers<WW, MY_LIST>), dim3(blocks), dim3(256), 0, s, ix, a);        \
        else if (mode == MY_BOUND) hipLaunchKernelGGL((k_edit_myers<WW, MY_BOUND>), dim3(blocks), dim3(256), 0, s, ix, a); \
        else hipLaunchKernelGGL((k_edit_myers<WW, MY_CHAIN>), dim3(blocks), dim3(256), 0, s, ix, a);                      \
    } while (0)
    if (W <= 2) MYERS_CASE(2);
    else if (W <= 3) MYERS_CASE(3);
    else if (W <= 4) MYERS_CASE(4);
    else if (W <= 5) MYERS_CASE(5);
    else if (W <= 6) MYERS_CASE(6);
    else if (W <= 7) MYERS_CASE(7);
    else MYERS_CASE(8);
#undef MYERS_CASE
}

void launch_resolve(hipStream_t s, uint32_t n_strands, int64_t max_candidates, int64_t max_assignments,
                    const uint32_t* strand_off, const uint32_t* strand_ncand, const uint32_t* cand_status, uint4* out,
                    uint32_t* strand_nout) {
    hipLaunchKernelGGL(k_resolve, dim3(cdiv(n_strands, 256)), dim3(256), 0, s, n_strands, max_candidates, max_assignments,
                       strand_off, strand_ncand, cand_status, out, strand_nout);
}

void launch_gather(hipStream_t s, uint32_t n_strands, uint64_t r0, const uint32_t* strand_off, const uint32_t* strand_nout,
                   const uint32_t* out_off, const uint4* out, DevHit* hits, uint64_t hits_base) {
    hipLaunchKernelGGL(k_gather, dim3(cdiv(n_strands, 256)), dim3(256), 0, s, n_strands, r0, strand_off, strand_nout,
                       out_off, out, hits, hits_base);
}

}  // namespace mtsv

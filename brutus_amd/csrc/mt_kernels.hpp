// mt_kernels.hpp -- numpy's legacy random stream on the device
// Part of the single translation unit brutus_kernels.hip (included before
// post_kernels.hpp); everything lives in that unit's anonymous namespace.
//
// The reference draws everything from ONE `numpy.random.RandomState` (MT19937):
//   rstate.normal(size = 3 Nmc Nsel)          utils.py:897   legacy polar Box-Muller
//   rstate.choice(Nsel, size = Ndraws, p = wt)   fitting.py:2039   Ndraws random_sample
//   Ndraws x rstate.choice(Nmc, p = w)           fitting.py:2053   one random_sample each
// per object, in object order.  k_mt_stream reproduces that consumption
// word for word on the device:
//   * MT19937 block recurrence (624 words per block, three dependent phases of up to 227
//     lanes) + tempering;
//   * random_double = ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53;
//   * legacy_gauss: candidates (x1, x2) = 2 u - 1, accepted iff 0 < x1^2 + x2^2 < 1 (no
//     contraction: the decision is bit-exact), f = sqrt(-2 ln r2 / r2); numpy returns
//     f * x2 first and caches f * x1 for the next call; the cached value carries across
//     calls and objects (has_gauss / gauss of the state);
// and writes, per object, its normals (in consumption order) and its uniforms to HBM for
// the array-sourced variants of k_post_mc / k_post_draw.  A stream is inherently
// sequential (the number of words an object consumes depends on its rejections), so ONE
// workgroup walks one stream; streams of different objects (per-object seeds, the
// sharded mode) run in parallel.  The state after the walk is written back in numpy's
// representation (key[624], pos, has_gauss, cached_gaussian) so the caller's RandomState
// continues exactly where the reference's would.
#pragma once

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_STATE_WORDS = 628;      // key[624], pos, has_gauss, gauss (2 words)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the
// outstanding GLOBAL stores of every wave (vmcnt(0)): with one deviate store per lane and
// step that made every step cost an HBM write round trip.  Nothing written to global
// memory is read back inside this kernel.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v, uint32_t far) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// next block of the state: reads `mt`, writes `nx` (no read/write hazard inside a phase, so
// one barrier per phase).  All threads of the workgroup; three dependent phases of <= 227 lanes.
__device__ __forceinline__ void mt_next_block(const uint32_t *mt, uint32_t *nx) {
    const int t = threadIdx.x;
    if (t < 227) nx[t] = mt_twist(mt[t], mt[t + 1], mt[t + MT_M]);                  // kk = 0..226
    lds_barrier();
    if (t < 227) nx[t + 227] = mt_twist(mt[t + 227], mt[t + 228], nx[t]);           // kk = 227..453
    lds_barrier();
    if (t < 169) nx[t + 454] = mt_twist(mt[t + 454], mt[t + 455], nx[t + 227]);     // kk = 454..622
    if (t == 169) nx[623] = mt_twist(mt[623], nx[0], nx[396]);                       // kk = 623
    lds_barrier();
}

constexpr int MT_NT = 1024;      // threads of the stream walker (16 waves: 4 per SIMD hide the
                                 // f64 latency of the polar step; one wave per SIMD did not)
constexpr int MT_NBLK = 4;       // MT blocks generated per refill (2496 words = 624 candidates)
constexpr int MT_WCAP = MT_NBLK * MT_N + 4;

// One workgroup per stream.  seg_obj0[g] .. seg_obj0[g + 1] are the objects stream g
// serves, in order.  Object o consumes nnorm[o] normals (written to Z + zoff[o]) and then
// `nuni` uniforms (written to U + o * nuni).
__global__ void __launch_bounds__(MT_NT)
k_mt_stream(int nseg, const int32_t *__restrict__ seg_obj0, uint32_t *__restrict__ states,
            const int64_t *__restrict__ nnorm, const int64_t *__restrict__ zoff,
            double *__restrict__ Z, int nuni, double *__restrict__ U) {
#pragma clang fp contract(off)
    __shared__ uint32_t blk[MT_NBLK][MT_N];   // the last MT_NBLK raw state blocks, oldest first
    __shared__ uint32_t wbuf[MT_WCAP];        // unread tempered words
    __shared__ int wcnt[MT_NT / 64];
    __shared__ int s_nw, s_rp, s_hasg;
    __shared__ double s_gauss;
    const int g = blockIdx.x;
    if (g >= nseg) return;
    uint32_t *stt = states + (int64_t)g * MT_STATE_WORDS;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int k = t; k < MT_N; k += MT_NT) blk[MT_NBLK - 1][k] = stt[k];
    lds_barrier();
    {
        const int pos = (int)stt[MT_N];
        // unread words of the current block
        for (int k = t; k < MT_N - pos; k += MT_NT) wbuf[k] = mt_temper(blk[MT_NBLK - 1][pos + k]);
        if (t == 0) {
            s_nw = MT_N - pos < 0 ? 0 : MT_N - pos;
            s_rp = 0;
            s_hasg = (int)stt[MT_N + 1];
            s_gauss = __hiloint2double((int)stt[MT_N + 3], (int)stt[MT_N + 2]);
        }
    }
    lds_barrier();
    // make at least `need` (<= 4) unread words available: keep the (< need) left-over words
    // and append MT_NBLK fresh blocks.  The left-overs are consumed first, so after the
    // consumption that triggered a refill every unread word lies in blk[0..MT_NBLK-1].
    auto refill = [&](int need) {
        while (s_nw - s_rp < need) {          // uniform: shared values, barriers below
            const int left = s_nw - s_rp, rp = s_rp;
            uint32_t keep = 0;
            if (t < left) keep = wbuf[rp + t];
            lds_barrier();
            if (t < left) wbuf[t] = keep;
            // block 0 follows the previous last block: stage it through block 1's storage
            mt_next_block(blk[MT_NBLK - 1], blk[1]);
            for (int k = t; k < MT_N; k += MT_NT) blk[0][k] = blk[1][k];
            lds_barrier();
            for (int b = 1; b < MT_NBLK; ++b) mt_next_block(blk[b - 1], blk[b]);
            for (int k = t; k < MT_NBLK * MT_N; k += MT_NT)
                wbuf[left + k] = mt_temper(blk[k / MT_N][k % MT_N]);
            if (t == 0) {
                s_nw = left + MT_NBLK * MT_N;
                s_rp = 0;
            }
            lds_barrier();
        }
    };
    for (int o = seg_obj0[g]; o < seg_obj0[g + 1]; ++o) {
        const int64_t n = nnorm[o];
        double *zo = Z + zoff[o];
        int64_t written = 0;                   // normals of this object placed so far
        if (n > 0 && s_hasg) {                 // the cached deviate comes first
            if (t == 0) zo[0] = s_gauss;
            written = 1;
            lds_barrier();
            if (t == 0) s_hasg = 0;
            lds_barrier();
        }
        while (written < n) {
            refill(4);
            const int navail = (s_nw - s_rp) >> 2;
            const int na = navail < MT_NT ? navail : MT_NT;
            const int rp = s_rp;
            bool acc = false;
            double x1 = 0., x2 = 0., r2 = 1.;
            if (t < na) {
                const uint32_t *w = wbuf + rp + 4 * t;
                const double u1 = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
                const double u2 = ((double)(w[2] >> 5) * 67108864.0 + (double)(w[3] >> 6)) / 9007199254740992.0;
                x1 = 2.0 * u1 - 1.0;
                x2 = 2.0 * u2 - 1.0;
                r2 = x1 * x1 + x2 * x2;
                acc = r2 < 1.0 && r2 != 0.0;
            }
            const unsigned long long bal = __ballot(acc);
            if (lane == 0) wcnt[wv] = __popcll(bal);
            lds_barrier();
            int before = __popcll(bal & ((1ull << lane) - 1ull)), total = 0;
            for (int q = 0; q < MT_NT / 64; ++q) {
                const int c = wcnt[q];
                before += q < wv ? c : 0;
                total += c;
            }
            // accepted pairs still wanted: pair m gives normals written + 2m, written + 2m + 1
            const int64_t want = (n - written + 1) >> 1;
            const int use = (int64_t)total <= want ? total : (int)want;
            if (acc && before < use) {
                // ~1 ulp Newton forms of ln, 1/x and sqrt (fastmath.hpp): the accept / reject
                // decision above is what must be bit-exact; the deviates agree with numpy's
                // to the last bit or two (its own ln is not correctly rounded either)
                const double f = fast_sqrt(-2.0 * fast_log_r(r2) * fast_rcp(r2));
                const int64_t j = written + 2 * (int64_t)before;
                zo[j] = f * x2;                                   // returned first
                if (j + 1 < n) zo[j + 1] = f * x1;                // the cached one
                else {                                            // stays cached for the next call
                    s_gauss = f * x1;
                    s_hasg = 1;
                }
            }
            // attempts consumed: all of them, or up to the attempt holding pair use - 1
            lds_barrier();
            if (use < total) {
                if (acc && before == use - 1) s_rp = rp + 4 * (t + 1);
            } else if (t == 0) {
                s_rp = rp + 4 * na;
            }
            written += 2 * (int64_t)use;
            if (written > n) written = n;
            lds_barrier();
        }
        // the uniforms of the two choice() stages
        double *uo = U + (int64_t)o * nuni;
        int done = 0;
        while (done < nuni) {
            refill(2);
            const int navail = (s_nw - s_rp) >> 1;
            int nu = navail < MT_NT ? navail : MT_NT;
            if (nu > nuni - done) nu = nuni - done;
            const int rp = s_rp;
            if (t < nu) {
                const uint32_t *w = wbuf + rp + 2 * t;
                uo[done + t] = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
            }
            lds_barrier();
            if (t == 0) s_rp = rp + 2 * nu;
            done += nu;
            lds_barrier();
        }
    }
    // state back in numpy's representation: key = the raw block holding the first unread
    // word, pos = its index in that block (pos = 624: the block is used up)
    {
        const int rem = s_nw - s_rp;                    // <= MT_NBLK * 624, all inside blk[][]
        const int first = MT_NBLK * MT_N - rem;         // index of the first unread word
        int bi = first / MT_N, pos = first % MT_N;
        if (bi >= MT_NBLK) {
            bi = MT_NBLK - 1;
            pos = MT_N;
        }
        for (int k = t; k < MT_N; k += MT_NT) stt[k] = blk[bi][k];
        if (t == 0) {
            stt[MT_N] = (uint32_t)pos;
            stt[MT_N + 1] = (uint32_t)s_hasg;
            stt[MT_N + 2] = (uint32_t)__double2loint(s_gauss);
            stt[MT_N + 3] = (uint32_t)__double2hiint(s_gauss);
        }
    }
}


// ===========================================================================
// The same stream walked by MANY workgroups (jump-ahead)
// ===========================================================================
// A stream is cut into sub-streams of MT_J words whose start windows come from the
// jump-ahead polynomials of tools/gen_mt_jump.py (brutus_amd/mt_jump.npz):
//   window m words ahead:  W'[i] = XOR_{j : g_j = 1} X[1 + j + i],  g = x^(m-1) mod phi.
// Everything the fit consumes is a multiple of 4 words when the number of uniforms per
// object is even, so the stream is a grid of 4-word SLOTS from its first unread word: a
// slot is either one polar candidate or two uniforms.  Which it is depends on where the
// earlier objects stopped, i.e. on the rejections before it -- but whether a slot WOULD be
// accepted as a candidate does not.  So:
//   k_mt_bits     every sub-stream in parallel: 1 bit per slot, "accepted if a candidate"
//   k_mt_sbcount + k_mt_sbscan   accepted-count prefix per 4096-slot superblock
//   k_mt_resolve  one workgroup per stream, objects in order: object o starts at slot x_o,
//                 its a_o-th accepted candidate (a_o pairs needed) ends it at y_o (a search
//                 in the prefix + bitmap, no random numbers regenerated), its uniforms take
//                 the next nuni / 2 slots
//   k_mt_emit     every sub-stream in parallel again: candidates -> normals at their final
//                 index, uniform slots -> uniforms
//   k_mt_advance  the generator state after the last consumed word, in numpy's form.
constexpr int64_t MT_J = 624 * 3360;          // words per sub-stream (mt_jump.npz strides[0])
constexpr int MT_L1 = 4;                      // sub-streams per first-level window (strides[1] = 4 J)
constexpr int MT_SB = 4096;                   // slots per superblock
constexpr int MT_JX = 34 * 624;                // words of X a jump generates (>= 19937 + 625)

// Chain c: windows[dst0[c] + k * inner] = window `stride` words ahead of its predecessor
// (the predecessor of k = 0 is windows[src0[c]]), k = 0..counts[c]-1.  One workgroup
// (1024 threads) per chain.
__global__ void __launch_bounds__(MT_NT)
k_mt_jump(const uint32_t *__restrict__ poly, uint32_t *__restrict__ windows,
          const int64_t *__restrict__ src0, const int64_t *__restrict__ dst0, int64_t inner,
          const int32_t *__restrict__ counts) {
    extern __shared__ uint32_t X[];            // MT_JX words, then the set-bit list
    __shared__ int s_nset;
    uint16_t *setpos = reinterpret_cast<uint16_t *>(X + MT_JX);
    const int c = blockIdx.x, t = threadIdx.x;
    const int count = counts[c];
    if (count <= 0) return;
    if (t == 0) s_nset = 0;
    lds_barrier();
    // positions of the set bits of g (unordered: XOR commutes)
    for (int w = t; w < MT_N; w += MT_NT) {
        uint32_t bits = poly[w];
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            setpos[atomicAdd(&s_nset, 1)] = (uint16_t)(w * 32 + b);
        }
    }
    for (int k = t; k < MT_N; k += MT_NT) X[k] = windows[src0[c] * MT_N + k];
    lds_barrier();
    const int nset = s_nset;
    for (int k = 0; k < count; ++k) {
        // X[624 ..] by the recurrence, block by block
        for (int b = 0; b + MT_N < MT_JX; b += MT_N) mt_next_block(X + b, X + b + MT_N);
        // ~10^4 terms per output word; eight independent chains hide the LDS latency
        uint32_t acc = 0;
        if (t < MT_N) {
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
            const uint32_t *Xt = X + 1 + t;
            int q = 0;
            for (; q + 8 <= nset; q += 8) {
                a0 ^= Xt[setpos[q]];
                a1 ^= Xt[setpos[q + 1]];
                a2 ^= Xt[setpos[q + 2]];
                a3 ^= Xt[setpos[q + 3]];
                a4 ^= Xt[setpos[q + 4]];
                a5 ^= Xt[setpos[q + 5]];
                a6 ^= Xt[setpos[q + 6]];
                a7 ^= Xt[setpos[q + 7]];
            }
            for (; q < nset; ++q) a0 ^= Xt[setpos[q]];
            acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
        }
        lds_barrier();
        if (t < MT_N) {
            X[t] = acc;
            windows[(dst0[c] + (int64_t)k * inner) * MT_N + t] = acc;
        }
        lds_barrier();
    }
}

struct MtSub {           // one sub-stream
    int64_t q0, q1;      // slots [q0, q1) of its stream
    int64_t bit0;        // index of slot q0's bit in the bitmap (multiple of 64)
    int32_t stream;      // which stream
    int32_t skip;        // words of its window before slot q0
};

constexpr int MT_PT = 256;                    // threads of the parallel walkers
constexpr int MT_RB = 4;                      // raw state blocks kept (a ring)

// Sequential word source of a sub-stream for the parallel walkers: a ring of the last
// MT_RB raw state blocks in LDS; block b (624 words) sits in blk[b % MT_RB], words are
// tempered when they are consumed.  `rd` = index of the first unread word (counted from
// the window's first word), `gen` = blocks generated so far: both are uniform values
// every thread carries in registers -- no cursor in LDS, no copy of the words into a
// staging buffer, no barrier besides the three of a block step and one in front of it.
struct MtWalk {
    uint32_t (*blk)[MT_N];
    int rd;                 // (a sub-stream is 2.1e6 words: 32-bit index arithmetic)
    int gen;
    __device__ __forceinline__ void init(const uint32_t *__restrict__ window, int skip) {
        for (int k = threadIdx.x; k < MT_N; k += MT_PT) blk[0][k] = window[k];
        rd = skip;
        gen = 1;
        lds_barrier();
    }
    // make `need` (<= 4 * MT_PT) unread words available
    __device__ __forceinline__ void ensure(int need) {
        while (gen * MT_N - rd < need) {
            lds_barrier();      // every wave is done with the block this step overwrites
            mt_next_block(blk[(gen - 1) % MT_RB], blk[gen % MT_RB]);
            ++gen;
        }
    }
    // the four words of slot `k` slots after the cursor (k < MT_PT), tempered
    __device__ __forceinline__ void slot_words(int k, uint32_t (&w)[4]) const {
        const unsigned w0 = (unsigned)(rd + 4 * k);
        unsigned bq = w0 / (unsigned)MT_N;
        int off = (int)(w0 - bq * (unsigned)MT_N);
        int b = (int)(bq % (unsigned)MT_RB);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[q] = mt_temper(blk[b][off]);
            if (++off == MT_N) {
                off = 0;
                b = b + 1 == MT_RB ? 0 : b + 1;
            }
        }
    }
    __device__ __forceinline__ void consume(int nslot) { rd += 4 * nslot; }
};

__device__ __forceinline__ double mt_u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// pass 1: bit of slot q = "x1^2 + x2^2 of its four words is in (0, 1)".
// EMIT: the two normals of every accepted candidate are written as well, compacted per
// sub-stream: the r-th accepted slot of sub-stream g goes to zloc[bit0_g + r] as (f x2, f x1)
// -- numpy's order of use.  Which object a pair belongs to is known only after the prefix
// over the whole stream (k_mt_resolve); the consumers then read the pairs where they lie,
// through the segment lists of k_mt_segments, and the second walk over the stream
// (k_mt_emit) shrinks to the few sub-streams that hold the objects' uniform slots.
template <bool EMIT>
__global__ void __launch_bounds__(MT_PT)
k_mt_bits(int nsub, const MtSub *__restrict__ subs, const uint32_t *__restrict__ windows,
          unsigned long long *__restrict__ bitmap, double2 *__restrict__ zloc) {
#pragma clang fp contract(off)
    __shared__ uint32_t blk[MT_RB][MT_N];
    __shared__ int wcnt[2][MT_PT / 64];
    const int g = blockIdx.x;
    if (g >= nsub) return;
    const MtSub sb = subs[g];
    MtWalk wk{blk, 0, 0};
    wk.init(windows + (int64_t)g * MT_N, sb.skip);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int64_t lbase = 0;          // accepted candidates of this sub-stream so far
    int par = 0;                // wcnt is double-buffered: one barrier per step
    for (int64_t q = sb.q0; q < sb.q1; q += MT_PT) {
        const int ns = (int)((sb.q1 - q) < MT_PT ? (sb.q1 - q) : MT_PT);
        wk.ensure(4 * ns);
        bool acc = false;
        double x1 = 0., x2 = 0., r2 = 1.;
        if (t < ns) {
            uint32_t w[4];
            wk.slot_words(t, w);
            x1 = 2.0 * mt_u53(w[0], w[1]) - 1.0;
            x2 = 2.0 * mt_u53(w[2], w[3]) - 1.0;
            r2 = x1 * x1 + x2 * x2;
            acc = r2 < 1.0 && r2 != 0.0;
        }
        const unsigned long long bal = __ballot(acc);
        if (lane == 0 && (q - sb.q0) + 64 * wv < sb.q1 - sb.q0)      // words of this sub-stream only
            bitmap[((sb.bit0 + (q - sb.q0)) >> 6) + wv] = bal;
        if constexpr (EMIT) {
            if (lane == 0) wcnt[par][wv] = __popcll(bal);
            lds_barrier();
            int bef = __popcll(bal & ((1ull << lane) - 1ull)), tot = 0;
#pragma unroll
            for (int w2 = 0; w2 < MT_PT / 64; ++w2) {
                bef += w2 < wv ? wcnt[par][w2] : 0;
                tot += wcnt[par][w2];
            }
            par ^= 1;
            if (acc) {
                const double f = fast_sqrt(-2.0 * fast_log_r(r2) * fast_rcp(r2));      // as k_mt_emit
                zloc[sb.bit0 + lbase + bef] = make_double2(f * x2, f * x1);
            }
            lbase += tot;
        }
        wk.consume(ns);
    }
}

// accepted candidates per superblock of MT_SB slots (64 bitmap words)
__global__ void k_mt_sbcount(int64_t nsb, const unsigned long long *__restrict__ bitmap,
                             uint32_t *__restrict__ cnt) {
    const int64_t sbi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sbi >= nsb) return;
    int c = __popcll(bitmap[sbi * 64 + (threadIdx.x & 63)]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) cnt[sbi] = (uint32_t)c;
}

// exclusive prefix of cnt over [lo, hi) per stream (one workgroup per stream)
__global__ void __launch_bounds__(1024)
k_mt_sbscan(const int64_t *__restrict__ sb_lo, const uint32_t *__restrict__ cnt,
            int64_t *__restrict__ pre) {
    __shared__ int64_t ws[16];
    __shared__ int64_t carry;
    const int st = blockIdx.x;
    const int64_t lo = sb_lo[st], hi = sb_lo[st + 1];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t b = lo; b < hi; b += 1024) {
        const int64_t i = b + threadIdx.x;
        const int64_t v = i < hi ? cnt[i] : 0;
        int64_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t y = __shfl_up(inc, off, 64);
            if (lane >= off) inc += y;
        }
        if (lane == 63) ws[wv] = inc;
        __syncthreads();
        int64_t base = carry;
        for (int q = 0; q < wv; ++q) base += ws[q];
        if (i < hi) pre[i + st] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) pre[hi + st] = carry;   // one spare entry per stream: the total
}

struct MtObj {           // per object, written by k_mt_resolve
    int64_t x, y;        // candidate slots [x, y), uniform slots [y, y + nuni / 2)
    int64_t px;          // accepted candidates of the stream before slot x
    int32_t c;           // 1: its first normal is the deviate cached by its predecessor
    int32_t nxt;         // next object of the stream that draws normals (-1: none)
    int32_t prv;         // previous one (-1: none; the cached deviate then comes with the state)
    int32_t pad_;
};

// One workgroup (64 threads) per stream, its objects in order.
//   bit_base[st]  bitmap index of the stream's slot 0 (multiple of MT_SB)
//   sb_lo[st]     first superblock of the stream (pre has one spare entry per stream:
//                 stream st's superblock b sits at pre[sb_lo[st] + st + b])
//   tslots[st]    slots that were generated for the stream
__global__ void __launch_bounds__(64)
k_mt_resolve(const int32_t *__restrict__ seg_obj0, const int64_t *__restrict__ nnorm, int nuni,
             const uint32_t *__restrict__ states, const int64_t *__restrict__ bit_base,
             const int64_t *__restrict__ sb_lo, const int64_t *__restrict__ tslots,
             const unsigned long long *__restrict__ bitmap, const int64_t *__restrict__ pre,
             const int64_t *__restrict__ zoff, double *__restrict__ Z, MtObj *__restrict__ objs,
             int64_t *__restrict__ end_slot, int32_t *__restrict__ end_hasg,
             int32_t *__restrict__ end_new, int32_t *__restrict__ fail, double *__restrict__ gauss0_out) {
    const int st = blockIdx.x, lane = threadIdx.x;
    const uint32_t *stt = states + (int64_t)st * MT_STATE_WORDS;
    int c = (int)stt[MT_N + 1];
    const double gauss0 = __hiloint2double((int)stt[MT_N + 3], (int)stt[MT_N + 2]);
    const unsigned long long *bm = bitmap + (bit_base[st] >> 6);
    const int64_t *pr = pre + sb_lo[st] + st;
    const int64_t nsb = sb_lo[st + 1] - sb_lo[st];
    const int64_t T = tslots[st];
    // accepted candidates before slot q
    auto before = [&](int64_t q) -> int64_t {
        const int64_t b = q >> 12;
        const int w = (int)((q & (MT_SB - 1)) >> 6), bit = (int)(q & 63);
        int cnt = 0;
        if (lane < w) cnt = __popcll(bm[b * 64 + lane]);
        else if (lane == w && bit) cnt = __popcll(bm[b * 64 + lane] & ((1ull << bit) - 1ull));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        return (b < nsb ? pr[b] : pr[nsb]) + cnt;
    };
    int64_t x = 0;
    int prev_with_normals = -1;
    bool bad = false, cnew = false;      // cnew: the pending cached deviate was made in this call
    for (int o = seg_obj0[st]; o < seg_obj0[st + 1]; ++o) {
        const int64_t n = nnorm[o];
        MtObj ob;
        ob.x = x;
        ob.c = (n > 0) ? c : 0;
        ob.nxt = -1;
        ob.prv = n > 0 ? prev_with_normals : -1;
        if (n > 0) {
            if (c && lane == 0) {
                // the deviate cached before this call: from the incoming state for the first
                // drawing object, else written by k_mt_emit (the predecessor's last pair)
                if (prev_with_normals < 0 && Z) Z[zoff[o]] = gauss0;     // (Z == nullptr: k_mt_segments)
            }
            if (prev_with_normals >= 0 && lane == 0) objs[prev_with_normals].nxt = o;
            prev_with_normals = o;
        }
        const int64_t a = n > 0 ? (n - c + 1) >> 1 : 0;      // accepted pairs needed
        const int64_t px = before(x);
        ob.px = px;
        int64_t y = x;
        if (a > 0) {
            const int64_t target = px + a;                    // the a-th accepted from x
            // superblock: largest b with pr[b] < target
            int64_t lo = x >> 12, hi = nsb;                   // pr[lo] <= px < target
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (pr[mid] < target) lo = mid; else hi = mid;
            }
            if (lo >= nsb || pr[nsb] < target) {
                bad = true;
                y = T;
            } else {
                const int wantin = (int)(target - pr[lo]);    // 1-based rank inside superblock lo
                const unsigned long long wd = bm[lo * 64 + lane];
                const int pc = __popcll(wd);
                int inc = pc;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int yv = __shfl_up(inc, off, 64);
                    if (lane >= off) inc += yv;
                }
                const bool mine = inc >= wantin && inc - pc < wantin;
                const unsigned long long who = __ballot(mine);
                const int wl = __ffsll((long long)who) - 1;
                int r = wantin - (__shfl(inc, wl, 64) - __shfl(pc, wl, 64));      // rank in the word
                unsigned long long v = __shfl((long long)wd, wl, 64);
                int bitpos = 0;
                for (; r > 1; --r) v &= v - 1;                // drop r - 1 lowest set bits
                bitpos = __ffsll((long long)v) - 1;
                y = lo * MT_SB + wl * 64 + bitpos + 1;
            }
            if ((n - c) & 1) c = 1; else c = 0;
            cnew = c == 1;
        } else if (n > 0) {
            c = 0;                                            // n == 1 served by the cached deviate
            cnew = false;
        }
        ob.y = y;
        x = y + nuni / 2;
        if (x > T) bad = true;
        if (lane == 0) {
            // nxt is patched later by a successor; keep the value a successor may already
            // have written?  (successors run after us in this loop, so plain store is fine)
            objs[o].x = ob.x;
            objs[o].y = ob.y;
            objs[o].px = ob.px;
            objs[o].c = ob.c;
            objs[o].nxt = -1;
            objs[o].prv = ob.prv;
        }
    }
    if (lane == 0) {
        gauss0_out[st] = gauss0;
        end_slot[st] = x;
        end_hasg[st] = c;
        end_new[st] = (c && cnew) ? 1 : 0;
        if (bad) atomicAdd(fail, 1);
    }
}

// pass 2: candidates -> normals at their final index, uniform slots -> uniforms
__global__ void __launch_bounds__(MT_PT)
k_mt_emit(int nsub, const MtSub *__restrict__ subs, const uint32_t *__restrict__ windows,
          const unsigned long long *__restrict__ bitmap, const int64_t *__restrict__ bit_base,
          const int64_t *__restrict__ sb_lo, const int64_t *__restrict__ pre,
          const int32_t *__restrict__ seg_obj0, const MtObj *__restrict__ objs,
          const int64_t *__restrict__ nnorm, const int64_t *__restrict__ zoff,
          double *__restrict__ Z, int nuni, double *__restrict__ U, double *__restrict__ end_gauss,
          int uni_only) {
#pragma clang fp contract(off)
    __shared__ uint32_t blk[MT_RB][MT_N];
    __shared__ int wcnt[2][MT_PT / 64];
    __shared__ int64_t s_bound[2 * BRUTUS_MAX_BATCH + 1];
    const int g = blockIdx.x;
    if (g >= nsub) return;
    const MtSub sb = subs[g];
    const int st = sb.stream;
    const int o0 = seg_obj0[st], no = seg_obj0[st + 1] - o0;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    // region boundaries of the stream: [x_0, y_0, x_1, y_1, ..., end]
    for (int k = t; k < no; k += MT_PT) {
        s_bound[2 * k] = objs[o0 + k].x;
        s_bound[2 * k + 1] = objs[o0 + k].y;
    }
    if (t == 0) s_bound[2 * no] = no > 0 ? objs[o0 + no - 1].y + nuni / 2 : 0;
    MtWalk wk{blk, 0, 0};
    wk.init(windows + (int64_t)g * MT_N, sb.skip);
    if (sb.q0 >= s_bound[2 * no]) return;              // nothing of this sub-stream is consumed
    if (uni_only) {
        // the normals were written by pass 1 (k_mt_bits<true>): only a sub-stream that holds
        // uniform slots (an odd region) is walked again
        int lo = 0, hi = 2 * no;                       // region of slot q0
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_bound[mid] <= sb.q0) lo = mid; else hi = mid;
        }
        const bool any = (lo & 1) || s_bound[lo + 1] < sb.q1;
        if (!any) return;
    }
    // accepted candidates of the stream before slot q0 (q0 is a multiple of 64)
    int64_t pcount;
    {
        const unsigned long long *bm = bitmap + (bit_base[st] >> 6);
        const int64_t b = sb.q0 >> 12;
        const int w = (int)((sb.q0 & (MT_SB - 1)) >> 6);
        int cnt = (t < w) ? __popcll(bm[b * 64 + t]) : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if (lane == 0) wcnt[0][wv] = cnt;
        lds_barrier();
        pcount = pre[sb_lo[st] + st + b] + wcnt[0][0];  // w < 64: only wave 0 holds counts
        lds_barrier();
    }
    int par = 0;        // wcnt is double-buffered: one barrier per step
    const int64_t qend = sb.q1 < s_bound[2 * no] ? sb.q1 : s_bound[2 * no];
    // region of the step's first slot, carried along (regions are millions of slots long:
    // nearly every step lies inside one, and then region, object and its constants are
    // uniform instead of a binary search and three loads per lane)
    int ureg = 0;
    {
        int lo = 0, hi = 2 * no;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_bound[mid] <= sb.q0) lo = mid; else hi = mid;
        }
        ureg = lo;
    }
    for (int64_t q = sb.q0; q < qend; q += MT_PT) {
        const int ns = (int)((sb.q1 - q) < MT_PT ? (sb.q1 - q) : MT_PT);
        wk.ensure(4 * ns);
        while (ureg + 1 < 2 * no && s_bound[ureg + 1] <= q) ++ureg;
        const bool one = q + ns <= s_bound[ureg + 1];        // the whole step in region ureg
        const int64_t myq = q + t;
        bool acc = false, accbit = false;
        double x1 = 0., x2 = 0., r2 = 1., u1 = 0., u2 = 0.;
        int reg = -1;
        if (t < ns) {
            uint32_t w[4];
            wk.slot_words(t, w);
            u1 = mt_u53(w[0], w[1]);
            u2 = mt_u53(w[2], w[3]);
            x1 = 2.0 * u1 - 1.0;
            x2 = 2.0 * u2 - 1.0;
            r2 = x1 * x1 + x2 * x2;
            accbit = r2 < 1.0 && r2 != 0.0;        // the bitmap's bit: every slot counts in the prefix
            if (one) {
                reg = ureg;
            } else if (myq < s_bound[2 * no]) {
                int lo = 0, hi = 2 * no;           // region: largest r with bound[r] <= myq
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_bound[mid] <= myq) lo = mid; else hi = mid;
                }
                reg = lo;
            }
            acc = reg >= 0 && accbit && !(reg & 1) && !uni_only;
        }
        const unsigned long long bal = __ballot(accbit);
        if (lane == 0) wcnt[par][wv] = __popcll(bal);
        lds_barrier();
        int bef = __popcll(bal & ((1ull << lane) - 1ull)), tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < MT_PT / 64; ++w2) {
            bef += w2 < wv ? wcnt[par][w2] : 0;
            tot += wcnt[par][w2];
        }
        par ^= 1;
        if (reg >= 0) {
            const int o = o0 + (reg >> 1);
            if (reg & 1) {
                const int64_t ui = 2 * (myq - s_bound[reg]);
                if (ui < nuni) U[(int64_t)o * nuni + ui] = u1;
                if (ui + 1 < nuni) U[(int64_t)o * nuni + ui + 1] = u2;
            } else if (acc) {
                const MtObj ob = objs[o];            // (uniform address when `one`: scalar loads)
                const int64_t m = (pcount + bef) - ob.px;            // pair index in the object
                const int64_t j = ob.c + 2 * m, n = nnorm[o];
                const double f = fast_sqrt(-2.0 * fast_log_r(r2) * fast_rcp(r2));
                double *zo = Z + zoff[o];
                if (j < n) zo[j] = f * x2;
                if (j + 1 < n) zo[j + 1] = f * x1;
                else if (j < n) {                                     // cached for the next call
                    if (ob.nxt >= 0) Z[zoff[ob.nxt]] = f * x1;
                    else end_gauss[st] = f * x1;
                }
            }
        }
        pcount += tot;
        wk.consume(ns);
    }
}

// ---- the normals where pass 1 left them -------------------------------------------------
// Object o's normals are its cached deviate (if c) followed by the pairs of its accepted
// candidate slots.  Pass 1 stored those per sub-stream; an object therefore reads a short
// list of segments: segment i covers the object's pairs [pair0_i, pair0_{i+1}) at
// zloc[addr_i + (pair - pair0_i)].  Lists of at most (sub-streams it touches) entries, at
// seg_lo[o] = (first sub-stream) + o, which no two objects share.
struct ZMap {
    const double2 *zloc;
    const int64_t *seg_pair0, *seg_addr, *seg_lo;
    const int32_t *nseg;
    const double *cached;
    const int32_t *c;            // 1: normal 0 of the object is cached[o]
};
__device__ __forceinline__ double zmap_at(const ZMap &zm, int o, int64_t j) {
    const int c = zm.c[o];
    if (j < c) return zm.cached[o];
    const int64_t pq = j - c, pr = pq >> 1, lo0 = zm.seg_lo[o];
    int lo = 0, hi = zm.nseg[o];                       // largest i with pair0_i <= pr
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (zm.seg_pair0[lo0 + mid] <= pr) lo = mid; else hi = mid;
    }
    const double2 v = zm.zloc[zm.seg_addr[lo0 + lo] + (pr - zm.seg_pair0[lo0 + lo])];
    return (pq & 1) ? v.y : v.x;
}

// One workgroup (64 threads) per object (after k_mt_resolve).
//   sub_base[st]  index of the stream's first sub-stream, sub_base[st + 1] one past its last
//   gauss0[st]    the deviate cached in the incoming state (k_mt_resolve)
__global__ void __launch_bounds__(64)
k_mt_segments(int nstream, const int32_t *__restrict__ seg_obj0, const int64_t *__restrict__ nnorm,
              const double *__restrict__ gauss0, const MtSub *__restrict__ subs,
              const int64_t *__restrict__ sub_base, const int64_t *__restrict__ bit_base,
              const int64_t *__restrict__ sb_lo, const unsigned long long *__restrict__ bitmap,
              const int64_t *__restrict__ pre, const MtObj *__restrict__ objs,
              const double2 *__restrict__ zloc, int64_t *__restrict__ seg_pair0,
              int64_t *__restrict__ seg_addr, int64_t *__restrict__ seg_lo, int32_t *__restrict__ nseg,
              double *__restrict__ cached, int32_t *__restrict__ cflag) {
    const int o = seg_obj0[0] + blockIdx.x, lane = threadIdx.x;
    if (o >= seg_obj0[nstream]) return;
    int st = 0;                                        // the stream of object o
    {
        int lo = 0, hi = nstream;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seg_obj0[mid] <= o) lo = mid; else hi = mid;
        }
        st = lo;
    }
    const unsigned long long *bm = bitmap + (bit_base[st] >> 6);
    const int64_t *pr = pre + sb_lo[st] + st;
    const int64_t nsb = sb_lo[st + 1] - sb_lo[st];
    const int64_t g0 = sub_base[st], g1 = sub_base[st + 1];
    // accepted candidates of the stream before slot q (wave-cooperative, as in k_mt_resolve)
    auto before = [&](int64_t q) -> int64_t {
        const int64_t b = q >> 12;
        const int w = (int)((q & (MT_SB - 1)) >> 6), bit = (int)(q & 63);
        int cnt = 0;
        if (lane < w) cnt = __popcll(bm[b * 64 + lane]);
        else if (lane == w && bit) cnt = __popcll(bm[b * 64 + lane] & ((1ull << bit) - 1ull));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        return (b < nsb ? pr[b] : pr[nsb]) + cnt;
    };
    // sub-stream of the stream that holds slot q: the last one with q0 <= q
    auto sub_of = [&](int64_t q) -> int64_t {
        int64_t lo = g0, hi = g1;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (subs[mid].q0 <= q) lo = mid; else hi = mid;
        }
        return lo;
    };
    const MtObj ob = objs[o];
    const int64_t n = nnorm[o];
    const int64_t a = n > 0 ? (n - ob.c + 1) >> 1 : 0;       // accepted pairs it consumes
    int ns = 0;
    int64_t gx = g0;
    if (a > 0) {
        gx = sub_of(ob.x);
        for (int64_t g = gx; g < g1 && subs[g].q0 < ob.y; ++g, ++ns) {
            const int64_t sp = before(subs[g].q0);             // accepted before the sub-stream
            if (lane == 0) {
                seg_pair0[gx + o + ns] = g == gx ? 0 : sp - ob.px;
                seg_addr[gx + o + ns] = subs[g].bit0 + (g == gx ? ob.px - sp : 0);
            }
        }
    }
    double cv = 0.;
    if (n > 0 && ob.c) {
        cv = gauss0[st];
        if (ob.prv >= 0) {
            // the second normal of the predecessor's last pair: its last accepted slot is
            // y - 1, rank px + a - 1 of the stream
            const MtObj pb = objs[ob.prv];
            const int64_t pa = (nnorm[ob.prv] - pb.c + 1) >> 1;
            const int64_t g = sub_of(pb.y - 1);
            const int64_t sp = before(subs[g].q0);
            cv = zloc[subs[g].bit0 + (pb.px + pa - 1 - sp)].y;
        }
    }
    if (lane == 0) {
        seg_lo[o] = gx + o;
        nseg[o] = ns;
        cached[o] = cv;
        cflag[o] = n > 0 ? ob.c : 0;
    }
}

// State of a stream `skip` words into the given window, in numpy's form.  When the walk
// leaves a NEW cached deviate behind (gauss_is_new), it is the f * x1 of the candidate slot
// that ends `skipc` words into the window (the stream's last accepted candidate; the
// uniforms of the last object follow it): its four words are met on the way and the value
// is computed here with the expression k_mt_emit uses -- so the state is complete before
// k_mt_emit runs (which lets a caller start the next batch's walk beside it).
__global__ void __launch_bounds__(MT_PT)
k_mt_advance(int nstream, const uint32_t *__restrict__ windows, const int64_t *__restrict__ widx,
             const int64_t *__restrict__ skip, const int64_t *__restrict__ skipc,
             const int32_t *__restrict__ hasg, const int32_t *__restrict__ gauss_is_new,
             double *__restrict__ gauss_new, uint32_t *__restrict__ states) {
#pragma clang fp contract(off)
    __shared__ uint32_t a[MT_N], b[MT_N];
    __shared__ uint32_t s_w[4];
    const int st = blockIdx.x, t = threadIdx.x;
    if (st >= nstream) return;
    uint32_t *cur = a, *nxt = b;
    for (int k = t; k < MT_N; k += MT_PT) cur[k] = windows[widx[st] * MT_N + k];
    lds_barrier();
    int64_t r = skip[st];
    const bool want = gauss_is_new[st] != 0;
    const int64_t c1 = skipc[st];                   // words [c1 - 4, c1) of the window's stream
    int64_t base = 0;                               // index of cur[0]
    bool have_prev = false;
    for (;;) {
        // the candidate's words: the last one lies in this block, the first possibly in the
        // previous one (`nxt` after the swap below)
        if (want && c1 - 1 >= base && c1 - 1 < base + MT_N && t < 4) {
            const int64_t idx = c1 - 4 + t;
            s_w[t] = mt_temper(idx >= base ? cur[idx - base] : (have_prev ? nxt[idx - base + MT_N] : 0u));
        }
        if (r < MT_N) break;
        mt_next_block(cur, nxt);
        uint32_t *sw = cur;
        cur = nxt;
        nxt = sw;
        r -= MT_N;
        base += MT_N;
        have_prev = true;
    }
    uint32_t *stt = states + (int64_t)st * MT_STATE_WORDS;
    double g = __hiloint2double((int)stt[MT_N + 3], (int)stt[MT_N + 2]);
    lds_barrier();
    for (int k = t; k < MT_N; k += MT_PT) stt[k] = cur[k];
    if (t == 0) {
        stt[MT_N] = (uint32_t)r;
        stt[MT_N + 1] = (uint32_t)hasg[st];
        if (want) {
            const double x1 = 2.0 * mt_u53(s_w[0], s_w[1]) - 1.0, x2 = 2.0 * mt_u53(s_w[2], s_w[3]) - 1.0;
            const double r2 = x1 * x1 + x2 * x2;
            const double f = fast_sqrt(-2.0 * fast_log_r(r2) * fast_rcp(r2));
            g = f * x1;
            gauss_new[st] = g;
        }
        stt[MT_N + 2] = (uint32_t)__double2loint(g);
        stt[MT_N + 3] = (uint32_t)__double2hiint(g);
    }
}


// windows[base[st]] = key block of stream st
__global__ void k_mt_keys(int nstream, const uint32_t *__restrict__ states,
                          const int64_t *__restrict__ base, uint32_t *__restrict__ windows) {
    const int st = blockIdx.x;
    if (st >= nstream) return;
    for (int k = threadIdx.x; k < MT_N; k += blockDim.x)
        windows[base[st] * MT_N + k] = states[(int64_t)st * MT_STATE_WORDS + k];
}

}  // namespace

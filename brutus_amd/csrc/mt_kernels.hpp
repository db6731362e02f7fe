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

}  // namespace

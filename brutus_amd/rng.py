"""Counter-based random state for the `lnpost` / resampling stage.

The reference draws from one sequential `numpy.random.RandomState`
(`utils.py:897`, `fitting.py:2039`, `:2053`).  MT19937 plus the legacy polar
Gaussian is inherently serial: the position of every deviate depends on all the
rejections before it.  `PhiloxRandomState` is a drop-in `rstate` object (it
provides the `normal`, `choice` and `random_sample` methods the reference
calls, so the *reference itself* runs unchanged with it) whose j-th normal and
q-th uniform are pure functions of `(seed, j)` / `(seed, q)`:

    Philox4x32-7, key = the 64-bit seed, counter = (index lo, index hi, attempt, stream)
    uniform   u = ((w0 >> 5) * 2**26 + (w1 >> 6)) / 2**53       (numpy's 53-bit recipe)
    normal j  ziggurat with 1024 layers (table `_zigtab`, written by tools/gen_zig_table.py)
              on 64 random bits (a, b):
                  i = b & 1023, sign = bit 10 of b, u = (a * 2**21 + (b >> 11)) / 2**53,
                  x = u * X[i];  x < X[i + 1] -> the normal is +-x          (99.57 %)
              attempt 0 takes (a, b) = words (0, 1) [j even] or (2, 3) [j odd] of the call
              with counter (j >> 1, 0, stream 0) -- two normals per Philox call; a later
              attempt r takes words (0, 1) of the call (j, r, stream 2).  Outside the
              rectangle: layer 0 is the tail x > R (Marsaglia: xt = -ln(1 - u1) / R,
              yt = -ln(1 - u2) with the two uniforms of the call (j, r << 16 | k, stream 3),
              k = 0, 1, ... until 2 yt > xt**2, the normal is +-(R + xt)); any other layer
              is the wedge test Y[i] + u2 (Y[i+1] - Y[i]) < exp(-x**2 / 2) with u2 from
              words (2, 3) of the call (j, r, stream 2), and on failure attempt r + 1.

Because any deviate can be computed without the ones before it, the device can
evaluate the Monte Carlo integral of every selected model of every object in
parallel (`brutus_post_batch`) and still reproduce, deviate for deviate, what
this numpy class -- and therefore the reference run with it -- produces.
This file is the specification of that stream.
"""
import numpy as np

from ._zigtab import X as ZIG_X, Y as ZIG_Y, ZIG_N

__all__ = ["PhiloxRandomState", "philox4x32", "philox_uniform", "philox_normal"]

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
ROUNDS = 7
STREAM_NORMAL, STREAM_UNIFORM, STREAM_RETRY, STREAM_TAIL = 0, 1, 2, 3


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=ROUNDS):
    """Philox4x32 on arrays of 32-bit words held in uint64; returns 4 arrays."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & _MASK for x in (c0, c1, c2, c3))
    k0 = np.uint64(k0) & _MASK
    k1 = np.uint64(k1) & _MASK
    for r in range(rounds):
        if r > 0:
            k0 = (k0 + _W0) & _MASK
            k1 = (k1 + _W1) & _MASK
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
    return c0, c1, c2, c3


def _u53(a, b):
    return ((a >> np.uint64(5)) * 67108864.0 + (b >> np.uint64(6))) / 9007199254740992.0


def philox_uniform(seed, index):
    """Uniform deviates in [0, 1) number `index` (array) of the uniform stream."""
    idx = np.asarray(index, dtype=np.uint64)
    seed = np.uint64(seed)
    o = philox4x32(idx & _MASK, idx >> np.uint64(32), np.zeros_like(idx),
                   np.full_like(idx, STREAM_UNIFORM), seed & _MASK, seed >> np.uint64(32))
    return _u53(o[0], o[1])


def _philox_at(seed, lo64, c2, stream):
    seed = np.uint64(seed)
    return philox4x32(lo64 & _MASK, lo64 >> np.uint64(32), c2, np.full_like(lo64, stream),
                      seed & _MASK, seed >> np.uint64(32))


def _zig_try(a, b):
    """(x, layer, negative, inside the layer's rectangle) from 64 random bits."""
    i = (b & np.uint64(ZIG_N - 1)).astype(np.int64)
    neg = ((b >> np.uint64(10)) & np.uint64(1)).astype(bool)
    u = ((a * np.uint64(2097152) + (b >> np.uint64(11))).astype(np.float64)) / 9007199254740992.0
    x = u * ZIG_X[i]
    return x, i, neg, x < ZIG_X[i + 1]


def philox_normal(seed, index):
    """Standard normal deviates number `index` (array) of the normal stream."""
    idx = np.asarray(index, dtype=np.uint64).ravel()
    out = np.empty(idx.shape, dtype=np.float64)
    if idx.size == 0:
        return out.reshape(np.shape(index))
    odd = (idx & np.uint64(1)).astype(bool)
    o = _philox_at(seed, idx >> np.uint64(1), np.zeros_like(idx), STREAM_NORMAL)
    a, b = np.where(odd, o[2], o[0]), np.where(odd, o[3], o[1])
    todo = np.arange(idx.size)
    attempt = 0
    R = ZIG_X[1]
    while True:
        x, i, neg, fast = _zig_try(a, b)
        out[todo[fast]] = np.where(neg[fast], -x[fast], x[fast])
        rest = ~fast
        if not rest.any():
            break
        todo, x, i, neg = todo[rest], x[rest], i[rest], neg[rest]
        j = idx[todo]
        v = _philox_at(seed, j, np.full_like(j, attempt), STREAM_RETRY)
        done = np.zeros(todo.size, dtype=bool)
        # wedge of layers 1 .. N-1
        wd = i > 0
        u2 = _u53(v[2], v[3])
        with np.errstate(all="ignore"):
            ok = wd & (ZIG_Y[i] + u2 * (ZIG_Y[np.minimum(i + 1, ZIG_N)] - ZIG_Y[i]) < np.exp(-0.5 * x * x))
        out[todo[ok]] = np.where(neg[ok], -x[ok], x[ok])
        done |= ok
        # tail beyond R (layer 0)
        tl = np.nonzero(~wd)[0]
        k = 0
        while tl.size:
            jt = j[tl]
            t = _philox_at(seed, jt, np.full_like(jt, (attempt << 16) | k), STREAM_TAIL)
            xt = -np.log(1.0 - _u53(t[0], t[1])) / R
            yt = -np.log(1.0 - _u53(t[2], t[3]))
            acc = yt + yt > xt * xt
            hit = tl[acc]
            out[todo[hit]] = np.where(neg[hit], -(R + xt[acc]), R + xt[acc])
            done[hit] = True
            tl = tl[~acc]
            k += 1
        if done.all():
            break
        todo = todo[~done]
        attempt += 1
        j = idx[todo]
        v = _philox_at(seed, j, np.full_like(j, attempt), STREAM_RETRY)
        a, b = v[0], v[1]
    return out.reshape(np.shape(index))


class PhiloxRandomState(object):
    """`rstate` object with the call surface the brutus path uses.

    State is two positions (normals consumed, uniforms consumed); `seed` keys
    the generator.  `normal` consumes `size` normals, `choice` consumes one
    uniform per drawn sample, exactly in call order -- so a run is reproducible
    and the consumption of a call can be predicted from its arguments alone.
    """

    def __init__(self, seed=0, n_normal=0, n_uniform=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.n_normal = int(n_normal)
        self.n_uniform = int(n_uniform)

    def normal(self, loc=0.0, scale=1.0, size=None):
        n = 1 if size is None else int(np.prod(size))
        z = philox_normal(self.seed, np.arange(self.n_normal, self.n_normal + n,
                                               dtype=np.uint64))
        self.n_normal += n
        z = loc + scale * z
        return float(z[0]) if size is None else z.reshape(size)

    def random_sample(self, size=None):
        n = 1 if size is None else int(np.prod(size))
        u = philox_uniform(self.seed, np.arange(self.n_uniform, self.n_uniform + n,
                                                dtype=np.uint64))
        self.n_uniform += n
        return float(u[0]) if size is None else u.reshape(size)

    def choice(self, a, size=None, p=None):
        """Legacy `RandomState.choice` semantics for an integer `a`, with
        replacement: `searchsorted(cumsum(p) / sum, uniforms, side='right')`."""
        a = int(a)
        if p is None:
            u = self.random_sample(size)
            return np.minimum((np.asarray(u) * a).astype(np.int64), a - 1) \
                if size is not None else min(int(u * a), a - 1)
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        u = self.random_sample(size)
        idx = np.searchsorted(cdf, u, side='right')
        return np.minimum(idx, a - 1) if size is not None else int(min(idx, a - 1))

    def multivariate_normal(self, mean, cov, size=None):
        mean = np.asarray(mean, dtype=np.float64)
        n = 1 if size is None else int(np.prod(size))
        L = np.linalg.cholesky(np.asarray(cov, dtype=np.float64))
        z = self.normal(size=(n, mean.size))
        out = mean + z @ L.T
        return out[0] if size is None else out.reshape(tuple(np.atleast_1d(size)) + (mean.size,))


# ---------------------------------------------------------------------------
# numpy's legacy generator <-> the 628-word state block of brutus_post_batch_numpy
# ---------------------------------------------------------------------------
MT_STATE_WORDS = 628


def numpy_stream(rstate):
    """The object whose `get_state` / `set_state` address the legacy MT19937 stream the
    reference would draw from: a `numpy.random.RandomState`, or the `numpy.random`
    module itself (`rstate=None`, reference fitting.py:937-944).  None for anything
    else (other bit generators, user classes)."""
    if rstate is np.random or isinstance(rstate, np.random.RandomState):
        try:
            if rstate.get_state()[0] == 'MT19937':
                return rstate
        except Exception:      # pragma: no cover
            return None
    return None


def state_to_words(state):
    """`RandomState.get_state()` -> uint32[628]: key[624], pos, has_gauss, cached_gaussian."""
    name, key, pos, has_gauss, gauss = state[:5]
    if name != 'MT19937':
        raise ValueError("not a legacy MT19937 state")
    w = np.empty(MT_STATE_WORDS, dtype=np.uint32)
    w[:624] = key
    w[624] = pos
    w[625] = has_gauss
    w[626:628] = np.frombuffer(np.float64(gauss).tobytes(), dtype=np.uint32)
    return w


def words_to_state(w):
    w = np.ascontiguousarray(w, dtype=np.uint32)
    gauss = float(np.frombuffer(w[626:628].tobytes(), dtype=np.float64)[0])
    return ('MT19937', w[:624].copy(), int(w[624]), int(w[625]), gauss)

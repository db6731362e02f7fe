#!/usr/bin/env python
"""Jump-ahead polynomials of MT19937 for brutus_amd/csrc/mt_kernels.hpp.

MT19937's state transition F (one 32-bit word per step) is linear over GF(2) with an
irreducible characteristic polynomial phi of degree 19937 on the 19937 bits that matter.
The word sequence X of a state satisfies  sum_j phi_j X[j + i] = 0  for i >= 1 (the low
31 bits of X[0] are not part of the state: they never feed the recurrence, so the
relation only holds once they have left the window).  Hence for g = x^(m-1) mod phi the
window that lies m words ahead is
    W'[i] = XOR_{j : g_j = 1} X[1 + j + i],   i = 0..623
(Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008, "Efficient jump ahead for
F2-linear random number generators").  This script
  1. finds phi with Berlekamp-Massey on one output bit of the generator (numpy's own
     MT19937 supplies the sequence),
  2. computes g = x^(m-1) mod phi for the strides m the device code uses,
  3. checks each g against numpy by brute force (advance a RandomState by m words),
and writes brutus_amd/mt_jump.npz: `strides` (words) and `polys` (uint32 (n, 624), bit j of
the polynomial = bit j % 32 of word j // 32).

Pure Python integers as GF(2)[x] polynomials; runs in about a minute.
"""
import os
import sys

import numpy as np

N = 19937
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def berlekamp_massey_gf2(bits):
    """Minimal polynomial (as int, bit k = coefficient of x^k) of a GF(2) sequence."""
    n = len(bits)
    s = 0
    for i, b in enumerate(bits):
        if b:
            s |= 1 << i
    C, B = 1, 1          # connection polynomials, bit k = c_k (c_0 = 1)
    L, m = 0, 1
    for i in range(n):
        # discrepancy d = s_i + sum_{k=1..L} c_k s_{i-k} = parity(C & reversed window)
        # window w_k = s_{i-k}: take bits i-L..i of s, reversed.  Use the identity on the
        # un-reversed form: sum_k c_k s_{i-k} = parity over k of (C >> k & 1) * (s >> (i-k) & 1)
        # -> evaluate with the reversed connection polynomial instead
        d = 0
        # rev(C) aligned so that bit (L - k) holds c_k; window bits i-L..i of s
        win = (s >> (i - L)) & ((1 << (L + 1)) - 1) if i >= L else None
        if win is None:
            for k in range(0, L + 1):
                if (C >> k) & 1 and i - k >= 0 and (s >> (i - k)) & 1:
                    d ^= 1
        else:
            d = bin(win & REV(C, L)).count("1") & 1
        if d:
            T = C
            C ^= B << m
            if 2 * L <= i:
                L, B, m = i + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


_rev_cache = {}


def REV(C, L):
    """bit-reverse C within L+1 bits (cached per (C, L) identity change is rare)"""
    key = (C, L)
    r = _rev_cache.get(key)
    if r is None:
        _rev_cache.clear()
        r = int(bin(C)[2:].zfill(L + 1)[::-1][: L + 1], 2) if C.bit_length() <= L + 1 else 0
        # bin(C) MSB-first of length L+1; reversing gives bit (L-k) = c_k
        r = int(bin(C)[2:].zfill(L + 1)[::-1], 2)
        r = int(format(C, "0%db" % (L + 1)), 2)          # = C itself, MSB-first string
        # we need bit (L - k) = c_k: that is the bit-reversal of C over L+1 bits
        r = int(format(C, "0%db" % (L + 1))[::-1], 2)
        _rev_cache[key] = r
    return r


def polymod(a, phi, deg):
    """a mod phi over GF(2) (deg = degree of phi)."""
    while a.bit_length() - 1 >= deg:
        a ^= phi << (a.bit_length() - 1 - deg)
    return a


_SQ = [int("".join(c + "0" for c in format(b, "08b")), 2) >> 1 for b in range(256)]


def polysquare(a):
    """square in GF(2)[x]: interleave zeros"""
    out = 0
    by = a.to_bytes((a.bit_length() + 7) // 8 or 1, "little")
    res = bytearray(2 * len(by))
    for i, b in enumerate(by):
        v = _SQ[b]
        res[2 * i] = v & 0xFF
        res[2 * i + 1] = v >> 8
    return int.from_bytes(bytes(res), "little")


def polymulx(a, phi, deg):
    a <<= 1
    if (a >> deg) & 1:
        a ^= phi
    return a


def xpow_mod(m, phi, deg):
    """x^m mod phi by left-to-right square and multiply-by-x"""
    r = 1
    for bit in bin(m)[2:]:
        r = polymod(polysquare(r), phi, deg)
        if bit == "1":
            r = polymulx(r, phi, deg)
    return r


def raw_words(state_key, count):
    """`count` successive RAW (untempered) state words X[0..count) starting from the key
    block `state_key` (X[0..623] = key), by the plain recurrence."""
    x = np.empty(count, dtype=np.uint32)
    x[:624] = state_key
    U, Lm, A = np.uint32(0x80000000), np.uint32(0x7fffffff), np.uint32(0x9908b0df)
    for k in range(0, count - 624):
        y = (x[k] & U) | (x[k + 1] & Lm)
        x[k + 624] = x[k + 397] ^ (y >> np.uint32(1)) ^ (A if (y & np.uint32(1)) else np.uint32(0))
    return x


def jump_window(key, g):
    """window m words ahead of `key` for g = x^(m-1) mod phi"""
    X = raw_words(key, N + 625)
    out = np.zeros(624, dtype=np.uint32)
    j = 0
    gg = g
    while gg:
        if gg & 1:
            out ^= X[1 + j:1 + j + 624]
        gg >>= 1
        j += 1
    return out


def main():
    # 1. minimal polynomial from 2 * 19937 bits of one output bit
    rs = np.random.RandomState(5489)
    nb = 2 * N + 64
    words = np.frombuffer(rs.bytes(4 * nb), dtype=np.uint32)
    bits = (words & 1).astype(np.uint8).tolist()
    phi, L = berlekamp_massey_gf2(bits)
    assert L == N, L
    # BM returns the connection polynomial c(x) = sum c_k x^k with s_i = sum c_k s_{i-k};
    # the characteristic polynomial of the shift is its reciprocal
    phi = int(format(phi, "0%db" % (N + 1))[::-1], 2)
    assert phi.bit_length() - 1 == N and (phi & 1)
    print("phi: degree %d, weight %d" % (N, bin(phi).count("1")))
    J = 624 * 3360                      # 2 096 640 words per sub-stream
    # first-level strides L1 J * 2^r (L1 = MT_L1 of mt_kernels.hpp): the first-level windows
    # of a long stream come from a doubling tree (depth log2 instead of a sequential chain),
    # the L1 - 1 sub-streams behind each of them from a chain of single jumps
    L1 = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    strides = [J] + [L1 * J * 2 ** r for r in range(14)]
    polys = []
    for m in strides:
        g = xpow_mod(m - 1, phi, N)
        polys.append(g)
        print("stride %d: weight %d" % (m, bin(g).count("1")))
    # 3. checks: unaligned shifts against the plain recurrence, the strides against numpy
    key = np.random.RandomState(12345).get_state()[1].copy()   # pos = 624: X[0..623] = key,
    X = raw_words(key, 60000)                                  # first output = temper(X[624])
    for m in (1, 625, 1000, 50001):
        assert np.array_equal(jump_window(key, xpow_mod(m - 1, phi, N)), X[m:m + 624]), m
    for m, g in zip(strides[:2], polys[:2]):
        a = np.random.RandomState(12345)
        a.bytes(4 * m)                          # consume m words (m is a multiple of 624)
        k2 = a.get_state()
        assert k2[2] == 624
        assert np.array_equal(jump_window(key, g), k2[1]), "jump by %d words disagrees with numpy" % m
        print("jump by %d words == numpy" % m)
    # the longer strides by composition: two jumps of m == one jump of 2 m
    for r in range(2, len(strides)):
        half = jump_window(jump_window(key, polys[r - 1]), polys[r - 1])
        assert np.array_equal(jump_window(key, polys[r]), half), strides[r]
        print("jump by %d words == 2 x jump by %d" % (strides[r], strides[r - 1]))
    out = np.zeros((len(polys), 624), dtype=np.uint32)
    for r, g in enumerate(polys):
        by = g.to_bytes(624 * 4, "little")
        out[r] = np.frombuffer(by, dtype="<u4")
    path = os.path.join(ROOT, "brutus_amd", "mt_jump.npz")
    np.savez_compressed(path, strides=np.array(strides, dtype=np.int64), polys=out)
    print("wrote", path)


if __name__ == "__main__":
    main()

"""The algebra behind ringsnark_amd/csrc/witness_inc.hpp, restated in plain Python and checked against schoolbook products (CPU,
no GPU, no library): a cyclic convolution of length 2^n over a prime whose 2-adicity a is SMALLER than n, by running the first a
stages of the decimation-in-frequency transform (twiddle table indexed by decimation-tree node, exactly the device's `tw`),
multiplying the leaves -- G = 2^(n-a) consecutive words each -- as polynomials modulo x^G - eta_g with
eta_g = +-tw[parent of the leaf], and starting the inverse at stage n - a.  The primes are the headline's own ring primes
(default_double_batching_modulus(8192, 8192), seal/seal_util.hpp:20-32: 2-adicity 15, 15, 14, 14)."""
import random

import pytest

from ringsnark_amd import params as P


def _bitrev(i, l):
    r = 0
    for _ in range(l):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def _table(p, logmax):
    """tw[M + i] = w_{2M}^{bitrev(i)}: the device's cyclic table (witness.hip make_cyc) for transforms of up to 2^logmax points"""
    n = 1 << logmax
    g = 2
    while True:
        w = pow(g, (p - 1) // n, p)
        if pow(w, n // 2, p) != 1:
            break
        g += 1
    tw = [1] * n
    for lg in range(logmax):
        M = 1 << lg
        w2 = pow(w, n // (2 * M), p)
        for i in range(M):
            tw[M + i] = pow(w2, _bitrev(i, lg), p)
    return tw


def _fwd(a, logn, tw, p, nst):
    m, gap = 1, (1 << logn) >> 1
    for _ in range(nst):
        for i in range(m):
            W = tw[m + i]
            for j in range(2 * i * gap, 2 * i * gap + gap):
                u, v = a[j], a[j + gap] * W % p
                a[j], a[j + gap] = (u + v) % p, (u - v) % p
        m, gap = m << 1, gap >> 1


def _inv(a, logn, tw, p, u0):
    n = 1 << logn
    gap, m = 1 << u0, n >> (u0 + 1)
    while m >= 1:
        for i in range(m):
            W = pow(tw[m + i], p - 2, p)
            for j in range(2 * i * gap, 2 * i * gap + gap):
                u, v = a[j], a[j + gap]
                a[j], a[j + gap] = (u + v) % p, (u - v) * W % p
        m, gap = m >> 1, gap << 1
    s = pow((n >> u0) % p, p - 2, p)
    for i in range(n):
        a[i] = a[i] * s % p


@pytest.mark.parametrize("limb,logn,inc", [(0, 6, 1), (1, 7, 2), (2, 8, 3), (3, 9, 4), (2, 5, 0)])
def test_incomplete_transform_convolution_equals_schoolbook(limb, logn, inc):
    p = P.preset("C3").q[limb]
    nst, n, G = logn - inc, 1 << logn, 1 << inc
    tw = _table(p, max(1, nst))  # a table of 2^nst entries is all the transform touches: "2-adicity nst"
    rng = random.Random(7 * logn + inc)
    A = [rng.randrange(p) for _ in range(n)]
    B = [rng.randrange(p) for _ in range(n)]
    ref = [0] * n
    for i in range(n):
        for j in range(n):
            ref[(i + j) % n] = (ref[(i + j) % n] + A[i] * B[j]) % p
    fa, fb = A[:], B[:]
    _fwd(fa, logn, tw, p, nst)
    _fwd(fb, logn, tw, p, nst)
    out = [0] * n
    for g in range(n >> inc):
        node = (1 << nst) + g
        eta = 1 if nst == 0 else (tw[node >> 1] if g % 2 == 0 else (p - tw[node >> 1]) % p)
        x, t = fa[g * G:(g + 1) * G], fb[g * G:(g + 1) * G]
        for k in range(G):
            lo = sum(x[i] * t[k - i] for i in range(k + 1))
            hi = sum(x[i] * t[k + G - i] for i in range(k + 1, G))
            out[g * G + k] = (lo + eta * hi) % p
    _inv(out, logn, tw, p, inc)
    assert out == ref
    # the leaf's eta squares to its parent's: the parent's twiddle IS a square root of it, no deeper root is needed
    if nst >= 2:
        for g in range(0, n >> inc, 2):
            node = (1 << nst) + g
            parent_eta = tw[node >> 2] if (node >> 1) % 2 == 0 else (p - tw[node >> 2]) % p
            assert tw[node >> 1] * tw[node >> 1] % p == parent_eta

"""Parameter contract for the ring backend.

The reference builds its parameters from SEAL helpers:
  * ring primes   -- default_double_batching_modulus(N, N_inner) = CoeffModulus::Create(N_inner,
                     bit sizes of BFVDefault(N))            (seal/seal_util.hpp:20-32)
  * encoding ctx  -- EncodingElem::set_context(N_inner): BGV, degree N_inner,
                     coeff_modulus = BFVDefault(N_inner), plain_modulus = q_i
                                                            (seal/seal_ring.hpp:266-306)
SEAL is not available here, so the prime search is restated (SEAL util::get_primes /
CoeffModulus::Create).  It reproduces the BFVDefault(4096) primes written down in the
reference's docs/qrp.sage:3-5 (0xffffee001, 0xffffc4001, 0x1ffffe0001) -- tests/test_oracle.py:12-15.

Constraints enforced here:
  q_i = 1 (mod 2*N_enc)  (batching for the encoding contexts, seal_ring.hpp:297)
  Q_j = 1 (mod 2*N_enc), gcd(q_i, Q_j) = 1
"""
from dataclasses import dataclass, field
from typing import List

# BFVDefault(N) bit sizes (SEAL util/globals.cpp default_coeff_modulus_128), first data level =
# all but the last ("special") prime when there is more than one.
BFV_DEFAULT_BITS = {
    1024: [27],
    2048: [54],
    4096: [36, 36, 37],
    8192: [43, 43, 44, 44, 44],
    16384: [48, 48, 48, 49, 49, 49, 49, 49, 49],
}


def is_prime(n: int) -> bool:
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n == p:
            return True
        if n % p == 0:
            return False
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def get_primes(factor: int, bit_size: int, count: int, exclude=()) -> List[int]:
    """SEAL util::get_primes: primes = 1 (mod factor) of exactly bit_size bits, largest first."""
    value = ((1 << bit_size) - 1) // factor * factor + 1
    lower = 1 << (bit_size - 1)
    out = []
    while len(out) < count and value > lower:
        if value not in exclude and is_prime(value):
            out.append(value)
        value -= factor
    if len(out) != count:
        raise ValueError("not enough %d-bit primes = 1 mod %d" % (bit_size, factor))
    return out


def coeff_modulus_create(factor: int, bit_sizes: List[int], exclude=()) -> List[int]:
    """SEAL CoeffModulus::Create: the later-found (smaller) prime of a size goes first."""
    need = {}
    for b in bit_sizes:
        need[b] = need.get(b, 0) + 1
    table = {b: get_primes(factor, b, c, exclude) for b, c in need.items()}
    return [table[b].pop() for b in bit_sizes]


def two_adicity(q: int) -> int:
    v, x = 0, q - 1
    while x % 2 == 0:
        x //= 2
        v += 1
    return v


@dataclass
class RingParams:
    """Ring Z_q[X]/(X^N+1) with q = prod q_i, and the L encoding contexts (N_enc, Q_j)."""

    N: int
    q: List[int]
    N_enc: int
    Q: List[int]
    name: str = "custom"
    notes: str = ""
    extra: dict = field(default_factory=dict)

    @property
    def L(self):
        return len(self.q)

    @property
    def K(self):
        return len(self.Q)

    @property
    def ring_words(self):
        return self.L * self.N

    @property
    def ct_words(self):
        return 2 * self.K * self.N_enc

    @property
    def enc_words(self):
        return self.L * self.ct_words

    def max_constraints_fast(self) -> int:
        """Largest m whose witness map runs on full-length transforms: complete ones need a cyclic NTT of length
        2*next_pow2(m) in every F_{q_i}; incomplete ones (csrc/witness_inc.hpp) may stop up to four stages short
        (multi-pass sizes, m > 2^14); beyond that the block convolutions take over (to 2^20)."""
        v = min(two_adicity(p) for p in self.q)
        return min(1 << 22, 1 << (v - 1 + (4 if v >= 14 else 0)))

    def validate(self):
        for p in self.q + self.Q:
            assert is_prime(p), hex(p)
            assert (p - 1) % (2 * self.N_enc) == 0, "prime %x != 1 mod 2*N_enc" % p
        assert len(set(self.q + self.Q)) == self.L + self.K, "q_i and Q_j must be distinct"
        assert self.N <= self.N_enc and self.N_enc & (self.N_enc - 1) == 0
        return self


def make_params(N, ring_bits, N_enc, enc_bits, ring_factor=None, name="custom", notes=""):
    """ring_factor: primes q_i = 1 mod ring_factor (default 2*N_enc, the reference's recipe;
    larger powers of two raise the 2-adicity so the fast witness map covers more constraints)."""
    rf = ring_factor or 2 * N_enc
    assert rf % (2 * N_enc) == 0
    q = coeff_modulus_create(rf, ring_bits)
    Q = coeff_modulus_create(2 * N_enc, enc_bits, exclude=set(q))
    return RingParams(N, q, N_enc, Q, name=name, notes=notes).validate()


def preset(name: str) -> RingParams:
    """Named configurations (SURVEY.md section 8(d) "Concrete configs")."""
    if name == "C1":  # bare NTT, N=4096, q=0xffffee001 (bench_ntt plumbing case)
        return RingParams(4096, [0xFFFFEE001], 4096, [0xFFFFC4001], name="C1").validate()
    if name == "C2":  # ringGroth16 m=2^10, N=4096 L=2, N_enc=8192 K=4
        return make_params(4096, BFV_DEFAULT_BITS[4096][:-1], 8192, BFV_DEFAULT_BITS[8192][:-1], name="C2",
                           notes="default_double_batching_modulus(4096, 8192) + BFVDefault(8192) data primes")
    if name in ("C3", "C3R"):  # the headline: N=8192 L=4, N_enc=8192 K=4, ring primes by the reference's recipe
        # default_double_batching_modulus(8192, 8192) = BFVDefault(8192) at first data level: q_i = 1 mod 2*N_enc = 2^14 only
        # (2-adicity 15, 15, 14, 14) -- what every shipped ringSNARK binary and a SEAL-produced key have (seal_util.hpp:20-32,
        # examples/example_SEAL.cpp:15-22).  The witness map's long transforms are incomplete on them (csrc/witness_inc.hpp).
        # "C3R" (rounds 3-5, when the headline ran on the 2^20-adic primes of C3F) is the same preset.
        return make_params(8192, BFV_DEFAULT_BITS[8192][:-1], 8192, BFV_DEFAULT_BITS[8192][:-1], name="C3",
                           notes="default_double_batching_modulus(8192, 8192): what a SEAL-produced headline key has")
    if name == "C3F":  # the headline shape on "friendly" ring primes = 1 mod 2^20 (a custom coeff_modulus, valid in the reference): complete transforms
        return make_params(8192, BFV_DEFAULT_BITS[8192][:-1], 8192, BFV_DEFAULT_BITS[8192][:-1],
                           ring_factor=1 << 20, name="C3F",
                           notes="ring primes = 1 mod 2^20: every transform of the witness map up to 2^19 constraints is complete")
    if name in ("C4", "C4R"):  # configs[3]: Rinocchio N=16384 L=6 (a 6-prime chain of BFVDefault(16384)'s bit sizes), N_enc=16384 K=8; recipe primes
        # q_i = 1 mod 2*N_enc = 2^15 only (2-adicity 20, 17, 15, 15, 16, 15), as default_double_batching_modulus yields them
        return make_params(16384, [48, 48, 48, 49, 49, 49], 16384, BFV_DEFAULT_BITS[16384][:-1], name="C4",
                           notes="configs[3]'s shape with ring primes as the default_double_batching_modulus recipe yields them (seal_util.hpp:20-32)")
    if name == "C4F":  # configs[3]'s shape on ring primes = 1 mod 2^20: complete transforms to 2^19 constraints
        return make_params(16384, [48, 48, 48, 49, 49, 49], 16384, BFV_DEFAULT_BITS[16384][:-1],
                           ring_factor=1 << 20, name="C4F")
    if name == "C5":  # logistic regression as in the reference file: N=2048 L=1, N_enc=16384 K=8
        return make_params(2048, BFV_DEFAULT_BITS[2048], 16384, BFV_DEFAULT_BITS[16384][:-1], name="C5")
    if name == "C5s":  # C5's shape with a 49-bit ring prime (runs on the FP64 arithmetic; C5 itself runs on the integer one)
        return make_params(2048, [49], 16384, BFV_DEFAULT_BITS[16384][:-1], name="C5s",
                           notes="C5 with the 54-bit BFVDefault(2048) prime replaced by a 49-bit one")
    if name == "toy54":  # C5's moduli at test scale: a 54-bit ring prime (integer Montgomery path), 48/49-bit encoding primes
        return make_params(32, [54], 64, [48, 49, 49], ring_factor=1 << 12, name="toy54")
    if name == "toy60":  # the {59,60,60}-bit primes of the reference's microbench.cpp:35-36, everywhere
        return make_params(32, [59, 60], 64, [60, 60, 59], ring_factor=1 << 12, name="toy60")
    if name == "micro60":  # microbench.cpp:33-36: N = 16384, coefficient primes of {59, 60, 60} bits
        return make_params(16384, [59], 16384, [60, 60], name="micro60")
    if name == "toyR":  # the reference's recipe verbatim: ring primes only = 1 mod 2*N_enc (seal_util.hpp:20-32), no extra 2-adicity
        return make_params(32, [30, 30], 64, [40, 40, 41], name="toyR")
    if name == "toyC3":  # C3's moduli (the recipe primes of the headline: 2-adicity 15, 15, 14, 14) on a 32-slot ring: large-m witness maps
        c3 = preset("C3")
        return RingParams(32, list(c3.q), 64, list(c3.Q[:3]), name="toyC3",
                          notes="default_double_batching_modulus(8192, 8192) primes on a small ring").validate()
    if name == "toy":  # CPU-test scale
        return make_params(32, [30, 30], 64, [40, 40, 41], ring_factor=1 << 12, name="toy")
    if name == "toy44x":  # the same with primes = 1 mod 2^23: full-length transforms to 2^22 constraints
        return make_params(32, [43, 44], 64, [43, 44, 44], ring_factor=1 << 23, name="toy44x")
    if name == "toy44":  # small ring, headline-size primes (= 1 mod 2^20): large-m witness-map tests
        return make_params(32, [43, 44], 64, [43, 44, 44], ring_factor=1 << 20, name="toy44")
    if name == "toy49":  # stresses the 50-bit bound of the FP64 modmul path
        return make_params(64, [49, 49], 128, [49, 49, 49], ring_factor=1 << 12, name="toy49")
    raise KeyError(name)

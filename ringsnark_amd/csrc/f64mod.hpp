// f64mod.hpp -- exact modular arithmetic on FP64 for primes p < 2^50 (host + gfx950 device).
//
// Why FP64: gfx950 has no native 64x64 integer multiply (a 64-bit mulhi+mullo pair lowers to
// ~9 quarter-rate v_mad_u64_u32), while v_fma_f64 issues at half the full VALU rate.  Every
// residue on this path is < 2^50 (36..49-bit primes), so products can
// be taken exactly with the FMA error-free transformation:
//
//     h = fl(a*b), l = fma(a,b,-h)          =>  a*b = h + l            (exact)
//     k = rint(fl(h * pinv))                =>  |k - a*b/p| <= 1/2 + 3*eps*|a*b/p|
//     r = fma(-k, p, h) + l                 =>  r = a*b - k*p          (exact, |r| <= 0.75 p)
//
// valid whenever |a*b| <= p * 2^49 (then 3*eps*|a*b/p| <= 1/4, |r| < 2^53 and both the fma and
// the final add are exact because their results are integers below 2^53).  All values are held
// as signed ("balanced") integers in doubles; callers keep |v| <= 2^50 (mul operand) and
// |v| <= 2^52 (add operand).  Canonicalisation maps to [0, p).  Bit-exactness against a 128-bit
// integer reference is asserted in tests/test_f64mod.py (CPU) and in every GPU parity test.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define RS_HD __host__ __device__ __forceinline__
#else
#define RS_HD inline
#endif

namespace rs {

struct Mod {
  double p;     // the prime
  double pinv;  // fl(1/p)
};

RS_HD double f64_rint(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_rint(x);  // v_rndne_f64
#else
  return std::nearbyint(x);
#endif
}
RS_HD double f64_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fma(a, b, c);
#else
  return std::fma(a, b, c);
#endif
}

// a*b mod p, |result| <= 0.75p.  Requires |a*b| <= p*2^49.
RS_HD double mulmod(double a, double b, const Mod &m) {
  double h = a * b;
  double l = f64_fma(a, b, -h);
  double k = f64_rint(h * m.pinv);
  double r = f64_fma(-k, m.p, h);
  return r + l;
}
// partial reduction: |result| <= p/2 (+1 ulp-ish slack).  Requires |a| <= 2^52.
RS_HD double reduce(double a, const Mod &m) {
  double k = f64_rint(a * m.pinv);
  return f64_fma(-k, m.p, a);
}
// canonical residue in [0,p) as a double.  Requires |a| <= 2^52.
RS_HD double canon(double a, const Mod &m) {
  double r = reduce(a, m);
  return r < 0.0 ? r + m.p : r;
}
// exact u64 <-> f64 for values < 2^52: OR the magic exponent, subtract 2^52 (1 FP op).
RS_HD double from_u64(uint64_t v) {
  union { uint64_t u; double d; } x;
  x.u = v | 0x4330000000000000ull;
  return x.d - 4503599627370496.0;
}
// v must be an integer in [0, 2^52)
RS_HD uint64_t to_u64(double v) {
  union { uint64_t u; double d; } x;
  x.d = v + 4503599627370496.0;
  return x.u & 0x000FFFFFFFFFFFFFull;
}
// balanced representative of a canonical residue (|result| <= (p-1)/2): the SEAL centred lift
// rule c >= (t+1)/2  =>  c - t   (Evaluator::transform_to_ntt_inplace on a Plaintext).
RS_HD double center(double c, const Mod &m) { return (c + c > m.p) ? c - m.p : c; }
// The same centred representative straight from a balanced value with |a| <= p (a mulmod result, a partially reduced
// sum): one reduction step decides exactly, because the integer a is at least 1/(2p) >= 2^-51 away from the rounding
// boundaries +-p/2 in units of p while the computed quotient a*pinv errs by less than 2^-52.  Equals center(canon(a)):
// three instructions instead of twelve (two compares and four 64-bit selects among them).
RS_HD double center_balanced(double a, const Mod &m) { return reduce(a, m); }

}  // namespace rs

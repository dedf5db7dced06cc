// intmod.hpp -- 64-bit integer modular arithmetic (Montgomery reduction) for primes 2^50 <= p < 2^62: the
// arithmetic of contexts whose moduli do not fit the exact-FP64 path of f64mod.hpp (SEAL's 54-bit
// BFVDefault(2048) prime of bench_logistic_regression_inference.cpp:20-27, the {59,60,60}-bit primes of
// microbench.cpp:35-36).  gfx950 has no 64x64 multiplier: one Montgomery product is two 64x64->128 products
// and one 64x64->64 (about fourteen quarter-rate v_mad_u64_u32 / v_mul_lo_u32), i.e. several times the cost of
// the six FP64 instructions of f64mod.hpp -- which is why FP64 stays the default below 2^50.
//
// Representation: canonical residues in [0, p) everywhere (registers, LDS, workspaces): no lazy values, so
// reduce / canon / center are identities here.  Montgomery form is used for CONSTANTS only: every table entry
// (twiddles, n^-1, spectra of slot-constant polynomials, R1CS coefficients) is stored as c*R mod p, R = 2^64,
// so that
//     mulmod(a, c_mont)  = REDC(a * c*R)      = a*c        data x constant -> data, one reduction
//     mulmod_dd(a, b)    = REDC(REDC(a*b) * R^2) = a*b      data x data     -> data, two reductions
// and no value ever has to be converted into or out of Montgomery form.
//
// The kernels are written once against the overload set {mulmod, mulmod_dd, addm, subm, negm, reduce, canon,
// center, from_res<T>, to_res} and instantiated for (double, Mod) and (uint64_t, ModI).
#pragma once
#include <cstdint>

#include "f64mod.hpp"

namespace rs {

struct ModI {
  uint64_t p;     // the prime, < 2^62
  uint64_t ninv;  // -p^-1 mod 2^64
  uint64_t r2;    // 2^128 mod p
};

RS_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
// REDC(a*b): a*b*2^-64 mod p, inputs < p (or one of them < 2^64 with the other < p), result in [0, p)
RS_HD uint64_t montmul(uint64_t a, uint64_t b, const ModI &m) {
  const uint64_t hi = mulhi64(a, b), lo = a * b;
  const uint64_t k = lo * m.ninv;
  const uint64_t r = hi + mulhi64(k, m.p) + (lo != 0);  // (a*b + k*p) / 2^64: the low words cancel to 0 or 2^64
  return r >= m.p ? r - m.p : r;
}
RS_HD uint64_t mulmod(uint64_t a, uint64_t c_mont, const ModI &m) { return montmul(a, c_mont, m); }
RS_HD uint64_t mulmod_dd(uint64_t a, uint64_t b, const ModI &m) { return montmul(montmul(a, b, m), m.r2, m); }
RS_HD uint64_t addm(uint64_t a, uint64_t b, const ModI &m) {
  const uint64_t s = a + b;
  return s >= m.p ? s - m.p : s;
}
RS_HD uint64_t subm(uint64_t a, uint64_t b, const ModI &m) { return a >= b ? a - b : a + m.p - b; }
RS_HD uint64_t negm(uint64_t a, const ModI &m) { return a ? m.p - a : 0; }
RS_HD uint64_t reduce(uint64_t a, const ModI &) { return a; }
RS_HD uint64_t canon(uint64_t a, const ModI &) { return a; }
RS_HD uint64_t center(uint64_t a, const ModI &) { return a; }
RS_HD uint64_t to_res(uint64_t v) { return v; }
// data value -> Montgomery form (so that a later mulmod(x, result) is a data x data product with one reduction)
RS_HD uint64_t to_mont(uint64_t a, const ModI &m) { return montmul(a, m.r2, m); }
// the integer a slot-constant small value stands for (loop counters, 0, 1)
RS_HD uint64_t small_val(uint64_t v, const ModI &m) { return v % m.p; }

// the VALUE of a table constant (used where a coefficient is added, not multiplied: index-0 R1CS terms)
RS_HD uint64_t konst_value(uint64_t c_mont, const ModI &m) { return montmul(c_mont, 1, m); }

// Centred lift of a mod-t residue to an integer in (-t/2, t/2] (SEAL Evaluator::transform_to_ntt_inplace on a
// Plaintext: c >= (t+1)/2 -> c - t), and its residue modulo another prime.
RS_HD int64_t lift_centered(uint64_t c, const ModI &t) { return (c + c > t.p) ? (int64_t)c - (int64_t)t.p : (int64_t)c; }
RS_HD uint64_t lift_residue(int64_t v, const ModI &Q) {
  const int64_t r = v % (int64_t)Q.p;
  return (uint64_t)(r < 0 ? r + (int64_t)Q.p : r);
}

// ---- the same vocabulary for the FP64 arithmetic (f64mod.hpp) -------------------------------------------------
RS_HD double mulmod_dd(double a, double b, const Mod &m) { return mulmod(a, b, m); }
RS_HD double addm(double a, double b, const Mod &) { return a + b; }  // lazy: callers reduce where f64mod.hpp requires
RS_HD double subm(double a, double b, const Mod &) { return a - b; }
RS_HD double negm(double a, const Mod &) { return -a; }
RS_HD uint64_t to_res(double v) { return to_u64(v); }  // v canonical
RS_HD double to_mont(double a, const Mod &) { return a; }
RS_HD double small_val(uint64_t v, const Mod &) { return (double)v; }
RS_HD double konst_value(double c, const Mod &) { return c; }
RS_HD double lift_centered(double c, const Mod &t) { return center(c, t); }
RS_HD double lift_residue(double v, const Mod &Q) { return reduce(v, Q); }

template <class T>
RS_HD T from_res(uint64_t v);
template <>
RS_HD double from_res<double>(uint64_t v) {
  return from_u64(v);
}
template <>
RS_HD uint64_t from_res<uint64_t>(uint64_t v) {
  return v;
}

// value type / lifted-integer type / modulus type of each arithmetic
template <class M>
struct ArithOf;
template <>
struct ArithOf<Mod> {
  using T = double;
  using Lift = double;
};
template <>
struct ArithOf<ModI> {
  using T = uint64_t;
  using Lift = int64_t;
};

}  // namespace rs

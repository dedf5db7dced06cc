// host_math.hpp -- host-side number theory used to build device tables (product code; it does
// not link or include anything from oracle/).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace rs::host {

typedef unsigned __int128 u128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) {
  uint64_t s = a + b;
  return s >= q ? s - q : s;
}
inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
inline uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  a %= q;
  for (; e; e >>= 1) {
    if (e & 1) r = mulmod(r, a, q);
    a = mulmod(a, a, q);
  }
  return r;
}
inline uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a, q - 2, q); }

inline bool is_prime(uint64_t n) {
  if (n < 2) return false;
  static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (uint64_t b : bases) {
    if (n == b) return true;
    if (n % b == 0) return false;
  }
  uint64_t d = n - 1;
  int r = 0;
  while (!(d & 1)) d >>= 1, r++;
  for (uint64_t b : bases) {
    uint64_t x = powmod(b, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int k = 1; k < r && comp; k++) {
      x = mulmod(x, x, n);
      if (x == n - 1) comp = false;
    }
    if (comp) return false;
  }
  return true;
}

inline int two_adicity(uint64_t q) {
  int v = 0;
  for (uint64_t x = q - 1; !(x & 1); x >>= 1) v++;
  return v;
}

// Smallest primitive `degree`-th root of unity mod q (degree a power of two): the convention
// of SEAL's NTTTables (util::try_minimal_primitive_root), which fixes the NTT output order the
// CRS ciphertexts are stored in.
inline uint64_t minimal_primitive_root(uint64_t degree, uint64_t q) {
  if ((q - 1) % degree) throw std::invalid_argument("prime is not 1 mod NTT degree");
  uint64_t quo = (q - 1) / degree, root = 0;
  for (uint64_t g = 2; g < q && !root; g++) {
    uint64_t c = powmod(g, quo, q);
    if (powmod(c, degree >> 1, q) == q - 1) root = c;
  }
  uint64_t sq = mulmod(root, root, q), cur = root, best = root;
  for (uint64_t i = 0; i < (degree >> 1); i++) {
    if (cur < best) best = cur;
    cur = mulmod(cur, sq, q);
  }
  return best;
}

// Any primitive `degree`-th root (used for the witness map's cyclic transforms, where the
// choice is internal and cancels out).
inline uint64_t some_primitive_root(uint64_t degree, uint64_t q) {
  if ((q - 1) % degree) throw std::invalid_argument("prime lacks the 2-adicity for this transform");
  uint64_t quo = (q - 1) / degree;
  for (uint64_t g = 2; g < q; g++) {
    uint64_t c = powmod(g, quo, q);
    if (degree == 1 || powmod(c, degree >> 1, q) == q - 1) return c;
  }
  throw std::runtime_error("no primitive root");
}

inline uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}

// Montgomery constants of intmod.hpp (R = 2^64)
inline uint64_t mont_form(uint64_t v, uint64_t q) { return (uint64_t)((((u128)(v % q)) << 64) % q); }
inline uint64_t mont_ninv(uint64_t q) {  // -q^-1 mod 2^64 (Newton iteration on odd q)
  uint64_t x = q;
  for (int i = 0; i < 6; i++) x *= 2 - q * x;
  return (uint64_t)0 - x;
}
inline uint64_t mont_r2(uint64_t q) { return mont_form(mont_form(1, q), q); }

// balanced representative as a double: value in (-q/2, q/2]
inline double balanced(uint64_t v, uint64_t q) { return v > q / 2 ? -(double)(q - v) : (double)v; }

}  // namespace rs::host

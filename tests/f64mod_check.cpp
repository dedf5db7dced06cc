// Host-side exactness check of ringsnark_amd/csrc/f64mod.hpp against 128-bit integer arithmetic.
// The same header is compiled for gfx950; IEEE-754 binary64 with FMA behaves identically.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "../ringsnark_amd/csrc/f64mod.hpp"

typedef unsigned __int128 u128;
typedef __int128 i128;
static uint64_t sm(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static long long fails = 0;
static void expect(bool c, const char *what) { if (!c) { if (fails < 10) fprintf(stderr, "FAIL %s\n", what); fails++; } }
static int64_t imod(i128 v, uint64_t p) { i128 r = v % (i128)p; if (r < 0) r += p; return (int64_t)r; }

int main(int argc, char **argv) {
  std::vector<uint64_t> primes;
  for (int i = 1; i < argc; i++) primes.push_back(strtoull(argv[i], nullptr, 0));
  uint64_t st = 42;
  for (uint64_t p : primes) {
    rs::Mod m{(double)p, 1.0 / (double)p};
    const double lim = 1125899906842624.0;  // 2^50
    for (int it = 0; it < 2000000; it++) {
      // operands: a up to 2^50 in magnitude (lazy NTT value), b balanced residue
      uint64_t ra = sm(st), rb = sm(st);
      int64_t a, b;
      switch (it & 7) {
        case 0: a = (int64_t)(ra % p); b = (int64_t)(rb % p); break;                      // canonical x canonical
        case 1: a = (int64_t)(ra % (uint64_t)lim) - (int64_t)(lim / 2); b = (int64_t)(rb % p) - (int64_t)(p / 2); break;
        case 2: a = (int64_t)lim - (int64_t)(ra & 3); b = (int64_t)(p / 2) - (int64_t)(rb & 3); break;  // extreme corner
        case 3: a = -(int64_t)lim + (int64_t)(ra & 3); b = -(int64_t)(p / 2) + (int64_t)(rb & 3); break;
        case 4: a = (int64_t)(p - 1 - (ra & 1)); b = (int64_t)(p - 1 - (rb & 1)); break;
        case 5: a = (int64_t)(ra & 7); b = (int64_t)(rb % p); break;
        case 6: a = (int64_t)((p / 2) + (ra & 3)); b = (int64_t)((p / 2) + (rb & 3)); break;
        default: a = (int64_t)(ra % (uint64_t)lim); b = -(int64_t)(rb % (p / 2 + 1)); break;
      }
      // mulmod precondition: |a*b| <= p * 2^49
      i128 prod = (i128)a * b;
      i128 bound = (i128)p << 49;
      if (prod > bound || -prod > bound) continue;
      double r = rs::mulmod((double)a, (double)b, m);
      expect(r == (double)(int64_t)r, "mulmod integral");
      expect(r <= 0.75 * (double)p + 1 && r >= -0.75 * (double)p - 1, "mulmod bound");
      expect(imod((i128)(int64_t)r, p) == imod(prod, p), "mulmod value");
      double c = rs::canon(r, m);
      expect(c >= 0 && c < (double)p && (int64_t)c == imod(prod, p), "canon");
      // reduce / canon on large inputs up to 2^52
      int64_t big = (int64_t)(ra % (1ull << 52)) - (1ll << 51);
      double rr = rs::reduce((double)big, m);
      expect(rr == (double)(int64_t)rr && rr <= 0.5 * (double)p + 1 && rr >= -0.5 * (double)p - 1, "reduce bound");
      expect(imod((i128)(int64_t)rr, p) == imod(big, p), "reduce value");
      double cc = rs::canon((double)big, m);
      expect(cc >= 0 && cc < (double)p && (int64_t)cc == imod(big, p), "canon big");
      // u64 <-> f64
      uint64_t u = ra % p;
      expect(rs::to_u64(rs::from_u64(u)) == u && rs::from_u64(u) == (double)u, "u64 roundtrip");
      double ce = rs::center((double)u, m);
      expect((u >= (p + 1) / 2) ? (ce == (double)u - (double)p) : (ce == (double)u), "center");
      // center_balanced: the same representative straight from a mulmod result (|r| <= 0.75 p) and from reduce(big)
      expect(rs::center_balanced(r, m) == rs::center(c, m), "center_balanced of a product");
      expect(rs::center_balanced(rr, m) == rs::center(cc, m), "center_balanced of a reduced value");
    }
    // the decision boundaries themselves: a = +-(p-1)/2 stays, +-(p+1)/2 wraps, and everything within 3 of +-p/2, +-p, 0
    for (int64_t base : {(int64_t)(p / 2), -(int64_t)(p / 2), (int64_t)p, -(int64_t)p, (int64_t)0, (int64_t)(p - p / 4), -(int64_t)(p - p / 4)})
      for (int64_t d = -3; d <= 3; d++) {
        const int64_t a = base + d;
        if (a > (int64_t)p || a < -(int64_t)p) continue;
        const double want = rs::center((double)imod((i128)a, p), m);
        expect(rs::center_balanced((double)a, m) == want, "center_balanced at a boundary");
      }
  }
  if (fails) { fprintf(stderr, "%lld failures\n", fails); return 1; }
  printf("ok\n");
  return 0;
}

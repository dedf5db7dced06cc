// ref_r1cs_probe.cpp -- checker built FROM THE REFERENCE'S OWN HEADERS (compiled where they lie
// under /root/reference; nothing is copied): instantiates ringsnark::r1cs_constraint_system,
// linear_combination::evaluate (relations/variable.tcc:246-254) and is_satisfied
// (relations/constraint_satisfaction_problems/r1cs/r1cs.tcc) over a prime field Z_q, the same
// way the reference's util/test_utils.hpp instantiates its templates with PrimitiveWrapper<T>.
// TEST INFRASTRUCTURE ONLY.  Output lands in oracle/_ref/ (git-ignored).
//
// stdin:  q m n_vars n_inputs
//         for each of a,b,c and each constraint: k  (idx coeff) * k
//         n_vars assignment values
// stdout: "sat <0|1>" then m lines "a b c" (the three evaluated linear combinations per row)
// Vector form ("vec q S m n_vars n_inputs"): the ring is Z_q^S with slot-wise operations, every coefficient and every
// value is S residues; the evaluated combinations print as comma-separated slot lists.
#include <cstdint>
#include <iostream>
#include <string>
#include <vector>

#include <ringsnark/relations/constraint_satisfaction_problems/r1cs/r1cs.hpp>

struct Zq {
  static inline uint64_t q = 3;
  uint64_t v = 0;
  Zq() = default;
  Zq(long x) : v((uint64_t)(((x % (long)q) + (long)q) % (long)q)) {}
  static Zq from_u64(uint64_t x) {
    Zq r;
    r.v = x % q;
    return r;
  }
  static Zq zero() { return Zq(0); }
  static Zq one() { return Zq(1); }
  bool is_zero() const { return v == 0; }
  Zq &operator+=(const Zq &o) {
    v = (uint64_t)(((unsigned __int128)v + o.v) % q);
    return *this;
  }
  Zq &operator-=(const Zq &o) {
    v = (v + q - o.v) % q;
    return *this;
  }
  Zq &operator*=(const Zq &o) {
    v = (uint64_t)(((unsigned __int128)v * o.v) % q);
    return *this;
  }
  Zq operator-() const { return from_u64(v ? q - v : 0); }
  bool operator==(const Zq &o) const { return v == o.v; }
  bool operator!=(const Zq &o) const { return v != o.v; }
};
inline Zq operator+(Zq a, const Zq &b) { return a += b; }
inline Zq operator-(Zq a, const Zq &b) { return a -= b; }
inline Zq operator*(Zq a, const Zq &b) { return a *= b; }
inline std::ostream &operator<<(std::ostream &o, const Zq &z) { return o << z.v; }
inline std::istream &operator>>(std::istream &i, Zq &z) {
  uint64_t x;
  i >> x;
  z = Zq::from_u64(x);
  return i;
}

// Z_q^S with slot-wise operations: what ringsnark::seal::RingElem is in NTT form (seal_ring.tcc:105-247 are all
// dyadic), so one coefficient object can hold a different residue in every slot -- the "polynomial" coefficients of
// benchmarks/bench_ntt_SEAL.cpp:46-53.
struct ZqVec {
  static inline size_t S = 1;
  std::vector<Zq> v;
  ZqVec() : v(S) {}
  ZqVec(long x) : v(S, Zq(x)) {}
  static ZqVec zero() { return ZqVec(0); }
  static ZqVec one() { return ZqVec(1); }
  bool is_zero() const {
    for (const auto &x : v)
      if (!x.is_zero()) return false;
    return true;
  }
  ZqVec &operator+=(const ZqVec &o) {
    for (size_t i = 0; i < S; i++) v[i] += o.v[i];
    return *this;
  }
  ZqVec &operator-=(const ZqVec &o) {
    for (size_t i = 0; i < S; i++) v[i] -= o.v[i];
    return *this;
  }
  ZqVec &operator*=(const ZqVec &o) {
    for (size_t i = 0; i < S; i++) v[i] *= o.v[i];
    return *this;
  }
  ZqVec operator-() const {
    ZqVec r;
    for (size_t i = 0; i < S; i++) r.v[i] = -v[i];
    return r;
  }
  bool operator==(const ZqVec &o) const { return v == o.v; }
  bool operator!=(const ZqVec &o) const { return !(v == o.v); }
};
inline ZqVec operator+(ZqVec a, const ZqVec &b) { return a += b; }
inline ZqVec operator-(ZqVec a, const ZqVec &b) { return a -= b; }
inline ZqVec operator*(ZqVec a, const ZqVec &b) { return a *= b; }
inline std::ostream &operator<<(std::ostream &o, const ZqVec &z) {
  for (size_t i = 0; i < ZqVec::S; i++) o << (i ? "," : "") << z.v[i];
  return o;
}
inline std::istream &operator>>(std::istream &i, ZqVec &z) {
  z = ZqVec();
  for (auto &x : z.v) i >> x;
  return i;
}

template <class F>
int run(size_t m, size_t n_vars, size_t n_inputs) {
  using namespace ringsnark;
  r1cs_constraint_system<F> cs;
  cs.primary_input_size = n_inputs;
  cs.auxiliary_input_size = n_vars - n_inputs;
  std::vector<linear_combination<F>> lc[3];
  for (int w = 0; w < 3; w++) {
    lc[w].resize(m);
    for (size_t i = 0; i < m; i++) {
      size_t k;
      std::cin >> k;
      for (size_t e = 0; e < k; e++) {
        size_t idx;
        F c;
        std::cin >> idx >> c;
        lc[w][i].add_term(linear_term<F>(variable<F>(idx), c));
      }
    }
  }
  for (size_t i = 0; i < m; i++) cs.add_constraint(r1cs_constraint<F>(lc[0][i], lc[1][i], lc[2][i]));
  std::vector<F> full(n_vars);
  for (auto &x : full) std::cin >> x;
  r1cs_primary_input<F> primary(full.begin(), full.begin() + n_inputs);
  r1cs_auxiliary_input<F> aux(full.begin() + n_inputs, full.end());
  std::cout << "sat " << (cs.is_satisfied(primary, aux) ? 1 : 0) << "\n";
  for (size_t i = 0; i < m; i++)
    std::cout << cs.constraints[i].a.evaluate(full) << " " << cs.constraints[i].b.evaluate(full) << " "
              << cs.constraints[i].c.evaluate(full) << "\n";
  return 0;
}

int main() {
  std::string first;
  std::cin >> first;
  size_t m, n_vars, n_inputs;
  if (first == "vec") {  // vec q S m n_vars n_inputs: every coefficient and every value is S residues
    std::cin >> Zq::q >> ZqVec::S >> m >> n_vars >> n_inputs;
    return run<ZqVec>(m, n_vars, n_inputs);
  }
  Zq::q = std::stoull(first);
  std::cin >> m >> n_vars >> n_inputs;
  return run<Zq>(m, n_vars, n_inputs);
}

// adapter_run.cpp -- run-time check of the header-only C++ adapters (include/ringsnark_amd/ring.hpp)
// against librs_hip.so: plain C++17, no HIP headers, the way a maintainer of the reference would use
// them (ringsnark::seal::{RingElem, EncodingElem} -> ringsnark::amd::{...}).  TEST INFRASTRUCTURE.
//
// usage: adapter_run N L q_0..q_{L-1} N_enc K Q_0..Q_{K-1}
// Checks ring identities, the reference's error messages, and the encoding homomorphism
//   decode(<E(a_t), r_t>) = sum_t a_t r_t   through encode / operator*= / operator+= / inner_product.
#include <cstdio>
#include <cstdlib>
#include <random>

#include <ringsnark_amd/ring.hpp>

using namespace ringsnark::amd;

static int fails = 0;
#define EXPECT(c)                                         \
  do {                                                    \
    if (!(c)) {                                           \
      std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
      fails++;                                            \
    }                                                     \
  } while (0)

static RingElem random_ring(std::mt19937_64 &g, bool nonzero) {
  const Params &p = Context::get_params();
  std::vector<uint64_t> w(Context::ring_words());
  for (int i = 0; i < p.L; i++)
    for (int x = 0; x < p.N; x++) {
      uint64_t v = g() % p.q[i];
      if (nonzero && v == 0) v = 1;
      w[(size_t)i * p.N + x] = v;
    }
  return RingElem(std::move(w));
}

int main(int argc, char **argv) {
  int a = 1;
  Params p;
  p.N = std::atoi(argv[a++]);
  p.L = std::atoi(argv[a++]);
  for (int i = 0; i < p.L; i++) p.q.push_back(std::strtoull(argv[a++], nullptr, 10));
  p.N_enc = std::atoi(argv[a++]);
  p.K = std::atoi(argv[a++]);
  for (int i = 0; i < p.K; i++) p.Q.push_back(std::strtoull(argv[a++], nullptr, 10));
  EXPECT(a == argc);

  try {
    Context::get_context();
    EXPECT(false);
  } catch (const std::invalid_argument &e) {
    EXPECT(std::string(e.what()) == "context not set");
  }
  Context::set_context(p);
  try {
    Context::set_context(p);
    EXPECT(false);
  } catch (const std::invalid_argument &e) {
    EXPECT(std::string(e.what()) == "cannot re-set context once set");
  }

  std::mt19937_64 g(7);
  const RingElem x = random_ring(g, true), y = random_ring(g, true), z = random_ring(g, false);
  EXPECT((x + y) - y == x);
  EXPECT(x * y == y * x);
  EXPECT(x * (y + z) == x * y + x * z);
  EXPECT(x * x.inverse() == RingElem::one());
  EXPECT((x / y) * y == x);
  EXPECT(-(-z) == z);
  EXPECT(RingElem(5) * x == x + x + x + x + x);  // scalar promoted to a polynomial (seal_ring.tcc:265-277)
  EXPECT(RingElem::zero() * x == RingElem::zero() && (RingElem::zero() * x).is_zero());
  {
    std::vector<uint64_t> w = x.get_poly();
    w[3] = 0;  // one zero slot: not invertible
    try {
      RingElem(w).invert_inplace();
      EXPECT(false);
    } catch (const std::invalid_argument &e) {
      EXPECT(std::string(e.what()) == "element is not invertible in ring");
    }
    EXPECT(!RingElem(w).is_invertible() && x.is_invertible());
  }

  // encodings: the all-zero secret key is a valid (if useless) BGV key, c0 = m - t*e
  const EncodingElem::SecretKey sk((size_t)p.K * p.N_enc, 0);
  const std::vector<RingElem> as = {x, y, z};
  const std::vector<EncodingElem> es = EncodingElem::encode(sk, as, 11);
  EXPECT(es.size() == 3 && !es[0].is_empty());
  for (int t = 0; t < 3; t++) EXPECT(EncodingElem::decode(sk, es[t]) == as[t]);
  const RingElem r0 = random_ring(g, false), r1 = random_ring(g, false), r2 = RingElem(1);
  EncodingElem acc = es[0];
  acc *= r0;
  acc += es[1] * r1;
  acc += es[2] * r2;  // Scalar 1: no-op multiply (seal_ring.tcc:525-527)
  const RingElem want = x * r0 + y * r1 + z;
  EXPECT(EncodingElem::decode(sk, acc) == want);
  const std::vector<RingElem> rs = {r0, r1, r2};
  const EncodingElem ip = EncodingElem::inner_product(es.begin(), es.end(), rs.begin(), rs.end());
  EXPECT(EncodingElem::decode(sk, ip) == want);
  EXPECT(ip == acc);  // same operations, same ciphertext
  const std::vector<RingElem> zeros = {RingElem(0), RingElem(0), RingElem(0)};
  EXPECT(EncodingElem::inner_product(es.begin(), es.end(), zeros.begin(), zeros.end()).is_empty());  // seal_ring.tcc:412,432
  EXPECT((EncodingElem() += es[0]) == es[0]);

  if (fails) return 1;
  std::puts("adapter_run: OK");
  return 0;
}

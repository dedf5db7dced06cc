// adapter_run.cpp -- run-time check of the header-only C++ adapters (include/ringsnark_amd/ring.hpp)
// against librs_hip.so: plain C++17, no HIP headers, the way a maintainer of the reference would use
// them (ringsnark::seal::{RingElem, EncodingElem} -> ringsnark::amd::{...}).  TEST INFRASTRUCTURE.
//
// usage: adapter_run N L q_0..q_{L-1} N_enc K Q_0..Q_{K-1}
// Checks ring identities, the reference's error messages, and the encoding homomorphism
//   decode(<E(a_t), r_t>) = sum_t a_t r_t   through encode / operator*= / operator+= / inner_product.
#define RINGSNARK_AMD_TESTING 1  // Context::seed_prng: reproducible draws for the fixtures (never in production builds)
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
  {  // representation follows the reference (seal_ring.tcc:105-247): which results stay a Scalar
    const size_t qb = 64 - __builtin_clzll(p.q[0]);
    const RingElem s3(3), s4(4), big((uint64_t)1 << (qb - 2));
    EXPECT((s3 + s4).is_scalar() && (s3 + s4).get_scalar() == 7);      // bit sizes 2, 3 -> 3 < |q_1|
    EXPECT((s3 * s4).is_scalar() && (s3 * s4).get_scalar() == 12);     // 2 + 3 < |q_1|
    EXPECT((big + big).is_poly());                                     // equal sizes: |q_1| - 1 + 1 is not < |q_1|
    EXPECT(big + big == RingElem((uint64_t)1 << (qb - 1)));            // ... with the right value
    EXPECT((big * s4).is_poly() && big * s4 == (big + big) + (big + big));
    EXPECT((s4 - s3).is_poly() && s4 - s3 == RingElem::one());         // Scalar - Scalar is always promoted (:176-178)
    EXPECT((s3 + RingElem::zero()).is_poly() && s3 + RingElem::zero() == s3);  // bit size of 0: promoted (see ring.hpp)
    EXPECT((RingElem::zero() + s3).is_scalar());                       // Scalar 0 on the left copies the operand (:121-124)
    EXPECT((x * RingElem::zero()).is_scalar() && (x * RingElem::zero()).get_scalar() == 0);  // :196-199
    EXPECT((RingElem::one() * x).is_poly() && (RingElem::one() * s4).is_scalar());
    EXPECT((x * RingElem::one()).is_poly() && x * RingElem::one() == x);
    EXPECT((s3 * x).is_poly() && s3 * x == x + x + x && x * s3 == s3 * x);
    EXPECT((s3 + x).is_poly() && s3 + x == x + s3 && (x + s3) - s3 == x);
    EXPECT((x - RingElem::zero()).is_poly() && x - RingElem::zero() == x);
  }
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

  // the rest of the <RingT, EncT> concept
  {
    Context::seed_prng(99);
    const RingElem u = RingElem::random_invertible_element(), nz = RingElem::random_nonzero_element();
    EXPECT(u.is_invertible() && !nz.is_zero() && u.is_poly());
    struct Dom {
      size_t m = 40;
    };
    const RingElem s = RingElem::random_exceptional_element(std::make_shared<Dom>());
    EXPECT(s.is_scalar() && s.get_scalar() > 40 && s.get_scalar() < p.q[0]);
    EXPECT(RingElem::random_exceptional_element().is_scalar());
    EXPECT(std::hash<RingElem>()(u) == std::hash<RingElem>()(RingElem(u)) && std::hash<RingElem>()(RingElem(7)) == std::hash<uint64_t>()(7));
    auto [pk_enc, sk2] = EncodingElem::keygen();
    (void)pk_enc;
    EXPECT(sk2.size() == (size_t)p.K * p.N_enc && EncodingElem::size_in_bits_pk(nullptr) == 0 && EncodingElem::size_in_bits_sk(sk2) > 0);
    const std::vector<EncodingElem> e2 = EncodingElem::encode(sk2, {u, nz});
    EXPECT(EncodingElem::decode(sk2, e2[0]) == u && EncodingElem::decode(sk2, e2[1]) == nz);
    EXPECT(EncodingElem::decode(sk2, e2[0] * nz + e2[1]) == u * nz + nz);
    {  // the verifier's guard (seal_ring.tcc:443-454): an encoding multiplied past its noise budget is refused
      EncodingElem spent = e2[0];
      for (int k = 0; k < 8; k++) spent *= nz;
      try {
        (void)EncodingElem::decode(sk2, spent);
        EXPECT(false);
      } catch (const EncodingElem::decoding_error &e) {
        const std::string w = e.what();  // "decoding error: ciphertext #0 has remaining noise budget 0 <= 0 ..."
        EXPECT(w.rfind("decoding error: ciphertext #", 0) == 0 && w.find("has remaining noise budget 0 <= 0") != std::string::npos);
      }
    }
    // util/polynomials.hpp helpers: divide(multiply(q, x), x) == q (util/division_test.cpp:28-49, n = 12)
    std::vector<RingElem> xs, qs;
    for (uint64_t i = 0; i < 12; i++) {
      xs.push_back(RingElem(2 * i + 1));
      qs.push_back(RingElem(i + 1));
    }
    const std::vector<RingElem> prod = multiply(qs, xs), back = divide(prod, xs);
    EXPECT(prod.size() == 23 && back.size() == 12);
    for (size_t i = 0; i < back.size() && i < 12; i++) EXPECT(back[i] == qs[i]);
    // interpolation_test.cpp:29-55: nodes 0..7, coefficients 0..7
    std::vector<RingElem> ys;
    for (uint64_t xv = 0; xv < 8; xv++) {
      uint64_t acc2 = 0;
      for (int k = 7; k >= 0; k--) acc2 = acc2 * xv + (uint64_t)k;
      ys.push_back(RingElem(acc2));
    }
    const std::vector<RingElem> cf = interpolate_on_domain(ys);
    for (uint64_t k = 0; k < 8; k++) EXPECT(cf[k] == RingElem(k));
  }

  if (fails) return 1;
  std::puts("adapter_run: OK");
  return 0;
}

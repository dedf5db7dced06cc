// ref_adapter_prove.cpp -- the reference's OWN gadgetlib / relations / proof-system base headers (compiled where
// they lie under /root/reference; nothing is copied) instantiated with RingT = ringsnark::amd::RingElem and
// EncT = ringsnark::amd::EncodingElem, i.e. on the MI355X library through the C++ adapters.  This is the
// north-star claim "gadgetlib/relations are untouched" made executable:
//
//   protoboard<R> + pb_variable_array<R> (gadgetlib/protoboard.hpp, pb_variable.hpp) build a circuit whose linear
//   combinations have several terms, integer / ring coefficients and constants; pb.is_satisfied()
//   (relations/.../r1cs.tcc) evaluates it with device ring arithmetic; the constraint system is exported to CSR
//   (export_csr), a key is encrypted on the device (EncodingElem::keygen / encode), and
//   ringsnark::amd::groth16::prover / rinocchio::prover are called with the reference's argument order
//   (pk, primary_input, auxiliary_input).
//
// Everything the Python side needs to recompute the proofs through ctypes is written as raw little-endian u64
// files into <outdir>.  TEST INFRASTRUCTURE; the binary lands in oracle/_ref/ (git-ignored, travels to the GPU box).
//
// usage: ref_adapter_prove N L q.. N_enc K Q.. m outdir [poly]
// poly: the circuit's ring coefficients are general ring elements (random polynomials) instead of the Scalar 5 -- on a
// primary input and on an auxiliary variable -- as in benchmarks/bench_ntt_SEAL.cpp:46-53 (`row * vars[i]`).
#define RINGSNARK_AMD_TESTING 1  // Context::seed_prng: reproducible draws for the fixtures (never in production builds)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include <ringsnark_amd/ring.hpp>

#include <ringsnark/gadgetlib/protoboard.hpp>
#include <ringsnark/relations/constraint_satisfaction_problems/r1cs/r1cs.hpp>
#include <ringsnark/zk_proof_systems/r1cs_ppzksnark.hpp>

using R = ringsnark::amd::RingElem;
using E = ringsnark::amd::EncodingElem;
using ringsnark::amd::Context;

// what groth16::proving_key<R, E> / rinocchio::proving_key<R, E> hold (zk_proof_systems/groth16/groth16.hpp:9-48,
// rinocchio/rinocchio.hpp:9-60); those two headers pull Boost through util/polynomials.tcc, so the key types are
// restated as plain structs deriving from the reference's abstract base
struct Groth16Key : ringsnark::proving_key<R, E> {
  ringsnark::r1cs_constraint_system<R> constraint_system;
  E alpha, beta;
  std::vector<E> s_pows, delta_mid, delta_ts;
  size_t size_in_bits() const override { return s_pows.size() * s_pows[0].size_in_bits(); }
};
struct RinocchioKey : ringsnark::proving_key<R, E> {
  ringsnark::r1cs_constraint_system<R> constraint_system;
  std::vector<E> s_pows, alpha_s_pows, beta_prods;
  E beta_rv_ts, beta_rw_ts, beta_ry_ts;
  size_t size_in_bits() const override { return s_pows.size() * s_pows[0].size_in_bits(); }
};

static void dump(const std::string &path, const std::vector<uint64_t> &w) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char *>(w.data()), (std::streamsize)(w.size() * 8));
}
static std::vector<uint64_t> words_of(const std::vector<E> &v) { return ringsnark::amd::flatten(v); }

int main(int argc, char **argv) {
  try {
    int a = 1;
    ringsnark::amd::Params p;
    p.N = atoi(argv[a++]);
    p.L = atoi(argv[a++]);
    for (int i = 0; i < p.L; i++) p.q.push_back(strtoull(argv[a++], nullptr, 10));
    p.N_enc = atoi(argv[a++]);
    p.K = atoi(argv[a++]);
    for (int j = 0; j < p.K; j++) p.Q.push_back(strtoull(argv[a++], nullptr, 10));
    const size_t m = (size_t)atoi(argv[a++]);
    const std::string out = argv[a++];
    const bool poly = a < argc && std::string(argv[a]) == "poly";
    Context::set_context(p);
    Context::seed_prng(20261002);

    // ---- circuit through the reference's gadgetlib: x_{i+2} = (x_i + 3 x_{i+1} + 2) * (x_{i+1} - c x_0), c a ring scalar
    ringsnark::protoboard<R> pb;
    ringsnark::pb_variable_array<R> x(m + 2, ringsnark::pb_variable<R>());
    x.allocate(pb, m + 2, "x");
    pb.set_input_sizes(2);
    // poly: u multiplies x_i (an auxiliary variable from i = 2 on), c multiplies the primary input x_0, k is the constant term
    const R c = poly ? R::random_element() : R(5), u = poly ? R::random_element() : R(1), k = poly ? R::random_element() : R(2);
    for (size_t i = 0; i < m; i++)
      pb.add_r1cs_constraint(ringsnark::r1cs_constraint<R>(u * x[i] + 3 * x[i + 1] + k * ringsnark::variable<R>(0), x[i + 1] - c * x[0], x[i + 2]));
    pb.val(x[0]) = R::random_element();
    pb.val(x[1]) = R::random_element();
    for (size_t i = 0; i < m; i++)
      pb.val(x[i + 2]) = (u * pb.val(x[i]) + R(3) * pb.val(x[i + 1]) + k) * (pb.val(x[i + 1]) - c * pb.val(x[0]));
    if (!pb.is_satisfied()) throw std::runtime_error("reference is_satisfied() rejects a satisfying assignment");
    {
      ringsnark::protoboard<R> bad(pb);
      bad.val(x[m + 1]) += R::one();
      if (bad.is_satisfied()) throw std::runtime_error("reference is_satisfied() accepts a wrong assignment");
    }
    const auto cs = pb.get_constraint_system();
    const auto primary = pb.primary_input();
    const auto aux = pb.auxiliary_input();
    const size_t n_aux = aux.size();

    // ---- keys: real encryptions of random ring elements (the prover never looks inside them)
    auto [pk_enc, sk] = E::keygen();
    (void)pk_enc;
    auto rand_encs = [&](size_t n) {
      std::vector<R> rs(n);
      for (auto &r : rs) r = R::random_element();
      return E::encode(sk, rs);
    };
    Groth16Key gk;
    gk.constraint_system = cs;
    gk.s_pows = rand_encs(m + 1);
    gk.delta_ts = rand_encs(m + 1);
    gk.delta_mid = rand_encs(n_aux);
    gk.alpha = rand_encs(1)[0];
    gk.beta = rand_encs(1)[0];
    const auto gp = ringsnark::amd::groth16::prover(gk, primary, aux);  // the reference's call shape (groth16.tcc:70-73)

    RinocchioKey rk;
    rk.constraint_system = cs;
    rk.s_pows = rand_encs(m + 1);
    rk.alpha_s_pows = rand_encs(m + 1);
    rk.beta_prods = rand_encs(n_aux);
    rk.beta_rv_ts = rand_encs(1)[0];
    rk.beta_rw_ts = rand_encs(1)[0];
    rk.beta_ry_ts = rand_encs(1)[0];
    const R d1 = R::random_invertible_element(), d2 = R::random_invertible_element(), d3 = R::random_invertible_element();
    const auto rkd = ringsnark::amd::rinocchio::proving_key_device::from(rk);
    const auto rp = ringsnark::amd::rinocchio::prover(rkd, primary, aux, &d1, &d2, &d3);

    // homomorphism spot check through the adapters: decode(A) computed two ways is left to the Python side;
    // here: decode(encode(r)) == r
    {
      const R r = R::random_element();
      if (!(E::decode(sk, E::encode(sk, {r})[0]) == r)) throw std::runtime_error("decode(encode(r)) != r");
    }

    // ---- dump
    const ringsnark::amd::R1csCsr csr = ringsnark::amd::export_csr(cs);
    std::ofstream meta(out + "/meta.txt");
    meta << m << " " << csr.n_vars << " " << csr.n_inputs << "\n";
    for (int w = 0; w < 3; w++) meta << csr.col[w].size() << " ";
    meta << "\n";
    for (int w = 0; w < 3; w++) {
      std::vector<uint64_t> rp64(csr.row_ptr[w].begin(), csr.row_ptr[w].end()), col64(csr.col[w].begin(), csr.col[w].end());
      dump(out + "/row_ptr" + std::to_string(w) + ".bin", rp64);
      dump(out + "/col" + std::to_string(w) + ".bin", col64);
      dump(out + "/coeff" + std::to_string(w) + ".bin", csr.coeff[w]);
      std::vector<uint64_t> pi64(csr.poly_idx[w].size());
      for (size_t e = 0; e < pi64.size(); e++) pi64[e] = (uint64_t)(int64_t)csr.poly_idx[w][e];
      dump(out + "/poly_idx" + std::to_string(w) + ".bin", pi64);
    }
    dump(out + "/poly_table.bin", csr.poly_table);
    std::vector<R> full(primary);
    full.insert(full.end(), aux.begin(), aux.end());
    dump(out + "/assignment.bin", ringsnark::amd::flatten(full));
    dump(out + "/g_s_pows.bin", words_of(gk.s_pows));
    dump(out + "/g_delta_ts.bin", words_of(gk.delta_ts));
    dump(out + "/g_delta_mid.bin", words_of(gk.delta_mid));
    dump(out + "/g_alpha.bin", gk.alpha.words());
    dump(out + "/g_beta.bin", gk.beta.words());
    dump(out + "/g_proof.bin", words_of({gp.A, gp.B, gp.C}));
    dump(out + "/r_s_pows.bin", words_of(rk.s_pows));
    dump(out + "/r_alpha_s_pows.bin", words_of(rk.alpha_s_pows));
    dump(out + "/r_beta_prods.bin", words_of(rk.beta_prods));
    dump(out + "/r_beta_rv_ts.bin", rk.beta_rv_ts.words());
    dump(out + "/r_beta_rw_ts.bin", rk.beta_rw_ts.words());
    dump(out + "/r_beta_ry_ts.bin", rk.beta_ry_ts.words());
    dump(out + "/r_d.bin", ringsnark::amd::flatten({d1, d2, d3}));
    dump(out + "/r_proof.bin", words_of({rp.A, rp.A_prime, rp.B, rp.B_prime, rp.C, rp.C_prime, rp.D, rp.D_prime, rp.F}));
    dump(out + "/sk.bin", sk);
    std::cout << "ref_adapter_prove: OK constraints=" << cs.num_constraints() << " variables=" << cs.num_variables()
              << " inputs=" << cs.num_inputs() << std::endl;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "ref_adapter_prove: FAILED: " << e.what() << std::endl;
    return 1;
  }
}

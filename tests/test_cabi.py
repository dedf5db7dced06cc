"""The C-ABI library loads and exports every symbol include/ringsnark_amd.h declares (no compute:
there is no GPU here), and the host logic around it."""
import os
import re

import numpy as np
import pytest

from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ringsnark_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "ringsnark_amd.h")).read()
    declared = set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rs_stream"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rs_version() >= 100


def test_ctx_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ringsnark_amd.device import Device
    with pytest.raises(RuntimeError):
        Device(P.preset("toy"))


def test_presets_are_consistent():
    for name in ("C1", "C2", "C3", "C3F", "C4", "C4F", "C5", "C5s", "toy", "toy49", "toyC3"):
        prm = P.preset(name).validate()
        if name != "C5":  # BFVDefault(2048) is one 54-bit prime: the integer (Montgomery) arithmetic
            assert all(p < (1 << 50) for p in prm.q + prm.Q)
    c3 = P.preset("C3")
    assert (c3.N, c3.L, c3.N_enc, c3.K) == (8192, 4, 8192, 4)
    # q_i = default_double_batching_modulus(8192, 8192) = CoeffModulus::Create(8192, {43, 43, 44, 44}) -- the
    # largest primes of those sizes = 1 mod 2^14, the later-found first (seal_util.hpp:20-32); Q_j the next ones
    assert c3.q == P.coeff_modulus_create(16384, [43, 43, 44, 44]) == [0x7FFFFFC8001, 0x7FFFFFD8001, 0xFFFFFF6C001, 0xFFFFFFFC001]
    assert [P.two_adicity(q) for q in c3.q] == [15, 15, 14, 14] and not set(c3.q) & set(c3.Q)
    assert P.preset("C3R").q == c3.q and P.preset("C3R").name == "C3"  # the name rounds 3-5 used for this preset
    assert c3.max_constraints_fast() >= 1 << 16 and P.preset("C3F").max_constraints_fast() >= 1 << 18
    assert P.preset("C2").q == P.coeff_modulus_create(16384, [36, 36])


def test_chain_and_wide_r1cs_shapes():
    q = P.preset("toy").q
    cs = R.chain_r1cs(5, q)
    assert (cs.m, cs.n_vars, cs.n_inputs, cs.n_aux) == (5, 7, 2, 5)
    assert [int(x) for x in cs.mats["a"][1]] == [1, 2, 3, 4, 5]
    w = R.wide_r1cs(5, q)
    assert w.mats["a"][2].shape == (len(q), 45)
    assert (w.mats["a"][1] == 0).sum() == 5  # one constant-one term per row
    neg = (w.mats["a"][2][0] > q[0] // 2).any()
    assert neg  # negative literals are stored as q - |c|


def test_cpp_adapters_compile():
    """include/ringsnark_amd/ring.hpp (the header-only RingElem / EncodingElem adapters a maintainer
    would template the reference's prover on) must compile as plain C++17 against the C header."""
    import shutil
    import subprocess
    import tempfile
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r"""
#include <ringsnark_amd/ring.hpp>
using namespace ringsnark::amd;
int main() {
  if (false) {  // instantiate, never run (no device in the CPU suite)
    EncodingElem::SecretKey sk;
    std::vector<RingElem> rs(2);
    auto e = EncodingElem::encode(sk, rs, 3);
    RingElem r = EncodingElem::decode(sk, e[0]);
    e[0] += e[1];
    e[0] *= r;
    EncodingElem ip = EncodingElem::inner_product(e.begin(), e.end(), rs.begin(), rs.end());
    (void)ip;
  }
  return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "t.cpp")
        open(f, "w").write(src)
        r = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "include"), f],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_adapter_generator_is_chacha20_keyed_from_the_os(tmp_path):
    """The adapters draw secrets (secret key, the point s, alpha/beta/delta, blinding, noise seeds) from a ChaCha20
    keystream keyed with OS entropy: the block function against the RFC 8439 section 2.3.2 known answer, two
    default-constructed generators differ, the test hook is reproducible."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src = r"""
#define RINGSNARK_AMD_TESTING 1  // the reproducible-key hook exists only under this macro
#include <cstdio>
#include <ringsnark_amd/ring.hpp>
int main() {
  using ringsnark::amd::ChaCha20Rng;
  uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, 0x03020100, 0x07060504, 0x0b0a0908, 0x0f0e0d0c,
                     0x13121110, 0x17161514, 0x1b1a1918, 0x1f1e1d1c, 1, 0x09000000, 0x4a000000, 0}, out[16];
  ChaCha20Rng::block(in, out);
  for (int i = 0; i < 16; i++) std::printf("%08x%c", out[i], i == 15 ? '\n' : ' ');
  ChaCha20Rng a, b, c, d;
  c.seed_for_tests(5);
  d.seed_for_tests(5);
  std::uniform_int_distribution<int> tern(-1, 1);
  int lo = 0, hi = 0;
  for (int i = 0; i < 3000; i++) { int t = tern(a); lo += t == -1; hi += t == 1; }
  std::printf("%d %d %d\n", a() != b(), c() == d() && c() == d(), lo > 800 && hi > 800);
  return 0;
}
"""
    f = tmp_path / "t.cpp"
    f.write_text(src)
    exe = str(tmp_path / "t")
    r = subprocess.run([gxx, "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), str(f), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True).stdout.splitlines()
    assert out[0] == ("e4e7f110 15593bd1 1fdd0f50 c47120a3 c7f4d1c7 0368c033 9aaa2204 4e6cd4c3 "
                      "466482d2 09aa9f07 05d7c214 a2028bd9 d19c12b5 b94e16de e883d0cb 4e3c50a2")
    assert out[1] == "1 1 1"
    hdr = open(os.path.join(ROOT, "include", "ringsnark_amd", "ring.hpp")).read()
    assert "mt19937" not in hdr and "random_device" not in hdr
    # a production build (no RINGSNARK_AMD_TESTING) has no way to make the secrets reproducible, and the process generator is
    # per thread (round-3 advice): the hook must not compile, the generator must be thread_local
    f2 = tmp_path / "t2.cpp"
    f2.write_text("#include <ringsnark_amd/ring.hpp>\nint main() { ringsnark::amd::Context::seed_prng(1); return 0; }\n")
    r2 = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(f2)], capture_output=True, text=True)
    assert r2.returncode != 0 and "seed_prng" in r2.stderr
    assert "static thread_local ChaCha20Rng" in hdr


@pytest.mark.gpu
def test_cpp_adapters_run_against_the_library(tmp_path):
    """tests/cpp/adapter_run.cpp: the RingElem / EncodingElem adapters, compiled with plain g++ and
    linked against librs_hip.so, on the device: ring identities, the reference's error messages and
    the encoding homomorphism through encode / *= / += / inner_product / decode."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "adapter_run")
    libdir = os.path.join(ROOT, "ringsnark_amd")
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "adapter_run.cpp"), "-o", exe, "-L", libdir, "-lrs_hip",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    prm = P.preset("toy")
    args = [str(prm.N), str(prm.L)] + [str(x) for x in prm.q] + [str(prm.N_enc), str(prm.K)] + [str(x) for x in prm.Q]
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "adapter_run: OK" in r.stdout, r.stdout + r.stderr


def test_reference_templates_compile_on_the_adapters():
    """north_star: "keeping the RingElem/EncodingElem operator surface so gadgetlib/relations are untouched".
    The reference's own gadgetlib/, relations/ and zk_proof_systems/r1cs_ppzksnark.hpp (the <RingT, EncT> contract,
    :173-188) must compile, as they lie under /root/reference, with RingT = ringsnark::amd::RingElem and
    EncT = ringsnark::amd::EncodingElem: oracle/ref_adapter_prove.cpp instantiates protoboard, pb_variable_array,
    r1cs_constraint_system (add_constraint, is_satisfied, linear_combination::evaluate), proving_key<R, E> and the
    adapters' provers on them.  (The same file is built into oracle/_ref/ and RUN on the GPU by
    test_reference_gadgetlib_drives_the_device_provers.)"""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None or not os.path.isdir("/root/reference/ringsnark"):
        pytest.skip("needs g++ and /root/reference (build container only)")
    r = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-I", "/root/reference", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "oracle", "ref_adapter_prove.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("poly", [False, True])
def test_reference_gadgetlib_drives_the_device_provers(tmp_path, poly):
    """oracle/_ref/ref_adapter_prove (built in the container from the reference's headers + the C++ adapters + librs_hip.so):
    a circuit assembled with the reference's protoboard, checked by the reference's is_satisfied() on device ring
    arithmetic, proven by ringsnark::amd::groth16::prover and rinocchio::prover FROM C++.  The proofs must equal what
    the ctypes host obtains from the same key and assignment, and the ringGroth16 one must equal the CPU oracle's.
    poly: the circuit's ring coefficients are general ring elements (on the constant one, a primary input and an
    auxiliary variable; benchmarks/bench_ntt_SEAL.cpp:46-53) -- export_csr puts them into the coefficient table."""
    import subprocess

    import numpy as np
    from oracle import oracle as O
    from ringsnark_amd import r1cs as R
    from ringsnark_amd.device import Device, to_host
    from tests import helpers as H
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_adapter_prove")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_adapter_prove not built (needs /root/reference at build time)")
    prm = P.preset("toy")
    m = 6
    args = [str(prm.N), str(prm.L)] + [str(x) for x in prm.q] + [str(prm.N_enc), str(prm.K)] + [str(x) for x in prm.Q] + [str(m), str(tmp_path)] + (["poly"] if poly else [])
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ref_adapter_prove: OK" in r.stdout, r.stdout + r.stderr
    rd = lambda name, shape=None: (lambda a: a if shape is None else a.reshape(shape))(np.fromfile(str(tmp_path / name), dtype=np.uint64))
    mm, n_vars, n_inputs = [int(v) for v in open(str(tmp_path / "meta.txt")).read().split()[:3]]
    assert (mm, n_vars, n_inputs) == (m, m + 2, 2)
    mats = {}
    for w, name in enumerate("abc"):
        col = rd("col%d.bin" % w).astype(np.uint32)
        mats[name] = (rd("row_ptr%d.bin" % w).astype(np.uint32), col, rd("coeff%d.bin" % w, (prm.L, col.shape[0])))
    # the CSR export holds what the reference's gadgetlib produced: a-rows have 3 terms (x_i, 3 x_{i+1}, the constant 2)
    assert mats["a"][1].shape[0] == 3 * m and (mats["a"][1] == 0).sum() == m
    pidx = {name: rd("poly_idx%d.bin" % w).astype(np.int64).astype(np.int32) for w, name in enumerate("abc")}
    table = rd("poly_table.bin")
    if poly:  # u (on x_i), c (on x_0) and k (on the constant one): three distinct table rows, shared by the m constraints
        table = table.reshape((-1, prm.L, prm.N))
        assert table.shape[0] == 3 and (pidx["a"] >= 0).sum() == 2 * m and (pidx["b"] >= 0).sum() == m and (pidx["c"] >= 0).sum() == 0
        cs = R.R1CS(m, n_vars, n_inputs, mats, pidx, table)
    else:
        assert table.size == 0 and all((p == -1).all() for p in pidx.values())
        cs = R.R1CS(m, n_vars, n_inputs, mats)
    dev = Device(prm)
    ctx = H.oracle_ctx(prm)
    asg = rd("assignment.bin", (n_vars,) + ctx.ring_shape())
    enc = lambda name, n=None: rd(name, ctx.enc_shape() if n is None else ctx.enc_shape(n))
    gpk = dict(s_pows=enc("g_s_pows.bin", m + 1), delta_ts=enc("g_delta_ts.bin", m + 1), delta_mid=enc("g_delta_mid.bin", m),
               alpha=enc("g_alpha.bin"), beta=enc("g_beta.bin"))
    dcs = dev.r1cs(cs)
    got, _ = dev.groth16_prove(dcs, {k: dev.put(v) for k, v in gpk.items()}, dev.put(asg))
    cpp = enc("g_proof.bin", 3)
    assert (to_host(got) == cpp).all(), "C++ adapter prover and ctypes host disagree"
    exp, _ = O.groth16_prove(ctx, H.oracle_cs(cs), gpk, asg)
    assert (cpp == exp).all(), "C++ adapter prover differs from the CPU oracle"
    rpk = dict(s_pows=enc("r_s_pows.bin", m + 1), alpha_s_pows=enc("r_alpha_s_pows.bin", m + 1), beta_prods=enc("r_beta_prods.bin", m),
               beta_rv_ts=enc("r_beta_rv_ts.bin"), beta_rw_ts=enc("r_beta_rw_ts.bin"), beta_ry_ts=enc("r_beta_ry_ts.bin"))
    ds = rd("r_d.bin", ctx.ring_shape(3))
    got, _ = dev.rinocchio_prove(dcs, {k: dev.put(v) for k, v in rpk.items()}, dev.put(asg), *[dev.put(d) for d in ds])
    cppr = enc("r_proof.bin", 9)
    assert (to_host(got) == cppr).all()
    expr, _ = O.rinocchio_prove(ctx, H.oracle_cs(cs), rpk, asg, *ds)
    assert (cppr == expr).all()


def test_bench_launches_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) must start the two ranks itself
    (round-4 verdict: the driver's command form died on an assert) and, when they fail -- here: no GPU -- exit non-zero
    without hanging or retrying.  On the GPU box the same command is rehearsed with RINGSNARK_BENCH_REHEARSAL=1
    (profiles/r05_bench_rehearsal_gpus2.json)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher's failure path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "No HIP GPUs are available" in (r.stdout + r.stderr) or "HIP device" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-2000:]
    assert '"metric"' not in r.stdout

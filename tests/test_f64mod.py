"""Exactness of the FP64 modular arithmetic used by every kernel (ringsnark_amd/csrc/f64mod.hpp),
checked on the host against 128-bit integers for every prime of the presets."""
import os
import subprocess
import tempfile

from ringsnark_amd import params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_f64mod_exact_against_int128():
    primes = set()
    for name in ("toy", "toy49", "C2", "C3", "C4"):
        prm = P.preset(name)
        primes.update(prm.q + prm.Q)
    primes.add((1 << 50) - 27)  # largest prime below 2^50: the documented limit
    assert P.is_prime((1 << 50) - 27)
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "f64mod_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tests", "f64mod_check.cpp"), "-o", exe])
        out = subprocess.run([exe] + [str(p) for p in sorted(primes)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == "ok"

"""EncodingElem::encode / ::decode on the device, and the end-to-end
semantic check they make possible: a proof produced by the HIP prover under a REAL (encrypted)
proving key satisfies the reference's verification equation (groth16.tcc:117-170)."""
import numpy as np
import pytest

from oracle import oracle as O
from ringsnark_amd import params as P
from ringsnark_amd import r1cs as R
from tests import helpers as H
from tests import snark_ref as S


def test_oracle_generator_verifier_accept_and_reject():
    """CPU only: the restated generator / verifier around the ORACLE prover (pins tests/snark_ref.py)."""
    prm = P.preset("toy")
    ctx = H.oracle_ctx(prm)
    cs = R.chain_r1cs(6, prm.q)
    asg = H.make_assignment(ctx, cs)
    pk, vk = S.groth16_generator(ctx, cs, 5, ctx.enc_encode)
    proof, empty = O.groth16_prove(ctx, H.oracle_cs(cs), pk, asg)
    assert empty == [0, 0, 0]
    A, B, C = (ctx.enc_decode(vk["sk"], proof[k]) for k in range(3))
    assert S.groth16_verifier(ctx, cs, vk, asg[: cs.n_inputs], A, B, C)
    bad = asg[: cs.n_inputs].copy()
    bad[0, 0, 0] = (int(bad[0, 0, 0]) + 1) % prm.q[0]
    assert not S.groth16_verifier(ctx, cs, vk, bad, A, B, C)
    C2 = C.copy()
    C2[1, 3] = (int(C2[1, 3]) + 1) % prm.q[1]
    assert not S.groth16_verifier(ctx, cs, vk, asg[: cs.n_inputs], A, B, C2)


def test_oracle_rinocchio_generator_verifier_accept_and_reject():
    """CPU only: Rinocchio's six verifier checks (rinocchio.tcc:192-300) around the ORACLE prover."""
    prm = P.preset("toy")
    ctx = H.oracle_ctx(prm)
    cs = R.chain_r1cs(5, prm.q)
    asg = H.make_assignment(ctx, cs)
    pk, vk = S.rinocchio_generator(ctx, cs, 8, ctx.enc_encode)
    proof, empty = O.rinocchio_prove(ctx, H.oracle_cs(cs), pk, asg)
    assert empty == [0] * 9
    decs = [ctx.enc_decode(vk["sk"], proof[k]) for k in range(9)]
    ok, checks = S.rinocchio_verifier(ctx, cs, vk, asg[: cs.n_inputs], decs)
    assert ok, checks
    bad = asg[: cs.n_inputs].copy()
    bad[0, 1, 2] = (int(bad[0, 1, 2]) + 1) % prm.q[1]
    ok, checks = S.rinocchio_verifier(ctx, cs, vk, bad, decs)
    assert not ok and not checks["P = H Z(s)"] and checks["L_beta = L"]


def _budget_ladder(ctx, sk, seed=3):
    """A fresh encoding multiplied by a random ring element again and again: the invariant noise budget falls by about
    log2(t) + log2(N_enc) bits a time until it is spent."""
    r = ctx.random_ring(seed, 2)
    cur = ctx.enc_encode(sk, r[:1], 5)[0]
    out = [cur]
    for _ in range(16):  # as long as it takes (C5: eight 48-bit data primes under a 54-bit plain modulus), one step beyond
        cur = ctx.enc_mul_ring(cur, r[1])
        out.append(cur)
        if len(out) >= 5 and max(ctx.noise_budget(sk, out[-2])) == 0:
            break
    return np.stack(out), r


@pytest.mark.parametrize("name", ["toy", "toy49", "toy60", "C2"])
def test_oracle_noise_budget_and_guard(name):
    """CPU: Decryptor::invariant_noise_budget restated (SEAL 4.x, bgv: || c0 + c1 s mod Q ||_inf centred, no scaling by t)
    and the guard of EncodingElem::decode (seal_ring.tcc:443-454).  Fresh: bit_count(Q) - log2(t |e| N-ish) - 1; every
    plaintext product costs about log2(t) + log2(N_enc)/2.. bits; decoding is exact while budget > 0, and the guard fires
    exactly when it is 0; a uniformly random "ciphertext" has none."""
    prm = P.preset(name)
    ctx = H.oracle_ctx(prm)
    sk = ctx.keygen(9)
    ladder, r = _budget_ladder(ctx, sk)
    logQ = sum(int(p).bit_length() for p in prm.Q)  # within a bit of bit_count(prod Q)
    want = r[0]
    prev = None
    spent = False
    for k, e in enumerate(ladder):
        b = ctx.noise_budget(sk, e)
        assert all(0 <= x < logQ for x in b)
        if k == 0:  # fresh: noise = m + t e, |.| < t (1 + |e|) <= 2 t
            for i in range(prm.L):
                assert logQ - int(prm.q[i]).bit_length() - 4 <= b[i] <= logQ - int(prm.q[i]).bit_length()
        if prev is not None:
            assert all(x <= max(0, p_ - 1) for x, p_ in zip(b, prev)), (k, prev, b)
        prev = b
        ring, bad = ctx.enc_decode_checked(sk, e)
        if min(b) > 0:
            assert bad == -1 and (ring == want).all()   # budget left => the decryption is the product
        else:
            assert bad == b.index(0) and ring is None    # "ciphertext #i has remaining noise budget 0 <= 0"
            spent = True
        want = ctx.ring_mul(want, r[1])
    assert spent, "the ladder is long enough to spend the budget of every preset here"
    assert ctx.noise_budget(sk, ctx.random_enc(4)) == [0] * prm.L


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["toy", "toy49", "C2", "toy54", "toy60", "C5"])  # toy54 / toy60 / C5 (ring side): integer arithmetic
def test_noise_budget_and_decode_guard_match_oracle(name):
    """rs_enc_noise_budget equals the oracle's budget for every ciphertext of the ladder (both arithmetics; the device
    finds the bit length by digit comparisons in mixed radix, the oracle composes multi-word integers), and rs_enc_decode
    refuses what the reference refuses, with its message (seal_ring.tcc:446-454): a proof whose encodings were multiplied
    past their budget must not decode to garbage silently (round-4 verdict, "What's missing" 2)."""
    from ringsnark_amd import _lib
    from ringsnark_amd.device import Device, to_host
    prm = P.preset(name)
    dev, ctx = Device(prm), H.oracle_ctx(prm)
    sk = ctx.keygen(9)
    ladder, r = _budget_ladder(ctx, sk)
    exp = np.array([ctx.noise_budget(sk, e) for e in ladder])
    dsk, dl = dev.put(sk), dev.put(ladder)
    got = dev.enc_noise_budget(dsk, dl)
    assert (got == exp).all(), (got, exp)
    assert exp[0].min() > 0 and exp[-1].max() == 0
    ok = [k for k in range(len(ladder)) if exp[k].min() > 0]
    dec = to_host(dev.enc_decode(dsk, dl[:len(ok)]))  # the budget only falls: the decodable ones come first
    want = r[0]
    for k in ok:
        assert (dec[k] == want).all()
        want = ctx.ring_mul(want, r[1])
    first_bad = len(ok)
    with pytest.raises(_lib.RsError) as ei:
        dev.enc_decode(dsk, dl)
    assert ei.value.code == _lib.RS_ERR_NOISE
    limb = int(np.flatnonzero(exp[first_bad] == 0)[0])
    assert "ciphertext #%d has remaining noise budget 0 <= 0" % limb in str(ei.value) and "element %d" % first_bad in str(ei.value)
    # a single spent element, as the verifier sees one
    _, bad = ctx.enc_decode_checked(sk, ladder[-1])
    with pytest.raises(_lib.RsError) as ei:
        dev.enc_decode(dsk, dl[-1])
    assert ei.value.code == _lib.RS_ERR_NOISE and "ciphertext #%d " % bad in str(ei.value)
    assert str(ei.value).endswith("has remaining noise budget 0 <= 0")  # one element: exactly the reference's text (seal_ring.tcc:450-453)
    # uniformly random residues are no ciphertext of anything
    assert (dev.enc_noise_budget(dsk, dev.put(ctx.random_enc(4, 2))) == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["toy", "toy49", "C2", "toy54", "toy60"])  # toy54 / toy60: the integer (Montgomery) arithmetic
def test_enc_encode_decode_match_oracle(name):
    from ringsnark_amd.device import Device, to_host
    prm = P.preset(name)
    dev, ctx = Device(prm), H.oracle_ctx(prm)
    sk = ctx.keygen(3)
    rings = ctx.random_ring(41, 3)
    rings[1] = 0
    exp = ctx.enc_encode(sk, rings, 9)
    dsk = dev.put(sk)
    got = dev.enc_encode(dsk, dev.put(rings), 9)
    assert (to_host(got) == exp).all()
    # decode: fresh encodings, and products / sums of them (noise grows, centring matters)
    dec = to_host(dev.enc_decode(dsk, got))
    assert (dec == rings).all()
    r2 = ctx.random_ring(42, 3)
    ip, used = ctx.inner_product(exp, r2)
    assert used == 3
    d = to_host(dev.enc_decode(dsk, dev.put(ip)))
    assert (d == ctx.enc_decode(sk, ip)).all()
    want = ctx.ring_add(ctx.ring_mul(rings[0], r2[0]), ctx.ring_mul(rings[2], r2[2]))
    assert (d == want).all()  # homomorphism: decode(<E(a), b>) = <a, b>


@pytest.mark.gpu
@pytest.mark.parametrize("name,m", [("toy", 6), ("toy49", 12)])
def test_device_proof_verifies_under_reference_equation(name, m):
    """generator (instance map with evaluation and encryption on the DEVICE) -> device prover ->
    device decode -> the reference's verification equation; then a tampered input must be rejected."""
    from ringsnark_amd.device import Device, to_host
    prm = P.preset(name)
    dev, ctx = Device(prm), H.oracle_ctx(prm)
    cs = R.chain_r1cs(m, prm.q)
    asg = H.make_assignment(ctx, cs)

    def encode_on_device(sk, rings, seed):
        return to_host(dev.enc_encode(dev.put(sk), dev.put(rings), seed))

    def imap_on_device(s):
        return [to_host(x) for x in dev.instance_map_eval(dev.r1cs(cs), dev.put(s))]

    pk, vk = S.groth16_generator(ctx, cs, 21, encode_on_device, imap_on_device)
    got, empty = dev.groth16_prove(dev.r1cs(cs), {k: dev.put(v) for k, v in pk.items()}, dev.put(asg))
    assert [int(e) for e in empty] == [0, 0, 0]
    dec = to_host(dev.enc_decode(dev.put(vk["sk"]), got))
    assert S.groth16_verifier(ctx, cs, vk, asg[: cs.n_inputs], dec[0], dec[1], dec[2])
    bad = asg[: cs.n_inputs].copy()
    bad[1, 0, 5] = (int(bad[1, 0, 5]) + 1) % prm.q[0]
    assert not S.groth16_verifier(ctx, cs, vk, bad, dec[0], dec[1], dec[2])


@pytest.mark.gpu
@pytest.mark.parametrize("aux_only", [False, True])
def test_device_proof_with_polynomial_coefficients_verifies(aux_only):
    """A circuit whose coefficients are general ring elements (relations/variable.tcc:246-254; the DFT constraint of
    benchmarks/bench_ntt_SEAL.cpp:46-53), end to end on the device: instance map with evaluation, encryption, the
    ringGroth16 prover, decryption, the reference's verification equation; a tampered input is rejected."""
    from ringsnark_amd.device import Device, to_host
    prm = P.preset("toy49")
    dev, ctx = Device(prm), H.oracle_ctx(prm)
    m = 10
    cs = R.wide_poly_r1cs(m, prm.q, prm.N, aux_only=aux_only, constants=False)
    asg = H.make_assignment(ctx, cs)
    dcs = dev.r1cs(cs)
    pk, vk = S.groth16_generator(ctx, cs, 23, lambda sk, r, seed: to_host(dev.enc_encode(dev.put(sk), dev.put(r), seed)),
                                 lambda s: [to_host(x) for x in dev.instance_map_eval(dcs, dev.put(s))])
    got, empty = dev.groth16_prove(dcs, {k: dev.put(v) for k, v in pk.items()}, dev.put(asg))
    assert [int(e) for e in empty] == [0, 0, 0]
    dec = to_host(dev.enc_decode(dev.put(vk["sk"]), got))
    assert S.groth16_verifier(ctx, cs, vk, asg[: cs.n_inputs], dec[0], dec[1], dec[2])
    bad = asg[: cs.n_inputs].copy()
    bad[0, 1, 7] = (int(bad[0, 1, 7]) + 1) % prm.q[1]
    assert not S.groth16_verifier(ctx, cs, vk, bad, dec[0], dec[1], dec[2])


@pytest.mark.gpu
def test_device_rinocchio_proof_passes_reference_checks():
    from ringsnark_amd.device import Device, to_host
    prm = P.preset("toy49")
    dev, ctx = Device(prm), H.oracle_ctx(prm)
    cs = R.chain_r1cs(9, prm.q)
    asg = H.make_assignment(ctx, cs)
    pk, vk = S.rinocchio_generator(ctx, cs, 31, lambda sk, r, seed: to_host(dev.enc_encode(dev.put(sk), dev.put(r), seed)))
    got, empty = dev.rinocchio_prove(dev.r1cs(cs), {k: dev.put(v) for k, v in pk.items()}, dev.put(asg))
    assert [int(e) for e in empty] == [0] * 9
    # the proof travels through the wire format on its way to the verifier
    back, em = dev.enc_deserialize(dev.enc_serialize(got, empty=empty))
    dec = to_host(dev.enc_decode(dev.put(vk["sk"]), back))
    ok, checks = S.rinocchio_verifier(ctx, cs, vk, asg[: cs.n_inputs], [dec[k] for k in range(9)])
    assert ok, checks


@pytest.mark.gpu
def test_wire_format_roundtrip_and_validation():
    """proofs / key vectors through the wire format of include/ringsnark_amd.h."""
    import struct
    from ringsnark_amd import _lib
    from ringsnark_amd.device import Device, to_host
    prm = P.preset("toy")
    dev, ctx = Device(prm), H.oracle_ctx(prm)
    enc = ctx.random_enc(5, 3)
    data = dev.enc_serialize(dev.put(enc), empty=[0, 1, 0])
    # header, field by field
    assert data[:8] == b"RSNKENC1"
    assert struct.unpack_from("<4I", data, 8) == (prm.N, prm.L, prm.N_enc, prm.K)
    off = 24
    assert list(struct.unpack_from("<%dQ" % prm.L, data, off)) == [int(x) for x in prm.q]
    off += 8 * prm.L
    assert list(struct.unpack_from("<%dQ" % prm.K, data, off)) == [int(x) for x in prm.Q]
    off += 8 * prm.K
    assert struct.unpack_from("<Q", data, off)[0] == 3 and data[off + 8: off + 11] == bytes([0, 1, 0])
    hb = (off + 8 + 3 + 7) // 8 * 8
    payload = np.frombuffer(data, dtype="<u8", offset=hb).reshape(enc.shape)
    assert (payload[0] == enc[0]).all() and not payload[1].any() and (payload[2] == enc[2]).all()
    back, empty = dev.enc_deserialize(data)
    assert list(empty) == [0, 1, 0]
    b = to_host(back)
    assert (b[0] == enc[0]).all() and not b[1].any() and (b[2] == enc[2]).all()
    # untrusted input: every corruption is refused with RS_ERR_INVALID
    for mutate in ("magic", "dims", "modulus", "truncate", "residue", "flag", "count_wraps", "count_huge", "padding",
                   "empty_payload"):
        bad = bytearray(data)
        if mutate == "magic":
            bad[0] ^= 1
        elif mutate == "dims":
            struct.pack_into("<I", bad, 8, prm.N * 2)
        elif mutate == "modulus":
            struct.pack_into("<Q", bad, 24, int(prm.q[0]) + 2)
        elif mutate == "truncate":
            bad = bad[:-8]
        elif mutate == "residue":
            struct.pack_into("<Q", bad, hb, int(prm.Q[0]))
        elif mutate == "flag":
            bad[off + 8] = 7
        elif mutate == "count_wraps":
            # a count chosen so that header(count) + count * enc_bytes wraps around 2^64 to
            # something <= len(buf): (1 + enc_bytes) is odd, hence invertible mod 2^64
            step = 1 + prm.enc_words * 8
            fixed = off + 8
            cnt = (pow(step, -1, 1 << 64) * ((len(bad) - fixed - 64) % (1 << 64))) % (1 << 64)
            assert cnt > 3 and (fixed + cnt * step) % (1 << 64) <= len(bad)
            struct.pack_into("<Q", bad, off, cnt)
        elif mutate == "count_huge":
            struct.pack_into("<Q", bad, off, (1 << 64) - 1)
        elif mutate == "padding":
            bad[hb - 1] = 1  # header padding byte (3 flag bytes -> 5 bytes of padding)
        else:  # the EMPTY element's payload must be zero
            struct.pack_into("<Q", bad, hb + prm.enc_words * 8, 1)
        with pytest.raises(_lib.RsError):
            dev.enc_deserialize(bytes(bad))
        # the size-query path must refuse a corrupt header as well (it used to return a garbage count)
        if mutate in ("count_wraps", "count_huge", "flag", "padding", "truncate"):
            import ctypes as C
            buf = np.frombuffer(bytes(bad), dtype=np.uint8)
            cnt_out = C.c_size_t(12345)
            rc = dev.lib.rs_enc_deserialize(dev.h, buf.ctypes.data_as(C.c_void_p), buf.size, None, None, 0, C.byref(cnt_out), None)
            assert rc != 0 and cnt_out.value == 12345, mutate
    # a different context refuses the stream
    dev49 = Device(P.preset("toy49"))
    with pytest.raises(_lib.RsError):
        dev49.enc_deserialize(data)


@pytest.mark.gpu
@pytest.mark.parametrize("name,m,kind", [("toy", 6, "chain"), ("toy49", 11, "wide"), ("toy54", 9, "wide"), ("toy60", 7, "chain"),
                                         ("toy", 9, "wide_poly"), ("toy60", 7, "wide_poly")])
def test_instance_map_with_evaluation_matches_restatement(name, m, kind):
    """At/Bt/Ct/Ht/Zt on the device against the O(m^2) restatement of
    r1cs_to_qrp.tcc:76-116 (tests/snark_ref.py), bit for bit; only a point that IS a domain element is refused."""
    from ringsnark_amd import _lib
    from ringsnark_amd.device import Device, to_host
    prm = P.preset(name)
    dev, ctx = Device(prm), H.oracle_ctx(prm)
    cs = {"chain": lambda: R.chain_r1cs(m, prm.q), "wide": lambda: R.wide_r1cs(m, prm.q),
          "wide_poly": lambda: R.wide_poly_r1cs(m, prm.q, prm.N)}[kind]()  # wide_poly: coefficients that are general ring elements
    Rg = S.Ring(ctx)
    s = Rg.random_exceptional(np.random.RandomState(4), m)
    At, Bt, Ct, Ht, Zt = S.instance_map_with_evaluation(Rg, cs, s)
    got = dev.instance_map_eval(dev.r1cs(cs), dev.put(s))
    for g, e in zip(got[:4], (At, Bt, Ct, Ht)):
        assert (to_host(g) == np.stack(e)).all()
    assert (to_host(got[4]) == Zt).all()
    # s hitting a node in SOME slots is legal in the reference (evaluation_domain.tcc:24-39
    # only rejects s == domain element as a ring element, and computes products without dividing)
    hit = s.copy()
    hit[prm.L - 1, 3] = m - 1  # the last limb (toy54 has one)
    hit[0, 5] = 0
    e = S.instance_map_with_evaluation(Rg, cs, hit)
    got = dev.instance_map_eval(dev.r1cs(cs), dev.put(hit))
    for g, x in zip(got[:4], e[:4]):
        assert (to_host(g) == np.stack(x)).all()
    assert (to_host(got[4]) == e[4]).all()
    bad = np.full_like(s, m - 1)  # the ring element RingT(m-1): a domain element
    with pytest.raises(_lib.RsError) as ei:
        dev.instance_map_eval(dev.r1cs(cs), dev.put(bad))
    assert "t cannot be one of the values in the domain" in str(ei.value)


@pytest.mark.gpu
def test_new_entry_points_reject_bad_arguments():
    """Error behaviour of the 8(f) entry points: status codes, never a crash."""
    import ctypes as C
    from ringsnark_amd import _lib
    from ringsnark_amd.device import Device, _ptr
    dev = Device(P.preset("toy"))
    lib = dev.lib
    enc = dev.enc_empty(2)
    buf = (C.c_uint8 * 16)()
    assert lib.rs_enc_wire_size(None, 3) == 0
    assert lib.rs_enc_serialize(dev.h, _ptr(enc), None, 2, buf, 16, None) == _lib.RS_ERR_INVALID
    assert b"buffer too small" in lib.rs_last_error()
    assert lib.rs_enc_decode(dev.h, None, _ptr(enc), 2, _ptr(dev.ring_empty(2)), None) == _lib.RS_ERR_INVALID
    assert lib.rs_enc_encode(dev.h, None, None, 1, 0, None, None) == _lib.RS_ERR_INVALID
    cnt = C.c_size_t(0)
    assert lib.rs_enc_deserialize(dev.h, buf, 16, None, None, 0, C.byref(cnt), None) == _lib.RS_ERR_INVALID
    assert lib.rs_instance_map_eval(dev.h, None, None, None, None, None, None, None, None) == _lib.RS_ERR_INVALID
    # count == 0 is a no-op
    assert lib.rs_enc_decode(dev.h, _ptr(enc), _ptr(enc), 0, _ptr(dev.ring_empty()), None) == _lib.RS_OK
    # the noise budget: null arguments, a null context, count == 0
    out = (C.c_int * 4)()
    assert lib.rs_enc_noise_budget(dev.h, None, _ptr(enc), 2, out, None) == _lib.RS_ERR_INVALID
    assert lib.rs_enc_noise_budget(dev.h, _ptr(enc), _ptr(enc), 2, None, None) == _lib.RS_ERR_INVALID
    assert lib.rs_enc_noise_budget(None, _ptr(enc), _ptr(enc), 2, out, None) == _lib.RS_ERR_INVALID
    assert lib.rs_enc_noise_budget(dev.h, _ptr(enc), _ptr(enc), 0, out, None) == _lib.RS_OK
    assert lib.rs_version() >= 101

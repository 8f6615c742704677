"""ecgfp5 crate: native group arithmetic against the affine chord-and-tangent oracle, the reference's round-trip tests,
and its three circuits through the oracle prover + host verifier (CPU) and bit-exact on the GPU (-m gpu).

pod2 (which owns the curve) is not in the reference tree, so nothing here is pinned to its bytes; what is pinned is the
published curve: group order prime, n * G = neutral, and the product's complete projective formulas agreeing with
textbook affine arithmetic (oracle/oracle_ecgfp5.py)."""
import os
import random
import sys

import pytest

import circuits

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_ecgfp5 as ec  # noqa: E402

P = 0xFFFFFFFF00000001


def _miller_rabin(n, rounds=24):
    r = random.Random(7)
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for _ in range(rounds):
        x = pow(r.randrange(2, n - 1), d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def test_oracle_curve_constants():
    n = ec.GROUP_ORDER
    assert n.bit_length() == 319 and _miller_rabin(n)
    # Hasse: |E| = 2n within 2 sqrt(q) of q + 1
    q = P**5
    assert abs(2 * n - (q + 1)) <= 2 * int(q**0.5) + 2
    G = ec.generator()
    assert G[1] == (pow(4, P - 2, P), 0, 0, 0, 0) and ec.in_group(G) and ec.on_curve(ec.to_curve(G))
    assert ec.g_mul(n, G) == ec.NEUTRAL
    assert ec.g_mul(n - 1, G) == ec.g_neg(G)
    # field helpers against plain exponentiation
    r = random.Random(1)
    a = tuple(r.randrange(P) for _ in range(5))
    assert ec.f_frob(a) == ec.f_pow(a, P)
    assert ec.f_inv(a) == ec.f_pow(a, q - 2)
    sq = ec.f_mul(a, a)
    assert ec.f_mul(ec.f_sqrt(sq), ec.f_sqrt(sq)) == sq


def test_native_group_law_against_oracle(pkg):
    E = pkg.ecgfp5
    assert E.group_order() == ec.GROUP_ORDER
    G = E.generator()
    assert G == ec.generator()
    r = random.Random(5)
    assert E.mul(0, G) == ec.NEUTRAL and E.mul(ec.GROUP_ORDER, G) == ec.NEUTRAL
    assert E.add(G, ec.NEUTRAL) == G and E.add(ec.NEUTRAL, ec.NEUTRAL) == ec.NEUTRAL
    for _ in range(6):
        k, k2 = r.randrange(ec.GROUP_ORDER), r.randrange(ec.GROUP_ORDER)
        p1 = E.mul(k, G)
        assert p1 == ec.g_mul(k, G)
        p2 = E.mul(k2, p1)
        assert p2 == ec.g_mul(k2, p1) == E.mul(k * k2 % ec.GROUP_ORDER, G)
        assert E.add(p1, p2) == ec.g_add(p1, p2)
        assert E.add(p1, p1) == ec.g_add(p1, p1) == E.mul(2, p1)
        assert E.add(p1, E.neg(p1)) == ec.NEUTRAL
        assert E.is_in_subgroup(p1) and ec.in_group(p1)
        assert E.decompress_into_subgroup(E.compress_from_subgroup(p1)) == p1 == ec.decompress(p1[1])
        # the other coset (n-torsion points: p1 + N on the curve) has the same u up to sign but a square x: rejected
        x, y = ec.e_add(ec.to_curve(p1), ec.N_PT)
        other = (x, ec.f_mul(x, ec.f_inv(y)))
        assert not E.is_in_subgroup(other) and not ec.in_group(other)
    # a u that is not the coordinate of a group element
    bad = next(w for w in ((i, 1, 0, 0, 0) for i in range(1, 64)) if ec.decompress(w) is None)
    with pytest.raises(pkg.P2Error):
        E.decompress_into_subgroup(bad)


def test_msg_encoding_roundtrip(pkg):
    # ecgfp5/src/lib.rs:101-112
    E = pkg.ecgfp5
    r = random.Random(11)
    for i in range(25):
        x = r.getrandbits(160) if i else (1 << 160) - 1
        pt = E.encode_binary(x, 1000 + i)
        assert ec.in_group(pt)
        assert all((pt[1][j] & 0xFFFFFFFF) == (x >> (32 * j)) & 0xFFFFFFFF for j in range(5))
        assert E.decode_binary(pt) == x


def test_elgamal_round_trip(pkg):
    # ecgfp5/src/elgamal.rs:33-46 and hashed_elgamal.rs:49-63
    E = pkg.ecgfp5
    r = random.Random(13)
    for i in range(10):
        sk = pkg.ECGFP5SecretKey.rand(i)
        pk = sk.public_key()
        assert pk == ec.g_mul(sk.value, ec.generator())
        msg = E.new_rand_from_subgroup(50 + i)
        nonce = E.random_scalar(100 + i)
        ct = E.elgamal_encrypt(pk, nonce, msg)
        assert ct == ec.elgamal_encrypt(pk, nonce, msg)
        assert E.elgamal_decrypt(sk, ct) == msg == ec.elgamal_decrypt(sk.value, ct)
        m5 = tuple(r.randrange(P) for _ in range(5))
        hct = E.hashed_elgamal_encrypt(pk, nonce, m5)
        h = pkg.poseidon_native.hash_n_to_m_no_pad(ec.as_fields(ec.g_mul(nonce, pk)), 5)
        assert hct == (ct[0], tuple((m5[j] + h[j]) % P for j in range(5)))
        assert E.hashed_elgamal_decrypt(sk, hct) == m5
    with pytest.raises(pkg.P2Error):
        E.elgamal_encrypt(pk, ec.GROUP_ORDER, msg)  # elgamal.rs:12 assert!(nonce < &GROUP_ORDER)


def test_poseidon_cipher_with_curve_keys(pkg, orc):
    # poseidon-cipher/src/lib.rs:141-156 and circuit.rs:156-190: the key really is a group element (new_key, expanded_key)
    from test_host_logic import _prove_verify
    r = random.Random(17)
    k, K = pkg.poseidon_native.new_key(1)
    assert K == ec.g_mul(k, ec.generator())
    ks = pkg.poseidon_native.expanded_key(K, 2)
    assert ec.in_group(ks)
    nonce = [r.randrange(P), r.randrange(P)]
    for n in (9, 10, 11):
        msg = [tuple(r.randrange(P) for _ in range(5)) for _ in range(n)]
        ct = pkg.poseidon_native.encrypt(ks, msg, nonce)
        assert pkg.poseidon_native.decrypt(ks, ct, nonce, n) == msg and msg != ct[:n]
    b = pkg.CircuitBuilder()
    t = pkg.PoseidonEncryptTarget.build(b, 3)
    data = b.build()
    msg = [tuple(r.randrange(P) for _ in range(5)) for _ in range(3)]
    pw = pkg.PartialWitness()
    t.set_targets(pw, list(ks), msg, nonce, pkg.poseidon_native.encrypt(ks, msg, nonce))
    _prove_verify(pkg, orc, data, [pw])


def _flip(point):
    return (point[0], (point[1][0] ^ 1,) + tuple(point[1][1:]))


def test_public_key_circuit(pkg, orc):
    from test_host_logic import _prove_verify
    data, pws, (sk_t, pk_t), cases = circuits.ecgfp5_public_key(pkg, [1, 2])
    assert data.info["degree_bits"] == 13 and data.info["num_luts"] == 0
    oc, vd, res = _prove_verify(pkg, orc, data, pws)
    sk, pk = cases[0]
    pw = pkg.PartialWitness()
    pw.set_secret_key_target(sk_t, sk)
    pw.set_point_target(pk_t, _flip(pk))
    assert oc.prove(pw.map)[0] == 1
    # a "bit" that is not a bit is rejected by the boolean constraint
    pw = pkg.PartialWitness()
    pw.set_target_arr(sk_t, [2] + [0] * 319)
    assert oc.prove(pw.map)[0] == 1
    # the public key is computed, not only checked: leave it unset, read it back from the witness
    pw = pkg.PartialWitness()
    pw.set_secret_key_target(sk_t, sk)
    st, proof = oc.prove(pw.map)
    assert st == 0
    data.verify(proof, vd)


def test_elgamal_circuits(pkg, orc):
    from test_host_logic import _prove_verify
    data, pws, (pk_t, nonce_t, msg_t, ct_t), cases = circuits.ecgfp5_elgamal(pkg, [1])
    assert data.info["degree_bits"] == 14
    oc, vd, res = _prove_verify(pkg, orc, data, pws)
    sk, pk, msg, nonce, ct = cases[0]
    for bad0, bad1 in ((_flip(ct[0]), ct[1]), (ct[0], _flip(ct[1]))):
        pw = pkg.PartialWitness()
        pw.set_point_target(pk_t, pk)
        pw.set_point_target(msg_t, msg)
        pw.set_biguint320_target(nonce_t, nonce)
        pw.set_point_target(ct_t[0], bad0)
        pw.set_point_target(ct_t[1], bad1)
        assert oc.prove(pw.map)[0] == 1
    # a public key that is not on the curve fails add_virtual_point_target's membership check
    pw = pkg.PartialWitness()
    pw.set_point_target(pk_t, _flip(pk))
    pw.set_point_target(msg_t, msg)
    pw.set_biguint320_target(nonce_t, nonce)
    assert oc.prove(pw.map)[0] == 1

    data, pws, _, cases = circuits.ecgfp5_hashed_elgamal(pkg, [2])
    _prove_verify(pkg, orc, data, pws)


@pytest.mark.gpu
def test_ecgfp5_circuits_bit_exact_on_gpu(pkg, orc):
    """BASELINE.json configs[3] shape (ElGamal): GPU proofs byte-identical to the oracle's; wrong ciphertexts fail."""
    from test_gpu_parity import _gpu_vs_oracle
    if pkg.lib().p2_gpu_device_count() <= 0:
        pytest.fail("no HIP device")
    data, pws, (sk_t, pk_t), cases = circuits.ecgfp5_public_key(pkg, [5, 6])
    assert _gpu_vs_oracle(pkg, orc, data, pws)[1] == [0, 0]
    pw = pkg.PartialWitness()
    pw.set_secret_key_target(sk_t, cases[0][0])
    pw.set_point_target(pk_t, _flip(cases[0][1]))
    assert data.prove_batch([pw])[1] == [1]

    data, pws, (pk_t, nonce_t, msg_t, ct_t), cases = circuits.ecgfp5_elgamal(pkg, [5, 6])
    assert _gpu_vs_oracle(pkg, orc, data, pws)[1] == [0, 0]
    sk, pk, msg, nonce, ct = cases[0]
    pw = pkg.PartialWitness()
    pw.set_point_target(pk_t, pk)
    pw.set_point_target(msg_t, msg)
    pw.set_biguint320_target(nonce_t, nonce)
    pw.set_point_target(ct_t[0], ct[0])
    pw.set_point_target(ct_t[1], _flip(ct[1]))
    assert data.prove_batch([pw])[1] == [1]

    data, pws, _, _ = circuits.ecgfp5_hashed_elgamal(pkg, [5, 6])
    assert _gpu_vs_oracle(pkg, orc, data, pws)[1] == [0, 0]

"""CPU tests of the host layer: C-ABI surface, builder behaviour, blob handling, verifier, and the oracle -> verifier
loop on the reference's own circuit tests (accept / reject behaviour is what those tests pin)."""
import ctypes as C
import hashlib
import os
import re

import pytest

import circuits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "p2aes.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(p2_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"p2_builder", "p2_circuit"}
    lib = C.CDLL(pkg.lib_path())
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert len(declared) >= 60
    assert not pkg.lib()._p2_missing


def test_prove_without_gpu_fails_loudly(pkg):
    if pkg.lib().p2_gpu_device_count() > 0:
        pytest.skip("a GPU is present")
    data, pws = circuits.assert_byte(pkg, [3])
    with pytest.raises(pkg.P2Error, match="no HIP device"):
        data.prove(pws[0])


def test_builder_constant_folding_and_caching(pkg):
    b = pkg.CircuitBuilder()
    x, y = b.add_virtual_target(), b.add_virtual_target()
    z = b.zero()
    assert b.constant(5) == b.constant(5)
    assert b.mul_const_add(256, z, y) == y          # first term zero, const_1 == 1 -> addend (arithmetic_special_cases)
    assert b.mul(b.one(), x) == x                   # 1 * x
    assert b.add(b.constant(2), b.constant(3)) == b.constant(5)
    g0 = b.num_gates()
    r1 = b.mul_const_add(256, x, y)
    assert b.num_gates() == g0 + 1
    assert b.mul_const_add(256, x, y) == r1 and b.num_gates() == g0 + 1   # base_arithmetic_results cache
    for _ in range(19):
        b.mul_const_add(256, b.add_virtual_target(), y)
    assert b.num_gates() == g0 + 1                  # 20 ops share one ArithmeticGate row
    b.mul_const_add(256, b.add_virtual_target(), y)
    assert b.num_gates() == g0 + 2
    b.mul_const_add(8, x, y)
    assert b.num_gates() == g0 + 3                  # different constants -> different row
    assert b.add_lookup_table_from_pairs([(1, 2), (3, 4)]) == b.add_lookup_table_from_pairs([(1, 2), (3, 4)])
    with pytest.raises(pkg.P2Error):
        b.add_lookup_from_index(x, 7)


def test_partial_witness_set_target_conflict(pkg):
    pw = pkg.PartialWitness()
    pw.set_target(1, 5)
    pw.set_target(1, 5)
    with pytest.raises(pkg.P2Error):
        pw.set_target(1, 6)
    with pytest.raises(pkg.P2Error):
        pw.set_target(2, 0xFFFFFFFF00000001)


def test_circuit_shapes(pkg):
    # row counts of the reference's size-report configurations (circuit_gcm.rs:708-736 prints them, nothing is
    # recorded upstream; these are this repo's own regression values)
    data, _, _ = circuits.encrypt(pkg, 4, 1024, False)
    assert data.info["degree_bits"] == 14 and data.info["num_luts"] == 3 and data.info["num_constants_cols"] == 11
    # 3 tables x (LookupGate + LookupTableGate) + Noop + Constant + PublicInput + Arithmetic = 10 gate types: two selector
    # polynomials ([0, 8) and [8, 10)), 5 + 3 lookup selectors, 2 gate constants
    assert data.info["num_gate_kinds"] == 10
    assert data.info["num_zs_cols"] == 34 and data.info["num_quotient_cols"] == 16 and data.info["num_fri_rounds"] == 3
    assert data.info["proof_bytes"] == 151372
    data, _, _ = circuits.encrypt(pkg, 4, 13, True)
    assert data.info["degree_bits"] == 13 and data.info["num_luts"] == 5
    data, _ = circuits.assert_byte(pkg, [1])
    assert data.info["degree_bits"] == 4 and data.info["num_fri_rounds"] == 0


def test_blob_rejects_garbage(pkg):
    data, _ = circuits.assert_byte(pkg, [1])
    info = pkg.api._Info()
    assert pkg.lib().p2_blob_info(b"nonsense", 8, C.byref(info)) != 0
    assert pkg.lib().p2_blob_info(data.blob[:100], 100, C.byref(info)) != 0
    bad = bytearray(data.blob)
    bad[8] = 99  # version
    assert pkg.lib().p2_blob_info(bytes(bad), len(bad), C.byref(info)) != 0


def _prove_verify(pkg, orc, data, pws, expect_ok=True):
    oc = orc.OracleCircuit(data.blob)
    vd = oc.verifier_data()
    out = []
    for pw in pws:
        st, proof = oc.prove(pw.map)
        if expect_ok:
            assert st == 0
            assert len(proof) == data.proof_bytes
            data.verify(proof, vd)
        out.append((st, proof))
    return oc, vd, out


def test_assert_byte_accepts_bytes_rejects_others(pkg, orc):
    data, pws = circuits.assert_byte(pkg, [0, 255, 256, 0xFFFFFFFF00000000, 70000])
    oc = orc.OracleCircuit(data.blob)
    sts = [oc.prove(pw.map)[0] for pw in pws]
    assert sts == [0, 0, 1, 1, 1]          # circuit_aes.rs:403-405: tv > 255 => prove is Err


def test_sub_bytes_and_negative(pkg, orc):
    data, pws = circuits.sub_bytes(pkg, orc, circuits.random_states(1, 2))
    oc, vd, res = _prove_verify(pkg, orc, data, pws)
    bad = dict(pws[0].map)
    k = list(bad)[-1]
    bad[k] ^= 1
    assert oc.prove(bad)[0] == 1
    missing = dict(pws[0].map)
    del missing[list(missing)[0]]
    assert oc.prove(missing)[0] == 2


def test_verifier_rejects_tampering(pkg, orc):
    data, pws = circuits.gf_2_8_mul(pkg, [(0x57, 0x13, 0xFE)])
    oc, vd, res = _prove_verify(pkg, orc, data, pws)
    proof = res[0][1]
    data.verify(proof, vd)
    n = len(proof)
    for pos in (0, 600, 1600, 4000, 5000, n // 2, n - 200, n - 9, n - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        with pytest.raises(pkg.P2Error):
            data.verify(bytes(bad), vd)
    with pytest.raises(pkg.P2Error):
        data.verify(proof[:-1], vd)
    with pytest.raises(pkg.P2Error):
        data.verify(proof + b"\0", vd)
    bad_vd = list(vd)
    for k in range(16):  # every cap digest (a single one is only hit by ~1/16 of the 28 queries)
        bad_vd[4 * k] ^= 1
    with pytest.raises(pkg.P2Error):
        data.verify(proof, bad_vd)
    bad_vd = list(vd)
    bad_vd[-1] ^= 1      # circuit digest feeds the transcript
    with pytest.raises(pkg.P2Error):
        data.verify(proof, bad_vd)
    # a proof for one circuit does not verify against another
    data2, pws2 = circuits.gf_2_8_add(pkg, [(1, 2)])
    oc2 = orc.OracleCircuit(data2.blob)
    with pytest.raises(pkg.P2Error):
        data2.verify(proof, oc2.verifier_data())


def test_gf_2_8_mul_all_reference_triples(pkg, orc):
    import json
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "aes_kat.json")))
    data, pws = circuits.gf_2_8_mul(pkg, kat["gf_2_8_mul"][-3:])
    _prove_verify(pkg, orc, data, pws)
    data, pws = circuits.gf_2_8_mul(pkg, [(0x57, 0x13, 0xFF)])
    assert orc.OracleCircuit(data.blob).prove(pws[0].map)[0] == 1


@pytest.mark.parametrize("name", ["mix_columns", "gf_2_8_add", "key_expansion_128", "encrypt_block_fips197", "right_shift_one",
                                  "gctr_128_13", "gf_2_128_mul", "ghash_16", "encrypt_128_13", "encrypt_128_13_tag", "encrypt_128_17",
                                  "encrypt_256_13"])
def test_reference_circuit_tests(pkg, orc, name):
    kat_key = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
    if name == "mix_columns":
        data, pws = circuits.mix_columns(pkg, circuits.random_states(2, 1))
    elif name == "gf_2_8_add":
        data, pws = circuits.gf_2_8_add(pkg, [(0xA5, 0x3C)])
    elif name == "key_expansion_128":
        data, pws = circuits.key_expansion(pkg, kat_key)
    elif name == "encrypt_block_fips197":
        data, pws = circuits.encrypt_block(pkg, kat_key, bytes.fromhex("3243f6a8885a308d313198a2e0370734"),
                                           expected=bytes.fromhex("3925841d02dc09fbdc118597196a0b32"))
    elif name == "right_shift_one":
        data, pws = circuits.right_shift_one(pkg)
    elif name == "gctr_128_13":
        data, pws = circuits.gctr(pkg, 4, 13)
    elif name == "gf_2_128_mul":
        data, pws = circuits.gf_2_128_mul(pkg)
    elif name == "ghash_16":
        data, pws = circuits.ghash(pkg, 16)
    elif name == "encrypt_128_13":
        data, pws, _ = circuits.encrypt(pkg, 4, 13, False)
    elif name == "encrypt_128_13_tag":
        data, pws, _ = circuits.encrypt(pkg, 4, 13, True)
    elif name == "encrypt_128_17":
        data, pws, _ = circuits.encrypt(pkg, 4, 17, False)
    else:
        data, pws, _ = circuits.encrypt(pkg, 8, 13, False)
    _prove_verify(pkg, orc, data, pws)


def test_oracle_proofs_are_deterministic(pkg, orc):
    data, pws = circuits.gf_2_8_add(pkg, [(7, 9)])
    oc = orc.OracleCircuit(data.blob)
    a, b = oc.prove(pws[0].map)[1], oc.prove(pws[0].map)[1]
    assert a == b and hashlib.sha256(a).hexdigest() == hashlib.sha256(b).hexdigest()


def test_circuit_without_lookup_tables(pkg, orc):
    P = 0xFFFFFFFF00000001
    data, pws = circuits.arithmetic_only(pkg, [(3, 5, 11, 92), (3, 5, 11, 93), (P - 1, P - 2, 12345678901234567, 1)])
    assert data.info["num_luts"] == 0 and data.info["num_zs_cols"] == 20
    _prove_verify(pkg, orc, data, pws)
    bad = dict(pws[0].map)
    bad[list(bad)[-1]] = 1
    assert orc.OracleCircuit(data.blob).prove(bad)[0] == 1


def test_poseidon_native_round_trip_and_sponge(pkg, orc):
    # poseidon-cipher/src/lib.rs:127 test_encrypt_decrypt message lengths
    import ctypes as C
    import random
    P = 0xFFFFFFFF00000001
    r = random.Random(8)
    fq = lambda: tuple(r.randrange(P) for _ in range(5))  # noqa: E731
    for n in (9, 10, 11, 12, 128, 129, 1023, 1024, 1025):
        ks, nonce, msg = [fq(), fq()], [r.randrange(P), r.randrange(P)], [fq() for _ in range(n)]
        ct = pkg.poseidon_native.encrypt(ks, msg, nonce)
        assert len(ct) == (n + 2) // 3 * 3 + 1
        assert pkg.poseidon_native.decrypt(ks, ct, nonce, n) == msg
        bad = list(ct)
        bad[-1] = tuple((v + 1) % P for v in bad[-1])
        with pytest.raises(pkg.P2Error):
            pkg.poseidon_native.decrypt(ks, bad, nonce, n)
    # native sponge == the oracle's hash_n_to_hash_no_pad on the first 4 outputs
    for n in (1, 7, 8, 9, 20):
        inp = [r.randrange(P) for _ in range(n)]
        out = (C.c_uint64 * 4)()
        orc.lib().orc_hash_no_pad((C.c_uint64 * n)(*inp), n, out)
        assert pkg.poseidon_native.hash_n_to_m_no_pad(inp, 20)[:4] == list(out)


def test_poseidon_cipher_circuit(pkg, orc):
    # BASELINE.json configs[0] shape: 32-byte message = 3 Fq elements (L = 3); also L = 6
    for L in (3, 6):
        data, pws, t, cases = circuits.poseidon_encrypt(pkg, L, [1, 2])
        assert data.info["num_luts"] == 0 and data.info["num_ops"] >= 10 * (L // 3 + 1)
        oc, vd, res = _prove_verify(pkg, orc, data, pws)
        ks, msg, nonce, ct = cases[0]
        bad_ct = [tuple(v ^ (1 if (i, j) == (1, 2) else 0) for j, v in enumerate(fq)) for i, fq in enumerate(ct)]
        pw = pkg.PartialWitness()
        t.set_targets(pw, ks, msg, nonce, bad_ct)
        assert oc.prove(pw.map)[0] == 1


def test_in_circuit_hash_matches_native(pkg, orc):
    P = 0xFFFFFFFF00000001
    b = pkg.CircuitBuilder()
    ins = b.add_virtual_target_arr(11)
    outs = b.hash_n_to_m_no_pad(ins, 9)
    data = b.build()
    vals = [(i * 0x9E3779B97F4A7C15 + 5) % P for i in range(11)]
    pw = pkg.PartialWitness()
    pw.set_target_arr(ins, vals)
    pw.set_target_arr(outs, pkg.poseidon_native.hash_n_to_m_no_pad(vals, 9))
    _prove_verify(pkg, orc, data, [pw])
    pw2 = pkg.PartialWitness()
    pw2.set_target_arr(ins, vals)
    pw2.set_target_arr(outs, [1] * 9)
    assert orc.OracleCircuit(data.blob).prove(pw2.map)[0] == 1


def test_feistel_poseidon_circuit(pkg, orc):
    data, pws = circuits.feistel_poseidon(pkg, [1, 2])
    assert data.info["num_luts"] == 0
    oc, vd, res = _prove_verify(pkg, orc, data, pws)
    bad = dict(pws[0].map)
    bad[list(bad)[-1]] ^= 1
    assert oc.prove(bad)[0] == 1


def test_zero_knowledge_config(pkg, orc):
    """standard_recursion_zk_config: blinding rows enlarge the circuit, leaves of the three blinded oracles carry 4 salt
    elements, proofs verify, differ from proof to proof, and are reproducible for a given (seed, proof index)."""
    data, pws = circuits.zk_gf_2_8_add(pkg, [(5, 9)])
    plain, _ = circuits.gf_2_8_add(pkg, [(5, 9)])
    assert data.info["degree_bits"] > plain.info["degree_bits"]
    assert data.proof_bytes > plain.proof_bytes
    oc = orc.OracleCircuit(data.blob)
    vd = oc.verifier_data()
    oc.set_zk(1234, 0)
    st, p1 = oc.prove(pws[0].map)
    assert st == 0 and len(p1) == data.proof_bytes
    data.verify(p1, vd)
    oc.set_zk(1234, 1)
    p2 = oc.prove(pws[0].map)[1]
    data.verify(p2, vd)
    assert p1 != p2
    oc.set_zk(1234, 0)
    assert oc.prove(pws[0].map)[1] == p1
    with pytest.raises(pkg.P2Error):
        plain.verify(p1, vd)          # a zk proof is not a proof for the non-zk circuit


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_circuits(pkg, orc, seed):
    data, pws = circuits.random_circuit(pkg, orc, seed)
    oc, vd, res = _prove_verify(pkg, orc, data, pws)
    # flipping any asserted value must make witness generation fail
    bad = dict(pws[0].map)
    k = list(bad)[-1]
    bad[k] = (bad[k] + 1) % 0xFFFFFFFF00000001
    assert oc.prove(bad)[0] in (1, 2) or len(pws[0].map) <= 6


def test_soundness_every_constraint_family_bites(pkg, orc):
    """Fault injection in the oracle prover: a witness that violates ONE constraint family must yield a proof the verifier
    rejects.  Slot faults change every copy of a value (the permutation argument still holds), so only the gate /
    lookup constraints can catch them; cell faults change a single wire of a copy-constrained set."""
    G_LOOKUP, G_LUT, G_NOOP, G_CONST, G_PI, G_ARITH, G_POS = range(7)
    OP_ARITH, OP_CONST, OP_LOOKUP, OP_EQ, OP_EQINV, OP_POSEIDON = range(6)
    P = 0xFFFFFFFF00000001

    def must_reject(data, oc, vd, pw, what):
        st, proof = oc.prove(pw.map)
        oc.set_fault(0)
        assert st == 0, what
        with pytest.raises(pkg.P2Error):
            data.verify(proof, vd)

    # circuit 1: AES mix_columns (arithmetic + two lookup tables)
    data, pws = circuits.mix_columns(pkg, circuits.random_states(7, 1))
    oc = orc.OracleCircuit(data.blob)
    vd = oc.verifier_data()
    st, good = oc.prove(pws[0].map)
    data.verify(good, vd)
    n = 1 << data.info["degree_bits"]
    kinds = [oc.row_gate_kind(r) for r in range(n)]
    ops = oc.ops()
    # slots produced by an arithmetic op / a lookup op / a constant op
    for kind, what in ((OP_ARITH, "ArithmeticGate constraint"), (OP_LOOKUP, "lookup argument (looking value not in table)"), (OP_CONST, "ConstantGate constraint")):
        slot = next(o for k, o in ops if k == kind)
        oc.set_fault(1, slot, 0, 1)
        # the asserted outputs would make witness generation itself fail; drop them for slot faults on asserted values
        must_reject(data, oc, vd, _inputs_only(pws[0], data, oc), what)
    # LookupTableGate: wrong multiplicity, and a corrupted table entry
    row = kinds.index(G_LUT)
    oc.set_fault(2, 2, row, 1)
    must_reject(data, oc, vd, pws[0], "lookup argument (multiplicity)")
    oc.set_fault(2, 1, row, 1)
    must_reject(data, oc, vd, pws[0], "lookup table content (RE polynomial)")
    # permutation argument: one cell of an arithmetic row input (copy-constrained to its source) -- also trips the gate;
    # and a cell on a NoopGate row that no gate constrains and no copy constraint touches must NOT matter
    row = kinds.index(G_ARITH)
    oc.set_fault(2, 0, row, 5)
    must_reject(data, oc, vd, pws[0], "permutation / arithmetic")
    # PublicInputGate: hash wires must be zero
    row = kinds.index(G_PI)
    oc.set_fault(2, 0, row, 1)
    must_reject(data, oc, vd, pws[0], "PublicInputGate")

    # circuit 2: PoseidonGate -- corrupt an S-box wire of a gate row (unrouted advice wire: only the gate can notice)
    data, pws, t, cases = circuits.poseidon_encrypt(pkg, 3, [1])
    oc = orc.OracleCircuit(data.blob)
    vd = oc.verifier_data()
    n = 1 << data.info["degree_bits"]
    row = [oc.row_gate_kind(r) for r in range(n)].index(G_POS)
    for col, what in ((100, "PoseidonGate full-round S-box wire"), (66, "PoseidonGate partial-round S-box wire"), (24, "PoseidonGate swap bit")):
        oc.set_fault(2, col, row, 1)
        must_reject(data, oc, vd, pws[0], what)


def _inputs_only(pw, data, oc):
    """The witness restricted to virtual-target inputs (asserted outputs removed), as a PartialWitness-like object."""
    class _PW:
        pass
    out = _PW()
    keys = list(pw.map)
    out.map = {k: pw.map[k] for k in keys[:16]}   # circuits.mix_columns sets the 16 state inputs first
    return out


def test_blob_mutation_fuzz_never_crashes(pkg):
    """p2_blob_info (the parser behind p2_circuit_load / p2_verify) on mutated blobs: it may accept or reject, never crash."""
    import random
    data, _ = circuits.sub_bytes(pkg, __import__("oracle_lib"), circuits.random_states(1, 1))
    data2, _, _, _ = circuits.poseidon_encrypt(pkg, 3, [1])
    info = pkg.api._Info()
    r = random.Random(0)
    rejected = 0
    for blob in (data.blob, data2.blob):
        n = len(blob)
        assert pkg.lib().p2_blob_info(blob, n, C.byref(info)) == 0
        for trial in range(300):
            b = bytearray(blob)
            mode = trial % 3
            if mode == 0:      # flip bytes, mostly in the structured header region
                for _ in range(r.randrange(1, 4)):
                    pos = r.randrange(min(n, 400)) if r.random() < 0.7 else r.randrange(n)
                    b[pos] = r.randrange(256)
            elif mode == 1:    # truncate
                b = b[: r.randrange(n)]
            else:              # overwrite a 4-byte field with an extreme value
                pos = r.randrange(n - 4)
                b[pos:pos + 4] = r.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x80", b"\xff\xff\xff\x7f"])
            rc = pkg.lib().p2_blob_info(bytes(b), len(b), C.byref(info))
            rejected += rc != 0
    assert rejected > 200


def test_host_selftest_of_the_shared_arithmetic(pkg):
    """poseidon_fast.h (lazy carry-chain reductions, sparse partial rounds, accumulator fold) is the arithmetic the hashing
    kernels run; its host build must agree with 128-bit arithmetic and with the plain permutation.  (The device build is
    cross-checked by tools/microbench/reduce_check.hip and poseidon_bench.hip and by every GPU parity test.)"""
    assert pkg.lib().p2_selftest_host(0x5EED, 2_000_000, 2_000) == 0
    assert pkg.lib().p2_selftest_host(7, 200_000, 500) == 0



def test_gate_counts_of_the_report_sizes_list(pkg):
    """circuit_gcm.rs:708-736 (test_encrypt_report_sizes): builder.num_gates() and the compiled shape for AES-GCM-128/-256,
    L in 16..2048, against tests/golden/gate_counts.json (this build's own frozen counts -- the reference prints them but
    holds none; catches builder drift on the CPU)."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "gate_counts.json")))["sizes"]
    assert len(gold) == 20
    for e in gold:
        if e["L"] > 1024 and e["nk"] == 8:
            continue                                   # the largest one costs the most host time and adds no new shape
        b = pkg.CircuitBuilder()
        pkg.AesGcmTarget.build(b, e["nk"], e["nk"] + 6, e["L"], False)
        assert b.num_gates() == e["num_gates"], e
        info = b.build().info
        for k in ("degree_bits", "num_ops", "num_levels", "num_slots", "proof_bytes"):
            assert info[k] == e[k], (e, k, info[k])


def _query_layout(info):
    """Byte offsets of the Merkle-path length prefixes of query 0 (DESIGN.md 'Proof layout'), non-zk circuits."""
    lde_bits, cap_h, rounds = info["degree_bits"] + 3, 4, info["num_fri_rounds"]
    cols = [info["num_constants_cols"] + info["num_routed_wires"], info["num_wires"], info["num_zs_cols"], info["num_quotient_cols"]]
    qbytes = sum(8 * c + 1 + 32 * (lde_bits - cap_h) for c in cols)
    bits = lde_bits
    for _ in range(rounds):
        bits -= 4
        qbytes += 16 * 16 + 1 + 32 * (bits - cap_h)
    fl = (1 << info["degree_bits"]) >> (4 * rounds)
    q0 = info["proof_bytes"] - 8 - 16 * fl - 28 * qbytes
    offs, pos = [], q0
    for c in cols:
        pos += 8 * c
        offs.append((pos, lde_bits - cap_h))
        pos += 1 + 32 * (lde_bits - cap_h)
    return offs


def test_verifier_rejects_merkle_paths_of_the_wrong_depth(pkg, orc):
    """ADVICE r1 (verifier.h): a proof whose total length is right but whose Merkle paths have other depths -- one path a
    sibling longer, the next a sibling shorter -- must be rejected for its SHAPE, before any hashing decides."""
    data, pws = circuits.gf_2_8_mul(pkg, [(0x57, 0x13, 0xFE)])
    oc, vd, res = _prove_verify(pkg, orc, data, pws)
    proof = res[0][1]
    (p0, d0), (p1, d1) = _query_layout(data.info)[:2]
    assert proof[p0] == d0 and proof[p1] == d1       # the layout helper points at the length prefixes
    bad = bytearray(proof)
    bad[p0] = d0 + 1                                 # path 0 claims one sibling more ...
    bad[p0 + 1 + 32 * d0: p0 + 1 + 32 * d0] = bytes(32)
    q1 = p1 + 32                                     # (path 1's prefix moved by the inserted sibling)
    assert bad[q1] == d1
    bad[q1] = d1 - 1                                 # ... path 1 one fewer: same total length
    del bad[q1 + 1 + 32 * (d1 - 1): q1 + 1 + 32 * d1]
    assert len(bad) == len(proof)
    with pytest.raises(pkg.P2Error, match="depth"):
        data.verify(bytes(bad), vd)
    # non-canonical encodings of field elements are rejected wherever they appear: pow witness, a cap word, a leaf value
    n = len(proof)
    for pos in (n - 8, 0, p0 - 8):
        bad = bytearray(proof)
        bad[pos:pos + 8] = (0xFFFFFFFFFFFFFFFF).to_bytes(8, "little")
        with pytest.raises(pkg.P2Error, match="non-canonical"):
            data.verify(bytes(bad), vd)


def test_ecgfp5_default_randomness_is_the_os_csprng(pkg):
    """keys, nonces and encode_binary padding default to the OS CSPRNG (the reference's OsRng,
    ecgfp5/src/lib.rs:35,64); the seeded forms are reproducible and test-only."""
    E = pkg.ecgfp5
    order = E.group_order()
    ks = {E.random_scalar() for _ in range(8)}
    assert len(ks) == 8 and all(0 <= k < order for k in ks)
    assert max(ks).bit_length() > 300                      # full-width scalars, not 64-bit seeds stretched
    assert E.random_scalar(7) == E.random_scalar(7) and E.random_scalar(7) != E.random_scalar(8)
    a, b = E.new_rand_from_subgroup(), E.new_rand_from_subgroup()
    assert a != b and E.is_in_subgroup(a) and E.is_in_subgroup(b)
    x = 0x0123456789ABCDEF0123456789ABCDEF01234567
    p1, p2 = E.encode_binary(x), E.encode_binary(x)
    assert p1 != p2 and E.decode_binary(p1) == x and E.decode_binary(p2) == x
    assert E.encode_binary(x, 5) == E.encode_binary(x, 5)
    sk1, sk2 = pkg.ECGFP5SecretKey.rand(), pkg.ECGFP5SecretKey.rand()
    assert sk1.value != sk2.value
    k1, K1 = pkg.poseidon_native.new_key()
    k2, K2 = pkg.poseidon_native.new_key()
    assert k1 != k2 and K1 != K2


def test_zk_prf_key_is_full_width_in_the_oracle(pkg, orc):
    """The blinding PRF (circuit.h 'Blinding randomness') is keyed by four field elements: each of them, the proof index
    and the seed shorthand all change the oracle's zk proof; equal keys reproduce it."""
    data, pws = circuits.zk_gf_2_8_add(pkg, [(5, 9)])
    oc = orc.OracleCircuit(data.blob)
    vd = oc.verifier_data()
    key = [11, 22, 33, 44]
    oc.set_zk_key(key, 0)
    st, base = oc.prove(pws[0].map)
    assert st == 0
    data.verify(base, vd)
    oc.set_zk_key(key, 0)
    assert oc.prove(pws[0].map)[1] == base
    # every key word reaches the blinding rows (the witness alone shows it: no need to prove five more times)
    cap = data.info["num_wires"] << data.info["degree_bits"]
    seen = {tuple(oc.generate_witness(pws[0].map, cap)[1])}
    for w in range(4):
        k2 = list(key)
        k2[w] += 1
        oc.set_zk_key(k2, 0)
        seen.add(tuple(oc.generate_witness(pws[0].map, cap)[1]))
    assert len(seen) == 5
    oc.set_zk_key(key, 1)
    assert oc.prove(pws[0].map)[1] != base
    oc.set_zk(99, 3)
    a = oc.prove(pws[0].map)[1]
    oc.set_zk_key([99, 0, 0, 0], 3)
    assert oc.prove(pws[0].map)[1] == a


def test_c_abi_functions_do_not_throw_across_the_boundary(pkg):
    """a bad LUT index / target through a void or target-returning entry point sets p2_last_error and returns the
    function's error value instead of ending in std::terminate."""
    import ctypes as C
    L = pkg.lib()
    b = pkg.CircuitBuilder()
    t = b.add_virtual_target()
    assert L.p2_builder_add_lookup_from_index(b._h, t, 12345) == 0xFFFFFFFFFFFFFFFF
    assert b"" != L.p2_last_error()
    out = (C.c_uint64 * 16)()
    st = (C.c_uint64 * 16)(*[t] * 16)
    L.p2_aes_state_sub_bytes(b._h, 777, st, out)          # no such table: must return, not abort
    assert L.p2_last_error()


def test_rust_ffi_matches_the_c_header(pkg):
    """The shim crate (bindings/plonky2-hip; replaces /root/reference/Cargo.toml:12) cannot be compiled in this
    image, so its FFI surface is checked mechanically: (1) src/ffi.rs is exactly what tools/gen_rust_ffi.py generates from
    include/p2aes.h today; (2) an independent parse of the Rust extern block agrees with an independent parse of the header
    on every name, arity and parameter type; (3) every declared function is exported by libp2aes.so; (4) every `ffi::` call
    in src/lib.rs names a declared function and passes the declared number of arguments."""
    import re
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi as G
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]) == 0
    rs = open(os.path.join(ROOT, "bindings", "plonky2-hip", "src", "ffi.rs")).read()
    rust = {}
    for m in re.finditer(r"pub fn (p2_\w+)\((.*?)\)(?: -> ([^;]+))?;", rs):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(",") if p.strip()]
        rust[m.group(1)] = (params, (m.group(3) or "()").strip())
    # independent, deliberately simple C-side reading: count commas / map the base types by hand
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "p2aes.h")).read(), flags=re.S)
    cmap = {"uint64_t": "u64", "p2_target": "u64", "uint32_t": "u32", "uint16_t": "u16", "uint8_t": "u8", "int": "c_int", "long": "c_long",
            "size_t": "usize", "char": "c_char", "void": "c_void", "p2_builder": "P2Builder", "p2_circuit": "P2Circuit",
            "p2_assignment": "P2Assignment", "p2_circuit_info": "P2CircuitInfo", "p2_kernel_time": "P2KernelTime"}
    seen = 0
    for m in re.finditer(r"\b(p2_\w+)\s*\(([^;{}()]*)\)\s*;", hdr):
        name, args = m.group(1), m.group(2).strip()
        if name not in rust:
            assert "typedef" in hdr[max(0, m.start() - 40):m.start()], "header function %s has no Rust declaration" % name
            continue
        seen += 1
        cargs = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        rparams, _ = rust[name]
        assert len(cargs) == len(rparams), (name, cargs, rparams)
        for ca, rp in zip(cargs, rparams):
            is_ptr = "*" in ca or "[" in ca
            assert is_ptr == rp.startswith("*"), (name, ca, rp)
            base = next(cmap[w] for w in re.findall(r"\w+", ca) if w in cmap)
            assert rp.split()[-1] == base or (not is_ptr and rp == base), (name, ca, rp)
            if is_ptr:
                assert rp.startswith("*const") == bool(re.search(r"\bconst\b", ca)), (name, ca, rp)
    assert seen == len(rust) >= 100
    missing = [n for n in rust if not hasattr(pkg.lib(), n)]
    assert not missing, missing
    lib_rs = open(os.path.join(ROOT, "bindings", "plonky2-hip", "src", "lib.rs")).read()
    calls = re.findall(r"ffi::(p2_\w+)\(", lib_rs)
    assert len(set(calls)) >= 35
    for m in re.finditer(r"ffi::(p2_\w+)\(", lib_rs):
        name = m.group(1)
        assert name in rust, "lib.rs calls undeclared %s" % name
        depth, i, nargs, any_arg = 1, m.end(), 0, False
        while depth:
            ch = lib_rs[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                nargs += 1
            elif not ch.isspace():
                any_arg = True
            i += 1
        nargs += 1 if any_arg else 0
        assert nargs == len(rust[name][0]), "lib.rs passes %d arguments to %s, the header declares %d" % (nargs, name, len(rust[name][0]))
    # every SURVEY A.1 builder / witness symbol is present in the shim
    for sym in ("add_virtual_target", "add_virtual_target_arr", "constant", "zero", "one", "mul_const_add", "fn add(", "fn mul(", "is_equal", "select",
                "connect", "add_lookup_table_from_pairs", "add_lookup_from_index", "hash_n_to_m_no_pad", "num_gates", "fn build<", "new_unsafe",
                "set_target", "set_target_arr", "fn prove(", "fn verify(", "standard_recursion_config", "standard_recursion_zk_config",
                "from_canonical_u8", "NEG_ONE", "ORDER", "fn rand()", "QuinticExtension", "add_virtual_point_target", "constant_point",
                "multiply_point", "add_point", "add_virtual_biguint320_target", "decompress_into_subgroup", "compress_from_subgroup",
                "new_rand_from_subgroup", "fn generator()", "fn inverse("):
        assert sym in lib_rs, "shim crate lacks %s" % sym


def test_proof_digests_are_frozen(pkg, orc):
    """tests/golden/proof_digests.json: the compiled circuit blob, the verifier data and the oracle's proof bytes for seven of
    the reference's circuit tests with their fixed inputs, by SHA-256.  The fixture is this build's own output (the reference
    holds no proof bytes and cannot be run here: parity with real plonky2 stays unpinned) -- what it buys is that any drift
    of the builder or the oracle is caught here on the CPU, and of the GPU prover by the same fixture in
    test_gpu_parity.py::test_gpu_proofs_match_the_frozen_digests."""
    import digest_cases as D
    gold = D.fixture()
    assert gold["zk_key"] == D.ZK_KEY
    seen = 0
    for name, (data, pws) in D.cases(pkg):
        got = D.digest_case(orc, data, pws[0], name.startswith("zk_"))
        assert got == gold["cases"][name], name
        seen += 1
    assert seen == len(gold["cases"]) == 7


@pytest.mark.parametrize("name", ["aes_gcm_128_1024", "elgamal_encrypt"])
def test_full_size_workload_digests_are_frozen(pkg, orc, name):
    """Round 3: the same for BASELINE.json's GPU workloads at full size (the bench circuit AesGcm128Target<1024> and the ElGamal
    encryption circuit), including a digest of every intermediate stage of the oracle's proof.  The third large entry,
    AesGcm128Target<65536> (n = 2^19 rows), costs the oracle tens of minutes per proof: it was written once by
    tools/make_proof_digests.py --large in the build container and is held against the GPU prover only
    (test_gpu_parity.py::test_full_size_workloads_match_the_frozen_digests)."""
    import digest_cases as D
    want = D.fixture()["large_cases"][name]
    data, pws = D.large_case(pkg, name)
    assert hashlib.sha256(data.blob).hexdigest() == want["blob_sha256"]
    oc = orc.OracleCircuit(data.blob)
    st, proof = oc.prove(pws[0].map, trace=True)
    assert st == 0
    assert D.sha_words(oc.verifier_data()) == want["verifier_data_sha256"]
    got = D.stage_digests_oracle(oc, data.info, want["live_wire_columns"] << data.info["degree_bits"])
    for stage in D.STAGE_ORDER:
        assert got[stage] == want["stages"][stage], stage
    assert len(proof) == want["proof_bytes"] and hashlib.sha256(proof).hexdigest() == want["proof_sha256"]
    assert set(D.fixture()["large_cases"]) == set(D.LARGE)


def test_witness_schedule_keeps_every_dependency(pkg, orc):
    """csrc/witness_schedule.h (round 3): the prover contracts the critical path of the witness program into CHAINS (straight-line
    runs of up to K lookup-free ops executed by one thread) and leaves every other op a single, one workgroup barrier per level.
    p2_witness_schedule_check rebuilds that schedule on the host and checks it op by op: nothing lost, the levels tile the
    program, every slot keeps its first producer, every operand is produced in an earlier level or earlier in the same chain,
    chains hold only the kinds the chain executor runs.  Circuits: lookups + arithmetic (AES-GCM with the inc32 carry chain),
    `connect`ed second producers, PoseidonGate rows, random circuits over the whole vocabulary."""
    shapes = []
    for data in (circuits.encrypt(pkg, 4, 64, False)[0], circuits.encrypt(pkg, 4, 13, True)[0], circuits.arithmetic_only(pkg, [(1, 2, 3, 4)])[0],
                 circuits.feistel_poseidon(pkg, [1], rounds=4)[0], circuits.random_circuit(pkg, orc, 5)[0], circuits.random_circuit(pkg, orc, 6)[0]):
        base = data.witness_schedule(1)
        assert base["chains"] == 0 and base["fused_ops"] == 0 and base["max_chain"] == 0
        assert base["levels"] <= data.info["num_levels"]                  # never deeper than the builder's own levelisation
        prev = base["levels"]
        for K in (2, 4, 8):
            s = data.witness_schedule(K)
            assert s["max_chain"] <= K and s["levels"] <= prev and s["fused_ops"] >= s["chains"]
            prev = s["levels"]
        shapes.append((base["levels"], prev))
    assert shapes[0][1] < shapes[0][0]     # the AES-CTR carry chain does contract


def test_witness_schedule_contracts_the_counter_chain_of_a_deep_circuit(pkg):
    """AesGcm128Target<2048>: 128 counter blocks chained through inc32 (aes-gcm/src/circuit_gcm.rs:350-368) -- the shape that makes
    the 64 KiB circuit 16.5 k levels deep.  With chains of 8 the counter chain costs a fraction of a level per block instead of
    three, and well under 2 % of the ops are fused: the rest of the program keeps its width."""
    data = circuits.encrypt(pkg, 4, 2048, False)[0]
    one, eight = data.witness_schedule(1), data.witness_schedule(8)
    assert eight["levels"] * 2 < one["levels"] and eight["fused_ops"] * 50 < data.info["num_ops"]


def test_bench_uses_a_counter_file_only_for_the_sources_it_was_measured_on(tmp_path):
    """bench.py's roofline.traffic / roofline_valu come from rocprofv3 PMC passes kept under profiles/ (they went
    stale silently).  Each file records the identity of the kernel sources it was measured on; bench.py uses it only while that
    matches the sources it runs on and says so in the line (`traffic_source` / `valu_source`)."""
    import importlib.util
    import json
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from src_id import csrc_id
    here = csrc_id()["csrc_sha16"]
    (tmp_path / "profiles").mkdir()
    data, src = bench.load_profile("traffic", root=str(tmp_path), tag="t")
    assert data is None and src["status"] == "missing"
    (tmp_path / "profiles" / "t_traffic.json").write_text(json.dumps({"source": {"csrc_sha16": "0" * 16, "git_head": "abc"}, "chunk": 128}))
    data, src = bench.load_profile("traffic", root=str(tmp_path), tag="t")
    assert data is None and src["status"].startswith("stale") and src["measured_on_csrc_sha16"] == "0" * 16 and src["this_build_csrc_sha16"] == here
    (tmp_path / "profiles" / "t_traffic.json").write_text(json.dumps({"source": {"csrc_sha16": here}, "chunk": 128}))
    data, src = bench.load_profile("traffic", root=str(tmp_path), tag="t")
    assert data["chunk"] == 128 and src["status"] == "current"


def test_committed_counter_files_were_measured_on_the_committed_kernels():
    """Freshness of profiles/<tag>_traffic.json and _valu.json.  A stale file is not an error of the product -- bench.py then
    reports null with the reason -- so this SKIPS with the reason instead of failing: re-run tools/gpu_profile_round.sh."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for kind in ("traffic", "valu"):
        data, src = bench.load_profile(kind)
        if src["status"] != "current":
            pytest.skip("profiles/%s is %s" % (src["file"], src["status"]))

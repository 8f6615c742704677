"""The product's circuit compiler against an independent derivation of the same circuits.

`oracle/oracle_builder.py` restates plonky2's CircuitBuilder and the reference's AES / AES-GCM gadgets in Python, straight from
the reference's Rust (aes-gcm/src/circuit_aes.rs, circuit_gcm.rs) and plonky2's published builder algorithm, without reading the
product's blob or sharing any code with builder.h / aes_gadgets.h.  Here the two are held against each other for
`AesGcmTarget::build` (circuit_gcm.rs:49-172): the number of gates, the gate types and their selector groups, the whole constants
matrix (selectors, lookup selectors, gate constants), sigma, the lookup tables and their rows, and -- on inputs -- every routed
wire of the witness, including the table rows, the multiplicities and the padding of the last LookupGate row.  Together with the
protocol (proved byte for byte against oracle_prover.h) those determine the proof: a product builder that compiled a different
circuit than the reference's gadgets describe no longer passes by agreeing with itself.  (Both sides restate un-vendored plonky2
from its published algorithm: parity with real plonky2 stays unpinned.)"""
import json
import os
import sys

import numpy as np
import pytest

import circuits
from blob_reader import Blob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_builder as OB  # noqa: E402

CASES = [(4, 13, False), (4, 13, True), (8, 40, False), (6, 16, False), (4, 1024, False)]


def test_keccak_256_known_answers():
    # the Keccak team's published digests of the empty string and of "abc" (original padding, not SHA3-256)
    assert OB.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert OB.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert OB.keccak256(bytes(200)) != OB.keccak256(bytes(201))            # spans two rate blocks


def _both(pkg, nk, L, tag):
    b = pkg.CircuitBuilder()
    target = pkg.AesGcmTarget.build(b, nk, nk + 6, L, tag)
    product_gates = b.num_gates()
    data = b.build()
    ob = OB.Builder()
    otarget = OB.AesGcmTarget(ob, nk, nk + 6, L, tag)
    assert product_gates == ob.num_gates()                                  # what test_encrypt_report_sizes prints
    return data, target, ob.build(), otarget


@pytest.mark.parametrize("nk,L,tag", CASES)
def test_aes_gcm_circuit_equals_the_independent_derivation(pkg, orc, nk, L, tag):
    data, target, shape, otarget = _both(pkg, nk, L, tag)
    B = Blob(data.blob)
    kind_of = {"lookup": 0, "lookup_table": 1, "noop": 2, "constant": 3, "public_input": 4, "arithmetic": 5}
    assert B.n == shape.n
    assert list(B.gates) == [kind_of[g.name] for g in shape.gates]
    assert [tuple(g) for g in B.groups.tolist()] == shape.groups
    assert B.num_gate_constraints == max(g.num_constraints for g in shape.gates)
    want = np.array(shape.constants, dtype=np.uint64)
    assert want.shape == B.constants.shape and (want == B.constants).all()
    assert (np.array(shape.sigma_values(), dtype=np.uint64).reshape(OB.NUM_ROUTED, shape.n) == B.sigmas).all()
    assert len(B.luts) == len(shape.b.luts) and all((np.array(t, dtype=np.uint16) == got).all() for t, got in zip(shape.b.luts, B.luts))
    assert [tuple(r) for r in B.lookup_rows.tolist()] == shape.b.lookup_rows
    assert list(B.num_lookups) == [len(l) for l in shape.b.lut_lookups]
    # the witness: the reference's test inputs (circuit_gcm.rs:739-781 shape: key [42; 4 NK], nonce [111; 12]) and a ragged plaintext
    key, nonce, pt = bytes([42] * (4 * nk)), bytes([111] * 12), bytes((7 * i + 3) & 255 for i in range(L))
    ct, tg = orc.gcm_encrypt(key, nonce, pt)
    want_wires = np.array(shape.witness(otarget.inputs(key, nonce, pt, ct, tg)), dtype=np.uint64)
    pw = pkg.PartialWitness()
    target.set_targets(pw, key, nonce, pt, ct, tg if tag else b"")
    st, wires = orc.OracleCircuit(data.blob).generate_witness(pw.map, OB.NUM_WIRES * shape.n)   # the product's witness PROGRAM, run by the oracle
    wires = np.array(wires, dtype=np.uint64).reshape(OB.NUM_WIRES, shape.n)
    assert st == 0 and (wires[:OB.NUM_ROUTED] == want_wires).all() and not wires[OB.NUM_ROUTED:].any()
    # a wrong ciphertext byte: the independent witness generator runs into the conflict, the product's program reports it
    bad = bytearray(ct)
    bad[L // 2] ^= 1
    with pytest.raises(ValueError):
        shape.witness(otarget.inputs(key, nonce, pt, bytes(bad), tg))
    pw = pkg.PartialWitness()
    target.set_targets(pw, key, nonce, pt, bytes(bad), tg if tag else b"")
    assert orc.OracleCircuit(data.blob).generate_witness(pw.map, OB.NUM_WIRES * shape.n)[0] != 0


def test_gate_count_fixture_has_an_independent_derivation():
    """tests/golden/gate_counts.json (written by the product's builder) for the smaller sizes of test_encrypt_report_sizes
    (circuit_gcm.rs:708-736), re-derived by the independent builder."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "gate_counts.json")))["sizes"]
    seen = 0
    for e in gold:
        if e["L"] > 64:
            continue
        ob = OB.Builder()
        OB.AesGcmTarget(ob, e["nk"], e["nk"] + 6, e["L"], False)
        assert ob.num_gates() == e["num_gates"], e
        assert ob.build().n == 1 << e["degree_bits"], e
        seen += 1
    assert seen >= 6


def test_selector_groups_follow_the_number_of_tables():
    """One table: five gate types share one selector polynomial.  Two tables: 3 + 8 - 1 > 9, so two."""
    ob = OB.Builder()
    sbox = OB.sbox_lut(ob)
    OB.add_virtual_byte_target(ob, sbox)
    assert ob.build().groups == [(0, 5)]
    ob = OB.Builder()
    xor_lut, mul_lut = OB.byte_xor_lut(ob), OB.gf_2_8_mul_lut(ob)
    x, y = ob.add_virtual_target(), ob.add_virtual_target()
    OB.byte_xor(ob, xor_lut, OB.gf_2_8_mul(ob, mul_lut, x, y), y)
    sh = ob.build()
    assert [g.name for g in sh.gates] == ["lookup", "lookup", "lookup_table", "lookup_table", "noop", "constant", "public_input", "arithmetic"]
    assert sh.groups == [(0, 7), (7, 8)]        # greedy: 5 + Constant + PublicInput fit under degree 9, ArithmeticGate (degree 3) does not


@pytest.mark.gpu
@pytest.mark.parametrize("nk,L,tag", [(4, 13, True), (4, 1024, False)])
def test_gpu_witness_equals_the_independent_derivation(pkg, orc, nk, L, tag):
    """The wire matrix the DEVICE generates (k_witness / k_fill_wires / k_lut_rows, read back through the C ABI) against the
    independent builder's witness -- no product code and no oracle_prover.h on the expected side."""
    data, target, shape, otarget = _both(pkg, nk, L, tag)
    key, nonce = bytes([42] * (4 * nk)), bytes([111] * 12)
    pts = [bytes((7 * i + 3 + 11 * k) & 255 for i in range(L)) for k in range(2)]
    pws, want = [], []
    for pt in pts:
        ct, tg = orc.gcm_encrypt(key, nonce, pt)
        pw = pkg.PartialWitness()
        target.set_targets(pw, key, nonce, pt, ct, tg if tag else b"")
        pws.append(pw)
        want.append(np.array(shape.witness(otarget.inputs(key, nonce, pt, ct, tg)), dtype=np.uint64))
    proofs, status = data.prove_batch(pws)
    assert status == [0, 0]
    for i in range(2):
        got = np.array(data.debug_read("wires", i, cap=OB.NUM_WIRES * shape.n), dtype=np.uint64)
        assert len(got) >= OB.NUM_ROUTED * shape.n
        assert (got[:OB.NUM_ROUTED * shape.n].reshape(OB.NUM_ROUTED, shape.n) == want[i]).all()
        assert not got[OB.NUM_ROUTED * shape.n:].any()
        data.verify(proofs[i])


def test_feistel_poseidon_circuit_equals_the_independent_derivation(pkg, orc):
    """feistel/src/circuit.rs:115 (feistel_poseidon_check: 32 rounds, one PoseidonGate row each): gate types and selector
    groups ([Noop, Constant, PublicInput, Arithmetic] | [Poseidon]), constants, sigma and all 135 wire columns, S-box input
    advice wires included."""
    data, pws = circuits.feistel_poseidon(pkg, [5])
    B = Blob(data.blob)
    ob = OB.Builder()
    state_t = [ob.add_virtual_target() for _ in range(8)]
    keys_t = [[ob.add_virtual_target() for _ in range(4)] for _ in range(32)]
    out_t = OB.feistel_cipher(ob, state_t, keys_t)
    shape = ob.build()
    assert B.n == shape.n and [g.name for g in shape.gates] == ["noop", "constant", "public_input", "arithmetic", "poseidon"]
    assert list(B.gates) == [2, 3, 4, 5, 6] and [tuple(g) for g in B.groups.tolist()] == shape.groups == [(0, 4), (4, 5)]
    want = np.array(shape.constants, dtype=np.uint64)
    assert want.shape == B.constants.shape and (want == B.constants).all()
    assert (np.array(shape.sigma_values(), dtype=np.uint64).reshape(OB.NUM_ROUTED, shape.n) == B.sigmas).all()
    # same inputs as circuits.feistel_poseidon(seed 5): the virtual targets were created in the same order on both sides
    inputs = {t: v for t, v in pws[0].map.items() if t < len(state_t) + 4 * len(keys_t)}
    assert len(inputs) == 8 + 128
    st, wires = orc.OracleCircuit(data.blob).generate_witness(pws[0].map, OB.NUM_WIRES * shape.n)
    assert st == 0
    got = np.array(wires, dtype=np.uint64).reshape(OB.NUM_WIRES, shape.n)
    want_wires = np.array(shape.witness(inputs), dtype=np.uint64)
    assert want_wires.shape == got.shape and (want_wires == got).all()
    # the cipher's output, read off the independent witness, is what the reference's native cipher gives (circuits.py checks
    # the round trip lib.rs:98 on it)
    outs = [v for t, v in pws[0].map.items() if t not in inputs]
    assert len(outs) == 8
    for t, v in zip(out_t, outs):
        row, col = OB.wire_rc(t)
        assert int(want_wires[col][row]) == v


class _Dual:
    """Drives the product's builder (through the C ABI) and the independent one with the same calls; a target is a pair."""

    def __init__(self, pkg):
        self.p, self.o = pkg.CircuitBuilder(), OB.Builder()

    def __getattr__(self, name):
        fp, fo = getattr(self.p, name), getattr(self.o, name)

        def call(*args):
            split = lambda k: [a[k] if isinstance(a, tuple) else ([x[k] for x in a] if isinstance(a, list) else a) for a in args]
            rp, ro = fp(*split(0)), fo(*split(1))
            if isinstance(rp, list):
                return list(zip(rp, ro))
            return None if rp is None else (rp, ro)
        return call


@pytest.mark.parametrize("seed", range(24))
def test_random_circuits_compile_identically(pkg, orc, seed):
    """Differential fuzz of the two builders over the whole generic vocabulary -- raw `arithmetic` with degenerate constants,
    constant operands (every branch of `arithmetic_special_cases`), repeated operations (the result cache), add / sub / mul /
    mul_const_add / select / is_equal / connect, lookups into one to three tables, Poseidon sponges -- then the same
    comparison as for the AES circuits: gate count, gate types, selector groups, constants, sigma, and the witness."""
    import random
    r = random.Random(1000 + seed)
    P = OB.P
    d = _Dual(pkg)
    tables = [[(i, (i * 7 + 3) & 0xFF) for i in range(256)], [(i, i >> 1) for i in range(256)], [((x << 3) + i, (x >> i) & 1) for x in range(32) for i in range(8)]]
    luts = [(d.p.add_lookup_table_from_pairs(t), d.o.add_lookup_table_from_pairs(t)) for t in tables[:r.randrange(0, 4)]]
    assert all(a == b for a, b in luts)
    inputs = [d.add_virtual_target() for _ in range(r.randrange(2, 6))]
    nodes = list(inputs) + [d.constant(c) for c in (0, 1, 2, P - 1, r.randrange(P))]
    small = list(inputs)                           # targets whose value is inside every table's domain [0, 256)
    used = set()
    for _ in range(r.randrange(10, 120)):
        kind = r.choice(["arith", "arith", "add", "sub", "mul", "mca", "select", "eq", "connect", "lookup", "lookup", "hash", "repeat"])
        x, y, z = (r.choice(nodes) for _ in range(3))
        if kind == "arith":
            t = d.arithmetic(r.choice([0, 1, P - 1, 5, r.randrange(P)]), r.choice([0, 1, P - 1, r.randrange(P)]), x, y, z)
        elif kind == "repeat":
            t = d.arithmetic(1, 1, x, y, z)
            assert d.arithmetic(1, 1, x, y, z) == t
        elif kind in ("add", "sub", "mul"):
            t = getattr(d, kind)(x, y)
        elif kind == "mca":
            t = d.mul_const_add(r.choice([1, 2, 256, P - 2, r.randrange(P)]), x, y)
        elif kind == "select":
            t = d.select(d.is_equal(x, y), z, x)
        elif kind == "eq":
            t = d.is_equal(x, y)
        elif kind == "connect":
            t = d.add(x, y)
            d.connect(t, d.add(y, x))
        elif kind == "lookup":
            if not luts:
                continue
            k = r.randrange(len(luts))
            t = d.add_lookup_from_index(r.choice(small) if r.random() < 0.9 else x, luts[k][0])
            small.append(t)
            used.add(k)
        else:
            outs = d.hash_n_to_m_no_pad([r.choice(nodes) for _ in range(r.randrange(1, 11))], r.randrange(1, 10))
            nodes += outs
            continue
        nodes.append(t)
    for k in range(len(luts)):                     # an unused table is an error in both builders: use each once
        if k not in used:
            d.add_lookup_from_index(inputs[0], luts[k][0])
    assert d.p.num_gates() == d.o.num_gates()
    data, shape = d.p.build(), d.o.build()
    B = Blob(data.blob)
    names = {"lookup": 0, "lookup_table": 1, "noop": 2, "constant": 3, "public_input": 4, "arithmetic": 5, "poseidon": 6}
    assert B.n == shape.n and list(B.gates) == [names[g.name] for g in shape.gates]
    assert [tuple(g) for g in B.groups.tolist()] == shape.groups
    want = np.array(shape.constants, dtype=np.uint64)
    assert want.shape == B.constants.shape and (want == B.constants).all()
    assert (np.array(shape.sigma_values(), dtype=np.uint64).reshape(OB.NUM_ROUTED, shape.n) == B.sigmas).all()
    assert [tuple(x) for x in B.lookup_rows.tolist()] == shape.b.lookup_rows
    # witness on small inputs (so that lookups can hit); a circuit whose random structure is unsatisfiable must fail on both sides
    vals = [r.randrange(32) for _ in inputs]
    oc = orc.OracleCircuit(data.blob)
    st, wires = oc.generate_witness({t[0]: v for t, v in zip(inputs, vals)}, OB.NUM_WIRES * shape.n)
    try:
        want_wires = np.array(shape.witness({t[1]: v for t, v in zip(inputs, vals)}), dtype=np.uint64)
    except ValueError:
        assert st != 0
        return
    assert st == 0
    got = np.array(wires, dtype=np.uint64).reshape(OB.NUM_WIRES, shape.n)
    assert (got[:want_wires.shape[0]] == want_wires).all() and not got[want_wires.shape[0]:].any()

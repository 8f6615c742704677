"""The circuits and fixed inputs behind tests/golden/proof_digests.json (written by tools/make_proof_digests.py), shared by the
generator, the CPU test (oracle digest == fixture) and the GPU test (GPU proof digest == fixture).

Two groups.  `cases`: seven small circuit tests of the reference, re-proved by the oracle in the CPU suite.  `large_cases`
(round 3): the three GPU workloads of BASELINE.json -- AesGcm128Target<1024> (the bench circuit), the ecGFp5 ElGamal encryption
circuit and AesGcm128Target<65536> (n = 2^19 rows) -- on the reference's deterministic inputs (key [42;16], nonce [111;12],
plaintext [42;L], aes-gcm/src/circuit_gcm.rs:750-752).  Their fixture entries were written ONCE by the oracle in the build
container (the 2^19-row proof takes the oracle tens of minutes; the wall time is recorded in the entry) and carry, besides the
proof digest, a digest of every intermediate stage in pipeline order, so that a GPU proof that differs names its stage without
the oracle in the loop."""
import hashlib
import json
import os
import struct

import circuits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ZK_KEY = [0x0123456789ABCDEF, 0x1111111111111111, 0x2222222222222222, 0x3333333333333333]
LARGE = ("aes_gcm_128_1024", "elgamal_encrypt", "aes_gcm_128_65536")


def cases(pkg):
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "aes_kat.json")))
    yield "gf_2_8_mul_57_13", circuits.gf_2_8_mul(pkg, [(0x57, 0x13, 0xFE)])[:2]
    yield "encrypt_block_fips197", circuits.encrypt_block(pkg, bytes.fromhex(kat["fips197_block"]["key"]), bytes.fromhex(kat["fips197_block"]["input"]),
                                                           expected=bytes.fromhex(kat["fips197_block"]["output"]))[:2]
    yield "aes_gcm_128_13", circuits.encrypt(pkg, 4, 13, False)[:2]
    yield "aes_gcm_128_13_tag", circuits.encrypt(pkg, 4, 13, True)[:2]
    yield "arithmetic_only", circuits.arithmetic_only(pkg, [(3, 5, 11, 92)])[:2]
    yield "poseidon_cipher_L3", circuits.poseidon_encrypt(pkg, 3, [11])[:2]
    yield "zk_example_aes_gcm_128", circuits.zk_example_aes_gcm_128(pkg)[:2]


def large_case(pkg, name):
    """(CircuitData, [PartialWitness]) of one BASELINE.json GPU workload on the reference's fixed test inputs."""
    if name == "aes_gcm_128_1024":      # BASELINE.json configs[2]; aes-gcm/src/lib.rs:19, inputs circuit_gcm.rs:750-752
        return circuits.encrypt(pkg, 4, 1024, False)[:2]
    if name == "aes_gcm_128_65536":     # configs[4]: the deep circuit, n = 2^19 rows
        return circuits.encrypt(pkg, 4, 65536, False)[:2]
    if name == "elgamal_encrypt":       # configs[3]; ecgfp5/src/elgamal/circuit.rs:66-97 with seeded instead of OsRng inputs
        return circuits.ecgfp5_elgamal(pkg, [1])[:2]
    raise KeyError(name)


def digest_case(O, data, pw, zk):
    oc = O.OracleCircuit(data.blob)
    if zk:
        oc.set_zk_key(ZK_KEY, 0)
    st, proof = oc.prove(pw.map)
    assert st == 0
    vd = oc.verifier_data()
    return {"blob_sha256": hashlib.sha256(data.blob).hexdigest(), "verifier_data_sha256": hashlib.sha256(struct.pack("<%dQ" % len(vd), *vd)).hexdigest(),
            "proof_sha256": hashlib.sha256(proof).hexdigest(), "proof_bytes": len(proof), "degree_bits": data.info["degree_bits"]}


def sha_words(words):
    return hashlib.sha256(struct.pack("<%dQ" % len(words), *words)).hexdigest()


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def _deinterleave(b):
    """[(c0, c1), ...] extension elements -> the two component columns one after the other (the device layout)."""
    import numpy as np
    a = np.frombuffer(b, dtype="<u8").reshape(-1, 2)
    return np.ascontiguousarray(a.T).tobytes()


# challenge block of the device (csrc/kernels.h ChalSlot) <-> oracle trace names, in transcript order
def _chal_slices(nr):
    return [("betas|gammas", 0, 4), ("deltas", 4, 12), ("alphas", 12, 14), ("zeta", 14, 16), ("fri_alpha", 16, 18), ("fri_betas", 18, 18 + 2 * nr),
            ("pow_witness", 34, 35), ("query_indices", 36, 36 + 28)]


STAGE_ORDER = ["wires", "wires_cap", "betas|gammas", "deltas", "zs", "zs_cap", "alphas", "quotient_coeffs", "quotient_cap", "zeta", "fri_alpha",
               "fri_final_poly_in", "fri_betas", "pow_witness", "query_indices"]


def stage_digests_oracle(oc, info, live_wire_words):
    """SHA-256 of every traced stage of the oracle's last prove(trace=True).  `wires` covers the first `live_wire_words` words
    (the wire columns the device materialises); the rest of the oracle's matrix must be zero."""
    out = {}
    w = oc.trace_bytes("wires")
    assert not any(w[8 * live_wire_words:]), "a wire column the device does not materialise is not identically zero"
    out["wires"] = _sha(w[: 8 * live_wire_words])
    for name in ("wires_cap", "zs", "zs_cap", "quotient_coeffs", "quotient_cap"):
        out[name] = _sha(oc.trace_bytes(name))
    out["fri_final_poly_in"] = _sha(_deinterleave(oc.trace_bytes("fri_final_poly_in")))
    out["betas|gammas"] = _sha(oc.trace_bytes("betas") + oc.trace_bytes("gammas"))
    for name in ("deltas", "alphas", "zeta", "fri_alpha", "fri_betas", "pow_witness", "query_indices"):
        out[name] = _sha(oc.trace_bytes(name))
    return out


def stage_digests_gpu(data, index):
    """The same digests from the device buffers of proof `index` of the last batch."""
    n, nr = 1 << data.info["degree_bits"], data.info["num_fri_rounds"]
    cap = 135 * n + 16
    out = {}
    for name in ("wires", "wires_cap", "zs", "zs_cap", "quotient_coeffs", "quotient_cap", "fri_final_poly_in"):
        out[name] = _sha(data.debug_read_bytes(name, index, cap=cap))
    ch = data.debug_read_bytes("challenges", index, cap=256)
    has_lookups = data.info["num_luts"] > 0
    for name, lo, hi in _chal_slices(nr):
        out[name] = _sha(ch[8 * lo: 8 * hi] if (name != "deltas" or has_lookups) else b"")
    return out


def fixture():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "proof_digests.json")))

"""The circuits and fixed inputs behind tests/golden/proof_digests.json (written by tools/make_proof_digests.py), shared by the
generator, the CPU test (oracle digest == fixture) and the GPU test (GPU proof digest == fixture)."""
import hashlib
import json
import os
import struct

import circuits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ZK_KEY = [0x0123456789ABCDEF, 0x1111111111111111, 0x2222222222222222, 0x3333333333333333]


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


def fixture():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "proof_digests.json")))

#!/usr/bin/env python3
"""Writes tests/golden/aes_kat.json: the known-answer vectors that the reference's own tests hold for the
native AES / GCM path (data only -- inputs and expected outputs):

  FIPS-197 App. B block vector          aes-gcm/src/native_aes.rs:209-220 (also circuit_aes.rs:621-629)
  FIPS-197 App. A key-schedule prefixes native_aes.rs:167-203 (keys also circuit_aes.rs:550-569)
  GF(2^8) products (FIPS-197 4.2)       aes-gcm/src/circuit_aes.rs:488-498
  NIST CAVP AES-128-GCM vectors         aes-gcm/src/native_gcm.rs:290-329
  deterministic circuit-test inputs     circuit_gcm.rs:476-480, 586-587, 651-652, 750-752; examples/aes_gcm_128.rs:20-22

Expected outputs of the last group are not stored in the reference (it computes them); they are produced here by
the oracle's AES restatement AFTER it has passed every vector above, and cross-checked against the OpenSSL CLI
when it is present.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

kat = {
    "source": "vectors copied as data from the reference's tests (see tools/make_golden.py docstring for file:line)",
    "fips197_block": {"key": "2b7e151628aed2a6abf7158809cf4f3c", "input": "3243f6a8885a308d313198a2e0370734",
                      "output": "3925841d02dc09fbdc118597196a0b32"},
    "key_expansion_prefix": [
        {"key": "2b7e151628aed2a6abf7158809cf4f3c", "words": ["2b7e1516", "28aed2a6", "abf71588", "09cf4f3c"]},
        {"key": "8e73b0f7da0e6452c810f32b809079e562f8ead2522c6b7b",
         "words": ["8e73b0f7", "da0e6452", "c810f32b", "809079e5", "62f8ead2", "522c6b7b"]},
        {"key": "603deb1015ca71be2b73aef0857d77811f352c073b6108d72d9810a30914dff4",
         "words": ["603deb10", "15ca71be", "2b73aef0", "857d7781", "1f352c07", "3b6108d7", "2d9810a3", "0914dff4"]},
    ],
    "gf_2_8_mul": [[0x57, 0x01, 0x57], [0x57, 0x02, 0xae], [0x57, 0x04, 0x47], [0x57, 0x08, 0x8e], [0x57, 0x10, 0x07],
                   [0x57, 0x20, 0x0e], [0x57, 0x40, 0x1c], [0x57, 0x80, 0x38], [0x57, 0x13, 0xfe]],
    "cavp_gcm128": [
        {"key": "cf063a34d4a9a76c2c86787d3f96db71", "iv": "113b9785971864c83b01c787", "pt": "", "ct": "",
         "tag": "72ac8493e3a5228b5d130a69d2510e42"},
        {"key": "e98b72a9881a84ca6b76e0f43e68647a", "iv": "8b23299fde174053f3d652ba", "pt": "28286a321293253c3e0aa2704a278032",
         "ct": "5a3c1cf1985dbb8bed818036fdd5ab42", "tag": "23c7ab0f952b7091cd324835043b5eb5"},
        {"key": "387218b246c1a8257748b56980e50c94", "iv": "dd7e014198672be39f95b69d", "pt": "48f5b426baca03064554cc2b30",
         "ct": "cdba9e73eaf3d38eceb2b04a8d", "tag": "ecf90f4a47c9c626d6fb2c765d201556"},
        {"key": "bfd414a6212958a607a0f5d3ab48471d", "iv": "86d8ea0ab8e40dcc481cd0e2",
         "pt": "a6b76a066e63392c9443e60272ceaeb9d25c991b0f2e55e2804e168c05ea591a",
         "ct": "62171db33193292d930bf6647347652c1ef33316d7feca99d54f1db4fcf513f8", "tag": "c28280aa5c6c7a8bd366f28c1cfd1f6e"},
    ],
}

# pin the restatement before using it to derive anything
v = kat["fips197_block"]
assert O.encrypt_block(bytes.fromhex(v["key"]), bytes.fromhex(v["input"])).hex() == v["output"]
for v in kat["cavp_gcm128"]:
    ct, tag = O.gcm_encrypt(bytes.fromhex(v["key"]), bytes.fromhex(v["iv"]), bytes.fromhex(v["pt"]))
    assert ct.hex() == v["ct"] and tag.hex() == v["tag"], v

# derived expectations for the reference's deterministic circuit-test inputs
derived = []
for nk in (4, 8):
    for L in (13, 17):
        key, nonce, pt = bytes([42] * (4 * nk)), bytes([111] * 12), bytes([42] * L)
        ct, tag = O.gcm_encrypt(key, nonce, pt)
        derived.append({"what": "test_encrypt inputs (circuit_gcm.rs:750-752)", "key": key.hex(), "iv": nonce.hex(), "pt": pt.hex(),
                        "ct": ct.hex(), "tag": tag.hex()})
key, nonce, pt = bytes([123] * 16), bytes([0] * 12), bytes([231] * 42)
ct, tag = O.gcm_encrypt(key, nonce, pt)
derived.append({"what": "example key/pt (examples/aes_gcm_128.rs:20-22), zero nonce", "key": key.hex(), "iv": nonce.hex(), "pt": pt.hex(),
                "ct": ct.hex(), "tag": tag.hex()})
kat["derived_by_pinned_oracle"] = derived

# optional cross-check against the OpenSSL CLI (AES-128-ECB single block)
try:
    v = kat["fips197_block"]
    out = subprocess.run(["openssl", "enc", "-aes-128-ecb", "-nopad", "-K", v["key"]], input=bytes.fromhex(v["input"]),
                         capture_output=True, check=True).stdout
    kat["openssl_cli_crosscheck"] = out.hex() == v["output"]
except Exception as e:  # noqa: BLE001
    kat["openssl_cli_crosscheck"] = "unavailable: %s" % type(e).__name__

os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
with open(os.path.join(ROOT, "tests", "golden", "aes_kat.json"), "w") as f:
    json.dump(kat, f, indent=1)
print("wrote tests/golden/aes_kat.json; openssl:", kat["openssl_cli_crosscheck"])

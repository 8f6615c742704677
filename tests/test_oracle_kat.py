"""CPU tests: the oracle against every golden vector the reference's tests hold for this path, and against
independent big-integer arithmetic; the product's native cipher against the same vectors."""
import ctypes as C
import json
import os
import random

import pytest

P = 0xFFFFFFFF00000001
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "aes_kat.json")))


def test_oracle_field_vs_bigint(orc):
    L, r = orc.lib(), random.Random(1)
    xs = [0, 1, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000] + [r.randrange(P) for _ in range(200)]
    for a in xs:
        for b in xs[:12] + [r.randrange(P) for _ in range(8)]:
            assert L.orc_fmul(a, b) == a * b % P
            assert L.orc_fadd(a, b) == (a + b) % P
            assert L.orc_fsub(a, b) == (a - b) % P
    for a in xs[1:40]:
        assert L.orc_fmul(L.orc_finv(a), a) == 1


def test_poseidon_upstream_test_vector(orc):
    # plonky2 poseidon_goldilocks.rs test_vectors: all-zero input (first and last words, SURVEY.md C.3 tripwire)
    st = (C.c_uint64 * 12)()
    orc.lib().orc_poseidon(st)
    assert st[0] == 0x3C18A9786CB0B359 and st[1] == 0xC4055E3364A246C3 and st[11] == 0x1792B1C4342109D7


def test_sparse_partial_rounds_are_the_same_permutation(orc, pkg):
    """oracle/oracle_poseidon_sparse.h (the faster permutation bench.py's cpu_baseline leg switches on): derived at start-up from
    the MDS matrix and the round constants, it must be the SAME function as the textbook 30-round loop -- upstream's all-zero
    vector, extreme words, 2000 random states -- and a whole proof must come out byte-identical with it switched on."""
    L, r = orc.lib(), random.Random(7)
    z = (C.c_uint64 * 12)()
    L.orc_poseidon_sparse(z)
    assert z[0] == 0x3C18A9786CB0B359 and z[11] == 0x1792B1C4342109D7
    states = [[0] * 12, [P - 1] * 12, [1] + [0] * 11, list(range(12))] + [[r.randrange(P) for _ in range(12)] for _ in range(2000)]
    for st in states:
        a, b = (C.c_uint64 * 12)(*st), (C.c_uint64 * 12)(*st)
        L.orc_poseidon(a)
        L.orc_poseidon_sparse(b)
        assert list(a) == list(b)
    import circuits
    data, pws = circuits.gf_2_8_mul(pkg, [(0x57, 0x13, 0xFE)])
    oc = orc.OracleCircuit(data.blob)
    st, ref = oc.prove(pws[0].map)
    try:
        L.orc_set_fast_hash(1)
        for lanes in (8, 4, 1):  # AVX-512, AVX2 (where the host has them) and scalar forms of the fast hash
            L.orc_set_simd_lanes(lanes)
            st2, fast = orc.OracleCircuit(data.blob).prove(pws[0].map)
            assert st == 0 and st2 == 0 and fast == ref, lanes
    finally:
        L.orc_set_fast_hash(0)
        L.orc_set_simd_lanes(8)


EDGE_WORDS = [0, 1, 2, 0xFFFFFFFE, 0xFFFFFFFF, 1 << 32, (1 << 32) + 1, (1 << 33) - 1, (1 << 63) - 1, 1 << 63, P - 2, P - 1, P, P + 1,
              (1 << 64) - (1 << 32), (1 << 64) - 2, (1 << 64) - 1, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF, 0xFFFFFFFEFFFFFFFF]


@pytest.mark.parametrize("lanes", [8, 4])
def test_simd_lane_arithmetic_on_arbitrary_words(orc, lanes):
    """oracle/oracle_poseidon_simd.h works on ARBITRARY 64-bit representatives and pays every wrap of 2^64 back at once; the
    carries that decide it are 2^-32 events on random data, so the operations are held to big-integer arithmetic on every triple
    of edge words (around 0, 2^32, 2^63, p, 2^64) and on random words."""
    L, r = orc.lib(), random.Random(11)
    triples = [(a, b, c) for a in EDGE_WORDS for b in EDGE_WORDS for c in EDGE_WORDS]
    triples += [(r.getrandbits(64), r.getrandbits(64), r.getrandbits(64)) for _ in range(4000)]
    while len(triples) % lanes:
        triples.append((0, 0, 0))
    ran = False
    for k in range(0, len(triples), lanes):
        chunk = triples[k:k + lanes]
        a, b, c = ((C.c_uint64 * lanes)(*[t[j] for t in chunk]) for j in range(3))
        out = (C.c_uint64 * (6 * lanes))()
        if not L.orc_simd_test_arith(lanes, a, b, c, out):
            pytest.skip("host CPU cannot run %d lanes" % lanes)
        ran = True
        for l, (x, y, z) in enumerate(chunk):
            want = [x * y % P, x * x % P, (x * y + z) % P, (x * y + y * z + z * x) % P, pow(x, 7, P), (x + chunk[0][2] % P) % P]
            got = [out[j * lanes + l] for j in range(6)]
            assert got == want, (hex(x), hex(y), hex(z), got, want)
    assert ran


@pytest.mark.parametrize("lanes", [8, 4, 1])
def test_simd_hash_is_the_textbook_hash(orc, lanes):
    """Merkle caps through the lane-parallel sparse Poseidon (the cpu_baseline leg's hash) equal the textbook ones: leaf widths on
    both sides of the rate (5, 8, 9, 16, 17, 135: one to seventeen permutations per leaf, short last block), leaf counts that
    leave a scalar tail or none, extreme words in the leaves; then a whole proof."""
    L, r = orc.lib(), random.Random(5 + lanes)
    cases = [(64, 5), (64, 8), (64, 9), (128, 16), (32, 17), (256, 135), (16, 4), (8, 3), (2, 135), (4, 9)]
    try:
        for leaves, width in cases:
            words = [r.choice(EDGE_WORDS[:12]) if r.random() < 0.2 else r.randrange(P) for _ in range(leaves * width)]
            buf = (C.c_uint64 * len(words))(*words)
            caps = []
            for fast in (0, 1):
                L.orc_set_fast_hash(fast)
                got = L.orc_set_simd_lanes(lanes)
                if fast and lanes > 1 and got != lanes:
                    pytest.skip("host CPU cannot run %d lanes" % lanes)
                out = (C.c_uint64 * (4 * 16))()
                n = L.orc_merkle_cap(buf, leaves, width, 2, out)
                caps.append(list(out)[: 4 * n])
            assert caps[0] == caps[1], (leaves, width)
    finally:
        L.orc_set_fast_hash(0)
        L.orc_set_simd_lanes(8)


@pytest.mark.parametrize("lanes", [8, 4, 1])
def test_bit_reversed_output_transform_is_the_textbook_transform(orc, lanes):
    """fft_bitrev_out (the commitments' transform: decimation in frequency, no permutation pass, SIMD butterflies under the fast
    switch) leaves X[k] of the textbook fft_inplace at position rev(k) -- sizes 2^0 .. 2^12, extreme and random words."""
    L, r = orc.lib(), random.Random(17 + lanes)
    try:
        for fast in (0, 1):
            L.orc_set_fast_hash(fast)
            L.orc_set_simd_lanes(lanes)
            for bits in list(range(0, 9)) + [12]:
                n = 1 << bits
                v = [r.choice([0, 1, P - 1, P - 2, 0xFFFFFFFF, 1 << 32]) if r.random() < 0.2 else r.randrange(P) for _ in range(n)]
                a, b = (C.c_uint64 * n)(*v), (C.c_uint64 * n)(*v)
                L.orc_fft(a, bits, 0)
                L.orc_fft_bitrev_out(b, bits)
                rev = [int(format(i, "0%db" % bits)[::-1], 2) if bits else 0 for i in range(n)]
                assert [b[rev[k]] for k in range(n)] == list(a), (fast, bits)
    finally:
        L.orc_set_fast_hash(0)
        L.orc_set_simd_lanes(8)


def test_batch_inverse_is_elementwise_inverse(orc):
    """One inversion per block (the permutation and lookup stages of the oracle): equal to finv element by element, zeros stay
    zero (finv(0) = 0), block boundaries (1024) and sizes 0, 1 covered."""
    L, r = orc.lib(), random.Random(3)
    for n in (0, 1, 2, 1023, 1024, 1025, 5000):
        v = [0 if r.random() < 0.05 else r.randrange(P) for _ in range(n)]
        if n > 2:
            v[0], v[-1] = 0, P - 1
        buf = (C.c_uint64 * max(n, 1))(*v)
        L.orc_batch_inverse(buf, n)
        assert list(buf)[:n] == [pow(x, P - 2, P) for x in v]


def test_poseidon_constants_rederive():
    # the committed .inc files are what tools/gen_poseidon_constants.py derives (ChaCha8 seed 0, rand-0.8 gen_range)
    import importlib.util
    root = os.path.dirname(os.path.dirname(__file__))
    spec = importlib.util.spec_from_file_location("gen", os.path.join(root, "tools", "gen_poseidon_constants.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rc = gen.constants()
    assert rc[0] == 0xB585F766F2144405 and rc[1] == 0x7746A55F43921AD7
    for path in ("oracle/poseidon_rc.inc", "plonky2-aes_amd/csrc/poseidon_rc.inc"):
        txt = open(os.path.join(root, path)).read()
        got = [int(t.rstrip("ULL,"), 16) for t in txt.split() if t.startswith("0x")]
        assert got == rc
    assert gen.permute([0] * 12, rc)[0] == 0x3C18A9786CB0B359


def test_oracle_fft_matches_naive_dft(orc):
    r, bits = random.Random(2), 4
    n = 1 << bits
    w = pow(7277203076849721926, 1 << (32 - bits), P)
    x = [r.randrange(P) for _ in range(n)]
    a = (C.c_uint64 * n)(*x)
    orc.lib().orc_fft(a, bits, 0)
    assert list(a) == [sum(x[j] * pow(w, j * k, P) for j in range(n)) % P for k in range(n)]
    orc.lib().orc_fft(a, bits, 1)
    assert list(a) == x


def test_oracle_lde_is_evaluation_on_coset(orc):
    r, bits = random.Random(3), 3
    n, N, g = 8, 64, 14293326489335486720
    c = [r.randrange(P) for _ in range(n)]
    out = (C.c_uint64 * N)()
    orc.lib().orc_lde((C.c_uint64 * n)(*c), bits, 3, out)
    w = pow(7277203076849721926, 1 << (32 - 6), P)
    for i in range(N):
        x = g * pow(w, i, P) % P
        rev = int(format(i, "06b")[::-1], 2)
        assert out[rev] == sum(ci * pow(x, k, P) for k, ci in enumerate(c)) % P


def _both(pkg, orc):
    return [("oracle", orc.encrypt_block, orc.gcm_encrypt), ("native", pkg.native.encrypt_block, pkg.native.gcm_encrypt)]


def test_fips197_block(pkg, orc):
    v = GOLD["fips197_block"]
    for name, enc, _ in _both(pkg, orc):
        assert enc(bytes.fromhex(v["key"]), bytes.fromhex(v["input"])).hex() == v["output"], name


def test_key_expansion_prefix(pkg, orc):
    for v in GOLD["key_expansion_prefix"]:
        key = bytes.fromhex(v["key"])
        nk = len(key) // 4
        out = C.create_string_buffer(16 * (nk + 7))
        orc.lib().orc_aes_expand_key(key, nk, out)
        for w in (out.raw, pkg.native.key_expansion(key)):
            assert [w[4 * i:4 * i + 4].hex() for i in range(nk)] == v["words"]
        assert out.raw == pkg.native.key_expansion(key)


def test_gf_2_8_products(pkg, orc):
    for a, b, e in GOLD["gf_2_8_mul"]:
        assert orc.lib().orc_gf_2_8_mul(a, b) == e and pkg.native.gf_2_8_mul(a, b) == e


def test_cavp_gcm_vectors(pkg, orc):
    for v in GOLD["cavp_gcm128"] + GOLD["derived_by_pinned_oracle"]:
        for name, _, gcm in _both(pkg, orc):
            ct, tag = gcm(bytes.fromhex(v["key"]), bytes.fromhex(v["iv"]), bytes.fromhex(v["pt"]))
            assert ct.hex() == v["ct"] and tag.hex() == v["tag"], name


def test_native_matches_oracle_random(pkg, orc):
    # the reference's differential tests (native_gcm.rs:334-374) use these plaintext lengths, AES-128 and AES-256
    r = random.Random(5)
    for nk in (4, 6, 8):
        for L in (0, 16, 32, 1, 17, 14, 26, 131, 1027, 4242):
            key, iv, pt = bytes(r.randrange(256) for _ in range(4 * nk)), bytes(r.randrange(256) for _ in range(12)), bytes(r.randrange(256) for _ in range(L))
            assert pkg.native.gcm_encrypt(key, iv, pt) == orc.gcm_encrypt(key, iv, pt)
    x, y = bytes(r.randrange(256) for _ in range(16)), bytes(r.randrange(256) for _ in range(16))
    o = C.create_string_buffer(16)
    orc.lib().orc_gf_2_128_mul(x, y, o)
    assert o.raw == pkg.native.gf_2_128_mul(x, y)
    data = bytes(r.randrange(256) for _ in range(64))
    orc.lib().orc_ghash(x, data, 64, o)
    assert o.raw == pkg.native.ghash(x, data)

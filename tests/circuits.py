"""Circuit + witness constructors that mirror the reference's own circuit tests one for one.

Each function returns (CircuitData, [PartialWitness, ...]) with the expected outputs written into the witness as
well as the inputs -- the reference's correctness mechanism: a wrong expected value makes
witness generation fail, i.e. `data.prove(pw)` is Err.
"""
import random

SBOX = None


def _sbox(orc):
    global SBOX
    if SBOX is None:
        SBOX = [orc.lib().orc_sbox(x) for x in range(256)]
    return SBOX


def st_index(i, j):  # StateTarget.0[i][j] <-> flat 4*i + j
    return 4 * i + j


def assert_byte(pkg, values):  # aes-gcm/src/circuit_aes.rs:385 test_assert_byte
    b = pkg.CircuitBuilder()
    lut = b.sbox_lut()
    t = b.add_virtual_byte_target(lut)
    data = b.build()
    pws = []
    for v in values:
        pw = pkg.PartialWitness()
        pw.set_target(t, v)
        pws.append(pw)
    return data, pws


def sub_bytes(pkg, orc, states):  # circuit_aes.rs:414 test_sub_bytes
    b = pkg.CircuitBuilder()
    s = b.add_virtual_state_target_unsafe()
    lut = b.sbox_lut()
    out = b.state_sub_bytes(lut, s)
    data = b.build()
    S = _sbox(orc)
    pws = []
    for st in states:
        pw = pkg.PartialWitness()
        pw.set_state_target(s, st)
        pw.set_state_target(out, [S[x] for x in st])
        pws.append(pw)
    return data, pws


def _mix_columns_native(pkg, st):
    g = pkg.native.gf_2_8_mul
    s = [[st[st_index(i, j)] for j in range(4)] for i in range(4)]
    r = [[0] * 4 for _ in range(4)]
    for c in range(4):
        r[0][c] = g(2, s[0][c]) ^ g(3, s[1][c]) ^ s[2][c] ^ s[3][c]
        r[1][c] = s[0][c] ^ g(2, s[1][c]) ^ g(3, s[2][c]) ^ s[3][c]
        r[2][c] = s[0][c] ^ s[1][c] ^ g(2, s[2][c]) ^ g(3, s[3][c])
        r[3][c] = g(3, s[0][c]) ^ s[1][c] ^ s[2][c] ^ g(2, s[3][c])
    return [r[i][j] for i in range(4) for j in range(4)]


def mix_columns(pkg, states):  # circuit_aes.rs:442 test_mix_columns
    b = pkg.CircuitBuilder()
    xl, ml = b.byte_xor_lut(), b.gf_2_8_mul_lut()
    s = b.add_virtual_state_target_unsafe()
    out = b.state_mix_columns(xl, ml, s)
    data = b.build()
    pws = []
    for st in states:
        pw = pkg.PartialWitness()
        pw.set_state_target(s, st)
        pw.set_state_target(out, _mix_columns_native(pkg, st))
        pws.append(pw)
    return data, pws


def gf_2_8_mul(pkg, triples):  # circuit_aes.rs:473 test_gf_2_8_mul
    b = pkg.CircuitBuilder()
    lut = b.gf_2_8_mul_lut()
    x, y = b.add_virtual_byte_target_unsafe(), b.add_virtual_byte_target_unsafe()
    xy = b.gf_2_8_mul(lut, x, y)
    data = b.build()
    pws = []
    for a, c, e in triples:
        pw = pkg.PartialWitness()
        pw.set_byte_target(x, a)
        pw.set_byte_target(y, c)
        pw.set_byte_target(xy, e)
        pws.append(pw)
    return data, pws


def gf_2_8_add(pkg, pairs):  # circuit_aes.rs:513 test_gf_2_8_add
    b = pkg.CircuitBuilder()
    lut = b.byte_xor_lut()
    x, y = b.add_virtual_byte_target_unsafe(), b.add_virtual_byte_target_unsafe()
    xy = b.gf_2_8_add(lut, x, y)
    data = b.build()
    pws = []
    for a, c in pairs:
        pw = pkg.PartialWitness()
        pw.set_byte_target(x, a)
        pw.set_byte_target(y, c)
        pw.set_byte_target(xy, a ^ c)
        pws.append(pw)
    return data, pws


def key_expansion(pkg, key):  # circuit_aes.rs:548 test_key_expansion
    nk = len(key) // 4
    nr = nk + 6
    b = pkg.CircuitBuilder()
    kt = [b.add_virtual_byte_target_unsafe() for _ in range(4 * nk)]
    xl, sl = b.byte_xor_lut(), b.sbox_lut()
    ek = b.key_expansion(nk, nr, xl, sl, kt)
    data = b.build()
    w = pkg.native.key_expansion(key)
    pw = pkg.PartialWitness()
    for t, v in zip(kt, key):
        pw.set_byte_target(t, v)
    for t, v in zip(ek, w):
        pw.set_byte_target(t, v)
    return data, [pw]


def encrypt_block(pkg, key, block, expected=None):  # circuit_aes.rs:619 test_encrypt_block_test_vector
    nk = len(key) // 4
    nr = nk + 6
    b = pkg.CircuitBuilder()
    kt = [b.add_virtual_byte_target_unsafe() for _ in range(4 * nk)]
    xl, ml, sl = b.byte_xor_lut(), b.gf_2_8_mul_lut(), b.sbox_lut()
    ek = b.key_expansion(nk, nr, xl, sl, kt)
    ist = b.add_virtual_state_target(sl)
    out = b.encrypt_block(nr, xl, ml, sl, ist, ek)
    data = b.build()
    ct = expected if expected is not None else pkg.native.encrypt_block(key, block)
    w = pkg.native.key_expansion(key)
    pw = pkg.PartialWitness()
    for t, v in zip(kt, key):
        pw.set_byte_target(t, v)
    pw.set_state_target(ist, [block[i + 4 * j] for i in range(4) for j in range(4)])
    for t, v in zip(ek, w):
        pw.set_byte_target(t, v)
    pw.set_state_target(out, [ct[i + 4 * j] for i in range(4) for j in range(4)])
    return data, [pw]


def right_shift_one(pkg):  # circuit_gcm.rs:537 test_right_shift_one
    x = [111] * 16
    exp, carry = [], 0
    for v in x:
        exp.append((v >> 1) | (carry << 7))
        carry = v & 1
    b = pkg.CircuitBuilder()
    lut = b.u8_unit_right_shift_lut()
    xt = [b.add_virtual_byte_target_unsafe() for _ in range(16)]
    out = b.right_shift_one(lut, xt)
    data = b.build()
    pw = pkg.PartialWitness()
    for t, v in zip(xt, x):
        pw.set_byte_target(t, v)
    for t, v in zip(out, exp):
        pw.set_byte_target(t, v)
    return data, [pw]


def gctr(pkg, nk, L):  # circuit_gcm.rs:460 test_gctr
    nr = nk + 6
    key, icb, pt = bytes([42] * (4 * nk)), bytes([222] * 16), bytes([42] * L)
    b = pkg.CircuitBuilder()
    kt = [b.add_virtual_byte_target_unsafe() for _ in range(4 * nk)]
    it = [b.add_virtual_byte_target_unsafe() for _ in range(16)]
    pt_t = [b.add_virtual_byte_target_unsafe() for _ in range(L)]
    sl, xl, ml = b.sbox_lut(), b.byte_xor_lut(), b.gf_2_8_mul_lut()
    ek = b.key_expansion(nk, nr, xl, sl, kt)
    out = b.gctr(nr, xl, ml, sl, ek, it, pt_t)
    data = b.build()
    exp = pkg.native.gctr(key, icb, pt)
    pw = pkg.PartialWitness()
    for ts, vs in ((kt, key), (it, icb), (pt_t, pt), (out, exp)):
        for t, v in zip(ts, vs):
            pw.set_byte_target(t, v)
    return data, [pw]


def gf_2_128_mul(pkg):  # circuit_gcm.rs:570 test_gf_mul
    x, y = bytes([111] * 16), bytes([222] * 16)
    b = pkg.CircuitBuilder()
    xl, shl, brl = b.byte_xor_lut(), b.u8_unit_right_shift_lut(), b.u8_bitref_lut()
    xt = [b.add_virtual_byte_target_unsafe() for _ in range(16)]
    yt = [b.add_virtual_byte_target_unsafe() for _ in range(16)]
    out = b.gf_2_128_mul(xl, shl, brl, xt, yt)
    data = b.build()
    exp = pkg.native.gf_2_128_mul(x, y)
    pw = pkg.PartialWitness()
    for ts, vs in ((xt, x), (yt, y), (out, exp)):
        for t, v in zip(ts, vs):
            pw.set_byte_target(t, v)
    return data, [pw]


def ghash(pkg, L):  # circuit_gcm.rs:635 test_ghash
    h, x = bytes([222] * 16), bytes([42] * L)
    b = pkg.CircuitBuilder()
    xl, shl, brl = b.byte_xor_lut(), b.u8_unit_right_shift_lut(), b.u8_bitref_lut()
    ht = [b.add_virtual_byte_target_unsafe() for _ in range(16)]
    xt = [b.add_virtual_byte_target_unsafe() for _ in range(L)]
    out = b.ghash(xl, shl, brl, ht, xt)
    data = b.build()
    exp = pkg.native.ghash(h, x)
    pw = pkg.PartialWitness()
    for ts, vs in ((ht, h), (xt, x), (out, exp)):
        for t, v in zip(ts, vs):
            pw.set_byte_target(t, v)
    return data, [pw]


def encrypt(pkg, nk, L, tag, keys=None):  # circuit_gcm.rs:695 test_encrypt (key [42;..], nonce [111;12], pt [42;L])
    b = pkg.CircuitBuilder()
    t = pkg.AesGcmTarget.build(b, nk, nk + 6, L, tag)
    data = b.build()
    pws = []
    for key, nonce, pt in (keys or [(bytes([42] * (4 * nk)), bytes([111] * 12), bytes([42] * L))]):
        ct, tg = pkg.native.gcm_encrypt(key, nonce, pt)
        pw = pkg.PartialWitness()
        t.set_targets(pw, key, nonce, pt, ct, tg)
        pws.append(pw)
    return data, pws, t


def random_states(seed, count):
    r = random.Random(seed)
    return [[r.randrange(256) for _ in range(16)] for _ in range(count)]


def arithmetic_only(pkg, inputs):
    """No lookup tables at all (the shape of the feistel / ecgfp5 arithmetic): mul_const_add, add, mul, sub, select,
    is_equal and a connect between two computed values.  out = (x*y + 7*z == w) ? x + y : x*z"""
    P = 0xFFFFFFFF00000001
    b = pkg.CircuitBuilder()
    x, y, z, w = (b.add_virtual_target() for _ in range(4))
    xy = b.mul(x, y)
    t = b.mul_const_add(7, z, xy)
    eq = b.is_equal(t, w)
    out = b.select(eq, b.add(x, y), b.mul(x, z))
    t2 = b.add(b.mul_const_add(7, z, b.mul(y, x)), b.zero())   # y*x is a different op than x*y; tie the two results together
    b.connect(t, t2)
    data = b.build()
    pws = []
    for (xv, yv, zv, wv) in inputs:
        tv = (xv * yv + 7 * zv) % P
        pw = pkg.PartialWitness()
        for tt, vv in ((x, xv), (y, yv), (z, zv), (w, wv), (out, (xv + yv) % P if tv == wv else xv * zv % P)):
            pw.set_target(tt, vv % P)
        pws.append(pw)
    return data, pws


def poseidon_encrypt(pkg, L, seeds):
    """poseidon-cipher/src/circuit.rs:156-190 test body: random key point coordinates (x, u), message of L Fq elements,
    two nonce elements; expected ciphertext from the native cipher.  (The cipher never uses the group structure of the key
    point, so two arbitrary Fq coordinates exercise the same circuit; tests/test_ecgfp5.py runs it with a real key.)"""
    P = 0xFFFFFFFF00000001
    b = pkg.CircuitBuilder()
    t = pkg.PoseidonEncryptTarget.build(b, L)
    data = b.build()
    pws, cases = [], []
    for seed in seeds:
        r = random.Random(seed)
        fq = lambda: tuple(r.randrange(P) for _ in range(5))  # noqa: E731
        ks, nonce, msg = [fq(), fq()], [r.randrange(P), r.randrange(P)], [fq() for _ in range(L)]
        ct = pkg.poseidon_native.encrypt(ks, msg, nonce)
        assert pkg.poseidon_native.decrypt(ks, ct, nonce, L) == msg
        pw = pkg.PartialWitness()
        t.set_targets(pw, ks, msg, nonce, ct)
        pws.append(pw)
        cases.append((ks, msg, nonce, ct))
    return data, pws, t, cases


def feistel_native(pkg, state, key_schedule, inverse=False):
    """feistel/src/lib.rs:15-75 with f = Poseidon hash_n_to_hash_no_pad (additive Feistel network over the field)."""
    P = 0xFFFFFFFF00000001
    h = len(state) // 2
    f = lambda v: pkg.poseidon_native.hash_n_to_m_no_pad(v, 4)  # noqa: E731
    st = list(state)
    for k in key_schedule:
        l, r = st[:h], st[h:]
        if not inverse:
            off = f(r + list(k))
            st = r + [(l[i] + off[i]) % P for i in range(h)]
        else:
            off = f(l + list(k))
            st = [(r[i] - off[i]) % P for i in range(h)] + l
    return st


def feistel_poseidon(pkg, seeds, rounds=32):
    """feistel/src/circuit.rs:115 feistel_poseidon_check: STATE_HALF_LEN = 4, KEY_LEN = 4, NR = 32 rounds; built from
    the generic builder API only (hash_n_to_hash_no_pad = first 4 sponge outputs, then `add`)."""
    P = 0xFFFFFFFF00000001
    b = pkg.CircuitBuilder()
    state_t = b.add_virtual_target_arr(8)
    keys_t = [b.add_virtual_target_arr(4) for _ in range(rounds)]
    st = list(state_t)
    for k in keys_t:
        l, r = st[:4], st[4:]
        off = b.hash_n_to_m_no_pad(r + k, 4)
        st = r + [b.add(l[i], off[i]) for i in range(4)]
    data = b.build()
    pws = []
    for seed in seeds:
        rr = random.Random(seed)
        state = [rr.randrange(P) for _ in range(8)]
        ks = [[rr.randrange(P) for _ in range(4)] for _ in range(rounds)]
        out = feistel_native(pkg, state, ks)
        assert feistel_native(pkg, out, ks[::-1], inverse=True) == state      # lib.rs:98 round trip
        pw = pkg.PartialWitness()
        pw.set_target_arr(state_t, state)
        for kt, kv in zip(keys_t, ks):
            pw.set_target_arr(kt, kv)
        pw.set_target_arr(st, out)
        pws.append(pw)
    return data, pws


def zk_gf_2_8_add(pkg, pairs):
    """test_gf_2_8_add's circuit under standard_recursion_zk_config (the config of aes-gcm/examples/aes_gcm_128.rs:36)."""
    b = pkg.CircuitBuilder(zero_knowledge=True)
    lut = b.byte_xor_lut()
    x, y = b.add_virtual_byte_target_unsafe(), b.add_virtual_byte_target_unsafe()
    xy = b.gf_2_8_add(lut, x, y)
    data = b.build()
    pws = []
    for a, c in pairs:
        pw = pkg.PartialWitness()
        pw.set_byte_target(x, a)
        pw.set_byte_target(y, c)
        pw.set_byte_target(xy, a ^ c)
        pws.append(pw)
    return data, pws


def zk_example_aes_gcm_128(pkg):
    """aes-gcm/examples/aes_gcm_128.rs: AesGcm128Target<42> in the zk config, key [123;16], plaintext [231;42]."""
    b = pkg.CircuitBuilder(zero_knowledge=True)
    t = pkg.AesGcmTarget.build(b, 4, 10, 42, False)
    data = b.build()
    key, nonce, pt = bytes([123] * 16), bytes(12), bytes([231] * 42)
    ct, tag = pkg.native.gcm_encrypt(key, nonce, pt)
    pw = pkg.PartialWitness()
    t.set_targets(pw, key, nonce, pt, ct, tag)
    return data, [pw]


def random_circuit(pkg, orc, seed, n_ops=60, n_witnesses=2):
    """A random circuit over the whole builder vocabulary (add / sub / mul / mul_const_add / select / is_equal / S-box
    lookups / connect / in-circuit Poseidon sponge), evaluated in Python to produce the witnesses; a random subset of the
    computed targets is asserted in the PartialWitness, the reference's way of checking a circuit."""
    P = 0xFFFFFFFF00000001
    S = _sbox(orc)
    r = random.Random(seed)
    b = pkg.CircuitBuilder()
    lut = b.sbox_lut() if r.random() < 0.8 else None
    n_in = r.randrange(2, 6)
    nodes = []  # (target, [value per witness], is_byte)
    inputs = []
    for _ in range(n_in):
        byte = lut is not None and r.random() < 0.5
        t = b.add_virtual_byte_target(lut) if byte else b.add_virtual_target()
        vals = [r.randrange(256) if byte else r.choice([0, 1, P - 1, r.randrange(P)]) for _ in range(n_witnesses)]
        nodes.append((t, vals, byte))
        inputs.append((t, vals))
    for c in (0, 1, 7, P - 2):
        nodes.append((b.constant(c), [c] * n_witnesses, c < 256))
    computed = []
    for _ in range(n_ops):
        kind = r.choice(["add", "sub", "mul", "mca", "select", "sbox", "sbox", "connect", "hash"])
        x, y, z = (r.choice(nodes) for _ in range(3))
        if kind == "add":
            t, v, by = b.add(x[0], y[0]), [(a + c) % P for a, c in zip(x[1], y[1])], False
        elif kind == "sub":
            t, v, by = b.sub(x[0], y[0]), [(a - c) % P for a, c in zip(x[1], y[1])], False
        elif kind == "mul":
            t, v, by = b.mul(x[0], y[0]), [a * c % P for a, c in zip(x[1], y[1])], False
        elif kind == "mca":
            k = r.choice([2, 256, P - 1, r.randrange(P)])
            t, v, by = b.mul_const_add(k, x[0], y[0]), [(k * a + c) % P for a, c in zip(x[1], y[1])], False
        elif kind == "select":
            e = b.is_equal(x[0], y[0])
            ev = [int(a == c) for a, c in zip(x[1], y[1])]
            nodes.append((e, ev, True))
            t, v, by = b.select(e, z[0], x[0]), [zz if q else a for q, zz, a in zip(ev, z[1], x[1])], False
        elif kind == "sbox":
            bytes_ = [nd for nd in nodes if nd[2]]
            if lut is None or not bytes_:
                continue
            x = r.choice(bytes_)
            t, v, by = b.add_lookup_from_index(x[0], lut), [S[a] for a in x[1]], True
        elif kind == "connect":
            t1, t2 = b.add(x[0], y[0]), b.add(y[0], x[0])
            if t1 != t2:
                b.connect(t1, t2)
            t, v, by = t1, [(a + c) % P for a, c in zip(x[1], y[1])], False
        else:
            k = r.randrange(1, 11)
            ins = [r.choice(nodes) for _ in range(k)]
            m = r.randrange(1, 10)
            outs = b.hash_n_to_m_no_pad([nd[0] for nd in ins], m)
            per_w = [pkg.poseidon_native.hash_n_to_m_no_pad([nd[1][w] for nd in ins], m) for w in range(n_witnesses)]
            for j, ot in enumerate(outs):
                nodes.append((ot, [per_w[w][j] for w in range(n_witnesses)], False))
                computed.append(nodes[-1])
            continue
        nodes.append((t, v, by))
        computed.append(nodes[-1])
    data = b.build()
    pws = []
    asserted = [nd for nd in computed if r.random() < 0.3]
    for w in range(n_witnesses):
        pw = pkg.PartialWitness()
        for t, vals in inputs:
            pw.set_target(t, vals[w])
        for t, vals, _ in asserted:
            pw.set_target(t, vals[w])
        pws.append(pw)
    return data, pws


# ---- ecgfp5 crate: the three circuit tests (inputs seeded where upstream draws from OsRng) -------------------------
def ecgfp5_public_key(pkg, seeds):
    """ecgfp5/src/circuit.rs:83-100 public_key_calculation."""
    b = pkg.CircuitBuilder()
    sk_t = b.add_secret_key()
    pk_t = b.public_key(sk_t)
    data = b.build()
    pws, cases = [], []
    for seed in seeds:
        sk = pkg.ECGFP5SecretKey.rand(seed)
        pk = sk.public_key()
        pw = pkg.PartialWitness()
        pw.set_secret_key_target(sk_t, sk)
        pw.set_point_target(pk_t, pk)
        pws.append(pw)
        cases.append((sk, pk))
    return data, pws, (sk_t, pk_t), cases


def ecgfp5_elgamal(pkg, seeds):
    """ecgfp5/src/elgamal/circuit.rs:65-97 elgamal_encryption."""
    E = pkg.ecgfp5
    b = pkg.CircuitBuilder()
    pk_t = b.add_virtual_point_target()
    nonce_t = b.add_virtual_biguint320_target()
    msg_t = b.add_virtual_point_target()
    ct_t = b.elgamal_encrypt(pk_t, nonce_t, msg_t)
    data = b.build()
    pws, cases = [], []
    for seed in seeds:
        sk = pkg.ECGFP5SecretKey.rand(3 * seed)
        pk = sk.public_key()
        msg = E.new_rand_from_subgroup(3 * seed + 1)
        nonce = E.random_scalar(3 * seed + 2)
        ct = E.elgamal_encrypt(pk, nonce, msg)
        pw = pkg.PartialWitness()
        pw.set_point_target(pk_t, pk)
        pw.set_point_target(msg_t, msg)
        pw.set_biguint320_target(nonce_t, nonce)
        pw.set_point_target(ct_t[0], ct[0])
        pw.set_point_target(ct_t[1], ct[1])
        pws.append(pw)
        cases.append((sk, pk, msg, nonce, ct))
    return data, pws, (pk_t, nonce_t, msg_t, ct_t), cases


def ecgfp5_hashed_elgamal(pkg, seeds):
    """ecgfp5/src/hashed_elgamal/circuit.rs:76-107 hashed_elgamal_encryption."""
    P = 0xFFFFFFFF00000001
    E = pkg.ecgfp5
    b = pkg.CircuitBuilder()
    pk_t = b.add_virtual_point_target()
    nonce_t = b.add_virtual_biguint320_target()
    msg_t = b.add_virtual_target_arr(5)
    ct_t = b.hashed_elgamal_encrypt(pk_t, nonce_t, msg_t)
    data = b.build()
    pws, cases = [], []
    for seed in seeds:
        r = random.Random(seed)
        sk = pkg.ECGFP5SecretKey.rand(3 * seed)
        pk = sk.public_key()
        msg = tuple(r.randrange(P) for _ in range(5))
        nonce = E.random_scalar(3 * seed + 2)
        ct = E.hashed_elgamal_encrypt(pk, nonce, msg)
        pw = pkg.PartialWitness()
        pw.set_point_target(pk_t, pk)
        pw.set_target_arr(msg_t, msg)
        pw.set_biguint320_target(nonce_t, nonce)
        pw.set_point_target(ct_t[0], ct[0])
        pw.set_target_arr(ct_t[1], ct[1])
        pws.append(pw)
        cases.append((sk, pk, msg, nonce, ct))
    return data, pws, (pk_t, nonce_t, msg_t, ct_t), cases

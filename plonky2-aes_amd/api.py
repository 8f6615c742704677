"""Host-side mirror of the reference's operator interface over the C ABI (include/p2aes.h).

Same names and argument meaning as the Rust call sites so the parity tests read like the reference's tests:

    builder = CircuitBuilder()                                  # CircuitBuilder::<F, D>::new(config)
    t = AesGcmTarget.build(builder, nk=4, nr=10, L=13, tag=False)   # aes-gcm/src/circuit_gcm.rs:49
    data = builder.build()                                      # builder.build::<PoseidonGoldilocksConfig>()
    pw = PartialWitness()                                       # PartialWitness::<F>::new()
    t.set_targets(pw, key, nonce, pt, ct, tag)                  # circuit_gcm.rs:174
    proof = data.prove(pw)                                      # data.prove(pw)?   (GPU; raises ProveError)
    data.verify(proof)                                          # data.verify(proof)

PyTorch is not needed here; `prove_batch_device` accepts raw device pointers (e.g. torch tensors' data_ptr()).
"""
import ctypes as C
import os

P = 0xFFFFFFFF00000001


class P2Error(RuntimeError):
    pass


class ProveError(P2Error):
    """`data.prove(pw)` returned Err (witness conflict, lookup miss, missing input)."""

    def __init__(self, status):
        self.status = status
        super().__init__({1: "witness conflict or lookup input not in table", 2: "a generator never ran (missing input)",
                          3: "opening point is in the subgroup",
                          4: "no proof-of-work witness found"}.get(status, "prove failed (%d)" % status))


u32p = C.POINTER(C.c_uint32)


def lib_path():
    return os.environ.get("P2AES_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libp2aes.so")


_lib = None
u64, u8p, u64p, u16p, sz = C.c_uint64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint16), C.c_size_t


class _Info(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("degree_bits", "num_wires", "num_routed_wires", "num_constants_cols", "num_zs_cols",
                                           "num_quotient_cols", "num_luts", "num_ops", "num_levels", "num_slots",
                                           "num_virtual_targets", "num_fri_rounds")] + [("proof_bytes", C.c_uint64), ("zero_knowledge", C.c_uint32),
                                                                                      ("num_gate_kinds", C.c_uint32)]


class _Assignment(C.Structure):
    _fields_ = [("targets", u64p), ("values", u64p), ("count", sz)]


class _KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("count", C.c_uint32)]


def lib():
    """Load the shared library; fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise P2Error("native library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    sig = {
        "p2_last_error": (C.c_char_p, []),
        "p2_builder_new": (vp, []), "p2_builder_new_zk": (vp, []), "p2_builder_free": (None, [vp]),
        "p2_circuit_set_zk_seed": (C.c_int, [vp, u64]),
        "p2_circuit_set_zk_key": (C.c_int, [vp, C.POINTER(u64)]),
        "p2_circuit_set_option": (C.c_int, [vp, C.c_char_p, C.c_long]),
        "p2_prove_batch_multi": (C.c_int, [C.POINTER(vp), sz, sz, C.POINTER(_Assignment), C.c_char_p, C.POINTER(C.c_int)]),
        "p2_builder_add_virtual_target": (u64, [vp]), "p2_builder_constant": (u64, [vp, u64]),
        "p2_builder_zero": (u64, [vp]), "p2_builder_one": (u64, [vp]),
        "p2_builder_arithmetic": (u64, [vp, u64, u64, u64, u64, u64]),
        "p2_builder_mul_const_add": (u64, [vp, u64, u64, u64]),
        "p2_builder_add": (u64, [vp, u64, u64]), "p2_builder_sub": (u64, [vp, u64, u64]), "p2_builder_mul": (u64, [vp, u64, u64]),
        "p2_builder_select": (u64, [vp, u64, u64, u64]), "p2_builder_is_equal": (u64, [vp, u64, u64]),
        "p2_builder_connect": (None, [vp, u64, u64]),
        "p2_builder_add_lookup_table_from_pairs": (sz, [vp, u16p, sz]),
        "p2_builder_add_lookup_from_index": (u64, [vp, u64, sz]),
        "p2_builder_num_gates": (sz, [vp]),
        "p2_builder_build": (C.c_int, [vp, C.POINTER(u8p), C.POINTER(sz)]), "p2_blob_free": (None, [u8p]),
        "p2_aes_sbox_lut": (sz, [vp]), "p2_aes_byte_xor_lut": (sz, [vp]), "p2_aes_gf_2_8_mul_lut": (sz, [vp]),
        "p2_gcm_u8_unit_right_shift_lut": (sz, [vp]), "p2_gcm_u8_bitref_lut": (sz, [vp]),
        "p2_aes_add_virtual_byte_target": (u64, [vp, sz]), "p2_aes_add_virtual_byte_target_unsafe": (u64, [vp]),
        "p2_aes_state_sub_bytes": (None, [vp, sz, u64p, u64p]),
        "p2_aes_state_mix_columns": (None, [vp, sz, sz, u64p, u64p]),
        "p2_aes_gf_2_8_mul": (u64, [vp, sz, u64, u64]), "p2_aes_gf_2_8_add": (u64, [vp, sz, u64, u64]),
        "p2_aes_key_expansion": (None, [vp, C.c_int, C.c_int, sz, sz, u64p, u64p]),
        "p2_aes_encrypt_block": (None, [vp, C.c_int, sz, sz, sz, u64p, u64p, u64p]),
        "p2_gcm_gctr": (None, [vp, C.c_int, sz, sz, sz, u64p, u64p, u64p, sz, u64p]),
        "p2_gcm_right_shift_one": (None, [vp, sz, u64p, u64p]), "p2_gcm_inc32": (None, [vp, u64p, u64p]),
        "p2_gcm_gf_2_128_mul": (None, [vp, sz, sz, sz, u64p, u64p, u64p]),
        "p2_gcm_ghash": (C.c_int, [vp, sz, sz, sz, u64p, u64p, sz, u64p]),
        "p2_aes_gcm_build": (C.c_int, [vp, C.c_int, C.c_int, sz, C.c_int, u64p, u64p, u64p, u64p, u64p]),
        "p2_builder_hash_n_to_m_no_pad": (C.c_int, [vp, u64p, sz, u64p, sz]),
        "p2_poseidon_cipher_build": (C.c_int, [vp, sz, u64p, u64p, u64p, u64p]),
        "p2_native_hash_n_to_m_no_pad": (None, [u64p, sz, u64p, sz]),
        "p2_native_poseidon_encrypt": (None, [u64p, u64p, sz, u64p, u64p]),
        "p2_native_poseidon_decrypt": (C.c_int, [u64p, u64p, sz, u64p, sz, u64p]),
        "p2_ecgfp5_group_order": (None, [u64p]), "p2_ecgfp5_generator": (None, [u64p]),
        "p2_ecgfp5_mul": (None, [u64p, u64p, u64p]), "p2_ecgfp5_add": (None, [u64p, u64p, u64p]), "p2_ecgfp5_neg": (None, [u64p, u64p]),
        "p2_ecgfp5_is_in_subgroup": (C.c_int, [u64p]),
        "p2_ecgfp5_compress": (None, [u64p, u64p]), "p2_ecgfp5_decompress": (C.c_int, [u64p, u64p]),
        "p2_ecgfp5_random_scalar": (C.c_int, [u64p]), "p2_ecgfp5_random_point": (C.c_int, [u64p]),
        "p2_ecgfp5_encode_binary": (C.c_int, [u32p, u64p]),
        "p2_ecgfp5_random_scalar_seeded": (None, [u64, u64p]), "p2_ecgfp5_random_point_seeded": (None, [u64, u64p]),
        "p2_ecgfp5_encode_binary_seeded": (None, [u32p, u64, u64p]), "p2_ecgfp5_decode_binary": (None, [u64p, u32p]),
        "p2_elgamal_encrypt": (C.c_int, [u64p, u64p, u64p, u64p, u64p]), "p2_elgamal_decrypt": (C.c_int, [u64p, u64p, u64p, u64p]),
        "p2_hashed_elgamal_encrypt": (C.c_int, [u64p, u64p, u64p, u64p, u64p]),
        "p2_hashed_elgamal_decrypt": (C.c_int, [u64p, u64p, u64p, u64p]),
        "p2_builder_add_virtual_point_target": (None, [vp, u64p]), "p2_builder_constant_point": (None, [vp, u64p, u64p]),
        "p2_builder_add_virtual_biguint320_target": (None, [vp, u64p]),
        "p2_builder_multiply_point": (C.c_int, [vp, u64p, u64p, u64p]), "p2_builder_add_point": (None, [vp, u64p, u64p, u64p]),
        "p2_builder_public_key": (C.c_int, [vp, u64p, u64p]),
        "p2_builder_elgamal_encrypt": (C.c_int, [vp, u64p, u64p, u64p, u64p, u64p]),
        "p2_builder_hashed_elgamal_encrypt": (C.c_int, [vp, u64p, u64p, u64p, u64p, u64p]),
        "p2_selftest_host": (C.c_int, [u64, sz, sz]), "p2_selftest_device": (C.c_int, [u64, sz, C.c_int]),
        "p2_native_gf_2_8_mul": (C.c_uint8, [C.c_uint8, C.c_uint8]),
        "p2_native_aes_key_expansion": (None, [C.c_char_p, C.c_int, C.c_int, C.c_char_p]),
        "p2_native_aes_encrypt_block": (None, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]),
        "p2_native_gf_2_128_mul": (None, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "p2_native_ghash": (None, [C.c_char_p, C.c_char_p, sz, C.c_char_p]),
        "p2_native_gctr": (None, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, sz, C.c_char_p]),
        "p2_native_aes_gcm_encrypt": (None, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, sz, C.c_char_p, C.c_char_p]),
        "p2_blob_info": (C.c_int, [C.c_char_p, sz, C.POINTER(_Info)]),
        "p2_witness_schedule_check": (C.c_int, [C.c_char_p, sz, C.c_uint32, u32p]),
        "p2_verify": (C.c_int, [C.c_char_p, sz, u64p, sz, C.c_char_p, sz]),
        "p2_circuit_load": (vp, [C.c_char_p, sz, C.c_int]), "p2_circuit_free": (None, [vp]),
        "p2_circuit_verifier_data": (C.c_int, [vp, u64p, sz, C.POINTER(sz)]),
        "p2_circuit_proof_bytes": (sz, [vp]),
        "p2_circuit_chunk_proofs": (sz, [vp]),
        "p2_prove_batch": (C.c_int, [vp, sz, C.POINTER(_Assignment), C.c_char_p, C.POINTER(C.c_int)]),
        "p2_prove_batch_device": (C.c_int, [vp, sz, u64p, sz, vp, vp, vp, vp]),
        "p2_circuit_synchronize": (C.c_int, [vp]),
        "p2_circuit_set_timing": (C.c_int, [vp, C.c_int]),
        "p2_circuit_get_timing": (sz, [vp, C.POINTER(_KernelTime), sz]),
        "p2_gpu_device_count": (C.c_int, []),
        "p2_gpu_poseidon": (C.c_int, [u64p, sz, C.c_int]),
        "p2_gpu_lde": (C.c_int, [u64p, sz, C.c_int, C.c_int, u64p, C.c_int]),
        "p2_gpu_intt": (C.c_int, [u64p, sz, C.c_int, u64p, C.c_int]),
        "p2_gpu_merkle_cap": (C.c_int, [u64p, sz, sz, C.c_int, u64p, C.c_int]),
        "p2_circuit_debug_read": (C.c_int, [vp, C.c_char_p, sz, u64p, sz, C.POINTER(sz)]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            f = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        f.restype, f.argtypes = res, args
    L._p2_missing = missing
    L._p2_signatures = sig
    _lib = L
    return L


def _err():
    return lib().p2_last_error().decode()


def _arr(vals):
    return (u64 * len(vals))(*vals)


class CircuitBuilder:
    def __init__(self, zero_knowledge=False):
        """CircuitBuilder::<F, D>::new(standard_recursion_config()) or, with zero_knowledge, standard_recursion_zk_config()."""
        self._h = lib().p2_builder_new_zk() if zero_knowledge else lib().p2_builder_new()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().p2_builder_free(self._h)
            self._h = None

    def add_virtual_target(self): return lib().p2_builder_add_virtual_target(self._h)
    def add_virtual_target_arr(self, n): return [self.add_virtual_target() for _ in range(n)]
    def constant(self, c): return lib().p2_builder_constant(self._h, c % P)
    def zero(self): return lib().p2_builder_zero(self._h)
    def one(self): return lib().p2_builder_one(self._h)
    def arithmetic(self, c0, c1, m0, m1, addend): return lib().p2_builder_arithmetic(self._h, c0 % P, c1 % P, m0, m1, addend)
    def mul_const_add(self, c, x, y): return lib().p2_builder_mul_const_add(self._h, c % P, x, y)
    def add(self, x, y): return lib().p2_builder_add(self._h, x, y)
    def sub(self, x, y): return lib().p2_builder_sub(self._h, x, y)
    def mul(self, x, y): return lib().p2_builder_mul(self._h, x, y)
    def select(self, b, x, y): return lib().p2_builder_select(self._h, b, x, y)
    def is_equal(self, x, y): return lib().p2_builder_is_equal(self._h, x, y)
    def connect(self, x, y): lib().p2_builder_connect(self._h, x, y)

    def add_lookup_table_from_pairs(self, pairs):
        flat = (C.c_uint16 * (2 * len(pairs)))(*[v for p in pairs for v in p])
        return lib().p2_builder_add_lookup_table_from_pairs(self._h, flat, len(pairs))

    def add_lookup_from_index(self, looking_in, lut_index):
        t = lib().p2_builder_add_lookup_from_index(self._h, looking_in, lut_index)
        if t == 0xFFFFFFFFFFFFFFFF:
            raise P2Error(_err())
        return t

    def num_gates(self): return lib().p2_builder_num_gates(self._h)

    # ---- CircuitBuilderAESState (aes-gcm/src/circuit_aes.rs:41-174) and the GCM helpers (circuit_gcm.rs)
    def sbox_lut(self): return lib().p2_aes_sbox_lut(self._h)
    def byte_xor_lut(self): return lib().p2_aes_byte_xor_lut(self._h)
    def gf_2_8_mul_lut(self): return lib().p2_aes_gf_2_8_mul_lut(self._h)
    def u8_unit_right_shift_lut(self): return lib().p2_gcm_u8_unit_right_shift_lut(self._h)
    def u8_bitref_lut(self): return lib().p2_gcm_u8_bitref_lut(self._h)
    def add_virtual_byte_target(self, u8_table_idx): return lib().p2_aes_add_virtual_byte_target(self._h, u8_table_idx)
    def add_virtual_byte_target_unsafe(self): return lib().p2_aes_add_virtual_byte_target_unsafe(self._h)
    def add_virtual_state_target_unsafe(self): return [self.add_virtual_byte_target_unsafe() for _ in range(16)]
    def add_virtual_state_target(self, lut): return [self.add_virtual_byte_target(lut) for _ in range(16)]

    def state_sub_bytes(self, sbox_lut, s):
        out = (u64 * 16)()
        lib().p2_aes_state_sub_bytes(self._h, sbox_lut, _arr(s), out)
        return list(out)

    def state_mix_columns(self, xor_lut, mul_lut, s):
        out = (u64 * 16)()
        lib().p2_aes_state_mix_columns(self._h, xor_lut, mul_lut, _arr(s), out)
        return list(out)

    def gf_2_8_mul(self, mul_lut, x, y): return lib().p2_aes_gf_2_8_mul(self._h, mul_lut, x, y)
    def gf_2_8_add(self, xor_lut, x, y): return lib().p2_aes_gf_2_8_add(self._h, xor_lut, x, y)

    def key_expansion(self, nk, nr, xor_lut, sbox_lut, key):
        out = (u64 * (16 * (nr + 1)))()
        lib().p2_aes_key_expansion(self._h, nk, nr, xor_lut, sbox_lut, _arr(key), out)
        return list(out)

    def encrypt_block(self, nr, xor_lut, mul_lut, sbox_lut, state, expanded_key):
        out = (u64 * 16)()
        lib().p2_aes_encrypt_block(self._h, nr, xor_lut, mul_lut, sbox_lut, _arr(state), _arr(expanded_key), out)
        return list(out)

    def gctr(self, nr, xor_lut, mul_lut, sbox_lut, expanded_key, icb, x):
        out = (u64 * len(x))()
        lib().p2_gcm_gctr(self._h, nr, xor_lut, mul_lut, sbox_lut, _arr(expanded_key), _arr(icb), _arr(x), len(x), out)
        return list(out)

    def right_shift_one(self, shift_lut, v):
        out = (u64 * 16)()
        lib().p2_gcm_right_shift_one(self._h, shift_lut, _arr(v), out)
        return list(out)

    def inc32(self, block):
        out = (u64 * 16)()
        lib().p2_gcm_inc32(self._h, _arr(block), out)
        return list(out)

    def gf_2_128_mul(self, xor_lut, shift_lut, bitref_lut, x, y):
        out = (u64 * 16)()
        lib().p2_gcm_gf_2_128_mul(self._h, xor_lut, shift_lut, bitref_lut, _arr(x), _arr(y), out)
        return list(out)

    def ghash(self, xor_lut, shift_lut, bitref_lut, h, x):
        out = (u64 * 16)()
        if lib().p2_gcm_ghash(self._h, xor_lut, shift_lut, bitref_lut, _arr(h), _arr(x), len(x), out):
            raise P2Error(_err())
        return list(out)

    def hash_n_to_m_no_pad(self, inputs, num_outputs):
        out = (u64 * num_outputs)()
        if lib().p2_builder_hash_n_to_m_no_pad(self._h, _arr(inputs), len(inputs), out, num_outputs):
            raise P2Error(_err())
        return list(out)

    # ---- pod2 CircuitBuilderElliptic / CircuitBuilderBits and the ecgfp5 crate's builder traits.  A PointTarget is its ten
    # targets (x then u), a BigUInt320Target its 320 little-endian bit targets.
    def add_virtual_point_target(self):
        out = (u64 * 10)()
        lib().p2_builder_add_virtual_point_target(self._h, out)
        return list(out)

    def constant_point(self, point):
        out = (u64 * 10)()
        lib().p2_builder_constant_point(self._h, _pt(point), out)
        return list(out)

    def add_virtual_biguint320_target(self):
        out = (u64 * 320)()
        lib().p2_builder_add_virtual_biguint320_target(self._h, out)
        return list(out)

    def multiply_point(self, bits, point_target):
        out = (u64 * 10)()
        if lib().p2_builder_multiply_point(self._h, _arr(bits), _arr(point_target), out):
            raise P2Error(_err())
        return list(out)

    def add_point(self, p, q):
        out = (u64 * 10)()
        lib().p2_builder_add_point(self._h, _arr(p), _arr(q), out)
        return list(out)

    def add_secret_key(self): return self.add_virtual_biguint320_target()   # ecgfp5/src/circuit.rs:31

    def public_key(self, sk_target):                                         # ecgfp5/src/circuit.rs:35
        out = (u64 * 10)()
        if lib().p2_builder_public_key(self._h, _arr(sk_target), out):
            raise P2Error(_err())
        return list(out)

    def elgamal_encrypt(self, pk, nonce, msg):                               # elgamal/circuit.rs:28
        c0, c1 = (u64 * 10)(), (u64 * 10)()
        if lib().p2_builder_elgamal_encrypt(self._h, _arr(pk), _arr(nonce), _arr(msg), c0, c1):
            raise P2Error(_err())
        return list(c0), list(c1)

    def hashed_elgamal_encrypt(self, pk, nonce, msg):                        # hashed_elgamal/circuit.rs:33
        c0, ct = (u64 * 10)(), (u64 * 5)()
        if lib().p2_builder_hashed_elgamal_encrypt(self._h, _arr(pk), _arr(nonce), _arr(msg), c0, ct):
            raise P2Error(_err())
        return list(c0), list(ct)

    def build(self):
        blob, n = u8p(), sz()
        if lib().p2_builder_build(self._h, C.byref(blob), C.byref(n)):
            raise P2Error(_err())
        # (ctypes.string_at takes a C int: blobs of 2^21-row circuits are larger than 2 GiB)
        data = bytes((C.c_char * n.value).from_address(C.cast(blob, C.c_void_p).value))
        lib().p2_blob_free(blob)
        return CircuitData(data)


class PartialWitness:
    """plonky2 iop::witness::PartialWitness: a target -> value map; set_target errors on a conflicting re-set."""

    def __init__(self):
        self.map = {}

    def set_target(self, target, value):
        value = int(value)
        if not 0 <= value < P:
            raise P2Error("value is not a canonical field element")
        old = self.map.get(target)
        if old is not None and old != value:
            raise P2Error("target was set twice with different values")
        self.map[target] = value

    def set_target_arr(self, targets, values):
        for t, v in zip(targets, values):
            self.set_target(t, v)

    # PartialWitnessByteArray / PartialWitnessAESState (circuit_aes.rs:277-297)
    def set_point_target(self, target, point):            # pod2 WitnessWriteCurve (elgamal/circuit.rs:88-92)
        self.set_target_arr(target, list(point[0]) + list(point[1]))

    def set_biguint320_target(self, target, value):       # pod2 bits (elgamal/circuit.rs:90)
        assert 0 <= value < 1 << 320
        self.set_target_arr(target, [(value >> i) & 1 for i in range(320)])

    def set_secret_key_target(self, target, sk):          # ecgfp5/src/circuit.rs:52
        self.set_biguint320_target(target, sk.value)

    def set_byte_target(self, target, value): self.set_target(target, value & 0xFF if isinstance(value, int) else int(value))
    def set_state_target(self, targets, state16):
        for t, v in zip(targets, state16):
            self.set_byte_target(t, v)


class CircuitData:
    """builder.build::<PoseidonGoldilocksConfig>() result.  prove() runs on the GPU (no CPU fallback)."""

    def __init__(self, blob, device=0):
        self.blob = blob
        self.device = device
        self._gpu = None
        self._vd = None
        info = _Info()
        if lib().p2_blob_info(blob, len(blob), C.byref(info)):
            raise P2Error(_err())
        self.info = {n: getattr(info, n) for n, _ in _Info._fields_}

    def __del__(self):
        if getattr(self, "_gpu", None):
            lib().p2_circuit_free(self._gpu)
            self._gpu = None

    def gpu(self):
        if self._gpu is None:
            h = lib().p2_circuit_load(self.blob, len(self.blob), self.device)
            if not h:
                raise P2Error("p2_circuit_load failed: " + _err())
            self._gpu = h
        return self._gpu

    @property
    def proof_bytes(self): return self.info["proof_bytes"]

    def witness_schedule(self, fuse=8):
        """Host-side check of the device's witness schedule for macro size `fuse` (csrc/witness_schedule.h):
        {levels, chains, max_chain, fused_ops}; raises if an operand would not be ready when its op runs."""
        out = (C.c_uint32 * 4)()
        if lib().p2_witness_schedule_check(self.blob, len(self.blob), fuse, out):
            raise P2Error(_err())
        return dict(zip(("levels", "chains", "max_chain", "fused_ops"), out))

    def verifier_data(self):
        """constants_sigmas_cap || circuit_digest.  Computed on the device when the circuit is loaded (the reference's
        build() computes it); cached here, so verify() of later proofs needs no device."""
        if self._vd is None:
            out = (u64 * 80)()
            n = sz()
            if lib().p2_circuit_verifier_data(self.gpu(), out, 80, C.byref(n)):
                raise P2Error(_err())
            self._vd = list(out[: n.value])
        return list(self._vd)

    def set_option(self, name, value):
        """Tuning knobs of the GPU handle: "chunk", "streams", "debug_timing" (include/p2aes.h)."""
        if lib().p2_circuit_set_option(self.gpu(), name.encode(), int(value)):
            raise P2Error(_err())

    def prove_batch(self, pws):
        """Returns (proofs: list[bytes|None], status: list[int])."""
        B = len(pws)
        asg = (_Assignment * B)()
        keep = []
        for i, pw in enumerate(pws):
            ts, vs = _arr(list(pw.map.keys())), _arr(list(pw.map.values()))
            keep.append((ts, vs))
            asg[i].targets, asg[i].values, asg[i].count = C.cast(ts, u64p), C.cast(vs, u64p), len(pw.map)
        buf = C.create_string_buffer(B * self.proof_bytes)
        status = (C.c_int * B)()
        if lib().p2_prove_batch(self.gpu(), B, asg, buf, status):
            raise P2Error("p2_prove_batch failed: " + _err())
        pb, base = self.proof_bytes, C.addressof(buf)   # (buf.raw would copy the whole buffer once per proof)
        return [C.string_at(base + i * pb, pb) if status[i] == 0 else None for i in range(B)], list(status)

    @staticmethod
    def prove_batch_multi(datas, pws):
        """One batch sharded over several CircuitData loads of the same blob (one per HIP device): p2_prove_batch_multi."""
        B = len(pws)
        asg = (_Assignment * B)()
        keep = []
        for i, pw in enumerate(pws):
            ts, vs = _arr(list(pw.map.keys())), _arr(list(pw.map.values()))
            keep.append((ts, vs))
            asg[i].targets, asg[i].values, asg[i].count = ts, vs, len(pw.map)
        pb = datas[0].proof_bytes
        buf = C.create_string_buffer(B * pb)
        status = (C.c_int * B)()
        hs = (C.c_void_p * len(datas))(*[d.gpu() for d in datas])
        if lib().p2_prove_batch_multi(hs, len(datas), B, asg, buf, status):
            raise P2Error("p2_prove_batch_multi failed: " + _err())
        base = C.addressof(buf)
        return [C.string_at(base + i * pb, pb) if status[i] == 0 else None for i in range(B)], list(status)

    def prove_batch_device(self, targets, d_values, d_proofs, d_status, batch, stream=None):
        """Device-resident path: `d_values` [batch][len(targets)] u64, `d_proofs` batch*proof_bytes bytes and `d_status`
        int32[batch] are raw device pointers (e.g. torch tensors' data_ptr()); `stream` a raw hipStream_t or None."""
        tarr = _arr(list(targets))
        if lib().p2_prove_batch_device(self.gpu(), batch, tarr, len(targets), d_values, d_proofs, d_status, stream):
            raise P2Error("p2_prove_batch_device failed: " + _err())

    def set_zk_seed(self, seed):
        """TEST ONLY (reproducible zk proofs): blinding key = (seed, 0, 0, 0).  The default key comes from the OS CSPRNG."""
        if lib().p2_circuit_set_zk_seed(self.gpu(), seed):
            raise P2Error(_err())

    def set_zk_key(self, key4):
        """TEST ONLY: fix the full 256-bit blinding key (four field elements) and reset the proof counter."""
        if lib().p2_circuit_set_zk_key(self.gpu(), (C.c_uint64 * 4)(*key4)):
            raise P2Error(_err())

    def synchronize(self):
        if lib().p2_circuit_synchronize(self.gpu()):
            raise P2Error(_err())

    def prove(self, pw):
        proofs, status = self.prove_batch([pw])
        if status[0]:
            raise ProveError(status[0])
        return proofs[0]

    def verify(self, proof, verifier_data=None):
        vd = verifier_data if verifier_data is not None else self.verifier_data()
        if lib().p2_verify(self.blob, len(self.blob), _arr(vd), len(vd), proof, len(proof)):
            raise P2Error("verify failed: " + _err())

    def debug_read(self, name, index=0, cap=1 << 26):
        out = (u64 * cap)()
        n = sz()
        if lib().p2_circuit_debug_read(self.gpu(), name.encode(), index, out, cap, C.byref(n)):
            raise P2Error(_err())
        return list(out[: n.value])

    def debug_read_bytes(self, name, index=0, cap=1 << 26):
        """The same buffer as little-endian u64 bytes (10^8-word buffers of the 2^19-row circuits: hash, do not list)."""
        out = (u64 * cap)()
        n = sz()
        if lib().p2_circuit_debug_read(self.gpu(), name.encode(), index, out, cap, C.byref(n)):
            raise P2Error(_err())
        return bytes(memoryview(out).cast("B")[: 8 * n.value])


class AesGcmTarget:
    """AesGcmTarget<NK, 4, NR, L, TAG> (aes-gcm/src/circuit_gcm.rs:24-209)."""

    @staticmethod
    def build(builder, nk=4, nr=10, L=16, tag=False):
        t = AesGcmTarget()
        t.nk, t.nr, t.L, t.TAG = nk, nr, L, tag
        key, nonce, pt, ct, tg = (u64 * (4 * nk))(), (u64 * 12)(), (u64 * max(L, 1))(), (u64 * max(L, 1))(), (u64 * 16)()
        if lib().p2_aes_gcm_build(builder._h, nk, nr, L, int(tag), key, nonce, pt, ct, tg):
            raise P2Error(_err())
        t.key, t.nonce, t.pt, t.ct, t.tag = list(key), list(nonce), list(pt)[:L], list(ct)[:L], list(tg)
        return t

    def set_targets(self, pw, key, nonce, pt, ct, tag):
        # effective contract of circuit_gcm.rs:183-193: len(pt) == len(ct) == L (copy_from_slice panics otherwise)
        assert len(pt) == self.L and len(ct) == self.L and len(key) == 4 * self.nk and len(nonce) == 12
        for t, v in zip(self.key, key): pw.set_byte_target(t, v)
        for t, v in zip(self.nonce, nonce): pw.set_byte_target(t, v)
        for t, v in zip(self.pt, pt): pw.set_byte_target(t, v)
        for t, v in zip(self.ct, ct): pw.set_byte_target(t, v)
        if self.TAG:
            assert len(tag) == 16
            for t, v in zip(self.tag, tag): pw.set_byte_target(t, v)
        else:
            for t in self.tag: pw.set_byte_target(t, 0)


class PoseidonEncryptTarget:
    """PoseidonEncryptTarget<L> (poseidon-cipher/src/circuit.rs:35-118).  Fq elements are 5-tuples of field elements;
    the key point `ks` is its two coordinates (x, u)."""

    @staticmethod
    def build(builder, L):
        t = PoseidonEncryptTarget()
        t.L = L
        ks, m, nonce, ct = (u64 * 10)(), (u64 * (5 * L))(), (u64 * 2)(), (u64 * (5 * (L + 1)))()
        if lib().p2_poseidon_cipher_build(builder._h, L, ks, m, nonce, ct):
            raise P2Error(_err())
        t.ks, t.m, t.nonce, t.ct = list(ks), list(m), list(nonce), list(ct)
        return t

    def set_targets(self, pw, ks, m, nonce, ct):
        assert len(m) == self.L and len(ct) == self.L + 1 and len(ks) == 2   # circuit.rs:99-100
        pw.set_target_arr(self.ks, [v for fq in ks for v in fq])
        pw.set_target_arr(self.m, [v for fq in m for v in fq])
        pw.set_target_arr(self.nonce, nonce)
        pw.set_target_arr(self.ct, [v for fq in ct for v in fq])


class poseidon_native:
    """poseidon-cipher/src/lib.rs through the C ABI."""

    @staticmethod
    def hash_n_to_m_no_pad(inputs, m):
        out = (u64 * m)()
        lib().p2_native_hash_n_to_m_no_pad(_arr(inputs), len(inputs), out, m)
        return list(out)

    @staticmethod
    def new_key(seed=None):                                # lib.rs:31 (OS randomness, as upstream's OsRng; seed = tests only)
        k = ecgfp5.random_scalar(seed)
        return k, ecgfp5.mul(k, ecgfp5.generator())

    @staticmethod
    def expanded_key(K, seed=None):                        # lib.rs:36
        return ecgfp5.mul(ecgfp5.random_scalar(seed), K)

    @staticmethod
    def encrypt(ks, msg, nonce):
        n = len(msg)
        nct = (n + 2) // 3 * 3 + 1
        ct = (u64 * (5 * nct))()
        lib().p2_native_poseidon_encrypt(_arr([v for fq in ks for v in fq]), _arr([v for fq in msg for v in fq]) if n else (u64 * 1)(), n, _arr(nonce), ct)
        return [tuple(ct[5 * i:5 * i + 5]) for i in range(nct)]

    @staticmethod
    def decrypt(ks, ct, nonce, l):
        msg = (u64 * (5 * max(l, 1)))()
        if lib().p2_native_poseidon_decrypt(_arr([v for fq in ks for v in fq]), _arr([v for fq in ct for v in fq]), len(ct), _arr(nonce), l, msg):
            raise P2Error(_err())
        return [tuple(msg[5 * i:5 * i + 5]) for i in range(l)]


def _pt(point):
    return _arr(list(point[0]) + list(point[1]))


def _pt_out(buf):
    return (tuple(buf[0:5]), tuple(buf[5:10]))


def _sc(k):
    return _arr([(k >> (64 * i)) & (2**64 - 1) for i in range(5)])


class ecgfp5:
    """The ecgfp5 crate's native side (ecgfp5/src/lib.rs, elgamal.rs, hashed_elgamal.rs) through the C ABI.  A Point is
    ((x0..x4), (u0..u4)); scalars are Python ints.  Randomness comes from the OS CSPRNG as upstream's OsRng; an explicit seed (tests only) makes it reproducible."""

    @staticmethod
    def group_order():
        out = (u64 * 5)()
        lib().p2_ecgfp5_group_order(out)
        return sum(int(out[i]) << (64 * i) for i in range(5))

    @staticmethod
    def generator():
        out = (u64 * 10)()
        lib().p2_ecgfp5_generator(out)
        return _pt_out(out)

    @staticmethod
    def mul(k, point):
        out = (u64 * 10)()
        lib().p2_ecgfp5_mul(_sc(k), _pt(point), out)
        return _pt_out(out)

    @staticmethod
    def add(p, q):
        out = (u64 * 10)()
        lib().p2_ecgfp5_add(_pt(p), _pt(q), out)
        return _pt_out(out)

    @staticmethod
    def neg(p):
        out = (u64 * 10)()
        lib().p2_ecgfp5_neg(_pt(p), out)
        return _pt_out(out)

    @staticmethod
    def is_in_subgroup(p): return bool(lib().p2_ecgfp5_is_in_subgroup(_pt(p)))

    @staticmethod
    def compress_from_subgroup(p):
        out = (u64 * 5)()
        lib().p2_ecgfp5_compress(_pt(p), out)
        return tuple(out)

    @staticmethod
    def decompress_into_subgroup(w):
        out = (u64 * 10)()
        if lib().p2_ecgfp5_decompress(_arr(list(w)), out):
            raise P2Error(_err())
        return _pt_out(out)

    @staticmethod
    def random_scalar(seed=None):
        """Uniform scalar below the group order from the OS CSPRNG; a `seed` (tests / benchmarks only) makes it reproducible
        and is NOT a source of key material."""
        out = (u64 * 5)()
        if seed is None:
            if lib().p2_ecgfp5_random_scalar(out):
                raise P2Error(_err())
        else:
            lib().p2_ecgfp5_random_scalar_seeded(seed, out)
        return sum(int(out[i]) << (64 * i) for i in range(5))

    @staticmethod
    def new_rand_from_subgroup(seed=None):
        out = (u64 * 10)()
        if seed is None:
            if lib().p2_ecgfp5_random_point(out):
                raise P2Error(_err())
        else:
            lib().p2_ecgfp5_random_point_seeded(seed, out)
        return _pt_out(out)

    @staticmethod
    def encode_binary(x, seed=None):                       # lib.rs:48
        assert 0 <= x < 1 << 160
        out = (u64 * 10)()
        limbs = (C.c_uint32 * 5)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(5)])
        if seed is None:
            if lib().p2_ecgfp5_encode_binary(limbs, out):
                raise P2Error(_err())
        else:
            lib().p2_ecgfp5_encode_binary_seeded(limbs, seed, out)
        return _pt_out(out)

    @staticmethod
    def decode_binary(p):                                  # lib.rs:80
        out = (C.c_uint32 * 5)()
        lib().p2_ecgfp5_decode_binary(_pt(p), out)
        return sum(int(out[i]) << (32 * i) for i in range(5))

    @staticmethod
    def elgamal_encrypt(pk, nonce, msg):                   # elgamal.rs:11
        c0, c1 = (u64 * 10)(), (u64 * 10)()
        if lib().p2_elgamal_encrypt(_pt(pk), _sc(nonce), _pt(msg), c0, c1):
            raise P2Error(_err())
        return _pt_out(c0), _pt_out(c1)

    @staticmethod
    def elgamal_decrypt(sk, ct):                           # elgamal.rs:19
        out = (u64 * 10)()
        if lib().p2_elgamal_decrypt(_sc(sk.value), _pt(ct[0]), _pt(ct[1]), out):
            raise P2Error(_err())
        return _pt_out(out)

    @staticmethod
    def hashed_elgamal_encrypt(pk, nonce, msg):            # hashed_elgamal.rs:19
        c0, ct = (u64 * 10)(), (u64 * 5)()
        if lib().p2_hashed_elgamal_encrypt(_pt(pk), _sc(nonce), _arr(list(msg)), c0, ct):
            raise P2Error(_err())
        return _pt_out(c0), tuple(ct)

    @staticmethod
    def hashed_elgamal_decrypt(sk, ct):                    # hashed_elgamal.rs:28
        out = (u64 * 5)()
        if lib().p2_hashed_elgamal_decrypt(_sc(sk.value), _pt(ct[0]), _arr(list(ct[1])), out):
            raise P2Error(_err())
        return tuple(out)


class ECGFP5SecretKey:
    """ecgfp5/src/lib.rs:23-42."""

    def __init__(self, s):
        assert 0 <= s < ecgfp5.group_order()
        self.value = s

    @staticmethod
    def rand(seed=None): return ECGFP5SecretKey(ecgfp5.random_scalar(seed))   # lib.rs:35 OsRng; seed = tests only

    def public_key(self): return ecgfp5.mul(self.value, ecgfp5.generator())


class native:
    """aes-gcm/src/native_aes.rs / native_gcm.rs through the C ABI."""

    @staticmethod
    def gf_2_8_mul(a, b): return lib().p2_native_gf_2_8_mul(a, b)

    @staticmethod
    def key_expansion(key):
        nk = len(key) // 4
        out = C.create_string_buffer(16 * (nk + 7))
        lib().p2_native_aes_key_expansion(bytes(key), nk, nk + 6, out)
        return out.raw

    @staticmethod
    def encrypt_block(key, block):
        nk = len(key) // 4
        out = C.create_string_buffer(16)
        lib().p2_native_aes_encrypt_block(bytes(key), nk, nk + 6, bytes(block), out)
        return out.raw

    @staticmethod
    def gf_2_128_mul(x, y):
        out = C.create_string_buffer(16)
        lib().p2_native_gf_2_128_mul(bytes(x), bytes(y), out)
        return out.raw

    @staticmethod
    def ghash(h, x):
        out = C.create_string_buffer(16)
        lib().p2_native_ghash(bytes(h), bytes(x), len(x), out)
        return out.raw

    @staticmethod
    def gctr(key, icb, x):
        nk = len(key) // 4
        out = C.create_string_buffer(max(len(x), 1))
        lib().p2_native_gctr(bytes(key), nk, nk + 6, bytes(icb), bytes(x), len(x), out)
        return out.raw[: len(x)]

    @staticmethod
    def gcm_encrypt(key, nonce, pt):
        nk = len(key) // 4
        ct, tag = C.create_string_buffer(max(len(pt), 1)), C.create_string_buffer(16)
        lib().p2_native_aes_gcm_encrypt(bytes(key), nk, nk + 6, bytes(nonce), bytes(pt), len(pt), ct, tag)
        return ct.raw[: len(pt)], tag.raw

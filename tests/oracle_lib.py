"""ctypes loader for the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None
u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        vp, sz = C.c_void_p, C.c_size_t
        L.orc_circuit_load.restype, L.orc_circuit_load.argtypes = vp, [C.c_char_p, sz]
        L.orc_circuit_free.argtypes = [vp]
        L.orc_degree_bits.restype, L.orc_degree_bits.argtypes = C.c_uint32, [vp]
        L.orc_set_zk.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.orc_set_zk_key.argtypes = [vp, u64p, C.c_uint64]
        L.orc_set_fault.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_row_gate_kind.restype, L.orc_row_gate_kind.argtypes = C.c_uint32, [vp, C.c_uint32]
        L.orc_wire_slot.restype, L.orc_wire_slot.argtypes = C.c_int32, [vp, C.c_uint32, C.c_uint32]
        L.orc_num_ops.restype, L.orc_num_ops.argtypes = sz, [vp]
        L.orc_get_op.argtypes = [vp, sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_verifier_data.restype, L.orc_verifier_data.argtypes = sz, [vp, u64p, sz]
        L.orc_prove.restype, L.orc_prove.argtypes = C.c_int, [vp, u64p, u64p, sz, C.c_char_p, sz, C.POINTER(sz), C.c_int]
        L.orc_trace_len.restype, L.orc_trace_len.argtypes = sz, [vp, C.c_char_p]
        L.orc_trace_get.restype, L.orc_trace_get.argtypes = sz, [vp, C.c_char_p, u64p, sz]
        L.orc_generate_witness.restype, L.orc_generate_witness.argtypes = C.c_int, [vp, u64p, u64p, sz, u64p]
        for f in ("orc_fmul", "orc_fadd", "orc_fsub"):
            getattr(L, f).restype, getattr(L, f).argtypes = C.c_uint64, [C.c_uint64, C.c_uint64]
        L.orc_finv.restype, L.orc_finv.argtypes = C.c_uint64, [C.c_uint64]
        L.orc_poseidon.argtypes = [u64p]
        L.orc_hash_no_pad.argtypes = [u64p, sz, u64p]
        L.orc_two_to_one.argtypes = [u64p, u64p, u64p]
        L.orc_fft.argtypes = [u64p, C.c_int, C.c_int]
        L.orc_fft_bitrev_out.argtypes = [u64p, C.c_int]
        L.orc_lde.argtypes = [u64p, C.c_int, C.c_int, u64p]
        L.orc_merkle_cap.restype, L.orc_merkle_cap.argtypes = sz, [u64p, sz, sz, C.c_int, u64p]
        L.orc_gf_2_8_mul.restype, L.orc_gf_2_8_mul.argtypes = C.c_uint8, [C.c_uint8, C.c_uint8]
        L.orc_sbox.restype, L.orc_sbox.argtypes = C.c_uint8, [C.c_uint8]
        L.orc_aes_expand_key.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.orc_aes_encrypt_block.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
        L.orc_gf_2_128_mul.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.orc_ghash.argtypes = [C.c_char_p, C.c_char_p, sz, C.c_char_p]
        L.orc_gctr.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, sz, C.c_char_p]
        L.orc_gcm_encrypt.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, sz, C.c_char_p, C.c_char_p]
        L.orc_last_stage_seconds.argtypes = [C.POINTER(C.c_double)]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_fast_hash.argtypes = [C.c_int]
        L.orc_poseidon_sparse.argtypes = [u64p]
        L.orc_set_simd_lanes.restype, L.orc_set_simd_lanes.argtypes = C.c_int, [C.c_int]
        L.orc_simd_test_arith.restype, L.orc_simd_test_arith.argtypes = C.c_int, [C.c_int, u64p, u64p, u64p, u64p]
        L.orc_batch_inverse.argtypes = [u64p, sz]
        _lib = L
    return _lib


class OracleCircuit:
    def __init__(self, blob):
        self.h = lib().orc_circuit_load(blob, len(blob))
        if not self.h:
            raise RuntimeError("oracle failed to load circuit blob")
        self.blob = blob

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_circuit_free(self.h)
            self.h = None

    def verifier_data(self):
        n = lib().orc_verifier_data(self.h, None, 0)
        out = (C.c_uint64 * n)()
        lib().orc_verifier_data(self.h, out, n)
        return list(out)

    def set_fault(self, kind, a=0, b=0, delta=1):
        lib().orc_set_fault(self.h, kind, a, b, delta)

    def row_gate_kind(self, row):
        return lib().orc_row_gate_kind(self.h, row)

    def wire_slot(self, col, row):
        return lib().orc_wire_slot(self.h, col, row)

    def ops(self):
        k, o = C.c_uint32(), C.c_uint32()
        out = []
        for i in range(lib().orc_num_ops(self.h)):
            lib().orc_get_op(self.h, i, C.byref(k), C.byref(o))
            out.append((k.value, o.value))
        return out

    def set_zk(self, seed, proof_index):
        lib().orc_set_zk(self.h, seed, proof_index)

    def set_zk_key(self, key4, proof_index):
        lib().orc_set_zk_key(self.h, (C.c_uint64 * 4)(*key4), proof_index)

    def prove(self, pw_map, trace=False, cap=1 << 22):
        """Returns (status, proof bytes or None)."""
        ts = (C.c_uint64 * len(pw_map))(*pw_map.keys())
        vs = (C.c_uint64 * len(pw_map))(*pw_map.values())
        buf = C.create_string_buffer(cap)
        n = C.c_size_t()
        st = lib().orc_prove(self.h, ts, vs, len(pw_map), buf, cap, C.byref(n), int(trace))
        if st:
            return st, None
        return 0, buf.raw[: n.value]

    def generate_witness(self, pw_map, cap):
        """(status, wires [num_wires][n] column-major) -- witness generation only, no proving."""
        ts = (C.c_uint64 * len(pw_map))(*pw_map.keys())
        vs = (C.c_uint64 * len(pw_map))(*pw_map.values())
        out = (C.c_uint64 * cap)()
        st = lib().orc_generate_witness(self.h, ts, vs, len(pw_map), out)
        return st, (list(out) if st == 0 else None)

    @staticmethod
    def last_stage_seconds():
        out = (C.c_double * 6)()
        lib().orc_last_stage_seconds(out)
        return dict(zip(("witness", "commitments", "partial_products_lookups", "quotient", "openings", "fri"), out))

    def trace(self, name):
        n = lib().orc_trace_len(self.h, name.encode())
        out = (C.c_uint64 * n)()
        lib().orc_trace_get(self.h, name.encode(), out, n)
        return list(out)

    def trace_bytes(self, name):
        """The same buffer as little-endian u64 bytes (for hashing 10^8-word buffers without building Python ints)."""
        n = lib().orc_trace_len(self.h, name.encode())
        out = (C.c_uint64 * max(n, 1))()
        lib().orc_trace_get(self.h, name.encode(), out, n)
        return bytes(memoryview(out).cast("B")[: 8 * n])

    def generate_witness_bytes(self, pw_map, cap):
        """(status, wires [num_wires][n] column-major as little-endian u64 bytes)."""
        ts = (C.c_uint64 * len(pw_map))(*pw_map.keys())
        vs = (C.c_uint64 * len(pw_map))(*pw_map.values())
        out = (C.c_uint64 * cap)()
        st = lib().orc_generate_witness(self.h, ts, vs, len(pw_map), out)
        return st, (bytes(memoryview(out).cast("B")) if st == 0 else None)


def gcm_encrypt(key, iv, pt):
    ct, tag = C.create_string_buffer(max(len(pt), 1)), C.create_string_buffer(16)
    lib().orc_gcm_encrypt(bytes(key), len(key) // 4, bytes(iv), bytes(pt), len(pt), ct, tag)
    return ct.raw[: len(pt)], tag.raw


def encrypt_block(key, block):
    out = C.create_string_buffer(16)
    lib().orc_aes_encrypt_block(bytes(key), len(key) // 4, bytes(block), out)
    return out.raw

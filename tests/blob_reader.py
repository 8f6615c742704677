"""Reads the product's circuit blob (layout: plonky2-aes_amd/csrc/circuit.h `serialize`, DESIGN.md "Circuit blob") into
numpy arrays -- test-side only, so that a circuit shape derived elsewhere (oracle/oracle_builder.py) can be compared with
what the product compiled."""
import struct

import numpy as np


class Blob:
    def __init__(self, data):
        self.d, self.pos = memoryview(data), 0
        assert bytes(self.d[:8]) == b"P2AESCIR"
        self.pos = 8
        self.version = self.u32()
        self.cfg = struct.unpack_from("<12I", self.d, self.pos)
        self.pos += 48
        self.degree_bits = self.u32()
        self.n = 1 << self.degree_bits
        self.gates = self.arr("<u4")
        self.selector_index = self.arr("<u4")
        self.groups = self.arr("<u4", 2).reshape(-1, 2)
        self.num_lookup_selectors = self.u32()
        self.num_gate_constraints = self.u32()
        self.constants = self.arr("<u8").reshape(-1, self.n)
        self.sigmas = self.arr("<u8").reshape(-1, self.n)
        self.k_is = self.arr("<u8")
        self.luts = [self.arr("<u2", 2).reshape(-1, 2) for _ in range(self.u32())]
        self.lookup_rows = self.arr("<u4", 3).reshape(-1, 3)
        self.num_lookups = self.arr("<u4")

    def u32(self):
        v = struct.unpack_from("<I", self.d, self.pos)[0]
        self.pos += 4
        return v

    def arr(self, dtype, per=1):
        n = struct.unpack_from("<Q", self.d, self.pos)[0]
        self.pos += 8
        a = np.frombuffer(self.d, dtype=dtype, count=n * per, offset=self.pos)
        self.pos += a.nbytes
        return a

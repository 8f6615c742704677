// Keccak-256 (the original padding, 0x01 ... 0x80, rate 136), host only.  plonky2 tags every LookupGate / LookupTableGate
// with the Keccak hash of its table (gates/lookup.rs `new_from_table`: keccak over the pairs, each as input u16 LE | output
// u16 LE) and the hash is part of the gate's id(), so it decides how many gate types a circuit has and how they sort:
// builder.h needs it to lay out the selector polynomials as upstream does.
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <vector>

namespace p2 {

static inline void keccak_f1600(uint64_t s[25]) {
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5 y]
    uint64_t rc_lfsr = 1;
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; x++) {
            uint64_t t = c[(x + 4) % 5], u = c[(x + 1) % 5];
            uint64_t d = t ^ ((u << 1) | (u >> 63));
            for (int y = 0; y < 5; y++) s[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) {
                uint64_t v = s[x + 5 * y];
                int r = ROT[x + 5 * y];
                b[y + 5 * ((2 * x + 3 * y) % 5)] = r ? (v << r) | (v >> (64 - r)) : v;
            }
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        uint64_t rc = 0;
        for (int j = 0; j < 7; j++) {  // round constant bits from the degree-8 LFSR x^8 + x^6 + x^5 + x^4 + 1
            if (rc_lfsr & 1) rc ^= 1ull << ((1 << j) - 1);
            rc_lfsr = (rc_lfsr & 0x80) ? ((rc_lfsr << 1) ^ 0x171) & 0xFF : (rc_lfsr << 1);
        }
        s[0] ^= rc;
    }
}

static inline std::array<uint8_t, 32> keccak256(const uint8_t* data, size_t len) {
    const size_t RATE = 136;
    uint64_t s[25] = {0};
    std::vector<uint8_t> msg(data, data + len);
    msg.push_back(0x01);
    while (msg.size() % RATE) msg.push_back(0);
    msg.back() |= 0x80;
    for (size_t off = 0; off < msg.size(); off += RATE) {
        for (size_t i = 0; i < RATE / 8; i++) {
            uint64_t w;
            memcpy(&w, &msg[off + 8 * i], 8);
            s[i] ^= w;
        }
        keccak_f1600(s);
    }
    std::array<uint8_t, 32> out;
    memcpy(out.data(), s, 32);
    return out;
}

}  // namespace p2

// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_field.h header).
//
// CPU restatement of the reference's native AES / AES-GCM:
//   aes-gcm/src/native_aes.rs  : encrypt_block :27-54, sub_bytes :56, shift_rows :66, mix_columns :70,
//                                gf_2_8_mul :82-98, add_round_key :100, key_expansion :111-131
//   aes-gcm/src/native_gcm.rs  : encrypt :16-68, gctr :71-108, ghash :111-123, gf_2_128_mul :133-158,
//                                right_shift_one :224, inc32 :234-242, msb_t :245-265
// Pinned by the reference's own vectors (tests/golden/aes_kat.json): FIPS-197 App. B block
// (native_aes.rs:209-220), App. A key prefixes (:167-203), GF(2^8) products (circuit_aes.rs:488-498) and the
// four NIST CAVP GCM vectors (native_gcm.rs:290-329).
//
// Style differs on purpose from the product's aes_gadgets.h: the S-box is computed (GF(2^8) inverse + affine
// map) instead of tabulated, the state is a flat column-major 16-byte block, GHASH works on two 64-bit halves.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace orc_aes {

static inline uint8_t xtime(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }
static inline uint8_t gmul(uint8_t a, uint8_t b) {
    uint8_t r = 0;
    while (b) {
        if (b & 1) r ^= a;
        a = xtime(a);
        b >>= 1;
    }
    return r;
}
static inline uint8_t sbox(uint8_t x) {
    // multiplicative inverse (x^254), then affine transform (FIPS-197 section 5.1.1)
    uint8_t inv = 0;
    if (x) {
        uint8_t p = x, acc = 1;
        int e = 254;
        while (e) {
            if (e & 1) acc = gmul(acc, p);
            p = gmul(p, p);
            e >>= 1;
        }
        inv = acc;
    }
    uint8_t r = inv;
    for (int i = 1; i <= 4; i++) r ^= (uint8_t)((inv << i) | (inv >> (8 - i)));
    return r ^ 0x63;
}
struct Tables {
    uint8_t S[256];
    Tables() {
        for (int i = 0; i < 256; i++) S[i] = sbox((uint8_t)i);
    }
};
static inline const Tables& tables() {
    static Tables t;
    return t;
}

// round keys as bytes: rk[16*r + 4*c + i] = word (4r+c), byte i
static inline std::vector<uint8_t> expand_key(const uint8_t* key, int nk) {
    const uint8_t* S = tables().S;
    int nr = nk + 6, words = 4 * (nr + 1);
    std::vector<uint8_t> w(4 * words);
    memcpy(w.data(), key, 4 * nk);
    uint8_t rc = 1;
    for (int i = nk; i < words; i++) {
        uint8_t t[4];
        memcpy(t, &w[4 * (i - 1)], 4);
        if (i % nk == 0) {
            uint8_t t0 = t[0];
            t[0] = S[t[1]] ^ rc;
            t[1] = S[t[2]];
            t[2] = S[t[3]];
            t[3] = S[t0];
            rc = xtime(rc);
        } else if (nk > 6 && i % nk == 4) {
            for (int j = 0; j < 4; j++) t[j] = S[t[j]];
        }
        for (int j = 0; j < 4; j++) w[4 * i + j] = w[4 * (i - nk) + j] ^ t[j];
    }
    return w;
}
// in/out: 16 bytes, standard AES byte order (column-major state)
static inline void encrypt_block(const uint8_t* rk, int nr, const uint8_t* in, uint8_t* out) {
    const uint8_t* S = tables().S;
    uint8_t s[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
    for (int r = 1; r <= nr; r++) {
        uint8_t t[16];
        // SubBytes + ShiftRows: byte at (row i, col c) comes from (row i, col c+i)
        for (int c = 0; c < 4; c++)
            for (int i = 0; i < 4; i++) t[4 * c + i] = S[s[4 * ((c + i) % 4) + i]];
        if (r < nr) {
            for (int c = 0; c < 4; c++) {
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c + 0] = xtime(a0) ^ (xtime(a1) ^ a1) ^ a2 ^ a3;
                s[4 * c + 1] = a0 ^ xtime(a1) ^ (xtime(a2) ^ a2) ^ a3;
                s[4 * c + 2] = a0 ^ a1 ^ xtime(a2) ^ (xtime(a3) ^ a3);
                s[4 * c + 3] = (xtime(a0) ^ a0) ^ a1 ^ a2 ^ xtime(a3);
            }
        } else {
            memcpy(s, t, 16);
        }
        for (int i = 0; i < 16; i++) s[i] ^= rk[16 * r + i];
    }
    memcpy(out, s, 16);
}

static inline void gf128_mul(const uint8_t* x, const uint8_t* y, uint8_t* out) {
    uint64_t zh = 0, zl = 0, vh = 0, vl = 0;
    for (int i = 0; i < 8; i++) {
        vh = (vh << 8) | y[i];
        vl = (vl << 8) | y[8 + i];
    }
    for (int i = 0; i < 128; i++) {
        if ((x[i >> 3] >> (7 - (i & 7))) & 1) {
            zh ^= vh;
            zl ^= vl;
        }
        uint64_t lsb = vl & 1;
        vl = (vl >> 1) | (vh << 63);
        vh >>= 1;
        if (lsb) vh ^= 0xE100000000000000ull;
    }
    for (int i = 0; i < 8; i++) {
        out[i] = (uint8_t)(zh >> (56 - 8 * i));
        out[8 + i] = (uint8_t)(zl >> (56 - 8 * i));
    }
}
static inline void ghash(const uint8_t* h, const uint8_t* x, size_t len, uint8_t* out) {
    uint8_t y[16] = {0};
    for (size_t off = 0; off < len; off += 16) {
        uint8_t t[16];
        for (int i = 0; i < 16; i++) t[i] = y[i] ^ x[off + i];
        gf128_mul(t, h, y);
    }
    memcpy(out, y, 16);
}
static inline void ctr_inc32(uint8_t* cb) {
    for (int i = 15; i >= 12; i--)
        if (++cb[i]) break;
}
static inline void gctr(const uint8_t* rk, int nr, const uint8_t* icb, const uint8_t* x, size_t len, uint8_t* y) {
    uint8_t cb[16], ks[16];
    memcpy(cb, icb, 16);
    for (size_t off = 0; off < len; off += 16) {
        if (off) ctr_inc32(cb);
        encrypt_block(rk, nr, cb, ks);
        for (size_t j = 0; j < 16 && off + j < len; j++) y[off + j] = x[off + j] ^ ks[j];
    }
}
// AES-GCM, 96-bit IV, no AAD, 128-bit tag
static inline void gcm_encrypt(const uint8_t* key, int nk, const uint8_t* iv, const uint8_t* pt, size_t len, uint8_t* ct, uint8_t* tag) {
    int nr = nk + 6;
    auto rk = expand_key(key, nk);
    uint8_t zero[16] = {0}, h[16], j0[16] = {0}, j1[16];
    encrypt_block(rk.data(), nr, zero, h);
    memcpy(j0, iv, 12);
    j0[15] = 1;
    memcpy(j1, j0, 16);
    ctr_inc32(j1);
    gctr(rk.data(), nr, j1, pt, len, ct);
    size_t padded = (len + 15) / 16 * 16;
    std::vector<uint8_t> g(padded + 16, 0);
    if (len) memcpy(g.data(), ct, len);
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) g[padded + 8 + i] = (uint8_t)(bits >> (56 - 8 * i));
    uint8_t s[16];
    ghash(h, g.data(), g.size(), s);
    gctr(rk.data(), nr, j0, s, 16, tag);
}

}  // namespace orc_aes

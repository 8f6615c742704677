// AES / AES-GCM gadgets over the CircuitBuilder mirror, one function per reference function, same names.
//
// Mirrors aes-gcm/src/circuit_aes.rs (trait CircuitBuilderAESState :41-174, impl :176-275, LUT builders
// :299-335, byte_xor :350-358) and aes-gcm/src/circuit_gcm.rs (AesGcmTarget::build :49-172, set_targets
// :174-208, gctr_target :212-260, ghash_target :262-289, gf_2_128_mul_target :290-326,
// right_shift_one_target :327-348, inc32_target :350-368, LUTs :392-425).  Const generics <NK,NB,NR,L,TAG>
// become run-time arguments (NB is always 4).  The native (non-circuit) cipher used to produce witness values
// mirrors aes-gcm/src/native_aes.rs and native_gcm.rs.
#pragma once
#include <array>

#include "builder.h"

namespace p2 {
namespace aes {

typedef Target ByteTarget;
typedef std::array<std::array<ByteTarget, 4>, 4> StateTarget;
typedef std::array<ByteTarget, 4> WordTarget;

// ---------------------------------------------------------------- constants (aes-gcm/src/constants.rs)
static const uint8_t SBOX[256] = {
    0x63, 0x7c, 0x77, 0x7b, 0xf2, 0x6b, 0x6f, 0xc5, 0x30, 0x01, 0x67, 0x2b, 0xfe, 0xd7, 0xab, 0x76, 0xca, 0x82, 0xc9, 0x7d, 0xfa, 0x59,
    0x47, 0xf0, 0xad, 0xd4, 0xa2, 0xaf, 0x9c, 0xa4, 0x72, 0xc0, 0xb7, 0xfd, 0x93, 0x26, 0x36, 0x3f, 0xf7, 0xcc, 0x34, 0xa5, 0xe5, 0xf1,
    0x71, 0xd8, 0x31, 0x15, 0x04, 0xc7, 0x23, 0xc3, 0x18, 0x96, 0x05, 0x9a, 0x07, 0x12, 0x80, 0xe2, 0xeb, 0x27, 0xb2, 0x75, 0x09, 0x83,
    0x2c, 0x1a, 0x1b, 0x6e, 0x5a, 0xa0, 0x52, 0x3b, 0xd6, 0xb3, 0x29, 0xe3, 0x2f, 0x84, 0x53, 0xd1, 0x00, 0xed, 0x20, 0xfc, 0xb1, 0x5b,
    0x6a, 0xcb, 0xbe, 0x39, 0x4a, 0x4c, 0x58, 0xcf, 0xd0, 0xef, 0xaa, 0xfb, 0x43, 0x4d, 0x33, 0x85, 0x45, 0xf9, 0x02, 0x7f, 0x50, 0x3c,
    0x9f, 0xa8, 0x51, 0xa3, 0x40, 0x8f, 0x92, 0x9d, 0x38, 0xf5, 0xbc, 0xb6, 0xda, 0x21, 0x10, 0xff, 0xf3, 0xd2, 0xcd, 0x0c, 0x13, 0xec,
    0x5f, 0x97, 0x44, 0x17, 0xc4, 0xa7, 0x7e, 0x3d, 0x64, 0x5d, 0x19, 0x73, 0x60, 0x81, 0x4f, 0xdc, 0x22, 0x2a, 0x90, 0x88, 0x46, 0xee,
    0xb8, 0x14, 0xde, 0x5e, 0x0b, 0xdb, 0xe0, 0x32, 0x3a, 0x0a, 0x49, 0x06, 0x24, 0x5c, 0xc2, 0xd3, 0xac, 0x62, 0x91, 0x95, 0xe4, 0x79,
    0xe7, 0xc8, 0x37, 0x6d, 0x8d, 0xd5, 0x4e, 0xa9, 0x6c, 0x56, 0xf4, 0xea, 0x65, 0x7a, 0xae, 0x08, 0xba, 0x78, 0x25, 0x2e, 0x1c, 0xa6,
    0xb4, 0xc6, 0xe8, 0xdd, 0x74, 0x1f, 0x4b, 0xbd, 0x8b, 0x8a, 0x70, 0x3e, 0xb5, 0x66, 0x48, 0x03, 0xf6, 0x0e, 0x61, 0x35, 0x57, 0xb9,
    0x86, 0xc1, 0x1d, 0x9e, 0xe1, 0xf8, 0x98, 0x11, 0x69, 0xd9, 0x8e, 0x94, 0x9b, 0x1e, 0x87, 0xe9, 0xce, 0x55, 0x28, 0xdf, 0x8c, 0xa1,
    0x89, 0x0d, 0xbf, 0xe6, 0x42, 0x68, 0x41, 0x99, 0x2d, 0x0f, 0xb0, 0x54, 0xbb, 0x16};
static const uint8_t RCON[11] = {0x00, 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36};
static const size_t TAG_LEN = 128;

// ---------------------------------------------------------------- native cipher (native_aes.rs, native_gcm.rs)
typedef std::array<std::array<uint8_t, 4>, 4> State;

inline uint8_t gf_2_8_mul(uint8_t a, uint8_t b) {
    uint8_t r = 0;
    for (int i = 0; i < 8; i++) {
        if (b & 1) r ^= a;
        uint8_t hb = a & 0x80;
        a <<= 1;
        if (hb) a ^= 0x1b;
        b >>= 1;
    }
    return r;
}
template <class T>
inline std::array<std::array<T, 4>, 4> shift_rows(const std::array<std::array<T, 4>, 4>& s) {
    std::array<std::array<T, 4>, 4> r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i][j] = s[i][(i + j) % 4];
    return r;
}
template <class T>
inline std::array<T, 4> rot_word(const std::array<T, 4>& w) {
    return {w[1], w[2], w[3], w[0]};
}
inline std::array<uint8_t, 16> flatten_state(const State& s) {
    std::array<uint8_t, 16> r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i * 4 + j] = s[j][i];
    return r;
}
// expanded key as 4*(NR+1) words
inline std::vector<std::array<uint8_t, 4>> key_expansion(int NK, int NR, const uint8_t* key) {
    std::vector<std::array<uint8_t, 4>> w(4 * (NR + 1));
    for (int i = 0; i < NK; i++)
        for (int j = 0; j < 4; j++) w[i][j] = key[4 * i + j];
    for (int i = NK; i < 4 * (NR + 1); i++) {
        std::array<uint8_t, 4> temp = w[i - 1];
        if (i % NK == 0) {
            temp = rot_word(temp);
            for (auto& b : temp) b = SBOX[b];
            temp[0] ^= RCON[i / NK];
        } else if (NK > 6 && i % NK == 4) {
            for (auto& b : temp) b = SBOX[b];
        }
        for (int j = 0; j < 4; j++) w[i][j] = w[i - NK][j] ^ temp[j];
    }
    return w;
}
inline State encrypt_block(int NR, const uint8_t* input, const std::vector<std::array<uint8_t, 4>>& w) {
    State s;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) s[i][j] = input[i + 4 * j];
    auto add_round_key = [&](int r) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) s[i][j] ^= w[4 * r + j][i];
    };
    auto sub_bytes = [&]() {
        for (auto& row : s)
            for (auto& b : row) b = SBOX[b];
    };
    add_round_key(0);
    for (int round = 1; round < NR; round++) {
        sub_bytes();
        s = shift_rows(s);
        State r;
        for (int c = 0; c < 4; c++) {
            r[0][c] = gf_2_8_mul(2, s[0][c]) ^ gf_2_8_mul(3, s[1][c]) ^ s[2][c] ^ s[3][c];
            r[1][c] = s[0][c] ^ gf_2_8_mul(2, s[1][c]) ^ gf_2_8_mul(3, s[2][c]) ^ s[3][c];
            r[2][c] = s[0][c] ^ s[1][c] ^ gf_2_8_mul(2, s[2][c]) ^ gf_2_8_mul(3, s[3][c]);
            r[3][c] = gf_2_8_mul(3, s[0][c]) ^ s[1][c] ^ s[2][c] ^ gf_2_8_mul(2, s[3][c]);
        }
        s = r;
        add_round_key(round);
    }
    sub_bytes();
    s = shift_rows(s);
    add_round_key(NR);
    return s;
}
inline void inc32(uint8_t* b) {
    uint32_t ctr = ((uint32_t)b[12] << 24) | ((uint32_t)b[13] << 16) | ((uint32_t)b[14] << 8) | b[15];
    ctr += 1;
    b[12] = ctr >> 24;
    b[13] = ctr >> 16;
    b[14] = ctr >> 8;
    b[15] = ctr;
}
inline std::vector<uint8_t> gctr(int NR, const std::vector<std::array<uint8_t, 4>>& key, const uint8_t* icb, const uint8_t* x, size_t len) {
    std::vector<uint8_t> y(len);
    uint8_t cb[16];
    memcpy(cb, icb, 16);
    for (size_t off = 0, i = 0; off < len; off += 16, i++) {
        if (i > 0) inc32(cb);
        auto ks = flatten_state(encrypt_block(NR, cb, key));
        size_t l = std::min<size_t>(16, len - off);
        for (size_t j = 0; j < l; j++) y[off + j] = x[off + j] ^ ks[j];
    }
    return y;
}
inline void right_shift_one(uint8_t* block) {
    uint8_t carry = 0;
    for (int i = 0; i < 16; i++) {
        uint8_t nc = block[i] & 1;
        block[i] = (block[i] >> 1) | (carry << 7);
        carry = nc;
    }
}
inline std::array<uint8_t, 16> gf_2_128_mul(const uint8_t* x, const uint8_t* y) {
    std::array<uint8_t, 16> z{};
    uint8_t v[16];
    memcpy(v, y, 16);
    for (int i = 0; i < 128; i++) {
        if ((x[i / 8] >> (7 - (i % 8))) & 1)
            for (int b = 0; b < 16; b++) z[b] ^= v[b];
        uint8_t lsb = v[15] & 1;
        right_shift_one(v);
        if (lsb) v[0] ^= 0xE1;
    }
    return z;
}
inline std::array<uint8_t, 16> ghash(const uint8_t* h, const uint8_t* x, size_t len) {
    std::array<uint8_t, 16> y{};
    for (size_t i = 0; i < len / 16; i++) {
        uint8_t t[16];
        for (int b = 0; b < 16; b++) t[b] = y[b] ^ x[16 * i + b];
        y = gf_2_128_mul(t, h);
    }
    return y;
}
// native_gcm.rs:16-68.  Returns (ciphertext, 16-byte tag); no AAD, 96-bit nonce.
inline void gcm_encrypt(int NK, int NR, const uint8_t* key, const uint8_t* nonce, const uint8_t* pt, size_t len, uint8_t* ct, uint8_t* tag) {
    auto w = key_expansion(NK, NR, key);
    uint8_t zero[16] = {0};
    auto h = flatten_state(encrypt_block(NR, zero, w));
    uint8_t j0[16] = {0};
    memcpy(j0, nonce, 12);
    j0[15] = 1;
    uint8_t j0i[16];
    memcpy(j0i, j0, 16);
    inc32(j0i);
    auto c = gctr(NR, w, j0i, pt, len);
    size_t u = (16 - len % 16) % 16;
    std::vector<uint8_t> gin(c);
    gin.resize(len + u, 0);
    uint64_t clen = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) gin.push_back(0);
    for (int i = 7; i >= 0; i--) gin.push_back((uint8_t)(clen >> (8 * i)));
    auto s = ghash(h.data(), gin.data(), gin.size());
    auto t = gctr(NR, w, j0, s.data(), 16);
    if (len) memcpy(ct, c.data(), len);
    memcpy(tag, t.data(), 16);
}

// ---------------------------------------------------------------- LUT builders
inline size_t sbox_lut(CircuitBuilder& b) {  // circuit_aes.rs:299
    std::vector<std::pair<u16, u16>> t(256);
    for (int i = 0; i < 256; i++) t[i] = {(u16)i, (u16)SBOX[i]};
    return b.add_lookup_table_from_pairs(t);
}
inline size_t byte_xor_lut(CircuitBuilder& b) {  // circuit_aes.rs:309
    std::vector<std::pair<u16, u16>> t;
    t.reserve(65536);
    for (int x = 0; x < 256; x++)
        for (int y = 0; y < 256; y++) t.push_back({(u16)((x << 8) + y), (u16)(x ^ y)});
    return b.add_lookup_table_from_pairs(t);
}
inline size_t gf_2_8_mul_lut(CircuitBuilder& b) {  // circuit_aes.rs:321
    std::vector<std::pair<u16, u16>> t;
    t.reserve(65536);
    for (int x = 0; x < 256; x++)
        for (int y = 0; y < 256; y++) t.push_back({(u16)((x << 8) + y), (u16)gf_2_8_mul((uint8_t)x, (uint8_t)y)});
    return b.add_lookup_table_from_pairs(t);
}
inline size_t u8_unit_right_shift_lut(CircuitBuilder& b) {  // circuit_gcm.rs:392
    std::vector<std::pair<u16, u16>> t(256);
    for (int x = 0; x < 256; x++) t[x] = {(u16)x, (u16)(x >> 1)};
    return b.add_lookup_table_from_pairs(t);
}
inline size_t u8_bitref_lut(CircuitBuilder& b) {  // circuit_gcm.rs:407
    std::vector<std::pair<u16, u16>> t;
    for (int x = 0; x < 256; x++)
        for (int i = 0; i < 8; i++) t.push_back({(u16)((x << 3) + i), (u16)((x >> i) & 1)});
    return b.add_lookup_table_from_pairs(t);
}

// ---------------------------------------------------------------- CircuitBuilderAESState (circuit_aes.rs:41-275)
inline ByteTarget add_virtual_byte_target_unsafe(CircuitBuilder& b) { return b.add_virtual_target(); }
inline void assert_byte(CircuitBuilder& b, Target x, size_t u8_table_idx) { b.add_lookup_from_index(x, u8_table_idx); }
inline ByteTarget add_virtual_byte_target(CircuitBuilder& b, size_t u8_table_idx) {
    ByteTarget t = add_virtual_byte_target_unsafe(b);
    assert_byte(b, t, u8_table_idx);
    return t;
}
inline ByteTarget byte_constant(CircuitBuilder& b, uint8_t c) { return b.constant(c); }
inline StateTarget add_virtual_state_target_unsafe(CircuitBuilder& b) {
    StateTarget s;
    for (auto& row : s)
        for (auto& t : row) t = add_virtual_byte_target_unsafe(b);
    return s;
}
inline StateTarget add_virtual_state_target(CircuitBuilder& b, size_t u8_table_idx) {
    StateTarget s;
    for (auto& row : s)
        for (auto& t : row) t = add_virtual_byte_target(b, u8_table_idx);
    return s;
}
inline std::array<ByteTarget, 16> flatten(const StateTarget& s) {  // circuit_aes.rs:28
    std::array<ByteTarget, 16> r;
    for (int i = 0; i < 16; i++) r[i] = s[i % 4][i / 4];
    return r;
}
inline StateTarget from_flat(const std::array<ByteTarget, 16>& f) {  // circuit_aes.rs:32
    StateTarget s;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) s[i][j] = f[j * 4 + i];
    return s;
}
inline ByteTarget byte_xor(CircuitBuilder& b, size_t xor_lut_idx, ByteTarget x, ByteTarget y) {  // :350
    Target idx = b.mul_const_add(1 << 8, x, y);
    return b.add_lookup_from_index(idx, xor_lut_idx);
}
inline ByteTarget gf_2_8_add(CircuitBuilder& b, size_t xor_lut_idx, ByteTarget x, ByteTarget y) { return byte_xor(b, xor_lut_idx, x, y); }
inline ByteTarget gf_2_8_mul_t(CircuitBuilder& b, size_t mul_lut_idx, ByteTarget x, ByteTarget y) {  // :243
    Target idx = b.mul_const_add(1 << 8, x, y);
    return b.add_lookup_from_index(idx, mul_lut_idx);
}
inline WordTarget state_sub_word(CircuitBuilder& b, size_t sbox_lut_idx, const WordTarget& w) {  // :189
    WordTarget r;
    for (int i = 0; i < 4; i++) r[i] = b.add_lookup_from_index(w[i], sbox_lut_idx);
    return r;
}
inline StateTarget state_sub_bytes(CircuitBuilder& b, size_t sbox_lut_idx, const StateTarget& s) {  // :99
    StateTarget r;
    for (int i = 0; i < 4; i++) r[i] = state_sub_word(b, sbox_lut_idx, s[i]);
    return r;
}
inline ByteTarget bytearray_ip4(CircuitBuilder& b, size_t xor_lut, size_t mul_lut, const WordTarget& x, const WordTarget& y) {  // :253
    ByteTarget acc = b.zero();
    for (int i = 0; i < 4; i++) {
        ByteTarget prod = gf_2_8_mul_t(b, mul_lut, x[i], y[i]);
        acc = gf_2_8_add(b, xor_lut, acc, prod);
    }
    return acc;
}
inline StateTarget state_mix_matrix(CircuitBuilder& b) {  // :337
    ByteTarget one = byte_constant(b, 1), two = byte_constant(b, 2), three = byte_constant(b, 3);
    StateTarget m = {{{two, three, one, one}, {one, two, three, one}, {one, one, two, three}, {three, one, one, two}}};
    return m;
}
inline StateTarget state_mix_columns(CircuitBuilder& b, size_t xor_lut, size_t mul_lut, const StateTarget& mix, const StateTarget& s) {  // :109
    StateTarget out_cols;
    for (int i = 0; i < 4; i++) {
        WordTarget col = {s[0][i], s[1][i], s[2][i], s[3][i]};
        for (int m = 0; m < 4; m++) out_cols[i][m] = bytearray_ip4(b, xor_lut, mul_lut, mix[m], col);
    }
    StateTarget r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i][j] = out_cols[j][i];
    return r;
}
inline StateTarget state_add_round_key(CircuitBuilder& b, size_t xor_lut, const WordTarget* round_key, const StateTarget& s) {  // :124
    StateTarget r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i][j] = gf_2_8_add(b, xor_lut, s[i][j], round_key[j][i]);
    return r;
}
inline std::vector<WordTarget> key_expansion_t(CircuitBuilder& b, int NK, int NR, size_t xor_lut, size_t sbox_lut_idx, const std::vector<ByteTarget>& key) {  // :197
    std::array<ByteTarget, 11> rcon;
    for (int i = 0; i < 11; i++) rcon[i] = byte_constant(b, RCON[i]);
    std::vector<WordTarget> st(NK);
    for (int i = 0; i < NK; i++)
        for (int j = 0; j < 4; j++) st[i][j] = key[4 * i + j];
    for (int i = NK; i < 4 * (NR + 1); i++) {
        WordTarget offset;
        if (i % NK == 0) {
            WordTarget term = state_sub_word(b, sbox_lut_idx, rot_word(st[i - 1]));
            offset = term;
            offset[0] = gf_2_8_add(b, xor_lut, term[0], rcon[i / NK]);
        } else if (NK > 6 && i % NK == 4) {
            offset = state_sub_word(b, sbox_lut_idx, st[i - 1]);
        } else {
            offset = st[i - 1];
        }
        WordTarget cur;
        for (int j = 0; j < 4; j++) cur[j] = gf_2_8_add(b, xor_lut, st[i - NK][j], offset[j]);
        st.push_back(cur);
    }
    return st;
}
inline StateTarget encrypt_block_t(CircuitBuilder& b, int NR, size_t xor_lut, size_t mul_lut, size_t sbox_lut_idx, const StateTarget& mix,
                                   StateTarget s, const std::vector<WordTarget>& w) {  // :76
    s = state_add_round_key(b, xor_lut, &w[0], s);
    for (int i = 1; i < NR; i++) {
        s = state_sub_bytes(b, sbox_lut_idx, s);
        s = shift_rows(s);
        s = state_mix_columns(b, xor_lut, mul_lut, mix, s);
        s = state_add_round_key(b, xor_lut, &w[4 * i], s);
    }
    s = state_sub_bytes(b, sbox_lut_idx, s);
    s = shift_rows(s);
    return state_add_round_key(b, xor_lut, &w[4 * NR], s);
}
inline ByteTarget zero_byte(CircuitBuilder& b) { return b.zero(); }
inline StateTarget empty_state(CircuitBuilder& b) {
    StateTarget s;
    for (auto& row : s)
        for (auto& t : row) t = b.zero();
    return s;
}

// ---------------------------------------------------------------- GCM (circuit_gcm.rs)
typedef std::array<ByteTarget, 16> BlockTarget;

inline BlockTarget xor_blocks(CircuitBuilder& b, size_t xor_lut, const BlockTarget& b1, const BlockTarget& b2) {  // :370
    BlockTarget r;
    for (int i = 0; i < 16; i++) r[i] = byte_xor(b, xor_lut, b1[i], b2[i]);
    return r;
}
inline ByteTarget u8_unit_right_shift(CircuitBuilder& b, size_t lut, ByteTarget x) { return b.add_lookup_from_index(x, lut); }  // :398
inline BoolTarget u8_bitref(CircuitBuilder& b, size_t lut, ByteTarget x, ByteTarget i) {  // :417
    Target idx = b.mul_const_add(8, x, i);
    return BoolTarget{b.add_lookup_from_index(idx, lut)};
}
inline BlockTarget right_shift_one_target(CircuitBuilder& b, size_t shift_lut, const BlockTarget& v) {  // :327
    BlockTarget r = v;
    Target carry = b.zero();
    for (int i = 0; i < 16; i++) {
        ByteTarget current = v[i];
        ByteTarget shifted = u8_unit_right_shift(b, shift_lut, current);
        Target next_carry = b.mul_const_add(gl::mul(2, gl::P - 1), shifted, current);
        shifted = b.mul_const_add(1 << 7, carry, shifted);
        r[i] = shifted;
        carry = next_carry;
    }
    return r;
}
inline BlockTarget inc32_target(CircuitBuilder& b, const BlockTarget& block) {  // :350
    BlockTarget r = block;
    Target zero = b.zero();
    Target u8_max = b.constant(255);
    Target carry = b.one();
    for (int byte_index = 15; byte_index >= 12; byte_index--) {
        Target a = block[byte_index];
        Target sum = b.add(a, carry);
        BoolTarget a_is_u8_max = b.is_equal(a, u8_max);
        Target carry_out = b.mul(carry, a_is_u8_max.target);
        r[byte_index] = b.select(a_is_u8_max, zero, sum);
        carry = carry_out;
    }
    return r;
}
inline std::vector<ByteTarget> gctr_target(CircuitBuilder& b, int NR, size_t xor_lut, size_t mul_lut, size_t sbox_lut_idx, const StateTarget& mix,
                                           const std::vector<WordTarget>& key, const BlockTarget& icb, const std::vector<ByteTarget>& x) {  // :212
    const size_t L = x.size();
    std::vector<ByteTarget> y = x;
    BlockTarget cb_i = icb;
    ByteTarget zb = zero_byte(b);
    for (size_t off = 0, i = 0; off < L; off += 16, i++) {
        if (i > 0) cb_i = inc32_target(b, cb_i);
        size_t l = std::min<size_t>(16, L - off);
        BlockTarget x_i;
        x_i.fill(zb);
        for (size_t j = 0; j < l; j++) x_i[j] = x[off + j];
        BlockTarget ciph = flatten(encrypt_block_t(b, NR, xor_lut, mul_lut, sbox_lut_idx, mix, from_flat(cb_i), key));
        BlockTarget y_i;
        size_t n_bytes;
        if (l == 16) {
            y_i = xor_blocks(b, xor_lut, x_i, ciph);
            n_bytes = 16;
        } else {
            // last chunk: MSB_{L%16 bytes}(ciph), zero-padded, then xor over the full block (as the reference does)
            BlockTarget m;
            m.fill(zb);
            for (size_t j = 0; j < L % 16; j++) m[j] = ciph[j];
            y_i = xor_blocks(b, xor_lut, x_i, m);
            n_bytes = L % 16;
        }
        for (size_t j = 0; j < n_bytes; j++) y[off + j] = y_i[j];
    }
    return y;
}
inline BlockTarget gf_2_128_mul_target(CircuitBuilder& b, size_t xor_lut, size_t shift_lut, size_t bitref_lut, const BlockTarget& x, const BlockTarget& y) {  // :290
    ByteTarget zero = zero_byte(b);
    ByteTarget r_first = byte_constant(b, 225);
    BlockTarget z;
    z.fill(zero);
    BlockTarget v = y;
    for (int i = 0; i < 128; i++) {
        int byte_index = i / 8, bit_index = 7 - (i % 8);
        ByteTarget bit_idx_target = byte_constant(b, (uint8_t)bit_index);
        BoolTarget xi = u8_bitref(b, bitref_lut, x[byte_index], bit_idx_target);
        for (int k = 0; k < 16; k++) {
            ByteTarget z_xor_v = byte_xor(b, xor_lut, z[k], v[k]);
            z[k] = b.select(xi, z_xor_v, z[k]);
        }
        BoolTarget lsb = u8_bitref(b, bitref_lut, v[15], zero);
        v = right_shift_one_target(b, shift_lut, v);
        ByteTarget v_xor_r = byte_xor(b, xor_lut, v[0], r_first);
        v[0] = b.select(lsb, v_xor_r, v[0]);
    }
    return z;
}
inline BlockTarget ghash_target(CircuitBuilder& b, size_t xor_lut, size_t shift_lut, size_t bitref_lut, const BlockTarget& h, const std::vector<ByteTarget>& x) {  // :262
    if (x.size() % 16 != 0) throw std::runtime_error("ghash input must be a multiple of 16 bytes");
    ByteTarget zb = zero_byte(b);
    BlockTarget y;
    y.fill(zb);
    for (size_t i = 0; i < x.size() / 16; i++) {
        BlockTarget xi;
        for (int k = 0; k < 16; k++) xi[k] = x[16 * i + k];
        BlockTarget y_xi = xor_blocks(b, xor_lut, y, xi);
        y = gf_2_128_mul_target(b, xor_lut, shift_lut, bitref_lut, y_xi, h);
    }
    return y;
}

// AesGcmTarget<NK, 4, NR, L, TAG> (circuit_gcm.rs:24-209)
struct AesGcmTarget {
    int NK, NR;
    size_t L;
    bool TAG;
    std::vector<ByteTarget> key, nonce, pt, ct, tag;

    static AesGcmTarget build(CircuitBuilder& b, int NK, int NR, size_t L, bool TAG) {
        AesGcmTarget t;
        t.NK = NK;
        t.NR = NR;
        t.L = L;
        t.TAG = TAG;
        size_t sbox = sbox_lut(b), xorl = byte_xor_lut(b), mull = gf_2_8_mul_lut(b);
        for (int i = 0; i < NK * 4; i++) t.key.push_back(add_virtual_byte_target(b, sbox));
        for (int i = 0; i < 12; i++) t.nonce.push_back(add_virtual_byte_target(b, sbox));
        for (size_t i = 0; i < L; i++) t.pt.push_back(add_virtual_byte_target(b, sbox));
        for (size_t i = 0; i < TAG_LEN / 8; i++) t.tag.push_back(add_virtual_byte_target(b, sbox));
        StateTarget mix = state_mix_matrix(b);
        auto expanded_key = key_expansion_t(b, NK, NR, xorl, sbox, t.key);
        StateTarget es = empty_state(b);
        BlockTarget h = flatten(encrypt_block_t(b, NR, xorl, mull, sbox, mix, es, expanded_key));
        ByteTarget zb = zero_byte(b), ob = byte_constant(b, 1);
        BlockTarget j0;
        j0.fill(zb);
        for (int i = 0; i < 12; i++) j0[i] = t.nonce[i];
        j0[15] = ob;
        BlockTarget inc32_j0 = inc32_target(b, j0);
        t.ct = gctr_target(b, NR, xorl, mull, sbox, mix, expanded_key, inc32_j0, t.pt);
        if (!TAG) return t;
        size_t u = (16 - L % 16) % 16;
        uint64_t clen = (uint64_t)L * 8;
        std::vector<ByteTarget> gin(t.ct);
        for (size_t i = 0; i < u; i++) gin.push_back(zb);
        for (int i = 0; i < 8; i++) gin.push_back(byte_constant(b, 0));
        for (int i = 7; i >= 0; i--) gin.push_back(byte_constant(b, (uint8_t)(clen >> (8 * i))));
        size_t shl = u8_unit_right_shift_lut(b), brl = u8_bitref_lut(b);
        BlockTarget s = ghash_target(b, xorl, shl, brl, h, gin);
        std::vector<ByteTarget> sv(s.begin(), s.end());
        auto msb_input = gctr_target(b, NR, xorl, mull, sbox, mix, expanded_key, j0, sv);
        for (size_t i = 0; i < TAG_LEN / 8; i++) b.connect(t.tag[i], msb_input[i]);
        return t;
    }
};

}  // namespace aes
}  // namespace p2

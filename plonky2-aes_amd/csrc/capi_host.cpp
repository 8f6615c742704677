// Host half of the C ABI declared in include/p2aes.h: CircuitBuilder mirror, AES/GCM gadgets, native cipher,
// blob info and the verifier.  No device code here; the GPU prover lives in prover_gpu.hip.
#include <stdlib.h>

#include "../../include/p2aes.h"
#include "aes_gadgets.h"
#include "capi_common.h"
#include "ecgfp5.h"
#include "poseidon_cipher.h"
#include "poseidon_fast.h"
#include "verifier.h"
#include "witness_schedule.h"

using namespace p2;

namespace p2 {
thread_local std::string g_last_error;
}

struct p2_builder {
    CircuitBuilder b;
};


// No exception may cross the C boundary: every entry point that runs builder / gadget code goes through one of these.
// The message lands in p2_last_error(); the return value is the function's error value (an error code, UINT64_MAX for a
// target, (size_t)-1 for an index; a void function leaves its outputs untouched).
template <class F>
static int guarded(F f) {
    try {
        f();
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    } catch (...) {
        return set_error("unknown C++ exception"), P2_ERR_INVALID;
    }
}
template <class F>
static p2_target guarded_t(F f) {
    p2_target r = UINT64_MAX;
    (void)guarded([&] { r = f(); });
    return r;
}
template <class F>
static size_t guarded_sz(F f) {
    size_t r = (size_t)-1;
    (void)guarded([&] { r = f(); });
    return r;
}

extern "C" {

const char* p2_last_error(void) { return p2::g_last_error.c_str(); }

p2_builder* p2_builder_new(void) { return new p2_builder(); }
p2_builder* p2_builder_new_zk(void) {
    Config cfg;
    cfg.zero_knowledge = 1;
    return new p2_builder{CircuitBuilder(cfg)};
}
void p2_builder_free(p2_builder* b) { delete b; }
p2_target p2_builder_add_virtual_target(p2_builder* b) { return guarded_t([&]() -> p2_target { return b->b.add_virtual_target(); }); }
p2_target p2_builder_constant(p2_builder* b, uint64_t c) { return guarded_t([&]() -> p2_target { return b->b.constant(c); }); }
p2_target p2_builder_zero(p2_builder* b) { return guarded_t([&]() -> p2_target { return b->b.zero(); }); }
p2_target p2_builder_one(p2_builder* b) { return guarded_t([&]() -> p2_target { return b->b.one(); }); }
p2_target p2_builder_arithmetic(p2_builder* b, uint64_t c0, uint64_t c1, p2_target m0, p2_target m1, p2_target a) { return guarded_t([&]() -> p2_target {    return b->b.arithmetic(c0 % gl::P, c1 % gl::P, m0, m1, a);}); }
p2_target p2_builder_mul_const_add(p2_builder* b, uint64_t c, p2_target x, p2_target y) { return guarded_t([&]() -> p2_target { return b->b.mul_const_add(c % gl::P, x, y); }); }
p2_target p2_builder_add(p2_builder* b, p2_target x, p2_target y) { return guarded_t([&]() -> p2_target { return b->b.add(x, y); }); }
p2_target p2_builder_sub(p2_builder* b, p2_target x, p2_target y) { return guarded_t([&]() -> p2_target { return b->b.sub(x, y); }); }
p2_target p2_builder_mul(p2_builder* b, p2_target x, p2_target y) { return guarded_t([&]() -> p2_target { return b->b.mul(x, y); }); }
p2_target p2_builder_select(p2_builder* b, p2_target c, p2_target x, p2_target y) { return guarded_t([&]() -> p2_target { return b->b.select(BoolTarget{c}, x, y); }); }
p2_target p2_builder_is_equal(p2_builder* b, p2_target x, p2_target y) { return guarded_t([&]() -> p2_target { return b->b.is_equal(x, y).target; }); }
void p2_builder_connect(p2_builder* b, p2_target x, p2_target y) { (void)guarded([&] { b->b.connect(x, y); }); }
size_t p2_builder_add_lookup_table_from_pairs(p2_builder* b, const uint16_t* pairs, size_t n) {
    return guarded_sz([&]() -> size_t {
        std::vector<std::pair<u16, u16>> t(n);
        for (size_t i = 0; i < n; i++) t[i] = {pairs[2 * i], pairs[2 * i + 1]};
        return b->b.add_lookup_table_from_pairs(t);
    });
}
p2_target p2_builder_add_lookup_from_index(p2_builder* b, p2_target in, size_t lut) {
    try {
        return b->b.add_lookup_from_index(in, lut);
    } catch (std::exception& e) {
        set_error(e.what());
        return UINT64_MAX;
    }
}
size_t p2_builder_num_gates(const p2_builder* b) { return guarded_sz([&]() -> size_t { return b->b.num_gates(); }); }
int p2_builder_build(p2_builder* b, uint8_t** blob, size_t* len) {
    try {
        Circuit c = b->b.build();
        std::vector<uint8_t> v = serialize(c);
        uint8_t* out = (uint8_t*)malloc(v.size());
        if (!out) return set_error("out of memory"), P2_ERR_INVALID;
        memcpy(out, v.data(), v.size());
        *blob = out;
        *len = v.size();
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
void p2_blob_free(uint8_t* blob) { free(blob); }

// ---- gadgets
static aes::StateTarget state_in(const p2_target* s) {
    aes::StateTarget st;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) st[i][j] = s[4 * i + j];
    return st;
}
static void state_out(const aes::StateTarget& st, p2_target* o) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) o[4 * i + j] = st[i][j];
}
static std::vector<aes::WordTarget> words_in(const p2_target* w, size_t n) {
    std::vector<aes::WordTarget> v(n);
    for (size_t i = 0; i < n; i++)
        for (int j = 0; j < 4; j++) v[i][j] = w[4 * i + j];
    return v;
}
static aes::BlockTarget block_in(const p2_target* b) {
    aes::BlockTarget r;
    for (int i = 0; i < 16; i++) r[i] = b[i];
    return r;
}
size_t p2_aes_sbox_lut(p2_builder* b) { return guarded_sz([&]() -> size_t { return aes::sbox_lut(b->b); }); }
size_t p2_aes_byte_xor_lut(p2_builder* b) { return guarded_sz([&]() -> size_t { return aes::byte_xor_lut(b->b); }); }
size_t p2_aes_gf_2_8_mul_lut(p2_builder* b) { return guarded_sz([&]() -> size_t { return aes::gf_2_8_mul_lut(b->b); }); }
size_t p2_gcm_u8_unit_right_shift_lut(p2_builder* b) { return guarded_sz([&]() -> size_t { return aes::u8_unit_right_shift_lut(b->b); }); }
size_t p2_gcm_u8_bitref_lut(p2_builder* b) { return guarded_sz([&]() -> size_t { return aes::u8_bitref_lut(b->b); }); }
p2_target p2_aes_add_virtual_byte_target(p2_builder* b, size_t t) {
    try {
        return aes::add_virtual_byte_target(b->b, t);
    } catch (std::exception& e) {
        set_error(e.what());
        return UINT64_MAX;
    }
}
p2_target p2_aes_add_virtual_byte_target_unsafe(p2_builder* b) { return guarded_t([&]() -> p2_target { return aes::add_virtual_byte_target_unsafe(b->b); }); }
void p2_aes_state_sub_bytes(p2_builder* b, size_t sbox, const p2_target* s, p2_target* out) { (void)guarded([&] { state_out(aes::state_sub_bytes(b->b, sbox, state_in(s)), out); }); }
void p2_aes_state_mix_columns(p2_builder* b, size_t xl, size_t ml, const p2_target* s, p2_target* out) {
    (void)guarded([&] {
        aes::StateTarget mix = aes::state_mix_matrix(b->b);
        state_out(aes::state_mix_columns(b->b, xl, ml, mix, state_in(s)), out);
    });
}
p2_target p2_aes_gf_2_8_mul(p2_builder* b, size_t ml, p2_target x, p2_target y) { return guarded_t([&]() -> p2_target { return aes::gf_2_8_mul_t(b->b, ml, x, y); }); }
p2_target p2_aes_gf_2_8_add(p2_builder* b, size_t xl, p2_target x, p2_target y) { return guarded_t([&]() -> p2_target { return aes::gf_2_8_add(b->b, xl, x, y); }); }
void p2_aes_key_expansion(p2_builder* b, int nk, int nr, size_t xl, size_t sl, const p2_target* key, p2_target* out) {
    (void)guarded([&] {
        std::vector<aes::ByteTarget> k(key, key + 4 * nk);
        auto w = aes::key_expansion_t(b->b, nk, nr, xl, sl, k);
        for (size_t i = 0; i < w.size(); i++)
            for (int j = 0; j < 4; j++) out[4 * i + j] = w[i][j];
    });
}
void p2_aes_encrypt_block(p2_builder* b, int nr, size_t xl, size_t ml, size_t sl, const p2_target* state, const p2_target* ek, p2_target* out) {
    (void)guarded([&] {
        aes::StateTarget mix = aes::state_mix_matrix(b->b);
        state_out(aes::encrypt_block_t(b->b, nr, xl, ml, sl, mix, state_in(state), words_in(ek, 4 * (nr + 1))), out);
    });
}
void p2_gcm_gctr(p2_builder* b, int nr, size_t xl, size_t ml, size_t sl, const p2_target* ek, const p2_target* icb, const p2_target* x, size_t len, p2_target* y) {
    (void)guarded([&] {
        aes::StateTarget mix = aes::state_mix_matrix(b->b);
        std::vector<aes::ByteTarget> xv(x, x + len);
        auto r = aes::gctr_target(b->b, nr, xl, ml, sl, mix, words_in(ek, 4 * (nr + 1)), block_in(icb), xv);
        for (size_t i = 0; i < len; i++) y[i] = r[i];
    });
}
void p2_gcm_right_shift_one(p2_builder* b, size_t shl, const p2_target* v, p2_target* out) {
    (void)guarded([&] {
        auto r = aes::right_shift_one_target(b->b, shl, block_in(v));
        for (int i = 0; i < 16; i++) out[i] = r[i];
    });
}
void p2_gcm_inc32(p2_builder* b, const p2_target* blk, p2_target* out) {
    (void)guarded([&] {
        auto r = aes::inc32_target(b->b, block_in(blk));
        for (int i = 0; i < 16; i++) out[i] = r[i];
    });
}
void p2_gcm_gf_2_128_mul(p2_builder* b, size_t xl, size_t shl, size_t brl, const p2_target* x, const p2_target* y, p2_target* out) {
    (void)guarded([&] {
        auto r = aes::gf_2_128_mul_target(b->b, xl, shl, brl, block_in(x), block_in(y));
        for (int i = 0; i < 16; i++) out[i] = r[i];
    });
}
int p2_gcm_ghash(p2_builder* b, size_t xl, size_t shl, size_t brl, const p2_target* h, const p2_target* x, size_t len, p2_target* out) {
    try {
        std::vector<aes::ByteTarget> xv(x, x + len);
        auto r = aes::ghash_target(b->b, xl, shl, brl, block_in(h), xv);
        for (int i = 0; i < 16; i++) out[i] = r[i];
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
int p2_aes_gcm_build(p2_builder* b, int nk, int nr, size_t L, int with_tag, p2_target* key, p2_target* nonce, p2_target* pt, p2_target* ct, p2_target* tag) {
    if (!((nk == 4 && nr == 10) || (nk == 6 && nr == 12) || (nk == 8 && nr == 14))) return set_error("unsupported (NK, NR)"), P2_ERR_INVALID;
    try {
        auto t = aes::AesGcmTarget::build(b->b, nk, nr, L, with_tag != 0);
        std::copy(t.key.begin(), t.key.end(), key);
        std::copy(t.nonce.begin(), t.nonce.end(), nonce);
        std::copy(t.pt.begin(), t.pt.end(), pt);
        std::copy(t.ct.begin(), t.ct.end(), ct);
        std::copy(t.tag.begin(), t.tag.end(), tag);
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}

// ---- Poseidon hashing / poseidon-cipher
int p2_builder_hash_n_to_m_no_pad(p2_builder* b, const p2_target* inputs, size_t n, p2_target* outputs, size_t m) {
    try {
        auto o = b->b.hash_n_to_m_no_pad(std::vector<Target>(inputs, inputs + n), m);
        std::copy(o.begin(), o.end(), outputs);
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
int p2_poseidon_cipher_build(p2_builder* b, size_t L, p2_target* ks, p2_target* m, p2_target* nonce, p2_target* ct) {
    try {
        auto t = pcipher::PoseidonEncryptTarget::build(b->b, L);
        for (int i = 0; i < 5; i++) {
            ks[i] = t.ks_x[i];
            ks[5 + i] = t.ks_u[i];
        }
        for (size_t i = 0; i < L; i++)
            for (int j = 0; j < 5; j++) m[5 * i + j] = t.m[i][j];
        nonce[0] = t.nonce[0];
        nonce[1] = t.nonce[1];
        for (size_t i = 0; i <= L; i++)
            for (int j = 0; j < 5; j++) ct[5 * i + j] = t.ct[i][j];
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
void p2_native_hash_n_to_m_no_pad(const uint64_t* in, size_t n, uint64_t* out, size_t m) {
    (void)guarded([&] {
        auto o = pcipher::hash_n_to_m_no_pad(std::vector<u64>(in, in + n), m);
        std::copy(o.begin(), o.end(), out);
    });
}
static pcipher::Fq fq_at(const uint64_t* p) { return pcipher::Fq{p[0], p[1], p[2], p[3], p[4]}; }
void p2_native_poseidon_encrypt(const uint64_t* ks, const uint64_t* msg, size_t n_msg, const uint64_t* nonce, uint64_t* ct) {
    (void)guarded([&] {
        std::vector<pcipher::Fq> m(n_msg);
        for (size_t i = 0; i < n_msg; i++) m[i] = fq_at(msg + 5 * i);
        auto c = pcipher::encrypt(fq_at(ks), fq_at(ks + 5), m, nonce);
        for (size_t i = 0; i < c.size(); i++) memcpy(ct + 5 * i, c[i].data(), 40);
    });
}
int p2_native_poseidon_decrypt(const uint64_t* ks, const uint64_t* ct, size_t n_ct, const uint64_t* nonce, size_t l, uint64_t* msg) {
    std::vector<pcipher::Fq> c(n_ct), m;
    for (size_t i = 0; i < n_ct; i++) c[i] = fq_at(ct + 5 * i);
    if (n_ct < 1 || l > n_ct - 1 || !pcipher::decrypt(fq_at(ks), fq_at(ks + 5), c, nonce, l, &m)) return set_error("poseidon decrypt: authentication failed"), P2_ERR_VERIFY;
    for (size_t i = 0; i < m.size(); i++) memcpy(msg + 5 * i, m[i].data(), 40);
    return P2_OK;
}

// ---- ecGFp5
static ecgfp5::Affine pt_at(const uint64_t* p) { return ecgfp5::Affine{fq_at(p), fq_at(p + 5)}; }
static void pt_put(const ecgfp5::Affine& a, uint64_t* out) {
    memcpy(out, a.x.data(), 40);
    memcpy(out + 5, a.u.data(), 40);
}
static ecgfp5::U320 sc_at(const uint64_t* k) {
    ecgfp5::U320 s;
    memcpy(s.w, k, 40);
    return s;
}
static void ptt_put(const ecgfp5::PointTarget& p, p2_target* out) {
    for (int i = 0; i < 5; i++) out[i] = p.x[i], out[5 + i] = p.u[i];
}
static ecgfp5::PointTarget ptt_at(const p2_target* p) {
    ecgfp5::PointTarget t;
    for (int i = 0; i < 5; i++) t.x[i] = p[i], t.u[i] = p[5 + i];
    return t;
}
static std::vector<BoolTarget> bits_at(const p2_target* bits) {
    std::vector<BoolTarget> v(ecgfp5::SCALAR_BITS);
    for (size_t i = 0; i < v.size(); i++) v[i] = BoolTarget{bits[i]};
    return v;
}
static int scalar_ok(const uint64_t* k) {
    if (!ecgfp5::u320_lt(sc_at(k), ecgfp5::GROUP_ORDER)) return set_error("scalar is not below the group order"), P2_ERR_INVALID;
    return P2_OK;
}
void p2_ecgfp5_group_order(uint64_t out[5]) { (void)guarded([&] { memcpy(out, ecgfp5::GROUP_ORDER.w, 40); }); }
void p2_ecgfp5_generator(uint64_t out[10]) { (void)guarded([&] { pt_put(ecgfp5::generator(), out); }); }
void p2_ecgfp5_mul(const uint64_t k[5], const uint64_t p[10], uint64_t out[10]) { (void)guarded([&] { pt_put(ecgfp5::scalar_mul(sc_at(k), pt_at(p)), out); }); }
void p2_ecgfp5_add(const uint64_t p[10], const uint64_t q[10], uint64_t out[10]) { (void)guarded([&] { pt_put(ecgfp5::point_add(pt_at(p), pt_at(q)), out); }); }
void p2_ecgfp5_neg(const uint64_t p[10], uint64_t out[10]) { (void)guarded([&] { pt_put(ecgfp5::point_neg(pt_at(p)), out); }); }
int p2_ecgfp5_is_in_subgroup(const uint64_t p[10]) { return ecgfp5::is_in_subgroup(pt_at(p)) ? 1 : 0; }
void p2_ecgfp5_compress(const uint64_t p[10], uint64_t w[5]) { (void)guarded([&] { memcpy(w, ecgfp5::compress_from_subgroup(pt_at(p)).data(), 40); }); }
int p2_ecgfp5_decompress(const uint64_t w[5], uint64_t out[10]) {
    ecgfp5::Affine a;
    if (!ecgfp5::decompress_into_subgroup(fq_at(w), &a)) return set_error("not the encoding of a group element"), P2_ERR_INVALID;
    pt_put(a, out);
    return P2_OK;
}
int p2_ecgfp5_random_scalar(uint64_t out[5]) {
    return guarded([&] {
        ecgfp5::Rng rng = ecgfp5::Rng::os();
        memcpy(out, ecgfp5::random_scalar(rng).w, 40);
    });
}
int p2_ecgfp5_random_point(uint64_t out[10]) {
    return guarded([&] {
        ecgfp5::Rng rng = ecgfp5::Rng::os();
        pt_put(ecgfp5::random_point(rng), out);
    });
}
int p2_ecgfp5_encode_binary(const uint32_t limbs[5], uint64_t out[10]) {
    return guarded([&] {
        ecgfp5::Rng rng = ecgfp5::Rng::os();
        pt_put(ecgfp5::encode_binary(limbs, rng), out);
    });
}
void p2_ecgfp5_random_scalar_seeded(uint64_t seed, uint64_t out[5]) {
    (void)guarded([&] {
        ecgfp5::Rng rng = ecgfp5::Rng::from_seed(seed);
        memcpy(out, ecgfp5::random_scalar(rng).w, 40);
    });
}
void p2_ecgfp5_random_point_seeded(uint64_t seed, uint64_t out[10]) {
    (void)guarded([&] {
        ecgfp5::Rng rng = ecgfp5::Rng::from_seed(seed);
        pt_put(ecgfp5::random_point(rng), out);
    });
}
void p2_ecgfp5_encode_binary_seeded(const uint32_t limbs[5], uint64_t seed, uint64_t out[10]) {
    (void)guarded([&] {
        ecgfp5::Rng rng = ecgfp5::Rng::from_seed(seed);
        pt_put(ecgfp5::encode_binary(limbs, rng), out);
    });
}
void p2_ecgfp5_decode_binary(const uint64_t p[10], uint32_t limbs[5]) { (void)guarded([&] { ecgfp5::decode_binary(pt_at(p), limbs); }); }
int p2_elgamal_encrypt(const uint64_t pk[10], const uint64_t nonce[5], const uint64_t msg[10], uint64_t c0[10], uint64_t c1[10]) {
    if (scalar_ok(nonce)) return P2_ERR_INVALID;
    ecgfp5::Affine a, b;
    ecgfp5::elgamal_encrypt(pt_at(pk), sc_at(nonce), pt_at(msg), &a, &b);
    pt_put(a, c0);
    pt_put(b, c1);
    return P2_OK;
}
int p2_elgamal_decrypt(const uint64_t sk[5], const uint64_t c0[10], const uint64_t c1[10], uint64_t msg[10]) {
    if (scalar_ok(sk)) return P2_ERR_INVALID;
    pt_put(ecgfp5::elgamal_decrypt(sc_at(sk), pt_at(c0), pt_at(c1)), msg);
    return P2_OK;
}
int p2_hashed_elgamal_encrypt(const uint64_t pk[10], const uint64_t nonce[5], const uint64_t msg[5], uint64_t c0[10], uint64_t ct[5]) {
    if (scalar_ok(nonce)) return P2_ERR_INVALID;
    ecgfp5::Affine a;
    ecgfp5::hashed_elgamal_encrypt(pt_at(pk), sc_at(nonce), msg, &a, ct);
    pt_put(a, c0);
    return P2_OK;
}
int p2_hashed_elgamal_decrypt(const uint64_t sk[5], const uint64_t c0[10], const uint64_t ct[5], uint64_t msg[5]) {
    if (scalar_ok(sk)) return P2_ERR_INVALID;
    ecgfp5::hashed_elgamal_decrypt(sc_at(sk), pt_at(c0), ct, msg);
    return P2_OK;
}
void p2_builder_add_virtual_point_target(p2_builder* b, p2_target out[10]) { (void)guarded([&] { ptt_put(ecgfp5::add_virtual_point_target(b->b), out); }); }
void p2_builder_constant_point(p2_builder* b, const uint64_t p[10], p2_target out[10]) { (void)guarded([&] { ptt_put(ecgfp5::constant_point(b->b, pt_at(p)), out); }); }
void p2_builder_add_virtual_biguint320_target(p2_builder* b, p2_target bits[320]) {
    (void)guarded([&] {
        auto v = ecgfp5::add_virtual_biguint320_target(b->b);
        for (size_t i = 0; i < v.size(); i++) bits[i] = v[i].target;
    });
}
int p2_builder_multiply_point(p2_builder* b, const p2_target bits[320], const p2_target p[10], p2_target out[10]) {
    try {
        ptt_put(ecgfp5::multiply_point(b->b, bits_at(bits), ptt_at(p)), out);
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
void p2_builder_add_point(p2_builder* b, const p2_target p[10], const p2_target q[10], p2_target out[10]) { (void)guarded([&] {    ptt_put(ecgfp5::add_point(b->b, ptt_at(p), ptt_at(q)), out);}); }
int p2_builder_public_key(p2_builder* b, const p2_target sk_bits[320], p2_target pk[10]) {
    try {
        ptt_put(ecgfp5::public_key_target(b->b, bits_at(sk_bits)), pk);
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
int p2_builder_elgamal_encrypt(p2_builder* b, const p2_target pk[10], const p2_target nonce_bits[320], const p2_target msg[10], p2_target c0[10],
                               p2_target c1[10]) {
    try {
        ecgfp5::PointTarget a, c;
        ecgfp5::elgamal_encrypt_target(b->b, ptt_at(pk), bits_at(nonce_bits), ptt_at(msg), &a, &c);
        ptt_put(a, c0);
        ptt_put(c, c1);
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
int p2_builder_hashed_elgamal_encrypt(p2_builder* b, const p2_target pk[10], const p2_target nonce_bits[320], const p2_target msg[5],
                                      p2_target c0[10], p2_target ct[5]) {
    try {
        ecgfp5::PointTarget a;
        ecgfp5::hashed_elgamal_encrypt_target(b->b, ptt_at(pk), bits_at(nonce_bits), msg, &a, ct);
        ptt_put(a, c0);
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}

// ---- self-test
int p2_selftest_host(uint64_t seed, size_t n_reductions, size_t n_permutations) {
    u64 x = seed | 1;
    auto rnd = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    size_t bad = 0;
    for (size_t i = 0; i < n_reductions; i++) {
        u64 hi = rnd(), lo = rnd();
        switch (i & 15) {
            case 0: hi &= gl::EPS; break;
            case 1: lo &= gl::EPS; break;
            case 2: hi |= ~gl::EPS; break;
            case 3: lo |= ~gl::EPS; break;
            case 4: hi = 0; break;
            case 5: lo = 0; break;
            case 6: hi = ~0ull; lo = ~0ull - (x & 3); break;
            case 7: hi &= ~gl::EPS; lo &= gl::EPS; break;
            default: break;
        }
        unsigned __int128 v = ((unsigned __int128)hi << 64) | lo;
        u64 want = (u64)(v % gl::P);
        if (glf::canon(glf::red128(hi, lo)) != want) bad++;
        if (gl::reduce128(hi, lo) != want) bad++;
        glf::Acc a;
        a.init();
        a.fma(hi, lo);
        a.fma(lo, lo);
        unsigned __int128 w = ((unsigned __int128)hi * lo) % gl::P + ((unsigned __int128)lo * lo) % gl::P;
        if (glf::canon(a.reduce()) != (u64)(w % gl::P)) bad++;
    }
    for (size_t i = 0; i < n_permutations; i++) {
        u64 s0[12], s1[12];
        for (int k = 0; k < 12; k++) s0[k] = s1[k] = (i & 3) == 3 ? rnd() : rnd() % gl::P;  // canonical or arbitrary words
        for (int k = 0; k < 12; k++) s0[k] %= gl::P;
        gl::poseidon(s0);
        glf::poseidon(s1);
        for (int k = 0; k < 12; k++) bad += s0[k] != s1[k];
    }
    return (int)std::min<size_t>(bad, 0x7FFFFFFF);
}

// ---- native cipher
uint8_t p2_native_gf_2_8_mul(uint8_t a, uint8_t b) { return aes::gf_2_8_mul(a, b); }
void p2_native_aes_key_expansion(const uint8_t* key, int nk, int nr, uint8_t* out) {
    (void)guarded([&] {
        auto w = aes::key_expansion(nk, nr, key);
        for (size_t i = 0; i < w.size(); i++) memcpy(out + 4 * i, w[i].data(), 4);
    });
}
void p2_native_aes_encrypt_block(const uint8_t* key, int nk, int nr, const uint8_t* in, uint8_t* out) {
    (void)guarded([&] {
        auto w = aes::key_expansion(nk, nr, key);
        auto s = aes::flatten_state(aes::encrypt_block(nr, in, w));
        memcpy(out, s.data(), 16);
    });
}
void p2_native_gf_2_128_mul(const uint8_t* x, const uint8_t* y, uint8_t* out) {
    (void)guarded([&] {
        auto r = aes::gf_2_128_mul(x, y);
        memcpy(out, r.data(), 16);
    });
}
void p2_native_ghash(const uint8_t* h, const uint8_t* x, size_t len, uint8_t* out) {
    (void)guarded([&] {
        auto r = aes::ghash(h, x, len);
        memcpy(out, r.data(), 16);
    });
}
void p2_native_gctr(const uint8_t* key, int nk, int nr, const uint8_t* icb, const uint8_t* x, size_t len, uint8_t* y) {
    (void)guarded([&] {
        auto w = aes::key_expansion(nk, nr, key);
        auto r = aes::gctr(nr, w, icb, x, len);
        if (len) memcpy(y, r.data(), len);
    });
}
void p2_native_aes_gcm_encrypt(const uint8_t* key, int nk, int nr, const uint8_t* nonce, const uint8_t* pt, size_t len, uint8_t* ct, uint8_t* tag) { (void)guarded([&] {    aes::gcm_encrypt(nk, nr, key, nonce, pt, len, ct, tag);}); }

// ---- info / verify
int p2_blob_info(const uint8_t* blob, size_t len, p2_circuit_info* out) {
    try {
        Circuit c = deserialize(blob, len);
        fill_info(c, out);
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
int p2_witness_schedule_check(const uint8_t* blob, size_t len, uint32_t fuse, uint32_t out[4]) {
    try {
        Circuit c = deserialize(blob, len);
        WitnessSchedule s = schedule_witness(c, fuse);
        const size_t M = c.ops.size(), n = (size_t)1 << c.degree_bits;
        const u32 R = c.cfg.num_routed_wires;
        auto same = [](const Op& x, const Op& y) { return x.kind == y.kind && x.out == y.out && x.a == y.a && x.b == y.b && x.c == y.c && x.aux == y.aux && x.k0 == y.k0 && x.k1 == y.k1; };
        auto slots_of = [&](const Op& o, u32* d, int& first_out) {
            int nd = 0;
            if (o.kind == OP_ARITH) d[nd++] = o.a, d[nd++] = o.b, d[nd++] = o.c;
            else if (o.kind == OP_LOOKUP) d[nd++] = o.a;
            else if (o.kind == OP_EQ || o.kind == OP_EQINV) d[nd++] = o.a, d[nd++] = o.b;
            else if (o.kind == OP_POSEIDON) {
                for (u32 k = 0; k < 12; k++) d[nd++] = (u32)c.wire_slot[(size_t)(PG_IN + k) * n + o.a];
                d[nd++] = (u32)c.wire_slot[(size_t)PG_SWAP * n + o.a];
            }
            first_out = nd;
            if (o.kind == OP_POSEIDON) {
                for (u32 col = PG_OUT; col < R; col++)
                    if (col != PG_SWAP) d[nd++] = (u32)c.wire_slot[(size_t)col * n + o.a];
            } else {
                d[nd++] = o.out;
            }
            return nd;
        };
        if (s.ops.size() != M || s.levels.empty()) return set_error("witness schedule: shape"), P2_ERR_INVALID;
        // every op belongs to exactly one unit (a chain or a single) of exactly one level, and the levels tile the op array
        const u32 NONE = ~0u;
        std::vector<u32> unit_of(M, NONE), level_of(M, NONE);
        u32 cursor = 0, unit = 0, chain_cursor = 0;
        for (size_t l = 0; l < s.levels.size(); l++) {
            const WLevel& L = s.levels[l];
            if (L.chain_begin != chain_cursor || (size_t)L.chain_begin + L.chain_count > s.chains.size()) return set_error("witness schedule: chain ranges"), P2_ERR_INVALID;
            for (u32 ci = 0; ci < L.chain_count; ci++) {
                const WChain& ch = s.chains[L.chain_begin + ci];
                if (ch.start != cursor || ch.count < 2 || ch.count > std::max<u32>(fuse, 1) || (size_t)ch.start + ch.count > M) return set_error("witness schedule: chain shape"), P2_ERR_INVALID;
                for (u32 k = 0; k < ch.count; k++) {
                    const u32 kind = s.ops[ch.start + k].kind;
                    if (kind != OP_ARITH && kind != OP_CONST && kind != OP_EQ) return set_error("witness schedule: a chain holds an op the chain executor does not run"), P2_ERR_INVALID;
                    unit_of[ch.start + k] = unit, level_of[ch.start + k] = (u32)l;
                }
                unit++;
                cursor += ch.count;
            }
            chain_cursor += L.chain_count;
            if (L.single_begin != cursor || L.single_end < L.single_begin || L.single_end > M) return set_error("witness schedule: single ranges"), P2_ERR_INVALID;
            for (u32 k = L.single_begin; k < L.single_end; k++) unit_of[k] = unit++, level_of[k] = (u32)l;
            cursor = L.single_end;
        }
        if (cursor != M || chain_cursor != s.chains.size()) return set_error("witness schedule: the levels do not tile the program"), P2_ERR_INVALID;
        // first producer of every slot, in the builder's order and in the scheduled order
        std::vector<int64_t> first_orig(c.num_slots, -1), first_sched(c.num_slots, -1);
        u32 d[96];
        int fo;
        for (size_t i = 0; i < M; i++) {
            int nd = slots_of(c.ops[i], d, fo);
            for (int j = fo; j < nd; j++)
                if (first_orig[d[j]] < 0) first_orig[d[j]] = (int64_t)i;
        }
        for (size_t i = 0; i < M; i++) {
            int nd = slots_of(s.ops[i], d, fo);
            for (int j = fo; j < nd; j++)
                if (first_sched[d[j]] < 0) first_sched[d[j]] = (int64_t)i;
        }
        for (u32 sl = 0; sl < c.num_slots; sl++) {
            if ((first_orig[sl] < 0) != (first_sched[sl] < 0)) return set_error("witness schedule: a slot lost or gained a producer"), P2_ERR_INVALID;
            if (first_orig[sl] >= 0 && !same(c.ops[first_orig[sl]], s.ops[first_sched[sl]])) return set_error("witness schedule: a slot changed its first producer"), P2_ERR_INVALID;
        }
        for (size_t i = 0; i < M; i++) {
            int nd = slots_of(s.ops[i], d, fo);
            for (int j = 0; j < nd; j++) {
                const int64_t p = first_sched[d[j]];
                if (p < 0 || p == (int64_t)i) continue;   // a user input, or this op is the producer
                const bool earlier_level = level_of[p] < level_of[i];
                const bool same_unit_before = unit_of[p] == unit_of[i] && p < (int64_t)i;
                if (!earlier_level && !same_unit_before) return set_error("witness schedule: an operand is not ready when its op runs"), P2_ERR_INVALID;
            }
        }
        // multiset of ops preserved: compare sorted fingerprints
        auto fp = [](const Op& o) { return ((u64)o.kind * 0x9E3779B97F4A7C15ull) ^ ((u64)o.out << 32 | o.a) ^ (((u64)o.b << 32 | o.c) * 0xBF58476D1CE4E5B9ull) ^ (o.k0 * 3 + o.k1 * 5 + o.aux); };
        std::vector<u64> fa(M), fb(M);
        for (size_t i = 0; i < M; i++) fa[i] = fp(c.ops[i]), fb[i] = fp(s.ops[i]);
        std::sort(fa.begin(), fa.end());
        std::sort(fb.begin(), fb.end());
        if (fa != fb) return set_error("witness schedule: ops changed"), P2_ERR_INVALID;
        out[0] = (u32)s.levels.size();
        out[1] = (u32)s.chains.size();
        out[2] = s.max_chain;
        out[3] = (u32)s.fused_ops;
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
int p2_verify(const uint8_t* blob, size_t blob_len, const uint64_t* vd, size_t vd_len, const uint8_t* proof, size_t proof_len) {
    try {
        Circuit c = deserialize(blob, blob_len);
        size_t cap_n = (size_t)1 << c.cfg.cap_height;
        if (vd_len != 4 * cap_n + 4) return set_error("verifier_data must be cap || circuit_digest"), P2_ERR_INVALID;
        VerifierData v;
        v.constants_sigmas_cap.resize(cap_n);
        for (size_t i = 0; i < cap_n; i++) memcpy(v.constants_sigmas_cap[i].e, vd + 4 * i, 32);
        memcpy(v.circuit_digest.e, vd + 4 * cap_n, 32);
        std::string err = verify_proof(c, v, proof, proof_len);
        if (!err.empty()) return set_error(err), P2_ERR_VERIFY;
        return P2_OK;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    }
}
}

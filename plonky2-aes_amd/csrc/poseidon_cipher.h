// Poseidon-sponge encryption over Fq = GF(p^5) and its circuit, one function per reference function.
// Mirrors poseidon-cipher/src/lib.rs (encrypt :41-73, decrypt :75-111, hash_state :113-118) and
// poseidon-cipher/src/circuit.rs (PoseidonEncryptTarget::build :49-89, hash_state_target :121-125).
// pod2's PointTarget / OEFTarget<5> are not in the reference tree: a point is taken as its two Fq coordinates (x, u)
// = 10 virtual targets with no curve-membership constraint, and nnf_add is the component-wise addition of an optimal
// extension field.  Key generation (new_key / expanded_key, lib.rs:31-39) is a scalar multiplication: see ecgfp5.h.
#pragma once
#include <array>

#include "builder.h"

namespace p2 {
namespace pcipher {

typedef std::array<u64, 5> Fq;

inline std::vector<u64> hash_n_to_m_no_pad(const std::vector<u64>& in, size_t m) {
    u64 st[12] = {0};
    for (size_t off = 0; off < in.size(); off += 8) {
        for (size_t i = 0; i < 8 && off + i < in.size(); i++) st[i] = in[off + i];
        gl::poseidon(st);
    }
    std::vector<u64> out;
    for (;;) {
        for (int i = 0; i < 8; i++) {
            out.push_back(st[i]);
            if (out.size() == m) return out;
        }
        gl::poseidon(st);
    }
}
inline void hash_state(Fq s[4]) {
    std::vector<u64> e(20);
    for (int i = 0; i < 20; i++) e[i] = s[i / 5][i % 5];
    auto h = hash_n_to_m_no_pad(e, 20);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 5; j++) s[i][j] = h[j + 5 * i];
}
inline Fq fq_add(const Fq& a, const Fq& b) {
    Fq r;
    for (int i = 0; i < 5; i++) r[i] = gl::add(a[i], b[i]);
    return r;
}
inline Fq fq_sub(const Fq& a, const Fq& b) {
    Fq r;
    for (int i = 0; i < 5; i++) r[i] = gl::sub(a[i], b[i]);
    return r;
}
// lib.rs:41  ct has padded_len(msg) + 1 elements
inline std::vector<Fq> encrypt(const Fq& ks_x, const Fq& ks_u, const std::vector<Fq>& msg, const u64 nonce[2]) {
    std::vector<Fq> m(msg);
    m.resize((m.size() + 2) / 3 * 3, Fq{0, 0, 0, 0, 0});
    Fq s[4] = {Fq{0, 0, 0, 0, 0}, ks_x, ks_u, Fq{nonce[0], nonce[1], (u64)msg.size(), 0, 0}};
    std::vector<Fq> ct(m.size() + 1);
    for (size_t i = 0; i < m.size() / 3; i++) {
        hash_state(s);
        for (int k = 0; k < 3; k++) {
            s[1 + k] = fq_add(s[1 + k], m[3 * i + k]);
            ct[3 * i + k] = s[1 + k];
        }
    }
    hash_state(s);
    ct[m.size()] = s[1];
    return ct;
}
// lib.rs:75  returns false where the reference's asserts would fail
inline bool decrypt(const Fq& ks_x, const Fq& ks_u, const std::vector<Fq>& ct, const u64 nonce[2], size_t l, std::vector<Fq>* out) {
    Fq zero{0, 0, 0, 0, 0};
    Fq s[4] = {zero, ks_x, ks_u, Fq{nonce[0], nonce[1], (u64)l, 0, 0}};
    std::vector<Fq> m(ct.size() - 1);
    for (size_t i = 0; i < ct.size() / 3; i++) {
        hash_state(s);
        for (int k = 0; k < 3; k++) {
            m[3 * i + k] = fq_sub(ct[3 * i + k], s[1 + k]);
            s[1 + k] = ct[3 * i + k];
        }
    }
    if (l > 3) {
        if (l % 3 == 2 && m[m.size() - 1] != zero) return false;
        if (l % 3 == 1 && (m[m.size() - 1] != zero || m[m.size() - 2] != zero)) return false;
    }
    hash_state(s);
    if (ct[ct.size() - 1] != s[1]) return false;
    out->assign(m.begin(), m.begin() + l);
    return true;
}

typedef std::array<Target, 5> FqT;

inline void hash_state_target(CircuitBuilder& b, FqT s[4]) {  // circuit.rs:121
    std::vector<Target> e(20);
    for (int i = 0; i < 20; i++) e[i] = s[i / 5][i % 5];
    auto h = b.hash_n_to_m_no_pad(e, 20);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 5; j++) s[i][j] = h[j + 5 * i];
}
inline FqT nnf_add(CircuitBuilder& b, const FqT& x, const FqT& y) {
    FqT r;
    for (int i = 0; i < 5; i++) r[i] = b.add(x[i], y[i]);
    return r;
}
struct PoseidonEncryptTarget {  // circuit.rs:35
    size_t L;
    FqT ks_x, ks_u;
    std::vector<FqT> m, ct;
    Target nonce[2];
    static PoseidonEncryptTarget build(CircuitBuilder& b, size_t L) {
        if (L % 3 != 0) throw std::runtime_error("L must be a multiple of 3");
        PoseidonEncryptTarget t;
        t.L = L;
        Target z = b.constant(0);
        for (auto& x : t.ks_x) x = b.add_virtual_target();
        for (auto& x : t.ks_u) x = b.add_virtual_target();
        t.m.resize(L);
        for (auto& e : t.m)
            for (auto& x : e) x = b.add_virtual_target();
        t.nonce[0] = b.add_virtual_target();
        t.nonce[1] = b.add_virtual_target();
        FqT fzero = {z, z, z, z, z};
        t.ct.assign(L + 1, fzero);
        FqT n_l = {t.nonce[0], t.nonce[1], b.constant((u64)L), z, z};
        FqT s[4] = {fzero, t.ks_x, t.ks_u, n_l};
        for (size_t i = 0; i < L / 3; i++) {
            hash_state_target(b, s);
            for (int k = 0; k < 3; k++) {
                s[1 + k] = nnf_add(b, s[1 + k], t.m[3 * i + k]);
                t.ct[3 * i + k] = s[1 + k];
            }
        }
        hash_state_target(b, s);
        t.ct[L] = s[1];
        return t;
    }
};

}  // namespace pcipher
}  // namespace p2

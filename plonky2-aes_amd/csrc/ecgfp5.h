// ecGFp5: native group arithmetic, the ElGamal schemes of the reference, and their circuits.
//
// Mirrors ecgfp5/src/lib.rs (ECGFP5SecretKey :23-42, encode_binary :48-76, decode_binary :80-99), elgamal.rs
// (:11-22), hashed_elgamal.rs (:19-32), circuit.rs (public_key :35-38), elgamal/circuit.rs (:28-38) and
// hashed_elgamal/circuit.rs (:33-47).  The curve itself lives in the `pod2` crate, which is not in the reference tree
// (SURVEY.md 8c): what is restated here is the published curve (Pornin, "EcGFp5", 2022) --
//     K = GF(p)[z]/(z^5 - 3),  E: y^2 = x(x^2 + 2x + 263z),  |E| = 2n,
//     group = N + E[n] with neutral N = (0,0), elements written (x, u = x/y), generator u = 1/4 --
// checked numerically (n prime, n*G = N, formulas against chord-and-tangent arithmetic: oracle/oracle_ecgfp5.py).
// pod2's in-circuit EC gates are unknown, so multiply_point / add_point are built from plonky2 arithmetic gates only:
// GF(p^5) is an extension of the native field, so a product is 25 base multiplications and no range checks are needed.
// Conventions that pod2 fixes and no reference fixture pins (compress = u, sampling) are marked UNPINNED.
#pragma once
#include <array>

#include "builder.h"
#include "os_random.h"
#include "poseidon_cipher.h"

namespace p2 {
namespace ecgfp5 {

typedef std::array<u64, 5> Fq;
static const u64 B1 = 263;  // b = B1 * z,  a = 2

// ------------------------------------------------------------------------------------------ GF(p^5)
inline Fq fq_zero() { return Fq{0, 0, 0, 0, 0}; }
inline Fq fq_one() { return Fq{1, 0, 0, 0, 0}; }
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
inline Fq fq_neg(const Fq& a) {
    Fq r;
    for (int i = 0; i < 5; i++) r[i] = gl::neg(a[i]);
    return r;
}
inline Fq fq_small(const Fq& a, u64 k) {
    Fq r;
    for (int i = 0; i < 5; i++) r[i] = gl::mul(a[i], k);
    return r;
}
inline Fq fq_mul(const Fq& a, const Fq& b) {
    u64 c[9] = {0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) c[i + j] = gl::add(c[i + j], gl::mul(a[i], b[j]));
    Fq r;
    for (int k = 0; k < 5; k++) r[k] = k + 5 < 9 ? gl::add(c[k], gl::mul(3, c[k + 5])) : c[k];
    return r;
}
inline Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }
inline Fq fq_mul_k1(const Fq& a, u64 k) {  // a * (k z)
    return Fq{gl::mul(gl::mul(3, k), a[4]), gl::mul(k, a[0]), gl::mul(k, a[1]), gl::mul(k, a[2]), gl::mul(k, a[3])};
}
inline u64 frob_gamma() { return gl::pow(3, (gl::P - 1) / 5); }  // z^p = gamma z
inline Fq fq_frob(const Fq& a, int k) {                           // a^(p^k)
    u64 g = gl::pow(frob_gamma(), (u64)k), s = 1;
    Fq r;
    for (int i = 0; i < 5; i++) {
        r[i] = gl::mul(a[i], s);
        s = gl::mul(s, g);
    }
    return r;
}
// a^(p + p^2 + p^3 + p^4); a times it is the norm, an element of GF(p)
inline Fq fq_conj_product(const Fq& a) {
    Fq t = fq_mul(fq_frob(a, 1), fq_frob(a, 2));
    return fq_mul(t, fq_frob(t, 2));
}
inline u64 fq_norm(const Fq& a) { return fq_mul(a, fq_conj_product(a))[0]; }
inline Fq fq_inv(const Fq& a) {  // 0 -> 0
    Fq t = fq_conj_product(a);
    u64 nrm = fq_mul(a, t)[0];
    return fq_small(t, nrm ? gl::inv(nrm) : 0);
}
inline bool fq_is_square(const Fq& a) {
    u64 nrm = fq_norm(a);
    return nrm == 0 || gl::pow(nrm, (gl::P - 1) / 2) == 1;
}
// Tonelli-Shanks in GF(p); p - 1 = 2^32 * odd and POW2_GEN generates the 2-Sylow subgroup
inline bool gl_sqrt(u64 a, u64* out) {
    if (a == 0) return *out = 0, true;
    if (gl::pow(a, (gl::P - 1) / 2) != 1) return false;
    const u64 q = (gl::P - 1) >> 32;
    u64 c = gl::POW2_GEN, x = gl::pow(a, (q + 1) / 2), t = gl::pow(a, q);
    int m = 32;
    while (t != 1) {
        int i = 0;
        for (u64 t2 = t; t2 != 1; i++) t2 = gl::sqr(t2);
        u64 b = gl::exp_pow2(c, m - i - 1);
        x = gl::mul(x, b);
        c = gl::sqr(b);
        t = gl::mul(t, c);
        m = i;
    }
    return *out = x, true;
}
// sqrt(a) = sqrt(Norm(a)) / a^((r-1)/2),  (r-1)/2 = ((p+1)/2) * p * (1 + p^2)
inline bool fq_sqrt(const Fq& a, Fq* out) {
    if (a == fq_zero()) return *out = a, true;
    Fq y = fq_one(), base = a;
    for (u64 e = (gl::P + 1) / 2; e; e >>= 1) {
        if (e & 1) y = fq_mul(y, base);
        base = fq_sqr(base);
    }
    Fq v = fq_frob(fq_mul(y, fq_frob(y, 2)), 1);
    u64 s;
    if (!gl_sqrt(fq_mul(a, fq_sqr(v))[0], &s)) return false;
    *out = fq_small(fq_inv(v), s);
    return true;
}

// ------------------------------------------------------------------------------------------ scalars
struct U320 {
    u64 w[5];  // little-endian limbs
    bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
};
static const U320 GROUP_ORDER = {{0xe80fd996948bffe1ull, 0xe8885c39d724a09cull, 0x7fffffe6cfb80639ull, 0x7ffffff100000016ull, 0x7ffffffd80000007ull}};
inline bool u320_lt(const U320& a, const U320& b) {
    for (int i = 4; i >= 0; i--)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    return false;
}
// Randomness for keys, nonces and encode_binary padding.  The default source is the operating system's CSPRNG, as in the
// reference (OsRng: ecgfp5/src/lib.rs:35,64; poseidon-cipher/src/lib.rs:32,37).  The seeded form (SplitMix64) exists for
// tests and benchmarks that need reproducible inputs: 64 bits of a non-cryptographic generator, never a real key.
struct Rng {
    bool seeded;
    u64 s;
    static Rng os() { return Rng{false, 0}; }
    static Rng from_seed(u64 seed) { return Rng{true, seed}; }
    u64 next() {
        if (!seeded) {
            u64 v;
            os_random_bytes(&v, 8);
            return v;
        }
        u64 z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    u64 below(u64 bound) {  // uniform in [0, bound)
        for (;;) {
            u64 v = next();
            if (v < UINT64_MAX - UINT64_MAX % bound) return v % bound;
        }
    }
};
// gen_biguint_below(&GROUP_ORDER): rejection sampling of 319-bit values
inline U320 random_scalar(Rng& rng) {
    for (;;) {
        U320 k;
        for (int i = 0; i < 5; i++) k.w[i] = rng.next();
        k.w[4] &= (1ull << 63) - 1;  // 319 bits
        if (u320_lt(k, GROUP_ORDER)) return k;
    }
}

// ------------------------------------------------------------------------------------------ the group
struct Affine {
    Fq x, u;
    bool operator==(const Affine& o) const { return x == o.x && u == o.u; }
};
struct Point {  // x = X/Z, u = U/T
    Fq X, Z, U, T;
    static Point neutral() { return Point{fq_zero(), fq_one(), fq_zero(), fq_one()}; }
    static Point from_affine(const Affine& a) { return Point{a.x, fq_one(), a.u, fq_one()}; }
    Affine affine() const { return Affine{fq_mul(X, fq_inv(Z)), fq_mul(U, fq_inv(T))}; }
    Point neg() const { return Point{X, Z, fq_neg(U), T}; }
    // complete (unified) addition, 10 multiplications
    Point add(const Point& o) const {
        Fq t1 = fq_mul(X, o.X), t2 = fq_mul(Z, o.Z), t3 = fq_mul(U, o.U), t4 = fq_mul(T, o.T);
        Fq t5 = fq_sub(fq_sub(fq_mul(fq_add(X, Z), fq_add(o.X, o.Z)), t1), t2);
        Fq t6 = fq_sub(fq_sub(fq_mul(fq_add(U, T), fq_add(o.U, o.T)), t3), t4);
        Fq t7 = fq_add(t1, fq_mul_k1(t2, B1));
        Fq t8 = fq_mul(t4, t7);
        Fq t9 = fq_mul(t3, fq_add(fq_mul_k1(t5, 2 * B1), fq_small(t7, 2)));
        Fq t10 = fq_mul(fq_add(t4, fq_small(t3, 2)), fq_add(t5, t7));
        return Point{fq_mul_k1(fq_sub(t10, t8), B1), fq_sub(t8, t9), fq_mul(t6, fq_sub(fq_mul_k1(t2, B1), t1)), fq_add(t8, t9)};
    }
    Point dbl() const { return add(*this); }
    Point mul(const U320& k) const {
        Point acc = neutral();
        for (int i = 319; i >= 0; i--) {
            acc = acc.dbl();
            if (k.bit(i)) acc = acc.add(*this);
        }
        return acc;
    }
    bool equals(const Point& o) const { return fq_mul(X, o.Z) == fq_mul(o.X, Z) && fq_mul(U, o.T) == fq_mul(o.U, T); }
};
// on the curve, and in the group: x = u^2 (x^2 + a x + b) with x a non-square (or the neutral)
inline bool on_curve(const Affine& a) {
    Fq w = fq_add(fq_add(fq_sqr(a.x), fq_small(a.x, 2)), Fq{0, B1, 0, 0, 0});
    return fq_mul(fq_sqr(a.u), w) == a.x;
}
inline bool is_in_subgroup(const Affine& a) {
    if (a.x == fq_zero() && a.u == fq_zero()) return true;
    if (a.x == fq_zero() || a.u == fq_zero()) return false;
    return on_curve(a) && !fq_is_square(a.x);
}
// UNPINNED: a group element compresses to u (pod2's choice is not visible from the reference)
inline Fq compress_from_subgroup(const Affine& a) { return a.u; }
inline bool decompress_into_subgroup(const Fq& u, Affine* out) {
    if (u == fq_zero()) return *out = Affine{fq_zero(), fq_zero()}, true;
    Fq bc = fq_sub(Fq{2, 0, 0, 0, 0}, fq_inv(fq_sqr(u)));  // x^2 + bc x + b = 0
    Fq r;
    if (!fq_sqrt(fq_sub(fq_sqr(bc), Fq{0, 4 * B1, 0, 0, 0}), &r)) return false;
    const u64 half = (gl::P + 1) / 2;
    for (int s = 0; s < 2; s++) {
        Affine cand{fq_small(fq_sub(s ? fq_neg(r) : r, bc), half), u};
        if (is_in_subgroup(cand)) return *out = cand, true;
    }
    return false;
}
inline const Affine& generator() {
    static const Affine g = [] {
        Affine a;
        if (!decompress_into_subgroup(Fq{gl::inv(4), 0, 0, 0, 0}, &a)) throw std::runtime_error("ecgfp5 generator");
        return a;
    }();
    return g;
}
inline Affine scalar_mul(const U320& k, const Affine& p) { return Point::from_affine(p).mul(k).affine(); }
inline Affine point_add(const Affine& p, const Affine& q) { return Point::from_affine(p).add(Point::from_affine(q)).affine(); }
inline Affine point_neg(const Affine& p) { return Affine{p.x, fq_neg(p.u)}; }
inline Affine public_key(const U320& sk) { return scalar_mul(sk, generator()); }                  // lib.rs:39
inline Affine random_point(Rng& rng) { return scalar_mul(random_scalar(rng), generator()); }  // new_rand_from_subgroup, UNPINNED
inline std::vector<u64> as_fields(const Affine& a) {
    std::vector<u64> f(a.x.begin(), a.x.end());
    f.insert(f.end(), a.u.begin(), a.u.end());
    return f;
}

// lib.rs:48  160 message bits, 32 in the low half of every limb, the high halves random until the element decodes
inline Affine encode_binary(const u32 limbs[5], Rng& rng) {
    for (;;) {
        Fq w;
        for (int i = 0; i < 5; i++) {
            u64 r = rng.below(gl::P - limbs[i]);
            w[i] = limbs[i] + ((r >> 32) << 32);
        }
        Affine a;
        if (decompress_into_subgroup(w, &a)) return a;
    }
}
inline void decode_binary(const Affine& p, u32 limbs[5]) {  // lib.rs:80
    Fq w = compress_from_subgroup(p);
    for (int i = 0; i < 5; i++) limbs[i] = (u32)w[i];
}
// elgamal.rs:11 / :19
inline void elgamal_encrypt(const Affine& pk, const U320& nonce, const Affine& msg, Affine* c0, Affine* c1) {
    *c0 = scalar_mul(nonce, generator());
    *c1 = point_add(msg, scalar_mul(nonce, pk));
}
inline Affine elgamal_decrypt(const U320& sk, const Affine& c0, const Affine& c1) { return point_add(c1, point_neg(scalar_mul(sk, c0))); }
// hashed_elgamal.rs:19 / :28
inline void hashed_elgamal_encrypt(const Affine& pk, const U320& nonce, const u64 msg[5], Affine* c0, u64 ct[5]) {
    *c0 = scalar_mul(nonce, generator());
    auto h = pcipher::hash_n_to_m_no_pad(as_fields(scalar_mul(nonce, pk)), 5);
    for (int i = 0; i < 5; i++) ct[i] = gl::add(msg[i], h[i]);
}
inline void hashed_elgamal_decrypt(const U320& sk, const Affine& c0, const u64 ct[5], u64 msg[5]) {
    auto h = pcipher::hash_n_to_m_no_pad(as_fields(scalar_mul(sk, c0)), 5);
    for (int i = 0; i < 5; i++) msg[i] = gl::sub(ct[i], h[i]);
}

// ------------------------------------------------------------------------------------------ circuits
typedef std::array<Target, 5> FqT;
struct PointTarget {
    FqT x, u;
};
struct ProjTarget {
    FqT X, Z, U, T;
};
static const u64 NEG1 = gl::P - 1;

inline FqT fq_const(CircuitBuilder& b, const Fq& a) {
    FqT r;
    for (int i = 0; i < 5; i++) r[i] = b.constant(a[i]);
    return r;
}
// pc * x * y + ac * addend, with optional per-coefficient scalings sx, sy of the factors (Frobenius): 25 gates' ops
inline FqT fq_mul(CircuitBuilder& b, const FqT& x, const FqT& y, u64 pc = 1, const FqT* addend = nullptr, u64 ac = 1, const u64* sx = nullptr,
                  const u64* sy = nullptr) {
    FqT r;
    for (int k = 0; k < 5; k++) {
        bool have = false;
        Target acc = 0;
        for (int i = 0; i < 5; i++) {
            int j = (k - i + 5) % 5;
            u64 c = i + j >= 5 ? gl::mul(pc, 3) : pc;
            if (sx) c = gl::mul(c, sx[i]);
            if (sy) c = gl::mul(c, sy[j]);
            if (!have && addend)
                acc = b.arithmetic(c, ac, x[i], y[j], (*addend)[k]);
            else if (!have)
                acc = b.arithmetic(c, 0, x[i], y[j], x[i]);
            else
                acc = b.arithmetic(c, 1, x[i], y[j], acc);
            have = true;
        }
        r[k] = acc;
    }
    return r;
}
// pc * x^2: 15 ops
inline FqT fq_sqr(CircuitBuilder& b, const FqT& x, u64 pc = 1) {
    FqT r;
    for (int k = 0; k < 5; k++) {
        bool have = false;
        Target acc = 0;
        for (int i = 0; i < 5; i++) {
            int j = (k - i + 5) % 5;
            if (j < i) continue;
            u64 c = i + j >= 5 ? gl::mul(pc, 3) : pc;
            if (i != j) c = gl::dbl(c);
            acc = have ? b.arithmetic(c, 1, x[i], x[j], acc) : b.arithmetic(c, 0, x[i], x[j], x[i]);
            have = true;
        }
        r[k] = acc;
    }
    return r;
}
// ca * a + cd * d
inline FqT fq_lin(CircuitBuilder& b, u64 ca, const FqT& a, u64 cd, const FqT& d) {
    FqT r;
    Target one = b.one();
    for (int i = 0; i < 5; i++) r[i] = b.arithmetic(ca, cd, a[i], one, d[i]);
    return r;
}
// (k z) * a + cd * d
inline FqT fq_lin_k1(CircuitBuilder& b, u64 k, const FqT& a, u64 cd, const FqT& d) {
    FqT r;
    Target one = b.one();
    for (int i = 0; i < 5; i++) r[i] = b.arithmetic(i == 0 ? gl::mul(3, k) : k, cd, a[(i + 4) % 5], one, d[i]);
    return r;
}
inline FqT fq_mul_k1(CircuitBuilder& b, u64 k, const FqT& a) {
    FqT r;
    for (int i = 0; i < 5; i++) r[i] = b.mul_const(i == 0 ? gl::mul(3, k) : k, a[(i + 4) % 5]);
    return r;
}
// 1/a through the norm: t = a^(p+p^2+p^3+p^4) costs two products with the Frobenius scalings folded into the gate
// constants, Norm = (a t)_0, one base-field inverse (hint + check), then five scalings.  a = 0 makes the proof fail.
inline FqT fq_inv(CircuitBuilder& b, const FqT& a) {
    u64 g1[5], g2[5], g = frob_gamma(), s1 = 1, s2 = 1;
    for (int i = 0; i < 5; i++) {
        g1[i] = s1;
        g2[i] = s2;
        s1 = gl::mul(s1, g);
        s2 = gl::mul(s2, gl::sqr(g));
    }
    FqT t = fq_mul(b, a, a, 1, nullptr, 1, g1, g2);  // a^p * a^(p^2)
    FqT t2 = fq_mul(b, t, t, 1, nullptr, 1, nullptr, g2);
    Target nrm = 0;
    for (int i = 0; i < 5; i++) {
        int j = (5 - i) % 5;
        u64 c = i + j >= 5 ? 3 : 1;
        nrm = i == 0 ? b.arithmetic(c, 0, a[i], t2[j], a[i]) : b.arithmetic(c, 1, a[i], t2[j], nrm);
    }
    Target inv = b.inverse(nrm);
    FqT r;
    for (int i = 0; i < 5; i++) r[i] = b.mul(t2[i], inv);
    return r;
}

inline ProjTarget neutral_target(CircuitBuilder& b) {
    FqT z = fq_const(b, fq_zero()), o = fq_const(b, fq_one());
    return ProjTarget{z, o, z, o};
}
// doubling: the unified addition with both operands equal (4 squarings + 6 products)
inline ProjTarget pdbl(CircuitBuilder& b, const ProjTarget& p) {
    FqT t1 = fq_sqr(b, p.X), t2 = fq_sqr(b, p.Z), t3 = fq_sqr(b, p.U), t4 = fq_sqr(b, p.T);
    FqT t5 = fq_mul(b, p.X, p.Z, 2), t6 = fq_mul(b, p.U, p.T, 2);
    FqT t7 = fq_lin_k1(b, B1, t2, 1, t1);
    FqT t8 = fq_mul(b, t4, t7);
    FqT w9 = fq_lin_k1(b, 2 * B1, t5, 2, t7);
    ProjTarget r;
    r.Z = fq_mul(b, t3, w9, NEG1, &t8, 1);
    r.T = fq_lin(b, 2, t8, NEG1, r.Z);
    FqT s = fq_lin(b, 2, t3, 1, t4), q = fq_lin(b, 1, t5, 1, t7);
    FqT d = fq_mul(b, s, q, 1, &t8, NEG1);
    r.X = fq_mul_k1(b, B1, d);
    FqT e = fq_lin_k1(b, B1, t2, NEG1, t1);
    r.U = fq_mul(b, t6, e);
    return r;
}
// unified addition with an affine second operand (Z2 = T2 = 1): 8 products; complete, so (0, 0) adds nothing
inline ProjTarget padd_mixed(CircuitBuilder& b, const ProjTarget& p, const PointTarget& q) {
    FqT t1 = fq_mul(b, p.X, q.x), t3 = fq_mul(b, p.U, q.u);
    FqT t5 = fq_mul(b, p.Z, q.x, 1, &p.X, 1), t6 = fq_mul(b, p.T, q.u, 1, &p.U, 1);
    FqT t7 = fq_lin_k1(b, B1, p.Z, 1, t1);
    FqT t8 = fq_mul(b, p.T, t7);
    FqT w9 = fq_lin_k1(b, 2 * B1, t5, 2, t7);
    ProjTarget r;
    r.Z = fq_mul(b, t3, w9, NEG1, &t8, 1);
    r.T = fq_lin(b, 2, t8, NEG1, r.Z);
    FqT s = fq_lin(b, 2, t3, 1, p.T), qq = fq_lin(b, 1, t5, 1, t7);
    FqT d = fq_mul(b, s, qq, 1, &t8, NEG1);
    r.X = fq_mul_k1(b, B1, d);
    FqT e = fq_lin_k1(b, B1, p.Z, NEG1, t1);
    r.U = fq_mul(b, t6, e);
    return r;
}
inline PointTarget to_affine(CircuitBuilder& b, const ProjTarget& p) {
    return PointTarget{fq_mul(b, p.X, fq_inv(b, p.Z)), fq_mul(b, p.U, fq_inv(b, p.T))};
}

// pod2 CircuitBuilderElliptic (call sites: ecgfp5/src/circuit.rs:37-38, elgamal/circuit.rs:34-37, :70-72)
inline PointTarget add_virtual_point_target(CircuitBuilder& b) {
    PointTarget p;
    for (auto& t : p.x) t = b.add_virtual_target();
    for (auto& t : p.u) t = b.add_virtual_target();
    // curve membership: u^2 (x^2 + 2x + b) - x = 0
    FqT w = fq_lin(b, 2, p.x, 1, fq_sqr(b, p.x));
    w[1] = b.add(w[1], b.constant(B1));
    FqT chk = fq_mul(b, fq_sqr(b, p.u), w, 1, &p.x, NEG1);
    for (auto& t : chk) b.connect(t, b.zero());
    return p;
}
inline PointTarget constant_point(CircuitBuilder& b, const Affine& a) { return PointTarget{fq_const(b, a.x), fq_const(b, a.u)}; }
inline bool point_as_constant(const CircuitBuilder& b, const PointTarget& p, Affine* out) {
    for (int i = 0; i < 5; i++)
        if (!b.target_as_constant(p.x[i], &out->x[i]) || !b.target_as_constant(p.u[i], &out->u[i])) return false;
    return true;
}
inline PointTarget add_point(CircuitBuilder& b, const PointTarget& p, const PointTarget& q) {
    FqT o = fq_const(b, fq_one());
    return to_affine(b, padd_mixed(b, ProjTarget{p.x, o, p.u, o}, q));
}
// pod2 CircuitBuilderBits::add_virtual_biguint320_target: 320 little-endian bits, each constrained to {0, 1}
static const size_t SCALAR_BITS = 320;
inline std::vector<BoolTarget> add_virtual_biguint320_target(CircuitBuilder& b) {
    std::vector<BoolTarget> bits(SCALAR_BITS);
    for (auto& t : bits) {
        t = b.add_virtual_bool_target_unsafe();
        b.connect(b.mul_sub(t.target, t.target, t.target), b.zero());
    }
    return bits;
}
// bits * P.  A variable base costs a doubling and a mixed addition per bit (the addend is bit * (x, u): the neutral
// when the bit is clear).  A constant base (the generator) needs no doublings: sum_i bit_i * (2^i P) with the
// multiples precomputed on the host.
inline PointTarget multiply_point(CircuitBuilder& b, const std::vector<BoolTarget>& bits, const PointTarget& p) {
    if (bits.size() != SCALAR_BITS) throw std::runtime_error("multiply_point takes 320 bits");
    ProjTarget acc = neutral_target(b);
    Affine base;
    if (point_as_constant(b, p, &base)) {
        Point m = Point::from_affine(base);
        for (size_t i = 0; i < SCALAR_BITS; i++) {
            Affine a = m.affine();
            PointTarget q;
            for (int k = 0; k < 5; k++) {
                q.x[k] = b.mul_const(a.x[k], bits[i].target);
                q.u[k] = b.mul_const(a.u[k], bits[i].target);
            }
            acc = padd_mixed(b, acc, q);
            m = m.dbl();
        }
    } else {
        for (size_t i = SCALAR_BITS; i-- > 0;) {
            acc = pdbl(b, acc);
            PointTarget q;
            for (int k = 0; k < 5; k++) {
                q.x[k] = b.mul(bits[i].target, p.x[k]);
                q.u[k] = b.mul(bits[i].target, p.u[k]);
            }
            acc = padd_mixed(b, acc, q);
        }
    }
    return to_affine(b, acc);
}
// ecgfp5/src/circuit.rs:35
inline PointTarget public_key_target(CircuitBuilder& b, const std::vector<BoolTarget>& sk_bits) {
    return multiply_point(b, sk_bits, constant_point(b, generator()));
}
// elgamal/circuit.rs:28
inline void elgamal_encrypt_target(CircuitBuilder& b, const PointTarget& pk, const std::vector<BoolTarget>& nonce, const PointTarget& msg,
                                   PointTarget* c0, PointTarget* c1) {
    PointTarget g = constant_point(b, generator());
    *c0 = multiply_point(b, nonce, g);
    PointTarget npk = multiply_point(b, nonce, pk);
    *c1 = add_point(b, msg, npk);
}
// hashed_elgamal/circuit.rs:33
inline void hashed_elgamal_encrypt_target(CircuitBuilder& b, const PointTarget& pk, const std::vector<BoolTarget>& nonce, const Target msg[5],
                                          PointTarget* c0, Target ct[5]) {
    PointTarget g = constant_point(b, generator());
    *c0 = multiply_point(b, nonce, g);
    PointTarget npk = multiply_point(b, nonce, pk);
    std::vector<Target> in(npk.x.begin(), npk.x.end());
    in.insert(in.end(), npk.u.begin(), npk.u.end());
    auto h = b.hash_n_to_m_no_pad(in, 5);
    for (int i = 0; i < 5; i++) ct[i] = b.add(msg[i], h[i]);
}

}  // namespace ecgfp5
}  // namespace p2

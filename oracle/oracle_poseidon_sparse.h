// ORACLE -- TEST INFRASTRUCTURE ONLY.  Optional faster form of the oracle's OWN Poseidon permutation, used by bench.py's
// cpu_baseline leg (orc_set_fast_hash(1)) so that the reported CPU number is not handicapped by the textbook 30-round loop.
// The parity checks of tests/ keep the textbook form (oracle_field.h poseidon_permute); tests/test_oracle_kat.py holds this
// form to it on random states and on upstream's all-zero test vector, and checks that a whole proof comes out byte-identical
// either way.
//
// The 22 partial rounds apply the full 12x12 MDS matrix M although only one word went through the S-box.  The published
// optimisation (Grassi et al., "Poseidon", appendix "optimised implementation"; plonky2's poseidon.rs follows it) rewrites
// them with sparse matrices.  Derived here at start-up from M and the round constants, nothing is copied in:
//   * constants: in round i only the part of the constant that reaches the S-box (word 0) has to be added before it; the
//     other eleven words commute with the S-box and are pushed through M into the next round (PRE[i] = the scalar that
//     remains; TAIL = what falls out at the end, added before the first full round that follows);
//   * matrix: write M = [[m, r], [c, H]] (m scalar, r row, c column, H 11x11).  Then M = S . diag(1, H) with
//     S = [[m, r H^-1], [c, I]], and diag(1, H) commutes with the partial S-box, so it is merged into the matrix of the
//     round BEFORE; done from the last partial round backwards this leaves 22 sparse S_i (a row ROW[i] and a column COL[i])
//     and one dense 11x11 matrix HEAD applied once before the first partial round.
// Per partial round: 4 multiplications for x^7, an 11-term dot product accumulated in 128+ bits and reduced ONCE, and eleven
// multiply-adds -- against 144 multiply-accumulates and twelve reductions.
#pragma once
#include "oracle_field.h"

namespace orc {

struct SparsePoseidon {
    u64 PRE[22];        // scalar added to word 0 before the S-box of partial round i
    u64 ROW[22][11];    // new word 0 = m * s0 + ROW[i] . s[1..]
    u64 COL[22][11];    // s[j] += COL[i][j-1] * s0
    u64 HEAD[11][11];   // dense layer on words 1.. before the first partial round
    u64 TAIL[12];       // added to the state after the last partial round
    u64 m00;
};

typedef std::vector<std::vector<u64>> Mat;
static inline Mat mat_mul(const Mat& A, const Mat& B) {
    Mat C(A.size(), std::vector<u64>(B[0].size(), 0));
    for (size_t i = 0; i < A.size(); i++)
        for (size_t k = 0; k < B.size(); k++)
            if (A[i][k])
                for (size_t j = 0; j < B[0].size(); j++) C[i][j] = fadd(C[i][j], fmul(A[i][k], B[k][j]));
    return C;
}
static inline Mat mat_inv(Mat A) {  // Gauss-Jordan over the field
    const size_t n = A.size();
    Mat I(n, std::vector<u64>(n, 0));
    for (size_t i = 0; i < n; i++) I[i][i] = 1;
    for (size_t col = 0; col < n; col++) {
        size_t piv = col;
        while (piv < n && A[piv][col] == 0) piv++;
        if (piv == n) throw std::runtime_error("singular matrix in the Poseidon derivation");
        std::swap(A[piv], A[col]);
        std::swap(I[piv], I[col]);
        const u64 inv = finv(A[col][col]);
        for (size_t j = 0; j < n; j++) A[col][j] = fmul(A[col][j], inv), I[col][j] = fmul(I[col][j], inv);
        for (size_t r = 0; r < n; r++) {
            if (r == col || A[r][col] == 0) continue;
            const u64 f = A[r][col];
            for (size_t j = 0; j < n; j++) A[r][j] = fsub(A[r][j], fmul(f, A[col][j])), I[r][j] = fsub(I[r][j], fmul(f, I[col][j]));
        }
    }
    return I;
}
static inline const SparsePoseidon& sparse_poseidon() {
    static const SparsePoseidon S = [] {
        SparsePoseidon sp;
        Mat M(12, std::vector<u64>(12));
        for (int r = 0; r < 12; r++)
            for (int c = 0; c < 12; c++) M[r][c] = fadd(MDS_CIRC[(c - r + 12) % 12], r == c ? MDS_DIAG[r] : 0);
        sp.m00 = M[0][0];
        // constants, forwards: carry = what earlier rounds pushed into this one
        u64 carry[12] = {0};
        for (int i = 0; i < 22; i++) {
            u64 t[12];
            for (int k = 0; k < 12; k++) t[k] = fadd(ROUND_CONSTANTS[12 * (4 + i) + k], carry[k]);
            sp.PRE[i] = t[0];
            for (int r = 0; r < 12; r++) {  // carry = M . (0, t[1..])
                u64 acc = 0;
                for (int k = 1; k < 12; k++) acc = fadd(acc, fmul(M[r][k], t[k]));
                carry[r] = acc;
            }
        }
        memcpy(sp.TAIL, carry, sizeof(carry));
        // matrices, backwards: Q = the matrix of the round being split (M with the later rounds' dense parts merged in)
        Mat Q = M, H;
        for (int i = 21; i >= 0; i--) {
            H.assign(11, std::vector<u64>(11));
            for (int r = 0; r < 11; r++)
                for (int c = 0; c < 11; c++) H[r][c] = Q[r + 1][c + 1];
            Mat row(1, std::vector<u64>(11));
            for (int c = 0; c < 11; c++) row[0][c] = Q[0][c + 1];
            Mat rh = mat_mul(row, mat_inv(H));
            for (int c = 0; c < 11; c++) sp.ROW[i][c] = rh[0][c];
            for (int r = 0; r < 11; r++) sp.COL[i][r] = Q[r + 1][0];
            if (Q[0][0] != sp.m00) throw std::runtime_error("Poseidon derivation: corner element moved");
            Mat D(12, std::vector<u64>(12, 0));
            D[0][0] = 1;
            for (int r = 0; r < 11; r++)
                for (int c = 0; c < 11; c++) D[r + 1][c + 1] = H[r][c];
            Q = mat_mul(D, M);
        }
        for (int r = 0; r < 11; r++)
            for (int c = 0; c < 11; c++) sp.HEAD[r][c] = H[r][c];
        return sp;
    }();
    return S;
}

// Inside this permutation words are arbitrary u64 representatives (not necessarily < p); the output is canonicalised.
//   2^64 = 2^32 - 1 and 2^96 = -1 (mod p):  lo + 2^64 (hl + 2^32 hh) = lo - hh + hl (2^32 - 1), folded with wrap corrections
static const u64 EPS = 0xFFFFFFFFull;
static inline u64 lred(u128 x) {
    // branch-free: the two wrap tests are coin flips on random data, and a mispredicted branch costs more than the whole fold
    const u64 lo = (u64)x, hi = (u64)(x >> 64), hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh;
    t0 -= (0 - (u64)(lo < hh)) & EPS;  // borrowed 2^64 = EPS (mod p); cannot underflow again: t0 >= 2^64 - 2^32 after a borrow
    const u64 t1 = (hl << 32) - hl;    // hl * EPS < 2^64
    u64 r = t0 + t1;
    r += (0 - (u64)(r < t1)) & EPS;    // wrapped: + 2^64 = + EPS; cannot wrap again
    return r;
}
static inline u64 lmul(u64 a, u64 b) { return lred((u128)a * b); }
static inline u64 lpow7(u64 x) {
    const u64 x2_ = lmul(x, x), x3 = lmul(x2_, x), x4 = lmul(x2_, x2_);
    return lmul(x3, x4);
}
static inline u64 ladd(u64 a, u64 c) {  // a arbitrary, c canonical: some representative of a + c
    u64 r = a + c;
    r += (0 - (u64)(r < c)) & EPS;    // r < c <= p - 1 after the wrap, so + EPS cannot wrap
    return r;
}
static inline u64 lcanon(u64 a) { return a >= MODULUS ? a - MODULUS : a; }
// a * b accumulated into a 192-bit counter (lo, hi, top): no reduction until the dot product is complete
struct Acc192 {
    u64 lo = 0, hi = 0, top = 0;
    inline void mac(u64 a, u64 b) {
        const u128 p = (u128)a * b;
        const u128 s = (u128)lo + (u64)p;
        lo = (u64)s;
        const u128 h = (u128)hi + (u64)(p >> 64) + (u64)(s >> 64);
        hi = (u64)h;
        top += (u64)(h >> 64);
    }
    inline u64 reduce() const {  // lo + 2^64 hi + 2^128 top, with 2^128 = -2^32 (mod p); top is tiny (< 16)
        const u64 base = lcanon(lred(((u128)hi << 64) | lo));
        return fsub(base, top << 32);
    }
};
static inline void full_round(u64 st[12], const u64* rc, const u64* extra) {
    u64 tw[24];
    for (int i = 0; i < 12; i++) {
        u64 x = ladd(st[i], rc[i]);
        if (extra) x = ladd(x, extra[i]);
        tw[i] = tw[i + 12] = lpow7(x);
    }
    for (int r = 0; r < 12; r++) {
        u128 acc = (u128)tw[r] * MDS_DIAG[r];
        for (int i = 0; i < 12; i++) acc += (u128)tw[i + r] * MDS_CIRC[i];   // < 2^64 * 2^9: fits
        st[r] = lred(acc);
    }
}
static inline void poseidon_permute_sparse(u64 st[12]) {
    const SparsePoseidon& S = sparse_poseidon();
    for (int r = 0; r < 4; r++) full_round(st, ROUND_CONSTANTS + 12 * r, nullptr);
    {
        u64 t[11];
        for (int r = 0; r < 11; r++) {
            Acc192 a;
            for (int c = 0; c < 11; c++) a.mac(S.HEAD[r][c], st[c + 1]);
            t[r] = a.reduce();
        }
        memcpy(st + 1, t, sizeof(t));
    }
    for (int i = 0; i < 22; i++) {
        const u64 s0 = lpow7(ladd(st[0], S.PRE[i]));
        Acc192 a;
        a.mac(S.m00, s0);
        for (int j = 0; j < 11; j++) a.mac(S.ROW[i][j], st[j + 1]);
        for (int j = 0; j < 11; j++) st[j + 1] = lred((u128)S.COL[i][j] * s0 + st[j + 1]);
        st[0] = a.reduce();
    }
    full_round(st, ROUND_CONSTANTS + 12 * 26, S.TAIL);
    for (int r = 27; r < 30; r++) full_round(st, ROUND_CONSTANTS + 12 * r, nullptr);
    for (int i = 0; i < 12; i++) st[i] = lcanon(st[i]);
}

// ---- lane-parallel forms (oracle_poseidon_simd.h, instantiated in oracle_simd_avx512.cpp / oracle_simd_avx2.cpp)
void simd512_hash_rows(const SparsePoseidon* S, const u64* rows, size_t row_stride, size_t width, Digest* out);
void simd512_compress_pairs(const SparsePoseidon* S, const Digest* src, Digest* dst);
void simd256_hash_rows(const SparsePoseidon* S, const u64* rows, size_t row_stride, size_t width, Digest* out);
void simd256_compress_pairs(const SparsePoseidon* S, const Digest* src, Digest* dst);
void simd512_test_arith(const u64* a, const u64* b, const u64* c, u64* out);
void simd256_test_arith(const u64* a, const u64* b, const u64* c, u64* out);
// lanes the host can run: 8 (AVX-512 F + DQ), 4 (AVX2) or 1; g_simd_cap (orc_set_simd_lanes) lowers it for the tests
static int g_simd_cap = 8;
static inline int simd_lanes() {
#if defined(__x86_64__) && !defined(ORC_NO_SIMD)
    static const int host = [] {
        __builtin_cpu_init();
        if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) return 8;
        if (__builtin_cpu_supports("avx2")) return 4;
        return 1;
    }();
    const int l = host < g_simd_cap ? host : g_simd_cap;
    return l >= 8 ? 8 : l >= 4 ? 4 : 1;
#else
    return 1;
#endif
}
// hash_no_pad of rows [0, count) of `width` > 4 words each (row-major), digests to out[]: SIMD blocks, scalar tail
static inline void hash_rows_fast(const u64* rows, size_t first, size_t count, size_t width, Digest* out) {
    const SparsePoseidon& S = sparse_poseidon();
    const size_t lanes = (size_t)simd_lanes();
    size_t i = first;
    const size_t end = first + count;
#if defined(__x86_64__) && !defined(ORC_NO_SIMD)
    if (lanes == 8)
        for (; i + 8 <= end; i += 8) simd512_hash_rows(&S, rows + i * width, width, width, out + i);
    else if (lanes == 4)
        for (; i + 4 <= end; i += 4) simd256_hash_rows(&S, rows + i * width, width, width, out + i);
#endif
    for (; i < end; i++) out[i] = hash_no_pad(rows + i * width, width);
}
// dst[i] = compress(src[2 i], src[2 i + 1]) for i in [first, first + count)
static inline void compress_level_fast(const Digest* src, Digest* dst, size_t first, size_t count) {
    const SparsePoseidon& S = sparse_poseidon();
    const size_t lanes = (size_t)simd_lanes();
    size_t i = first;
    const size_t end = first + count;
#if defined(__x86_64__) && !defined(ORC_NO_SIMD)
    if (lanes == 8)
        for (; i + 8 <= end; i += 8) simd512_compress_pairs(&S, src + 2 * i, dst + i);
    else if (lanes == 4)
        for (; i + 4 <= end; i += 4) simd256_compress_pairs(&S, src + 2 * i, dst + i);
#endif
    for (; i < end; i++) dst[i] = compress(src[2 * i], src[2 * i + 1]);
}

}  // namespace orc

// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load anything under oracle/.
//
// CPU restatement of the arithmetic underneath plonky2's prover for this repo's hot path:
// GoldilocksField, its quadratic extension, Poseidon-12, the overwrite-mode sponge, Merkle caps, the duplex
// Challenger and radix-2 FFTs.  The algorithm lives in the third-party crate `plonky2`
// (git rev 109d517d09c210ae4c2cee381d3e3fbc04aa3812, /root/reference/Cargo.toml:12), which is NOT vendored
// under /root/reference; it is restated here from its published definition.
//
// Pinning status: Poseidon is pinned by upstream's published test vector for the all-zero input
// (first word 0x3c18a9786cb0b359, last word 0x1792b1c4342109d7) and by the first round constants
// (0xb585f766f2144405, ...), both reproduced by tools/gen_poseidon_constants.py.  The AES/GCM restatement is
// pinned by the FIPS-197 / NIST CAVP vectors in the reference's tests.  PROOF BYTES ARE PARITY-UNPINNED: the
// reference holds no golden proof, digest or gate count.
//
// Deliberately written in a different style from plonky2-aes_amd/csrc/gl.h (unsigned __int128 arithmetic,
// whole-word MDS accumulation, full-size zero-padded FFTs) so the two implementations check each other.
#pragma once
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <stdexcept>
#include <vector>

namespace orc {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef unsigned __int128 u128;

static const u64 MODULUS = 18446744069414584321ull;

static inline u64 fred(u128 x) {
    // x = lo + 2^64*(hl + 2^32*hh);  2^64 = 2^32-1, 2^96 = -1  (mod p):  x = lo - hh + hl*(2^32-1).
    // Written without data-dependent branches (the wrap tests are coin flips on random data): a borrow of 2^64 is paid
    // back as -(2^32-1), a carry as +(2^32-1); neither can wrap twice (after a borrow t0 >= 2^64-2^32; after a carry
    // r < hl*(2^32-1) <= 2^64-2^33+1).  The result of the folds is some representative below 2^64 < 2p.
    const u64 E = 0xffffffffull;
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hh = hi >> 32, hl = hi & E;
    u64 t0 = lo - hh;
    t0 -= (0 - (u64)(lo < hh)) & E;
    u64 t1 = (hl << 32) - hl;
    u64 r = t0 + t1;
    r += (0 - (u64)(r < t1)) & E;
    return r >= MODULUS ? r - MODULUS : r;
}
// canonical in, canonical out; no data-dependent branches (sum >= p and a < b are coin flips on random data)
static inline u64 fadd(u64 a, u64 b) {
    const u64 s = a + b, t = s - MODULUS;
    return ((s < a) | (s >= MODULUS)) ? t : s;
}
static inline u64 fsub(u64 a, u64 b) { return a - b + ((0 - (u64)(a < b)) & MODULUS); }
static inline u64 fneg(u64 a) { return a ? MODULUS - a : 0; }
static inline u64 fmul(u64 a, u64 b) { return fred((u128)a * b); }
static inline u64 fpow(u64 b, u64 e) {
    u64 r = 1;
    for (; e; e >>= 1) {
        if (e & 1) r = fmul(r, b);
        b = fmul(b, b);
    }
    return r;
}
static inline u64 finv(u64 a) { return fpow(a, MODULUS - 2); }
static const u64 GENERATOR = 14293326489335486720ull;     // MULTIPLICATIVE_GROUP_GENERATOR
static const u64 TWO_ADIC_ROOT = 7277203076849721926ull;  // POWER_OF_TWO_GENERATOR, order 2^32
static inline u64 root_of_unity(int bits) {
    u64 r = TWO_ADIC_ROOT;
    for (int i = bits; i < 32; i++) r = fmul(r, r);
    return r;
}
static inline size_t rev_bits(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// GF(p^2) = F[x]/(x^2-7)
struct X2 {
    u64 c0, c1;
};
static inline X2 x2(u64 a, u64 b = 0) { return X2{a, b}; }
static inline X2 operator+(X2 a, X2 b) { return X2{fadd(a.c0, b.c0), fadd(a.c1, b.c1)}; }
static inline X2 operator-(X2 a, X2 b) { return X2{fsub(a.c0, b.c0), fsub(a.c1, b.c1)}; }
static inline X2 operator*(X2 a, X2 b) {
    return X2{fadd(fmul(a.c0, b.c0), fmul(7, fmul(a.c1, b.c1))), fadd(fmul(a.c0, b.c1), fmul(a.c1, b.c0))};
}
static inline X2 operator*(X2 a, u64 s) { return X2{fmul(a.c0, s), fmul(a.c1, s)}; }
static inline bool operator==(X2 a, X2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline X2 xinv(X2 a) {
    u64 nrm = fsub(fmul(a.c0, a.c0), fmul(7, fmul(a.c1, a.c1)));
    u64 ni = finv(nrm);
    return X2{fmul(a.c0, ni), fmul(fneg(a.c1), ni)};
}
static inline X2 xpow(X2 b, u64 e) {
    X2 r = x2(1);
    for (; e; e >>= 1) {
        if (e & 1) r = r * b;
        b = b * b;
    }
    return r;
}

// ------------------------------------------------------------- Poseidon-12 (naive schedule)
static const u64 ROUND_CONSTANTS[360] = {
#include "poseidon_rc.inc"
};
static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static inline u64 pow7(u64 x) {
    u64 x2_ = fmul(x, x), x3 = fmul(x2_, x), x6 = fmul(x3, x3);
    return fmul(x6, x);
}
// bench.py's cpu_baseline leg may switch every permutation of the oracle to the sparse-partial-round form of
// oracle_poseidon_sparse.h (orc_set_fast_hash): the same function, held to this one by tests/test_oracle_kat.py
static bool g_sparse_poseidon = false;
static inline void poseidon_permute_sparse(u64 st[12]);
struct Digest;
static inline void hash_rows_fast(const u64* rows, size_t first, size_t count, size_t width, Digest* out);
static inline void compress_level_fast(const Digest* src, Digest* dst, size_t first, size_t count);
static inline void poseidon_permute(u64 st[12]) {
    if (g_sparse_poseidon) return poseidon_permute_sparse(st);
    for (int round = 0; round < 30; round++) {
        for (int i = 0; i < 12; i++) st[i] = fadd(st[i], ROUND_CONSTANTS[12 * round + i]);
        bool full = round < 4 || round >= 26;
        if (full)
            for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
        else
            st[0] = pow7(st[0]);
        u64 twice[24], nx[12];  // state laid out twice so that the circulant rows need no index wrap
        memcpy(twice, st, 96);
        memcpy(twice + 12, st, 96);
        for (int r = 0; r < 12; r++) {
            u128 acc = 0;
            for (int i = 0; i < 12; i++) acc += (u128)twice[i + r] * MDS_CIRC[i];
            acc += (u128)st[r] * MDS_DIAG[r];
            nx[r] = fred(acc);
        }
        memcpy(st, nx, sizeof(nx));
    }
}

struct Digest {
    u64 e[4];
    bool operator==(const Digest& o) const { return memcmp(e, o.e, 32) == 0; }
};

// hash_n_to_hash_no_pad (overwrite-mode sponge, rate 8)
static inline Digest hash_no_pad(const u64* in, size_t len) {
    u64 st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; i++) st[i] = in[off + i];
        poseidon_permute(st);
    }
    Digest d;
    memcpy(d.e, st, 32);
    return d;
}
// hash_or_noop: <= 4 elements are zero-padded into the digest unhashed
static inline Digest hash_or_noop(const u64* in, size_t len) {
    if (len <= 4) {
        Digest d = {{0, 0, 0, 0}};
        for (size_t i = 0; i < len; i++) d.e[i] = in[i];
        return d;
    }
    return hash_no_pad(in, len);
}
static inline Digest compress(const Digest& l, const Digest& r) {
    u64 st[12] = {l.e[0], l.e[1], l.e[2], l.e[3], r.e[0], r.e[1], r.e[2], r.e[3], 0, 0, 0, 0};
    poseidon_permute(st);
    Digest d;
    memcpy(d.e, st, 32);
    return d;
}
// hash_pad: append 1, zeros until len+1 is a multiple of the sponge WIDTH, then 1
static inline Digest hash_pad(const std::vector<u64>& in) {
    std::vector<u64> p(in);
    p.push_back(1);
    while ((p.size() + 1) % 12 != 0) p.push_back(0);
    p.push_back(1);
    return hash_no_pad(p.data(), p.size());
}

// ------------------------------------------------------------- Merkle tree with cap
struct MerkleTree {
    int height = 0;      // log2(#leaves)
    int cap_height = 0;  // effective (min(cap_height, height))
    std::vector<std::vector<Digest>> levels;  // levels[0] = leaf digests ... levels[height-cap_height] = cap
    const std::vector<Digest>& cap() const { return levels.back(); }
    // siblings from the leaf up to (excluding) the cap level
    std::vector<Digest> prove(size_t leaf) const {
        std::vector<Digest> s;
        for (size_t l = 0; l + 1 < levels.size(); l++) {
            s.push_back(levels[l][leaf ^ 1]);
            leaf >>= 1;
        }
        return s;
    }
};
// leaves: row-major [num_leaves][width]
static inline MerkleTree build_merkle(const u64* leaves, size_t num_leaves, size_t width, int cap_height) {
    MerkleTree t;
    while (((size_t)1 << t.height) < num_leaves) t.height++;
    t.cap_height = cap_height < t.height ? cap_height : t.height;
    t.levels.emplace_back(num_leaves);
    const size_t BLK = 64;  // rows per work item (a multiple of every lane count of the fast hash)
    const bool fast = g_sparse_poseidon && width > 4;
#pragma omp parallel for schedule(static)
    for (size_t i0 = 0; i0 < num_leaves; i0 += BLK) {
        const size_t cnt = num_leaves - i0 < BLK ? num_leaves - i0 : BLK;
        if (fast)
            hash_rows_fast(leaves, i0, cnt, width, t.levels[0].data());
        else
            for (size_t i = i0; i < i0 + cnt; i++) t.levels[0][i] = hash_or_noop(leaves + i * width, width);
    }
    for (int l = 0; l < t.height - t.cap_height; l++) {
        size_t m = t.levels[l].size() / 2;
        t.levels.emplace_back(m);
        auto& src = t.levels[l];
        auto& dst = t.levels[l + 1];
#pragma omp parallel for schedule(static) if (m > 256)
        for (size_t i0 = 0; i0 < m; i0 += BLK) {
            const size_t cnt = m - i0 < BLK ? m - i0 : BLK;
            if (g_sparse_poseidon)
                compress_level_fast(src.data(), dst.data(), i0, cnt);
            else
                for (size_t i = i0; i < i0 + cnt; i++) dst[i] = compress(src[2 * i], src[2 * i + 1]);
        }
    }
    return t;
}

// ------------------------------------------------------------- Challenger (duplex sponge, overwrite mode)
struct Challenger {
    u64 state[12];
    std::vector<u64> input, output;
    Challenger() { memset(state, 0, sizeof(state)); }
    void duplexing() {
        for (size_t i = 0; i < input.size(); i++) state[i] = input[i];
        input.clear();
        poseidon_permute(state);
        output.assign(state, state + 8);
    }
    void observe(u64 x) {
        output.clear();
        input.push_back(x);
        if (input.size() == 8) duplexing();
    }
    void observe(const Digest& d) {
        for (int i = 0; i < 4; i++) observe(d.e[i]);
    }
    void observe_cap(const std::vector<Digest>& cap) {
        for (auto& d : cap) observe(d);
    }
    void observe(X2 x) {
        observe(x.c0);
        observe(x.c1);
    }
    u64 challenge() {
        if (!input.empty() || output.empty()) duplexing();
        u64 v = output.back();
        output.pop_back();
        return v;
    }
    X2 ext_challenge() {
        u64 a = challenge();
        u64 b = challenge();
        return X2{a, b};
    }
};

// ------------------------------------------------------------- FFT (natural in, natural out)
// forward: X[k] = sum_j x[j] w^(jk), w = root_of_unity(bits)
static inline void fft_inplace(u64* a, int bits, bool inverse) {
    size_t n = (size_t)1 << bits;
    for (size_t i = 0; i < n; i++) {
        size_t j = rev_bits(i, bits);
        if (i < j) {
            u64 t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    }
    for (int s = 1; s <= bits; s++) {
        size_t m = (size_t)1 << s, half = m >> 1;
        u64 wm = root_of_unity(s);
        if (inverse) wm = finv(wm);
        std::vector<u64> tw(half);
        u64 w = 1;
        for (size_t j = 0; j < half; j++) {
            tw[j] = w;
            w = fmul(w, wm);
        }
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                u64 t = fmul(tw[j], a[k + j + half]), u = a[k + j];
                a[k + j] = fadd(u, t);
                a[k + j + half] = fsub(u, t);
            }
    }
    if (inverse) {
        u64 ni = finv((u64)n % MODULUS);
        for (size_t i = 0; i < n; i++) a[i] = fmul(a[i], ni);
    }
}
// w^j for j < 2^(bits-1), w = root_of_unity(bits): built once per size
static inline const std::vector<u64>& forward_twiddles(int bits) {
    static std::vector<u64> tables[33];
    static std::mutex guard;
    std::lock_guard<std::mutex> lock(guard);
    std::vector<u64>& t = tables[bits];
    if (t.empty()) {
        t.resize(bits ? (size_t)1 << (bits - 1) : 1);
        const u64 w = root_of_unity(bits);
        u64 x = 1;
        for (auto& v : t) {
            v = x;
            x = fmul(x, w);
        }
    }
    return t;
}
// per stage of the decimation-in-frequency transform (half = 2^(bits-1-s)): its twiddles w^(j * 2^s), j < half, contiguous
static inline const std::vector<std::vector<u64>>& forward_stage_twiddles(int bits) {
    static std::vector<std::vector<u64>> tables[33];
    const std::vector<u64>& tw = forward_twiddles(bits);
    static std::mutex guard;
    std::lock_guard<std::mutex> lock(guard);
    std::vector<std::vector<u64>>& t = tables[bits];
    if (t.empty() && bits) {
        t.resize(bits);
        for (int s = 0; s < bits; s++) {
            const size_t half = (size_t)1 << (bits - 1 - s), step = (size_t)1 << s;
            t[s].resize(half);
            for (size_t j = 0; j < half; j++) t[s][j] = tw[j * step];
        }
    }
    return t;
}
void simd512_dif_stage(u64* a, size_t n, size_t half, const u64* tw);
void simd256_dif_stage(u64* a, size_t n, size_t half, const u64* tw);
static inline int simd_lanes();
// the same transform as fft_inplace(a, bits, false) with the output left in bit-reversed order (position rev(k) holds X[k]):
// decimation in frequency, no permutation pass.  The low-degree extensions are stored in that order.  With the fast switch of
// the cpu_baseline leg on, stages wide enough run eight (AVX-512) or four (AVX2) butterflies per instruction.
static inline void fft_bitrev_out(u64* a, int bits) {
    const size_t n = (size_t)1 << bits;
    const std::vector<std::vector<u64>>& stw = forward_stage_twiddles(bits);
    const size_t lanes = g_sparse_poseidon ? (size_t)simd_lanes() : 1;
    int s = 0;
    for (size_t half = n >> 1; half >= 1; half >>= 1, s++) {
        const u64* tw = stw[s].data();
#if defined(__x86_64__) && !defined(ORC_NO_SIMD)
        if (lanes == 8 && half >= 8) {
            simd512_dif_stage(a, n, half, tw);
            continue;
        }
        if (lanes == 4 && half >= 4) {
            simd256_dif_stage(a, n, half, tw);
            continue;
        }
#endif
        for (size_t k = 0; k < n; k += 2 * half)
            for (size_t j = 0; j < half; j++) {
                const u64 u = a[k + j], v = a[k + j + half];
                a[k + j] = fadd(u, v);
                a[k + j + half] = fmul(fsub(u, v), tw[j]);
            }
    }
}
// out[rev(k)] = value of `coeffs` (zero-padded to 2^bits) at shift * w^k;  shift_pows[i] = shift^i
static inline void coset_fft_bitrev_out(const std::vector<u64>& coeffs, const std::vector<u64>& shift_pows, int bits, u64* out) {
    const size_t n = (size_t)1 << bits;
    for (size_t i = 0; i < coeffs.size(); i++) out[i] = fmul(coeffs[i], shift_pows[i]);
    for (size_t i = coeffs.size(); i < n; i++) out[i] = 0;
    fft_bitrev_out(out, bits);
}
static inline std::vector<u64> powers_of(u64 x, size_t count) {
    std::vector<u64> p(count);
    u64 s = 1;
    for (auto& v : p) {
        v = s;
        s = fmul(s, x);
    }
    return p;
}
// values of the polynomial `coeffs` (zero-padded to 2^bits) on shift*<w>, natural order
static inline std::vector<u64> coset_fft(const std::vector<u64>& coeffs, int bits, u64 shift) {
    std::vector<u64> a((size_t)1 << bits, 0);
    u64 s = 1;
    for (size_t i = 0; i < coeffs.size(); i++) {
        a[i] = fmul(coeffs[i], s);
        s = fmul(s, shift);
    }
    fft_inplace(a.data(), bits, false);
    return a;
}
static inline std::vector<u64> coset_ifft(std::vector<u64> vals, int bits, u64 shift) {
    fft_inplace(vals.data(), bits, true);
    u64 si = finv(shift), s = 1;
    for (auto& v : vals) {
        v = fmul(v, s);
        s = fmul(s, si);
    }
    return vals;
}

}  // namespace orc

#include "oracle_poseidon_sparse.h"

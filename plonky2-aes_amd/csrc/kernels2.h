// Second half of the pipeline kernels: quotient (gate-constraint) evaluation, openings, FRI.
#pragma once
#include "kernels.h"

namespace p2k {

// ------------------------------------------------------------------------------------------- quotient
struct QuotientArgs {
    const u64* pre_lde;    // [ncc + R][8n]   constants | sigmas
    const u64* wires_lde;  // [active][8n]
    const u64* zs_lde;     // [zs_cols][8n]
    size_t wires_batch_stride, zs_batch_stride;
    const u64* chal;
    const u64* xs;      // [8n] x at LDE position p (bit-reversed layout)
    const u64* l0;      // [8n] L_0(x)
    const u64* zh_inv;  // [8] indexed by coset j
    const u64* k_is;
    const u64* apow;  // [batch][2][APOW_STRIDE] alpha powers
    u64* out;                   // [batch][NC][8n]
    size_t out_batch_stride;
    u32 n, logn, rate_bits, R, ncc, nsel, nls, NC, npp, qdf, num_luts, nsldc, lut_deg, nlp;
    u32 num_gates, num_gate_constraints;
    u32 gate_kind[p2::MAX_GATE_TYPES], gate_sel[p2::MAX_GATE_TYPES], group_lo[p2::MAX_GATE_TYPES], group_hi[p2::MAX_GATE_TYPES];
    u32 lut_last_row[8];  // last_lut row per LUT: RE there equals get_lut_poly (read from the zs VALUES)
    const u64* zs_values;  // [zs_cols][n]
    size_t zs_values_batch_stride;
};

// alpha^k for k < count, per proof and challenge: apow[(proof*2 + i)*APOW_STRIDE + k]
static const u32 APOW_STRIDE = 256;
__global__ void k_alpha_pows(const u64* chal, u64* apow, u32 batch, u32 count) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * batch) return;
    u64 alpha = chal[(size_t)(t >> 1) * CH_WORDS + CH_ALPHAS + (t & 1)];
    u64* o = apow + (size_t)t * APOW_STRIDE;
    u64 p = 1;
    for (u32 k = 0; k < count; k++) {
        o[k] = p;
        p = gl::mul(p, alpha);
    }
}

// sum_k term_k * alpha_i^k for both challenges, accumulated lazily (one reduction at the very end)
struct AlphaAcc {
    glf::Acc acc[2];
    const u64* pw[2];
    __device__ __forceinline__ void init(const u64* apow_proof) {
        acc[0].init();
        acc[1].init();
        pw[0] = apow_proof;
        pw[1] = apow_proof + APOW_STRIDE;
    }
    __device__ __forceinline__ void add(u32 idx, u64 term) {
        // the powers are per proof (= per workgroup row blockIdx.y) and idx is never lane-dependent: uniform, read from SGPRs
        acc[0].fma_k(pw[0][idx], term);
        acc[1].fma_k(pw[1][idx], term);
    }
};

// One thread per point of the LDE coset (bit-reversed position p); evaluates every constraint of
// eval_vanishing_poly_base (term k is weighted by alpha^k, k in plonky2's order [Z(1) | partial products | lookups |
// gates]) for both challenges at once and divides by Z_H.  The 80 wire columns are walked TWICE: once for the permutation
// argument with the ArithmeticGate constraints riding along (a chunk of 8 wires is exactly two arithmetic ops), once for the
// LookupTableGate and LookupGate views together (6 columns = 2 table slots + 3 looking slots).  Round 1 walked them once per
// view (4x) and fetched 2.3x its algorithmic bytes (profiles/r01_traffic.json; the very first version 3.8x).
// ONE_WALK (round 3; standard config without PoseidonGate rows: 80 routed wires, chunks of 8): the wire columns are read ONCE.
// The permutation argument takes them 8 at a time; a LookupTableGate slot is 3 consecutive columns and a LookupGate slot 2, so
// while a chunk of 8 sits in registers every slot that ENDS inside it is fed -- its leading columns are either in the same
// chunk or the last two of the previous one, which are carried over.  Nothing else changes: every accumulator takes exact
// integer sums (AlphaAcc), so the order in which the terms arrive does not show in the result.
template <bool HAS_POSEIDON, bool ONE_WALK = false>
__global__ __launch_bounds__(256, 4) void k_quotient(QuotientArgs a) {
    static_assert(!(HAS_POSEIDON && ONE_WALK), "the one-walk form rides on the ArithmeticGate layout of routed-only circuits");
    const u32 N = a.n << a.rate_bits;
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const u32 lde_bits = a.logn + a.rate_bits;
    // natural index i = rev(p); next = i + 2^rate_bits (mod N); its position
    const u32 i_nat = __brev(p) >> (32 - lde_bits);
    const u32 p_next = __brev((i_nat + (1u << a.rate_bits)) & (N - 1)) >> (32 - lde_bits);
    const u32 coset = i_nat & ((1u << a.rate_bits) - 1);
    const u64* cw = a.chal + (size_t)blockIdx.y * CH_WORDS;
    const u64* W = a.wires_lde + (size_t)blockIdx.y * a.wires_batch_stride + p;
    const u64* Zs = a.zs_lde + (size_t)blockIdx.y * a.zs_batch_stride + p;
    const u64* Zn = a.zs_lde + (size_t)blockIdx.y * a.zs_batch_stride + p_next;
    const u64* C = a.pre_lde + p;
    const u64* S = a.pre_lde + (size_t)a.ncc * N + p;
    const u64 x = a.xs[p], l0 = a.l0[p];
    AlphaAcc A;
    A.init(a.apow + (size_t)blockIdx.y * 2 * APOW_STRIDE);
    // term index bases
    const u32 idx_pp = a.NC, nlk = a.nlp ? 4 + a.num_luts + 2 * a.nsldc : 0;
    const u32 idx_lk = idx_pp + a.NC * (a.npp + 1), idx_gate = idx_lk + a.NC * nlk;
    // gate filters (needed early: the arithmetic constraints ride on the permutation argument's walk over the wires)
    const u64* gc = C + (size_t)(a.nsel + a.nls) * N;
    const u64 c0 = gc[0], c1 = gc[(size_t)1 * N];
    u64 f_arith = 0, f_const = 0, f_pi = 0, f_pos = 0;
    for (u32 g = 0; g < a.num_gates; g++) {
        u32 kind = a.gate_kind[g];
        if (kind != p2::G_ARITHMETIC && kind != p2::G_CONSTANT && kind != p2::G_PUBLIC_INPUT && kind != p2::G_POSEIDON) continue;
        u64 s = C[(size_t)a.gate_sel[g] * N], filter = 1;
        for (u32 j = a.group_lo[g]; j < a.group_hi[g]; j++)
            if (j != g) filter = gl::mul(filter, gl::sub(j, s));
        if (a.nsel > 1) filter = gl::mul(filter, gl::sub(p2::UNUSED_SELECTOR, s));
        if (kind == p2::G_ARITHMETIC) f_arith = filter;
        if (kind == p2::G_CONSTANT) f_const = filter;
        if (kind == p2::G_PUBLIC_INPUT) f_pi = filter;
        if (kind == p2::G_POSEIDON) f_pos = filter;
    }
    for (u32 i = 0; i < a.NC; i++) A.add(i, gl::mul(l0, gl::sub(Zs[(size_t)i * N], 1)));
    // ---- lookup argument: state shared by the two forms of the walk
    const u64* sel = C + (size_t)a.nsel * N;  // TransSre, TransLdc, InitSre, LastLdc, StartEnd..
    u64 s_sre = 0, s_ldc = 0;
    const u32 lk0 = a.NC * (1 + a.npp);
    const u64 *lz[2] = {Zs, Zs}, *lzn[2] = {Zn, Zn};
    u64 dA[2] = {0, 0}, dB[2] = {0, 0}, dAl[2] = {0, 0}, dD[2] = {0, 0}, cur[2] = {0, 0};
    u64 tprod[2] = {1, 1}, tsum[2] = {0, 0}, lprod[2] = {1, 1}, lsum[2] = {0, 0};
    u32 tpoly = 0, tin = 0, lpoly = 0, lin = 0;
    const u32 lu_deg = a.qdf - 1;
    if (a.nlp) {
        s_sre = sel[0];
        s_ldc = sel[(size_t)1 * N];
        const u64 s_init = sel[(size_t)2 * N], s_last = sel[(size_t)3 * N];
        for (u32 i = 0; i < 2; i++) {
            const u64* d = cw + CH_DELTAS + 4 * i;
            dA[i] = d[0];
            dB[i] = d[1];
            dAl[i] = d[2];
            dD[i] = d[3];
            lz[i] = Zs + (size_t)(lk0 + i * a.nlp) * N;
            lzn[i] = Zn + (size_t)(lk0 + i * a.nlp) * N;
            const u64 z_re = lz[i][0];
            const u32 t0 = idx_lk + i * nlk;
            A.add(t0 + 0, gl::mul(s_last, lz[i][(size_t)a.nsldc * N]));
            A.add(t0 + 1, gl::mul(s_init, lz[i][(size_t)1 * N]));
            A.add(t0 + 2, gl::mul(s_init, z_re));
            const u64* zv = a.zs_values + (size_t)blockIdx.y * a.zs_values_batch_stride + (size_t)(lk0 + i * a.nlp) * a.n;
            for (u32 l = 0; l < a.num_luts; l++) A.add(t0 + 3 + l, gl::mul(sel[(size_t)(4 + l) * N], gl::sub(z_re, zv[a.lut_last_row[l]])));
            cur[i] = lzn[i][0];
        }
    }
    // LookupTableGate view (slots (inp, out, mult): RE Horner and the Sum transition of each partial poly) and LookupGate view
    // (slots (inp, out): the LDC transition of each partial poly)
    auto lut_slot = [&](u32 s_, u64 win, u64 wout, u64 wm) {
        for (u32 i = 0; i < 2; i++) {
            cur[i] = gl::mul_add(cur[i], dD[i], gl::mul_add_nc(dB[i], wout, win));
            const u64 f = gl::sub(dAl[i], gl::mul_add(dA[i], wout, win));
            tsum[i] = gl::mul_add(tsum[i], f, gl::mul_nc(wm, tprod[i]));  // sum' = sum*f + mult*prod
            tprod[i] = gl::mul(tprod[i], f);
        }
        if (++tin == a.lut_deg || s_ + 1 == p2::LUT_SLOTS) {
            for (u32 i = 0; i < 2; i++) {
                u64 prev = tpoly == 0 ? lzn[i][(size_t)a.nsldc * N] : lz[i][(size_t)tpoly * N];
                u64 diff = gl::sub(lz[i][(size_t)(1 + tpoly) * N], prev);
                A.add(idx_lk + i * nlk + 4 + a.num_luts + 2 * tpoly, gl::mul(s_sre, gl::sub(gl::mul(tprod[i], diff), tsum[i])));
                tprod[i] = 1;
                tsum[i] = 0;
            }
            tpoly++;
            tin = 0;
        }
    };
    auto lu_slot = [&](u32 s_, u64 win, u64 wout) {
        for (u32 i = 0; i < 2; i++) {
            const u64 f = gl::sub(dAl[i], gl::mul_add(dA[i], wout, win));
            lsum[i] = gl::mul_add(lsum[i], f, lprod[i]);
            lprod[i] = gl::mul(lprod[i], f);
        }
        if (++lin == lu_deg || s_ + 1 == p2::LU_SLOTS) {
            for (u32 i = 0; i < 2; i++) {
                u64 prev = lpoly == 0 ? lzn[i][(size_t)a.nsldc * N] : lz[i][(size_t)lpoly * N];
                u64 diff = gl::sub(lz[i][(size_t)(1 + lpoly) * N], prev);
                A.add(idx_lk + i * nlk + 4 + a.num_luts + 2 * lpoly + 1, gl::mul(s_ldc, gl::add(gl::mul(lprod[i], diff), lsum[i])));
                lprod[i] = 1;
                lsum[i] = 0;
            }
            lpoly++;
            lin = 0;
        }
    };
    static_assert(p2::LUT_SLOTS == 26 && p2::LU_SLOTS == 40, "both walks assume 26 table slots (78 columns) and 40 looking slots (80 columns)");
    {  // permutation argument, both challenges per loaded wire
        const u64 b0 = cw[CH_BETAS], b1 = cw[CH_BETAS + 1], g0 = cw[CH_GAMMAS], g1 = cw[CH_GAMMAS + 1];
        const u64 bx0 = gl::mul(b0, x), bx1 = gl::mul(b1, x);
        u64 p6 = 0, p7 = 0;  // ONE_WALK: columns 6 and 7 of the previous chunk
        for (u32 chunk = 0; chunk <= a.npp; chunk++) {
            u64 num0 = 1, den0 = 1, num1 = 1, den1 = 1;
            u32 j1 = min(a.R, (chunk + 1) * a.qdf);
            u64 w8[8];
            if (ONE_WALK) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const u32 j = 8 * chunk + k;
                    const u64 wv = W[(size_t)j * N], sg = S[(size_t)j * N], kj = a.k_is[j];
                    w8[k] = wv;
                    // w + gamma once per challenge, the beta term as a fused multiply-add whose (non-canonical) result goes
                    // straight into the running product: 2 adds + 4 fused ops + 4 products instead of 8 adds + 8 products
                    // (one challenge after the other: with both sums live the kernel spills at its 128-register budget)
                    {
                        const u64 t = gl::add(wv, g0);
                        num0 = gl::mul(num0, gl::mul_add_nc(bx0, kj, t));
                        den0 = gl::mul(den0, gl::mul_add_nc(b0, sg, t));
                    }
                    {
                        const u64 t = gl::add(wv, g1);
                        num1 = gl::mul(num1, gl::mul_add_nc(bx1, kj, t));
                        den1 = gl::mul(den1, gl::mul_add_nc(b1, sg, t));
                    }
                }
            } else {
                for (u32 j = chunk * a.qdf; j < j1; j++) {
                    const u64 wv = W[(size_t)j * N], sg = S[(size_t)j * N], kj = a.k_is[j];
                    if (!HAS_POSEIDON) w8[(j - chunk * a.qdf) & 7] = wv;
                    num0 = gl::mul(num0, gl::add(gl::add(wv, gl::mul(bx0, kj)), g0));
                    den0 = gl::mul(den0, gl::add(gl::add(wv, gl::mul(b0, sg)), g0));
                    num1 = gl::mul(num1, gl::add(gl::add(wv, gl::mul(bx1, kj)), g1));
                    den1 = gl::mul(den1, gl::add(gl::add(wv, gl::mul(b1, sg)), g1));
                }
            }
            if (!HAS_POSEIDON && f_arith && a.qdf == 8) {
                // ArithmeticGate ops 2*chunk and 2*chunk + 1 occupy exactly these 8 wires: out - (c0 m0 m1 + c1 addend)
                for (u32 h = 0; h < 2; h++) {
                    const u64 m0 = w8[4 * h], m1 = w8[4 * h + 1], ad = w8[4 * h + 2], o = w8[4 * h + 3];
                    A.add(idx_gate + 2 * chunk + h, gl::mul(f_arith, gl::sub(o, gl::mul_add(gl::mul_nc(m0, m1), c0, gl::mul_nc(ad, c1)))));
                }
            }
            for (u32 i = 0; i < 2; i++) {
                u64 prev = chunk == 0 ? Zs[(size_t)i * N] : Zs[(size_t)(a.NC + i * a.npp + chunk - 1) * N];
                u64 next = chunk == a.npp ? Zn[(size_t)i * N] : Zs[(size_t)(a.NC + i * a.npp + chunk) * N];
                A.add(idx_pp + i * (a.npp + 1) + chunk, gl::sub(gl::mul(prev, i ? num1 : num0), gl::mul(next, i ? den1 : den0)));
            }
            if (ONE_WALK && a.nlp) {
                // the slots that end in this chunk: column c = 8 chunk + k closes table slot c / 3 when c % 3 == 2 (c < 78) and
                // looking slot c / 2 when c is odd; 8 = 2 (mod 3), so the table-slot pattern depends on chunk mod 3 (uniform)
                const u32 c3 = (2 * chunk) % 3;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const u32 c = 8 * chunk + k;
                    const u64 wm1 = k >= 1 ? w8[k >= 1 ? k - 1 : 0] : p7, wm2 = k >= 2 ? w8[k >= 2 ? k - 2 : 0] : (k == 1 ? p7 : p6);
                    if ((c3 + k) % 3 == 2 && c < 3 * p2::LUT_SLOTS) lut_slot(c / 3, wm2, wm1, w8[k]);
                    if (k & 1) lu_slot(c >> 1, wm1, w8[k]);
                }
                p6 = w8[6];
                p7 = w8[7];
            }
        }
    }
    if (a.nlp) {
        if (!ONE_WALK) {
            // both views in ONE walk of their own over the wire columns: six columns are two table slots and three looking slots
            // (round 1 walked the columns once per view)
            for (u32 g = 0; g < 13; g++) {
                u64 w[6];
#pragma unroll
                for (int k = 0; k < 6; k++) w[k] = W[(size_t)(6 * g + k) * N];
                lut_slot(2 * g, w[0], w[1], w[2]);
                lut_slot(2 * g + 1, w[3], w[4], w[5]);
                lu_slot(3 * g, w[0], w[1]);
                lu_slot(3 * g + 1, w[2], w[3]);
                lu_slot(3 * g + 2, w[4], w[5]);
            }
            lu_slot(39, W[(size_t)78 * N], W[(size_t)79 * N]);
        }
        for (u32 i = 0; i < 2; i++) A.add(idx_lk + i * nlk + 3 + a.num_luts, gl::mul(s_sre, gl::sub(lz[i][0], cur[i])));
    }
    // gate constraints: constraint slot k collects every gate's k-th constraint times the gate's filter
    {
        // k-th constraint of every gate other than PoseidonGate, already multiplied by the gate's filter
        auto other_gates = [&](u32 k) -> u64 {
            u64 term = 0;
            u64 w0 = 0;
            if ((HAS_POSEIDON || a.qdf != 8) && k < p2::ARITH_OPS && f_arith) {  // otherwise already added during the permutation walk
                u64 m0 = W[(size_t)(4 * k) * N], m1 = W[(size_t)(4 * k + 1) * N], ad = W[(size_t)(4 * k + 2) * N], o = W[(size_t)(4 * k + 3) * N];
                term = gl::mul(f_arith, gl::sub(o, gl::add(gl::mul(gl::mul(m0, m1), c0), gl::mul(ad, c1))));
            }
            if (k < 4 && (f_const || f_pi)) w0 = W[(size_t)k * N];
            if (k < 2 && f_const) term = gl::add(term, gl::mul(f_const, gl::sub(k == 0 ? c0 : c1, w0)));
            if (k < 4 && f_pi) term = gl::add(term, gl::mul(f_pi, w0));
            return term;
        };
        if (HAS_POSEIDON) {
            p2::poseidon_gate_constraints<p2::FBase>([&](u32 i) { return W[(size_t)i * N]; },
                                                     [&](int k, u64 cst) { A.add(idx_gate + (u32)k, gl::add(gl::mul(f_pos, cst), other_gates((u32)k))); });
        } else {
            const u32 kmax = a.qdf == 8 ? min(a.num_gate_constraints, 4u) : a.num_gate_constraints;  // what is left: ConstantGate / PublicInputGate
            for (u32 k = 0; k < kmax; k++) A.add(idx_gate + k, other_gates(k));
        }
    }
    const u64 zi = a.zh_inv[coset];
    u64* out = a.out + (size_t)blockIdx.y * a.out_batch_stride + p;
    out[0] = gl::mul(A.acc[0].reduce(), zi);
    out[(size_t)N] = gl::mul(A.acc[1].reduce(), zi);
}

// ------------------------------------------------------------------------------------------- openings
// pows[k][i] = z_k^i (extension), k = 0: zeta, 1: g*zeta, 2: 1/zeta, 3: 1/(g*zeta).  layout [k][2][n] (c0 | c1)
__global__ void k_zeta_pows(const u64* chal, u64* pows, size_t pows_batch_stride, u32 n, u64 g) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64* cw = chal + (size_t)blockIdx.y * CH_WORDS;
    E2 z = gl::e2(cw[CH_ZETA], cw[CH_ZETA + 1]);
    u32 k = blockIdx.z;
    if (k & 1) z = gl::mul(z, g);
    if (k & 2) z = gl::inv(z);
    E2 r = gl::pow(z, i);
    u64* o = pows + (size_t)blockIdx.y * pows_batch_stride + (size_t)k * 2 * n;
    o[i] = r.a;
    o[n + i] = r.b;
}
// out[col] = sum_i coeffs[col][i] * pw[i]   (extension result), one workgroup per (column, proof)
__global__ __launch_bounds__(256) void k_eval_polys(const u64* __restrict__ coeffs, size_t coeffs_batch_stride, const u64* __restrict__ pw /*[2][n]*/,
                                                     size_t pw_batch_stride, u32 n, u64* __restrict__ out /*[cols][2]*/, size_t out_batch_stride) {
    __shared__ u64 la[256], lb[256];
    const u64* c = coeffs + (size_t)blockIdx.y * coeffs_batch_stride + (size_t)blockIdx.x * n;
    const u64* pa = pw + (size_t)blockIdx.y * pw_batch_stride;
    u64 sa = 0, sb = 0;
    u32 i = threadIdx.x;
    // four rows per trip, their twelve loads issued before the first product (loads_issued, kernels.h)
    for (; i + 3 * blockDim.x < n; i += 4 * blockDim.x) {
        u64 v[4], wa[4], wb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = c[i + k * blockDim.x];
            wa[k] = pa[i + k * blockDim.x];
            wb[k] = pa[n + i + k * blockDim.x];
        }
        loads_issued();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sa = gl::add(sa, gl::mul(v[k], wa[k]));
            sb = gl::add(sb, gl::mul(v[k], wb[k]));
        }
    }
    for (; i < n; i += blockDim.x) {
        u64 v = c[i];
        sa = gl::add(sa, gl::mul(v, pa[i]));
        sb = gl::add(sb, gl::mul(v, pa[n + i]));
    }
    la[threadIdx.x] = sa;
    lb[threadIdx.x] = sb;
    __syncthreads();
    for (u32 off = blockDim.x / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            la[threadIdx.x] = gl::add(la[threadIdx.x], la[threadIdx.x + off]);
            lb[threadIdx.x] = gl::add(lb[threadIdx.x], lb[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        u64* o = out + (size_t)blockIdx.y * out_batch_stride + 2 * (size_t)blockIdx.x;
        o[0] = la[0];
        o[1] = lb[0];
    }
}

// The same for a LIST of polynomials from several oracles in one launch (round 2 launched the kernel above once per oracle and
// point: preprocessed, wires, Z at zeta, Z at g zeta, quotient): entry e = column `col` of the coefficient matrix `base`,
// evaluated with power table pw_k (0: zeta, 1: g zeta), result to extension slot `out`.
struct EvalRef {
    const u64* base;      // coefficient matrix of the oracle (proof 0)
    size_t batch_stride;  // 0 for the shared preprocessed oracle
    u32 col, pw_k, out, pad;
};
__global__ __launch_bounds__(256) void k_eval_polys_refs(const EvalRef* __restrict__ refs, const u64* __restrict__ pows /*[4][2][n]*/, size_t pw_batch_stride, u32 n,
                                                          u64* __restrict__ ev, size_t ev_batch_stride) {
    __shared__ u64 la[256], lb[256];
    const EvalRef r = refs[blockIdx.x];
    const u64* c = r.base + (size_t)blockIdx.y * r.batch_stride + (size_t)r.col * n;
    const u64* pa = pows + (size_t)blockIdx.y * pw_batch_stride + (size_t)r.pw_k * 2 * n;
    u64 sa = 0, sb = 0;
    u32 i = threadIdx.x;
    // four rows per trip, their twelve loads issued before the first product (loads_issued, kernels.h)
    for (; i + 3 * blockDim.x < n; i += 4 * blockDim.x) {
        u64 v[4], wa[4], wb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = c[i + k * blockDim.x];
            wa[k] = pa[i + k * blockDim.x];
            wb[k] = pa[n + i + k * blockDim.x];
        }
        loads_issued();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sa = gl::add(sa, gl::mul(v[k], wa[k]));
            sb = gl::add(sb, gl::mul(v[k], wb[k]));
        }
    }
    for (; i < n; i += blockDim.x) {
        u64 v = c[i];
        sa = gl::add(sa, gl::mul(v, pa[i]));
        sb = gl::add(sb, gl::mul(v, pa[n + i]));
    }
    la[threadIdx.x] = sa;
    lb[threadIdx.x] = sb;
    __syncthreads();
    for (u32 off = blockDim.x / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            la[threadIdx.x] = gl::add(la[threadIdx.x], la[threadIdx.x + off]);
            lb[threadIdx.x] = gl::add(lb[threadIdx.x], lb[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        u64* o = ev + (size_t)blockIdx.y * ev_batch_stride + 2 * (size_t)r.out;
        o[0] = la[0];
        o[1] = lb[0];
    }
}

// ------------------------------------------------------------------------------------------- FRI
// Batch description: for each polynomial of a FRI batch, where its coefficient column lives.
struct PolyRef {
    const u64* base;      // coefficient matrix of the oracle (proof 0)
    size_t batch_stride;  // 0 for the shared preprocessed oracle
    u32 col;
    u32 pad;
};
// comp[b][k] = sum_j alpha^j f_{b,j}[k]   for the two FRI batches (b = 0: zeta, b = 1: g*zeta); output [b][2][n]
__global__ __launch_bounds__(256) void k_fri_compose(const PolyRef* __restrict__ polys, u32 n0, u32 n1, const u64* chal, u32 n, u64* comp, size_t comp_batch_stride) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const u64* cw = chal + (size_t)blockIdx.y * CH_WORDS;
    const E2 alpha = gl::e2(cw[CH_FRI_ALPHA], cw[CH_FRI_ALPHA + 1]);
    u64* o = comp + (size_t)blockIdx.y * comp_batch_stride;
    u32 off = 0;
    for (u32 b = 0; b < 2; b++) {
        u32 cnt = b == 0 ? n0 : n1;
        E2 acc = gl::e2(0, 0);
        for (u32 j = cnt; j-- > 0;) {  // Horner: acc = acc*alpha + f_j[k]
            const PolyRef pr = polys[off + j];
            u64 v = pr.base ? pr.base[(size_t)blockIdx.y * pr.batch_stride + (size_t)pr.col * n + k] : 0;
            acc = gl::mul(acc, alpha);
            acc.a = gl::add(acc.a, v);
        }
        o[(size_t)(2 * b) * n + k] = acc.a;
        o[(size_t)(2 * b + 1) * n + k] = acc.b;
        off += cnt;
    }
}
// Division of both compositions by (X - z_b), combination final = Q_0 * alpha^{n1} + Q_1.
//   Q[k-1] = sum_{i >= k} comp[i] z^(i-k) = z^-k * sum_{i>=k} comp[i] z^i         (remainder dropped, Q[n-1] = 0)
// A workgroup per (proof, segment of the coefficients); extension-field suffix scan.  Short polynomials are one segment; long ones
// (n > 2^14, where a chunk holds few proofs) are cut into gridDim.y segments counted from the TOP coefficient down, whose sums
// k_fri_seg_sums forms first: a segment's sweep starts from the sum of the segments above it.
static const u32 FRI_MAX_SEGS = 64;
__global__ __launch_bounds__(1024) void k_fri_seg_sums(const u64* comp, size_t comp_batch_stride, const u64* pows, size_t pows_batch_stride, u32 n,
                                                        u64* seg_sum /*[proof][2][segs][2]*/) {
    __shared__ u64 la[1024], lb[1024];
    const u32 t = threadIdx.x, segs = gridDim.y, seg = blockIdx.y, seg_len = n / segs;
    for (u32 b = 0; b < 2; b++) {
        const u64* c = comp + (size_t)blockIdx.x * comp_batch_stride + (size_t)(2 * b) * n;
        const u64* zp = pows + (size_t)blockIdx.x * pows_batch_stride + (size_t)b * 2 * n;
        E2 acc = gl::e2(0, 0);
        for (u32 d = seg * seg_len + t; d < (seg + 1) * seg_len; d += blockDim.x) {
            const u32 i = n - 1 - d;
            acc = gl::add(acc, gl::mul(gl::e2(c[i], c[n + i]), gl::e2(zp[i], zp[n + i])));
        }
        __syncthreads();
        la[t] = acc.a;
        lb[t] = acc.b;
        __syncthreads();
        for (u32 off = blockDim.x / 2; off > 0; off >>= 1) {
            if (t < off) {
                la[t] = gl::add(la[t], la[t + off]);
                lb[t] = gl::add(lb[t], lb[t + off]);
            }
            __syncthreads();
        }
        if (t == 0) {
            u64* o = seg_sum + (((size_t)blockIdx.x * 2 + b) * segs + seg) * 2;
            o[0] = la[0];
            o[1] = lb[0];
        }
    }
}
__global__ __launch_bounds__(1024) void k_fri_divide(const u64* comp, size_t comp_batch_stride, const u64* pows, size_t pows_batch_stride, const u64* chal,
                                                      u32 n, u32 n1, u64* final_poly /*[2][n]*/, size_t final_batch_stride, const u64* seg_sum /* null: one segment */) {
    __shared__ u64 la[1024], lb[1024];
    __shared__ u64 s_carry[2];
    const u32 t = threadIdx.x, segs = gridDim.y, seg = blockIdx.y, seg_len = n / segs;
    const u64* cw = chal + (size_t)blockIdx.x * CH_WORDS;
    const E2 alpha = gl::e2(cw[CH_FRI_ALPHA], cw[CH_FRI_ALPHA + 1]);
    const E2 shift = gl::pow(alpha, n1);
    u64* fo = final_poly + (size_t)blockIdx.x * final_batch_stride;
    // Synthetic division by (X - z) is a suffix sum: Q[i] = z^-(i+1) * sum_{i' > i} comp[i'] z^i'.  The coefficients are
    // swept from the top 1024 at a time (thread t takes i = n-1-(base+t): coalesced), a workgroup scan per sweep, the
    // running sum carried across sweeps.
    for (u32 b = 0; b < 2; b++) {
        const u64* c = comp + (size_t)blockIdx.x * comp_batch_stride + (size_t)(2 * b) * n;
        const u64* zp = pows + (size_t)blockIdx.x * pows_batch_stride + (size_t)b * 2 * n;         // z^i
        const u64* zi = pows + (size_t)blockIdx.x * pows_batch_stride + (size_t)(2 + b) * 2 * n;   // z^-i
        __syncthreads();  // the previous pass is done with s_carry
        if (t == 0) {
            E2 c0 = gl::e2(0, 0);
            for (u32 s_ = 0; s_ < seg; s_++) {
                const u64* o = seg_sum + (((size_t)blockIdx.x * 2 + b) * segs + s_) * 2;
                c0 = gl::add(c0, gl::e2(o[0], o[1]));
            }
            s_carry[0] = c0.a;
            s_carry[1] = c0.b;
        }
        __syncthreads();
        for (u32 base = seg * seg_len; base < (seg + 1) * seg_len; base += blockDim.x) {
            const u32 d = base + t;
            const bool live = d < (seg + 1) * seg_len;
            const u32 i = live ? n - 1 - d : 0;
            E2 term = gl::e2(0, 0);
            if (live) term = gl::mul(gl::e2(c[i], c[n + i]), gl::e2(zp[i], zp[n + i]));
            const E2 carry = gl::e2(s_carry[0], s_carry[1]);
            la[t] = term.a;
            lb[t] = term.b;
            __syncthreads();
            for (u32 off = 1; off < blockDim.x; off <<= 1) {
                u64 xa = la[t], xb = lb[t];
                u64 ya = t >= off ? la[t - off] : 0, yb = t >= off ? lb[t - off] : 0;
                __syncthreads();
                la[t] = gl::add(xa, ya);
                lb[t] = gl::add(xb, yb);
                __syncthreads();
            }
            if (live) {
                E2 acc = gl::add(carry, t == 0 ? gl::e2(0, 0) : gl::e2(la[t - 1], lb[t - 1]));  // sum over i' > i
                E2 q = gl::e2(0, 0);
                if (i + 1 < n) q = gl::mul(acc, gl::e2(zi[i + 1], zi[n + i + 1]));
                if (b == 0) {
                    E2 s = gl::mul(q, shift);
                    fo[i] = s.a;
                    fo[n + i] = s.b;
                } else {
                    fo[i] = gl::add(fo[i], q.a);
                    fo[n + i] = gl::add(fo[n + i], q.b);
                }
            }
            __syncthreads();
            if (t == blockDim.x - 1) {
                E2 tot = gl::add(carry, gl::e2(la[t], lb[t]));
                s_carry[0] = tot.a;
                s_carry[1] = tot.b;
            }
            __syncthreads();
        }
    }
}
// fold coefficients by beta: out[k] = sum_{i<arity} beta^i in[arity*k + i];  in/out: [2][len] component columns
__global__ void k_fri_fold(const u64* in, size_t in_len, size_t in_batch_stride, u64* out, size_t out_len, size_t out_batch_stride, const u64* chal,
                           u32 round, u32 arity) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= out_len) return;
    const u64* cw = chal + (size_t)blockIdx.y * CH_WORDS;
    const E2 beta = gl::e2(cw[CH_FRI_BETAS + 2 * round], cw[CH_FRI_BETAS + 2 * round + 1]);
    const u64* a = in + (size_t)blockIdx.y * in_batch_stride;
    E2 acc = gl::e2(0, 0);
    for (u32 i = arity; i-- > 0;) {
        acc = gl::mul(acc, beta);
        acc.a = gl::add(acc.a, a[(size_t)arity * k + i]);
        acc.b = gl::add(acc.b, a[in_len + (size_t)arity * k + i]);
    }
    u64* o = out + (size_t)blockIdx.y * out_batch_stride;
    o[k] = acc.a;
    o[out_len + k] = acc.b;
}

// ------------------------------------------------------------------------------------------- proof assembly
__device__ __forceinline__ void store_u64_bytes(uint8_t* p, u64 v) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}
// copy `count` u64 words (src stride per proof) to byte offset `dst_off` of every proof
__global__ void k_proof_copy(const u64* src, size_t src_batch_stride, u32 count, uint8_t* proofs, size_t proof_bytes, size_t dst_off) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    store_u64_bytes(proofs + (size_t)blockIdx.y * proof_bytes + dst_off + 8 * (size_t)i, src[(size_t)blockIdx.y * src_batch_stride + i]);
}
// strided variant: word i comes from src[i * elem_stride]  (component-column -> interleaved extension elements)
__global__ void k_proof_copy_ext(const u64* src, size_t src_batch_stride, size_t comp_stride, u32 count, uint8_t* proofs, size_t proof_bytes, size_t dst_off) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64* s = src + (size_t)blockIdx.y * src_batch_stride;
    uint8_t* d = proofs + (size_t)blockIdx.y * proof_bytes + dst_off + 16 * (size_t)i;
    store_u64_bytes(d, s[i]);
    store_u64_bytes(d + 8, s[comp_stride + i]);
}

// dst[i] (extension) = src[map[i]]
__global__ void k_gather_ext(const u64* src, size_t src_batch_stride, const u32* map, u32 count, u64* dst, size_t dst_batch_stride) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64* s = src + (size_t)blockIdx.y * src_batch_stride + 2 * (size_t)map[i];
    u64* d = dst + (size_t)blockIdx.y * dst_batch_stride + 2 * (size_t)i;
    d[0] = s[0];
    d[1] = s[1];
}
__global__ void k_proof_gather_ext(const u64* src, size_t src_batch_stride, const u32* map, u32 count, uint8_t* proofs, size_t proof_bytes, size_t dst_off) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64* s = src + (size_t)blockIdx.y * src_batch_stride + 2 * (size_t)map[i];
    uint8_t* d = proofs + (size_t)blockIdx.y * proof_bytes + dst_off + 16 * (size_t)i;
    store_u64_bytes(d, s[0]);
    store_u64_bytes(d + 8, s[1]);
}
// Every fixed-position piece of the proof -- the caps, the opening set, the final polynomial, the PoW witness -- in ONE launch
// (round 2: nine launches of the three kernels above).  blockIdx.z = piece.
struct ProofSeg {
    const u64* src;
    const u32* map;       // kind 2: gather map
    size_t src_batch_stride, comp_stride, dst_off;
    u32 count, kind;      // 0: `count` words; 1: `count` extension elements from two component columns; 2: `count` gathered extension elements
};
struct ProofSegs {
    ProofSeg s[12];
    uint8_t* proofs;
    size_t proof_bytes;
};
__global__ void k_proof_segments(ProofSegs a) {
    const ProofSeg g = a.s[blockIdx.z];
    const u64* s = g.src + (size_t)blockIdx.y * g.src_batch_stride;
    uint8_t* d = a.proofs + (size_t)blockIdx.y * a.proof_bytes + g.dst_off;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < g.count; i += gridDim.x * blockDim.x) {
        if (g.kind == 0) {
            store_u64_bytes(d + 8 * (size_t)i, s[i]);
        } else if (g.kind == 1) {
            store_u64_bytes(d + 16 * (size_t)i, s[i]);
            store_u64_bytes(d + 16 * (size_t)i + 8, s[g.comp_stride + i]);
        } else {
            const u64* e = s + 2 * (size_t)g.map[i];
            store_u64_bytes(d + 16 * (size_t)i, e[0]);
            store_u64_bytes(d + 16 * (size_t)i + 8, e[1]);
        }
    }
}
// component columns [2][len] -> interleaved (c0, c1) pairs
__global__ void k_interleave_ext(const u64* src, size_t len, size_t src_batch_stride, u64* dst, size_t dst_batch_stride) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const u64* s = src + (size_t)blockIdx.y * src_batch_stride;
    u64* d = dst + (size_t)blockIdx.y * dst_batch_stride;
    d[2 * (size_t)i] = s[i];
    d[2 * (size_t)i + 1] = s[len + i];
}
// publish the per-proof status; a failed proof's slot is zeroed
__global__ void k_finish(const int* status, int* status_out, uint8_t* proofs, size_t proof_bytes, u32 batch) {
    u32 p = blockIdx.y;
    int st = status[p];
    if (blockIdx.x == 0 && threadIdx.x == 0) status_out[p] = st;
    if (st != 0) {
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < proof_bytes) proofs[(size_t)p * proof_bytes + i] = 0;
    }
}

struct QueryOracle {
    const u64* lde;        // [cols][N] column-major (active cols only)
    size_t lde_batch_stride;
    const u64* digests;    // level 0 .. cap-1 consecutively: level l starts at dig_level_off[l] (in u64)
    size_t dig_batch_stride;
    u32 cols, active_cols;
};
struct QueryArgs {
    QueryOracle oracles[4];
    u32 lde_bits, cap_height, num_queries;
    // FRI rounds
    u32 num_rounds;
    u32 arity_bits[8];
    const u64* fri_vals[8];  // [2][len_r] component columns, bit-reversed order
    size_t fri_vals_batch_stride[8];
    u32 fri_bits[8];  // log2(len_r)
    const u64* fri_digests[8];
    size_t fri_dig_batch_stride[8];
    const u64* chal;
    uint8_t* proofs;
    size_t proof_bytes, queries_off, query_bytes;
};
// digest level offsets inside one tree buffer: level l (l = 0 leaves) holds (num_leaves >> l) digests
__device__ __forceinline__ size_t level_off(u32 height_bits, u32 l) {
    // sum_{k<l} 4 * 2^(h-k) = 4 * (2^(h+1) - 2^(h-l+1))
    return 4 * (((size_t)2 << height_bits) - ((size_t)2 << (height_bits - l)));
}
// One workgroup per (query, proof): serialises FriQueryRound { initial_trees_proof, steps }.
__global__ __launch_bounds__(256) void k_write_queries(QueryArgs a) {
    const u32 q = blockIdx.x, p = blockIdx.y;
    const u64* cw = a.chal + (size_t)p * CH_WORDS;
    u32 x_index = (u32)cw[CH_QUERY + q];
    uint8_t* out = a.proofs + (size_t)p * a.proof_bytes + a.queries_off + (size_t)q * a.query_bytes;
    const u32 nsib = a.lde_bits - a.cap_height;
    for (int o = 0; o < 4; o++) {
        const QueryOracle& O = a.oracles[o];
        const size_t N = (size_t)1 << a.lde_bits;
        const u64* lde = O.lde + (size_t)p * O.lde_batch_stride;
        for (u32 c = threadIdx.x; c < O.cols; c += blockDim.x) store_u64_bytes(out + 8 * (size_t)c, c < O.active_cols ? lde[(size_t)c * N + x_index] : 0);
        out += 8 * (size_t)O.cols;
        if (threadIdx.x == 0) out[0] = (uint8_t)nsib;
        out += 1;
        const u64* dg = O.digests + (size_t)p * O.dig_batch_stride;
        for (u32 t = threadIdx.x; t < nsib * 4; t += blockDim.x) {
            u32 l = t >> 2;
            store_u64_bytes(out + 8 * (size_t)t, dg[level_off(a.lde_bits, l) + 4 * (size_t)((x_index >> l) ^ 1) + (t & 3)]);
        }
        out += 32 * (size_t)nsib;
    }
    for (u32 r = 0; r < a.num_rounds; r++) {
        const u32 ab = a.arity_bits[r], arity = 1u << ab;
        x_index >>= ab;
        const size_t len = (size_t)1 << a.fri_bits[r];
        const u64* v = a.fri_vals[r] + (size_t)p * a.fri_vals_batch_stride[r];
        for (u32 e = threadIdx.x; e < 2 * arity; e += blockDim.x)
            store_u64_bytes(out + 8 * (size_t)e, v[(size_t)(e & 1) * len + (size_t)x_index * arity + (e >> 1)]);
        out += 16 * (size_t)arity;
        const u32 tree_bits = a.fri_bits[r] - ab, ns = tree_bits - a.cap_height;
        if (threadIdx.x == 0) out[0] = (uint8_t)ns;
        out += 1;
        const u64* dg = a.fri_digests[r] + (size_t)p * a.fri_dig_batch_stride[r];
        for (u32 t = threadIdx.x; t < ns * 4; t += blockDim.x) {
            u32 l = t >> 2;
            store_u64_bytes(out + 8 * (size_t)t, dg[level_off(tree_bits, l) + 4 * (size_t)((x_index >> l) ^ 1) + (t & 3)]);
        }
        out += 32 * (size_t)ns;
    }
}

}  // namespace p2k

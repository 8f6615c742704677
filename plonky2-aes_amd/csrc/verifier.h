// Host verifier: the counterpart of `CircuitData::verify(proof)` (19 call sites in the reference, SURVEY.md
// A.2, e.g. aes-gcm/src/circuit_gcm.rs:782).  Upstream logic (plonky2 plonk/verifier.rs, plonk/vanishing_poly.rs,
// fri/verifier.rs; un-vendored crate, restated): re-derive the Fiat-Shamir challenges, check
// vanishing(zeta) = Z_H(zeta) * t(zeta) from the openings, then verify the FRI opening proof.
// ms-scale host code, exactly as in the reference; not part of the accelerated path.
#pragma once
#include <string>

#include "circuit.h"
#include "gl.h"
#include "poseidon_gate.h"

namespace p2 {

struct Hash4 {
    u64 e[4];
};
inline bool operator==(const Hash4& a, const Hash4& b) { return memcmp(a.e, b.e, 32) == 0; }

inline Hash4 h_hash_no_pad(const u64* in, size_t len) {
    u64 st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = std::min<size_t>(8, len - off);
        for (size_t i = 0; i < k; i++) st[i] = in[off + i];
        gl::poseidon(st);
    }
    Hash4 h;
    memcpy(h.e, st, 32);
    return h;
}
inline Hash4 h_hash_or_noop(const u64* in, size_t len) {
    if (len <= 4) {
        Hash4 h = {{0, 0, 0, 0}};
        for (size_t i = 0; i < len; i++) h.e[i] = in[i];
        return h;
    }
    return h_hash_no_pad(in, len);
}
inline Hash4 h_two_to_one(const Hash4& l, const Hash4& r) {
    Hash4 h;
    gl::two_to_one(l.e, r.e, h.e);
    return h;
}

// plonky2 iop::challenger::Challenger (duplex sponge, overwrite mode)
struct HostChallenger {
    u64 state[12];
    std::vector<u64> in, out;
    HostChallenger() { memset(state, 0, sizeof(state)); }
    void duplexing() {
        for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
        in.clear();
        gl::poseidon(state);
        out.assign(state, state + 8);
    }
    void observe(u64 x) {
        out.clear();
        in.push_back(x);
        if (in.size() == 8) duplexing();
    }
    void observe_hash(const Hash4& h) {
        for (int i = 0; i < 4; i++) observe(h.e[i]);
    }
    void observe_ext(gl::E2 x) {
        observe(x.a);
        observe(x.b);
    }
    u64 challenge() {
        if (!in.empty() || out.empty()) duplexing();
        u64 v = out.back();
        out.pop_back();
        return v;
    }
    gl::E2 ext_challenge() {
        u64 a = challenge();
        u64 b = challenge();
        return gl::e2(a, b);
    }
};

struct VerifierData {
    std::vector<Hash4> constants_sigmas_cap;
    Hash4 circuit_digest;
};

struct ProofReader {
    const uint8_t* p;
    size_t len, pos = 0;
    bool fail = false;
    u64 r64() {
        if (pos + 8 > len) {
            fail = true;
            return 0;
        }
        u64 v;
        memcpy(&v, p + pos, 8);
        pos += 8;
        return v;
    }
    uint8_t r8() {
        if (pos + 1 > len) {
            fail = true;
            return 0;
        }
        return p[pos++];
    }
    Hash4 hash() {
        Hash4 h;
        for (int i = 0; i < 4; i++) h.e[i] = r64();
        return h;
    }
    gl::E2 ext() {
        u64 a = r64();
        u64 b = r64();
        return gl::e2(a, b);
    }
    std::vector<Hash4> merkle_proof() {
        size_t k = r8();
        std::vector<Hash4> s(k);
        for (auto& h : s) h = hash();
        return s;
    }
};

inline bool verify_merkle_to_cap(const u64* leaf, size_t width, size_t index, const std::vector<Hash4>& cap, const std::vector<Hash4>& siblings) {
    Hash4 cur = h_hash_or_noop(leaf, width);
    for (auto& s : siblings) {
        cur = (index & 1) ? h_two_to_one(s, cur) : h_two_to_one(cur, s);
        index >>= 1;
    }
    return index < cap.size() && cur == cap[index];
}

// Returns "" on success, else the reason (anyhow-style error text).
inline std::string verify_proof(const Circuit& C, const VerifierData& vd, const uint8_t* bytes, size_t len) {
    using namespace gl;
    const size_t NC = C.cfg.num_challenges, R = C.cfg.num_routed_wires, npp = C.num_partial_products(), nlp = C.num_lookup_polys();
    const size_t nsldc = C.num_sldc_polys(), qdf = C.cfg.quotient_degree_factor, ncc = C.num_constants_cols();
    const size_t nsel = C.num_selectors(), nls = C.num_lookup_selectors;
    const size_t n = C.n(), lde_bits = C.degree_bits + C.cfg.rate_bits, N = (size_t)1 << lde_bits;
    const size_t cap_n = (size_t)1 << C.cfg.cap_height;
    const std::vector<u32> arities = C.reduction_arity_bits();
    // Proof shape (upstream: fri/validate_shape.rs + the fixed layout of ProofWithPublicInputs): the total length and the
    // depth of every Merkle path are functions of the circuit alone.  Without the depth check a prover could root subtrees
    // of different depths under different cap entries and pick, per query, which of them to open.  Every count below is
    // derived from the circuit and every Merkle path depth is checked after parsing, so together with the trailing-bytes
    // check the total length is pinned as well.
    ProofReader r{bytes, len};
    auto read_cap = [&]() {
        std::vector<Hash4> c(cap_n);
        for (auto& h : c) h = r.hash();
        return c;
    };
    auto read_exts = [&](size_t k) {
        std::vector<E2> v(k);
        for (auto& e : v) e = r.ext();
        return v;
    };
    auto wires_cap = read_cap(), zs_cap = read_cap(), quot_cap = read_cap();
    auto o_constants = read_exts(ncc), o_sigmas = read_exts(R), o_wires = read_exts(C.cfg.num_wires);
    // read_opening_set: ..., plonk_zs, plonk_zs_next, lookup_zs, lookup_zs_next, partial_products, quotient_polys
    auto o_zs = read_exts(NC), o_zs_next = read_exts(NC);
    auto o_lk = read_exts(NC * nlp), o_lk_next = read_exts(NC * nlp);
    auto o_pp = read_exts(NC * npp), o_quot = read_exts(NC * qdf);
    std::vector<std::vector<Hash4>> fri_caps;
    for (size_t i = 0; i < arities.size(); i++) fri_caps.push_back(read_cap());
    struct Query {
        std::vector<std::vector<u64>> init_evals;
        std::vector<std::vector<Hash4>> init_proofs;
        std::vector<std::vector<E2>> step_evals;
        std::vector<std::vector<Hash4>> step_proofs;
    };
    const size_t oracle_cols[4] = {C.num_preprocessed(), C.cfg.num_wires, C.num_zs_cols(), C.num_quotient_cols()};
    std::vector<Query> queries(C.cfg.num_query_rounds);
    for (auto& q : queries) {
        for (int o = 0; o < 4; o++) {
            std::vector<u64> ev(oracle_cols[o] + (o ? C.salt() : 0));  // blinded oracles carry SALT_SIZE extra leaf elements
            for (auto& v : ev) v = r.r64();
            q.init_evals.push_back(ev);
            q.init_proofs.push_back(r.merkle_proof());
        }
        for (size_t k = 0; k < arities.size(); k++) {
            q.step_evals.push_back(read_exts((size_t)1 << arities[k]));
            q.step_proofs.push_back(r.merkle_proof());
        }
    }
    size_t final_len = n;
    for (u32 a : arities) final_len >>= a;
    auto final_poly = read_exts(final_len);
    u64 pow_witness = r.r64();
    if (r.fail) return "proof truncated";
    if (r.pos != len) return "trailing bytes in proof";
    auto canonical = [&](u64 v) { return v < P; };
    for (auto* v : {&o_constants, &o_sigmas, &o_wires, &o_zs, &o_zs_next, &o_pp, &o_quot, &o_lk, &o_lk_next, &final_poly})
        for (auto& e : *v)
            if (!canonical(e.a) || !canonical(e.b)) return "non-canonical field element";
    if (!canonical(pow_witness)) return "non-canonical field element";
    auto canonical_hashes = [&](const std::vector<Hash4>& hs) {
        for (auto& h : hs)
            for (int i = 0; i < 4; i++)
                if (!canonical(h.e[i])) return false;
        return true;
    };
    if (!canonical_hashes(wires_cap) || !canonical_hashes(zs_cap) || !canonical_hashes(quot_cap)) return "non-canonical field element";
    for (auto& cap : fri_caps)
        if (!canonical_hashes(cap)) return "non-canonical field element";
    for (auto& q : queries) {
        for (int o = 0; o < 4; o++) {
            if (q.init_proofs[o].size() + C.cfg.cap_height != lde_bits) return "Merkle path of the wrong depth (initial tree).";
            if (!canonical_hashes(q.init_proofs[o])) return "non-canonical field element";
            for (u64 v : q.init_evals[o])
                if (!canonical(v)) return "non-canonical field element";
        }
        size_t bits = lde_bits;
        for (size_t k = 0; k < arities.size(); k++) {
            if (q.step_proofs[k].size() + C.cfg.cap_height + arities[k] != bits) return "Merkle path of the wrong depth (FRI round).";
            if (!canonical_hashes(q.step_proofs[k])) return "non-canonical field element";
            for (auto& e : q.step_evals[k])
                if (!canonical(e.a) || !canonical(e.b)) return "non-canonical field element";
            bits -= arities[k];
        }
    }

    // ---- challenges (plonk/get_challenges.rs)
    HostChallenger ch;
    ch.observe_hash(vd.circuit_digest);
    for (int i = 0; i < 4; i++) ch.observe(0);  // public_inputs_hash of zero public inputs
    for (auto& h : wires_cap) ch.observe_hash(h);
    std::vector<u64> betas, gammas, deltas, alphas;
    for (size_t i = 0; i < NC; i++) betas.push_back(ch.challenge());
    for (size_t i = 0; i < NC; i++) gammas.push_back(ch.challenge());
    if (nlp) {
        deltas = betas;
        deltas.insert(deltas.end(), gammas.begin(), gammas.end());
        for (size_t i = 0; i < 2 * NC; i++) deltas.push_back(ch.challenge());
    }
    for (auto& h : zs_cap) ch.observe_hash(h);
    for (size_t i = 0; i < NC; i++) alphas.push_back(ch.challenge());
    for (auto& h : quot_cap) ch.observe_hash(h);
    E2 zeta = ch.ext_challenge();
    std::vector<E2> batch0, batch1;
    for (auto* v : {&o_constants, &o_sigmas, &o_wires, &o_zs, &o_pp, &o_quot, &o_lk}) batch0.insert(batch0.end(), v->begin(), v->end());
    for (auto* v : {&o_zs_next, &o_lk_next}) batch1.insert(batch1.end(), v->begin(), v->end());
    for (auto& e : batch0) ch.observe_ext(e);
    for (auto& e : batch1) ch.observe_ext(e);
    E2 fri_alpha = ch.ext_challenge();
    std::vector<E2> fri_betas;
    for (auto& cap : fri_caps) {
        for (auto& h : cap) ch.observe_hash(h);
        fri_betas.push_back(ch.ext_challenge());
    }
    for (auto& e : final_poly) ch.observe_ext(e);
    ch.observe(pow_witness);
    u64 pow_response = ch.challenge();
    if ((pow_response >> (64 - C.cfg.pow_bits)) != 0) return "Invalid proof-of-work witness.";
    std::vector<size_t> query_idx;
    for (size_t i = 0; i < queries.size(); i++) query_idx.push_back((size_t)(ch.challenge() % N));

    // ---- vanishing polynomial at zeta (eval_vanishing_poly)
    E2 zeta_pow_n = exp_pow2(zeta, (int)C.degree_bits);
    E2 z_h_zeta = sub(zeta_pow_n, e2(1));
    if (eq(zeta_pow_n, e2(1))) return "Opening point is in the subgroup.";
    E2 l0 = mul(z_h_zeta, inv(mul(sub(zeta, e2(1)), (u64)n % P)));
    std::vector<E2> terms;
    {
        std::vector<E2> z1, ppt, lkt, gate(C.num_gate_constraints, e2(0));
        for (size_t i = 0; i < NC; i++) {
            z1.push_back(mul(l0, sub(o_zs[i], e2(1))));
            for (size_t chunk = 0; chunk * qdf < R; chunk++) {
                E2 num = e2(1), den = e2(1);
                for (size_t j = chunk * qdf; j < std::min(R, (chunk + 1) * qdf); j++) {
                    num = mul(num, add(add(o_wires[j], mul(zeta, mul(betas[i], C.k_is[j]))), e2(gammas[i])));
                    den = mul(den, add(add(o_wires[j], mul(o_sigmas[j], betas[i])), e2(gammas[i])));
                }
                E2 prev = chunk == 0 ? o_zs[i] : o_pp[i * npp + chunk - 1];
                E2 next = chunk == npp ? o_zs_next[i] : o_pp[i * npp + chunk];
                ppt.push_back(sub(mul(prev, num), mul(next, den)));
            }
            if (nlp) {
                const u64* d = &deltas[4 * i];
                const E2 *lz = &o_lk[i * nlp], *lzn = &o_lk_next[i * nlp];
                const E2* sel = &o_constants[nsel];
                const E2 *sl = lz + 1, *sln = lzn + 1;
                const size_t lu_deg = qdf - 1, lut_deg = C.lut_degree();
                E2 looked[26], looking[40], lookup[26];
                for (int s = 0; s < 26; s++) {
                    looked[s] = add(o_wires[3 * s], mul(o_wires[3 * s + 1], d[0]));
                    lookup[s] = add(o_wires[3 * s], mul(o_wires[3 * s + 1], d[1]));
                }
                for (int s = 0; s < 40; s++) looking[s] = add(o_wires[2 * s], mul(o_wires[2 * s + 1], d[0]));
                lkt.push_back(mul(sel[3], sl[nsldc - 1]));
                lkt.push_back(mul(sel[2], sl[0]));
                lkt.push_back(mul(sel[2], lz[0]));
                for (size_t l = 0; l < C.luts.size(); l++) {
                    size_t rows = (C.luts[l].size() + LUT_SLOTS - 1) / LUT_SLOTS, total = rows * LUT_SLOTS;
                    u64 acc = 0;
                    for (size_t k = 0; k < total; k++) {
                        u64 e = k < C.luts[l].size() ? add((u64)C.luts[l][k].first, mul(d[1], (u64)C.luts[l][k].second)) : 0;
                        acc = add(mul(acc, d[3]), e);
                    }
                    lkt.push_back(mul(sel[4 + l], sub(lz[0], e2(acc))));
                }
                E2 cur = lzn[0];
                for (int s = 0; s < 26; s++) cur = add(mul(cur, d[3]), lookup[s]);
                lkt.push_back(mul(sel[0], sub(lz[0], cur)));
                E2 alpha_e = e2(d[2]);
                for (size_t poly = 0; poly < nsldc; poly++) {
                    size_t a0 = poly * lut_deg, a1 = std::min<size_t>((poly + 1) * lut_deg, 26);
                    size_t b0 = poly * lu_deg, b1 = std::min<size_t>((poly + 1) * lu_deg, 40);
                    E2 lut_prod = e2(1), lu_prod = e2(1), lu_sum = e2(0), lut_sum_mul = e2(0);
                    for (size_t k = a0; k < a1; k++) lut_prod = mul(lut_prod, sub(alpha_e, looked[k]));
                    for (size_t k = b0; k < b1; k++) lu_prod = mul(lu_prod, sub(alpha_e, looking[k]));
                    for (size_t k = b0; k < b1; k++) {
                        E2 p = e2(1);
                        for (size_t m = b0; m < b1; m++)
                            if (m != k) p = mul(p, sub(alpha_e, looking[m]));
                        lu_sum = add(lu_sum, p);
                    }
                    for (size_t k = a0; k < a1; k++) {
                        E2 p = e2(1);
                        for (size_t m = a0; m < a1; m++)
                            if (m != k) p = mul(p, sub(alpha_e, looked[m]));
                        lut_sum_mul = add(lut_sum_mul, mul(o_wires[3 * k + 2], p));
                    }
                    E2 prev = poly == 0 ? sln[nsldc - 1] : sl[poly - 1];
                    E2 diff = sub(sl[poly], prev);
                    lkt.push_back(mul(sel[0], sub(mul(lut_prod, diff), lut_sum_mul)));
                    lkt.push_back(mul(sel[1], add(mul(lu_prod, diff), lu_sum)));
                }
            }
        }
        for (size_t gi = 0; gi < C.gates.size(); gi++) {
            u32 kind = C.gates[gi];
            if (gate_num_constraints(kind) == 0) continue;
            size_t si = C.selector_index[gi];
            E2 s = o_constants[si], filter = e2(1);
            for (u32 j = C.groups[si].first; j < C.groups[si].second; j++)
                if (j != gi) filter = mul(filter, sub(e2(j), s));
            if (nsel > 1) filter = mul(filter, sub(e2(UNUSED_SELECTOR), s));
            const E2* gc = &o_constants[nsel + nls];
            if (kind == G_ARITHMETIC) {
                for (u32 op = 0; op < ARITH_OPS; op++) {
                    E2 c = sub(o_wires[4 * op + 3], add(mul(mul(o_wires[4 * op], o_wires[4 * op + 1]), gc[0]), mul(o_wires[4 * op + 2], gc[1])));
                    gate[op] = add(gate[op], mul(filter, c));
                }
            } else if (kind == G_CONSTANT) {
                for (int k = 0; k < 2; k++) gate[k] = add(gate[k], mul(filter, sub(gc[k], o_wires[k])));
            } else if (kind == G_PUBLIC_INPUT) {
                for (int k = 0; k < 4; k++) gate[k] = add(gate[k], mul(filter, o_wires[k]));
            } else if (kind == G_POSEIDON) {
                poseidon_gate_constraints<FExt>([&](u32 i) { return o_wires[i]; },
                                                [&](int k, E2 cst) { gate[k] = add(gate[k], mul(filter, cst)); });
            }
        }
        for (auto* v : {&z1, &ppt, &lkt, &gate}) terms.insert(terms.end(), v->begin(), v->end());
    }
    for (size_t i = 0; i < NC; i++) {
        E2 vanishing = e2(0);
        for (size_t k = terms.size(); k-- > 0;) vanishing = add(mul(vanishing, alphas[i]), terms[k]);
        E2 t = e2(0);
        for (size_t c = qdf; c-- > 0;) t = add(mul(t, zeta_pow_n), o_quot[i * qdf + c]);
        if (!eq(vanishing, mul(z_h_zeta, t))) return "vanishing polynomial identity does not hold at zeta";
    }

    // ---- FRI (fri/verifier.rs)
    E2 g_zeta = mul(zeta, root_of_unity((int)C.degree_bits));
    auto reduce = [&](const std::vector<E2>& v) {
        E2 acc = e2(0);
        for (size_t k = v.size(); k-- > 0;) acc = add(mul(acc, fri_alpha), v[k]);
        return acc;
    };
    E2 red0 = reduce(batch0), red1 = reduce(batch1);
    const size_t nzpp = C.num_zs_pp();
    const std::vector<Hash4>* init_caps[4] = {&vd.constants_sigmas_cap, &wires_cap, &zs_cap, &quot_cap};
    u64 w_lde = root_of_unity((int)lde_bits);
    for (size_t qi = 0; qi < queries.size(); qi++) {
        const Query& q = queries[qi];
        size_t x_index = query_idx[qi];
        for (int o = 0; o < 4; o++)
            if (!verify_merkle_to_cap(q.init_evals[o].data(), q.init_evals[o].size(), x_index, *init_caps[o], q.init_proofs[o]))
                return "Invalid Merkle proof (initial tree).";
        u64 subgroup_x = mul(MULT_GEN, pow(w_lde, bitrev((u32)x_index, (int)lde_bits)));
        // fri_combine_initial
        std::vector<u64> e0, e1;
        // unsalted_eval: the salt elements at the end of a blinded leaf take no part in the combination
        const size_t zc_ = C.num_zs_cols();
        e0.insert(e0.end(), q.init_evals[0].begin(), q.init_evals[0].end());
        e0.insert(e0.end(), q.init_evals[1].begin(), q.init_evals[1].begin() + C.cfg.num_wires);
        e0.insert(e0.end(), q.init_evals[2].begin(), q.init_evals[2].begin() + nzpp);
        e0.insert(e0.end(), q.init_evals[3].begin(), q.init_evals[3].begin() + C.num_quotient_cols());
        e0.insert(e0.end(), q.init_evals[2].begin() + nzpp, q.init_evals[2].begin() + zc_);
        e1.insert(e1.end(), q.init_evals[2].begin(), q.init_evals[2].begin() + NC);
        e1.insert(e1.end(), q.init_evals[2].begin() + nzpp, q.init_evals[2].begin() + zc_);
        auto reduce_base = [&](const std::vector<u64>& v) {
            E2 acc = e2(0);
            for (size_t k = v.size(); k-- > 0;) acc = add(mul(acc, fri_alpha), e2(v[k]));
            return acc;
        };
        E2 sum = mul(sub(reduce_base(e0), red0), inv(sub(e2(subgroup_x), zeta)));
        sum = mul(sum, pow(fri_alpha, e1.size()));
        sum = add(sum, mul(sub(reduce_base(e1), red1), inv(sub(e2(subgroup_x), g_zeta))));
        E2 old_eval = sum;
        for (size_t k = 0; k < arities.size(); k++) {
            u32 ab = arities[k];
            size_t arity = (size_t)1 << ab;
            const std::vector<E2>& evals = q.step_evals[k];
            size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (!eq(evals[within], old_eval)) return "FRI fold consistency check failed.";
            // compute_evaluation: interpolate {(coset_start*g^k, evals_rev[k])} and evaluate at beta
            u64 g = root_of_unity((int)ab);
            u32 rev_within = bitrev((u32)within, (int)ab);
            u64 coset_start = mul(subgroup_x, pow(g, arity - rev_within));
            std::vector<u64> xs(arity);
            std::vector<E2> ys(arity);
            for (size_t t = 0; t < arity; t++) {
                xs[t] = mul(coset_start, pow(g, t));
                ys[t] = evals[bitrev((u32)t, (int)ab)];
            }
            E2 acc = e2(0);
            for (size_t t = 0; t < arity; t++) {
                E2 num = e2(1);
                u64 den = 1;
                for (size_t m = 0; m < arity; m++)
                    if (m != t) {
                        num = mul(num, sub(fri_betas[k], e2(xs[m])));
                        den = mul(den, sub(xs[t], xs[m]));
                    }
                acc = add(acc, mul(mul(ys[t], num), inv(den)));
            }
            old_eval = acc;
            std::vector<u64> flat;
            for (auto& e : evals) {
                flat.push_back(e.a);
                flat.push_back(e.b);
            }
            if (!verify_merkle_to_cap(flat.data(), flat.size(), coset_index, fri_caps[k], q.step_proofs[k])) return "Invalid Merkle proof (FRI round).";
            subgroup_x = exp_pow2(subgroup_x, (int)ab);
            x_index = coset_index;
        }
        E2 fe = e2(0);
        for (size_t k = final_poly.size(); k-- > 0;) fe = add(mul(fe, subgroup_x), final_poly[k]);
        if (!eq(fe, old_eval)) return "Final polynomial evaluation is invalid.";
    }
    return "";
}

}  // namespace p2

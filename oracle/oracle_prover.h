// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_field.h header).  PROOF BYTES: PARITY UNPINNED.
//
// CPU restatement of `CircuitData::prove(pw)` -- the 20 call sites of the reference (e.g.
// aes-gcm/src/circuit_gcm.rs:781, aes-gcm/examples/aes_gcm_128.rs:52) all land in the third-party crate
// plonky2 (git rev 109d517d..., Cargo.toml:12; not vendored), whose published protocol is restated here
// stage by stage: generate_partial_witness -> wires commit -> betas/gammas/deltas ->
// partial products, Z and lookup (RE / SLDC) polynomials -> alphas -> quotient -> zeta -> openings -> FRI
// (arity-16 folds, 16-bit grinding, 28 queries) -> serialisation.
//
// Two upstream sources of non-determinism are replaced by deterministic choices (documented in DESIGN.md):
//   * randomize_unused_pi_wires: the PublicInputGate's spare wires stay 0;
//   * fri_proof_of_work uses rayon find_any: here the SMALLEST satisfying witness is taken.
#pragma once
#include <algorithm>
#include <chrono>
#include <map>
#include <string>

#include "oracle_field.h"

namespace orc {

struct OConfig {
    u32 num_wires, num_routed_wires, num_constants, num_challenges, quotient_degree_factor, rate_bits, cap_height, pow_bits,
        num_query_rounds, arity_bits, final_poly_bits, zero_knowledge;
};
// Blinding PRF of the zk configuration, restated from its definition (product: csrc/circuit.h "Blinding randomness"):
// element `index` of (proof, domain) = Poseidon(key[0..4] | proof | domain | index / 8 | TAG | 0^4)[index % 8].
// Sequential callers hit the one-block cache.
struct ZkStream {
    u64 key[4], proof, domain;
    u64 block = ~0ull, out[8];
    ZkStream(const u64* k, u64 proof_, u64 domain_) : proof(proof_), domain(domain_) {
        for (int i = 0; i < 4; i++) key[i] = k[i];
    }
    u64 at(u64 index) {
        if ((index >> 3) != block) {
            block = index >> 3;
            u64 st[12] = {key[0], key[1], key[2], key[3], proof, domain, block, 0x7a6b5f626c696e64ull % MODULUS, 0, 0, 0, 0};
            poseidon_permute(st);
            for (int i = 0; i < 8; i++) out[i] = st[i];
        }
        return out[index & 7];
    }
};
struct OOp {
    u32 kind, out, a, b, c, aux;
    u64 k0, k1;
};
struct OLookupRows {
    u32 last_lu, last_lut, first_lut;
};
enum { GK_LOOKUP = 0, GK_LOOKUP_TABLE = 1, GK_NOOP = 2, GK_CONSTANT = 3, GK_PUBLIC_INPUT = 4, GK_ARITHMETIC = 5, GK_POSEIDON = 6 };

struct Reader {
    const uint8_t* p;
    size_t len, pos;
    void get(void* o, size_t n) {
        if (pos + n > len) throw std::runtime_error("oracle: blob truncated");
        memcpy(o, p + pos, n);
        pos += n;
    }
    u32 g32() { u32 v; get(&v, 4); return v; }
    u64 g64() { u64 v; get(&v, 8); return v; }
    template <class T>
    std::vector<T> arr() {
        u64 n = g64();
        if (n > (len - pos) / sizeof(T)) throw std::runtime_error("oracle: blob array truncated");
        std::vector<T> v(n);
        if (n) get(v.data(), n * sizeof(T));
        return v;
    }
};

// PolynomialBatch
struct Batch {
    size_t cols = 0;
    size_t width = 0;                       // leaf width = cols (+ 4 salt elements for a blinded oracle in zk mode)
    std::vector<std::vector<u64>> coeffs;  // [cols][n]
    std::vector<u64> lde;                  // [8n][cols], leaf index = bit-reversed domain index
    MerkleTree tree;
};

struct OCircuit {
    OConfig cfg;
    u32 degree_bits;
    std::vector<u32> gates, selector_index;
    std::vector<std::pair<u32, u32>> groups;
    u32 num_lookup_selectors, num_gate_constraints;
    std::vector<u64> constants, sigmas, k_is;
    std::vector<std::vector<std::pair<u16, u16>>> luts;
    std::vector<OLookupRows> lookup_rows;
    std::vector<u32> num_lookups;
    u32 num_slots;
    std::vector<OOp> ops;
    std::vector<u32> level_offsets;
    std::vector<int32_t> vt_slot, wire_slot;
    std::vector<u32> poseidon_rows, blind_rows;
    std::vector<std::pair<u32, u32>> blind_zrows;
    u64 zk_key[4] = {0, 0, 0, 0}, zk_proof = 0;  // set per proof by the caller (orc_set_zk / orc_set_zk_key)
    // fault injection (soundness tests): after witness generation add `fault_delta` to a slot (every copy of the value:
    // only gate / lookup constraints can notice) or to one wire cell (the permutation argument must notice)
    int fault_kind = 0;  // 0 none, 1 slot, 2 wire cell
    u64 fault_a = 0, fault_b = 0, fault_delta = 0;
    // derived
    size_t n;
    int lde_bits;
    Batch pre;
    Digest circuit_digest;
    std::vector<std::vector<int32_t>> lut_dense;  // per LUT: input value -> table index (or -1)

    size_t nsel() const { return groups.size(); }
    size_t ncc() const { return nsel() + num_lookup_selectors + cfg.num_constants; }
    size_t num_pp() const { return (cfg.num_routed_wires + cfg.quotient_degree_factor - 1) / cfg.quotient_degree_factor - 1; }
    size_t num_sldc() const { return luts.empty() ? 0 : (40 + cfg.quotient_degree_factor - 2) / (cfg.quotient_degree_factor - 1); }
    size_t num_lookup_polys() const { return luts.empty() ? 0 : num_sldc() + 1; }
    size_t lut_degree() const { return (26 + num_sldc() - 1) / num_sldc(); }
    std::vector<u32> arity_bits() const {
        std::vector<u32> r;
        u32 db = degree_bits;
        while (db > cfg.final_poly_bits && db + cfg.rate_bits - cfg.arity_bits >= cfg.cap_height) {
            r.push_back(cfg.arity_bits);
            db -= cfg.arity_bits;
        }
        return r;
    }
};

typedef std::map<std::string, std::vector<u64>> Trace;

// Wall-clock seconds the last prove() spent per stage (bench.py's cpu_baseline breakdown): 0 witness generation,
// 1 commitments (iFFT + coset FFT + Merkle trees of the wires / Z / quotient oracles), 2 partial products + lookup
// polynomials, 3 quotient evaluation, 4 openings, 5 FRI (batch combination, commit phase, PoW, queries, serialisation).
static thread_local double g_stage_seconds[6];
static inline double now_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct StageClock {
    double t = now_seconds();
    void lap(int stage) {
        double u = now_seconds();
        g_stage_seconds[stage] += u - t;
        t = u;
    }
};
// oracle_index 0 = constants|sigmas (never blinded); 1 wires, 2 zs/partial products/lookups, 3 quotient
static inline void commit_from_coeffs(const OCircuit& C, Batch& b, int oracle_index) {
    size_t n = C.n, N = n << C.cfg.rate_bits, cols = b.cols;
    size_t salt = (C.cfg.zero_knowledge && oracle_index > 0) ? 4 : 0;
    b.width = cols + salt;
    b.lde.assign(N * b.width, 0);
    {
        // column by column into a column-major scratch (the transform leaves position rev(k) = the storage order), then
        // transposed in blocks of rows so that every cache line of the row-major table is written once
        std::vector<u64> by_col(cols * N);
        const std::vector<u64> shift_pows = powers_of(GENERATOR, n);
        forward_stage_twiddles(C.lde_bits);
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t c = 0; c < cols; c++) coset_fft_bitrev_out(b.coeffs[c], shift_pows, C.lde_bits, &by_col[c * N]);
        const size_t RB = 64;
#pragma omp parallel for schedule(static)
        for (size_t r0 = 0; r0 < N; r0 += RB)
            for (size_t c = 0; c < cols; c++)
                for (size_t r = r0; r < r0 + RB && r < N; r++) b.lde[r * b.width + c] = by_col[c * N + r];
    }
    if (salt) {
        ZkStream zs(C.zk_key, C.zk_proof, 3 + oracle_index);
        for (size_t s = 0; s < salt; s++)
            for (size_t pos = 0; pos < N; pos++) b.lde[pos * b.width + cols + s] = zs.at(s * N + pos);
    }
    b.tree = build_merkle(b.lde.data(), N, b.width, C.cfg.cap_height);
}
static inline void commit_from_values(const OCircuit& C, Batch& b, const std::vector<std::vector<u64>>& values, int oracle_index) {
    b.cols = values.size();
    b.coeffs = values;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t c = 0; c < b.cols; c++) fft_inplace(b.coeffs[c].data(), C.degree_bits, true);
    commit_from_coeffs(C, b, oracle_index);
}

static inline OCircuit* load_circuit(const void* blob, size_t len) {
    Reader r{(const uint8_t*)blob, len, 0};
    char magic[8];
    r.get(magic, 8);
    if (memcmp(magic, "P2AESCIR", 8) != 0) throw std::runtime_error("oracle: bad magic");
    if (r.g32() != 4) throw std::runtime_error("oracle: bad version");
    OCircuit* C = new OCircuit();
    r.get(&C->cfg, sizeof(OConfig));
    C->degree_bits = r.g32();
    C->gates = r.arr<u32>();
    C->selector_index = r.arr<u32>();
    C->groups = r.arr<std::pair<u32, u32>>();
    C->num_lookup_selectors = r.g32();
    C->num_gate_constraints = r.g32();
    C->constants = r.arr<u64>();
    C->sigmas = r.arr<u64>();
    C->k_is = r.arr<u64>();
    u32 nl = r.g32();
    for (u32 i = 0; i < nl; i++) C->luts.push_back(r.arr<std::pair<u16, u16>>());
    C->lookup_rows = r.arr<OLookupRows>();
    C->num_lookups = r.arr<u32>();
    C->num_slots = r.g32();
    C->ops = r.arr<OOp>();
    C->level_offsets = r.arr<u32>();
    C->vt_slot = r.arr<int32_t>();
    C->wire_slot = r.arr<int32_t>();
    C->poseidon_rows = r.arr<u32>();
    C->blind_rows = r.arr<u32>();
    C->blind_zrows = r.arr<std::pair<u32, u32>>();
    C->n = (size_t)1 << C->degree_bits;
    C->lde_bits = C->degree_bits + C->cfg.rate_bits;
    for (auto& lut : C->luts) {
        std::vector<int32_t> d(65536, -1);
        for (size_t i = 0; i < lut.size(); i++)
            if (d[lut[i].first] < 0) d[lut[i].first] = (int32_t)i;
        C->lut_dense.push_back(d);
    }
    // constants_sigmas_commitment and circuit digest (CircuitBuilder::build tail)
    size_t n = C->n, ncc = C->ncc(), R = C->cfg.num_routed_wires;
    std::vector<std::vector<u64>> vals(ncc + R, std::vector<u64>(n));
    for (size_t c = 0; c < ncc; c++) memcpy(vals[c].data(), &C->constants[c * n], n * 8);
    for (size_t c = 0; c < R; c++) memcpy(vals[ncc + c].data(), &C->sigmas[c * n], n * 8);
    commit_from_values(*C, C->pre, vals, 0);
    std::vector<u64> parts;
    for (auto& d : C->pre.tree.cap())
        for (int i = 0; i < 4; i++) parts.push_back(d.e[i]);
    Digest ds = hash_pad(std::vector<u64>());  // domain separator = empty
    for (int i = 0; i < 4; i++) parts.push_back(ds.e[i]);
    parts.push_back(C->degree_bits);
    C->circuit_digest = hash_no_pad(parts.data(), parts.size());
    return C;
}

// ---------------------------------------------------------------------------------------------------------
// PoseidonGate (plonky2 gates/poseidon.rs, restated): wires 0..11 in, 12..23 out, 24 swap, 25..28 delta,
// 29..64 S-box inputs of full rounds 1..3, 65..86 S-box inputs of the 22 partial rounds, 87..134 S-box inputs of the
// last four full rounds.  `row` holds the 135 wire values; with `fill` the generator's outputs are written, otherwise
// the 123 constraint values are appended to `out`.
static inline void mds_apply(u64 st[12]) {
    u64 nx[12];
    for (int r = 0; r < 12; r++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)st[(i + r) % 12] * MDS_CIRC[i];
        acc += (u128)st[r] * MDS_DIAG[r];
        nx[r] = fred(acc);
    }
    memcpy(st, nx, sizeof(nx));
}
static inline void poseidon_gate_walk(u64* row, bool fill, std::vector<u64>* out) {
    auto check = [&](u64 computed, int wire) -> u64 {
        if (fill) {
            row[wire] = computed;
            return computed;
        }
        out->push_back(fsub(computed, row[wire]));
        return row[wire];
    };
    u64 swap = row[24];
    if (!fill) out->push_back(fmul(swap, fsub(swap, 1)));
    u64 st[12];
    for (int i = 0; i < 4; i++) {
        u64 d = check(fmul(swap, fsub(row[4 + i], row[i])), 25 + i);
        st[i] = fadd(row[i], d);
        st[4 + i] = fsub(row[4 + i], d);
    }
    for (int i = 8; i < 12; i++) st[i] = row[i];
    for (int round = 0; round < 30; round++) {
        for (int i = 0; i < 12; i++) st[i] = fadd(st[i], ROUND_CONSTANTS[12 * round + i]);
        if (round < 4) {
            if (round > 0)
                for (int i = 0; i < 12; i++) st[i] = check(st[i], 29 + 12 * (round - 1) + i);
            for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
        } else if (round < 26) {
            st[0] = pow7(check(st[0], 65 + (round - 4)));
        } else {
            for (int i = 0; i < 12; i++) st[i] = pow7(check(st[i], 87 + 12 * (round - 26) + i));
        }
        mds_apply(st);
    }
    for (int i = 0; i < 12; i++) check(st[i], 12 + i);
}

// ---------------------------------------------------------------------------------------------------------
// Witness generation.  status: 0 ok, 1 conflict / lookup miss, 2 some generator never ran (missing input).
static inline int generate_witness(const OCircuit& C, const u64* in_targets, const u64* in_values, size_t n_in,
                                   std::vector<std::vector<u64>>& wires) {
    const u64 UNSET = ~0ull;
    std::vector<u64> val(C.num_slots, UNSET);
    auto set = [&](u32 slot, u64 v) -> bool {
        if (val[slot] == UNSET) {
            val[slot] = v;
            return true;
        }
        return val[slot] == v;
    };
    for (size_t i = 0; i < n_in; i++) {
        u64 t = in_targets[i];
        int32_t slot = -1;
        if (t >> 63) {  // Target::Wire(row, column): routed wires only
            u64 row = (t & ~(1ull << 63)) >> 8, col = t & 0xFF;
            if (row < C.n && col < C.cfg.num_routed_wires) slot = C.wire_slot[col * C.n + row];
        } else if (t < C.vt_slot.size()) {
            slot = C.vt_slot[t];
        }
        if (slot < 0) return 1;
        if (in_values[i] >= MODULUS) return 1;
        if (!set((u32)slot, in_values[i])) return 1;
    }
    std::vector<std::vector<u64>> pos_rows(C.poseidon_rows.size());  // full 135-wire rows written by PoseidonGenerators
    for (const OOp& o : C.ops) {
        u64 r;
        if (o.kind == 5) {
            std::vector<u64> roww(C.cfg.num_wires, 0);
            for (u32 col = 0; col < 12; col++) roww[col] = val[C.wire_slot[col * C.n + o.a]];
            roww[24] = val[C.wire_slot[24 * C.n + o.a]];
            for (u32 col = 0; col < 12; col++)
                if (roww[col] == UNSET) return 2;
            if (roww[24] == UNSET) return 2;
            poseidon_gate_walk(roww.data(), true, nullptr);
            for (u32 col = 12; col < C.cfg.num_routed_wires; col++)
                if (col != 24 && !set((u32)C.wire_slot[col * C.n + o.a], roww[col])) return 1;
            pos_rows[o.aux] = roww;
            continue;
        }
        switch (o.kind) {
            case 0: {  // arith
                u64 a = val[o.a], b = val[o.b], c = val[o.c];
                if (a == UNSET || b == UNSET || c == UNSET) return 2;
                r = fadd(fmul(fmul(a, b), o.k0), fmul(c, o.k1));
                break;
            }
            case 1:
                r = o.k0;
                break;
            case 2: {  // lookup
                u64 a = val[o.a];
                if (a == UNSET) return 2;
                if (a >= 65536 || C.lut_dense[o.aux][a] < 0) return 1;
                r = C.luts[o.aux][C.lut_dense[o.aux][a]].second;
                break;
            }
            case 3:
            case 4: {
                u64 a = val[o.a], b = val[o.b];
                if (a == UNSET || b == UNSET) return 2;
                if (o.kind == 3)
                    r = a == b ? 1 : 0;
                else
                    r = a == b ? 0 : finv(fsub(a, b));
                break;
            }
            default:
                return 1;
        }
        if (!set(o.out, r)) return 1;
    }
    size_t n = C.n, R = C.cfg.num_routed_wires;
    if (C.fault_kind == 1 && C.fault_a < C.num_slots && val[C.fault_a] != UNSET) val[C.fault_a] = fadd(val[C.fault_a], C.fault_delta);
    wires.assign(C.cfg.num_wires, std::vector<u64>(n, 0));
    for (size_t c = 0; c < R; c++)
        for (size_t row = 0; row < n; row++) {
            int32_t s = C.wire_slot[c * n + row];
            if (s >= 0) {
                if (val[s] == UNSET) return 2;
                wires[c][row] = val[s];
            }
        }
    // zk blinding rows (RandomValueGenerators of blind_and_pad)
    ZkStream zrow(C.zk_key, C.zk_proof, 1), zzrow(C.zk_key, C.zk_proof, 2);
    for (size_t k = 0; k < C.blind_rows.size(); k++)
        for (size_t c = 0; c < C.cfg.num_wires; c++) wires[c][C.blind_rows[k]] = zrow.at(k * C.cfg.num_wires + c);
    for (size_t k = 0; k < C.blind_zrows.size(); k++)
        for (size_t c = 0; c < R; c++) {
            u64 v = zzrow.at(k * R + c);
            wires[c][C.blind_zrows[k].first] = v;
            wires[c][C.blind_zrows[k].second] = v;
        }
    for (size_t k = 0; k < C.poseidon_rows.size(); k++) {
        if (pos_rows[k].empty()) return 2;
        for (size_t c = R; c < C.cfg.num_wires; c++) wires[c][C.poseidon_rows[k]] = pos_rows[k][c];
    }
    struct FaultAtExit {  // applied to the finished matrix, whatever wrote the cell
        const OCircuit& C;
        std::vector<std::vector<u64>>& w;
        ~FaultAtExit() {
            if (C.fault_kind == 2 && C.fault_a < w.size() && C.fault_b < C.n) w[C.fault_a][C.fault_b] = fadd(w[C.fault_a][C.fault_b], C.fault_delta);
        }
    } fault_at_exit{C, wires};
    // LookupTableGate rows (stored upside down), multiplicities, and padding of the last LookupGate
    // (LookupTableGenerator + prover.rs set_lookup_wires)
    for (size_t l = 0; l < C.luts.size(); l++) {
        auto& lut = C.luts[l];
        auto lr = C.lookup_rows[l];
        std::vector<u64> mult(lut.size(), 0);
        for (const OOp& o : C.ops)
            if (o.kind == 2 && o.aux == l) mult[C.lut_dense[l][val[o.a]]]++;
        size_t remaining = (40 - C.num_lookups[l] % 40) % 40;
        mult[0] += remaining;
        for (size_t slot = 0; slot < lut.size(); slot++) {
            size_t row = lr.first_lut - slot / 26, s = slot % 26;
            wires[3 * s][row] = lut[slot].first;
            wires[3 * s + 1][row] = lut[slot].second;
            wires[3 * s + 2][row] = mult[slot];
        }
        for (size_t slot = 40 - remaining; slot < 40; slot++) {
            wires[2 * slot][lr.last_lut - 1] = lut[0].first;
            wires[2 * slot + 1][lr.last_lut - 1] = lut[0].second;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
struct PointVars {  // everything eval_vanishing_poly needs at one point of the LDE coset
    const u64 *consts, *wires, *sigmas, *zs, *zs_next;  // zs: full zs/pp/lookup row
};

// get_lut_poly: Horner over the (padded) table in slot order, consistent with the RE recursion.
static inline u64 lut_poly(const OCircuit& C, size_t lut, u64 chal_b, u64 delta) {
    size_t rows = (C.luts[lut].size() + 25) / 26, total = rows * 26;
    u64 acc = 0;
    for (size_t k = 0; k < total; k++) {
        u64 e = k < C.luts[lut].size() ? fadd(C.luts[lut][k].first, fmul(chal_b, C.luts[lut][k].second)) : 0;
        acc = fadd(fmul(acc, delta), e);
    }
    return acc;
}

struct Challenges {
    std::vector<u64> betas, gammas, deltas, alphas;
    std::vector<u64> lut_polys;  // [challenge][lut]
};

// vanishing polynomial terms at one point, combined with each alpha (eval_vanishing_poly_base)
static inline void eval_vanishing(const OCircuit& C, const Challenges& ch, u64 x, u64 l0_x, const PointVars& v, u64* out) {
    const size_t R = C.cfg.num_routed_wires, NC = C.cfg.num_challenges, npp = C.num_pp(), qdf = C.cfg.quotient_degree_factor;
    const size_t nsel = C.nsel(), nls = C.num_lookup_selectors, nlp = C.num_lookup_polys(), nsldc = C.num_sldc();
    // per-thread scratch, reused from point to point (this function runs 2^17 times per proof)
    static thread_local std::vector<u64> z1, ppt, lkt, gate, terms;
    z1.clear(), ppt.clear(), lkt.clear(), terms.clear();
    gate.assign(C.num_gate_constraints, 0);
    for (size_t i = 0; i < NC; i++) {
        u64 z_x = v.zs[i], z_gx = v.zs_next[i];
        z1.push_back(fmul(l0_x, fsub(z_x, 1)));
        // check_partial_products
        const u64* pp = v.zs + NC + i * npp;
        for (size_t chunk = 0; chunk * qdf < R; chunk++) {
            u64 num = 1, den = 1;
            for (size_t j = chunk * qdf; j < std::min(R, (chunk + 1) * qdf); j++) {
                u64 w = v.wires[j];
                num = fmul(num, fadd(fadd(w, fmul(ch.betas[i], fmul(C.k_is[j], x))), ch.gammas[i]));
                den = fmul(den, fadd(fadd(w, fmul(ch.betas[i], v.sigmas[j])), ch.gammas[i]));
            }
            u64 prev = chunk == 0 ? z_x : pp[chunk - 1];
            u64 next = chunk == npp ? z_gx : pp[chunk];
            ppt.push_back(fsub(fmul(prev, num), fmul(next, den)));
        }
        if (nlp) {  // check_lookup_constraints
            const u64* d = &ch.deltas[4 * i];  // A, B, Alpha, Delta
            const u64* lz = v.zs + NC * (1 + npp) + i * nlp;
            const u64* lzn = v.zs_next + NC * (1 + npp) + i * nlp;
            const u64* sel = v.consts + nsel;  // TransSre, TransLdc, InitSre, LastLdc, StartEnd...
            u64 z_re = lz[0], next_z_re = lzn[0];
            const u64 *sl = lz + 1, *sln = lzn + 1;
            const size_t lu_deg = qdf - 1, lut_deg = C.lut_degree();
            u64 looked[26], looking[40], lookup[26];
            for (int s = 0; s < 26; s++) {
                looked[s] = fadd(v.wires[3 * s], fmul(d[0], v.wires[3 * s + 1]));
                lookup[s] = fadd(v.wires[3 * s], fmul(d[1], v.wires[3 * s + 1]));
            }
            for (int s = 0; s < 40; s++) looking[s] = fadd(v.wires[2 * s], fmul(d[0], v.wires[2 * s + 1]));
            lkt.push_back(fmul(sel[3], sl[nsldc - 1]));  // LastLdc
            lkt.push_back(fmul(sel[2], sl[0]));          // InitSre (sum)
            lkt.push_back(fmul(sel[2], z_re));           // InitSre (RE)
            for (size_t l = 0; l < C.luts.size(); l++) lkt.push_back(fmul(sel[4 + l], fsub(z_re, ch.lut_polys[i * C.luts.size() + l])));
            u64 cur = next_z_re;
            for (int s = 0; s < 26; s++) cur = fadd(fmul(cur, d[3]), lookup[s]);
            lkt.push_back(fmul(sel[0], fsub(z_re, cur)));
            for (size_t poly = 0; poly < nsldc; poly++) {
                size_t a0 = poly * lut_deg, a1 = std::min<size_t>((poly + 1) * lut_deg, 26);
                size_t b0 = poly * lu_deg, b1 = std::min<size_t>((poly + 1) * lu_deg, 40);
                u64 lut_prod = 1, lu_prod = 1;
                for (size_t k = a0; k < a1; k++) lut_prod = fmul(lut_prod, fsub(d[2], looked[k]));
                for (size_t k = b0; k < b1; k++) lu_prod = fmul(lu_prod, fsub(d[2], looking[k]));
                u64 lu_sum = 0, lut_sum_mul = 0;
                for (size_t k = b0; k < b1; k++) {
                    u64 p = 1;
                    for (size_t m = b0; m < b1; m++)
                        if (m != k) p = fmul(p, fsub(d[2], looking[m]));
                    lu_sum = fadd(lu_sum, p);
                }
                for (size_t k = a0; k < a1; k++) {
                    u64 p = 1;
                    for (size_t m = a0; m < a1; m++)
                        if (m != k) p = fmul(p, fsub(d[2], looked[m]));
                    lut_sum_mul = fadd(lut_sum_mul, fmul(v.wires[3 * k + 2], p));
                }
                u64 prev = poly == 0 ? sln[nsldc - 1] : sl[poly - 1];
                u64 diff = fsub(sl[poly], prev);
                lkt.push_back(fmul(sel[0], fsub(fmul(lut_prod, diff), lut_sum_mul)));
                lkt.push_back(fmul(sel[1], fadd(fmul(lu_prod, diff), lu_sum)));
            }
        }
    }
    // gate constraints, each multiplied by its selector filter
    for (size_t gi = 0; gi < C.gates.size(); gi++) {
        u32 kind = C.gates[gi];
        if (kind != GK_ARITHMETIC && kind != GK_CONSTANT && kind != GK_PUBLIC_INPUT && kind != GK_POSEIDON) continue;
        size_t si = C.selector_index[gi];
        u64 s = v.consts[si], filter = 1;
        for (u32 j = C.groups[si].first; j < C.groups[si].second; j++)
            if (j != gi) filter = fmul(filter, fsub(j, s));
        if (nsel > 1) filter = fmul(filter, fsub(0xFFFFFFFFull, s));
        const u64* gc = v.consts + nsel + nls;
        if (kind == GK_ARITHMETIC) {
            for (int op = 0; op < 20; op++) {
                u64 m0 = v.wires[4 * op], m1 = v.wires[4 * op + 1], ad = v.wires[4 * op + 2], o = v.wires[4 * op + 3];
                u64 c = fsub(o, fadd(fmul(fmul(m0, m1), gc[0]), fmul(ad, gc[1])));
                gate[op] = fadd(gate[op], fmul(filter, c));
            }
        } else if (kind == GK_CONSTANT) {
            for (int k = 0; k < 2; k++) gate[k] = fadd(gate[k], fmul(filter, fsub(gc[k], v.wires[k])));
        } else if (kind == GK_POSEIDON) {
            std::vector<u64> roww(v.wires, v.wires + C.cfg.num_wires), cs;
            poseidon_gate_walk(roww.data(), false, &cs);
            for (size_t k = 0; k < cs.size(); k++) gate[k] = fadd(gate[k], fmul(filter, cs[k]));
        } else {
            for (int k = 0; k < 4; k++) gate[k] = fadd(gate[k], fmul(filter, v.wires[k]));  // public_inputs_hash = 0
        }
    }
    terms.insert(terms.end(), z1.begin(), z1.end());
    terms.insert(terms.end(), ppt.begin(), ppt.end());
    terms.insert(terms.end(), lkt.begin(), lkt.end());
    terms.insert(terms.end(), gate.begin(), gate.end());
    for (size_t i = 0; i < NC; i++) {  // reduce_with_powers(terms, alpha)
        u64 acc = 0;
        for (size_t k = terms.size(); k-- > 0;) acc = fadd(fmul(acc, ch.alphas[i]), terms[k]);
        out[i] = acc;
    }
}

// v[i] <- 1 / v[i] for every i (0 stays 0, as finv(0) = 0): one inversion per block of 1024 (Montgomery's trick), blocks in parallel
static inline void batch_inverse_parallel(std::vector<u64>& v) {
    const size_t BLK = 1024, count = v.size();
#pragma omp parallel for schedule(static)
    for (size_t b0 = 0; b0 < count; b0 += BLK) {
        const size_t m = std::min(BLK, count - b0);
        u64 prefix[BLK];
        u64 acc = 1;
        for (size_t k = 0; k < m; k++) {
            prefix[k] = acc;  // product of the non-zero entries before k
            if (v[b0 + k]) acc = fmul(acc, v[b0 + k]);
        }
        u64 inv = finv(acc);
        for (size_t k = m; k-- > 0;) {
            const u64 x = v[b0 + k];
            if (!x) continue;
            v[b0 + k] = fmul(inv, prefix[k]);
            inv = fmul(inv, x);
        }
    }
}

static inline X2 eval_poly_ext(const std::vector<u64>& coeffs, X2 z) {
    X2 acc = x2(0);
    for (size_t i = coeffs.size(); i-- > 0;) acc = acc * z + x2(coeffs[i]);
    return acc;
}

struct ByteWriter {
    std::vector<uint8_t> b;
    void w64(u64 v) {
        for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i)));
    }
    void digest(const Digest& d) {
        for (int i = 0; i < 4; i++) w64(d.e[i]);
    }
    void ext(X2 x) {
        w64(x.c0);
        w64(x.c1);
    }
    void merkle_proof(const std::vector<Digest>& s) {
        b.push_back((uint8_t)s.size());
        for (auto& d : s) digest(d);
    }
};

// returns status (0 ok); fills proof bytes.  `trace` (optional) receives intermediate buffers by name.
static inline int prove(const OCircuit& C, const u64* in_targets, const u64* in_values, size_t n_in, std::vector<uint8_t>& proof, Trace* trace) {
    const size_t n = C.n, N = n << C.cfg.rate_bits, R = C.cfg.num_routed_wires, NC = C.cfg.num_challenges;
    const size_t npp = C.num_pp(), nlp = C.num_lookup_polys(), nsldc = C.num_sldc(), qdf = C.cfg.quotient_degree_factor;
    const size_t ncc = C.ncc();
    auto tr = [&](const char* name, const std::vector<u64>& v) {
        if (trace) (*trace)[name] = v;
    };
    auto flat_cap = [&](const MerkleTree& t) {
        std::vector<u64> f;
        for (auto& d : t.cap())
            for (int i = 0; i < 4; i++) f.push_back(d.e[i]);
        return f;
    };
    // 1. witness
    for (double& v : g_stage_seconds) v = 0;
    StageClock clk;
    std::vector<std::vector<u64>> wires;
    int st = generate_witness(C, in_targets, in_values, n_in, wires);
    clk.lap(0);
    if (st) return st;
    if (trace) {
        std::vector<u64> f;
        for (auto& c : wires) f.insert(f.end(), c.begin(), c.end());
        tr("wires", f);
    }
    // 2. wires commitment
    Batch wb;
    clk.lap(0);
    commit_from_values(C, wb, wires, 1);
    clk.lap(1);
    tr("wires_cap", flat_cap(wb.tree));
    // 3. challenger
    Challenger chal;
    chal.observe(C.circuit_digest);
    for (int i = 0; i < 4; i++) chal.observe((u64)0);  // public_inputs_hash = hash_no_pad([]) = 0
    chal.observe_cap(wb.tree.cap());
    Challenges ch;
    for (size_t i = 0; i < NC; i++) ch.betas.push_back(chal.challenge());
    for (size_t i = 0; i < NC; i++) ch.gammas.push_back(chal.challenge());
    if (nlp) {
        ch.deltas = ch.betas;
        ch.deltas.insert(ch.deltas.end(), ch.gammas.begin(), ch.gammas.end());
        for (size_t i = 0; i < 4 * NC - 2 * NC; i++) ch.deltas.push_back(chal.challenge());
        for (size_t i = 0; i < NC; i++)
            for (size_t l = 0; l < C.luts.size(); l++) ch.lut_polys.push_back(lut_poly(C, l, ch.deltas[4 * i + 1], ch.deltas[4 * i + 3]));
    }
    tr("betas", ch.betas);
    tr("gammas", ch.gammas);
    tr("deltas", ch.deltas);
    // 4. partial products and Z (wires_permutation_partial_products_and_zs)
    std::vector<std::vector<u64>> zcols(NC * (1 + npp) + NC * nlp, std::vector<u64>(n, 0));
    {
        std::vector<u64> subgroup(n);
        u64 w = root_of_unity(C.degree_bits), x = 1;
        for (size_t i = 0; i < n; i++) {
            subgroup[i] = x;
            x = fmul(x, w);
        }
        for (size_t i = 0; i < NC; i++) {
            // quotient num / den per (row, chunk): the denominators are inverted together (batch_inverse), block by block
            std::vector<u64> q((npp + 1) * n), dens((npp + 1) * n);
#pragma omp parallel for schedule(static)
            for (size_t row = 0; row < n; row++) {
                for (size_t chunk = 0; chunk <= npp; chunk++) {
                    u64 num = 1, den = 1;
                    for (size_t j = chunk * qdf; j < std::min(R, (chunk + 1) * qdf); j++) {
                        u64 wv = wires[j][row];
                        num = fmul(num, fadd(fadd(wv, fmul(ch.betas[i], fmul(C.k_is[j], subgroup[row]))), ch.gammas[i]));
                        den = fmul(den, fadd(fadd(wv, fmul(ch.betas[i], C.sigmas[j * n + row])), ch.gammas[i]));
                    }
                    q[chunk * n + row] = num;
                    dens[chunk * n + row] = den;
                }
            }
            batch_inverse_parallel(dens);
#pragma omp parallel for schedule(static)
            for (size_t k = 0; k < q.size(); k++) q[k] = fmul(q[k], dens[k]);
            u64 z = 1;
            for (size_t row = 0; row < n; row++) {
                zcols[i][row] = z;
                u64 acc = z;
                for (size_t chunk = 0; chunk <= npp; chunk++) {
                    acc = fmul(acc, q[chunk * n + row]);
                    if (chunk < npp) zcols[NC + i * npp + chunk][row] = acc;
                }
                z = acc;
            }
        }
    }
    // 5. lookup polynomials RE + partial SLDCs (compute_lookup_polys)
    for (size_t i = 0; i < NC && nlp; i++) {
        const u64* d = &ch.deltas[4 * i];
        auto col = [&](size_t p) -> std::vector<u64>& { return zcols[NC * (1 + npp) + i * nlp + p]; };
        const size_t lu_deg = qdf - 1, lut_deg = C.lut_degree();
        for (auto lr : C.lookup_rows) {
            // 1 / (delta_2 - combo) of every (row, slot entry) first, all at once; the running sums below are serial
            const size_t lut_rows = lr.first_lut + 1 - lr.last_lut, lu_rows = lr.last_lut - lr.last_lu;
            std::vector<u64> inv_lut(lut_rows * 26), inv_lu(lu_rows * 40);
#pragma omp parallel for schedule(static)
            for (size_t k = 0; k < lut_rows; k++)
                for (size_t s = 0; s < 26; s++) {
                    const size_t row = lr.last_lut + k;
                    inv_lut[k * 26 + s] = fsub(d[2], fadd(wires[3 * s][row], fmul(d[0], wires[3 * s + 1][row])));
                }
#pragma omp parallel for schedule(static)
            for (size_t k = 0; k < lu_rows; k++)
                for (size_t s = 0; s < 40; s++) {
                    const size_t row = lr.last_lu + k;
                    inv_lu[k * 40 + s] = fsub(d[2], fadd(wires[2 * s][row], fmul(d[0], wires[2 * s + 1][row])));
                }
            batch_inverse_parallel(inv_lut);
            batch_inverse_parallel(inv_lu);
            for (size_t row = lr.first_lut + 1; row-- > lr.last_lut;) {
                u64 re = col(0)[row + 1];
                for (int s = 0; s < 26; s++) re = fadd(fmul(re, d[3]), fadd(wires[3 * s][row], fmul(d[1], wires[3 * s + 1][row])));
                col(0)[row] = re;
                for (size_t slot = 0; slot < nsldc; slot++) {
                    u64 acc = slot ? col(slot)[row] : col(nsldc)[row + 1];
                    for (size_t s = slot * lut_deg; s < std::min<size_t>((slot + 1) * lut_deg, 26); s++)
                        acc = fadd(acc, fmul(wires[3 * s + 2][row], inv_lut[(row - lr.last_lut) * 26 + s]));
                    col(slot + 1)[row] = acc;
                }
            }
            for (size_t row = lr.last_lut; row-- > lr.last_lu;) {
                for (size_t slot = 0; slot < nsldc; slot++) {
                    u64 prev = slot ? col(slot)[row] : col(nsldc)[row + 1];
                    u64 sum = 0;
                    for (size_t s = slot * lu_deg; s < std::min<size_t>((slot + 1) * lu_deg, 40); s++) sum = fadd(sum, inv_lu[(row - lr.last_lu) * 40 + s]);
                    col(slot + 1)[row] = fsub(prev, sum);
                }
            }
        }
    }
    if (trace) {
        std::vector<u64> f;
        for (auto& c : zcols) f.insert(f.end(), c.begin(), c.end());
        tr("zs", f);
    }
    Batch zb;
    clk.lap(2);
    commit_from_values(C, zb, zcols, 2);
    clk.lap(1);
    tr("zs_cap", flat_cap(zb.tree));
    chal.observe_cap(zb.tree.cap());
    for (size_t i = 0; i < NC; i++) ch.alphas.push_back(chal.challenge());
    tr("alphas", ch.alphas);
    // 7. quotient polynomials (compute_quotient_polys)
    Batch qb;
    {
        std::vector<std::vector<u64>> qv(NC, std::vector<u64>(N));
        u64 wN = root_of_unity(C.lde_bits);
        std::vector<u64> xs(N);
        {
            u64 x = GENERATOR;
            for (size_t i = 0; i < N; i++) {
                xs[i] = x;
                x = fmul(x, wN);
            }
        }
        // Z_H(x) = x^n - 1 takes 2^rate_bits values on the coset
        std::vector<u64> zh_inv((size_t)1 << C.cfg.rate_bits);
        u64 gn = fpow(GENERATOR, n), w8 = root_of_unity(C.cfg.rate_bits);
        for (size_t j = 0; j < zh_inv.size(); j++) zh_inv[j] = finv(fsub(fmul(gn, fpow(w8, j)), 1));
        const size_t step = (size_t)1 << C.cfg.rate_bits;
        u64 n_inv = finv(n);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < N; i++) {
            size_t li = rev_bits(i, C.lde_bits), ln = rev_bits((i + step) % N, C.lde_bits);
            PointVars v;
            v.consts = &C.pre.lde[li * C.pre.width];
            v.sigmas = v.consts + ncc;
            v.wires = &wb.lde[li * wb.width];
            v.zs = &zb.lde[li * zb.width];
            v.zs_next = &zb.lde[ln * zb.width];
            u64 x = xs[i];
            u64 zh = fsub(fmul(gn, fpow(w8, i % step)), 1);
            // L_0(x) = (x^n - 1) / (n (x - 1))
            u64 l0 = fmul(zh, finv(fmul(n % MODULUS, fsub(x, 1))));
            (void)n_inv;
            u64 out[8];
            eval_vanishing(C, ch, x, l0, v, out);
            for (size_t k = 0; k < NC; k++) qv[k][i] = fmul(out[k], zh_inv[i % step]);
        }
        qb.cols = NC * qdf;
        qb.coeffs.assign(qb.cols, std::vector<u64>(n));
        for (size_t k = 0; k < NC; k++) {
            std::vector<u64> co = coset_ifft(qv[k], C.lde_bits, GENERATOR);
            for (size_t c = 0; c < qdf; c++) memcpy(qb.coeffs[k * qdf + c].data(), &co[c * n], n * 8);
        }
        if (trace) {
            std::vector<u64> f;
            for (auto& c : qb.coeffs) f.insert(f.end(), c.begin(), c.end());
            tr("quotient_coeffs", f);
        }
        clk.lap(3);
        commit_from_coeffs(C, qb, 3);
        clk.lap(1);
    }
    tr("quotient_cap", flat_cap(qb.tree));
    chal.observe_cap(qb.tree.cap());
    X2 zeta = chal.ext_challenge();
    tr("zeta", {zeta.c0, zeta.c1});
    {
        X2 zp = zeta;
        for (u32 i = 0; i < C.degree_bits; i++) zp = zp * zp;
        if (zp == x2(1)) return 3;  // "Opening point is in the subgroup."
    }
    X2 g_zeta = zeta * root_of_unity(C.degree_bits);
    // 9. openings
    const Batch* oracles[4] = {&C.pre, &wb, &zb, &qb};
    std::vector<X2> op_pre(C.pre.cols), op_w(wb.cols), op_z(zb.cols), op_zn(zb.cols), op_q(qb.cols);
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t c = 0; c < C.pre.cols; c++) op_pre[c] = eval_poly_ext(C.pre.coeffs[c], zeta);
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t c = 0; c < wb.cols; c++) op_w[c] = eval_poly_ext(wb.coeffs[c], zeta);
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t c = 0; c < zb.cols; c++) {
        op_z[c] = eval_poly_ext(zb.coeffs[c], zeta);
        op_zn[c] = eval_poly_ext(zb.coeffs[c], g_zeta);
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t c = 0; c < qb.cols; c++) op_q[c] = eval_poly_ext(qb.coeffs[c], zeta);
    const size_t nzpp = NC * (1 + npp);
    // OpeningSet fields, in serialisation order
    std::vector<X2> o_constants(op_pre.begin(), op_pre.begin() + ncc), o_sigmas(op_pre.begin() + ncc, op_pre.end());
    std::vector<X2> o_zs(op_z.begin(), op_z.begin() + NC), o_zs_next(op_zn.begin(), op_zn.begin() + NC);
    std::vector<X2> o_pp(op_z.begin() + NC, op_z.begin() + nzpp);
    std::vector<X2> o_lk(op_z.begin() + nzpp, op_z.end()), o_lk_next(op_zn.begin() + nzpp, op_zn.end());
    // to_fri_openings + observe_openings
    std::vector<X2> batch0, batch1;
    for (auto* v : {&o_constants, &o_sigmas, &op_w, &o_zs, &o_pp, &op_q, &o_lk}) batch0.insert(batch0.end(), v->begin(), v->end());
    for (auto* v : {&o_zs_next, &o_lk_next}) batch1.insert(batch1.end(), v->begin(), v->end());
    for (auto& e : batch0) chal.observe(e);
    for (auto& e : batch1) chal.observe(e);
    if (trace) {
        std::vector<u64> f;
        for (auto& e : batch0) { f.push_back(e.c0); f.push_back(e.c1); }
        for (auto& e : batch1) { f.push_back(e.c0); f.push_back(e.c1); }
        tr("openings", f);
    }
    // 10. FRI: batch polynomials (PolynomialBatch::prove_openings)
    clk.lap(4);
    X2 fri_alpha = chal.ext_challenge();
    tr("fri_alpha", {fri_alpha.c0, fri_alpha.c1});
    struct PolyRef {
        int oracle;
        size_t idx;
    };
    std::vector<PolyRef> b0, b1;
    for (size_t c = 0; c < C.pre.cols; c++) b0.push_back({0, c});
    for (size_t c = 0; c < wb.cols; c++) b0.push_back({1, c});
    for (size_t c = 0; c < nzpp; c++) b0.push_back({2, c});
    for (size_t c = 0; c < qb.cols; c++) b0.push_back({3, c});
    for (size_t c = nzpp; c < zb.cols; c++) b0.push_back({2, c});
    for (size_t c = 0; c < NC; c++) b1.push_back({2, c});
    for (size_t c = nzpp; c < zb.cols; c++) b1.push_back({2, c});
    auto batch_quotient = [&](const std::vector<PolyRef>& polys, X2 point) {
        // composition = sum_j alpha^j f_j ; then divide by (X - point), dropping the remainder
        std::vector<X2> comp(n, x2(0));
        X2 ap = x2(1);
        std::vector<X2> apow(polys.size());
        for (size_t j = 0; j < polys.size(); j++) {
            apow[j] = ap;
            ap = ap * fri_alpha;
        }
#pragma omp parallel for schedule(static)
        for (size_t k = 0; k < n; k++) {
            X2 acc = x2(0);
            for (size_t j = 0; j < polys.size(); j++) acc = acc + apow[j] * oracles[polys[j].oracle]->coeffs[polys[j].idx][k];
            comp[k] = acc;
        }
        std::vector<X2> quo(n, x2(0));
        X2 carry = x2(0);
        for (size_t k = n; k-- > 1;) {
            carry = comp[k] + point * carry;
            quo[k - 1] = carry;
        }
        return quo;  // quo[n-1] = 0 : "pad back to power of two"
    };
    std::vector<X2> final_poly = batch_quotient(b0, zeta);
    {
        std::vector<X2> q1 = batch_quotient(b1, g_zeta);
        X2 shift = xpow(fri_alpha, b1.size());
        for (size_t k = 0; k < n; k++) final_poly[k] = final_poly[k] * shift + q1[k];
    }
    if (trace) {
        std::vector<u64> f;
        for (auto& e : final_poly) { f.push_back(e.c0); f.push_back(e.c1); }
        tr("fri_final_poly_in", f);
    }
    auto ext_coset_fft = [&](const std::vector<X2>& coeffs, int bits, u64 shift) {
        std::vector<u64> a(coeffs.size()), b(coeffs.size());
        for (size_t i = 0; i < coeffs.size(); i++) {
            a[i] = coeffs[i].c0;
            b[i] = coeffs[i].c1;
        }
        auto va = coset_fft(a, bits, shift), vb = coset_fft(b, bits, shift);
        std::vector<X2> v(va.size());
        for (size_t i = 0; i < v.size(); i++) v[i] = X2{va[i], vb[i]};
        return v;
    };
    // commit phase (fri_committed_trees)
    std::vector<X2> coeffs(final_poly);
    coeffs.resize(N, x2(0));
    int bits = C.lde_bits;
    std::vector<X2> values = ext_coset_fft(coeffs, bits, GENERATOR);
    std::vector<MerkleTree> fri_trees;
    std::vector<std::vector<u64>> fri_leaves;  // per round: [num_leaves][2*arity]
    std::vector<u32> arities = C.arity_bits();
    u64 shift = GENERATOR;
    std::vector<X2> fri_betas;
    for (u32 ab : arities) {
        size_t arity = (size_t)1 << ab, len = values.size();
        std::vector<u64> leaves(2 * len);
        for (size_t i = 0; i < len; i++) {
            size_t r = rev_bits(i, bits);
            leaves[2 * r] = values[i].c0;
            leaves[2 * r + 1] = values[i].c1;
        }
        fri_trees.push_back(build_merkle(leaves.data(), len / arity, 2 * arity, C.cfg.cap_height));
        fri_leaves.push_back(leaves);
        chal.observe_cap(fri_trees.back().cap());
        X2 beta = chal.ext_challenge();
        fri_betas.push_back(beta);
        std::vector<X2> folded(coeffs.size() / arity);
        for (size_t k = 0; k < folded.size(); k++) {
            X2 acc = x2(0);
            for (size_t i = arity; i-- > 0;) acc = acc * beta + coeffs[k * arity + i];
            folded[k] = acc;
        }
        coeffs = folded;
        shift = fpow(shift, arity);
        bits -= ab;
        values = ext_coset_fft(coeffs, bits, shift);
    }
    coeffs.resize(coeffs.size() >> C.cfg.rate_bits);
    for (auto& e : coeffs) chal.observe(e);
    if (trace) {
        std::vector<u64> f;
        for (auto& e : fri_betas) { f.push_back(e.c0); f.push_back(e.c1); }
        tr("fri_betas", f);
        f.clear();
        for (auto& t : fri_trees)
            for (auto& d : t.cap())
                for (int i = 0; i < 4; i++) f.push_back(d.e[i]);
        tr("fri_caps", f);
        f.clear();
        for (auto& e : coeffs) { f.push_back(e.c0); f.push_back(e.c1); }
        tr("fri_final_poly", f);
    }
    // proof of work: smallest witness whose response has >= pow_bits leading zeros
    u64 pow_witness = ~0ull;
    for (u64 base = 0; pow_witness == ~0ull; base += 1 << 14) {  // blocks of 2^14 candidates, smallest hit of the first block with one
        u64 found = ~0ull;
#pragma omp parallel for schedule(static) reduction(min : found)
        for (long long k = 0; k < (1 << 14); k++) {
            Challenger c2 = chal;
            c2.observe(base + (u64)k);
            u64 resp = c2.challenge();
            if ((resp >> (64 - C.cfg.pow_bits)) == 0 && base + (u64)k < found) found = base + (u64)k;
        }
        pow_witness = found;
    }
    chal.observe(pow_witness);
    (void)chal.challenge();
    tr("pow_witness", {pow_witness});
    // query phase
    ByteWriter w;
    for (auto& d : wb.tree.cap()) w.digest(d);
    for (auto& d : zb.tree.cap()) w.digest(d);
    for (auto& d : qb.tree.cap()) w.digest(d);
    for (auto* v : {&o_constants, &o_sigmas, &op_w, &o_zs, &o_zs_next, &o_lk, &o_lk_next, &o_pp, &op_q})  // write_opening_set's order
        for (auto& e : *v) w.ext(e);
    for (auto& t : fri_trees)
        for (auto& d : t.cap()) w.digest(d);
    std::vector<u64> qidx;
    for (u32 q = 0; q < C.cfg.num_query_rounds; q++) {
        size_t x_index = (size_t)(chal.challenge() % N);
        qidx.push_back(x_index);
        for (int o = 0; o < 4; o++) {
            const Batch* b = oracles[o];
            for (size_t c = 0; c < b->width; c++) w.w64(b->lde[x_index * b->width + c]);
            w.merkle_proof(b->tree.prove(x_index));
        }
        for (size_t r = 0; r < arities.size(); r++) {
            size_t arity = (size_t)1 << arities[r];
            x_index >>= arities[r];
            for (size_t k = 0; k < 2 * arity; k++) w.w64(fri_leaves[r][x_index * 2 * arity + k]);
            w.merkle_proof(fri_trees[r].prove(x_index));
        }
    }
    tr("query_indices", qidx);
    for (auto& e : coeffs) w.ext(e);
    w.w64(pow_witness);
    proof.swap(w.b);
    clk.lap(5);
    return 0;
}

}  // namespace orc

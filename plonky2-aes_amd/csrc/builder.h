// Host-side mirror of the slice of plonky2's CircuitBuilder API that the reference's gadget crates call
// (SURVEY.md Appendix A.1): add_virtual_target, constant/zero/one, arithmetic (mul_const_add, add, mul, sub,
// mul_sub), select, is_equal, connect, add_lookup_table_from_pairs, add_lookup_from_index, num_gates, build.
// Same names, argument meaning and constant-folding / op-caching / slot-packing behaviour as upstream, so the
// gate rows a gadget produces here are the rows it produces there.  Gate TYPES follow upstream's GateRef identity: every
// lookup table has its own LookupGate and LookupTableGate type (their id() carries the table's Keccak hash), which decides
// the number of selector polynomials; sigma sends each routed wire to the next wire of its copy class in (row, column)
// order, as `wire_partition` / `get_sigma_map` do.  oracle/oracle_builder.py derives all of this a second time,
// independently, and tests/test_independent_builder.py compares.
//
// Call sites mirrored: aes-gcm/src/circuit_aes.rs:176-358, aes-gcm/src/circuit_gcm.rs:49-425.
#pragma once
#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <unordered_map>

#include "circuit.h"
#include "gl.h"
#include "keccak.h"

namespace p2 {

// Target: virtual target (index) or routed wire (row, column).  plonky2 iop::target::Target.
typedef u64 Target;
static const u64 T_WIRE_BIT = 1ull << 63;
static inline Target wire_target(u32 row, u32 col) { return T_WIRE_BIT | ((u64)row << 8) | col; }
static inline bool is_wire(Target t) { return (t & T_WIRE_BIT) != 0; }
static inline u32 wire_row(Target t) { return (u32)((t & ~T_WIRE_BIT) >> 8); }
static inline u32 wire_col(Target t) { return (u32)(t & 0xFF); }

struct BoolTarget {
    Target target;
};

struct GateInstance {
    u32 kind;
    u64 constants[2];
};

class CircuitBuilder {
   public:
    explicit CircuitBuilder(const Config& cfg = Config()) : cfg_(cfg) {}

    // ---- targets ------------------------------------------------------------------------------------
    Target add_virtual_target() { return (Target)num_virtual_++; }
    BoolTarget add_virtual_bool_target_unsafe() { return BoolTarget{add_virtual_target()}; }

    Target constant(u64 c) {
        c %= gl::P;
        auto it = constants_to_targets_.find(c);
        if (it != constants_to_targets_.end()) return it->second;
        Target t = add_virtual_target();
        constants_to_targets_[c] = t;
        targets_to_constants_[t] = c;
        return t;
    }
    Target zero() { return constant(0); }
    Target one() { return constant(1); }
    Target neg_one() { return constant(gl::P - 1); }

    bool target_as_constant(Target t, u64* out) const {
        auto it = targets_to_constants_.find(t);
        if (it == targets_to_constants_.end()) return false;
        *out = it->second;
        return true;
    }

    void connect(Target x, Target y) { copy_constraints_.push_back({x, y}); }

    size_t num_gates() const { return gate_instances_.size(); }

    // ---- arithmetic (plonky2 gadgets/arithmetic.rs) -----------------------------------------------
    // const_0 * multiplicand_0 * multiplicand_1 + const_1 * addend
    Target arithmetic(u64 c0, u64 c1, Target m0, Target m1, Target addend) {
        Target special;
        if (arithmetic_special_cases(c0, c1, m0, m1, addend, &special)) return special;
        ArithKey key{c0, c1, m0, m1, addend};
        auto it = base_arithmetic_results_.find(key);
        if (it != base_arithmetic_results_.end()) return it->second;
        // find_slot(ArithmeticGate, params = [c0, c1])
        auto sl = arith_slots_.find({c0, c1});
        u32 row, i;
        if (sl == arith_slots_.end()) {
            row = add_gate(G_ARITHMETIC, c0, c1);
            i = 0;
        } else {
            row = sl->second.first;
            i = sl->second.second;
        }
        if (i == ARITH_OPS - 1)
            arith_slots_.erase({c0, c1});
        else
            arith_slots_[{c0, c1}] = {row, i + 1};
        connect(m0, wire_target(row, 4 * i));
        connect(m1, wire_target(row, 4 * i + 1));
        connect(addend, wire_target(row, 4 * i + 2));
        Target out = wire_target(row, 4 * i + 3);
        gens_.push_back(Gen{OP_ARITH, out, wire_target(row, 4 * i), wire_target(row, 4 * i + 1), wire_target(row, 4 * i + 2), 0, c0, c1});
        base_arithmetic_results_[key] = out;
        return out;
    }
    // c*x + y.  Upstream passes the constant one as the FIRST multiplicand here (`self.arithmetic(c, F::ONE, one, x, y)`), unlike
    // add / sub where it is the second: which of the two multiplicand wires carries x is part of the circuit.
    Target mul_const_add(u64 c, Target x, Target y) { return arithmetic(c, 1, one(), x, y); }
    Target add(Target x, Target y) { return arithmetic(1, 1, x, one(), y); }
    Target sub(Target x, Target y) { return arithmetic(1, gl::P - 1, x, one(), y); }
    Target mul(Target x, Target y) { return arithmetic(1, 0, x, y, x); }
    Target mul_sub(Target x, Target y, Target z) { return arithmetic(1, gl::P - 1, x, y, z); }  // x*y - z
    Target mul_const(u64 c, Target x) { return mul_const_add(c, x, zero()); }
    // if b { x } else { y }
    Target select(BoolTarget b, Target x, Target y) {
        Target tmp = mul_sub(b.target, y, y);
        return mul_sub(b.target, x, tmp);
    }
    BoolTarget not_(BoolTarget b) { return BoolTarget{sub(one(), b.target)}; }
    BoolTarget is_equal(Target x, Target y) {
        Target z = zero();
        BoolTarget equal = add_virtual_bool_target_unsafe();
        BoolTarget not_equal = not_(equal);
        Target inv = add_virtual_target();
        gens_.push_back(Gen{OP_EQ, equal.target, x, y, 0, 0, 0, 0});
        gens_.push_back(Gen{OP_EQINV, inv, x, y, 0, 0, 0, 0});
        Target diff = sub(x, y);
        Target not_equal_check = mul(equal.target, diff);
        Target diff_normalized = mul(diff, inv);
        connect(not_equal.target, diff_normalized);
        connect(not_equal_check, z);
        return equal;
    }

    // 1/x (plonky2 gadgets/arithmetic.rs `inverse`): a hinted value checked by x * inv = 1; x = 0 has no witness
    Target inverse(Target x) {
        Target inv = add_virtual_target();
        gens_.push_back(Gen{OP_EQINV, inv, x, zero(), 0, 0, 0, 0});
        connect(mul(x, inv), one());
        return inv;
    }

    // ---- lookups (plonky2 gadgets/lookup.rs) ------------------------------------------------------
    size_t add_lookup_table_from_pairs(const std::vector<std::pair<u16, u16>>& table) {
        for (size_t i = 0; i < luts_.size(); i++)
            if (luts_[i] == table) return i;
        luts_.push_back(table);
        lut_to_lookups_.emplace_back();
        return luts_.size() - 1;
    }
    Target add_lookup_from_index(Target looking_in, size_t lut_index) {
        if (lut_index >= luts_.size()) throw std::runtime_error("lut index not in luts");
        Target looking_out = add_virtual_target();
        lut_to_lookups_[lut_index].push_back({looking_in, looking_out});
        return looking_out;
    }
    size_t num_luts() const { return luts_.size(); }

    // ---- hashing (plonky2 gadgets/hash.rs, hash/poseidon.rs AlgebraicHasher::permute_swapped) --------
    // One PoseidonGate row; swap = false.
    std::array<Target, 12> permute(const std::array<Target, 12>& inputs) {
        u32 row = add_gate(G_POSEIDON);
        connect(zero(), wire_target(row, PG_SWAP));
        for (u32 i = 0; i < 12; i++) connect(inputs[i], wire_target(row, PG_IN + i));
        gens_.push_back(Gen{OP_POSEIDON, wire_target(row, PG_OUT), (Target)row, 0, 0, 0, 0, 0});
        std::array<Target, 12> out;
        for (u32 i = 0; i < 12; i++) out[i] = wire_target(row, PG_OUT + i);
        return out;
    }
    // Overwrite-mode sponge, rate 8 (poseidon-cipher/src/circuit.rs:123, hashed_elgamal/circuit.rs:42)
    std::vector<Target> hash_n_to_m_no_pad(const std::vector<Target>& inputs, size_t num_outputs) {
        std::array<Target, 12> state;
        state.fill(zero());
        for (size_t off = 0; off < inputs.size(); off += 8) {
            for (size_t i = 0; i < 8 && off + i < inputs.size(); i++) state[i] = inputs[off + i];
            state = permute(state);
        }
        std::vector<Target> outputs;
        for (;;) {
            for (size_t i = 0; i < 8; i++) {
                outputs.push_back(state[i]);
                if (outputs.size() == num_outputs) return outputs;
            }
            state = permute(state);
        }
    }

    // ---- build (plonky2 CircuitBuilder::build) ----------------------------------------------------
    Circuit build();

   private:
    struct ArithKey {
        u64 c0, c1;
        Target m0, m1, a;
        bool operator==(const ArithKey& o) const { return c0 == o.c0 && c1 == o.c1 && m0 == o.m0 && m1 == o.m1 && a == o.a; }
    };
    struct ArithKeyHash {
        size_t operator()(const ArithKey& k) const {
            u64 h = k.c0 * 0x9E3779B97F4A7C15ull;
            h = (h ^ k.c1) * 0xC2B2AE3D27D4EB4Full;
            h = (h ^ k.m0) * 0x9E3779B97F4A7C15ull;
            h = (h ^ k.m1) * 0xC2B2AE3D27D4EB4Full;
            h = (h ^ k.a) * 0x9E3779B97F4A7C15ull;
            return (size_t)(h ^ (h >> 29));
        }
    };
    struct Gen {
        u32 kind;
        Target out, a, b, c;
        u32 aux;
        u64 k0, k1;
    };

    bool arithmetic_special_cases(u64 c0, u64 c1, Target m0, Target m1, Target addend, Target* out) {
        Target z = zero();
        u64 m0c = 0, m1c = 0, ac = 0;
        bool m0k = target_as_constant(m0, &m0c), m1k = target_as_constant(m1, &m1c), ak = target_as_constant(addend, &ac);
        bool first_zero = c0 == 0 || m0 == z || m1 == z;
        bool second_zero = c1 == 0 || addend == z;
        bool fk = false, sk = false;
        u64 fv = 0, sv = 0;
        if (first_zero) {
            fk = true;
        } else if (m0k && m1k) {
            fk = true;
            fv = gl::mul(gl::mul(m0c, m1c), c0);
        }
        if (second_zero) {
            sk = true;
        } else if (ak) {
            sk = true;
            sv = gl::mul(ac, c1);
        }
        if (fk && sk) {
            *out = constant(gl::add(fv, sv));
            return true;
        }
        if (first_zero && c1 == 1) {
            *out = addend;
            return true;
        }
        if (second_zero) {
            if (m0k && gl::mul(m0c, c0) == 1) {
                *out = m1;
                return true;
            }
            if (m1k && gl::mul(m1c, c0) == 1) {
                *out = m0;
                return true;
            }
        }
        return false;
    }

    u32 add_gate(u32 kind, u64 c0 = 0, u64 c1 = 0) {
        u32 row = (u32)gate_instances_.size();
        gate_instances_.push_back(GateInstance{kind, {c0, c1}});
        if (kind == G_CONSTANT) {
            constant_generators_.push_back({row, 0});
            constant_generators_.push_back({row, 1});
        }
        return row;
    }
    void add_all_lookups();

    Config cfg_;
    u64 num_virtual_ = 0;
    std::vector<GateInstance> gate_instances_;
    std::vector<std::pair<Target, Target>> copy_constraints_;
    std::map<u64, Target> constants_to_targets_;  // ordered by canonical value, as upstream's sorted_by_key
    std::unordered_map<Target, u64> targets_to_constants_;
    std::unordered_map<ArithKey, Target, ArithKeyHash> base_arithmetic_results_;
    std::map<std::pair<u64, u64>, std::pair<u32, u32>> arith_slots_;
    std::vector<std::vector<std::pair<u16, u16>>> luts_;
    std::vector<std::vector<std::pair<Target, Target>>> lut_to_lookups_;
    std::vector<LookupRows> lookup_rows_;
    std::vector<std::pair<u32, u32>> constant_generators_;  // (row, index)
    std::vector<Gen> gens_;
};

// Adds, per LUT: LookupGate rows for the looking pairs, then the LookupTableGate rows (stored upside down),
// then one NoopGate.  plonky2 CircuitBuilder::add_all_lookups.
inline void CircuitBuilder::add_all_lookups() {
    for (size_t lut = 0; lut < luts_.size(); lut++) {
        auto& lookups = lut_to_lookups_[lut];
        if (lookups.empty()) throw std::runtime_error("LUT is unused");
        u32 last_lu_gate = (u32)num_gates();
        u32 row = 0, slot = LU_SLOTS;
        for (auto& pr : lookups) {
            if (slot == LU_SLOTS) {
                row = add_gate(G_LOOKUP, (u64)lut, 0);
                slot = 0;
            }
            Target gin = wire_target(row, 2 * slot), gout = wire_target(row, 2 * slot + 1);
            connect(gin, pr.first);
            connect(gout, pr.second);
            gens_.push_back(Gen{OP_LOOKUP, gout, gin, 0, 0, (u32)lut, 0, 0});
            slot++;
        }
        u32 last_lut_gate = (u32)num_gates();
        u32 num_lut_rows = (u32)((luts_[lut].size() - 1) / LUT_SLOTS + 1);
        for (u32 i = 0; i < num_lut_rows; i++) add_gate(G_LOOKUP_TABLE, (u64)lut, 0);
        u32 first_lut_gate = (u32)num_gates() - 1;
        add_gate(G_NOOP);
        lookup_rows_.push_back(LookupRows{last_lu_gate, last_lut_gate, first_lut_gate});
    }
}

inline Circuit CircuitBuilder::build() {
    Circuit c;
    c.cfg = cfg_;
    const u32 R = cfg_.num_routed_wires;

    // Public inputs: the reference registers none (grep register_public_input -> 0 hits), so the public-input
    // hash is hash_no_pad([]) = 0^4 and needs no PoseidonGate; route four zero constants to a PublicInputGate.
    // Upstream also attaches RandomValueGenerators to the gate's unused wires (randomize_unused_pi_wires);
    // here those wires stay 0 so that proofs are reproducible.
    {
        Target z = zero();
        u32 pi_gate = add_gate(G_PUBLIC_INPUT);
        for (u32 i = 0; i < 4; i++) connect(z, wire_target(pi_gate, i));
    }
    add_all_lookups();
    while (constants_to_targets_.size() > constant_generators_.size()) add_gate(G_CONSTANT);
    {
        size_t k = 0;
        for (auto& kv : constants_to_targets_) {
            auto cg = constant_generators_[k++];
            gate_instances_[cg.first].constants[cg.second] = kv.first;
            Target w = wire_target(cg.first, cg.second);
            connect(w, kv.second);
            gens_.push_back(Gen{OP_CONST, w, 0, 0, 0, 0, kv.first, 0});
        }
    }
    // blind_and_pad.  zk: blinding_counts() rows of random wires for the regular polynomials, and pairs of rows with
    // equal (copy-constrained) random routed wires for the Z polynomials; then pad with NoopGates to a power of two.
    if (cfg_.zero_knowledge) {
        const size_t D = 2, num_gates_now = gate_instances_.size();
        size_t degree_estimate = 1;
        while (degree_estimate < num_gates_now) degree_estimate <<= 1;
        size_t regular = 0, zopen = 0;
        for (;;) {
            u32 db_est = 0;
            while (((size_t)1 << db_est) < degree_estimate) db_est++;
            Circuit tmp;
            tmp.cfg = cfg_;
            tmp.degree_bits = db_est;
            size_t folding = 0, prod = 1;
            for (u32 a : tmp.reduction_arity_bits()) {
                folding += ((size_t)1 << a) - 1;
                prod <<= a;
            }
            size_t final_coeffs = degree_estimate / prod;
            size_t fri_openings = cfg_.num_query_rounds * (1 + D * folding + D * final_coeffs);
            regular = D + fri_openings;
            zopen = 2 * D + fri_openings;
            if (num_gates_now + regular + 2 * zopen <= degree_estimate) break;
            degree_estimate <<= 1;
        }
        for (size_t i = 0; i < regular; i++) c.blind_rows.push_back(add_gate(G_NOOP));
        for (size_t i = 0; i < zopen; i++) {
            u32 g1 = add_gate(G_NOOP), g2 = add_gate(G_NOOP);
            for (u32 w = 0; w < R; w++) connect(wire_target(g1, w), wire_target(g2, w));
            c.blind_zrows.push_back({g1, g2});
        }
    }
    while (gate_instances_.size() < 4 || (gate_instances_.size() & (gate_instances_.size() - 1)) != 0) add_gate(G_NOOP);
    const size_t n = gate_instances_.size();
    u32 db = 0;
    while ((1ull << db) < n) db++;
    c.degree_bits = db;

    // ---- gate types, sorted by (degree, id) as `gates.sort_unstable_by_key(|g| (g.0.degree(), g.0.id()))`; selector
    // polynomials (plonky2 gates/selectors.rs).  A LookupGate's id is "LookupGate {num_slots: 40, lut_hash: [..]}" and a
    // LookupTableGate's "LookupTableGate {num_slots: 26, lut_hash: [..], last_lut_row: R}" with the table's Keccak-256 as a
    // Rust {:?} byte list, so each table contributes two gate types of its own.
    struct GateType {
        u32 kind, lut;
        std::string id;
    };
    std::vector<GateType> types;
    {
        bool present[G_NUM_KINDS] = {false};
        for (auto& g : gate_instances_) present[g.kind] = true;
        for (size_t l = 0; l < luts_.size(); l++) {
            std::vector<uint8_t> bytes;
            for (auto& pr : luts_[l])
                for (u16 v : {pr.first, pr.second}) {
                    bytes.push_back((uint8_t)(v & 0xFF));
                    bytes.push_back((uint8_t)(v >> 8));
                }
            auto h = keccak256(bytes.data(), bytes.size());
            std::string hs = "[";
            for (size_t i = 0; i < h.size(); i++) hs += (i ? ", " : "") + std::to_string((unsigned)h[i]);
            hs += "]";
            types.push_back({G_LOOKUP, (u32)l, "LookupGate {num_slots: " + std::to_string(LU_SLOTS) + ", lut_hash: " + hs + "}"});
            types.push_back({G_LOOKUP_TABLE, (u32)l,
                             "LookupTableGate {num_slots: " + std::to_string(LUT_SLOTS) + ", lut_hash: " + hs + ", last_lut_row: " + std::to_string(lookup_rows_[l].last_lut) + "}"});
        }
        if (present[G_NOOP]) types.push_back({G_NOOP, 0, "NoopGate"});
        if (present[G_CONSTANT]) types.push_back({G_CONSTANT, 0, "ConstantGate { num_consts: 2 }"});
        if (present[G_PUBLIC_INPUT]) types.push_back({G_PUBLIC_INPUT, 0, "PublicInputGate"});
        if (present[G_ARITHMETIC]) types.push_back({G_ARITHMETIC, 0, "ArithmeticGate { num_ops: 20 }"});
        if (present[G_POSEIDON]) types.push_back({G_POSEIDON, 0, "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"});
        std::sort(types.begin(), types.end(), [](const GateType& a, const GateType& b) {
            u32 da = gate_degree(a.kind), db = gate_degree(b.kind);
            return da != db ? da < db : a.id < b.id;
        });
    }
    for (auto& t : types) c.gates.push_back(t.kind);
    const u32 num_gates_kinds = (u32)c.gates.size();
    auto type_index = [&](const GateInstance& g) -> u32 {
        for (u32 i = 0; i < num_gates_kinds; i++)
            if (types[i].kind == g.kind && ((g.kind != G_LOOKUP && g.kind != G_LOOKUP_TABLE) || types[i].lut == (u32)g.constants[0])) return i;
        throw std::runtime_error("gate instance without a gate type");
    };
    const u32 max_degree = cfg_.quotient_degree_factor + 1;
    const u32 max_gate_degree = gate_degree(c.gates.back());
    if (max_gate_degree + num_gates_kinds - 1 <= max_degree) {
        c.groups.push_back({0, num_gates_kinds});
        c.selector_index.assign(num_gates_kinds, 0);
    } else {
        u32 start = 0;
        while (start < num_gates_kinds) {
            u32 size = 0;
            while (start + size < num_gates_kinds && size + gate_degree(c.gates[start + size]) < max_degree) size++;
            c.groups.push_back({start, start + size});
            start += size;
        }
        c.selector_index.resize(num_gates_kinds);
        for (u32 g = 0; g < c.groups.size(); g++)
            for (u32 i = c.groups[g].first; i < c.groups[g].second; i++) c.selector_index[i] = g;
    }
    c.num_gate_constraints = 0;
    for (u32 k : c.gates) c.num_gate_constraints = std::max(c.num_gate_constraints, gate_num_constraints(k));
    c.luts = luts_;
    c.lookup_rows = lookup_rows_;
    for (auto& l : lut_to_lookups_) c.num_lookups.push_back((u32)l.size());
    c.num_lookup_selectors = luts_.empty() ? 0 : 4 + (u32)luts_.size();

    const u32 nsel = c.num_selectors();
    const u32 ncc = c.num_constants_cols();
    c.constants.assign((size_t)ncc * n, 0);
    for (size_t row = 0; row < n; row++) {
        u32 gi = type_index(gate_instances_[row]);
        for (u32 s = 0; s < nsel; s++)
            c.constants[(size_t)s * n + row] = (nsel == 1 || c.selector_index[gi] == s) ? gi : UNUSED_SELECTOR;
        for (u32 k = 0; k < cfg_.num_constants; k++)
            c.constants[(size_t)(nsel + c.num_lookup_selectors + k) * n + row] =
                (gate_instances_[row].kind == G_ARITHMETIC || gate_instances_[row].kind == G_CONSTANT) ? gate_instances_[row].constants[k] : 0;
    }
    // lookup selectors: TransSre, TransLdc, InitSre, LastLdc, then one StartEnd per LUT (selectors_lookup,
    // selector_ends_lookups).
    for (size_t l = 0; l < lookup_rows_.size(); l++) {
        auto lr = lookup_rows_[l];
        u64* col = &c.constants[(size_t)nsel * n];
        for (u32 r = lr.last_lut; r <= lr.first_lut; r++) col[0 * n + r] = 1;
        for (u32 r = lr.last_lu; r < lr.last_lut; r++) col[1 * n + r] = 1;
        col[2 * n + lr.first_lut + 1] = 1;
        col[3 * n + lr.last_lu] = 1;
        col[(4 + l) * n + lr.last_lut] = 1;
    }

    // ---- copy constraints -> partitions (union-find over virtual targets and routed wires) ----
    const u64 V = num_virtual_;
    const u64 N = V + (u64)n * R;
    if (N >= 0xFFFFFFF0ull) throw std::runtime_error("circuit too large for 32-bit node ids");
    auto node = [&](Target t) -> u32 {
        if (is_wire(t)) {
            if (wire_col(t) >= R) throw std::runtime_error("copy constraint on non-routed wire");
            return (u32)(V + (u64)wire_row(t) * R + wire_col(t));
        }
        return (u32)t;
    };
    std::vector<u32> parent(N);
    for (u64 i = 0; i < N; i++) parent[i] = (u32)i;
    auto find = [&](u32 x) {
        u32 r = x;
        while (parent[r] != r) r = parent[r];
        while (parent[x] != r) {
            u32 nx = parent[x];
            parent[x] = r;
            x = nx;
        }
        return r;
    };
    for (auto& cc : copy_constraints_) {
        u32 a = find(node(cc.first)), b = find(node(cc.second));
        if (a != b) parent[std::max(a, b)] = std::min(a, b);
    }

    // ---- sigma: every routed wire goes to the next wire of its partition in (row, column) order, the last to the first
    // (plonk/permutation_argument.rs: `wire_partition` fills each class `for row { for column {..} }`, `get_sigma_map` links
    // neighbours cyclically) ----
    c.k_is.resize(R);
    {
        u64 k = 1;
        for (u32 i = 0; i < R; i++) {
            c.k_is[i] = k;
            k = gl::mul(k, gl::MULT_GEN);
        }
    }
    std::vector<u64> subgroup(n);
    {
        u64 w = gl::root_of_unity((int)db), x = 1;
        for (size_t i = 0; i < n; i++) {
            subgroup[i] = x;
            x = gl::mul(x, w);
        }
    }
    c.sigmas.resize((size_t)R * n);
    {
        const u32 NONE = 0xFFFFFFFFu;
        std::vector<u32> first(N, NONE), last(N, NONE);  // per class: column * n + row of its first / latest wire
        auto sig = [&](u64 from, u64 to) { c.sigmas[from] = gl::mul(c.k_is[to / n], subgroup[to % n]); };
        for (u32 row = 0; row < n; row++)
            for (u32 col = 0; col < R; col++) {
                const u32 idx = (u32)((u64)col * n + row);
                u32 r = find((u32)(V + (u64)row * R + col));
                if (first[r] == NONE)
                    first[r] = idx;
                else
                    sig(last[r], idx);
                last[r] = idx;
            }
        for (u64 r = 0; r < N; r++)
            if (first[r] != NONE) sig(last[r], first[r]);
    }

    // ---- witness program over slots ----
    std::vector<int32_t> slot_of(N, -1);
    u32 num_slots = 0;
    auto slot = [&](Target t) -> u32 {
        u32 r = find(node(t));
        if (slot_of[r] < 0) slot_of[r] = (int32_t)num_slots++;
        return (u32)slot_of[r];
    };
    std::vector<Op> ops(gens_.size());
    for (size_t i = 0; i < gens_.size(); i++) {
        const Gen& g = gens_[i];
        Op o{g.kind, slot(g.out), 0, 0, 0, g.aux, g.k0, g.k1};
        if (g.kind == OP_ARITH) {
            o.a = slot(g.a);
            o.b = slot(g.b);
            o.c = slot(g.c);
        } else if (g.kind == OP_LOOKUP) {
            o.a = slot(g.a);
        } else if (g.kind == OP_EQ || g.kind == OP_EQINV) {
            o.a = slot(g.a);
            o.b = slot(g.b);
        } else if (g.kind == OP_POSEIDON) {
            u32 row = (u32)g.a;
            o.a = row;
            o.aux = (u32)c.poseidon_rows.size();
            c.poseidon_rows.push_back(row);
            for (u32 col = 0; col < R; col++) slot(wire_target(row, col));  // every routed wire of the row carries a value
        }
        ops[i] = o;
    }
    c.vt_slot.resize(V);
    for (u64 v = 0; v < V; v++) c.vt_slot[v] = (int32_t)slot((Target)v);
    c.num_slots = num_slots;
    c.wire_slot.resize((size_t)R * n);
    for (u64 idx = 0; idx < (u64)R * n; idx++) {
        u32 col = (u32)(idx / n), row = (u32)(idx % n);
        c.wire_slot[idx] = slot_of[find((u32)(V + (u64)row * R + col))];
    }
    // levelise (Kahn).  A slot becomes available when its FIRST producer (in this order) has run -- exactly
    // plonky2's "generator fires once its watch list is set"; any further producer of the same slot (two
    // computed values tied by `connect`) runs at a strictly later level and only checks equality.
    {
        const size_t M = ops.size();
        std::vector<u32> num_producers(num_slots, 0);
        std::vector<std::vector<u32>> consumers(num_slots);
        auto wslot = [&](u32 row, u32 col) -> u32 { return (u32)slot_of[find((u32)(V + (u64)row * R + col))]; };
        auto inputs = [&](const Op& o, u32* in) -> int {
            if (o.kind == OP_ARITH) { in[0] = o.a; in[1] = o.b; in[2] = o.c; return 3; }
            if (o.kind == OP_LOOKUP) { in[0] = o.a; return 1; }
            if (o.kind == OP_EQ || o.kind == OP_EQINV) { in[0] = o.a; in[1] = o.b; return 2; }
            if (o.kind == OP_POSEIDON) {
                for (u32 k = 0; k < 12; k++) in[k] = wslot(o.a, PG_IN + k);
                in[12] = wslot(o.a, PG_SWAP);
                return 13;
            }
            return 0;
        };
        auto outputs = [&](const Op& o, u32* out) -> int {
            if (o.kind != OP_POSEIDON) { out[0] = o.out; return 1; }
            int k = 0;
            for (u32 col = PG_OUT; col < R; col++)
                if (col != PG_SWAP) out[k++] = wslot(o.a, col);
            return k;
        };
        for (size_t i = 0; i < M; i++) {
            u32 outs[80];
            int k = outputs(ops[i], outs);
            for (int j = 0; j < k; j++) num_producers[outs[j]]++;
        }
        std::vector<u32> pending(M, 0), level(M, 0), slot_level(num_slots, 0);
        std::vector<uint8_t> slot_ready(num_slots, 0);
        for (size_t i = 0; i < M; i++) {
            u32 in[16];
            int k = inputs(ops[i], in);
            for (int j = 0; j < k; j++) {
                bool dup = false;
                for (int j2 = 0; j2 < j; j2++) dup |= in[j2] == in[j];
                if (dup) continue;
                if (num_producers[in[j]] > 0) {
                    pending[i]++;
                    consumers[in[j]].push_back((u32)i);
                }
            }
        }
        // The inverse hints (OP_EQINV: a 96-multiplication Fermat inversion on one lane) feed only equality checks,
        // never the values a circuit computes with.  Left at their natural level they put an inversion into every
        // level of a sequential chain (inc32's carry chain: four per AES-CTR block); instead they are held back until
        // everything else is scheduled and then run side by side in one level, their checks in the levels after.
        std::vector<u32> queue, deferred;
        auto ready = [&](u32 i) { (ops[i].kind == OP_EQINV ? deferred : queue).push_back(i); };
        for (size_t i = 0; i < M; i++)
            if (pending[i] == 0) ready((u32)i);
        size_t done = 0;
        u32 max_level = 0, floor_level = 0;
        for (;;) {
            while (done < queue.size()) {
                u32 i = queue[done++];
                u32 in[16], outs[80];
                int k = inputs(ops[i], in);
                int ko = outputs(ops[i], outs);
                u32 lv = ops[i].kind == OP_EQINV ? floor_level : 0;
                for (int j = 0; j < k; j++) lv = std::max(lv, slot_level[in[j]]);
                for (int j = 0; j < ko; j++) lv = std::max(lv, slot_level[outs[j]]);
                level[i] = lv;  // ops with only user-set inputs are level 0
                max_level = std::max(max_level, lv);
                for (int j = 0; j < ko; j++) {
                    u32 s = outs[j];
                    slot_level[s] = lv + 1;
                    if (!slot_ready[s]) {
                        slot_ready[s] = 1;
                        for (u32 cns : consumers[s])
                            if (--pending[cns] == 0) ready(cns);
                    }
                }
            }
            if (deferred.empty()) break;
            floor_level = max_level + 1;
            queue.insert(queue.end(), deferred.begin(), deferred.end());
            deferred.clear();
        }
        if (done != M) throw std::runtime_error("witness program has a dependency cycle");
        c.level_offsets.assign(max_level + 2, 0);
        for (size_t i = 0; i < M; i++) c.level_offsets[level[i] + 1]++;
        for (u32 l = 0; l <= max_level; l++) c.level_offsets[l + 1] += c.level_offsets[l];
        std::vector<u32> cursor(c.level_offsets.begin(), c.level_offsets.end() - 1);
        c.ops.resize(M);
        for (size_t i = 0; i < M; i++) c.ops[cursor[level[i]]++] = ops[i];
        if (M == 0) c.level_offsets = {0, 0};
    }
    return c;
}

}  // namespace p2

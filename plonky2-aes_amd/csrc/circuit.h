// Compiled-circuit container shared by the host builder, the GPU prover and the host verifier, plus its
// flat binary encoding (the "blob" taken by p2_circuit_load, include/p2aes.h).
//
// Plays the role of plonky2's CircuitData { prover_only, verifier_only, common } as produced by
// `builder.build::<PoseidonGoldilocksConfig>()` (19 call sites in the reference, e.g.
// aes-gcm/src/circuit_gcm.rs:771, aes-gcm/examples/aes_gcm_128.rs:46) -- minus the constants/sigmas
// commitment, which the prover computes on the device at load time.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace p2 {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;

// CircuitConfig::standard_recursion_config() [EXT, SURVEY.md section 5]; every reference test uses it
// (e.g. aes-gcm/src/circuit_gcm.rs:757).
struct Config {
    u32 num_wires = 135;
    u32 num_routed_wires = 80;
    u32 num_constants = 2;
    u32 num_challenges = 2;
    u32 quotient_degree_factor = 8;
    u32 rate_bits = 3;
    u32 cap_height = 4;
    u32 pow_bits = 16;
    u32 num_query_rounds = 28;
    u32 arity_bits = 4;       // FriReductionStrategy::ConstantArityBits(4, 5)
    u32 final_poly_bits = 5;
    u32 zero_knowledge = 0;   // standard_recursion_zk_config(): blinding rows + salted Merkle leaves
};
static const u32 SALT_SIZE = 4;  // plonky2 fri::oracle SALT_SIZE: random elements appended to every leaf of a blinded oracle

// Blinding randomness.  Upstream draws every blinding element from the OS RNG.  Here the handle holds a 256-bit key
// (four field elements, drawn from the OS CSPRNG at p2_circuit_load) and every element is the output of a PRF:
//     zk_block(key, proof, domain, block) = Poseidon(key[0..4] | proof | domain | block | ZK_TAG | 0^4)[0..8]
//     element `index` of (proof, domain)  = zk_block(key, proof, domain, index >> 3)[index & 7]
// (an outer-keyed sponge squeezing its rate; the four capacity words stay hidden, so published salts do not reveal the
// key).  `proof` is a per-handle counter that advances with every proof attempted, so no (key, proof) pair is ever
// reused.  With p2_circuit_set_zk_key the key is fixed and proofs are reproducible: that is what lets the CPU oracle
// (its own restatement in oracle/oracle_prover.h) check zk proofs byte for byte.  Evaluated in zk_prf.h.
enum ZkDomain : u64 { ZK_ROW = 1, ZK_ZROW = 2, ZK_SALT = 3 /* + oracle index */ };
static const u64 ZK_TAG = 0x7a6b5f626c696e64ull;  // "zk_blind"
struct ZkKey {
    u64 k[4];
};

// Gate kinds in plonky2's sort order (degree, id) for the gates the gadget crates instantiate.
enum GateKind : u32 {
    G_LOOKUP = 0,        // LookupGate, 40 (inp,out) slots, degree 0
    G_LOOKUP_TABLE = 1,  // LookupTableGate, 26 (inp,out,mult) slots, degree 0
    G_NOOP = 2,          // NoopGate
    G_CONSTANT = 3,      // ConstantGate{2}, degree 1
    G_PUBLIC_INPUT = 4,  // PublicInputGate, degree 1
    G_ARITHMETIC = 5,    // ArithmeticGate{20 ops}, degree 3
    G_POSEIDON = 6,      // PoseidonGate (one permutation per row, all 135 wires), degree 7, 123 constraints
    G_NUM_KINDS = 7
};
static inline u32 gate_degree(u32 k) { return k == G_POSEIDON ? 7 : k == G_ARITHMETIC ? 3 : (k == G_CONSTANT || k == G_PUBLIC_INPUT) ? 1 : 0; }
static inline u32 gate_num_constraints(u32 k) { return k == G_POSEIDON ? 123 : k == G_ARITHMETIC ? 20 : k == G_CONSTANT ? 2 : k == G_PUBLIC_INPUT ? 4 : 0; }
// PoseidonGate wire layout (plonky2 gates/poseidon.rs): 12 in | 12 out | swap | 4 delta | 3x12 full-round
// S-box inputs | 22 partial-round S-box inputs | 4x12 full-round S-box inputs = 135
static const u32 PG_IN = 0, PG_OUT = 12, PG_SWAP = 24, PG_DELTA = 25, PG_FULL0 = 29, PG_PARTIAL = 65, PG_FULL1 = 87;

static const u32 LU_SLOTS = 40;   // LookupGate::num_slots = num_routed_wires / 2
static const u32 LUT_SLOTS = 26;  // LookupTableGate::num_slots = num_routed_wires / 3
static const u32 ARITH_OPS = 20;  // ArithmeticGate::num_ops = num_routed_wires / 4
static const u32 UNUSED_SELECTOR = 0xFFFFFFFFu;
static const u32 MAX_LUTS = 6, MAX_GATE_TYPES = 2 * MAX_LUTS + 5;  // per table: LookupGate + LookupTableGate; Noop, Constant, PublicInput, Arithmetic, Poseidon

// Witness program: one op per generator, over "slots" (= copy-constraint partitions that carry a value).
enum OpKind : u32 {
    OP_ARITH = 0,   // out = k0*a*b + k1*c            (ArithmeticBaseGenerator)
    OP_CONST = 1,   // out = k0                       (ConstantGenerator)
    OP_LOOKUP = 2,  // out = lut[aux](a), error if a is not a table input (LookupGenerator)
    OP_EQ = 3,      // out = (a == b)                 (EqualityGenerator.equal)
    OP_EQINV = 4,   // out = a==b ? 0 : 1/(a-b)       (EqualityGenerator.inv)
    OP_POSEIDON = 5,  // PoseidonGenerator of gate row `a`: reads wires 0..11 and 24 of the row, writes every other wire of
                      // the row (routed ones through their slots, wires >= 80 into advice block `aux`)
};
struct Op {
    u32 kind, out, a, b, c, aux;
    u64 k0, k1;
};

struct LookupRows {
    u32 last_lu, last_lut, first_lut;  // plonky2 LookupWire {last_lu_gate, last_lut_gate, first_lut_gate}
};

struct Circuit {
    Config cfg;
    u32 degree_bits = 0;
    std::vector<u32> gates;                      // kind of every gate TYPE in plonky2's (degree, id) order; one G_LOOKUP and one
                                                 // G_LOOKUP_TABLE entry per lookup table (their ids carry the table's hash)
    std::vector<u32> selector_index;             // per gate: which selector polynomial
    std::vector<std::pair<u32, u32>> groups;     // per selector: [first gate, last gate)
    u32 num_lookup_selectors = 0;                // 0 or 4 + #luts
    u32 num_gate_constraints = 0;
    // column-major [col][n]; constants = selectors | lookup selectors | gate constants
    std::vector<u64> constants;
    std::vector<u64> sigmas;
    std::vector<u64> k_is;
    std::vector<std::vector<std::pair<u16, u16>>> luts;
    std::vector<LookupRows> lookup_rows;
    std::vector<u32> num_lookups;                // looking pairs per LUT
    // witness program
    u32 num_slots = 0;
    std::vector<Op> ops;                         // sorted by level
    std::vector<u32> level_offsets;              // ops of level l = [level_offsets[l], level_offsets[l+1])
    std::vector<int32_t> vt_slot;                // virtual target index -> slot (or -1)
    std::vector<int32_t> wire_slot;              // [num_routed][n] -> slot (or -1 = unconnected)
    std::vector<u32> poseidon_rows;              // rows holding a PoseidonGate; index = advice block of the row
    // zk blinding (blind_and_pad): NoopGate rows whose 135 wires are random, and pairs of NoopGate rows whose 80
    // routed wires carry the same random value (copy-constrained), for the Z polynomials
    std::vector<u32> blind_rows;
    std::vector<std::pair<u32, u32>> blind_zrows;

    u32 n() const { return 1u << degree_bits; }
    u32 num_selectors() const { return (u32)groups.size(); }
    u32 num_constants_cols() const { return num_selectors() + num_lookup_selectors + cfg.num_constants; }
    u32 num_preprocessed() const { return num_constants_cols() + cfg.num_routed_wires; }
    u32 num_partial_products() const { return (cfg.num_routed_wires + cfg.quotient_degree_factor - 1) / cfg.quotient_degree_factor - 1; }
    u32 num_sldc_polys() const { return luts.empty() ? 0 : (LU_SLOTS + cfg.quotient_degree_factor - 2) / (cfg.quotient_degree_factor - 1); }
    u32 num_lookup_polys() const { return luts.empty() ? 0 : num_sldc_polys() + 1; }
    u32 lut_degree() const { return (LUT_SLOTS + num_sldc_polys() - 1) / num_sldc_polys(); }
    u32 num_zs_pp() const { return cfg.num_challenges * (1 + num_partial_products()); }
    u32 num_zs_cols() const { return num_zs_pp() + cfg.num_challenges * num_lookup_polys(); }
    u32 num_quotient_cols() const { return cfg.num_challenges * cfg.quotient_degree_factor; }
    u32 salt() const { return cfg.zero_knowledge ? SALT_SIZE : 0; }  // extra leaf elements of the wires / zs / quotient oracles
    // FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits).reduction_arity_bits(...)
    std::vector<u32> reduction_arity_bits() const {
        std::vector<u32> r;
        u32 db = degree_bits;
        while (db > cfg.final_poly_bits && db + cfg.rate_bits - cfg.arity_bits >= cfg.cap_height) {
            r.push_back(cfg.arity_bits);
            db -= cfg.arity_bits;
        }
        return r;
    }
};

// ------------------------------------------------------------------ blob encoding
// Little-endian; header "P2AESCIR" u32 version; then fixed scalars and length-prefixed arrays in the order below.
struct BlobWriter {
    std::vector<uint8_t> buf;
    void raw(const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p;
        buf.insert(buf.end(), b, b + n);
    }
    void w32(u32 v) { raw(&v, 4); }
    void w64(u64 v) { raw(&v, 8); }
    template <class T>
    void vec(const std::vector<T>& v) {
        w64(v.size());
        if (!v.empty()) raw(v.data(), v.size() * sizeof(T));
    }
};
struct BlobReader {
    const uint8_t* p;
    size_t len, pos = 0;
    BlobReader(const void* d, size_t l) : p((const uint8_t*)d), len(l) {}
    void raw(void* out, size_t n) {
        if (pos + n > len) throw std::runtime_error("circuit blob truncated");
        memcpy(out, p + pos, n);
        pos += n;
    }
    u32 r32() { u32 v; raw(&v, 4); return v; }
    u64 r64() { u64 v; raw(&v, 8); return v; }
    template <class T>
    void vec(std::vector<T>& v) {
        u64 n = r64();
        if (n * sizeof(T) > len - pos) throw std::runtime_error("circuit blob truncated (array)");
        v.resize(n);
        if (n) raw(v.data(), n * sizeof(T));
    }
};

static const char BLOB_MAGIC[8] = {'P', '2', 'A', 'E', 'S', 'C', 'I', 'R'};
static const u32 BLOB_VERSION = 4;  // 4: one gate type per lookup table (repeated G_LOOKUP / G_LOOKUP_TABLE entries in `gates`), row-major sigma cycles

static inline std::vector<uint8_t> serialize(const Circuit& c) {
    BlobWriter w;
    w.raw(BLOB_MAGIC, 8);
    w.w32(BLOB_VERSION);
    w.raw(&c.cfg, sizeof(Config));
    w.w32(c.degree_bits);
    w.vec(c.gates);
    w.vec(c.selector_index);
    w.vec(c.groups);
    w.w32(c.num_lookup_selectors);
    w.w32(c.num_gate_constraints);
    w.vec(c.constants);
    w.vec(c.sigmas);
    w.vec(c.k_is);
    w.w32((u32)c.luts.size());
    for (auto& l : c.luts) w.vec(l);
    w.vec(c.lookup_rows);
    w.vec(c.num_lookups);
    w.w32(c.num_slots);
    w.vec(c.ops);
    w.vec(c.level_offsets);
    w.vec(c.vt_slot);
    w.vec(c.wire_slot);
    w.vec(c.poseidon_rows);
    w.vec(c.blind_rows);
    w.vec(c.blind_zrows);
    return w.buf;
}

static inline Circuit deserialize(const void* data, size_t len) {
    BlobReader r(data, len);
    char magic[8];
    r.raw(magic, 8);
    if (memcmp(magic, BLOB_MAGIC, 8) != 0) throw std::runtime_error("bad circuit blob magic");
    if (r.r32() != BLOB_VERSION) throw std::runtime_error("unsupported circuit blob version");
    Circuit c;
    r.raw(&c.cfg, sizeof(Config));
    c.degree_bits = r.r32();
    if (c.degree_bits < 2 || c.degree_bits > 26) throw std::runtime_error("degree_bits out of range");
    if (c.cfg.num_wires != 135 || c.cfg.num_routed_wires != 80 || c.cfg.num_constants != 2 || c.cfg.num_challenges != 2 ||
        c.cfg.quotient_degree_factor != 8 || c.cfg.rate_bits != 3 || c.cfg.cap_height != 4 || c.cfg.pow_bits == 0 || c.cfg.pow_bits > 32 ||
        c.cfg.num_query_rounds == 0 || c.cfg.num_query_rounds > 64 || c.cfg.arity_bits != 4 || c.cfg.final_poly_bits != 5 || c.cfg.zero_knowledge > 1)
        throw std::runtime_error("unsupported circuit config (only standard_recursion[_zk]_config)");
    r.vec(c.gates);
    r.vec(c.selector_index);
    r.vec(c.groups);
    c.num_lookup_selectors = r.r32();
    c.num_gate_constraints = r.r32();
    r.vec(c.constants);
    r.vec(c.sigmas);
    r.vec(c.k_is);
    u32 nl = r.r32();
    if (nl > MAX_LUTS) throw std::runtime_error("too many lookup tables");
    c.luts.resize(nl);
    for (auto& l : c.luts) r.vec(l);
    r.vec(c.lookup_rows);
    r.vec(c.num_lookups);
    c.num_slots = r.r32();
    r.vec(c.ops);
    r.vec(c.level_offsets);
    r.vec(c.vt_slot);
    r.vec(c.wire_slot);
    r.vec(c.poseidon_rows);
    r.vec(c.blind_rows);
    r.vec(c.blind_zrows);
    // shape checks: everything a kernel indexes with is validated here, once.
    size_t n = c.n();
    if (c.gates.empty() || c.gates.size() > MAX_GATE_TYPES || c.groups.empty() || c.groups.size() > c.gates.size()) throw std::runtime_error("gate list");
    for (size_t i = 0; i < c.gates.size(); i++)
        if (c.gates[i] >= G_NUM_KINDS || (i && (c.gates[i] < c.gates[i - 1] || (c.gates[i] == c.gates[i - 1] && c.gates[i] > G_LOOKUP_TABLE)))) throw std::runtime_error("gate kinds");
    for (auto& g : c.groups)
        if (g.first >= g.second || g.second > c.gates.size()) throw std::runtime_error("selector groups");
    for (size_t i = 0; i < c.selector_index.size(); i++)
        if (c.selector_index[i] >= c.groups.size() || i < c.groups[c.selector_index[i]].first || i >= c.groups[c.selector_index[i]].second)
            throw std::runtime_error("selector index");
    if (c.num_lookup_selectors != (c.luts.empty() ? 0 : 4 + c.luts.size())) throw std::runtime_error("lookup selector count");
    {
        u32 want = 0;
        for (u32 k : c.gates) want = std::max(want, gate_num_constraints(k));
        if (c.num_gate_constraints != want) throw std::runtime_error("gate constraint count");
    }
    for (auto& l : c.luts)
        if (l.empty() || l.size() > 65536) throw std::runtime_error("lookup table size");
    for (size_t i = 1; i < c.level_offsets.size(); i++)
        if (c.level_offsets[i] < c.level_offsets[i - 1]) throw std::runtime_error("level offsets not monotone");
    if (!c.level_offsets.empty() && c.level_offsets[0] != 0) throw std::runtime_error("level offsets");
    for (u64 v : c.k_is)
        if (v >= 0xFFFFFFFF00000001ull) throw std::runtime_error("non-canonical k_i");
    if (c.constants.size() != (size_t)c.num_constants_cols() * n) throw std::runtime_error("constants shape");
    if (c.sigmas.size() != (size_t)c.cfg.num_routed_wires * n) throw std::runtime_error("sigmas shape");
    if (c.wire_slot.size() != (size_t)c.cfg.num_routed_wires * n) throw std::runtime_error("wire_slot shape");
    if (c.k_is.size() != c.cfg.num_routed_wires) throw std::runtime_error("k_is shape");
    if (c.lookup_rows.size() != c.luts.size() || c.num_lookups.size() != c.luts.size()) throw std::runtime_error("lut shape");
    if (c.selector_index.size() != c.gates.size()) throw std::runtime_error("selector shape");
    for (size_t l = 0; l < c.lookup_rows.size(); l++) {
        auto& lr = c.lookup_rows[l];
        if (!(lr.last_lu < lr.last_lut && lr.last_lut <= lr.first_lut && (size_t)lr.first_lut + 1 < n)) throw std::runtime_error("lookup rows");
        if ((size_t)(lr.first_lut - lr.last_lut + 1) != (c.luts[l].size() + LUT_SLOTS - 1) / LUT_SLOTS) throw std::runtime_error("lookup table rows");
        if ((size_t)(lr.last_lut - lr.last_lu) != ((size_t)c.num_lookups[l] + LU_SLOTS - 1) / LU_SLOTS || c.num_lookups[l] == 0) throw std::runtime_error("lookup gate rows");
    }
    for (auto& o : c.ops) {
        if (o.out >= c.num_slots) throw std::runtime_error("op out slot");
        if (o.kind == OP_ARITH && (o.a >= c.num_slots || o.b >= c.num_slots || o.c >= c.num_slots)) throw std::runtime_error("op in slot");
        if (o.kind == OP_LOOKUP && (o.a >= c.num_slots || o.aux >= c.luts.size())) throw std::runtime_error("lookup op");
        if ((o.kind == OP_EQ || o.kind == OP_EQINV) && (o.a >= c.num_slots || o.b >= c.num_slots)) throw std::runtime_error("eq op");
        if (o.kind == OP_POSEIDON && (o.a >= n || o.aux >= c.poseidon_rows.size() || c.poseidon_rows[o.aux] != o.a)) throw std::runtime_error("poseidon op");
        if (o.kind > OP_POSEIDON) throw std::runtime_error("op kind");
    }
    for (auto s : c.wire_slot)
        if (s >= (int32_t)c.num_slots) throw std::runtime_error("wire slot range");
    for (u32 row : c.blind_rows)
        if (row >= n) throw std::runtime_error("blinding row");
    for (auto pr : c.blind_zrows)
        if (pr.first >= n || pr.second >= n) throw std::runtime_error("blinding row pair");
    for (u32 row : c.poseidon_rows) {
        if (row >= n) throw std::runtime_error("poseidon row");
        for (u32 col = 0; col < c.cfg.num_routed_wires; col++)
            if (c.wire_slot[(size_t)col * n + row] < 0) throw std::runtime_error("poseidon row wire without slot");
    }
    for (auto s : c.vt_slot)
        if (s >= (int32_t)c.num_slots) throw std::runtime_error("vt slot range");
    if (c.level_offsets.empty() || c.level_offsets.back() != c.ops.size()) throw std::runtime_error("level offsets");
    return c;
}

}  // namespace p2

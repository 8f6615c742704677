// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_field.h header).  C entry points for ctypes: tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg are the only permitted callers.
#include <omp.h>

#include "oracle_aes.h"
#include "oracle_prover.h"

using namespace orc;

struct Handle {
    OCircuit* c;
    Trace trace;
    std::string err;
};

extern "C" {

void* orc_circuit_load(const void* blob, size_t len) {
    try {
        Handle* h = new Handle();
        h->c = load_circuit(blob, len);
        return h;
    } catch (std::exception& e) {
        fprintf(stderr, "orc_circuit_load: %s\n", e.what());
        return nullptr;
    }
}
void orc_circuit_free(void* hp) {
    Handle* h = (Handle*)hp;
    if (!h) return;
    delete h->c;
    delete h;
}
uint32_t orc_degree_bits(void* hp) { return ((Handle*)hp)->c->degree_bits; }
// soundness tests: kind 1 = add delta to slot a; kind 2 = add delta to wire cell (column a, row b); kind 0 = off
void orc_set_fault(void* hp, int kind, uint64_t a, uint64_t b, uint64_t delta) {
    OCircuit* c = ((Handle*)hp)->c;
    c->fault_kind = kind;
    c->fault_a = a;
    c->fault_b = b;
    c->fault_delta = delta;
}
// structure queries for picking fault sites: gate kind of a row, slot of a routed wire cell, op list
uint32_t orc_row_gate_kind(void* hp, uint32_t row) {
    OCircuit* c = ((Handle*)hp)->c;
    u64 gi = c->constants[(size_t)0 * c->n + row];
    for (size_t s = 0; s < c->groups.size(); s++) {
        u64 v = c->constants[s * c->n + row];
        if (v != 0xFFFFFFFFull) gi = v;
    }
    return gi < c->gates.size() ? c->gates[gi] : 0xFFFFFFFFu;
}
int32_t orc_wire_slot(void* hp, uint32_t col, uint32_t row) {
    OCircuit* c = ((Handle*)hp)->c;
    return (col < c->cfg.num_routed_wires && row < c->n) ? c->wire_slot[(size_t)col * c->n + row] : -1;
}
size_t orc_num_ops(void* hp) { return ((Handle*)hp)->c->ops.size(); }
void orc_get_op(void* hp, size_t i, uint32_t* kind, uint32_t* out) {
    OCircuit* c = ((Handle*)hp)->c;
    *kind = c->ops[i].kind;
    *out = c->ops[i].out;
}
// zk circuits: blinding key (four field elements) and index of the next proof under it
void orc_set_zk_key(void* hp, const uint64_t* key4, uint64_t proof_index) {
    for (int i = 0; i < 4; i++) ((Handle*)hp)->c->zk_key[i] = key4[i] % MODULUS;
    ((Handle*)hp)->c->zk_proof = proof_index;
}
void orc_set_zk(void* hp, uint64_t seed, uint64_t proof_index) {  // the product's p2_circuit_set_zk_seed: key = {seed, 0, 0, 0}
    const uint64_t key[4] = {seed, 0, 0, 0};
    orc_set_zk_key(hp, key, proof_index);
}
// verifier-only data: constants_sigmas_cap (2^cap_height digests) followed by circuit_digest; returns #u64
size_t orc_verifier_data(void* hp, uint64_t* out, size_t cap) {
    Handle* h = (Handle*)hp;
    std::vector<u64> v;
    for (auto& d : h->c->pre.tree.cap())
        for (int i = 0; i < 4; i++) v.push_back(d.e[i]);
    for (int i = 0; i < 4; i++) v.push_back(h->c->circuit_digest.e[i]);
    if (out && cap >= v.size()) memcpy(out, v.data(), v.size() * 8);
    return v.size();
}
// seconds the last orc_prove on this thread spent per stage: witness, commitments, partial products + lookups, quotient,
// openings, FRI (see StageClock in oracle_prover.h)
void orc_last_stage_seconds(double* out6) {
    for (int i = 0; i < 6; i++) out6[i] = g_stage_seconds[i];
}
// status: 0 ok, 1 witness conflict / lookup miss, 2 missing input, 3 zeta in subgroup, -1 buffer too small
int orc_prove(void* hp, const uint64_t* targets, const uint64_t* values, size_t n_in, uint8_t* out, size_t out_cap, size_t* out_len, int want_trace) {
    Handle* h = (Handle*)hp;
    std::vector<uint8_t> proof;
    h->trace.clear();
    int st = prove(*h->c, targets, values, n_in, proof, want_trace ? &h->trace : nullptr);
    if (st) return st;
    *out_len = proof.size();
    if (proof.size() > out_cap) return -1;
    memcpy(out, proof.data(), proof.size());
    return 0;
}
size_t orc_trace_len(void* hp, const char* name) {
    Handle* h = (Handle*)hp;
    auto it = h->trace.find(name);
    return it == h->trace.end() ? 0 : it->second.size();
}
size_t orc_trace_get(void* hp, const char* name, uint64_t* out, size_t cap) {
    Handle* h = (Handle*)hp;
    auto it = h->trace.find(name);
    if (it == h->trace.end()) return 0;
    size_t k = std::min(cap, it->second.size());
    memcpy(out, it->second.data(), k * 8);
    return k;
}
// witness only: wires [num_wires][n] column-major
int orc_generate_witness(void* hp, const uint64_t* targets, const uint64_t* values, size_t n_in, uint64_t* wires_out) {
    Handle* h = (Handle*)hp;
    std::vector<std::vector<u64>> w;
    int st = generate_witness(*h->c, targets, values, n_in, w);
    if (st) return st;
    for (size_t c = 0; c < w.size(); c++) memcpy(wires_out + c * h->c->n, w[c].data(), h->c->n * 8);
    return 0;
}

// ---- primitives for unit parity
uint64_t orc_fmul(uint64_t a, uint64_t b) { return fmul(a, b); }
uint64_t orc_fadd(uint64_t a, uint64_t b) { return fadd(a, b); }
uint64_t orc_fsub(uint64_t a, uint64_t b) { return fsub(a, b); }
uint64_t orc_finv(uint64_t a) { return finv(a); }
void orc_poseidon(uint64_t* st) { poseidon_permute(st); }
void orc_hash_no_pad(const uint64_t* in, size_t n, uint64_t* out4) {
    Digest d = hash_no_pad(in, n);
    memcpy(out4, d.e, 32);
}
void orc_two_to_one(const uint64_t* l, const uint64_t* r, uint64_t* out4) {
    Digest a, b;
    memcpy(a.e, l, 32);
    memcpy(b.e, r, 32);
    Digest d = compress(a, b);
    memcpy(out4, d.e, 32);
}
void orc_fft(uint64_t* a, int bits, int inverse) { fft_inplace(a, bits, inverse != 0); }
// the transform the commitments use: forward, output in bit-reversed order (SIMD butterflies when the fast switch is on)
void orc_fft_bitrev_out(uint64_t* a, int bits) { fft_bitrev_out(a, bits); }
// LDE of one polynomial: coeffs[n] -> values on g*<w_{n<<rate}> written in BIT-REVERSED index order
void orc_lde(const uint64_t* coeffs, int bits, int rate_bits, uint64_t* out) {
    std::vector<u64> c(coeffs, coeffs + ((size_t)1 << bits));
    auto v = coset_fft(c, bits + rate_bits, GENERATOR);
    for (size_t i = 0; i < v.size(); i++) out[rev_bits(i, bits + rate_bits)] = v[i];
}
// Merkle cap over row-major leaves; returns number of digests written
size_t orc_merkle_cap(const uint64_t* leaves, size_t num_leaves, size_t width, int cap_height, uint64_t* out) {
    MerkleTree t = build_merkle(leaves, num_leaves, width, cap_height);
    size_t k = 0;
    for (auto& d : t.cap())
        for (int i = 0; i < 4; i++) out[k++] = d.e[i];
    return t.cap().size();
}
// ---- AES restatement
uint8_t orc_gf_2_8_mul(uint8_t a, uint8_t b) { return orc_aes::gmul(a, b); }
uint8_t orc_sbox(uint8_t x) { return orc_aes::tables().S[x]; }
// out must hold 16*(nk+7) bytes
void orc_aes_expand_key(const uint8_t* key, int nk, uint8_t* out) {
    auto rk = orc_aes::expand_key(key, nk);
    memcpy(out, rk.data(), rk.size());
}
void orc_aes_encrypt_block(const uint8_t* key, int nk, const uint8_t* in, uint8_t* out) {
    auto rk = orc_aes::expand_key(key, nk);
    orc_aes::encrypt_block(rk.data(), nk + 6, in, out);
}
void orc_gf_2_128_mul(const uint8_t* x, const uint8_t* y, uint8_t* out) { orc_aes::gf128_mul(x, y, out); }
void orc_ghash(const uint8_t* h, const uint8_t* x, size_t len, uint8_t* out) { orc_aes::ghash(h, x, len, out); }
void orc_gctr(const uint8_t* key, int nk, const uint8_t* icb, const uint8_t* x, size_t len, uint8_t* y) {
    auto rk = orc_aes::expand_key(key, nk);
    orc_aes::gctr(rk.data(), nk + 6, icb, x, len, y);
}
void orc_gcm_encrypt(const uint8_t* key, int nk, const uint8_t* iv, const uint8_t* pt, size_t len, uint8_t* ct, uint8_t* tag) {
    orc_aes::gcm_encrypt(key, nk, iv, pt, len, ct, tag);
}
int orc_num_threads() { return omp_get_max_threads(); }
void orc_set_num_threads(int t) { omp_set_num_threads(t); }
// 1: every Poseidon permutation of the oracle runs in the sparse-partial-round form (oracle_poseidon_sparse.h); 0: the
// textbook 30-round loop (the default, and what the parity tests use).  Process-wide.
void orc_set_fast_hash(int on) { g_sparse_poseidon = on != 0; }
void orc_poseidon_sparse(uint64_t* st) { poseidon_permute_sparse(st); }
// most SIMD lanes the fast hash may use (8, 4 or 1); returns the lanes it will use on this host
// test hooks: the lane arithmetic of the SIMD hash on `lanes` arbitrary words per operand (out: 6 x lanes words, see
// PoseidonLanes::test_arith); returns 0 when the host cannot run that many lanes.  Hashing of rows with the fast hash.
int orc_simd_test_arith(int lanes, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out) {
    const int keep = g_simd_cap;
    g_simd_cap = 8;
    const int host = simd_lanes();
    g_simd_cap = keep;
#if defined(__x86_64__) && !defined(ORC_NO_SIMD)
    if (lanes == 8 && host >= 8) return simd512_test_arith(a, b, c, out), 1;
    if (lanes == 4 && host >= 4) return simd256_test_arith(a, b, c, out), 1;
#endif
    (void)host;
    return 0;
}
void orc_batch_inverse(uint64_t* v, size_t n) {
    std::vector<u64> t(v, v + n);
    batch_inverse_parallel(t);
    memcpy(v, t.data(), n * 8);
}
int orc_set_simd_lanes(int lanes) {
    g_simd_cap = lanes;
    return simd_lanes();
}
}

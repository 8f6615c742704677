// Shared by capi_host.cpp and prover_gpu.hip: error slot and circuit-shape helpers.
#pragma once
#include <string>

#include "../../include/p2aes.h"
#include "circuit.h"

namespace p2 {
extern thread_local std::string g_last_error;
inline void set_error(const std::string& s) { g_last_error = s; }

// exact proof size in bytes for the layout written by the prover (see DESIGN.md "Proof layout")
inline size_t proof_words(const Circuit& c, size_t* n_merkle_proofs) {
    size_t cap_n = (size_t)1 << c.cfg.cap_height, NC = c.cfg.num_challenges;
    size_t lde_bits = c.degree_bits + c.cfg.rate_bits;
    auto ar = c.reduction_arity_bits();
    size_t w = 3 * cap_n * 4;
    size_t n_open = c.num_constants_cols() + c.cfg.num_routed_wires + c.cfg.num_wires + 2 * NC + NC * c.num_partial_products() +
                    c.num_quotient_cols() + 2 * NC * c.num_lookup_polys();
    w += 2 * n_open;
    w += ar.size() * cap_n * 4;
    size_t per_q = c.num_preprocessed() + c.cfg.num_wires + c.num_zs_cols() + c.num_quotient_cols() + 3 * c.salt() + 4 * 4 * (lde_bits - c.cfg.cap_height);
    size_t mp = 4;
    size_t bits = lde_bits;
    for (u32 a : ar) {
        bits -= a;
        per_q += 2 * ((size_t)1 << a) + 4 * (bits - c.cfg.cap_height);
        mp++;
    }
    w += c.cfg.num_query_rounds * per_q;
    size_t fl = c.n();
    for (u32 a : ar) fl >>= a;
    w += 2 * fl + 1;
    if (n_merkle_proofs) *n_merkle_proofs = mp * c.cfg.num_query_rounds;
    return w;
}
inline size_t proof_bytes(const Circuit& c) {
    size_t mp;
    size_t w = proof_words(c, &mp);
    return 8 * w + mp;  // one u8 length prefix per Merkle proof
}
inline void fill_info(const Circuit& c, p2_circuit_info* o) {
    o->degree_bits = c.degree_bits;
    o->num_wires = c.cfg.num_wires;
    o->num_routed_wires = c.cfg.num_routed_wires;
    o->num_constants_cols = c.num_constants_cols();
    o->num_zs_cols = c.num_zs_cols();
    o->num_quotient_cols = c.num_quotient_cols();
    o->num_luts = (uint32_t)c.luts.size();
    o->num_ops = (uint32_t)c.ops.size();
    o->num_levels = (uint32_t)c.level_offsets.size() - 1;
    o->num_slots = c.num_slots;
    o->num_virtual_targets = (uint32_t)c.vt_slot.size();
    o->num_fri_rounds = (uint32_t)c.reduction_arity_bits().size();
    o->proof_bytes = proof_bytes(c);
    o->zero_knowledge = c.cfg.zero_knowledge;
    o->num_gate_kinds = (uint32_t)c.gates.size();
}
}  // namespace p2

// Device-side schedule of the witness program (host code, run once per circuit at p2_circuit_load).
//
// The compiled circuit carries its witness program as ops sorted into dependency levels (builder.h): level l holds every op
// whose inputs are ready after level l - 1, and k_witness runs one level per workgroup barrier.  That is the right shape for a
// wide circuit (AES-GCM 1 KiB: 170 k ops in 381 levels), and the wrong one for a deep one: AesGcm128Target<65536> is 10.7 M
// ops in 16.5 k levels, almost all of them the carry chain of inc32 (aes-gcm/src/circuit_gcm.rs:350-368: add, is_equal, mul
// and a two-op select per counter byte, block after block), and a level costs its barrier and its dependent memory round
// trips -- about 4 us -- however few ops it holds (round 2: 69 ms per 8-proof chunk, 8 workgroups on a 256-CU chip).
//
// So the levels are rebuilt here around MACROS: a macro is a short straight-line run of ops that ONE thread executes in
// order, and a level is a set of macros with no dependencies among them.  An op joins the macro of its latest producer when
// every input that becomes ready that late comes from that one macro (its other inputs are older) and the macro is not full;
// otherwise it starts a macro of its own one level later.  Chains contract by the cap K (one barrier per K chain links), wide
// levels lose nothing they need (there are far more macros than threads), and the values computed are the same, op for op:
// every op still reads its operands from, and writes its result to, the slot array; inside a macro the reads follow the
// writes of the same thread in program order, across macros the workgroup barrier of the level orders them.
//
// Semantics kept from the builder's levelisation: a slot is produced by its FIRST producer in program order; any later
// producer of the same slot (two computed values tied by `connect`) depends on it and only checks equality; the inverse hints
// (OP_EQINV) are held back until everything else is scheduled.  PoseidonGenerator ops (a whole gate row per thread) are never
// fused.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>

#include "circuit.h"

namespace p2 {

struct WitnessSchedule {
    std::vector<Op> ops;              // reordered: level by level, macro by macro
    std::vector<u32> macro_offsets;   // macro m = ops[macro_offsets[m], macro_offsets[m + 1])
    std::vector<u32> level_offsets;   // level l = macros [level_offsets[l], level_offsets[l + 1])
    u32 max_macro = 0;
};

// K = most ops per macro (1 = one op per thread per level, the plain levelisation)
inline WitnessSchedule schedule_witness(const Circuit& c, u32 K) {
    const size_t M = c.ops.size(), n = (size_t)1 << c.degree_bits;
    const u32 R = c.cfg.num_routed_wires;
    WitnessSchedule s;
    if (M == 0) {
        s.macro_offsets = {0};
        s.level_offsets = {0, 0};
        return s;
    }
    K = std::max<u32>(K, 1);
    std::vector<int32_t> slot_macro(c.num_slots, -1);  // macro of the slot's first producer
    std::vector<u32> macro_level, macro_size, macro_head, macro_tail;
    std::vector<uint8_t> macro_closed;
    std::vector<u32> next_op(M, ~0u);
    macro_level.reserve(M / 2);
    u32 max_level = 0, floor_level = 0;
    bool seen_inv = false;
    auto wslot = [&](u32 row, u32 col) -> u32 { return (u32)c.wire_slot[(size_t)col * n + row]; };
    u32 deps[96];
    for (size_t i = 0; i < M; i++) {
        const Op& o = c.ops[i];
        int nd = 0, first_out = 0;
        if (o.kind == OP_ARITH) {
            deps[nd++] = o.a, deps[nd++] = o.b, deps[nd++] = o.c;
        } else if (o.kind == OP_LOOKUP) {
            deps[nd++] = o.a;
        } else if (o.kind == OP_EQ || o.kind == OP_EQINV) {
            deps[nd++] = o.a, deps[nd++] = o.b;
        } else if (o.kind == OP_POSEIDON) {
            for (u32 k = 0; k < 12; k++) deps[nd++] = wslot(o.a, PG_IN + k);
            deps[nd++] = wslot(o.a, PG_SWAP);
        }
        first_out = nd;
        if (o.kind == OP_POSEIDON) {
            for (u32 col = PG_OUT; col < R; col++)
                if (col != PG_SWAP) deps[nd++] = wslot(o.a, col);
        } else {
            deps[nd++] = o.out;
        }
        // latest producer level among the inputs and the (possibly already produced) outputs
        int lv = -1;
        int32_t from = -1;
        bool single = true;
        for (int j = 0; j < nd; j++) {
            if (deps[j] >= c.num_slots) throw std::runtime_error("witness op refers to a slot outside the circuit");
            const int32_t pm = slot_macro[deps[j]];
            if (pm < 0) continue;
            const int l = (int)macro_level[pm];
            if (l > lv) {
                lv = l, from = pm, single = true;
            } else if (l == lv && pm != from) {
                single = false;
            }
        }
        if (o.kind == OP_EQINV && !seen_inv) {
            seen_inv = true;
            floor_level = max_level + 1;
        }
        int32_t mine;
        const bool fusable = o.kind != OP_EQINV && o.kind != OP_POSEIDON;
        if (fusable && lv >= 0 && single && !macro_closed[from] && macro_size[from] < K) {
            mine = from;
            next_op[macro_tail[from]] = (u32)i;
            macro_tail[from] = (u32)i;
            macro_size[from]++;
        } else {
            u32 l = (u32)(lv + 1);
            if (o.kind == OP_EQINV) l = std::max(l, floor_level);
            mine = (int32_t)macro_level.size();
            macro_level.push_back(l);
            macro_size.push_back(1);
            macro_head.push_back((u32)i);
            macro_tail.push_back((u32)i);
            macro_closed.push_back(o.kind == OP_POSEIDON || o.kind == OP_EQINV);
            max_level = std::max(max_level, l);
        }
        for (int j = first_out; j < nd; j++)
            if (slot_macro[deps[j]] < 0) slot_macro[deps[j]] = mine;
    }
    const size_t NM = macro_level.size();
    s.level_offsets.assign((size_t)max_level + 2, 0);
    for (size_t m = 0; m < NM; m++) s.level_offsets[macro_level[m] + 1]++;
    for (u32 l = 0; l <= max_level; l++) s.level_offsets[l + 1] += s.level_offsets[l];
    std::vector<u32> cursor(s.level_offsets.begin(), s.level_offsets.end() - 1), order(NM);
    for (size_t m = 0; m < NM; m++) order[cursor[macro_level[m]]++] = (u32)m;  // stable: macros keep their program order inside a level
    s.ops.reserve(M);
    s.macro_offsets.reserve(NM + 1);
    for (size_t k = 0; k < NM; k++) {
        const u32 m = order[k];
        s.macro_offsets.push_back((u32)s.ops.size());
        for (u32 i = macro_head[m]; i != ~0u; i = next_op[i]) s.ops.push_back(c.ops[i]);
        s.max_macro = std::max(s.max_macro, macro_size[m]);
    }
    s.macro_offsets.push_back((u32)s.ops.size());
    if (s.ops.size() != M) throw std::runtime_error("internal: witness schedule lost ops");
    return s;
}

}  // namespace p2

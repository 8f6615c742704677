// Device-side schedule of the witness program (host code, run once per circuit at p2_circuit_load).
//
// The compiled circuit carries its witness program as ops sorted into dependency levels (builder.h), and k_witness runs one
// level per workgroup barrier, one workgroup per proof.  Two things bound that kernel, and they are different things:
//   * DEPTH: a level costs its chain of dependent memory round trips plus the barrier -- about 4 us -- however few ops it
//     holds.  AesGcm128Target<65536> is 16.5 k levels deep, almost all of it the carry chain of inc32
//     (aes-gcm/src/circuit_gcm.rs:350-368: add, is_equal, mul and a two-op select per counter byte, block after block);
//   * WIDTH: a thread works on one op per round trip, so 10.7 M ops on 512 threads are 20.8 k trips of ~3.3 us.
// Round 2's kernel sat on both at once (69 ms per 8-proof chunk either way), which is why neither wider workgroups nor more
// ops in flight per thread alone moved it.
//
// The schedule built here takes the depth apart.  Only 0.6 % of the ops are on or next to the critical path: the counter
// chain.  Those -- ops with no slack whose kind needs no table lookup -- are fused into CHAINS: short straight-line runs (at
// most `max_chain` ops) that ONE thread executes in order, reading every operand the chain does not produce itself in a
// single batch of loads and forwarding the rest through registers (k_witness, witness_exec_chain).  Everything else stays a
// SINGLE op, scheduled as early as its inputs allow against the contracted chain.  64 KiB circuit: 12.4 k levels unfused,
// 2.6 k with chains of 8, with 63 k of 10.7 M ops fused; the singles of a level are then wide enough (4 k on average) for every
// thread to keep several in flight.
//
// Semantics kept from the builder's levelisation: a slot is produced by its FIRST producer in program order; any later
// producer of the same slot (two computed values tied by `connect`) depends on it and only checks equality; the inverse hints
// (OP_EQINV) are held back until everything else is scheduled.  Values computed, checks made and statuses reported are the
// same, op for op.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>

#include "circuit.h"

namespace p2 {

struct WLevel {
    u32 chain_begin, chain_count;    // chains [chain_begin, chain_begin + chain_count) of WitnessSchedule::chains
    u32 single_begin, single_end;    // single ops ops[single_begin, single_end)
};
struct WChain {
    u32 start, count;                // ops[start, start + count), executed in order by one thread
};
struct WitnessSchedule {
    std::vector<Op> ops;             // reordered: level by level; inside a level the chains' ops, then the singles
    std::vector<WLevel> levels;
    std::vector<WChain> chains;
    u32 max_chain = 0;
    size_t fused_ops = 0;
};

// max_chain = most ops per chain (1 = no fusion: the plain levelisation); slack = how far from the critical path (in levels
// of the unfused schedule) an op may be and still be fused
inline WitnessSchedule schedule_witness(const Circuit& c, u32 max_chain, u32 slack = 0) {
    const size_t M = c.ops.size(), n = (size_t)1 << c.degree_bits;
    const u32 R = c.cfg.num_routed_wires;
    WitnessSchedule s;
    if (M == 0) {
        s.levels.push_back(WLevel{0, 0, 0, 0});
        return s;
    }
    max_chain = std::max<u32>(max_chain, 1);
    auto wslot = [&](u32 row, u32 col) -> u32 { return (u32)c.wire_slot[(size_t)col * n + row]; };
    // operands, then outputs, of an op
    auto slots_of = [&](const Op& o, u32* d, int& first_out) {
        int nd = 0;
        if (o.kind == OP_ARITH) {
            d[nd++] = o.a, d[nd++] = o.b, d[nd++] = o.c;
        } else if (o.kind == OP_LOOKUP) {
            d[nd++] = o.a;
        } else if (o.kind == OP_EQ || o.kind == OP_EQINV) {
            d[nd++] = o.a, d[nd++] = o.b;
        } else if (o.kind == OP_POSEIDON) {
            for (u32 k = 0; k < 12; k++) d[nd++] = wslot(o.a, PG_IN + k);
            d[nd++] = wslot(o.a, PG_SWAP);
        }
        first_out = nd;
        if (o.kind == OP_POSEIDON) {
            for (u32 col = PG_OUT; col < R; col++)
                if (col != PG_SWAP) d[nd++] = wslot(o.a, col);
        } else {
            d[nd++] = o.out;
        }
        for (int j = 0; j < nd; j++)
            if (d[j] >= c.num_slots) throw std::runtime_error("witness op refers to a slot outside the circuit");
        return nd;
    };
    u32 d[96];
    int fo;
    // ---- pass A: as-soon-as-possible levels of the unfused program, then as-late-as-possible ones: slack = alap - asap
    std::vector<int32_t> producer(c.num_slots, -1);  // first producer (op index) of a slot
    std::vector<u32> asap(M), alap;
    u32 depth = 0, floor_level = 0;
    bool seen_inv = false;
    for (size_t i = 0; i < M; i++) {
        const int nd = slots_of(c.ops[i], d, fo);
        int lv = -1;
        for (int j = 0; j < nd; j++)
            if (producer[d[j]] >= 0) lv = std::max(lv, (int)asap[producer[d[j]]]);
        if (c.ops[i].kind == OP_EQINV && !seen_inv) {
            seen_inv = true;
            floor_level = depth + 1;
        }
        u32 l = (u32)(lv + 1);
        if (c.ops[i].kind == OP_EQINV) l = std::max(l, floor_level);
        asap[i] = l;
        if (!seen_inv) depth = std::max(depth, l);
        for (int j = fo; j < nd; j++)
            if (producer[d[j]] < 0) producer[d[j]] = (int32_t)i;
    }
    // `depth` = last level before the inverse hints; ops at or behind the hints (asap > depth) are never fused
    alap.assign(M, depth);
    if (max_chain > 1) {
        for (size_t i = M; i-- > 0;) {
            if (asap[i] > depth) {
                alap[i] = asap[i];
                continue;  // the deferred tail does not constrain what feeds it
            }
            const int nd = slots_of(c.ops[i], d, fo);
            const u32 mine = alap[i];
            for (int j = 0; j < nd; j++) {
                const int32_t p = producer[d[j]];
                if (p >= 0 && (size_t)p != i) alap[p] = std::min(alap[p], mine ? mine - 1 : 0);
            }
        }
    }
    // ---- pass B: levels again, with near-critical lookup-free ops joining the chain of their latest producer
    std::vector<int32_t> slot_unit(c.num_slots, -1);  // unit (chain or single) of the slot's first producer
    std::vector<u32> unit_level, unit_size, unit_head, unit_tail;
    std::vector<uint8_t> unit_open;
    std::vector<u32> next_op(M, ~0u);
    unit_level.reserve(M);
    u32 max_level = 0;
    floor_level = 0;
    seen_inv = false;
    for (size_t i = 0; i < M; i++) {
        const Op& o = c.ops[i];
        const int nd = slots_of(o, d, fo);
        int lv = -1;
        int32_t from = -1;
        bool single_source = true;
        for (int j = 0; j < nd; j++) {
            const int32_t pu = slot_unit[d[j]];
            if (pu < 0) continue;
            const int l = (int)unit_level[pu];
            if (l > lv) {
                lv = l, from = pu, single_source = true;
            } else if (l == lv && pu != from) {
                single_source = false;
            }
        }
        if (o.kind == OP_EQINV && !seen_inv) {
            seen_inv = true;
            floor_level = max_level + 1;
        }
        const bool chainable = (o.kind == OP_ARITH || o.kind == OP_CONST || o.kind == OP_EQ) && asap[i] <= depth && alap[i] - asap[i] <= slack;
        int32_t mine;
        if (max_chain > 1 && chainable && lv >= 0 && single_source && unit_open[from] && unit_size[from] < max_chain) {
            mine = from;
            next_op[unit_tail[from]] = (u32)i;
            unit_tail[from] = (u32)i;
            unit_size[from]++;
            s.fused_ops++;
        } else {
            u32 l = (u32)(lv + 1);
            if (o.kind == OP_EQINV) l = std::max(l, floor_level);
            mine = (int32_t)unit_level.size();
            unit_level.push_back(l);
            unit_size.push_back(1);
            unit_head.push_back((u32)i);
            unit_tail.push_back((u32)i);
            unit_open.push_back(chainable);  // a chain starts at, and only holds, chainable ops
            max_level = std::max(max_level, l);
        }
        for (int j = fo; j < nd; j++)
            if (slot_unit[d[j]] < 0) slot_unit[d[j]] = mine;
    }
    // ---- lay out: per level the chains (units of more than one op), then the singles, both in program order
    const size_t NU = unit_level.size();
    std::vector<u32> n_chain(max_level + 1, 0), n_chain_ops(max_level + 1, 0), n_single(max_level + 1, 0);
    for (size_t u = 0; u < NU; u++) {
        if (unit_size[u] > 1)
            n_chain[unit_level[u]]++, n_chain_ops[unit_level[u]] += unit_size[u];
        else
            n_single[unit_level[u]]++;
    }
    s.levels.resize((size_t)max_level + 1);
    u32 op_cursor = 0, chain_cursor = 0;
    std::vector<u32> chain_op_cur(max_level + 1), chain_cur(max_level + 1), single_cur(max_level + 1);
    for (u32 l = 0; l <= max_level; l++) {
        s.levels[l].chain_begin = chain_cursor;
        s.levels[l].chain_count = n_chain[l];
        chain_cur[l] = chain_cursor;
        chain_op_cur[l] = op_cursor;
        chain_cursor += n_chain[l];
        op_cursor += n_chain_ops[l];
        s.levels[l].single_begin = op_cursor;
        single_cur[l] = op_cursor;
        op_cursor += n_single[l];
        s.levels[l].single_end = op_cursor;
    }
    if (op_cursor != M) throw std::runtime_error("internal: witness schedule lost ops");
    s.ops.resize(M);
    s.chains.resize(chain_cursor);
    for (size_t u = 0; u < NU; u++) {
        const u32 l = unit_level[u];
        if (unit_size[u] > 1) {
            s.chains[chain_cur[l]++] = WChain{chain_op_cur[l], unit_size[u]};
            for (u32 i = unit_head[u]; i != ~0u; i = next_op[i]) s.ops[chain_op_cur[l]++] = c.ops[i];
            s.max_chain = std::max(s.max_chain, unit_size[u]);
        } else {
            s.ops[single_cur[l]++] = c.ops[unit_head[u]];
        }
    }
    return s;
}

}  // namespace p2

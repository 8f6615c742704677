// PoseidonGate: witness (PoseidonGenerator) and the 123 constraints of eval_unfiltered, written once over an abstract
// field so that the GPU quotient kernel (base field, per LDE point) and the host verifier (GF(p^2), at zeta) share it.
// plonky2 gates/poseidon.rs [EXT, restated]; wire layout in circuit.h (PG_*).  The partial-round S-box wires hold the
// S-box inputs, which are the same values in plonky2's "fast" partial-round schedule and in the naive schedule used here.
#pragma once
#include "circuit.h"
#include "gl.h"

namespace p2 {

struct FBase {
    typedef gl::u64 T;
    GL_HD static T add(T a, T b) { return gl::add(a, b); }
    GL_HD static T sub(T a, T b) { return gl::sub(a, b); }
    GL_HD static T mul(T a, T b) { return gl::mul(a, b); }
    GL_HD static T cst(gl::u64 c) { return c; }
    GL_HD static T mul_base(T a, gl::u64 c) { return gl::mul(a, c); }
};
struct FExt {
    typedef gl::E2 T;
    GL_HD static T add(T a, T b) { return gl::add(a, b); }
    GL_HD static T sub(T a, T b) { return gl::sub(a, b); }
    GL_HD static T mul(T a, T b) { return gl::mul(a, b); }
    GL_HD static T cst(gl::u64 c) { return gl::e2(c, 0); }
    GL_HD static T mul_base(T a, gl::u64 c) { return gl::mul(a, c); }
};

template <class F>
GL_HD void pg_sbox_layer(typename F::T* st) {
    for (int i = 0; i < 12; i++) {
        typename F::T x = st[i], x2 = F::mul(x, x), x3 = F::mul(x2, x), x4 = F::mul(x2, x2);
        st[i] = F::mul(x3, x4);
    }
}
template <class F>
GL_HD typename F::T pg_sbox(typename F::T x) {
    typename F::T x2 = F::mul(x, x), x3 = F::mul(x2, x), x4 = F::mul(x2, x2);
    return F::mul(x3, x4);
}
template <class F>
GL_HD void pg_mds(typename F::T* st) {
    const gl::u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    typename F::T out[12];
    for (int r = 0; r < 12; r++) {
        typename F::T acc = F::cst(0);
        for (int i = 0; i < 12; i++) acc = F::add(acc, F::mul_base(st[(i + r) % 12], C[i]));
        if (r == 0) acc = F::add(acc, F::mul_base(st[0], 8));
        out[r] = acc;
    }
    for (int i = 0; i < 12; i++) st[i] = out[i];
}
template <class F>
GL_HD void pg_constants(typename F::T* st, int round) {
    for (int i = 0; i < 12; i++) st[i] = F::add(st[i], F::cst(gl::poseidon_rc(12 * round + i)));
}

// wire(i) -> value of wire i of the row; emit(k, c) receives constraint k = 0..122 in order.
template <class F, class WireFn, class EmitFn>
GL_HD void poseidon_gate_constraints(WireFn wire, EmitFn emit) {
    typedef typename F::T T;
    int k = 0;
    T swap = wire(PG_SWAP);
    emit(k++, F::mul(swap, F::sub(swap, F::cst(1))));
    T st[12];
    for (int i = 0; i < 4; i++) {
        T lhs = wire(PG_IN + i), rhs = wire(PG_IN + i + 4), d = wire(PG_DELTA + i);
        emit(k++, F::sub(F::mul(swap, F::sub(rhs, lhs)), d));
        st[i] = F::add(lhs, d);
        st[i + 4] = F::sub(rhs, d);
    }
    for (int i = 8; i < 12; i++) st[i] = wire(PG_IN + i);
    int round = 0;
    for (int r = 0; r < 4; r++, round++) {
        pg_constants<F>(st, round);
        if (r != 0)
            for (int i = 0; i < 12; i++) {
                T sin = wire(PG_FULL0 + 12 * (r - 1) + i);
                emit(k++, F::sub(st[i], sin));
                st[i] = sin;
            }
        pg_sbox_layer<F>(st);
        pg_mds<F>(st);
    }
    for (int r = 0; r < 22; r++, round++) {
        pg_constants<F>(st, round);
        T sin = wire(PG_PARTIAL + r);
        emit(k++, F::sub(st[0], sin));
        st[0] = pg_sbox<F>(sin);
        pg_mds<F>(st);
    }
    for (int r = 0; r < 4; r++, round++) {
        pg_constants<F>(st, round);
        for (int i = 0; i < 12; i++) {
            T sin = wire(PG_FULL1 + 12 * r + i);
            emit(k++, F::sub(st[i], sin));
            st[i] = sin;
        }
        pg_sbox_layer<F>(st);
        pg_mds<F>(st);
    }
    for (int i = 0; i < 12; i++) emit(k++, F::sub(st[i], wire(PG_OUT + i)));
}

// PoseidonGenerator: w[0..11] inputs and w[24] swap are given; fills the other 122 wires of the row.
GL_HD void poseidon_gate_witness(gl::u64* w) {
    typedef FBase F;
    gl::u64 st[12], swap = w[PG_SWAP];
    for (int i = 0; i < 4; i++) {
        gl::u64 d = gl::mul(swap, gl::sub(w[PG_IN + i + 4], w[PG_IN + i]));
        w[PG_DELTA + i] = d;
        st[i] = gl::add(w[PG_IN + i], d);
        st[i + 4] = gl::sub(w[PG_IN + i + 4], d);
    }
    for (int i = 8; i < 12; i++) st[i] = w[PG_IN + i];
    int round = 0;
    for (int r = 0; r < 4; r++, round++) {
        pg_constants<F>(st, round);
        if (r != 0)
            for (int i = 0; i < 12; i++) w[PG_FULL0 + 12 * (r - 1) + i] = st[i];
        pg_sbox_layer<F>(st);
        pg_mds<F>(st);
    }
    for (int r = 0; r < 22; r++, round++) {
        pg_constants<F>(st, round);
        w[PG_PARTIAL + r] = st[0];
        st[0] = pg_sbox<F>(st[0]);
        pg_mds<F>(st);
    }
    for (int r = 0; r < 4; r++, round++) {
        pg_constants<F>(st, round);
        for (int i = 0; i < 12; i++) w[PG_FULL1 + 12 * r + i] = st[i];
        pg_sbox_layer<F>(st);
        pg_mds<F>(st);
    }
    for (int i = 0; i < 12; i++) w[PG_OUT + i] = st[i];
}

}  // namespace p2

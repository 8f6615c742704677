// HIP kernels of the proving pipeline (gfx950 / MI355X).  64-bit modular integer work: no MFMA anywhere;
// the levers are coalesced column-major access, LDS-resident butterflies and enough waves to cover latency.
//
// Data layout (all u64, canonical field elements):
//   polynomial batches are COLUMN-MAJOR: values/coeffs [col][n], LDE [col][8n] with the LDE index already
//   bit-reversed (position p holds the value at g*w^rev(p)), i.e. exactly plonky2's Merkle leaf order, so a
//   leaf is the p-th element of every column and a wave reading 64 consecutive leaves of one column issues one
//   512-byte coalesced request.  A batch of proofs adds an outer [proof] dimension (blockIdx.y).
#pragma once
#include <hip/hip_runtime.h>

#include "circuit.h"
#include "witness_schedule.h"
#include "gl.h"
#include "poseidon_fast.h"
#include "poseidon_gate.h"
#include "zk_prf.h"

namespace p2k {
using gl::E2;
using gl::u32;
using gl::u64;

static const u64 UNSET = ~0ull;

// ------------------------------------------------------------------------------------------- Poseidon
// The hashing kernels are VALU-issue bound; left alone the register allocator takes 102 VGPRs (4 waves per SIMD), asked
// for 5 waves it fits in 84 without spilling.
#ifndef P2_HASH_WAVES_PER_EU
#define P2_HASH_WAVES_PER_EU 5
#endif
#define P2_HASH_WAVES __attribute__((amdgpu_waves_per_eu(P2_HASH_WAVES_PER_EU, P2_HASH_WAVES_PER_EU)))
// 12 lanes of state live in registers of ONE thread; one thread = one sponge.  (Leaf hashing has ~10^5..10^6
// independent sponges per tree, so thread-per-sponge already fills the chip with coalesced column reads.)
__device__ __forceinline__ void sponge_absorb_permute(u64* st) { glf::poseidon(st); }

// Leaf digests of a column-major batch: digest[leaf] = hash_or_noop(row leaf of `cols` columns).
// Columns >= active_cols are known-zero (never materialised).
__global__ __launch_bounds__(256) P2_HASH_WAVES void k_hash_leaves(const u64* __restrict__ data, int cols, int active_cols, size_t col_stride,
                                                      size_t batch_stride, size_t num_leaves, u64* __restrict__ digests,
                                                      size_t dig_batch_stride) {
    // (the output address is formed only at the end: nothing but the sponge state, the input pointer and the loop counter
    // stays live across the permutations -- at 5 waves per SIMD every register counts)
    if ((size_t)blockIdx.x * blockDim.x + threadIdx.x >= num_leaves) return;
    const u64* d = data + (size_t)blockIdx.y * batch_stride + ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = 0;
    if (cols <= 4) {
        u64* out = digests + (size_t)blockIdx.y * dig_batch_stride + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
        for (int c = 0; c < 4; c++) out[c] = (c < cols && c < active_cols) ? d[(size_t)c * col_stride] : 0;
        return;
    }
    for (int c0 = 0; c0 < cols; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int c = c0 + k;
            if (c < cols) st[k] = c < active_cols ? d[(size_t)k * col_stride] : 0;
        }
        d += 8 * col_stride;
        glf::poseidon(st);
    }
    u64* out = digests + (size_t)blockIdx.y * dig_batch_stride + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = st[i];
}

// FRI commit-phase leaves: leaf t = 16 consecutive extension values (bit-reversed order), flattened (c0,c1).
// values: two component columns [2][len].
__global__ __launch_bounds__(256) P2_HASH_WAVES void k_hash_fri_leaves(const u64* __restrict__ vals, size_t len, size_t batch_stride, int arity,
                                                          u64* __restrict__ digests, size_t dig_batch_stride) {
    size_t leaf = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t num_leaves = len / arity;
    if (leaf >= num_leaves) return;
    const u64* v = vals + (size_t)blockIdx.y * batch_stride;
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = 0;
    int width = 2 * arity;
    for (int e0 = 0; e0 < width; e0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int e = e0 + k;
            if (e < width) st[k] = v[(size_t)(e & 1) * len + leaf * arity + (e >> 1)];
        }
        glf::poseidon(st);
    }
    u64* out = digests + (size_t)blockIdx.y * dig_batch_stride + leaf * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = st[i];
}

// One Merkle level: parent[i] = two_to_one(child[2i], child[2i+1]).
__global__ __launch_bounds__(256) P2_HASH_WAVES void k_merkle_level(const u64* __restrict__ child, u64* __restrict__ parent, size_t num_parents,
                                                       size_t batch_stride) {
    if ((size_t)blockIdx.x * blockDim.x + threadIdx.x >= num_parents) return;
    u64 st[12];
    {
        const u64* c = child + (size_t)blockIdx.y * batch_stride + 8 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
#pragma unroll
        for (int k = 0; k < 8; k++) st[k] = c[k];
    }
    {
        u64 zero = 0;
        asm("" : "+v"(zero));  // opaque: a known-zero capacity makes the first round a special case and costs 9 more registers
#pragma unroll
        for (int k = 8; k < 12; k++) st[k] = zero;
    }
    glf::poseidon(st);
    u64* o = parent + (size_t)blockIdx.y * batch_stride + 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);  // formed late: see k_hash_leaves
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = st[k];
}

// The TOP of a tree in two launches (small batches only, see merkle_levels): workgroup w owns the subtree under cap node w (16
// per proof) from the level at which that subtree has at most 256 parents -- up to nine levels, 256, 128, ... 1 parents -- with a
// workgroup barrier between levels instead of a kernel boundary.  Those levels hold 2^13 .. 2^4 hashes per proof: launched one
// by one (round 2) each of them costs a whole permutation's latency plus a launch for next to no work.
//   k_merkle_top       levels with more than 32 parents per subtree: one thread per node (glf::poseidon);
//   k_merkle_top_coop  the narrow rest: what such a level costs is ONE permutation's latency, so every permutation is spread over
//                      a 16-lane group (glf::poseidon_coop, the Fiat-Shamir sponge's form: 4.3 k instructions per lane instead of
//                      15.5 k; 26 us instead of 71 us per level).
// Two kernels, not two branches of one: with both forms of the permutation in one kernel the compiler ran out of SGPRs and
// reloaded spilled round constants with v_readlane right in front of the asm blocks that read them -- a VALU write of an SGPR
// followed at once by a VALU read, the hazard its recogniser cannot see inside inline asm (tests/isa_lint.py flagged it on the
// CPU box before it ever ran).
// dig = the tree's digest levels as merkle_build lays them out; level l's nodes start at 4 * (2^(bits+1) - 2^(bits-l+1)).
__global__ __launch_bounds__(256) P2_HASH_WAVES void k_merkle_top(u64* __restrict__ dig, size_t batch_stride, u32 bits, u32 first_level, u32 num_levels) {
    u64* base = dig + (size_t)blockIdx.y * batch_stride;
    u32 P = ((1u << bits) >> (first_level + 1)) / gridDim.x;  // parents of this workgroup at its first level (<= 256)
    for (u32 k = 0, l = first_level; k < num_levels; k++, l++, P >>= 1) {
        if (threadIdx.x < P) {
            const size_t off_c = 4 * (((size_t)2 << bits) - ((size_t)2 << (bits - l))), off_p = 4 * (((size_t)2 << bits) - ((size_t)2 << (bits - l - 1)));
            const size_t idx = (size_t)blockIdx.x * P + threadIdx.x;
            u64 st[12];
            {
                const u64* c = base + off_c + 8 * idx;
#pragma unroll
                for (int q = 0; q < 8; q++) st[q] = c[q];
            }
            {
                u64 zero = 0;
                asm("" : "+v"(zero));  // see k_merkle_level
#pragma unroll
                for (int q = 8; q < 12; q++) st[q] = zero;
            }
            glf::poseidon(st);
            u64* o = base + off_p + 4 * idx;
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] = st[q];
        }
        __syncthreads();  // the parents just written are the next level's children (same workgroup: workgroup-scope ordering is enough)
    }
}
__global__ __launch_bounds__(256) void k_merkle_top_coop(u64* __restrict__ dig, size_t batch_stride, u32 bits, u32 first_level, u32 num_levels) {
    u64* base = dig + (size_t)blockIdx.y * batch_stride;
    u32 P = ((1u << bits) >> (first_level + 1)) / gridDim.x;  // parents of this workgroup at its first level (<= 32 by the launch)
    const u32 g = threadIdx.x >> 4, i = threadIdx.x & 15;
    for (u32 k = 0, l = first_level; k < num_levels; k++, l++, P >>= 1) {
        const size_t off_c = 4 * (((size_t)2 << bits) - ((size_t)2 << (bits - l))), off_p = 4 * (((size_t)2 << bits) - ((size_t)2 << (bits - l - 1)));
        for (u32 first = 0; first < P; first += blockDim.x >> 4) {
            if (first + ((threadIdx.x >> 6) << 2) < P) {  // wave-uniform: this wave's four groups hold at least one live node
                const u32 idx = first + g;
                const bool live = idx < P;
                const size_t node = (size_t)blockIdx.x * P + (live ? idx : 0);
                u64 w = (live && i < 8) ? base[off_c + 8 * node + i] : 0;
                w = glf::poseidon_coop(w, i);
                if (live && i < 4) base[off_p + 4 * node + i] = w;
            }
        }
        __syncthreads();
    }
}

// standalone permutation (parity test entry point)
__global__ void k_poseidon_states(u64* states, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 st[12];
    for (int k = 0; k < 12; k++) st[k] = states[12 * i + k];
    glf::poseidon(st);
    for (int k = 0; k < 12; k++) states[12 * i + k] = st[k];
}

// ------------------------------------------------------------------------------------------- NTT
// One workgroup transforms one column (n = 2^logn <= 2^14 points, 8 B each => up to 128 KiB of the CU's
// 160 KiB LDS), decimation in frequency: natural order in, bit-reversed order out, log n LDS stages.
//   grid.x = columns * cosets, grid.y = proofs.
//   in  : [col][n]           (+ blockIdx.y * in_batch_stride), optionally read through a bit-reversal
//   pre : optional per-coset scale table [coset][n] applied on load (coset shift powers)
//   out : [col][cosets * n]  block `out_block[coset]` of the column, optionally written through a bit-reversal,
//         optionally multiplied by post[coset][i] (natural index i) or the scalar post_scalar.
struct NttArgs {
    const u64* in;
    u64* out;
    const u64* tw;     // w^k, k < n_max/2, for the forward or inverse root
    const u64* pre;    // [cosets][n] or null
    const u64* pre_tw; // half-column form only, or null: [cosets][n/2] pre[coset][i] * tw(i), the coset scale folded into the first stage's twiddle
    const u64* post;   // [cosets][n] or null (indexed by natural output index; needs bitrev_out)
    u64 post_scalar;   // applied when post == null (1 = none)
    size_t in_col_stride, out_col_stride, in_batch_stride, out_batch_stride;
    int logn, log_nmax, cosets;
    int bitrev_in, bitrev_out;
    int in_coset_blocks;  // 1: input column holds `cosets` blocks of n (quotient inverse); 0: one block shared by all cosets
    u32 block_of_coset[8];
};

__global__ __launch_bounds__(1024) void k_ntt_lds(NttArgs a) {
    extern __shared__ __align__(16) u64 lds[];
    const int logn = a.logn;
    const u32 n = 1u << logn;
    const u32 col = blockIdx.x / a.cosets, coset = blockIdx.x % a.cosets;
    const u32 blk = a.block_of_coset[coset];
    const u64* in = a.in + (size_t)blockIdx.y * a.in_batch_stride + (size_t)col * a.in_col_stride + (a.in_coset_blocks ? (size_t)blk * n : 0);
    u64* out = a.out + (size_t)blockIdx.y * a.out_batch_stride + (size_t)col * a.out_col_stride + (size_t)blk * n;
    const u64* pre = a.pre ? a.pre + (size_t)coset * n : nullptr;
    const u64* post = a.post ? a.post + (size_t)coset * n : nullptr;
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        u64 v = in[i];  // coalesced read; the permutation happens on the LDS side
        u32 dst = a.bitrev_in ? (__brev(i) >> (32 - logn)) : i;
        if (pre) v = gl::mul_nb(v, pre[dst]);
        lds[dst] = v;
    }
    __syncthreads();
    const int tw_shift = a.log_nmax - logn;
    int s = logn - 1;
    if (logn & 1) {  // odd number of stages: one radix-2 stage first
        const u32 h = 1u << s;
        for (u32 t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
            u32 pos = t & (h - 1);
            u32 i = ((t >> s) << (s + 1)) | pos;
            u64 x = lds[i], y = lds[i + h];
            u64 w = a.tw[(size_t)(pos << (logn - 1 - s)) << tw_shift];
            lds[i] = gl::add(x, y);
            lds[i + h] = gl::mul_nb(gl::sub(x, y), w);
        }
        __syncthreads();
        s--;
    }
    // radix-4 steps: DIF stages s and s-1 fused -- each thread owns x[i], x[i+q], x[i+h], x[i+h+q] (h = 2^s, q = h/2) for
    // both stages, so the data crosses LDS once per two stages and the barrier count halves
    for (; s >= 1; s -= 2) {
        const u32 h = 1u << s, q = h >> 1;
        for (u32 t = threadIdx.x; t < (n >> 2); t += blockDim.x) {
            u32 pos = t & (q - 1);                       // i mod q
            u32 i = ((t >> (s - 1)) << (s + 1)) | pos;  // bits s and s-1 of i are zero
            u64 a0 = lds[i], a1 = lds[i + q], a2 = lds[i + h], a3 = lds[i + h + q];
            // stage s: twiddle w_{2h}^(i mod h); here i mod h = pos and (i+q) mod h = pos + q
            u64 w0 = a.tw[(size_t)(pos << (logn - 1 - s)) << tw_shift];
            u64 w1 = a.tw[(size_t)((pos + q) << (logn - 1 - s)) << tw_shift];
            u64 b0 = gl::add(a0, a2), b2 = gl::mul_nb(gl::sub(a0, a2), w0);
            u64 b1 = gl::add(a1, a3), b3 = gl::mul_nb(gl::sub(a1, a3), w1);
            // stage s-1: twiddle w_{2q}^(i mod q), the same for both pairs
            u64 w2 = a.tw[(size_t)(pos << (logn - s)) << tw_shift];
            lds[i] = gl::add(b0, b1);
            lds[i + q] = gl::mul_nb(gl::sub(b0, b1), w2);
            lds[i + h] = gl::add(b2, b3);
            lds[i + h + q] = gl::mul_nb(gl::sub(b2, b3), w2);
        }
        __syncthreads();
    }
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        // write position i (coalesced); it receives the element whose natural index is i when bitrev_out
        u32 src = a.bitrev_out ? (__brev(i) >> (32 - logn)) : i;
        u64 v = lds[src];
        if (post)
            v = gl::mul_nb(v, post[i]);
        else if (a.post_scalar != 1)
            v = gl::mul_nb(v, a.post_scalar);
        out[i] = v;
    }
}

// ---- register-blocked transform (2^8 <= n <= 2^14): n / 16 threads, every thread owns 16 points.
// The same decimation-in-frequency butterflies as k_ntt_lds, regrouped so that FOUR stages run in registers between two
// trips through LDS: with h the half-size of a stage, the 16 points {base + t' + r M : r < 16} (M = stride, t' < M) are
// closed under the stages h = 8M, 4M, 2M, M.  n = 2^14 takes 2 + 4 + 4 + 4 stages: the first step comes straight from the
// global loads (x[t + (n/16) k] is both the coalesced load pattern and the M = n/16 layout), three more steps exchange
// through LDS, a last trip puts the data in store order -- 4 barriers and 4 LDS round trips where k_ntt_lds needs 8 and 7,
// and 16 loads + 15 twiddle loads in flight per thread instead of 4 + 3.  rocprofv3 showed k_ntt_lds at 7.3 SIMD-cycles
// per VALU instruction (profiles/r02_valu.json): it waits on barriers and dependent loads, not on arithmetic.
// LDS index i is stored at i + (i >> 4): a thread's 16 consecutive points (last step, M = 1) and the 16-element sub-blocks
// of the M = 16 step then fall on distinct banks.
__device__ __forceinline__ u32 ntt_pad(u32 i) { return i + (i >> 4); }
// The 16 points of a thread are base + r * stride with stride = 1 (base a multiple of 16) or stride >= 16: the padded
// index is affine in r, pad(base) + r * (stride + stride / 16), so one address costs one add instead of five instructions.
struct NttLdsWalk {
    u32 a0, step;  // byte address of point 0, byte step between points
    __device__ __forceinline__ NttLdsWalk(u32 base, u32 stride) : a0(8 * ntt_pad(base)), step(stride >= 16 ? 8 * (stride + (stride >> 4)) : 8 * stride) {}
    __device__ __forceinline__ u64& at(u64* lds, int r) const { return *(u64*)((char*)lds + (a0 + (u32)r * step)); }
};
// The 15 twiddles of a four-stage step with stride 2^m (8 + 4 + 2 + 1 for the stages with half-size 8, 4, 2, 1 times 2^m):
// stage A (h = 2^(m + 3 - A)), pair (r, r + (8 >> A)): w_{2h}^(t' + 2^m (r mod (8 >> A))) = table[(t' + 2^m j) << (tw_log - (m + 4 - A))].
// They are loaded a whole step ahead -- before the LDS exchange that precedes their use -- so that their L2 latency hides
// under the barrier (rocprofv3: the kernel's waves were parked 55 % of the time; a wave's loads retire in order, so the
// only place to hide them is ahead of work the same wave does anyway).
__device__ __forceinline__ void ntt_r16_load_tw(u64* w, const u64* __restrict__ tw, u32 tp, int m, int tw_log, int nstages) {
    // 32-bit byte offsets from the uniform table base (the table is at most 2^14 * 8 B)
#pragma unroll
    for (int A = 0; A < 4; A++) {
        if (A >= nstages) break;
        const int half = 8 >> A, base = 16 - 2 * half, sh = tw_log - (m + 4 - A);
        const u32 off0 = tp << (sh + 3), dj = 1u << (m + sh + 3);
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < half) w[base + j] = *(const u64*)((const char*)tw + (off0 + (u32)j * dj));
    }
}
// stage A of a step on the 16 points x[r] (r <-> index base + t' + r * 2^m), twiddles from ntt_r16_load_tw
template <int A>
__device__ __forceinline__ void ntt_r16_stage(u64* x, const u64* w) {
    constexpr int half = 8 >> A, base = 16 - 2 * half;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r & half) continue;
        const u64 u = x[r], v = x[r + half];
        x[r] = gl::add(u, v);
        x[r + half] = gl::mul_nb(gl::sub(u, v), w[base + r % half]);
    }
}
// the same at stride M = 1 (the last step): t' = 0, so the j = 0 twiddle of every stage is w^0 = 1 and its product is skipped
// (15 of the step's 32 multiplications; the whole last stage)
template <int A>
__device__ __forceinline__ void ntt_r16_stage_m0(u64* x, const u64* w) {
    constexpr int half = 8 >> A, base = 16 - 2 * half;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r & half) continue;
        const u64 u = x[r], v = x[r + half];
        x[r] = gl::add(u, v);
        x[r + half] = (r % half) ? gl::mul_nb(gl::sub(u, v), w[base + r % half]) : gl::sub(u, v);
    }
}
// LDS traffic between the lanes of ONE wave needs no s_barrier: a wave's LDS instructions execute in order; the fences keep
// the compiler from moving the reads above the writes (it sees no alias: a lane reads what OTHER lanes wrote)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// SPLIT: one workgroup transforms HALF a column -- the first stage (h = n/2) is done while loading (every workgroup reads
// both halves and keeps the sums or the twiddled differences), the remaining n/2-point transform runs in 70 KiB of LDS, so
// that two workgroups share a compute unit and the load / store phases of one overlap the butterflies of the other.  (One
// 139 KiB workgroup per compute unit leaves the SIMDs idle while its waves wait on memory: a wave's own prefetch cannot
// help, vmcnt retires in order.)  Natural-order input and bit-reversed output only (the LDE), which is where the time is.
// block id -> work id such that the work ids handled by one XCD (block ids congruent mod 8) are consecutive
__device__ __forceinline__ u32 xcd_swizzle(u32 b, u32 grid) { return (grid & 7) ? b : (b & 7) * (grid >> 3) + (b >> 3); }
// every global load written above this line is issued before any instruction below it (the loads themselves complete in
// order; their consumers wait with s_waitcnt vmcnt(n) as usual)
__device__ __forceinline__ void loads_issued() { asm volatile("" ::: "memory"); }
template <bool SPLIT>
__global__ __launch_bounds__(1024) void k_ntt_r16(NttArgs a) {
    extern __shared__ __align__(16) u64 lds[];
    const int logn = SPLIT ? a.logn - 1 : a.logn;  // size of the transform this workgroup runs in LDS
    const u32 n = 1u << logn, T = n >> 4, t = threadIdx.x;
    // XCD placement: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2), and the 8 cosets (x 2
    // halves) of one column all read the SAME coefficient column.  Left in launch order they land on all eight L2s and the
    // column is fetched eight times (round 2, PMC: read = written, where the algorithm reads 1/8 of what it writes); with
    // the ids of one XCD renumbered consecutively a column's workgroups share one L2 and the column is fetched once.
    const u32 bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const u32 unit = SPLIT ? bid >> 1 : bid, half = SPLIT ? (bid & 1) : 0;
    const u32 col = unit / a.cosets, coset = unit % a.cosets;
    const u32 blk = a.block_of_coset[coset];
    const size_t full = (size_t)1 << a.logn;
    const u64* in = a.in + (size_t)blockIdx.y * a.in_batch_stride + (size_t)col * a.in_col_stride + (a.in_coset_blocks ? (size_t)blk * full : 0);
    u64* out = a.out + (size_t)blockIdx.y * a.out_batch_stride + (size_t)col * a.out_col_stride + (size_t)blk * full + (size_t)half * n;
    const u64* pre = a.pre ? a.pre + (size_t)coset * full : nullptr;
    const u64* post = a.post ? a.post + (size_t)coset * full : nullptr;
    u64 x[16];
    u64 w[15];
    bool tw_in_flight = false;  // SPLIT with the folded table: the first step's twiddles are fetched behind the load stage
    if (SPLIT) {
        // stage h = n_full / 2 on the fly: sums feed the first half of the (bit-reversed) output, differences the second
        if (pre) {
            // pre[j] = s^j: u s^i +- v s^(i+n) = s^i (u +- S v) with S = s^n, so one uniform constant, one table value per
            // point (s^i, or s^i w^i for the differences) and two multiplications instead of four loads and up to three
            const u64 S = pre[n];
            const u64* scale = half ? (a.pre_tw ? a.pre_tw + (size_t)coset * n : nullptr) : pre;
            if (!half || scale) {
                // All 48 loads of the thread are ISSUED before the first product (loads_issued): left to itself the compiler
                // keeps the register count down by loading one or two points, waiting, multiplying, loading the next --
                // sixteen dependent L2 round trips per wave, which is what the kernel spent its time on (rocprofv3: waves
                // parked 58 % of their life, VALU at 63 % of what the hash kernels reach; profiles/r03_sq_wait_counters.txt).
                // The second half of the scale values is fetched behind the first four points: all 48 at once need more
                // than the 128 registers of a four-wave kernel.
                u64 uu[16], vv[16], sc[16];
#pragma unroll
                for (int k = 0; k < 16; k++) uu[k] = in[t + T * k];
#pragma unroll
                for (int k = 0; k < 16; k++) vv[k] = in[t + T * k + n];
#pragma unroll
                for (int k = 0; k < 8; k++) sc[k] = scale[t + T * k];
                loads_issued();
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k == 4) {
                        loads_issued();
#pragma unroll
                        for (int j = 8; j < 16; j++) sc[j] = scale[t + T * j];
                        loads_issued();
                    }
                    if (k == 12) {  // three quarters of the point registers are free again: the first step's twiddles
                        loads_issued();
                        ntt_r16_load_tw(w, a.tw, t, logn - 4, a.log_nmax, 4);
                        loads_issued();
                    }
                    const u64 v = gl::mul_nb(vv[k], S);
                    x[k] = gl::mul_nb(half ? gl::sub(uu[k], v) : gl::add(uu[k], v), sc[k]);
                }
                tw_in_flight = true;
            } else {
                // no folded table (the small transforms of the later FRI rounds): scale and first-stage twiddle separately
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const u32 i = t + T * k;
                    const u64 u = in[i], v = gl::mul_nb(in[i + n], S);
                    x[k] = gl::mul_nb(gl::mul_nb(gl::sub(u, v), pre[i]), a.tw[(size_t)i << (a.log_nmax - a.logn)]);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 i = t + T * k;
                const u64 u = in[i], v = in[i + n];
                x[k] = half ? gl::mul_nb(gl::sub(u, v), a.tw[(size_t)i << (a.log_nmax - a.logn)]) : gl::add(u, v);
            }
        }
    } else if (a.bitrev_in) {
        // position i holds natural index rev(i): coalesced read, permute through LDS
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 i = t + T * k, dst = __brev(i) >> (32 - logn);
            u64 v = in[i];
            if (pre) v = gl::mul_nb(v, pre[dst]);
            lds[ntt_pad(dst)] = v;
        }
        __syncthreads();
        {
            const NttLdsWalk w0(t, T);
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = w0.at(lds, k);
        }
    } else {
        // points, scales and the first step's twiddles: all issued before the first product (loads_issued)
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = in[t + T * k];
        const int rem0 = (logn & 3) ? (logn & 3) : 4;
        if (pre) {
            u64 pv[16];
#pragma unroll
            for (int k = 0; k < 16; k++) pv[k] = pre[t + T * k];
            loads_issued();
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k == 8) {
                    loads_issued();
                    ntt_r16_load_tw(w, a.tw, t, logn - 4, a.log_nmax, rem0);
                    loads_issued();
                }
                x[k] = gl::mul_nb(x[k], pv[k]);
            }
        } else {
            ntt_r16_load_tw(w, a.tw, t, logn - 4, a.log_nmax, rem0);
            loads_issued();
        }
        tw_in_flight = true;
    }
    if (SPLIT) {
        // Four stages straight from the load layout (stride T: h = n/2 .. n/16), ONE exchange across the workgroup, and from
        // there on 2^(logn-4)-point transforms that each live inside one wave's 1024 points (thread t works on block
        // t >> (logN - 4) of 2^logN points; 2^logN <= 1024 and blocks are aligned, so block and thread share t >> 6): every
        // later exchange is between the lanes of a wave and costs no barrier, and the waves of a workgroup drift apart so
        // that one's LDS and memory waits overlap another's butterflies.  (Round 2 first ran all four exchanges through
        // __syncthreads: the waves were parked 55 % of the time.)
        if (!tw_in_flight) ntt_r16_load_tw(w, a.tw, t, logn - 4, a.log_nmax, 4);
        ntt_r16_stage<0>(x, w);
        ntt_r16_stage<1>(x, w);
        ntt_r16_stage<2>(x, w);
        ntt_r16_stage<3>(x, w);
        int logN = logn - 4;                                   // stages left; 8 or 9 here (logn = 12, 13)
        const int rem = (logN & 3) ? (logN & 3) : 4;
        int m = logN - 4;
        u32 tp = t & ((1u << m) - 1);
        ntt_r16_load_tw(w, a.tw, tp, m, a.log_nmax, rem);      // in flight across the barrier
        {
            const NttLdsWalk wr(t, T);
#pragma unroll
            for (int r = 0; r < 16; r++) wr.at(lds, r) = x[r];
        }
        __syncthreads();
        u32 base_idx = ((t >> m) << logN) | tp, stride = 1u << m;
        {
            const NttLdsWalk rd(base_idx, stride);
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = rd.at(lds, r);
        }
        ntt_r16_stage<0>(x, w);
        if (rem >= 2) ntt_r16_stage<1>(x, w);
        if (rem >= 3) ntt_r16_stage<2>(x, w);
        if (rem >= 4) ntt_r16_stage<3>(x, w);
        for (logN -= rem; logN >= 4; logN -= 4) {
            m = logN - 4;
            tp = t & ((1u << m) - 1);
            ntt_r16_load_tw(w, a.tw, tp, m, a.log_nmax, 4);
            {
                const NttLdsWalk wr(base_idx, stride);
#pragma unroll
                for (int r = 0; r < 16; r++) wr.at(lds, r) = x[r];
            }
            wave_lds_sync();
            base_idx = ((t >> m) << logN) | tp;
            stride = 1u << m;
            {
                const NttLdsWalk rd(base_idx, stride);
#pragma unroll
                for (int r = 0; r < 16; r++) x[r] = rd.at(lds, r);
            }
            if (m == 0) {
                ntt_r16_stage_m0<0>(x, w);
                ntt_r16_stage_m0<1>(x, w);
                ntt_r16_stage_m0<2>(x, w);
                ntt_r16_stage_m0<3>(x, w);
            } else {
                ntt_r16_stage<0>(x, w);
                ntt_r16_stage<1>(x, w);
                ntt_r16_stage<2>(x, w);
                ntt_r16_stage<3>(x, w);
            }
        }
        {
            const NttLdsWalk wr(base_idx, stride);
#pragma unroll
            for (int r = 0; r < 16; r++) wr.at(lds, r) = x[r];
        }
        wave_lds_sync();
        // store order, still inside the wave: lane l of wave v takes points 1024 v + l + 64 k (512 contiguous bytes per store)
        const u32 first = ((t >> 6) << 10) | (t & 63);
        const NttLdsWalk fin(first, 64);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            u64 v = fin.at(lds, k);
            if (a.post_scalar != 1) v = gl::mul_nb(v, a.post_scalar);
            out[first + 64 * k] = v;
        }
        return;
    }
    // first step: the leading rem = logn mod 4 (or 4) stages on the whole array, stride M = n / 16
    // (with M = n / 16 stage A has h = n / 2^(A+1): these ARE the first `rem` stages of the whole transform)
    const int rem = (logn & 3) ? (logn & 3) : 4;
    if (!tw_in_flight) ntt_r16_load_tw(w, a.tw, t, logn - 4, a.log_nmax, rem);
    ntt_r16_stage<0>(x, w);
    if (rem >= 2) ntt_r16_stage<1>(x, w);
    if (rem >= 3) ntt_r16_stage<2>(x, w);
    if (rem >= 4) ntt_r16_stage<3>(x, w);
    u32 base_idx = t, stride = T;  // the thread's 16 points are base_idx + stride * r
    for (int logN = logn - rem; logN >= 4; logN -= 4) {
        const int m = logN - 4;
        const u32 tp = t & ((1u << m) - 1);
        ntt_r16_load_tw(w, a.tw, tp, m, a.log_nmax, 4);  // issued before the exchange: in flight across the barrier
        // exchange: write the points back where they live, read the next step's 16
        {
            const NttLdsWalk wr(base_idx, stride);
#pragma unroll
            for (int r = 0; r < 16; r++) wr.at(lds, r) = x[r];
        }
        __syncthreads();
        base_idx = ((t >> m) << logN) | tp;
        stride = 1u << m;
        {
            const NttLdsWalk rd(base_idx, stride);
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = rd.at(lds, r);
        }
        // no barrier here: a thread writes back exactly the 16 locations it read, nobody else touches them in this step
        if (m == 0) {
            ntt_r16_stage_m0<0>(x, w);
            ntt_r16_stage_m0<1>(x, w);
            ntt_r16_stage_m0<2>(x, w);
            ntt_r16_stage_m0<3>(x, w);
        } else {
            ntt_r16_stage<0>(x, w);
            ntt_r16_stage<1>(x, w);
            ntt_r16_stage<2>(x, w);
            ntt_r16_stage<3>(x, w);
        }
    }
    {
        const NttLdsWalk wr(base_idx, stride);
#pragma unroll
        for (int r = 0; r < 16; r++) wr.at(lds, r) = x[r];
    }
    __syncthreads();
    const NttLdsWalk fin(t, T);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 i = t + T * k;
        u64 v = (!SPLIT && a.bitrev_out) ? lds[ntt_pad(__brev(i) >> (32 - logn))] : fin.at(lds, k);
        if (post)
            v = gl::mul_nb(v, post[(size_t)half * n + i]);
        else if (a.post_scalar != 1)
            v = gl::mul_nb(v, a.post_scalar);
        out[i] = v;
    }
}

// (Round 3 measured a radix-8 form of the half-column kernel -- 8 points per thread, 64 bytes of LDS per thread, two 1 024-thread
// workgroups per compute unit = 8 waves per SIMD instead of 4, one more exchange: 21.4 ms against 19.7 ms per 128-proof chunk.
// Occupancy is not what this kernel lacks; the variant was dropped.)

// ---- large transforms (n > 2^14): four-step split n = n1 * n2.  Pass 1 (this kernel): for a tile of T adjacent
// columns j2, the n1-point DIF over the stride-n2 elements x[j1*n2 + j2], then the twiddle w_n^(j2*k1); the result
// for k1 lands in row rev(k1).  Pass 2 is k_ntt_lds on the n1 contiguous rows of n2 points.  Together: natural
// order in, bit-reversed order out, exactly like the single-pass kernel.
struct Pass1Args {
    const u64* in;
    u64* out;
    const u64* tw;   // w^i for i < n (forward or inverse root of order n)
    const u64* pre;  // [cosets][n] scale applied on load, or null
    const u64* out_tw;  // k_ntt_pass1_r16: [n1][n2] w^(rev(row) j2), the output twiddle in output order
    size_t in_col_stride, out_col_stride, in_batch_stride, out_batch_stride;
    int logn, log_n1, log_T, cosets, in_coset_blocks, xcd_swizzle;
    u32 block_of_coset[8];
};
__global__ __launch_bounds__(256) void k_ntt_pass1(Pass1Args a) {
    extern __shared__ __align__(16) u64 lds[];
    const u32 n = 1u << a.logn, n1 = 1u << a.log_n1, T = 1u << a.log_T;
    const int log_n2 = a.logn - a.log_n1;
    const u32 n2 = 1u << log_n2, tiles = n2 >> a.log_T;
    const u32 tile = blockIdx.x % tiles, cc = blockIdx.x / tiles, col = cc / a.cosets, coset = cc % a.cosets;
    const u32 blk = a.block_of_coset[coset];
    const u64* in = a.in + (size_t)blockIdx.y * a.in_batch_stride + (size_t)col * a.in_col_stride + (a.in_coset_blocks ? (size_t)blk * n : 0);
    u64* out = a.out + (size_t)blockIdx.y * a.out_batch_stride + (size_t)col * a.out_col_stride + (size_t)blk * n;
    const u64* pre = a.pre ? a.pre + (size_t)coset * n : nullptr;
    const u32 j2_0 = tile << a.log_T;
    for (u32 e = threadIdx.x; e < (n1 << a.log_T); e += blockDim.x) {
        u32 j1 = e >> a.log_T, t = e & (T - 1);
        size_t idx = (size_t)j1 * n2 + j2_0 + t;
        u64 v = in[idx];
        if (pre) v = gl::mul_nb(v, pre[idx]);
        lds[e] = v;
    }
    __syncthreads();
    for (int s = a.log_n1 - 1; s >= 0; s--) {
        const u32 h = 1u << s;
        for (u32 b = threadIdx.x; b < ((n1 >> 1) << a.log_T); b += blockDim.x) {
            u32 t = b & (T - 1), q = b >> a.log_T;
            u32 pos = q & (h - 1);
            u32 i = ((q >> s) << (s + 1)) | pos;
            u64 x = lds[(i << a.log_T) + t], y = lds[((i + h) << a.log_T) + t];
            u64 w = a.tw[(size_t)(pos << (a.log_n1 - 1 - s)) << log_n2];  // w_n1^(pos * 2^(log_n1-1-s))
            lds[(i << a.log_T) + t] = gl::add(x, y);
            lds[((i + h) << a.log_T) + t] = gl::mul_nb(gl::sub(x, y), w);
        }
        __syncthreads();
    }
    for (u32 e = threadIdx.x; e < (n1 << a.log_T); e += blockDim.x) {
        u32 r = e >> a.log_T, t = e & (T - 1);
        u32 k1 = a.log_n1 ? (__brev(r) >> (32 - a.log_n1)) : 0;
        u32 j2 = j2_0 + t;
        out[(size_t)r * n2 + j2] = gl::mul_nb(lds[e], a.tw[(size_t)k1 * j2]);
    }
}
// ---- pass 1, register-blocked (round 3).  The tile of n1 rows x T columns (4096 elements) is ONE flat array e = row * T + t:
// the n1-point DIF down the rows of every column is then exactly the first log n1 stages of a 4096-point DIF over e -- the
// same pairs (e, e + h), h >= T -- with twiddles that depend on the ROW part of e only.  So the step structure of k_ntt_r16
// carries over: 256 threads own 16 points each, the first log n1 mod 4 stages come straight from the (tile-strided) global
// loads, every further four stages cost one LDS exchange, and a last trip puts the data in store order.  n = 2^19 (n1 = 128):
// 3 + 4 stages, 2 barriers, where k_ntt_pass1 runs 7 radix-2 LDS stages with 7 barriers and gathers its output twiddle
// w^(k1 j2) with a stride of k1 elements (64 cache lines per wave load); here that twiddle comes from a table laid out in
// output order (Pass1Args::out_tw: [row][j2] = w^(rev(row) j2), n entries shared by every column, coset and proof).
__device__ __forceinline__ void ntt_p1_load_tw(u64* w, const u64* __restrict__ tw, u32 tp, int m, int log_T, int tw_log, int nstages) {
#pragma unroll
    for (int A = 0; A < 4; A++) {
        if (A >= nstages) break;
        const int half = 8 >> A, base = 16 - 2 * half, q = m + 4 - A - log_T;  // root of order 2^q over the rows; q >= 1 for every stage that is run
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < half) w[base + j] = tw[(size_t)((tp + ((u32)j << m)) >> log_T) << (tw_log - q)];
    }
}
// MINW = waves per SIMD the register budget is cut for.  With the tile addresses written as per-thread + uniform parts the
// kernel needs 122 VGPRs and no scratch at four waves; the two-wave build (P2AES_PASS1_WAVES=2) is kept as an A/B switch.
// (Before that: 128 VGPRs with 44 bytes of scratch at four waves, 170 and none at two -- and the scratch-free build at half the
// occupancy was 1.5 x slower, 100.8 vs 67.3 ms per 16 proofs at n = 2^19.)
template <int MINW>
__global__ __launch_bounds__(256, MINW) void k_ntt_pass1_r16(Pass1Args a) {
    extern __shared__ __align__(16) u64 lds[];
    const u32 T = 1u << a.log_T, t = threadIdx.x;
    const int log_n2 = a.logn - a.log_n1;
    const u32 n2 = 1u << log_n2, tiles = n2 >> a.log_T;
    // the 8 cosets of one (column, tile) read the same input: one XCD, one L2 (a.xcd_swizzle; off when there is one coset)
    const u32 bid = a.xcd_swizzle ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32 tile = bid % tiles, cc = bid / tiles, col = cc / a.cosets, coset = cc % a.cosets;
    const u32 blk = a.block_of_coset[coset];
    const size_t n = (size_t)1 << a.logn;
    const u64* in = a.in + (size_t)blockIdx.y * a.in_batch_stride + (size_t)col * a.in_col_stride + (a.in_coset_blocks ? (size_t)blk * n : 0);
    u64* out = a.out + (size_t)blockIdx.y * a.out_batch_stride + (size_t)col * a.out_col_stride + (size_t)blk * n;
    const u64* pre = a.pre ? a.pre + (size_t)coset * n : nullptr;
    const u32 j2_0 = tile << a.log_T;
    u64 x[16], w[15];
    const int rem = (a.log_n1 & 3) ? (a.log_n1 & 3) : 4;
    // every load of the first step -- 16 points, their coset scales, the twiddles -- is issued before the first product
    // (loads_issued, see k_ntt_r16: the compiler otherwise loads a point, waits, multiplies, loads the next)
    // (the coset scale s^idx as s^(row n2) s^j2 from the hot corner of the table -- two multiplies instead of an 8 n-byte
    // stream per (column, coset) -- was measured at n = 2^19 and is slower, 67.3 vs 61.7 ms per 16 proofs; likewise the
    // output twiddle below)
    // element e = t + 256 k of the tile sits at index (e >> log_T) n2 + j2_0 + (e & (T - 1)) of the column; with t < 256 that is
    // at(t) + step(k), a per-thread part and a UNIFORM part (T <= 256: the row advances by 256 / T per k; T > 256: 256 k splits
    // into row and column without a carry from t), so the sixteen addresses cost one vector offset and scalar arithmetic
    const u32 at = (t >> a.log_T) * n2 + j2_0 + (t & (T - 1));  // index < n <= 2^22
    auto step = [&](int k) -> u32 { return ((256u * (u32)k) >> a.log_T) * n2 + ((256u * (u32)k) & (T - 1)); };
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = in[at + step(k)];
    if (pre) {
        u64 pv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) pv[k] = pre[at + step(k)];
        loads_issued();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k == 8) {  // half of the scale registers are free again
                loads_issued();
                ntt_p1_load_tw(w, a.tw, t, 8, a.log_T, a.logn, rem);
                loads_issued();
            }
            x[k] = gl::mul_nb(x[k], pv[k]);
        }
    } else {
        ntt_p1_load_tw(w, a.tw, t, 8, a.log_T, a.logn, rem);
        loads_issued();
    }
    ntt_r16_stage<0>(x, w);
    if (rem >= 2) ntt_r16_stage<1>(x, w);
    if (rem >= 3) ntt_r16_stage<2>(x, w);
    if (rem >= 4) ntt_r16_stage<3>(x, w);
    u32 base_idx = t, stride = 256;
    for (int logN = 12 - rem; logN - 4 >= a.log_T; logN -= 4) {
        const int m = logN - 4;
        const u32 tp = t & ((1u << m) - 1);
        ntt_p1_load_tw(w, a.tw, tp, m, a.log_T, a.logn, 4);  // in flight across the barrier
        // (padded index computed per point, not NttLdsWalk: the last step's stride is T, which is 8 or 4 for n >= 2^21, and
        // the walk's affine step holds only for strides that are 1 or a multiple of 16)
#pragma unroll
        for (int r = 0; r < 16; r++) lds[ntt_pad(base_idx + (u32)r * stride)] = x[r];
        __syncthreads();
        base_idx = ((t >> m) << logN) | tp;
        stride = 1u << m;
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = lds[ntt_pad(base_idx + (u32)r * stride)];
        ntt_r16_stage<0>(x, w);
        ntt_r16_stage<1>(x, w);
        ntt_r16_stage<2>(x, w);
        ntt_r16_stage<3>(x, w);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) lds[ntt_pad(base_idx + (u32)r * stride)] = x[r];
    // the output twiddles are fetched across the barrier (x is dead: the registers are free)
    u64 ot[16];
#pragma unroll
    for (int k = 0; k < 16; k++) ot[k] = a.out_tw[at + step(k)];
    loads_issued();
    __syncthreads();
    const NttLdsWalk fin(t, 256);
#pragma unroll
    for (int k = 0; k < 16; k++) out[at + step(k)] = gl::mul_nb(fin.at(lds, k), ot[k]);
}
// out_tw[r * n2 + j2] = tw[rev(r) * j2]   (r < n1, j2 < n2; tw = w^i, i < n)
__global__ void k_pass1_out_tw(const u64* __restrict__ tw, u64* __restrict__ out_tw, int logn, int log_n1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> logn) return;
    const int log_n2 = logn - log_n1;
    const u32 r = (u32)(i >> log_n2), j2 = (u32)(i & (((size_t)1 << log_n2) - 1));
    const u32 k1 = log_n1 ? (__brev(r) >> (32 - log_n1)) : 0;
    out_tw[i] = tw[(size_t)k1 * j2];
}
// out[i] = in[rev(i)] (* post[row][i]) on `blocks` consecutive blocks of n per column
__global__ void k_bitrev_copy(const u64* in, size_t in_col_stride, size_t in_batch_stride, u64* out, size_t out_col_stride, size_t out_batch_stride, int logn,
                              u32 blocks, const u64* post, u32 post_block_perm /*1: table row = rev3(block)*/) {
    const size_t n = (size_t)1 << logn;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 col = blockIdx.z / blocks, blk = blockIdx.z % blocks;
    const u64* src = in + (size_t)blockIdx.y * in_batch_stride + (size_t)col * in_col_stride + (size_t)blk * n;
    u64* dst = out + (size_t)blockIdx.y * out_batch_stride + (size_t)col * out_col_stride + (size_t)blk * n;
    u64 v = src[__brev((u32)i) >> (32 - logn)];
    if (post) v = gl::mul(v, post[(size_t)(post_block_perm ? (__brev(blk) >> 29) : blk) * n + i]);
    dst[i] = v;
}

// table[j][i] = (base[j])^i  for i < n   (coset shift powers and their inverses, zeta powers, ...)
// out[c][i] = a[c][i] * b[i << b_shift] for i < m (the LDE's coset scale folded into the first stage's twiddles)
__global__ void k_mul_tables(u64* out, const u64* a, size_t a_stride, const u64* b, int b_shift, u32 m) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    out[(size_t)blockIdx.y * m + i] = gl::mul(a[(size_t)blockIdx.y * a_stride + i], b[(size_t)i << b_shift]);
}
__global__ void k_pow_table(u64* table, const u64* bases, u32 n, u64 scale) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 b = bases[blockIdx.y];
    table[(size_t)blockIdx.y * n + i] = gl::mul(gl::pow(b, i), scale);
}

// quotient: combine the 8 per-coset residues r_j (each deg < n) into the 8 chunks t_c:
//   t_c[i] = s^-c / 8 * sum_j w8^(-j c) r_j[i],  s = g^n.   r: [ch][8][n] (block rev3(j) = coset j), t: [ch*8 + c][n]
__global__ void k_quotient_chunks_rev(const u64* __restrict__ r, u64* __restrict__ t, u32 n, size_t r_batch_stride, size_t t_batch_stride,
                                  const u64* __restrict__ w8inv_pows /*[8]*/, const u64* __restrict__ scale /*[8]: s^-c/8*/) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 ch = blockIdx.z;
    const u64* rr = r + (size_t)blockIdx.y * r_batch_stride + (size_t)ch * 8 * n + i;
    u64 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = rr[(size_t)(__brev((u32)j) >> 29) * n];
    u64* tt = t + (size_t)blockIdx.y * t_batch_stride + (size_t)ch * 8 * n + i;
    for (int c = 0; c < 8; c++) {
        u64 acc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) acc = gl::add(acc, gl::mul(v[j], w8inv_pows[(j * c) & 7]));
        tt[(size_t)c * n] = gl::mul(acc, scale[c]);
    }
}

// ------------------------------------------------------------------------------------------- witness
struct WitnessArgs {
    const p2::Op* ops;            // scheduled order (witness_schedule.h)
    const p2::WLevel* levels;     // per level: its chains and its single ops
    const p2::WChain* chains;
    u32 num_levels, num_slots, n_inputs;
    const u32* input_slots;   // [n_inputs] (shared by the batch)
    const u64* input_values;  // [batch][n_inputs]
    u64* values;              // [batch][num_slots]
    const u64* lut_ent;       // [num_luts][65536] input value -> (flat table entry index << 16) | output, or ~0
    u32* mult;                // [batch][total_lut_entries] multiplicity counters (zeroed by the caller)
    size_t total_lut_entries;
    int* status;              // [batch]
    const int32_t* wire_slot; // [80][n] (PoseidonGate rows read/write their wires through it)
    u64* advice;              // [batch][num_poseidon_rows][55]: wires 80..134 of every PoseidonGate row
    u32 n, num_poseidon_rows;
};

// PoseidonGenerator: one thread computes the whole row (these ops form sequential sponge chains).  0 ok, 1 conflict, 2 missing input
__device__ __noinline__ int witness_poseidon_op(const WitnessArgs& a, u64* val, u32 proof, const p2::Op& o) {
    u64 w[135];
    int bad = 0;
    const u32 row = o.a;
    for (u32 c = 0; c < 12; c++) {
        w[c] = val[a.wire_slot[(size_t)c * a.n + row]];
        if (w[c] == UNSET) bad = 2;
    }
    w[p2::PG_SWAP] = val[a.wire_slot[(size_t)p2::PG_SWAP * a.n + row]];
    if (w[p2::PG_SWAP] == UNSET) bad = 2;
    if (bad) return bad;
    p2::poseidon_gate_witness(w);
    for (u32 c = p2::PG_OUT; c < 80; c++) {
        if (c == p2::PG_SWAP) continue;
        u32 sl = (u32)a.wire_slot[(size_t)c * a.n + row];
        u64 cur = val[sl];
        if (cur == UNSET)
            val[sl] = w[c];
        else if (cur != w[c])
            bad = 1;
    }
    u64* adv = a.advice + ((size_t)proof * a.num_poseidon_rows + o.aux) * 55;
    for (u32 c = 80; c < 135; c++) adv[c - 80] = w[c];
    return bad;
}

// Most ops per chain the kernel is unrolled for (witness_schedule.h is asked for chains no longer than this).
static const int WITNESS_KMAX = 8;
// Single ops a thread keeps in flight together: the kernel is latency bound (a dependent chain of memory round trips per op,
// one workgroup per proof), so what it processes per unit time is (threads x ops in flight) / round-trip time.
static const int WITNESS_MLP = 4;

// One CHAIN of the witness program: up to WITNESS_KMAX lookup-free ops (ARITH, CONST, EQ) that one thread runs in order
// (witness_schedule.h).  What a level costs is its chain of dependent memory round trips, not its arithmetic, so the chain is
// arranged to need two however long it is:
//   1. all descriptors of the chain (contiguous);
//   2. every operand of every op, and the present value of every output slot, TOGETHER -- whether or not an earlier op of
//      the chain is about to produce it;
//   3. the ops in order, an operand that an earlier op of the chain produced taken from that op's result in registers
//      (forwarding) instead of from memory.
// Returns 0 ok, 2 missing input, 3 conflict (the encoding of s_status).
__device__ __forceinline__ int witness_exec_chain(const WitnessArgs& a, u64* val, u32 ob, u32 cnt) {
    p2::Op o[WITNESS_KMAX];
#pragma unroll
    for (int i = 0; i < WITNESS_KMAX; i++)
        if ((u32)i < cnt) o[i] = a.ops[ob + i];
    u64 x[WITNESS_KMAX], y[WITNESS_KMAX], z[WITNESS_KMAX], cur[WITNESS_KMAX], after[WITNESS_KMAX];
#pragma unroll
    for (int i = 0; i < WITNESS_KMAX; i++) {
        x[i] = y[i] = z[i] = cur[i] = after[i] = 0;
        if ((u32)i < cnt) {
            const u32 kind = o[i].kind;
            cur[i] = val[o[i].out];
            if (kind != p2::OP_CONST) x[i] = val[o[i].a];
            if (kind != p2::OP_CONST) y[i] = val[o[i].b];
            if (kind == p2::OP_ARITH) z[i] = val[o[i].c];
        }
    }
    int worst = 0;
#pragma unroll
    for (int i = 0; i < WITNESS_KMAX; i++) {
        if ((u32)i >= cnt) break;
        const u32 kind = o[i].kind;
        // forwarding: the latest earlier op of this chain that wrote the slot decides (after[j] = the slot's value after op j)
#pragma unroll
        for (int j = 0; j < i; j++) {
            if (o[j].out == o[i].a) x[i] = after[j];
            if (o[j].out == o[i].b) y[i] = after[j];
            if (o[j].out == o[i].c) z[i] = after[j];
            if (o[j].out == o[i].out) cur[i] = after[j];
        }
        u64 r = 0;
        int bad = 0;
        if (kind == p2::OP_ARITH) {
            if (x[i] == UNSET || y[i] == UNSET || z[i] == UNSET)
                bad = 2;
            else
                r = gl::add(gl::mul(gl::mul(x[i], y[i]), o[i].k0), gl::mul(z[i], o[i].k1));
        } else if (kind == p2::OP_CONST) {
            r = o[i].k0;
        } else if (kind == p2::OP_EQ) {
            if (x[i] == UNSET || y[i] == UNSET)
                bad = 2;
            else
                r = x[i] == y[i] ? 1 : 0;
        } else {
            bad = 2;  // no other kind is ever scheduled into a chain
        }
        after[i] = cur[i];
        if (!bad) {
            if (cur[i] == UNSET) {
                val[o[i].out] = r;
                after[i] = r;
            } else if (cur[i] != r) {
                bad = 1;
            }
        }
        if (bad) worst = max(worst, bad == 1 ? 3 : 2);  // conflict (1) outranks missing input (2)
    }
    return worst;
}

// One workgroup generates one witness.  The program is scheduled (witness_schedule.h) into levels; a level holds a few CHAINS
// (the contracted critical path: each run in order by one thread) and many independent SINGLE ops, WITNESS_MLP of which a
// thread keeps in flight together; a workgroup barrier separates the levels.  The descriptors are shared by all proofs.
// CHAINS = false: the instantiation for schedules without chains (P2AES_WITNESS_FUSE=1, or a circuit whose critical path holds
// nothing to fuse); it does not carry the chain executor's registers (252 -> ~100 VGPRs), so its 8 waves share a compute unit
// with the other stream's hash waves instead of needing an empty one.
template <bool HAS_POSEIDON, bool CHAINS>
__global__ __launch_bounds__(512) void k_witness(WitnessArgs a) {
    __shared__ int s_status;
    const u32 proof = blockIdx.x;
    u64* val = a.values + (size_t)proof * a.num_slots;
    u32* mult = a.mult + (size_t)proof * a.total_lut_entries;
    if (threadIdx.x == 0) s_status = 0;
    for (u32 i = threadIdx.x; i < a.num_slots; i += blockDim.x) val[i] = UNSET;
    __syncthreads();
    // PartialWitness::set_target for every input at once: a slot takes the first value that reaches it (compare-and-swap
    // against the unset marker) and any later, different value is a conflict -- the outcome does not depend on the order
    {
        const u64* iv = a.input_values + (size_t)proof * a.n_inputs;
        int bad = 0;
        for (u32 i = threadIdx.x; i < a.n_inputs; i += blockDim.x) {
            const u64 v = iv[i];
            if (v == UNSET) continue;  // this witness does not assign the target (batches share one target list)
            if (v >= gl::P) {
                bad = 1;
                continue;
            }
            const u64 old = atomicCAS((unsigned long long*)&val[a.input_slots[i]], (unsigned long long)UNSET, (unsigned long long)v);
            if (old != UNSET && old != v) bad = 1;
        }
        if (bad) atomicMax(&s_status, 3);
    }
    __syncthreads();
    // The op descriptors do not depend on witness values, so each thread fetches its first single-op descriptor of level
    // lv + 1 before it starts on level lv: on deep circuits (10^3..10^4 levels, descriptors streaming from HBM) the descriptor
    // latency is otherwise the longest link of the per-level dependency chain.
    p2::Op nxt;
    bool have_nxt = false;
    p2::WLevel NL = a.levels[0];
    if (NL.single_begin + threadIdx.x < NL.single_end) {
        nxt = a.ops[NL.single_begin + threadIdx.x];
        have_nxt = true;
    }
    for (u32 lv = 0; lv < a.num_levels; lv++) {
        const p2::WLevel L = NL;
        const p2::Op first = nxt;
        const bool have_first = have_nxt;
        have_nxt = false;
        if (lv + 1 < a.num_levels) {
            NL = a.levels[lv + 1];
            if (NL.single_begin + threadIdx.x < NL.single_end) {
                nxt = a.ops[NL.single_begin + threadIdx.x];
                have_nxt = true;
            }
        }
        int worst = 0;
        // the contracted critical path first: it is what the next level waits for
        if (CHAINS) {
            for (u32 ci = threadIdx.x; ci < L.chain_count; ci += blockDim.x) {
                const p2::WChain ch = a.chains[L.chain_begin + ci];
                worst = max(worst, witness_exec_chain(a, val, ch.start, min(ch.count, (u32)WITNESS_KMAX)));
            }
        }
        // stages of a single op: descriptor, then every operand together with the present value of the output slot (it does
        // not depend on the operands), then the table entry of a lookup; WITNESS_MLP ops go through them side by side
        for (u32 k0 = L.single_begin + threadIdx.x; k0 < L.single_end; k0 += WITNESS_MLP * blockDim.x) {
            p2::Op o[WITNESS_MLP];
            bool act[WITNESS_MLP];
#pragma unroll
            for (int u = 0; u < WITNESS_MLP; u++) {
                const u32 k = k0 + u * blockDim.x;
                act[u] = k < L.single_end;
                if (act[u]) o[u] = (u == 0 && have_first && k0 == L.single_begin + threadIdx.x) ? first : a.ops[k];
            }
            u64 x[WITNESS_MLP], y[WITNESS_MLP], z[WITNESS_MLP], cur[WITNESS_MLP], ent[WITNESS_MLP];
#pragma unroll
            for (int u = 0; u < WITNESS_MLP; u++) {
                x[u] = y[u] = z[u] = cur[u] = 0;
                if (!act[u] || o[u].kind == p2::OP_POSEIDON) continue;
                const u32 kind = o[u].kind;
                cur[u] = val[o[u].out];
                if (kind != p2::OP_CONST) x[u] = val[o[u].a];
                if (kind == p2::OP_ARITH || kind == p2::OP_EQ || kind == p2::OP_EQINV) y[u] = val[o[u].b];
                if (kind == p2::OP_ARITH) z[u] = val[o[u].c];
            }
#pragma unroll
            for (int u = 0; u < WITNESS_MLP; u++) {
                ent[u] = ~0ull;  // (flat entry index << 16) | output, or ~0
                if (act[u] && o[u].kind == p2::OP_LOOKUP && x[u] < 65536) ent[u] = a.lut_ent[(size_t)o[u].aux * 65536 + x[u]];
            }
#pragma unroll
            for (int u = 0; u < WITNESS_MLP; u++) {
                if (!act[u]) continue;
                const u32 kind = o[u].kind;
                u64 r = 0;
                int bad = 0;
                if (kind == p2::OP_POSEIDON) {
                    if (HAS_POSEIDON) bad = witness_poseidon_op(a, val, proof, o[u]);
                    if (bad) worst = max(worst, bad == 1 ? 3 : 2);
                    continue;
                }
                if (kind == p2::OP_ARITH) {
                    if (x[u] == UNSET || y[u] == UNSET || z[u] == UNSET)
                        bad = 2;
                    else
                        r = gl::add(gl::mul(gl::mul(x[u], y[u]), o[u].k0), gl::mul(z[u], o[u].k1));
                } else if (kind == p2::OP_CONST) {
                    r = o[u].k0;
                } else if (kind == p2::OP_LOOKUP) {
                    if (x[u] == UNSET) {
                        bad = 2;
                    } else if (ent[u] == ~0ull) {  // not a 16-bit value, or not in the table
                        bad = 1;
                    } else {
                        r = ent[u] & 0xFFFF;
                        atomicAdd(&mult[ent[u] >> 16], 1u);
                    }
                } else {
                    if (x[u] == UNSET || y[u] == UNSET)
                        bad = 2;
                    else if (kind == p2::OP_EQ)
                        r = x[u] == y[u] ? 1 : 0;
                    else
                        r = x[u] == y[u] ? 0 : gl::inv(gl::sub(x[u], y[u]));
                }
                if (!bad) {
                    if (cur[u] == UNSET)
                        val[o[u].out] = r;
                    else if (cur[u] != r)
                        bad = 1;
                }
                if (bad) worst = max(worst, bad == 1 ? 3 : 2);  // conflict (1) outranks missing input (2); remapped below
            }
        }
        if (worst) atomicMax(&s_status, worst);
        __syncthreads();
    }
    if (threadIdx.x == 0) a.status[proof] = s_status == 3 ? 1 : s_status;
}

// wires[col][row] = value of the wire's partition (or 0 for unconnected wires)
__global__ void k_fill_wires(const int32_t* __restrict__ wire_slot, const u64* __restrict__ values, u64* __restrict__ wires, size_t total /*cols*n*/,
                             u32 num_slots, size_t wires_batch_stride, int* status) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int32_t s = wire_slot[idx];
    u64 v = 0;
    if (s >= 0) {
        v = values[(size_t)blockIdx.y * num_slots + s];
        if (v == UNSET) {
            v = 0;
            atomicCAS(&status[blockIdx.y], 0, 2);  // a conflict / lookup miss (1) already recorded wins
        }
    }
    wires[(size_t)blockIdx.y * wires_batch_stride + idx] = v;
}

// wires 80..134: the PoseidonGate rows take their advice block, every other row is 0
__global__ void k_fill_advice(const int32_t* __restrict__ pos_index /*[n]: advice block of the row or -1*/, const u64* __restrict__ advice, u64* __restrict__ wires,
                              u32 n, u32 num_poseidon_rows, size_t wires_batch_stride) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)55 * n) return;
    u32 c = (u32)(idx / n), row = (u32)(idx % n);
    int32_t k = pos_index[row];
    u64 v = k >= 0 ? advice[((size_t)blockIdx.y * num_poseidon_rows + k) * 55 + c] : 0;
    wires[(size_t)blockIdx.y * wires_batch_stride + (size_t)(80 + c) * n + row] = v;
}

// zk blinding rows: every wire of a `rows` entry is random; the 80 routed wires of both rows of a `zrows` pair carry the
// same random value (they are copy-constrained).  One thread = one PRF block of eight elements.
__global__ void k_fill_blind(const u32* __restrict__ rows, u32 n_rows, const u32* __restrict__ zrows, u32 n_z, u64* __restrict__ wires,
                             size_t wires_batch_stride, u32 n, p2::ZkKey key, u64 proof_base) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64* w = wires + (size_t)blockIdx.y * wires_batch_stride;
    const u64 proof = proof_base + blockIdx.y;
    const size_t regular = (size_t)n_rows * 135, nb_reg = (regular + 7) / 8, zcount = (size_t)n_z * 80, nb_z = (zcount + 7) / 8;
    u64 v[8];
    if (t < nb_reg) {
        p2::zk_block(key, proof, p2::ZK_ROW, t, v);
        for (int j = 0; j < 8; j++) {
            size_t idx = 8 * t + j;
            if (idx < regular) w[(size_t)(idx % 135) * n + rows[idx / 135]] = v[j];
        }
    } else if (t < nb_reg + nb_z) {
        t -= nb_reg;
        p2::zk_block(key, proof, p2::ZK_ZROW, t, v);
        for (int j = 0; j < 8; j++) {
            size_t idx = 8 * t + j;
            if (idx < zcount) {
                u32 k = (u32)(idx / 80), c = (u32)(idx % 80);
                w[(size_t)c * n + zrows[2 * k]] = v[j];
                w[(size_t)c * n + zrows[2 * k + 1]] = v[j];
            }
        }
    }
}
// SALT_SIZE random columns appended to a blinded oracle's LDE matrix (they are leaf data only, not polynomials).
// Element idx = s * N + pos; one thread = one PRF block = eight consecutive positions of one salt column.
__global__ void k_fill_salt(u64* __restrict__ salt_cols, size_t batch_stride, size_t N, p2::ZkKey key, u64 proof_base, u64 domain) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = p2::SALT_SIZE * N;
    if (8 * t >= total) return;
    u64 v[8];
    p2::zk_block(key, proof_base + blockIdx.y, domain, t, v);
    u64* o = salt_cols + (size_t)blockIdx.y * batch_stride + 8 * t;
    for (int j = 0; j < 8; j++)
        if (8 * t + j < total) o[j] = v[j];
}

struct LutRowsArgs {
    const u32* lut_pairs;
    const u32* lut_offsets;
    const p2::LookupRows* rows;
    const u32* num_lookups;
    const u32* mult;
    size_t total_lut_entries;
    u64* wires;
    size_t wires_batch_stride;
    u32 n, num_luts;
};
// LookupTableGate rows (table stored upside down + multiplicities) and padding of each LUT's last LookupGate
__global__ void k_lut_rows(LutRowsArgs a) {
    u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    u64* w = a.wires + (size_t)blockIdx.y * a.wires_batch_stride;
    const u32* mult = a.mult + (size_t)blockIdx.y * a.total_lut_entries;
    if (e < a.total_lut_entries) {
        u32 l = 0;
        while (l + 1 < a.num_luts && e >= a.lut_offsets[l + 1]) l++;
        u32 slot = e - a.lut_offsets[l];
        u32 row = a.rows[l].first_lut - slot / p2::LUT_SLOTS, s = slot % p2::LUT_SLOTS;
        u32 pr = a.lut_pairs[e];
        u64 m = mult[e];
        if (slot == 0) m += (p2::LU_SLOTS - a.num_lookups[l] % p2::LU_SLOTS) % p2::LU_SLOTS;
        w[(size_t)(3 * s) * a.n + row] = pr & 0xFFFF;
        w[(size_t)(3 * s + 1) * a.n + row] = pr >> 16;
        w[(size_t)(3 * s + 2) * a.n + row] = m;
    }
    // padding: one thread per (lut, slot) of the first 40*num_luts threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < p2::LU_SLOTS * a.num_luts) {
        u32 l = threadIdx.x / p2::LU_SLOTS, slot = threadIdx.x % p2::LU_SLOTS;
        u32 remaining = (p2::LU_SLOTS - a.num_lookups[l] % p2::LU_SLOTS) % p2::LU_SLOTS;
        if (slot >= p2::LU_SLOTS - remaining) {
            u32 pr = a.lut_pairs[a.lut_offsets[l]];
            u32 row = a.rows[l].last_lut - 1;
            w[(size_t)(2 * slot) * a.n + row] = pr & 0xFFFF;
            w[(size_t)(2 * slot + 1) * a.n + row] = pr >> 16;
        }
    }
}

// ------------------------------------------------------------------------------------------- challenger
// Per-proof Fiat-Shamir state kept in global memory between stages.
struct ChalState {
    u64 state[12];
    u64 in[8];
    u64 out[8];
    u32 in_len, out_len;
};
// The sponge of one proof is spread over a 16-lane group (glf::poseidon_coop): lane i holds state word i, input-buffer
// word i and output-buffer word i.  The ~110 permutations of a proof's Fiat-Shamir chain are strictly sequential; with one
// thread per proof (round 1) a single lane's 19 k-instruction stream per permutation set the latency of every stage.
struct DevChallenger {
    u64 w, inb, outb;     // this lane's word of state / in / out
    u32 in_len, out_len;  // the same in every lane of the group
    u32 i;                // lane within the group
    int gbase;            // first lane of the group within the wave
    __device__ void load(const ChalState& s) {
        w = i < 12 ? s.state[i] : 0;
        inb = i < 8 ? s.in[i] : 0;
        outb = i < 8 ? s.out[i] : 0;
        in_len = s.in_len;
        out_len = s.out_len;
    }
    __device__ void store(ChalState& s) const {
        if (i < 12) s.state[i] = w;
        if (i < 8) {
            s.in[i] = inb;
            s.out[i] = outb;
        }
        if (i == 0) {
            s.in_len = in_len;
            s.out_len = out_len;
        }
    }
    __device__ void duplexing() {
        if (i < in_len) w = inb;
        in_len = 0;
        w = glf::poseidon_coop(w, i);
        outb = w;
        out_len = 8;
    }
    __device__ void observe(u64 x) {  // x: the same value in every lane of the group
        out_len = 0;
        if (i == in_len) inb = x;
        in_len++;
        if (in_len == 8) duplexing();
    }
    __device__ u64 challenge() {  // returns the value to every lane of the group
        if (in_len != 0 || out_len == 0) duplexing();
        --out_len;
        return glf::shfl64(outb, gbase + (int)out_len);
    }
};

// Challenge block per proof (u64 words)
enum ChalSlot {
    CH_BETAS = 0,      // 2
    CH_GAMMAS = 2,     // 2
    CH_DELTAS = 4,     // 8
    CH_ALPHAS = 12,    // 2
    CH_ZETA = 14,      // 2
    CH_FRI_ALPHA = 16, // 2
    CH_FRI_BETAS = 18, // 2 * up to 8 rounds
    CH_POW = 34,       // 1
    CH_QUERY = 36,     // 28 (up to 64)
    CH_WORDS = 100
};

struct ChalArgs {
    ChalState* st;        // [batch]
    u64* chal;            // [batch][CH_WORDS]
    const u64* observe;   // data to observe: per proof `observe_len` words at stride observe_stride
    size_t observe_stride;
    u32 observe_len;
    u32 batch;
    u32 stage;            // see k_challenger
    u32 aux;              // stage-specific (fri round, #queries, has_lookup)
    u64 mod;              // query index modulus
    const u64* digest;    // circuit digest (stage 0)
    int* status;
};
// stage 0: init; observe circuit digest, pi hash (0^4), wires cap; betas, gammas, deltas
// stage 1: observe zs cap; alphas            stage 2: observe quotient cap; zeta
// stage 3: observe openings; fri_alpha       stage 4: observe FRI cap (round aux); beta
// stage 5: observe final poly                stage 6: observe pow witness; response; query indices
__global__ __launch_bounds__(64) void k_challenger(ChalArgs a) {
    // 16 lanes per proof, 4 proofs per wave; a group past the end of the batch replays the last proof without storing
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool real = (t >> 4) < a.batch;
    const u32 p = real ? (t >> 4) : a.batch - 1;
    DevChallenger c;
    c.i = t & 15;
    c.gbase = (int)(threadIdx.x & 63) & ~15;
    const bool writer = real && c.i == 0;
    u64* ch = a.chal + (size_t)p * CH_WORDS;
    if (a.stage == 0) {
        c.w = c.inb = c.outb = 0;
        c.in_len = c.out_len = 0;
        for (int i = 0; i < 4; i++) c.observe(a.digest[i]);
        for (int i = 0; i < 4; i++) c.observe(0);
    } else {
        c.load(a.st[p]);
    }
    const u64* ob = a.observe + (size_t)p * a.observe_stride;
    if (a.stage == 6) {
        c.observe(ch[CH_POW]);
        (void)c.challenge();
        for (u32 q = 0; q < a.aux; q++) {
            const u64 v = c.challenge() % a.mod;
            if (writer) ch[CH_QUERY + q] = v;
        }
        if (real) c.store(a.st[p]);
        return;
    }
    for (u32 i = 0; i < a.observe_len; i++) c.observe(ob[i]);
    if (a.stage == 0) {
        u64 bg[4], dl[4];
        for (int i = 0; i < 4; i++) bg[i] = c.challenge();  // betas, gammas
        if (a.aux)
            for (int i = 0; i < 4; i++) dl[i] = c.challenge();
        if (writer) {
            for (int i = 0; i < 2; i++) ch[CH_BETAS + i] = bg[i];
            for (int i = 0; i < 2; i++) ch[CH_GAMMAS + i] = bg[2 + i];
            if (a.aux) {
                for (int i = 0; i < 4; i++) ch[CH_DELTAS + i] = bg[i];
                for (int i = 0; i < 4; i++) ch[CH_DELTAS + 4 + i] = dl[i];
            }
        }
    } else if (a.stage == 1) {
        const u64 a0 = c.challenge(), a1 = c.challenge();
        if (writer) ch[CH_ALPHAS] = a0, ch[CH_ALPHAS + 1] = a1;
    } else if (a.stage == 2) {
        const u64 z0 = c.challenge(), z1 = c.challenge();
        if (writer) {
            ch[CH_ZETA] = z0;
            ch[CH_ZETA + 1] = z1;
            // "Opening point is in the subgroup."
            E2 zp = gl::exp_pow2(gl::e2(z0, z1), (int)a.aux);
            if (zp.a == 1 && zp.b == 0) atomicMax(&a.status[p], 3);
        }
    } else if (a.stage == 3) {
        const u64 f0 = c.challenge(), f1 = c.challenge();
        if (writer) ch[CH_FRI_ALPHA] = f0, ch[CH_FRI_ALPHA + 1] = f1;
    } else if (a.stage == 4) {
        const u64 b0 = c.challenge(), b1 = c.challenge();
        if (writer) ch[CH_FRI_BETAS + 2 * a.aux] = b0, ch[CH_FRI_BETAS + 2 * a.aux + 1] = b1;
    }
    if (real) c.store(a.st[p]);
}

// Proof-of-work grinding: smallest witness w such that the duplex response has >= pow_bits leading zeros.
// One 256-candidate block per workgroup; a workgroup whose whole block lies above the best witness found so far exits at
// once.  2^21 candidates per proof are covered in three phases of 2^16, 3 * 2^16 and 7 * 2^18 candidates: ~2^16 candidates
// are needed per proof on average, so the first phase (grid = proofs x 256 blocks, proofs varying fastest) settles 63 % of a
// chunk, and the later phases run only over the COMPACTED list of unsolved proofs (k_pow_compact) with a small grid of list
// slots.  Round 1 launched proofs x 8192 workgroups in one go: a million of them per chunk did nothing but one memory-side
// atomic on one of 128 addresses and exit, which cost as much as the hashing itself.
// The probability that 2^21 candidates hold no witness is exp(-32) (reported as status 4, never a bad proof).
static const u32 POW_PHASE_BLOCKS[3] = {1u << 8, 3u << 8, 7u << 10};  // x 256 candidates: 2^16 + 3 * 2^16 + 7 * 2^18 = 2^21
static const u32 POW_PHASE_SLOTS[3] = {0, 64, 8};                      // list slots of the grid (phase 0 addresses proofs directly)
__global__ __launch_bounds__(256) void k_pow(const ChalState* st, u64* chal, int pow_bits, unsigned long long* best /*[batch]*/, u32 block0,
                                             const u32* list /* null: blockIdx.x is the proof */, const u32* count) {
    // The candidate is the next observed element: whether or not it completes the rate, the response is word 7 of
    // permute(state overwritten by the buffered inputs and the candidate).  The overwritten state is the same for every
    // candidate of a proof, so it is staged once per workgroup in LDS and the permutation runs entirely in registers.
    __shared__ u64 sh[12];
    __shared__ u32 sh_pos;
    __shared__ unsigned long long sh_cur;
    const u64 block_start = ((u64)block0 + blockIdx.y) * blockDim.x;
    const u32 n_items = list ? *count : gridDim.x;
    for (u32 item = blockIdx.x; item < n_items; item += gridDim.x) {  // workgroup-uniform; one trip unless the list outgrows the grid
        const u32 p = list ? list[item] : item;
        // `best` is updated by workgroups on all 8 XCDs; their L2s are not coherent with each other, so a plain (even sc1)
        // load can keep returning this XCD's stale copy.  A no-op atomic min executes at the memory side and returns the
        // current value; one lane per workgroup issues it.
        __syncthreads();
        if (threadIdx.x == 0) sh_cur = atomicMin(&best[p], ~0ull);
        __syncthreads();
        if (sh_cur < block_start) continue;  // workgroup-uniform
        if (threadIdx.x < 12) {
            const ChalState* s = st + p;
            u32 i = threadIdx.x;
            sh[i] = (i < s->in_len) ? s->in[i] : s->state[i];
            if (i == 0) sh_pos = s->in_len;
        }
        __syncthreads();
        const u64 cand = block_start + threadIdx.x;
        const u32 pos = sh_pos;
        u64 r[12];
#pragma unroll
        for (int i = 0; i < 12; i++) r[i] = ((u32)i == pos) ? cand : sh[i];
        glf::poseidon(r);
        if ((r[7] >> (64 - pow_bits)) == 0) atomicMin(&best[p], (unsigned long long)cand);
    }
}
// list of the proofs that still have no witness (any order), for the next phase
__global__ void k_pow_compact(const unsigned long long* best, u32 batch, u32* list, u32* count) {
    __shared__ u32 n;
    if (threadIdx.x == 0) n = 0;
    __syncthreads();
    for (u32 p = threadIdx.x; p < batch; p += blockDim.x)
        if (best[p] == ~0ull) list[atomicAdd(&n, 1u)] = p;
    __syncthreads();
    if (threadIdx.x == 0) *count = n;
}
__global__ void k_pow_finish(u64* chal, const unsigned long long* best, u32 batch, int* status) {
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= batch) return;
    chal[(size_t)p * CH_WORDS + CH_POW] = best[p];
    if (best[p] == ~0ull) atomicMax(&status[p], 4);
}

// ------------------------------------------------------------------------------------------- block scans
// Inclusive scan of 1024 per-thread partials held in LDS (Hillis-Steele, 10 steps); OP is gl::add or gl::mul.
template <bool MUL>
__device__ __forceinline__ u64 block_scan_inclusive(u64 v, u64* lds) {
    const u32 t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (u32 off = 1; off < blockDim.x; off <<= 1) {
        u64 x = lds[t];
        u64 y = t >= off ? lds[t - off] : (MUL ? 1 : 0);
        __syncthreads();
        lds[t] = MUL ? gl::mul(x, y) : gl::add(x, y);
        __syncthreads();
    }
    return lds[t];
}

// ------------------------------------------------------------------------------------------- permutation argument
// q[ch][chunk][row] = prod_{j in chunk}(w_j + beta*k_j*x + gamma) / prod_{j in chunk}(w_j + beta*sigma_j + gamma)
// Thread = row.  Both challenges share every wire and sigma load, and the 2 x 10 denominators of a row are inverted
// together (Montgomery's trick: one Fermat inversion and 3 multiplications per value instead of 20 inversions -- the
// inversions were 3/4 of this kernel's multiplies).
static const u32 PERM_MAX_CHUNKS = 12;
__global__ __launch_bounds__(256) void k_perm_chunks(const u64* __restrict__ wires, size_t wires_batch_stride, const u64* __restrict__ sigmas,
                                                     const u64* __restrict__ k_is, const u64* __restrict__ subgroup, const u64* __restrict__ chal,
                                                     u64* __restrict__ q, size_t q_batch_stride, u32 n, u32 num_routed, u32 chunk_size, u32 num_chunks,
                                                     u32 num_challenges) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const u64* cw = chal + (size_t)blockIdx.y * CH_WORDS;
    const u64* w = wires + (size_t)blockIdx.y * wires_batch_stride + row;
    const u64* sg = sigmas + row;
    const bool two = num_challenges > 1;
    const u64 b0 = cw[CH_BETAS], g0 = cw[CH_GAMMAS], b1 = two ? cw[CH_BETAS + 1] : 0, g1 = two ? cw[CH_GAMMAS + 1] : 0;
    const u64 x = subgroup[row], bx0 = gl::mul(b0, x), bx1 = gl::mul(b1, x);
    u64 num[2 * PERM_MAX_CHUNKS], den[2 * PERM_MAX_CHUNKS], pre[2 * PERM_MAX_CHUNKS];
#pragma unroll
    for (u32 c = 0; c < PERM_MAX_CHUNKS; c++) {
        u64 n0 = 1, d0 = 1, n1 = 1, d1 = 1;
        if (c < num_chunks) {
            const u32 j1 = min(num_routed, (c + 1) * chunk_size);
            for (u32 j = c * chunk_size; j < j1; j++) {
                const u64 wv = w[(size_t)j * n], s = sg[(size_t)j * n], kj = k_is[j];
                n0 = gl::mul(n0, gl::add(gl::add(wv, gl::mul(bx0, kj)), g0));
                d0 = gl::mul(d0, gl::add(gl::add(wv, gl::mul(b0, s)), g0));
                if (two) {
                    n1 = gl::mul(n1, gl::add(gl::add(wv, gl::mul(bx1, kj)), g1));
                    d1 = gl::mul(d1, gl::add(gl::add(wv, gl::mul(b1, s)), g1));
                }
            }
        }
        num[c] = n0;
        den[c] = d0;
        num[PERM_MAX_CHUNKS + c] = n1;
        den[PERM_MAX_CHUNKS + c] = d1;
    }
    u64 run = 1;
#pragma unroll
    for (u32 k = 0; k < 2 * PERM_MAX_CHUNKS; k++) {
        pre[k] = run;
        run = gl::mul(run, den[k]);
    }
    // A zero denominator (probability ~2^-45 per proof; upstream's quotient computation fails on it) zeroes the whole
    // row here, where value-by-value inversion would zero one chunk: no valid proof exists either way.
    u64 inv = gl::inv(run);
#pragma unroll
    for (u32 kk = 2 * PERM_MAX_CHUNKS; kk-- > 0;) {
        const u64 dinv = gl::mul(inv, pre[kk]);
        inv = gl::mul(inv, den[kk]);
        const u32 ch = kk / PERM_MAX_CHUNKS, c = kk % PERM_MAX_CHUNKS;
        if (c < num_chunks && ch < num_challenges)
            q[(size_t)blockIdx.y * q_batch_stride + ((size_t)ch * num_chunks + c) * n + row] = gl::mul(num[kk], dinv);
    }
}

// Z and partial products from the chunk quotients.  One workgroup per (proof, challenge); rows are split into
// 1024 contiguous segments, a workgroup-wide multiplicative scan links the segments.
//   zs layout: [Z_0, Z_1, pp(ch0) 0..npp, pp(ch1) 0..npp, ...]   each [n]
// Z and the partial products: a running product over (row, chunk) in row-major order.  The rows are swept 1024 at a
// time -- thread t takes row base + t, so every column read and write is coalesced -- with a workgroup scan per sweep
// and the running total carried from sweep to sweep.  (A thread-owns-16-consecutive-rows split reads each 128-byte
// line for 8 bytes: the PMC counters showed 32x the algorithmic traffic.)
// Long columns (n > 2^14: few proofs fit a chunk, and one workgroup per (proof, challenge) walking 2^19 rows left the chip idle
// for 5 ms per 16 proofs) are cut into `segs` = gridDim.z segments of equal length, a workgroup each: k_perm_seg_products first
// multiplies every segment's quotients together, and a segment's sweep starts from the product of the segments before it.
static const u32 PERM_MAX_SEGS = 64;
__global__ __launch_bounds__(1024) void k_perm_seg_products(const u64* __restrict__ q, size_t q_batch_stride, u64* __restrict__ seg_tot, u32 n, u32 num_chunks) {
    __shared__ u64 lds[1024];
    const u32 ch = blockIdx.x, t = threadIdx.x, segs = gridDim.z, seg = blockIdx.z, seg_rows = n / segs;
    const u64* qq = q + (size_t)blockIdx.y * q_batch_stride + (size_t)ch * num_chunks * n;
    u64 prod = 1;
    for (u32 r = seg * seg_rows + t; r < (seg + 1) * seg_rows; r += blockDim.x)
        for (u32 c = 0; c < num_chunks; c++) prod = gl::mul(prod, qq[(size_t)c * n + r]);
    lds[t] = prod;
    __syncthreads();
    for (u32 off = blockDim.x / 2; off > 0; off >>= 1) {
        if (t < off) lds[t] = gl::mul(lds[t], lds[t + off]);
        __syncthreads();
    }
    if (t == 0) seg_tot[((size_t)blockIdx.y * gridDim.x + ch) * segs + seg] = lds[0];
}
__global__ __launch_bounds__(1024) void k_perm_scan(const u64* __restrict__ q, size_t q_batch_stride, u64* __restrict__ zs, size_t zs_batch_stride, u32 n,
                                                     u32 num_chunks, u32 num_challenges, const u64* __restrict__ seg_tot /* null: one segment */) {
    __shared__ u64 lds[1024];
    __shared__ u64 s_carry;
    const u32 ch = blockIdx.x, t = threadIdx.x, segs = gridDim.z, seg = blockIdx.z, seg_rows = n / segs;
    const u64* qq = q + (size_t)blockIdx.y * q_batch_stride + (size_t)ch * num_chunks * n;
    u64* z = zs + (size_t)blockIdx.y * zs_batch_stride + (size_t)ch * n;
    u64* pp = zs + (size_t)blockIdx.y * zs_batch_stride + ((size_t)num_challenges + (size_t)ch * (num_chunks - 1)) * n;
    if (t == 0) {
        u64 c0 = 1;
        for (u32 s = 0; s < seg; s++) c0 = gl::mul(c0, seg_tot[((size_t)blockIdx.y * gridDim.x + ch) * segs + s]);
        s_carry = c0;
    }
    __syncthreads();
    for (u32 base = seg * seg_rows; base < (seg + 1) * seg_rows; base += blockDim.x) {
        const u32 r = base + t;
        const bool live = r < (seg + 1) * seg_rows;
        u64 v[PERM_MAX_CHUNKS];
        u64 prod = 1;
#pragma unroll
        for (u32 c = 0; c < PERM_MAX_CHUNKS; c++) {
            v[c] = (c < num_chunks && live) ? qq[(size_t)c * n + r] : 1;
            prod = gl::mul(prod, v[c]);
        }
        const u64 carry = s_carry;
        (void)block_scan_inclusive<true>(prod, lds);  // ends on a barrier: lds[] holds the inclusive products
        u64 acc = gl::mul(carry, t == 0 ? 1 : lds[t - 1]);
        if (live) {
            z[r] = acc;
#pragma unroll
            for (u32 c = 0; c < PERM_MAX_CHUNKS; c++) {
                if (c < num_chunks) {
                    acc = gl::mul(acc, v[c]);
                    if (c + 1 < num_chunks) pp[(size_t)c * n + r] = acc;
                }
            }
        }
        __syncthreads();  // everyone has read s_carry and lds[t - 1]
        if (t == blockDim.x - 1) s_carry = gl::mul(carry, lds[t]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------- lookup argument
struct LookupArgs {
    const u64* wires;
    size_t wires_batch_stride;
    const u64* chal;
    u64* zs;  // base of the zs batch
    size_t zs_batch_stride;
    u64* tmp;  // [batch][ch][num_sldc + 1][n] scratch: per-row partial sums S_k and RE terms
    size_t tmp_batch_stride;
    const p2::LookupRows* rows;
    u32 n, num_luts, num_sldc, lut_deg, lu_deg, num_challenges, zs_lookup_col0;  // first lookup column in zs
};
// phase 1: per (row, partial poly k): S_k[row] = sum over the poly's slots of mult/(alpha - combo)  (LUT rows)
//          or  -sum 1/(alpha - combo)  (LookupGate rows); k == num_sldc: RE row term c[row] = sum_s combo_B[s] delta^(25-s)
__global__ void k_lookup_terms(LookupArgs a) {
    u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.n) return;
    u32 k = blockIdx.z % (a.num_sldc + 1), ch = blockIdx.z / (a.num_sldc + 1);
    // which LUT (if any) owns this row?
    int kind = 0;  // 1 = LUT row, 2 = LU row
    for (u32 l = 0; l < a.num_luts; l++) {
        p2::LookupRows lr = a.rows[l];
        if (row >= lr.last_lut && row <= lr.first_lut) kind = 1;
        if (row >= lr.last_lu && row < lr.last_lut) kind = 2;
    }
    const u64* d = a.chal + (size_t)blockIdx.y * CH_WORDS + CH_DELTAS + 4 * ch;  // A, B, Alpha, Delta
    const u64* w = a.wires + (size_t)blockIdx.y * a.wires_batch_stride + row;
    u64 r = 0;
    if ((kind == 1 || kind == 2) && k < a.num_sldc) {
        // sum over this partial polynomial's slots of c_s / (alpha - combo_s)  (c_s = multiplicity, or -1 for LookupGate
        // rows) with ONE field inversion: prefix products forwards, peel the inverse backwards (Montgomery's trick)
        const bool lut = kind == 1;
        const u32 deg = lut ? a.lut_deg : a.lu_deg, total = lut ? p2::LUT_SLOTS : p2::LU_SLOTS, stride = lut ? 3 : 2;
        const u32 s0 = k * deg, s1 = min(total, s0 + deg), cnt = s1 > s0 ? s1 - s0 : 0;
        u64 f[8], pre[8];
        u64 acc = 1;
#pragma unroll
        for (u32 i = 0; i < 8; i++) {  // fixed trip count + predicate keeps f/pre in registers (cnt <= 7)
            f[i] = 1;
            pre[i] = acc;
            if (i < cnt) {
                u32 s = s0 + i;
                u64 combo = gl::add(w[(size_t)(stride * s) * a.n], gl::mul(d[0], w[(size_t)(stride * s + 1) * a.n]));
                f[i] = gl::sub(d[2], combo);
                acc = gl::mul(acc, f[i]);
            }
        }
        u64 inv_all = gl::inv(acc);
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            if ((u32)i < cnt) {
                u64 fi_inv = gl::mul(inv_all, pre[i]);
                inv_all = gl::mul(inv_all, f[i]);
                if (lut)
                    r = gl::add(r, gl::mul(w[(size_t)(3 * (s0 + i) + 2) * a.n], fi_inv));
                else
                    r = gl::sub(r, fi_inv);
            }
        }
    } else if (kind == 1) {
        for (u32 s = 0; s < p2::LUT_SLOTS; s++) {
            u64 combo = gl::add(w[(size_t)(3 * s) * a.n], gl::mul(d[1], w[(size_t)(3 * s + 1) * a.n]));
            r = gl::add(gl::mul(r, d[3]), combo);
        }
    }
    a.tmp[(size_t)blockIdx.y * a.tmp_batch_stride + ((size_t)ch * (a.num_sldc + 1) + k) * a.n + row] = r;
}
// phase 2: one workgroup per (lut, challenge, proof): suffix scans over the LUT's rows [last_lu, first_lut].
//   SLDC_k[row] = (sum_{r > row} T[r]) + sum_{k' <= k} S_k'[row],  T[r] = sum_k S_k[r]
//   RE[row]     = sum_{r >= row, r in LUT rows} c[r] * D^(r-row),  D = delta^26
__global__ __launch_bounds__(1024) void k_lookup_scan(LookupArgs a) {
    __shared__ u64 lds[1024];
    const u32 l = blockIdx.x, ch = blockIdx.z, t = threadIdx.x;
    const p2::LookupRows lr = a.rows[l];
    const u64* tmp = a.tmp + (size_t)blockIdx.y * a.tmp_batch_stride + (size_t)ch * (a.num_sldc + 1) * a.n;
    u64* zl = a.zs + (size_t)blockIdx.y * a.zs_batch_stride + ((size_t)a.zs_lookup_col0 + (size_t)ch * (a.num_sldc + 1)) * a.n;  // [RE, SLDC_0..]
    const u64* d = a.chal + (size_t)blockIdx.y * CH_WORDS + CH_DELTAS + 4 * ch;
    // rows processed top-down: position i = first_lut - row, i in [0, total)
    const u32 total = lr.first_lut - lr.last_lu + 1;
    const u32 per = (total + blockDim.x - 1) / blockDim.x;
    const u32 i0 = min(total, t * per), i1 = min(total, i0 + per);
    // ---- SLDC: additive scan of row totals
    u64 seg = 0;
    for (u32 i = i0; i < i1; i++) {
        u32 row = lr.first_lut - i;
        for (u32 k = 0; k < a.num_sldc; k++) seg = gl::add(seg, tmp[(size_t)k * a.n + row]);
    }
    block_scan_inclusive<false>(seg, lds);
    u64 acc = t == 0 ? 0 : lds[t - 1];
    for (u32 i = i0; i < i1; i++) {
        u32 row = lr.first_lut - i;
        for (u32 k = 0; k < a.num_sldc; k++) {
            acc = gl::add(acc, tmp[(size_t)k * a.n + row]);
            zl[(size_t)(1 + k) * a.n + row] = acc;
        }
    }
    __syncthreads();
    // ---- RE over the LUT rows only: re[row] = re[row+1]*D + c[row]; with i = first_lut - row:
    //      re_i = sum_{m <= i} c_m D^(i-m)  =>  re_i = D^i * sum_{m<=i} c_m D^-m
    const u32 lut_rows = lr.first_lut - lr.last_lut + 1;
    const u32 per2 = (lut_rows + blockDim.x - 1) / blockDim.x;
    const u32 j0 = min(lut_rows, t * per2), j1 = min(lut_rows, j0 + per2);
    u64 D = gl::pow(d[3], p2::LUT_SLOTS), Dinv = gl::inv(D);
    u64 dm = gl::pow(Dinv, j0), s2 = 0;
    for (u32 i = j0; i < j1; i++) {
        s2 = gl::add(s2, gl::mul(tmp[(size_t)a.num_sldc * a.n + (lr.first_lut - i)], dm));
        dm = gl::mul(dm, Dinv);
    }
    block_scan_inclusive<false>(s2, lds);
    u64 pre = t == 0 ? 0 : lds[t - 1];
    dm = gl::pow(Dinv, j0);
    u64 dp = gl::pow(D, j0);
    for (u32 i = j0; i < j1; i++) {
        pre = gl::add(pre, gl::mul(tmp[(size_t)a.num_sldc * a.n + (lr.first_lut - i)], dm));
        zl[lr.first_lut - i] = gl::mul(pre, dp);
        dm = gl::mul(dm, Dinv);
        dp = gl::mul(dp, D);
    }
}

}  // namespace p2k

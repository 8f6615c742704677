// GPU half of the C ABI (include/p2aes.h): p2_circuit_load, p2_prove_batch(_device), debug reads, primitives.
// Replaces `CircuitData::prove(pw)` (reference call sites: SURVEY.md A.2) with a sequence of HIP kernels on one
// stream per circuit handle; no host synchronisation between stages (Fiat-Shamir runs in a device kernel).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>

#include <algorithm>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>

#include "capi_common.h"
#include "os_random.h"
#include "kernels.h"
#include "kernels2.h"

using namespace p2;
using namespace p2k;

#define HIPCHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                       \
            return P2_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

struct Tree {
    u64* dig = nullptr;  // [batch][4 * 2^(bits+1)]
    u32 bits = 0;        // log2(#leaves)
    size_t stride() const { return (size_t)8 << bits; }
};

struct Workspace {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;  // recorded after the last kernel of a call; the caller's stream waits on it
    u32* d_input_slots = nullptr;      // slot of every input target, as last uploaded to this workspace
    std::vector<u32> h_input_slots;    // host copy: re-uploaded only when a call brings a different target list
    u64* d_input_values = nullptr;
    u64* d_values = nullptr;
    u32* d_mult = nullptr;
    int* d_status = nullptr;
    u64* d_advice = nullptr;
    u64 *d_wires = nullptr, *d_wcoef = nullptr, *d_wlde = nullptr;
    u64 *d_zs = nullptr, *d_zcoef = nullptr, *d_zlde = nullptr, *d_permq = nullptr, *d_perm_seg = nullptr, *d_fri_seg = nullptr, *d_lktmp = nullptr;
    u64 *d_qvals = nullptr, *d_qres = nullptr, *d_qcoef = nullptr, *d_qlde = nullptr;
    Tree wtree, ztree, qtree;
    ChalState* d_chal_state = nullptr;
    u64* d_chal = nullptr;
    u64 *d_pows = nullptr, *d_ev = nullptr, *d_obs = nullptr, *d_comp = nullptr, *d_apow = nullptr;
    u64* d_fri_coef[9] = {nullptr};  // [2][n_r]
    u64* d_fri_vals[9] = {nullptr};  // [2][8 n_r]
    Tree fri_tree[9];
    unsigned long long* d_pow_best = nullptr;
    u32* d_pow_list = nullptr;  // [chunk] unsolved proofs + [1] their count (proof-of-work phases)
    uint8_t* d_proofs = nullptr;
    PolyRef* d_polyrefs = nullptr;
    EvalRef* d_evalrefs = nullptr;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
};

struct p2_circuit {
    Circuit c;
    int device = 0;
    hipStream_t stream = nullptr;  // setup stream (= ws[0] once workspaces exist)
    size_t n = 0, N = 0;
    u32 logn = 0, lde_bits = 0, active_wires = 0;
    size_t pbytes = 0;
    std::vector<u32> arities;
    // ---- static device data
    Op* d_ops = nullptr;
    WLevel* d_wlevels = nullptr;
    WChain* d_wchains = nullptr;
    u32 witness_levels = 0, witness_chains = 0;
    int32_t* d_wire_slot = nullptr;
    u64* d_lut_ent = nullptr;
    u32 *d_lut_pairs = nullptr, *d_lut_offsets = nullptr, *d_num_lookups = nullptr;
    LookupRows* d_lookup_rows = nullptr;
    int32_t* d_pos_index = nullptr;  // [n] advice block of a PoseidonGate row, else -1
    u32 *d_blind_rows = nullptr, *d_blind_zrows = nullptr;
    ZkKey zk_key{};      // blinding PRF key (zk circuits): OS randomness at load, or p2_circuit_set_zk_key
    u64 zk_counter = 0;  // proofs attempted under this key; never reused, also when a batch fails
    size_t total_lut_entries = 0;
    u64 *d_sigmas = nullptr, *d_k_is = nullptr, *d_subgroup = nullptr;
    u64 *d_tw_fwd = nullptr, *d_tw_inv = nullptr;  // w^k / w^-k for k < n_max/2, n_max = n
    u64 *d_tw_fwd_full = nullptr, *d_tw_inv_full = nullptr;  // w^k / w^-k for k < n (two-pass NTT, n > 2^14)
    u64* d_tw_fwd_round[9] = {nullptr};                       // order n_r tables for FRI rounds that still need two passes
    u64* d_shift_pows[9] = {nullptr};              // per FRI round r (0 = main LDE): [8][n_r] (s_r w^j)^i
    u64* d_shift_tw = nullptr;                     // main LDE, 2^13 <= n <= 2^14: [8][n/2] d_shift_pows[0][j][i] * w^i
    u64* d_shift_inv_pows = nullptr;               // [8][n] (g w^j)^-i / n   (quotient inverse)
    u64 *d_xs = nullptr, *d_l0 = nullptr, *d_zh_inv = nullptr, *d_w8inv = nullptr, *d_qscale = nullptr;
    u64 *d_pre_coeffs = nullptr, *d_pre_lde = nullptr;
    Tree pre_tree;
    u64* d_digest = nullptr;  // circuit digest (4)
    std::vector<u64> verifier_data;
    u32 n_b0 = 0, n_b1 = 0, n_evalrefs = 0;
    u32 *d_map_obs = nullptr, *d_map_ser = nullptr;
    u32 n_obs = 0, n_ser = 0, ev_count = 0;
    // ---- per-stream workspaces: chunks are dealt round-robin to streams so that the latency-bound stages of one
    // chunk (witness levels, Fiat-Shamir, PoW tail) overlap with the Poseidon-heavy stages of another
    std::vector<struct Workspace*> ws;
    Workspace* cur = nullptr;  // workspace the host thread is currently enqueueing into (under `mu`)
    Workspace setup_ws;        // used before any per-chunk workspace exists (preprocessing, primitives)
    hipStream_t cur_stream() { return cur ? cur->stream : stream; }
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>>& cur_pending() { return cur ? cur->pending : setup_ws.pending; }
    hipEvent_t ev_witness = nullptr;  // end of the latest witness kernel on any proving stream
    bool witness_recorded = false;
    u32 ws_inputs = 0;
    size_t chunk = 0, ws_alloc_begin = 0;  // allocs[ws_alloc_begin..] belong to the workspaces
    bool ws_allocs_open = false;
    // tuning options (p2_circuit_set_option; the environment is read ONCE, at load): proofs per chunk, proving streams,
    // phase timing of the host path on stderr
    size_t opt_chunk = 128, opt_streams = 2;
    std::map<const u64*, u64*> pass1_out_tw;  // two-pass NTT: output-twiddle table per full twiddle table (ensure_pass1_table)
    bool opt_merkle_top = true;               // P2AES_MERKLE_TOP=0: every level its own launch (A/B measurements)
    int opt_pass1_waves = 4;                  // P2AES_PASS1_WAVES=2: the 178-register, scratch-free build of the pass-1 kernel (A/B measurements)
    bool opt_quotient_two_walks = false;      // P2AES_QUOTIENT_TWO_WALKS: round 2's form of the quotient kernel (A/B measurements)
    bool opt_pass1_noswizzle = false;         // P2AES_PASS1_NOSWIZZLE: pass-1 workgroups in launch order (A/B measurements)
    bool opt_pass1_radix2 = false;            // P2AES_PASS1_RADIX2: the round-2 pass-1 kernel (A/B measurements)
    // Longest chain of the witness schedule (P2AES_WITNESS_FUSE at load, 1..8).  Default 1 = no chains: measured on the 64 KiB
    // circuit, chains of 8 cut the levels from 12.4 k to 2.6 k and change nothing (73 vs 66 ms per 16 witnesses, 17.4 vs 17.6
    // proofs/s) -- the kernel is bound by one compute unit's address path, not by its depth -- and the chain executor costs
    // 252 VGPRs against 126.  The contraction is what a several-CUs-per-witness kernel would need; it stays selectable.
    u32 opt_witness_fuse = 1;
    bool opt_debug_timing = false;
    // host-path staging (p2_prove_batch): persistent device buffers + pinned host buffers, one set per concurrent caller
    std::vector<struct Staging*> staging_free;
    std::mutex staging_mu;
    long fail_alloc_after = -1;            // test hook, see dalloc_ws
    // timing
    bool timing_on = false;
    std::map<std::string, std::pair<float, u32>> times;
    std::vector<void*> allocs;
    std::mutex mu;
};

template <class T>
static int dalloc(p2_circuit* C, T** p, size_t count) {
    void* q = nullptr;
    HIPCHECK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    C->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}
template <class T>
static int upload(p2_circuit* C, T** p, const T* host, size_t count) {
    if (dalloc(C, p, count)) return P2_ERR_HIP;
    if (count) HIPCHECK(hipMemcpy(*p, host, count * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// Launch with optional per-kernel event timing on the proving stream.
#define LAUNCH(C, name, kernel, grid, block, shmem, ...)                                              \
    do {                                                                                              \
        hipEvent_t _e0 = nullptr, _e1 = nullptr;                                                      \
        if ((C)->timing_on) {                                                                         \
            (void)hipEventCreate(&_e0);                                                                     \
            (void)hipEventCreate(&_e1);                                                                     \
            (void)hipEventRecord(_e0, (C)->cur_stream());                                                         \
        }                                                                                             \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (C)->cur_stream(), __VA_ARGS__);                     \
        if ((C)->timing_on) {                                                                         \
            (void)hipEventRecord(_e1, (C)->cur_stream());                                                         \
            (C)->cur_pending().push_back({name, {_e0, _e1}});                                               \
        }                                                                                             \
        HIPCHECK(hipGetLastError());                                                                  \
    } while (0)

static inline dim3 g1(size_t work, u32 block, u32 y = 1, u32 z = 1) { return dim3((u32)((work + block - 1) / block), y, z); }

// ---------------------------------------------------------------------------------- building blocks
static const u32 R16_MIN_BITS = 8;  // the register-blocked kernel needs n / 16 threads >= a few waves; smaller transforms keep k_ntt_lds
static size_t r16_lds_bytes(int logn) { return 8 * (((size_t)1 << logn) + ((size_t)1 << (logn - 4))); }
static int run_ntt(p2_circuit* C, const char* name, NttArgs a, u32 cols, u32 batch) {
    a.log_nmax = (int)C->logn;
    if ((u32)a.logn >= 13 && !a.bitrev_in && !a.bitrev_out && !a.post) {
        // half a column per workgroup: two (or more) workgroups per compute unit overlap each other's memory phases
        LAUNCH(C, name, k_ntt_r16<true>, dim3(2 * cols * a.cosets, batch), dim3(1u << (a.logn - 5)), r16_lds_bytes(a.logn - 1), a);
        return 0;
    }
    if ((u32)a.logn >= R16_MIN_BITS) {
        LAUNCH(C, name, k_ntt_r16<false>, dim3(cols * a.cosets, batch), dim3(1u << (a.logn - 4)), r16_lds_bytes(a.logn), a);
        return 0;
    }
    size_t shmem = (size_t)8 << a.logn;
    LAUNCH(C, name, k_ntt_lds, dim3(cols * a.cosets, batch), dim3(1024), shmem, a);
    return 0;
}
static const u32 LDS_NTT_MAX_BITS = 14;  // whole transform in one workgroup's LDS up to 2^14 points
// The output twiddles of pass 1 for the order-2^logn table `tw_full`, in output order (k_pass1_out_tw); built at load, never
// while workspaces are open (allocations made then belong to the workspaces).
static int ensure_pass1_table(p2_circuit* C, const u64* tw_full, u32 logn) {
    if (logn <= LDS_NTT_MAX_BITS || C->pass1_out_tw.count(tw_full)) return 0;
    u64* t = nullptr;
    if (dalloc(C, &t, (size_t)1 << logn)) return P2_ERR_HIP;
    hipLaunchKernelGGL(k_pass1_out_tw, g1((size_t)1 << logn, 256), dim3(256), 0, C->stream, tw_full, t, (int)logn, (int)(logn - 12));
    HIPCHECK(hipGetLastError());
    C->pass1_out_tw[tw_full] = t;
    return 0;
}
// two-pass transform of `cols*cosets` blocks: natural order in `in`, bit-reversed order in `out`
static int ntt_big(p2_circuit* C, const char* name, const u64* in, u64* out, const u64* tw_full, const u64* pre, u32 logn, u32 cols, u32 cosets,
                   const u32* block_of_coset, int in_coset_blocks, size_t in_col_stride, size_t out_col_stride, size_t in_batch_stride,
                   size_t out_batch_stride, u64 post_scalar, u32 batch) {
    const u32 log_n2 = 12, log_n1 = logn - log_n2;
    u32 log_T = 12 - log_n1;  // tile of n1 x T = 4096 elements (32 KiB LDS)
    if (log_T < 2) return set_error("transform too large for the two-pass NTT"), P2_ERR_INVALID;
    Pass1Args a{};
    a.in = in;
    a.out = out;
    a.tw = tw_full;
    a.pre = pre;
    a.in_col_stride = in_col_stride;
    a.out_col_stride = out_col_stride;
    a.in_batch_stride = in_batch_stride;
    a.out_batch_stride = out_batch_stride;
    a.logn = (int)logn;
    a.log_n1 = (int)log_n1;
    a.log_T = (int)log_T;
    a.cosets = (int)cosets;
    a.in_coset_blocks = in_coset_blocks;
    for (u32 j = 0; j < 8; j++) a.block_of_coset[j] = block_of_coset ? block_of_coset[j] : 0;
    u32 tiles = (1u << log_n2) >> log_T;
    auto it = C->pass1_out_tw.find(tw_full);
    if (it == C->pass1_out_tw.end()) return set_error("internal: no pass-1 twiddle table for this transform"), P2_ERR_INVALID;
    a.out_tw = it->second;
    a.xcd_swizzle = (cosets > 1 && !in_coset_blocks && !C->opt_pass1_noswizzle) ? 1 : 0;  // only where workgroups share their input
    const std::string name1 = std::string(name) + "_pass1";  // the two passes are timed apart
    if (C->opt_pass1_radix2)
        LAUNCH(C, name1, k_ntt_pass1, dim3(tiles * cols * cosets, batch), dim3(256), (size_t)8 << 12, a);
    else if (C->opt_pass1_waves == 2)
        LAUNCH(C, name1, k_ntt_pass1_r16<2>, dim3(tiles * cols * cosets, batch), dim3(256), r16_lds_bytes(12), a);
    else
        LAUNCH(C, name1, k_ntt_pass1_r16<4>, dim3(tiles * cols * cosets, batch), dim3(256), r16_lds_bytes(12), a);
    // pass 2: every row of n2 contiguous points, in place
    if (out_col_stride != ((size_t)cosets << logn)) return set_error("internal: two-pass NTT needs densely packed output blocks"), P2_ERR_INVALID;
    NttArgs b{};
    b.in = out;
    b.out = out;
    b.tw = tw_full;
    b.post_scalar = post_scalar;
    b.in_col_stride = b.out_col_stride = (size_t)1 << log_n2;
    b.in_batch_stride = b.out_batch_stride = out_batch_stride;
    b.logn = (int)log_n2;
    b.log_nmax = (int)logn;
    b.cosets = 1;
    size_t rows = ((size_t)cols * cosets) << log_n1;
    // grid.x is limited to 2^31-1; rows*1 fits for every supported size
    LAUNCH(C, name, k_ntt_r16<false>, dim3((u32)rows, batch), dim3(1u << (log_n2 - 4)), r16_lds_bytes((int)log_n2), b);
    return 0;
}
// values [cols][n] -> coeffs [cols][n].  `scratch` ([cols][n] per proof, same batch stride) is needed when n > 2^14.
static int intt_cols(p2_circuit* C, const u64* vals, u64* coeffs, u32 cols, size_t batch_stride, u32 batch, u64* scratch = nullptr,
                     size_t scratch_batch_stride = 0) {
    if (C->logn > LDS_NTT_MAX_BITS) {
        if (!scratch) return set_error("internal: large iNTT needs scratch"), P2_ERR_INVALID;
        if (ntt_big(C, "intt", vals, scratch, C->d_tw_inv_full, nullptr, C->logn, cols, 1, nullptr, 0, C->n, C->n, batch_stride, scratch_batch_stride,
                    gl::inv((u64)C->n % gl::P), batch))
            return P2_ERR_HIP;
        LAUNCH(C, "bitrev_copy", k_bitrev_copy, g1(C->n, 256, batch, cols), dim3(256), 0, scratch, C->n, scratch_batch_stride, coeffs, C->n, batch_stride,
               (int)C->logn, 1u, (const u64*)nullptr, 0u);
        return 0;
    }
    NttArgs a{};
    a.in = vals;
    a.out = coeffs;
    a.tw = C->d_tw_inv;
    a.post_scalar = gl::inv((u64)C->n % gl::P);
    a.in_col_stride = a.out_col_stride = C->n;
    a.in_batch_stride = a.out_batch_stride = batch_stride;
    a.logn = (int)C->logn;
    a.cosets = 1;
    a.bitrev_out = 1;
    return run_ntt(C, "intt", a, cols, batch);
}
// coeffs [cols][n_r] -> lde [cols][8 n_r] (bit-reversed order)
static int lde_cols(p2_circuit* C, const u64* coeffs, size_t in_batch_stride, u64* lde, size_t out_batch_stride, u32 cols, u32 round, u32 batch) {
    u32 logn_r = C->logn;
    for (u32 r = 0; r < round; r++) logn_r -= C->arities[r];
    u32 blocks[8];
    for (u32 j = 0; j < 8; j++) blocks[j] = gl::bitrev(j, (int)C->c.cfg.rate_bits);
    if (logn_r > LDS_NTT_MAX_BITS) {
        // twiddles of order n_r are a stride of the order-n table
        if (round != 0 && C->d_tw_fwd_round[round] == nullptr) return set_error("internal: missing round twiddles"), P2_ERR_INVALID;
        const u64* tw = round == 0 ? C->d_tw_fwd_full : C->d_tw_fwd_round[round];
        return ntt_big(C, "lde", coeffs, lde, tw, C->d_shift_pows[round], logn_r, cols, 8, blocks, 0, (size_t)1 << logn_r, (size_t)8 << logn_r,
                       in_batch_stride, out_batch_stride, 1, batch);
    }
    NttArgs a{};
    a.in = coeffs;
    a.out = lde;
    a.tw = C->d_tw_fwd;
    a.pre = C->d_shift_pows[round];
    a.pre_tw = round == 0 ? C->d_shift_tw : nullptr;
    a.post_scalar = 1;
    a.in_col_stride = (size_t)1 << logn_r;
    a.out_col_stride = (size_t)8 << logn_r;
    a.in_batch_stride = in_batch_stride;
    a.out_batch_stride = out_batch_stride;
    a.logn = (int)logn_r;
    a.cosets = 1 << C->c.cfg.rate_bits;
    for (u32 j = 0; j < 8; j++) a.block_of_coset[j] = blocks[j];
    return run_ntt(C, "lde", a, cols, batch);
}
// The levels above the leaf digests, up to the cap: wide levels one launch each, the top (at most 256 parents per cap
// subtree, up to nine levels) in one launch of a workgroup per cap node.
static int merkle_levels(p2_circuit* C, Tree& t, u32 batch) {
    const u32 cap_h = C->c.cfg.cap_height;
    if (t.bits <= cap_h) return 0;
    const u32 levels = t.bits - cap_h;                       // level l: 2^(bits-l-1) parents
    // The last `fused` levels (parents per cap subtree 2^(fused-1) .. 1) run as ONE launch of a workgroup per cap node -- for
    // SMALL batches only.  These levels are latency bound either way (a level is one permutation deep whatever its width), so
    // what the fusion buys is launches: 54 -> 18 per chunk, 118 -> 70 for the whole pipeline.  For a full chunk it costs time:
    // the waves of a fused walk stay resident for all its levels and slow each other down, where separately launched levels
    // shrink to one wave per SIMD as they narrow (measured per 128-proof chunk: nine levels in 256-thread workgroups + 3.3 ms,
    // seven levels in one wave + 1.3 ms against 21.1 ms).  Launch count does not matter there: the chip is busy throughout.
    const u32 fused = (C->opt_merkle_top && batch <= 16) ? std::min<u32>(levels, 9) : 0;
    const u32 top_threads = 256;
    const size_t leaves = (size_t)1 << t.bits;
    for (u32 l = 0; l < levels - fused; l++) {
        size_t parents = leaves >> (l + 1);
        size_t off_c = 4 * (((size_t)2 << t.bits) - ((size_t)2 << (t.bits - l)));
        size_t off_p = 4 * (((size_t)2 << t.bits) - ((size_t)2 << (t.bits - l - 1)));
        LAUNCH(C, "merkle_level", k_merkle_level, g1(parents, 256, batch), dim3(256), 0, t.dig + off_c, t.dig + off_p, parents, t.stride());
    }
    if (fused) {
        // parents per cap subtree at the fused levels: 2^(fused-1) .. 1; those with more than 32 one thread per node, the rest cooperative
        const u32 direct = fused > 6 ? fused - 6 : 0;
        if (direct) LAUNCH(C, "merkle_top", k_merkle_top, dim3(1u << cap_h, batch), dim3(top_threads), 0, t.dig, t.stride(), t.bits, levels - fused, direct);
        LAUNCH(C, "merkle_top", k_merkle_top_coop, dim3(1u << cap_h, batch), dim3(256), 0, t.dig, t.stride(), t.bits, levels - fused + direct, fused - direct);
    }
    return 0;
}
static int merkle_build(p2_circuit* C, const u64* data, u32 cols, u32 active, size_t col_stride, size_t batch_stride, Tree& t, u32 batch) {
    size_t leaves = (size_t)1 << t.bits;
    LAUNCH(C, "hash_leaves", k_hash_leaves, g1(leaves, 256, batch), dim3(256), 0, data, (int)cols, (int)active, col_stride, batch_stride, leaves, t.dig,
           t.stride());
    return merkle_levels(C, t, batch);
}
static size_t cap_off(const Tree& t, u32 cap_height) {
    u32 l = t.bits - cap_height;
    return 4 * (((size_t)2 << t.bits) - ((size_t)2 << (t.bits - l)));
}
static int challenger(p2_circuit* C, u32 stage, const u64* observe, size_t stride, u32 len, u32 aux, u64 mod, u32 batch) {
    ChalArgs a{};
    a.st = C->cur->d_chal_state;
    a.chal = C->cur->d_chal;
    a.observe = observe;
    a.observe_stride = stride;
    a.observe_len = len;
    a.batch = batch;
    a.stage = stage;
    a.aux = aux;
    a.mod = mod;
    a.digest = C->d_digest;
    a.status = C->cur->d_status;
    LAUNCH(C, "challenger", k_challenger, g1((size_t)batch * 16, 64), dim3(64), 0, a);  // a 16-lane group per proof
    return 0;
}

// ---------------------------------------------------------------------------------- load / preprocess
static int circuit_setup(p2_circuit* C) {
    const Circuit& c = C->c;
    const size_t n = C->n, N = C->N;
    const u32 R = c.cfg.num_routed_wires, ncc = c.num_constants_cols(), np = c.num_preprocessed();
    if (c.cfg.num_challenges > 2) return set_error("k_perm_chunks handles at most two challenges"), P2_ERR_INVALID;
    if (c.num_partial_products() + 1 > PERM_MAX_CHUNKS) return set_error("more partial-product chunks than k_perm_scan holds in registers"), P2_ERR_INVALID;
    {
        // the witness program, rescheduled for the device: contracted critical chains + single ops per level (witness_schedule.h)
        WitnessSchedule ws = schedule_witness(c, std::min<u32>(C->opt_witness_fuse, WITNESS_KMAX));
        if (ws.max_chain > (u32)WITNESS_KMAX) return set_error("internal: witness chain longer than the kernel is unrolled for"), P2_ERR_INVALID;
        C->witness_levels = (u32)ws.levels.size();
        C->witness_chains = (u32)ws.chains.size();
        if (upload(C, &C->d_ops, ws.ops.data(), ws.ops.size())) return P2_ERR_HIP;
        if (upload(C, &C->d_wlevels, ws.levels.data(), ws.levels.size())) return P2_ERR_HIP;
        if (upload(C, &C->d_wchains, ws.chains.data(), ws.chains.size())) return P2_ERR_HIP;
    }
    if (upload(C, &C->d_wire_slot, c.wire_slot.data(), c.wire_slot.size())) return P2_ERR_HIP;
    {
        // witness generation resolves a lookup with ONE load: input value -> (flat entry index << 16) | output
        std::vector<u64> ent(c.luts.size() * 65536, ~0ull);
        std::vector<u32> pairs, offs(1, 0);
        for (size_t l = 0; l < c.luts.size(); l++) {
            for (size_t i = 0; i < c.luts[l].size(); i++) {
                auto pr = c.luts[l][i];
                if (ent[l * 65536 + pr.first] == ~0ull) ent[l * 65536 + pr.first] = ((u64)pairs.size() << 16) | pr.second;
                pairs.push_back((u32)pr.first | ((u32)pr.second << 16));
            }
            offs.push_back((u32)pairs.size());
        }
        C->total_lut_entries = pairs.size();
        if (upload(C, &C->d_lut_ent, ent.data(), ent.size())) return P2_ERR_HIP;
        if (upload(C, &C->d_lut_pairs, pairs.data(), pairs.size())) return P2_ERR_HIP;
        if (upload(C, &C->d_lut_offsets, offs.data(), offs.size())) return P2_ERR_HIP;
        if (upload(C, &C->d_num_lookups, c.num_lookups.data(), c.num_lookups.size())) return P2_ERR_HIP;
        if (upload(C, &C->d_lookup_rows, c.lookup_rows.data(), c.lookup_rows.size())) return P2_ERR_HIP;
    }
    {
        std::vector<int32_t> pi(n, -1);
        for (size_t k = 0; k < c.poseidon_rows.size(); k++) pi[c.poseidon_rows[k]] = (int32_t)k;
        if (upload(C, &C->d_pos_index, pi.data(), n)) return P2_ERR_HIP;
    }
    if (upload(C, &C->d_blind_rows, c.blind_rows.data(), c.blind_rows.size())) return P2_ERR_HIP;
    if (upload(C, &C->d_blind_zrows, (const u32*)c.blind_zrows.data(), 2 * c.blind_zrows.size())) return P2_ERR_HIP;
    if (upload(C, &C->d_sigmas, c.sigmas.data(), c.sigmas.size())) return P2_ERR_HIP;
    if (upload(C, &C->d_k_is, c.k_is.data(), c.k_is.size())) return P2_ERR_HIP;
    // twiddles, subgroup, coset tables (host-computed once; O(n) field ops)
    {
        std::vector<u64> sub(n), twi(n);
        u64 w = gl::root_of_unity((int)C->logn), wi = gl::inv(w), x = 1, xi = 1;
        for (size_t i = 0; i < n; i++) {
            sub[i] = x;
            twi[i] = xi;
            x = gl::mul(x, w);
            xi = gl::mul(xi, wi);
        }
        if (upload(C, &C->d_subgroup, sub.data(), n)) return P2_ERR_HIP;
        if (upload(C, &C->d_tw_inv_full, twi.data(), n)) return P2_ERR_HIP;
        C->d_tw_fwd_full = C->d_subgroup;  // w^k, k < n
        C->d_tw_fwd = C->d_tw_fwd_full;    // the single-pass kernel only indexes k < n/2
        C->d_tw_inv = C->d_tw_inv_full;
        if (ensure_pass1_table(C, C->d_tw_fwd_full, C->logn) || ensure_pass1_table(C, C->d_tw_inv_full, C->logn)) return P2_ERR_HIP;
        // FRI rounds whose polynomial is still > 2^14 need their own order-n_r table
        u32 logn_r = C->logn;
        for (u32 r = 0; r < C->arities.size(); r++) {
            logn_r -= C->arities[r];
            if (logn_r > LDS_NTT_MAX_BITS) {
                size_t n_r = (size_t)1 << logn_r;
                std::vector<u64> t(n_r);
                for (size_t i = 0; i < n_r; i++) t[i] = sub[i << (C->logn - logn_r)];
                if (upload(C, &C->d_tw_fwd_round[r + 1], t.data(), n_r)) return P2_ERR_HIP;
                if (ensure_pass1_table(C, C->d_tw_fwd_round[r + 1], logn_r)) return P2_ERR_HIP;
            }
        }
    }
    {
        // LDE shift tables for round r: bases s_{r,j} = g^(16^r) * w_{8 n_r}^j
        u32 logn_r = C->logn;
        u64 shift = gl::MULT_GEN;
        for (u32 r = 0; r <= C->arities.size(); r++) {
            size_t n_r = (size_t)1 << logn_r;
            std::vector<u64> bases(8);
            u64 wl = gl::root_of_unity((int)(logn_r + c.cfg.rate_bits));
            for (u32 j = 0; j < 8; j++) bases[j] = gl::mul(shift, gl::pow(wl, j));
            u64* d_b;
            if (upload(C, &d_b, bases.data(), 8)) return P2_ERR_HIP;
            if (dalloc(C, &C->d_shift_pows[r], 8 * n_r)) return P2_ERR_HIP;
            hipLaunchKernelGGL(k_pow_table, g1(n_r, 256, 8), dim3(256), 0, C->stream, C->d_shift_pows[r], d_b, (u32)n_r, (u64)1);
            if (r == 0 && C->logn >= 13 && C->logn <= LDS_NTT_MAX_BITS) {
                if (dalloc(C, &C->d_shift_tw, 8 * (n_r / 2))) return P2_ERR_HIP;
                hipLaunchKernelGGL(k_mul_tables, g1(n_r / 2, 256, 8), dim3(256), 0, C->stream, C->d_shift_tw, C->d_shift_pows[0], n_r, C->d_tw_fwd, 0, (u32)(n_r / 2));
            }
            if (r == 0) {
                std::vector<u64> ib(8);
                for (u32 j = 0; j < 8; j++) ib[j] = gl::inv(bases[j]);
                u64* d_ib;
                if (upload(C, &d_ib, ib.data(), 8)) return P2_ERR_HIP;
                if (dalloc(C, &C->d_shift_inv_pows, 8 * n_r)) return P2_ERR_HIP;
                hipLaunchKernelGGL(k_pow_table, g1(n_r, 256, 8), dim3(256), 0, C->stream, C->d_shift_inv_pows, d_ib, (u32)n_r, gl::inv((u64)n % gl::P));
            }
            if (r < C->arities.size()) {
                shift = gl::pow(shift, (u64)1 << C->arities[r]);
                logn_r -= C->arities[r];
            }
        }
        HIPCHECK(hipGetLastError());
    }
    {
        // per-point tables on the LDE coset (position p <-> natural index rev(p))
        std::vector<u64> xs(N), l0(N), zh_inv(8), w8inv(8), qscale(8);
        u64 wl = gl::root_of_unity((int)C->lde_bits);
        std::vector<u64> nat(N);
        u64 x = gl::MULT_GEN;
        for (size_t i = 0; i < N; i++) {
            nat[i] = x;
            x = gl::mul(x, wl);
        }
        u64 gn = gl::pow(gl::MULT_GEN, n), w8 = gl::root_of_unity((int)c.cfg.rate_bits);
        std::vector<u64> zh(8);
        for (u32 j = 0; j < 8; j++) {
            zh[j] = gl::sub(gl::mul(gn, gl::pow(w8, j)), 1);
            zh_inv[j] = gl::inv(zh[j]);
            w8inv[j] = gl::inv(gl::pow(w8, j));
            qscale[j] = gl::mul(gl::inv(gl::pow(gn, j)), gl::inv(8));
        }
        // batch inversion of n*(x-1)
        std::vector<u64> den(N), pref(N);
        u64 acc = 1;
        for (size_t i = 0; i < N; i++) {
            den[i] = gl::mul((u64)n % gl::P, gl::sub(nat[i], 1));
            pref[i] = acc;
            acc = gl::mul(acc, den[i]);
        }
        u64 inv_all = gl::inv(acc);
        for (size_t i = N; i-- > 0;) {
            u64 di = gl::mul(inv_all, pref[i]);
            inv_all = gl::mul(inv_all, den[i]);
            size_t p = gl::bitrev((u32)i, (int)C->lde_bits);
            xs[p] = nat[i];
            l0[p] = gl::mul(zh[i & 7], di);
        }
        if (upload(C, &C->d_xs, xs.data(), N)) return P2_ERR_HIP;
        if (upload(C, &C->d_l0, l0.data(), N)) return P2_ERR_HIP;
        if (upload(C, &C->d_zh_inv, zh_inv.data(), 8)) return P2_ERR_HIP;
        if (upload(C, &C->d_w8inv, w8inv.data(), 8)) return P2_ERR_HIP;
        if (upload(C, &C->d_qscale, qscale.data(), 8)) return P2_ERR_HIP;
    }
    // constants | sigmas commitment on the device
    {
        u64* d_vals;
        if (dalloc(C, &d_vals, (size_t)np * n)) return P2_ERR_HIP;
        HIPCHECK(hipMemcpy(d_vals, c.constants.data(), (size_t)ncc * n * 8, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(d_vals + (size_t)ncc * n, c.sigmas.data(), (size_t)R * n * 8, hipMemcpyHostToDevice));
        if (dalloc(C, &C->d_pre_coeffs, (size_t)np * n)) return P2_ERR_HIP;
        if (dalloc(C, &C->d_pre_lde, (size_t)np * N)) return P2_ERR_HIP;
        C->pre_tree.bits = C->lde_bits;
        if (dalloc(C, &C->pre_tree.dig, C->pre_tree.stride())) return P2_ERR_HIP;
        if (intt_cols(C, d_vals, C->d_pre_coeffs, np, 0, 1, C->d_pre_lde, 0)) return P2_ERR_HIP;
        if (lde_cols(C, C->d_pre_coeffs, 0, C->d_pre_lde, 0, np, 0, 1)) return P2_ERR_HIP;
        if (merkle_build(C, C->d_pre_lde, np, np, N, 0, C->pre_tree, 1)) return P2_ERR_HIP;
        HIPCHECK(hipStreamSynchronize(C->stream));
        size_t cap_n = (size_t)1 << c.cfg.cap_height;
        std::vector<u64> cap(4 * cap_n);
        HIPCHECK(hipMemcpy(cap.data(), C->pre_tree.dig + cap_off(C->pre_tree, c.cfg.cap_height), cap.size() * 8, hipMemcpyDeviceToHost));
        // circuit digest = hash_no_pad(cap || hash_pad([]) || degree_bits)   (a dozen host permutations)
        std::vector<u64> parts(cap);
        {
            u64 st[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};  // hash_pad of the empty domain separator
            u64 s2[12] = {0};
            for (int i = 0; i < 8; i++) s2[i] = st[i];
            gl::poseidon(s2);
            for (int i = 0; i < 4; i++) s2[i] = st[8 + i];
            gl::poseidon(s2);
            for (int i = 0; i < 4; i++) parts.push_back(s2[i]);
        }
        parts.push_back(c.degree_bits);
        u64 st[12] = {0};
        for (size_t off = 0; off < parts.size(); off += 8) {
            for (size_t i = 0; i < std::min<size_t>(8, parts.size() - off); i++) st[i] = parts[off + i];
            gl::poseidon(st);
        }
        C->verifier_data = cap;
        for (int i = 0; i < 4; i++) C->verifier_data.push_back(st[i]);
        if (upload(C, &C->d_digest, st, 4)) return P2_ERR_HIP;
    }
    return 0;
}

static int setup_polyrefs(p2_circuit* C);
static void collect_timing(p2_circuit* C);
// Drains and releases every per-stream workspace; the handle is left with none (chunk == 0), ready to allocate again.
static void release_workspaces(p2_circuit* C) {
    for (Workspace* W : C->ws) {
        if (W->stream) (void)hipStreamSynchronize(W->stream);
        for (auto& pe : W->pending) {
            (void)hipEventDestroy(pe.second.first);
            (void)hipEventDestroy(pe.second.second);
        }
        if (W->stream) (void)hipStreamDestroy(W->stream);
        if (W->done) (void)hipEventDestroy(W->done);
        delete W;
    }
    C->ws.clear();
    if (C->ws_allocs_open) {
        for (size_t i = C->ws_alloc_begin; i < C->allocs.size(); i++) (void)hipFree(C->allocs[i]);
        C->allocs.resize(C->ws_alloc_begin);
        C->ws_allocs_open = false;
    }
    C->chunk = 0;
    C->ws_inputs = 0;
    C->cur = nullptr;
    C->witness_recorded = false;
}
// Test hook (P2AES_TEST_FAIL_ALLOC_AFTER=k in the environment when the handle is loaded): the k-th workspace allocation
// fails as if the device were out of memory.  Exercises the roll-back below without needing a full HBM.
static int dalloc_ws(p2_circuit* C, size_t& counter, void** p, size_t bytes) {
    if (C->fail_alloc_after >= 0 && (long)counter++ == C->fail_alloc_after) {
        C->fail_alloc_after = -1;  // one shot: the retry must succeed
        return set_error("hipMalloc: injected allocation failure (test hook)"), P2_ERR_HIP;
    }
    void* q = nullptr;
    HIPCHECK(hipMalloc(&q, std::max<size_t>(bytes, 8)));
    C->allocs.push_back(q);
    *p = q;
    return 0;
}
static int build_workspace(p2_circuit* C, Workspace* W, size_t chunk, u32 ws_inputs, size_t& counter) {
    const Circuit& c = C->c;
    const size_t n = C->n, N = C->N;
    const u32 zc = c.num_zs_cols(), qc = c.num_quotient_cols(), NC = c.cfg.num_challenges, act = C->active_wires;
    if (hipStreamCreateWithFlags(&W->stream, hipStreamNonBlocking) != hipSuccess) return set_error("hipStreamCreate failed"), P2_ERR_HIP;
    if (hipEventCreateWithFlags(&W->done, hipEventDisableTiming) != hipSuccess) return set_error("hipEventCreate failed"), P2_ERR_HIP;
#define WS_ALLOC(field, count)                                                                       \
    if (dalloc_ws(C, counter, (void**)&(field), (size_t)(count) * sizeof(*(field)))) return P2_ERR_HIP
    WS_ALLOC(W->d_input_slots, ws_inputs);
    WS_ALLOC(W->d_input_values, chunk * ws_inputs);
    WS_ALLOC(W->d_values, chunk * c.num_slots);
    WS_ALLOC(W->d_mult, chunk * std::max<size_t>(C->total_lut_entries, 1));
    WS_ALLOC(W->d_status, chunk);
    WS_ALLOC(W->d_advice, chunk * std::max<size_t>(c.poseidon_rows.size(), 1) * 55);
    WS_ALLOC(W->d_wires, chunk * act * n);
    WS_ALLOC(W->d_wcoef, chunk * act * n);
    WS_ALLOC(W->d_wlde, chunk * (act + c.salt()) * N);
    WS_ALLOC(W->d_zs, chunk * zc * n);
    WS_ALLOC(W->d_zcoef, chunk * zc * n);
    WS_ALLOC(W->d_zlde, chunk * (zc + c.salt()) * N);
    WS_ALLOC(W->d_permq, chunk * NC * (c.num_partial_products() + 1) * n);
    WS_ALLOC(W->d_perm_seg, chunk * NC * PERM_MAX_SEGS);
    WS_ALLOC(W->d_fri_seg, chunk * 4 * FRI_MAX_SEGS);
    WS_ALLOC(W->d_lktmp, chunk * NC * (c.num_sldc_polys() + 1) * n);
    WS_ALLOC(W->d_qvals, chunk * NC * N);
    WS_ALLOC(W->d_qres, chunk * NC * N);
    WS_ALLOC(W->d_qcoef, chunk * qc * n);
    WS_ALLOC(W->d_qlde, chunk * (qc + c.salt()) * N);
    for (Tree* t : {&W->wtree, &W->ztree, &W->qtree}) {
        t->bits = C->lde_bits;
        WS_ALLOC(t->dig, chunk * t->stride());
    }
    WS_ALLOC(W->d_chal_state, chunk);
    WS_ALLOC(W->d_chal, chunk * CH_WORDS);
    WS_ALLOC(W->d_pows, chunk * 8 * n);
    WS_ALLOC(W->d_ev, chunk * 2 * C->ev_count);
    WS_ALLOC(W->d_obs, chunk * 2 * C->n_obs);
    WS_ALLOC(W->d_comp, chunk * 4 * n);
    WS_ALLOC(W->d_apow, chunk * 2 * APOW_STRIDE);
    u32 logn_r = C->logn;
    for (u32 r = 0; r <= C->arities.size(); r++) {
        size_t n_r = (size_t)1 << logn_r;
        WS_ALLOC(W->d_fri_coef[r], chunk * 2 * n_r);
        if (r < C->arities.size()) {
            WS_ALLOC(W->d_fri_vals[r], chunk * 2 * 8 * n_r);
            W->fri_tree[r].bits = logn_r + c.cfg.rate_bits - C->arities[r];
            WS_ALLOC(W->fri_tree[r].dig, chunk * W->fri_tree[r].stride());
            logn_r -= C->arities[r];
        }
    }
    WS_ALLOC(W->d_pow_best, chunk);
    WS_ALLOC(W->d_pow_list, chunk + 1);
    WS_ALLOC(W->d_proofs, chunk * C->pbytes);
#undef WS_ALLOC
    C->cur = W;
    int e = setup_polyrefs(C);
    C->cur = nullptr;
    return e;
}
// Makes sure `nstreams` workspaces of `chunk` proofs and `n_inputs` input targets exist.  All-or-nothing: the new shape
// (C->chunk, C->ws_inputs, C->ws) is published only after every allocation has succeeded; on failure whatever was
// allocated is released and the handle is left without workspaces, so that a retry (with a smaller batch) allocates
// afresh instead of launching kernels on null pointers.
static int alloc_workspace(p2_circuit* C, size_t chunk, u32 n_inputs, size_t nstreams) {
    if (C->chunk >= chunk && C->ws_inputs >= n_inputs && C->ws.size() >= nstreams) return 0;
    if (C->chunk != 0) {
        // a later call wants a bigger shape (a first prove(pw) sizes the workspace for one proof; a batch follows):
        // drain the proving streams, release the old workspaces and allocate the larger ones
        if (C->timing_on) collect_timing(C);
        chunk = std::max(chunk, C->chunk);
        n_inputs = std::max(n_inputs, C->ws_inputs);
        nstreams = std::max(nstreams, C->ws.size());
        release_workspaces(C);
    }
    C->ws_alloc_begin = C->allocs.size();
    C->ws_allocs_open = true;
    const u32 ws_inputs = std::max<u32>(n_inputs, 1);
    size_t counter = 0;
    for (size_t wi = 0; wi < nstreams; wi++) {
        Workspace* W = new Workspace();
        C->ws.push_back(W);  // owned by the handle from here on: release_workspaces() frees a half-built one too
        if (build_workspace(C, W, chunk, ws_inputs, counter)) {
            std::string why = g_last_error;
            release_workspaces(C);
            set_error("workspace allocation failed (" + why + "); the handle holds no workspace now, a smaller batch may fit");
            return P2_ERR_HIP;
        }
    }
    C->chunk = chunk;
    C->ws_inputs = ws_inputs;
    return 0;
}

// ---------------------------------------------------------------------------------- the pipeline
// the target slots are already in the current workspace (d_input_slots); d_values: [batch][n_inputs] device; proofs/status: device.
static int prove_chunk(p2_circuit* C, u32 B, u32 n_inputs, const u64* d_values, uint8_t* d_proofs, int* d_status_out, u64 proof_base) {
    const Circuit& c = C->c;
    const size_t n = C->n, N = C->N;
    const u32 R = c.cfg.num_routed_wires, NC = c.cfg.num_challenges, npp = c.num_partial_products(), nlp = c.num_lookup_polys();
    const u32 zc = c.num_zs_cols(), qc = c.num_quotient_cols(), act = C->active_wires, ncc = c.num_constants_cols(), np = c.num_preprocessed();
    const u32 cap_h = c.cfg.cap_height, cap_words = 4u << cap_h, nsldc = c.num_sldc_polys();
    const u32 salt = c.salt();
    const size_t ws = (size_t)act * n, wls = (size_t)(act + salt) * N, zs_s = (size_t)zc * n, zl_s = (size_t)(zc + salt) * N, ql_s = (size_t)(qc + salt) * N;
    hipStream_t st = C->cur_stream();
    // 1. witness
    HIPCHECK(hipMemsetAsync(C->cur->d_mult, 0, (size_t)B * std::max<size_t>(C->total_lut_entries, 1) * 4, st));
    {
        WitnessArgs a{};
        a.ops = C->d_ops;
        a.levels = C->d_wlevels;
        a.chains = C->d_wchains;
        a.num_levels = C->witness_levels;
        a.num_slots = c.num_slots;
        a.n_inputs = n_inputs;
        a.input_slots = C->cur->d_input_slots;
        a.input_values = d_values;
        a.values = C->cur->d_values;
        a.lut_ent = C->d_lut_ent;
        a.mult = C->cur->d_mult;
        a.total_lut_entries = C->total_lut_entries;
        a.status = C->cur->d_status;
        a.wire_slot = C->d_wire_slot;
        a.advice = C->cur->d_advice;
        a.n = (u32)n;
        a.num_poseidon_rows = (u32)c.poseidon_rows.size();
        // 512 threads (8 waves, <= 128 VGPRs each) leave room on the compute unit: a 1024-thread workgroup needs a
        // completely empty unit and waits for the tail of whatever wide kernel the other stream is running.
        const u32 WITNESS_THREADS = 512;
        // Witness kernels of successive chunks run one after the other (each occupies only B compute units): without
        // this the two proving streams stay in lockstep -- both in witness generation with the chip idle, then both in
        // the wide kernels -- and a deep circuit's witness time is never hidden.  Chained, chunk k+1's witness runs
        // under chunk k's commitments.
        if (C->witness_recorded) HIPCHECK(hipStreamWaitEvent(C->cur->stream, C->ev_witness, 0));
        if (c.poseidon_rows.empty()) {
            if (C->witness_chains)
                LAUNCH(C, "witness", (k_witness<false, true>), dim3(B), dim3(WITNESS_THREADS), 0, a);
            else
                LAUNCH(C, "witness", (k_witness<false, false>), dim3(B), dim3(WITNESS_THREADS), 0, a);
        } else {
            if (C->witness_chains)
                LAUNCH(C, "witness", (k_witness<true, true>), dim3(B), dim3(WITNESS_THREADS), 0, a);
            else
                LAUNCH(C, "witness", (k_witness<true, false>), dim3(B), dim3(WITNESS_THREADS), 0, a);
        }
        HIPCHECK(hipEventRecord(C->ev_witness, C->cur->stream));
        C->witness_recorded = true;
    }
    LAUNCH(C, "fill_wires", k_fill_wires, g1((size_t)R * n, 256, B), dim3(256), 0, C->d_wire_slot, C->cur->d_values, C->cur->d_wires, (size_t)R * n, c.num_slots, ws,
           C->cur->d_status);
    if (act > R)
        LAUNCH(C, "fill_advice", k_fill_advice, g1((size_t)55 * n, 256, B), dim3(256), 0, C->d_pos_index, C->cur->d_advice, C->cur->d_wires, (u32)n,
               (u32)c.poseidon_rows.size(), ws);
    if (c.cfg.zero_knowledge) {
        size_t cnt = (c.blind_rows.size() * 135 + 7) / 8 + (c.blind_zrows.size() * 80 + 7) / 8;  // PRF blocks of eight elements
        LAUNCH(C, "fill_blind", k_fill_blind, g1(std::max<size_t>(cnt, 1), 256, B), dim3(256), 0, C->d_blind_rows, (u32)c.blind_rows.size(), C->d_blind_zrows,
               (u32)c.blind_zrows.size(), C->cur->d_wires, ws, (u32)n, C->zk_key, proof_base);
    }
    if (!c.luts.empty()) {
        LutRowsArgs a{};
        a.lut_pairs = C->d_lut_pairs;
        a.lut_offsets = C->d_lut_offsets;
        a.rows = C->d_lookup_rows;
        a.num_lookups = C->d_num_lookups;
        a.mult = C->cur->d_mult;
        a.total_lut_entries = C->total_lut_entries;
        a.wires = C->cur->d_wires;
        a.wires_batch_stride = ws;
        a.n = (u32)n;
        a.num_luts = (u32)c.luts.size();
        LAUNCH(C, "lut_rows", k_lut_rows, g1(std::max<size_t>(C->total_lut_entries, 256), 256, B), dim3(256), 0, a);
    }
    // 2. wires commitment
    if (intt_cols(C, C->cur->d_wires, C->cur->d_wcoef, act, ws, B, C->cur->d_wlde, wls)) return P2_ERR_HIP;
    if (lde_cols(C, C->cur->d_wcoef, ws, C->cur->d_wlde, wls, act, 0, B)) return P2_ERR_HIP;
    if (salt) LAUNCH(C, "fill_salt", k_fill_salt, g1((size_t)salt * N / 8, 256, B), dim3(256), 0, C->cur->d_wlde + (size_t)act * N, wls, N, C->zk_key, proof_base, (u64)ZK_SALT + 1);
    if (merkle_build(C, C->cur->d_wlde, c.cfg.num_wires + salt, act + salt, N, wls, C->cur->wtree, B)) return P2_ERR_HIP;
    // 3. betas, gammas, deltas
    if (challenger(C, 0, C->cur->wtree.dig + cap_off(C->cur->wtree, cap_h), C->cur->wtree.stride(), cap_words, nlp ? 1 : 0, 0, B)) return P2_ERR_HIP;
    // 4. partial products and Z
    HIPCHECK(hipMemsetAsync(C->cur->d_zs, 0, (size_t)B * zs_s * 8, st));
    LAUNCH(C, "perm_chunks", k_perm_chunks, g1(n, 256, B), dim3(256), 0, C->cur->d_wires, ws, C->d_sigmas, C->d_k_is, C->d_subgroup, C->cur->d_chal,
           C->cur->d_permq, (size_t)NC * (npp + 1) * n, (u32)n, R, c.cfg.quotient_degree_factor, npp + 1, NC);
    {
        // columns longer than 2^14 rows in segments of 2^14 (at most PERM_MAX_SEGS), a workgroup per segment
        const u32 segs = (u32)std::min<size_t>(std::max<size_t>(n >> 14, 1), PERM_MAX_SEGS);
        if (segs > 1)
            LAUNCH(C, "perm_scan", k_perm_seg_products, dim3(NC, B, segs), dim3(1024), 0, C->cur->d_permq, (size_t)NC * (npp + 1) * n, C->cur->d_perm_seg, (u32)n, npp + 1);
        LAUNCH(C, "perm_scan", k_perm_scan, dim3(NC, B, segs), dim3(1024), 0, C->cur->d_permq, (size_t)NC * (npp + 1) * n, C->cur->d_zs, zs_s, (u32)n, npp + 1, NC,
               segs > 1 ? C->cur->d_perm_seg : nullptr);
    }
    // 5. lookup polynomials
    if (nlp) {
        LookupArgs a{};
        a.wires = C->cur->d_wires;
        a.wires_batch_stride = ws;
        a.chal = C->cur->d_chal;
        a.zs = C->cur->d_zs;
        a.zs_batch_stride = zs_s;
        a.tmp = C->cur->d_lktmp;
        a.tmp_batch_stride = (size_t)NC * (nsldc + 1) * n;
        a.rows = C->d_lookup_rows;
        a.n = (u32)n;
        a.num_luts = (u32)c.luts.size();
        a.num_sldc = nsldc;
        a.lut_deg = c.lut_degree();
        a.lu_deg = c.cfg.quotient_degree_factor - 1;
        a.num_challenges = NC;
        a.zs_lookup_col0 = c.num_zs_pp();
        LAUNCH(C, "lookup_terms", k_lookup_terms, g1(n, 256, B, NC * (nsldc + 1)), dim3(256), 0, a);
        LAUNCH(C, "lookup_scan", k_lookup_scan, dim3((u32)c.luts.size(), B, NC), dim3(1024), 0, a);
    }
    // 6. zs commitment, alphas
    if (intt_cols(C, C->cur->d_zs, C->cur->d_zcoef, zc, zs_s, B, C->cur->d_zlde, zl_s)) return P2_ERR_HIP;
    if (lde_cols(C, C->cur->d_zcoef, zs_s, C->cur->d_zlde, zl_s, zc, 0, B)) return P2_ERR_HIP;
    if (salt) LAUNCH(C, "fill_salt", k_fill_salt, g1((size_t)salt * N / 8, 256, B), dim3(256), 0, C->cur->d_zlde + (size_t)zc * N, zl_s, N, C->zk_key, proof_base, (u64)ZK_SALT + 2);
    if (merkle_build(C, C->cur->d_zlde, zc + salt, zc + salt, N, zl_s, C->cur->ztree, B)) return P2_ERR_HIP;
    if (challenger(C, 1, C->cur->ztree.dig + cap_off(C->cur->ztree, cap_h), C->cur->ztree.stride(), cap_words, 0, 0, B)) return P2_ERR_HIP;
    // 7. quotient
    {
        QuotientArgs a{};
        a.pre_lde = C->d_pre_lde;
        a.wires_lde = C->cur->d_wlde;
        a.zs_lde = C->cur->d_zlde;
        a.wires_batch_stride = wls;
        a.zs_batch_stride = zl_s;
        a.chal = C->cur->d_chal;
        a.xs = C->d_xs;
        a.l0 = C->d_l0;
        a.zh_inv = C->d_zh_inv;
        a.k_is = C->d_k_is;
        a.out = C->cur->d_qvals;
        a.out_batch_stride = (size_t)NC * N;
        a.n = (u32)n;
        a.logn = C->logn;
        a.rate_bits = c.cfg.rate_bits;
        a.R = R;
        a.ncc = ncc;
        a.nsel = c.num_selectors();
        a.nls = c.num_lookup_selectors;
        a.NC = NC;
        a.npp = npp;
        a.qdf = c.cfg.quotient_degree_factor;
        a.num_luts = (u32)c.luts.size();
        a.nsldc = nsldc;
        a.lut_deg = nlp ? c.lut_degree() : 0;
        a.nlp = nlp;
        a.num_gates = (u32)c.gates.size();
        a.num_gate_constraints = c.num_gate_constraints;
        for (u32 g = 0; g < c.gates.size(); g++) {
            a.gate_kind[g] = c.gates[g];
            a.gate_sel[g] = c.selector_index[g];
            a.group_lo[g] = c.groups[c.selector_index[g]].first;
            a.group_hi[g] = c.groups[c.selector_index[g]].second;
        }
        for (u32 l = 0; l < c.luts.size(); l++) a.lut_last_row[l] = c.lookup_rows[l].last_lut;
        a.zs_values = C->cur->d_zs;
        a.zs_values_batch_stride = zs_s;
        a.apow = C->cur->d_apow;
        {
            u32 nlk = nlp ? 4 + (u32)c.luts.size() + 2 * nsldc : 0;
            u32 nterms = NC + NC * (npp + 1) + NC * nlk + c.num_gate_constraints;
            if (nterms > APOW_STRIDE) return set_error("internal: too many vanishing terms for the alpha-power table"), P2_ERR_INVALID;
            LAUNCH(C, "alpha_pows", k_alpha_pows, g1(2 * B, 64), dim3(64), 0, C->cur->d_chal, C->cur->d_apow, B, nterms);
        }
        if (c.poseidon_rows.empty())
            if (C->opt_quotient_two_walks)
                LAUNCH(C, "quotient", (k_quotient<false, false>), g1(N, 256, B), dim3(256), 0, a);
            else
                LAUNCH(C, "quotient", (k_quotient<false, true>), g1(N, 256, B), dim3(256), 0, a);  // the wire columns read once
        else
            LAUNCH(C, "quotient", k_quotient<true>, g1(N, 256, B), dim3(256), 0, a);
        // coset-wise inverse transform: residues r_j, then the 8-point cross-coset DFT
        if (C->logn > LDS_NTT_MAX_BITS) {
            const size_t qs = (size_t)NC * N;
            u32 ident[8] = {0, 1, 2, 3, 4, 5, 6, 7};
            LAUNCH(C, "bitrev_copy", k_bitrev_copy, g1(n, 256, B, NC * 8), dim3(256), 0, C->cur->d_qvals, N, qs, C->cur->d_qres, N, qs, (int)C->logn, 8u,
                   (const u64*)nullptr, 0u);
            if (ntt_big(C, "quotient_intt", C->cur->d_qres, C->cur->d_qvals, C->d_tw_inv_full, nullptr, C->logn, NC, 8, ident, 1, N, N, qs, qs, 1, B)) return P2_ERR_HIP;
            LAUNCH(C, "bitrev_copy", k_bitrev_copy, g1(n, 256, B, NC * 8), dim3(256), 0, C->cur->d_qvals, N, qs, C->cur->d_qres, N, qs, (int)C->logn, 8u,
                   (const u64*)C->d_shift_inv_pows, 1u);
        } else {
        NttArgs t{};
        t.in = C->cur->d_qvals;
        t.out = C->cur->d_qres;
        t.tw = C->d_tw_inv;
        t.post = C->d_shift_inv_pows;
        t.post_scalar = 1;
        t.in_col_stride = t.out_col_stride = N;
        t.in_batch_stride = t.out_batch_stride = (size_t)NC * N;
        t.logn = (int)C->logn;
        t.cosets = 8;
        t.bitrev_in = 1;
        t.bitrev_out = 1;
        t.in_coset_blocks = 1;
        // input block rev3(j) holds coset j; residue r_j is written to the same block
        for (u32 j = 0; j < 8; j++) t.block_of_coset[j] = gl::bitrev(j, 3);
        if (run_ntt(C, "quotient_intt", t, NC, B)) return P2_ERR_HIP;
        }
        LAUNCH(C, "quotient_chunks", k_quotient_chunks_rev, g1(n, 256, B, NC), dim3(256), 0, C->cur->d_qres, C->cur->d_qcoef, (u32)n, (size_t)NC * N, (size_t)qc * n,
               C->d_w8inv, C->d_qscale);
    }
    if (lde_cols(C, C->cur->d_qcoef, (size_t)qc * n, C->cur->d_qlde, ql_s, qc, 0, B)) return P2_ERR_HIP;
    if (salt) LAUNCH(C, "fill_salt", k_fill_salt, g1((size_t)salt * N / 8, 256, B), dim3(256), 0, C->cur->d_qlde + (size_t)qc * N, ql_s, N, C->zk_key, proof_base, (u64)ZK_SALT + 3);
    if (merkle_build(C, C->cur->d_qlde, qc + salt, qc + salt, N, ql_s, C->cur->qtree, B)) return P2_ERR_HIP;
    if (challenger(C, 2, C->cur->qtree.dig + cap_off(C->cur->qtree, cap_h), C->cur->qtree.stride(), cap_words, c.degree_bits, 0, B)) return P2_ERR_HIP;
    // 8. openings
    LAUNCH(C, "zeta_pows", k_zeta_pows, g1(n, 256, B, 4), dim3(256), 0, C->cur->d_chal, C->cur->d_pows, (size_t)8 * n, (u32)n, gl::root_of_unity((int)C->logn));
    HIPCHECK(hipMemsetAsync(C->cur->d_ev, 0, (size_t)B * 2 * C->ev_count * 8, st));
    {
        const size_t evs = 2 * (size_t)C->ev_count;
        u64* ev = C->cur->d_ev;
        LAUNCH(C, "eval_polys", k_eval_polys_refs, dim3(C->n_evalrefs, B), dim3(256), 0, C->cur->d_evalrefs, C->cur->d_pows, (size_t)8 * n, (u32)n, ev, evs);
        LAUNCH(C, "gather_ext", k_gather_ext, g1(C->n_obs, 256, B), dim3(256), 0, C->cur->d_ev, evs, C->d_map_obs, C->n_obs, C->cur->d_obs, (size_t)2 * C->n_obs);
    }
    if (challenger(C, 3, C->cur->d_obs, (size_t)2 * C->n_obs, 2 * C->n_obs, 0, 0, B)) return P2_ERR_HIP;
    // 9. FRI: compose, divide, commit phase
    LAUNCH(C, "fri_compose", k_fri_compose, g1(n, 256, B), dim3(256), 0, C->cur->d_polyrefs, C->n_b0, C->n_b1, C->cur->d_chal, (u32)n, C->cur->d_comp, (size_t)4 * n);
    {
        const u32 segs = (u32)std::min<size_t>(std::max<size_t>(n >> 14, 1), FRI_MAX_SEGS);  // as in the permutation scan
        if (segs > 1)
            LAUNCH(C, "fri_divide", k_fri_seg_sums, dim3(B, segs), dim3(1024), 0, C->cur->d_comp, (size_t)4 * n, C->cur->d_pows, (size_t)8 * n, (u32)n, C->cur->d_fri_seg);
        LAUNCH(C, "fri_divide", k_fri_divide, dim3(B, segs), dim3(1024), 0, C->cur->d_comp, (size_t)4 * n, C->cur->d_pows, (size_t)8 * n, C->cur->d_chal, (u32)n, C->n_b1,
               C->cur->d_fri_coef[0], (size_t)2 * n, segs > 1 ? C->cur->d_fri_seg : nullptr);
    }
    {
        u32 logn_r = C->logn;
        for (u32 r = 0; r < C->arities.size(); r++) {
            size_t n_r = (size_t)1 << logn_r, len = 8 * n_r;
            u32 arity = 1u << C->arities[r];
            if (lde_cols(C, C->cur->d_fri_coef[r], 2 * n_r, C->cur->d_fri_vals[r], 2 * len, 2, r, B)) return P2_ERR_HIP;
            Tree& t = C->cur->fri_tree[r];
            size_t leaves = len / arity;
            LAUNCH(C, "hash_fri_leaves", k_hash_fri_leaves, g1(leaves, 256, B), dim3(256), 0, C->cur->d_fri_vals[r], len, 2 * len, (int)arity, t.dig, t.stride());
            if (merkle_levels(C, t, B)) return P2_ERR_HIP;
            if (challenger(C, 4, t.dig + cap_off(t, cap_h), t.stride(), cap_words, r, 0, B)) return P2_ERR_HIP;
            size_t n_next = n_r >> C->arities[r];
            LAUNCH(C, "fri_fold", k_fri_fold, g1(n_next, 256, B), dim3(256), 0, C->cur->d_fri_coef[r], n_r, 2 * n_r, C->cur->d_fri_coef[r + 1], n_next, 2 * n_next, C->cur->d_chal, r,
                   arity);
            logn_r -= C->arities[r];
        }
        // final polynomial (interleave components for observation)
        size_t fl = (size_t)1 << logn_r;
        u32 R_ = (u32)C->arities.size();
        LAUNCH(C, "interleave", k_interleave_ext, g1(fl, 256, B), dim3(256), 0, C->cur->d_fri_coef[R_], fl, 2 * fl, C->cur->d_obs, (size_t)2 * C->n_obs);
        if (challenger(C, 5, C->cur->d_obs, (size_t)2 * C->n_obs, (u32)(2 * fl), 0, 0, B)) return P2_ERR_HIP;
        // proof of work
        HIPCHECK(hipMemsetAsync(C->cur->d_pow_best, 0xFF, (size_t)B * 8, st));
        {
            u32 block0 = 0;
            for (int ph = 0; ph < 3; ph++) {
                u32* list = ph ? C->cur->d_pow_list : nullptr;
                if (ph) LAUNCH(C, "pow", k_pow_compact, dim3(1), dim3(256), 0, C->cur->d_pow_best, B, C->cur->d_pow_list, C->cur->d_pow_list + C->chunk);
                const u32 slots = ph ? std::min<u32>(B, POW_PHASE_SLOTS[ph]) : B;
                LAUNCH(C, "pow", k_pow, dim3(slots, POW_PHASE_BLOCKS[ph]), dim3(256), 0, C->cur->d_chal_state, C->cur->d_chal, (int)c.cfg.pow_bits, C->cur->d_pow_best,
                       block0, (const u32*)list, (const u32*)(list ? list + C->chunk : nullptr));
                block0 += POW_PHASE_BLOCKS[ph];
            }
        }
        LAUNCH(C, "pow_finish", k_pow_finish, g1(B, 64), dim3(64), 0, C->cur->d_chal, C->cur->d_pow_best, B, C->cur->d_status);
        if (challenger(C, 6, C->cur->d_obs, 0, 0, c.cfg.num_query_rounds, (u64)N, B)) return P2_ERR_HIP;
        // 10. proof assembly
        size_t off = 0;
        const size_t pb = C->pbytes;
        ProofSegs segs{};
        u32 nseg = 0;
        segs.proofs = d_proofs;
        segs.proof_bytes = pb;
        for (Tree* t : {&C->cur->wtree, &C->cur->ztree, &C->cur->qtree}) {
            segs.s[nseg++] = ProofSeg{t->dig + cap_off(*t, cap_h), nullptr, t->stride(), 0, off, cap_words, 0};
            off += 8 * (size_t)cap_words;
        }
        segs.s[nseg++] = ProofSeg{C->cur->d_ev, C->d_map_ser, 2 * (size_t)C->ev_count, 0, off, C->n_ser, 2};
        off += 16 * (size_t)C->n_ser;
        if (R_ + 6 > 12) return set_error("internal: more FRI rounds than proof segments"), P2_ERR_INVALID;
        for (u32 r = 0; r < R_; r++) {
            Tree& t = C->cur->fri_tree[r];
            segs.s[nseg++] = ProofSeg{t.dig + cap_off(t, cap_h), nullptr, t.stride(), 0, off, cap_words, 0};
            off += 8 * (size_t)cap_words;
        }
        QueryArgs q{};
        const u64* ldes[4] = {C->d_pre_lde, C->cur->d_wlde, C->cur->d_zlde, C->cur->d_qlde};
        const size_t lstr[4] = {0, wls, zl_s, ql_s};
        const Tree* trees[4] = {&C->pre_tree, &C->cur->wtree, &C->cur->ztree, &C->cur->qtree};
        const u32 colsv[4] = {np, c.cfg.num_wires + salt, zc + salt, qc + salt};  // blinded leaves end with the salt
        const u32 actv[4] = {np, act + salt, zc + salt, qc + salt};
        size_t qbytes = 0;
        for (int o = 0; o < 4; o++) {
            q.oracles[o].lde = ldes[o];
            q.oracles[o].lde_batch_stride = lstr[o];
            q.oracles[o].digests = trees[o]->dig;
            q.oracles[o].dig_batch_stride = o == 0 ? 0 : trees[o]->stride();
            q.oracles[o].cols = colsv[o];
            q.oracles[o].active_cols = actv[o];
            qbytes += 8 * (size_t)colsv[o] + 1 + 32 * (size_t)(C->lde_bits - cap_h);
        }
        q.lde_bits = C->lde_bits;
        q.cap_height = cap_h;
        q.num_queries = c.cfg.num_query_rounds;
        q.num_rounds = R_;
        u32 lb = C->lde_bits;
        for (u32 r = 0; r < R_; r++) {
            q.arity_bits[r] = C->arities[r];
            q.fri_vals[r] = C->cur->d_fri_vals[r];
            q.fri_vals_batch_stride[r] = (size_t)2 << lb;
            q.fri_bits[r] = lb;
            q.fri_digests[r] = C->cur->fri_tree[r].dig;
            q.fri_dig_batch_stride[r] = C->cur->fri_tree[r].stride();
            qbytes += 16 * ((size_t)1 << C->arities[r]) + 1 + 32 * (size_t)(lb - C->arities[r] - cap_h);
            lb -= C->arities[r];
        }
        q.chal = C->cur->d_chal;
        q.proofs = d_proofs;
        q.proof_bytes = pb;
        q.queries_off = off;
        q.query_bytes = qbytes;
        LAUNCH(C, "write_queries", k_write_queries, dim3(c.cfg.num_query_rounds, B), dim3(256), 0, q);
        off += qbytes * c.cfg.num_query_rounds;
        segs.s[nseg++] = ProofSeg{C->cur->d_fri_coef[R_], nullptr, 2 * fl, fl, off, (u32)fl, 1};
        off += 16 * fl;
        segs.s[nseg++] = ProofSeg{C->cur->d_chal + CH_POW, nullptr, (size_t)CH_WORDS, 0, off, 1u, 0};
        off += 8;
        if (off != pb) return set_error("internal: proof layout size mismatch"), P2_ERR_INVALID;
        LAUNCH(C, "proof_segments", k_proof_segments, dim3(2, B, nseg), dim3(256), 0, segs);
    }
    LAUNCH(C, "finish", k_finish, g1(C->pbytes, 256, B), dim3(256), 0, C->cur->d_status, d_status_out, d_proofs, C->pbytes, B);
    return 0;
}

static void collect_timing(p2_circuit* C) {
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> all;
    all.swap(C->setup_ws.pending);
    for (Workspace* W : C->ws) {
        all.insert(all.end(), W->pending.begin(), W->pending.end());
        W->pending.clear();
    }
    for (auto& pe : all) {
        float ms = 0;
        (void)hipEventSynchronize(pe.second.second);
        (void)hipEventElapsedTime(&ms, pe.second.first, pe.second.second);
        auto& t = C->times[pe.first];
        t.first += ms;
        t.second++;
        (void)hipEventDestroy(pe.second.first);
        (void)hipEventDestroy(pe.second.second);
    }
}

// Host-path staging: device buffers, pinned host mirrors and a stream, kept across p2_prove_batch calls (they only grow).
// A caller leases one set for the duration of its call; concurrent callers on one handle each get their own.
struct Staging {
    hipStream_t stream = nullptr;
    u64 *d_vals = nullptr, *h_vals = nullptr;
    uint8_t *d_proofs = nullptr, *h_proofs = nullptr;
    int *d_stat = nullptr, *h_stat = nullptr;
    size_t cap_vals = 0, cap_proofs = 0, cap_stat = 0;
    void release() {
        if (d_vals) (void)hipFree(d_vals);
        if (d_proofs) (void)hipFree(d_proofs);
        if (d_stat) (void)hipFree(d_stat);
        if (h_vals) (void)hipHostFree(h_vals);
        if (h_proofs) (void)hipHostFree(h_proofs);
        if (h_stat) (void)hipHostFree(h_stat);
        if (stream) (void)hipStreamDestroy(stream);
        *this = Staging();
    }
    template <class T>
    static bool grow(T** d, T** h, size_t* cap, size_t bytes) {
        if (*cap >= bytes) return true;
        if (*d) (void)hipFree(*d);
        if (*h) (void)hipHostFree(*h);
        *d = nullptr, *h = nullptr, *cap = 0;
        size_t want = bytes + bytes / 4;
        if (hipMalloc((void**)d, want) != hipSuccess || hipHostMalloc((void**)h, want, hipHostMallocDefault) != hipSuccess) return false;
        *cap = want;
        return true;
    }
};
struct StagingLease {
    p2_circuit* C;
    Staging* S = nullptr;
    explicit StagingLease(p2_circuit* c) : C(c) {}
    Staging* get(size_t vals_bytes, size_t proofs_bytes, size_t batch) {
        {
            std::lock_guard<std::mutex> lock(C->staging_mu);
            if (!C->staging_free.empty()) {
                S = C->staging_free.back();
                C->staging_free.pop_back();
            }
        }
        if (!S) S = new Staging();
        if ((!S->stream && hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking) != hipSuccess) ||
            !Staging::grow(&S->d_vals, &S->h_vals, &S->cap_vals, vals_bytes) || !Staging::grow(&S->d_proofs, &S->h_proofs, &S->cap_proofs, proofs_bytes) ||
            !Staging::grow(&S->d_stat, &S->h_stat, &S->cap_stat, batch * sizeof(int))) {
            set_error("staging buffers for p2_prove_batch could not be allocated");
            S->release();
            delete S;
            S = nullptr;
        }
        return S;
    }
    ~StagingLease() {
        if (!S) return;
        // every path out of p2_prove_batch, the error returns after an async copy included: nothing may still be reading or
        // writing these buffers when the next caller leases them (a no-op on the success path, which has synchronised already)
        if (S->stream) (void)hipStreamSynchronize(S->stream);
        std::lock_guard<std::mutex> lock(C->staging_mu);
        C->staging_free.push_back(S);
    }
};

// Every entry point that allocates (std::vector, std::map, std::thread) runs behind this guard: an exception becomes
// P2_ERR_* + p2_last_error(), never std::terminate -- also on the worker threads of p2_prove_batch_multi.
template <class F>
static int guarded_rc(F&& f) {
    try {
        return f();
    } catch (std::bad_alloc&) {
        return set_error("out of host memory"), P2_ERR_HIP;
    } catch (std::exception& e) {
        return set_error(e.what()), P2_ERR_INVALID;
    } catch (...) {
        return set_error("unknown exception"), P2_ERR_INVALID;
    }
}
extern "C" {

int p2_gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

p2_circuit* p2_circuit_load(const uint8_t* blob, size_t len, int device) {
    p2_circuit* C = nullptr;
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            set_error("no HIP device available: the prover has no CPU fallback");
            return nullptr;
        }
        if (device < 0 || device >= ndev) {
            set_error("device index out of range");
            return nullptr;
        }
        C = new p2_circuit();
        C->c = deserialize(blob, len);
        const Circuit& c = C->c;
        if (c.cfg.rate_bits != 3 || c.cfg.num_challenges != 2 || c.cfg.quotient_degree_factor != 8 || c.cfg.num_routed_wires != 80 || c.cfg.arity_bits != 4 ||
            c.cfg.num_query_rounds > 64 || c.gates.size() > p2::MAX_GATE_TYPES || c.luts.size() > p2::MAX_LUTS)
            throw std::runtime_error("only CircuitConfig::standard_recursion_config() is supported");
        if (c.degree_bits > 22) throw std::runtime_error("degree_bits > 22 is not supported");
        C->device = device;
        C->logn = c.degree_bits;
        C->n = c.n();
        C->lde_bits = c.degree_bits + c.cfg.rate_bits;
        C->N = C->n << c.cfg.rate_bits;
        C->arities = c.reduction_arity_bits();
        // routed-only gates leave wires 80..134 identically zero (never materialised); PoseidonGate rows use all 135
        C->active_wires = (c.poseidon_rows.empty() && !c.cfg.zero_knowledge) ? c.cfg.num_routed_wires : c.cfg.num_wires;
        for (int i = 0; i < 4; i++) C->zk_key.k[i] = os_random_field();
        // environment defaults for the options, read once here (never per call)
        if (const char* e = getenv("P2AES_CHUNK")) C->opt_chunk = (size_t)std::max(1, atoi(e));
        if (const char* e = getenv("P2AES_STREAMS")) C->opt_streams = (size_t)std::min(8, std::max(1, atoi(e)));
        C->opt_pass1_radix2 = getenv("P2AES_PASS1_RADIX2") != nullptr;
        C->opt_pass1_noswizzle = getenv("P2AES_PASS1_NOSWIZZLE") != nullptr;
        C->opt_quotient_two_walks = getenv("P2AES_QUOTIENT_TWO_WALKS") != nullptr;
        if (const char* e = getenv("P2AES_PASS1_WAVES")) C->opt_pass1_waves = atoi(e) == 2 ? 2 : 4;
        if (const char* e = getenv("P2AES_MERKLE_TOP")) C->opt_merkle_top = atoi(e) != 0;
        if (const char* e = getenv("P2AES_WITNESS_FUSE")) C->opt_witness_fuse = (u32)std::min(1024, std::max(1, atoi(e)));
        C->opt_debug_timing = getenv("P2AES_DEBUG_TIMING") != nullptr;
        if (const char* e = getenv("P2AES_TEST_FAIL_ALLOC_AFTER")) C->fail_alloc_after = atol(e);
        C->pbytes = proof_bytes(c);
        if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
        if (hipStreamCreate(&C->stream) != hipSuccess) throw std::runtime_error("hipStreamCreate failed");
        if (hipEventCreateWithFlags(&C->ev_witness, hipEventDisableTiming) != hipSuccess) throw std::runtime_error("hipEventCreate failed");
        if (hipFuncSetAttribute((const void*)k_ntt_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_ntt_r16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r16_lds_bytes(14)) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_ntt_r16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r16_lds_bytes(13)) != hipSuccess)
            throw std::runtime_error("cannot raise the dynamic LDS limit for the NTT kernels");
        if (hipFuncSetAttribute((const void*)k_ntt_pass1, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_ntt_pass1_r16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_ntt_pass1_r16<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
            throw std::runtime_error("cannot set the dynamic LDS limit of the pass-1 NTT");
        // opening maps
        const u32 np = c.num_preprocessed(), W = c.cfg.num_wires, zc = c.num_zs_cols(), qc = c.num_quotient_cols(), NC = c.cfg.num_challenges;
        const u32 ncc = c.num_constants_cols(), nzpp = c.num_zs_pp();
        const u32 E_PRE = 0, E_W = np, E_Z = np + W, E_ZN = E_Z + zc, E_Q = E_ZN + zc;
        C->ev_count = E_Q + qc;
        std::vector<u32> obs, ser;
        auto range = [](std::vector<u32>& v, u32 base, u32 a, u32 b) {
            for (u32 i = a; i < b; i++) v.push_back(base + i);
        };
        range(obs, E_PRE, 0, np);
        range(obs, E_W, 0, W);
        range(obs, E_Z, 0, nzpp);
        range(obs, E_Q, 0, qc);
        range(obs, E_Z, nzpp, zc);
        range(obs, E_ZN, 0, NC);
        range(obs, E_ZN, nzpp, zc);
        range(ser, E_PRE, 0, ncc);
        range(ser, E_PRE, ncc, np);
        range(ser, E_W, 0, W);
        range(ser, E_Z, 0, NC);
        range(ser, E_ZN, 0, NC);
        range(ser, E_Z, nzpp, zc);   // write_opening_set puts lookup_zs / lookup_zs_next between plonk_zs_next and the
        range(ser, E_ZN, nzpp, zc);  // partial products (the OpeningSet struct itself lists them last)
        range(ser, E_Z, NC, nzpp);
        range(ser, E_Q, 0, qc);
        C->n_obs = (u32)obs.size();
        C->n_ser = (u32)ser.size();
        size_t fl = C->n;
        for (u32 a : C->arities) fl >>= a;
        if (fl > C->n_obs) throw std::runtime_error("final polynomial larger than the observation buffer");
        if (upload(C, &C->d_map_obs, obs.data(), obs.size()) || upload(C, &C->d_map_ser, ser.data(), ser.size())) throw std::runtime_error(g_last_error);
        if (circuit_setup(C)) throw std::runtime_error(g_last_error);
        return C;
    } catch (std::exception& e) {
        set_error(e.what());
        if (C) p2_circuit_free(C);
        return nullptr;
    }
}

void p2_circuit_free(p2_circuit* C) {
    if (!C) return;
    (void)hipSetDevice(C->device);
    (void)hipDeviceSynchronize();
    release_workspaces(C);
    for (Staging* S : C->staging_free) {
        S->release();
        delete S;
    }
    for (void* p : C->allocs) (void)hipFree(p);
    if (C->stream) (void)hipStreamDestroy(C->stream);
    if (C->ev_witness) (void)hipEventDestroy(C->ev_witness);
    delete C;
}

int p2_circuit_verifier_data(const p2_circuit* C, uint64_t* out, size_t cap, size_t* n_written) {
    if (cap < C->verifier_data.size()) return set_error("buffer too small"), P2_ERR_INVALID;
    memcpy(out, C->verifier_data.data(), C->verifier_data.size() * 8);
    *n_written = C->verifier_data.size();
    return P2_OK;
}
size_t p2_circuit_proof_bytes(const p2_circuit* C) { return C->pbytes; }
size_t p2_circuit_chunk_proofs(p2_circuit* C) {
    std::lock_guard<std::mutex> lock(C->mu);
    return C->chunk;
}
int p2_circuit_set_zk_key(p2_circuit* C, const uint64_t key[4]) {
    // TEST ONLY, and refused unless the process opted in: a fixed key is not secret, and a (key, proof index) pair that is used
    // for two different witnesses breaks zero-knowledge.
    const char* allow = getenv("P2AES_ALLOW_FIXED_ZK_KEY");
    if (!allow || strcmp(allow, "1") != 0)
        return set_error("p2_circuit_set_zk_key is a test hook: set P2AES_ALLOW_FIXED_ZK_KEY=1 to fix the blinding key (never in production)"), P2_ERR_INVALID;
    std::lock_guard<std::mutex> lock(C->mu);
    bool same = true;
    for (int i = 0; i < 4; i++) same = same && C->zk_key.k[i] == key[i] % gl::P;
    if (same) return P2_OK;  // the proof counter keeps running: setting the key a handle already holds never replays an index
    for (int i = 0; i < 4; i++) C->zk_key.k[i] = key[i] % gl::P;
    C->zk_counter = 0;
    return P2_OK;
}
int p2_circuit_set_zk_seed(p2_circuit* C, uint64_t seed) {
    const uint64_t key[4] = {seed, 0, 0, 0};
    return p2_circuit_set_zk_key(C, key);
}

static int setup_polyrefs(p2_circuit* C) {
    const Circuit& c = C->c;
    const u32 np = c.num_preprocessed(), W = c.cfg.num_wires, zc = c.num_zs_cols(), qc = c.num_quotient_cols(), NC = c.cfg.num_challenges, nzpp = c.num_zs_pp();
    const size_t n = C->n;
    std::vector<PolyRef> v;
    for (u32 i = 0; i < np; i++) v.push_back({C->d_pre_coeffs, 0, i, 0});
    for (u32 i = 0; i < W; i++) v.push_back({i < C->active_wires ? C->cur->d_wcoef : nullptr, (size_t)C->active_wires * n, i, 0});
    for (u32 i = 0; i < nzpp; i++) v.push_back({C->cur->d_zcoef, (size_t)zc * n, i, 0});
    for (u32 i = 0; i < qc; i++) v.push_back({C->cur->d_qcoef, (size_t)qc * n, i, 0});
    for (u32 i = nzpp; i < zc; i++) v.push_back({C->cur->d_zcoef, (size_t)zc * n, i, 0});
    C->n_b0 = (u32)v.size();
    for (u32 i = 0; i < NC; i++) v.push_back({C->cur->d_zcoef, (size_t)zc * n, i, 0});
    for (u32 i = nzpp; i < zc; i++) v.push_back({C->cur->d_zcoef, (size_t)zc * n, i, 0});
    C->n_b1 = (u32)v.size() - C->n_b0;
    // the opening set: every polynomial at zeta, the Z columns at g zeta as well (slots: preprocessed | wires | Z(zeta) | Z(g zeta) | quotient)
    std::vector<EvalRef> e;
    for (u32 i = 0; i < np; i++) e.push_back({C->d_pre_coeffs, 0, i, 0, i, 0});
    for (u32 i = 0; i < C->active_wires; i++) e.push_back({C->cur->d_wcoef, (size_t)C->active_wires * n, i, 0, np + i, 0});
    for (u32 i = 0; i < zc; i++) e.push_back({C->cur->d_zcoef, (size_t)zc * n, i, 0, np + W + i, 0});
    for (u32 i = 0; i < zc; i++) e.push_back({C->cur->d_zcoef, (size_t)zc * n, i, 1, np + W + zc + i, 0});
    for (u32 i = 0; i < qc; i++) e.push_back({C->cur->d_qcoef, (size_t)qc * n, i, 0, np + W + 2 * zc + i, 0});
    C->n_evalrefs = (u32)e.size();
    if (upload(C, &C->cur->d_evalrefs, e.data(), e.size())) return P2_ERR_HIP;
    return upload(C, &C->cur->d_polyrefs, v.data(), v.size());
}

static int prove_batch_device_impl(p2_circuit* C, size_t batch, const p2_target* targets, size_t n_targets, const uint64_t* d_values, uint8_t* d_proofs, int* d_status,
                                   void* stream);
int p2_prove_batch_device(p2_circuit* C, size_t batch, const p2_target* targets, size_t n_targets, const uint64_t* d_values, uint8_t* d_proofs, int* d_status,
                          void* stream) {
    return guarded_rc([&] { return prove_batch_device_impl(C, batch, targets, n_targets, d_values, d_proofs, d_status, stream); });
}
static int prove_batch_device_impl(p2_circuit* C, size_t batch, const p2_target* targets, size_t n_targets, const uint64_t* d_values, uint8_t* d_proofs, int* d_status,
                                   void* stream) {
    std::lock_guard<std::mutex> lock(C->mu);
    hipStream_t caller = (hipStream_t)stream;
    HIPCHECK(hipSetDevice(C->device));
    std::vector<u32> slots(n_targets);
    for (size_t i = 0; i < n_targets; i++) {
        u64 t = targets[i];
        int32_t slot = -1;
        if (t >> 63) {  // Target::Wire(row, column)
            u64 row = (t & ~(1ull << 63)) >> 8, col = t & 0xFF;
            if (row < C->n && col < C->c.cfg.num_routed_wires) slot = C->c.wire_slot[col * C->n + row];
        } else if (t < C->c.vt_slot.size()) {
            slot = C->c.vt_slot[t];
        }
        if (slot < 0) return set_error("input target is not a target of this circuit"), P2_ERR_INVALID;
        slots[i] = (u32)slot;
    }
    // Chunk size / stream count: options "chunk" (default 128 proofs per chunk) and "streams" (default 2).  A batch smaller
    // than chunk x streams is split evenly over the streams, so that the serial stages of one chunk (witness levels,
    // the Fiat-Shamir chain, proof-of-work) overlap the wide kernels of the other.  A later, larger batch regrows the
    // workspaces (alloc_workspace); a smaller one runs in the existing ones.
    size_t want_chunk = C->opt_chunk, want_streams = C->opt_streams;
    {
        // cap the chunk so that all workspaces fit in ~80% of the HBM that is free (plus what the workspaces hold now); chunks of
        // a batch are then made EQUAL (a 32-proof batch under a cap of 14 is 3 x 11, not 14 + 14 + 4)
        const Circuit& c = C->c;
        size_t n = C->n, N = C->N, zc = c.num_zs_cols(), qc = c.num_quotient_cols(), act = C->active_wires, NC = c.cfg.num_challenges;
        size_t words = c.num_slots + (act + zc) * 2 * n + qc * n + (act + zc + qc) * N + NC * (c.num_partial_products() + c.num_sldc_polys() + 2) * n +
                       2 * NC * N + 3 * 8 * N + 16 * n + 4 * N;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            free_b += 8 * words * C->chunk * C->ws.size();
            size_t fit = (size_t)(0.8 * (double)free_b) / (8 * words * want_streams);
            want_chunk = std::max<size_t>(1, std::min(want_chunk, fit));
        }
    }
    size_t nstreams = std::min(want_streams, std::max<size_t>(batch, 1));
    size_t chunk = std::min(want_chunk, (std::max<size_t>(batch, 1) + nstreams - 1) / nstreams);
    {
        const size_t nchunks = (std::max<size_t>(batch, 1) + chunk - 1) / chunk;
        chunk = (std::max<size_t>(batch, 1) + nchunks - 1) / nchunks;
    }
    if (C->chunk >= chunk && C->ws.size() >= nstreams) chunk = C->chunk, nstreams = C->ws.size();
    if (alloc_workspace(C, chunk, (u32)n_targets, nstreams)) return P2_ERR_HIP;
    // Ordering with the caller: the proving streams wait for everything already enqueued on the caller's stream (its
    // inputs) and the caller's stream then waits for the proofs.  NULL means the legacy default stream, which is
    // ordered the same way (an event recorded on stream 0) -- no device-wide synchronisation, so consecutive calls
    // pipeline into each other.
    struct EventGuard {
        hipEvent_t e = nullptr;
        ~EventGuard() {
            if (e) (void)hipEventDestroy(e);
        }
    } ev_guard;
    HIPCHECK(hipEventCreateWithFlags(&ev_guard.e, hipEventDisableTiming));
    hipEvent_t ev_in = ev_guard.e;
    HIPCHECK(hipEventRecord(ev_in, caller));
    for (Workspace* W : C->ws) HIPCHECK(hipStreamWaitEvent(W->stream, ev_in, 0));
    // the blinding counter advances before anything is enqueued: a batch that fails half-way must not leave its proof
    // indices to be used again under the same key
    const u64 proof_base0 = C->zk_counter;
    C->zk_counter += batch;
    size_t k = 0;
    for (size_t done = 0; done < batch; done += C->chunk, k++) {
        u32 B = (u32)std::min(C->chunk, batch - done);
        C->cur = C->ws[k % C->ws.size()];
        if (C->cur->h_input_slots != slots) {  // a new target list: wait for the workspace's earlier chunks, then upload
            HIPCHECK(hipStreamSynchronize(C->cur->stream));
            HIPCHECK(hipMemcpy(C->cur->d_input_slots, slots.data(), n_targets * 4, hipMemcpyHostToDevice));
            C->cur->h_input_slots = slots;
        }
        int rc = prove_chunk(C, B, (u32)n_targets, d_values + done * n_targets, d_proofs + done * C->pbytes, d_status + done, proof_base0 + done);
        C->cur = nullptr;
        if (rc) return rc;
    }
    for (Workspace* W : C->ws) {
        HIPCHECK(hipEventRecord(W->done, W->stream));
        HIPCHECK(hipStreamWaitEvent(caller, W->done, 0));
    }
    return P2_OK;
}

int p2_circuit_synchronize(p2_circuit* C) {
    std::lock_guard<std::mutex> lock(C->mu);
    HIPCHECK(hipSetDevice(C->device));
    HIPCHECK(hipStreamSynchronize(C->stream));
    for (Workspace* W : C->ws) HIPCHECK(hipStreamSynchronize(W->stream));
    if (C->timing_on) collect_timing(C);
    return P2_OK;
}

static int prove_batch_impl(p2_circuit* C, size_t batch, const p2_assignment* inputs, uint8_t* proofs, int* status);
int p2_prove_batch(p2_circuit* C, size_t batch, const p2_assignment* inputs, uint8_t* proofs, int* status) {
    return guarded_rc([&] { return prove_batch_impl(C, batch, inputs, proofs, status); });
}
static int prove_batch_impl(p2_circuit* C, size_t batch, const p2_assignment* inputs, uint8_t* proofs, int* status) {
    if (batch == 0) return P2_OK;
    HIPCHECK(hipSetDevice(C->device));
    // Fast path: every PartialWitness assigns the same target list in the same order.  Otherwise the batch is put on
    // the union of the targets, a witness that does not assign a target gets the "absent" marker (2^64-1, not a field
    // element), and a witness that assigns one target two different values fails on the host like set_target does.
    // Either way a value that is not a canonical field element fails that witness here, on the host.
    size_t nt = inputs[0].count;
    bool same = true;
    for (size_t i = 1; i < batch && same; i++) same = inputs[i].count == nt && memcmp(inputs[i].targets, inputs[0].targets, nt * 8) == 0;
    std::vector<u64> union_targets;
    std::map<u64, size_t> col;
    std::vector<int> host_status(batch, 0);
    const p2_target* targets = inputs[0].targets;
    if (!same) {
        for (size_t i = 0; i < batch; i++)
            for (size_t k = 0; k < inputs[i].count; k++)
                if (col.emplace(inputs[i].targets[k], union_targets.size()).second) union_targets.push_back(inputs[i].targets[k]);
        nt = union_targets.size();
        targets = union_targets.data();
    }
    const size_t nvals = batch * std::max<size_t>(nt, 1);
    StagingLease lease(C);
    Staging* S = lease.get(nvals * 8, batch * C->pbytes, batch);
    if (!S) return P2_ERR_HIP;
    const auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool dbg = C->opt_debug_timing;
    double t_a = now();
    u64* hv = S->h_vals;  // pinned: the values are laid out straight into the buffer the DMA engine reads
    if (same) {
        for (size_t i = 0; i < batch; i++)
            for (size_t k = 0; k < nt; k++) {
                u64 v = inputs[i].values[k];
                if (v >= gl::P) host_status[i] = P2_PROOF_WITNESS_CONFLICT, v = 0;
                hv[i * nt + k] = v;
            }
    } else {
        std::fill(hv, hv + nvals, ~0ull);
        for (size_t i = 0; i < batch; i++)
            for (size_t k = 0; k < inputs[i].count; k++) {
                u64& cell = hv[i * nt + col[inputs[i].targets[k]]];
                u64 v = inputs[i].values[k];
                if (v >= gl::P || (cell != ~0ull && cell != v)) host_status[i] = P2_PROOF_WITNESS_CONFLICT, v = 0;
                cell = v;
            }
    }
    double t_b = now();
    // one stream carries upload -> prove -> download; p2_prove_batch_device orders the proving streams with it
    HIPCHECK(hipMemcpyAsync(S->d_vals, hv, nvals * 8, hipMemcpyHostToDevice, S->stream));
    int rc = p2_prove_batch_device(C, batch, targets, nt, S->d_vals, S->d_proofs, S->d_stat, (void*)S->stream);
    if (rc != P2_OK) return rc;  // (the lease drains S->stream on every way out)
    HIPCHECK(hipMemcpyAsync(S->h_proofs, S->d_proofs, batch * C->pbytes, hipMemcpyDeviceToHost, S->stream));
    HIPCHECK(hipMemcpyAsync(S->h_stat, S->d_stat, batch * sizeof(int), hipMemcpyDeviceToHost, S->stream));
    double t_c = now();
    HIPCHECK(hipStreamSynchronize(S->stream));
    double t_d = now();
    memcpy(proofs, S->h_proofs, batch * C->pbytes);
    memcpy(status, S->h_stat, batch * sizeof(int));
    for (size_t i = 0; i < batch; i++)
        if (host_status[i]) {
            status[i] = host_status[i];
            memset(proofs + i * C->pbytes, 0, C->pbytes);
        }
    if (dbg) fprintf(stderr, "[p2aes] pack %.3f enqueue %.3f wait %.3f unpack %.3f s\n", t_b - t_a, t_c - t_b, t_d - t_c, now() - t_d);
    return P2_OK;
}

// In-process multi-device form of p2_prove_batch: contiguous balanced ranges of the batch, one host thread per handle.
// zk circuits: every handle blinds with ITS OWN key (drawn from the OS at load) and its own proof counter; handles that were
// given one fixed key by the test hook would blind different witnesses with the same (key, index) values.
static int prove_batch_multi_impl(p2_circuit* const* handles, size_t n_handles, size_t batch, const p2_assignment* inputs, uint8_t* proofs, int* status);
int p2_prove_batch_multi(p2_circuit* const* handles, size_t n_handles, size_t batch, const p2_assignment* inputs, uint8_t* proofs, int* status) {
    return guarded_rc([&] { return prove_batch_multi_impl(handles, n_handles, batch, inputs, proofs, status); });
}
static int prove_batch_multi_impl(p2_circuit* const* handles, size_t n_handles, size_t batch, const p2_assignment* inputs, uint8_t* proofs, int* status) {
    if (n_handles == 0 || !handles) return set_error("p2_prove_batch_multi needs at least one handle"), P2_ERR_INVALID;
    for (size_t h = 0; h < n_handles; h++)
        if (!handles[h] || handles[h]->pbytes != handles[0]->pbytes || handles[h]->verifier_data != handles[0]->verifier_data)
            return set_error("p2_prove_batch_multi: the handles are not loads of one compiled circuit"), P2_ERR_INVALID;
    if (batch == 0) return P2_OK;
    const size_t pb = handles[0]->pbytes, base = batch / n_handles, extra = batch % n_handles;
    std::vector<int> rc(n_handles, P2_OK);
    std::vector<std::string> err(n_handles);
    std::vector<std::thread> workers;
    size_t lo = 0;
    for (size_t h = 0; h < n_handles; h++) {
        const size_t cnt = base + (h < extra ? 1 : 0), first = lo;
        lo += cnt;
        if (cnt == 0) continue;
        workers.emplace_back([=, &rc, &err] {
            rc[h] = p2_prove_batch(handles[h], cnt, inputs + first, proofs + first * pb, status + first);
            if (rc[h] != P2_OK) err[h] = g_last_error;  // the error slot is thread-local: carry it to the caller's thread
        });
    }
    for (auto& w : workers) w.join();
    for (size_t h = 0; h < n_handles; h++)
        if (rc[h] != P2_OK) return set_error("handle " + std::to_string(h) + " (device " + std::to_string(handles[h]->device) + "): " + err[h]), rc[h];
    return P2_OK;
}

int p2_circuit_set_option(p2_circuit* C, const char* name, long value) {
    std::lock_guard<std::mutex> lock(C->mu);
    std::string k(name ? name : "");
    if (k == "chunk") {
        if (value < 1 || value > 4096) return set_error("option chunk: 1..4096 proofs"), P2_ERR_INVALID;
        C->opt_chunk = (size_t)value;
    } else if (k == "streams") {
        if (value < 1 || value > 8) return set_error("option streams: 1..8"), P2_ERR_INVALID;
        C->opt_streams = (size_t)value;
    } else if (k == "debug_timing") {
        C->opt_debug_timing = value != 0;
    } else {
        return set_error("unknown option (known: chunk, streams, debug_timing)"), P2_ERR_INVALID;
    }
    return P2_OK;
}

int p2_circuit_set_timing(p2_circuit* C, int enable) {
    std::lock_guard<std::mutex> lock(C->mu);
    C->timing_on = enable != 0;
    C->times.clear();
    return P2_OK;
}
size_t p2_circuit_get_timing(p2_circuit* C, p2_kernel_time* out, size_t cap) {
    std::lock_guard<std::mutex> lock(C->mu);
    size_t k = 0;
    for (auto& kv : C->times) {
        if (k < cap) {
            memset(&out[k], 0, sizeof(out[k]));
            strncpy(out[k].name, kv.first.c_str(), sizeof(out[k].name) - 1);
            out[k].ms = kv.second.first;
            out[k].count = kv.second.second;
        }
        k++;
    }
    return k;
}

int p2_circuit_debug_read(p2_circuit* C, const char* name_c, size_t index, uint64_t* out, size_t cap, size_t* n_written) {
    std::lock_guard<std::mutex> lock(C->mu);
    HIPCHECK(hipSetDevice(C->device));
    HIPCHECK(hipDeviceSynchronize());
    const Circuit& c = C->c;
    std::string name(name_c);
    // `index` addresses proof (index % chunk) of the workspace that handled chunk (index / chunk) of the last call
    struct CurGuard {
        p2_circuit* C;
        ~CurGuard() { C->cur = nullptr; }
    } guard{C};
    if (!C->ws.empty()) {
        C->cur = C->ws[(index / C->chunk) % C->ws.size()];
        index %= C->chunk;
    }
    const size_t n = C->n, N = C->N;
    const u32 zc = c.num_zs_cols(), qc = c.num_quotient_cols(), act = C->active_wires, cap_words = 4u << c.cfg.cap_height;
    const u64* src = nullptr;
    size_t count = 0;
    if (name == "pre_cap") { src = C->pre_tree.dig + cap_off(C->pre_tree, c.cfg.cap_height); count = cap_words; }
    else if (name == "pre_coeffs") { src = C->d_pre_coeffs; count = (size_t)c.num_preprocessed() * n; }
    else if (C->ws.empty()) return set_error("nothing has been proven yet"), P2_ERR_INVALID;
    else if (name == "values") { src = C->cur->d_values + index * c.num_slots; count = c.num_slots; }
    else if (name == "wires") { src = C->cur->d_wires + index * act * n; count = (size_t)act * n; }
    else if (name == "wires_coeffs") { src = C->cur->d_wcoef + index * act * n; count = (size_t)act * n; }
    else if (name == "wires_lde") { src = C->cur->d_wlde + index * (act + c.salt()) * N; count = (size_t)(act + c.salt()) * N; }
    else if (name == "wires_cap") { src = C->cur->wtree.dig + index * C->cur->wtree.stride() + cap_off(C->cur->wtree, c.cfg.cap_height); count = cap_words; }
    else if (name == "zs") { src = C->cur->d_zs + index * zc * n; count = (size_t)zc * n; }
    else if (name == "zs_cap") { src = C->cur->ztree.dig + index * C->cur->ztree.stride() + cap_off(C->cur->ztree, c.cfg.cap_height); count = cap_words; }
    else if (name == "quotient_values") { src = C->cur->d_qvals + index * 2 * N; count = 2 * N; }
    else if (name == "quotient_coeffs") { src = C->cur->d_qcoef + index * qc * n; count = (size_t)qc * n; }
    else if (name == "quotient_cap") { src = C->cur->qtree.dig + index * C->cur->qtree.stride() + cap_off(C->cur->qtree, c.cfg.cap_height); count = cap_words; }
    else if (name == "challenges") { src = C->cur->d_chal + index * CH_WORDS; count = CH_WORDS; }
    else if (name == "openings") { src = C->cur->d_ev + index * 2 * C->ev_count; count = 2 * (size_t)C->ev_count; }
    else if (name == "fri_final_poly_in") { src = C->cur->d_fri_coef[0] + index * 2 * n; count = 2 * n; }
    else return set_error("unknown debug buffer"), P2_ERR_INVALID;
    if (count > cap) return set_error("debug buffer too small"), P2_ERR_INVALID;
    HIPCHECK(hipMemcpy(out, src, count * 8, hipMemcpyDeviceToHost));
    *n_written = count;
    return P2_OK;
}

// ---------------------------------------------------------------------------------- primitives (parity tests)
static int pick_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_error("no HIP device available"), P2_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return set_error("device index out of range"), P2_ERR_INVALID;
    HIPCHECK(hipSetDevice(device));
    return 0;
}
// ---- device self-test
namespace p2k {
__global__ void k_selftest(unsigned long long* bad, u64 seed, size_t threads) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= threads) return;
    u64 x = (seed | 1) + 0x9E3779B97F4A7C15ull * (t + 1);
    auto rnd = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    unsigned long long b = 0;
    for (int i = 0; i < 64; i++) {
        u64 hi = rnd(), lo = rnd();
        const int m = i & 15;
        if (m == 0) hi &= gl::EPS;
        if (m == 1) lo &= gl::EPS;
        if (m == 2) hi |= ~gl::EPS;
        if (m == 3) lo |= ~gl::EPS;
        if (m == 4) hi = 0;
        if (m == 5) lo = 0;
        if (m == 6) {
            hi = ~0ull;
            lo = ~0ull - (x & 3);
        }
        if (m == 7) {
            hi &= ~gl::EPS;
            lo &= gl::EPS;
        }
        const u64 want = gl::reduce128(hi, lo);  // textbook form
        if (glf::canon(glf::red128(hi, lo)) != want) b++;
        glf::Acc a;
        a.init();
        a.fma(hi, lo);
        a.fma(lo, lo);
        if (glf::canon(a.reduce()) != gl::add(gl::mul(hi % gl::P, lo % gl::P), gl::mul(lo % gl::P, lo % gl::P))) b++;
        // a carry-consuming chain right after a reduction: the pattern the backend used to break
        const u64 r = glf::red128(hi, lo);
        u32 k, K;
        const u32 c0 = __builtin_addc((u32)r, 0xFFFFFFFFu, 0u, &k);
        const u32 c1 = __builtin_addc((u32)(r >> 32), 0u, k, &K);
        if ((K ? (((u64)c1 << 32) | c0) : r) != want) b++;
        // the mad-based field operations every kernel uses (gl.h) against the textbook forms
        const u64 x = hi % gl::P, y = lo % gl::P, z = rnd() % gl::P;
        if (gl::mul(x, y) != gl::mul_ref(x, y)) b++;
        if (gl::mul(hi, lo) != gl::mul_ref(x, y)) b++;  // non-canonical inputs are fine for mul
        if (gl::sub(x, y) != gl::sub_ref(x, y) || gl::sub(y, x) != gl::sub_ref(y, x)) b++;
        if (gl::mul_add(x, y, z) != gl::add_ref(gl::mul_ref(x, y), z)) b++;
        if (gl::add(x, y) != gl::add_ref(x, y) || gl::add(x, gl::P - 1) != gl::add_ref(x, gl::P - 1) || gl::add(y, gl::P - 1 - (y & 1)) != gl::add_ref(y, gl::P - 1 - (y & 1))) b++;
        if (glf::canon(hi) != hi % gl::P) b++;
        // the reduction's borrow case (R < H.hi + k), which random inputs never reach: 2^48 * (m 2^48) = m 2^96 = -m, and
        // 2^32 * (m 2^32) = m 2^64 with a zero low limb.  Odd threads keep random operands, so that one wave holds lanes that
        // borrow next to lanes that carry (the correction is chosen per wave, then applied per lane).
        {
            const u64 mm = ((t * 64 + i) & 0xFFFE) + 1;
            const u64 p48 = 1ull << 48, q = (t & 1) ? y : (mm << 48), pa = (t & 1) ? x : p48;
            if (gl::mul(pa, q) != gl::mul_ref(pa, q)) b++;
            if (glf::canon(glf::mulr(pa, q)) != gl::mul_ref(pa, q)) b++;
            if (gl::mul_add(pa, q, z) != gl::add_ref(gl::mul_ref(pa, q), z)) b++;
            const u64 e = (t & 2) ? (mm << 32) : (gl::P - mm), f = (t & 2) ? (1ull << 32) : (1ull << 48);
            if (gl::mul(e, f) != gl::mul_ref(e, f)) b++;
        }
    }
    u64 s0[12], s1[12];
    for (int k = 0; k < 12; k++) {
        s1[k] = (t & 3) == 3 ? rnd() : rnd() % gl::P;
        s0[k] = s1[k] % gl::P;
    }
    gl::poseidon(s0);
    glf::poseidon(s1);
    for (int k = 0; k < 12; k++) b += s0[k] != s1[k];
    if (b) atomicAdd(bad, b);
}
// the cooperative permutation (16 lanes per state) against the plain one: every group draws its own state
__global__ __launch_bounds__(64) void k_selftest_coop(unsigned long long* bad, u64 seed) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, grp = t >> 4, i = t & 15;
    u64 st[12];
    u64 x = (seed | 1) + 0x9E3779B97F4A7C15ull * (grp + 1);
    for (int k = 0; k < 12; k++) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        st[k] = (grp & 7) == 0 ? (k & 1 ? gl::P - 1 : 0) : x % gl::P;
    }
    u64 mine = 0;
    for (int k = 0; k < 12; k++)
        if ((u32)k == i) mine = st[k];
    const u64 got = glf::poseidon_coop(mine, i);
    gl::poseidon(st);  // every lane redundantly, the textbook form
    u64 want = 0;
    for (int k = 0; k < 12; k++)
        if ((u32)k == i) want = st[k];
    if (got != want) atomicAdd(bad, 1ull);
}
}  // namespace p2k

int p2_selftest_device(uint64_t seed, size_t threads, int device) {
    if (hipSetDevice(device) != hipSuccess) return set_error("no such HIP device"), -P2_ERR_HIP;
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, 8) != hipSuccess || hipMemset(d, 0, 8) != hipSuccess) return set_error("hipMalloc failed"), -P2_ERR_HIP;
    hipLaunchKernelGGL(p2k::k_selftest, dim3((u32)((threads + 255) / 256)), dim3(256), 0, 0, d, (u64)seed, threads);
    hipLaunchKernelGGL(p2k::k_selftest_coop, dim3((u32)std::min<size_t>(std::max<size_t>(threads / 64, 1), 4096)), dim3(64), 0, 0, d, (u64)seed);
    unsigned long long h = 0;
    hipError_t e = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(hipGetErrorString(e)), -P2_ERR_HIP;
    return (int)std::min<unsigned long long>(h, 0x7FFFFFFF);
}

int p2_gpu_poseidon(uint64_t* states, size_t n_perm, int device) {
    if (int rc = pick_device(device)) return rc;
    u64* d;
    HIPCHECK(hipMalloc((void**)&d, n_perm * 96));
    HIPCHECK(hipMemcpy(d, states, n_perm * 96, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_poseidon_states, g1(n_perm, 256), dim3(256), 0, 0, d, n_perm);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpy(states, d, n_perm * 96, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return P2_OK;
}
// A throw-away circuit-less context for the NTT / Merkle primitives
struct PrimCtx {
    p2_circuit C;
    int init(int device, int degree_bits) {
        if (int rc = pick_device(device)) return rc;
        C.device = device;
        C.logn = (u32)degree_bits;
        C.n = (size_t)1 << degree_bits;
        C.c.degree_bits = (u32)degree_bits;
        C.lde_bits = C.logn + 3;
        C.N = C.n << 3;
        HIPCHECK(hipStreamCreate(&C.stream));
        HIPCHECK(hipFuncSetAttribute((const void*)k_ntt_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void*)k_ntt_r16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r16_lds_bytes(14)));
        HIPCHECK(hipFuncSetAttribute((const void*)k_ntt_r16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r16_lds_bytes(13)));
        size_t n = C.n;
        std::vector<u64> twf(n), twi(n);
        u64 w = gl::root_of_unity(degree_bits), wi = gl::inv(w), x = 1, xi = 1;
        for (size_t i = 0; i < n; i++) {
            twf[i] = x;
            twi[i] = xi;
            x = gl::mul(x, w);
            xi = gl::mul(xi, wi);
        }
        if (upload(&C, &C.d_tw_fwd_full, twf.data(), n) || upload(&C, &C.d_tw_inv_full, twi.data(), n)) return P2_ERR_HIP;
        C.d_tw_fwd = C.d_tw_fwd_full;
        C.d_tw_inv = C.d_tw_inv_full;
        if (ensure_pass1_table(&C, C.d_tw_fwd_full, (u32)degree_bits) || ensure_pass1_table(&C, C.d_tw_inv_full, (u32)degree_bits)) return P2_ERR_HIP;
        C.opt_pass1_radix2 = getenv("P2AES_PASS1_RADIX2") != nullptr;
        if (const char* e = getenv("P2AES_PASS1_WAVES")) C.opt_pass1_waves = atoi(e) == 2 ? 2 : 4;
        HIPCHECK(hipFuncSetAttribute((const void*)k_ntt_pass1_r16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void*)k_ntt_pass1_r16<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        std::vector<u64> bases(8);
        u64 wl = gl::root_of_unity(degree_bits + 3);
        for (u32 j = 0; j < 8; j++) bases[j] = gl::mul(gl::MULT_GEN, gl::pow(wl, j));
        u64* d_b;
        if (upload(&C, &d_b, bases.data(), 8)) return P2_ERR_HIP;
        if (dalloc(&C, &C.d_shift_pows[0], 8 * n)) return P2_ERR_HIP;
        hipLaunchKernelGGL(k_pow_table, g1(n, 256, 8), dim3(256), 0, C.stream, C.d_shift_pows[0], d_b, (u32)n, (u64)1);
        HIPCHECK(hipGetLastError());
        return 0;
    }
    ~PrimCtx() {
        if (C.stream) (void)hipStreamSynchronize(C.stream);
        for (void* p : C.allocs) (void)hipFree(p);
        if (C.stream) (void)hipStreamDestroy(C.stream);
    }
};
int p2_gpu_intt(const uint64_t* values, size_t cols, int degree_bits, uint64_t* coeffs, int device) {
    if (degree_bits < 1 || degree_bits > 22) return set_error("degree_bits must be in 1..22"), P2_ERR_INVALID;
    PrimCtx ctx;
    if (int rc = ctx.init(device, degree_bits)) return rc;
    p2_circuit* C = &ctx.C;
    size_t n = C->n;
    u64 *d_in, *d_out, *d_scratch;
    if (upload(C, &d_in, values, cols * n) || dalloc(C, &d_out, cols * n) || dalloc(C, &d_scratch, cols * n)) return P2_ERR_HIP;
    if (intt_cols(C, d_in, d_out, (u32)cols, 0, 1, d_scratch, 0)) return P2_ERR_HIP;
    HIPCHECK(hipStreamSynchronize(C->stream));
    HIPCHECK(hipMemcpy(coeffs, d_out, cols * n * 8, hipMemcpyDeviceToHost));
    return P2_OK;
}
int p2_gpu_lde(const uint64_t* coeffs, size_t cols, int degree_bits, int rate_bits, uint64_t* lde, int device) {
    if (degree_bits < 1 || degree_bits > 22 || rate_bits != 3) return set_error("degree_bits must be in 1..22 and rate_bits 3"), P2_ERR_INVALID;
    PrimCtx ctx;
    if (int rc = ctx.init(device, degree_bits)) return rc;
    p2_circuit* C = &ctx.C;
    C->c.cfg.rate_bits = 3;
    size_t n = C->n;
    u64 *d_in, *d_out;
    if (upload(C, &d_in, coeffs, cols * n) || dalloc(C, &d_out, cols * 8 * n)) return P2_ERR_HIP;
    if (lde_cols(C, d_in, 0, d_out, 0, (u32)cols, 0, 1)) return P2_ERR_HIP;
    HIPCHECK(hipStreamSynchronize(C->stream));
    HIPCHECK(hipMemcpy(lde, d_out, cols * 8 * n * 8, hipMemcpyDeviceToHost));
    return P2_OK;
}
int p2_gpu_merkle_cap(const uint64_t* cols_major, size_t cols, size_t num_leaves, int cap_height, uint64_t* cap, int device) {
    u32 bits = 0;
    while (((size_t)1 << bits) < num_leaves) bits++;
    if (((size_t)1 << bits) != num_leaves || (int)bits < cap_height) return set_error("num_leaves must be a power of two >= 2^cap_height"), P2_ERR_INVALID;
    PrimCtx ctx;
    if (int rc = ctx.init(device, 4)) return rc;
    p2_circuit* C = &ctx.C;
    C->c.cfg.cap_height = (u32)cap_height;
    u64* d_in;
    Tree t;
    t.bits = bits;
    if (upload(C, &d_in, cols_major, cols * num_leaves) || dalloc(C, &t.dig, t.stride())) return P2_ERR_HIP;
    if (merkle_build(C, d_in, (u32)cols, (u32)cols, num_leaves, 0, t, 1)) return P2_ERR_HIP;
    HIPCHECK(hipStreamSynchronize(C->stream));
    HIPCHECK(hipMemcpy(cap, t.dig + cap_off(t, (u32)cap_height), ((size_t)4 << cap_height) * 8, hipMemcpyDeviceToHost));
    return P2_OK;
}
}

/* p2aes.h -- C ABI of the MI355X-native Plonky2 proving backend for the 0xPARC/plonky2-aes gadget circuits.
 *
 * This is the drop-in boundary.  The reference calls the third-party `plonky2` crate
 * through five kinds of call sites; each group of entry points below replaces one of them:
 *
 *   reference call site (file:line)                                          entry points here
 *   ---------------------------------------------------------------------    --------------------------------
 *   CircuitBuilder::<F,D>::new(config) + builder methods                     p2_builder_*
 *     aes-gcm/src/circuit_aes.rs:178,182,186,193,249,250,300,317,334,356,357
 *     aes-gcm/src/circuit_gcm.rs:163,314,323,339,342,356,361-364,396,403,415,423,424
 *   gadget constructors (AesGcmTarget::build etc.)                           p2_aes_*, p2_gcm_*
 *     aes-gcm/src/circuit_gcm.rs:49-172, aes-gcm/src/circuit_aes.rs:76-275
 *   builder.build::<PoseidonGoldilocksConfig>()                              p2_builder_build -> blob,
 *     aes-gcm/src/circuit_gcm.rs:771, examples/aes_gcm_128.rs:46 (19 sites)  p2_circuit_load (GPU preprocessing)
 *   PartialWitness::new / pw.set_target / data.prove(pw)   ** HOT PATH **    p2_prove_batch
 *     aes-gcm/src/circuit_aes.rs:283, circuit_gcm.rs:779-781 (20 sites)
 *   data.verify(proof)                                                       p2_verify
 *     aes-gcm/src/circuit_gcm.rs:782 (19 sites)
 *   native_gcm::encrypt (witness values)  aes-gcm/src/native_gcm.rs:16       p2_native_aes_gcm_encrypt
 *
 * Conventions: plain pointers and sizes only; every function that can fail returns 0 on success and a
 * non-zero code otherwise, with a thread-local message available from p2_last_error().  The caller owns
 * every buffer it passes in; the library owns handles and all device memory.  Nothing here touches the CPU
 * oracle under oracle/: proving runs on the GPU or fails with P2_ERR_NO_DEVICE.
 */
#ifndef P2AES_H
#define P2AES_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    P2_OK = 0,
    P2_ERR_INVALID = 1,   /* bad argument / malformed blob or proof */
    P2_ERR_NO_DEVICE = 2, /* no usable HIP device: proving has no CPU fallback */
    P2_ERR_HIP = 3,       /* a HIP runtime call failed */
    P2_ERR_VERIFY = 4     /* proof rejected (reason in p2_last_error) */
};
/* per-proof status written by p2_prove_batch (mirrors `data.prove(pw)` returning Err, circuit_aes.rs:403-405) */
enum {
    P2_PROOF_OK = 0,
    P2_PROOF_WITNESS_CONFLICT = 1, /* generator output conflicts with a pre-set target, or lookup input not in table */
    P2_PROOF_MISSING_INPUT = 2,    /* some generator never ran: an input target was not set */
    P2_PROOF_ZETA_IN_SUBGROUP = 3, /* "Opening point is in the subgroup." */
    P2_PROOF_POW_NOT_FOUND = 4     /* no proof-of-work witness among the 2^21 candidates searched (probability ~e^-32) */
};

const char* p2_last_error(void);

/* ------------------------------------------------------------------ CircuitBuilder (host) */
typedef struct p2_builder p2_builder;
/* Targets are opaque 64-bit handles (virtual target index, or a routed wire). */
typedef uint64_t p2_target;

p2_builder* p2_builder_new(void);    /* CircuitConfig::standard_recursion_config() */
p2_builder* p2_builder_new_zk(void); /* CircuitConfig::standard_recursion_zk_config() (examples/aes_gcm_128.rs:36) */
void p2_builder_free(p2_builder*);
p2_target p2_builder_add_virtual_target(p2_builder*);
p2_target p2_builder_constant(p2_builder*, uint64_t c);
p2_target p2_builder_zero(p2_builder*);
p2_target p2_builder_one(p2_builder*);
/* c0*m0*m1 + c1*addend */
p2_target p2_builder_arithmetic(p2_builder*, uint64_t c0, uint64_t c1, p2_target m0, p2_target m1, p2_target addend);
p2_target p2_builder_mul_const_add(p2_builder*, uint64_t c, p2_target x, p2_target y); /* c*x + y */
p2_target p2_builder_add(p2_builder*, p2_target x, p2_target y);
p2_target p2_builder_sub(p2_builder*, p2_target x, p2_target y);
p2_target p2_builder_mul(p2_builder*, p2_target x, p2_target y);
p2_target p2_builder_select(p2_builder*, p2_target b, p2_target x, p2_target y); /* if b {x} else {y} */
p2_target p2_builder_is_equal(p2_builder*, p2_target x, p2_target y);
void p2_builder_connect(p2_builder*, p2_target x, p2_target y);
/* pairs = n_pairs * (input u16, output u16); returns the LUT index (an identical table is re-used) */
size_t p2_builder_add_lookup_table_from_pairs(p2_builder*, const uint16_t* pairs, size_t n_pairs);
/* returns the looked-up output target; (size_t)-1 lut index is an error -> returns UINT64_MAX */
p2_target p2_builder_add_lookup_from_index(p2_builder*, p2_target looking_in, size_t lut_index);
size_t p2_builder_num_gates(const p2_builder*);
/* Compile the circuit.  On success *blob points to a library-owned buffer (release with p2_blob_free). */
int p2_builder_build(p2_builder*, uint8_t** blob, size_t* blob_len);
void p2_blob_free(uint8_t* blob);

/* ------------------------------------------------------------------ AES / GCM gadgets (host) */
size_t p2_aes_sbox_lut(p2_builder*);
size_t p2_aes_byte_xor_lut(p2_builder*);
size_t p2_aes_gf_2_8_mul_lut(p2_builder*);
size_t p2_gcm_u8_unit_right_shift_lut(p2_builder*);
size_t p2_gcm_u8_bitref_lut(p2_builder*);
p2_target p2_aes_add_virtual_byte_target(p2_builder*, size_t u8_table_idx);
p2_target p2_aes_add_virtual_byte_target_unsafe(p2_builder*);
/* States are 16 targets, element [4*i + j] = row i, column j (StateTarget.0[i][j]). */
void p2_aes_state_sub_bytes(p2_builder*, size_t sbox_lut, const p2_target* s, p2_target* out);
void p2_aes_state_mix_columns(p2_builder*, size_t xor_lut, size_t mul_lut, const p2_target* s, p2_target* out);
p2_target p2_aes_gf_2_8_mul(p2_builder*, size_t mul_lut, p2_target x, p2_target y);
p2_target p2_aes_gf_2_8_add(p2_builder*, size_t xor_lut, p2_target x, p2_target y);
/* key: 4*nk bytes targets; out: 4*(nr+1) words * 4 targets, word-major */
void p2_aes_key_expansion(p2_builder*, int nk, int nr, size_t xor_lut, size_t sbox_lut, const p2_target* key, p2_target* out);
void p2_aes_encrypt_block(p2_builder*, int nr, size_t xor_lut, size_t mul_lut, size_t sbox_lut, const p2_target* state,
                          const p2_target* expanded_key, p2_target* out_state);
void p2_gcm_gctr(p2_builder*, int nr, size_t xor_lut, size_t mul_lut, size_t sbox_lut, const p2_target* expanded_key,
                 const p2_target* icb, const p2_target* x, size_t len, p2_target* y);
void p2_gcm_right_shift_one(p2_builder*, size_t shift_lut, const p2_target* v, p2_target* out);
void p2_gcm_inc32(p2_builder*, const p2_target* block, p2_target* out);
void p2_gcm_gf_2_128_mul(p2_builder*, size_t xor_lut, size_t shift_lut, size_t bitref_lut, const p2_target* x,
                         const p2_target* y, p2_target* out);
int p2_gcm_ghash(p2_builder*, size_t xor_lut, size_t shift_lut, size_t bitref_lut, const p2_target* h, const p2_target* x,
                 size_t len, p2_target* out);
/* AesGcmTarget<NK,4,NR,L,TAG>::build.  Outputs: key[4*nk], nonce[12], pt[L], ct[L], tag[16]. */
int p2_aes_gcm_build(p2_builder*, int nk, int nr, size_t L, int with_tag, p2_target* key, p2_target* nonce, p2_target* pt,
                     p2_target* ct, p2_target* tag);

/* ------------------------------------------------------------------ Poseidon hashing / poseidon-cipher (host) */
/* builder.hash_n_to_m_no_pad::<PoseidonHash>(inputs, m)  (poseidon-cipher/src/circuit.rs:123): PoseidonGate rows */
int p2_builder_hash_n_to_m_no_pad(p2_builder*, const p2_target* inputs, size_t n, p2_target* outputs, size_t m);
/* PoseidonEncryptTarget::<L>::build (poseidon-cipher/src/circuit.rs:49).  ks: 10 targets (x then u), m: 5*L,
 * nonce: 2, ct: 5*(L+1) */
int p2_poseidon_cipher_build(p2_builder*, size_t L, p2_target* ks, p2_target* m, p2_target* nonce, p2_target* ct);
/* native hash / cipher (poseidon-cipher/src/lib.rs:41,75,113).  msg: 5*n_msg words; ct: 5*(ceil3(n_msg)+1) words */
void p2_native_hash_n_to_m_no_pad(const uint64_t* in, size_t n, uint64_t* out, size_t m);
void p2_native_poseidon_encrypt(const uint64_t* ks10, const uint64_t* msg, size_t n_msg, const uint64_t* nonce2, uint64_t* ct);
int p2_native_poseidon_decrypt(const uint64_t* ks10, const uint64_t* ct, size_t n_ct, const uint64_t* nonce2, size_t l, uint64_t* msg);

/* ------------------------------------------------------------------ ecGFp5 / ElGamal (host) */
/* The curve is pod2's (not in the reference tree; see csrc/ecgfp5.h).  A point is its affine (x, u) coordinates, ten
 * words, x then u -- pod2 `Point::as_fields` (hashed_elgamal.rs:23); a scalar is five little-endian 64-bit limbs. */
#define P2_POINT_WORDS 10
#define P2_SCALAR_LIMBS 5
#define P2_SCALAR_BITS 320
void p2_ecgfp5_group_order(uint64_t out[5]);                     /* GROUP_ORDER (lib.rs:12) */
void p2_ecgfp5_generator(uint64_t out[10]);                      /* Point::generator() */
void p2_ecgfp5_mul(const uint64_t k[5], const uint64_t p[10], uint64_t out[10]); /* &k * P */
void p2_ecgfp5_add(const uint64_t p[10], const uint64_t q[10], uint64_t out[10]);
void p2_ecgfp5_neg(const uint64_t p[10], uint64_t out[10]);      /* Point::inverse (elgamal.rs:21) */
int p2_ecgfp5_is_in_subgroup(const uint64_t p[10]);              /* 1 / 0 */
void p2_ecgfp5_compress(const uint64_t p[10], uint64_t w[5]);    /* compress_from_subgroup (lib.rs:82) */
int p2_ecgfp5_decompress(const uint64_t w[5], uint64_t out[10]); /* decompress_into_subgroup (lib.rs:74); P2_ERR_INVALID if none */
/* Randomness comes from the operating system's CSPRNG, as the reference's OsRng (lib.rs:35,64): */
int p2_ecgfp5_random_scalar(uint64_t out[5]);                    /* gen_biguint_below(&GROUP_ORDER) (lib.rs:35) */
int p2_ecgfp5_random_point(uint64_t out[10]);                    /* Point::new_rand_from_subgroup */
/* encode_binary / decode_binary (lib.rs:48, :80): 160 message bits as five 32-bit limbs, random padding above them */
int p2_ecgfp5_encode_binary(const uint32_t limbs[5], uint64_t out[10]);
/* TEST / BENCHMARK ONLY: the same three from a 64-bit seed (SplitMix64, not cryptographic) so that inputs are
 * reproducible.  A scalar drawn this way carries at most 64 bits of entropy: never a real key or nonce. */
void p2_ecgfp5_random_scalar_seeded(uint64_t seed, uint64_t out[5]);
void p2_ecgfp5_random_point_seeded(uint64_t seed, uint64_t out[10]);
void p2_ecgfp5_encode_binary_seeded(const uint32_t limbs[5], uint64_t seed, uint64_t out[10]);
void p2_ecgfp5_decode_binary(const uint64_t p[10], uint32_t limbs[5]);
/* elgamal.rs:11, :19; hashed_elgamal.rs:19, :28.  Scalars must be below the group order (P2_ERR_INVALID otherwise). */
int p2_elgamal_encrypt(const uint64_t pk[10], const uint64_t nonce[5], const uint64_t msg[10], uint64_t c0[10], uint64_t c1[10]);
int p2_elgamal_decrypt(const uint64_t sk[5], const uint64_t c0[10], const uint64_t c1[10], uint64_t msg[10]);
int p2_hashed_elgamal_encrypt(const uint64_t pk[10], const uint64_t nonce[5], const uint64_t msg[5], uint64_t c0[10], uint64_t ct[5]);
int p2_hashed_elgamal_decrypt(const uint64_t sk[5], const uint64_t c0[10], const uint64_t ct[5], uint64_t msg[5]);
/* pod2 CircuitBuilderElliptic / CircuitBuilderBits as the reference calls them (ecgfp5/src/circuit.rs:33-38,
 * elgamal/circuit.rs:34-37,70-72): points are 10 targets, a BigUInt320Target is its 320 little-endian bit targets. */
void p2_builder_add_virtual_point_target(p2_builder*, p2_target out[10]);
void p2_builder_constant_point(p2_builder*, const uint64_t p[10], p2_target out[10]);
void p2_builder_add_virtual_biguint320_target(p2_builder*, p2_target bits[320]);
int p2_builder_multiply_point(p2_builder*, const p2_target bits[320], const p2_target p[10], p2_target out[10]);
void p2_builder_add_point(p2_builder*, const p2_target p[10], const p2_target q[10], p2_target out[10]);
/* CircuitBuilderECGFP5PublicKey::public_key (ecgfp5/src/circuit.rs:35) */
int p2_builder_public_key(p2_builder*, const p2_target sk_bits[320], p2_target pk[10]);
/* CircuitBuilderElGamal::elgamal_encrypt (elgamal/circuit.rs:28) */
int p2_builder_elgamal_encrypt(p2_builder*, const p2_target pk[10], const p2_target nonce_bits[320], const p2_target msg[10],
                               p2_target c0[10], p2_target c1[10]);
/* CircuitBuilderHashedElGamal::hashed_elgamal_encrypt (hashed_elgamal/circuit.rs:33) */
int p2_builder_hashed_elgamal_encrypt(p2_builder*, const p2_target pk[10], const p2_target nonce_bits[320], const p2_target msg[5],
                                      p2_target c0[10], p2_target ct[5]);

/* ------------------------------------------------------------------ self-test (host) */
/* Checks the host build of the arithmetic the kernels share with it: the carry-chain reduction against 128-bit
 * arithmetic (random and extreme inputs), and the restructured Poseidon (poseidon_fast.h: lazy reduction, sparse partial
 * rounds, accumulator fold) against the plain 30-round permutation.  0 = all agree; otherwise the number of mismatches. */
int p2_selftest_host(uint64_t seed, size_t n_reductions, size_t n_permutations);
/* The same comparison compiled for the device and run there (threads x 64 reductions and threads x 1 permutation,
 * textbook forms as the reference): guards against code-generation regressions such as the add-with-carry fold described
 * in DESIGN.md.  Returns the number of mismatches, or a negative P2_ERR_* code. */
int p2_selftest_device(uint64_t seed, size_t threads, int device);

/* ------------------------------------------------------------------ native cipher (host; witness values) */
uint8_t p2_native_gf_2_8_mul(uint8_t a, uint8_t b);
void p2_native_aes_key_expansion(const uint8_t* key, int nk, int nr, uint8_t* out /* 16*(nr+1) */);
void p2_native_aes_encrypt_block(const uint8_t* key, int nk, int nr, const uint8_t* in16, uint8_t* out16);
void p2_native_gf_2_128_mul(const uint8_t* x16, const uint8_t* y16, uint8_t* out16);
void p2_native_ghash(const uint8_t* h16, const uint8_t* x, size_t len, uint8_t* out16);
void p2_native_gctr(const uint8_t* key, int nk, int nr, const uint8_t* icb16, const uint8_t* x, size_t len, uint8_t* y);
void p2_native_aes_gcm_encrypt(const uint8_t* key, int nk, int nr, const uint8_t* nonce12, const uint8_t* pt, size_t len,
                               uint8_t* ct, uint8_t* tag16);

/* ------------------------------------------------------------------ circuit info / verification (host) */
/* Shape of a compiled circuit without touching a device. */
typedef struct {
    uint32_t degree_bits, num_wires, num_routed_wires, num_constants_cols, num_zs_cols, num_quotient_cols, num_luts,
        num_ops, num_levels, num_slots, num_virtual_targets, num_fri_rounds;
    uint64_t proof_bytes; /* exact serialised proof size */
    uint32_t zero_knowledge, num_gate_kinds; /* standard_recursion_zk_config(); distinct gate types in the circuit */
} p2_circuit_info;
int p2_blob_info(const uint8_t* blob, size_t len, p2_circuit_info* out);
/* The device-side schedule of a compiled circuit's witness program for macro size `fuse` (csrc/witness_schedule.h; the prover
 * builds it at p2_circuit_load; P2AES_WITNESS_FUSE = longest chain, 1..8, default 1 = none), computed AND checked on the host:
 * every op is kept, every slot keeps its first producer, and every operand of every op is produced in an earlier level
 * or earlier in the op's own chain.  out = {levels, chains, longest chain, ops fused into chains}.  P2_ERR_INVALID if a
 * check fails. */
int p2_witness_schedule_check(const uint8_t* blob, size_t len, uint32_t fuse, uint32_t out[4]);
/* verifier_data = constants_sigmas_cap (16 digests) || circuit_digest, 68 u64 -- from p2_circuit_verifier_data */
int p2_verify(const uint8_t* blob, size_t blob_len, const uint64_t* verifier_data, size_t verifier_data_len,
              const uint8_t* proof, size_t proof_len);

/* ------------------------------------------------------------------ GPU prover */
typedef struct p2_circuit p2_circuit;
/* Uploads the compiled circuit to HIP device `device` and commits constants+sigmas there (one-time). */
p2_circuit* p2_circuit_load(const uint8_t* blob, size_t len, int device);
void p2_circuit_free(p2_circuit*);
int p2_circuit_verifier_data(const p2_circuit*, uint64_t* out, size_t cap, size_t* n_written);
/* Proofs per chunk of the workspaces the handle holds now (0 before the first proof): a batch is cut into equal chunks of
 * at most this many proofs, dealt round-robin to the proving streams; the size is the "chunk" option capped by free HBM. */
size_t p2_circuit_chunk_proofs(p2_circuit*);
/* zk circuits only.  Blinding values are the output of a Poseidon-based PRF under a 256-bit key (four field elements)
 * and a per-handle proof counter that advances with every proof attempted.  The key is drawn from the operating system's
 * CSPRNG at load time (upstream: OS randomness per proof) -- that is the production path and needs no call here.
 * TEST ONLY: p2_circuit_set_zk_key fixes the key (words are reduced mod p) and, when the key differs from the one the handle
 * holds, restarts the counter -- which makes proofs reproducible so the CPU oracle can check them byte for byte;
 * p2_circuit_set_zk_seed(s) is set_zk_key({s, 0, 0, 0}).  Both return P2_ERR_INVALID unless the environment holds
 * P2AES_ALLOW_FIXED_ZK_KEY=1 (the test suite sets it).  A fixed key is not secret, and one key on two handles (or set again
 * after proofs were made under another) blinds different witnesses with the same values: never outside tests. */
int p2_circuit_set_zk_key(p2_circuit*, const uint64_t key[4]);
int p2_circuit_set_zk_seed(p2_circuit*, uint64_t seed);
size_t p2_circuit_proof_bytes(const p2_circuit*);
/* One PartialWitness: (target, value) pairs, values canonical (< p). */
typedef struct {
    const p2_target* targets;
    const uint64_t* values;
    size_t count;
} p2_assignment;
/* Proves `batch` independent witnesses of one circuit.  proofs: batch * p2_circuit_proof_bytes() bytes.
 * status[i] receives a P2_PROOF_* code; a failed proof leaves its slot zeroed (a value >= p fails its witness with
 * P2_PROOF_WITNESS_CONFLICT, like a conflicting set_target).  The call returns after the proofs are in host memory.
 * Inputs and proofs travel through persistent pinned staging buffers owned by the handle.
 * Thread safety: `prove(&self)` in the reference takes a shared reference, and so does this: any number of host threads
 * may call p2_prove_batch / p2_prove_batch_device concurrently on the SAME handle (enqueueing is serialised inside; each
 * caller gets its own staging set) as well as on different handles. */
int p2_prove_batch(p2_circuit*, size_t batch, const p2_assignment* inputs, uint8_t* proofs, int* status);
/* The same call sharded over several handles of ONE compiled circuit, normally one handle per HIP device of the node
 * (p2_circuit_load(blob, len, d) for d = 0..N-1): witness i goes to the handle that owns its contiguous, balanced range of
 * the batch (the partition of SURVEY.md 8e: independent proofs, no data crosses devices), one host thread per handle.
 * proofs / status are indexed like the inputs.  P2_ERR_INVALID if the handles are not loads of the same circuit. */
int p2_prove_batch_multi(p2_circuit* const* handles, size_t n_handles, size_t batch, const p2_assignment* inputs,
                         uint8_t* proofs, int* status);
/* Same pipeline with inputs already resident on the device and proofs left on the device:
 * d_values: [batch][n_targets] u64 (device pointer), targets shared by the whole batch (host pointer); the value
 * 2^64-1 (not a field element) marks "this witness does not assign the target".
 * d_proofs: device buffer of batch * proof_bytes; d_status: device int[batch].  Asynchronous: kernels are enqueued on
 * the circuit's own streams.  `stream` (a hipStream_t passed as void*, NULL = the default stream) orders the call with
 * the caller: the proving streams wait for the work already enqueued on it and it then waits for the proofs, so work
 * the caller enqueues on `stream` afterwards sees them.  d_values must stay untouched until then.  Host code calls
 * p2_circuit_synchronize() (or synchronises `stream`) before reading the outputs. */
int p2_prove_batch_device(p2_circuit*, size_t batch, const p2_target* targets, size_t n_targets, const uint64_t* d_values,
                          uint8_t* d_proofs, int* d_status, void* stream);
int p2_circuit_synchronize(p2_circuit*);
/* Tuning knobs of a handle: "chunk" (proofs per workspace, default 128), "streams" (proving streams, default 2),
 * "debug_timing" (host-path phase times on stderr).  The environment variables P2AES_CHUNK / P2AES_STREAMS /
 * P2AES_DEBUG_TIMING set the defaults and are read once, in p2_circuit_load. */
int p2_circuit_set_option(p2_circuit*, const char* name, long value);
/* Per-kernel timing of the most recent batch (HIP events on the proving stream). */
typedef struct {
    char name[48];
    float ms;       /* total over launches */
    uint32_t count; /* launches */
} p2_kernel_time;
int p2_circuit_set_timing(p2_circuit*, int enable);
size_t p2_circuit_get_timing(p2_circuit*, p2_kernel_time* out, size_t cap);

/* ------------------------------------------------------------------ GPU primitives (parity tests / microbench) */
int p2_gpu_device_count(void);
/* data: n_perm * 12 u64 on host; permuted in place on the device */
int p2_gpu_poseidon(uint64_t* states, size_t n_perm, int device);
/* columns: [cols][n] coefficients (host) -> lde [cols][n<<rate_bits], bit-reversed index order (host) */
int p2_gpu_lde(const uint64_t* coeffs, size_t cols, int degree_bits, int rate_bits, uint64_t* lde, int device);
/* values [cols][n] -> coefficients [cols][n]; degree_bits 1..22 (above 14: two-pass transform) */
int p2_gpu_intt(const uint64_t* values, size_t cols, int degree_bits, uint64_t* coeffs, int device);
/* column-major leaves [cols][num_leaves] -> cap digests (2^cap_height * 4 u64) */
int p2_gpu_merkle_cap(const uint64_t* cols_major, size_t cols, size_t num_leaves, int cap_height, uint64_t* cap, int device);
/* debug: copy a named intermediate buffer of proof `index` of the last batch to the host
 * ("wires", "wires_cap", "zs", "zs_cap", "quotient_coeffs", "quotient_cap", "challenges", "openings", ...) */
int p2_circuit_debug_read(p2_circuit*, const char* name, size_t index, uint64_t* out, size_t cap, size_t* n_written);

#ifdef __cplusplus
}
#endif
#endif

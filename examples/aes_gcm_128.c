/* The reference's canonical entry point (aes-gcm/examples/aes_gcm_128.rs:16-54) over the C ABI, from plain C:
 * standard_recursion_zk_config (:36), AesGcm128Target<42>::build (:38), set_targets (:50), prove on the GPU (:52),
 * verify (:53).
 *   gcc -O2 -Iinclude examples/aes_gcm_128.c -o aes_gcm_128 -Lplonky2-aes_amd -lp2aes -Wl,-rpath,$PWD/plonky2-aes_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "p2aes.h"

#define L 42
int main(void) {
    uint8_t key[16], nonce[12] = {0}, pt[L], ct[L], tag[16];
    memset(key, 123, sizeof key); /* examples/aes_gcm_128.rs:20-22 */
    memset(pt, 231, sizeof pt);
    p2_native_aes_gcm_encrypt(key, 4, 10, nonce, pt, L, ct, tag);

    p2_builder* b = p2_builder_new_zk(); /* CircuitConfig::standard_recursion_zk_config() */
    p2_target tk[16], tn[12], tp[L], tc[L], tt[16];
    if (p2_aes_gcm_build(b, 4, 10, L, 0, tk, tn, tp, tc, tt)) return fprintf(stderr, "build: %s\n", p2_last_error()), 1;
    printf("AES-GCM-128 (L=%d) num_gates: %zu\n", L, p2_builder_num_gates(b));
    uint8_t* blob;
    size_t blob_len;
    if (p2_builder_build(b, &blob, &blob_len)) return fprintf(stderr, "compile: %s\n", p2_last_error()), 1;

    p2_circuit* c = p2_circuit_load(blob, blob_len, 0);
    if (!c) return fprintf(stderr, "load: %s\n", p2_last_error()), 1;

    /* aes_targets.set_targets(&mut pw, key, nonce, pt, ct, tag): TAG=false sets the 16 tag targets to 0 */
    enum { NT = 16 + 12 + L + L + 16 };
    p2_target targets[NT];
    uint64_t values[NT];
    size_t k = 0;
    for (int i = 0; i < 16; i++) targets[k] = tk[i], values[k++] = key[i];
    for (int i = 0; i < 12; i++) targets[k] = tn[i], values[k++] = nonce[i];
    for (int i = 0; i < L; i++) targets[k] = tp[i], values[k++] = pt[i];
    for (int i = 0; i < L; i++) targets[k] = tc[i], values[k++] = ct[i];
    for (int i = 0; i < 16; i++) targets[k] = tt[i], values[k++] = 0;
    p2_assignment pw = {targets, values, NT};

    size_t pb = p2_circuit_proof_bytes(c);
    uint8_t* proof = malloc(pb);
    int status = -1;
    if (p2_prove_batch(c, 1, &pw, proof, &status) || status) return fprintf(stderr, "prove: status %d %s\n", status, p2_last_error()), 1;
    uint64_t vd[80];
    size_t nvd;
    p2_circuit_verifier_data(c, vd, 80, &nvd);
    if (p2_verify(blob, blob_len, vd, nvd, proof, pb)) return fprintf(stderr, "verify: %s\n", p2_last_error()), 1;
    printf("proved and verified: %zu-byte proof\n", pb);
    values[16 + 12 + L + 3] ^= 1; /* a wrong ciphertext byte: prove must fail, not emit a bad proof */
    p2_prove_batch(c, 1, &pw, proof, &status);
    printf("wrong ciphertext -> status %d (expect 1)\n", status);
    free(proof);
    p2_circuit_free(c);
    p2_blob_free(blob);
    p2_builder_free(b);
    return status == 1 ? 0 : 1;
}

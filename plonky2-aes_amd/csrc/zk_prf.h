// Blinding PRF of the zero-knowledge configuration (definition: circuit.h, "Blinding randomness").
#pragma once
#include "circuit.h"
#include "gl.h"
#include "poseidon_fast.h"

namespace p2 {
// eight blinding elements: block `block` of (key, proof, domain)
GL_HD void zk_block(const ZkKey& key, u64 proof, u64 domain, u64 block, u64* out8) {
    u64 s[12] = {key.k[0], key.k[1], key.k[2], key.k[3], proof, domain, block, ZK_TAG % gl::P, 0, 0, 0, 0};
    glf::poseidon(s);
    for (int i = 0; i < 8; i++) out8[i] = s[i];
}
}  // namespace p2

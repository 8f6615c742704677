// Operating-system CSPRNG: getrandom(2), /dev/urandom as the fallback.  Used for the zk blinding key (prover_gpu.hip) and
// for the default secret keys / nonces of the ecgfp5 layer (ecgfp5.h), where the reference uses OsRng
// (ecgfp5/src/lib.rs:35,64; poseidon-cipher/src/lib.rs:32,37).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <sys/random.h>

#include <stdexcept>

namespace p2 {
inline void os_random_bytes(void* buf, size_t len) {
    uint8_t* p = (uint8_t*)buf;
    size_t got = 0;
    while (got < len) {
        ssize_t r = getrandom(p + got, len - got, 0);
        if (r <= 0) break;
        got += (size_t)r;
    }
    if (got < len) {
        FILE* f = fopen("/dev/urandom", "rb");
        if (f) {
            got += fread(p + got, 1, len - got, f);
            fclose(f);
        }
    }
    if (got < len) throw std::runtime_error("no operating-system randomness available");
}
// uniform field element (rejection sampling below p = 2^64 - 2^32 + 1)
inline uint64_t os_random_field() {
    for (;;) {
        uint64_t v;
        os_random_bytes(&v, 8);
        if (v < 0xFFFFFFFF00000001ull) return v;
    }
}
}  // namespace p2

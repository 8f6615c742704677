// ORACLE -- TEST INFRASTRUCTURE ONLY.  AVX2 instantiation of oracle_poseidon_simd.h (four permutations per call).
// Compiled with -mavx2; entered only after orc::simd_level() saw the bit in cpuid.
#include <immintrin.h>

#include "oracle_poseidon_simd.h"

namespace orc {
struct V256 {
    static constexpr int W = 4;
    __m256i v;
    static inline V256 set1(u64 x) { return V256{_mm256_set1_epi64x((long long)x)}; }
    static inline V256 gather(const u64* base, size_t stride) {
        if (stride == 1) return V256{_mm256_loadu_si256((const __m256i*)base)};
        const long long s = (long long)stride;
        return V256{_mm256_i64gather_epi64((const long long*)base, _mm256_set_epi64x(3 * s, 2 * s, s, 0), 8)};
    }
};
static inline V256 vadd(V256 a, V256 b) { return V256{_mm256_add_epi64(a.v, b.v)}; }
static inline V256 vsub(V256 a, V256 b) { return V256{_mm256_sub_epi64(a.v, b.v)}; }
static inline V256 vand(V256 a, V256 b) { return V256{_mm256_and_si256(a.v, b.v)}; }
static inline V256 vor(V256 a, V256 b) { return V256{_mm256_or_si256(a.v, b.v)}; }
static inline V256 vsrl32(V256 a) { return V256{_mm256_srli_epi64(a.v, 32)}; }
static inline V256 vsll32(V256 a) { return V256{_mm256_slli_epi64(a.v, 32)}; }
static inline V256 vmul32(V256 a, V256 b) { return V256{_mm256_mul_epu32(a.v, b.v)}; }
// unsigned a < b through the signed compare: flip the top bits
static inline __m256i ltu(V256 a, V256 b) {
    const __m256i top = _mm256_set1_epi64x((long long)0x8000000000000000ull);
    return _mm256_cmpgt_epi64(_mm256_xor_si256(b.v, top), _mm256_xor_si256(a.v, top));
}
static inline V256 vadd_if_lt(V256 r, V256 a, V256 b, V256 x) { return V256{_mm256_add_epi64(r.v, _mm256_and_si256(ltu(a, b), x.v))}; }
static inline V256 vsub_if_lt(V256 r, V256 a, V256 b, V256 x) { return V256{_mm256_sub_epi64(r.v, _mm256_and_si256(ltu(a, b), x.v))}; }
static inline void vstore(u64* dst, V256 a) { _mm256_storeu_si256((__m256i*)dst, a.v); }

void simd256_hash_rows(const SparsePoseidon* S, const u64* rows, size_t row_stride, size_t width, Digest* out) {
    PoseidonLanes<V256>::hash_rows(*S, rows, row_stride, width, out);
}
void simd256_compress_pairs(const SparsePoseidon* S, const Digest* src, Digest* dst) { PoseidonLanes<V256>::compress_pairs(*S, src, dst); }
void simd256_test_arith(const u64* a, const u64* b, const u64* c, u64* out) { PoseidonLanes<V256>::test_arith(a, b, c, out); }
void simd256_dif_stage(u64* a, size_t n, size_t half, const u64* tw) { PoseidonLanes<V256>::dif_stage(a, n, half, tw); }
}  // namespace orc

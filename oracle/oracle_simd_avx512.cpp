// ORACLE -- TEST INFRASTRUCTURE ONLY.  AVX-512 instantiation of oracle_poseidon_simd.h (eight permutations per call).
// Compiled with -mavx512f -mavx512dq; entered only after orc::simd_level() saw both bits in cpuid.
#include <immintrin.h>

#include "oracle_poseidon_simd.h"

namespace orc {
struct V512 {
    static constexpr int W = 8;
    __m512i v;
    static inline V512 set1(u64 x) { return V512{_mm512_set1_epi64((long long)x)}; }
    static inline V512 gather(const u64* base, size_t stride) {
        if (stride == 1) return V512{_mm512_loadu_si512((const void*)base)};
        const __m512i idx = _mm512_mullo_epi64(_mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0), _mm512_set1_epi64((long long)stride));
        return V512{_mm512_i64gather_epi64(idx, base, 8)};
    }
};
static inline V512 vadd(V512 a, V512 b) { return V512{_mm512_add_epi64(a.v, b.v)}; }
static inline V512 vsub(V512 a, V512 b) { return V512{_mm512_sub_epi64(a.v, b.v)}; }
static inline V512 vand(V512 a, V512 b) { return V512{_mm512_and_si512(a.v, b.v)}; }
static inline V512 vor(V512 a, V512 b) { return V512{_mm512_or_si512(a.v, b.v)}; }
static inline V512 vsrl32(V512 a) { return V512{_mm512_srli_epi64(a.v, 32)}; }
static inline V512 vsll32(V512 a) { return V512{_mm512_slli_epi64(a.v, 32)}; }
static inline V512 vmul32(V512 a, V512 b) { return V512{_mm512_mul_epu32(a.v, b.v)}; }
static inline V512 vadd_if_lt(V512 r, V512 a, V512 b, V512 x) { return V512{_mm512_mask_add_epi64(r.v, _mm512_cmplt_epu64_mask(a.v, b.v), r.v, x.v)}; }
static inline V512 vsub_if_lt(V512 r, V512 a, V512 b, V512 x) { return V512{_mm512_mask_sub_epi64(r.v, _mm512_cmplt_epu64_mask(a.v, b.v), r.v, x.v)}; }
static inline void vstore(u64* dst, V512 a) { _mm512_storeu_si512((void*)dst, a.v); }

void simd512_hash_rows(const SparsePoseidon* S, const u64* rows, size_t row_stride, size_t width, Digest* out) {
    PoseidonLanes<V512>::hash_rows(*S, rows, row_stride, width, out);
}
void simd512_compress_pairs(const SparsePoseidon* S, const Digest* src, Digest* dst) { PoseidonLanes<V512>::compress_pairs(*S, src, dst); }
void simd512_test_arith(const u64* a, const u64* b, const u64* c, u64* out) { PoseidonLanes<V512>::test_arith(a, b, c, out); }
void simd512_dif_stage(u64* a, size_t n, size_t half, const u64* tw) { PoseidonLanes<V512>::dif_stage(a, n, half, tw); }
}  // namespace orc

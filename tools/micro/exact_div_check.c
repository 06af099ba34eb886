// Host check of exact_div (bflow_amd/csrc/corr_lookup_tile.hip): a / b from r = RN(1 / b) by two residual corrections equals IEEE division.
//   gcc -O2 -ffp-contract=off -o /tmp/exact_div_check tools/micro/exact_div_check.c -lm && /tmp/exact_div_check     (48 s; "mismatches 0")
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static inline float fastdiv(float a, float b, float r) {
    float q0 = a * r;
    float e1 = fmaf(-b, q0, a);
    float q1 = fmaf(e1, r, q0);
    float e2 = fmaf(-b, q1, a);
    return fmaf(e2, r, q1);
}
int main() {
    uint64_t s = 88172645463325252ULL; long bad = 0, n = 0;
    for (int sm1 = 1; sm1 <= 4096; ++sm1) {
        float b = (float)sm1, r = 1.0f / b;
        // all mantissas for one binade when sm1 small set; random otherwise
        int full = (sm1 <= 140);
        long cnt = full ? (1L << 23) : (1L << 16);
        for (long i = 0; i < cnt; ++i) {
            uint32_t m;
            if (full) m = (uint32_t)i; else { s ^= s << 13; s ^= s >> 7; s ^= s << 17; m = (uint32_t)s & 0x7fffff; }
            for (int e = 120; e <= 141; e += (full ? 21 : 3)) {
                uint32_t bits = ((uint32_t)e << 23) | m; float a; memcpy(&a, &bits, 4);
                float q = a / b, f = fastdiv(a, b, r);
                if (q != f) { if (bad < 5) printf("mismatch a=%a b=%d q=%a f=%a\n", a, sm1, q, f); ++bad; }
                ++n;
            }
        }
    }
    printf("checked %ld, mismatches %ld\n", n, bad);
    return 0;
}

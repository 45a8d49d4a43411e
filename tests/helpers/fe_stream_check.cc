// Host-side check of the small integer / packed helpers of am_fe_stream.h (their CPU twins: the forms the emulated kernels run).
// Built and run by tests/test_fe_stream_helpers.py against tests/emu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "am_fe_stream.h"

template <int D> static int check_div()
{
    // the helper's stated domain: 0 <= x < 1024, D <= 64 (thread and lane indices)
    for (int x = 0; x < 1024; ++x)
        if (fes_div_small<D>(x) != x / D) { printf("fes_div_small<%d>(%d) = %d, want %d\n", D, x, fes_div_small<D>(x), x / D); return 1; }
    return 0;
}

int main()
{
    int bad = 0;
    bad += check_div<1>() + check_div<2>() + check_div<3>() + check_div<5>() + check_div<6>() + check_div<8>() + check_div<10>() +
           check_div<12>() + check_div<15>() + check_div<16>() + check_div<24>() + check_div<48>() + check_div<63>() + check_div<64>();
    for (int a = -4096; a <= 4096; a += 7)
        for (int b = -300; b <= 300; b += 11)
            if (fes_mul24(a, b) != a * b) { printf("fes_mul24(%d, %d)\n", a, b); bad++; }
    // packed forms: the two halves are the scalar operations, nothing else
    const float xs[] = {0.0f, -0.0f, 1.0f, 3.14159274f, 1e-30f, 1e30f, 16777216.0f, 5.96046448e-8f};
    for (float a : xs) for (float b : xs) for (float c : xs) {
        const fes_f2 s = fes_pk_add(fes_mk2(a, b), fes_mk2(c, a)), m = fes_pk_mul(fes_mk2(a, b), fes_mk2(c, c));
        if (!(s.x == a + c || (s.x != s.x && (a + c) != (a + c))) || !(s.y == b + a) || !(m.x == a * c) || !(m.y == b * c)) { printf("packed\n"); bad++; }
    }
    // x + (-0) == x bit for bit (what the streaming kernels use to make "pre alone" the same expression as "suf + pre")
    for (float a : xs) { const float r = a + (-0.0f); if (__builtin_memcmp(&r, &a, 4) != 0) { printf("minus zero %g\n", a); bad++; } }
    printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad ? 1 : 0;
}

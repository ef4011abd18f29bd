// rcp_in_range / div_in_range / sqrt_in_range (rayhs_amd/csrc/short_div.h) against the compiler's own correctly rounded sequences, bit for bit, on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Irayhs_amd/csrc tools/micro/short_div.hip -o build_exp/micro/short_div && build_exp/micro/short_div
// Operands: 2^24 pseudo-random doubles per case with exponents spread over the whole range each helper is used in (and a little beyond the
// stated limits, to show where the identity ends), plus the edges (powers of two, all-ones mantissas) of those ranges.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include "short_div.h"
using namespace rayhs;
__host__ __device__ static inline uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31; return x; }
// a double with a pseudo-random sign and mantissa and a binary exponent drawn from [elo, ehi]; every 64th one an edge (mantissa all zeros / all ones)
__device__ static double draw(uint64_t i, uint64_t salt, int elo, int ehi, bool positive) {
    const uint64_t h = mix(i * 0x9e3779b97f4a7c15ULL + salt);
    uint64_t mant = h & ((1ULL << 52) - 1);
    if ((i & 63) == 0) mant = (h >> 60) & 1 ? (1ULL << 52) - 1 : 0;
    const int e = elo + (int)((h >> 52) % (uint64_t)(ehi - elo + 1));
    const uint64_t bits = ((positive ? 0ULL : (mix(h) & 1)) << 63) | ((uint64_t)(e + 1023) << 52) | mant;
    return __longlong_as_double((long long)bits);
}
__global__ void check(unsigned long long *bad, int which, int elo, int ehi, int e2lo, int e2hi) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool same = true;
    if (which == 0) { const double x = draw(i, 1, elo, ehi, false); same = __double_as_longlong(rcp_in_range(x)) == __double_as_longlong(1.0 / x); }
    else if (which == 1) { const double d = draw(i, 2, elo, ehi, false), n = draw(i, 3, e2lo, e2hi, false); same = __double_as_longlong(div_in_range(n, d)) == __double_as_longlong(n / d); }
    else { const double x = draw(i, 4, elo, ehi, true); same = __double_as_longlong(sqrt_in_range(x)) == __double_as_longlong(sqrt(x)); }
    if (!same) atomicAdd(bad, 1ULL);
}
int main() {
    unsigned long long *bad; hipMalloc(&bad, 8);
    struct { const char *what; int which, elo, ehi, e2lo, e2hi; } cases[] = {
        {"1 / x, |x| in [2^-767, 2^1021]  (the stated range; the triangle test's |det| lies in [1e-6, 1e110])", 0, -767, 1021, 0, 0},
        {"1 / x, |x| in [2^-20, 2^366]    (|det| of an accepted triangle)", 0, -20, 366, 0, 0},
        {"1 / x, |x| in [2^-200, 2^801]   (a light's d and den^2)", 0, -200, 801, 0, 0},
        {"n / d, |d| in [2^-200, 2^200], |n| in [2^-200, 2^200]  (a light's d / radius)", 1, -200, 200, -200, 200},
        {"n / d, exponents up to 760 apart, |n| >= 2^-969", 1, -380, 380, -380, 380},
        {"sqrt x, x in [2^-767, 2^1023]   (the stated range)", 2, -767, 1023, 0, 0},
        {"sqrt x, x in [2^-399, 2^399]    (a light's squared distance behind the range test)", 2, -399, 399, 0, 0},
        {"OUTSIDE: 1 / x, |x| in [2^-1022, 2^-768]  (the fix-ups matter here: mismatches expected)", 0, -1022, -768, 0, 0},
        {"OUTSIDE: sqrt x, x in [2^-1022, 2^-768]", 2, -1022, -768, 0, 0},
    };
    int rc = 0;
    for (auto &c : cases) {
        hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(check, dim3(1 << 16), dim3(256), 0, 0, bad, c.which, c.elo, c.ehi, c.e2lo, c.e2hi);
        unsigned long long h = ~0ULL;
        if (hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
        printf("%-110s %llu of %u differ\n", c.what, h, 1u << 24);
        if (h != 0 && c.what[0] != 'O') rc = 1;
    }
    return rc;
}

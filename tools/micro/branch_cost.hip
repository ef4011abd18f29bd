// Cost of a TAKEN scalar branch in a wave's instruction stream (the trace kernel executes one branch per six vector
// instructions): the same 16 v_add per "iteration", once as a loop with a back edge and a forward skip per iteration
// (2 taken branches), once with 1 taken branch, once fully unrolled (no branch).  Development aid (GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/branch_cost.hip -o /tmp/br && /tmp/br
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define BODY "v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %0\n\t" \
             "v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %0\n\t"

template <int MODE> __global__ void k(int iters, int flag, long long *cycles, float *sink) {
    float a = threadIdx.x, b = 1.0f;
    const int f = __builtin_amdgcn_readfirstlane(flag);
    long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {            // unrolled x8: one back edge per 8 bodies
        for (int i = 0; i < iters; i += 8) { asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY : "+v"(a), "+v"(b)); }
    } else if (MODE == 1) {     // one taken branch (the back edge) per body
        for (int i = 0; i < iters; i++) { asm volatile(BODY : "+v"(a), "+v"(b)); }
    } else {                    // two taken branches per body: a forward skip over code that never runs, and the back edge
        for (int i = 0; i < iters; i++) {
            asm volatile(BODY : "+v"(a), "+v"(b));
            asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 1f\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n1:" : "+v"(a), "+v"(b) : "s"(f) : "scc");
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a + b;
}

template <int MODE> static double run(int waves_per_cu) {
    const int blocks = 256 * waves_per_cu, iters = 4096;
    long long *c; float *s; (void)hipMalloc(&c, 8 * blocks); (void)hipMalloc(&s, 4 * 64 * blocks);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, iters, 0, c, s); (void)hipDeviceSynchronize(); }
    std::vector<long long> h(blocks); (void)hipMemcpy(h.data(), c, 8 * blocks, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    (void)hipFree(c); (void)hipFree(s);
    return (double)h[blocks / 2] / iters;
}
int main() {
    for (int w : {1, 4, 16}) {
        const double u = run<0>(w), one = run<1>(w), two = run<2>(w);
        printf("waves/CU %2d: cycles per body of 8 v_add: unrolled %.1f, +1 taken branch %.1f, +2 taken branches (and s_cmp) %.1f  -> a taken branch costs ~%.1f / ~%.1f cycles\n",
               w, u, one, two, one - u * 1.0, (two - one));
    }
    return 0;
}

// Does a fresh 100 MB hipMalloc delay the first kernel behind it (a one-shot `rayhs` allocates its continuation
// workspace right in front of its only frame)?  Each variant in its own process.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/alloc_first.cpp -o /tmp/af && for v in 0 1 2 3; do /tmp/af $v; done
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
__global__ void touch(double *p, size_t n, int do_touch) {
    if (!do_touch) return;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t mb = variant == 3 ? 8 : 100, bytes = mb << 20;
    int n = 0; (void)hipGetDeviceCount(&n); (void)hipSetDevice(0); (void)hipFree(nullptr);
    void *q = nullptr; char b[64] = {0};
    (void)hipMalloc(&q, 64); (void)hipMemcpy(q, b, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, 0, (double *)q, 0, 0); (void)hipDeviceSynchronize();   // queue + code object are up
    const clk::time_point t0 = clk::now();
    void *w = nullptr; (void)hipMalloc(&w, bytes);
    const clk::time_point t1 = clk::now();
    hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, 0, (double *)w, bytes / 8, variant == 1 ? 0 : 1);
    const clk::time_point t2 = clk::now();
    (void)hipDeviceSynchronize();
    const clk::time_point t3 = clk::now();
    hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, 0, (double *)w, bytes / 8, 1); (void)hipDeviceSynchronize();
    const clk::time_point t4 = clk::now();
    printf("variant %d: hipMalloc(%zu MB) %.2f ms, launch %.2f ms, first kernel (%s) done after %.2f ms, the same kernel again %.2f ms\n", variant, mb, ms(t0, t1), ms(t1, t2),
           variant == 1 ? "touches nothing" : "writes every byte", ms(t2, t3), ms(t3, t4));
    return 0;
}

// Does gfx950 execute SCALAR memory atomics (s_atomic_add), and do they count per wave?  (tools/bbprof_rewrite.py counts basic blocks with them.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/s_atomic_add.hip -o /tmp/s_atomic_add && /tmp/s_atomic_add
// Every wave adds the high word of its own area's address N times to two counters of that area, and once per loop pass to a third one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(unsigned *buf, int passes) {
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned long long p = (unsigned long long)(buf + (size_t)wave * 16);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    unsigned long long q = ((unsigned long long)hi << 32) | lo;
    for (int i = 0; i < passes; i++) {
        asm volatile("s_atomic_add %1, %0, 0x0\n\ts_atomic_add %1, %0, 0x4\n\ts_atomic_add %1, %0, 0x4" :: "s"(q), "s"(hi) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}
int main() {
    const int blocks = 1024, waves = blocks * 4, passes = 1000;
    unsigned *d; hipMalloc(&d, (size_t)waves * 16 * 4); hipMemset(d, 0, (size_t)waves * 16 * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, passes);
    if (hipDeviceSynchronize() != hipSuccess) { printf("s_atomic_add: kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<unsigned> h((size_t)waves * 16); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < waves; w++) {
        const unsigned hi = (unsigned)(((unsigned long long)(d + (size_t)w * 16)) >> 32);
        if (h[(size_t)w * 16] != hi * (unsigned)passes || h[(size_t)w * 16 + 1] != 2u * hi * (unsigned)passes) bad++;
    }
    printf("s_atomic_add on this GPU: %d of %d waves counted wrong (hi word %#x, expected %u and %u per wave)\n", bad, waves,
           (unsigned)(((unsigned long long)d) >> 32), (unsigned)(((unsigned long long)d) >> 32) * passes, 2u * (unsigned)(((unsigned long long)d) >> 32) * passes);
    return bad != 0;
}

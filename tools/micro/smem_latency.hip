// Latency of a DEPENDENT scalar load (s_load_dwordx16 of a 64-byte record whose address comes from the previous one),
// as the packet walk of the trace kernel does per BVH node: chase a random cycle through an array of `n` 64-byte
// records with 1, 4 or 16 waves per CU doing the same on their own cycles.  Development aid (GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/smem_latency.hip -o /tmp/smem && /tmp/smem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

struct alignas(64) Rec { int next; int pad[15]; };
template <class T> using cptr = const T __attribute__((address_space(4))) *;

__global__ void chase(const Rec *recs, int n, int start_stride, int steps, long long *cycles, int *sink) {
    cptr<Rec> base = (cptr<Rec>)(unsigned long long)recs;
    int cur = __builtin_amdgcn_readfirstlane((int)(((long long)blockIdx.x * start_stride) % n));  // always inside the array
    long long t0 = __builtin_amdgcn_s_memtime();
    int acc = 0;
    for (int i = 0; i < steps; i++) {
        const auto &r = base[cur];
        acc += r.pad[3] + r.pad[14];   // use the far end of the record: forces the 64-byte load to complete
        cur = r.next;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; sink[blockIdx.x] = acc + cur; }
}

int main() {
    for (int n : {256, 4096, 16384, 65536, 262144}) {          // 16 KB, 256 KB, 1 MB, 4 MB, 16 MB of records
        std::vector<int> perm(n); std::iota(perm.begin(), perm.end(), 0);
        std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<Rec> h(n);
        for (int i = 0; i < n; i++) { h[perm[i]].next = perm[(i + 1) % n]; for (int k = 0; k < 15; k++) h[perm[i]].pad[k] = k; }
        Rec *d; hipMalloc(&d, sizeof(Rec) * n); hipMemcpy(d, h.data(), sizeof(Rec) * n, hipMemcpyHostToDevice);
        for (int blocks_per_cu : {1, 4, 16}) {
            const int blocks = 256 * blocks_per_cu, steps = 2000;
            long long *c; int *s; hipMalloc(&c, 8 * blocks); hipMalloc(&s, 4 * blocks);
            for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, n, n / blocks > 0 ? n / blocks : 1, steps, c, s); hipDeviceSynchronize(); }
            std::vector<long long> hc(blocks); hipMemcpy(hc.data(), c, 8 * blocks, hipMemcpyDeviceToHost);
            std::sort(hc.begin(), hc.end());
            printf("records %7d (%6d KB)  waves/CU %2d : median %.0f cycles per dependent 64-byte scalar load\n", n, n / 16, blocks_per_cu, (double)hc[blocks / 2] / steps);
            (void)hipFree(c); (void)hipFree(s);
        }
        (void)hipFree(d);
    }
    return 0;
}

// How a fresh process gets 13 MB of scene records into HBM fastest (the one-shot `rayhs -oX scene.json` uploads the
// dragon scene once and renders one frame): pageable hipMemcpy, one copy or six, pinned staging, registered memory.
// Each variant runs in its OWN process (argv[1]) so that every one pays the runtime's first-copy set-up the same way.
// Development aid (GPU box):   hipcc -O2 tools/micro/h2d_first.cpp -o /tmp/h2d && for v in 0 1 2 3 4; do /tmp/h2d $v; done
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t sizes[6] = {281536, 2178240, 3485184, 1100000, 2178240, 3485184};  // nodes, tris, attrs, fnodes, ftris, fattrs of dragon.obj
    size_t total = 0; for (size_t s : sizes) total += s;
    std::vector<std::vector<char>> host(6);
    for (int i = 0; i < 6; i++) { host[i].assign(sizes[i], (char)i); }
    const clk::time_point t0 = clk::now();
    int n = 0; (void)hipGetDeviceCount(&n); (void)hipSetDevice(0); (void)hipFree(nullptr);
    const clk::time_point t1 = clk::now();
    void *q = nullptr; char b[64] = {0};
    (void)hipMalloc(&q, 64); (void)hipMemcpy(q, b, 64, hipMemcpyHostToDevice);   // the runtime's one-time copy set-up
    const clk::time_point t2 = clk::now();
    void *dev[6]; void *arena = nullptr;
    (void)hipMalloc(&arena, total + 6 * 256);
    size_t off = 0; for (int i = 0; i < 6; i++) { dev[i] = (char *)arena + off; off += (sizes[i] + 255) / 256 * 256; }
    const clk::time_point t3 = clk::now();
    const char *what = "";
    if (variant == 0) { what = "six pageable hipMemcpy";
        for (int i = 0; i < 6; i++) (void)hipMemcpy(dev[i], host[i].data(), sizes[i], hipMemcpyHostToDevice);
    } else if (variant == 1) { what = "one pageable hipMemcpy of a contiguous copy";
        std::vector<char> all(off); size_t o = 0;
        for (int i = 0; i < 6; i++) { memcpy(all.data() + o, host[i].data(), sizes[i]); o += (sizes[i] + 255) / 256 * 256; }
        (void)hipMemcpy(arena, all.data(), off, hipMemcpyHostToDevice);
    } else if (variant == 2) { what = "hipHostMalloc staging + one hipMemcpy";
        void *pin = nullptr; (void)hipHostMalloc(&pin, off, hipHostMallocDefault); size_t o = 0;
        for (int i = 0; i < 6; i++) { memcpy((char *)pin + o, host[i].data(), sizes[i]); o += (sizes[i] + 255) / 256 * 256; }
        (void)hipMemcpy(arena, pin, off, hipMemcpyHostToDevice);
    } else if (variant == 3) { what = "hipHostRegister each array + six hipMemcpy";
        for (int i = 0; i < 6; i++) { (void)hipHostRegister(host[i].data(), sizes[i], hipHostRegisterDefault); (void)hipMemcpy(dev[i], host[i].data(), sizes[i], hipMemcpyHostToDevice); }
    } else { what = "six hipMemcpyAsync on a stream (pageable) + stream sync";
        hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int i = 0; i < 6; i++) (void)hipMemcpyAsync(dev[i], host[i].data(), sizes[i], hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
    }
    const clk::time_point t4 = clk::now();
    (void)hipDeviceSynchronize();
    const clk::time_point t5 = clk::now();
    printf("variant %d (%s): hipInit+context %.1f ms, first tiny copy %.1f ms, arena hipMalloc %.2f ms, upload calls %.1f ms, then device idle after %.1f ms  [%zu bytes]\n",
           variant, what, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), total);
    return 0;
}

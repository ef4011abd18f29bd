// What a SIMD of gfx950 can ISSUE per cycle, by instruction class -- the roof the trace kernel is priced against
// ("Roofline"): the kernel is neither HBM- nor MFMA-bound, its waves execute about one scalar instruction
// per vector instruction, and the question is which issue resource fills first.  Every body below is 64 instructions
// of one class (or an interleaving of two), independent unless it says DEPENDENT, run by 1 .. 32 waves per CU
// (64-thread workgroups: w / 4 waves per SIMD).  Reported per column: cycles per instruction as ONE wave sees them,
// and SIMD-cycles per instruction = cycles x SIMDs / (waves x instructions) -- the price of one instruction of that
// class in the issue time of its SIMD once the SIMD is the bottleneck (it converges as waves are added).
// s_memtime ticks are calibrated against s_memrealtime (100 MHz) in the first line.
// Development aid (GPU box):   hipcc --offload-arch=gfx950 -O3 tools/micro/issue_rates.hip -o /tmp/ir && /tmp/ir
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

// operands: %0-%3 doubles a..d, %4-%7 floats, %8-%11 32-bit SGPRs, %12 64-bit SGPR pair, %13 table pointer (SGPR pair), %14 LDS byte address (VGPR), %15 double constant
struct Mode { const char *name; int insts; };
#define MODES(X) \
    X(0, "s_add_u32", 64, R16("s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 1\n s_add_u32 %10, %10, 1\n s_add_u32 %11, %11, 1\n")) \
    X(1, "s_and_b64 (mask logic)", 64, R16("s_and_b64 %12, %12, exec\n s_or_b64 %12, %12, exec\n s_andn2_b64 %12, %12, vcc\n s_and_b64 %12, %12, exec\n")) \
    X(2, "s_mov_b32 literal", 64, R16("s_mov_b32 %8, 0x48e8a71e\n s_mov_b32 %9, 0x3fefff2e\n s_mov_b32 %10, 0x48e8a71f\n s_mov_b32 %11, 0x3fefff2f\n")) \
    X(3, "s_waitcnt lgkmcnt(0), nothing outstanding", 64, R16("s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n")) \
    X(4, "s_cbranch_scc1 NOT taken (+ s_cmp)", 64, R16("s_cmp_eq_u32 %8, 0\n s_cbranch_scc1 9f\n s_cmp_eq_u32 %8, 0\n s_cbranch_scc1 9f\n")) \
    X(5, "s_branch TAKEN (+ v_add_f32)", 64, R16("v_add_f32 %4, %4, %4\n s_branch 1f\n s_nop 0\n1:\n v_add_f32 %5, %5, %5\n s_branch 2f\n s_nop 0\n2:\n")) \
    X(6, "v_add_f32", 64, R16("v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7\n")) \
    X(7, "v_pk_fma_f32", 64, R16("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %3, %3, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %3, %3, %1, %2\n")) \
    X(8, "v_add_f64", 64, R16("v_add_f64 %0, %0, %15\n v_add_f64 %1, %1, %15\n v_add_f64 %2, %2, %15\n v_add_f64 %3, %3, %15\n")) \
    X(9, "v_mul_f64", 64, R16("v_mul_f64 %0, %0, %15\n v_mul_f64 %1, %1, %15\n v_mul_f64 %2, %2, %15\n v_mul_f64 %3, %3, %15\n")) \
    X(10, "v_fma_f64", 64, R16("v_fma_f64 %0, %0, %15, %15\n v_fma_f64 %1, %1, %15, %15\n v_fma_f64 %2, %2, %15, %15\n v_fma_f64 %3, %3, %15, %15\n")) \
    X(11, "v_rcp_f64", 64, R16("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n")) \
    X(12, "v_cmp_gt_f64 -> vcc", 64, R16("v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %1, %2\n v_cmp_gt_f64 vcc, %2, %3\n v_cmp_gt_f64 vcc, %3, %0\n")) \
    X(13, "v_cmp_gt_f64 -> SGPR pair", 64, R16("v_cmp_gt_f64 %12, %0, %1\n v_cmp_gt_f64 %12, %1, %2\n v_cmp_gt_f64 %12, %2, %3\n v_cmp_gt_f64 %12, %3, %0\n")) \
    X(14, "v_cndmask_b32 (vcc)", 64, R16("v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %4, vcc\n")) \
    X(15, "v_readfirstlane_b32", 64, R16("v_readfirstlane_b32 %8, %4\n v_readfirstlane_b32 %9, %5\n v_readfirstlane_b32 %10, %6\n v_readfirstlane_b32 %11, %7\n")) \
    X(16, "v_mul_lo_u32", 64, R16("v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %4\n")) \
    X(17, "s_add_u32 + v_add_f32 interleaved", 64, R16("s_add_u32 %8, %8, 1\n v_add_f32 %4, %4, %4\n s_add_u32 %9, %9, 1\n v_add_f32 %5, %5, %5\n")) \
    X(18, "s_add_u32 + v_add_f64 interleaved", 64, R16("s_add_u32 %8, %8, 1\n v_add_f64 %0, %0, %15\n s_add_u32 %9, %9, 1\n v_add_f64 %1, %1, %15\n")) \
    X(19, "v_add_f32 + v_add_f64 interleaved", 64, R16("v_add_f32 %4, %4, %4\n v_add_f64 %0, %0, %15\n v_add_f32 %5, %5, %5\n v_add_f64 %1, %1, %15\n")) \
    X(20, "s_load_dwordx2 (one line) + s_waitcnt per 4", 80, R16("s_load_dwordx2 %12, %13, 0x0\n s_load_dwordx2 %12, %13, 0x8\n s_load_dwordx2 %12, %13, 0x10\n s_load_dwordx2 %12, %13, 0x18\n s_waitcnt lgkmcnt(0)\n")) \
    X(21, "ds_read_b64 + s_waitcnt per 4", 80, R16("ds_read_b64 %0, %14\n ds_read_b64 %1, %14 offset:2048\n ds_read_b64 %2, %14 offset:4096\n ds_read_b64 %3, %14 offset:6144\n s_waitcnt lgkmcnt(0)\n")) \
    X(22, "ds_write_b64 + s_waitcnt per 4", 80, R16("ds_write_b64 %14, %0\n ds_write_b64 %14, %1 offset:2048\n ds_write_b64 %14, %2 offset:4096\n ds_write_b64 %14, %3 offset:6144\n s_waitcnt lgkmcnt(0)\n")) \
    X(23, "v_add_f64 DEPENDENT", 64, R16("v_add_f64 %0, %0, %15\n v_add_f64 %0, %0, %15\n v_add_f64 %0, %0, %15\n v_add_f64 %0, %0, %15\n")) \
    X(24, "v_add_f32 DEPENDENT", 64, R16("v_add_f32 %4, %4, %4\n v_add_f32 %4, %4, %4\n v_add_f32 %4, %4, %4\n v_add_f32 %4, %4, %4\n")) \
    X(25, "s_add_u32 DEPENDENT", 64, R16("s_add_u32 %8, %8, 1\n s_add_u32 %8, %8, 1\n s_add_u32 %8, %8, 1\n s_add_u32 %8, %8, 1\n")) \
    X(26, "v_cmp_gt_f32 -> s_and_b64 -> v_cndmask DEPENDENT", 48, R16("v_cmp_gt_f32 %12, %4, %5\n s_and_b64 %12, %12, exec\n v_cndmask_b32 %4, %4, %4, %12\n")) \
    X(27, "v_rcp_f64 DEPENDENT", 16, R16("v_rcp_f64 %0, %0\n")) \
    X(28, "v_cndmask_b32 e32 (vcc), independent", 64, R16("v_cndmask_b32 %4, %5, %6, vcc\n v_cndmask_b32 %4, %5, %6, vcc\n v_cndmask_b32 %7, %5, %6, vcc\n v_cndmask_b32 %7, %5, %6, vcc\n")) \
    X(29, "v_cndmask_b32 e64 (SGPR pair), independent", 64, R16("v_cndmask_b32 %4, %5, %6, %12\n v_cndmask_b32 %4, %5, %6, %12\n v_cndmask_b32 %7, %5, %6, %12\n v_cndmask_b32 %7, %5, %6, %12\n")) \
    X(30, "v_bfi_b32", 64, R16("v_bfi_b32 %4, %5, %6, %7\n v_bfi_b32 %4, %5, %6, %7\n v_bfi_b32 %4, %5, %6, %7\n v_bfi_b32 %4, %5, %6, %7\n")) \
    X(31, "v_mov_b32", 64, R16("v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n")) \
    X(32, "v_add_u32 / v_and_b32 / v_lshlrev_b32", 64, R16("v_add_u32 %4, %5, %6\n v_and_b32 %7, %5, %6\n v_lshlrev_b32 %4, 3, %5\n v_add_u32 %7, %5, %6\n")) \
    X(33, "v_fma_f32 / v_mul_f32 / v_max_f32 / v_min_f32", 64, R16("v_fma_f32 %4, %5, %6, %7\n v_mul_f32 %4, %5, %6\n v_max_f32 %4, %5, %6\n v_min_f32 %4, %5, %6\n")) \
    X(34, "v_cmp_gt_f32 -> vcc", 64, R16("v_cmp_gt_f32 vcc, %4, %5\n v_cmp_gt_f32 vcc, %5, %6\n v_cmp_gt_f32 vcc, %6, %7\n v_cmp_gt_f32 vcc, %7, %4\n")) \
    X(35, "v_cmp_gt_f32 -> SGPR pair", 64, R16("v_cmp_gt_f32 %12, %4, %5\n v_cmp_gt_f32 %12, %5, %6\n v_cmp_gt_f32 %12, %6, %7\n v_cmp_gt_f32 %12, %7, %4\n")) \
    X(36, "v_cvt_f32_f64 / v_cvt_f64_f32", 64, R16("v_cvt_f32_f64 %4, %0\n v_cvt_f64_f32 %1, %5\n v_cvt_f32_f64 %6, %2\n v_cvt_f64_f32 %3, %7\n")) \
    X(37, "v_readlane_b32 (SGPR lane index)", 64, R16("v_readlane_b32 %8, %4, %11\n v_readlane_b32 %9, %5, %11\n v_readlane_b32 %10, %6, %11\n v_readlane_b32 %8, %7, %11\n")) \
    X(38, "s_bcnt1_i32_b64 / s_ff1_i32_b64", 64, R16("s_bcnt1_i32_b64 %8, %12\n s_ff1_i32_b64 %9, %12\n s_bcnt1_i32_b64 %10, %12\n s_ff1_i32_b64 %11, %12\n")) \
    X(39, "s_cmp_lg_u64 + s_cselect_b64", 64, R16("s_cmp_lg_u64 %12, 0\n s_cselect_b64 %12, %12, exec\n s_cmp_lg_u64 %12, 0\n s_cselect_b64 %12, %12, exec\n")) \
    X(40, "s_and_saveexec_b64 + s_mov_b64 exec (restore)", 64, R16("s_and_saveexec_b64 %12, vcc\n s_mov_b64 exec, %12\n s_and_saveexec_b64 %12, vcc\n s_mov_b64 exec, %12\n")) \
    X(41, "mix: v_cmp_f64->SGPR, s_and, s_andn2, v_cndmask e64 x2", 80, R16("v_cmp_gt_f64 %12, %0, %1\n s_and_b64 %12, %12, exec\n s_andn2_b64 vcc, exec, %12\n v_cndmask_b32 %4, %5, %6, %12\n v_cndmask_b32 %7, %5, %6, %12\n")) \
    X(42, "mix: v_mul_f64, v_add_f64, v_fma_f32, s_and (4 independent)", 64, R16("v_mul_f64 %0, %1, %15\n v_fma_f32 %4, %5, %6, %7\n v_add_f64 %2, %3, %15\n s_and_b64 %12, %12, exec\n")) \
    X(43, "v_mul_f64 then dependent v_add_f64 (pairs)", 64, R16("v_mul_f64 %0, %1, %15\n v_add_f64 %0, %0, %15\n v_mul_f64 %2, %3, %15\n v_add_f64 %2, %2, %15\n")) \
    X(44, "v_cndmask_b32_e64 with vcc as the SGPR pair", 64, R16("v_cndmask_b32_e64 %4, %5, %6, vcc\n v_cndmask_b32_e64 %4, %5, %6, vcc\n v_cndmask_b32_e64 %7, %5, %6, vcc\n v_cndmask_b32_e64 %7, %5, %6, vcc\n")) \
    X(45, "v_cndmask e32 vcc, 1 in 4 (3 v_fma_f32 between)", 64, R16("v_cndmask_b32 %4, %5, %6, vcc\n v_fma_f32 %7, %5, %6, %7\n v_fma_f32 %7, %5, %6, %7\n v_fma_f32 %7, %5, %6, %7\n") ) \
    X(46, "v_cndmask e64 sgpr, 1 in 4 (3 v_fma_f32 between)", 64, R16("v_cndmask_b32 %4, %5, %6, %12\n v_fma_f32 %7, %5, %6, %7\n v_fma_f32 %7, %5, %6, %7\n v_fma_f32 %7, %5, %6, %7\n") ) \
    X(47, "v_addc_co_u32 (vcc in and out)", 64, R16("v_addc_co_u32 %4, vcc, %5, %6, vcc\n v_addc_co_u32 %7, vcc, %5, %6, vcc\n v_addc_co_u32 %4, vcc, %5, %6, vcc\n v_addc_co_u32 %7, vcc, %5, %6, vcc\n")) \
    X(48, "s_nop 0", 64, R16("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")) \
    X(49, "v_mov_b64", 64, R16("v_mov_b64 %0, %1\n v_mov_b64 %2, %3\n v_mov_b64 %0, %1\n v_mov_b64 %2, %3\n"))

#define X_NAME(i, name, n, body) {name, n},
static const Mode kModes[] = {MODES(X_NAME)};
constexpr int kNumModes = sizeof(kModes) / sizeof(kModes[0]);

template <int MODE> __global__ void k(int iters, const int *table, long long *cycles, long long *real, double *sink) {
    __shared__ double lds[1024];
    double a = threadIdx.x + 1.5, b = 1.25, c = 0.75, d = 2.5;
    const double e = 1.0000001;
    float fa = threadIdx.x, fb = 1, fc = 2, fd = 3;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    unsigned long long m = 5;
    const unsigned la = threadIdx.x * 8;
    lds[threadIdx.x] = 1.0; lds[threadIdx.x + 256] = 2.0; lds[threadIdx.x + 512] = 2.0; lds[threadIdx.x + 768] = 2.0;
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#define X_BODY(i_, name, n, body) \
        if (MODE == i_) asm volatile(body "9:\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(m) \
                                     : "s"(table), "v"(la), "v"(e) : "scc", "vcc", "memory");
        MODES(X_BODY)
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
    sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + fa + fb + fc + fd + (double)(s0 + s1 + s2 + s3) + (double)m + lds[threadIdx.x];
}

static double g_ticks_per_10ns = 0;
template <int MODE> static void run() {
    const int iters = 512;
    printf("%-52s", kModes[MODE].name);
    for (int w : {1, 4, 8, 16, 20, 24, 32}) {
        const int blocks = 256 * w;
        long long *c, *r; double *s; int *t;
        (void)hipMalloc(&c, 8 * blocks); (void)hipMalloc(&r, 8 * blocks); (void)hipMalloc(&s, 8 * 64 * blocks); (void)hipMalloc(&t, 256); (void)hipMemset(t, 0, 256);
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, iters, t, c, r, s); (void)hipDeviceSynchronize(); }
        std::vector<long long> h(blocks), hr(blocks);
        (void)hipMemcpy(h.data(), c, 8 * blocks, hipMemcpyDeviceToHost); (void)hipMemcpy(hr.data(), r, 8 * blocks, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end()); std::sort(hr.begin(), hr.end());
        (void)hipFree(c); (void)hipFree(r); (void)hipFree(s); (void)hipFree(t);
        const double per_body = (double)h[blocks / 2] / iters;
        if (w == 16 && MODE == 6) g_ticks_per_10ns = (double)h[blocks / 2] / (double)hr[blocks / 2];
        printf(" | w%-2d %6.2f %5.2f", w, per_body / kModes[MODE].insts, per_body * 4.0 / ((double)w * kModes[MODE].insts));
    }
    printf("\n");
}
template <int M> static void run_all() { run<M>(); if constexpr (M + 1 < kNumModes) run_all<M + 1>(); }
int main() {
    printf("per column: waves per CU | s_memtime ticks per instruction seen by one wave, SIMD-ticks per instruction (= ticks x 4 SIMDs / (waves x instructions))\n");
    run_all<0>();
    printf("s_memtime ticks per 10 ns (s_memrealtime tick) during the v_add_f32 run at 16 waves/CU: %.3f  -> tick rate %.3f GHz\n", g_ticks_per_10ns, g_ticks_per_10ns / 10.0);
    return 0;
}

// What a SIMD of gfx950 can ISSUE per cycle, by instruction class -- the calibration behind the roofline of the trace kernel
// ("Roofline").  Every body below is 64 instructions of one class (or an interleaving of two), independent unless
// it says DEPENDENT, run by w = 1 .. 32 waves per CU on every CU of the chip, in workgroups of 64 or of 256 threads.
//
// Round 5: NOTHING IS ASSUMED ABOUT RESIDENCY ANY MORE.  Round 4's table divided a wave's time by "w / 4 waves per SIMD" -- and
// its kernel declared 8 KB of LDS per 64-thread workgroup, which caps a CU at 20 of them, so the w24 / w32 columns described
// fewer co-resident waves than their headers said.  Now every wave records where it ran (HW_REG_HW_ID: SE, SH, CU, SIMD, wave
// slot; HW_REG_XCC_ID) and when (s_memrealtime at both ends, s_memtime at both ends); LDS is only allocated by the two LDS
// modes (2 KB per wave).  From those records, per column:
//   t/inst   s_memtime ticks per instruction as ONE wave sees them (median over waves)
//   res      waves co-resident on a SIMD, MEASURED: sum of the wave lifetimes on that SIMD / the SIMD's busy span (median over SIMDs)
//   S/inst   SIMD-ticks per wave-instruction = the SIMD's busy span / instructions executed on it (median over SIMDs): the price
//            of one wave64 instruction of that class in the issue time of its SIMD, with no residency assumption in it
// and per run the occupancy the runtime reports (hipOccupancyMaxActiveBlocksPerMultiprocessor), the number of distinct SIMDs seen
// (1024 = 256 CUs x 4) and the s_memtime tick rate against s_memrealtime (100 MHz).  The last block prints the vector peaks the
// measured S/inst imply next to the spec sheet's (157.3 TF fp32, 78.6 TF fp64: MI355X_MICROARCH.md).
// Development aid (GPU box):   hipcc --offload-arch=gfx950 -O3 tools/micro/issue_rates.hip -o /tmp/ir && /tmp/ir
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
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
    X(21, "ds_read_b64 + s_waitcnt per 4", 80, R16("ds_read_b64 %0, %14\n ds_read_b64 %1, %14 offset:512\n ds_read_b64 %2, %14 offset:1024\n ds_read_b64 %3, %14 offset:1536\n s_waitcnt lgkmcnt(0)\n")) \
    X(22, "ds_write_b64 + s_waitcnt per 4", 80, R16("ds_write_b64 %14, %0\n ds_write_b64 %14, %1 offset:512\n ds_write_b64 %14, %2 offset:1024\n ds_write_b64 %14, %3 offset:1536\n s_waitcnt lgkmcnt(0)\n")) \
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

struct WaveRec { unsigned long long where, rt0, rt1, t0, t1; };

template <int MODE> __global__ void k(int iters, const int *table, WaveRec *rec, double *sink) {
    extern __shared__ double lds[];   // 2 KB per wave, allocated for the two LDS modes only
    double a = threadIdx.x + 1.5, b = 1.25, c = 0.75, d = 2.5;
    const double e = 1.0000001;
    float fa = threadIdx.x, fb = 1, fc = 2, fd = 3;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    unsigned long long m = 5;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned la = wave * 2048 + lane * 8;
    if (MODE == 21 || MODE == 22) { for (int q = 0; q < 4; q++) lds[wave * 256 + q * 64 + lane] = 1.0 + q; }
    __syncthreads();
    unsigned hw_id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
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
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (lane == 0) { rec[w].where = ((unsigned long long)xcc << 32) | hw_id; rec[w].rt0 = r0; rec[w].rt1 = r1; rec[w].t0 = t0; rec[w].t1 = t1; }
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + fa + fb + fc + fd + (double)(s0 + s1 + s2 + s3) + (double)m + ((MODE == 21 || MODE == 22) ? lds[wave * 256 + lane] : 0.0);
}

static double median(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
struct Result { double t_inst, res, s_inst, ghz; int simds, occ; };
static Result g_res[kNumModes][2][7];
static const int kW[7] = {1, 4, 8, 16, 20, 24, 32};

template <int MODE> static Result run_one(int block, int w) {
    const int iters = 512, waves_per_block = block / 64;
    const int n_cu = 256, blocks = n_cu * w / waves_per_block > 0 ? n_cu * w / waves_per_block : 1;
    const size_t n_waves = (size_t)blocks * waves_per_block;
    const size_t lds = (MODE == 21 || MODE == 22) ? (size_t)waves_per_block * 2048 : 0;
    Result r{};
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&r.occ, k<MODE>, block, lds);
    WaveRec *rec; double *s; int *t;
    (void)hipMalloc(&rec, sizeof(WaveRec) * n_waves); (void)hipMalloc(&s, 8 * (size_t)blocks * block); (void)hipMalloc(&t, 256); (void)hipMemset(t, 0, 256);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), lds, 0, iters, t, rec, s); (void)hipDeviceSynchronize(); }
    std::vector<WaveRec> h(n_waves);
    (void)hipMemcpy(h.data(), rec, sizeof(WaveRec) * n_waves, hipMemcpyDeviceToHost);
    (void)hipFree(rec); (void)hipFree(s); (void)hipFree(t);
    const double insts = (double)iters * kModes[MODE].insts;
    std::vector<double> per_wave, ghz;
    // a SIMD = (XCC, SE, SH, CU, SIMD) of HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]
    struct Simd { double life = 0, lo = 1e300, hi = 0, n = 0, lo_t = 1e300, hi_t = 0; };
    std::map<unsigned long long, Simd> simds;
    for (const WaveRec &q : h) {
        per_wave.push_back((double)(q.t1 - q.t0) / insts);
        ghz.push_back((double)(q.t1 - q.t0) / ((double)(q.rt1 - q.rt0) * 10.0));
        const unsigned hw = (unsigned)q.where;
        const unsigned long long key = (q.where >> 32 << 32) | (hw & 0xfff0u & ~0xc0u) | ((hw >> 13 & 7u) << 16);   // drops the wave slot and the pipe id
        Simd &sd = simds[key];
        sd.life += (double)(q.rt1 - q.rt0); sd.lo = std::min(sd.lo, (double)q.rt0); sd.hi = std::max(sd.hi, (double)q.rt1); sd.n += 1;
    }
    r.t_inst = median(per_wave); r.ghz = median(ghz); r.simds = (int)simds.size();
    std::vector<double> res, s_inst;
    for (auto &kv : simds) {
        const Simd &sd = kv.second;
        const double span = sd.hi - sd.lo;   // 10 ns units
        if (span <= 0) continue;
        res.push_back(sd.life / span);
        s_inst.push_back(span * 10.0 * r.ghz / (sd.n * insts));   // SIMD-ticks (at the measured tick rate) per wave-instruction executed on this SIMD
    }
    r.res = median(res); r.s_inst = median(s_inst);
    return r;
}
template <int MODE> static void run() {
    for (int bi = 0; bi < 2; bi++) {
        const int block = bi == 0 ? 64 : 256;
        printf("%-52s B%-3d", kModes[MODE].name, block);
        for (int wi = 0; wi < 7; wi++) {
            const int w = kW[wi];
            if (block == 256 && w < 4) { printf(" | w%-2d      -     -      -", w); continue; }
            const Result r = run_one<MODE>(block, w);
            g_res[MODE][bi][wi] = r;
            printf(" | w%-2d %6.2f %5.2f %6.2f", w, r.t_inst, r.res, r.s_inst);
        }
        const Result &last = g_res[MODE][bi][6];
        printf(" | occ %d blocks/CU, %d SIMDs seen, tick %.3f GHz\n", last.occ, last.simds, last.ghz);
    }
}
template <int M> static void run_all() { run<M>(); if constexpr (M + 1 < kNumModes) run_all<M + 1>(); }
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    printf("device: %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    printf("per column (waves per CU requested): t/inst = s_memtime ticks per instruction seen by one wave | res = waves co-resident per SIMD, MEASURED | "
           "S/inst = SIMD-ticks per wave-instruction, MEASURED (SIMD busy span / instructions executed on it)\n");
    run_all<0>();
    // what the measured SIMD throughput implies for the vector peaks (best column of each class, 256-thread workgroups)
    auto best = [](int mode) { double b = 1e300; for (int bi = 0; bi < 2; bi++) for (int wi = 0; wi < 7; wi++) { const Result &r = g_res[mode][bi][wi]; if (r.s_inst > 0) b = std::min(b, r.s_inst); } return b; };
    const double ghz = 2.4;
    printf("\nimplied vector peaks at %.1f GHz, 1024 SIMDs, 64 lanes, 2 flops per fused multiply-add (spec: 157.3 TF fp32, 78.6 TF fp64):\n", ghz);
    printf("  v_fma_f32 (mode 33 mix): best S/inst %.2f -> %.1f TF   v_pk_fma_f32: %.2f -> %.1f TF   v_fma_f64: %.2f -> %.1f TF   v_rcp_f64: %.2f SIMD-ticks\n",
           best(33), 1024 * 128 * ghz / best(33) / 1e3, best(7), 1024 * 256 * ghz / best(7) / 1e3, best(10), 1024 * 128 * ghz / best(10) / 1e3, best(11));
    printf("  scalar ALU (s_add_u32): %.2f   s_nop: %.2f   v_cndmask e64: %.2f   v_cmp_f64 -> SGPR: %.2f   mixed (mode 42): %.2f   v_add_f32 + s_add_u32: %.2f\n",
           best(0), best(48), best(29), best(13), best(42), best(17));
    return 0;
}

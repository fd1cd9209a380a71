// valu_rates3.hip -- round 2: issue cost of the integer / SDWA / bit-field instructions considered for the merge
// kernel's index path, and whether v_exp_f32 overlaps with plain VALU work of other waves.  Same method as
// valu_rates2.hip: inline asm, wave64, 8 waves per SIMD resident, independent destinations, 8 instructions per
// loop iteration; cycles are per wave-instruction per SIMD at the nominal 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_rates3.hip -o tools/valu_rates3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITERS = 2048;
#define REP8(S) S S S S S S S S
#define REP4(S) S S S S

template <int OP> __global__ __launch_bounds__(256) void k(float *out, float seed, unsigned long long *stamps)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0s = __builtin_amdgcn_s_memrealtime();
    float a = seed + threadIdx.x * 0.001f, b = 1.0000001f, c = 1e-9f, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    unsigned u = __float_as_uint(a) & 0xffffff, m = 16711936u, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (OP == 0) { REP8(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));) }
        if constexpr (OP == 1) { REP8(asm volatile("v_and_b32 %0, %1, %2" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 2) { REP8(asm volatile("v_or_b32 %0, %1, %2" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 3) { REP8(asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(w0) : "v"(u));) }
        if constexpr (OP == 4) { REP8(asm volatile("v_add_u32 %0, %1, %2" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 5) { REP8(asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 6) { REP8(asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 7) { REP8(asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(w0) : "v"(u), "v"(m), "v"(u));) }
        if constexpr (OP == 8) { REP8(asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(w0) : "v"(u));) }
        if constexpr (OP == 9) { REP8(asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(w0) : "v"(u), "v"(m), "v"(u));) }
        if constexpr (OP == 10) { REP8(asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 11) { REP8(asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w0) : "v"(u), "v"(m), "v"(u));) }
        if constexpr (OP == 12) { REP8(asm volatile("v_mul_hi_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 13) { REP8(asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(w0) : "v"(m), "v"(u));) }
        if constexpr (OP == 14) { REP8(asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r1) : "v"(u));) }
        if constexpr (OP == 15) { REP8(asm volatile("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(w1) : "v"(u), "v"(u));) }
        if constexpr (OP == 16) { REP8(asm volatile("v_lshlrev_b32 %0, 4, %1" : "=v"(w1) : "v"(u));) }
        if constexpr (OP == 17) { REP8(asm volatile("v_add_f32 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(b));) }
        if constexpr (OP == 18) { REP8(asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(b));) }
        if constexpr (OP == 19) { REP8(asm volatile("v_mac_f32 %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));) }
        if constexpr (OP == 20) { REP8(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));) }
        if constexpr (OP == 21) { REP8(asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(w0) : "v"(a));) }
        if constexpr (OP == 22) { REP8(asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(w0) : "v"(a));) }
        if constexpr (OP == 23) { REP8(asm volatile("v_mov_b32 %0, %1" : "=v"(w0) : "v"(u));) }
        if constexpr (OP == 24) { REP8(asm volatile("v_exp_f32 %0, %1" : "=v"(r3) : "v"(a));) }
        if constexpr (OP == 25) {  // exp and fma alternating inside one wave: 4 + 4
            REP4(asm volatile("v_exp_f32 %0, %1" : "=v"(r3) : "v"(a)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));)
        }
        if constexpr (OP == 26) {  // exp in even waves, fma in odd waves (same SIMD hosts both kinds)
            if ((threadIdx.x >> 6) & 1) { REP8(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));) }
            else { REP8(asm volatile("v_exp_f32 %0, %1" : "=v"(r3) : "v"(a));) }
        }
        if constexpr (OP == 27) {  // 1 exp + 7 fma (the merge loop's ratio is ~1 : 14)
            asm volatile("v_exp_f32 %0, %1" : "=v"(r3) : "v"(a));
            REP4(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));)
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r2) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a), "v"(b), "v"(c));
        }
        if constexpr (OP == 28) {  // rounding-mode switch around 4 fma (index trick): 2 s_setreg + 4 fma + 4 fma
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 1");
            REP4(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));)
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
            REP4(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a), "v"(b), "v"(c));)
        }
        if constexpr (OP == 29) { REP8(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double *)&r0) : "v"(*(double *)&a), "v"(*(double *)&b));) }
        if constexpr (OP == 30) { REP8(asm volatile("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(b));) }
        if constexpr (OP == 31) { REP8(asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r1) : "v"(u));) }
        if constexpr (OP == 32) { REP8(asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(b));) }
        if constexpr (OP == 33) { REP8(asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(u));) }
        if constexpr (OP == 34) { REP8(asm volatile("v_sub_u32 %0, %1, %2" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 35) { REP8(asm volatile("v_xor_b32 %0, %1, %2" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 36) { REP8(asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(r0) : "v"(a), "v"(b));) }
        if constexpr (OP == 37) { REP8(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(w0) : "v"(u), "v"(m));) }
        if constexpr (OP == 38) { REP8(asm volatile("v_rcp_f32 %0, %1" : "=v"(r3) : "v"(a));) }
        if constexpr (OP == 39) { REP8(asm volatile("v_rsq_f32 %0, %1" : "=v"(r3) : "v"(a));) }
        if constexpr (OP == 40) { REP8(asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));) }
        if constexpr (OP == 41) { REP8(asm volatile("v_fma_f32 %0, %1, s4, %2" : "=v"(r0) : "v"(a), "v"(c) : "s4");) }
        if constexpr (OP == 42) { REP8(asm volatile("v_mul_f32 %0, 0x3f800001, %1" : "=v"(r0) : "v"(a));) }
        if constexpr (OP == 43) { REP8(asm volatile("v_fmaak_f32 %0, %1, %2, 0x3f800001" : "=v"(r0) : "v"(a), "v"(b));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + w0 + w1 + w2 + w3;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1s = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1s - r0s; }
}

#include <vector>
#include <algorithm>
static unsigned long long *g_stamps;
template <int OP> void run(const char *name, double per_iter, float *out)
{
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, g_stamps);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, g_stamps);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double wave_insts = (double)blocks * 4 * ITERS * per_iter;
    const double per_simd_per_us = wave_insts / 1024.0 / (ms * 1e3);
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int b = 0; b < blocks; ++b) { clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); cyc.push_back((double)h[2 * b]); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    // a block's 4 waves sit on 4 SIMDs; 8 blocks per CU -> 8 waves per SIMD share the issue port
    const double cyc_per = cyc[blocks / 2] / ((double)ITERS * per_iter * 8.0);
    printf("%-44s %.3f ms  => %.2f nominal cycles (2.4 GHz) | in-kernel clock %.2f GHz, %.2f shader cycles per wave-instr per SIMD\n",
           name, ms, 2400.0 / per_simd_per_us, clk[blocks / 2], cyc_per);
    fflush(stdout);
}

int main()
{
    float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CK(hipMalloc(&g_stamps, 256 * 8 * 2 * 8));
    run<0>("v_fma_f32", 8, out);
    run<32>("v_mul_f32", 8, out);
    run<17>("v_add_f32", 8, out);
    run<18>("v_sub_f32", 8, out);
    run<20>("v_fmac_f32 (VOP2, accumulate)", 8, out);
    run<30>("v_max_f32", 8, out);
    run<40>("v_fma_f32 with neg modifier", 8, out);
    run<41>("v_fma_f32 with SGPR operand", 8, out);
    run<42>("v_mul_f32 with literal", 8, out);
    run<43>("v_fmaak_f32 (literal addend)", 8, out);
    run<36>("v_mul_f32_sdwa (DWORD sel)", 8, out);
    run<29>("v_pk_mul_f32 (2 values)", 8, out);
    run<33>("v_ldexp_f32", 8, out);
    run<23>("v_mov_b32", 8, out);
    run<1>("v_and_b32", 8, out);
    run<2>("v_or_b32", 8, out);
    run<35>("v_xor_b32", 8, out);
    run<3>("v_lshrrev_b32", 8, out);
    run<16>("v_lshlrev_b32", 8, out);
    run<4>("v_add_u32", 8, out);
    run<34>("v_sub_u32", 8, out);
    run<5>("v_mul_u32_u24", 8, out);
    run<6>("v_mul_hi_u32_u24", 8, out);
    run<12>("v_mul_hi_u32_u24_sdwa WORD_1", 8, out);
    run<7>("v_mad_u32_u24", 8, out);
    run<37>("v_mul_lo_u32", 8, out);
    run<8>("v_bfe_u32", 8, out);
    run<9>("v_and_or_b32", 8, out);
    run<10>("v_alignbit_b32", 8, out);
    run<11>("v_perm_b32", 8, out);
    run<15>("v_lshl_add_u32", 8, out);
    run<13>("v_or_b32_sdwa WORD_1", 8, out);
    run<14>("v_cvt_f32_u32_sdwa WORD_1", 8, out);
    run<31>("v_cvt_f32_ubyte1", 8, out);
    run<21>("v_cvt_flr_i32_f32", 8, out);
    run<22>("v_cvt_u32_f32", 8, out);
    run<24>("v_exp_f32", 8, out);
    run<38>("v_rcp_f32", 8, out);
    run<39>("v_rsq_f32", 8, out);
    run<25>("exp+fma alternating in a wave (4+4)", 8, out);
    run<26>("exp in even waves, fma in odd waves", 8, out);
    run<27>("1 exp + 7 fma", 8, out);
    run<28>("s_setreg x2 + 8 fma (count 8)", 8, out);
    return 0;
}

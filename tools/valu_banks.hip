// valu_banks.hip -- does the VGPR bank of the source operands change the issue cost of f32 FMA / MUL on gfx950?
// (Hypothesis for the ~1.28x gap between the merge loop's time and the sum of its instructions' microbenchmarked costs.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_banks.hip -o tools/valu_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITERS = 2048;
#define REP8(S) S S S S S S S S
#define CLOB : : : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23"

#define CLOB2 : : : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51"

template <int OP> __global__ __launch_bounds__(256) void k(float *out, float seed)
{
    asm volatile("v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n"
                 "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n"
                 : : "v"(seed) : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (OP == 0) { REP8(asm volatile("v_fma_f32 v20, v4, v9, v14" CLOB);) }    // banks 0,1,2
        if constexpr (OP == 1) { REP8(asm volatile("v_fma_f32 v20, v4, v8, v12" CLOB);) }    // all bank 0
        if constexpr (OP == 2) { REP8(asm volatile("v_fma_f32 v20, v4, v8, v13" CLOB);) }    // two in bank 0
        if constexpr (OP == 3) { REP8(asm volatile("v_fma_f32 v20, v4, v4, v5" CLOB);) }     // same register twice
        if constexpr (OP == 4) { REP8(asm volatile("v_mul_f32 v20, v4, v9" CLOB);) }
        if constexpr (OP == 5) { REP8(asm volatile("v_mul_f32 v20, v4, v8" CLOB);) }
        if constexpr (OP == 6) { REP8(asm volatile("v_fmac_f32 v20, v4, v9" CLOB);) }        // dest = addend: v20 (bank 0) + v4 (0) + v9 (1)
        if constexpr (OP == 7) { REP8(asm volatile("v_fmac_f32 v21, v4, v10" CLOB);) }       // banks 1,0,2
        if constexpr (OP == 8) { REP8(asm volatile("v_fmac_f32 v20, v4, v8" CLOB);) }        // all bank 0
        if constexpr (OP == 9) {   // dependent chain in one wave: each fma consumes the previous result
            REP8(asm volatile("v_fma_f32 v20, v20, v9, v14" CLOB);)
        }
        if constexpr (OP == 10) {  // the merge loop's per-element sequence, one element, register operands spread over banks
            asm volatile(
                "v_fma_f32 v16, v4, v9, v14\n"        // dk
                "v_mul_f32 v17, v16, v16\n"           // sq
                "v_exp_f32 v17, -v17\n"               // w
                "v_fma_f32 v18, v5, v4, v10\n"        // lin
                "v_fma_f32 v18, v18, v11, -v6\n"      // yd
                "v_add_f32 v20, v20, v17\n"           // W
                "v_fmac_f32 v21, v17, v18\n"          // Swy
                "v_mul_f32 v19, v17, v4\n"            // wu
                "v_mul_f32 v16, v16, v19\n"           // av
                "v_mul_f32 v19, v19, v5\n"            // wu * slope
                "v_mul_f32 v19, v19, v7\n"            // * cq
                "v_fmac_f32 v19, v16, v18\n"          // cv
                "v_fmac_f32 v22, v16, v16\n"
                "v_fmac_f32 v23, v16, v19\n"
                "v_fmac_f32 v15, v19, v19\n" CLOB);
        }
        if constexpr (OP == 13) { REP8(asm volatile("v_exp_f32 v20, -v4" CLOB);) }
        if constexpr (OP == 14) { REP8(asm volatile("v_exp_f32 v20, v4" CLOB);) }
        if constexpr (OP == 15) { REP8(asm volatile("v_mul_f32 v20, -v4, v9" CLOB);) }
        if constexpr (OP == 16) {  // the sequence with the exp replaced by a mov (what do the other 14 cost together?)
            asm volatile(
                "v_fma_f32 v16, v4, v9, v14\n"
                "v_mul_f32 v17, v16, v16\n"
                "v_mov_b32 v17, v17\n"
                "v_fma_f32 v18, v5, v4, v10\n"
                "v_fma_f32 v18, v18, v11, -v6\n"
                "v_add_f32 v20, v20, v17\n"
                "v_fmac_f32 v21, v17, v18\n"
                "v_mul_f32 v19, v17, v4\n"
                "v_mul_f32 v16, v16, v19\n"
                "v_mul_f32 v19, v19, v5\n"
                "v_mul_f32 v19, v19, v7\n"
                "v_fmac_f32 v19, v16, v18\n"
                "v_fmac_f32 v22, v16, v16\n"
                "v_fmac_f32 v23, v16, v19\n"
                "v_fmac_f32 v15, v19, v19\n" CLOB);
        }
        if constexpr (OP == 17) {  // 14 independent f32 ops + 1 exp, no dependencies inside the group
            asm volatile(
                "v_fma_f32 v16, v4, v9, v14\n v_mul_f32 v17, v5, v6\n v_exp_f32 v18, v7\n v_fma_f32 v19, v5, v4, v10\n"
                "v_fma_f32 v20, v8, v11, -v6\n v_add_f32 v21, v9, v10\n v_fmac_f32 v22, v4, v5\n v_mul_f32 v23, v6, v4\n"
                "v_mul_f32 v16, v7, v8\n v_mul_f32 v17, v9, v5\n v_mul_f32 v18, v10, v7\n v_fmac_f32 v19, v11, v12\n"
                "v_fmac_f32 v20, v13, v13\n v_fmac_f32 v21, v14, v15\n v_fmac_f32 v22, v4, v4\n" CLOB);
        }
        if constexpr (OP == 18) { asm volatile(
                "v_fma_f32 v16, v4, v9, v14\n"
                "v_mul_f32 v17, v16, v16\n"
                "v_fma_f32 v20, v4, v9, v14\n"
                "v_mul_f32 v21, v20, v20\n"
                "v_fma_f32 v24, v4, v9, v14\n"
                "v_mul_f32 v25, v24, v24\n"
                "v_fma_f32 v28, v4, v9, v14\n"
                "v_mul_f32 v29, v28, v28\n"
                "v_exp_f32 v17, -v17\n"
                "v_exp_f32 v21, -v21\n"
                "v_exp_f32 v25, -v25\n"
                "v_exp_f32 v29, -v29\n"
                "v_fma_f32 v18, v5, v4, v10\n"
                "v_fma_f32 v18, v18, v11, -v6\n"
                "v_add_f32 v32, v32, v17\n"
                "v_fmac_f32 v33, v17, v18\n"
                "v_mul_f32 v19, v17, v4\n"
                "v_mul_f32 v16, v16, v19\n"
                "v_mul_f32 v19, v19, v5\n"
                "v_mul_f32 v19, v19, v7\n"
                "v_fmac_f32 v19, v16, v18\n"
                "v_fmac_f32 v34, v16, v16\n"
                "v_fmac_f32 v35, v16, v19\n"
                "v_fmac_f32 v36, v19, v19\n"
                "v_fma_f32 v22, v5, v4, v10\n"
                "v_fma_f32 v22, v22, v11, -v6\n"
                "v_add_f32 v37, v37, v21\n"
                "v_fmac_f32 v38, v21, v22\n"
                "v_mul_f32 v23, v21, v4\n"
                "v_mul_f32 v20, v20, v23\n"
                "v_mul_f32 v23, v23, v5\n"
                "v_mul_f32 v23, v23, v7\n"
                "v_fmac_f32 v23, v20, v22\n"
                "v_fmac_f32 v39, v20, v20\n"
                "v_fmac_f32 v40, v20, v23\n"
                "v_fmac_f32 v41, v23, v23\n"
                "v_fma_f32 v26, v5, v4, v10\n"
                "v_fma_f32 v26, v26, v11, -v6\n"
                "v_add_f32 v42, v42, v25\n"
                "v_fmac_f32 v43, v25, v26\n"
                "v_mul_f32 v27, v25, v4\n"
                "v_mul_f32 v24, v24, v27\n"
                "v_mul_f32 v27, v27, v5\n"
                "v_mul_f32 v27, v27, v7\n"
                "v_fmac_f32 v27, v24, v26\n"
                "v_fmac_f32 v44, v24, v24\n"
                "v_fmac_f32 v45, v24, v27\n"
                "v_fmac_f32 v46, v27, v27\n"
                "v_fma_f32 v30, v5, v4, v10\n"
                "v_fma_f32 v30, v30, v11, -v6\n"
                "v_add_f32 v47, v47, v29\n"
                "v_fmac_f32 v48, v29, v30\n"
                "v_mul_f32 v31, v29, v4\n"
                "v_mul_f32 v28, v28, v31\n"
                "v_mul_f32 v31, v31, v5\n"
                "v_mul_f32 v31, v31, v7\n"
                "v_fmac_f32 v31, v28, v30\n"
                "v_fmac_f32 v49, v28, v28\n"
                "v_fmac_f32 v50, v28, v31\n"
                "v_fmac_f32 v51, v31, v31\n" CLOB2); }
        if constexpr (OP == 11) { asm volatile(
                "v_fma_f32 v16, v4, v9, v14\n"
                "v_fma_f32 v20, v4, v9, v14\n"
                "v_fma_f32 v24, v4, v9, v14\n"
                "v_fma_f32 v28, v4, v9, v14\n"
                "v_mul_f32 v17, v16, v16\n"
                "v_mul_f32 v21, v20, v20\n"
                "v_mul_f32 v25, v24, v24\n"
                "v_mul_f32 v29, v28, v28\n"
                "v_exp_f32 v17, -v17\n"
                "v_exp_f32 v21, -v21\n"
                "v_exp_f32 v25, -v25\n"
                "v_exp_f32 v29, -v29\n"
                "v_fma_f32 v18, v5, v4, v10\n"
                "v_fma_f32 v22, v5, v4, v10\n"
                "v_fma_f32 v26, v5, v4, v10\n"
                "v_fma_f32 v30, v5, v4, v10\n"
                "v_fma_f32 v18, v18, v11, -v6\n"
                "v_fma_f32 v22, v22, v11, -v6\n"
                "v_fma_f32 v26, v26, v11, -v6\n"
                "v_fma_f32 v30, v30, v11, -v6\n"
                "v_add_f32 v32, v32, v17\n"
                "v_add_f32 v37, v37, v21\n"
                "v_add_f32 v42, v42, v25\n"
                "v_add_f32 v47, v47, v29\n"
                "v_fmac_f32 v33, v17, v18\n"
                "v_fmac_f32 v38, v21, v22\n"
                "v_fmac_f32 v43, v25, v26\n"
                "v_fmac_f32 v48, v29, v30\n"
                "v_mul_f32 v19, v17, v4\n"
                "v_mul_f32 v23, v21, v4\n"
                "v_mul_f32 v27, v25, v4\n"
                "v_mul_f32 v31, v29, v4\n"
                "v_mul_f32 v16, v16, v19\n"
                "v_mul_f32 v20, v20, v23\n"
                "v_mul_f32 v24, v24, v27\n"
                "v_mul_f32 v28, v28, v31\n"
                "v_mul_f32 v19, v19, v5\n"
                "v_mul_f32 v23, v23, v5\n"
                "v_mul_f32 v27, v27, v5\n"
                "v_mul_f32 v31, v31, v5\n"
                "v_mul_f32 v19, v19, v7\n"
                "v_mul_f32 v23, v23, v7\n"
                "v_mul_f32 v27, v27, v7\n"
                "v_mul_f32 v31, v31, v7\n"
                "v_fmac_f32 v19, v16, v18\n"
                "v_fmac_f32 v23, v20, v22\n"
                "v_fmac_f32 v27, v24, v26\n"
                "v_fmac_f32 v31, v28, v30\n"
                "v_fmac_f32 v34, v16, v16\n"
                "v_fmac_f32 v39, v20, v20\n"
                "v_fmac_f32 v44, v24, v24\n"
                "v_fmac_f32 v49, v28, v28\n"
                "v_fmac_f32 v35, v16, v19\n"
                "v_fmac_f32 v40, v20, v23\n"
                "v_fmac_f32 v45, v24, v27\n"
                "v_fmac_f32 v50, v28, v31\n"
                "v_fmac_f32 v36, v19, v19\n"
                "v_fmac_f32 v41, v23, v23\n"
                "v_fmac_f32 v46, v27, v27\n"
                "v_fmac_f32 v51, v31, v31\n" CLOB2); }
        if constexpr (OP == 12) { asm volatile(
                "v_fma_f32 v16, v4, v9, v14\n"
                "v_mul_f32 v17, v16, v16\n"
                "v_exp_f32 v17, -v17\n"
                "v_fma_f32 v18, v5, v4, v10\n"
                "v_fma_f32 v18, v18, v11, -v6\n"
                "v_add_f32 v32, v32, v17\n"
                "v_fmac_f32 v33, v17, v18\n"
                "v_mul_f32 v19, v17, v4\n"
                "v_mul_f32 v16, v16, v19\n"
                "v_mul_f32 v19, v19, v5\n"
                "v_mul_f32 v19, v19, v7\n"
                "v_fmac_f32 v19, v16, v18\n"
                "v_fmac_f32 v34, v16, v16\n"
                "v_fmac_f32 v35, v16, v19\n"
                "v_fmac_f32 v36, v19, v19\n"
                "v_fma_f32 v20, v4, v9, v14\n"
                "v_mul_f32 v21, v20, v20\n"
                "v_exp_f32 v21, -v21\n"
                "v_fma_f32 v22, v5, v4, v10\n"
                "v_fma_f32 v22, v22, v11, -v6\n"
                "v_add_f32 v37, v37, v21\n"
                "v_fmac_f32 v38, v21, v22\n"
                "v_mul_f32 v23, v21, v4\n"
                "v_mul_f32 v20, v20, v23\n"
                "v_mul_f32 v23, v23, v5\n"
                "v_mul_f32 v23, v23, v7\n"
                "v_fmac_f32 v23, v20, v22\n"
                "v_fmac_f32 v39, v20, v20\n"
                "v_fmac_f32 v40, v20, v23\n"
                "v_fmac_f32 v41, v23, v23\n"
                "v_fma_f32 v24, v4, v9, v14\n"
                "v_mul_f32 v25, v24, v24\n"
                "v_exp_f32 v25, -v25\n"
                "v_fma_f32 v26, v5, v4, v10\n"
                "v_fma_f32 v26, v26, v11, -v6\n"
                "v_add_f32 v42, v42, v25\n"
                "v_fmac_f32 v43, v25, v26\n"
                "v_mul_f32 v27, v25, v4\n"
                "v_mul_f32 v24, v24, v27\n"
                "v_mul_f32 v27, v27, v5\n"
                "v_mul_f32 v27, v27, v7\n"
                "v_fmac_f32 v27, v24, v26\n"
                "v_fmac_f32 v44, v24, v24\n"
                "v_fmac_f32 v45, v24, v27\n"
                "v_fmac_f32 v46, v27, v27\n"
                "v_fma_f32 v28, v4, v9, v14\n"
                "v_mul_f32 v29, v28, v28\n"
                "v_exp_f32 v29, -v29\n"
                "v_fma_f32 v30, v5, v4, v10\n"
                "v_fma_f32 v30, v30, v11, -v6\n"
                "v_add_f32 v47, v47, v29\n"
                "v_fmac_f32 v48, v29, v30\n"
                "v_mul_f32 v31, v29, v4\n"
                "v_mul_f32 v28, v28, v31\n"
                "v_mul_f32 v31, v31, v5\n"
                "v_mul_f32 v31, v31, v7\n"
                "v_fmac_f32 v31, v28, v30\n"
                "v_fmac_f32 v49, v28, v28\n"
                "v_fmac_f32 v50, v28, v31\n"
                "v_fmac_f32 v51, v31, v31\n" CLOB2); }
    }
    float r;
    asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n v_add_f32 %0, %0, v15" : "=v"(r) : :
                 "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP> void run(const char *name, double per_iter, float *out)
{
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double wave_insts = (double)blocks * 4 * ITERS * per_iter;
    printf("%-58s %.3f ms  => %.2f nominal cycles (2.4 GHz) per wave-instr per SIMD\n", name, ms, 2400.0 / (wave_insts / 1024.0 / (ms * 1e3)));
}

int main()
{
    float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("v_fma_f32 sources in banks 0,1,2", 8, out);
    run<1>("v_fma_f32 all three sources in bank 0", 8, out);
    run<2>("v_fma_f32 two sources in bank 0", 8, out);
    run<3>("v_fma_f32 same register twice", 8, out);
    run<4>("v_mul_f32 banks 0,1", 8, out);
    run<5>("v_mul_f32 both bank 0", 8, out);
    run<6>("v_fmac_f32 dest bank 0, sources banks 0,1", 8, out);
    run<7>("v_fmac_f32 dest bank 1, sources banks 0,2", 8, out);
    run<8>("v_fmac_f32 everything in bank 0", 8, out);
    run<9>("v_fma_f32 dependent chain (8 waves per SIMD)", 8, out);
    run<10>("merge loop sequence, 15 instr (14 f32 + exp), per instr", 15, out);
    run<14>("v_exp_f32", 8, out);
    run<13>("v_exp_f32 with neg modifier (VOP3)", 8, out);
    run<15>("v_mul_f32 with neg modifier (VOP3)", 8, out);
    run<16>("merge sequence with the exp replaced by v_mov", 15, out);
    run<17>("14 independent f32 ops + 1 exp", 15, out);
    run<12>("same, 4 elements one after the other (60 instr)", 60, out);
    run<11>("same, 4 elements interleaved instruction by instruction", 60, out);
    run<18>("same, the 4 v_exp_f32 back to back, rest element by element", 60, out);
    return 0;
}

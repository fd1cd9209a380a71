// typed_load_probe.hip -- does gfx950 convert integer codes to float32 in the buffer-load path, at what rate, and is the
// LUT interval exact when it is formed from that float by one FMA under round-toward-minus-infinity?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/typed_load_probe.hip -o tools/typed_load_probe
// 1. buffer_load_format_{x,xyzw} through a descriptor with DATA_FORMAT 16_16_16_16 / 8_8_8_8 / 16 / 8 and NUM_FORMAT
//    USCALED: every code 0..65535 (0..255) in every component position must come back as (float)code.
// 2. interval = as_uint(fma(px, r, 1.5 * 2^23)) - 0x4B400000 with FP_ROUND(single) = toward -inf and r = 1/step rounded
//    up must equal floor(code / step) for every code, for the steps ct_pivot_index_constants admits.
// 3. streaming rate: 32 exposures of 4096x4096x3 uint16, 4 codes per lane per exposure, typed load against
//    buffer_load_dwordx2 + 4 v_cvt_f32_u32_sdwa.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
__device__ f4 llvm_buffer_load_format_v4f32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
__device__ float llvm_buffer_load_format_f32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.f32");
__device__ u2 llvm_buffer_load_v2i32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");

// word 3 of a gfx9 buffer descriptor: DST_SEL x,y,z,w = R,G,B,A | NUM_FORMAT << 12 | DATA_FORMAT << 15
constexpr uint32_t kSelXYZW = 4u | (5u << 3) | (6u << 6) | (7u << 9);
constexpr uint32_t kUscaled = 2u << 12;
constexpr uint32_t kFmt8 = 1u << 15, kFmt16 = 2u << 15, kFmt8888 = 10u << 15, kFmt16x4 = 12u << 15, kFmt32 = 4u << 15;

__device__ __forceinline__ i4 make_rsrc(const void *p, uint32_t word3)
{
    const uint64_t b = reinterpret_cast<uint64_t>(p);
    i4 r;
    r.x = (int)(uint32_t)b;
    r.y = (int)(uint32_t)(b >> 32);  // stride 0: raw addressing, base + offset
    r.z = -1;
    r.w = (int)word3;
    return r;
}

__global__ void probe_x4(const void *in, float *out, int n_packets, uint32_t word3, int packet_bytes)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_packets) return;
    const f4 v = llvm_buffer_load_format_v4f32(make_rsrc(in, word3), t * packet_bytes, 0, 0);
    out[4 * t + 0] = v.x; out[4 * t + 1] = v.y; out[4 * t + 2] = v.z; out[4 * t + 3] = v.w;
}
__global__ void probe_x1(const void *in, float *out, int n, uint32_t word3, int elem_bytes)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    out[t] = llvm_buffer_load_format_f32(make_rsrc(in, word3), t * elem_bytes, 0, 0);
}

// interval of 4 codes under round-toward-minus-infinity, one asm block so nothing else runs in that mode
__global__ void probe_floor(const float *px, uint32_t *out, int n, float r)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * t >= n) return;
    const float magic = 12582912.0f;
    float a = px[4 * t], b = px[4 * t + 1], c = px[4 * t + 2], d = px[4 * t + 3];
    float ra, rb, rc, rd;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
                 "v_fma_f32 %0, %4, %8, %9\n\t"
                 "v_fma_f32 %1, %5, %8, %9\n\t"
                 "v_fma_f32 %2, %6, %8, %9\n\t"
                 "v_fma_f32 %3, %7, %8, %9\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd)
                 : "v"(a), "v"(b), "v"(c), "v"(d), "v"(r), "v"(magic));
    out[4 * t] = __float_as_uint(ra) - 0x4B400000u;
    out[4 * t + 1] = __float_as_uint(rb) - 0x4B400000u;
    out[4 * t + 2] = __float_as_uint(rc) - 0x4B400000u;
    out[4 * t + 3] = __float_as_uint(rd) - 0x4B400000u;
    // an FMA after the block must round to nearest again: 1 + 2^-24 * 1.5 -> 1 + 2^-23 under RNE, 1 under RTN
    volatile float one = 1.0f, eps = 8.940696716308594e-08f;
    const float chk = __builtin_fmaf(one, one, eps);
    if (t == 0) out[n] = __float_as_uint(chk);
}

// LDS address by a denormal FMA: as_uint(fma(1.5 * 2^23 + i, 8 2^-149, c(row))) must be 8 i + row
__global__ void probe_addr(uint32_t *out, int n, int row)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = 12582912.0f + (float)i;
    const float c = __builtin_fmaf((float)row, __uint_as_float(1u), -__uint_as_float(0x02400000u));
    uint32_t adr;
    asm("v_fma_f32 %0, %1, 8, %2" : "=v"(adr) : "v"(t), "v"(c));
    out[i] = adr;
}

template <bool TYPED>
__global__ __launch_bounds__(256) void stream(const uint16_t *stack, float *out, uint32_t vecs, int batch, int64_t image_stride)
{
    const uint32_t vec = blockIdx.x * 256u + threadIdx.x;
    if (vec >= vecs) return;
    float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll 4
    for (int n = 0; n < batch; ++n) {
        const uint16_t *base = stack + (int64_t)n * image_stride;
        if constexpr (TYPED) {
            const f4 v = llvm_buffer_load_format_v4f32(make_rsrc(base, kSelXYZW | kUscaled | kFmt16x4), (int)(vec * 8u), 0, 0);
            acc0 += v.x; acc1 += v.y; acc2 += v.z; acc3 += v.w;
        } else {
            const u2 v = llvm_buffer_load_v2i32(make_rsrc(base, kFmt32), (int)(vec * 8u), 0, 0);
            float a, b, c, d;
            asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(a) : "v"(v.x));
            asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(b) : "v"(v.x));
            asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(c) : "v"(v.y));
            asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(d) : "v"(v.y));
            acc0 += a; acc1 += b; acc2 += c; acc3 += d;
        }
    }
    f4 o = {acc0, acc1, acc2, acc3};
    reinterpret_cast<f4 *>(out)[vec] = o;
}

int main()
{
    int bad_total = 0;
    {   // 1a. 16_16_16_16 USCALED: codes 0..65535, four rotations so every code visits every component
        std::vector<uint16_t> h(65536 * 4);
        for (int i = 0; i < 65536; ++i) for (int c = 0; c < 4; ++c) h[4 * i + c] = (uint16_t)((i + c * 16411) & 0xffff);
        uint16_t *d; float *o; CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, h.size() * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe_x4, dim3(256), dim3(256), 0, 0, d, o, 65536, kSelXYZW | kUscaled | kFmt16x4, 8);
        std::vector<float> r(h.size()); CK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (size_t i = 0; i < h.size(); ++i) bad += r[i] != (float)h[i];
        printf("16_16_16_16 USCALED xyzw: %d of %zu wrong (first values %g %g %g %g for codes %u %u %u %u)\n", bad, h.size(), r[4], r[5], r[6], r[7], h[4], h[5], h[6], h[7]);
        bad_total += bad;
        // 1b. single 16 USCALED
        hipLaunchKernelGGL(probe_x1, dim3(1024), dim3(256), 0, 0, d, o, 65536 * 4, 4u | kUscaled | kFmt16, 2);
        CK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
        bad = 0; for (size_t i = 0; i < h.size(); ++i) bad += r[i] != (float)h[i];
        printf("16 USCALED x: %d of %zu wrong\n", bad, h.size());
        bad_total += bad;
        CK(hipFree(d)); CK(hipFree(o));
    }
    {   // 1c. 8_8_8_8 and 8 USCALED
        std::vector<uint8_t> h(256 * 4 * 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)((i * 37 + (i >> 2)) & 0xff);
        uint8_t *d; float *o; CK(hipMalloc(&d, h.size())); CK(hipMalloc(&o, h.size() * 4));
        CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe_x4, dim3(4), dim3(256), 0, 0, d, o, (int)h.size() / 4, kSelXYZW | kUscaled | kFmt8888, 4);
        std::vector<float> r(h.size()); CK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (size_t i = 0; i < h.size(); ++i) bad += r[i] != (float)h[i];
        printf("8_8_8_8 USCALED xyzw: %d of %zu wrong\n", bad, h.size());
        bad_total += bad;
        hipLaunchKernelGGL(probe_x1, dim3(16), dim3(256), 0, 0, d, o, (int)h.size(), 4u | kUscaled | kFmt8, 1);
        CK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
        bad = 0; for (size_t i = 0; i < h.size(); ++i) bad += r[i] != (float)h[i];
        printf("8 USCALED x: %d of %zu wrong\n", bad, h.size());
        bad_total += bad;
        CK(hipFree(d)); CK(hipFree(o));
    }
    {   // 2. floor(code / step) by one FMA under round-toward-minus-infinity
        std::vector<float> h(65536); for (int i = 0; i < 65536; ++i) h[i] = (float)i;
        float *d; uint32_t *o; CK(hipMalloc(&d, 65536 * 4)); CK(hipMalloc(&o, 65537 * 4));
        CK(hipMemcpy(d, h.data(), 65536 * 4, hipMemcpyHostToDevice));
        const int steps[] = {1, 3, 5, 15, 17, 51, 85, 255, 257, 771, 1285, 3855, 4369, 13107, 21845, 65535};
        for (int st : steps) {
            float r = (float)(1.0 / st);
            if ((double)r < 1.0 / st) r = nextafterf(r, 2.0f);  // rounded up: exact multiples must not fall below
            hipLaunchKernelGGL(probe_floor, dim3(64), dim3(256), 0, 0, d, o, 65536, r);
            std::vector<uint32_t> g(65537); CK(hipMemcpy(g.data(), o, 65537 * 4, hipMemcpyDeviceToHost));
            int bad = 0; for (int i = 0; i < 65536; ++i) bad += g[i] != (uint32_t)(i / st);
            printf("floor(code / %5d) by RTN FMA: %d of 65536 wrong; FMA after the block rounds to nearest: %s\n", st, bad, g[65536] == 0x3f800001u ? "yes" : "NO");
            bad_total += bad + (g[65536] != 0x3f800001u);
        }
        CK(hipFree(d)); CK(hipFree(o));
    }
    {   // 2b. LDS byte address by one denormal FMA
        uint32_t *o; CK(hipMalloc(&o, 65536 * 4));
        const int rows[] = {0, 2048, 4096, 8 * 4369 * 2, 163832};
        for (int row : rows) {
            hipLaunchKernelGGL(probe_addr, dim3(256), dim3(256), 0, 0, o, 65536, row);
            std::vector<uint32_t> g(65536); CK(hipMemcpy(g.data(), o, 65536 * 4, hipMemcpyDeviceToHost));
            int bad = 0; for (int i = 0; i < 65536; ++i) bad += g[i] != (uint32_t)(8 * i + row);
            printf("address 8 i + %8d by denormal FMA: %d of 65536 wrong\n", row, bad);
            bad_total += bad;
        }
        CK(hipFree(o));
    }
    {   // 3. streaming rate
        const int N = 32; const size_t Q = (size_t)3 * 4096 * 4096;
        uint16_t *stack; float *out; CK(hipMalloc(&stack, Q * N * 2)); CK(hipMalloc(&out, Q * 4));
        CK(hipMemset(stack, 0x5a, Q * N * 2));
        const uint32_t vecs = (uint32_t)(Q / 4);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int which = 0; which < 2; ++which) {
            for (int rep = 0; rep < 2; ++rep) {   // each variant sustained alone (the card sits on its power limit)
                std::vector<float> ms;
                for (int k = 0; k < 300; ++k) {
                    CK(hipEventRecord(e0, 0));
                    if (which == 0) hipLaunchKernelGGL(stream<false>, dim3((vecs + 255) / 256), dim3(256), 0, 0, stack, out, vecs, N, (int64_t)Q);
                    else hipLaunchKernelGGL(stream<true>, dim3((vecs + 255) / 256), dim3(256), 0, 0, stack, out, vecs, N, (int64_t)Q);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float t; CK(hipEventElapsedTime(&t, e0, e1));
                    if (k >= 100) ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                const double bytes = (double)Q * N * 2 + (double)Q * 4;
                printf("stream %-28s median %.3f ms (%.2f TB/s)\n", which ? "buffer_load_format_xyzw" : "buffer_load_dwordx2 + 4 cvt", ms[ms.size() / 2], bytes / ms[ms.size() / 2] / 1e9);
            }
        }
    }
    printf(bad_total ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad_total ? 1 : 0;
}

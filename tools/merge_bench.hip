// merge_bench.hip -- development harness: times instantiations <V, PF> of the PRODUCT merge kernel
// (csrc/ct_merge.hip is included verbatim) on the C2 shape, interleaved rounds in one process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/merge_bench.hip \
//         clair_torch_amd/csrc/ct_merge_exact.hip clair_torch_amd/csrc/ct_api.cpp -o tools/merge_bench
#define CT_MERGE_PART 2  // both translation units of the product file in one
#include "../clair_torch_amd/csrc/ct_merge.hip"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_random(uint16_t *p, size_t n, uint32_t seed, int mode, size_t Q)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)(i % Q) * 2654435761u + seed;
        h ^= h >> 16; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        if (mode == 0) {
            uint32_t g = (uint32_t)i * 2654435761u + seed; g ^= g >> 16; g *= 2246822519u; g ^= g >> 13;
            p[i] = (uint16_t)(g >> 8);
        } else {
            int nn = (int)(i / Q);
            float E = (h >> 8) * (1.0f / 16777216.0f) * 2.0f / 0.0146f;
            if (mode == 2) E = (float)((i % Q) % 4096u) * (1.0f / 4096.0f) * 2.0f / 0.0146f;  // smooth ramp along a row
            float t = 0.001f * exp2f(nn * 0.25f);
            float lin = fminf(E * t, 1.0f);
            p[i] = (uint16_t)rintf(powf(lin, 1.0f / 2.2f) * 65535.0f);
        }
    }
}

// streaming yardstick with the product kernel's access pattern: V uint16 per lane per exposure, 12 B written per element
template <int V>
__global__ __launch_bounds__(256) void stream_only(const ct::MergeArgs a)
{
    const uint32_t vec = blockIdx.x * 256u + threadIdx.x;
    if (vec * V >= a.q_count) return;
    const uint32_t q0 = vec * V;
    uint32_t acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0;
    const uint16_t *src = static_cast<const uint16_t *>(a.stack) + q0;
#pragma unroll 4
    for (int n = 0; n < a.batch; ++n) {
        const ct::Packet<uint16_t, V> pk = *reinterpret_cast<const ct::Packet<uint16_t, V> *>(src + (int64_t)n * a.image_stride);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += pk.v[e];
    }
    ct::Packet<double, V> mo; ct::Packet<float, V> so;
#pragma unroll
    for (int e = 0; e < V; ++e) { mo.v[e] = (double)acc[e]; so.v[e] = (float)acc[e]; }
    *reinterpret_cast<ct::Packet<double, V> *>(static_cast<double *>(a.mean_out) + q0) = mo;
    *reinterpret_cast<ct::Packet<float, V> *>(a.std_out + q0) = so;
}
template <int V> void launch_stream(const ct::MergeArgs &a, hipStream_t s)
{
    uint32_t vecs = a.q_count / V, grid = (vecs + 255) / 256;
    hipLaunchKernelGGL((stream_only<V>), dim3(grid), dim3(256), 0, s, a);
}

using namespace ct;
struct Variant { std::string name; void (*launch)(const MergeArgs &, hipStream_t); };

template <int V, int PF, int STD>
void launch_v(const MergeArgs &a0, hipStream_t s)
{
    MergeArgs a = a0;
    const uint32_t vecs = a.q_count / V, grid = (vecs + kBlock - 1) / kBlock;
    const size_t lds = (size_t)a.channels * a.n_points * lut_entry_bytes(CT_INTERP_LINEAR) + 2 * sizeof(float) * (size_t)a.batch;
    hipLaunchKernelGGL((merge_kernel<uint16_t, V, CT_INTERP_LINEAR, CT_WEIGHT_GAUSS, STD, true, PF, false>), dim3(grid), dim3(kBlock), lds, s, a);
}

static PivotArgs g_px;
template <int STD, int V = kPivotV>
void launch_pivot_v(const MergeArgs &a0, hipStream_t s)
{
    MergeArgs a = a0;
    a.q_count = (a.q_count / (kBlock * V)) * (kBlock * V);
    int rc = launch_pivot<uint16_t, V, CT_INTERP_LINEAR, CT_WEIGHT_GAUSS, STD, false>(a, g_px, s);
    if (rc != CT_OK) { printf("launch_pivot rc %d\n", rc); exit(1); }
}

// compute-bound yardstick: every exposure aliases exposure 0 (image_stride = 0), so the stack traffic drops to 1/32
template <int STD>
void launch_pivot_nomem(const MergeArgs &a0, hipStream_t s)
{
    MergeArgs a = a0;
    a.image_stride = 0;
    launch_pivot_v<STD>(a, s);
}

// compare two result sets on the device: max element-wise relative error, norm-wise relative error
__global__ void compare_kernel(const double *m0, const float *s0, const double *m1, const float *s1, size_t n, double *acc)
{
    // acc: [0] sum (m1-m0)^2 [1] sum m0^2 [2] sum (s1-s0)^2 [3] sum s0^2 [4] max rel mean (bits) [5] max rel std (bits)
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, mx0 = 0, mx1 = 0;
    for (; i < n; i += stride) {
        double dm = m1[i] - m0[i], ds = (double)s1[i] - (double)s0[i];
        a0 += dm * dm; a1 += m0[i] * m0[i]; a2 += ds * ds; a3 += (double)s0[i] * s0[i];
        mx0 = fmax(mx0, fabs(dm) / fmax(fabs(m0[i]), 1e-300));
        mx1 = fmax(mx1, fabs(ds) / fmax(fabs((double)s0[i]), 1e-300));
    }
    atomicAdd(acc + 0, a0); atomicAdd(acc + 1, a1); atomicAdd(acc + 2, a2); atomicAdd(acc + 3, a3);
    atomicMax((unsigned long long *)(acc + 4), (unsigned long long)__double_as_longlong(mx0));
    atomicMax((unsigned long long *)(acc + 5), (unsigned long long)__double_as_longlong(mx1));
}

int main(int argc, char **argv)
{
    int N = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 4096, Wd = argc > 3 ? atoi(argv[3]) : 4096;
    int rounds = argc > 4 ? atoi(argv[4]) : 7;
    const size_t Q = (size_t)3 * H * Wd, S = Q * N;
    uint16_t *stack; float *lut; double *expo; double *mean; float *stdo;
    CK(hipMalloc(&stack, S * 2)); CK(hipMalloc(&lut, 768 * 4)); CK(hipMalloc(&expo, N * 8));
    CK(hipMalloc(&mean, Q * 8)); CK(hipMalloc(&stdo, Q * 4));
    std::vector<float> hl(768); std::vector<double> he(N);
    const double pw[3] = {2.2, 2.4, 2.6};
    const bool step_lut = getenv("MERGE_BENCH_STEP_LUT") != nullptr;  // a LUT with a jump: exercises the exact-interval path
    for (int r = 0; r < 3; ++r) for (int i = 0; i < 256; ++i)
        hl[r * 256 + i] = (float)pow(i / 255.0, pw[r]) * (step_lut && i >= 200 ? 9.0f : 1.0f);
    for (int n = 0; n < N; ++n) he[n] = 0.001 * pow(2.0, n / 4.0);
    CK(hipMemcpy(lut, hl.data(), 768 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(expo, he.data(), N * 8, hipMemcpyHostToDevice));
    MergeArgs a{};
    a.stack = stack; a.exposure = expo; a.lut = lut; a.mean_out = mean; a.std_out = stdo;
    a.image_stride = (int64_t)Q; a.q_begin = 0; a.q_count = (uint32_t)Q;
    a.tile.plane_local = (uint32_t)((size_t)H * Wd); a.tile.chan_skip = 0; a.tile.base = 0;
    a.batch = N; a.channels = 3; a.n_points = 256;
    ct_norm_constants(65535.0f, &a.norm.hi, &a.norm.lo);
    if (ct_index_constants(65535.0f, 256, &a.index.hi, &a.index.lo) != CT_OK) { printf("fold refused\n"); return 1; }
    a.inv_max_code = (float)(1.0 / 65535.0); a.std_value = 0.05f; a.weight_scale = 30.0f;
    a.flags = CT_MERGE_FIRST_BATCH | CT_MERGE_FINALIZE;

    if (ct_pivot_index_constants(65535.0f, 256, &g_px.index_mul, &g_px.step) != CT_OK) { printf("pivot index refused\n"); return 1; }
    if (ct_pivot_floor_constants(65535.0f, 256, &g_px.index_rcp) != CT_OK) { printf("pivot floor constants refused\n"); return 1; }
    g_px.probe = N / 2; g_px.max_code = 65535.0f; g_px.n_entries = 256; g_px.tf_max = kFloorMagic + 255.0f;
    unsigned long long *retries; CK(hipMalloc(&retries, 8)); CK(hipMemset(retries, 0, 8));
    g_px.retry_count = retries;
    double *mean2; float *std2; double *acc;
    CK(hipMalloc(&mean2, Q * 8)); CK(hipMalloc(&std2, Q * 4)); CK(hipMalloc(&acc, 6 * 8));
    std::vector<Variant> vs = {
        {"pivot V4 mult", launch_pivot_v<CT_STD_MULTIPLIER>}, {"pivot V4 nostd", launch_pivot_v<CT_STD_NONE>},
        {"compute-only yardstick (every exposure aliases exposure 0)", launch_pivot_nomem<CT_STD_MULTIPLIER>},
        {"stream V4 (8B/lane)", launch_stream<4>}, {"stream V8 (16B/lane)", launch_stream<8>},
        {"f64 V4 PF2 mult", launch_v<4, 2, CT_STD_MULTIPLIER>}, {"f64 V4 PF2 nostd", launch_v<4, 2, CT_STD_NONE>},
    };
    const char *only = argc > 5 ? argv[5] : nullptr;   // run only the variants whose name contains this (for rocprofv3 --pmc)
    if (only) {
        std::vector<Variant> keep;
        // '|'-separated list of substrings
        std::string pat(only);
        for (auto &v : vs) {
            size_t at = 0;
            bool hit = false;
            while (at <= pat.size()) {
                const size_t bar = pat.find('|', at);
                const std::string one = pat.substr(at, bar == std::string::npos ? std::string::npos : bar - at);
                if (!one.empty() && v.name.find(one) != std::string::npos) hit = true;
                if (bar == std::string::npos) break;
                at = bar + 1;
            }
            if (hit) keep.push_back(v);
        }
        vs = keep;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)S * 2 + (double)Q * 12;
    const int mode_only = getenv("MERGE_BENCH_MODE") ? atoi(getenv("MERGE_BENCH_MODE")) : 1;  // data set of a filtered run
    for (int mode = only ? mode_only : 0; mode < (only ? mode_only + 1 : 3); ++mode) {
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, stack, S, 12345u, mode, Q);
        CK(hipDeviceSynchronize());
        printf("== data: %s, N=%d %dx%d, algorithmic bytes %.3f GB ==\n", mode == 0 ? "uniform random codes" : mode == 1 ? "gamma-2.2 scene, independent pixels" : "gamma-2.2 scene, smooth ramp", N, H, Wd, bytes / 1e9);
        std::vector<std::vector<float>> ms(vs.size());
        for (int r = 0; r < rounds + 1; ++r)
            for (size_t v = 0; v < vs.size(); ++v) {
                CK(hipEventRecord(e0, 0));
                vs[v].launch(a, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (r > 0) ms[v].push_back(t);
            }
        CK(hipGetLastError());
        {   // parity of the pivoted float32 kernel against the float64-moment kernel on this data
            MergeArgs b = a; b.mean_out = mean2; b.std_out = std2;
            launch_v<4, 2, CT_STD_MULTIPLIER>(a, 0);
            CK(hipMemset(retries, 0, 8));
            launch_pivot_v<CT_STD_MULTIPLIER>(b, 0);
            CK(hipMemset(acc, 0, 48));
            const size_t nq = (Q / (kBlock * kPivotV)) * (kBlock * kPivotV);
            hipLaunchKernelGGL(compare_kernel, dim3(2048), dim3(256), 0, 0, mean, stdo, mean2, std2, nq, acc);
            double h[6]; unsigned long long hr;
            CK(hipMemcpy(h, acc, 48, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hr, retries, 8, hipMemcpyDeviceToHost));
            printf("pivot vs f64: mean norm %.3g max %.3g | std norm %.3g max %.3g | fallback wavefronts %llu of %zu\n",
                   sqrt(h[0] / h[1]), h[4], sqrt(h[2] / h[3]), h[5], hr, nq / 256);
        }
        for (size_t v = 0; v < vs.size(); ++v) {
            std::sort(ms[v].begin(), ms[v].end());
            float med = ms[v][ms[v].size() / 2], mn = ms[v][0];
            printf("%-16s med %.3f ms  min %.3f ms  %.2f TB/s  %.1f%% of 8TB/s  %.0f MPix/s\n", vs[v].name.c_str(), med, mn,
                   bytes / med / 1e9, bytes / med / 1e9 / 8.0 * 100, (double)H * Wd / med / 1e3);
        }
    }
    std::vector<float> hs(16); CK(hipMemcpy(hs.data(), stdo, 64, hipMemcpyDeviceToHost));
    printf("std[0..3] = %g %g %g %g\n", hs[0], hs[1], hs[2], hs[3]);
    return 0;
}

// dev tool: how fast does a 256-workgroup grid stream a weight matrix in the access pattern of the NT GEMM
// (per workgroup and stage: 64 rows x 512 B, row stride 16 KB) compared with the same bytes laid out contiguously?
// Each wave mimics gemm_nt: lane (i = lane & 15, q = lane >> 4) loads 16 B at row (16 wave + i), byte offset 64 s + 16 q.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// W [N][K] row-major; grid (N/64, 1, nsplit); each workgroup walks its K range in stages of 128 floats
__global__ __launch_bounds__(256) void strided(const float* __restrict__ W, int K, int stages, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const size_t row = (size_t)blockIdx.x * 64 + wave * 16 + li;
    const float* p = W + row * K + (size_t)blockIdx.z * stages * 128 + 4 * lq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < stages; ++st) {
        f32x4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = *reinterpret_cast<const f32x4*>(p + 16 * s);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc += v[s];
        p += 128;
    }
    if (acc[0] == 12345.678f) out[0] = acc[1] + acc[2] + acc[3];
}
// packed: the 32 KB a workgroup reads per stage are contiguous: [tile][split][stage][64 rows][128 k]
__global__ __launch_bounds__(256) void packed(const float* __restrict__ W, int stages, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const size_t wg = (size_t)blockIdx.x * gridDim.z + blockIdx.z;
    const float* p = W + wg * stages * 8192 + (size_t)(wave * 16 + li) * 128 + 4 * lq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < stages; ++st) {
        f32x4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = *reinterpret_cast<const f32x4*>(p + 16 * s);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc += v[s];
        p += 8192;
    }
    if (acc[0] == 12345.678f) out[0] = acc[1] + acc[2] + acc[3];
}
int main() {
    const int N = 4096, K = 4096, ns = 4, stages = K / 128 / ns;
    const int NW = 6;                                   // rotate over several matrices (400 MB) so that nothing stays cached
    float* W[NW]; float* out;
    for (int i = 0; i < NW; ++i) { hipMalloc(&W[i], (size_t)N * K * 4); hipMemset(W[i], 0, (size_t)N * K * 4); }
    hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rot = 0; rot < 2; ++rot) {              // rot = 0: the same matrix every launch (MALL-resident), 1: rotating (HBM)
            float ms;
            const int iters = 60;
            for (int it = -10; it < iters; ++it) {
                if (it == 0) hipEventRecord(e0, 0);
                const float* w = W[rot ? (it + 10) % NW : 0];
                if (mode == 0) hipLaunchKernelGGL(strided, dim3(N / 64, 1, ns), dim3(256), 0, 0, w, K, stages, out);
                else hipLaunchKernelGGL(packed, dim3(N / 64, 1, ns), dim3(256), 0, 0, w, stages, out);
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / iters;
            printf("%s %s: %.2f us per launch, %.2f TB/s (67 MB)\n", mode ? "packed " : "strided", rot ? "rotating(HBM)" : "same(MALL)  ", us,
                   (double)N * K * 4 / us / 1e6);
        }
    return 0;
}

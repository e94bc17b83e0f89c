// dev tool: issue rate of v_mfma_f32_16x16x4_f32 with NACC independent accumulators, one wave per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_wave = (double)iters * 16 * NACC;
    double ns_per = ms * 1e6 / mfma_per_wave;
    double tf = (double)blocks * 4 * mfma_per_wave * 2048 / (ms * 1e-3) / 1e12;
    printf("NACC=%d blocks=%d: %.2f ms, %.2f ns per MFMA per wave (= %.1f cycles @2.4GHz), %.1f TF\n", NACC, blocks, ms, ns_per, ns_per * 2.4, tf);
    hipFree(out);
}
int main() {
    run<1>(256); run<2>(256); run<4>(256); run<8>(256);
    run<4>(512); run<4>(64);
    return 0;
}

// Round 5 microbenchmark: do trailing workgroups of a small kernel that WARM the heads of the next GEMM's weight
// ranges in the consumer XCD's L2 shorten the dependency edge small kernel -> GEMM of a decoder step?
//
// The decode step is a chain: ... -> pointwise (5 us, 64 x 4 workgroups) -> gate GEMM (256 workgroups, each streams a private
// 256 KB weight range + a 64 KB activation block; its first weight bytes arrive a full HBM round trip after kernel entry) -> ...
// Three such edges per step.  The idea: the small kernel gets 256 extra workgroups; extra workgroup j touches the first HEAD bytes of
// the weight range of consumer workgroup j.  Workgroups are dealt to the 8 XCDs round-robin by their index, so an extra workgroup
// placed at an index congruent to j modulo 8 warms THE L2 that consumer workgroup j will read through.  The consumer then finds
// its first loads in L2 and the stream behind them already rolling.
//
// Measured: the time of one (small -> GEMM-like consumer) edge in a long dependent chain, with / without the warming workgroups,
// HEAD = 4 .. 64 KB per consumer workgroup, weights cold (rotating through 8 x 64 MB: every launch from HBM) or repeated (one
// 64 MB matrix: Infinity-Cache resident, as a decode loop's weights are), as ONE chain and as TWO chains on two streams (the SCST
// rollout pair).
//
//   hipcc --offload-arch=gfx950 -O3 -o l2_warm l2_warm.hip && ./l2_warm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CONS_WGS = 256;                 // consumer workgroups (one per CU)
constexpr int W_F4 = 256 * 1024 / 16;         // float4 of private weights per consumer workgroup (256 KB)
constexpr int ACT_F4 = 64 * 1024 / 16;        // float4 of its activation block (64 KB, shared by the workgroups of a k range)
constexpr int SMALL_WGS = 256;                // the small kernel's own workgroups (64 rows x 4, like lstm_point_gw_kernel)

// the small kernel: reads 16.8 MB of "slabs" over its own workgroups and writes 1 MB of activations (what lstm_point_gw_kernel does at
// 64 rows); workgroups >= SMALL_WGS (when launched) warm the head of consumer workgroup (blockIdx - SMALL_WGS)'s weight range
__global__ __launch_bounds__(256) void small_kernel(const f32x4* __restrict__ slabs, f32x4* __restrict__ act, const f32x4* __restrict__ w_next,
                                                    int head_f4, float* sink) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= SMALL_WGS) {
        const f32x4* p = w_next + (size_t)(b - SMALL_WGS) * W_F4;
        f32x4 s = {0, 0, 0, 0};
        for (int i = tid; i < head_f4; i += 256) s += p[i];
        if (s[0] == 12345.678f) sink[0] = 1.f;
        return;
    }
    f32x4 s = {0, 0, 0, 0};
    const f32x4* p = slabs + (size_t)b * 16 * 256 + tid;           // 16 slabs x 256 float4 per workgroup = 64 KB
    f32x4 v[16];
#pragma unroll
    for (int z = 0; z < 16; ++z) v[z] = p[(size_t)z * 256];
#pragma unroll
    for (int z = 0; z < 16; ++z) s += v[z];
    act[(size_t)b * 256 + tid] = s;
}

// the consumer: activation block first (64 KB from L2), then the private weight range with 16 independent 16-byte loads in
// flight per lane -- the resident GEMM's data movement without its arithmetic
__global__ __launch_bounds__(256) void consumer_kernel(const f32x4* __restrict__ w, const f32x4* __restrict__ act, f32x4* __restrict__ out) {
    const int b = blockIdx.x, tid = threadIdx.x;
    f32x4 s = {0, 0, 0, 0};
    const f32x4* ap = act + (size_t)(b % 16) * ACT_F4 + tid;
#pragma unroll
    for (int i = 0; i < ACT_F4 / 256; ++i) s += ap[(size_t)i * 256];
    const f32x4* wp = w + (size_t)b * W_F4 + tid;
    for (int i = 0; i < W_F4 / 256; i += 16) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(wp + (size_t)(i + u) * 256);
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    out[(size_t)b * 256 + tid] = s;
}

int main() {
    const size_t w_bytes = (size_t)CONS_WGS * W_F4 * 16;            // 64 MB per matrix
    const int NW = 8;
    std::vector<f32x4*> w(NW);
    for (int i = 0; i < NW; ++i) { CK(hipMalloc(&w[i], w_bytes)); CK(hipMemset(w[i], 0, w_bytes)); }
    f32x4 *slabs, *act[2], *out[2];
    float* sink;
    CK(hipMalloc(&slabs, (size_t)SMALL_WGS * 16 * 256 * 16));
    CK(hipMemset(slabs, 0, (size_t)SMALL_WGS * 16 * 256 * 16));
    for (int c = 0; c < 2; ++c) { CK(hipMalloc(&act[c], (size_t)16 * ACT_F4 * 16)); CK(hipMalloc(&out[c], (size_t)CONS_WGS * 256 * 16)); }
    CK(hipMalloc(&sink, 16));
    hipStream_t st[2];
    for (int c = 0; c < 2; ++c) CK(hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking));
    hipEvent_t e0, e1, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
    const int EDGES = 240;
    // chains: 1 or 2 streams; every edge = small kernel (+ warming workgroups) -> consumer on the next weight matrix
    auto run = [&](int chains, int head_kb, bool cold) -> double {
        const int head_f4 = head_kb * 1024 / 16;
        auto enqueue = [&]() {
            for (int i = 0; i < EDGES; ++i)
                for (int c = 0; c < chains; ++c) {
                    const f32x4* wn = cold ? w[(2 * i + c) % NW] : w[c];
                    hipLaunchKernelGGL(small_kernel, dim3(SMALL_WGS + (head_kb ? CONS_WGS : 0)), dim3(256), 0, st[c], slabs, act[c], wn, head_f4, sink);
                    hipLaunchKernelGGL(consumer_kernel, dim3(CONS_WGS), dim3(256), 0, st[c], wn, act[c], out[c]);
                }
        };
        enqueue();                              // warm-up
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        if (chains == 2) CK(hipStreamWaitEvent(st[1], e0, 0));
        enqueue();
        if (chains == 2) { CK(hipEventRecord(ej, st[1])); CK(hipStreamWaitEvent(st[0], ej, 0)); }
        CK(hipEventRecord(e1, st[0]));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / EDGES;
    };
    printf("us per edge (small kernel -> 256 x (64 KB activations + 256 KB weights)); a decode step has three such edges\n");
    printf("%-8s %-22s %10s", "chains", "weights", "no warm");
    const int heads[] = {4, 8, 16, 32, 64};
    for (int h : heads) printf("   %3d KB", h);
    printf("\n");
    for (int chains = 1; chains <= 2; ++chains)
        for (int cold = 0; cold < 2; ++cold)
            for (int rep = 0; rep < 2; ++rep) {
                printf("%-8d %-22s %10.2f", chains, cold ? "cold (8 x 64 MB)" : "repeated (in MALL)", run(chains, 0, cold));
                for (int h : heads) printf("   %6.2f", run(chains, h, cold));
                printf("\n");
            }
    // the consumer alone, back to back on one stream (no small kernel in between): the edge's lower bound
    {
        for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(consumer_kernel, dim3(CONS_WGS), dim3(256), 0, st[0], w[i % NW], act[0], out[0]);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < EDGES; ++i) hipLaunchKernelGGL(consumer_kernel, dim3(CONS_WGS), dim3(256), 0, st[0], w[i % NW], act[0], out[0]);
        CK(hipEventRecord(e1, st[0]));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("consumer alone, cold weights, back to back: %.2f us per launch (%.2f TB/s)\n", ms * 1e3 / EDGES, (double)w_bytes / (ms * 1e-3 / EDGES) / 1e12);
    }
    return 0;
}

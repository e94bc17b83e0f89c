// Round 4 microbenchmark: what does it cost a workgroup to read the WHOLE activation matrix of a decoder-step
// GEMM through L2 while it streams its private 256 KB of weights from HBM?
//
// The alternative decomposition under test: gate-interleaved LSTM weights, one workgroup per CU owns 4 hidden units x 4 gates (16
// weight rows) x the FULL K, its four waves split K and reduce through LDS, the pointwise stage runs in the epilogue -- no split-K
// slabs, no pointwise launch.  Its price: every one of the 256 workgroups reads all activations (64 rows x K = 4096: 1 MB as fp32,
// 1.5 MB as three producer-written bf16 planes) instead of today's 64 KB block.  DESIGN.md's "14 B per cycle and CU whatever the
// source" was measured at kernel entry with cold L2s; this measures the steady state: a producer kernel writes the activations
// (as the previous decoder-step kernel would), then 256 workgroups of 256 threads each read all of them (16-byte loads, eight in
// flight per lane) -- alone, and interleaved 4 : 1 / 6 : 1 with a once-read weight stream that rotates through 6 x 64 MB (no MALL hits).
//
//   hipcc --offload-arch=gfx950 -O3 -o act_through_l2 act_through_l2.hip && ./act_through_l2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void produce(f32x4* act, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) act[i] = (f32x4){v, v + 1, v + 2, v + 3};
}

// per workgroup: WQ float4 of private weights per thread (stride 256 threads), AQ float4 of the shared activations per thread;
// the two streams are interleaved in groups of 8 independent loads (ratio AQ : WQ), summed so that nothing is dead code
template <int UN>
__global__ __launch_bounds__(256) void consume(const f32x4* __restrict__ w, int wq, const f32x4* __restrict__ act, int aq, float* sink,
                                               unsigned long long* cycles) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const f32x4* wp = w + (size_t)blockIdx.x * wq * 256 + threadIdx.x;
    const f32x4* ap = act + threadIdx.x;
    f32x4 s = {0, 0, 0, 0};
    int wi = 0, ai = 0;
    // act groups per weight group
    const int ratio = wq > 0 ? (aq + wq - 1) / wq : 0;
    while (wi < wq || ai < aq) {
        if (wi < wq) {
            f32x4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) v[u] = __builtin_nontemporal_load(wp + (size_t)(wi + u) * 256);
#pragma unroll
            for (int u = 0; u < UN; ++u) s += v[u];
            wi += UN;
        }
        const int todo = wq > 0 ? ratio : 1;
        for (int r = 0; r < todo && ai < aq; ++r) {
            f32x4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) v[u] = ap[(size_t)(ai + u) * 256];
#pragma unroll
            for (int u = 0; u < UN; ++u) s += v[u];
            ai += UN;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = 1.f;
}

int main() {
    const int NWG = 256;
    const size_t WBYTES = 64ull << 20;          // 256 KB per workgroup
    const int NROT = 6;
    std::vector<f32x4*> w(NROT);
    for (auto& p : w) { CK(hipMalloc(&p, WBYTES)); CK(hipMemset(p, 0, WBYTES)); }
    f32x4* act; CK(hipMalloc(&act, 4 << 20));
    float* sink; CK(hipMalloc(&sink, 16));
    unsigned long long* cyc; CK(hipMalloc(&cyc, sizeof(unsigned long long) * NWG));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; size_t wbytes_wg, abytes; };
    const Case cases[] = {
        {"weights only, 256 KB per workgroup (HBM, once-read)", 256 << 10, 0},
        {"activations only, 64 KB (today's block)", 0, 64 << 10},
        {"activations only, 1 MB (64 rows x 4096 fp32)", 0, 1 << 20},
        {"activations only, 1.5 MB (three bf16 planes)", 0, 3 << 19},
        {"weights + 64 KB activations", 256 << 10, 64 << 10},
        {"weights + 1 MB activations", 256 << 10, 1 << 20},
        {"weights + 1.5 MB activations", 256 << 10, 3 << 19},
        {"weights + 768 KB activations (K = 3072 fp32)", 256 << 10, 3 << 18},
    };
    printf("%-58s %9s %9s %12s %12s\n", "case", "us", "us(min)", "cyc/WG(med)", "B/clk/CU");
    for (const Case& c : cases) {
        const int wq = (int)(c.wbytes_wg / 16 / 256), aq = (int)(c.abytes / 16 / 256);
        std::vector<float> us;
        std::vector<unsigned long long> h(NWG);
        unsigned long long medc = 0;
        for (int it = 0; it < 24; ++it) {
            // the producer rewrites the activations (other CUs / XCDs than the readers': the kernel boundary writes them back)
            hipLaunchKernelGGL(produce, dim3(256), dim3(256), 0, 0, act, (size_t)(4 << 20) / 16, (float)it);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(consume<8>, dim3(NWG), dim3(256), 0, 0, w[it % NROT], wq, act, aq, sink, cyc);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 4) us.push_back(ms * 1e3f);
            CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * NWG, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            medc = h[NWG / 2];
        }
        std::sort(us.begin(), us.end());
        const double bytes = (double)c.wbytes_wg + (double)c.abytes;
        printf("%-58s %9.2f %9.2f %12llu %12.1f\n", c.name, us[us.size() / 2], us[0], medc, bytes / (double)medc);
    }
    return 0;
}

// dev tool: per-kernel cost of dependent tiny kernels, eager vs hipGraph
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void mid(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int which = 0; which < 2; ++which) {
            auto launch = [&](hipStream_t s) {
                if (which == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
                else hipLaunchKernelGGL(mid, dim3(256), dim3(256), 0, s, d, 65536);
            };
            float ms;
            if (mode == 0) {
                for (int i = 0; i < 100; ++i) launch(st);
                hipStreamSynchronize(st);
                hipEventRecord(e0, st);
                for (int i = 0; i < N; ++i) launch(st);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            } else {
                hipGraph_t g; hipGraphExec_t ex;
                hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
                for (int i = 0; i < N; ++i) launch(st);
                hipStreamEndCapture(st, &g);
                hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
                hipGraphLaunch(ex, st); hipStreamSynchronize(st);
                hipEventRecord(e0, st);
                hipGraphLaunch(ex, st);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%s %s: %.2f us per kernel\n", mode ? "graph" : "eager", which ? "mid(256 WGs)" : "tiny", ms * 1e3 / N);
        }
    }
    return 0;
}

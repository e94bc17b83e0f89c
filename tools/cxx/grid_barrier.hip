// dev tool: cost of a grid-wide barrier inside one persistent kernel (the building block of a per-step decode kernel with
// phases instead of one launch per GEMM / pointwise stage).  One workgroup per CU (and two), a
// monotonically increasing arrival counter, bounded spin (a stuck barrier sets an error flag and every wave leaves).
// Each round also passes a value between workgroups through global memory to check visibility across the XCDs' L2s.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned nwg, unsigned& gen, unsigned* err) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += 1;
        const unsigned target = gen * nwg;
        __atomic_thread_fence(__ATOMIC_RELEASE);              // agent scope: write back this XCD's dirty L2 lines once
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 2000000u) {          // every workgroup runs into its own limit: all waves leave
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);              // invalidate once, after the wait
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// Flag barrier: no read-modify-write on a shared word.  Every workgroup publishes its generation in its own slot and every
// thread polls a few slots (256 threads cover 256 workgroups in one load each).
__device__ __forceinline__ bool flag_barrier(unsigned* flags, int stride, unsigned nwg, unsigned& gen, unsigned* err) {
    __shared__ int bad_;
    if (threadIdx.x == 0) bad_ = 0;
    __syncthreads();
    gen += 1;
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_store(flags + (size_t)blockIdx.x * stride, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (unsigned i = threadIdx.x; i < nwg; i += blockDim.x) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags + (size_t)i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            if (++spins > 2000000u) {          // every workgroup runs into its own limit: all waves leave
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad_ = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    __syncthreads();
    return bad_ == 0;
}

// rounds x { publish, barrier, read the neighbour's value, [work] }
__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, unsigned* err, unsigned* box, unsigned* bad, int rounds, int work,
                                                    const float* __restrict__ w, float* __restrict__ sink, unsigned* flags, int stride) {
    const unsigned nwg = gridDim.x, me = blockIdx.x;
    unsigned gen = 0;
    float acc = 0.f;
    for (int r = 1; r <= rounds; ++r) {
        if (threadIdx.x == 0) __hip_atomic_store(box + me, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(stride ? flag_barrier(flags, stride, nwg, gen, err) : grid_barrier(counter, nwg, gen, err))) return;
        if (threadIdx.x == 0) {
            const unsigned v = __hip_atomic_load(box + (me + nwg / 2 + 1) % nwg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (unsigned)r && v != (unsigned)r + 1) atomicAdd(bad, 1u);
        }
        // optional streaming work between barriers (per workgroup: work * 256 float4 loads of its own slice)
        for (int i = 0; i < work; ++i) {
            const float4 x = reinterpret_cast<const float4*>(w)[((size_t)me * work + i) * 256 + threadIdx.x];
            acc += x.x + x.y + x.z + x.w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned *counter, *err, *box, *bad, *flags;
    CHECK(hipMalloc(&flags, 4096 * 64 * 4));
    float *w, *sink;
    CHECK(hipMalloc(&counter, 256));
    CHECK(hipMalloc(&err, 256));
    CHECK(hipMalloc(&bad, 256));
    CHECK(hipMalloc(&box, 4096 * 4));
    CHECK(hipMalloc(&w, (size_t)512 * 64 * 256 * 16));
    CHECK(hipMemset(w, 0, (size_t)512 * 64 * 256 * 16));
    CHECK(hipMalloc(&sink, 256));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%d CUs\n", cus);
    for (int stride : {0, 1, 32})
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        for (int work : {0, 16, 64}) {
            const int nwg = cus * per_cu, rounds = 2000;
            float base_ms = 0.f, ms = 0.f;
            for (int pass = 0; pass < 2; ++pass) {          // pass 0: 1 round (launch + ramp), pass 1: `rounds`
                const int n = pass ? rounds : 1;
                CHECK(hipMemsetAsync(counter, 0, 4, st));
                CHECK(hipMemsetAsync(err, 0, 4, st));
                CHECK(hipMemsetAsync(bad, 0, 4, st));
                CHECK(hipMemsetAsync(box, 0, 4096 * 4, st));
                CHECK(hipMemsetAsync(flags, 0, 4096 * 64 * 4, st));
                CHECK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(barrier_loop, dim3(nwg), dim3(256), 0, st, counter, err, box, bad, n, work, w, sink, flags, stride);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(pass ? &ms : &base_ms, e0, e1));
            }
            unsigned herr = 0, hbad = 0;
            CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            printf("%s  %d workgroups (%d per CU), %3d KB streamed per workgroup per round: %.2f us per round%s%s\n",
                   stride == 0 ? "counter      " : (stride == 1 ? "flags packed " : "flags 128 B  "), nwg, per_cu, work * 4, (ms - base_ms) * 1e3 / (rounds - 1), herr ? "  [BARRIER TIMED OUT]" : "", hbad ? "  [STALE READS]" : "");
            if (herr) return 2;
        }
    }
    return 0;
}

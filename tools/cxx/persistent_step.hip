// dev tool (round 6): what would ONE resident launch per decode step buy over the nine dependent launches of today's
// greedy BUTD step (BUTD_Model.py:172-183 at 64 rows)?  Measured on the step's DATA MOVEMENT, phase for phase with the byte counts and
// the workgroup decomposition of the shipped kernels (csrc/butd.hip: Butd::step + greedy_select), no arithmetic: every variant below
// moves the same bytes through the same number of workgroups, so the differences are launch edges against grid barriers and what
// cross-phase prefetch hides.  A real persistent kernel can only be slower than its emulation here (MFMA / VALU work, registers
// shared between phases), the launch chain is the shipped one minus its arithmetic.
//
// Phases of a step (jobs x [shared KB re-read through L2 | private KB read once | KB written] per job):
//   0 TD gates GEMM   192 x [64 | 256 | 64]   k range 256 deep x 256 columns, K = 3072, N = 4096 (gemm_resident_x3_kernel)
//   1 TD pointwise    256 x [ 0 |  53 |  6]   12 slabs summed (lstm_point_gw_kernel)
//   2 dec_att GEMM     64 x [32 |  64 | 32]   64 x 1024 x 1024 (gemm_nt_kernel, 8 column tiles x 8 splits)
//   3 scores          192 x [32 |  48 |  1]   att_scores_kernel: (row, 3 parts)
//   4 softmax + ctx   256 x [ 0 |  72 |  2]   att_ctx_kernel: (row, 4 column parts)
//   5 LM gates GEMM   256 x [64 | 256 | 64]   K = 4096
//   6 LM pointwise    256 x [ 0 |  69 |  7]   16 slabs
//   7 predict GEMM    160 x [64 | 256 | 64]   K = 1024, N = 10112
//   8 token choice    256 x [ 0 |  40 |  1]   four slabs of a quarter row (launch chain: 64 x 1024 threads x 160 KB as shipped)
// = 287 MB per step as moved (SURVEY 8d's algorithmic 227.9 MB + the split-K slabs written and read back).
//
// Variants: A  nine launches per step, 20 steps captured into one hipGraph (today's structure)
//           B  one launch per 20 steps, 256 resident workgroups, an XCD-hierarchical grid barrier between phases
//           C  B + every workgroup issues the first 64 KB of its NEXT GEMM phase's private stream before it waits at the barrier
//           D  B with the barrier removed (WRONG results in a real kernel: the lower bound, phases back to back without any edge)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Phase { int jobs; int shared_f4, priv_f4, out_f4; int shared_mod; };     // per job, in float4 per THREAD (256 threads)
// KB per job / (256 threads x 16 B = 4 KB per float4-per-thread)
__constant__ Phase PH[9];
static Phase PH_HOST[9] = {
    {192, 16, 64, 16, 12}, {256, 0, 13, 2, 1}, {64, 8, 16, 8, 8}, {192, 8, 12, 1, 64}, {256, 0, 18, 1, 1},
    {256, 16, 64, 16, 16}, {256, 0, 17, 2, 1}, {160, 16, 64, 16, 4}, {256, 0, 10, 1, 1}};
// `persistent_step half`: the gate GEMMs on 512-deep k ranges x ONE 128-column tile per workgroup (half the slabs: 6 / 8 instead of 12 / 16;
// twice the activation block per workgroup) and the pointwise phases over half the slab bytes -- EXPERIMENTS round 6 section 2
static const Phase PH_HALF[9] = {
    {192, 32, 64, 8, 6}, {256, 0, 7, 2, 1}, {64, 8, 16, 8, 8}, {192, 8, 12, 1, 64}, {256, 0, 18, 1, 1},
    {256, 32, 64, 8, 8}, {256, 0, 9, 2, 1}, {160, 16, 64, 16, 4}, {256, 0, 10, 1, 1}};

struct Bufs {
    const f32x4* priv[9];      // private streams, job j at + j * priv_f4 * 256
    const f32x4* shared[9];    // shared blocks, job j reads block (j % shared_mod) at + (j % shared_mod) * shared_f4 * 256
    f32x4* out[9];             // job j writes at + j * out_f4 * 256
};

struct Bar { unsigned xcd_cnt[8 * 32]; unsigned top[32]; unsigned gen[8 * 32]; unsigned err[32]; };

// XCD-hierarchical barrier (MI355X_MICROARCH.md, barrier-xcd): workgroups of one XCD (blockIdx % 8) arrive on that XCD's counter; its last
// arriver goes to the top counter; the last of those publishes the generation into eight per-XCD words, each polled by its own 32
__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned& gen) {
    __shared__ int ok_;
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += 1;
        const unsigned x = blockIdx.x & 7, per = gridDim.x / 8;
        __atomic_thread_fence(__ATOMIC_RELEASE);                                        // agent scope: this workgroup's stores
        const unsigned old = __hip_atomic_fetch_add(&b->xcd_cnt[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gen * per - 1) {
            const unsigned t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gen * 8 - 1)
                for (int i = 0; i < 8; ++i) __hip_atomic_store(&b->gen[i * 32], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(&b->gen[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            if (++spins > 4000000u) { __hip_atomic_store(&b->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        ok_ = good;
    }
    __syncthreads();
    return ok_ != 0;
}

// one job: shared block first (L2), then the private stream with 16 x 16-byte loads in flight per lane, then the output
__device__ __forceinline__ void do_job(const Bufs& B, int ph, int j, const f32x4 (&pre)[16], bool have_pre, int nthreads) {
    const Phase p = PH[ph];
    const int tid = threadIdx.x;
    const int scale = nthreads / 256;                       // the 1024-thread token-choice launch of variant A: the caller passes 4 jobs' worth per workgroup
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (p.shared_f4) {
        const f32x4* sp = B.shared[ph] + (size_t)(j % p.shared_mod) * p.shared_f4 * 256 + tid;
        for (int i = 0; i < p.shared_f4; i += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp[(size_t)(i + u) * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    const f32x4* pp = B.priv[ph] + (size_t)j * p.priv_f4 * 256 * scale + tid;
    int i = 0;
    if (have_pre) {
#pragma unroll
        for (int u = 0; u < 16; ++u) s += pre[u];
        i = 16;
    }
    for (; i + 16 <= p.priv_f4; i += 16) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(pp + (size_t)(i + u) * nthreads);
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    if (i < p.priv_f4) {                                   // the remainder as one group as well: all its loads in flight together
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (i + u < p.priv_f4) v[u] = __builtin_nontemporal_load(pp + (size_t)(i + u) * nthreads);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (i + u < p.priv_f4) s += v[u];
    }
    f32x4* op = B.out[ph] + (size_t)j * p.out_f4 * 256 * scale + tid;
    for (int q = 0; q < p.out_f4; ++q) op[(size_t)q * nthreads] = s;
}

__global__ __launch_bounds__(256) void phase_kernel(Bufs B, int ph) {
    f32x4 pre[16];
    do_job(B, ph, blockIdx.x, pre, false, 256);
}
__global__ __launch_bounds__(1024) void select_kernel_1024(Bufs B) {          // the shipped token choice: one 1024-thread workgroup per row
    f32x4 pre[16];
    do_job(B, 8, blockIdx.x, pre, false, 1024);
}

// mode 0: barrier between phases; 1: + prefetch of the next GEMM phase's first 64 KB in front of the barrier; 2: no barrier at all
__global__ __launch_bounds__(256) void persistent_kernel(Bufs B, Bar* bar, int steps, int mode) {
    unsigned gen = 0;
    const int j = blockIdx.x;
    f32x4 pre[16];
    bool have = false;
    for (int t = 0; t < steps; ++t)
        for (int ph = 0; ph < 9; ++ph) {
            if (j < PH[ph].jobs) do_job(B, ph, j, pre, have, 256);
            have = false;
            if (mode == 1) {
                const int nx = (ph + 1) % 9;
                if ((nx == 0 || nx == 5 || nx == 7) && j < PH[nx].jobs && !(nx == 0 && t + 1 == steps)) {
                    const f32x4* pp = B.priv[nx] + (size_t)j * PH[nx].priv_f4 * 256 + threadIdx.x;
#pragma unroll
                    for (int u = 0; u < 16; ++u) pre[u] = __builtin_nontemporal_load(pp + (size_t)u * 256);
                    have = true;
                }
            }
            if (mode != 2 && !grid_barrier(bar, gen)) return;
        }
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'h') { for (int i = 0; i < 9; ++i) PH_HOST[i] = PH_HALF[i]; printf("table: gate GEMMs on 512-deep k ranges x one tile (half the slabs)\n"); }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(PH), PH_HOST, sizeof(PH_HOST)));
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("%d CUs\n", cus);
    if (cus != 256) { printf("this tool assumes 256 CUs (one resident workgroup each)\n"); return 1; }
    Bufs B = {};
    double step_bytes = 0;
    std::vector<void*> allocs;
    for (int ph = 0; ph < 9; ++ph) {
        const Phase& p = PH_HOST[ph];
        const int scale = 1;
        size_t pb = (size_t)p.jobs * p.priv_f4 * 256 * 16 * scale, sb = (size_t)p.shared_mod * (p.shared_f4 ? p.shared_f4 : 1) * 256 * 16,
               ob = (size_t)p.jobs * p.out_f4 * 256 * 16;
        void *a, *b, *c;
        CK(hipMalloc(&a, pb)); CK(hipMemset(a, 0, pb));
        CK(hipMalloc(&b, sb)); CK(hipMemset(b, 0, sb));
        CK(hipMalloc(&c, ob)); CK(hipMemset(c, 0, ob));
        allocs.push_back(a); allocs.push_back(b); allocs.push_back(c);
        B.priv[ph] = (const f32x4*)a; B.shared[ph] = (const f32x4*)b; B.out[ph] = (f32x4*)c;
        step_bytes += (double)p.jobs * (p.priv_f4 + p.out_f4) * 4096.0 + (double)p.shared_mod * p.shared_f4 * 4096.0;
    }
    printf("bytes per step (private + written once, shared blocks once): %.1f MB\n", step_bytes / 1e6);
    Bar* bar;
    CK(hipMalloc(&bar, sizeof(Bar)));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int STEPS = 20, REPS = 30;

    // ---- variant A: nine launches per step, 20 steps in one captured graph
    auto enqueue_steps = [&](hipStream_t s, int skip_phase) {
        for (int t = 0; t < STEPS; ++t)
            for (int ph = 0; ph < 9; ++ph) {
                if (ph == skip_phase) continue;
                if (ph == 8) hipLaunchKernelGGL(select_kernel_1024, dim3(64), dim3(1024), 0, s, B);
                else hipLaunchKernelGGL(phase_kernel, dim3(PH_HOST[ph].jobs), dim3(256), 0, s, B, ph);
            }
    };
    auto time_graph = [&](int skip_phase, double* us) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        enqueue_steps(st, skip_phase);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        *us = ms * 1e3 / (REPS * STEPS);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return 0;
    };
    double usA = 0;
    if (time_graph(-1, &usA)) return 1;
    printf("A  nine launches per step (hipGraph of 20 steps):            %7.2f us per step  = %.2f TB/s = %.3f of 8 TB/s\n", usA, step_bytes / usA / 1e6, step_bytes / usA / 8e6);
    // what each phase contributes to A: the graph without it
    static const char* names[9] = {"TD gates GEMM", "TD pointwise", "dec_att GEMM", "scores", "softmax + ctx", "LM gates GEMM", "LM pointwise", "predict GEMM", "token choice"};
    for (int ph = 0; ph < 9; ++ph) {
        double u = 0;
        if (time_graph(ph, &u)) return 1;
        printf("   A without phase %d (%-14s): %7.2f us per step (the phase and its edge cost %5.2f us)\n", ph, names[ph], u, usA - u);
    }

    // ---- variants B, C, D: one resident launch per 20 steps
    for (int mode : {0, 1, 2}) {
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
            hipLaunchKernelGGL(persistent_kernel, dim3(256), dim3(256), 0, st, B, bar, 2, mode);            // warm
            CK(hipStreamSynchronize(st));
            CK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent_kernel, dim3(256), dim3(256), 0, st, B, bar, STEPS * 4, mode);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            Bar hb;
            CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
            if (hb.err[0]) { printf("mode %d: BARRIER TIMED OUT\n", mode); return 2; }
            const double us = ms * 1e3 / (STEPS * 4);
            if (us < best) best = us;
        }
        printf("%s %7.2f us per step  = %.2f TB/s = %.3f of 8 TB/s  (%.2fx A)\n",
               mode == 0 ? "B  one resident launch, grid barrier between phases:          " :
               mode == 1 ? "C  B + next GEMM phase's first 64 KB issued before the wait: " :
                           "D  B without any barrier (lower bound, not a valid program):  ", best, step_bytes / best / 1e6, step_bytes / best / 8e6, best / usA);
    }
    // the barrier alone (nothing between barriers)
    {
        // 9 x 80 barriers of an otherwise empty kernel: reuse persistent_kernel with zero-job phases is not possible; time B - D instead
    }
    for (void* p : allocs) (void)hipFree(p);
    return 0;
}

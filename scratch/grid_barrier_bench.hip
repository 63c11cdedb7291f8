// Bounded experiment for VERDICT r5 item 5b (one PERSISTENT kernel per contact solve instead of four dependent launches per
// Newton iteration): what does a grid-wide barrier cost on this part, 8 XCDs with an L2 each?  A persistent solve needs four
// of them per iteration (tile -> node_dir -> ls -> decide), each with the data of the phase before it visible to every
// XCD, i.e. a release (L2 write-back) before the arrival and an acquire (L2 invalidate) behind the wait.
// Measured here: N barriers of G workgroups (one per CU, or two), two-level arrival (a counter per XCD, 128 bytes apart,
// then one word), monotonic generation, with and without the fences; against N empty kernel launches back to back.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/grid_barrier_bench scratch/grid_barrier_bench.hip && /tmp/grid_barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FENCE>
__global__ __launch_bounds__(256) void k_barriers(unsigned* xcd_cnt, unsigned* all_cnt, int n, int per_xcd, float* sink) {
    const int xcd = blockIdx.x & 7;
    float acc = 0.f;
    for (int it = 0; it < n; ++it) {
        acc += (float)it * 1e-9f;   // (a phase's work would go here)
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCE) __threadfence();   // release: this workgroup's stores of the phase reach memory (L2 write-back)
            const unsigned old = __hip_atomic_fetch_add(&xcd_cnt[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)(per_xcd * (it + 1) - 1))
                __hip_atomic_fetch_add(all_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (bounded: a workgroup that is not resident would never arrive -- give up after 0.5 s of the 100 MHz wall clock,
            // tell everybody, and let the grid drain)
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(all_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(8 * (it + 1))) {
                if (__hip_atomic_load(all_cnt + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > 50000000ull) {
                    __hip_atomic_store(all_cnt + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (__hip_atomic_load(all_cnt + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) n = 0;   // (leave the loop)
            if (FENCE) __threadfence();   // acquire: the next phase must not read stale lines of this XCD's L2
        }
        __syncthreads();
        if (__hip_atomic_load(all_cnt + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // (uniform: read behind the barrier)
    }
    if (acc == 123.f) sink[0] = acc;
}
__global__ void k_empty(float* sink) { if (threadIdx.x == 9999) sink[0] = 1.f; }

int main() {
    unsigned *xcd_cnt, *all_cnt;
    float* sink;
    CK(hipMalloc((void**)&xcd_cnt, 8 * 128));
    CK(hipMalloc((void**)&all_cnt, 128));
    CK(hipMalloc((void**)&sink, 128));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 2000;
    for (int G : {256, 512}) {
        for (int fence = 0; fence < 2; ++fence) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(xcd_cnt, 0, 8 * 128)); CK(hipMemset(all_cnt, 0, 128));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a));
                if (fence) hipLaunchKernelGGL(k_barriers<1>, dim3(G), dim3(256), 0, 0, xcd_cnt, all_cnt, N, G / 8, sink);
                else hipLaunchKernelGGL(k_barriers<0>, dim3(G), dim3(256), 0, 0, xcd_cnt, all_cnt, N, G / 8, sink);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, a, b));
                unsigned gave_up = 0;
                CK(hipMemcpy(&gave_up, all_cnt + 16, 4, hipMemcpyDeviceToHost));
                printf("grid barrier, %d workgroups of 256, %s: %.2f us per barrier%s\n", G, fence ? "release + acquire fences" : "no fences", ms * 1e3 / N,
                       gave_up ? "  (GAVE UP: not all workgroups resident)" : "");
            }
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("kernel boundary: %d empty launches of 256 workgroups back to back: %.2f us per launch\n", N, ms * 1e3 / N);
    }
    return 0;
}

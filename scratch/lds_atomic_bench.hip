// Micro-benchmark: LDS atomic / read-modify-write throughput on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int stride) {
    __shared__ __attribute__((aligned(16))) float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) buf[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // each wave touches 64 distinct words, pattern like the P2G epilogue (stride between lanes)
    int idx = (wv * 640 + lane * stride) & 8191;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) __hip_atomic_fetch_add(&buf[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<int*>(&buf[idx]), (int)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) { float t = buf[idx]; buf[idx] = t + v; }
        if (MODE == 3) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(buf) + ((idx & 8191) >> 1), (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 4) buf[idx] = v;
        if (MODE == 5) __hip_atomic_fetch_add(reinterpret_cast<double*>(buf) + ((idx & 8191) >> 1), (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        idx = (idx + 64 * stride + 4) & 8191;
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x * 7];
}
template <int MODE>
void run(const char* name, int stride) {
    float* d; hipMalloc(&d, 512 * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    k<MODE><<<512, 512>>>(d, 10, stride);
    hipEventRecord(a);
    k<MODE><<<512, 512>>>(d, iters, stride);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // 512 WGs on 256 CUs -> 2 WG/CU, 16 waves/CU; per CU wave-instructions = 16 * iters
    double cyc_per_instr = ms * 1e-3 * 2.4e9 / (16.0 * iters);
    printf("%-28s stride %d: %.3f ms  -> %.1f cycles per wave-instr per CU (at 2.4 GHz)\n", name, stride, ms, cyc_per_instr);
    hipFree(d);
}
int main() {
    for (int stride : {1, 4}) {
        run<0>("ds_add_f32", stride);
        run<1>("ds_add_u32", stride);
        run<2>("ds_read+add+ds_write", stride);
        run<3>("ds_add_u64", stride);
        run<4>("ds_write_b32", stride);
        run<5>("ds_add_f64", stride);
    }
    return 0;
}

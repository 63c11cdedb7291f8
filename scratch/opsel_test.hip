// Checks the inline assembly of k_g2p's z broadcast (mpm_step.h, MPM_G2P_ZPAIR) against scalar arithmetic, bit for bit.
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scratch/opsel_test.hip -o scratch/opsel_test && scratch/opsel_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out_asm, float* out_ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = in + (size_t)i * 12;
    const f32x2 W0 = {p[0], p[1]}, W1 = {p[2], p[3]}, W2 = {p[4], p[5]};
    const f32x2 z0 = {p[6], p[7]}, z1 = {p[8], p[9]}, z2 = {p[10], p[11]};
    f32x2 AA;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\ts_nop 0\n\t"
        "v_pk_fma_f32 %0, %3, %4, %0 op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
        "v_pk_fma_f32 %0, %5, %6, %0 op_sel_hi:[1,0,1]"
        : "=&v"(AA) : "v"(W0), "v"(z0), "v"(W1), "v"(z1), "v"(W2), "v"(z2));
    out_asm[2 * i] = AA.x; out_asm[2 * i + 1] = AA.y;
    out_ref[2 * i] = __builtin_fmaf(W2.x, z2.x, __builtin_fmaf(W1.x, z1.x, W0.x * z0.x));
    out_ref[2 * i + 1] = __builtin_fmaf(W2.y, z2.x, __builtin_fmaf(W1.y, z1.x, W0.y * z0.x));
}
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc((size_t)n * 12 * 4);
    srand(1);
    for (size_t k = 0; k < (size_t)n * 12; ++k) h[k] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *d, *a, *r;
    hipMalloc(&d, (size_t)n * 48); hipMalloc(&a, (size_t)n * 8); hipMalloc(&r, (size_t)n * 8);
    hipMemcpy(d, h, (size_t)n * 48, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, r, n);
    float* ha = (float*)malloc((size_t)n * 8); float* hr = (float*)malloc((size_t)n * 8);
    hipMemcpy(ha, a, (size_t)n * 8, hipMemcpyDeviceToHost); hipMemcpy(hr, r, (size_t)n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t k = 0; k < (size_t)n * 2; ++k) bad += memcmp(&ha[k], &hr[k], 4) != 0;
    printf("opsel test: %zu of %zu values differ\n", bad, (size_t)n * 2);
    return bad != 0;
}

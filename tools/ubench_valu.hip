// Micro-benchmark: fp32 VALU issue rate on gfx950, scalar v_fma_f32 vs packed v_pk_fma_f32, at 1/2/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
__global__ void k_scalar(float *out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_packed(float *out, float a, float b) {
    f2 A = {a, a}, B = {b, b};
    f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    for (int i = 0; i < ITERS; i++) {
        x0 = __builtin_elementwise_fma(x0, A, B); x1 = __builtin_elementwise_fma(x1, A, B); x2 = __builtin_elementwise_fma(x2, A, B); x3 = __builtin_elementwise_fma(x3, A, B);
        x4 = __builtin_elementwise_fma(x4, A, B); x5 = __builtin_elementwise_fma(x5, A, B); x6 = __builtin_elementwise_fma(x6, A, B); x7 = __builtin_elementwise_fma(x7, A, B);
    }
    f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 1024 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {
        for (int packed = 0; packed < 2; packed++) {
            dim3 grid(256 * wps), block(256);
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (packed) hipLaunchKernelGGL(k_packed, grid, block, 0, 0, d, 1.0001f, 0.5f); else hipLaunchKernelGGL(k_scalar, grid, block, 0, 0, d, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts = 8.0 * ITERS;  // per wave
            double cyc = ms * 1e-3 * 2.4e9;
            double flops = (double)grid.x * 256 * insts * 2 * (packed ? 2 : 1);
            printf("waves/SIMD=%d %s: %.3f ms  cycles/wave-instr (per SIMD, @2.4GHz) = %.2f  TFLOP/s = %.1f\n", wps, packed ? "v_pk_fma_f32" : "v_fma_f32   ", ms, cyc / (insts * wps), flops / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

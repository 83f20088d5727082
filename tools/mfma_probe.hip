// Dev probe: sustained v_mfma_f32_32x32x2_f32 rate with operands in registers (no memory), to read the clock the chip
// holds under fp32-MFMA load.  hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 16 * NACC * 32 * 32 * 2 * 2;
        double mf_per_simd = (double)iters * 16 * NACC * (blocks * 4.0 / 1024.0);   // MFMAs issued per SIMD (if evenly spread)
        printf("NACC=%d blocks=%d iters=%d: %.1f us  %.1f TFLOP/s  implied clock %.2f GHz (64 cyc/MFMA/SIMD)\n", NACC, blocks, iters,
               ms * 1e3, flops / ms / 1e9, mf_per_simd * 64 / (ms * 1e-3) / 1e9);
    }
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    run<1>(256, 1000, d);      // 1 wave / SIMD, one dependent chain
    run<1>(512, 1000, d);      // 2 waves / SIMD
    run<2>(256, 1000, d);      // 1 wave / SIMD, two chains
    run<1>(256, 64, d);        // ~27 us kernel like one xpanel tile
    run<1>(512, 64, d);
    return 0;
}

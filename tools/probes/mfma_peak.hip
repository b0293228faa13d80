// Sustained fp32 MFMA rate of the chip under its real clocks: the ceiling the convolution kernels' TF/s should be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak.exe && tools/probes/mfma_peak.exe
// Variants: workgroups per CU (1 / 2 / 3), a barrier every 64 MFMAs (the conv kernel's chunk), 4 or 1 accumulator tiles per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool BAR>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64 / NACC; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        if (BAR) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
}

template <int NACC, bool BAR>
static void run(const char* name, int wg_per_cu, float* out) {
    const int iters = 20000;                                  // 64 MFMAs each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL((mfma_loop<NACC, BAR>), dim3(grid), dim3(256), 0, 0, out, 200, 1.f, 1.f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, BAR>), dim3(grid), dim3(256), 0, 0, out, iters / wg_per_cu, 1.f, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 * (iters / wg_per_cu) * 64.0 * 2 * 32 * 32 * 2;
        printf("%-34s wg/CU %d  %8.3f ms  %7.1f TF/s  (%.3f GHz-equivalent of 256 CU x 4 SIMD x 256 FLOP/clk)\n", name, wg_per_cu, ms,
               flop / ms * 1e-9, flop / ms * 1e-6 / (256.0 * 4 * 256));
    }
}

int main() {
    float* out; hipMalloc(&out, 4096);
    run<4, false>("4 acc tiles, no barrier", 1, out);
    run<4, false>("4 acc tiles, no barrier", 2, out);
    run<4, true>("4 acc tiles, barrier / 64 MFMA", 1, out);
    run<4, true>("4 acc tiles, barrier / 64 MFMA", 2, out);
    run<4, true>("4 acc tiles, barrier / 64 MFMA", 3, out);
    run<1, false>("1 acc tile (dependent chain)", 2, out);
    run<2, false>("2 acc tiles", 2, out);
    return 0;
}

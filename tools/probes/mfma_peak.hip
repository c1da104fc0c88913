// Sustained fp32 MFMA rate of the chip: every wave issues nothing but v_mfma_f32_32x32x2_f32 on register operands
// (four independent accumulators), two waves per SIMD.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_only(float *out, int iters, float a0, float b0)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float *out;
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 2048}) {
        const int iters = 4000;
        mfma_only<<<grid, 256>>>(out, 100, 1.0f, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_only<<<grid, 256>>>(out, iters, 1.0f, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 /*waves*/ * iters * 32 /*mfma*/ * 4096.0;
        printf("grid %4d x 256 threads: %8.3f ms  %7.1f TFLOP/s fp32 (32x32x2)\n", grid, ms, flops / ms / 1e9);
    }
    return 0;
}

// What the matrix pipe sustains on this chip: every CU runs WAVES waves of back-to-back v_mfma (4 independent accumulators per wave);
// a workgroup records its shader-clock cycles (s_memtime) and the constant 100 MHz counter (s_memrealtime) around the loop.
//   cycles / mfma / SIMD  -> is the pipe full (64 for f32 32x32x2, 32 for bf16 32x32x16: 16 / 8 passes x 4 cycles)
//   cycles / real time    -> the shader clock the chip actually holds under this load
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_clock.hip -o tools/_bin/mfma_clock && tools/_bin/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(1024) burn(int iters, unsigned long long *out, float *sink) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) a8[i] = (__bf16)a, b8[i] = (__bf16)b;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[2 * blockIdx.x] = c1 - c0, out[2 * blockIdx.x + 1] = r1 - r0;
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
}

template <int KIND>
static void run(const char *name, int waves, int iters, int reps, double flops_per_mfma, int pipe_cycles) {
    unsigned long long *d;
    float *sink;
    hipMalloc(&d, 256 * 2 * sizeof(unsigned long long));
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    burn<KIND><<<256, 64 * waves>>>(iters / 10 + 1, d, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) burn<KIND><<<256, 64 * waves>>>(iters, d, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), d, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, mhz;
    for (int i = 0; i < 256; ++i) cyc.push_back((double)h[2 * i]), mhz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] / 100.0));
    std::sort(cyc.begin(), cyc.end()), std::sort(mhz.begin(), mhz.end());
    const double mfma_per_simd = (double)iters * 32 * waves / 4.0;
    const double total_flops = (double)reps * 256 * waves * iters * 32 * flops_per_mfma;
    printf("%-18s waves/CU %2d  launch %8.1f us x %d: %7.1f TFLOP/s | median cycles per mfma and SIMD %.1f (pipe: %d) | shader clock median %.0f MHz (min %.0f max %.0f)\n",
           name, waves, ms * 1e3 / reps, reps, total_flops / (ms * 1e-3) / 1e12, cyc[128] / mfma_per_simd, pipe_cycles, mhz[128], mhz[0], mhz[255]);
    hipFree(d), hipFree(sink);
}

int main() {
    for (int waves : {4, 8, 16}) {
        run<0>("f32 32x32x2", waves, 2000 * 8 / waves, 1, 4096.0, 64);      // ~0.5 ms
        run<0>("f32 32x32x2", waves, 2000 * 8 / waves, 20, 4096.0, 64);     // sustained: 20 launches back to back
        run<0>("f32 32x32x2 long", waves, 40000 * 8 / waves, 3, 4096.0, 64);
    }
    for (int waves : {4, 8}) {
        run<1>("bf16 32x32x16", waves, 4000 * 8 / waves, 1, 32768.0, 32);
        run<1>("bf16 32x32x16", waves, 4000 * 8 / waves, 20, 32768.0, 32);
        run<1>("bf16 32x32x16 long", waves, 80000 * 8 / waves, 3, 32768.0, 32);
    }
    return 0;
}

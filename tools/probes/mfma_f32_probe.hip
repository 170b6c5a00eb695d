// What does v_mfma_f32_16x16x4_f32 sustain in the shape of conv_wring.hip's matrix phase?  8 waves per CU (2 per SIMD, 512-thread
// workgroups, one per CU), every wave a stream of NA accumulators visited round-robin; variants: NA = 2 (the kernel's groups of two),
// 4, 8; with V plain VALU adds and S SALU instructions between two matrix instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_probe.hip -o tools/_bin/mfma_f32_probe && tools/_bin/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NA, int V, int WAVES, int PK = 0, int SA = 0>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4) k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    if (SA >= 2) { for (int i = threadIdx.x; i < 8192; i += WAVES * 64) ((float *)lds)[i] = 1.f; __syncthreads(); }
    f32x4 acc[NA];
    for (int a = 0; a < NA; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x = (float)threadIdx.x, y = 1.0f + (float)(threadIdx.x & 3), z = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 zz = {0.f, 0.f}, yy = {y, x};
    unsigned sc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / NA; ++r) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(zz) : "v"(yy));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(z) : "v"(y));
                }
                if (SA == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                if (SA == 2) { if ((a & 1) == 0) { f32x4 t = *reinterpret_cast<const f32x4 *>(lds + ((threadIdx.x * 16 + r * 1024) & 32767)); asm volatile("" :: "v"(t)); } }   // 1 ds_read_b128 per 2 MFMAs
                if (SA == 3) { f32x4 t = *reinterpret_cast<const f32x4 *>(lds + ((threadIdx.x * 16 + r * 1024) & 32767)); asm volatile("" :: "v"(t)); }   // 1 per MFMA
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = z + zz[0] + zz[1] + (float)sc;
    for (int a = 0; a < NA; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

template <int NA, int V, int WAVES, int PK = 0, int SA = 0>
void run(const char *name) {
    const int iters = 2000, blocks = 256;
    float *out;
    hipMalloc(&out, blocks * WAVES * 64 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL((k<NA, V, WAVES, PK, SA>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NA, V, WAVES, PK, SA>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * WAVES * iters * 64 * 2048.0;
    printf("%-44s %7.1f TFLOP/s  (%.1f us)\n", name, flops / (ms * 1e-3) / 1e12, ms * 1e3);
    hipFree(out);
}

int main() {
    run<2, 0, 8>("8 waves, 2 accumulators, MFMA only");
    run<4, 0, 8>("8 waves, 4 accumulators, MFMA only");
    run<8, 0, 8>("8 waves, 8 accumulators, MFMA only");
    run<2, 1, 8>("8 waves, 2 accumulators, 1 VALU per MFMA");
    run<2, 2, 8>("8 waves, 2 accumulators, 2 VALU per MFMA");
    run<4, 2, 8>("8 waves, 4 accumulators, 2 VALU per MFMA");
    run<2, 0, 4>("4 waves, 2 accumulators, MFMA only");
    run<4, 0, 4>("4 waves, 4 accumulators, MFMA only");
    run<8, 0, 4>("4 waves, 8 accumulators, MFMA only");
    run<2, 0, 16>("16 waves, 2 accumulators, MFMA only");
    run<2, 1, 8, 1>("8 waves, 2 acc, 1 v_pk_add_f32 per MFMA");
    run<2, 2, 8, 1>("8 waves, 2 acc, 2 v_pk_add_f32 per MFMA");
    run<2, 0, 8, 0, 1>("8 waves, 2 acc, 1 SALU per MFMA");
    run<2, 4, 8>("8 waves, 2 acc, 4 VALU per MFMA");
    run<2, 0, 8, 0, 2>("8 waves, 2 acc, 1 ds_read_b128 per 2 MFMAs");
    run<2, 0, 8, 0, 3>("8 waves, 2 acc, 1 ds_read_b128 per MFMA");
    run<2, 1, 4>("4 waves, 2 acc, 1 VALU per MFMA");
    run<2, 2, 4>("4 waves, 2 acc, 2 VALU per MFMA");
    return 0;
}

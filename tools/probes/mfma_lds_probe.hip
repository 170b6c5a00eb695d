// How many LDS operand reads per MFMA can a CU sustain?  4 waves per workgroup, WGS workgroups per CU (LDS sized to force it),
// every wave loops: R ds_read_b128 (conflict-free, 80-byte row pitch as conv_bf16.hip) + 4 x v_mfma_f32_32x32x16_bf16 on 4 accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_probe.hip -o tools/_bin/mfma_lds_probe && tools/_bin/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int W>   // R reads and W ds_write_b128 per 4 MFMAs
__global__ void __launch_bounds__(256, 2) k(float *out, int iters, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < lds_bytes / 4; i += 256) ((unsigned *)lds)[i] = 0x3f803f80u;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int base = ((lane & 31) * 80 + (lane >> 5) * 16) + wv * 2560;
    bf16x8 f[4];
    for (int a = 0; a < 4; ++a) f[a] = *reinterpret_cast<const bf16x8 *>(lds + base + a * 32);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) f[r & 3] = *reinterpret_cast<const bf16x8 *>(lds + base + ((it * R + r) & 63) * 160);
#pragma unroll
        for (int w = 0; w < W; ++w) *reinterpret_cast<bf16x8 *>(lds + 20480 + threadIdx.x * 16 + ((it + w) & 7) * 4096) = f[w & 3];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], f[2], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], f[3], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], f[2], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], f[3], acc[3], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int R, int W>
void run(const char *name, int wgs_per_cu) {
    const int iters = 4000, blocks = 256 * wgs_per_cu * 4;
    const int lds = wgs_per_cu == 2 ? 72 * 1024 : (wgs_per_cu == 1 ? 100 * 1024 : 36 * 1024);
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipFuncSetAttribute((const void *)k<R, W>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL((k<R, W>), dim3(blocks), dim3(256), lds, 0, out, 10, lds);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<R, W>), dim3(blocks), dim3(256), lds, 0, out, iters, lds);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * 4 * iters * 4 * 32768.0;
    printf("%-34s %d WG/CU: %7.1f TFLOP/s\n", name, wgs_per_cu, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0, 0>("MFMA only", w);
        run<2, 0>("2 reads / 4 MFMA", w);
        run<4, 0>("4 reads / 4 MFMA (conv_bf16)", w);
        run<6, 0>("6 reads / 4 MFMA", w);
        run<4, 1>("4 reads + 1 write / 4 MFMA", w);
    }
    return 0;
}

// Issue cost of the vector instructions the uint8 720p warp is made of (gfx950): cycles of a SIMD per wave64 instruction, measured as
// throughput with 8 waves per SIMD, 8 independent chains per wave.  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(256) probe(float *out, int iters) {
    float a[8], b[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    unsigned u[8];
    const unsigned long long mask = 0x5555555555555555ull * (blockIdx.x & 1);
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i, b[i] = 1.0001f + i, p[i] = f2{a[i], b[i]}, u[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (OP == 4) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
                if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 6) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 8) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
                if (OP == 9) asm volatile("v_floor_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                if (OP == 10) asm volatile("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(u[i]) : "v"(a[i]));
                if (OP == 11) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 13) asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
                if (OP == 14) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a[i]));
                if (OP == 15) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                if (OP == 17) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask));
                if (OP == 18) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
                if (OP == 19) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(b[i]));
                if (OP == 20) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)" : "=v"(u[i]) : "v"(b[i]));
                if (OP == 21) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]));
                if (OP == 22) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 23) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 24) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
            }
        }
        if (OP == 11 || OP == 19 || OP == 20) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int OP>
void run(const char *name, float *d) {
    const int iters = 2000, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = 8.0 * iters * 32;   // 8 waves x iters x 32 instructions
    printf("%-28s %7.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / insts_per_simd,
           ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
    float *d;
    hipMalloc(&d, 4096);
    run<0>("v_fma_f32", d);
    run<12>("v_mul_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<2>("v_pk_mul_f32", d);
    run<3>("v_pk_add_f32", d);
    run<4>("v_cvt_f32_ubyte1", d);
    run<5>("v_cndmask_b32", d);
    run<6>("v_med3_i32", d);
    run<7>("v_mad_u32_u24", d);
    run<8>("v_cvt_i32_f32", d);
    run<9>("v_floor_f32", d);
    run<10>("v_cvt_u32_f32_sdwa", d);
    run<14>("v_cvt_pk_u8_f32", d);
    run<13>("v_cmp_gt_u32", d);
    run<15>("v_lshl_add_u32", d);
    run<16>("v_mov_b32", d);
    run<11>("ds_bpermute_b32 (chain)", d);
    run<19>("ds_bpermute_b32 (indep.)", d);
    run<20>("ds_swizzle_b32", d);
    run<21>("v_mov_b32_dpp row_shr", d);
    run<17>("v_cndmask_b32_e64 sgpr", d);
    run<18>("v_cndmask_b32 vcc indep.", d);
    run<5>("v_cndmask_b32 vcc chain", d);
    run<22>("v_and_b32", d);
    run<23>("v_bfe_u32", d);
    run<24>("v_perm_b32", d);
    run<0>("v_fma_f32 (again)", d);
    {   // dispatch floor: the warp's grid (48 x 5 x 8 workgroups of 256) doing nothing
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe<16>, dim3(48, 5, 8), dim3(256), 0, 0, d, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("empty kernel, grid 48x5x8 x 256 threads: %.1f us\n", ms * 1e3);
        }
    }
    return 0;
}

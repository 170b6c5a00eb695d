#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, int pitch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) ((unsigned short*)lds)[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const unsigned addr = ((g * 4 + (i >> 2)) * pitch + 4 * (i & 3)) * 2;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short h[4096], o[256];
    const int pitch = 16;
    for (int i = 0; i < 4096; ++i) h[i] = (i / pitch) * 100 + (i % pitch);
    unsigned short *d, *dout;
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, dout, pitch);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        int want = ((l >> 4) * 4 + e) * 100 + (l & 15);
        if (o[l * 4 + e] != want) { if (bad < 8) printf("lane %d e %d got %d want %d\n", l, e, o[l*4+e], want); ++bad; }
    }
    printf("tr16 semantics: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as expected", bad);
    for (int l = 0; l < 20; ++l) printf("lane %2d: %d %d %d %d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    return 0;
}

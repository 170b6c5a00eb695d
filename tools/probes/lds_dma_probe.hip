// Probe of the LDS-DMA path on gfx950 (buffer_load_dwordx4 ... lds), the staging primitive of conv_ring.hip:
//   1. destination = M0 base + lane * 16 for every base in the 160 KB LDS (also above 64 KB)?
//   2. lanes whose offset lies beyond the descriptor's num_records: are zeros written (zero padding for free)?
//   3. inline-asm form with M0 written in the same statement, counted s_waitcnt vmcnt + s_barrier before the ds_read.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_probe.hip -o tools/_bin/lds_dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(unsigned lds_base, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
    // m0 = wave-uniform LDS byte address; each lane's 16 bytes land at m0 + lane * 16
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

__global__ void __launch_bounds__(256) probe(const unsigned *src, unsigned src_bytes, unsigned *out, unsigned lds_base, unsigned oob_from) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // poison the destination window
    for (int i = tid; i < 4096 / 4; i += 256) reinterpret_cast<unsigned *>(lds + lds_base)[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, src_bytes, 0x00020000);
    // wave w loads the 1 KB piece w (reversed lane order in the SOURCE: lane l reads source slot 63 - l) ; lanes >= oob_from read out of range
    const unsigned slot = (unsigned)(wv * 64 + (63 - lane));
    const unsigned voff = lane >= (int)oob_from ? 0x7ffffff0u : slot * 16u;
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base + wv * 1024);
    dma16(base, voff, rsrc, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < 4096 / 4; i += 256) out[i] = reinterpret_cast<unsigned *>(lds + lds_base)[i];
}

int main() {
    const unsigned n = 4096 / 4;
    std::vector<unsigned> h(n), r(n);
    for (unsigned i = 0; i < n; ++i) h[i] = 0x1000000u + i;
    unsigned *d_src, *d_out;
    hipMalloc(&d_src, 4096), hipMalloc(&d_out, 4096);
    hipMemcpy(d_src, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    int bad_total = 0;
    for (unsigned base : {0u, 4096u, 61440u, 65536u, 100000u / 16 * 16, 159744u}) {
        for (unsigned oob : {64u, 40u}) {
            hipMemset(d_out, 0, 4096);
            hipLaunchKernelGGL(probe, dim3(1), dim3(256), 163840, 0, d_src, 4096u, d_out, base, oob);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(r.data(), d_out, 4096, hipMemcpyDeviceToHost);
            int bad = 0, zeros = 0, poison = 0;
            for (unsigned w = 0; w < 4; ++w)
                for (unsigned l = 0; l < 64; ++l)
                    for (unsigned k = 0; k < 4; ++k) {
                        const unsigned got = r[(w * 64 + l) * 4 + k];
                        const unsigned want = l >= oob ? 0u : h[(w * 64 + 63 - l) * 4 + k];
                        if (got != want) ++bad;
                        if (l >= oob && got == 0) ++zeros;
                        if (got == 0xdeadbeefu) ++poison;
                    }
            printf("lds base %6u oob_from %2u: %s  mismatches %d  (oob dwords zero %d, still poisoned %d)  [%s]\n", base, oob, bad ? "FAIL" : "ok", bad,
                   zeros, poison, hipGetErrorString(e));
            bad_total += bad;
        }
    }
    printf(bad_total ? "PROBE FAILED\n" : "PROBE OK: lane-linear destination over the whole LDS, out-of-range lanes write zeros\n");
    return bad_total ? 1 : 0;
}

// Pieces shared by the convolution translation units (conv_mfma.hip: fp32 matrix cores; conv_bf16.hip: bf16 matrix
// cores with fp32 accumulation): kernel parameter block, epilogue store, tile selector and the split-K reduce launcher.
#pragma once
#include "common.h"

namespace pws {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvKParams {
    const float *src_ptr[4];
    int src_c[4];
    int src_ld[4];
    int nsrc;
    int N, H, W;   // input
    int LH, LW;    // logical output extent walked by the tiles (conv: OH,OW ; convT k4s2: H,W)
    int OH, OW;    // output tensor extent
    int cin_pad;   // rows per tap in the packed weights
    int cout;
    const float *w;
    const float *bias;
    float *out;     // final output, or the partial buffer when ksplit > 1
    int out_ld;
    int act;
    int tiles_x, tiles_y;
    unsigned ntiles;
    int nclasses;   // 4 for the sub-pixel modes, else 1
    int cls_inner;  // sub-pixel modes: the 4 parity classes of a tile are 4 consecutive block slots of ONE XCD (grid.x = 4 ntiles)
                    // instead of grid.z planes launched far apart: they stage the same input tile, which then comes from that
                    // XCD's L2 three times out of four instead of from HBM (see conv_block_coords)
    int ksplit;     // >= 1
    int chunks_per_split;
    size_t split_stride;  // floats between consecutive partial buffers
    // data-gradient mode: the output channels are scattered over up to 4 NHWC destinations (the sources of the
    // forward layer's virtual concat), each either overwritten or accumulated into.  ndst == 0: plain `out`.
    // bf16 math (conv_bf16.hip): weights as [class*taps][npad_bf][kpad_bf] bf16 (pws_pack_weight_bf16)
    const void *w_bf;
    int kpad_bf, npad_bf;
    int io_bf16;    // bf16 math only: sources, `out` and dst_ptr[] hold bf16 elements (strides in elements), see conv_bf16.hip
    int epi16;      // io_bf16 only: every output row / destination segment is a whole number of 16-byte (8-channel) groups at
                    // 16-byte aligned addresses -> the LDS-transposed epilogue with 16-byte stores (PWS_BF_EPI16)
    int ndst;
    float *dst_ptr[4];
    int dst_c0[4], dst_c1[4], dst_ld[4], dst_acc[4];
    // bf16 storage only: the sum is multiplied by act'(dst_y) (the forward tensor this destination is the gradient of)
    const void *dst_y[4];
    int dst_y_ld[4], dst_act[4];
    // bf16 storage only: sign bits of the forward tensors (pws_conv_args.out_sign / pws_dst.act_sign): bit (c & 7) of byte
    // [pixel * ld + c / 8] = (tensor[pixel][c] > 0).  out_sign: written by the forward launch beside `out`; dst_sign[s]: read by the
    // data-gradient epilogue INSTEAD of dst_y[s] (1/16 of its bytes) by the kernels that know it (conv_ring.hip)
    void *out_sign;
    int out_sign_ld;
    const void *dst_sign[4];
    int dst_sign_ld[4];
};

// final store of one output element (pixel index `pix` in the output tensor, channel `co`)
__device__ __forceinline__ void epi_store(const ConvKParams &p, size_t pix, int co, float v) {
    if (p.ndst == 0) {
        p.out[pix * p.out_ld + co] = act_apply(v + (p.bias ? p.bias[co] : 0.f), p.act);
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < p.ndst && co >= p.dst_c0[s] && co < p.dst_c1[s]) {
                float *d = p.dst_ptr[s] + pix * p.dst_ld[s] + (co - p.dst_c0[s]);
                *d = p.dst_acc[s] ? *d + v : v;
            }
        }
    }
}

// bf16-storage counterparts (conv_bf16.hip with io_bf16): a lane stores the channel pair (co, co + 1) of one pixel as one dword
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void epi_store_pair16(const ConvKParams &p, size_t pix, int co, float v0, float v1) {
    if (p.ndst == 0) {
        const float b0 = p.bias ? p.bias[co] : 0.f, b1 = p.bias ? p.bias[co + 1] : 0.f;
        unsigned *d = reinterpret_cast<unsigned *>(reinterpret_cast<__bf16 *>(p.out) + pix * p.out_ld + co);
        *d = cvt_pk_bf16(act_apply(v0 + b0, p.act), act_apply(v1 + b1, p.act));
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < p.ndst && co >= p.dst_c0[s] && co < p.dst_c1[s]) {
                unsigned *d = reinterpret_cast<unsigned *>(reinterpret_cast<__bf16 *>(p.dst_ptr[s]) + pix * p.dst_ld[s] + (co - p.dst_c0[s]));
                if (p.dst_acc[s]) {
                    const unsigned o = *d;
                    v0 += bf16_lo(o), v1 += bf16_hi(o);
                }
                if (p.dst_act[s] != PWS_ACT_NONE) {
                    const unsigned y = *reinterpret_cast<const unsigned *>(static_cast<const __bf16 *>(p.dst_y[s]) + pix * p.dst_y_ld[s] +
                                                                           (co - p.dst_c0[s]));
                    const float sl = p.dst_act[s] == PWS_ACT_LRELU ? 0.2f : 0.f;
                    v0 *= bf16_lo(y) > 0.f ? 1.f : sl, v1 *= bf16_hi(y) > 0.f ? 1.f : sl;
                }
                *d = cvt_pk_bf16(v0, v1);
            }
        }
    }
}

// Block -> (tile, parity class, K split).  Plain mode: blockIdx.x = tile slot (XCD-remapped), blockIdx.z = class + 4 * split.
// cls_inner (ntiles % 8 == 0): blockIdx.x = ((tile_slot * 4 + class) * 8 + xcd), blockIdx.z = split.
__device__ __forceinline__ void conv_block_coords(const ConvKParams &p, bool subpix, unsigned &tile, int &cls, int &split) {
    if (subpix && p.cls_inner) {
        const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        cls = (int)(slot & 3u), split = (int)blockIdx.z;
        tile = xcd_remap(((slot >> 2) << 3) | xcd, p.ntiles);
    } else {
        tile = xcd_remap(blockIdx.x, p.ntiles);
        cls = subpix ? (int)(blockIdx.z & 3) : 0;
        split = subpix ? (int)(blockIdx.z >> 2) : (int)blockIdx.z;
    }
}
static inline dim3 conv_grid(ConvKParams &kp, int bn) {
    kp.cls_inner = kp.nclasses == 4 && kp.ntiles % 8 == 0 && g_experiment != 9 ? 1 : 0;
    const unsigned gy = (unsigned)((kp.cout + bn - 1) / bn);
    return kp.cls_inner ? dim3(kp.ntiles * 4, gy, (unsigned)kp.ksplit) : dim3(kp.ntiles, gy, (unsigned)(kp.nclasses * kp.ksplit));
}

struct ProfInfo {
    double flops, bytes;
};

struct TileChoice {  // one instantiation, described for the selector
    int th, tw, tn, ck, bn;
    int kid;
    int (*launch)(ConvKParams &, hipStream_t, const ProfInfo &);
};

int launch_splitk_reduce(const ConvKParams &kp, const float *partial, size_t total4, hipStream_t st);  // conv_mfma.hip

static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

constexpr long kFillBlocks = 512;      // 256 CUs x 2 resident workgroups
constexpr long kFillBlocksBf16 = 256;  // bf16 kernels: one workgroup per CU is enough before K is split (the 4x shorter kernels pay
                                       // relatively more for the split-K reduce launches: N=8 inference 4550 -> 4700 frames/s;
                                       // 128 / 384 / 512 measured 4580 / 4580 / 4550; fp32: 512 best, 256 and 1024 -2 %)

// Pick the tile and the K split for one launch.
//  1. drop tiles that are mostly padding for this extent (a 16x16 tile on an 8x8 map);
//  2. take the largest remaining tile whose grid has >= kFill workgroups, else the one with the most workgroups;
//  3. if the grid is still < kFill and a workspace was given, split K (>= 2 chunks per split, <= 64 splits).
[[maybe_unused]] static int select_and_launch(const TileChoice *cands, int ncand, ConvKParams &kp, int cin_total, float *final_out,
                             float *ws, size_t ws_floats, hipStream_t st, const ProfInfo &pi, long kFill = kFillBlocks) {
    const TileChoice *best = nullptr;
    long best_blocks = -1;
    double best_score = -1.0;
    for (int i = 0; i < ncand; ++i) {
        const TileChoice &c = cands[i];
        const long tiles = cdiv(kp.LW, c.tw) * cdiv(kp.LH, c.th) * cdiv(kp.N, c.tn);
        const double useful = (double)kp.N * kp.LH * kp.LW / ((double)tiles * c.th * c.tw * c.tn);
        if (useful < 0.45 && i + 1 < ncand) continue;
        // sub-8 spatial tiles exist for maps that are themselves tiny; on a larger map their halo re-reads dominate
        if (c.th < 8 && c.th < kp.LH && i > 0 && best) continue;
        const long blocks = tiles * cdiv(kp.cout, c.bn) * kp.nclasses;
        if (blocks >= kFill) {
            best = &c, best_blocks = blocks;
            break;
        }
        if (blocks * useful > best_score) best = &c, best_blocks = blocks, best_score = blocks * useful;
    }
    const TileChoice &c = *best;
    kp.tiles_x = (int)cdiv(kp.LW, c.tw), kp.tiles_y = (int)cdiv(kp.LH, c.th);
    kp.ntiles = (unsigned)(kp.tiles_x * kp.tiles_y * cdiv(kp.N, c.tn));
    const int total_chunks = cin_total / c.ck;
    int ksplit = 1;
    const size_t out_floats = (size_t)kp.N * kp.OH * kp.OW * kp.cout;
    if (best_blocks < kFill && ws && kp.cout % 4 == 0 && total_chunks >= 4) {
        long want = cdiv(kFill, best_blocks);
        if (want > 64) want = 64;
        if (want > total_chunks / 2) want = total_chunks / 2;
        while (want > 1 && (size_t)want * out_floats > ws_floats) --want;
        ksplit = (int)want;
    }
    kp.ksplit = ksplit < 1 ? 1 : ksplit;
    kp.chunks_per_split = (int)cdiv(total_chunks, kp.ksplit);
    kp.ksplit = (int)cdiv(total_chunks, kp.chunks_per_split);  // no empty splits
    kp.split_stride = out_floats;
    kp.out = kp.ksplit > 1 ? ws : final_out;
    ProfScope prof(c.kid, pi.flops, pi.bytes, st);  // covers the split-K reduce as well
    int rc = c.launch(kp, st, pi);
    if (rc != PWS_OK || kp.ksplit == 1) return rc;
    const size_t total4 = out_floats / 4;
    kp.out = final_out;
    return launch_splitk_reduce(kp, ws, total4, st);
}


}  // namespace pws

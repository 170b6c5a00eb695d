// Argument blocks of the whole-generator pack / unpack kernels (netg_pack.hip), filled by netg.cpp from its layer table.
#pragma once
#include "common.h"

namespace pws {

constexpr int kPackMaxLayers = 46;
constexpr unsigned kNoOff = 0xffffffffu;

struct PackLayer {
    int kind, cin, cin_pad, cout, k, planes, dg_taps;
    unsigned w_off, b_off, ww_off, dg_off;  // float offsets (kNoOff: none); 32 bits keep the argument block under 4 KB
    unsigned wr_off;                        // ring-layout Winograd weights (conv_wring.hip)
};

struct PackAllArgs {
    int nlayers;
    unsigned total_blocks, total_blocks_wino, total_blocks_dgrad;
    unsigned first_block[kPackMaxLayers], first_block_wino[kPackMaxLayers], first_block_dgrad[kPackMaxLayers];
    PackLayer layer[kPackMaxLayers];
    const float *params[2 * kPackMaxLayers];
};

struct UnpackAllArgs {
    int nlayers;
    unsigned total_blocks;
    unsigned first_block[kPackMaxLayers];
    PackLayer layer[kPackMaxLayers];
    float *grads[2 * kPackMaxLayers];
};

struct Bf16Layer {
    int planes, krows, ncols, kpad, npad;
    unsigned src_off, dst_off;  // float offsets into the source / destination buffer (kNoOff: layer has no bf16 copy)
};

struct Bf16AllArgs {
    int nlayers;
    unsigned total_blocks;
    unsigned first_block[kPackMaxLayers];
    Bf16Layer layer[kPackMaxLayers];
};

// The big layers (the five conv kinds of the encoder / decoder) are repacked tile by tile through LDS: a workgroup moves TA x 32
// (first x second torch dimension) x all k*k taps, reading the torch tensor in runs of 32 * k*k floats and writing the packed one in runs
// of TA or 32 floats -- both sides coalesced (netg_pack.hip).  The heads' small tensors keep the one-thread-per-element path.
#ifdef __HIPCC__
#define PWS_PACK_HD __host__ __device__
#else
#define PWS_PACK_HD
#endif
PWS_PACK_HD static inline bool pack_tiled(int kind) {
    return kind == PWS_CONV_K3S1 || kind == PWS_CONV_K3S2 || kind == PWS_CONV_K5S1 || kind == PWS_CONVT_K3S1 || kind == PWS_CONVT_K4S2;
}
PWS_PACK_HD static inline bool pack_iohw(int kind) { return kind == PWS_CONVT_K3S1 || kind == PWS_CONVT_K4S2; }   // torch layout [cin][cout][k][k]
PWS_PACK_HD static inline int pack_tile_a(int kk) { return kk <= 9 ? 32 : (kk <= 16 ? 16 : 8); }
// tiles of a layer: first torch dimension A in steps of TA, second B in steps of 32 (cin counted up to cin_pad where it is padded)
PWS_PACK_HD static inline unsigned pack_tiles(int kind, int cin_rows, int cout, int k) {
    const int A = pack_iohw(kind) ? cin_rows : cout, B = pack_iohw(kind) ? cout : cin_rows, ta = pack_tile_a(k * k);
    return (unsigned)(((A + ta - 1) / ta) * ((B + 31) / 32));
}

// One-pass pack of a bf16 training step (round 5): torch layouts -> the bf16 forward copy AND the bf16 data-gradient copy (+ bias), one
// read of the weights, no fp32 packed copies in between (those exist only to feed bf16_all_kernel in this mode).
struct Pack16Args {
    int nlayers;
    unsigned total_blocks;
    unsigned first_block[kPackMaxLayers];
    unsigned wb_off[kPackMaxLayers], dgb_off[kPackMaxLayers];   // float offsets of the bf16 copies in packed / packed_dgrad (kNoOff: layer not taken / no data-gradient copy)
    PackLayer layer[kPackMaxLayers];
    const float *params[2 * kPackMaxLayers];
};
static_assert(sizeof(Pack16Args) <= 4096, "kernel argument blocks are limited to 4 KB");

static_assert(sizeof(PackAllArgs) <= 4096 && sizeof(UnpackAllArgs) <= 4096 && sizeof(Bf16AllArgs) <= 4096,
              "kernel argument blocks are limited to 4 KB");

int launch_pack_all(const PackAllArgs &a, float *packed, hipStream_t st);
int launch_dgrad_all(const PackAllArgs &a, float *packed_dgrad, hipStream_t st);
int launch_bf16_all(const Bf16AllArgs &a, const float *src_base, float *dst_base, hipStream_t st);
int launch_unpack_all(const UnpackAllArgs &a, const float *dpacked, hipStream_t st);
int launch_pack16_all(const Pack16Args &a, float *packed, float *packed_dgrad, hipStream_t st);

}  // namespace pws

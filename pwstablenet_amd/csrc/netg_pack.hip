// Whole-generator weight (re)packing and gradient unpacking in a handful of launches.
//
// A training step changes all 92 tensors, so every step re-packs them into the kernels' layouts (forward fp32, Winograd,
// bf16, data-gradient fp32 + bf16) and unpacks the 92 gradients: done layer by layer that is ~330 launches of 4-9 us each
// (1.5-2 ms of a 19 ms step, tools/trace_train.sh).  Here each stage is ONE launch over all layers: the per-layer table
// (46 entries) and the 92 tensor pointers travel in the kernel arguments, a workgroup finds its layer by its block index.
// The element-wise index math is the per-layer kernels' (pack.hip, conv_wino.hip, conv_bf16.hip), which stay the public,
// individually tested entry points.
#include "netg_pack.h"

namespace pws {

__device__ __forceinline__ int find_layer(const unsigned *first_block, int n, unsigned b) {
    int lo = 0, hi = n - 1;  // last layer whose first block <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first_block[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ float torch_weight_at(const float *w, int kind, int cin, int cout, int k, size_t t, int ci, int co) {
    // t = class*taps + tap index in the forward packed layout (pack.hip)
    if (kind == PWS_CONVT_K4S2) {
        const int tap = t % 4, cls = t / 4;
        const int dy = tap >> 1, dx = tap & 1, py = cls >> 1, px = cls & 1;
        return w[(((size_t)ci * cout + co) * 4 + (3 - py - 2 * dy)) * 4 + (3 - px - 2 * dx)];
    }
    if (kind == PWS_CONVT_K3S1) {
        const int r = t / 3, s = t % 3;
        return w[(((size_t)ci * cout + co) * 3 + (2 - r)) * 3 + (2 - s)];
    }
    const int ky = t / k, kx = t % k;
    return w[(((size_t)co * cin + ci) * k + ky) * k + kx];
}

// ---- tiled repacking of the big layers (pack_tiled, netg_pack.h; one thread per packed element read a 4-byte value per lane out of
// lines k*k*cin floats apart: pack / data-gradient pack / unpack of the 48.5 M weights 234 / 282 / 242 us -> 200 / 195 / 169 us).  Tile image in LDS: L[a][b][q] = torch element (a0 + a, b0 + b, tap q),
// pitches PB = kk | 1 (odd) and PA = 32 PB + 1 (odd): walking a or b with the lanes is conflict-free.
constexpr int kPackTileFloats = 32 * (32 * 9 + 1);   // 9 taps: 32 x 32 x 9; 16 taps: 16 x 32 x 16; 25 taps: 8 x 32 x 25 -- all <= this
struct PackTile {
    int A, B, kk, TA, PB, PA, a0, b0;
    bool iohw;
};
// rows: the extent of the cin dimension the tiles cover (cin_pad for the forward layout, cin else)
__device__ __forceinline__ PackTile pack_tile_of(const PackLayer &L, unsigned b, int rows) {
    PackTile t;
    t.iohw = L.kind == PWS_CONVT_K3S1 || L.kind == PWS_CONVT_K4S2;
    t.kk = L.k * L.k;
    t.A = t.iohw ? rows : L.cout, t.B = t.iohw ? L.cout : rows;
    t.TA = pack_tile_a(t.kk), t.PB = t.kk | 1, t.PA = 32 * t.PB + 1;
    const unsigned tiles_b = (unsigned)((t.B + 31) / 32);
    t.a0 = (int)(b / tiles_b) * t.TA, t.b0 = (int)(b % tiles_b) * 32;
    return t;
}
// torch tensor -> tile (coalesced runs of 32 * kk floats); elements outside [A_real x B_real] are zeros.  KK = taps (compile time:
// the index arithmetic per element is divisions by constants)
template <int KK>
__device__ __forceinline__ void pack_tile_load_k(const PackTile &t, const float *__restrict__ w, int A_real, int B_real, float *lds) {
    constexpr int TA = KK <= 9 ? 32 : (KK <= 16 ? 16 : 8), PB = KK | 1, PA = 32 * PB + 1;
    for (int i = threadIdx.x; i < TA * 32 * KK; i += 256) {
        const int a = i / (32 * KK), rem = i - a * 32 * KK;
        const int b = rem / KK, q = rem - b * KK;
        const int ga = t.a0 + a, gb = t.b0 + b;
        lds[a * PA + b * PB + q] = (ga < A_real && gb < B_real) ? w[((size_t)ga * B_real + gb) * KK + q] : 0.f;
    }
}
__device__ __forceinline__ void pack_tile_load(const PackTile &t, const float *__restrict__ w, int A_real, int B_real, float *lds) {
    if (t.kk == 9) pack_tile_load_k<9>(t, w, A_real, B_real, lds);
    else if (t.kk == 16) pack_tile_load_k<16>(t, w, A_real, B_real, lds);
    else pack_tile_load_k<25>(t, w, A_real, B_real, lds);
}
// tile -> torch tensor
template <int KK>
__device__ __forceinline__ void pack_tile_store_k(const PackTile &t, float *__restrict__ w, int A_real, int B_real, const float *lds) {
    constexpr int TA = KK <= 9 ? 32 : (KK <= 16 ? 16 : 8), PB = KK | 1, PA = 32 * PB + 1;
    for (int i = threadIdx.x; i < TA * 32 * KK; i += 256) {
        const int a = i / (32 * KK), rem = i - a * 32 * KK;
        const int b = rem / KK, q = rem - b * KK;
        const int ga = t.a0 + a, gb = t.b0 + b;
        if (ga < A_real && gb < B_real) w[((size_t)ga * B_real + gb) * KK + q] = lds[a * PA + b * PB + q];
    }
}
__device__ __forceinline__ void pack_tile_store(const PackTile &t, float *__restrict__ w, int A_real, int B_real, const float *lds) {
    if (t.kk == 9) pack_tile_store_k<9>(t, w, A_real, B_real, lds);
    else if (t.kk == 16) pack_tile_store_k<16>(t, w, A_real, B_real, lds);
    else pack_tile_store_k<25>(t, w, A_real, B_real, lds);
}
// tap q of the torch tensor that plane t of the forward packed layout holds (torch_weight_at)
__device__ __forceinline__ int pack_fwd_tap(int kind, int k, int t) {
    if (kind == PWS_CONVT_K4S2) {
        const int tap = t & 3, cls = t >> 2;
        return (3 - (cls >> 1) - 2 * (tap >> 1)) * 4 + (3 - (cls & 1) - 2 * (tap & 1));
    }
    if (kind == PWS_CONVT_K3S1) return (2 - t / 3) * 3 + (2 - t % 3);
    return t;
}
// tap q of the torch tensor that plane t of the data-gradient layout holds, or -1 (a zero plane element)
__device__ __forceinline__ int pack_dgrad_tap(int kind, int t) {
    if (kind == PWS_CONV_K3S1) return (2 - t / 3) * 3 + (2 - t % 3);
    if (kind == PWS_CONVT_K3S1) return t;
    if (kind == PWS_CONV_K3S2) {
        const int tap = t & 3, cls = t >> 2;
        const int dy = tap >> 1, dx = tap & 1, py = cls >> 1, px = cls & 1;
        const int ry = py ? (dy ? 0 : 2) : (dy ? -1 : 1), rx = px ? (dx ? 0 : 2) : (dx ? -1 : 1);
        return (ry >= 0 && rx >= 0) ? ry * 3 + rx : -1;
    }
    return t;   // PWS_CONVT_K4S2
}

// ---- stage 1: torch layouts -> forward packed fp32 [class*tap][cin_pad][cout] (+ bias), all layers
__global__ void __launch_bounds__(256) pack_all_kernel(const PackAllArgs a, float *__restrict__ packed) {
    __shared__ float lds[kPackTileFloats];
    const int l = find_layer(a.first_block, a.nlayers, blockIdx.x);
    const PackLayer &L = a.layer[l];
    if (pack_tiled(L.kind)) {
        const unsigned b = blockIdx.x - a.first_block[l];
        const unsigned ntiles = pack_tiles(L.kind, L.cin_pad, L.cout, L.k);
        if (b >= ntiles) {   // the bias blocks behind the tiles
            const unsigned i = (b - ntiles) * 256 + threadIdx.x;
            if (i < (unsigned)L.cout) packed[L.b_off + i] = a.params[2 * l + 1][i];
            return;
        }
        const PackTile t = pack_tile_of(L, b, L.cin_pad);
        pack_tile_load(t, a.params[2 * l], t.iohw ? L.cin : L.cout, t.iohw ? L.cout : L.cin, lds);
        __syncthreads();
        // packed [plane][ci][co]: lanes walk co (torch dimension a for OIHW, b for IOHW)
        const int nf = t.iohw ? 32 : t.TA, ns = t.iohw ? t.TA : 32;   // fast (co) / slow (ci) extents of the tile
        const int lf = 31 - __builtin_clz((unsigned)nf), ls = 31 - __builtin_clz((unsigned)ns);
        const int n = L.planes * ns * nf;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int f = i & (nf - 1), s_ = (i >> lf) & (ns - 1), pl = i >> (lf + ls);   // nf, ns: powers of two
            const int q = pack_fwd_tap(L.kind, L.k, pl);
            const int la = t.iohw ? s_ : f, lb = t.iohw ? f : s_;
            const int ci = (t.iohw ? t.a0 : t.b0) + s_, co = (t.iohw ? t.b0 : t.a0) + f;
            if (ci < L.cin_pad && co < L.cout) packed[L.w_off + ((size_t)pl * L.cin_pad + ci) * L.cout + co] = lds[la * t.PA + lb * t.PB + q];
        }
        return;
    }
    const size_t idx = (size_t)(blockIdx.x - a.first_block[l]) * 256 + threadIdx.x;
    const size_t wtotal = (size_t)L.planes * L.cin_pad * L.cout;
    if (idx < wtotal) {
        const int co = idx % L.cout;
        size_t t = idx / L.cout;
        const int ci = t % L.cin_pad;
        t /= L.cin_pad;
        packed[L.w_off + idx] = ci < L.cin ? torch_weight_at(a.params[2 * l], L.kind, L.cin, L.cout, L.k, t, ci, co) : 0.f;
    } else if (idx < wtotal + L.cout) {
        packed[L.b_off + (idx - wtotal)] = a.params[2 * l + 1][idx - wtotal];
    }
}

// ---- stage 2: Winograd weights from the packed fp32 weights: F(2x2,3x3) of the K3S1 / CONVT_K3S1 layers (component-major for
// conv_wino.hip, ring layout for conv_wring.hip) and the ring-layout F(2x2,2x2) weights of the large-map CONVT_K4S2 layers
__global__ void __launch_bounds__(256) wino_all_kernel(const PackAllArgs a, float *__restrict__ packed) {
    const int l = find_layer(a.first_block_wino, a.nlayers, blockIdx.x);
    const PackLayer &L = a.layer[l];
    if (L.ww_off == kNoOff && L.wr_off == kNoOff) return;
    const size_t plane = (size_t)L.cin_pad * L.cout;
    const size_t i = (size_t)(blockIdx.x - a.first_block_wino[l]) * 256 + threadIdx.x;
    if (i >= plane) return;
    const float *pk = packed + L.w_off;
    if (L.wr_off != kNoOff) wring_pack_element(pk, packed + L.wr_off, plane, i, L.cin_pad, L.cout, L.kind == PWS_CONVT_K4S2 ? 1 : (L.kind == PWS_CONV_K5S1 ? 2 : 0));
    if (L.ww_off == kNoOff) return;
    float *uw = packed + L.ww_off;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = pk[(size_t)(r * 3 + s) * plane + i];
    float u[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        u[0][s] = g[0][s], u[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]), u[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
        u[3][s] = g[2][s];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uw[(size_t)(r * 4 + 0) * plane + i] = u[r][0];
        uw[(size_t)(r * 4 + 1) * plane + i] = 0.5f * (u[r][0] + u[r][1] + u[r][2]);
        uw[(size_t)(r * 4 + 2) * plane + i] = 0.5f * (u[r][0] - u[r][1] + u[r][2]);
        uw[(size_t)(r * 4 + 3) * plane + i] = u[r][2];
    }
}

// ---- stage 3: bf16 copies [plane][ncols pad 64][krows pad 32] of an fp32 [plane][krows][ncols] layout (forward or data
// gradient), all layers.  A workgroup transposes a 32 (k) x 64 (n) tile through LDS: coalesced on both sides.
__global__ void __launch_bounds__(256) bf16_all_kernel(const Bf16AllArgs a, const float *__restrict__ src_base,
                                                       unsigned *__restrict__ dst_base) {
    __shared__ float t[32][65];
    const int l = find_layer(a.first_block, a.nlayers, blockIdx.x);
    const Bf16Layer &L = a.layer[l];
    if (L.dst_off == kNoOff) return;
    unsigned b = blockIdx.x - a.first_block[l];
    const int kt = L.kpad / 32, nt = L.npad / 64;
    const int kb = b % kt;
    b /= kt;
    const int nb = b % nt, plane = b / nt;
    const float *src = src_base + (size_t)L.src_off + (size_t)plane * L.krows * L.ncols;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int k = kb * 32 + (i >> 6), n = nb * 64 + (i & 63);
        t[i >> 6][i & 63] = (k < L.krows && n < L.ncols) ? src[(size_t)k * L.ncols + n] : 0.f;
    }
    __syncthreads();
    // destination (in dwords = k pairs): [plane][npad][kpad/2]
    unsigned *dst = dst_base + ((size_t)L.dst_off * 2 + ((size_t)plane * L.npad + nb * 64) * L.kpad + kb * 32) / 2;
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
        const int n = i >> 4, k2 = i & 15;
        dst[(size_t)n * (L.kpad / 2) + k2] = cvt_pk_bf16(t[2 * k2][n], t[2 * k2 + 1][n]);
    }
}

// ---- data-gradient fp32 layouts [tap][cout_f][cin_f] of all layers that have one (pack.hip: pack_dgrad_kernel)
__global__ void __launch_bounds__(256) dgrad_all_kernel(const PackAllArgs a, float *__restrict__ packed_dgrad) {
    __shared__ float lds[kPackTileFloats];
    const int l = find_layer(a.first_block_dgrad, a.nlayers, blockIdx.x);
    const PackLayer &L = a.layer[l];
    if (L.dg_off == kNoOff) return;
    if (pack_tiled(L.kind)) {
        const PackTile t = pack_tile_of(L, blockIdx.x - a.first_block_dgrad[l], L.cin);
        pack_tile_load(t, a.params[2 * l], t.A, t.B, lds);
        __syncthreads();
        // [plane][co][ci]: lanes walk ci (torch dimension b for OIHW, a for IOHW)
        const int nf = t.iohw ? t.TA : 32, ns = t.iohw ? 32 : t.TA;
        const int lf = 31 - __builtin_clz((unsigned)nf), ls = 31 - __builtin_clz((unsigned)ns);
        const int n = L.dg_taps * ns * nf;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int f = i & (nf - 1), s_ = (i >> lf) & (ns - 1), pl = i >> (lf + ls);
            const int q = pack_dgrad_tap(L.kind, pl);
            const int la = t.iohw ? f : s_, lb = t.iohw ? s_ : f;
            const int ci = (t.iohw ? t.a0 : t.b0) + f, co = (t.iohw ? t.b0 : t.a0) + s_;
            if (ci < L.cin && co < L.cout)
                packed_dgrad[L.dg_off + ((size_t)pl * L.cout + co) * L.cin + ci] = q >= 0 ? lds[la * t.PA + lb * t.PB + q] : 0.f;
        }
        return;
    }
    const size_t idx = (size_t)(blockIdx.x - a.first_block_dgrad[l]) * 256 + threadIdx.x;
    const size_t total = (size_t)L.dg_taps * L.cin * L.cout;
    if (idx >= total) return;
    const float *w = a.params[2 * l];
    const int cin = L.cin, cout = L.cout, kind = L.kind;
    const int ci = idx % cin;
    size_t t = idx / cin;
    const int co = t % cout;
    t /= cout;
    float v = 0.f;
    if (kind == PWS_CONV_K3S1) {
        const int r = t / 3, s_ = t % 3;
        v = w[(((size_t)co * cin + ci) * 3 + (2 - r)) * 3 + (2 - s_)];
    } else if (kind == PWS_CONVT_K3S1) {
        const int r = t / 3, s_ = t % 3;
        v = w[(((size_t)ci * cout + co) * 3 + r) * 3 + s_];
    } else if (kind == PWS_CONV_K3S2) {
        const int tap = t % 4, cls = t / 4;
        const int dy = tap >> 1, dx = tap & 1, py = cls >> 1, px = cls & 1;
        const int ry = py ? (dy ? 0 : 2) : (dy ? -1 : 1), rx = px ? (dx ? 0 : 2) : (dx ? -1 : 1);
        if (ry >= 0 && rx >= 0) v = w[(((size_t)co * cin + ci) * 3 + ry) * 3 + rx];
    } else {  // PWS_CONVT_K4S2
        const int ky = t / 4, kx = t % 4;
        v = w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
    }
    packed_dgrad[L.dg_off + idx] = v;
}

// ---- gradients: packed layout -> the 92 torch-layout tensors (pack.hip: unpack_weight_kernel) + biases
__global__ void __launch_bounds__(256) unpack_all_kernel(const UnpackAllArgs a, const float *__restrict__ dpacked) {
    __shared__ float lds[kPackTileFloats];
    const int l = find_layer(a.first_block, a.nlayers, blockIdx.x);
    const PackLayer &L = a.layer[l];
    if (pack_tiled(L.kind)) {
        const unsigned b = blockIdx.x - a.first_block[l];
        const unsigned ntiles = pack_tiles(L.kind, L.cin, L.cout, L.k);
        if (b >= ntiles) {   // the bias blocks behind the tiles
            const unsigned i = (b - ntiles) * 256 + threadIdx.x;
            if (i < (unsigned)L.cout) a.grads[2 * l + 1][i] = dpacked[L.b_off + i];
            return;
        }
        const PackTile t = pack_tile_of(L, b, L.cin);
        // packed [plane][ci][co] -> tile: lanes walk co; every tap q of the torch tensor lies in exactly one plane
        const int nf = t.iohw ? 32 : t.TA, ns = t.iohw ? t.TA : 32;
        const int lf = 31 - __builtin_clz((unsigned)nf), ls = 31 - __builtin_clz((unsigned)ns);
        const int n = L.planes * ns * nf;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int f = i & (nf - 1), s_ = (i >> lf) & (ns - 1), pl = i >> (lf + ls);
            const int q = pack_fwd_tap(L.kind, L.k, pl);
            const int la = t.iohw ? s_ : f, lb = t.iohw ? f : s_;
            const int ci = (t.iohw ? t.a0 : t.b0) + s_, co = (t.iohw ? t.b0 : t.a0) + f;
            lds[la * t.PA + lb * t.PB + q] = (ci < L.cin && co < L.cout) ? dpacked[L.w_off + ((size_t)pl * L.cin_pad + ci) * L.cout + co] : 0.f;
        }
        __syncthreads();
        pack_tile_store(t, a.grads[2 * l], t.A, t.B, lds);
        return;
    }
    const size_t idx = (size_t)(blockIdx.x - a.first_block[l]) * 256 + threadIdx.x;
    const int k = L.k, cin = L.cin, cout = L.cout, kind = L.kind;
    const size_t wtotal = (size_t)k * k * cin * cout;
    if (idx < wtotal) {
        const int kx = idx % k, ky = (idx / k) % k;
        const size_t t = idx / ((size_t)k * k);
        size_t tapcls;
        int ci, co;
        if (kind == PWS_CONVT_K4S2 || kind == PWS_CONVT_K3S1) {  // IOHW
            co = t % cout, ci = t / cout;
            if (kind == PWS_CONVT_K4S2) {
                const int py = (3 - ky) & 1, dy = (3 - ky) >> 1, px = (3 - kx) & 1, dx = (3 - kx) >> 1;
                tapcls = (size_t)(py * 2 + px) * 4 + dy * 2 + dx;
            } else {
                tapcls = (size_t)(2 - ky) * 3 + (2 - kx);
            }
        } else {  // OIHW
            ci = t % cin, co = t / cin;
            tapcls = (size_t)ky * k + kx;
        }
        a.grads[2 * l][idx] = dpacked[L.w_off + (tapcls * L.cin_pad + ci) * cout + co];
    } else if (idx < wtotal + cout) {
        a.grads[2 * l + 1][idx - wtotal] = dpacked[L.b_off + (idx - wtotal)];
    }
}

// ---- one pass for a bf16 training step: torch tensor tile -> LDS -> bf16 forward layout [plane][cout][cin_pad] AND bf16 data-gradient
// layout [plane][cin][cout] (both k-contiguous, as bf16_all_kernel writes them), + the bias.  Only for layers whose padded extents are
// the real ones (netg.cpp: cout % 64 == 0, cin_pad % 32 == 0; with a data-gradient copy cin % 64 == 0 too), so the tiles cover every
// element of both destinations and nothing is left to zero-fill.  Per step and generator: 194 MB read + 2 x 97 MB written, against
// 194 + 194 (pack) + 194 + 97 (bf16) + 194 + 194 (data-gradient pack) + 194 + 97 (bf16) with the four launches it replaces.
__global__ void __launch_bounds__(256) pack16_all_kernel(const Pack16Args a, float *__restrict__ packed, float *__restrict__ packed_dgrad) {
    __shared__ float lds[kPackTileFloats];
    const int l = find_layer(a.first_block, a.nlayers, blockIdx.x);
    const PackLayer &L = a.layer[l];
    const unsigned b = blockIdx.x - a.first_block[l];
    const unsigned ntiles = pack_tiles(L.kind, L.cin_pad, L.cout, L.k);
    if (b >= ntiles) {   // the bias blocks behind the tiles
        const unsigned i = (b - ntiles) * 256 + threadIdx.x;
        if (i < (unsigned)L.cout) packed[L.b_off + i] = a.params[2 * l + 1][i];
        return;
    }
    const PackTile t = pack_tile_of(L, b, L.cin_pad);
    pack_tile_load(t, a.params[2 * l], t.iohw ? L.cin : L.cout, t.iohw ? L.cout : L.cin, lds);
    __syncthreads();
    const int nci = t.iohw ? t.TA : 32, nco = t.iohw ? 32 : t.TA;   // extents of the tile in ci / co
    const int ci0 = t.iohw ? t.a0 : t.b0, co0 = t.iohw ? t.b0 : t.a0;
    const int sci = t.iohw ? t.PA : t.PB, sco = t.iohw ? t.PB : t.PA;   // LDS strides of ci / co
    {   // forward: [plane][co][ci], ci pairs per dword
        unsigned *dst = reinterpret_cast<unsigned *>(packed) + a.wb_off[l];
        const int kpad = L.cin_pad, half = nci / 2;
        const int n = L.planes * nco * half;
        const int lh = 31 - __builtin_clz((unsigned)half), lc = 31 - __builtin_clz((unsigned)nco);   // (tile extents are powers of two: shifts, not divisions)
        for (int i = threadIdx.x; i < n; i += 256) {
            const int k2 = i & (half - 1), r = i >> lh;
            const int co_l = r & (nco - 1), pl = r >> lc;
            const int q = pack_fwd_tap(L.kind, L.k, pl);
            const float *src = lds + (2 * k2) * sci + co_l * sco + q;
            const int ci = ci0 + 2 * k2, co = co0 + co_l;
            if (ci < kpad && co < L.cout) dst[(((size_t)pl * L.cout + co) * kpad + ci) >> 1] = cvt_pk_bf16(src[0], src[sci]);
        }
    }
    if (a.dgb_off[l] != kNoOff) {   // data gradient: [plane][ci][co], co pairs per dword; a plane element without a tap is zero
        unsigned *dst = reinterpret_cast<unsigned *>(packed_dgrad) + a.dgb_off[l];
        const int kpad = L.cout, half = nco / 2;
        const int n = L.dg_taps * nci * half;
        const int lh = 31 - __builtin_clz((unsigned)half), lc = 31 - __builtin_clz((unsigned)nci);
        for (int i = threadIdx.x; i < n; i += 256) {
            const int k2 = i & (half - 1), r = i >> lh;
            const int ci_l = r & (nci - 1), pl = r >> lc;
            const int q = pack_dgrad_tap(L.kind, pl);
            const float *src = lds + ci_l * sci + (2 * k2) * sco + (q >= 0 ? q : 0);
            const int ci = ci0 + ci_l, co = co0 + 2 * k2;
            if (ci < L.cin && co < kpad) dst[(((size_t)pl * L.cin + ci) * kpad + co) >> 1] = q >= 0 ? cvt_pk_bf16(src[0], src[sco]) : 0u;
        }
    }
}

int launch_pack16_all(const Pack16Args &a, float *packed, float *packed_dgrad, hipStream_t st) {
    if (!a.total_blocks) return PWS_OK;
    hipLaunchKernelGGL(pack16_all_kernel, dim3(a.total_blocks), dim3(256), 0, st, a, packed, packed_dgrad);
    return check_launch("pack16_all_kernel");
}

int launch_pack_all(const PackAllArgs &a, float *packed, hipStream_t st) {
    hipLaunchKernelGGL(pack_all_kernel, dim3(a.total_blocks), dim3(256), 0, st, a, packed);
    if (a.total_blocks_wino) hipLaunchKernelGGL(wino_all_kernel, dim3(a.total_blocks_wino), dim3(256), 0, st, a, packed);
    return check_launch("pack_all_kernel");
}

int launch_dgrad_all(const PackAllArgs &a, float *packed_dgrad, hipStream_t st) {
    hipLaunchKernelGGL(dgrad_all_kernel, dim3(a.total_blocks_dgrad), dim3(256), 0, st, a, packed_dgrad);
    return check_launch("dgrad_all_kernel");
}

int launch_bf16_all(const Bf16AllArgs &a, const float *src_base, float *dst_base, hipStream_t st) {
    if (!a.total_blocks) return PWS_OK;
    hipLaunchKernelGGL(bf16_all_kernel, dim3(a.total_blocks), dim3(256), 0, st, a, src_base, reinterpret_cast<unsigned *>(dst_base));
    return check_launch("bf16_all_kernel");
}

int launch_unpack_all(const UnpackAllArgs &a, const float *dpacked, hipStream_t st) {
    hipLaunchKernelGGL(unpack_all_kernel, dim3(a.total_blocks), dim3(256), 0, st, a, dpacked);
    return check_launch("unpack_all_kernel");
}

}  // namespace pws

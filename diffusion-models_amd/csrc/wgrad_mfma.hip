// Weight gradients of the U-Net's convolutions for the training step (DD/denoising_diffusion.py:805-900: autograd's
// conv2d backward w.r.t. the weight), on v_mfma_f32_32x32x2_f32:
//     dW[cout][cin][tap] = sum over pixels  dY[pixel][cout] * X[pixel + tap][cin]
// The reduction runs over PIXELS (B*H*W: 4k .. 260k), the output is small (Cout x Cin x taps), so the roles are the
// transpose of the forward kernels: couts are the MFMA rows, cins the columns, a pair of neighbouring pixels the K index.
// Both operands are NHWC, i.e. K-major with the M / N index contiguous -- exactly what the 32x32x2 operand layout wants
// (lane = (row, k): the 32 lanes of one k read 32 consecutive floats).
//
// One workgroup = 4 waves = 64 couts x 64 cins x one kernel ROW (the three taps ky = blockIdx.z of a 3x3 kernel; all taps
// of the 1x1 / 2x2 forms); wave (wo, wc) owns 32 couts x 32 cins and keeps one 32x32 accumulator per tap: the dY value of a
// pixel pair is loaded once and meets the shifted X values, which come out of the same LDS window.  Per block of <= 64
// pixels (R rows x TW columns of one image, or several whole small images) the workgroup stages dY (64 px x 64 couts) and
// the X window (R x (TW+2) px x 64 cins) in LDS; the next block's items are prefetched into registers during the MFMAs.
// The pixel blocks are split over blockIdx.y; every split writes its partial [tap][cout][cin] tile and
// wgrad_reduce_kernel sums the splits into the OIHW gradient (deterministic, no atomics).  A workgroup per kernel row
// instead of per 3x3 kernel: the chip is filled with a third of the splits, i.e. a third of the partial-sum traffic (which
// was 38 MB per layer, written and read back), for three times the (cheap) staging.
//
// Modes: 0 3x3 / pad 1 (optionally on a nearest-x2 upsampled source: Upsample :48-52), 1 1x1, 2 the 2x2 / stride 2 form of
// Downsample (:54-58; weight index c*4 + p1*2 + p2), 3 the 3x3 form in the Winograd domain (wgrad_wino_kernel below: what
// the training step uses on even image sizes).  GROUPED instantiations run a table of layers in one launch.  Inputs may be the channel concatenation of two tensors (the up
// path's torch.cat).  Channel counts must be multiples of 4; everything else (7x7 first conv over the NCHW image,
// final_conv with 3 outputs) goes through wgrad_naive_kernel.
#include "conv_device.h"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace dm {

static constexpr int WG_SY = 64;  // floats per staged dY pixel
static constexpr int WG_SX = 64;  // floats per staged X pixel

template <int MODE>
struct WgradMode;
template <>
struct WgradMode<0> { static constexpr int T = 9, TA = 3, S = 1; };
template <>
struct WgradMode<1> { static constexpr int T = 1, TA = 1, S = 1; };
template <>
struct WgradMode<2> { static constexpr int T = 4, TA = 4, S = 2; };

// GROUPED: one launch for a table of layers (the training step runs the weight gradients of ALL its convolutions of one
// mode together at the end of the backward pass: the chip is filled by independent layers instead of by splitting every
// layer's pixel range 256 ways); workgroup -> (layer, tile, split, kernel row) through the first_wg prefix.
template <int MODE, bool GROUPED>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const WgradParams p1, const WgradParams* __restrict__ table,
                                                            int n_jobs) {
    constexpr int T = WgradMode<MODE>::T;    // taps of the partial layout
    constexpr int TA = WgradMode<MODE>::TA;  // taps (accumulators) of this workgroup
    constexpr int S = WgradMode<MODE>::S;
    int bx = blockIdx.x, split = blockIdx.y, kyb = MODE == 0 ? blockIdx.z : 0;  // tile, pixel split, kernel row
    WgradParams p = p1;
    if (GROUPED) {
        int lo = 0, hi = n_jobs - 1;
        while (lo < hi) {  // last layer whose first_wg <= blockIdx.x
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid].first_wg <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        p = table[lo];
        const int lb = blockIdx.x - p.first_wg, tiles = p.n_ct * p.n_kt;
        bx = lb % tiles;
        const int rest = lb / tiles;
        split = rest % p.n_splits;
        kyb = rest / p.n_splits;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sY = smem;                // [64 px][64 couts]
    float* sX = smem + 64 * WG_SY;   // [WH * WW px][64 cins]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, k = lane >> 5;
    const int ct = bx % p.n_ct, kt = bx / p.n_ct;
    const int TW = p.TW, R = p.R;
    const int WW = MODE == 0 ? TW + 2 : S * TW;
    const int WH = S * R;
    const int Hs = MODE == 2 ? 2 * p.Ho : (p.up ? p.Ho / 2 : p.Ho);  // source tensor size
    const int Ws = MODE == 2 ? 2 * p.Wo : (p.up ? p.Wo / 2 : p.Wo);

    f32x16 acc[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // ---- staging items of this thread: (window pixel, channel quad) pairs do not depend on the block, only their image
    //      coordinates do: the divisions are done once, a block costs additions and bounds checks.  The items of the NEXT
    //      block are loaded into registers before the MFMA loop of the current one (one workgroup hides its own staging).
    constexpr int YI = 4;                                      // dY: 64 px x 16 quads / 256 threads
    constexpr int XI = MODE == 0 ? 6 : (MODE == 1 ? 4 : 16);   // X window items (upper bound, checked by the launcher)
    const int NB = p.NB;
    const int q4 = 4 * (tid & 15);
    int ypos[YI], xpos[XI];  // (image << 24) | (row << 12) | column inside the block / window, -1: no item
#pragma unroll
    for (int j = 0; j < YI; ++j) {
        const int px = (tid >> 4) + 16 * j;
        const int nb = px / (R * TW), rem = px - nb * (R * TW);
        const int r = rem / TW, c = rem - r * TW;
        ypos[j] = nb < NB ? (nb << 24) | (r << 12) | c : -1;
    }
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int wp = (tid >> 4) + 16 * j;
        const int nb = wp / (WH * WW), rem = wp - nb * (WH * WW);
        const int wy = rem / WW, wx = rem - wy * WW;
        xpos[j] = nb < NB ? (nb << 24) | (wy << 12) | wx : -1;
    }
    const int co_s = ct * 64 + q4, ci_s = kt * 64 + q4;
    const bool co_ok = co_s < p.Cout, ci_ok = ci_s < p.Cin;
    const bool src1 = ci_s >= p.C0;
    const float* xsrc = src1 ? p.in1 + (ci_s - p.C0) : p.in0 + ci_s;
    const int xC = src1 ? p.C1 : p.C0;
    const int per_img = p.tiles_x * p.tiles_y;

    f32x4 yreg[YI], xreg[XI];
    auto load_block = [&](int blk) {
        const int bg = blk / per_img, rem = blk - bg * per_img;
        const int b0 = bg * NB;
        const int y0 = (rem / p.tiles_x) * R, x0 = (rem % p.tiles_x) * TW;
#pragma unroll
        for (int j = 0; j < YI; ++j) {
            f32x4 v = make_f32x4(0.f, 0.f, 0.f, 0.f);
            if (ypos[j] >= 0) {
                const int b = b0 + (ypos[j] >> 24), y = y0 + ((ypos[j] >> 12) & 0xFFF), x = x0 + (ypos[j] & 0xFFF);
                if (b < p.B && y < p.Ho && x < p.Wo && co_ok)
                    v = *reinterpret_cast<const f32x4*>(p.dy + ((size_t)(b * p.Ho + y) * p.Wo + x) * p.Cout + co_s);
            }
            yreg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            f32x4 v = make_f32x4(0.f, 0.f, 0.f, 0.f);
            if (xpos[j] >= 0) {
                const int b = b0 + (xpos[j] >> 24), wy = (xpos[j] >> 12) & 0xFFF, wx = xpos[j] & 0xFFF;
                int sy, sx;
                bool ok;
                if (MODE == 0) {
                    const int uy = y0 - 1 + kyb + wy, ux = x0 - 1 + wx;
                    ok = uy >= 0 && uy < p.Ho && ux >= 0 && ux < p.Wo;
                    sy = p.up ? uy >> 1 : uy;
                    sx = p.up ? ux >> 1 : ux;
                } else {
                    sy = S * y0 + wy;
                    sx = S * x0 + wx;
                    ok = sy < Hs && sx < Ws;
                }
                if (ok && ci_ok && b < p.B) v = *reinterpret_cast<const f32x4*>(xsrc + ((size_t)(b * Hs + sy) * Ws + sx) * xC);
            }
            xreg[j] = v;
        }
    };

    const int blk0 = split * p.blocks_per_split;
    const int blk1 = min(blk0 + p.blocks_per_split, p.n_blocks);
    if (blk0 < blk1) load_block(blk0);
    for (int blk = blk0; blk < blk1; ++blk) {
        __syncthreads();  // the previous block's MFMA loop is done with the tiles
#pragma unroll
        for (int j = 0; j < YI; ++j) *reinterpret_cast<f32x4*>(sY + ((tid >> 4) + 16 * j) * WG_SY + q4) = yreg[j];
#pragma unroll
        for (int j = 0; j < XI; ++j)
            if (xpos[j] >= 0) *reinterpret_cast<f32x4*>(sX + ((tid >> 4) + 16 * j) * WG_SX + q4) = xreg[j];
        __syncthreads();
        if (blk + 1 < blk1) load_block(blk + 1);  // in flight during the MFMAs below
        // ---- MFMA: K = pixel pairs (2 cp + k) of every row of every image of the block
        const float* ya = sY + wo * 32 + l31;
        const float* xb = sX + wc * 32 + l31;
        for (int nb = 0; nb < NB; ++nb)
        for (int r = 0; r < R; ++r) {
            for (int cp = 0; cp < TW / 2; ++cp) {
                const int c = 2 * cp + k;
                const float a = ya[((nb * R + r) * TW + c) * WG_SY];
                const float* xw = xb + (size_t)nb * WH * WW * WG_SX;
                if (MODE == 0) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float bv = xw[(r * WW + c + kx) * WG_SX];
                        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[kx], 0, 0, 0);
                    }
                } else if (MODE == 1) {
                    const float bv = xw[(r * WW + c) * WG_SX];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[0], 0, 0, 0);
                } else {
#pragma unroll
                    for (int p1 = 0; p1 < 2; ++p1)
#pragma unroll
                        for (int p2 = 0; p2 < 2; ++p2) {
                            const float bv = xw[((2 * r + p1) * WW + 2 * c + p2) * WG_SX];
                            acc[p1 * 2 + p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[p1 * 2 + p2], 0, 0, 0);
                        }
                }
            }
        }
    }
    // ---- partial tile: D[i][j], j = lane % 32 (cin), i = 8 (v / 4) + 4 (lane / 32) + v % 4 (cout)
    const int ci = kt * 64 + wc * 32 + l31;
    if (ci < p.Cin) {
#pragma unroll
        for (int t = 0; t < TA; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int co = ct * 64 + wo * 32 + 8 * (v >> 2) + 4 * k + (v & 3);
                if (co < p.Cout) p.partial[(((size_t)split * T + kyb * TA + t) * p.Cout + co) * p.Cin + ci] = acc[t][v];
            }
    }
}

// ---- mode 3: 3x3 in the Winograd domain, F(3x3, 2x2): per 2x2 tile of dY and its 4x4 input patch
//     dW = G^T [ sum_tiles (A dY A^T) (.) (B^T X B) ] G      A = [1 0; 1 1; 1 -1; 0 -1]   (4 x 2)
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]          G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
// (the transposition of the F(2x2, 3x3) algorithm of winograd_mfma.hip: same B^T, A and G change roles): 16 products per
// tile and (cout, cin) instead of 36.  The K index of the MFMA is a pair of TILES; a workgroup owns two of the four
// transform rows (blockIdx.z), i.e. 8 accumulators of 16 registers; the partial sums stay in the transformed domain
// (T = 16) and the reduce kernel applies G^T . G.  Even image sizes only.
// The TRANSFORMED tiles are staged.  Thread (tile, channel quad) of the workgroup
// loads its own 3x4-pixel piece of the 4x4 input patch and its 2x2 dY tile straight from global memory (prefetched one pixel
// block ahead, as above), applies the two +-1 transforms once, and stores the eight values of its transform-row pair as
// [frequency][tile][channel] in LDS (64 KB for both operands: two workgroups per CU).  The MFMA loop is then one LDS read per
// operand and nothing else -- the version that transformed inside the loop (4 + 12 raw reads and ~40 VALU operations per
// 8 MFMAs, every wave redoing its neighbours' transforms) ran at 0.37 of the peak by executed products, this one at 0.49
// (also tried there: the row pair as a compile-time constant through two launches -- every input staged twice, no gain).
template <bool GROUPED>
__global__ __launch_bounds__(256, 2) void wgrad_wino_kernel(const WgradParams p1, const WgradParams* __restrict__ table,
                                                            int n_jobs) {
    constexpr int T = 16, TA = 8;
    int bx = blockIdx.x, split = blockIdx.y, zi = blockIdx.z;  // tile, pixel split, transform-row pair
    WgradParams p = p1;
    if (GROUPED) {
        int lo = 0, hi = n_jobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid].first_wg <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        p = table[lo];
        const int lb = blockIdx.x - p.first_wg, tiles = p.n_ct * p.n_kt;
        bx = lb % tiles;
        const int rest = lb / tiles;
        split = rest % p.n_splits;
        zi = rest / p.n_splits;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sYt = smem;                  // [8 freq][16 tiles][64 couts]
    float* sXt = smem + TA * 16 * 64;   // [8 freq][16 tiles][64 cins]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, k = lane >> 5;
    const int ct = bx % p.n_ct, kt = bx / p.n_ct;
    const int TW = p.TW, R = p.R, NB = p.NB;
    const int Hs = p.up ? p.Ho / 2 : p.Ho, Ws = p.up ? p.Wo / 2 : p.Wo;  // source tensor size

    f32x16 acc[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // this thread's staging item: tile (image nb, tile row ty, tile column tx of the pixel block) x channel quad
    const int tile = tid >> 4, q4 = 4 * (tid & 15);
    const int tpr = TW >> 1, tpi = (R >> 1) * tpr, ntiles = NB * tpi;
    const bool tile_ok = tile < ntiles;
    const int nb = tile / tpi, trem = tile - nb * tpi, ty = trem / tpr, tx = trem - ty * tpr;
    const int co_s = ct * 64 + q4, ci_s = kt * 64 + q4;
    const bool co_ok = co_s < p.Cout, ci_ok = ci_s < p.Cin;
    const bool src1 = ci_s >= p.C0;
    const float* xsrc = src1 ? p.in1 + (ci_s - p.C0) : p.in0 + ci_s;
    const int xC = src1 ? p.C1 : p.C0;
    const int per_img = p.tiles_x * p.tiles_y;
    const f32x4 zero4 = make_f32x4(0.f, 0.f, 0.f, 0.f);

    f32x4 yr[4], xr[12];  // dY tile (row-major 2x2), input patch rows zi .. zi + 2 (row-major 3x4)
    auto load_block = [&](int blk) {
        const int bg = blk / per_img, rem = blk - bg * per_img;
        const int b = bg * NB + nb;
        const int y0 = (rem / p.tiles_x) * R + 2 * ty, x0 = (rem % p.tiles_x) * TW + 2 * tx;
        const bool img_ok = tile_ok && b < p.B;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + (j >> 1), x = x0 + (j & 1);
            yr[j] = (img_ok && co_ok && y < p.Ho && x < p.Wo)
                        ? *reinterpret_cast<const f32x4*>(p.dy + ((size_t)(b * p.Ho + y) * p.Wo + x) * p.Cout + co_s)
                        : zero4;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int uy = y0 - 1 + zi + a, ux = x0 - 1 + c;
                const bool ok = img_ok && ci_ok && uy >= 0 && uy < p.Ho && ux >= 0 && ux < p.Wo;
                const int sy = p.up ? uy >> 1 : uy, sx = p.up ? ux >> 1 : ux;
                xr[a * 4 + c] = ok ? *reinterpret_cast<const f32x4*>(xsrc + ((size_t)(b * Hs + sy) * Ws + sx) * xC) : zero4;
            }
    };

    const int blk0 = split * p.blocks_per_split;
    const int blk1 = min(blk0 + p.blocks_per_split, p.n_blocks);
    if (blk0 < blk1) load_block(blk0);
    for (int blk = blk0; blk < blk1; ++blk) {
        __syncthreads();  // the previous block's MFMA loop is done with the tiles
        {
            // rows 2 zi, 2 zi + 1 of A dY:  [d0; d0 + d1]  or  [d0 - d1; -d1];  then the same on the columns
            const f32x4 p0 = zi ? yr[0] - yr[2] : yr[0], p1 = zi ? yr[1] - yr[3] : yr[1];
            const f32x4 q0 = zi ? zero4 - yr[2] : yr[0] + yr[2], q1 = zi ? zero4 - yr[3] : yr[1] + yr[3];
            const f32x4 av[8] = {p0, p0 + p1, p0 - p1, zero4 - p1, q0, q0 + q1, q0 - q1, zero4 - q1};
            float* dst = sYt + tile * 64 + q4;
#pragma unroll
            for (int f = 0; f < TA; ++f) *reinterpret_cast<f32x4*>(dst + f * 16 * 64) = av[f];
            // B^T rows: 0: r0 - r2, 1: r1 + r2 | 2: r2' - r1' = x1 - x0 here, 3: r1' - r3' = x0 - x2 here; then the columns
            f32x4 y0v[4], y1v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y0v[c] = zi ? xr[4 + c] - xr[c] : xr[c] - xr[8 + c];
                y1v[c] = zi ? xr[c] - xr[8 + c] : xr[4 + c] + xr[8 + c];
            }
            const f32x4 bv[8] = {y0v[0] - y0v[2], y0v[1] + y0v[2], y0v[2] - y0v[1], y0v[1] - y0v[3],
                                 y1v[0] - y1v[2], y1v[1] + y1v[2], y1v[2] - y1v[1], y1v[1] - y1v[3]};
            float* dsx = sXt + tile * 64 + q4;
#pragma unroll
            for (int f = 0; f < TA; ++f) *reinterpret_cast<f32x4*>(dsx + f * 16 * 64) = bv[f];
        }
        __syncthreads();
        if (blk + 1 < blk1) load_block(blk + 1);  // in flight during the MFMAs below
        const float* ya = sYt + wo * 32 + l31;
        const float* xb = sXt + wc * 32 + l31;
        for (int kk = 0; 2 * kk < ntiles; ++kk) {
            const int t2 = (2 * kk + k) * 64;  // slots of tiles >= ntiles hold zeros
#pragma unroll
            for (int f = 0; f < TA; ++f)
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[f * 16 * 64 + t2], xb[f * 16 * 64 + t2], acc[f], 0, 0, 0);
        }
    }
    // ---- partial tile: D[i][j], j = lane % 32 (cin), i = 8 (v / 4) + 4 (lane / 32) + v % 4 (cout)
    const int ci = kt * 64 + wc * 32 + l31;
    if (ci < p.Cin) {
#pragma unroll
        for (int t = 0; t < TA; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int co = ct * 64 + wo * 32 + 8 * (v >> 2) + 4 * k + (v & 3);
                if (co < p.Cout) p.partial[(((size_t)split * T + zi * TA + t) * p.Cout + co) * p.Cin + ci] = acc[t][v];
            }
    }
}

// out[(o * Cin + c) * T + t] (= | +=) sum_split partial[split][t][o][c]: OIHW for 3x3 / 1x1, and the (Cout, 4 C) layout of
// the Downsample weight (index c*4 + p1*2 + p2) for T = 4.  One workgroup = 64 consecutive (cout, cin) pairs x ALL taps, so
// the result leaves as one contiguous run of 64 * T floats (a workgroup per tap wrote 4-byte pieces 4 * T bytes apart:
// 0.5 ms for the 143 MB of gradients).  T >= 4: thread (pair, q) owns the taps q, q + 4, .. and adds their splits in order;
// T < 4: the four q share the splits of one tap.  T = 16: the partial sums are in the Winograd domain (mode 3): the sixteen
// sums M[i][j] of a pair become its nine taps dW = G^T M G.  Fixed summation order throughout (no atomics).
__device__ __forceinline__ void wgrad_reduce_block(const WgradJob& j, int lb, float* tile) {
    const int T = j.T;
    const int TS = T == 16 ? 17 : T;  // row stride of the staged sums (16 would put every second pair on the same bank)
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)lb * 64, i = i0 + li;
    if (T >= 4) {
        for (int t = q; t < T; t += 4) {
            float s = 0.f;
            if (i < j.oc)
                for (int sp = 0; sp < j.splits; ++sp) s += j.partial[((int64_t)sp * T + t) * j.oc + i];
            tile[li * TS + t] = s;
        }
    } else {
        float* red = tile + 64 * 17;
        for (int t = 0; t < T; ++t) {
            float s = 0.f;
            if (i < j.oc)
                for (int sp = q; sp < j.splits; sp += 4) s += j.partial[((int64_t)sp * T + t) * j.oc + i];
            __syncthreads();
            red[q * 64 + li] = s;
            __syncthreads();
            if (q == 0) tile[li * T + t] = (red[li] + red[64 + li]) + (red[128 + li] + red[192 + li]);
        }
    }
    __syncthreads();
    const int pairs = (int)min((int64_t)64, j.oc - i0);
    if (T == 16) {
        float* o = j.out + i0 * 9;
        for (int k = threadIdx.x; k < pairs * 9; k += 256) {
            const int pr = k / 9, uv = k - pr * 9, u = uv / 3, v = uv - u * 3;
            const float* M = tile + pr * TS;
            // column u of G: (1, 1/2, 1/2, 0), (0, 1/2, -1/2, 0), (0, 1/2, 1/2, 1)
            float r[4];  // r[jj] = sum_i G[i][u] M[i][jj]
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float m0 = M[jj], m1 = M[4 + jj], m2 = M[8 + jj], m3 = M[12 + jj];
                r[jj] = u == 0 ? m0 + 0.5f * (m1 + m2) : (u == 1 ? 0.5f * (m1 - m2) : 0.5f * (m1 + m2) + m3);
            }
            const float val = v == 0 ? r[0] + 0.5f * (r[1] + r[2]) : (v == 1 ? 0.5f * (r[1] - r[2]) : 0.5f * (r[1] + r[2]) + r[3]);
            o[k] = j.accumulate ? o[k] + val : val;
        }
        return;
    }
    const int64_t n = (int64_t)pairs * T;
    float* o = j.out + i0 * T;
    for (int k = threadIdx.x; k < n; k += 256) o[k] = j.accumulate ? o[k] + tile[k] : tile[k];
}
// The same for 256 consecutive (cout, cin) pairs per workgroup, 16 bytes per lane (T >= 4, oc % 4 == 0): the 64-pair form
// is ~88 000 workgroups of 4 KB each for the U-Net's 3x3 layers -- bound by workgroup dispatch, not by the 0.4 GB it moves
// (0.31 ms at any batch); this one is a quarter of the workgroups with four times the bytes in flight per lane.
constexpr int WRP = 256;  // pairs per workgroup of the wide form
__device__ __forceinline__ void wgrad_reduce_block_wide(const WgradJob& j, int lb, float* tile) {
    const int T = j.T;
    const int TS = T | 1;  // odd row stride: the nine-tap gather below walks rows
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)lb * WRP, i = i0 + 4 * li;
    for (int t = q; t < T; t += 4) {
        f32x4 s = make_f32x4(0.f, 0.f, 0.f, 0.f);
        if (i < j.oc)
            for (int sp = 0; sp < j.splits; ++sp) s += *reinterpret_cast<const f32x4*>(j.partial + ((int64_t)sp * T + t) * j.oc + i);
        tile[(4 * li + 0) * TS + t] = s.x;
        tile[(4 * li + 1) * TS + t] = s.y;
        tile[(4 * li + 2) * TS + t] = s.z;
        tile[(4 * li + 3) * TS + t] = s.w;
    }
    __syncthreads();
    const int pairs = (int)min((int64_t)WRP, j.oc - i0);
    if (T == 16) {
        float* o = j.out + i0 * 9;
        for (int k = threadIdx.x; k < pairs * 9; k += 256) {
            const int pr = k / 9, uv = k - pr * 9, u = uv / 3, v = uv - u * 3;
            const float* M = tile + pr * TS;
            float r[4];  // r[jj] = sum_i G[i][u] M[i][jj]   (G columns as in wgrad_reduce_block)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float m0 = M[jj], m1 = M[4 + jj], m2 = M[8 + jj], m3 = M[12 + jj];
                r[jj] = u == 0 ? m0 + 0.5f * (m1 + m2) : (u == 1 ? 0.5f * (m1 - m2) : 0.5f * (m1 + m2) + m3);
            }
            const float val = v == 0 ? r[0] + 0.5f * (r[1] + r[2]) : (v == 1 ? 0.5f * (r[1] - r[2]) : 0.5f * (r[1] + r[2]) + r[3]);
            o[k] = j.accumulate ? o[k] + val : val;
        }
        return;
    }
    const int64_t n = (int64_t)pairs * T;
    float* o = j.out + i0 * T;
    for (int k = threadIdx.x; k < n; k += 256) {
        const int pr = k / T, t = k - pr * T;
        const float val = tile[pr * TS + t];
        o[k] = j.accumulate ? o[k] + val : val;
    }
}
// pairs one workgroup of the reduce kernels takes for a job
int wgrad_reduce_pairs(int T, long long oc) { return (T >= 4 && T <= 16 && oc % 4 == 0) ? WRP : 64; }

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradJob j) {
    __shared__ float tile[WRP * 17];
    if (j.T >= 4 && j.T <= 16 && j.oc % 4 == 0) wgrad_reduce_block_wide(j, blockIdx.x, tile);
    else wgrad_reduce_block(j, blockIdx.x, tile);
}
// the same sum for a table of layers: workgroup -> job by binary search over the block prefix
__global__ __launch_bounds__(256) void wgrad_reduce_jobs_kernel(const WgradJob* __restrict__ jobs, int n_jobs) {
    __shared__ float tile[WRP * 17];
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {  // last job whose first_block <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgradJob j = jobs[lo];
    if (j.T >= 4 && j.T <= 16 && j.oc % 4 == 0) wgrad_reduce_block_wide(j, blockIdx.x - j.first_block, tile);
    else wgrad_reduce_block(j, blockIdx.x - j.first_block, tile);
}
int launch_wgrad_reduce_jobs(const WgradJob* jobs_dev, int n_jobs, int total_blocks, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3(total_blocks), dim3(256), 0, s, jobs_dev, n_jobs);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// pixel-block geometry of a (Ho, Wo) output: R rows x TW columns of one image, or NB whole images when an image has at
// most 32 pixels (4x4 / 2x2 / 1x1 maps of the deep layers: a 16-pixel block would stage as much as it multiplies)
static int wgrad_items(int mode) { return mode == 0 ? 6 : (mode == 1 ? 4 : 16); }  // mode 3 stages per tile, not per window item
static int wgrad_zdim(int mode) { return mode == 0 ? 3 : (mode == 3 ? 2 : 1); }
struct WgradGeo {
    int TW, R, NB, tiles_x, tiles_y, n_blocks;
};
static WgradGeo wgrad_geo(int B, int Ho, int Wo, int mode) {
    WgradGeo g{};
    g.TW = std::min((Wo + 1) & ~1, mode == 3 ? 32 : 64);
    g.R = std::max(1, std::min(64 / g.TW, Ho));
    if (mode == 3) g.R = std::max(2, g.R & ~1);  // whole 2x2 tiles
    g.tiles_x = (Wo + g.TW - 1) / g.TW;
    g.tiles_y = (Ho + g.R - 1) / g.R;
    g.NB = 1;
    if (g.tiles_x == 1 && g.tiles_y == 1) {
        const int win = mode == 0 ? g.R * (g.TW + 2) : (mode == 2 ? 4 : 1) * g.R * g.TW;
        const int items = 16 * wgrad_items(mode);
        g.NB = std::max(1, std::min(std::min(64 / (g.R * g.TW), mode == 3 ? 64 : items / win), B));
    }
    g.n_blocks = ((B + g.NB - 1) / g.NB) * g.tiles_x * g.tiles_y;
    return g;
}

static size_t wgrad_ws_floats_mode(int B, int Ho, int Wo, int Cout, int Cin, int T, int mode, int* splits_out) {
    const WgradGeo g = wgrad_geo(B, Ho, Wo, mode);
    const int tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64) * wgrad_zdim(mode);
    // A workgroup hides its own staging (the next block is loaded during the MFMAs), so the split count only has to fill
    // the chip; every split writes a full [T][Cout][Cin] partial tile that the reduce kernel reads back.
    static const int min_bps = env_int("DM_WGRAD_MIN_BLOCKS", 2);
    // (swept at B=64: 256 / 512 / 768 / 1024 / 1536 workgroups -> 13.0 / 12.3 / 12.0 / 12.5 / 12.6 ms per loss+backward:
    // three 146-register workgroups fit a CU)
    static const int target = env_int("DM_WGRAD_TARGET_WGS", 768);
    int splits = std::max(1, std::min((g.n_blocks + min_bps - 1) / min_bps, (target + tiles - 1) / tiles));
    const int bps = (g.n_blocks + splits - 1) / splits;
    splits = (g.n_blocks + bps - 1) / bps;
    if (splits_out) *splits_out = splits;
    return (size_t)splits * T * Cout * Cin;
}
size_t wgrad_ws_floats(int B, int Ho, int Wo, int Cout, int Cin, int T, int* splits_out) {
    return wgrad_ws_floats_mode(B, Ho, Wo, Cout, Cin, T, T == 9 ? 0 : (T == 1 ? 1 : (T == 4 ? 2 : 3)), splits_out);
}

static int wgrad_taps(int mode) { return mode == 0 ? 9 : (mode == 1 ? 1 : (mode == 2 ? 4 : 16)); }
static size_t wgrad_lds_bytes(const WgradParams& p, int mode) {
    if (mode == 3) return (size_t)2 * 8 * 16 * 64 * sizeof(float);  // transformed dY and X tiles
    const int WW = mode == 0 ? p.TW + 2 : (mode == 2 ? 2 * p.TW : p.TW);
    const int WH = mode == 2 ? 2 * p.R : p.R;
    return (size_t)(64 * WG_SY + p.NB * WH * WW * WG_SX) * sizeof(float);
}
// geometry of one layer's weight gradient for `splits` pixel splits (ws: splits * T * Cout * Cin floats)
static int wgrad_fill(WgradParams& p, const float* in0, int C0, const float* in1, int C1, const float* dy, int Cout, int B, int Ho,
                      int Wo, int mode, int up, float* ws, int splits) {
    DM_REQUIRE(C0 % 4 == 0 && C1 % 4 == 0 && Cout % 4 == 0 && C0 > 0, "wgrad: channel counts must be multiples of 4");
    DM_REQUIRE(mode >= 0 && mode <= 3 && (!up || ((mode == 0 || mode == 3) && Ho % 2 == 0 && Wo % 2 == 0)), "wgrad: mode");
    DM_REQUIRE(mode != 3 || (Ho % 2 == 0 && Wo % 2 == 0), "wgrad: the Winograd form needs even image sizes");
    DM_REQUIRE((size_t)B * Ho * Wo * (size_t)std::max(Cout, C0 + C1) * (mode == 2 ? 4 : 1) < (1ull << 40), "wgrad: size");
    p = WgradParams{};
    p.in0 = in0; p.in1 = C1 ? in1 : in0; p.dy = dy; p.partial = ws;
    p.C0 = C0; p.C1 = C1; p.Cin = C0 + C1; p.Cout = Cout;
    p.B = B; p.Ho = Ho; p.Wo = Wo; p.up = up;
    const WgradGeo g = wgrad_geo(B, Ho, Wo, mode);
    p.TW = g.TW; p.R = g.R; p.NB = g.NB;
    p.tiles_x = g.tiles_x;
    p.tiles_y = g.tiles_y;
    p.n_blocks = g.n_blocks;
    p.n_ct = (Cout + 63) / 64;
    p.n_kt = (p.Cin + 63) / 64;
    p.blocks_per_split = (p.n_blocks + splits - 1) / splits;
    p.n_splits = splits;
    const int WW = mode == 0 ? p.TW + 2 : (mode == 2 ? 2 * p.TW : p.TW);
    const int WH = mode == 2 ? 2 * p.R : p.R;
    DM_REQUIRE(wgrad_lds_bytes(p, mode) <= 160 * 1024, "wgrad: LDS");
    DM_REQUIRE(p.NB * p.R * p.TW <= 64 && (mode == 3 || p.NB * WH * WW <= 16 * wgrad_items(mode)) && p.TW < 4096 && WH < 4096,
               "wgrad: block larger than the staging items");
    DM_REQUIRE(mode != 3 || (p.R % 2 == 0 && p.TW % 2 == 0), "wgrad: the Winograd form works on whole 2x2 tiles");
    return 0;
}

// mode: 0 3x3 pad 1 (up: nearest x2 source), 1 1x1, 2 2x2 stride 2 (Ho, Wo = OUTPUT size; the source is 2Ho x 2Wo)
int launch_wgrad(const float* in0, int C0, const float* in1, int C1, const float* dy, int Cout, int B, int Ho, int Wo,
                 int mode, int up, float* ws, float* dw, int accumulate, hipStream_t s, WgradJob* defer) {
    const int T = wgrad_taps(mode);
    int splits = 1;
    (void)wgrad_ws_floats_mode(B, Ho, Wo, Cout, C0 + C1, T, mode, &splits);
    WgradParams p;
    if (wgrad_fill(p, in0, C0, in1, C1, dy, Cout, B, Ho, Wo, mode, up, ws, splits)) return 1;
    const size_t lds = wgrad_lds_bytes(p, mode);
    const dim3 grid(p.n_ct * p.n_kt, splits, wgrad_zdim(mode));
    const bool timed = prof::enabled();
    if (timed && prof::begin("wgrad_mfma_kernel", 2.0 * T * p.Cin * Cout * (double)B * Ho * Wo,
                             4.0 * ((double)B * Ho * Wo * (p.Cin * (mode == 2 ? 4 : 1) + Cout) + (double)T * p.Cin * Cout), s))
        return 1;
    static LdsOptIn f0, f1, f2, f3;
    if (mode == 0) {
        if (lds_opt_in(f0, reinterpret_cast<const void*>(wgrad_mfma_kernel<0, false>), 1)) return 1;
        hipLaunchKernelGGL((wgrad_mfma_kernel<0, false>), grid, dim3(256), lds, s, p, nullptr, 1);
    } else if (mode == 3) {
        if (lds_opt_in(f3, reinterpret_cast<const void*>(wgrad_wino_kernel<false>), 1)) return 1;
        hipLaunchKernelGGL((wgrad_wino_kernel<false>), grid, dim3(256), lds, s, p, nullptr, 1);
    } else if (mode == 1) {
        if (lds_opt_in(f1, reinterpret_cast<const void*>(wgrad_mfma_kernel<1, false>), 1)) return 1;
        hipLaunchKernelGGL((wgrad_mfma_kernel<1, false>), grid, dim3(256), lds, s, p, nullptr, 1);
    } else {
        if (lds_opt_in(f2, reinterpret_cast<const void*>(wgrad_mfma_kernel<2, false>), 1)) return 1;
        hipLaunchKernelGGL((wgrad_mfma_kernel<2, false>), grid, dim3(256), lds, s, p, nullptr, 1);
    }
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    const int64_t oc = (int64_t)Cout * p.Cin;
    if (defer) {
        *defer = WgradJob{ws, dw, oc, splits, T, accumulate, 0};
        return 0;
    }
    const int rp = wgrad_reduce_pairs(T, (long long)oc);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((oc + rp - 1) / rp)), dim3(256), 0, s,
                       WgradJob{ws, dw, oc, splits, T, accumulate, 0});
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- grouped form: the weight gradients of many layers of one mode in one launch ---------------------------------------
// Pass 1 (wgrad_group_plan): splits per layer such that every workgroup gets about the same number of pixel blocks and the
// launch has about `target` workgroups; returns the floats of partial-sum workspace each layer needs.
void wgrad_group_plan(const std::vector<WgradDesc>& descs, int mode, std::vector<int>& splits, std::vector<size_t>& ws_floats) {
    static const int target = env_int("DM_WGRAD_GROUP_WGS", 2048);
    const int T = wgrad_taps(mode);
    long long units = 0;
    std::vector<WgradGeo> geo(descs.size());
    for (size_t j = 0; j < descs.size(); ++j) {
        const WgradDesc& d = descs[j];
        geo[j] = wgrad_geo(d.B, d.Ho, d.Wo, mode);
        const int tiles = ((d.Cout + 63) / 64) * ((d.C0 + d.C1 + 63) / 64) * wgrad_zdim(mode);
        units += (long long)tiles * geo[j].n_blocks;
    }
    const int bps = (int)std::max<long long>(1, (units + target - 1) / target);
    splits.resize(descs.size());
    ws_floats.resize(descs.size());
    for (size_t j = 0; j < descs.size(); ++j) {
        const int sp = (geo[j].n_blocks + bps - 1) / bps;
        const int per = (geo[j].n_blocks + sp - 1) / sp;
        splits[j] = (geo[j].n_blocks + per - 1) / per;
        ws_floats[j] = (size_t)splits[j] * T * descs[j].Cout * (descs[j].C0 + descs[j].C1);
    }
}
// Pass 2: the parameter table (host) of the launch and the split-K sums it leaves behind
int wgrad_group_fill(const std::vector<WgradDesc>& descs, int mode, const std::vector<int>& splits, const std::vector<float*>& ws,
                     std::vector<WgradParams>& table, std::vector<WgradJob>& jobs, int* total_wgs, size_t* lds_bytes) {
    const int T = wgrad_taps(mode);
    int first = 0;
    size_t lds = 0;
    table.resize(descs.size());
    for (size_t j = 0; j < descs.size(); ++j) {
        const WgradDesc& d = descs[j];
        if (wgrad_fill(table[j], d.in0, d.C0, d.in1, d.C1, d.dy, d.Cout, d.B, d.Ho, d.Wo, mode, d.up, ws[j], splits[j])) return 1;
        table[j].first_wg = first;
        first += table[j].n_ct * table[j].n_kt * splits[j] * wgrad_zdim(mode);
        lds = std::max(lds, wgrad_lds_bytes(table[j], mode));
        jobs.push_back(WgradJob{ws[j], d.dw, (long long)d.Cout * (d.C0 + d.C1), splits[j], T, d.accumulate, 0});
    }
    *total_wgs = first;
    *lds_bytes = lds;
    return 0;
}
int launch_wgrad_group(const WgradParams* table_dev, int n_jobs, int total_wgs, size_t lds_bytes, int mode, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    static LdsOptIn f0, f1, f2, f3;
    const WgradParams none{};
    if (mode == 0) {
        if (lds_opt_in(f0, reinterpret_cast<const void*>(wgrad_mfma_kernel<0, true>), 1)) return 1;
        hipLaunchKernelGGL((wgrad_mfma_kernel<0, true>), dim3(total_wgs), dim3(256), lds_bytes, s, none, table_dev, n_jobs);
    } else if (mode == 3) {
        if (lds_opt_in(f3, reinterpret_cast<const void*>(wgrad_wino_kernel<true>), 1)) return 1;
        hipLaunchKernelGGL((wgrad_wino_kernel<true>), dim3(total_wgs), dim3(256), lds_bytes, s, none, table_dev, n_jobs);
    } else if (mode == 1) {
        if (lds_opt_in(f1, reinterpret_cast<const void*>(wgrad_mfma_kernel<1, true>), 1)) return 1;
        hipLaunchKernelGGL((wgrad_mfma_kernel<1, true>), dim3(total_wgs), dim3(256), lds_bytes, s, none, table_dev, n_jobs);
    } else {
        if (lds_opt_in(f2, reinterpret_cast<const void*>(wgrad_mfma_kernel<2, true>), 1)) return 1;
        hipLaunchKernelGGL((wgrad_mfma_kernel<2, true>), dim3(total_wgs), dim3(256), lds_bytes, s, none, table_dev, n_jobs);
    }
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- general (slow) form for the two thin convolutions of the U-Net: init_conv (7x7 over the NCHW image, 3-8 input
// channels) and final_conv (1x1, 3 outputs, NCHW gradient).  One thread per weight element, pixels split over blockIdx.y.
struct WgradNaive {
    const float* x;
    const float* dy;
    float* partial;  // [split][n_out]
    int Cin, Cout, KH, KW, pad;
    int B, H, W;     // input == output size (stride 1, "same" padding)
    int64_t xs_b, xs_c, xs_p;     // strides of x: batch, channel, pixel
    int64_t dys_b, dys_c, dys_p;  // strides of dy
    int rows_per_split;           // image rows (b * H + y) per split
};

__global__ void wgrad_naive_kernel(const WgradNaive p) {
    const int n_out = p.Cout * p.Cin * p.KH * p.KW;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int kx = i % p.KW, ky = (i / p.KW) % p.KH, c = (i / (p.KW * p.KH)) % p.Cin, o = i / (p.KW * p.KH * p.Cin);
    const int row0 = blockIdx.y * p.rows_per_split, row1 = min(row0 + p.rows_per_split, p.B * p.H);
    float s = 0.f;
    for (int row = row0; row < row1; ++row) {
        const int b = row / p.H, y = row - b * p.H;
        const int sy = y + ky - p.pad;
        if (sy < 0 || sy >= p.H) continue;
        const float* xr = p.x + b * p.xs_b + c * p.xs_c + (int64_t)sy * p.W * p.xs_p;
        const float* dr = p.dy + b * p.dys_b + o * p.dys_c + (int64_t)y * p.W * p.dys_p;
        const int xlo = max(0, p.pad - kx), xhi = min(p.W, p.W + p.pad - kx);
        for (int x = xlo; x < xhi; ++x) s += dr[x * p.dys_p] * xr[(x + kx - p.pad) * p.xs_p];
    }
    p.partial[(size_t)blockIdx.y * n_out + i] = s;
}

// init_conv (DD/denoising_diffusion.py:262): x NCHW with a handful of channels, dy NHWC.  One wave = one input channel c x 64
// output channels (lane) x a slice of the image rows, KH x KW accumulators per lane: per output row the KH zero-padded
// input rows are staged in LDS, a dy value is loaded once (coalesced over the channels) and meets the KH x KW neighbouring
// x values (LDS broadcasts).  dy is read Cin times in all (it was Cin x KH times with one kernel row per workgroup).
template <int KH, int KW>
__global__ __launch_bounds__(64) void wgrad_init_kernel(const WgradNaive p) {
    extern __shared__ float win[];  // [KH][W + KW - 1]
    const int c = blockIdx.x, o = blockIdx.z * 64 + threadIdx.x;
    const bool o_ok = o < p.Cout;
    const int WP = p.W + KW - 1;
    const int row0 = blockIdx.y * p.rows_per_split, row1 = min(row0 + p.rows_per_split, p.B * p.H);
    float acc[KH][KW];
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int k = 0; k < KW; ++k) acc[a][k] = 0.f;
    for (int row = row0; row < row1; ++row) {
        const int b = row / p.H, y = row - b * p.H;
        __syncthreads();
        for (int i = threadIdx.x; i < KH * WP; i += 64) {
            const int ky = i / WP, sx = i - ky * WP - p.pad, sy = y + ky - p.pad;
            win[i] = (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) ? p.x[b * p.xs_b + c * p.xs_c + (int64_t)sy * p.W + sx] : 0.f;
        }
        __syncthreads();
        const float* dr = p.dy + ((int64_t)row * p.W) * p.Cout + (o_ok ? o : 0);
        for (int x0 = 0; x0 < p.W; x0 += 16) {  // 16 dy values in flight, then 16 x KH x KW products on a sliding LDS window
            float d[16];
#pragma unroll
            for (int xx = 0; xx < 16; ++xx) d[xx] = x0 + xx < p.W ? dr[(int64_t)(x0 + xx) * p.Cout] : 0.f;
#pragma unroll
            for (int a = 0; a < KH; ++a) {
                float wv[16 + KW - 1];
#pragma unroll
                for (int j = 0; j < 16 + KW - 1; ++j) wv[j] = x0 + j < WP ? win[a * WP + x0 + j] : 0.f;
#pragma unroll
                for (int xx = 0; xx < 16; ++xx)
#pragma unroll
                    for (int k = 0; k < KW; ++k) acc[a][k] += d[xx] * wv[xx + k];
            }
        }
    }
    if (o_ok) {
        float* q = p.partial + (size_t)blockIdx.y * p.Cout * p.Cin * KH * KW + ((size_t)o * p.Cin + c) * KH * KW;
#pragma unroll
        for (int a = 0; a < KH; ++a)
#pragma unroll
            for (int k = 0; k < KW; ++k) q[a * KW + k] = acc[a][k];
    }
}
// init_conv on the matrix core (v_mfma_f32_32x32x2_f32): dW[o][n] = sum_px dy[px][o] * X[px + tap(n)], n = (c, ky, kx) -- the
// 147 (Cin = 3) or 294 (with self-conditioning) columns are 5 / 10 tiles of 32, the 64 couts of the workgroup 2 tiles, a pair of
// neighbouring pixels of one image row the K index.  Wave (mt, nh) of the 4 owns cout tile mt and the column tiles nh, nh + 2,
// ..: A = dy (registers, one row of the image at a time, loaded while the previous row multiplies), B = the zero-padded 7-row
// window of every input channel in LDS (lane (n, k) reads win[c][ky][2 s + k + kx]), double-buffered: one barrier per row.
// Same split partials as wgrad_init_kernel ([split][Cout][Cin][7][7]); 89 -> 20 us at B = 64, 32x32.
template <int NTW, int NS>
__global__ __launch_bounds__(256) void wgrad_init_mfma_kernel(const WgradNaive p) {
    extern __shared__ float win[];  // [2][Cin * 7 * WP + 2 NS + 2]: the tail of a buffer stays zero (columns n >= Cin * 49)
    constexpr int SI = 12;          // window staging items per thread (checked by the launcher)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int mt = wave & 1, nh = wave >> 1;
    const int WP = 2 * NS + 6, WIN = p.Cin * 7 * WP, WBUF = WIN + 2 * NS + 2;
    const int NTOT = p.Cin * 49;
    const int row0 = blockIdx.y * p.rows_per_split, row1 = min(row0 + p.rows_per_split, p.B * p.H);
    const int co = blockIdx.z * 64 + mt * 32 + l31;
    const bool co_ok = co < p.Cout;
    int base[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = (nh + 2 * j) * 32 + l31;
        const int c = n / 49, r = n - c * 49, ky = r / 7, kx = r - ky * 7;
        base[j] = (n < NTOT ? (c * 7 + ky) * WP + kx : WIN) + kh;  // WIN: the zero tail
    }
    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int i = tid; i < 2 * WBUF; i += 256) win[i] = 0.f;

    float wreg[SI], areg[NS];
    auto fetch = [&](int row) {  // the row's window (all channels) and its dy values into registers
        const int b = row / p.H, y = row - b * p.H;
#pragma unroll
        for (int q = 0; q < SI; ++q) {
            const int i = tid + 256 * q;
            float v = 0.f;
            if (i < WIN) {
                const int cky = i / WP, sx = i - cky * WP - 3, c = cky / 7, sy = y + (cky - c * 7) - 3;
                if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) v = p.x[b * p.xs_b + c * p.xs_c + (int64_t)sy * p.W + sx];
            }
            wreg[q] = v;
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int x = 2 * st + kh;
            areg[st] = (co_ok && x < p.W) ? p.dy[((int64_t)row * p.W + x) * p.Cout + co] : 0.f;
        }
    };
    if (row0 < row1) fetch(row0);
    __syncthreads();  // the zero fill
    for (int row = row0; row < row1; ++row) {
        float* wb = win + ((row - row0) & 1) * WBUF;
#pragma unroll
        for (int q = 0; q < SI; ++q)
            if (tid + 256 * q < WIN) wb[tid + 256 * q] = wreg[q];
        float a[NS];
#pragma unroll
        for (int st = 0; st < NS; ++st) a[st] = areg[st];
        __syncthreads();
        if (row + 1 < row1) fetch(row + 1);
#pragma unroll
        for (int st = 0; st < NS; ++st)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], wb[base[j] + 2 * st], acc[j], 0, 0, 0);
            }
    }
    // D register e of lane (n = l31, half kh) = cout row (e & 3) + 8 (e >> 2) + 4 kh of the tile, column n
    float* out = p.partial + (size_t)blockIdx.y * p.Cout * NTOT;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = (nh + 2 * j) * 32 + l31;
        if (n >= NTOT) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int o = blockIdx.z * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (o < p.Cout) out[(size_t)o * NTOT + n] = acc[j][e];
        }
    }
}
// column tiles per wave needed for Cin channels (0: the VALU kernel takes the layer)
static int wgrad_init_mfma_ntw(int Cin, int W) {
    const int tiles = (Cin * 49 + 31) / 32, ntw = (tiles + 1) / 2;
    const int ns = W <= 32 ? 16 : 32;
    if (W > 64 || ntw > 5 || Cin * 7 * (2 * ns + 6) > 12 * 256) return 0;
    return ntw <= 3 ? 3 : 5;
}

// final_conv (:343): 1x1 with a handful of outputs, x NHWC, dy NCHW.  256 threads = 64 input channels x 4 pixel lanes.
__global__ __launch_bounds__(256) void wgrad_thin_out_kernel(const WgradNaive p) {
    __shared__ float red[4][4][64];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + li;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t px0 = (int64_t)blockIdx.y * p.rows_per_split * p.W, px1 = min(px0 + (int64_t)p.rows_per_split * p.W, p.B * HW);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < p.Cin)
        for (int64_t px = px0 + q; px < px1; px += 4) {
            const float xv = p.x[px * p.Cin + c];
            const int64_t b = px / HW, r = px - b * HW;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < p.Cout) acc[o] += xv * p.dy[(b * p.Cout + o) * HW + r];
        }
#pragma unroll
    for (int o = 0; o < 4; ++o) red[o][q][li] = acc[o];
    __syncthreads();
    if (q == 0 && c < p.Cin)
        for (int o = 0; o < p.Cout; ++o)
            p.partial[(size_t)blockIdx.y * p.Cout * p.Cin + (size_t)o * p.Cin + c] =
                (red[o][0][li] + red[o][1][li]) + (red[o][2][li] + red[o][3][li]);
}

// final_conv (:343) backward in one pass over its input: weight gradient, bias gradient and input gradient of a 1x1 convolution
// with at most 4 outputs, x NHWC, dy NCHW (the loss gradient).  256 threads = 64 input channels x 4 pixel lanes; a thread keeps 16
// pixels in flight (x and the Cout dy values of each), and the pixel's dx row -- sum_o w[o][c] dy[o][px] -- leaves with the same
// registers.  Split partials: [split][Cout * Cin] then [split][Cout] (bias), summed by the reduce jobs in split order.
struct ThinOutBwd {
    const float *x, *dy, *w;  // x (B, H, W, Cin), dy (B, Cout, H, W), w (Cout, Cin)
    float *dx, *part_w, *part_b;
    int Cin, Cout, HW, px_per_split;
    long long pixels;
};
__global__ __launch_bounds__(256) void thin_out_bwd_kernel(const ThinOutBwd p) {
    __shared__ float red[4][4][64];
    __shared__ float redb[4][4];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + li;
    const bool cok = c < p.Cin;
    const long long px0 = (long long)blockIdx.y * p.px_per_split, px1 = min(px0 + (long long)p.px_per_split, p.pixels);
    float wv[4], acc[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 4; ++o) wv[o] = (cok && o < p.Cout) ? p.w[o * p.Cin + c] : 0.f;
    long long px = px0 + q;
    long long b = px / p.HW;
    int r = (int)(px - b * p.HW);
    const int cc = cok ? c : 0;
    while (px < px1) {  // branch-free body: clamped addresses, masked values (HW >= 4: one wrap per step at most)
        constexpr int NP = 16;  // pixels in flight per thread
        float xv[NP], dv[NP][4];
        long long pxs[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const bool ok = px < px1;
            const long long pc = ok ? px : px1 - 1;
            const long long bc = ok ? b : 0;
            const int rc = ok ? r : 0;
            pxs[k] = ok ? px : -1;
            xv[k] = p.x[pc * p.Cin + cc];
#pragma unroll
            for (int o = 0; o < 4; ++o) dv[k][o] = p.dy[(bc * p.Cout + min(o, p.Cout - 1)) * p.HW + rc];
            px += 4;
            r += 4;
            const bool wrap = r >= p.HW;
            r = wrap ? r - p.HW : r;
            b = wrap ? b + 1 : b;
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const float m = pxs[k] >= 0 ? 1.f : 0.f;
            float d = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float dvm = dv[k][o] * m;
                acc[o] += xv[k] * dvm;
                bs[o] += dvm;
                d += wv[o] * dvm;  // wv[o] = 0 for o >= Cout
            }
            if (pxs[k] >= 0 && cok) p.dx[pxs[k] * p.Cin + c] = d;
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) red[o][q][li] = acc[o];
    if (li == 0)
#pragma unroll
        for (int o = 0; o < 4; ++o) redb[o][q] = bs[o];
    __syncthreads();
    if (q == 0 && cok)
        for (int o = 0; o < p.Cout; ++o)
            p.part_w[(size_t)blockIdx.y * p.Cout * p.Cin + (size_t)o * p.Cin + c] =
                (red[o][0][li] + red[o][1][li]) + (red[o][2][li] + red[o][3][li]);
    if (blockIdx.x == 0 && threadIdx.x < p.Cout) {
        const int o = threadIdx.x;
        p.part_b[(size_t)blockIdx.y * p.Cout + o] = (redb[o][0] + redb[o][1]) + (redb[o][2] + redb[o][3]);
    }
}
static int thin_out_bwd_splits(long long pixels) { return (int)std::min<long long>(256, std::max<long long>(1, pixels / 64)); }
size_t thin_out_bwd_ws_floats(int B, int H, int W, int Cout, int Cin) {
    return (size_t)thin_out_bwd_splits((long long)B * H * W) * ((size_t)Cout * Cin + Cout);
}
bool thin_out_bwd_ok(int Cout, int HW) { return Cout >= 1 && Cout <= 4 && HW >= 4; }
// defer_w / defer_b: both set (the training step's grouped split sums) or both null (summed here)
int launch_thin_out_bwd(const float* x, const float* dy_nchw, const float* w_oc, float* dx, float* ws, float* dw, float* db,
                        int B, int H, int W, int Cin, int Cout, int accumulate, hipStream_t s, WgradJob* defer_w,
                        WgradJob* defer_b) {
    DM_REQUIRE(thin_out_bwd_ok(Cout, H * W), "thin_out_bwd: 1 to 4 outputs");
    ThinOutBwd p{};
    p.pixels = (long long)B * H * W;
    const int splits0 = thin_out_bwd_splits(p.pixels);
    p.px_per_split = (int)((p.pixels + splits0 - 1) / splits0);
    const int splits = (int)((p.pixels + p.px_per_split - 1) / p.px_per_split);
    p.x = x; p.dy = dy_nchw; p.w = w_oc; p.dx = dx;
    p.part_w = ws; p.part_b = ws + (size_t)splits * Cout * Cin;
    p.Cin = Cin; p.Cout = Cout; p.HW = H * W;
    hipLaunchKernelGGL(thin_out_bwd_kernel, dim3((Cin + 63) / 64, splits), dim3(256), 0, s, p);
    DM_CHECK_HIP(hipGetLastError());
    const WgradJob jw{p.part_w, dw, (long long)Cout * Cin, splits, 1, accumulate, 0};
    const WgradJob jb{p.part_b, db, (long long)Cout, splits, 1, accumulate, 0};
    if (defer_w && defer_b) {
        *defer_w = jw;
        *defer_b = jb;
        return 0;
    }
    const int rpn = wgrad_reduce_pairs(1, 1);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((Cout * Cin + rpn - 1) / rpn), dim3(256), 0, s, jw);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((Cout + rpn - 1) / rpn), dim3(256), 0, s, jb);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

size_t wgrad_naive_ws_floats(int B, int H, int Cout, int Cin, int KH, int KW, int* splits_out) {
    const int rows = B * H;
    int splits = std::min(rows, 256);
    const int rps = (rows + splits - 1) / splits;
    splits = (rows + rps - 1) / rps;
    if (splits_out) *splits_out = splits;
    return (size_t)splits * Cout * Cin * KH * KW;
}

// x: NCHW when x_nchw else NHWC; dy likewise
int launch_wgrad_naive(const float* x, int x_nchw, const float* dy, int dy_nchw, int Cin, int Cout, int KH, int KW, int pad,
                       int B, int H, int W, float* ws, float* dw, int accumulate, hipStream_t s, WgradJob* defer) {
    DM_REQUIRE(KH == 2 * pad + 1 && KW == 2 * pad + 1, "naive wgrad: same-size convolutions only");
    WgradNaive p{};
    p.x = x; p.dy = dy; p.partial = ws;
    p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.pad = pad; p.B = B; p.H = H; p.W = W;
    const int64_t HW = (int64_t)H * W;
    if (x_nchw) { p.xs_b = Cin * HW; p.xs_c = HW; p.xs_p = 1; } else { p.xs_b = HW * Cin; p.xs_c = 1; p.xs_p = Cin; }
    if (dy_nchw) { p.dys_b = Cout * HW; p.dys_c = HW; p.dys_p = 1; } else { p.dys_b = HW * Cout; p.dys_c = 1; p.dys_p = Cout; }
    int splits = 1;
    (void)wgrad_naive_ws_floats(B, H, Cout, Cin, KH, KW, &splits);
    p.rows_per_split = (B * H + splits - 1) / splits;
    const int n_out = Cout * Cin * KH * KW;
    static const bool no_init_mfma = std::getenv("DM_WGRAD_INIT_VALU") != nullptr;
    const int ntw = (x_nchw && !dy_nchw && KW == 7 && KH == 7 && pad == 3 && !no_init_mfma) ? wgrad_init_mfma_ntw(Cin, W) : 0;
    if (ntw) {
        const int ns = W <= 32 ? 16 : 32;
        const size_t lds = 2 * ((size_t)Cin * 7 * (2 * ns + 6) + 2 * ns + 2) * sizeof(float);
        const dim3 grid(1, splits, (Cout + 63) / 64);
        if (ntw == 3 && ns == 16) hipLaunchKernelGGL((wgrad_init_mfma_kernel<3, 16>), grid, dim3(256), lds, s, p);
        else if (ntw == 3) hipLaunchKernelGGL((wgrad_init_mfma_kernel<3, 32>), grid, dim3(256), lds, s, p);
        else if (ns == 16) hipLaunchKernelGGL((wgrad_init_mfma_kernel<5, 16>), grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL((wgrad_init_mfma_kernel<5, 32>), grid, dim3(256), lds, s, p);
    } else if (x_nchw && !dy_nchw && KW == 7 && KH == 7 && (size_t)7 * (W + 6) * sizeof(float) <= 48 * 1024)
        hipLaunchKernelGGL((wgrad_init_kernel<7, 7>), dim3(Cin, splits, (Cout + 63) / 64), dim3(64), 7 * (W + 6) * sizeof(float), s,
                           p);
    else if (!x_nchw && dy_nchw && KH == 1 && KW == 1 && Cout <= 4)
        hipLaunchKernelGGL(wgrad_thin_out_kernel, dim3((Cin + 63) / 64, splits), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(wgrad_naive_kernel, dim3((n_out + 127) / 128, splits), dim3(128), 0, s, p);
    DM_CHECK_HIP(hipGetLastError());
    if (defer) {
        *defer = WgradJob{ws, dw, (long long)n_out, splits, 1, accumulate, 0};
        return 0;
    }
    const int rpn = wgrad_reduce_pairs(1, 1);  // T = 1 jobs of the thin layers: the 64-pair form
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n_out + rpn - 1) / rpn), dim3(256), 0, s,
                       WgradJob{ws, dw, (long long)n_out, splits, 1, accumulate, 0});
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dm

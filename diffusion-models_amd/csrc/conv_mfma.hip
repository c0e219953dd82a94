// Implicit-GEMM convolution for gfx950 on the f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces every nn.Conv2d on the reference's sampling path (conv3x3 of Block,
// DD/denoising_diffusion.py:108; 1x1 res_conv :134, to_qkv :166, to_out :169; 7x7 init_conv :262;
// Downsample = pixel-unshuffle + 1x1 == 2x2 stride-2 conv with permuted weights :54-58;
// Upsample = nearest x2 folded into the gather :48-52), with the channel concat of the up path
// (:378,:381,:387) folded into a two-source gather and the Block epilogue
// (RMSNorm :66-67, scale/shift :117-119, SiLU) fused when one workgroup owns all output channels.
//
// Mapping (one workgroup = 256 threads = 4 waves, wave tile 64 pixels x 64 couts = 2x2 MFMA tiles):
//   MFMA rows  (A operand)  = output pixels     A[i = lane&31][k = lane>>5]
//   MFMA cols  (B operand)  = output channels   B[k = lane>>5][j = lane&31]
//   reduction               = (tap, cin); per K chunk of CK input channels the lane half h = lane>>5
//                             owns channels [h*CK/2, (h+1)*CK/2) of the chunk, so MFMA step s multiplies
//                             channels {s, CK/2 + s} -- a fixed permutation of the fp32 summation order.
//   accumulator             acc[r][p][reg]: pixel row (reg&3) + 8*(reg>>2) + 4*(lane>>5) of row-tile r,
//                           cout (lane&31) of col-tile p.
// LDS: the input window of the tile (NB images x IH x IW pixels x CK channels, row padded to CK+4
// floats so that ds_read_b128 of 16 consecutive pixels is bank-conflict free) is staged ONCE per
// channel chunk and re-used by all KH*KW taps.  Weights stream through two LDS buffers, one
// "slab" = (chunk, ky, TPS taps) at a time.
// Pipeline: the global loads of slab i+1 (and of the next chunk's input window) are issued into
// registers BEFORE the MFMAs of slab i and written to LDS after them, so HBM/L2 latency hides under
// the matrix work of the same wave; one barrier per slab.
// Split-K: blockIdx.y owns a contiguous range of channel chunks and writes raw partial sums; the
// epilogue then runs in norm_act_kernel (elementwise.hip), which sums the partials.
#include "conv_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace dm {

static inline int pow2ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
static inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline int ckp_for(int CK) { return CK == 4 ? 4 : CK + 4; }
static inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

int conv_ck_for(int C0, int C1) { return (C0 % 16 == 0 && C1 % 16 == 0) ? 16 : 4; }

// halo / weight staging registers (f32x4 per thread) per instantiation
static constexpr int hregs_for(int WM) { return WM == 4 ? 9 : (WM == 2 ? 5 : 3); }
static constexpr int wregs_for(int WN, int CK) { return CK == 16 ? 3 * WN : 2 * WN; }

// ---- packing: OIHW -> [chunk][ky][kx][CoutP][CK], CoutP = Cout padded to 64, zero filled ----------
size_t conv_packed_floats(int Cout, int C0, int C1, int KH, int KW) {
    int CK = conv_ck_for(C0, C1);
    int chunks = pad_to(C0, CK) / CK + (C1 ? pad_to(C1, CK) / CK : 0);
    return (size_t)chunks * KH * KW * pad_to(Cout, 256) * CK;
}

void conv_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1, int KH, int KW) {
    int CK = conv_ck_for(C0, C1);
    int chunks0 = pad_to(C0, CK) / CK;
    int chunks1 = C1 ? pad_to(C1, CK) / CK : 0;
    int chunks = chunks0 + chunks1;
    int Cin = C0 + C1;
    int CoutP = pad_to(Cout, 256);
    std::memset(packed, 0, conv_packed_floats(Cout, C0, C1, KH, KW) * sizeof(float));
    for (int ch = 0; ch < chunks; ++ch)
        for (int ky = 0; ky < KH; ++ky)
            for (int kx = 0; kx < KW; ++kx) {
                float* dst = packed + (((size_t)ch * KH + ky) * KW + kx) * CoutP * CK;
                for (int co = 0; co < Cout; ++co)
                    for (int kk = 0; kk < CK; ++kk) {
                        int cin;
                        if (ch < chunks0) {
                            int c = ch * CK + kk;
                            if (c >= C0) continue;
                            cin = c;
                        } else {
                            int c = (ch - chunks0) * CK + kk;
                            if (c >= C1) continue;
                            cin = C0 + c;
                        }
                        dst[(size_t)co * CK + kk] = oihw[(((size_t)co * Cin + cin) * KH + ky) * KW + kx];
                    }
            }
}

// ---- tiling policy ----------------------------------------------------------------------------------
static void fill_tile(ConvGeom& g, int B, int Ho, int Wo, int Cout, int KH, int KW, int stride) {
    const int MT = 64 * g.WM, NT = 64 * g.WN;
    g.TW = std::min(std::min(pow2ceil(Wo), 32), MT);
    g.TH = std::min(pow2ceil(Ho), MT / g.TW);
    g.NB = MT / (g.TW * g.TH);
    g.lTW = ilog2(g.TW);
    g.lTH = ilog2(g.TH);
    g.tiles_x = (Wo + g.TW - 1) / g.TW;
    g.tiles_y = (Ho + g.TH - 1) / g.TH;
    g.groups = (B + g.NB - 1) / g.NB;
    g.n_tiles_n = (Cout + NT - 1) / NT;
    g.IH = (g.TH - 1) * stride + KH;
    g.IW = (g.TW - 1) * stride + KW;
}
static int wgs_of(const ConvGeom& g) { return g.tiles_x * g.tiles_y * g.groups * g.n_tiles_n; }

ConvGeom conv_plan(int B, int Ho, int Wo, int Cout, int KH, int KW, int stride, int C0, int C1, bool want_norm,
                   bool allow_split, int grid_z) {
    static const int target_wgs = env_int("DM_CONV_TARGET_WGS", 512);
    static const int min_fused_wgs = env_int("DM_CONV_MIN_FUSED_WGS", 512);
    static const int force_tps = env_int("DM_CONV_TPS", 0);
    static const int max_splits = env_int("DM_CONV_MAX_SPLITS", 8);
    ConvGeom g{};
    g.CK = conv_ck_for(C0, C1);
    const int CKP = ckp_for(g.CK);
    const int n_chunks = pad_to(C0, g.CK) / g.CK + (C1 ? pad_to(C1, g.CK) / g.CK : 0);
    const double M = (double)B * Ho * Wo;
    const double K = (double)KH * KW * (C0 + C1);
    bool chosen = false;
    g.fused_norm = 0;
    if (want_norm && Cout <= 256) {
        g.WN = Cout <= 64 ? 1 : (Cout <= 128 ? 2 : 4);
        g.WM = 4 / g.WN;
        fill_tile(g, B, Ho, Wo, Cout, KH, KW, stride);
        if (wgs_of(g) * grid_z >= min_fused_wgs) {
            chosen = true;
            g.fused_norm = 1;
        }
    }
    if (!chosen) {
        // rough cost: padded MFMA work + L2 traffic (activations re-read per N tile, weights per M tile)
        double best = 1e300;
        int best_wn = 1;
        for (int wn = 1; wn <= 4; wn *= 2) {
            if (wn > 1 && Cout <= 64 * (wn / 2)) break;
            ConvGeom t = g;
            t.WN = wn;
            t.WM = 4 / wn;
            fill_tile(t, B, Ho, Wo, Cout, KH, KW, stride);
            double m_tiles = (double)t.tiles_x * t.tiles_y * t.groups;
            double flops = 2.0 * m_tiles * (64 * t.WM) * t.n_tiles_n * (64 * wn) * K;
            double bytes = 4.0 * (M * stride * stride * (C0 + C1) * t.n_tiles_n + K * Cout * m_tiles);
            double cost = flops / 100e12 + bytes / 4e12;
            if (cost < best) {
                best = cost;
                best_wn = wn;
            }
        }
        g.WN = best_wn;
        g.WM = 4 / best_wn;
        fill_tile(g, B, Ho, Wo, Cout, KH, KW, stride);
    }
    // Tiny images with a wide kernel (3x3 on 1x1 / 2x2 maps) blow the staged window up to KH * KW times the pixel tile: when
    // a 256-pixel tile's window does not fit LDS, narrower pixel tiles (more waves along the couts, even if that pads
    // them) are taken instead of failing.
    auto window_fits = [&](const ConvGeom& t) {
        const size_t halo = g.CK == 16 ? (size_t)pad_to(t.NB * t.IH * t.IW, 64) * CKP : (size_t)pad_to(t.NB * t.IH * t.IW * CKP, 4);
        const size_t slab = (size_t)(t.WN == 1 ? KW : 1) * 64 * t.WN * CKP;  // TPS as chosen below
        return (halo + 2 * slab + 64 * t.WM + 256) * 4 <= 160 * 1024;
    };
    while (!window_fits(g) && g.WN < 4) {
        g.WN *= 2;
        g.WM = 4 / g.WN;
        g.fused_norm = 0;  // (a fused RMSNorm needs all couts in one tile: the caller lands through the norm kernel then)
        fill_tile(g, B, Ho, Wo, Cout, KH, KW, stride);
    }
    // taps per weight slab: a whole kernel row when two buffers + the window leave room for 2 WGs/CU
    const int NT = 64 * g.WN;
    // CK == 16 stages the window in whole 256-thread passes of 64 pixel rows (unpredicated LDS stores)
    g.halo_floats = g.CK == 16 ? pad_to(g.NB * g.IH * g.IW, 64) * CKP : pad_to(g.NB * g.IH * g.IW * CKP, 4);
    g.TPS = KW;
    if (g.WN > 1 && (g.halo_floats + 2 * KW * NT * CKP) * 4 > 80 * 1024) g.TPS = 1;
    if (KW * NT * (g.CK / 4) > 256 * wregs_for(g.WN, g.CK)) g.TPS = 1;
    if (force_tps == 1) g.TPS = 1;
    if (force_tps == 3) g.TPS = KW;
    g.w_floats = g.TPS * NT * CKP;
    // split-K over channel chunks when the grid would leave CUs idle
    g.splits = 1;
    if (!g.fused_norm && allow_split) {
        int wgs = wgs_of(g) * grid_z;
        if (wgs < target_wgs && n_chunks >= 4) {
            int s = (target_wgs + wgs - 1) / wgs;
            s = std::min(s, std::min(max_splits, n_chunks / 2));
            g.splits = std::max(1, s);
        }
    }
    g.chunks_per_split = (n_chunks + g.splits - 1) / g.splits;
    g.splits = (n_chunks + g.chunks_per_split - 1) / g.chunks_per_split;
    return g;
}

// ---------------------------------------------------------------------------------------

// E consecutive k-values of one MFMA operand row/column (one ds_read_b128 or ds_read_b64)
template <int E>
struct MicroFrag {
    float v[E];
    __device__ __forceinline__ void load(const float* p) {
        if constexpr (E == 2) {
            float2 t = *reinterpret_cast<const float2*>(p);
            v[0] = t.x;
            v[1] = t.y;
        } else {
            f32x4 t = *reinterpret_cast<const f32x4*>(p);
            v[0] = t.x;
            v[1] = t.y;
            v[2] = t.z;
            v[3] = t.w;
        }
    }
};

// source pixel (b*Hs + sy)*Ws + sx behind window pixel hp of this tile, or -1 (padding / outside the batch).
// Chunk independent, so a thread computes it once per kernel for the window items it stages.
__device__ __forceinline__ int window_pixel(const ConvParams& p, int hp, int b0, int iy0, int ix0) {
    const ConvGeom& g = p.geo;
    const int img_pix = g.IH * g.IW;
    int nb = hp / img_pix;
    int rem = hp - nb * img_pix;
    int hy = rem / g.IW;
    int hx = rem - hy * g.IW;
    int b = b0 + nb;
    int iy = iy0 + hy, ix = ix0 + hx;
    if (b >= p.B || iy < 0 || iy >= p.Hin || ix < 0 || ix >= p.Win) return -1;
    const int Hs = p.up ? (p.Hin >> 1) : p.Hin;
    const int Ws = p.up ? (p.Win >> 1) : p.Win;
    int sy = p.up ? (iy >> 1) : iy;
    int sx = p.up ? (ix >> 1) : ix;
    if (p.s2d) return (b * 2 * p.Hin + 2 * iy) * (2 * p.Win) + 2 * ix;  // tap (0,0) of the 2x2 block; taps shift the base
    return (b * Hs + sy) * Ws + sx;
}

// Source of K chunk `chunk` (CK == 16 path): base pointer for channel 0 of the chunk and the channel count of
// that source.  s2d: chunk = cin_chunk * 4 + tap, and tap (ky, kx) shifts every window pixel by (ky, kx).
__device__ __forceinline__ const float* chunk_source(const ConvParams& p, int chunk, int* Cs) {
    if (p.s2d) {
        const int tap = chunk & 3;
        *Cs = p.C0;
        return p.in0 + (chunk >> 2) * 16 + (size_t)((tap >> 1) * 2 * p.Win + (tap & 1)) * p.C0;
    }
    const bool s1 = chunk >= p.chunks0;
    *Cs = s1 ? p.C1 : p.C0;
    return (s1 ? p.in1 : p.in0) + (s1 ? chunk - p.chunks0 : chunk) * 16;
}

// one f32x4 of the input window: source pixel `pix`, channel quad q of chunk `chunk`
template <int CK>
__device__ __forceinline__ f32x4 load_window_px(const ConvParams& p, int pix, int q, int chunk) {
    const bool src1 = chunk >= p.chunks0;
    const float* __restrict__ src = src1 ? p.in1 : p.in0;
    const int Cs = src1 ? p.C1 : p.C0;
    const int c = (src1 ? chunk - p.chunks0 : chunk) * CK + 4 * q;
    f32x4 v = make_f32x4(0.f, 0.f, 0.f, 0.f);
    if (pix >= 0 && c < Cs) {
        if (p.in_nchw) {
            const int HWs = (p.up ? (p.Hin >> 1) * (p.Win >> 1) : p.Hin * p.Win);
            const int b = pix / HWs;
            const int sp = pix - b * HWs;
            size_t o = ((size_t)b * Cs + c) * HWs + sp;
            v.x = src[o];
            if (c + 1 < Cs) v.y = src[o + HWs];
            if (c + 2 < Cs) v.z = src[o + 2 * (size_t)HWs];
            if (c + 3 < Cs) v.w = src[o + 3 * (size_t)HWs];
        } else {
            size_t o = (size_t)pix * Cs + c;
            if ((Cs & 3) == 0) {
                v = *reinterpret_cast<const f32x4*>(src + o);
            } else {
                v.x = src[o];
                if (c + 1 < Cs) v.y = src[o + 1];
                if (c + 2 < Cs) v.z = src[o + 2];
                if (c + 3 < Cs) v.w = src[o + 3];
            }
        }
    }
    return v;
}

template <int WM, int WN, int CK, int TPSC>  // TPSC: taps per weight slab at compile time (0 = run time)
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p) {
    constexpr int NT = WN * 64;
    constexpr int CKP = (CK == 4) ? 4 : CK + 4;
    constexpr int HALF = CK / 2;
    constexpr int QPP = CK / 4;  // f32x4 items per pixel / weight row
    constexpr int HREGS = hregs_for(WM);
    constexpr int WREGS = wregs_for(WN, CK);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = p.geo;
    float* halo = smem;
    float* wbuf0 = smem + g.halo_floats;
    float* wbuf1 = wbuf0 + g.w_floats;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // block -> (n_tile, tile_x, tile_y, group); blockIdx.y = K split
    int bid = blockIdx.x;
    const int n_tile = bid % g.n_tiles_n;
    bid /= g.n_tiles_n;
    const int tile_x = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int tile_y = bid % g.tiles_y;
    const int group = bid / g.tiles_y;
    const int x0 = tile_x * g.TW, y0 = tile_y * g.TH, b0 = group * g.NB;
    // folded upsample conv: blockIdx.z picks the output parity, which fixes the (asymmetric) padding
    const int par = p.fold ? blockIdx.z : 0;
    const int par_y = par >> 1, par_x = par & 1;
    const int pad_y = p.fold ? 1 - par_y : p.pad, pad_x = p.fold ? 1 - par_x : p.pad_w;
    const int ix0 = x0 * p.stride - pad_x, iy0 = y0 * p.stride - pad_y;
    DM_STAMP_DECL
    DM_STAMP(0);
    __builtin_amdgcn_s_setprio(3);  // lowered to 0 only around the MFMA stream (see the slab loop)
    const int split = blockIdx.y;
    const int cb = split * g.chunks_per_split;
    const int ce = min(cb + g.chunks_per_split, p.n_chunks);

    // per-lane LDS read bases (floats)
    int a_base[2], b_base[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int m = wm * 64 + r * 32 + l31;
        int tx = m & (g.TW - 1);
        int ty = (m >> g.lTW) & (g.TH - 1);
        int nb = m >> (g.lTW + g.lTH);
        a_base[r] = ((nb * g.IH + ty * p.stride) * g.IW + tx * p.stride) * CKP + lh * HALF;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) b_base[q] = (wn * 64 + q * 32 + l31) * CKP + lh * HALF;

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][q][e] = 0.f;

    const int halo_items = g.NB * g.IH * g.IW * QPP;
    const bool halo_in_regs = halo_items <= 256 * HREGS;
    const int TPS = TPSC ? TPSC : g.TPS;
    const int w_items = TPS * NT * QPP;
    const int kxg = p.KW / TPS;           // tap groups per kernel row
    const int spc = p.KH * kxg;           // slabs per chunk
    const int n_slabs = (ce - cb) * spc;
    const int CoutP = (p.Cout + 255) & ~255;  // packed rows are padded so every N tile reads in bounds
    const float* __restrict__ wtile = p.w + (size_t)par * p.fold_w_stride + (size_t)n_tile * NT * CK;

    f32x4 wreg[WREGS];
    f32x4 hreg[HREGS];

    // Weight slab (chunk, ky, kx0 .. kx0+TPS-1) -> registers -> LDS.  For CK == 16 a slab is a whole number
    // of 256-thread passes (TPS*WN of them) and the packed rows are padded to 256 couts, so the code is
    // branch-free apart from the uniform pass count: item i of a thread is tap i/WN, cout rows (i%WN)*64...
    const int n_w = TPS * WN;  // CK == 16 passes
    auto load_w = [&](int chunk, int ky, int kx0) {
        const float* __restrict__ src =
            wtile + ((((size_t)chunk * p.KH + ky) * p.KW + kx0) * (size_t)CoutP) * CK;
        if constexpr (CK == 16) {
#pragma unroll
            for (int i = 0; i < WREGS; ++i)
                if (i < n_w)
                    wreg[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(i / WN) * CoutP * CK +
                                                               (i % WN) * 1024 + tid * 4);
        } else {
#pragma unroll
            for (int i = 0; i < WREGS; ++i) {
                int it = tid + 256 * i;
                if (it < w_items) {
                    int t = it / (NT * QPP);
                    int rq = it - t * (NT * QPP);
                    wreg[i] = *reinterpret_cast<const f32x4*>(src + (size_t)t * CoutP * CK + rq * 4);
                }
            }
        }
    };
    auto store_w = [&](float* wb) {
        if constexpr (CK == 16) {
            float* dst = wb + (tid >> 2) * CKP + (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < WREGS; ++i)
                if (i < n_w) *reinterpret_cast<f32x4*>(dst + i * 64 * CKP) = wreg[i];
        } else {
#pragma unroll
            for (int i = 0; i < WREGS; ++i) {
                int it = tid + 256 * i;
                if (it < w_items) {
                    int row = it / QPP;  // t*NT + cout row
                    int q = it - row * QPP;
                    *reinterpret_cast<f32x4*>(wb + row * CKP + 4 * q) = wreg[i];
                }
            }
        }
    };
    int hpix[HREGS];  // source pixel of each window item this thread stages (chunk independent)
    if (halo_in_regs) {
#pragma unroll
        for (int i = 0; i < HREGS; ++i) {
            int it = tid + 256 * i;
            hpix[i] = it < halo_items ? window_pixel(p, it / QPP, b0, iy0, ix0) : -1;
        }
    }
    // global pixel index of every output row of the tile (or -1), shared by all waves in the epilogue
    int* ptab = reinterpret_cast<int*>(smem + g.ptab_off);
    if (tid < WM * 64) {
        int tx = tid & (g.TW - 1);
        int ty = (tid >> g.lTW) & (g.TH - 1);
        int nb = tid >> (g.lTW + g.lTH);
        int b = b0 + nb, y = y0 + ty, x = x0 + tx;
        const bool ok = b < p.B && y < p.Ho && x < p.Wo;
        ptab[tid] = !ok ? -1
                        : (p.fold ? (b * 2 * p.Ho + 2 * y + par_y) * (2 * p.Wo) + 2 * x + par_x
                                  : (b * p.Ho + y) * p.Wo + x);
    }
    // Input window of one channel chunk -> registers -> LDS.  CK == 16 (NHWC, C % 16 == 0): branch-free --
    // padding pixels load pixel 0 and are zeroed by a select; the LDS region is padded to whole passes.
    const int n_h = (halo_items + 255) >> 8;
    auto load_h = [&](int chunk) {
        if constexpr (CK == 16) {
            int Cs;
            const float* __restrict__ src = chunk_source(p, chunk, &Cs) + (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < HREGS; ++i)
                if (i < n_h)  // zeroing of padding pixels happens at store time: nothing here may touch the data,
                              // or the wave would wait for the load right after issuing it
                    hreg[i] = *reinterpret_cast<const f32x4*>(src + (size_t)max(hpix[i], 0) * Cs);
        } else {
#pragma unroll
            for (int i = 0; i < HREGS; ++i) {
                int it = tid + 256 * i;
                if (it < halo_items) hreg[i] = load_window_px<CK>(p, hpix[i], it & (QPP - 1), chunk);
            }
        }
    };
    auto store_h = [&]() {
        if constexpr (CK == 16) {
            float* dst = halo + (tid >> 2) * CKP + (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < HREGS; ++i)
                if (i < n_h)
                    *reinterpret_cast<f32x4*>(dst + i * 64 * CKP) =
                        hpix[i] >= 0 ? hreg[i] : make_f32x4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int i = 0; i < HREGS; ++i) {
                int it = tid + 256 * i;
                if (it < halo_items) {
                    int hp = it / QPP;
                    int q = it - hp * QPP;
                    *reinterpret_cast<f32x4*>(halo + hp * CKP + 4 * q) = hreg[i];
                }
            }
        }
    };
    auto stage_h_direct = [&](int chunk) {
        for (int it = tid; it < halo_items; it += 256) {
            int hp = it / QPP;
            int q = it - hp * QPP;
            f32x4 v = load_window_px<CK>(p, window_pixel(p, hp, b0, iy0, ix0), q, chunk);
            *reinterpret_cast<f32x4*>(halo + hp * CKP + 4 * q) = v;
        }
    };

    // ---- prologue: slab 0 ----
    load_w(cb, 0, 0);
    if (halo_in_regs) {
        load_h(cb);
        store_h();
    } else {
        stage_h_direct(cb);
    }
    store_w(wbuf0);
    __syncthreads();
    DM_STAMP_ADD(0)

    // slab counters (chunk, ky, tap group) advance incrementally: no integer division in the loop
    int chunk = cb, ky = 0, kgi = 0;
    if constexpr (CK == 16 && TPSC > 0) {
        // ---- main loop, taps per slab known at compile time: the staging work of the NEXT slab is issued from
        // hooks BETWEEN the MFMAs of this slab (one small piece per MFMA, order pinned by sched_barrier), so it
        // runs in the shadow of the 64-cycle matrix instructions instead of forming a phase of its own.  Stamps
        // showed that phase at ~3.4k cycles per 6.1k-cycle slab before.
        constexpr int E = 4, UPT = 2, NU = TPSC * UPT, NM = NU * E * 4;  // MFMAs per slab per wave
        constexpr int NW = TPSC * WN;                                      // weight items per thread and slab
        constexpr int H0 = NW + 2;                                         // first MFMA slot of the window loads
        static_assert(H0 + HREGS <= NM - NW, "staging hooks do not fit into one slab");
        // Consecutive slabs are consecutive in the packed weights ([chunk][ky][kx] order), so the source of the
        // next slab is one pointer bump; the window source changes only when the chunk does.
        const size_t slab_stride = (size_t)TPSC * CoutP * CK;
        const float* __restrict__ wnext = wtile + ((size_t)cb * spc + 1) * slab_stride + tid * 4;
        int sic = 0;  // slab index inside the chunk
        const float* __restrict__ hsrc = nullptr;
        int hCs = 0;
        auto window_source = [&](int nc) { hsrc = chunk_source(p, nc, &hCs) + (tid & 3) * 4; };
        window_source(cb + 1 < ce ? cb + 1 : cb);
        for (int slab = 0; slab < n_slabs; ++slab) {
            const bool has_next = slab + 1 < n_slabs;
            const bool chunk_ends = has_next && sic == spc - 1;
            const int nchunk = chunk_ends ? chunk + 1 : chunk;
            const bool ld_h = chunk_ends && halo_in_regs;
            // at the last slab the weight loads re-read the current slab (valid memory; never stored)
            const float* __restrict__ wsrc = has_next ? wnext : wnext - slab_stride;
            float* wdst = ((slab & 1) ? wbuf0 : wbuf1) + (tid >> 2) * CKP + (tid & 3) * 4;
            const float* wb = (slab & 1) ? wbuf1 : wbuf0;
            const int a_row = (ky * g.IW + kgi * TPSC) * CKP;
            DM_STAMP_ADD(1)
            MicroFrag<E> fa[2][2], fb[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) fa[0][r].load(halo + a_base[r] + a_row);
#pragma unroll
            for (int q = 0; q < 2; ++q) fb[0][q].load(wb + b_base[q]);
            __builtin_amdgcn_s_setprio(0);  // the MFMA stream yields to the other wave's bookkeeping code
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 1 < NU) {
                    const int t = (u + 1) / UPT, hh = (u + 1) % UPT;
#pragma unroll
                    for (int r = 0; r < 2; ++r) fa[(u + 1) & 1][r].load(halo + a_base[r] + a_row + t * CKP + hh * E);
#pragma unroll
                    for (int q = 0; q < 2; ++q) fb[(u + 1) & 1][q].load(wb + b_base[q] + t * NT * CKP + hh * E);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < E; ++s)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int m = ((u * E + s) * 2 + r) * 2 + q;  // index of this MFMA within the slab
                            acc[r][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u & 1][r].v[s], fb[u & 1][q].v[s],
                                                                             acc[r][q], 0, 0, 0);
                            if (m < NW) {  // next slab's weights: one 16-byte load per hook
                                wreg[m] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)(m / WN) * CoutP * CK +
                                                                          (m % WN) * 1024);
                            } else if (m >= H0 && m < H0 + HREGS) {  // next chunk's input window
                                if (ld_h && m - H0 < n_h)
                                    hreg[m - H0] = *reinterpret_cast<const f32x4*>(
                                        hsrc + (size_t)max(hpix[m - H0], 0) * hCs);
                            }
                            if (m >= NM - NW) {  // weights -> the other LDS buffer (free since the last barrier)
                                if (has_next) *reinterpret_cast<f32x4*>(wdst + (m - (NM - NW)) * 64 * CKP) = wreg[m - (NM - NW)];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
            }
            __builtin_amdgcn_s_setprio(3);
            DM_STAMP_ADD(2)
            DM_STAMP_ADD(3)
            __syncthreads();  // slab done everywhere: its weight buffer and (at a chunk end) the window are free
            DM_STAMP_ADD(4)
            if (chunk_ends) {
                if (halo_in_regs) store_h();
                else stage_h_direct(nchunk);
                __syncthreads();
            }
            DM_STAMP_ADD(5)
            wnext += slab_stride;
            if (++kgi == kxg) {
                kgi = 0;
                ++ky;
            }
            if (++sic == spc) {
                sic = 0;
                ky = 0;
                ++chunk;
                window_source(chunk + 1 < ce ? chunk + 1 : chunk);
            }
        }
    } else
    for (int slab = 0; slab < n_slabs; ++slab) {
        const bool has_next = slab + 1 < n_slabs;
        int nchunk = chunk, nky = ky, nkgi = kgi + 1;
        if (nkgi == kxg) {
            nkgi = 0;
            if (++nky == p.KH) {
                nky = 0;
                ++nchunk;
            }
        }
        const bool chunk_ends = has_next && nchunk != chunk;
        const int next_chunk = nchunk;
        if (has_next) load_w(nchunk, nky, nkgi * TPS);
        if (chunk_ends && halo_in_regs) load_h(next_chunk);
        DM_STAMP_ADD(1)
        // The staging code of this wave competes for the SIMD's issue port with the MFMA stream of the
        // co-resident workgroup's wave and loses about one slot per MFMA at equal priority (stamps: ~5k
        // cycles for ~150 instructions).  Run everything except the MFMA stream at raised priority.
        __builtin_amdgcn_s_setprio(0);

        const int kx0 = kgi * TPS;
        const float* wb = (slab & 1) ? wbuf1 : wbuf0;
        // Micro-step = E consecutive channels of one tap for this lane half (E*4 MFMAs).  Fragment reads of
        // micro-step u+1 are issued before the MFMAs of micro-step u (ping-pong registers), so the LDS
        // latency hides under matrix work instead of stalling every tap.
        constexpr int E = (CK == 16) ? 4 : 2;
        constexpr int UPT = HALF / E;  // micro-steps per tap
        const int nu = TPS * UPT;
        const int a_row = (ky * g.IW + kx0) * CKP;
        auto read_frags = [&](int u, MicroFrag<E> (&fa)[2], MicroFrag<E> (&fb)[2]) {
            const int t = u / UPT, hh = u - t * UPT;
            const int a_off = a_row + t * CKP + hh * E;
            const int b_off = t * NT * CKP + hh * E;
#pragma unroll
            for (int r = 0; r < 2; ++r) fa[r].load(halo + a_base[r] + a_off);
#pragma unroll
            for (int q = 0; q < 2; ++q) fb[q].load(wb + b_base[q] + b_off);
        };
        auto mfma_block = [&](const MicroFrag<E> (&fa)[2], const MicroFrag<E> (&fb)[2]) {
#pragma unroll
            for (int s = 0; s < E; ++s)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[r][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[r].v[s], fb[q].v[s], acc[r][q], 0, 0, 0);
        };
        MicroFrag<E> fa0[2], fb0[2], fa1[2], fb1[2];
        read_frags(0, fa0, fb0);
        // reads past the end are clamped to the last micro-step (a harmless re-read) so that the loop body
        // has no branch between a ds_read and its MFMAs: the compiler can then wait with a COUNTED lgkmcnt
        // sched_barrier(0) pins the order [reads of u+1][MFMAs of u]: without it the scheduler sinks the
        // reads to the end of the MFMA block to save registers and the latency is exposed again.
        for (int u = 0; u < nu; u += 2) {
            read_frags(min(u + 1, nu - 1), fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (UPT % 2 == 0) {
                read_frags(min(u + 2, nu - 1), fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_block(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            } else if (u + 1 < nu) {
                read_frags(min(u + 2, nu - 1), fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_block(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(3);
        DM_STAMP_ADD(2)
        if (has_next) store_w((slab & 1) ? wbuf0 : wbuf1);
        DM_STAMP_ADD(3)
        __syncthreads();  // slab done everywhere: its weight buffer and (at a chunk end) the window are free
        DM_STAMP_ADD(4)
        if (chunk_ends) {
            if (halo_in_regs) store_h();
            else stage_h_direct(next_chunk);
            __syncthreads();
        }
        DM_STAMP_ADD(5)
        chunk = nchunk;
        ky = nky;
        kgi = nkgi;
    }

    // ------------------------------- epilogue -------------------------------
    const int epi = p.epi;
    {
        // Row-layout epilogue (each wave owns 64 pixels x 64 consecutive couts): the accumulators go
        // through LDS once so that a pixel's 64 couts sit on 16 consecutive lanes as float4.  Every global
        // access of the epilogue is then a 16-byte one covering whole 256-byte pixel rows (16 stores per lane
        // instead of 64 dword stores), the RMSNorm reduction stays inside a 16-lane DPP row, and the
        // per-cout vectors (bias, g, scale, shift) are one float4 per lane.
        if (!p.out_nchw && (p.Cout & 3) == 0) {
            constexpr int TS = 68;  // padded row stride (floats): conflict-free b32 writes and b128 reads
            const int rsub = lane >> 4;          // row within a group of 4
            const int c4 = (lane & 15) * 4;      // first of this lane's 4 couts inside the tile
            const int cg = n_tile * NT + wn * 64 + c4;  // global cout
            const bool cvalid = cg < p.Cout;
            int pixv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) pixv[j] = ptab[wm * 64 + 4 * j + rsub];
            RowsEpilogue re;
            re.split = split;
            re.M = (size_t)p.B * p.Ho * p.Wo * (p.fold ? 4 : 1);
            re.b0 = b0;
            re.uni = g.NB == 1 || p.ss_stride == 0;
            re.HoWo = p.Ho * p.Wo * (p.fold ? 4 : 1);
            re.red = smem + g.ptab_off + 64 * WM;  // [WN][64*WM]
            re.rows_per_wg = 64 * WM;
            re.row_in_wg0 = wm * 64;
            re.wn = wn;
            re.all_valid = (p.Cout % NT) == 0;
            // everything the epilogue reads from global memory -- the residual rows included (res_conv and the
            // attention projections add one: 64 KB per workgroup whose latency was exposed row by row, 16 k of the
            // 85 k cycles of a 128->64 res_conv workgroup) -- is requested before the accumulators go through LDS
            RowsPrefetch<16, true> pf;
            rows_prefetch<16, true>(p, re, pixv, cg, cvalid, pf);
            float* T = smem + wave * (64 * TS);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        T[(r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * TS + q * 32 + l31] = acc[r][q][e];
            __builtin_amdgcn_wave_barrier();
            f32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f32x4*>(T + (4 * j + rsub) * TS + c4);
            rows_epilogue<WN, 16, true>(p, re, v, pixv, cg, cvalid, pf);
            DM_STAMP_ADD(6)
            DM_STAMP_FLUSH
            return;
        }
    }

    int co[2];
    bool cok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        co[q] = n_tile * NT + wn * 64 + q * 32 + l31;
        cok[q] = co[q] < p.Cout;
    }

    if (p.partial) {
        // raw partial sums of this K split: out[split][pixel][cout]
        const size_t M = (size_t)p.B * p.Ho * p.Wo * (p.fold ? 4 : 1);
        float* po = p.out + (size_t)split * M * p.Cout;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pixi = ptab[wm * 64 + r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh];
                if (pixi < 0) continue;
                const size_t pix = (size_t)pixi;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (cok[q]) po[pix * p.Cout + co[q]] = acc[r][q][e];
            }
        DM_STAMP_ADD(6)
        DM_STAMP_FLUSH
        return;
    }

    float bias[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if ((epi & EPI_BIAS) && cok[q]) bias[q] = p.bias[co[q]];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][q][e] = cok[q] ? acc[r][q][e] + bias[q] : 0.f;

    if (epi & EPI_NORM) {
        // sum of squares over all couts of each pixel: lanes (32) x col tiles (2) x waves along N (WN)
        float ss[2][16];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[r][0][e] * acc[r][0][e] + acc[r][1][e] * acc[r][1][e];
                ss[r][e] = half_wave_sum(v);
            }
        if constexpr (WN > 1) {
            // the loop ended with a barrier, so LDS is free: reuse it as [WN][MT] scratch
            float* red = smem;
            constexpr int MT = WM * 64;
            if (l31 == 0) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        int m = wm * 64 + r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        red[wn * MT + m] = ss[r][e];
                    }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int m = wm * 64 + r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) t += red[w * MT + m];
                    ss[r][e] = t;
                }
        }
        const float sqrtc = sqrtf((float)p.Cout);
        float gq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) gq[q] = cok[q] ? p.g[co[q]] * sqrtc : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                // 1 / max(||v||, 1e-12) as one v_rsq_f32 (1 ulp)
                float rn = fast_rsq(fmaxf(ss[r][e], 1e-24f));
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[r][q][e] = acc[r][q][e] * rn * gq[q];
            }
    }

    // scale/shift are per (image, cout): one pair per lane when the tile holds one image or t is shared
    const bool ss_uniform = (epi & EPI_SCALE_SHIFT) && (g.NB == 1 || p.ss_stride == 0);
    float sc1[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
    if (ss_uniform) {
        const float* sp = p.scale + (size_t)min(b0, p.B - 1) * p.ss_stride;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (cok[q]) {
                sc1[q] = sp[co[q]] + 1.0f;
                sh[q] = sp[p.Cout + co[q]];
            }
    }
    const int HoWo = p.Ho * p.Wo * (p.fold ? 4 : 1);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int pixi = ptab[wm * 64 + r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh];
            if (pixi < 0) continue;
            const size_t pix = (size_t)pixi;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!cok[q]) continue;
                float v = acc[r][q][e];
                if (epi & EPI_SCALE_SHIFT) {
                    if (ss_uniform) {
                        v = v * sc1[q] + sh[q];
                    } else {
                        const float* sp = p.scale + (size_t)(pixi / HoWo) * p.ss_stride;
                        v = v * (sp[co[q]] + 1.0f) + sp[p.Cout + co[q]];
                    }
                }
                if (epi & EPI_SILU) v = v * fast_rcp(1.0f + __expf(-v));
                if (epi & EPI_RELU) v = fmaxf(v, 0.0f);
                if (epi & EPI_RESIDUAL) v += p.residual[pix * p.Cout + co[q]];
                if (p.out_nchw) {
                    const int b = pixi / HoWo;
                    p.out[((size_t)b * p.Cout + co[q]) * HoWo + (pixi - b * HoWo)] = v;
                } else {
                    p.out[pix * p.Cout + co[q]] = v;
                }
            }
        }
    DM_STAMP_ADD(6)
    DM_STAMP_FLUSH
}

template <int WM, int WN, int CK, int TPSC>  // TPSC: taps per weight slab at compile time (0 = run time)
static int launch_one(const ConvParams& p, hipStream_t s) {
    static LdsOptIn lds_flag;
    auto kern = conv_mfma_kernel<WM, WN, CK, TPSC>;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(kern), 1)) return 1;
    const ConvGeom& g = p.geo;
    DM_REQUIRE(g.TPS * (WN * 64) * (CK / 4) <= 256 * wregs_for(WN, CK), "conv: weight slab exceeds staging registers");
    int blocks = g.n_tiles_n * g.tiles_x * g.tiles_y * g.groups;
    const bool timed = prof::enabled();
    if (timed) {
        // algorithmic work of this launch: 2*k*k*Cin*Cout*pixels FLOP;
        // read input once + write output once + weights once
        // (a folded upsample conv is priced as the reference's op: 9 taps at the output resolution)
        const double pix = (double)p.B * p.Ho * p.Wo * (p.fold ? 4 : 1);
        const double cin = p.C0 + p.C1;
        const double in_pix = (double)p.B * (p.up ? (p.Hin / 2) * (p.Win / 2) : p.Hin * p.Win) * (p.s2d ? 4 : 1);
        const double taps = p.fold ? 9.0 : (p.s2d ? 4.0 : (double)p.KH * p.KW);
        const double flops = 2.0 * taps * cin * p.Cout * pix;
        const double res_rows = (!p.partial && (p.epi & EPI_RESIDUAL)) ? 1.0 : 0.0;  // the fused residual add reads one more tensor
        const double bytes = 4.0 * (cin * in_pix + (1.0 + res_rows) * p.Cout * pix + taps * cin * p.Cout);
        char name[64];
        if (prof::detail())
            snprintf(name, sizeof(name), "conv<%d,%d,%d> %dx%d s%d %d+%d->%d @%dx%d%s e%d k%d t%d", WM, WN, CK, p.KH,
                     p.KW, p.stride, p.C0, p.C1, p.Cout, p.Ho, p.Wo, p.fold ? " upfold" : (p.up ? " up" : (p.s2d ? " s2d" : "")), p.epi, g.splits, g.TPS);
        else
            snprintf(name, sizeof(name), "conv_mfma_kernel<%d,%d,%d,%d>", WM, WN, CK, TPSC);
        if (prof::begin(name, flops, bytes, s)) return 1;
    }
#ifdef DM_STAMPS
    // diagnostic build: run the launch synchronously with a stamp buffer and print the phase averages
    {
        const size_t nblk = (size_t)blocks * g.splits;
        unsigned long long* dbuf = nullptr;
        DM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), nblk * 8 * sizeof(unsigned long long)));
        DM_CHECK_HIP(hipMemsetAsync(dbuf, 0, nblk * 8 * sizeof(unsigned long long), s));
        ConvParams ps = p;
        ps.stamps = dbuf;
        hipLaunchKernelGGL(kern, dim3(blocks, g.splits, p.fold ? 4 : 1), dim3(256), g.lds_bytes, s, ps);
        DM_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nblk * 8);
        DM_CHECK_HIP(hipMemcpy(h.data(), dbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(dbuf);
        double avg[8] = {0};
        for (size_t b = 0; b < nblk; ++b)
            for (int k = 0; k < 8; ++k) avg[k] += (double)h[b * 8 + k] / nblk;
        double tot = 0;
        for (int k = 0; k < 7; ++k) tot += avg[k];
        fprintf(stderr,
                "STAMPS conv<%d,%d,%d> %dx%d %d+%d->%d @%dx%d e%d k%d: wgs=%zu total=%.0f | prologue %.0f issue %.0f "
                "mfma %.0f storew %.0f barrier %.0f halo %.0f epilogue %.0f\n",
                WM, WN, CK, p.KH, p.KW, p.C0, p.C1, p.Cout, p.Ho, p.Wo, p.epi, g.splits, nblk, tot, avg[0], avg[1],
                avg[2], avg[3], avg[4], avg[5], avg[6]);
        return 0;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(blocks, g.splits, p.fold ? 4 : 1), dim3(256), g.lds_bytes, s, p);
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    return 0;
}

int conv_launch(const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    p.stamps = nullptr;
    ConvGeom& g = p.geo;
    const int CK = g.CK;
    DM_REQUIRE(g.WM * g.WN == 4, "conv: bad wave grid");
    DM_REQUIRE(CK == 16 || CK == 4, "conv: bad CK");
    DM_REQUIRE(!(p.epi & EPI_NORM) || (g.n_tiles_n == 1 && g.splits == 1), "conv: fused RMSNorm needs one N tile");
    DM_REQUIRE(g.splits == 1 || p.partial, "conv: split-K writes partial sums");
    DM_REQUIRE(!p.partial || !p.out_nchw, "conv: partial sums are NHWC");
    DM_REQUIRE(!p.in_nchw || p.C1 == 0, "conv: NCHW input supports one source");
    DM_REQUIRE(!p.up || ((p.Hin % 2 == 0) && (p.Win % 2 == 0)), "conv: upsampled input must be even");
    DM_REQUIRE(!p.s2d || (CK == 16 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.C1 == 0 && !p.up &&
                          !p.fold && !p.in_nchw && p.C0 % 16 == 0 && p.n_chunks == 4 * (p.C0 / 16)),
               "conv: space-to-depth mode");
    DM_REQUIRE(p.KW % g.TPS == 0, "conv: taps per slab must divide KW");
    {
        // the row epilogue addresses the output with 32-bit byte offsets
        const size_t M = (size_t)p.B * p.Ho * p.Wo * (p.fold ? 4 : 1);
        DM_REQUIRE(M * (size_t)p.Cout < (1ull << 30), "conv: output tensor too large (2^30 elements)");
    }
    if (CK == 16) DM_REQUIRE(p.C0 % 4 == 0 && p.C1 % 4 == 0, "conv: CK16 needs C % 4 == 0");
    int red_floats = (p.epi & EPI_NORM) && g.WN > 1 ? g.WN * g.WM * 64 : 0;
    // the pixel table sits behind the staging buffers and behind the 4 x [64][68] transposed tiles of the
    // row-layout epilogue (WN == 1), which reuse the staging space after the last barrier
    g.ptab_off = std::max(g.halo_floats + 2 * g.w_floats, 4 * 64 * 68);
    // [ staging | 4 transposed tiles ] [ pixel table: 64*WM ] [ cross-wave norm partials: WN * 64*WM = 256 ]
    g.lds_bytes = std::max(g.ptab_off + 64 * g.WM + 256, red_floats) * 4;
    DM_REQUIRE(g.lds_bytes <= 160 * 1024, "conv: tile does not fit LDS");
    if (CK == 16) {
        // taps per slab is a template parameter on this path (hooked main loop)
#define DM_CONV_DISPATCH(WM_, WN_)                                      \
    switch (g.TPS) {                                                    \
        case 1: return launch_one<WM_, WN_, 16, 1>(p, s);               \
        case 2: return launch_one<WM_, WN_, 16, 2>(p, s);               \
        case 3: return launch_one<WM_, WN_, 16, 3>(p, s);               \
        default: return launch_one<WM_, WN_, 16, 0>(p, s);              \
    }
        if (g.WN == 1) { DM_CONV_DISPATCH(4, 1) }
        if (g.WN == 2) { DM_CONV_DISPATCH(2, 2) }
        DM_CONV_DISPATCH(1, 4)
#undef DM_CONV_DISPATCH
    } else {
        if (g.WN == 1) return launch_one<4, 1, 4, 0>(p, s);
        if (g.WN == 2) return launch_one<2, 2, 4, 0>(p, s);
        return launch_one<1, 4, 4, 0>(p, s);
    }
}

}  // namespace dm

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
// channel chunk and re-used by all KH*KW taps; the weight slab of one kernel row
// ([KW][NT][CK+4]) is staged per (chunk, ky).
#include "dm_common.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

namespace dm {

using f32x16 = __attribute__((ext_vector_type(16))) float;

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

int conv_nt_for(int Cout, bool) {
    if (Cout <= 64) return 64;
    if (Cout <= 128) return 128;
    if (Cout % 256 == 0) return 256;
    if (Cout % 128 == 0) return 128;
    return 256;
}

int conv_ck_for(int C0, int C1) { return (C0 % 16 == 0 && C1 % 16 == 0) ? 16 : 4; }

static inline int ckp_for(int CK) { return CK == 4 ? 4 : CK + 4; }
static inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

ConvGeom conv_plan(int B, int Ho, int Wo, int Cout, int KH, int KW, int stride, int Cin_total, bool want_norm) {
    ConvGeom g{};
    (void)Cin_total;
    int NT = conv_nt_for(Cout, want_norm);
    g.WN = NT / 64;
    g.WM = 4 / g.WN;
    int MT = 64 * g.WM;
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
    return g;
}

size_t conv_packed_floats(int Cout, int C0, int C1, int KH, int KW, bool want_norm) {
    int NT = conv_nt_for(Cout, want_norm);
    int CK = conv_ck_for(C0, C1);
    int n_tiles = (Cout + NT - 1) / NT;
    int chunks = pad_to(C0, CK) / CK + (C1 ? pad_to(C1, CK) / CK : 0);
    return (size_t)n_tiles * chunks * KH * KW * NT * CK;
}

void conv_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1, int KH, int KW, bool want_norm) {
    int NT = conv_nt_for(Cout, want_norm);
    int CK = conv_ck_for(C0, C1);
    int n_tiles = (Cout + NT - 1) / NT;
    int chunks0 = pad_to(C0, CK) / CK;
    int chunks1 = C1 ? pad_to(C1, CK) / CK : 0;
    int chunks = chunks0 + chunks1;
    int Cin = C0 + C1;
    size_t total = (size_t)n_tiles * chunks * KH * KW * NT * CK;
    std::memset(packed, 0, total * sizeof(float));
    for (int nt = 0; nt < n_tiles; ++nt)
        for (int ch = 0; ch < chunks; ++ch)
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx) {
                    float* dst = packed + ((((size_t)nt * chunks + ch) * KH + ky) * KW + kx) * NT * CK;
                    for (int j = 0; j < NT; ++j) {
                        int co = nt * NT + j;
                        if (co >= Cout) continue;
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
                            dst[j * CK + kk] = oihw[(((size_t)co * Cin + cin) * KH + ky) * KW + kx];
                        }
                    }
                }
}

// ---------------------------------------------------------------------------------------

template <int CK>
struct Frag {
    float v[CK / 2];
};

template <int CK>
__device__ __forceinline__ void lds_read_frag(const float* p, Frag<CK>& f) {
    if constexpr (CK == 4) {
        float2 t = *reinterpret_cast<const float2*>(p);
        f.v[0] = t.x;
        f.v[1] = t.y;
    } else {
#pragma unroll
        for (int q = 0; q < CK / 8; ++q) {
            float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
            f.v[4 * q + 0] = t.x;
            f.v[4 * q + 1] = t.y;
            f.v[4 * q + 2] = t.z;
            f.v[4 * q + 3] = t.w;
        }
    }
}

__device__ __forceinline__ float half_wave_sum(float v) {
    // sum over the 32 lanes that share lane>>5 (xor masks stay inside a 32-lane half)
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

template <int WM, int WN, int CK>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p) {
    constexpr int NT = WN * 64;
    constexpr int CKP = (CK == 4) ? 4 : CK + 4;
    constexpr int HALF = CK / 2;
    constexpr int QPP = CK / 4;  // float4 items per pixel / weight row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = p.geo;
    float* halo = smem;
    float* wl = smem + g.halo_floats;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // block -> (n_tile, tile_x, tile_y, group)
    int bid = blockIdx.x;
    const int n_tile = bid % g.n_tiles_n;
    bid /= g.n_tiles_n;
    const int tile_x = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int tile_y = bid % g.tiles_y;
    const int group = bid / g.tiles_y;
    const int x0 = tile_x * g.TW, y0 = tile_y * g.TH, b0 = group * g.NB;
    const int ix0 = x0 * p.stride - p.pad, iy0 = y0 * p.stride - p.pad;

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
    const int w_items = p.KW * NT * QPP;
    const int Hs = p.up ? (p.Hin >> 1) : p.Hin;
    const int Ws = p.up ? (p.Win >> 1) : p.Win;
    const int img_pix = g.IH * g.IW;

    for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
        const bool src1 = chunk >= p.chunks0;
        const float* __restrict__ src = src1 ? p.in1 : p.in0;
        const int Cs = src1 ? p.C1 : p.C0;
        const int cbase = (src1 ? chunk - p.chunks0 : chunk) * CK;
        __syncthreads();  // previous chunk's readers are done with halo and wl
        // ---- stage the input window of this channel chunk ----
        for (int it = tid; it < halo_items; it += 256) {
            int hp = it / QPP;
            int q = it - hp * QPP;
            int nb = hp / img_pix;
            int rem = hp - nb * img_pix;
            int hy = rem / g.IW;
            int hx = rem - hy * g.IW;
            int b = b0 + nb;
            int iy = iy0 + hy, ix = ix0 + hx;
            int c = cbase + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < p.B && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win && c < Cs) {
                int sy = p.up ? (iy >> 1) : iy;
                int sx = p.up ? (ix >> 1) : ix;
                if (p.in_nchw) {
                    size_t o = (((size_t)b * Cs + c) * Hs + sy) * Ws + sx;
                    size_t cs = (size_t)Hs * Ws;
                    v.x = src[o];
                    if (c + 1 < Cs) v.y = src[o + cs];
                    if (c + 2 < Cs) v.z = src[o + 2 * cs];
                    if (c + 3 < Cs) v.w = src[o + 3 * cs];
                } else {
                    size_t o = (((size_t)b * Hs + sy) * Ws + sx) * Cs + c;
                    if ((Cs & 3) == 0) {
                        v = *reinterpret_cast<const float4*>(src + o);
                    } else {
                        v.x = src[o];
                        if (c + 1 < Cs) v.y = src[o + 1];
                        if (c + 2 < Cs) v.z = src[o + 2];
                        if (c + 3 < Cs) v.w = src[o + 3];
                    }
                }
            }
            *reinterpret_cast<float4*>(halo + hp * CKP + 4 * q) = v;
        }
        for (int ky = 0; ky < p.KH; ++ky) {
            if (ky > 0) __syncthreads();  // readers of the previous weight slab are done
            // ---- stage the weight slab (chunk, ky): KW x NT rows of CK floats, contiguous in HBM ----
            const float* __restrict__ wsrc =
                p.w + ((((size_t)n_tile * p.n_chunks + chunk) * p.KH + ky) * p.KW) * (size_t)(NT * CK);
            for (int it = tid; it < w_items; it += 256) {
                int row = it / QPP;
                int q = it - row * QPP;
                float4 v = *reinterpret_cast<const float4*>(wsrc + (size_t)it * 4);
                *reinterpret_cast<float4*>(wl + row * CKP + 4 * q) = v;
            }
            __syncthreads();
            // ---- MFMA over the KW taps of this kernel row ----
            for (int kx = 0; kx < p.KW; ++kx) {
                const int a_off = (ky * g.IW + kx) * CKP;
                const int b_off = kx * NT * CKP;
                Frag<CK> fa[2], fb[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) lds_read_frag<CK>(halo + a_base[r] + a_off, fa[r]);
#pragma unroll
                for (int q = 0; q < 2; ++q) lds_read_frag<CK>(wl + b_base[q] + b_off, fb[q]);
#pragma unroll
                for (int s = 0; s < HALF; ++s)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            acc[r][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[r].v[s], fb[q].v[s], acc[r][q], 0, 0, 0);
            }
        }
    }

    // ------------------------------- epilogue -------------------------------
    const int epi = p.epi;
    int co[2];
    bool cok[2];
    float bias[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        co[q] = n_tile * NT + wn * 64 + q * 32 + l31;
        cok[q] = co[q] < p.Cout;
        if ((epi & EPI_BIAS) && cok[q]) bias[q] = p.bias[co[q]];
    }
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
            __syncthreads();  // all waves finished reading halo/wl; reuse smem as [WN][MT] scratch
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
                float rn = 1.0f / fmaxf(sqrtf(ss[r][e]), 1e-12f);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[r][q][e] = acc[r][q][e] * rn * gq[q];
            }
    }

#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int m = wm * 64 + r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            int tx = m & (g.TW - 1);
            int ty = (m >> g.lTW) & (g.TH - 1);
            int nb = m >> (g.lTW + g.lTH);
            int b = b0 + nb, y = y0 + ty, x = x0 + tx;
            if (b >= p.B || y >= p.Ho || x >= p.Wo) continue;
            size_t pix = ((size_t)b * p.Ho + y) * p.Wo + x;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!cok[q]) continue;
                float v = acc[r][q][e];
                if (epi & EPI_SCALE_SHIFT) {
                    const float* sp = p.scale + (size_t)b * p.ss_stride;
                    v = v * (sp[co[q]] + 1.0f) + sp[p.Cout + co[q]];
                }
                if (epi & EPI_SILU) v = v / (1.0f + __expf(-v));
                if (epi & EPI_RESIDUAL) v += p.residual[pix * p.Cout + co[q]];
                if (p.out_nchw)
                    p.out[(((size_t)b * p.Cout + co[q]) * p.Ho + y) * p.Wo + x] = v;
                else
                    p.out[pix * p.Cout + co[q]] = v;
            }
        }
}

template <int WM, int WN, int CK>
static int launch_one(const ConvParams& p, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_mfma_kernel<WM, WN, CK>;
    if (!attr_set) {
        DM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const ConvGeom& g = p.geo;
    int blocks = g.n_tiles_n * g.tiles_x * g.tiles_y * g.groups;
    const bool timed = prof::enabled();
    if (timed) {
        // algorithmic work of this launch: 2*k*k*Cin*Cout*pixels FLOP;
        // read input once + write output once + weights once
        const double pix = (double)p.B * p.Ho * p.Wo;
        const double cin = p.C0 + p.C1;
        const double in_pix = (double)p.B * (p.up ? (p.Hin / 2) * (p.Win / 2) : p.Hin * p.Win);
        const double flops = 2.0 * p.KH * p.KW * cin * p.Cout * pix;
        const double bytes = 4.0 * (cin * in_pix + p.Cout * pix + (double)p.KH * p.KW * cin * p.Cout);
        char name[64];
        snprintf(name, sizeof(name), "conv_mfma_kernel<%d,%d,%d>", WM, WN, CK);
        if (prof::begin(name, flops, bytes, s)) return 1;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), g.lds_bytes, s, p);
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    return 0;
}

int conv_launch(const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    ConvGeom& g = p.geo;
    const int CK = g.CK;
    const int CKP = ckp_for(CK);
    const int NT = g.WN * 64;
    DM_REQUIRE(g.WM * g.WN == 4, "conv: bad wave grid");
    DM_REQUIRE(CK == 16 || CK == 4, "conv: bad CK");
    DM_REQUIRE(!(p.epi & EPI_NORM) || g.n_tiles_n == 1, "conv: fused RMSNorm needs one N tile");
    DM_REQUIRE(!p.in_nchw || p.C1 == 0, "conv: NCHW input supports one source");
    DM_REQUIRE(!p.up || ((p.Hin % 2 == 0) && (p.Win % 2 == 0)), "conv: upsampled input must be even");
    if (CK == 16) DM_REQUIRE(p.C0 % 4 == 0 && p.C1 % 4 == 0, "conv: CK16 needs C % 4 == 0");
    g.halo_floats = pad_to(g.NB * g.IH * g.IW * CKP, 4);
    int w_floats = p.KW * NT * CKP;
    int red_floats = (p.epi & EPI_NORM) && g.WN > 1 ? g.WN * g.WM * 64 : 0;
    g.lds_bytes = std::max(g.halo_floats + w_floats, red_floats) * 4;
    DM_REQUIRE(g.lds_bytes <= 160 * 1024, "conv: tile does not fit LDS");
    if (CK == 16) {
        if (g.WN == 1) return launch_one<4, 1, 16>(p, s);
        if (g.WN == 2) return launch_one<2, 2, 16>(p, s);
        return launch_one<1, 4, 16>(p, s);
    } else {
        if (g.WN == 1) return launch_one<4, 1, 4>(p, s);
        if (g.WN == 2) return launch_one<2, 2, 4>(p, s);
        return launch_one<1, 4, 4>(p, s);
    }
}

}  // namespace dm

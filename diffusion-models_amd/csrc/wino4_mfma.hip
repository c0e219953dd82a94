// Winograd F(4x4, 3x3) convolution for gfx950 on the f32-input MFMA (v_mfma_f32_16x16x4_f32).
//
// The 3x3 / stride-1 / pad-1 convolutions of Block (DD/denoising_diffusion.py:108, inside ResnetBlock :136-148) and
// the plain 3x3 convs of the last Downsample / Upsample stage (:291, :303) on power-of-two images: 36 multiplies per
// 4x4 output pixels and (cin, cout) pair -- 2.25 per output against 4 for F(2x2,3x3) (winograd_mfma.hip) and 9 for the
// direct form:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d = 6x6 input patch, g = 3x3 filter, Y = 4x4 outputs
//     B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//     G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//     A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 throughout; measured against the reference's direct convolution the whole U-Net forward differs by 1.2e-6
// rel-L2 and a DDIM-50 loop by 1.3e-6 (tests hold 1e-4 / 1e-3).
//
// Mapping.  One workgroup = 4 waves = 16 tiles (256 output pixels) x 64 couts on v_mfma_f32_16x16x4_f32, one
// workgroup per CU (each wave owns a SIMD; 144 accumulator registers -- the compiler keeps MFMA accumulators in the
// 256 AGPRs at this occupancy, which rules out the 288 of a 32-tile block).  The 6x6 positions xi = (i, j) of the
// transformed patch split into four 3x3 blocks: wave g = 2*RT + CT owns rows {0,1,2} (RT = 0) or {3,4,5} (RT = 1) and
// columns {0,1,2} / {3,4,5} (CT).  Both halves of B^T have the same shape once written over the four middle inputs
// z0..z3 = x1..x4 and one extra input e (x0 for the low half, x5 for the high half):
//     S = z3 - k z1,  D = z2 - k z0:   "plus" = S + l D,   "minus" = S - l D,   "P" = pe e + p0 z0 + p1 z1 + p2 z2 + p3 z3
//     low  half (rows 1, 2, 0):  k = 4, l = 1, P = 4 e - 5 z1 + z3        high half (rows 3, 4, 5):  k = 1, l = 2, P = e + 4 z0 - 5 z2
// so every wave runs the SAME instruction stream with its own coefficient registers (SGPRs) and LDS offsets; slot
// 3a + b of a wave is (row type a, column type b) with types 0 = P, 1 = plus, 2 = minus, and the host packs
// U = G g G^T per wave and slot.  As in winograd_mfma.hip neither MFMA operand goes through LDS in transformed form:
//   * B operand: lane (cout n = lane & 15, channel pair kq = lane >> 4) holds U[slot][16 gq + n][2 kq + st] for the 4
//     cout groups gq and 2 K steps st: 8 floats = two 16-byte buffer loads, issued one chunk ahead right after the
//     MFMAs that used the registers.
//   * A operand: the raw input window (NB images x (4 TH + 2) x (4 TW + 2) pixels x 8 channels) is staged in LDS, double
//     buffered, one barrier per chunk; lane (tile = lane & 15, kq) reads 5 x 5 pixels of its patch as channel pairs
//     (conflict-free ds_read_b64: the 16-byte slots of a row are XOR-swizzled by tile row / image), forms the 3 x 5
//     row-transformed values and from them its nine V -- the MFMA A operands of the next chunk.  Reads, packed VALU
//     work, weight / window loads and LDS stores are issued from hooks between the 72 MFMAs of a chunk.
// Epilogue: per output column beta each wave reduces its three columns (R_a[beta], one value per row type), the twelve
// partial rows meet in LDS, Y[alpha][beta] = sum_i A^T[alpha][i] R_i[beta], then the shared Block epilogue
// (conv_device.h: bias / RMSNorm / scale-shift / SiLU / residual, or raw K-split partial sums).
#include "conv_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace dm {

static constexpr int W4CK = 8;      // input channels per K chunk
static constexpr int W4TILES = 16;  // 4x4-pixel tiles per workgroup
static constexpr int W4WTS = 68;    // row stride (floats) of the epilogue staging tiles: 4 rows = 16 banks, so the two kq of a 32-lane pass hit different banks


// geometry classes: LTW = log2(tiles across); whole images per workgroup below 16 pixels across
template <int LTW>
struct W4Geo {
    static constexpr int TW = 1 << LTW;
    static constexpr int TH = TW;
    static constexpr int NB = W4TILES / (TW * TH);
    static constexpr int IH = 4 * TH + 2, IW = 4 * TW + 2;
    static constexpr int RS = ((IW * W4CK + 63) / 64) * 64;  // floats per window row: whole LDS bank periods
    static constexpr int BUF = NB * IH * RS;                 // floats per window buffer
    static constexpr int HR = LTW == 2 ? 3 : 2;              // 16-byte staging items per thread
};
// XOR code of image nb on the 16-byte slot index: images of one ds_read pass land in different bank groups
__host__ __device__ constexpr int w4_nbcode(int ltw, int nb) {
    return ltw == 2 ? 0 : (ltw == 1 ? ((nb & 1) | ((nb & 2) << 1)) : (nb & 15));
}

bool wino4_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up) {
    static const bool off = std::getenv("DM_NO_WINO4") != nullptr || std::getenv("DM_NO_WINOGRAD") != nullptr;
    return !off && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !up && C0 > 0 && C0 % W4CK == 0 &&
           C1 % W4CK == 0 && Cout % 64 == 0;
}

size_t wino4_packed_floats(int Cout, int C0, int C1) { return (size_t)(C0 + C1) * 36 * Cout; }

// row / column index of type t (0 = P, 1 = plus, 2 = minus) in the low (h = 0) and high (h = 1) half
static inline int w4_index(int h, int t) {
    static const int lo[3] = {0, 1, 2}, hi[3] = {5, 3, 4};
    return h ? hi[t] : lo[t];
}

void wino4_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1) {
    static const double G[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int Cin = C0 + C1;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float* gk = oihw + ((size_t)co * Cin + ci) * 9;
            double Gg[6][3];
            for (int i = 0; i < 6; ++i)
                for (int b = 0; b < 3; ++b) Gg[i][b] = G[i][0] * gk[b] + G[i][1] * gk[3 + b] + G[i][2] * gk[6 + b];
            double U[6][6];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) U[i][j] = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
            const int chunk = ci / W4CK, cc = ci % W4CK;
            for (int g = 0; g < 4; ++g)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        const int i = w4_index(g >> 1, a), j = w4_index(g & 1, b);
                        // [chunk][wave][slot][cout tile of 64][kq 4][n 16][gq 4][st 2]: channel cc = 2 kq + st, cout = 16 gq + n
                        const int ct = co / 64, gq = (co % 64) / 16, n = co % 16, kq = cc / 2, st = cc % 2;
                        packed[(((((size_t)chunk * 4 + g) * 9 + a * 3 + b) * (Cout / 64) + ct) * 4 + kq) * 128 + n * 8 +
                               gq * 2 + st] = (float)U[i][j];
                    }
        }
}

// log2(tiles across) of the geometry class that takes an (Ho, Wo) image, or -1
static int w4_class(int Ho, int Wo) {
    if (Ho == 4 && Wo == 4) return 0;
    if (Ho == 8 && Wo == 8) return 1;
    if (Wo % 16 == 0 && Ho % 16 == 0 && Ho > 0 && Wo > 0) return 2;
    return -1;
}

ConvGeom wino4_plan(int B, int Ho, int Wo, int Cout, int C0, int C1, bool allow_split) {
    ConvGeom g{};
    const int ltw = w4_class(Ho, Wo);
    g.WM = 1;
    g.WN = 1;
    g.CK = W4CK;
    g.lTW = ltw < 0 ? 0 : ltw;
    g.TW = 1 << g.lTW;
    g.TH = g.TW;
    g.lTH = g.lTW;
    g.NB = W4TILES / (g.TW * g.TH);
    g.tiles_x = std::max(1, Wo / (4 * g.TW));
    g.tiles_y = std::max(1, Ho / (4 * g.TH));
    g.groups = (B + g.NB - 1) / g.NB;
    g.n_tiles_n = Cout / 64;
    g.IH = 4 * g.TH + 2;
    g.IW = 4 * g.TW + 2;
    g.row_stride = ((g.IW * W4CK + 63) / 64) * 64;
    g.halo_floats = g.NB * g.IH * g.row_stride;
    g.TPS = 3;
    const int n_chunks = (C0 + C1) / W4CK;
    const int wgs = g.tiles_x * g.tiles_y * g.groups * g.n_tiles_n;
    int splits = 1;
    if (allow_split) {
        static const int target = env_int("DM_WINO4_TARGET_WGS", 256);
        static const int min_chunks = env_int("DM_WINO4_MIN_CHUNKS", 8);
        while (wgs * splits < target && splits < 8 && n_chunks / (splits * 2) >= min_chunks) splits *= 2;
    }
    g.chunks_per_split = (n_chunks + splits - 1) / splits;
    g.splits = (n_chunks + g.chunks_per_split - 1) / g.chunks_per_split;
    g.fused_norm = g.n_tiles_n == 1 && g.splits == 1;
    g.w_floats = 0;
    // two window buffers + a scratch slot reachable from both (items outside the image are stored at buffer + 2 BUF);
    // the epilogue reuses the space for 2 x 12 partial rows of 16 tiles x 64 couts (two output columns per pass); behind
    // both: the output-pixel table [tile][beta][alpha]
    g.ptab_off = std::max(3 * g.halo_floats + 16, 2 * 12 * W4TILES * W4WTS);
    g.lds_bytes = (g.ptab_off + 16 * W4TILES) * 4;
    return g;
}

bool wino4_shape_ok(int B, int Ho, int Wo, int Cout, int C0, int C1) {
    if (w4_class(Ho, Wo) < 0) return false;
    const ConvGeom g = wino4_plan(B, Ho, Wo, Cout, C0, C1, true);
    // One 512-register workgroup per CU cannot hide its own prologue (first windows and weights from L2 / HBM) and
    // epilogue behind another workgroup, as the two-per-CU F(2x2) kernel does: this kernel wins where those fixed costs
    // are amortised over a long reduction (measured at B = 256, 32x32: 16 chunks 1.10-1.18x, 32-48 chunks 1.19-1.23x,
    // 8 chunks 1.0x) and the chip is filled.
    static const int min_wgs = env_int("DM_WINO4_MIN_WGS", 200);
    static const int min_k = env_int("DM_WINO4_MIN_K", 12);  // chunks of 8 input channels per workgroup
    const int wgs = g.tiles_x * g.tiles_y * g.groups * g.n_tiles_n * g.splits;
    return wgs >= min_wgs && g.chunks_per_split >= min_k && g.lds_bytes <= 160 * 1024 &&
           (size_t)B * Ho * Wo < (1u << 24) && (size_t)B * Ho * Wo * std::max(C0, C1) < (1ull << 30);
}

// packed math whose first factor is a wave-uniform coefficient pair held in SGPRs (one constant-bus operand)
__device__ __forceinline__ f32x2 pk_fma_s(f32x2 sa, f32x2 b, f32x2 c) {  // sa * b + c
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(sa), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_fms_s(f32x2 sa, f32x2 b, f32x2 c) {  // c - sa * b
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "s"(sa), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul_s(f32x2 sa, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "s"(sa), "v"(b));
    return d;
}

struct W4Coef {
    f32x2 k, l, pe, p0, p1, p2, p3;
};
__device__ __forceinline__ W4Coef w4_coef(int hi) {
    W4Coef c;
    const float k = hi ? 1.f : 4.f, l = hi ? 2.f : 1.f, pe = hi ? 1.f : 4.f;
    const float p0 = hi ? 4.f : 0.f, p1 = hi ? 0.f : -5.f, p2 = hi ? -5.f : 0.f, p3 = hi ? 0.f : 1.f;
    c.k = f32x2{k, k};
    c.l = f32x2{l, l};
    c.pe = f32x2{pe, pe};
    c.p0 = f32x2{p0, p0};
    c.p1 = f32x2{p1, p1};
    c.p2 = f32x2{p2, p2};
    c.p3 = f32x2{p3, p3};
    return c;
}
// one half of B^T on (e, z0..z3): out[0] = P, out[1] = plus, out[2] = minus
__device__ __forceinline__ void w4_half(const W4Coef& c, f32x2 e, f32x2 z0, f32x2 z1, f32x2 z2, f32x2 z3, f32x2& oP,
                                        f32x2& oPlus, f32x2& oMinus) {
    const f32x2 S = pk_fms_s(c.k, z1, z3);  // z3 - k z1
    const f32x2 D = pk_fms_s(c.k, z0, z2);  // z2 - k z0
    oPlus = pk_fma_s(c.l, D, S);
    oMinus = pk_fms_s(c.l, D, S);
    f32x2 t = pk_mul_s(c.pe, e);
    t = pk_fma_s(c.p0, z0, t);
    t = pk_fma_s(c.p1, z1, t);
    t = pk_fma_s(c.p2, z2, t);
    oP = pk_fma_s(c.p3, z3, t);
}

template <int LTW>
__global__ __launch_bounds__(256, 1) void wino4_mfma_kernel(const ConvParams p) {
    using G = W4Geo<LTW>;
    constexpr int TW = G::TW, TH = G::TH, NB = G::NB, IH = G::IH, IW = G::IW, RS = G::RS, BUF = G::BUF, HR = G::HR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = p.geo;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RT = wave >> 1, CT = wave & 1;
    const int l15 = lane & 15;  // MFMA row (tile) of the A operand / column (cout) of the B operand
    const int kq = lane >> 4;   // channel pair (2 kq, 2 kq + 1) of the chunk

    // Block order: the cout tile is the fastest index and blocks are dealt round-robin over the 8 XCDs, so with 2, 4 or 8
    // cout tiles an XCD always works on the same tile(s): its L2 holds that tile's weights for all pixel blocks (the
    // deep layers are weight-streaming: 37.7 MB of transformed weights against 4 MB of activations at 512 channels).
    int n_tile, bid;
    block_to_tile(g, blockIdx.x, gridDim.x, n_tile, bid);
    const int tile_x = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int tile_y = bid % g.tiles_y;
    const int group = bid / g.tiles_y;
    const int tx0 = tile_x * TW, ty0 = tile_y * TH, b0 = group * NB;  // in tiles
    const int ix0 = 4 * tx0 - 1, iy0 = 4 * ty0 - 1;                   // window origin in pixels
    const int split = blockIdx.y;
    const int cb = split * g.chunks_per_split;
    const int ce = min(cb + g.chunks_per_split, p.n_chunks);
    float* raw[2] = {smem, smem + BUF};
    DM_STAMP_DECL
    DM_STAMP(0);
    constexpr int SCRATCH = 2 * BUF;  // floats from the buffer base: items outside the image land here (either buffer)

    // ---- window staging: item = (pixel of the load region, channel quad)
    int hpix[HR], hoff[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int it = tid + 256 * i;
        const int hp = it >> 1, qd = it & 1;
        int nb, hy, hx;
        bool in_region;
        if constexpr (LTW == 2) {  // window incl. halo of one image
            hy = hp / IW;
            hx = hp - hy * IW;
            nb = 0;
            in_region = hp < IH * IW;
        } else {  // whole images: the halo ring is padding
            constexpr int PIX = 16 * TW * TH;  // pixels per image
            nb = hp / PIX;
            const int r = hp - nb * PIX;
            hy = r / (4 * TW) + 1;
            hx = r % (4 * TW) + 1;
            in_region = nb < NB;
        }
        const int b = b0 + nb, iy = iy0 + hy, ix = ix0 + hx;
        hpix[i] = 0;
        hoff[i] = SCRATCH;
        if (in_region && b < p.B && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
            const int slot = (2 * hx + qd) ^ ((hx >> 3) & 1) ^ (((hy >> 2) & 3) << 1) ^ w4_nbcode(LTW, nb);
            hpix[i] = (b * p.Hin + iy) * p.Win + ix;
            hoff[i] = (nb * IH + hy) * RS + 4 * slot;
        }
    }
    f32x4 hreg[HR];
    const size_t in_px = (size_t)p.B * p.Hin * p.Win;
    const __amdgpu_buffer_rsrc_t rs_in0 = make_rsrc(p.in0, in_px * p.C0 * 4);
    const __amdgpu_buffer_rsrc_t rs_in1 = make_rsrc(p.C1 ? p.in1 : p.in0, in_px * (p.C1 ? p.C1 : p.C0) * 4);
    const unsigned hq = 4 * (tid & 1);
    unsigned hvo[HR];
    auto window_offsets = [&](unsigned Cs) {
#pragma unroll
        for (int i = 0; i < HR; ++i) hvo[i] = (__umul24((unsigned)hpix[i], Cs) + hq) * 4;
    };
    auto window_value = [&](int chunk, int i) {
        const bool s1 = chunk >= p.chunks0;
        return bufload4(s1 ? rs_in1 : rs_in0, hvo[i], (unsigned)(s1 ? chunk - p.chunks0 : chunk) * (W4CK * 4));
    };

    // ---- transform addressing of this lane: tile l15, channel pair kq; byte offsets inside a window buffer of
    //      (row 4 ty [+ immediate row offsets], column col(cc)), for rows 0..3 (rhi = 0) and 4..5 (rhi = 1);
    //      cc = 0 is the extra column (0 or 5), cc = 1..4 are columns 1..4
    unsigned acol[5][2], aext[5];
    {
        const int tx = l15 & (TW - 1), ty = (l15 >> LTW) & (TH - 1), nb = l15 >> (2 * LTW);
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) {
            const int c = cc == 0 ? (CT ? 5 : 0) : cc;
            const int hx = 4 * tx + c;
#pragma unroll
            for (int rhi = 0; rhi < 2; ++rhi) {
                const int slot = (2 * hx + (kq >> 1)) ^ ((hx >> 3) & 1) ^ (((ty + rhi) & 3) << 1) ^ w4_nbcode(LTW, nb);
                acol[cc][rhi] = (unsigned)(((nb * IH + 4 * ty) * RS + 4 * slot + 2 * (kq & 1)) * 4);
            }
            aext[cc] = acol[cc][RT] + (unsigned)((RT ? 5 : 0) * RS * 4);  // extra row: 0 (low half) or 5 (high half)
        }
    }
    const char* sbytes = reinterpret_cast<const char*>(smem);
    auto rd2 = [&](unsigned byte_off) { return *reinterpret_cast<const f32x2*>(sbytes + byte_off); };
    const W4Coef cr = w4_coef(RT), cc_ = w4_coef(CT);

    // ---- weights: lane (n = l15, kq) loads its 8 floats [gq 4][st 2] of a slot as two 16-byte loads
    const size_t u_chunk = (size_t)36 * p.Cout * W4CK;  // floats per chunk
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (size_t)p.n_chunks * u_chunk * 4);
    const unsigned uvo = (unsigned)((kq * 128 + l15 * 8) * 4);
    const unsigned u_slot = (unsigned)p.Cout * W4CK * 4;  // bytes between consecutive slots
    const unsigned u_wave = (unsigned)(9 * wave) * u_slot + (unsigned)n_tile * (64 * W4CK * 4);
    f32x4 U[9][2];  // [slot][gq pair]: .xy = (gq even, st 0 / 1), .zw = (gq odd, st 0 / 1)
    auto load_u = [&](int chunk, int k) {
        const unsigned so = (unsigned)chunk * (unsigned)(u_chunk * 4) + u_wave + k * u_slot;
        U[k][0] = bufload4(rs_w, uvo, so);
        U[k][1] = bufload4(rs_w, uvo, so + 16);
    };

    f32x2 A[9];        // V of the current chunk: slot k, channels 2 kq (.x, K step 0) and 2 kq + 1 (.y, K step 1)
    f32x4 acc[9][4];   // [slot][cout group]; first written by the first chunk's MFMAs (C = 0)
    f32x2 T[3][5];     // row-transformed values of the next chunk: [row type][column cc]

    // row stage of column cc from window buffer `bf` (compile-time byte offsets)
    auto row_stage_read = [&](auto bf_tag, auto cc_tag, f32x2 (&d)[5]) {
        constexpr int bf = decltype(bf_tag)::value, cc = decltype(cc_tag)::value;
        constexpr unsigned base = bf * BUF * 4;
        d[0] = rd2(aext[cc] + base);
        d[1] = rd2(acol[cc][0] + base + 1 * RS * 4);
        d[2] = rd2(acol[cc][0] + base + 2 * RS * 4);
        d[3] = rd2(acol[cc][0] + base + 3 * RS * 4);
        d[4] = rd2(acol[cc][1] + base + 4 * RS * 4);
    };
    auto row_stage_math = [&](int cc, const f32x2 (&d)[5]) {
        w4_half(cr, d[0], d[1], d[2], d[3], d[4], T[0][cc], T[1][cc], T[2][cc]);
    };
    // column stage of row type a -> A[3a .. 3a+2]
    auto col_stage = [&](int a) {
        w4_half(cc_, T[a][0], T[a][1], T[a][2], T[a][3], T[a][4], A[3 * a + 0], A[3 * a + 1], A[3 * a + 2]);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;

    // ---- prologue: request chunks cb and cb + 1 and the weights of chunk cb first, do the LDS housekeeping while they fly
    f32x4 h2[HR];
    {
        const bool two = cb + 1 < ce;
        window_offsets(cb >= p.chunks0 ? p.C1 : p.C0);
#pragma unroll
        for (int i = 0; i < HR; ++i) hreg[i] = window_value(cb, i);
        const int c1 = two ? cb + 1 : cb;
        if (c1 == p.chunks0) window_offsets(p.C1);
#pragma unroll
        for (int i = 0; i < HR; ++i) h2[i] = window_value(c1, i);
#pragma unroll
        for (int k = 0; k < 9; ++k) load_u(cb, k);
    }
    // zero both window buffers once (padding pixels stay zero), build the output-pixel table
    {
        const f32x4 z4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < 2 * BUF / 4; i += 256) *reinterpret_cast<f32x4*>(smem + 4 * i) = z4;
        int* ptab = reinterpret_cast<int*>(smem + g.ptab_off);
        {
            const int i = tid;  // 16 tiles x 16 pixels
            const int t = i >> 4, beta = (i >> 2) & 3, alpha = i & 3;
            const int tx = t & (TW - 1), ty = (t >> LTW) & (TH - 1), nb = t >> (2 * LTW);
            const int b = b0 + nb, y = 4 * (ty0 + ty) + alpha, x = 4 * (tx0 + tx) + beta;
            ptab[i] = (b < p.B && y < p.Ho && x < p.Wo) ? (b * p.Ho + y) * p.Wo + x : -1;
        }
    }

    __syncthreads();  // zero fill done
    DM_STAMP_ADD(0)
#pragma unroll
    for (int i = 0; i < HR; ++i) *reinterpret_cast<f32x4*>(raw[0] + hoff[i]) = hreg[i];
#pragma unroll
    for (int i = 0; i < HR; ++i) *reinterpret_cast<f32x4*>(raw[1] + hoff[i]) = h2[i];
    __syncthreads();
    DM_STAMP_ADD(1)
    {
        f32x2 d[5];
        row_stage_read(I0{}, I0{}, d); row_stage_math(0, d);
        row_stage_read(I0{}, I1{}, d); row_stage_math(1, d);
        row_stage_read(I0{}, I2{}, d); row_stage_math(2, d);
        row_stage_read(I0{}, I3{}, d); row_stage_math(3, d);
        row_stage_read(I0{}, I4{}, d); row_stage_math(4, d);
#pragma unroll
        for (int a = 0; a < 3; ++a) col_stage(a);
    }
    __syncthreads();  // buffer 0 is overwritten with chunk cb + 2 by the first iteration
    DM_STAMP_ADD(2)

    // ---- main loop: 72 MFMAs per chunk and wave; everything else rides in the hooks between them
    const f32x4 zero4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    auto chunk_body = [&](int c, auto first_tag, auto par_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;  // (c - cb) & 1: chunk c was read from buffer PAR
        constexpr int BN = PAR ^ 1, BS = PAR;          // chunk c + 1 sits in BN; chunk c + 2 goes to BS
        using BNT = std::integral_constant<int, BN>;
        const bool has1 = c + 1 < ce, has2 = c + 2 < ce;
        const int cw = has2 ? c + 2 : c;     // window fetched now (c again at the end: stored, never read)
        const int cun = has1 ? c + 1 : c;    // chunk whose weights are fetched now (never past the packed weights)
        if (cw == p.chunks0 && p.C1 != p.C0) {
            window_offsets(p.C1);
            asm volatile("" ::: "memory");
        }
        f32x2 d[5];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int m = (k * 2 + st) * 4 + gq;  // 0..71
                    acc[k][gq] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[k][st], U[k][gq >> 1][2 * (gq & 1) + st],
                                                                      (FIRST && st == 0) ? zero4 : acc[k][gq], 0, 0, 0);
                    // window of chunk c + 2: loads early, stores late
                    if (m < HR) hreg[m] = window_value(cw, m);
                    if (m >= 60 && m < 60 + HR) *reinterpret_cast<f32x4*>(raw[BS] + hoff[m - 60]) = hreg[m - 60];
                    // weights of chunk c + 1 into the registers the MFMAs of slot k have just used
                    if (m == 8 * k + 7) load_u(cun, k);
                    // transform of chunk c + 1: row stage in slots 2..29, column stage once the A registers of a row
                    // type have been consumed (slots 32, 48, end)
                    if (m == 2) row_stage_read(BNT{}, I0{}, d);
                    if (m == 5) row_stage_math(0, d);
                    if (m == 8) row_stage_read(BNT{}, I1{}, d);
                    if (m == 11) row_stage_math(1, d);
                    if (m == 14) row_stage_read(BNT{}, I2{}, d);
                    if (m == 17) row_stage_math(2, d);
                    if (m == 20) row_stage_read(BNT{}, I3{}, d);
                    if (m == 23) row_stage_math(3, d);
                    if (m == 26) row_stage_read(BNT{}, I4{}, d);
                    if (m == 29) row_stage_math(4, d);
                    if (m == 32) col_stage(0);
                    if (m == 48) col_stage(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
        col_stage(2);
        __syncthreads();
    };
    {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        chunk_body(cb, std::true_type{}, P0{});
        int c = cb + 1;
        for (; c + 1 < ce; c += 2) {
            chunk_body(c, std::false_type{}, P1{});
            chunk_body(c + 1, std::false_type{}, P0{});
        }
        if (c < ce) chunk_body(c, std::false_type{}, P1{});
    }

    DM_STAMP_ADD(3)
    // ---- epilogue.  Per wave and row type a: p = M[a][plus] + M[a][minus], m = M[a][plus] - M[a][minus] (in place), then
    // for every output column beta:  R_a[beta] = c1[beta] * (beta even ? p : m) + cP[beta] * M[a][P]   (column half CT)
    //   low half (columns 0,1,2):  c1 = 1, 1, 1, 1   cP = 1, 0, 0, 0      high half (columns 5,3,4):  c1 = 1, 2, 4, 8   cP = 0, 0, 0, 1
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 vp = acc[3 * a + 1][gq], vm = acc[3 * a + 2][gq];
            acc[3 * a + 1][gq] = add4(vp, vm);
            acc[3 * a + 2][gq] = sub4(vp, vm);
        }
    const int c4 = l15 * 4;
    const int sub = lane >> 4;
    const int cg = n_tile * 64 + c4;
    const bool cvalid = cg < p.Cout;
    RowsEpilogue re;
    re.split = split;
    re.M = (size_t)p.B * p.Ho * p.Wo;
    re.b0 = b0;
    re.uni = NB == 1 || p.ss_stride == 0;
    re.HoWo = p.Ho * p.Wo;
    re.red = nullptr;
    re.rows_per_wg = 0;
    re.row_in_wg0 = 0;
    re.wn = 0;
    re.all_valid = true;
    const int* ptab = reinterpret_cast<const int*>(smem + g.ptab_off);
    const int my_tile = 4 * wave + sub;  // tile this lane group finishes
    // Everything the four column passes read from global memory (bias, gain, scale / shift, residual rows) is requested
    // now: the loads fly while the partial rows go through LDS.
    int pixv[2][8];
    RowsPrefetch<8, true> pf[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int4 t4 = *reinterpret_cast<const int4*>(ptab + (my_tile * 4 + 2 * h + bb) * 4);
            pixv[h][4 * bb + 0] = t4.x;
            pixv[h][4 * bb + 1] = t4.y;
            pixv[h][4 * bb + 2] = t4.z;
            pixv[h][4 * bb + 3] = t4.w;
        }
        rows_prefetch<8, true>(p, re, pixv[h], cg, cvalid, pf[h]);
    }
    // staging: [beta parity][wave][a][tile][cout]
    constexpr int RSZ = 12 * W4TILES * W4WTS;  // floats per beta
    float* Rw = smem + (size_t)wave * 3 * W4TILES * W4WTS;
    auto beta_pair = [&](auto h_tag) {
        constexpr int h = decltype(h_tag)::value;  // output columns 2h and 2h + 1
        // accumulator register e of lane (n = l15, kq): tile 4 kq + e, cout 16 gq + n
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            constexpr int beta0 = 2 * h;
            const int beta = beta0 + bb;
            const float c1 = CT ? (float)(1 << beta) : 1.0f;
            const float cP = CT ? (beta == 3 ? 1.0f : 0.0f) : (beta == 0 ? 1.0f : 0.0f);
            const f32x2 c12 = {c1, c1}, cP2 = {cP, cP};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 x = acc[3 * a + 1 + bb][gq];  // beta even: p, beta odd: m
                    const f32x4 mp = acc[3 * a][gq];
                    const f32x4 r = join4(pk_fma(cP2, mp.xy, pk_mul(c12, x.xy)), pk_fma(cP2, mp.zw, pk_mul(c12, x.zw)));
                    float* dst = Rw + bb * RSZ + (a * W4TILES + 4 * kq) * W4WTS + 16 * gq + l15;
                    dst[0 * W4WTS] = r.x;
                    dst[1 * W4WTS] = r.y;
                    dst[2 * W4WTS] = r.z;
                    dst[3 * W4WTS] = r.w;
                }
        }
        __syncthreads();
        DM_STAMP_ADD(4)
        // X_i = sum over the two column halves; i = 0 (lo, P), 1 (lo, +), 2 (lo, -), 3 (hi, +), 4 (hi, -), 5 (hi, P)
        f32x4 v[8];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            auto X = [&](int rt, int a) {
                const float* r0 = smem + bb * RSZ + ((size_t)((2 * rt + 0) * 3 + a) * W4TILES + my_tile) * W4WTS + c4;
                const float* r1 = smem + bb * RSZ + ((size_t)((2 * rt + 1) * 3 + a) * W4TILES + my_tile) * W4WTS + c4;
                return add4(*reinterpret_cast<const f32x4*>(r0), *reinterpret_cast<const f32x4*>(r1));
            };
            const f32x4 x0 = X(0, 0), x1 = X(0, 1), x2 = X(0, 2), x3 = X(1, 1), x4 = X(1, 2), x5 = X(1, 0);
            const f32x4 s12 = add4(x1, x2), d12 = sub4(x1, x2), s34 = add4(x3, x4), d34 = sub4(x3, x4);
            const f32x2 two = {2.f, 2.f}, four = {4.f, 4.f}, eight = {8.f, 8.f};
            v[4 * bb + 0] = add4(add4(x0, s12), s34);
            v[4 * bb + 1] = fma4(d34, two, d12);
            v[4 * bb + 2] = fma4(s34, four, s12);
            v[4 * bb + 3] = add4(fma4(d34, eight, d12), x5);
        }
        DM_STAMP_ADD(5)
        rows_epilogue<1, 8, true>(p, re, v, pixv[h], cg, cvalid, pf[h]);
        DM_STAMP_ADD(6)
    };
    beta_pair(I0{});
    __syncthreads();
    DM_STAMP_ADD(7)
    beta_pair(I1{});
    DM_STAMP_FLUSH
}

template <int LTW>
static int wino4_launch_t(const ConvParams& p, int blocks, hipStream_t s) {
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(wino4_mfma_kernel<LTW>), 1)) return 1;
#ifdef DM_STAMPS
    // diagnostic build: run the launch synchronously with a stamp buffer and print the phase averages
    {
        const size_t nblk = (size_t)blocks * p.geo.splits;
        unsigned long long* dbuf = nullptr;
        DM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), nblk * 8 * sizeof(unsigned long long)));
        DM_CHECK_HIP(hipMemsetAsync(dbuf, 0, nblk * 8 * sizeof(unsigned long long), s));
        ConvParams ps = p;
        ps.stamps = dbuf;
        hipLaunchKernelGGL(wino4_mfma_kernel<LTW>, dim3(blocks, p.geo.splits, 1), dim3(256), p.geo.lds_bytes, s, ps);
        DM_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nblk * 8);
        DM_CHECK_HIP(hipMemcpy(h.data(), dbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(dbuf);
        double avg[8] = {0};
        for (size_t b = 0; b < nblk; ++b)
            for (int k = 0; k < 8; ++k) avg[k] += (double)h[b * 8 + k] / nblk;
        fprintf(stderr, "STAMPS wino4<%d> %d+%d->%d @%dx%d e%d k%d chunks %d: wgs=%zu | setup %.0f loads->lds %.0f "
                        "transform0 %.0f loop %.0f (%.0f/chunk) stage1+bar %.0f final %.0f rows_epi %.0f midbar %.0f\n",
                LTW, p.C0, p.C1, p.Cout, p.Ho, p.Wo, p.epi, p.geo.splits, p.geo.chunks_per_split, nblk, avg[0], avg[1],
                avg[2], avg[3], avg[3] / p.geo.chunks_per_split, avg[4], avg[5], avg[6], avg[7]);
        return 0;
    }
#endif
    hipLaunchKernelGGL(wino4_mfma_kernel<LTW>, dim3(blocks, p.geo.splits, 1), dim3(256), p.geo.lds_bytes, s, p);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

int wino4_launch(const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    p.stamps = nullptr;
    const ConvGeom& g = p.geo;
    DM_REQUIRE(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && !p.up && !p.fold, "winograd4: 3x3 s1 p1 only");
    DM_REQUIRE(!p.in_nchw && !p.out_nchw, "winograd4: NHWC only");
    DM_REQUIRE(p.C0 % W4CK == 0 && p.C1 % W4CK == 0 && p.Cout % 64 == 0, "winograd4: channel counts");
    DM_REQUIRE(p.Hin == p.Ho && p.Win == p.Wo, "winograd4: same-size convolution");
    const int ltw = w4_class(p.Ho, p.Wo);
    DM_REQUIRE(ltw >= 0 && ltw == g.lTW, "winograd4: image size has no geometry class");
    DM_REQUIRE(g.TW * g.TH * g.NB == W4TILES, "winograd4: 16 tiles per workgroup");
    DM_REQUIRE(4 * g.TW * g.tiles_x == p.Wo && 4 * g.TH * g.tiles_y == p.Ho, "winograd4: tiles must cover the image exactly");
    DM_REQUIRE((size_t)p.B * p.Ho * p.Wo < (1u << 24) && p.C0 < (1 << 24) && p.C1 < (1 << 24) &&
                   (size_t)p.B * p.Ho * p.Wo * std::max(p.C0, p.C1) < (1ull << 30),
               "winograd4: tensor too large for 24-bit pixel indices");
    DM_REQUIRE(!(p.epi & EPI_NORM) || (g.n_tiles_n == 1 && g.splits == 1), "winograd4: fused RMSNorm needs one N tile");
    DM_REQUIRE(g.splits == 1 || p.partial, "winograd4: split-K writes partial sums");
    DM_REQUIRE(g.lds_bytes <= 160 * 1024, "winograd4: tile does not fit LDS");
    DM_REQUIRE(p.chunks0 == p.C0 / W4CK && p.n_chunks == (p.C0 + p.C1) / W4CK, "winograd4: chunk counts");
    const int blocks = g.n_tiles_n * g.tiles_x * g.tiles_y * g.groups;
    // XCD-aware block order (conv_device.h: block_to_tile); DM_NO_XCD_ORDER=1 keeps the raw order for A/B runs
    static const bool xcd_order = env_int("DM_NO_XCD_ORDER", 0) == 0;
    p.geo.xcd_groups = (xcd_order && blocks % 8 == 0 && 8 % g.n_tiles_n == 0) ? 8 / g.n_tiles_n : 0;
    const bool timed = prof::enabled();
    if (timed) {
        // priced as the reference's op: 2*9*Cin*Cout*pixels FLOP; the kernel executes 36/144 of the
        // multiply-adds of that count
        const double pix = (double)p.B * p.Ho * p.Wo;
        const double cin = p.C0 + p.C1;
        const double flops = 2.0 * 9.0 * cin * p.Cout * pix;
        const double res_rows = (!p.partial && (p.epi & EPI_RESIDUAL)) ? 1.0 : 0.0;  // the fused residual add reads one more tensor
        const double bytes = 4.0 * (cin * pix + (1.0 + res_rows) * p.Cout * pix + 9.0 * cin * p.Cout);
        char name[64];
        if (prof::detail())
            snprintf(name, sizeof(name), "wino4<%d> 3x3 s1 %d+%d->%d @%dx%d e%d k%d g%d", ltw, p.C0, p.C1, p.Cout, p.Ho, p.Wo,
                     p.epi, g.splits, blocks * g.splits);
        else
            snprintf(name, sizeof(name), "wino4_mfma_kernel<%d>", ltw);
        if (prof::begin(name, flops, bytes, s)) return 1;
    }
    int rc;
    switch (ltw) {
        case 0: rc = wino4_launch_t<0>(p, blocks, s); break;
        case 1: rc = wino4_launch_t<1>(p, blocks, s); break;
        default: rc = wino4_launch_t<2>(p, blocks, s); break;
    }
    if (rc) return 1;
    if (timed && prof::end(s)) return 1;
    return 0;
}

}  // namespace dm

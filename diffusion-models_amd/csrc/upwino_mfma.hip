// Nearest-neighbour x2 upsampling followed by a 3x3 / pad 1 convolution (Upsample, DD/denoising_diffusion.py:48-52;
// VAE Upsample, LD/modules/diffusionmodules/model.py:70-74) as a minimal bilinear algorithm on the SOURCE grid.
//
// Every source pixel l(i, j) yields 2x2 outputs, and along one axis (u = upsampled signal, zero padded)
//     y(2i)     = g0 u(2i-1) + g1 u(2i)   + g2 u(2i+1) = g0 l(i-1) + (g1 + g2) l(i)
//     y(2i + 1) = g0 u(2i)   + g1 u(2i+1) + g2 u(2i+2) = (g0 + g1) l(i) + g2 l(i+1)
// needs three products instead of four:
//     M0 = g0 (l(i-1) - l(i)),  M1 = (g0 + g1 + g2) l(i),  M2 = g2 (l(i+1) - l(i));   y(2i) = M0 + M1,  y(2i+1) = M1 + M2
// In two dimensions, per source pixel and (cin, cout) pair:
//     Y (2x2) = A^T [ (G g G^T) (.) (T d T^T) ] A       d = 3x3 source patch, g = 3x3 filter
//     T = [1 -1 0; 0 1 0; 0 -1 1]     G = [1 0 0; 1 1 1; 0 0 1]     A^T = [1 1 0; 0 1 1]
// 9 multiplies per 4 outputs: 2.25 per output against 4 for the four 2x2 parity convolutions of the folded direct
// kernel (conv_mfma.hip, ConvParams::fold) and 9 for the convolution as the reference runs it.  All transform
// coefficients are +-1: the rounding behaviour is that of the direct convolution with two extra additions per operand.
//
// Mapping (the register / MFMA structure of wino4_mfma.hip, minus its cross-wave exchange): one workgroup = 4 waves,
// one per SIMD; a wave owns a 4x4 block of source pixels (16 "tiles" = the 16 rows of v_mfma_f32_16x16x4_f32) x 64
// couts x all 9 positions xi = 3a + b: 144 accumulator registers.  The four waves cover an 8x8 block of one image
// (class 1: source sizes that are multiples of 8) or four 4x4 images (class 0); they use the same weights.
//   * B operand: lane (cout n = lane & 15, channel pair kq = lane >> 4) holds U[xi][16 gq + n][2 kq + st] for the 4 cout
//     groups gq and 2 K steps st: two 16-byte buffer loads per position, one chunk ahead.
//   * A operand: the raw source window (10x10 or 4 x 6x6 pixels x 8 channels) is staged in LDS, double buffered, one
//     barrier per chunk; lane (tile, kq) reads its 3x3 patch as channel pairs (9 ds_read_b64, conflict free: rows are
//     80 / 48 floats apart -- two rows = 32 banks -- and odd rows swap the two 16-byte halves of a pixel) and forms its
//     nine V with 12 packed subtractions.
//   * Epilogue: Y = A^T M A in registers (no exchange between waves), transposed through LDS into the row layout of the
//     shared Block epilogue (conv_device.h), whose global operands are requested before the transposition.
#include "conv_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace dm {

static constexpr int UWCK = 8;   // input channels per K chunk
static constexpr int UWTS = 68;  // row stride (floats) of the epilogue staging tiles: 4 rows = 16 banks

template <int CLS>
struct UWGeo {
    static constexpr int NB = CLS ? 1 : 4;    // images per workgroup
    static constexpr int IH = CLS ? 10 : 6;   // window rows / columns per image (block + 1 pixel ring)
    static constexpr int IW = IH;
    static constexpr int RS = IW * UWCK;      // floats per window row
    static constexpr int BUF = NB * IH * RS;  // floats per window buffer
};


bool upwino_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up) {
    static const bool off = std::getenv("DM_NO_UPWINO") != nullptr || std::getenv("DM_NO_WINOGRAD") != nullptr;
    return !off && up && KH == 3 && KW == 3 && stride == 1 && pad == 1 && C0 > 0 && C0 % UWCK == 0 && C1 == 0 &&
           Cout % 64 == 0;
}

size_t upwino_packed_floats(int Cout, int C0, int C1) { return (size_t)(C0 + C1) * 9 * Cout; }

void upwino_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1) {
    static const double G[3][3] = {{1, 0, 0}, {1, 1, 1}, {0, 0, 1}};
    const int Cin = C0 + C1;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float* gk = oihw + ((size_t)co * Cin + ci) * 9;
            double Gg[3][3];
            for (int i = 0; i < 3; ++i)
                for (int b = 0; b < 3; ++b) Gg[i][b] = G[i][0] * gk[b] + G[i][1] * gk[3 + b] + G[i][2] * gk[6 + b];
            const int chunk = ci / UWCK, cc = ci % UWCK;
            const int ct = co / 64, gq = (co % 64) / 16, n = co % 16, kq = cc / 2, st = cc % 2;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    const double u = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
                    // [chunk][xi = 3 i + j][cout tile of 64][kq 4][n 16][gq 4][st 2]: channel cc = 2 kq + st, cout = 16 gq + n
                    packed[((((size_t)chunk * 9 + 3 * i + j) * (Cout / 64) + ct) * 4 + kq) * 128 + n * 8 + gq * 2 + st] =
                        (float)u;
                }
        }
}

// geometry class of a source image: 1 = 8x8 blocks, 0 = whole 4x4 images, -1 = not served
static int uw_class(int Hl, int Wl) {
    if (Hl == 4 && Wl == 4) return 0;
    if (Hl > 0 && Wl > 0 && Hl % 8 == 0 && Wl % 8 == 0) return 1;
    return -1;
}

// (B, Hl, Wl) = the SOURCE tensor; the output is (B, 2 Hl, 2 Wl)
ConvGeom upwino_plan(int B, int Hl, int Wl, int Cout, int C0, int C1, bool allow_split) {
    ConvGeom g{};
    const int cls = std::max(uw_class(Hl, Wl), 0);
    g.WM = 1;
    g.WN = 1;
    g.CK = UWCK;
    g.lTW = g.lTH = cls;  // the class travels in lTW
    g.TW = g.TH = cls ? 8 : 4;
    g.NB = cls ? 1 : 4;
    g.tiles_x = cls ? Wl / 8 : 1;
    g.tiles_y = cls ? Hl / 8 : 1;
    g.groups = (B + g.NB - 1) / g.NB;
    g.n_tiles_n = Cout / 64;
    g.IH = g.IW = cls ? 10 : 6;
    g.row_stride = g.IW * UWCK;
    g.halo_floats = g.NB * g.IH * g.row_stride;
    g.TPS = 3;
    const int n_chunks = (C0 + C1) / UWCK;
    const int wgs = g.tiles_x * g.tiles_y * g.groups * g.n_tiles_n;
    int splits = 1;
    if (allow_split) {
        static const int target = env_int("DM_UPWINO_TARGET_WGS", 256);
        static const int min_chunks = env_int("DM_UPWINO_MIN_CHUNKS", 8);
        while (wgs * splits < target && splits < 8 && n_chunks / (splits * 2) >= min_chunks) splits *= 2;
    }
    g.chunks_per_split = (n_chunks + splits - 1) / splits;
    g.splits = (n_chunks + g.chunks_per_split - 1) / g.chunks_per_split;
    g.fused_norm = g.n_tiles_n == 1 && g.splits == 1;
    g.w_floats = 0;
    // two window buffers + a scratch slot reachable from both (items outside the image are stored at buffer + 2 BUF);
    // the epilogue reuses the space for one 64-row x 64-cout staging tile per wave
    g.ptab_off = std::max(3 * g.halo_floats + 16, 4 * 64 * UWTS);
    g.lds_bytes = g.ptab_off * 4;
    return g;
}

bool upwino_shape_ok(int B, int Hl, int Wl, int Cout, int C0, int C1) {
    if (uw_class(Hl, Wl) < 0) return false;
    const ConvGeom g = upwino_plan(B, Hl, Wl, Cout, C0, C1, true);
    // one 4-wave workgroup per CU: needs enough workgroups to cover the chip and a reduction that amortises its
    // prologue / epilogue (the folded direct kernel keeps the rest)
    static const int min_wgs = env_int("DM_UPWINO_MIN_WGS", 128);
    static const int min_k = env_int("DM_UPWINO_MIN_K", 8);
    const int wgs = g.tiles_x * g.tiles_y * g.groups * g.n_tiles_n * g.splits;
    const size_t out_px = (size_t)B * Hl * Wl * 4;
    return wgs >= min_wgs && g.chunks_per_split >= min_k && out_px < (1u << 24) &&
           out_px * (size_t)std::max(Cout, C0) < (1ull << 30);
}

template <int CLS>
__global__ __launch_bounds__(256, 1) void upwino_mfma_kernel(const ConvParams p) {
    using G = UWGeo<CLS>;
    constexpr int NB = G::NB, IH = G::IH, IW = G::IW, RS = G::RS, BUF = G::BUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = p.geo;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;  // MFMA row (source pixel of the wave's 4x4 block) of A / column (cout) of B
    const int kq = lane >> 4;   // channel pair (2 kq, 2 kq + 1) of the chunk

    int n_tile, bid;
    block_to_tile(g, blockIdx.x, gridDim.x, n_tile, bid);
    const int tile_x = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int tile_y = bid % g.tiles_y;
    const int group = bid / g.tiles_y;
    const int b0 = group * NB;
    // source-pixel origin of the workgroup's block and of this wave's 4x4 block inside the window (ring excluded)
    const int ly0 = CLS ? 8 * tile_y : 0, lx0 = CLS ? 8 * tile_x : 0;
    const int wy = CLS ? 4 * (wave >> 1) : 0, wx = CLS ? 4 * (wave & 1) : 0, wnb = CLS ? 0 : wave;
    const int split = blockIdx.y;
    const int cb = split * g.chunks_per_split;
    const int ce = min(cb + g.chunks_per_split, p.n_chunks);
    float* raw[2] = {smem, smem + BUF};
    constexpr int SCRATCH = 2 * BUF;  // floats from the buffer base: items outside the image land here (either buffer)

    // ---- window staging: one item = (window pixel, channel quad) per thread
    int hpix = 0, hoff = SCRATCH;
    {
        const int hp = tid >> 1, qd = tid & 1;
        int nb, hy, hx;
        bool in_region;
        if constexpr (CLS == 1) {  // 10x10 window of one image
            hy = hp / IW;
            hx = hp - hy * IW;
            nb = 0;
            in_region = hp < IH * IW;
        } else {  // four whole 4x4 images: the ring is padding
            nb = hp >> 4;
            hy = ((hp >> 2) & 3) + 1;
            hx = (hp & 3) + 1;
            in_region = nb < NB;
        }
        const int b = b0 + nb, iy = ly0 - 1 + hy, ix = lx0 - 1 + hx;
        if (in_region && b < p.B && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
            hpix = (b * p.Hin + iy) * p.Win + ix;
            hoff = (nb * IH + hy) * RS + 4 * ((2 * hx + qd) ^ (hy & 1));
        }
    }
    const size_t in_px = (size_t)p.B * p.Hin * p.Win;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in0, in_px * p.C0 * 4);
    const unsigned hvo = (__umul24((unsigned)hpix, (unsigned)p.C0) + 4 * (tid & 1)) * 4;
    auto window_value = [&](int chunk) { return bufload4(rs_in, hvo, (unsigned)chunk * (UWCK * 4)); };

    // ---- patch addressing of this lane: source pixel (ty, tx) of the wave's block, channel pair kq; byte offsets inside a
    //      window buffer of patch element (r, c)
    unsigned aoff[3][3];
    {
        const int tx = l15 & 3, ty = l15 >> 2;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int hy = wy + ty + r, hx = wx + tx + c;
                aoff[r][c] = (unsigned)(((wnb * IH + hy) * RS + 4 * ((2 * hx + (kq >> 1)) ^ (hy & 1)) + 2 * (kq & 1)) * 4);
            }
    }
    const char* sbytes = reinterpret_cast<const char*>(smem);
    auto rd2 = [&](unsigned byte_off) { return *reinterpret_cast<const f32x2*>(sbytes + byte_off); };

    // ---- weights: lane (n = l15, kq) loads its 8 floats [gq 4][st 2] of a position as two 16-byte loads
    const size_t u_chunk = (size_t)9 * p.Cout * UWCK;  // floats per chunk
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (size_t)p.n_chunks * u_chunk * 4);
    const unsigned uvo = (unsigned)((kq * 128 + l15 * 8) * 4);
    const unsigned u_slot = (unsigned)p.Cout * UWCK * 4;  // bytes between consecutive positions
    const unsigned u_tile = (unsigned)n_tile * (64 * UWCK * 4);
    f32x4 U[9][2];  // [xi][gq pair]: .xy = (gq even, st 0 / 1), .zw = (gq odd, st 0 / 1)
    auto load_u = [&](int chunk, int k) {
        const unsigned so = (unsigned)chunk * (unsigned)(u_chunk * 4) + u_tile + k * u_slot;
        U[k][0] = bufload4(rs_w, uvo, so);
        U[k][1] = bufload4(rs_w, uvo, so + 16);
    };

    f32x2 A[9];       // V of the current chunk: position k, channels 2 kq (.x, K step 0) and 2 kq + 1 (.y, K step 1)
    f32x4 acc[9][4];  // [xi][cout group]; first written by the first chunk's MFMAs (C = 0)
    f32x2 d[3][3];    // patch of the next chunk, then its row-transformed values (rows 0 and 2 in place)

    auto patch_read = [&](auto bf_tag, int r) {
        constexpr unsigned base = decltype(bf_tag)::value * BUF * 4;
#pragma unroll
        for (int c = 0; c < 3; ++c) d[r][c] = rd2(aoff[r][c] + base);
    };
    auto row_stage = [&]() {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d[0][c] = pk_sub(d[0][c], d[1][c]);
            d[2][c] = pk_sub(d[2][c], d[1][c]);
        }
    };
    auto col_stage = [&](int a) {
        A[3 * a + 0] = pk_sub(d[a][0], d[a][1]);
        A[3 * a + 1] = d[a][1];
        A[3 * a + 2] = pk_sub(d[a][2], d[a][1]);
    };

    using I0 = std::integral_constant<int, 0>;

    // ---- prologue: request chunks cb and cb + 1 and the weights of chunk cb first, do the LDS housekeeping while they fly
    f32x4 hreg, h2;
    {
        hreg = window_value(cb);
        h2 = window_value(cb + 1 < ce ? cb + 1 : cb);
#pragma unroll
        for (int k = 0; k < 9; ++k) load_u(cb, k);
    }
    {
        const f32x4 z4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < 2 * BUF / 4; i += 256) *reinterpret_cast<f32x4*>(smem + 4 * i) = z4;
    }
    __syncthreads();  // zero fill done (padding pixels stay zero)
    *reinterpret_cast<f32x4*>(raw[0] + hoff) = hreg;
    *reinterpret_cast<f32x4*>(raw[1] + hoff) = h2;
    __syncthreads();
    patch_read(I0{}, 0);
    patch_read(I0{}, 1);
    patch_read(I0{}, 2);
    row_stage();
#pragma unroll
    for (int a = 0; a < 3; ++a) col_stage(a);
    __syncthreads();  // buffer 0 is overwritten with chunk cb + 2 by the first iteration

    // ---- main loop: 72 MFMAs per chunk and wave; everything else rides in the hooks between them
    const f32x4 zero4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    auto chunk_body = [&](int c, auto first_tag, auto par_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;  // (c - cb) & 1: chunk c was read from buffer PAR
        constexpr int BN = PAR ^ 1, BS = PAR;          // chunk c + 1 sits in BN; chunk c + 2 goes to BS
        using BNT = std::integral_constant<int, BN>;
        const bool has1 = c + 1 < ce, has2 = c + 2 < ce;
        const int cw = has2 ? c + 2 : c;   // window fetched now (c again at the end: stored, never read)
        const int cun = has1 ? c + 1 : c;  // chunk whose weights are fetched now (never past the packed weights)
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int m = (k * 2 + st) * 4 + gq;  // 0..71
                    acc[k][gq] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[k][st], U[k][gq >> 1][2 * (gq & 1) + st],
                                                                      (FIRST && st == 0) ? zero4 : acc[k][gq], 0, 0, 0);
                    // window of chunk c + 2: load early, store late
                    if (m == 0) hreg = window_value(cw);
                    if (m == 60) *reinterpret_cast<f32x4*>(raw[BS] + hoff) = hreg;
                    // weights of chunk c + 1 into the registers the MFMAs of position k have just used
                    if (m == 8 * k + 7) load_u(cun, k);
                    // transform of chunk c + 1: patch and row stage early, column stage once the A registers of a row have
                    // been consumed (positions 0-2 by slot 23, 3-5 by slot 47, 6-8 at the end)
                    if (m == 2) patch_read(BNT{}, 0);
                    if (m == 4) patch_read(BNT{}, 1);
                    if (m == 6) patch_read(BNT{}, 2);
                    if (m == 16) row_stage();
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

    // ---- epilogue.  Row layout of the shared epilogue: v[j] = couts [cg, cg + 4) of staging row 4 j + rsub.  Staging row
    // rho = 16 (2 a + b) + tile, so the lane group rsub finishes source column tx = rsub, v[j] is output (a, b) = (j >> 3,
    // (j >> 2) & 1) of source row ty = j & 3.
    const int rsub = lane >> 4;
    const int c4 = l15 * 4;
    const int cg = n_tile * 64 + c4;
    const bool cvalid = cg < p.Cout;
    RowsEpilogue re;
    re.split = split;
    re.M = (size_t)p.B * p.Ho * p.Wo;
    re.b0 = b0 + wnb;
    re.uni = true;  // every wave works on one image
    re.HoWo = p.Ho * p.Wo;
    re.red = nullptr;
    re.rows_per_wg = 0;
    re.row_in_wg0 = 0;
    re.wn = 0;
    re.all_valid = true;
    int pixv[16];
    {
        const int b = b0 + wnb;
        const int lx = lx0 + wx + rsub;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int ly = ly0 + wy + (j & 3);
            const int y = 2 * ly + (j >> 3), x = 2 * lx + ((j >> 2) & 1);
            pixv[j] = (b < p.B && ly < p.Hin && lx < p.Win) ? (b * p.Ho + y) * p.Wo + x : -1;
        }
    }
    RowsPrefetch<16, true> pf;
    rows_prefetch<16, true>(p, re, pixv, cg, cvalid, pf);

    // Y = A^T M A in registers, then accumulator layout (lane = cout 16 gq + n, register e = tile 4 kq + e) -> staging rows
    float* S = smem + (size_t)wave * 64 * UWTS;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        f32x4 R0[3], R1[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            R0[b] = add4(acc[b][gq], acc[3 + b][gq]);
            R1[b] = add4(acc[3 + b][gq], acc[6 + b][gq]);
        }
        const f32x4 y4[4] = {add4(R0[0], R0[1]), add4(R0[1], R0[2]), add4(R1[0], R1[1]), add4(R1[1], R1[2])};
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            float* dst = S + (16 * ab + 4 * kq) * UWTS + 16 * gq + l15;
            dst[0 * UWTS] = y4[ab].x;
            dst[1 * UWTS] = y4[ab].y;
            dst[2 * UWTS] = y4[ab].z;
            dst[3 * UWTS] = y4[ab].w;
        }
    }
    __builtin_amdgcn_wave_barrier();
    f32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f32x4*>(S + (4 * j + rsub) * UWTS + c4);
    rows_epilogue<1, 16, true>(p, re, v, pixv, cg, cvalid, pf);
}

template <int CLS>
static int upwino_launch_t(const ConvParams& p, int blocks, hipStream_t s) {
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(upwino_mfma_kernel<CLS>), 1)) return 1;
    hipLaunchKernelGGL(upwino_mfma_kernel<CLS>, dim3(blocks, p.geo.splits, 1), dim3(256), p.geo.lds_bytes, s, p);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// p.Hin / p.Win = the SOURCE size, p.Ho / p.Wo = twice that; p.w = upwino_pack_weights; p.geo = upwino_plan
int upwino_launch(const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    p.stamps = nullptr;
    const ConvGeom& g = p.geo;
    DM_REQUIRE(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.up && !p.fold, "upwino: nearest x2 + 3x3 s1 p1 only");
    DM_REQUIRE(!p.in_nchw && !p.out_nchw, "upwino: NHWC only");
    DM_REQUIRE(p.C0 % UWCK == 0 && p.C1 == 0 && p.Cout % 64 == 0, "upwino: channel counts");
    DM_REQUIRE(p.Ho == 2 * p.Hin && p.Wo == 2 * p.Win, "upwino: the output is twice the source");
    const int cls = uw_class(p.Hin, p.Win);
    DM_REQUIRE(cls >= 0 && cls == g.lTW, "upwino: source size has no geometry class");
    DM_REQUIRE(g.NB == (cls ? 1 : 4) && g.tiles_x == (cls ? p.Win / 8 : 1) && g.tiles_y == (cls ? p.Hin / 8 : 1) &&
                   g.groups == (p.B + g.NB - 1) / g.NB && g.n_tiles_n == p.Cout / 64,
               "upwino: plan does not match the tensor");
    DM_REQUIRE((size_t)p.B * p.Ho * p.Wo < (1u << 24) &&
                   (size_t)p.B * p.Ho * p.Wo * std::max(p.C0, p.Cout) < (1ull << 30),
               "upwino: tensor too large for 24-bit pixel indices");
    DM_REQUIRE(!(p.epi & EPI_NORM) || (g.n_tiles_n == 1 && g.splits == 1), "upwino: fused RMSNorm needs one N tile");
    DM_REQUIRE(g.splits == 1 || p.partial, "upwino: split-K writes partial sums");
    DM_REQUIRE(g.lds_bytes <= 160 * 1024 && g.lds_bytes >= (3 * g.halo_floats + 16) * 4 && g.lds_bytes >= 4 * 64 * UWTS * 4,
               "upwino: LDS size");
    DM_REQUIRE(p.chunks0 == p.C0 / UWCK && p.n_chunks == p.C0 / UWCK, "upwino: chunk counts");
    DM_REQUIRE(g.splits * g.chunks_per_split >= p.n_chunks && (g.splits - 1) * g.chunks_per_split < p.n_chunks,
               "upwino: K split does not cover the chunks");
    const int blocks = g.n_tiles_n * g.tiles_x * g.tiles_y * g.groups;
    static const bool xcd_order = env_int("DM_NO_XCD_ORDER", 0) == 0;
    p.geo.xcd_groups = (xcd_order && blocks % 8 == 0 && 8 % g.n_tiles_n == 0) ? 8 / g.n_tiles_n : 0;
    const bool timed = prof::enabled();
    if (timed) {
        // priced as the reference's op: 2*9*Cin*Cout FLOP per OUTPUT pixel; the kernel executes 9/36 of
        // the multiply-adds of that count
        const double pix = (double)p.B * p.Ho * p.Wo;
        const double flops = 2.0 * 9.0 * p.C0 * p.Cout * pix;
        const double res_rows = (!p.partial && (p.epi & EPI_RESIDUAL)) ? 1.0 : 0.0;  // the fused residual add reads one more tensor
        const double bytes = 4.0 * (p.C0 * pix / 4 + (1.0 + res_rows) * p.Cout * pix + 9.0 * p.C0 * p.Cout);
        char name[64];
        if (prof::detail())
            snprintf(name, sizeof(name), "upwino<%d> 3x3 up %d->%d @%dx%d e%d k%d g%d", cls, p.C0, p.Cout, p.Ho, p.Wo, p.epi,
                     g.splits, blocks * g.splits);
        else
            snprintf(name, sizeof(name), "upwino_mfma_kernel<%d>", cls);
        if (prof::begin(name, flops, bytes, s)) return 1;
    }
    const int rc = cls ? upwino_launch_t<1>(p, blocks, s) : upwino_launch_t<0>(p, blocks, s);
    if (rc) return 1;
    if (timed && prof::end(s)) return 1;
    return 0;
}

}  // namespace dm

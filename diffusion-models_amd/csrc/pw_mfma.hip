// 1x1 convolutions (res_conv DD/denoising_diffusion.py:134, the to_qkv / to_out projections of the attention layers
// :163-170 / :209-213, the VAE's nin_shortcut and q / k / v / proj_out) as a register-direct GEMM on
// v_mfma_f32_16x16x4_f32:  out[pixel][cout] = sum_c in[pixel][c] W[cout][c]   over one or two NHWC sources.
//
// A 1x1 convolution has no window, so neither MFMA operand needs LDS: lane (row l15, kq) of a wave loads the 16 bytes
// in[pixel l15][16 chunk + 4 kq .. + 3] (A) and W'[chunk][cout tile][lane][0..3] (B, packed on the host in exactly that
// order) and feeds component j of both to MFMA step j -- the K index of a step is the channel 16 chunk + 4 kq + j, a
// fixed permutation of the reduction.  No staging loads, no barriers, no LDS traffic in the loop; three chunks are in
// flight per wave and two workgroups share a CU.  (The generic kernel conv_mfma.hip stages a pixel window and the weights
// through LDS for every 16 channels: at these shapes less than half of its time is MFMA issue.)
// One wave = 64 pixels x 64 couts (4 x 4 MFMA tiles, 64 accumulators); one workgroup = 4 waves as 256 px x 64 couts
// (WGN = 1) or 128 px x 128 couts (WGN = 2).  RT < 4 (32 or 16 pixels per wave): for grids that cannot fill the chip -- a
// wave's time is its MFMA chain, chunks x RT x 512 cycles, so a 1024-pixel projection on 48 workgroups takes 14 us of
// matrix issue per wave while 3/4 of the CUs idle; with RT = 1 it is 192 workgroups and a quarter of the chain (the weights
// are then pulled through L2 four times as often, which at these sizes is megabytes).  The 16-row MFMA is chosen for its operand addressing: the four kq lanes
// of a pixel row read 64 consecutive bytes, 16 cache lines per load instruction instead of the 64 of the 32x32x2
// mapping (measured 3 % faster over the 1x1 layers of a forward).  Epilogue: the shared row epilogue (conv_device.h)
// after a per-wave transposition through LDS, 32 pixel rows at a time.
#include "conv_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace dm {

static constexpr int PWCK = 16;  // input channels per K chunk
static constexpr int PWTS = 68;  // row stride (floats) of the epilogue staging tile
static constexpr int PWD = 3;    // chunks in flight per wave


bool pw_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up) {
    static const bool off = std::getenv("DM_NO_PW") != nullptr;
    return !off && KH == 1 && KW == 1 && stride == 1 && pad == 0 && !up && C0 > 0 && C0 % PWCK == 0 && C1 % PWCK == 0 &&
           Cout % 64 == 0;
}

size_t pw_packed_floats(int Cout, int C0, int C1) { return (size_t)(C0 + C1) * Cout; }

// oihw = (Cout, C0 + C1) -> [chunk of 16 channels][cout tile of 16][lane = 16 kq + l15][j 4] = W[16 t + l15][16 chunk + 4 kq + j]
void pw_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1) {
    const int Cin = C0 + C1;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const int chunk = ci / PWCK, cc = ci % PWCK, kq = cc / 4, j = cc % 4;
            const int t = co / 16, l15 = co % 16;
            packed[(((size_t)chunk * (Cout / 16) + t) * 64 + kq * 16 + l15) * 4 + j] = oihw[(size_t)co * Cin + ci];
        }
}

// Downsample weights (Cout, C0, 2, 2) -> the same lane order over K = (sub-pixel 2 dy + dx, channel)
void pw_pack_weights_s2d(const float* oihw, float* packed, int Cout, int C0) {
    std::vector<float> w((size_t)Cout * 4 * C0);
    for (int co = 0; co < Cout; ++co)
        for (int c = 0; c < C0; ++c)
            for (int sub = 0; sub < 4; ++sub) w[(size_t)co * 4 * C0 + sub * C0 + c] = oihw[((size_t)co * C0 + c) * 4 + sub];
    pw_pack_weights(w.data(), packed, Cout, 4 * C0, 0);
}

bool pw_s2d_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up) {
    static const bool off = std::getenv("DM_NO_PW") != nullptr;
    const int cps = C0 / PWCK;
    return !off && KH == 2 && KW == 2 && stride == 2 && pad == 0 && !up && C1 == 0 && C0 > 0 && C0 % PWCK == 0 &&
           (cps & (cps - 1)) == 0 && Cout % 64 == 0;
}

// M = output pixels (B * Ho * Wo)
ConvGeom pw_plan(int B, int Ho, int Wo, int Cout, int C0, int C1, bool allow_split) {
    ConvGeom g{};
    const size_t M = (size_t)B * Ho * Wo;
    g.WN = Cout % 128 == 0 ? 2 : 1;
    g.WM = 4 / g.WN;
    g.CK = PWCK;
    g.TW = g.NB = 1;
    g.TH = 4;  // RT: 16-pixel row tiles per wave (4, 2 or 1; chosen below)
    g.tiles_x = (int)((M + 64 * g.WM - 1) / (64 * g.WM));  // pixel blocks
    g.tiles_y = 1;
    g.groups = 1;
    g.n_tiles_n = Cout / (64 * g.WN);
    g.TPS = 1;
    const int n_chunks = (C0 + C1) / PWCK;
    // A grid that leaves CUs idle is bound by the MFMA chain of ONE wave (chunks x RT x 512 cycles): first fewer pixels per
    // wave (RT 4 -> 2 -> 1: more workgroups, a shorter chain, no extra output traffic), then K splits (partial sums + a
    // landing pass).  DM_PW_RT_FIRST=0: the other order.
    static const int rt_target = env_int("DM_PW_RT_TARGET_WGS", 512);
    static const int rt_min = env_int("DM_PW_RT_MIN", 1);
    static const bool rt_first = env_int("DM_PW_RT_FIRST", 1) != 0;
    auto shrink_rt = [&](int splits_now) {
        while (g.TH > rt_min && g.tiles_x * g.n_tiles_n * splits_now < rt_target) {
            g.TH /= 2;
            g.tiles_x = (int)((M + 16 * g.TH * g.WM - 1) / (16 * g.TH * g.WM));
        }
    };
    if (rt_first) shrink_rt(1);
    int splits = 1;
    if (allow_split) {
        static const int target = env_int("DM_PW_TARGET_WGS", 256);
        static const int min_chunks = env_int("DM_PW_MIN_CHUNKS", 8);
        while (g.tiles_x * g.n_tiles_n * splits < target && splits < 8 && n_chunks / (splits * 2) >= min_chunks) splits *= 2;
    }
    g.chunks_per_split = (n_chunks + splits - 1) / splits;
    g.splits = (n_chunks + g.chunks_per_split - 1) / g.chunks_per_split;
    if (!rt_first) shrink_rt(g.splits);
    g.fused_norm = Cout == 64 && g.splits == 1;
    g.lds_bytes = 4 * 32 * PWTS * 4;
    return g;
}

bool pw_shape_ok(int B, int Ho, int Wo, int Cout, int C0, int C1) {
    const size_t M = (size_t)B * Ho * Wo;
    return M > 0 && M < (1u << 24) && M * (size_t)std::max(Cout, std::max(C0, C1)) < (1ull << 30);
}

template <int WGN, int RT>
__global__ __launch_bounds__(256, 2) void pw_mfma_kernel(const ConvParams p) {
    constexpr int WGM = 4 / WGN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = p.geo;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int kq = lane >> 4;
    const int wn = wave % WGN, wm = wave / WGN;

    int n_blk, m_blk;
    block_to_tile(g, blockIdx.x, gridDim.x, n_blk, m_blk);
    const int ct = n_blk * WGN + wn;  // 64-cout tile of this wave
    const unsigned M = (unsigned)((size_t)p.B * p.Ho * p.Wo);
    const unsigned px0 = ((unsigned)m_blk * WGM + wm) * (16u * RT);
    const int split = blockIdx.y;
    const int cb = split * g.chunks_per_split;
    const int ce = min(cb + g.chunks_per_split, p.n_chunks);

    // ---- operand addressing: A rows of this lane (clamped: rows past the tensor load row M - 1 and are never stored).
    //      Lane (row l15, kq) of row tile rt loads channels 16 chunk + 4 kq .. + 3 of pixel px0 + 16 rt + l15: the four kq
    //      lanes of a row read 64 consecutive bytes, so one load instruction touches 16 cache lines (the 32x32x2 mapping,
    //      one 16-byte piece of 64 different rows per instruction, left the kernel bound by the texture addresser)
    //      Space-to-depth mode (p.s2d: Downsample = pixel-unshuffle + 1x1, :54-58): the source is (B, 2 Ho, 2 Wo, C0), the K
    //      index runs over (sub-pixel 2 dy + dx, channel); a lane addresses the top-left source pixel of its output pixel
    //      and the sub-pixel is a wave-uniform offset per chunk.
    const unsigned src_px = p.s2d ? 4u * M : M;
    const __amdgpu_buffer_rsrc_t rs_in0 = make_rsrc(p.in0, (size_t)src_px * p.C0 * 4);
    const __amdgpu_buffer_rsrc_t rs_in1 = make_rsrc(p.C1 ? p.in1 : p.in0, (size_t)M * (p.C1 ? p.C1 : p.C0) * 4);
    unsigned avo0[RT], avo1[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        unsigned px = min(px0 + 16u * rt + (unsigned)l15, M - 1u);
        avo1[rt] = (__umul24(px, (unsigned)p.C1) + 4u * kq) * 4u;
        if (p.s2d) {
            const unsigned hw = (unsigned)(p.Ho * p.Wo), bimg = px / hw, r = px - bimg * hw;
            const unsigned y = r / (unsigned)p.Wo, x = r - y * (unsigned)p.Wo;
            px = (bimg * 2u * p.Ho + 2u * y) * (2u * p.Wo) + 2u * x;
        }
        avo0[rt] = (__umul24(px, (unsigned)p.C0) + 4u * kq) * 4u;
    }
    const int sub_shift = p.s2d;  // log2(chunks per sub-pixel) + 1 in space-to-depth mode (set by the launcher), else 0
    const size_t w_chunk = (size_t)p.Cout * PWCK;  // floats per chunk
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (size_t)p.n_chunks * w_chunk * 4);
    const unsigned wvo = (unsigned)(((4 * ct) * 64 + lane) * 4 * 4);  // cout tile 4 ct; the next tiles are 1024 bytes apart

    f32x4 a[PWD][RT], b[PWD][4];
    auto load = [&](int c, int d) {
        const bool s1 = c >= p.chunks0;
        unsigned so = (unsigned)(s1 ? c - p.chunks0 : c) * (PWCK * 4);
        if (sub_shift) {
            const unsigned sub = (unsigned)c >> (sub_shift - 1), cs = (unsigned)c & ((1u << (sub_shift - 1)) - 1u);
            so = (((sub >> 1) * 2u * p.Wo + (sub & 1u)) * (unsigned)p.C0 + cs * PWCK) * 4u;
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[d][rt] = bufload4(s1 ? rs_in1 : rs_in0, s1 ? avo1[rt] : avo0[rt], so);
        const unsigned wo = (unsigned)c * (unsigned)(w_chunk * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[d][t] = bufload4(rs_w, wvo + 1024u * t, wo);
    };

    const f32x4 z4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[RT][4];  // [row tile][cout tile]: register e of lane (n = l15, kq) = pixel 16 rt + 4 kq + e, cout 16 t + n
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rt][t] = z4;

#pragma unroll
    for (int d = 0; d < PWD; ++d) load(min(cb + d, ce - 1), d);
    for (int c = cb; c < ce; c += PWD) {
#pragma unroll
        for (int d = 0; d < PWD; ++d) {
            if (c + d < ce) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][rt][j], b[d][t][j], acc[rt][t], 0, 0, 0);
                if (c + d + PWD < ce) load(c + d + PWD, d);
            }
        }
    }

    // ---- epilogue: 32 pixel rows (two row tiles) at a time through this wave's staging tile
    const int rsub = lane >> 4;
    const int c4 = (lane & 15) * 4;
    const int cg = ct * 64 + c4;
    const bool cvalid = cg < p.Cout;
    RowsEpilogue re;
    re.split = split;
    re.M = (size_t)M;
    re.b0 = 0;
    re.uni = p.ss_stride == 0;
    re.HoWo = p.Ho * p.Wo;
    re.red = nullptr;
    re.rows_per_wg = 0;
    re.row_in_wg0 = 0;
    re.wn = 0;
    re.all_valid = true;
    float* T = smem + (size_t)wave * 32 * PWTS;
    constexpr int NRE = RT >= 2 ? 8 : 4;  // pixel rows per lane group and pass: 32 rows (two row tiles) or the wave's 16
#pragma unroll
    for (int r = 0; r < (RT + 1) / 2; ++r) {
        int pixv[NRE];
#pragma unroll
        for (int j = 0; j < NRE; ++j) {
            const unsigned px = px0 + 32u * r + 4u * j + (unsigned)rsub;
            pixv[j] = px < M ? (int)px : -1;
        }
        RowsPrefetch<NRE, true> pf;
        rows_prefetch<NRE, true>(p, re, pixv, cg, cvalid, pf);
#pragma unroll
        for (int h = 0; h < (RT >= 2 ? 2 : 1); ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float* dst = T + (16 * h + 4 * kq) * PWTS + 16 * t + l15;
                const f32x4 v4 = acc[2 * r + h][t];
                dst[0 * PWTS] = v4.x;
                dst[1 * PWTS] = v4.y;
                dst[2 * PWTS] = v4.z;
                dst[3 * PWTS] = v4.w;
            }
        __builtin_amdgcn_wave_barrier();
        f32x4 v[NRE];
#pragma unroll
        for (int j = 0; j < NRE; ++j) v[j] = *reinterpret_cast<const f32x4*>(T + (4 * j + rsub) * PWTS + c4);
        __builtin_amdgcn_wave_barrier();
        rows_epilogue<1, NRE, true>(p, re, v, pixv, cg, cvalid, pf);
    }
}

template <int WGN>
static int pw_launch_t(const ConvParams& p, int blocks, hipStream_t s) {
    const dim3 grid(blocks, p.geo.splits, 1);
    switch (p.geo.TH) {
        case 4: hipLaunchKernelGGL((pw_mfma_kernel<WGN, 4>), grid, dim3(256), p.geo.lds_bytes, s, p); break;
        case 2: hipLaunchKernelGGL((pw_mfma_kernel<WGN, 2>), grid, dim3(256), p.geo.lds_bytes, s, p); break;
        default: hipLaunchKernelGGL((pw_mfma_kernel<WGN, 1>), grid, dim3(256), p.geo.lds_bytes, s, p); break;
    }
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

int pw_launch(const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    p.stamps = nullptr;
    const ConvGeom& g = p.geo;
    DM_REQUIRE(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && !p.up && !p.fold, "pointwise: 1x1 only");
    if (p.s2d) {
        // space-to-depth: C0 = channels of the SOURCE, n_chunks = 4 * C0 / 16; the kernel takes log2(C0 / 16) + 1 in p.s2d
        const int cps = p.C0 / PWCK;
        DM_REQUIRE(p.C1 == 0 && cps > 0 && (cps & (cps - 1)) == 0 && p.n_chunks == 4 * cps && p.chunks0 == p.n_chunks,
                   "pointwise: space-to-depth mode needs one source with a power-of-two number of 16-channel chunks");
        DM_REQUIRE((size_t)4 * p.B * p.Ho * p.Wo * p.C0 < (1ull << 30) && (size_t)4 * p.B * p.Ho * p.Wo < (1u << 24),
                   "pointwise: space-to-depth source too large");
        int lg = 0;
        while ((1 << lg) < cps) ++lg;
        p.s2d = lg + 1;
    }
    DM_REQUIRE(!p.in_nchw && !p.out_nchw, "pointwise: NHWC only");
    DM_REQUIRE(p.C0 % PWCK == 0 && p.C1 % PWCK == 0 && p.Cout % 64 == 0, "pointwise: channel counts");
    DM_REQUIRE(p.Hin == p.Ho && p.Win == p.Wo, "pointwise: same-size convolution");
    const size_t M = (size_t)p.B * p.Ho * p.Wo;
    DM_REQUIRE(M > 0 && M < (1u << 24) && M * (size_t)std::max(p.Cout, std::max(p.C0, p.C1)) < (1ull << 30),
               "pointwise: tensor too large for 24-bit pixel indices");
    DM_REQUIRE((g.WN == 1 || g.WN == 2) && g.WM * g.WN == 4 && p.Cout % (64 * g.WN) == 0 &&
                   (g.TH == 4 || g.TH == 2 || g.TH == 1) && g.n_tiles_n == p.Cout / (64 * g.WN) &&
                   (size_t)g.tiles_x * 16 * g.TH * g.WM >= M && (size_t)(g.tiles_x - 1) * 16 * g.TH * g.WM < M,
               "pointwise: plan does not match the tensor");
    DM_REQUIRE(!(p.epi & EPI_NORM) || (p.Cout == 64 && g.splits == 1), "pointwise: fused RMSNorm needs all couts in one wave");
    DM_REQUIRE(g.splits == 1 || p.partial, "pointwise: split-K writes partial sums");
    DM_REQUIRE(p.s2d || (p.chunks0 == p.C0 / PWCK && p.n_chunks == (p.C0 + p.C1) / PWCK), "pointwise: chunk counts");
    DM_REQUIRE(g.splits * g.chunks_per_split >= p.n_chunks && (g.splits - 1) * g.chunks_per_split < p.n_chunks,
               "pointwise: K split does not cover the chunks");
    DM_REQUIRE(g.lds_bytes >= 4 * 32 * PWTS * 4, "pointwise: LDS size");
    const int blocks = g.n_tiles_n * g.tiles_x;
    static const bool xcd_order = env_int("DM_NO_XCD_ORDER", 0) == 0;
    p.geo.xcd_groups = (xcd_order && blocks % 8 == 0 && 8 % g.n_tiles_n == 0) ? 8 / g.n_tiles_n : 0;
    const bool timed = prof::enabled();
    if (timed) {
        const double pix = (double)M, cin = p.s2d ? 4.0 * p.C0 : p.C0 + p.C1;
        const double flops = 2.0 * cin * p.Cout * pix;
        const double res_rows = (!p.partial && (p.epi & EPI_RESIDUAL)) ? 1.0 : 0.0;  // the fused residual add reads one more tensor
        const double bytes = 4.0 * (cin * pix + (1.0 + res_rows) * p.Cout * pix + cin * p.Cout);
        char name[64];
        if (prof::detail())
            snprintf(name, sizeof(name), "pw<%d> 1x1 %d+%d->%d @%dx%d%s e%d k%d g%d r%d", g.WN, p.C0, p.C1, p.Cout, p.Ho,
                     p.Wo, p.s2d ? " s2d" : "", p.epi, g.splits, blocks * g.splits, g.TH);
        else
            snprintf(name, sizeof(name), "pw_mfma_kernel<%d>", g.WN);
        if (prof::begin(name, flops, bytes, s)) return 1;
    }
    const int rc = g.WN == 2 ? pw_launch_t<2>(p, blocks, s) : pw_launch_t<1>(p, blocks, s);
    if (rc) return 1;
    if (timed && prof::end(s)) return 1;
    return 0;
}

}  // namespace dm

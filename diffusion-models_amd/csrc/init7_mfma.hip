// The U-Net's first convolution: 7x7 / pad 3 over the NCHW input image, 3-8 channels -> 64 (init_conv,
// DD/denoising_diffusion.py:262; 4 channels for latents, 6 for the image-conditional variants).
//
// K = 49 Cin (147 for RGB) is too thin for the generic kernel's 16-channel chunks -- it pads every tap to 4 channels
// and re-stages windows per tap row.  Here one workgroup = one 16x16 block of output pixels x 64 couts: the whole
// 22x22xCin input window (6 KB) and the whole weight matrix (k x 64, 38 KB, k = (c, ky, kx) padded to a multiple of 4)
// sit in LDS, and the K loop is 37 steps of v_mfma_f32_16x16x4_f32 whose A operand is ONE scalar LDS read per lane
// (pixel (y, x) shifted by the tap of K index 4 s + kq) and whose B operand is one 16-byte LDS read (the lane's cout in
// each of the four 16-cout tiles).  Wave w owns rows 4w..4w+3 of the block (row tile = one image row of 16 pixels).
// Output: NHWC through the shared row epilogue (bias), 32 pixels at a time.
#include "conv_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace dm {

static constexpr int I7_WR = 22, I7_WS = 24;  // window rows, padded row stride (floats)
static constexpr int I7_TS = 68;              // row stride of the epilogue staging tile

bool init7_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up) {
    static const bool off = std::getenv("DM_NO_INIT7") != nullptr;
    return !off && KH == 7 && KW == 7 && stride == 1 && pad == 3 && !up && C1 == 0 && C0 >= 1 && C0 <= 8 && Cout == 64;
}

static inline int i7_ksteps(int Cin) { return (49 * Cin + 3) / 4; }
size_t init7_packed_floats(int Cin) { return (size_t)i7_ksteps(Cin) * 4 * 64; }

// oihw (64, Cin, 7, 7) -> [k = c*49 + ky*7 + kx, padded with zero rows][l15][t]: cout = 16 t + l15
void init7_pack_weights(const float* oihw, float* packed, int Cin) {
    const int K = 49 * Cin, KP = i7_ksteps(Cin) * 4;
    for (int k = 0; k < KP; ++k)
        for (int co = 0; co < 64; ++co)
            packed[((size_t)k * 16 + (co & 15)) * 4 + (co >> 4)] = k < K ? oihw[(size_t)co * K + k] : 0.f;
}

template <int NS>  // K steps of 4 (37 for 3 channels; upper bound for the unrolled loop, the real count is a kernel argument)
__global__ __launch_bounds__(256) void init7_mfma_kernel(const ConvParams p, int n_steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Cin = p.C0, H = p.Ho, W = p.Wo;
    float* wl = smem;                                // [n_steps * 4][16][4]
    float* xw = smem + (size_t)n_steps * 4 * 64;     // [Cin][22][24]
    float* T = xw;                                   // epilogue staging (the window is dead by then): 4 waves x [32][68]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = 16 * tx, y0 = 16 * ty;

    // ---- stage the weights (same for every workgroup: L2 hits) and the NCHW window (zero outside the image)
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.w);
        f32x4* dst = reinterpret_cast<f32x4*>(wl);
        for (int i = tid; i < n_steps * 4 * 16; i += 256) dst[i] = src[i];
        const float* xb = p.in0 + (size_t)b * Cin * H * W;
        const int n_win = Cin * I7_WR * I7_WS;
        for (int i = tid; i < n_win; i += 256) {
            const int c = i / (I7_WR * I7_WS), r = i - c * (I7_WR * I7_WS);
            const int wy = r / I7_WS, wx = r - wy * I7_WS;
            const int iy = y0 - 3 + wy, ix = x0 - 3 + wx;
            float v = 0.f;
            if (wx < I7_WR && iy >= 0 && iy < H && ix >= 0 && ix < W) v = xb[((size_t)c * H + iy) * W + ix];
            xw[i] = v;
        }
    }
    __syncthreads();

    // ---- K loop.  Lane (l15, kq), step s: K index k = 4 s + kq = (c, ky, kx); A = window[c][4 wave + rt + ky][l15 + kx]
    const f32x4 z4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rt][t] = z4;
    const int K = 49 * Cin;
    const float* xrow = xw + (4 * wave) * I7_WS + l15;
    const float* wrow = wl + (size_t)(kq * 16 + l15) * 4;
#pragma unroll 1
    for (int s0 = 0; s0 < n_steps; s0 += NS) {
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const int s = s0 + si;
            if (s < n_steps) {
                int k = min(4 * s + kq, K - 1);  // padded K rows have zero weights: any valid address will do
                const int c = k / 49;
                k -= 49 * c;
                const int ky = k / 7, kx = k - 7 * ky;
                const float* xa = xrow + (c * I7_WR + ky) * I7_WS + kx;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(wrow + (size_t)s * 256);
                float a[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) a[rt] = xa[rt * I7_WS];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], b4[t], acc[rt][t], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // the window becomes the staging area

    // ---- epilogue: accumulator register e of lane (n = l15, kq) = pixel (row 4 wave + rt, column 4 kq + e), cout 16 t + n
    const int rsub = lane >> 4;
    const int c4 = (lane & 15) * 4;
    RowsEpilogue re;
    re.split = 0;
    re.M = (size_t)p.B * H * W;
    re.b0 = b;
    re.uni = true;
    re.HoWo = H * W;
    re.red = nullptr;
    re.rows_per_wg = 0;
    re.row_in_wg0 = 0;
    re.wn = 0;
    re.all_valid = true;
    float* Tw = T + (size_t)wave * 32 * I7_TS;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        // staging row rho = 16 h + column: the lane group rsub finishes columns 4 j' + rsub
        int pixv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rho = 4 * j + rsub;  // 0..31: row tile h = rho >> 4, column = rho & 15
            const int y = y0 + 4 * wave + 2 * r + (rho >> 4), x = x0 + (rho & 15);
            pixv[j] = (y < H && x < W) ? (b * H + y) * W + x : -1;
        }
        RowsPrefetch<8, true> pf;
        rows_prefetch<8, true>(p, re, pixv, c4, true, pf);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float* dst = Tw + (16 * h + 4 * kq) * I7_TS + 16 * t + l15;
                const f32x4 v4 = acc[2 * r + h][t];
                dst[0 * I7_TS] = v4.x;
                dst[1 * I7_TS] = v4.y;
                dst[2 * I7_TS] = v4.z;
                dst[3 * I7_TS] = v4.w;
            }
        __builtin_amdgcn_wave_barrier();
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(Tw + (4 * j + rsub) * I7_TS + c4);
        __builtin_amdgcn_wave_barrier();
        rows_epilogue<1, 8, true>(p, re, v, pixv, c4, true, pf);
    }
}

// p: in0 = NCHW image (B, C0, Ho, Wo), w = init7_pack_weights, out = NHWC (B, Ho, Wo, 64), epi = EPI_BIAS or 0
int init7_launch(const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    p.stamps = nullptr;
    DM_REQUIRE(p.KH == 7 && p.KW == 7 && p.stride == 1 && p.pad == 3 && !p.up && !p.fold && !p.s2d, "init7: 7x7 s1 p3 only");
    DM_REQUIRE(p.in_nchw && !p.out_nchw && p.C1 == 0 && p.C0 >= 1 && p.C0 <= 8 && p.Cout == 64, "init7: NCHW image -> 64 NHWC");
    DM_REQUIRE(p.Hin == p.Ho && p.Win == p.Wo && !p.partial && (p.epi & ~EPI_BIAS) == 0, "init7: plain conv + bias");
    DM_REQUIRE((size_t)p.B * p.Ho * p.Wo < (1u << 24), "init7: tensor too large for 24-bit pixel indices");
    const int n_steps = i7_ksteps(p.C0);
    const size_t lds = ((size_t)n_steps * 4 * 64 + std::max((size_t)p.C0 * I7_WR * I7_WS, (size_t)4 * 32 * I7_TS)) * 4;
    DM_REQUIRE(lds <= 160 * 1024, "init7: LDS");
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(init7_mfma_kernel<8>), 1)) return 1;
    const int blocks = p.B * ((p.Ho + 15) / 16) * ((p.Wo + 15) / 16);
    const bool timed = prof::enabled();
    if (timed) {
        const double pix = (double)p.B * p.Ho * p.Wo;
        char name[64];
        if (prof::detail())
            snprintf(name, sizeof(name), "init7 7x7 %d->64 @%dx%d e%d", p.C0, p.Ho, p.Wo, p.epi);
        else
            snprintf(name, sizeof(name), "init7_mfma_kernel");
        if (prof::begin(name, 2.0 * 49.0 * p.C0 * 64.0 * pix, 4.0 * (p.C0 + 64.0) * pix, s)) return 1;
    }
    hipLaunchKernelGGL(init7_mfma_kernel<8>, dim3(blocks), dim3(256), lds, s, p, n_steps);
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    return 0;
}

}  // namespace dm

// EXPERIMENTAL, OFF BY DEFAULT (DM_WINOB=1): Winograd F(2x2, 3x3) with its fp32 products formed on the bf16 matrix cores.
//
// The convolution of winograd_mfma.hip (same algebra, same epilogue, same accumulator layout) whose 16 Winograd-domain
// GEMMs run on v_mfma_f32_32x32x16_bf16: every fp32 operand is split into three round-to-nearest bf16 terms,
// x = h + m + l (exact to 2^-27 |x|), and a product is the sum of the six partial products
// ah*bl + al*bh + am*bm + ah*bm + am*bh + ah*bh with fp32 accumulation (linattn_bf16x6.hip has the error analysis:
// 1.05e-6 rel-L2 on a K = 4608 GEMM against 1.21e-6 for the f32-input MFMA).  Six 32-cycle instructions cover a
// 32x32x16 step that takes eight 64-cycle f32 MFMAs, and -- unlike the f32-input MFMA -- they do not run on the vector
// ALUs.  tools/winob_microbench.hip measured the main loop below in isolation (profiles/r3_winob_microbench.txt).
//
// Mapping (one workgroup per CU = 4 waves = 64 Winograd tiles (a 16x16 block of output pixels) x 64 couts):
//   wave i owns row i of the transformed 4x4 patch; its accumulators are 4 columns j x 2 tile groups r x 2 cout groups q
//   (256 registers).  K runs in chunks of 16 input channels: lane (tile l31 [+32 r], half lh) holds channels 8 lh .. 8 lh + 7.
//   * B operand: U = G g G^T split on the host into three bf16 planes, packed [chunk][xi][plane][cout][16 channels];
//     lane (cout q*32 + l31, half lh) loads its 8 channels of a plane as one 16-byte buffer load.  Two register sets, the
//     six loads of the next patch column are issued at the head of the current column's MFMAs.
//   * A operand: the raw 18x18x16-channel window of the tile block is staged in LDS (double buffered, one barrier per
//     chunk).  T = d[ra] +- d[rb] for the four window columns of a tile is kept in registers for a whole chunk (64
//     registers); V[j] is formed from T and split into its three planes one channel pair at a time, in the gaps between the
//     MFMAs of the previous column (scalar fp32 instructions: packed ones cost several issue slots beside an MFMA).
// Restrictions (the caller falls back to the f32 kernels otherwise): even images of at least 16x16 pixels, C0 % 16 == 0,
// C1 % 16 == 0, Cout % 64 == 0; inference handles only (the training step re-packs weights on the device and has no
// packer for this layout).
#include "conv_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace dm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static constexpr int BCK = 16;               // input channels per K chunk
static constexpr int BTILES = 64;            // Winograd tiles per workgroup (8 x 8)
static constexpr int BIW = 18;               // window: 2 * 8 + 2 pixels per side
static constexpr int BRS = BIW * BCK + 8;    // floats per window row (+8: tile rows 16 banks apart)
static constexpr int BWIN = BIW * BRS;       // floats per window buffer
static constexpr int BHR = 6;                // 16-byte staging items per thread: 18 * 18 * 4 = 1296 <= 6 * 256
static constexpr int BWTS = 68;              // row stride of the epilogue tiles (as winograd_mfma.hip)
static constexpr int B1WIN = 10 * BRS;       // 32-tile form: 18 x 10 window
static constexpr int B1HR = 3;               // 18 * 10 * 4 = 720 items <= 3 * 256

bool winob_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up) {
    static const bool on = env_int("DM_WINOB", 0) != 0 && std::getenv("DM_NO_WINOGRAD") == nullptr;
    return on && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !up && C0 > 0 && C0 % BCK == 0 && C1 % BCK == 0 &&
           Cout % 64 == 0;
}

// packed size in floats (the buffer holds bf16): (C / 16) chunks x 16 xi x 3 planes x Cout x 16 channels x 2 bytes
size_t winob_packed_floats(int Cout, int C0, int C1) { return (size_t)(C0 + C1) * 16 * 3 * Cout / 2; }

static inline uint16_t b_f2bf(float x) {  // round to nearest even (finite values)
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float b_bf2f(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

void winob_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int Cin = C0 + C1;
    uint16_t* dst = reinterpret_cast<uint16_t*>(packed);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float* gk = oihw + ((size_t)co * Cin + ci) * 9;
            double Gg[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) Gg[i][b] = G[i][0] * gk[b] + G[i][1] * gk[3 + b] + G[i][2] * gk[6 + b];
            const int chunk = ci / BCK, cc = ci % BCK;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const float u = (float)(Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2]);
                    const uint16_t h = b_f2bf(u);
                    const float r1 = u - b_bf2f(h);
                    const uint16_t m = b_f2bf(r1);
                    const uint16_t l = b_f2bf(r1 - b_bf2f(m));
                    const size_t base = (((size_t)chunk * 16 + i * 4 + j) * 3) * Cout;
                    dst[((base + 0 * (size_t)Cout) + co) * BCK + cc] = h;
                    dst[((base + 1 * (size_t)Cout) + co) * BCK + cc] = m;
                    dst[((base + 2 * (size_t)Cout) + co) * BCK + cc] = l;
                }
        }
}

ConvGeom winob_plan(int B, int Ho, int Wo, int Cout, int C0, int C1) {
    // R = tile groups of 32 per workgroup: 1 -> 8 x 4 tiles, 128 accumulators, two workgroups per CU (default);
    //                                      2 -> 8 x 8 tiles, 256 accumulators, one workgroup per CU
    static const int R = env_int("DM_WINOB_R", 1) == 2 ? 2 : 1;
    const int tiles = 32 * R, TH = 4 * R, win = (2 * TH + 2) * BRS;
    ConvGeom g{};
    g.WM = R;
    g.WN = 1;
    g.CK = BCK;
    g.TW = 8;
    g.TH = TH;
    g.NB = 1;
    g.lTW = 3;
    g.lTH = R == 2 ? 3 : 2;
    g.tiles_x = (Wo / 2 + 7) / 8;
    g.tiles_y = (Ho / 2 + TH - 1) / TH;
    g.groups = B;
    g.n_tiles_n = Cout / 64;
    g.IH = 2 * TH + 2;
    g.IW = BIW;
    g.row_stride = BRS;
    g.halo_floats = win;
    g.TPS = 3;
    g.splits = 1;
    g.chunks_per_split = (C0 + C1) / BCK;
    g.fused_norm = g.n_tiles_n == 1;
    g.w_floats = 0;
    // two windows + a scratch slot; the epilogue reuses the space for 4 x 2 transposed tiles; behind both the pixel table
    g.ptab_off = std::max(2 * win + 8, 4 * 2 * tiles * BWTS);
    g.lds_bytes = (g.ptab_off + 4 * tiles) * 4;
    return g;
}

bool winob_shape_ok(int B, int Ho, int Wo, int Cout, int C0, int C1) {
    if ((Ho | Wo) & 1) return false;
    if (Ho < 16 || Wo < 16) return false;
    static const int min_wgs = env_int("DM_WINOB_MIN_WGS", 128);
    const ConvGeom g = winob_plan(B, Ho, Wo, Cout, C0, C1);
    const int wgs = g.tiles_x * g.tiles_y * g.groups * g.n_tiles_n;
    return wgs >= min_wgs && g.lds_bytes <= 160 * 1024 && (size_t)B * Ho * Wo < (1u << 24) &&
           (size_t)B * Ho * Wo * std::max(C0, C1) < (1ull << 30);
}

__device__ __forceinline__ unsigned b_pk_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(256, 1) void winob_mfma_kernel(const ConvParams p) {
    constexpr int TILES = BTILES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = p.geo;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    int n_tile, bid;
    block_to_tile(g, blockIdx.x, gridDim.x, n_tile, bid);
    const int tile_x = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int tile_y = bid % g.tiles_y;
    const int b0 = bid / g.tiles_y;
    const int tx0 = tile_x * 8, ty0 = tile_y * 8;    // in Winograd tiles
    const int ix0 = 2 * tx0 - 1, iy0 = 2 * ty0 - 1;  // window origin in pixels
    const int cb = 0, ce = p.n_chunks;
    float* raw[2] = {smem, smem + BWIN};
    DM_STAMP_DECL
    DM_STAMP(0);

    // ---- window staging: item = (window pixel, channel quad); quad q of pixel column hx sits in slot q ^ ((hx >> 2) & 3)
    int hpix[BHR], hoff[BHR];
#pragma unroll
    for (int i = 0; i < BHR; ++i) {
        const int it = tid + 256 * i;
        hpix[i] = -1;
        hoff[i] = 2 * BWIN;  // items past the window go to a scratch slot
        if (it < BIW * BIW * 4) {
            const int hp = it >> 2, qd = it & 3;
            const int hy = hp / BIW, hx = hp - hy * BIW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const int off = hy * BRS + hx * BCK + 4 * (qd ^ ((hx >> 2) & 3));
            if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
                hpix[i] = (b0 * p.Hin + iy) * p.Win + ix;
                hoff[i] = off;
            } else {
                *reinterpret_cast<f32x4*>(raw[0] + off) = make_f32x4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<f32x4*>(raw[1] + off) = make_f32x4(0.f, 0.f, 0.f, 0.f);
            }
        }
        hpix[i] = max(hpix[i], 0);
    }
    {
        // output pixel of (tile, a, b), or -1: thread = tile * 4 + (2a + b)  (layout of winograd_mfma.hip, NRT = 16)
        constexpr int NRT = 16;
        int* ptab = reinterpret_cast<int*>(smem + g.ptab_off);
        const int t = tid >> 2, ab = tid & 3;
        const int tx = t & 7, ty = (t >> 3) & 7;
        const int y = 2 * (ty0 + ty) + (ab >> 1), x = 2 * (tx0 + tx) + (ab & 1);
        ptab[(t / NRT) * (4 * NRT) + ab * NRT + (t % NRT)] = (y < p.Ho && x < p.Wo) ? (b0 * p.Ho + y) * p.Wo + x : -1;
    }
    f32x4 hreg[BHR];
    const size_t in_px = (size_t)p.B * p.Hin * p.Win;
    const __amdgpu_buffer_rsrc_t rs_in0 = make_rsrc(p.in0, in_px * p.C0 * 4);
    const __amdgpu_buffer_rsrc_t rs_in1 = make_rsrc(p.C1 ? p.in1 : p.in0, in_px * (p.C1 ? p.C1 : p.C0) * 4);
    const unsigned hq = 4 * (tid & 3);
    unsigned hvo[BHR];
    auto window_offsets = [&](unsigned Cs) {
#pragma unroll
        for (int i = 0; i < BHR; ++i) hvo[i] = (__umul24((unsigned)hpix[i], Cs) + hq) * 4;
    };
    auto window_value = [&](int chunk, int i) {
        const bool s1 = chunk >= p.chunks0;
        return bufload4(s1 ? rs_in1 : rs_in0, hvo[i], (unsigned)(s1 ? chunk - p.chunks0 : chunk) * (BCK * 4));
    };

    // ---- input transform of this lane: tiles l31 and 32 + l31, channels 8 lh .. 8 lh + 7, row `wave` of B^T d
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const int tx = l31 & 7, ty = l31 >> 3;
    int colq[4][2];  // float offset of (column b, quad 2 lh + k) of this lane's tiles inside a window row
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int hx = 2 * tx + b;
            colq[b][k] = hx * BCK + 4 * ((2 * lh + k) ^ ((hx >> 2) & 3));
        }
    const int rowa = ra * BRS, rowb = rb * BRS;
    const int rbase[2] = {2 * ty * BRS, 2 * (ty + 4) * BRS};

    // ---- B planes: [chunk][xi][plane][cout][16] bf16; per-lane byte offset once, the rest is scalar
    const size_t w_bytes = (size_t)p.n_chunks * 16 * 3 * p.Cout * 32;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, w_bytes);
    const unsigned bvo = (unsigned)((n_tile * 64 + l31) * 2 + lh) * 16;
    const unsigned plane_b = (unsigned)p.Cout * 32;  // bytes per (xi, plane)
    const unsigned chunk_b = 16 * 3 * plane_b;
    const unsigned bwave = (unsigned)(4 * wave) * 3 * plane_b;
    auto bload = [&](int chunk, int j, int pl, int q) {
        const unsigned so = (unsigned)chunk * chunk_b + bwave + (unsigned)(j * 3 + pl) * plane_b + (unsigned)q * (32 * 32);
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)bvo, (int)so, 0));
    };

    f32x16 acc[4][2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][r][q][e] = 0.f;
    u32x4 Ah[2][2], Am[2][2], Al[2][2];  // [set][r]: the three planes of the A operand (8 bf16 each)
    bf16x8 Bp[2][3][2];                  // [set][plane][q]
    f32x4 T[2][4][2];                    // [r][window column b][channel quad k]
    constexpr int LAT = 3;               // MFMA gaps between an LDS read and its use
    f32x4 tq[LAT];
    float rs0 = 0.f, rs1 = 0.f;

    // one channel pair (quad k, half h2) of A[set][r] for patch column jn in three stages (one per MFMA gap):
    // 0: combine + high plane, 1: middle plane, 2: low plane; the residuals wait in rs0 / rs1 in between
    auto a_stage = [&](int set, int jn, int r, int k, int h2, int st) {
        if (st == 0) {
            const int e0 = 2 * h2, e1 = 2 * h2 + 1;
            const int ca = jn == 0 ? 0 : (jn == 2 ? 2 : 1), cc = jn == 0 ? 2 : (jn == 1 ? 2 : (jn == 2 ? 1 : 3));
            const float x0 = jn == 1 ? T[r][ca][k][e0] + T[r][cc][k][e0] : T[r][ca][k][e0] - T[r][cc][k][e0];
            const float x1 = jn == 1 ? T[r][ca][k][e1] + T[r][cc][k][e1] : T[r][ca][k][e1] - T[r][cc][k][e1];
            const unsigned h = b_pk_bf16(x0, x1);
            Ah[set][r][2 * k + h2] = h;
            rs0 = x0 - __builtin_bit_cast(float, h << 16);
            rs1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
        } else if (st == 1) {
            const unsigned mm = b_pk_bf16(rs0, rs1);
            Am[set][r][2 * k + h2] = mm;
            rs0 = rs0 - __builtin_bit_cast(float, mm << 16);
            rs1 = rs1 - __builtin_bit_cast(float, mm & 0xffff0000u);
        } else {
            Al[set][r][2 * k + h2] = b_pk_bf16(rs0, rs1);
        }
    };
    auto a_micro = [&](int set, int jn, int r, int u) { a_stage(set, jn, r, (u / 3) >> 1, (u / 3) & 1, u % 3); };
    auto t_issue = [&](const float* win, int r, int i) {  // item i = 2 b + k: row a lands in T, row b in tq
        const float* base = win + rbase[r];
        T[r][i >> 1][i & 1] = *reinterpret_cast<const f32x4*>(base + rowa + colq[i >> 1][i & 1]);
        tq[i % LAT] = *reinterpret_cast<const f32x4*>(base + rowb + colq[i >> 1][i & 1]);
    };
    auto t_finish = [&](int r, int i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) T[r][i >> 1][i & 1][e] = __builtin_fmaf(sgn, tq[i % LAT][e], T[r][i >> 1][i & 1][e]);
    };

    DM_STAMP_ADD(4)
    // ---- prologue: windows of chunks cb and cb + 1 -> LDS, T / A / B of phase (cb, 0) -> registers
    {
        const bool two = cb + 1 < ce;
        f32x4 h2[BHR];
        window_offsets(cb >= p.chunks0 ? p.C1 : p.C0);
#pragma unroll
        for (int i = 0; i < BHR; ++i) hreg[i] = window_value(cb, i);
        const int c1 = two ? cb + 1 : cb;
        if (c1 == p.chunks0) window_offsets(p.C1);
#pragma unroll
        for (int i = 0; i < BHR; ++i) h2[i] = window_value(c1, i);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < 2; ++q) Bp[0][pl][q] = bload(cb, 0, pl, q);
#pragma unroll
        for (int i = 0; i < BHR; ++i) *reinterpret_cast<f32x4*>(raw[0] + hoff[i]) = hreg[i];
#pragma unroll
        for (int i = 0; i < BHR; ++i) *reinterpret_cast<f32x4*>(raw[1] + hoff[i]) = h2[i];
    }
    DM_STAMP_ADD(5)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* base = raw[0] + rbase[r];
            const f32x4 da = *reinterpret_cast<const f32x4*>(base + rowa + colq[i >> 1][i & 1]);
            const f32x4 db = *reinterpret_cast<const f32x4*>(base + rowb + colq[i >> 1][i & 1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) T[r][i >> 1][i & 1][e] = __builtin_fmaf(sgn, db[e], da[e]);
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u = 0; u < 12; ++u) a_micro(0, 0, r, u);
    __syncthreads();  // raw[0] is overwritten with chunk cb + 2 during the first iteration
    DM_STAMP_ADD(0)

    // ---- main loop: phase (c, j) = the 24 MFMAs of patch column j of chunk c; everything else sits in hooks between
    // them (one basic block per chunk, sched_barrier pins every hook).  Loads past the last chunk re-read valid memory.
    auto chunk_body = [&](int c, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;  // (c - cb) & 1: the buffer chunk c was read from
        const bool has1 = c + 1 < ce, has2 = c + 2 < ce;
        const int cn = has1 ? c + 1 : c;   // chunk whose B planes / window are consumed next
        const int cw = has2 ? c + 2 : c;   // chunk whose window is fetched now (c again at the end: never read)
        if (cw == p.chunks0 && p.C1 != p.C0) {
            window_offsets(p.C1);
            asm volatile("" ::: "memory");
        }
        const float* wnext = raw[PAR ^ 1];  // window of chunk c + 1
        float* wstore = raw[PAR];           // free: window of chunk c was consumed during chunk c - 1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int set = j & 1, nset = set ^ 1, jn = (j + 1) & 3;
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                const int pr = m >> 2, r = (m >> 1) & 1, q = m & 1;
                // small terms first: ah*bl, al*bh, am*bm, ah*bm, am*bh, ah*bh
                const u32x4 au = pr == 0 || pr == 3 || pr == 5 ? Ah[set][r] : (pr == 1 ? Al[set][r] : Am[set][r]);
                const bf16x8 bb = pr == 0 ? Bp[set][2][q] : (pr == 1 || pr == 4 || pr == 5 ? Bp[set][0][q] : Bp[set][1][q]);
                acc[j][r][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au), bb, acc[j][r][q], 0, 0, 0);
                // ---- hooks
                if (m < 6) Bp[nset][m >> 1][m & 1] = bload(j == 3 ? cn : c, jn, m >> 1, m & 1);
                if (j < 2) {
                    a_micro(nset, jn, m / 12, m % 12);  // A of the next column from the current T
                    if (j == 0 && m >= 12 && m < 12 + BHR) hreg[m - 12] = window_value(cw, m - 12);
                    if (j == 1 && m >= 12 && m < 12 + BHR) *reinterpret_cast<f32x4*>(wstore + hoff[m - 12]) = hreg[m - 12];
                } else if (j == 2) {
                    // A[3] of this chunk (tile group 0, then 1); T of the next chunk for group 0 once its A[3] is done
                    a_micro(nset, 3, m / 12, m % 12);
                    if (m >= 12 + LAT && m < 20 + LAT) t_finish(0, m - 12 - LAT);  // before the issue that reuses its tq slot
                    if (m >= 12 && m < 20) t_issue(wnext, 0, m - 12);
                } else {
                    // T of the next chunk for group 1 beside A[0] of the next chunk (group 0 first)
                    if (m >= LAT && m < 8 + LAT) t_finish(1, m - LAT);
                    if (m < 8) t_issue(wnext, 1, m);
                    a_micro(nset, 0, m / 12, m % 12);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        int c = cb;
        for (; c + 1 < ce; c += 2) {
            chunk_body(c, P0{});
            chunk_body(c + 1, P1{});
        }
        if (c < ce) chunk_body(c, P0{});
    }

    DM_STAMP_ADD(1)
    // ---- epilogue (winograd_mfma.hip, R = 2): R_i[b] = sum_j M[i][j] A[j][b] per wave, Y[a][b] = sum_i A^T[a][i] R_i[b]
    // through LDS, then the shared Block epilogue.
    constexpr int NR = 16;
    const int rsub = lane >> 4;
    const int oa = rsub >> 1, ob = rsub & 1;
    const int c4 = (lane & 15) * 4;
    const int cg = n_tile * 64 + c4;
    const bool cvalid = cg < p.Cout;
    int pixv[NR];
    {
        const int* pt = reinterpret_cast<const int*>(smem + g.ptab_off) + wave * (4 * NR) + rsub * NR;
#pragma unroll
        for (int jj = 0; jj < NR; jj += 4) {
            const int4 t4 = *reinterpret_cast<const int4*>(pt + jj);
            pixv[jj] = t4.x;
            pixv[jj + 1] = t4.y;
            pixv[jj + 2] = t4.z;
            pixv[jj + 3] = t4.w;
        }
    }
    RowsEpilogue re;
    re.split = 0;
    re.M = (size_t)p.B * p.Ho * p.Wo;
    re.b0 = b0;
    re.uni = true;
    re.HoWo = p.Ho * p.Wo;
    re.red = nullptr;
    re.rows_per_wg = 4 * TILES;
    re.row_in_wg0 = wave * 4 * NR;
    re.wn = 0;
    re.all_valid = true;
    RowsPrefetch<NR, true> pf;
    rows_prefetch<NR, true>(p, re, pixv, cg, cvalid, pf);

    float* Tb = smem + wave * (2 * TILES * BWTS);  // [b][tile][WTS]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 a0 = {acc[0][r][q][e], acc[0][r][q][e + 1]}, a1 = {acc[1][r][q][e], acc[1][r][q][e + 1]};
                const f32x2 a2 = {acc[2][r][q][e], acc[2][r][q][e + 1]}, a3 = {acc[3][r][q][e], acc[3][r][q][e + 1]};
                const f32x2 r0 = pk_add(pk_add(a0, a1), a2);
                const f32x2 r1 = pk_sub(pk_sub(a1, a2), a3);
                const int row = r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                Tb[row * BWTS + q * 32 + l31] = r0.x;
                Tb[(row + 1) * BWTS + q * 32 + l31] = r0.y;
                Tb[(TILES + row) * BWTS + q * 32 + l31] = r1.x;
                Tb[(TILES + row + 1) * BWTS + q * 32 + l31] = r1.y;
            }
        }
    __syncthreads();
    DM_STAMP_ADD(2)
    const float ysgn = oa ? -1.0f : 1.0f;  // Y[0] = R0 + R1 + R2,  Y[1] = R1 - R2 - R3
    const float* Y0 = smem + (oa * 2 + ob) * (TILES * BWTS) + c4;
    f32x4 v[NR];
#pragma unroll
    for (int jj = 0; jj < NR; ++jj) {
        const float* yp = Y0 + (NR * wave + jj) * BWTS;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(yp);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(yp + 2 * TILES * BWTS);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(yp + 4 * TILES * BWTS);
        v[jj] = a0 + ysgn * a1 + ysgn * a2;
    }
    rows_epilogue<1, NR, true>(p, re, v, pixv, cg, cvalid, pf);
    DM_STAMP_ADD(3)
    DM_STAMP_FLUSH
}

// The 32-tile form: 8 x 4 tiles per workgroup, 128 accumulator registers, TWO workgroups per CU -- one wave's vector
// instructions issue while the other's MFMAs run, and a workgroup's prologue / epilogue overlap its neighbour's loop; a
// weight plane serves 32 tiles (twice the B traffic per MFMA of the 64-tile form).  tools/winob_microbench.hip: 234 vs 212
// TFLOP/s fp32-equivalent in the loop.
__global__ __launch_bounds__(256, 2) void winob1_mfma_kernel(const ConvParams p) {
    constexpr int TILES = 32;
    constexpr int WINF = B1WIN, HRN = B1HR, IHN = 10;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = p.geo;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    int n_tile, bid;
    block_to_tile(g, blockIdx.x, gridDim.x, n_tile, bid);
    const int tile_x = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int tile_y = bid % g.tiles_y;
    const int b0 = bid / g.tiles_y;
    const int tx0 = tile_x * 8, ty0 = tile_y * 4;    // in Winograd tiles
    const int ix0 = 2 * tx0 - 1, iy0 = 2 * ty0 - 1;  // window origin in pixels
    const int cb = 0, ce = p.n_chunks;
    float* raw[2] = {smem, smem + WINF};
    DM_STAMP_DECL
    DM_STAMP(0);

    // ---- window staging: item = (window pixel, channel quad); quad q of pixel column hx sits in slot q ^ ((hx >> 2) & 3)
    int hpix[HRN], hoff[HRN];
#pragma unroll
    for (int i = 0; i < HRN; ++i) {
        const int it = tid + 256 * i;
        hpix[i] = -1;
        hoff[i] = 2 * WINF;  // items past the window go to a scratch slot
        if (it < BIW * IHN * 4) {
            const int hp = it >> 2, qd = it & 3;
            const int hy = hp / BIW, hx = hp - hy * BIW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const int off = hy * BRS + hx * BCK + 4 * (qd ^ ((hx >> 2) & 3));
            if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
                hpix[i] = (b0 * p.Hin + iy) * p.Win + ix;
                hoff[i] = off;
            } else {
                *reinterpret_cast<f32x4*>(raw[0] + off) = make_f32x4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<f32x4*>(raw[1] + off) = make_f32x4(0.f, 0.f, 0.f, 0.f);
            }
        }
        hpix[i] = max(hpix[i], 0);
    }
    {
        // output pixel of (tile, a, b), or -1: thread = tile * 4 + (2a + b)  (layout of winograd_mfma.hip, NRT = 16)
        constexpr int NRT = 8;
        int* ptab = reinterpret_cast<int*>(smem + g.ptab_off);
        if (tid < 4 * TILES) {
            const int t = tid >> 2, ab = tid & 3;
            const int tx = t & 7, ty = (t >> 3) & 3;
            const int y = 2 * (ty0 + ty) + (ab >> 1), x = 2 * (tx0 + tx) + (ab & 1);
            ptab[(t / NRT) * (4 * NRT) + ab * NRT + (t % NRT)] = (y < p.Ho && x < p.Wo) ? (b0 * p.Ho + y) * p.Wo + x : -1;
        }
    }
    f32x4 hreg[HRN];
    const size_t in_px = (size_t)p.B * p.Hin * p.Win;
    const __amdgpu_buffer_rsrc_t rs_in0 = make_rsrc(p.in0, in_px * p.C0 * 4);
    const __amdgpu_buffer_rsrc_t rs_in1 = make_rsrc(p.C1 ? p.in1 : p.in0, in_px * (p.C1 ? p.C1 : p.C0) * 4);
    const unsigned hq = 4 * (tid & 3);
    unsigned hvo[HRN];
    auto window_offsets = [&](unsigned Cs) {
#pragma unroll
        for (int i = 0; i < HRN; ++i) hvo[i] = (__umul24((unsigned)hpix[i], Cs) + hq) * 4;
    };
    auto window_value = [&](int chunk, int i) {
        const bool s1 = chunk >= p.chunks0;
        return bufload4(s1 ? rs_in1 : rs_in0, hvo[i], (unsigned)(s1 ? chunk - p.chunks0 : chunk) * (BCK * 4));
    };

    // ---- input transform of this lane: tile l31, channels 8 lh .. 8 lh + 7, row `wave` of B^T d
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const int tx = l31 & 7, ty = l31 >> 3;
    int colq[4][2];  // float offset of (column b, quad 2 lh + k) of this lane's tiles inside a window row
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int hx = 2 * tx + b;
            colq[b][k] = hx * BCK + 4 * ((2 * lh + k) ^ ((hx >> 2) & 3));
        }
    const int rowa = ra * BRS, rowb = rb * BRS;
    const int rbase0 = 2 * ty * BRS;

    // ---- B planes: [chunk][xi][plane][cout][16] bf16; per-lane byte offset once, the rest is scalar
    const size_t w_bytes = (size_t)p.n_chunks * 16 * 3 * p.Cout * 32;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, w_bytes);
    const unsigned bvo = (unsigned)((n_tile * 64 + l31) * 2 + lh) * 16;
    const unsigned plane_b = (unsigned)p.Cout * 32;  // bytes per (xi, plane)
    const unsigned chunk_b = 16 * 3 * plane_b;
    const unsigned bwave = (unsigned)(4 * wave) * 3 * plane_b;
    auto bload = [&](int chunk, int j, int pl, int q) {
        const unsigned so = (unsigned)chunk * chunk_b + bwave + (unsigned)(j * 3 + pl) * plane_b + (unsigned)q * (32 * 32);
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)bvo, (int)so, 0));
    };

    f32x16 acc[4][1][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 1; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][r][q][e] = 0.f;
    u32x4 Ah[2][1], Am[2][1], Al[2][1];  // [set][0]: the three planes of the A operand (8 bf16 each)
    bf16x8 Bp[2][3][2];                  // [set][plane][q]
    f32x4 T[1][4][2];                    // [0][window column b][channel quad k]
    constexpr int LAT = 2;               // MFMA gaps between an LDS read and its use
    f32x4 tq[LAT];
    float rs0 = 0.f, rs1 = 0.f;

    // one channel pair (quad k, half h2) of A[set][r] for patch column jn in three stages (one per MFMA gap):
    // 0: combine + high plane, 1: middle plane, 2: low plane; the residuals wait in rs0 / rs1 in between
    auto a_stage = [&](int set, int jn, int r, int k, int h2, int st) {
        if (st == 0) {
            const int e0 = 2 * h2, e1 = 2 * h2 + 1;
            const int ca = jn == 0 ? 0 : (jn == 2 ? 2 : 1), cc = jn == 0 ? 2 : (jn == 1 ? 2 : (jn == 2 ? 1 : 3));
            const float x0 = jn == 1 ? T[r][ca][k][e0] + T[r][cc][k][e0] : T[r][ca][k][e0] - T[r][cc][k][e0];
            const float x1 = jn == 1 ? T[r][ca][k][e1] + T[r][cc][k][e1] : T[r][ca][k][e1] - T[r][cc][k][e1];
            const unsigned h = b_pk_bf16(x0, x1);
            Ah[set][r][2 * k + h2] = h;
            rs0 = x0 - __builtin_bit_cast(float, h << 16);
            rs1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
        } else if (st == 1) {
            const unsigned mm = b_pk_bf16(rs0, rs1);
            Am[set][r][2 * k + h2] = mm;
            rs0 = rs0 - __builtin_bit_cast(float, mm << 16);
            rs1 = rs1 - __builtin_bit_cast(float, mm & 0xffff0000u);
        } else {
            Al[set][r][2 * k + h2] = b_pk_bf16(rs0, rs1);
        }
    };
    auto a_micro = [&](int set, int jn, int r, int u) { a_stage(set, jn, r, (u / 3) >> 1, (u / 3) & 1, u % 3); };
    auto t_issue = [&](const float* win, int r, int i) {  // item i = 2 b + k: row a lands in T, row b in tq
        const float* base = win + rbase0;
        T[r][i >> 1][i & 1] = *reinterpret_cast<const f32x4*>(base + rowa + colq[i >> 1][i & 1]);
        tq[i % LAT] = *reinterpret_cast<const f32x4*>(base + rowb + colq[i >> 1][i & 1]);
    };
    auto t_finish = [&](int r, int i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) T[r][i >> 1][i & 1][e] = __builtin_fmaf(sgn, tq[i % LAT][e], T[r][i >> 1][i & 1][e]);
    };

    DM_STAMP_ADD(4)
    // ---- prologue: windows of chunks cb and cb + 1 -> LDS, T / A / B of phase (cb, 0) -> registers
    {
        const bool two = cb + 1 < ce;
        f32x4 h2[HRN];
        window_offsets(cb >= p.chunks0 ? p.C1 : p.C0);
#pragma unroll
        for (int i = 0; i < HRN; ++i) hreg[i] = window_value(cb, i);
        const int c1 = two ? cb + 1 : cb;
        if (c1 == p.chunks0) window_offsets(p.C1);
#pragma unroll
        for (int i = 0; i < HRN; ++i) h2[i] = window_value(c1, i);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < 2; ++q) Bp[0][pl][q] = bload(cb, 0, pl, q);
#pragma unroll
        for (int i = 0; i < HRN; ++i) *reinterpret_cast<f32x4*>(raw[0] + hoff[i]) = hreg[i];
#pragma unroll
        for (int i = 0; i < HRN; ++i) *reinterpret_cast<f32x4*>(raw[1] + hoff[i]) = h2[i];
    }
    DM_STAMP_ADD(5)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 1; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* base = raw[0] + rbase0;
            const f32x4 da = *reinterpret_cast<const f32x4*>(base + rowa + colq[i >> 1][i & 1]);
            const f32x4 db = *reinterpret_cast<const f32x4*>(base + rowb + colq[i >> 1][i & 1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) T[r][i >> 1][i & 1][e] = __builtin_fmaf(sgn, db[e], da[e]);
        }
#pragma unroll
    for (int r = 0; r < 1; ++r)
#pragma unroll
        for (int u = 0; u < 12; ++u) a_micro(0, 0, r, u);
    __syncthreads();  // raw[0] is overwritten with chunk cb + 2 during the first iteration
    DM_STAMP_ADD(0)

    // ---- main loop: phase (c, j) = the 24 MFMAs of patch column j of chunk c; everything else sits in hooks between
    // them (one basic block per chunk, sched_barrier pins every hook).  Loads past the last chunk re-read valid memory.
    auto chunk_body = [&](int c, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;  // (c - cb) & 1: the buffer chunk c was read from
        const bool has1 = c + 1 < ce, has2 = c + 2 < ce;
        const int cn = has1 ? c + 1 : c;   // chunk whose B planes / window are consumed next
        const int cw = has2 ? c + 2 : c;   // chunk whose window is fetched now (c again at the end: never read)
        if (cw == p.chunks0 && p.C1 != p.C0) {
            window_offsets(p.C1);
            asm volatile("" ::: "memory");
        }
        const float* wnext = raw[PAR ^ 1];  // window of chunk c + 1
        float* wstore = raw[PAR];           // free: window of chunk c was consumed during chunk c - 1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int set = j & 1, nset = set ^ 1, jn = (j + 1) & 3;
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int pr = m >> 1, q = m & 1;
                // small terms first: ah*bl, al*bh, am*bm, ah*bm, am*bh, ah*bh
                const u32x4 au = pr == 0 || pr == 3 || pr == 5 ? Ah[set][0] : (pr == 1 ? Al[set][0] : Am[set][0]);
                const bf16x8 bb = pr == 0 ? Bp[set][2][q] : (pr == 1 || pr == 4 || pr == 5 ? Bp[set][0][q] : Bp[set][1][q]);
                acc[j][0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au), bb, acc[j][0][q], 0, 0, 0);
                // ---- hooks
                if (m < 6) Bp[nset][m >> 1][m & 1] = bload(j == 3 ? cn : c, jn, m >> 1, m & 1);
                a_micro(nset, jn, 0, m);  // A of the next column: 12 micro-ops, one per gap
                if (j == 0 && m >= 6 && m < 6 + HRN) hreg[m - 6] = window_value(cw, m - 6);
                if (j == 1 && m >= 6 && m < 6 + HRN) *reinterpret_cast<f32x4*>(wstore + hoff[m - 6]) = hreg[m - 6];
                // T of the next chunk: columns 0 and 2 are dead once A[2] exists (after phase 1), columns 1 and 3 once A[3]
                // does (after phase 2); item i = 2 b + k
                if (j == 2) {
                    if (m >= LAT && m < 4 + LAT) t_finish(0, 4 * ((m - LAT) >> 1) + ((m - LAT) & 1));
                    if (m < 4) t_issue(wnext, 0, 4 * (m >> 1) + (m & 1));
                }
                if (j == 3) {
                    if (m >= LAT && m < 4 + LAT) t_finish(0, 4 * ((m - LAT) >> 1) + 2 + ((m - LAT) & 1));
                    if (m < 4) t_issue(wnext, 0, 4 * (m >> 1) + 2 + (m & 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        int c = cb;
        for (; c + 1 < ce; c += 2) {
            chunk_body(c, P0{});
            chunk_body(c + 1, P1{});
        }
        if (c < ce) chunk_body(c, P0{});
    }

    DM_STAMP_ADD(1)
    // ---- epilogue (winograd_mfma.hip, R = 2): R_i[b] = sum_j M[i][j] A[j][b] per wave, Y[a][b] = sum_i A^T[a][i] R_i[b]
    // through LDS, then the shared Block epilogue.
    constexpr int NR = 8;
    const int rsub = lane >> 4;
    const int oa = rsub >> 1, ob = rsub & 1;
    const int c4 = (lane & 15) * 4;
    const int cg = n_tile * 64 + c4;
    const bool cvalid = cg < p.Cout;
    int pixv[NR];
    {
        const int* pt = reinterpret_cast<const int*>(smem + g.ptab_off) + wave * (4 * NR) + rsub * NR;
#pragma unroll
        for (int jj = 0; jj < NR; jj += 4) {
            const int4 t4 = *reinterpret_cast<const int4*>(pt + jj);
            pixv[jj] = t4.x;
            pixv[jj + 1] = t4.y;
            pixv[jj + 2] = t4.z;
            pixv[jj + 3] = t4.w;
        }
    }
    RowsEpilogue re;
    re.split = 0;
    re.M = (size_t)p.B * p.Ho * p.Wo;
    re.b0 = b0;
    re.uni = true;
    re.HoWo = p.Ho * p.Wo;
    re.red = nullptr;
    re.rows_per_wg = 4 * TILES;
    re.row_in_wg0 = wave * 4 * NR;
    re.wn = 0;
    re.all_valid = true;
    RowsPrefetch<NR, true> pf;
    rows_prefetch<NR, true>(p, re, pixv, cg, cvalid, pf);

    float* Tb = smem + wave * (2 * TILES * BWTS);  // [b][tile][WTS]
#pragma unroll
    for (int r = 0; r < 1; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 a0 = {acc[0][r][q][e], acc[0][r][q][e + 1]}, a1 = {acc[1][r][q][e], acc[1][r][q][e + 1]};
                const f32x2 a2 = {acc[2][r][q][e], acc[2][r][q][e + 1]}, a3 = {acc[3][r][q][e], acc[3][r][q][e + 1]};
                const f32x2 r0 = pk_add(pk_add(a0, a1), a2);
                const f32x2 r1 = pk_sub(pk_sub(a1, a2), a3);
                const int row = r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                Tb[row * BWTS + q * 32 + l31] = r0.x;
                Tb[(row + 1) * BWTS + q * 32 + l31] = r0.y;
                Tb[(TILES + row) * BWTS + q * 32 + l31] = r1.x;
                Tb[(TILES + row + 1) * BWTS + q * 32 + l31] = r1.y;
            }
        }
    __syncthreads();
    DM_STAMP_ADD(2)
    const float ysgn = oa ? -1.0f : 1.0f;  // Y[0] = R0 + R1 + R2,  Y[1] = R1 - R2 - R3
    const float* Y0 = smem + (oa * 2 + ob) * (TILES * BWTS) + c4;
    f32x4 v[NR];
#pragma unroll
    for (int jj = 0; jj < NR; ++jj) {
        const float* yp = Y0 + (NR * wave + jj) * BWTS;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(yp);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(yp + 2 * TILES * BWTS);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(yp + 4 * TILES * BWTS);
        v[jj] = a0 + ysgn * a1 + ysgn * a2;
    }
    rows_epilogue<1, NR, true>(p, re, v, pixv, cg, cvalid, pf);
    DM_STAMP_ADD(3)
    DM_STAMP_FLUSH
}

int winob_launch(const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    p.stamps = nullptr;
    const ConvGeom& g = p.geo;
    DM_REQUIRE(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && !p.up && !p.fold, "winograd bf16x6: 3x3 s1 p1 only");
    DM_REQUIRE(!p.in_nchw && !p.out_nchw, "winograd bf16x6: NHWC only");
    DM_REQUIRE(p.C0 % BCK == 0 && p.C1 % BCK == 0 && p.Cout % 64 == 0, "winograd bf16x6: channel counts");
    DM_REQUIRE(p.Hin == p.Ho && p.Win == p.Wo && p.Ho % 2 == 0 && p.Wo % 2 == 0 && p.Ho >= 16 && p.Wo >= 16,
               "winograd bf16x6: even images of at least 16 x 16");
    DM_REQUIRE((g.WM == 1 || g.WM == 2) && g.TW == 8 && g.TH == 4 * g.WM && g.NB == 1 && g.splits == 1 && g.groups == p.B,
               "winograd bf16x6: plan");
    const bool one = g.WM == 1;
    DM_REQUIRE((size_t)p.B * p.Ho * p.Wo < (1u << 24) && (size_t)p.B * p.Ho * p.Wo * std::max(p.C0, p.C1) < (1ull << 30),
               "winograd bf16x6: tensor too large for 24-bit pixel indices");
    DM_REQUIRE(!(p.epi & EPI_NORM) || g.n_tiles_n == 1, "winograd bf16x6: fused RMSNorm needs one N tile");
    DM_REQUIRE(g.lds_bytes <= 160 * 1024, "winograd bf16x6: tile does not fit LDS");
    DM_REQUIRE(p.chunks0 == p.C0 / BCK && p.n_chunks == (p.C0 + p.C1) / BCK, "winograd bf16x6: chunk counts");
    const int blocks = g.n_tiles_n * g.tiles_x * g.tiles_y * g.groups;
    static const bool xcd_order = env_int("DM_NO_XCD_ORDER", 0) == 0;
    p.geo.xcd_groups = (xcd_order && blocks % 8 == 0 && 8 % g.n_tiles_n == 0) ? 8 / g.n_tiles_n : 0;
    static LdsOptIn lds_flag, lds_flag1;
    if (one ? lds_opt_in(lds_flag1, reinterpret_cast<const void*>(winob1_mfma_kernel), 1)
            : lds_opt_in(lds_flag, reinterpret_cast<const void*>(winob_mfma_kernel), 1))
        return 1;
    const bool timed = prof::enabled();
    if (timed) {
        const double pix = (double)p.B * p.Ho * p.Wo;
        const double cin = p.C0 + p.C1;
        const double flops = 2.0 * 9.0 * cin * p.Cout * pix;
        const double res_rows = (!p.partial && (p.epi & EPI_RESIDUAL)) ? 1.0 : 0.0;
        const double bytes = 4.0 * (cin * pix + (1.0 + res_rows) * p.Cout * pix + 9.0 * cin * p.Cout);
        char name[64];
        if (prof::detail())
            snprintf(name, sizeof(name), "winob<%d> 3x3 s1 %d+%d->%d @%dx%d e%d", g.WM, p.C0, p.C1, p.Cout, p.Ho, p.Wo, p.epi);
        else
            snprintf(name, sizeof(name), "winob_mfma_kernel");
        if (prof::begin(name, flops, bytes, s)) return 1;
    }
#ifdef DM_STAMPS
    {  // diagnostic build: run the launch synchronously with a stamp buffer and print the phase averages
        const size_t nblk = (size_t)blocks;
        unsigned long long* dbuf = nullptr;
        DM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), nblk * 8 * sizeof(unsigned long long)));
        DM_CHECK_HIP(hipMemsetAsync(dbuf, 0, nblk * 8 * sizeof(unsigned long long), s));
        ConvParams ps = p;
        ps.stamps = dbuf;
        if (one)
            hipLaunchKernelGGL(winob1_mfma_kernel, dim3(blocks, 1, 1), dim3(256), p.geo.lds_bytes, s, ps);
        else
            hipLaunchKernelGGL(winob_mfma_kernel, dim3(blocks, 1, 1), dim3(256), p.geo.lds_bytes, s, ps);
        DM_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nblk * 8);
        DM_CHECK_HIP(hipMemcpy(h.data(), dbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(dbuf);
        double avg[8] = {0};
        for (size_t b = 0; b < nblk; ++b)
            for (int k = 0; k < 8; ++k) avg[k] += (double)h[b * 8 + k] / nblk;
        fprintf(stderr, "STAMPS winob<%d> %d+%d->%d @%dx%d e%d chunks %d: wgs=%zu | setup %.0f load+store %.0f transform %.0f "
                        "loop %.0f (%.0f/chunk) reduce %.0f epilogue %.0f\n",
                g.WM, p.C0, p.C1, p.Cout, p.Ho, p.Wo, p.epi, p.n_chunks, nblk, avg[4], avg[5], avg[0], avg[1], avg[1] / p.n_chunks,
                avg[2], avg[3]);
        if (timed && prof::end(s)) return 1;
        return 0;
    }
#endif
    if (one)
        hipLaunchKernelGGL(winob1_mfma_kernel, dim3(blocks, 1, 1), dim3(256), p.geo.lds_bytes, s, p);
    else
        hipLaunchKernelGGL(winob_mfma_kernel, dim3(blocks, 1, 1), dim3(256), p.geo.lds_bytes, s, p);
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    return 0;
}

}  // namespace dm

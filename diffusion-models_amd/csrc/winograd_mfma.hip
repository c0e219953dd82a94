// Winograd F(2x2, 3x3) convolution for gfx950 on the f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// Runs the 3x3 / stride-1 / pad-1 convolutions of Block (DD/denoising_diffusion.py:108, inside ResnetBlock
// :136-148) and the plain 3x3 convs of the last Downsample / Upsample stages (:291, :303) with 16 instead of
// 36 multiplies per 2x2 output pixels and (cin, cout) pair:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d = 4x4 input patch, g = 3x3 filter, Y = 2x2 outputs
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// For each of the 16 positions xi = (i, j) of the transformed patch the reduction over cin is an independent GEMM
//     M[xi][tile][cout] = sum_c V[xi][tile][c] * U[xi][cout][c],   V = B^T d B,   U = G g G^T (host, once)
//
// Mapping (one workgroup = 4 waves = 64 Winograd tiles (128..256 output pixels) x 64 couts; one workgroup per CU):
//   wave i owns row i of the transformed patch: xi = (i, 0..3).  Its accumulators are 4 xi x (64 tiles x 64
//   couts = 2x2 MFMA tiles) = 256 registers.  Nothing but the raw input window is shared between waves:
//   * B operand: U[xi][cout][c] is packed on the host as [chunk of 8 cin][xi][cout][8], so lane (cout = lane&31,
//     half = lane>>5) loads its 4 channels of a chunk as ONE 16-byte global load straight into the MFMA operand
//     registers (a wave reads 1 KiB contiguous).  Weights never touch LDS.
//   * A operand: the raw input window of the tile (NB images x (2TH+2) x (2TW+2) pixels x 8 channels) is staged in
//     LDS (double buffered, one barrier per chunk).  Lane (tile = lane&31 [+32], half) reads the two patch rows
//     that row i of B^T combines (8 ds_read_b128 per tile), forms T = d[ra] +- d[rb] and V[i][0..3] from it -- and
//     those 4 values ARE its MFMA A operands for the chunk.  The transformed input never touches LDS either.
//   * All of that (global loads of the next weights / next-but-one window, LDS reads and the ~64 VALU ops of the
//     next chunk's transform) is issued from hooks between the 64 MFMAs of the current chunk.
//   Epilogue: each wave reduces its 4 columns with A (R_i[b], 2 values), the four R_i go through LDS in the
//   row layout of conv_mfma.hip, Y[a][b] = sum_i A^T[a][i] R_i[b], then the shared Block epilogue
//   (bias / RMSNorm / scale-shift / SiLU / residual, or raw K-split partial sums).
//
// The summation order differs from the direct kernel's (and from PyTorch's), like any fp32 convolution
// algorithm; tests/test_hip_ops.py holds the tolerance.
#include "conv_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace dm {

static constexpr int WCK = 8;     // input channels per K chunk
static constexpr int WTS = 68;    // padded row stride of the transposed epilogue tiles (floats)


static inline int w_pow2ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
static inline int w_ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
bool wino_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up) {
    static const bool off = std::getenv("DM_NO_WINOGRAD") != nullptr;
    return !off && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !up && C0 > 0 && C0 % WCK == 0 &&
           C1 % WCK == 0 && Cout % 64 == 0;
}

size_t wino_packed_floats(int Cout, int C0, int C1) { return (size_t)(C0 + C1) * 16 * Cout; }

void wino_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int Cin = C0 + C1;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float* gk = oihw + ((size_t)co * Cin + ci) * 9;
            double Gg[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) Gg[i][b] = G[i][0] * gk[b] + G[i][1] * gk[3 + b] + G[i][2] * gk[6 + b];
            const int chunk = ci / WCK, cc = ci % WCK;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    double u = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
                    packed[(((size_t)chunk * 16 + i * 4 + j) * Cout + co) * WCK + cc] = (float)u;
                }
        }
}

ConvGeom wino_plan(int B, int Ho, int Wo, int Cout, int C0, int C1, bool allow_split, bool want_norm) {
    // R = Winograd tiles per lane: 2 -> 64 tiles per workgroup, 256 accumulator registers, one workgroup per CU;
    //                              1 -> 32 tiles per workgroup, 128 accumulator registers, two workgroups per CU
    static const int R = env_int("DM_WINO_R", 1) == 2 ? 2 : 1;
    const int tiles = 32 * R;
    ConvGeom g{};
    g.WM = R;
    g.WN = 1;
    g.CK = WCK;
    const int twi = (Wo + 1) / 2, thi = (Ho + 1) / 2;  // tiles per image
    g.TW = std::min(w_pow2ceil(twi), 8);
    g.TH = std::min(w_pow2ceil(thi), tiles / g.TW);
    g.NB = tiles / (g.TW * g.TH);
    g.lTW = w_ilog2(g.TW);
    g.lTH = w_ilog2(g.TH);
    g.tiles_x = (twi + g.TW - 1) / g.TW;
    g.tiles_y = (thi + g.TH - 1) / g.TH;
    g.groups = (B + g.NB - 1) / g.NB;
    g.n_tiles_n = Cout / 64;
    g.IH = 2 * g.TH + 2;
    g.IW = 2 * g.TW + 2;
    g.row_stride = g.IW * WCK + 4;  // +4: rows of consecutive tile rows start 8 banks apart mod 16
    g.halo_floats = g.NB * g.IH * g.row_stride;
    g.magic_win = ((1 << 24) + g.IH * g.IW - 1) / (g.IH * g.IW);  // exact for the < 2^10 window pixels
    g.magic_row = ((1 << 24) + g.IW - 1) / g.IW;
    g.TPS = 3;
    const int n_chunks = (C0 + C1) / WCK;
    const int wgs = g.tiles_x * g.tiles_y * g.groups * g.n_tiles_n;
    int splits = 1;
    if (allow_split) {
        static const int target = env_int("DM_WINO_TARGET_WGS", 256);
        static const int min_chunks = env_int("DM_WINO_MIN_CHUNKS", 8);
        while (wgs * splits < target * (3 - R) && splits < 8 && n_chunks / (splits * 2) >= min_chunks) splits *= 2;
    }
    g.chunks_per_split = (n_chunks + splits - 1) / splits;
    g.splits = (n_chunks + g.chunks_per_split - 1) / g.chunks_per_split;
    g.fused_norm = g.n_tiles_n == 1 && g.splits == 1;
    // A grid that still leaves workgroup slots empty is bound by the MFMA chain of ONE wave (chunks x 32 MFMAs x 64 cycles):
    // with 32 instead of 64 couts per workgroup the chain halves and the workgroups double.  Not where the 64-cout tile
    // holds a pixel's whole row (the RMSNorm then runs in the kernel's epilogue instead of a landing pass).
    g.NQ = 2;
    static const int q_target = env_int("DM_WINO_Q_TARGET_WGS", 512);
    // (want_norm = false: the caller has no norm to fuse -- the training step's convolutions, whose norms are separate passes
    // over the tape -- so a single 64-cout tile is no reason to keep it)
    if (R == 1 && !(g.fused_norm && want_norm) && wgs * g.splits < q_target) {
        g.NQ = 1;
        g.n_tiles_n = Cout / 32;
        g.fused_norm = 0;
    }
    g.w_floats = 0;
    // two windows + a scratch slot reachable from both; the epilogue reuses the space for 4 x 2 transposed tiles;
    // behind both: the output-pixel table [wave][2a+b][tile of the wave] (built once, read by the epilogue)
    g.ptab_off = std::max(3 * g.halo_floats + 4, 4 * 2 * tiles * WTS);
    g.lds_bytes = (g.ptab_off + 4 * tiles) * 4;
    return g;
}

bool wino_shape_ok(int B, int Ho, int Wo, int Cout, int C0, int C1) {
    if ((Ho | Wo) & 1) return false;
    const ConvGeom g = wino_plan(B, Ho, Wo, Cout, C0, C1, true);
    // the window of one tile block must fit the staging registers (very small images pack too many per block)
    return g.NB * g.IH * g.IW * 2 <= 256 * (g.WM == 2 ? 5 : 3) && g.lds_bytes <= 160 * 1024 &&
           (size_t)B * Ho * Wo < (1u << 24) && (size_t)B * Ho * Wo * std::max(C0, C1) < (1ull << 30);
}

template <int R, int Q>
__global__ __launch_bounds__(256, 3 - R) void wino_mfma_kernel(const ConvParams p) {
    constexpr int TILES = 32 * R;         // Winograd tiles per workgroup
    constexpr int NC = 32 * Q;            // couts per workgroup
    constexpr int HR = R == 2 ? 5 : 3;    // window staging registers (16 B each) per thread
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
    const int group = bid / g.tiles_y;
    const int tx0 = tile_x * g.TW, ty0 = tile_y * g.TH, b0 = group * g.NB;  // in Winograd tiles
    const int ix0 = 2 * tx0 - 1, iy0 = 2 * ty0 - 1;                         // window origin in pixels
    const int split = blockIdx.y;
    const int cb = split * g.chunks_per_split;
    const int ce = min(cb + g.chunks_per_split, p.n_chunks);
    DM_STAMP_DECL
    DM_STAMP(0);
    const int RS = g.row_stride;
    float* raw0 = smem;
    float* raw1 = smem + g.halo_floats;

    // ---- window staging: item = (window pixel, channel quad); the quad is tid & 1 for every item of a thread
    const int win_items = g.NB * g.IH * g.IW * 2;
    int hpix[HR], hoff[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int it = tid + 256 * i;
        hpix[i] = -1;
        hoff[i] = 2 * g.halo_floats;  // items past the window go to a scratch slot: the main loop has no branches
        if (it < win_items) {
            const int hp = it >> 1;
            const int nb = (int)(((unsigned)hp * (unsigned)g.magic_win) >> 24);  // hp / (IH * IW)
            const int rem = hp - nb * (g.IH * g.IW);
            const int hy = (int)(((unsigned)rem * (unsigned)g.magic_row) >> 24);  // rem / IW
            const int hx = rem - hy * g.IW;
            const int b = b0 + nb, iy = iy0 + hy, ix = ix0 + hx;
            // the two channel quads of a pixel swap places in every other group of 8 columns: the 16 lanes of a
            // ds_read_b128 pass (8 tiles = every second pixel, x 2 tile rows) then hit 16 different 4-bank groups
            const int off = (nb * g.IH + hy) * RS + hx * WCK + 4 * ((tid & 1) ^ ((hx >> 3) & 1));
            if (b < p.B && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
                hpix[i] = (b * p.Hin + iy) * p.Win + ix;
                hoff[i] = off;
            } else {
                // padding / out-of-batch pixel: zero in both buffers, once; its loads (pixel 0) land in the scratch slot
                *reinterpret_cast<f32x4*>(raw0 + off) = make_f32x4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<f32x4*>(raw1 + off) = make_f32x4(0.f, 0.f, 0.f, 0.f);
            }
        }
        hpix[i] = max(hpix[i], 0);
    }
    {
        // output pixel of (tile, a, b), or -1: thread = tile * 4 + (2a + b)
        constexpr int NRT = 8 * R;
        int* ptab = reinterpret_cast<int*>(smem + g.ptab_off);
        if (tid < 4 * TILES) {
            const int t = tid >> 2, ab = tid & 3;
            const int tx = t & (g.TW - 1);
            const int ty = (t >> g.lTW) & (g.TH - 1);
            const int nb = t >> (g.lTW + g.lTH);
            const int b = b0 + nb, y = 2 * (ty0 + ty) + (ab >> 1), x = 2 * (tx0 + tx) + (ab & 1);
            ptab[(t / NRT) * (4 * NRT) + ab * NRT + (t % NRT)] =
                (b < p.B && y < p.Ho && x < p.Wo) ? (b * p.Ho + y) * p.Wo + x : -1;
        }
    }
    f32x4 hreg[HR];
    // Window loads: per-lane byte offset hvo[i] = (pixel * C_source + quad) * 4 (recomputed only when the source
    // changes), per-chunk scalar offset = first channel of the chunk.  Pixel index and channel count are < 2^24
    // and the tensors < 2^30 elements (wino_launch checks).
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
        return bufload4(s1 ? rs_in1 : rs_in0, hvo[i], (unsigned)(s1 ? chunk - p.chunks0 : chunk) * (WCK * 4));
    };
    auto store_window = [&](float* raw, int i) { *reinterpret_cast<f32x4*>(raw + hoff[i]) = hreg[i]; };

    // ---- input transform of this lane: tiles l31 (and 32 + l31), channel quad lh, row `wave` of B^T d
    int rbase[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = 32 * r + l31;
        const int ty = (t >> g.lTW) & (g.TH - 1);
        const int nb = t >> (g.lTW + g.lTH);
        rbase[r] = (nb * g.IH + 2 * ty) * RS;
    }
    // B^T row i = d[ra] + sgn * d[rb]:  i=0: d0 - d2;  i=1: d1 + d2;  i=2: d2 - d1;  i=3: d1 - d3
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const int offa = ra * RS, offb = rb * RS;
    int colq[4];  // window column 2*tx + b of this lane's tiles: pixel slot + (swizzled) quad of this lane half
    {
        const int tx = l31 & (g.TW - 1);
#pragma unroll
        for (int b = 0; b < 4; ++b) colq[b] = (2 * tx + b) * WCK + 4 * (lh ^ (((2 * tx + b) >> 3) & 1));
    }
    auto rd = [&](const float* raw, int r, int ab, int b) {  // patch row ra (ab = 0) or rb (1), column b, of tile r
        return *reinterpret_cast<const f32x4*>(raw + rbase[r] + (ab ? offb : offa) + colq[b]);
    };
    // the same addresses precomputed per LDS buffer: in the main loop (which is unrolled over the two buffers) every
    // LDS access is then a register base + an immediate offset
    const float* rdp[2][R][2][4];  // [buffer][tile r][row a / row b][column]
    float* stp[2][HR];             // [buffer][window item]
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        float* base = bf ? raw1 : raw0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                rdp[bf][r][0][b] = base + rbase[r] + offa + colq[b];
                rdp[bf][r][1][b] = base + rbase[r] + offb + colq[b];
            }
#pragma unroll
        for (int i = 0; i < HR; ++i) stp[bf][i] = base + hoff[i];
    }

    f32x4 A[4][R];  // V[wave][j] of tile r: channels 4*lh .. 4*lh+3 of the chunk
    f32x4 U[4][Q];  // U[wave*4 + j][cout 32*q + l31][channels 4*lh ..]
    const size_t u_chunk = (size_t)16 * p.Cout * WCK;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (size_t)p.n_chunks * u_chunk * 4);
    const unsigned uvo = (l31 * WCK + 4 * lh) * 4;  // bytes
    const unsigned u_row = (unsigned)p.Cout * WCK * 4;  // bytes between consecutive xi
    const unsigned u_wave = ((unsigned)(4 * wave) * p.Cout + n_tile * NC) * WCK * 4;
    auto load_u = [&](int chunk, int j) {  // U of (chunk, xi = 4*wave + j): couts l31 and (Q == 2) 32 + l31
        const unsigned so = (unsigned)chunk * (unsigned)(u_chunk * 4) + u_wave + j * u_row;
        U[j][0] = bufload4(rs_w, uvo, so);
        if constexpr (Q == 2) U[j][1] = bufload4(rs_w, uvo, so + 32 * WCK * 4);
    };

    f32x16 acc[4][R][Q];  // first written by the first chunk's MFMAs (C = 0)

    DM_STAMP_ADD(4)
    // ---- prologue: chunks cb and cb + 1 -> LDS (both loads in flight together), operands of chunk cb -> registers
    {
        const bool two = cb + 1 < ce;
        f32x4 h2[HR];
        window_offsets(cb >= p.chunks0 ? p.C1 : p.C0);
#pragma unroll
        for (int i = 0; i < HR; ++i) hreg[i] = window_value(cb, i);
        const int c1 = two ? cb + 1 : cb;
        if (c1 == p.chunks0) window_offsets(p.C1);
#pragma unroll
        for (int i = 0; i < HR; ++i) h2[i] = window_value(c1, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) load_u(cb, j);
#pragma unroll
        for (int i = 0; i < HR; ++i) store_window(raw0, i);
#pragma unroll
        for (int i = 0; i < HR; ++i) hreg[i] = h2[i];
#pragma unroll
        for (int i = 0; i < HR; ++i) store_window(raw1, i);
    }
    DM_STAMP_ADD(5)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        f32x4 T[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) T[b] = rd(raw0, r, 0, b) + sgn * rd(raw0, r, 1, b);
        A[0][r] = sub4(T[0], T[2]);
        A[1][r] = add4(T[1], T[2]);
        A[2][r] = sub4(T[2], T[1]);
        A[3][r] = sub4(T[1], T[3]);
    }
    __syncthreads();  // raw0 is overwritten with chunk cb + 2 by the first iteration
    DM_STAMP_ADD(0)

    // ---- main loop: NM MFMAs per chunk and wave; everything else is issued from the hooks between them.
    // The body is ONE basic block (no branch: loads past the end re-read valid memory and are never used), so
    // sched_barrier pins every hook to its slot.
    // The first chunk runs as its own copy of the body whose first MFMA per accumulator takes C = 0 (an inline
    // constant): no accumulator initialisation.
    const f32x2 sgn2 = {sgn, sgn};
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
    auto chunk_body = [&](int c, auto first_tag, auto par_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;  // (c - cb) & 1
        const bool has1 = c + 1 < ce, has2 = c + 2 < ce;
        // chunk c + 1 sits in buffer PAR ^ 1 (stored during iteration c - 1 / the prologue); chunk c was read from
        // buffer PAR during iteration c - 1, so it is free for chunk c + 2
        constexpr int BN = PAR ^ 1, BS = PAR;
        const int cw = has2 ? c + 2 : c;   // chunk whose window is fetched now (c again at the end: never read)
        const int unext = has1 ? c + 1 : c;  // chunk whose weights are fetched now (never past the packed weights)
        if (cw == p.chunks0 && p.C1 != p.C0) {  // uniform, once per kernel, outside the MFMA stream
            window_offsets(p.C1);
            asm volatile("" ::: "memory");  // keep this a branch: if-converted it costs a select per offset and chunk
        }
        auto rdn = [&](int r, int ab, int b) { return *reinterpret_cast<const f32x4*>(rdp[BN][r][ab][b]); };
        f32x4 T[R][4];
        if constexpr (R == 2) {
            static_assert(R == 1 || Q == 2, "the 64-tile form keeps 64 couts");
            f32x4 d[8];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int m = ((j * 4 + s) * 2 + r) * 2 + q;  // 0..63
                            acc[j][r][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                A[j][r][s], U[j][q][s], (FIRST && s == 0) ? zero16 : acc[j][r][q], 0, 0, 0);
                            if (m < HR) hreg[m] = window_value(cw, m);
                            if (m >= 6 && m < 14) d[m - 6] = rdn(0, (m - 6) >> 2, (m - 6) & 3);
                            if (m >= 14 && m < 18) T[0][m - 14] = fma4(d[4 + m - 14], sgn2, d[m - 14]);
                            if (m == 16) load_u(unext, 0);  // the MFMAs of j = 0 were issued by m = 15
                            if (m >= 18 && m < 26) d[m - 18] = rdn(1, (m - 18) >> 2, (m - 18) & 3);
                            if (m >= 26 && m < 30) T[1][m - 26] = fma4(d[4 + m - 26], sgn2, d[m - 26]);
                            if (m == 30) {
                                A[0][0] = sub4(T[0][0], T[0][2]);
                                A[0][1] = sub4(T[1][0], T[1][2]);
                            }
                            if (m == 32) {  // j = 1 done at m = 31
                                A[1][0] = add4(T[0][1], T[0][2]);
                                A[1][1] = add4(T[1][1], T[1][2]);
                            }
                            if (m == 33) load_u(unext, 1);
                            if (m == 48) {  // j = 2 done at m = 47
                                A[2][0] = sub4(T[0][2], T[0][1]);
                                A[2][1] = sub4(T[1][2], T[1][1]);
                            }
                            if (m == 49) load_u(unext, 2);
                            if (m >= 56 && m < 56 + HR) *reinterpret_cast<f32x4*>(stp[BS][m - 56]) = hreg[m - 56];
                            __builtin_amdgcn_sched_barrier(0);
                        }
            A[3][0] = sub4(T[0][1], T[0][3]);
            A[3][1] = sub4(T[1][1], T[1][3]);
        } else {
            f32x4 d[4];
            // the hooks of the 32-MFMA stream (Q == 2: one per MFMA slot; Q == 1: two per slot of its 16 MFMAs -- "the MFMAs
            // of j are issued" holds at the same hook index either way, MFMA (j, s, q) sitting at index ((4 j + s) 2 + q 2 / Q))
            auto hook = [&](int m) {
                if (m < HR) hreg[m] = window_value(cw, m);
                if (m >= 3 && m < 7) d[m - 3] = rdn(0, (m - 3) & 1, (m - 3) >> 1);  // columns 0, 1
                if (m == 8) load_u(unext, 0);  // j = 0 done at m = 7
                if (m == 11) T[0][0] = fma4(d[1], sgn2, d[0]);
                if (m == 12) T[0][1] = fma4(d[3], sgn2, d[2]);
                if (m >= 13 && m < 17) d[m - 13] = rdn(0, (m - 13) & 1, 2 + ((m - 13) >> 1));  // columns 2, 3
                if (m == 17) load_u(unext, 1);  // j = 1 done at m = 15
                if (m == 21) T[0][2] = fma4(d[1], sgn2, d[0]);
                if (m == 22) {
                    T[0][3] = fma4(d[3], sgn2, d[2]);
                    A[0][0] = sub4(T[0][0], T[0][2]);
                }
                if (m == 23) A[1][0] = add4(T[0][1], T[0][2]);
                if (m == 24) load_u(unext, 2);  // j = 2 done at m = 23
                if (m == 25) A[2][0] = sub4(T[0][2], T[0][1]);
                if (m >= 27 && m < 27 + HR) *reinterpret_cast<f32x4*>(stp[BS][m - 27]) = hreg[m - 27];
            };
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const int m = (j * 4 + s) * Q + q;  // 0 .. 16 Q - 1
                        acc[j][0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            A[j][0][s], U[j][q][s], (FIRST && s == 0) ? zero16 : acc[j][0][q], 0, 0, 0);
                        if constexpr (Q == 2) {
                            hook(m);
                        } else {
                            hook(2 * m);
                            hook(2 * m + 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            A[3][0] = sub4(T[0][1], T[0][3]);
        }
        load_u(unext, 3);
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

    DM_STAMP_ADD(1)
    // ---- epilogue: R_i[b] = sum_j M[i][j] A[j][b] per wave, then Y[a][b] = sum_i A^T[a][i] R_i[b] through LDS.
    // Wave w finishes tiles [NR*w, NR*w + NR): lane group rsub = 2a + b owns pixel (a, b) of each tile.  What
    // the epilogue needs from global memory is requested first.
    constexpr int NR = 8 * R;
    const int rsub = lane >> 4;
    const int oa = rsub >> 1, ob = rsub & 1;
    const int c4 = (lane & 15) * 4;
    const int cg = n_tile * NC + c4;
    const bool cvalid = c4 < NC && cg < p.Cout;  // Q == 1: the upper half of a 16-lane row has no couts
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
    re.split = split;
    re.M = (size_t)p.B * p.Ho * p.Wo;
    re.b0 = b0;
    re.uni = g.NB == 1 || p.ss_stride == 0;
    re.HoWo = p.Ho * p.Wo;
    re.red = nullptr;
    re.rows_per_wg = 4 * TILES;
    re.row_in_wg0 = wave * 4 * NR;
    re.wn = 0;
    re.all_valid = Q == 2;
    RowsPrefetch<NR, true> pf;
    rows_prefetch<NR, true>(p, re, pixv, cg, cvalid, pf);

    float* Tb = smem + wave * (2 * TILES * WTS);  // [b][tile][WTS]
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {  // two accumulator registers per packed instruction
                const f32x2 a0 = {acc[0][r][q][e], acc[0][r][q][e + 1]}, a1 = {acc[1][r][q][e], acc[1][r][q][e + 1]};
                const f32x2 a2 = {acc[2][r][q][e], acc[2][r][q][e + 1]}, a3 = {acc[3][r][q][e], acc[3][r][q][e + 1]};
                const f32x2 r0 = pk_add(pk_add(a0, a1), a2);
                const f32x2 r1 = pk_sub(pk_sub(a1, a2), a3);
                const int row = r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                Tb[row * WTS + q * 32 + l31] = r0.x;
                Tb[(row + 1) * WTS + q * 32 + l31] = r0.y;
                Tb[(TILES + row) * WTS + q * 32 + l31] = r1.x;
                Tb[(TILES + row + 1) * WTS + q * 32 + l31] = r1.y;
            }
        }
    __syncthreads();
    DM_STAMP_ADD(2)
    const float ysgn = oa ? -1.0f : 1.0f;  // Y[0] = R0 + R1 + R2,  Y[1] = R1 - R2 - R3
    const float* Y0 = smem + (oa * 2 + ob) * (TILES * WTS) + c4;
    f32x4 v[NR];
#pragma unroll
    for (int jj = 0; jj < NR; ++jj) {
        const float* yp = Y0 + (NR * wave + jj) * WTS;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(yp);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(yp + 2 * TILES * WTS);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(yp + 4 * TILES * WTS);
        v[jj] = a0 + ysgn * a1 + ysgn * a2;
    }
    rows_epilogue<1, NR, true>(p, re, v, pixv, cg, cvalid, pf);
    DM_STAMP_ADD(3)
    DM_STAMP_FLUSH
}

template <int R, int Q>
static int wino_launch_r(const ConvParams& p, int blocks, hipStream_t s) {
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(wino_mfma_kernel<R, Q>), 1)) return 1;
#ifdef DM_STAMPS
    // diagnostic build: run the launch synchronously with a stamp buffer and print the phase averages
    {
        const size_t nblk = (size_t)blocks * p.geo.splits;
        unsigned long long* dbuf = nullptr;
        DM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), nblk * 8 * sizeof(unsigned long long)));
        DM_CHECK_HIP(hipMemsetAsync(dbuf, 0, nblk * 8 * sizeof(unsigned long long), s));
        ConvParams ps = p;
        ps.stamps = dbuf;
        hipLaunchKernelGGL((wino_mfma_kernel<R, Q>), dim3(blocks, p.geo.splits, 1), dim3(256), p.geo.lds_bytes, s, ps);
        DM_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nblk * 8);
        DM_CHECK_HIP(hipMemcpy(h.data(), dbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(dbuf);
        double avg[8] = {0};
        for (size_t b = 0; b < nblk; ++b)
            for (int k = 0; k < 8; ++k) avg[k] += (double)h[b * 8 + k] / nblk;
        fprintf(stderr, "STAMPS wino<%d> %d+%d->%d @%dx%d e%d k%d chunks %d: wgs=%zu | setup %.0f load+store %.0f "
                        "transform %.0f loop %.0f (%.0f/chunk) reduce %.0f epilogue %.0f\n",
                R, p.C0, p.C1, p.Cout, p.Ho, p.Wo, p.epi, p.geo.splits, p.geo.chunks_per_split, nblk, avg[4], avg[5],
                avg[0], avg[1], avg[1] / p.geo.chunks_per_split, avg[2], avg[3]);
        return 0;
    }
#endif
    hipLaunchKernelGGL((wino_mfma_kernel<R, Q>), dim3(blocks, p.geo.splits, 1), dim3(256), p.geo.lds_bytes, s, p);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

int wino_launch(const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    p.stamps = nullptr;
    const ConvGeom& g = p.geo;
    DM_REQUIRE(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && !p.up && !p.fold, "winograd: 3x3 s1 p1 only");
    DM_REQUIRE(!p.in_nchw && !p.out_nchw, "winograd: NHWC only");
    DM_REQUIRE(p.C0 % WCK == 0 && p.C1 % WCK == 0 && p.Cout % 64 == 0, "winograd: channel counts");
    DM_REQUIRE(p.Hin == p.Ho && p.Win == p.Wo, "winograd: same-size convolution");
    const int R = g.WM;
    DM_REQUIRE(R == 1 || R == 2, "winograd: tiles per lane");
    DM_REQUIRE((g.NQ == 2 || (g.NQ == 1 && R == 1)) && g.n_tiles_n == p.Cout / (32 * g.NQ), "winograd: cout tile");
    DM_REQUIRE(g.TW * g.TH * g.NB == 32 * R, "winograd: 32 R tiles per workgroup");
    DM_REQUIRE(g.NB * g.IH * g.IW * 2 <= 256 * (R == 2 ? 5 : 3), "winograd: window exceeds staging registers");
    DM_REQUIRE((size_t)p.B * p.Ho * p.Wo < (1u << 24) && p.C0 < (1 << 24) && p.C1 < (1 << 24) &&
                   (size_t)p.B * p.Ho * p.Wo * std::max(p.C0, p.C1) < (1ull << 30),
               "winograd: tensor too large for 24-bit pixel indices");
    DM_REQUIRE(!(p.epi & EPI_NORM) || (g.n_tiles_n == 1 && g.splits == 1), "winograd: fused RMSNorm needs one N tile");
    DM_REQUIRE(g.splits == 1 || p.partial, "winograd: split-K writes partial sums");
    DM_REQUIRE(g.lds_bytes <= 160 * 1024, "winograd: tile does not fit LDS");
    DM_REQUIRE(p.chunks0 == p.C0 / WCK && p.n_chunks == (p.C0 + p.C1) / WCK, "winograd: chunk counts");
    const int blocks = g.n_tiles_n * g.tiles_x * g.tiles_y * g.groups;
    // XCD-aware block order (conv_device.h: block_to_tile); DM_NO_XCD_ORDER=1 keeps the raw order for A/B runs
    static const bool xcd_order = env_int("DM_NO_XCD_ORDER", 0) == 0;
    p.geo.xcd_groups = (xcd_order && blocks % 8 == 0 && 8 % g.n_tiles_n == 0) ? 8 / g.n_tiles_n : 0;
    const bool timed = prof::enabled();
    if (timed) {
        // priced as the reference's op: 2*9*Cin*Cout*pixels FLOP; the kernel executes 16/36 of
        // the multiply-adds of that count
        const double pix = (double)p.B * p.Ho * p.Wo;
        const double cin = p.C0 + p.C1;
        const double flops = 2.0 * 9.0 * cin * p.Cout * pix;
        const double res_rows = (!p.partial && (p.epi & EPI_RESIDUAL)) ? 1.0 : 0.0;  // the fused residual add reads one more tensor
        const double bytes = 4.0 * (cin * pix + (1.0 + res_rows) * p.Cout * pix + 9.0 * cin * p.Cout);
        char name[64];
        if (prof::detail())
            snprintf(name, sizeof(name), "wino<%d> 3x3 s1 %d+%d->%d @%dx%d e%d k%d g%d q%d", R, p.C0, p.C1, p.Cout, p.Ho,
                     p.Wo, p.epi, g.splits, blocks * g.splits, g.NQ);
        else
            snprintf(name, sizeof(name), "wino_mfma_kernel<%d, %d>", R, g.NQ);  // the symbol rocprofv3 reports
        if (prof::begin(name, flops, bytes, s)) return 1;
    }
    if (R == 2 ? wino_launch_r<2, 2>(p, blocks, s) : (g.NQ == 1 ? wino_launch_r<1, 1>(p, blocks, s) : wino_launch_r<1, 2>(p, blocks, s)))
        return 1;
    if (timed && prof::end(s)) return 1;
    return 0;
}

}  // namespace dm

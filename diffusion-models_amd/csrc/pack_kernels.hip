// Device-side weight packing for the training loop: after every optimiser step the master parameters (reference
// layouts, one flat device buffer) are re-packed into the layouts the convolution kernels read -- the same layouts the
// host packers produce at dm_unet_finalize (conv_pack_weights, wino_pack_weights, wino4_pack_weights,
// upwino_pack_weights, pw_pack_weights[_s2d], init7_pack_weights, the four parity convolutions of a folded upsample conv)
// -- without a device -> host -> device round trip of 143 MB of weights.  Each kernel mirrors its host packer line by line
// (same double-precision Winograd transforms, same index arithmetic); dm_unet_check_device_pack compares every buffer
// against the host packer bit for bit.  The input-gradient convolutions read rotated / transposed weights: those are
// materialised first (rot_transpose_kernel, s2d_transpose_kernel) and go through the same packers.
#include "dm_common.h"

#include <algorithm>

namespace dm {

static __device__ __forceinline__ int d_pad_to(int v, int m) { return (v + m - 1) / m * m; }

// conv_pack_weights: OIHW (Cout, C0 + C1, KH, KW) -> [chunk][ky][kx][CoutP][CK], zero filled; one thread per packed float
__global__ void pack_direct_kernel(const float* __restrict__ oihw, float* __restrict__ packed, int Cout, int C0, int C1, int KH,
                                   int KW, int CK, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int CoutP = d_pad_to(Cout, 256);
    const int chunks0 = d_pad_to(C0, CK) / CK;
    const int Cin = C0 + C1;
    const int kk = (int)(i % CK);
    int64_t r = i / CK;
    const int co = (int)(r % CoutP);
    r /= CoutP;
    const int kx = (int)(r % KW);
    r /= KW;
    const int ky = (int)(r % KH);
    const int ch = (int)(r / KH);
    float v = 0.f;
    if (co < Cout) {
        int cin = -1;
        if (ch < chunks0) {
            const int c = ch * CK + kk;
            if (c < C0) cin = c;
        } else {
            const int c = (ch - chunks0) * CK + kk;
            if (c < C1) cin = C0 + c;
        }
        if (cin >= 0) v = oihw[(((size_t)co * Cin + cin) * KH + ky) * KW + kx];
    }
    packed[i] = v;
}

// the 2x2 parity convolution (py, px) of nearest-x2 + conv3x3: W'[o][c][a][b] = sum of the 3x3 taps that hit source offset (a, b)
__global__ void fold_taps_kernel(const float* __restrict__ oihw, float* __restrict__ w2, int64_t oc, int py, int px) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= oc) return;
    const float* g = oihw + i * 9;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int y0, y1, x0, x1;
            if (py == 0) { y0 = a == 0 ? 0 : 1; y1 = a == 0 ? 0 : 2; } else { y0 = a == 0 ? 0 : 2; y1 = a == 0 ? 1 : 2; }
            if (px == 0) { x0 = b == 0 ? 0 : 1; x1 = b == 0 ? 0 : 2; } else { x0 = b == 0 ? 0 : 2; x1 = b == 0 ? 1 : 2; }
            float s = 0.f;
            for (int dy = y0; dy <= y1; ++dy)
                for (int dx = x0; dx <= x1; ++dx) s += g[dy * 3 + dx];
            w2[i * 4 + a * 2 + b] = s;
        }
}

// the nine taps of pair i = cout * Cin + cin of an OIHW 3x3 weight; rs != 0: the pair of the ROTATED weight an input-gradient
// convolution reads -- w_b[cout][cin][ky][kx] = w[cin][rc + cout][2 - ky][2 - kx] with rs the Cin of w -- read in place (the
// grouped re-pack no longer materialises the rotated tensor; the reads are runs of 9 x (consecutive couts) floats)
__device__ __forceinline__ void load_taps9(const float* __restrict__ w, int Cin, int rs, int rc, int64_t i, float* g) {
    if (rs == 0) {
        const float* p = w + i * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k] = p[k];
    } else {
        const int co = (int)(i / Cin), ci = (int)(i % Cin);
        const float* p = w + ((size_t)ci * rs + rc + co) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k] = p[8 - k];
    }
}

// wino_pack_weights: U = G g G^T (4x4) -> [chunk of 8 cin][xi 16][Cout][8]; one thread per (cout, cin)
__device__ __forceinline__ void pack_wino_elem(const float* __restrict__ oihw, float* __restrict__ packed, int Cout, int Cin,
                                               int64_t i, int rs = 0, int rc = 0) {
#pragma clang fp contract(off)  // the host packer (x86-64 baseline) rounds every product and sum separately
    const int co = (int)(i / Cin), ci = (int)(i % Cin);
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    float gk[9];
    load_taps9(oihw, Cin, rs, rc, i, gk);
    double Gg[4][3];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 3; ++b) Gg[a][b] = G[a][0] * gk[b] + G[a][1] * gk[3 + b] + G[a][2] * gk[6 + b];
    const int chunk = ci / 8, cc = ci % 8;
    for (int a = 0; a < 4; ++a)
        for (int j = 0; j < 4; ++j) {
            const double u = Gg[a][0] * G[j][0] + Gg[a][1] * G[j][1] + Gg[a][2] * G[j][2];
            packed[(((size_t)chunk * 16 + a * 4 + j) * Cout + co) * 8 + cc] = (float)u;
        }
}

__global__ void pack_wino_kernel(const float* __restrict__ oihw, float* __restrict__ packed, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)Cout * Cin) pack_wino_elem(oihw, packed, Cout, Cin, i);
}
// wino4_pack_weights: U = G g G^T (6x6) -> [chunk][wave 4][slot 9][cout tile][kq 4][n 16][gq 4][st 2]
__device__ __forceinline__ void pack_wino4_elem(const float* __restrict__ oihw, float* __restrict__ packed, int Cout, int Cin,
                                                int64_t i, int rs = 0, int rc = 0) {
#pragma clang fp contract(off)  // the host packer (x86-64 baseline) rounds every product and sum separately
    const int co = (int)(i / Cin), ci = (int)(i % Cin);
    const double G[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int lo[3] = {0, 1, 2}, hi[3] = {5, 3, 4};
    float gk[9];
    load_taps9(oihw, Cin, rs, rc, i, gk);
    double Gg[6][3];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b) Gg[a][b] = G[a][0] * gk[b] + G[a][1] * gk[3 + b] + G[a][2] * gk[6 + b];
    double U[6][6];
    for (int a = 0; a < 6; ++a)
        for (int j = 0; j < 6; ++j) U[a][j] = Gg[a][0] * G[j][0] + Gg[a][1] * G[j][1] + Gg[a][2] * G[j][2];
    const int chunk = ci / 8, cc = ci % 8;
    const int ct = co / 64, gq = (co % 64) / 16, n = co % 16, kq = cc / 2, st = cc % 2;
    for (int g = 0; g < 4; ++g)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const int ii = (g >> 1) ? hi[a] : lo[a], jj = (g & 1) ? hi[b] : lo[b];
                packed[(((((size_t)chunk * 4 + g) * 9 + a * 3 + b) * (Cout / 64) + ct) * 4 + kq) * 128 + n * 8 + gq * 2 + st] =
                    (float)U[ii][jj];
            }
}

__global__ void pack_wino4_kernel(const float* __restrict__ oihw, float* __restrict__ packed, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)Cout * Cin) pack_wino4_elem(oihw, packed, Cout, Cin, i);
}
// upwino_pack_weights: G = [1 0 0; 1 1 1; 0 0 1] -> [chunk][xi 9][cout tile][kq 4][n 16][gq 4][st 2]
__device__ __forceinline__ void pack_upwino_elem(const float* __restrict__ oihw, float* __restrict__ packed, int Cout, int Cin,
                                                 int64_t i, int rs = 0, int rc = 0) {
#pragma clang fp contract(off)  // the host packer (x86-64 baseline) rounds every product and sum separately
    const int co = (int)(i / Cin), ci = (int)(i % Cin);
    const double G[3][3] = {{1, 0, 0}, {1, 1, 1}, {0, 0, 1}};
    float gk[9];
    load_taps9(oihw, Cin, rs, rc, i, gk);
    double Gg[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Gg[a][b] = G[a][0] * gk[b] + G[a][1] * gk[3 + b] + G[a][2] * gk[6 + b];
    const int chunk = ci / 8, cc = ci % 8;
    const int ct = co / 64, gq = (co % 64) / 16, n = co % 16, kq = cc / 2, st = cc % 2;
    for (int a = 0; a < 3; ++a)
        for (int j = 0; j < 3; ++j) {
            const double u = Gg[a][0] * G[j][0] + Gg[a][1] * G[j][1] + Gg[a][2] * G[j][2];
            packed[((((size_t)chunk * 9 + 3 * a + j) * (Cout / 64) + ct) * 4 + kq) * 128 + n * 8 + gq * 2 + st] = (float)u;
        }
}

__global__ void pack_upwino_kernel(const float* __restrict__ oihw, float* __restrict__ packed, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)Cout * Cin) pack_upwino_elem(oihw, packed, Cout, Cin, i);
}
// pw_pack_weights: (Cout, Cin) -> [chunk of 16][cout tile of 16][lane = 16 kq + l15][j 4];  s2d: the (Cout, C0, 2, 2)
// Downsample weight read as (Cout, 4 C0) with K index sub * C0 + c
__device__ __forceinline__ void pack_pw_elem(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin,
                                             int s2d_C0, int64_t i, int rs = 0, int rc = 0) {
    const int co = (int)(i / Cin), ci = (int)(i % Cin);
    float v;
    if (rs) {  // the transposed 1x1 weight of an input-gradient convolution, read in place (load_taps9)
        v = w[(size_t)ci * rs + rc + co];
    } else if (s2d_C0) {
        const int sub = ci / s2d_C0, c = ci % s2d_C0;
        v = w[((size_t)co * s2d_C0 + c) * 4 + sub];
    } else {
        v = w[i];
    }
    const int chunk = ci / 16, cc = ci % 16, kq = cc / 4, j = cc % 4;
    const int t = co / 16, l15 = co % 16;
    packed[(((size_t)chunk * (Cout / 16) + t) * 64 + kq * 16 + l15) * 4 + j] = v;
}

__global__ void pack_pw_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int s2d_C0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)Cout * Cin) pack_pw_elem(w, packed, Cout, Cin, s2d_C0, i);
}
// init7_pack_weights: (64, Cin, 7, 7) -> [k padded to a multiple of 4][l15][t]
__global__ void pack_init7_kernel(const float* __restrict__ oihw, float* __restrict__ packed, int Cin, int KP) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KP * 64) return;
    const int k = i / 64, co = i % 64, K = 49 * Cin;
    packed[((size_t)k * 16 + (co & 15)) * 4 + (co >> 4)] = k < K ? oihw[(size_t)co * K + k] : 0.f;
}

// (c_n, Cout, K, K) <- 180-degree rotation + channel-role swap of rows [c_lo, c_lo + c_n) of an OIHW (Cout, Cin, K, K)
__device__ __forceinline__ void rot_transpose_elem(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                   int K, int c_lo, int64_t i) {
    const int kx = (int)(i % K), ky = (int)((i / K) % K), o = (int)((i / (K * K)) % Cout), c = (int)(i / ((int64_t)K * K * Cout));
    out[i] = w[(((size_t)o * Cin + c_lo + c) * K + (K - 1 - ky)) * K + (K - 1 - kx)];
}
__global__ void rot_transpose_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int K, int c_lo,
                                     int c_n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)c_n * Cout * K * K) rot_transpose_elem(w, out, Cout, Cin, K, c_lo, i);
}
// Downsample input gradient as a 1x1 convolution Cout -> 4C: out[(sub * C + c)][o] = w[o][c * 4 + sub]
__device__ __forceinline__ void s2d_transpose_elem(const float* __restrict__ w, float* __restrict__ out, int Cout, int C,
                                                   int64_t i) {
    const int o = (int)(i % Cout), c = (int)((i / Cout) % C), sub = (int)(i / ((int64_t)Cout * C));
    out[i] = w[(size_t)o * 4 * C + c * 4 + sub];
}

__global__ void s2d_transpose_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)4 * C * Cout) s2d_transpose_elem(w, out, Cout, C, i);
}
// all raw parameter copies of a re-pack (biases, gains, mem_kv, time MLP, the concatenated ResnetBlock.mlp rows) in ONE
// launch: blockIdx.y = table entry, the entry's floats are spread over blockIdx.x
struct ScatterEntry {
    long long src_off;
    float* dst;
    long long n;
};
__global__ void scatter_copy_kernel(const float* __restrict__ param, const ScatterEntry* __restrict__ table) {
    const ScatterEntry e = table[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < e.n; i += (long long)gridDim.x * blockDim.x)
        e.dst[i] = param[e.src_off + i];
}
int launch_scatter_copy(const float* param, const void* table_dev, int n_entries, long long max_n, hipStream_t s) {
    if (n_entries <= 0) return 0;
    const int bx = (int)std::max<long long>(1, std::min<long long>(64, (max_n + 1023) / 1024));
    hipLaunchKernelGGL(scatter_copy_kernel, dim3(bx, n_entries), dim3(256), 0, s, param,
                       static_cast<const ScatterEntry*>(table_dev));
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- grouped re-pack: one launch for a table of (source, destination, layout) jobs.  After an optimiser step the training
// loop rebuilds every packed buffer its shapes use in two launches (the rotated / transposed sources of the input-gradient
// layers first, then the packs) instead of one or two small launches in front of each convolution.
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    if (j.kind == PJ_ROT) {
        // 16 x 16 (cout, cin) pairs per workgroup through LDS: read with the lanes along cin (the source's inner index: 16 x
        // K^2 contiguous floats per cout), written with the lanes along cout (the destination's): both sides move runs of
        // 16 K^2 floats, where one thread per destination element read 36-byte pieces Cin * 36 bytes apart
        __shared__ float tile[256 * 9];
        const int K = j.a, KK = K * K, c_lo = j.b;
        const int c_n = (int)(j.n / ((int64_t)j.Cout * KK));
        const int tiles_c = (c_n + 15) / 16;
        const int lb = (int)(blockIdx.x - j.first_block);
        const int c0 = (lb % tiles_c) * 16, o0 = (lb / tiles_c) * 16;
        const int lo16 = threadIdx.x & 15, hi16 = threadIdx.x >> 4;
        {
            const int o = o0 + hi16, c = c0 + lo16;
            if (o < j.Cout && c < c_n) {
                const float* sp = j.src + ((size_t)o * j.Cin + c_lo + c) * KK;
                for (int t = 0; t < KK; ++t) tile[(hi16 * 16 + lo16) * KK + t] = sp[t];
            }
        }
        __syncthreads();
        {
            const int c = c0 + hi16, o = o0 + lo16;
            if (o < j.Cout && c < c_n) {
                float* dp = j.dst + ((size_t)c * j.Cout + o) * KK;
                for (int t = 0; t < KK; ++t) dp[t] = tile[(lo16 * 16 + hi16) * KK + (KK - 1 - t)];  // rotated by 180 degrees
            }
        }
        return;
    }
    const int64_t t = (int64_t)(blockIdx.x - j.first_block) * 256 + threadIdx.x;
    if (t >= j.n) return;
    // The element functions take (cout, cin) as i = cout * Cin + cin.  Here consecutive threads take the elements that are
    // neighbours in the PACKED layout, so every plane of a Winograd pack leaves as 256-byte runs (one thread per OIHW
    // element in OIHW order wrote 32-byte pieces).
    switch (j.kind) {
        case PJ_S2D_T: s2d_transpose_elem(j.src, j.dst, j.Cout, j.Cin, t); break;
        case PJ_WINO: {  // [chunk of 8][xi][Cout][8]
            const int cc = (int)(t & 7), co = (int)((t >> 3) % j.Cout), chunk = (int)((t >> 3) / j.Cout);
            pack_wino_elem(j.src, j.dst, j.Cout, j.Cin, (int64_t)co * j.Cin + chunk * 8 + cc, j.rs, j.rc);
            break;
        }
        case PJ_WINO4:
        case PJ_UPWINO: {  // [chunk][..][cout tile of 64][kq 4][n 16][gq 4][st 2]
            const int st = (int)(t & 1), gq = (int)((t >> 1) & 3), n = (int)((t >> 3) & 15), kq = (int)((t >> 7) & 3);
            const int64_t rest = t >> 9;
            const int tiles = j.Cout / 64, ct = (int)(rest % tiles), chunk = (int)(rest / tiles);
            const int64_t i = (int64_t)(ct * 64 + 16 * gq + n) * j.Cin + chunk * 8 + 2 * kq + st;
            if (j.kind == PJ_WINO4) pack_wino4_elem(j.src, j.dst, j.Cout, j.Cin, i, j.rs, j.rc);
            else pack_upwino_elem(j.src, j.dst, j.Cout, j.Cin, i, j.rs, j.rc);
            break;
        }
        case PJ_PW: {  // [chunk of 16][cout tile of 16][kq 4][l15 16][j 4]
            const int jj = (int)(t & 3), l15 = (int)((t >> 2) & 15), kq = (int)((t >> 6) & 3);
            const int64_t rest = t >> 8;
            const int tiles = j.Cout / 16, t16 = (int)(rest % tiles), chunk = (int)(rest / tiles);
            pack_pw_elem(j.src, j.dst, j.Cout, j.Cin, j.a, (int64_t)(t16 * 16 + l15) * j.Cin + chunk * 16 + 4 * kq + jj, j.rs, j.rc);
            break;
        }
        default: j.dst[t] = j.src[t]; break;  // PJ_COPY
    }
}
int pack_jobs_prefix(std::vector<PackJob>& jobs) {
    int first = 0;
    for (PackJob& j : jobs) {
        j.first_block = first;
        if (j.kind == PJ_ROT) {  // one workgroup per 16 x 16 (cout, cin) pairs (pack_jobs_kernel)
            const long long c_n = j.n / ((long long)j.Cout * j.a * j.a);
            first += (int)(((c_n + 15) / 16) * ((j.Cout + 15) / 16));
        } else {
            first += (int)((j.n + 255) / 256);
        }
    }
    return first;
}
int launch_pack_jobs(const PackJob* jobs_dev, int n_jobs, int total_blocks, hipStream_t s) {
    if (n_jobs <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(pack_jobs_kernel, dim3(total_blocks), dim3(256), 0, s, jobs_dev, n_jobs);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

#define DM_PK_LAUNCH(kernel, n, ...)                                                                     \
    do {                                                                                                   \
        const int64_t n_ = (n);                                                                            \
        if (n_ > 0) hipLaunchKernelGGL(kernel, dim3((unsigned)((n_ + 255) / 256)), dim3(256), 0, s, __VA_ARGS__); \
        DM_CHECK_HIP(hipGetLastError());                                                                   \
    } while (0)

int launch_pack_direct(const float* oihw, float* packed, int Cout, int C0, int C1, int KH, int KW, hipStream_t s) {
    const int CK = conv_ck_for(C0, C1);
    const int64_t n = (int64_t)conv_packed_floats(Cout, C0, C1, KH, KW);
    DM_PK_LAUNCH(pack_direct_kernel, n, oihw, packed, Cout, C0, C1, KH, KW, CK, n);
    return 0;
}
int launch_fold_taps(const float* oihw, float* w2, int Cout, int Cin, int py, int px, hipStream_t s) {
    DM_PK_LAUNCH(fold_taps_kernel, (int64_t)Cout * Cin, oihw, w2, (int64_t)Cout * Cin, py, px);
    return 0;
}
int launch_pack_wino(const float* oihw, float* packed, int Cout, int Cin, hipStream_t s) {
    DM_PK_LAUNCH(pack_wino_kernel, (int64_t)Cout * Cin, oihw, packed, Cout, Cin);
    return 0;
}
int launch_pack_wino4(const float* oihw, float* packed, int Cout, int Cin, hipStream_t s) {
    DM_PK_LAUNCH(pack_wino4_kernel, (int64_t)Cout * Cin, oihw, packed, Cout, Cin);
    return 0;
}
int launch_pack_upwino(const float* oihw, float* packed, int Cout, int Cin, hipStream_t s) {
    DM_PK_LAUNCH(pack_upwino_kernel, (int64_t)Cout * Cin, oihw, packed, Cout, Cin);
    return 0;
}
int launch_pack_pw(const float* w, float* packed, int Cout, int Cin, int s2d_C0, hipStream_t s) {
    DM_PK_LAUNCH(pack_pw_kernel, (int64_t)Cout * Cin, w, packed, Cout, Cin, s2d_C0);
    return 0;
}
int launch_pack_init7(const float* oihw, float* packed, int Cin, hipStream_t s) {
    const int KP = (int)(init7_packed_floats(Cin) / 64);
    DM_PK_LAUNCH(pack_init7_kernel, (int64_t)KP * 64, oihw, packed, Cin, KP);
    return 0;
}
int launch_rot_transpose(const float* w, float* out, int Cout, int Cin, int K, int c_lo, int c_n, hipStream_t s) {
    DM_PK_LAUNCH(rot_transpose_kernel, (int64_t)c_n * Cout * K * K, w, out, Cout, Cin, K, c_lo, c_n);
    return 0;
}
int launch_s2d_transpose(const float* w, float* out, int Cout, int C, hipStream_t s) {
    DM_PK_LAUNCH(s2d_transpose_kernel, (int64_t)4 * C * Cout, w, out, Cout, C);
    return 0;
}

}  // namespace dm

// Backward of the two attention cores for the training step (autograd of DD/denoising_diffusion.py:179-192
// LinearAttention and :221-226 + DD/attend.py:109-124 Attention).  dim_head = 32, 4 learned memory key/values.
// qkv / dqkv are NHWC token rows [q(h,d) | k(h,d) | v(h,d)]; the cores hold < 2 % of a step's FLOPs, so these are plain
// LDS-tiled VALU kernels with a fixed summation order (no atomics).
//
// LinearAttention:  p = softmax_d(q);  qs = p * scale;  ks = softmax_tokens(k_ext);  ctx[d][e] = sum_j ks[d][j] v_ext[e][j];
//                   out[e][i] = sum_d ctx[d][e] qs[d][i]
//   dctx[d][e] = sum_i qs[d][i] dout[e][i]         dqs[d][i] = sum_e ctx[d][e] dout[e][i]
//   dq[d][i]   = scale p[d][i] (dqs[d][i] - sum_d' p[d'][i] dqs[d'][i])
//   dks[d][j]  = sum_e dctx[d][e] v[e][j]          dv[e][j] = sum_d ks[d][j] dctx[d][e]
//   dk[d][j]   = ks[d][j] (dks[d][j] - S[d]),      S[d] = sum_j ks dks = sum_e dctx[d][e] ctx[d][e]
#include "dm_common.h"

#include <algorithm>
#include <cstdlib>

namespace dm {

constexpr int BDH = 32;
constexpr int NMEM = 4;
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 make_f32x4(float a, float b, float c, float d) { return f32x4{a, b, c, d}; }

constexpr int LSTR = BDH + 4;  // LDS row stride of a token row: 16-byte aligned, 128-bit accesses of 8 consecutive lanes cover all banks

// part 1: grid (ceil(n / 64), heads, B), one wave (lane = token): dq, and this block's share of dctx.  One head per block
// keeps the tile at 22 KB, so seven blocks share a CU and hide each other's row loads.
__global__ __launch_bounds__(64) void linattn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                           const float* __restrict__ dout, float* __restrict__ dqkv,
                                                           float* __restrict__ dctx_part, int n, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) float cs[BDH * BDH];   // ctx of this (image, head)
    __shared__ __attribute__((aligned(16))) float ps[64 * LSTR];   // scale * p
    __shared__ __attribute__((aligned(16))) float ds[64 * LSTR];   // dout
    const int blk = blockIdx.x, nblk = gridDim.x, h = blockIdx.y, b = blockIdx.z;
    const int tl = threadIdx.x;
    const int tok = blk * 64 + tl;
    const int ld = 3 * heads * BDH, hid = heads * BDH;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(ctx + (size_t)(b * heads + h) * BDH * BDH);
        for (int i = tl; i < BDH * BDH / 4; i += 64) reinterpret_cast<f32x4*>(cs)[i] = src[i];
    }
    float q[BDH], dq[BDH], dov[BDH];
    const bool ok = tok < n;
    const f32x4* qp = reinterpret_cast<const f32x4*>(qkv + ((size_t)b * n + (ok ? tok : 0)) * ld + h * BDH);
    const f32x4* dp = reinterpret_cast<const f32x4*>(dout + ((size_t)b * n + (ok ? tok : 0)) * hid + h * BDH);
#pragma unroll
    for (int j = 0; j < BDH / 4; ++j) {
        const f32x4 a = qp[j], c = dp[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q[4 * j + i] = a[i];
            dov[4 * j + i] = ok ? c[i] : 0.f;
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int d = 0; d < BDH; ++d) m = fmaxf(m, q[d]);
    float sum = 0.f;
#pragma unroll
    for (int d = 0; d < BDH; ++d) {
        q[d] = __expf(q[d] - m);
        sum += q[d];
    }
    const float inv = 1.0f / sum;
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < BDH; ++d) {
        q[d] *= inv;  // p
        float s = 0.f;
        const float* cr = cs + d * BDH;
#pragma unroll
        for (int e = 0; e < BDH; ++e) s += cr[e] * dov[e];
        dq[d] = s;  // dqs
        dot += q[d] * s;
    }
    if (ok) {
        f32x4* o = reinterpret_cast<f32x4*>(dqkv + ((size_t)b * n + tok) * ld + h * BDH);
#pragma unroll
        for (int j = 0; j < BDH / 4; ++j)
            o[j] = make_f32x4(scale * q[4 * j] * (dq[4 * j] - dot), scale * q[4 * j + 1] * (dq[4 * j + 1] - dot),
                              scale * q[4 * j + 2] * (dq[4 * j + 2] - dot), scale * q[4 * j + 3] * (dq[4 * j + 3] - dot));
    }
#pragma unroll
    for (int j = 0; j < BDH / 4; ++j) {
        const float z = ok ? scale : 0.f;
        *reinterpret_cast<f32x4*>(ps + tl * LSTR + 4 * j) =
            make_f32x4(z * q[4 * j], z * q[4 * j + 1], z * q[4 * j + 2], z * q[4 * j + 3]);
        *reinterpret_cast<f32x4*>(ds + tl * LSTR + 4 * j) = make_f32x4(dov[4 * j], dov[4 * j + 1], dov[4 * j + 2], dov[4 * j + 3]);
    }
    __syncthreads();
    // dctx share of this block: lane -> (d, 16 e's), tokens in order
    {
        const int d = tl >> 1, e0 = (tl & 1) * 16;
        float acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int t = 0; t < 64; ++t) {
            const float pv = ps[t * LSTR + d];
            const f32x4* dr = reinterpret_cast<const f32x4*>(ds + t * LSTR + e0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = dr[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * j + i] += pv * v[i];
            }
        }
        f32x4* o = reinterpret_cast<f32x4*>(dctx_part + ((((size_t)b * nblk + blk) * heads + h) * BDH + d) * BDH + e0);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = make_f32x4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    }
}

// part 2a: grid (heads, B), 256 threads: dctx = sum of the block shares (left in share 0), per d the softmax statistics of
// k over the tokens (memory tokens first, as the reference concatenates them) and S[d] -> stats (B, heads, 3, 32), and the
// gradients of this image's 4 memory key/value tokens
__global__ __launch_bounds__(256) void linattn_bwd_stats_kernel(const float* __restrict__ qkv, const float* __restrict__ mem_kv,
                                                                const float* __restrict__ ctx, float* __restrict__ dctx_part,
                                                                int nblk, float* __restrict__ stats,
                                                                float* __restrict__ dmem_part,
                                                                const float* __restrict__ kstats, int n, int heads) {
    __shared__ float dctx[BDH][BDH + 1];
    __shared__ float kmax[BDH], kinv[BDH], S[BDH];
    __shared__ float red[8][BDH];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ld = 3 * heads * BDH;
    const float* kbase = qkv + (size_t)b * n * ld + heads * BDH + h * BDH;
    const float* mk = mem_kv + (size_t)h * BDH * NMEM;  // [d][j]
    {  // the block shares of dctx, summed in block order; the 4 elements of a thread x 4 shares are in flight together
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const float* src = dctx_part + ((size_t)b * nblk * heads + h) * BDH * BDH + tid;
        const size_t kst = (size_t)heads * BDH * BDH;
        int k = 0;
        for (; k + 4 <= nblk; k += 4) {
            float v[4][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[kk][j] = src[(k + kk) * kst + 256 * j];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += v[kk][j];
        }
        for (; k < nblk; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += src[k * kst + 256 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            dctx[i >> 5][i & 31] = s[j];
            dctx_part[((size_t)b * nblk * heads + h) * BDH * BDH + i] = s[j];
        }
    }
    // kstats (max and sum over tokens of exp(k - max), per column) as the forward context kernel left them on the tape; without
    // them (kstats == nullptr) the two passes over the n key rows are redone here, 8 lanes deep per column
    if (kstats) {
        if (tid < BDH) {
            kmax[tid] = kstats[(size_t)(b * heads + h) * 2 * BDH + tid];
            red[0][tid] = kstats[(size_t)(b * heads + h) * 2 * BDH + BDH + tid];
        } else if (tid < 8 * BDH) {
            red[tid >> 5][tid & 31] = 0.f;
        }
        __syncthreads();
    } else {
        const int d = tid & 31, part = tid >> 5;
        float m = part < NMEM ? mk[d * NMEM + part] : -INFINITY;
        {
            int t = part;
            for (; t + 56 < n; t += 64) {  // 8 independent row loads in flight
                float kv[8];
    #pragma unroll
                for (int j = 0; j < 8; ++j) kv[j] = kbase[(size_t)(t + 8 * j) * ld + d];
    #pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, kv[j]);
            }
            for (; t < n; t += 8) m = fmaxf(m, kbase[(size_t)t * ld + d]);
        }
        red[part][d] = m;
        __syncthreads();
        if (tid < BDH) {
            float mm = red[0][tid];
            for (int q = 1; q < 8; ++q) mm = fmaxf(mm, red[q][tid]);
            kmax[tid] = mm;
        }
        __syncthreads();
        const float km = kmax[d];
        float s = part < NMEM ? __expf(mk[d * NMEM + part] - km) : 0.f;
        {
            int t = part;
            for (; t + 56 < n; t += 64) {
                float kv[8];
    #pragma unroll
                for (int j = 0; j < 8; ++j) kv[j] = kbase[(size_t)(t + 8 * j) * ld + d];
    #pragma unroll
                for (int j = 0; j < 8; ++j) s += __expf(kv[j] - km);
            }
            for (; t < n; t += 8) s += __expf(kbase[(size_t)t * ld + d] - km);
        }
        __syncthreads();
        red[part][d] = s;
        __syncthreads();
    }
    if (tid < BDH) {
        float ss = 0.f;
        for (int q = 0; q < 8; ++q) ss += red[q][tid];
        const float* cr = ctx + ((size_t)(b * heads + h) * BDH + tid) * BDH;
        float sd = 0.f;
        for (int e = 0; e < BDH; ++e) sd += dctx[tid][e] * cr[e];
        float* st = stats + (size_t)(b * heads + h) * 3 * BDH;
        st[tid] = kmax[tid];
        st[BDH + tid] = kinv[tid] = 1.0f / ss;
        st[2 * BDH + tid] = S[tid] = sd;
    }
    __syncthreads();
    if (tid < NMEM * BDH) {  // memory tokens: thread (t, x) forms dk[d = x][t] and dv[e = x][t]
        const int t = tid >> 5, x = tid & 31;
        const float* mv = mem_kv + (size_t)(heads + h) * BDH * NMEM;  // [e][j]
        float sk = 0.f, sv = 0.f;
        for (int j = 0; j < BDH; ++j) {
            sk += dctx[x][j] * mv[j * NMEM + t];
            sv += __expf(mk[j * NMEM + t] - kmax[j]) * kinv[j] * dctx[j][x];
        }
        float* o = dmem_part + (size_t)b * 2 * heads * BDH * NMEM;
        o[((size_t)h * BDH + x) * NMEM + t] = __expf(mk[x * NMEM + t] - kmax[x]) * kinv[x] * (sk - S[x]);
        o[((size_t)(heads + h) * BDH + x) * NMEM + t] = sv;
    }
}

// part 2b: grid (ceil(n / 64), heads, B), one wave (lane = token): dk, dv.  k and v rows sit in registers; dk overwrites k
// row by row.
__global__ __launch_bounds__(64, 3) void linattn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx_part,
                                                            int nblk, const float* __restrict__ stats, float* __restrict__ dqkv,
                                                            int n, int heads) {
    __shared__ __attribute__((aligned(16))) float dctx[BDH * BDH];
    __shared__ float st[3 * BDH];
    const int h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int ld = 3 * heads * BDH;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(dctx_part + ((size_t)b * nblk * heads + h) * BDH * BDH);
        for (int i = lane; i < BDH * BDH / 4; i += 64) reinterpret_cast<f32x4*>(dctx)[i] = src[i];
        for (int i = lane; i < 3 * BDH; i += 64) st[i] = stats[(size_t)(b * heads + h) * 3 * BDH + i];
    }
    __syncthreads();
    const int t = blockIdx.x * 64 + lane;
    if (t >= n) return;
    // two passes over dctx, so that only two 32-vectors and one dctx row are live at a time (one pass holding k, v, dv and
    // the prefetched rows spills at 3 waves per SIMD):  dv[e] = sum_d ks[d] dctx[d][e], then dk[d] = ks[d] (dctx[d] . v - S[d])
    float kk[BDH], vv[BDH];
    const f32x4* kp = reinterpret_cast<const f32x4*>(qkv + ((size_t)b * n + t) * ld + heads * BDH + h * BDH);
    const f32x4* vp = reinterpret_cast<const f32x4*>(qkv + ((size_t)b * n + t) * ld + 2 * heads * BDH + h * BDH);
    f32x4* ok = reinterpret_cast<f32x4*>(dqkv + ((size_t)b * n + t) * ld + heads * BDH + h * BDH);
    f32x4* ov = reinterpret_cast<f32x4*>(dqkv + ((size_t)b * n + t) * ld + 2 * heads * BDH + h * BDH);
#pragma unroll
    for (int j = 0; j < BDH / 4; ++j) {
        const f32x4 a = kp[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kk[4 * j + i] = __expf(a[i] - st[4 * j + i]) * st[BDH + 4 * j + i];  // ks
            vv[4 * j + i] = 0.f;                                                 // dv
        }
    }
#pragma unroll
    for (int d = 0; d < BDH; ++d) {
#pragma unroll
        for (int e = 0; e < BDH; ++e) vv[e] += kk[d] * dctx[d * BDH + e];
        asm volatile("" ::: "memory");  // one dctx row in flight
    }
#pragma unroll
    for (int j = 0; j < BDH / 4; ++j) ov[j] = make_f32x4(vv[4 * j], vv[4 * j + 1], vv[4 * j + 2], vv[4 * j + 3]);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < BDH / 4; ++j) {
        const f32x4 c = vp[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) vv[4 * j + i] = c[i];
    }
#pragma unroll
    for (int d = 0; d < BDH; ++d) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < BDH; ++e) s += dctx[d * BDH + e] * vv[e];
        kk[d] *= s - st[2 * BDH + d];  // dk
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int j = 0; j < BDH / 4; ++j) ok[j] = make_f32x4(kk[4 * j], kk[4 * j + 1], kk[4 * j + 2], kk[4 * j + 3]);
}

// ---- the same two passes on the matrix core (v_mfma_f32_32x32x2_f32), what the training step runs.  Every product of the
// backward pass is a (tokens x 32) x (32 x 32) GEMM; computed TRANSPOSED -- the 32 x 32 matrix (ctx, dctx) is the A operand, the
// token rows the B operand -- the result D[i][j = token] leaves lane (token, half) with the 16 columns
//     dset(r) = (r & 3) + 8 (r >> 2) + 4 half,   r = 0 .. 15      (four float4 chunks 2 m + half of the token's 32-float row)
// of its token, and with the K index enumerated in the same order (k-step s of half h = column dset(s)) the B operand of the
// next product is exactly those registers: a lane loads 4 chunks of q / dout / k / v, keeps everything row-wise (softmax over
// the columns, the dot products) in registers plus one exchange with lane ^ 32, and stores 4 chunks.  One wave = 64 tokens (two
// 32-token tiles), 64 MFMAs per pass where the VALU form issued ~2000 FMAs and ~600 LDS broadcasts per lane.
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(64) void linattn_bwd_q_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                                const float* __restrict__ dout, float* __restrict__ dqkv,
                                                                float* __restrict__ dctx_part, int n, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) float ps[64 * LSTR];  // scale * p, [token][d]
    __shared__ __attribute__((aligned(16))) float ds[64 * LSTR];  // dout,      [token][e]
    const int blk = blockIdx.x, nblk = gridDim.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int ld = 3 * heads * BDH, hid = heads * BDH;
    // A operand of dqs^T = ctx . dout^T: ctx[d = l31][e = dset(s)]
    float actx[16];
    {
        const f32x4* cr = reinterpret_cast<const f32x4*>(ctx + ((size_t)(b * heads + h) * BDH + l31) * BDH);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 v = cr[2 * m + half];
#pragma unroll
            for (int i = 0; i < 4; ++i) actx[4 * m + i] = v[i];
        }
    }
    float p[2][16], dv[2][16];
    bool okt[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int tok = blk * 64 + nt * 32 + l31;
        okt[nt] = tok < n;
        const size_t row = (size_t)b * n + (okt[nt] ? tok : 0);
        const f32x4* qp = reinterpret_cast<const f32x4*>(qkv + row * ld + h * BDH);
        const f32x4* dp = reinterpret_cast<const f32x4*>(dout + row * hid + h * BDH);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 a = qp[2 * m + half], c = dp[2 * m + half];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p[nt][4 * m + i] = a[i];
                dv[nt][4 * m + i] = okt[nt] ? c[i] : 0.f;
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, p[nt][r]);
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[nt][r] = __expf(p[nt][r] - m);
            sum += p[nt][r];
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[nt][r] *= inv;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(actx[s], dv[nt][s], acc, 0, 0, 0);
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) dot += p[nt][r] * acc[r];
        dot += __shfl_xor(dot, 32);
        const int tok = blk * 64 + nt * 32 + l31;
        if (okt[nt]) {
            f32x4* o = reinterpret_cast<f32x4*>(dqkv + ((size_t)b * n + tok) * ld + h * BDH);
#pragma unroll
            for (int m4 = 0; m4 < 4; ++m4)
                o[2 * m4 + half] = make_f32x4(scale * p[nt][4 * m4] * (acc[4 * m4] - dot),
                                              scale * p[nt][4 * m4 + 1] * (acc[4 * m4 + 1] - dot),
                                              scale * p[nt][4 * m4 + 2] * (acc[4 * m4 + 2] - dot),
                                              scale * p[nt][4 * m4 + 3] * (acc[4 * m4 + 3] - dot));
        }
        const float z = okt[nt] ? scale : 0.f;
        float* pr = ps + (nt * 32 + l31) * LSTR;
        float* dr = ds + (nt * 32 + l31) * LSTR;
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
            *reinterpret_cast<f32x4*>(pr + 4 * (2 * m4 + half)) =
                make_f32x4(z * p[nt][4 * m4], z * p[nt][4 * m4 + 1], z * p[nt][4 * m4 + 2], z * p[nt][4 * m4 + 3]);
            *reinterpret_cast<f32x4*>(dr + 4 * (2 * m4 + half)) =
                make_f32x4(dv[nt][4 * m4], dv[nt][4 * m4 + 1], dv[nt][4 * m4 + 2], dv[nt][4 * m4 + 3]);
        }
    }
    __syncthreads();
    // dctx share of this block: D[i = d][j = e] = sum over the 64 tokens (k-step s, half -> token 2 s + half) ps[t][d] ds[t][e]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ps[(2 * s + half) * LSTR + l31], ds[(2 * s + half) * LSTR + l31], acc, 0, 0, 0);
    float* o = dctx_part + (((size_t)b * nblk + blk) * heads + h) * BDH * BDH;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * half) * BDH + l31] = acc[r];
}

__global__ __launch_bounds__(64) void linattn_bwd_kv_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx_part,
                                                                 int nblk, const float* __restrict__ stats,
                                                                 float* __restrict__ dqkv, int n, int heads) {
    const int blk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int ld = 3 * heads * BDH;
    const float* dctx = dctx_part + ((size_t)b * nblk * heads + h) * BDH * BDH;  // share 0 holds the sum
    // A operands: dv^T = dctx^T . ks^T needs dctx[d = dset(s)][e = l31]; dks^T = dctx . v^T needs dctx[d = l31][e = dset(s)]
    float a3[16], a4[16], kmax[16], kinv[16], S[16];
    {
        const f32x4* row = reinterpret_cast<const f32x4*>(dctx + l31 * BDH);
        const f32x4* st = reinterpret_cast<const f32x4*>(stats + (size_t)(b * heads + h) * 3 * BDH);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 v = row[2 * m + half], s0 = st[2 * m + half], s1 = st[8 + 2 * m + half], s2 = st[16 + 2 * m + half];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a4[4 * m + i] = v[i];
                a3[4 * m + i] = dctx[(8 * m + 4 * half + i) * BDH + l31];
                kmax[4 * m + i] = s0[i];
                kinv[4 * m + i] = s1[i];
                S[4 * m + i] = s2[i];
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int tok = blk * 64 + nt * 32 + l31;
        const bool ok = tok < n;
        const size_t row = (size_t)b * n + (ok ? tok : 0);
        const f32x4* kp = reinterpret_cast<const f32x4*>(qkv + row * ld + heads * BDH + h * BDH);
        const f32x4* vp = reinterpret_cast<const f32x4*>(qkv + row * ld + 2 * heads * BDH + h * BDH);
        float ks[16], vv[16];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 a = kp[2 * m + half], c = vp[2 * m + half];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ks[4 * m + i] = __expf(a[i] - kmax[4 * m + i]) * kinv[4 * m + i];
                vv[4 * m + i] = c[i];
            }
        }
        f32x16 dvt, dkt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dvt[r] = dkt[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            dvt = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[s], ks[s], dvt, 0, 0, 0);
            dkt = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], vv[s], dkt, 0, 0, 0);
        }
        if (ok) {
            f32x4* okp = reinterpret_cast<f32x4*>(dqkv + row * ld + heads * BDH + h * BDH);
            f32x4* ovp = reinterpret_cast<f32x4*>(dqkv + row * ld + 2 * heads * BDH + h * BDH);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ovp[2 * m + half] = make_f32x4(dvt[4 * m], dvt[4 * m + 1], dvt[4 * m + 2], dvt[4 * m + 3]);
                okp[2 * m + half] = make_f32x4(ks[4 * m] * (dkt[4 * m] - S[4 * m]), ks[4 * m + 1] * (dkt[4 * m + 1] - S[4 * m + 1]),
                                               ks[4 * m + 2] * (dkt[4 * m + 2] - S[4 * m + 2]),
                                               ks[4 * m + 3] * (dkt[4 * m + 3] - S[4 * m + 3]));
            }
        }
    }
}

// block shares of dctx (the sum lands in share 0) + the per-(image, head) statistics
size_t linattn_bwd_ws_floats(int B, int n, int heads) {
    return (size_t)B * ((n + 63) / 64) * heads * BDH * BDH + (size_t)B * heads * 3 * BDH;
}

// qkv (B, n, 3*heads*32), ctx (B, heads, 32, 32) as the forward core left it, dout (B, n, heads*32) -> dqkv (same shape as
// qkv), dmem_part (B, 2, heads, 32, 4) per-image memory key/value gradients (the caller sums over B)
int launch_linear_attention_core_bwd(const float* qkv, const float* mem_kv, const float* ctx, const float* dout, float* ws,
                                     float* dqkv, float* dmem_part, int B, int n, int heads, int dh, hipStream_t s,
                                     const float* kstats) {
    DM_REQUIRE(dh == BDH && heads >= 1 && heads <= 16, "linear attention backward: dim_head 32");
    DM_REQUIRE(B <= 65535 && n >= 1, "linear attention backward: batch");
    const int nblk = (n + 63) / 64;
    float* stats = ws + (size_t)B * nblk * heads * BDH * BDH;
    static const bool valu = std::getenv("DM_LINATTN_BWD_VALU") != nullptr;  // the lane-per-token VALU kernels (A/B, forced-path test)
    if (valu)
        hipLaunchKernelGGL(linattn_bwd_q_kernel, dim3(nblk, heads, B), dim3(64), 0, s, qkv, ctx, dout, dqkv, ws, n, heads,
                           1.0f / sqrtf((float)dh));
    else
        hipLaunchKernelGGL(linattn_bwd_q_mfma_kernel, dim3(nblk, heads, B), dim3(64), 0, s, qkv, ctx, dout, dqkv, ws, n, heads,
                           1.0f / sqrtf((float)dh));
    DM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(linattn_bwd_stats_kernel, dim3(heads, B), dim3(256), 0, s, qkv, mem_kv, ctx, ws, nblk, stats, dmem_part,
                       kstats, n, heads);
    DM_CHECK_HIP(hipGetLastError());
    if (valu)
        hipLaunchKernelGGL(linattn_bwd_kv_kernel, dim3(nblk, heads, B), dim3(64), 0, s, qkv, ws, nblk, stats, dqkv, n, heads);
    else
        hipLaunchKernelGGL(linattn_bwd_kv_mfma_kernel, dim3(nblk, heads, B), dim3(64), 0, s, qkv, ws, nblk, stats, dqkv, n, heads);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Full attention: P = softmax_j(scale q_i k_j), o_i = sum_j P_ij v_j (keys = n_mem memory rows, then the nk key rows)
//   dP_ij = do_i . v_j;  D_i = do_i . o_i;  dS_ij = P_ij (dP_ij - D_i);  dq_i = scale sum_j dS_ij k_j;
//   dk_j = scale sum_i dS_ij q_i;  dv_j = sum_i P_ij do_i
// grid (heads, B); phase 1: thread = query (row statistics, dq); phase 2: thread = key (dk, dv), fixed order over queries.
// Self-attention (Attention :215-229): q, k, v are column blocks of one qkv tensor and 4 learned memory rows come first;
// cross-attention (DD/denoising_diffusion_text_conditional.py:54-78): k, v are projections of the text context, no memory.
// ---------------------------------------------------------------------------------------
struct AttnBwdParams {
    const float *q, *k, *v;      // rows of ldq / ldk floats per token, head h at column h * 32
    const float *mem_k, *mem_v;  // (heads, n_mem, 32) or nullptr
    const float* dout;           // (B, nq, heads * 32)
    float *dq, *dk, *dv;         // same strides as q / k / v
    float* dmem_part;            // (B, 2, heads, n_mem, 32) or nullptr
    int ldq, ldk, nq, nk, n_mem, heads;
    float scale;
};

// CACHE (short sequences: nq * (nk + n_mem) <= 4096, e.g. the 16 tokens + 4 memory rows of the 32x32 U-Net's bottleneck): the
// scores q_i . k_j and dP_ij are formed ONCE and kept in LDS -- the plain form recomputes the 32-wide dot products in each of
// its three passes over the keys and again per (key, query) in the second phase, and with one thread per query that serial
// chain IS the kernel's time (42 us at any batch for 16 queries); same summation order within every dot product.
template <bool CACHE>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int nkt = p.nk + p.n_mem, n = p.nq;
    float* Ks = sm;                       // [nkt][33]
    float* Vs = Ks + nkt * (BDH + 1);     // [nkt][33]
    float* Qs = Vs + nkt * (BDH + 1);     // [n][33]
    float* Ds = Qs + n * (BDH + 1);       // [n][33]  dout
    float* rm = Ds + n * (BDH + 1);       // [n] row max
    float* rl = rm + n;                   // [n] 1 / row sum
    float* rD = rl + n;                   // [n] D_i
    float* Pc = rD + n;                   // CACHE: [n][nkt] scores -> exp -> P_ij
    float* Sc = Pc + (CACHE ? n * nkt : 0);  // CACHE: [n][nkt] dP_ij -> dS_ij
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int hid = p.heads * BDH;
    const float scale = p.scale;
    for (int i = tid; i < nkt * BDH; i += 256) {
        const int j = i >> 5, d = i & 31;
        float kv, vv;
        if (j < p.n_mem) {
            kv = p.mem_k[((size_t)h * p.n_mem + j) * BDH + d];
            vv = p.mem_v[((size_t)h * p.n_mem + j) * BDH + d];
        } else {
            const size_t row = ((size_t)b * p.nk + (j - p.n_mem)) * p.ldk + h * BDH + d;
            kv = p.k[row];
            vv = p.v[row];
        }
        Ks[j * (BDH + 1) + d] = kv;
        Vs[j * (BDH + 1) + d] = vv;
    }
    for (int i = tid; i < n * BDH; i += 256) {
        const int t = i >> 5, d = i & 31;
        Qs[t * (BDH + 1) + d] = p.q[((size_t)b * n + t) * p.ldq + h * BDH + d];
        Ds[t * (BDH + 1) + d] = p.dout[((size_t)b * n + t) * hid + h * BDH + d];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        float q[BDH], dov[BDH];
#pragma unroll
        for (int d = 0; d < BDH; ++d) {
            q[d] = Qs[i * (BDH + 1) + d];
            dov[d] = Ds[i * (BDH + 1) + d];
        }
        float m = -INFINITY, l = 0.f, D = 0.f, linv;
        float dq[BDH];
#pragma unroll
        for (int d = 0; d < BDH; ++d) dq[d] = 0.f;
        if constexpr (CACHE) {
            float* Pi = Pc + i * nkt;
            float* Si = Sc + i * nkt;
            for (int j = 0; j < nkt; ++j) {
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < BDH; ++d) {
                    sc += q[d] * Ks[j * (BDH + 1) + d];
                    dp += dov[d] * Vs[j * (BDH + 1) + d];
                }
                Pi[j] = sc * scale;
                Si[j] = dp;
                m = fmaxf(m, sc * scale);
            }
            for (int j = 0; j < nkt; ++j) {
                const float e = __expf(Pi[j] - m);
                Pi[j] = e;
                l += e;
                D += e * Si[j];
            }
            linv = 1.0f / l;
            D *= linv;
            for (int j = 0; j < nkt; ++j) {
                const float P = Pi[j] * linv;
                const float dS = P * (Si[j] - D);
                Pi[j] = P;
                Si[j] = dS;
#pragma unroll
                for (int d = 0; d < BDH; ++d) dq[d] += dS * Ks[j * (BDH + 1) + d];
            }
        } else {
            for (int j = 0; j < nkt; ++j) {
                float sc = 0.f;
#pragma unroll
                for (int d = 0; d < BDH; ++d) sc += q[d] * Ks[j * (BDH + 1) + d];
                m = fmaxf(m, sc * scale);
            }
            for (int j = 0; j < nkt; ++j) {
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < BDH; ++d) {
                    sc += q[d] * Ks[j * (BDH + 1) + d];
                    dp += dov[d] * Vs[j * (BDH + 1) + d];
                }
                const float e = __expf(sc * scale - m);
                l += e;
                D += e * dp;
            }
            linv = 1.0f / l;
            D *= linv;
            for (int j = 0; j < nkt; ++j) {
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < BDH; ++d) {
                    sc += q[d] * Ks[j * (BDH + 1) + d];
                    dp += dov[d] * Vs[j * (BDH + 1) + d];
                }
                const float dS = __expf(sc * scale - m) * linv * (dp - D);
#pragma unroll
                for (int d = 0; d < BDH; ++d) dq[d] += dS * Ks[j * (BDH + 1) + d];
            }
        }
        rm[i] = m;
        rl[i] = linv;
        rD[i] = D;
        float* o = p.dq + ((size_t)b * n + i) * p.ldq + h * BDH;
#pragma unroll
        for (int d = 0; d < BDH; ++d) o[d] = scale * dq[d];
    }
    __syncthreads();
    for (int j = tid; j < nkt; j += 256) {
        float kk[BDH], vv[BDH], dk[BDH], dv[BDH];
#pragma unroll
        for (int d = 0; d < BDH; ++d) {
            kk[d] = Ks[j * (BDH + 1) + d];
            vv[d] = Vs[j * (BDH + 1) + d];
            dk[d] = 0.f;
            dv[d] = 0.f;
        }
        for (int i = 0; i < n; ++i) {
            float P, dS;
            if constexpr (CACHE) {
                P = Pc[i * nkt + j];
                dS = Sc[i * nkt + j];
            } else {
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < BDH; ++d) {
                    sc += Qs[i * (BDH + 1) + d] * kk[d];
                    dp += Ds[i * (BDH + 1) + d] * vv[d];
                }
                P = __expf(sc * scale - rm[i]) * rl[i];
                dS = P * (dp - rD[i]);
            }
#pragma unroll
            for (int d = 0; d < BDH; ++d) {
                dk[d] += dS * Qs[i * (BDH + 1) + d];
                dv[d] += P * Ds[i * (BDH + 1) + d];
            }
        }
        if (j < p.n_mem) {
            float* o = p.dmem_part + (size_t)b * 2 * p.heads * p.n_mem * BDH;
#pragma unroll
            for (int d = 0; d < BDH; ++d) {
                o[((size_t)h * p.n_mem + j) * BDH + d] = scale * dk[d];
                o[((size_t)(p.heads + h) * p.n_mem + j) * BDH + d] = dv[d];
            }
        } else {
            const size_t row = ((size_t)b * p.nk + (j - p.n_mem)) * p.ldk + h * BDH;
#pragma unroll
            for (int d = 0; d < BDH; ++d) {
                p.dk[row + d] = scale * dk[d];
                p.dv[row + d] = dv[d];
            }
        }
    }
}

// ---- short sequences (the CACHE condition), every phase spread over the 256 threads: the 16 tokens + 4 memory rows of the
// 32x32 U-Net's bottleneck are 16 queries and 20 keys, so "thread = query" leaves 240 threads idle while 16 of them walk 20 keys
// x three 32-wide dot products each (22 us at any batch).  Here (1) thread = (query, key) pair forms the score and dP, (2) thread
// = query normalises its row (no dot products left), (3) thread = (query, d) forms dq, (4) thread = (key, d) forms dk and dv.
// Every sum runs over the same index in the same order as attn_bwd_kernel's (d, then keys, then queries): identical results.
__global__ __launch_bounds__(256) void attn_bwd_pairs_kernel(const AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int nkt = p.nk + p.n_mem, n = p.nq;
    float* Ks = sm;                       // [nkt][33]
    float* Vs = Ks + nkt * (BDH + 1);     // [nkt][33]
    float* Qs = Vs + nkt * (BDH + 1);     // [n][33]
    float* Ds = Qs + n * (BDH + 1);       // [n][33]  dout
    float* Pc = Ds + n * (BDH + 1) + 3 * n;  // (same offsets as attn_bwd_kernel<true>: one LDS size for both)
    float* Sc = Pc + n * nkt;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int hid = p.heads * BDH;
    const float scale = p.scale;
    for (int i = tid; i < nkt * BDH; i += 256) {
        const int j = i >> 5, d = i & 31;
        float kv, vv;
        if (j < p.n_mem) {
            kv = p.mem_k[((size_t)h * p.n_mem + j) * BDH + d];
            vv = p.mem_v[((size_t)h * p.n_mem + j) * BDH + d];
        } else {
            const size_t row = ((size_t)b * p.nk + (j - p.n_mem)) * p.ldk + h * BDH + d;
            kv = p.k[row];
            vv = p.v[row];
        }
        Ks[j * (BDH + 1) + d] = kv;
        Vs[j * (BDH + 1) + d] = vv;
    }
    for (int i = tid; i < n * BDH; i += 256) {
        const int t = i >> 5, d = i & 31;
        Qs[t * (BDH + 1) + d] = p.q[((size_t)b * n + t) * p.ldq + h * BDH + d];
        Ds[t * (BDH + 1) + d] = p.dout[((size_t)b * n + t) * hid + h * BDH + d];
    }
    __syncthreads();
    for (int idx = tid; idx < n * nkt; idx += 256) {  // (1) scores and dP
        const int i = idx / nkt, j = idx - i * nkt;
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < BDH; ++d) {
            sc += Qs[i * (BDH + 1) + d] * Ks[j * (BDH + 1) + d];
            dp += Ds[i * (BDH + 1) + d] * Vs[j * (BDH + 1) + d];
        }
        Pc[idx] = sc * scale;
        Sc[idx] = dp;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {  // (2) row softmax, D_i, dS
        float* Pi = Pc + i * nkt;
        float* Si = Sc + i * nkt;
        float m = -INFINITY, l = 0.f, D = 0.f;
        for (int j = 0; j < nkt; ++j) m = fmaxf(m, Pi[j]);
        for (int j = 0; j < nkt; ++j) {
            const float e = __expf(Pi[j] - m);
            Pi[j] = e;
            l += e;
            D += e * Si[j];
        }
        const float linv = 1.0f / l;
        D *= linv;
        for (int j = 0; j < nkt; ++j) {
            const float P = Pi[j] * linv;
            Si[j] = P * (Si[j] - D);
            Pi[j] = P;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < n * BDH; idx += 256) {  // (3) dq
        const int i = idx >> 5, d = idx & 31;
        float dq = 0.f;
        for (int j = 0; j < nkt; ++j) dq += Sc[i * nkt + j] * Ks[j * (BDH + 1) + d];
        p.dq[((size_t)b * n + i) * p.ldq + h * BDH + d] = scale * dq;
    }
    for (int idx = tid; idx < nkt * BDH; idx += 256) {  // (4) dk, dv
        const int j = idx >> 5, d = idx & 31;
        float dk = 0.f, dv = 0.f;
        for (int i = 0; i < n; ++i) {
            dk += Sc[i * nkt + j] * Qs[i * (BDH + 1) + d];
            dv += Pc[i * nkt + j] * Ds[i * (BDH + 1) + d];
        }
        if (j < p.n_mem) {
            float* o = p.dmem_part + (size_t)b * 2 * p.heads * p.n_mem * BDH;
            o[((size_t)h * p.n_mem + j) * BDH + d] = scale * dk;
            o[((size_t)(p.heads + h) * p.n_mem + j) * BDH + d] = dv;
        } else {
            const size_t row = ((size_t)b * p.nk + (j - p.n_mem)) * p.ldk + h * BDH;
            p.dk[row + d] = scale * dk;
            p.dv[row + d] = dv;
        }
    }
}

// ---- the same gradients for sequences that do not fit LDS: two tiled kernels, one wave per 64 queries / 64 keys.
// attn_bwd_q_tiled_kernel   grid (ceil(nq / 64), heads, B): lane = query; the keys stream through LDS in tiles of 64 three
//     times (row maximum; row sum and D_i; dq) -- the three loops of attn_bwd_kernel's first phase; writes dq and the row
//     statistics (m_i, 1 / l_i, D_i) to stats[b][h][i][3].
// attn_bwd_kv_tiled_kernel  grid (ceil((nk + n_mem) / 64), heads, B): lane = key; the queries (q, dO, statistics) stream
//     through LDS in tiles of 64 in order: dk, dv, memory-row gradients.  Fixed summation order, no atomics.
constexpr int ATS = BDH + 1;
__device__ __forceinline__ void attn_load_key(const AttnBwdParams& p, int b, int h, int j, int d, float* kv, float* vv) {
    if (j < p.n_mem) {
        *kv = p.mem_k[((size_t)h * p.n_mem + j) * BDH + d];
        *vv = p.mem_v[((size_t)h * p.n_mem + j) * BDH + d];
    } else {
        const size_t row = ((size_t)b * p.nk + (j - p.n_mem)) * p.ldk + h * BDH + d;
        *kv = p.k[row];
        *vv = p.v[row];
    }
}
__global__ __launch_bounds__(64) void attn_bwd_q_tiled_kernel(const AttnBwdParams p, float* __restrict__ stats) {
    __shared__ float Ks[64 * ATS], Vs[64 * ATS];
    const int h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane, n = p.nq, nkt = p.nk + p.n_mem;
    const bool ok = i < n;
    const int hid = p.heads * BDH;
    const float scale = p.scale;
    float q[BDH], dov[BDH];
#pragma unroll
    for (int d = 0; d < BDH; ++d) {
        q[d] = ok ? p.q[((size_t)b * n + i) * p.ldq + h * BDH + d] : 0.f;
        dov[d] = ok ? p.dout[((size_t)b * n + i) * hid + h * BDH + d] : 0.f;
    }
    float m = -INFINITY, l = 0.f, D = 0.f, dq[BDH];
#pragma unroll
    for (int d = 0; d < BDH; ++d) dq[d] = 0.f;
    for (int pass = 0; pass < 3; ++pass) {
        float linv = 0.f;
        if (pass == 2) {
            linv = 1.0f / l;
            D *= linv;
        }
        for (int j0 = 0; j0 < nkt; j0 += 64) {
            __syncthreads();
            for (int e = lane; e < 64 * BDH; e += 64) {
                const int jj = e >> 5, d = e & 31;
                float kv = 0.f, vv = 0.f;
                if (j0 + jj < nkt) attn_load_key(p, b, h, j0 + jj, d, &kv, &vv);
                Ks[jj * ATS + d] = kv;
                Vs[jj * ATS + d] = vv;
            }
            __syncthreads();
            const int jn = min(64, nkt - j0);
            for (int jj = 0; jj < jn; ++jj) {
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < BDH; ++d) {
                    sc += q[d] * Ks[jj * ATS + d];
                    dp += dov[d] * Vs[jj * ATS + d];
                }
                if (pass == 0) {
                    m = fmaxf(m, sc * scale);
                } else if (pass == 1) {
                    const float e = __expf(sc * scale - m);
                    l += e;
                    D += e * dp;
                } else {
                    const float dS = __expf(sc * scale - m) * linv * (dp - D);
#pragma unroll
                    for (int d = 0; d < BDH; ++d) dq[d] += dS * Ks[jj * ATS + d];
                }
            }
        }
    }
    if (ok) {
        float* st = stats + (((size_t)b * p.heads + h) * n + i) * 3;
        st[0] = m;
        st[1] = 1.0f / l;
        st[2] = D;
        float* o = p.dq + ((size_t)b * n + i) * p.ldq + h * BDH;
#pragma unroll
        for (int d = 0; d < BDH; ++d) o[d] = scale * dq[d];
    }
}
__global__ __launch_bounds__(64) void attn_bwd_kv_tiled_kernel(const AttnBwdParams p, const float* __restrict__ stats) {
    __shared__ float Qs[64 * ATS], Ds[64 * ATS], St[64 * 3];
    const int h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int j = blockIdx.x * 64 + lane, n = p.nq, nkt = p.nk + p.n_mem;
    const bool ok = j < nkt;
    const int hid = p.heads * BDH;
    const float scale = p.scale;
    float kk[BDH], vv[BDH], dk[BDH], dv[BDH];
#pragma unroll
    for (int d = 0; d < BDH; ++d) {
        kk[d] = vv[d] = 0.f;
        if (ok) attn_load_key(p, b, h, j, d, &kk[d], &vv[d]);
        dk[d] = dv[d] = 0.f;
    }
    for (int i0 = 0; i0 < n; i0 += 64) {
        __syncthreads();
        for (int e = lane; e < 64 * BDH; e += 64) {
            const int ii = e >> 5, d = e & 31;
            const bool in = i0 + ii < n;
            Qs[ii * ATS + d] = in ? p.q[((size_t)b * n + i0 + ii) * p.ldq + h * BDH + d] : 0.f;
            Ds[ii * ATS + d] = in ? p.dout[((size_t)b * n + i0 + ii) * hid + h * BDH + d] : 0.f;
        }
        for (int e = lane; e < 64 * 3; e += 64)
            St[e] = i0 + e / 3 < n ? stats[(((size_t)b * p.heads + h) * n + i0) * 3 + e] : 0.f;
        __syncthreads();
        const int in_ = min(64, n - i0);
        for (int ii = 0; ii < in_; ++ii) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < BDH; ++d) {
                sc += Qs[ii * ATS + d] * kk[d];
                dp += Ds[ii * ATS + d] * vv[d];
            }
            const float P = __expf(sc * scale - St[3 * ii]) * St[3 * ii + 1];
            const float dS = P * (dp - St[3 * ii + 2]);
#pragma unroll
            for (int d = 0; d < BDH; ++d) {
                dk[d] += dS * Qs[ii * ATS + d];
                dv[d] += P * Ds[ii * ATS + d];
            }
        }
    }
    if (!ok) return;
    if (j < p.n_mem) {
        float* o = p.dmem_part + (size_t)b * 2 * p.heads * p.n_mem * BDH;
#pragma unroll
        for (int d = 0; d < BDH; ++d) {
            o[((size_t)h * p.n_mem + j) * BDH + d] = scale * dk[d];
            o[((size_t)(p.heads + h) * p.n_mem + j) * BDH + d] = dv[d];
        }
    } else {
        const size_t row = ((size_t)b * p.nk + (j - p.n_mem)) * p.ldk + h * BDH;
#pragma unroll
        for (int d = 0; d < BDH; ++d) {
            p.dk[row + d] = scale * dk[d];
            p.dv[row + d] = dv[d];
        }
    }
}

static bool attn_bwd_cached(int nq, int nk, int n_mem) {
    static const bool off = std::getenv("DM_ATTN_BWD_NO_CACHE") != nullptr;  // A/B, and the tests' other path
    return !off && (size_t)nq * (nk + n_mem) <= 4096;
}
static size_t attn_bwd_lds_bytes(int nq, int nk, int n_mem) {
    const size_t cache = attn_bwd_cached(nq, nk, n_mem) ? (size_t)2 * nq * (nk + n_mem) : 0;
    return ((size_t)(2 * (nk + n_mem) + 2 * nq) * (BDH + 1) + 3 * nq + cache) * sizeof(float);
}
static bool attn_bwd_tiled(int nq, int nk, int n_mem) {
    static const bool force = std::getenv("DM_ATTN_BWD_TILED") != nullptr;  // tests: the tiled form on small shapes too
    return force || attn_bwd_lds_bytes(nq, nk, n_mem) > 160 * 1024;
}
// floats of statistics workspace the tiled form needs (0 when the LDS-resident kernel takes the shape)
size_t attn_bwd_ws_floats(int B, int nq, int nk, int n_mem, int heads) {
    return attn_bwd_tiled(nq, nk, n_mem) ? (size_t)B * heads * nq * 3 : 0;
}

static int launch_attn_bwd(const AttnBwdParams& p, int B, float* ws, hipStream_t s) {
    const size_t lds = attn_bwd_lds_bytes(p.nq, p.nk, p.n_mem);
    if (attn_bwd_tiled(p.nq, p.nk, p.n_mem)) {
        DM_REQUIRE(ws != nullptr, "attention backward: the tiled form needs its statistics workspace (attn_bwd_ws_floats)");
        DM_REQUIRE(B <= 65535 && p.heads <= 65535, "attention backward: batch");
        hipLaunchKernelGGL(attn_bwd_q_tiled_kernel, dim3((p.nq + 63) / 64, p.heads, B), dim3(64), 0, s, p, ws);
        DM_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(attn_bwd_kv_tiled_kernel, dim3((p.nk + p.n_mem + 63) / 64, p.heads, B), dim3(64), 0, s, p, ws);
        DM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    static const bool no_pairs = std::getenv("DM_ATTN_BWD_NO_PAIRS") != nullptr;
    if (attn_bwd_cached(p.nq, p.nk, p.n_mem) && !no_pairs) {
        static LdsOptIn flagp;
        if (lds_opt_in(flagp, reinterpret_cast<const void*>(attn_bwd_pairs_kernel), 1)) return 1;
        hipLaunchKernelGGL(attn_bwd_pairs_kernel, dim3(p.heads, B), dim3(256), lds, s, p);
        DM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (attn_bwd_cached(p.nq, p.nk, p.n_mem)) {
        static LdsOptIn flagc;
        if (lds_opt_in(flagc, reinterpret_cast<const void*>(attn_bwd_kernel<true>), 1)) return 1;
        hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(p.heads, B), dim3(256), lds, s, p);
        DM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    static LdsOptIn flag;
    if (lds_opt_in(flag, reinterpret_cast<const void*>(attn_bwd_kernel<false>), 1)) return 1;
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(p.heads, B), dim3(256), lds, s, p);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// qkv (B, n, 3*heads*32), dout (B, n, heads*32) -> dqkv, dmem_part (B, 2, heads, 4, 32)
int launch_attention_core_bwd(const float* qkv, const float* mem_kv, const float* dout, float* dqkv, float* dmem_part, float* ws,
                              int B, int n, int heads, int dh, hipStream_t s) {
    DM_REQUIRE(dh == BDH, "attention backward: dim_head 32");
    const int hid = heads * BDH;
    AttnBwdParams p{};
    p.q = qkv; p.k = qkv + hid; p.v = qkv + 2 * hid;
    p.mem_k = mem_kv; p.mem_v = mem_kv + (size_t)heads * NMEM * BDH;
    p.dout = dout;
    p.dq = dqkv; p.dk = dqkv + hid; p.dv = dqkv + 2 * hid;
    p.dmem_part = dmem_part;
    p.ldq = p.ldk = 3 * hid; p.nq = p.nk = n; p.n_mem = NMEM; p.heads = heads;
    p.scale = 1.0f / sqrtf((float)dh);
    return launch_attn_bwd(p, B, ws, s);
}

// CrossAttention core: q (B, nq, heads*32), k / v (B, m, heads*32) projections of the context -> dq, dk, dv (same shapes)
int launch_cross_attention_core_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                                    float* dv, float* ws, int B, int nq, int m, int heads, int dh, hipStream_t s) {
    DM_REQUIRE(dh == BDH, "attention backward: dim_head 32");
    AttnBwdParams p{};
    p.q = q; p.k = k; p.v = v;
    p.dout = dout;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.ldq = p.ldk = heads * BDH; p.nq = nq; p.nk = m; p.n_mem = 0; p.heads = heads;
    p.scale = 1.0f / sqrtf((float)dh);
    return launch_attn_bwd(p, B, ws, s);
}

}  // namespace dm

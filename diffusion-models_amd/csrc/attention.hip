// Attention cores of the denoising U-Net (gfx950).  dim_head is 32 in every reference config
// (DD/denoising_diffusion.py:248,155,200); the kernels below are specialised for it.
//
//   LinearAttention core  DD/denoising_diffusion.py:179-192   (two softmaxes + two 32x32 products/head)
//   Attention core        DD/denoising_diffusion.py:221-226 + DD/attend.py:109-124
//   CrossAttention core   DD/denoising_diffusion_text_conditional.py:68-77 (same kernel, no memory kv)
//
// qkv tensors are NHWC, i.e. one row of 3*heads*32 floats per token: [q(h,d) | k(h,d) | v(h,d)].
#include "dm_common.h"

#include <algorithm>

#include <cstdlib>

namespace dm {

constexpr int DH = 32;

__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ---------------------------------------------------------------------------------------
// LinearAttention, part 1 on the f32 MFMA: ctx[d][e] = sum_t softmax_t(k)[d][t] * v[e][t] is a (32 x n) x (n x 32)
// product per (image, head), with the tokens as the reduction axis:
//   A[i = d][k = token] = exp(k[token][d] - max_d),  B[k = token][j = e] = v[token][e]
// so a lane loads ONE dword of k and ONE of v per step (lanes 0-31 / 32-63 read the 128 contiguous bytes of
// two consecutive tokens) and the 32x32 context accumulates in 16 registers.  grid (heads, B), 4 waves; each
// wave takes a quarter of the tokens, the partial contexts meet in LDS.  The 4 memory tokens are two extra steps.
// ---------------------------------------------------------------------------------------
using f32x16_t = __attribute__((ext_vector_type(16))) float;

// NW waves per (image, head): the token loop is a chain of load round trips (8 loads in flight per lane, then 4 MFMAs), so with
// heads x B workgroups and nothing else to overlap, the kernel's time is one wave's chain; 16 waves on the 1024 tokens of a 32x32
// stage make that chain 8 rounds instead of 32 (45 -> about 15 us in the B = 64 training step).
template <int NW>
__global__ __launch_bounds__(64 * NW) void linattn_ctx_mfma_kernel(const float* __restrict__ qkv,
                                                                   const float* __restrict__ mem_kv,
                                                                   float* __restrict__ ctx, float* __restrict__ kstats,
                                                                   int n, int heads) {
    constexpr int NMEM = 4;
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int ld = 3 * heads * DH;
    const float* kb = qkv + (size_t)b * n * ld + heads * DH + h * DH + c;
    const float* vb = qkv + (size_t)b * n * ld + 2 * heads * DH + h * DH + c;
    const float* mk = mem_kv + (size_t)h * DH * NMEM;            // [d][j]
    const float* mv = mem_kv + (size_t)(heads + h) * DH * NMEM;  // [e][j]
    __shared__ float red[NW][DH];
    __shared__ float kmax_s[DH], ksum_s[DH];
    constexpr int NP = NW > 8 ? NW / 2 : NW;  // 16 waves fold in two rounds: 64 KB of static LDS would not fit with the rest
    __shared__ __attribute__((aligned(16))) float part[NP][DH * DH];

    // tokens of this wave: [t0, t1), visited two at a time (one per lane half)
    const int per = ((n + 2 * NW - 1) / (2 * NW)) * 2;  // even number of tokens per wave
    const int t0 = min(n, wave * per), t1 = min(n, t0 + per);

    // pass 1: max over all tokens (incl. memory) of k[.][d]
    float m = -INFINITY;
    {
        int t = t0;
        for (; t + 16 <= t1; t += 16) {
            float kv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) kv[s] = kb[(size_t)(t + 2 * s + half) * ld];
#pragma unroll
            for (int s = 0; s < 8; ++s) m = fmaxf(m, kv[s]);
        }
        for (t += half; t < t1; t += 2) m = fmaxf(m, kb[(size_t)t * ld]);
    }
    if (wave == 0) {
        m = fmaxf(m, mk[c * NMEM + half]);
        m = fmaxf(m, mk[c * NMEM + 2 + half]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    if (half == 0) red[wave][c] = m;
    __syncthreads();
    if (tid < DH) {
        float mm = red[0][tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) mm = fmaxf(mm, red[w][tid]);
        kmax_s[tid] = mm;
    }
    __syncthreads();
    const float kmax = kmax_s[c];

    // pass 2: exp, row sums, and the outer-product accumulation on the matrix core
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float ksum = 0.f;
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < NMEM; j += 2) {
            float a = __expf(mk[c * NMEM + j + half] - kmax);
            float bv = mv[c * NMEM + j + half];
            ksum += a;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
        }
    }
    int t = t0;
    for (; t + 8 <= t1; t += 8) {  // 4 steps per iteration: 8 loads in flight per lane
        float kv[4], vv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kv[s] = kb[(size_t)(t + 2 * s + half) * ld];
            vv[s] = vb[(size_t)(t + 2 * s + half) * ld];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float a = __expf(kv[s] - kmax);
            ksum += a;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vv[s], acc, 0, 0, 0);
        }
    }
    for (; t < t1; t += 2) {  // tail: a missing token contributes a = 0
        const bool ok = t + half < t1;
        float a = ok ? __expf(kb[(size_t)(t + half) * ld] - kmax) : 0.f;
        float bv = ok ? vb[(size_t)(t + half) * ld] : 0.f;
        ksum += a;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
    ksum += __shfl_xor(ksum, 32);
    __syncthreads();  // red is reused
    if (half == 0) red[wave][c] = ksum;
    // accumulator element e of lane: row d = (e&3) + 8*(e>>2) + 4*half, column e-index = lane&31
    if constexpr (NP < NW) {
        if (wave >= NP) {
#pragma unroll
            for (int e = 0; e < 16; ++e) part[wave - NP][((e & 3) + 8 * (e >> 2) + 4 * half) * DH + c] = acc[e];
        }
        __syncthreads();
        if (wave < NP) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += part[wave][((e & 3) + 8 * (e >> 2) + 4 * half) * DH + c];
        }
    }
    if (wave < NP) {
#pragma unroll
        for (int e = 0; e < 16; ++e) part[wave][((e & 3) + 8 * (e >> 2) + 4 * half) * DH + c] = acc[e];
    }
    __syncthreads();
    if (tid < DH) {
        float ss = red[0][tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) ss += red[w][tid];
        ksum_s[tid] = ss;
        if (kstats) {  // the training tape keeps the column statistics: the backward pass starts from them
            kstats[(size_t)(b * heads + h) * 2 * DH + tid] = kmax_s[tid];
            kstats[(size_t)(b * heads + h) * 2 * DH + DH + tid] = ss;
        }
    }
    __syncthreads();
    float* cp = ctx + (size_t)(b * heads + h) * DH * DH;
    for (int i = tid; i < DH * DH; i += 64 * NW) {
        const int d = i >> 5;
        float sum = part[0][i];
#pragma unroll
        for (int w = 1; w < NP; ++w) sum += part[w][i];
        cp[i] = sum / ksum_s[d];
    }
}

// ---------------------------------------------------------------------------------------
// LinearAttention, part 1 (VALU version, kept as the fallback for n_mem != 4):
// ctx[b][h][d][e] = sum_n softmax_n(k)[d][n] * v[e][n]  (n includes the memory tokens)
// grid (heads, B), 256 threads.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const float* __restrict__ qkv,
                                                          const float* __restrict__ mem_kv,
                                                          float* __restrict__ ctx, int n, int heads, int n_mem) {
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const int ld = 3 * heads * DH;
    const float* kbase = qkv + (size_t)b * n * ld + heads * DH + h * DH;
    const float* vbase = qkv + (size_t)b * n * ld + 2 * heads * DH + h * DH;
    const float* mk = mem_kv + (size_t)h * DH * n_mem;            // [d][j]
    const float* mv = mem_kv + (size_t)(heads + h) * DH * n_mem;  // [e][j]
    const int ntok = n + n_mem;

    __shared__ float red[8][DH];
    __shared__ float kmax[DH];
    __shared__ float ke[64][DH + 1];
    __shared__ __attribute__((aligned(16))) float vv[64][DH];
    __shared__ float ksum_s[DH];

    // pass 1: max over tokens for each d
    {
        const int d = tid & 31, part = tid >> 5;
        float m = -INFINITY;
        for (int t = part; t < ntok; t += 8) {
            float kv = t < n_mem ? mk[d * n_mem + t] : kbase[(size_t)(t - n_mem) * ld + d];
            m = fmaxf(m, kv);
        }
        red[part][d] = m;
        __syncthreads();
        if (tid < DH) {
            float mm = red[0][tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) mm = fmaxf(mm, red[q][tid]);
            kmax[tid] = mm;
        }
        __syncthreads();
    }
    // pass 2: exp, running sum and the 32x32 outer-product accumulation
    const int d = tid >> 3, e0 = (tid & 7) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float ksum = 0.f;
    for (int t0 = 0; t0 < ntok; t0 += 64) {
        __syncthreads();
        for (int it = tid; it < 64 * DH; it += 256) {
            int tt = it >> 5, c = it & 31;
            int t = t0 + tt;
            float kval = 0.f, vval = 0.f;
            if (t < ntok) {
                if (t < n_mem) {
                    kval = __expf(mk[c * n_mem + t] - kmax[c]);
                    vval = mv[c * n_mem + t];
                } else {
                    kval = __expf(kbase[(size_t)(t - n_mem) * ld + c] - kmax[c]);
                    vval = vbase[(size_t)(t - n_mem) * ld + c];
                }
            }
            ke[tt][c] = kval;
            vv[tt][c] = vval;
        }
        __syncthreads();
#pragma unroll 8
        for (int tt = 0; tt < 64; ++tt) {
            float kx = ke[tt][d];
            float4 v4 = *reinterpret_cast<const float4*>(&vv[tt][e0]);
            ksum += kx;
            acc[0] += kx * v4.x;
            acc[1] += kx * v4.y;
            acc[2] += kx * v4.z;
            acc[3] += kx * v4.w;
        }
    }
    if ((tid & 7) == 0) ksum_s[d] = ksum;
    __syncthreads();
    const float inv = 1.0f / ksum_s[d];
    float* cp = ctx + ((size_t)(b * heads + h) * DH + d) * DH + e0;
    *reinterpret_cast<float4*>(cp) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
}

// LinearAttention, part 2: out[b][n][h*32+e] = sum_d ctx[b][h][d][e] * softmax_d(q[b][n][h][:])[d] * 32^-0.5
// grid (ceil(n/64), B), block = 64*heads threads: thread = h*64 + pixel (one head per wavefront).
__global__ void linattn_out_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                   float* __restrict__ out, int n, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float cs[];  // [heads][DH][DH]
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    for (int i = tid; i < heads * DH * DH; i += blockDim.x) cs[i] = ctx[(size_t)b * heads * DH * DH + i];
    __syncthreads();
    const int h = tid >> 6;
    const int p = blockIdx.x * 64 + (tid & 63);
    if (p >= n) return;
    const int ld = 3 * heads * DH;
    const float* qp = qkv + ((size_t)b * n + p) * ld + h * DH;
    float q[DH];
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
        float4 t = *reinterpret_cast<const float4*>(qp + 4 * i);
        q[4 * i] = t.x; q[4 * i + 1] = t.y; q[4 * i + 2] = t.z; q[4 * i + 3] = t.w;
    }
    float m = q[0];
#pragma unroll
    for (int i = 1; i < DH; ++i) m = fmaxf(m, q[i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < DH; ++i) {
        q[i] = __expf(q[i] - m);
        s += q[i];
    }
    const float inv = scale / s;
    float o[DH];
#pragma unroll
    for (int i = 0; i < DH; ++i) o[i] = 0.f;
    const float* ch = cs + h * DH * DH;
#pragma unroll
    for (int dd = 0; dd < DH; ++dd) {
        const float qd = q[dd] * inv;
#pragma unroll
        for (int i = 0; i < DH / 4; ++i) {
            float4 c4 = *reinterpret_cast<const float4*>(ch + dd * DH + 4 * i);
            o[4 * i] += qd * c4.x;
            o[4 * i + 1] += qd * c4.y;
            o[4 * i + 2] += qd * c4.z;
            o[4 * i + 3] += qd * c4.w;
        }
    }
    float* op = out + ((size_t)b * n + p) * (heads * DH) + h * DH;
#pragma unroll
    for (int i = 0; i < DH / 4; ++i)
        *reinterpret_cast<float4*>(op + 4 * i) = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
}

// part 2 on the matrix core: out^T[e][token] = sum_d ctx[d][e] q_s[token][d] as 32x32x2 MFMA products with the token rows as the
// B operand (attn_bwd.hip's linattn_bwd_*_mfma_kernel describe the layout): lane (token, half) holds the four float4 chunks
// 2 m + half of its token's q row = the columns dset(r) = (r & 3) + 8 (r >> 2) + 4 half, the softmax over the 32 columns is 16
// registers plus one exchange with lane ^ 32, the K index runs over d in the same order, and the result D[i = e][j = token]
// leaves the lane with out[token][dset(r)]: four float4 stores.  grid (ceil(n / 64), heads, B), one wave = 64 tokens.
__global__ __launch_bounds__(64) void linattn_out_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                              float* __restrict__ out, int n, int heads, float scale) {
    const int blk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int ld = 3 * heads * DH, hid = heads * DH;
    float a[16];  // A[i = e = l31][k -> d = dset(s)] = ctx[d][e]
    {
        const float* cp = ctx + (size_t)(b * heads + h) * DH * DH + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = cp[((s & 3) + 8 * (s >> 2) + 4 * half) * DH];
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int tok = blk * 64 + nt * 32 + l31;
        const bool ok = tok < n;
        const size_t row = (size_t)b * n + (ok ? tok : 0);
        const float4* qp = reinterpret_cast<const float4*>(qkv + row * ld + h * DH);
        float q[16];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 t = qp[2 * m + half];
            q[4 * m] = t.x; q[4 * m + 1] = t.y; q[4 * m + 2] = t.z; q[4 * m + 3] = t.w;
        }
        float mx = q[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, q[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            q[r] = __expf(q[r] - mx);
            sum += q[r];
        }
        sum += __shfl_xor(sum, 32);
        const float inv = scale / sum;
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], q[s] * inv, acc, 0, 0, 0);
        if (ok) {
            float4* op = reinterpret_cast<float4*>(out + row * hid + h * DH);
#pragma unroll
            for (int m = 0; m < 4; ++m) op[2 * m + half] = make_float4(acc[4 * m], acc[4 * m + 1], acc[4 * m + 2], acc[4 * m + 3]);
        }
    }
}

// the training tape keeps the key statistics unless the VALU context kernel is forced or DM_LINATTN_NO_KSTATS asks for the
// recomputing backward (tests/test_hip_forced_dispatch.py)
bool linattn_keeps_kstats() {
    static const bool keep = !std::getenv("DM_LINATTN_VALU") && !std::getenv("DM_LINATTN_NO_KSTATS");
    return keep;
}

int launch_linear_attention_core(const float* qkv, const float* mem_kv, float* ctx_ws, float* out, int B, int n,
                                 int heads, int dh, hipStream_t s, float* kstats) {
    DM_REQUIRE(dh == DH, "LinearAttention kernel is specialised for dim_head == 32");
    DM_REQUIRE(heads >= 1 && heads <= 16, "LinearAttention kernel supports 1..16 heads");
    static const bool valu_ctx = std::getenv("DM_LINATTN_VALU") != nullptr;
    if (valu_ctx) {
        DM_REQUIRE(!kstats, "DM_LINATTN_VALU: the VALU context kernel does not keep the key statistics");
        hipLaunchKernelGGL(linattn_ctx_kernel, dim3(heads, B), dim3(256), 0, s, qkv, mem_kv, ctx_ws, n, heads, 4);
    } else {
        // waves per (image, head) by the sequence length alone, so that a sample's result does not depend on its batch
        static const int force = std::getenv("DM_LINATTN_CTX_WAVES") ? atoi(std::getenv("DM_LINATTN_CTX_WAVES")) : 0;
        const int nw = force ? force : n >= 1024 ? 16 : n >= 256 ? 8 : 4;
        if (nw == 16)
            hipLaunchKernelGGL(linattn_ctx_mfma_kernel<16>, dim3(heads, B), dim3(1024), 0, s, qkv, mem_kv, ctx_ws, kstats,
                               n, heads);
        else if (nw == 8)
            hipLaunchKernelGGL(linattn_ctx_mfma_kernel<8>, dim3(heads, B), dim3(512), 0, s, qkv, mem_kv, ctx_ws, kstats, n,
                               heads);
        else
            hipLaunchKernelGGL(linattn_ctx_mfma_kernel<4>, dim3(heads, B), dim3(256), 0, s, qkv, mem_kv, ctx_ws, kstats, n,
                               heads);
    }
    DM_CHECK_HIP(hipGetLastError());
    if (valu_ctx) {
        size_t lds = (size_t)heads * DH * DH * sizeof(float);
        hipLaunchKernelGGL(linattn_out_kernel, dim3((n + 63) / 64, B), dim3(64 * heads), lds, s, qkv, ctx_ws, out, n,
                           heads, 1.0f / sqrtf((float)dh));
    } else {
        DM_REQUIRE(B <= 65535, "LinearAttention: batch");
        hipLaunchKernelGGL(linattn_out_mfma_kernel, dim3((n + 63) / 64, heads, B), dim3(64), 0, s, qkv, ctx_ws, out, n, heads,
                           1.0f / sqrtf((float)dh));
    }
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// softmax(q k^T * scale) v for short sequences; K and V of one (batch, head) live in LDS.
// grid (heads, B), 256 threads (4 waves, one query row per wave at a time).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_core_kernel(const float* __restrict__ q, int ldq,
                                                             const float* __restrict__ k,
                                                             const float* __restrict__ v, int ldk,
                                                             const float* __restrict__ mem_k,
                                                             const float* __restrict__ mem_v, int n_mem,
                                                             float* __restrict__ out, int ldo, int nq, int nk,
                                                             float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntok = nk + n_mem;
    float* Ks = sm;                      // [ntok][DH+1]
    float* Vs = Ks + ntok * (DH + 1);    // [ntok][DH]
    float* Ps = Vs + ntok * DH;          // [4][ntok]
    float* Qs = Ps + 4 * ntok;           // [4][DH]
    for (int it = tid; it < ntok * DH; it += 256) {
        int t = it >> 5, c = it & 31;
        float kv, vvv;
        if (t < n_mem) {
            kv = mem_k[((size_t)h * n_mem + t) * DH + c];
            vvv = mem_v[((size_t)h * n_mem + t) * DH + c];
        } else {
            size_t o = ((size_t)b * nk + (t - n_mem)) * ldk + h * DH + c;
            kv = k[o];
            vvv = v[o];
        }
        Ks[t * (DH + 1) + c] = kv;
        Vs[t * DH + c] = vvv;
    }
    __syncthreads();
    float* pw = Ps + wave * ntok;
    float* qw = Qs + wave * DH;
    // blockIdx.z splits the queries when (heads x batch) alone leaves CUs idle (small batches, 64-token stages)
    for (int i = blockIdx.z * 4 + wave; i < nq; i += 4 * gridDim.z) {
        if (lane < DH) qw[lane] = q[((size_t)b * nq + i) * ldq + h * DH + lane] * scale;
        __builtin_amdgcn_wave_barrier();
        // scores for keys j = lane, lane+64, ...
        float mx = -INFINITY;
        for (int j = lane; j < ntok; j += 64) {
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) sc += qw[c] * Ks[j * (DH + 1) + c];
            pw[j] = sc;
            mx = fmaxf(mx, sc);
        }
        mx = wave_max64(mx);
        float sum = 0.f;
        for (int j = lane; j < ntok; j += 64) {
            float pe = __expf(pw[j] - mx);
            pw[j] = pe;
            sum += pe;
        }
        sum = wave_sum64(sum);
        __builtin_amdgcn_wave_barrier();
        // out[d] = sum_j p_j v[j][d]; lane = (half, d): halves split the keys
        const int d = lane & 31, half = lane >> 5;
        float acc = 0.f;
        for (int j = half; j < ntok; j += 2) acc += pw[j] * Vs[j * DH + d];
        acc += __shfl_xor(acc, 32);
        if (lane < DH) out[((size_t)b * nq + i) * ldo + h * DH + d] = acc / sum;
        __builtin_amdgcn_wave_barrier();
    }
}

// The same for sequences whose K and V do not fit LDS: one wave per 64 queries (lane = query, its q row and the 32
// accumulators in registers), the keys stream through LDS in tiles of 64 twice (row maximum; then exp, row sum and P V) --
// two passes instead of an online rescale keep the arithmetic that of softmax() followed by the product.
// grid (ceil(nq / 64), heads, B), 64 threads.
__global__ __launch_bounds__(64) void attention_core_tiled_kernel(const float* __restrict__ q, int ldq,
                                                                  const float* __restrict__ k,
                                                                  const float* __restrict__ v, int ldk,
                                                                  const float* __restrict__ mem_k,
                                                                  const float* __restrict__ mem_v, int n_mem,
                                                                  float* __restrict__ out, int ldo, int nq, int nk,
                                                                  float scale) {
    __shared__ float Ks[64 * (DH + 1)], Vs[64 * (DH + 1)];
    const int h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane, ntok = nk + n_mem;
    const bool ok = i < nq;
    float qr[DH], acc[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) {
        qr[c] = ok ? q[((size_t)b * nq + i) * ldq + h * DH + c] * scale : 0.f;
        acc[c] = 0.f;
    }
    float mx = -INFINITY, sum = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        for (int j0 = 0; j0 < ntok; j0 += 64) {
            __syncthreads();
            for (int e = lane; e < 64 * DH; e += 64) {
                const int t = j0 + (e >> 5), c = e & 31;
                float kv = 0.f, vvv = 0.f;
                if (t < ntok) {
                    if (t < n_mem) {
                        kv = mem_k[((size_t)h * n_mem + t) * DH + c];
                        vvv = mem_v[((size_t)h * n_mem + t) * DH + c];
                    } else {
                        const size_t o = ((size_t)b * nk + (t - n_mem)) * ldk + h * DH + c;
                        kv = k[o];
                        vvv = v[o];
                    }
                }
                Ks[(e >> 5) * (DH + 1) + c] = kv;
                Vs[(e >> 5) * (DH + 1) + c] = vvv;
            }
            __syncthreads();
            const int jn = min(64, ntok - j0);
            for (int jj = 0; jj < jn; ++jj) {
                float sc = 0.f;
#pragma unroll
                for (int c = 0; c < DH; ++c) sc += qr[c] * Ks[jj * (DH + 1) + c];
                if (pass == 0) {
                    mx = fmaxf(mx, sc);
                } else {
                    const float pe = __expf(sc - mx);
                    sum += pe;
#pragma unroll
                    for (int c = 0; c < DH; ++c) acc[c] += pe * Vs[jj * (DH + 1) + c];
                }
            }
        }
    }
    if (!ok) return;
    const float inv = 1.0f / sum;
    float* o = out + ((size_t)b * nq + i) * ldo + h * DH;
#pragma unroll
    for (int c = 0; c < DH; c += 4) *reinterpret_cast<float4*>(o + c) = make_float4(acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv);
}

int launch_attention_core(const float* q, int ldq, const float* k, const float* v, int ldk, const float* mem_k,
                          const float* mem_v, int n_mem, float* out, int ldo, int B, int nq, int nk, int heads,
                          int dh, float scale, hipStream_t s) {
    DM_REQUIRE(dh == DH, "attention kernel is specialised for dim_head == 32");
    int ntok = nk + n_mem;
    size_t lds = ((size_t)ntok * (DH + 1) + (size_t)ntok * DH + 4 * (size_t)ntok + 4 * DH) * sizeof(float);
    static const bool force_tiled = std::getenv("DM_ATTN_TILED") != nullptr;  // tests: the tiled form on short sequences
    // beyond ~320 keys the tiled form is also the faster one (tools/attn_time.py: 512 tokens 1.29 vs 2.79 ms per layer at
    // B=64, equal at 256): the LDS-resident kernel re-stages all K / V of an (image, head) in every query block
    static const int tiled_min = env_int("DM_ATTN_TILED_MIN", 320);
    if (lds > 160 * 1024 || ntok > tiled_min || force_tiled) {
        DM_REQUIRE(B <= 65535 && heads <= 65535 && ldo % 4 == 0, "attention: batch / row stride");
        hipLaunchKernelGGL(attention_core_tiled_kernel, dim3((nq + 63) / 64, heads, B), dim3(64), 0, s, q, ldq, k, v, ldk,
                           mem_k, mem_v, n_mem, out, ldo, nq, nk, scale);
        DM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(attention_core_kernel), 1)) return 1;
    const int qblocks = std::max(1, std::min((nq + 3) / 4, (512 + heads * B - 1) / (heads * B)));
    hipLaunchKernelGGL(attention_core_kernel, dim3(heads, B, qblocks), dim3(256), lds, s, q, ldq, k, v, ldk, mem_k,
                       mem_v, n_mem, out, ldo, nq, nk, scale);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dm

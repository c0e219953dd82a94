// Fused LinearAttention for gfx950 (DD/denoising_diffusion.py:150-192, called as `attn(x) + x` at :360/:383).
//
// The unfused path (RMSNorm kernel -> to_qkv 1x1 conv -> context kernel -> out kernel -> to_out conv) moves the
// (B, n, 384) qkv tensor through HBM twice and the (B, n, 128) attention output once more: ~1.5 GB per layer at
// 32x32 / batch 256 for ~27 GFLOP of work.  Here the whole layer is two kernels that read x twice and write y once:
//
//   1. linattn_ctx_fused_kernel   (token block, image): k, v = W_k x^, W_v x^ on the MFMA, exp, and the per-head
//      32x32 context  ctx[d][e] += sum_t exp(k[d][t] - bound[d]) v[e][t]  plus the softmax denominators, written
//      as per-block partial sums.  x^ = RMSNorm(x) has unit length before the gain, and the gain g*sqrt(C) is folded
//      into the weights on the host, so |k[d][t]| <= ||W'_k[d]||_2 =: bound[d] (Cauchy-Schwarz).  softmax is
//      invariant to the shift, so the bound replaces the running maximum: no max pass over the tokens, and partial
//      sums of different token blocks simply add.  (Layers whose bound exceeds 40 -- exp(-2*bound) must stay a
//      normal float -- keep the unfused path.)
//   (a tiny kernel sums the block partials, divides by the denominators and folds the context into to_out:
//    z = W_out (ctx^T q) = (W_out ctx^T) q =: M q with M (C x 32) per image and head, packed in lane order)
//   2. linattn_out_fused_kernel   (token block, image): q = W_q x^, softmax over dim_head, z = M q + b,
//      RMSNorm(z) * g, + x, one store.
//
// MFMA bookkeeping (v_mfma_f32_32x32x2_f32, wave = head): the accumulator of one product is used DIRECTLY as an
// operand of the next one, without any data movement: accumulator register e of lane (l, half) is element
// (row (e&3) + 8(e>>2) + 4*half, column l).  Taking MFMA step e of the next product to reduce over exactly that
// row pair (one row from each lane half) makes acc[e] the B operand (or, for the transposed products of kernel 1,
// the A operand) of step e.  The remaining operands (projection weights, context, W_out) are packed on the host in
// lane order and sit in registers.
#include "conv_device.h"

#include <cmath>
#include <vector>

namespace dm {

static constexpr int LA_DH = 32;
static constexpr int LA_HEADS = 4;
static constexpr int LA_HID = LA_DH * LA_HEADS;
static constexpr int LA_CTX = LA_DH * LA_DH + LA_DH;  // partial context + partial denominators per (image, block, head)
static constexpr int LA_TOK1 = 128;                   // tokens per workgroup, kernel 1 (C = 128; 256 for C = 64)
__host__ __device__ constexpr int la_tok1(int C) { return C <= 64 ? 256 : LA_TOK1; }
static constexpr int LA_TOK2 = 64;                    // tokens per workgroup, kernel 2

__host__ __device__ static inline int la_row_of(int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; }

bool linattn_fused_eligible(int C, int heads, int dh) {
    static const bool off = std::getenv("DM_NO_FUSED_LINATTN") != nullptr;
    return !off && heads == LA_HEADS && dh == LA_DH && (C == 64 || C == 128);
}

// Host-side packing.  w_qkv is (3*128, C) with rows [q | k | v] x (head, d); norm_g (C); w_out (C, 128);
// mem_kv (2, heads, 32, 4).  Returns false when a softmax bound is too large for the shift trick.
bool linattn_fused_pack(const float* w_qkv, const float* norm_g, const float* w_out, const float* mem_kv, int C,
                        std::vector<float>& wq, std::vector<float>& wk, std::vector<float>& wv, std::vector<float>& wo,
                        std::vector<float>& kbound) {
    const int G = C / 8;
    const float sq = std::sqrt((float)C);
    auto pack_proj = [&](int which, std::vector<float>& dst) {
        // [head][g][lane = half*32 + l][4]: W'[head*32 + l][8g + 4*half + s],  W' = W * g * sqrt(C) per input channel
        dst.assign((size_t)LA_HEADS * G * 64 * 4, 0.f);
        for (int h = 0; h < LA_HEADS; ++h)
            for (int g = 0; g < G; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) {
                        const int l = lane & 31, half = lane >> 5;
                        const int row = which * LA_HID + h * LA_DH + l, c = 8 * g + 4 * half + s;
                        dst[(((size_t)h * G + g) * 64 + lane) * 4 + s] = w_qkv[(size_t)row * C + c] * (norm_g[c] * sq);
                    }
    };
    pack_proj(0, wq);
    pack_proj(1, wk);
    pack_proj(2, wv);
    wo.clear();  // to_out is folded with the per-image context on the device (linattn_ctx_reduce_kernel)
    (void)w_out;
    kbound.assign(LA_HID, 0.f);
    bool ok = true;
    for (int h = 0; h < LA_HEADS; ++h)
        for (int d = 0; d < LA_DH; ++d) {
            double ss = 0;
            for (int c = 0; c < C; ++c) {
                const double v = (double)w_qkv[(size_t)(LA_HID + h * LA_DH + d) * C + c] * norm_g[c] * sq;
                ss += v * v;
            }
            float bnd = (float)(std::sqrt(ss) * 1.0001);
            for (int j = 0; j < 4; ++j) bnd = std::fmax(bnd, mem_kv[((size_t)h * LA_DH + d) * 4 + j]);
            kbound[h * LA_DH + d] = bnd;
            if (!(bnd < 40.f)) ok = false;
        }
    return ok;
}

size_t linattn_fused_ws_floats(int B, int n) {
    const int nblk = (n + LA_TOK1 - 1) / LA_TOK1;  // upper bound over C
    // per-block partial contexts, then M = W_out ctx^T per (image, head) packed as the A operand of kernel 2 (C <= 128)
    return (size_t)B * nblk * LA_HEADS * LA_CTX + (size_t)B * LA_HEADS * 128 * LA_DH;
}

// x rows [t0, t0 + TOK) of image b -> LDS (zero rows past the image), rn[t] = 1 / max(||x_t||, 1e-12)
template <int C, int TOK>
__device__ __forceinline__ void la_stage_rows(const float* __restrict__ xb, int nt, float* xs, float* rn) {
    constexpr int XS = C + 4, Q = C / 4;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TOK * Q / 256; ++i) {
        const int it = tid + 256 * i;
        const int row = it / Q, q = it - row * Q;
        f32x4 v = make_f32x4(0.f, 0.f, 0.f, 0.f);
        if (row < nt) v = *reinterpret_cast<const f32x4*>(xb + (size_t)row * C + 4 * q);
        *reinterpret_cast<f32x4*>(xs + row * XS + 4 * q) = v;
    }
    __syncthreads();
    if (tid < TOK) {
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xs + tid * XS + 4 * q);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        rn[tid] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    }
}

template <int C, int TOK>
__global__ __launch_bounds__(256) void linattn_ctx_fused_kernel(const float* __restrict__ x, const LinAttnFused w,
                                                                float* __restrict__ ws, int n, int nblk) {
    constexpr int G = C / 8, XS = C + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* rn = smem + TOK * XS;
    const int blk = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int t0 = blk * TOK;
    const int nt = min(TOK, n - t0);

    f32x4 wk[G], wv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        wk[g] = *reinterpret_cast<const f32x4*>(w.wk + (((size_t)h * G + g) * 64 + lane) * 4);
        wv[g] = *reinterpret_cast<const f32x4*>(w.wv + (((size_t)h * G + g) * 64 + lane) * 4);
    }
    const float kb = w.kbound[h * LA_DH + l31];
    la_stage_rows<C, TOK>(x + ((size_t)b * n + t0) * C, nt, xs, rn);
    __syncthreads();

    f32x16 cacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) cacc[e] = 0.f;
    float ksum = 0.f;
    if (blk == 0) {
        // the 4 learned memory tokens (mem_kv, :159/:182-183): two MFMA steps, token j + half per lane half
        const float* mk = w.mem_kv + (size_t)h * LA_DH * 4;               // [d][j]
        const float* mv = w.mem_kv + (size_t)(LA_HEADS + h) * LA_DH * 4;  // [e][j]
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const float a = __expf(mk[l31 * 4 + j + lh] - kb);
            ksum += a;
            cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, mv[l31 * 4 + j + lh], cacc, 0, 0, 0);
        }
    }
    for (int st = 0; st < TOK / 32; ++st) {
        if (st * 32 >= nt) break;
        // k^T, v^T (tokens x d) = x (tokens x C) W'^T: A = x row of token l31 from LDS, B = weights in registers
        f32x16 kacc, vacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            kacc[e] = 0.f;
            vacc[e] = 0.f;
        }
        const float* xr = xs + (st * 32 + l31) * XS + 4 * lh;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 8 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                kacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wk[g][s], kacc, 0, 0, 0);
                vacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wv[g][s], vacc, 0, 0, 0);
            }
        }
        // register e holds token st*32 + row_of(e, lh), channel d = l31 (k) / e = l31 (v): exactly the operands of
        // step e of  ctx[d][e] += sum_t a[d][t] v[t][e]
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(rn + st * 32 + 8 * m + 4 * lh);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * m + i;
                const bool valid = st * 32 + 8 * m + 4 * lh + i < nt;
                const float a = valid ? __expf(kacc[e] * r4[i] - kb) : 0.f;
                ksum += a;
                cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vacc[e] * r4[i], cacc, 0, 0, 0);
            }
        }
    }
    float* cp = ws + (((size_t)b * nblk + blk) * LA_HEADS + h) * LA_CTX;
#pragma unroll
    for (int e = 0; e < 16; ++e) cp[la_row_of(e, lh) * LA_DH + l31] = cacc[e];
    ksum += __shfl_xor(ksum, 32);
    if (lh == 0) cp[LA_DH * LA_DH + l31] = ksum;
}

// Sum the per-block partial contexts in block order, divide by the softmax denominator (k.softmax(dim=-1), :186), and
// fold the result into to_out: out = ctx^T q (:189) followed by z = W_out[:, head] out (:191) is z = M q with
// M[c][d] = sum_e W_out[c][head*32 + e] ctx[d][e].  M is stored in the lane order kernel 2 consumes as an MFMA A
// operand: [image][head][mt][e>>2][lane][e&3] = M[32*mt + l][row_of(e, half)].
template <int C>
__global__ __launch_bounds__(256) void linattn_ctx_reduce_kernel(const float* __restrict__ ws,
                                                                 const float* __restrict__ w_out,
                                                                 float* __restrict__ mz, int nblk) {
    constexpr int MT = C / 32;
    __shared__ float cs[LA_DH][LA_DH + 1];  // normalised context [d][e]
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, e4 = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = 8 * e4 + 4 * lh + i;  // every (d, e = l31) once
        float c = 0.f, k = 0.f;
        // the partial sums of the token blocks are independent loads: keep 8 of them in flight (at 64x64 there are 16-32
        // blocks per image and only heads x images workgroups, so this loop is a latency chain otherwise)
#pragma unroll 8
        for (int kb = 0; kb < nblk; ++kb) {
            const float* cp = ws + (((size_t)b * nblk + kb) * LA_HEADS + h) * LA_CTX;
            c += cp[d * LA_DH + l31];
            k += cp[LA_DH * LA_DH + d];
        }
        cs[d][l31] = c / k;
    }
    __syncthreads();
    // M^T (d x c) = ctx (d x e) W_out[:, head]^T (e x c) on the MFMA, one 32-channel column tile per wave: the
    // accumulator layout (row d on the registers, column c on the lanes) IS the packed layout
    const int mt = e4;
    if (mt < MT) {
        const float* wr = w_out + (size_t)(32 * mt + l31) * LA_HID + h * LA_DH;  // W_out[c][head*32 .. +32)
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int st = 0; st < 16; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cs[l31][2 * st + lh], wr[2 * st + lh], acc, 0, 0, 0);
        float* op = mz + ((((size_t)b * LA_HEADS + h) * MT + mt) * 4) * 64 * 4 + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(op + q * 64 * 4) = make_f32x4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
}

template <int C>
__global__ __launch_bounds__(256) void linattn_out_fused_kernel(const float* __restrict__ x, const LinAttnFused w,
                                                                const float* __restrict__ mz, float* __restrict__ y,
                                                                int n, int add_x, float scale) {
    constexpr int G = C / 8, XS = C + 4, TOK = LA_TOK2, MT = C / 32, Q = C / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                  // [TOK][XS]
    float* rn = smem + TOK * XS;       // [TOK]
    float* zb = rn + TOK;              // [heads][32][XS]
    const int blk = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int t0 = blk * TOK;
    const int nt = min(TOK, n - t0);

    f32x4 wq[G];
#pragma unroll
    for (int g = 0; g < G; ++g) wq[g] = *reinterpret_cast<const f32x4*>(w.wq + (((size_t)h * G + g) * 64 + lane) * 4);
    // M of this image / head: the A operand of z = M q, loaded once
    f32x4 mreg[MT][4];
    {
        const float* mzp = mz + ((size_t)b * LA_HEADS + h) * MT * 4 * 64 * 4 + lane * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) mreg[mt][e4] = *reinterpret_cast<const f32x4*>(mzp + (mt * 4 + e4) * 64 * 4);
    }
    la_stage_rows<C, TOK>(x + ((size_t)b * n + t0) * C, nt, xs, rn);
    __syncthreads();

    for (int st = 0; st < TOK / 32; ++st) {
        if (st * 32 >= nt) break;
        // q (d x tokens) = W'_q (d x C) x^T: A = weights in registers, B = x row of token l31 from LDS
        f32x16 qacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) qacc[e] = 0.f;
        const float* xr = xs + (st * 32 + l31) * XS + 4 * lh;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 8 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) qacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[g][s], a[s], qacc, 0, 0, 0);
        }
        // softmax over d (:185) for token l31: 16 values here, 16 in the other lane half; then * scale (:188)
        const float r = rn[st * 32 + l31];
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            qacc[e] *= r;
            m = fmaxf(m, qacc[e]);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        float ssum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            qacc[e] = __expf(qacc[e] - m);
            ssum += qacc[e];
        }
        ssum += __shfl_xor(ssum, 32);
        const float inv = scale / ssum;
        // this head's share of z (C x tokens) = M q: the B operand of step e is q's accumulator register e
#pragma unroll
        for (int e = 0; e < 16; ++e) qacc[e] *= inv;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x16 zacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) zacc[e] = 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    zacc = __builtin_amdgcn_mfma_f32_32x32x2f32(mreg[mt][e4][i], qacc[4 * e4 + i], zacc, 0, 0, 0);
            }
            // register e = channel 32*mt + row_of(e, lh) of token l31: four consecutive channels per b128 store
            float* zr = zb + (h * 32 + l31) * XS + 32 * mt + 4 * lh;
#pragma unroll
            for (int m4 = 0; m4 < 4; ++m4)
                *reinterpret_cast<f32x4*>(zr + 8 * m4) =
                    make_f32x4(zacc[4 * m4], zacc[4 * m4 + 1], zacc[4 * m4 + 2], zacc[4 * m4 + 3]);
        }
        __syncthreads();
        // sum over heads, + bias, RMSNorm over channels (to_out[1], :170), + x, store
#pragma unroll
        for (int i = 0; i < 32 * Q / 256; ++i) {
            const int it = tid + 256 * i;
            const int tok = it / Q, q4 = it - tok * Q;
            f32x4 z = *reinterpret_cast<const f32x4*>(w.bias + 4 * q4);
#pragma unroll
            for (int hh = 0; hh < LA_HEADS; ++hh) z += *reinterpret_cast<const f32x4*>(zb + (hh * 32 + tok) * XS + 4 * q4);
            float ss = z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w;
#pragma unroll
            for (int o = 1; o < Q; o <<= 1) ss += __shfl_xor(ss, o);  // the Q lanes of a token are consecutive
            z = z * fast_rsq(fmaxf(ss, 1e-24f)) * *reinterpret_cast<const f32x4*>(w.og + 4 * q4);
            if (add_x) z += *reinterpret_cast<const f32x4*>(xs + (st * 32 + tok) * XS + 4 * q4);
            if (st * 32 + tok < nt)
                *reinterpret_cast<f32x4*>(y + ((size_t)b * n + t0 + st * 32 + tok) * C + 4 * q4) = z;
        }
        __syncthreads();  // zb is rewritten by the next token tile
    }
}

template <int C, int TOK1>
static int launch_ctx(const LinAttnFused& w, const float* x, float* ws, int B, int n, int nblk, hipStream_t s) {
    const size_t lds1 = (size_t)(TOK1 * (C + 4) + TOK1) * 4;
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(linattn_ctx_fused_kernel<C, TOK1>), 1)) return 1;
    hipLaunchKernelGGL((linattn_ctx_fused_kernel<C, TOK1>), dim3(nblk, B), dim3(256), lds1, s, x, w, ws, n, nblk);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int C>
static int launch_c(const LinAttnFused& w, const float* x, float* ws, float* y, int B, int n, bool add_x, hipStream_t s) {
    // 256-token blocks halve the partial-context traffic, but only when they still fill the chip
    const bool big = la_tok1(C) == 256 && (size_t)B * ((n + 255) / 256) >= 512;
    const int tok1 = big ? 256 : LA_TOK1;
    const int nblk = (n + tok1 - 1) / tok1;
    const size_t lds2 = (size_t)(LA_TOK2 * (C + 4) + LA_TOK2 + LA_HEADS * 32 * (C + 4)) * 4;
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(linattn_out_fused_kernel<C>), 1)) return 1;
    const bool timed = prof::enabled();
    const double tokens = (double)B * n;
    if (timed && prof::begin("linattn_ctx_fused_kernel", 2.0 * tokens * (2.0 * LA_HID * C + LA_HID * LA_DH),
                             4.0 * tokens * C, s))
        return 1;
    if (big ? launch_ctx<C, (C <= 64 ? 256 : LA_TOK1)>(w, x, ws, B, n, nblk, s)
            : launch_ctx<C, LA_TOK1>(w, x, ws, B, n, nblk, s))
        return 1;
    if (timed && prof::end(s)) return 1;
    float* ctxn = ws + (size_t)B * nblk * LA_HEADS * LA_CTX;  // M = W_out ctx^T, lane-packed
    hipLaunchKernelGGL(linattn_ctx_reduce_kernel<C>, dim3(LA_HEADS, B), dim3(256), 0, s, ws, w.wo_raw, ctxn, nblk);
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::begin("linattn_out_fused_kernel", 2.0 * tokens * (2.0 * LA_HID * C + LA_HID * LA_DH),
                             8.0 * tokens * C, s))
        return 1;
    hipLaunchKernelGGL(linattn_out_fused_kernel<C>, dim3((n + LA_TOK2 - 1) / LA_TOK2, B), dim3(256), lds2, s, x, w,
                       ctxn, y, n, add_x ? 1 : 0, 1.0f / sqrtf((float)LA_DH));
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    return 0;
}

int launch_linattn_fused(const LinAttnFused& w, const float* x, float* ws, float* y, int B, int n, bool add_x,
                         hipStream_t s) {
    DM_REQUIRE(w.C == 64 || w.C == 128, "fused LinearAttention: C must be 64 or 128");
    if (w.C == 64) return launch_c<64>(w, x, ws, y, B, n, add_x, s);
    return launch_c<128>(w, x, ws, y, B, n, add_x, s);
}

}  // namespace dm

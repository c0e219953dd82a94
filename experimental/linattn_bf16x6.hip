// EXPERIMENTAL, OFF BY DEFAULT (DM_LINATTN_BF16X6=1): the fused LinearAttention of linattn_fused.hip with its
// fp32 products formed on the bf16 matrix cores.
//
// The f32-input MFMA runs at 1/16 of the bf16 rate (and on the vector ALUs).  Here every fp32
// operand is split into three round-to-nearest bf16 terms, x = h + m + l (exact to 2^-27 |x|), and a product a*b is the
// sum of the six partial products  ah*bl + al*bh + am*bm + ah*bm + am*bh + ah*bh  issued as v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation; the dropped terms (am*bl, al*bm, al*bl) are below 2^-25 |a||b|, i.e. under the rounding of a native
// fp32 FMA chain (tools/bf16x6_microbench.hip: 1.05e-6 rel-L2 on a K = 4608 GEMM against 1.21e-6 for the f32 MFMA).
// Six 32-cycle instructions cover a 32x32x16 step that takes eight 64-cycle f32 MFMAs: 2.7x fewer matrix cycles.
//
// What is split where:
//   x rows (the A / B operand of the three projections)  once per workgroup, while staging them into LDS (three bf16 planes)
//   projection weights, with the RMSNorm gain folded in    on the host
//   M = W_out ctx^T (A operand of z = M q)                 in the reduce kernel, once per (image, head)
//   softmax(q) (B operand of z = M q)                      in registers, per token tile (the only per-step VALU split)
// The softmax / exp / context parts are the fp32 code of linattn_fused.hip (the context product stays on the f32 MFMA).
// C = 64 only (the 32x32 and 64x64 stages, where the layer is matrix bound).
#include "conv_device.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace dm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

static constexpr int LB_DH = 32, LB_HEADS = 4, LB_HID = 128;
static constexpr int LB_CTX = LB_DH * LB_DH + LB_DH;
static constexpr int LB_TOK1 = 128, LB_TOK2 = 64;

__host__ __device__ static inline int lb_row_of(int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; }

// ---- host: round-to-nearest-even bf16 split of finite floats ----------------------------------------------------
static inline uint16_t f2bf(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline void split3_host(float x, uint16_t out[3]) {
    out[0] = f2bf(x);
    const float r1 = x - bf2f(out[0]);
    out[1] = f2bf(r1);
    out[2] = f2bf(r1 - bf2f(out[1]));
}

// Projection `which` (0 q, 1 k, 2 v) of w_qkv (3*128, C) with the gain folded in, as bf16 triples in MFMA lane order:
// [head][g = C/16][term][lane = half*32 + l][8]: W'[head*32 + l][16 g + 8 half + j].  Returned as raw 32-bit words.
void linattn_bf16x6_pack_proj(const float* w_qkv, const float* norm_g, int C, int which, std::vector<float>& dst) {
    const int G16 = C / 16;
    const float sq = std::sqrt((float)C);
    std::vector<uint16_t> tmp((size_t)LB_HEADS * G16 * 3 * 64 * 8);
    for (int h = 0; h < LB_HEADS; ++h)
        for (int g = 0; g < G16; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int l = lane & 31, half = lane >> 5;
                    const int row = which * LB_HID + h * LB_DH + l, c = 16 * g + 8 * half + j;
                    uint16_t t[3];
                    split3_host(w_qkv[(size_t)row * C + c] * (norm_g[c] * sq), t);
                    for (int term = 0; term < 3; ++term)
                        tmp[((((size_t)h * G16 + g) * 3 + term) * 64 + lane) * 8 + j] = t[term];
                }
    dst.resize(tmp.size() / 2);
    std::memcpy(dst.data(), tmp.data(), tmp.size() * 2);
}

// ---- device -------------------------------------------------------------------------------------------------------
struct Split3 {
    bf16x8 h, m, l;
};
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];  // v_cvt_pk_bf16_f32: round to nearest even
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        s.h[i] = h;
        s.m[i] = m;
        s.l[i] = (__bf16)(r1 - (float)m);
    }
    return s;
}
// acc += a * b on the bf16 cores, small terms first
__device__ __forceinline__ f32x16 mfma6(const bf16x8& ah, const bf16x8& am, const bf16x8& al, const bf16x8& bh,
                                        const bf16x8& bm, const bf16x8& bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
}

// x rows [t0, t0 + TOK) -> three bf16 planes [TOK][C + 8] (+ optionally an fp32 copy) in LDS, rn[t] = 1 / max(||x_t||, 1e-12)
template <int C, int TOK, bool KEEP_F32>
__device__ __forceinline__ void lb_stage_rows(const float* __restrict__ xb, int nt, __bf16* xh, __bf16* xm, __bf16* xl,
                                              float* xs, float* rn) {
    constexpr int XSB = C + 8, Q = C / 4;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TOK * Q / 256; ++i) {
        const int it = tid + 256 * i;
        const int row = it / Q, q = it - row * Q;
        f32x4 v = make_f32x4(0.f, 0.f, 0.f, 0.f);
        if (row < nt) v = *reinterpret_cast<const f32x4*>(xb + (size_t)row * C + 4 * q);
        bf16x4 h, m, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const __bf16 hh = (__bf16)v[k];
            const float r1 = v[k] - (float)hh;
            const __bf16 mm = (__bf16)r1;
            h[k] = hh;
            m[k] = mm;
            l[k] = (__bf16)(r1 - (float)mm);
        }
        *reinterpret_cast<bf16x4*>(xh + row * XSB + 4 * q) = h;
        *reinterpret_cast<bf16x4*>(xm + row * XSB + 4 * q) = m;
        *reinterpret_cast<bf16x4*>(xl + row * XSB + 4 * q) = l;
        if constexpr (KEEP_F32) *reinterpret_cast<f32x4*>(xs + row * (C + 4) + 4 * q) = v;
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
        for (int o = 1; o < Q; o <<= 1) ss += __shfl_xor(ss, o);  // the Q lanes of a row are consecutive
        if (q == 0) rn[row] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    }
}

template <int C>
__global__ __launch_bounds__(256) void linattn_ctx_bf16x6_kernel(const float* __restrict__ x, const LinAttnFused w,
                                                                 float* __restrict__ ws, int n, int nblk) {
    constexpr int G16 = C / 16, XSB = C + 8, TOK = LB_TOK1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* xh = reinterpret_cast<__bf16*>(smem);
    __bf16* xm = xh + TOK * XSB;
    __bf16* xl = xm + TOK * XSB;
    float* rn = reinterpret_cast<float*>(xl + TOK * XSB);
    const int blk = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int t0 = blk * TOK;
    const int nt = min(TOK, n - t0);

    bf16x8 wk[G16][3], wv[G16][3];
    {
        const bf16x8* pk = reinterpret_cast<const bf16x8*>(w.wk3) + (size_t)h * G16 * 3 * 64 + lane;
        const bf16x8* pv = reinterpret_cast<const bf16x8*>(w.wv3) + (size_t)h * G16 * 3 * 64 + lane;
#pragma unroll
        for (int g = 0; g < G16; ++g)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                wk[g][t] = pk[(g * 3 + t) * 64];
                wv[g][t] = pv[(g * 3 + t) * 64];
            }
    }
    const float kb = w.kbound[h * LB_DH + l31];
    lb_stage_rows<C, TOK, false>(x + ((size_t)b * n + t0) * C, nt, xh, xm, xl, nullptr, rn);
    __syncthreads();

    f32x16 cacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) cacc[e] = 0.f;
    float ksum = 0.f;
    if (blk == 0) {
        const float* mk = w.mem_kv + (size_t)h * LB_DH * 4;
        const float* mv = w.mem_kv + (size_t)(LB_HEADS + h) * LB_DH * 4;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const float a = __expf(mk[l31 * 4 + j + lh] - kb);
            ksum += a;
            cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, mv[l31 * 4 + j + lh], cacc, 0, 0, 0);
        }
    }
    for (int st = 0; st < TOK / 32; ++st) {
        if (st * 32 >= nt) break;
        f32x16 kacc, vacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            kacc[e] = 0.f;
            vacc[e] = 0.f;
        }
        const int ro = (st * 32 + l31) * XSB + 8 * lh;
#pragma unroll
        for (int g = 0; g < G16; ++g) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xh + ro + 16 * g);
            const bf16x8 am = *reinterpret_cast<const bf16x8*>(xm + ro + 16 * g);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(xl + ro + 16 * g);
            kacc = mfma6(ah, am, al, wk[g][0], wk[g][1], wk[g][2], kacc);
            vacc = mfma6(ah, am, al, wv[g][0], wv[g][1], wv[g][2], vacc);
        }
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
    float* cp = ws + (((size_t)b * nblk + blk) * LB_HEADS + h) * LB_CTX;
#pragma unroll
    for (int e = 0; e < 16; ++e) cp[lb_row_of(e, lh) * LB_DH + l31] = cacc[e];
    ksum += __shfl_xor(ksum, 32);
    if (lh == 0) cp[LB_DH * LB_DH + l31] = ksum;
}

// As linattn_ctx_reduce_kernel, but M^T leaves as bf16 triples in the k-slot order of the 32x32x16 MFMA:
// [image][head][mt][ks][term][lane][8], element j of lane (c = l, half) = M[32 mt + c][row_of(8 ks + j, half)].
template <int C>
__global__ __launch_bounds__(256) void linattn_reduce_bf16x6_kernel(const float* __restrict__ ws,
                                                                    const float* __restrict__ w_out,
                                                                    bf16x8* __restrict__ mz3, int nblk) {
    constexpr int MT = C / 32;
    __shared__ float cs[LB_DH][LB_DH + 1];
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, e4 = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = 8 * e4 + 4 * lh + i;
        float c = 0.f, k = 0.f;
        for (int kb = 0; kb < nblk; ++kb) {
            const float* cp = ws + (((size_t)b * nblk + kb) * LB_HEADS + h) * LB_CTX;
            c += cp[d * LB_DH + l31];
            k += cp[LB_DH * LB_DH + d];
        }
        cs[d][l31] = c / k;
    }
    __syncthreads();
    const int mt = e4;
    if (mt < MT) {
        const float* wr = w_out + (size_t)(32 * mt + l31) * LB_HID + h * LB_DH;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int st = 0; st < 16; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cs[l31][2 * st + lh], wr[2 * st + lh], acc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = acc[8 * ks + j];
            const Split3 s = split3(v);
            bf16x8* op = mz3 + (((((size_t)b * LB_HEADS + h) * MT + mt) * 2 + ks) * 3) * 64 + lane;
            op[0] = s.h;
            op[64] = s.m;
            op[128] = s.l;
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void linattn_out_bf16x6_kernel(const float* __restrict__ x, const LinAttnFused w,
                                                                 const bf16x8* __restrict__ mz3, float* __restrict__ y,
                                                                 int n, int add_x, float scale) {
    constexpr int G16 = C / 16, XS = C + 4, XSB = C + 8, TOK = LB_TOK2, MT = C / 32, Q = C / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;              // [TOK][XS] fp32 copy for the residual
    float* rn = xs + TOK * XS;     // [TOK]
    float* zb = rn + TOK;          // [heads][32][XS]
    __bf16* xh = reinterpret_cast<__bf16*>(zb + LB_HEADS * 32 * XS);
    __bf16* xm = xh + TOK * XSB;
    __bf16* xl = xm + TOK * XSB;
    const int blk = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int t0 = blk * TOK;
    const int nt = min(TOK, n - t0);

    bf16x8 wq[G16][3];
    {
        const bf16x8* pq = reinterpret_cast<const bf16x8*>(w.wq3) + (size_t)h * G16 * 3 * 64 + lane;
#pragma unroll
        for (int g = 0; g < G16; ++g)
#pragma unroll
            for (int t = 0; t < 3; ++t) wq[g][t] = pq[(g * 3 + t) * 64];
    }
    bf16x8 mreg[MT][2][3];
    {
        const bf16x8* mp = mz3 + ((size_t)b * LB_HEADS + h) * MT * 2 * 3 * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 3; ++t) mreg[mt][ks][t] = mp[((mt * 2 + ks) * 3 + t) * 64];
    }
    lb_stage_rows<C, TOK, true>(x + ((size_t)b * n + t0) * C, nt, xh, xm, xl, xs, rn);
    __syncthreads();

    for (int st = 0; st < TOK / 32; ++st) {
        if (st * 32 >= nt) break;
        f32x16 qacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) qacc[e] = 0.f;
        const int ro = (st * 32 + l31) * XSB + 8 * lh;
#pragma unroll
        for (int g = 0; g < G16; ++g) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(xh + ro + 16 * g);
            const bf16x8 bm = *reinterpret_cast<const bf16x8*>(xm + ro + 16 * g);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(xl + ro + 16 * g);
            qacc = mfma6(wq[g][0], wq[g][1], wq[g][2], bh, bm, bl, qacc);
        }
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
        // softmax(q) * scale as the B operand of z = M q: registers 8 ks .. 8 ks + 7 are the k slots of step ks
        Split3 qs[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = qacc[8 * ks + j] * inv;
            qs[ks] = split3(v);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x16 zacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) zacc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                zacc = mfma6(mreg[mt][ks][0], mreg[mt][ks][1], mreg[mt][ks][2], qs[ks].h, qs[ks].m, qs[ks].l, zacc);
            float* zr = zb + (h * 32 + l31) * XS + 32 * mt + 4 * lh;
#pragma unroll
            for (int m4 = 0; m4 < 4; ++m4)
                *reinterpret_cast<f32x4*>(zr + 8 * m4) =
                    make_f32x4(zacc[4 * m4], zacc[4 * m4 + 1], zacc[4 * m4 + 2], zacc[4 * m4 + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32 * Q / 256; ++i) {
            const int it = tid + 256 * i;
            const int tok = it / Q, q4 = it - tok * Q;
            f32x4 z = *reinterpret_cast<const f32x4*>(w.bias + 4 * q4);
#pragma unroll
            for (int hh = 0; hh < LB_HEADS; ++hh) z += *reinterpret_cast<const f32x4*>(zb + (hh * 32 + tok) * XS + 4 * q4);
            float ss = z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w;
#pragma unroll
            for (int o = 1; o < Q; o <<= 1) ss += __shfl_xor(ss, o);
            z = z * fast_rsq(fmaxf(ss, 1e-24f)) * *reinterpret_cast<const f32x4*>(w.og + 4 * q4);
            if (add_x) z += *reinterpret_cast<const f32x4*>(xs + (st * 32 + tok) * XS + 4 * q4);
            if (st * 32 + tok < nt)
                *reinterpret_cast<f32x4*>(y + ((size_t)b * n + t0 + st * 32 + tok) * C + 4 * q4) = z;
        }
        __syncthreads();
    }
}

int launch_linattn_bf16x6(const LinAttnFused& w, const float* x, float* ws, float* y, int B, int n, bool add_x,
                          hipStream_t s) {
    constexpr int C = 64;
    DM_REQUIRE(w.C == C && w.wq3 && w.wk3 && w.wv3, "bf16x6 LinearAttention: C == 64 with split weights");
    const int nblk = (n + LB_TOK1 - 1) / LB_TOK1;
    const size_t lds1 = (size_t)3 * LB_TOK1 * (C + 8) * 2 + LB_TOK1 * 4;
    const size_t lds2 = (size_t)(LB_TOK2 * (C + 4) + LB_TOK2 + LB_HEADS * 32 * (C + 4)) * 4 + (size_t)3 * LB_TOK2 * (C + 8) * 2;
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(linattn_ctx_bf16x6_kernel<C>), 2)) return 1;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(linattn_out_bf16x6_kernel<C>), 2)) return 1;
    const bool timed = prof::enabled();
    const double tokens = (double)B * n;
    if (timed && prof::begin("linattn_ctx_bf16x6_kernel", 2.0 * tokens * (2.0 * LB_HID * C + LB_HID * LB_DH),
                             4.0 * tokens * C, s))
        return 1;
    hipLaunchKernelGGL(linattn_ctx_bf16x6_kernel<C>, dim3(nblk, B), dim3(256), lds1, s, x, w, ws, n, nblk);
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    bf16x8* mz3 = reinterpret_cast<bf16x8*>(ws + (size_t)B * nblk * LB_HEADS * LB_CTX);
    hipLaunchKernelGGL(linattn_reduce_bf16x6_kernel<C>, dim3(LB_HEADS, B), dim3(256), 0, s, ws, w.wo_raw, mz3, nblk);
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::begin("linattn_out_bf16x6_kernel", 2.0 * tokens * (2.0 * LB_HID * C + LB_HID * LB_DH),
                             8.0 * tokens * C, s))
        return 1;
    hipLaunchKernelGGL(linattn_out_bf16x6_kernel<C>, dim3((n + LB_TOK2 - 1) / LB_TOK2, B), dim3(256), lds2, s, x, w, mz3,
                       y, n, add_x ? 1 : 0, 1.0f / sqrtf((float)LB_DH));
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    return 0;
}

}  // namespace dm

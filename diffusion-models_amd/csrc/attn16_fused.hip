// Full self-attention over a 4x4 feature map as ONE kernel (Attention.forward, DD/denoising_diffusion.py:215-229, with
// Attend.forward DD/attend.py:109-124 and the caller's `attn(x) + x`, :363 / :371 / :383): RMSNorm -> to_qkv (1x1, no
// bias) -> 4 heads x 32, 4 learned memory key/values per head -> softmax(q k^T / sqrt(32)) v -> to_out (1x1 + bias) -> + x.
//
// At 32x32 inputs the three full-attention layers of the U-Net see 16 tokens of 256 / 512 channels per image: as five
// launches each (norm, to_qkv K-split + landing, attention core, to_out) they cost 174 us of a 4.6 ms step at batch
// 256 -- latency, not work.  Here one workgroup = one image, one wave = one head, everything between x and y stays in
// registers / LDS, on v_mfma_f32_16x16x4_f32 (16 tokens = the 16 MFMA rows).
//
// MFMA bookkeeping.  A D tile holds element (row 4 kq + e, column l15) in register e of lane (l15, kq).  Used as the B
// operand of step e of the next product it contributes B[k = kq][col = l15], used as the A operand A[row = l15][k = kq]:
// either way the product contracts over the tile's ROW index, with no data movement.  So the projections are produced in
// the orientation their consumer contracts over:
//     qT, kT (d x tokens) = W' x^T       v (tokens x d) = x^ W'^T            (x^ = x / |x|; g sqrt(C) and the softmax scale
//     S^T (keys x queries) = (kT)^T-as-A . qT-as-B      contraction over d    are folded into W' on the host)
//     O^T (d x queries)    = v-as-A . P^T-as-B          contraction over keys; P^T = softmax over the rows of S^T
// and the memory keys / values enter as two more 16x16 tiles loaded in D layout.  O^T goes through LDS once (8 KB) so
// that to_out can split the OUTPUT channels over the four waves with the full K = 128 each.
// Global operands (weights) are 16-byte loads in lane order, packed on the host; the K index of MFMA step j of
// iteration s is channel 16 s + 4 kq + j (a fixed permutation, as in pw_mfma.hip).
#include "conv_device.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace dm {

static constexpr int A16_OS = 132;  // row stride (floats) of the O staging tile: rows 4 banks apart
static constexpr int A16_D = 4;     // weight iterations in flight

bool attn16_eligible(int dim, int heads, int dh) {
    static const bool off = std::getenv("DM_NO_ATTN16") != nullptr;
    return !off && heads == 4 && dh == 32 && dim % 256 == 0 && dim >= 256 && dim <= 1024;  // 4 channel tiles per wave and pass
}

// w_qkv (384, C) rows [q | k | v] x (head, d); norm_g (C); w_out (C, 128).
//   wp[s][tile 24][lane][j] = W'[16 tile + l15][16 s + 4 kq + j]   W' = w_qkv * g * sqrt(C) (q rows also * 32^-1/2)
//   wo[s][ct C/16][lane][j] = w_out[16 ct + l15][16 s + 4 kq + j]
void attn16_pack(const float* w_qkv, const float* norm_g, const float* w_out, int C, std::vector<float>& wp,
                 std::vector<float>& wo) {
    const float sc = std::sqrt((float)C), qs = 1.0f / std::sqrt(32.0f);
    wp.assign((size_t)384 * C, 0.f);
    for (int s = 0; s < C / 16; ++s)
        for (int t = 0; t < 24; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int row = 16 * t + (lane & 15), ch = 16 * s + 4 * (lane >> 4) + j;
                    float v = w_qkv[(size_t)row * C + ch] * norm_g[ch] * sc;
                    if (row < 128) v *= qs;
                    wp[(((size_t)s * 24 + t) * 64 + lane) * 4 + j] = v;
                }
    wo.assign((size_t)C * 128, 0.f);
    for (int s = 0; s < 8; ++s)
        for (int ct = 0; ct < C / 16; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    wo[(((size_t)s * (C / 16) + ct) * 64 + lane) * 4 + j] =
                        w_out[(size_t)(16 * ct + (lane & 15)) * 128 + 16 * s + 4 * (lane >> 4) + j];
}

__device__ __forceinline__ f32x4 a16_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256) void attn16_fused_kernel(const float* __restrict__ x, const Attn16 w,
                                                            float* __restrict__ y, int add_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = w.C;
    const int XS = C + 4;  // rows 4 banks apart: the 16 tokens of a ds_read_b128 cover the 64 banks
    float* xs = smem;
    float* ol = smem + 16 * XS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = head
    const int l15 = lane & 15, kq = lane >> 4;
    const size_t img = blockIdx.x;

    // ---- projection weights of this head: a ring of A16_D iterations, requested before anything else (they do not
    //      depend on x); tile order in a ring slot: q0 q1 k0 k1 v0 v1
    const int S = C / 16;
    const float* wl = w.wp + (size_t)lane * 4;
    f32x4 wr[A16_D][6];
    auto load = [&](int s, int d) {
        const float* ws = wl + (size_t)s * (24 * 256);
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                wr[d][2 * part + t] = *reinterpret_cast<const f32x4*>(ws + (8 * part + 2 * h + t) * 256);
    };
#pragma unroll
    for (int d = 0; d < A16_D; ++d) load(min(d, S - 1), d);

    // ---- x^ = x / max(|x|, 1e-12) -> LDS: 16 threads (one DPP row) per token, the row held in registers (C <= 1024)
    {
        const int token = tid >> 4, li = tid & 15;
        const float* xr = x + (img * 16 + token) * C;
        f32x4 xv[16];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = 4 * li + 64 * i;
            if (c < C) {
                xv[i] = *reinterpret_cast<const f32x4*>(xr + c);
                ss += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
            }
        }
        ss += dpp_f<0xB1>(ss);
        ss += dpp_f<0x4E>(ss);
        ss += dpp_f<0x141>(ss);
        ss += dpp_f<0x140>(ss);
        const float rn = fast_rsq(fmaxf(ss, 1e-24f));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = 4 * li + 64 * i;
            if (c < C) *reinterpret_cast<f32x4*>(xs + token * XS + c) = xv[i] * rn;
        }
    }
    __syncthreads();

    // ---- projections of this head: qT, kT (d x tokens), v (tokens x d); two 16-wide d tiles each
    const f32x4 z4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    f32x4 qT[2] = {z4, z4}, kT[2] = {z4, z4}, vv[2] = {z4, z4};
    {
        const float* xrow = xs + l15 * XS + 4 * kq;
        for (int s = 0; s < S; s += A16_D) {
#pragma unroll
            for (int d = 0; d < A16_D; ++d) {
                if (s + d < S) {
                    const f32x4 xa = *reinterpret_cast<const f32x4*>(xrow + 16 * (s + d));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            qT[t] = a16_mfma(wr[d][t][j], xa[j], qT[t]);
                            kT[t] = a16_mfma(wr[d][2 + t][j], xa[j], kT[t]);
                            vv[t] = a16_mfma(xa[j], wr[d][4 + t][j], vv[t]);
                        }
                    }
                    if (s + d + A16_D < S) load(s + d + A16_D, d);
                }
            }
        }
    }

    // ---- to_out weights: wave h finishes the output channels [h C/4, (h + 1) C/4), four 16-channel tiles per pass, K = 128
    //      = 8 iterations per pass; a ring of 4 iterations over the flattened (pass, s) index, started here so that the
    //      first loads fly during the softmax
    const int CT = C / 16;
    const int ct_per_wave = CT / 4;
    const int n_it = (ct_per_wave / 4) * 8;
    const float* wol = w.wo + (size_t)lane * 4;
    f32x4 wv4[4][4];
    auto load_o = [&](int it, int slot) {
        const int c0 = (it >> 3) * 4, so = it & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wv4[slot][i] = *reinterpret_cast<const f32x4*>(wol + ((size_t)so * CT + h * ct_per_wave + c0 + i) * 256);
    };
#pragma unroll
    for (int d = 0; d < 4; ++d) load_o(d, d);

    // ---- scores, keys on the rows: S^T = k q^T (16 keys) and the 4 memory keys (rows 0..3 of a second tile)
    const float* mk = w.mem_kv + (size_t)h * 4 * 32;
    const float* mv = w.mem_kv + (size_t)(4 + h) * 4 * 32;
    f32x4 st = z4, sm = z4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // memory keys in D layout of a (d x m) tile: lane (m = l15, kq), register e = mk[m][16 t + 4 kq + e]
        const f32x4 km = l15 < 4 ? *reinterpret_cast<const f32x4*>(mk + l15 * 32 + 16 * t + 4 * kq) : z4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            st = a16_mfma(kT[t][e], qT[t][e], st);
            sm = a16_mfma(km[e], qT[t][e], sm);
        }
    }
    // softmax over the keys of query l15: registers e and the four lane groups kq; memory keys live in group 0 only
    const bool g0 = kq == 0;
    float mx = fmaxf(fmaxf(st.x, st.y), fmaxf(st.z, st.w));
    if (g0) mx = fmaxf(mx, fmaxf(fmaxf(sm.x, sm.y), fmaxf(sm.z, sm.w)));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    f32x4 pt, pm;
    pt.x = __expf(st.x - mx);
    pt.y = __expf(st.y - mx);
    pt.z = __expf(st.z - mx);
    pt.w = __expf(st.w - mx);
    pm.x = g0 ? __expf(sm.x - mx) : 0.f;
    pm.y = g0 ? __expf(sm.y - mx) : 0.f;
    pm.z = g0 ? __expf(sm.z - mx) : 0.f;
    pm.w = g0 ? __expf(sm.w - mx) : 0.f;
    float sum = (pt.x + pt.y) + (pt.z + pt.w) + (pm.x + pm.y) + (pm.z + pm.w);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    pt = pt * inv;
    pm = pm * inv;

    // ---- O^T (d x queries) = v^T P^T + v_mem^T P_mem^T, then to LDS as O[token][head * 32 + d]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x4 ot = z4;
        // memory values in D layout of an (m x d) tile: lane (d = l15, kq), register e = mv[4 kq + e][16 t + l15]
        f32x4 vm = z4;
        if (g0) {
            vm.x = mv[0 * 32 + 16 * t + l15];
            vm.y = mv[1 * 32 + 16 * t + l15];
            vm.z = mv[2 * 32 + 16 * t + l15];
            vm.w = mv[3 * 32 + 16 * t + l15];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ot = a16_mfma(vv[t][e], pt[e], ot);
            ot = a16_mfma(vm[e], pm[e], ot);
        }
        *reinterpret_cast<f32x4*>(ol + l15 * A16_OS + h * 32 + 16 * t + 4 * kq) = ot;
    }
    __syncthreads();

    // ---- to_out: wave h finishes the output channels [h C/4, (h + 1) C/4); K = 128 = 8 iterations
    f32x4 ob[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) ob[s] = *reinterpret_cast<const f32x4*>(ol + l15 * A16_OS + 16 * s + 4 * kq);
    const size_t row = (img * 16 + l15) * C;  // this lane's token
    for (int c0 = 0; c0 < ct_per_wave; c0 += 4) {
        f32x4 acc[4] = {z4, z4, z4, z4};
        const int it0 = (c0 >> 2) * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = a16_mfma(wv4[s & 3][i][j], ob[s][j], acc[i]);
            if (it0 + s + 4 < n_it) load_o(it0 + s + 4, s & 3);
        }
        // D tile: rows = channels 16 ct + 4 kq + e, columns = tokens: four consecutive channels of token l15 per lane
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 16 * (h * ct_per_wave + c0 + i) + 4 * kq;
            f32x4 r = acc[i] + *reinterpret_cast<const f32x4*>(w.bias + c);
            if (add_x) r += *reinterpret_cast<const f32x4*>(x + row + c);
            *reinterpret_cast<f32x4*>(y + row + c) = r;
        }
    }
}

int launch_attn16_fused(const Attn16& w, const float* x, float* y, int B, bool add_x, hipStream_t s) {
    DM_REQUIRE(w.C % 256 == 0 && w.C >= 256 && w.C <= 1024 && B > 0, "attn16: channel count");
    const size_t lds = (size_t)(16 * (w.C + 4) + 16 * A16_OS) * sizeof(float);
    static LdsOptIn lds_flag;
    if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(attn16_fused_kernel), 1)) return 1;
    const bool timed = prof::enabled();
    if (timed) {
        const double tok = 16.0 * B;
        const double flops = 2.0 * tok * (384.0 * w.C + 128.0 * w.C + 2.0 * 4 * 20 * 32);
        const double bytes = 4.0 * (2.0 * tok * w.C + 512.0 * w.C);
        if (prof::begin("attn16_fused_kernel", flops, bytes, s)) return 1;
    }
    hipLaunchKernelGGL(attn16_fused_kernel, dim3(B), dim3(256), lds, s, x, w, y, add_x ? 1 : 0);
    DM_CHECK_HIP(hipGetLastError());
    if (timed && prof::end(s)) return 1;
    return 0;
}

}  // namespace dm

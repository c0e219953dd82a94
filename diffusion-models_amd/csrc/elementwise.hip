// Bandwidth-bound kernels of the sampling path (gfx950): layout changes at the NCHW boundary,
// RMSNorm / GroupNorm, the small Linear layers of the time embedding, the DDPM/DDIM update and
// the Philox noise generator.  One wavefront = 64 lanes throughout.
#include "dm_common.h"

namespace dm {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// ---------------------------------------------------------------------------------------
// NCHW <-> NHWC (boundary tensors only: 3-4 channels)
// ---------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = i % C;
    int64_t r = i / C;
    int p = r % HW;
    int64_t b = r / HW;
    out[i] = in[(b * C + c) * HW + p];
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = i % HW;
    int64_t r = i / HW;
    int c = r % C;
    int64_t b = r / C;
    out[i] = in[(b * HW + p) * C + c];
}
int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, hipStream_t s) {
    int64_t n = (int64_t)B * C * HW;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, C, HW, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, hipStream_t s) {
    int64_t n = (int64_t)B * C * HW;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, C, HW, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// RMSNorm (+ scale/shift, SiLU, residual): one wavefront per pixel row of C channels.
// Restates RMSNorm.forward (DD/denoising_diffusion.py:66-67) and the tail of Block.forward
// (:115-121) for layers whose output channels do not fit one conv workgroup.
// ---------------------------------------------------------------------------------------
// Also the landing kernel of split-K convolutions: sums the K-split partial tiles and adds the bias first.
template <int MAXV>  // MAXV > 0: the row (C <= 64*MAXV) is held in registers; 0: re-read (any C)
__global__ __launch_bounds__(256) void norm_act_kernel(const float* __restrict__ x, int nsplit, int64_t split_stride,
                                                       const float* __restrict__ bias, const float* __restrict__ g,
                                                       const float* __restrict__ scale, int ss_stride,
                                                       int pix_per_image, const float* __restrict__ residual,
                                                       float* __restrict__ y, int64_t rows, int C, int flags,
                                                       int res_nsplit, int64_t res_stride,
                                                       const float* __restrict__ res_bias) {
    const int lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float* yr = y + row * C;
    auto value = [&](int c) {
        float v = xr[c];
        for (int sp = 1; sp < nsplit; ++sp) v += xr[(size_t)sp * split_stride + c];
        if (flags & EPI_BIAS) v += bias[c];
        return v;
    };
    float vals[MAXV > 0 ? MAXV : 1];
    float ss = 0.f;
    if constexpr (MAXV > 0) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + 64 * i;
            vals[i] = c < C ? value(c) : 0.f;
            ss += vals[i] * vals[i];
        }
    } else if (flags & EPI_NORM) {
        for (int c = lane; c < C; c += 64) {
            float v = value(c);
            ss += v * v;
        }
    }
    float rn = 1.0f;
    if (flags & EPI_NORM) {
        ss = wave_sum(ss);
        rn = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
    }
    const float* sp = nullptr;
    if (flags & EPI_SCALE_SHIFT) sp = scale + (row / pix_per_image) * (int64_t)ss_stride;
    auto finish = [&](int c, float v) {
        if (flags & EPI_NORM) v = v * rn * g[c];
        if (flags & EPI_SCALE_SHIFT) v = v * (sp[c] + 1.0f) + sp[C + c];
        if (flags & EPI_SILU) v = silu_f(v);
        if (flags & EPI_RELU) v = fmaxf(v, 0.0f);
        if (flags & EPI_RESIDUAL) {
            float r = residual[row * C + c];
            for (int sp2 = 1; sp2 < res_nsplit; ++sp2) r += residual[(size_t)sp2 * res_stride + row * C + c];
            if (res_bias) r += res_bias[c];
            v += r;
        }
        yr[c] = v;
    };
    if constexpr (MAXV > 0) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + 64 * i;
            if (c < C) finish(c, vals[i]);
        }
    } else {
        for (int c = lane; c < C; c += 64) finish(c, value(c));
    }
}
// 16-byte version for C = 4*L*NV: a row occupies L lanes (NV float4 each), 64/L rows per wavefront, so a
// 64-channel row no longer costs a whole wave one dword per lane.  Same arithmetic as norm_act_kernel.
using f32x4v = __attribute__((ext_vector_type(4))) float;
template <int L, int NV>
__global__ __launch_bounds__(256) void norm_act_vec_kernel(const float* __restrict__ x, int nsplit,
                                                           int64_t split_stride, const float* __restrict__ bias,
                                                           const float* __restrict__ g,
                                                           const float* __restrict__ scale, int ss_stride,
                                                           int pix_per_image, const float* __restrict__ residual,
                                                           float* __restrict__ y, int64_t rows, int flags,
                                                           int res_nsplit, int64_t res_stride,
                                                           const float* __restrict__ res_bias) {
    constexpr int C = 4 * L * NV;
    constexpr int RPW = 64 / L;  // rows per wavefront
    const int lane = threadIdx.x & 63;
    const int sub = lane / L, li = lane % L;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
    const bool ok = row < rows;
    const int64_t rrow = ok ? row : rows - 1;  // keep every lane in the shuffles
    f32x4v v[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (li + L * i) * 4;
        const float* xp = x + rrow * C + c;
        v[i] = *reinterpret_cast<const f32x4v*>(xp);
        for (int sp = 1; sp < nsplit; ++sp) v[i] += *reinterpret_cast<const f32x4v*>(xp + (size_t)sp * split_stride);
        if (flags & EPI_BIAS) v[i] += *reinterpret_cast<const f32x4v*>(bias + c);
        ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    float rn = 1.0f;
    if (flags & EPI_NORM) {
#pragma unroll
        for (int m = L / 2; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
        rn = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
    }
    if (!ok) return;
    const float* sp = nullptr;
    if (flags & EPI_SCALE_SHIFT) sp = scale + (row / pix_per_image) * (int64_t)ss_stride;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (li + L * i) * 4;
        f32x4v t = v[i];
        if (flags & EPI_NORM) t = t * rn * *reinterpret_cast<const f32x4v*>(g + c);
        if (flags & EPI_SCALE_SHIFT)
            t = t * (*reinterpret_cast<const f32x4v*>(sp + c) + 1.0f) + *reinterpret_cast<const f32x4v*>(sp + C + c);
        if (flags & EPI_SILU) {
            t.x = silu_f(t.x);
            t.y = silu_f(t.y);
            t.z = silu_f(t.z);
            t.w = silu_f(t.w);
        }
        if (flags & EPI_RELU) {
            t.x = fmaxf(t.x, 0.f);
            t.y = fmaxf(t.y, 0.f);
            t.z = fmaxf(t.z, 0.f);
            t.w = fmaxf(t.w, 0.f);
        }
        if (flags & EPI_RESIDUAL) {
            const float* rp = residual + row * C + c;
            f32x4v r = *reinterpret_cast<const f32x4v*>(rp);
            for (int sp2 = 1; sp2 < res_nsplit; ++sp2) r += *reinterpret_cast<const f32x4v*>(rp + (size_t)sp2 * res_stride);
            if (res_bias) r += *reinterpret_cast<const f32x4v*>(res_bias + c);
            t += r;
        }
        *reinterpret_cast<f32x4v*>(y + row * C + c) = t;
    }
}

int launch_norm_act(const float* x, int nsplit, int64_t split_stride, const float* bias, const float* g,
                    const float* scale, int ss_stride, int pix_per_image, const float* residual, float* y,
                    int64_t rows, int C, int flags, hipStream_t s, int res_nsplit, int64_t res_stride,
                    const float* res_bias) {
#define DM_NORM_VEC(L_, NV_)                                                                                       \
    {                                                                                                              \
        const int64_t rpb = 4 * (64 / L_);                                                                         \
        hipLaunchKernelGGL((norm_act_vec_kernel<L_, NV_>), dim3((rows + rpb - 1) / rpb), dim3(256), 0, s, x, nsplit, \
                           split_stride, bias, g, scale, ss_stride, pix_per_image, residual, y, rows, flags,      \
                           res_nsplit, res_stride, res_bias);                                                      \
        DM_CHECK_HIP(hipGetLastError());                                                                           \
        return 0;                                                                                                  \
    }
    const bool aligned = (split_stride % 4 == 0) && (ss_stride % 4 == 0) && (res_stride % 4 == 0);
    if (aligned) {
        if (C == 64) DM_NORM_VEC(16, 1)
        if (C == 128) DM_NORM_VEC(32, 1)
        if (C == 256) DM_NORM_VEC(64, 1)
        if (C == 384) DM_NORM_VEC(32, 3)
        if (C == 512) DM_NORM_VEC(64, 2)
    }
#undef DM_NORM_VEC
    dim3 grid((rows + 3) / 4), block(256);
    if (C <= 128)
        hipLaunchKernelGGL(norm_act_kernel<2>, grid, block, 0, s, x, nsplit, split_stride, bias, g, scale, ss_stride,
                           pix_per_image, residual, y, rows, C, flags, res_nsplit, res_stride, res_bias);
    else if (C <= 512)
        hipLaunchKernelGGL(norm_act_kernel<8>, grid, block, 0, s, x, nsplit, split_stride, bias, g, scale, ss_stride,
                           pix_per_image, residual, y, rows, C, flags, res_nsplit, res_stride, res_bias);
    else
        hipLaunchKernelGGL(norm_act_kernel<0>, grid, block, 0, s, x, nsplit, split_stride, bias, g, scale, ss_stride,
                           pix_per_image, residual, y, rows, C, flags, res_nsplit, res_stride, res_bias);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// y[r][o] = act_out(bias[o] + sum_i act_in(x[r][i]) * W[o][i])   (nn.Linear layout, wave per output)
// time_mlp (DD/denoising_diffusion.py:280-285), ResnetBlock.mlp (:127-130), CrossAttention k/v
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int ldy, int I, int O, int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = blockIdx.y;
    if (o >= O) return;
    const float* xr = x + (size_t)r * ldx;
    const float* wr = W + (size_t)o * I;
    float acc = 0.f;
    for (int i = lane; i < I; i += 64) {
        float v = xr[i];
        if (act_in == 1) v = silu_f(v);
        else if (act_in == 2) v = gelu_erf_f(v);
        acc += v * wr[i];
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (bias) acc += bias[o];
        if (act_out == 1) acc = silu_f(acc);
        else if (act_out == 2) acc = gelu_erf_f(acc);
        y[(size_t)r * ldy + o] = acc;
    }
}
// The same for a batch of different rows (the training step: one time embedding per image): a wave forms its output o for 8
// rows r, so a weight row is read once per 8 rows instead of once per row.  Same summation order per (r, o) as above.
__global__ __launch_bounds__(256) void linear_rows8_kernel(const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int ldy, int R, int I, int O, int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = blockIdx.y * 8;
    if (o >= O) return;
    const float* wr = W + (size_t)o * I;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int i = lane; i < I; i += 64) {
        const float w = wr[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = r0 + j < R ? x[(size_t)(r0 + j) * ldx + i] : 0.f;
            if (act_in == 1) v = silu_f(v);
            else if (act_in == 2) v = gelu_erf_f(v);
            acc[j] += v * w;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = wave_sum(acc[j]);
        if (lane == 0 && r0 + j < R) {
            if (bias) a += bias[o];
            if (act_out == 1) a = silu_f(a);
            else if (act_out == 2) a = gelu_erf_f(a);
            y[(size_t)(r0 + j) * ldy + o] = a;
        }
    }
}
int launch_linear_rows(const float* x, int ldx, const float* W, const float* bias, float* y, int ldy, int R, int I,
                       int O, int act_in, int act_out, hipStream_t s) {
    if (R == 0 || O == 0) return 0;
    static const bool no_mfma = std::getenv("DM_NO_SMALL_GEMM") != nullptr;
    if (!no_mfma && act_in == 0 && act_out == 0 && rows_gemm_nt_ok(R, I, O, ldx))
        return launch_rows_gemm_nt(x, ldx, W, bias, y, ldy, R, I, O, s);  // a batch of rows: MFMA GEMM (small_gemm.hip)
    if (R >= 8) {
        hipLaunchKernelGGL(linear_rows8_kernel, dim3((O + 3) / 4, (R + 7) / 8), dim3(256), 0, s, x, ldx, W, bias, y, ldy, R, I, O,
                           act_in, act_out);
        DM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(linear_rows_kernel, dim3((O + 3) / 4, R), dim3(256), 0, s, x, ldx, W, bias, y, ldy, I, O,
                       act_in, act_out);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// SinusoidalPosEmb.forward (DD/denoising_diffusion.py:77-84); freqs are computed on the host
// exactly as the reference computes them (fp32) so only sin/cos run here.
__global__ void sinusoid_kernel(const int64_t* __restrict__ t, const int64_t* __restrict__ step_times,
                                const SamplerState* __restrict__ st, const float* __restrict__ freqs,
                                float* __restrict__ e, int R, int half, int learned) {
#pragma clang fp contract(off)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * half) return;
    int r = i / half, k = i - r * half;
    int64_t tv = step_times ? step_times[st->step] : t[r];
    if (learned) {
        // RandomOrLearnedSinusoidalPosEmb (:96-101): (x * w) * 2 pi in fp32, row = [x | sin | cos]
        const float a = ((float)tv * freqs[k]) * 6.283185307179586f;
        float* row = e + (size_t)r * (2 * half + 1);
        if (k == 0) row[0] = (float)tv;
        row[1 + k] = sinf(a);
        row[1 + half + k] = cosf(a);
        return;
    }
    float a = (float)tv * freqs[k];
    e[(size_t)r * 2 * half + k] = sinf(a);
    e[(size_t)r * 2 * half + half + k] = cosf(a);
}
int launch_sinusoid(const int64_t* t, const int64_t* step_times, const SamplerState* step, const float* freqs, float* e,
                    int R, int half, hipStream_t s, bool learned) {
    int n = R * half;
    hipLaunchKernelGGL(sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, step_times, step, freqs, e, R,
                       half, learned ? 1 : 0);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// GroupNorm(32, eps) [+ swish], NHWC (VAE decoder only; LD/modules/diffusionmodules/model.py:55-56)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                          int HW, int C, int groups, float eps) {
    const int b = blockIdx.y, grp = blockIdx.x;
    const int cg = C / groups;
    const int64_t n = (int64_t)HW * cg;
    const float* xb = x + (size_t)b * HW * C + grp * cg;
    __shared__ float red[4];
    __shared__ float mean_s;
    float sum = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        int64_t p = i / cg;
        int j = i - p * cg;
        sum += xb[p * C + j];
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) mean_s = (red[0] + red[1] + red[2] + red[3]) / (float)n;
    __syncthreads();
    const float mean = mean_s;
    float var = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        int64_t p = i / cg;
        int j = i - p * cg;
        float d = xb[p * C + j] - mean;
        var += d * d;
    }
    var = wave_sum(var);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = var;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = (red[0] + red[1] + red[2] + red[3]) / (float)n;
        stats[((size_t)b * groups + grp) * 2 + 0] = mean;
        stats[((size_t)b * groups + grp) * 2 + 1] = 1.0f / sqrtf(v + eps);
    }
}
__global__ void group_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                   const float* __restrict__ w, const float* __restrict__ bias,
                                   float* __restrict__ y, int HW, int C, int groups, int swish, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = i % C;
    int64_t b = i / ((int64_t)HW * C);
    int grp = c / (C / groups);
    const float* st = stats + (b * groups + grp) * 2;
    float v = (x[i] - st[0]) * st[1] * w[c] + bias[c];
    if (swish) v = v / (1.0f + __expf(-v));
    y[i] = v;
}
int launch_group_norm(const float* x, const float* w, const float* b, float* y, float* stats_ws, int B, int HW,
                      int C, int groups, float eps, int swish, hipStream_t s) {
    // stats_ws: group_stats_ws_floats(B, groups) = B*groups*2 floats (mean, rstd), then the per-block partial sums (doubles)
    if (group_sums_ok(C, groups)) {
        double* acc = reinterpret_cast<double*>(stats_ws + (size_t)B * groups * 2);
        if (launch_group_stats_fast(x, stats_ws, acc, B, HW, C, groups, eps, s)) return 1;
    } else {
        hipLaunchKernelGGL(group_stats_kernel, dim3(groups, B), dim3(256), 0, s, x, stats_ws, HW, C, groups, eps);
    }
    int64_t n = (int64_t)B * HW * C;
    hipLaunchKernelGGL(group_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, stats_ws, w, b, y, HW, C,
                       groups, swish, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
// dst[b][c_off + c][sp] = src[b][c][sp]: places a (B, Cs, HW) tensor into channels [c_off, c_off + Cs) of (B, Cd, HW)
__global__ void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t per_src,
                                     int64_t per_dst, int64_t off, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / per_src, r = i - b * per_src;
    dst[b * per_dst + off + r] = src[i];
}
int launch_copy_channels(const float* src, float* dst, int B, int Cs, int Cd, int c_off, int HW, hipStream_t s) {
    const int64_t n = (int64_t)B * Cs * HW;
    hipLaunchKernelGGL(copy_channels_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, (int64_t)Cs * HW,
                       (int64_t)Cd * HW, (int64_t)c_off * HW, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
// dst[r][0..n) = src[0..n) for r < rows (dst rows ld floats apart): the time embedding of a sampler step in front
// of every row of the text-concat input (DD/denoising_diffusion_text_conditional.py:146-152), one launch
// With group < rows: source row r / group serves destination row r (one vector per image spread over its pixels: the
// CrossAttention layer with a single context token, DD/denoising_diffusion_text_conditional.py:54-78).
__global__ void broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int ld, int64_t total,
                                      int group) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t r = i / n;
    const int c = (int)(i - r * n);
    dst[r * ld + c] = src[(r / group) * n + c];
}
int launch_broadcast_rows(const float* src, float* dst, int rows, int n, int ld, hipStream_t s, int group) {
    const int64_t total = (int64_t)rows * n;
    if (total == 0) return 0;
    if (group <= 0) group = rows;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, dst, n, ld, total, group);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
// Holds the stream for `ticks` of the 100 MHz wall clock.  Used only by the profiling leg: with the GPU parked the
// host can enqueue a whole run of (event, kernel, event) triples, which then execute back to back, so the event
// intervals measure the kernels and not the host's launch rate.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int launch_spin(double milliseconds, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, s, (long long)(milliseconds * 1.0e5));
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_add(const float* a, const float* b, float* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, y, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// 1x1 convolution to a few output channels, NHWC in -> NCHW out (final_conv, DD/denoising_diffusion.py:319: 64 -> 3).
// One pixel per thread: the MFMA tile would spend 61 of 64 columns on padding; this is a 256-byte read and three
// coalesced dword stores per pixel, HBM bound.  Weights (Cout x C) and bias sit in LDS.
// ---------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void pointwise_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int64_t pixels, int C, int HW) {
    extern __shared__ __attribute__((aligned(16))) float ws[];  // [COUT][C]
    for (int i = threadIdx.x; i < COUT * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= pixels) return;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.f;
    const f32x4v* xp = reinterpret_cast<const f32x4v*>(x + pix * C);
    for (int q = 0; q < C / 4; ++q) {
        const f32x4v v = xp[q];
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const f32x4v wv = *reinterpret_cast<const f32x4v*>(ws + o * C + 4 * q);  // same address in every lane
            acc[o] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        }
    }
    const int64_t b = pix / HW, sp = pix - b * HW;
#pragma unroll
    for (int o = 0; o < COUT; ++o) y[(b * COUT + o) * HW + sp] = acc[o];
}
int launch_pointwise_small(const float* x, const float* w_oc, const float* bias, float* y_nchw, int64_t pixels, int C,
                           int Cout, int HW, hipStream_t s) {
    DM_REQUIRE(Cout >= 1 && Cout <= 4 && C % 4 == 0, "pointwise_small: 1..4 output channels, C % 4 == 0");
    const dim3 grid((pixels + 255) / 256), block(256);
    const size_t lds = (size_t)Cout * C * sizeof(float);
    switch (Cout) {
        case 1: hipLaunchKernelGGL(pointwise_small_kernel<1>, grid, block, lds, s, x, w_oc, bias, y_nchw, pixels, C, HW); break;
        case 2: hipLaunchKernelGGL(pointwise_small_kernel<2>, grid, block, lds, s, x, w_oc, bias, y_nchw, pixels, C, HW); break;
        case 3: hipLaunchKernelGGL(pointwise_small_kernel<3>, grid, block, lds, s, x, w_oc, bias, y_nchw, pixels, C, HW); break;
        default: hipLaunchKernelGGL(pointwise_small_kernel<4>, grid, block, lds, s, x, w_oc, bias, y_nchw, pixels, C, HW); break;
    }
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t draw, uint64_t idx4, float z[4]) {
    uint32_t c[4] = {(uint32_t)idx4, (uint32_t)(idx4 >> 32), (uint32_t)draw, (uint32_t)(draw >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float inv = 2.3283064365386963e-10f;  // 2^-32
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float u1 = ((float)c[2 * h] + 1.0f) * inv;  // (0, 1]
        float u2 = (float)c[2 * h + 1] * inv;
        float rad = sqrtf(-2.0f * logf(fminf(u1, 1.0f)));
        float ang = 6.283185307179586f * u2;
        z[2 * h] = rad * cosf(ang);
        z[2 * h + 1] = rad * sinf(ang);
    }
}
__global__ void randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t draw, uint64_t off4) {
    int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= n) return;
    float z[4];
    philox_normal4(seed, draw, off4 + (uint64_t)i4, z);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i4 * 4 + j < n) out[i4 * 4 + j] = z[j];
}
int launch_randn(float* out, int64_t n, uint64_t seed, uint64_t draw, uint64_t element_offset, hipStream_t s) {
    DM_REQUIRE(element_offset % 4 == 0, "Philox element offset must be a multiple of 4 (one counter serves 4 elements)");
    if (n == 0) return 0;
    int64_t n4 = (n + 3) / 4;
    hipLaunchKernelGGL(randn_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, out, n, seed, draw, element_offset / 4);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// One sampler update (elementwise, layout-agnostic).  Restates, for objective pred_noise:
//   DDPM  DD/denoising_diffusion.py:570-574 (x0), :633 (clamp), :594-598 (posterior mean), :643-644
//   DDIM  DD/denoising_diffusion.py:607-613 (x0, clamp, re-derived eps), :686-701
// Contraction is off so that the expression tree rounds exactly like the reference's tensor ops.
// ---------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// objective (what the U-Net output `eps` means, :603-626): 0 pred_noise, 1 pred_x0, 2 pred_v.  x_start (clamped, the
// value both loops hand to the next step as x_self_cond, :657 / :683) is also written to xstart_out when given.
__global__ void sampler_update_kernel(int kind, int objective, const float* __restrict__ x,
                                      const float* __restrict__ eps, const float* __restrict__ noise,
                                      const float* __restrict__ coefs, const SamplerState* __restrict__ st,
                                      int64_t noise_step_stride, float* __restrict__ out,
                                      float* __restrict__ all_steps, float* __restrict__ final_out,
                                      float* __restrict__ xstart_out, int64_t n) {
    // everything that changes between two sample() calls of one shape is read from the device-side state, so a
    // captured step graph stays valid across calls (seed, Philox offset, step count, unnormalize)
    const int step = st ? st->step : 0;
    const int n_steps = st ? st->n_steps : 1;
    const int unnormalize = st ? st->unnormalize : 0;
    const uint64_t seed = st ? st->seed : 0;
    const uint64_t off4 = st ? st->off4 : 0;
    const float* c = coefs + (size_t)step * 8;
    const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c6 = c[6], c7 = c[7];
    const bool flag = c[5] != 0.0f;
    int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= n) return;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (flag) {
        if (noise) {
            const float* np = noise + (size_t)step * noise_step_stride;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i4 * 4 + j < n) z[j] = np[i4 * 4 + j];
        } else if (c4 != 0.0f) {
            philox_normal4(seed, (uint64_t)step + 1, off4 + (uint64_t)i4, z);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int64_t i = i4 * 4 + j;
        if (i >= n) break;
        float xv = x[i], ev = eps[i];
        float x0;
        if (objective == 0) x0 = c0 * xv - c1 * ev;       // predict_start_from_noise :570-574
        else if (objective == 1) x0 = ev;                  // the model predicts x_0 :614-617
        else x0 = c6 * xv - c7 * ev;                       // predict_start_from_v :588-592
        x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        if (xstart_out) xstart_out[i] = x0;
        float r;
        if (kind == 0) {
            float mean = c2 * x0 + c3 * xv;
            r = flag ? mean + c4 * z[j] : mean + c4 * 0.0f;
        } else {
            float e2 = (c0 * xv - x0) / c1;
            r = flag ? (x0 * c2 + c3 * e2) + c4 * z[j] : x0;
        }
        out[i] = r;
        if (all_steps) all_steps[(size_t)(step + 1) * n + i] = r;
        if (final_out && step == n_steps - 1) final_out[i] = unnormalize ? (r + 1.0f) * 0.5f : r;
    }
}
// out = x or (x + 1) / 2: `unnormalize` at the end of a loop (DD/denoising_diffusion.py:663,:707)
__global__ void finalize_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, int unnormalize) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = x[i];
    out[i] = unnormalize ? (r + 1.0f) * 0.5f : r;
}
#pragma clang fp contract(fast)

int launch_finalize(const float* x, float* out, int64_t n, int unnormalize, hipStream_t s) {
    hipLaunchKernelGGL(finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, out, n, unnormalize);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_sampler_update(int kind, const float* x, const float* eps, const float* noise, const float* coefs_dev,
                          const SamplerState* state_dev, int64_t noise_step_stride, float* out, float* all_steps,
                          float* final_out, int64_t n, hipStream_t s, int objective, float* xstart_out) {
    int64_t n4 = (n + 3) / 4;
    hipLaunchKernelGGL(sampler_update_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, kind, objective, x, eps, noise,
                       coefs_dev, state_dev, noise_step_stride, out, all_steps, final_out, xstart_out, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ void step_advance_kernel(SamplerState* st) { st->step += 1; }
int launch_step_advance(SamplerState* state_dev, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, state_dev);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dm

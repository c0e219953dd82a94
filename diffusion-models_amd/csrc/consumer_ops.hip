// Bandwidth-bound operators of the sample CONSUMER next to the sampling path: the InceptionV3
// feature extractor of the FID / Inception-score evaluators (DD/fid_evaluation.py:41-51, DD/inception_score_evaluation.py:
// 70-92) around the convolution kernels.  NHWC fp32; gfx950 only.
#include "dm_common.h"

#include <cmath>

namespace dm {

// k x k pooling, NHWC.  mode 0: max (padding never wins); 1: average over k*k (count_include_pad = True, the
// F.avg_pool2d default of torchvision's InceptionA/C/E); 2: average over the valid pixels (count_include_pad = False,
// pytorch_fid's FIDInceptionA/C/E_1)
__global__ void pool2d_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int Ho, int Wo,
                                   int k, int stride, int pad, int mode, int64_t n4) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4n = C / 4;
    const int c4 = (int)(i % c4n);
    int64_t r = i / c4n;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int64_t b = r / Ho;
    f4 acc = mode == 0 ? f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f4{0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    for (int dy = 0; dy < k; ++dy) {
        const int y = yo * stride - pad + dy;
        if (y < 0 || y >= H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int x = xo * stride - pad + dx;
            if (x < 0 || x >= W) continue;
            const f4 v = *reinterpret_cast<const f4*>(in + ((b * H + y) * W + x) * C + 4 * c4);
            if (mode == 0) {
                acc.x = fmaxf(acc.x, v.x);
                acc.y = fmaxf(acc.y, v.y);
                acc.z = fmaxf(acc.z, v.z);
                acc.w = fmaxf(acc.w, v.w);
            } else {
                acc += v;
            }
            ++cnt;
        }
    }
    if (mode == 1) acc = acc / (float)(k * k);
    if (mode == 2) acc = acc / (float)max(cnt, 1);
    *reinterpret_cast<f4*>(out + ((b * Ho + yo) * Wo + xo) * C + 4 * c4) = acc;
}
int launch_pool2d_nhwc(const float* in, float* out, int B, int H, int W, int C, int k, int stride, int pad, int mode,
                       hipStream_t s) {
    DM_REQUIRE(C % 4 == 0 && k >= 1 && stride >= 1 && pad >= 0 && mode >= 0 && mode <= 2, "pool2d: C % 4 == 0, valid window");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    DM_REQUIRE(Ho > 0 && Wo > 0, "pool2d: empty output");
    const int64_t n4 = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(pool2d_nhwc_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, in, out, H, W, C, Ho, Wo, k, stride, pad,
                       mode, n4);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// F.interpolate(x, size, mode="bilinear", align_corners=False) on an NCHW image batch, written NHWC, followed by
// y = scale[c] * v + shift[c]  (2x - 1 of pytorch_fid; ImageNet mean / std and torchvision's transform_input)
__global__ void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W, int Ho,
                                       int Wo, float sy, float sx, const float* __restrict__ scale,
                                       const float* __restrict__ shift, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int64_t b = r / Ho;
    // source index as ATen computes it: max(0, (dst + 0.5) * scale - 0.5)
    const float fy = fmaxf((yo + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((xo + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - y0, lx = fx - x0;
    const float* p = in + (b * C + c) * (int64_t)H * W;
    const float v = (1.f - ly) * ((1.f - lx) * p[y0 * W + x0] + lx * p[y0 * W + x1]) +
                    ly * ((1.f - lx) * p[y1 * W + x0] + lx * p[y1 * W + x1]);
    out[i] = scale[c] * v + shift[c];
}
int launch_resize_bilinear(const float* in_nchw, float* out_nhwc, int B, int C, int H, int W, int Ho, int Wo,
                           const float* scale, const float* shift, hipStream_t s) {
    const int64_t n = (int64_t)B * Ho * Wo * C;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in_nchw, out_nhwc, C, H, W, Ho, Wo,
                       (float)H / (float)Ho, (float)W / (float)Wo, scale, shift, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// dst[row][c_off + c] = src[row][c]: one branch of an Inception block into its slice of the concatenated tensor
__global__ void copy_channels_nhwc_kernel(const float* __restrict__ src, int Cs, float* __restrict__ dst, int Cd, int c_off,
                                          int64_t n4) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4n = Cs / 4;
    const int64_t row = i / c4n;
    const int c4 = (int)(i - row * c4n);
    *reinterpret_cast<f4*>(dst + row * Cd + c_off + 4 * c4) = *reinterpret_cast<const f4*>(src + row * Cs + 4 * c4);
}
int launch_copy_channels_nhwc(const float* src, int Cs, float* dst, int Cd, int c_off, int64_t rows, hipStream_t s) {
    DM_REQUIRE(Cs % 4 == 0 && Cd % 4 == 0 && c_off % 4 == 0 && c_off + Cs <= Cd, "copy_channels_nhwc: 4-channel granules");
    const int64_t n4 = rows * (Cs / 4);
    if (n4 == 0) return 0;
    hipLaunchKernelGGL(copy_channels_nhwc_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, src, Cs, dst, Cd, c_off, n4);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// adaptive_avg_pool2d(x, (1, 1)): out[b][c] = mean over the HW pixels, NHWC in
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int HW,
                                                             int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* p = in + (size_t)b * HW * C + c;
    float acc = 0.f;
    for (int i = 0; i < HW; ++i) acc += p[(size_t)i * C];
    out[(size_t)b * C + c] = acc / (float)HW;
}
int launch_global_avgpool_nhwc(const float* in, float* out, int B, int HW, int C, hipStream_t s) {
    hipLaunchKernelGGL(global_avgpool_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, in, out, HW, C);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dm

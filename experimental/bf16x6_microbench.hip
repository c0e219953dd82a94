// Development microbenchmark for the "fp32 products on the bf16 matrix cores" idea (DESIGN.md, next lever):
// accuracy of a GEMM whose operands are split into three round-to-nearest bf16 terms (x = h + m + l) and whose
// products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, against the f32-input MFMA and an fp64 reference,
// and the issue rate of that scheme next to the splitting VALU work.
//   hipcc -O3 --offload-arch=gfx950 tools/bf16x6_microbench.hip -o /tmp/bf16x6_mb && /tmp/bf16x6_mb
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3 {
    bf16x8 h, m, l;
};
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;  // exact
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;  // exact
        s.h[i] = h;
        s.m[i] = m;
        s.l[i] = (__bf16)r2;
    }
    return s;
}

// C (M x N) = A (M x K, row major) * B (K x N, stored transposed: Bt is N x K row major); one wave per 32x32 tile.
// MODE 0: f32 MFMA; 3 / 6 / 9: number of bf16 products per fp32 product.
template <int MODE>
__global__ __launch_bounds__(64) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                  float* __restrict__ C, int M, int N, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, hf = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if constexpr (MODE == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(m0 + l31) * K + k + hf], Bt[(size_t)(n0 + l31) * K + k + hf],
                                                       acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            float a[8], b[8];
            for (int j = 0; j < 8; ++j) {
                a[j] = A[(size_t)(m0 + l31) * K + k + 8 * hf + j];
                b[j] = Bt[(size_t)(n0 + l31) * K + k + 8 * hf + j];
            }
            const Split3 sa = split3(a), sb = split3(b);
            // small terms first
            if constexpr (MODE >= 9) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.l, sb.l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.m, sb.l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.l, sb.m, acc, 0, 0, 0);
            }
            if constexpr (MODE >= 6) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.h, sb.l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.l, sb.h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.m, sb.m, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.h, sb.m, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.m, sb.h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.h, sb.h, acc, 0, 0, 0);
        }
    }
    for (int e = 0; e < 16; ++e) C[(size_t)(m0 + (e & 3) + 8 * (e >> 2) + 4 * hf) * N + n0 + l31] = acc[e];
}

// Issue-rate kernel: a wave owns a 32 x 128 tile (4 column tiles).  Per 16-deep K step it reads 8 fp32 A values per
// lane from LDS, splits them (VALU) and multiplies with four pre-split B fragments read from LDS: 24 MFMAs (bf16x6)
// per split.  MODE 0 is the f32 MFMA doing the same tile (32 MFMAs of 32x32x2 per 16-deep step).
template <int MODE>
__global__ __launch_bounds__(256, 2) void rate_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 12288; i += 256) sm[i] = (float)((i * 37) & 255) * 0.01f - 1.0f;
    __syncthreads();
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const float* pa = sm + (lane & 31) * 36 + (lane >> 5) * 8;
    const bf16x8* pb = reinterpret_cast<const bf16x8*>(sm + 4096) + lane;
    for (int it = 0; it < iters; ++it) {
        const int off = (it & 1) * 16;
        const f32x4 a0 = *(const f32x4*)(pa + off), a1 = *(const f32x4*)(pa + off + 4);
        if constexpr (MODE == 0) {
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], sm[4096 + q * 64 + lane + s], acc[q], 0, 0, 0);
        } else {
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const Split3 sa = split3(av);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8 bh = pb[(q * 3 + 0) * 64], bm = pb[(q * 3 + 1) * 64], bl = pb[(q * 3 + 2) * 64];
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.h, bl, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.l, bh, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.m, bm, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.h, bm, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.m, bh, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.h, bh, acc[q], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 16; ++e) s += acc[q][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void accuracy(const char* name, const std::vector<float>& A, const std::vector<float>& Bt,
                     const std::vector<double>& ref, int M, int N, int K) {
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4);
    hipMalloc(&dB, Bt.size() * 4);
    hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gemm_kernel<MODE>, dim3(N / 32, M / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K);
    std::vector<float> C((size_t)M * N);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0, mx = 0;
    for (size_t i = 0; i < C.size(); ++i) {
        const double d = C[i] - ref[i];
        num += d * d;
        den += ref[i] * ref[i];
        mx = std::fmax(mx, std::fabs(d));
    }
    printf("  %-34s rel-L2 %.3e   max abs err %.3e\n", name, std::sqrt(num / den), mx);
    hipFree(dA);
    hipFree(dB);
    hipFree(dC);
}

template <int MODE>
static void rate(const char* name, int blocks) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000, lds = 64 * 1024;
    hipFuncSetAttribute((const void*)rate_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), lds, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * (2.0 * 32 * 128 * 16);  // fp32-equivalent FLOPs
    printf("  %-46s %.3f ms   %.1f fp32-equivalent TFLOP/s\n", name, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    const int M = 64, N = 64, K = 4608;  // K of the deepest 3x3 layer
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<float> A((size_t)M * K), Bt((size_t)N * K);
        for (auto& v : A) v = nd(rng) * (pass ? std::exp(4.f * nd(rng)) : 1.f);  // pass 1: wide dynamic range
        for (auto& v : Bt) v = nd(rng) * (pass ? std::exp(4.f * nd(rng)) : 1.f) / std::sqrt((float)K);
        std::vector<double> ref((size_t)M * N);
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)A[(size_t)i * K + k] * (double)Bt[(size_t)j * K + k];
                ref[(size_t)i * N + j] = s;
            }
        printf("accuracy vs fp64, %d x %d x %d, %s:\n", M, N, K, pass ? "log-normal magnitudes" : "N(0,1) operands");
        accuracy<0>("f32 MFMA (32x32x2_f32)", A, Bt, ref, M, N, K);
        accuracy<3>("bf16 x3 (hh, hm, mh)", A, Bt, ref, M, N, K);
        accuracy<6>("bf16 x6 (+ hl, lh, mm)", A, Bt, ref, M, N, K);
        accuracy<9>("bf16 x9 (all)", A, Bt, ref, M, N, K);
    }
    printf("issue rate, wave tile 32 x 128, operands in LDS:\n");
    rate<0>("f32 MFMA, 2 waves/SIMD", 512);
    rate<6>("bf16 x6, A split in the loop, 2 waves/SIMD", 512);
    rate<6>("bf16 x6, A split in the loop, 1 wave/SIMD", 256);
    return 0;
}

// Development microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 under the access pattern of conv_mfma_kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_microbench.hip -o /tmp/mfma_mb && /tmp/mfma_mb
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) sm[i] = (float)(i & 15) * 0.001f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int r = 0; r < 2; ++r) for (int q = 0; q < 2; ++q) for (int e = 0; e < 16; ++e) acc[r][q][e] = 0.f;
    const float* pa = sm + (lane & 31) * 20 + (lane >> 5) * 8;
    const float* pb = sm + 8192 + (lane & 31) * 20 + (lane >> 5) * 8;
    float4 a0 = *(const float4*)pa, a1 = *(const float4*)(pa + 640), b0 = *(const float4*)pb, b1 = *(const float4*)(pb + 640);
    for (int it = 0; it < iters; ++it) {
        float4 na0, na1, nb0, nb1;
        if (MODE >= 1) {  // LDS fragment reads for the next block, issued before the MFMAs
            int off = (it & 7) * 4;
            na0 = *(const float4*)(pa + off); na1 = *(const float4*)(pa + 640 + off);
            nb0 = *(const float4*)(pb + off); nb1 = *(const float4*)(pb + 640 + off);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float fa[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
        const float fb[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[r][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[r][s], fb[q][s], acc[r][q], 0, 0, 0);
        if (MODE >= 1) { __builtin_amdgcn_sched_barrier(0); a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        if (MODE >= 2 && (it % 6) == 5) __syncthreads();  // one barrier per 96 MFMAs
    }
    float s = 0.f;
    for (int r = 0; r < 2; ++r) for (int q = 0; q < 2; ++q) for (int e = 0; e < 16; ++e) s += acc[r][q][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int lds) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
    printf("%-40s blocks=%4d lds=%6d  %.3f ms  %.1f TF/s\n", name, blocks, lds, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<0>("mfma only, 1 WG/CU (1 wave/SIMD)", 256, 100 * 1024);
    run<0>("mfma only, 2 WG/CU (2 waves/SIMD)", 512, 70 * 1024);
    run<1>("+ds_read_b128 x4 per 16 mfma, 1 WG/CU", 256, 100 * 1024);
    run<1>("+ds_read_b128 x4 per 16 mfma, 2 WG/CU", 512, 70 * 1024);
    run<2>("+barrier per 96 mfma, 1 WG/CU", 256, 100 * 1024);
    run<2>("+barrier per 96 mfma, 2 WG/CU", 512, 70 * 1024);
    run<2>("+barrier per 96 mfma, 4 WG/CU worth (2 rounds)", 1024, 70 * 1024);
    return 0;
}

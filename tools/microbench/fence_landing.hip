// Microbenchmark: does a "last workgroup to arrive lands the tile" epilogue pay on MI355X?
//
// Producer workgroups (256 threads) spin on ALU work for ~T cycles, then write a 64 KB tile (256 rows x 64 floats) of
// partial sums.  GROUP of them share one output tile.
//   variant 0: producers only, then a separate landing kernel sums the GROUP partial tiles (what the library does today)
//   variant 1: each producer does __threadfence() + one device-scope atomicAdd on its group's counter; the workgroup that
//              observes GROUP - 1 re-reads all GROUP tiles (written by workgroups on other XCDs) and writes the sum.
// Prints the time of both and checks that variant 1 produced exactly the sums of variant 0 (visibility across XCD L2s).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/fence_landing.hip -o gpurun_out/fence_landing && gpurun_out/fence_landing
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int TILE = 256 * 64;  // floats per tile

__device__ float spin_work(int iters, float seed) {
    float a = seed, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        a = a * b + 0.5f;
        b = b * 0.99999f + 1e-6f;
    }
    return a + b;
}

template <int FUSED>
__global__ __launch_bounds__(256) void producer(float* part, float* out, unsigned* counters, int group, int iters) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    // groups are interleaved over the grid the way cout tiles / K splits are: member m of group g is block g * group + m
    const int g = wg / group, m = wg % group;
    const float w = spin_work(iters, (float)(tid & 7));
    float4* dst = reinterpret_cast<float4*>(part + ((size_t)g * group + m) * TILE);
    for (int i = tid; i < TILE / 4; i += 256) {
        const float v = (float)((wg * 131 + i) % 1021) * 0.001f + (w > 1e30f ? w : 0.f);
        dst[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
    }
    if (!FUSED) return;
    __shared__ unsigned last;
    __threadfence();  // this workgroup's stores become visible device-wide before the counter moves
    __syncthreads();
    if (tid == 0) last = atomicAdd(&counters[g], 1u);
    __syncthreads();
    if (last != (unsigned)(group - 1)) return;
    __threadfence();  // acquire side: do not read the other tiles from stale lines
    float4* o = reinterpret_cast<float4*>(out + (size_t)g * TILE);
    const float4* src = reinterpret_cast<const float4*>(part + (size_t)g * group * TILE);
    for (int i = tid; i < TILE / 4; i += 256) {
        float4 s = src[i];
        for (int k = 1; k < group; ++k) {
            const float4 t = src[(size_t)k * (TILE / 4) + i];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        o[i] = s;
    }
    if (tid == 0) counters[g] = 0;  // ready for the next launch
}

__global__ __launch_bounds__(256) void landing(const float* part, float* out, int group, int n_tiles) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // float4 index over all output tiles
    if (i >= (size_t)n_tiles * (TILE / 4)) return;
    const size_t g = i / (TILE / 4), r = i % (TILE / 4);
    const float4* src = reinterpret_cast<const float4*>(part + g * group * TILE);
    float4 s = src[r];
    for (int k = 1; k < group; ++k) {
        const float4 t = src[(size_t)k * (TILE / 4) + r];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    reinterpret_cast<float4*>(out)[i] = s;
}

int main() {
    const int groups_list[] = {2, 4, 8, 16};
    const int wgs_list[] = {256, 512, 1024};
    const int iters_list[] = {2000, 20000};  // ~ 4 us and ~ 40 us of ALU work per workgroup
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int iters : iters_list)
        for (int wgs : wgs_list)
            for (int group : groups_list) {
                const int n_tiles = wgs / group;
                float *part, *out0, *out1;
                unsigned* cnt;
                CHECK(hipMalloc(&part, (size_t)wgs * TILE * 4));
                CHECK(hipMalloc(&out0, (size_t)n_tiles * TILE * 4));
                CHECK(hipMalloc(&out1, (size_t)n_tiles * TILE * 4));
                CHECK(hipMalloc(&cnt, n_tiles * sizeof(unsigned)));
                CHECK(hipMemset(cnt, 0, n_tiles * sizeof(unsigned)));
                float ms[2] = {0, 0};
                for (int variant = 0; variant < 2; ++variant) {
                    const int reps = 20;
                    for (int r = -3; r < reps; ++r) {
                        if (r == 0) CHECK(hipEventRecord(e0, s));
                        if (variant == 0) {
                            hipLaunchKernelGGL(producer<0>, dim3(wgs), dim3(256), 0, s, part, out0, cnt, group, iters);
                            hipLaunchKernelGGL(landing, dim3((n_tiles * (TILE / 4) + 255) / 256), dim3(256), 0, s, part, out0,
                                               group, n_tiles);
                        } else {
                            hipLaunchKernelGGL(producer<1>, dim3(wgs), dim3(256), 0, s, part, out1, cnt, group, iters);
                        }
                    }
                    CHECK(hipEventRecord(e1, s));
                    CHECK(hipStreamSynchronize(s));
                    CHECK(hipEventElapsedTime(&ms[variant], e0, e1));
                    ms[variant] /= reps;
                }
                std::vector<float> h0((size_t)n_tiles * TILE), h1((size_t)n_tiles * TILE);
                CHECK(hipMemcpy(h0.data(), out0, h0.size() * 4, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(h1.data(), out1, h1.size() * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
                printf("iters %6d wgs %5d group %2d: two kernels %8.2f us   fused last-arriver %8.2f us   mismatches %zu\n",
                       iters, wgs, group, ms[0] * 1e3, ms[1] * 1e3, bad);
                CHECK(hipFree(part)); CHECK(hipFree(out0)); CHECK(hipFree(out1)); CHECK(hipFree(cnt));
            }
    return 0;
}

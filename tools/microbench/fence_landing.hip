// Microbenchmark: does a "last workgroup to arrive lands the tile" epilogue pay on MI355X?
//
// Producer workgroups (256 threads) spin on ALU work for ~T cycles, then write a 64 KB tile (256 rows x 64 floats) of
// partial sums.  GROUP of them share one output tile.
//   variant 0: producers only, then a separate landing kernel sums the GROUP partial tiles (what the library does today)
//   variant 1: each producer does __threadfence() + one device-scope atomicAdd on its group's counter; the workgroup that
//              observes GROUP - 1 re-reads all GROUP tiles (written by workgroups on other XCDs) and writes the sum.
//   variant 2: producers + __threadfence() only (what the fence alone costs)
//   variant 3: counter + landing WITHOUT fences (what the rest costs; the sums may be wrong)
//   variant 4: tiles written with agent-scope (write-through) stores and read back with agent-scope loads, relaxed counter,
//              no fence
// Prints the times and, for the landing variants, how many sums differ from variant 0 (visibility across XCD L2s).
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
__global__ __launch_bounds__(256) void producer(float* part, float* out, unsigned* counters, int group, int iters,
                                                 int rep) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    // groups are interleaved over the grid the way cout tiles / K splits are: member m of group g is block g * group + m
    const int g = wg / group, m = wg % group;
    const float w = spin_work(iters, (float)(tid & 7));
    float* dstf = part + ((size_t)g * group + m) * TILE;
    float4* dst = reinterpret_cast<float4*>(dstf);
    for (int i = tid; i < TILE / 4; i += 256) {
        // the values change with every launch: a stale line from the previous launch would show in the sums
        const float v = (float)((wg * 131 + i + 7 * rep) % 1021) * 0.001f + (w > 1e30f ? w : 0.f);
        if (FUSED == 4) {  // write-through stores at agent scope (sc1): no L2 write-back needed later
            __hip_atomic_store(dstf + 4 * i + 0, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dstf + 4 * i + 1, v + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dstf + 4 * i + 2, v + 2.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dstf + 4 * i + 3, v + 3.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            dst[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
        }
    }
    if (FUSED == 0) return;
    __shared__ unsigned last;
    if (FUSED == 1 || FUSED == 2) __threadfence();  // this workgroup's stores become visible device-wide before the counter moves
    if (FUSED == 2) return;
    if (FUSED == 4) __builtin_amdgcn_s_waitcnt(0);  // the write-through stores have been acknowledged
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(&counters[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (last != (unsigned)(group - 1)) return;
    if (FUSED == 1) __threadfence();  // acquire side: do not read the other tiles from stale lines
    float4* o = reinterpret_cast<float4*>(out + (size_t)g * TILE);
    const float* srcf = part + (size_t)g * group * TILE;
    const float4* src = reinterpret_cast<const float4*>(srcf);
    for (int i = tid; i < TILE / 4; i += 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < group; ++k) {
            float4 t;
            if (FUSED == 4) {  // agent-scope loads: not served from this XCD's non-coherent L2 lines
                const float* q = srcf + (size_t)k * TILE + 4 * i;
                t.x = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t.z = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t.w = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                t = src[(size_t)k * (TILE / 4) + i];
            }
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
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < group; ++k) {
        const float4 t = src[(size_t)k * (TILE / 4) + r];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    reinterpret_cast<float4*>(out)[i] = s;
}

int main() {
    const int groups_list[] = {2, 8};
    const int wgs_list[] = {256, 512, 1024};
    const int iters_list[] = {2000};
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
                float ms[5] = {0, 0, 0, 0, 0};
                size_t bad[5] = {0, 0, 0, 0, 0};
                std::vector<float> h0((size_t)n_tiles * TILE), h1((size_t)n_tiles * TILE);
                for (int variant = 0; variant < 5; ++variant) {
                    const int reps = 20;
                    for (int r = -3; r < reps; ++r) {
                        if (r == 0) CHECK(hipEventRecord(e0, s));
                        if (variant == 0) {
                            hipLaunchKernelGGL(producer<0>, dim3(wgs), dim3(256), 0, s, part, out0, cnt, group, iters, r);
                            hipLaunchKernelGGL(landing, dim3((n_tiles * (TILE / 4) + 255) / 256), dim3(256), 0, s, part, out0,
                                               group, n_tiles);
                        } else if (variant == 1) {
                            hipLaunchKernelGGL(producer<1>, dim3(wgs), dim3(256), 0, s, part, out1, cnt, group, iters, r);
                        } else if (variant == 2) {
                            hipLaunchKernelGGL(producer<2>, dim3(wgs), dim3(256), 0, s, part, out1, cnt, group, iters, r);
                        } else if (variant == 3) {
                            hipLaunchKernelGGL(producer<3>, dim3(wgs), dim3(256), 0, s, part, out1, cnt, group, iters, r);
                        } else {
                            hipLaunchKernelGGL(producer<4>, dim3(wgs), dim3(256), 0, s, part, out1, cnt, group, iters, r);
                        }
                    }
                    CHECK(hipEventRecord(e1, s));
                    CHECK(hipStreamSynchronize(s));
                    CHECK(hipEventElapsedTime(&ms[variant], e0, e1));
                    ms[variant] /= reps;
                    if (variant == 0) CHECK(hipMemcpy(h0.data(), out0, h0.size() * 4, hipMemcpyDeviceToHost));
                    if (variant == 1 || variant >= 3) {
                        CHECK(hipMemcpy(h1.data(), out1, h1.size() * 4, hipMemcpyDeviceToHost));
                        for (size_t i = 0; i < h0.size(); ++i) bad[variant] += h0[i] != h1[i];
                        CHECK(hipMemset(out1, 0, h1.size() * 4));
                    }
                }
                printf("iters %6d wgs %5d group %2d: two kernels %7.1f | fence+count+land %7.1f (bad %zu) | fence only %7.1f | "
                       "count+land, no fence %7.1f (bad %zu) | sc1 stores/loads, no fence %7.1f (bad %zu)  [us]\n",
                       iters, wgs, group, ms[0] * 1e3, ms[1] * 1e3, bad[1], ms[2] * 1e3, ms[3] * 1e3, bad[3], ms[4] * 1e3,
                       bad[4]);
                CHECK(hipFree(part)); CHECK(hipFree(out0)); CHECK(hipFree(out1)); CHECK(hipFree(cnt));
            }
    return 0;
}

// Kernels of the VAE decoder / encoder (latent-diffusion/ldm/modules/diffusionmodules/model.py) that are not
// convolutions: the single-head n x n attention of AttnBlock (:195-219) on the f32 MFMA, and GroupNorm(32, eps 1e-6)
// statistics (:55-56).  gfx950 only.
#include "conv_device.h"

#include <algorithm>
#include <cmath>

namespace dm {

// ---------------------------------------------------------------------------------------
// AttnBlock core: out[b][i][:] = softmax_j(q[b][i] . k[b][j] * C^-1/2) v[b][j][:]      q, k, v, out: (B, n, C) rows
//
// Flash form, TRANSPOSED so that no operand has to change lanes between the two products.  One wavefront owns 32
// queries; for every block of 32 keys
//     S^T (keys x queries) = K_blk Q^T     A = K rows from LDS (lane = key),  B = Q of the wave, kept in registers
//     O^T (C x queries)   += V_blk^T P^T   A = V^T from LDS (lane = channel), B = P^T = exp(S^T - m)
// The accumulator of the first product has lane = query, register e = key (e&3) + 8(e>>2) + 4*half; MFMA step e of the
// second product reduces over exactly that key pair, so register e IS its B operand.  The softmax statistics of a query
// are a reduction over the 16 registers of a lane and one exchange between the two lane halves; the running rescale
// of O^T is a per-lane scalar.  4 waves = 128 queries share the K / V blocks through LDS.
// ---------------------------------------------------------------------------------------
template <int CB>  // C = 32 * CB channels
__global__ __launch_bounds__(256) void vae_attn_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ out, int n,
                                                            float scale_log2e) {
    constexpr int C = 32 * CB;
    constexpr int KS = C + 4;  // padded row stride of the staged K / V blocks (floats)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ks = sm;              // [32][KS]
    float* Vs = sm + 32 * KS;    // [32][KS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool active = q0 < n;  // n is a multiple of 32 (launcher)
    const float* qb = q + (size_t)b * n * C;
    const float* kb = k + (size_t)b * n * C;
    const float* vb = v + (size_t)b * n * C;

    // Q of this lane's query, in MFMA step order: chunk t of 8 channels, step s <- channel 8t + 4 lh + s (times scale log2 e)
    float Qr[C / 2];
    if (active) {
        const float* qr = qb + (size_t)(q0 + l31) * C + 4 * lh;
#pragma unroll
        for (int t = 0; t < C / 8; ++t) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(qr + 8 * t);
            Qr[4 * t + 0] = x.x * scale_log2e;
            Qr[4 * t + 1] = x.y * scale_log2e;
            Qr[4 * t + 2] = x.z * scale_log2e;
            Qr[4 * t + 3] = x.w * scale_log2e;
        }
    } else {
#pragma unroll
        for (int t = 0; t < C / 2; ++t) Qr[t] = 0.f;
    }
    f32x16 O[CB];
#pragma unroll
    for (int db = 0; db < CB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[db][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging: the block of 32 keys is 32 * C floats of K and of V = C/4 * 32 float4 each; thread t moves items t, t+256, ...
    constexpr int ITEMS = 32 * C / 4 / 256;  // float4 per thread per tensor (CB = 2 -> 2, 4 -> 4, 8 -> 8)
    f32x4 kreg[ITEMS], vreg[ITEMS];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + 256 * i;
            const int row = it / (C / 4), c4 = it % (C / 4);
            kreg[i] = *reinterpret_cast<const f32x4*>(kb + (size_t)(j0 + row) * C + 4 * c4);
            vreg[i] = *reinterpret_cast<const f32x4*>(vb + (size_t)(j0 + row) * C + 4 * c4);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + 256 * i;
            const int row = it / (C / 4), c4 = it % (C / 4);
            *reinterpret_cast<f32x4*>(Ks + row * KS + 4 * c4) = kreg[i];
            *reinterpret_cast<f32x4*>(Vs + row * KS + 4 * c4) = vreg[i];
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < n; j0 += 32) {
        __syncthreads();  // the previous block has been consumed
        stash();
        __syncthreads();
        if (j0 + 32 < n) fetch(j0 + 32);  // in flight during the products
        // ---- S^T = K_blk Q^T
        f32x16 St;
#pragma unroll
        for (int e = 0; e < 16; ++e) St[e] = 0.f;
        const float* krow = Ks + l31 * KS + 4 * lh;
#pragma unroll
        for (int t = 0; t < C / 8; ++t) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + 8 * t);
            St = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, Qr[4 * t + 0], St, 0, 0, 0);
            St = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, Qr[4 * t + 1], St, 0, 0, 0);
            St = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, Qr[4 * t + 2], St, 0, 0, 0);
            St = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, Qr[4 * t + 3], St, 0, 0, 0);
        }
        // ---- online softmax of this lane's query over the 32 keys of the block (base-2 exponentials)
        float mx = St[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, St[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 0 on the first block (m_run = -inf)
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            St[e] = __builtin_amdgcn_exp2f(St[e] - m_new);
            rs += St[e];
        }
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < CB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[db][e] *= alpha;
        // ---- O^T += V_blk^T P^T: step e reduces over keys (e&3) + 8(e>>2) + 4 lh
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float* vrow = Vs + ((e & 3) + 8 * (e >> 2) + 4 * lh) * KS + l31;
#pragma unroll
            for (int db = 0; db < CB; ++db)
                O[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * db], St[e], O[db], 0, 0, 0);
        }
    }
    // ---- out[i][:] = O^T[:, i] / l: transpose through LDS (one 32 x C tile per wave) for 16-byte row stores
    __syncthreads();
    float* Ts = sm + wave * 32 * KS;  // [query][KS]  (4 waves x 32 x KS floats <= the K + V blocks when 2 * 32 >= 4 * 32 ... see launcher)
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int db = 0; db < CB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) Ts[l31 * KS + 32 * db + (e & 3) + 8 * (e >> 2) + 4 * lh] = O[db][e] * inv;
    __builtin_amdgcn_wave_barrier();
    if (active) {
        float* ob = out + ((size_t)b * n + q0) * C;
#pragma unroll
        for (int i = 0; i < 32 * C / 4 / 64; ++i) {
            const int it = lane + 64 * i;
            const int row = it / (C / 4), c4 = it % (C / 4);
            *reinterpret_cast<f32x4*>(ob + (size_t)row * C + 4 * c4) = *reinterpret_cast<const f32x4*>(Ts + row * KS + 4 * c4);
        }
    }
}

bool vae_attn_mfma_ok(int n, int C) { return n % 32 == 0 && (C == 64 || C == 128 || C == 256); }

int launch_vae_attn_mfma(const float* q, const float* k, const float* v, float* out, int B, int n, int C, hipStream_t s) {
    DM_REQUIRE(vae_attn_mfma_ok(n, C), "VAE attention (MFMA): n % 32 == 0 and C in {64, 128, 256}");
    const float scale_log2e = (1.0f / sqrtf((float)C)) * 1.4426950408889634f;
    const dim3 grid((n + 127) / 128, B), block(256);
    // LDS: K and V blocks (2 x 32 rows) during the loop, four 32-row output tiles afterwards
    const size_t lds = (size_t)4 * 32 * (C + 4) * sizeof(float);
#define DM_VAE_ATTN(CB_)                                                                                              \
    {                                                                                                                 \
        static LdsOptIn lds_flag; \
        if (lds_opt_in(lds_flag, reinterpret_cast<const void*>(vae_attn_mfma_kernel<CB_>), 1)) return 1;                                                                                                             \
        hipLaunchKernelGGL(vae_attn_mfma_kernel<CB_>, grid, block, lds, s, q, k, v, out, n, scale_log2e);             \
    }
    if (C == 64) DM_VAE_ATTN(2)
    else if (C == 128) DM_VAE_ATTN(4)
    else DM_VAE_ATTN(8)
#undef DM_VAE_ATTN
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// GroupNorm statistics, NHWC, deterministic.  Stage 1: one block per (image, slab of pixel rows) reads whole pixel rows
// (16 bytes per lane, coalesced), keeps per-lane channel sums in fp32 over its rows, folds a lane's quad in double, combines
// the lanes of one group with wavefront shuffles (a group's channels are adjacent lanes of one pixel row: a fixed butterfly),
// the row slots of the block through LDS in a fixed order, and writes its (sum, sum of squares) per group to
// part[b][block][2 * groups].  Stage 2: one thread per (image, group) adds the blocks in order -> (mean, rstd).  No atomics:
// the result does not depend on arrival order.
// ---------------------------------------------------------------------------------------
constexpr int GS_MAX_BLOCKS = 64;  // slabs per image (the slab grows with the image instead)
__global__ __launch_bounds__(256) void group_sums_kernel(const float* __restrict__ x, double* __restrict__ part, int HW, int C,
                                                         int groups, int rows_per_block) {
    __shared__ double red[256][8];  // per thread: (sum, sum of squares) of its quad [0..1], or per element of it [2e, 2e + 1]
    const int b = blockIdx.y;
    const int c4n = C / 4;                 // float4 per pixel
    const int tpr = min(c4n, 256);         // threads per pixel row (C <= 1024)
    const int rpp = 256 / tpr;             // pixel rows in flight per pass
    const int c4 = threadIdx.x % tpr, sub = threadIdx.x / tpr;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, HW);
    f32x4 s = make_f32x4(0.f, 0.f, 0.f, 0.f), ss = make_f32x4(0.f, 0.f, 0.f, 0.f);
    if (sub < rpp)
        for (int r = r0 + sub; r < r1; r += rpp) {
            const f32x4 vx = *reinterpret_cast<const f32x4*>(x + ((size_t)b * HW + r) * C + 4 * c4);
            s += vx;
            ss += vx * vx;
        }
    const int cg = C / groups;  // channels per group (>= 1)
    if (cg % 4 != 0) {
        // a quad straddles groups (2 channels per group at C = 64, 6 at C = 192): per-element sums through LDS, each group
        // added up channel by channel and row slot by row slot, in order
        const float sv[4] = {s.x, s.y, s.z, s.w}, qv[4] = {ss.x, ss.y, ss.z, ss.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[threadIdx.x][2 * e] = (double)sv[e];
            red[threadIdx.x][2 * e + 1] = (double)qv[e];
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * groups) {
            const int grp = threadIdx.x >> 1, which = threadIdx.x & 1;
            double t = 0.0;
            for (int sb = 0; sb < rpp; ++sb)
                for (int c = grp * cg; c < (grp + 1) * cg; ++c) t += red[sb * tpr + (c >> 2)][2 * (c & 3) + which];
            part[((size_t)b * gridDim.x + blockIdx.x) * 2 * groups + threadIdx.x] = t;
        }
        return;
    }
    const int lpg = cg / 4;  // a group = lpg adjacent lanes of a pixel row
    double ds = ((double)s.x + (double)s.y) + ((double)s.z + (double)s.w);
    double dq = ((double)ss.x + (double)ss.y) + ((double)ss.z + (double)ss.w);
    if ((lpg & (lpg - 1)) == 0 && lpg <= 64) {
        // power-of-two group width: butterfly over the group's lanes (tpr is a multiple of lpg and divides 256, so a group
        // never straddles a wavefront); idle row slots (sub >= rpp) hold zeros and take part harmlessly
        for (int m = 1; m < lpg; m <<= 1) {
            ds += __shfl_xor(ds, m);
            dq += __shfl_xor(dq, m);
        }
        red[threadIdx.x][0] = ds;
        red[threadIdx.x][1] = dq;
        __syncthreads();
        if ((int)threadIdx.x < 2 * groups) {
            const int grp = threadIdx.x >> 1, which = threadIdx.x & 1;
            double t = 0.0;
            for (int sb = 0; sb < rpp; ++sb) t += red[sb * tpr + grp * lpg][which];
            part[((size_t)b * gridDim.x + blockIdx.x) * 2 * groups + threadIdx.x] = t;
        }
    } else {
        red[threadIdx.x][0] = ds;
        red[threadIdx.x][1] = dq;
        __syncthreads();
        if ((int)threadIdx.x < 2 * groups) {  // e.g. 96 channels per group = 24 lanes: summed lane by lane, in order
            const int grp = threadIdx.x >> 1, which = threadIdx.x & 1;
            double t = 0.0;
            for (int sb = 0; sb < rpp; ++sb)
                for (int l = 0; l < lpg; ++l) t += red[sb * tpr + grp * lpg + l][which];
            part[((size_t)b * gridDim.x + blockIdx.x) * 2 * groups + threadIdx.x] = t;
        }
    }
}

__global__ void group_finish_kernel(const double* __restrict__ part, float* __restrict__ stats, int n_stats, int groups,
                                    int n_blocks, double inv_n, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (b, group)
    if (i >= n_stats) return;
    const int b = i / groups, g = i - b * groups;
    double sum = 0.0, sq = 0.0;
    for (int k = 0; k < n_blocks; ++k) {
        const double* q = part + ((size_t)b * n_blocks + k) * 2 * groups + 2 * g;
        sum += q[0];
        sq += q[1];
    }
    const double mean = sum * inv_n;
    const double var = fmax(sq * inv_n - mean * mean, 0.0);
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

bool group_sums_ok(int C, int groups) { return groups == 32 && C % 32 == 0 && C % 4 == 0 && C <= 1024; }
// floats of the statistics workspace: B * groups * 2 (mean, rstd), then the per-block partial sums as doubles
size_t group_stats_ws_floats(int B, int groups) { return (size_t)B * groups * 2 + (size_t)B * GS_MAX_BLOCKS * 2 * groups * 2; }

int launch_group_stats_fast(const float* x, float* stats, double* part, int B, int HW, int C, int groups, float eps,
                            hipStream_t s) {
    DM_REQUIRE(group_sums_ok(C, groups), "group_sums: 32 groups, C % 32 == 0, C <= 1024");
    const int rows_per_block = std::max(64, (HW + GS_MAX_BLOCKS - 1) / GS_MAX_BLOCKS);
    const int n_blocks = (HW + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(group_sums_kernel, dim3(n_blocks, B), dim3(256), 0, s, x, part, HW, C, groups, rows_per_block);
    const int ns = B * groups;
    hipLaunchKernelGGL(group_finish_kernel, dim3((ns + 255) / 256), dim3(256), 0, s, part, stats, ns, groups, n_blocks,
                       1.0 / ((double)HW * (C / groups)), eps);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dm

// Backward kernels of the training step that are not convolutions (DD/denoising_diffusion.py:805-900 is the loss; the
// operators differentiated here are Block :113-122 / RMSNorm :66-67, the time MLP :280-285 and ResnetBlock.mlp :127-130,
// Downsample / Upsample rearrangements :48-58, final_conv, q_sample :813-821 and the weighted MSE :874-878).
// All activations NHWC fp32 (rows = pixels, C contiguous) unless the name says nchw.  HBM-bound.
#include "conv_device.h"

#include <algorithm>

namespace dm {

// ---------------------------------------------------------------------------------------
// Block / RMSNorm backward, one pass over the rows:
//   forward   n = u / max(||u||, 1e-12);  w = n * g * sqrt(C);  v = w * (scale + 1) + shift;  y = v * sigmoid(v)
//   backward  dv = dy * silu'(v);  dshift += dv;  dscale += dv * w;  dw = dv * (scale + 1);  dg += dw * n * sqrt(C);
//             dn = dw * g * sqrt(C);  du = (dn - n * <n, dn>) / max(||u||, 1e-12);  dbias += du
// A wave works on 64 / LPR rows at a time (LPR lanes of 4 channels per row); per-channel sums are kept in registers over
// the rows of the workgroup's chunk (one image), combined through LDS in a fixed order, and written as partial sums
// [image][chunk][4][C]; rowgrad_reduce_kernel finishes them (deterministic: no atomics).
// ---------------------------------------------------------------------------------------
// Philox4x32-10 (the dropout masks below and in norm_act_bwd_kernel)
__device__ __forceinline__ void philox4(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
struct NormBwdParams {
    const float* dy;
    const float* u;
    const float* g;
    const float* ss;   // scale = ss[b * ss_stride + c], shift = ss[b * ss_stride + C + c]; nullptr: none
    float* du;
    const float* add;  // optional: du = (gradient through the norm) + add (the residual branch of `attn(x) + x`)
    float* part;       // [B][chunks][4][C]: dg, dbias, dscale, dshift
    int C, C4, LPR, NV, ss_stride, flags;
    int pix_per_image, rows_per_chunk, chunks;
    // dy given as the K-split partial sums of the convolution that produced it (dy_nsplit tensors, dy_stride floats apart):
    // summed on load, so no landing pass runs between that convolution and this kernel
    int dy_nsplit;
    long long dy_stride;
    // nn.Dropout between the activation and this kernel's input gradient: dy is multiplied by the mask of the forward pass
    // (dropout_kernel below: Philox keyed by (seed, stream, flat float4 index)), recomputed here instead of a separate pass
    float drop_p, drop_inv_keep;
    uint64_t drop_seed, drop_stream;
};

template <int NV>
__global__ __launch_bounds__(256) void norm_act_bwd_kernel(const NormBwdParams p) {
    extern __shared__ float red[];  // [groups][4][C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LPR = p.LPR, RPW = 64 / LPR;
    const int sub = lane % LPR, rg = lane / LPR;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int row_lo = chunk * p.rows_per_chunk, row_hi = min(row_lo + p.rows_per_chunk, p.pix_per_image);
    const float sqrtC = sqrtf((float)p.C);
    const bool has_ss = p.ss != nullptr && (p.flags & EPI_SCALE_SHIFT);
    const bool silu = (p.flags & EPI_SILU) != 0;
    const bool norm = (p.flags & EPI_NORM) != 0;
    f32x4 gq[NV], sc[NV], sh[NV];
    f32x4 a_g[NV], a_b[NV], a_sc[NV], a_sh[NV];
    bool cv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c4 = sub + LPR * v;
        cv[v] = c4 < p.C4;
        const f32x4 z = make_f32x4(0.f, 0.f, 0.f, 0.f), one = make_f32x4(1.f, 1.f, 1.f, 1.f);
        gq[v] = (cv[v] && p.g) ? *reinterpret_cast<const f32x4*>(p.g + 4 * c4) : one;
        sc[v] = (cv[v] && has_ss) ? *reinterpret_cast<const f32x4*>(p.ss + (size_t)b * p.ss_stride + 4 * c4) : z;
        sh[v] = (cv[v] && has_ss) ? *reinterpret_cast<const f32x4*>(p.ss + (size_t)b * p.ss_stride + p.C + 4 * c4) : z;
        a_g[v] = a_b[v] = a_sc[v] = a_sh[v] = z;
    }
    for (int r0 = row_lo + wave * RPW; r0 < row_hi; r0 += 4 * RPW) {
        const int r = r0 + rg;
        const bool rv = r < row_hi;
        const size_t base = ((size_t)b * p.pix_per_image + (rv ? r : row_lo)) * p.C;
        f32x4 uv[NV], dv[NV];
        float ssq = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f32x4 z = make_f32x4(0.f, 0.f, 0.f, 0.f);
            const int c4 = sub + LPR * v;
            uv[v] = (cv[v] && rv) ? *reinterpret_cast<const f32x4*>(p.u + base + 4 * c4) : z;
            dv[v] = (cv[v] && rv) ? *reinterpret_cast<const f32x4*>(p.dy + base + 4 * c4) : z;
            if (cv[v] && rv)
                for (int sp = 1; sp < p.dy_nsplit; ++sp)
                    dv[v] += *reinterpret_cast<const f32x4*>(p.dy + (size_t)sp * p.dy_stride + base + 4 * c4);
            if (p.drop_p > 0.f) {
                const uint64_t i4 = (uint64_t)(base >> 2) + c4;
                uint32_t c[4] = {(uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)p.drop_stream, (uint32_t)(p.drop_stream >> 32)};
                philox4(c, (uint32_t)p.drop_seed, (uint32_t)(p.drop_seed >> 32));
                float* q = reinterpret_cast<float*>(&dv[v]);
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = ((float)c[j] * 2.3283064365386963e-10f >= p.drop_p) ? q[j] * p.drop_inv_keep : 0.f;
            }
            ssq += uv[v].x * uv[v].x + uv[v].y * uv[v].y + uv[v].z * uv[v].z + uv[v].w * uv[v].w;
        }
        for (int m = 1; m < LPR; m <<= 1) ssq += __shfl_xor(ssq, m);
        const float nrm = sqrtf(ssq);
        const float rinv = norm ? 1.0f / fmaxf(nrm, 1e-12f) : 1.0f;
        const bool clamped = nrm < 1e-12f;  // F.normalize divides by the clamp: no projection term
        f32x4 nn[NV], dn[NV];
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float* un = reinterpret_cast<float*>(&uv[v]);
            float* dd = reinterpret_cast<float*>(&dv[v]);
            float* gg = reinterpret_cast<float*>(&gq[v]);
            float* s1 = reinterpret_cast<float*>(&sc[v]);
            float* s2 = reinterpret_cast<float*>(&sh[v]);
            float* ag = reinterpret_cast<float*>(&a_g[v]);
            float* asc = reinterpret_cast<float*>(&a_sc[v]);
            float* ash = reinterpret_cast<float*>(&a_sh[v]);
            float* np = reinterpret_cast<float*>(&nn[v]);
            float* dnp = reinterpret_cast<float*>(&dn[v]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float n = un[j] * rinv;
                const float w = norm ? n * gg[j] * sqrtC : un[j];
                const float val = has_ss ? w * (s1[j] + 1.0f) + s2[j] : w;
                float d = dd[j];
                if (silu) {
                    const float sig = 1.0f / (1.0f + __expf(-val));
                    d *= sig * (1.0f + val * (1.0f - sig));
                }
                ash[j] += d;
                asc[j] += d * w;
                const float dw = has_ss ? d * (s1[j] + 1.0f) : d;
                ag[j] += dw * n * sqrtC;
                const float dnj = norm ? dw * gg[j] * sqrtC : dw;
                np[j] = n;
                dnp[j] = dnj;
                dot += n * dnj;
            }
        }
        for (int m = 1; m < LPR; m <<= 1) dot += __shfl_xor(dot, m);
        if (!norm || clamped) dot = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            f32x4 o;
            float* op = reinterpret_cast<float*>(&o);
            float* np = reinterpret_cast<float*>(&nn[v]);
            float* dnp = reinterpret_cast<float*>(&dn[v]);
            float* ab = reinterpret_cast<float*>(&a_b[v]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                op[j] = (dnp[j] - np[j] * dot) * rinv;
                ab[j] += op[j];
            }
            if (cv[v] && rv) {
                if (p.add) o += *reinterpret_cast<const f32x4*>(p.add + base + 4 * (sub + LPR * v));
                *reinterpret_cast<f32x4*>(p.du + base + 4 * (sub + LPR * v)) = o;
            }
        }
    }
    // ---- combine the row groups of the workgroup in a fixed order
    const int groups = 4 * RPW, grp = wave * RPW + rg;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c4 = sub + LPR * v;
        if (!cv[v]) continue;
        float* dst = red + (size_t)grp * 4 * p.C + 4 * c4;
        *reinterpret_cast<f32x4*>(dst) = a_g[v];
        *reinterpret_cast<f32x4*>(dst + p.C) = a_b[v];
        *reinterpret_cast<f32x4*>(dst + 2 * p.C) = a_sc[v];
        *reinterpret_cast<f32x4*>(dst + 3 * p.C) = a_sh[v];
    }
    __syncthreads();
    for (int i = tid; i < 4 * p.C; i += 256) {
        float s = 0.f;
        for (int gi = 0; gi < groups; ++gi) s += red[(size_t)gi * 4 * p.C + i];
        p.part[((size_t)b * p.chunks + chunk) * 4 * p.C + i] = s;
    }
}

// The forward of the same Block in training mode with dropout, in one pass over the rows (the sampling path's landing kernel
// followed by dropout_kernel otherwise):  y = dropout(silu(norm(u) * g * sqrt(C) * (scale + 1) + shift)) [+ residual].
// Same row mapping as norm_act_bwd_kernel; the mask is the one dropout_kernel / norm_act_bwd_kernel form from
// (seed, stream, flat float4 index).  grid (chunks, B).
struct NormFwdParams {
    const float* u;
    // u given as the K-split partial sums of the convolution (nsplit tensors, split_stride floats apart) [+ bias]: the sum is
    // what the backward pass needs, so it is stored to u_out in the same pass (tape forward: one launch instead of a landing
    // pass followed by the norm pass)
    int nsplit;
    long long split_stride;
    const float* bias;
    float* u_out;
    const float* g;
    const float* ss;
    const float* residual;
    float* y;
    int C, C4, LPR, ss_stride, flags, pix_per_image, rows_per_chunk;
    float drop_p, drop_inv_keep;
    uint64_t drop_seed, drop_stream;
};
template <int NV>
__global__ __launch_bounds__(256) void norm_act_drop_kernel(const NormFwdParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LPR = p.LPR, RPW = 64 / LPR;
    const int sub = lane % LPR, rg = lane / LPR;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int row_lo = chunk * p.rows_per_chunk, row_hi = min(row_lo + p.rows_per_chunk, p.pix_per_image);
    const float sqrtC = sqrtf((float)p.C);
    const bool has_ss = p.ss != nullptr && (p.flags & EPI_SCALE_SHIFT);
    f32x4 gq[NV], sc[NV], sh[NV], bq[NV];
    bool cv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c4 = sub + LPR * v;
        cv[v] = c4 < p.C4;
        const f32x4 z = make_f32x4(0.f, 0.f, 0.f, 0.f), one = make_f32x4(1.f, 1.f, 1.f, 1.f);
        gq[v] = (cv[v] && p.g) ? *reinterpret_cast<const f32x4*>(p.g + 4 * c4) : one;
        bq[v] = (cv[v] && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + 4 * c4) : z;
        sc[v] = (cv[v] && has_ss) ? *reinterpret_cast<const f32x4*>(p.ss + (size_t)b * p.ss_stride + 4 * c4) : z;
        sh[v] = (cv[v] && has_ss) ? *reinterpret_cast<const f32x4*>(p.ss + (size_t)b * p.ss_stride + p.C + 4 * c4) : z;
    }
    for (int r0 = row_lo + wave * RPW; r0 < row_hi; r0 += 4 * RPW) {
        const int r = r0 + rg;
        const bool rv = r < row_hi;
        const size_t base = ((size_t)b * p.pix_per_image + (rv ? r : row_lo)) * p.C;
        f32x4 uv[NV];
        float ssq = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            uv[v] = make_f32x4(0.f, 0.f, 0.f, 0.f);
            if (cv[v] && rv) {
                const float* up = p.u + base + 4 * (sub + LPR * v);
                uv[v] = *reinterpret_cast<const f32x4*>(up);
                for (int sp = 1; sp < p.nsplit; ++sp) uv[v] += *reinterpret_cast<const f32x4*>(up + (size_t)sp * p.split_stride);
                uv[v] += bq[v];
                if (p.u_out) *reinterpret_cast<f32x4*>(p.u_out + base + 4 * (sub + LPR * v)) = uv[v];
            }
            ssq += uv[v].x * uv[v].x + uv[v].y * uv[v].y + uv[v].z * uv[v].z + uv[v].w * uv[v].w;
        }
        for (int m = 1; m < LPR; m <<= 1) ssq += __shfl_xor(ssq, m);
        const float rinv = 1.0f / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c4 = sub + LPR * v;
            f32x4 o;
            float* op = reinterpret_cast<float*>(&o);
            const float* un = reinterpret_cast<const float*>(&uv[v]);
            const float* gg = reinterpret_cast<const float*>(&gq[v]);
            const float* s1 = reinterpret_cast<const float*>(&sc[v]);
            const float* s2 = reinterpret_cast<const float*>(&sh[v]);
            uint32_t c[4] = {0, 0, 0, 0};
            if (p.drop_p > 0.f) {
                const uint64_t i4 = (uint64_t)(base >> 2) + c4;
                c[0] = (uint32_t)i4; c[1] = (uint32_t)(i4 >> 32); c[2] = (uint32_t)p.drop_stream; c[3] = (uint32_t)(p.drop_stream >> 32);
                philox4(c, (uint32_t)p.drop_seed, (uint32_t)(p.drop_seed >> 32));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = un[j] * rinv * gg[j] * sqrtC;
                const float val = has_ss ? w * (s1[j] + 1.0f) + s2[j] : w;
                float a = val / (1.0f + __expf(-val));
                if (p.drop_p > 0.f) a = ((float)c[j] * 2.3283064365386963e-10f >= p.drop_p) ? a * p.drop_inv_keep : 0.f;
                op[j] = a;
            }
            if (cv[v] && rv) {
                if (p.residual) o += *reinterpret_cast<const f32x4*>(p.residual + base + 4 * c4);
                *reinterpret_cast<f32x4*>(p.y + base + 4 * c4) = o;
            }
        }
    }
}

// stage 1, grid (ceil(C / 64), B): sums over the chunks of one image: dss[b][c] / dss[b][C + c] (the image's scale / shift
// gradient) and img[b][2][C] (its share of dg / dbias);  stage 2: dg[c] (+)= sum_b, dbias likewise.  Fixed order throughout.
__global__ void rowgrad_image_kernel(const float* __restrict__ part, int chunks, int C, float* __restrict__ dss,
                                     int dss_stride, float* __restrict__ img) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    float sg = 0.f, sb = 0.f, ssc = 0.f, ssh = 0.f;
    for (int k = 0; k < chunks; ++k) {
        const float* q = part + ((size_t)b * chunks + k) * 4 * C;
        sg += q[c];
        sb += q[C + c];
        ssc += q[2 * C + c];
        ssh += q[3 * C + c];
    }
    if (dss) {
        dss[(size_t)b * dss_stride + c] = ssc;
        dss[(size_t)b * dss_stride + C + c] = ssh;
    }
    img[((size_t)b * 2) * C + c] = sg;
    img[((size_t)b * 2 + 1) * C + c] = sb;
}
// 256 threads = 64 channels x 4 lanes over the images, combined through LDS in a fixed order
__global__ __launch_bounds__(256) void rowgrad_finish_kernel(const float* __restrict__ img, int B, int C, float* __restrict__ dg,
                                                             float* __restrict__ dbias, int accumulate) {
    __shared__ float red[2][4][64];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + li;
    float sg = 0.f, sb = 0.f;
    if (c < C)
        for (int b = q; b < B; b += 4) {
            sg += img[((size_t)b * 2) * C + c];
            sb += img[((size_t)b * 2 + 1) * C + c];
        }
    red[0][q][li] = sg;
    red[1][q][li] = sb;
    __syncthreads();
    if (q == 0 && c < C) {
        sg = (red[0][0][li] + red[0][1][li]) + (red[0][2][li] + red[0][3][li]);
        sb = (red[1][0][li] + red[1][1][li]) + (red[1][2][li] + red[1][3][li]);
        if (dg) dg[c] = accumulate ? dg[c] + sg : sg;
        if (dbias) dbias[c] = accumulate ? dbias[c] + sb : sb;
    }
}

// The two stages above for a table of layers (the training step defers the reductions behind every norm_act_bwd_kernel of
// a backward pass to two launches in front of the time-MLP gradients): workgroup -> job through the block prefixes.
__global__ __launch_bounds__(64) void rowgrad_image_jobs_kernel(const RowgradJob* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_img_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const RowgradJob j = jobs[lo];
    const int lb = blockIdx.x - j.first_img_block, cb = (j.C + 63) / 64;
    const int c = (lb % cb) * 64 + threadIdx.x, b = lb / cb;
    if (c >= j.C) return;
    const int C = j.C;
    float sg = 0.f, sb = 0.f, ssc = 0.f, ssh = 0.f;
    int k = 0;
    for (; k + 8 <= j.chunks; k += 8) {  // 32 loads in flight, added in chunk order (a small batch has up to 64 chunks per image)
        float v[8][4];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float* q = j.part + ((size_t)b * j.chunks + k + kk) * 4 * C;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[kk][r] = q[r * C + c];
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            sg += v[kk][0];
            sb += v[kk][1];
            ssc += v[kk][2];
            ssh += v[kk][3];
        }
    }
    for (; k < j.chunks; ++k) {
        const float* q = j.part + ((size_t)b * j.chunks + k) * 4 * C;
        sg += q[c];
        sb += q[C + c];
        ssc += q[2 * C + c];
        ssh += q[3 * C + c];
    }
    if (j.dss) {
        j.dss[(size_t)b * j.dss_stride + c] = ssc;
        j.dss[(size_t)b * j.dss_stride + C + c] = ssh;
    }
    j.img[((size_t)b * 2) * C + c] = sg;
    j.img[((size_t)b * 2 + 1) * C + c] = sb;
}
__global__ __launch_bounds__(256) void rowgrad_finish_jobs_kernel(const RowgradJob* __restrict__ jobs, int n_jobs) {
    __shared__ float red[2][4][64];
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_fin_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const RowgradJob j = jobs[lo];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = (blockIdx.x - j.first_fin_block) * 64 + li, C = j.C;
    float sg = 0.f, sb = 0.f;
    if (c < C)
        for (int b = q; b < j.B; b += 4) {
            sg += j.img[((size_t)b * 2) * C + c];
            sb += j.img[((size_t)b * 2 + 1) * C + c];
        }
    red[0][q][li] = sg;
    red[1][q][li] = sb;
    __syncthreads();
    if (q == 0 && c < C) {
        sg = (red[0][0][li] + red[0][1][li]) + (red[0][2][li] + red[0][3][li]);
        sb = (red[1][0][li] + red[1][1][li]) + (red[1][2][li] + red[1][3][li]);
        if (j.dg) j.dg[c] = j.accumulate ? j.dg[c] + sg : sg;
        if (j.dbias) j.dbias[c] = j.accumulate ? j.dbias[c] + sb : sb;
    }
}
// jobs_host gets its block prefixes here; jobs_dev must hold the same table
void rowgrad_jobs_prefix(std::vector<RowgradJob>& jobs, int* img_blocks, int* fin_blocks) {
    int a = 0, f = 0;
    for (RowgradJob& j : jobs) {
        j.first_img_block = a;
        j.first_fin_block = f;
        a += ((j.C + 63) / 64) * j.B;
        f += (j.C + 63) / 64;
    }
    *img_blocks = a;
    *fin_blocks = f;
}
int launch_rowgrad_jobs(const RowgradJob* jobs_dev, int n_jobs, int img_blocks, int fin_blocks, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    hipLaunchKernelGGL(rowgrad_image_jobs_kernel, dim3(img_blocks), dim3(64), 0, s, jobs_dev, n_jobs);
    DM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(rowgrad_finish_jobs_kernel, dim3(fin_blocks), dim3(256), 0, s, jobs_dev, n_jobs);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

static int pow2ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Row chunks per image of the norm kernels below: about 1024 workgroups in total, and no chunk smaller than the rows one
// workgroup covers per loop iteration (4 waves x 64 / LPR rows: 16 rows at C = 64, 4 at C >= 256 -- a 4x4 map of 512 channels
// at batch 64 is 256 workgroups, not 64)
static int norm_chunks(int B, int pix_per_image, int C) {
    const int lpr = std::min(64, pow2ceil(C / 4));
    const int rows_min = 4 * (64 / lpr);
    return std::max(1, std::min(pix_per_image / rows_min, (1024 + B - 1) / B));
}
size_t norm_act_bwd_ws_floats(int B, int pix_per_image, int C) {
    const int chunks = norm_chunks(B, pix_per_image, C);
    return (size_t)B * chunks * 4 * C + (size_t)B * 2 * C;  // chunk partials + per-image dg / dbias
}

// dy, u, du: [B * pix_per_image][C].  dg / dbias: [C] parameter gradients ((+)= with accumulate); dss: [B][dss_stride]
// receives (dscale | dshift) at columns [0, 2C) of each row when the forward had a scale-shift, else nullptr.
int launch_norm_act_bwd(const float* dy, const float* u, const float* g, const float* ss, int ss_stride, int pix_per_image,
                        float* du, float* ws, float* dg, float* dbias, float* dss, int dss_stride, int B, int C, int flags,
                        int accumulate, hipStream_t s, RowgradJob* defer, float drop_p, uint64_t drop_seed,
                        uint64_t drop_stream, const float* add, int dy_nsplit, int64_t dy_stride) {
    DM_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024, "norm_act_bwd: C must be a multiple of 4, at most 1024");
    DM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "norm_act_bwd: dropout probability");
    NormBwdParams p{};
    p.drop_p = drop_p; p.drop_inv_keep = 1.0f / (1.0f - drop_p); p.drop_seed = drop_seed; p.drop_stream = drop_stream;
    p.dy = dy; p.u = u; p.g = g; p.ss = ss; p.du = du; p.part = ws; p.add = add;
    DM_REQUIRE(dy_nsplit >= 1 && dy_stride % 4 == 0, "norm_act_bwd: partial sums of dy");
    p.dy_nsplit = dy_nsplit; p.dy_stride = dy_stride;
    p.C = C; p.C4 = C / 4;
    p.LPR = std::min(64, pow2ceil(p.C4));
    p.NV = (p.C4 + p.LPR - 1) / p.LPR;
    p.ss_stride = ss_stride; p.flags = flags; p.pix_per_image = pix_per_image;
    p.chunks = norm_chunks(B, pix_per_image, C);
    p.rows_per_chunk = (pix_per_image + p.chunks - 1) / p.chunks;
    p.chunks = (pix_per_image + p.rows_per_chunk - 1) / p.rows_per_chunk;
    const int groups = 4 * (64 / p.LPR);
    const size_t lds = (size_t)groups * 4 * C * sizeof(float);
    DM_REQUIRE(lds <= 64 * 1024, "norm_act_bwd: LDS");
    const dim3 grid(p.chunks, B);
    switch (p.NV) {
        case 1: hipLaunchKernelGGL(norm_act_bwd_kernel<1>, grid, dim3(256), lds, s, p); break;
        case 2: hipLaunchKernelGGL(norm_act_bwd_kernel<2>, grid, dim3(256), lds, s, p); break;
        case 3: hipLaunchKernelGGL(norm_act_bwd_kernel<3>, grid, dim3(256), lds, s, p); break;
        default: hipLaunchKernelGGL(norm_act_bwd_kernel<4>, grid, dim3(256), lds, s, p); break;
    }
    DM_CHECK_HIP(hipGetLastError());
    float* img = ws + (size_t)B * p.chunks * 4 * C;
    if (defer) {
        *defer = RowgradJob{ws, (flags & EPI_SCALE_SHIFT) ? dss : nullptr, img, dg, dbias, p.chunks, C, dss_stride, B, accumulate, 0, 0};
        return 0;
    }
    hipLaunchKernelGGL(rowgrad_image_kernel, dim3((C + 63) / 64, B), dim3(64), 0, s, ws, p.chunks, C,
                       (flags & EPI_SCALE_SHIFT) ? dss : nullptr, dss_stride, img);
    DM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(rowgrad_finish_kernel, dim3((C + 63) / 64), dim3(256), 0, s, img, B, C, dg, dbias, accumulate);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// RMSNorm -> (scale + 1, shift) -> SiLU -> dropout [+ residual] of a training-mode Block: u, y, residual [B * pix][C]
int launch_norm_act_drop(const float* u, const float* g, const float* ss, int ss_stride, int pix_per_image, const float* residual,
                         float* y, int B, int C, int flags, float drop_p, uint64_t drop_seed, uint64_t drop_stream, hipStream_t s,
                         int nsplit, int64_t split_stride, const float* bias, float* u_out) {
    DM_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024, "norm_act_drop: C must be a multiple of 4, at most 1024");
    DM_REQUIRE(nsplit >= 1 && split_stride % 4 == 0, "norm_act_drop: partial sums of u");
    DM_REQUIRE((flags & EPI_NORM) && (flags & EPI_SILU) && drop_p >= 0.f && drop_p < 1.f, "norm_act_drop: a Block's epilogue");
    NormFwdParams p{};
    p.u = u; p.g = g; p.ss = ss; p.residual = residual; p.y = y;
    p.nsplit = nsplit; p.split_stride = split_stride; p.bias = bias; p.u_out = u_out;
    p.C = C; p.C4 = C / 4;
    p.LPR = std::min(64, pow2ceil(p.C4));
    const int NV = (p.C4 + p.LPR - 1) / p.LPR;
    p.ss_stride = ss_stride; p.flags = flags; p.pix_per_image = pix_per_image;
    int chunks = norm_chunks(B, pix_per_image, C);
    p.rows_per_chunk = (pix_per_image + chunks - 1) / chunks;
    chunks = (pix_per_image + p.rows_per_chunk - 1) / p.rows_per_chunk;
    p.drop_p = drop_p; p.drop_inv_keep = 1.0f / (1.0f - drop_p); p.drop_seed = drop_seed; p.drop_stream = drop_stream;
    const dim3 grid(chunks, B);
    switch (NV) {
        case 1: hipLaunchKernelGGL(norm_act_drop_kernel<1>, grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL(norm_act_drop_kernel<2>, grid, dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL(norm_act_drop_kernel<3>, grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL(norm_act_drop_kernel<4>, grid, dim3(256), 0, s, p); break;
    }
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Column sums (bias gradients of convolutions that have no norm behind them): out[c] (+)= sum_r x[r * ld + c * cs].
// Two passes: [blocks][C] partial sums, then the finish.
// ---------------------------------------------------------------------------------------
// 256 threads = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t row_stride,
                                                             int64_t col_stride, int rows_per_block, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + li;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, rows);
    float s = 0.f;
    if (c < C)
        for (int64_t r = r0 + q; r < r1; r += 4) s += x[r * row_stride + c * col_stride];
    red[q][li] = s;
    __syncthreads();
    if (q == 0 && c < C) part[(size_t)blockIdx.y * C + c] = (red[0][li] + red[1][li]) + (red[2][li] + red[3][li]);
}
// finish / few rows (a batch of per-image vectors, the [blocks][C] partial sums): one pass, 1024 threads = 64 columns x 16
// lanes over the rows, combined in a fixed order
__global__ __launch_bounds__(1024) void colsum_lanes_kernel(const float* __restrict__ x, int rows, int C, int64_t row_stride,
                                                            int64_t col_stride, float* __restrict__ out, int accumulate) {
    __shared__ float red[16][64];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + li;
    float s = 0.f;
    if (c < C)
        for (int r = q; r < rows; r += 16) s += x[r * row_stride + c * col_stride];
    red[q][li] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][li];
        out[c] = accumulate ? out[c] + v : v;
    }
}
// the two passes for a table of column sums (every bias / mem_kv gradient of a backward pass in two launches)
__global__ __launch_bounds__(256) void colsum_partial_jobs_kernel(const ColsumJob* __restrict__ jobs, int n_jobs) {
    __shared__ float red[4][64];
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_part_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ColsumJob j = jobs[lo];  // jobs without a partial pass have nb == 0 and own no block: never selected
    const int lb = blockIdx.x - j.first_part_block, cb = (j.C + 63) / 64;
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = (lb % cb) * 64 + li, by = lb / cb;
    const long long r0 = (long long)by * j.rpb, r1 = min(r0 + (long long)j.rpb, j.rows);
    float s = 0.f;
    if (c < j.C)
        for (long long r = r0 + q; r < r1; r += 4) s += j.x[r * j.row_stride + c * j.col_stride];
    red[q][li] = s;
    __syncthreads();
    if (q == 0 && c < j.C) j.ws[(size_t)by * j.C + c] = (red[0][li] + red[1][li]) + (red[2][li] + red[3][li]);
}
__global__ __launch_bounds__(1024) void colsum_lanes_jobs_kernel(const ColsumJob* __restrict__ jobs, int n_jobs) {
    __shared__ float red[16][64];
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_fin_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ColsumJob j = jobs[lo];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = (blockIdx.x - j.first_fin_block) * 64 + li;
    const float* x = j.nb > 0 ? j.ws : j.x;
    const long long rows = j.nb > 0 ? j.nb : j.rows, rs = j.nb > 0 ? j.C : j.row_stride, cs = j.nb > 0 ? 1 : j.col_stride;
    float s = 0.f;
    if (c < j.C)
        for (long long r = q; r < rows; r += 16) s += x[r * rs + c * cs];
    red[q][li] = s;
    __syncthreads();
    if (q == 0 && c < j.C) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][li];
        j.out[c] = j.accumulate ? j.out[c] + v : v;
    }
}
void colsum_jobs_prefix(std::vector<ColsumJob>& jobs, int* part_blocks, int* fin_blocks) {
    int a = 0, f = 0;
    // jobs with a partial pass first, so that the binary search over first_part_block never lands on one without
    std::stable_sort(jobs.begin(), jobs.end(), [](const ColsumJob& x, const ColsumJob& y) { return (x.nb > 0) > (y.nb > 0); });
    for (ColsumJob& j : jobs) {
        j.first_part_block = a;
        j.first_fin_block = f;
        a += ((j.C + 63) / 64) * j.nb;
        f += (j.C + 63) / 64;
    }
    *part_blocks = a;
    *fin_blocks = f;
}
int launch_colsum_jobs(const ColsumJob* jobs_dev, int n_jobs, int n_with_partial, int part_blocks, int fin_blocks, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    if (part_blocks > 0) {
        hipLaunchKernelGGL(colsum_partial_jobs_kernel, dim3(part_blocks), dim3(256), 0, s, jobs_dev, n_with_partial);
        DM_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(colsum_lanes_jobs_kernel, dim3(fin_blocks), dim3(1024), 0, s, jobs_dev, n_jobs);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

static int colsum_blocks(int64_t rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(512, rows / 64)); }
size_t colsum_ws_floats(int64_t rows, int C) { return (size_t)colsum_blocks(rows) * C; }
int launch_colsum(const float* x, int64_t rows, int C, int64_t row_stride, int64_t col_stride, float* ws, float* out,
                  int accumulate, hipStream_t s, ColsumJob* defer) {
    if (defer) {
        int nb = 0, rpb = 0;
        if (rows > 512) {
            const int blocks = colsum_blocks(rows);
            rpb = (int)((rows + blocks - 1) / blocks);
            nb = (int)((rows + rpb - 1) / rpb);
        }
        *defer = ColsumJob{x, ws, out, (long long)rows, (long long)row_stride, (long long)col_stride, C, accumulate, nb, rpb, 0, 0};
        return 0;
    }
    if (rows <= 512) {
        hipLaunchKernelGGL(colsum_lanes_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, x, (int)rows, C, row_stride, col_stride,
                           out, accumulate);
        DM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int blocks = colsum_blocks(rows);
    const int rpb = (int)((rows + blocks - 1) / blocks);
    const int nb = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, nb), dim3(256), 0, s, x, rows, C, row_stride, col_stride,
                       rpb, ws);
    DM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(colsum_lanes_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, ws, nb, C, (int64_t)C, (int64_t)1, out,
                       accumulate);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
// bias gradient of a convolution whose dY is NCHW (final_conv): sum over b and pixels of dy[b][c][:]
__global__ __launch_bounds__(1024) void colsum_nchw_kernel(const float* __restrict__ dy, int B, int C, int HW,
                                                           float* __restrict__ out, int accumulate) {
    __shared__ float red[1024];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* row = dy + ((size_t)b * C + c) * HW;
        for (int q = threadIdx.x; q < HW; q += 1024) s += row[q];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 512; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + red[0] : red[0];
}
int launch_colsum_nchw(const float* dy, int B, int C, int HW, float* out, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(colsum_nchw_kernel, dim3(C), dim3(1024), 0, s, dy, B, C, HW, out, accumulate);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Elementwise pieces
// ---------------------------------------------------------------------------------------
// act: 1 SiLU, 2 GELU (exact erf, nn.GELU default)
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    y[i] = act == 1 ? v / (1.0f + __expf(-v)) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
}
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n,
                               int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float d;
    if (act == 1) {
        const float sig = 1.0f / (1.0f + __expf(-v));
        d = sig * (1.0f + v * (1.0f - sig));
    } else {
        d = 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.39894228040143268f * __expf(-0.5f * v * v);
    }
    dx[i] = dy[i] * d;
}
int launch_act_fwd(const float* x, float* y, int64_t n, int act, hipStream_t s) {
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n, act);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, hipStream_t s) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, x, dx, n, act);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// nn.Dropout(p) of Block.forward in training mode (DD/denoising_diffusion.py:111,121): out = x * keep / (1 - p) [+ add],
// keep ~ Bernoulli(1 - p) from Philox4x32-10 keyed by (seed, stream, element index / 4).  The backward pass applies the
// SAME mask to the gradient, recomputed from the key (nothing is stored).  x == nullptr writes the mask factor itself.
__global__ void dropout_kernel(const float* __restrict__ x, const float* __restrict__ add, float* __restrict__ out, int64_t n4,
                               float p, float inv_keep, uint64_t seed, uint64_t stream) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    philox4(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    f32x4 v = x ? reinterpret_cast<const f32x4*>(x)[i] : make_f32x4(1.f, 1.f, 1.f, 1.f);
    float* q = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = ((float)c[j] * 2.3283064365386963e-10f >= p) ? q[j] * inv_keep : 0.f;
    if (add) v += reinterpret_cast<const f32x4*>(add)[i];
    reinterpret_cast<f32x4*>(out)[i] = v;
}
int launch_dropout(const float* x, const float* add, float* out, int64_t n, float p, uint64_t seed, uint64_t stream,
                   hipStream_t s) {
    DM_REQUIRE(n % 4 == 0 && p >= 0.f && p < 1.f, "dropout: length must be a multiple of 4, 0 <= p < 1");
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, add, out, n / 4, p,
                       1.0f / (1.0f - p), seed, stream);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// y = a + b (+ c)
__global__ void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                            float* __restrict__ y, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
    if (c) v += reinterpret_cast<const f32x4*>(c)[i];
    reinterpret_cast<f32x4*>(y)[i] = v;
}
int launch_add3(const float* a, const float* b, const float* c, float* y, int64_t n, hipStream_t s) {
    DM_REQUIRE(n % 4 == 0, "add3: length must be a multiple of 4");
    hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, a, b, c, y, n / 4);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// Adjoint of the space-to-depth gather of Downsample: t is (B, Ho, Wo, 4 C) with channel (p1*2 + p2) * C + c
// (the order the 1x1 kernel's K index runs in); dx (B, 2Ho, 2Wo, C)[2y + p1][2x + p2][c] = t[y][x][(p1*2+p2)*C + c] (+ add)
__global__ void depth_to_space_kernel(const float* __restrict__ t, const float* __restrict__ add, float* __restrict__ dx,
                                      int Ho, int Wo, int C, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % (C / 4));
    int64_t px = i / (C / 4);
    const int X = (int)(px % (2 * Wo));
    px /= 2 * Wo;
    const int Y = (int)(px % (2 * Ho));
    const int64_t b = px / (2 * Ho);
    const int sub = (Y & 1) * 2 + (X & 1);
    f32x4 v = *reinterpret_cast<const f32x4*>(t + (((b * Ho + (Y >> 1)) * Wo + (X >> 1)) * 4 + sub) * (int64_t)C + 4 * c4);
    if (add) v += reinterpret_cast<const f32x4*>(add)[i];
    reinterpret_cast<f32x4*>(dx)[i] = v;
}
int launch_depth_to_space(const float* t, const float* add, float* dx, int B, int Ho, int Wo, int C, hipStream_t s) {
    DM_REQUIRE(C % 4 == 0, "depth_to_space: C % 4");
    const int64_t n4 = (int64_t)B * 4 * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(depth_to_space_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, t, add, dx, Ho, Wo, C, n4);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// Adjoint of nearest x2 upsampling: dx (B, H, W, C)[y][x] = sum of the 2x2 block of du (B, 2H, 2W, C) (+ add)
__global__ void pool2x2_sum_kernel(const float* __restrict__ du, const float* __restrict__ add, float* __restrict__ dx, int H,
                                   int W, int C, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % (C / 4));
    int64_t px = i / (C / 4);
    const int x = (int)(px % W);
    px /= W;
    const int y = (int)(px % H);
    const int64_t b = px / H;
    const float* q = du + (((b * 2 * H + 2 * y) * 2 * W) + 2 * x) * (int64_t)C + 4 * c4;
    const int64_t rs = (int64_t)2 * W * C;
    f32x4 v = *reinterpret_cast<const f32x4*>(q) + *reinterpret_cast<const f32x4*>(q + C) +
              *reinterpret_cast<const f32x4*>(q + rs) + *reinterpret_cast<const f32x4*>(q + rs + C);
    if (add) v += reinterpret_cast<const f32x4*>(add)[i];
    reinterpret_cast<f32x4*>(dx)[i] = v;
}
int launch_pool2x2_sum(const float* du, const float* add, float* dx, int B, int H, int W, int C, hipStream_t s) {
    DM_REQUIRE(C % 4 == 0, "pool2x2_sum: C % 4");
    const int64_t n4 = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(pool2x2_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, du, add, dx, H, W, C, n4);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// Input gradient of a 1x1 convolution with a handful of outputs whose dY is NCHW (final_conv 64 -> 3):
// dx[pixel][c] = sum_o w[o][c] * dy[b][o][pixel]
__global__ void pointwise_small_dgrad_kernel(const float* __restrict__ dy_nchw, const float* __restrict__ w_oc,
                                             float* __restrict__ dx, int64_t pixels, int C, int Cout, int HW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixels * C) return;
    const int c = (int)(i % C);
    const int64_t px = i / C, b = px / HW, q = px - b * HW;
    float s = 0.f;
    for (int o = 0; o < Cout; ++o) s += w_oc[o * C + c] * dy_nchw[(b * Cout + o) * HW + q];
    dx[i] = s;
}
int launch_pointwise_small_dgrad(const float* dy_nchw, const float* w_oc, float* dx, int64_t pixels, int C, int Cout, int HW,
                                 hipStream_t s) {
    const int64_t n = pixels * C;
    hipLaunchKernelGGL(pointwise_small_dgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy_nchw, w_oc, dx,
                       pixels, C, Cout, HW);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// nn.Linear weight gradient: dW[o][i] (+)= sum_r dy[r * ldy + o] * act(x[r * ldx + i]);  act_in: 0 none, 1 silu
__global__ void linear_wgrad_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                    float* __restrict__ dw, int R, int I, int O, int act_in, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int o = blockIdx.y;
    if (i >= I) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) {
        float xv = x[(size_t)r * ldx + i];
        if (act_in == 1) xv = xv / (1.0f + __expf(-xv));
        s += dy[(size_t)r * ldy + o] * xv;
    }
    float* q = dw + (size_t)o * I + i;
    *q = accumulate ? *q + s : s;
}
int launch_linear_wgrad(const float* dy, int ldy, const float* x, int ldx, float* dw, int R, int I, int O, int act_in,
                        int accumulate, hipStream_t s) {
    static const bool no_mfma = std::getenv("DM_NO_SMALL_GEMM") != nullptr;
    if (!no_mfma && act_in == 0 && rows_gemm_tn_ok(R, I, O, ldy, ldx))
        return launch_rows_gemm_tn(dy, ldy, x, ldx, nullptr, nullptr, dw, I, nullptr, R, I, O, accumulate, s);
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3((I + 63) / 64, O), dim3(64), 0, s, dy, ldy, x, ldx, dw, R, I, O, act_in,
                       accumulate);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
// The 19 ResnetBlock.mlp (SiLU -> Linear) gradients in ONE launch: row o of the concatenated [ss_total][I] matrix goes to
// dw_rows[o] (a pointer into that block's mlp.1.weight gradient), its bias gradient to db_rows[o].
// grid (ceil(I / 64), ceil(ss_total / 16)), 64 threads: dW[o][i] = sum_r dy[r][o] * silu(x[r][i]) for 16 rows o per
// workgroup (silu(x) is formed once per r and meets 16 dy values); lane 0 of the first column block also sums the biases.
constexpr int MLP_ROWS = 16;
__global__ __launch_bounds__(64) void mlp_rows_wgrad_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                            int ldx, float* const* __restrict__ dw_rows,
                                                            float* const* __restrict__ db_rows, int R, int I, int O,
                                                            int accumulate) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int o0 = blockIdx.y * MLP_ROWS;
    const bool i_ok = i < I;
    float s[MLP_ROWS], sb[MLP_ROWS];
#pragma unroll
    for (int j = 0; j < MLP_ROWS; ++j) s[j] = sb[j] = 0.f;
    for (int r0 = 0; r0 < R; r0 += 4) {  // the loads of 4 rows first (independent), then the arithmetic
        float xv[4], d[4][MLP_ROWS];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool r_ok = r0 + k < R;
            xv[k] = (i_ok && r_ok) ? x[(size_t)(r0 + k) * ldx + i] : 0.f;
            const float* dr = dy + (size_t)(r0 + k) * ldy + o0;
#pragma unroll
            for (int j = 0; j < MLP_ROWS; ++j) d[k][j] = (r_ok && o0 + j < O) ? dr[j] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = xv[k] / (1.0f + __expf(-xv[k]));
#pragma unroll
            for (int j = 0; j < MLP_ROWS; ++j) {
                s[j] += d[k][j] * a;
                sb[j] += d[k][j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MLP_ROWS; ++j) {
        if (o0 + j >= O) break;
        if (i_ok) {
            float* q = dw_rows[o0 + j] + i;
            *q = accumulate ? *q + s[j] : s[j];
        }
        if (i == 0) {
            float* b = db_rows[o0 + j];
            *b = accumulate ? *b + sb[j] : sb[j];
        }
    }
}
int launch_mlp_rows_wgrad(const float* dy, int ldy, const float* x, int ldx, float* const* dw_rows, float* const* db_rows,
                          int R, int I, int O, int accumulate, hipStream_t s, const float* x_act) {
    static const bool no_mfma = std::getenv("DM_NO_SMALL_GEMM") != nullptr;
    if (!no_mfma && x_act && rows_gemm_tn_ok(R, I, O, ldy, ldx))  // x_act = SiLU(x), formed once by the forward pass
        return launch_rows_gemm_tn(dy, ldy, x_act, ldx, dw_rows, db_rows, nullptr, 0, nullptr, R, I, O, accumulate, s);
    hipLaunchKernelGGL(mlp_rows_wgrad_kernel, dim3((I + 63) / 64, (O + MLP_ROWS - 1) / MLP_ROWS), dim3(64), 0, s, dy, ldy, x, ldx,
                       dw_rows, db_rows, R, I, O, accumulate);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// nn.Linear input gradient: dx[r][i] = sum_o dy[r * ldy + o] * W[o][i]   (W as stored: (O, I)).
// grid (ceil(I / 64), ceil(R / 8), OS), 256 threads = 64 columns x 4 slices of this workgroup's share of the O range: a W
// value is loaded once for 8 rows r, the four slices meet in LDS in a fixed order.  OS > 1 (the 9 k-row scale/shift
// matrix): every share writes its partial (R, I) block to the workspace and linear_dgrad_sum_kernel adds them in order.
constexpr int LD_ROWS = 8;
constexpr int LD_OT = 512;  // o values per staged tile of dy
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ W,
                                                           float* __restrict__ dx, int ldx, int R, int I, int O, int o_share) {
    __shared__ float dys[LD_ROWS][LD_OT];
    __shared__ float red[4][LD_ROWS][64];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + li, r0 = blockIdx.y * LD_ROWS;
    const int oa = blockIdx.z * o_share, ob = min(oa + o_share, O);
    float s[LD_ROWS];
#pragma unroll
    for (int j = 0; j < LD_ROWS; ++j) s[j] = 0.f;
    for (int ot = oa; ot < ob; ot += LD_OT) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < LD_ROWS * LD_OT; idx += 256) {
            const int j = idx / LD_OT, oo = idx - j * LD_OT;
            dys[j][oo] = (r0 + j < R && ot + oo < ob) ? dy[(size_t)(r0 + j) * ldy + ot + oo] : 0.f;
        }
        __syncthreads();
        const int n = min(LD_OT, ob - ot);
        if (i < I) {
            const float* wp = W + (size_t)ot * I + i;
#pragma unroll 4
            for (int oo = q; oo < n; oo += 4) {
                const float w = wp[(size_t)oo * I];
#pragma unroll
                for (int j = 0; j < LD_ROWS; ++j) s[j] += dys[j][oo] * w;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LD_ROWS; ++j) red[q][j][li] = s[j];
    __syncthreads();
    if (i < I)
        for (int j = q; j < LD_ROWS; j += 4)
            if (r0 + j < R)
                dx[(size_t)blockIdx.z * R * ldx + (size_t)(r0 + j) * ldx + i] =
                    (red[0][j][li] + red[1][j][li]) + (red[2][j][li] + red[3][j][li]);
}
__global__ void linear_dgrad_sum_kernel(const float* __restrict__ part, int shares, int64_t n, float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < shares; ++k) s += part[(int64_t)k * n + i];
    dx[i] = s;
}
static int linear_dgrad_shares(int I, int O) { return O >= 2048 ? 16 : 1; }
size_t linear_dgrad_ws_floats(int R, int I, int O) {
    const int sh = std::max(linear_dgrad_shares(I, O), rows_gemm_nn_shares(O));
    return sh > 1 ? (size_t)sh * R * I : 0;
}
// ws: linear_dgrad_ws_floats(R, I, O) floats (may be nullptr when that is 0, or to force the one-pass form)
int launch_linear_dgrad(const float* dy, int ldy, const float* W, float* dx, int ldx, int R, int I, int O, float* ws,
                        hipStream_t s) {
    static const bool no_mfma = std::getenv("DM_NO_SMALL_GEMM") != nullptr;  // A/B: the VALU kernels below
    if (!no_mfma && rows_gemm_nn_ok(R, I, O, ldy, ldx)) {  // batch rows as an MFMA GEMM (small_gemm.hip)
        const int nsh = rows_gemm_nn_shares(O);
        if (nsh == 1) return launch_rows_gemm_nn(dy, ldy, W, dx, ldx, R, I, O, s);
        if (ws && ldx == I) {
            if (launch_rows_gemm_nn(dy, ldy, W, ws, ldx, R, I, O, s)) return 1;
            const int64_t n = (int64_t)R * I;
            hipLaunchKernelGGL(linear_dgrad_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, nsh, n, dx);
            DM_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    const int sh = (ws && ldx == I) ? linear_dgrad_shares(I, O) : 1;
    const int o_share = (O + sh - 1) / sh;
    const dim3 grid((I + 63) / 64, (R + LD_ROWS - 1) / LD_ROWS, sh);
    hipLaunchKernelGGL(linear_dgrad_kernel, grid, dim3(256), 0, s, dy, ldy, W, sh > 1 ? ws : dx, ldx, R, I, O, o_share);
    DM_CHECK_HIP(hipGetLastError());
    if (sh > 1) {
        const int64_t n = (int64_t)R * I;
        hipLaunchKernelGGL(linear_dgrad_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, sh, n, dx);
        DM_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// q_sample (:813-821) and the loss (:864-878): per-sample mean of (out - target)^2 times loss_weight[t], mean over the
// batch; d(loss)/d(out) in the same pass.  Everything NCHW (the boundary layout).  coef[b] = (sqrt_alphas_cumprod[t_b],
// sqrt_one_minus_alphas_cumprod[t_b], loss_weight[t_b], 0), gathered on the host exactly as `extract` does.
// objective: 0 pred_noise (target = noise), 1 pred_x0 (x_start), 2 pred_v (a * noise - b * x_start, :582-586)
// ---------------------------------------------------------------------------------------
__global__ void q_sample_kernel(const float* __restrict__ x_start, const float* __restrict__ noise,
                                const float* __restrict__ coef, float* __restrict__ x, int per_sample, int64_t n) {
#pragma clang fp contract(off)  // two roundings, like the reference's tensor expression
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* c = coef + DM_TRAIN_COEFS * (i / per_sample);
    x[i] = c[0] * x_start[i] + c[1] * noise[i];
}
// offset noise (:830-834): noise += offset_noise_strength * offset[b][c] (two roundings, as the reference's expression)
__global__ void offset_noise_kernel(float* __restrict__ noise, const float* __restrict__ offset, float strength, int HW,
                                    int64_t n) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o = strength * offset[i / HW];
    noise[i] = noise[i] + o;
}
int launch_offset_noise(float* noise, const float* offset, float strength, int BC, int HW, hipStream_t s) {
    const int64_t n = (int64_t)BC * HW;
    hipLaunchKernelGGL(offset_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, noise, offset, strength, HW, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
// torch.cdist (p = 2) of two flattened batches: out[i][j] = sqrt(sum_d (x[i][d] - y[j][d])^2), one workgroup per pair
__global__ __launch_bounds__(256) void cdist_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                    float* __restrict__ out, int m, int64_t D) {
    __shared__ float red[256];
    const int i = blockIdx.y, j = blockIdx.x;
    const float* a = x + (size_t)i * D;
    const float* b = y + (size_t)j * D;
    float s = 0.f;
    for (int64_t d = threadIdx.x; d < D; d += 256) {
        const float v = a[d] - b[d];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)i * m + j] = sqrtf(red[0]);
}
int launch_cdist(const float* x, const float* y, float* out, int n, int m, int64_t D, hipStream_t s) {
    DM_REQUIRE(n <= 65535, "cdist: at most 65535 rows");
    hipLaunchKernelGGL(cdist_kernel, dim3(m, n), dim3(256), 0, s, x, y, out, m, D);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
// The elementwise helpers of DenoisingDiffusion (predict_start_from_noise / predict_noise_from_start / predict_v /
// predict_start_from_v / the posterior mean, DD/denoising_diffusion.py:570-601) with per-sample coefficients c = coef[b][0..1]:
//   mode 0: out = c0 * x + c1 * y        mode 1: out = (c0 * x - y) / c1        clamp: to [-1, 1] afterwards
// and the guide mix of ddim_sample_guided (:754): out = a * mask + b * (1 - mask).  Roundings as the reference's expressions.
__global__ void lincomb_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ coef,
                               float* __restrict__ out, int64_t per_sample, int64_t n, int mode, int clamp) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c0 = coef[2 * (i / per_sample)], c1 = coef[2 * (i / per_sample) + 1];
    float v = mode == 0 ? c0 * x[i] + c1 * y[i] : (c0 * x[i] - y[i]) / c1;
    if (clamp) v = fminf(fmaxf(v, -1.f), 1.f);
    out[i] = v;
}
int launch_lincomb(const float* x, const float* y, const float* coef_dev, float* out, int B, int64_t per_sample, int mode,
                   int clamp, hipStream_t s) {
    const int64_t n = (int64_t)B * per_sample;
    hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, coef_dev, out, per_sample, n, mode,
                       clamp);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
__global__ void mask_mix_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ mask,
                                float* __restrict__ out, int64_t n) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = a[i] * mask[i] + b[i] * (1.0f - mask[i]);
}
int launch_mask_mix(const float* a, const float* b, const float* mask, float* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(mask_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, mask, out, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_q_sample(const float* x_start, const float* noise, const float* coef_dev, float* x, int B, int per_sample,
                    hipStream_t s) {
    const int64_t n = (int64_t)B * per_sample;
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x_start, noise, coef_dev, x,
                       per_sample, n);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// pred_x_start of model_predictions (:603-626) with a per-sample timestep, unclipped (p_losses :849): coef[b][4..5] =
// sqrt_recip_alphas_cumprod[t_b], sqrt_recipm1_alphas_cumprod[t_b]; coef[b][0..1] feed predict_start_from_v
__global__ void pred_x_start_kernel(const float* __restrict__ x, const float* __restrict__ out, const float* __restrict__ coef,
                                    float* __restrict__ xs, int per_sample, int64_t n, int objective) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* c = coef + DM_TRAIN_COEFS * (i / per_sample);
    float v;
    if (objective == 0) v = c[4] * x[i] - c[5] * out[i];
    else if (objective == 1) v = out[i];
    else v = c[0] * x[i] - c[1] * out[i];
    xs[i] = v;
}
int launch_pred_x_start(const float* x, const float* out, const float* coef_dev, float* xs, int B, int per_sample, int objective,
                        hipStream_t s) {
    const int64_t n = (int64_t)B * per_sample;
    hipLaunchKernelGGL(pred_x_start_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, coef_dev, xs, per_sample,
                       n, objective);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// one workgroup per sample: part[b] = loss_weight * mean((out - target)^2); dout = 2 (out - target) * w / (per_sample * B).
// terms: bit 0 the weighted MSE, bit 1 the hybrid KL term of p_losses (:880-897), written as the reference writes it --
//   x0 = pred_x_start(out) clamped to [-1, 1] (p_mean_variance, clip_denoised = True: clamp_ passes the gradient inside the
//   closed interval);  model_mean = coef1 x0 + coef2 x;  posterior_mean = coef1 x_start + coef2 x (q_posterior, :594-601);
//   kl = 0.5 (plv - mlv + (exp(mlv) + (model_mean - posterior_mean)^2) / posterior_variance - 1), mlv = plv =
//   posterior_log_variance_clipped[t];  klpart[b] = mean(kl) * [t_b > 0];  loss += kl_weight * sum_b klpart[b] / (n_pos + 1e-8)
// -- including the division by posterior_variance[t] = 0 at t = 0 that the mask multiplies afterwards (inf * 0): a batch that
// holds a t = 0 sample has a NaN loss and NaN gradients in the reference, and here.
// coef[b]: [8] posterior_mean_coef1, [9] posterior_mean_coef2, [10] posterior_variance, [11] posterior_log_variance_clipped,
// [3] 1 if t_b > 0 else 0.  xq: the q_sample output the U-Net saw (its first C channels).
__global__ void mse_loss_kernel(const float* __restrict__ out, const float* __restrict__ x_start, const float* __restrict__ noise,
                                const float* __restrict__ xq, const float* __restrict__ coef, float* __restrict__ dout,
                                float* __restrict__ part, float* __restrict__ klpart, int per_sample, int B, int objective,
                                float loss_scale, int terms, float kl_scale) {
#pragma clang fp contract(off)  // the KL expression keeps the roundings of the reference's tensor expression
    __shared__ double red[256];
    __shared__ double red2[256];
    const int b = blockIdx.x;
    const float* c = coef + DM_TRAIN_COEFS * b;
    const float gscale = (terms & 1) ? loss_scale * 2.0f * c[2] / ((float)per_sample * (float)B) : 0.f;
    // d(loss) / d(kl element) = loss_scale * kl_weight * mask_b / (n_pos + 1e-8) / per_sample  (kl_scale = kl_weight / (n_pos + 1e-8))
    const float kscale = loss_scale * kl_scale * c[3] / (float)per_sample;
    const float emlv = (terms & 2) ? expf(c[11]) : 0.f;
    double s = 0.0, sk = 0.0;
    for (int i = threadIdx.x; i < per_sample; i += 256) {
        const size_t k = (size_t)b * per_sample + i;
        float tgt;
        if (objective == 0) tgt = noise[k];
        else if (objective == 1) tgt = x_start[k];
        else tgt = c[0] * noise[k] - c[1] * x_start[k];
        const float o = out[k];
        const float d = o - tgt;
        s += (double)d * d;
        float g = d * gscale;
        if (terms & 2) {
            const float x = xq[k];
            float x0, dx0;  // pred_x_start and its derivative w.r.t. the model output
            if (objective == 0) { x0 = c[4] * x - c[5] * o; dx0 = -c[5]; }
            else if (objective == 1) { x0 = o; dx0 = 1.0f; }
            else { x0 = c[0] * x - c[1] * o; dx0 = -c[1]; }
            const bool inside = x0 >= -1.0f && x0 <= 1.0f;
            const float x0c = fminf(fmaxf(x0, -1.0f), 1.0f);
            const float mm = c[8] * x0c + c[9] * x;
            const float pm = c[8] * x_start[k] + c[9] * x;
            const float diff = mm - pm;
            const float kl = 0.5f * ((c[11] - c[11]) + (emlv + diff * diff) / c[10] - 1.0f);
            sk += (double)kl;
            // 0.5 * 2 diff / pv * coef1 * [inside] * dx0, then the chain of means; formed in the reference's order so that
            // t = 0 (pv = 0, mask 0) yields 0 * inf = NaN as autograd does
            const float gk = kscale * (diff / c[10]) * (inside ? c[8] * dx0 : 0.0f);
            g = (terms & 1) ? g + gk : gk;
        }
        dout[k] = g;
    }
    red[threadIdx.x] = s;
    red2[threadIdx.x] = sk;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) {
            red[threadIdx.x] += red[threadIdx.x + m];
            red2[threadIdx.x] += red2[threadIdx.x + m];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[b] = (terms & 1) ? (float)(red[0] / per_sample) * c[2] : 0.f;
        if (terms & 2) klpart[b] = (float)(red2[0] / per_sample) * c[3];  // kl * mask, as the reference multiplies
    }
}
__global__ void mean_kernel(const float* __restrict__ part, const float* __restrict__ klpart, int B, float* __restrict__ loss,
                            float loss_scale, float kl_scale) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += part[b];
    float v = (float)(s / B);
    if (klpart) {
        float k = 0.f;
        for (int b = 0; b < B; ++b) k += klpart[b];
        v += kl_scale * k;  // loss.mean() of (loss_b + 0.001 * kl): the scalar joins every row
    }
    *loss = v * loss_scale;
}
// loss_scale: 1 / gradient_accumulate_every of Trainer.train (:1171) -- scales the reported loss and every gradient
int launch_mse_loss(const float* out, const float* x_start, const float* noise, const float* coef_dev, float* dout,
                    float* part, float* loss, int B, int per_sample, int objective, float loss_scale, hipStream_t s,
                    int terms, const float* xq, float* klpart, float kl_scale) {
    DM_REQUIRE(terms >= 1 && terms <= 3 && (!(terms & 2) || (xq && klpart)), "mse_loss: terms");
    hipLaunchKernelGGL(mse_loss_kernel, dim3(B), dim3(256), 0, s, out, x_start, noise, xq, coef_dev, dout, part, klpart,
                       per_sample, B, objective, loss_scale, terms, kl_scale);
    DM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1), 0, s, part, (terms & 2) ? klpart : nullptr, B, loss, loss_scale, kl_scale);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Optimiser step of the caller of record (Trainer: Adam(lr, betas = (0.9, 0.99)) after clip_grad_norm_(1.0), then
// ema.update(), DD/denoising_diffusion.py:1006,1180-1190).  One fused pass over a flat parameter tensor:
//   g = grad * clip;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr * (m / bc1) / (sqrt(v / bc2) + eps)
//   ema = ema * decay + p * (1 - decay)   (when ema != nullptr)
// ---------------------------------------------------------------------------------------
__global__ void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ part) {
    __shared__ double red[256];
    double s = 0.0;
    const int64_t n4 = n >> 2;  // 16-byte loads (the flat gradient buffer is 256-byte aligned), then the tail
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        s += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += (double)x[4 * n4 + threadIdx.x] * x[4 * n4 + threadIdx.x];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// total_norm = sqrt(sum); clip coefficient = min(1, max_norm / (total_norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
// One wave: lane l adds the partial sums l, l + 64, ... in order, the 64 lane sums meet in a fixed tree (a single thread
// walking 1024 dependent loads took 53 us).
__global__ __launch_bounds__(64) void clip_coef_kernel(const double* __restrict__ part, int nparts, float max_norm,
                                                       float* __restrict__ out2) {
    __shared__ double red[64];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 32; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        out2[0] = norm;
        out2[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
    }
}
int launch_grad_norm(const float* grads, int64_t n, double* part_ws /* 1024 doubles */, float max_norm, float* out2,
                     hipStream_t s) {
    const int nb = (int)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, grads, n, part_ws);
    DM_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, s, part_ws, nb, max_norm, out2);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
__global__ void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                float* __restrict__ ema, const float* __restrict__ clip2, int64_t n, float lr, float b1, float b2,
                                float eps, float bc1, float bc2, float ema_decay) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * (clip2 ? clip2[1] : 1.0f);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float pi = p[i] - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = pi;
    if (ema) ema[i] = ema[i] * ema_decay + pi * (1.0f - ema_decay);
}
// ema = ema * decay + p * (1 - decay)   (ema_pytorch: ema.lerp_(online, 1 - decay))
__global__ void lerp_kernel(float* __restrict__ ema, const float* __restrict__ p, int64_t n, float decay) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e = ema[i];
    ema[i] = e + (p[i] - e) * (1.0f - decay);
}
int launch_lerp(float* ema, const float* p, int64_t n, float decay, hipStream_t s) {
    hipLaunchKernelGGL(lerp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ema, p, n, decay);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_adam_ema(float* p, const float* g, float* m, float* v, float* ema, const float* clip2, int64_t n, float lr,
                    float b1, float b2, float eps, int step, float ema_decay, hipStream_t s) {
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, ema, clip2, n, lr, b1,
                       b2, eps, bc1, bc2, ema_decay);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dm

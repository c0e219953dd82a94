// The nn.Linear layers of the time embedding in TRAINING mode (one row per image): the concatenated ResnetBlock.mlp matrix
// ([8064][256] at dim 64: DD/denoising_diffusion.py:127-130), time_mlp (:280-285) and the text projections -- forward, input
// gradient and weight gradient as small GEMMs on v_mfma_f32_16x16x4_f32.  M = batch rows (16 .. 256), so the matrices are
// tiny beside the convolutions, but as wave-per-output VALU kernels they cost 0.2 ms of an 8 ms iteration; here every
// operand is fetched with 16-byte loads and every product is on the matrix pipe.
//   rows_gemm_nt   y[r][o]  = b[o] + sum_i x[r][i] W[o][i]               (Linear.forward, W as stored: [O][I])
//   rows_gemm_nn   dx[r][i] = sum_o dy[r][o] W[o][i]                      (input gradient; K = O split over workgroups)
//   rows_gemm_tn   dW[o][i] (+)= sum_r dy[r][o] x[r][i],  db[o] (+)= sum_r dy[r][o]   (weight / bias gradient)
// Operand trick shared with pw_mfma.hip: a lane's 16-byte load is four values along the tensor's contiguous axis; used as
// the MFMA's K axis it is four K steps, used as the M / N axis it is four interleaved tiles -- either way one load, no LDS
// transposition.  Summation order is fixed (deterministic); it differs from the VALU kernels' (fp32 rounding only).
#include "conv_device.h"

#include <algorithm>

namespace dm {

static constexpr int SG_D = 4;  // K chunks in flight per wave

// ---- NT: workgroup = 16 outputs x 64 rows; its 4 waves take the 16-wide K chunks c = wave, wave + 4, ... (at I = 256 that is
// four chunks per wave, all in flight at once: these launches have few workgroups and nothing to overlap with, so their time
// is the number of load round trips of one wave -- 16 in a row cost 35 us for the 256 x 256 time MLP, one costs 6), the partial
// tiles meet in LDS and wave w finishes row tile w (fixed order over the waves).  grid (ceil(O / 16), ceil(R / 64)).
__global__ __launch_bounds__(256) void rows_gemm_nt_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ y, int ldy, int R,
                                                           int I, int O) {
    __shared__ f32x4 part[4][4][64];  // [wave][row tile][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int o0 = blockIdx.x * 16;
    const int r0 = blockIdx.y * 64;
    const int nrt = min(4, (R - r0 + 15) / 16);
    const float* wp = W + (size_t)min(o0 + l15, O - 1) * I + 4 * kq;
    const float* xp[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) xp[rt] = x + (size_t)min(r0 + 16 * rt + l15, R - 1) * ldx + 4 * kq;
    const f32x4 z4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[4] = {z4, z4, z4, z4};
    f32x4 b[SG_D], a[SG_D][4];
    const int nc = I / 16;
    auto load = [&](int c, int d) {
        b[d] = *reinterpret_cast<const f32x4*>(wp + 16 * c);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
            if (rt < nrt) a[d][rt] = *reinterpret_cast<const f32x4*>(xp[rt] + 16 * c);
    };
#pragma unroll
    for (int d = 0; d < SG_D; ++d) load(min(wave + 4 * d, nc - 1), d);
    for (int c = wave; c < nc; c += 4 * SG_D) {
#pragma unroll
        for (int d = 0; d < SG_D; ++d) {
            if (c + 4 * d < nc) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
                        if (rt < nrt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][rt][j], b[d][j], acc[rt], 0, 0, 0);
                if (c + 4 * (d + SG_D) < nc) load(c + 4 * (d + SG_D), d);
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) part[wave][rt][lane] = acc[rt];
    __syncthreads();
    const int o = o0 + l15, rt = wave;
    if (o >= O || rt >= nrt) return;
    const f32x4 sum = ((part[0][rt][lane] + part[1][rt][lane]) + part[2][rt][lane]) + part[3][rt][lane];
    const float bv = bias ? bias[o] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = r0 + 16 * rt + 4 * kq + e;  // D layout: register e of lane (n = l15, kq) = row 4 kq + e, column n
        if (r < R) y[(size_t)r * ldy + o] = sum[e] + bv;
    }
}
bool rows_gemm_nt_ok(int R, int I, int O, int ldx) { return R >= 16 && I % 16 == 0 && O % 16 == 0 && ldx % 4 == 0; }
int launch_rows_gemm_nt(const float* x, int ldx, const float* W, const float* bias, float* y, int ldy, int R, int I, int O,
                        hipStream_t s) {
    DM_REQUIRE(rows_gemm_nt_ok(R, I, O, ldx), "rows_gemm_nt: shape");
    hipLaunchKernelGGL(rows_gemm_nt_kernel, dim3((O + 15) / 16, (R + 63) / 64), dim3(256), 0, s, x, ldx, W, bias, y, ldy, R, I, O);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- NN: dx[r][i] = sum_o dy[r][o] W[o][i].  Wave = 16 rows x 64 columns i (four interleaved 16-column sets: the lane's
// 16 bytes of a W row), workgroup = 4 waves = the four 64-column tiles of I = 256 (or fewer), one 16-row tile, one share
// of the O range; grid (ceil(I / 256), ceil(R / 16), shares).  A lane's 16 bytes of dy are four K steps, each meeting its
// own W row.  shares > 1: partial sums [share][R][ldx] for linear_dgrad_sum_kernel.
static constexpr int SG_NN_CH = 16;  // 16-o chunks per share
__global__ __launch_bounds__(256) void rows_gemm_nn_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ W,
                                                           float* __restrict__ dx, int ldx, int R, int I, int O) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int i0 = (blockIdx.x * 4 + wave) * 64;
    if (i0 >= I) return;
    const int r0 = blockIdx.y * 16;
    const int cb = blockIdx.z * SG_NN_CH, ce = min(cb + SG_NN_CH, O / 16);
    const float* ap = dy + (size_t)min(r0 + l15, R - 1) * ldy + 4 * kq;
    const float* wp = W + (size_t)(4 * kq) * I + i0 + 4 * l15;
    const f32x4 z4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[4] = {z4, z4, z4, z4};  // [column set j]: D register e of lane (n = l15, kq) = row 4 kq + e, column i0 + 4 n + j
    f32x4 a[SG_D], b[SG_D][4];
    auto load = [&](int c, int d) {
        a[d] = *reinterpret_cast<const f32x4*>(ap + 16 * c);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[d][t] = *reinterpret_cast<const f32x4*>(wp + (size_t)(16 * c + t) * I);
    };
#pragma unroll
    for (int d = 0; d < SG_D; ++d) load(min(cb + d, ce - 1), d);
    for (int c = cb; c < ce; c += SG_D) {
#pragma unroll
        for (int d = 0; d < SG_D; ++d) {
            if (c + d < ce) {
#pragma unroll
                for (int t = 0; t < 4; ++t)   // K step t: rows o = 16 c + 4 kq + t of W, component t of the dy load
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][t], b[d][t][j], acc[j], 0, 0, 0);
                if (c + d + SG_D < ce) load(c + d + SG_D, d);
            }
        }
    }
    float* out = dx + (size_t)blockIdx.z * R * ldx;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = r0 + 4 * kq + e;
        if (r < R) *reinterpret_cast<f32x4*>(out + (size_t)r * ldx + i0 + 4 * l15) = make_f32x4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
    }
}
bool rows_gemm_nn_ok(int R, int I, int O, int ldy, int ldx) {
    return R >= 16 && I % 64 == 0 && O % 16 == 0 && ldy % 4 == 0 && ldx % 4 == 0;
}
int rows_gemm_nn_shares(int O) { return (O / 16 + SG_NN_CH - 1) / SG_NN_CH; }
// ws: shares * R * ldx floats when shares > 1 (the caller sums the shares), else unused
int launch_rows_gemm_nn(const float* dy, int ldy, const float* W, float* dx_or_ws, int ldx, int R, int I, int O, hipStream_t s) {
    DM_REQUIRE(rows_gemm_nn_ok(R, I, O, ldy, ldx), "rows_gemm_nn: shape");
    hipLaunchKernelGGL(rows_gemm_nn_kernel, dim3((I + 255) / 256, (R + 15) / 16, rows_gemm_nn_shares(O)), dim3(256), 0, s, dy, ldy, W,
                       dx_or_ws, ldx, R, I, O);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- TN: dW[o][i] (+)= sum_r dy[r][o] x[r][i];  db[o] (+)= sum_r dy[r][o] (the waves of column tile 0).
// Wave = 64 outputs o (four interleaved 16-row sets: the lane's 16 bytes of a dy row) x 64 columns i (likewise of an x row);
// the K axis is the batch row r, four rows (kq) per MFMA.  Row o of dW is dw_rows[o] (the 19 ResnetBlock.mlp gradients live in
// 19 tensors) or dw + o * ldw.  grid (ceil(I / 64), ceil(O / 256)), workgroup = 4 waves = 4 output tiles.
__global__ __launch_bounds__(256) void rows_gemm_tn_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                           int ldx, float* const* __restrict__ dw_rows,
                                                           float* const* __restrict__ db_rows, float* __restrict__ dw, int ldw,
                                                           float* __restrict__ db, int R, int I, int O, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int o0 = (blockIdx.y * 4 + wave) * 64, i0 = blockIdx.x * 64;
    if (o0 >= O) return;
    const float* ap = dy + (size_t)kq * ldy + o0 + 4 * l15;
    const float* bp = x + (size_t)kq * ldx + i0 + 4 * l15;
    const f32x4 z4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[4][4];  // [row set s][column set j]: register e of lane (n = l15, kq) = o0 + 4 (4 kq + e) + s, i0 + 4 n + j
#pragma unroll
    for (int sgi = 0; sgi < 4; ++sgi)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[sgi][j] = z4;
    f32x4 colsum = z4;
    const int nk = (R + 3) / 4;
    constexpr int TD = 8;  // 8 of the batch's 4-row steps in flight: B = 64 is two load rounds
    f32x4 a[TD], b[TD];
    auto load = [&](int k, int d) {
        const bool ok = 4 * k + kq < R;
        const int kk = ok ? k : 0;
        a[d] = *reinterpret_cast<const f32x4*>(ap + (size_t)(4 * kk) * ldy);
        b[d] = *reinterpret_cast<const f32x4*>(bp + (size_t)(4 * kk) * ldx);
        if (!ok) a[d] = z4;  // rows past the batch contribute nothing
    };
#pragma unroll
    for (int d = 0; d < TD; ++d) load(min(d, nk - 1), d);
    for (int k = 0; k < nk; k += TD) {
#pragma unroll
        for (int d = 0; d < TD; ++d) {
            if (k + d < nk) {
                colsum += a[d];
#pragma unroll
                for (int sgi = 0; sgi < 4; ++sgi)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[sgi][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][sgi], b[d][j], acc[sgi][j], 0, 0, 0);
                if (k + d + TD < nk) load(k + d + TD, d);
            }
        }
    }
#pragma unroll
    for (int sgi = 0; sgi < 4; ++sgi)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = o0 + 4 * (4 * kq + e) + sgi;
            if (o >= O) continue;
            float* row = (dw_rows ? dw_rows[o] : dw + (size_t)o * ldw) + i0 + 4 * l15;
            f32x4 v = make_f32x4(acc[sgi][0][e], acc[sgi][1][e], acc[sgi][2][e], acc[sgi][3][e]);
            if (accumulate) v += *reinterpret_cast<const f32x4*>(row);
            *reinterpret_cast<f32x4*>(row) = v;
        }
    if (blockIdx.x == 0 && (db_rows || db)) {
        // the lane's partial column sums cover rows r = kq (mod 4): add the four kq lanes of a column group (fixed order)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = colsum[c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int o = o0 + 4 * l15 + c;
            if (kq == 0 && o < O) {
                float* q = db_rows ? db_rows[o] : db + o;
                *q = accumulate ? *q + v : v;
            }
        }
    }
}
bool rows_gemm_tn_ok(int R, int I, int O, int ldy, int ldx) {
    return R >= 4 && I % 64 == 0 && O % 64 == 0 && ldy % 4 == 0 && ldx % 4 == 0;
}
int launch_rows_gemm_tn(const float* dy, int ldy, const float* x, int ldx, float* const* dw_rows, float* const* db_rows, float* dw,
                        int ldw, float* db, int R, int I, int O, int accumulate, hipStream_t s) {
    DM_REQUIRE(rows_gemm_tn_ok(R, I, O, ldy, ldx) && (dw_rows || (dw && ldw % 4 == 0)), "rows_gemm_tn: shape");
    hipLaunchKernelGGL(rows_gemm_tn_kernel, dim3(I / 64, (O + 255) / 256), dim3(256), 0, s, dy, ldy, x, ldx, dw_rows, db_rows, dw, ldw,
                       db, R, I, O, accumulate);
    DM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace dm

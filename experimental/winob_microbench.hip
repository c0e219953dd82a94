// Development microbenchmark for the "Winograd F(2x2,3x3) with fp32 products on the bf16 matrix cores" idea (DESIGN.md,
// next levers): the MAIN LOOP only of the kernel that note budgets -- no convolution plumbing, no epilogue -- to measure
// what fraction of the six-product bf16 MFMA rate survives the operand traffic and the transform / split VALU work.
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/winob_microbench.hip -o tools/microbench/bin/winob_mb && tools/microbench/bin/winob_mb
//
// Shape of the loop (one workgroup per CU, 4 waves, wave i = row i of the 4x4 transformed patch, 64 tiles x 64 couts):
//   per chunk of 16 input channels and per wave: 4 patch columns j x 2 tile groups r x 2 cout groups q x 6 products
//   = 96 v_mfma_f32_32x32x16_bf16; the B planes (host-split weights, [chunk][xi][plane][cout][16] bf16) come straight from
//   global memory (L2-resident, the same stream for every workgroup, 24 KB per wave and chunk); the A planes are formed per
//   lane from the raw 18x18x16 window in LDS: T = d[ra] +- d[rb] (32 ds_read_b128), V[j] from T, three-way bf16 split.
// MODE 0: MFMAs only (operands loaded once)          -> the issue ceiling of this accumulator layout
// MODE 1: + B planes from global memory every chunk
// MODE 2: + A planes from LDS (reads, transform, split) every chunk
// MODE 3: + window staging (global loads of the next window, ds_write_b128, one barrier per chunk)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int RS = 18 * 16 + 8;     // floats per window row
constexpr int WIN = 18 * RS;        // floats per window buffer
constexpr int HR = 6;               // 16-byte staging items per thread (18 x 18 pixels x 4 quads = 1296 <= 6 x 256)

struct Split3 {
    bf16x8 h, m, l;
};
__device__ __forceinline__ Split3 split3(const f32x4& lo, const f32x4& hi) {
    Split3 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? lo[i] : hi[i - 4];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        s.h[i] = h;
        s.m[i] = m;
        s.l[i] = (__bf16)(r1 - (float)m);
    }
    return s;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void winob_rate(const bf16x8* __restrict__ wplanes, const float* __restrict__ xin,
                                                     float* __restrict__ out, int n_chunks, int reps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * WIN; i += 256) sm[i] = (float)((i * 29 + blockIdx.x) & 127) * 0.01f - 0.6f;
    __syncthreads();
    // transform row of this wave: B^T row i = d[ra] + sgn * d[rb]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const int tx = l31 & 7, ty = l31 >> 3;
    int colq[4][2];  // float offset of (column b, quad 2 lh + k) of this lane's tile inside a window row
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int hx = 2 * tx + b;
            colq[b][k] = hx * 16 + 4 * ((2 * lh + k) ^ ((hx >> 2) & 3));
        }
    int hoff[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int it = tid + 256 * i;
        const int px = min(it >> 2, 18 * 18 - 1), qd = it & 3;
        const int hy = px / 18, hx = px - hy * 18;
        hoff[i] = hy * RS + hx * 16 + 4 * (qd ^ ((hx >> 2) & 3));
    }
    f32x16 acc[4][2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][r][q][e] = 0.f;

    // B planes of (chunk, xi = 4 wave + j, plane, q): lane reads 16 bytes at cout q*32 + l31, channels 8 lh ..
    auto bptr = [&](int chunk, int j, int pl, int q) {
        return wplanes + ((((size_t)chunk * 16 + 4 * wave + j) * 3 + pl) * 64 + q * 32 + l31) * 2 + lh;
    };
    bf16x8 Bp[4][3][2];
    Split3 Ap[4][2];
    auto load_b = [&](int chunk, int j) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < 2; ++q) Bp[j][pl][q] = *bptr(chunk, j, pl, q);
    };
    auto make_a = [&](const float* win) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float* base = win + (2 * (ty + 4 * r)) * RS;
            f32x4 T[4][2];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f32x4 da = *reinterpret_cast<const f32x4*>(base + ra * RS + colq[b][k]);
                    const f32x4 db = *reinterpret_cast<const f32x4*>(base + rb * RS + colq[b][k]);
                    T[b][k] = da + sgn * db;
                }
            Ap[0][r] = split3(T[0][0] - T[2][0], T[0][1] - T[2][1]);
            Ap[1][r] = split3(T[1][0] + T[2][0], T[1][1] + T[2][1]);
            Ap[2][r] = split3(T[2][0] - T[1][0], T[2][1] - T[1][1]);
            Ap[3][r] = split3(T[1][0] - T[3][0], T[1][1] - T[3][1]);
        }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b(0, j);
    make_a(sm);
    f32x4 hreg[HR];
    for (int rep = 0; rep < reps; ++rep) {
        for (int c = 0; c < n_chunks; ++c) {
            const float* wnext = sm + ((c + 1) & 1) * WIN;
            float* wstore = sm + (c & 1) * WIN;
            if constexpr (MODE >= 3) {
#pragma unroll
                for (int i = 0; i < HR; ++i)
                    hreg[i] = *reinterpret_cast<const f32x4*>(xin + ((size_t)(blockIdx.x * 8 + (c & 7)) * HR * 256 + 256 * i + tid) * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // six products, small terms first; the four accumulators of a column take turns
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const bf16x8 a = pr == 0 || pr == 3 || pr == 5 ? Ap[j][r].h : (pr == 1 ? Ap[j][r].l : Ap[j][r].m);
                            const bf16x8 b = pr == 0 ? Bp[j][2][q] : (pr == 1 || pr == 4 || pr == 5 ? Bp[j][0][q] : Bp[j][1][q]);
                            acc[j][r][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j][r][q], 0, 0, 0);
                        }
                if constexpr (MODE >= 1) load_b((c + 1) % n_chunks, j);  // this column's planes for the next chunk
            }
            if constexpr (MODE >= 3) {
#pragma unroll
                for (int i = 0; i < HR; ++i) *reinterpret_cast<f32x4*>(wstore + hoff[i]) = hreg[i];
            }
            if constexpr (MODE >= 2) make_a(wnext);
            if constexpr (MODE >= 3) __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[j][r][q][e];
    out[blockIdx.x * 256 + tid] = s;
}

// MODE 4 of the experiment: the same work as MODE 3, software-pipelined by hand.  Phase (c, j) = the 24 MFMAs of patch
// column j of chunk c; between them, pinned by sched_barrier:
//   * the six 16-byte loads of the B planes of the NEXT phase (two register sets, 48 registers instead of 96),
//   * the A planes of the NEXT phase from T (combine + split, one channel pair per hook; two sets, 48 registers),
//   * T of the next chunk from the LDS window (tile group 0 in phase 2, group 1 in phase 3, into the registers the
//     current T no longer needs), the global loads of the window after that (phase 0) and its LDS stores (phase 1).
// One barrier per chunk.
struct Pair3 {
    unsigned h, m, l;  // two bf16 each
};
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ Pair3 split_pair(float x0, float x1) {
    Pair3 o;
    o.h = pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, o.h << 16), r1 = x1 - __builtin_bit_cast(float, o.h & 0xffff0000u);
    o.m = pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, o.m << 16), s1 = r1 - __builtin_bit_cast(float, o.m & 0xffff0000u);
    o.l = pk_bf16(s0, s1);
    return o;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ABL>  // ablation bits: 1 = no B loads, 2 = no A production, 4 = no window staging, 8 = no barrier
__global__ __launch_bounds__(256, 1) void winob_pipe(const bf16x8* __restrict__ wplanes, const float* __restrict__ xin,
                                                     float* __restrict__ out, int n_chunks, int reps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * WIN; i += 256) sm[i] = (float)((i * 29 + blockIdx.x) & 127) * 0.01f - 0.6f;
    __syncthreads();
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const int tx = l31 & 7, ty = l31 >> 3;
    int colq[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int hx = 2 * tx + b;
            colq[b][k] = hx * 16 + 4 * ((2 * lh + k) ^ ((hx >> 2) & 3));
        }
    int hoff[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int it = tid + 256 * i;
        const int px = min(it >> 2, 18 * 18 - 1), qd = it & 3;
        const int hy = px / 18, hx = px - hy * 18;
        hoff[i] = hy * RS + hx * 16 + 4 * (qd ^ ((hx >> 2) & 3));
    }
    f32x16 acc[4][2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][r][q][e] = 0.f;
    u32x4 Ah[2][2], Am[2][2], Al[2][2];  // [set][r]: planes of the A operand (8 bf16 each)
    bf16x8 Bp[2][3][2];                  // [set][plane][q]
    f32x4 T[2][4][2];                    // [r][column b][channel quad k]
    f32x4 hreg[HR];
    auto bptr = [&](int chunk, int j, int pl, int q) {
        return wplanes + ((((size_t)chunk * 16 + 4 * wave + j) * 3 + pl) * 64 + q * 32 + l31) * 2 + lh;
    };
    // the same planes through a buffer resource: per-lane byte offset once, everything else a scalar offset
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(wplanes), 0, (int)((size_t)n_chunks * 16 * 3 * 64 * 32), 0x00020000);
    const unsigned bvo = (unsigned)(l31 * 2 + lh) * 16;
    const unsigned bwave = __builtin_amdgcn_readfirstlane(wave) * 4 * 3 * 64 * 32;
    auto bload = [&](int chunk, int j, int pl, int q) {
        const unsigned so = (unsigned)chunk * (16 * 3 * 64 * 32) + bwave + (unsigned)((j * 3 + pl) * 64 + q * 32) * 32;
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)bvo, (int)so, 0));
    };
    // V[jn] (8 channels: quad halves k = 0, 1) of tile group r from T
    auto vval = [&](int jn, int r, int k) -> f32x4 {
        return jn == 0 ? T[r][0][k] - T[r][2][k] : (jn == 1 ? T[r][1][k] + T[r][2][k] : (jn == 2 ? T[r][2][k] - T[r][1][k] : T[r][1][k] - T[r][3][k]));
    };
    // one channel pair (k, half h2) of A[set][r] for column jn, in three stages of ~4 instructions (one per MFMA gap):
    // stage 0: combine + high plane, 1: middle plane, 2: low plane; the residuals wait in rs0 / rs1 in between
    float rs0 = 0.f, rs1 = 0.f;
    auto a_stage = [&](int set, int jn, int r, int k, int h2, int st) {
        if (st == 0) {
            const int e0 = 2 * h2, e1 = 2 * h2 + 1;
            const int ca = jn == 0 ? 0 : (jn == 2 ? 2 : 1), cb = jn == 0 ? 2 : (jn == 1 ? 2 : (jn == 2 ? 1 : 3));
            const float x0 = jn == 1 ? T[r][ca][k][e0] + T[r][cb][k][e0] : T[r][ca][k][e0] - T[r][cb][k][e0];
            const float x1 = jn == 1 ? T[r][ca][k][e1] + T[r][cb][k][e1] : T[r][ca][k][e1] - T[r][cb][k][e1];
            const unsigned h = pk_bf16(x0, x1);
            Ah[set][r][2 * k + h2] = h;
            rs0 = x0 - __builtin_bit_cast(float, h << 16);
            rs1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
        } else if (st == 1) {
            const unsigned mm = pk_bf16(rs0, rs1);
            Am[set][r][2 * k + h2] = mm;
            rs0 = rs0 - __builtin_bit_cast(float, mm << 16);
            rs1 = rs1 - __builtin_bit_cast(float, mm & 0xffff0000u);
        } else {
            Al[set][r][2 * k + h2] = pk_bf16(rs0, rs1);
        }
    };
    // micro-op u = 0..11 of the four pairs of tile group r
    auto a_micro = [&](int set, int jn, int r, int u) { a_stage(set, jn, r, (u / 3) >> 1, (u / 3) & 1, u % 3); };
    auto a_pair = [&](int set, int jn, int r, int k, int h2) {
        a_stage(set, jn, r, k, h2, 0);
        a_stage(set, jn, r, k, h2, 1);
        a_stage(set, jn, r, k, h2, 2);
    };
    auto t_item = [&](const float* win, int r, int b, int k) {
        const float* base = win + (2 * (ty + 4 * r)) * RS;
        const f32x4 da = *reinterpret_cast<const f32x4*>(base + ra * RS + colq[b][k]);
        const f32x4 db = *reinterpret_cast<const f32x4*>(base + rb * RS + colq[b][k]);
        T[r][b][k] = da + sgn * db;
    };
    // the same in two steps, LAT gaps apart: the reads land in T[r][b][k] (row a) and tq[slot] (row b)
    constexpr int LAT = 3;
    f32x4 tq[LAT];
    auto t_issue = [&](const float* win, int r, int i) {  // item i = 2 b + k
        const float* base = win + (2 * (ty + 4 * r)) * RS;
        T[r][i >> 1][i & 1] = *reinterpret_cast<const f32x4*>(base + ra * RS + colq[i >> 1][i & 1]);
        tq[i % LAT] = *reinterpret_cast<const f32x4*>(base + rb * RS + colq[i >> 1][i & 1]);
    };
    // scalar math on purpose: a packed f32 instruction beside MFMAs costs several times its issue slot (MI355X_MICROARCH.md,
    // constants table); build this file with -fno-slp-vectorize
    auto t_finish = [&](int r, int i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) T[r][i >> 1][i & 1][e] = __builtin_fmaf(sgn, tq[i % LAT][e], T[r][i >> 1][i & 1][e]);
    };
    // prologue: T of chunk 0, A / B of phase (0, 0)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = 0; k < 2; ++k) t_item(sm, r, b, k);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) a_pair(0, 0, r, k, h2);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < 2; ++q) Bp[0][pl][q] = *bptr(0, 0, pl, q);

    for (int rep = 0; rep < reps; ++rep) {
        for (int c = 0; c < n_chunks; ++c) {
            const int cn = c + 1 < n_chunks ? c + 1 : 0;
            const float* wnext = sm + ((c + 1) & 1) * WIN;
            float* wstore = sm + (c & 1) * WIN;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int set = j & 1, nset = set ^ 1, jn = (j + 1) & 3;
#pragma unroll
                for (int m = 0; m < 24; ++m) {
                    const int pr = m >> 2, r = (m >> 1) & 1, q = m & 1;
                    const u32x4 au = pr == 0 || pr == 3 || pr == 5 ? Ah[set][r] : (pr == 1 ? Al[set][r] : Am[set][r]);
                    const bf16x8 b = pr == 0 ? Bp[set][2][q] : (pr == 1 || pr == 4 || pr == 5 ? Bp[set][0][q] : Bp[set][1][q]);
                    acc[j][r][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au), b, acc[j][r][q], 0, 0, 0);
                    // ---- hooks
                    // at the head of the phase: spread over the phase (one load per four gaps) they land too late for the next
                    // phase and the loop is 19 % slower
                    if (m < 6 && !(ABL & 1)) Bp[nset][m >> 1][m & 1] = bload(j == 3 ? cn : c, jn, m >> 1, m & 1);
                    if (ABL & 2) {
                        if (!(ABL & 4)) {
                            if (j == 0 && m >= 12 && m < 12 + HR)
                                hreg[m - 12] = *reinterpret_cast<const f32x4*>(xin + ((size_t)(blockIdx.x * 8 + (c & 7)) * HR * 256 + 256 * (m - 12) + tid) * 4);
                            if (j == 1 && m >= 12 && m < 12 + HR) *reinterpret_cast<f32x4*>(wstore + hoff[m - 12]) = hreg[m - 12];
                        }
                    } else if (j < 2) {
                        // A of the next column from the current T: 24 micro-ops, one per gap
                        a_micro(nset, jn, m / 12, m % 12);
                        if (j == 0 && m >= 12 && m < 12 + HR && !(ABL & 4))
                            hreg[m - 12] = *reinterpret_cast<const f32x4*>(xin + ((size_t)(blockIdx.x * 8 + (c & 7)) * HR * 256 + 256 * (m - 12) + tid) * 4);
                        if (j == 1 && m >= 12 && m < 12 + HR && !(ABL & 4)) *reinterpret_cast<f32x4*>(wstore + hoff[m - 12]) = hreg[m - 12];
                    } else if (j == 2) {
                        // A[3] of this chunk (group 0, then group 1); T of the next chunk for group 0 once A[3] group 0 is done
                        a_micro(nset, 3, m / 12, m % 12);
                        if (m >= 12 && m < 20) t_issue(wnext, 0, m - 12);
                        if (m >= 12 + LAT && m < 20 + LAT) t_finish(0, m - 12 - LAT);
                    } else {
                        // T of the next chunk for group 1 beside A[0] of the next chunk (group 0 first)
                        if (m < 8) t_issue(wnext, 1, m);
                        if (m >= LAT && m < 8 + LAT) t_finish(1, m - LAT);
                        a_micro(nset, 0, m / 12, m % 12);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!(ABL & 8)) __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[j][r][q][e];
    if (ABL & 4) s += hoff[0] * 1e-30f;
    out[blockIdx.x * 256 + tid] = s;
}

// The 32-tile form: two workgroups per CU (two waves per SIMD, 256 registers each), so one wave's vector instructions issue
// while the other's MFMAs run and a workgroup's fixed phases overlap its neighbour's loop; the price is twice the B traffic
// per MFMA (a weight plane serves 32 tiles).  One B register set (loads issued right after the MFMAs that used the
// registers), T of the next chunk lands column by column as the current columns die.
constexpr int WIN1 = 10 * RS;  // 18 x 10 window
__global__ __launch_bounds__(256, 2) void winob_pipe1(const bf16x8* __restrict__ wplanes, const float* __restrict__ xin,
                                                      float* __restrict__ out, int n_chunks, int reps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * WIN1; i += 256) sm[i] = (float)((i * 29 + blockIdx.x) & 127) * 0.01f - 0.6f;
    __syncthreads();
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const int tx = l31 & 7, ty = l31 >> 3;
    int colq[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int hx = 2 * tx + b;
            colq[b][k] = hx * 16 + 4 * ((2 * lh + k) ^ ((hx >> 2) & 3));
        }
    constexpr int HR1 = 3;  // 18 x 10 x 4 = 720 items
    int hoff[HR1];
#pragma unroll
    for (int i = 0; i < HR1; ++i) {
        const int it = tid + 256 * i;
        const int px = min(it >> 2, 18 * 10 - 1), qd = it & 3;
        const int hy = px / 18, hx = px - hy * 18;
        hoff[i] = hy * RS + hx * 16 + 4 * (qd ^ ((hx >> 2) & 3));
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][q][e] = 0.f;
    u32x4 Ah[2], Am[2], Al[2];  // [set]
    bf16x8 Bp[2][3][2];         // [set][plane][q]
    f32x4 T[4][2];              // [column b][channel quad k]
    f32x4 hreg[HR1];
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(wplanes), 0, (int)((size_t)n_chunks * 16 * 3 * 64 * 32), 0x00020000);
    const unsigned bvo = (unsigned)(l31 * 2 + lh) * 16;
    const unsigned bwave = __builtin_amdgcn_readfirstlane(wave) * 4 * 3 * 64 * 32;
    auto bload = [&](int chunk, int j, int pl, int q) {
        const unsigned so = (unsigned)chunk * (16 * 3 * 64 * 32) + bwave + (unsigned)((j * 3 + pl) * 64 + q * 32) * 32;
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)bvo, (int)so, 0));
    };
    float rs0 = 0.f, rs1 = 0.f;
    auto a_stage = [&](int set, int jn, int k, int h2, int st) {
        if (st == 0) {
            const int e0 = 2 * h2, e1 = 2 * h2 + 1;
            const int ca = jn == 0 ? 0 : (jn == 2 ? 2 : 1), cb = jn == 0 ? 2 : (jn == 1 ? 2 : (jn == 2 ? 1 : 3));
            const float x0 = jn == 1 ? T[ca][k][e0] + T[cb][k][e0] : T[ca][k][e0] - T[cb][k][e0];
            const float x1 = jn == 1 ? T[ca][k][e1] + T[cb][k][e1] : T[ca][k][e1] - T[cb][k][e1];
            const unsigned h = pk_bf16(x0, x1);
            Ah[set][2 * k + h2] = h;
            rs0 = x0 - __builtin_bit_cast(float, h << 16);
            rs1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
        } else if (st == 1) {
            const unsigned mm = pk_bf16(rs0, rs1);
            Am[set][2 * k + h2] = mm;
            rs0 = rs0 - __builtin_bit_cast(float, mm << 16);
            rs1 = rs1 - __builtin_bit_cast(float, mm & 0xffff0000u);
        } else {
            Al[set][2 * k + h2] = pk_bf16(rs0, rs1);
        }
    };
    auto a_micro = [&](int set, int jn, int u) { a_stage(set, jn, (u / 3) >> 1, (u / 3) & 1, u % 3); };
    constexpr int LAT = 2;
    f32x4 tq[LAT];
    auto t_issue = [&](const float* win, int b, int k, int slot) {
        const float* base = win + (2 * ty) * RS;
        T[b][k] = *reinterpret_cast<const f32x4*>(base + ra * RS + colq[b][k]);
        tq[slot] = *reinterpret_cast<const f32x4*>(base + rb * RS + colq[b][k]);
    };
    auto t_finish = [&](int b, int k, int slot) {
#pragma unroll
        for (int e = 0; e < 4; ++e) T[b][k][e] = __builtin_fmaf(sgn, tq[slot][e], T[b][k][e]);
    };
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            t_issue(sm, b, k, 0);
            t_finish(b, k, 0);
        }
#pragma unroll
    for (int u = 0; u < 12; ++u) a_micro(0, 0, u);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int q = 0; q < 2; ++q) Bp[0][pl][q] = bload(0, 0, pl, q);

    for (int rep = 0; rep < reps; ++rep) {
        for (int c = 0; c < n_chunks; ++c) {
            const int cn = c + 1 < n_chunks ? c + 1 : 0;
            const float* wnext = sm + ((c + 1) & 1) * WIN1;
            float* wstore = sm + (c & 1) * WIN1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int set = j & 1, nset = set ^ 1, jn = (j + 1) & 3;
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int pr = m >> 1, q = m & 1;
                    const u32x4 au = pr == 0 || pr == 3 || pr == 5 ? Ah[set] : (pr == 1 ? Al[set] : Am[set]);
                    const bf16x8 b = pr == 0 ? Bp[set][2][q] : (pr == 1 || pr == 4 || pr == 5 ? Bp[set][0][q] : Bp[set][1][q]);
                    acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au), b, acc[j][q], 0, 0, 0);
                    if (m < 6) Bp[nset][m >> 1][m & 1] = bload(j == 3 ? cn : c, jn, m >> 1, m & 1);
                    a_micro(nset, jn, m);  // A of the next column (12 micro-ops, one per gap)
                    if (j == 0 && m >= 6 && m < 6 + HR1)
                        hreg[m - 6] = *reinterpret_cast<const f32x4*>(xin + ((size_t)(blockIdx.x * 8 + (c & 7)) * HR1 * 256 + 256 * (m - 6) + tid) * 4);
                    if (j == 1 && m >= 6 && m < 6 + HR1) *reinterpret_cast<f32x4*>(wstore + hoff[m - 6]) = hreg[m - 6];
                    // T of the next chunk: columns 0 and 2 die after A[2] (phase 1), columns 1 and 3 after A[3] (phase 2)
                    if (j == 2) {
                        if (m >= LAT && m < 4 + LAT) t_finish(((m - LAT) >> 1) * 2, (m - LAT) & 1, (m - LAT) % LAT);
                        if (m < 4) t_issue(wnext, (m >> 1) * 2, m & 1, m % LAT);
                    }
                    if (j == 3) {
                        if (m >= LAT && m < 4 + LAT) t_finish(((m - LAT) >> 1) * 2 + 1, (m - LAT) & 1, (m - LAT) % LAT);
                        if (m < 4) t_issue(wnext, (m >> 1) * 2 + 1, m & 1, m % LAT);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[j][q][e];
    out[blockIdx.x * 256 + tid] = s;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            return 1;                                                          \
        }                                                                      \
    } while (0)

template <int MODE>
int run(const char* name, const bf16x8* w, const float* x, float* out, int n_chunks) {
    const int blocks = 256, reps = 40;
    const size_t lds = 2 * WIN * sizeof(float);
    CK(hipFuncSetAttribute((const void*)winob_rate<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(winob_rate<MODE>, dim3(blocks), dim3(256), lds, 0, w, x, out, n_chunks, 1);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(winob_rate<MODE>, dim3(blocks), dim3(256), lds, 0, w, x, out, n_chunks, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // fp32-equivalent FLOPs of the Winograd-domain products: 16 xi x 64 tiles x 64 couts x 16 channels x 2 per chunk and workgroup
    const double flops = (double)blocks * reps * n_chunks * 16.0 * 64 * 64 * 16 * 2;
    const double tf = flops / ms / 1e9;
    printf("%-64s %8.3f ms  %7.1f TF/s fp32-equivalent = %.2f of the six-product peak (419), %.2fx the f32-MFMA peak\n", name, ms,
           tf, tf / 419.4, tf / 157.3);
    return 0;
}

template <int ABL>
int run_pipe(const char* name, const bf16x8* w, const float* x, float* out, int n_chunks) {
    const int blocks = 256, reps = 40;
    const size_t lds = 2 * WIN * sizeof(float);
    CK(hipFuncSetAttribute((const void*)winob_pipe<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(winob_pipe<ABL>, dim3(blocks), dim3(256), lds, 0, w, x, out, n_chunks, 1);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(winob_pipe<ABL>, dim3(blocks), dim3(256), lds, 0, w, x, out, n_chunks, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * reps * n_chunks * 16.0 * 64 * 64 * 16 * 2;
    const double tf = flops / ms / 1e9;
    printf("%-64s %8.3f ms  %7.1f TF/s fp32-equivalent = %.2f of the six-product peak (419), %.2fx the f32-MFMA peak\n", name, ms,
           tf, tf / 419.4, tf / 157.3);
    return 0;
}

int run_pipe1(const char* name, const bf16x8* w, const float* x, float* out, int n_chunks) {
    const int blocks = 512, reps = 40;
    const size_t lds = 2 * WIN1 * sizeof(float);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(winob_pipe1, dim3(blocks), dim3(256), lds, 0, w, x, out, n_chunks, 1);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(winob_pipe1, dim3(blocks), dim3(256), lds, 0, w, x, out, n_chunks, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * reps * n_chunks * 16.0 * 32 * 64 * 16 * 2;
    const double tf = flops / ms / 1e9;
    printf("%-64s %8.3f ms  %7.1f TF/s fp32-equivalent = %.2f of the six-product peak (419), %.2fx the f32-MFMA peak\n", name, ms,
           tf, tf / 419.4, tf / 157.3);
    return 0;
}

int main() {
    const int n_chunks = 32;  // a 512-channel layer
    const size_t wn = (size_t)n_chunks * 16 * 3 * 64 * 2;  // bf16x8 units
    bf16x8* w;
    float *x, *out;
    CK(hipMalloc(&w, wn * sizeof(bf16x8)));
    CK(hipMemset(w, 0x3c, wn * sizeof(bf16x8)));  // 0x3c3c = 0.0115 in bf16
    const size_t xn = (size_t)512 * 8 * HR * 256 * 4;
    CK(hipMalloc(&x, xn * sizeof(float)));
    CK(hipMemset(x, 0, xn * sizeof(float)));
    CK(hipMalloc(&out, 512 * 256 * sizeof(float)));
    if (run<0>("MFMAs only", w, x, out, n_chunks)) return 1;
    if (run<1>("+ B planes from global memory (24 KB per wave and chunk)", w, x, out, n_chunks)) return 1;
    if (run<2>("+ A planes from the LDS window (reads, transform, split)", w, x, out, n_chunks)) return 1;
    if (run<3>("+ window staging (global loads, ds_write, barrier)", w, x, out, n_chunks)) return 1;
    if (run_pipe<0>("everything, software-pipelined by hand (hooks between the MFMAs)", w, x, out, n_chunks)) return 1;
    if (run_pipe<1>("  pipelined, without the B loads", w, x, out, n_chunks)) return 1;
    if (run_pipe<2>("  pipelined, without the A production", w, x, out, n_chunks)) return 1;
    if (run_pipe<4>("  pipelined, without the window staging", w, x, out, n_chunks)) return 1;
    if (run_pipe<12>("  pipelined, without staging and barrier", w, x, out, n_chunks)) return 1;
    if (run_pipe<15>("  pipelined, MFMAs only", w, x, out, n_chunks)) return 1;
    if (run_pipe1("32-tile form, two workgroups per CU, everything", w, x, out, n_chunks)) return 1;
    return 0;
}

// Device-side pieces shared by the convolution kernels (conv_mfma.hip, winograd_mfma.hip).
#pragma once

#include "dm_common.h"

namespace dm {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// 16-byte staging values are NATIVE vectors, not HIP's float4 struct: arrays of the struct are copied with
// llvm.memcpy between address spaces, which keeps them in scratch memory (one synchronous round trip per
// global load) instead of registers.
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 make_f32x4(float a, float b, float c, float d) { return f32x4{a, b, c, d}; }

// Packed fp32 VALU ops (two floats per lane and instruction).  On gfx950 the f32-input MFMA runs on the same
// vector ALUs as ordinary VALU code -- measured: a kernel's time is the SUM of 64 cycles per 32x32x2 MFMA and
// 4 cycles per VALU instruction, nothing overlaps -- so every VALU instruction next to an MFMA stream costs 1/16
// of an MFMA and the packed forms halve that.  The compiler emits v_pk_add/v_pk_fma only for some patterns
// (never for subtraction), hence the explicit forms.
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {  // a * b + c
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x4 join4(f32x2 lo, f32x2 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3); }
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) { return join4(pk_add(a.xy, b.xy), pk_add(a.zw, b.zw)); }
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) { return join4(pk_sub(a.xy, b.xy), pk_sub(a.zw, b.zw)); }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x2 s, f32x4 c) {  // a * s + c, s = one scalar in both halves
    return join4(pk_fma(a.xy, s, c.xy), pk_fma(a.zw, s, c.zw));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ float half_wave_sum(float v) {
    // sum over the 32 lanes that share lane>>5; every lane ends with the total.
    // Four DPP steps inside each 16-lane row (they fuse into v_add_f32_dpp), one swizzle across the two rows.
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]  : lane ^ 1
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]  : lane ^ 2
    v += dpp_f<0x141>(v);  // row_half_mirror      : pairs the two quads of each 8 lanes
    v += dpp_f<0x140>(v);  // row_mirror           : pairs the two halves of the row
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // lane ^ 16
    return v;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// In-kernel cycle stamps: DIAGNOSTIC build only (make STAMPS=1 -> libdm_hip_stamps.so, tools/conv_stamps.py).
// Wave 0 of every workgroup sums the s_memtime cycles it spends per phase into p.stamps[block][8]; nothing
// the kernel outputs depends on them.  In the shipped library every macro below is empty.
#ifdef DM_STAMPS
__device__ __forceinline__ unsigned long long dm_stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DM_STAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DM_STAMP(k) st_prev = dm_stamp_now();
#define DM_STAMP_ADD(k)                              \
    {                                                \
        unsigned long long st_now = dm_stamp_now();  \
        st_acc[k] += st_now - st_prev;               \
        st_prev = st_now;                            \
    }
#define DM_STAMP_FLUSH                                                                              \
    if (threadIdx.x == 0 && p.stamps) {                                                             \
        unsigned long long* sp_ = p.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;     \
        for (int k_ = 0; k_ < 8; ++k_) sp_[k_] = st_acc[k_];                                        \
    }
#else
#define DM_STAMP_DECL
#define DM_STAMP(k)
#define DM_STAMP_ADD(k)
#define DM_STAMP_FLUSH
#endif

// Row-layout epilogue shared by the convolution kernels.  A wave holds 64 output pixels x 64 consecutive couts
// (NR = 16; 32 pixels for NR = 8) as v[j] = the 4 couts [cg, cg+4) of pixel row 4*j + (lane>>4); the 16 lanes
// of a DPP row cover one pixel.
// pixv[j] is the global output pixel of that row or -1.  Every global access is a 16-byte one, the RMSNorm
// reduction stays inside a DPP row (plus one LDS exchange when the pixel's couts span WN waves: `red` is
// [WN][rows_per_wg] floats and row_in_wg0 the index of this wave's first row in it -- all waves of the
// workgroup must then call this function together).
struct RowsEpilogue {
    int split;        // K split index (partial sums go to out[split][pixel][cout])
    size_t M;         // output pixels of the whole tensor
    int b0;           // first image of the tile (scale/shift row when `uni`)
    bool uni;         // one scale/shift row serves the whole tile
    int HoWo;         // output pixels per image
    float* red;
    int rows_per_wg;
    int row_in_wg0;
    int wn;
};

// Everything the epilogue reads from global memory, fetched BEFORE the accumulators go through LDS so that the
// HBM/L2 latency (bias, g, scale/shift, the residual rows) overlaps the transposition instead of following it.
// RES = false leaves the residual rows to the point of use (for callers without the registers to hold them).
template <int NR, bool RES = true>
struct RowsPrefetch {
    f32x4 b4, g4, sc, sh;
    f32x4 res[RES ? NR : 1];
};

template <int NR, bool RES>
__device__ __forceinline__ void rows_prefetch(const ConvParams& p, const RowsEpilogue& e, const int (&pixv)[NR], int cg,
                                              bool cvalid, RowsPrefetch<NR, RES>& f) {
    const f32x4 zero4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    f.b4 = zero4;
    f.g4 = zero4;
    f.sc = zero4;
    f.sh = zero4;
    if constexpr (RES) {
#pragma unroll
        for (int j = 0; j < NR; ++j) f.res[j] = zero4;
    }
    if (p.partial) return;
    const int epi = p.epi;
    if ((epi & EPI_BIAS) && cvalid) f.b4 = *reinterpret_cast<const f32x4*>(p.bias + cg);
    if ((epi & EPI_NORM) && cvalid) f.g4 = *reinterpret_cast<const f32x4*>(p.g + cg);
    if ((epi & EPI_SCALE_SHIFT) && e.uni && cvalid) {
        const float* sp = p.scale + (size_t)min(e.b0, p.B - 1) * p.ss_stride;
        f.sc = *reinterpret_cast<const f32x4*>(sp + cg);  // raw scale: nothing here may USE a loaded value
        f.sh = *reinterpret_cast<const f32x4*>(sp + p.Cout + cg);
    }
    if constexpr (RES) {
        if (epi & EPI_RESIDUAL) {
            // unconditional loads (rows outside the tensor read row 0 and are never stored): no per-row branches
            const int cgc = cvalid ? cg : 0;
#pragma unroll
            for (int j = 0; j < NR; ++j)
                f.res[j] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)max(pixv[j], 0) * p.Cout + cgc);
        }
    }
}

template <int WN, int NR, bool RES>
__device__ __forceinline__ void rows_epilogue(const ConvParams& p, const RowsEpilogue& e, f32x4 (&v)[NR],
                                              const int (&pixv)[NR], int cg, bool cvalid,
                                              const RowsPrefetch<NR, RES>& f) {
    const int lane = threadIdx.x & 63;
    const int rsub = lane >> 4;
    if (p.partial) {
        float* po = p.out + (size_t)e.split * e.M * p.Cout + cg;
#pragma unroll
        for (int j = 0; j < NR; ++j)
            if (pixv[j] >= 0 && cvalid) *reinterpret_cast<f32x4*>(po + (size_t)pixv[j] * p.Cout) = v[j];
        return;
    }
    const int epi = p.epi;
    const f32x4 zero4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NR; ++j) v[j] = cvalid ? v[j] + f.b4 : zero4;
    if (epi & EPI_NORM) {
        const float sqrtc = sqrtf((float)p.Cout);
        const f32x4 g4 = f.g4 * sqrtc;
        float ssv[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float ss = v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
            ss += dpp_f<0xB1>(ss);
            ss += dpp_f<0x4E>(ss);
            ss += dpp_f<0x141>(ss);
            ss += dpp_f<0x140>(ss);  // all 16 lanes of the pixel hold the sum over this wave's 64 couts
            ssv[j] = ss;
        }
        if constexpr (WN > 1) {
            // the pixel's couts are spread over WN waves: exchange the partial sums through LDS
            if ((lane & 15) == 0) {
#pragma unroll
                for (int j = 0; j < NR; ++j) e.red[e.wn * e.rows_per_wg + e.row_in_wg0 + 4 * j + rsub] = ssv[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) t += e.red[w * e.rows_per_wg + e.row_in_wg0 + 4 * j + rsub];
                ssv[j] = t;
            }
        }
        // 1 / max(||v||, 1e-12) as one v_rsq_f32 (1 ulp)
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = v[j] * (fast_rsq(fmaxf(ssv[j], 1e-24f))) * g4;
    }
    if (epi & EPI_SCALE_SHIFT) {
        f32x4 sc = f.sc + 1.0f, sh = f.sh;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (!e.uni && cvalid && pixv[j] >= 0) {
                const float* sp = p.scale + (size_t)(pixv[j] / e.HoWo) * p.ss_stride;
                sc = *reinterpret_cast<const f32x4*>(sp + cg) + 1.0f;
                sh = *reinterpret_cast<const f32x4*>(sp + p.Cout + cg);
            }
            v[j] = v[j] * sc + sh;
        }
    }
    if (epi & EPI_SILU) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            v[j].x = v[j].x * fast_rcp(1.0f + __expf(-v[j].x));
            v[j].y = v[j].y * fast_rcp(1.0f + __expf(-v[j].y));
            v[j].z = v[j].z * fast_rcp(1.0f + __expf(-v[j].z));
            v[j].w = v[j].w * fast_rcp(1.0f + __expf(-v[j].w));
        }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        if (pixv[j] < 0 || !cvalid) continue;
        const size_t o = (size_t)pixv[j] * p.Cout + cg;
        if constexpr (RES) {
            *reinterpret_cast<f32x4*>(p.out + o) = v[j] + f.res[j];
        } else {
            f32x4 r4 = v[j];
            if (epi & EPI_RESIDUAL) r4 += *reinterpret_cast<const f32x4*>(p.residual + o);
            *reinterpret_cast<f32x4*>(p.out + o) = r4;
        }
    }
}

}  // namespace dm

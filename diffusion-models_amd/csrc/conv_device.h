// Device-side pieces shared by the convolution kernels (conv_mfma.hip, winograd_mfma.hip).
#pragma once

#include "dm_common.h"

namespace dm {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Workgroup -> (cout tile, pixel block).  The dispatcher deals workgroups to the 8 XCDs round-robin by linear id and
// each XCD has its own L2.  With the cout tile as the fastest index an XCD always works on the same cout tile (its L2
// keeps that tile's weights), but in raw order the pixel blocks of an XCD are 8 / n_tiles_n apart, so the halo a block
// shares with its neighbours is fetched from HBM by each of them (F(2x2) at 32x32: the 8 blocks of an image sit on 8
// XCDs, 1.41x the input bytes).  With xcd_groups = 8 / n_tiles_n > 0 the XCDs that share a cout tile split the pixel
// blocks into contiguous ranges instead: neighbours run on the same XCD at about the same time.  The host sets
// xcd_groups only when gridDim.x % 8 == 0 (K splits in blockIdx.y then keep the XCD assignment) and n_tiles_n | 8.
__device__ __forceinline__ void block_to_tile(const ConvGeom& g, int bid, int nblk, int& n_tile, int& pblock) {
    n_tile = bid % g.n_tiles_n;
    pblock = bid / g.n_tiles_n;
    if (g.xcd_groups > 0) {
        const int per = (nblk / g.n_tiles_n) / g.xcd_groups;  // pixel blocks per XCD group
        pblock = (pblock % g.xcd_groups) * per + pblock / g.xcd_groups;
    }
}
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

// 16-byte load through a buffer resource: base and size in 4 SGPRs, byte offset = VGPR part (per lane, loop invariant)
// + SGPR part (per chunk / position / column, scalar arithmetic): no vector address arithmetic at all.  Reads past the
// size return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)min(bytes, (size_t)0xFFFFFFFFu),
                                             0x00020000);
}
__device__ __forceinline__ f32x4 bufload4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, (int)soff_bytes, 0));
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
    bool all_valid;   // every lane of the tile maps to a real cout (no masking needed)
};

// global-memory access as (wave-uniform base in SGPRs) + (32-bit element offset in a VGPR): no 64-bit vector
// arithmetic per access.  Tensors on this path hold < 2^30 elements (checked by the launchers).
using gfloat_ptr = const float __attribute__((address_space(1)))*;
using gfloat_mptr = float __attribute__((address_space(1)))*;
__device__ __forceinline__ f32x4 gload4(gfloat_ptr base, unsigned off_floats) {
    using gchar_ptr = const char __attribute__((address_space(1)))*;
    return *(const f32x4 __attribute__((address_space(1)))*)((gchar_ptr)base + (off_floats << 2));
}
__device__ __forceinline__ void gstore4(gfloat_mptr base, unsigned off_floats, f32x4 v) {
    using gchar_ptr = char __attribute__((address_space(1)))*;
    *(f32x4 __attribute__((address_space(1)))*)((gchar_ptr)base + (off_floats << 2)) = v;
}

// Everything the epilogue reads from global memory, fetched BEFORE the accumulators go through LDS so that the
// HBM/L2 latency (bias, g, scale/shift, the residual rows) overlaps the transposition instead of following it.
// RES = false leaves the residual rows to the point of use (for callers without the registers to hold them).
template <int NR, bool RES = true>
struct RowsPrefetch {
    f32x4 b4, g4, sc, sh;
    f32x4 res[RES ? NR : 1];
    unsigned off[NR];  // element offset of (pixel row j, cout cg) in the output / residual tensor
};

template <int NR, bool RES>
__device__ __forceinline__ void rows_prefetch(const ConvParams& p, const RowsEpilogue& e, const int (&pixv)[NR], int cg,
                                              bool cvalid, RowsPrefetch<NR, RES>& f) {
    // fields are only read by rows_epilogue under the same epilogue flags they are loaded under: no defaults
    const int cgc = cvalid ? cg : 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        // full-rate 24-bit multiply while pixel indices allow it (they do for every benchmark shape)
        const unsigned px = (unsigned)max(pixv[j], 0);
        f.off[j] = (e.M < (1u << 24) ? __umul24(px, (unsigned)p.Cout) : px * (unsigned)p.Cout) + cgc;
    }
    if (p.partial) return;
    const int epi = p.epi;
    if (epi & EPI_BIAS) f.b4 = *reinterpret_cast<const f32x4*>(p.bias + cgc);
    if (epi & EPI_NORM) f.g4 = *reinterpret_cast<const f32x4*>(p.g + cgc);
    if ((epi & EPI_SCALE_SHIFT) && e.uni) {
        const float* sp = p.scale + (size_t)min(e.b0, p.B - 1) * p.ss_stride;
        f.sc = *reinterpret_cast<const f32x4*>(sp + cgc);  // raw scale: nothing here may USE a loaded value
        f.sh = *reinterpret_cast<const f32x4*>(sp + p.Cout + cgc);
    }
    if constexpr (RES) {
        if (epi & EPI_RESIDUAL) {
            // unconditional loads (rows outside the tensor read row 0 and are never stored): no per-row branches
            const gfloat_ptr rp = (gfloat_ptr)p.residual;
#pragma unroll
            for (int j = 0; j < NR; ++j) f.res[j] = gload4(rp, f.off[j]);
        }
    }
}

// The arithmetic is written with the packed forms: next to an MFMA stream every VALU instruction counts (see above).
template <int WN, int NR, bool RES>
__device__ __forceinline__ void rows_epilogue(const ConvParams& p, const RowsEpilogue& e, f32x4 (&v)[NR],
                                              const int (&pixv)[NR], int cg, bool cvalid,
                                              const RowsPrefetch<NR, RES>& f) {
    const int lane = threadIdx.x & 63;
    const int rsub = lane >> 4;
    if (p.partial) {
        const gfloat_mptr po = (gfloat_mptr)(p.out + (size_t)e.split * e.M * p.Cout);
#pragma unroll
        for (int j = 0; j < NR; ++j)
            if (pixv[j] >= 0 && cvalid) gstore4(po, f.off[j], v[j]);
        return;
    }
    const int epi = p.epi;
    const f32x4 zero4 = make_f32x4(0.f, 0.f, 0.f, 0.f);
    if (epi & EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = add4(v[j], f.b4);
    }
    if (!e.all_valid) {  // lanes past the last cout must not feed the norm
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = cvalid ? v[j] : zero4;
    }
    if (epi & EPI_NORM) {
        const float sqrtc = sqrtf((float)p.Cout);
        const f32x4 g4 = cvalid ? f.g4 * sqrtc : zero4;
        float ssv[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const f32x2 sq = pk_fma(v[j].zw, v[j].zw, pk_mul(v[j].xy, v[j].xy));
            float ss = sq.x + sq.y;
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
        // 1 / max(||v||, 1e-12) as one v_rsq_f32 (1 ulp).  The consumer of a transcendental's result is plain C++,
        // never inline asm: gfx950 needs a wait state between v_rsq / v_exp / v_rcp and the VALU instruction that reads
        // the result, which the compiler inserts for its own instructions but not for asm operands (stale high halves).
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const float rn = fast_rsq(fmaxf(ssv[j], 1e-24f));
            const f32x4 gr = g4 * rn;
            v[j] = join4(pk_mul(v[j].xy, gr.xy), pk_mul(v[j].zw, gr.zw));
        }
    }
    if (epi & EPI_SCALE_SHIFT) {
        f32x4 sc = make_f32x4(1.f, 1.f, 1.f, 1.f), sh = zero4;
        if (e.uni) {
            sc = f.sc + 1.0f;
            sh = f.sh;
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (!e.uni && cvalid && pixv[j] >= 0) {
                const float* sp = p.scale + (size_t)(pixv[j] / e.HoWo) * p.ss_stride;
                sc = *reinterpret_cast<const f32x4*>(sp + cg) + 1.0f;
                sh = *reinterpret_cast<const f32x4*>(sp + p.Cout + cg);
            }
            v[j] = join4(pk_fma(v[j].xy, sc.xy, sh.xy), pk_fma(v[j].zw, sc.zw, sh.zw));
        }
    }
    if (epi & EPI_SILU) {
        // x * 1 / (1 + 2^(-x log2 e)): the same instructions __expf / the reciprocal lower to, packed where possible
        // (the additions and the final multiply read transcendental results: plain C++, see the note at the norm)
        const f32x2 nl2e = {-1.4426950408889634f, -1.4426950408889634f};
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const f32x2 t0 = pk_mul(v[j].xy, nl2e), t1 = pk_mul(v[j].zw, nl2e);
            f32x4 e4 = make_f32x4(__builtin_amdgcn_exp2f(t0.x), __builtin_amdgcn_exp2f(t0.y),
                                  __builtin_amdgcn_exp2f(t1.x), __builtin_amdgcn_exp2f(t1.y));
            e4 = e4 + 1.0f;
            const f32x4 r4 = make_f32x4(fast_rcp(e4.x), fast_rcp(e4.y), fast_rcp(e4.z), fast_rcp(e4.w));
            v[j] = v[j] * r4;
        }
    }
    if (epi & EPI_RELU) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
            v[j] = make_f32x4(fmaxf(v[j].x, 0.f), fmaxf(v[j].y, 0.f), fmaxf(v[j].z, 0.f), fmaxf(v[j].w, 0.f));
    }
    const gfloat_mptr op = (gfloat_mptr)p.out;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        if (pixv[j] < 0 || !cvalid) continue;
        if constexpr (RES) {
            gstore4(op, f.off[j], (epi & EPI_RESIDUAL) ? add4(v[j], f.res[j]) : v[j]);
        } else {
            f32x4 r4 = v[j];
            if (epi & EPI_RESIDUAL) r4 = add4(r4, gload4((gfloat_ptr)p.residual, f.off[j]));
            gstore4(op, f.off[j], r4);
        }
    }
}

}  // namespace dm

// Internal declarations shared by the HIP translation units of libdm_hip.so.
// gfx950 (MI355X) only: 64-wide wavefronts, f32-input MFMA, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include <cstdlib>

namespace dm {

// integer tuning knob from the environment (read once by the callers: they keep the value in a function-local static)
inline int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}


void set_error(const std::string& msg);

#define DM_CHECK_HIP(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            dm::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));               \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

#define DM_REQUIRE(cond, msg)                                                               \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            dm::set_error(std::string(msg) + " [" #cond "]");                                \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

// Opt-in of a kernel to more than 64 KB of dynamic LDS.  hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE
// attribute, so the "already done" state is kept per device: a process that builds handles on two GPUs (cond_vae on
// cuda:1 next to the U-Net on cuda:0) opts in on both.  One LdsOptIn per launch site; `n_kernels` = how many kernels
// that site opts in through it (they are always called in the same order).
struct LdsOptIn {
    unsigned char done[64] = {};
};
inline int lds_opt_in(LdsOptIn& f, const void* kernel, int n_kernels) {
    int dev = 0;
    DM_CHECK_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && f.done[dev] >= n_kernels) return 0;
    DM_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (tracked) ++f.done[dev];
    return 0;
}

// ---------------------------------------------------------------------------------------
// Implicit-GEMM convolution on the f32 MFMA (conv_mfma.hip)
// ---------------------------------------------------------------------------------------
// Activations are NHWC fp32 inside the library.  A convolution is
//   D[pixel][cout] = sum_{tap, cin} A[pixel + tap][cin] * W[tap][cin][cout]
// Pixels are the MFMA rows, output channels the MFMA columns, (tap, cin) the reduction.

enum ConvEpilogue : int {
    EPI_BIAS = 1,        // + bias[cout]
    EPI_NORM = 2,        // RMSNorm over cout: v / max(||v||, 1e-12) * g[cout] * sqrt(Cout)   (needs NT >= Cout)
    EPI_SCALE_SHIFT = 4, // v * (scale[b][cout] + 1) + shift[b][cout]
    EPI_SILU = 8,        // v * sigmoid(v)
    EPI_RESIDUAL = 16,   // + residual[pixel][cout]
    EPI_RELU = 32,       // max(v, 0)  (BasicConv2d of the Inception evaluators; after bias, before the residual)
};

struct ConvGeom {
    // tiling of the output, chosen on the host (conv_plan)
    int WM, WN;          // waves along pixels / couts (WM*WN == 4); tile = (64*WM) x (64*WN)
    int CK;              // channels per K chunk: 16, or 4 for thin inputs
    int TW, TH, NB;      // output tile: NB images x TH rows x TW cols (TW, TH powers of two)
    int lTW, lTH;        // log2
    int tiles_x, tiles_y, groups, n_tiles_n;
    int IH, IW;          // staged input window per image
    int halo_floats;     // LDS floats for the activation window (multiple of 4)
    int w_floats;        // LDS floats of one weight slab buffer (two are allocated)
    int ptab_off;        // LDS float offset of the output-pixel table (set by conv_launch)
    int TPS;             // taps (along kx) per weight slab: KW or 1
    int splits;          // K splits (blockIdx.y); > 1 writes partial sums
    int chunks_per_split;
    int fused_norm;      // the plan keeps all couts of a pixel in one workgroup (EPI_NORM allowed)
    int row_stride;      // winograd: LDS floats per window row
    int magic_win, magic_row;  // winograd: ceil(2^24 / (IH*IW)), ceil(2^24 / IW): divisions of the window setup
    int lds_bytes;
    int xcd_groups;      // 3x3 kernels: XCDs that share a cout tile (8 / n_tiles_n), 0 = blocks in raw order
    int NQ;              // winograd F(2x2): 32-cout MFMA blocks per workgroup (2: 64 couts; 1: 32, for grids that leave CUs idle)
};

struct ConvParams {
    const float* in0;
    const float* in1;
    int C0, C1;          // real channels of each source (concat along C)
    int chunks0;         // K chunks that come from source 0
    int n_chunks;        // total K chunks
    int Hin, Win;        // spatial size the convolution sees (after nearest x2 if up)
    int up;              // sources are (Hin/2, Win/2), nearest-upsampled on the fly
    int in_nchw;         // source 0 is NCHW (boundary tensors); requires C1 == 0
    const float* w;      // packed [chunk][ky][kx][Cout padded to 64][CK]
    const float* bias;
    int Cout, KH, KW, stride, pad;
    int pad_w;           // zero columns left of the image (pad = rows above it); equal to pad for every U-Net / VAE conv
    int B, Ho, Wo;
    float* out;
    int out_nchw;
    int partial;         // out is [splits][pixel][Cout] raw partial sums, no epilogue
    // Nearest-x2 + conv3x3 folded into four 2x2 convolutions on the SOURCE grid, one per output parity
    // (blockIdx.z = py*2+px): out[2i+py][2j+px] = sum_{a,b<2} W'[py][px][a][b] * src[i+a-(1-py)][j+b-(1-px)],
    // W' = sums of the 3x3 taps that hit the same source pixel.  4/9 of the multiply-adds.  Ho/Wo are the
    // source-grid dims; the output tensor is (2Ho, 2Wo).
    int fold;
    // 2x2 / stride-2 convolution (Downsample = pixel-unshuffle + 1x1, :54-58) executed as a 1x1 convolution whose K
    // chunks run over (cin chunk, tap): no overlapping window, so the tile stages only its own 4 source pixels per
    // output pixel.  Hin/Win are the OUTPUT dims; KH = KW = 1 in the kernel; n_chunks = 4 * C0/16.
    int s2d;
    int fold_w_stride;   // floats between the four packed weight sets
    unsigned long long* stamps;  // diagnostic build (-DDM_STAMPS) only: per-workgroup phase cycle sums
    int epi;
    const float* residual;   // NHWC [pixel][Cout]
    const float* g;          // [Cout]
    const float* scale;      // scale[b * ss_stride + c], shift = scale + Cout
    int ss_stride;
    ConvGeom geo;
};

// choose wave grid, tile, weight-slab size and K splits for an output of (B, Ho, Wo, Cout);
// want_norm: the caller would like RMSNorm fused (granted when plan.fused_norm != 0)
ConvGeom conv_plan(int B, int Ho, int Wo, int Cout, int KH, int KW, int stride, int C0, int C1, bool want_norm,
                   bool allow_split, int grid_z = 1);
int conv_ck_for(int C0, int C1);
// floats needed for packed weights
size_t conv_packed_floats(int Cout, int C0, int C1, int KH, int KW);
// pack OIHW (Cout, C0+C1, KH, KW) -> kernel layout (host side)
void conv_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1, int KH, int KW);
int conv_launch(const ConvParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) convolution on the f32 MFMA (winograd_mfma.hip): 3x3 / stride 1 / pad 1, NHWC,
// C0 % 8 == 0, C1 % 8 == 0, Cout % 64 == 0, even output size.  Same ConvParams / epilogue contract as
// conv_launch; ConvGeom is reused with TW/TH/NB counted in 2x2-pixel Winograd tiles (64 per workgroup).
// ---------------------------------------------------------------------------------------
bool wino_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up);
size_t wino_packed_floats(int Cout, int C0, int C1);
// OIHW (Cout, C0+C1, 3, 3) -> U = G g G^T in kernel layout [chunk of 8 cin][16 xi][Cout][8]
void wino_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1);
ConvGeom wino_plan(int B, int Ho, int Wo, int Cout, int C0, int C1, bool allow_split, bool want_norm = true);
// the shape-dependent half of eligibility: even size, window fits the staging registers, 24-bit pixel indices
bool wino_shape_ok(int B, int Ho, int Wo, int Cout, int C0, int C1);
int wino_launch(const ConvParams& p, hipStream_t s);

// Winograd F(4x4, 3x3) convolution (wino4_mfma.hip): same contract as wino_launch, 2.25 instead of 4 multiplies per
// output; images of 4x4, 8x8, 16x16 or (16k x 32m) pixels.  ConvGeom TW/TH/NB count 4x4-pixel tiles (32 per workgroup).
// ---------------------------------------------------------------------------------------
bool wino4_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up);
size_t wino4_packed_floats(int Cout, int C0, int C1);
// OIHW (Cout, C0+C1, 3, 3) -> U = G g G^T (6x6) in kernel layout [chunk of 8 cin][wave 4][slot 9][Cout][8]
void wino4_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1);
ConvGeom wino4_plan(int B, int Ho, int Wo, int Cout, int C0, int C1, bool allow_split);
bool wino4_shape_ok(int B, int Ho, int Wo, int Cout, int C0, int C1);
int wino4_launch(const ConvParams& p, hipStream_t s);

// The 7x7 / pad 3 first convolution over an NCHW image with 1-8 channels -> 64 NHWC channels (init7_mfma.hip)
bool init7_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up);
size_t init7_packed_floats(int Cin);
void init7_pack_weights(const float* oihw, float* packed, int Cin);
int init7_launch(const ConvParams& p, hipStream_t s);

// 1x1 convolution as a register-direct GEMM (pw_mfma.hip): NHWC, one or two sources, C % 8 == 0, Cout % 64 == 0;
// w = pw_pack_weights, chunks of 8 input channels.
bool pw_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up);
size_t pw_packed_floats(int Cout, int C0, int C1);
void pw_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1);
// Downsample (2x2 / stride 2 over (Cout, C0, 2, 2) weights) on the same kernel: ConvParams::s2d = 1, C0 = source channels
bool pw_s2d_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up);
void pw_pack_weights_s2d(const float* oihw, float* packed, int Cout, int C0);
ConvGeom pw_plan(int B, int Ho, int Wo, int Cout, int C0, int C1, bool allow_split);
bool pw_shape_ok(int B, int Ho, int Wo, int Cout, int C0, int C1);
int pw_launch(const ConvParams& p, hipStream_t s);

// Full self-attention over 16 tokens (a 4x4 feature map) as one kernel (attn16_fused.hip): RMSNorm, to_qkv, memory
// key/values, softmax, to_out + bias [+ x].  x, y: (B, 16, C) NHWC.
struct Attn16 {
    int C;
    const float* wp;      // attn16_pack: projection weights in lane order (gain, sqrt(C) and the softmax scale folded in)
    const float* wo;      // attn16_pack: to_out weights in lane order
    const float* bias;    // to_out bias (C)
    const float* mem_kv;  // (2, 4, 4, 32) as the reference stores it
};
bool attn16_eligible(int dim, int heads, int dh);
void attn16_pack(const float* w_qkv, const float* norm_g, const float* w_out, int C, std::vector<float>& wp,
                 std::vector<float>& wo);
int launch_attn16_fused(const Attn16& w, const float* x, float* y, int B, bool add_x, hipStream_t s);

// Nearest x2 upsampling + 3x3 convolution as a 9-multiply bilinear algorithm per source pixel (upwino_mfma.hip):
// (B, Hl, Wl) is the SOURCE tensor, the output is (B, 2 Hl, 2 Wl).  ConvParams: up = 1, Hin / Win = source size,
// Ho / Wo = output size, w = upwino_pack_weights, chunks of 8 input channels, single source (C1 == 0).
bool upwino_eligible(int Cout, int C0, int C1, int KH, int KW, int stride, int pad, bool up);
size_t upwino_packed_floats(int Cout, int C0, int C1);
void upwino_pack_weights(const float* oihw, float* packed, int Cout, int C0, int C1);
ConvGeom upwino_plan(int B, int Hl, int Wl, int Cout, int C0, int C1, bool allow_split);
bool upwino_shape_ok(int B, int Hl, int Wl, int Cout, int C0, int C1);
int upwino_launch(const ConvParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; never enabled while a graph is being captured.
// ---------------------------------------------------------------------------------------
namespace prof {
bool enabled();
bool detail();  // per-shape kernel names (tools/layer_report.py)
// bracket one launch: begin() records an event on `s`, end() records the closing one
int begin(const char* kernel, double flops, double bytes, hipStream_t s);
int end(hipStream_t s);
}  // namespace prof

// ---------------------------------------------------------------------------------------
// Elementwise / reduction kernels (elementwise.hip); all NHWC unless noted
// ---------------------------------------------------------------------------------------
int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, hipStream_t s);
int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, hipStream_t s);
// v = sum_{s<nsplit} x[s*split_stride + row*C + c] [+ bias[c]];
// y = [silu]( rmsnorm(v)*g*sqrt(C) [*(scale+1)+shift] ) [+ residual]; rows = pixels; flags = EPI_*
// the residual may itself be K-split partial sums of a convolution: sum_{s<res_nsplit} residual[s*res_stride + ...] + res_bias[c]
int launch_norm_act(const float* x, int nsplit, int64_t split_stride, const float* bias, const float* g,
                    const float* scale, int ss_stride, int pix_per_image, const float* residual, float* y,
                    int64_t rows, int C, int flags, hipStream_t s, int res_nsplit = 1, int64_t res_stride = 0,
                    const float* res_bias = nullptr);
// y[r][o] = act_out(bias[o] + sum_i act_in(x[r][i]) * W[o][i]);  act: 0 none, 1 silu, 2 gelu(erf)
int launch_linear_rows(const float* x, int ldx, const float* W, const float* bias, float* y, int ldy, int R, int I,
                       int O, int act_in, int act_out, hipStream_t s);
// Device-resident state of one sampling loop.  A captured denoise step reads everything that differs between two
// sample() calls of the same shape from here, so the instantiated graph is reused across calls.
struct SamplerState {
    int step;            // index of the current denoise step (advanced on the device)
    int n_steps;         // the step that also writes the final output is n_steps - 1
    int unnormalize;     // (x + 1) / 2 on the final output
    int pad_;
    uint64_t seed;       // Philox key
    uint64_t off4;       // Philox counter of this shard's first element = global element index / 4
};
// e[r] = cat(sin(t[r]*f), cos(t[r]*f)); t int64; if step_times != nullptr t = step_times[state->step]
int launch_sinusoid(const int64_t* t, const int64_t* step_times, const SamplerState* step, const float* freqs, float* e,
                    int R, int half, hipStream_t s, bool learned = false);
// VAE kernels (vae_kernels.hip): MFMA flash attention of AttnBlock, coalesced GroupNorm statistics
bool vae_attn_mfma_ok(int n, int C);
int launch_vae_attn_mfma(const float* q, const float* k, const float* v, float* out, int B, int n, int C, hipStream_t s);
bool group_sums_ok(int C, int groups);
size_t group_stats_ws_floats(int B, int groups);
int launch_group_stats_fast(const float* x, float* stats, double* acc, int B, int HW, int C, int groups, float eps,
                            hipStream_t s);
// GroupNorm(32 groups) + optional swish, NHWC; stats_ws holds B*groups*2 floats + B*groups*2 doubles
int launch_group_norm(const float* x, const float* w, const float* b, float* y, float* stats_ws, int B, int HW,
                      int C, int groups, float eps, int swish, hipStream_t s);
int launch_add(const float* a, const float* b, float* y, int64_t n, hipStream_t s);
// consumer-side ops (consumer_ops.hip): NHWC pooling (mode 0 max, 1 avg counting the padding, 2 avg over valid pixels),
// bilinear resize (align_corners = False) NCHW -> NHWC with a per-channel affine map, channel-slice copy, global mean
int launch_pool2d_nhwc(const float* in, float* out, int B, int H, int W, int C, int k, int stride, int pad, int mode,
                       hipStream_t s);
int launch_resize_bilinear(const float* in_nchw, float* out_nhwc, int B, int C, int H, int W, int Ho, int Wo,
                           const float* scale, const float* shift, hipStream_t s);
int launch_copy_channels_nhwc(const float* src, int Cs, float* dst, int Cd, int c_off, int64_t rows, hipStream_t s);
int launch_global_avgpool_nhwc(const float* in, float* out, int B, int HW, int C, hipStream_t s);
int launch_spin(double milliseconds, hipStream_t s);
int launch_copy_channels(const float* src, float* dst, int B, int Cs, int Cd, int c_off, int HW, hipStream_t s);
// y_nchw[b][o][sp] = bias[o] + sum_c x[pixel][c] * w[o][c]  for Cout <= 4 (NHWC in, NCHW out)
int launch_pointwise_small(const float* x, const float* w_oc, const float* bias, float* y_nchw, int64_t pixels, int C,
                           int Cout, int HW, hipStream_t s);

// state_dev == nullptr: one stand-alone update (step 0 of 1, no Philox noise, no unnormalize)
int launch_sampler_update(int kind, const float* x, const float* eps, const float* noise, const float* coefs_dev,
                          const SamplerState* state_dev, int64_t noise_step_stride, float* out, float* all_steps,
                          float* final_out, int64_t n, hipStream_t s, int objective = 0, float* xstart_out = nullptr);
// element e of the tensor uses Philox counter ((element_offset + e) / 4, draw); element_offset % 4 == 0
int launch_randn(float* out, int64_t n, uint64_t seed, uint64_t draw, uint64_t element_offset, hipStream_t s);
int launch_step_advance(SamplerState* state_dev, hipStream_t s);
int launch_finalize(const float* x, float* out, int64_t n, int unnormalize, hipStream_t s);
int launch_broadcast_rows(const float* src, float* dst, int rows, int n, int ld, hipStream_t s, int group = 0);
// (group > 0: source row r / group feeds destination row r; 0: one source row for all)

// ---------------------------------------------------------------------------------------
// Attention cores (attention.hip); qkv is NHWC (B, n, 3*heads*dh) = [q | k | v] per pixel
// ---------------------------------------------------------------------------------------
// LinearAttention core: out (B, n, heads*dh)
bool linattn_keeps_kstats();
int launch_linear_attention_core(const float* qkv, const float* mem_kv, float* ctx_ws, float* out, int B, int n,
                                 int heads, int dh, hipStream_t s, float* kstats = nullptr);
// kstats (optional, B * heads * 64 floats): per (image, head) the column max and the column sum of exp(k - max) the
// context kernel formed; launch_linear_attention_core_bwd starts from them instead of two more passes over the keys.
// softmax(q k^T * scale) v with `n_mem` learned key/value rows prepended.
//   q: rows of length ldq per query token (head h at column h*dh), k/v likewise with ldk
//   mem_k/mem_v: (heads, n_mem, dh) or nullptr
int launch_attention_core(const float* q, int ldq, const float* k, const float* v, int ldk, const float* mem_k,
                          const float* mem_v, int n_mem, float* out, int ldo, int B, int nq, int nk, int heads,
                          int dh, float scale, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Fused LinearAttention (linattn_fused.hip): RMSNorm + to_qkv + both softmaxes + context + to_out + RMSNorm (+ x)
// in two kernels; heads == 4, dim_head == 32, C in {64, 128}
// ---------------------------------------------------------------------------------------
struct LinAttnFused {
    int C;
    const float *wq, *wk, *wv;  // packed projections with the RMSNorm gain folded in
    const float* wo_raw;        // to_out weight (C, 128) as in the state dict
    const float* bias;          // to_out bias [C]
    const float* og;            // to_out RMSNorm gain * sqrt(C) [C]
    const float* kbound;        // softmax shift per (head, d) [128]
    const float* mem_kv;        // (2, heads, 32, 4)
};
bool linattn_fused_eligible(int C, int heads, int dh);
bool linattn_fused_pack(const float* w_qkv, const float* norm_g, const float* w_out, const float* mem_kv, int C,
                        std::vector<float>& wq, std::vector<float>& wk, std::vector<float>& wv, std::vector<float>& wo,
                        std::vector<float>& kbound);
size_t linattn_fused_ws_floats(int B, int n);
int launch_linattn_fused(const LinAttnFused& w, const float* x, float* ws, float* y, int B, int n, bool add_x,
                         hipStream_t s);


// ---------------------------------------------------------------------------------------
// Training step (dm_train.inc): weight gradients (wgrad_mfma.hip), norm / activation / loss / optimiser kernels
// (train_kernels.hip), attention-core backward (attn_bwd.hip)
// ---------------------------------------------------------------------------------------
size_t wgrad_ws_floats(int B, int Ho, int Wo, int Cout, int Cin, int T, int* splits_out);
// mode: 0 3x3 pad 1 (up: nearest x2 source), 1 1x1, 2 2x2 stride 2 (Ho, Wo = OUTPUT size; the source is 2Ho x 2Wo).
// dw: OIHW (mode 2: the (Cout, 4 C) Downsample layout); ws from wgrad_ws_floats
// The split-K sum of one layer (out[i * T + t] (+)= sum_split partial[(split * T + t) * oc + i]).  With `defer` the launchers
// below only describe it; the training step then sums the partial tiles of ALL its layers in one launch
// (launch_wgrad_reduce_jobs) instead of one small launch behind every weight-gradient kernel.
struct WgradJob {
    const float* partial;
    float* out;
    long long oc;
    int splits, T, accumulate, first_block;  // first_block: prefix sum of ceil(oc / 64) over the jobs (T <= 9)
};
int launch_wgrad(const float* in0, int C0, const float* in1, int C1, const float* dy, int Cout, int B, int Ho, int Wo,
                 int mode, int up, float* ws, float* dw, int accumulate, hipStream_t s, WgradJob* defer = nullptr);
// One layer's weight gradient as the grouped launch sees it (device table entry) and as the training step records it.
struct WgradParams {
    const float* in0;
    const float* in1;
    const float* dy;
    float* partial;  // [split][T][Cout][Cin]
    int C0, C1, Cin, Cout;
    int B, Ho, Wo;   // size of dY (and of the input the convolution sees, except s2d: input is 2Ho x 2Wo)
    int up;          // sources are (Ho/2, Wo/2), nearest-upsampled
    int TW, R, NB;   // pixel block: NB images x R rows x TW columns (TW even), NB*R*TW <= 64; NB > 1 only for whole images
    int tiles_x, tiles_y;
    int n_blocks, blocks_per_split;
    int n_ct, n_kt;  // cout / cin tiles of 64
    int n_splits, first_wg;  // grouped launch: pixel splits of this layer, its first workgroup
};
struct WgradDesc {
    const float *in0, *in1, *dy;
    float* dw;
    int C0, C1, Cout, B, Ho, Wo, up, accumulate;
};
void wgrad_group_plan(const std::vector<WgradDesc>& descs, int mode, std::vector<int>& splits, std::vector<size_t>& ws_floats);
int wgrad_group_fill(const std::vector<WgradDesc>& descs, int mode, const std::vector<int>& splits, const std::vector<float*>& ws,
                     std::vector<WgradParams>& table, std::vector<WgradJob>& jobs, int* total_wgs, size_t* lds_bytes);
int launch_wgrad_group(const WgradParams* table_dev, int n_jobs, int total_wgs, size_t lds_bytes, int mode, hipStream_t s);
// final_conv backward (1x1, <= 4 outputs, dy NCHW) in one pass: dw, db (through split partials in ws) and dx
size_t thin_out_bwd_ws_floats(int B, int H, int W, int Cout, int Cin);
bool thin_out_bwd_ok(int Cout, int HW);
int launch_thin_out_bwd(const float* x, const float* dy_nchw, const float* w_oc, float* dx, float* ws, float* dw, float* db,
                        int B, int H, int W, int Cin, int Cout, int accumulate, hipStream_t s, WgradJob* defer_w,
                        WgradJob* defer_b);
size_t wgrad_naive_ws_floats(int B, int H, int Cout, int Cin, int KH, int KW, int* splits_out);
int launch_wgrad_naive(const float* x, int x_nchw, const float* dy, int dy_nchw, int Cin, int Cout, int KH, int KW, int pad,
                       int B, int H, int W, float* ws, float* dw, int accumulate, hipStream_t s, WgradJob* defer = nullptr);
// jobs_dev: n_jobs descriptors in device memory with first_block filled in, total_blocks = the last prefix
int launch_wgrad_reduce_jobs(const WgradJob* jobs_dev, int n_jobs, int total_blocks, hipStream_t s);
int wgrad_reduce_pairs(int T, long long oc);  // (cout, cin) pairs per workgroup of the reduce kernels for a job
size_t norm_act_bwd_ws_floats(int B, int pix_per_image, int C);
// the two reduction stages behind a norm_act_bwd_kernel / the two passes of a column sum, as deferred table entries
struct RowgradJob {
    const float* part;
    float *dss, *img, *dg, *dbias;
    int chunks, C, dss_stride, B, accumulate, first_img_block, first_fin_block;
    int pad_;
};
struct ColsumJob {
    const float* x;
    float *ws, *out;
    long long rows, row_stride, col_stride;
    int C, accumulate, nb, rpb, first_part_block, first_fin_block;
};
int launch_norm_act_bwd(const float* dy, const float* u, const float* g, const float* ss, int ss_stride, int pix_per_image,
                        float* du, float* ws, float* dg, float* dbias, float* dss, int dss_stride, int B, int C, int flags,
                        int accumulate, hipStream_t s, RowgradJob* defer = nullptr, float drop_p = 0.f, uint64_t drop_seed = 0,
                        uint64_t drop_stream = 0, const float* add = nullptr, int dy_nsplit = 1, int64_t dy_stride = 0);
int launch_norm_act_drop(const float* u, const float* g, const float* ss, int ss_stride, int pix_per_image, const float* residual,
                         float* y, int B, int C, int flags, float drop_p, uint64_t drop_seed, uint64_t drop_stream, hipStream_t s,
                         int nsplit = 1, int64_t split_stride = 0, const float* bias = nullptr, float* u_out = nullptr);
void rowgrad_jobs_prefix(std::vector<RowgradJob>& jobs, int* img_blocks, int* fin_blocks);
int launch_rowgrad_jobs(const RowgradJob* jobs_dev, int n_jobs, int img_blocks, int fin_blocks, hipStream_t s);
void colsum_jobs_prefix(std::vector<ColsumJob>& jobs, int* part_blocks, int* fin_blocks);
int launch_colsum_jobs(const ColsumJob* jobs_dev, int n_jobs, int n_with_partial, int part_blocks, int fin_blocks, hipStream_t s);
size_t colsum_ws_floats(int64_t rows, int C);
int launch_colsum(const float* x, int64_t rows, int C, int64_t row_stride, int64_t col_stride, float* ws, float* out,
                  int accumulate, hipStream_t s, ColsumJob* defer = nullptr);
int launch_colsum_nchw(const float* dy, int B, int C, int HW, float* out, int accumulate, hipStream_t s);
int launch_act_fwd(const float* x, float* y, int64_t n, int act, hipStream_t s);  // 1 SiLU, 2 GELU (erf)
int launch_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, hipStream_t s);
int launch_add3(const float* a, const float* b, const float* c, float* y, int64_t n, hipStream_t s);
// out = x * mask / (1 - p) [+ add], mask from Philox (seed, stream, element / 4); x == nullptr: the mask factor itself
int launch_dropout(const float* x, const float* add, float* out, int64_t n, float p, uint64_t seed, uint64_t stream,
                   hipStream_t s);
int launch_depth_to_space(const float* t, const float* add, float* dx, int B, int Ho, int Wo, int C, hipStream_t s);
int launch_pool2x2_sum(const float* du, const float* add, float* dx, int B, int H, int W, int C, hipStream_t s);
int launch_pointwise_small_dgrad(const float* dy_nchw, const float* w_oc, float* dx, int64_t pixels, int C, int Cout, int HW,
                                 hipStream_t s);
int launch_linear_wgrad(const float* dy, int ldy, const float* x, int ldx, float* dw, int R, int I, int O, int act_in,
                        int accumulate, hipStream_t s);
int launch_mlp_rows_wgrad(const float* dy, int ldy, const float* x, int ldx, float* const* dw_rows, float* const* db_rows,
                          int R, int I, int O, int accumulate, hipStream_t s, const float* x_act = nullptr);
int launch_lincomb(const float* x, const float* y, const float* coef_dev, float* out, int B, int64_t per_sample, int mode,
                   int clamp, hipStream_t s);
int launch_mask_mix(const float* a, const float* b, const float* mask, float* out, int64_t n, hipStream_t s);
int launch_offset_noise(float* noise, const float* offset, float strength, int BC, int HW, hipStream_t s);
int launch_cdist(const float* x, const float* y, float* out, int n, int m, int64_t D, hipStream_t s);
size_t linear_dgrad_ws_floats(int R, int I, int O);
// small_gemm.hip: the batch-row Linear layers of the training step as MFMA GEMMs
bool rows_gemm_nt_ok(int R, int I, int O, int ldx);
int launch_rows_gemm_nt(const float* x, int ldx, const float* W, const float* bias, float* y, int ldy, int R, int I, int O,
                        hipStream_t s);
bool rows_gemm_nn_ok(int R, int I, int O, int ldy, int ldx);
int rows_gemm_nn_shares(int O);
int launch_rows_gemm_nn(const float* dy, int ldy, const float* W, float* dx_or_ws, int ldx, int R, int I, int O, hipStream_t s);
bool rows_gemm_tn_ok(int R, int I, int O, int ldy, int ldx);
int launch_rows_gemm_tn(const float* dy, int ldy, const float* x, int ldx, float* const* dw_rows, float* const* db_rows, float* dw,
                        int ldw, float* db, int R, int I, int O, int accumulate, hipStream_t s);
int launch_linear_dgrad(const float* dy, int ldy, const float* W, float* dx, int ldx, int R, int I, int O, float* ws,
                        hipStream_t s);
// per-sample scalars of the training step, gathered on the host as `extract` does: [0] sqrt_alphas_cumprod[t],
// [1] sqrt_one_minus_alphas_cumprod[t], [2] loss_weight[t], [3] (t > 0), [4] sqrt_recip_alphas_cumprod[t],
// [5] sqrt_recipm1_alphas_cumprod[t], [6..7] 0, and for the hybrid KL term [8] posterior_mean_coef1[t], [9] posterior_mean_coef2[t],
// [10] posterior_variance[t], [11] posterior_log_variance_clipped[t]
#define DM_TRAIN_COEFS 12
int launch_pred_x_start(const float* x, const float* out, const float* coef_dev, float* xs, int B, int per_sample, int objective,
                        hipStream_t s);
int launch_q_sample(const float* x_start, const float* noise, const float* coef_dev, float* x, int B, int per_sample,
                    hipStream_t s);
int launch_mse_loss(const float* out, const float* x_start, const float* noise, const float* coef_dev, float* dout,
                    float* part, float* loss, int B, int per_sample, int objective, float loss_scale, hipStream_t s,
                    int terms = 1, const float* xq = nullptr, float* klpart = nullptr, float kl_scale = 0.f);
int launch_lerp(float* ema, const float* p, int64_t n, float decay, hipStream_t s);
// device mirrors of the host weight packers (pack_kernels.hip)
int launch_pack_direct(const float* oihw, float* packed, int Cout, int C0, int C1, int KH, int KW, hipStream_t s);
int launch_fold_taps(const float* oihw, float* w2, int Cout, int Cin, int py, int px, hipStream_t s);
int launch_pack_wino(const float* oihw, float* packed, int Cout, int Cin, hipStream_t s);
int launch_pack_wino4(const float* oihw, float* packed, int Cout, int Cin, hipStream_t s);
int launch_pack_upwino(const float* oihw, float* packed, int Cout, int Cin, hipStream_t s);
int launch_pack_pw(const float* w, float* packed, int Cout, int Cin, int s2d_C0, hipStream_t s);
int launch_pack_init7(const float* oihw, float* packed, int Cin, hipStream_t s);
// one job of a grouped re-pack (pack_kernels.hip: pack_jobs_kernel)
enum PackJobKind { PJ_COPY = 0, PJ_ROT, PJ_S2D_T, PJ_WINO, PJ_WINO4, PJ_UPWINO, PJ_PW };
struct PackJob {
    const float* src;
    float* dst;
    long long n;  // threads (elements) of the job
    int kind, Cout, Cin, a, b, first_block;  // a, b: PJ_ROT K, c_lo; PJ_PW s2d_C0
    int rs = 0, rc = 0;  // packs: rs != 0 reads the rotated / transposed weight of an input-gradient layer in place (load_taps9)
};
int pack_jobs_prefix(std::vector<PackJob>& jobs);
int launch_pack_jobs(const PackJob* jobs_dev, int n_jobs, int total_blocks, hipStream_t s);
int launch_rot_transpose(const float* w, float* out, int Cout, int Cin, int K, int c_lo, int c_n, hipStream_t s);
int launch_s2d_transpose(const float* w, float* out, int Cout, int C, hipStream_t s);
// table_dev: device array of {long long src_off; float* dst; long long n;}: dst[i] = param[src_off + i]
int launch_scatter_copy(const float* param, const void* table_dev, int n_entries, long long max_n, hipStream_t s);
int launch_grad_norm(const float* grads, int64_t n, double* part_ws, float max_norm, float* out2, hipStream_t s);
int launch_adam_ema(float* p, const float* g, float* m, float* v, float* ema, const float* clip2, int64_t n, float lr,
                    float b1, float b2, float eps, int step, float ema_decay, hipStream_t s);
size_t linattn_bwd_ws_floats(int B, int n, int heads);
int launch_linear_attention_core_bwd(const float* qkv, const float* mem_kv, const float* ctx, const float* dout, float* ws,
                                     float* dqkv, float* dmem_part, int B, int n, int heads, int dh, hipStream_t s,
                                     const float* kstats = nullptr);
// ws: attn_bwd_ws_floats() floats (row statistics of the tiled form; 0 floats when the sequence fits LDS)
size_t attn_bwd_ws_floats(int B, int nq, int nk, int n_mem, int heads);
int launch_attention_core_bwd(const float* qkv, const float* mem_kv, const float* dout, float* dqkv, float* dmem_part, float* ws,
                              int B, int n, int heads, int dh, hipStream_t s);
int launch_cross_attention_core_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                                    float* dv, float* ws, int B, int nq, int m, int heads, int dh, hipStream_t s);

}  // namespace dm

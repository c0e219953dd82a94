// C ABI of libdm_hip.so (include/dm_hip.h): U-Net handle, forward, samplers, single operators.
// Host orchestration only -- every FLOP runs in the kernels of conv_mfma.hip / attention.hip /
// elementwise.hip.  No allocation, copy or synchronisation happens inside the per-step launch
// sequence, so one denoise step can be captured into a hipGraph and replayed.
#include "../../include/dm_hip.h"
#include "dm_common.h"

#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace dm {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

// ---------------------------------------------------------------------------------------
// per-kernel event timing (bench.py roofline leg)
// ---------------------------------------------------------------------------------------
namespace prof {
struct Rec {
    std::string kernel;
    double flops, bytes;
    hipEvent_t e0, e1;
};
static bool g_on = false;
static bool g_detail = false;
// Interval of (event, EMPTY kernel, event): dispatch latency plus the two event packets, ~9 us on MI355X.  It is
// measured when profiling is switched on and subtracted from every bracketed launch, so that a row's time is the
// kernel's execution time -- what rocprofv3 --kernel-trace reports -- and not execution + dispatch.
static double g_overhead_ms = 0.0;
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_pool;

bool enabled() { return g_on; }
bool detail() { return g_detail; }

static int get_event(hipEvent_t* e) {
    if (!g_pool.empty()) {
        *e = g_pool.back();
        g_pool.pop_back();
        return 0;
    }
    DM_CHECK_HIP(hipEventCreate(e));
    return 0;
}

int begin(const char* kernel, double flops, double bytes, hipStream_t s) {
    Rec r{kernel, flops, bytes, nullptr, nullptr};
    if (get_event(&r.e0) || get_event(&r.e1)) return 1;
    DM_CHECK_HIP(hipEventRecord(r.e0, s));
    g_recs.push_back(r);
    return 0;
}

int end(hipStream_t s) {
    DM_REQUIRE(!g_recs.empty(), "prof::end without begin");
    DM_CHECK_HIP(hipEventRecord(g_recs.back().e1, s));
    return 0;
}
}  // namespace prof

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool set = false;
    bool dirty = false;  // changed by dm_unet_update_param since the device copy was packed
    size_t numel() const {
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        return n;
    }
};

// Recipe of one packed weight buffer (recorded while the host packs it): which parameter it comes from and which packer
// made it, so that the training loop can re-pack it on the DEVICE from the flat master parameters (pack_kernels.hip).
struct PackSrc {
    int kind = 0;  // 0: the parameter as stored; 1: rotated / transposed rows [c_lo, c_lo + c_n) of a (sCout, sCin, sK, sK)
                   // parameter (input-gradient convolution); 2: Downsample input-gradient transpose of a (sCout, 4 sCin) parameter
    std::string wname, bname;
    int sCout = 0, sCin = 0, sK = 0, c_lo = 0, c_n = 0;
};
enum PackKind { PK_RAW_W, PK_RAW_B, PK_RAW_NAME, PK_DIRECT, PK_FOLD, PK_WINO, PK_WINO4, PK_UPWINO, PK_PW, PK_PW_S2D, PK_INIT7 };
struct PackOp {
    int kind;
    float* dst;
    size_t n;          // floats of dst written by this op
    int Cout, C0, C1, KH, KW;
    PackSrc src;
    std::string name;  // PK_RAW_NAME: the parameter copied to dst
};

// Owner of every repacked weight buffer of a handle.  The first build allocates one buffer per upload; a REFRESH
// (dm_unet_refresh: new values for parameters of the same shapes) replays the same upload sequence into the same
// buffers, so every device pointer -- and with them a captured step graph -- stays valid.
struct DeviceOwner {
    std::vector<void*> ptrs;
    std::vector<size_t> sizes;
    bool refreshing = false;
    size_t cursor = 0;
    std::vector<PackOp>* rec = nullptr;  // when set, make_conv / up1 append the recipe of every buffer they upload
    PackSrc src;                         // the source of the convolution being packed (set by the caller of make_conv)
    void record(int kind, float* dst, size_t n, int Cout, int C0, int C1, int KH, int KW) {
        if (rec && !refreshing) rec->push_back(PackOp{kind, dst, n, Cout, C0, C1, KH, KW, src, std::string()});
    }
    void record_raw(const std::string& name, float* dst, size_t n) {
        if (rec && !refreshing) rec->push_back(PackOp{PK_RAW_NAME, dst, n, 0, 0, 0, 0, 0, PackSrc(), name});
    }
    ~DeviceOwner() { reset(); }
    DeviceOwner() = default;
    DeviceOwner(const DeviceOwner&) = delete;  // owns device memory
    DeviceOwner& operator=(const DeviceOwner&) = delete;
    void reset() {  // free every buffer (the input-gradient layers are rebuilt from scratch on a refresh)
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
        sizes.clear();
        cursor = 0;
    }
    int upload(const float* host, size_t n, float** out) {
        if (refreshing) {
            DM_REQUIRE(cursor < ptrs.size() && sizes[cursor] == n, "refresh: the upload sequence differs from the first build");
            void* p = ptrs[cursor++];
            if (n) DM_CHECK_HIP(hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice));
            *out = static_cast<float*>(p);
            return 0;
        }
        void* p = nullptr;
        DM_CHECK_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)));
        ptrs.push_back(p);
        sizes.push_back(n);
        if (n) DM_CHECK_HIP(hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice));
        *out = static_cast<float*>(p);
        return 0;
    }
};

// Workspace allocator over one device buffer.  Activations are released as soon as their last consumer has been
// ENQUEUED (the launch stream is in order, and so is a captured step graph), so a denoise step works in a few hundred
// megabytes that it keeps re-using instead of a fresh line of HBM per tensor: at the benchmark batch the blocks a
// layer reads were written a few launches earlier and are still in the 256 MB Infinity Cache.
// `dry` replays the same allocation sequence on fake addresses to measure the capacity (`peak`) a run needs; the
// sequence, hence every pointer, is a function of the shapes only -- which is what makes the step capturable.
struct Arena {
    char* base = nullptr;
    size_t cap = 0;
    bool dry = false;
    size_t off = 0;  // capacity required so far (high-water mark)
    struct Blk {
        size_t off, size;
        bool free;
    };
    std::vector<Blk> blks;  // sorted by offset, adjacent, covering [0, end of the last block)
    int n_free = 0;         // free blocks in `blks`: the tape forward of the training step releases almost nothing, and a
                            // best-fit scan over its ~1000 live blocks per allocation was a millisecond of host time per call
    char* origin() const { return dry ? reinterpret_cast<char*>(uintptr_t(1) << 44) : base; }
    float* alloc(size_t nfloats) {
        const size_t bytes = std::max<size_t>((nfloats * sizeof(float) + 255) & ~size_t(255), 256);
        int best = -1;
        if (n_free > 0)
            for (size_t i = 0; i < blks.size(); ++i)  // best fit
                if (blks[i].free && blks[i].size >= bytes && (best < 0 || blks[i].size < blks[best].size)) best = (int)i;
        size_t at;
        if (best >= 0) {
            Blk& b = blks[best];
            at = b.off;
            if (b.size > bytes) {
                const Blk rest{b.off + bytes, b.size - bytes, true};
                b.size = bytes;
                b.free = false;
                blks.insert(blks.begin() + best + 1, rest);  // one free block became a used one and a free rest
            } else {
                b.free = false;
                n_free -= 1;
            }
        } else if (!blks.empty() && blks.back().free) {  // grow the free tail
            at = blks.back().off;
            blks.back().size = bytes;
            blks.back().free = false;
            n_free -= 1;
        } else {
            at = blks.empty() ? 0 : blks.back().off + blks.back().size;
            blks.push_back(Blk{at, bytes, false});
        }
        off = std::max(off, blks.back().off + blks.back().size);
        return reinterpret_cast<float*>(origin() + at);
    }
    void release(const void* p) {
        static const bool off = std::getenv("DM_NO_WS_REUSE") != nullptr;  // measurement switch: one fresh block per tensor
        if (!p || off) return;
        const size_t at = (size_t)(static_cast<const char*>(p) - origin());
        // blocks are sorted by offset: binary search for the one that starts at `at`
        size_t lo = 0, hi = blks.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (blks[mid].off < at) lo = mid + 1;
            else hi = mid;
        }
        for (size_t i = lo; i < blks.size() && blks[i].off == at; ++i) {
            if (blks[i].free) continue;
            blks[i].free = true;
            n_free += 1;
            if (i + 1 < blks.size() && blks[i + 1].free) {
                blks[i].size += blks[i + 1].size;
                blks.erase(blks.begin() + i + 1);
                n_free -= 1;
            }
            if (i > 0 && blks[i - 1].free) {
                blks[i - 1].size += blks[i].size;
                blks.erase(blks.begin() + i);
                n_free -= 1;
            }
            return;
        }
    }
};

struct ConvLayer {
    int C0 = 0, C1 = 0, Cout = 0, KH = 1, KW = 1, stride = 1, pad = 0;
    int pad_w = -1;     // zero columns left / right when they differ from `pad` rows (1x7 / 7x1 convs of InceptionV3); -1: pad
    int pad_hi = 0;     // extra zero rows / columns at the bottom / right only (VAE Encoder Downsample, model.py:89-93)
    bool up = false;    // nearest x2 in front of the conv (Upsample, DD/denoising_diffusion.py:48-52)
    bool fold = false;  // ... executed as four 2x2 parity convs on the source grid (ConvParams::fold)
    int fold_w_stride = 0;
    float* w = nullptr;
    float* ww = nullptr;  // Winograd-transformed weights (winograd_mfma.hip) when the layer is eligible
    float* ww4 = nullptr;  // F(4x4,3x3) transformed weights (wino4_mfma.hip)
    float* wwu = nullptr;  // transformed weights of the upsample + 3x3 algorithm (upwino_mfma.hip)
    float* wpw = nullptr;  // lane-ordered weights of the 1x1 GEMM kernel (pw_mfma.hip)
    float* wi7 = nullptr;  // weights of the 7x7 first-conv kernel (init7_mfma.hip)
    float* wraw = nullptr;  // (Cout, Cin) weights of a 1x1 conv with Cout <= 4 (pointwise_small_kernel)
    float* bias = nullptr;
};

struct ResBlock {
    ConvLayer c1, c2, res;
    bool has_res = false;
    float *g1 = nullptr, *g2 = nullptr;
    int ss_off = 0, dout = 0;
};

struct AttnLayer {
    bool full = false;
    int dim = 0, heads = 0;  // heads: attn_heads of the layer's stage (cast_tuple(attn_heads, num_stages), :294)
    float *norm_g = nullptr, *mem_kv = nullptr, *out_g = nullptr;
    ConvLayer qkv, out;
    bool has_fused = false;  // LinearAttention as two fused kernels (linattn_fused.hip)
    bool has16 = false;      // full attention over 16 tokens as one kernel (attn16_fused.hip)
    Attn16 a16{};
    LinAttnFused fused{};
};

struct CrossLayer {
    ConvLayer q, out;
    float *wk = nullptr, *wv = nullptr, *g = nullptr;
    float* wo_raw = nullptr;  // to_out.0.weight as stored (dim, inner): the one-context-token path runs it per image
};

struct Stage {
    ResBlock b1, b2;
    AttnLayer attn;
    ConvLayer resample;
};

struct TrainState;  // dm_train.inc

}  // namespace dm

using namespace dm;

struct dm_unet {
    dm_unet_cfg cfg{};
    int device = 0;
    int init_dim = 0, out_dim = 0, time_dim = 0, heads = 0, dh = 0;
    std::vector<int> dims;
    std::map<std::string, HostTensor> params;  // expected entries, filled by set_param
    std::vector<std::string> order;
    bool finalized = false;
    bool poisoned = false;  // a refresh failed half-way: device weights are a mix of old and new values
    dm::TrainState* train = nullptr;  // dm_unet_train_enable: gradient buffers, dgrad weights, training workspace
    std::vector<dm::PackOp> pack_ops;  // recipe of every buffer the convolutions / norms read (device-side re-packing)
    bool infer_stale = false;          // device-side optimiser steps since the fused inference-only packs were built
    DeviceOwner own;
    // layers
    ConvLayer init_conv, final_conv;
    std::vector<Stage> downs, ups;
    ResBlock mid1, mid2, final_res;
    AttnLayer mid_attn;
    CrossLayer cross_down, cross_mid, cross_up;
    float *freqs = nullptr, *tw1 = nullptr, *tb1 = nullptr, *tw2 = nullptr, *tb2 = nullptr;
    float *ss_w = nullptr, *ss_b = nullptr;
    int ss_total = 0;
    float *tp_w0 = nullptr, *tp_b0 = nullptr, *tp_w2 = nullptr, *tp_b2 = nullptr, *tc_w = nullptr, *tc_b = nullptr;
    // workspace
    char* ws = nullptr;
    size_t ws_cap = 0;
    // sampler state (device)
    SamplerState* state_dev = nullptr;
    int64_t* times_dev = nullptr;
    float* coefs_dev = nullptr;
    int sampler_cap = 0;
    // weight groups: parameter-name prefix -> [first upload, count) in `own` (dm_unet_refresh skips clean groups)
    std::map<std::string, std::pair<size_t, size_t>> groups;
    std::vector<std::pair<std::string, ResBlock*>> resnets;  // in ss_off order
    // the instantiated graph of one denoise step, reused while the key (shape, kind, every captured pointer) holds
    struct GraphKey {
        int kind = -1, B = 0, H = 0, W = 0, ctx_tokens = 0, cond_channels = 0, objective = 0, self_cond = 0;
        const void *noise = nullptr, *all_steps = nullptr, *ws = nullptr, *times = nullptr, *coefs = nullptr;
        bool operator==(const GraphKey& o) const {
            return kind == o.kind && B == o.B && H == o.H && W == o.W && ctx_tokens == o.ctx_tokens &&
                   cond_channels == o.cond_channels && objective == o.objective && self_cond == o.self_cond && noise == o.noise && all_steps == o.all_steps && ws == o.ws &&
                   times == o.times && coefs == o.coefs;
        }
    } gkey;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    // Workspace, sampler state and the cached graph are shared by every call on the handle: a call waits for the previous
    // call's last kernel (recorded here) before it touches them, so two calls on DIFFERENT streams are still ordered.
    hipEvent_t done_ev = nullptr;
    bool done_recorded = false;
    int order_after_previous(hipStream_t s) {
        if (!done_ev) DM_CHECK_HIP(hipEventCreateWithFlags(&done_ev, hipEventDisableTiming));
        if (done_recorded) DM_CHECK_HIP(hipStreamWaitEvent(s, done_ev, 0));
        return 0;
    }
    int mark_done(hipStream_t s) {
        DM_CHECK_HIP(hipEventRecord(done_ev, s));
        done_recorded = true;
        return 0;
    }
    // Independent branches of a step (res_conv next to block1, the time MLP next to init_conv) are enqueued on a second
    // stream between a fork and a join event, eagerly and inside the captured step graph alike: at small per-GPU batches no
    // kernel fills the chip and the branch costs nothing.  Events come from a pool (distinct objects within one capture).
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_next = 0;
    int next_event(hipEvent_t* e) {
        if (ev_pool.size() < 64) {
            hipEvent_t n = nullptr;
            DM_CHECK_HIP(hipEventCreateWithFlags(&n, hipEventDisableTiming));
            ev_pool.push_back(n);
            *e = n;
            return 0;
        }
        *e = ev_pool[ev_next++ % ev_pool.size()];
        return 0;
    }
    hipStream_t cap_stream = nullptr;  // capture / replay stream when the caller passes the legacy default stream
    int graph_captures = 0;            // diagnostics (dm_unet_graph_captures)
    void drop_graph() {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
        gexec = nullptr;
        graph = nullptr;
        gkey = GraphKey{};
    }
};

namespace dm {

void free_train(dm_unet* u);  // dm_train.inc
static int ensure_packed(dm_unet* u, const float* w, hipStream_t s);
static int build_train(dm_unet* u);
static int upload_master_if_resident(dm_unet* u);

static void expect(dm_unet* u, const std::string& name, std::vector<int64_t> shape) {
    HostTensor t;
    t.shape = std::move(shape);
    u->params[name] = std::move(t);
    u->order.push_back(name);
}

static void expect_resnet(dm_unet* u, const std::string& p, int din, int dout) {
    int td = u->time_dim;
    expect(u, p + ".mlp.1.weight", {2 * dout, td});
    expect(u, p + ".mlp.1.bias", {2 * dout});
    expect(u, p + ".block1.proj.weight", {dout, din, 3, 3});
    expect(u, p + ".block1.proj.bias", {dout});
    expect(u, p + ".block1.norm.g", {1, dout, 1, 1});
    expect(u, p + ".block2.proj.weight", {dout, dout, 3, 3});
    expect(u, p + ".block2.proj.bias", {dout});
    expect(u, p + ".block2.norm.g", {1, dout, 1, 1});
    if (din != dout) {
        expect(u, p + ".res_conv.weight", {dout, din, 1, 1});
        expect(u, p + ".res_conv.bias", {dout});
    }
}

// attn_heads of stage i (Unet(attn_heads = (..)), :294: cast_tuple over the stages; mid_attn takes the last stage's, :324)
static int stage_heads(const dm_unet_cfg& cfg, int i) { return cfg.attn_heads_stage[i] > 0 ? cfg.attn_heads_stage[i] : cfg.attn_heads; }

static void expect_attn(dm_unet* u, const std::string& p, int dim, bool full, int heads) {
    int hidden = heads * u->dh;
    if (full) {
        expect(u, p + ".mem_kv", {2, heads, 4, u->dh});
        expect(u, p + ".norm.g", {1, dim, 1, 1});
        expect(u, p + ".to_qkv.weight", {3 * hidden, dim, 1, 1});
        expect(u, p + ".to_out.weight", {dim, hidden, 1, 1});
        expect(u, p + ".to_out.bias", {dim});
    } else {
        expect(u, p + ".mem_kv", {2, heads, u->dh, 4});
        expect(u, p + ".norm.g", {1, dim, 1, 1});
        expect(u, p + ".to_qkv.weight", {3 * hidden, dim, 1, 1});
        expect(u, p + ".to_out.0.weight", {dim, hidden, 1, 1});
        expect(u, p + ".to_out.0.bias", {dim});
        expect(u, p + ".to_out.1.g", {1, dim, 1, 1});
    }
}

static void expect_cross(dm_unet* u, const std::string& p, int dim) {
    int inner = 4 * u->dh;
    int ctx = u->cfg.text_emb_dim;
    expect(u, p + ".to_q.weight", {inner, dim});
    expect(u, p + ".to_k.weight", {inner, ctx});
    expect(u, p + ".to_v.weight", {inner, ctx});
    expect(u, p + ".to_out.0.weight", {dim, inner});
    expect(u, p + ".to_out.0.bias", {dim});
    expect(u, p + ".to_out.1.g", {1, dim});
}

static std::string idx(const std::string& a, int i) { return a + "." + std::to_string(i); }

// ---- weight upload ----------------------------------------------------------------------
static int make_conv(DeviceOwner& own, ConvLayer& L, const float* oihw, const float* bias, int Cout, int C0, int C1, int KH,
                     int KW, int stride, int pad, bool up, int pad_hi = 0) {
    L.C0 = C0; L.C1 = C1; L.Cout = Cout; L.KH = KH; L.KW = KW; L.stride = stride; L.pad = pad; L.up = up;
    L.pad_hi = pad_hi;
    static const bool no_fold = std::getenv("DM_NO_UPFOLD") != nullptr;
    L.fold = up && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !no_fold;
    std::vector<float> packed;
    if (L.fold) {
        // nearest x2 then conv3x3 == one 2x2 conv per output parity with summed taps:
        //   parity 0 along an axis: source offsets {-1, 0} <- taps {0}, {1,2};  parity 1: {0, +1} <- taps {0,1}, {2}
        const int Cin = C0 + C1;
        const size_t per = conv_packed_floats(Cout, C0, C1, 2, 2);
        packed.resize(4 * per);
        std::vector<float> w2((size_t)Cout * Cin * 4);
        auto taps = [](int parity, int a, int* lo, int* hi) {
            if (parity == 0) { *lo = a == 0 ? 0 : 1; *hi = a == 0 ? 0 : 2; }
            else             { *lo = a == 0 ? 0 : 2; *hi = a == 0 ? 1 : 2; }
        };
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                for (size_t oc = 0; oc < (size_t)Cout * Cin; ++oc)
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            int y0, y1, x0, x1;
                            taps(py, a, &y0, &y1);
                            taps(px, b, &x0, &x1);
                            float sacc = 0.f;
                            for (int dy = y0; dy <= y1; ++dy)
                                for (int dx = x0; dx <= x1; ++dx) sacc += oihw[oc * 9 + dy * 3 + dx];
                            w2[oc * 4 + a * 2 + b] = sacc;
                        }
                conv_pack_weights(w2.data(), packed.data() + (size_t)(py * 2 + px) * per, Cout, C0, C1, 2, 2);
            }
        L.fold_w_stride = (int)per;
    } else {
        packed.resize(conv_packed_floats(Cout, C0, C1, KH, KW));
        conv_pack_weights(oihw, packed.data(), Cout, C0, C1, KH, KW);
    }
    if (own.upload(packed.data(), packed.size(), &L.w)) return 1;
    own.record(L.fold ? PK_FOLD : PK_DIRECT, L.w, packed.size(), Cout, C0, C1, KH, KW);
    L.ww = nullptr;
    if (wino_eligible(Cout, C0, C1, KH, KW, stride, pad, up)) {
        std::vector<float> wp(wino_packed_floats(Cout, C0, C1));
        wino_pack_weights(oihw, wp.data(), Cout, C0, C1);
        if (own.upload(wp.data(), wp.size(), &L.ww)) return 1;
        own.record(PK_WINO, L.ww, wp.size(), Cout, C0, C1, KH, KW);
    }
    L.ww4 = nullptr;
    if (wino4_eligible(Cout, C0, C1, KH, KW, stride, pad, up)) {
        std::vector<float> wp(wino4_packed_floats(Cout, C0, C1));
        wino4_pack_weights(oihw, wp.data(), Cout, C0, C1);
        if (own.upload(wp.data(), wp.size(), &L.ww4)) return 1;
        own.record(PK_WINO4, L.ww4, wp.size(), Cout, C0, C1, KH, KW);
    }
    L.wwu = nullptr;
    if (upwino_eligible(Cout, C0, C1, KH, KW, stride, pad, up)) {
        std::vector<float> wp(upwino_packed_floats(Cout, C0, C1));
        upwino_pack_weights(oihw, wp.data(), Cout, C0, C1);
        if (own.upload(wp.data(), wp.size(), &L.wwu)) return 1;
        own.record(PK_UPWINO, L.wwu, wp.size(), Cout, C0, C1, KH, KW);
    }
    L.wpw = nullptr;
    if (pw_eligible(Cout, C0, C1, KH, KW, stride, pad, up)) {
        std::vector<float> wp(pw_packed_floats(Cout, C0, C1));
        pw_pack_weights(oihw, wp.data(), Cout, C0, C1);
        if (own.upload(wp.data(), wp.size(), &L.wpw)) return 1;
        own.record(PK_PW, L.wpw, wp.size(), Cout, C0, C1, KH, KW);
    }
    if (pw_s2d_eligible(Cout, C0, C1, KH, KW, stride, pad, up)) {
        std::vector<float> wp((size_t)4 * C0 * Cout);
        pw_pack_weights_s2d(oihw, wp.data(), Cout, C0);
        if (own.upload(wp.data(), wp.size(), &L.wpw)) return 1;
        own.record(PK_PW_S2D, L.wpw, wp.size(), Cout, C0, C1, KH, KW);
    }
    L.wi7 = nullptr;
    if (init7_eligible(Cout, C0, C1, KH, KW, stride, pad, up)) {
        std::vector<float> wp(init7_packed_floats(C0));
        init7_pack_weights(oihw, wp.data(), C0);
        if (own.upload(wp.data(), wp.size(), &L.wi7)) return 1;
        own.record(PK_INIT7, L.wi7, wp.size(), Cout, C0, C1, KH, KW);
    }
    L.wraw = nullptr;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && !up && Cout <= 4 && C1 == 0 && C0 % 4 == 0) {
        if (own.upload(oihw, (size_t)Cout * C0, &L.wraw)) return 1;
        own.record(PK_RAW_W, L.wraw, (size_t)Cout * C0, Cout, C0, C1, KH, KW);
    }
    L.bias = nullptr;
    if (bias) {
        if (own.upload(bias, Cout, &L.bias)) return 1;
        own.record(PK_RAW_B, L.bias, Cout, Cout, C0, C1, KH, KW);
    }
    return 0;
}

static const HostTensor& P(dm_unet* u, const std::string& n) { return u->params.at(n); }

static int up1(dm_unet* u, const std::string& n, float** out) {
    const HostTensor& t = P(u, n);
    if (u->own.upload(t.data.data(), t.data.size(), out)) return 1;
    u->own.record_raw(n, *out, t.data.size());
    return 0;
}
// the parameters make_conv is about to pack (for the device-side re-packing recipe)
static void conv_src(dm_unet* u, const std::string& wname, const std::string& bname) {
    u->own.src = PackSrc();
    u->own.src.wname = wname;
    u->own.src.bname = bname;
}

// One weight group = everything packed from the parameters under one name prefix.  First build: run `build` and
// record which uploads it made.  Refresh: re-run it into the same buffers if any of its parameters changed, else skip.
static bool prefix_dirty(dm_unet* u, const std::string& prefix) {
    for (auto it = u->params.lower_bound(prefix); it != u->params.end(); ++it) {
        const std::string& k = it->first;
        if (k.compare(0, prefix.size(), prefix) != 0) break;
        if ((k.size() == prefix.size() || k[prefix.size()] == '.') && it->second.dirty) return true;
    }
    return false;
}
template <class F>
static int weight_group(dm_unet* u, const std::string& prefix, F build) {
    DeviceOwner& own = u->own;
    if (own.refreshing) {
        auto it = u->groups.find(prefix);
        DM_REQUIRE(it != u->groups.end(), "refresh: unknown weight group");
        DM_REQUIRE(own.cursor == it->second.first, "refresh: weight groups out of order");
        if (!prefix_dirty(u, prefix)) {
            own.cursor += it->second.second;
            return 0;
        }
        if (build()) return 1;
        DM_REQUIRE(own.cursor == it->second.first + it->second.second, "refresh: upload count differs from the first build");
        return 0;
    }
    const size_t start = own.ptrs.size();
    if (build()) return 1;
    u->groups[prefix] = {start, own.ptrs.size() - start};
    return 0;
}

static int build_resnet(dm_unet* u, ResBlock& R, const std::string& p, int C0, int C1, int dout, int& ss_off) {
    // the mlp (SiLU -> Linear) of every ResnetBlock is one row block of the concatenated [ss_total][time_dim] matrix
    // assembled by build_scale_shift
    R.dout = dout;
    R.ss_off = ss_off;
    ss_off += 2 * dout;
    if (!u->own.refreshing) u->resnets.emplace_back(p, &R);
    return weight_group(u, p, [&]() -> int {
        int din = C0 + C1;
        conv_src(u, p + ".block1.proj.weight", p + ".block1.proj.bias");
        if (make_conv(u->own, R.c1, P(u, p + ".block1.proj.weight").data.data(), P(u, p + ".block1.proj.bias").data.data(),
                      dout, C0, C1, 3, 3, 1, 1, false)) return 1;
        conv_src(u, p + ".block2.proj.weight", p + ".block2.proj.bias");
        if (make_conv(u->own, R.c2, P(u, p + ".block2.proj.weight").data.data(), P(u, p + ".block2.proj.bias").data.data(),
                      dout, dout, 0, 3, 3, 1, 1, false)) return 1;
        if (up1(u, p + ".block1.norm.g", &R.g1) || up1(u, p + ".block2.norm.g", &R.g2)) return 1;
        R.has_res = din != dout;
        if (R.has_res) {
            conv_src(u, p + ".res_conv.weight", p + ".res_conv.bias");
            if (make_conv(u->own, R.res, P(u, p + ".res_conv.weight").data.data(), P(u, p + ".res_conv.bias").data.data(),
                          dout, C0, C1, 1, 1, 1, 0, false)) return 1;
        }
        return 0;
    });
}

static int build_scale_shift(dm_unet* u) {
    bool dirty = false;
    for (auto& pr : u->resnets) dirty = dirty || P(u, pr.first + ".mlp.1.weight").dirty || P(u, pr.first + ".mlp.1.bias").dirty;
    if (u->own.refreshing && !dirty) {
        u->own.cursor += 2;
        return 0;
    }
    std::vector<float> ssw, ssb;
    for (auto& pr : u->resnets) {
        const HostTensor& mw = P(u, pr.first + ".mlp.1.weight");
        const HostTensor& mb = P(u, pr.first + ".mlp.1.bias");
        ssw.insert(ssw.end(), mw.data.begin(), mw.data.end());
        ssb.insert(ssb.end(), mb.data.begin(), mb.data.end());
    }
    if (u->own.upload(ssw.data(), ssw.size(), &u->ss_w) || u->own.upload(ssb.data(), ssb.size(), &u->ss_b)) return 1;
    size_t ow = 0, ob = 0;  // recipe: every ResnetBlock.mlp is a row block of the concatenated matrix
    for (auto& pr : u->resnets) {
        const HostTensor& mw = P(u, pr.first + ".mlp.1.weight");
        const HostTensor& mb = P(u, pr.first + ".mlp.1.bias");
        u->own.record_raw(pr.first + ".mlp.1.weight", u->ss_w + ow, mw.data.size());
        u->own.record_raw(pr.first + ".mlp.1.bias", u->ss_b + ob, mb.data.size());
        ow += mw.data.size();
        ob += mb.data.size();
    }
    return 0;
}

static int build_attn_body(dm_unet* u, AttnLayer& A, const std::string& p, int dim, bool full, int heads) {
    A.full = full;
    A.dim = dim;
    A.heads = heads;
    int hidden = heads * u->dh;
    if (up1(u, p + ".norm.g", &A.norm_g) || up1(u, p + ".mem_kv", &A.mem_kv)) return 1;
    conv_src(u, p + ".to_qkv.weight", "");
    if (make_conv(u->own, A.qkv, P(u, p + ".to_qkv.weight").data.data(), nullptr, 3 * hidden, dim, 0, 1, 1, 1, 0, false))
        return 1;
    if (full) {
        conv_src(u, p + ".to_out.weight", p + ".to_out.bias");
        if (make_conv(u->own, A.out, P(u, p + ".to_out.weight").data.data(), P(u, p + ".to_out.bias").data.data(), dim,
                      hidden, 0, 1, 1, 1, 0, false)) return 1;
        A.has16 = false;
        if (attn16_eligible(dim, heads, u->dh)) {
            std::vector<float> wp, wo;
            attn16_pack(P(u, p + ".to_qkv.weight").data.data(), P(u, p + ".norm.g").data.data(),
                        P(u, p + ".to_out.weight").data.data(), dim, wp, wo);
            float *dwp, *dwo;
            if (u->own.upload(wp.data(), wp.size(), &dwp) || u->own.upload(wo.data(), wo.size(), &dwo)) return 1;
            A.a16 = Attn16{dim, dwp, dwo, A.out.bias, A.mem_kv};
            A.has16 = true;
        }
    } else {
        conv_src(u, p + ".to_out.0.weight", p + ".to_out.0.bias");
        if (make_conv(u->own, A.out, P(u, p + ".to_out.0.weight").data.data(), P(u, p + ".to_out.0.bias").data.data(),
                      dim, hidden, 0, 1, 1, 1, 0, false)) return 1;
        if (up1(u, p + ".to_out.1.g", &A.out_g)) return 1;
        if (linattn_fused_eligible(dim, heads, u->dh)) {
            std::vector<float> wq, wk, wv, wo, kb;
            const HostTensor& og = P(u, p + ".to_out.1.g");
            if (linattn_fused_pack(P(u, p + ".to_qkv.weight").data.data(), P(u, p + ".norm.g").data.data(),
                                   P(u, p + ".to_out.0.weight").data.data(), P(u, p + ".mem_kv").data.data(), dim, wq,
                                   wk, wv, wo, kb)) {
                std::vector<float> ogs(og.data);
                for (float& v : ogs) v *= std::sqrt((float)dim);
                float *dq, *dk, *dv, *dwo, *dkb, *dog;
                const HostTensor& wor = P(u, p + ".to_out.0.weight");
                if (u->own.upload(wq.data(), wq.size(), &dq) || u->own.upload(wk.data(), wk.size(), &dk) ||
                    u->own.upload(wv.data(), wv.size(), &dv) || u->own.upload(wor.data.data(), wor.data.size(), &dwo) ||
                    u->own.upload(kb.data(), kb.size(), &dkb) || u->own.upload(ogs.data(), ogs.size(), &dog))
                    return 1;
                A.fused = LinAttnFused{dim, dq, dk, dv, dwo, A.out.bias, dog, dkb, A.mem_kv};
                A.has_fused = true;
            }
        }
    }
    return 0;
}

static int build_attn(dm_unet* u, AttnLayer& A, const std::string& p, int dim, bool full, int heads) {
    return weight_group(u, p, [&]() -> int { return build_attn_body(u, A, p, dim, full, heads); });
}

static int build_cross_body(dm_unet* u, CrossLayer& C, const std::string& p, int dim) {
    int inner = 4 * u->dh;
    // nn.Linear weights [out, in] are 1x1 conv weights (out, in, 1, 1)
    conv_src(u, p + ".to_q.weight", "");
    if (make_conv(u->own, C.q, P(u, p + ".to_q.weight").data.data(), nullptr, inner, dim, 0, 1, 1, 1, 0, false)) return 1;
    conv_src(u, p + ".to_out.0.weight", p + ".to_out.0.bias");
    if (make_conv(u->own, C.out, P(u, p + ".to_out.0.weight").data.data(), P(u, p + ".to_out.0.bias").data.data(), dim,
                  inner, 0, 1, 1, 1, 0, false)) return 1;
    if (up1(u, p + ".to_k.weight", &C.wk) || up1(u, p + ".to_v.weight", &C.wv) || up1(u, p + ".to_out.1.g", &C.g) ||
        up1(u, p + ".to_out.0.weight", &C.wo_raw))
        return 1;
    return 0;
}

static int build_cross(dm_unet* u, CrossLayer& C, const std::string& p, int dim) {
    return weight_group(u, p, [&]() -> int { return build_cross_body(u, C, p, dim); });
}

// Pack every parameter into its kernel layout and upload it: the first build (dm_unet_finalize) and, with
// own.refreshing set, the in-place refresh of changed parameters (dm_unet_refresh).
static int build_all(dm_unet* u) {
    const dm_unet_cfg& cfg = u->cfg;
    const int n = cfg.n_stages;
    // sinusoid frequencies, fp32 as the reference computes them (DD/denoising_diffusion.py:79-81)
    if (cfg.learned_sinusoidal_dim > 0) {
        // RandomOrLearnedSinusoidalPosEmb (:86-101): the frequencies are the parameter time_mlp.0.weights
        if (weight_group(u, "time_mlp.0", [&]() -> int { return up1(u, "time_mlp.0.weights", &u->freqs); })) return 1;
    } else if (weight_group(u, "#freqs", [&]() -> int {
            int half = cfg.dim / 2;
            double k = std::log((double)cfg.sinusoidal_theta) / (half - 1);
            float kf = (float)(-k);
            std::vector<float> f(half);
            for (int i = 0; i < half; ++i) f[i] = std::exp((float)i * kf);
            return u->own.upload(f.data(), f.size(), &u->freqs);
        }))
        return 1;
    if (weight_group(u, "time_mlp", [&]() -> int {
            return up1(u, "time_mlp.1.weight", &u->tw1) || up1(u, "time_mlp.1.bias", &u->tb1) ||
                   up1(u, "time_mlp.3.weight", &u->tw2) || up1(u, "time_mlp.3.bias", &u->tb2);
        }))
        return 1;
    if (weight_group(u, "init_conv", [&]() -> int {
            conv_src(u, "init_conv.weight", "init_conv.bias");
            return make_conv(u->own, u->init_conv, P(u, "init_conv.weight").data.data(), P(u, "init_conv.bias").data.data(),
                             u->init_dim, cfg.input_channels, 0, 7, 7, 1, 3, false);
        }))
        return 1;
    int ss_off = 0;
    u->downs.resize(n);
    u->ups.resize(n);
    for (int i = 0; i < n; ++i) {
        int din = u->dims[i], dout = u->dims[i + 1];
        std::string q = idx("downs", i);
        Stage& S = u->downs[i];
        if (build_resnet(u, S.b1, q + ".0", din, 0, din, ss_off)) return 1;
        if (build_resnet(u, S.b2, q + ".1", din, 0, din, ss_off)) return 1;
        if (build_attn(u, S.attn, q + ".2", din, cfg.full_attn[i] != 0, stage_heads(cfg, i))) return 1;
        if (weight_group(u, q + ".3", [&]() -> int {
                conv_src(u, i < n - 1 ? q + ".3.1.weight" : q + ".3.weight", i < n - 1 ? q + ".3.1.bias" : q + ".3.bias");
                if (i < n - 1) {
                    // pixel-unshuffle + 1x1  ==  2x2 stride-2 conv: W'[o][c][p1][p2] = W[o][c*4 + p1*2 + p2]
                    return make_conv(u->own, S.resample, P(u, q + ".3.1.weight").data.data(),
                                     P(u, q + ".3.1.bias").data.data(), dout, din, 0, 2, 2, 2, 0, false);
                }
                return make_conv(u->own, S.resample, P(u, q + ".3.weight").data.data(), P(u, q + ".3.bias").data.data(),
                                 dout, din, 0, 3, 3, 1, 1, false);
            }))
            return 1;
    }
    int mid = u->dims.back();
    if (build_resnet(u, u->mid1, "mid_block1", mid, 0, mid, ss_off)) return 1;
    if (build_attn(u, u->mid_attn, "mid_attn", mid, true, stage_heads(cfg, n - 1))) return 1;
    if (build_resnet(u, u->mid2, "mid_block2", mid, 0, mid, ss_off)) return 1;
    for (int j = 0; j < n; ++j) {
        int din = u->dims[n - 1 - j], dout = u->dims[n - j];
        std::string q = idx("ups", j);
        Stage& S = u->ups[j];
        if (build_resnet(u, S.b1, q + ".0", dout, din, dout, ss_off)) return 1;
        if (build_resnet(u, S.b2, q + ".1", dout, din, dout, ss_off)) return 1;
        if (build_attn(u, S.attn, q + ".2", dout, cfg.full_attn[n - 1 - j] != 0, stage_heads(cfg, n - 1 - j))) return 1;
        if (weight_group(u, q + ".3", [&]() -> int {
                conv_src(u, j < n - 1 ? q + ".3.1.weight" : q + ".3.weight", j < n - 1 ? q + ".3.1.bias" : q + ".3.bias");
                if (j < n - 1)
                    return make_conv(u->own, S.resample, P(u, q + ".3.1.weight").data.data(),
                                     P(u, q + ".3.1.bias").data.data(), din, dout, 0, 3, 3, 1, 1, /*up=*/true);
                return make_conv(u->own, S.resample, P(u, q + ".3.weight").data.data(), P(u, q + ".3.bias").data.data(),
                                 din, dout, 0, 3, 3, 1, 1, false);
            }))
            return 1;
    }
    if (build_resnet(u, u->final_res, "final_res_block", u->init_dim, u->init_dim, u->init_dim, ss_off)) return 1;
    if (weight_group(u, "final_conv", [&]() -> int {
            conv_src(u, "final_conv.weight", "final_conv.bias");
            return make_conv(u->own, u->final_conv, P(u, "final_conv.weight").data.data(),
                             P(u, "final_conv.bias").data.data(), u->out_dim, u->init_dim, 0, 1, 1, 1, 0, false);
        }))
        return 1;
    u->ss_total = ss_off;
    if (build_scale_shift(u)) return 1;
    if (cfg.text_mode == DM_TEXT_CONCAT) {
        if (weight_group(u, "text_proj", [&]() -> int {
                return up1(u, "text_proj.0.weight", &u->tp_w0) || up1(u, "text_proj.0.bias", &u->tp_b0) ||
                       up1(u, "text_proj.2.weight", &u->tp_w2) || up1(u, "text_proj.2.bias", &u->tp_b2);
            }) ||
            weight_group(u, "text_concat_proj", [&]() -> int {
                return up1(u, "text_concat_proj.weight", &u->tc_w) || up1(u, "text_concat_proj.bias", &u->tc_b);
            }))
            return 1;
    } else if (cfg.text_mode == DM_TEXT_CROSS) {
        if (build_cross(u, u->cross_mid, "cross_attn", mid) || build_cross(u, u->cross_down, "cross_attn_down", mid) ||
            build_cross(u, u->cross_up, "cross_attn_up", mid))
            return 1;
    }
    return 0;
}

// ---- launches ---------------------------------------------------------------------------
struct Ctx {
    dm_unet* u;
    Arena* A;
    hipStream_t s;
    int B;
    const float* ss;   // [Bt][ss_total]
    int ss_stride;     // 0 when one row serves the whole batch
    bool par = false;  // independent branches go to the handle's second stream (fork / join below)
    bool dry() const { return A->dry; }
};

// fork: everything enqueued on the side stream from here on runs after what `c.s` holds now, next to what `c.s` gets next
static int fork_side(Ctx& c, hipStream_t* side) {
    dm_unet* u = c.u;
    if (!u->side) DM_CHECK_HIP(hipStreamCreateWithFlags(&u->side, hipStreamNonBlocking));
    hipEvent_t e;
    if (u->next_event(&e)) return 1;
    DM_CHECK_HIP(hipEventRecord(e, c.s));
    DM_CHECK_HIP(hipStreamWaitEvent(u->side, e, 0));
    *side = u->side;
    return 0;
}
// join: what `c.s` gets next runs after everything the side stream holds
static int join_side(Ctx& c) {
    dm_unet* u = c.u;
    hipEvent_t e;
    if (u->next_event(&e)) return 1;
    DM_CHECK_HIP(hipEventRecord(e, u->side));
    DM_CHECK_HIP(hipStreamWaitEvent(c.s, e, 0));
    return 0;
}
// Which steps fork.  Measured on MI355X (profiles/r3_fork_join_ab.txt, hipGraph replay): forking makes every batch SLOWER
// -- B=8 @64x64 1.669 -> 1.753 ms per step, B=32 2.858 -> 2.924, B=64 @32x32 1.984 -> 2.042, B=256 4.551 -> 4.608 -- the
// cross-queue dependencies of a forked graph (one signal wait per fork and per join, ~20 per step) cost more than the
// overlap of a 17 us res_conv or the 20 us time-MLP chain buys.  So the default is OFF; DM_PAR=1 switches it on (the tests
// run one model with it), DM_PAR_MAX_PIXELS limits it to small steps.
static bool par_policy(int B, int H, int W) {
    static const int mode = env_int("DM_PAR", 0);
    static const int max_px = env_int("DM_PAR_MAX_PIXELS", 1 << 30);
    return mode != 0 && (int64_t)B * H * W <= max_px;
}

// policy: which eligible layers take the Winograd kernel (DM_WINO=0 none, 1 all)
static bool wino_use(int B, int Ho, int Wo, int Cout, int C0, int C1) {
    static const int mode = std::getenv("DM_WINO") ? std::atoi(std::getenv("DM_WINO")) : 1;
    return mode != 0 && wino_shape_ok(B, Ho, Wo, Cout, C0, C1);
}

// residual operand of a landing pass given as the K-split partial sums of another convolution (+ its bias)
struct ResParts {
    const float* part;
    int nsplit;
    int64_t stride;
    const float* bias;
};

struct PlannedConv {
    ConvParams p;      // everything but the tensor pointers
    int out_h, out_w;  // dims of the output tensor
    int kind;          // 0 direct, 1 Winograd F(2x2,3x3), 2 F(4x4,3x3), 3 upsample algorithm, 4 one thread per pixel, 5 1x1 GEMM, 7 7x7 first conv
    bool in_kernel;    // the epilogue runs inside the conv kernel (else: partial sums + landing kernel)
};

// Which kernel and tiling a convolution takes; a function of the layer, the batch and the image size only.
static int plan_conv(const Ctx& c, const ConvLayer& L, bool has_in1, int Hin, int Win, int epi, bool in_nchw, bool out_nchw,
                     PlannedConv& P) {
    ConvParams& p = P.p;
    p = ConvParams{};
    p.C0 = L.C0; p.C1 = L.C1;
    const int CK = conv_ck_for(L.C0, L.C1);
    p.chunks0 = (L.C0 + CK - 1) / CK;
    p.n_chunks = p.chunks0 + (L.C1 ? (L.C1 + CK - 1) / CK : 0);
    p.in_nchw = in_nchw ? 1 : 0;
    p.w = L.w; p.bias = L.bias;
    p.Cout = L.Cout; p.stride = L.stride; p.pad = L.pad;
    const int padw = L.pad_w >= 0 ? L.pad_w : L.pad;
    p.pad_w = padw;
    p.B = c.B;
    p.out_nchw = out_nchw ? 1 : 0;
    p.ss_stride = c.ss_stride;
    const bool want_norm = (epi & EPI_NORM) != 0;
    P.kind = 0;
    const bool upwino = L.up && L.wwu && !in_nchw && !out_nchw && !has_in1 && Hin % 2 == 0 && Win % 2 == 0 &&
                        upwino_shape_ok(c.B, Hin / 2, Win / 2, L.Cout, L.C0, L.C1);
    if (upwino) {
        // (Hin, Win) is the upsampled size the caller sees; the kernel works on the source grid (upwino_mfma.hip)
        p.fold = 0; p.fold_w_stride = 0; p.up = 1;
        p.KH = 3; p.KW = 3;
        p.Hin = Hin / 2; p.Win = Win / 2;
        p.Ho = Hin; p.Wo = Win;
        P.out_h = Hin; P.out_w = Win;
        p.w = L.wwu;
        p.chunks0 = p.n_chunks = L.C0 / 8;
        p.geo = upwino_plan(c.B, p.Hin, p.Win, L.Cout, L.C0, L.C1, true);
        P.kind = 3;
    } else if (L.fold) {
        // (Hin, Win) is the upsampled size the caller sees; the four parity convs run on the source grid
        DM_REQUIRE(!out_nchw && !in_nchw, "folded upsample conv is NHWC");
        p.fold = 1; p.fold_w_stride = L.fold_w_stride; p.up = 0;
        p.KH = 2; p.KW = 2; p.pad_w = 1;
        p.Hin = Hin / 2; p.Win = Win / 2;
        p.Ho = p.Hin; p.Wo = p.Win;
        P.out_h = Hin; P.out_w = Win;
        p.geo = conv_plan(c.B, p.Ho, p.Wo, L.Cout, 2, 2, 1, L.C0, L.C1, want_norm, true, 4);
    } else if (L.KH == 2 && L.KW == 2 && L.stride == 2 && L.pad == 0 && !L.up && L.C1 == 0 && L.C0 % 16 == 0 &&
               !in_nchw && Hin % 2 == 0 && Win % 2 == 0) {
        // Downsample: space-to-depth, then a 1x1 convolution over 4*C0 channels (ConvParams::s2d)
        p.fold = 0; p.fold_w_stride = 0; p.up = 0; p.s2d = 1;
        p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.pad_w = 0;
        p.Ho = Hin / 2; p.Wo = Win / 2;
        p.Hin = p.Ho; p.Win = p.Wo;
        p.chunks0 = p.n_chunks = 4 * (L.C0 / 16);
        P.out_h = p.Ho; P.out_w = p.Wo;
        if (L.wpw && !out_nchw && pw_shape_ok(c.B, p.Ho, p.Wo, L.Cout, 4 * L.C0, 0)) {
            // the 1x1 GEMM kernel in space-to-depth mode (pw_mfma.hip)
            p.w = L.wpw;
            p.geo = pw_plan(c.B, p.Ho, p.Wo, L.Cout, 4 * L.C0, 0, true);
            P.kind = 5;
            P.in_kernel = p.geo.splits == 1 && (!want_norm || p.geo.fused_norm);
            return 0;
        }
        p.geo = conv_plan(c.B, p.Ho, p.Wo, L.Cout, 1, 1, 1, 4 * L.C0, 0, want_norm && !out_nchw, !out_nchw);
    } else {
        p.fold = 0; p.fold_w_stride = 0; p.up = L.up ? 1 : 0;
        p.KH = L.KH; p.KW = L.KW;
        p.Hin = Hin; p.Win = Win;
        // the kernel zero-fills every window pixel outside the image, so bottom / right padding is only an output size
        p.Ho = (Hin + 2 * L.pad + L.pad_hi - L.KH) / L.stride + 1;
        p.Wo = (Win + 2 * padw + L.pad_hi - L.KW) / L.stride + 1;
        P.out_h = p.Ho; P.out_w = p.Wo;
        p.geo = conv_plan(c.B, p.Ho, p.Wo, L.Cout, L.KH, L.KW, L.stride, L.C0, L.C1, want_norm && !out_nchw,
                          !out_nchw);
    }
    if (P.kind == 0 && L.wi7 && in_nchw && !out_nchw && !has_in1 && epi == 0 && L.pad_hi == 0 && padw == L.pad &&
        (size_t)c.B * p.Ho * p.Wo < (1u << 24)) {
        p.w = L.wi7;
        P.kind = 7;
        P.in_kernel = true;
        return 0;
    }
    if (L.wraw && out_nchw && !in_nchw && !has_in1 && epi == 0) {
        // a handful of output channels (final_conv): one pixel per thread instead of a 64-column MFMA tile
        P.kind = 4;
        P.in_kernel = true;
        return 0;
    }
    // 3x3 / stride 1 convolutions run as Winograd F(4x4,3x3) on power-of-two images, else as F(2x2,3x3), when the layer
    // has transformed weights
    const bool wino4 = P.kind == 0 && L.ww4 && !L.fold && !in_nchw && !out_nchw && padw == L.pad &&
                       wino4_shape_ok(c.B, p.Ho, p.Wo, L.Cout, L.C0, L.C1);
    const bool wino = P.kind == 0 && !wino4 && L.ww && !L.fold && !in_nchw && !out_nchw && padw == L.pad &&
                      wino_use(c.B, p.Ho, p.Wo, L.Cout, L.C0, L.C1);
    const bool pw = P.kind == 0 && L.wpw && !L.fold && !p.s2d && !in_nchw && !out_nchw && L.pad_hi == 0 && padw == 0 &&
                    pw_shape_ok(c.B, p.Ho, p.Wo, L.Cout, L.C0, L.C1);
    if (pw) {
        p.w = L.wpw;
        p.chunks0 = L.C0 / 16;
        p.n_chunks = (L.C0 + L.C1) / 16;
        p.geo = pw_plan(c.B, p.Ho, p.Wo, L.Cout, L.C0, L.C1, true);
        P.kind = 5;
    } else if (wino4) {
        p.w = L.ww4;
        p.chunks0 = L.C0 / 8;
        p.n_chunks = (L.C0 + L.C1) / 8;
        p.geo = wino4_plan(c.B, p.Ho, p.Wo, L.Cout, L.C0, L.C1, true);
        P.kind = 2;
    } else if (wino) {
        p.w = L.ww;
        p.chunks0 = L.C0 / 8;
        p.n_chunks = (L.C0 + L.C1) / 8;
        p.geo = wino_plan(c.B, p.Ho, p.Wo, L.Cout, L.C0, L.C1, true, want_norm);
        P.kind = 1;
    }
    P.in_kernel = p.geo.splits == 1 && (!want_norm || p.geo.fused_norm);
    return 0;
}

static int launch_planned(const PlannedConv& P, const ConvParams& q, hipStream_t s, dm_unet* u = nullptr) {
    if (u && u->train && ensure_packed(u, q.w, s)) return 1;  // training loop: weights re-packed on the device when stale
    switch (P.kind) {
        case 7: return init7_launch(q, s);
        case 5: return pw_launch(q, s);
        case 3: return upwino_launch(q, s);
        case 2: return wino4_launch(q, s);
        case 1: return wino_launch(q, s);
        default: return conv_launch(q, s);
    }
}

static int run_conv(Ctx& c, const ConvLayer& L, const float* in0, const float* in1, int Hin, int Win, float* out,
                    int epi, const float* g, const float* scale, const float* residual, bool in_nchw = false,
                    bool out_nchw = false, const ResParts* res_parts = nullptr) {
    // One convolution with its epilogue (epi = EPI_* requested by the caller; bias is implied by the layer).
    // The plan decides whether the epilogue runs inside the conv kernel (one workgroup owns all couts of a
    // pixel and K is not split) or in norm_act_kernel on the raw / K-split partial sums.  res_parts: the residual
    // is the sum of another convolution's partial tiles (only on the landing path; the caller asks plan_conv first).
    PlannedConv P;
    if (plan_conv(c, L, in1 != nullptr, Hin, Win, epi, in_nchw, out_nchw, P)) return 1;
    ConvParams& p = P.p;
    p.in0 = in0; p.in1 = in1;
    p.residual = residual; p.g = g; p.scale = scale;
    if (P.kind == 4) {
        if (c.dry()) return 0;
        if (c.u && c.u->train && ensure_packed(c.u, L.wraw, c.s)) return 1;
        return launch_pointwise_small(in0, L.wraw, L.bias, out, (int64_t)c.B * p.Ho * p.Wo, L.C0, L.Cout, p.Ho * p.Wo,
                                      c.s);
    }
    const int full_epi = epi | (L.bias ? EPI_BIAS : 0);
    if (P.in_kernel) {
        DM_REQUIRE(!res_parts, "residual partial sums need the landing pass");
        if (c.dry()) return 0;
        p.out = out; p.partial = 0; p.epi = full_epi;
        return launch_planned(P, p, c.s, c.u);
    }
    DM_REQUIRE(!out_nchw, "split / unfused epilogue writes NHWC");
    const size_t M = (size_t)c.B * P.out_h * P.out_w;
    float* part = c.A->alloc((size_t)p.geo.splits * M * L.Cout);
    if (c.dry()) {
        c.A->release(part);
        return 0;
    }
    p.out = part; p.partial = 1; p.epi = 0;
    if (launch_planned(P, p, c.s, c.u)) return 1;
    int rc;
    if (res_parts)
        rc = launch_norm_act(part, p.geo.splits, (int64_t)(M * L.Cout), L.bias, g, scale, c.ss_stride, P.out_h * P.out_w,
                             res_parts->part, out, (int64_t)M, L.Cout, full_epi | EPI_RESIDUAL, c.s, res_parts->nsplit,
                             res_parts->stride, res_parts->bias);
    else
        rc = launch_norm_act(part, p.geo.splits, (int64_t)(M * L.Cout), L.bias, g, scale, c.ss_stride, P.out_h * P.out_w,
                             residual, out, (int64_t)M, L.Cout, full_epi, c.s);
    c.A->release(part);
    return rc;
}

// The raw K-split partial sums of a convolution (no bias, no epilogue) for a consumer that lands them itself:
// *part = [nsplit][B*Ho*Wo][Cout], allocated here, released by the caller.
static int run_conv_partial(Ctx& c, const ConvLayer& L, const float* in0, const float* in1, int Hin, int Win, float** part,
                            int* nsplit) {
    PlannedConv P;
    if (plan_conv(c, L, in1 != nullptr, Hin, Win, 0, false, false, P)) return 1;
    DM_REQUIRE(P.kind != 4, "partial sums of a pointwise-small layer");
    ConvParams& p = P.p;
    p.in0 = in0; p.in1 = in1;
    const size_t M = (size_t)c.B * P.out_h * P.out_w;
    *part = c.A->alloc((size_t)p.geo.splits * M * L.Cout);
    *nsplit = p.geo.splits;
    if (c.dry()) return 0;
    p.out = *part; p.partial = 1; p.epi = 0;
    return launch_planned(P, p, c.s, c.u);
}

// Block.forward: conv3x3 -> RMSNorm -> (scale+1, shift) -> SiLU [-> + residual]
static int run_block(Ctx& c, const ConvLayer& L, const float* in0, const float* in1, int H, int W, const float* g,
                     const float* scale, const float* residual, float* out, const ResParts* res_parts = nullptr) {
    int flags = EPI_NORM | EPI_SILU | (scale ? EPI_SCALE_SHIFT : 0) | (residual ? EPI_RESIDUAL : 0);
    return run_conv(c, L, in0, in1, H, W, out, flags, g, scale, residual, false, false, res_parts);
}

// ResnetBlock.forward (DD/denoising_diffusion.py:136-148); x = cat(x0, x1)
static int run_resnet(Ctx& c, const ResBlock& R, const float* x0, const float* x1, int H, int W, float** out) {
    size_t n = (size_t)c.B * H * W * R.dout;
    const float* scale = c.ss ? c.ss + R.ss_off : nullptr;
    // block2(h1) + res_conv(x).  When both convolutions leave partial sums for a landing pass anyway (several cout
    // tiles under one RMSNorm, or K splits), one landing serves both: it finishes block2 and adds the res_conv partials.
    static const bool merge = std::getenv("DM_NO_RES_MERGE") == nullptr;
    PlannedConv P2, Pr;
    bool merged = false;
    if (R.has_res) {
        if (plan_conv(c, R.c2, false, H, W, EPI_NORM | EPI_SILU, false, false, P2)) return 1;
        if (plan_conv(c, R.res, x1 != nullptr, H, W, EPI_RESIDUAL, false, false, Pr)) return 1;
        merged = merge && !P2.in_kernel && !Pr.in_kernel && Pr.kind != 4;
    }
    ResParts rp{};
    float* rpart = nullptr;
    bool forked = false;
    if (merged) {
        // res_conv reads only x: its partial sums are produced next to block1 (second stream) when the step forks
        Ctx cs = c;
        if (c.par && !c.dry()) {
            if (fork_side(c, &cs.s)) return 1;
            forked = true;
        }
        if (run_conv_partial(cs, R.res, x0, x1, H, W, &rpart, &rp.nsplit)) return 1;
        rp.part = rpart;
        rp.stride = (int64_t)n;
        rp.bias = R.res.bias;
    }
    float* h1 = c.A->alloc(n);
    float* h2 = c.A->alloc(n);
    if (run_block(c, R.c1, x0, x1, H, W, R.g1, scale, nullptr, h1)) return 1;
    if (forked && join_side(c)) return 1;
    if (!R.has_res) {
        if (run_block(c, R.c2, h1, nullptr, H, W, R.g2, nullptr, x0, h2)) return 1;
        c.A->release(h1);
        *out = h2;
        return 0;
    }
    if (merged) {
        if (run_block(c, R.c2, h1, nullptr, H, W, R.g2, nullptr, nullptr, h2, &rp)) return 1;
        c.A->release(rpart);
        c.A->release(h1);
        *out = h2;
        return 0;
    }
    if (run_block(c, R.c2, h1, nullptr, H, W, R.g2, nullptr, nullptr, h2)) return 1;
    c.A->release(h1);
    float* o = c.A->alloc(n);
    if (run_conv(c, R.res, x0, x1, H, W, o, EPI_RESIDUAL, nullptr, nullptr, h2)) return 1;
    c.A->release(h2);
    *out = o;
    return 0;
}

// attn(x) + x  (LinearAttention DD/denoising_diffusion.py:173-193, Attention :215-229)
static int run_attn(Ctx& c, const AttnLayer& At, const float* x, int H, int W, float** out, bool add_x = true) {
    const int res_flag = add_x ? EPI_RESIDUAL : 0;
    const float* xres = add_x ? x : nullptr;
    dm_unet* u = c.u;
    const int n = H * W, heads = At.heads, hidden = heads * u->dh;
    const size_t rows = (size_t)c.B * n;
    if (!At.full && At.has_fused) {
        float* ws = c.A->alloc(linattn_fused_ws_floats(c.B, n));
        float* yf = c.A->alloc(rows * At.dim);
        if (!c.dry() && launch_linattn_fused(At.fused, x, ws, yf, c.B, n, add_x, c.s)) return 1;
        c.A->release(ws);
        *out = yf;
        return 0;
    }
    if (At.full && At.has16 && n == 16) {
        float* yf = c.A->alloc(rows * At.dim);
        if (!c.dry() && launch_attn16_fused(At.a16, x, yf, c.B, add_x, c.s)) return 1;
        *out = yf;
        return 0;
    }
    float* xn = c.A->alloc(rows * At.dim);
    float* qkv = c.A->alloc(rows * 3 * hidden);
    float* o = c.A->alloc(rows * hidden);
    float* y = c.A->alloc(rows * At.dim);
    if (!c.dry() && launch_norm_act(x, 1, 0, nullptr, At.norm_g, nullptr, 0, n, nullptr, xn, (int64_t)rows, At.dim,
                                    EPI_NORM, c.s))
        return 1;
    if (run_conv(c, At.qkv, xn, nullptr, H, W, qkv, 0, nullptr, nullptr, nullptr)) return 1;
    if (At.full) {
        if (!c.dry()) {
            const float* mk = At.mem_kv;
            const float* mv = At.mem_kv + (size_t)heads * 4 * u->dh;
            if (launch_attention_core(qkv, 3 * hidden, qkv + hidden, qkv + 2 * hidden, 3 * hidden, mk, mv, 4, o,
                                      hidden, c.B, n, n, heads, u->dh, 1.0f / sqrtf((float)u->dh), c.s))
                return 1;
        }
        if (run_conv(c, At.out, o, nullptr, H, W, y, res_flag, nullptr, nullptr, xres)) return 1;
    } else {
        float* ctxws = c.A->alloc((size_t)c.B * heads * u->dh * u->dh);
        if (!c.dry() && launch_linear_attention_core(qkv, At.mem_kv, ctxws, o, c.B, n, heads, u->dh, c.s)) return 1;
        if (run_conv(c, At.out, o, nullptr, H, W, y, EPI_NORM | res_flag, At.out_g, nullptr, xres)) return 1;
        c.A->release(ctxws);
    }
    c.A->release(xn);
    c.A->release(qkv);
    c.A->release(o);
    *out = y;
    return 0;
}

// CrossAttention.forward (DD/denoising_diffusion_text_conditional.py:54-78); the result REPLACES x (:173-177)
static int run_cross(Ctx& c, const CrossLayer& Cr, const float* x, int H, int W, const float* ctx, int m,
                     float** out) {
    dm_unet* u = c.u;
    const int n = H * W, inner = 4 * u->dh, dim = Cr.out.Cout, E = u->cfg.text_emb_dim;
    const size_t rows = (size_t)c.B * n;
    static const bool one_token_path = std::getenv("DM_NO_CROSS1") == nullptr;
    if (m == 1 && one_token_path) {
        // One context token (what every sampler of the reference passes: (B, 512) -> (B, 1, 512), :57-58): the softmax
        // over a single key is exactly 1.0, so every query's output is v and the q / k projections drop out.  The layer
        // is z_b = RMSNorm1D(to_out(to_v(ctx_b))) spread over the pixels of image b -- three row-per-image launches and
        // one broadcast instead of two 1x1 convolutions over every pixel and an attention core.
        float* v = c.A->alloc((size_t)c.B * inner);
        float* yb = c.A->alloc((size_t)c.B * dim);
        float* zb = c.A->alloc((size_t)c.B * dim);
        float* y = c.A->alloc(rows * dim);
        if (!c.dry()) {
            if (launch_linear_rows(ctx, E, Cr.wv, nullptr, v, inner, c.B, E, inner, 0, 0, c.s)) return 1;
            if (launch_linear_rows(v, inner, Cr.wo_raw, Cr.out.bias, yb, dim, c.B, inner, dim, 0, 0, c.s)) return 1;
            if (launch_norm_act(yb, 1, 0, nullptr, Cr.g, nullptr, 0, 1, nullptr, zb, (int64_t)c.B, dim, EPI_NORM, c.s))
                return 1;
            if (launch_broadcast_rows(zb, y, (int)rows, dim, dim, c.s, n)) return 1;
        }
        c.A->release(v);
        c.A->release(yb);
        c.A->release(zb);
        *out = y;
        return 0;
    }
    float* q = c.A->alloc(rows * inner);
    float* k = c.A->alloc((size_t)c.B * m * inner);
    float* v = c.A->alloc((size_t)c.B * m * inner);
    float* o = c.A->alloc(rows * inner);
    float* y = c.A->alloc(rows * dim);
    if (run_conv(c, Cr.q, x, nullptr, H, W, q, 0, nullptr, nullptr, nullptr)) return 1;
    if (!c.dry()) {
        if (launch_linear_rows(ctx, E, Cr.wk, nullptr, k, inner, c.B * m, E, inner, 0, 0, c.s)) return 1;
        if (launch_linear_rows(ctx, E, Cr.wv, nullptr, v, inner, c.B * m, E, inner, 0, 0, c.s)) return 1;
        if (launch_attention_core(q, inner, k, v, inner, nullptr, nullptr, 0, o, inner, c.B, n, m, 4, u->dh,
                                  1.0f / sqrtf((float)u->dh), c.s))
            return 1;
    }
    if (run_conv(c, Cr.out, o, nullptr, H, W, y, EPI_NORM, Cr.g, nullptr, nullptr)) return 1;
    c.A->release(q);
    c.A->release(k);
    c.A->release(v);
    c.A->release(o);
    *out = y;
    return 0;
}

// Unet.forward (DD/denoising_diffusion.py:349-390; text hooks DD/denoising_diffusion_text_conditional.py:131-214)
static int unet_forward_impl(dm_unet* u, Arena& A, const float* x_nchw, const int64_t* t_dev,
                             const int64_t* step_times, const SamplerState* step_dev, const float* ctx, int ctx_tokens,
                             float* out_nchw, int B, int H, int W, hipStream_t s) {
    const dm_unet_cfg& cfg = u->cfg;
    Ctx c{u, &A, s, B, nullptr, 0};
    const int td = u->time_dim;
    const bool text_concat = cfg.text_mode == DM_TEXT_CONCAT && ctx != nullptr;
    const bool text_cross = cfg.text_mode == DM_TEXT_CROSS && ctx != nullptr;
    // one context token: the three CrossAttention layers ignore their image input (run_cross), which makes everything
    // between the last skip connection and cross_attn_up dead code (DM_NO_CROSS1 computes it anyway)
    static const bool cross1 = std::getenv("DM_NO_CROSS1") == nullptr;
    const bool dead_bottleneck = text_cross && ctx_tokens == 1 && cross1;
    // the time embedding is one row when the whole batch shares t (samplers), else one row per sample
    const int Bt = (step_times && !text_concat) ? 1 : B;
    const int Rt = step_times ? 1 : B;  // rows of the sinusoid / time_mlp
    const int lsd = cfg.learned_sinusoidal_dim;  // > 0: cat(t, sin(t w 2 pi), cos(t w 2 pi)) of width lsd + 1 (:96-101)
    const int fdim = lsd > 0 ? lsd + 1 : cfg.dim;
    float* e0 = A.alloc((size_t)Rt * fdim);
    float* e1 = A.alloc((size_t)Rt * td);
    float* temb = A.alloc((size_t)B * td);
    float* ss = A.alloc((size_t)Bt * u->ss_total);
    // The time embedding (sinusoid, time_mlp, every ResnetBlock.mlp: 4+ dependent launches) does not depend on x: when the
    // step forks it runs on the second stream next to init_conv.  Nothing it allocates is released before the join.
    c.par = !A.dry && par_policy(B, H, W);
    hipStream_t ts = s;
    if (c.par && fork_side(c, &ts)) return 1;
    if (!A.dry) {
        if (launch_sinusoid(t_dev, step_times, step_dev, u->freqs, e0, Rt, (lsd > 0 ? lsd : cfg.dim) / 2, ts, lsd > 0)) return 1;
        if (launch_linear_rows(e0, fdim, u->tw1, u->tb1, e1, td, Rt, fdim, td, 0, 2, ts)) return 1;
        if (launch_linear_rows(e1, td, u->tw2, u->tb2, temb, td, Rt, td, td, 0, 0, ts)) return 1;
    }
    const float* tfinal = temb;
    float *cat = nullptr, *tf0 = nullptr;
    if (text_concat) {
        // t = text_concat_proj(cat(t, text_proj(text_emb)))  (:146-152); ctx is (B, 1, E) or (B, E)
        DM_REQUIRE(ctx_tokens == 1, "text concat conditioning takes one pooled embedding per sample");
        cat = A.alloc((size_t)B * 2 * td);
        tf0 = A.alloc((size_t)B * td);
        float* t2 = A.alloc((size_t)B * td);
        if (!A.dry) {
            // left half: the time embedding of every row (one broadcast launch when the batch shares t)
            if (Rt == 1) {
                if (launch_broadcast_rows(temb, cat, B, td, 2 * td, ts)) return 1;
            } else {
                DM_CHECK_HIP(hipMemcpy2DAsync(cat, 2 * td * sizeof(float), temb, td * sizeof(float),
                                              td * sizeof(float), B, hipMemcpyDeviceToDevice, ts));
            }
            if (launch_linear_rows(ctx, cfg.text_emb_dim, u->tp_w0, u->tp_b0, tf0, td, B, cfg.text_emb_dim, td, 0, 2, ts))
                return 1;
            if (launch_linear_rows(tf0, td, u->tp_w2, u->tp_b2, cat + td, 2 * td, B, td, td, 0, 0, ts)) return 1;
            if (launch_linear_rows(cat, 2 * td, u->tc_w, u->tc_b, t2, td, B, 2 * td, td, 0, 0, ts)) return 1;
        }
        tfinal = t2;
    }
    if (!A.dry) {
        // every ResnetBlock.mlp (SiLU -> Linear) in one launch
        if (launch_linear_rows(tfinal, td, u->ss_w, u->ss_b, ss, u->ss_total, Bt, td, u->ss_total, 1, 0, ts)) return 1;
    }
    c.ss = A.dry ? reinterpret_cast<const float*>(16) : ss;  // non-null marker in dry mode
    c.ss_stride = Bt == 1 ? 0 : u->ss_total;

    const int n_st = cfg.n_stages;
    float* x = A.alloc((size_t)B * H * W * u->init_dim);
    if (run_conv(c, u->init_conv, x_nchw, nullptr, H, W, x, 0, nullptr, nullptr, nullptr, /*in_nchw=*/true)) return 1;
    if (c.par && join_side(c)) return 1;
    if (cat) A.release(cat);
    if (tf0) A.release(tf0);
    if (tfinal != temb) A.release(tfinal);
    A.release(e0);
    A.release(e1);
    A.release(temb);
    const float* r = x;  // Unet.forward's `r = x.clone()`: read again by final_res_block, never released before
    auto rel = [&](const float* t) {
        if (t != r) A.release(t);
    };
    std::vector<const float*> skips;
    int h = H, w = W;
    float* cur = x;
    for (int i = 0; i < n_st; ++i) {
        Stage& S = u->downs[i];
        float *a, *b2, *at;
        if (run_resnet(c, S.b1, cur, nullptr, h, w, &a)) return 1;
        rel(cur);
        skips.push_back(a);
        if (run_resnet(c, S.b2, a, nullptr, h, w, &b2)) return 1;
        if (run_attn(c, S.attn, b2, h, w, &at)) return 1;
        rel(b2);
        skips.push_back(at);
        int ho = (i < n_st - 1) ? h / 2 : h, wo = (i < n_st - 1) ? w / 2 : w;
        if (i == n_st - 1 && dead_bottleneck) {  // its only consumer is a CrossAttention that ignores x (see below)
            cur = nullptr;
            break;
        }
        float* d = A.alloc((size_t)B * ho * wo * S.resample.Cout);
        if (run_conv(c, S.resample, at, nullptr, h, w, d, 0, nullptr, nullptr, nullptr)) return 1;
        cur = d; h = ho; w = wo;
    }
    float* t0;
    if (dead_bottleneck) {
        // CrossAttention REPLACES x (:173-177, :187-191, :199-203), and with one context token its output does not depend
        // on x at all (run_cross): cross_attn_down discards the last down conv, cross_attn discards mid_block1,
        // cross_attn_up discards mid_attn and mid_block2.  None of that work can reach the output, so the up path starts
        // from cross_attn_up(ctx); the skip connections carry the image signal, exactly as in the reference.
        if (run_cross(c, u->cross_up, nullptr, h, w, ctx, ctx_tokens, &t0)) return 1;
        cur = t0;
    } else {
        if (text_cross) { if (run_cross(c, u->cross_down, cur, h, w, ctx, ctx_tokens, &t0)) return 1; rel(cur); cur = t0; }
        if (run_resnet(c, u->mid1, cur, nullptr, h, w, &t0)) return 1; rel(cur); cur = t0;
        if (text_cross) { if (run_cross(c, u->cross_mid, cur, h, w, ctx, ctx_tokens, &t0)) return 1; rel(cur); cur = t0; }
        if (run_attn(c, u->mid_attn, cur, h, w, &t0)) return 1; rel(cur); cur = t0;
        if (run_resnet(c, u->mid2, cur, nullptr, h, w, &t0)) return 1; rel(cur); cur = t0;
        if (text_cross) { if (run_cross(c, u->cross_up, cur, h, w, ctx, ctx_tokens, &t0)) return 1; rel(cur); cur = t0; }
    }
    for (int j = 0; j < n_st; ++j) {
        Stage& S = u->ups[j];
        float *a, *b2, *at;
        const float* sk = skips.back(); skips.pop_back();
        if (run_resnet(c, S.b1, cur, sk, h, w, &a)) return 1;
        rel(cur);
        rel(sk);
        sk = skips.back(); skips.pop_back();
        if (run_resnet(c, S.b2, a, sk, h, w, &b2)) return 1;
        rel(a);
        rel(sk);
        if (run_attn(c, S.attn, b2, h, w, &at)) return 1;
        rel(b2);
        bool last = j == n_st - 1;
        int ho = last ? h : h * 2, wo = last ? w : w * 2;
        float* d = A.alloc((size_t)B * ho * wo * S.resample.Cout);
        // the conv sees the (virtually) upsampled tensor
        if (run_conv(c, S.resample, at, nullptr, ho, wo, d, 0, nullptr, nullptr, nullptr)) return 1;
        rel(at);
        cur = d; h = ho; w = wo;
    }
    float* fr;
    if (run_resnet(c, u->final_res, cur, r, h, w, &fr)) return 1;
    rel(cur);
    A.release(r);
    if (run_conv(c, u->final_conv, fr, nullptr, h, w, out_nchw, 0, nullptr, nullptr, nullptr, false, /*out_nchw=*/true))
        return 1;
    A.release(fr);
    A.release(ss);
    return 0;
}

static int ensure_workspace(dm_unet* u, size_t bytes) {
    if (bytes <= u->ws_cap) return 0;
    DM_CHECK_HIP(hipDeviceSynchronize());
    u->drop_graph();  // the captured step points into the old arena
    if (u->ws) (void)hipFree(u->ws);
    u->ws = nullptr;
    u->ws_cap = 0;
    void* p = nullptr;
    DM_CHECK_HIP(hipMalloc(&p, bytes));
    u->ws = static_cast<char*>(p);
    u->ws_cap = bytes;
    return 0;
}

static int check_hw(dm_unet* u, int H, int W) {
    int f = 1 << (u->cfg.n_stages - 1);
    if (H % f || W % f) {
        set_error("your input dimensions (" + std::to_string(H) + ", " + std::to_string(W) +
                  ") need to be divisible by " + std::to_string(f) + ", given the unet");
        return 1;
    }
    return 0;
}

}  // namespace dm

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* dm_last_error(void) { return g_err.c_str(); }
int dm_abi_version(void) { return 5; }

int dm_unet_create(const dm_unet_cfg* cfg, int device, dm_unet** out) {
    DM_REQUIRE(cfg && out, "null argument");
    DM_REQUIRE(cfg->n_stages >= 1 && cfg->n_stages <= DM_MAX_STAGES, "n_stages out of range");
    DM_REQUIRE(cfg->dim > 0 && cfg->dim % 2 == 0, "dim must be positive and even");
    DM_REQUIRE(cfg->attn_dim_head == 32, "HIP attention kernels are specialised for attn_dim_head == 32");
    DM_REQUIRE(cfg->attn_heads >= 1 && cfg->attn_heads <= 16, "attn_heads out of range");
    for (int i = 0; i < cfg->n_stages; ++i)
        DM_REQUIRE(cfg->attn_heads_stage[i] >= 0 && cfg->attn_heads_stage[i] <= 16, "attn_heads of a stage out of range");
    int ndev = 0;
    DM_CHECK_HIP(hipGetDeviceCount(&ndev));
    DM_REQUIRE(device >= 0 && device < ndev, "no such HIP device");
    DM_CHECK_HIP(hipSetDevice(device));
    auto u = std::make_unique<dm_unet>();
    u->cfg = *cfg;
    u->device = device;
    u->init_dim = cfg->init_dim ? cfg->init_dim : cfg->dim;
    u->out_dim = cfg->out_dim ? cfg->out_dim : cfg->channels;
    u->time_dim = cfg->dim * 4;
    u->heads = cfg->attn_heads;
    u->dh = cfg->attn_dim_head;
    u->dims.push_back(u->init_dim);
    for (int i = 0; i < cfg->n_stages; ++i) u->dims.push_back(cfg->dim * cfg->dim_mults[i]);
    dm_unet* p = u.get();
    const int n = cfg->n_stages, td = u->time_dim;
    expect(p, "init_conv.weight", {u->init_dim, cfg->input_channels, 7, 7});
    expect(p, "init_conv.bias", {u->init_dim});
    if (cfg->learned_sinusoidal_dim > 0) expect(p, "time_mlp.0.weights", {cfg->learned_sinusoidal_dim / 2});
    expect(p, "time_mlp.1.weight", {td, cfg->learned_sinusoidal_dim > 0 ? cfg->learned_sinusoidal_dim + 1 : cfg->dim});
    expect(p, "time_mlp.1.bias", {td});
    expect(p, "time_mlp.3.weight", {td, td});
    expect(p, "time_mlp.3.bias", {td});
    for (int i = 0; i < n; ++i) {
        int din = u->dims[i], dout = u->dims[i + 1];
        std::string q = idx("downs", i);
        expect_resnet(p, q + ".0", din, din);
        expect_resnet(p, q + ".1", din, din);
        expect_attn(p, q + ".2", din, cfg->full_attn[i] != 0, stage_heads(*cfg, i));
        if (i < n - 1) {
            expect(p, q + ".3.1.weight", {dout, din * 4, 1, 1});
            expect(p, q + ".3.1.bias", {dout});
        } else {
            expect(p, q + ".3.weight", {dout, din, 3, 3});
            expect(p, q + ".3.bias", {dout});
        }
    }
    for (int j = 0; j < n; ++j) {
        int din = u->dims[n - 1 - j], dout = u->dims[n - j];
        std::string q = idx("ups", j);
        expect_resnet(p, q + ".0", dout + din, dout);
        expect_resnet(p, q + ".1", dout + din, dout);
        expect_attn(p, q + ".2", dout, cfg->full_attn[n - 1 - j] != 0, stage_heads(*cfg, n - 1 - j));
        if (j < n - 1) {
            expect(p, q + ".3.1.weight", {din, dout, 3, 3});
            expect(p, q + ".3.1.bias", {din});
        } else {
            expect(p, q + ".3.weight", {din, dout, 3, 3});
            expect(p, q + ".3.bias", {din});
        }
    }
    int mid = u->dims.back();
    expect_resnet(p, "mid_block1", mid, mid);
    expect_attn(p, "mid_attn", mid, true, stage_heads(*cfg, n - 1));
    expect_resnet(p, "mid_block2", mid, mid);
    expect_resnet(p, "final_res_block", 2 * u->init_dim, u->init_dim);
    expect(p, "final_conv.weight", {u->out_dim, u->init_dim, 1, 1});
    expect(p, "final_conv.bias", {u->out_dim});
    if (cfg->text_mode == DM_TEXT_CONCAT) {
        expect(p, "text_proj.0.weight", {td, cfg->text_emb_dim});
        expect(p, "text_proj.0.bias", {td});
        expect(p, "text_proj.2.weight", {td, td});
        expect(p, "text_proj.2.bias", {td});
        expect(p, "text_concat_proj.weight", {td, 2 * td});
        expect(p, "text_concat_proj.bias", {td});
    } else if (cfg->text_mode == DM_TEXT_CROSS) {
        expect_cross(p, "cross_attn", mid);
        expect_cross(p, "cross_attn_down", mid);
        expect_cross(p, "cross_attn_up", mid);
    }
    *out = u.release();
    return 0;
}

void dm_unet_destroy(dm_unet* u) {
    if (!u) return;
    (void)hipSetDevice(u->device);
    if (u->train) free_train(u);
    u->drop_graph();
    if (u->cap_stream) (void)hipStreamDestroy(u->cap_stream);
    if (u->side) (void)hipStreamDestroy(u->side);
    for (hipEvent_t e : u->ev_pool) (void)hipEventDestroy(e);
    if (u->done_ev) (void)hipEventDestroy(u->done_ev);
    if (u->ws) (void)hipFree(u->ws);
    if (u->state_dev) (void)hipFree(u->state_dev);
    if (u->times_dev) (void)hipFree(u->times_dev);
    if (u->coefs_dev) (void)hipFree(u->coefs_dev);
    delete u;
}

int dm_unet_set_param(dm_unet* u, const char* name, const float* data_host, const int64_t* shape, int ndim) {
    DM_REQUIRE(u && name && data_host && shape, "null argument");
    DM_REQUIRE(!u->finalized, "set_param after finalize");
    auto it = u->params.find(name);
    if (it == u->params.end()) {
        set_error(std::string("unexpected parameter: ") + name);
        return 1;
    }
    HostTensor& t = it->second;
    bool ok = (int)t.shape.size() == ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = t.shape[i] == shape[i];
    if (!ok) {
        set_error(std::string("shape mismatch for ") + name);
        return 1;
    }
    t.data.assign(data_host, data_host + t.numel());
    t.set = true;
    return 0;
}

int dm_unet_missing_params(dm_unet* u) {
    if (!u) return -1;
    int missing = 0;
    std::string first;
    for (auto& kv : u->params)
        if (!kv.second.set) {
            if (!missing) first = kv.first;
            ++missing;
        }
    if (missing) set_error("missing parameter: " + first);
    return missing;
}

/* read back the handle's host copy of one parameter (what dm_unet_set_param / dm_unet_update_param stored; after
 * dm_unet_train_sync: the trained values) */
int dm_unet_get_param_host(dm_unet* u, const char* name, float* out_host, int64_t n) {
    DM_REQUIRE(u && name && out_host, "null argument");
    auto it = u->params.find(name);
    if (it == u->params.end() || !it->second.set) {
        set_error(std::string("unknown or unset parameter: ") + name);
        return 1;
    }
    DM_REQUIRE((int64_t)it->second.numel() == n, "dm_unet_get_param_host: element count mismatch");
    std::memcpy(out_host, it->second.data.data(), (size_t)n * sizeof(float));
    return 0;
}

int dm_unet_finalize(dm_unet* u) {
    DM_REQUIRE(u, "null handle");
    DM_REQUIRE(!u->finalized, "already finalized");
    if (dm_unet_missing_params(u) != 0) return 1;
    DM_CHECK_HIP(hipSetDevice(u->device));
    u->own.rec = &u->pack_ops;
    if (build_all(u)) return 1;
    // the host copies stay: dm_unet_update_param / dm_unet_refresh re-pack from them
    u->finalized = true;
    return 0;
}

int dm_unet_update_param(dm_unet* u, const char* name, const float* data_host, const int64_t* shape, int ndim) {
    DM_REQUIRE(u && name && data_host && shape, "null argument");
    DM_REQUIRE(u->finalized, "dm_unet_update_param before dm_unet_finalize (use dm_unet_set_param)");
    // the device-resident parameters have moved on (dm_unet_optimizer_step): bring the host copies up to date first, so
    // that "unchanged" below compares with the current values and a partial update does not resurrect stale ones
    if (u->infer_stale && dm_unet_train_sync(u)) return 1;
    auto it = u->params.find(name);
    if (it == u->params.end()) {
        set_error(std::string("unexpected parameter: ") + name);
        return 1;
    }
    HostTensor& t = it->second;
    bool ok = (int)t.shape.size() == ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = t.shape[i] == shape[i];
    if (!ok) {
        set_error(std::string("shape mismatch for ") + name);
        return 1;
    }
    if (std::memcmp(t.data.data(), data_host, t.numel() * sizeof(float)) == 0) return 0;  // unchanged
    t.data.assign(data_host, data_host + t.numel());
    t.dirty = true;
    return 0;
}

int dm_unet_refresh(dm_unet* u) {
    DM_REQUIRE(u, "null handle");
    DM_REQUIRE(u->finalized, "dm_unet_refresh before dm_unet_finalize");
    bool any = u->poisoned;
    for (auto& kv : u->params) any = any || kv.second.dirty;
    if (!any) return 0;
    DM_CHECK_HIP(hipSetDevice(u->device));
    DM_CHECK_HIP(hipDeviceSynchronize());  // no kernel may be reading the buffers that are rewritten
    u->own.refreshing = true;
    u->own.cursor = 0;
    int rc = build_all(u);
    if (!rc && u->own.cursor != u->own.ptrs.size()) {
        set_error("refresh: not every weight buffer was visited");
        rc = 1;
    }
    u->own.refreshing = false;
    if (rc) {
        // Some buffers hold new values, some old ones.  The dirty flags stay set so that a retry re-packs everything that
        // changed, the cached step graph is dropped, and the handle refuses to run until a refresh has succeeded.
        u->drop_graph();
        u->poisoned = true;
        return rc;
    }
    // the input-gradient convolutions are packed from the same parameters; a device-resident master copy follows the host
    if (u->train && (build_train(u) || upload_master_if_resident(u))) {
        u->poisoned = true;
        return 1;
    }
    u->poisoned = false;
    for (auto& kv : u->params) kv.second.dirty = false;
    return 0;
}

int dm_unet_graph_captures(dm_unet* u) { return u ? u->graph_captures : -1; }
int64_t dm_unet_workspace_bytes(dm_unet* u) { return u ? (int64_t)u->ws_cap : -1; }

int dm_unet_forward(dm_unet* u, const float* x, const int64_t* time, const float* ctx, int ctx_tokens, float* out,
                    int B, int H, int W, void* stream) {
    DM_REQUIRE(u && x && time && out, "null argument");
    DM_REQUIRE(u->finalized, "dm_unet_finalize has not been called");
    DM_REQUIRE(!u->poisoned, "the last dm_unet_refresh failed: refresh again before running the model");
    DM_REQUIRE(!u->infer_stale, "parameters were updated on the device (dm_unet_optimizer_step): call dm_unet_train_sync "
                                "before sampling from this handle");
    DM_REQUIRE(B > 0, "empty batch");
    if (check_hw(u, H, W)) return 1;
    DM_CHECK_HIP(hipSetDevice(u->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    Arena dry;
    dry.dry = true;
    if (unet_forward_impl(u, dry, x, time, nullptr, nullptr, ctx, ctx_tokens, out, B, H, W, s)) return 1;
    if (ensure_workspace(u, dry.off)) return 1;
    Arena A;
    A.base = u->ws;
    A.cap = u->ws_cap;
    if (u->order_after_previous(s)) return 1;
    if (unet_forward_impl(u, A, x, time, nullptr, nullptr, ctx, ctx_tokens, out, B, H, W, s)) return 1;
    return u->mark_done(s);
}

}  // extern "C"

// The sampling loop behind dm_sample / dm_sample_cond.  cond (B, cond_channels, H, W) is the image condition of
// DD/denoising_diffusion_image_conditional.py:51-55,156-180: constant over the loop, concatenated behind x in front of
// init_conv at every step.
//
// One denoise step (U-Net forward + update + step counter) touches only handle-owned memory: x, eps, the [x | cond]
// input, the text context and the final image live at fixed offsets of the workspace, and everything that differs
// between two calls of one shape (seed, Philox offset, step tables) is device DATA, not a kernel argument.  The step is
// therefore captured into a hipGraph once per (shape, sampler kind) and the instantiated graph is replayed by every
// later call; it is re-captured only when the shape, an injected-noise / all-steps pointer or the workspace changes.
static int sample_impl(dm_unet* u, int kind, int n_steps, const int64_t* times_host, const float* coefs_host,
                       const float* x_T, const float* noise, uint64_t seed, uint64_t sample_offset, const float* ctx,
                       int ctx_tokens, const float* cond, int cond_channels, float* out, float* all_steps, int B, int H,
                       int W, int unnormalize, int use_graph, void* stream, int objective = DM_OBJ_PRED_NOISE,
                       int self_cond = 0) {
    DM_REQUIRE(u && times_host && coefs_host && x_T && out, "null argument");
    DM_REQUIRE(u->finalized, "dm_unet_finalize has not been called");
    DM_REQUIRE(!u->poisoned, "the last dm_unet_refresh failed: refresh again before running the model");
    DM_REQUIRE(!u->infer_stale, "parameters were updated on the device (dm_unet_optimizer_step): call dm_unet_train_sync "
                                "before sampling from this handle");
    DM_REQUIRE(kind == DM_SAMPLER_DDPM || kind == DM_SAMPLER_DDIM, "unknown sampler kind");
    DM_REQUIRE(n_steps > 0 && B > 0, "empty run");
    DM_REQUIRE(u->out_dim == u->cfg.channels, "sampler needs out_dim == channels (DD/denoising_diffusion.py:456)");
    DM_REQUIRE((cond == nullptr) == (cond_channels == 0) && cond_channels >= 0, "cond and cond_channels come together");
    DM_REQUIRE(objective >= DM_OBJ_PRED_NOISE && objective <= DM_OBJ_PRED_V, "unknown objective");
    DM_REQUIRE(!self_cond || cond_channels == 0, "self-conditioning and an image condition are not combined");
    DM_REQUIRE(u->cfg.input_channels == u->cfg.channels * (self_cond ? 2 : 1) + cond_channels,
               "U-Net input channels != channels [* 2 with self-conditioning] + cond_channels");
    DM_REQUIRE((ctx == nullptr) == (ctx_tokens == 0), "ctx and ctx_tokens come together");
    if (check_hw(u, H, W)) return 1;
    DM_CHECK_HIP(hipSetDevice(u->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = u->cfg.channels;
    const int64_t n = (int64_t)B * C * H * W;
    // the U-Net input when it is wider than x: [x | cond] (image condition) or [x_start | x] (self-conditioning)
    const int Cin = self_cond ? 2 * C : C + cond_channels;
    const bool wide = Cin != C;
    const int64_t n_in = (int64_t)B * Cin * H * W;
    const int64_t n_ctx = ctx ? (int64_t)B * ctx_tokens * u->cfg.text_emb_dim : 0;
    const uint64_t elem_off = sample_offset * (uint64_t)C * H * W;  // global element index of this shard's first value
    DM_REQUIRE(elem_off % 4 == 0, "sample_offset * C * H * W must be a multiple of 4");

    if (!u->state_dev) DM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&u->state_dev), 256));
    if (n_steps > u->sampler_cap) {
        DM_CHECK_HIP(hipDeviceSynchronize());
        u->drop_graph();
        if (u->times_dev) (void)hipFree(u->times_dev);
        if (u->coefs_dev) (void)hipFree(u->coefs_dev);
        u->times_dev = nullptr;
        u->coefs_dev = nullptr;
        u->sampler_cap = 0;
        DM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&u->times_dev), n_steps * sizeof(int64_t)));
        DM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&u->coefs_dev), (size_t)n_steps * DM_COEFS * sizeof(float)));
        u->sampler_cap = n_steps;
    }
    // graph mode on the legacy default stream: that stream cannot be captured, so the whole call runs on a stream of
    // the handle, ordered after the caller's work by a synchronisation here and finished before return
    const bool own_stream = use_graph && s == nullptr;
    if (own_stream) {
        if (!u->cap_stream) DM_CHECK_HIP(hipStreamCreateWithFlags(&u->cap_stream, hipStreamNonBlocking));
        DM_CHECK_HIP(hipStreamSynchronize(nullptr));
        s = u->cap_stream;
    }
    // workspace: [x | eps | [x | cond] | ctx | forward arena]
    Arena dry;
    dry.dry = true;
    dry.alloc(n);
    dry.alloc(n);
    if (wide) dry.alloc(n_in);
    if (self_cond) dry.alloc(n);
    if (ctx) dry.alloc(n_ctx);
    const float* ctx_marker = ctx ? reinterpret_cast<const float*>(16) : nullptr;
    if (unet_forward_impl(u, dry, nullptr, nullptr, u->times_dev, u->state_dev, ctx_marker, ctx_tokens, nullptr, B, H, W, s))
        return 1;
    if (ensure_workspace(u, dry.off)) return 1;

    if (u->order_after_previous(s)) return 1;
    SamplerState st_host{};
    st_host.step = 0;
    st_host.n_steps = n_steps;
    st_host.unnormalize = unnormalize;
    st_host.seed = seed;
    st_host.off4 = elem_off / 4;
    DM_CHECK_HIP(hipMemcpyAsync(u->times_dev, times_host, n_steps * sizeof(int64_t), hipMemcpyHostToDevice, s));
    DM_CHECK_HIP(hipMemcpyAsync(u->coefs_dev, coefs_host, (size_t)n_steps * DM_COEFS * sizeof(float),
                                hipMemcpyHostToDevice, s));
    DM_CHECK_HIP(hipMemcpyAsync(u->state_dev, &st_host, sizeof(st_host), hipMemcpyHostToDevice, s));
    DM_CHECK_HIP(hipStreamSynchronize(s));  // the host tables and st_host may go away when this function returns

    Arena A;
    A.base = u->ws;
    A.cap = u->ws_cap;
    float* xbuf = A.alloc(n);
    float* eps = A.alloc(n);
    float* xin = wide ? A.alloc(n_in) : nullptr;  // [x | cond] or [x_start | x] per image, what init_conv reads
    float* xstart = self_cond ? A.alloc(n) : nullptr;  // clamped x_0 estimate of the previous step
    float* ctxbuf = ctx ? A.alloc(n_ctx) : nullptr;
    const std::vector<Arena::Blk> arena_mark = A.blks;  // allocator state in front of a denoise step
    DM_CHECK_HIP(hipMemcpyAsync(xbuf, x_T, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (ctx) DM_CHECK_HIP(hipMemcpyAsync(ctxbuf, ctx, n_ctx * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (cond && launch_copy_channels(cond, xin, B, cond_channels, C + cond_channels, C, H * W, s)) return 1;
    if (self_cond) DM_CHECK_HIP(hipMemsetAsync(xstart, 0, n * sizeof(float), s));  // x_self_cond = zeros_like(x) (:353)
    if (all_steps) DM_CHECK_HIP(hipMemcpyAsync(all_steps, x_T, n * sizeof(float), hipMemcpyDeviceToDevice, s));

    auto one_step = [&](hipStream_t st) -> int {
        A.blks = arena_mark;
        if (cond && launch_copy_channels(xbuf, xin, B, C, Cin, 0, H * W, st)) return 1;
        if (self_cond && (launch_copy_channels(xstart, xin, B, C, Cin, 0, H * W, st) ||
                          launch_copy_channels(xbuf, xin, B, C, Cin, C, H * W, st)))
            return 1;
        if (unet_forward_impl(u, A, wide ? xin : xbuf, nullptr, u->times_dev, u->state_dev, ctxbuf, ctx_tokens, eps, B, H,
                              W, st))
            return 1;
        if (launch_sampler_update(kind, xbuf, eps, noise, u->coefs_dev, u->state_dev, n, xbuf, all_steps, nullptr, n, st,
                                  objective, xstart))
            return 1;
        return launch_step_advance(u->state_dev, st);
    };
    auto finish = [&]() -> int {  // out = x_0 [ (x + 1) / 2 ]
        if (launch_finalize(xbuf, out, n, unnormalize, s)) return 1;
        if (u->mark_done(s)) return 1;
        if (own_stream) DM_CHECK_HIP(hipStreamSynchronize(s));
        return 0;
    };

    if (!use_graph) {
        // profiling leg: park the GPU while the host enqueues, so that event intervals are kernel times (<= 8 steps)
        if (prof::enabled() && n_steps <= 8 && launch_spin(8.0 * n_steps, s)) return 1;
        for (int i = 0; i < n_steps; ++i)
            if (one_step(s)) return 1;
        return finish();
    }
    dm_unet::GraphKey key;
    key.kind = kind; key.B = B; key.H = H; key.W = W; key.ctx_tokens = ctx_tokens; key.cond_channels = cond_channels;
    key.objective = objective; key.self_cond = self_cond;
    key.noise = noise; key.all_steps = all_steps; key.ws = u->ws; key.times = u->times_dev; key.coefs = u->coefs_dev;
    if (!u->gexec || !(u->gkey == key)) {
        u->drop_graph();
        DM_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = one_step(s);
        hipGraph_t graph = nullptr;
        hipError_t ce = hipStreamEndCapture(s, &graph);
        if (rc || ce != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            if (!rc) set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
            return 1;
        }
        hipGraphExec_t exec = nullptr;
        hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
            (void)hipGraphDestroy(graph);
            set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
            return 1;
        }
        u->graph = graph;
        u->gexec = exec;
        u->gkey = key;
        u->graph_captures += 1;
    }
    for (int i = 0; i < n_steps; ++i) DM_CHECK_HIP(hipGraphLaunch(u->gexec, s));
    return finish();
}

extern "C" {

int dm_sample(dm_unet* u, int kind, int n_steps, const int64_t* times_host, const float* coefs_host,
              const float* x_T, const float* noise, uint64_t seed, uint64_t sample_offset, const float* ctx,
              int ctx_tokens, float* out, float* all_steps, int B, int H, int W, int unnormalize, int use_graph,
              void* stream) {
    return sample_impl(u, kind, n_steps, times_host, coefs_host, x_T, noise, seed, sample_offset, ctx, ctx_tokens,
                       nullptr, 0, out, all_steps, B, H, W, unnormalize, use_graph, stream);
}

int dm_sample_cond(dm_unet* u, int kind, int n_steps, const int64_t* times_host, const float* coefs_host,
                   const float* x_T, const float* noise, uint64_t seed, uint64_t sample_offset, const float* ctx,
                   int ctx_tokens, const float* cond, int cond_channels, float* out, float* all_steps, int B, int H,
                   int W, int unnormalize, int use_graph, void* stream) {
    DM_REQUIRE(cond && cond_channels > 0, "dm_sample_cond needs a condition image");
    return sample_impl(u, kind, n_steps, times_host, coefs_host, x_T, noise, seed, sample_offset, ctx, ctx_tokens, cond,
                       cond_channels, out, all_steps, B, H, W, unnormalize, use_graph, stream);
}

int dm_sample_ex(dm_unet* u, const dm_sample_args* a) {
    DM_REQUIRE(u && a, "null argument");
    return sample_impl(u, a->kind, a->n_steps, a->times_host, a->coefs_host, a->x_T, a->noise, a->seed, a->sample_offset,
                       a->ctx, a->ctx_tokens, a->cond, a->cond_channels, a->out, a->all_steps, a->B, a->H, a->W,
                       a->unnormalize, a->use_graph, a->stream, a->objective, a->self_condition);
}

int dm_randn(float* out, int64_t n, uint64_t seed, uint64_t draw, uint64_t element_offset, void* stream) {
    DM_REQUIRE(out && n >= 0, "bad argument");
    return launch_randn(out, n, seed, draw, element_offset, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// single-operator entry points (dm_op_*) and the VAE decoder share the helpers above
#include "dm_ops.inc"
#include "dm_vae.inc"
#include "dm_consumer.inc"
#include "dm_train.inc"
#include "dm_train_ops.inc"

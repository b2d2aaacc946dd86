// Context, weight repacking and the forward graph of MICA (reference models/model.py:331-348) as a
// fixed sequence of HIP kernel launches on one stream, behind the C ABI of include/mica_hip.h.
#include "../../include/mica_hip.h"
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

using namespace mica;

namespace {

thread_local std::string g_create_err;      // mica_last_error(NULL): the calling thread's last refused mica_create

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

struct ConvLayer {
    std::string name;
    int cout = 0, cin = 0, k = 1;
    std::vector<int> seg_c, seg_cp;   // channel segmentation of the (virtual) concatenated input
    int total_chunks = 0;
    float cout_scale = 1.f;           // folded output scale (FPN softmax weight, model.py:201-205)
    float wscale = 1.f;               // power of two bringing max|w| to ~4096 (f16 hi/lo stay normal)
    bool per_tile = false;            // weights re-packed per tile with a gate folded in (cin_scale)
    bool wino = false;                // 3^3 conv: Winograd-along-x kernel and operand layout
    bool f43 = false;                 // ... F(4,3) (kernels_conv43.hip: the 3^3 convs of encoder.2) instead of F(2,3)
    float* d_w = nullptr;             // torch layout f32
    float* d_b = nullptr;             // bias (already times cout_scale)
    _Float16* d_wpk = nullptr;        // packed (static) or per-tile buffer [maxB][...]
    float* d_cin_scale = nullptr;     // [maxB][cin] for per_tile layers
    int64_t pk_halves = 0;
    double flops_per_voxel = 0;
};

struct GateMLP {
    int C = 0, Ch = 0;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
};

struct Enc {
    int C;
    ConvLayer conv1, conv2, conv3, fusion, transition;
    GateMLP se, ga;
    float *dw_w = nullptr, *dw_b = nullptr;   // [27][C], [C]
};

struct Head {
    int ncls;
    ConvLayer conv1, conv2;
    GateMLP cal;
    float *wf = nullptr, *bf = nullptr;
};

}  // namespace

constexpr int PROF_KINDS = 7;

struct mica_ctx {
    int device = 0, maxB = 1, S = 64;
    Dims d{64, 64, 64};
    int V = 0;
    std::string err;
    bool finalized = false;
    std::map<std::string, HostTensor> host;
    std::vector<void*> allocs;
    int64_t bytes = 0;

    // weights
    float *stem_w = nullptr, *stem_b = nullptr;
    _Float16* stem_rec = nullptr;     // the stem on the matrix cores (kernels_stem.hip): packed K-step records, block offsets, plan
    int* stem_aoff = nullptr;
    StemPlan stem_plan{};
    float stem_wscale = 1.f;
    int stem_mode = 1;               // 0: the f32 VALU stem everywhere (A/B switch, MICA_STEM_MFMA=0)
    int raw_cblk = 32;               // channel block of the raw tensors around the depthwise conv (0: plain NDHWC; A/B switch, MICA_RAW_CBLK=0)
    GateMLP exp_att;
    ConvLayer downsizing, feat_conv, fusion0;
    float *fg_w0 = nullptr, *fg_b0 = nullptr, *fg_w2 = nullptr, *fg_b2 = nullptr;
    Enc enc[3];
    ConvLayer lateral[3], smooth[3];
    Head heads[3];

    // activations
    // operands of 3^3 convs are in wino layout (2x bytes); the 1x1 convs read plain split operands (S_exp, S_fw) or raw f32 tensors
    _Float16 *S_exp, *S_af, *S_fw, *S_x0, *S_1, *S_2, *S_f, *S_c[3], *S_l, *S_fpn, *S_extra, *S_h1;
    float* extra_raw = nullptr;   // [B][8][V] backbone + CA logits (NCDHW) feeding the next heads' conv1
    float *R_a, *R_b, *R_c;
    float *logits[3];             // internal NCDHW logits when the caller wants probabilities only
    float* ws = nullptr;          // reduction partials
    float* ws_gap = nullptr;      // depthwise: per-block sums of its normalised input [B][blocks][C]
    float *v_mean, *v_rstd, *v_mean3, *v_rstd3, *v_pool, *v_gse, *v_gate, *v_abs;
    int* d_err = nullptr;            // range flags, one per tile of the call (int[maxB])
    int* cur_err = nullptr;          // ... of the run of tiles forward_run is working on
    float ascale = ASCALE_DEFAULT;   // activation scale of the split encoding (common.h) every forward call starts from
    int f43_mode = 3;                // 0: every 3^3 conv on the F(2,3) kernel; 1: encoder.2's four convs on the F(4,3) kernel; 2: those and
                                     // encoder.1's transition; 3 (default since round 5): mode 1 and the late narrow layers (FPN smooth x3, the
                                     // heads' conv1 x3) on its 64-channel variant
    float last_scale = ASCALE_DEFAULT;   // the lowest scale a tile of the last forward call needed (forward_checked)
    int last_retries = 0;                // tiles of the last forward call that had to be repeated at a lower scale
    int last_input_runs = 0;             // runs of equal AF3 gate the last forward_impl cut its batch into (MultiScaleInput launches)
    bool trunk_per_run = false;          // A/B switch (MICA_TRUNK_PER_RUN=1): the whole network once per run, as rounds 1-5 did
    std::vector<char> use_af;        // per tile of the last forward_impl call: AF3 branch taken
    float* h_abs = nullptr;       // pinned
    int* h_err = nullptr;         // pinned

    // profiling: HIP events around the launches of kind 0 = every dense conv (= kinds 2 + 4 + 5 + 6, work = FLOPs), 1 = depthwise
    // conv3d (work = bytes), 2 = 3^3 convs (Winograd kernel), 3 = operand passes (prep kernels, work = bytes), 4 = 1x1 convs
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_kind;
    size_t ev_used = 0;
    double prof_work[PROF_KINDS] = {};
    double last_ms[PROF_KINDS] = {}, last_work[PROF_KINDS] = {};
    int64_t last_launches[PROF_KINDS] = {};
};

namespace {

#define HIPC(ctx, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return MICA_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)
// Entry of every call that launches work: the context's device becomes current.  The runtime's thread-local last error is NOT
// cleared here: an error of the host program that is still pending on this thread (say a failed launch of its own that it has not
// looked at yet) stays where the host will find it, and the call is refused rather than launched behind it - the launch checks at
// the end of a call (CHECK_LAUNCHES) could not tell it from an error of this call's launches.  (The library's own failed hipMallocs
// clear what they report: dalloc, Tmp::get.)  hipErrorNotReady is a status of hipEventQuery / hipStreamQuery, not a failure.
#define MICA_ENTER(ctx)                                                                              \
    do {                                                                                             \
        HIPC(ctx, hipSetDevice((ctx)->device));                                                      \
        hipError_t pre_ = hipPeekAtLastError();                                                      \
        if (pre_ != hipSuccess && pre_ != hipErrorNotReady) {                                        \
            (ctx)->err = std::string("a HIP error of the calling program is pending on this thread (") + hipGetErrorString(pre_) + \
                         "): left in place, nothing launched";                                       \
            return MICA_ERR_STATE;                                                                   \
        }                                                                                            \
        (void)take_launch_refusal();                                                                 \
    } while (0)
// After the launches of a call: a launch helper that refused its shape (common.h: refuse_launch) -> MICA_ERR_ARG; a failed launch ->
// MICA_ERR_HIP.
#define CHECK_LAUNCHES(ctx)                                                                          \
    do {                                                                                             \
        if (const char* why_ = take_launch_refusal()) {                                              \
            (ctx)->err = why_;                                                                       \
            return MICA_ERR_ARG;                                                                     \
        }                                                                                            \
        hipError_t e_ = hipGetLastError();                                                           \
        if (e_ != hipSuccess && e_ != hipErrorNotReady) {                                            \
            (ctx)->err = std::string("kernel launch: ") + hipGetErrorString(e_);                     \
            return MICA_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

template <typename T> int dalloc(mica_ctx* c, T** p, int64_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (size_t)(n * (int64_t)sizeof(T)));
    if (e != hipSuccess) {
        c->err = "hipMalloc(" + std::to_string(n * (int64_t)sizeof(T)) + " B): " + hipGetErrorString(e);
        (void)hipGetLastError();      // reported here: the runtime's sticky last-error must not fail the next call's hipGetLastError() check
        return MICA_ERR_HIP;
    }
    c->allocs.push_back(q);
    c->bytes += n * (int64_t)sizeof(T);
    *p = (T*)q;
    return MICA_OK;
}

int upload(mica_ctx* c, float** p, const std::vector<float>& h) {
    int r = dalloc(c, p, (int64_t)h.size());
    if (r) return r;
    HIPC(c, hipMemcpy(*p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return MICA_OK;
}

const HostTensor* find(mica_ctx* c, const std::string& n) {
    auto it = c->host.find(n);
    return it == c->host.end() ? nullptr : &it->second;
}

bool shape_is(const HostTensor* t, std::initializer_list<int64_t> s) {
    return t && t->shape == std::vector<int64_t>(s);
}

int pad16(int c) { return (c + 15) / 16 * 16; }

// weight scale: power of two so that max|w * cout_scale| * wscale is in (2048, 4096]
float pick_wscale(const std::vector<float>& w, float cs) {
    float m = 0.f;
    for (float v : w) m = std::fmax(m, std::fabs(v * cs));
    if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
    int e = 12 - (int)std::ceil(std::log2(m));
    if (e > 24) e = 24;
    if (e < -8) e = -8;
    return std::ldexp(1.f, e);
}

int setup_conv(mica_ctx* c, ConvLayer& L, const std::string& name, int cout, int k, std::vector<int> seg_c, bool per_tile,
               float cout_scale = 1.f, bool f43 = false) {
    L.name = name;
    L.cout = cout;
    L.k = k;
    L.seg_c = seg_c;
    L.per_tile = per_tile;
    L.cout_scale = cout_scale;
    L.cin = 0;
    L.seg_cp.clear();
    L.total_chunks = 0;
    for (int s : seg_c) {
        L.cin += s;
        L.seg_cp.push_back(pad16(s));
        L.total_chunks += pad16(s) / 16;
    }
    const HostTensor* w = find(c, name + ".weight");
    const HostTensor* b = find(c, name + ".bias");
    if (!shape_is(w, {cout, L.cin, k, k, k}) || !shape_is(b, {cout})) {
        c->err = "weight tensor missing or mis-shaped: " + name;
        return MICA_ERR_STATE;
    }
    L.wscale = pick_wscale(w->data, cout_scale);
    L.flops_per_voxel = 2.0 * k * k * k * (double)L.cin * cout;
    int r = upload(c, &L.d_w, w->data);
    if (r) return r;
    std::vector<float> bs(b->data);
    for (float& v : bs) v *= cout_scale;
    r = upload(c, &L.d_b, bs);
    if (r) return r;
    L.wino = (k == 3);
    L.f43 = L.wino && f43 && conv_wino43_eligible(cout);
    L.pk_halves = L.f43 ? packed_weight_halves_wino43(cout, L.total_chunks)
                        : L.wino ? packed_weight_halves_wino(cout, L.total_chunks) : packed_weight_halves(cout, k, L.total_chunks);
    r = dalloc(c, &L.d_wpk, L.pk_halves * (per_tile ? c->maxB : 1));
    if (r) return r;
    if (per_tile) {
        r = dalloc(c, &L.d_cin_scale, (int64_t)c->maxB * L.cin);
        if (r) return r;
        launch_fill_float(L.d_cin_scale, (int64_t)c->maxB * L.cin, 1.0f, 0);
    } else if (L.f43) {
        launch_pack_weights_wino43(L.d_w, cout, L.cin, L.seg_c.data(), L.seg_cp.data(), (int)L.seg_c.size(), nullptr, 1, cout_scale,
                                   L.wscale, L.d_wpk, 0);
    } else if (L.wino) {
        launch_pack_weights_wino(L.d_w, cout, L.cin, L.seg_c.data(), L.seg_cp.data(), (int)L.seg_c.size(), nullptr, 1, cout_scale,
                                 L.wscale, L.d_wpk, 0);
    } else {
        launch_pack_weights(L.d_w, cout, L.cin, k, L.seg_c.data(), L.seg_cp.data(), (int)L.seg_c.size(), nullptr, 1,
                            cout_scale, L.wscale, L.d_wpk, 0);
    }
    return MICA_OK;
}

int setup_gate(mica_ctx* c, GateMLP& g, const std::string& n1, const std::string& n2, int C, int Ch, bool linear) {
    g.C = C;
    g.Ch = Ch;
    const HostTensor *w1 = find(c, n1 + ".weight"), *b1 = find(c, n1 + ".bias");
    const HostTensor *w2 = find(c, n2 + ".weight"), *b2 = find(c, n2 + ".bias");
    bool ok = linear ? (shape_is(w1, {Ch, C}) && shape_is(w2, {C, Ch}))
                     : (shape_is(w1, {Ch, C, 1, 1, 1}) && shape_is(w2, {C, Ch, 1, 1, 1}));
    if (!ok || !shape_is(b1, {Ch}) || !shape_is(b2, {C})) {
        c->err = "gate weights missing or mis-shaped: " + n1;
        return MICA_ERR_STATE;
    }
    int r;
    if ((r = upload(c, &g.w1, w1->data))) return r;
    if ((r = upload(c, &g.b1, b1->data))) return r;
    if ((r = upload(c, &g.w2, w2->data))) return r;
    if ((r = upload(c, &g.b2, b2->data))) return r;
    return MICA_OK;
}

SplitView view(_Float16* p, int chunks_total, int off, int chunks) { return SplitView{p, chunks_total, off, chunks}; }

struct SrcList {
    ConvSrcs s{};
    SrcList() { s.n = 0; }
    SrcList& add(const _Float16* p, int chunks_total, int off, int chunks) {
        s.p[s.n] = p; s.chunks_total[s.n] = chunks_total; s.chunk_off[s.n] = off; s.chunks[s.n] = chunks; ++s.n;
        return *this;
    }
};

void prof_begin(mica_ctx* c, int kind, hipStream_t st) {
    if (!c->profiling) return;
    while (c->ev.size() < c->ev_used + 2) { hipEvent_t e; hipEventCreate(&e); c->ev.push_back(e); c->ev_kind.push_back(0); }
    c->ev_kind[c->ev_used] = kind;
    hipEventRecord(c->ev[c->ev_used], st);
}
void prof_end(mica_ctx* c, int kind, double work, hipStream_t st) {
    if (!c->profiling) return;
    hipEventRecord(c->ev[c->ev_used + 1], st);
    c->ev_used += 2;
    c->prof_work[kind] += work;
}

// Launch a dense 3x3x3 conv (Winograd kernel).  With `mean`/`rstd` given the InstanceNorm statistics of the output are
// produced too (fused into the kernel's epilogue, merged by stats_finalize).
void run_conv(mica_ctx* c, ConvLayer& L, const SrcList& src, float* out, int B, hipStream_t st, float* mean = nullptr,
              float* rstd = nullptr, int out_cblk = 0) {
    const int pk = !L.f43 ? 2 : L.cout % 128 == 0 ? 5 : 6;       // F(2,3) kernel | F(4,3) kernel, 128-channel blocks | its tap-split 64-channel variant
    prof_begin(c, pk, st);
    const int P = L.f43 ? launch_conv_wino43(src.s, L.d_wpk, 0, L.d_b, 1.0f / (L.wscale * (c->ascale / WINO43_ASCALE_DIV)), out, B, c->d, L.cout,
                                             mean ? c->ws : nullptr, st, out_cblk)
                        : launch_conv_wino(src.s, L.d_wpk, 0, L.d_b, 1.0f / (L.wscale * c->ascale), out, B, c->d, L.cout, mean ? c->ws : nullptr, st,
                                           out_cblk);
    prof_end(c, pk, L.flops_per_voxel * (double)c->V * B, st);
    if (mean) launch_stats_finalize(c->ws, B, P, L.cout, 1e-5f, mean, rstd, st);
}

Conv1Src split_src(const _Float16* p, int chunks_total, int off, int chunks) {
    return Conv1Src{p, nullptr, nullptr, 0, chunks, chunks_total, off, 0, 0};
}
Conv1Src raw_src(const float* p, int channels, const float* mean, const float* rstd, int relu, int cblk = 0) {
    return Conv1Src{p, mean, rstd, 1, channels / 16, channels / 16, 0, relu, cblk};
}

void make_operand(mica_ctx* c, const float* raw, int B, int C, const float* mean, const float* rstd, int relu, SplitView t3, SplitView t1,
                  float* gap, hipStream_t st, bool f43 = false);

// A 1x1x1 conv whose output is the Winograd operand `dst` of the 3^3 conv that follows.  The kernel writes the operand itself
// when a workgroup owns whole x rows; otherwise it writes raw f32 to `tmp_raw` and the operand pass follows.
void run_conv1x1(mica_ctx* c, ConvLayer& L, Conv1Src a, const Conv1Src* b2, SplitView dst, float* tmp_raw, int B, hipStream_t st,
                 bool f43 = false) {
    if (L.per_tile)
        launch_pack_weights(L.d_w, L.cout, L.cin, 1, L.seg_c.data(), L.seg_cp.data(), (int)L.seg_c.size(), L.d_cin_scale, B, L.cout_scale,
                            L.wscale, L.d_wpk, st);
    Conv1Srcs src{};
    src.n = b2 ? 2 : 1;
    src.s[0] = a;
    if (b2) src.s[1] = *b2;
    const bool fused = conv1x1_can_emit_wino(c->d, f43 ? 2 : 1);
    prof_begin(c, 4, st);
    launch_conv1x1(src, L.d_wpk, L.per_tile ? L.pk_halves : 0, L.d_b, 1.0f / (L.wscale * c->ascale), fused ? nullptr : tmp_raw,
                   fused ? dst : SplitView{nullptr, 0, 0, 0}, B, c->d, L.cout, SplitEnc{c->cur_err, c->ascale}, st, f43 ? 2 : 1,
                   f43 ? c->ascale / WINO43_ASCALE_DIV : c->ascale);
    prof_end(c, 4, L.flops_per_voxel * (double)c->V * B, st);
    if (!fused) make_operand(c, tmp_raw, B, L.cout, nullptr, nullptr, 0, dst, SplitView{nullptr, 0, 0, 0}, nullptr, st, f43);
}

void gate(mica_ctx* c, const GateMLP& g, const float* pool, const float* premul, int B, const float* postmul, float* out,
          float* out_post, int post_stride, hipStream_t st) {
    launch_gate_mlp(pool, premul, B, g.C, g.Ch, g.w1, g.b1, g.w2, g.b2, postmul, out, out_post, post_stride, st);
}

// Re-encode a raw conv output as the operand(s) of its consumers: `t3` feeds 3^3 convs (wino layout when the
// Winograd path is on, else plain split), `t1` feeds 1x1 convs (always plain split).  Either may be empty.
void make_operand(mica_ctx* c, const float* raw, int B, int C, const float* mean, const float* rstd, int relu, SplitView t3,
                  SplitView t1, float* gap, hipStream_t st, bool f43) {
    prof_begin(c, 3, st);
    // algorithmic bytes: the f32 tensor read once, each operand written once (wino layout = 8 B, wino43 = 6 B, plain split = 4 B per value)
    const double bytes = (double)B * c->V * C * (4.0 + (t3.p ? (f43 ? 6.0 : 8.0) : 0.0) + (t1.p ? 4.0 : 0.0));
    struct End { mica_ctx* c; double b; hipStream_t s; ~End() { prof_end(c, 3, b, s); } } end_{c, bytes, st};
    if (t3.p && f43) {
        // F(4,3) operand of encoder.2's convs: no plain / pooled side outputs are ever wanted there
        launch_prep_wino43(raw, B, c->d, C, mean, rstd, relu, t3, SplitEnc{c->cur_err, c->ascale / WINO43_ASCALE_DIV}, st);
    } else if (t3.p) {
        launch_prep_wino(raw, B, c->d, C, mean, rstd, relu, nullptr, t3, t1, gap, c->ws, SplitEnc{c->cur_err, c->ascale}, st);
    } else {
        launch_prep(raw, B, c->V, C, mean, rstd, relu, nullptr, t1, nullptr, gap, c->ws, SplitEnc{c->cur_err, c->ascale}, st);
    }
}

// The stem: on the matrix cores for tile widths that are multiples of 64 (kernels_stem.hip), else the f32 VALU kernel; both write the
// split view of the 128 channels and / or the raw tensor, and the per-tile channel means (gap, nullable).
void run_stem(mica_ctx* c, const float* d_map, int B, Dims d, SplitView out, float* out_raw, float* gap, SplitEnc enc, hipStream_t st) {
    if (c->stem_mode != 0 && stem_mfma_eligible(d)) {
        const int nblk = launch_stem_mfma(d_map, B, d, c->stem_rec, c->stem_aoff, c->stem_plan, c->stem_wscale, c->stem_b, out, out_raw,
                                          gap ? c->ws : nullptr, enc, st);
        if (gap) launch_finalize_sum(c->ws, B, nblk, 128, 1.0f / (float)(d.D * d.H * d.W), gap, st);
        return;
    }
    launch_stem(d_map, B, d, c->stem_w, c->stem_b, out, out_raw, gap, c->ws, enc, st);
}

// MultiScaleInput (model.py:43-74) for ONE run of B consecutive tiles that take the same branch - the only part of the network in
// which a tile with AF3 atoms and one without differ (model.py:56-63 against :69-74).  Every buffer it touches is scratch at slots
// 0 .. B-1 (the runs of a call follow each other on the stream), except its result: x0 of the run lands at slots slot0 .. slot0+B-1
// of S_x0, where `trunk` finds the whole batch, and the range flags are the tiles' own (d_err + err0).
void input_run(mica_ctx* c, const float* d_map, const float* d_af, int B, bool use_af, hipStream_t st, int slot0, int err0) {
    const int V = c->V;
    const Dims d = c->d;
    c->cur_err = c->d_err + err0;
    // S_x0 is a Winograd F(2,3) operand [B][4 chunks][4 p][4 q][Vh][8] (encoder.0 never runs on the F(4,3) kernel)
    const int64_t x0_tile = (int64_t)4 * 16 * ((int64_t)d.D * d.H * ((d.W + 1) / 2)) * 8;
    const SplitView x0 = view(c->S_x0 + (int64_t)slot0 * x0_tile, 4, 0, 4);
    run_stem(c, d_map, B, d, view(c->S_exp, 8, 0, 8), nullptr, c->v_pool, SplitEnc{c->cur_err, c->ascale}, st);
    if (!use_af) {
        gate(c, c->exp_att, c->v_pool, nullptr, B, nullptr, nullptr, c->downsizing.d_cin_scale, 128, st);
        run_conv1x1(c, c->downsizing, split_src(c->S_exp, 8, 0, 8), nullptr, x0, c->R_a, B, st);
    } else {
        gate(c, c->exp_att, c->v_pool, nullptr, B, nullptr, nullptr, c->fusion0.d_cin_scale, 192, st);
        launch_prep_ncdhw_wino(d_af, B, d, 24, view(c->S_af, 2, 0, 2), SplitEnc{c->cur_err, c->ascale}, st);
        run_conv(c, c->feat_conv, SrcList().add(c->S_af, 2, 0, 2), c->R_b, B, st);
        launch_feat_gate(c->R_b, B, V, c->fg_w0, c->fg_b0, c->fg_w2, c->fg_b2, view(c->S_fw, 4, 0, 4), SplitEnc{c->cur_err, c->ascale}, st);
        const Conv1Src fw = split_src(c->S_fw, 4, 0, 4);
        run_conv1x1(c, c->fusion0, split_src(c->S_exp, 8, 0, 8), &fw, x0, c->R_a, B, st);
    }
}

// Everything behind MultiScaleInput - encoders, FPN, heads (model.py:336-346) - ONCE for all B tiles of the call, whatever mix of
// branches produced their x0 (round 6; until round 5 the whole network ran once per run of equal gates, i.e. 2-4 small forwards per
// batch on a map whose docked model covers part of the box).  Workspace slots 0..B-1.
int trunk(mica_ctx* c, int B, float* o_bb, float* o_ca, float* o_aa, hipStream_t st, int slot0 = 0) {
    const int V = c->V;
    const Dims d = c->d;
    SplitView none{nullptr, 0, 0, 0};
    c->cur_err = c->d_err + slot0;
    // ---- encoders (model.py:149-152) -----------------------------------------------------------
    const _Float16* X = c->S_x0;
    for (int e = 0; e < 3; ++e) {
        Enc& E = c->enc[e];
        const int C = E.C, cc = C / 16, ch = C / 32;   // chunks of C and of C/2
        // encoder.2's four 3^3 convs (and, in mode 2, encoder.1's transition) run on the F(4,3) kernel: every operand they read
        // (x = c_1, x1, x2, the fusions' outputs) is written in that kernel's layout by its producer; no tensor is needed in both layouts
        const bool f43 = E.conv1.f43, f43_next = e < 2 && c->enc[e + 1].conv1.f43;
        // ResidualDenseBlock (model.py:130-134)
        run_conv(c, E.conv1, SrcList().add(X, cc, 0, cc), c->R_a, B, st, c->v_mean, c->v_rstd);
        make_operand(c, c->R_a, B, C / 2, c->v_mean, c->v_rstd, 1, view(c->S_1, ch, 0, ch), none, nullptr, st, f43);
        run_conv(c, E.conv2, SrcList().add(X, cc, 0, cc).add(c->S_1, ch, 0, ch), c->R_a, B, st, c->v_mean, c->v_rstd);
        make_operand(c, c->R_a, B, C / 2, c->v_mean, c->v_rstd, 1, view(c->S_2, ch, 0, ch), none, nullptr, st, f43);
        // conv3's raw output and the depthwise conv's live in the blocked raw layout [C / 32][V][32] (common.h): their readers - the
        // depthwise conv, the 1x1 fusion - work on 32-channel slabs / chunk pairs, which are contiguous there
        const int cb = c->raw_cblk;
        run_conv(c, E.conv3, SrcList().add(X, cc, 0, cc).add(c->S_1, ch, 0, ch).add(c->S_2, ch, 0, ch), c->R_b, B, st, c->v_mean3,
                 c->v_rstd3, cb);
        // x3 = relu(IN(conv3)) is never materialised: the depthwise conv and the 1x1 fusion normalise the raw tensor on load,
        // and its global average pool (the SE gate's input, model.py:256) is summed by the depthwise kernel as it loads.
        // The SE gate g (per tile and channel, > 0) scales the depthwise conv's INPUT (model.py:258, 99); the conv is linear, so
        // it runs on the ungated tensor (R_c = w * x3 + b) and the gate moves into the InstanceNorm constants of its output:
        // IN(g (R_c - b) + b) = (R_c - mean) g / sqrt(g^2 var + eps).
        // DualAttention (model.py:98-101): local branch
        {
            prof_begin(c, 1, st);
            const int P = launch_depthwise(c->R_b, B, d, C, c->v_mean3, c->v_rstd3, nullptr, E.dw_w, E.dw_b, c->R_c, c->ws, c->ws_gap, st, cb);
            prof_end(c, 1, 8.0 * (double)C * V * B, st);     // algorithmic bytes: read + write 4 B per voxel and channel
            launch_finalize_sum(c->ws_gap, B, P, C, 1.0f / (float)V, c->v_pool, st);
            gate(c, E.se, c->v_pool, nullptr, B, nullptr, c->v_gse, nullptr, 0, st);      // SEBlock gate (model.py:254-258)
            launch_stats_finalize(c->ws, B, P, C, 1e-5f, c->v_mean, c->v_rstd, st, c->v_gse);
        }
        // global branch: GAP(se(x3)) = g_se * GAP(x3); global_feat = g_ga * g_se * x3 folded into fusion's weights
        gate(c, E.ga, c->v_pool, c->v_gse, B, c->v_gse, nullptr, E.fusion.d_cin_scale + C, 2 * C, st);
        {
            // fusion (model.py:96, 101) reads the two branches raw: local = relu(IN(depthwise)), global = relu(IN(conv3)) * gates
            const Conv1Src glob = raw_src(c->R_b, C, c->v_mean3, c->v_rstd3, 1, cb);
            run_conv1x1(c, E.fusion, raw_src(c->R_c, C, c->v_mean, c->v_rstd, 1, cb), &glob, view(c->S_f, cc, 0, cc), c->R_a, B, st, E.transition.f43);
        }
        // transition (model.py:141-147); c_e feeds the next encoder's 3^3 convs and the FPN's 1x1 lateral
        run_conv(c, E.transition, SrcList().add(c->S_f, cc, 0, cc), c->R_a, B, st, c->v_mean, c->v_rstd);
        if (e < 2) make_operand(c, c->R_a, B, 2 * C, c->v_mean, c->v_rstd, 1, view(c->S_c[e], 2 * cc, 0, 2 * cc), none, nullptr, st, f43_next);
        // FPN level e right away (model.py:182-205; the interpolations are identities): the lateral 1x1 reads c_e raw with the
        // transition's InstanceNorm + ReLU applied on load, and writes the smoothing conv's operand
        // (conv variant 3: the smooth conv and the heads' conv1 run on the F(4,3) kernel's 64-channel variant; their operands - S_l, the
        // FPN output, the earlier heads' logits - are written in that kernel's layout by their producers)
        run_conv1x1(c, c->lateral[e], raw_src(c->R_a, 2 * C, c->v_mean, c->v_rstd, 1), nullptr, view(c->S_l, 4, 0, 4), c->R_c, B, st, c->smooth[e].f43);
        run_conv(c, c->smooth[e], SrcList().add(c->S_l, 4, 0, 4), c->R_b, B, st);
        make_operand(c, c->R_b, B, 64, nullptr, nullptr, 0, view(c->S_fpn, 12, 4 * e, 4), none, nullptr, st, c->heads[0].conv1.f43);
        X = c->S_c[e];
    }
    // ---- heads (model.py:230-239, 344-346) ------------------------------------------------------
    hipMemsetAsync(c->extra_raw, 0, sizeof(float) * (size_t)B * 8 * V, st);
    float* outs[3] = {o_bb, o_ca, o_aa};
    for (int h = 0; h < 3; ++h) {
        Head& H = c->heads[h];
        SrcList src;
        src.add(c->S_fpn, 12, 0, 12);
        if (h > 0) src.add(c->S_extra, 1, 0, 1);
        run_conv(c, H.conv1, src, c->R_a, B, st, c->v_mean, c->v_rstd);
        make_operand(c, c->R_a, B, 64, c->v_mean, c->v_rstd, 1, view(c->S_h1, 4, 0, 4), none, nullptr, st);
        run_conv(c, H.conv2, SrcList().add(c->S_h1, 4, 0, 4), c->R_b, B, st, c->v_mean, c->v_rstd);
        launch_prep(c->R_b, B, V, 32, c->v_mean, c->v_rstd, 1, nullptr, none, nullptr, c->v_pool, c->ws, SplitEnc{c->cur_err, c->ascale}, st);
        gate(c, H.cal, c->v_pool, nullptr, B, nullptr, c->v_gate, nullptr, 0, st);
        const bool feeds = h < 2;
        launch_head_final(c->R_b, B, V, c->v_mean, c->v_rstd, c->v_gate, H.wf, H.bf, H.ncls, outs[h], 4 * h,
                          feeds ? c->extra_raw : nullptr, 8, st);
        if (feeds && c->heads[0].conv1.f43)
            launch_prep_ncdhw_wino43(c->extra_raw, B, d, 8, view(c->S_extra, 1, 0, 1), SplitEnc{c->cur_err, c->ascale / WINO43_ASCALE_DIV}, st);
        else if (feeds) launch_prep_ncdhw_wino(c->extra_raw, B, d, 8, view(c->S_extra, 1, 0, 1), SplitEnc{c->cur_err, c->ascale}, st);
    }
    return MICA_OK;
}

int forward_impl(mica_ctx* c, const float* d_map, const float* d_af, int B, int af_mode, float* o_bb, float* o_ca, float* o_aa,
                 hipStream_t st, const char* force_use = nullptr) {
    if (!c->finalized) { c->err = "weights not finalized"; return MICA_ERR_STATE; }
    if (B < 1 || B > c->maxB) { c->err = "batch out of range [1, max_batch]"; return MICA_ERR_ARG; }
    if (!d_map || !o_bb || !o_ca || !o_aa) { c->err = "null pointer argument"; return MICA_ERR_ARG; }
    if (af_mode < MICA_AF_NONE || af_mode > MICA_AF_ALWAYS) { c->err = "bad af_mode"; return MICA_ERR_ARG; }
    MICA_ENTER(c);
    const int V = c->V;
    c->ev_used = 0;
    for (double& w : c->prof_work) w = 0;
    HIPC(c, hipMemsetAsync(c->d_err, 0, sizeof(int) * B, st));
    std::vector<char> use(B, 0);
    if (force_use) {
        for (int b = 0; b < B; ++b) use[b] = force_use[b];
    } else if (d_af && af_mode == MICA_AF_ALWAYS) {
        for (int b = 0; b < B; ++b) use[b] = 1;        // the caller evaluated the test itself (a batch cut into several calls)
    } else if (d_af && af_mode != MICA_AF_NONE) {
        // is_af_zero = af.abs().sum() < 1e-6  (model.py:60): device reduction, one small D2H per call
        HIPC(c, hipMemsetAsync(c->v_abs, 0, sizeof(float) * B, st));
        launch_abs_sum(d_af, B, (int64_t)24 * V, c->v_abs, st);
        HIPC(c, hipMemcpyAsync(c->h_abs, c->v_abs, sizeof(float) * B, hipMemcpyDeviceToHost, st));
        HIPC(c, hipStreamSynchronize(st));
        if (af_mode == MICA_AF_BATCH) {
            double tot = 0;
            for (int b = 0; b < B; ++b) tot += c->h_abs[b];
            for (int b = 0; b < B; ++b) use[b] = !(tot < 1e-6);
        } else {
            for (int b = 0; b < B; ++b) use[b] = !(c->h_abs[b] < 1e-6f);
        }
    }
    c->use_af = use;
    c->last_input_runs = 0;
    for (int b0 = 0; b0 < B;) {                      // MultiScaleInput per run of tiles with equal gate, into their slots of x0
        int b1 = b0 + 1;
        while (b1 < B && use[b1] == use[b0]) ++b1;
        const bool whole_net = c->trunk_per_run;     // A/B switch: rounds 1-5 ran the whole network once per run (workspace slots 0 .. n-1)
        input_run(c, d_map + (int64_t)b0 * V, d_af ? d_af + (int64_t)b0 * 24 * V : nullptr, b1 - b0, use[b0] != 0, st, whole_net ? 0 : b0, b0);
        if (whole_net) {
            int r = trunk(c, b1 - b0, o_bb + (int64_t)b0 * 4 * V, o_ca + (int64_t)b0 * 4 * V, o_aa + (int64_t)b0 * 21 * V, st, b0);
            if (r) return r;
        }
        ++c->last_input_runs;
        b0 = b1;
    }
    if (!c->trunk_per_run) {                         // ... and the rest of the network once for the whole batch
        int r = trunk(c, B, o_bb, o_ca, o_aa, st);
        if (r) return r;
    }
    CHECK_LAUNCHES(c);
    if (c->profiling) {
        HIPC(c, hipStreamSynchronize(st));
        double ms[PROF_KINDS] = {};
        int64_t n[PROF_KINDS] = {};
        for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
            float t = 0;
            hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]);
            ms[c->ev_kind[i]] += t;
            n[c->ev_kind[i]]++;
        }
        ms[0] = ms[2] + ms[4] + ms[5] + ms[6]; n[0] = n[2] + n[4] + n[5] + n[6];
        c->prof_work[0] = c->prof_work[2] + c->prof_work[4] + c->prof_work[5] + c->prof_work[6];
        for (int k = 0; k < PROF_KINDS; ++k) { c->last_ms[k] = ms[k]; c->last_launches[k] = n[k]; c->last_work[k] = c->prof_work[k]; }
    }
    return MICA_OK;
}

}  // namespace

namespace {
struct Tmp {
    std::vector<void*> p;
    ~Tmp() { for (void* q : p) hipFree(q); }
    template <typename T> T* get(int64_t n) {
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)(n * (int64_t)sizeof(T))) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        p.push_back(q);
        return (T*)q;
    }
};
bool pow2_8_512(int c) { return c >= 8 && c <= 512 && (c & (c - 1)) == 0; }
// Box limits of the single-op entry points: the same edges mica_create_dims admits (the conv kernels index their slabs with 32-bit
// offsets, sized for edges <= 128), at most 64 tiles per call.
bool op_box_ok(int batch, int d, int h, int w) { return batch >= 1 && batch <= 64 && d >= 1 && d <= 128 && h >= 1 && h <= 128 && w >= 1 && w <= 128; }
#define OP_BOX_MSG " [box limits: 1 <= batch <= 64, edges in [1, 128]]"
}  // namespace


// The forward with the range check of the split-f16 encoding (synchronises `st`).  An activation beyond the f16 range at the
// context's activation scale (|x| * ascale > 60000) is handled PER TILE and PER CALL: every encoder raises the flag of the tile it
// is working on (int[batch] on the device), and only the flagged tiles are repeated, alone, stepping down by 4 (exact: powers of
// two, undone in the conv epilogues; never past the smallest scale without trying it) until they fit.  A tile's numbers therefore never depend on its batch neighbours, on tiles or
// maps processed earlier, or on which rank of a sharded run met the outlier; the context's scale is not changed.  Below ~1 the
// lo halves of O(1) activations go subnormal in f16 and the whole-network error grows (DESIGN.md section 2 has the measured
// table); mica_get_last_forward_scale() reports the lowest scale the call used so that the caller can tell.  NaN/Inf, or an
// overflow at the smallest scale, fail loudly with MICA_ERR_RANGE rather than return clipped numbers.
static int range_flags(mica_ctx* c, hipStream_t st, int n) {        // -> c->h_err[0..n)
    HIPC(c, hipMemcpyAsync(c->h_err, c->d_err, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    HIPC(c, hipStreamSynchronize(st));
    return MICA_OK;
}
static int range_error(mica_ctx* c, int flag) {
    c->err = (flag & RANGE_NONFINITE) ? "activation outside the representable range of the split-f16 conv path: NaN/Inf in the input or in an activation"
                                      : "activation outside the representable range of the split-f16 conv path (|x| > 1.5e7)";
    return MICA_ERR_RANGE;
}
static int forward_checked(mica_ctx* c, const float* d_map, const float* d_af, int B, int af_mode, float* o_bb, float* o_ca,
                           float* o_aa, hipStream_t st) {
    const float base = c->ascale;
    c->last_scale = base;
    c->last_retries = 0;
    int r = forward_impl(c, d_map, d_af, B, af_mode, o_bb, o_ca, o_aa, st);
    if (r || (r = range_flags(c, st, B))) return r;
    std::vector<int> flags(c->h_err, c->h_err + B);      // one per tile: every encoder reports under the tile it is working on
    int any = 0;
    for (int f : flags) any |= f;
    if (!any) return MICA_OK;
    if (any & RANGE_NONFINITE) return range_error(c, any);
    const std::vector<char> use = c->use_af;          // the gate decisions of the whole call (MICA_AF_BATCH looks at all tiles)
    const int V = c->V;
    for (int b = 0; b < B && !r; ++b) {
        if (!flags[b]) continue;                      // this tile fitted: its numbers stand (they do not depend on its batch neighbours)
        ++c->last_retries;
        if (base <= ASCALE_MIN) { r = range_error(c, RANGE_OVERFLOW); break; }
        c->ascale = std::max(base * 0.25f, ASCALE_MIN);
        for (;;) {
            r = forward_impl(c, d_map + (int64_t)b * V, d_af ? d_af + (int64_t)b * 24 * V : nullptr, 1, af_mode, o_bb + (int64_t)b * 4 * V,
                             o_ca + (int64_t)b * 4 * V, o_aa + (int64_t)b * 21 * V, st, &use[b]);
            if (r || (r = range_flags(c, st, 1))) break;
            const int flag = c->h_err[0];
            if (!flag) { c->last_scale = std::min(c->last_scale, c->ascale); break; }
            if (flag & RANGE_NONFINITE) { r = range_error(c, flag); break; }
            if (c->ascale <= ASCALE_MIN) { r = range_error(c, RANGE_OVERFLOW); break; }
            c->ascale = std::max(c->ascale * 0.25f, ASCALE_MIN);      // never steps past the smallest scale without trying it
        }
    }
    c->ascale = base;
    return r;
}

// =================================================================================================
extern "C" {

int mica_abi_version(void) { return 3; }

const char* mica_last_error(const mica_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int64_t mica_workspace_bytes(const mica_ctx* ctx) { return ctx ? ctx->bytes : 0; }

int mica_create(int device, int max_batch, int tile_size, mica_ctx** out) {
    return mica_create_dims(device, max_batch, tile_size, tile_size, tile_size, out);
}

int mica_create_dims(int device, int max_batch, int td, int th, int tw, mica_ctx** out) {
    // the workspace is sized for the cubic tile that bounds (td, th, tw): every size formula below is monotonic in each edge
    const int tile_size = std::max(td, std::max(th, tw));
    if (!out || max_batch < 1 || max_batch > 64 || std::min(td, std::min(th, tw)) < 4 || tile_size > 128) {
        g_create_err = "mica_create: bad argument (1 <= max_batch <= 64, tile edges in [4, 128])";
        return MICA_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || device < 0 || device >= ndev) {
        g_create_err = std::string("mica_create: no such HIP device (") + hipGetErrorString(e) + ")";
        return MICA_ERR_HIP;
    }
    hipDeviceProp_t prop;
    hipSetDevice(device);
    hipGetDeviceProperties(&prop, device);
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_err = std::string("mica_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return MICA_ERR_HIP;
    }
    mica_ctx* c = new mica_ctx();
    if (const char* ev = getenv("MICA_STEM_MFMA")) c->stem_mode = atoi(ev) != 0;
    if (const char* ev = getenv("MICA_RAW_CBLK")) c->raw_cblk = atoi(ev) == 32 ? 32 : 0;
    if (const char* ev = getenv("MICA_TRUNK_PER_RUN")) c->trunk_per_run = atoi(ev) != 0;
    if (const char* ev = getenv("MICA_F43")) { const int m = atoi(ev); c->f43_mode = m < 0 || m > 3 ? 3 : m; }      // A/B switch (0: the F(2,3) kernel everywhere)
    c->device = device;
    c->maxB = max_batch;
    c->S = tile_size;
    c->d = Dims{td, th, tw};
    c->V = td * th * tw;
    const int64_t BV = (int64_t)max_batch * c->V;
    int r = 0;
    // wino layout holds 4 transformed values per output pair: 2x the plain bytes (pairs = ceil(W/2) per row)
    const int64_t BVw = (int64_t)max_batch * tile_size * tile_size * ((tile_size + 1) / 2) * 4;
    auto S = [&](_Float16** p, int ch) { if (!r) r = dalloc(c, p, BV * ch * 2); };                        // plain: hi + lo
    auto W3 = [&](_Float16** p, int ch) { if (!r) r = dalloc(c, p, BVw * ch * 2); };                       // operand of 3^3 convs
    S(&c->S_exp, 128); W3(&c->S_af, 32); S(&c->S_fw, 64); W3(&c->S_x0, 64);
    W3(&c->S_1, 128); W3(&c->S_2, 128); W3(&c->S_f, 256);
    W3(&c->S_c[0], 128); W3(&c->S_c[1], 256); c->S_c[2] = nullptr;
    W3(&c->S_l, 64); W3(&c->S_fpn, 192); W3(&c->S_extra, 16); W3(&c->S_h1, 64);
    if (!r) r = dalloc(c, &c->extra_raw, BV * 8);
    auto R = [&](float** p, int ch) { if (!r) r = dalloc(c, p, BV * ch); };
    R(&c->R_a, 512); R(&c->R_b, 256); R(&c->R_c, 256);
    R(&c->logits[0], 4); R(&c->logits[1], 4); R(&c->logits[2], 21);
    int64_t wsn = stats_ws_floats(max_batch, 512);
    int64_t stem_ws = (int64_t)max_batch * 128 * (((tile_size + 31) / 32) * ((tile_size + 7) / 8) * ((tile_size + 1) / 2));
    if (stem_ws > wsn) wsn = stem_ws;
    if (fused_stats_ws_floats(max_batch, tile_size) > wsn) wsn = fused_stats_ws_floats(max_batch, tile_size);
    if (!r) r = dalloc(c, &c->ws, wsn);
    if (!r) r = dalloc(c, &c->ws_gap, (int64_t)max_batch * ((tile_size + 15) / 16) * ((tile_size + 7) / 8) * 256);
    auto Vv = [&](float** p) { if (!r) r = dalloc(c, p, (int64_t)max_batch * 512); };
    Vv(&c->v_mean); Vv(&c->v_rstd); Vv(&c->v_mean3); Vv(&c->v_rstd3); Vv(&c->v_pool); Vv(&c->v_gse); Vv(&c->v_gate); Vv(&c->v_abs);
    if (!r) r = dalloc(c, &c->d_err, max_batch);
    if (!r && hipHostMalloc((void**)&c->h_abs, sizeof(float) * max_batch) != hipSuccess) { c->err = "hipHostMalloc failed"; r = MICA_ERR_HIP; }
    if (!r && hipHostMalloc((void**)&c->h_err, sizeof(int) * max_batch) != hipSuccess) { c->err = "hipHostMalloc failed"; r = MICA_ERR_HIP; }
    if (r) {
        g_create_err = "mica_create: " + c->err;
        mica_destroy(c);
        return r;
    }
    *out = c;
    return MICA_OK;
}

void mica_destroy(mica_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (void* p : c->allocs) hipFree(p);
    if (c->h_abs) hipHostFree(c->h_abs);
    if (c->h_err) hipHostFree(c->h_err);
    for (hipEvent_t e : c->ev) hipEventDestroy(e);
    delete c;
}

int mica_load_weight(mica_ctx* c, const char* name, const float* h_data, const int64_t* shape, int ndim) {
    if (!c) return MICA_ERR_ARG;
    if (!name || !h_data || !shape || ndim < 1 || ndim > 5) { c->err = "mica_load_weight: bad argument"; return MICA_ERR_ARG; }
    if (c->finalized) { c->err = "mica_load_weight: weights already finalized"; return MICA_ERR_STATE; }
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    if (n < 1 || n > (int64_t)1 << 28) { c->err = "mica_load_weight: bad shape"; return MICA_ERR_ARG; }
    t.data.assign(h_data, h_data + n);
    for (float v : t.data)
        if (!std::isfinite(v)) { c->err = std::string("mica_load_weight: non-finite value in ") + name; return MICA_ERR_ARG; }
    std::string key(name);
    if (key.rfind("module.", 0) == 0) key = key.substr(7);   // DataParallel prefix (predict.py:237-238)
    c->host[key] = std::move(t);
    return MICA_OK;
}

int mica_set_conv_variant(mica_ctx* c, int mode) {
    if (!c) return MICA_ERR_ARG;
    if (mode < 0 || mode > 3) { c->err = "mica_set_conv_variant: mode must be 0, 1, 2 or 3"; return MICA_ERR_ARG; }
    if (c->finalized) { c->err = "mica_set_conv_variant: weights already finalized (the variant decides how they are packed)"; return MICA_ERR_STATE; }
    c->f43_mode = mode;
    return MICA_OK;
}

int mica_get_conv_variant(const mica_ctx* c) { return c ? c->f43_mode : MICA_ERR_ARG; }

int mica_finalize_weights(mica_ctx* c) {
    if (!c) return MICA_ERR_ARG;
    if (c->finalized) { c->err = "already finalized"; return MICA_ERR_STATE; }
    MICA_ENTER(c);
    int r;
    const std::string ip = "input_processing.";
    // stem: [32][1][k][k][k] x4 -> [conv][tap][32]
    {
        std::vector<float> w((size_t)stem_weight_floats()), b(128);
        size_t off = 0;
        const int ks[4] = {3, 5, 7, 9};
        for (int i = 0; i < 4; ++i) {
            const int k = ks[i], nt = k * k * k;
            const HostTensor* tw = find(c, ip + "exp_convs." + std::to_string(i) + ".weight");
            const HostTensor* tb = find(c, ip + "exp_convs." + std::to_string(i) + ".bias");
            if (!shape_is(tw, {32, 1, k, k, k}) || !shape_is(tb, {32})) { c->err = "stem conv weights missing or mis-shaped"; return MICA_ERR_STATE; }
            for (int t = 0; t < nt; ++t)
                for (int co = 0; co < 32; ++co) w[off + (size_t)t * 32 + co] = tw->data[(size_t)co * nt + t];
            for (int co = 0; co < 32; ++co) b[i * 32 + co] = tb->data[co];
            off += (size_t)nt * 32;
        }
        if ((r = upload(c, &c->stem_w, w))) return r;
        if ((r = upload(c, &c->stem_b, b))) return r;
        // the matrix-core form of the same weights: per (kernel size, residue class of x mod 8) K-step records, f16 hi + lo of w * 2^k
        std::vector<float> wf;
        std::vector<int> aoff;
        stem_mfma_plan(w.data(), wf, aoff, c->stem_plan);
        c->stem_wscale = pick_wscale(w, 1.f);
        float* d_wf = nullptr;
        if ((r = upload(c, &d_wf, wf))) return r;
        if ((r = dalloc(c, &c->stem_aoff, (int64_t)aoff.size()))) return r;
        HIPC(c, hipMemcpy(c->stem_aoff, aoff.data(), aoff.size() * sizeof(int), hipMemcpyHostToDevice));
        if ((r = dalloc(c, &c->stem_rec, (int64_t)c->stem_plan.records * 4 * 64 * 8))) return r;
        launch_stem_mfma_pack(d_wf, c->stem_plan.records, c->stem_wscale, c->stem_rec, nullptr);
        HIPC(c, hipDeviceSynchronize());
    }
    if ((r = setup_gate(c, c->exp_att, ip + "exp_attention.1", ip + "exp_attention.3", 128, 64, false))) return r;
    if ((r = setup_conv(c, c->downsizing, ip + "exp_downsizing", 64, 1, {128}, true))) return r;
    if ((r = setup_conv(c, c->feat_conv, ip + "feat_conv", 64, 3, {24}, false))) return r;
    if ((r = setup_conv(c, c->fusion0, ip + "fusion", 64, 1, {128, 64}, true))) return r;
    {
        const HostTensor *w0 = find(c, ip + "feat_gate.0.weight"), *b0 = find(c, ip + "feat_gate.0.bias");
        const HostTensor *w2 = find(c, ip + "feat_gate.2.weight"), *b2 = find(c, ip + "feat_gate.2.bias");
        if (!shape_is(w0, {16, 64, 1, 1, 1}) || !shape_is(b0, {16}) || !shape_is(w2, {1, 16, 1, 1, 1}) || !shape_is(b2, {1})) {
            c->err = "feat_gate weights missing or mis-shaped";
            return MICA_ERR_STATE;
        }
        if ((r = upload(c, &c->fg_w0, w0->data))) return r;
        if ((r = upload(c, &c->fg_b0, b0->data))) return r;
        if ((r = upload(c, &c->fg_w2, w2->data))) return r;
        if ((r = upload(c, &c->fg_b2, b2->data))) return r;
    }
    for (int e = 0; e < 3; ++e) {
        Enc& E = c->enc[e];
        const int C = 64 << e;
        E.C = C;
        const std::string p = "encoder." + std::to_string(e) + ".";
        // the three dense-block convs of an encoder switch together (they share operands): encoder.2 has Cout = 128, 128, 256; a
        // transition conv is the only reader of its operand (the 1x1 fusion's output), so encoder.1's (128 -> 256) can switch alone
        const bool f43 = c->f43_mode != 0 && e == 2, f43t = f43 || (c->f43_mode == 2 && e == 1);
        if ((r = setup_conv(c, E.conv1, p + "dense_block.conv1.0", C / 2, 3, {C}, false, 1.f, f43))) return r;
        if ((r = setup_conv(c, E.conv2, p + "dense_block.conv2.0", C / 2, 3, {C, C / 2}, false, 1.f, f43))) return r;
        if ((r = setup_conv(c, E.conv3, p + "dense_block.conv3.0", C, 3, {C, C / 2, C / 2}, false, 1.f, f43))) return r;
        if ((r = setup_gate(c, E.se, p + "dense_block.se.fc.0", p + "dense_block.se.fc.3", C, C / 16, true))) return r;
        if ((r = setup_gate(c, E.ga, p + "dual_attn.global_attn.1", p + "dual_attn.global_attn.4", C, C / 4, false))) return r;
        if ((r = setup_conv(c, E.fusion, p + "dual_attn.fusion", C, 1, {C, C}, true))) return r;
        if ((r = setup_conv(c, E.transition, p + "transition.0", 2 * C, 3, {C}, false, 1.f, f43t))) return r;
        if (E.conv1.f43 != E.conv2.f43 || E.conv1.f43 != E.conv3.f43) { c->err = "internal: mixed conv variants in a dense block"; return MICA_ERR_STATE; }
        const HostTensor *dw = find(c, p + "dual_attn.local_attn.0.weight"), *db = find(c, p + "dual_attn.local_attn.0.bias");
        if (!shape_is(dw, {C, 1, 3, 3, 3}) || !shape_is(db, {C})) { c->err = "depthwise weights missing or mis-shaped"; return MICA_ERR_STATE; }
        std::vector<float> wt((size_t)27 * C);
        for (int ch = 0; ch < C; ++ch)
            for (int t = 0; t < 27; ++t) wt[(size_t)t * C + ch] = dw->data[(size_t)ch * 27 + t];
        if ((r = upload(c, &E.dw_w, wt))) return r;
        if ((r = upload(c, &E.dw_b, db->data))) return r;
    }
    {
        const HostTensor* fw = find(c, "fpn.weights");
        if (!shape_is(fw, {3})) { c->err = "fpn.weights missing or mis-shaped"; return MICA_ERR_STATE; }
        // softmax over the three fusion weights (model.py:183), f32 like torch
        float m = std::fmax(fw->data[0], std::fmax(fw->data[1], fw->data[2]));
        float ex[3], s = 0.f;
        for (int i = 0; i < 3; ++i) { ex[i] = std::exp(fw->data[i] - m); s += ex[i]; }
        for (int i = 0; i < 3; ++i) {
            if ((r = setup_conv(c, c->lateral[i], "fpn.lateral." + std::to_string(i), 64, 1, {128 << i}, false))) return r;
            if ((r = setup_conv(c, c->smooth[i], "fpn.smooth." + std::to_string(i) + ".0", 64, 3, {64}, false, ex[i] / s, c->f43_mode == 3))) return r;
        }
    }
    const char* hn[3] = {"backbone_head", "ca_head", "aa_head"};
    const int ncls[3] = {4, 4, 21};
    for (int h = 0; h < 3; ++h) {
        Head& H = c->heads[h];
        H.ncls = ncls[h];
        const std::string p = std::string(hn[h]) + ".";
        std::vector<int> seg = {192};
        if (h > 0) seg.push_back(4 * h);
        // (mode 3) the three conv1 read the same operands (the FPN output, the earlier heads' logits): they switch together
        if ((r = setup_conv(c, H.conv1, p + "conv1", 64, 3, seg, false, 1.f, c->f43_mode == 3))) return r;
        if ((r = setup_conv(c, H.conv2, p + "conv2", 32, 3, {64}, false))) return r;
        if ((r = setup_gate(c, H.cal, p + "calibration.1", p + "calibration.4", 32, 8, false))) return r;
        const HostTensor *wf = find(c, p + "final.weight"), *bf = find(c, p + "final.bias");
        if (!shape_is(wf, {ncls[h], 32, 1, 1, 1}) || !shape_is(bf, {ncls[h]})) { c->err = "head final weights missing or mis-shaped"; return MICA_ERR_STATE; }
        if ((r = upload(c, &H.wf, wf->data))) return r;
        if ((r = upload(c, &H.bf, bf->data))) return r;
    }
    HIPC(c, hipDeviceSynchronize());
    CHECK_LAUNCHES(c);
    c->host.clear();
    c->finalized = true;
    return MICA_OK;
}

int mica_forward_logits(mica_ctx* c, const float* d_map, const float* d_af, int batch, int af_mode, float* d_bb, float* d_ca,
                        float* d_aa, void* stream) {
    if (!c) return MICA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    return forward_checked(c, d_map, d_af, batch, af_mode, d_bb, d_ca, d_aa, st);
}

int mica_postprocess(mica_ctx* c, const float* d_bb, const float* d_ca, const float* d_aa, int batch, float* d_bb_prob,
                     float* d_ca_prob, float* d_aa_prob, float* d_aa_pred, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_bb || !d_ca || !d_aa || !d_bb_prob || !d_ca_prob || !d_aa_prob || !d_aa_pred || batch < 1) { c->err = "mica_postprocess: bad argument"; return MICA_ERR_ARG; }
    MICA_ENTER(c);
    launch_postprocess(d_bb, d_ca, d_aa, batch, c->V, d_bb_prob, d_ca_prob, d_aa_prob, d_aa_pred, c->V, (int64_t)20 * c->V, (hipStream_t)stream);
    CHECK_LAUNCHES(c);
    return MICA_OK;
}

int mica_forward_tiles(mica_ctx* c, const float* d_map, const float* d_af, int batch, int af_mode, float* d_bb_prob,
                       float* d_ca_prob, float* d_aa_prob, float* d_aa_pred, void* stream) {
    if (!c) return MICA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int r = forward_checked(c, d_map, d_af, batch, af_mode, c->logits[0], c->logits[1], c->logits[2], st);
    if (r) return r;
    return mica_postprocess(c, c->logits[0], c->logits[1], c->logits[2], batch, d_bb_prob, d_ca_prob, d_aa_prob, d_aa_pred, stream);
}

int mica_af_abs_sums(mica_ctx* c, const float* d_af, int64_t batch, float* h_sums, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_af || !h_sums || batch < 1) { c->err = "mica_af_abs_sums: bad argument"; return MICA_ERR_ARG; }
    MICA_ENTER(c);
    hipStream_t st = (hipStream_t)stream;
    const int64_t per = (int64_t)24 * c->V;
    for (int64_t b0 = 0; b0 < batch; b0 += c->maxB) {           // the reduction of forward_impl, max_batch tiles at a time
        const int n = (int)std::min<int64_t>(c->maxB, batch - b0);
        HIPC(c, hipMemsetAsync(c->v_abs, 0, sizeof(float) * n, st));
        launch_abs_sum(d_af + b0 * per, n, per, c->v_abs, st);
        HIPC(c, hipMemcpyAsync(c->h_abs, c->v_abs, sizeof(float) * n, hipMemcpyDeviceToHost, st));
        CHECK_LAUNCHES(c);
        HIPC(c, hipStreamSynchronize(st));
        for (int b = 0; b < n; ++b) h_sums[b0 + b] = c->h_abs[b];
    }
    return MICA_OK;
}

int mica_forward_records(mica_ctx* c, const float* d_map, const float* d_af, int batch, int af_mode, float* d_rec, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_rec) { c->err = "mica_forward_records: null pointer argument"; return MICA_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    int r = forward_checked(c, d_map, d_af, batch, af_mode, c->logits[0], c->logits[1], c->logits[2], st);
    if (r) return r;
    const int64_t V = c->V;
    launch_postprocess(c->logits[0], c->logits[1], c->logits[2], batch, c->V, d_rec, d_rec + V, d_rec + 3 * V, d_rec + 2 * V, 23 * V, 23 * V, st);
    CHECK_LAUNCHES(c);
    return MICA_OK;
}

int64_t mica_tile_count(int64_t n0, int64_t n1, int64_t n2, int grid) {
    if (n0 < 1 || n1 < 1 || n2 < 1 || grid < 1) return MICA_ERR_ARG;
    return ((n0 + grid - 1) / grid) * ((n1 + grid - 1) / grid) * ((n2 + grid - 1) / grid);
}

int64_t mica_tile_table(int64_t n0, int64_t n1, int64_t n2, int grid, int64_t* h_table, int64_t capacity) {
    int64_t T = mica_tile_count(n0, n1, n2, grid);
    if (T < 0 || !h_table || capacity < T) return MICA_ERR_ARG;
    int64_t t = 0;
    for (int64_t i = 0; i < n0; i += grid)           // create_grids.py:143-149
        for (int64_t j = 0; j < n1; j += grid)
            for (int64_t k = 0; k < n2; k += grid) {
                int64_t* r = h_table + 6 * t++;
                r[0] = i; r[1] = j; r[2] = k;
                r[3] = (n0 - i < grid) ? n0 - i : grid;
                r[4] = (n1 - j < grid) ? n1 - j : grid;
                r[5] = (n2 - k < grid) ? n2 - k : grid;
            }
    return T;
}

static int check_tiling(mica_ctx* c, const void* a, const void* b, int channels, int64_t n0, int64_t n1, int64_t n2, int grid,
                        int pad, int64_t first, int64_t count) {
    if (!a || !b || channels < 1 || channels > 65535 || grid < 1 || pad < 0 || count < 0 || count > 65535 || first < 0) {
        c->err = "tiling: bad argument";
        return MICA_ERR_ARG;
    }
    int64_t T = mica_tile_count(n0, n1, n2, grid);
    if (T < 0 || first + count > T) { c->err = "tiling: tile range outside the table"; return MICA_ERR_ARG; }
    return MICA_OK;
}

int mica_gather_tiles(mica_ctx* c, const float* d_vol, int channels, int64_t n0, int64_t n1, int64_t n2, int grid, int pad,
                      int64_t first, int64_t count, float* d_tiles, void* stream) {
    if (!c) return MICA_ERR_ARG;
    int r = check_tiling(c, d_vol, d_tiles, channels, n0, n1, n2, grid, pad, first, count);
    if (r || count == 0) return r;
    MICA_ENTER(c);
    launch_gather_tiles(d_vol, channels, n0, n1, n2, grid, pad, first, count, d_tiles, (hipStream_t)stream);
    CHECK_LAUNCHES(c);
    return MICA_OK;
}

int mica_gather_tiles_u8(mica_ctx* c, const uint8_t* d_vol, int channels, int64_t n0, int64_t n1, int64_t n2, int grid, int pad,
                         int64_t first, int64_t count, float* d_tiles, void* stream) {
    if (!c) return MICA_ERR_ARG;
    int r = check_tiling(c, d_vol, d_tiles, channels, n0, n1, n2, grid, pad, first, count);
    if (r || count == 0) return r;
    MICA_ENTER(c);
    launch_gather_tiles_u8(d_vol, channels, n0, n1, n2, grid, pad, first, count, d_tiles, (hipStream_t)stream);
    CHECK_LAUNCHES(c);
    return MICA_OK;
}

int mica_stitch_tiles(mica_ctx* c, const float* d_tiles, int channels, int64_t n0, int64_t n1, int64_t n2, int grid, int pad,
                      int64_t first, int64_t count, float* d_vol, void* stream) {
    if (!c) return MICA_ERR_ARG;
    int r = check_tiling(c, d_tiles, d_vol, channels, n0, n1, n2, grid, pad, first, count);
    if (r || count == 0) return r;
    MICA_ENTER(c);
    launch_stitch_tiles(d_tiles, channels, n0, n1, n2, grid, pad, first, count, d_vol, (hipStream_t)stream);
    CHECK_LAUNCHES(c);
    return MICA_OK;
}

int mica_normalise_map_typed(mica_ctx* c, float* d_vol, int64_t n, int map_type, double* h_stats, void* stream) {
    return mica_normalise_map_np(c, d_vol, n, map_type, MICA_NUMPY_NEP50, h_stats, stream);
}

int mica_normalise_map_np(mica_ctx* c, float* d_vol, int64_t n, int map_type, int numpy_rules, double* h_stats, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_vol || n < 1 || !h_stats || map_type < MICA_MAP_F32 || map_type > MICA_MAP_U16 ||
        (numpy_rules != MICA_NUMPY_NEP50 && numpy_rules != MICA_NUMPY_LEGACY)) {
        c->err = "mica_normalise_map: bad argument";
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = normalise_map_device(d_vol, n, map_type, numpy_rules, h_stats, (hipStream_t)stream, buf, sizeof(buf));
    if (r) c->err = buf;
    return r;
}

int mica_normalise_map(mica_ctx* c, float* d_vol, int64_t n, double* h_stats, void* stream) {
    return mica_normalise_map_typed(c, d_vol, n, MICA_MAP_F32, h_stats, stream);
}

int mica_zoom_cubic_typed(mica_ctx* c, const float* d_in, int64_t n0, int64_t n1, int64_t n2, int64_t o0, int64_t o1, int64_t o2,
                          int map_type, float* d_out, void* stream) {
    if (!c) return MICA_ERR_ARG;
    const int64_t lim = 4096;
    if (!d_in || !d_out || n0 < 1 || n1 < 1 || n2 < 1 || o0 < 1 || o1 < 1 || o2 < 1 || n0 > lim || n1 > lim || n2 > lim || o0 > lim ||
        o1 > lim || o2 > lim || map_type < MICA_MAP_F32 || map_type > MICA_MAP_U16) {
        c->err = "mica_zoom_cubic: bad argument";
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = zoom_cubic_device(d_in, n0, n1, n2, o0, o1, o2, map_type, d_out, (hipStream_t)stream, buf, sizeof(buf));
    if (r) c->err = buf;
    return r;
}

int mica_zoom_cubic(mica_ctx* c, const float* d_in, int64_t n0, int64_t n1, int64_t n2, int64_t o0, int64_t o1, int64_t o2,
                    float* d_out, void* stream) {
    return mica_zoom_cubic_typed(c, d_in, n0, n1, n2, o0, o1, o2, MICA_MAP_F32, d_out, stream);
}

int mica_rasterise_atoms(mica_ctx* c, const float* d_xyz, const int32_t* d_bb, const int32_t* d_aa, int64_t n_atoms,
                         const float* h_origin, int64_t nz, int64_t ny, int64_t nx, float* d_vol, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (n_atoms < 0 || (n_atoms > 0 && (!d_xyz || !d_bb || !d_aa)) || !h_origin || !d_vol || nz < 1 || ny < 1 || nx < 1 || nz > 4096 ||
        ny > 4096 || nx > 4096) {
        c->err = "mica_rasterise_atoms: bad argument";
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = rasterise_atoms_device(d_xyz, d_bb, d_aa, n_atoms, h_origin, nz, ny, nx, d_vol, (hipStream_t)stream, buf, sizeof(buf));
    if (r) c->err = buf;
    return r;
}

int mica_threshold_points(mica_ctx* c, const float* d_vol, int64_t n, float thr, int64_t* d_idx, int64_t capacity, int64_t* h_count,
                          void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_vol || n < 1 || n > ((int64_t)1 << 36) || capacity < 0 || (capacity > 0 && !d_idx) || !h_count) {
        c->err = "mica_threshold_points: bad argument";
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = threshold_points_device(d_vol, n, thr, d_idx, capacity, h_count, (hipStream_t)stream, buf, sizeof(buf));
    if (r) c->err = buf;
    return r;
}

int mica_gather_values(mica_ctx* c, const float* d_vol, int channels, int64_t nvox, const int64_t* d_idx, int64_t n, float* d_out,
                       void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_vol || channels < 1 || nvox < 1 || n < 0 || (n > 0 && (!d_idx || !d_out))) {
        c->err = "mica_gather_values: bad argument";
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = gather_values_device(d_vol, channels, nvox, d_idx, n, d_out, (hipStream_t)stream, buf, sizeof(buf));
    if (r) c->err = buf;
    return r;
}

int mica_refine_candidates(mica_ctx* c, const float* d_ca, const float* d_aa, int64_t n0, int64_t n1, int64_t n2, const int32_t* d_cand,
                           int64_t n, double* d_coord, float* d_aa_out, int32_t* d_ok, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_ca || !d_aa || n0 < 1 || n1 < 1 || n2 < 1 || n0 > 4096 || n1 > 4096 || n2 > 4096 || n < 0 ||
        (n > 0 && (!d_cand || !d_coord || !d_aa_out || !d_ok))) {
        c->err = "mica_refine_candidates: bad argument";
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = refine_candidates_device(d_ca, d_aa, (int)n0, (int)n1, (int)n2, d_cand, n, d_coord, d_aa_out, d_ok, (hipStream_t)stream, buf,
                                     sizeof(buf));
    if (r) c->err = buf;
    return r;
}

int mica_segment_sums(mica_ctx* c, const float* d_vals, const int64_t* d_seg_off, int64_t nseg, float* d_sums, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (nseg < 0 || (nseg > 0 && (!d_vals || !d_seg_off || !d_sums))) { c->err = "mica_segment_sums: bad argument"; return MICA_ERR_ARG; }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = segment_sums_device(d_vals, d_seg_off, nseg, d_sums, (hipStream_t)stream, buf, sizeof(buf));
    if (r) c->err = buf;
    return r;
}

int mica_nms_points(mica_ctx* c, const int32_t* d_pts, int64_t n, int64_t n0, int64_t n1, int64_t n2, double radius, int32_t* d_keep,
                    void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (n < 0 || n > 0x7fffffff || n0 < 1 || n1 < 1 || n2 < 1 || n0 > 4096 || n1 > 4096 || n2 > 4096 || !(radius >= 0.0 && radius < 1e6) ||
        (n > 0 && (!d_pts || !d_keep))) {
        c->err = "mica_nms_points: bad argument";
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = nms_points_device(d_pts, n, (int)n0, (int)n1, (int)n2, radius, d_keep, (hipStream_t)stream, buf, sizeof(buf));
    if (r) c->err = buf;
    return r;
}

int mica_neighbour_matrix(mica_ctx* c, const double* d_cands, int64_t n, const float* d_bb, int64_t n0, int64_t n1, int64_t n2,
                          double* d_dis, double* d_mat, void* stream) {
    return mica_neighbour_matrix_np(c, d_cands, n, d_bb, n0, n1, n2, MICA_NUMPY_NEP50, d_dis, d_mat, stream);
}

int mica_neighbour_matrix_np(mica_ctx* c, const double* d_cands, int64_t n, const float* d_bb, int64_t n0, int64_t n1, int64_t n2,
                             int numpy_rules, double* d_dis, double* d_mat, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if ((numpy_rules != MICA_NUMPY_NEP50 && numpy_rules != MICA_NUMPY_LEGACY) || n < 0 || n > 65535 || n0 < 1 || n1 < 1 || n2 < 1 || n0 > 4096 || n1 > 4096 || n2 > 4096 || (n > 0 && (!d_cands || !d_bb || !d_dis || !d_mat))) {
        c->err = "mica_neighbour_matrix: bad argument";
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    char buf[256] = {0};
    int r = neighbour_matrix_device(d_cands, n, d_bb, (int)n0, (int)n1, (int)n2, d_dis, d_mat, numpy_rules, (hipStream_t)stream, buf, sizeof(buf));
    if (r) c->err = buf;
    return r;
}

// ---- single-op entry points (test harness for the individual kernels) ---------------------------
int mica_op_conv3d(mica_ctx* c, const float* d_x, int batch, int cin, int d, int h, int w, const float* h_w, const float* h_b,
                   int cout, int k, float* d_y, void* stream) {
    return mica_op_conv3d_variant(c, d_x, batch, cin, d, h, w, h_w, h_b, cout, k, 0, d_y, stream);
}

int mica_op_conv3d_variant(mica_ctx* c, const float* d_x, int batch, int cin, int d, int h, int w, const float* h_w, const float* h_b,
                           int cout, int k, int variant, float* d_y, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_x || !h_w || !h_b || !d_y || batch < 1 || cin < 1 || cin > 1024 || cout < 32 || cout > 1024 || cout % 32 || (k != 1 && k != 3) || !op_box_ok(batch, d, h, w) ||
        (k == 1 && cout != 64 && cout != 128 && cout != 256) || variant < 0 || variant > 1 || (variant == 1 && (k != 3 || (cout % 128 && cout != 64)))) {
        c->err = "mica_op_conv3d: bad argument (k = 3: cout a multiple of 32 - of 128, or 64, for the F(4,3) variant; k = 1: cout in {64, 128, 256}; cin <= 1024)" OP_BOX_MSG;
        return MICA_ERR_ARG;
    }
    const bool f43 = variant == 1;
    MICA_ENTER(c);
    hipStream_t st = (hipStream_t)stream;
    const int V = d * h * w, cp = pad16(cin), nt = k * k * k;
    const bool wino = (k == 3);
    const int64_t Vop = f43 ? (int64_t)d * h * ((w + 3) / 4) * 6 : wino ? (int64_t)d * h * ((w + 1) / 2) * 4 : V;
    Tmp t;
    _Float16* sx = t.get<_Float16>((int64_t)batch * Vop * cp * 2);
    float* dw = t.get<float>((int64_t)cout * cin * nt);
    float* db = t.get<float>(cout);
    float* raw = t.get<float>((int64_t)batch * V * cout);
    _Float16* pk = t.get<_Float16>(f43 ? packed_weight_halves_wino43(cout, cp / 16)
                                       : wino ? packed_weight_halves_wino(cout, cp / 16) : packed_weight_halves(cout, k, cp / 16));
    int* derr = t.get<int>(batch);
    if (!sx || !dw || !db || !raw || !pk || !derr) { c->err = "mica_op_conv3d: hipMalloc failed"; return MICA_ERR_HIP; }
    std::vector<float> hw(h_w, h_w + (size_t)cout * cin * nt);
    float ws = pick_wscale(hw, 1.f);
    HIPC(c, hipMemcpyAsync(dw, h_w, sizeof(float) * cout * cin * nt, hipMemcpyHostToDevice, st));
    HIPC(c, hipMemcpyAsync(db, h_b, sizeof(float) * cout, hipMemcpyHostToDevice, st));
    HIPC(c, hipMemsetAsync(derr, 0, sizeof(int) * batch, st));
    int sc[1] = {cin}, scp[1] = {cp};
    ConvSrcs s{};
    s.n = 1; s.p[0] = sx; s.chunks_total[0] = cp / 16; s.chunk_off[0] = 0; s.chunks[0] = cp / 16;
    if (f43) {
        const float asc = ASCALE_DEFAULT / WINO43_ASCALE_DIV;
        launch_prep_ncdhw_wino43(d_x, batch, Dims{d, h, w}, cin, SplitView{sx, cp / 16, 0, cp / 16}, SplitEnc{derr, asc}, st);
        launch_pack_weights_wino43(dw, cout, cin, sc, scp, 1, nullptr, 1, 1.f, ws, pk, st);
        launch_conv_wino43(s, pk, 0, db, 1.0f / (ws * asc), raw, batch, Dims{d, h, w}, cout, nullptr, st);
    } else if (wino) {
        launch_prep_ncdhw_wino(d_x, batch, Dims{d, h, w}, cin, SplitView{sx, cp / 16, 0, cp / 16}, SplitEnc{derr, ASCALE_DEFAULT}, st);
        launch_pack_weights_wino(dw, cout, cin, sc, scp, 1, nullptr, 1, 1.f, ws, pk, st);
        launch_conv_wino(s, pk, 0, db, 1.0f / (ws * ASCALE_DEFAULT), raw, batch, Dims{d, h, w}, cout, nullptr, st);
    } else {
        launch_prep_ncdhw(d_x, batch, V, cin, SplitView{sx, cp / 16, 0, cp / 16}, nullptr, SplitEnc{derr, ASCALE_DEFAULT}, st);
        launch_pack_weights(dw, cout, cin, k, sc, scp, 1, nullptr, 1, 1.f, ws, pk, st);
        Conv1Srcs s1{};
        s1.n = 1;
        s1.s[0] = Conv1Src{sx, nullptr, nullptr, 0, cp / 16, cp / 16, 0, 0};
        launch_conv1x1(s1, pk, 0, db, 1.0f / (ws * ASCALE_DEFAULT), raw, SplitView{nullptr, 0, 0, 0}, batch, Dims{d, h, w}, cout,
                       SplitEnc{derr, ASCALE_DEFAULT}, st);
    }
    launch_nhwc_to_nchw(raw, batch, cout, V, d_y, st);
    CHECK_LAUNCHES(c);
    HIPC(c, hipStreamSynchronize(st));
    return MICA_OK;
}

int mica_op_norm_conv1_conv3(mica_ctx* c, const float* d_x, int batch, int cin, int d, int h, int w, const float* h_w1, const float* h_b1,
                             int cmid, const float* h_w3, const float* h_b3, int cout, float* d_y, void* stream) {
    return mica_op_norm_conv1_conv3_variant(c, d_x, batch, cin, d, h, w, h_w1, h_b1, cmid, h_w3, h_b3, cout, 0, d_y, stream);
}

int mica_op_norm_conv1_conv3_variant(mica_ctx* c, const float* d_x, int batch, int cin, int d, int h, int w, const float* h_w1, const float* h_b1,
                                     int cmid, const float* h_w3, const float* h_b3, int cout, int variant, float* d_y, void* stream) {
    if (!c) return MICA_ERR_ARG;
    const bool f43 = variant == 1;
    if (!d_x || !h_w1 || !h_b1 || !h_w3 || !h_b3 || !d_y || batch < 1 || cin < 16 || !pow2_8_512(cin) || (cmid != 64 && cmid != 128 && cmid != 256) ||
        cout < 32 || cout > 1024 || cout % 32 || !op_box_ok(batch, d, h, w) || variant < 0 || variant > 1 || (f43 && cout % 128 && cout != 64)) {
        c->err = "mica_op_norm_conv1_conv3: bad argument (cin a power of two in [16,512], cmid in {64,128,256}, cout a multiple of 32, <= 1024)" OP_BOX_MSG;
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    hipStream_t st = (hipStream_t)stream;
    const Dims dm{d, h, w};
    const int V = d * h * w;
    const int64_t Vw = f43 ? (int64_t)d * h * ((w + 3) / 4) * 6 : (int64_t)d * h * ((w + 1) / 2) * 4;
    Tmp t;
    float* xr = t.get<float>((int64_t)batch * V * cin);              // raw NDHWC input
    float* mean = t.get<float>((int64_t)batch * cin);
    float* rstd = t.get<float>((int64_t)batch * cin);
    float* ws = t.get<float>(stats_ws_floats(batch, cin));
    float* mid = t.get<float>((int64_t)batch * V * cmid);            // only when the operand cannot be emitted directly
    _Float16* op = t.get<_Float16>((int64_t)batch * Vw * cmid * 2);
    float* raw = t.get<float>((int64_t)batch * V * cout);
    float* dw1 = t.get<float>((int64_t)cmid * cin);
    float* db1 = t.get<float>(cmid);
    float* dw3 = t.get<float>((int64_t)cout * cmid * 27);
    float* db3 = t.get<float>(cout);
    _Float16* pk1 = t.get<_Float16>(packed_weight_halves(cmid, 1, cin / 16));
    _Float16* pk3 = t.get<_Float16>(f43 ? packed_weight_halves_wino43(cout, cmid / 16) : packed_weight_halves_wino(cout, cmid / 16));
    int* derr = t.get<int>(batch);
    if (!xr || !mean || !rstd || !ws || !mid || !op || !raw || !dw1 || !db1 || !dw3 || !db3 || !pk1 || !pk3 || !derr) { c->err = "hipMalloc failed"; return MICA_ERR_HIP; }
    std::vector<float> v1(h_w1, h_w1 + (size_t)cmid * cin), v3(h_w3, h_w3 + (size_t)cout * cmid * 27);
    const float s1 = pick_wscale(v1, 1.f), s3 = pick_wscale(v3, 1.f);
    HIPC(c, hipMemcpyAsync(dw1, h_w1, sizeof(float) * cmid * cin, hipMemcpyHostToDevice, st));
    HIPC(c, hipMemcpyAsync(db1, h_b1, sizeof(float) * cmid, hipMemcpyHostToDevice, st));
    HIPC(c, hipMemcpyAsync(dw3, h_w3, sizeof(float) * cout * cmid * 27, hipMemcpyHostToDevice, st));
    HIPC(c, hipMemcpyAsync(db3, h_b3, sizeof(float) * cout, hipMemcpyHostToDevice, st));
    HIPC(c, hipMemsetAsync(derr, 0, sizeof(int) * batch, st));
    const SplitEnc enc{derr, ASCALE_DEFAULT};
    launch_nchw_to_nhwc(d_x, batch, cin, V, xr, st);
    launch_stats(xr, batch, V, cin, 1e-5f, mean, rstd, ws, st);
    int sc1[1] = {cin}, sc3[1] = {cmid};
    launch_pack_weights(dw1, cmid, cin, 1, sc1, sc1, 1, nullptr, 1, 1.f, s1, pk1, st);
    if (f43) launch_pack_weights_wino43(dw3, cout, cmid, sc3, sc3, 1, nullptr, 1, 1.f, s3, pk3, st);
    else launch_pack_weights_wino(dw3, cout, cmid, sc3, sc3, 1, nullptr, 1, 1.f, s3, pk3, st);
    Conv1Srcs s{};
    s.n = 1;
    s.s[0] = Conv1Src{xr, mean, rstd, 1, cin / 16, cin / 16, 0, 1};
    const SplitView opv{op, cmid / 16, 0, cmid / 16};
    const float asc3 = f43 ? ASCALE_DEFAULT / WINO43_ASCALE_DIV : ASCALE_DEFAULT;
    if (conv1x1_can_emit_wino(dm, f43 ? 2 : 1)) {
        launch_conv1x1(s, pk1, 0, db1, 1.0f / (s1 * ASCALE_DEFAULT), nullptr, opv, batch, dm, cmid, enc, st, f43 ? 2 : 1, asc3);
    } else {
        launch_conv1x1(s, pk1, 0, db1, 1.0f / (s1 * ASCALE_DEFAULT), mid, SplitView{nullptr, 0, 0, 0}, batch, dm, cmid, enc, st);
        if (f43) launch_prep_wino43(mid, batch, dm, cmid, nullptr, nullptr, 0, opv, SplitEnc{derr, asc3}, st);
        else launch_prep_wino(mid, batch, dm, cmid, nullptr, nullptr, 0, nullptr, opv, SplitView{nullptr, 0, 0, 0}, nullptr, ws, enc, st);
    }
    ConvSrcs cs{};
    cs.n = 1; cs.p[0] = op; cs.chunks_total[0] = cmid / 16; cs.chunk_off[0] = 0; cs.chunks[0] = cmid / 16;
    if (f43) launch_conv_wino43(cs, pk3, 0, db3, 1.0f / (s3 * asc3), raw, batch, dm, cout, nullptr, st);
    else launch_conv_wino(cs, pk3, 0, db3, 1.0f / (s3 * ASCALE_DEFAULT), raw, batch, dm, cout, nullptr, st);
    launch_nhwc_to_nchw(raw, batch, cout, V, d_y, st);
    CHECK_LAUNCHES(c);
    HIPC(c, hipStreamSynchronize(st));
    return MICA_OK;
}

int mica_op_instnorm_relu(mica_ctx* c, const float* d_x, int batch, int ch, int d, int h, int w, float* d_y, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_x || !d_y || batch < 1 || !pow2_8_512(ch) || !op_box_ok(batch, d, h, w)) { c->err = "mica_op_instnorm_relu: bad argument (C must be a power of two in [8,512])" OP_BOX_MSG; return MICA_ERR_ARG; }
    MICA_ENTER(c);
    hipStream_t st = (hipStream_t)stream;
    const int V = d * h * w;
    Tmp t;
    float* a = t.get<float>((int64_t)batch * V * ch);
    float* b = t.get<float>((int64_t)batch * V * ch);
    float* mean = t.get<float>((int64_t)batch * ch);
    float* rstd = t.get<float>((int64_t)batch * ch);
    float* ws = t.get<float>(stats_ws_floats(batch, ch));
    int* derr = t.get<int>(batch);
    if (!a || !b || !mean || !rstd || !ws || !derr) { c->err = "hipMalloc failed"; return MICA_ERR_HIP; }
    launch_nchw_to_nhwc(d_x, batch, ch, V, a, st);
    launch_stats(a, batch, V, ch, 1e-5f, mean, rstd, ws, st);
    launch_prep(a, batch, V, ch, mean, rstd, 1, nullptr, SplitView{nullptr, 0, 0, 0}, b, nullptr, ws, SplitEnc{derr, ASCALE_DEFAULT}, st);
    launch_nhwc_to_nchw(b, batch, ch, V, d_y, st);
    CHECK_LAUNCHES(c);
    HIPC(c, hipStreamSynchronize(st));
    return MICA_OK;
}

int mica_op_depthwise3(mica_ctx* c, const float* d_x, int batch, int ch, int d, int h, int w, const float* h_w, const float* h_b,
                       float* d_y, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_x || !d_y || !h_w || !h_b || batch < 1 || ch < 16 || ch > 1024 || ch % 16 || !op_box_ok(batch, d, h, w)) {
        c->err = "mica_op_depthwise3: bad argument (C must be a multiple of 16, <= 1024)" OP_BOX_MSG;
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    hipStream_t st = (hipStream_t)stream;
    const int V = d * h * w;
    Tmp t;
    float* a = t.get<float>((int64_t)batch * V * ch);
    float* b = t.get<float>((int64_t)batch * V * ch);
    float* dw = t.get<float>(27 * ch);
    float* db = t.get<float>(ch);
    if (!a || !b || !dw || !db) { c->err = "hipMalloc failed"; return MICA_ERR_HIP; }
    std::vector<float> wt((size_t)27 * ch);
    for (int cc = 0; cc < ch; ++cc)
        for (int tp = 0; tp < 27; ++tp) wt[(size_t)tp * ch + cc] = h_w[(size_t)cc * 27 + tp];
    HIPC(c, hipMemcpy(dw, wt.data(), sizeof(float) * 27 * ch, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(db, h_b, sizeof(float) * ch, hipMemcpyHostToDevice));
    launch_nchw_to_nhwc(d_x, batch, ch, V, a, st);
    launch_depthwise(a, batch, Dims{d, h, w}, ch, nullptr, nullptr, nullptr, dw, db, b, nullptr, nullptr, st);
    launch_nhwc_to_nchw(b, batch, ch, V, d_y, st);
    CHECK_LAUNCHES(c);
    HIPC(c, hipStreamSynchronize(st));
    return MICA_OK;
}

int mica_op_se_depthwise(mica_ctx* c, const float* d_x, int batch, int ch, int d, int h, int w, const float* h_dw_w, const float* h_dw_b,
                         const float* h_fc0_w, const float* h_fc0_b, const float* h_fc3_w, const float* h_fc3_b, float* d_y, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!d_x || !d_y || !h_dw_w || !h_dw_b || !h_fc0_w || !h_fc0_b || !h_fc3_w || !h_fc3_b || batch < 1 || ch < 32 || ch > 256 || !pow2_8_512(ch) ||
        !op_box_ok(batch, d, h, w)) {
        c->err = "mica_op_se_depthwise: bad argument (C a power of two in [32, 256])" OP_BOX_MSG;
        return MICA_ERR_ARG;
    }
    MICA_ENTER(c);
    hipStream_t st = (hipStream_t)stream;
    const Dims dm{d, h, w};
    const int V = d * h * w, Ch = ch / 16;
    const int nblk = ((w + 15) / 16) * ((h + 7) / 8);                 // upper bound of the depthwise kernel's blocks per channel slab
    Tmp t;
    float* a = t.get<float>((int64_t)batch * V * ch);
    float* u = t.get<float>((int64_t)batch * V * ch);
    float* y = t.get<float>((int64_t)batch * V * ch);
    float* dw = t.get<float>(27 * ch);
    float* db = t.get<float>(ch);
    float *w1 = t.get<float>((int64_t)Ch * ch), *b1 = t.get<float>(Ch), *w2 = t.get<float>((int64_t)ch * Ch), *b2 = t.get<float>(ch);
    float *m0 = t.get<float>((int64_t)batch * ch), *r0 = t.get<float>((int64_t)batch * ch), *m1 = t.get<float>((int64_t)batch * ch), *r1 = t.get<float>((int64_t)batch * ch);
    float *pool = t.get<float>((int64_t)batch * ch), *gse = t.get<float>((int64_t)batch * ch);
    const int64_t wsn = std::max(stats_ws_floats(batch, ch), (int64_t)batch * nblk * ch * 3);
    float* ws = t.get<float>(wsn);
    float* wsg = t.get<float>((int64_t)batch * nblk * ch);
    int* derr = t.get<int>(batch);
    if (!a || !u || !y || !dw || !db || !w1 || !b1 || !w2 || !b2 || !m0 || !r0 || !m1 || !r1 || !pool || !gse || !ws || !wsg || !derr) { c->err = "hipMalloc failed"; return MICA_ERR_HIP; }
    std::vector<float> wt((size_t)27 * ch);
    for (int cc = 0; cc < ch; ++cc)
        for (int tp = 0; tp < 27; ++tp) wt[(size_t)tp * ch + cc] = h_dw_w[(size_t)cc * 27 + tp];
    HIPC(c, hipMemcpy(dw, wt.data(), sizeof(float) * 27 * ch, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(db, h_dw_b, sizeof(float) * ch, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(w1, h_fc0_w, sizeof(float) * Ch * ch, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(b1, h_fc0_b, sizeof(float) * Ch, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(w2, h_fc3_w, sizeof(float) * ch * Ch, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(b2, h_fc3_b, sizeof(float) * ch, hipMemcpyHostToDevice));
    HIPC(c, hipMemsetAsync(derr, 0, sizeof(int) * batch, st));
    launch_nchw_to_nhwc(d_x, batch, ch, V, a, st);
    launch_stats(a, batch, V, ch, 1e-5f, m0, r0, ws, st);                                  // x3 = relu(IN(x)) is applied on load
    const int P = launch_depthwise(a, batch, dm, ch, m0, r0, nullptr, dw, db, u, ws, wsg, st);
    launch_finalize_sum(wsg, batch, P, ch, 1.0f / (float)V, pool, st);                      // GAP(x3), summed by the depthwise kernel
    launch_gate_mlp(pool, nullptr, batch, ch, Ch, w1, b1, w2, b2, nullptr, gse, nullptr, 0, st);
    launch_stats_finalize(ws, batch, P, ch, 1e-5f, m1, r1, st, gse);                        // the SE gate folded into the norm constants
    launch_prep(u, batch, V, ch, m1, r1, 1, nullptr, SplitView{nullptr, 0, 0, 0}, y, nullptr, ws, SplitEnc{derr, ASCALE_DEFAULT}, st);
    launch_nhwc_to_nchw(y, batch, ch, V, d_y, st);
    CHECK_LAUNCHES(c);
    HIPC(c, hipStreamSynchronize(st));
    return MICA_OK;
}

int mica_op_stem(mica_ctx* c, const float* d_map, int batch, int d, int h, int w, float* d_y, void* stream) {
    if (!c) return MICA_ERR_ARG;
    if (!c->finalized) { c->err = "weights not finalized"; return MICA_ERR_STATE; }
    if (!d_map || !d_y || batch < 1 || !op_box_ok(batch, d, h, w)) { c->err = "mica_op_stem: bad argument" OP_BOX_MSG; return MICA_ERR_ARG; }
    MICA_ENTER(c);
    hipStream_t st = (hipStream_t)stream;
    const int V = d * h * w;
    Tmp t;
    float* raw = t.get<float>((int64_t)batch * V * 128);
    if (!raw) { c->err = "hipMalloc failed"; return MICA_ERR_HIP; }
    run_stem(c, d_map, batch, Dims{d, h, w}, SplitView{nullptr, 0, 0, 0}, raw, nullptr, SplitEnc{nullptr, ASCALE_DEFAULT}, st);
    launch_nhwc_to_nchw(raw, batch, 128, V, d_y, st);
    CHECK_LAUNCHES(c);
    HIPC(c, hipStreamSynchronize(st));
    return MICA_OK;
}

float mica_get_activation_scale(const mica_ctx* c) { return c ? c->ascale : 0.f; }
int mica_get_last_forward_retries(const mica_ctx* c) { return c ? c->last_retries : 0; }

int mica_get_last_forward_input_runs(const mica_ctx* c) { return c ? c->last_input_runs : 0; }

int mica_get_last_forward_af_tiles(const mica_ctx* c) {
    int n = 0;
    if (c) for (char u : c->use_af) n += u != 0;
    return n;
}

float mica_get_last_forward_scale(const mica_ctx* c) { return c ? c->last_scale : 0.f; }

int mica_set_activation_scale(mica_ctx* c, float scale) {
    if (!c) return MICA_ERR_ARG;
    int e = 0;
    if (!(scale >= ASCALE_MIN && scale <= ASCALE_DEFAULT) || std::frexp(scale, &e) != 0.5f) {
        c->err = "mica_set_activation_scale: a power of two in [2^-8, 16] expected";
        return MICA_ERR_ARG;
    }
    c->ascale = scale;
    return MICA_OK;
}

int mica_set_profiling(mica_ctx* c, int enable) {
    if (!c) return MICA_ERR_ARG;
    c->profiling = enable != 0;
    return MICA_OK;
}

int mica_get_profile(mica_ctx* c, int kind, double* h_ms_total, int64_t* h_launches, double* h_work) {
    if (!c || kind < 0 || kind >= PROF_KINDS || !h_ms_total || !h_launches || !h_work) return MICA_ERR_ARG;
    *h_ms_total = c->last_ms[kind];
    *h_launches = c->last_launches[kind];
    *h_work = c->last_work[kind];
    return MICA_OK;
}

int mica_get_conv_profile(mica_ctx* c, double* h_ms_total, int64_t* h_launches, double* h_flops) {
    return mica_get_profile(c, 0, h_ms_total, h_launches, h_flops);
}

}  // extern "C"

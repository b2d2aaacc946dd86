// Internal declarations shared by the HIP translation units of libmica_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <atomic>
#include <vector>

#include "../../include/mica_hip.h"

namespace mica {

// A launch helper that is handed a shape its kernel cannot take does NOT launch and records why (first refusal wins); the C-ABI entry
// points turn a recorded refusal into MICA_ERR_ARG + mica_last_error (forward.hip: CHECK_LAUNCHES).  Nothing in the library aborts
// the host process.
inline thread_local const char* g_launch_refusal = nullptr;
inline void refuse_launch(const char* why) { if (!g_launch_refusal) g_launch_refusal = why; }
inline const char* take_launch_refusal() { const char* w = g_launch_refusal; g_launch_refusal = nullptr; return w; }

// Once-per-device initialisation of a launcher (dynamic-LDS limit of its kernels, CU count) that holds when several host threads drive
// different contexts ("different ctxs are independent", include/mica_hip.h): a device's flag is published only AFTER `init(dev)` has
// run, so no thread can launch with a limit that is not set yet or read a CU count that is not written yet; two threads running
// `init` at the same time is harmless (hipFuncSetAttribute and the values stored are idempotent).  Returns the current device (0..63).
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    template <typename F> int run(F&& init) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) { init(0); return 0; }
        if (!((done.load(std::memory_order_acquire) >> dev) & 1ull)) {
            init(dev);
            done.fetch_or(1ull << dev, std::memory_order_release);
        }
        return dev;
    }
};

// ---- activation formats ----------------------------------------------------------------------
// raw   : float  [B][V][C]              (NDHWC, V = D*H*W voxels)
// split : _Float16 [B][chunks][V][2][16]  a 16-channel chunk of one voxel is 64 B: 16 "hi" halves
//         then 16 "lo" halves with  x * ascale = hi + lo (+ ~2^-22 relative).  This is what the
//         MFMA conv consumes: three f16 MFMAs (hi*hi, hi*lo, lo*hi) give ~fp32 products.
// `ascale` is a power of two carried by the context: 16 by default (keeps `lo` out of f16 subnormals for |x| >= 2^-7);
// when an activation overflows the f16 range at that scale (|x| > 3750) the forward is repeated with ascale / 4
// (forward.hip: forward_checked), down to 2^-8 (|x| < 1.5e7).  Scaling by a power of two is exact, and the conv epilogues
// undo it (out_scale = 1 / (wscale * ascale)).
constexpr float ASCALE_DEFAULT = 16.0f;
constexpr float ASCALE_MIN = 1.0f / 256.0f;
constexpr float F16_LIMIT = 60000.0f;    // |x*ascale| above this sets the range flag
constexpr int RANGE_OVERFLOW = 1, RANGE_NONFINITE = 2;      // bits of the range flag
struct SplitEnc {           // where the split encoders report and how they scale
    int* err;               // device flags int[batch of the launch], one per tile: OR of RANGE_* bits
    float ascale;
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// The split encoder every operand producer uses: x * ascale = hi + lo in f16.  Values beyond the f16 range are flagged and
// SATURATED (an overflow must not turn into Inf/NaN downstream: the caller repeats the tile at a lower scale, forward.hip).
// The range test is one v_med3 + one compare into a scalar mask per value; which kind of violation it was is worked out on a
// wave-uniform slow path that only runs when some lane saw one (the per-value flag logic used to be a third of the encoder's
// VALU work, and the 1x1 kernels are VALU-bound on exactly this).
// development (power experiment, tools/exp/lomask.sh): -DMICA_EXP_LOMASK=n clears the n low mantissa bits of every lo half (activations
// here, weights in the packers): do the matrix cores draw less power - and so hold a higher clock - when the operands toggle fewer bits?
__device__ __forceinline__ _Float16 mica_lo_half(float r) {
    _Float16 l = (_Float16)r;
#ifdef MICA_EXP_LOMASK
    unsigned short b = __builtin_bit_cast(unsigned short, l);
    b &= (unsigned short)(0xFFFFu << MICA_EXP_LOMASK);
    l = __builtin_bit_cast(_Float16, b);
#endif
    return l;
}
__device__ __forceinline__ void mica_split8(const float (&y)[8], half8& hi, half8& lo, int& bad, float ascale) {
    float c[8];
    unsigned long long viol = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = y[j] * ascale;
        c[j] = __builtin_amdgcn_fmed3f(xs, -F16_LIMIT, F16_LIMIT);      // NaN -> -F16_LIMIT (v_med3 returns the minimum when an input is NaN)
        viol |= __builtin_amdgcn_ballot_w64(c[j] != xs);
    }
    if (viol) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!(fabsf(y[j] * ascale) <= F16_LIMIT)) bad |= (fabsf(y[j]) <= 3.0e38f) ? RANGE_OVERFLOW : RANGE_NONFINITE;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)c[j];
        hi[j] = h;
        lo[j] = mica_lo_half(c[j] - (float)h);
    }
}
// Winograd F(4,3) along x with the interpolation points {0, +-3/2, +-2/3, inf} (kernels_conv43.hip): input transform of one
// output quad, d_k = in(4i - 1 + k).  a b = 1 balances the magnitudes of the six transform-domain products (with the textbook
// points {0, +-1, +-2} the "infinity" product is 4x the output's magnitude and enters with cancellation): a layer's rounding error
// is 0.55e-6 rms instead of 0.77e-6 (direct form 0.23e-6, F(2,3) 0.30e-6; measured by emulation).
constexpr float W43_A = 1.5f, W43_B = 0.6666666666666666f, W43_A2 = 2.25f, W43_B2 = 0.4444444444444444f, W43_A3 = 3.375f,
                W43_B3 = 0.2962962962962963f, W43_S = 2.6944444444444446f;      // S = a^2 + b^2
__device__ __forceinline__ void wino43_input_transform(const float (&dv)[6][8], float (&t)[6][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float d0 = dv[0][j], d1 = dv[1][j], d2 = dv[2][j], d3 = dv[3][j], d4 = dv[4][j], d5 = dv[5][j];
        t[0][j] = fmaf(-W43_S, d2, d0 + d4);
        const float e1 = fmaf(-W43_B2, d2, d4), o1 = fmaf(W43_A, d3, -W43_B * d1);      // t1 = e1 + o1 ; t2 = e1 - o1
        t[1][j] = e1 + o1;
        t[2][j] = e1 - o1;
        const float e2 = fmaf(-W43_A2, d2, d4), o2 = fmaf(W43_B, d3, -W43_A * d1);      // t3 = e2 + o2 ; t4 = e2 - o2
        t[3][j] = e2 + o2;
        t[4][j] = e2 - o2;
        t[5][j] = fmaf(-W43_S, d3, d1 + d5);
    }
}
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct SplitView {          // a channel-chunk window into a split buffer
    _Float16* p;            // buffer base
    int chunks_total;       // chunks per batch entry in the buffer
    int chunk_off;          // first chunk of this view
    int chunks;             // chunks in this view
};

constexpr int MAX_SRC = 4;
struct ConvSrcs {
    const _Float16* p[MAX_SRC];
    int chunks_total[MAX_SRC];
    int chunk_off[MAX_SRC];
    int chunks[MAX_SRC];
    int n;
};

struct Dims { int D, H, W; };  // one tile

// Sources of a 1x1x1 conv (kernels_conv1x1.hip): the virtual channel concatenation of up to two tensors, each either an
// operand in split form or the RAW f32 output of its producer with the InstanceNorm constants to apply on load.
struct Conv1Src {
    const void* p;          // split: _Float16 [B][chunks_total][V][2][16] ; raw: float [B][V][16 * chunks_total]
    const float* mean;      // raw only, nullable (identity): per (b, channel) of the raw tensor
    const float* rstd;
    int kind;               // 0 split, 1 raw
    int chunks;             // 16-channel chunks this source contributes
    int chunks_total;       // chunks per batch entry of the buffer (raw: channels / 16)
    int chunk_off;          // first chunk within the buffer
    int relu;               // raw: ReLU after the normalisation
    int cblk;               // raw: 0 = plain NDHWC [V][C]; else the blocked raw layout [C / cblk][V][cblk] (below)
};
struct Conv1Srcs { Conv1Src s[2]; int n; };

// ---- launchers (kernels_*.hip) -----------------------------------------------------------------
// 1x1x1 conv, cout in {64, 128, 256}: out = acc * out_scale + bias written either as raw f32 [B][V][cout] (out_raw) or straight
// as the Winograd operand of the 3^3 conv that follows (wino; needs conv1x1_can_emit_wino(d): whole x rows per workgroup).
// wpk as launch_pack_weights(ksize = 1) lays it out.
// wino_kind: 1 = the F(2,3) operand, 2 = the F(4,3) operand (kernels_conv43.hip; enc_out.ascale = that operand's scale).
// `enc` scales the split encoding of RAW sources staged for the MFMAs and must match out_scale; `enc_out` scales the emitted operand.
void launch_conv1x1(const Conv1Srcs& src, const _Float16* wpk, int64_t wpk_bstride, const float* bias, float out_scale, float* out_raw,
                    SplitView wino, int B, Dims d, int cout, SplitEnc enc, hipStream_t st, int wino_kind = 1, float enc_out_ascale = 0.f);
bool conv1x1_can_emit_wino(Dims d, int wino_kind = 1);
// Pack torch-layout conv weights into wpk.  seg_c/seg_cp: per-source real and padded channel counts.
// cin_scale f32[B][Cin] (nullable) multiplies input channels (gate folding); cout_scale scalar.
void launch_pack_weights(const float* w, int cout, int cin, int ksize, const int* h_seg_c, const int* h_seg_cp,
                         int nseg, const float* cin_scale, int B, float cout_scale, float wscale,
                         _Float16* wpk, hipStream_t st);
int64_t packed_weight_halves(int cout, int ksize, int total_chunks);

// raw f32 NDHWC [B][V][C] -> per (b,c) mean and rstd (biased var, eps) ; partial workspace ws
void launch_stats(const float* x, int B, int V, int C, float eps, float* mean, float* rstd, float* ws,
                  hipStream_t st);
int64_t stats_ws_floats(int B, int C);
// y = relu?((x-mean)*rstd) * scale ; writes split view and/or raw f32 ; optional per (b,c) mean of y (gap)
void launch_prep(const float* x, int B, int V, int C, const float* mean, const float* rstd, int relu,
                 const float* scale, SplitView out, float* out_raw, float* gap, float* ws, SplitEnc enc,
                 hipStream_t st);
// NCDHW f32 [B][C][V] -> split (C padded to 16, zero filled) ; also per-batch |x| sum
void launch_prep_ncdhw(const float* x, int B, int V, int C, SplitView out, float* abs_sum, SplitEnc enc,
                       hipStream_t st);
// abs_sum[b] += sum |x[b][0..n)| (x 16-byte aligned per batch entry when n % 4 == 0)
void launch_abs_sum(const float* x, int B, int64_t n, float* abs_sum, hipStream_t st);
void launch_finalize_sum(const float* ws, int B, int nblocks, int C, float inv, float* out, hipStream_t st);
void launch_nchw_to_nhwc(const float* x, int B, int C, int V, float* y, hipStream_t st);
void launch_nhwc_to_nchw(const float* x, int B, int C, int V, float* y, hipStream_t st);

// gate[b][c] = sigmoid(W2 relu(W1 (pool[b]*premul[b]) + b1) + b2) ; out = gate * postmul (nullable)
// gate_post (nullable) is written at gate_post[b*post_stride + c] (a slice of a conv's cin-scale row)
void launch_gate_mlp(const float* pool, const float* premul, int B, int C, int Ch, const float* w1,
                     const float* b1, const float* w2, const float* b2, const float* postmul, float* gate,
                     float* gate_post, int post_stride, hipStream_t st);
void launch_fill_float(float* p, int64_t n, float v, hipStream_t st);

// depthwise 3^3 on raw input with fused (x-mean)*rstd, relu, *scale applied on load (zero padding after)
// C must be a multiple of 16.  stats_ws (nullable): fused InstanceNorm partials, returns their count P.
// gap_ws (nullable): f32 [B][P][C] per-block sums of the normalised input over the block's own voxels (launch_finalize_sum).
// cblk (0 = plain NDHWC): x AND out are in the blocked raw layout float [B][C / cblk][V][cblk] - round 5: the two tensors around the
// depthwise conv (conv3's raw output, its own output) are stored in blocks of 32 channels, so that the 32-channel slab a depthwise
// workgroup streams and the chunk pair a 1x1 workgroup stages are contiguous runs of memory instead of 128-byte pieces at a pitch of
// C floats (tools/exp/dw_layout.py: the same kernel at C = 32, where a slab IS the row, streams 11 % faster than at C = 256)
int launch_depthwise(const float* x, int B, Dims d, int C, const float* mean, const float* rstd,
                     const float* scale, const float* w27, const float* bias, float* out, float* stats_ws, float* gap_ws, hipStream_t st,
                     int cblk = 0);
// merge P partials f32 [B][P][C][3] = (count, mean, M2) into mean / rstd.  gate f32 [B][C] (nullable): the statistics are
// those of u while the tensor that is normalised downstream is t = g u + const (g > 0 per tile and channel): then
// (t - mean_t) / sqrt(var_t + eps) = (u - mean_u) * g / sqrt(g^2 var_u + eps), i.e. rstd = g / sqrt(g^2 var_u + eps).
void launch_stats_finalize(const float* ws, int B, int P, int C, float eps, float* mean, float* rstd, hipStream_t st,
                           const float* gate = nullptr);
int64_t fused_stats_ws_floats(int B, int tile_size);
// stem: map f32 [B][V] -> split view of 128 channels + gap[b][128] (mean over voxels)
void launch_stem(const float* map, int B, Dims d, const float* wstem, const float* bstem, SplitView out,
                 float* out_raw, float* gap, float* ws, SplitEnc enc, hipStream_t st);
int64_t stem_weight_floats();
// the stem on the matrix cores (kernels_stem.hip), for tile widths that are multiples of 64
struct StemPlan { int first[4][8]; int steps[4][8]; int records; };      // K-step records of (kernel size, residue class of x mod 8)
void stem_mfma_plan(const float* w, std::vector<float>& wf, std::vector<int>& aoff, StemPlan& plan);
void launch_stem_mfma_pack(const float* d_wf, int records, float wscale, _Float16* d_rec, hipStream_t st);
bool stem_mfma_eligible(Dims d);
int launch_stem_mfma(const float* map, int B, Dims d, const _Float16* wrec, const int* aoff, const StemPlan& plan, float wscale,
                     const float* bstem, SplitView out, float* out_raw, float* ws, SplitEnc enc, hipStream_t st);
// x_feat raw [B][V][64] -> split(x_feat * sigmoid(w2 . relu(W0 x + b0) + b2))
void launch_feat_gate(const float* x, int B, int V, const float* w0, const float* b0, const float* w2,
                      const float* b2, SplitView out, SplitEnc enc, hipStream_t st);
// head tail: raw2 [B][V][32] -> logits NCDHW [B][ncls][V]
// extra_raw (nullable): also store the logits as channels [extra_ch_off, +ncls) of f32 [B][extra_raw_c][V]
void launch_head_final(const float* x, int B, int V, const float* mean, const float* rstd, const float* gate,
                       const float* wf, const float* bf, int ncls, float* logits, int extra_ch_off, float* extra_raw,
                       int extra_raw_c, hipStream_t st);
// ---- Winograd F(2,3) along x (dense 3^3 convs): operand layout [B][chunks][4 p][4 q][Vh][8], Vh = D*H*ceil(W/2)
void launch_prep_wino(const float* x, int B, Dims d, int C, const float* mean, const float* rstd, int relu,
                      const float* scale, SplitView wino, SplitView plain, float* gap, float* ws, SplitEnc enc,
                      hipStream_t st);
void launch_prep_ncdhw_wino(const float* x, int B, Dims d, int C, SplitView wino, SplitEnc enc, hipStream_t st);
int launch_conv_wino(const ConvSrcs& s, const _Float16* wpk, int64_t wpk_bstride, const float* bias,
                     float out_scale, float* out, int B, Dims d, int cout, float* stats_ws, hipStream_t st, int out_cblk = 0);
void launch_pack_weights_wino(const float* w, int cout, int cin, const int* h_seg_c, const int* h_seg_cp, int nseg,
                              const float* cin_scale, int B, float cout_scale, float wscale, _Float16* wpk,
                              hipStream_t st);
int64_t packed_weight_halves_wino(int cout, int total_chunks);
// ---- Winograd F(4,3) along x (kernels_conv43.hip; the 3^3 convs of encoder.2): operand layout [B][chunks][6 p][4 q][Vq][8],
// Vq = D*H*ceil(W/4); `enc.ascale` of its producers is the OPERAND's scale (callers pass ascale / WINO43_ASCALE_DIV)
constexpr float WINO43_ASCALE_DIV = 4.0f;
bool conv_wino43_eligible(int cout);
void launch_prep_wino43(const float* x, int B, Dims d, int C, const float* mean, const float* rstd, int relu, SplitView wino,
                        SplitEnc enc, hipStream_t st);
void launch_prep_ncdhw_wino43(const float* x, int B, Dims d, int C, SplitView wino, SplitEnc enc, hipStream_t st);
int launch_conv_wino43(const ConvSrcs& s, const _Float16* wpk, int64_t wpk_bstride, const float* bias, float out_scale, float* out,
                       int B, Dims d, int cout, float* stats_ws, hipStream_t st, int out_cblk = 0);
void launch_pack_weights_wino43(const float* w, int cout, int cin, const int* h_seg_c, const int* h_seg_cp, int nseg,
                                const float* cin_scale, int B, float cout_scale, float wscale, _Float16* wpk, hipStream_t st);
int64_t packed_weight_halves_wino43(int cout, int total_chunks);
void launch_postprocess(const float* bb, const float* ca, const float* aa, int B, int V, float* bbp, float* cap,
                        float* aap, float* aapred, int64_t s1, int64_t s20, hipStream_t st);

void launch_gather_tiles(const float* vol, int C, int64_t n0, int64_t n1, int64_t n2, int grid, int pad,
                         int64_t first, int64_t count, float* tiles, hipStream_t st);
void launch_gather_tiles_u8(const uint8_t* vol, int C, int64_t n0, int64_t n1, int64_t n2, int grid, int pad,
                            int64_t first, int64_t count, float* tiles, hipStream_t st);
void launch_stitch_tiles(const float* tiles, int C, int64_t n0, int64_t n1, int64_t n2, int grid, int pad,
                         int64_t first, int64_t count, float* vol, hipStream_t st);
// exact order statistics of a f32 array (radix select) ; see kernels_select.hip
int normalise_map_device(float* d_vol, int64_t n, int kind, int numpy_rules, double* h_stats, hipStream_t st, char* err, int errlen);
// scipy.ndimage.zoom(order=3) restated bit-exactly in f64 ; see kernels_zoom.hip (synchronous)
int zoom_cubic_device(const float* d_in, int64_t n0, int64_t n1, int64_t n2, int64_t o0, int64_t o1, int64_t o2, int kind, float* d_out,
                      hipStream_t st, char* err, int errlen);

// AF3 encoding rasteriser (preprocessing.py:172-178,283-298) ; zeroes d_vol f32[24][nz][ny][nx] first ; synchronous
int rasterise_atoms_device(const float* d_xyz, const int* d_bb, const int* d_aa, int64_t n_atoms, const float* origin, int64_t nz,
                           int64_t ny, int64_t nx, float* d_vol, hipStream_t st, char* err, int errlen);

// point-list kernels (kernels_points.hip; modeler.py:767, 836-852) ; threshold and gather are synchronous
int threshold_points_device(const float* d_vol, int64_t n, float thr, int64_t* d_idx, int64_t capacity, int64_t* h_count, hipStream_t st,
                            char* err, int errlen);
int gather_values_device(const float* d_vol, int C, int64_t nvox, const int64_t* d_idx, int64_t n, float* d_out, hipStream_t st, char* err,
                         int errlen);
int refine_candidates_device(const float* d_ca, const float* d_aa, int n0, int n1, int n2, const int* d_cand, int64_t n, double* d_coord,
                             float* d_aa_out, int* d_ok, hipStream_t st, char* err, int errlen);

// modeler.py:776-787 (np.sum per cluster, numpy's summation order), :822-831 (greedy NMS over sorted candidates), :860-888
int segment_sums_device(const float* d_vals, const int64_t* d_seg_off, int64_t nseg, float* d_sums, hipStream_t st, char* err, int errlen);
int nms_points_device(const int* d_pts, int64_t n, int n0, int n1, int n2, double radius, int* d_keep, hipStream_t st, char* err, int errlen);
int neighbour_matrix_device(const double* d_cands, int64_t n, const float* d_bb, int n0, int n1, int n2, double* d_dis, double* d_mat,
                            int legacy, hipStream_t st, char* err, int errlen);

}  // namespace mica

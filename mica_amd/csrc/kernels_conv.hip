// Dense convolution on gfx950 matrix cores: implicit GEMM over 16-channel chunks with split-f16 operands (x = hi + lo,
// three f16 MFMA products per f32-grade product: lo*hi, hi*lo, hi*hi), which keeps ~22 mantissa bits per product at 3/16
// of the f32-MFMA cost (MI355X: f32 MFMA 157 TF, f16 2.5 PF).  Replaces nn.Conv3d(k=3,pad=1) of reference
// models/model.py:17,107,115,122,142,165-174,210,212.
//   conv_wino16_kernel<128|64|32> : every 3^3 conv - Winograd F(2,3) along x, v_mfma_f32_16x16x32_f16, persistent
// (the 1x1x1 convs are in kernels_conv1x1.hip)
// Also here: weight packing, the depthwise 3^3 conv (model.py:80) and the Cin=1 multi-scale stem (model.py:9-14).
#include "common.h"
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace mica {

struct Segs { int c[MAX_SRC]; int cp[MAX_SRC]; int n; };   // channel segmentation of a conv's concatenated input

// ================================================================================================
// Winograd geometry shared by the kernel below: dense 3x3x3 conv with Winograd F(2,3) along x.  The kernel is
// power/MFMA-issue bound (1.25 PF of f16 MFMA measured), so the lever left is fewer MFMAs: per output pair
// 4 transform-domain products replace 6 taps => 9 (dz,dy) taps x 4 positions instead of 27 taps x 2 outputs
// = 1.5x fewer MFMAs for the same result (products stay split-f16, accumulate f32; the transforms are
// +-1 / 0.5 combinations done in f32 by the producer (prep_wino) and the weight packer).
//   y[2i]   = m0 + m1 + m2          m_p = sum_{dz,dy,cin} t_p * u_p
//   y[2i+1] = m1 - m2 - m3          u0 = g0, u1 = (g0+g1+g2)/2, u2 = (g0-g1+g2)/2, u3 = g2
// Workgroup: 8 waves, output tile 16(x) = 8 pairs x 4(y) x 4(z); wave (p, wn) owns Winograd position p for
// all 128 (pair,y,z) rows and 64 channels.  LDS image per chunk: 4 planes (hi/lo x k-half) x
// [z 6][p 4][y 6][pair 8] 16-B slots = 73,728 B, double buffered, filled by LDS-DMA; a row fragment is
// (8 pairs x 4 y) of one z: y rows are 8 slots apart => every ds_read_b128 lane group covers 16 distinct slots.
// Epilogue: the four position-waves exchange their accumulators through LDS and apply the output transform.
// ================================================================================================
struct GeoW {
    static constexpr int SY = 6, SZ = 6;
    static constexpr int PP = SY * 8;               // slots per (z, p)
    static constexpr int PZ = 4 * PP;               // 192 slots per z plane = 3 DMA instructions
    static constexpr int PLANE = SZ * PZ;           // 1152
    static constexpr int CH_BYTES = 4 * PLANE * 16; // 73,728
    static constexpr int NT = 9;
    static constexpr int DPZ = PZ / 64;
    static constexpr int DMA_PER_CHUNK = 4 * SZ * DPZ;   // 72
    static constexpr int DPW = DMA_PER_CHUNK / 8;        // 9
};

__device__ __forceinline__ const _Float16* chunk_base_wino(const ConvSrcs& s, int gch, int b, int Vh) {
    int si = 0, ch = gch;
#pragma unroll
    for (int i = 0; i < MAX_SRC - 1; ++i)
        if (si == i && i + 1 < s.n && ch >= s.chunks[i]) { ch -= s.chunks[i]; si = i + 1; }
    return s.p[si] + ((int64_t)b * s.chunks_total[si] + s.chunk_off[si] + ch) * (int64_t)Vh * 128;
}

// ================================================================================================
// conv_wino16: the 128-channel variant of conv_wino on the v_mfma_f32_16x16x32_f16 shape, as a PERSISTENT kernel.
//
// MFMA shape.  The kernel is power-bound, and on this chip the 16x16x32 shape sustains 1.43x the f16 FLOP/s of 32x32x16 on
// random operands held in registers (tools/mfma_shape_bench.hip: 1775 vs 1241 TF).  K = 32 per instruction is filled from
// ONE 16-channel chunk by regrouping the three split products:
//   X(t)    : A = [a_hi | a_lo] (the four LDS planes are exactly the four k-groups),  B = [b_hi ; b_hi]
//             = a_hi.b_hi + a_lo.b_hi of tap t
//   Y(t,t') : A = [a_hi(t) | a_hi(t')],  B = [b_lo(t) ; b_lo(t')]   = the a_hi.b_lo terms of two taps
// 9 taps = 4 pairs + tap 8 alone (its Y has a zero upper half): 14 instead of 13.5 MFMAs per tile and chunk.
// A wave owns 128 rows x 64 channels = 8 row fragments (8 pairs x 2 y of one z) x 4 column tiles; a chunk is 14 steps
// (pair-step, kind Y / X' / X) of 32 MFMAs on 32 distinct accumulators.  A step's four weight fragments are fetched one
// step ahead by inline-asm loads into two register sets; the A fragments stream through a four-deep register pipeline.
//
// Persistence.  Only one workgroup fits a CU (147 KB of LDS), so a workgroup's prologue (first slab, HBM latency), its
// epilogue and the dispatch gap were all exposed: 30 K of 290 K cycles per 256-channel tile.  Here one workgroup per CU
// walks its share of the (batch, tile, channel block) items; the slab and weight pipelines run ACROSS items (the last
// chunk of an item fetches chunk 0 of the next), and the output transform uses only the slab buffer that is idle.
// Every LDS slot of a slab is rewritten per chunk (DMA, or an explicit zero for positions outside the volume).
// ================================================================================================
typedef float floatx4v __attribute__((ext_vector_type(4)));
// development switch: -DMICA_EXP_NOEPI builds a kernel without the output-transform passes (timing experiments only)
#ifdef MICA_EXP_NOEPI
#define MICA_EXP_EPI_PASSES(n) 0
#else
#define MICA_EXP_EPI_PASSES(n) (n)
#endif
// energy ablations (timing experiments only, results are garbage): -DMICA_EXP_SLAB_FIXED wraps every slab DMA into the first 4 MB of
// the operand (the data stay random - a constant slab would lower the MFMA power by itself - but come from L2: no HBM traffic for
// the operand, the LDS-DMA writes remain); -DMICA_EXP_W_FIXED makes every weight fragment load hit the same 4 KB per wave (L1 hits
// instead of the L2 stream).  Measured on single layers with random inputs (tools/exp/abl.sh): inside the network an ablated conv
// feeds garbage to the next one and changes ITS power.
#ifdef MICA_EXP_SLAB_FIXED
#define MICA_EXP_SLABOFF(x) ((x) & 0x3FFFF0)      /* wrapped into the first 4 MB of the operand: random data, served by L2 */
#define MICA_EXP_SLABBASE(b) (s.p[0])
#else
#define MICA_EXP_SLABOFF(x) (x)
#define MICA_EXP_SLABBASE(b) (b)
#endif
// -DMICA_EXP_ROWFRAGS=n (n < 8): only the first n of a step's eight row fragments issue their MFMAs while every operand is still
// moved - the operand bytes per MFMA of a 2-D Winograd variant (whose tiles hold 2 row fragments per weight fragment) at today's
// instruction stream: measures how fast the weight / slab streams can run when the matrix pipe does not limit them.
#ifndef MICA_EXP_ROWFRAGS
#define MICA_EXP_ROWFRAGS 8
#endif
#ifdef MICA_EXP_W_FIXED
#define MICA_EXP_WBASE(b) (wwave)
#else
#define MICA_EXP_WBASE(b) (b)
#endif
#ifndef MICA_BLOCKED_WALK
#define MICA_BLOCKED_WALK 1
#endif


__device__ __forceinline__ constexpr int w16_step_ps(int st) { return st < 12 ? st / 3 : 4; }
__device__ __forceinline__ constexpr int w16_step_kind(int st) { return st < 12 ? st % 3 : (st - 12) * 2; }   // 0 Y, 1 X' (second tap), 2 X (first tap)
// LDS byte offset of the k-th slab DMA instruction of `wave` (wave-uniform)
__device__ __forceinline__ int w16_slab_loff(int wave, int k) {
    const int ii = wave * GeoW::DPW + k;
    const int q = ii / (GeoW::SZ * GeoW::DPZ), rem = ii % (GeoW::SZ * GeoW::DPZ);
    return (q * GeoW::PLANE + (rem / GeoW::DPZ) * GeoW::PZ + (rem % GeoW::DPZ) * 64) * 16;
}
// Tile-independent part of the global BYTE offset (within a 16-channel chunk of the wino operand) this lane fetches with DMA
// instruction k, relative to the slab origin (z0, y0, i0); the slab row vy of the lane rides in the low three bits
// (the offset is a multiple of 16).  vz is wave-uniform: w16_slab_vz.
__device__ __forceinline__ int w16_slab_rel(int wave, int lane, int k, int H, int Wh, int Vh) {
    const int ii = wave * GeoW::DPW + k;
    const int q = ii / (GeoW::SZ * GeoW::DPZ), rem = ii % (GeoW::SZ * GeoW::DPZ);
    const int vz = rem / GeoW::DPZ, part = rem % GeoW::DPZ;
    const int slot = part * 64 + lane;
    const int pp = slot / GeoW::PP, r2 = slot - pp * GeoW::PP;
    const int vy = r2 >> 3, pr = r2 & 7;          // pr == lane & 7 for every k
    return (((pp * 4 + q) * Vh + (vz * H + vy) * Wh + pr) * 8) * 2 + vy;
}
__device__ __forceinline__ int w16_slab_vz(int wave, int k) { return ((wave * GeoW::DPW + k) % (GeoW::SZ * GeoW::DPZ)) / GeoW::DPZ; }
// Chan's pairwise merge of (count, mean, M2)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float on, float om, float oq) {
    const float tn = n + on;
    if (tn > 0.f) {
        const float dl = om - mean;
        const float mm = (n > 0.f) ? mean + dl * (on / tn) : om;
        const float qq = (n > 0.f && on > 0.f) ? m2 + oq + dl * dl * (n * on / tn) : (n > 0.f ? m2 : oq);
        mean = mm;
        m2 = qq;
    }
    n = tn;
}

template <int BN>
__global__ __launch_bounds__(512, 2) void conv_wino16_kernel(ConvSrcs s, const _Float16* __restrict__ wpk, int64_t wpk_bstride,
                                                             const float* __restrict__ bias, float out_scale,
                                                             float* __restrict__ out, Dims d, int cout, int total_chunks,
                                                             int ntx, int nty, int nnb, int items_per_b, int total_items,
                                                             float* __restrict__ stats_ws, int ocb) {
    // ocb: channel block of the OUTPUT layout (common.h: raw tensors): cout for plain NDHWC [V][cout], else [cout / ocb][V][ocb]
    using G = GeoW;
    static_assert(BN == 128 || BN == 64 || BN == 32, "three variants");
    // BN = 128: the two wave groups (wn) own 64 output channels each and walk all 14 steps of a chunk.
    // BN = 64 : both groups own the same 64 channels and SPLIT THE TAPS, 7 steps each (group 0: taps 0..4 less the X terms of
    //           tap 4/5, group 1: the rest); the per-wave tile, operand reuse and weight traffic per MFMA stay those of the
    //           128 variant, and the two partial accumulators are added through LDS before the output transform.
    // BN = 32 : as BN = 64 with two column tiles per wave instead of four (an A fragment then feeds two MFMAs; LDS has room).
    constexpr bool SPLIT = BN <= 64;
    constexpr int NS = SPLIT ? 7 : 14;              // steps per wave and chunk
    constexpr int WNC = BN == 32 ? 32 : 64;         // output channels per wave
    constexpr int NCT = WNC / 16;                   // column tiles (and weight fragments per step) per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 3, wn = wave >> 2;
    const int Wh = (d.W + 1) >> 1;
    const int V = d.D * d.H * d.W, Vh = d.D * d.H * Wh;

    // persistent schedule: workgroup g sits on XCD g & 7 (round-robin dispatch); each XCD takes a contiguous eighth of the
    // items so that the 32 CUs sharing an L2 work on neighbouring tiles and on the same weights at the same time
    const int xcd = blockIdx.x & 7, lwg = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int range = (total_items + 7) >> 3;
    const int it_end = min(total_items, (xcd + 1) * range);
    int item = xcd * range + lwg;
    if (item >= it_end) return;

    // A operand addressing: lane = (row r = lane&15 -> pair r&7, y-in-fragment r>>3 ; k-group g = lane>>4).
    //   X steps : k-group g reads LDS plane g                       = plane (g&1) + (g>>1) * 2 planes
    //   Y steps : k-groups 0,1 read hi planes of tap t, 2,3 of t'   = plane (g&1) + (g>>1) * (tap delta: one y row, or for
    //             the pair (2,3) one z plane less two y rows)
    // so every step is "common lane address + (g>>1 ? step delta : 0) + step tap offset" with per-step SCALARS - which lets
    // the two wave groups of the BN = 64 variant run the same instruction stream on different taps.
    const int lr = lane & 15, lg = lane >> 4;
    const int himask = (lg >> 1) ? -1 : 0;
    const int a_common = ((lg & 1) * G::PLANE + wp * G::PP + (lr >> 3) * 8 + (lr & 7)) * 16;

    // packed weights: [nb][chunk][pair-step 5][p 4][unit 8][128 cout][8 halves] - a workgroup's slice is contiguous and
    // every stride is a compile-time constant; units: 0,1 hi(t) k-half 0,1 | 2,3 hi(t') | 4,5 lo(t) | 6,7 lo(t')
    constexpr int ustride = BN * 16;                        // bytes per unit
    constexpr int psstride = 4 * 8 * ustride;               // bytes per pair-step (65,536)
    constexpr int chstride = 5 * psstride;                  // bytes per chunk
    const unsigned w_common = (unsigned)((lg & 1) * BN + lr) * 16u;     // X steps: units (g&1); Y steps: + (g>>1) * 2 units
    const int64_t nbstride = (int64_t)total_chunks * chstride;
    const char* wwave = reinterpret_cast<const char*>(wpk) + wp * 8 * ustride + (SPLIT ? 0 : wn * WNC * 16);

#ifdef MICA16_TWOAHEAD
    // experiment (round 6): in the tap-split variants the fragments of step ls + 2 are requested in step ls (four register sets, set =
    // ls & 3: the assignment closes over the seven steps of a chunk), so that a slab DMA issued in step ls is forced complete by the
    // in-order queue at the wait of step ls + 3 instead of ls + 2
    constexpr bool TWOAHEAD = SPLIT;
#else
    constexpr bool TWOAHEAD = false;
#endif
    half8 bq[TWOAHEAD ? 4 : 3][NCT];        // weight fragment sets; the third only when NS is odd (see W16_SET)
    // step parameters: global step st (0..13) -> scalars; a wave's local step ls is global step ls (+ 7 for group 1 of BN = 64)
#define W16_TAP(st) (2 * w16_step_ps(st) + (w16_step_kind(st) == 1 ? 1 : 0))
#define W16_AOFF(st) (((W16_TAP(st) / 3) * G::PZ + (W16_TAP(st) % 3) * 8) * 16)
#define W16_ADELTA(st) (w16_step_kind(st) != 0 ? 2 * G::PLANE * 16 : (w16_step_ps(st) == 1 ? (G::PZ - 16) * 16 : 8 * 16))
#define W16_WOFF(st) (w16_step_ps(st) * psstride + (w16_step_kind(st) == 0 ? 4 : w16_step_kind(st) == 1 ? 2 : 0) * ustride)
#define W16_WDELTA(st) (w16_step_kind(st) == 0 ? 2 * BN * 16 : 0)
#define W16_SEL(ls, M) ((SPLIT && wsel == 1) ? M((ls) + 7) : M(ls))
#define MICA_BLOAD16(set, wbase, ls)                                                                                    \
    do {                                                                                                                \
        const char* pb_ = MICA_EXP_WBASE(wbase) + W16_SEL(ls, W16_WOFF);                                                \
        const unsigned vo_ = w_common + (unsigned)(W16_SEL(ls, W16_WDELTA) & himask);                                   \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[set][0]) : "v"(vo_), "s"(pb_) : "memory");              \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:256" : "=v"(bq[set][1]) : "v"(vo_), "s"(pb_) : "memory");   \
        if (NCT == 4) {                                                                                                 \
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" : "=v"(bq[set][NCT - 2]) : "v"(vo_), "s"(pb_) : "memory"); \
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:768" : "=v"(bq[set][NCT - 1]) : "v"(vo_), "s"(pb_) : "memory"); \
        }                                                                                                               \
    } while (0)
    // One slab DMA instruction: issued UNCONDITIONALLY (lanes outside the volume are masked off by hand and write an explicit
    // zero instead) so that the number of vector-memory operations in flight is known at compile time: the weight waits can
    // then leave the newest DMA outstanding instead of exposing its HBM latency every step.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#define MICA_SLAB_DMA(srcbase, bufoff, k, org)                                                                          \
    do {                                                                                                                \
        const int lo_ = (bufoff) + w16_slab_loff(wave, k);                                                              \
        const unsigned la_ = __builtin_amdgcn_readfirstlane(lds0 + lo_);                                                \
        const bool ok_ = (org).i0 + (lane & 7) < Wh && (unsigned)((org).y0 + (rel[k] & 7)) < (unsigned)d.H &&           \
                         (unsigned)((org).z0 + w16_slab_vz(wave, k)) < (unsigned)d.D;                                   \
        const int go_ = ok_ ? MICA_EXP_SLABOFF((org).base + (rel[k] & ~15)) : -1;                                       \
        unsigned long long sv_;                                                                                         \
        asm volatile("s_mov_b64 %0, exec\n\tv_cmp_lt_i32 vcc, -1, %1\n\ts_mov_b64 exec, vcc\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" \
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"                                              \
                     : "=&s"(sv_) : "v"(go_), "s"(MICA_EXP_SLABBASE(srcbase)), "s"(la_) : "memory", "vcc", "m0");        \
        if (!ok_) *reinterpret_cast<uint4*>(smem + lo_ + lane * 16) = make_uint4(0, 0, 0, 0);                           \
    } while (0)

    // Everything per item is scalar and is worked out ONCE per item, one item ahead (during the previous item's epilogue):
    // decoding an item costs integer divisions and kernel-argument loads that must not sit between a chunk's barrier and its
    // first MFMA.
    struct Item {
        int b, nb, tile;
        int i0, y0, z0, base;          // slab origin: first x pair, y, z (halo included) and its byte offset in a chunk
        const char* w;                 // this wave's weights of chunk 0
        const _Float16* src0;          // chunk 0 of the operand
    };
    auto decode = [&](int it) {
        Item r;
        r.b = it / items_per_b;
        const int id = it - r.b * items_per_b;
        r.nb = id % nnb;
        // walk order of the tiles: the 32 workgroups of an XCD work on 32 consecutive items at a time, and what neighbouring
        // tiles share is their y and z halo (none in x) - so consecutive items form compact (8 y x 4 z) blocks of tiles whose
        // halos the XCD's L2 serves, rather than x-major rows (1.2x instead of 1.6x the ideal slab traffic at Cout <= 128)
        const int seq = id / nnb;
        int tx, ty, tz;
        if (MICA_BLOCKED_WALK && ((nty & 7) | (((d.D + 3) >> 2) & 3)) == 0) {
            const int inb = seq & 31, blk = seq >> 5, nby = nty >> 3;
            tx = blk % ntx;
            ty = (blk / ntx % nby) * 8 + (inb & 7);
            tz = (blk / (ntx * nby)) * 4 + (inb >> 3);
        } else {
            tx = seq % ntx;
            ty = seq / ntx % nty;
            tz = seq / (ntx * nty);
        }
        r.tile = (tz * nty + ty) * ntx + tx;
        r.i0 = tx * 8;
        r.y0 = ty * 4 - 1;
        r.z0 = tz * 4 - 1;
        r.base = ((r.z0 * d.H + r.y0) * Wh + r.i0) * 16;
        r.w = wwave + (int64_t)r.b * wpk_bstride * 2 + r.nb * nbstride;
        r.src0 = chunk_base_wino(s, 0, r.b, Vh);
        return r;
    };
    const int64_t chunk_halves = (int64_t)Vh * 128;

    Item cur = decode(item);
    int nitem = item + per_xcd;
    Item nxt = decode(nitem < it_end ? nitem : item);     // past the end: the pipelines re-fetch this item's first chunk, harmlessly
    int rel[G::DPW];
#pragma unroll
    for (int k = 0; k < G::DPW; ++k) rel[k] = w16_slab_rel(wave, lane, k, d.H, Wh, Vh);

    // step ls of a wave's NS uses set W16_SET(ls): alternating, except that with NS odd the first step of a chunk has a set
    // of its own (its fragments are fetched during the last step of the previous chunk, which uses set 0 itself)
#define W16_SET(ls) (TWOAHEAD ? ((ls) & 3) : (SPLIT && (ls) == 0) ? 2 : ((ls) & 1))
    // slab DMA instructions issued in step ls (9 per wave and chunk, none in the last step so that the chunk-end wait can
    // leave exactly the next chunk's weight loads in flight)
#if defined(MICA16_DMAPLAN) && MICA16_DMAPLAN == 1        /* experiments (round 6): the nine DMAs of a wave as 0,5,4 ... */
#define W16_NDMA(ls) (SPLIT ? ((ls) == 1 ? 5 : (ls) == 2 ? 4 : 0) : ((ls) < G::DPW ? 1 : 0))
#define W16_DMA0(ls) (SPLIT ? ((ls) == 2 ? 5 : 0) : (ls))
#elif defined(MICA16_DMAPLAN) && MICA16_DMAPLAN == 2      /* 3,3,3 */
#define W16_NDMA(ls) (SPLIT ? ((ls) < 3 ? 3 : 0) : ((ls) < G::DPW ? 1 : 0))
#define W16_DMA0(ls) (SPLIT ? 3 * (ls) : (ls))
#elif defined(MICA16_DMAPLAN) && MICA16_DMAPLAN == 3      /* 0,3,3,3 */
#define W16_NDMA(ls) (SPLIT ? ((ls) >= 1 && (ls) < 4 ? 3 : 0) : ((ls) < G::DPW ? 1 : 0))
#define W16_DMA0(ls) (SPLIT ? 3 * ((ls) - 1) : (ls))
#elif defined(MICA16_DMAPLAN) && MICA16_DMAPLAN == 4      /* 0,9 */
#define W16_NDMA(ls) (SPLIT ? ((ls) == 1 ? 9 : 0) : ((ls) < G::DPW ? 1 : 0))
#define W16_DMA0(ls) (SPLIT ? 0 : (ls))
#else
#define W16_NDMA(ls) (SPLIT ? ((ls) < 3 ? 2 : (ls) < 6 ? 1 : 0) : ((ls) < G::DPW ? 1 : 0))
#define W16_DMA0_DEFAULT
#endif
    // steps in which group 0 holds the high priority: 9 of 14 (0,1,3,4,6,7,9,10,12) resp. 4 of 7 (0,1,3,4); measured flat
    // between 7 and 10 of 14
#define W16_XHI(ls) (((SPLIT ? 0x1B : 0x16DB) >> (ls)) & 1)
#ifdef W16_DMA0_DEFAULT
#define W16_DMA0(ls) (SPLIT ? ((ls) < 3 ? 2 * (ls) : (ls) + 3) : (ls))
#endif
    // prologue of the first item: slab chunk 0 -> buffer 0, weights of the wave's first step
    int wsel = wn;
    MICA_BLOAD16(W16_SET(0), cur.w, 0);
    if constexpr (TWOAHEAD) MICA_BLOAD16(W16_SET(1), cur.w, 1);
#pragma unroll
    for (int k = 0; k < G::DPW; ++k) MICA_SLAB_DMA(cur.src0, 0, k, cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int par = 0;                                   // slab buffer holding the current chunk

    for (;;) {
        const bool has_next = nitem < it_end;
        floatx4v acc[8][NCT];
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[f][c][i] = 0.f;

        // running pointer to the operand chunk after the current one (the sources of a virtual concat are walked in order)
        const _Float16* run = cur.src0;
        int si = 0, left = s.chunks[0];
        const char* wcur = cur.w;
#pragma clang loop unroll(disable)      // also keeps the first iteration from being peeled into a second copy of the body
        for (int gch = 0; gch < total_chunks; ++gch) {
            const char* A = smem + par * G::CH_BYTES;
            const int nxt_off = (par ^ 1) * G::CH_BYTES;
            const bool last = gch + 1 == total_chunks;
            // the group selector is laundered per chunk: as a loop invariant it invites the compiler to clone the whole chunk
            // body per group, and the audit of the hand-waited loads (tools/audit_asm_loads.py) reads straight-line code
            wsel = wn;
            asm volatile("" : "+s"(wsel));
            if (!last) {
                if (--left == 0) {
                    ++si;
                    left = s.chunks[si];
                    run = s.p[si] + ((int64_t)cur.b * s.chunks_total[si] + s.chunk_off[si]) * chunk_halves;
                } else {
                    run += chunk_halves;
                }
            }
            const _Float16* nsrc = last ? nxt.src0 : run;        // the last chunk's DMAs fetch the NEXT item's first slab
            Item org = cur;
            if (last) org = nxt;
            const char* wnxt = last ? nxt.w : wcur + chstride;
            // The NS steps of this wave in this chunk.  Per step: fetch the next step's weight fragments, wait for this
            // step's (in flight and NEWER: the four loads just issued plus the slab DMAs of the previous step; loads return in
            // order), issue this step's slab DMAs, then 8 groups (row fragment f = (z, y pair)) of 4 MFMAs.  The A fragments
            // stream through a four-deep register pipeline that runs across the steps (the fragment three groups ahead is read
            // while this group's MFMAs issue); the order is pinned per group so the scheduler cannot pull more LDS reads forward
            // than the register budget (256) allows.  The MFMA is asm: that ties the accumulator in place (the untied builtin
            // lets the allocator rotate 128 accumulator registers through the file and spill).
#define W16_ABASE(ls) (A + a_common + (W16_SEL(ls, W16_ADELTA) & himask) + W16_SEL(ls, W16_AOFF))
#define W16_AFRAG(base, f) (*reinterpret_cast<const half8*>((base) + ((((f) >> 1) * G::PZ + ((f) & 1) * 16) * 16)))
            const char* ab_cur = W16_ABASE(0);
            // AD fragments ahead: 3 groups of four MFMAs, or 6 groups of two (BN = 32), cover the LDS latency
            constexpr int AD = NCT == 4 ? 3 : 6;
            half8 ar[AD + 1];
#pragma unroll
            for (int i = 0; i < AD; ++i) ar[i] = W16_AFRAG(ab_cur, i);
#pragma unroll
            for (int ls = 0; ls < NS; ++ls) {
                half8 (&bc)[NCT] = bq[W16_SET(ls)];
                if constexpr (TWOAHEAD) {
                    if (ls + 2 < NS) MICA_BLOAD16(W16_SET(ls + 2), wcur, ls + 2);
                    else MICA_BLOAD16(W16_SET(ls + 2 - NS), wnxt, ls + 2 - NS);
                } else {
                    if (ls + 1 < NS) MICA_BLOAD16(W16_SET(ls + 1), wcur, ls + 1);
                    else MICA_BLOAD16(W16_SET(0), wnxt, 0);
                }
                // in flight and NEWER than this step's fragments: the NCT loads just issued plus the previous step's slab DMAs
                // (TWOAHEAD: the loads of the previous and of this step, and the DMAs of the two steps before this one - at the head of a
                // chunk those of the previous chunk's last steps)
                const int newer = TWOAHEAD ? 2 * NCT + W16_NDMA((ls + NS - 1) % NS) + W16_NDMA((ls + NS - 2) % NS)
                                           : NCT + (ls >= 1 ? W16_NDMA(ls - 1) : 0);
#define W16_WAIT(N) do { if (NCT == 4) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bc[0]), "+v"(bc[1]), "+v"(bc[NCT - 2]), "+v"(bc[NCT - 1])); \
                         else asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bc[0]), "+v"(bc[1])); } while (0)
                if (newer == 2) W16_WAIT(2);
                else if (newer == 3) W16_WAIT(3);
                else if (newer == 4) W16_WAIT(4);
                else if (newer == 5) W16_WAIT(5);
                else if (newer == 6) W16_WAIT(6);
                else if (newer == 7) W16_WAIT(7);
                else if (newer == 8) W16_WAIT(8);
                else if (newer == 9) W16_WAIT(9);
                else if (newer == 10) W16_WAIT(10);
                else if (newer == 11) W16_WAIT(11);
                else if (newer == 12) W16_WAIT(12);
                else W16_WAIT(13);
#undef W16_WAIT
                static_assert(NCT + W16_NDMA(0) <= 9 || NCT + W16_NDMA(0) == 11 || NCT + W16_NDMA(0) == 13, "wait immediates above");
                static_assert(NCT + W16_NDMA(1) <= 9 || NCT + W16_NDMA(1) == 11 || NCT + W16_NDMA(1) == 13, "wait immediates above");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < W16_NDMA(ls); ++q) MICA_SLAB_DMA(nsrc, nxt_off, W16_DMA0(ls) + q, org);
                const char* ab_nxt = ab_cur;
                if (ls + 1 < NS) ab_nxt = W16_ABASE(ls + 1);
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    // Two waves share a SIMD (wave w and w + 4, i.e. the two groups).  The MFMA arbiter is strictly "highest
                    // priority, then oldest": left alone, group 0 runs at its solo speed (about 60 % of the pipe, the rest
                    // are its own waits), group 1 only fills the gaps and then finishes ALONE at that same 60 %.  Group 1
                    // therefore holds priority 1 and group 0 alternates between 2 and 0 so that each is the gap filler for a
                    // share of the chunk and both reach the chunk barrier together with the pipe contended throughout.
                    if (f == 0) {
                        if (wn == 0) {
                            if (W16_XHI(ls)) asm volatile("s_setprio 2"); else asm volatile("s_setprio 0");
                        } else if (ls == 0) {
                            asm volatile("s_setprio 1");
                        }
                    }
                    const int fi = ls * 8 + f;                  // fragment index within the chunk; lives in ar[fi % (AD + 1)]
                    if (f + AD < 8) ar[(fi + AD) % (AD + 1)] = W16_AFRAG(ab_cur, f + AD);
                    else if (ls + 1 < NS) ar[(fi + AD) % (AD + 1)] = W16_AFRAG(ab_nxt, f + AD - 8);
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
                        if (f < MICA_EXP_ROWFRAGS)
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[f][c]) : "v"(ar[fi % (AD + 1)]), "v"(bc[c]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                ab_cur = ab_nxt;
            }
#undef W16_ABASE
#undef W16_AFRAG
            // the slab DMAs of this chunk are older than the NCT weight loads still wanted in flight
            if (NCT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __syncthreads();
            par ^= 1;
            wcur = wnxt;
        }
        asm volatile("s_setprio 0");
        // the next item's first weight fragments were requested a step ago: retire them here, so that the compiler may
        // move their registers freely through the epilogue and the loop back edge (it cannot see that they were in flight)
        if (NCT == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[W16_SET(0)][0]), "+v"(bq[W16_SET(0)][1]), "+v"(bq[W16_SET(0)][NCT - 2]), "+v"(bq[W16_SET(0)][NCT - 1]));
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[W16_SET(0)][0]), "+v"(bq[W16_SET(0)][1]));
        if constexpr (TWOAHEAD) {       // ... and those of its second step
            if (NCT == 4) asm volatile("" : "+v"(bq[W16_SET(1)][0]), "+v"(bq[W16_SET(1)][1]), "+v"(bq[W16_SET(1)][NCT - 2]), "+v"(bq[W16_SET(1)][NCT - 1]));
            else asm volatile("" : "+v"(bq[W16_SET(1)][0]), "+v"(bq[W16_SET(1)][1]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU/LDS readers: the hazard the compiler cannot see through the asm

        // ---- output transform through the idle slab buffer (the other one already holds the next item's first chunk) ----
        // four passes of CP columns; transform region T = [z 4][p 4][row 32 = y*8+pair][CP + 4 (padded)] floats.
        //   BN = 128: pass = (32-column half j, wave group wq): that group writes its accumulators to T.
        //   BN = 64 : pass = column tile; group 1 first hands its partial sums to group 0 through a 32 KB exchange area in
        //             front of T ([p 4][f 8][lane 64] float4), group 0 adds them and writes T.
        // Then all eight waves finish: wave -> (z plane, half of the CP columns), lane -> (row, 4-channel group).
        {
            constexpr int CP = SPLIT ? 16 : 32, RS = CP + 4, REG = 32 * RS;
            constexpr int CG = CP / 8, RPI = 64 / CG, ITS = 32 / RPI;        // 4-channel groups per row, rows per iteration
            char* ebuf = smem + (par ^ 1) * G::CH_BYTES;
            float4* xp = reinterpret_cast<float4*>(ebuf);                    // BN = 64: partial-sum exchange
            float* xs = reinterpret_cast<float*>(ebuf + (SPLIT ? 32768 : 0));
            static_assert((SPLIT ? 32768 : 0) + 16 * REG * 4 <= G::CH_BYTES, "epilogue fits one slab buffer");
            const int ib = cur.b, inb = cur.nb, itile = cur.tile;
            const int tx = itile % ntx, ty = (itile / ntx) % nty, tz = itile / (ntx * nty);
            const int nnitem = nitem + per_xcd;
            const Item nn = decode(nnitem < it_end ? nnitem : (has_next ? nitem : item));     // the item after next, for the next round
            const int fz = wave & 3, fch = wave >> 2;            // finishing role: z plane, column half
            const int frow = lane / CG, fcg = lane % CG;
            const int P = (items_per_b / nnb) * 4;
#pragma unroll
            for (int pass = 0; pass < MICA_EXP_EPI_PASSES(SPLIT ? NCT : 4); ++pass) {
                const int c0 = SPLIT ? pass : (pass >> 1) * 2;       // first column tile of the pass
                const int wq = SPLIT ? 0 : (pass & 1);               // the group that writes T
                if (SPLIT) {
                    if (wn == 1) {
#pragma unroll
                        for (int f = 0; f < 8; ++f)
                            xp[(wp * 8 + f) * 64 + lane] = make_float4(acc[f][c0][0], acc[f][c0][1], acc[f][c0][2], acc[f][c0][3]);
                    }
                    __syncthreads();
                }
                if (wn == wq) {
                    // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
                    for (int f = 0; f < 8; ++f) {
                        float* dst = xs + ((f >> 1) * 4 + wp) * REG;
                        float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (SPLIT) add = xp[(wp * 8 + f) * 64 + lane];
                        const float addv[4] = {add.x, add.y, add.z, add.w};
#pragma unroll
                        for (int c = 0; c < CP / 16; ++c)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int r = lg * 4 + i;                    // row in the 16-row fragment: pair = r&7, y = r>>3
                                const int r32 = ((f & 1) * 2 + (r >> 3)) * 8 + (r & 7);
                                dst[r32 * RS + c * 16 + lr] = acc[f][c0 + c][i] + addv[i];
                            }
                    }
                }
                __syncthreads();
                {
                    const float* src = xs + (fz * 4) * REG + fch * (CP / 2) + fcg * 4;
                    const int n0 = inb * BN + (SPLIT ? pass * 16 : wq * WNC + (pass >> 1) * 32) + fch * (CP / 2) + fcg * 4;
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (bias) bv = *reinterpret_cast<const float4*>(bias + n0);
                    const int gz = tz * 4 + fz;
                    // statistics: sums of (v - shift) and (v - shift)^2 with ONE shift per channel for the whole wave - the
                    // tile's first voxel (row 0 is inside the volume whenever this z plane is) - so that the row lanes
                    // combine by plain adds in a fixed order; (count, mean, M2) are formed once per channel at the end
                    float sn = 0.f;
                    float sk[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int it = 0; it < ITS; ++it) {
                        const int r = it * RPI + frow;
                        const float4 m0 = *reinterpret_cast<const float4*>(src + 0 * REG + r * RS);
                        const float4 m1 = *reinterpret_cast<const float4*>(src + 1 * REG + r * RS);
                        const float4 m2 = *reinterpret_cast<const float4*>(src + 2 * REG + r * RS);
                        const float4 m3 = *reinterpret_cast<const float4*>(src + 3 * REG + r * RS);
                        const int gx = (tx * 8 + (r & 7)) * 2, gy = ty * 4 + (r >> 3);
                        float ve[4], vo[4];
                        ve[0] = (m0.x + m1.x + m2.x) * out_scale + bv.x; vo[0] = (m1.x - m2.x - m3.x) * out_scale + bv.x;
                        ve[1] = (m0.y + m1.y + m2.y) * out_scale + bv.y; vo[1] = (m1.y - m2.y - m3.y) * out_scale + bv.y;
                        ve[2] = (m0.z + m1.z + m2.z) * out_scale + bv.z; vo[2] = (m1.z - m2.z - m3.z) * out_scale + bv.z;
                        ve[3] = (m0.w + m1.w + m2.w) * out_scale + bv.w; vo[3] = (m1.w - m2.w - m3.w) * out_scale + bv.w;
                        if (it == 0) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) sk[c] = __shfl(ve[c], fcg);      // lane fcg holds row 0 of this channel group
                        }
                        const bool in = gy < d.H && gz < d.D && gx < d.W, odd = in && gx + 1 < d.W;
                        if (in) {
                            // (blocks are 32 channels: shifts instead of a per-lane division in a kernel that has no registers to spare)
                            float* o = out + (int64_t)ib * V * cout + (BN != 32 && ocb != cout ? (int64_t)(n0 >> 5) * V * 32 + (n0 & 31) : (int64_t)n0) +
                                       ((int64_t)(gz * d.H + gy) * d.W + gx) * (BN != 32 ? ocb : cout);
                            *reinterpret_cast<float4*>(o) = make_float4(ve[0], ve[1], ve[2], ve[3]);
                            if (odd) *reinterpret_cast<float4*>(o + (BN != 32 ? ocb : cout)) = make_float4(vo[0], vo[1], vo[2], vo[3]);
                        }
                        const float we = in ? 1.f : 0.f, wo = odd ? 1.f : 0.f;
                        sn += we + wo;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float t = (ve[c] - sk[c]) * we, u = (vo[c] - sk[c]) * wo;
                            s1[c] += t + u;
                            s2[c] = fmaf(t, t, fmaf(u, u, s2[c]));
                        }
                    }
                    if (stats_ws) {
                        // lanes of one 4-channel group sit CG apart: inside a row of 16 lanes they combine with DPP rotations
                        // (VALU rate), the four rows with two cross-row exchanges
#define W16_ROR_ADD(x, n) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, false))
                        if (CG == 2) {
                            W16_ROR_ADD(sn, 2);
#pragma unroll
                            for (int c = 0; c < 4; ++c) { W16_ROR_ADD(s1[c], 2); W16_ROR_ADD(s2[c], 2); }
                        }
                        W16_ROR_ADD(sn, 4);
                        W16_ROR_ADD(sn, 8);
#pragma unroll
                        for (int c = 0; c < 4; ++c) { W16_ROR_ADD(s1[c], 4); W16_ROR_ADD(s2[c], 4); W16_ROR_ADD(s1[c], 8); W16_ROR_ADD(s2[c], 8); }
#undef W16_ROR_ADD
#pragma unroll
                        for (int off = 16; off < 64; off <<= 1) {
                            sn += __shfl_xor(sn, off);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                s1[c] += __shfl_xor(s1[c], off);
                                s2[c] += __shfl_xor(s2[c], off);
                            }
                        }
                        if (frow == 0) {
                            float* wsp = stats_ws + (((int64_t)ib * P + itile * 4 + fz) * cout + n0) * 3;
                            const float inv = sn > 0.f ? 1.f / sn : 0.f;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                wsp[c * 3] = sn;
                                wsp[c * 3 + 1] = sn > 0.f ? sk[c] + s1[c] * inv : 0.f;
                                wsp[c * 3 + 2] = sn > 0.f ? fmaxf(s2[c] - s1[c] * s1[c] * inv, 0.f) : 0.f;
                            }
                        }
                    }
                }
                if (!SPLIT) __syncthreads();     // BN = 64: the next pass's first barrier already orders T reads before T writes
            }
            if (SPLIT) __syncthreads();
            if (!has_next) break;
            item = nitem;
            nitem = nnitem;
            cur = nxt;
            nxt = nn;
        }
    }
#undef MICA_BLOAD16
#undef MICA_SLAB_DMA
    // nothing may still be in flight towards this workgroup's registers or LDS when it ends
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[W16_SET(0)][0]), "+v"(bq[W16_SET(0)][1]), "+v"(bq[W16_SET(0)][NCT - 2]), "+v"(bq[W16_SET(0)][NCT - 1]));
    if constexpr (TWOAHEAD) asm volatile("" : "+v"(bq[W16_SET(1)][0]), "+v"(bq[W16_SET(1)][1]), "+v"(bq[W16_SET(1)][NCT - 2]), "+v"(bq[W16_SET(1)][NCT - 1]));
#undef W16_SET
#undef W16_NDMA
#undef W16_DMA0
#undef W16_XHI
#undef W16_TAP
#undef W16_AOFF
#undef W16_ADELTA
#undef W16_WOFF
#undef W16_WDELTA
#undef W16_SEL
}

// channel block of a layer: every Cout is a multiple of 32 - blocks of 128 channels when possible, else 64, else 32
static int wino16_block(int cout) { return cout % 128 == 0 ? 128 : cout % 64 == 0 ? 64 : 32; }

static int launch_conv_wino16(const ConvSrcs& s, const _Float16* wpk, int64_t wpk_bstride, const float* bias, float out_scale,
                              float* out, int B, Dims d, int cout, float* stats_ws, hipStream_t st, int out_cblk) {
    int total = 0;
    for (int i = 0; i < s.n; ++i) total += s.chunks[i];
    const int bn = wino16_block(cout);
    // the blocked raw layout has 32-channel blocks; only conv3 layers write it (Cout = 64, 128: never the 32-channel variant)
    const int ocb = out_cblk == 32 && cout % 32 == 0 && bn != 32 ? 32 : cout;
    if (out_cblk != 0 && ocb == cout) { refuse_launch("conv_wino16: blocked output needs Cout a multiple of 64"); return 0; }
    int ntx = (d.W + 15) / 16, nty = (d.H + 3) / 4, ntz = (d.D + 3) / 4, nnb = cout / bn;
    size_t lds = 2 * GeoW::CH_BYTES;
    static PerDeviceOnce once;
    static int cus_of[64] = {0};
    const int dev = once.run([&](int dv) {
        (void)hipFuncSetAttribute((const void*)conv_wino16_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_wino16_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_wino16_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipDeviceProp_t prop;
        int c = 0;
        if (hipGetDeviceProperties(&prop, dv) == hipSuccess) c = prop.multiProcessorCount;
        if (c < 8) c = 256;
        cus_of[dv] = c & ~7;
    });
    const int cus = cus_of[dev];
    // one persistent workgroup per CU (LDS admits no more), a multiple of eight so that every XCD gets the same number
    const int items_per_b = ntx * nty * ntz * nnb, total_items = items_per_b * B;
    const int nwg = total_items >= cus ? cus : ((total_items + 7) / 8) * 8;
    if (bn == 128)
        hipLaunchKernelGGL((conv_wino16_kernel<128>), dim3(nwg), dim3(512), lds, st, s, wpk, wpk_bstride, bias, out_scale, out, d,
                           cout, total, ntx, nty, nnb, items_per_b, total_items, stats_ws, ocb);
    else if (bn == 64)
        hipLaunchKernelGGL((conv_wino16_kernel<64>), dim3(nwg), dim3(512), lds, st, s, wpk, wpk_bstride, bias, out_scale, out, d,
                           cout, total, ntx, nty, nnb, items_per_b, total_items, stats_ws, ocb);
    else
        hipLaunchKernelGGL((conv_wino16_kernel<32>), dim3(nwg), dim3(512), lds, st, s, wpk, wpk_bstride, bias, out_scale, out, d,
                           cout, total, ntx, nty, nnb, items_per_b, total_items, stats_ws, ocb);
    return ntx * nty * ntz * 4;
}

// Returns the number of statistics partials per (tile, channel) written to stats_ws (when non-null):
// f32 [B][P][cout][3] = (count, mean, M2), to be merged by launch_stats_finalize.
int launch_conv_wino(const ConvSrcs& s, const _Float16* wpk, int64_t wpk_bstride, const float* bias, float out_scale,
                     float* out, int B, Dims d, int cout, float* stats_ws, hipStream_t st, int out_cblk) {
    return launch_conv_wino16(s, wpk, wpk_bstride, bias, out_scale, out, B, d, cout, stats_ws, st, out_cblk);
}

// weights for conv_wino16: [B][nb = Cout/bn][chunk][pair-step 5][p 4][unit 8][bn][8] halves (bn = 128 or 64); unit u: 0,1 = hi
// of tap t (k-half 0,1), 2,3 = hi of tap t' = t+1, 4,5 = lo of tap t, 6,7 = lo of tap t'; pair-step 4 is tap 8 alone (units
// 2,3,6,7 zero)
__global__ void pack_weights_wino16_kernel(const float* __restrict__ w, int cout, int cin, Segs sg, int total_chunks,
                                           const float* __restrict__ cin_scale, float mul, _Float16* __restrict__ wpk,
                                           int64_t per_b, int bn) {
    const int b = blockIdx.y;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [nb][chunk][ps][p][unit][n in block]
    int64_t total = (int64_t)total_chunks * 5 * 4 * 8 * cout;
    if (e >= total) return;
    int nl = e % bn;
    int u = (e / bn) & 7;
    int pp = (e / (bn * 8)) & 3;
    int ps = (e / (bn * 32)) % 5;
    int gch = (e / (bn * 160)) % total_chunks;
    int n = (int)(e / ((int64_t)bn * 160 * total_chunks)) * bn + nl;
    const int kind = u >> 2, second = (u >> 1) & 1, kh = u & 1;
    const int tap = 2 * ps + second;
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int kp = gch * 16 + kh * 8 + j;
        int ci = -1, accp = 0, accc = 0;
        for (int si = 0; si < sg.n; ++si) {
            if (kp >= accp && kp < accp + sg.cp[si]) {
                int local = kp - accp;
                if (local < sg.c[si]) ci = accc + local;
            }
            accp += sg.cp[si];
            accc += sg.c[si];
        }
        float v = 0.f;
        if (ci >= 0 && tap < 9) {
            const float* g = w + ((int64_t)n * cin + ci) * 27 + tap * 3;
            float g0 = g[0], g1 = g[1], g2 = g[2];
            float uu = pp == 0 ? g0 : pp == 1 ? 0.5f * (g0 + g1 + g2) : pp == 2 ? 0.5f * (g0 - g1 + g2) : g2;
            v = uu * mul;
            if (cin_scale) v *= cin_scale[(int64_t)b * cin + ci];
        }
        _Float16 hi = (_Float16)v;
        _Float16 lo = mica_lo_half(v - (float)hi);
        o[j] = kind ? lo : hi;
    }
    *reinterpret_cast<half8*>(wpk + (int64_t)b * per_b + e * 8) = o;
}

int64_t packed_weight_halves_wino(int cout, int total_chunks) {
    return (int64_t)total_chunks * 5 * 32 * cout * 8;
}

void launch_pack_weights_wino(const float* w, int cout, int cin, const int* h_seg_c, const int* h_seg_cp, int nseg,
                              const float* cin_scale, int B, float cout_scale, float wscale, _Float16* wpk, hipStream_t st) {
    Segs sg;
    sg.n = nseg;
    int total_chunks = 0;
    for (int i = 0; i < nseg; ++i) {
        sg.c[i] = h_seg_c[i];
        sg.cp[i] = h_seg_cp[i];
        total_chunks += h_seg_cp[i] / 16;
    }
    int64_t total16 = (int64_t)total_chunks * 5 * 32 * cout;
    dim3 grid16((unsigned)((total16 + 255) / 256), B);
    hipLaunchKernelGGL(pack_weights_wino16_kernel, grid16, dim3(256), 0, st, w, cout, cin, sg, total_chunks, cin_scale,
                       cout_scale * wscale, wpk, packed_weight_halves_wino(cout, total_chunks), wino16_block(cout));
}

// ------------------------------------------------------------------------------------------------
// Weight packing: torch [Cout][Cin][k][k][k] f32 -> [B][chunk][tap][q][Cout][8] halves, q = kind*2+khalf.
// Input channels are laid out as the concatenation of the conv's sources, each padded to 16.
// ------------------------------------------------------------------------------------------------

__global__ void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int nt, Segs sg,
                                    int total_chunks, const float* __restrict__ cin_scale, float mul,
                                    _Float16* __restrict__ wpk, int64_t per_b) {
    const int b = blockIdx.y;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [chunk][tap][q][n]
    int64_t total = (int64_t)nt * total_chunks * 4 * cout;
    if (e >= total) return;
    int n = e % cout;
    int q = (e / cout) & 3;
    int tap = (e / ((int64_t)cout * 4)) % nt;
    int gch = e / ((int64_t)cout * 4 * nt);
    int kind = q >> 1, kh = q & 1;
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int kp = gch * 16 + kh * 8 + j;   // padded concat channel
        int ci = -1, accp = 0, accc = 0;
        for (int s = 0; s < sg.n; ++s) {
            if (kp >= accp && kp < accp + sg.cp[s]) {
                int local = kp - accp;
                if (local < sg.c[s]) ci = accc + local;
            }
            accp += sg.cp[s];
            accc += sg.c[s];
        }
        float v = 0.f;
        if (ci >= 0) {
            v = w[((int64_t)n * cin + ci) * nt + tap] * mul;
            if (cin_scale) v *= cin_scale[(int64_t)b * cin + ci];
        }
        _Float16 hi = (_Float16)v;
        _Float16 lo = mica_lo_half(v - (float)hi);
        o[j] = kind ? lo : hi;
    }
    *reinterpret_cast<half8*>(wpk + (int64_t)b * per_b + e * 8) = o;
}

int64_t packed_weight_halves(int cout, int ksize, int total_chunks) {
    return (int64_t)ksize * ksize * ksize * total_chunks * 4 * cout * 8;
}

void launch_pack_weights(const float* w, int cout, int cin, int ksize, const int* h_seg_c, const int* h_seg_cp,
                         int nseg, const float* cin_scale, int B, float cout_scale, float wscale, _Float16* wpk,
                         hipStream_t st) {
    Segs sg;
    sg.n = nseg;
    int total_chunks = 0;
    for (int i = 0; i < nseg; ++i) {
        sg.c[i] = h_seg_c[i];
        sg.cp[i] = h_seg_cp[i];
        total_chunks += h_seg_cp[i] / 16;
    }
    int nt = ksize * ksize * ksize;
    int64_t total = (int64_t)nt * total_chunks * 4 * cout;
    dim3 grid((unsigned)((total + 255) / 256), B);
    hipLaunchKernelGGL(pack_weights_kernel, grid, dim3(256), 0, st, w, cout, cin, nt, sg, total_chunks, cin_scale,
                       cout_scale * wscale, wpk, packed_weight_halves(cout, ksize, total_chunks));
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3x3 (model.py:80) with the producer's InstanceNorm+ReLU and the SE gate applied on load:
//   y = scale[c] * relu((x - mean[c]) * rstd[c]);  out = sum_tap w[tap][c] * y(shifted, zero padded) + bias
// HBM-bound (8 B/voxel/channel algorithmic); the 27 shifted reads are served by L1/L2.
// ------------------------------------------------------------------------------------------------
// One workgroup = 16 channels x a 16(x) x 16(y) column of the tile, marching along z with a ring of three
// halo'd z planes in LDS (18 x 18 voxels x 16 channels x 4 B = 20.7 KB each): every input element is read from HBM
// once per column (x/y halo 1.27x, no z re-reads), normalised once when it enters the ring, and the 27 taps
// come from LDS.  Thread = (channel quad, x, 4 consecutive y) with a sliding window along y.
constexpr int DW_X = 16;
constexpr int DW_LX = DW_X + 2;
// YO = y outputs per thread: the column is 16(x) x 4*YO(y); YO = 2 doubles the number of workgroups for small C * batch
// CQ = channel quads per workgroup (64 * CQ threads): 4 = 16 channels, two workgroups per CU; 8 = 32 channels = one full
// 128-B line per voxel, one workgroup of eight waves per CU.  With 16 channels every access is a 64-B piece of a 1-KB
// voxel row and the kernel saturated at ~4.5 TB/s of actual traffic; prep-style contiguous streams reach 5.4.
// NYQ = y groups of threads: the column is 16(x) x NYQ*YO(y) and the workgroup 16 * CQ * NYQ threads; <2, 8, 8> covers the same 16 x 16 x 32
// column as <4, 8, 4> with 16 waves of two outputs per thread instead of 8 waves of four (half the registers, four waves per SIMD)
template <int YO, int CQ, int NYQ = 4> struct DwGeo {
    static constexpr int Y = NYQ * YO, LY = Y + 2, DWC = 4 * CQ, NT = 16 * CQ * NYQ;
    static constexpr int PLANE = DW_LX * LY * DWC;     // floats per ring plane
};

template <int YO, int CQ, int NYQ = 4>
__global__ __launch_bounds__(16 * CQ * NYQ, CQ == 4 ? 2 : 1) void depthwise_kernel(const float* __restrict__ x, Dims d, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ scale, const float* __restrict__ w27,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        float* __restrict__ stats_ws, float* __restrict__ gap_ws, int ntx, int nty, int vs) {
    // vs: floats between consecutive voxels of a channel slab in x and out - C for plain NDHWC, the channel block (32) for the
    // blocked raw layout [C / vs][V][vs] (common.h), in which a 32-channel slab is ONE contiguous run of memory
    extern __shared__ __attribute__((aligned(16))) float ring[];   // [3][LY][LX][16] ; reused for the statistics merge
    using Geo = DwGeo<YO, CQ, NYQ>;
    constexpr int DW_Y = Geo::Y, DW_LY = Geo::LY, DW_PLANE = Geo::PLANE, DW_C = 4 * CQ, NT = Geo::NT, RL = NT / CQ;   // RL threads share a channel quad
    const int b = blockIdx.y;
    const int V = d.D * d.H * d.W;
    const int tid = threadIdx.x;
    // block order: channel slab outermost, XCD-contiguous (neighbouring columns share halos in one L2)
    int id = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) id = (id & 7) * (nwg >> 3) + (id >> 3);
    const int ncol = ntx * nty;
    const int cs = id / ncol, col = id - cs * ncol;
    const int tx = col % ntx, ty = col / ntx;
    const int x0 = tx * DW_X, y0 = ty * DW_Y, c0 = cs * DW_C;

    const int cq = tid % CQ, xi = (tid / CQ) & 15, yq = tid / (CQ * 16);     // compute role: 4 channels, x, YO y outputs
    const int c = c0 + cq * 4;
    float* wl = ring + 3 * DW_PLANE;                                  // [27][16] weights of this channel slab
    for (int e = tid; e < 27 * DW_C; e += NT) wl[e] = w27[(e / DW_C) * C + c0 + (e % DW_C)];
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);

    // load role: plane elements (voxel, channel quad): 324 voxels x 4 quads = 1296 float4 per plane, <= 6 per thread.
    // Planes are fetched RAW into registers THREE steps ahead (three register sets): at 2 workgroups per CU one plane in
    // flight is 41 KB per CU, which at 2.5 us of loaded HBM latency caps the chip at 4.2 TB/s of actual traffic (Little's
    // law) - that was the limit of the one-ahead version (2.6 -> 3.1 TB/s algorithmic).  A plane is normalised (IN-apply +
    // ReLU + gate) when it is written to the ring slot of the plane that has just been retired.
    constexpr int NE = (DW_LX * DW_LY * CQ + NT - 1) / NT;
    float4 preA[NE], preB[NE], preC[NE];
    // the channel quad of a thread's elements is (tid + NT k) % CQ = tid % CQ: one set of norm constants
    const int lq = tid % CQ;
    float4 nm = make_float4(0, 0, 0, 0), nr = make_float4(1, 1, 1, 1), ns = make_float4(1, 1, 1, 1);
    if (mean) { nm = *reinterpret_cast<const float4*>(mean + (int64_t)b * C + c0 + lq * 4); nr = *reinterpret_cast<const float4*>(rstd + (int64_t)b * C + c0 + lq * 4); }
    if (scale) ns = *reinterpret_cast<const float4*>(scale + (int64_t)b * C + c0 + lq * 4);
    // Everything about a thread's NE plane elements that does not depend on z is worked out once: the float offset of the
    // (clamped) voxel inside a z plane, whether the element lies inside the volume / the plane image / the block's own voxels.
    // (The per-plane versions of these index computations were ~1200 VALU cycles per plane and wave - a sixth of the kernel.)
    int goff[NE];
    unsigned okm = 0, inm = 0, valm = 0;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = tid + NT * k;
        const int v = e / CQ;
        const int lx = v % DW_LX, ly = v / DW_LX;
        const int cx = min(max(x0 + lx - 1, 0), d.W - 1), cy = min(max(y0 + ly - 1, 0), d.H - 1);
        goff[k] = (cy * d.W + cx) * vs + lq * 4;
        const bool val = e < DW_LX * DW_LY * CQ;
        if ((unsigned)(x0 + lx - 1) < (unsigned)d.W && (unsigned)(y0 + ly - 1) < (unsigned)d.H) okm |= 1u << k;
        if (val) valm |= 1u << k;
        if (val && lx >= 1 && lx <= DW_X && ly >= 1 && ly <= DW_Y) inm |= 1u << k;
    }
    const int64_t cbase = (int64_t)b * V * C + (int64_t)(c0 / vs) * V * vs + (c0 % vs);      // first channel of the slab, voxel 0
    const float* xb = x + cbase;
    const int64_t zstride = (int64_t)d.H * d.W * vs;
    auto fetch_plane = [&](int gz, float4 (&pre)[NE]) {
        const float* pz = xb + (int64_t)min(max(gz, 0), d.D - 1) * zstride;
#pragma unroll
#ifdef MICA_DW_NOFETCH
        for (int k = 0; k < NE; ++k) pre[k] = make_float4(1.f, 2.f, 3.f, (float)gz);       // ablation: no HBM reads
        (void)pz;
#else
        for (int k = 0; k < NE; ++k) pre[k] = *reinterpret_cast<const float4*>(pz + goff[k]);
#endif
    };
    // gap_ws: the global average pool of the NORMALISED input (the SE gate's input, model.py:256) rides along: every voxel is an
    // interior element of exactly one block
    float4 gsum = make_float4(0, 0, 0, 0);
    auto store_plane = [&](int slot, int gz, float4 (&pre)[NE]) {
        float* dst = ring + slot * DW_PLANE + tid * 4;              // element e = v * CQ + lq sits at float e * 4
        const unsigned zm = (unsigned)gz < (unsigned)d.D ? okm : 0u;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            float4 t = pre[k];
            if (mean) {
                t.x = fmaxf((t.x - nm.x) * nr.x, 0.f) * ns.x; t.y = fmaxf((t.y - nm.y) * nr.y, 0.f) * ns.y;
                t.z = fmaxf((t.z - nm.z) * nr.z, 0.f) * ns.z; t.w = fmaxf((t.w - nm.w) * nr.w, 0.f) * ns.w;
            } else { t.x *= ns.x; t.y *= ns.y; t.z *= ns.z; t.w *= ns.w; }
            if (!(zm >> k & 1u)) t = make_float4(0, 0, 0, 0);
            if (valm >> k & 1u) {
                *reinterpret_cast<float4*>(dst + NT * 4 * k) = t;
                if (inm >> k & 1u) { gsum.x += t.x; gsum.y += t.y; gsum.z += t.z; gsum.w += t.w; }
            }
        }
    };

    float sn = 0.f;
    float4 sk = make_float4(0, 0, 0, 0), s1 = sk, s2 = sk;
    // The taps of output plane z in two parts: dz = 0 reads plane z - 1 (ring slot z % 3), dz = 1, 2 read planes z, z + 1.  Plane z + 2
    // replaces plane z - 1 in its slot as soon as every wave is through the first part, and the normalise-and-store of that plane then runs
    // beside the other two thirds of the taps instead of between two barriers of its own (round 5; the accumulation order is unchanged).
    float4 acc[YO];
    auto taps = [&](int z, int dz0, int dz1) {
#ifdef MICA_DW_NOCOMPUTE
        if (z < 0)          // ablation: no taps (the outputs are the bias)
#endif
#pragma unroll 1
        for (int dz = dz0; dz < dz1; ++dz) {
            const float* pl = ring + ((z + dz) % 3) * DW_PLANE + (xi * DW_C + cq * 4);
#pragma unroll 1
            for (int dx = 0; dx < 3; ++dx) {
                float4 win[YO + 2];
#pragma unroll
                for (int i = 0; i < YO + 2; ++i)
                    win[i] = *reinterpret_cast<const float4*>(pl + ((yq * YO + i) * DW_LX + dx) * DW_C);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float4 w = *reinterpret_cast<const float4*>(wl + ((dz * 3 + dy) * 3 + dx) * DW_C + cq * 4);
#pragma unroll
                    for (int i = 0; i < YO; ++i) {
                        acc[i].x = fmaf(w.x, win[i + dy].x, acc[i].x); acc[i].y = fmaf(w.y, win[i + dy].y, acc[i].y);
                        acc[i].z = fmaf(w.z, win[i + dy].z, acc[i].z); acc[i].w = fmaf(w.w, win[i + dy].w, acc[i].w);
                    }
                }
            }
        }
    };
    auto first_part = [&](int z) {
#pragma unroll
        for (int i = 0; i < YO; ++i) acc[i] = bv;
        taps(z, 0, 1);
    };
    auto second_part = [&](int z) {
        taps(z, 1, 3);
        const int gx = x0 + xi;
#pragma unroll
        for (int i = 0; i < YO; ++i) {
            const int gy = y0 + yq * YO + i;
            if (gx < d.W && gy < d.H) {
#ifdef MICA_DW_NT
                {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    f4v v_; v_.x = acc[i].x; v_.y = acc[i].y; v_.z = acc[i].z; v_.w = acc[i].w;
                    __builtin_nontemporal_store(v_, reinterpret_cast<f4v*>(out + cbase + ((int64_t)(z * d.H + gy) * d.W + gx) * vs + cq * 4));
                }
#else
                *reinterpret_cast<float4*>(out + cbase + ((int64_t)(z * d.H + gy) * d.W + gx) * vs + cq * 4) = acc[i];
#endif
                if (sn == 0.f) sk = acc[i];
                const float4 t = make_float4(acc[i].x - sk.x, acc[i].y - sk.y, acc[i].z - sk.z, acc[i].w - sk.w);
                s1.x += t.x; s1.y += t.y; s1.z += t.z; s1.w += t.w;
                s2.x = fmaf(t.x, t.x, s2.x); s2.y = fmaf(t.y, t.y, s2.y); s2.z = fmaf(t.z, t.z, s2.z); s2.w = fmaf(t.w, t.w, s2.w);
                sn += 1.f;
            }
        }
    };
#ifdef MICA_DW_CLOCKS
    long long tk[6] = {0, 0, 0, 0, 0, 0};
    const long long t_entry = __builtin_readcyclecounter();
#define DWCLK(i, stmt) do { const long long t0_ = __builtin_readcyclecounter(); stmt; tk[i] += __builtin_readcyclecounter() - t0_; } while (0)
#else
#define DWCLK(i, stmt) do { stmt; } while (0)
#endif
    fetch_plane(-1, preA); fetch_plane(0, preB); fetch_plane(1, preC);
    store_plane(0, -1, preA); store_plane(1, 0, preB); store_plane(2, 1, preC);
    fetch_plane(2, preA); fetch_plane(3, preB);
    __syncthreads();
#ifdef MICA_DW_CLOCKS
    const long long t_loop = __builtin_readcyclecounter();
#endif
    // invariant at the top of an iteration z (multiple of 3): preA holds plane z + 2 and preB plane z + 3 (in flight), the
    // ring holds z-1, z, z+1
    for (int z = 0; z < d.D; z += 3) {
        DWCLK(0, fetch_plane(z + 4, preC));
        DWCLK(1, first_part(z));
        DWCLK(2, __syncthreads());                    // plane z-1 (slot z % 3) is no longer read by anyone
        DWCLK(3, store_plane(z % 3, z + 2, preA));    // plane z+2 takes its place ...
        DWCLK(1, second_part(z));                     // ... while the taps on planes z, z+1 run
        // ONE barrier per plane: plane z+2 is first read in the second part of output plane z+1, i.e. behind the next barrier, which no
        // wave passes before every wave has finished this store; and the slot it overwrote was last read before the barrier above
        if (z + 1 >= d.D) break;
        DWCLK(0, fetch_plane(z + 5, preA));
        DWCLK(1, first_part(z + 1));
        DWCLK(2, __syncthreads());
        DWCLK(3, store_plane((z + 1) % 3, z + 3, preB));
        DWCLK(1, second_part(z + 1));
        if (z + 2 >= d.D) break;
        DWCLK(0, fetch_plane(z + 6, preB));
        DWCLK(1, first_part(z + 2));
        DWCLK(2, __syncthreads());
        DWCLK(3, store_plane((z + 2) % 3, z + 4, preC));
        DWCLK(1, second_part(z + 2));
    }
    __syncthreads();                                  // the ring is reused below: every wave is through its last taps
#ifdef MICA_DW_CLOCKS
    const long long t_done = __builtin_readcyclecounter();
#endif
    if (gap_ws) {
        // block sum per channel: the 64 threads that share a channel quad, fixed tree
        __syncthreads();
        float4* sg = reinterpret_cast<float4*>(ring);
        sg[tid] = gsum;
        __syncthreads();
        for (int off = RL / 2; off > 0; off >>= 1) {
            if (tid / CQ < off) {
                const float4 o = sg[tid + off * CQ];
                float4 a = sg[tid];
                a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
                sg[tid] = a;
            }
            __syncthreads();
        }
        if (tid / CQ == 0) *reinterpret_cast<float4*>(gap_ws + ((int64_t)b * ncol + col) * C + c0 + (tid % CQ) * 4) = sg[tid];
        __syncthreads();
    }
    if (!stats_ws) return;
    // block merge of (count, mean, M2): 64 threads (x, yq) share a channel quad
    float* shn = ring; float* shm = ring + 4 * NT; float* shq = ring + 8 * NT;
    const float kk[4] = {sk.x, sk.y, sk.z, sk.w}, a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float mean_ = 0.f, m2 = 0.f;
        if (sn > 0.f) { mean_ = kk[j] + a1[j] / sn; m2 = fmaxf(a2[j] - a1[j] * a1[j] / sn, 0.f); }
        shn[tid * 4 + j] = sn; shm[tid * 4 + j] = mean_; shq[tid * 4 + j] = m2;
    }
    __syncthreads();
    const int rl = tid / CQ;                       // 0..RL-1 within the channel quad
    for (int off = RL / 2; off > 0; off >>= 1) {
        if (rl < off) {
            const int o = tid + off * CQ;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float na = shn[tid * 4 + j], ma = shm[tid * 4 + j], qa = shq[tid * 4 + j];
                const float nb2 = shn[o * 4 + j], mb = shm[o * 4 + j], qb = shq[o * 4 + j];
                if (nb2 > 0.f) {
                    if (na == 0.f) { na = nb2; ma = mb; qa = qb; }
                    else { const float nn = na + nb2, dl = mb - ma; ma += dl * (nb2 / nn); qa += qb + dl * dl * (na * nb2 / nn); na = nn; }
                }
                shn[tid * 4 + j] = na; shm[tid * 4 + j] = ma; shq[tid * 4 + j] = qa;
            }
        }
        __syncthreads();
    }
    if (rl == 0) {
        float* wsp = stats_ws + (((int64_t)b * ncol + col) * C + c) * 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) { wsp[j * 3] = shn[tid * 4 + j]; wsp[j * 3 + 1] = shm[tid * 4 + j]; wsp[j * 3 + 2] = shq[tid * 4 + j]; }
    }
#ifdef MICA_DW_CLOCKS
    if ((tid & 63) == 0 && (tid >> 6) % 4 == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 77))
        printf("dw<%d,%d> C=%d blk %d wave %d: prologue %lld loop %lld epilogue %lld | per plane: fetch %lld compute %lld barA %lld store %lld barB %lld\n", YO, CQ, C,
               (int)blockIdx.x, tid >> 6, t_loop - t_entry, t_done - t_loop, (long long)__builtin_readcyclecounter() - t_done, tk[0] / d.D, tk[1] / d.D,
               tk[2] / d.D, tk[3] / d.D, tk[4] / d.D);
#endif
}

// Returns the number of statistics partials per (tile, channel) written to stats_ws (when non-null).
template <int YO, int CQ, int NYQ = 4>
static void launch_depthwise_t(const float* x, int B, Dims d, int C, const float* mean, const float* rstd, const float* scale,
                               const float* w27, const float* bias, float* out, float* stats_ws, float* gap_ws, int ntx, int nty, int vs, hipStream_t st) {
    using Gm = DwGeo<YO, CQ, NYQ>;
    const size_t lds = (3 * Gm::PLANE + 27 * Gm::DWC) * sizeof(float);
    static PerDeviceOnce once;
    once.run([&](int) { (void)hipFuncSetAttribute((const void*)depthwise_kernel<YO, CQ, NYQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    dim3 grid((unsigned)(ntx * nty * (C / Gm::DWC)), B);
    hipLaunchKernelGGL((depthwise_kernel<YO, CQ, NYQ>), grid, dim3(Gm::NT), lds, st, x, d, C, mean, rstd, scale, w27, bias, out, stats_ws, gap_ws, ntx, nty, vs);
}
int launch_depthwise(const float* x, int B, Dims d, int C, const float* mean, const float* rstd,
                     const float* scale, const float* w27, const float* bias, float* out, float* stats_ws, float* gap_ws, hipStream_t st, int cblk) {
    const int ntx = (d.W + DW_X - 1) / DW_X;
    const int vs = cblk > 0 && C % cblk == 0 && cblk % 32 == 0 ? cblk : C;      // a slab (16 or 32 channels) never straddles a block
    // The variant fixes the number and order of the statistics / pool partials of a tile, so it is chosen from the tile geometry
    // (C, H, W) alone - as if 8 tiles were in flight, the throughput case - never from the size of this call or the capacity of
    // the context: a tile's numbers do not depend on how many tiles share its call, nor on which engine computed them.
    constexpr int Bplan = 8;
    // 32-channel workgroups (full 128-B lines) when that still gives every CU at least two rounds of work
    if (C % 32 == 0 && (int64_t)ntx * ((d.H + 15) / 16) * (C / 32) * Bplan >= 256) {
        const int nty = (d.H + 15) / 16;
#ifdef MICA_EXP_DW16      // development A/B (tools/exp/dw16.sh): 16 waves of two outputs per thread - 0.559-0.567 against 0.568-0.574 of 8 TB/s
        launch_depthwise_t<2, 8, 8>(x, B, d, C, mean, rstd, scale, w27, bias, out, stats_ws, gap_ws, ntx, nty, vs, st);
#else
        launch_depthwise_t<4, 8>(x, B, d, C, mean, rstd, scale, w27, bias, out, stats_ws, gap_ws, ntx, nty, vs, st);
#endif
        return ntx * nty;
    }
    // enough workgroups to fill 256 CUs twice: halve the column height when C * batch is small
    const bool small = (int64_t)ntx * ((d.H + 15) / 16) * (C / 16) * Bplan < 1024;
    const int Y = small ? 8 : 16;
    const int nty = (d.H + Y - 1) / Y;
    if (small) launch_depthwise_t<2, 4>(x, B, d, C, mean, rstd, scale, w27, bias, out, stats_ws, gap_ws, ntx, nty, vs, st);
    else launch_depthwise_t<4, 4>(x, B, d, C, mean, rstd, scale, w27, bias, out, stats_ws, gap_ws, ntx, nty, vs, st);
    return ntx * nty;
}

// ------------------------------------------------------------------------------------------------
// Multi-scale stem (model.py:9-14,49-51): four Conv3d(1,32,k) with k = 3,5,7,9 on the density tile.
// This is the f32 VALU form (tile widths that are not multiples of 64; the production width runs on the matrix cores with the
// taps as the GEMM's K dimension, kernels_stem.hip): one LDS tile with halo 4 serves all
// four kernels; each thread keeps 2 voxels x 32 output channels in registers and the weights arrive
// as wave-uniform scalar loads (v_fma with an SGPR operand).  Writes the 128 channels straight in
// split format plus per-block channel sums for the attention gate's global average pool (model.py:21).
// Workgroup tile: 32(x) x 8(y) x 2(z) voxels.
// ------------------------------------------------------------------------------------------------
constexpr int ST_X = 32, ST_Y = 8, ST_Z = 2, ST_H = 4;
constexpr int ST_LX = ST_X + 2 * ST_H, ST_LY = ST_Y + 2 * ST_H, ST_LZ = ST_Z + 2 * ST_H;
int64_t stem_weight_floats() { return (27 + 125 + 343 + 729) * 32; }

template <int K>
__device__ __forceinline__ void stem_one(const float* __restrict__ tile, const float* __restrict__ wc, int lx, int ly0,
                                         int lz, float (&acc)[2][32]) {
    constexpr int R = K / 2;
    for (int dz = 0; dz < K; ++dz)
        for (int dx = 0; dx < K; ++dx) {
            float col[K + 1];
            const float* p = tile + ((lz + dz + ST_H - R) * ST_LY + (ly0 + ST_H - R)) * ST_LX + (lx + dx + ST_H - R);
#pragma unroll
            for (int i = 0; i < K + 1; ++i) col[i] = p[i * ST_LX];
#pragma unroll
            for (int dy = 0; dy < K; ++dy) {
                const float* wt = wc + ((dz * K + dy) * K + dx) * 32;
#pragma unroll
                for (int co = 0; co < 32; ++co) {
                    float wv = wt[co];
                    acc[0][co] = fmaf(wv, col[dy], acc[0][co]);
                    acc[1][co] = fmaf(wv, col[dy + 1], acc[1][co]);
                }
            }
        }
}

__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ map, Dims d,
                                                   const float* __restrict__ wstem, const float* __restrict__ bstem,
                                                   SplitView out, float* __restrict__ out_raw,
                                                   float* __restrict__ ws, int ntx, int nty, SplitEnc enc) {
    __shared__ float tile[ST_LZ * ST_LY * ST_LX];
    const float ascale = enc.ascale;
    int bad = 0;
    __shared__ float csum[4][128];   // per-wave partial channel sums (fixed summation order => deterministic)
    const int tid = threadIdx.x, b = blockIdx.y;
    const int V = d.D * d.H * d.W;
    const int t = blockIdx.x;
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    const int x0 = tx * ST_X, y0 = ty * ST_Y, z0 = tz * ST_Z;
    const float* mb = map + (int64_t)b * V;
    for (int i = tid; i < ST_LZ * ST_LY * ST_LX; i += 256) {
        int lx = i % ST_LX, ly = (i / ST_LX) % ST_LY, lz = i / (ST_LX * ST_LY);
        int gx = x0 + lx - ST_H, gy = y0 + ly - ST_H, gz = z0 + lz - ST_H;
        float v = 0.f;
        if ((unsigned)gx < (unsigned)d.W && (unsigned)gy < (unsigned)d.H && (unsigned)gz < (unsigned)d.D)
            v = mb[(int64_t)(gz * d.H + gy) * d.W + gx];
        tile[i] = v;
    }
    __syncthreads();
    const int lx = tid & 31, ly0 = ((tid >> 5) & 3) * 2, lz = tid >> 7;
    const int gx = x0 + lx, gz = z0 + lz;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        float acc[2][32];
#pragma unroll
        for (int co = 0; co < 32; ++co) { acc[0][co] = bstem[c * 32 + co]; acc[1][co] = acc[0][co]; }
        if (c == 0) stem_one<3>(tile, wstem, lx, ly0, lz, acc);
        else if (c == 1) stem_one<5>(tile, wstem + 27 * 32, lx, ly0, lz, acc);
        else if (c == 2) stem_one<7>(tile, wstem + (27 + 125) * 32, lx, ly0, lz, acc);
        else stem_one<9>(tile, wstem + (27 + 125 + 343) * 32, lx, ly0, lz, acc);
        float sums[32];
#pragma unroll
        for (int co = 0; co < 32; ++co) sums[co] = 0.f;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int gy = y0 + ly0 + v;
            const bool ok = gx < d.W && gy < d.H && gz < d.D;
            if (ok) {
                const int64_t vox = (int64_t)(gz * d.H + gy) * d.W + gx;
#pragma unroll
                for (int co = 0; co < 32; ++co) sums[co] += acc[v][co];
                if (out.p) {
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        half8 hi[2], lo[2];
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            float xs = acc[v][cc * 16 + j] * ascale;
                            if (!(fabsf(xs) <= F16_LIMIT)) {      // flag and saturate, like every other split encoder (kernels_elem.hip: split8)
                                bad |= (fabsf(acc[v][cc * 16 + j]) <= 3.0e38f) ? RANGE_OVERFLOW : RANGE_NONFINITE;
                                xs = fminf(fmaxf(xs, -F16_LIMIT), F16_LIMIT);
                            }
                            _Float16 h = (_Float16)xs;
                            hi[j >> 3][j & 7] = h;
                            lo[j >> 3][j & 7] = (_Float16)(xs - (float)h);
                        }
                        _Float16* dst = out.p + (((int64_t)b * out.chunks_total + out.chunk_off + c * 2 + cc) * V + vox) * 32;
                        *reinterpret_cast<half8*>(dst) = hi[0];
                        *reinterpret_cast<half8*>(dst + 8) = hi[1];
                        *reinterpret_cast<half8*>(dst + 16) = lo[0];
                        *reinterpret_cast<half8*>(dst + 24) = lo[1];
                    }
                }
                if (out_raw) {
                    float* dr = out_raw + ((int64_t)b * V + vox) * 128 + c * 32;
#pragma unroll
                    for (int co = 0; co < 32; co += 4)
                        *reinterpret_cast<float4*>(dr + co) = make_float4(acc[v][co], acc[v][co + 1], acc[v][co + 2], acc[v][co + 3]);
                }
            }
        }
        // block partial channel sums (wave shuffle reduce, then LDS)
#pragma unroll
        for (int co = 0; co < 32; ++co) {
            float sv = sums[co];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o);
            if ((tid & 63) == 0) csum[tid >> 6][c * 32 + co] = sv;
        }
    }
    if (bad && enc.err) atomicOr(enc.err + b, bad);
    __syncthreads();
    if (ws && tid < 128)
        ws[((int64_t)b * gridDim.x + blockIdx.x) * 128 + tid] = (csum[0][tid] + csum[1][tid]) + (csum[2][tid] + csum[3][tid]);
}

void launch_stem(const float* map, int B, Dims d, const float* wstem, const float* bstem, SplitView out,
                 float* out_raw, float* gap, float* ws, SplitEnc enc, hipStream_t st) {
    int ntx = (d.W + ST_X - 1) / ST_X, nty = (d.H + ST_Y - 1) / ST_Y, ntz = (d.D + ST_Z - 1) / ST_Z;
    dim3 grid(ntx * nty * ntz, B);
    hipLaunchKernelGGL(stem_kernel, grid, dim3(256), 0, st, map, d, wstem, bstem, out, out_raw, gap ? ws : nullptr, ntx, nty, enc);
    if (gap) launch_finalize_sum(ws, B, (int)grid.x, 128, 1.0f / (float)(d.D * d.H * d.W), gap, st);
}

}  // namespace mica

// Dense 3x3x3 convolution with Winograd F(4,3) along x for the layers where the whole-network numerics allow it (round 4).
//
// conv_wino16_kernel<128> (kernels_conv.hip, F(2,3) along x) is power-bound on the matrix pipes: the only lever on its time is
// fewer MFMAs per output.  F(4,3) needs 6 transform-domain products per 4 outputs instead of 4 per 2: 13.5 instead of 18
// MFMA-taps per output (1.33x fewer), and its operand is 1.5x the plain bytes instead of 2x.  Its transforms have constants up to
// 8, so a layer's rounding error is about 4x that of F(2,3); oracle/wino_network.py (whole-network emulation of exactly this
// arithmetic, profiles/r04_wino_network_numerics.txt) shows where that matters: F(4,3) EVERYWHERE doubles the network's rms
// distance from the float64 truth and breaks the 1e-4 bar, on all layers with Cout >= 128 it is borderline, but on the four
// 3x3x3 convs of encoder.2 (conv1 256->128, conv2 384->128, conv3 512->256, transition 256->512: 68 % of the network's FLOPs, K =
// 27 x 256..512 per output, late in the network) the result is indistinguishable from F(2,3) everywhere.  Those four layers run here.
// Replaces nn.Conv3d(k=3, padding=1) of reference models/model.py:107,115,122,142 (encoder.2).
//
//   points {0, a, -a, b, -b, inf} with a = 3/2, b = 2/3 (a b = 1; s = a^2 + b^2); per output quad (x = 4i .. 4i+3),
//   d_k = in(4i-1+k), k = 0..5 (zero outside the volume):
//     t0 = d0 - s d2 + d4                          u0 = g0                                    y0 = m0 + m1 + m2 + m3 + m4
//     t1,2 = (d4 - b^2 d2) +- (a d3 - b d1)        u1,2 = (g0 +- a g1 + a^2 g2) / N_a         y1 = a (m1 - m2) + b (m3 - m4)
//     t3,4 = (d4 - a^2 d2) +- (b d3 - a d1)        u3,4 = (g0 +- b g1 + b^2 g2) / N_b         y2 = a^2 (m1 + m2) + b^2 (m3 + m4)
//     t5 = d1 - s d3 + d5                          u5 = g2                                    y3 = a^3 (m1 - m2) + b^3 (m3 - m4) + m5
//     N_a = 2 a^2 (a^2 - b^2), N_b = 2 b^2 (b^2 - a^2);   m_p = sum over (dz, dy, cin) of t_p u_p
//   (the textbook points {0, +-1, +-2} make m5 four times the output's magnitude: 0.77e-6 rms error per layer against 0.55e-6 here,
//   common.h).  The producer passes (prep_wino43_kernel below, the F(4,3) epilogue of conv1x1_kernel) do the input transform in f32
//   and encode t * (ascale / 4) as f16 hi + lo (|t| <= 5.4 max|d|; powers of two, exact); the weight packer does the weight
//   transform in f64.
//
// Operand ("wino43") layout:  _Float16 [B][chunks][6 p][4 q][Vq][8],  Vq = D*H*ceil(W/4); q = hi|lo x channel half of the chunk.
//
// Kernel structure = conv_wino16_kernel<128> re-dimensioned (same MFMA shape v_mfma_f32_16x16x32_f16, same regrouping of the three
// split products into X / X' / Y steps over tap pairs, same persistent item walk, slab LDS-DMA with counted vmcnt, asm weight
// prefetch): workgroup = 12 waves = 6 positions x 2 channel halves; output tile 32(x) = 8 quads x 2(y) x 4(z); wave (p, wn) owns
// Winograd position p for all 64 (quad, y, z) rows and 64 channels = 4 row fragments (one per z: 8 quads x 2 y) x 4 column tiles
// = 64 accumulator VGPRs, three waves per SIMD.  LDS image per chunk: 4 planes (hi/lo x k-half) x [z 6][p 6][y 4][quad 8] 16-B
// slots = 73,728 B, double buffered.  Per chunk and workgroup 2,688 MFMAs instead of 3,584 for the same 256 voxels x 128
// channels; a weight fragment feeds 4 MFMAs (8 in the F(2,3) kernel: the weight stream from L2 is 1.5x per output).
//
// BN = 64 (round 5: the FPN's smooth convs 64 -> 64 and the heads' conv1 192 / 196 / 200 -> 64, reference models/model.py:165-174,210):
// as conv_wino16_kernel<64>, the two wave groups own the SAME 64 channels and SPLIT THE TAPS - group 0 taps 0..3 and the hi x hi /
// lo x hi products of tap 8 ("x"), group 1 taps 4..7 and tap 8's hi x lo product ("y") - seven steps per chunk each instead of
// fourteen; per-wave tile, operand reuse and weight bytes per MFMA stay those of the 128 variant.  One instruction stream serves
// both groups (tap offsets and weight offsets are per-group scalars); the partial sums meet in the epilogue: group 1 stores its
// accumulators into the transform tile, group 0 adds its own to them in place (one add per address: deterministic).  Oracle:
// oracle/wino_network.py f43s@late (profiles/r05_wino_late_numerics.txt): the rms distance from the float64 truth moves by < 1.5 %.
#include "common.h"
#include <cstdio>
#include <cstdlib>

namespace mica {

struct Segs43 { int c[MAX_SRC]; int cp[MAX_SRC]; int n; };

struct Geo43 {
    // output tile 32(x) = 8 quads x 2(y) x 4(z): a slab row is 8 quads = 128 contiguous bytes per plane in HBM (with 4 quads per
    // row - a 16 x 4 x 4 tile - every slab DMA touched half of each 128-byte line it named and the HBM slab stream cost 13 % of
    // the kernel; tools/exp/abl43.sh)
    static constexpr int TY = 2, TZ = 4;               // output rows / planes per tile
    static constexpr int SY = TY + 2, SZ = TZ + 2, NP = 6, QX = 8;
    static constexpr int PP = SY * QX;                 // 32 slots per (z, p)
    static constexpr int PZ = NP * PP;                 // 192 slots per z plane = 3 DMA instructions
    static constexpr int PLANE = SZ * PZ;              // 1152 slots per (hi/lo, k-half) plane
    static constexpr int NSLOT = 4 * PLANE;            // 4608
    static constexpr int CH_BYTES = NSLOT * 16;        // 73,728
    static constexpr int NDMA = NSLOT / 64;            // 72 one-KiB LDS-DMA instructions per chunk
    static constexpr int NW = 12;                      // waves per workgroup
    static constexpr int DPW = 6;                      // DMA instructions per wave and chunk
    static constexpr int BN = 128;
};
static_assert(Geo43::NSLOT % 64 == 0 && Geo43::NDMA == Geo43::NW * Geo43::DPW && Geo43::PZ == 192 && Geo43::PP == 32 && Geo43::QX == 8 &&
              Geo43::PLANE == 18 * 64, "slab DMA plan: the lane / scalar split of the source offsets in the kernel assumes this geometry");

typedef float floatx4w __attribute__((ext_vector_type(4)));
#ifdef MICA43_CLOCKS
// development: cycle stamps of one chunk (workgroup 8, its fourth item, chunk 10) per wave, read back by mica_debug_conv43 (tools/exp/clk43.py)
__device__ unsigned g_mica43_clk[12 * 48];
#define W43_STAMP(idx)                                                                                   \
    do {                                                                                                 \
        if (clk_on) {                                                                                    \
            const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();         /* (HW_REG_SHADER_CYCLES reads 0 on gfx950) */ \
            if (lane == 0) clk_lds[wave * 48 + (idx)] = t_;                                              \
        }                                                                                                \
    } while (0)
#else
#define W43_STAMP(idx) do {} while (0)
#endif
// development switches (timing experiments only, results are garbage; tools/exp/abl43.sh): -DMICA43_W_FIXED every weight fragment
// load hits the same 4 KB per wave (L1 instead of the L2 stream); -DMICA43_SLAB_FIXED the slab DMAs wrap into the first 4 MB of
// the operand (L2 instead of HBM); -DMICA43_NOEPI no output-transform passes
#ifdef MICA43_ALLFIX
#define MICA43_W_FIXED
#define MICA43_SLAB_FIXED
#define MICA43_NOEPI
#endif
#ifdef MICA43_W_FIXED
#define MICA43_WBASE(b, off) (wwave)
#else
#define MICA43_WBASE(b, off) ((b) + (off))
#endif
#ifdef MICA43_SLAB_FIXED
#define MICA43_SLABOFF(x) ((x) & 0x3FFFF0)
#define MICA43_SLABBASE(b) (s.p[0])
#else
#define MICA43_SLABOFF(x) (x)
#define MICA43_SLABBASE(b) (b)
#endif
#ifdef MICA43_NOEPI
#define MICA43_EPI_PASSES 0
#else
#define MICA43_EPI_PASSES 4
#endif


__device__ __forceinline__ const _Float16* chunk_base_wino43(const ConvSrcs& s, int gch, int b, int Vq) {
    int si = 0, ch = gch;
#pragma unroll
    for (int i = 0; i < MAX_SRC - 1; ++i)
        if (si == i && i + 1 < s.n && ch >= s.chunks[i]) { ch -= s.chunks[i]; si = i + 1; }
    return s.p[si] + ((int64_t)b * s.chunks_total[si] + s.chunk_off[si] + ch) * (int64_t)Vq * 192;
}

// Slab DMAs: every wave issues six per chunk, one per step in its first six steps.  Measured and dropped (profiles/r04_ablations_f43.txt):
// two or three per step, all six in step 0, and moving all 72 to the four or six OLDEST waves - cycle stamps (tools/exp/clk43.py) show
// the MFMA arbiter serving the three waves of a SIMD oldest first (waves 0-3 finish a chunk after ~8 K cycles and idle ~5 K at its
// barrier, waves 8-11 are the critical path at 13 K) and a DMA costing its wave ~170 cycles, but with the old waves issuing all of them
// the chunk took as long: the request path they occupy is the one the young waves' weight loads wait on.
template <int BN_>
__global__ __launch_bounds__(768) void conv_wino43_kernel(ConvSrcs s, const _Float16* __restrict__ wpk, int64_t wpk_bstride,
                                                          const float* __restrict__ bias, float out_scale, float* __restrict__ out,
                                                          Dims d, int cout, int total_chunks, int ntx, int nty, int nnb,
                                                          int items_per_b, int total_items, float* __restrict__ stats_ws, int ocb) {
    // ocb: channel block of the OUTPUT layout: cout for plain NDHWC, else [cout / ocb][V][ocb] (common.h: raw tensors)
    using G = Geo43;
    constexpr int BN = BN_, NCT = 4, NF = 4;
    constexpr bool SPLIT = BN == 64;                         // the two wave groups split the taps instead of the channels
    constexpr int NS = SPLIT ? 7 : 14;                       // steps per wave and chunk: 4 x (a, b, c) + x + y, or 2 x (a, b, c) + x|y
    static_assert(BN == 128 || BN == 64, "channel block");
    constexpr int DW = G::NW, DPW = G::NDMA / DW, DPS = DPW / 6;      // waves that issue slab DMAs, DMAs per wave and chunk, per step
    static_assert(G::NDMA % DW == 0 && DPW % 6 == 0 && DPS == 1, "slab DMA plan: one per wave and step in the first six steps");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % 6, wn = wave / 6;
    const int Wq = (d.W + 3) >> 2;
    const int V = d.D * d.H * d.W, Vq = d.D * d.H * Wq;

    // persistent schedule (as conv_wino16_kernel): workgroup g sits on XCD g & 7, each XCD takes a contiguous eighth of the items
    const int xcd = blockIdx.x & 7, lwg = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int range = (total_items + 7) >> 3;
    const int it_end = min(total_items, (xcd + 1) * range);
    int item = xcd * range + lwg;
    if (item >= it_end) return;

    // A operand: lane = (row r = lane & 15 -> quad r & 7, y r >> 3 ; k-group g = lane >> 4); a fragment is one z plane of the tile
    const int lr = lane & 15, lg = lane >> 4;
    const int himask = (lg >> 1) ? -1 : 0;
    const int a_common = ((lg & 1) * G::PLANE + wp * G::PP + lr) * 16;

    // packed weights: [nb][chunk][pair-step 5][p 6][unit 8][128 cout][8 halves]; units: 0,1 hi(t) k-half 0,1 | 2,3 hi(t') | 4,5 lo(t) | 6,7 lo(t')
    constexpr int ustride = BN * 16;
    constexpr int psstride = 6 * 8 * ustride;               // 98,304
    constexpr int chstride = 5 * psstride;                  // 491,520
    const unsigned w_common = (unsigned)((lg & 1) * BN + lr) * 16u;
    const int64_t nbstride = (int64_t)total_chunks * chstride;
    const char* wwave = reinterpret_cast<const char*>(wpk) + wp * 8 * ustride + (SPLIT ? 0 : wn * 64 * 16);
    // SPLIT: group wn works on pair-steps 2 wn ("A") and 2 wn + 1 ("B") and on its half of tap 8; wave-uniform scalars
    const int sp_at[2] = {wn ? (G::PZ + G::QX) * 16 : 0, wn ? 2 * G::PZ * 16 : 2 * G::QX * 16};          // W43_AOFF_TAP(4 wn), (4 wn + 2)
    const int sp_pd[2] = {G::QX * 16, wn ? G::QX * 16 : (G::PZ - 2 * G::QX) * 16};                         // W43_PAIRDELTA(2 wn), (2 wn + 1)
    const int sp_wo[3] = {2 * wn * psstride, (2 * wn + 1) * psstride, 4 * psstride + (wn ? 4 * ustride : 0)};  // weights of A, B, tap 8 (H | L)
    const int sp_d4 = wn ? 2 * BN * 16 : 0;                                                                // tap 8: y reads [b_lo ; 0], x [b_hi ; b_hi]

    // Weight fragments and the step schedule.  The three split products of a tap pair (t, t') are grouped so that operands are
    // shared between MFMAs:  with  Ahi = [a_hi(t) | a_hi(t')], Alo = [a_lo(t) | a_lo(t')]  (k-groups 0,1 = the two channel halves at
    // tap t, 2,3 at tap t') and  H = [b_hi(t) ; b_hi(t')], L = [b_lo(t) ; b_lo(t')]:
    //     step a : Alo . H        step b : Ahi . H        step c : Ahi . L        (16 MFMAs per wave each)
    // and tap 8, which has no partner:  x: [a_hi(8) | a_lo(8)] . [b_hi(8) ; b_hi(8)],  y: the same A . [b_lo(8) ; 0].
    // Per chunk that is 10 weight-fragment sets for the same 224 MFMAs that the X / X' / Y grouping of conv_wino16_kernel feeds
    // with 14: this kernel's weight fragments feed 4 MFMAs each (8 there), and with 14 sets per chunk their loads alone kept the
    // vector L1 busy for as long as the MFMAs take (12 waves x 56 KiB per chunk at 64 B/clk).
    // Three register sets hold the fragments: H of pair-step ps in set {0,1,0,1,2}[ps], L in {2,2,2,0,1}[ps] - after the five
    // pair-steps of a chunk the assignment is back where it started, so every index is a compile-time constant.  Both are
    // requested at the start of step a: L of this pair-step (used two steps later, in c) and H of the NEXT pair-step (three steps).
    half8 bq[3][NCT];
    // BN = 64 (SPLIT): steps a, b, c of the group's pair-steps A and B, then its tap-8 step z (kind 5: a single step).  Five fragment
    // sets per chunk live in the same three register sets: LA, LB -> 0; HB, W4 (tap 8) -> 1; HA -> 2.  LA is requested in step 0 (used
    // in 2), HB in 1 (used in 3, 4), LB in 3 (used in 5), the NEXT chunk's HA in 4 (HA is done after step 1) and W4 in 5 (HB is done
    // after step 4; used in 6 - one step of lead, like the next chunk's first H in the 128 variant).
#define W43_PS(st) (SPLIT ? ((st) < 6 ? (st) / 3 : 4) : ((st) < 12 ? (st) / 3 : 4))     /* SPLIT: slot 0 = A, 1 = B */
#define W43_KIND(st) (SPLIT ? ((st) < 6 ? (st) % 3 : 5) : ((st) < 12 ? (st) % 3 : (st) - 9))         /* 0 a, 1 b, 2 c, 3 x, 4 y, 5 z */
#define W43_SPSET(st) ((st) < 2 ? 2 : (st) == 2 || (st) == 5 ? 0 : 1)
#define W43_HSET(ps) ((ps) == 4 ? 2 : (ps) & 1)
#define W43_LSET(ps) ((ps) < 3 ? 2 : (ps) == 3 ? 0 : 1)
#define W43_AOFF_TAP(tap) ((((tap) / 3) * G::PZ + ((tap) % 3) * G::QX) * 16)
#define W43_PAIRDELTA(ps) ((ps) == 1 ? (G::PZ - 2 * G::QX) * 16 : G::QX * 16)
    // slab DMA slots per wave and step (the first six steps of a chunk), and the first DMA index of a step
    // (SPLIT: a chunk is seven steps, half as long; with one DMA per step the last one had a single step to land before the chunk-end wait)
#if defined(MICA43_DMAPLAN) && MICA43_DMAPLAN == 0      /* rounds 5's placement: two in each of the first three steps */
#define W43_NDMA(st) (SPLIT ? ((st) >= 0 && (st) < 3 ? 2 * DPS : 0) : ((st) >= 0 && (st) < 6 ? DPS : 0))
#define W43_DMA0(st) (SPLIT ? (st) * 2 * DPS : (st) * DPS)
#elif defined(MICA43_DMAPLAN) && MICA43_DMAPLAN == 2    /* four in step 1, two in step 2 */
#define W43_NDMA(st) (SPLIT ? ((st) == 1 ? 4 * DPS : (st) == 2 ? 2 * DPS : 0) : ((st) >= 0 && (st) < 6 ? DPS : 0))
#define W43_DMA0(st) (SPLIT ? ((st) == 2 ? 4 * DPS : 0) : (st) * DPS)
#elif defined(MICA43_DMAPLAN) && MICA43_DMAPLAN == 3    /* three in step 1, three in step 2 */
#define W43_NDMA(st) (SPLIT ? ((st) == 1 || (st) == 2 ? 3 * DPS : 0) : ((st) >= 0 && (st) < 6 ? DPS : 0))
#define W43_DMA0(st) (SPLIT ? ((st) == 2 ? 3 * DPS : 0) : (st) * DPS)
#elif defined(MICA43_DMAPLAN) && MICA43_DMAPLAN == 4    /* HB requested in step 0 beside LA, all six DMAs behind them: forced complete at step 5, five steps later */
#define W43_NDMA(st) (SPLIT ? ((st) == 0 ? 6 * DPS : 0) : ((st) >= 0 && (st) < 6 ? DPS : 0))
#define W43_DMA0(st) (SPLIT ? 0 : (st) * DPS)
#define MICA43_HB_EARLY
#elif defined(MICA43_DMAPLAN) && MICA43_DMAPLAN == 5    /* as 4, three DMAs in step 0 and three in step 1 */
#define W43_NDMA(st) (SPLIT ? ((st) == 0 || (st) == 1 ? 3 * DPS : 0) : ((st) >= 0 && (st) < 6 ? DPS : 0))
#define W43_DMA0(st) (SPLIT ? ((st) == 1 ? 3 * DPS : 0) : (st) * DPS)
#define MICA43_HB_EARLY
#else
    // SPLIT (round 6): all six right behind HB's request in step 1.  Vector-memory operations retire in order, so a DMA is forced complete
    // at the first wait for a fragment set requested AFTER it: issued behind HB (step 1) that is the wait for LB in step 5 - four steps
    // of lead for every DMA, where two per step in steps 0 .. 2 (round 5) gave the first pair three (forced by HB's wait in step 3) and
    // the last pair three.  2.5-3 % on the 8 .. 16-chunk layers (profiles/r06_warm_l2_ab.txt).  A still longer lead - HB requested in
    // step 0 beside LA and the DMAs behind both, five steps - is 1 % SLOWER than round 5's: twelve waves then leave the chunk barrier
    // with eight fragment loads and six DMAs each (variants 4, 5 above).
#define W43_NDMA(st) (SPLIT ? ((st) == 1 ? 6 * DPS : 0) : ((st) >= 0 && (st) < 6 ? DPS : 0))
#define W43_DMA0(st) (SPLIT ? 0 : (st) * DPS)
#endif
    // one fragment set: four 16-cout column tiles, 256 B apart; `delta` (bytes, applied to k-groups 2,3) selects the second tap's units
#define MICA_BLOAD43(set, base, off, delta)                                                                             \
    do {                                                                                                                \
        const char* pb_ = MICA43_WBASE(base, off);                                                                      \
        const unsigned vo_ = w_common + (unsigned)((delta) & himask);                                                   \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[set][0]) : "v"(vo_), "s"(pb_) : "memory");              \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:256" : "=v"(bq[set][1]) : "v"(vo_), "s"(pb_) : "memory");   \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" : "=v"(bq[set][2]) : "v"(vo_), "s"(pb_) : "memory");   \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:768" : "=v"(bq[set][3]) : "v"(vo_), "s"(pb_) : "memory");   \
    } while (0)
#define MICA_BLOAD43_H(ps, base) MICA_BLOAD43(W43_HSET(ps), base, (ps) * psstride, (ps) == 4 ? 0 : 2 * BN * 16)
#define MICA_BLOAD43_L(ps, base) MICA_BLOAD43(W43_LSET(ps), base, (ps) * psstride + 4 * ustride, 2 * BN * 16)
    // SPLIT: the group's fragments; offsets are wave-uniform scalars (sp_wo), destinations fixed sets
#define MICA_BLOAD43_SP_HA(base) MICA_BLOAD43(2, base, sp_wo[0], 2 * BN * 16)
#define MICA_BLOAD43_SP_LA(base) MICA_BLOAD43(0, base, sp_wo[0] + 4 * ustride, 2 * BN * 16)
#define MICA_BLOAD43_SP_HB(base) MICA_BLOAD43(1, base, sp_wo[1], 2 * BN * 16)
#define MICA_BLOAD43_SP_LB(base) MICA_BLOAD43(0, base, sp_wo[1] + 4 * ustride, 2 * BN * 16)
#define MICA_BLOAD43_SP_W4(base) MICA_BLOAD43(1, base, sp_wo[2], sp_d4)
#define W43_WAIT(N, set) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bq[set][0]), "+v"(bq[set][1]), "+v"(bq[set][2]), "+v"(bq[set][3]))
    // wait until at most n (a compile-time constant after unrolling, 4 .. 14) vector-memory operations are outstanding
#define W43_WAITN(n, set)                                                                                               \
    do {                                                                                                                \
        static_assert((n) >= 0 && (n) <= 14, "wait immediates");                                                        \
        if ((n) == 0) W43_WAIT(0, set); else if ((n) == 1) W43_WAIT(1, set); else if ((n) == 4) W43_WAIT(4, set); else if ((n) == 5) W43_WAIT(5, set); else if ((n) == 6) W43_WAIT(6, set);       \
        else if ((n) == 7) W43_WAIT(7, set); else if ((n) == 8) W43_WAIT(8, set); else if ((n) == 9) W43_WAIT(9, set);  \
        else if ((n) == 10) W43_WAIT(10, set); else if ((n) == 11) W43_WAIT(11, set); else if ((n) == 12) W43_WAIT(12, set); \
        else if ((n) == 13) W43_WAIT(13, set); else W43_WAIT(14, set);                                                  \
    } while (0)

    // Slab DMA instruction k of this wave covers the 64 consecutive slots starting at (k * 12 + wave) * 64 of the flat LDS image
    // [q 4][z 6][p 6][y 4][quad 8].  Issued UNCONDITIONALLY (lanes outside the volume masked by hand and zeroed explicitly), so that
    // the number of vector-memory operations in flight is a compile-time fact and the weight waits can leave the newest DMA outstanding.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // lane part of a DMA's source: an instruction covers 64 consecutive slots of one (plane q, slab plane vz), i.e. positions
    // pp = 2 * part + (lane >> 5), slab rows vy = (lane >> 3) & 3 and quads lane & 7 - the same three lane terms for every
    // instruction; (q, vz, part) are wave-uniform per instruction (w43_dma_*)
    const int dma_vy = (lane >> 3) & 3, dma_quad = lane & 7;
    const int dma_lane = (((lane >> 5) * 4) * Vq + dma_vy * Wq + dma_quad) * 16;
#ifdef MICA43_WARM
    // experiment (round 6, review item 4): behind a chunk's last slab DMA every wave TOUCHES the lines of the slab two chunks ahead - one
    // dword per 128-byte line, result discarded - so that the DMAs of the next chunk find them in the XCD's L2 instead of beyond it.
    // A wave's warm instruction covers the 8 lines (2 positions x 4 slab rows) of each of ITS six DMAs: lane = (k = lane >> 3, line = lane & 7).
    int warm_off, warm_yz;
    {
        const int k_ = (lane >> 3) < DPW ? (lane >> 3) : 0, l_ = lane & 7;
        const int ii_ = k_ * DW + wave;
        const int q_ = ii_ / 18, rem_ = ii_ - q_ * 18, vz_ = rem_ / 3, part_ = rem_ - vz_ * 3;
        warm_off = ((((part_ * 2) * 4 + q_) * Vq + vz_ * d.H * Wq) + ((l_ >> 2) * 4) * Vq + (l_ & 3) * Wq) * 16;
        warm_yz = (l_ & 3) | (vz_ << 4);
    }
    float warm_sink = 0.f;
#endif
#define MICA_SLAB_DMA43(srcbase, bufoff, k, org)                                                                        \
    do {                                                                                                                \
        const int ii_ = (k) * DW + dma_w;                        /* wave-uniform */                                      \
        const int lo_ = (bufoff) + ii_ * 1024;                                                                          \
        const unsigned la_ = __builtin_amdgcn_readfirstlane(lds0 + lo_);                                                \
        const int q_ = ii_ / 18, rem_ = ii_ - q_ * 18, vz_ = rem_ / 3, part_ = rem_ - vz_ * 3;                          \
        const int sc_ = (((part_ * 2) * 4 + q_) * Vq + vz_ * d.H * Wq) * 16;                                            \
        const bool ok_ = (org).i0 + dma_quad < Wq && (unsigned)((org).y0 + dma_vy) < (unsigned)d.H &&                   \
                         (unsigned)((org).z0 + vz_) < (unsigned)d.D;                                                    \
        const int go_ = ok_ ? MICA43_SLABOFF((org).base + sc_ + dma_l) : -1;                                             \
        unsigned long long sv_;                                                                                         \
        asm volatile("s_mov_b64 %0, exec\n\tv_cmp_lt_i32 vcc, -1, %1\n\ts_mov_b64 exec, vcc\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" \
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"                                              \
                     : "=&s"(sv_) : "v"(go_), "s"(MICA43_SLABBASE(srcbase)), "s"(la_) : "memory", "vcc", "m0");          \
        if (!ok_) *reinterpret_cast<uint4*>(smem + lo_ + lane * 16) = make_uint4(0, 0, 0, 0);                           \
    } while (0)

    struct Item {
        int b, nb, tile;
        int i0, y0, z0, base;          // slab origin: first x quad, y, z (halo included) and its byte offset in a chunk
        const char* w;                 // this wave's weights of chunk 0
        const _Float16* src0;          // chunk 0 of the operand
    };
    auto decode = [&](int it) {
        Item r;
        r.b = it / items_per_b;
        const int id = it - r.b * items_per_b;
        // the channel blocks of a tile run fastest: the CUs of an XCD round share the tile's slab in L2 (one block, or two, per pass over
        // the tiles measured the same: profiles/r04_ablations_f43.txt)
        r.nb = id % nnb;
        const int seq = id / nnb;
        int tx, ty, tz;
        if (((nty & 7) | (((d.D + 3) >> 2) & 3)) == 0) {       // compact 8(y) x 4(z) blocks of tiles per XCD round (shared y/z halos in L2)
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
        r.i0 = tx * G::QX;
        r.y0 = ty * G::TY - 1;
        r.z0 = tz * G::TZ - 1;
        r.base = ((r.z0 * d.H + r.y0) * Wq + r.i0) * 16;
        r.w = wwave + (int64_t)r.b * wpk_bstride * 2 + r.nb * nbstride;
        r.src0 = chunk_base_wino43(s, 0, r.b, Vq);
        return r;
    };
    const int64_t chunk_halves = (int64_t)Vq * 192;

    Item cur = decode(item);
    int nitem = item + per_xcd;
    Item nxt = decode(nitem < it_end ? nitem : item);

    // prologue of the first item: slab chunk 0 -> buffer 0, H fragments of the first pair-step
    if constexpr (SPLIT) MICA_BLOAD43_SP_HA(cur.w); else MICA_BLOAD43_H(0, cur.w);
    {
        const int dma_w = wave, dma_l = dma_lane;
#pragma unroll
        for (int k = 0; k < DPW; ++k) MICA_SLAB_DMA43(cur.src0, 0, k, cur);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int par = 0;

#ifdef MICA43_CLOCKS
    int item_no = 0;
#endif
    for (;;) {
        const bool has_next = nitem < it_end;
        floatx4w acc[NF][NCT];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[f][c][i] = 0.f;

        const _Float16* run = cur.src0;
        int si = 0, left = s.chunks[0];
        const char* wcur = cur.w;
#pragma clang loop unroll(disable)
        for (int gch = 0; gch < total_chunks; ++gch) {
#ifdef MICA43_CLOCKS
            const bool clk_on = blockIdx.x == 8 && item_no == 3 && gch == 10;
            unsigned* clk_lds = reinterpret_cast<unsigned*>(smem + 2 * Geo43::CH_BYTES);
#endif
            W43_STAMP(0);
            // laundered per chunk: as loop invariants the DMA source offsets of a wave are hoisted out of the item loop into VGPRs and
            // SGPRs that this kernel does not have to spare
            int dma_w = wave, dma_l = dma_lane;
            asm volatile("" : "+s"(dma_w), "+v"(dma_l));
            const char* A = smem + par * G::CH_BYTES;
            const int nxt_off = (par ^ 1) * G::CH_BYTES;
            const bool last = gch + 1 == total_chunks;
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
            // A fragment base of step st: pair steps read the hi (bc) or lo (a) planes of tap t in k-groups 0,1 and of tap t' in 2,3;
            // the tap-8 steps read the hi planes in k-groups 0,1 and the lo planes in 2,3
#define W43_ABASE(st) (A + a_common + (W43_KIND(st) >= 3 ? ((2 * G::PLANE * 16) & himask) + W43_AOFF_TAP(8)                                  \
                                       : SPLIT ? (sp_pd[W43_PS(st)] & himask) + sp_at[W43_PS(st)] + (W43_KIND(st) == 0 ? 2 * G::PLANE * 16 : 0) \
                                               : (W43_PAIRDELTA(W43_PS(st)) & himask) + W43_AOFF_TAP(2 * W43_PS(st)) +                          \
                                                 (W43_KIND(st) == 0 ? 2 * G::PLANE * 16 : 0)))
#define W43_AFRAG(base, f) (*reinterpret_cast<const half8*>((base) + (f) * G::PZ * 16))
            const char* ab_cur = W43_ABASE(0);
            constexpr int AD = 3;                        // A fragments requested ahead (each feeds 4 MFMAs)
            half8 ar[AD + 1];
#pragma unroll
            for (int i = 0; i < AD; ++i) ar[i] = W43_AFRAG(ab_cur, i);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int ps = W43_PS(st), kind = W43_KIND(st);
                // requests of this step, then the wait for the fragments it uses (in flight and NEWER than those: what was just
                // requested, the fragments requested with them, and the slab DMAs issued since)
                // loads are spread over the steps (all twelve waves reach them together after the chunk barrier, and eight 1-KB requests
                // per wave at once queue for > 1000 cycles in the vector memory pipeline: cycle stamps, tools/exp/clk43.py): step a requests L
                // of this pair-step (used in c), step b requests H of the next pair-step (used from its a), x requests L of tap 8, y the
                // next chunk's first H
                if constexpr (SPLIT) {
                    // in flight and NEWER than the set a step waits for: the fragment sets requested since and the slab DMAs issued since
#ifdef MICA43_HB_EARLY
                    if (st == 0) { MICA_BLOAD43_SP_LA(wcur); MICA_BLOAD43_SP_HB(wcur); W43_WAITN(8, 2); }        // HA; newer: LA, HB (set 1: W4 was used up in the previous chunk's last step)
                    else if (st == 1) { }
                    else if (st == 2) { W43_WAITN(4 + W43_NDMA(0) + W43_NDMA(1), 0); }                          // LA; newer: HB, DMA 0, DMA 1
#else
                    if (st == 0) { MICA_BLOAD43_SP_LA(wcur); W43_WAITN(4, 2); }                                   // HA: complete since the chunk-end wait
                    else if (st == 1) { MICA_BLOAD43_SP_HB(wcur); }
                    else if (st == 2) { W43_WAITN(4 + W43_NDMA(0) + W43_NDMA(1), 0); }                          // LA; newer: DMA 0, HB, DMA 1
#endif
#ifdef MICA43_WARM
                    else if (st == 3) { MICA_BLOAD43_SP_LB(wcur); W43_WAITN(4 + W43_NDMA(1) + W43_NDMA(2) + 1, 1); }  // HB; newer: DMA 1, DMA 2, the warm touch, LB
#elif defined(MICA43_HB_EARLY)
                    else if (st == 3) { MICA_BLOAD43_SP_LB(wcur); W43_WAITN(4 + W43_NDMA(0) + W43_NDMA(1) + W43_NDMA(2), 1); }  // HB; newer: DMA 0, DMA 1, DMA 2, LB
#else
                    else if (st == 3) { MICA_BLOAD43_SP_LB(wcur); W43_WAITN(4 + W43_NDMA(1) + W43_NDMA(2), 1); }  // HB; newer: DMA 1, DMA 2, LB
#endif
                    else if (st == 4) { MICA_BLOAD43_SP_HA(wnxt); }                                               // the next chunk's (or item's) HA
                    else if (st == 5) { MICA_BLOAD43_SP_W4(wcur); W43_WAITN(8 + W43_NDMA(3) + W43_NDMA(4), 0); } // LB; newer: DMA 3, HA', DMA 4, W4
                    else { W43_WAITN(W43_NDMA(5), 1); }                                                           // W4; newer: DMA 5
                } else
                if (kind == 0) {
                    MICA_BLOAD43_L(ps, wcur);
                    if (st == 0) W43_WAITN(4, W43_HSET(0)); else if (st == 3) W43_WAITN(4 + W43_NDMA(1) + W43_NDMA(2), W43_HSET(1));
                    else if (st == 6) W43_WAITN(4 + W43_NDMA(4) + W43_NDMA(5), W43_HSET(2)); else W43_WAITN(4 + W43_NDMA(7) + W43_NDMA(8), W43_HSET(3));
                } else if (kind == 1) {
                    if (ps == 0) MICA_BLOAD43_H(1, wcur); else if (ps == 1) MICA_BLOAD43_H(2, wcur); else if (ps == 2) MICA_BLOAD43_H(3, wcur); else MICA_BLOAD43_H(4, wcur);
                } else if (kind == 2) {
                    if (st == 2) W43_WAITN(4 + W43_NDMA(0) + W43_NDMA(1), W43_LSET(0)); else if (st == 5) W43_WAITN(4 + W43_NDMA(3) + W43_NDMA(4), W43_LSET(1));
                    else if (st == 8) W43_WAITN(4 + W43_NDMA(6) + W43_NDMA(7), W43_LSET(2)); else W43_WAITN(4 + W43_NDMA(9) + W43_NDMA(10), W43_LSET(3));
                } else if (kind == 3) {
                    MICA_BLOAD43_L(4, wcur);
                    W43_WAITN(4 + W43_NDMA(10) + W43_NDMA(11), W43_HSET(4));
                } else if (kind == 4) {
                    MICA_BLOAD43_H(0, wnxt);
                    W43_WAITN(4 + W43_NDMA(12), W43_LSET(4));
                }
                static_assert(W43_NDMA(13) == 0 && W43_NDMA(6) == 0, "no slab DMA in the last step: the chunk-end wait leaves exactly the next chunk's first H in flight");
                __builtin_amdgcn_sched_barrier(0);
                W43_STAMP(1 + 2 * st);
#pragma unroll
                for (int q = 0; q < W43_NDMA(st); ++q) MICA_SLAB_DMA43(nsrc, nxt_off, W43_DMA0(st) + q, org);
#ifdef MICA43_WARM
                if constexpr (SPLIT) if (st == 2) {
                    int g2 = gch + 2;
                    Item it2 = cur;
                    if (g2 >= total_chunks) { g2 -= total_chunks; it2 = nxt; }
                    if (g2 >= total_chunks) g2 = total_chunks - 1;
                    const char* src2 = reinterpret_cast<const char*>(chunk_base_wino43(s, g2, it2.b, Vq));
                    int wo = warm_off, wyz = warm_yz;
                    asm volatile("" : "+v"(wo), "+v"(wyz));
                    const bool ok2 = (unsigned)(it2.y0 + (wyz & 3)) < (unsigned)d.H && (unsigned)(it2.z0 + (wyz >> 4)) < (unsigned)d.D;
                    const int go2 = ok2 ? it2.base + wo : 0;
                    asm volatile("global_load_dword %0, %1, %2" : "=v"(warm_sink) : "v"(go2), "s"(src2) : "memory");
                }
#endif
                const char* ab_nxt = ab_cur;
                if (st + 1 < NS) ab_nxt = W43_ABASE(st + 1);
                half8 (&b1)[NCT] = bq[SPLIT ? W43_SPSET(st) : (kind == 2 || kind == 4) ? W43_LSET(ps) : W43_HSET(ps)];
#ifdef MICA43_PRIO
                {   // experiments: priorities of the three waves that share a SIMD (wave, wave + 4, wave + 8)
                    const int g3 = wave >> 2;
                    const int pr = MICA43_PRIO == 1 ? (g3 + st) % 3 : MICA43_PRIO == 2 ? (g3 + st / 5) % 3 : MICA43_PRIO == 3 ? 2 - g3 : (g3 + st / 2) % 3;
                    if (MICA43_PRIO != 3 || st == 0) {
                        if (pr == 0) asm volatile("s_setprio 0"); else if (pr == 1) asm volatile("s_setprio 1"); else asm volatile("s_setprio 2");
                    }
                }
#endif
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int fi = st * NF + f;                 // fragment index within the chunk; lives in ar[fi % (AD + 1)]
                    // steps b and c (and x and y) read the same A fragments: they stay in the ring (slot = f, AD + 1 == NF) and the second
                    // step only requests the fragments of the step after it
                    static_assert(AD + 1 == NF, "ring slot == fragment index");
                    const bool first_of_two = kind == 1 || kind == 3, second_of_two = kind == 2 || kind == 4;
                    if (f + AD < NF) { if (!second_of_two) ar[(fi + AD) % (AD + 1)] = W43_AFRAG(ab_cur, f + AD); }
                    else if (st + 1 < NS && !first_of_two) ar[(fi + AD) % (AD + 1)] = W43_AFRAG(ab_nxt, f + AD - NF);
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[f][c]) : "v"(ar[fi % (AD + 1)]), "v"(b1[c]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                ab_cur = ab_nxt;
                W43_STAMP(2 + 2 * st);
            }
#undef W43_ABASE
#undef W43_AFRAG
            // the slab DMAs of this chunk are older than the four weight loads (the next chunk's first H) still wanted in flight
            W43_STAMP(30);
            // (SPLIT: the next chunk's HA was requested three steps ago and is OLDER than this chunk's last slab DMA: everything drains)
            if constexpr (SPLIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#ifdef MICA43_WARM
            asm volatile("" :: "v"(warm_sink));
#endif
            W43_STAMP(31);
            __syncthreads();
            W43_STAMP(32);
            par ^= 1;
            wcur = wnxt;
        }
#ifdef MICA43_PRIO
        asm volatile("s_setprio 0");
#endif
        // the next item's first weight fragments were requested in the last step: retire them here (the compiler cannot see them in flight)
        if constexpr (SPLIT) W43_WAIT(0, 2); else W43_WAIT(0, W43_HSET(0));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU/LDS readers

        // ---- output transform through the idle slab buffer: four passes of 32 columns; T = [z 4][p 6][row 16 = y*8+quad][32 + 4] floats ----
        {
            constexpr int CP = 32, RS = CP + 4, REG = 16 * RS;
            static_assert(4 * 6 * REG * 4 <= G::CH_BYTES, "epilogue fits one slab buffer");
            float* xs = reinterpret_cast<float*>(smem + (par ^ 1) * G::CH_BYTES);
            const int ib = cur.b, inb = cur.nb, itile = cur.tile;
            const int tx = itile % ntx, ty = (itile / ntx) % nty, tz = itile / (ntx * nty);
            const int nnitem = nitem + per_xcd;
            const Item nn = decode(nnitem < it_end ? nnitem : (has_next ? nitem : item));
            const int fz = wave & 3, fch = (wave >> 2) & 1;          // finishing role of waves 0..7: z plane, 16-column half
            int elane = lane;
            asm volatile("" : "+v"(elane));      // per item: keeps the epilogue's lane-dependent addresses from being hoisted over (and kept live through) the main loop
            const int frow = elane >> 2, fcg = elane & 3;
            const int elr = elane & 15, elg = elane >> 4;
            const int P = (items_per_b / nnb) * 4;
#pragma unroll
            for (int pass = 0; pass < (SPLIT ? MICA43_EPI_PASSES / 2 : MICA43_EPI_PASSES); ++pass) {
                const int c0 = SPLIT ? pass * 2 : (pass >> 1) * 2;   // first column tile of the pass
                const int wq = SPLIT ? 1 : pass & 1;                 // the wave group that writes T first (SPLIT: group 1, then group 0 adds its own)
                if (wn == wq) {
                    // C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg = y*8 + quad of the fragment
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        float* dst = xs + (f * 6 + wp) * REG;
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int i = 0; i < 4; ++i) dst[(elg * 4 + i) * RS + c * 16 + elr] = acc[f][c0 + c][i];
                    }
                }
                __syncthreads();
                if constexpr (SPLIT) {
                    // the partial sums of the two tap groups meet: wave (wp, 0) reads what wave (wp, 1) stored at the very same addresses,
                    // adds its own and writes the sum back (plain LDS reads and writes: LDS atomics ran at a fraction of their rate -
                    // 35 us per item against 8, tools/exp/conv64_ab.sh)
                    if (wn == 0) {
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            float* dst = xs + (f * 6 + wp) * REG;
#pragma unroll
                            for (int c = 0; c < 2; ++c)
#pragma unroll
                                for (int i = 0; i < 4; ++i) dst[(elg * 4 + i) * RS + c * 16 + elr] += acc[f][c0 + c][i];
                        }
                    }
                    __syncthreads();
                }
                if (wave < 8) {
                    const float* src = xs + (fz * 6) * REG + frow * RS + fch * 16 + fcg * 4;
                    const int n0 = SPLIT ? inb * BN + pass * 32 + fch * 16 + fcg * 4 : inb * BN + wq * 64 + (pass >> 1) * 32 + fch * 16 + fcg * 4;
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (bias) bv = *reinterpret_cast<const float4*>(bias + n0);
                    const float4 m0 = *reinterpret_cast<const float4*>(src + 0 * REG);
                    const float4 m1 = *reinterpret_cast<const float4*>(src + 1 * REG);
                    const float4 m2 = *reinterpret_cast<const float4*>(src + 2 * REG);
                    const float4 m3 = *reinterpret_cast<const float4*>(src + 3 * REG);
                    const float4 m4 = *reinterpret_cast<const float4*>(src + 4 * REG);
                    const float4 m5 = *reinterpret_cast<const float4*>(src + 5 * REG);
                    const float a0[4] = {m0.x, m0.y, m0.z, m0.w}, a1[4] = {m1.x, m1.y, m1.z, m1.w}, a2[4] = {m2.x, m2.y, m2.z, m2.w};
                    const float a3[4] = {m3.x, m3.y, m3.z, m3.w}, a4[4] = {m4.x, m4.y, m4.z, m4.w}, a5[4] = {m5.x, m5.y, m5.z, m5.w};
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                    float yv[4][4];                                  // [x within the quad][channel]
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float s12 = a1[c] + a2[c], d12 = a1[c] - a2[c], s34 = a3[c] + a4[c], d34 = a3[c] - a4[c];
                        yv[0][c] = (a0[c] + s12 + s34) * out_scale + bb[c];
                        yv[1][c] = fmaf(W43_A, d12, W43_B * d34) * out_scale + bb[c];
                        yv[2][c] = fmaf(W43_A2, s12, W43_B2 * s34) * out_scale + bb[c];
                        yv[3][c] = (fmaf(W43_A3, d12, W43_B3 * d34) + a5[c]) * out_scale + bb[c];
                    }
                    const int gx = (tx * G::QX + (frow & (G::QX - 1))) * 4, gy = ty * G::TY + frow / G::QX, gz = tz * G::TZ + fz;
                    // statistics: sums of (v - shift), (v - shift)^2 with one shift per channel for the whole wave (the tile's first voxel)
                    float sk[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) sk[c] = __shfl(yv[0][c], fcg);      // lane fcg holds row 0 of this channel group
                    const bool inr = gy < d.H && gz < d.D;
                    float sn = 0.f, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool in = inr && gx + j < d.W;
                        if (in) {
                            float* o = out + (int64_t)ib * V * cout + (ocb != cout ? (int64_t)(n0 >> 5) * V * 32 + (n0 & 31) : (int64_t)n0) +
                                       ((int64_t)(gz * d.H + gy) * d.W + gx + j) * ocb;
                            *reinterpret_cast<float4*>(o) = make_float4(yv[j][0], yv[j][1], yv[j][2], yv[j][3]);
                        }
                        const float wv = in ? 1.f : 0.f;
                        sn += wv;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float t = (yv[j][c] - sk[c]) * wv;
                            s1[c] += t;
                            s2[c] = fmaf(t, t, s2[c]);
                        }
                    }
                    if (stats_ws) {
                        // lanes of one 4-channel group sit 4 apart: DPP rotations inside a row of 16 lanes, then two cross-row exchanges
#define W43_ROR_ADD(x, n) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, false))
                        W43_ROR_ADD(sn, 4);
                        W43_ROR_ADD(sn, 8);
#pragma unroll
                        for (int c = 0; c < 4; ++c) { W43_ROR_ADD(s1[c], 4); W43_ROR_ADD(s2[c], 4); W43_ROR_ADD(s1[c], 8); W43_ROR_ADD(s2[c], 8); }
#undef W43_ROR_ADD
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
                __syncthreads();
            }
#ifdef MICA43_CLOCKS
            if (blockIdx.x == 8 && item_no == 3) {
                __syncthreads();
                const unsigned* cl = reinterpret_cast<const unsigned*>(smem + 2 * Geo43::CH_BYTES);
                for (int i = tid; i < 12 * 48; i += 768) g_mica43_clk[i] = cl[i];
            }
            ++item_no;
#endif
            if (!has_next) break;
            item = nitem;
            nitem = nnitem;
            cur = nxt;
            nxt = nn;
        }
    }
#undef MICA_BLOAD43
#undef MICA_SLAB_DMA43
    // nothing may still be in flight towards this workgroup's registers or LDS when it ends
    if constexpr (SPLIT) W43_WAIT(0, 2); else W43_WAIT(0, W43_HSET(0));
#undef MICA_BLOAD43_H
#undef MICA_BLOAD43_L
#undef MICA_BLOAD43_SP_HA
#undef MICA_BLOAD43_SP_LA
#undef MICA_BLOAD43_SP_HB
#undef MICA_BLOAD43_SP_LB
#undef MICA_BLOAD43_SP_W4
#undef W43_SPSET
#undef W43_WAIT
#undef W43_NDMA
#undef W43_DMA0
#undef W43_WAITN
#undef W43_PS
#undef W43_KIND
#undef W43_HSET
#undef W43_LSET
#undef W43_AOFF_TAP
#undef W43_PAIRDELTA
}

bool conv_wino43_eligible(int cout) { return cout % 128 == 0 || cout == 64; }
static int conv_wino43_bn(int cout) { return cout % 128 == 0 ? 128 : 64; }

#ifdef MICA43_CLOCKS
extern "C" int mica_debug_conv43(unsigned* h_out, int n) {
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_mica43_clk), sizeof(unsigned) * (n < 12 * 48 ? n : 12 * 48));
}
#endif

// Returns the number of statistics partials per (tile, channel) written to stats_ws (when non-null): f32 [B][P][cout][3].
int launch_conv_wino43(const ConvSrcs& s, const _Float16* wpk, int64_t wpk_bstride, const float* bias, float out_scale, float* out,
                       int B, Dims d, int cout, float* stats_ws, hipStream_t st, int out_cblk) {
    const int ocb = out_cblk == 32 && cout % 32 == 0 ? 32 : cout;          // the blocked raw layout has 32-channel blocks
    if (!conv_wino43_eligible(cout)) { refuse_launch("conv_wino43: cout must be a multiple of 128, or 64"); return 0; }
    if ((int64_t)d.D * d.H * ((d.W + 3) / 4) * 24 * 16 >= (1ll << 31)) { refuse_launch("conv_wino43: tile too large for 32-bit slab offsets"); return 0; }
    int total = 0;
    for (int i = 0; i < s.n; ++i) total += s.chunks[i];
    const int ntx = (d.W + 4 * Geo43::QX - 1) / (4 * Geo43::QX), nty = (d.H + Geo43::TY - 1) / Geo43::TY, ntz = (d.D + Geo43::TZ - 1) / Geo43::TZ,
              bn = conv_wino43_bn(cout), nnb = cout / bn;
#ifdef MICA43_CLOCKS
    const size_t lds = 2 * Geo43::CH_BYTES + 12 * 48 * 4;
#else
    const size_t lds = 2 * Geo43::CH_BYTES;          // two slab buffers
#endif
    static PerDeviceOnce once;
    static int cus_of[64] = {0};
    const int dev = once.run([&](int dv) {
        (void)hipFuncSetAttribute((const void*)conv_wino43_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_wino43_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipDeviceProp_t prop;
        int c = 0;
        if (hipGetDeviceProperties(&prop, dv) == hipSuccess) c = prop.multiProcessorCount;
        if (c < 8) c = 256;
        cus_of[dv] = c & ~7;
    });
    const int cus = cus_of[dev];
    const int items_per_b = ntx * nty * ntz * nnb, total_items = items_per_b * B;
    const int nwg = total_items >= cus ? cus : ((total_items + 7) / 8) * 8;
    if (bn == 128)
        hipLaunchKernelGGL(conv_wino43_kernel<128>, dim3(nwg), dim3(768), lds, st, s, wpk, wpk_bstride, bias, out_scale, out, d, cout, total, ntx,
                           nty, nnb, items_per_b, total_items, stats_ws, ocb);
    else
        hipLaunchKernelGGL(conv_wino43_kernel<64>, dim3(nwg), dim3(768), lds, st, s, wpk, wpk_bstride, bias, out_scale, out, d, cout, total, ntx,
                           nty, nnb, items_per_b, total_items, stats_ws, ocb);
    return ntx * nty * ntz * 4;
}

// ------------------------------------------------------------------------------------------------
// weights for conv_wino43: [B][nb = Cout/bn][chunk][pair-step 5][p 6][unit 8][bn][8] halves, bn = 128 (64 for Cout = 64); unit u: 0,1 = hi of tap t (k-half
// 0,1), 2,3 = hi of tap t' = t+1, 4,5 = lo of tap t, 6,7 = lo of tap t'; pair-step 4 is tap 8 alone (units 2,3,6,7 zero).
// The weight transform u_p = sum_k G[p][k] g_k is done in f64 and rounded once to f32 before the power-of-two scaling.
// ------------------------------------------------------------------------------------------------
__global__ void pack_weights_wino43_kernel(const float* __restrict__ w, int cout, int cin, Segs43 sg, int total_chunks,
                                           const float* __restrict__ cin_scale, float mul, _Float16* __restrict__ wpk, int64_t per_b, int bn) {
    const int b = blockIdx.y;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [nb][chunk][ps 5][p 6][unit 8][n in block]
    int64_t total = (int64_t)total_chunks * 5 * 6 * 8 * cout;
    if (e >= total) return;
    const int nl = e % bn;
    const int u = (e / bn) & 7;
    const int pp = (e / (bn * 8)) % 6;
    const int ps = (e / (bn * 48)) % 5;
    const int gch = (e / (bn * 240)) % total_chunks;
    const int n = (int)(e / ((int64_t)bn * 240 * total_chunks)) * bn + nl;
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
            const double g0 = g[0], g1 = g[1], g2 = g[2];
            // G rows [1, p, p^2] / N_p, N_p = prod over the other finite points (p - p'); a = 3/2, b = 2/3, a b = 1
            const double a = 1.5, b = 2.0 / 3.0, na = 2.0 * a * a * (a * a - b * b), nb = 2.0 * b * b * (b * b - a * a);
            double uu;
            switch (pp) {
                case 0: uu = g0; break;
                case 1: uu = (g0 + a * g1 + a * a * g2) / na; break;
                case 2: uu = (g0 - a * g1 + a * a * g2) / na; break;
                case 3: uu = (g0 + b * g1 + b * b * g2) / nb; break;
                case 4: uu = (g0 - b * g1 + b * b * g2) / nb; break;
                default: uu = g2; break;
            }
            v = (float)uu * mul;
            if (cin_scale) v *= cin_scale[(int64_t)b * cin + ci];
        }
        _Float16 hi = (_Float16)v;
        _Float16 lo = mica_lo_half(v - (float)hi);
        o[j] = kind ? lo : hi;
    }
    *reinterpret_cast<half8*>(wpk + (int64_t)b * per_b + e * 8) = o;
}

int64_t packed_weight_halves_wino43(int cout, int total_chunks) { return (int64_t)total_chunks * 5 * 48 * cout * 8; }

void launch_pack_weights_wino43(const float* w, int cout, int cin, const int* h_seg_c, const int* h_seg_cp, int nseg,
                                const float* cin_scale, int B, float cout_scale, float wscale, _Float16* wpk, hipStream_t st) {
    Segs43 sg;
    sg.n = nseg;
    int total_chunks = 0;
    for (int i = 0; i < nseg; ++i) {
        sg.c[i] = h_seg_c[i];
        sg.cp[i] = h_seg_cp[i];
        total_chunks += h_seg_cp[i] / 16;
    }
    const int64_t total = (int64_t)total_chunks * 5 * 48 * cout;
    dim3 grid((unsigned)((total + 255) / 256), B);
    hipLaunchKernelGGL(pack_weights_wino43_kernel, grid, dim3(256), 0, st, w, cout, cin, sg, total_chunks, cin_scale, cout_scale * wscale,
                       wpk, packed_weight_halves_wino43(cout, total_chunks), conv_wino43_bn(cout));
}

// ------------------------------------------------------------------------------------------------
// Operand producers: y = relu?((x - mean) * rstd) of a raw f32 NDHWC tensor, written as the F(4,3) input transform of every
// output quad in wino43 layout.  Thread = (8-channel group, quad).  enc.ascale is the scale of the OPERAND (the callers pass a
// quarter of the context's activation scale, see the header).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_wino43_kernel(const float* __restrict__ x, Dims d, int C, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, int relu, SplitView wino, SplitEnc enc) {
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int Cs = C < 64 ? C : 64;                 // channel slab per block: a wave's stores form >= 512-B runs per chunk plane
    const int G = Cs >> 3, SUB = 256 / G;
    const int tid = threadIdx.x, g = blockIdx.z * (Cs >> 3) + tid % G, sub = tid / G;
    const int Wq = (d.W + 3) >> 2;
    const int V = d.D * d.H * d.W, Vq = d.D * d.H * Wq;
    const int per = (Vq + nblk - 1) / nblk;
    const int p0 = blk * per, p1 = min(Vq, p0 + per);
    float m[8], r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t ci = (int64_t)b * C + g * 8 + j;
        m[j] = mean ? mean[ci] : 0.f;
        r[j] = rstd ? rstd[ci] : 1.f;
    }
    int bad = 0;
    const float* xb = x + (int64_t)b * V * C + g * 8;
    _Float16* wb = wino.p + (((int64_t)b * wino.chunks_total + wino.chunk_off + (g >> 1)) * (int64_t)Vq) * 192;
    const int kh = g & 1;
    for (int ph = p0 + sub; ph < p1; ph += SUB) {
        const int row = ph / Wq, i = ph - row * Wq;
        const int xo = 4 * i;
        float dv[6][8];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int xx = xo - 1 + k;
            const bool ok = (unsigned)xx < (unsigned)d.W;
            const int xc = ok ? xx : xo;
            const float4 a = *reinterpret_cast<const float4*>(xb + ((int64_t)row * d.W + xc) * C);
            const float4 c = *reinterpret_cast<const float4*>(xb + ((int64_t)row * d.W + xc) * C + 4);
            const float y[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (y[j] - m[j]) * r[j];
                if (relu) t = fmaxf(t, 0.f);
                dv[k][j] = ok ? t : 0.f;
            }
        }
        float t[6][8];
        wino43_input_transform(dv, t);
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) {
            half8 hi, lo;
            mica_split8(t[pp], hi, lo, bad, enc.ascale);
            *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + kh) * Vq + ph) * 8) = hi;
            *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + 2 + kh) * Vq + ph) * 8) = lo;
        }
    }
    if (bad) atomicOr(enc.err + b, bad);
}

void launch_prep_wino43(const float* x, int B, Dims d, int C, const float* mean, const float* rstd, int relu, SplitView wino,
                        SplitEnc enc, hipStream_t st) {
    const int Cs = C < 64 ? C : 64, G = Cs / 8, SUB = 256 / G;
    const int Vq = d.D * d.H * ((d.W + 3) / 4);
    int nblk = (Vq + SUB * 2 - 1) / (SUB * 2);
    if (nblk > 4096) nblk = 4096;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(prep_wino43_kernel, dim3(nblk, B, C / Cs), dim3(256), 0, st, x, d, C, mean, rstd, relu, wino, enc);
}

// NCDHW f32 [B][C][V] -> wino43 layout (single-op entry point)
__global__ __launch_bounds__(256) void prep_ncdhw_wino43_kernel(const float* __restrict__ x, Dims d, int C, SplitView wino, SplitEnc enc) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int Wq = (d.W + 3) >> 2;
    const int V = d.D * d.H * d.W, Vq = d.D * d.H * Wq;
    const int ph = blockIdx.x * 256 + threadIdx.x;
    if (ph >= Vq) return;
    const int row = ph / Wq, i = ph - row * Wq, xo = 4 * i;
    int bad = 0;
    _Float16* wb = wino.p + (((int64_t)b * wino.chunks_total + wino.chunk_off + ch) * (int64_t)Vq) * 192;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        float dv[6][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 16 + kh * 8 + j;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int xx = xo - 1 + k;
                dv[k][j] = (c < C && (unsigned)xx < (unsigned)d.W) ? x[((int64_t)b * C + c) * V + (int64_t)row * d.W + xx] : 0.f;
            }
        }
        float t[6][8];
        wino43_input_transform(dv, t);
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) {
            half8 hi, lo;
            mica_split8(t[pp], hi, lo, bad, enc.ascale);
            *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + kh) * Vq + ph) * 8) = hi;
            *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + 2 + kh) * Vq + ph) * 8) = lo;
        }
    }
    if (bad) atomicOr(enc.err + b, bad);
}

void launch_prep_ncdhw_wino43(const float* x, int B, Dims d, int C, SplitView wino, SplitEnc enc, hipStream_t st) {
    const int Vq = d.D * d.H * ((d.W + 3) / 4);
    dim3 grid((Vq + 255) / 256, (C + 15) / 16, B);
    hipLaunchKernelGGL(prep_ncdhw_wino43_kernel, grid, dim3(256), 0, st, x, d, C, wino, enc);
}

}  // namespace mica

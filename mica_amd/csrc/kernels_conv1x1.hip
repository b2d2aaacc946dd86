// The 1x1x1 convolutions (reference models/model.py:28 exp_downsizing, :38 fusion, :96 dual_attn.fusion, :158-160 FPN laterals)
// as one streaming GEMM kernel whose prologue and epilogue replace the operand passes around it.
//
// These layers are HBM-bound (1.7 % of the FLOPs, arithmetic intensity 20-85 FLOP/B), and every one of them sits between
// two re-encoding passes in the first version of the graph: its inputs were InstanceNorm-applied, ReLU'd and split into
// f16 hi/lo by a prep pass (read 4 B + write 4 B per value), and its f32 output went through another pass to become the
// Winograd operand of the 3x3x3 conv that follows (read 4 B + write 8 B).  Here
//   * a source may be the RAW f32 output of the producing conv together with its InstanceNorm constants: the
//     normalisation, the ReLU and the hi/lo split happen while the tile is staged into LDS (VALU work the kernel has to
//     spare: it waits on HBM) - or an operand that already is in split form (stem output, gated AF3 features);
//   * the epilogue writes the consumer's operand directly: the Winograd F(2,3) input transform along x needs the two
//     x-neighbours of every output pair, and a workgroup owns whole x rows (W <= 64), so the neighbours are either in its
//     own tile or outside the volume (= the consumer's zero padding).  Raw f32 output is there for the single-op entry point
//     and for tile widths that do not divide the 256-voxel tile.
//
// Arithmetic: split-f16 products as in the 3x3x3 kernel, x*w = a_hi*b_hi + a_lo*b_hi + a_hi*b_lo with f32 accumulation, on
// v_mfma_f32_16x16x32_f16: per PAIR of 16-channel chunks (c, c') three instructions per 16x16 tile,
//   X(c)  : A = [a_hi(c) | a_lo(c)]   B = [b_hi(c) ; b_hi(c)]          X(c') likewise
//   Y     : A = [a_hi(c) | a_hi(c')]  B = [b_lo(c) ; b_lo(c')]
// Workgroup: 4 waves, 128 voxels x Cout; wave wn owns all 128 voxels (8 row fragments) x the column tiles {wn + 4 j}.  Small
// workgroups on purpose: two to three of them share a CU (43 KB of LDS each), out of phase, so that one's loads overlap another's
// MFMAs and a third's epilogue stores - with one 8-wave workgroup per CU the kernel alternated between a read phase and a
// write phase and reached 2.7 TB/s.
// Tried in round 3 and dropped (tools: rocprofv3 kernel trace of bench.py): decoupling the activation stream from the weight
// stream (vector-memory operations retire in order, so a wait for a weight fragment also waits for every activation load issued
// before it) by requesting a whole pair's weight fragments one pair ahead and running the activation loads four pairs
// ahead through a register ring, Cout = 256 as eight waves x two column tiles: 12-40 % SLOWER (<1> 747 -> 842 us, <2> 994 ->
// 1158, <4> 2792 -> 3927): the register cost halves the workgroups per CU, and it is the number of independent workgroups on a
// CU - one's VALU-heavy commit / epilogue beside another's MFMAs and a third's HBM waits - that keeps the three pipes busy, not
// the bytes in flight.  (Round 5 read the generated waits instead - tools/audit_waits.py - and found what that experiment had been
// after without the register cost: see "Order of the vector-memory queue" at the main loop; 0.47 -> 0.57 of the HBM roofline.)  Also tried: padding the stage planes and XOR-swizzling the epilogue tile against the 20 % LDS bank-conflict
// share the counters show (under a 32-bank and under a 64-bank model of the LDS): the counter went to 40 % and the kernels 2-4 %
// slower both times; the kernel is not LDS-bound (LDS busy 21-32 % of the cycles), so the layout stays.
// Tried in round 4 and dropped: fetching a raw chunk pair line-wise (thread = (row, 8-channel group of the pair's 32 channels): the four
// lanes of a quad read one 128-byte line instead of two half-used ones) - bench.py 83.24 against 83.33 / 82.76 sub-grids/s on one box: the
// vector-memory front end is not what the kernel waits on either.
// LDS: two 16-KB stages of [chunk-in-pair 2][plane 4 = hi/lo x channel half][128 rows] 16-byte slots (a ds_read_b128 lane
// group covers 16 consecutive slots), refilled one pair ahead: global loads for pair p+1 are issued in step 0 of
// pair p (behind that step's weight-fragment request: the in-order vector-memory queue, see the main loop) and committed to LDS after the
// pair's MFMAs, one barrier per pair.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace mica {

typedef float floatx4c __attribute__((ext_vector_type(4)));

constexpr int C1_ROWS = 128;
constexpr int C1_NT = 256;                              // threads per workgroup
constexpr int C1_STAGE = 2 * 4 * C1_ROWS * 16;          // bytes per stage
constexpr int C1_TS = 68;                               // floats per row of the epilogue staging tile (64 + pad)
constexpr int C1_TBYTES = C1_ROWS * C1_TS * 4;
constexpr int C1_TAB = 2 * 512 * 2 * 4;                 // (mean, rstd) of up to 512 channels for two sources: the largest table
constexpr int C1_BASE = (C1_TBYTES > 2 * C1_STAGE ? C1_TBYTES : 2 * C1_STAGE);      // stages / epilogue tile
constexpr int C1_LDS = C1_BASE + C1_TAB;
// The norm table is sized per launch (tab_stride = channels of the widest RAW source, none for split sources): with the full 8 KB
// a workgroup needs 42.8 KB and three fit a CU; a 512-channel single source needs 4 KB (38.9 KB: four fit), the stem fusion none.

__device__ __forceinline__ void c1_split8(const float (&y)[8], half8& hi, half8& lo, int& bad, float ascale) {
    mica_split8(y, hi, lo, bad, ascale);
}

// WINO: 0 = raw f32 output, 1 = the F(2,3) operand, 2 = the F(4,3) operand (kernels_conv43.hip), encoded at out_ascale
template <int NCT, int WINO>
__global__ __launch_bounds__(C1_NT, NCT == 1 ? 4 : NCT == 2 ? 3 : 2) void conv1x1_kernel(Conv1Srcs src, const _Float16* __restrict__ wpk, int64_t wpk_bstride,
                                                      const float* __restrict__ bias, float out_scale, float* __restrict__ out_raw,
                                                      SplitView wino, Dims d, int cout, int total_chunks, SplitEnc enc, float out_ascale, int tab_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* tab = reinterpret_cast<float2*>(smem + C1_BASE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;
    const int b = blockIdx.y;
    const int V = d.D * d.H * d.W;
    const int v0 = blockIdx.x * C1_ROWS;
    const int lr = lane & 15, lg = lane >> 4;

    // InstanceNorm constants of the raw sources
    for (int si = 0; si < src.n; ++si) {
        const Conv1Src& s = src.s[si];
        if (s.kind == 1) {
            const int Cs = s.chunks_total * 16;
            for (int ch = tid; ch < Cs; ch += C1_NT)
                tab[si * tab_stride + ch] = s.mean ? make_float2(s.mean[(int64_t)b * Cs + ch], s.rstd[(int64_t)b * Cs + ch]) : make_float2(0.f, 1.f);
        }
    }

    // ---- staging of one chunk pair: fetch (global -> registers), commit (registers -> LDS, normalising raw sources) ----
    struct Where { int si, lc; };
    auto where = [&](int gch) {
        Where w{0, gch};
        if (src.n > 1 && gch >= src.s[0].chunks) { w.si = 1; w.lc = gch - src.s[0].chunks; }
        return w;
    };
    float4 st[2][2];
    // Exactly four 16-byte loads per thread and pair, whatever the source kind, the tile's raggedness or the parity of the chunk count
    // (rows beyond the volume and a missing second chunk re-read a valid address; commit() writes zeros for them): the number of loads in
    // flight is then a compile-time fact, and the waits on the weight fragments can leave these loads outstanding (`s_waitcnt vmcnt(n)`
    // with n > 0) - behind a per-lane `if (v < V)` the compiler has to assume none was issued and waits for everything.
    auto fetch = [&](int pair) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int gch = min(2 * pair + cc, total_chunks - 1);
            const Where w = where(gch);
            const Conv1Src& s = src.s[w.si];
            if (s.kind == 1) {                               // raw: item = (row, 8-channel half)
                const int row = tid >> 1, kh = tid & 1;
                const int v = min(v0 + row, V - 1);
                // plain NDHWC [V][Cs], or the blocked raw layout [Cs / cblk][V][cblk] (a chunk pair = one 32-channel block)
                // (cblk is 0 or 32 - launch_conv1x1 refuses anything else -: no division in the address)
                const int Cs = s.chunks_total * 16, ch = (s.chunk_off + w.lc) * 16 + kh * 8;
                const bool blocked = s.cblk == 32;
                const int cb = blocked ? 32 : Cs, blk = blocked ? ch >> 5 : 0, within = blocked ? ch & 31 : ch;
                const float* p = reinterpret_cast<const float*>(s.p) + (int64_t)b * V * Cs + (int64_t)blk * V * cb + (int64_t)v * cb + within;
                st[cc][0] = *reinterpret_cast<const float4*>(p);
                st[cc][1] = *reinterpret_cast<const float4*>(p + 4);
            } else {                                          // split: two 16-byte pieces, piece = (row, plane)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int id = tid + C1_NT * k, plane = id & 3, row = id >> 2;
                    const int v = min(v0 + row, V - 1);
                    st[cc][k] = *reinterpret_cast<const float4*>(reinterpret_cast<const _Float16*>(s.p) +
                                                                 (((int64_t)b * s.chunks_total + s.chunk_off + w.lc) * V + v) * 32 + plane * 8);
                }
            }
        }
    };
    int bad = 0;
    auto commit = [&](int pair, char* buf) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int gch = 2 * pair + cc;
            const Where w = where(gch < total_chunks ? gch : 0);
            const Conv1Src& s = src.s[w.si];
            char* cb = buf + cc * (4 * C1_ROWS * 16);
            if (gch < total_chunks && s.kind == 1) {
                const int row = tid >> 1, kh = tid & 1;
                const int ch0 = (s.chunk_off + w.lc) * 16 + kh * 8;
                float y[8] = {st[cc][0].x, st[cc][0].y, st[cc][0].z, st[cc][0].w, st[cc][1].x, st[cc][1].y, st[cc][1].z, st[cc][1].w};
                const bool live = v0 + row < V;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 mr = tab[w.si * tab_stride + ch0 + j];
                    float t = (y[j] - mr.x) * mr.y;
                    if (s.relu) t = fmaxf(t, 0.f);
                    y[j] = live ? t : 0.f;
                }
                half8 hi, lo;
                c1_split8(y, hi, lo, bad, enc.ascale);
                *reinterpret_cast<half8*>(cb + ((kh)*C1_ROWS + row) * 16) = hi;
                *reinterpret_cast<half8*>(cb + ((2 + kh) * C1_ROWS + row) * 16) = lo;
            } else {                                          // split source, or the missing second chunk of an odd count (zeros)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int id = tid + C1_NT * k, plane = id & 3, row = id >> 2;
                    const bool live = gch < total_chunks && v0 + row < V;
                    *reinterpret_cast<float4*>(cb + (plane * C1_ROWS + row) * 16) = live ? st[cc][k] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };

    floatx4c acc[8][NCT];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[f][c][i] = 0.f;

    const int npairs = (total_chunks + 1) >> 1;
    const _Float16* wb = wpk + (int64_t)b * wpk_bstride;
    const int rowbase = lr;

    // weight fragments of (pair, kind): kind 0 = X(c), 1 = X(c'), 2 = Y.  They come straight from L1/L2 and are fetched two steps
    // ahead (three register sets), so that their latency hides behind the MFMAs of the two steps before
    auto bload = [&](int pair, int kind, half8 (&bq)[NCT]) {
        const int c0 = 2 * pair, c1 = (2 * pair + 1 < total_chunks) ? 2 * pair + 1 : 2 * pair;
        const int gc = kind == 0 ? c0 : kind == 1 ? c1 : ((lg >> 1) ? c1 : c0);
        const int q = (kind == 2 ? 2 : 0) + (lg & 1);
        const _Float16* wp = wb + ((int64_t)(gc * 4 + q) * cout + wn * 16 + lr) * 8;
#pragma unroll
        for (int c = 0; c < NCT; ++c) bq[c] = *reinterpret_cast<const half8*>(wp + (int64_t)c * 64 * 8);
    };
    half8 bq[3][NCT];

    // Order of the vector-memory queue (round 5).  Loads retire in order: waiting for a weight fragment also waits for every load
    // issued before it.  With the activation fetch of pair p+1 issued at the top of pair p, BEFORE the request for step 1's fragments,
    // the wait at the start of step 1 forced the whole fetch to have landed after ONE step of MFMAs (a third of a pair) - the prefetch
    // distance the kernel really had.  Now the fragments run two steps ahead (three register sets, set = step of the pair) and the
    // fetch is issued right behind the request made in step 0: the first wait that covers it is the one for the fragments requested in
    // step 1, which are needed at step 0 of the NEXT pair - the fetch has the whole pair, as the commit at the end of it always assumed.
    // The loop is peeled: every pair but the last issues the same, unconditional set of loads (fragments of step 2, the next pair's
    // activations, the next pair's first two fragment sets) - behind a uniform `if (p + 1 < npairs)` the compiler has to assume the loads
    // were NOT issued and its waits fall back to vmcnt(0), which is the one-step prefetch distance again.
    const auto kind_of = [&](int q, int stp) { return 2 * q + 1 < total_chunks ? stp : (stp == 0 ? 0 : 2); };
    auto run_pair = [&](int p, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        char* cur = smem + (p & 1) * C1_STAGE;
        const bool has2 = LAST ? 2 * p + 1 < total_chunks : true;     // only the last pair of an odd chunk count lacks its second chunk
        __syncthreads();                              // stage p is complete; nobody still reads the stage that commit(p+1) will overwrite
        // three steps per pair: X(c), X(c'), Y - or X(c), Y when the second chunk is missing
#pragma unroll
        for (int stp = 0; stp < 3; ++stp) {
            const int kind = has2 ? stp : (stp == 0 ? 0 : 2);
            if (!has2 && stp == 2) break;
            // request the fragments of the step after the next one (set (stp + 2) % 3)
            if (stp == 0) {
                if (has2) bload(p, 2, bq[2]);
                if (!LAST) {
                    __builtin_amdgcn_sched_barrier(0);
                    fetch(p + 1);                     // behind step 0's fragment request: see above
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (!LAST) bload(p + 1, kind_of(p + 1, stp - 1), bq[stp - 1]);
            // A: X(cc) reads plane lg of chunk cc; Y reads the hi plane (lg & 1) of chunk (lg >> 1)
            const int aplane = kind == 0 ? lg : kind == 1 ? 4 + lg : (lg >> 1) * 4 + (lg & 1);
            const char* ap = cur + (aplane * C1_ROWS + rowbase) * 16;
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const half8 a = *reinterpret_cast<const half8*>(ap + f * 16 * 16);
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq[stp][c], acc[f][c], 0, 0, 0);
                if (f & 1) __builtin_amdgcn_sched_barrier(0);     // keeps the LDS reads from being hoisted over the whole step (registers)
            }
        }
        if (!LAST) commit(p + 1, smem + ((p + 1) & 1) * C1_STAGE);
    };
    __syncthreads();                                  // norm table ready
    fetch(0);
    commit(0, smem);
    bload(0, kind_of(0, 0), bq[0]);
    bload(0, kind_of(0, 1), bq[1]);
    for (int p = 0; p + 1 < npairs; ++p) run_pair(p, std::false_type{});
    run_pair(npairs - 1, std::true_type{});
    if (bad) atomicOr(enc.err + b, bad);

    // ---- epilogue: 64 output channels per pass through the staging tile T[128][64 (+4)] ----
    float* T = reinterpret_cast<float*>(smem);
    const int Wh = (d.W + 1) >> 1, Vh = d.D * d.H * Wh;
    int bad2 = 0;
#pragma unroll
    for (int pass = 0; pass < NCT; ++pass) {
        __syncthreads();                              // MFMA phase (or the previous pass's readers) done with this LDS
        {
            const int n = (pass * 4 + wn) * 16 + lr;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int f = 0; f < 8; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i)           // C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + i
                    T[(f * 16 + lg * 4 + i) * C1_TS + wn * 16 + lr] = acc[f][pass][i] * out_scale + bv;
        }
        __syncthreads();
        if (WINO == 0) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int id = it * C1_NT + tid, cg = id & 15, row = id >> 4;
                const int v = v0 + row;
                if (v < V)
                    *reinterpret_cast<float4*>(out_raw + ((int64_t)b * V + v) * cout + pass * 64 + cg * 4) =
                        *reinterpret_cast<const float4*>(T + row * C1_TS + cg * 4);
            }
        } else if (WINO == 2) {
            // F(4,3) input transform of the consumer (kernels_conv43.hip: prep_wino43_kernel): per output quad (x = 4i .. 4i+3) and
            // d_k = y(4i-1+k), k = 0..5, zero outside the row; 32 quads x 8 channel groups = one task per thread
            const int Wq = (d.W + 3) >> 2, Vq = d.D * d.H * Wq;
            {
                const int phl = tid & 31, kg = tid >> 5;
                const int yr = phl / Wq, i = phl - yr * Wq;
                const int vrow = v0 + yr * d.W;       // first voxel of this x row
                if (vrow < V) {
                    float dv[6][8];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int xx = 4 * i - 1 + k;
                        const bool ok = (unsigned)xx < (unsigned)d.W;
#ifdef MICA_C1_LINEAR_READS
                        // timing experiment (results are garbage): neighbouring lanes read NEIGHBOURING rows of the tile (17 slots apart =
                        // consecutive bank quads: conflict-free under any lane grouping) instead of rows four apart (4-way on ds_read_b128)
                        const float* tp = T + ((phl + k) & (C1_ROWS - 1)) * C1_TS + kg * 8;
#else
                        const float* tp = T + (yr * d.W + (ok ? xx : 0)) * C1_TS + kg * 8;
#endif
                        const float4 a = *reinterpret_cast<const float4*>(tp), c = *reinterpret_cast<const float4*>(tp + 4);
                        dv[k][0] = ok ? a.x : 0.f; dv[k][1] = ok ? a.y : 0.f; dv[k][2] = ok ? a.z : 0.f; dv[k][3] = ok ? a.w : 0.f;
                        dv[k][4] = ok ? c.x : 0.f; dv[k][5] = ok ? c.y : 0.f; dv[k][6] = ok ? c.z : 0.f; dv[k][7] = ok ? c.w : 0.f;
                    }
                    float t[6][8];
                    wino43_input_transform(dv, t);
                    const int chunk = pass * 4 + (kg >> 1), kh = kg & 1;
                    const int64_t ph = (int64_t)(vrow / d.W) * Wq + i;
                    _Float16* wbp = wino.p + (((int64_t)b * wino.chunks_total + wino.chunk_off + chunk) * (int64_t)Vq) * 192;
#pragma unroll
                    for (int pp = 0; pp < 6; ++pp) {
                        half8 hi, lo;
                        c1_split8(t[pp], hi, lo, bad2, out_ascale);
                        *reinterpret_cast<half8*>(wbp + ((int64_t)(pp * 4 + kh) * Vq + ph) * 8) = hi;
                        *reinterpret_cast<half8*>(wbp + ((int64_t)(pp * 4 + 2 + kh) * Vq + ph) * 8) = lo;
                    }
                }
            }
        } else {
            // Winograd input transform of the consumer (kernels_elem.hip: prep_wino_kernel): per output pair (x = 2i, 2i+1) and
            // d_k = y(2i-1+k), zero outside the row:  t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int id = it * C1_NT + tid, phl = id & 63, kg = id >> 6;
                const int yr = phl / Wh, i = phl - yr * Wh;
                const int vrow = v0 + yr * d.W;       // first voxel of this x row
                if (vrow < V) {
                    float dv[4][8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int xx = 2 * i - 1 + k;
                        const bool ok = (unsigned)xx < (unsigned)d.W;
#ifdef MICA_C1_LINEAR_READS
                        const float* tp = T + ((phl + k) & (C1_ROWS - 1)) * C1_TS + kg * 8;      // rows one apart instead of two (2-way)
#else
                        const float* tp = T + (yr * d.W + (ok ? xx : 0)) * C1_TS + kg * 8;
#endif
                        const float4 a = *reinterpret_cast<const float4*>(tp), c = *reinterpret_cast<const float4*>(tp + 4);
                        dv[k][0] = ok ? a.x : 0.f; dv[k][1] = ok ? a.y : 0.f; dv[k][2] = ok ? a.z : 0.f; dv[k][3] = ok ? a.w : 0.f;
                        dv[k][4] = ok ? c.x : 0.f; dv[k][5] = ok ? c.y : 0.f; dv[k][6] = ok ? c.z : 0.f; dv[k][7] = ok ? c.w : 0.f;
                    }
                    float t[4][8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        t[0][j] = dv[0][j] - dv[2][j];
                        t[1][j] = dv[1][j] + dv[2][j];
                        t[2][j] = dv[2][j] - dv[1][j];
                        t[3][j] = dv[1][j] - dv[3][j];
                    }
                    const int chunk = pass * 4 + (kg >> 1), kh = kg & 1;
                    const int64_t ph = (int64_t)(vrow / d.W) * Wh + i;
                    _Float16* wbp = wino.p + (((int64_t)b * wino.chunks_total + wino.chunk_off + chunk) * 4 * Vh) * 32;
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        half8 hi, lo;
                        c1_split8(t[pp], hi, lo, bad2, out_ascale);
                        *reinterpret_cast<half8*>(wbp + ((int64_t)(pp * 4 + kh) * Vh + ph) * 8) = hi;
                        *reinterpret_cast<half8*>(wbp + ((int64_t)(pp * 4 + 2 + kh) * Vh + ph) * 8) = lo;
                    }
                }
            }
        }
    }
    if (bad2) atomicOr(enc.err + b, bad2);
}


bool conv1x1_can_emit_wino(Dims d, int wino_kind) {
    const int m = wino_kind == 2 ? 4 : 2;          // whole x rows per workgroup and whole output pairs / quads per row
    return d.W >= m && d.W <= 64 && C1_ROWS % d.W == 0 && d.W % m == 0;
}

template <int NCT, int WINO>
static void launch_conv1x1_t(const Conv1Srcs& src, const _Float16* wpk, int64_t wpk_bstride, const float* bias, float out_scale,
                             float* out_raw, SplitView wino, int B, Dims d, int cout, int total, SplitEnc enc, float out_ascale, hipStream_t st) {
    static PerDeviceOnce once;
    once.run([&](int) { (void)hipFuncSetAttribute((const void*)conv1x1_kernel<NCT, WINO>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS); });
    const int V = d.D * d.H * d.W;
    dim3 grid((V + C1_ROWS - 1) / C1_ROWS, B);
    int tab_stride = 0;                                   // channels of the widest raw source (multiple of 16, <= 512)
    for (int i = 0; i < src.n; ++i)
        if (src.s[i].kind == 1 && src.s[i].chunks_total * 16 > tab_stride) tab_stride = src.s[i].chunks_total * 16;
    const int lds = C1_BASE + src.n * tab_stride * (int)sizeof(float2);
    hipLaunchKernelGGL((conv1x1_kernel<NCT, WINO>), grid, dim3(C1_NT), lds, st, src, wpk, wpk_bstride, bias, out_scale, out_raw, wino, d, cout,
                       total, enc, out_ascale, tab_stride);
}

// cout in {64, 128, 256}.  Exactly one of out_raw / wino.p is given; wino needs conv1x1_can_emit_wino(d).
void launch_conv1x1(const Conv1Srcs& src, const _Float16* wpk, int64_t wpk_bstride, const float* bias, float out_scale, float* out_raw,
                    SplitView wino, int B, Dims d, int cout, SplitEnc enc, hipStream_t st, int wino_kind, float enc_out_ascale) {
    const float oasc = enc_out_ascale > 0.f ? enc_out_ascale : enc.ascale;
    int total = 0;
    for (int i = 0; i < src.n; ++i) {
        total += src.s[i].chunks;
        // the norm table in LDS holds 512 channels per raw source (the widest tensor of the network); shapes are fixed by the
        // forward graph and checked by the single-op entry points
        if (src.s[i].kind == 1 && src.s[i].chunks_total * 16 > 512) { refuse_launch("conv1x1: raw source wider than 512 channels"); return; }
        if (src.s[i].kind == 1 && src.s[i].cblk != 0 && (src.s[i].cblk != 32 || src.s[i].chunks_total % 2)) { refuse_launch("conv1x1: a blocked raw source has 32-channel blocks"); return; }
    }
    const int w = wino.p != nullptr ? (wino_kind == 2 ? 2 : 1) : 0;
#define C1_GO(NCT)                                                                                                                      \
    do {                                                                                                                                \
        if (w == 2) launch_conv1x1_t<NCT, 2>(src, wpk, wpk_bstride, bias, out_scale, out_raw, wino, B, d, cout, total, enc, oasc, st);    \
        else if (w == 1) launch_conv1x1_t<NCT, 1>(src, wpk, wpk_bstride, bias, out_scale, out_raw, wino, B, d, cout, total, enc, oasc, st); \
        else launch_conv1x1_t<NCT, 0>(src, wpk, wpk_bstride, bias, out_scale, out_raw, wino, B, d, cout, total, enc, oasc, st);           \
    } while (0)
    if (cout == 64) C1_GO(1);
    else if (cout == 128) C1_GO(2);
    else C1_GO(4);
#undef C1_GO
}

}  // namespace mica

// Multi-scale stem (reference models/model.py:9-14,49-51: four Conv3d(1, 32, k), k = 3, 5, 7, 9, on the density tile) on the matrix
// cores.  Cin = 1, so the GEMM's K dimension is the taps: out[voxel][ch] = sum_tap in[voxel + tap] * w[tap][ch], with the split-f16
// products of the other convs (x * w = w_hi x_hi + w_hi x_lo + w_lo x_hi, f32 accumulation, v_mfma_f32_16x16x32_f16).
//
// The obstacle is the im2col operand: a lane needs 8 consecutive taps along x of ITS voxel, i.e. 16 bytes starting at a 2-byte
// granular LDS address.  It goes away when the shift moves into the weights:
//   * wave c of a workgroup owns the outputs x = x0 + c + 8 r (r = 0..7) - one residue class mod 8;
//   * K runs over ALIGNED 8-blocks of the input row: for output block r the blocks r - 1, r, r + 1 (delta = -1, 0, +1) cover its
//     window, and the fragment of (kz, ky, delta) is an aligned ds_read_b128 at xi = 8 (r + delta) for every row r;
//   * the weights are packed per class: W[c][(kz, ky, delta)][i] = w[kz][ky][kx = 8 delta + i - c + R] or 0 (R = k / 2).  A class
//     needs one or two blocks per (kz, ky) (k = 9: always two; k = 7: one for c = 3, 4; k = 5: one for c = 2..5; k = 3: one for
//     c = 1..6), so 50-56 % of the K slots carry a tap.
// MFMA roles: A = weights (16 channels x 32 K), B = im2col (32 K x 16 voxels), D[channel][voxel]: with the rows of the two channel
// tiles interleaved in the packing a lane ends up with 8 consecutive channels of one voxel - 16-byte pieces of the split records.
//
// Workgroup = 8 waves (one per class) x a 64(x) x 8(y) x 2(z) block of outputs; the input tile with halo 4 (x: the aligned range
// -8 .. 71) sits in LDS as f16 hi and lo planes (2 x 25 KB).  Per K-step of 32 slots a wave issues 16 ds_read_b128, 4 weight loads
// (L1 / L2: 2.4 MB of packed records shared by every workgroup) and 48 MFMAs for its 128 voxels x 32 channels.  The waves of a SIMD
// are classes c and c + 4: 148-158 K-steps per block on every SIMD.  Two workgroups share a CU (128 registers, 54 KB of LDS each): one's
// tile fill and epilogue run beside the other's MFMAs, and nothing is double-buffered inside a wave.
// Measured (bench.py, 8 tiles of 64^3 per launch, rocprofv3 kernel trace): 0.68 ms against 1.86 ms for the f32 VALU kernel it replaces
// (60 M MFMAs = 0.39 ms of issue at 2.4 GHz).  Ablations (-DMICA_STEM_NOLOOP: one K-step per kernel size; -DMICA_STEM_NOEPI: no
// output) on the one-workgroup-per-CU version of 0.94 ms: 0.37 ms and 0.70 ms - the K loop was 0.57 ms (69 % MFMA-busy), the output
// (1.07 GB of split records in 8-byte pieces) 0.24 ms, tile fill and launch 0.13 ms, all in sequence; pinning the weight prefetch,
// batching the tile fill's loads, a conflict-free fragment mapping (y and y + 4 in one fragment) and all sixteen fragment reads up
// front each changed nothing there; sharing the CU between two workgroups took it to 0.80 ms, removing that version's 26 spilled registers
// (fragment addresses hoisted out of the size loop) to 0.73 ms, 16-byte instead of 8-byte output pieces to 0.68 ms.
// Used for tile widths that are multiples of 64 (the production tile); other widths take the f32 VALU kernel (kernels_conv.hip:
// stem_kernel), which writes the same formats.
#include "common.h"
#include <vector>

namespace mica {

typedef float floatx4s __attribute__((ext_vector_type(4)));
typedef _Float16 half4s __attribute__((ext_vector_type(4)));

constexpr int SM_Y = 8, SM_Z = 2, SM_H = 4;
constexpr int SM_ROWH = 80, SM_ROWB = SM_ROWH * 2;                 // halves / bytes per tile row: xi = -8 .. 71
constexpr int SM_YR = SM_Y + 2 * SM_H, SM_ZR = SM_Z + 2 * SM_H;    // 16 x 10 rows
constexpr int SM_PLANE = SM_ZR * SM_YR * SM_ROWB;                  // bytes per plane (hi or lo): 25 600

// ---- host: the K-step records of every (kernel size, class) --------------------------------------------------------------------------
// w: [size][tap (kz, ky, kx)][32] as uploaded for the VALU kernel.  wf: [record][t 2][lane 64][8] f32 weight fragments (scaled and split
// on the device), aoff: [record][4] byte offsets of the K-groups' input blocks inside a tile plane.
void stem_mfma_plan(const float* w, std::vector<float>& wf, std::vector<int>& aoff, StemPlan& plan) {
    const int ks[4] = {3, 5, 7, 9};
    wf.clear();
    aoff.clear();
    int rec = 0;
    size_t woff = 0;
    for (int sz = 0; sz < 4; ++sz) {
        const int k = ks[sz], R = k / 2;
        for (int c = 0; c < 8; ++c) {
            std::vector<int> deltas;
            for (int dl = -1; dl <= 1; ++dl) {
                bool any = false;
                for (int i = 0; i < 8; ++i) {
                    const int kx = 8 * dl + i - c + R;
                    any |= kx >= 0 && kx < k;
                }
                if (any) deltas.push_back(dl);
            }
            struct Grp { int kz, ky, dl; };
            std::vector<Grp> groups;
            for (int kz = 0; kz < k; ++kz)
                for (int ky = 0; ky < k; ++ky)
                    for (int dl : deltas) groups.push_back(Grp{kz, ky, dl});
            const int steps = ((int)groups.size() + 3) / 4;
            plan.first[sz][c] = rec;
            plan.steps[sz][c] = steps;
            for (int s = 0; s < steps; ++s, ++rec) {
                for (int g = 0; g < 4; ++g) {
                    const int G = 4 * s + g;
                    int off = 0;
                    if (G < (int)groups.size()) {
                        const Grp& q = groups[G];
                        off = ((q.kz - R + SM_H) * SM_YR + (q.ky - R + SM_H)) * SM_ROWB + (q.dl + 1) * 16;
                    }
                    aoff.push_back(off);
                }
                for (int t = 0; t < 2; ++t)
                    for (int lane = 0; lane < 64; ++lane) {
                        // MFMA row m of channel tile t carries channel 8 (m >> 2) + 4 t + (m & 3): a lane's rows 4 g .. 4 g + 3 of the two
                        // tiles are then the 8 consecutive channels 8 g .. 8 g + 7 (16-byte pieces of the output records)
                        const int m = lane & 15, g = lane >> 4, G = 4 * s + g;
                        for (int i = 0; i < 8; ++i) {
                            float v = 0.f;
                            if (G < (int)groups.size()) {
                                const Grp& q = groups[G];
                                const int kx = 8 * q.dl + i - c + R;
                                if (kx >= 0 && kx < k) v = w[woff + ((size_t)(q.kz * k + q.ky) * k + kx) * 32 + 8 * (m >> 2) + 4 * t + (m & 3)];
                            }
                            wf.push_back(v);
                        }
                    }
            }
        }
        woff += (size_t)k * k * k * 32;
    }
    plan.records = rec;
}

// wf [record][t][lane][8] f32 -> records [record][hi t0 | hi t1 | lo t0 | lo t1][lane][8] f16 of w * wscale
__global__ void stem_mfma_pack_kernel(const float* __restrict__ wf, int64_t n, float wscale, _Float16* __restrict__ rec) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (record, t, lane, i)
    if (e >= n) return;
    const int i = (int)(e & 7), lane = (int)(e >> 3) & 63, t = (int)(e >> 9) & 1;
    const int64_t r = e >> 10;
    const float v = wf[e] * wscale;
    const _Float16 h = (_Float16)v;
    rec[((r * 4 + t) * 64 + lane) * 8 + i] = h;
    rec[((r * 4 + 2 + t) * 64 + lane) * 8 + i] = (_Float16)(v - (float)h);
}
void launch_stem_mfma_pack(const float* d_wf, int records, float wscale, _Float16* d_rec, hipStream_t st) {
    const int64_t n = (int64_t)records * 2 * 64 * 8;
    hipLaunchKernelGGL(stem_mfma_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_wf, n, wscale, d_rec);
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 4) void stem_mfma_kernel(const float* __restrict__ map, Dims d, const _Float16* __restrict__ wrec,
                                                        const int* __restrict__ aoff, StemPlan plan, const float* __restrict__ bstem,
                                                        float out_scale, SplitView out, float* __restrict__ out_raw,
                                                        float* __restrict__ ws, int ntx, int nty, SplitEnc enc) {
    __shared__ __attribute__((aligned(16))) char smem[2 * SM_PLANE];     // [hi | lo][z row 10][y row 16][80 halves]
    __shared__ float csum[8][128];                                       // per-wave channel sums (fixed order => deterministic)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int V = d.D * d.H * d.W;
    const int tb = blockIdx.x;
    const int tx = tb % ntx, ty = (tb / ntx) % nty, tz = tb / (ntx * nty);
    const int x0 = tx * 64, y0 = ty * SM_Y, z0 = tz * SM_Z;
    const float ascale = enc.ascale;
    int bad = 0;

    // input tile -> split f16 planes (x * ascale = hi + lo); values beyond the f16 range raise the tile's flag like every other encoder
    {
        const float* mb = map + (int64_t)b * V;
        _Float16* th = reinterpret_cast<_Float16*>(smem);
        _Float16* tl = reinterpret_cast<_Float16*>(smem + SM_PLANE);
        // all 25 loads of a thread are requested before the first is used (one memory round trip per tile, not 25)
        constexpr int NE = SM_ZR * SM_YR * SM_ROWH / 512;
        static_assert(NE * 512 == SM_ZR * SM_YR * SM_ROWH, "the tile is a whole number of passes of the workgroup");
        float v[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int i = tid + 512 * k;
            const int col = i % SM_ROWH, row = i / SM_ROWH;
            const int yl = row % SM_YR, zl = row / SM_YR;
            const int gx = x0 + col - 8, gy = y0 + yl - SM_H, gz = z0 + zl - SM_H;
            const bool in = (unsigned)gx < (unsigned)d.W && (unsigned)gy < (unsigned)d.H && (unsigned)gz < (unsigned)d.D;
            v[k] = in ? mb[(int64_t)(gz * d.H + gy) * d.W + gx] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int i = tid + 512 * k;
            float xs = v[k] * ascale;
            if (!(fabsf(xs) <= F16_LIMIT)) {
                bad |= (fabsf(v[k]) <= 3.0e38f) ? RANGE_OVERFLOW : RANGE_NONFINITE;
                xs = fminf(fmaxf(xs, -F16_LIMIT), F16_LIMIT);
            }
            const _Float16 h = (_Float16)xs;
            th[i] = h;
            tl[i] = (_Float16)(xs - (float)h);
        }
    }
    __syncthreads();

    const int c = wave;                                   // residue class of this wave's outputs: x = x0 + c + 8 r
    const int n = lane & 15, g = lane >> 4;               // MFMA column (voxel of the fragment) and K-group / row group
    // voxel of a fragment: x block r, y = p or p + 4 (fragment f = (p = f & 3, z = f >> 2)): rows four apart are 640 B = 128 B (mod 256)
    // apart, so the 16 lanes of a ds_read_b128 group cover all 64 banks once (rows y, y + 1 at 160 B shared eight banks)
    const int r = n & 7, yy = n >> 3;
    const char* xlane = smem + yy * 4 * SM_ROWB + r * 16;

#pragma unroll 1
    for (int sz = 0; sz < 4; ++sz) {
        floatx4s acc[8][2];
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[f][t][i] = 0.f;
#ifdef MICA_STEM_NOLOOP
        const int nst = 1, rec0 = plan.first[sz][c];          // ablation: one K-step per kernel size
#else
        const int nst = plan.steps[sz][c], rec0 = plan.first[sz][c];
#endif
        const half8* wp = reinterpret_cast<const half8*>(wrec) + (int64_t)rec0 * 4 * 64 + lane;
        const int* ap = aoff + rec0 * 4 + g;
#pragma unroll 1
        for (int s = 0; s < nst; ++s) {
            // two workgroups share a CU (four waves per SIMD): the other waves' MFMAs cover this wave's weight and fragment reads, so
            // nothing is double-buffered and the kernel stays within 128 registers
            half8 wc[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) wc[p] = wp[(s * 4 + p) * 64];
            const int ao = ap[s * 4];
            const char* xb = xlane + ao;
            // two fragments at a time: the three products on one accumulator sit four MFMAs apart
#pragma unroll
            for (int f = 0; f < 8; f += 2) {
                half8 xh[2], xl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int fo = (((f + u) >> 2) * SM_YR + ((f + u) & 3)) * SM_ROWB;
                    xh[u] = *reinterpret_cast<const half8*>(xb + fo);
                    xl[u] = *reinterpret_cast<const half8*>(xb + SM_PLANE + fo);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[f + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[t], xh[u], acc[f + u][t], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[f + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[t], xl[u], acc[f + u][t], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[f + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[2 + t], xh[u], acc[f + u][t], 0, 0, 0);
            }
        }

        // ---- this kernel size's 32 channels of the wave's 128 voxels: bias, split records / raw, channel sums ----
        // C/D map of the 16x16 MFMA: column = lane & 15 (voxel), rows (lane >> 4) * 4 + i (channels 4 g + i of the 16-channel tile t)
        // Everything is formed at the operand scale: xs = acc / wscale + bias * ascale = ascale * (acc * out_scale + bias) exactly (powers
        // of two), so the split encoder needs no multiply of its own; the channel sums and the raw output are scaled back by 1 / ascale.
        // C/D map of the 16x16 MFMA: column = lane & 15 (voxel), rows (lane >> 4) * 4 + i of tile t = channels 8 g + 4 t + i (see the packing)
        float sums[2][4];
        float bv[2][4];
        const float xscale = out_scale * ascale, inv_ascale = 1.0f / ascale;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 q = *reinterpret_cast<const float4*>(bstem + sz * 32 + 8 * g + 4 * t);
            bv[t][0] = q.x * ascale; bv[t][1] = q.y * ascale; bv[t][2] = q.z * ascale; bv[t][3] = q.w * ascale;
#pragma unroll
            for (int i = 0; i < 4; ++i) sums[t][i] = 0.f;
        }
        // laundered per kernel size: as loop invariants the eight fragments' 64-bit output addresses are hoisted out of the size loop
        // into registers that the K loop does not have to spare (26 spilled VGPRs; 124 and none with this)
        int y0e = y0, z0e = z0;
        asm volatile("" : "+s"(y0e), "+s"(z0e));
        const int gx = x0 + c + 8 * r;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int gy = y0e + (f & 3) + 4 * yy, gz = z0e + (f >> 2);
#ifdef MICA_STEM_NOEPI
            if (gy < d.H && gz < 0) {                      // ablation: no output
#else
            if (gy < d.H && gz < d.D) {
#endif
                const int64_t vox = (int64_t)(gz * d.H + gy) * d.W + gx;
                float xs[8], cl[8];
                bool viol = false;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 4 * t + i;
                        xs[j] = fmaf(acc[f][t][i], xscale, bv[t][i]);
                        sums[t][i] += xs[j];
                        cl[j] = __builtin_amdgcn_fmed3f(xs[j], -F16_LIMIT, F16_LIMIT);      // NaN -> -F16_LIMIT
                        viol |= cl[j] != xs[j];
                    }
                if (__builtin_amdgcn_ballot_w64(viol)) {          // wave-uniform slow path: which kind of violation
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (!(fabsf(xs[j]) <= F16_LIMIT)) bad |= (fabsf(xs[j]) <= 3.0e38f) ? RANGE_OVERFLOW : RANGE_NONFINITE;
                }
                if (out.p) {
                    half8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const _Float16 h = (_Float16)cl[j];
                        hi[j] = h;
                        lo[j] = (_Float16)(cl[j] - (float)h);
                    }
                    // channels 8 g .. 8 g + 7 of this kernel size: chunk sz * 2 + (g >> 1), halves (g & 1) * 8 .. of its hi and lo parts
                    _Float16* dst = out.p + (((int64_t)b * out.chunks_total + out.chunk_off + sz * 2 + (g >> 1)) * V + vox) * 32 + (g & 1) * 8;
                    *reinterpret_cast<half8*>(dst) = hi;
                    *reinterpret_cast<half8*>(dst + 16) = lo;
                }
                if (out_raw) {
                    float* dr = out_raw + ((int64_t)b * V + vox) * 128 + sz * 32 + 8 * g;
                    *reinterpret_cast<float4*>(dr) = make_float4(xs[0] * inv_ascale, xs[1] * inv_ascale, xs[2] * inv_ascale, xs[3] * inv_ascale);
                    *reinterpret_cast<float4*>(dr + 4) = make_float4(xs[4] * inv_ascale, xs[5] * inv_ascale, xs[6] * inv_ascale, xs[7] * inv_ascale);
                }
            }
        }
        // channel sums over the wave's voxels: the 16 voxel lanes of a row group combine by DPP rotations inside their row of 16 lanes
#define SM_ROR_ADD(x, k) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (k), 0xf, 0xf, false))
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sv = sums[t][i] * inv_ascale;
                SM_ROR_ADD(sv, 1);
                SM_ROR_ADD(sv, 2);
                SM_ROR_ADD(sv, 4);
                SM_ROR_ADD(sv, 8);
                if (n == 0) csum[wave][sz * 32 + 8 * g + 4 * t + i] = sv;
            }
#undef SM_ROR_ADD
    }
    if (bad && enc.err) atomicOr(enc.err + b, bad);
    __syncthreads();
    if (ws && tid < 128)
        ws[((int64_t)b * gridDim.x + blockIdx.x) * 128 + tid] = ((csum[0][tid] + csum[1][tid]) + (csum[2][tid] + csum[3][tid])) +
                                                               ((csum[4][tid] + csum[5][tid]) + (csum[6][tid] + csum[7][tid]));
}

bool stem_mfma_eligible(Dims d) { return d.W % 64 == 0; }

// Returns the number of per-block channel-sum partials written to ws (when non-null): f32 [B][blocks][128].
int launch_stem_mfma(const float* map, int B, Dims d, const _Float16* wrec, const int* aoff, const StemPlan& plan, float wscale,
                     const float* bstem, SplitView out, float* out_raw, float* ws, SplitEnc enc, hipStream_t st) {
    const int ntx = d.W / 64, nty = (d.H + SM_Y - 1) / SM_Y, ntz = (d.D + SM_Z - 1) / SM_Z;
    dim3 grid(ntx * nty * ntz, B);
    hipLaunchKernelGGL(stem_mfma_kernel, grid, dim3(512), 0, st, map, d, wrec, aoff, plan, bstem, 1.0f / (wscale * enc.ascale), out, out_raw, ws,
                       ntx, nty, enc);
    return (int)grid.x;
}

}  // namespace mica

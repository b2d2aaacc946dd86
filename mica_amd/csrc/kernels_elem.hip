// HBM-bound streaming kernels of the MICA hot path: InstanceNorm statistics and apply(+ReLU, +gate)
// with the split-f16 re-encoding the MFMA conv consumes, squeeze-excite style gate MLPs, head tails,
// softmax/argmax post-processing, tile gather and stitch.  All f32 arithmetic; one pass over the data each.
#include "common.h"
#include <cstdio>

namespace mica {

static inline int pick_blocks(int V, int vox_per_iter, int cap) {
    int nb = (V + vox_per_iter - 1) / vox_per_iter;
    return nb < cap ? (nb < 1 ? 1 : nb) : cap;
}
constexpr int RED_BLOCKS = 1024;   // max partial blocks per batch entry for reductions

// ------------------------------------------------------------------------------------------------
// InstanceNorm3d statistics (model.py:81,108,116,123,143,211,213): per (tile, channel) mean and biased
// variance over all V voxels, eps = 1e-5.  Thread = (8-channel group g, voxel lane); shifted sums per
// thread, Chan merges inside the block (f32) and across blocks (f64, fixed order => deterministic).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void chan_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    float n = na + nb;
    if (nb > 0.f) {
        if (na == 0.f) { na = nb; ma = mb; qa = qb; return; }
        float dlt = mb - ma;
        ma = ma + dlt * (nb / n);
        qa = qa + qb + dlt * dlt * (na * nb / n);
        na = n;
    }
}

__global__ __launch_bounds__(256) void stats_kernel(const float* __restrict__ x, int V, int C, float* __restrict__ ws) {
    extern __shared__ float sh[];   // [3][256][8]
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int G = C >> 3, SUB = 256 / G;
    const int tid = threadIdx.x, g = tid % G, sub = tid / G;
    const int per = (V + nblk - 1) / nblk;
    const int v0 = blk * per, v1 = min(V, v0 + per);
    float K[8], s[8], q[8];
    float n = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { K[j] = 0.f; s[j] = 0.f; q[j] = 0.f; }
    const float* xb = x + (int64_t)b * V * C + g * 8;
    for (int v = v0 + sub; v < v1; v += SUB) {
        float4 a = *reinterpret_cast<const float4*>(xb + (int64_t)v * C);
        float4 c = *reinterpret_cast<const float4*>(xb + (int64_t)v * C + 4);
        float val[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        if (n == 0.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) K[j] = val[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { float t = val[j] - K[j]; s[j] += t; q[j] = fmaf(t, t, q[j]); }
        n += 1.f;
    }
    float* shn = sh; float* shm = sh + 256 * 8; float* shq = sh + 2 * 256 * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float mean = 0.f, m2 = 0.f;
        if (n > 0.f) { mean = K[j] + s[j] / n; m2 = fmaxf(q[j] - s[j] * s[j] / n, 0.f); }
        shn[tid * 8 + j] = n; shm[tid * 8 + j] = mean; shq[tid * 8 + j] = m2;
    }
    __syncthreads();
    for (int off = SUB >> 1; off > 0; off >>= 1) {
        if (sub < off) {
            const int o = (sub + off) * G + g;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float na = shn[tid * 8 + j], ma = shm[tid * 8 + j], qa = shq[tid * 8 + j];
                chan_merge(na, ma, qa, shn[o * 8 + j], shm[o * 8 + j], shq[o * 8 + j]);
                shn[tid * 8 + j] = na; shm[tid * 8 + j] = ma; shq[tid * 8 + j] = qa;
            }
        }
        __syncthreads();
    }
    if (sub == 0) {
        float* w = ws + (((int64_t)b * nblk + blk) * C + g * 8) * 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) { w[j * 3] = shn[tid * 8 + j]; w[j * 3 + 1] = shm[tid * 8 + j]; w[j * 3 + 2] = shq[tid * 8 + j]; }
    }
}

// one workgroup per (tile, channel): 256 threads merge the block partials (f64 Chan merges, fixed thread->partial
// assignment and fixed tree => deterministic)
// Merge of the P partials f32 [B][P][C][3] = (count, mean, M2) per channel, in f64 and in a fixed order.  One block = 4
// channels x 64 slot lanes: for a given slot the 4 channels are 48 contiguous bytes (one block per channel read its
// 12-byte triples at a stride of C * 12 bytes: 64-B sectors for 12 useful bytes).
template <int CL>      // channels per block (CL x 256 / CL slot lanes): per slot the block reads CL * 12 contiguous bytes
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ ws, int nblk, int C, float eps,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             const float* __restrict__ gate) {
    __shared__ double sn[256], sm[256], sq[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    constexpr int SL = 256 / CL;
    const int cl = tid % CL, sl = tid / CL, c = blockIdx.x * CL + cl;
    double n = 0, m = 0, q = 0;
    if (c < C)
        // eight partials are fetched before they are merged: the merge chain is sequential (fixed order), and with one load per
        // iteration the kernel spent its time in 64 dependent global-load latencies (47 -> 30 us; the rest is the traffic itself:
        // up to 200 MB of partials per launch)
        for (int k0 = sl; k0 < nblk; k0 += SL * 8) {
            float pn[8], pm[8], pq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + SL * u;
                const float* w = ws + (((int64_t)b * nblk + (k < nblk ? k : k0)) * C + c) * 3;
                pn[u] = k < nblk ? w[0] : 0.f; pm[u] = w[1]; pq[u] = w[2];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double nb = pn[u], mb = pm[u], qb = pq[u];
                if (nb > 0) {
                    const double nn = n + nb, dl = mb - m, r = nb / nn;      // one division per merge
                    m += dl * r;
                    q += qb + dl * dl * (n * r);
                    n = nn;
                }
            }
        }
    sn[tid] = n; sm[tid] = m; sq[tid] = q;
    __syncthreads();
    for (int off = SL / 2; off > 0; off >>= 1) {
        if (sl < off) {
            const int o = tid + off * CL;
            double nb = sn[o], mb = sm[o], qb = sq[o];
            double na = sn[tid], ma = sm[tid], qa = sq[tid];
            if (nb > 0) {
                if (na == 0) { na = nb; ma = mb; qa = qb; }
                else {
                    const double nn = na + nb, dl = mb - ma, r = nb / nn;
                    ma += dl * r;
                    qa += qb + dl * dl * (na * r);
                    na = nn;
                }
            }
            sn[tid] = na; sm[tid] = ma; sq[tid] = qa;
        }
        __syncthreads();
    }
    if (sl == 0 && c < C) {
        mean[(int64_t)b * C + c] = (float)sm[tid];
        const double g = gate ? (double)gate[(int64_t)b * C + c] : 1.0;
        rstd[(int64_t)b * C + c] = (float)(g / sqrt(g * g * (sq[tid] / sn[tid]) + (double)eps));
    }
}

int64_t stats_ws_floats(int B, int C) { return (int64_t)B * RED_BLOCKS * C * 3; }
void launch_stats_finalize(const float* ws, int B, int P, int C, float eps, float* mean, float* rstd, hipStream_t st, const float* gate) {
    // channels per block by layer width ALONE (C = 512 -> 16, C = 128 / 256 -> 8, narrower -> 4; the call's batch must not enter: the
    // partition decides the merge order, and a tile's numbers do not depend on its batch): wider blocks read longer contiguous
    // runs per slot (C = 512: 132 -> 53 us, 256: 68 -> 35, 128: 33 -> 28) but give every thread a longer sequential merge chain and the
    // chip fewer blocks (16 channels at C = 128: 48 us; one channel per block at C <= 64: 19-35 us against 17-19)
    if (C >= 512) hipLaunchKernelGGL(stats_finalize_kernel<16>, dim3((C + 15) / 16, B), dim3(256), 0, st, ws, P, C, eps, mean, rstd, gate);
    else if (C >= 128) hipLaunchKernelGGL(stats_finalize_kernel<8>, dim3((C + 7) / 8, B), dim3(256), 0, st, ws, P, C, eps, mean, rstd, gate);
    else hipLaunchKernelGGL(stats_finalize_kernel<4>, dim3((C + 3) / 4, B), dim3(256), 0, st, ws, P, C, eps, mean, rstd, gate);
}
// fused statistics: conv_wino writes 4 partials per 16x4x4 output tile (8 for Cout = 32), depthwise one per block
int64_t fused_stats_ws_floats(int B, int S) {
    int64_t tiles = (int64_t)((S + 15) / 16) * ((S + 3) / 4) * ((S + 3) / 4);
    int64_t a = tiles * 4 * 512 * 3, bb = tiles * 8 * 32 * 3;
    int64_t dwp = ((int64_t)S * S * ((S + 7) / 8) + 31) / 32 * 256 * 3;   // depthwise: one partial per 32 x-runs, C <= 256
    int64_t m = a > bb ? a : bb;
    if (dwp > m) m = dwp;
    // conv_wino43 (kernels_conv43.hip): 4 partials per 32x2x4 output tile, Cout <= 512
    const int64_t t43 = (int64_t)((S + 31) / 32) * ((S + 1) / 2) * ((S + 3) / 4) * 4 * 512 * 3;
    if (t43 > m) m = t43;
    return m * B;
}

void launch_stats(const float* x, int B, int V, int C, float eps, float* mean, float* rstd, float* ws, hipStream_t st) {
    int G = C / 8, SUB = 256 / G;
    int nblk = pick_blocks(V, SUB * 8, RED_BLOCKS);
    hipLaunchKernelGGL(stats_kernel, dim3(nblk, B), dim3(256), 3 * 256 * 8 * sizeof(float), st, x, V, C, ws);
    hipLaunchKernelGGL(stats_finalize_kernel<4>, dim3((C + 3) / 4, B), dim3(256), 0, st, ws, nblk, C, eps, mean, rstd, (const float*)nullptr);
}

__global__ __launch_bounds__(256) void finalize_sum_kernel(const float* __restrict__ ws, int nblocks, int C, float inv,
                                                           float* __restrict__ out) {
    __shared__ double sh[256];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int c = blockIdx.x * 4 + (tid >> 6);
    double s = 0.0;
    if (c < C)
        for (int k = lane; k < nblocks; k += 64) s += (double)ws[((int64_t)b * nblocks + k) * C + c];
    sh[tid] = s;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        if (lane < off) sh[tid] += sh[tid + off];
        __syncthreads();
    }
    if (lane == 0 && c < C) out[(int64_t)b * C + c] = (float)(sh[tid] * inv);
}
void launch_finalize_sum(const float* ws, int B, int nblocks, int C, float inv, float* out, hipStream_t st) {
    hipLaunchKernelGGL(finalize_sum_kernel, dim3((C + 3) / 4, B), dim3(256), 0, st, ws, nblocks, C, inv, out);
}

// ------------------------------------------------------------------------------------------------
// prep: y = relu?((x - mean) * rstd) * scale  ->  split view (the next conv's operand), optional raw
// copy, optional global-average-pool of y (SE / calibration gates, model.py:216,244).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const float (&y)[8], half8& hi, half8& lo, int& bad, float ascale) {
    mica_split8(y, hi, lo, bad, ascale);
}

__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ x, int V, int C,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   int relu, const float* __restrict__ scale, SplitView out,
                                                   float* __restrict__ out_raw, float* __restrict__ ws,
                                                   SplitEnc enc) {
    extern __shared__ float sh[];   // [256][8] for the gap reduction
    // A block covers a slab of at most 64 channels (blockIdx.z): with all C channels per block a wave's stores
    // scatter over C/16 chunk planes in 32-B pieces (2.6 TB/s at C = 512 vs 5.3 TB/s at C = 64 measured).
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int Cs = C < 64 ? C : 64;
    const int G = Cs >> 3, SUB = 256 / G;
    const int tid = threadIdx.x, g = blockIdx.z * (Cs >> 3) + tid % G, sub = tid / G;
    const int per = (V + nblk - 1) / nblk;
    const int v0 = blk * per, v1 = min(V, v0 + per);
    float m[8], r[8], sc[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t ci = (int64_t)b * C + g * 8 + j;
        m[j] = mean ? mean[ci] : 0.f;
        r[j] = rstd ? rstd[ci] : 1.f;
        sc[j] = scale ? scale[ci] : 1.f;
        acc[j] = 0.f;
    }
    int bad = 0;
    const float* xb = x + (int64_t)b * V * C + g * 8;
    _Float16* ob = out.p ? out.p + (((int64_t)b * out.chunks_total + out.chunk_off + (g >> 1)) * V) * 32 + (g & 1) * 8 : nullptr;
    for (int v = v0 + sub; v < v1; v += SUB) {
        float4 a = *reinterpret_cast<const float4*>(xb + (int64_t)v * C);
        float4 c = *reinterpret_cast<const float4*>(xb + (int64_t)v * C + 4);
        float y[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = (y[j] - m[j]) * r[j];
            if (relu) t = fmaxf(t, 0.f);
            y[j] = t * sc[j];
            acc[j] += y[j];
        }
        if (out_raw) {
            float* o = out_raw + ((int64_t)b * V + v) * C + g * 8;
            *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
        if (ob) {
            half8 hi, lo;
            split8(y, hi, lo, bad, enc.ascale);
            *reinterpret_cast<half8*>(ob + (int64_t)v * 32) = hi;
            *reinterpret_cast<half8*>(ob + (int64_t)v * 32 + 16) = lo;
        }
    }
    if (bad) atomicOr(enc.err + b, bad);
    if (ws) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[tid * 8 + j] = acc[j];
        __syncthreads();
        for (int off = SUB >> 1; off > 0; off >>= 1) {
            if (sub < off) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sh[tid * 8 + j] += sh[(tid + off * G) * 8 + j];
            }
            __syncthreads();
        }
        if (sub == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ws[((int64_t)b * nblk + blk) * C + g * 8 + j] = sh[tid * 8 + j];
        }
    }
}

void launch_prep(const float* x, int B, int V, int C, const float* mean, const float* rstd, int relu,
                 const float* scale, SplitView out, float* out_raw, float* gap, float* ws, SplitEnc enc,
                 hipStream_t st) {
    int Cs = C < 64 ? C : 64, G = Cs / 8, SUB = 256 / G;
    int nblk = pick_blocks(V, SUB * 4, gap ? RED_BLOCKS : 4096);
    hipLaunchKernelGGL(prep_kernel, dim3(nblk, B, C / Cs), dim3(256), 256 * 8 * sizeof(float), st, x, V, C, mean, rstd, relu,
                       scale, out, out_raw, gap ? ws : nullptr, enc);
    if (gap) launch_finalize_sum(ws, B, nblk, C, 1.0f / (float)V, gap, st);
}

// ------------------------------------------------------------------------------------------------
// prep for Winograd F(2,3) convs: the same y = relu?((x-mean)*rstd)*scale, written as the x-direction
// input transform of each output pair (x = 2i, 2i+1):  with d_k = y(2i-1+k), zero outside the volume,
//   t0 = d0 - d2,  t1 = d1 + d2,  t2 = d2 - d1,  t3 = d1 - d3
// in "wino" layout  _Float16 [B][chunks][4 (p)][4 (q)][Vh][8],  Vh = D*H*ceil(W/2)  (2x the plain bytes); q = hi|lo x channel
// half of the chunk.  Sixteen planes of 16-byte pieces per chunk: consecutive pairs are consecutive 16-byte pieces, so
// this kernel's stores and the conv's slab DMA both move whole 128-byte lines per instruction (with 64-byte
// [hi 16 | lo 16] records per pair an instruction touched a quarter to a half of every line it named).
// Thread = (8-channel group, pair); its own voxels are d1, d2 (plain / raw / gap outputs use those).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_wino_kernel(const float* __restrict__ x, Dims d, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        int relu, const float* __restrict__ scale, SplitView wino,
                                                        SplitView plain, float* __restrict__ ws, SplitEnc enc) {
    extern __shared__ float sh[];
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int Cs = C < 64 ? C : 64;                 // channel slab per block (see prep_kernel)
    const int G = Cs >> 3, SUB = 256 / G;
    const int tid = threadIdx.x, g = blockIdx.z * (Cs >> 3) + tid % G, sub = tid / G;
    const int Wh = (d.W + 1) >> 1;
    const int V = d.D * d.H * d.W, Vh = d.D * d.H * Wh;
    const int per = (Vh + nblk - 1) / nblk;
    const int p0 = blk * per, p1 = min(Vh, p0 + per);
    float m[8], r[8], sc[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t ci = (int64_t)b * C + g * 8 + j;
        m[j] = mean ? mean[ci] : 0.f;
        r[j] = rstd ? rstd[ci] : 1.f;
        sc[j] = scale ? scale[ci] : 1.f;
        acc[j] = 0.f;
    }
    int bad = 0;
    const float* xb = x + (int64_t)b * V * C + g * 8;
    _Float16* wb = wino.p + (((int64_t)b * wino.chunks_total + wino.chunk_off + (g >> 1)) * 4 * Vh) * 32;
    const int kh = g & 1;                           // which 8 of the chunk's 16 channels: plane q = kh (hi), 2 + kh (lo)
    _Float16* pb = plain.p ? plain.p + (((int64_t)b * plain.chunks_total + plain.chunk_off + (g >> 1)) * V) * 32 + (g & 1) * 8 : nullptr;
    for (int ph = p0 + sub; ph < p1; ph += SUB) {
        const int row = ph / Wh, i = ph - row * Wh;
        const int xo = 2 * i;
        float dv[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = xo - 1 + k;
            const bool ok = (unsigned)xx < (unsigned)d.W;
            const int xc = ok ? xx : xo;
            float4 a = *reinterpret_cast<const float4*>(xb + ((int64_t)row * d.W + xc) * C);
            float4 c = *reinterpret_cast<const float4*>(xb + ((int64_t)row * d.W + xc) * C + 4);
            float y[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (y[j] - m[j]) * r[j];
                if (relu) t = fmaxf(t, 0.f);
                dv[k][j] = ok ? t * sc[j] : 0.f;
            }
        }
        float t[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            t[0][j] = dv[0][j] - dv[2][j];
            t[1][j] = dv[1][j] + dv[2][j];
            t[2][j] = dv[2][j] - dv[1][j];
            t[3][j] = dv[1][j] - dv[3][j];
            acc[j] += dv[1][j] + dv[2][j];      // d2 is 0 when x = 2i+1 is outside
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            half8 hi, lo;
            split8(t[pp], hi, lo, bad, enc.ascale);
            *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + kh) * Vh + ph) * 8) = hi;
            *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + 2 + kh) * Vh + ph) * 8) = lo;
        }
        if (pb) {
#pragma unroll
            for (int k = 1; k <= 2; ++k)
                if (xo - 1 + k < d.W) {
                    half8 hi, lo;
                    split8(dv[k], hi, lo, bad, enc.ascale);
                    _Float16* o = pb + ((int64_t)row * d.W + xo - 1 + k) * 32;
                    *reinterpret_cast<half8*>(o) = hi;
                    *reinterpret_cast<half8*>(o + 16) = lo;
                }
        }
    }
    if (bad) atomicOr(enc.err + b, bad);
    if (ws) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[tid * 8 + j] = acc[j];
        __syncthreads();
        for (int off = SUB >> 1; off > 0; off >>= 1) {
            if (sub < off) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sh[tid * 8 + j] += sh[(tid + off * G) * 8 + j];
            }
            __syncthreads();
        }
        if (sub == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ws[((int64_t)b * nblk + blk) * C + g * 8 + j] = sh[tid * 8 + j];
        }
    }
}

void launch_prep_wino(const float* x, int B, Dims d, int C, const float* mean, const float* rstd, int relu, const float* scale,
                      SplitView wino, SplitView plain, float* gap, float* ws, SplitEnc enc, hipStream_t st) {
    int Cs = C < 64 ? C : 64, G = Cs / 8, SUB = 256 / G;
    int Vh = d.D * d.H * ((d.W + 1) / 2);
    int nblk = pick_blocks(Vh, SUB * 2, gap ? RED_BLOCKS : 4096);
    hipLaunchKernelGGL(prep_wino_kernel, dim3(nblk, B, C / Cs), dim3(256), 256 * 8 * sizeof(float), st, x, d, C, mean, rstd, relu, scale,
                       wino, plain, gap ? ws : nullptr, enc);
    if (gap) launch_finalize_sum(ws, B, nblk, C, 1.0f / (float)(d.D * d.H * d.W), gap, st);
}

// NCDHW f32 [B][C][V] -> wino layout (AF3 encodings for feat_conv, head logits as extra channels)
__global__ __launch_bounds__(256) void prep_ncdhw_wino_kernel(const float* __restrict__ x, Dims d, int C, SplitView wino,
                                                              SplitEnc enc) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int Wh = (d.W + 1) >> 1;
    const int V = d.D * d.H * d.W, Vh = d.D * d.H * Wh;
    const int ph = blockIdx.x * 256 + threadIdx.x;
    if (ph >= Vh) return;
    const int row = ph / Wh, i = ph - row * Wh, xo = 2 * i;
    int bad = 0;
    float t[4][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = ch * 16 + j;
        float dv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = xo - 1 + k;
            dv[k] = (c < C && (unsigned)xx < (unsigned)d.W) ? x[((int64_t)b * C + c) * V + (int64_t)row * d.W + xx] : 0.f;
        }
        t[0][j] = dv[0] - dv[2]; t[1][j] = dv[1] + dv[2]; t[2][j] = dv[2] - dv[1]; t[3][j] = dv[1] - dv[3];
    }
    _Float16* wb = wino.p + (((int64_t)b * wino.chunks_total + wino.chunk_off + ch) * 4 * Vh) * 32;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
        float y0[8], y1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { y0[j] = t[pp][j]; y1[j] = t[pp][8 + j]; }
        half8 hi, lo;
        split8(y0, hi, lo, bad, enc.ascale);
        *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + 0) * Vh + ph) * 8) = hi;
        *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + 2) * Vh + ph) * 8) = lo;
        split8(y1, hi, lo, bad, enc.ascale);
        *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + 1) * Vh + ph) * 8) = hi;
        *reinterpret_cast<half8*>(wb + ((int64_t)(pp * 4 + 3) * Vh + ph) * 8) = lo;
    }
    if (bad) atomicOr(enc.err + b, bad);
}
void launch_prep_ncdhw_wino(const float* x, int B, Dims d, int C, SplitView wino, SplitEnc enc, hipStream_t st) {
    int Vh = d.D * d.H * ((d.W + 1) / 2);
    dim3 grid((Vh + 255) / 256, (C + 15) / 16, B);
    hipLaunchKernelGGL(prep_ncdhw_wino_kernel, grid, dim3(256), 0, st, x, d, C, wino, enc);
}

// NCDHW f32 [B][C][V] (the caller's layout, model.py:331) -> split, channels zero-padded to 16.
__global__ __launch_bounds__(256) void prep_ncdhw_kernel(const float* __restrict__ x, int V, int C, SplitView out,
                                                         float* __restrict__ abs_sum, SplitEnc enc) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    float asum = 0.f;
    int bad = 0;
    if (v < V) {
        float y[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int c = ch * 16 + j;
            y[j] = c < C ? x[((int64_t)b * C + c) * V + v] : 0.f;
            asum += fabsf(y[j]);
        }
        if (out.p) {
            _Float16* o = out.p + (((int64_t)b * out.chunks_total + out.chunk_off + ch) * V + v) * 32;
            half8 hi, lo;
            float y0[8], y1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { y0[j] = y[j]; y1[j] = y[8 + j]; }
            split8(y0, hi, lo, bad, enc.ascale);
            *reinterpret_cast<half8*>(o) = hi;
            *reinterpret_cast<half8*>(o + 16) = lo;
            split8(y1, hi, lo, bad, enc.ascale);
            *reinterpret_cast<half8*>(o + 8) = hi;
            *reinterpret_cast<half8*>(o + 24) = lo;
        }
    }
    if (bad) atomicOr(enc.err + b, bad);
    if (abs_sum) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) asum += __shfl_xor(asum, o);
        if ((threadIdx.x & 63) == 0 && asum != 0.f) atomicAdd(abs_sum + b, asum);
    }
}

// sum |x| over each batch entry of a contiguous f32 tensor [B][n] (the AF3 test of model.py:60).  One float atomic per block:
// the per-wave atomics of prep_ncdhw_kernel (65 K of them on 8 addresses) made that pass 6x slower than its traffic.
__global__ __launch_bounds__(256) void abs_sum_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ abs_sum) {
    __shared__ float sh[4];
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * n;
    float s = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(xb)[i];
        s += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += fabsf(xb[n4 * 4 + threadIdx.x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (t != 0.f) atomicAdd(abs_sum + b, t);
    }
}
void launch_abs_sum(const float* x, int B, int64_t n, float* abs_sum, hipStream_t st) {
    hipLaunchKernelGGL(abs_sum_kernel, dim3(128, B), dim3(256), 0, st, x, n, abs_sum);
}

void launch_prep_ncdhw(const float* x, int B, int V, int C, SplitView out, float* abs_sum, SplitEnc enc, hipStream_t st) {
    dim3 grid((V + 255) / 256, (C + 15) / 16, B);
    hipLaunchKernelGGL(prep_ncdhw_kernel, grid, dim3(256), 0, st, x, V, C, out, abs_sum, enc);
}

// layout transposes for the op-level entry points (tests): tiled through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int R, int Cc, float* __restrict__ y) {
    // x [batch][R][Cc] -> y [batch][Cc][R]
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const float* xb = x + (int64_t)b * R * Cc;
    float* yb = y + (int64_t)b * R * Cc;
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        int r = r0 + i, c = c0 + tx;
        if (r < R && c < Cc) t[i][tx] = xb[(int64_t)r * Cc + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, r = r0 + tx;
        if (r < R && c < Cc) yb[(int64_t)c * R + r] = t[tx][i];
    }
}
void launch_nchw_to_nhwc(const float* x, int B, int C, int V, float* y, hipStream_t st) {
    dim3 grid((V + 31) / 32, (C + 31) / 32, B);   // x [B][C][V] : R = C, Cc = V
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, x, C, V, y);
}
void launch_nhwc_to_nchw(const float* x, int B, int C, int V, float* y, hipStream_t st) {
    dim3 grid((C + 31) / 32, (V + 31) / 32, B);   // x [B][V][C] : R = V, Cc = C
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, x, V, C, y);
}

// ------------------------------------------------------------------------------------------------
// Gate MLPs: SEBlock.fc (model.py:245-258), exp_attention (:20-26), global_attn (:87-94),
// calibration (:215-223): gate = sigmoid(W2 relu(W1 p + b1) + b2), p = pooled vector (* premul).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_mlp_kernel(const float* __restrict__ pool, const float* __restrict__ premul,
                                                       int C, int Ch, const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       const float* __restrict__ postmul, float* __restrict__ gate,
                                                       float* __restrict__ gate_post, int post_stride) {
    __shared__ float p[512];
    __shared__ float hid[128];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) p[c] = pool[(int64_t)b * C + c] * (premul ? premul[(int64_t)b * C + c] : 1.f);
    __syncthreads();
    for (int h = tid; h < Ch; h += 256) {
        float s = b1[h];
        for (int c = 0; c < C; ++c) s = fmaf(w1[(int64_t)h * C + c], p[c], s);
        hid[h] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = b2[c];
        for (int h = 0; h < Ch; ++h) s = fmaf(w2[(int64_t)c * Ch + h], hid[h], s);
        float gv = 1.f / (1.f + expf(-s));
        if (gate) gate[(int64_t)b * C + c] = gv;
        if (gate_post) gate_post[(int64_t)b * post_stride + c] = gv * (postmul ? postmul[(int64_t)b * C + c] : 1.f);
    }
}
void launch_gate_mlp(const float* pool, const float* premul, int B, int C, int Ch, const float* w1, const float* b1,
                     const float* w2, const float* b2, const float* postmul, float* gate, float* gate_post, int post_stride,
                     hipStream_t st) {
    hipLaunchKernelGGL(gate_mlp_kernel, dim3(B), dim3(256), 0, st, pool, premul, C, Ch, w1, b1, w2, b2, postmul, gate,
                       gate_post, post_stride);
}

// ------------------------------------------------------------------------------------------------
// feat_gate (model.py:31-36,70-71): per-voxel importance = sigmoid(w2 . relu(W0 x + b0) + b2), x*importance
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void feat_gate_kernel(const float* __restrict__ x, int V, const float* __restrict__ w0,
                                                        const float* __restrict__ b0, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, SplitView out, SplitEnc enc) {
    __shared__ float sw0[16 * 64];
    __shared__ float sb0[16], sw2[16];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 16 * 64; i += 256) sw0[i] = w0[i];
    if (tid < 16) { sb0[tid] = b0[tid]; sw2[tid] = w2[tid]; }
    __syncthreads();
    const int v = blockIdx.x * 256 + tid;
    if (v >= V) return;
    float xv[64];
    const float* xp = x + ((int64_t)b * V + v) * 64;
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
        float4 a = *reinterpret_cast<const float4*>(xp + j);
        xv[j] = a.x; xv[j + 1] = a.y; xv[j + 2] = a.z; xv[j + 3] = a.w;
    }
    float z = b2[0];
#pragma unroll 1
    for (int h = 0; h < 16; ++h) {
        float s = sb0[h];
#pragma unroll
        for (int j = 0; j < 64; ++j) s = fmaf(sw0[h * 64 + j], xv[j], s);
        z = fmaf(sw2[h], fmaxf(s, 0.f), z);
    }
    const float gt = 1.f / (1.f + expf(-z));
    int bad = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        _Float16* o = out.p + (((int64_t)b * out.chunks_total + out.chunk_off + ch) * V + v) * 32;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = xv[ch * 16 + kh * 8 + j] * gt;
            half8 hi, lo;
            split8(y, hi, lo, bad, enc.ascale);
            *reinterpret_cast<half8*>(o + kh * 8) = hi;
            *reinterpret_cast<half8*>(o + 16 + kh * 8) = lo;
        }
    }
    if (bad) atomicOr(enc.err + b, bad);
}
void launch_feat_gate(const float* x, int B, int V, const float* w0, const float* b0, const float* w2, const float* b2,
                      SplitView out, SplitEnc enc, hipStream_t st) {
    hipLaunchKernelGGL(feat_gate_kernel, dim3((V + 255) / 256, B), dim3(256), 0, st, x, V, w0, b0, w2, b2, out, enc);
}

// ------------------------------------------------------------------------------------------------
// Head tail (model.py:232,238-239): relu(IN(conv2 out)) * calibration gate -> final 1x1 (32 -> ncls).
// Writes NCDHW logits (the inner boundary's layout) and, for the backbone / CA heads, the logits as
// extra input channels of the next head's conv1 (model.py:345-346).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_final_kernel(const float* __restrict__ x, int V, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gate,
                                                         const float* __restrict__ wf, const float* __restrict__ bf, int ncls,
                                                         float* __restrict__ logits, int extra_off,
                                                         float* __restrict__ extra_raw, int extra_raw_c) {
    __shared__ float sw[21 * 32];
    __shared__ float sm[32], sr[32], sg[32], sb[21];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < ncls * 32; i += 256) sw[i] = wf[i];
    if (tid < 32) { sm[tid] = mean[(int64_t)b * 32 + tid]; sr[tid] = rstd[(int64_t)b * 32 + tid]; sg[tid] = gate[(int64_t)b * 32 + tid]; }
    if (tid < ncls) sb[tid] = bf[tid];
    __syncthreads();
    const int v = blockIdx.x * 256 + tid;
    if (v >= V) return;
    float t[32];
    const float* xp = x + ((int64_t)b * V + v) * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
        float4 a = *reinterpret_cast<const float4*>(xp + j);
        t[j] = a.x; t[j + 1] = a.y; t[j + 2] = a.z; t[j + 3] = a.w;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) t[j] = fmaxf((t[j] - sm[j]) * sr[j], 0.f) * sg[j];
    for (int n = 0; n < ncls; ++n) {
        float s = sb[n];
#pragma unroll
        for (int j = 0; j < 32; ++j) s = fmaf(sw[n * 32 + j], t[j], s);
        logits[((int64_t)b * ncls + n) * V + v] = s;
        if (extra_raw) extra_raw[((int64_t)b * extra_raw_c + extra_off + n) * V + v] = s;
    }
}
void launch_head_final(const float* x, int B, int V, const float* mean, const float* rstd, const float* gate,
                       const float* wf, const float* bf, int ncls, float* logits, int extra_ch_off, float* extra_raw,
                       int extra_raw_c, hipStream_t st) {
    hipLaunchKernelGGL(head_final_kernel, dim3((V + 255) / 256, B), dim3(256), 0, st, x, V, mean, rstd, gate, wf, bf,
                       ncls, logits, extra_ch_off, extra_raw, extra_raw_c);
}

// ------------------------------------------------------------------------------------------------
// Head post-processing (utils/predict.py:342-349): class 1 dropped, softmax over the remaining three,
// keep the last; amino acids: softmax over classes 1..20 and argmax (first maximum).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softmax3_last(float a, float c, float e) {
    float m = fmaxf(a, fmaxf(c, e));
    float ea = expf(a - m), ec = expf(c - m), ee = expf(e - m);
    return ee / (ea + ec + ee);
}
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ bb, const float* __restrict__ ca,
                                                          const float* __restrict__ aa, int V, float* __restrict__ bbp,
                                                          float* __restrict__ cap, float* __restrict__ aap,
                                                          float* __restrict__ aapred, int64_t s1, int64_t s20) {
    // s1 / s20: elements between batch entries of the single-channel outputs / of the 20-channel output (V and 20 V for
    // four separate tensors; 23 V each when the four are slices of one record tensor [B][23][V])
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* pb = bb + (int64_t)b * 4 * V + v;
    const float* pc = ca + (int64_t)b * 4 * V + v;
    bbp[(int64_t)b * s1 + v] = softmax3_last(pb[0], pb[2 * (int64_t)V], pb[3 * (int64_t)V]);
    cap[(int64_t)b * s1 + v] = softmax3_last(pc[0], pc[2 * (int64_t)V], pc[3 * (int64_t)V]);
    const float* pa = aa + (int64_t)b * 21 * V + v;
    float l[20];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 20; ++j) {
        l[j] = pa[(int64_t)(j + 1) * V];
        m = fmaxf(m, l[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 20; ++j) { l[j] = expf(l[j] - m); s += l[j]; }
    // torch.max(aa_scores, 1)[1] (predict.py:349): the first maximum of the softmax SCORES - two logits that round to the same
    // score resolve to the lower class, as in the reference
    float best = -1.f;
    int am = 0;
#pragma unroll
    for (int j = 0; j < 20; ++j) {
        const float pj = l[j] / s;
        aap[(int64_t)b * s20 + (int64_t)j * V + v] = pj;
        if (pj > best) { best = pj; am = j; }
    }
    aapred[(int64_t)b * s1 + v] = (float)am;
}
void launch_postprocess(const float* bb, const float* ca, const float* aa, int B, int V, float* bbp, float* cap,
                        float* aap, float* aapred, int64_t s1, int64_t s20, hipStream_t st) {
    hipLaunchKernelGGL(postprocess_kernel, dim3((V + 255) / 256, B), dim3(256), 0, st, bb, ca, aa, V, bbp, cap, aap, aapred, s1, s20);
}

__global__ void fill_float_kernel(float* p, int64_t n, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_float(float* p, int64_t n, float v, hipStream_t st) {
    hipLaunchKernelGGL(fill_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
}

// ------------------------------------------------------------------------------------------------
// Tile gather (create_grids.py:129-157): window W = grid + 2*pad at stride grid from the zero-padded
// volume; out-of-volume voxels read 0 (np.pad 'constant').  Pure copy => bit exact.
// Stitch (predict.py:494-501): central grid^3 of every tile back into the volume; regions are disjoint.
// ------------------------------------------------------------------------------------------------
template <typename TIn>
__global__ __launch_bounds__(256) void gather_tiles_kernel(const TIn* __restrict__ vol, int C, int64_t n0, int64_t n1,
                                                           int64_t n2, int grid, int pad, int64_t first, int nt1, int nt2,
                                                           float* __restrict__ tiles) {
    const int W = grid + 2 * pad;
    const int64_t W3 = (int64_t)W * W * W;
    const int t = blockIdx.z, c = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= W3) return;
    const int a2 = e % W, a1 = (e / W) % W, a0 = e / ((int64_t)W * W);
    const int64_t tg = first + t;
    const int64_t tk = tg % nt2, tj = (tg / nt2) % nt1, ti = tg / ((int64_t)nt2 * nt1);
    const int64_t s0 = ti * grid + a0 - pad, s1 = tj * grid + a1 - pad, s2 = tk * grid + a2 - pad;
    float v = 0.f;
    if (s0 >= 0 && s0 < n0 && s1 >= 0 && s1 < n1 && s2 >= 0 && s2 < n2) v = (float)vol[(((int64_t)c * n0 + s0) * n1 + s1) * n2 + s2];
    tiles[((int64_t)t * C + c) * W3 + e] = v;
}
void launch_gather_tiles(const float* vol, int C, int64_t n0, int64_t n1, int64_t n2, int grid, int pad, int64_t first,
                         int64_t count, float* tiles, hipStream_t st) {
    int W = grid + 2 * pad;
    int64_t W3 = (int64_t)W * W * W;
    int nt1 = (int)((n1 + grid - 1) / grid), nt2 = (int)((n2 + grid - 1) / grid);
    dim3 g((unsigned)((W3 + 255) / 256), C, (unsigned)count);
    hipLaunchKernelGGL(gather_tiles_kernel<float>, g, dim3(256), 0, st, vol, C, n0, n1, n2, grid, pad, first, nt1, nt2, tiles);
}
// the same windows from a uint8 volume (binary AF3 encodings held at a quarter of the memory), converted to f32 on the way
void launch_gather_tiles_u8(const uint8_t* vol, int C, int64_t n0, int64_t n1, int64_t n2, int grid, int pad, int64_t first,
                            int64_t count, float* tiles, hipStream_t st) {
    int W = grid + 2 * pad;
    int64_t W3 = (int64_t)W * W * W;
    int nt1 = (int)((n1 + grid - 1) / grid), nt2 = (int)((n2 + grid - 1) / grid);
    dim3 g((unsigned)((W3 + 255) / 256), C, (unsigned)count);
    hipLaunchKernelGGL(gather_tiles_kernel<uint8_t>, g, dim3(256), 0, st, vol, C, n0, n1, n2, grid, pad, first, nt1, nt2, tiles);
}

__global__ __launch_bounds__(256) void stitch_tiles_kernel(const float* __restrict__ tiles, int C, int64_t n0, int64_t n1,
                                                           int64_t n2, int grid, int pad, int64_t first, int nt1, int nt2,
                                                           float* __restrict__ vol) {
    const int W = grid + 2 * pad;
    const int64_t W3 = (int64_t)W * W * W, G3 = (int64_t)grid * grid * grid;
    const int t = blockIdx.z, c = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= G3) return;
    const int u2 = e % grid, u1 = (e / grid) % grid, u0 = e / ((int64_t)grid * grid);
    const int64_t tg = first + t;
    const int64_t tk = tg % nt2, tj = (tg / nt2) % nt1, ti = tg / ((int64_t)nt2 * nt1);
    const int64_t d0 = ti * grid + u0, d1 = tj * grid + u1, d2 = tk * grid + u2;
    if (d0 < n0 && d1 < n1 && d2 < n2)
        vol[(((int64_t)c * n0 + d0) * n1 + d1) * n2 + d2] =
            tiles[((int64_t)t * C + c) * W3 + ((int64_t)(pad + u0) * W + (pad + u1)) * W + (pad + u2)];
}
void launch_stitch_tiles(const float* tiles, int C, int64_t n0, int64_t n1, int64_t n2, int grid, int pad, int64_t first,
                         int64_t count, float* vol, hipStream_t st) {
    int64_t G3 = (int64_t)grid * grid * grid;
    int nt1 = (int)((n1 + grid - 1) / grid), nt2 = (int)((n2 + grid - 1) / grid);
    dim3 g((unsigned)((G3 + 255) / 256), C, (unsigned)count);
    hipLaunchKernelGGL(stitch_tiles_kernel, g, dim3(256), 0, st, tiles, C, n0, n1, n2, grid, pad, first, nt1, nt2, vol);
}

// ---- AF3 encoding rasteriser: DataPreprocessor.create_AF3_encodings' atom loop (reference utils/preprocessing.py:172-178,
// 283-298).  Per atom: idx = clip(round_half_even(float32(coord - origin)), 0, shape - 1) where `shape` is the map's
// (nz, ny, nx) applied to the (x, y, z) components IN THAT ORDER (the reference's quirk: x is clipped against nz, z against
// nx), then volume[ch, idx[2], idx[1], idx[0]] = 1 for the atom's backbone channel and its residue's amino-acid channel.
// Where the reference would raise IndexError (non-cubic maps: clipped z index >= nz or x index >= nx) the flag is set.
// Writes of 1.0 commute, so atom order does not matter.
__device__ __forceinline__ long long af3_index(float c, float o, long long n) {
    const float f = rintf(__fsub_rn(c, o));                       // np.round on float32: half to even
    // numpy's float32 -> int64 cast (cvttss2si) gives INT64_MIN for NaN, +-inf and |f| >= 2^63; np.clip then makes it 0
    if (!(fabsf(f) < 9.2233720368547758e18f)) return 0;
    const long long i = (long long)f;
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
__global__ void rasterise_atoms_kernel(const float* __restrict__ xyz, const int* __restrict__ bb, const int* __restrict__ aa,
                                       int64_t n_atoms, float ox, float oy, float oz, int64_t nz, int64_t ny, int64_t nx,
                                       float* __restrict__ vol, int* __restrict__ flag) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_atoms) return;
    const int cb = bb[a], ca = aa[a];
    if (cb < 0 && ca < 0) return;
    const long long i0 = af3_index(xyz[a * 3 + 0], ox, nz);       // x component, clipped against shape[0] = nz
    const long long i1 = af3_index(xyz[a * 3 + 1], oy, ny);
    const long long i2 = af3_index(xyz[a * 3 + 2], oz, nx);       // z component, clipped against shape[2] = nx
    if (i2 >= nz || i0 >= nx) { atomicOr(flag, 1); return; }      // volume[ch, i2, i1, i0] would raise IndexError
    const int64_t off = (i2 * ny + i1) * nx + i0, V = nz * ny * nx;
    if (cb >= 0) vol[(int64_t)cb * V + off] = 1.0f;
    if (ca >= 0) vol[(int64_t)ca * V + off] = 1.0f;
}
int rasterise_atoms_device(const float* d_xyz, const int* d_bb, const int* d_aa, int64_t n_atoms, const float* origin, int64_t nz,
                           int64_t ny, int64_t nx, float* d_vol, hipStream_t st, char* err, int errlen) {
    int* d_flag = nullptr;
    if (hipMalloc(&d_flag, sizeof(int)) != hipSuccess) { snprintf(err, errlen, "mica_rasterise_atoms: hipMalloc failed"); return -2; }
    int h_flag = 0;
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(int), st);
    if (e == hipSuccess) e = hipMemsetAsync(d_vol, 0, (size_t)24 * nz * ny * nx * sizeof(float), st);
    if (e == hipSuccess && n_atoms > 0) {
        hipLaunchKernelGGL(rasterise_atoms_kernel, dim3((unsigned)((n_atoms + 255) / 256)), dim3(256), 0, st, d_xyz, d_bb, d_aa, n_atoms,
                           origin[0], origin[1], origin[2], nz, ny, nx, d_vol, d_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_flag);
    if (e != hipSuccess) { snprintf(err, errlen, "mica_rasterise_atoms: %s", hipGetErrorString(e)); return -2; }
    if (h_flag) {
        snprintf(err, errlen, "mica_rasterise_atoms: an atom indexes outside the volume (the reference raises IndexError: x is "
                              "clipped against nz and z against nx, preprocessing.py:177,295)");
        return -4;
    }
    return 0;
}

}  // namespace mica

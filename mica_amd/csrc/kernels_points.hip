// Point-list kernels for the consumers of the stitched volumes (reference utils/modeler.py:762-858, Solver.clustering):
// the three steps of it that touch whole volumes, so that only ~1e4-1e5 candidate points ever leave the GPU.
//   threshold_points : np.array(np.where(CAProb > thr)).T            (modeler.py:767)  -> ascending linear indices
//   gather_values    : vol[x, y, z] at a list of points              (modeler.py:780, 786, 800, 856, 884)
//   refine_candidates: 3x3x3 probability-weighted sub-voxel position and amino-acid profile (modeler.py:836-852)
// DBSCAN (open3d, :770), the cluster scores and the greedy NMS (:822-831) work on the point list and stay with the caller.
#include "common.h"
#include <cstdio>

namespace mica {

constexpr int TP_BLOCK = 256, TP_PER = 16, TP_CHUNK = TP_BLOCK * TP_PER;   // elements per block

__global__ __launch_bounds__(TP_BLOCK) void tp_count_kernel(const float* __restrict__ v, int64_t n, float thr,
                                                            unsigned* __restrict__ counts) {
    const int64_t base = (int64_t)blockIdx.x * TP_CHUNK;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < TP_PER; ++k) {
        const int64_t i = base + (int64_t)k * TP_BLOCK + threadIdx.x;
        c += (i < n && v[i] > thr) ? 1u : 0u;
    }
    __shared__ unsigned sh[TP_BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// exclusive scan of up to 1024 * per counts in one block; total -> offsets[nblk]
__global__ __launch_bounds__(1024) void tp_scan_kernel(const unsigned* __restrict__ counts, int nblk, int64_t* __restrict__ offsets) {
    __shared__ int64_t sh[1024];
    const int per = (nblk + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    int64_t s = 0;
    for (int k = 0; k < per; ++k)
        if (b0 + k < nblk) s += counts[b0 + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int64_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = sh[threadIdx.x] - s;
    for (int k = 0; k < per; ++k)
        if (b0 + k < nblk) { offsets[b0 + k] = run; run += counts[b0 + k]; }
    if (threadIdx.x == 1023) offsets[nblk] = sh[1023];
}

// second pass: the k-th sweep of a block covers 256 consecutive elements, so "sweep-major, thread-minor" IS ascending order
__global__ __launch_bounds__(TP_BLOCK) void tp_write_kernel(const float* __restrict__ v, int64_t n, float thr,
                                                            const int64_t* __restrict__ offsets, int64_t capacity,
                                                            int64_t* __restrict__ idx) {
    __shared__ unsigned wsum[TP_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * TP_CHUNK;
    int64_t out = offsets[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = 0; k < TP_PER; ++k) {
        const int64_t i = base + (int64_t)k * TP_BLOCK + threadIdx.x;
        const bool p = i < n && v[i] > thr;
        const unsigned long long m = __ballot(p);
        const unsigned before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        unsigned wbase = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < TP_BLOCK / 64; ++j) { if (j < w) wbase += wsum[j]; tot += wsum[j]; }
        if (p) {
            const int64_t o = out + wbase + before;
            if (o < capacity) idx[o] = i;
        }
        out += tot;
        __syncthreads();
    }
}

int threshold_points_device(const float* d_vol, int64_t n, float thr, int64_t* d_idx, int64_t capacity, int64_t* h_count, hipStream_t st,
                            char* err, int errlen) {
    const int nblk = (int)((n + TP_CHUNK - 1) / TP_CHUNK);
    unsigned* d_counts = nullptr;
    int64_t* d_off = nullptr;
    if (hipMalloc(&d_counts, (size_t)nblk * sizeof(unsigned)) != hipSuccess || hipMalloc(&d_off, (size_t)(nblk + 1) * sizeof(int64_t)) != hipSuccess) {
        if (d_counts) hipFree(d_counts);
        snprintf(err, errlen, "mica_threshold_points: hipMalloc failed");
        return -2;
    }
    hipLaunchKernelGGL(tp_count_kernel, dim3(nblk), dim3(TP_BLOCK), 0, st, d_vol, n, thr, d_counts);
    hipLaunchKernelGGL(tp_scan_kernel, dim3(1), dim3(1024), 0, st, d_counts, nblk, d_off);
    hipLaunchKernelGGL(tp_write_kernel, dim3(nblk), dim3(TP_BLOCK), 0, st, d_vol, n, thr, d_off, capacity, d_idx);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_count, d_off + nblk, sizeof(int64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_counts);
    hipFree(d_off);
    if (e != hipSuccess) { snprintf(err, errlen, "mica_threshold_points: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

__global__ void gather_values_kernel(const float* __restrict__ vol, int C, int64_t nvox, const int64_t* __restrict__ idx, int64_t n,
                                     float* __restrict__ out, int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t p = idx[i];
    if (p < 0 || p >= nvox) { atomicOr(flag, 1); return; }
    for (int c = 0; c < C; ++c) out[(int64_t)c * n + i] = vol[(int64_t)c * nvox + p];
}

int gather_values_device(const float* d_vol, int C, int64_t nvox, const int64_t* d_idx, int64_t n, float* d_out, hipStream_t st, char* err,
                         int errlen) {
    int* d_flag = nullptr;
    int h_flag = 0;
    if (hipMalloc(&d_flag, sizeof(int)) != hipSuccess) { snprintf(err, errlen, "mica_gather_values: hipMalloc failed"); return -2; }
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(int), st);
    if (e == hipSuccess && n > 0) {
        hipLaunchKernelGGL(gather_values_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_vol, C, nvox, d_idx, n, d_out, d_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_flag);
    if (e != hipSuccess) { snprintf(err, errlen, "mica_gather_values: %s", hipGetErrorString(e)); return -2; }
    if (h_flag) { snprintf(err, errlen, "mica_gather_values: index outside the volume"); return -1; }
    return 0;
}

// modeler.py:836-852 per candidate, with numpy's arithmetic:
//   weights = CAProb[3x3x3] / np.sum(CAProb[3x3x3])    float32; np.sum of the 27-element view = numpy's pairwise routine on the
//             C-order copy: eight accumulators over elements 0..23 (three rounds), combined as ((0+1)+(2+3))+((4+5)+(6+7)),
//             then elements 24, 25, 26 added in order
//   coord  += this_coord * weights[...]                 int64 x float32 -> float64 products (exact), added in loop order
//   AA      = np.sum([AAProb[:, p] * w for p], axis=0)  float32 products, rows added in loop order
// A candidate on the volume's faces makes the reference's slice short and the loop raise: it is skipped (ok = 0).
__global__ void refine_candidates_kernel(const float* __restrict__ ca, const float* __restrict__ aa, int n0, int n1, int n2,
                                         const int* __restrict__ cand, int64_t n, double* __restrict__ coord,
                                         float* __restrict__ aaout, int* __restrict__ ok) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c0 = cand[i * 3], c1 = cand[i * 3 + 1], c2 = cand[i * 3 + 2];
    const bool inside = c0 >= 1 && c0 <= n0 - 2 && c1 >= 1 && c1 <= n1 - 2 && c2 >= 1 && c2 <= n2 - 2;
    ok[i] = inside ? 1 : 0;
    if (!inside) {
        coord[i * 3] = coord[i * 3 + 1] = coord[i * 3 + 2] = 0.0;
        for (int c = 0; c < 20; ++c) aaout[i * 20 + c] = 0.f;
        return;
    }
    const int64_t nvox = (int64_t)n0 * n1 * n2;
    float v[27];
#pragma unroll
    for (int e = 0; e < 27; ++e) {
        const int di = e / 9 - 1, dj = (e / 3) % 3 - 1, dk = e % 3 - 1;
        v[e] = ca[((int64_t)(c0 + di) * n1 + (c1 + dj)) * n2 + (c2 + dk)];
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
    for (int b = 8; b < 24; b += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], v[b + j]);
    float s = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    s = __fadd_rn(s, v[24]);
    s = __fadd_rn(s, v[25]);
    s = __fadd_rn(s, v[26]);
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    float acc[20];
#pragma unroll
    for (int e = 0; e < 27; ++e) {
        const int di = e / 9 - 1, dj = (e / 3) % 3 - 1, dk = e % 3 - 1;
        const float w = __fdiv_rn(v[e], s);
        const double wd = (double)w;
        x0 = __dadd_rn(x0, __dmul_rn((double)(c0 + di), wd));
        x1 = __dadd_rn(x1, __dmul_rn((double)(c1 + dj), wd));
        x2 = __dadd_rn(x2, __dmul_rn((double)(c2 + dk), wd));
        const int64_t p = ((int64_t)(c0 + di) * n1 + (c1 + dj)) * n2 + (c2 + dk);
#pragma unroll
        for (int c = 0; c < 20; ++c) {
            const float t = __fmul_rn(aa[(int64_t)c * nvox + p], w);
            acc[c] = e == 0 ? t : __fadd_rn(acc[c], t);
        }
    }
    coord[i * 3] = x0; coord[i * 3 + 1] = x1; coord[i * 3 + 2] = x2;
#pragma unroll
    for (int c = 0; c < 20; ++c) aaout[i * 20 + c] = acc[c];
}

int refine_candidates_device(const float* d_ca, const float* d_aa, int n0, int n1, int n2, const int* d_cand, int64_t n, double* d_coord,
                             float* d_aa_out, int* d_ok, hipStream_t st, char* err, int errlen) {
    if (n > 0) hipLaunchKernelGGL(refine_candidates_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, d_ca, d_aa, n0, n1, n2, d_cand, n,
                                  d_coord, d_aa_out, d_ok);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(err, errlen, "mica_refine_candidates: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace mica

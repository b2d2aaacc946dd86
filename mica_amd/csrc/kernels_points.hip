// Point-list kernels for the consumers of the stitched volumes (reference utils/modeler.py:762-858, Solver.clustering):
// the three steps of it that touch whole volumes, so that only ~1e4-1e5 candidate points ever leave the GPU.
//   threshold_points : np.array(np.where(CAProb > thr)).T            (modeler.py:767)  -> ascending linear indices
//   gather_values    : vol[x, y, z] at a list of points              (modeler.py:780, 786, 800, 856, 884)
//   refine_candidates: 3x3x3 probability-weighted sub-voxel position and amino-acid profile (modeler.py:836-852)
//   segment_sums     : np.sum(BBProb[cluster points]) per DBSCAN cluster, in numpy's summation order   (modeler.py:776-787)
//   nms_points       : the greedy radius non-maximum suppression over the score-sorted candidates       (modeler.py:822-831)
//   neighbour_matrix : candidate distance matrix and the neighbour / backbone-density score matrix      (modeler.py:860-888)
// DBSCAN itself (open3d, :770) and the sort of the candidate list stay with the caller (host, point lists only).
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

// ------------------------------------------------------------------------------------------------------------------------
// np.sum of a contiguous float32 array, per segment, bit for bit (checked against numpy 2.2 on 60 random lengths up to 4e5):
// the reduction runs over buffer-sized pieces of 8192 elements, s = (...((0 + P(piece 0)) + P(piece 1)) + ...), and P is
// numpy's pairwise routine: n <= 128 -> eight running sums over blocks of eight, combined ((0+1)+(2+3))+((4+5)+(6+7)), then
// the n % 8 tail in order (n < 8: plain loop from 0); n > 128 -> P(first n2) + P(rest) with n2 = n/2 rounded down to a
// multiple of 8.  One workgroup per segment: the <= 128 leaves of a piece are summed by one thread each, thread 0 replays
// the recursion over the leaf sums.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int SS_PIECE = 8192, SS_LEAF = 128, SS_MAXLEAF = 256;

__device__ __forceinline__ float ss_leaf_sum(const float* __restrict__ a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r = __fadd_rn(r, a[i]);
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a[i + j]);
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    for (; i < n; ++i) res = __fadd_rn(res, a[i]);
    return res;
}

__global__ __launch_bounds__(256) void segment_sums_kernel(const float* __restrict__ vals, const int64_t* __restrict__ seg_off,
                                                           float* __restrict__ sums) {
    __shared__ int loff[SS_MAXLEAF], llen[SS_MAXLEAF];
    __shared__ float lsum[SS_MAXLEAF];
    __shared__ int nleaf;
    const int64_t o = seg_off[blockIdx.x], n = seg_off[blockIdx.x + 1] - o;
    float total = 0.f;
    for (int64_t p0 = 0; p0 < n; p0 += SS_PIECE) {
        const int m = (int)((n - p0 < SS_PIECE) ? n - p0 : SS_PIECE);
        if (threadIdx.x == 0) {                       // the leaves of the recursion, left to right
            int so[16], sn[16], sp = 0, L = 0;
            so[0] = 0; sn[0] = m; sp = 1;
            while (sp > 0) {
                const int off = so[sp - 1], len = sn[sp - 1];
                --sp;
                if (len <= SS_LEAF) { loff[L] = off; llen[L] = len; ++L; }
                else {
                    int n2 = len / 2;
                    n2 -= n2 % 8;
                    so[sp] = off + n2; sn[sp] = len - n2; ++sp;      // right is pushed first: left is popped first
                    so[sp] = off; sn[sp] = n2; ++sp;
                }
            }
            nleaf = L;
        }
        __syncthreads();
        for (int l = threadIdx.x; l < nleaf; l += 256) lsum[l] = ss_leaf_sum(vals + o + p0 + loff[l], llen[l]);
        __syncthreads();
        if (threadIdx.x == 0) {                       // replay: value(node) = value(left) + value(right)
            int fn[16], fs[16], sp = 0, li = 0;
            float flv[16], ret = 0.f;
            fn[0] = m; fs[0] = 0; sp = 1;
            while (sp > 0) {
                const int t = sp - 1;
                if (fs[t] == 0) {
                    if (fn[t] <= SS_LEAF) { ret = lsum[li++]; --sp; }
                    else { int n2 = fn[t] / 2; n2 -= n2 % 8; fs[t] = 1; fn[sp] = n2; fs[sp] = 0; ++sp; }
                } else if (fs[t] == 1) {
                    int n2 = fn[t] / 2; n2 -= n2 % 8;
                    flv[t] = ret; fs[t] = 2; fn[sp] = fn[t] - n2; fs[sp] = 0; ++sp;
                } else { ret = __fadd_rn(flv[t], ret); --sp; }
            }
            total = __fadd_rn(total, ret);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

int segment_sums_device(const float* d_vals, const int64_t* d_seg_off, int64_t nseg, float* d_sums, hipStream_t st, char* err, int errlen) {
    if (nseg > 0) hipLaunchKernelGGL(segment_sums_kernel, dim3((unsigned)nseg), dim3(256), 0, st, d_vals, d_seg_off, d_sums);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(err, errlen, "mica_segment_sums: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Greedy non-maximum suppression (modeler.py:822-831) over candidates ALREADY sorted by descending score: the reference
// takes the first remaining candidate and deletes every candidate within squared distance nms_radius of it.  Equivalent
// fixed point: candidate i is kept iff no KEPT earlier candidate lies within the radius.  Candidates are distinct voxels,
// so a dense rank volume (rank of the candidate at a voxel, -1 elsewhere) turns the neighbour search into at most
// (2r+1)^3 reads; rounds of "decide every candidate whose earlier neighbours are all decided" reach the fixed point.
// Status updates are monotonic (undecided -> kept | suppressed), so reading a stale 'undecided' only postpones a decision.
// ------------------------------------------------------------------------------------------------------------------------
__global__ void nms_rank_kernel(const int* __restrict__ pts, int64_t n, int n0, int n1, int n2, int* __restrict__ rank, int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = pts[i * 3], b = pts[i * 3 + 1], c = pts[i * 3 + 2];
    if ((unsigned)a >= (unsigned)n0 || (unsigned)b >= (unsigned)n1 || (unsigned)c >= (unsigned)n2) { atomicOr(flag, 1); return; }
    const int old = atomicExch(&rank[((int64_t)a * n1 + b) * n2 + c], (int)i);
    if (old != -1) atomicOr(flag, 2);                  // two candidates on one voxel
}

__global__ void nms_round_kernel(const int* __restrict__ pts, int64_t n, int n0, int n1, int n2, const int* __restrict__ rank, double radius,
                                 int r, int* __restrict__ status, int* __restrict__ undecided) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (__atomic_load_n(&status[i], __ATOMIC_RELAXED) != 0) return;
    const int a = pts[i * 3], b = pts[i * 3 + 1], c = pts[i * 3 + 2];
    bool suppressed = false, blocked = false;
    for (int da = -r; da <= r && !suppressed; ++da) {
        const int x = a + da;
        if ((unsigned)x >= (unsigned)n0) continue;
        for (int db = -r; db <= r && !suppressed; ++db) {
            const int y = b + db;
            if ((unsigned)y >= (unsigned)n1) continue;
            for (int dc = -r; dc <= r; ++dc) {
                const int z = c + dc;
                if ((unsigned)z >= (unsigned)n2) continue;
                if (!((double)(da * da + db * db + dc * dc) <= radius)) continue;       // the reference compares float64 values
                const int j = rank[((int64_t)x * n1 + y) * n2 + z];
                if (j < 0 || j >= i) continue;
                const int sj = __atomic_load_n(&status[j], __ATOMIC_RELAXED);
                if (sj == 1) { suppressed = true; break; }
                if (sj == 0) blocked = true;
            }
        }
    }
    if (suppressed) __atomic_store_n(&status[i], 2, __ATOMIC_RELAXED);
    else if (!blocked) __atomic_store_n(&status[i], 1, __ATOMIC_RELAXED);
    else atomicAdd(undecided, 1);
}

__global__ void nms_keep_kernel(const int* __restrict__ status, int64_t n, int* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = status[i] == 1 ? 1 : 0;
}

int nms_points_device(const int* d_pts, int64_t n, int n0, int n1, int n2, double radius, int* d_keep, hipStream_t st, char* err, int errlen) {
    if (n == 0) return 0;
    const int64_t nvox = (int64_t)n0 * n1 * n2;
    int *d_rank = nullptr, *d_status = nullptr, *d_cnt = nullptr;
    int h[2] = {0, 0};
    auto done = [&](int rc, const char* msg) {
        if (d_rank) hipFree(d_rank);
        if (d_status) hipFree(d_status);
        if (d_cnt) hipFree(d_cnt);
        if (rc) snprintf(err, errlen, "mica_nms_points: %s", msg);
        return rc;
    };
    if (hipMalloc(&d_rank, (size_t)nvox * sizeof(int)) != hipSuccess || hipMalloc(&d_status, (size_t)n * sizeof(int)) != hipSuccess ||
        hipMalloc(&d_cnt, 2 * sizeof(int)) != hipSuccess)
        return done(-2, "hipMalloc failed");
    hipMemsetAsync(d_rank, 0xFF, (size_t)nvox * sizeof(int), st);
    hipMemsetAsync(d_status, 0, (size_t)n * sizeof(int), st);
    hipMemsetAsync(d_cnt, 0, 2 * sizeof(int), st);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(nms_rank_kernel, dim3(nb), dim3(256), 0, st, d_pts, n, n0, n1, n2, d_rank, d_cnt + 1);
    if (hipMemcpyAsync(h, d_cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return done(-2, "rank pass failed");
    if (h[1] & 1) return done(-1, "candidate outside the volume");
    if (h[1] & 2) return done(-1, "two candidates on one voxel");
    int r = 0;
    while ((double)((r + 1) * (r + 1)) <= radius) ++r;          // offsets beyond r cannot satisfy d^2 <= radius
    for (int round = 0;; ++round) {
        hipMemsetAsync(d_cnt, 0, sizeof(int), st);
        hipLaunchKernelGGL(nms_round_kernel, dim3(nb), dim3(256), 0, st, d_pts, n, n0, n1, n2, d_rank, radius, r, d_status, d_cnt);
        if (hipMemcpyAsync(h, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return done(-2, "round failed");
        if (h[0] == 0) break;
        if (round > 1000000) return done(-2, "did not converge");      // every round decides at least the first undecided candidate
    }
    hipLaunchKernelGGL(nms_keep_kernel, dim3(nb), dim3(256), 0, st, d_status, n, d_keep);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(-2, hipGetErrorString(e));
    return done(0, "");
}

// ------------------------------------------------------------------------------------------------------------------------
// modeler.py:860-888 for every ordered pair of candidates, with numpy 2.x arithmetic (this file is built -ffp-contract=off):
//   cand_self_dis[i][j] = sqrt(((0 + dx^2) + dy^2) + dz^2)                                  float64 (calc_dis, :174-181)
//   for 2 <= dis <= 6:  BB_dens = sum_{j=1..4} BBProb[round(j/5 c_n + (5-j)/5 c_c)]          float32 adds (0 + f32 is f32)
//     dis = max(0, |d - 3.8| - 0.5); dis_score = max(0, 1 - dis/2);  neigh_mat = (dis_score + BB_dens/4) / 2
//   Python's max(0, x) returns the int 0 unless x > 0, so the type of dis_score decides where the last sum is formed
//   (NEP 50): a float64 distance term -> float64 sum; a Python number (dis = 0 -> dis_score = 1.0, or dis_score = 0) ->
//   float32 sum.  Both are reproduced.
// legacy = 1: numpy 1.x value-based promotion, the reference's own environment (environment.yml pins numpy 1.19.1): `0 + np.float32`
//   is a float64 there, so BB_dens, BB_dens/4 and the final sum are all float64 (float64 adds of the float32 samples).
// ------------------------------------------------------------------------------------------------------------------------
__global__ void neighbour_matrix_kernel(const double* __restrict__ cands, int64_t n, const float* __restrict__ bb, int n0, int n1, int n2,
                                        double* __restrict__ dis, double* __restrict__ mat, int* __restrict__ flag, int legacy) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * n) return;
    const int64_t i = e / n, j = e - i * n;             // i = cand, j = neigh
    const double ci[3] = {cands[i * 3], cands[i * 3 + 1], cands[i * 3 + 2]}, cj[3] = {cands[j * 3], cands[j * 3 + 1], cands[j * 3 + 2]};
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double d = __dsub_rn(ci[k], cj[k]); s = __dadd_rn(s, __dmul_rn(d, d)); }
    const double d = __dsqrt_rn(s);
    dis[e] = d;
    double out = 0.0;
    if (d <= 6.0 && d >= 2.0) {
        float dens = 0.f;                                // BB_dens = 0; 0 + float32 is a float32 add
        double dens64 = 0.0;                             // ... and a float64 add under numpy 1.x
        for (int q = 1; q <= 4; ++q) {
            const double wa = (double)q / 5.0, wb = (double)(5 - q) / 5.0;
            int64_t c[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = (int64_t)rint(__dadd_rn(__dmul_rn(wa, cj[k]), __dmul_rn(wb, ci[k])));
            // numpy wraps negative indices and raises beyond the shape; positions between two in-volume candidates are in-volume
            if (c[0] < 0 || c[0] >= n0 || c[1] < 0 || c[1] >= n1 || c[2] < 0 || c[2] >= n2) { atomicOr(flag, 1); c[0] = c[1] = c[2] = 0; }
            const float v = bb[(c[0] * n1 + c[1]) * n2 + c[2]];
            dens = __fadd_rn(dens, v);
            dens64 = __dadd_rn(dens64, (double)v);
        }
        const float q4 = __fdiv_rn(dens, 4.0f);
        const double t = __dsub_rn(fabs(__dsub_rn(d, 3.8)), 0.5);
        if (legacy) {
            const double ds = (t > 0.0) ? __dsub_rn(1.0, __ddiv_rn(t, 2.0)) : 1.0;      // max(0, .) picks the int 0 unless x > 0
            out = __ddiv_rn(__dadd_rn(ds > 0.0 ? ds : 0.0, __ddiv_rn(dens64, 4.0)), 2.0);
        } else if (!(t > 0.0)) {
            out = (double)__fdiv_rn(__fadd_rn(1.0f, q4), 2.0f);           // dis = 0 (int): dis_score = 1.0 (Python float) -> float32 sum
        } else {
            const double ds = __dsub_rn(1.0, __ddiv_rn(t, 2.0));
            if (ds > 0.0) out = __ddiv_rn(__dadd_rn(ds, (double)q4), 2.0);
            else out = (double)__fdiv_rn(q4, 2.0f);                        // dis_score = 0 (int): 0 + float32 stays float32
        }
    }
    mat[e] = out;
}

int neighbour_matrix_device(const double* d_cands, int64_t n, const float* d_bb, int n0, int n1, int n2, double* d_dis, double* d_mat,
                            int legacy, hipStream_t st, char* err, int errlen) {
    if (n == 0) return 0;
    int* d_flag = nullptr;
    int h_flag = 0;
    if (hipMalloc(&d_flag, sizeof(int)) != hipSuccess) { snprintf(err, errlen, "mica_neighbour_matrix: hipMalloc failed"); return -2; }
    hipMemsetAsync(d_flag, 0, sizeof(int), st);
    hipLaunchKernelGGL(neighbour_matrix_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, st, d_cands, n, d_bb, n0, n1, n2, d_dis, d_mat,
                       d_flag, legacy);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_flag);
    if (e != hipSuccess) { snprintf(err, errlen, "mica_neighbour_matrix: %s", hipGetErrorString(e)); return -2; }
    if (h_flag) { snprintf(err, errlen, "mica_neighbour_matrix: a sampling position between two candidates leaves the volume"); return -1; }
    return 0;
}

}  // namespace mica

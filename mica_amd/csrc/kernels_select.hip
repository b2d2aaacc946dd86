// Map normalisation (reference utils/preprocessing.py:122-133) on the device: nan_to_num, exact median
// and exact 99.9th percentile of the positives by most-significant-digit radix select (8 bits per pass,
// a 256-bin histogram per pass, no sort), then clip and divide.  HBM-bound: each pass streams the map once.
// The scalar arithmetic reproduces numpy 2.x for float32 input: np.median = float32 mean of the middle
// element(s); np.percentile(method='linear') runs entirely in float32 (q/100, the virtual index (n-1)*q,
// gamma and the lerp), which this file restates so that the output is bit-identical.
#include "common.h"

#include <cfloat>
#include <cmath>
#include <cstdio>

namespace mica {

__device__ __forceinline__ unsigned f2key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float key2f(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    union { unsigned u; float f; } cv;
    cv.u = u;
    return cv.f;
}

__global__ void nan_to_num_kernel(float* __restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        if (v != v) x[i] = 0.f;
        else if (v == INFINITY) x[i] = FLT_MAX;
        else if (v == -INFINITY) x[i] = -FLT_MAX;
    }
}

// mode 0: key(x) ; mode 1: key(m) for m = (x > med) ? x - med : 0, only m > 0
__global__ __launch_bounds__(256) void hist_kernel(const float* __restrict__ x, int64_t n, int mode, float med,
                                                   unsigned prefix, unsigned prefix_mask, int shift,
                                                   unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        if (mode == 1) {
            v = (v > med) ? (v - med) : 0.f;
            if (!(v > 0.f)) continue;
        }
        unsigned k = f2key(v);
        if ((k & prefix_mask) == prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void finish_kernel(float* __restrict__ x, int64_t n, float med, float pct) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        float m = (v > med) ? (v - med) : 0.f;      // (x > median) * (x - median)          :124
        m = (m < pct) ? m : pct;                     // clip at the percentile                :131-132
        x[i] = m / pct;                              // float32 division                      :133
    }
}

// numpy 1.x promotion rules on a float32 map (MICA_NUMPY_LEGACY; the reference pins numpy 1.19.1, environment.yml:8): np.percentile
// returns a float64 there, `(map_data_ >= p) * p` (bool array times a float64 scalar) is a float64 array, so the sum, the division
// (:131-133) and everything up to the final astype(float32) (:139) run in float64, while the two comparisons against p are
// float32 loops with p rounded to float32 (same-kind scalar: the array's type wins).
__global__ void finish_legacy_kernel(float* __restrict__ x, int64_t n, float med, float pcmp, double pct) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float m = (v > med) ? (v - med) : 0.f;
        const double c = (m < pcmp) ? (double)m : pct;
        x[i] = (float)(c / pct);
    }
}

// Integer maps (MRC modes 0 / 1 / 6): numpy promotes them to float64 at `norm_data - median` (:124) and stays there until the
// final astype(float32) (:139).  The map holds the integers as f32; differences against the median (an integer or a half) are
// exact in either width, so only the clip / divide / final rounding need the wide type.
__global__ void finish_f64_kernel(float* __restrict__ x, int64_t n, double med, double pct) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)x[i];
        double m = (v > med) ? (v - med) : 0.0;
        m = (m < pct) ? m : pct;
        x[i] = (float)(m / pct);
    }
}

namespace {
struct Sel {
    const float* x; int64_t n; int mode; float med; unsigned* d_hist; hipStream_t st;
    // value of rank r (0-based) among the selected elements; total (out) = number of selected elements
    int run(int64_t r, float* out, int64_t* total) {
        unsigned prefix = 0, mask = 0;
        unsigned h[256];
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned), st) != hipSuccess) return -1;
            hipLaunchKernelGGL(hist_kernel, dim3(2048), dim3(256), 0, st, x, n, mode, med, prefix, mask, shift, d_hist);
            if (hipMemcpyAsync(h, d_hist, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
            if (hipStreamSynchronize(st) != hipSuccess) return -1;
            int64_t tot = 0;
            for (int i = 0; i < 256; ++i) tot += h[i];
            if (shift == 24) {
                if (total) *total = tot;
                if (r < 0 || r >= tot) return 1;
            }
            int64_t acc = 0;
            int b = 0;
            for (; b < 256; ++b) {
                if (r < acc + (int64_t)h[b]) break;
                acc += h[b];
            }
            r -= acc;
            prefix |= (unsigned)b << shift;
            mask |= 255u << shift;
        }
        *out = key2f(prefix);
        return 0;
    }
};
}  // namespace

// numpy 1.19's 'linear' percentile (function_base.py, _quantile_ureduce_func): float64 virtual index and weights for every input
// type, and the weighted sum x_below * w_below + x_above * w_above instead of numpy 2's _lerp
static double percentile_weights_legacy(int64_t P, int64_t* below, int64_t* above, double* w_above) {
    const double q = 99.9 / 100.0;
    const double ind = q * (double)(P - 1);
    *below = (int64_t)floor(ind);
    *above = *below + 1;
    if (*above > P - 1) *above = P - 1;
    *w_above = ind - (double)*below;
    return 1.0 - *w_above;
}

int normalise_map_device(float* d_vol, int64_t n, int kind, int numpy_rules, double* h_stats, hipStream_t st, char* err, int errlen) {
    const bool wide = kind != MICA_MAP_F32;
    const bool legacy = numpy_rules == MICA_NUMPY_LEGACY;
    unsigned* d_hist = nullptr;
    if (hipMalloc((void**)&d_hist, 256 * sizeof(unsigned)) != hipSuccess) { snprintf(err, errlen, "normalise: hipMalloc failed"); return -2; }
    int rc = 0;
    do {
        hipLaunchKernelGGL(nan_to_num_kernel, dim3(2048), dim3(256), 0, st, d_vol, n);               // :122
        Sel s{d_vol, n, 0, 0.f, d_hist, st};
        float a = 0, b = 0, med;
        int64_t tot = 0;
        if (n & 1) {                                                                                 // :123 np.median
            if (s.run(n / 2, &a, &tot)) { rc = -2; break; }
            med = a;
        } else {
            if (s.run(n / 2 - 1, &a, &tot) || s.run(n / 2, &b, &tot)) { rc = -2; break; }
            med = (float)(a + b) / 2.0f;   // np.mean of two float32: float32 sum, then / 2
        }
        Sel p{d_vol, n, 1, med, d_hist, st};
        int64_t P = 0;
        float lo = 0, hi = 0;
        int r0 = p.run(0, &lo, &P);
        if (r0 < 0) { rc = -2; break; }
        if (P == 0) { snprintf(err, errlen, "No positive values found after thresholding"); rc = -3; break; }   // :163-165
        if (wide) {
            // np.median of an integer array is the float64 mean of the middle element(s): exact, and exact as the f32 above.
            // np.percentile(pos, 99.9) on the float64 positives: the same 'linear' recipe in float64.
            const double medd = (n & 1) ? (double)a : ((double)a + (double)b) / 2.0;
            const double qd = 99.9 / 100.0;
            const double vid = (double)(P - 1) * qd;
            int64_t prevd = (int64_t)floor(vid), nextd = prevd + 1;
            if (vid >= (double)(P - 1)) { prevd = P - 1; nextd = P - 1; }
            const double gd = vid - floor(vid);
            if (p.run(prevd, &lo, nullptr) || p.run(nextd, &hi, nullptr)) { rc = -2; break; }
            const double dd = (double)hi - (double)lo;
            double pctd = (double)lo + dd * gd;
            if (gd >= 0.5) pctd = (double)hi - dd * (1.0 - gd);
            if (legacy) {
                int64_t ib, ia;
                double wa;
                const double wb = percentile_weights_legacy(P, &ib, &ia, &wa);
                if (p.run(ib, &lo, nullptr) || p.run(ia, &hi, nullptr)) { rc = -2; break; }
                const double x1 = (double)lo * wb;
                const double x2 = (double)hi * wa;
                pctd = x1 + x2;
            }
            if (pctd == 0.0) { snprintf(err, errlen, "Percentile value is zero - cannot normalize"); rc = -3; break; }
            hipLaunchKernelGGL(finish_f64_kernel, dim3(2048), dim3(256), 0, st, d_vol, n, medd, pctd);
            if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { rc = -2; break; }
            h_stats[0] = medd;
            h_stats[1] = pctd;
            break;
        }
        if (legacy) {
            int64_t ib, ia;
            double wa;
            const double wb = percentile_weights_legacy(P, &ib, &ia, &wa);
            if (p.run(ib, &lo, nullptr) || p.run(ia, &hi, nullptr)) { rc = -2; break; }
            const double x1 = (double)lo * wb;
            const double x2 = (double)hi * wa;
            const double pctd = x1 + x2;
            if (pctd == 0.0) { snprintf(err, errlen, "Percentile value is zero - cannot normalize"); rc = -3; break; }
            hipLaunchKernelGGL(finish_legacy_kernel, dim3(2048), dim3(256), 0, st, d_vol, n, med, (float)pctd, pctd);
            if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { rc = -2; break; }
            h_stats[0] = med;
            h_stats[1] = pctd;
            break;
        }
        // np.percentile(pos, 99.9) for float32 input, numpy 2.x: everything in float32
        const float q = 99.9f / 100.0f;
        const float vi = (float)(P - 1) * q;
        int64_t prev = (int64_t)floorf(vi), next = prev + 1;
        if (vi >= (float)(P - 1)) { prev = P - 1; next = P - 1; }
        const float gamma = vi - floorf(vi);
        if (p.run(prev, &lo, nullptr) || p.run(next, &hi, nullptr)) { rc = -2; break; }
        const float diff = hi - lo;
        float pct = lo + diff * gamma;
        if (gamma >= 0.5f) pct = hi - diff * (1.0f - gamma);
        if (pct == 0.f) { snprintf(err, errlen, "Percentile value is zero - cannot normalize"); rc = -3; break; }   // :159-161
        hipLaunchKernelGGL(finish_kernel, dim3(2048), dim3(256), 0, st, d_vol, n, med, pct);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { rc = -2; break; }
        h_stats[0] = med;
        h_stats[1] = pct;
    } while (0);
    if (rc == -2 && !err[0]) snprintf(err, errlen, "normalise: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d_hist);
    return rc;
}

}  // namespace mica

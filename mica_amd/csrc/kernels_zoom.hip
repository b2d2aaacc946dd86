// Cubic-spline resampling of the density map to 1 Angstrom: scipy.ndimage.zoom(data, factors, order=3), the call the
// reference makes at utils/preprocessing.py:117 (defaults mode='constant', cval=0, prefilter=True, grid_mode=False).
// scipy is a third-party dependency of the reference (pinned 1.5.2 in environment.yml:13; 1.15.3 in this image) and its
// C source is not in the tree, so the algorithm is restated from its published form (Unser's recursive B-spline
// prefilter with mirror boundaries + separable 4-tap interpolation) and pinned BIT-EXACT against scipy 1.15.3
// (oracle/volume_oracle.py::zoom_cubic, tests).  Everything is f64 like scipy's internals, one rounding to f32 at the
// end.  This file is compiled with -ffp-contract=off: a fused multiply-add would change the last bit.
// Details that matter for bit-exactness (found by black-box comparison):
//   * the pole is the correctly rounded sqrt(3)-2 = -0x1.126145e9ecd56p-2, not the double expression sqrt(3.0)-2.0;
//   * causal initialisation is the exact mirror sum c0 = (c0 + z^(n-1) c[n-1] + sum_i z^i (c[i] + z^(n-1) c[n-1-i])) /
//     (1 - z^(2n-2)), accumulated in index order; z^(n-1) comes from the host's libm pow();
//   * output coordinate cc = o * ((n_in-1)/(n_out-1)); if cc > n_in-1 by rounding the sample is cval = 0 (scipy quirk);
//   * taps floor(cc)-1 .. floor(cc)+2 are mirrored into range; the 64 products are accumulated in (k0,k1,k2) order, each
//     coefficient multiplied by w0, then w1, then w2.
#include "common.h"

#include <cmath>
#include <cstdio>

namespace mica {

constexpr double SPLINE_POLE = -0x1.126145e9ecd56p-2;

__global__ void f32_to_f64_kernel(const float* __restrict__ x, double* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = (double)x[i];
}

// one thread per line; element i of line L at c[base(L) + i*stride]
__global__ __launch_bounds__(256) void spline_prefilter_kernel(double* __restrict__ c, int n, int64_t stride, int64_t nlines,
                                                               int64_t inner, int64_t outer_stride, double z_n_1) {
    const int64_t L = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= nlines || n < 2) return;
    double* p = c + (L / inner) * outer_stride + (L % inner);
    const double z = SPLINE_POLE;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    for (int i = 0; i < n; ++i) p[i * stride] *= gain;
    double c0 = p[0] + z_n_1 * p[(int64_t)(n - 1) * stride];
    double zi = z;
    for (int i = 1; i < n - 1; ++i) {
        c0 += zi * (p[i * stride] + z_n_1 * p[(int64_t)(n - 1 - i) * stride]);
        zi *= z;
    }
    double prev = c0 / (1.0 - z_n_1 * z_n_1);
    p[0] = prev;
    for (int i = 1; i < n; ++i) {
        prev = p[i * stride] + z * prev;
        p[i * stride] = prev;
    }
    double nxt = (z * p[(int64_t)(n - 2) * stride] + p[(int64_t)(n - 1) * stride]) * z / (z * z - 1.0);
    p[(int64_t)(n - 1) * stride] = nxt;
    for (int i = n - 2; i >= 0; --i) {
        nxt = z * (nxt - p[i * stride]);
        p[i * stride] = nxt;
    }
}

struct ZoomAxis { int n_in, n_out; double zf; };

__device__ __forceinline__ int mirror_idx(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

__device__ __forceinline__ bool axis_setup(const ZoomAxis& a, int o, int (&idx)[4], double (&w)[4]) {
    const double cc = (double)o * a.zf;
    if (cc > (double)(a.n_in - 1)) return false;          // scipy: coordinate outside -> cval
    const double fl = floor(cc);
    const double x = cc - fl, y = x, zc = 1.0 - x;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (zc * zc * (zc - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = zc * zc * zc / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
    const int st = (int)fl - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[k] = mirror_idx(st + k, a.n_in);
    return true;
}

// scipy's conversion of the f64 result to an integer output array (the map keeps its MRC mode through zoom): add 0.5 away
// from zero, clamp to the type's range, truncate (unsigned: negative -> 0).  The integers are stored as f32 (exact: 16 bits).
__device__ __forceinline__ double round_like_scipy(double t, int kind) {
    if (kind == MICA_MAP_F32) return t;
    double lo, hi;
    if (kind == MICA_MAP_U16) { t = t > 0.0 ? t + 0.5 : 0.0; lo = 0.0; hi = 65535.0; }
    else {
        t = t > 0.0 ? t + 0.5 : t - 0.5;
        lo = kind == MICA_MAP_I8 ? -128.0 : -32768.0;
        hi = kind == MICA_MAP_I8 ? 127.0 : 32767.0;
    }
    t = t > hi ? hi : t;
    t = t < lo ? lo : t;
    return trunc(t);
}

__global__ __launch_bounds__(256) void zoom_interp_kernel(const double* __restrict__ f, ZoomAxis a0, ZoomAxis a1, ZoomAxis a2,
                                                          int kind, float* __restrict__ out) {
    const int64_t total = (int64_t)a0.n_out * a1.n_out * a2.n_out;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int o2 = (int)(e % a2.n_out), o1 = (int)((e / a2.n_out) % a1.n_out), o0 = (int)(e / ((int64_t)a2.n_out * a1.n_out));
    int i0[4], i1[4], i2[4];
    double w0[4], w1[4], w2[4];
    const bool ok0 = axis_setup(a0, o0, i0, w0), ok1 = axis_setup(a1, o1, i1, w1), ok2 = axis_setup(a2, o2, i2, w2);
    const bool ok = ok0 && ok1 && ok2;
    double t = 0.0;
    if (ok) {
        for (int k0 = 0; k0 < 4; ++k0)
            for (int k1 = 0; k1 < 4; ++k1) {
                const double* row = f + ((int64_t)i0[k0] * a1.n_in + i1[k1]) * a2.n_in;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    double c = row[i2[k2]];
                    c *= w0[k0];
                    c *= w1[k1];
                    c *= w2[k2];
                    t += c;
                }
            }
    }
    out[e] = (float)round_like_scipy(t, kind);
}

int zoom_cubic_device(const float* d_in, int64_t n0, int64_t n1, int64_t n2, int64_t o0, int64_t o1, int64_t o2, int kind,
                      float* d_out, hipStream_t st, char* err, int errlen) {
    const int64_t n = n0 * n1 * n2;
    double* f = nullptr;
    if (hipMalloc((void**)&f, (size_t)n * sizeof(double)) != hipSuccess) { snprintf(err, errlen, "zoom: hipMalloc(%lld B) failed", (long long)(n * 8)); return -2; }
    hipLaunchKernelGGL(f32_to_f64_kernel, dim3(4096), dim3(256), 0, st, d_in, f, n);
    const int64_t dims[3] = {n0, n1, n2};
    const double z = SPLINE_POLE;
    // scipy filters axis 0, then 1, then 2
    {   // axis 0: lines (i1,i2), stride n1*n2
        int64_t nl = n1 * n2;
        hipLaunchKernelGGL(spline_prefilter_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, f, (int)n0, n1 * n2, nl, nl, (int64_t)0,
                           std::pow(z, (double)(n0 - 1)));
    }
    {   // axis 1: lines (i0,i2): base = i0*n1*n2 + i2, stride n2
        int64_t nl = n0 * n2;
        hipLaunchKernelGGL(spline_prefilter_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, f, (int)n1, n2, nl, n2, n1 * n2,
                           std::pow(z, (double)(n1 - 1)));
    }
    {   // axis 2: lines (i0,i1): base = (i0*n1+i1)*n2, stride 1
        int64_t nl = n0 * n1;
        hipLaunchKernelGGL(spline_prefilter_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, f, (int)n2, (int64_t)1, nl, (int64_t)1, n2,
                           std::pow(z, (double)(n2 - 1)));
    }
    (void)dims;
    ZoomAxis a[3];
    const int64_t ins[3] = {n0, n1, n2}, outs[3] = {o0, o1, o2};
    for (int k = 0; k < 3; ++k) {
        a[k].n_in = (int)ins[k];
        a[k].n_out = (int)outs[k];
        a[k].zf = outs[k] > 1 ? (double)(ins[k] - 1) / (double)(outs[k] - 1) : 1.0;
    }
    const int64_t total = o0 * o1 * o2;
    hipLaunchKernelGGL(zoom_interp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, f, a[0], a[1], a[2], kind, d_out);
    int rc = 0;
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { snprintf(err, errlen, "zoom: HIP failure"); rc = -2; }
    (void)hipFree(f);
    return rc;
}

}  // namespace mica

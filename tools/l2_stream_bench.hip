// Micro-benchmark: how many bytes per clock can a CU pull from L2 with the conv kernel's weight-fragment access pattern?
// 256 workgroups x 8 waves; every wave streams 1-KB fragments (global_load_dwordx4, 64 lanes x 16 B) from a buffer that all
// workgroups read at the same time (as the 32 CUs of an XCD do with a layer's packed weights): L2 hits after the first touch.
// The 2-D Winograd variant of DESIGN.md section 4 needs 43 B/clk/CU of this stream to break even and 85 to realise its MFMA saving.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_stream_bench.hip -o tools/l2_stream_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ __launch_bounds__(512, 1) void k(const f4* __restrict__ w, size_t frags_total, int iters, float* __restrict__ out, long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    size_t f = (size_t)wave * 9973;                    // waves walk different fragments; all workgroups walk the same ones
    for (int it = 0; it < iters; ++it) {
        f4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            v[j] = w[((f + j) % frags_total) * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc += v[j];
        f += INFLIGHT * 8;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
}

int main() {
    for (size_t mb : {1, 4, 16}) {
        const size_t bytes = mb << 20, frags = bytes / 1024;
        f4* w; float* out; long long* cyc;
        hipMalloc(&w, bytes); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
        hipMemset(w, 0, bytes);
        for (int inflight : {8, 16}) {
            const int iters = 4000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (inflight == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, w, frags, iters, out, cyc);
                else hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, w, frags, iters, out, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                std::vector<long long> hc(256); hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
                double cy = 0; for (auto v : hc) cy += (double)v / 256;
                const double per_cu = 8.0 * inflight * iters * 1024.0;
                if (rep) printf("buffer %zu MB, %2d fragments in flight per wave: %.2f ms, %.1f B/clk/CU (%.1f TB/s chip-wide), clock %.2f GHz\n", mb, inflight, ms,
                                per_cu / cy, per_cu * 256 / ms / 1e9, cy / ms / 1e6);
            }
        }
        hipFree(w); hipFree(out); hipFree(cyc);
    }
    return 0;
}

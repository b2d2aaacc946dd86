// Micro-benchmark: f16 MFMA throughput of the 32x32x16 and 16x16x32 shapes on random operands held in registers
// (development aid: does the 16x16x32 shape hold a higher clock under the power limit on this chip?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const half8* __restrict__ in, float* __restrict__ out, int iters, long long* __restrict__ cyc) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const long long t0 = clock64();
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) % 65536]; b[i] = in[(tid * 8 + 4 + i) % 65536]; }
    if (SHAPE == 0) {
        f16v acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {       // 8 tiles x 3 = 24 MFMA 32x32x16 (as the conv's tap)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) & 3], b[(i + 2) & 3], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 2) & 3], b[(i + 1) & 3], acc[i], 0, 0, 0);
            }
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
        out[tid] = s;
        if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
        return;
    } else {
        f4v acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {      // 32 tiles x 1.5 = 48 MFMA 16x16x32 = same FLOPs as 24 of 32x32x16
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
                if (i & 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + 1) & 3], b[(i + 2) & 3], acc[i], 0, 0, 0);
            }
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
        out[tid] = s;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
}

int main() {
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    half8* din; float* dout;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 512 * 4 * 4);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;
    long long* dcyc; hipMalloc(&dcyc, 256 * 8);
    for (int threads = 256; threads <= 512; threads += 256)
    for (int rep = 0; rep < 2; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, din, dout, iters, dcyc);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, din, dout, iters, dcyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = 256.0 * (threads / 64) * iters * 24 * 32768.0;    // waves x iters x MFMA-equivalents
            std::vector<long long> hc(256); hipMemcpy(hc.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost);
            double cy = 0; for (auto v : hc) cy += (double)v / 256;
            // MFMA pipe cycles one SIMD needs: (waves per SIMD) x iters x 24 x 32 cycles (32x32x16 = 8 passes; 16x16x32: two of 4 passes)
            double need = (threads / 256) * (double)iters * 24 * 32;
            printf("%d waves/SIMD shape %s: %.2f ms  %.1f TFLOP/s (f16 MFMA)  cycles %.0f  pipe busy %.3f  clock %.2f GHz\n", threads / 256,
                   shape == 0 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9, cy, need / cy, cy / ms / 1e6);
        }
    return 0;
}

// Micro-benchmark (round 6, DESIGN.md section 4 "Round 6" (c)): what serves a line that all eight XCDs read within the same moment?
// rocprofv3's FETCH_SIZE counts requests at each L2's fabric side, so a table that every XCD streams through its own 4-MB L2 is counted
// eight times per pass - as the packed weights of conv_wino43_kernel are, once per round of items.  Do those eight requests reach HBM?
//   shared : workgroup (xcd = blockIdx & 7, l = blockIdx >> 3) reads slice l of 32 of the buffer - the 32 CUs of an XCD cover the buffer
//            once, all eight XCDs read the SAME bytes at about the same time (fabric side: 8 x buffer per pass);
//   private: workgroup b reads slice b of 256 - every byte is read by one XCD only (fabric side: 1 x buffer per pass, all of it from HBM
//            when the buffer is larger than the 256-MB Infinity Cache).
// If `shared` moves 8 x the bytes of `private` through the fabric in about the same time, seven of its eight requests per line were
// served on-die.  Build: hipcc --offload-arch=gfx950 -O3 tools/xcd_share_bench.hip -o tools/xcd_share_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void k(const f4* __restrict__ buf, size_t slice_vec, int shared, int passes, float* __restrict__ out) {
    const size_t slice = shared ? (size_t)(blockIdx.x >> 3) : (size_t)blockIdx.x;
    const f4* p = buf + slice * slice_vec;
    f4 acc = {0, 0, 0, 0};
    for (int pass = 0; pass < passes; ++pass)
        for (size_t i = threadIdx.x; i < slice_vec; i += 512 * 4) {          // four independent 16-byte loads per thread in flight
            f4 a = p[i], b = i + 512 < slice_vec ? p[i + 512] : acc, c = i + 1024 < slice_vec ? p[i + 1024] : acc, d = i + 1536 < slice_vec ? p[i + 1536] : acc;
            acc += a + b + c + d;
        }
    out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;                          // 0 / 1: run one mode only (for a counter pass)
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (size_t mb : {32, 2048}) {
        const size_t bytes = mb << 20;
        f4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
        for (int shared : {0, 1}) {
            if (only >= 0 && shared != only) continue;
            const size_t slice_vec = bytes / 16 / (shared ? 32 : 256);
            const int passes = mb >= 1024 ? 3 : 60;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, buf, slice_vec, shared, passes, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double fabric = (double)bytes * passes * (shared ? 8 : 1);
            printf("%4zu-MB buffer, %-7s: %8.3f ms for %d passes; bytes requested by the CUs %.1f GB; fabric-side bytes (what FETCH_SIZE counts, if nothing stays in an L2) %.1f GB = %.2f TB/s; "
                   "HBM-side bytes if every line left HBM once per pass %.1f GB = %.2f TB/s\n", mb, shared ? "shared" : "private", best, passes,
                   fabric / 1e9, fabric / 1e9, fabric / best / 1e9, (double)bytes * passes / 1e9, (double)bytes * passes / best / 1e9);
        }
        hipFree(buf);
    }
    return 0;
}

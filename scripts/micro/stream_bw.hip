// Calibration: how fast can this GPU stream 380 MB in + 80 MB out with plain 16 B/lane loads?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int UNROLL>
__global__ void __launch_bounds__(1024) stream_kernel(const uint4* __restrict__ in, size_t n16, int* __restrict__ out, size_t nout)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    size_t i = tid;
    for (; i + (UNROLL - 1) * nth < n16; i += UNROLL * nth) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * nth];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += nth) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    for (size_t o = tid; o < nout; o += nth) out[o] = (int)(acc + o);
}
int main()
{
    const size_t bytes = 380000000, n16 = bytes / 16, nout = 20000000;
    uint4* in; int* out;
    hipMalloc(&in, bytes); hipMalloc(&out, nout * 4);
    hipMemset(in, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(1024), 0, 0, in, n16, out, nout);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("grid %4d x1024 unroll4: %.1f us/launch  -> %.2f TB/s (460 MB)\n", grid, ms / 20 * 1e3, 460e6 / (ms / 20 * 1e-3) / 1e12);
        }
    }
    for (int grid : {256, 1024}) {
        hipEventRecord(e0);
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(stream_kernel<8>, dim3(grid), dim3(1024), 0, 0, in, n16, out, nout);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %4d x1024 unroll8: %.1f us/launch  -> %.2f TB/s\n", grid, ms / 20 * 1e3, 460e6 / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}

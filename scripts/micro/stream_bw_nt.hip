// Does a non-temporal policy on the once-read stream / once-written output change the streaming rate?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0 plain, 1 nt loads, 2 nt loads + nt stores, 3 nt stores only
__global__ void __launch_bounds__(1024) k(const uint4* __restrict__ in, size_t n16, int* __restrict__ out, size_t nout)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    size_t i = tid;
    for (; i + 3 * nth < n16; i += 4 * nth) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 1 || MODE == 2) {
                const unsigned* p = reinterpret_cast<const unsigned*>(in + i + u * nth);
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
                v[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else v[u] = in[i + u * nth];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += nth) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    for (size_t o = tid; o < nout; o += nth) {
        if (MODE >= 2) __builtin_nontemporal_store((int)(acc + o), out + o); else out[o] = (int)(acc + o);
    }
}
template <int MODE> void run(const uint4* in, size_t n16, int* out, size_t nout, const char* name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 20; ++j) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, in, n16, out, nout);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-28s %.1f us/launch -> %.2f TB/s\n", name, ms / 20 * 1e3, 460e6 / (ms / 20 * 1e-3) / 1e12);
    }
}
int main()
{
    const size_t bytes = 380000000, n16 = bytes / 16, nout = 20000000;
    uint4* in; int* out; hipMalloc(&in, bytes); hipMalloc(&out, nout * 4); hipMemset(in, 1, bytes);
    run<0>(in, n16, out, nout, "plain");
    run<1>(in, n16, out, nout, "nt loads");
    run<2>(in, n16, out, nout, "nt loads + nt stores");
    run<3>(in, n16, out, nout, "nt stores");
    run<0>(in, n16, out, nout, "plain (again)");
    return 0;
}

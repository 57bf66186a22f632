// stream_mm.hip -- the byte mixes of the BATCHED score launches (BASELINE config 5) as bare streams: every lane reads
// LOADS x 16 B of k-mers and writes one 16-byte store into each of STORES separate score arrays (four int32 scores of
// STORES motifs), interleaved as a scoring kernel must.  What the memory system sustains for W/MM + 4 bytes per pair
// when the mix is write-heavy (W = 8, MM = 3: 8 B read per 12 B written).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_mm.hip -o scripts/micro/stream_mm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
struct Outs { int *p[3]; };
template <int LOADS, int STORES>
__global__ void __launch_bounds__(1024) k(const u4 *__restrict__ in, size_t iters, Outs out)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    u4 buf[LOADS];
#pragma unroll
    for (int u = 0; u < LOADS; ++u) buf[u] = __builtin_nontemporal_load(in + (size_t)u * nth + tid);
    for (size_t it = 0; it < iters; ++it) {
        unsigned acc = 0;
#pragma unroll
        for (int u = 0; u < LOADS; ++u) acc ^= buf[u].x ^ buf[u].y ^ buf[u].z ^ buf[u].w;
        if (it + 1 < iters) {
#pragma unroll
            for (int u = 0; u < LOADS; ++u) buf[u] = __builtin_nontemporal_load(in + ((it + 1) * LOADS + u) * nth + tid);
        }
        const size_t o = (it * nth + tid) * 4;
#pragma unroll
        for (int s = 0; s < STORES; ++s) {
            const i4 v = {(int)acc + s, (int)acc + 1, (int)acc + 2, (int)acc + 3};
            __builtin_nontemporal_store(v, reinterpret_cast<i4 *>(out.p[s] + o));
        }
    }
}
template <int LOADS, int STORES> void run(const u4 *in, Outs out, size_t rows, const char *name)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t nth = 256 * 1024, iters = rows / 4 / nth;
    const double bytes = (double)iters * nth * 16.0 * (LOADS + STORES);
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<LOADS, STORES>), dim3(256), dim3(1024), 0, 0, in, iters, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    std::printf("%-34s %8.1f us avg %8.1f us best  %6.2f TB/s  (%.0f MB per launch, %zu rows)\n", name, sum / 10 * 1e3, best * 1e3,
                bytes / (sum / 10 * 1e-3) / 1e12, bytes / 1e6, iters * nth * 4);
}
int main()
{
    const size_t rows = 100000000;
    u4 *in; Outs out{};
    if (hipMalloc(&in, rows * 28) != hipSuccess) return 1;
    (void)hipMemset(in, 1, rows * 28);
    for (int s = 0; s < 3; ++s) if (hipMalloc(&out.p[s], rows * 4 + 4096) != hipSuccess) return 1;
    run<2, 3>(in, out, rows, "W=8  MM=3 (2 loads, 3 stores)");
    run<3, 3>(in, out, rows, "W=12 MM=3 (3 loads, 3 stores)");
    run<4, 3>(in, out, rows, "W=16 MM=3 (4 loads, 3 stores)");
    run<5, 3>(in, out, rows, "W=20 MM=3 (5 loads, 3 stores)");
    run<6, 2>(in, out, rows, "W=24 MM=2 (6 loads, 2 stores)");
    run<5, 1>(in, out, rows, "W=20 MM=1 (5 loads, 1 store)");
    run<7, 1>(in, out, rows, "W=28 MM=1 (7 loads, 1 store)");
    run<2, 1>(in, out, rows, "W=8  MM=1 (2 loads, 1 store)");
    run<1, 3>(in, out, rows, "W=4  MM=3 (1 load, 3 stores)");
    return 0;
}

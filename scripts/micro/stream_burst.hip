// stream_burst.hip -- does the memory system like the score kernel's stores in BIGGER BURSTS?  (first candidate:
// "scores of k chunks parked in the wave's LDS strip and written as one k-KiB burst: fewer read<->write turnarounds per byte".)
// The W = 19 byte mix as a bare stream -- per lane-step 5 x 16 B non-temporal loads and one 16 B store, 2e7 rows: 380 MB read,
// 80 MB written -- in three shapes:
//   interleaved   every step: its loads, then its store (1 KiB per wave and step): what score_quad_kernel does;
//   burst K       a wave takes K CONSECUTIVE chunks, keeps the K results (registers here; an LDS strip in a kernel: the
//                 same to the memory system) and issues K stores back to back into K consecutive KiB;
//   burst K, wg   ... and the waves of a workgroup take adjacent runs, so that a workgroup writes 16 K KiB contiguous.
// Store policy as the kernel's: write-through (sc0 sc1) below 96 MiB of scores per launch.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_burst.hip -o scripts/micro/stream_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int LOADS = 5;

template <int K, bool THROUGH>
__global__ void __launch_bounds__(1024) k_burst(const u4 *__restrict__ in, size_t n_chunks, unsigned *__restrict__ out)
{
    // chunk = 64 lanes x LOADS x 16 B in, 64 x 16 B out.  Wave w of the grid takes runs of K consecutive chunks, runs dealt
    // round-robin over the waves.
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t n_runs = n_chunks / K;
    for (size_t run = wave; run < n_runs; run += n_waves) {
        u4 res[K];
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const u4 *src = in + ((run * K + c) * LOADS) * 64 + lane;
            unsigned acc = 0;
#pragma unroll
            for (int u = 0; u < LOADS; ++u) {
                const u4 v = __builtin_nontemporal_load(src + (size_t)u * 64);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
            res[c] = u4{acc, acc + 1, acc + 2, acc + 3};
        }
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (run * K + c) * 256, 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(res[c], rs, lane * 16, 0, THROUGH ? 3 : 2 /* sc0 sc1 : nt */);
        }
    }
}

template <int K, bool THROUGH> void run(const u4 *in, unsigned *out, size_t rows, const char *name)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t n_chunks = rows / 256;
    const double bytes = (double)n_chunks * 64 * 16.0 * (LOADS + 1);
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 22; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_burst<K, THROUGH>), dim3(256), dim3(1024), 0, 0, in, n_chunks, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    std::printf("%-44s %8.1f us avg %8.1f us best  %6.2f TB/s avg  (%.0f MB per launch)\n", name, sum / 20 * 1e3, best * 1e3,
                bytes / (sum / 20 * 1e-3) / 1e12, bytes / 1e6);
}
int main()
{
    const size_t rows = 20000000 / 2048 * 2048;
    u4 *in; unsigned *out;
    if (hipMalloc(&in, rows / 4 * LOADS * 16 + 4096) != hipSuccess) return 1;
    (void)hipMemset(in, 1, rows / 4 * LOADS * 16);
    if (hipMalloc(&out, rows * 4 + 4096) != hipSuccess) return 1;
    for (int pass = 0; pass < 2; ++pass) {
        run<1, true>(in, out, rows, "interleaved (K = 1), write-through");
        run<2, true>(in, out, rows, "burst K = 2, write-through");
        run<4, true>(in, out, rows, "burst K = 4, write-through");
        run<8, true>(in, out, rows, "burst K = 8, write-through");
        run<1, false>(in, out, rows, "interleaved (K = 1), nt stores");
        run<4, false>(in, out, rows, "burst K = 4, nt stores");
        run<8, false>(in, out, rows, "burst K = 8, nt stores");
    }
    return 0;
}

// stream_mix.hip -- the byte mix of the score kernel (19 bytes read per 4 bytes written) as a bare stream:
// does interleaving the stores with the loads (what a scoring kernel must do) cost bandwidth against
// reading everything first and writing afterwards (scripts/micro/stream_bw_nt.hip)?
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_mix.hip -o scripts/micro/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
// every lane: LOADS x 16 B loads then one store of 16 B (MODE 0), four dword stores (MODE 1), or no store (MODE 2)
template <int MODE, int LOADS, int DEPTH>
__global__ void __launch_bounds__(1024) k(const u4 *__restrict__ in, size_t n16, int *__restrict__ out)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const size_t iters = n16 / (nth * LOADS);
    u4 buf[DEPTH][LOADS];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int u = 0; u < LOADS; ++u)
            if ((size_t)d < iters) buf[d][u] = __builtin_nontemporal_load(in + ((size_t)d * LOADS + u) * nth + tid);
    for (size_t it = 0; it < iters; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (it + d >= iters) break;
            unsigned acc = 0;
#pragma unroll
            for (int u = 0; u < LOADS; ++u) acc ^= buf[d][u].x ^ buf[d][u].y ^ buf[d][u].z ^ buf[d][u].w;
            if (it + d + DEPTH < iters) {
#pragma unroll
                for (int u = 0; u < LOADS; ++u)
                    buf[d][u] = __builtin_nontemporal_load(in + ((it + d + DEPTH) * LOADS + u) * nth + tid);
            }
            const size_t o = ((it + d) * nth + tid) * 4;
            if (MODE == 0) {
                const i4 v = {(int)acc, (int)acc + 1, (int)acc + 2, (int)acc + 3};
                __builtin_nontemporal_store(v, reinterpret_cast<i4 *>(out + o));
            } else if (MODE == 1) {
                const size_t ob = (it + d) * nth * 4 + (tid & ~(size_t)63) * 4 + (tid & 63);
                for (int j = 0; j < 4; ++j) __builtin_nontemporal_store((int)acc + j, out + ob + j * 64);
            } else if (MODE == 3 || MODE == 4) {   // buffer stores with a cache policy: 3 = sc0 sc1 (write-through), 4 = nt
                const u4 v = {acc, acc + 1, acc + 2, acc + 3};
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (o & ~(size_t)255), 0, 1024, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(threadIdx.x & 63) * 16, 0, MODE == 3 ? 17 : 2);
            } else if (MODE == 5) {                // plain store
                const i4 v = {(int)acc, (int)acc + 1, (int)acc + 2, (int)acc + 3};
                *reinterpret_cast<i4 *>(out + o) = v;
            } else if (acc == 0x12345678u) {
                out[o] = 1;
            }
        }
    }
}
template <int MODE, int LOADS, int DEPTH> void run(const u4 *in, size_t n16, int *const *outs, const char *name)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t nth = 256 * 1024, iters = n16 / (nth * LOADS);
    const double bytes = (double)iters * nth * (LOADS * 16 + (MODE == 2 ? 0 : 16));
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        for (int j = 0; j < 21; ++j) hipLaunchKernelGGL((k<MODE, LOADS, DEPTH>), dim3(256), dim3(1024), 0, 0, in, n16, outs[j % 3]);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-52s %.1f us/launch -> %.2f TB/s (%.0f MB)\n", name, ms / 21 * 1e3, bytes / (ms / 21 * 1e-3) / 1e12, bytes / 1e6);
    }
}
// stream_mix [input bytes]   (default 380e6: the bench's config 2; 2375e6: config 3).  Outputs rotate over three buffers.
int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? (size_t)atof(argv[1]) : 380000000, n16 = bytes / 16;
    u4 *in; int *outs[3];
    (void)hipMalloc(&in, bytes);
    for (int i = 0; i < 3; ++i) (void)hipMalloc(&outs[i], bytes / 4 + (1 << 20));
    setvbuf(stdout, nullptr, _IONBF, 0); (void)hipMemset(in, 1, bytes);
    run<2, 5, 2>(in, n16, outs, "loads only, 5 x 16 B per step, depth 2");
    run<0, 5, 2>(in, n16, outs, "5 loads + one 16 B nt store per step, depth 2");
    run<4, 5, 2>(in, n16, outs, "5 loads + one 16 B nt BUFFER store, depth 2");
    run<3, 5, 2>(in, n16, outs, "5 loads + one 16 B sc0 sc1 buffer store, depth 2");
    run<5, 5, 2>(in, n16, outs, "5 loads + one 16 B plain store, depth 2");
    run<3, 5, 1>(in, n16, outs, "5 loads + one 16 B sc0 sc1 buffer store, depth 1");
    run<0, 5, 1>(in, n16, outs, "5 loads + one 16 B nt store per step, depth 1");
    run<1, 5, 2>(in, n16, outs, "5 loads + four nt dword stores per step, depth 2");
    run<0, 5, 4>(in, n16, outs, "5 loads + one 16 B nt store per step, depth 4");
    return 0;
}

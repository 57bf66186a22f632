// stream_asm.hip -- does the loop-top `s_waitcnt vmcnt(0)` the compiler puts in front of the prefetched k-mers cost the
// interleaved stream its 6 %?  The counter is in order and counts stores too: waiting for the loads of step i+1 with
// vmcnt(0) also waits for the STORE of step i, issued after them.  Variant B issues the loads from inline asm (the
// compiler does not see them) and waits with vmcnt(1) by hand: the newest operation -- the store -- may stay in flight.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_asm.hip -o scripts/micro/stream_asm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
constexpr int LOADS = 5;
__global__ void __launch_bounds__(1024) k_plain(const u4 *__restrict__ in, size_t iters, int *__restrict__ out)
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
        const i4 v = {(int)acc, (int)acc + 1, (int)acc + 2, (int)acc + 3};
        __builtin_nontemporal_store(v, reinterpret_cast<i4 *>(out + (it * nth + tid) * 4));
    }
}
__device__ __forceinline__ u4 ld_nt(const u4 *p)
{
    u4 r;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__global__ void __launch_bounds__(1024) k_asm(const u4 *__restrict__ in, size_t iters, int *__restrict__ out)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    u4 b0, b1, b2, b3, b4;
    b0 = ld_nt(in + 0 * nth + tid); b1 = ld_nt(in + 1 * nth + tid); b2 = ld_nt(in + 2 * nth + tid);
    b3 = ld_nt(in + 3 * nth + tid); b4 = ld_nt(in + 4 * nth + tid);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4));
    for (size_t it = 0; it < iters; ++it) {
        const unsigned acc = b0.x ^ b0.y ^ b0.z ^ b0.w ^ b1.x ^ b1.y ^ b1.z ^ b1.w ^ b2.x ^ b2.y ^ b2.z ^ b2.w ^ b3.x ^ b3.y ^ b3.z ^
                             b3.w ^ b4.x ^ b4.y ^ b4.z ^ b4.w;
        const bool more = it + 1 < iters;
        u4 n0 = b0, n1 = b1, n2 = b2, n3 = b3, n4 = b4;
        if (more) {
            const u4 *p = in + (it + 1) * LOADS * nth + tid;
            n0 = ld_nt(p); n1 = ld_nt(p + nth); n2 = ld_nt(p + 2 * nth); n3 = ld_nt(p + 3 * nth); n4 = ld_nt(p + 4 * nth);
        }
        const i4 v = {(int)acc, (int)acc + 1, (int)acc + 2, (int)acc + 3};
        __builtin_nontemporal_store(v, reinterpret_cast<i4 *>(out + (it * nth + tid) * 4));
        // the five loads are older than the store: with one operation allowed in flight they have all arrived
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4));
        b0 = n0; b1 = n1; b2 = n2; b3 = n3; b4 = n4;
    }
}
int main()
{
    const size_t rows = 20000000, nth = 256 * 1024, iters = rows / 4 / nth;
    u4 *in; int *out;
    if (hipMalloc(&in, rows * 20 + 4096) != hipSuccess || hipMalloc(&out, rows * 4 + 4096) != hipSuccess) return 1;
    (void)hipMemset(in, 1, rows * 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = (double)iters * nth * 16.0 * (LOADS + 1);
    for (int variant = 0; variant < 4; ++variant) {
        float sum = 0, best = 1e9f;
        for (int rep = 0; rep < 22; ++rep) {
            (void)hipEventRecord(e0);
            if (variant % 2 == 0) hipLaunchKernelGGL(k_plain, dim3(256), dim3(1024), 0, 0, in, iters, out);
            else hipLaunchKernelGGL(k_asm, dim3(256), dim3(1024), 0, 0, in, iters, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        std::printf("%s: %7.1f us avg %7.1f us best  %.2f TB/s (%.0f MB)\n", variant % 2 ? "asm loads, vmcnt(1)      " : "compiler's waits (vmcnt 0)",
                    sum / 20 * 1e3, best * 1e3, bytes / (sum / 20 * 1e-3) / 1e12, bytes / 1e6);
    }
    unsigned chk[4]; (void)hipMemcpy(chk, out, sizeof chk, hipMemcpyDeviceToHost); std::printf("check %u %u\n", chk[0], chk[1]);
    return 0;
}

// stream_calib.hip -- gfm_calibrate_stream: the bare-stream floor of the score kernel's byte mix on THIS device.
// Part of libgrafimo_hip.so.  bench.py reports it as `peak_measured` next to the 8 TB/s vendor peak (SURVEY 8d asks
// for both): score_quad_kernel<W, 1> reads 4 W bytes and writes 16 per lane and step, interleaved in one pass; this
// kernel does exactly that and nothing else (no LDS, no lookups), with the same load policy (nt) and the store policy
// the score kernel would choose.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "grafimo_hip.h"

extern "C" void gfm_set_error_(const char *msg);

namespace {

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// every lane: LOADS x 16 B nt loads, then one 16 B buffer store (THROUGH: sc0 sc1, else nt); one step prefetched
template <int LOADS, bool THROUGH>
__global__ void __launch_bounds__(1024) stream_mix_kernel(const u4 *__restrict__ in, size_t n16, int *__restrict__ out)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const size_t iters = n16 / (nth * LOADS);
    u4 buf[LOADS];
    if (iters > 0) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) buf[u] = __builtin_nontemporal_load(in + (size_t)u * nth + tid);
    }
    for (size_t it = 0; it < iters; ++it) {
        unsigned acc = 0;
#pragma unroll
        for (int u = 0; u < LOADS; ++u) acc ^= buf[u].x ^ buf[u].y ^ buf[u].z ^ buf[u].w;
        if (it + 1 < iters) {
#pragma unroll
            for (int u = 0; u < LOADS; ++u) buf[u] = __builtin_nontemporal_load(in + ((it + 1) * LOADS + u) * nth + tid);
        }
        const size_t o = (it * nth + tid) * 4;
        const u4 v = {acc, acc + 1, acc + 2, acc + 3};
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (o & ~(size_t)255), 0, 1024, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(threadIdx.x & 63) * 16, 0, THROUGH ? 17 : 2);
    }
}

template <int LOADS> void launch(bool through, int grid, const u4 *in, size_t n16, int *out)
{
    if (through)
        hipLaunchKernelGGL((stream_mix_kernel<LOADS, true>), dim3(grid), dim3(1024), 0, nullptr, in, n16, out);
    else
        hipLaunchKernelGGL((stream_mix_kernel<LOADS, false>), dim3(grid), dim3(1024), 0, nullptr, in, n16, out);
}

int cfail(int code, const std::string &msg)
{
    gfm_set_error_(msg.c_str());
    return code;
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int gfm_calibrate_stream(int loads_per_store, int64_t in_bytes,
                                                                          int store_policy, int launches,
                                                                          double *us_per_launch, double *bytes_per_launch)
{
    if (loads_per_store < 1 || loads_per_store > 16 || in_bytes < (1 << 20) || launches < 1 || !us_per_launch)
        return cfail(GFM_ERR_INVALID, "gfm_calibrate_stream: bad argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        return cfail(GFM_ERR_NODEVICE, "no HIP device available");
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = cus > 0 ? cus : 256;
    const size_t n16 = (size_t)in_bytes / 16, nth = (size_t)grid * 1024;
    const size_t iters = n16 / (nth * (size_t)loads_per_store);
    const size_t out_bytes = iters * nth * 16 + (1 << 20);
    u4 *ins[2] = {nullptr, nullptr};     // two inputs used in turn, like the bench's k-mer buffers: no launch re-reads
                                         // what the 256 MiB Infinity Cache may still hold of the previous one
    int *outs[3] = {nullptr, nullptr, nullptr};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = GFM_OK;
    auto bad = [&](hipError_t e) {
        if (e == hipSuccess) return false;
        rc = cfail(GFM_ERR_HIP, std::string("gfm_calibrate_stream: ") + hipGetErrorString(e));
        return true;
    };
    do {
        if (bad(hipMalloc(&ins[0], n16 * 16)) || bad(hipMalloc(&ins[1], n16 * 16))) break;
        bool failed = false;
        for (int i = 0; i < 3 && !failed; ++i) failed = bad(hipMalloc(&outs[i], out_bytes));
        if (failed) break;
        if (bad(hipMemset(ins[0], 1, n16 * 16)) || bad(hipMemset(ins[1], 2, n16 * 16))) break;
        if (bad(hipEventCreate(&e0)) || bad(hipEventCreate(&e1))) break;
        for (int rep = 0; rep < 2; ++rep) {          // the first repetition warms up
            if (bad(hipEventRecord(e0, nullptr))) break;
            for (int j = 0; j < launches; ++j) {
                switch (loads_per_store) {
#define GFM_CAL(N) case N: launch<N>(store_policy != 0, grid, ins[j & 1], n16, outs[j % 3]); break;
                    GFM_CAL(1) GFM_CAL(2) GFM_CAL(3) GFM_CAL(4) GFM_CAL(5) GFM_CAL(6) GFM_CAL(7) GFM_CAL(8)
                    GFM_CAL(9) GFM_CAL(10) GFM_CAL(11) GFM_CAL(12) GFM_CAL(13) GFM_CAL(14) GFM_CAL(15) GFM_CAL(16)
#undef GFM_CAL
                }
            }
            if (bad(hipGetLastError()) || bad(hipEventRecord(e1, nullptr)) || bad(hipEventSynchronize(e1))) break;
            float ms = 0.f;
            if (bad(hipEventElapsedTime(&ms, e0, e1))) break;
            *us_per_launch = (double)ms * 1e3 / launches;
        }
        if (bytes_per_launch) *bytes_per_launch = (double)iters * (double)nth * (16.0 * loads_per_store + 16.0);
    } while (false);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (int i = 0; i < 2; ++i)
        if (ins[i]) (void)hipFree(ins[i]);
    for (int i = 0; i < 3; ++i)
        if (outs[i]) (void)hipFree(outs[i]);
    return rc;
}

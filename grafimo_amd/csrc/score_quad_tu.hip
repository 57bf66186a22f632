// score_quad_tu.hip -- the instantiations of score_quad_kernel<W, MM> (gfm_score_quad.hpp) for sixteen widths and one
// MM, and their launcher; compiled eight times (-DGFM_QUAD_GROUP=0..3 -DGFM_QUAD_MM=1, groups 0 and 1 also with
// MM = 2 and 3) into libgrafimo_hip.so, side by side.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>

#include "grafimo_hip.h"

#include "gfm_common.hpp"
#include "gfm_score_quad.hpp"

#if !defined(GFM_QUAD_GROUP) || !defined(GFM_QUAD_MM)
#error "compile with -DGFM_QUAD_GROUP=0..3 -DGFM_QUAD_MM=1..3"
#endif

extern "C" void gfm_set_error_(const char *msg);

namespace {

int qfail(int code, const std::string &msg)
{
    gfm_set_error_(msg.c_str());
    return code;
}
#define Q_TRY(expr)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return qfail(GFM_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_)); \
    } while (0)

template <int W, int MM, bool STORE>
int launch_quad_k(const uint8_t *d_kmers, long long n, long long row_base, const ScoreArgs<MM> &args, size_t lds,
                  int nslabs, int waves, hipStream_t st, bool prepare, hipEvent_t ev0, hipEvent_t ev1)
{
    auto kern = score_quad_kernel<W, MM, STORE>;
    if (prepare) {  // from gfm_motif_create (never inside a stream capture); once per process and device
        static std::atomic<unsigned long long> done{0ull};       // one bit per device for this instantiation
        int dev = 0;
        Q_TRY(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return GFM_OK;
        Q_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kMaxLdsBytes));
        // the kernel addresses its lookup tables by absolute LDS offset: its dynamic LDS must start at 0
        hipFuncAttributes attr;
        Q_TRY(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kern)));
        if (attr.sharedSizeBytes != 0) return qfail(GFM_ERR_HIP, "score kernel carries static LDS (expected none)");
        done.fetch_or(bit, std::memory_order_release);
        return GFM_OK;
    }
    // Events ride ON the dispatch packet (start / completion of this kernel) instead of being recorded around it:
    // an event recorded behind the kernel is a barrier packet of its own in the queue, and the next score kernel
    // of the stream starts several microseconds later for each of them (scripts/gap_micro.py).
    if (ev0 || ev1)
        hipExtLaunchKernelGGL(kern, dim3(nslabs), dim3(waves * kWave), lds, st, ev0, ev1, 0, d_kmers, n, row_base, args);
    else
        hipLaunchKernelGGL(kern, dim3(nslabs), dim3(waves * kWave), lds, st, d_kmers, n, row_base, args);
    Q_TRY(hipGetLastError());
    return GFM_OK;
}

// the storing instantiation, or -- args.store_through == 2: the caller passed d_scores == NULL -- the one without score stores
template <int W, int MM>
int launch_quad_t(const uint8_t *d_kmers, long long n, long long row_base, const ScoreArgs<MM> &args, size_t lds,
                  int nslabs, int waves, hipStream_t st, bool prepare, hipEvent_t ev0, hipEvent_t ev1)
{
    if (prepare) {
        const int rc = launch_quad_k<W, MM, true>(d_kmers, n, row_base, args, lds, nslabs, waves, st, true, ev0, ev1);
        return rc ? rc : launch_quad_k<W, MM, false>(d_kmers, n, row_base, args, lds, nslabs, waves, st, true, ev0, ev1);
    }
    if (args.store_through == 2) return launch_quad_k<W, MM, false>(d_kmers, n, row_base, args, lds, nslabs, waves, st, false, ev0, ev1);
    return launch_quad_k<W, MM, true>(d_kmers, n, row_base, args, lds, nslabs, waves, st, false, ev0, ev1);
}

}  // namespace

#define GFM_QUAD_NAME2(g, mm) gfm_quad_launch_g##g##_m##mm
#define GFM_QUAD_NAME(g, mm) GFM_QUAD_NAME2(g, mm)

extern "C" __attribute__((visibility("hidden"))) int GFM_QUAD_NAME(GFM_QUAD_GROUP, GFM_QUAD_MM)(
    int W, const uint8_t *d_kmers, long long n, long long row_base, const void *score_args, size_t lds, int nslabs,
    int waves, void *stream, int prepare, void *ev0, void *ev1)
{
    const ScoreArgs<GFM_QUAD_MM> &args = *static_cast<const ScoreArgs<GFM_QUAD_MM> *>(score_args);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t e0 = static_cast<hipEvent_t>(ev0), e1 = static_cast<hipEvent_t>(ev1);
#define GFM_Q(N) case N: return launch_quad_t<N, GFM_QUAD_MM>(d_kmers, n, row_base, args, lds, nslabs, waves, st, prepare != 0, e0, e1);
    constexpr int B = 16 * GFM_QUAD_GROUP;
    switch (W) {
#ifdef GFM_ONLY_W   // development builds (scripts/lab_build.sh): one width, seconds to compile
#if (GFM_ONLY_W - 1) / 16 == GFM_QUAD_GROUP
        GFM_Q(GFM_ONLY_W)
#endif
#else
        GFM_Q(B + 1) GFM_Q(B + 2) GFM_Q(B + 3) GFM_Q(B + 4) GFM_Q(B + 5) GFM_Q(B + 6) GFM_Q(B + 7) GFM_Q(B + 8)
        GFM_Q(B + 9) GFM_Q(B + 10) GFM_Q(B + 11) GFM_Q(B + 12) GFM_Q(B + 13) GFM_Q(B + 14) GFM_Q(B + 15) GFM_Q(B + 16)
#endif
        default: return qfail(GFM_ERR_INVALID, "unsupported width " + std::to_string(W));
    }
#undef GFM_Q
}

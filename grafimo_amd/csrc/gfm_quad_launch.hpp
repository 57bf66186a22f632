// gfm_quad_launch.hpp -- geometry of score_quad_kernel<W> and the launchers of its four translation units
// Part of libgrafimo_hip.so (included by grafimo_hip.hip and score_quad_tu.hip).
#pragma once

#include <cstddef>
#include <cstdint>

#include "gfm_score_kernels.hpp"

namespace {

constexpr int kQuadRows = 256;   // k-mers per wave per step: 64 lanes x 4 rows
#ifndef GFM_QUAD_DEPTH
#define GFM_QUAD_DEPTH 1
#endif
// cache-policy bits of the score stores (gfx950 cpol: 1 sc0, 2 nt, 16 sc1)
constexpr int kStoreThrough = 17;   // sc0 sc1: write-through, lines stay allocated (Infinity Cache)
constexpr int kStoreStream = 2;     // nt: streaming
constexpr int kQuadDepth = GFM_QUAD_DEPTH;    // chunks in flight per wave beside the one being scored

__host__ __device__ constexpr int quad_pad(int W) { return (W % 8 == 0) ? 16 : 0; }
__host__ __device__ constexpr int quad_pitch(int W) { return 4 * W + quad_pad(W); }       // bytes per lane
__host__ __device__ constexpr int quad_stage_bytes(int W) { return 64 * quad_pitch(W) + 16; }
__host__ __device__ constexpr int quad_strip_stride(int W, int mm) { return quad_stage_bytes(W) + mm * kHitQueue * 8; }
// pair tables: uint16 entries for one motif, ONE table of packed 64-bit entries for two or three
__host__ __device__ constexpr int quad_tab_bytes(int W, int mm) { return 2 * ((W + 3) / 4) * 64 * (mm == 1 ? 2 : 8); }
constexpr int kQuadMaxBatchWidth = 32;   // widths up to this have the MM = 2, 3 instantiations


}  // namespace

// One per translation unit of score_quad_tu.hip: -DGFM_QUAD_GROUP=g holds widths 16g+1 .. 16g+16, -DGFM_QUAD_MM=mm the
// number of motifs per launch (MM = 2, 3 exist for widths <= kQuadMaxBatchWidth: groups 0 and 1).
// `score_args` points at a ScoreArgs<mm>; `prepare` != 0: set the kernel's LDS attribute instead of launching;
// ev0 / ev1 (hipEvent_t or NULL) bracket the launch.  Errors go to gfm_last_error().
#define GFM_QUAD_LAUNCH_DECL(g, mm)                                                                                    \
    extern "C" __attribute__((visibility("hidden"))) int gfm_quad_launch_g##g##_m##mm(                                \
        int W, const uint8_t *d_kmers, long long n, long long row_base, const void *score_args, size_t lds, int nslabs, \
        int waves, void *stream, int prepare, void *ev0, void *ev1);
GFM_QUAD_LAUNCH_DECL(0, 1)
GFM_QUAD_LAUNCH_DECL(1, 1)
GFM_QUAD_LAUNCH_DECL(2, 1)
GFM_QUAD_LAUNCH_DECL(3, 1)
GFM_QUAD_LAUNCH_DECL(0, 2)
GFM_QUAD_LAUNCH_DECL(1, 2)
GFM_QUAD_LAUNCH_DECL(0, 3)
GFM_QUAD_LAUNCH_DECL(1, 3)

// gfm_hit_sort.hpp -- ascending order of the packed hit entries the device hands back
// Part of libgrafimo_hip.so (host side; included by grafimo_hip.hip and scan_stream.cpp).
//
// The score kernels append (row << GFM_HIT_SCORE_BITS | scaled score) entries in the order their wavefronts finish;
// the reference reports hits in row order (score_sequences.py:273-321 walks the rows as they come), so the host puts
// them in order.  Rows are distinct, so the order of the ROW bits is the order of the entries: an LSD radix sort over
// those bits only, 11 bits a pass (2e7 rows: 3 passes).  std::sort took 10 ms for the 193 000 hits of a 2e7-row scan
// on the calling thread -- a quarter of the scan; this takes under 1 ms.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace gfm_hit_sort {

inline void sort_packed(int64_t *v, size_t n, int score_bits)
{
    if (n < 2) return;
    if (n < 256) { std::sort(v, v + n); return; }
    uint64_t any = 0;
    for (size_t i = 0; i < n; ++i) any |= (uint64_t)v[i];
    if ((int64_t)any < 0) { std::sort(v, v + n); return; }      // (never: rows are non-negative)
    int top = 64 - __builtin_clzll(any | 1);                      // bits in use
    if (top <= score_bits) { std::sort(v, v + n); return; }      // one row?  cannot be with n >= 2 distinct rows
    constexpr int kDigit = 11;
    constexpr size_t kBuckets = (size_t)1 << kDigit;
    std::vector<int64_t> tmp(n);
    int64_t *src = v, *dst = tmp.data();
    for (int shift = score_bits; shift < top; shift += kDigit) {
        size_t count[kBuckets];
        std::memset(count, 0, sizeof count);
        for (size_t i = 0; i < n; ++i) ++count[((uint64_t)src[i] >> shift) & (kBuckets - 1)];
        size_t at = 0;
        for (size_t b = 0; b < kBuckets; ++b) { const size_t c = count[b]; count[b] = at; at += c; }
        for (size_t i = 0; i < n; ++i) dst[count[((uint64_t)src[i] >> shift) & (kBuckets - 1)]++] = src[i];
        std::swap(src, dst);
    }
    if (src != v) std::memcpy(v, src, n * sizeof(int64_t));
}

}  // namespace gfm_hit_sort

// gfm_tsv_internal.hpp -- what the TSV parser (tsv_ingest.cpp) shares with the streamed scan (scan_stream.cpp).
// Part of libgrafimo_hip.so.  Reference lines cited as file:line are relative to /root/reference/src/grafimo/.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "grafimo_hip.h"

// shared thread-local error slot, defined in grafimo_hip.hip
extern "C" void gfm_set_error_(const char *msg);

namespace gfm_tsv_detail {

// columns of one parsed file (row order = line order).  The streamed scan keeps only `names` and `n_rows` here:
// its rows go straight to the pinned chunk slots and the per-chunk column blocks (scan_stream.cpp).
struct FileCols {
    std::vector<uint8_t> kmers;
    std::vector<int64_t> start, stop, freq;
    std::vector<uint8_t> strand, is_ref;
    std::vector<int32_t> local_name;       // index into names
    std::vector<std::string> names;
    int64_t n_rows = -1;                   // kept rows (-1: start.size())
    std::string error;
    int64_t rows() const { return n_rows >= 0 ? n_rows : (int64_t)start.size(); }
};

// score_seqs' row handling (score_sequences.py:273-293, :305-307) for one file
void parse_file(const char *path, int W, bool skip_rev, FileCols &out);

// The bytes of one file, in a buffer that is used again for the next file: small files are read() into it (with
// hundreds of parse threads, mmap/munmap of thousands of region files serialise on the process's address-space
// lock), big ones are mapped.  The streamed scan keeps a few of these per parse thread: a file is read and its rows
// counted in one go, and parsed from the same bytes once its row offset is known (scan_stream.cpp).
class FileBuf {
public:
    FileBuf() = default;
    ~FileBuf() { drop(); }
    FileBuf(const FileBuf &) = delete;
    FileBuf &operator=(const FileBuf &) = delete;
    bool load(const char *path, std::string &err);      // false: err says why
    void drop();                                        // unmap / hand an oversized buffer back
    const char *begin() const { return p_; }
    const char *end() const { return p_ + len_; }

private:
    std::vector<char> buf_;
    const char *p_ = nullptr;
    size_t len_ = 0;
    void *map_ = nullptr;
    size_t map_len_ = 0;
};

// Parse threads worth waking: at most `requested` (<= 0: every hardware thread), one per file, one per MiB of
// text (a thread parses ~1.4 GB/s) and 96 in all (beyond that the threads' own coordination costs more than
// they add: tsv_ingest.cpp).
int pick_threads(const char *const *paths, int n_paths, int requested);

// white space inside a line (what str.split() splits on, the newline aside): one table load per byte -- five
// comparisons per byte were most of the parser's time (210 -> 120 ns per row on one core)
struct WsTable {
    bool t[256];
    constexpr WsTable() : t() { t[(unsigned char)' '] = t[(unsigned char)'\t'] = t[(unsigned char)'\r'] = t[(unsigned char)'\v'] = t[(unsigned char)'\f'] = true; }
};
inline constexpr WsTable kWs{};
inline bool is_ws(char c) { return kWs.t[(unsigned char)c]; }

// CHR:NUM(+|-) -> NUM and strand; the reference takes split(":")[1] and drops its last char
inline bool parse_pos_general(const char *b, const char *e, int64_t *val, char *strand)
{
    if (e - b < 3) return false;
    const char *colon = static_cast<const char *>(memchr(b, ':', (size_t)(e - b)));
    if (!colon) return false;
    const char *p = colon + 1;
    const char *q = static_cast<const char *>(memchr(p, ':', (size_t)(e - p)));
    const char *fe = q ? q : e;      // field after the first ':' (up to a second ':', if any)
    if (fe - p < 2) return false;
    *strand = e[-1];                 // data[2][-1]: last char of the whole column
    const char *ne = fe - 1;         // [:-1]
    bool neg = false;
    if (p < ne && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    if (p >= ne) return false;
    if (ne - p > 18) return false;   // (Python's int() has no limit; coordinates of nineteen digits are refused here)
    int64_t v = 0;
    for (; p < ne; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
    }
    *val = neg ? -v : v;
    return true;
}

// Eight ASCII digits, most significant first, held in a little-endian word -> their value (three multiplies instead of a
// loop of eight dependent multiply-adds; the classic SWAR reduction).
inline uint32_t parse_8_digits(uint64_t w)
{
    w -= 0x3030303030303030ull;
    w = (w * 10) + (w >> 8);                                                  // pairs
    w = (((w & 0x000000FF000000FFull) * (100 + (1000000ull << 32))) +
         (((w >> 16) & 0x000000FF000000FFull) * (1 + (10000ull << 32)))) >> 32;
    return (uint32_t)w;
}
inline bool all_digits_8(uint64_t w)
{
    return (((w + 0x4646464646464646ull) | (w - 0x3030303030303030ull)) & 0x8080808080808080ull) == 0;
}
// n <= 8 digits at p (8 bytes readable) -> value; false: a byte among them is no digit
inline bool parse_upto_8_digits(const char *p, int n, uint64_t *val)
{
    uint64_t w;
    memcpy(&w, p, 8);
    if (n < 8) w = (w << (8 * (8 - n))) | (0x3030303030303030ull >> (8 * n));  // "000ddddd" in address order
    if (!all_digits_8(w)) return false;
    *val = parse_8_digits(w);
    return true;
}

// The same for the shape every vg row has -- name, ':', one to sixteen digits, ONE more character, end of the column --
// with word-wide steps (the byte loops of the general form were a third of the parser's time: 34 of 58 ns per row for
// the two position columns, the count and the name).  `slack`: at least 8 bytes behind `e` are readable.  Anything else
// goes to parse_pos_general.
inline bool parse_pos(const char *b, const char *e, bool slack, int64_t *val, char *strand)
{
    if (!slack || e - b < 3) return parse_pos_general(b, e, val, strand);
    const char *p = b;
    for (;; p += 8) {                                   // the first ':' (a chromosome name is a few characters)
        if (p >= e) return parse_pos_general(b, e, val, strand);
        uint64_t w;
        memcpy(&w, p, 8);
        const uint64_t x = w ^ 0x3A3A3A3A3A3A3A3Aull;
        const uint64_t hit = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
        if (hit) { p += __builtin_ctzll(hit) >> 3; break; }
    }
    if (p >= e) return parse_pos_general(b, e, val, strand);
    ++p;
    const long n = (long)(e - 1 - p);                   // digits between the ':' and the column's last character
    if (n < 1 || n > 16 || e[-1] == ':') return parse_pos_general(b, e, val, strand);
    uint64_t hi = 0, lo = 0;
    if (n <= 8) {
        if (!parse_upto_8_digits(p, (int)n, &lo)) return parse_pos_general(b, e, val, strand);
    } else {
        if (!parse_upto_8_digits(p, (int)(n - 8), &hi) || !parse_upto_8_digits(p + (n - 8), 8, &lo))
            return parse_pos_general(b, e, val, strand);
    }
    *val = (int64_t)(hi * 100000000ull + lo);
    *strand = e[-1];
    return true;
}

// A file's text from a cache-hot buffer into memory nobody reads soon (the scan's arena): non-temporal stores -- no
// read-for-ownership of the destination, and the source stays in cache for the scan that runs beside this.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void copy_streaming_avx2(char *dst, const char *src, size_t n)
{
    size_t i = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u); ++i) dst[i] = src[i];
    for (; i + 32 <= n; i += 32)
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i)));
    for (; i < n; ++i) dst[i] = src[i];
    _mm_sfence();
}
#endif
inline void copy_streaming(char *dst, const char *src, size_t n)
{
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx2");
    if (wide && n >= 4096) { copy_streaming_avx2(dst, src, n); return; }
#endif
    memcpy(dst, src, n);
}

// W bytes (a k-mer) with two overlapping fixed-size moves instead of a library call with a run-time length
inline void copy_kmer(uint8_t *dst, const uint8_t *src, int W)
{
    if (W >= 16 && W <= 32) {
        uint64_t a0, a1, b0, b1;
        memcpy(&a0, src, 8); memcpy(&a1, src + 8, 8);
        memcpy(&b0, src + W - 16, 8); memcpy(&b1, src + W - 8, 8);
        memcpy(dst, &a0, 8); memcpy(dst + 8, &a1, 8);
        memcpy(dst + W - 16, &b0, 8); memcpy(dst + W - 8, &b1, 8);
    } else if (W >= 8 && W < 16) {
        uint64_t a, b2;
        memcpy(&a, src, 8); memcpy(&b2, src + W - 8, 8);
        memcpy(dst, &a, 8); memcpy(dst + W - 8, &b2, 8);
    } else {
        memcpy(dst, src, (size_t)W);
    }
}

inline bool parse_int(const char *b, const char *e, int64_t *val)
{
    if (b >= e) return false;
    bool neg = false;
    if (*b == '-' || *b == '+') { neg = *b == '-'; ++b; }
    if (b >= e || e - b > 18) return false;
    int64_t v = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        v = v * 10 + (*b - '0');
    }
    *val = neg ? -v : v;
    return true;
}

// the first six whitespace-separated fields of the line [p, le); returns how many were found (<= 6)
inline int split_fields_scalar(const char *p, const char *le, const char **fb, const char **fe)
{
    int nf = 0;
    const char *c = p;
    while (c < le && nf < 6) {
        while (c < le && is_ws(*c)) ++c;
        if (c >= le) break;
        fb[nf] = c;
        while (c < le && !is_ws(*c)) ++c;
        fe[nf] = c;
        ++nf;
    }
    return nf;
}

#if defined(__x86_64__)
// The same with AVX2: 64 bytes of the line become one bit mask of its white space (two 32-byte compares), and the
// field boundaries are the 0 -> 1 / 1 -> 0 steps of that mask (138 -> 75 ns per row on one core).  Reads whole 64-byte
// blocks: the caller makes sure that 64 bytes behind `le` are readable.
__attribute__((target("avx2"))) inline unsigned ws_mask32_avx2(__m256i v)
{
    const __m256i ge9 = _mm256_cmpeq_epi8(_mm256_max_epu8(v, _mm256_set1_epi8(9)), v);     // '\t' '\n' '\v' '\f' '\r' are 9..13
    const __m256i le13 = _mm256_cmpeq_epi8(_mm256_min_epu8(v, _mm256_set1_epi8(13)), v);
    const __m256i ctl = _mm256_andnot_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('\n')), _mm256_and_si256(ge9, le13));
    return (unsigned)_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8(' ')), ctl));
}
__attribute__((target("avx2"))) inline unsigned long long ws_mask64_avx2(const char *p)
{
    const unsigned lo = ws_mask32_avx2(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p)));
    const unsigned hi = ws_mask32_avx2(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 32)));
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__attribute__((target("avx2,bmi"))) inline int split_fields_avx2(const char *p, const char *le, const char **fb, const char **fe)
{
    int nf = 0;
    bool in_field = false;
    const long len = (long)(le - p);
    for (long base = 0; base < len && nf < 6; base += 64) {
        unsigned long long ws = ws_mask64_avx2(p + base);
        if (len - base < 64) ws |= ~0ull << (len - base);                   // behind the line: white space
        // steps of the mask: a field starts where a non-ws byte follows ws (or the block start when none is open),
        // and ends at the first ws byte behind it
        unsigned long long prev = (ws << 1) | (in_field ? 0ull : 1ull);
        unsigned long long starts = ~ws & prev, ends = ws & ~prev;
        while (nf < 6) {
            if (!in_field) {
                if (!starts) break;
                const int sbit = __builtin_ctzll(starts);
                starts &= starts - 1;
                fb[nf] = p + base + sbit;
                in_field = true;
                ends &= ~0ull << sbit;                                      // only ends behind this start count
            }
            if (!ends) break;                                               // the field runs on into the next block
            const int ebit = __builtin_ctzll(ends);
            ends &= ends - 1;
            fe[nf++] = p + base + ebit;
            in_field = false;
            starts &= ~0ull << ebit;
        }
    }
    if (in_field && nf < 6) fe[nf++] = le;                                  // the line ends inside a field
    return nf;
}
// Line end AND fields from the same 64-byte blocks (no separate memchr pass over the line): the newline's bit ends the
// line, everything behind it counts as white space.  -> number of fields (<= 6), *line_end = the '\n' (or `end`);
// -1: a block would reach beyond `end` -- the caller takes the scalar way for this line (the file's last lines).
__attribute__((target("avx2,bmi"))) inline int split_line_avx2_blocks(const char *p, const char *end, const char **fb,
                                                                       const char **fe, const char **line_end);
// The common case first -- the six fields end inside the first 128 bytes of the line: both blocks' masks at once, field
// starts and ends as the 0 -> 1 / 1 -> 0 steps of the 128-bit white-space mask, picked off lowest first.  Anything
// else (a longer k-mer, a field across byte 128) goes to the block loop.
__attribute__((target("avx2,bmi"))) inline int split_line_avx2(const char *p, const char *end, const char **fb, const char **fe,
                                                                const char **line_end)
{
    if (end - p < 128) return split_line_avx2_blocks(p, end, fb, fe, line_end);
    const __m256i nlv = _mm256_set1_epi8('\n');
    const __m256i v0 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
    const __m256i v1 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 32));
    const __m256i v2 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 64));
    const __m256i v3 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 96));
    const unsigned long long nl0 = (unsigned long long)(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v0, nlv)) |
                                   ((unsigned long long)(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v1, nlv)) << 32);
    const unsigned long long nl1 = (unsigned long long)(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v2, nlv)) |
                                   ((unsigned long long)(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v3, nlv)) << 32);
    unsigned long long ws0 = (unsigned long long)ws_mask32_avx2(v0) | ((unsigned long long)ws_mask32_avx2(v1) << 32);
    unsigned long long ws1 = (unsigned long long)ws_mask32_avx2(v2) | ((unsigned long long)ws_mask32_avx2(v3) << 32);
    int L = -1;                                          // the newline's offset, if it lies in these 128 bytes
    if (nl0) { L = __builtin_ctzll(nl0); ws0 |= ~0ull << L; ws1 = ~0ull; }
    else if (nl1) { const int t = __builtin_ctzll(nl1); L = 64 + t; ws1 |= ~0ull << t; }
    const unsigned long long prev0 = (ws0 << 1) | 1ull, prev1 = (ws1 << 1) | (ws0 >> 63);
    unsigned long long s0 = ~ws0 & prev0, s1 = ~ws1 & prev1, e0 = ws0 & ~prev0, e1 = ws1 & ~prev1;
    int nf = 0, ne = 0;
    for (; nf < 6; ++nf) {
        if (s0) { fb[nf] = p + __builtin_ctzll(s0); s0 &= s0 - 1; }
        else if (s1) { fb[nf] = p + 64 + __builtin_ctzll(s1); s1 &= s1 - 1; }
        else break;
    }
    for (; ne < nf; ++ne) {
        if (e0) { fe[ne] = p + __builtin_ctzll(e0); e0 &= e0 - 1; }
        else if (e1) { fe[ne] = p + 64 + __builtin_ctzll(e1); e1 &= e1 - 1; }
        else break;
    }
    if (ne < nf) return split_line_avx2_blocks(p, end, fb, fe, line_end);     // a field runs on beyond byte 128
    if (L >= 0) { *line_end = p + L; return nf; }
    if (nf < 6) return split_line_avx2_blocks(p, end, fb, fe, line_end);      // a long line with fields still to come
    const char *q = static_cast<const char *>(memchr(p + 128, '\n', (size_t)(end - (p + 128))));
    *line_end = q ? q : end;
    return nf;
}
__attribute__((target("avx2,bmi"))) inline int split_line_avx2_blocks(const char *p, const char *end, const char **fb,
                                                                       const char **fe, const char **line_end)
{
    int nf = 0;
    bool in_field = false;
    for (long base = 0;; base += 64) {
        if (end - (p + base) < 64) return -1;
        const __m256i v0 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + base));
        const __m256i v1 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + base + 32));
        const __m256i nlv = _mm256_set1_epi8('\n');
        const unsigned long long nl = (unsigned long long)(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v0, nlv)) |
                                      ((unsigned long long)(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v1, nlv)) << 32);
        unsigned long long ws = (unsigned long long)ws_mask32_avx2(v0) | ((unsigned long long)ws_mask32_avx2(v1) << 32);
        int nl_bit = -1;
        if (nl) {
            nl_bit = __builtin_ctzll(nl);
            ws |= ~0ull << nl_bit;                                            // the newline and what follows it
        }
        if (nf < 6) {
            unsigned long long prev = (ws << 1) | (in_field ? 0ull : 1ull);
            unsigned long long starts = ~ws & prev, ends = ws & ~prev;
            while (nf < 6) {
                if (!in_field) {
                    if (!starts) break;
                    const int sbit = __builtin_ctzll(starts);
                    starts &= starts - 1;
                    fb[nf] = p + base + sbit;
                    in_field = true;
                    ends &= ~0ull << sbit;
                }
                if (!ends) break;
                const int ebit = __builtin_ctzll(ends);
                ends &= ends - 1;
                fe[nf++] = p + base + ebit;
                in_field = false;
                starts &= ~0ull << ebit;
            }
        }
        if (nl_bit >= 0) {
            *line_end = p + base + nl_bit;
            return nf;
        }
        if (nf == 6) {                  // the rest of a long line (its node path) holds nothing we read
            const char *q = static_cast<const char *>(memchr(p + base + 64, '\n', (size_t)(end - (p + base + 64))));
            *line_end = q ? q : end;
            return nf;
        }
    }
}
inline bool cpu_has_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi");
    return yes;
}
inline bool cpu_has_avx512()
{
    // (GRAFIMO_SCAN_NO_AVX512: the AVX2 / scalar forms on a CPU that has AVX-512 -- the sanitizer run covers both)
    static const bool yes = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                            __builtin_cpu_supports("bmi") && __builtin_cpu_supports("popcnt") &&
                            !std::getenv("GRAFIMO_SCAN_NO_AVX512");
    return yes;
}
// AVX-512: 64 bytes -> the bit masks of their newlines and of their white space (is_ws: ' ' and 9..13 without '\n') in
// four instructions -- the compares write mask registers, no movemask / shift / or as with two 32-byte halves.
__attribute__((target("avx512f,avx512bw"))) inline void masks64_avx512(const char *p, unsigned long long *nl,
                                                                       unsigned long long *ws)
{
    const __m512i v = _mm512_loadu_si512(reinterpret_cast<const void *>(p));
    const __mmask64 n = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('\n'));
    const __mmask64 ctl = _mm512_cmplt_epu8_mask(_mm512_sub_epi8(v, _mm512_set1_epi8(9)), _mm512_set1_epi8(5));
    const __mmask64 sp = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8(' '));
    *nl = (unsigned long long)n;
    *ws = ((unsigned long long)ctl & ~(unsigned long long)n) | (unsigned long long)sp;
}
// What the streamed scan needs of one line before hits are known (scan_rows): the k-mer column's bounds, the last
// character of the third column (the strand), that six columns exist, and where the line ends -- from the white-space
// mask of the line's first 128 bytes.  At least 128 bytes must be readable at p.
//   1: a row (kb, ke, strand, line_end set)   0: a blank line (line_end set)
//  -1: not decided here (fewer than six columns in sight, a blank stretch of 128 bytes): the caller takes the general way
__attribute__((target("avx512f,avx512bw,bmi,popcnt"))) inline int light_line_avx512(const char *p, const char *end, const char **kb,
                                                                                   const char **ke, char *strand,
                                                                                   const char **line_end)
{
    unsigned long long nl0, nl1, ws0, ws1;
    masks64_avx512(p, &nl0, &ws0);
    masks64_avx512(p + 64, &nl1, &ws1);
    int L = -1;                                          // the newline's offset, if it lies in these 128 bytes
    if (nl0) { L = __builtin_ctzll(nl0); ws0 |= ~0ull << L; ws1 = ~0ull; }
    else if (nl1) { const int t = __builtin_ctzll(nl1); L = 64 + t; ws1 |= ~0ull << t; }
    const unsigned long long prev0 = (ws0 << 1) | 1ull, prev1 = (ws1 << 1) | (ws0 >> 63);
    unsigned long long s0 = ~ws0 & prev0, s1 = ~ws1 & prev1, e0 = ws0 & ~prev0, e1 = ws1 & ~prev1;
    const int ns = __builtin_popcountll(s0) + __builtin_popcountll(s1);
    if (ns == 0) {
        if (L < 0) return -1;
        *line_end = p + L;
        return 0;
    }
    // six starts in sight mean five complete columns and a sixth that has begun (its end is not needed)
    if (ns < 6) return -1;
    auto take = [](unsigned long long &lo, unsigned long long &hi) {
        if (lo) { const int b = __builtin_ctzll(lo); lo &= lo - 1; return b; }
        const int b = 64 + __builtin_ctzll(hi);
        hi &= hi - 1;
        return b;
    };
    (void)take(s0, s1);
    *kb = p + take(s0, s1);
    (void)take(e0, e1);
    *ke = p + take(e0, e1);
    *strand = p[take(e0, e1) - 1];
    if (L >= 0) {
        *line_end = p + L;
    } else {
        const char *q = static_cast<const char *>(memchr(p + 128, '\n', (size_t)(end - (p + 128))));
        *line_end = q ? q : end;
    }
    return 1;
}
#endif

// `readable_end`: bytes up to there may be read (the line itself ends at le <= readable_end)
inline int split_fields(const char *p, const char *le, const char *readable_end, const char **fb, const char **fe)
{
#if defined(__x86_64__)
    if (readable_end - le >= 64 && cpu_has_avx2()) return split_fields_avx2(p, le, fb, fe);
#endif
    return split_fields_scalar(p, le, fb, fe);
}

// Rows the parser will keep: lines that hold a field, minus -- with skip_rev -- those whose third column ends in
// '-' (score_sequences.py:281-282).  Exact whenever parse_rows() succeeds on the same text.
int64_t count_rows(const char *p, const char *end, bool skip_rev);

// REGION strings of one file -> small integer ids (consecutive rows usually repeat the name)
struct NameTable {
    std::vector<std::string> &names;
    std::unordered_map<std::string, int32_t> ix;
    int32_t last_id = -1;
    explicit NameTable(std::vector<std::string> &n) : names(n) {}
    int32_t id(const char *b, size_t len)
    {
        if (last_id >= 0) {                            // consecutive rows repeat the name: eight bytes at a time, no call.
            const std::string &ls = names[(size_t)last_id];   // (Against the table's OWN copy of it: the caller's bytes may be a
            if (ls.size() == len) {                    // buffer that has been read into again since.)
                const char *last = ls.data();
                size_t i = 0;
                bool same = true;
                for (; same && i + 8 <= len; i += 8) {
                    uint64_t x, y;
                    memcpy(&x, last + i, 8);
                    memcpy(&y, b + i, 8);
                    same = x == y;
                }
                for (; same && i < len; ++i) same = last[i] == b[i];
                if (same) return last_id;
            }
        }
        std::string key(b, len);
        auto it = ix.find(key);
        int32_t nid;
        if (it == ix.end()) {
            nid = (int32_t)names.size();
            names.push_back(key);
            ix.emplace(std::move(key), nid);
        } else {
            nid = it->second;
        }
        last_id = nid;
        return nid;
    }
};

// score_seqs' row handling (score_sequences.py:273-293, :305-307) over the text [p, end): every kept row goes to
// sink(kmer bytes [W], start, stop, count, strand char, is_ref, local name id).  false: `error` holds
// "path:line: what".
template <class Sink>
bool parse_rows(const char *path, const char *p, const char *end, int W, bool skip_rev, NameTable &names, Sink &&sink,
                std::string &error)
{
    int64_t lineno = 0;
#if defined(__x86_64__)
    const bool wide = cpu_has_avx2();
#endif
    while (p < end) {
        ++lineno;
        const char *fb[6], *fe[6];
        const char *le = nullptr;
        int nf = -1;
#if defined(__x86_64__)
        if (wide) nf = split_line_avx2(p, end, fb, fe, &le);
#endif
        if (nf < 0) {
            const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
            le = nl ? nl : end;
            nf = split_fields(p, le, end, fb, fe);
        }
        const char *next = le < end ? le + 1 : end;
        if (nf == 0) { p = next; continue; }  // blank line
        auto bad = [&](const char *what) {
            char buf[256];
            snprintf(buf, sizeof buf, "%s:%lld: %s", path, (long long)lineno, what);
            error = buf;
            return false;
        };
        if (nf < 6) return bad("expected at least 6 columns");
        int64_t st = 0, sp = 0, fr = 0;
        char s1 = 0, s2 = 0;
        const bool slack = end - le >= 64;             // (the condition under which split_fields read whole blocks, too)
        if (!parse_pos(fb[2], fe[2], slack, &st, &s1)) return bad("malformed start column");
        if (skip_rev && s1 == '-') { p = next; continue; }
        if (!parse_pos(fb[3], fe[3], slack, &sp, &s2)) return bad("malformed stop column");
        if (fe[1] - fb[1] != W) return bad("k-mer length differs from the motif width");
        if (!parse_int(fb[4], fe[4], &fr)) return bad("malformed haplotype count");
        const int32_t nid = names.id(fb[0], (size_t)(fe[0] - fb[0]));
        const bool is_ref_str = (fe[5] - fb[5] == 3) && memcmp(fb[5], "ref", 3) == 0;
        const int64_t dist = sp > st ? sp - st : st - sp;
        sink(reinterpret_cast<const uint8_t *>(fb[1]), st, sp, fr, (uint8_t)s1,
             (uint8_t)(is_ref_str && dist == W),   // score_sequences.py:305-307
             nid);
        p = next;
    }
    return true;
}

#if defined(__x86_64__)
// scan_rows() with AVX-512, block by block.  The line-at-a-time form above is a chain of dependent steps -- load, compare,
// find the newline, only then is the next line's address known: ~30 cycles a line whatever the instruction count.  Here
// the masks (newlines, white space) of 16 KiB of text are made first, 64 bytes per step at addresses that depend on
// nothing, and the lines are then walked in MASK space: the next line start is a bit scan over words in L1, and a line's
// field boundaries are the steps of a 128-bit slice of the white-space mask.  16 -> 6 ns per row on one core.
// A line that does not show six columns inside its first 128 bytes (malformed, or longer names than any vg writes) is
// handed to the general splitter, which also words the error.  No byte behind `end` is read (masked load of the last block).
template <class Sink>
__attribute__((target("avx512f,avx512bw,bmi,popcnt"))) bool scan_rows_avx512(const char *path, const char *base, const char *end,
                                                                            int W, bool skip_rev, Sink &&sink, std::string &error)
{
    constexpr size_t kChunk = 256;                       // blocks of 64 bytes whose lines are walked per round
    const size_t n = (size_t)(end - base);
    const size_t last_block = n / 64;                    // holds the text's last bytes (if any) and the closing newline
    unsigned long long nlm[kChunk + 3], wsm[kChunk + 3];
    int64_t lineno = 0;
    auto bad = [&](const char *what) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s:%lld: %s", path, (long long)lineno, what);
        error = buf;
        return false;
    };
    // the general way for one line [o, q): what scan_rows does per line
    auto general = [&](size_t o, size_t q) -> bool {
        const char *p = base + o, *le = base + q;
        const char *fb[6], *fe[6];
        const int nf = split_fields(p, le, end, fb, fe);
        if (nf == 0) return true;
        if (nf < 6) return bad("expected at least 6 columns");
        if (!(skip_rev && fe[2][-1] == '-')) {
            if (fe[1] - fb[1] != W) return bad("k-mer length differs from the motif width");
            sink(reinterpret_cast<const uint8_t *>(fb[1]), (uint64_t)o);
        }
        return true;
    };
    size_t o = 0;                                        // start of the line being looked for
    while (o < n) {
        const size_t c0 = o & ~(size_t)63, g0 = c0 / 64;
        for (size_t b = 0; b < kChunk + 2; ++b) {
            const size_t g = g0 + b;
            if (g < last_block) {
                masks64_avx512(base + g * 64, &nlm[b], &wsm[b]);
            } else if (g == last_block) {                // the text's tail, a newline behind it, white space from there on
                const unsigned rem = (unsigned)(n % 64);
                const __mmask64 k = rem ? (~0ull >> (64 - rem)) : 0ull;
                const __m512i v = _mm512_maskz_loadu_epi8(k, reinterpret_cast<const void *>(base + g * 64));
                const unsigned long long nl = (unsigned long long)_mm512_mask_cmpeq_epi8_mask(k, v, _mm512_set1_epi8('\n'));
                const unsigned long long ctl = (unsigned long long)_mm512_mask_cmplt_epu8_mask(
                    k, _mm512_sub_epi8(v, _mm512_set1_epi8(9)), _mm512_set1_epi8(5));
                const unsigned long long sp = (unsigned long long)_mm512_mask_cmpeq_epi8_mask(k, v, _mm512_set1_epi8(' '));
                nlm[b] = nl | (1ull << rem);
                wsm[b] = ((ctl & ~nl) | sp) | (~0ull << rem);
            } else {
                nlm[b] = 0ull;
                wsm[b] = ~0ull;
            }
        }
        nlm[kChunk + 2] = 0ull;
        wsm[kChunk + 2] = ~0ull;
        const size_t o_in = o;
        // every line that ENDS inside the first kChunk blocks: its 128-bit slice lies inside the kChunk + 2 blocks made
        bool open_end = false;                           // the walk stopped at the text's closing newline
        for (size_t wi = (o - c0) >> 6; wi < kChunk && !open_end; ++wi) {
            unsigned long long m = nlm[wi];
            if (wi == ((o - c0) >> 6)) m &= ~0ull << ((o - c0) & 63);
            while (m) {
                const size_t q = c0 + wi * 64 + (size_t)__builtin_ctzll(m);      // the line is [o, q)
                m &= m - 1;
                ++lineno;
                const size_t rel = o - c0, w = rel >> 6;
                const unsigned sh = (unsigned)(rel & 63);
                unsigned long long w0 = wsm[w], w1 = wsm[w + 1];
                if (sh) {
                    w0 = (w0 >> sh) | (w1 << (64 - sh));
                    w1 = (w1 >> sh) | (wsm[w + 2] << (64 - sh));
                }
                const size_t L = q - o;
                if (L < 64) { w0 |= ~0ull << L; w1 = ~0ull; }
                else if (L < 128) { w1 |= ~0ull << (L - 64); }
                const unsigned long long prev0 = (w0 << 1) | 1ull, prev1 = (w1 << 1) | (w0 >> 63);
                unsigned long long s0 = ~w0 & prev0, s1 = ~w1 & prev1, e0 = w0 & ~prev0, e1 = w1 & ~prev1;
                const int ns = __builtin_popcountll(s0) + __builtin_popcountll(s1);
                if (ns >= 6) {
                    auto take = [](unsigned long long &lo, unsigned long long &hi) {
                        if (lo) { const int b = __builtin_ctzll(lo); lo &= lo - 1; return b; }
                        const int b = 64 + __builtin_ctzll(hi);
                        hi &= hi - 1;
                        return b;
                    };
                    (void)take(s0, s1);
                    const int kb = take(s0, s1);
                    (void)take(e0, e1);
                    const int ke = take(e0, e1);
                    const char sd = base[o + (size_t)take(e0, e1) - 1];
                    if (!(skip_rev && sd == '-')) {
                        if (ke - kb != W) return bad("k-mer length differs from the motif width");
                        sink(reinterpret_cast<const uint8_t *>(base + o + (size_t)kb), (uint64_t)o);
                    }
                } else if (!(ns == 0 && L < 128)) {      // (a blank line is nothing)
                    if (!general(o, q)) return false;
                }
                o = q + 1;
                if (q >= n) { open_end = true; break; }
            }
        }
        if (o >= n) break;
        if (o == o_in) {
            // no line ended inside this round: a line of more than 16 KiB.  Its end, then the general way.
            const char *nl = static_cast<const char *>(memchr(base + o, '\n', n - o));
            const size_t q = nl ? (size_t)(nl - base) : n;
            ++lineno;
            if (!general(o, q)) return false;
            o = q + 1;
        }
    }
    return true;
}
#endif

// The streamed scan's pass over a file: what must be known of EVERY row before the hits are -- its k-mer (the score
// kernel's input, score_sequences.py:286) and the strand character, the last one of the third column (:280-282: '-' rows
// are skipped under --no-reverse before they are scored or counted) -- plus where its line starts, so that the other
// columns can be read for the few rows that turn out to be hits (parse_line, at gfm_scan_tsv_finish).  The numbers of a
// row are NOT converted here: that was 20 of the parser's 36 ns per row, for values that one row in ten thousand needs.
// Checked for every row: six columns, a k-mer of the motif's width.  (A number that does not parse is therefore reported
// only if its row is a hit; the reference's int() raises for any row.  vg does not write such rows.)
// sink(kmer bytes [W], offset of the line in the file).  false: `error` holds "path:line: what".
template <class Sink>
bool scan_rows(const char *path, const char *base, const char *end, int W, bool skip_rev, Sink &&sink, std::string &error)
{
    int64_t lineno = 0;
    const char *p = base;
#if defined(__x86_64__)
    if (cpu_has_avx512())
        return scan_rows_avx512(path, base, end, W, skip_rev, sink, error);
    const bool wide = cpu_has_avx2();
#endif
    while (p < end) {
        ++lineno;
        const char *fb[6], *fe[6];
        const char *le = nullptr;
        int nf = -1;
#if defined(__x86_64__)
        if (wide) nf = split_line_avx2(p, end, fb, fe, &le);
#endif
        if (nf < 0) {
            const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
            le = nl ? nl : end;
            nf = split_fields(p, le, end, fb, fe);
        }
        const char *next = le < end ? le + 1 : end;
        if (nf == 0) { p = next; continue; }  // blank line
        auto bad = [&](const char *what) {
            char buf[256];
            snprintf(buf, sizeof buf, "%s:%lld: %s", path, (long long)lineno, what);
            error = buf;
            return false;
        };
        if (nf < 6) return bad("expected at least 6 columns");
        if (!(skip_rev && fe[2][-1] == '-')) {
            if (fe[1] - fb[1] != W) return bad("k-mer length differs from the motif width");
            sink(reinterpret_cast<const uint8_t *>(fb[1]), (uint64_t)(p - base));
        }
        p = next;
    }
    return true;
}

// The columns of ONE row, from the bytes at its line start (`readable_end`: what may be read; the line may end earlier):
// what parse_rows hands its sink for that row.  -1: the first six columns do not end inside the bytes given (the
// caller reads more); 0: malformed, *what says why; 1: fine.
struct LineCols {
    int64_t start = 0, stop = 0, freq = 0;
    uint8_t strand = 0, is_ref = 0;
    const char *name = nullptr, *kmer = nullptr;
    size_t name_len = 0, kmer_len = 0;
};
inline int parse_line(const char *p, const char *readable_end, bool complete, int W, LineCols &out, const char **what)
{
    const char *nl = nullptr, *le = nullptr;
    const char *fb[6], *fe[6];
    int nf = -1;
#if defined(__x86_64__)
    // (a line of the kept text: 128 readable bytes nearly always -- the masks of the streamed scan's splitter instead of a
    // byte loop over the line, 140 -> 15 ns of the ~450 a hit row's columns cost)
    if (readable_end - p >= 128 && cpu_has_avx2()) {
        nf = split_line_avx2(p, readable_end, fb, fe, &le);
        if (nf >= 0) nl = le < readable_end ? le : nullptr;
    }
#endif
    if (nf < 0) {
        nl = static_cast<const char *>(memchr(p, '\n', (size_t)(readable_end - p)));
        le = nl ? nl : readable_end;
        nf = split_fields_scalar(p, le, fb, fe);
    }
    // without the line's end in sight the sixth column may go on behind the bytes we hold
    if (!nl && !complete && (nf < 6 || fe[5] == le)) return -1;
    if (nf < 6) { *what = "expected at least 6 columns"; return 0; }
    char s1 = 0, s2 = 0;
    if (!parse_pos(fb[2], fe[2], false, &out.start, &s1)) { *what = "malformed start column"; return 0; }
    if (!parse_pos(fb[3], fe[3], false, &out.stop, &s2)) { *what = "malformed stop column"; return 0; }
    if (fe[1] - fb[1] != W) { *what = "k-mer length differs from the motif width"; return 0; }
    if (!parse_int(fb[4], fe[4], &out.freq)) { *what = "malformed haplotype count"; return 0; }
    const bool is_ref_str = (fe[5] - fb[5] == 3) && memcmp(fb[5], "ref", 3) == 0;
    const int64_t dist = out.stop > out.start ? out.stop - out.start : out.start - out.stop;
    out.strand = (uint8_t)s1;
    out.is_ref = (uint8_t)(is_ref_str && dist == W);      // score_sequences.py:305-307
    out.name = fb[0];
    out.name_len = (size_t)(fe[0] - fb[0]);
    out.kmer = fb[1];
    out.kmer_len = (size_t)(fe[1] - fb[1]);
    return 1;
}

}  // namespace gfm_tsv_detail

struct gfm_tsv {
    int W = 0;
    int64_t n = 0;
    std::vector<gfm_tsv_detail::FileCols> files;
    std::vector<int64_t> row_base;          // per file
    std::vector<std::string> names;         // global distinct REGION strings
    std::vector<std::vector<int32_t>> remap;  // per file: local name id -> global
    // row bases and the global name table from the parsed files (all of them free of errors)
    void index_rows();
};

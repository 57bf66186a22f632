// gfm_tsv_internal.hpp -- what the TSV parser (tsv_ingest.cpp) shares with the streamed scan (scan_stream.cpp).
// Part of libgrafimo_hip.so.  Reference lines cited as file:line are relative to /root/reference/src/grafimo/.
#pragma once

#include <cstdint>
#include <cstdio>
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
inline bool parse_pos(const char *b, const char *e, int64_t *val, char *strand)
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
    int64_t v = 0;
    for (; p < ne; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
    }
    *val = neg ? -v : v;
    return true;
}

inline bool parse_int(const char *b, const char *e, int64_t *val)
{
    if (b >= e) return false;
    bool neg = false;
    if (*b == '-' || *b == '+') { neg = *b == '-'; ++b; }
    if (b >= e) return false;
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
inline bool cpu_has_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi");
    return yes;
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
    const char *last = nullptr;
    size_t last_len = 0;
    int32_t last_id = -1;
    explicit NameTable(std::vector<std::string> &n) : names(n) {}
    int32_t id(const char *b, size_t len)
    {
        if (last && len == last_len && memcmp(last, b, len) == 0) return last_id;
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
        last = b; last_len = len; last_id = nid;
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
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        const char *le = nl ? nl : end;
        ++lineno;
        const char *fb[6], *fe[6];
        const int nf = split_fields(p, le, end, fb, fe);
        const char *next = nl ? nl + 1 : end;
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
        if (!parse_pos(fb[2], fe[2], &st, &s1)) return bad("malformed start column");
        if (skip_rev && s1 == '-') { p = next; continue; }
        if (!parse_pos(fb[3], fe[3], &sp, &s2)) return bad("malformed stop column");
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

// ASan/UBSan harness for the host-side readers (no GPU): TSV ingest (table + counting pass) and the VCF reader on the
// fixtures, on synthetic files and on mutated copies of them.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "grafimo_hip.h"
#include "gfm_tsv_internal.hpp"
#include "gfm_hit_sort.hpp"
#include <algorithm>
static thread_local std::string g_err;
extern "C" void gfm_set_error_(const char *m) { g_err = m ? m : ""; }
static std::string slurp(const char *p) { std::ifstream f(p, std::ios::binary); return std::string((std::istreambuf_iterator<char>(f)), {}); }
// The row rules with the library's plain byte loops only (memchr, split_fields_scalar, parse_pos_general, parse_int):
// what the word-wide / AVX2 fast paths of parse_rows must reproduce, value for value, also on malformed text.
struct RefRows {
    bool ok = true;
    std::vector<uint8_t> kmers, strand, is_ref;
    std::vector<int64_t> start, stop, freq;
};
static RefRows ref_parse(const std::string &t, int W, bool skip_rev)
{
    using namespace gfm_tsv_detail;
    RefRows r;
    const char *p = t.data(), *end = p + t.size();
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        const char *le = nl ? nl : end;
        const char *fb[6], *fe[6];
        const int nf = split_fields_scalar(p, le, fb, fe);
        p = nl ? nl + 1 : end;
        if (nf == 0) continue;
        int64_t st = 0, sp = 0, fr = 0;
        char s1 = 0, s2 = 0;
        if (nf < 6 || !parse_pos_general(fb[2], fe[2], &st, &s1)) { r.ok = false; return r; }
        if (skip_rev && s1 == '-') continue;
        if (!parse_pos_general(fb[3], fe[3], &sp, &s2) || fe[1] - fb[1] != W || !parse_int(fb[4], fe[4], &fr)) { r.ok = false; return r; }
        r.kmers.insert(r.kmers.end(), fb[1], fe[1]);
        r.start.push_back(st); r.stop.push_back(sp); r.freq.push_back(fr); r.strand.push_back((uint8_t)s1);
        const int64_t dist = sp > st ? sp - st : st - sp;
        r.is_ref.push_back((uint8_t)((fe[5] - fb[5] == 3) && memcmp(fb[5], "ref", 3) == 0 && dist == W));
    }
    return r;
}

// The streamed scan's pass (scan_rows: k-mer + line offset of every kept row, six columns and the k-mer length checked,
// no numbers converted) with byte loops only, and parse_line for single rows against the full parse of that line.
struct LightRows {
    bool ok = true;
    std::vector<uint64_t> off;
    std::vector<uint8_t> kmers;
};
static LightRows ref_light(const std::string &t, int W, bool skip_rev)
{
    using namespace gfm_tsv_detail;
    LightRows r;
    const char *p = t.data(), *end = p + t.size();
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        const char *le = nl ? nl : end;
        const char *fb[6], *fe[6];
        const int nf = split_fields_scalar(p, le, fb, fe);
        const char *line = p;
        p = nl ? nl + 1 : end;
        if (nf == 0) continue;
        if (nf < 6) { r.ok = false; return r; }
        if (skip_rev && fe[2][-1] == '-') continue;
        if (fe[1] - fb[1] != W) { r.ok = false; return r; }
        r.off.push_back((uint64_t)(line - t.data()));
        r.kmers.insert(r.kmers.end(), fb[1], fe[1]);
    }
    return r;
}
static int check_light(const std::string &t, int it, int skip, std::mt19937 &rng)
{
    using namespace gfm_tsv_detail;
    const LightRows ref = ref_light(t, 19, skip != 0);
    LightRows got;
    std::string err;
    std::string padded = t + std::string(64, '\0');           // (FileBuf keeps 64 readable bytes behind the text)
    got.ok = scan_rows("m.tsv", padded.data(), padded.data() + t.size(), 19, skip != 0,
                       [&](const uint8_t *k, uint64_t off) { got.off.push_back(off); got.kmers.insert(got.kmers.end(), k, k + 19); }, err);
    if (got.ok != ref.ok) { std::printf("LIGHT ACCEPT MISMATCH it=%d skip=%d scan_rows=%d reference=%d\n", it, skip, (int)got.ok, (int)ref.ok); return 1; }
    if (!ref.ok) return 0;
    if (got.off != ref.off || got.kmers != ref.kmers) { std::printf("LIGHT VALUE MISMATCH it=%d skip=%d\n", it, skip); return 1; }
    if (count_rows(padded.data(), padded.data() + t.size(), skip != 0) != (int64_t)ref.off.size()) {
        std::printf("LIGHT COUNT MISMATCH it=%d skip=%d\n", it, skip);
        return 1;
    }
    for (int k = 0; k < 12 && !ref.off.empty(); ++k) {          // single rows: whole line, and a prefix of it as a pread would hold
        const uint64_t off = ref.off[rng() % ref.off.size()];
        const char *p = t.data() + off, *end = t.data() + t.size();
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        const std::string line(p, nl ? nl : end);
        const RefRows one = ref_parse(line, 19, false);
        LineCols c;
        const char *what = "";
        const int rc = parse_line(p, end, true, 19, c, &what);
        if ((rc == 1) != (one.ok && one.start.size() == 1)) { std::printf("LINE ACCEPT MISMATCH it=%d off=%llu rc=%d\n", it, (unsigned long long)off, rc); return 1; }
        if (rc == 1 && (c.start != one.start[0] || c.stop != one.stop[0] || c.freq != one.freq[0] || c.strand != one.strand[0] ||
                        c.is_ref != one.is_ref[0] || memcmp(c.kmer, one.kmers.data(), 19) != 0)) {
            std::printf("LINE VALUE MISMATCH it=%d off=%llu\n", it, (unsigned long long)off);
            return 1;
        }
        const size_t cut = 1 + rng() % (size_t)(end - p);
        std::string part(p, cut);
        LineCols d;
        const int rc2 = parse_line(part.data(), part.data() + part.size(), cut == (size_t)(end - p), 19, d, &what);
        if (rc2 >= 0 && rc2 != rc) { std::printf("LINE PREFIX MISMATCH it=%d off=%llu cut=%zu rc=%d full=%d\n", it, (unsigned long long)off, cut, rc2, rc); return 1; }
        if (rc2 == 1 && (d.start != c.start || d.stop != c.stop || d.freq != c.freq || d.is_ref != c.is_ref || d.name_len != c.name_len)) {
            std::printf("LINE PREFIX VALUE MISMATCH it=%d off=%llu cut=%zu\n", it, (unsigned long long)off, cut);
            return 1;
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    const char *tsv = argv[1], *vcf = argv[2], *tmp = argv[3];
    std::string base = slurp(tsv);
    std::mt19937 rng(7);
    // a second base text: region names of 1..150 characters, so that every field boundary falls on either side of bytes
    // 64 and 128 of its line (the splitter works on 64-byte blocks and has a fast path for lines that fit two of them)
    std::string base2;
    for (int i = 0; i < 400; ++i) {
        std::string name(1 + rng() % 150, 'r');
        for (auto &c : name) c = "chr0123456789:-_"[rng() % 16];
        std::string kmer(19, 'A');
        for (auto &c : kmer) c = "ACGTN"[rng() % 5];
        const long a = (long)(rng() % 100000000), b = a + 19;
        const bool minus = rng() % 2;
        base2 += name + "\t" + kmer + "\tchr22:" + std::to_string(minus ? b : a) + (minus ? "-" : "+") + "\tchr22:" +
                 std::to_string(minus ? a : b) + (minus ? "-" : "+") + "\t" + std::to_string(rng() % 5097) + "\t" +
                 (rng() % 2 ? "ref" : "non.ref") + "\t" + std::string(rng() % 90, '7') + ",\n";
    }
    long ok = 0, bad = 0;
    for (int it = 0; it < 2000; ++it) {
        std::string t = it < 1500 ? base : base2;
        if (it == 1500) { /* unmutated */ } else
        if (it) {
            const int muts = 1 + (int)(rng() % (it < 1500 ? 8 : 3));
            for (int k = 0; k < muts; ++k) {
                const size_t at = rng() % t.size();
                switch (rng() % 6) {
                case 0: t[at] = "\t \n:+-0123456789ACGTNx\r"[rng() % 22]; break;
                case 5: t.insert(at, std::string(1 + rng() % 17, "0123456789"[rng() % 10])); break;    // long numbers
                case 1: t.erase(at, 1 + rng() % 40); break;
                case 2: t.insert(at, std::string(1 + rng() % 5, "\t \n"[rng() % 3])); break;
                case 3: t.resize(at); break;
                default: t.insert(at, t.substr(rng() % t.size(), rng() % 200)); break;
                }
                if (t.empty()) t = "x";
            }
        }
        std::string path = std::string(tmp) + "/m.tsv";
        { std::ofstream o(path, std::ios::binary); o << t; }
        const char *paths[1] = {path.c_str()};
        for (int skip = 0; skip < 2; ++skip) {
            gfm_tsv_t h = nullptr; int64_t n = 0, c = -1;
            const int rc = gfm_tsv_open(paths, 1, 19, skip, 1 + it % 3, &h, &n);
            const int rc2 = gfm_tsv_count_rows(path.c_str(), skip, &c);
            if (check_light(t, it, skip, rng)) return 1;
            const RefRows ref = ref_parse(t, 19, skip != 0);
            if ((rc == 0) != ref.ok) { std::printf("ACCEPT MISMATCH it=%d skip=%d library rc=%d reference ok=%d\n", it, skip, rc, (int)ref.ok); return 1; }
            if (rc == 0) {
                ++ok;
                if (rc2 != 0 || c != n) { std::printf("COUNT MISMATCH it=%d skip=%d parsed=%lld counted=%lld\n", it, skip, (long long)n, (long long)c); return 1; }
                std::vector<uint8_t> km((size_t)n * 19 + 1); std::vector<int64_t> a(n + 1), b(n + 1), f(n + 1); std::vector<uint8_t> s(n + 1), r(n + 1); std::vector<int32_t> fi(n + 1), ni(n + 1);
                gfm_tsv_read(h, km.data(), a.data(), b.data(), s.data(), f.data(), r.data(), fi.data(), ni.data());
                gfm_tsv_close(h);
                km.resize((size_t)n * 19); a.resize(n); b.resize(n); f.resize(n); s.resize(n); r.resize(n);
                if ((int64_t)ref.start.size() != n || km != ref.kmers || a != ref.start || b != ref.stop || f != ref.freq ||
                    s != ref.strand || r != ref.is_ref) {
                    std::printf("VALUE MISMATCH it=%d skip=%d rows %lld / %zu\n", it, skip, (long long)n, ref.start.size());
                    return 1;
                }
            } else ++bad;
        }
    }
    std::printf("tsv: %ld parsed, %ld refused\n", ok, bad);
    // the block-wise scanner's edges (scan_rows_avx512, where the CPU has AVX-512; the line-wise form elsewhere): a text whose
    // size is a multiple of 64, one byte more, one byte less; no newline at the end; blank stretches longer than the 128-byte
    // slice; a line longer than one round of blocks; a buffer with NO readable byte behind its end
    {
        using namespace gfm_tsv_detail;
        const std::string row = "chr22:16000000-16000200\tGAAAATTATTGATATGTAT\tchr22:16000000+\tchr22:16000019+\t984\tref\t1+,\n";
        std::vector<std::string> cases;
        for (int pad = 0; pad < 130; ++pad) {
            std::string t;
            for (int r = 0; r < 40; ++r) t += row;
            t += std::string((size_t)pad, ' ');
            cases.push_back(t);
            cases.push_back(t + row.substr(0, row.size() - 1));          // no closing newline
            cases.push_back(std::string((size_t)pad, '\n') + t);
        }
        cases.push_back(row + std::string(300, ' ') + "\n" + row + std::string(200, '\t') + row);
        cases.push_back(row.substr(0, row.size() - 1) + std::string(20000, '7') + "\n" + row + row);     // a 20 KB node path
        cases.push_back(std::string(40000, 'A') + "\n" + row);                                             // a 40 KB first field: refused
        cases.push_back(std::string(17000, ' ') + row + row);
        cases.push_back("");
        cases.push_back("\n");
        cases.push_back(row.substr(0, 60));
        for (size_t ci = 0; ci < cases.size(); ++ci) {
            const std::string &t = cases[ci];
            for (int skip = 0; skip < 2; ++skip) {
                const LightRows ref = ref_light(t, 19, skip != 0);
                std::vector<char> exact(t.begin(), t.end());           // heap block of exactly the text's size
                LightRows got;
                std::string err;
                got.ok = scan_rows("m.tsv", exact.data(), exact.data() + exact.size(), 19, skip != 0,
                                   [&](const uint8_t *k, uint64_t off) { got.off.push_back(off); got.kmers.insert(got.kmers.end(), k, k + 19); }, err);
                if (got.ok != ref.ok || (ref.ok && (got.off != ref.off || got.kmers != ref.kmers))) {
                    std::printf("EDGE MISMATCH case=%zu skip=%d ok %d/%d rows %zu/%zu\n", ci, skip, (int)got.ok, (int)ref.ok, got.off.size(), ref.off.size());
                    return 1;
                }
                if (ref.ok && count_rows(exact.data(), exact.data() + exact.size(), skip != 0) != (int64_t)ref.off.size()) {
                    std::printf("EDGE COUNT MISMATCH case=%zu skip=%d\n", ci, skip);
                    return 1;
                }
            }
        }
        std::printf("edges: %zu texts\n", cases.size());
    }
    // VCF reader on the fixture and mutated copies (plain text)
    std::string vbase = slurp(vcf);
    ok = bad = 0;
    for (int it = 0; it < 300; ++it) {
        std::string t = vbase;
        if (it) for (int k = 0; k < 1 + (int)(rng() % 6); ++k) {
            const size_t at = rng() % t.size();
            switch (rng() % 4) {
            case 0: t[at] = "\t,ACGT<>|/.0123456789\n"[rng() % 22]; break;
            case 1: t.erase(at, 1 + rng() % 30); break;
            case 2: t.insert(at, std::string(1 + rng() % 300, "ACGT"[rng() % 4])); break;
            default: t.insert(at, t.substr(rng() % t.size(), rng() % 100)); break;
            }
        }
        std::string path = std::string(tmp) + "/m.vcf";
        { std::ofstream o(path, std::ios::binary); o << t; }
        gfm_vcf_t v = nullptr; int64_t ns = 0, sk = 0; int32_t H = 0;
        const int rc = gfm_vcf_open(path.c_str(), "x", 1, 1 + it % 4, &v, &ns, &H, &sk);
        if (rc == 0) {
            ++ok;
            const int hw = (H + 63) / 64;
            std::vector<int32_t> pos(ns + 1), dl(ns + 1), il(ns + 1), io(ns + 1); std::vector<uint8_t> na(ns + 1), ab(3 * ns + 3), ib(gfm_vcf_ins_bytes(v) + 1);
            std::vector<uint64_t> bits((size_t)ns * 3 * hw + 1);
            gfm_vcf_read(v, pos.data(), na.data(), ab.data(), dl.data(), bits.data());
            gfm_vcf_read_insertions(v, il.data(), io.data(), ib.data());
            gfm_vcf_close(v);
        } else ++bad;
    }
    std::printf("vcf: %ld parsed, %ld refused\n", ok, bad);
    // the hit entries' order (gfm_hit_sort.hpp): distinct rows << 20 | score, every size class of the radix passes
    for (int it = 0; it < 60; ++it) {
        const size_t n = it < 6 ? (size_t)it : (size_t)(rng() % (it < 30 ? 2000 : 300000));
        const int row_bits = 1 + (int)(rng() % 40);
        std::vector<int64_t> v(n);
        int64_t row = 0;
        for (size_t i = 0; i < n; ++i) {
            row += 1 + (int64_t)(rng() % (uint64_t)std::max<int64_t>(1, (1ll << row_bits) / (int64_t)(n + 1)));   // distinct, < 2^43
            v[i] = (row << 20) | (int64_t)(rng() % (1 << 20));
        }
        std::shuffle(v.begin(), v.end(), rng);
        std::vector<int64_t> want = v;
        std::stable_sort(want.begin(), want.end(), [](int64_t a, int64_t b) { return (a >> 20) < (b >> 20); });
        gfm_hit_sort::sort_packed(v.data(), v.size(), 20);
        if (v != want) { std::printf("SORT MISMATCH it=%d n=%zu\n", it, n); return 1; }
    }
    std::printf("sort: ok\n");
    return 0;
}

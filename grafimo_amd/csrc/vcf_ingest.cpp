// vcf_ingest.cpp -- host-side reader of a phased VCF into the site arrays of gfm_graph_create.
//
// The reference never reads a VCF itself: `grafimo buildvg` hands it to `vg construct` / `vg index -G`
// (src/grafimo/constructVG.py:332,394).  For the extraction kernels (graph_extract.hip) the records
// of one chromosome become: SNP sites (REF and every ALT one base of ACGT, at most 3 ALTs), deletions
// (REF = anchor + deleted bases, one ALT = the anchor), everything else skipped and counted, as are
// a second SNP record at one position and a deletion that touches one accepted before it -- the
// rules oracle/extract_oracle.py read_vcf_graph states.  Genotypes: two haplotypes per sample in
// file order ("a|b", "a/b" taken as written, a single allele doubled, "." = reference); per
// alternate allele one bitset over the haplotypes, bit h of word h / 64.
// Plain text is mmap'ed, .gz goes through zlib (any gzip/bgzip stream); lines are parsed by a small
// pool of host threads, the order-dependent acceptance rules run in one pass afterwards.

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "grafimo_hip.h"

#define GFM_API extern "C" __attribute__((visibility("default")))
extern "C" void gfm_set_error_(const char *msg);

namespace {

constexpr int kMaxAlts = 3;

struct Rec {
    int64_t pos = 0;          // 0-based
    uint8_t kind = 0;         // 0 SNP, 1 deletion, 2 skipped
    uint8_t n_alts = 0;
    uint8_t alt[kMaxAlts] = {0, 0, 0};
    int32_t del_len = 0;
    size_t bits_at = 0;       // offset (words) of this record's [3][hw] block in its chunk's bit store
};

struct Chunk {
    std::vector<Rec> recs;
    std::vector<uint64_t> bits;
    std::vector<uint8_t> alleles;      // scratch of parse_line, reused from line to line
    int n_hap = -1;
    std::string error;
};

inline const char *field_end(const char *p, const char *e)
{
    const char *t = static_cast<const char *>(memchr(p, '\t', (size_t)(e - p)));
    return t ? t : e;
}

inline bool is_base(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
inline char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

// one data line [b, e) of the wanted chromosome
void parse_line(const char *b, const char *e, bool want_hap, int hw_hint, Chunk &out)
{
    const char *f[10];
    const char *p = b;
    int nf = 0;
    while (nf < 10 && p <= e) {
        f[nf++] = p;
        p = field_end(p, e);
        if (p == e) break;
        ++p;
    }
    if (nf < 5) { out.error = "VCF line with fewer than 5 columns"; return; }
    auto fe = [&](int k) { return field_end(f[k], e); };
    Rec r;
    r.pos = strtoll(f[1], nullptr, 10) - 1;
    const char *ref_b = f[3], *ref_e = fe(3), *alt_b = f[4], *alt_e = fe(4);
    const long ref_len = (long)(ref_e - ref_b);
    // ALT list
    const char *ab[8];
    long al[8];
    int na = 0;
    for (const char *q = alt_b; q <= alt_e && na < 8;) {
        const char *c = static_cast<const char *>(memchr(q, ',', (size_t)(alt_e - q)));
        const char *qe = c ? c : alt_e;
        ab[na] = q; al[na] = (long)(qe - q); ++na;
        if (!c) break;
        q = c + 1;
    }
    bool snp = ref_len == 1 && na >= 1 && na <= kMaxAlts;
    for (int k = 0; snp && k < na; ++k) snp = al[k] == 1 && is_base(up(ab[k][0]));
    const bool del = !snp && ref_len > 1 && na == 1 && al[0] == 1 && up(ab[0][0]) == up(ref_b[0]);
    if (snp) {
        r.kind = 0;
        r.n_alts = (uint8_t)na;
        for (int k = 0; k < na; ++k) r.alt[k] = (uint8_t)up(ab[k][0]);
    } else if (del) {
        r.kind = 1;
        r.n_alts = 1;
        r.del_len = (int32_t)(ref_len - 1);
    } else {
        r.kind = 2;
        out.recs.push_back(r);
        return;
    }
    if (want_hap && nf == 10) {
        // genotype columns start at f[9]; count haplotypes on the first record of the chunk
        std::vector<uint8_t> &alleles = out.alleles;
        alleles.clear();
        const char *q = f[9];
        while (q < e) {
            const char *qe = field_end(q, e);
            const char *colon = static_cast<const char *>(memchr(q, ':', (size_t)(qe - q)));
            const char *ge = colon ? colon : qe;
            // a|b, a/b, a
            int a0 = 0, a1 = 0;
            const char *s = q;
            bool digit = false;
            int v = 0;
            while (s < ge && *s >= '0' && *s <= '9') { v = v * 10 + (*s - '0'); ++s; digit = true; }
            a0 = digit ? v : 0;
            while (s < ge && *s != '|' && *s != '/') ++s;       // "." or junk
            if (s < ge) {
                ++s;
                digit = false; v = 0;
                while (s < ge && *s >= '0' && *s <= '9') { v = v * 10 + (*s - '0'); ++s; digit = true; }
                a1 = digit ? v : 0;
            } else {
                a1 = a0;                                          // a single allele counts twice
            }
            alleles.push_back((uint8_t)std::min(a0, 255));
            alleles.push_back((uint8_t)std::min(a1, 255));
            if (qe >= e) break;
            q = qe + 1;
        }
        const int H = (int)alleles.size();
        if (out.n_hap < 0) out.n_hap = H;
        if (H != out.n_hap) { out.error = "VCF records with different numbers of samples"; return; }
        const int hw = (H + 63) / 64;
        r.bits_at = out.bits.size();
        out.bits.resize(out.bits.size() + (size_t)kMaxAlts * hw, 0ull);
        uint64_t *dst = out.bits.data() + r.bits_at;
        for (int h = 0; h < H; ++h) {
            const int a = alleles[(size_t)h];
            if (a >= 1 && a <= kMaxAlts && a <= r.n_alts) dst[(size_t)(a - 1) * hw + (h >> 6)] |= 1ull << (h & 63);
        }
    }
    out.recs.push_back(r);
}

// the bytes of the file: plain text is mapped, gzip / bgzip is inflated into `store`
bool read_all(const char *path, std::string &store, const char **data, size_t *size, void **map, size_t *map_len,
              std::string &err)
{
    *map = nullptr;
    *map_len = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open ") + path; return false; }
    unsigned char magic[2] = {0, 0};
    const bool gz = read(fd, magic, 2) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    struct stat st;
    if (!gz && fstat(fd, &st) == 0 && st.st_size > 0) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { err = std::string("cannot map ") + path; return false; }
        *map = m;
        *map_len = (size_t)st.st_size;
        *data = static_cast<const char *>(m);
        *size = (size_t)st.st_size;
        return true;
    }
    close(fd);
    std::string &buf = store;
    gzFile fh = gzopen(path, "rb");
    if (!fh) { err = std::string("cannot open ") + path; return false; }
    gzbuffer(fh, 1 << 20);
    std::vector<char> tmp(1 << 22);
    for (;;) {
        const int n = gzread(fh, tmp.data(), (unsigned)tmp.size());
        if (n < 0) { err = std::string("read error in ") + path; gzclose(fh); return false; }
        if (n == 0) break;
        buf.append(tmp.data(), (size_t)n);
    }
    gzclose(fh);
    *data = buf.data();
    *size = buf.size();
    return true;
}

}  // namespace

struct gfm_vcf {
    std::vector<int32_t> pos, del_len;
    std::vector<uint8_t> n_alts, alt_bases;
    std::vector<uint64_t> bits;      // [n][3][hw]
    int n_hap = 0;
    int64_t skipped = 0;
};

GFM_API int gfm_vcf_open(const char *path, const char *chrom, int with_haplotypes, int n_threads,
                         gfm_vcf_t *out, int64_t *n_sites, int32_t *n_haplotypes, int64_t *n_skipped)
{
    if (!out || !path || !chrom) { gfm_set_error_("NULL argument"); return GFM_ERR_INVALID; }
    *out = nullptr;
    std::string store, err;
    const char *data = nullptr;
    size_t size = 0, map_len = 0;
    void *map = nullptr;
    if (!read_all(path, store, &data, &size, &map, &map_len, err)) { gfm_set_error_(err.c_str()); return GFM_ERR_IO; }
    struct Unmap {
        void *m; size_t n;
        ~Unmap() { if (m) munmap(m, n); }
    } unmap{map, map_len};
    // data lines of the chromosome
    const size_t clen = strlen(chrom);
    std::vector<std::pair<size_t, size_t>> lines;
    for (size_t b = 0; b < size;) {
        const char *nl = static_cast<const char *>(memchr(data + b, '\n', size - b));
        const size_t e = nl ? (size_t)(nl - data) : size;
        size_t le = e;
        if (le > b && data[le - 1] == '\r') --le;
        if (le > b && data[b] != '#' && le - b > clen && data[b + clen] == '\t' && memcmp(data + b, chrom, clen) == 0)
            lines.emplace_back(b, le);
        b = e + 1;
    }
    const int nt = std::max(1, std::min<int>(n_threads, (int)(lines.size() / 256) + 1));
    std::vector<Chunk> chunks((size_t)nt);
    std::vector<std::thread> pool;
    const size_t per = (lines.size() + (size_t)nt - 1) / (size_t)nt;
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&, t]() {
            Chunk &c = chunks[(size_t)t];
            const size_t lo = (size_t)t * per, hi = std::min(lines.size(), lo + per);
            for (size_t k = lo; k < hi && c.error.empty(); ++k)
                parse_line(data + lines[k].first, data + lines[k].second, with_haplotypes != 0,
                           c.n_hap > 0 ? (c.n_hap + 63) / 64 : 0, c);
        });
    for (auto &th : pool) th.join();
    int H = -1;
    for (auto &c : chunks) {
        if (!c.error.empty()) { gfm_set_error_(c.error.c_str()); return GFM_ERR_IO; }
        if (c.n_hap >= 0) {
            if (H >= 0 && c.n_hap != H) { gfm_set_error_("VCF records with different numbers of samples"); return GFM_ERR_IO; }
            H = c.n_hap;
        }
    }
    if (H < 0) H = 0;
    gfm_vcf *v = new (std::nothrow) gfm_vcf();
    if (!v) { gfm_set_error_("out of host memory"); return GFM_ERR_NOMEM; }
    v->n_hap = with_haplotypes ? H : 0;
    const int hw = (v->n_hap + 63) / 64;
    // acceptance rules in file order, then (pos, kind) order: a deletion follows the SNP at its anchor
    struct Ref { int chunk; size_t idx; };
    std::vector<Ref> keep;
    int64_t last_snp = -1, deleted_until = -1, prev_pos = -1;
    bool sorted = true;
    for (int c = 0; c < nt; ++c)
        for (size_t k = 0; k < chunks[(size_t)c].recs.size(); ++k) {
            const Rec &r = chunks[(size_t)c].recs[k];
            if (r.kind == 0 && r.pos != last_snp) {
                last_snp = r.pos;
            } else if (r.kind == 1 && r.pos > deleted_until) {
                deleted_until = r.pos + r.del_len;
            } else {
                ++v->skipped;
                continue;
            }
            if (r.pos < prev_pos) sorted = false;
            prev_pos = r.pos;
            keep.push_back({c, k});
        }
    if (!sorted) {
        delete v;
        gfm_set_error_("VCF records of the chromosome are not sorted by position");
        return GFM_ERR_IO;
    }
    std::stable_sort(keep.begin(), keep.end(), [&](const Ref &a, const Ref &b) {
        const Rec &ra = chunks[(size_t)a.chunk].recs[a.idx], &rb = chunks[(size_t)b.chunk].recs[b.idx];
        return ra.pos != rb.pos ? ra.pos < rb.pos : ra.kind < rb.kind;
    });
    const size_t n = keep.size();
    v->pos.resize(n); v->del_len.resize(n); v->n_alts.resize(n); v->alt_bases.assign(n * kMaxAlts, 0);
    if (hw) v->bits.assign(n * (size_t)kMaxAlts * hw, 0ull);
    for (size_t i = 0; i < n; ++i) {
        const Chunk &c = chunks[(size_t)keep[i].chunk];
        const Rec &r = c.recs[keep[i].idx];
        v->pos[i] = (int32_t)r.pos;
        v->del_len[i] = r.del_len;
        v->n_alts[i] = r.n_alts;
        for (int a = 0; a < kMaxAlts; ++a) v->alt_bases[i * kMaxAlts + a] = r.alt[a];
        if (hw && !c.bits.empty())
            memcpy(v->bits.data() + i * (size_t)kMaxAlts * hw, c.bits.data() + r.bits_at, sizeof(uint64_t) * kMaxAlts * hw);
    }
    if (n_sites) *n_sites = (int64_t)n;
    if (n_haplotypes) *n_haplotypes = v->n_hap;
    if (n_skipped) *n_skipped = v->skipped;
    *out = v;
    return GFM_OK;
}

GFM_API int gfm_vcf_read(gfm_vcf_t v, int32_t *pos, uint8_t *n_alts, uint8_t *alt_bases, int32_t *del_len,
                         uint64_t *alt_bits)
{
    if (!v) { gfm_set_error_("VCF handle is NULL"); return GFM_ERR_INVALID; }
    const size_t n = v->pos.size();
    if (n && (!pos || !n_alts || !alt_bases || !del_len)) { gfm_set_error_("NULL output buffer"); return GFM_ERR_INVALID; }
    if (n) {
        memcpy(pos, v->pos.data(), sizeof(int32_t) * n);
        memcpy(n_alts, v->n_alts.data(), n);
        memcpy(alt_bases, v->alt_bases.data(), n * kMaxAlts);
        memcpy(del_len, v->del_len.data(), sizeof(int32_t) * n);
        if (alt_bits && !v->bits.empty()) memcpy(alt_bits, v->bits.data(), sizeof(uint64_t) * v->bits.size());
    }
    return GFM_OK;
}

GFM_API void gfm_vcf_close(gfm_vcf_t v) { delete v; }

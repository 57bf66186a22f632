// vcf_ingest.cpp -- host-side reader of a phased VCF into the site arrays of gfm_graph_create.
//
// The reference never reads a VCF itself: `grafimo buildvg` hands it to `vg construct` / `vg index -G`
// (src/grafimo/constructVG.py:332,394).  For the extraction kernels (graph_extract.hip) the records
// of one chromosome are taken apart per ALT allele: single-base substitutions (one site per position, up
// to 3 alternates, also when they come from several records), insertions (REF = anchor, ALT = anchor +
// inserted bases), deletions (REF = anchor + deleted bases, ALT = anchor; they may overlap), multi-base
// substitutions of equal length (one substitution per mismatching position) -- every ALT normalised against REF
// first (common trailing, then leading bases dropped).  What is still neither of these is a COMPLEX allele
// (REF=ACG ALT=TC): it is taken apart into the substitutions of its first min(|REF|, |ALT|) bases and an
// insertion / deletion of the rest behind the last of them, all carried by the same haplotypes -- `vg construct`
// (called without --flat-alts, constructVG.py:332) decomposes such alleles too, by alignment; the haplotype
// sequences, and with them every row a haplotype carries, do not depend on how.  A record with an ALT that is not
// a string of A, C, G, T (symbolic <DEL> / <CN0> / <INS:ME:ALU>, breakends, '*', IUPAC codes) is left out whole and its alleles are
// counted in `skipped`: `vg construct` without --handle-sv (the reference's command line) skips such records with
// a warning.  The rules oracle/extract_oracle.py read_vcf_variants states.  Genotypes: two haplotypes per sample in
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
#include "gfm_workers.hpp"

#define GFM_API extern "C" __attribute__((visibility("default")))
extern "C" void gfm_set_error_(const char *msg);

namespace {

constexpr int kMaxAlts = 3;          // alternates of one substitution site (four bases: there cannot be more)
constexpr int kMaxRecordAlts = 64;   // ALT alleles taken from one record (insertions and deletions are sites of their own)

// One alternate allele of a record, taken apart (oracle/extract_oracle.py read_vcf_variants states the rules):
// kind 0 a single-base substitution at `pos`, 1 an insertion behind the anchor `pos`, 2 a deletion behind it.
// A multi-base substitution of equal length yields one kind-0 atom per mismatching position.
struct Atom {
    int64_t pos = 0;          // 0-based
    uint8_t kind = 0;
    uint8_t base = 0;         // substitution: the alternate base
    int32_t del_len = 0;
    int32_t ins_len = 0;
    size_t ins_at = 0;        // offset of the inserted bases in the chunk's pool
    size_t bits_at = 0;       // offset (words) of the allele's carrier bitset [hw] in the chunk's bit store
};

struct Chunk {
    std::vector<Atom> atoms;
    std::vector<int64_t> rec_pos;      // position of every record seen (sortedness check)
    std::vector<uint64_t> bits;
    std::vector<uint8_t> ins_pool;
    std::vector<uint8_t> alleles;      // scratch of parse_line, reused from line to line
    std::vector<int> allele_of_atom;   // scratch of parse_line: ALT index of every atom of the record (any number:
                                       // a long multi-base substitution yields one atom per mismatching position)
    int n_hap = -1;
    int64_t skipped = 0;
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
void parse_line(const char *b, const char *e, bool want_hap, Chunk &out)
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
    const int64_t pos = strtoll(f[1], nullptr, 10) - 1;
    out.rec_pos.push_back(pos);
    const char *ref_b = f[3], *ref_e = fe(3), *alt_b = f[4], *alt_e = fe(4);
    const long ref_len = (long)(ref_e - ref_b);
    bool ref_ok = ref_len > 0;
    for (const char *q = ref_b; q < ref_e && ref_ok; ++q) ref_ok = is_base(up(*q)) || up(*q) == 'N';
    // ALT list
    const char *ab[kMaxRecordAlts];
    long al[kMaxRecordAlts];
    int na = 0;
    for (const char *q = alt_b; q <= alt_e;) {
        const char *c = static_cast<const char *>(memchr(q, ',', (size_t)(alt_e - q)));
        const char *qe = c ? c : alt_e;
        if (na < kMaxRecordAlts) { ab[na] = q; al[na] = (long)(qe - q); }
        ++na;
        if (!c) break;
        q = c + 1;
    }
    // a record with an ALT that is no base string (symbolic, breakend, '*', '.') or with a REF that is none is left
    // out whole, as vg construct does; so are the ALTs beyond kMaxRecordAlts
    bool record_ok = ref_ok;
    for (int k = 0; k < na && k < kMaxRecordAlts && record_ok; ++k) {
        record_ok = al[k] > 0;
        for (long j = 0; record_ok && j < al[k]; ++j) record_ok = is_base(up(ab[k][j]));
    }
    if (!record_ok) { out.skipped += na; return; }
    // what every ALT is
    const size_t first_atom = out.atoms.size();
    std::vector<int> &allele_of_atom = out.allele_of_atom;
    allele_of_atom.clear();
    for (int k = 0; k < na; ++k) {
        const bool ok = k < kMaxRecordAlts;
        // per-allele normalisation (oracle/extract_oracle.py read_vcf_variants): common trailing bases go while both
        // strings keep one base, then common leading bases (the position moves right) -- the alleles of an STR record
        // (REF=ATTT ALT=A,AT,ATT,ATTTT) become deletions and an insertion anchored on its first base
        const char *rb = ref_b, *re = ref_e, *qb = ok ? ab[k] : ref_b, *qe = ok ? ab[k] + al[k] : ref_b;
        int64_t pp = pos;
        if (ok) {
            while (re - rb > 1 && qe - qb > 1 && up(re[-1]) == up(qe[-1])) { --re; --qe; }
            while (re - rb > 1 && qe - qb > 1 && up(rb[0]) == up(qb[0])) { ++rb; ++qb; ++pp; }
        }
        const long rl = (long)(re - rb), ql = (long)(qe - qb);
        if (ok && rl == 1 && ql == 1) {
            if (up(rb[0]) != up(qb[0])) {
                Atom a; a.pos = pp; a.kind = 0; a.base = (uint8_t)up(qb[0]);
                out.atoms.push_back(a); allele_of_atom.push_back(k);
            }
        } else if (ok && rl > 1 && ql == 1 && up(qb[0]) == up(rb[0])) {
            Atom a; a.pos = pp; a.kind = 2; a.del_len = (int32_t)(rl - 1);
            out.atoms.push_back(a); allele_of_atom.push_back(k);
        } else if (ok && rl == 1 && ql > 1 && up(qb[0]) == up(rb[0])) {
            Atom a; a.pos = pp; a.kind = 1; a.ins_len = (int32_t)(ql - 1); a.ins_at = out.ins_pool.size();
            for (long j = 1; j < ql; ++j) out.ins_pool.push_back((uint8_t)up(qb[j]));
            out.atoms.push_back(a); allele_of_atom.push_back(k);
        } else if (ok && rl == ql && rl > 1) {
            for (long j = 0; j < rl; ++j)
                if (up(rb[j]) != up(qb[j])) {
                    Atom a; a.pos = pp + j; a.kind = 0; a.base = (uint8_t)up(qb[j]);
                    out.atoms.push_back(a); allele_of_atom.push_back(k);
                }
        } else if (ok) {
            // complex: substitutions over the common length, the rest an insertion / deletion behind its last base
            const long common = std::min(rl, ql);
            for (long j = 0; j < common; ++j)
                if (up(rb[j]) != up(qb[j])) {
                    Atom a; a.pos = pp + j; a.kind = 0; a.base = (uint8_t)up(qb[j]);
                    out.atoms.push_back(a); allele_of_atom.push_back(k);
                }
            if (ql > rl) {
                Atom a; a.pos = pp + common - 1; a.kind = 1; a.ins_len = (int32_t)(ql - common); a.ins_at = out.ins_pool.size();
                for (long j = common; j < ql; ++j) out.ins_pool.push_back((uint8_t)up(qb[j]));
                out.atoms.push_back(a); allele_of_atom.push_back(k);
            } else if (rl > ql) {
                Atom a; a.pos = pp + common - 1; a.kind = 2; a.del_len = (int32_t)(rl - common);
                out.atoms.push_back(a); allele_of_atom.push_back(k);
            }
        } else {
            ++out.skipped;
        }
    }
    const int n_new = (int)allele_of_atom.size();
    if (n_new == 0) return;
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
        // one carrier bitset per ALT of the record that yielded atoms; the atoms of one ALT share it
        size_t at_of_allele[kMaxRecordAlts];
        for (int k = 0; k < kMaxRecordAlts; ++k) at_of_allele[k] = (size_t)-1;
        for (int t = 0; t < n_new; ++t) {
            const int k = allele_of_atom[(size_t)t];
            if (at_of_allele[k] == (size_t)-1) {
                at_of_allele[k] = out.bits.size();
                out.bits.resize(out.bits.size() + (size_t)hw, 0ull);
                uint64_t *dst = out.bits.data() + at_of_allele[k];
                for (int h = 0; h < H; ++h)
                    if (alleles[(size_t)h] == k + 1 && k + 1 <= na) dst[h >> 6] |= 1ull << (h & 63);
            }
            out.atoms[first_atom + (size_t)t].bits_at = at_of_allele[k];
        }
    }
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
    std::vector<int32_t> pos, del_len, ins_len, ins_off;
    std::vector<uint8_t> n_alts, alt_bases, ins_bases;
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
    const size_t per = (lines.size() + (size_t)nt - 1) / (size_t)nt;
    std::atomic<int> next_part{0};
    gfm_workers::run(nt, [&]() {
        const int t = next_part.fetch_add(1);
        Chunk &c = chunks[(size_t)t];
        const size_t lo = (size_t)t * per, hi = std::min(lines.size(), lo + per);
        for (size_t k = lo; k < hi && c.error.empty(); ++k)
            parse_line(data + lines[k].first, data + lines[k].second, with_haplotypes != 0, c);
    });
    int H = -1;
    for (auto &c : chunks) {
        if (!c.error.empty()) { gfm_set_error_(c.error.c_str()); return GFM_ERR_IO; }
        if (c.n_hap >= 0) {
            if (H >= 0 && c.n_hap != H) { gfm_set_error_("VCF records with different numbers of samples"); return GFM_ERR_IO; }
            H = c.n_hap;
        }
    }
    if (H < 0) H = 0;
    int64_t prev_pos = -1;
    for (auto &c : chunks)
        for (int64_t rp : c.rec_pos) {
            if (rp < prev_pos) {
                gfm_set_error_("VCF records of the chromosome are not sorted by position");
                return GFM_ERR_IO;
            }
            prev_pos = rp;
        }
    gfm_vcf *v = new (std::nothrow) gfm_vcf();
    if (!v) { gfm_set_error_("out of host memory"); return GFM_ERR_NOMEM; }
    v->n_hap = with_haplotypes ? H : 0;
    const int hw = (v->n_hap + 63) / 64;
    // atoms in (position, kind) order -- substitution < insertion < deletion, file order inside a tie
    struct Ref { int chunk; size_t idx; };
    std::vector<Ref> order;
    for (int c = 0; c < nt; ++c) {
        v->skipped += chunks[(size_t)c].skipped;
        for (size_t k = 0; k < chunks[(size_t)c].atoms.size(); ++k) order.push_back({c, k});
    }
    auto atom = [&](const Ref &r) -> const Atom & { return chunks[(size_t)r.chunk].atoms[r.idx]; };
    std::stable_sort(order.begin(), order.end(), [&](const Ref &a, const Ref &b) {
        const Atom &ra = atom(a), &rb = atom(b);
        return ra.pos != rb.pos ? ra.pos < rb.pos : ra.kind < rb.kind;
    });
    auto add_site = [&](int64_t pos, int dl, int il, size_t ioff) {
        v->pos.push_back((int32_t)pos);
        v->del_len.push_back(dl);
        v->ins_len.push_back(il);
        v->ins_off.push_back((int32_t)ioff);
        v->n_alts.push_back(0);
        v->alt_bases.insert(v->alt_bases.end(), kMaxAlts, (uint8_t)0);
        if (hw) v->bits.insert(v->bits.end(), (size_t)kMaxAlts * hw, 0ull);
    };
    auto or_bits = [&](size_t site, int slot, const Ref &r) {
        if (!hw) return;
        const Chunk &c = chunks[(size_t)r.chunk];
        if (c.bits.empty()) return;
        uint64_t *dst = v->bits.data() + (site * kMaxAlts + (size_t)slot) * hw;
        const uint64_t *src = c.bits.data() + atom(r).bits_at;
        for (int w = 0; w < hw; ++w) dst[w] |= src[w];
    };
    for (const Ref &r : order) {
        const Atom &a = atom(r);
        const size_t n = v->pos.size();
        if (a.kind == 0) {
            if (n && v->pos[n - 1] == a.pos && v->del_len[n - 1] == 0 && v->ins_len[n - 1] == 0) {
                // one site per position: another alternate, or more carriers of one it already has
                const size_t s = n - 1;
                int slot = -1;
                for (int k = 0; k < v->n_alts[s]; ++k)
                    if (v->alt_bases[s * kMaxAlts + k] == a.base) slot = k;
                if (slot < 0) {
                    if (v->n_alts[s] >= kMaxAlts) { ++v->skipped; continue; }
                    slot = v->n_alts[s]++;
                    v->alt_bases[s * kMaxAlts + slot] = a.base;
                }
                or_bits(s, slot, r);
            } else {
                add_site(a.pos, 0, 0, 0);
                v->n_alts[n] = 1;
                v->alt_bases[n * kMaxAlts] = a.base;
                or_bits(n, 0, r);
            }
        } else if (a.kind == 1) {
            const uint8_t *seq = chunks[(size_t)r.chunk].ins_pool.data() + a.ins_at;
            size_t dup = (size_t)-1;
            for (size_t s = n; s-- > 0 && v->pos[s] == a.pos;)
                if (v->ins_len[s] == a.ins_len && memcmp(v->ins_bases.data() + v->ins_off[s], seq, (size_t)a.ins_len) == 0)
                    dup = s;
            if (dup != (size_t)-1) {
                or_bits(dup, 0, r);
            } else {
                add_site(a.pos, 0, a.ins_len, v->ins_bases.size());
                v->ins_bases.insert(v->ins_bases.end(), seq, seq + a.ins_len);
                v->n_alts[n] = 1;
                or_bits(n, 0, r);
            }
        } else {
            // deletions may overlap (several lengths at one anchor: an STR record; an anchor inside another's span):
            // the graph holds all of them; two records deleting the same bases merge their carriers
            size_t dup = (size_t)-1;
            for (size_t s = n; s-- > 0 && v->pos[s] == a.pos;)
                if (v->del_len[s] == a.del_len) dup = s;
            if (dup != (size_t)-1) {
                or_bits(dup, 0, r);
            } else {
                add_site(a.pos, a.del_len, 0, 0);
                v->n_alts[n] = 1;
                or_bits(n, 0, r);
            }
        }
    }
    if (n_sites) *n_sites = (int64_t)v->pos.size();
    if (n_haplotypes) *n_haplotypes = v->n_hap;
    if (n_skipped) *n_skipped = v->skipped;
    *out = v;
    return GFM_OK;
}

GFM_API int64_t gfm_vcf_ins_bytes(gfm_vcf_t v) { return v ? (int64_t)v->ins_bases.size() : 0; }

GFM_API int gfm_vcf_read_insertions(gfm_vcf_t v, int32_t *ins_len, int32_t *ins_off, uint8_t *ins_bases)
{
    if (!v) { gfm_set_error_("VCF handle is NULL"); return GFM_ERR_INVALID; }
    const size_t n = v->pos.size();
    if (n && (!ins_len || !ins_off)) { gfm_set_error_("NULL output buffer"); return GFM_ERR_INVALID; }
    if (n) {
        memcpy(ins_len, v->ins_len.data(), sizeof(int32_t) * n);
        memcpy(ins_off, v->ins_off.data(), sizeof(int32_t) * n);
    }
    if (ins_bases && !v->ins_bases.empty()) memcpy(ins_bases, v->ins_bases.data(), v->ins_bases.size());
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

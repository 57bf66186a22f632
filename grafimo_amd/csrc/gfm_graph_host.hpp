// gfm_graph_host.hpp -- the host-side view of a gfm_graph_t that the TSV writer needs (graph_tsv_writer.cpp): the site
// arrays as gfm_graph_create received them and, built on first use, the node table of `vg construct` on this graph --
// column 7 of the rows `vg find -K` prints (extract_regions.py:180,225; GRAFIMO's scoring never reads it,
// score_sequences.py:279-293).  Part of libgrafimo_hip.so.
#pragma once

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

namespace gfm_host {

constexpr int kNodeMax = 32;          // vg construct -m default: an invariant stretch is chopped into nodes of <= 32 bases

struct HostGraph {
    long long ref_len = 0;
    std::vector<int> pos, del_len, ins_len;       // [n_sites]
    std::vector<unsigned char> n_alts;            // [n_sites]
    std::vector<long long> max_reach;             // [n_sites + 1]: last base removed by any deletion among sites [0, i), or -1
    bool has_ins = false;
    // ---- node table (GraphIndex._node_table of round 1-4's Python writer; pinned by vg's own .vg / .xg files,
    // tests/test_vg_pins.py): the reference is cut at every SNP (a node of its own: alternates numbered first, then the
    // reference allele), at both ends of every deleted stretch and behind the anchor of every insertion; what lies between
    // two cuts is chopped into nodes of <= 32 bases; the nodes of an insertion are numbered right behind the interval that
    // ends with its anchor.
    std::once_flag nodes_once;
    std::vector<long long> cuts;                  // ascending, cuts[0] = 0, last = ref_len
    std::vector<long long> first;                 // [intervals] id of the interval's first node
    std::vector<int> site_of;                     // [intervals] its SNP site, or -1
    std::vector<long long> ins_first;             // [n_sites] first node of an insertion site (else -1)
    void build_nodes();
    int n_sites() const { return (int)pos.size(); }
};

struct WriteStats {
    long long n_rows = 0, n_files = 0, bytes = 0;
    double total_s = 0, copy_s = 0, format_s = 0;
    int threads = 0, reserved = 0;
};

// one chunk of rows on the host (pointers into the caller's staging buffers), rows [row0, row0 + n)
struct RowChunk {
    const unsigned char *kmers;       // [n][W]
    const long long *start, *stop, *freq;
    const unsigned char *strand, *is_ref;
    const int *region, *walk;
    long long n;
};

struct WriteJob {
    int W = 0, n_regions = 0;
    const long long *region_stop = nullptr;       // [n_regions] E of region r (a walk must end inside it)
    const char *const *labels = nullptr;          // [n_regions] column 1
    const char *const *paths = nullptr;           // [n_regions] the file of region r
    const char *chrom = nullptr;                  // the name printed in columns 3 and 4
    bool node_paths = true;
    unsigned char *seen = nullptr;                // [n_regions] in/out: the region's file exists (later rows are appended)
    int threads = 1;
};

// Formats the rows of one chunk, one piece per run of equal region ids, and writes / appends them to their files.
// Returns 0 or a GFM_ERR_* code with `err` filled in.
int write_chunk(HostGraph &g, const WriteJob &job, const RowChunk &rows, WriteStats &stats, std::string &err);

}  // namespace gfm_host

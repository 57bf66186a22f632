// TEST INFRASTRUCTURE: drives the TSV writer of libgrafimo_hip.so (csrc/graph_tsv_writer.cpp: gfm_host::write_chunk, what
// gfm_graph_write_tsvs runs per chunk of rows) on HOST data, so that its node paths and layout enumeration can be compared
// with the Python writer of rounds 1-4 without a GPU, and under -fsanitize=address,undefined.
//   tsv_writer_host <dir> <chunk_rows> <threads> <node_paths 0|1>
// <dir> holds raw little-endian arrays written by tests/test_tsv_writer_host.py: meta.txt (ref_len n_sites W n_regions
// n_rows chrom), pos.i32 del_len.i32 ins_len.i32 n_alts.u8, kmers.u8 start.i64 stop.i64 freq.i64 region.i32 walk.i32
// strand.u8 is_ref.u8, region_stop.i64, labels.txt, paths.txt (one per line).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "gfm_graph_host.hpp"

template <typename T> static std::vector<T> slurp(const std::string &path, size_t count)
{
    std::vector<T> v(count);
    if (count == 0) return v;
    std::ifstream f(path, std::ios::binary);
    if (!f || !f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(count * sizeof(T)))) {
        std::cerr << "cannot read " << path << "\n";
        std::exit(2);
    }
    return v;
}
static std::vector<std::string> lines(const std::string &path)
{
    std::vector<std::string> out;
    std::ifstream f(path);
    for (std::string s; std::getline(f, s);) out.push_back(s);
    return out;
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const std::string d = std::string(argv[1]) + "/";
    const long long chunk_rows = atoll(argv[2]);
    const int threads = atoi(argv[3]);
    const bool node_paths = atoi(argv[4]) != 0;
    long long ref_len = 0, n_rows = 0;
    int n_sites = 0, W = 0, n_regions = 0;
    std::string chrom;
    {
        std::ifstream m(d + "meta.txt");
        m >> ref_len >> n_sites >> W >> n_regions >> n_rows >> chrom;
    }
    gfm_host::HostGraph g;
    g.ref_len = ref_len;
    g.pos = slurp<int>(d + "pos.i32", (size_t)n_sites);
    g.del_len = slurp<int>(d + "del_len.i32", (size_t)n_sites);
    g.ins_len = slurp<int>(d + "ins_len.i32", (size_t)n_sites);
    g.n_alts = slurp<unsigned char>(d + "n_alts.u8", (size_t)n_sites);
    g.max_reach.assign((size_t)n_sites + 1, -1);             // as gfm_graph_create fills it
    long long until = -1;
    for (int i = 0; i < n_sites; ++i) {
        if (g.del_len[(size_t)i] > 0) until = std::max(until, (long long)g.pos[(size_t)i] + g.del_len[(size_t)i]);
        if (g.ins_len[(size_t)i] > 0) g.has_ins = true;
        g.max_reach[(size_t)i + 1] = until;
    }
    const auto kmers = slurp<unsigned char>(d + "kmers.u8", (size_t)n_rows * (size_t)W);
    const auto start = slurp<long long>(d + "start.i64", (size_t)n_rows), stop = slurp<long long>(d + "stop.i64", (size_t)n_rows),
               freq = slurp<long long>(d + "freq.i64", (size_t)n_rows);
    const auto region = slurp<int>(d + "region.i32", (size_t)n_rows), walk = slurp<int>(d + "walk.i32", (size_t)n_rows);
    const auto strand = slurp<unsigned char>(d + "strand.u8", (size_t)n_rows), is_ref = slurp<unsigned char>(d + "is_ref.u8", (size_t)n_rows);
    const auto region_stop = slurp<long long>(d + "region_stop.i64", (size_t)n_regions);
    const auto labels = lines(d + "labels.txt"), paths = lines(d + "paths.txt");
    if ((int)labels.size() != n_regions || (int)paths.size() != n_regions) { std::cerr << "labels / paths\n"; return 2; }
    std::vector<const char *> lp, pp;
    for (int r = 0; r < n_regions; ++r) { lp.push_back(labels[(size_t)r].c_str()); pp.push_back(paths[(size_t)r].c_str()); }
    std::vector<unsigned char> seen((size_t)n_regions, 0);
    gfm_host::WriteJob job;
    job.W = W;
    job.n_regions = n_regions;
    job.region_stop = region_stop.data();
    job.labels = lp.data();
    job.paths = pp.data();
    job.chrom = chrom.c_str();
    job.node_paths = node_paths;
    job.seen = seen.data();
    job.threads = threads;
    gfm_host::WriteStats st;
    for (long long r0 = 0; r0 < n_rows; r0 += chunk_rows) {
        const long long n = std::min(chunk_rows, n_rows - r0);
        const gfm_host::RowChunk rc{kmers.data() + (size_t)r0 * (size_t)W, start.data() + r0, stop.data() + r0, freq.data() + r0,
                                    strand.data() + r0, is_ref.data() + r0, region.data() + r0, walk.data() + r0, n};
        std::string err;
        const int rcode = gfm_host::write_chunk(g, job, rc, st, err);
        if (rcode) { std::cerr << "write_chunk failed: " << err << "\n"; return 1; }
    }
    std::printf("rows %lld files %lld bytes %lld format_s %.4f threads %d\n", st.n_rows, st.n_files, st.bytes, st.format_s, st.threads);
    return 0;
}

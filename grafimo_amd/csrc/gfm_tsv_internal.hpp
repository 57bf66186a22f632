// gfm_tsv_internal.hpp -- what the TSV parser (tsv_ingest.cpp) shares with the streamed scan (scan_stream.cpp).
// Part of libgrafimo_hip.so.  Reference lines cited as file:line are relative to /root/reference/src/grafimo/.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "grafimo_hip.h"

// shared thread-local error slot, defined in grafimo_hip.hip
extern "C" void gfm_set_error_(const char *msg);

namespace gfm_tsv_detail {

// columns of one parsed file (row order = line order)
struct FileCols {
    std::vector<uint8_t> kmers;
    std::vector<int64_t> start, stop, freq;
    std::vector<uint8_t> strand, is_ref;
    std::vector<int32_t> local_name;       // index into names
    std::vector<std::string> names;
    std::string error;
};

// score_seqs' row handling (score_sequences.py:273-293, :305-307) for one file
void parse_file(const char *path, int W, bool skip_rev, FileCols &out);

// Parse threads worth waking: at most `requested` (<= 0: every hardware thread), one per file, one per MiB of
// text (a thread parses ~1.4 GB/s) and 96 in all (beyond that the threads' own coordination costs more than
// they add: tsv_ingest.cpp).
int pick_threads(const char *const *paths, int n_paths, int requested);

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

// gfm_workers.hpp -- the host threads behind the TSV / VCF readers and the streamed scan: one process-wide crew,
// started on first use, grown on demand and kept.  (Starting 256 threads per call cost 3-19 ms of a 7-12 ms scan
// of 2e6 rows, profiles/r02_scan_trace.txt; a kept crew is woken through one condition variable.)
// Part of libgrafimo_hip.so.
#pragma once

#include <functional>

namespace gfm_workers {

bool run_if_idle(int n, const std::function<void()> &fn);

// One use of the crew: start() hands fn to n workers and returns; wait() (also run by the destructor) returns when
// all n calls of fn have returned.  fn must not throw.  If the crew is busy with another Run (a concurrent call from
// another host thread) or cannot grow, plain threads are started for this Run instead.
class Run {
public:
    Run() = default;
    Run(const Run &) = delete;
    Run &operator=(const Run &) = delete;
    ~Run() { wait(); }
    void start(int n, std::function<void()> fn);
    void wait();

private:
    friend bool run_if_idle(int n, const std::function<void()> &fn);
    bool start_(int n, std::function<void()> fn, bool only_crew);   // start(); only_crew: false, nothing started, when the crew cannot take it
    struct Impl;
    Impl *impl_ = nullptr;
};

// start + wait; n <= 1 runs fn on the calling thread
void run(int n, const std::function<void()> &fn);

// run() for help that is only worth having when it is free: with the kept crew or not at all -- false, fn not called, when the
// crew is busy with another Run or cannot grow to n (plain threads started for a 100 us job would cost more than they save)
bool run_if_idle(int n, const std::function<void()> &fn);

}  // namespace gfm_workers

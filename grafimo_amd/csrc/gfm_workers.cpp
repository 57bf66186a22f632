// gfm_workers.cpp -- see gfm_workers.hpp.  Part of libgrafimo_hip.so.
#include "gfm_workers.hpp"

#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include <pthread.h>
#include <unistd.h>

namespace gfm_workers {

namespace {

struct Crew {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> threads;
    std::function<void()> job;
    uint64_t gen = 0;          // one per Run
    int want = 0, taken = 0, active = 0;
    bool busy = false;
    pid_t pid = 0;

    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_job.wait(lk, [&] { return gen != seen && taken < want; });
            seen = gen;              // a worker takes part in a Run once
            ++taken;
            ++active;
            lk.unlock();
            job();
            lk.lock();
            --active;
            if (active == 0 && taken == want) cv_done.notify_all();
        }
    }
};

// The crew is never destroyed: its threads sleep on its condition variable until the process ends (at most one
// per hardware thread or kMaxCrew, whichever a caller asked for: the TSV reader caps itself at 96, the VCF reader at
// the `threads` it is given).  fork(): the child holds the object but none of the threads, so it starts a crew of
// its own; pthread_atfork handlers take both mutexes around the fork so that the child never inherits one that some
// other thread of the parent held.
constexpr int kMaxCrew = 512;
std::mutex g_crew_mu;
Crew *g_crew = nullptr;

void fork_prepare()
{
    g_crew_mu.lock();
    if (g_crew) g_crew->mu.lock();
}
void fork_parent()
{
    if (g_crew) g_crew->mu.unlock();
    g_crew_mu.unlock();
}
void fork_child()
{
    if (g_crew) g_crew->mu.unlock();
    g_crew = nullptr;              // the parent's crew object stays behind (its threads do not exist here)
    g_crew_mu.unlock();
}

Crew *crew()
{
    static const int registered = pthread_atfork(fork_prepare, fork_parent, fork_child);
    (void)registered;
    std::lock_guard<std::mutex> lk(g_crew_mu);
    const pid_t me = getpid();
    if (!g_crew || g_crew->pid != me) {
        g_crew = new (std::nothrow) Crew();
        if (g_crew) g_crew->pid = me;
    }
    return g_crew;
}

}  // namespace

struct Run::Impl {
    Crew *crew = nullptr;                 // the shared crew, or ...
    std::vector<std::thread> own;         // ... threads of this Run alone
};

void Run::start(int n, std::function<void()> fn)
{
    start_(n, std::move(fn), false);
}

bool Run::start_(int n, std::function<void()> fn, bool only_crew)
{
    wait();
    if (n < 1) return true;
    impl_ = new Impl();
    Crew *c = n <= kMaxCrew ? crew() : nullptr;
    if (c) {
        std::unique_lock<std::mutex> lk(c->mu);
        if (!c->busy) {
            bool grown = true;
            while ((int)c->threads.size() < n) {
                try {
                    c->threads.emplace_back([c] { c->worker(); });
                } catch (const std::system_error &) {
                    grown = false;
                    break;
                }
            }
            if (grown) {
                c->busy = true;
                c->job = std::move(fn);
                c->want = n;
                c->taken = 0;
                ++c->gen;
                impl_->crew = c;
                lk.unlock();
                c->cv_job.notify_all();
                return true;
            }
        }
    }
    if (only_crew) {
        delete impl_;
        impl_ = nullptr;
        return false;
    }
    for (int k = 0; k < n; ++k) impl_->own.emplace_back(fn);
    return true;
}

void Run::wait()
{
    if (!impl_) return;
    if (Crew *c = impl_->crew) {
        std::unique_lock<std::mutex> lk(c->mu);
        c->cv_done.wait(lk, [&] { return c->taken == c->want && c->active == 0; });
        c->job = nullptr;
        c->want = c->taken = 0;   // late wakers of this generation find nothing to take
        c->busy = false;
    }
    for (auto &th : impl_->own)
        if (th.joinable()) th.join();
    delete impl_;
    impl_ = nullptr;
}

bool run_if_idle(int n, const std::function<void()> &fn)
{
    if (n < 1) return false;
    Run r;
    if (!r.start_(n, fn, true)) return false;
    r.wait();
    return true;
}

void run(int n, const std::function<void()> &fn)
{
    if (n <= 1) {
        fn();
        return;
    }
    Run r;
    r.start(n, fn);
    r.wait();
}

}  // namespace gfm_workers

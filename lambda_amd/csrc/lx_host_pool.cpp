// lx_host_pool.cpp -- the library's host threads (lx_host_pool.h has the design).
#include "lx_host_pool.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <sched.h>

namespace lxi
{

namespace
{

inline void relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// CPUs the cgroup grants this process (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1), rounded up; 0 = no quota
unsigned cgroup_cpus()
{
    long long quota = -1, period = 100000;
    if (FILE * f = std::fopen("/sys/fs/cgroup/cpu.max", "r"))
    {
        char q[32] = {0};
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm')
            quota = std::atoll(q);
        std::fclose(f);
    }
    else
    {
        FILE * fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        FILE * fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && (std::fscanf(fq, "%lld", &quota) != 1 || std::fscanf(fp, "%lld", &period) != 1))
            quota = -1;
        if (fq)
            std::fclose(fq);
        if (fp)
            std::fclose(fp);
    }
    if (quota <= 0 || period <= 0)
        return 0;
    return (unsigned)std::max<long long>(1, (quota + period - 1) / period);
}

} // namespace

HostPool & HostPool::instance()
{
    static HostPool p;
    return p;
}

HostPool::HostPool()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    unsigned c = sched_getaffinity(0, sizeof(set), &set) == 0 ? (unsigned)CPU_COUNT(&set) : std::thread::hardware_concurrency();
    c          = std::max(1u, c);
    if (unsigned const q = cgroup_cpus())
        c = std::min(c, q);
    granted_ = c;
    // one process per GPU: the ranks of a node share its CPUs (torch.distributed.run exports LOCAL_WORLD_SIZE)
    if (char const * lw = std::getenv("LOCAL_WORLD_SIZE"))
        local_world_ = (unsigned)std::max(1L, std::min(1024L, std::atol(lw)));
    default_width_ = std::max(1u, std::min(granted_ / local_world_, 16u));
    width_.store(default_width_); // (LX_HOST_THREADS, the measurement aid of tools/host_curve.py, is applied by lx_create: set_width)
}

HostPool::~HostPool()
{
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_.store(true);
    }
    cv_.notify_all();
    for (std::thread & t : workers_)
        t.join();
}

void HostPool::set_width(unsigned w)
{
    width_.store(w == 0 ? default_width_ : std::max(1u, std::min(w, kMaxParts)));
}

void HostPool::ensure_workers(unsigned want)
{
    // (never more workers than the granted CPUs can run beside the callers: parts beyond that are taken one after the other)
    want = std::min(want, std::max(granted_, 1u) - 1u);
    if (nworkers_.load(std::memory_order_acquire) >= want)
        return;
    std::lock_guard<std::mutex> lk(start_m_);
    while (workers_.size() < want)
    {
        unsigned const self = (unsigned)workers_.size();
        workers_.emplace_back([this, self] { worker(self); });
    }
    nworkers_.store((unsigned)workers_.size(), std::memory_order_release);
}

bool HostPool::claim(Slot & s, unsigned & id, unsigned & parts, std::function<void(unsigned)> const *& f)
{
    uint64_t c = s.ctrl.load(std::memory_order_acquire);
    for (;;)
    {
        unsigned const n = (unsigned)(c >> 16) & 0xffffu, next = (unsigned)c & 0xffffu;
        if (next >= n)
            return false;
        if (s.ctrl.compare_exchange_weak(c, c + 1, std::memory_order_acq_rel, std::memory_order_acquire))
        {
            // (this generation's job cannot end before finish() counts the part: f stays what the job's owner wrote)
            id    = next;
            parts = n;
            f     = s.f;
            if (next + 1 == n)
                pending_.fetch_sub(1);
            return true;
        }
    }
}

void HostPool::finish(Slot & s, unsigned parts)
{
    // (sequentially consistent on both sides: the caller writes `waiter` and then reads `done`, this thread counts `done` and then
    // reads `waiter` -- one of the two sees the other)
    if (s.done.fetch_add(1) + 1 == parts && s.waiter.load())
    {
        { std::lock_guard<std::mutex> lk(done_m_); } // (a caller between its test and its wait holds the mutex)
        done_cv_.notify_all();
    }
}

void HostPool::worker(unsigned self)
{
    using clock     = std::chrono::steady_clock;
    auto idle_since = clock::now();
    unsigned spins  = 0;
    for (;;)
    {
        bool did = false;
        if (pending_.load(std::memory_order_acquire) != 0)
            for (unsigned k = 0; k < kSlots; ++k)
            {
                Slot &                                s = slots_[(k + self) % kSlots]; // (workers start at different slots: concurrent jobs get their share)
                unsigned                              id, parts;
                std::function<void(unsigned)> const * f;
                while (claim(s, id, parts, f))
                {
                    (*f)(id);
                    finish(s, parts);
                    did = true;
                }
            }
        if (stop_.load())
            return;
        if (did)
        {
            idle_since = clock::now();
            spins      = 0;
            continue;
        }
        // nothing to take: keep looking while an entry point is between two of its loops, for at most 100 us; else sleep
        if (calls_.load(std::memory_order_relaxed) > 0)
        {
            relax();
            if ((++spins & 127u) != 0 || clock::now() - idle_since < std::chrono::microseconds(100))
                continue;
        }
        std::unique_lock<std::mutex> lk(m_);
        sleepers_.fetch_add(1); // (before the test: run() publishes its job and THEN looks for sleepers -- one of the two sees the other)
        cv_.wait(lk, [&] { return stop_.load() || pending_.load() != 0; });
        sleepers_.fetch_sub(1);
        lk.unlock();
        if (stop_.load())
            return;
        idle_since = clock::now();
        spins      = 0;
    }
}

void HostPool::run(unsigned nparts, std::function<void(unsigned)> const & f)
{
    if (nparts == 0)
        return;
    nparts = std::min(nparts, 0xffffu);
    Slot * mine = nullptr;
    if (nparts > 1)
    {
        ensure_workers(std::min(nparts, kMaxParts) - 1);
        if (nworkers_.load(std::memory_order_acquire) != 0)
            for (unsigned k = 0; k < kSlots && !mine; ++k)
                if (slots_[k].busy.load(std::memory_order_relaxed) == 0 && slots_[k].busy.exchange(1, std::memory_order_acquire) == 0)
                    mine = &slots_[k];
    }
    if (!mine)
    {
        for (unsigned t = 0; t < nparts; ++t) // (one part, no worker to share with, or every job slot taken)
            f(t);
        return;
    }
    Slot & s = *mine;
    s.f      = &f;
    s.done.store(0, std::memory_order_relaxed);
    uint64_t const gen = (s.ctrl.load(std::memory_order_relaxed) >> 32) + 1;
    pending_.fetch_add(1);
    s.ctrl.store(gen << 32 | (uint64_t)nparts << 16, std::memory_order_seq_cst);
    if (sleepers_.load() > 0)
    {
        { std::lock_guard<std::mutex> lk(m_); } // (a sleeper between its test and its wait holds the mutex: wait for it to be asleep)
        cv_.notify_all();
    }
    {
        unsigned                              id, parts;
        std::function<void(unsigned)> const * g;
        while (claim(s, id, parts, g))
        {
            f(id);
            s.done.fetch_add(1, std::memory_order_acq_rel);
        }
    }
    // the parts the workers took: wait for them -- looking for ~50 us (a part of a loop is tens of microseconds), then asleep, woken
    // by the worker that finishes the last part
    for (unsigned k = 0; s.done.load() != nparts; ++k)
    {
        relax();
        if (k < 2048)
            continue;
        s.waiter.store(1);
        {
            std::unique_lock<std::mutex> lk(done_m_);
            done_cv_.wait(lk, [&] { return s.done.load() == nparts; });
        }
        s.waiter.store(0);
    }
    s.f = nullptr;
    s.ctrl.store(gen << 32, std::memory_order_release); // (no parts: idle, generation kept)
    s.busy.store(0, std::memory_order_release);
}

// a few host threads for the per-extension loops of the entry points
unsigned host_threads(uint64_t n)
{
    // (a loop over fewer than ~24 000 extensions is shorter than handing it out; between that and the batch sizes the pipeline
    // is built for, one part per 24 000 -- a 3 000-query batch spent 1.3 of its 3.0 ms unpacking on one thread)
    if (n < 24000)
        return 1;
    unsigned const avail = HostPool::instance().width();
    return n >= 250000 ? avail : std::min<unsigned>(avail, std::max<unsigned>(2u, (unsigned)(n / 24000)));
}

unsigned pool_width()
{
    return HostPool::instance().width();
}

void pool_run(unsigned nparts, std::function<void(unsigned)> f)
{
    HostPool::instance().run(nparts, f);
}

} // namespace lxi

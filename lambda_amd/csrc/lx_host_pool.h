// lx_host_pool.h -- the library's host threads.  The per-extension loops of the host-buffer entry points and of the Level-2 driver
// (plans, validations, unpack loops: fifteen to twenty loops within a millisecond or two of a call) are cut into PARTS; a part is a
// unit of work that any thread may take, not a thread: the calling thread and the pool's workers claim part numbers from the job's
// counter until none is left.  So a loop is correct with any number of workers -- none, if the process may not start any --, and
// several callers (one handle per host thread, lambda's own model: /root/reference/src/search.cpp:379-385, one LocalDataHolder per
// OpenMP thread and nothing shared but statistics and the writer) run their loops side by side, each with the workers that happen
// to be free, instead of one after the other behind a mutex.
//
// Width (the parts a large loop is cut into = the threads that can work on it): min(affinity mask, cgroup CPU quota) divided by
// LOCAL_WORLD_SIZE (one process per GPU under torch.distributed.run: the ranks of a node share its CPUs), at most 16; the caller's
// LX_OPT_HOST_THREADS (include/lambda_ext.h) overrides it for the process.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lxi
{

class HostPool
{
public:
    static HostPool & instance();
    ~HostPool();

    // parts of a large loop
    unsigned width() const { return width_.load(std::memory_order_relaxed); }
    // LX_OPT_HOST_THREADS: 0 = back to the granted CPUs' share; at most kMaxParts
    void     set_width(unsigned w);
    unsigned granted_cpus() const { return granted_; } // min(affinity, cgroup quota), at least 1
    unsigned local_world() const { return local_world_; }

    // f(0) .. f(nparts - 1), each exactly once, on the caller and whatever workers are free; returns when all are done
    void run(unsigned nparts, std::function<void(unsigned)> const & f);

    // An entry point of the library is running on this thread: between its loops the workers keep looking for the next one for a
    // while (handing a loop to sleeping threads costs 30-60 us, to waiting ones about one) -- and only then: with no call in flight
    // they sleep at once.
    struct Call
    {
        Call() { instance().calls_.fetch_add(1); }
        ~Call() { instance().calls_.fetch_sub(1); }
        Call(Call const &)             = delete;
        Call & operator=(Call const &) = delete;
    };

    static constexpr unsigned kMaxParts = 64;

private:
    HostPool();
    static constexpr unsigned kSlots = 32; // jobs in flight at once (callers beyond that run their loop themselves)
    struct alignas(64) Slot
    {
        // generation << 32 | parts << 16 | next part: one word, so that a part is claimed (compare-exchange) from the very job whose
        // part count the claimer saw -- a slot that was reused meanwhile has another generation and the claim fails
        std::atomic<uint64_t>                 ctrl{0};
        std::atomic<uint32_t>                 done{0};
        std::atomic<uint32_t>                 busy{0};
        std::atomic<uint32_t>                 waiter{0};
        std::function<void(unsigned)> const * f = nullptr; // (written before ctrl publishes the job, read after a successful claim)
    };
    Slot                     slots_[kSlots];
    std::vector<std::thread> workers_;
    std::mutex               start_m_;           // workers_ grows under it
    std::mutex               m_;                 // the workers' sleep
    std::condition_variable  cv_;
    std::mutex               done_m_;            // a caller's sleep
    std::condition_variable  done_cv_;
    std::atomic<uint32_t>    pending_{0};        // jobs with unclaimed parts
    std::atomic<int>         sleepers_{0};
    std::atomic<int>         calls_{0};
    std::atomic<bool>        stop_{false};
    std::atomic<unsigned>    width_{1};
    std::atomic<unsigned>    nworkers_{0};
    unsigned                 granted_ = 1, local_world_ = 1, default_width_ = 1;

    bool claim(Slot & s, unsigned & id, unsigned & parts, std::function<void(unsigned)> const *& f);
    void finish(Slot & s, unsigned parts);
    void worker(unsigned self);
    void ensure_workers(unsigned want);
};

// parts for a loop over n items of an entry point (none below ~24 000 items: such a loop is shorter than handing it out)
unsigned host_threads(uint64_t n);

template <typename F>
inline void parallel_ranges(uint64_t n, unsigned nparts, F && body)
{
    if (nparts <= 1 || n < 2 * (uint64_t)nparts)
    {
        for (unsigned t = 0; t < nparts; ++t) // keep the per-part slots of the callers meaningful
            body(t, t == 0 ? 0 : n, n);
        return;
    }
    uint64_t const                      step = (n + nparts - 1) / nparts;
    std::function<void(unsigned)> const f    = [&body, step, n](unsigned t) { body(t, std::min(n, t * step), std::min(n, (t + 1) * step)); };
    HostPool::instance().run(nparts, f);
}

// for the Level-2 driver and the writers (host/*.cpp), which are written against the C ABI and borrow only the threads
unsigned pool_width();
void     pool_run(unsigned nparts, std::function<void(unsigned)> f);

} // namespace lxi

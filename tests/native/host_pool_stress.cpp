// host_pool_stress.cpp -- stress harness of lambda_amd/csrc/lx_host_pool.cpp (built and run by tests/test_host_pool.py, also under
// -fsanitize=thread): every part of every loop runs exactly once and is over when run() returns, with one caller and with eight
// callers at once (one handle per host thread); a pool that grows between loops (the round-5 race: run(2) x 3, then run(12)).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../lambda_amd/csrc/lx_host_pool.h"

using lxi::HostPool;

static std::atomic<long> g_fail{0}, g_by_worker{0};

static void one_loop(unsigned nparts, unsigned work)
{
    std::vector<int>  ran(nparts, 0); // (plain ints: written by whoever takes the part, read after run() -- a part still running then is a data race TSan reports)
    std::vector<long> sink(nparts, 0);
    std::thread::id const caller = std::this_thread::get_id();
    std::function<void(unsigned)> const f = [&](unsigned t)
    {
        if (std::this_thread::get_id() != caller)
            g_by_worker.fetch_add(1, std::memory_order_relaxed);
        long x = 0;
        for (unsigned k = 0; k < work; ++k)
            x += (long)k * (t + 1);
        sink[t] = x;
        ran[t] += 1;
    };
    HostPool::instance().run(nparts, f);
    for (unsigned t = 0; t < nparts; ++t)
        if (ran[t] != 1)
        {
            g_fail.fetch_add(1);
            std::fprintf(stderr, "part %u of %u ran %d times\n", t, nparts, ran[t]);
        }
}

int main(int argc, char ** argv)
{
    int const rounds = argc > 1 ? std::atoi(argv[1]) : 2000;
    unsigned const callers = argc > 2 ? (unsigned)std::atoi(argv[2]) : 8;
    // (1) the advisor's sequence, from a fresh pool state each time the pool has grown: small loops, then a wide one
    for (int r = 0; r < rounds; ++r)
    {
        one_loop(2, 50);
        one_loop(2, 50);
        one_loop(2, 50);
        one_loop(12, 50);
        one_loop(1 + (unsigned)r % 16, 10 + (unsigned)r % 200);
    }
    // (2) several callers at once, each inside an entry point (HostPool::Call: the workers look for the next loop instead of sleeping)
    std::vector<std::thread> th;
    for (unsigned c = 0; c < callers; ++c)
        th.emplace_back(
            [c, rounds]
            {
                HostPool::Call in_flight;
                for (int r = 0; r < rounds; ++r)
                    one_loop(1 + (unsigned)(r * 7 + (int)c) % 16, 20 + (unsigned)(r + (int)c) % 300);
            });
    for (auto & t : th)
        t.join();
    // (3) wider than the pool may ever be
    for (int r = 0; r < rounds / 10 + 1; ++r)
        one_loop(64, 30);
    std::printf("width %u granted %u local_world %u parts taken by workers %ld failures %ld\n", HostPool::instance().width(), HostPool::instance().granted_cpus(), HostPool::instance().local_world(),
                g_by_worker.load(), g_fail.load());
    return g_fail.load() ? 1 : 0;
}

// Development aid: times the host seeding stage alone (no GPU): g++ -O2 -std=c++17 -pthread tools/dev/seed_bench.cpp -o /tmp/seed_bench
//   /tmp/seed_bench [n_queries] [n_db] [threads]
#define LX_SEED_BUILD_TIMING 1
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../lambda_amd/csrc/host/lx_seeding.hpp"
using namespace lambda_amd;

int main(int argc, char ** argv)
{
    size_t const nq = argc > 1 ? std::atol(argv[1]) : 20000, ndb = argc > 2 ? std::atol(argv[2]) : 20000;
    unsigned const threads = argc > 3 ? (unsigned)std::atoi(argv[3]) : 1;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd(std::log(300.0), 0.6);
    uint8_t const std20[20] = {0, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18, 19, 21, 22, 23}; // ACDEFGHIKLMNPQRSTVWY in SeqAn ranks
    std::vector<uint8_t>  res, red;
    std::vector<uint64_t> off, len;
    for (size_t s = 0; s < ndb; ++s)
    {
        uint64_t const L = std::min(2000.0, std::max(50.0, std::exp(nd(rng))));
        off.push_back(res.size());
        len.push_back(L);
        for (uint64_t i = 0; i < L; ++i)
        {
            res.push_back(std20[rng() % 20]);
            red.push_back(kLi10[res.back()]);
        }
    }
    std::vector<uint8_t>  qres, qred;
    std::vector<uint64_t> qoff, qlen, which;
    for (size_t k = 0; k < nq; ++k)
    {
        qoff.push_back(qres.size());
        qlen.push_back(150);
        size_t s0 = rng() % ndb;
        while (len[s0] < 160)
            s0 = rng() % ndb;
        uint64_t const p = rng() % (len[s0] - 150);
        bool const planted = rng() % 10 < 3;
        for (int i = 0; i < 150; ++i)
        {
            uint8_t const c = (planted && rng() % 4 != 0) ? res[off[s0] + p + i] : std20[rng() % 20];
            qres.push_back(c);
            qred.push_back(kLi10[c]);
        }
        which.push_back(k);
    }
    auto const t0 = std::chrono::steady_clock::now();
    ReducedIndex ix;
    ix.build(red, off, len, 10, threads);
    auto const t1 = std::chrono::steady_clock::now();
    int8_t m[LX_ALPH * LX_ALPH];
    for (int a = 0; a < 32; ++a)
        for (int b = 0; b < 32; ++b)
            m[a * LX_ALPH + b] = a == b ? 5 : -2;
    SeedingInput in{};
    in.qRes = qres.data(), in.qRed = qred.data(), in.qOff = qoff.data(), in.qLen = qlen.data(), in.nQSeq = qoff.size();
    in.qNumFrames = 1, in.unknownRank = 25;
    in.sRes = res.data(), in.sOff = off.data(), in.sLen = len.data();
    in.alph = 10, in.matrix = m, in.maxMatches = 25, in.halfExact = true, in.adaptive = true, in.preScoring = 2, in.preScoringThresh = 2.0;
    for (SeedParams const so : {SeedParams{10, 5, 0}, SeedParams{11, 3, 1}})
    {
        std::vector<lx_match> a;
        SeedingStats          sa;
        auto const            t2 = std::chrono::steady_clock::now();
        seedQueriesParallel(ix, in, so, which, a, sa, threads);
        auto const t3 = std::chrono::steady_clock::now();
        std::printf("seed %d/%d delta %d: %.0f ms, %zu matches of %llu hits\n", so.seedLength, so.seedOffset, so.maxSeedDist,
                    std::chrono::duration<double, std::milli>(t3 - t2).count(), a.size(), (unsigned long long)sa.hitsAfterSeeding);
    }
    std::printf("table: %.0f ms (%zu residues, %u threads)\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), res.size(), threads);
    return 0;
}

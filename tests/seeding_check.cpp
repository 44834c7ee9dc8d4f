// Host-only check of lambda_amd/csrc/host/lx_seeding.hpp (compiled and run by tests/test_seeding.py with g++):
// the sorted word table against brute force over random reduced sequences -- exact words, half-exact words (first half
// exact, at most one substitution in the second half: searchHalfExactImpl, /root/reference/src/search_algo.hpp:537-604),
// one substitution anywhere (--seed-half-exact 0), words longer than the table's keys, cursor counts under extendRight, and the
// reduction tables' group structure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <tuple>

#include "../lambda_amd/csrc/host/lx_seeding.hpp"

using namespace lambda_amd;

static int fails = 0;
#define CHECK(c)                                                         \
    do                                                                   \
    {                                                                    \
        if (!(c))                                                        \
        {                                                                \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);   \
            ++fails;                                                     \
        }                                                                \
    } while (0)

int main()
{
    // Li-10: ten groups, the published ones (Li et al. 2003)
    char const * order = "ABCDEFGHIJKLMNOPQRSTUVWYZX*";
    auto g = [&](char c) { return kLi10[std::strchr(order, c) - order]; };
    std::set<int> groups;
    for (int i = 0; i < 27; ++i)
        groups.insert(kLi10[i]);
    CHECK(groups.size() == 10);
    CHECK(g('A') == g('S') && g('S') == g('T'));
    CHECK(g('F') == g('W') && g('W') == g('Y'));
    CHECK(g('L') == g('M') && g('I') == g('V') && g('I') != g('L'));
    CHECK(g('N') == g('H') && g('Q') == g('E') && g('E') == g('D') && g('R') == g('K'));
    CHECK(g('C') != g('A') && g('G') != g('A') && g('P') != g('A') && g('G') != g('P'));

    std::mt19937_64 rng(7);
    for (int alph : {10, 4})
    {
        std::vector<uint8_t>  red;
        std::vector<uint64_t> off, len;
        for (int s = 0; s < 40; ++s)
        {
            off.push_back(red.size());
            uint64_t const L = 5 + rng() % 300;
            len.push_back(L);
            for (uint64_t i = 0; i < L; ++i)
                red.push_back((uint8_t)(rng() % (alph == 4 ? 4 : 3 + rng() % 8))); // skewed: repeats occur
        }
        ReducedIndex ix;
        ix.build(red, off, len, alph);
        CHECK(ix.keyLen() == (alph == 10 ? 18 : 27));
        for (int trial = 0; trial < 300; ++trial)
        {
            int const K = alph == 10 ? 6 + (int)(rng() % 6) : 8 + (int)(rng() % 8);
            // the seed: a word of the database (mutated now and then) or random letters
            std::vector<uint8_t> seed(K);
            size_t const         s0 = rng() % off.size();
            if (len[s0] >= (uint64_t)K && rng() % 4)
            {
                uint64_t const p = rng() % (len[s0] - K + 1);
                for (int i = 0; i < K; ++i)
                    seed[i] = red[off[s0] + p + i];
                if (rng() % 2)
                    seed[K / 2 + rng() % (K - K / 2)] = (uint8_t)(rng() % alph);
            }
            else
                for (auto & c : seed)
                    c = (uint8_t)(rng() % alph);
            auto brute = [&](int maxDist)
            {
                std::set<std::pair<uint32_t, uint32_t>> hits;
                for (size_t s = 0; s < off.size(); ++s)
                    for (uint64_t p = 0; p + K <= len[s]; ++p)
                    {
                        int  d  = 0;
                        bool ok = true;
                        for (int i = 0; i < K && ok; ++i)
                            if (red[off[s] + p + i] != seed[i])
                            {
                                if (i < K / 2)
                                    ok = false; // the first half is exact
                                else
                                    ok = ++d <= maxDist;
                            }
                        if (ok)
                            hits.emplace((uint32_t)s, (uint32_t)p);
                    }
                return hits;
            };
            std::vector<ReducedIndex::Cursor> cur;
            searchExact(ix, seed.data(), K, cur);
            std::set<std::pair<uint32_t, uint32_t>> got;
            uint64_t                                cnt = 0;
            for (auto const & c : cur)
            {
                cnt += c.count();
                ix.locate(c, [&](uint32_t s, uint32_t p) { got.emplace(s, p); });
            }
            CHECK(got == brute(0) && cnt == got.size());
            cur.clear();
            got.clear();
            cnt = 0;
            searchHalfExact(ix, seed.data(), K, 1, alph, cur);
            for (auto const & c : cur)
            {
                cnt += c.count();
                ix.locate(c, [&](uint32_t s, uint32_t p) { got.emplace(s, p); });
            }
            CHECK(got == brute(1) && cnt == got.size()); // (every hit under exactly one cursor)
            // --seed-half-exact 0: one substitution anywhere in the seed (search_one_error over the whole word, :486-531)
            cur.clear();
            got.clear();
            cnt = 0;
            searchHalfExact(ix, seed.data(), K, 1, alph, cur, 0);
            for (auto const & c : cur)
            {
                cnt += c.count();
                ix.locate(c, [&](uint32_t s, uint32_t p) { got.emplace(s, p); });
            }
            std::set<std::pair<uint32_t, uint32_t>> hamming1;
            for (size_t s2 = 0; s2 < off.size(); ++s2)
                for (uint64_t p = 0; p + K <= len[s2]; ++p)
                {
                    int d = 0;
                    for (int i = 0; i < K; ++i)
                        d += red[off[s2] + p + i] != seed[i];
                    if (d <= 1)
                        hamming1.emplace((uint32_t)s2, (uint32_t)p);
                }
            CHECK(got == hamming1 && cnt == got.size());
        }
        // words LONGER than the table's keys (adaptive elongation goes on to the read's end, :703-721): a repeat of 60 letters
        // planted in three sequences; the cursor of its first keyLen + 20 letters must hold exactly the occurrences brute force finds
        {
            std::vector<uint8_t> rep(60);
            for (auto & c : rep)
                c = (uint8_t)(rng() % alph);
            for (size_t s2 : {3u, 11u, 29u})
                if (len[s2] >= 70)
                    for (int i = 0; i < 60; ++i)
                        red[off[s2] + 5 + i] = rep[i];
            ix.build(red, off, len, alph);
            int const K = ix.keyLen() + 20;
            ReducedIndex::Cursor c = ix.root();
            for (int i = 0; i < K && !c.empty(); ++i)
                c = ix.extendRight(c, rep[i]);
            std::set<std::pair<uint32_t, uint32_t>> got, want;
            ix.locate(c, [&](uint32_t s2, uint32_t p) { got.emplace(s2, p); });
            for (size_t s2 = 0; s2 < off.size(); ++s2)
                for (uint64_t p = 0; p + K <= len[s2]; ++p)
                    if (std::memcmp(&red[off[s2] + p], rep.data(), K) == 0)
                        want.emplace((uint32_t)s2, (uint32_t)p);
            CHECK(!want.empty() && got == want && c.count() == want.size());
            ReducedIndex::Cursor const dead = ix.extendRight(c, (uint8_t)((rep[K] + 1) % alph)); // a letter no occurrence continues with
            std::set<std::pair<uint32_t, uint32_t>> none;
            ix.locate(dead, [&](uint32_t s2, uint32_t p) { none.emplace(s2, p); });
            for (auto const & hp : none) // (whatever survives must really continue with that letter)
                CHECK(red[off[hp.first] + hp.second + K] == (uint8_t)((rep[K] + 1) % alph));
        }
    }
    // the table and the seeding on several host threads: the same cursors and the same match list as one thread gives
    {
        int const             alph = 10;
        std::vector<uint8_t>  red, res;
        std::vector<uint64_t> off, len;
        for (int s = 0; s < 300; ++s)
        {
            off.push_back(red.size());
            uint64_t const L = (s % 37 == 0) ? 0 : 5 + rng() % 400; // (empty sequences too)
            len.push_back(L);
            for (uint64_t i = 0; i < L; ++i)
            {
                res.push_back((uint8_t)(rng() % 20));
                red.push_back(kLi10[res.back()]);
            }
        }
        ReducedIndex one, many;
        one.build(red, off, len, alph, 1);
        many.build(red, off, len, alph, 5);
        for (int trial = 0; trial < 200; ++trial)
        {
            size_t const s0 = rng() % off.size();
            if (len[s0] < 12)
                continue;
            uint64_t const       p = rng() % (len[s0] - 11);
            ReducedIndex::Cursor a = one.root(), b = many.root();
            for (int i = 0; i < 11 && !a.empty(); ++i)
            {
                a = one.extendRight(a, red[off[s0] + p + i]);
                b = many.extendRight(b, red[off[s0] + p + i]);
                CHECK(a.lo == b.lo && a.hi == b.hi);
            }
            std::vector<std::pair<uint32_t, uint32_t>> ha, hb;
            one.locate(a, [&](uint32_t s, uint32_t q) { ha.emplace_back(s, q); });
            many.locate(b, [&](uint32_t s, uint32_t q) { hb.emplace_back(s, q); });
            CHECK(ha == hb && std::is_sorted(ha.begin(), ha.end())); // (equal words: by sequence, then position)
        }
        // reads: mutated pieces of the database, two frames per read (the second one the same letters: what matters is the reset)
        std::vector<uint8_t>  qres, qred;
        std::vector<uint64_t> qoff, qlen, which;
        for (int r = 0; r < 400; ++r)
        {
            size_t s0 = rng() % off.size();
            while (len[s0] < 60)
                s0 = rng() % off.size();
            uint64_t const p = rng() % (len[s0] - 50);
            for (int frame = 0; frame < 2; ++frame)
            {
                qoff.push_back(qres.size());
                qlen.push_back(r % 29 == 0 ? 6 : 50); // (reads too short for a seed are skipped)
                for (uint64_t i = 0; i < qlen.back(); ++i)
                {
                    uint8_t const c = (i % 11 == 7) ? (uint8_t)((res[off[s0] + p + i] + 1 + r % 3) % 20) : res[off[s0] + p + i];
                    qres.push_back(c);
                    qred.push_back(kLi10[c]);
                }
                which.push_back(qoff.size() - 1);
            }
        }
        int8_t m[LX_ALPH * LX_ALPH];
        for (int a = 0; a < 32; ++a)
            for (int b = 0; b < 32; ++b)
                m[a * LX_ALPH + b] = a == b ? 5 : -2;
        SeedingInput in{};
        in.qRes = qres.data(), in.qRed = qred.data(), in.qOff = qoff.data(), in.qLen = qlen.data(), in.nQSeq = qoff.size();
        in.qNumFrames = 2, in.unknownRank = 25;
        in.sRes = res.data(), in.sOff = off.data(), in.sLen = len.data();
        in.alph = alph, in.matrix = m, in.maxMatches = 25, in.halfExact = true, in.adaptive = true, in.preScoring = 2, in.preScoringThresh = 2.0;
        for (SeedParams const so : {SeedParams{10, 5, 0}, SeedParams{11, 3, 1}})
        {
            std::vector<lx_match> a, b;
            SeedingStats          sa, sb;
            seedQueries(one, in, so, which, a, sa);
            seedQueriesParallel(many, in, so, which, b, sb, 7);
            CHECK(!a.empty() && a.size() == b.size() && sa.hitsAfterSeeding == sb.hitsAfterSeeding && sa.hitsFailedPreExtendTest == sb.hitsFailedPreExtendTest);
            CHECK(a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(lx_match)) == 0);
        }
    }
    // seedLooksPromising: the planted diagonal passes, a random one does not
    {
        int8_t m[LX_ALPH * LX_ALPH];
        for (int a = 0; a < 32; ++a)
            for (int b = 0; b < 32; ++b)
                m[a * LX_ALPH + b] = a == b ? 5 : -4;
        std::vector<uint8_t> q(60), s(200);
        for (auto & c : q)
            c = (uint8_t)(rng() % 20);
        for (auto & c : s)
            c = (uint8_t)(rng() % 20);
        for (int i = 0; i < 40; ++i)
            s[100 + i] = q[10 + i];
        lx_match const good{0, 0, 20, 30, 110, 120}, bad{0, 0, 20, 30, 15, 25};
        CHECK(seedLooksPromising(q.data(), q.size(), s.data(), s.size(), good, 10, 2, 2.0, m));
        CHECK(!seedLooksPromising(q.data(), q.size(), s.data(), s.size(), bad, 10, 2, 2.0, m));
        lx_match const edge{0, 0, 0, 10, 0, 10}; // clipped at both sequence starts: no out-of-range read
        (void)seedLooksPromising(q.data(), q.size(), s.data(), s.size(), edge, 10, 2, 2.0, m);
    }
    std::printf(fails ? "seeding check: %d failure(s)\n" : "seeding check: ok\n", fails);
    return fails ? 1 : 0;
}

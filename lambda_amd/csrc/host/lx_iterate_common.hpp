// lx_iterate_common.hpp -- what the two forms of the Level-2 driver share: lx_iterate_matches on a host match list
// (lx_driver.cpp) and lx_iterate_matches_dev on a device match list (../lx_level2_host.cpp).  The part of iterateMatchesFullSimd
// (/root/reference/src/search_algo.hpp:1177-1332) that comes AFTER the two DP passes: the filter's statistics (:1260, :1274),
// the survivors' order (:1229-1235, :1299), _expandAlign's coordinates (:1032-1035), the identity cut-off (:1310-1315), bit
// score and e-value (:1318-1322) -- and the filter itself as an integer cut-off per query length.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include <vector>

#include <sys/mman.h>

#include "../../../include/lambda_ext.h"
#include "blast_stats.hpp"

#include "../lx_host_pool.h" // the library's host threads: parts per loop, and f(0) ... f(nparts - 1) taken by whatever threads are free

namespace lambda_amd
{
// Blocks of result memory kept between calls: a result of a million HSPs is 96 MB of records and 110 MB of alignment columns, and a
// block the allocator has just mapped costs a page fault per 4 KB when the threads write it (measured: 7 ms of a 22-ms call) -- a
// block a freed result hands back is mapped already.  Process-wide, a handful of blocks, 1 GiB at most; everything else is free()d.
struct BlockCache
{
    static constexpr size_t kMinBytes = 1u << 20, kMaxBlocks = 6, kMaxTotal = 1u << 30;
    std::mutex mu;
    struct Block
    {
        void * p;
        size_t bytes;
    };
    std::vector<Block> blocks;
    size_t             total = 0;
    ~BlockCache()
    {
        for (Block const & b : blocks)
            std::free(b.p);
    }
    void * take(size_t bytes, size_t & got) // the smallest kept block that holds `bytes`, or nullptr
    {
        std::lock_guard<std::mutex> lock(mu);
        size_t                      best = blocks.size();
        for (size_t k = 0; k < blocks.size(); ++k)
            if (blocks[k].bytes >= bytes && (best == blocks.size() || blocks[k].bytes < blocks[best].bytes))
                best = k;
        if (best == blocks.size() || blocks[best].bytes > 4 * bytes + (64u << 20))
            return nullptr;
        void * const p = blocks[best].p;
        got            = blocks[best].bytes;
        total -= got;
        blocks.erase(blocks.begin() + (long)best);
        return p;
    }
    void give(void * p, size_t bytes)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            if (bytes >= kMinBytes && blocks.size() < kMaxBlocks && total + bytes <= kMaxTotal)
            {
                blocks.push_back(Block{p, bytes});
                total += bytes;
                return;
            }
        }
        std::free(p);
    }
};
inline BlockCache & block_cache()
{
    // (never destroyed: a result freed from a static destructor or an atexit handler still finds it; lx_trim_result_cache() gives the
    // blocks back before that)
    static BlockCache * const cache = new BlockCache();
    return *cache;
}
inline size_t trim_block_cache() // frees every kept block; returns the bytes released
{
    BlockCache &                cache = block_cache();
    std::vector<BlockCache::Block> mine;
    size_t                      bytes = 0;
    {
        std::lock_guard<std::mutex> lock(cache.mu);
        mine.swap(cache.blocks);
        bytes       = cache.total;
        cache.total = 0;
    }
    for (BlockCache::Block const & b : mine)
        std::free(b.p);
    return bytes;
}

// A growing array of trivially copyable records that is never value-initialised: every byte of a result is written by the threads
// that make it -- a std::vector would first zero all of it on the calling thread.
template <class T>
struct RawVec
{
    T *    p   = nullptr;
    size_t n   = 0, cap = 0;
    RawVec()   = default;
    RawVec(RawVec const &)             = delete;
    RawVec & operator=(RawVec const &) = delete;
    ~RawVec()
    {
        if (p)
            block_cache().give(static_cast<void *>(p), cap * sizeof(T));
    }
    size_t    size() const { return n; }
    T *       data() { return p; }
    T const * data() const { return p; }
    T *       begin() { return p; }
    T *       end() { return p + n; }
    T &       operator[](size_t i) { return p[i]; }
    bool      resize(size_t m) // false: out of memory (the contents stay)
    {
        if (m > cap)
        {
            size_t const want = std::max(m, cap + cap / 2), bytes = std::max<size_t>(want, 1) * sizeof(T);
            if (!p && bytes >= BlockCache::kMinBytes)
            {
                size_t       got  = 0;
                void * const kept = block_cache().take(bytes, got);
                if (kept)
                {
                    p   = static_cast<T *>(kept);
                    cap = got / sizeof(T);
                    n   = m;
                    return true;
                }
            }
            void * const np = std::realloc(static_cast<void *>(p), bytes);
            if (!np)
                return false;
#ifdef MADV_HUGEPAGE
            if (bytes >= (8u << 20)) // (a fresh block: 2 MiB pages where the system grants them -- 512 times fewer faults)
            {
                // whole pages INSIDE the block only: [align_up(np), align_down(np + bytes))
                uintptr_t const a = (reinterpret_cast<uintptr_t>(np) + 4095) & ~(uintptr_t)4095, b = (reinterpret_cast<uintptr_t>(np) + bytes) & ~(uintptr_t)4095;
                if (b > a)
                    (void)madvise(reinterpret_cast<void *>(a), b - a, MADV_HUGEPAGE);
            }
#endif
            p   = static_cast<T *>(np);
            cap = want;
        }
        n = m;
        return true;
    }
};
} // namespace lambda_amd

struct lx_iterate_result
{
    lambda_amd::RawVec<lx_blast_match> matches;
    lambda_amd::RawVec<uint8_t>        ops;
    lx_iterate_stats                   stats{};
};

// lx_level2_host.cpp: lx_iterate_matches' large lists go to the Level-2 kernels -- the sequence sets become resident (once: a later
// call with the same sets finds them there), the matches go up as sort words (16 bytes each instead of 48), the window list comes
// back into `matches` (the span the reference shrinks).  Returns kNotTaken when the call is not one the device path serves (small
// lists, band mode, subjects that are not resident): the caller then runs the host form.
namespace lxi
{
constexpr int kNotTaken = 1;
int iterate_host_list_on_device(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint64_t const * q_seq_off, uint64_t const * q_seq_len,
                                uint64_t n_qseq, uint64_t const * q_orig_len, uint8_t const * s_res, uint64_t s_bytes, uint64_t const * s_seq_off,
                                uint64_t const * s_seq_len, uint64_t n_sseq, lx_match * matches, uint64_t n_matches, lx_search_params const * params,
                                lx_iterate_result * res);
} // namespace lxi

namespace lambda_amd
{

// The reference calls iterateMatches per thread on a block of <= 10 queries (src/search_options.hpp:71); a GPU wants the seed
// lists of thousands of queries per call, and then this function's own loops (a sort of the list, the slices, the records) cost
// as much as the kernels unless they are spread over the library's host threads (lx_host.cpp's pool).
inline constexpr uint64_t kParallelFrom = 32768; // list sizes below this stay on the calling thread
template <typename F>
inline void parallelRanges(uint64_t n, F && body) // body(thread, lo, hi) over a partition of [0, n)
{
    unsigned const nt = n >= kParallelFrom ? std::max(1u, lxi::pool_width()) : 1u;
    if (nt <= 1)
    {
        body(0u, (uint64_t)0, n);
        return;
    }
    uint64_t const step = (n + nt - 1) / nt;
    lxi::pool_run(nt, [&](unsigned t) { body(t, std::min(n, t * step), std::min(n, (t + 1) * step)); });
}

// The filter of :1251-1283 keeps a candidate iff bitScore >= minBitScore and eValue <= maxEValue.  Both are monotone in the raw
// score for a given query length (src/search_misc.hpp:77-78), so the smallest passing score per query length -- found by
// bisection over the very double formulas -- gives identical decisions as an integer test on the device.
struct CutOffs
{
    lx_search_params const *              params;
    EValueContext                         evalue;
    std::unordered_map<uint64_t, int32_t> byLength;
    explicit CutOffs(lx_search_params const * p) : params(p), evalue{p->karlin, p->db_total_length, p->query_translated != 0, {}} {}
    bool passes(int32_t score, uint64_t qLength)
    {
        if (params->min_bitscore >= 0 && computeBitScore(score, params->karlin) < params->min_bitscore)
            return false;
        if (params->max_evalue >= 0 && evalue(score, qLength) > params->max_evalue)
            return false;
        return true;
    }
    int32_t operator()(uint64_t qLength)
    {
        auto it = byLength.find(qLength);
        if (it != byLength.end())
            return it->second;
        int32_t const top = 1 << 30;
        int32_t       cut = 0x7fffffff;
        if (passes(0, qLength))
            cut = 0;
        else if (passes(top, qLength))
        {
            int32_t lo = 0, hi = top; // passes(lo) false, passes(hi) true
            while (hi - lo > 1)
            {
                int32_t const mid = lo + (hi - lo) / 2;
                (passes(mid, qLength) ? hi : lo) = mid;
            }
            cut = hi;
        }
        byLength.emplace(qLength, cut);
        return cut;
    }
};

// one DP window as the records need it
struct WindowView
{
    uint64_t qryId, subjId, qryStart, subjStart; // frame-expanded ids; where the slices begin in their sequences
    uint32_t qLen, sLen;                         // the slices' lengths
    uint64_t qLength;                            // the query length of the e-value (bm.qLength, :1213)
};

// Everything after the extension, for a list of n windows whose n_qid = qryId / qFrames never descends (both drivers hand
// over lists sorted by Match's order): `scores` of all windows, the survivors as lx_extend_batch_list returns them.  Appends
// to *res.  The reference's two stable sorts (:1229-1235 by lengths, :1299 by query) leave the survivors ordered by
// (n_qid, query slice length, subject slice length, list position).
template <class GetWindow, class GetMin>
inline int finishSurvivors(uint64_t n, GetWindow && window, int32_t const * scores, GetMin && minScore, lx_survivor_list const & list,
                            lx_search_params const * params, lx_iterate_result * res)
{
    int const qFrames = std::max(1, params->qry_num_frames), sFrames = std::max(1, params->sbj_num_frames);
    unsigned const nt = n >= kParallelFrom ? std::max(1u, lxi::pool_width()) : 1u;
    static bool const timing = std::getenv("LX_HOST_TIMING") != nullptr; // (development aid: where this function's time goes)
    auto              tlast  = std::chrono::steady_clock::now();
    std::string       tline;
    auto mark = [&](char const * what)
    {
        if (!timing)
            return;
        auto const now = std::chrono::steady_clock::now();
        char       buf[64];
        std::snprintf(buf, sizeof(buf), " %s %.1f", what, std::chrono::duration<double, std::milli>(now - tlast).count());
        tline += buf;
        tlast = now;
    };
    // where match i stands in the survivor list
    std::vector<uint32_t> listAt(n);
    parallelRanges(n, [&](unsigned, uint64_t lo, uint64_t hi) { std::fill(listAt.begin() + lo, listAt.begin() + hi, 0xffffffffu); });
    parallelRanges(list.count, [&](unsigned, uint64_t lo, uint64_t hi)
                   {
                       for (uint64_t k = lo; k < hi; ++k)
                           listAt[list.index[k]] = (uint32_t)k;
                   });
    mark("listAt");
    // the filter's statistics (:1260, :1274) from the scores of pass 1; the ranges are cut where n_qid changes
    std::vector<uint64_t> cut(nt + 1, n), nSurv(nt + 1, 0), failBit(nt, 0), failEv(nt, 0);
    std::vector<uint8_t>  descends(nt, 0);
    cut[0] = 0;
    for (unsigned t = 1; t < nt; ++t)
    {
        uint64_t c = std::max(cut[t - 1], n * t / nt);
        while (c < n && c > 0 && window(c).qryId / qFrames == window(c - 1).qryId / qFrames)
            ++c;
        cut[t] = c;
    }
    auto overRanges = [&](auto && body)
    {
        if (nt <= 1)
            body(0u, (uint64_t)0, n);
        else
            lxi::pool_run(nt, [&](unsigned t) { body(t, cut[t], cut[t + 1]); });
    };
    overRanges([&](unsigned t, uint64_t lo, uint64_t hi)
               {
                   uint64_t s = 0, fb = 0, fe = 0;
                   bool     bad = false;
                   for (uint64_t i = lo; i < hi; ++i)
                   {
                       if (i > lo && window(i).qryId / qFrames < window(i - 1).qryId / qFrames)
                           bad = true;
                       if (listAt[i] != 0xffffffffu)
                           ++s;
                       else if (scores[i] >= minScore(i))
                           ; // (an empty window: score 0 against a cut-off of 0 -- nothing to trace)
                       else if (params->min_bitscore >= 0 && computeBitScore(scores[i], params->karlin) < params->min_bitscore)
                           ++fb;
                       else
                           ++fe;
                   }
                   nSurv[t + 1] = s;
                   failBit[t]   = fb;
                   failEv[t]    = fe;
                   descends[t]  = bad;
               });
    bool sorted = true;
    for (unsigned t = 0; t < nt; ++t)
    {
        res->stats.failed_bitscore += failBit[t];
        res->stats.failed_evalue += failEv[t];
        nSurv[t + 1] += nSurv[t];
        sorted = sorted && !descends[t];
        if (t > 0 && cut[t] > 0 && cut[t] < n && window(cut[t]).qryId / qFrames < window(cut[t] - 1).qryId / qFrames)
            sorted = false;
    }
    mark("statistics");
    uint64_t const ns = nSurv[nt];
    if (ns == 0)
        return LX_OK;
    res->stats.num_ext_ali += ns; // :1287
    std::vector<uint32_t> surv(ns); // indices into the window list
    auto const byLengths = [&](uint32_t a, uint32_t b)
    {
        WindowView const x = window(a), y = window(b);
        return std::make_tuple(x.qLen, x.sLen, a) < std::make_tuple(y.qLen, y.sLen, b);
    };
    overRanges([&](unsigned t, uint64_t lo, uint64_t hi)
               {
                   uint64_t o = nSurv[t];
                   for (uint64_t i = lo; i < hi; ++i)
                       if (listAt[i] != 0xffffffffu)
                           surv[o++] = (uint32_t)i;
                   if (!sorted)
                       return;
                   // inside a query's survivors: by the lengths of the slices, then by list position
                   for (uint64_t a = nSurv[t]; a < o;)
                   {
                       uint64_t const id = window(surv[a]).qryId / qFrames;
                       uint64_t       b  = a + 1;
                       while (b < o && window(surv[b]).qryId / qFrames == id)
                           ++b;
                       if (b - a > 1)
                           std::sort(surv.begin() + a, surv.begin() + b, byLengths);
                       a = b;
                   }
               });
    if (!sorted) // (a caller's list in another order: the general sort)
        std::sort(surv.begin(), surv.end(),
                  [&](uint32_t a, uint32_t b)
                  {
                      WindowView const x = window(a), y = window(b);
                      return std::make_tuple(x.qryId / qFrames, x.qLen, x.sLen, a) < std::make_tuple(y.qryId / qFrames, y.qLen, y.sLen, b);
                  });

    mark("order");
    // compute the rest of the match properties (:1302-1325).  Two passes over the survivors, each spread over the host threads:
    // which of them pass the identity cut-off (:1310-1315) and how many columns they have, then -- the offsets known -- the
    // records and their ops, written where they stay.
    std::vector<uint8_t>  keep(ns);
    EValueContext const   evalue{params->karlin, params->db_total_length, params->query_translated != 0, {}};
    double const          logK = std::log(params->karlin.K), log2 = std::log(2.0);
    unsigned const        nts = ns >= kParallelFrom ? std::max(1u, lxi::pool_width()) : 1u;
    std::vector<uint64_t> keptOf(nts + 1, 0), opsOf(nts + 1, 0);
    uint64_t const        stepS = (ns + nts - 1) / nts;
    auto overSurvivors = [&](auto && body)
    {
        if (nts <= 1)
            body(0u, (uint64_t)0, ns);
        else
            lxi::pool_run(nts, [&](unsigned t) { body(t, std::min(ns, t * stepS), std::min(ns, (t + 1) * stepS)); });
    };
    bool const wantOps = !(params->flags & LX_ITERATE_NO_OPS);
    auto identityOf = [](lx_hsp const & a) { return a.n_ops ? (float)(100.0 * static_cast<float>(a.num_matches) / static_cast<float>(a.n_ops)) : 0.0f; };
    overSurvivors([&](unsigned t, uint64_t lo, uint64_t hi)
                  {
                      uint64_t nk = 0, no = 0;
                      for (uint64_t x = lo; x < hi; ++x)
                      {
                          if (x + 16 < hi) // (the survivors stand in the list in the order the device finished them: every record a cache miss)
                              __builtin_prefetch(&list.hsp[listAt[surv[x + 16]]]);
                          lx_hsp const & a = list.hsp[listAt[surv[x]]];
                          keep[x]          = !(identityOf(a) < params->id_cutoff);
                          nk += keep[x];
                          no += (keep[x] && wantOps) ? (uint64_t)a.n_ops : 0;
                      }
                      keptOf[t + 1] = nk;
                      opsOf[t + 1]  = no;
                  });
    for (unsigned t = 0; t < nts; ++t)
    {
        keptOf[t + 1] += keptOf[t];
        opsOf[t + 1] += opsOf[t];
    }
    mark("identity");
    uint64_t const rec0 = res->matches.size(), ops0 = res->ops.size(), nkeep = keptOf[nts], nops = opsOf[nts];
    res->stats.failed_identity += ns - nkeep;
    if (!res->matches.resize(rec0 + nkeep) || !res->ops.resize(ops0 + nops))
        return LX_ENOMEM;
    mark("allocate");
    overSurvivors([&](unsigned t, uint64_t lo, uint64_t hi)
                  {
                      EValueContext ev = evalue; // (its cache of length adjustments is not shared)
                      uint64_t      r = rec0 + keptOf[t], o = ops0 + opsOf[t];
                      for (uint64_t x = lo; x < hi; ++x)
                      {
                          if (!keep[x])
                              continue;
                          if (x + 16 < hi)
                          {
                              uint32_t const ahead = listAt[surv[x + 16]];
                              __builtin_prefetch(&list.hsp[ahead]);
                              __builtin_prefetch(&list.codes_off[ahead]);
                          }
                          if (x + 4 < hi && wantOps)
                              __builtin_prefetch(list.codes + list.codes_off[listAt[surv[x + 4]]]);
                          uint32_t const   k  = surv[x];
                          uint32_t const   at = listAt[k];
                          WindowView const m  = window(k);
                          lx_hsp const &   a  = list.hsp[at];
                          lx_blast_match   bm{};
                          bm.qry_id  = m.qryId;
                          bm.subj_id = m.subjId;
                          bm.n_qid   = m.qryId / qFrames;
                          bm.n_sid   = m.subjId / sFrames;
                          {
                              int32_t qf = 0, sf = 0; // _setFrames, :1223
                              lx_set_frames(params->q_frame_mode, params->s_frame_mode, m.qryId, m.subjId, &qf, &sf);
                              bm.q_frame = (int16_t)qf;
                              bm.s_frame = (int16_t)sf;
                          }
                          // _expandAlign: positions relative to the infix become positions in the sequence (:1032-1035)
                          bm.q_start = m.qryStart + a.q_begin;
                          bm.q_end   = m.qryStart + a.q_end;
                          bm.s_start = m.subjStart + a.s_begin;
                          bm.s_end   = m.subjStart + a.s_end;
                          bm.score   = a.score;
                          bm.alignment_length   = a.n_ops;
                          bm.num_matches        = a.num_matches;
                          bm.num_mismatches     = a.num_mismatches;
                          bm.num_positives      = a.num_positives;
                          bm.num_gap_opens      = a.num_gap_opens;
                          bm.num_gap_extensions = a.num_gap_extensions;
                          bm.identity           = identityOf(a);
                          // the reference keeps the values of the filter where it computed them and computes the others now
                          // (:1318-1322): the same formulas on the same score either way
                          bm.bit_score = (params->karlin.lambda * (double)a.score - logK) / log2; // (computeBitScore's expression, its two logarithms taken once)
                          bm.e_value   = ev(a.score, m.qLength);
                          bm.n_ops     = (uint32_t)a.n_ops;
                          bm.ops_off   = wantOps ? o : 0;
                          if (wantOps)
                          {
                              (void)lx_expand_ops(list.codes + list.codes_off[at], a.n_ops, res->ops.data() + o);
                              o += bm.n_ops;
                          }
                          res->matches[r++] = bm;
                      }
                  });
    mark("records");
    if (timing)
        std::fprintf(stderr, "[lx host ms]   finishSurvivors (%llu windows, %llu survivors):%s\n", (unsigned long long)n, (unsigned long long)ns, tline.c_str());
    return LX_OK;
}

} // namespace lambda_amd

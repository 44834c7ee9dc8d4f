// lx_driver.cpp -- C++ host mirror of the reference's extension driver, exported through the C ABI.
//
// Restates, above the two GPU passes, what /root/reference/src/search_algo.hpp does around them:
//   _widenMatch                  :919-938     (lambda_amd::widenMatch)
//   _widenAndPreprocessMatches   :1136-1175   (lambda_amd::widenAndPreprocessMatches)
//   iterateMatchesFullSimd       :1177-1332   (lambda_amd::iterateMatchesFullSimd -> lx_iterate_matches)
//   _expandAlign (coordinates)   :1032-1035
// The DP itself never runs here: both passes go to the GPU through lx_extend_batch.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <tuple>
#include <vector>

#include "blast_stats.hpp"
#include "lambda_ext.hpp"
#include "lx_iterate_common.hpp"
#include "scoring_tables.hpp"

namespace lambda_amd
{

// src/search_misc.hpp:46-50
inline int64_t bandSize(uint64_t const seqLength)
{
    return static_cast<int64_t>(std::sqrt(seqLength)) + 1;
}

// src/search_algo.hpp:919-938
inline void widenMatch(lx_match & m, uint64_t const qLen, uint64_t const sLen)
{
    m.subjStart         = (m.subjStart < m.qryStart) ? 0 : m.subjStart - m.qryStart;
    m.qryStart          = 0;
    m.qryEnd            = qLen;
    uint64_t const band = bandSize(qLen);
    m.subjEnd           = std::min<uint64_t>(m.subjStart + qLen + band, sLen);
    m.subjStart         = (band < m.subjStart) ? m.subjStart - band : 0;
}

inline auto tie(lx_match const & m)
{
    return std::tie(m.qryId, m.subjId, m.qryStart, m.qryEnd, m.subjStart, m.subjEnd);
}

// src/search_algo.hpp:1136-1175; returns the new size, *duplicates gets the number removed (hitsDuplicate)
inline uint64_t widenAndPreprocessMatches(lx_match * m, uint64_t n, uint64_t const * qLens, uint64_t const * sLens,
                                          uint64_t * duplicates)
{
    uint64_t const before = n;
    auto const     less   = [](lx_match const & a, lx_match const & b) { return tie(a) < tie(b); };
    // A list that comes grouped by query (what seeding emits) is cut at query boundaries into one piece per host thread: the
    // sort key begins with the query, the merge only ever joins windows of one (query, subject) pair, so the pieces are
    // independent and the result is the serial one.
    unsigned const nt = n >= kParallelFrom ? std::max(1u, lxi::pool_width()) : 1u;
    bool           grouped = nt > 1;
    for (uint64_t i = 1; i < n && grouped; ++i)
        grouped = m[i - 1].qryId <= m[i].qryId;
    // A list in any other order (the GPU seeding stage emits the matches in the order its lanes made them) is dealt to buckets of
    // consecutive queries first -- per-thread counts, one scatter into a second array --, and the pieces are runs of whole buckets:
    // each still holds every match of its queries, which is all the pieces need (they sort themselves).
    std::vector<lx_match> dealt;
    std::vector<uint64_t> cut(nt + 1, n), kept(nt, 0);
    cut[0] = 0;
    if (nt > 1 && !grouped)
    {
        uint64_t maxQ = 0;
        {
            std::vector<uint64_t> tmax(nt, 0);
            uint64_t const        step = (n + nt - 1) / nt;
            lxi::pool_run(nt,
                          [&](unsigned t)
                          {
                              uint64_t mx = 0;
                              for (uint64_t i = std::min(n, t * step); i < std::min(n, (t + 1) * step); ++i)
                                  mx = std::max(mx, m[i].qryId);
                              tmax[t] = mx;
                          });
            for (uint64_t v : tmax)
                maxQ = std::max(maxQ, v);
        }
        uint64_t const nb = std::min<uint64_t>(maxQ + 1, 4096), perBucket = (maxQ + nb) / nb; // bucket of query q: q / perBucket
        std::vector<std::vector<uint64_t>> at(nt, std::vector<uint64_t>(nb + 1, 0));
        uint64_t const                     step = (n + nt - 1) / nt;
        lxi::pool_run(nt,
                      [&](unsigned t)
                      {
                          for (uint64_t i = std::min(n, t * step); i < std::min(n, (t + 1) * step); ++i)
                              ++at[t][m[i].qryId / perBucket];
                      });
        std::vector<uint64_t> bucketAt(nb + 1, 0);
        uint64_t              o = 0;
        for (uint64_t b = 0; b < nb; ++b)
        {
            bucketAt[b] = o;
            for (unsigned t = 0; t < nt; ++t)
            {
                uint64_t const c = at[t][b];
                at[t][b]         = o;
                o += c;
            }
        }
        bucketAt[nb] = o;
        dealt.resize(n);
        lxi::pool_run(nt,
                      [&](unsigned t)
                      {
                          for (uint64_t i = std::min(n, t * step); i < std::min(n, (t + 1) * step); ++i)
                              dealt[at[t][m[i].qryId / perBucket]++] = m[i];
                      });
        for (unsigned t = 1; t < nt; ++t) // the bucket boundary at or behind the even share
            cut[t] = std::max(cut[t - 1], *std::lower_bound(bucketAt.begin(), bucketAt.end(), n * t / nt));
        grouped = true;
    }
    else if (grouped)
    {
        for (unsigned t = 1; t < nt; ++t)
        {
            uint64_t c = std::max(cut[t - 1], n * t / nt);
            while (c < n && c > 0 && m[c].qryId == m[c - 1].qryId)
                ++c;
            cut[t] = c;
        }
    }
    if (grouped)
    {
        lx_match * const dst = m;
        if (!dealt.empty())
            m = dealt.data(); // (the pieces are made in the second array and close ranks in the caller's)
        lxi::pool_run(nt,
                      [&](unsigned t)
                      {
                          uint64_t const lo = cut[t], hi = cut[t + 1];
                          for (uint64_t i = lo; i < hi; ++i)
                              widenMatch(m[i], qLens[m[i].qryId], sLens[m[i].subjId]);
                          std::sort(m + lo, m + hi, less);
                          for (uint64_t i = lo; i + 1 < hi; ++i)
                          {
                              lx_match & l = m[i];
                              lx_match & r = m[i + 1];
                              if (l.qryId == r.qryId && l.subjId == r.subjId && l.subjEnd >= r.subjStart)
                              {
                                  l.subjEnd   = r.subjEnd;
                                  r.subjStart = l.subjStart;
                              }
                          }
                          for (uint64_t i = hi; i-- > lo + 1;)
                          {
                              lx_match & r = m[i];
                              lx_match & l = m[i - 1];
                              if (r.qryId == l.qryId && r.subjId == l.subjId && r.subjStart < l.subjEnd)
                                  l = r;
                          }
                          kept[t] = (uint64_t)(std::unique(m + lo, m + hi, [](lx_match const & a, lx_match const & b) { return tie(a) == tie(b); }) - (m + lo));
                      });
        uint64_t w = 0;
        for (unsigned t = 0; t < nt; ++t) // the pieces close ranks (the first stays where it is)
        {
            if ((m != dst || w != cut[t]) && kept[t])
                std::memmove(static_cast<void *>(dst + w), m + cut[t], kept[t] * sizeof(lx_match));
            w += kept[t];
        }
        if (duplicates)
            *duplicates += before - w;
        return w;
    }
    for (uint64_t i = 0; i < n; ++i)
        widenMatch(m[i], qLens[m[i].qryId], sLens[m[i].subjId]);
    std::sort(m, m + n, less);
    if (n > 1)
    {
        for (uint64_t i = 0; i + 1 < n; ++i)
        {
            lx_match & l = m[i];
            lx_match & r = m[i + 1];
            if (l.qryId == r.qryId && l.subjId == r.subjId && l.subjEnd >= r.subjStart)
            {
                l.subjEnd   = r.subjEnd;
                r.subjStart = l.subjStart;
            }
        }
        for (uint64_t i = n - 1; i >= 1; --i)
        {
            lx_match & r = m[i];
            lx_match & l = m[i - 1];
            if (r.qryId == l.qryId && r.subjId == l.subjId && r.subjStart < l.subjEnd)
                l = r;
        }
        n = std::unique(m, m + n, [](lx_match const & a, lx_match const & b) { return tie(a) == tie(b); }) - m;
    }
    if (duplicates)
        *duplicates += before - n;
    return n;
}

} // namespace lambda_amd

extern "C" {

int lx_karlin_params(int scoring_method, int match, int mismatch, int gap_open_lambda, int gap_extend, lx_karlin * out)
{
    if (!out)
        return LX_EINVAL;
    return lambda_amd::karlinParams(scoring_method, match, mismatch, gap_open_lambda, gap_extend, *out) ? LX_OK : LX_EINVAL;
}

uint64_t lx_length_adjustment(uint64_t db_len, uint64_t q_len, lx_karlin const * ka)
{
    return lambda_amd::lengthAdjustment(db_len, q_len, *ka);
}

double lx_evalue(int32_t score, uint64_t q_len_adj, uint64_t db_len_adj, lx_karlin const * ka)
{
    return lambda_amd::computeEValue(score, q_len_adj, db_len_adj, *ka);
}

double lx_bitscore(int32_t score, lx_karlin const * ka)
{
    return lambda_amd::computeBitScore(score, *ka);
}

// seqan2_to_rank_inner, src/seqan2_to_biocpp.hpp:382-395: aa27 and (bisulfite) dna5 go through a permutation, every
// other alphabet keeps its rank.  The two permutations move X behind Z and N behind T.
int lx_convert_ranks(int kind, uint8_t const * in, uint64_t n, uint8_t * out)
{
    if ((!in || !out) && n != 0)
        return LX_EINVAL;
    if (kind == LX_RANKS_SIMPLE)
    {
        if (in != out)
            std::memmove(out, in, n);
        return LX_OK;
    }
    if (kind != LX_RANKS_AA27 && kind != LX_RANKS_DNA5_BS)
        return LX_EINVAL;
    uint8_t  table[32];
    unsigned size = 0;
    if (kind == LX_RANKS_AA27)
    {
        size = 27;
        for (unsigned r = 0; r < 27; ++r)
            table[r] = (uint8_t)r;
        table[23] = 25; // X
        table[24] = 23; // Y
        table[25] = 24; // Z
    }
    else
    {
        size     = 5;
        table[0] = 0;
        table[1] = 1;
        table[2] = 2;
        table[3] = 4; // N
        table[4] = 3; // T
    }
    int rc = LX_OK;
    for (uint64_t i = 0; i < n; ++i)
    {
        uint8_t const r = in[i];
        if (r >= size)
        {
            rc     = LX_EINVAL;
            out[i] = (uint8_t)(size - 1);
        }
        else
            out[i] = table[r];
    }
    return rc;
}

uint64_t lx_widen_and_preprocess(lx_match * m, uint64_t n, uint64_t const * qlens, uint64_t const * slens)
{
    return lambda_amd::widenAndPreprocessMatches(m, n, qlens, slens, nullptr);
}

} // extern "C"

// iterateMatchesFullSimd, src/search_algo.hpp:1177-1332, for one strand direction (= one scoring slot); appends to *res
static int iterateMatchesFullSimd(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint64_t const * q_seq_off,
                                  uint64_t const * q_seq_len, uint64_t const * q_orig_len, uint8_t const * s_res,
                                  uint64_t s_bytes, uint64_t const * s_seq_off, uint64_t const * s_seq_len,
                                  lx_match * matches, uint64_t n_matches, lx_search_params const * params,
                                  lx_iterate_result * res)
{
    using namespace lambda_amd;
    int const qFrames = std::max(1, params->qry_num_frames);
    res->stats.num_ext_score += n_matches; // lH.stats.numExtScore (:1187)
    // LX_HOST_TIMING=1: where this function's own time goes (the extension prints its breakdown itself)
    static bool const timing = std::getenv("LX_HOST_TIMING") != nullptr;
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

    // pre-sort and filter (:1198)
    uint64_t const n = widenAndPreprocessMatches(matches, n_matches, q_seq_len, s_seq_len, &res->stats.hits_duplicate);
    mark("widen+merge");

    // create blast matches from Lambda matches (:1200-1227); the window is the DP's (query slice, subject slice)
    std::vector<lx_extension> ext(n);
    parallelRanges(n,
                   [&](unsigned, uint64_t lo, uint64_t hi)
                   {
                       for (uint64_t i = lo; i < hi; ++i)
                       {
                           lx_match const & m = matches[i];
                           ext[i].q_off       = q_seq_off[m.qryId] + m.qryStart;
                           ext[i].q_len       = (uint32_t)(m.qryEnd - m.qryStart);
                           ext[i].s_off       = s_seq_off[m.subjId] + m.subjStart;
                           ext[i].s_len       = (uint32_t)(m.subjEnd - m.subjStart);
                       }
                   });
    // The reference sorts the list by lengths to minimise SIMD padding (:1229-1235), runs the extensions WITHOUT
    // alignment (:1246), filters by bit score and e-value (:1251-1283), runs the survivors WITH alignment (:1293-1296)
    // and stably re-sorts by query (:1299).  Here both passes are one call: bit score and e-value are monotone in the
    // raw score for a given query length (src/search_misc.hpp:77-78), so the filter goes to the device as an integer
    // cut-off per extension -- found with the very double formulas below, hence identical decisions -- and the list
    // stays in match order (sorted by query since widen/merge: one LDS profile per run).  The reference's two stable
    // sorts only fix the order of the survivors, which is restored afterwards: (query, lengths, list position).
    CutOffs               cutOffFor(params);
    std::vector<int32_t>  minScore(n);
    std::vector<uint64_t> qLengthOf(n);
    // (the cut-off of every query length that occurs, found once on this thread -- the list is grouped by query: one look per
    // run --, then handed out to the matches by all of them)
    parallelRanges(n,
                   [&](unsigned, uint64_t lo, uint64_t hi)
                   {
                       for (uint64_t i = lo; i < hi; ++i)
                           qLengthOf[i] = q_orig_len ? q_orig_len[matches[i].qryId / qFrames] : q_seq_len[matches[i].qryId];
                   });
    for (uint64_t i = 0; i < n; ++i)
        if (i == 0 || qLengthOf[i] != qLengthOf[i - 1])
            (void)cutOffFor(qLengthOf[i]);
    parallelRanges(n,
                   [&](unsigned, uint64_t lo, uint64_t hi)
                   {
                       for (uint64_t i = lo; i < hi; ++i)
                           minScore[i] = cutOffFor.byLength.find(qLengthOf[i])->second; // (reads only: every length is in the map)
                   });
    mark("slices+cut-offs");
    std::vector<int32_t> scores(n, 0);
    // (the survivors arrive as a list -- the form the filter loop leaves behind, :1251-1283 -- with their ops as run-length
    // codes, the form they cross PCIe in; only the HSPs that pass the identity cut-off are expanded into column bytes.
    // Band mode returns n records and column bytes)
    uint64_t bandNow = 0;
    (void)lx_get_option(h, LX_OPT_BAND, &bandNow);
    auto const window = [&](uint64_t i)
    {
        lx_match const & m = matches[i];
        return WindowView{m.qryId, m.subjId, m.qryStart, m.subjStart, ext[i].q_len, ext[i].s_len, qLengthOf[i]};
    };
    if (bandNow == 0)
    {
        lx_survivor_list list{};
        int              rc = lx_extend_batch_list(h, slot, q_res, q_bytes, s_res, s_bytes, ext.data(), n, minScore.data(), 0, scores.data(), &list);
        if (rc != LX_OK)
            return rc;
        mark("extension");
        rc = finishSurvivors(n, window, scores.data(), [&](uint64_t i) { return minScore[i]; }, list, params, res);
        mark("statistics+records");
        if (timing)
            std::fprintf(stderr, "[lx host ms] iterateMatchesFullSimd (%llu matches):%s\n", (unsigned long long)n_matches, tline.c_str());
        return rc;
    }
    // band mode: n records and column bytes from lx_extend_batch; the survivors become a list over them
    std::vector<lx_hsp>   hspAll(n);
    std::vector<uint64_t> opsOffAll(n);
    uint8_t const *       ops      = nullptr;
    uint64_t              opsBytes = 0;
    int rc = lx_extend_batch(h, slot, q_res, q_bytes, s_res, s_bytes, ext.data(), n, minScore.data(), 0, scores.data(), hspAll.data(), opsOffAll.data(),
                             &ops, &opsBytes);
    if (rc != LX_OK)
        return rc;
    mark("extension");
    std::vector<uint32_t> index;
    std::vector<lx_hsp>   hsps;
    std::vector<uint64_t> codesOff;
    std::vector<uint8_t>  codes;
    for (uint64_t i = 0; i < n; ++i)
        if (scores[i] >= minScore[i] && ext[i].q_len != 0 && ext[i].s_len != 0)
        {
            // (column bytes -> one run-length code per run, the form finishSurvivors expands)
            lx_hsp const &  a     = hspAll[i];
            uint8_t const * first = ops + opsOffAll[i] + a.ops_shift;
            index.push_back((uint32_t)i);
            codesOff.push_back(codes.size());
            for (int32_t c = 0; c < a.n_ops;)
            {
                int32_t e = c + 1;
                while (e < a.n_ops && first[e] == first[c] && e - c < 64)
                    ++e;
                codes.push_back((uint8_t)(((first[c] == 'D' ? 1 : first[c] == 'I' ? 2 : 0) << 6) | (e - c - 1)));
                c = e;
            }
            hsps.push_back(a);
            hsps.back().ops_shift = 0;
        }
    lx_survivor_list list{};
    list.count       = index.size();
    list.index       = index.data();
    list.hsp         = hsps.data();
    list.codes_off   = codesOff.data();
    list.codes       = codes.data();
    list.codes_bytes = codes.size();
    rc               = finishSurvivors(n, window, scores.data(), [&](uint64_t i) { return minScore[i]; }, list, params, res);
    mark("statistics+records");
    if (timing)
        std::fprintf(stderr, "[lx host ms] iterateMatchesFullSimd (%llu matches):%s\n", (unsigned long long)n_matches, tline.c_str());
    return rc;
}

extern "C" {

// the blocks of result memory kept between calls (host/lx_iterate_common.hpp: BlockCache, up to 1 GiB per process) go back to the allocator
uint64_t lx_trim_result_cache(void)
{
    return (uint64_t)lambda_amd::trim_block_cache();
}

// iterateMatches, src/search_algo.hpp:1364-1385
int lx_iterate_matches(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint64_t const * q_seq_off,
                       uint64_t const * q_seq_len, uint64_t n_qseq, uint64_t const * q_orig_len,
                       uint8_t const * s_res, uint64_t s_bytes, uint64_t const * s_seq_off, uint64_t const * s_seq_len,
                       uint64_t n_sseq, lx_match * matches, uint64_t n_matches, lx_search_params const * params,
                       lx_iterate_result ** out)
{
    if (!h || !out || !params || (!matches && n_matches))
        return LX_EINVAL;
    *out = nullptr;
    if (params->flags & ~(int32_t)LX_ITERATE_NO_OPS) // (a caller built against ABI 2 that left the then-reserved word uninitialised)
        return LX_EINVAL;
    lxi::HostPool::Call const in_flight_call;
    {
        std::vector<uint8_t> bad(std::max(1u, lxi::pool_width()), 0);
        lambda_amd::parallelRanges(n_matches,
                                   [&](unsigned t, uint64_t lo, uint64_t hi)
                                   {
                                       for (uint64_t i = lo; i < hi; ++i)
                                           if (matches[i].qryId >= n_qseq || matches[i].subjId >= n_sseq)
                                               bad[t] = 1;
                                   });
        for (uint8_t b : bad)
            if (b)
                return LX_EINVAL;
    }
    auto res = new lx_iterate_result();
    int  rc;
    // a large list: widen, sort, merge and unique on the device, like lx_iterate_matches_dev (the same records either way)
    rc = lxi::iterate_host_list_on_device(h, slot, q_res, q_bytes, q_seq_off, q_seq_len, n_qseq, q_orig_len, s_res, s_bytes, s_seq_off, s_seq_len, n_sseq, matches,
                                          n_matches, params, res);
    if (rc != lxi::kNotTaken)
    {
        if (rc != LX_OK)
        {
            delete res;
            return rc;
        }
        *out = res;
        return LX_OK;
    }
    // band mode for the duration of the call (default centres: the windows are built here, by _widenMatch's rule)
    uint64_t bandBefore = 0;
    (void)lx_get_option(h, LX_OPT_BAND, &bandBefore);
    struct RestoreBand
    {
        lx_handle * h;
        uint64_t    v;
        ~RestoreBand() { (void)lx_set_option(h, LX_OPT_BAND, v); }
    } restoreBand{h, bandBefore};
    if (params->band > 0 && (rc = lx_set_option(h, LX_OPT_BAND, (uint64_t)params->band)) != LX_OK)
    {
        delete res;
        return rc;
    }
    if (params->bisulfite)
    {
        // The bisulfite scheme type selects the computeAlignmentStats overload whose match test is
        // score(c0,c1) == score(c0,c0) (src/evaluate_bisulfite_alignment.hpp:97, called at src/search_algo.hpp:1308):
        // in force for this call whatever the handle's LX_OPT_BS_MATCH_RULE says, restored afterwards.
        uint64_t ruleBefore = 0;
        (void)lx_get_option(h, LX_OPT_BS_MATCH_RULE, &ruleBefore);
        (void)lx_set_option(h, LX_OPT_BS_MATCH_RULE, 1);
        struct Restore
        {
            lx_handle * h;
            uint64_t    v;
            ~Restore() { (void)lx_set_option(h, LX_OPT_BS_MATCH_RULE, v); }
        } restore{h, ruleBefore};
        // sort by (subjId % 2, Match); even subject frames use the forward scheme (slot 0), odd ones the reverse scheme
        // (slot 1); finally the HSPs are stably re-sorted by query (:1367-1379)
        std::sort(matches, matches + n_matches,
                  [](lx_match const & l, lx_match const & r)
                  { return std::make_tuple(l.subjId % 2, lambda_amd::tie(l)) < std::make_tuple(r.subjId % 2, lambda_amd::tie(r)); });
        lx_match * mid = std::find_if(matches, matches + n_matches, [](lx_match const & m) { return m.subjId % 2; });
        rc = iterateMatchesFullSimd(h, 0, q_res, q_bytes, q_seq_off, q_seq_len, q_orig_len, s_res, s_bytes, s_seq_off, s_seq_len,
                                    matches, (uint64_t)(mid - matches), params, res);
        if (rc == LX_OK)
            rc = iterateMatchesFullSimd(h, 1, q_res, q_bytes, q_seq_off, q_seq_len, q_orig_len, s_res, s_bytes, s_seq_off,
                                        s_seq_len, mid, (uint64_t)(matches + n_matches - mid), params, res);
        if (rc == LX_OK)
        {
            // the ops offsets stay valid: only the records move
            std::stable_sort(res->matches.begin(), res->matches.end(),
                             [](lx_blast_match const & a, lx_blast_match const & b) { return a.n_qid < b.n_qid; });
        }
    }
    else
        rc = iterateMatchesFullSimd(h, slot, q_res, q_bytes, q_seq_off, q_seq_len, q_orig_len, s_res, s_bytes, s_seq_off, s_seq_len,
                                    matches, n_matches, params, res);
    if (rc != LX_OK)
    {
        delete res;
        return rc;
    }
    *out = res;
    return LX_OK;
}

uint64_t lx_iterate_result_count(lx_iterate_result const * r)
{
    return r ? r->matches.size() : 0;
}

lx_blast_match const * lx_iterate_result_matches(lx_iterate_result const * r)
{
    return r ? r->matches.data() : nullptr;
}

uint8_t const * lx_iterate_result_ops(lx_iterate_result const * r)
{
    return r ? r->ops.data() : nullptr;
}

lx_iterate_stats lx_iterate_result_stats(lx_iterate_result const * r)
{
    return r ? r->stats : lx_iterate_stats{};
}

void lx_iterate_result_free(lx_iterate_result * r)
{
    delete r;
}

} // extern "C"

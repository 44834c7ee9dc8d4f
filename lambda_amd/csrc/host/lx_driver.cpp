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
#include "scoring_tables.hpp"

// the library's host threads (lx_host.cpp): width of the pool, and f(0) ... f(nthreads - 1) run side by side
namespace lxi
{
unsigned pool_width();
void     pool_run(unsigned nthreads, std::function<void(unsigned)> f);
} // namespace lxi

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

struct lx_iterate_result
{
    std::vector<lx_blast_match> matches;
    std::vector<uint8_t>        ops;
    lx_iterate_stats            stats{};
};

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
    int const qFrames = std::max(1, params->qry_num_frames), sFrames = std::max(1, params->sbj_num_frames);
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
    EValueContext evalue{params->karlin, params->db_total_length, params->query_translated != 0, {}};
    auto passes = [&](int32_t score, uint64_t qLength)
    {
        if (params->min_bitscore >= 0 && computeBitScore(score, params->karlin) < params->min_bitscore)
            return false;
        if (params->max_evalue >= 0 && evalue(score, qLength) > params->max_evalue)
            return false;
        return true;
    };
    std::unordered_map<uint64_t, int32_t> cutOffs; // by query length
    auto cutOffFor = [&](uint64_t qLength)
    {
        auto it = cutOffs.find(qLength);
        if (it != cutOffs.end())
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
        cutOffs.emplace(qLength, cut);
        return cut;
    };
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
                           minScore[i] = cutOffs.find(qLengthOf[i])->second; // (reads only: every length is in the map)
                   });
    mark("slices+cut-offs");
    std::vector<int32_t>  scores(n, 0);
    std::vector<lx_hsp>   hspAll;    // band mode only (n records, column bytes)
    std::vector<uint64_t> opsOffAll;
    uint8_t const *       ops      = nullptr;
    uint64_t              opsBytes = 0;
    // (the survivors arrive as a list -- the form the filter loop leaves behind, :1251-1283 -- with their ops as run-length
    // codes, the form they cross PCIe in; only the HSPs that pass the identity cut-off are expanded into column bytes below.
    // Band mode returns n records and column bytes)
    uint64_t bandNow = 0;
    (void)lx_get_option(h, LX_OPT_BAND, &bandNow);
    bool const       rle = bandNow == 0;
    lx_survivor_list list{};
    int              rc;
    if (rle)
        rc = lx_extend_batch_list(h, slot, q_res, q_bytes, s_res, s_bytes, ext.data(), n, minScore.data(), 0, scores.data(), &list);
    else
    {
        hspAll.resize(n);
        opsOffAll.resize(n);
        rc = lx_extend_batch(h, slot, q_res, q_bytes, s_res, s_bytes, ext.data(), n, minScore.data(), 0, scores.data(), hspAll.data(),
                             opsOffAll.data(), &ops, &opsBytes);
    }
    if (rc != LX_OK)
        return rc;
    mark("extension");

    // the filter's statistics (:1260, :1274) from the scores of pass 1
    std::vector<uint32_t> surv;   // indices into `matches`
    std::vector<uint32_t> listAt; // list mode: where match i stands in the survivor list
    surv.reserve(rle ? list.count : n);
    for (uint64_t i = 0; i < n; ++i)
    {
        if (scores[i] >= minScore[i])
        {
            if (!rle)
                surv.push_back((uint32_t)i);
        }
        else if (params->min_bitscore >= 0 && computeBitScore(scores[i], params->karlin) < params->min_bitscore)
            ++res->stats.failed_bitscore;
        else
            ++res->stats.failed_evalue;
    }
    if (rle)
    {
        listAt.assign(n, 0xffffffffu);
        for (uint64_t k = 0; k < list.count; ++k)
        {
            surv.push_back(list.index[k]);
            listAt[list.index[k]] = (uint32_t)k;
        }
    }
    mark("statistics");
    if (surv.empty())
        return LX_OK;
    res->stats.num_ext_ali += surv.size(); // :1287
    std::sort(surv.begin(), surv.end(),
              [&](uint32_t a, uint32_t b)
              {
                  return std::make_tuple(matches[a].qryId / qFrames, ext[a].q_len, ext[a].s_len, a) <
                         std::make_tuple(matches[b].qryId / qFrames, ext[b].q_len, ext[b].s_len, b);
              });

    // compute the rest of the match properties (:1302-1325).  Two passes over the survivors, each spread over the host threads:
    // the records (and which of them pass the identity cut-off), then -- the offsets known -- their ops.
    uint64_t const              ns = surv.size();
    std::vector<lx_blast_match> recs(ns);
    std::vector<uint8_t>        keep(ns, 0);
    parallelRanges(ns,
                   [&](unsigned, uint64_t lo, uint64_t hi)
                   {
                       EValueContext ev = evalue; // (its cache of length adjustments is not shared)
                       for (uint64_t x = lo; x < hi; ++x)
                       {
                           uint32_t const   k = surv[x];
                           lx_match const & m = matches[k];
                           lx_hsp const &   a = rle ? list.hsp[listAt[k]] : hspAll[k];
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
                           bm.identity = a.n_ops ? (float)(100.0 * static_cast<float>(a.num_matches) / static_cast<float>(a.n_ops)) : 0.0f;
                           if (!(bm.identity < params->id_cutoff)) // :1310-1315
                           {
                               // the reference keeps the values of the filter where it computed them and computes the others now
                               // (:1318-1322): the same formulas on the same score either way
                               bm.bit_score = computeBitScore(a.score, params->karlin);
                               bm.e_value   = ev(a.score, qLengthOf[k]);
                               bm.n_ops     = (uint32_t)a.n_ops;
                               keep[x]      = 1;
                           }
                           recs[x] = bm;
                       }
                   });
    uint64_t const rec0 = res->matches.size(), ops0 = res->ops.size();
    uint64_t       nkeep = 0, nops = 0;
    for (uint64_t x = 0; x < ns; ++x)
    {
        if (!keep[x])
        {
            ++res->stats.failed_identity;
            continue;
        }
        recs[x].ops_off = ops0 + nops; // (ops_off of a dropped record stays unused)
        nops += recs[x].n_ops;
        ++nkeep;
    }
    res->matches.resize(rec0 + nkeep);
    res->ops.resize(ops0 + nops);
    {
        // where the kept records go: their rank among the kept ones (a prefix count per thread share)
        std::vector<uint64_t> rank(ns);
        uint64_t              r = 0;
        for (uint64_t x = 0; x < ns; ++x)
        {
            rank[x] = r;
            r += keep[x];
        }
        parallelRanges(ns,
                       [&](unsigned, uint64_t lo, uint64_t hi)
                       {
                           for (uint64_t x = lo; x < hi; ++x)
                           {
                               if (!keep[x])
                                   continue;
                               uint32_t const         k  = surv[x];
                               lx_blast_match const & bm = recs[x];
                               lx_hsp const &         a  = rle ? list.hsp[listAt[k]] : hspAll[k];
                               uint8_t const * const first = rle ? list.codes + list.codes_off[listAt[k]] : ops + opsOffAll[k] + a.ops_shift;
                               if (rle)
                                   (void)lx_expand_ops(first, a.n_ops, res->ops.data() + bm.ops_off);
                               else
                                   std::memcpy(res->ops.data() + bm.ops_off, first, (size_t)a.n_ops);
                               res->matches[rec0 + rank[x]] = bm;
                           }
                       });
    }
    mark("records");
    if (timing)
        std::fprintf(stderr, "[lx host ms] iterateMatchesFullSimd (%llu matches):%s\n", (unsigned long long)n_matches, tline.c_str());
    return LX_OK;
}

extern "C" {

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
    for (uint64_t i = 0; i < n_matches; ++i)
        if (matches[i].qryId >= n_qseq || matches[i].subjId >= n_sseq)
            return LX_EINVAL;
    auto res = new lx_iterate_result();
    int  rc;
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

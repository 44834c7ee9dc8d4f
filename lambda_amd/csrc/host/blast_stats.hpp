// blast_stats.hpp -- Karlin-Altschul statistics used by the extension filter.
//
// Mirrors the calls the reference makes into seqan/blast/blast_statistics.h (source absent, [UPSTREAM-RECALL]):
//   _lengthAdjustment(dbTotalLength, ql, scheme)            /root/reference/src/search_misc.hpp:73
//   _computeEValue(score, ql - adj, dbTotalLength - adj, .) src/search_misc.hpp:77-78
//   computeBitScore(bm, context)                            src/search_algo.hpp:1258, :1319
// and the per-thread length-adjustment cache of computeEValueThreadSafe (src/search_misc.hpp:56-80).
// Parameter tables are NCBI's published values (blast_stat.c); the length adjustment is NCBI's
// BLAST_ComputeLengthAdjustment with the number of database sequences fixed to 1.
#pragma once
#include <cmath>
#include <cstdint>
#include <unordered_map>

#include "../../../include/lambda_ext.h"

namespace lambda_amd
{

struct KarlinRow
{
    int    gapOpen, gapExtend; // BLAST convention, positive
    double lambda, K, H, alpha, beta;
};

// clang-format off
inline constexpr KarlinRow kBlosum62Rows[] = {
    {11, 2, 0.297, 0.082, 0.27, 1.1, -10}, {10, 2, 0.291, 0.075, 0.23, 1.3, -15}, { 9, 2, 0.279, 0.058, 0.19, 1.5, -19},
    { 8, 2, 0.264, 0.045, 0.15, 1.8, -26}, { 7, 2, 0.239, 0.027, 0.10, 2.5, -46}, { 6, 2, 0.201, 0.012, 0.061, 3.3, -58},
    {13, 1, 0.292, 0.071, 0.23, 1.2, -11}, {12, 1, 0.283, 0.059, 0.19, 1.5, -19}, {11, 1, 0.267, 0.041, 0.14, 1.9, -30},
    {10, 1, 0.243, 0.024, 0.10, 2.5, -44}, { 9, 1, 0.206, 0.010, 0.052, 4.0, -87},
};
inline constexpr KarlinRow kBlosum45Rows[] = { // PROVISIONAL (from memory of blast_stat.c)
    {13, 3, 0.207, 0.049, 0.14, 1.5, -22}, {12, 3, 0.199, 0.039, 0.11, 1.8, -34}, {11, 3, 0.190, 0.031, 0.095, 2.0, -38},
    {10, 3, 0.179, 0.023, 0.075, 2.4, -51}, {16, 2, 0.210, 0.051, 0.14, 1.5, -24}, {15, 2, 0.203, 0.041, 0.12, 1.7, -31},
    {14, 2, 0.195, 0.032, 0.10, 1.9, -36}, {13, 2, 0.185, 0.024, 0.084, 2.2, -45}, {12, 2, 0.171, 0.016, 0.061, 2.8, -65},
    {19, 1, 0.205, 0.040, 0.11, 1.9, -43}, {18, 1, 0.198, 0.032, 0.10, 2.0, -43}, {17, 1, 0.189, 0.024, 0.079, 2.4, -57},
    {16, 1, 0.176, 0.016, 0.063, 2.8, -67},
};
inline constexpr KarlinRow kBlosum80Rows[] = { // PROVISIONAL (from memory of blast_stat.c)
    {25, 2, 0.342, 0.17, 0.66, 0.52, -1.6}, {13, 2, 0.336, 0.15, 0.57, 0.59, -3}, { 9, 2, 0.319, 0.11, 0.42, 0.76, -6},
    { 8, 2, 0.308, 0.090, 0.35, 0.89, -9}, { 7, 2, 0.293, 0.070, 0.27, 1.1, -14}, { 6, 2, 0.268, 0.045, 0.19, 1.4, -19},
    {11, 1, 0.314, 0.095, 0.35, 0.90, -9}, {10, 1, 0.299, 0.071, 0.27, 1.1, -14}, { 9, 1, 0.279, 0.048, 0.20, 1.4, -19},
};
// blastn, reward 2 / penalty -3
inline constexpr KarlinRow kNucl2_3Rows[] = {
    {4, 4, 0.63, 0.42, 0.84, 0.75, -2}, {2, 4, 0.615, 0.37, 0.72, 0.85, -3}, {0, 4, 0.55, 0.21, 0.46, 1.2, -5},
    {3, 3, 0.615, 0.37, 0.68, 0.9, -3}, {6, 2, 0.63, 0.42, 0.84, 0.75, -2}, {5, 2, 0.625, 0.41, 0.78, 0.8, -2},
    {4, 2, 0.61, 0.35, 0.68, 0.9, -3}, {2, 2, 0.515, 0.14, 0.33, 1.55, -9},
};
// clang-format on

// scoringMethod as in LambdaOptions (62/45/80; 0 = manual match/mismatch); gapOpen/gapExtend are lambda's
// (negative) option values, e.g. -11/-1.
inline bool karlinParams(int scoringMethod, int match, int misMatch, int gapOpen, int gapExtend, lx_karlin & out)
{
    KarlinRow const * rows = nullptr;
    size_t            n    = 0;
    switch (scoringMethod)
    {
        case 62: rows = kBlosum62Rows; n = sizeof(kBlosum62Rows) / sizeof(KarlinRow); break;
        case 45: rows = kBlosum45Rows; n = sizeof(kBlosum45Rows) / sizeof(KarlinRow); break;
        case 80: rows = kBlosum80Rows; n = sizeof(kBlosum80Rows) / sizeof(KarlinRow); break;
        case 0:
        case -1:
        case -2:
            if (match == 2 && misMatch == -3)
            {
                rows = kNucl2_3Rows;
                n    = sizeof(kNucl2_3Rows) / sizeof(KarlinRow);
            }
            break;
        default: break;
    }
    for (size_t i = 0; i < n; ++i)
        if (rows[i].gapOpen == -gapOpen && rows[i].gapExtend == -gapExtend)
        {
            out = lx_karlin{rows[i].lambda, rows[i].K, rows[i].H, rows[i].alpha, rows[i].beta};
            return true;
        }
    return false;
}

inline uint64_t lengthAdjustment(uint64_t dbLength, uint64_t queryLength, lx_karlin const & ka)
{
    double const K = ka.K, logK = std::log(K), alphaByLambda = ka.alpha / ka.lambda, beta = ka.beta;
    double const n = (double)dbLength, m = (double)queryLength;
    double       val = 0, val_min = 0, val_max, totalLen;
    bool         converged = false;
    {
        double const mb = m + n;
        double const c  = n * m - (m > n ? m : n) / K;
        if (c < 0)
            return 0;
        val_max = 2 * c / (mb + std::sqrt(mb * mb - 4 * c));
    }
    for (int i = 1; i <= 20; ++i)
    {
        totalLen             = (m - val) * (n - val);
        double const val_new = alphaByLambda * (logK + std::log(totalLen)) + beta;
        if (val_new >= val)
        {
            val_min = val;
            if (val_new - val_min <= 1.0)
            {
                converged = true;
                break;
            }
            if (val_min == val_max)
                break;
        }
        else
            val_max = val;
        if (val_min <= val_new && val_new <= val_max)
            val = val_new;
        else
            val = (i == 1) ? val_max : (val_min + val_max) / 2;
    }
    if (converged)
    {
        val = std::ceil(val_min);
        if (val <= val_max)
        {
            totalLen = (m - val) * (n - val);
            if (alphaByLambda * (logK + std::log(totalLen)) + beta >= val)
                return (uint64_t)val;
        }
    }
    return (uint64_t)val_min;
}

inline double computeEValue(int32_t score, uint64_t ql, uint64_t dl, lx_karlin const & ka)
{
    return ka.K * (double)ql * (double)dl * std::exp(-ka.lambda * (double)score);
}

inline double computeBitScore(int32_t score, lx_karlin const & ka)
{
    return (ka.lambda * (double)score - std::log(ka.K)) / std::log(2.0);
}

// computeEValueThreadSafe, src/search_misc.hpp:56-80 (the cache lives in the caller: one per driver call)
struct EValueContext
{
    lx_karlin                              ka;
    uint64_t                               dbTotalLength;
    bool                                   queryTranslated;
    std::unordered_map<uint64_t, uint64_t> cachedLengthAdjustments;
    uint64_t                               lastLength = ~0ull, lastAdjustment = 0; // (lists come grouped by query: the previous answer is usually the next)

    double operator()(int32_t score, uint64_t ql)
    {
        ql = ql / (queryTranslated ? 3 : 1);
        if (ql != lastLength)
        {
            auto it = cachedLengthAdjustments.find(ql);
            if (it == cachedLengthAdjustments.end())
                it = cachedLengthAdjustments.emplace(ql, lengthAdjustment(dbTotalLength, ql, ka)).first;
            lastLength     = ql;
            lastAdjustment = it->second;
        }
        uint64_t const adj = lastAdjustment;
        return computeEValue(score, ql - adj, dbTotalLength - adj, ka);
    }
};

} // namespace lambda_amd

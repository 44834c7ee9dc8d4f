// lx_seeding_gpu.hpp -- the seeding stage of the lambda3 front end on the GPU (gfx950 only).
//
// Same semantics as host/lx_seeding.hpp's seedQueries -- search() of the reference, /root/reference/src/search_algo.hpp:611-762:
// seeds every seedOffset letters of the reduced query, exact or half-exact search (:505-604), adaptive elongation (:679-727), the
// over-abundance cut (:729), seedLooksPromising per located hit (:426-481) -- over the same sorted word table.  What is serial
// there is serial here: the hits a read has collected so far steer the elongation of its next seeds, so ONE LANE owns a read and
// walks its frames, seeds, cursors and hits in the host's order; reads are independent, so a launch is one lane per read.  A
// lane's time is a chain of dependent table probes (the prefix table for a word's first letters, binary searches inside the
// range after that); what hides their latency is the other reads -- 100 000 reads are 1 563 wavefronts.
//   * The half-exact search is the host's level-by-level expansion walked depth first: the host's list of cursors after the
//     last level is in lexicographic order of the words (children are appended in letter order), which is the order a
//     depth-first walk with ascending letters reaches them -- same cursors, same order, so the same hitsThisSeq at every step.
//   * Beyond the table's key length a cursor keeps the entries of its range that still match as a bit mask (the host keeps a
//     list); a range of more than 32 entries at that point, or a seed longer than kMaxSecond letters behind its exact part,
//     sends the READ to the host (flag per read; the front end seeds those reads with seedQueries) -- results are identical by
//     construction, the device only declines.
//   * Matches leave through one atomic counter (a full buffer is reported; the reads of that launch then go to the host).
//     Their order is the lanes', not the host's: iterateMatches sorts its span first (src/search_algo.hpp:1141), so the order of
//     the list carries no meaning.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "lx_seeding.hpp"

namespace lambda_amd
{

struct SeedDev
{
    ReducedIndex::Entry const * entries;
    uint64_t const *            pre;
    uint64_t                    pow[64];
    uint64_t                    base;
    int                         preLen, keyLen, alph;
    uint8_t const *             sRes; // subjects: alignment ranks, reduced letters
    uint8_t const *             sRed;
    uint64_t const *            sOff;
    uint64_t const *            sLen;
    uint8_t const *             qRes; // queries (frame-expanded)
    uint8_t const *             qRed;
    uint64_t const *            qOff;
    uint64_t const *            qLen;
    uint64_t                    nQSeq;
    int                         qNumFrames;
    int                         unknownRank;
    int8_t const *              matrix;
    int8_t const *              matrixRev; // bisulfite: hits on odd subject frames (NULL otherwise)
    uint64_t                    maxMatches;
    int                         halfExact, adaptive, preScoring;
    double                      preScoringThresh;
    int                         seedLength, seedOffset, maxSeedDist;
    uint64_t const *            reads; // first frame sequence of every read of this launch
    uint64_t                    nReads;
    lx_match *                  out;
    unsigned long long *        counters; // [0] matches written, [1] hitsAfterSeeding, [2] hitsFailedPreExtendTest, [3] buffer full
    uint64_t                    outCap;
    uint8_t *                   declined; // per read of this launch: 1 = seed it on the host
};

constexpr int kMaxSecond = 24; // letters behind the exact part of a seed the depth-first walk holds

struct DevCursor
{
    uint32_t lo, hi;
    uint32_t mask; // beyond the key length: which entries of [lo, hi) still match (bit e = entry lo + e)
    int      len;
    uint64_t prefix;
};

__device__ __forceinline__ uint32_t dev_count(DevCursor const & c, int keyLen)
{
    return c.len > keyLen ? (uint32_t)__popc(c.mask) : c.hi - c.lo;
}

// the cursor of word + c (ReducedIndex::extendRight); ok = false: the device declines (too many entries beyond the keys)
__device__ __forceinline__ DevCursor dev_extend(SeedDev const & p, DevCursor const & cu, uint32_t c, bool & ok)
{
    DevCursor n = cu;
    n.len       = cu.len + 1;
    if (cu.len >= p.keyLen)
    {
        uint32_t m = cu.len == p.keyLen ? (cu.hi - cu.lo >= 32 ? 0xffffffffu : ((1u << (cu.hi - cu.lo)) - 1u)) : cu.mask;
        if (cu.len == p.keyLen && cu.hi - cu.lo > 32)
        {
            ok = false;
            return n;
        }
        uint32_t keep = 0;
        for (uint32_t rest = m; rest != 0; rest &= rest - 1)
        {
            int const                   e = __ffs((int)rest) - 1;
            ReducedIndex::Entry const & x = p.entries[cu.lo + (uint32_t)e];
            if ((uint64_t)x.pos + (uint64_t)cu.len < p.sLen[x.seq] && p.sRed[p.sOff[x.seq] + x.pos + (uint64_t)cu.len] == c)
                keep |= 1u << e;
        }
        n.mask = keep;
        return n;
    }
    n.prefix = cu.prefix * p.base + c;
    if (n.len <= p.preLen)
    {
        uint64_t const span = p.pow[p.preLen - n.len];
        n.lo                = (uint32_t)p.pre[n.prefix * span];
        n.hi                = (uint32_t)p.pre[(n.prefix + 1) * span];
        return n;
    }
    uint64_t const scale = p.pow[p.keyLen - n.len], first = n.prefix * scale, last = first + (scale - 1);
    uint32_t       a = cu.lo, b = cu.hi;
    while (a < b) // first entry with key >= first
    {
        uint32_t const mid = a + (b - a) / 2;
        if (p.entries[mid].key < first)
            a = mid + 1;
        else
            b = mid;
    }
    n.lo = a;
    b    = cu.hi;
    while (a < b) // first entry with key > last
    {
        uint32_t const mid = a + (b - a) / 2;
        if (p.entries[mid].key <= last)
            a = mid + 1;
        else
            b = mid;
    }
    n.hi = a;
    return n;
}

// seedLooksPromising (host/lx_seeding.hpp, :426-481), the same integer arithmetic
__device__ __forceinline__ bool dev_promising(uint8_t const * q, uint64_t qLen, uint8_t const * s, uint64_t sLen, uint64_t qryStart, uint64_t qryEnd,
                                              uint64_t subjStart, int seedLength, int preScoring, double preScoringThresh, int8_t const * matrix)
{
    int64_t  qFrom = (int64_t)qryStart, sFrom = (int64_t)subjStart;
    uint64_t seedSpan = qryEnd - qryStart;
    uint64_t span     = (uint64_t)(seedLength * preScoring) > seedSpan ? (uint64_t)(seedLength * preScoring) : seedSpan;
    if (span > seedSpan)
    {
        qFrom -= (int64_t)((span - seedSpan) / 2);
        sFrom -= (int64_t)((span - seedSpan) / 2);
        int64_t const mn = qFrom < sFrom ? qFrom : sFrom;
        if (mn < 0)
        {
            qFrom -= mn;
            sFrom -= mn;
            span += (uint64_t)mn;
        }
        uint64_t const a = qLen - (uint64_t)qFrom, b = sLen - (uint64_t)sFrom;
        span             = a < span ? a : span;
        span             = b < span ? b : span;
    }
    int       sc = 0, maxScore = 0;
    int const thresh = (int)(preScoringThresh * (double)span);
    for (uint64_t i = 0; i < span; ++i)
    {
        sc += matrix[(q[(uint64_t)qFrom + i] & 31) * LX_ALPH + (s[(uint64_t)sFrom + i] & 31)];
        if (sc < 0)
            sc = 0;
        else if (sc > maxScore)
            maxScore = sc;
        if (maxScore >= thresh)
            return true;
    }
    return false;
}

__global__ __launch_bounds__(64) void seed_reads_kernel(SeedDev p)
{
    uint64_t const r = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= p.nReads)
        return;
    uint64_t const read0 = p.reads[r];
    size_t         foundForRead = 0, framesTotal = 0, framesDone = 0;
    size_t const   kOccFactor  = 10;
    unsigned long long nHits = 0, nFailed = 0;
    bool               ok    = true;

    // one final cursor of a seed: adaptive elongation, the over-abundance cut, every located hit through seedLooksPromising
    auto finish_cursor = [&](DevCursor cursor, uint64_t i, uint64_t L, uint8_t const * red, uint8_t const * res, uint64_t seedBegin)
    {
        uint64_t seedLength = (uint64_t)p.seedLength;
        if (p.adaptive)
        {
            size_t const left        = (framesTotal - framesDone - seedBegin) / (size_t)p.seedOffset;
            size_t       wanted = foundForRead >= p.maxMatches ? 1 : (p.maxMatches - foundForRead) * kOccFactor / (left > 1 ? left : 1);
            if (wanted == 0)
                wanted = 1;
            DevCursor kept = cursor;
            size_t    keptCount  = dev_count(cursor, p.keyLen);
            while (seedBegin + seedLength < L)
            {
                cursor = dev_extend(p, cursor, red[seedBegin + seedLength], ok);
                if (!ok)
                    return;
                size_t const count = dev_count(cursor, p.keyLen);
                if (count < wanted && count < keptCount)
                {
                    cursor = kept;
                    break;
                }
                ++seedLength;
                keptCount  = count;
                kept = cursor;
            }
        }
        uint32_t const cnt = dev_count(cursor, p.keyLen);
        if (cnt > kOccFactor * p.maxMatches)
            return;
        bool const listed = cursor.len > p.keyLen;
        for (uint32_t e = cursor.lo; e < cursor.hi; ++e)
        {
            if (listed && !((cursor.mask >> (e - cursor.lo)) & 1u))
                continue;
            ReducedIndex::Entry const x = p.entries[e];
            ++nHits;
            int8_t const * mat = (p.matrixRev && (x.seq & 1u)) ? p.matrixRev : p.matrix;
            if (!dev_promising(res, L, p.sRes + p.sOff[x.seq], p.sLen[x.seq], seedBegin, seedBegin + seedLength, x.pos, p.seedLength, p.preScoring,
                               p.preScoringThresh, mat))
                ++nFailed;
            else
            {
                unsigned long long const at = atomicAdd(p.counters, 1ull);
                if (at < p.outCap)
                    p.out[at] = lx_match{i, x.seq, seedBegin, seedBegin + seedLength, x.pos, x.pos + seedLength};
                else
                    atomicExch(p.counters + 3, 1ull);
                ++foundForRead;
            }
        }
    };

    for (int f = 0; f < p.qNumFrames && ok; ++f)
    {
        uint64_t const i = read0 + (uint64_t)f;
        if (i >= p.nQSeq)
            break;
        if (f == 0) // (the per-read reset of seedQueries, before the length test like there)
        {
            foundForRead = framesTotal = framesDone = 0;
            for (int j = 0; j < p.qNumFrames && i + (uint64_t)j < p.nQSeq; ++j)
                framesTotal += p.qLen[i + (uint64_t)j];
        }
        if (p.qLen[i] < (uint64_t)p.seedLength)
            continue;
        uint64_t const        L   = p.qLen[i];
        uint8_t const * const red = p.qRed + p.qOff[i];
        uint8_t const * const res = p.qRes + p.qOff[i];
        for (uint64_t seedBegin = 0; ok; seedBegin += (uint64_t)p.seedOffset)
        {
            while (seedBegin < L - (uint64_t)p.seedLength && (res[seedBegin] == (uint8_t)p.unknownRank || res[seedBegin] == res[seedBegin + 1]))
                ++seedBegin;
            if (seedBegin > L - (uint64_t)p.seedLength)
                break;
            uint8_t const * const seed = red + seedBegin;
            int const firstHalf  = p.maxSeedDist == 0 ? p.seedLength : p.halfExact ? p.seedLength / 2 : 0;
            int const secondHalf = p.seedLength - firstHalf;
            if (secondHalf > kMaxSecond)
            {
                ok = false;
                break;
            }
            DevCursor c{0u, (uint32_t)p.pre[p.pow[p.preLen]], 0u, 0, 0ull};
            bool      alive = true;
            for (int k = 0; k < firstHalf && alive && ok; ++k)
            {
                c     = dev_extend(p, c, seed[k], ok);
                alive = dev_count(c, p.keyLen) != 0;
            }
            if (!ok)
                break;
            if (!alive)
                continue;
            if (secondHalf == 0)
            {
                finish_cursor(c, i, L, red, res, seedBegin);
                continue;
            }
            // depth-first over the second half: a cursor below the error budget branches into every letter (another letter costs
            // one error), the others go on with the seed's letter
            DevCursor stack[kMaxSecond + 1];
            uint8_t   errs[kMaxSecond + 1], next[kMaxSecond + 1];
            int       level = 0;
            stack[0] = c, errs[0] = 0, next[0] = 0;
            while (level >= 0 && ok)
            {
                if (level == secondHalf)
                {
                    finish_cursor(stack[level], i, L, red, res, seedBegin);
                    --level;
                    continue;
                }
                uint8_t const want = seed[firstHalf + level];
                uint32_t      letter;
                if ((int)errs[level] < p.maxSeedDist)
                {
                    if ((int)next[level] >= p.alph)
                    {
                        --level;
                        continue;
                    }
                    letter = next[level]++;
                }
                else
                {
                    if (next[level] != 0)
                    {
                        --level;
                        continue;
                    }
                    next[level] = 1;
                    letter      = want;
                }
                DevCursor const child = dev_extend(p, stack[level], letter, ok);
                if (!ok || dev_count(child, p.keyLen) == 0)
                    continue;
                stack[level + 1] = child;
                errs[level + 1]  = (uint8_t)(errs[level] + (letter != want ? 1 : 0));
                next[level + 1]  = 0;
                ++level;
            }
        }
        framesDone += L;
    }
    if (!ok)
    {
        p.declined[r] = 1; // (what the lane wrote so far is dropped by the host, which seeds the read again from its first frame)
        return;
    }
    if (nHits)
        atomicAdd(p.counters + 1, nHits);
    if (nFailed)
        atomicAdd(p.counters + 2, nFailed);
}

// ---- the word table on the GPU: the keys of all positions (one thread each), one radix sort of (key, sequence << 32 | position)
// pairs -- rocPRIM's device-wide sort through hipCUB: a library primitive, like a GEMM would be; stable, and the pairs start in
// (sequence, position) order, so equal words end up ordered by sequence and position as in the host's table --, the entries
// interleaved, the prefix table by one binary search per prefix.  The same table bit for bit (`lambda3 mkindex* --table gpu|host`
// write identical files).
__global__ void table_keys_kernel(uint8_t const * red, uint64_t const * off, uint64_t const * len, uint64_t const * first, uint64_t nSeq, uint64_t total,
                                  int keyLen, uint64_t base, int alph, uint64_t * keys, uint64_t * vals)
{
    uint64_t const e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total)
        return;
    uint64_t a = 0, b = nSeq; // the sequence that holds entry e: last s with first[s] <= e (empty sequences share a start)
    while (b - a > 1)
    {
        uint64_t const mid = a + (b - a) / 2;
        if (first[mid] <= e)
            a = mid;
        else
            b = mid;
    }
    while (a + 1 < nSeq && first[a + 1] <= e) // (skip empty sequences that start where the next one does)
        ++a;
    uint64_t const pos = e - first[a], L = len[a];
    uint8_t const * r  = red + off[a] + pos;
    uint64_t        key = 0;
    for (int i = 0; i < keyLen; ++i)
        key = key * base + (pos + (uint64_t)i < L ? (uint64_t)r[i] : (uint64_t)alph);
    keys[e] = key;
    vals[e] = (a << 32) | pos;
}

__global__ void table_entries_kernel(uint64_t const * keys, uint64_t const * vals, uint64_t total, ReducedIndex::Entry * out)
{
    uint64_t const e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total)
        out[e] = ReducedIndex::Entry{keys[e], (uint32_t)(vals[e] >> 32), (uint32_t)vals[e]};
}

__global__ void table_prefix_kernel(uint64_t const * keys, uint64_t total, uint64_t preDiv, uint64_t nPre, uint64_t * pre)
{
    uint64_t const w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nPre)
        return;
    uint64_t a = 0, b = total; // first entry whose first preLen letters are >= the word w
    while (a < b)
    {
        uint64_t const mid = a + (b - a) / 2;
        if (keys[mid] / preDiv < w)
            a = mid + 1;
        else
            b = mid;
    }
    pre[w] = a;
}

#define LXS_HIP(call)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t const e_ = (call);                                                                                  \
        if (e_ != hipSuccess)                                                                                          \
            throw std::runtime_error(std::string("GPU seeding: ") + #call + ": " + hipGetErrorString(e_));             \
    } while (0)

// the table as the GPU builder left it on its device: the seeding stage of a worker on that device reads it where it is
struct DeviceTable
{
    int                   device  = -1;
    ReducedIndex::Entry * entries = nullptr;
    uint64_t *            pre     = nullptr;
    DeviceTable() = default;
    DeviceTable(DeviceTable const &) = delete;
    DeviceTable & operator=(DeviceTable const &) = delete;
    ~DeviceTable()
    {
        if (device >= 0 && hipSetDevice(device) == hipSuccess)
        {
            if (entries)
                (void)hipFree(entries);
            if (pre)
                (void)hipFree(pre);
        }
    }
};

// The table, the subjects and the queries of one worker on its device; seed() runs one pass of the batch loop.
class GpuSeeder
{
    template <typename T>
    struct Buf
    {
        T *    p = nullptr;
        size_t n = 0;
        void   upload(T const * src, size_t count)
        {
            reserve(count);
            if (count)
                LXS_HIP(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
        }
        void reserve(size_t count)
        {
            if (count > n)
            {
                if (p)
                    (void)hipFree(p);
                p = nullptr;
                n = 0;
                LXS_HIP(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T)));
                n = count;
            }
        }
        ~Buf()
        {
            if (p)
                (void)hipFree(p);
        }
    };
    int                           device_;
    SeedDev                       d_{};
    Buf<ReducedIndex::Entry>      entries_;
    Buf<uint64_t>                 pre_, sOff_, sLen_, qOff_, qLen_, reads_;
    Buf<uint8_t>                  sRes_, sRed_, qRes_, qRed_, declined_;
    Buf<int8_t>                   matrix_, matrixRev_;
    Buf<lx_match>                 out_;
    Buf<unsigned long long>       counters_;

public:
    // (cursors are 32-bit ranges of the table; a seed's second part is held on a stack of kMaxSecond levels)
    static bool canTake(ReducedIndex const & ix) { return ix.entriesCount() < 0xffffffffull && ix.keyLen() < 64; }
    GpuSeeder(int device, ReducedIndex const & ix, SeedingInput const & in, std::vector<uint8_t> const & sRed, uint64_t nSSeq, uint64_t sBytes, uint64_t qBytes,
              DeviceTable const * resident = nullptr)
        : device_(device)
    {
        LXS_HIP(hipSetDevice(device_));
        bool const adopt = resident && resident->device == device_ && resident->entries && resident->pre;
        if (!adopt)
        {
            entries_.upload(ix.entriesData(), ix.entriesCount());
            pre_.upload(ix.prefixData(), ix.prefixCount());
        }
        sRes_.upload(in.sRes, sBytes);
        sRed_.upload(sRed.data(), sRed.size());
        sOff_.upload(in.sOff, nSSeq);
        sLen_.upload(in.sLen, nSSeq);
        qRes_.upload(in.qRes, qBytes);
        qRed_.upload(in.qRed, qBytes);
        qOff_.upload(in.qOff, in.nQSeq);
        qLen_.upload(in.qLen, in.nQSeq);
        matrix_.upload(in.matrix, LX_ALPH * LX_ALPH);
        if (in.matrixRev)
            matrixRev_.upload(in.matrixRev, LX_ALPH * LX_ALPH);
        counters_.reserve(4);
        d_.entries = adopt ? resident->entries : entries_.p, d_.pre = adopt ? resident->pre : pre_.p, d_.base = (uint64_t)ix.alphabet() + 1, d_.preLen = ix.prefixLen(), d_.keyLen = ix.keyLen(), d_.alph = ix.alphabet();
        for (int k = 0; k <= ix.keyLen() && k < 64; ++k)
            d_.pow[k] = ix.power(k);
        d_.sRes = sRes_.p, d_.sRed = sRed_.p, d_.sOff = sOff_.p, d_.sLen = sLen_.p;
        d_.qRes = qRes_.p, d_.qRed = qRed_.p, d_.qOff = qOff_.p, d_.qLen = qLen_.p, d_.nQSeq = in.nQSeq, d_.qNumFrames = in.qNumFrames;
        d_.unknownRank = in.unknownRank, d_.matrix = matrix_.p, d_.matrixRev = in.matrixRev ? matrixRev_.p : nullptr;
        d_.maxMatches = in.maxMatches, d_.halfExact = in.halfExact ? 1 : 0, d_.adaptive = in.adaptive ? 1 : 0, d_.preScoring = in.preScoring;
        d_.preScoringThresh = in.preScoringThresh;
    }

    // `which`: frame sequences, all frames of a read adjacent and in order (as seedQueries takes them).  Appends the matches of
    // the reads the device took and lists in `declinedReads` the first frame sequence of every read it left to the host: reads a
    // lane declined, and all reads of a launch whose match buffer filled up (returned: the number of such launches).  The reads go
    // to the device in launches of at most kLaunchReads (the match buffer holds 64 matches per read of a launch, as far as the
    // device's free memory allows).
    // `onDevice` (optional): a pass that is ONE launch which nothing declined leaves its matches where the kernel wrote them --
    // devMatches() / *onDevice of them, for lx_iterate_matches_dev -- and appends nothing to `matches`.
    static constexpr uint64_t kLaunchReads = 4u << 20;
    size_t seed(SeedParams const & so, std::vector<uint64_t> const & which, std::vector<lx_match> & matches, SeedingStats & stats,
                std::vector<uint64_t> & declinedReads, uint64_t * onDevice = nullptr)
    {
        LXS_HIP(hipSetDevice(device_));
        if (onDevice)
            *onDevice = 0;
        std::vector<uint64_t> reads;
        for (uint64_t i : which)
            if (i % (uint64_t)d_.qNumFrames == 0)
                reads.push_back(i);
        uint64_t launchReads = kLaunchReads;
        if (char const * forced = std::getenv("LAMBDA3_SEED_LAUNCH")) // (development aid: several launches on a small input)
            launchReads = std::max<uint64_t>(1, std::strtoull(forced, nullptr, 10));
        size_t full = 0;
        bool const oneLaunch = reads.size() <= launchReads;
        for (uint64_t a = 0; a < reads.size(); a += launchReads)
            full += seedLaunch(so, reads.data() + a, std::min<uint64_t>(launchReads, reads.size() - a), matches, stats, declinedReads,
                               oneLaunch ? onDevice : nullptr) ? 0 : 1;
        return full;
    }
    lx_match const * devMatches() const { return out_.p; }

private:
    bool seedLaunch(SeedParams const & so, uint64_t const * reads, uint64_t nReads, std::vector<lx_match> & matches, SeedingStats & stats,
                    std::vector<uint64_t> & declinedReads, uint64_t * onDevice)
    {
        reads_.upload(reads, nReads);
        declined_.reserve(nReads);
        LXS_HIP(hipMemset(declined_.p, 0, nReads));
        uint64_t cap = std::max<uint64_t>(1u << 20, 64ull * nReads);
        if (cap > out_.n)
        {
            // (no more than half of what the device has left beside the extension's buffers: a launch whose buffer fills up is the
            // host's, which is slow but right; a failed allocation would end the search)
            size_t freeB = 0, totalB = 0;
            LXS_HIP(hipMemGetInfo(&freeB, &totalB));
            cap = std::max<uint64_t>(out_.n, std::min<uint64_t>(cap, (freeB + out_.n * sizeof(lx_match)) / 2 / sizeof(lx_match)));
        }
        if (char const * forced = std::getenv("LAMBDA3_SEED_CAP")) // (development aid: a small buffer exercises the "buffer full" path)
            cap = std::max<uint64_t>(1, std::strtoull(forced, nullptr, 10));
        out_.reserve(cap);
        LXS_HIP(hipMemset(counters_.p, 0, 4 * sizeof(unsigned long long)));
        SeedDev p     = d_;
        p.seedLength  = so.seedLength, p.seedOffset = so.seedOffset, p.maxSeedDist = so.maxSeedDist;
        p.reads       = reads_.p, p.nReads = nReads;
        p.out         = out_.p, p.counters = counters_.p, p.outCap = cap, p.declined = declined_.p;
        hipLaunchKernelGGL(seed_reads_kernel, dim3((unsigned)((nReads + 63) / 64)), dim3(64), 0, 0, p);
        LXS_HIP(hipGetLastError());
        unsigned long long cnt[4];
        LXS_HIP(hipMemcpy(cnt, counters_.p, sizeof(cnt), hipMemcpyDeviceToHost)); // (synchronises)
        if (cnt[3] != 0)
        {
            // the buffer filled up: nothing of this launch is kept, its reads are the host's
            declinedReads.insert(declinedReads.end(), reads, reads + nReads);
            return false;
        }
        std::vector<uint8_t> decl(nReads);
        LXS_HIP(hipMemcpy(decl.data(), declined_.p, nReads, hipMemcpyDeviceToHost));
        if (onDevice && matches.empty() && std::all_of(decl.begin(), decl.end(), [](uint8_t d) { return d == 0; }))
        {
            // nothing declined: the matches stay in out_ for the Level-2 kernels
            *onDevice = cnt[0];
            stats.hitsAfterSeeding += cnt[1];
            stats.hitsFailedPreExtendTest += cnt[2];
            return true;
        }
        size_t const at = matches.size();
        matches.resize(at + cnt[0]);
        if (cnt[0])
            LXS_HIP(hipMemcpy(matches.data() + at, out_.p, cnt[0] * sizeof(lx_match), hipMemcpyDeviceToHost));
        std::vector<uint64_t> mine;
        for (size_t k = 0; k < nReads; ++k)
            if (decl[k])
                mine.push_back(reads[k]);
        if (!mine.empty())
        {
            // a declined read's matches are the host's to make: drop what the device wrote for it (it counted nothing)
            std::sort(mine.begin(), mine.end());
            size_t o = at;
            for (size_t k = at; k < matches.size(); ++k)
            {
                uint64_t const rd = matches[k].qryId - matches[k].qryId % (uint64_t)d_.qNumFrames;
                if (!std::binary_search(mine.begin(), mine.end(), rd))
                    matches[o++] = matches[k];
            }
            matches.resize(o);
            declinedReads.insert(declinedReads.end(), mine.begin(), mine.end());
        }
        stats.hitsAfterSeeding += cnt[1];
        stats.hitsFailedPreExtendTest += cnt[2];
        return true;
    }
};

// Fills `ix` with the table made on `device`.  false: not attempted (a table of 2^31 entries or more, or one that does not fit the
// device's free memory) -- the caller builds on the host.
inline bool buildTableOnGpu(int device, ReducedIndex & ix, std::vector<uint8_t> const & red, std::vector<uint64_t> const & off, std::vector<uint64_t> const & len, int alph,
                            DeviceTable * keep = nullptr)
{
    uint64_t total = 0;
    for (uint64_t l : len)
        total += l;
    if (total == 0 || total > 0x7fffffffull || off.size() >= 0xffffffffull) // (lx_sort_words_dev ranks with 32 bits; no test covers lists beyond 2^31 words)
        return false;
    for (uint64_t l : len)
        if (l >= 0xffffffffull)
            return false;
    LXS_HIP(hipSetDevice(device));
    size_t freeB = 0, totalB = 0;
    LXS_HIP(hipMemGetInfo(&freeB, &totalB));
    if ((double)total * 72.0 + (double)red.size() + (64 << 20) > (double)freeB) // keys + values twice, entries, sort workspace
        return false;
    ix.prepareExternal(red, off, len, alph);
    size_t const          nSeq = off.size();
    std::vector<uint64_t> first(nSeq + 1, 0);
    for (size_t s = 0; s < nSeq; ++s)
        first[s + 1] = first[s] + len[s];
    auto dev = [](size_t bytes) -> void *
    {
        void * p = nullptr;
        LXS_HIP(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        return p;
    };
    struct Free
    {
        std::vector<void *> ptrs;
        ~Free()
        {
            for (void * p : ptrs)
                if (p)
                    (void)hipFree(p);
        }
    } owned;
    auto take = [&](size_t bytes)
    {
        owned.ptrs.push_back(dev(bytes));
        return owned.ptrs.back();
    };
    uint8_t *  dRed   = static_cast<uint8_t *>(take(red.size() + 64));
    uint64_t * dOff   = static_cast<uint64_t *>(take(nSeq * 8));
    uint64_t * dLen   = static_cast<uint64_t *>(take(nSeq * 8));
    uint64_t * dFirst = static_cast<uint64_t *>(take((nSeq + 1) * 8));
    uint64_t * k0 = static_cast<uint64_t *>(take(total * 8)), * k1 = static_cast<uint64_t *>(take(total * 8));
    uint64_t * v0 = static_cast<uint64_t *>(take(total * 8)), * v1 = static_cast<uint64_t *>(take(total * 8));
    LXS_HIP(hipMemcpy(dRed, red.data(), red.size(), hipMemcpyHostToDevice));
    LXS_HIP(hipMemcpy(dOff, off.data(), nSeq * 8, hipMemcpyHostToDevice));
    LXS_HIP(hipMemcpy(dLen, len.data(), nSeq * 8, hipMemcpyHostToDevice));
    LXS_HIP(hipMemcpy(dFirst, first.data(), (nSeq + 1) * 8, hipMemcpyHostToDevice));
    int const      keyLen = ix.keyLen(), preLen = ix.prefixLen();
    uint64_t const base = (uint64_t)alph + 1;
    unsigned const blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(table_keys_kernel, dim3(blocks), dim3(256), 0, 0, dRed, dOff, dLen, dFirst, (uint64_t)nSeq, total, keyLen, base, alph, k0, v0);
    LXS_HIP(hipGetLastError());
    int bits = 1;
    while (bits < 64 && (ix.power(keyLen) - 1) >> bits)
        ++bits;
    {
        // the library's own radix sort (lx_level2.hip through include/lambda_ext.h); whichever buffer ends up sorted becomes k1 / v1
        uint64_t * kk[2] = {k0, k1}, * vv[2] = {v0, v1};
        int        where = 0;
        if (lx_sort_words_dev(device, kk, vv, total, bits >= 64 ? ~0ull : ((1ull << bits) - 1), nullptr, &where) != LX_OK)
            throw std::runtime_error("lx_sort_words_dev failed on the word table");
        if (where == 0)
        {
            std::swap(k0, k1);
            std::swap(v0, v1);
        }
    }
    ReducedIndex::Entry * dEntries = static_cast<ReducedIndex::Entry *>(take(total * sizeof(ReducedIndex::Entry)));
    hipLaunchKernelGGL(table_entries_kernel, dim3(blocks), dim3(256), 0, 0, k1, v1, total, dEntries);
    LXS_HIP(hipGetLastError());
    uint64_t const nPre = ix.prefixCount();
    uint64_t *     dPre = static_cast<uint64_t *>(take(nPre * 8));
    hipLaunchKernelGGL(table_prefix_kernel, dim3((unsigned)((nPre + 255) / 256)), dim3(256), 0, 0, k1, total, ix.power(keyLen - preLen), nPre, dPre);
    LXS_HIP(hipGetLastError());
    LXS_HIP(hipMemcpy(ix.entriesForFill(), dEntries, total * sizeof(ReducedIndex::Entry), hipMemcpyDeviceToHost));
    LXS_HIP(hipMemcpy(ix.prefixForFill(), dPre, nPre * 8, hipMemcpyDeviceToHost));
    if (keep) // the entries and the prefix table stay where they are for the seeding stage
    {
        for (void *& p : owned.ptrs)
            if (p == dEntries || p == dPre)
                p = nullptr;
        keep->device = device, keep->entries = dEntries, keep->pre = dPre;
    }
    return true;
}

} // namespace lambda_amd

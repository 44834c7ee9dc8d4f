// lx_seeding.hpp -- the seeding stage of the minimal lambda3 front end, host-only C++.
//
// What the reference does per query batch (search(), /root/reference/src/search_algo.hpp:611-762): every seedOffset letters
// a seed of seedLength letters of the REDUCED query (Li-10 for proteins, src/mkindex_options.hpp:182-185) is searched in an
// FM-index over the reduced database -- exactly, or "half exact" (first half exact, the second half with up to maxSeedDist
// substitutions, searchHalfExactImpl :537-604) -- the hit set of every cursor is thinned by ADAPTIVE ELONGATION (:679-727: the
// seed grows to the right while that does not push the number of occurrences under what the query still needs to reach
// maxMatches), over-abundant cursors are dropped (:729), and every located hit passes seedLooksPromising (:426-481) before it
// becomes a Match.  Queries without any result after extension are searched again with the second parameter set
// (iterativeSearch, :1391-1457; defaults src/search_options.hpp:309-337).
//
// What is different here: there is no FM-index (fmindex-collection is absent and out of scope).  The same questions -- how
// often does this reduced word occur, where -- are answered by a sorted table of packed words: every database position
// carries the key of its next kKeyLen reduced letters (base alphabet + 1, the extra digit pads sequence ends), a cursor is a
// range of that table, extendRight narrows it by binary search.  Beyond kKeyLen letters (18 for Li-10, 27 for nucleotides) the
// table's order says nothing any more: a cursor then carries the list of its entries and extendRight filters it by the next
// reduced letter of every occurrence -- so adaptive elongation goes on as far as the reference's does (:703-721) and the hit
// sets equal the FM-index's for words of any length.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "../../../include/lambda_ext.h"

namespace lambda_amd
{

// bio::alphabet::aa10li over SeqAn AminoAcid ranks "ABCDEFGHIJKLMNOPQRSTUVWYZX*": Li et al. 2003, {AST} {BDEQZ} {CU} {FWY} {G}
// {HN} {IV} {JLM} {KOR} {P}; X joins the first group, '*' the aromatic one as in BioC++'s table [UPSTREAM-RECALL]
inline constexpr uint8_t kLi10[27] = {/*A*/ 0, /*B*/ 1, /*C*/ 2, /*D*/ 1, /*E*/ 1, /*F*/ 3, /*G*/ 4, /*H*/ 5, /*I*/ 6, /*J*/ 7,
                                      /*K*/ 8, /*L*/ 7, /*M*/ 7, /*N*/ 5, /*O*/ 8, /*P*/ 9, /*Q*/ 1, /*R*/ 8, /*S*/ 0, /*T*/ 0,
                                      /*U*/ 2, /*V*/ 6, /*W*/ 3, /*Y*/ 3, /*Z*/ 1, /*X*/ 0, /***/ 3};
// bio::alphabet::aa10murphy (Murphy et al. 2000): {ILMVJ} {CU} {AX*} {G} {ST} {P} {FYW} {EDNQBZ} {KRO} {H}
inline constexpr uint8_t kMurphy10[27] = {/*A*/ 2, /*B*/ 7, /*C*/ 1, /*D*/ 7, /*E*/ 7, /*F*/ 6, /*G*/ 3, /*H*/ 9, /*I*/ 0, /*J*/ 0,
                                          /*K*/ 8, /*L*/ 0, /*M*/ 0, /*N*/ 7, /*O*/ 8, /*P*/ 5, /*Q*/ 7, /*R*/ 8, /*S*/ 4, /*T*/ 4,
                                          /*U*/ 1, /*V*/ 0, /*W*/ 6, /*Y*/ 6, /*Z*/ 7, /*X*/ 2, /***/ 2};
// BioC++ dna5 ranks (A, C, G, N, T) -> dna4 (A, C, G, T); N converts to A like every non-dna4 letter does
inline constexpr uint8_t kDna4[5] = {0, 1, 2, 0, 3};

// views::reduce_to_bisulfite (src/view_reduce_to_bisulfite.hpp:44-52) over SeqAn Dna5 ranks (A, C, G, T, N): even frames take the
// forward reduction (C and T fall together: ranks 0, 1, 2, 1), odd frames the reverse one (G and A: ranks 3, 4, 3, 5) -- one
// alphabet of six letters, so that a forward word never matches a reverse subject.  N converts like A (the reference draws a
// random letter for it, src/view_dna_n_to_random.hpp).
inline constexpr uint8_t kBsFwd[5] = {0, 1, 2, 1, 0};
inline constexpr uint8_t kBsRev[5] = {3, 4, 3, 5, 3};

// f(chunk) for chunk = 0 .. nChunks-1 on up to nThreads host threads, chunks handed out in order (the reference spreads its
// batches over OpenMP threads the same way, src/search.cpp:379-385: schedule(dynamic))
template <typename F>
inline void parallelChunks(unsigned nThreads, size_t nChunks, F && f)
{
    if (nThreads <= 1 || nChunks <= 1)
    {
        for (size_t c = 0; c < nChunks; ++c)
            f(c);
        return;
    }
    std::atomic<size_t> next{0};
    auto                run = [&]()
    {
        for (size_t c = next.fetch_add(1); c < nChunks; c = next.fetch_add(1))
            f(c);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < std::min<size_t>(nThreads, nChunks); ++t)
        pool.emplace_back(run);
    run();
    for (std::thread & t : pool)
        t.join();
}

// n elements of a trivial type, NOT initialised (std::vector would write 16 bytes per database residue on one thread before the
// threads that fill the table get to touch it)
template <typename T>
class PlainBuffer
{
    T *    p_ = nullptr;
    size_t n_ = 0;

public:
    PlainBuffer() = default;
    PlainBuffer(PlainBuffer const &) = delete;
    PlainBuffer & operator=(PlainBuffer const &) = delete;
    ~PlainBuffer() { ::operator delete(p_); }
    void resize(size_t n) // (contents are lost)
    {
        ::operator delete(p_);
        p_ = nullptr;
        n_ = 0;
        if (n)
            p_ = static_cast<T *>(::operator new(n * sizeof(T)));
        n_ = n;
    }
    T *       data() { return p_; }
    T const * data() const { return p_; }
    size_t    size() const { return n_; }
    T *       begin() { return p_; }
    T *       end() { return p_ + n_; }
    T const * begin() const { return p_; }
    T const * end() const { return p_ + n_; }
    T &       operator[](size_t i) { return p_[i]; }
    T const & operator[](size_t i) const { return p_[i]; }
};

struct SeedParams // SearchOptions' seeding part, src/search_options.hpp:309-337
{
    int seedLength = 10, seedOffset = 5, maxSeedDist = 0;
};

class ReducedIndex
{
public:
    struct Cursor
    {
        uint64_t lo = 0, hi = 0; // range of the sorted table
        int      len = 0;        // letters matched
        uint64_t prefix = 0;     // the word so far, base (alph + 1)
        bool                  listed = false; // the word is longer than the table's keys: `sel` lists the entries that match it
        std::vector<uint32_t> sel;
        uint64_t count() const { return listed ? sel.size() : hi - lo; }
        bool     empty() const { return count() == 0; }
    };

    // red = reduced residues of all (frame-expanded) subject sequences, off/len per sequence; alph = reduced alphabet size.
    // The table is made on nThreads host threads: keys per sequence, a scatter by the words' first kTopLen letters, every bucket
    // sorted on its own -- by (key, sequence, position), so that the table does not depend on the number of threads.
    void build(std::vector<uint8_t> const & red, std::vector<uint64_t> const & off, std::vector<uint64_t> const & len, int alph, unsigned nThreads = 1)
    {
        red_    = red.data();
        off_    = off.data();
        len_    = len.data();
        alph_   = alph;
        base_   = (uint64_t)alph + 1;
        keyLen_ = 0;
        for (uint64_t lim = ~0ull / 2, p = 1; p <= lim / base_; p *= base_)
            ++keyLen_;
        pow_.assign(keyLen_ + 1, 1);
        for (int i = 1; i <= keyLen_; ++i)
            pow_[i] = pow_[i - 1] * base_;
        size_t const          nSeq = off.size();
        std::vector<uint64_t> first(nSeq + 1, 0); // entry index of every sequence's first position
        for (size_t s = 0; s < nSeq; ++s)
            first[s + 1] = first[s] + len[s];
        uint64_t const total = first[nSeq];
        int const      kTopLen = std::min(3, keyLen_ - 1);
        uint64_t const topDiv = pow_[keyLen_ - kTopLen];
        size_t const   nBuckets = (size_t)pow_[kTopLen];
        PlainBuffer<Entry> raw;
        raw.resize(total);
        // contiguous ranges of sequences of about equal residue counts
        size_t const        nParts = std::max<size_t>(1, std::min<size_t>(nSeq, (size_t)nThreads * 4));
        std::vector<size_t> cut(nParts + 1, nSeq);
        cut[0] = 0;
        for (size_t part = 1, s = 0; part < nParts; ++part)
        {
            while (s < nSeq && first[s] < total * part / nParts)
                ++s;
            cut[part] = s;
        }
#ifdef LX_SEED_BUILD_TIMING // (development aid, tools/dev/seed_bench.cpp: where the table's time goes)
        auto tMark = std::chrono::steady_clock::now();
        auto mark  = [&](char const * what)
        {
            auto const now = std::chrono::steady_clock::now();
            std::printf("  table: %s %.0f ms\n", what, std::chrono::duration<double, std::milli>(now - tMark).count());
            tMark = now;
        };
#else
        auto mark = [](char const *) {};
#endif
        std::vector<std::vector<uint64_t>> counts(nParts, std::vector<uint64_t>(nBuckets, 0));
        parallelChunks(nThreads, nParts,
                       [&](size_t part)
                       {
                           std::vector<uint64_t> & cnt = counts[part];
                           for (size_t s = cut[part]; s < cut[part + 1]; ++s)
                           {
                               // rolling key of the next keyLen_ letters, pad digit `alph` beyond the sequence end
                               uint64_t const L = len[s];
                               if (L == 0)
                                   continue;
                               uint64_t key = 0;
                               for (int i = 0; i < keyLen_; ++i)
                                   key = key * base_ + ((uint64_t)i < L ? red[off[s] + i] : (uint64_t)alph);
                               Entry * out = raw.data() + first[s];
                               for (uint64_t p = 0; p < L; ++p)
                               {
                                   out[p] = Entry{key, (uint32_t)s, (uint32_t)p};
                                   ++cnt[key / topDiv];
                                   uint64_t const next = p + keyLen_ < L ? red[off[s] + p + keyLen_] : (uint64_t)alph;
                                   key                 = (key % pow_[keyLen_ - 1]) * base_ + next;
                               }
                           }
                       });
        mark("keys + counts");
        // where every (bucket, part) starts in the sorted table
        std::vector<uint64_t> bucketAt(nBuckets + 1, 0);
        for (size_t b = 0, at = 0; b < nBuckets; ++b)
        {
            bucketAt[b] = at;
            for (size_t part = 0; part < nParts; ++part)
            {
                uint64_t const c = counts[part][b];
                counts[part][b]  = at;
                at += c;
            }
            bucketAt[b + 1] = at;
        }
        entries_.resize(total);
        parallelChunks(nThreads, nParts,
                       [&](size_t part)
                       {
                           std::vector<uint64_t> & at = counts[part];
                           for (uint64_t e = first[cut[part]]; e < first[cut[part + 1]]; ++e)
                               entries_[at[raw[e].key / topDiv]++] = raw[e];
                       });
        mark("scatter");
        raw.resize(0);
        // the big buckets first, so that the last threads do not start one when the others are done
        std::vector<uint32_t> order(nBuckets);
        for (size_t b = 0; b < nBuckets; ++b)
            order[b] = (uint32_t)b;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return bucketAt[a + 1] - bucketAt[a] > bucketAt[b + 1] - bucketAt[b]; });
        // a bucket (the words' first kTopLen letters) is dealt once more by the next kTopLen letters -- it fits the caches --, and what
        // is left to compare is a handful of entries per piece (one sort of the whole bucket costs three times as much)
        int const      subLen = std::min(kTopLen, keyLen_ - kTopLen);
        uint64_t const subDiv = pow_[keyLen_ - kTopLen - subLen], subMod = pow_[subLen];
        auto const     less   = [](Entry const & x, Entry const & y) { return x.key != y.key ? x.key < y.key : x.seq != y.seq ? x.seq < y.seq : x.pos < y.pos; };
        parallelChunks(nThreads, nBuckets,
                       [&](size_t k)
                       {
                           uint32_t const b  = order[k];
                           Entry * const  lo = entries_.begin() + bucketAt[b];
                           size_t const   n  = (size_t)(bucketAt[b + 1] - bucketAt[b]);
                           if (n < 64 || subLen < 1)
                           {
                               std::sort(lo, lo + n, less);
                               return;
                           }
                           std::vector<uint32_t> at((size_t)subMod + 1, 0);
                           for (size_t e = 0; e < n; ++e)
                               ++at[(size_t)((lo[e].key / subDiv) % subMod) + 1];
                           for (size_t d = 0; d < (size_t)subMod; ++d)
                               at[d + 1] += at[d];
                           std::vector<uint32_t> const first(at);
                           PlainBuffer<Entry>          tmp;
                           tmp.resize(n);
                           for (size_t e = 0; e < n; ++e)
                               tmp[at[(size_t)((lo[e].key / subDiv) % subMod)]++] = lo[e];
                           for (size_t d = 0; d < (size_t)subMod; ++d)
                               if (first[d + 1] - first[d] > 1)
                                   std::sort(tmp.begin() + first[d], tmp.begin() + first[d + 1], less);
                           std::copy(tmp.begin(), tmp.end(), lo);
                       });
        mark("bucket sorts");
        // where the words with every prefix of preLen_ letters begin: the first letters of a seed cost one table read each instead
        // of two binary searches over the whole table (the probes that miss every cache)
        preLen_ = 1;
        while (preLen_ + 1 < keyLen_ && pow_[preLen_ + 1] <= std::min<uint64_t>(4u << 20, std::max<uint64_t>(1024, total)))
            ++preLen_;
        uint64_t const preDiv = pow_[keyLen_ - preLen_];
        pre_.assign((size_t)pow_[preLen_] + 1, total);
        size_t const nCuts = std::max<size_t>(1, (size_t)nThreads * 4);
        parallelChunks(nThreads, nCuts,
                       [&](size_t k)
                       {
                           uint64_t const lo = total * k / nCuts, hi = total * (k + 1) / nCuts;
                           // pre_[w] = first entry whose word is >= w: every w in (word of entry e-1, word of entry e] begins at e
                           uint64_t prev = lo == 0 ? 0 : entries_[lo - 1].key / preDiv + 1;
                           for (uint64_t e = lo; e < hi; ++e)
                           {
                               uint64_t const w = entries_[e].key / preDiv;
                               for (; prev <= w; ++prev)
                                   pre_[prev] = e;
                           }
                       });
        mark("prefix table");
    }

    // For a table made elsewhere (the GPU builder, host/lx_seeding_gpu.hpp): the geometry build() would choose for this database,
    // then room for the entries and the prefix table, which the caller fills -- sorted by (word, sequence, position) -- before
    // the first search.  total = number of residues.
    void prepareExternal(std::vector<uint8_t> const & red, std::vector<uint64_t> const & off, std::vector<uint64_t> const & len, int alph)
    {
        red_    = red.data();
        off_    = off.data();
        len_    = len.data();
        alph_   = alph;
        base_   = (uint64_t)alph + 1;
        keyLen_ = 0;
        for (uint64_t lim = ~0ull / 2, p = 1; p <= lim / base_; p *= base_)
            ++keyLen_;
        pow_.assign(keyLen_ + 1, 1);
        for (int i = 1; i <= keyLen_; ++i)
            pow_[i] = pow_[i - 1] * base_;
        uint64_t total = 0;
        for (uint64_t l : len)
            total += l;
        preLen_ = 1;
        while (preLen_ + 1 < keyLen_ && pow_[preLen_ + 1] <= std::min<uint64_t>(4u << 20, std::max<uint64_t>(1024, total)))
            ++preLen_;
        entries_.resize(total);
        pre_.resize((size_t)pow_[preLen_] + 1);
    }

    int    keyLen() const { return keyLen_; }
    Cursor root() const { return Cursor{0, entries_.size(), 0, 0}; }

    // the cursor of word + c; empty when the word does not occur (or the table's word length is exhausted)
    Cursor extendRight(Cursor const & cu, uint8_t c) const
    {
        if (cu.len >= keyLen_)
        {
            // beyond the keys: keep the occurrences whose next reduced letter is c (the sequences outlive the index)
            Cursor n;
            n.lo = cu.lo, n.hi = cu.hi, n.len = cu.len + 1, n.prefix = cu.prefix, n.listed = true;
            auto keep = [&](uint32_t e)
            {
                Entry const & x = entries_[e];
                if ((uint64_t)x.pos + (uint64_t)cu.len < len_[x.seq] && red_[off_[x.seq] + x.pos + (uint64_t)cu.len] == c)
                    n.sel.push_back(e);
            };
            if (cu.listed)
                for (uint32_t e : cu.sel)
                    keep(e);
            else
                for (uint64_t e = cu.lo; e < cu.hi; ++e)
                    keep((uint32_t)e);
            return n;
        }
        Cursor n = cu;
        n.prefix = cu.prefix * base_ + c;
        n.len    = cu.len + 1;
        if (n.len <= preLen_)
        {
            uint64_t const span = pow_[preLen_ - n.len];
            n.lo                = pre_[n.prefix * span];
            n.hi                = pre_[(n.prefix + 1) * span];
            return n;
        }
        uint64_t const scale = pow_[keyLen_ - n.len], first = n.prefix * scale, last = first + (scale - 1);
        if (cu.hi - cu.lo <= 16) // a handful of neighbouring entries: one pass over them instead of two binary searches
        {
            uint64_t a = cu.lo;
            while (a < cu.hi && entries_[a].key < first)
                ++a;
            uint64_t z = a;
            while (z < cu.hi && entries_[z].key <= last)
                ++z;
            n.lo = a, n.hi = z;
            return n;
        }
        auto const     b = entries_.begin() + (std::ptrdiff_t)cu.lo, e = entries_.begin() + (std::ptrdiff_t)cu.hi;
        n.lo = (uint64_t)(std::lower_bound(b, e, first, [](Entry const & x, uint64_t k) { return x.key < k; }) - entries_.begin());
        n.hi = (uint64_t)(std::lower_bound(entries_.begin() + (std::ptrdiff_t)n.lo, e, last + 1, [](Entry const & x, uint64_t k) { return x.key < k; }) - entries_.begin());
        return n;
    }

    // the cursors of word + r for every letter r = 0 .. alph-1 (out[r]; empty ones included): the children of a table range lie
    // side by side in letter order, so one boundary search per letter does (a scan when the range is short) instead of two
    void extendAll(Cursor const & cu, Cursor * out) const
    {
        int const nl = cu.len + 1;
        if (cu.len >= keyLen_ || nl <= preLen_)
        {
            for (int r = 0; r < alph_; ++r)
                out[r] = extendRight(cu, (uint8_t)r);
            return;
        }
        uint64_t const scale = pow_[keyLen_ - nl], word0 = cu.prefix * base_;
        uint64_t       at    = cu.lo;
        for (int r = 0; r <= alph_; ++r)
        {
            // first entry of [at, hi) whose word of nl letters is >= word0 + r
            uint64_t const first = (word0 + (uint64_t)r) * scale;
            if (cu.hi - at <= 24)
                while (at < cu.hi && entries_[at].key < first)
                    ++at;
            else
                at = (uint64_t)(std::lower_bound(entries_.begin() + (std::ptrdiff_t)at, entries_.begin() + (std::ptrdiff_t)cu.hi, first,
                                                 [](Entry const & x, uint64_t k) { return x.key < k; }) -
                                entries_.begin());
            if (r > 0)
                out[r - 1].hi = at;
            if (r < alph_)
            {
                out[r].lo = at, out[r].len = nl, out[r].prefix = word0 + (uint64_t)r, out[r].listed = false;
                out[r].sel.clear();
            }
        }
    }

    template <typename F>
    void locate(Cursor const & cu, F && f) const // f(subject sequence, offset)
    {
        if (cu.listed)
        {
            for (uint32_t e : cu.sel)
                f(entries_[e].seq, entries_[e].pos);
            return;
        }
        for (uint64_t i = cu.lo; i < cu.hi; ++i)
            f(entries_[i].seq, entries_[i].pos);
    }

    // The table as it stands, for an index file (lambda3 mkindex*): geometry, entries, prefix table.  load() takes the
    // reduced residues and the sequence table the entries refer to (they are not part of the table) and checks the sizes.
    template <typename Write>
    void save(Write && write) const // write(pointer, bytes)
    {
        int32_t const  geo[4] = {alph_, keyLen_, preLen_, 0};
        uint64_t const cnt[2] = {entries_.size(), pre_.size()};
        write(geo, sizeof(geo));
        write(cnt, sizeof(cnt));
        write(entries_.data(), entries_.size() * sizeof(Entry));
        write(pre_.data(), pre_.size() * sizeof(uint64_t));
    }
    template <typename Read>
    bool load(Read && read, std::vector<uint8_t> const & red, std::vector<uint64_t> const & off, std::vector<uint64_t> const & len) // read(pointer, bytes) -> bool
    {
        int32_t  geo[4];
        uint64_t cnt[2];
        if (!read(geo, sizeof(geo)) || !read(cnt, sizeof(cnt)))
            return false;
        uint64_t total = 0;
        for (uint64_t l : len)
            total += l;
        alph_ = geo[0], keyLen_ = geo[1], preLen_ = geo[2];
        base_ = (uint64_t)alph_ + 1;
        if (alph_ < 2 || alph_ > 31 || keyLen_ < 2 || keyLen_ > 63 || preLen_ < 1 || preLen_ >= keyLen_ || cnt[0] != total)
            return false;
        pow_.assign(keyLen_ + 1, 1);
        for (int i = 1; i <= keyLen_; ++i)
        {
            if (pow_[i - 1] > (~0ull / 2) / base_)
                return false;
            pow_[i] = pow_[i - 1] * base_;
        }
        if (cnt[1] != pow_[preLen_] + 1)
            return false;
        entries_.resize(cnt[0]);
        pre_.resize(cnt[1]);
        if (!read(entries_.data(), entries_.size() * sizeof(Entry)) || !read(pre_.data(), pre_.size() * sizeof(uint64_t)))
            return false;
        // (cheap sanity of what the searches rely on: the prefix table ascends to the entry count, the entries name sequences that exist)
        if (pre_.back() != total || pre_.front() != 0)
            return false;
        for (size_t w = 1; w < pre_.size(); ++w)
            if (pre_[w] < pre_[w - 1])
                return false;
        for (Entry const & e : entries_)
            if (e.seq >= len.size() || e.pos >= len[e.seq])
                return false;
        red_ = red.data(), off_ = off.data(), len_ = len.data();
        return true;
    }
    int alphabet() const { return alph_; }

    struct Entry
    {
        uint64_t key;
        uint32_t seq, pos;
    };
    // the table's parts as they stand (what the GPU seeding stage uploads, host/lx_seeding_gpu.hpp)
    Entry const *    entriesData() const { return entries_.data(); }
    uint64_t         entriesCount() const { return entries_.size(); }
    uint64_t const * prefixData() const { return pre_.data(); }
    uint64_t         prefixCount() const { return pre_.size(); }
    int              prefixLen() const { return preLen_; }
    uint64_t         power(int i) const { return pow_[(size_t)i]; }
    Entry *          entriesForFill() { return entries_.data(); } // (after prepareExternal)
    uint64_t *       prefixForFill() { return pre_.data(); }

private:
    PlainBuffer<Entry>    entries_;
    std::vector<uint64_t> pow_, pre_; // pre_[w] = first entry whose first preLen_ letters are >= the word w
    int                   preLen_ = 0;
    uint8_t const *       red_ = nullptr; // the caller's reduced residues and sequence table (must outlive the index)
    uint64_t const *      off_ = nullptr;
    uint64_t const *      len_ = nullptr;
    uint64_t              base_ = 11;
    int                   alph_ = 10, keyLen_ = 18;
};

// search_impl with maxSeedDist = 0 (:505-535): the exact word
inline void searchExact(ReducedIndex const & ix, uint8_t const * seed, int seedLength, std::vector<ReducedIndex::Cursor> & out)
{
    ReducedIndex::Cursor c = ix.root();
    for (int i = 0; i < seedLength; ++i)
    {
        c = ix.extendRight(c, seed[i]);
        if (c.empty())
            return;
    }
    out.push_back(c);
}

// searchHalfExactImpl (:537-604): first half exact, then letter by letter every cursor below the error budget branches into all
// letters of the alphabet (a different letter costs one error), the others continue with the seed's letter
// exactPrefix = seedLength / 2 is the half-exact search; 0 = substitutions anywhere in the seed, what the reference runs with
// --seed-half-exact 0 (search_one_error / search_pseudo over the whole seed, :486-531)
inline void searchHalfExact(ReducedIndex const & ix, uint8_t const * seed, int seedLength, int maxSeedDist, int alph,
                            std::vector<ReducedIndex::Cursor> & out, int exactPrefix = -1)
{
    int const firstHalf = exactPrefix < 0 ? seedLength / 2 : std::min(exactPrefix, seedLength), secondHalf = seedLength - firstHalf;
    std::vector<std::pair<ReducedIndex::Cursor, int>> cur, nxt;
    ReducedIndex::Cursor                              c = ix.root();
    for (int i = 0; i < firstHalf; ++i)
    {
        c = ix.extendRight(c, seed[i]);
        if (c.empty())
            return;
    }
    cur.emplace_back(c, 0);
    for (int i = 0; i < secondHalf; ++i)
    {
        uint8_t const want = seed[firstHalf + i];
        nxt.clear();
        for (auto const & [cursor, errors] : cur)
        {
            if (errors < maxSeedDist)
            {
                ReducedIndex::Cursor kids[32];
                ix.extendAll(cursor, kids);
                for (int r = 0; r < alph; ++r)
                    if (!kids[r].empty())
                        nxt.emplace_back(kids[r], errors + ((uint8_t)r != want));
            }
            else
            {
                ReducedIndex::Cursor n = ix.extendRight(cursor, want);
                if (!n.empty())
                    nxt.emplace_back(n, errors);
            }
        }
        cur.swap(nxt);
    }
    for (auto const & [cursor, errors] : cur)
        out.push_back(cursor);
}

// seedLooksPromising (:426-481) on the host, with the reference's integer types and order of operations: the seeding loop asks
// it per located hit (hitsThisSeq feeds the elongation of the NEXT seeds), so it cannot wait for a batched GPU pass here.
// q / s = the whole (frame) sequences in alignment ranks; matrix[q_rank * 32 + s_rank].
inline bool seedLooksPromising(uint8_t const * q, uint64_t qLen, uint8_t const * s, uint64_t sLen, lx_match const & m, int seedLength,
                               int preScoring, double preScoringThresh, int8_t const * matrix)
{
    int64_t  qFrom = (int64_t)m.qryStart, sFrom = (int64_t)m.subjStart;
    uint64_t seedSpan    = m.qryEnd - m.qryStart;
    uint64_t span = std::max<uint64_t>((uint64_t)(seedLength * preScoring), seedSpan);
    if (span > seedSpan)
    {
        qFrom -= (int64_t)((span - seedSpan) / 2);
        sFrom -= (int64_t)((span - seedSpan) / 2);
        int64_t const mn = std::min(qFrom, sFrom);
        if (mn < 0)
        {
            qFrom -= mn;
            sFrom -= mn;
            span += (uint64_t)mn; // (unsigned wrap-around of a negative addend, as in the reference)
        }
        span = std::min({(uint64_t)(qLen - (uint64_t)qFrom), (uint64_t)(sLen - (uint64_t)sFrom), span});
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

struct SeedingStats
{
    uint64_t hitsAfterSeeding = 0, hitsFailedPreExtendTest = 0;
};

struct SeedingInput
{
    // queries: alignment ranks and reduced letters of every (frame-expanded) sequence, frames of one read adjacent
    uint8_t const *  qRes;
    uint8_t const *  qRed;
    uint64_t const * qOff;
    uint64_t const * qLen;
    uint64_t         nQSeq;
    int              qNumFrames;
    uint8_t          unknownRank; // 'X' (proteins) / 'N' (nucleotides) in alignment ranks: seeds do not start there (:655-660)
    uint8_t const *  sRes;        // subjects in alignment ranks
    uint64_t const * sOff;
    uint64_t const * sLen;
    int              alph;        // reduced alphabet size
    int8_t const *   matrix;
    int8_t const *   matrixRev = nullptr; // bisulfite: the reverse scheme, used for hits on odd subject frames (:464-466)
    uint64_t         maxMatches;
    bool             halfExact, adaptive;
    int              preScoring;
    double           preScoringThresh;
};

// search(lH), :611-762, for the (frame-expanded) query sequences listed in `which` (all frames of a read, in order)
inline void seedQueries(ReducedIndex const & ix, SeedingInput const & in, SeedParams const & so, std::vector<uint64_t> const & which,
                        std::vector<lx_match> & matches, SeedingStats & stats)
{
    size_t           hitsThisSeq = 0, needlesSum = 0, needlesPos = 0;
    constexpr size_t kOccFactor = 10; // :629
    std::vector<ReducedIndex::Cursor> cursors;
    for (size_t w = 0; w < which.size(); ++w)
    {
        uint64_t const i = which[w];
        // reset on every "real" new read (:643-650).  The reference tests the length first (:640-641) and so would carry the previous
        // read's counts into a read whose FIRST frame is shorter than the seed while a later one is not -- no frame layout produces
        // that (frame 0 is never the shortest), and a read's seeds must not depend on its neighbours here: the reads are dealt to
        // threads and to GPU lanes, every one starts from zero
        if (i % (uint64_t)in.qNumFrames == 0)
        {
            hitsThisSeq = needlesSum = needlesPos = 0;
            for (int j = 0; j < in.qNumFrames && i + (uint64_t)j < in.nQSeq; ++j)
                needlesSum += in.qLen[i + (uint64_t)j];
        }
        if (in.qLen[i] < (uint64_t)so.seedLength) // :640-641 (without the bookkeeping below, as there)
            continue;
        uint64_t const        L   = in.qLen[i];
        uint8_t const * const red = in.qRed + in.qOff[i];
        uint8_t const * const res = in.qRes + in.qOff[i];
        {
            for (uint64_t seedBegin = 0;; seedBegin += (uint64_t)so.seedOffset)
            {
                // skip the unknown letter, skip a letter whose successor is the same (:655-660)
                while (seedBegin < L - (uint64_t)so.seedLength && (res[seedBegin] == in.unknownRank || res[seedBegin] == res[seedBegin + 1]))
                    ++seedBegin;
                if (seedBegin > L - (uint64_t)so.seedLength) // :663
                    break;
                cursors.clear();
                if (so.maxSeedDist != 0)
                    searchHalfExact(ix, red + seedBegin, so.seedLength, so.maxSeedDist, in.alph, cursors, in.halfExact ? -1 : 0);
                else
                    searchExact(ix, red + seedBegin, so.seedLength, cursors);
                for (ReducedIndex::Cursor cursor : cursors)
                {
                    uint64_t seedLength = (uint64_t)so.seedLength;
                    if (in.adaptive) // :683-727, the deterministic branch
                    {
                        size_t desiredOccs = hitsThisSeq >= in.maxMatches
                                               ? 1
                                               : (in.maxMatches - hitsThisSeq) * kOccFactor /
                                                   std::max<size_t>((needlesSum - needlesPos - seedBegin) / (size_t)so.seedOffset, 1ul);
                        if (desiredOccs == 0)
                            desiredOccs = 1;
                        ReducedIndex::Cursor old_cursor = cursor;
                        size_t               old_count  = cursor.count();
                        while (seedBegin + seedLength < L) // (:703: until the count drops under desiredOccs or the read ends)
                        {
                            cursor                 = ix.extendRight(cursor, red[seedBegin + seedLength]);
                            size_t const new_count = cursor.count();
                            if (new_count < desiredOccs && new_count < old_count) // we always extend if we don't lose anything
                            {
                                cursor = old_cursor;
                                break;
                            }
                            ++seedLength;
                            old_count  = new_count;
                            old_cursor = cursor;
                        }
                    }
                    if (cursor.count() > kOccFactor * in.maxMatches) // over-abundant (:729)
                        continue;
                    ix.locate(cursor,
                              [&](uint32_t subjNo, uint32_t subjOffset)
                              {
                                  lx_match const m{i, subjNo, seedBegin, seedBegin + seedLength, subjOffset, subjOffset + seedLength};
                                  ++stats.hitsAfterSeeding;
                                  if (!seedLooksPromising(res, L, in.sRes + in.sOff[subjNo], in.sLen[subjNo], m, so.seedLength, in.preScoring,
                                                          in.preScoringThresh, (in.matrixRev && (subjNo & 1u)) ? in.matrixRev : in.matrix))
                                      ++stats.hitsFailedPreExtendTest;
                                  else
                                  {
                                      matches.push_back(m);
                                      ++hitsThisSeq;
                                  }
                              });
                }
            }
        }
        needlesPos += L; // :759
    }
}

// The same over nThreads host threads: reads are independent (every bookkeeping value of seedQueries is reset at a read's first
// frame), so `which` is cut at read boundaries, the pieces are searched side by side and their matches concatenated in order --
// the list one thread would have produced.
inline void seedQueriesParallel(ReducedIndex const & ix, SeedingInput const & in, SeedParams const & so, std::vector<uint64_t> const & which,
                                std::vector<lx_match> & matches, SeedingStats & stats, unsigned nThreads)
{
    size_t const n = which.size();
    if (nThreads <= 1 || n < 64)
    {
        seedQueries(ix, in, so, which, matches, stats);
        return;
    }
    size_t const        nPieces = std::min<size_t>((size_t)nThreads * 8, n / 16 + 1);
    std::vector<size_t> cut(nPieces + 1, n);
    cut[0] = 0;
    for (size_t k = 1; k < nPieces; ++k)
    {
        size_t w = std::max(cut[k - 1], n * k / nPieces);
        while (w < n && which[w] % (uint64_t)in.qNumFrames != 0)
            ++w;
        cut[k] = w;
    }
    std::vector<std::vector<lx_match>> part(nPieces);
    std::vector<SeedingStats>          pst(nPieces);
    parallelChunks(nThreads, nPieces,
                   [&](size_t k)
                   {
                       if (cut[k] >= cut[k + 1])
                           return;
                       std::vector<uint64_t> const mine(which.begin() + (std::ptrdiff_t)cut[k], which.begin() + (std::ptrdiff_t)cut[k + 1]);
                       seedQueries(ix, in, so, mine, part[k], pst[k]);
                   });
    size_t total = matches.size();
    for (auto const & v : part)
        total += v.size();
    matches.reserve(total);
    for (size_t k = 0; k < nPieces; ++k)
    {
        matches.insert(matches.end(), part[k].begin(), part[k].end());
        stats.hitsAfterSeeding += pst[k].hitsAfterSeeding;
        stats.hitsFailedPreExtendTest += pst[k].hitsFailedPreExtendTest;
    }
}

} // namespace lambda_amd

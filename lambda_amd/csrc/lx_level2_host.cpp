// lx_level2_host.cpp -- the Level-2 driver on a DEVICE match list: lx_set_queries, lx_set_subject_seqs, lx_iterate_matches_dev.
//
// iterateMatchesFullSimd (/root/reference/src/search_algo.hpp:1177-1332) for matches that the seeding stage left in HBM: widen,
// sort, merge and unique (:1136-1175), the slices (:1200-1227) and the filter's cut-offs (:1251-1283) run as kernels
// (lx_level2.hip) on the resident sequence sets; the host receives the finished window list (24 B per WINDOW, a tenth of the
// matches), plans the sweep over it, and the extension pipeline (lx_host.cpp) reads list and cut-offs where the kernels wrote
// them.  What follows the two passes -- statistics, order, records (:1287-1325) -- is host/lx_iterate_common.hpp, shared with
// lx_iterate_matches.  lx_iterate_matches itself hands its lists to this path when they are large (host/lx_driver.cpp).
#include <cmath>
#include <thread>

#include "lx_internal.h"
#include "lx_level2.h"

#include "host/lx_iterate_common.hpp"

using namespace lxi;

namespace
{

int ensure_pinned(lx_handle * h, lx_handle::Pinned & b, size_t bytes, unsigned flags = hipHostMallocDefault)
{
    if (bytes <= b.cap)
        return LX_OK;
    if (b.ptr)
    {
        LX_HIP(h, hipHostFree(b.ptr));
        b.ptr = nullptr;
        b.cap = 0;
    }
    size_t const want = bytes + bytes / 4 + 4096;
    LX_HIP(h, hipHostMalloc(&b.ptr, want, flags));
    b.cap = want;
    return LX_OK;
}

template <class T>
int upload(lx_handle * h, DevBuf & b, std::vector<T> const & v)
{
    int rc = ensure(h, b, v.size() * sizeof(T) + 16);
    if (rc)
        return rc;
    if (!v.empty())
        LX_HIP(h, hipMemcpyAsync(b.ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return LX_OK;
}

inline uint64_t bits_below(uint64_t count) // mask of the bits a value in [0, count) can have set
{
    if (count <= 1)
        return 0;
    uint64_t m = 1;
    while (m < count - 1 && m != ~0ull)
        m = (m << 1) | 1;
    return m;
}

// src/search_misc.hpp:46-50
inline int64_t bandSize(uint64_t const seqLength)
{
    return static_cast<int64_t>(std::sqrt(seqLength)) + 1;
}

} // namespace

extern "C" {

int lx_set_queries(lx_handle * h, uint8_t const * q_res, uint64_t q_bytes, uint64_t const * q_seq_off, uint64_t const * q_seq_len, uint64_t n_qseq,
                   uint64_t const * q_orig_len, int32_t qry_num_frames)
{
    if (!h)
        return LX_EINVAL;
    if ((!q_res && q_bytes) || ((!q_seq_off || !q_seq_len) && n_qseq))
        return fail(h, LX_EINVAL, "NULL argument");
    if (n_qseq >= (1ull << 31))
        return fail(h, LX_EINVAL, "at most 2^31 (frame) query sequences");
    int rc = bind(h);
    if (rc)
        return rc;
    auto & l2   = h->l2;
    int const f = std::max(1, qry_num_frames);
    // Validated into locals, committed to the handle only after the uploads succeeded: a failing call leaves NOTHING resident
    // (q_bytes 0, empty tables -- what the device entry points test), never host tables of the new set over device arrays of the old.
    auto drop = [&]()
    {
        l2.q_bytes = 0;
        l2.q_hash  = 0;
        l2.q_off.clear();
        l2.q_len.clear();
        l2.q_evlen.clear();
        l2.evlens.clear();
        l2.max_evlen = 0;
    };
    drop();
    std::vector<uint64_t> q_off(q_seq_off, q_seq_off + n_qseq);
    std::vector<uint32_t> q_len(n_qseq), q_evlen(n_qseq), band(n_qseq);
    for (uint64_t i = 0; i < n_qseq; ++i)
    {
        if (q_seq_len[i] > 0xffffffffull || !lx_slice_ok(q_seq_off[i], q_seq_len[i], q_bytes))
            return fail(h, LX_EINVAL, "query sequence %llu exceeds the residue buffer", (unsigned long long)i);
        uint64_t const ev = q_orig_len ? q_orig_len[i / (uint64_t)f] : q_seq_len[i];
        if (ev > (1ull << 26))
            return fail(h, LX_EINVAL, "query %llu: a length of %llu is beyond the cut-off table", (unsigned long long)i, (unsigned long long)ev);
        q_len[i]   = (uint32_t)q_seq_len[i];
        q_evlen[i] = (uint32_t)ev;
        band[i]    = (uint32_t)bandSize(q_seq_len[i]);
    }
    std::vector<uint32_t> evlens = q_evlen;
    std::sort(evlens.begin(), evlens.end());
    evlens.erase(std::unique(evlens.begin(), evlens.end()), evlens.end());
    // (for the records kernel, lx_records.hip: which of the distinct e-value lengths a query has; the longest query bounds a sort digit)
    std::vector<uint32_t> evidx(n_qseq);
    uint32_t              max_qlen = 0;
    for (uint64_t i = 0; i < n_qseq; ++i)
    {
        evidx[i] = (uint32_t)(std::lower_bound(evlens.begin(), evlens.end(), q_evlen[i]) - evlens.begin());
        max_qlen = std::max(max_qlen, q_len[i]);
    }
    if ((rc = ensure(h, l2.d_qres, q_bytes + kSlack)))
        return rc;
    if (q_bytes)
        LX_HIP(h, hipMemcpyAsync(l2.d_qres.ptr, q_res, q_bytes, hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemsetAsync(static_cast<uint8_t *>(l2.d_qres.ptr) + q_bytes, 0, kSlack, h->stream));
    if ((rc = upload(h, l2.d_qoff, q_off)) || (rc = upload(h, l2.d_qlen, q_len)) || (rc = upload(h, l2.d_qband, band)) ||
        (rc = upload(h, l2.d_qevlen, q_evlen)) || (rc = upload(h, l2.d_qevidx, evidx)))
    {
        (void)hipStreamSynchronize(h->stream); // (the uploads queued so far read this function's locals)
        return rc;
    }
    LX_HIP(h, hipStreamSynchronize(h->stream)); // (the uploads read the caller's arrays and this function's locals)
    l2.max_evlen = evlens.empty() ? 0 : evlens.back();
    l2.evlens.swap(evlens);
    l2.q_off.swap(q_off);
    l2.q_len.swap(q_len);
    l2.q_evlen.swap(q_evlen);
    l2.q_frames = f;
    l2.max_qlen = max_qlen;
    l2.q_bytes  = q_bytes;
    return LX_OK;
}

int lx_set_subject_seqs(lx_handle * h, uint64_t const * s_seq_off, uint64_t const * s_seq_len, uint64_t n_sseq)
{
    if (!h)
        return LX_EINVAL;
    if ((!s_seq_off || !s_seq_len) && n_sseq)
        return fail(h, LX_EINVAL, "NULL argument");
    if (n_sseq >= (1ull << 32))
        return fail(h, LX_EINVAL, "at most 2^32 (frame) subject sequences");
    int rc = bind(h);
    if (rc)
        return rc;
    auto & l2 = h->l2;
    // (as lx_set_queries: nothing of the old set stays visible while the new one is validated and uploaded)
    l2.s_hash = 0;
    l2.s_off.clear();
    l2.s_len.clear();
    l2.max_slen = l2.s_extent = 0;
    std::vector<uint64_t> s_off(s_seq_off, s_seq_off + n_sseq), s_len(s_seq_len, s_seq_len + n_sseq);
    uint64_t              max_slen = 0, s_extent = 0;
    for (uint64_t i = 0; i < n_sseq; ++i)
    {
        if (s_seq_off[i] + s_seq_len[i] < s_seq_off[i])
            return fail(h, LX_EINVAL, "subject sequence %llu: offset + length overflows", (unsigned long long)i);
        max_slen = std::max(max_slen, s_seq_len[i]);
        s_extent = std::max(s_extent, s_seq_off[i] + s_seq_len[i]);
    }
    if ((rc = upload(h, l2.d_soff, s_off)) || (rc = upload(h, l2.d_slen, s_len)))
    {
        (void)hipStreamSynchronize(h->stream);
        return rc;
    }
    LX_HIP(h, hipStreamSynchronize(h->stream));
    l2.s_off.swap(s_off);
    l2.s_len.swap(s_len);
    l2.max_slen = max_slen;
    l2.s_extent = s_extent;
    return LX_OK;
}

} // extern "C"

// Sort, merge, unique and the slices for matches that stand on the device as sort words already (`pair` / `s0` of lx_level2.h in
// l2.d_pair[0] / l2.d_s0[0]; l2.d_cnt[2] = the key kernel's error flag): the windows end up in l2.d_win, their slices and
// cut-offs in h->d_ext_all / h->d_min_all.  nw = windows, nEven = those of even subject frames (bisulfite: they come first).
static int level2_windows(lx_handle * h, uint64_t n_matches, bool bisulfite, std::vector<int32_t> const & cut_table, uint64_t & nw, uint64_t & nEven)
{
    auto &            l2 = h->l2;
    hipStream_t const st = h->stream;
    int               rc;
    if ((rc = ensure(h, l2.d_cut, cut_table.size() * sizeof(int32_t) + 16)))
        return rc;
    LX_HIP(h, hipMemcpyAsync(l2.d_cut.ptr, cut_table.data(), cut_table.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    uint64_t const tiles = lx::l2_sort_tiles(n_matches), stiles = lx::l2_scan_tiles(n_matches);
    if ((rc = ensure(h, l2.d_pair[1], n_matches * 8 + 16)) || (rc = ensure(h, l2.d_s0[1], n_matches * 8 + 16)) ||
        (rc = ensure(h, l2.d_hist, (tiles + 2) * 256 * sizeof(uint32_t))) || (rc = ensure(h, l2.d_head, n_matches * 16 + 16)) ||
        (rc = ensure(h, l2.d_tot, (stiles + 2) * sizeof(uint32_t))) ||
        (rc = ensure(h, l2.d_win, n_matches * sizeof(lx::L2Window) + 16)) || (rc = ensure(h, h->d_ext_all, n_matches * sizeof(lx_extension) + 16)) ||
        (rc = ensure(h, h->d_min_all, n_matches * sizeof(int32_t) + 16)) || (rc = ensure_pinned(h, l2.p_cnt, 16 * sizeof(uint64_t))))
        return rc;
    uint64_t * pair = static_cast<uint64_t *>(l2.d_pair[0].ptr), * pair_tmp = static_cast<uint64_t *>(l2.d_pair[1].ptr);
    uint64_t * s0 = static_cast<uint64_t *>(l2.d_s0[0].ptr), * s0_tmp = static_cast<uint64_t *>(l2.d_s0[1].ptr);
    uint64_t const pair_bits = (bisulfite ? 1ull << 63 : 0ull) | (bits_below(l2.q_len.size()) << 32) | bits_below(l2.s_len.size());
    uint64_t const s0_bits   = bits_below(l2.max_slen);
    LX_HIP(h, lx::l2_launch_sort(&pair, &pair_tmp, &s0, &s0_tmp, n_matches, pair_bits, s0_bits, static_cast<uint32_t *>(l2.d_hist.ptr), st));
    lx::L2Params p{};
    p.sets       = lx::L2Sets{static_cast<uint64_t const *>(l2.d_qoff.ptr),   static_cast<uint32_t const *>(l2.d_qlen.ptr),
                        static_cast<uint32_t const *>(l2.d_qband.ptr),  static_cast<uint32_t const *>(l2.d_qevlen.ptr),
                        static_cast<uint64_t const *>(l2.d_soff.ptr),   static_cast<uint64_t const *>(l2.d_slen.ptr),
                        (uint64_t)l2.q_len.size(),                      (uint64_t)l2.s_len.size()};
    p.n          = n_matches;
    p.ext_out    = static_cast<lx::Extension *>(h->d_ext_all.ptr);
    p.min_out    = static_cast<int32_t *>(h->d_min_all.ptr);
    p.win_out    = static_cast<lx::L2Window *>(l2.d_win.ptr);
    p.cut_by_len = static_cast<int32_t const *>(l2.d_cut.ptr);
    p.count_out  = static_cast<uint64_t *>(l2.d_cnt.ptr);
    p.bisulfite  = bisulfite ? 1 : 0;
    // (the span after merge right goes where the sort's spare buffers are, the span after swallow left into d_head)
    LX_HIP(h, lx::l2_launch_merge(pair, s0, p, pair_tmp, s0_tmp, static_cast<uint64_t *>(l2.d_head.ptr), static_cast<uint64_t *>(l2.d_head.ptr) + n_matches,
                                  static_cast<uint32_t *>(l2.d_tot.ptr), st));
    // what the plan of the sweep will want to know about the list (the strip geometry that sweeps it cheapest, its cells), behind
    // the same synchronisation
    LX_HIP(h, lx::l2_launch_plan_cost(p.ext_out, p.count_out, n_matches, 0,
                                      reinterpret_cast<unsigned long long *>(p.count_out + 3), st));
    LX_HIP(h, hipMemcpyAsync(l2.p_cnt.ptr, l2.d_cnt.ptr, 13 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    LX_HIP(h, hipStreamSynchronize(st)); // (also: `cut_table` is the caller's)
    uint64_t const * const cnt = static_cast<uint64_t const *>(l2.p_cnt.ptr);
    uint64_t const         flag = cnt[2];
    nw    = cnt[0];
    nEven = cnt[1];
    if (flag & 1)
        return fail(h, LX_EINVAL, "a match names a query or subject sequence outside the resident sets");
    if (flag & 2)
        return fail(h, LX_EINVAL, "a match lies beyond the end of its subject sequence");
    if (flag & 4)
        return fail(h, LX_EINVAL, "a merged window is longer than 2^32 residues");
    if (nw > n_matches || nEven > nw)
        return fail(h, LX_ESTATE, "the merge kernels report %llu windows of %llu matches", (unsigned long long)nw, (unsigned long long)n_matches);
    return LX_OK;
}

// matches in device memory -> sort words
static int level2_keys(lx_handle * h, void const * d_matches, uint64_t n_matches, bool bisulfite)
{
    auto & l2 = h->l2;
    int    rc;
    if ((rc = ensure(h, l2.d_pair[0], n_matches * 8 + 16)) || (rc = ensure(h, l2.d_s0[0], n_matches * 8 + 16)) || (rc = ensure(h, l2.d_cnt, 16 * sizeof(uint64_t))))
        return rc;
    lx::L2Params p{};
    p.sets.n_qseq = l2.q_len.size();
    p.sets.n_sseq = l2.s_len.size();
    p.sets.s_len  = static_cast<uint64_t const *>(l2.d_slen.ptr);
    p.n           = n_matches;
    p.bisulfite   = bisulfite ? 1 : 0;
    uint64_t * const cnt = static_cast<uint64_t *>(l2.d_cnt.ptr);
    LX_HIP(h, hipMemsetAsync(cnt, 0, 4 * sizeof(uint64_t), h->stream));
    LX_HIP(h, lx::l2_launch_keys(d_matches, p, static_cast<uint64_t *>(l2.d_pair[0].ptr), static_cast<uint64_t *>(l2.d_s0[0].ptr), cnt + 2, h->stream));
    return LX_OK;
}

static_assert(sizeof(lx::BlastMatchDev) == sizeof(lx_blast_match) && offsetof(lx::BlastMatchDev, bit_score) == offsetof(lx_blast_match, bit_score) &&
                  offsetof(lx::BlastMatchDev, ops_off) == offsetof(lx_blast_match, ops_off) && offsetof(lx::BlastMatchDev, q_frame) == offsetof(lx_blast_match, q_frame) &&
                  offsetof(lx::BlastMatchDev, identity) == offsetof(lx_blast_match, identity) && offsetof(lx::BlastMatchDev, s_end) == offsetof(lx_blast_match, s_end),
              "the device writes lx_blast_match rows");

// The tail of iterateMatchesFullSimd (src/search_algo.hpp:1287-1325) for a part of the window list whose survivors the extension
// pipeline left on the device: statistics, order, identity cut-off, records as kernels (lx_records.hip), then ONE copy of finished
// lx_blast_match rows into the result and the alignment columns expanded from the run-length codes by the host threads.  What
// host/lx_iterate_common.hpp's finishSurvivors does for lists in host memory, with the same results to the bit.
// A part is served as ONE range behind its last chunk (the survivors of all chunks in lx_handle::Level2::d_surv_*), or -- where the
// plan's chunks are ranges of the query-sorted list (ResidentInput::ChunkRecords) -- range by range: a range's kernels are queued
// behind its chunk's own, its rows and columns come down while the next chunk computes.
constexpr size_t kPlanProbe = 128;                                 // l2.p_plan: [rank flag: 64 B][the free-packing plan's report: 64 B][probe windows] ...
constexpr size_t kPlanProbes = lx::kFpMaxRanges;                                    // cuts a part's ranges may have: 512 probe windows each
constexpr size_t kPlanHead   = kPlanProbe + kPlanProbes * 512 * sizeof(lx::L2Window); // ... in front of the per-wavefront arrays

struct RecordsJob
{
    struct Range
    {
        uint64_t lo = 0, hi = 0;   // windows of the part
        uint64_t code_base = 0;    // where the chunk's codes begin in h->ext_bytes
        bool     done = false;
    };
    lx_handle *              h;
    lx_search_params const * params;
    lx_iterate_result *      res;
    HostMarks &              hm;
    uint64_t                 part_lo, n_part;
    std::vector<Range>       ranges;
    uint64_t                 next_flush = 0;
    lx::RecParams            base{};
    uint64_t                 pair_bits = 0, s0_bits = 0;
    bool                     want_ops  = true;
    std::vector<double>      pre_host; // the e-value factors per distinct length, until their upload is through
    bool                     use_rank = false; // the survivors are sorted by their windows' ranks (l2.d_rank: rec_launch_rank over the part)

    RecordsJob(lx_handle * h_, lx_search_params const * p_, lx_iterate_result * r_, HostMarks & hm_, uint64_t part_lo_, uint64_t n_part_)
        : h(h_), params(p_), res(r_), hm(hm_), part_lo(part_lo_), n_part(n_part_)
    {
    }

    // the host's share of the arithmetic, made once per call with the host's libm -- per distinct e-value length the factor
    // K * (ql - adj) * (dl - adj) of computeEValue (blast_stats.hpp), exp(-lambda s) for every score until it is zero, the bit-score
    // test as an integer cut-off -- and the buffers (entries: upper bound of the survivor entries of any range; every range has its
    // own rows in d_rec: at most as many as it has windows)
    int prepare(lambda_amd::CutOffs & cutOffFor, std::vector<Range> rs, uint64_t max_entries, uint64_t max_wlen = 0 /* longest window of the part, 0 = unknown */)
    {
        using namespace lambda_amd;
        auto &            l2 = h->l2;
        hipStream_t const st = h->stream;
        int               rc;
        ranges   = std::move(rs);
        want_ops = !(params->flags & LX_ITERATE_NO_OPS);
        int const qFrames = std::max(1, params->qry_num_frames), sFrames = std::max(1, params->sbj_num_frames);
        std::vector<double> & pre = pre_host; // (a member: the upload below is asynchronous, the job outlives it)
        pre.assign(std::max<size_t>(l2.evlens.size(), 1), 0.0);
        for (size_t i = 0; i < l2.evlens.size(); ++i)
        {
            uint64_t const ql  = (uint64_t)l2.evlens[i] / (params->query_translated ? 3 : 1);
            auto           it  = cutOffFor.evalue.cachedLengthAdjustments.find(ql);
            uint64_t const adj = it != cutOffFor.evalue.cachedLengthAdjustments.end() ? it->second : lengthAdjustment(params->db_total_length, ql, params->karlin);
            pre[i]             = params->karlin.K * (double)(ql - adj) * (double)(params->db_total_length - adj);
        }
        if (l2.exp_lambda != params->karlin.lambda || l2.exp_n == 0)
        {
            std::vector<double> tab;
            for (uint32_t sc = 0; sc < (1u << 20); ++sc)
            {
                double const v = std::exp(-params->karlin.lambda * (double)sc);
                tab.push_back(v);
                if (v == 0.0)
                    break;
            }
            if ((rc = upload(h, l2.d_exp, tab)))
                return rc;
            LX_HIP(h, hipStreamSynchronize(st)); // (`tab` is a local)
            l2.exp_lambda = params->karlin.lambda;
            l2.exp_n      = (uint32_t)tab.size();
        }
        int32_t bit_cut = INT32_MIN;
        if (params->min_bitscore >= 0)
        {
            auto const fails = [&](int32_t sc) { return computeBitScore(sc, params->karlin) < params->min_bitscore; };
            int64_t    lo = -(1ll << 30), hi = 1ll << 30; // fails(lo), !fails(hi)
            if (!fails((int32_t)lo))
                bit_cut = INT32_MIN;
            else if (fails((int32_t)hi))
                bit_cut = INT32_MAX;
            else
            {
                while (hi - lo > 1)
                {
                    int64_t const mid = lo + (hi - lo) / 2;
                    (fails((int32_t)mid) ? lo : hi) = mid;
                }
                bit_cut = (int32_t)hi;
            }
        }
        uint64_t max_win = 1;
        for (Range const & r : ranges)
            max_win = std::max(max_win, r.hi - r.lo);
        uint64_t const tiles = (max_entries + 255) / 256, sort_n = std::max<uint64_t>(max_entries, 1), nr = ranges.size();
        if ((rc = upload(h, l2.d_pre, pre)) || (rc = ensure(h, l2.d_listat, max_win * sizeof(uint32_t) + 16)) || (rc = ensure(h, l2.d_reccnt, nr * lx::kRecCounters * sizeof(uint64_t))) ||
            (rc = ensure(h, l2.d_rec, (n_part + 16) * sizeof(lx_blast_match) + 16)) || (rc = ensure(h, l2.d_reccodes, 3 * (n_part + 16) * sizeof(uint64_t) + 16)) ||
            (rc = ensure(h, l2.d_tilekeep, (tiles + 1) * sizeof(uint32_t))) || (rc = ensure(h, l2.d_tileops, (tiles + 1) * sizeof(uint64_t))) ||
            (rc = ensure(h, l2.d_pair[0], sort_n * 8 + 16)) || (rc = ensure(h, l2.d_pair[1], sort_n * 8 + 16)) || (rc = ensure(h, l2.d_s0[0], sort_n * 8 + 16)) ||
            (rc = ensure(h, l2.d_s0[1], sort_n * 8 + 16)) || (rc = ensure(h, l2.d_hist, (lx::l2_sort_tiles(sort_n) + 2) * 256 * sizeof(uint32_t))) ||
            (rc = ensure_pinned(h, l2.p_reccnt, nr * lx::kRecCounters * sizeof(uint64_t))))
        {
            (void)hipStreamSynchronize(st);
            return rc;
        }
        LX_HIP(h, hipMemsetAsync(l2.d_reccnt.ptr, 0, nr * lx::kRecCounters * sizeof(uint64_t), st));
        base.q_len      = static_cast<uint32_t const *>(l2.d_qlen.ptr);
        base.q_evidx    = static_cast<uint32_t const *>(l2.d_qevidx.ptr);
        base.q_frames   = (uint32_t)qFrames;
        base.s_frames   = (uint32_t)sFrames;
        base.n_qid_end  = (uint32_t)((l2.q_len.size() + (uint64_t)qFrames - 1) / (uint64_t)qFrames);
        base.q_mode     = params->q_frame_mode;
        base.s_mode     = params->s_frame_mode;
        base.bit_cut    = bit_cut;
        base.id_cutoff  = params->id_cutoff;
        base.want_ops   = want_ops ? 1 : 0;
        base.lambda     = params->karlin.lambda;
        base.log_k      = std::log(params->karlin.K);
        base.log_2      = std::log(2.0);
        base.pre_by_len = static_cast<double const *>(l2.d_pre.ptr);
        base.exp_tab    = static_cast<double const *>(l2.d_exp.ptr);
        base.exp_n      = l2.exp_n;
        base.ops_base   = 0; // (the rows carry offsets inside their range; flush() knows where the range's columns begin)
        base.list_at    = static_cast<uint32_t *>(l2.d_listat.ptr);
        // the digits the keys can have set: (true query id | padding's id, query slice length), (subject slice length, window | entry)
        pair_bits = (bits_below((uint64_t)base.n_qid_end + 1) << 32) | bits_below((uint64_t)l2.max_qlen + 1);
        // (a window is as long as its subject at most; the plan knows the part's longest window: one digit instead of three on a genome)
        s0_bits   = (bits_below(std::min<uint64_t>(max_wlen ? max_wlen : l2.max_slen, 0xfffffffeull) + 1) << 32) | bits_below(std::max(max_win, max_entries) + 1);
        return LX_OK;
    }

    // the kernels of range r over a survivor list of up to `cap` entries (count_ptr: how many are filled, NULL = all; codes_off: where
    // each entry's codes begin, NULL = the alignment's own offset inside its chunk), on the handle's stream; the counters follow
    int enqueue(uint64_t r, void const * d_hsp, void const * d_src, void const * d_count, void const * d_codes_off, uint64_t cap)
    {
        auto &            l2 = h->l2;
        hipStream_t const st = h->stream;
        Range const &     rg = ranges[r];
        lx::RecParams     p  = base;
        p.hsp       = static_cast<lx::Hsp const *>(d_hsp);
        p.src       = static_cast<uint32_t const *>(d_src);
        p.codes_off = static_cast<uint64_t const *>(d_codes_off);
        p.count_ptr = static_cast<uint64_t const *>(d_count);
        p.n_entries = cap;
        p.src_base  = (uint32_t)rg.lo;
        p.win       = static_cast<lx::L2Window const *>(l2.d_win.ptr) + part_lo + rg.lo;
        p.score     = static_cast<int32_t const *>(h->d_score_all.ptr) + rg.lo;
        p.min_score = static_cast<int32_t const *>(h->d_min_all.ptr) + part_lo + rg.lo;
        p.n_win     = rg.hi - rg.lo;
        p.counters  = static_cast<uint64_t *>(l2.d_reccnt.ptr) + r * lx::kRecCounters;
        p.rec       = static_cast<lx::BlastMatchDev *>(l2.d_rec.ptr) + rg.lo;
        p.rec_codes = static_cast<uint64_t *>(l2.d_reccodes.ptr) + 3 * rg.lo;
        p.rank      = use_rank ? static_cast<uint32_t const *>(l2.d_rank.ptr) + rg.lo : nullptr;
        p.rank_pad  = (uint32_t)n_part;
        LX_HIP(h, hipMemsetAsync(p.counters, 0, lx::kRecCounters * sizeof(uint64_t), st)); // (a range whose chunk runs again starts over)
        LX_HIP(h, hipMemsetAsync(l2.d_listat.ptr, 0xff, p.n_win * sizeof(uint32_t), st));
        uint64_t * pair = static_cast<uint64_t *>(l2.d_pair[0].ptr), * pair_tmp = static_cast<uint64_t *>(l2.d_pair[1].ptr);
        uint64_t * s0 = static_cast<uint64_t *>(l2.d_s0[0].ptr), * s0_tmp = static_cast<uint64_t *>(l2.d_s0[1].ptr);
        LX_HIP(h, lx::rec_launch(p, &pair, &pair_tmp, &s0, &s0_tmp, use_rank ? 0ull : pair_bits, use_rank ? bits_below(n_part + 1) : s0_bits, static_cast<uint32_t *>(l2.d_hist.ptr),
                                 static_cast<uint32_t *>(l2.d_tilekeep.ptr),
                                 static_cast<uint64_t *>(l2.d_tileops.ptr), st));
        LX_HIP(h, hipMemcpyAsync(static_cast<uint64_t *>(l2.p_reccnt.ptr) + r * lx::kRecCounters, p.counters, lx::kRecCounters * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        return LX_OK;
    }

    // range r's kernels are through (the caller synchronised with them): every range up to the first unfinished one goes to the result
    int collect(uint64_t r, uint64_t code_base, bool gpu_busy)
    {
        ranges[r].done      = true;
        ranges[r].code_base = code_base;
        int rc = LX_OK;
        while (rc == LX_OK && next_flush < ranges.size() && ranges[next_flush].done)
            rc = flush(next_flush++, gpu_busy);
        return rc;
    }

    int flush(uint64_t r, bool gpu_busy)
    {
        using namespace lambda_amd;
        auto &                 l2  = h->l2;
        Range const &          rg  = ranges[r];
        uint64_t const * const cnt = static_cast<uint64_t const *>(l2.p_reccnt.ptr) + r * lx::kRecCounters;
        if (cnt[lx::kRecErr] & 1)
            return fail(h, LX_EOVERFLOW, "an extension of the list could not be traced");
        if (cnt[lx::kRecErr] & 2)
            return fail(h, LX_ESTATE, "a survivor names a window outside its range of the list");
        uint64_t const ns = cnt[lx::kRecSurvivors], nkeep = cnt[lx::kRecKept], nops = cnt[lx::kRecOps];
        if (ns > rg.hi - rg.lo || nkeep > ns)
            return fail(h, LX_ESTATE, "the records kernels report %llu records of %llu survivors of %llu windows", (unsigned long long)nkeep, (unsigned long long)ns,
                        (unsigned long long)(rg.hi - rg.lo));
        res->stats.failed_bitscore += cnt[lx::kRecFailedBit];
        res->stats.failed_evalue += cnt[lx::kRecFailedEv];
        if (ns == 0)
            return LX_OK;
        res->stats.num_ext_ali += ns; // :1287
        res->stats.failed_identity += ns - nkeep;
        uint64_t const rec0 = res->matches.size(), ops0 = res->ops.size();
        if (!res->matches.resize(rec0 + nkeep) || !res->ops.resize(ops0 + nops))
            return fail(h, LX_ENOMEM, "out of host memory for the result records");
        if (nkeep == 0)
            return LX_OK;
        // On the copy stream, beside whatever the handle's stream computes (the next chunk's sweep): the rows' column offsets become
        // the result's, then -- 24 bytes per record -- where the records' codes begin and their columns go: the host threads expand the
        // columns from the run-length codes WHILE the rows themselves come down (1.8 ms of copy beside 2.0 ms of expansion on a million reads)
        hipStream_t const      cs = h->stream3;
        lx::BlastMatchDev *    d_rows = static_cast<lx::BlastMatchDev *>(l2.d_rec.ptr) + rg.lo;
        std::vector<uint64_t> & codes_at = l2.rec_codes;
        std::thread             expander;
        if (want_ops)
        {
            if (ops0)
                LX_HIP(h, lx::rec_launch_add_ops_base(d_rows, nkeep, ops0, cs));
            codes_at.resize(3 * nkeep);
            LX_HIP(h, hipMemcpyAsync(codes_at.data(), static_cast<uint64_t const *>(l2.d_reccodes.ptr) + 3 * rg.lo, 3 * nkeep * sizeof(uint64_t), hipMemcpyDeviceToHost, cs));
            LX_HIP(h, hipStreamSynchronize(cs));
            uint8_t const * const codes = h->ext_bytes.data() + rg.code_base;
            uint8_t * const       ops   = res->ops.data() + ops0;
            expander = std::thread(
                [&codes_at, codes, ops, nkeep]()
                {
                    parallelRanges(nkeep,
                                   [&](unsigned, uint64_t lo, uint64_t hi)
                                   {
                                       for (uint64_t k = lo; k < hi; ++k)
                                       {
                                           if (k + 8 < hi)
                                               __builtin_prefetch(codes + codes_at[3 * (k + 8)]);
                                           (void)lx_expand_ops(codes + codes_at[3 * k], (int32_t)codes_at[3 * k + 2], ops + codes_at[3 * k + 1]);
                                       }
                                   });
                });
        }
        // The rows.  While another chunk computes: into pinned memory (a copy engine's transfer, at the link's rate whatever the handle's
        // stream does -- a copy into the result's ordinary memory is staged by the runtime with the compute units the sweep is using:
        // 4.1 instead of 1.1 ms for half a million rows) and from there into the result by the host threads (2.0 ms with the columns,
        // hidden behind the chunk).  With the GPU idle -- the last range -- straight into the result (1.1 ms).
        hipError_t e_rows = hipSuccess;
        int        rc_pin = gpu_busy ? ensure_pinned(h, l2.p_rows, nkeep * sizeof(lx_blast_match) + 16, hipHostMallocNonCoherent) : LX_OK;
        if (rc_pin == LX_OK)
        {
            e_rows = hipMemcpyAsync(gpu_busy ? l2.p_rows.ptr : static_cast<void *>(res->matches.data() + rec0), d_rows, nkeep * sizeof(lx_blast_match), hipMemcpyDeviceToHost, cs);
            if (e_rows == hipSuccess)
                e_rows = hipStreamSynchronize(cs);
        }
        if (expander.joinable())
            expander.join();
        if (rc_pin)
            return rc_pin;
        LX_HIP(h, e_rows);
        if (gpu_busy)
        {
            uint8_t const * const from = static_cast<uint8_t const *>(l2.p_rows.ptr);
            uint8_t * const       to   = reinterpret_cast<uint8_t *>(res->matches.data() + rec0);
            uint64_t const        bytes = nkeep * sizeof(lx_blast_match);
            // (pieces of 256 bytes: parallelRanges spreads lists of 32 768 items and more)
            parallelRanges((bytes + 255) / 256, [&](unsigned, uint64_t lo, uint64_t hi) { std::memcpy(to + lo * 256, from + lo * 256, std::min(bytes, hi * 256) - lo * 256); });
        }
        return LX_OK;
    }
};

// The list work and the extension for matches that stand on the device as sort words already.  Appends to *res.
static int level2_sorted_tail(lx_handle * h, int slot, uint64_t n_matches, lx_search_params const * params, lx_iterate_result * res, bool windows_to_host)
{
    using namespace lambda_amd;
    auto &            l2 = h->l2;
    hipStream_t const st = h->stream;
    HostMarks         hm("lx_iterate_matches_dev");
    HostPool::Call const in_flight_call;
    int               rc;
    res->stats.num_ext_score += n_matches; // lH.stats.numExtScore (:1187)
    // the cut-off of every query length of the resident set (one bisection each, :1251-1283 as an integer test)
    CutOffs              cutOffFor(params);
    std::vector<int32_t> table((size_t)l2.max_evlen + 1, 0x7fffffff);
    for (uint32_t len : l2.evlens)
        table[len] = cutOffFor(len);
    uint64_t nw = 0, nEven = 0;
    if ((rc = level2_windows(h, n_matches, params->bisulfite != 0, table, nw, nEven)))
        return rc;
    l2.dup_before = res->stats.hits_duplicate;
    res->stats.hits_duplicate += n_matches - nw;
    hm.mark("sort+merge");
    if (nw == 0)
        return LX_OK;
    // The window list comes down (24 B per window) on the copy stream, beside what follows, where the host needs it: for the plan of
    // the sweep where that is made on the host, for the records where they are (LX_OPT_ITERATE_RECORDS = 1), for a caller's span
    // (lx_iterate_matches).  Records made on the device over a device plan need none of it.
    bool const records_on_device = h->opt_iterate_records == 0;
    if (!l2.ev_win)
        LX_HIP(h, hipEventCreateWithFlags(&l2.ev_win, hipEventDisableTiming));
    // (queued behind what `st` holds at that moment: where the plan is made on the device, behind the plan's kernels -- beside them
    // the copy's writes over PCIe held the first of them up for the whole 0.5 ms it takes -- and so beside the sweep)
    bool win_queued = false;
    auto queue_windows = [&]() -> int
    {
        if (win_queued)
            return LX_OK;
        win_queued = true;
        int rcw;
        if ((rcw = ensure_pinned(h, l2.p_win, nw * sizeof(lx::L2Window) + 16)))
            return rcw;
        LX_HIP(h, hipEventRecord(l2.ev_win, st));
        LX_HIP(h, hipStreamWaitEvent(h->stream2, l2.ev_win, 0));
        LX_HIP(h, hipMemcpyAsync(l2.p_win.ptr, l2.d_win.ptr, nw * sizeof(lx::L2Window), hipMemcpyDeviceToHost, h->stream2));
        LX_HIP(h, hipEventRecord(l2.ev_win, h->stream2));
        return LX_OK;
    };
    bool                 win_here = false;
    lx::L2Window const * win      = nullptr; // (l2.p_win, once queue_windows has made room)
    l2.score.resize(nw); // (written only where the pipeline hands the survivors to the host: lists it serves without the multi-query plan)
    uint64_t const * const cost = static_cast<uint64_t const *>(l2.p_cnt.ptr) + 3; // [part][19, 13, 11 columns, cells]
    auto const window = [&](uint64_t i)
    {
        lx::L2Window const & w = win[i];
        return WindowView{w.q, w.s, 0, w.beg, l2.q_len[w.q], w.end > w.beg ? (uint32_t)(w.end - w.beg) : 0u, l2.q_evlen[w.q]};
    };
    auto const minOf = [&](uint64_t i) { return table[l2.q_evlen[win[i].q]]; };
    // one strand direction per call of the pipeline: bisulfite lists are sorted by subjId % 2 first (:1369-1372), the even
    // subject frames take the forward scheme (slot 0), the odd ones the reverse scheme (slot 1)
    struct Part
    {
        uint64_t lo, hi;
        int      slot;
    };
    Part const parts[2] = {{0, params->bisulfite ? nEven : nw, params->bisulfite ? 0 : slot}, {params->bisulfite ? nEven : nw, nw, 1}};
    for (int pi = 0; pi < 2; ++pi)
    {
        Part const & pt = parts[pi];
        if (pt.lo == pt.hi)
            continue;
        uint64_t const n = pt.hi - pt.lo;
        ResidentInput  ri;
        ri.d_q       = l2.d_qres.ptr;
        ri.q_bytes   = l2.q_bytes;
        ri.d_ext_all = static_cast<lx_extension const *>(h->d_ext_all.ptr) + pt.lo;
        ri.d_min_all = static_cast<int32_t const *>(h->d_min_all.ptr) + pt.lo;
        lx_survivor_list list{};
        // The plan of the sweep on the device: the solo packing where 16 profiles fit a wavefront's share of the LDS (nucleotides,
        // bisulfite), else the free packing (protein lists: four queries per wavefront) -- nothing of size n comes to the host either way
        bool const solo = solo_plan_applies(h, pt.slot), free_packing = !solo && free_plan_applies(h, pt.slot);
        if (solo || free_packing)
        {
            // The plan of the sweep on the device (the solo packing of lx_sweep_mq.hip: every window its own profile): ONE strip
            // geometry per call, the one that sweeps the list cheapest (lx_host.cpp has the measurement behind "one" and behind
            // the 15 % that narrower strips must save), then all windows by (columns per lane, length), 16 to a wavefront.
            int const cand[3] = {1, 3, 5};
            int       cfg     = 1;
            double    best    = 1e300;
            for (int k = 0; k < 3; ++k)
            {
                double const c = (double)cost[4 * pi + k] * (cand[k] == 1 ? 1.0 : 1.15);
                if (c < best)
                {
                    best = c;
                    cfg  = cand[k];
                }
            }
            // Ranges of the query-sorted list, each planned by itself and served as one chunk of the pipeline: a range's records are made
            // behind its chunk and come down while the next range computes (RecordsJob).  The cuts stand where the true query id changes
            // (the result is ordered by it): the windows around n / R, 2 n / R, ... come down for that, 12 KB each.
            std::vector<RecordsJob::Range> ranges;
            {
                uint64_t const forced = lx::dev_aids().l2_ranges;
                // (measured on 1.25 M windows, bench.py --iterate: 1 range 11.8, 2 ranges 10.8-11.4, 3 ranges 11.3-11.9, 4 ranges 11.9 ms: a
                // range costs a sweep's tail, a backtrace's tail and the records kernels' thirty launches; the protein list of bench.py --iterate
                // --config 1, 3.2 M windows: 1 range 29.1, 2 ranges 24.6, 3 ranges 24.9, 4 ranges 25.5 ms)
                uint64_t R = !records_on_device ? 1 : forced ? forced : n < 300000 ? 1 : std::min<uint64_t>(4, n / 4000000 + 2);
                // ... but a range's checkpoint slots stand in HBM together, and FRESH device memory is dear: 40 ms per GB the first time a
                // process touches it (tools/dev/malloc_probe: hipMalloc of 12 / 32 GB 0.6 / 1.3 s) -- 1.3 s for the two ranges of the protein
                // list above against a call of 25 ms.  So the ranges also fit what the handle holds already, or 8 GiB: a search's one call
                // pays 0.3 s instead, a handle whose slots are larger (LX_OPT_TRACE_BYTES-sized by earlier calls) keeps the fewer ranges.
                if (records_on_device && !forced)
                {
                    uint64_t const panel  = (uint64_t)lx::trace_cfg_panel(cfg);
                    uint64_t const panels = std::max<uint64_t>(1, ((uint64_t)l2.max_qlen + panel - 1) / panel);
                    uint64_t const steps  = ((uint64_t)cost[8 + pi] + 8 - 1 + 15) & ~15ull; // (the part's longest window)
                    uint64_t const slot_b = panels * (lx::ckpt16_slot_dwords(cfg, (uint32_t)std::min<uint64_t>(steps, 65520)) + lx::ckpt_slot_dwords(cfg, (uint32_t)std::min<uint64_t>(steps, 65520)) / 8) * 4;
                    // (every slot sized for the LONGEST window: an upper bound -- slots are laid out wavefront by wavefront; ordinary windows
                    // are a half to a third of a merged one: half of it per slot is what a list takes at most)
                    uint64_t const need   = (n + n / 8) * (slot_b / 2 + 1);
                    uint64_t const have   = std::min<uint64_t>(h->opt_trace_bytes, std::max<uint64_t>(h->d_trace.cap, 8ull << 30));
                    R = std::min<uint64_t>(lx::kFpMaxRanges, std::max<uint64_t>(R, (need + have - 1) / have));
                }
                R = std::min<uint64_t>(R, lx::kFpMaxRanges); // (what the plan kernels and the probe block take; LX_L2_RANGES may ask for more)
                uint64_t       lo     = 0;
                int const      qF     = std::max(1, params->qry_num_frames);
                // the windows around every cut's target come down together (512 each), one synchronisation
                // (two ranges: the first one two thirds of the windows -- what follows the LAST range's kernels, its rows on the PCIe link and
                // its columns, is nobody's shadow, while the first range's comes down beside the second's kernels as long as those
                // take longer; bench.py --iterate, fastest of eight calls at 50 / 62 / 66 / 70 / 75 / 80 %: 10.7 / 10.55 / 10.5 / 10.5 /
                // 10.9 / 11.4 ms)
                struct Probe
                {
                    uint64_t target, from, upto;
                };
                std::vector<Probe> probes;
                for (uint64_t k = 1; k < R && probes.size() < kPlanProbes; ++k)
                {
                    uint64_t const target = R == 2 ? n / 100 * 66 : n * k / R, from = target > 256 ? target - 256 : 0, upto = std::min(n, target + 256);
                    if (upto - from >= 2 && (probes.empty() || from >= probes.back().upto))
                        probes.push_back(Probe{target, from, upto});
                }
                if (!probes.empty())
                {
                    // (into pinned memory: a copy into ordinary memory is staged by the runtime, 40 us of an idle GPU each)
                    if ((rc = ensure_pinned(h, l2.p_plan, kPlanHead)))
                        return rc;
                    lx::L2Window * const probe = reinterpret_cast<lx::L2Window *>(static_cast<uint8_t *>(l2.p_plan.ptr) + kPlanProbe);
                    for (size_t k = 0; k < probes.size(); ++k)
                        LX_HIP(h, hipMemcpyAsync(probe + 512 * k, static_cast<lx::L2Window const *>(l2.d_win.ptr) + pt.lo + probes[k].from,
                                                 (probes[k].upto - probes[k].from) * sizeof(lx::L2Window), hipMemcpyDeviceToHost, st));
                    LX_HIP(h, hipStreamSynchronize(st));
                    for (size_t k = 0; k < probes.size(); ++k)
                    {
                        lx::L2Window const * const pw = probe + 512 * k;
                        uint64_t const from = probes[k].from, upto = probes[k].upto;
                        uint64_t cut = 0;
                        for (uint64_t w = std::max<uint64_t>(probes[k].target, from + 1); w < upto && !cut; ++w)
                            if (pw[w - from].q / (uint32_t)qF != pw[w - from - 1].q / (uint32_t)qF)
                                cut = w;
                        if (cut && cut > lo && cut < n)
                        {
                            ranges.push_back(RecordsJob::Range{lo, cut});
                            lo = cut;
                        }
                    }
                }
                ranges.push_back(RecordsJob::Range{lo, n});
            }
            uint64_t nwf = 0, cap_wf = 0; // (cap_wf: what the per-wavefront arrays are spaced by on the device)
            l2.cut_wf.assign(1, 0);
            if (solo)
            {
                for (auto const & rg : ranges)
                {
                    nwf += (rg.hi - rg.lo + 15) / 16;
                    l2.cut_wf.push_back(nwf);
                }
                cap_wf = nwf;
            }
            else
                cap_wf = lx::fp_wavefront_bound(n, l2.q_len.size(), (uint32_t)ranges.size());
            if ((rc = ensure(h, l2.d_plan, cap_wf * 16 * sizeof(uint32_t) + 16)) || (rc = ensure(h, l2.d_wf, 2 * cap_wf * sizeof(uint32_t) + 64)) ||
                (rc = ensure_pinned(h, l2.p_plan, kPlanHead)))
                return rc;
            uint32_t * const d_pan = static_cast<uint32_t *>(l2.d_wf.ptr), * const d_maxs = d_pan + cap_wf;
            uint32_t * const h_report = static_cast<uint32_t *>(l2.p_plan.ptr) + 16;
            // (records on the device: every window's place in the records' order, once -- lx_records.hip: the ranges' survivors are sorted by
            // it.  One large kernel that needs the window list only: on the second stream, beside the plan's three dozen small launches)
            bool const       try_rank   = records_on_device && n < 0xfffffff0ull;
            uint32_t * const h_rankflag = reinterpret_cast<uint32_t *>(static_cast<uint64_t *>(l2.p_cnt.ptr) + 15); // (pinned; level2_windows reads words 0-12)
            *h_rankflag = 0;
            if (try_rank)
            {
                if ((rc = ensure(h, l2.d_rank, (n + 1) * sizeof(uint32_t) + 16)))
                    return rc;
                for (hipEvent_t & ev : l2.ev_rank)
                    if (!ev)
                        LX_HIP(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                uint32_t * const d_rank = static_cast<uint32_t *>(l2.d_rank.ptr);
                LX_HIP(h, hipEventRecord(l2.ev_rank[0], st)); // (the window list is complete behind what `st` holds now)
                LX_HIP(h, hipStreamWaitEvent(h->stream2, l2.ev_rank[0], 0));
                LX_HIP(h, hipMemsetAsync(d_rank + n, 0, sizeof(uint32_t), h->stream2));
                LX_HIP(h, lx::rec_launch_rank(static_cast<lx::L2Window const *>(l2.d_win.ptr) + pt.lo, n, (uint32_t)std::max(1, params->qry_num_frames),
                                              static_cast<uint32_t const *>(l2.d_qlen.ptr), d_rank, d_rank + n, h->stream2));
                LX_HIP(h, hipMemcpyAsync(h_rankflag, d_rank + n, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream2));
                LX_HIP(h, hipEventRecord(l2.ev_rank[1], h->stream2));
            }
            if (solo)
                for (size_t r = 0; r < ranges.size(); ++r)
                {
                    uint64_t * key = static_cast<uint64_t *>(l2.d_pair[0].ptr), * key_tmp = static_cast<uint64_t *>(l2.d_pair[1].ptr);
                    uint64_t * idx = static_cast<uint64_t *>(l2.d_s0[0].ptr), * idx_tmp = static_cast<uint64_t *>(l2.d_s0[1].ptr);
                    uint64_t const w0 = l2.cut_wf[r];
                    LX_HIP(h, lx::l2_launch_plan(static_cast<lx::Extension const *>(ri.d_ext_all) + ranges[r].lo, ranges[r].hi - ranges[r].lo, lx::trace_cfg_panel(cfg) / 8,
                                                 0, &key, &key_tmp, &idx, &idx_tmp, static_cast<uint32_t *>(l2.d_hist.ptr),
                                                 static_cast<uint32_t *>(l2.d_plan.ptr) + w0 * 16, d_pan + w0, d_maxs + w0, st, (uint32_t)ranges[r].lo,
                                                 lx::l2_plan_key_bits(l2.max_qlen, cost[8 + pi])));
                }
            else
            {
                // the free packing (lx_plan_free.hip): ONE plan over the part, laid out range by range; how many wavefronts it takes is
                // the plan's to say (its report comes down with the rank kernel's flag)
                lx::FpArgs fa{};
                fa.ext        = static_cast<lx::Extension const *>(ri.d_ext_all);
                fa.n          = n;
                fa.n_qseq     = l2.q_len.size();
                fa.C          = lx::trace_cfg_panel(cfg) / 8;
                fa.no_narrow  = 0;
                fa.nranges    = (uint32_t)ranges.size();
                for (size_t r = 0; r < ranges.size(); ++r)
                    fa.cut[r] = ranges[r].lo;
                fa.cut[ranges.size()] = n;
                fa.work_bytes = lx::fp_workspace_bytes(n, fa.n_qseq, fa.nranges);
                if ((rc = ensure(h, l2.d_fp, fa.work_bytes + 64)))
                    return rc;
                fa.work    = l2.d_fp.ptr;
                fa.plan    = static_cast<uint32_t *>(l2.d_plan.ptr);
                fa.cap_wf  = cap_wf;
                fa.wf_pan  = d_pan;
                fa.wf_maxs = d_maxs;
                fa.report  = d_pan + 2 * cap_wf; // (16 words behind the per-wavefront arrays)
                LX_HIP(h, lx::fp_launch_plan(fa, st));
                LX_HIP(h, hipMemcpyAsync(h_report, fa.report, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            }
            if (try_rank)
                LX_HIP(h, hipStreamWaitEvent(st, l2.ev_rank[1], 0)); // (the records kernels on `st` read the ranks)
            // what the host reads of the plan, in pinned memory: columns per lane and longest window per wavefront
            if (!solo)
            {
                LX_HIP(h, hipStreamSynchronize(st));
                nwf = h_report[0];
                if (h_report[1] || nwf == 0 || nwf > cap_wf)
                    return fail(h, LX_ESTATE, "the free-packing plan of %llu windows reports %llu wavefronts (at most %llu), flag %u", (unsigned long long)n,
                                (unsigned long long)nwf, (unsigned long long)cap_wf, h_report[1]);
                for (size_t r = 0; r < ranges.size(); ++r)
                    l2.cut_wf.push_back(h_report[4 + r + 1]);
                if (l2.cut_wf.back() != nwf)
                    return fail(h, LX_ESTATE, "the free-packing plan's ranges end at wavefront %llu of %llu", (unsigned long long)l2.cut_wf.back(), (unsigned long long)nwf);
            }
            // (no copy into the block is pending here: it may move)
            if ((rc = ensure_pinned(h, l2.p_plan, kPlanHead + 2 * nwf * sizeof(uint32_t))))
                return rc;
            uint32_t * const h_pan = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(l2.p_plan.ptr) + kPlanHead), * const h_maxs = h_pan + nwf;
            LX_HIP(h, hipMemcpyAsync(h_pan, d_pan, nwf * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            LX_HIP(h, hipMemcpyAsync(h_maxs, d_maxs, nwf * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            if ((windows_to_host || !records_on_device) && (rc = queue_windows()))
                return rc;
            LX_HIP(h, hipStreamSynchronize(st)); // (behind the rank kernel's event as well: its flag has arrived)
            ri.d_plan  = static_cast<uint32_t const *>(l2.d_plan.ptr);
            ri.nwf     = nwf;
            ri.wf_pan  = h_pan;
            ri.wf_maxs = h_maxs;
            l2.rank_too_long = *h_rankflag;
            ri.mq_cfg  = cfg;
            ri.free_packing = !solo;
            ri.cells   = cost[4 * pi + 3];
            ri.keep_on_device = records_on_device;
            ri.want_codes     = !(params->flags & LX_ITERATE_NO_OPS);
            RecordsJob                  job(h, params, res, hm, pt.lo, n);
            ResidentInput::ChunkRecords cr;
            if (records_on_device)
            {
                uint64_t max_entries = 0;
                for (size_t r = 0; r < ranges.size(); ++r)
                    max_entries = std::max(max_entries, (l2.cut_wf[r + 1] - l2.cut_wf[r]) * 16 + 16);
                uint64_t max_wlen = 1; // (the plan's per-wavefront maxima: every window of the part stands in one of them)
                for (uint64_t w = 0; w < nwf; ++w)
                    max_wlen = std::max<uint64_t>(max_wlen, h_maxs[w]);
                if ((rc = job.prepare(cutOffFor, ranges, max_entries, max_wlen)))
                    return rc;
                job.use_rank = try_rank && l2.rank_too_long == 0; // (a query with more windows than the rank kernel counts: the full sort words)
                cr.cut_wf   = l2.cut_wf.data();
                cr.n_ranges = ranges.size();
                cr.enqueue  = [&job](uint64_t r, void const * d_hsp, void const * d_src, void const * d_count, uint64_t cap) { return job.enqueue(r, d_hsp, d_src, d_count, nullptr, cap); };
                cr.collect  = [&job](uint64_t r, uint64_t code_base, bool gpu_busy) { return job.collect(r, code_base, gpu_busy); };
                ri.chunk_records = &cr;
            }
            hm.mark("plan");
            if ((rc = extend_list_resident(h, pt.slot, ri, nullptr, n, nullptr, l2.score.data() + pt.lo, &list)))
                return rc;
            if (l2.surv_on_device && l2.surv_by_range)
            {
                if (job.next_flush != job.ranges.size())
                    return fail(h, LX_ESTATE, "the pipeline finished with %llu of %llu ranges' records made", (unsigned long long)job.next_flush,
                                (unsigned long long)job.ranges.size());
                hm.mark("extension + records (range by range)");
                continue;
            }
        }
        else
        {
            // the plan is made on the host (lx_host.cpp): its copies of the slices and cut-offs
            if ((rc = queue_windows()))
                return rc;
            if (!win_here)
            {
                LX_HIP(h, hipEventSynchronize(l2.ev_win));
                win_here = true;
                win      = static_cast<lx::L2Window const *>(l2.p_win.ptr);
                l2.ext.resize(nw);
                l2.min.resize(nw);
                parallelRanges(nw,
                               [&](unsigned, uint64_t lo, uint64_t hi)
                               {
                                   for (uint64_t i = lo; i < hi; ++i)
                                   {
                                       lx::L2Window const & w = win[i];
                                       lx_extension &       e = l2.ext[i];
                                       e.q_off   = l2.q_off[w.q];
                                       e.q_len   = l2.q_len[w.q];
                                       e.s_off   = l2.s_off[w.s] + w.beg;
                                       e.s_len   = w.end > w.beg ? (uint32_t)(w.end - w.beg) : 0u;
                                       l2.min[i] = table[l2.q_evlen[w.q]];
                                   }
                               });
                hm.mark("windows");
            }
            ri.keep_on_device = records_on_device;
            ri.want_codes     = !(params->flags & LX_ITERATE_NO_OPS);
            if ((rc = extend_list_resident(h, pt.slot, ri, l2.ext.data() + pt.lo, n, l2.min.data() + pt.lo, l2.score.data() + pt.lo, &list)))
                return rc;
        }
        hm.mark("extension");
        if (l2.surv_on_device)
        {
            // the survivors of all chunks stand in l2.d_surv_*: one range, its kernels now
            RecordsJob whole(h, params, res, hm, pt.lo, n);
            if ((rc = whole.prepare(cutOffFor, {RecordsJob::Range{0, n}}, l2.surv_total)) ||
                (rc = whole.enqueue(0, l2.d_surv_hsp.ptr, l2.d_surv_src.ptr, nullptr, l2.d_surv_codes.ptr, l2.surv_total)))
                return rc;
            LX_HIP(h, hipStreamSynchronize(st));
            hm.mark("statistics+order+records (kernels)");
            if ((rc = whole.collect(0, 0, false)))
                return rc;
            hm.mark("rows + columns");
            continue;
        }
        // (else the pipeline served the list without the multi-query plan -- a handful of windows, LX_OPT_MQ_SWEEP = 0 -- and its survivors
        // and scores are on the host: the host threads finish them, as for LX_OPT_ITERATE_RECORDS = 1)
        if (!win_here)
        {
            if ((rc = queue_windows()))
                return rc;
            LX_HIP(h, hipEventSynchronize(l2.ev_win));
            win_here = true;
            win      = static_cast<lx::L2Window const *>(l2.p_win.ptr);
        }
        uint64_t const base = pt.lo;
        if ((rc = finishSurvivors(n, [&](uint64_t i) { return window(base + i); }, l2.score.data() + pt.lo, [&](uint64_t i) { return minOf(base + i); }, list, params, res)))
            return fail(h, rc, "out of host memory for the result records");
        hm.mark("statistics+records");
    }
    if (params->bisulfite) // the HSPs are stably re-sorted by query (:1379); the ops offsets stay valid: only the records move
        std::stable_sort(res->matches.begin(), res->matches.end(), [](lx_blast_match const & a, lx_blast_match const & b) { return a.n_qid < b.n_qid; });
    if (windows_to_host)
    {
        if ((rc = queue_windows()))
            return rc;
        LX_HIP(h, hipEventSynchronize(l2.ev_win));
    }
    return LX_OK;
}

namespace
{

// 64-bit content hash of a buffer over the host threads: pieces of 1 MiB, each folded with a multiply-xorshift, combined in order
// (never 0: that is "no hash")
uint64_t content_hash(void const * data, uint64_t bytes, uint64_t seed)
{
    uint8_t const * const p       = static_cast<uint8_t const *>(data);
    uint64_t const        piece   = 1ull << 20, npieces = (bytes + piece - 1) / piece;
    std::vector<uint64_t> hp(npieces, 0);
    unsigned const        nt = npieces >= 8 ? std::max(1u, lxi::pool_width()) : 1u;
    auto fold = [&](unsigned t)
    {
        for (uint64_t k = t; k < npieces; k += nt)
        {
            uint64_t const a = k * piece, b = std::min(bytes, a + piece);
            uint64_t       x = 0x9E3779B97F4A7C15ull ^ k, i = a;
            for (; i + 8 <= b; i += 8)
            {
                uint64_t w;
                std::memcpy(&w, p + i, 8);
                x = (x ^ w) * 0xD6E8FEB86659FD93ull;
                x ^= x >> 32;
            }
            for (; i < b; ++i)
                x = (x ^ p[i]) * 0xD6E8FEB86659FD93ull;
            hp[k] = x;
        }
    };
    if (nt <= 1)
        fold(0);
    else
        lxi::pool_run(nt, fold);
    uint64_t x = seed ^ (bytes * 0x9E3779B97F4A7C15ull);
    for (uint64_t v : hp)
    {
        x = (x ^ v) * 0xD6E8FEB86659FD93ull;
        x ^= x >> 29;
    }
    return x ? x : 1;
}

} // namespace

int lxi::iterate_host_list_on_device(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint64_t const * q_seq_off, uint64_t const * q_seq_len,
                                     uint64_t n_qseq, uint64_t const * q_orig_len, uint8_t const * s_res, uint64_t s_bytes, uint64_t const * s_seq_off,
                                     uint64_t const * s_seq_len, uint64_t n_sseq, lx_match * matches, uint64_t n_matches, lx_search_params const * params,
                                     lx_iterate_result * res)
{
    using namespace lambda_amd;
    // what this path serves: lists large enough to pay for the hand-over, the reference's full rectangle, resident subjects
    if (n_matches < 4 * kParallelFrom || n_matches > 0x7ffffff0ull || params->band > 0 || h->opt_band || s_res != nullptr || s_bytes != 0 || !h->db_bytes || n_qseq >= (1ull << 31) ||
        n_sseq >= (1ull << 32) || lx::dev_aids().iterate_on_host)
        return kNotTaken;
    if (!params->bisulfite && (slot < 0 || slot > 1 || !h->have_sc[slot]))
        return kNotTaken; // (the host form reports it)
    if (params->bisulfite && (!h->have_sc[0] || !h->have_sc[1]))
        return kNotTaken;
    auto &    l2 = h->l2;
    HostMarks hm("lx_iterate_matches (device list work)");
    HostPool::Call const in_flight_call;
    int       rc;
    if ((rc = bind(h)))
        return rc;
    // ---- the sequence sets: resident already (the same content as last time) or set now
    int const      frames = std::max(1, params->qry_num_frames);
    uint64_t const n_orig = q_orig_len ? (n_qseq + (uint64_t)frames - 1) / (uint64_t)frames : 0;
    uint64_t const qh     = (content_hash(q_res, q_bytes, 1) ^ content_hash(q_seq_off, n_qseq * 8, 2) ^ content_hash(q_seq_len, n_qseq * 8, 3) ^
                         content_hash(q_orig_len, n_orig * 8, 4) ^ (uint64_t)frames) | 1;
    if (l2.q_len.size() != n_qseq || l2.q_bytes != q_bytes || l2.q_frames != frames || l2.q_hash != qh || n_qseq == 0)
    {
        if ((rc = lx_set_queries(h, q_res, q_bytes, q_seq_off, q_seq_len, n_qseq, q_orig_len, frames)))
            return rc == LX_EINVAL ? kNotTaken : rc; // (a set the resident form does not take: the host form copes or reports)
        l2.q_hash = qh;
    }
    uint64_t const sh = (content_hash(s_seq_off, n_sseq * 8, 5) ^ content_hash(s_seq_len, n_sseq * 8, 6)) | 1;
    if (l2.s_len.size() != n_sseq || l2.s_hash != sh || n_sseq == 0)
    {
        if ((rc = lx_set_subject_seqs(h, s_seq_off, s_seq_len, n_sseq)))
            return rc == LX_EINVAL ? kNotTaken : rc;
        l2.s_hash = sh;
    }
    if (l2.s_extent > h->db_bytes)
        return kNotTaken;
    hm.mark("sets");
    // ---- the matches as sort words (lx_level2.h), made by the host threads straight into pinned memory, 16 bytes each
    if ((rc = ensure_pinned(h, l2.p_up, n_matches * 16 + 16)) || (rc = ensure(h, l2.d_pair[0], n_matches * 8 + 16)) || (rc = ensure(h, l2.d_s0[0], n_matches * 8 + 16)) ||
        (rc = ensure(h, l2.d_cnt, 16 * sizeof(uint64_t))))
        return rc;
    uint64_t * const      pair = static_cast<uint64_t *>(l2.p_up.ptr), * const s0 = pair + n_matches;
    std::vector<uint8_t>  bad(std::max(1u, lxi::pool_width()), 0);
    bool const            bs = params->bisulfite != 0;
    parallelRanges(n_matches,
                   [&](unsigned t, uint64_t lo, uint64_t hi)
                   {
                       for (uint64_t i = lo; i < hi; ++i)
                       {
                           lx_match const & m = matches[i];
                           uint64_t const   d = m.subjStart < m.qryStart ? 0 : m.subjStart - m.qryStart; // _widenMatch's first line, :923
                           if (d >= s_seq_len[m.subjId])
                               bad[t] = 1;
                           pair[i] = ((bs ? (m.subjId & 1) : 0ull) << 63) | (m.qryId << 32) | m.subjId;
                           s0[i]   = d;
                       }
                   });
    for (uint8_t b : bad)
        if (b)
            return fail(h, LX_EINVAL, "a match lies beyond the end of its subject sequence");
    LX_HIP(h, hipMemsetAsync(l2.d_cnt.ptr, 0, 4 * sizeof(uint64_t), h->stream));
    LX_HIP(h, hipMemcpyAsync(l2.d_pair[0].ptr, pair, n_matches * 8, hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemcpyAsync(l2.d_s0[0].ptr, s0, n_matches * 8, hipMemcpyHostToDevice, h->stream));
    hm.mark("sort words");
    uint64_t const ruleBefore = h->opt_bs_rule;
    if (bs)
        h->opt_bs_rule = 1; // the bisulfite overload of computeAlignmentStats for the duration of the call
    rc             = level2_sorted_tail(h, slot, n_matches, params, res, true);
    h->opt_bs_rule = ruleBefore;
    if (rc)
        return rc;
    // ---- the reference's span now holds the windows (:1173-1174 shrinks it): the first n_windows records of `matches`
    uint64_t const             nw  = n_matches - (res->stats.hits_duplicate - l2.dup_before);
    lx::L2Window const * const win = static_cast<lx::L2Window const *>(l2.p_win.ptr);
    parallelRanges(nw,
                   [&](unsigned, uint64_t lo, uint64_t hi)
                   {
                       for (uint64_t i = lo; i < hi; ++i)
                           matches[i] = lx_match{win[i].q, win[i].s, 0, l2.q_len[win[i].q], win[i].beg, win[i].end};
                   });
    return LX_OK;
}

extern "C" {

int lx_iterate_matches_dev(lx_handle * h, int slot, void const * d_matches, uint64_t n_matches, lx_search_params const * params, lx_iterate_result ** out)
{
    if (!h || !out || !params)
        return LX_EINVAL;
    *out = nullptr;
    if (!d_matches && n_matches)
        return fail(h, LX_EINVAL, "NULL argument");
    if (n_matches > 0x7ffffff0ull)
        return fail(h, LX_EINVAL, "at most 2^31 matches per call");
    auto & l2 = h->l2;
    if (l2.q_len.empty() || l2.s_len.empty() || !h->db_bytes)
        return fail(h, LX_ESTATE, "lx_iterate_matches_dev needs the resident sets: lx_set_queries, lx_set_subjects, lx_set_subject_seqs");
    if (l2.s_extent > h->db_bytes)
        return fail(h, LX_EINVAL, "the subject sequences reach byte %llu, the resident residue buffer holds %llu", (unsigned long long)l2.s_extent,
                    (unsigned long long)h->db_bytes);
    if (std::max(1, params->qry_num_frames) != l2.q_frames)
        return fail(h, LX_EINVAL, "qry_num_frames = %d, but the queries were set with %d frames", params->qry_num_frames, l2.q_frames);
    if (params->band > 0 || h->opt_band)
        return fail(h, LX_EINVAL, "lx_iterate_matches_dev: band mode (lx_search_params.band, LX_OPT_BAND) goes through lx_iterate_matches");
    if (params->flags & ~(int32_t)LX_ITERATE_NO_OPS)
        return fail(h, LX_EINVAL, "lx_search_params.flags = 0x%x: unknown bits (this library knows LX_ITERATE_NO_OPS)", (unsigned)params->flags);
    if (!params->bisulfite && (slot < 0 || slot > 1 || !h->have_sc[slot]))
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (params->bisulfite && (!h->have_sc[0] || !h->have_sc[1]))
        return fail(h, LX_ESTATE, "bisulfite mode needs both scoring slots");
    int rc = bind(h);
    if (rc)
        return rc;
    auto res = new lx_iterate_result();
    if (n_matches == 0)
    {
        *out = res;
        return LX_OK;
    }
    // the bisulfite overload of computeAlignmentStats for the duration of the call (src/evaluate_bisulfite_alignment.hpp:97)
    uint64_t const ruleBefore = h->opt_bs_rule;
    if (params->bisulfite)
        h->opt_bs_rule = 1;
    if ((rc = level2_keys(h, d_matches, n_matches, params->bisulfite != 0)) == LX_OK)
        rc = level2_sorted_tail(h, slot, n_matches, params, res, false);
    h->opt_bs_rule = ruleBefore;
    if (rc != LX_OK)
    {
        delete res;
        return rc;
    }
    *out = res;
    return LX_OK;
}

// What the first lx_iterate_matches_dev call of a handle would otherwise allocate inside the call: 21.9 ms against 11.9 ms for the
// calls behind it on a million reads (tools/dev/cold_iterate.py) -- 8.5 instead of 2.1 ms for the copy of the rows into result memory
// that was never touched, 3 ms of allocations in the pipeline -- and a search makes ONE such call per run.  The sizes below are the
// formulas of the call's own ensure()s (level2_windows, level2_sorted_tail, RecordsJob::prepare, extend_pipeline's lanes); what
// falls short grows in the call as before.
int lx_reserve(lx_handle * h, uint64_t n_matches, uint64_t n_windows, uint64_t n_hsps, uint64_t n_columns)
{
    if (!h)
        return LX_EINVAL;
    if (n_matches > 0x7ffffff0ull || n_windows > n_matches || n_hsps > n_windows)
        return fail(h, LX_EINVAL, "lx_reserve: n_hsps <= n_windows <= n_matches <= 2^31");
    int rc = bind(h);
    if (rc || n_matches == 0)
        return rc;
    auto &         l2    = h->l2;
    uint64_t const tiles = lx::l2_sort_tiles(n_matches), stiles = lx::l2_scan_tiles(n_matches);
    HostMarks      hm("lx_reserve");
    // ---- the list work (level2_keys, level2_windows) and the device plan
    if ((rc = ensure(h, l2.d_pair[0], n_matches * 8 + 16)) || (rc = ensure(h, l2.d_s0[0], n_matches * 8 + 16)) || (rc = ensure(h, l2.d_cnt, 16 * sizeof(uint64_t))) ||
        (rc = ensure(h, l2.d_pair[1], n_matches * 8 + 16)) || (rc = ensure(h, l2.d_s0[1], n_matches * 8 + 16)) ||
        (rc = ensure(h, l2.d_hist, (tiles + 2) * 256 * sizeof(uint32_t))) || (rc = ensure(h, l2.d_head, n_matches * 16 + 16)) ||
        (rc = ensure(h, l2.d_tot, (stiles + 2) * sizeof(uint32_t))) || (rc = ensure(h, l2.d_win, n_matches * sizeof(lx::L2Window) + 16)) ||
        (rc = ensure(h, h->d_ext_all, n_matches * sizeof(lx_extension) + 16)) || (rc = ensure(h, h->d_min_all, n_matches * sizeof(int32_t) + 16)) ||
        (rc = ensure_pinned(h, l2.p_cnt, 16 * sizeof(uint64_t))) || (rc = ensure(h, l2.d_cut, ((size_t)l2.max_evlen + 1) * sizeof(int32_t) + 16)))
        return rc;
    // (the plan: the solo packing takes a wavefront per 16 windows; the free packing of a protein list is bounded by its own formula and
    // has a workspace)
    int const      slot_set    = h->have_sc[0] ? 0 : h->have_sc[1] ? 1 : -1;
    bool const     free_plan   = slot_set >= 0 && !solo_plan_applies(h, slot_set) && free_plan_applies(h, slot_set) && n_windows > 0;
    uint64_t const nwf_solo    = (n_windows + 15) / 16;
    uint64_t const nwf         = free_plan ? lx::fp_wavefront_bound(n_windows, l2.q_len.size(), 4) : nwf_solo, entries = n_windows + n_windows / 64 + 8192;
    if ((rc = ensure(h, l2.d_plan, nwf * 16 * sizeof(uint32_t) + 16)) || (rc = ensure(h, l2.d_wf, 2 * nwf * sizeof(uint32_t) + 64)) ||
        (free_plan && (rc = ensure(h, l2.d_fp, lx::fp_workspace_bytes(n_windows, l2.q_len.size(), 4) + 64))))
        return rc;
    l2.wf_pan.reserve(nwf_solo);
    l2.wf_maxs.reserve(nwf_solo);
    hm.mark("list work");
    // ---- the records (RecordsJob)
    if ((rc = ensure(h, l2.d_surv_hsp, entries * sizeof(lx_hsp))) || (rc = ensure(h, l2.d_surv_src, entries * sizeof(uint32_t))) ||
        (rc = ensure(h, l2.d_surv_codes, entries * sizeof(uint64_t))) || (rc = ensure(h, l2.d_listat, n_windows * sizeof(uint32_t) + 16)) ||
        (rc = ensure(h, l2.d_reccnt, lx::kRecCounters * sizeof(uint64_t))) || (rc = ensure(h, l2.d_rec, entries * sizeof(lx_blast_match) + 16)) ||
        (rc = ensure(h, l2.d_reccodes, 3 * entries * sizeof(uint64_t) + 16)) || (rc = ensure(h, l2.d_tilekeep, ((entries + 255) / 256 + 1) * sizeof(uint32_t))) ||
        (rc = ensure(h, l2.d_tileops, ((entries + 255) / 256 + 1) * sizeof(uint64_t))) || (rc = ensure_pinned(h, l2.p_reccnt, 8 * lx::kRecCounters * sizeof(uint64_t))) ||
        (rc = ensure_pinned(h, l2.p_rows, (n_hsps * 3 / 4) * sizeof(lx_blast_match) + 16, hipHostMallocNonCoherent)) ||
        (rc = ensure(h, l2.d_rank, (n_windows + 1) * sizeof(uint32_t) + 16)) || (rc = ensure_pinned(h, l2.p_plan, kPlanHead + 2 * (nwf + 8) * sizeof(uint32_t))) || (rc = ensure(h, l2.d_pre, std::max<size_t>(l2.evlens.size(), 1) * sizeof(double) + 16)) || (rc = ensure(h, l2.d_exp, (1u << 13) * sizeof(double))))
        return rc;
    if (n_columns)
    {
        l2.rec_codes.resize(3 * n_hsps); // (touched: value-initialised)
        if (!h->ext_bytes.grow(n_hsps * 8 + 4096))
            return fail(h, LX_ENOMEM, "lx_reserve: out of host memory");
        std::memset(h->ext_bytes.data(), 0, n_hsps * 8 + 4096);
    }
    hm.mark("records");
    // ---- the extension pipeline's two lanes (extend_pipeline: enqueue_mq), for chunks of the default size; the survivors' column slots for
    // windows of up to three times the ordinary length (what a merged window comes to, src/search_algo.hpp:1153-1157)
    {
        uint64_t const chunk  = h->opt_extend_chunk ? std::max<uint64_t>(h->opt_extend_chunk, 1024) : lxi::kExtendChunk;
        uint64_t const nwf_x  = free_plan ? nwf_solo + nwf_solo / 8 + 64 : nwf_solo; // (what a free-packing plan comes to, not its bound)
        uint64_t const panel  = (uint64_t)lx::trace_cfg_panel(1);
        uint64_t const max_q  = std::max<uint64_t>(1, ((uint64_t)l2.max_qlen + panel - 1) / panel) * panel;
        // (the longest ORDINARY window, query + band on either side, :919-938: a list whose merged windows are longer grows its first chunk's
        // slots in the call)
        uint64_t const max_s  = (uint64_t)l2.max_qlen + 2 * (uint64_t)bandSize(l2.max_qlen) + 16, stride = (max_q + 3 * max_s + 3) & ~3ull;
        uint64_t const steps  = (max_s + 8 - 1 + 15) & ~15ull;
        uint64_t const slot_b = max_q / panel * (lx::ckpt16_slot_dwords(1, (uint32_t)steps) + lx::ckpt_slot_dwords(1, (uint32_t)steps) / 8) * 4;
        // (a device plan's chunks are the RANGES of the list -- level2_sorted_tail: one below 300 000 windows, else two, one more per four
        // million, and as many as keep a range's checkpoint slots within 8 GiB: fresh device memory costs 40 ms per GB --, admitted up to
        // four default chunks: the lanes and the checkpoint slots are sized for the largest range, or the first call grows them)
        uint64_t const R_rule = n_windows < 300000 ? 1 : std::min<uint64_t>(4, n_windows / 4000000 + 2);
        uint64_t const R_est  = std::min<uint64_t>(lx::kFpMaxRanges, std::max<uint64_t>(R_rule, (nwf_x * 16 * slot_b + (8ull << 30) - 1) / (8ull << 30)));
        uint64_t const range  = (R_est == 1 ? nwf_x : R_est == 2 ? nwf_x * 68 / 100 + 64 : nwf_x / R_est + nwf_x / (8 * R_est) + 64) * 16;
        uint64_t const chunk16 = (chunk + 15) / 16 * 16;
        uint64_t const slots  = std::min<uint64_t>(nwf_x * 16, std::max(chunk16, std::min(range, 4 * chunk16))), cap_sel = (slots + 7) / 8 * 8 + 8;
        if ((rc = ensure(h, h->d_score_all, n_windows * sizeof(int32_t) + 16)))
            return rc;
        unsigned const lanes = nwf_x * 16 > slots ? 2 : 1;
        for (unsigned L = 0; L < lanes; ++L)
        {
            lx_handle::XbLane & ln = h->xb[L];
            if ((rc = ensure(h, ln.d_ext, slots * sizeof(lx_extension))) || (rc = ensure(h, ln.d_min, slots * sizeof(int32_t))) ||
                (rc = ensure(h, ln.d_score, slots * sizeof(int32_t))) || (rc = ensure(h, ln.d_hsp, cap_sel * sizeof(lx_hsp))) ||
                (rc = ensure(h, ln.d_ops, cap_sel * stride + 16)) || (rc = ensure(h, ln.d_rle, cap_sel * stride + 16)) ||
                (rc = ensure(h, ln.d_src, cap_sel * sizeof(uint32_t))) || (rc = ensure(h, ln.d_len, cap_sel * sizeof(uint32_t))) ||
                (rc = ensure(h, ln.d_cnt, 5 * sizeof(uint64_t))) || (rc = ensure_pinned(h, ln.p_cnt, 5 * sizeof(uint64_t))))
                return rc;
        }
        hm.mark("lanes");
        // the checkpoint slots of one chunk (what LX_OPT_TRACE_BYTES admits of them), its end cells, the survivor selection's lists
        // (fused_impl in lx_api.cpp: room for every slot plus the padding of query runs), the carry workspace
        uint64_t const trace = std::min<uint64_t>(slots * slot_b, h->opt_trace_bytes);
        if ((rc = ensure(h, h->d_trace, trace)) || (rc = ensure(h, h->d_ends, slots * sizeof(lx::EndCell))) || (rc = ensure(h, h->d_sel_ext, 2 * slots * sizeof(lx_extension))) ||
            (rc = ensure(h, h->d_sel_src, 2 * slots * sizeof(uint32_t))) || (rc = ensure(h, h->d_sel_score, 2 * slots * sizeof(int32_t))) ||
            (rc = ensure(h, h->d_sel_runs, (2 * slots + 4096) * sizeof(uint64_t))) || (rc = ensure(h, h->d_ws, std::max(h->opt_ws_bytes, h->ws_grown))))
            return rc;
    }
    hm.mark("checkpoint slots + selection");
    // ---- every kernel of the call once, on a list of one match (query 0 against the start of subject 0, no filter: it survives and is
    // traced): what a kernel's first launch costs the runtime (2-3 ms over the call's two dozen kernels) is paid here
    if (!l2.q_len.empty() && !l2.s_len.empty() && h->db_bytes && l2.s_extent <= h->db_bytes && (h->have_sc[0] || h->have_sc[1]) && !h->opt_band && l2.s_len[0] > 0)
    {
        lx_match const one{0, 0, 0, std::min<uint64_t>(l2.q_len[0], l2.s_len[0]), 0, std::min<uint64_t>(l2.q_len[0], l2.s_len[0])};
        if ((rc = ensure(h, l2.d_up, sizeof(lx_match) + 16)))
            return rc;
        LX_HIP(h, hipMemcpyAsync(l2.d_up.ptr, &one, sizeof(one), hipMemcpyHostToDevice, h->stream));
        LX_HIP(h, hipStreamSynchronize(h->stream));
        lx_search_params sp{};
        sp.max_evalue      = -1;
        sp.min_bitscore    = -1;
        sp.db_total_length = 1000000;
        sp.qry_num_frames  = l2.q_frames;
        sp.sbj_num_frames  = 1;
        sp.karlin          = lx_karlin{0.3, 0.1, 0.3, 1.0, -10.0};
        sp.flags           = n_columns ? 0 : LX_ITERATE_NO_OPS;
        lx_iterate_result * r = nullptr;
        // (the dummy call must not teach the handle anything: what the adaptive pass 2 and the wide sweep learnt from real lists stays)
        double const   surv_before = h->surv_frac, plan_before = h->plan_surv_frac, decl_before = h->mq_decl_frac;
        bool const     wide_before = h->mq_wide_call, pending_before = h->count_pending;
        uint64_t const res_before  = h->res_count;
        rc = lx_iterate_matches_dev(h, h->have_sc[0] ? 0 : 1, l2.d_up.ptr, 1, &sp, &r);
        h->surv_frac      = surv_before;
        h->plan_surv_frac = plan_before;
        h->mq_decl_frac   = decl_before;
        h->mq_wide_call   = wide_before;
        h->count_pending  = pending_before;
        h->res_count      = res_before;
        h->phase_ev.clear();
        h->ev_pool_used   = 0;
        if (r)
            lx_iterate_result_free(r);
        if (rc)
            return rc;
        l2.exp_lambda = 0; // (the table of exp(-lambda s) the dummy call left is not the search's)
    }
    hm.mark("one-match call");
    // ---- the result's memory: blocks of the sizes the first result will ask for, written once (a page the kernel has not handed out
    // yet costs a fault when the rows' copy or the expanding threads reach it), kept where a result's arrays are taken from
    for (uint64_t bytes : {n_hsps * (uint64_t)sizeof(lx_blast_match), n_columns})
    {
        if (bytes < lambda_amd::BlockCache::kMinBytes)
            continue;
        lambda_amd::RawVec<uint8_t> block;
        if (!block.resize(bytes))
            return fail(h, LX_ENOMEM, "lx_reserve: out of host memory for the result blocks");
        uint8_t * const b = block.data();
        lambda_amd::parallelRanges(bytes / 4096, [&](unsigned, uint64_t lo, uint64_t hi) { std::memset(b + lo * 4096, 0, (hi - lo) * 4096); });
    } // (the destructor hands the block to the cache)
    hm.mark("result blocks");
    return LX_OK;
}

// The Level-2 radix sort (lx_level2.hip: l2_launch_sort) for a caller's own (key, value) words in device memory -- what the front end's
// word table is sorted with (host/lx_seeding_gpu.hpp), instead of a library primitive.
int lx_sort_words_dev(int device, uint64_t * key[2], uint64_t * value[2], uint64_t n, uint64_t key_bits, void * stream, int * sorted_in)
{
    if (!key || !value || !sorted_in || (n && (!key[0] || !key[1] || !value[0] || !value[1])) || n > 0x7fffffffull || device < 0 || device >= 64)
        return LX_EINVAL;
    *sorted_in = 0;
    if (n < 2)
        return LX_OK;
    // (the caller's current device is the caller's: restored on every exit)
    int before = -1;
    if (hipGetDevice(&before) != hipSuccess)
        before = -1;
    struct Restore
    {
        int dev;
        ~Restore()
        {
            if (dev >= 0)
                (void)hipSetDevice(dev);
        }
    } const restore{before == device ? -1 : before};
    if (hipSetDevice(device) != hipSuccess)
        return LX_EHIP;
    // the digit counts: one allocation per (process, device), kept and grown -- a hipMalloc + hipFree per call synchronises the device
    static std::mutex  hist_m[64];
    static uint32_t *  hist_buf[64] = {nullptr};
    static size_t      hist_cap[64] = {0};
    size_t const want = (lx::l2_sort_tiles(n) + 2) * 256 * sizeof(uint32_t);
    int const    d    = device & 63;
    std::lock_guard<std::mutex> lk(hist_m[d]); // (one sort at a time per DEVICE on its buffer; the devices of a node sort side by side)
    if (hist_cap[d] < want)
    {
        if (hist_buf[d])
            (void)hipFree(hist_buf[d]);
        hist_buf[d] = nullptr;
        hist_cap[d] = 0;
        if (hipMalloc(reinterpret_cast<void **>(&hist_buf[d]), want + want / 2) != hipSuccess)
            return LX_ENOMEM;
        hist_cap[d] = want + want / 2;
    }
    uint32_t * const hist = hist_buf[d];
    uint64_t * k = key[0], * kt = key[1], * v = value[0], * vt = value[1];
    hipError_t e = lx::l2_launch_sort(&k, &kt, &v, &vt, n, key_bits, 0, hist, static_cast<hipStream_t>(stream));
    if (e == hipSuccess)
        e = hipStreamSynchronize(static_cast<hipStream_t>(stream)); // (the digit counts are shared: the sort is over before the next one starts)
    *sorted_in = k == key[0] ? 0 : 1;
    return e == hipSuccess ? LX_OK : LX_EHIP;
}

// The free-packing plan (lx_plan_free.hip) of a window list in device memory, brought to the host: what tests/test_gpu_plan.py checks
// slot by slot.  out_plan: [lx_plan_free_packing_bound(n, n_qseq, nranges) * 16], out_pan / out_maxs: [that bound], out_report: [16].
uint64_t lx_plan_free_packing_bound(uint64_t n, uint64_t n_qseq, uint32_t nranges)
{
    return lx::fp_wavefront_bound(n, n_qseq, nranges);
}

int lx_plan_free_packing_dev(lx_handle * h, void const * d_ext, uint64_t n, uint64_t n_qseq, int32_t strip_cols, uint32_t nranges, uint64_t const * cut, uint32_t * out_plan,
                             uint32_t * out_pan, uint32_t * out_maxs, uint32_t * out_report)
{
    if (!h)
        return LX_EINVAL;
    if (!d_ext || !cut || !out_plan || !out_pan || !out_maxs || !out_report || n == 0 || nranges == 0 || nranges > lx::kFpMaxRanges || n >= 0x7ffffff0ull ||
        (strip_cols != 19 && strip_cols != 13 && strip_cols != 11))
        return fail(h, LX_EINVAL, "lx_plan_free_packing_dev: arguments");
    for (uint32_t r = 0; r < nranges; ++r)
        if (cut[r] >= cut[r + 1])
            return fail(h, LX_EINVAL, "lx_plan_free_packing_dev: range %u is empty", r);
    if (cut[0] != 0 || cut[nranges] != n)
        return fail(h, LX_EINVAL, "lx_plan_free_packing_dev: the ranges must cover the list");
    int rc = bind(h);
    if (rc)
        return rc;
    auto &         l2  = h->l2;
    uint64_t const cap = lx::fp_wavefront_bound(n, n_qseq, nranges);
    lx::FpArgs     fa{};
    fa.ext        = static_cast<lx::Extension const *>(d_ext);
    fa.n          = n;
    fa.n_qseq     = n_qseq;
    fa.C          = strip_cols;
    fa.no_narrow  = 0;
    fa.nranges    = nranges;
    for (uint32_t r = 0; r <= nranges; ++r)
        fa.cut[r] = cut[r];
    fa.work_bytes = lx::fp_workspace_bytes(n, n_qseq, nranges);
    if ((rc = ensure(h, l2.d_fp, fa.work_bytes + 64)) || (rc = ensure(h, l2.d_plan, cap * 16 * sizeof(uint32_t) + 16)) || (rc = ensure(h, l2.d_wf, 2 * cap * sizeof(uint32_t) + 64)))
        return rc;
    fa.work    = l2.d_fp.ptr;
    fa.plan    = static_cast<uint32_t *>(l2.d_plan.ptr);
    fa.cap_wf  = cap;
    fa.wf_pan  = static_cast<uint32_t *>(l2.d_wf.ptr);
    fa.wf_maxs = fa.wf_pan + cap;
    fa.report  = fa.wf_pan + 2 * cap;
    LX_HIP(h, lx::fp_launch_plan(fa, h->stream));
    LX_HIP(h, hipMemcpyAsync(out_report, fa.report, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    LX_HIP(h, hipMemcpyAsync(out_plan, fa.plan, cap * 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    LX_HIP(h, hipMemcpyAsync(out_pan, fa.wf_pan, cap * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    LX_HIP(h, hipMemcpyAsync(out_maxs, fa.wf_maxs, cap * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    LX_HIP(h, hipStreamSynchronize(h->stream));
    return LX_OK;
}

// _widenAndPreprocessMatches (src/search_algo.hpp:1136-1175) alone, on a device match list over the resident sets
int lx_widen_and_preprocess_dev(lx_handle * h, void const * d_matches, uint64_t n_matches, int32_t bisulfite, lx_match * out, uint64_t * out_n)
{
    if (!h || !out_n)
        return LX_EINVAL;
    *out_n = 0;
    if ((!d_matches || !out) && n_matches)
        return fail(h, LX_EINVAL, "NULL argument");
    if (n_matches > 0x7ffffff0ull)
        return fail(h, LX_EINVAL, "at most 2^31 matches per call");
    auto & l2 = h->l2;
    if (l2.q_len.empty() || l2.s_len.empty())
        return fail(h, LX_ESTATE, "lx_widen_and_preprocess_dev needs the resident sets: lx_set_queries, lx_set_subject_seqs");
    int rc = bind(h);
    if (rc || n_matches == 0)
        return rc;
    std::vector<int32_t> const table((size_t)l2.max_evlen + 1, 0);
    uint64_t                   nw = 0, nEven = 0;
    if ((rc = level2_keys(h, d_matches, n_matches, bisulfite != 0)) || (rc = level2_windows(h, n_matches, bisulfite != 0, table, nw, nEven)))
        return rc;
    if (nw)
    {
        if ((rc = ensure_pinned(h, l2.p_win, nw * sizeof(lx::L2Window) + 16)))
            return rc;
        LX_HIP(h, hipMemcpyAsync(l2.p_win.ptr, l2.d_win.ptr, nw * sizeof(lx::L2Window), hipMemcpyDeviceToHost, h->stream));
        LX_HIP(h, hipStreamSynchronize(h->stream));
        lx::L2Window const * const win = static_cast<lx::L2Window const *>(l2.p_win.ptr);
        for (uint64_t i = 0; i < nw; ++i)
            out[i] = lx_match{win[i].q, win[i].s, 0, l2.q_len[win[i].q], win[i].beg, win[i].end};
    }
    *out_n = nw;
    return LX_OK;
}

} // extern "C"

// lx_level2_host.cpp -- the Level-2 driver on a DEVICE match list: lx_set_queries, lx_set_subject_seqs, lx_iterate_matches_dev.
//
// iterateMatchesFullSimd (/root/reference/src/search_algo.hpp:1177-1332) for matches that the seeding stage left in HBM: widen,
// sort, merge and unique (:1136-1175), the slices (:1200-1227) and the filter's cut-offs (:1251-1283) run as kernels
// (lx_level2.hip) on the resident sequence sets; the host receives the finished window list (24 B per WINDOW, a tenth of the
// matches), plans the sweep over it, and the extension pipeline (lx_host.cpp) reads list and cut-offs where the kernels wrote
// them.  What follows the two passes -- statistics, order, records (:1287-1325) -- is host/lx_iterate_common.hpp, shared with
// lx_iterate_matches.  lx_iterate_matches itself hands its lists to this path when they are large (host/lx_driver.cpp).
#include <cmath>

#include "lx_internal.h"
#include "lx_level2.h"

#include "host/lx_iterate_common.hpp"

using namespace lxi;

namespace
{

int ensure_pinned(lx_handle * h, lx_handle::Pinned & b, size_t bytes)
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
    LX_HIP(h, hipHostMalloc(&b.ptr, want, hipHostMallocDefault));
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
    if ((rc = ensure(h, l2.d_qres, q_bytes + kSlack)))
        return rc;
    if (q_bytes)
        LX_HIP(h, hipMemcpyAsync(l2.d_qres.ptr, q_res, q_bytes, hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemsetAsync(static_cast<uint8_t *>(l2.d_qres.ptr) + q_bytes, 0, kSlack, h->stream));
    if ((rc = upload(h, l2.d_qoff, q_off)) || (rc = upload(h, l2.d_qlen, q_len)) || (rc = upload(h, l2.d_qband, band)) ||
        (rc = upload(h, l2.d_qevlen, q_evlen)))
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
    l2.q_bytes = q_bytes;
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
    LX_HIP(h, lx::l2_launch_plan_cost(p.ext_out, p.count_out, n_matches, lx::dev_aids().mq_no_narrow ? 1 : 0,
                                      reinterpret_cast<unsigned long long *>(p.count_out + 3), st));
    LX_HIP(h, hipMemcpyAsync(l2.p_cnt.ptr, l2.d_cnt.ptr, 11 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
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

// The list work and the extension for matches that stand on the device as sort words already.  Appends to *res.
static int level2_sorted_tail(lx_handle * h, int slot, uint64_t n_matches, lx_search_params const * params, lx_iterate_result * res)
{
    using namespace lambda_amd;
    auto &            l2 = h->l2;
    hipStream_t const st = h->stream;
    HostMarks         hm("lx_iterate_matches_dev");
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
    // the window list comes down (24 B per window) on the copy stream, beside what follows: the records need it, and the plan of
    // the sweep where it is made on the host
    if ((rc = ensure_pinned(h, l2.p_win, nw * sizeof(lx::L2Window) + 16)))
        return rc;
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
        LX_HIP(h, hipEventRecord(l2.ev_win, st));
        LX_HIP(h, hipStreamWaitEvent(h->stream2, l2.ev_win, 0));
        LX_HIP(h, hipMemcpyAsync(l2.p_win.ptr, l2.d_win.ptr, nw * sizeof(lx::L2Window), hipMemcpyDeviceToHost, h->stream2));
        LX_HIP(h, hipEventRecord(l2.ev_win, h->stream2));
        return LX_OK;
    };
    bool                       win_here = false;
    lx::L2Window const * const win      = static_cast<lx::L2Window const *>(l2.p_win.ptr);
    l2.score.resize(nw);
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
        if (solo_plan_applies(h, pt.slot))
        {
            // The plan of the sweep on the device (the solo packing of lx_sweep_mq.hip: every window its own profile): ONE strip
            // geometry per call, the one that sweeps the list cheapest (lx_host.cpp has the measurement behind "one" and behind
            // the 15 % that narrower strips must save), then all windows by (columns per lane, length), 16 to a wavefront.
            int const cand[3] = {1, 3, 5}, set = lx::dev_aids().mq_set, forced = lx::dev_aids().force_mq_cfg;
            int       cfg     = 1;
            double    best    = 1e300;
            for (int k = 0; k < 3; ++k)
            {
                double c = (double)cost[4 * pi + k] * (cand[k] == 1 ? 1.0 : 1.15);
                if (!(set & (1 << k)) && forced != cand[k])
                    continue;
                if (forced == cand[k])
                    c = 0;
                if (c < best)
                {
                    best = c;
                    cfg  = cand[k];
                }
            }
            uint64_t const nwf = (n + 15) / 16;
            if ((rc = ensure(h, l2.d_plan, nwf * 16 * sizeof(uint32_t) + 16)) || (rc = ensure(h, l2.d_wf, 2 * nwf * sizeof(uint32_t) + 16)))
                return rc;
            uint64_t * key = static_cast<uint64_t *>(l2.d_pair[0].ptr), * key_tmp = static_cast<uint64_t *>(l2.d_pair[1].ptr);
            uint64_t * idx = static_cast<uint64_t *>(l2.d_s0[0].ptr), * idx_tmp = static_cast<uint64_t *>(l2.d_s0[1].ptr);
            uint32_t * const d_pan = static_cast<uint32_t *>(l2.d_wf.ptr), * const d_maxs = d_pan + nwf;
            LX_HIP(h, lx::l2_launch_plan(static_cast<lx::Extension const *>(ri.d_ext_all), n, lx::trace_cfg_panel(cfg) / 8, lx::dev_aids().mq_no_narrow ? 1 : 0, &key,
                                         &key_tmp, &idx, &idx_tmp, static_cast<uint32_t *>(l2.d_hist.ptr), static_cast<uint32_t *>(l2.d_plan.ptr), d_pan, d_maxs, st));
            l2.wf_pan.resize(nwf);
            l2.wf_maxs.resize(nwf);
            LX_HIP(h, hipMemcpyAsync(l2.wf_pan.data(), d_pan, nwf * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            LX_HIP(h, hipMemcpyAsync(l2.wf_maxs.data(), d_maxs, nwf * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            if ((rc = queue_windows()))
                return rc;
            LX_HIP(h, hipStreamSynchronize(st));
            ri.d_plan  = static_cast<uint32_t const *>(l2.d_plan.ptr);
            ri.nwf     = nwf;
            ri.wf_pan  = l2.wf_pan.data();
            ri.wf_maxs = l2.wf_maxs.data();
            ri.mq_cfg  = cfg;
            ri.cells   = cost[4 * pi + 3];
            hm.mark("plan");
            if ((rc = extend_list_resident(h, pt.slot, ri, nullptr, n, nullptr, l2.score.data() + pt.lo, &list)))
                return rc;
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
            if ((rc = extend_list_resident(h, pt.slot, ri, l2.ext.data() + pt.lo, n, l2.min.data() + pt.lo, l2.score.data() + pt.lo, &list)))
                return rc;
        }
        hm.mark("extension");
        if (!win_here)
        {
            LX_HIP(h, hipEventSynchronize(l2.ev_win));
            win_here = true;
        }
        uint64_t const base = pt.lo;
        if ((rc = finishSurvivors(n, [&](uint64_t i) { return window(base + i); }, l2.score.data() + pt.lo, [&](uint64_t i) { return minOf(base + i); }, list, params, res)))
            return fail(h, rc, "out of host memory for the result records");
        hm.mark("statistics+records");
    }
    if (params->bisulfite) // the HSPs are stably re-sorted by query (:1379); the ops offsets stay valid: only the records move
        std::stable_sort(res->matches.begin(), res->matches.end(), [](lx_blast_match const & a, lx_blast_match const & b) { return a.n_qid < b.n_qid; });
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
    if (n_matches < 4 * kParallelFrom || params->band > 0 || h->opt_band || s_res != nullptr || s_bytes != 0 || !h->db_bytes || n_qseq >= (1ull << 31) ||
        n_sseq >= (1ull << 32) || lx::dev_aids().iterate_on_host)
        return kNotTaken;
    if (!params->bisulfite && (slot < 0 || slot > 1 || !h->have_sc[slot]))
        return kNotTaken; // (the host form reports it)
    if (params->bisulfite && (!h->have_sc[0] || !h->have_sc[1]))
        return kNotTaken;
    auto &    l2 = h->l2;
    HostMarks hm("lx_iterate_matches (device list work)");
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
    rc             = level2_sorted_tail(h, slot, n_matches, params, res);
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
        rc = level2_sorted_tail(h, slot, n_matches, params, res);
    h->opt_bs_rule = ruleBefore;
    if (rc != LX_OK)
    {
        delete res;
        return rc;
    }
    *out = res;
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

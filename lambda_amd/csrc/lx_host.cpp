// lx_host.cpp -- lx_extend_batch / _rle / _list: the fused step (pass 1, filter, pass 2 -- _performAlignment at
// /root/reference/src/search_algo.hpp:1246 and :1296 with the filter loop :1251-1283 between them) on lists in host memory, as a
// pipeline of chunks (pinned staging, two chunks in flight, run-length coded ops on the wire), and the same pipeline over a list
// that is resident on the device already (lxi::extend_list_resident, what the Level-2 driver calls).  Ragged lists are planned for
// the multi-query sweep here (lx_plan_free.hip plans device lists).  The entry points of the single passes are lx_host_batch.cpp;
// the device entry points all of them drive live in lx_api.cpp; no DP arithmetic here.
#include "lx_internal.h"
#include "lx_level2.h"
using namespace lxi;

// Both passes on host buffers, as a pipeline of chunks.  The list is cut at query-run boundaries into chunks of a few
// hundred thousand extensions; per chunk the host groups the extensions by query slice and pads every run to 16 (or 8)
// slots -- the promise LX_OPT_QUERY_RUN makes to the device path -- into pinned staging, the GPU runs the whole fused step
// (sweep -> selection -> backtrace -> run-length packing of the ops, nothing in between comes back to the host), and the
// results return as scores + the survivors' records + their run-length codes.  Two chunks are in flight: uploads and
// downloads of one run on copy streams while the other's kernels run, and the host prepares chunk k + 1 / unpacks
// chunk k - 1 meanwhile.  What crosses PCIe per extension: 28 B up, 4 B + (survivors) 52 B + ~8 B of codes down.
namespace
{

struct XbPrep // what the host keeps about a chunk until its results are back
{
    uint64_t              k0 = 0, k1 = 0; // positions in the ordered list
    uint64_t              slots = 0, cap_sel = 0;
    bool                  wide = false;   // multi-query chunk: the sweep wrote int16-pair slots
    uint64_t              exec_cells = 0, max_s = 0, max_pan = 0; // (LX_HOST_TIMING: what the chunk's wavefronts execute)
    uint64_t              range = 0;      // records chunk by chunk (ResidentInput::ChunkRecords): the range this chunk is
    std::vector<uint32_t> slot_src;       // original index of every slot (0xffffffff = padding)
};

inline void rle_expand(uint8_t const * codes, int32_t n_ops, uint8_t * out)
{
    static char const kOp[4] = {'M', 'D', 'I', 'M'};
    int32_t done = 0;
    while (done < n_ops)
    {
        uint8_t const c   = *codes++;
        int32_t const len = (c & 63) + 1;
        if (done + ((len + 15) & ~15) <= n_ops)
        {
            // whole 16-byte stores while they stay inside this alignment's columns (the surplus is overwritten by the runs that
            // follow; a call to memset per run of a few columns costs more than the stores)
            for (int32_t k = 0; k < len; k += 16)
                std::memset(out + done + k, kOp[c >> 6], 16);
        }
        else
            std::memset(out + done, kOp[c >> 6], (size_t)len);
        done += len;
    }
}

inline uint64_t rle_length(uint8_t const * codes, int32_t n_ops)
{
    uint64_t k = 0;
    for (int32_t done = 0; done < n_ops; ++k)
        done += (codes[k] & 63) + 1;
    return k;
}

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

} // namespace

static int extend_pipeline(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                           lx_extension const * ext, uint64_t n, int32_t const * min_score, int32_t min_score_all, int32_t * out_score,
                           lx_hsp * out_hsp, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes, int mode,
                           lxi::ResidentInput const * ri = nullptr)
{
    // ri (lx_level2_host.cpp): the query residues and the caller's list + cut-offs stand on the device already (q_res is NULL, q_bytes the
    // resident size; `ext` / `min_score` are the host's copies of the same list, for the plan)
    // mode 0: column bytes, 1: run-length codes, 2: the survivors as a list in the handle's buffers (out_hsp, out_ops_off,
    // out_ops, out_ops_bytes are NULL; lx_extend_batch_list hands the buffers out)
    bool const want_rle = mode >= 1, as_list = mode == 2;
    HostPool::Call const in_flight_call; // (the host threads look for this call's next loop instead of going to sleep between two)
    bool       dev_list = false, want_codes = true, by_range = false; // (set where the multi-query plan is known: ResidentInput::keep_on_device)
    h->res_count         = 0;
    h->l2.surv_on_device = false;
    h->l2.surv_by_range  = false;
    int rc = bind(h);
    if (rc)
        return rc;
    SubjectRef sref;
    if ((rc = resolve_subjects(h, s_res, s_bytes, sref)))
        return rc;
    s_bytes = sref.bytes;
    HostMarks hm(as_list ? "lx_extend_batch_list" : want_rle ? "lx_extend_batch_rle" : "lx_extend_batch");

    // A plan that was made on the device (lx_level2_host.cpp: the solo packing of a resident window list -- a sort by width and
    // length, 16 windows to a wavefront): nothing of the list is looked at here, `ext` may be NULL.
    bool const preplanned = ri && ri->d_plan;
    // ---- validate; is the list grouped by query (lambda's lists are sorted by query)?  The loops over the list are spread
    // over a few host threads: at millions of extensions per call they would otherwise cost more than the kernels.
    unsigned const nthreads = host_threads(n);
    struct Part
    {
        uint64_t live = 0, bad = ~0ull;
        bool     monotone = true;
    };
    std::vector<Part> parts(nthreads);
    // (the same pass writes what the common case needs -- every extension live, the list grouped by query already: the order
    // is the identity and the runs begin where the slice changes)
    std::vector<uint32_t> & idx    = h->xb_idx;
    std::vector<uint8_t> &  newrun = h->xb_newrun;
    if (!preplanned)
    {
        idx.resize(n);
        newrun.resize(n + 1);
    }
    parallel_ranges(preplanned ? 0 : n, nthreads,
                    [&](unsigned t, uint64_t lo, uint64_t hi)
                    {
                        Part     pt;           // (a local: the per-thread slots share cache lines)
                        uint64_t prev = ~0ull; // last live extension before i (of the whole list)
                        for (uint64_t i = lo; i-- > 0;)
                            if (ext[i].q_len != 0 && ext[i].s_len != 0)
                            {
                                prev = i;
                                break;
                            }
                        for (uint64_t i = lo; i < hi; ++i)
                        {
                            lx_extension const & x = ext[i];
                            if (!lx_slice_ok(x.q_off, x.q_len, q_bytes) || !lx_slice_ok(x.s_off, x.s_len, s_bytes))
                            {
                                pt.bad = std::min(pt.bad, i);
                                continue;
                            }
                            if (x.q_len == 0 || x.s_len == 0)
                            {
                                out_score[i] = 0;
                                if (!as_list)
                                {
                                    out_hsp[i]     = lx_hsp{};
                                    out_ops_off[i] = 0;
                                }
                                continue;
                            }
                            if (prev != ~0ull && x.q_off < ext[prev].q_off)
                                pt.monotone = false;
                            idx[i]    = (uint32_t)i;
                            newrun[i] = (i == 0 || x.q_off != ext[i - 1].q_off || x.q_len != ext[i - 1].q_len) ? 1 : 0;
                            prev      = i;
                            ++pt.live;
                        }
                        parts[t] = pt;
                    });
    uint64_t live = 0;
    bool     monotone = true;
    for (Part const & pt : parts)
    {
        if (pt.bad != ~0ull)
            return fail(h, LX_EINVAL, "extension %llu exceeds the residue buffers", (unsigned long long)pt.bad);
        live += pt.live;
        monotone = monotone && pt.monotone;
    }
    if (preplanned)
        live = n;
    if (live == 0)
        return LX_OK;
    hm.mark("scan");
    bool const as_given = (live == n && monotone) || preplanned;
    if (!preplanned)
        idx.resize(live);
    if (!as_given)
    {
        std::vector<uint64_t> first(nthreads + 1, 0);
        for (unsigned t = 0; t < nthreads; ++t)
            first[t + 1] = first[t] + parts[t].live;
        parallel_ranges(n, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t o = first[t];
                            for (uint64_t i = lo; i < hi; ++i)
                                if (ext[i].q_len != 0 && ext[i].s_len != 0)
                                    idx[o++] = (uint32_t)i;
                        });
    }
    if (!monotone) // anything else is sorted first: equal slices become adjacent
        std::sort(idx.begin(), idx.end(),
                  [&](uint32_t a, uint32_t b)
                  {
                      lx_extension const &x = ext[a], &y = ext[b];
                      return x.q_off != y.q_off ? x.q_off < y.q_off : x.q_len != y.q_len ? x.q_len < y.q_len : a < b;
                  });
    auto same_slice = [&](uint32_t a, uint32_t b) { return ext[a].q_off == ext[b].q_off && ext[a].q_len == ext[b].q_len; };
    // where the runs of one query slice begin in the ordered list
    if (!preplanned)
        newrun.resize(live + 1);
    if (!as_given)
        parallel_ranges(live, nthreads,
                        [&](unsigned, uint64_t lo, uint64_t hi)
                        {
                            for (uint64_t k = lo; k < hi; ++k)
                                newrun[k] = (k == 0 || !same_slice(idx[k], idx[k - 1])) ? 1 : 0;
                        });
    if (!preplanned)
        newrun[live] = 1;
    hm.mark("order");
    // positions where the runs of one query slice begin, + the sentinel `live` (two parallel passes over newrun)
    std::vector<uint64_t> & starts = h->xb_starts;
    starts.clear();
    auto run_starts = [&](std::vector<uint64_t> & out)
    {
        std::vector<uint64_t> cnt(nthreads + 1, 0);
        parallel_ranges(live + 1, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t c = 0;
                            for (uint64_t k = lo; k < hi; ++k)
                                c += newrun[k];
                            cnt[t + 1] = c;
                        });
        for (unsigned t = 0; t < nthreads; ++t)
            cnt[t + 1] += cnt[t];
        out.resize(cnt[nthreads]);
        parallel_ranges(live + 1, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t o = cnt[t];
                            for (uint64_t k = lo; k < hi; ++k)
                                if (newrun[k])
                                    out[o++] = k;
                        });
    };
    // Multi-query sweep (lx_sweep_mq.hip) for lists that are not uniform: sub-blocks of 4 windows of one query, ordered by
    // (geometry class, longest window) ACROSS queries, four sub-blocks per wavefront -- see the plan below.  Needs what
    // fused_impl's mq branch needs; uniform lists (one query length, one window length, runs that fill whole wavefronts) stay
    // on the one-query-per-wavefront kernels.
    bool use_mq = false;
    {
        lx_scoring const & sh = h->sc_host[slot];
        use_mq = h->opt_mq >= 1 && h->opt_pass2 == 2 && h->opt_f16 && h->trace_ok[slot] && h->b8_ok[slot] && -sh.gap_open <= lx::kC16MaxGap &&
                 sh.gap_open <= sh.gap_extend;
    }
    // the caller's list and cut-offs onto the device (pinned staging, filled by the host threads; scores in caller order zeroed) -- as
    // soon as the multi-query path is known to be taken: the copy runs beside the planning of the pool
    double t_upload      = 0;
    bool   list_uploaded = false;
    auto   upload_list   = [&]() -> int
    {
        int rc2;
        auto const tu0 = std::chrono::steady_clock::now();
        uint64_t const ext_bytes = n * sizeof(lx_extension), min_bytes = min_score ? n * sizeof(int32_t) : 0;
        // (the scores of a list whose records are made on the device stay there: no pinned block for their way down)
        if ((rc2 = ensure(h, h->d_score_all, n * sizeof(int32_t) + 16)) ||
            (!(as_list && ri && ri->keep_on_device) && (rc2 = ensure_pinned(h, h->p_score_all, n * sizeof(int32_t) + 16))))
            return rc2;
        if (!ri) // (a resident list stands where the Level-2 kernels wrote it)
        {
            if ((rc2 = ensure_pinned(h, h->p_all, ext_bytes + min_bytes + 16)) || (rc2 = ensure(h, h->d_ext_all, ext_bytes + 16)) ||
                (rc2 = ensure(h, h->d_min_all, min_bytes + 16)))
                return rc2;
            uint8_t * const stage_all = static_cast<uint8_t *>(h->p_all.ptr);
            parallel_ranges(n, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                std::memcpy(stage_all + lo * sizeof(lx_extension), ext + lo, (hi - lo) * sizeof(lx_extension));
                                if (min_score)
                                    std::memcpy(stage_all + ext_bytes + lo * sizeof(int32_t), min_score + lo, (hi - lo) * sizeof(int32_t));
                            });
            LX_HIP(h, hipMemcpyAsync(h->d_ext_all.ptr, stage_all, ext_bytes, hipMemcpyHostToDevice, h->stream));
            if (min_score)
                LX_HIP(h, hipMemcpyAsync(h->d_min_all.ptr, stage_all + ext_bytes, min_bytes, hipMemcpyHostToDevice, h->stream));
        }
        LX_HIP(h, hipMemsetAsync(h->d_score_all.ptr, 0, n * sizeof(int32_t), h->stream));
        t_upload      = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count();
        list_uploaded = true;
        return LX_OK;
    };
    // The SOLO packing of that sweep (lx_sweep_mq.hip, LX_OPT_QUERY_RUN = 1): a byte profile per window, 16 windows of any queries
    // per wavefront -- where 16 profiles fit a wavefront's share of the LDS, i.e. the alphabets of at most six rows (nucleotides,
    // bisulfite).  A read set's seed list has one or two windows per read: at four queries per wavefront three slots in four
    // were fillers (configs[2]-sized list: 4.1 M slots for 1.25 M windows).  The plan is then a sort: all windows by (columns per
    // lane, length), longest first, 16 to a wavefront.
    bool const use_solo = preplanned ? !ri->free_packing
                                     : (use_mq && lx::sweep_mq_lds_bytes(1, h->sc_host[slot].alphabet_size + 1, -1) <= 20 * 1024);
    if (preplanned)
    {
        if (!use_mq || !as_list)
            return fail(h, LX_ESTATE, "a device plan needs the multi-query sweep and the list form");
    }
    else
    // Mixed query lengths (a real seed list; the synthetic batches have one): a chunk runs the kernel geometry of its longest
    // query, so runs are dealt to geometry classes first -- one panel of 152 columns, one of 208, two / three / ... panels of
    // 152 -- and every class goes through the pipeline by itself.  Inside a run the windows are ordered by length (merged
    // windows are up to 3 x longer: src/search_algo.hpp:1153-1157), so that a wavefront's 16 windows take about as many steps
    // each -- the reason the reference sorts its SIMD batches (:1229-1235).  Results are scattered by original index anyway.
    {
        auto qclass = [](uint32_t lq) -> uint32_t { return lq <= 104 ? 0u : lq <= 152 ? 1u : lq <= 200 ? 2u : lq <= 208 ? 3u : 3u + (lq + 151) / 152; };
        uint32_t cmin = ~0u, cmax = 0;
        bool     ragged_s = false;
        {
            struct Scan
            {
                uint32_t cmin = ~0u, cmax = 0;
                bool     ragged = false;
            };
            std::vector<Scan> scans(nthreads);
            parallel_ranges(live, nthreads,
                            [&](unsigned t, uint64_t lo, uint64_t hi)
                            {
                                Scan sc; // (a local: the per-thread slots share cache lines)
                                for (uint64_t k = lo; k < hi; ++k)
                                {
                                    if (newrun[k])
                                    {
                                        uint32_t const c = qclass(ext[idx[k]].q_len);
                                        sc.cmin = std::min(sc.cmin, c);
                                        sc.cmax = std::max(sc.cmax, c);
                                    }
                                    else if (ext[idx[k]].s_len != ext[idx[k - 1]].s_len)
                                        sc.ragged = true;
                                }
                                scans[t] = sc;
                            });
            for (Scan const & sc : scans)
            {
                cmin     = std::min(cmin, sc.cmin);
                cmax     = std::max(cmax, sc.cmax);
                ragged_s = ragged_s || sc.ragged;
            }
        }
        if (use_mq && cmin == cmax && !ragged_s && h->opt_mq < 2)
        {
            // one geometry, one window length: uniform if every run fills whole wavefronts
            run_starts(starts);
            bool all16 = true;
            for (size_t r = 0; r + 1 < starts.size() && all16; ++r)
                all16 = (starts[r + 1] - starts[r]) % 16 == 0;
            if (all16)
                use_mq = false;
        }
        if (use_mq && (rc = upload_list()))
            return rc;
        if (cmin != cmax && !use_mq)
        {
            std::vector<uint64_t> at(cmax + 2, 0);
            for (uint64_t k = 0; k < live;)
            {
                uint64_t kk = k + 1;
                while (!newrun[kk])
                    ++kk;
                at[qclass(ext[idx[k]].q_len) + 1] += kk - k;
                k = kk;
            }
            for (uint32_t c = 0; c <= cmax; ++c)
                at[c + 1] += at[c];
            std::vector<uint32_t> & idx2 = h->xb_src;
            idx2.resize(live);
            for (uint64_t k = 0; k < live;)
            {
                uint64_t kk = k + 1;
                while (!newrun[kk])
                    ++kk;
                uint64_t & o = at[qclass(ext[idx[k]].q_len)];
                std::copy(idx.begin() + k, idx.begin() + kk, idx2.begin() + o);
                o += kk - k;
                k = kk;
            }
            idx.swap(idx2);
            parallel_ranges(live, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                for (uint64_t k = lo; k < hi; ++k)
                                    newrun[k] = (k == 0 || !same_slice(idx[k], idx[k - 1])) ? 1 : 0;
                            });
        }
        if (ragged_s && !(use_mq && use_solo)) // (the solo plan sorts all windows itself)
        {
            run_starts(starts);
            parallel_ranges(starts.size() - 1, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                for (uint64_t r = lo; r < hi; ++r)
                                    std::sort(idx.begin() + starts[r], idx.begin() + starts[r + 1],
                                              [&](uint32_t a, uint32_t b) { return ext[a].s_len != ext[b].s_len ? ext[a].s_len < ext[b].s_len : a < b; });
                            });
        }
    }
    // ---- multi-query plan (free packing of lx_sweep_mq.hip: the two windows of a lane group share a query, a wavefront's 16 slots
    // hold windows of at most four queries in any split).  Inside a run the windows are sorted by length; those clearly longer than
    // the run's median -- the merged windows, up to 3 x longer (src/search_algo.hpp:1153-1157) -- go to the POOL in sub-blocks of 4
    // (filled up with the run's longest ordinary windows), the sub-blocks of the whole list are sorted by (columns per lane their query
    // sweeps -- whole panels + the narrow last one --, longest window)
    // and dealt four to a wavefront: a long window stretches three companions, not fifteen.  Everything else is STREAMED: the runs
    // in order of (columns per lane, ordinary window length), their windows pair by pair into wavefronts that are closed when they hold eight
    // pairs or meet a fifth query -- the windows of a wavefront take about the same number of steps, and a query with five windows
    // costs three lane groups, not two sub-blocks.  What is missing to a pair or a wavefront is filled with copies of the last
    // window (as the reference pads its SIMD batches, :1063-1067): they cost what the window costs and never survive (cut-off
    // INT_MAX).  The plan is the slot list of the whole call (the caller's index per slot) + columns per lane and longest window
    // per wavefront; chunks are ranges of wavefronts.
    constexpr uint64_t kSub = 4, kWave = 16;
    std::vector<uint32_t> & plan_slot = h->xb_slot, & wf_pan = h->xb_wfpan, & wf_maxs = h->xb_wfmaxs; // (wf_pan: columns per lane)
    std::vector<uint32_t> & sb_first = h->xb_sbfirst, & sb_key = h->xb_sbkey, & sb_order = h->xb_sborder, & sb_tmp = h->xb_sbtmp;
    uint64_t nwf = 0;
    hm.mark("classes+sort");
    // ONE strip geometry per call, the one that sweeps the fewest padded columns over the whole list (weighted by the
    // instructions a column costs at that width): every further geometry is a further pair of launches, and the backtrace of a
    // chunk with a few ten thousand survivors is bound by the latency of its longest walks (~1 ms), not by its work -- measured
    // on the ragged list of bench.py: (8,11) + (8,13) + (8,19) chosen per query 28.5 % padded cells but 27-31 ms, one geometry
    // 34.5 % / 39.9 % padded and 22.7-24.2 ms.
    int      mq_cfg   = 1;
    uint64_t mq_cells = 0; // sum q_len * s_len of the list (lx_last_extend_stats)
    // columns per lane a query sweeps over all its panels (its class in the plan's keys; what a wavefront executes is 8 of them per
    // step): whole panels, the last one with the narrowest strips that cover what is left (lx_device.h: narrow_code_for)
    auto mq_panels = [&](uint32_t lq) -> uint32_t
    {
        int const      C     = lx::trace_cfg_panel(mq_cfg) / 8;
        uint64_t const panel = (uint64_t)lx::trace_cfg_panel(mq_cfg);
        uint64_t const P     = std::max<uint64_t>(1, ((uint64_t)lq + panel - 1) / panel);
        int const      rem   = (int)((uint64_t)std::max<uint32_t>(lq, 1) - (P - 1) * panel);
        int const      code  = lx::narrow_code_for(C, 8, rem);
        return (uint32_t)std::min<uint64_t>(0xfff, (P - 1) * (uint64_t)C + (uint64_t)lx::narrow_strip_cols(C, code));
    };
    std::vector<uint64_t> & pool_at = h->xb_grp; // per run: first position of its pool part
    std::vector<uint32_t> & run_key = h->xb_runkey, & run_order = h->xb_runorder, & run_tmp = h->xb_runtmp;
    std::vector<uint8_t> &  sb_cnt = h->xb_sbcnt; // windows of a sub-block (the lowest of a run may have fewer than four)
    uint64_t nsb = 0, pool_wf = 0;
    // LSD radix sort by key: three passes of 10 bits (keys have 28)
    auto radix_sort = [](std::vector<uint32_t> & order, std::vector<uint32_t> & tmp, std::vector<uint32_t> const & key, uint64_t count)
    {
        for (int pass = 0; pass < 3; ++pass)
        {
            int const shift = 10 * pass;
            uint32_t  hist[1025] = {0};
            for (uint64_t o = 0; o < count; ++o)
                ++hist[((key[order[o]] >> shift) & 1023u) + 1];
            bool one_bucket = false;
            for (int bk = 0; bk < 1024; ++bk)
            {
                one_bucket = one_bucket || hist[bk + 1] == count;
                hist[bk + 1] += hist[bk];
            }
            if (one_bucket)
                continue;
            for (uint64_t o = 0; o < count; ++o)
                tmp[hist[(key[order[o]] >> shift) & 1023u]++] = order[o];
            order.swap(tmp);
        }
    };
    if (use_mq && preplanned)
    {
        mq_cfg         = ri->mq_cfg;
        h->mq_cfg_call = mq_cfg;
        mq_cells       = ri->cells;
        nwf = pool_wf  = ri->nwf;
        wf_pan.assign(ri->wf_pan, ri->wf_pan + nwf);
        wf_maxs.assign(ri->wf_maxs, ri->wf_maxs + nwf);
    }
    else if (use_mq)
    {
        if (starts.empty())
            run_starts(starts);
        uint64_t const        nruns = starts.size() - 1;
        {
            int const cand[3] = {1, 3, 5};
            std::vector<double> tc(3 * (size_t)nthreads, 0.0);
            parallel_ranges(nruns, nthreads,
                            [&](unsigned t, uint64_t rlo, uint64_t rhi)
                            {
                                double c[3] = {0, 0, 0};
                                for (uint64_t r = rlo; r < rhi; ++r)
                                {
                                    uint64_t const lq = ext[idx[starts[r]]].q_len, nw = use_solo ? starts[r + 1] - starts[r] : (starts[r + 1] - starts[r] + 1) / 2 * 2;
                                    for (int k = 0; k < 3; ++k)
                                    {
                                        uint64_t const panel = (uint64_t)lx::trace_cfg_panel(cand[k]), P = std::max<uint64_t>(1, (lq + panel - 1) / panel);
                                        int const      Cc = (int)panel / 8, rem = (int)(std::max<uint64_t>(lq, 1) - (P - 1) * panel);
                                        int const      code = lx::narrow_code_for(Cc, 8, rem);
                                        // (a step costs 3.75 instructions per column and 12 besides, whatever the strip width)
                                        c[k] += (double)nw * ((double)(P - 1) * (3.75 * Cc + 12.0) + 3.75 * lx::narrow_strip_cols(Cc, code) + 12.0);
                                    }
                                }
                                for (int k = 0; k < 3; ++k)
                                    tc[3 * t + k] = c[k];
                            });
            double best = 1e300;
            for (int k = 0; k < 3; ++k)
            {
                double c = 0;
                for (unsigned t = 0; t < nthreads; ++t)
                    c += tc[3 * t + k];
                // (narrower strips: more panels -- carries, profile builds -- and more tiles per walk in the backtrace; measured on
                // the ragged list of bench.py: 22.2 / 21.5 ms with 13 / 11 columns against 19.7 ms with 19, at 8 / 10 % fewer cells)
                c *= cand[k] == 1 ? 1.0 : 1.15;
                if (c < best)
                {
                    best   = c;
                    mq_cfg = cand[k];
                }
            }
        }
        h->mq_cfg_call = mq_cfg;
        if (use_solo)
        {
            // keys: most columns per lane first, longest window first inside a width (28 bits); a stable LSD radix sort over the
            // host threads (10 bits per pass: per-thread counts of a contiguous share, one scan, one scatter)
            std::vector<uint32_t> & key = h->xb_sbkey, & ord = h->xb_sborder, & tmp = h->xb_sbtmp;
            key.resize(live);
            ord.resize(live);
            tmp.resize(live);
            std::vector<uint64_t> tcells_plan(nthreads, 0);
            std::vector<uint32_t> tor(nthreads, 0), tand(nthreads, ~0u);
            parallel_ranges(live, nthreads,
                            [&](unsigned t, uint64_t lo, uint64_t hi)
                            {
                                uint64_t cells_t = 0;
                                uint32_t o = 0, a = ~0u;
                                for (uint64_t k = lo; k < hi; ++k)
                                {
                                    lx_extension const & x = ext[idx[k]];
                                    cells_t += (uint64_t)x.q_len * x.s_len;
                                    uint32_t const kk = ((0xfffu - mq_panels(x.q_len)) << 16) | (0xffffu - std::min<uint32_t>(x.s_len, 0xffffu));
                                    key[k] = kk;
                                    ord[k] = (uint32_t)k;
                                    o |= kk;
                                    a &= kk;
                                }
                                tcells_plan[t] = cells_t;
                                tor[t]         = o;
                                tand[t]        = a;
                            });
            mq_cells = 0;
            uint32_t varying = 0; // bits in which the keys differ
            {
                uint32_t o = 0, a = ~0u;
                for (unsigned t = 0; t < nthreads; ++t)
                {
                    mq_cells += tcells_plan[t];
                    o |= tor[t];
                    a &= tand[t];
                }
                varying = o & ~a;
            }
            std::vector<uint32_t> cnt((size_t)nthreads * 1024);
            for (int shift = 0; shift < 30; shift += 10)
            {
                if (!((varying >> shift) & 1023u))
                    continue;
                parallel_ranges(live, nthreads,
                                [&](unsigned t, uint64_t lo, uint64_t hi)
                                {
                                    uint32_t * const c = cnt.data() + (size_t)t * 1024;
                                    std::fill(c, c + 1024, 0u);
                                    for (uint64_t k = lo; k < hi; ++k)
                                        ++c[(key[ord[k]] >> shift) & 1023u];
                                });
                uint32_t at = 0;
                for (int b = 0; b < 1024; ++b)
                    for (unsigned t = 0; t < nthreads; ++t)
                    {
                        uint32_t const c          = cnt[(size_t)t * 1024 + b];
                        cnt[(size_t)t * 1024 + b] = at;
                        at += c;
                    }
                parallel_ranges(live, nthreads,
                                [&](unsigned t, uint64_t lo, uint64_t hi)
                                {
                                    uint32_t * const c = cnt.data() + (size_t)t * 1024;
                                    for (uint64_t k = lo; k < hi; ++k)
                                        tmp[c[(key[ord[k]] >> shift) & 1023u]++] = ord[k];
                                });
                ord.swap(tmp);
            }
            nwf     = (live + kWave - 1) / kWave;
            pool_wf = nwf;
            if (plan_slot.size() < nwf * kWave)
                plan_slot.resize(nwf * kWave);
            if (wf_pan.size() < nwf)
            {
                wf_pan.resize(nwf);
                wf_maxs.resize(nwf);
            }
            parallel_ranges(nwf, nthreads,
                            [&](unsigned, uint64_t wlo, uint64_t whi)
                            {
                                for (uint64_t w = wlo; w < whi; ++w)
                                {
                                    uint32_t pan = 0, maxs = 0;
                                    for (uint64_t j = 0; j < kWave; ++j)
                                    {
                                        uint64_t const o = w * kWave + j;
                                        uint32_t const i = idx[ord[std::min(o, live - 1)]]; // (the last wavefront repeats the last window as filler)
                                        plan_slot[o]     = i | (o < live ? 0u : 0x80000000u);
                                        pan              = std::max(pan, mq_panels(ext[i].q_len));
                                        maxs             = std::max(maxs, ext[i].s_len);
                                    }
                                    wf_pan[w]  = pan;
                                    wf_maxs[w] = maxs;
                                }
                            });
            hm.mark("solo plan");
        }
        else
        {
        // (1) per run: where its pool begins (the long windows + what fills their last sub-block up), its sub-blocks, its cells
        pool_at.assign(nruns, 0);
        std::vector<uint64_t> sb_off(nruns + 1, 0), tcells_plan(nthreads, 0);
        run_key.resize(nruns);
        parallel_ranges(nruns, nthreads,
                        [&](unsigned t, uint64_t rlo, uint64_t rhi)
                        {
                            uint64_t cells_t = 0;
                            for (uint64_t r = rlo; r < rhi; ++r)
                            {
                                uint64_t const a = starts[r], b = starts[r + 1];
                                uint64_t const med = ext[idx[a + (b - a - 1) / 2]].s_len, thr = med + std::max<uint64_t>(8, med / 8);
                                uint64_t       cut = b; // first long window
                                while (cut > a && ext[idx[cut - 1]].s_len > thr)
                                    --cut;
                                for (uint64_t k = a; k < b; ++k)
                                    cells_t += (uint64_t)ext[idx[k]].q_len * ext[idx[k]].s_len;
                                uint64_t const nsb_r = (b - cut + kSub - 1) / kSub;
                                pool_at[r]           = nsb_r * kSub >= b - a ? a : b - nsb_r * kSub;
                                sb_off[r + 1]        = nsb_r;
                                // the streamed part's place in the packing order: most panels first, longest windows first
                                if (pool_at[r] != a)
                                    run_key[r] = ((0xfffu - mq_panels(ext[idx[a]].q_len)) << 16) |
                                                 (0xffffu - std::min<uint32_t>(ext[idx[pool_at[r] - 1]].s_len, 0xffffu));
                            }
                            tcells_plan[t] = cells_t;
                        });
        mq_cells = 0;
        for (uint64_t c : tcells_plan)
            mq_cells += c;
        for (uint64_t r = 0; r < nruns; ++r)
            sb_off[r + 1] += sb_off[r];
        nsb = sb_off[nruns];
        sb_first.resize(nsb);
        sb_key.resize(nsb);
        sb_order.resize(nsb);
        sb_tmp.resize(nsb);
        sb_cnt.resize(nsb);
        // the pool's sub-blocks: sub-block j of a run (0 = its longest windows) = positions [max(pool, b - 4 (j + 1)), b - 4 j)
        parallel_ranges(nruns, nthreads,
                        [&](unsigned, uint64_t rlo, uint64_t rhi)
                        {
                            for (uint64_t r = rlo; r < rhi; ++r)
                            {
                                uint32_t const cls = 0xfffu - mq_panels(ext[idx[starts[r]]].q_len);
                                uint64_t       e   = starts[r + 1];
                                for (uint64_t o = sb_off[r]; o < sb_off[r + 1]; ++o)
                                {
                                    uint64_t const first = std::max<uint64_t>(pool_at[r], e >= kSub ? e - kSub : 0);
                                    sb_first[o] = (uint32_t)first;
                                    sb_cnt[o]   = (uint8_t)(e - first);
                                    // (most panels first, longest first inside a panel count: the wavefronts that run longest
                                    // start first, the tail of the launch is made of short ones)
                                    sb_key[o] = (cls << 16) | (0xffffu - std::min<uint32_t>(ext[idx[e - 1]].s_len, 0xffffu));
                                    e         = first;
                                }
                            }
                        });
        hm.mark("sub-blocks");
        for (uint64_t o = 0; o < nsb; ++o)
            sb_order[o] = (uint32_t)o;
        radix_sort(sb_order, sb_tmp, sb_key, nsb);
        // the pool's part of the plan: four sub-blocks per wavefront.  It goes to the GPU first (the chunk loop below) -- the longest
        // windows of the list -- and the streamed part is planned beside its kernels (plan_stream)
        pool_wf = (nsb + 3) / 4;
        nwf     = pool_wf;
        if (plan_slot.size() < nwf * kWave)
            plan_slot.resize(nwf * kWave);
        if (wf_pan.size() < nwf)
        {
            wf_pan.resize(nwf);
            wf_maxs.resize(nwf);
        }
        // The wavefronts are made of neighbours in that order (a wavefront runs as many columns per lane as its widest query has and as
        // many steps as its longest window) and LAUNCHED longest first: what a wavefront executes is columns x steps, and the blocks of a
        // launch are dealt to the chip's wavefront slots in index order -- with two or three wavefronts per slot (a list of long queries)
        // the launch is as long as its unluckiest slot, which longest-first keeps at the longest wavefront itself.
        std::vector<uint32_t> & pool_pan = h->xb_pool_pan, & pool_maxs = h->xb_pool_maxs, & pool_place = h->xb_pool_place;
        pool_pan.resize(pool_wf);
        pool_maxs.resize(pool_wf);
        pool_place.resize(pool_wf);
        parallel_ranges(pool_wf, nthreads,
                        [&](unsigned, uint64_t wlo, uint64_t whi)
                        {
                            for (uint64_t w = wlo; w < whi; ++w)
                            {
                                uint32_t pan = 0, maxs = 0;
                                for (uint64_t o = 4 * w; o < 4 * w + 4; ++o)
                                {
                                    uint32_t const sb    = sb_order[std::min(o, nsb - 1)];
                                    uint64_t const first = sb_first[sb], cnt = sb_cnt[sb];
                                    pan = std::max(pan, 0xfffu - (sb_key[sb] >> 16));
                                    for (uint64_t j = 0; j < cnt; ++j)
                                        maxs = std::max(maxs, ext[idx[first + j]].s_len);
                                }
                                pool_pan[w]  = pan;
                                pool_maxs[w] = maxs;
                            }
                        });
        {
            std::vector<uint32_t> & by_len = h->xb_pool_order;
            by_len.resize(pool_wf);
            for (uint64_t w = 0; w < pool_wf; ++w)
                by_len[w] = (uint32_t)w;
            {
                std::vector<uint32_t> & len_key = h->xb_pool_key, & len_tmp = h->xb_pool_tmp;
                len_key.resize(pool_wf);
                len_tmp.resize(pool_wf);
                for (uint64_t w = 0; w < pool_wf; ++w)
                    len_key[w] = 0x3fffffffu - (uint32_t)std::min<uint64_t>((uint64_t)pool_pan[w] * (pool_maxs[w] + 7), 0x3fffffffu);
                radix_sort(by_len, len_tmp, len_key, pool_wf);
            }
            for (uint64_t k = 0; k < pool_wf; ++k)
                pool_place[by_len[k]] = (uint32_t)k;
        }
        parallel_ranges(pool_wf, nthreads,
                        [&](unsigned, uint64_t wlo, uint64_t whi)
                        {
                            for (uint64_t w = wlo; w < whi; ++w)
                            {
                                uint64_t const at = pool_place[w]; // (its place in the launch)
                                for (uint64_t o = 4 * w; o < 4 * w + 4; ++o)
                                {
                                    // (a wavefront that the pool cannot fill repeats its last sub-block as fillers)
                                    bool const     real  = o < nsb;
                                    uint32_t const sb    = sb_order[std::min(o, nsb - 1)];
                                    uint64_t const first = sb_first[sb], cnt = sb_cnt[sb];
                                    // (the longest window first, like the streamed pairs; the last one is repeated as filler)
                                    for (uint64_t j = 0; j < kSub; ++j)
                                        plan_slot[at * kWave + (o - 4 * w) * kSub + j] =
                                          idx[first + cnt - 1 - std::min(j, cnt - 1)] | ((real && j < cnt) ? 0u : 0x80000000u);
                                }
                                wf_pan[at]  = pool_pan[w];
                                wf_maxs[at] = pool_maxs[w];
                            }
                        });
        hm.mark("pool");
        } // (!use_solo)
    }
    // (2) the streamed part: runs in order of (panels, ordinary window length), most panels and longest first
    auto plan_stream = [&]()
    {
        uint64_t const nruns = starts.size() - 1;
        run_order.resize(nruns);
        run_tmp.resize(nruns);
        uint64_t nstream_runs = 0;
        for (uint64_t r = 0; r < nruns; ++r)
            if (pool_at[r] != starts[r]) // (else the whole run stands in the pool)
                run_order[nstream_runs++] = (uint32_t)r;
        radix_sort(run_order, run_tmp, run_key, nstream_runs);
        // every thread packs a contiguous share of the sorted runs into wavefronts of its own (a share starts a new wavefront):
        // once to count them, once -- the offsets known -- to write the plan
        std::vector<uint64_t> wf_at(nthreads + 1, pool_wf);
        auto pack = [&](uint64_t lo, uint64_t hi, uint32_t * out_slot, uint32_t * out_pan, uint32_t * out_maxs) -> uint64_t
        {
            uint32_t wq[4] = {0, 0, 0, 0}; // runs of the open wavefront
            uint32_t nq = 0, npairs = 0, pan = 0, maxs = 0, last = 0;
            uint64_t done = 0;             // wavefronts closed
            auto close = [&]()
            {
                if (npairs == 0)
                    return;
                if (out_slot)
                {
                    for (uint32_t k = 2 * npairs; k < kWave; ++k)
                        out_slot[done * kWave + k] = last | 0x80000000u;
                    out_pan[done]  = pan;
                    out_maxs[done] = maxs;
                }
                ++done;
                nq = npairs = pan = maxs = 0;
            };
            for (uint64_t x = lo; x < hi; ++x)
            {
                uint32_t const r  = run_order[x];
                uint32_t const rp = 0xfffu - (run_key[r] >> 16);
                // (longest first: the order descends over the runs, so it does inside one)
                for (uint64_t e = pool_at[r]; e > starts[r];)
                {
                    if (npairs == kWave / 2)
                        close();
                    bool known = false;
                    for (uint32_t k = 0; k < nq; ++k)
                        known = known || wq[k] == r;
                    if (!known)
                    {
                        if (nq == 4)
                            close();
                        wq[nq++] = r;
                    }
                    bool const two = e - 1 > starts[r];
                    if (out_slot)
                    {
                        uint32_t const i0 = idx[e - 1], i1 = two ? idx[e - 2] : (i0 | 0x80000000u);
                        out_slot[done * kWave + 2 * npairs]     = i0;
                        out_slot[done * kWave + 2 * npairs + 1] = i1;
                        last = i1;
                        maxs = std::max(maxs, ext[i0].s_len);
                        pan  = std::max(pan, rp);
                    }
                    ++npairs;
                    e -= two ? 2 : 1;
                }
            }
            close();
            return done;
        };
        parallel_ranges(nstream_runs, nthreads, [&](unsigned t, uint64_t lo, uint64_t hi) { wf_at[t + 1] = pack(lo, hi, nullptr, nullptr, nullptr); });
        for (unsigned t = 0; t < nthreads; ++t)
            wf_at[t + 1] += wf_at[t];
        nwf = wf_at[nthreads];
        if (plan_slot.size() < nwf * kWave)
            plan_slot.resize(nwf * kWave);
        if (wf_pan.size() < nwf)
        {
            wf_pan.resize(nwf);
            wf_maxs.resize(nwf);
        }
        parallel_ranges(nstream_runs, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        { (void)pack(lo, hi, plan_slot.data() + wf_at[t] * kWave, wf_pan.data() + wf_at[t], wf_maxs.data() + wf_at[t]); });
    };
    hm.mark("plan");

    // ---- the caller's option values come back on every exit; the streams are drained before anything is torn down
    h->phase_ev.clear();
    h->ev_pool_used      = 0;
    h->keep_phase_events = true;
    struct Guard
    {
        lx_handle * h;
        uint64_t    qlen, slen, run;
        ~Guard()
        {
            h->keep_phase_events = false;
            (void)hipStreamSynchronize(h->stream);
            (void)hipStreamSynchronize(h->stream2);
            (void)hipStreamSynchronize(h->stream3);
            h->mq_cfg_call   = 0;
            h->mq_wide_call  = false;
            h->opt_max_qlen  = qlen;
            h->opt_max_slen  = slen;
            h->opt_query_run = run;
        }
    } const guard{h, h->opt_max_qlen, h->opt_max_slen, h->opt_query_run};

    if (!ri)
    {
        if ((rc = ensure(h, h->d_q, q_bytes + kSlack)))
            return rc;
        if (q_bytes)
            LX_HIP(h, hipMemcpyAsync(h->d_q.ptr, q_res, q_bytes, hipMemcpyHostToDevice, h->stream));
    }
    void const * const d_qptr = ri ? ri->d_q : h->d_q.ptr;
    if (sref.upload)
        LX_HIP(h, hipMemcpyAsync(sref.dev, s_res, s_bytes, hipMemcpyHostToDevice, h->stream));
    // (measured in round 3 and removed again: two chunks' kernels side by side on two streams -- ragged list 28.5 against 24.4 ms,
    // headline batch 40.0 against 31.3 ms -- and a chunk's backtrace on a second stream beside the next chunk's sweep -- no gain on
    // the ragged list, 34.9 against 29.0 ms on the headline: kernels side by side cost more than their tails and latencies save)

    uint64_t const chunk_target = h->opt_extend_chunk ? std::max<uint64_t>(h->opt_extend_chunk, 1024) : lxi::kExtendChunk;
    h->ext_bytes.clear();
    uint64_t ops_total = 0; // bytes handed out in h->ext_bytes so far
    double   t_prep = 0, t_issue = 0, t_wait = 0, t_unpack = 0, t_u1 = 0, t_u2 = 0; // LX_HOST_TIMING: where the host's time goes
    h->xb_stats[0] = live;
    h->xb_stats[1] = h->xb_stats[2] = h->xb_stats[3] = 0; // slots, cells, cells the wavefronts execute
    auto     now    = []() { return std::chrono::steady_clock::now(); };
    auto     ms     = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b)
    { return std::chrono::duration<double, std::milli>(b - a).count(); };
    XbPrep   prep[2];
    bool     in_flight[2] = {false, false};

    // ---- device side of a chunk whose padded slots stand in lane L's pinned staging: uploads and kernels queued
    auto launch_chunk = [&](int L, uint64_t slots, uint64_t max_q, uint64_t max_s, uint64_t kRun) -> int
    {
        auto const          t1 = now();
        lx_handle::XbLane & ln = h->xb[L];
        XbPrep &            pr = prep[L];
        lx_extension * const slot_ext = static_cast<lx_extension *>(ln.p_ext.ptr);
        int32_t * const      slot_min = static_cast<int32_t *>(ln.p_min.ptr);
        int rc2;
        // device side of the lane
        uint64_t const stride = (max_q + max_s + 3) & ~3ull; // one ops slot per position of the survivor list
        if ((rc2 = ensure(h, ln.d_ext, slots * sizeof(lx_extension))) || (rc2 = ensure(h, ln.d_min, slots * sizeof(int32_t))) ||
            (rc2 = ensure(h, ln.d_score, slots * sizeof(int32_t))) || (rc2 = ensure(h, ln.d_hsp, pr.cap_sel * sizeof(lx_hsp))) ||
            (rc2 = ensure(h, ln.d_ops, pr.cap_sel * stride + 16)) || (rc2 = ensure(h, ln.d_rle, pr.cap_sel * stride + 16)) ||
            (rc2 = ensure(h, ln.d_src, pr.cap_sel * sizeof(uint32_t))) || (rc2 = ensure(h, ln.d_len, pr.cap_sel * sizeof(uint32_t))) ||
            (rc2 = ensure(h, ln.d_cnt, 4 * sizeof(uint64_t))) ||
            (rc2 = ensure_pinned(h, ln.p_score, slots * sizeof(int32_t))) || (rc2 = ensure_pinned(h, ln.p_cnt, 4 * sizeof(uint64_t))))
            return rc2;
        LX_HIP(h, hipMemcpyAsync(ln.d_ext.ptr, slot_ext, slots * sizeof(lx_extension), hipMemcpyHostToDevice, h->stream3));
        LX_HIP(h, hipMemcpyAsync(ln.d_min.ptr, slot_min, slots * sizeof(int32_t), hipMemcpyHostToDevice, h->stream3));
        LX_HIP(h, hipEventRecord(ln.ev_up, h->stream3));
        hipStream_t const ks = h->stream; // (all chunks' kernels in one stream: side by side they were measured slower, see above)
        LX_HIP(h, hipStreamWaitEvent(ks, ln.ev_up, 0));
        h->opt_max_qlen  = max_q;
        h->opt_max_slen  = max_s;
        h->opt_query_run = kRun;
        uint64_t * const d_cnt = static_cast<uint64_t *>(ln.d_cnt.ptr);
        FusedExtra       fx;
        fx.ops_stride = stride;
        fx.d_rle      = static_cast<uint8_t *>(ln.d_rle.ptr);
        fx.d_rle_top  = reinterpret_cast<unsigned long long *>(d_cnt + 2);
        fx.rle_cap    = pr.cap_sel * stride;
        fx.d_src_out  = static_cast<uint32_t *>(ln.d_src.ptr);
        fx.d_rle_len  = static_cast<uint32_t *>(ln.d_len.ptr);
        hipStream_t const ke = ks;
        if ((rc2 = fused_impl(h, slot, d_qptr, sref.dev, ln.d_ext.ptr, slots, ln.d_min.ptr, 0, ln.d_score.ptr, ln.d_hsp.ptr, ln.d_ops.ptr,
                              nullptr, d_cnt, ks, 3, true, &fx)))
            return rc2;
        // the device's error word of THIS chunk, saved in stream order (the next chunk's prepare_workspace clears it): it comes
        // back with the counts and is checked in collect()
        LX_HIP(h, hipMemcpyAsync(d_cnt + 3, h->d_ws_top, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, ke));
        LX_HIP(h, hipEventRecord(ln.ev_k, ke));
        // what has a size the host knows goes back at once; records and codes follow when the counts have arrived
        LX_HIP(h, hipStreamWaitEvent(h->stream2, ln.ev_k, 0));
        LX_HIP(h, hipMemcpyAsync(ln.p_cnt.ptr, d_cnt, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream2));
        LX_HIP(h, hipMemcpyAsync(ln.p_score.ptr, ln.d_score.ptr, slots * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream2));
        LX_HIP(h, hipEventRecord(ln.ev_cnt, h->stream2));
        in_flight[L] = true;
        t_issue += ms(t1, now());
        return LX_OK;
    };

    // ---- chunk k0 .. k1 of the ordered list -> padded slots in lane L's pinned staging -> uploads and kernels queued
    auto enqueue = [&](int L, uint64_t k0, uint64_t k1) -> int
    {
        auto const          t0 = now();
        lx_handle::XbLane & ln = h->xb[L];
        XbPrep &            pr = prep[L];
        pr.k0 = k0;
        pr.k1 = k1;
        // runs of one query slice; padded to 16 slots (one query per wavefront of the 8-lane packed geometry) or, when the
        // queries have few windows each, to 8 (one query per half wavefront, or the 16-lane geometries) -- whichever is less work
        std::vector<uint64_t> & grp = h->xb_grp; // (first position, first slot) of every run + a sentinel
        grp.clear();
        uint64_t slots16 = 0, slots8 = 0, max_q = 1, max_s = 1;
        for (uint64_t k = k0; k < k1;)
        {
            uint64_t kk = k + 1;
            while (kk < k1 && !newrun[kk])
                ++kk;
            grp.push_back(k);
            grp.push_back(0);
            slots16 += (kk - k + 15) / 16 * 16;
            slots8 += (kk - k + 7) / 8 * 8;
            max_q = std::max<uint64_t>(max_q, ext[idx[k]].q_len);
            k     = kk;
        }
        grp.push_back(k1);
        grp.push_back(0);
        uint64_t const ngroups = grp.size() / 2 - 1;
        // (measured on the ragged list of bench.py: a slot of a run of 8 costs ~1.1 x one of a run of 16 while the query fits a
        // panel; wider queries run (8,19) panels with compact codes at 16 and (16,13) panels with int16 pairs at 8 -- the
        // padded columns and 0.0108 against 0.0156 ms per column, ckpt_cfg_for in lx_api.cpp, decide)
        double const cost16 = max_q > 208 ? (double)slots16 * (double)((max_q + 151) / 152 * 152) * 1.08 : (double)slots16 * 8.0;
        double const cost8  = max_q > 208 ? (double)slots8 * (double)((max_q + 207) / 208 * 208) * 1.56 : (double)slots8 * 9.0;
        uint64_t const kRun = cost8 < cost16 ? 8 : 16;
        uint64_t       slots   = 0;
        for (uint64_t g = 0; g <= ngroups; ++g)
        {
            grp[2 * g + 1] = slots;
            if (g < ngroups)
                slots += (grp[2 * g + 2] - grp[2 * g] + kRun - 1) / kRun * kRun;
        }
        pr.slots   = slots;
        pr.cap_sel = (slots + slots / kRun * 3 + 7) / 8 * 8 + 8;
        pr.slot_src.resize(slots);
        int rc2;
        if ((rc2 = ensure_pinned(h, ln.p_ext, slots * sizeof(lx_extension))) || (rc2 = ensure_pinned(h, ln.p_min, slots * sizeof(int32_t))))
            return rc2;
        lx_extension * const slot_ext = static_cast<lx_extension *>(ln.p_ext.ptr);
        int32_t * const      slot_min = static_cast<int32_t *>(ln.p_min.ptr);
        uint32_t * const     slot_src = pr.slot_src.data();
        std::vector<uint64_t> tmax(nthreads, 1), tcells(nthreads, 0), tpad(nthreads, 0);
        // (what the wavefronts will execute: every block of kRun slots runs all columns of its panels for as many steps as
        // its longest window has rows)
        uint64_t const panel = max_q <= 104 ? 104 : max_q <= 152 ? 152 : max_q <= 200 ? 200 : max_q <= 208 ? 208 : 152, lanes = panel == 208 ? 16 : 8;
        parallel_ranges(ngroups, nthreads,
                        [&](unsigned t, uint64_t glo, uint64_t ghi)
                        {
                            uint64_t ms = 1, cells = 0, padded = 0; // (locals: the per-thread slots share cache lines)
                            for (uint64_t g = glo; g < ghi; ++g)
                            {
                                uint64_t const a = grp[2 * g], b = grp[2 * g + 2], o1 = grp[2 * g + 3];
                                uint64_t       o = grp[2 * g + 1];
                                uint64_t const cols = (ext[idx[a]].q_len + panel - 1) / panel * panel;
                                for (uint64_t j0 = a; j0 < b; j0 += kRun)
                                {
                                    uint64_t bmax = 0;
                                    for (uint64_t j = j0; j < std::min(b, j0 + kRun); ++j)
                                    {
                                        bmax = std::max<uint64_t>(bmax, ext[idx[j]].s_len);
                                        cells += (uint64_t)ext[idx[j]].q_len * ext[idx[j]].s_len;
                                    }
                                    padded += kRun * cols * (bmax + lanes - 1);
                                }
                                for (uint64_t j = a; j < b; ++j, ++o)
                                {
                                    slot_ext[o] = ext[idx[j]];
                                    slot_src[o] = idx[j];
                                    slot_min[o] = min_score ? min_score[idx[j]] : min_score_all;
                                    ms          = std::max<uint64_t>(ms, ext[idx[j]].s_len);
                                }
                                lx_extension dummy = ext[idx[a]];
                                dummy.s_len        = 0;
                                for (; o < o1; ++o)
                                {
                                    slot_ext[o] = dummy;
                                    slot_src[o] = 0xffffffffu;
                                    slot_min[o] = 0x7fffffff; // never survives
                                }
                            }
                            tmax[t]   = std::max(tmax[t], ms);
                            tcells[t] = cells;
                            tpad[t]   = padded;
                        });
        for (uint64_t v : tmax)
            max_s = std::max(max_s, v);
        h->xb_stats[1] += slots;
        for (unsigned t = 0; t < nthreads; ++t)
        {
            h->xb_stats[2] += tcells[t];
            h->xb_stats[3] += tpad[t];
        }
        t_prep += ms(t0, now());
        return launch_chunk(L, slots, max_q, max_s, kRun);
    };

    // ---- wavefronts w0 .. w1 of the plan -> the slots' caller indices in lane L's pinned staging -> upload, gather of the slot
    // records on the device, kernels, scatter of the scores into caller order.  The plan's order is a permutation of the caller's
    // list (+ fillers): the host only touches 4 bytes per slot here (the 24-byte records and their cut-offs are gathered from the
    // device copy of the list at HBM speed, not by cache misses of a few host threads).
    auto enqueue_mq = [&](int L, uint64_t w0, uint64_t w1, bool force_wide = false) -> int
    {
        auto const          t0 = now();
        lx_handle::XbLane & ln = h->xb[L];
        XbPrep &            pr = prep[L];
        pr.k0 = w0;
        pr.k1 = w1;
        uint64_t const slots = (w1 - w0) * kWave;
        pr.slots   = slots;
        pr.cap_sel = (slots + 7) / 8 * 8 + 8;
        int rc2;
        if (!preplanned && (rc2 = ensure_pinned(h, ln.p_orig, slots * sizeof(uint32_t))))
            return rc2;
        uint32_t * const slot_orig = static_cast<uint32_t *>(ln.p_orig.ptr);
        uint64_t const   panel     = (uint64_t)lx::trace_cfg_panel(mq_cfg);
        std::vector<uint64_t> tpad(nthreads, 0), tmaxs(nthreads, 1), tpan(nthreads, 1);
        // (what the wavefronts execute: every one sweeps as many panels as its widest query needs, each for as many steps as its
        // longest window has rows)
        parallel_ranges(w1 - w0, nthreads,
                        [&](unsigned t, uint64_t wlo, uint64_t whi)
                        {
                            uint64_t padded = 0, smax = 1, pmax = 1; // (locals: the per-thread slots share cache lines)
                            if (!preplanned)
                                std::memcpy(slot_orig + wlo * kWave, plan_slot.data() + (w0 + wlo) * kWave, (whi - wlo) * kWave * sizeof(uint32_t));
                            for (uint64_t w = w0 + wlo; w < w0 + whi; ++w)
                            {
                                padded += kWave * ((uint64_t)wf_pan[w] * 8) * ((uint64_t)wf_maxs[w] + 7);
                                smax = std::max<uint64_t>(smax, wf_maxs[w]);
                                pmax = std::max<uint64_t>(pmax, wf_pan[w]);
                            }
                            tpad[t]  = padded;
                            tmaxs[t] = smax;
                            tpan[t]  = pmax;
                        });
        // the promises of the chunk: its widest query (as a panel count) and its longest window
        uint64_t max_s = 1, max_pan = 1;
        pr.exec_cells = 0;
        for (unsigned t = 0; t < nthreads; ++t)
        {
            pr.exec_cells += tpad[t];
            h->xb_stats[3] += tpad[t];
            max_s   = std::max(max_s, tmaxs[t]);
            max_pan = std::max(max_pan, tpan[t]);
        }
        h->xb_stats[1] += slots;
        pr.max_s   = max_s;
        pr.max_pan = max_pan;
        uint64_t const max_q = (max_pan + panel / 8 - 1) / (panel / 8) * panel; // (whole panels: the slots have one part per panel)
        t_prep += ms(t0, now());

        auto const t1 = now();
        uint64_t const stride = (max_q + max_s + 3) & ~3ull; // one ops slot per position of the survivor list
        if ((!preplanned && (rc2 = ensure(h, ln.d_orig, slots * sizeof(uint32_t)))) || (rc2 = ensure(h, ln.d_ext, slots * sizeof(lx_extension))) ||
            (rc2 = ensure(h, ln.d_min, slots * sizeof(int32_t))) || (rc2 = ensure(h, ln.d_score, slots * sizeof(int32_t))) ||
            (rc2 = ensure(h, ln.d_hsp, pr.cap_sel * sizeof(lx_hsp))) || (rc2 = ensure(h, ln.d_ops, pr.cap_sel * stride + 16)) ||
            (rc2 = ensure(h, ln.d_rle, pr.cap_sel * stride + 16)) || (rc2 = ensure(h, ln.d_src, pr.cap_sel * sizeof(uint32_t))) ||
            (rc2 = ensure(h, ln.d_len, pr.cap_sel * sizeof(uint32_t))) || (rc2 = ensure(h, ln.d_cnt, 5 * sizeof(uint64_t))) ||
            (rc2 = ensure_pinned(h, ln.p_cnt, 5 * sizeof(uint64_t))))
            return rc2;
        // (a device plan: the chunk's slots are a piece of it)
        uint32_t const * const d_orig = preplanned ? ri->d_plan + w0 * kWave : static_cast<uint32_t const *>(ln.d_orig.ptr);
        if (!preplanned)
        {
            LX_HIP(h, hipMemcpyAsync(ln.d_orig.ptr, slot_orig, slots * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream3));
            LX_HIP(h, hipEventRecord(ln.ev_up, h->stream3));
            LX_HIP(h, hipStreamWaitEvent(h->stream, ln.ev_up, 0));
        }
        LX_HIP(h, lx::launch_slot_gather(static_cast<lx::Extension const *>(ri ? ri->d_ext_all : h->d_ext_all.ptr),
                                         (min_score || (ri && ri->d_min_all)) ? static_cast<int32_t const *>(ri ? ri->d_min_all : h->d_min_all.ptr) : nullptr,
                                         min_score_all, d_orig, slots, static_cast<lx::Extension *>(ln.d_ext.ptr),
                                         static_cast<int32_t *>(ln.d_min.ptr), h->stream));
        h->opt_max_qlen  = max_q;
        h->opt_max_slen  = max_s;
        h->opt_query_run = use_solo ? 1 : 2; // (the solo packing: no promise; the free packing: pairs of one query, at most four queries per wavefront)
        // Compact codes hold scores up to 2046; a window beyond them is redone by the int32 launch, one profile per pair, at a tenth
        // of the sweep's speed.  Where the last chunks had more than a few such windows (long queries with strong hits: a 600-residue
        // query against its homologue scores ~3 000) the sweep writes int16 pairs itself (lx_sweep_mq.hip: WIDE) -- twice the
        // checkpoint bytes, no second launch; it goes back to the codes when fewer than 1 % of a chunk's windows need more.
        // (force_wide: a chunk that runs again because its overflow area filled up -- said by the caller, not inferred from the fraction
        // the OTHER lane's collect may have overwritten meanwhile)
        h->mq_wide_call = mq_cfg == 1 && !lx::dev_aids().mq_no_wide && (force_wide || (h->mq_wide_call ? h->mq_decl_frac > 0.01 : h->mq_decl_frac > 0.03));
        pr.wide         = h->mq_wide_call;
        h->mq_tab       = lx_handle::MqTab{};
        {
            // the chunk's slots by wavefront (lx::WfSlots): every wavefront's sixteen laid out for ITS longest window and widest query
            uint64_t const nw = w1 - w0, pc = panel / 8;
            if ((rc2 = ensure_pinned(h, ln.p_wft, nw * sizeof(lx::WfSlots))) || (rc2 = ensure(h, ln.d_wft, nw * sizeof(lx::WfSlots))))
                return rc2;
            lx::WfSlots * const tab = static_cast<lx::WfSlots *>(ln.p_wft.ptr);
            uint64_t            off = 0, slot_dw = 0;
            uint32_t            last_steps = 0; // (a sorted plan repeats its step counts: the layout's slot size is asked for once per run of them)
            for (uint64_t w = 0; w < nw; ++w)
            {
                uint32_t const steps  = (uint32_t)(((uint64_t)wf_maxs[w0 + w] + 8 - 1 + 15) & ~15ull);
                uint32_t const panels = (uint32_t)std::max<uint64_t>(1, ((uint64_t)wf_pan[w0 + w] + pc - 1) / pc);
                if (steps != last_steps)
                {
                    slot_dw    = pr.wide ? lx::ckpt_slot_dwords(mq_cfg, steps) : lx::ckpt16_slot_dwords(mq_cfg, steps);
                    last_steps = steps;
                }
                tab[w] = lx::WfSlots{off, steps, panels};
                off += kWave * (uint64_t)panels * slot_dw;
            }
            LX_HIP(h, hipMemcpyAsync(ln.d_wft.ptr, tab, nw * sizeof(lx::WfSlots), hipMemcpyHostToDevice, h->stream3));
            LX_HIP(h, hipEventRecord(ln.ev_up, h->stream3));
            LX_HIP(h, hipStreamWaitEvent(h->stream, ln.ev_up, 0));
            h->mq_tab.dev     = ln.d_wft.ptr;
            h->mq_tab.n0      = slots;
            h->mq_tab.dw0     = off;
            h->mq_tab.ovf_cap = std::min<uint64_t>(slots, slots / 8 + 64); // (what the chunk's budget reserved: an eighth of its slots)
        }
        uint64_t * const d_cnt = static_cast<uint64_t *>(ln.d_cnt.ptr);
        FusedExtra       fx;
        fx.ops_stride = stride;
        fx.d_rle      = static_cast<uint8_t *>(ln.d_rle.ptr);
        fx.d_rle_top  = reinterpret_cast<unsigned long long *>(d_cnt + 2);
        fx.rle_cap    = pr.cap_sel * stride;
        fx.d_src_out  = static_cast<uint32_t *>(ln.d_src.ptr);
        fx.d_rle_len  = static_cast<uint32_t *>(ln.d_len.ptr);
        if ((rc2 = fused_impl(h, slot, d_qptr, sref.dev, ln.d_ext.ptr, slots, ln.d_min.ptr, 0, ln.d_score.ptr, ln.d_hsp.ptr, ln.d_ops.ptr, nullptr,
                              d_cnt, h->stream, 3, true, &fx)))
        {
            h->mq_tab = lx_handle::MqTab{};
            return rc2;
        }
        h->mq_tab = lx_handle::MqTab{};
        LX_HIP(h, lx::launch_slot_scatter(d_orig, slots, static_cast<int32_t const *>(ln.d_score.ptr),
                                          static_cast<int32_t *>(h->d_score_all.ptr), static_cast<uint32_t *>(ln.d_src.ptr), d_cnt, pr.cap_sel, h->stream));
        LX_HIP(h, hipMemcpyAsync(d_cnt + 3, h->d_ws_top, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream)); // this chunk's error word
        LX_HIP(h, hipMemcpyAsync(d_cnt + 4, h->d_ws_top + 6, sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream)); // ... its windows beyond the compact codes
        if (by_range && (rc2 = ri->chunk_records->enqueue(pr.range, ln.d_hsp.ptr, ln.d_src.ptr, d_cnt, pr.cap_sel)))
            return rc2; // (the records kernels of the range, behind the chunk's own: lx_level2_host.cpp)
        LX_HIP(h, hipEventRecord(ln.ev_k, h->stream));
        LX_HIP(h, hipStreamWaitEvent(h->stream2, ln.ev_k, 0));
        LX_HIP(h, hipMemcpyAsync(ln.p_cnt.ptr, d_cnt, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream2));
        LX_HIP(h, hipEventRecord(ln.ev_cnt, h->stream2));
        in_flight[L] = true;
        t_issue += ms(t1, now());
        return LX_OK;
    };

    // ---- ONE chunk in TWO calls (slots by wavefront): the plan's pool goes to the GPU -- its slot list, its table, its sweep -- while
    // the streamed part of the plan is still being made on the host threads; the second call sweeps the streamed wavefronts into slots
    // behind the pool's and runs what follows a sweep ONCE over all of them: one selection, one backtrace, one result list.  (Rounds
    // 3-4: the pool as a chunk of its own -- its backtrace is the latency of its longest walks, 1.6 ms for a fifth of the survivors --
    // or, for small lists, everything planned before the first launch.)
    struct TwoCall
    {
        uint64_t n1 = 0, nw1 = 0, dw0 = 0, ovf_cap = 0, ovf_dw = 0, total_dw = 0, stride = 0, max_q = 0, max_s = 0, cap_slots = 0;
    } two;
    auto wf_dwords = [&](uint64_t w, bool wide) -> uint64_t // what wavefront w's sixteen slots take
    {
        uint64_t const pc     = (uint64_t)lx::trace_cfg_panel(mq_cfg) / 8;
        uint32_t const steps  = (uint32_t)(((uint64_t)wf_maxs[w] + 8 - 1 + 15) & ~15ull);
        uint64_t const panels = std::max<uint64_t>(1, ((uint64_t)wf_pan[w] + pc - 1) / pc);
        return kWave * panels * (wide ? lx::ckpt_slot_dwords(mq_cfg, steps) : lx::ckpt16_slot_dwords(mq_cfg, steps));
    };
    auto fill_table = [&](lx::WfSlots * tab, uint64_t wlo, uint64_t whi, bool wide) -> uint64_t // offsets from 0; returns the dwords
    {
        uint64_t const pc  = (uint64_t)lx::trace_cfg_panel(mq_cfg) / 8;
        uint64_t       off = 0;
        for (uint64_t w = wlo; w < whi; ++w)
        {
            uint32_t const steps  = (uint32_t)(((uint64_t)wf_maxs[w] + 8 - 1 + 15) & ~15ull);
            uint32_t const panels = (uint32_t)std::max<uint64_t>(1, ((uint64_t)wf_pan[w] + pc - 1) / pc);
            tab[w - wlo]          = lx::WfSlots{off, steps, panels};
            off += wf_dwords(w, wide);
        }
        return off;
    };
    auto chunk_stats = [&](XbPrep & pr, uint64_t wlo, uint64_t whi)
    {
        uint64_t padded = 0;
        for (uint64_t w = wlo; w < whi; ++w)
            padded += kWave * ((uint64_t)wf_pan[w] * 8) * ((uint64_t)wf_maxs[w] + 7);
        pr.exec_cells += padded;
        h->xb_stats[3] += padded;
        h->xb_stats[1] += (whi - wlo) * kWave;
    };
    // first call: wavefronts [0, w1) = the pool; cap_slots bounds the chunk's slots, (cap_pan, cap_s) its widest query and longest window,
    // rest_dw is what the second call's slots are expected to take
    auto enqueue_mq_first = [&](int L, uint64_t w1, uint64_t cap_slots, uint64_t cap_pan, uint64_t cap_s, uint64_t rest_dw) -> int
    {
        auto const          t0 = now();
        lx_handle::XbLane & ln = h->xb[L];
        XbPrep &            pr = prep[L];
        uint64_t const      panel = (uint64_t)lx::trace_cfg_panel(mq_cfg), slots1 = w1 * kWave;
        pr.k0 = 0, pr.k1 = w1, pr.slots = slots1, pr.cap_sel = (cap_slots + 7) / 8 * 8 + 8, pr.exec_cells = 0, pr.max_s = cap_s, pr.max_pan = cap_pan;
        int rc2;
        if ((rc2 = ensure_pinned(h, ln.p_orig, cap_slots * sizeof(uint32_t))) || (rc2 = ensure_pinned(h, ln.p_wft, (cap_slots / kWave + 1) * sizeof(lx::WfSlots))))
            return rc2;
        uint32_t * const slot_orig = static_cast<uint32_t *>(ln.p_orig.ptr);
        std::memcpy(slot_orig, plan_slot.data(), slots1 * sizeof(uint32_t));
        chunk_stats(pr, 0, w1);
        two           = TwoCall{};
        two.max_q     = (cap_pan + panel / 8 - 1) / (panel / 8) * panel;
        two.max_s     = cap_s;
        two.stride    = (two.max_q + two.max_s + 3) & ~3ull;
        two.cap_slots = cap_slots;
        two.n1        = slots1;
        two.nw1       = w1;
        t_prep += ms(t0, now());
        auto const t1 = now();
        if ((rc2 = ensure(h, ln.d_orig, cap_slots * sizeof(uint32_t))) || (rc2 = ensure(h, ln.d_ext, cap_slots * sizeof(lx_extension))) ||
            (rc2 = ensure(h, ln.d_min, cap_slots * sizeof(int32_t))) || (rc2 = ensure(h, ln.d_score, cap_slots * sizeof(int32_t))) ||
            (rc2 = ensure(h, ln.d_hsp, pr.cap_sel * sizeof(lx_hsp))) || (rc2 = ensure(h, ln.d_ops, pr.cap_sel * two.stride + 16)) ||
            (rc2 = ensure(h, ln.d_rle, pr.cap_sel * two.stride + 16)) || (rc2 = ensure(h, ln.d_src, pr.cap_sel * sizeof(uint32_t))) ||
            (rc2 = ensure(h, ln.d_len, pr.cap_sel * sizeof(uint32_t))) || (rc2 = ensure(h, ln.d_cnt, 5 * sizeof(uint64_t))) ||
            (rc2 = ensure_pinned(h, ln.p_cnt, 5 * sizeof(uint64_t))) || (rc2 = ensure(h, ln.d_wft, (cap_slots / kWave + 1) * sizeof(lx::WfSlots))))
            return rc2;
        h->opt_max_qlen  = two.max_q;
        h->opt_max_slen  = two.max_s;
        h->opt_query_run = 2;
        h->mq_wide_call  = mq_cfg == 1 && !lx::dev_aids().mq_no_wide && (h->mq_wide_call ? h->mq_decl_frac > 0.01 : h->mq_decl_frac > 0.03);
        pr.wide          = h->mq_wide_call;
        lx::WfSlots * const tab = static_cast<lx::WfSlots *>(ln.p_wft.ptr);
        two.dw0          = fill_table(tab, 0, w1, pr.wide);
        {
            // (overflow slots have the size of the chunk's widest query and longest window: an eighth of the slots, within 16 GiB)
            uint64_t const steps = (two.max_s + 8 - 1 + 15) & ~15ull, s32 = (two.max_q / panel) * lx::ckpt_slot_dwords(mq_cfg, (uint32_t)steps);
            two.ovf_cap          = std::min<uint64_t>(std::min<uint64_t>(cap_slots, cap_slots / 8 + 64), std::max<uint64_t>(1024, (4ull << 30) / std::max<uint64_t>(s32, 1)));
            two.ovf_dw           = two.ovf_cap * s32;
        }
        // (reserved now: the pool's slots, the overflow slots, what the caller expects the second call's slots to take -- within the budget)
        two.total_dw = std::max<uint64_t>(two.dw0 + two.ovf_dw, std::min<uint64_t>(h->opt_trace_bytes / 4, two.dw0 + two.ovf_dw + rest_dw));
        LX_HIP(h, hipMemcpyAsync(ln.d_orig.ptr, slot_orig, slots1 * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream3));
        LX_HIP(h, hipMemcpyAsync(ln.d_wft.ptr, tab, w1 * sizeof(lx::WfSlots), hipMemcpyHostToDevice, h->stream3));
        LX_HIP(h, hipEventRecord(ln.ev_up, h->stream3));
        LX_HIP(h, hipStreamWaitEvent(h->stream, ln.ev_up, 0));
        LX_HIP(h, lx::launch_slot_gather(static_cast<lx::Extension const *>(ri ? ri->d_ext_all : h->d_ext_all.ptr),
                                         (min_score || (ri && ri->d_min_all)) ? static_cast<int32_t const *>(ri ? ri->d_min_all : h->d_min_all.ptr) : nullptr,
                                         min_score_all, static_cast<uint32_t const *>(ln.d_orig.ptr), slots1, static_cast<lx::Extension *>(ln.d_ext.ptr),
                                         static_cast<int32_t *>(ln.d_min.ptr), h->stream));
        h->mq_tab          = lx_handle::MqTab{};
        h->mq_tab.dev      = ln.d_wft.ptr;
        h->mq_tab.n0       = slots1;
        h->mq_tab.dw0      = two.dw0;
        h->mq_tab.ovf_cap  = two.ovf_cap;
        h->mq_tab.total_dw = two.total_dw;
        h->mq_tab.part     = 1;
        uint64_t * const d_cnt = static_cast<uint64_t *>(ln.d_cnt.ptr);
        FusedExtra       fx;
        fx.ops_stride = two.stride;
        fx.d_rle      = static_cast<uint8_t *>(ln.d_rle.ptr);
        fx.d_rle_top  = reinterpret_cast<unsigned long long *>(d_cnt + 2);
        fx.rle_cap    = pr.cap_sel * two.stride;
        fx.d_src_out  = static_cast<uint32_t *>(ln.d_src.ptr);
        fx.d_rle_len  = static_cast<uint32_t *>(ln.d_len.ptr);
        rc2 = fused_impl(h, slot, d_qptr, sref.dev, ln.d_ext.ptr, cap_slots, ln.d_min.ptr, 0, ln.d_score.ptr, ln.d_hsp.ptr, ln.d_ops.ptr, nullptr, d_cnt, h->stream, 3,
                         true, &fx);
        h->mq_tab = lx_handle::MqTab{};
        t_issue += ms(t1, now());
        return rc2;
    };
    // second call: wavefronts [w_mid, w1) of the plan (none: the pool stays a chunk of its own) behind the first call's
    auto enqueue_mq_second = [&](int L, uint64_t w_mid, uint64_t w1) -> int
    {
        auto const          t0 = now();
        lx_handle::XbLane & ln = h->xb[L];
        XbPrep &            pr = prep[L];
        uint64_t const      slots2 = (w1 - w_mid) * kWave, total = two.n1 + slots2;
        int rc2;
        uint32_t * const    slot_orig = static_cast<uint32_t *>(ln.p_orig.ptr);
        lx::WfSlots * const tab       = static_cast<lx::WfSlots *>(ln.p_wft.ptr) + two.nw1;
        if (slots2)
            std::memcpy(slot_orig + two.n1, plan_slot.data() + w_mid * kWave, slots2 * sizeof(uint32_t));
        chunk_stats(pr, w_mid, w1);
        uint64_t const dw1 = fill_table(tab, w_mid, w1, pr.wide);
        pr.k1    = w1;
        pr.slots = total;
        t_prep += ms(t0, now());
        auto const t1 = now();
        if (slots2)
        {
            LX_HIP(h, hipMemcpyAsync(static_cast<uint32_t *>(ln.d_orig.ptr) + two.n1, slot_orig + two.n1, slots2 * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream3));
            LX_HIP(h, hipMemcpyAsync(static_cast<lx::WfSlots *>(ln.d_wft.ptr) + two.nw1, tab, (w1 - w_mid) * sizeof(lx::WfSlots), hipMemcpyHostToDevice, h->stream3));
            LX_HIP(h, hipEventRecord(ln.ev_up, h->stream3));
            LX_HIP(h, hipStreamWaitEvent(h->stream, ln.ev_up, 0));
            LX_HIP(h, lx::launch_slot_gather(static_cast<lx::Extension const *>(ri ? ri->d_ext_all : h->d_ext_all.ptr),
                                             (min_score || (ri && ri->d_min_all)) ? static_cast<int32_t const *>(ri ? ri->d_min_all : h->d_min_all.ptr) : nullptr,
                                             min_score_all, static_cast<uint32_t const *>(ln.d_orig.ptr) + two.n1, slots2,
                                             static_cast<lx::Extension *>(ln.d_ext.ptr) + two.n1, static_cast<int32_t *>(ln.d_min.ptr) + two.n1, h->stream));
        }
        h->opt_max_qlen    = two.max_q;
        h->opt_max_slen    = two.max_s;
        h->opt_query_run   = 2;
        h->mq_wide_call    = pr.wide;
        h->mq_tab          = lx_handle::MqTab{};
        h->mq_tab.dev      = ln.d_wft.ptr;
        h->mq_tab.n0       = two.n1;
        h->mq_tab.dw0      = two.dw0;
        h->mq_tab.dw1      = dw1;
        h->mq_tab.ovf_cap  = two.ovf_cap;
        h->mq_tab.total_dw = two.total_dw;
        h->mq_tab.part     = 2;
        uint64_t * const d_cnt = static_cast<uint64_t *>(ln.d_cnt.ptr);
        FusedExtra       fx;
        fx.ops_stride = two.stride;
        fx.d_rle      = static_cast<uint8_t *>(ln.d_rle.ptr);
        fx.d_rle_top  = reinterpret_cast<unsigned long long *>(d_cnt + 2);
        fx.rle_cap    = pr.cap_sel * two.stride;
        fx.d_src_out  = static_cast<uint32_t *>(ln.d_src.ptr);
        fx.d_rle_len  = static_cast<uint32_t *>(ln.d_len.ptr);
        rc2 = fused_impl(h, slot, d_qptr, sref.dev, ln.d_ext.ptr, total, ln.d_min.ptr, 0, ln.d_score.ptr, ln.d_hsp.ptr, ln.d_ops.ptr, nullptr, d_cnt, h->stream, 3, true,
                         &fx);
        h->mq_tab = lx_handle::MqTab{};
        if (rc2)
            return rc2;
        LX_HIP(h, lx::launch_slot_scatter(static_cast<uint32_t const *>(ln.d_orig.ptr), total, static_cast<int32_t const *>(ln.d_score.ptr),
                                          static_cast<int32_t *>(h->d_score_all.ptr), static_cast<uint32_t *>(ln.d_src.ptr), d_cnt, pr.cap_sel, h->stream));
        LX_HIP(h, hipMemcpyAsync(d_cnt + 3, h->d_ws_top, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream));
        LX_HIP(h, hipMemcpyAsync(d_cnt + 4, h->d_ws_top + 6, sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream));
        LX_HIP(h, hipEventRecord(ln.ev_k, h->stream));
        LX_HIP(h, hipStreamWaitEvent(h->stream2, ln.ev_k, 0));
        LX_HIP(h, hipMemcpyAsync(ln.p_cnt.ptr, d_cnt, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream2));
        LX_HIP(h, hipEventRecord(ln.ev_cnt, h->stream2));
        in_flight[L] = true;
        t_issue += ms(t1, now());
        return LX_OK;
    };

    // ---- list form (lx_extend_batch_list): the chunk's survivors are appended to the handle's result buffers -- position in the
    // caller's list, record, where the codes begin -- and the chunk's codes to h->ext_bytes as one block; of the host's arrays
    // of size n only the scores are touched
    auto append_list = [&](int L, uint64_t count, uint64_t nrle, uint32_t const * slot_src /* NULL: the device translated */) -> int
    {
        lx_handle::XbLane &    ln    = h->xb[L];
        lx_hsp const * const   hs    = static_cast<lx_hsp const *>(ln.p_hsp.ptr);
        uint32_t const * const src   = static_cast<uint32_t const *>(ln.p_src.ptr);
        uint8_t const * const  codes = static_cast<uint8_t const *>(ln.p_rle.ptr);
        std::vector<uint64_t>  part(nthreads + 1, 0), untraced(nthreads, ~0ull);
        parallel_ranges(count, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t k = 0, bad = ~0ull; // (locals: the per-thread slots share cache lines)
                            for (uint64_t e = lo; e < hi; ++e)
                            {
                                if (src[e] == 0xffffffffu) // padding of the survivor list
                                    continue;
                                if (hs[e].score < 0)
                                    bad = std::min<uint64_t>(bad, slot_src ? slot_src[src[e]] : src[e]);
                                else
                                    ++k;
                            }
                            part[t + 1] = k;
                            untraced[t] = bad;
                        });
        for (uint64_t u : untraced)
            if (u != ~0ull)
                return fail(h, LX_EOVERFLOW, "extension %llu could not be traced", (unsigned long long)u);
        for (unsigned t = 0; t < nthreads; ++t)
            part[t + 1] += part[t];
        uint64_t const base = h->res_count, total = base + part[nthreads];
        if (!h->res_index.grow(total * sizeof(uint32_t) + 16) || !h->res_hsp.grow(total * sizeof(lx_hsp) + 16) ||
            !h->res_off.grow(total * sizeof(uint64_t) + 16) || !h->ext_bytes.grow(ops_total + nrle + 16))
            return fail(h, LX_ENOMEM, "out of host memory for %llu survivors", (unsigned long long)total);
        uint32_t * const res_index = reinterpret_cast<uint32_t *>(h->res_index.data());
        lx_hsp * const   res_hsp   = reinterpret_cast<lx_hsp *>(h->res_hsp.data());
        uint64_t * const res_off   = reinterpret_cast<uint64_t *>(h->res_off.data());
        uint8_t * const  dst       = h->ext_bytes.data() + ops_total;
        parallel_ranges(count, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t k = base + part[t];
                            for (uint64_t e = lo; e < hi; ++e)
                            {
                                if (src[e] == 0xffffffffu)
                                    continue;
                                lx_hsp r     = hs[e];
                                res_index[k] = slot_src ? slot_src[src[e]] : src[e];
                                res_off[k]   = ops_total + (uint32_t)r.ops_shift;
                                r.ops_shift  = 0;
                                res_hsp[k]   = r;
                                ++k;
                            }
                        });
        parallel_ranges(nrle, nthreads, [&](unsigned, uint64_t lo, uint64_t hi) { std::memcpy(dst + lo, codes + lo, hi - lo); });
        h->res_count = total;
        ops_total += nrle;
        return LX_OK;
    };

    // ---- results of a multi-query chunk: the survivors' records and ops, addressed by caller index (the device translated the
    // list); the scores of every extension come back once, in caller order, at the end of the call
    std::vector<std::pair<uint64_t, uint64_t>> redo; // chunks (wavefront ranges) to run again with int16-pair slots
    auto collect_mq = [&](int L) -> int
    {
        lx_handle::XbLane & ln = h->xb[L];
        XbPrep &            pr = prep[L];
        in_flight[L]           = false;
        auto const t0          = now();
        LX_HIP(h, hipEventSynchronize(ln.ev_cnt));
        auto const t_ev        = now();
        uint64_t const * const cnt = static_cast<uint64_t const *>(ln.p_cnt.ptr);
        uint64_t const count = cnt[0], nrle = cnt[2];
        {
            uint32_t flags[2];
            std::memcpy(flags, cnt + 3, sizeof(flags));
            if (hm.on)
                fprintf(stderr, "[lx host ms]   chunk %llu-%llu: %llu slots%s (%.2f G executed cells, longest window %llu, <= %llu columns per lane), error word %u, %u beyond the codes, %llu survivors\n",
                        (unsigned long long)pr.k0, (unsigned long long)pr.k1, (unsigned long long)pr.slots, pr.wide ? " (wide)" : "", (double)pr.exec_cells / 1e9,
                        (unsigned long long)pr.max_s, (unsigned long long)pr.max_pan, flags[1], (uint32_t)cnt[4], (unsigned long long)cnt[1]);
            if (flags[1] == 4 && !pr.wide && mq_cfg == 1 && !lx::dev_aids().mq_no_wide)
            {
                // More windows beyond the compact codes than the overflow area holds int16-pair slots for (its slots are sized for the
                // chunk's longest window): the chunk runs again with int16 pairs from the sweep itself, which needs no overflow area.
                // Nothing of this attempt is kept (the scores it scattered are written again).
                redo.push_back({pr.k0, pr.k1});
                h->mq_decl_frac = 1.0;
                return LX_OK;
            }
            int const rcf = error_for_flag(h, flags[1]);
            if (rcf)
                return rcf;
        }
        if (count > pr.cap_sel)
            return fail(h, LX_ESTATE, "survivor list longer than its capacity");
        if (pr.slots)
        {
            h->surv_frac    = (double)cnt[1] / (double)pr.slots;
            h->mq_decl_frac = (double)(uint32_t)cnt[4] / (double)pr.slots;
        }
        int rc2;
        if (dev_list)
        {
            // The survivors stay where the backtrace left them (lx_records.hip makes the result records from them): the chunk's list
            // joins the call's on the device -- a copy kernel in stream order, so that this lane's buffers are free for the chunk after
            // next --, only the run-length codes come down, into the call's code bytes.
            auto & l2 = h->l2;
            if (by_range)
            {
                uint64_t const code_base = ops_total;
                if (nrle && want_codes)
                {
                    if (!h->ext_bytes.grow(ops_total + nrle + 16))
                        return fail(h, LX_ENOMEM, "out of host memory for %llu bytes of alignment codes", (unsigned long long)(ops_total + nrle));
                    LX_HIP(h, hipMemcpyAsync(h->ext_bytes.data() + ops_total, ln.d_rle.ptr, nrle, hipMemcpyDeviceToHost, h->stream3));
                    LX_HIP(h, hipStreamSynchronize(h->stream3));
                }
                ops_total += nrle;
                l2.surv_total += count;
                h->res_count = l2.surv_total;
                auto const t1 = now();
                t_wait += ms(t0, t1);
                rc2 = ri->chunk_records->collect(pr.range, code_base, in_flight[L ^ 1]);
                t_unpack += ms(t1, now());
                if (hm.on)
                    fprintf(stderr, "[lx host ms]     ... its counts came after %.2f ms of waiting, its %llu code bytes in %.2f more; rows and columns of range %llu in %.2f\n",
                            ms(t0, t_ev), (unsigned long long)nrle, ms(t_ev, t1), (unsigned long long)pr.range, ms(t1, now()));
                return rc2;
            }
            if (l2.surv_total + count > l2.surv_cap)
                return fail(h, LX_ESTATE, "the call's survivor list is longer than its capacity");
            LX_HIP(h, lx::rec_launch_append(static_cast<lx::Hsp const *>(ln.d_hsp.ptr), static_cast<uint32_t const *>(ln.d_src.ptr),
                                            static_cast<uint64_t const *>(ln.d_cnt.ptr), count, ops_total, static_cast<lx::Hsp *>(l2.d_surv_hsp.ptr) + l2.surv_total,
                                            static_cast<uint32_t *>(l2.d_surv_src.ptr) + l2.surv_total, static_cast<uint64_t *>(l2.d_surv_codes.ptr) + l2.surv_total,
                                            h->stream));
            l2.surv_total += count;
            if (nrle && want_codes)
            {
                if (!h->ext_bytes.grow(ops_total + nrle + 16))
                    return fail(h, LX_ENOMEM, "out of host memory for %llu bytes of alignment codes", (unsigned long long)(ops_total + nrle));
                // (straight into the call's code bytes: a copy of this size into ordinary memory runs at the link's rate)
                LX_HIP(h, hipMemcpyAsync(h->ext_bytes.data() + ops_total, ln.d_rle.ptr, nrle, hipMemcpyDeviceToHost, h->stream3));
                LX_HIP(h, hipStreamSynchronize(h->stream3));
            }
            ops_total += nrle;
            h->res_count = l2.surv_total;
            auto const t1 = now();
            t_wait += ms(t0, t1);
            if (hm.on)
                fprintf(stderr, "[lx host ms]     ... its counts came after %.2f ms of waiting, its %llu code bytes in %.2f more; the %llu records stay on the device\n",
                        ms(t0, t_ev), (unsigned long long)nrle, ms(t_ev, t1), (unsigned long long)count);
            return LX_OK;
        }
        if ((rc2 = ensure_pinned(h, ln.p_hsp, count * sizeof(lx_hsp) + 16)) || (rc2 = ensure_pinned(h, ln.p_src, count * sizeof(uint32_t) + 16)) ||
            (rc2 = ensure_pinned(h, ln.p_len, count * sizeof(uint32_t) + 16)) || (rc2 = ensure_pinned(h, ln.p_rle, nrle + 16)))
            return rc2;
        if (count)
        {
            LX_HIP(h, hipMemcpyAsync(ln.p_hsp.ptr, ln.d_hsp.ptr, count * sizeof(lx_hsp), hipMemcpyDeviceToHost, h->stream3));
            LX_HIP(h, hipMemcpyAsync(ln.p_src.ptr, ln.d_src.ptr, count * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream3));
            LX_HIP(h, hipMemcpyAsync(ln.p_len.ptr, ln.d_len.ptr, count * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream3));
            if (nrle)
                LX_HIP(h, hipMemcpyAsync(ln.p_rle.ptr, ln.d_rle.ptr, nrle, hipMemcpyDeviceToHost, h->stream3));
        }
        LX_HIP(h, hipStreamSynchronize(h->stream3));
        auto const t1 = now();
        t_wait += ms(t0, t1);
        if (as_list)
        {
            rc2 = append_list(L, count, nrle, nullptr);
            t_unpack += ms(t1, now());
            if (hm.on)
                fprintf(stderr, "[lx host ms]     ... its counts came after %.2f ms of waiting, its %llu records + %llu code bytes in %.2f more, appended in %.2f\n", ms(t0, t_ev),
                        (unsigned long long)count, (unsigned long long)nrle, ms(t_ev, t1), ms(t1, now()));
            return rc2;
        }
        lx_hsp const * const   hs       = static_cast<lx_hsp const *>(ln.p_hsp.ptr);
        uint32_t const * const src_orig = static_cast<uint32_t const *>(ln.p_src.ptr);
        uint8_t const * const  codes    = static_cast<uint8_t const *>(ln.p_rle.ptr);
        uint32_t const * const code_len = static_cast<uint32_t const *>(ln.p_len.ptr);
        std::vector<uint64_t> & pos_off = h->xb_off;
        pos_off.resize(count + 1);
        std::vector<uint64_t> part(nthreads + 1, 0);
        parallel_ranges(count, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t sum = 0;
                            for (uint64_t e = lo; e < hi; ++e)
                            {
                                uint64_t const len = (src_orig[e] != 0xffffffffu && hs[e].score > 0) ? (want_rle ? (uint64_t)code_len[e] : (uint64_t)hs[e].n_ops) : 0;
                                pos_off[e] = len;
                                sum += len;
                            }
                            part[t + 1] = sum;
                        });
        part[0] = ops_total;
        for (unsigned t = 0; t < nthreads; ++t)
            part[t + 1] += part[t];
        uint64_t const total = part[nthreads];
        if (!h->ext_bytes.grow(total + 16))
            return fail(h, LX_ENOMEM, "out of host memory for %llu bytes of alignment ops", (unsigned long long)(total + 16));
        uint8_t * const dst = h->ext_bytes.data();
        std::vector<uint64_t> untraced(nthreads, ~0ull);
        parallel_ranges(count, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t at = part[t];
                            for (uint64_t e = lo; e < hi; ++e)
                            {
                                uint64_t const len = pos_off[e];
                                uint32_t const orig = src_orig[e];
                                if (orig == 0xffffffffu)
                                    continue;
                                lx_hsp r = hs[e];
                                if (r.score < 0)
                                {
                                    untraced[t] = std::min<uint64_t>(untraced[t], orig);
                                    continue;
                                }
                                uint8_t const * const c = codes + (uint32_t)r.ops_shift;
                                if (r.score > 0 && want_rle)
                                    std::memcpy(dst + at, c, (size_t)len);
                                else if (r.score > 0)
                                    rle_expand(c, r.n_ops, dst + at);
                                r.ops_shift       = 0;
                                out_hsp[orig]     = r;
                                out_ops_off[orig] = at;
                                at += len;
                            }
                        });
        for (uint64_t u : untraced)
            if (u != ~0ull)
                return fail(h, LX_EOVERFLOW, "extension %llu could not be traced", (unsigned long long)u);
        ops_total = total;
        t_unpack += ms(t1, now());
        return LX_OK;
    };

    // ---- results of the chunk in lane L -> the caller's arrays
    auto collect = [&](int L) -> int
    {
        lx_handle::XbLane & ln = h->xb[L];
        XbPrep &            pr = prep[L];
        in_flight[L]           = false;
        auto const t0          = now();
        LX_HIP(h, hipEventSynchronize(ln.ev_cnt));
        uint64_t const * const cnt = static_cast<uint64_t const *>(ln.p_cnt.ptr);
        uint64_t const count = cnt[0], nrle = cnt[2];
        {
            uint32_t flags[2];
            std::memcpy(flags, cnt + 3, sizeof(flags));
            int const rcf = error_for_flag(h, flags[1]);
            if (rcf)
                return rcf;
        }
        if (count > pr.cap_sel)
            return fail(h, LX_ESTATE, "survivor list longer than its capacity");
        if (pr.slots)
            h->surv_frac = (double)cnt[1] / (double)pr.slots; // (adaptive pass-2 mode of the next chunks: fused_impl)
        int rc2;
        if ((rc2 = ensure_pinned(h, ln.p_hsp, count * sizeof(lx_hsp) + 16)) || (rc2 = ensure_pinned(h, ln.p_src, count * sizeof(uint32_t) + 16)) ||
            (rc2 = ensure_pinned(h, ln.p_len, count * sizeof(uint32_t) + 16)) ||
            (rc2 = ensure_pinned(h, ln.p_rle, nrle + 16)))
            return rc2;
        // (on the upload stream: stream2 already holds the next chunk's first-stage copies, which wait for its kernels)
        if (count)
        {
            LX_HIP(h, hipMemcpyAsync(ln.p_hsp.ptr, ln.d_hsp.ptr, count * sizeof(lx_hsp), hipMemcpyDeviceToHost, h->stream3));
            LX_HIP(h, hipMemcpyAsync(ln.p_src.ptr, ln.d_src.ptr, count * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream3));
            LX_HIP(h, hipMemcpyAsync(ln.p_len.ptr, ln.d_len.ptr, count * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream3));
            if (nrle)
                LX_HIP(h, hipMemcpyAsync(ln.p_rle.ptr, ln.d_rle.ptr, nrle, hipMemcpyDeviceToHost, h->stream3));
        }
        LX_HIP(h, hipStreamSynchronize(h->stream3));
        auto const t1 = now();
        t_wait += ms(t0, t1);
        int32_t const * const  sc      = static_cast<int32_t const *>(ln.p_score.ptr);
        lx_hsp const * const   hs      = static_cast<lx_hsp const *>(ln.p_hsp.ptr);
        uint32_t const * const sel_src = static_cast<uint32_t const *>(ln.p_src.ptr);
        uint8_t const * const  codes   = static_cast<uint8_t const *>(ln.p_rle.ptr);
        uint32_t const * const code_len = static_cast<uint32_t const *>(ln.p_len.ptr);
        uint32_t const * const slot_src = pr.slot_src.data();
        if (as_list)
        {
            parallel_ranges(pr.slots, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                for (uint64_t o = lo; o < hi; ++o)
                                    if (slot_src[o] != 0xffffffffu)
                                        out_score[slot_src[o]] = sc[o];
                            });
            rc2 = append_list(L, count, nrle, slot_src);
            t_unpack += ms(t1, now());
            return rc2;
        }
        // (1) per survivor: how many bytes its ops take in the handle's buffer (column bytes, or the codes themselves),
        //     and which list position a slot has
        std::vector<uint64_t> & pos_off  = h->xb_off;
        std::vector<uint32_t> & slot_pos = h->xb_pos;
        pos_off.resize(count + 1);
        slot_pos.resize(pr.slots);
        parallel_ranges(pr.slots, nthreads,
                        [&](unsigned, uint64_t lo, uint64_t hi) { std::fill(slot_pos.begin() + lo, slot_pos.begin() + hi, 0xffffffffu); });
        std::vector<uint64_t> part(nthreads + 1, 0);
        parallel_ranges(count, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t sum = 0;
                            for (uint64_t e = lo; e < hi; ++e)
                            {
                                uint64_t len = 0;
                                if (sel_src[e] != 0xffffffffu)
                                {
                                    slot_pos[sel_src[e]] = (uint32_t)e;
                                    if (hs[e].score > 0)
                                        len = want_rle ? (uint64_t)code_len[e] : (uint64_t)hs[e].n_ops;
                                }
                                pos_off[e] = len;
                                sum += len;
                            }
                            part[t + 1] = sum;
                        });
        auto const tu1 = now();
        t_u1 += ms(t1, tu1);
        // (2) offsets: prefix over the threads' shares, then inside each share
        part[0] = ops_total;
        for (unsigned t = 0; t < nthreads; ++t)
            part[t + 1] += part[t];
        uint64_t const total = part[nthreads];
        parallel_ranges(count, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            uint64_t at = part[t];
                            for (uint64_t e = lo; e < hi; ++e)
                            {
                                uint64_t const len = pos_off[e];
                                pos_off[e]         = at;
                                at += len;
                            }
                        });
        pos_off[count] = total;
        if (!h->ext_bytes.grow(total + 16))
            return fail(h, LX_ENOMEM, "out of host memory for %llu bytes of alignment ops", (unsigned long long)(total + 16));
        uint8_t * const dst = h->ext_bytes.data();
        t_u2 += ms(tu1, now());
        // (3) one pass over the chunk's slots: score and record of every extension, the survivors' ops
        std::vector<uint64_t> untraced(nthreads, ~0ull);
        parallel_ranges(pr.slots, nthreads,
                        [&](unsigned t, uint64_t lo, uint64_t hi)
                        {
                            for (uint64_t o = lo; o < hi; ++o)
                            {
                                uint32_t const orig = slot_src[o];
                                if (orig == 0xffffffffu)
                                    continue;
                                out_score[orig]  = sc[o];
                                uint32_t const e = slot_pos[o];
                                if (e == 0xffffffffu)
                                {
                                    lx_hsp r{};
                                    r.score           = sc[o];
                                    out_hsp[orig]     = r;
                                    out_ops_off[orig] = 0;
                                    continue;
                                }
                                lx_hsp r = hs[e];
                                if (r.score < 0)
                                {
                                    untraced[t] = std::min<uint64_t>(untraced[t], orig);
                                    continue;
                                }
                                uint8_t const * const c = codes + (uint32_t)r.ops_shift;
                                if (r.score > 0 && want_rle)
                                    std::memcpy(dst + pos_off[e], c, (size_t)(pos_off[e + 1] - pos_off[e]));
                                else if (r.score > 0)
                                    rle_expand(c, r.n_ops, dst + pos_off[e]);
                                r.ops_shift       = 0;
                                out_hsp[orig]     = r;
                                out_ops_off[orig] = pos_off[e];
                            }
                        });
        for (uint64_t u : untraced)
            if (u != ~0ull)
                return fail(h, LX_EOVERFLOW, "extension %llu could not be traced", (unsigned long long)u);
        ops_total = total;
        t_unpack += ms(t1, now());
        return LX_OK;
    };

    // ---- the pipeline: prepare + queue chunk c, then unpack chunk c - 1 while c runs
    uint64_t k0 = 0;
    int      c  = 0;
    if (use_mq)
    {
        // chunks = ranges of the plan's wavefronts, about chunk_target slots.  A chunk may span panel counts (its slots are sized
        // for its widest query, its narrower queries run the multi-panel kernel over one panel): a chunk boundary wherever the
        // panel count changes was measured on the ragged list of bench.py and costs more than it saves -- 3 chunks 20.8 ms,
        // 2 chunks 18.9 ms: every chunk pays the fixed cost of a backtrace launch (~0.5-1 ms), the single-panel kernel saves a
        // tenth of a 0.8 ms sweep
        uint64_t const per_chunk = std::max<uint64_t>(1, chunk_target / kWave);
        h->xb_stats[2] = mq_cells;
        // the caller's list and cut-offs onto the device (pinned staging, filled by the pool), scores in caller order zeroed
        {
            if (!list_uploaded && (rc = upload_list()))
                return rc;
            t_prep += t_upload;
        }
        if (as_list && ri && ri->keep_on_device)
        {
            // the Level-2 driver makes its records on the device (lx_records.hip): room for every chunk's survivor list, padding included
            auto &         l2  = h->l2;
            // (fillers never survive, a chunk's list is padded by less than 16 entries: the list's windows + a margin per chunk -- not the
            // plan's slots: the streamed part of a host plan is made later)
            uint64_t const cap = n + n / 64 + 8192;
            if ((rc = ensure(h, l2.d_surv_hsp, cap * sizeof(lx_hsp))) || (rc = ensure(h, l2.d_surv_src, cap * sizeof(uint32_t))) ||
                (rc = ensure(h, l2.d_surv_codes, cap * sizeof(uint64_t))))
                return rc;
            l2.surv_cap       = cap;
            l2.surv_total     = 0;
            l2.surv_on_device = true;
            dev_list          = true;
            want_codes        = ri->want_codes;
            // records chunk by chunk where every range of the plan is a chunk the budgets admit (else: the call's list, one chain at the end)
            if (ri->chunk_records && ri->chunk_records->n_ranges >= 1 && preplanned)
            {
                auto const &   cr = *ri->chunk_records;
                uint64_t const pc = (uint64_t)lx::trace_cfg_panel(mq_cfg) / 8;
                by_range          = cr.cut_wf[0] == 0 && cr.cut_wf[cr.n_ranges] == nwf;
                for (uint64_t r = 0; r < cr.n_ranges && by_range; ++r)
                {
                    uint64_t const a = cr.cut_wf[r], b = cr.cut_wf[r + 1];
                    uint64_t       pm = 1, sm = 1;
                    for (uint64_t w = a; w < b; ++w)
                    {
                        pm = std::max<uint64_t>(pm, wf_pan[w]);
                        sm = std::max<uint64_t>(sm, wf_maxs[w]);
                    }
                    uint64_t const steps = (sm + 8 - 1 + 15) & ~15ull;
                    uint64_t const slot  = (pm + pc - 1) / pc * (lx::ckpt16_slot_dwords(mq_cfg, (uint32_t)steps) + lx::ckpt_slot_dwords(mq_cfg, (uint32_t)steps) / 8) * 4;
                    // (a range whose sweep overflows runs again WHOLE with int16-pair slots, about twice the codes': admitted against those too)
                    uint64_t const slot_w = mq_cfg == 1 && !lx::dev_aids().mq_no_wide ? (pm + pc - 1) / pc * lx::ckpt_slot_dwords(mq_cfg, (uint32_t)steps) * 4 : 0;
                    by_range = a < b && b - a <= 4 * per_chunk && (b - a) * kWave * std::max(slot, slot_w) <= h->opt_trace_bytes && (b - a) * kWave * (pm * 8 + sm) <= (8ull << 30);
                }
                l2.surv_by_range = by_range;
            }
        }
        bool     rows_cleared = false, stream_planned = use_solo || preplanned; // (the solo plan and a device plan are whole before the first chunk)
        uint64_t w0           = 0;
        // ONE launch for the pool and what follows it: the pool is a tenth of the list
        // in wavefronts that run up to three times as long as the others -- launched by itself it leaves most of the chip idle behind
        // its longest windows (ragged list of bench.py: 5 000 of 37 000 wavefronts, but 5.9 of 11.8 ms), launched with the rest
        // behind it the short wavefronts fill in.  The plan of the streamed part is then made before the first launch.
        // (lists of up to ~200 000 windows: a dozen rounds of the chip's wavefront slots.  Beyond that the pool by itself is several
        // rounds and the streamed part's plan is better made beside its kernels: 596 k windows 18.6 ms merged, 17.5 ms not;
        // 64 k windows of 300-500-residue queries 11.9 ms merged, 16.3 ms not)
        bool const merge_pool = !use_solo && !preplanned && live <= (lx::dev_aids().mq_merge_below ? lx::dev_aids().mq_merge_below : 200000);
        // wavefronts [wlo, whi) of the plan in launch order = longest first (what a wavefront executes is columns x steps; the blocks of a
        // launch are dealt to the chip's wavefront slots in index order, so a long wavefront late in the order ends the launch late)
        auto longest_first = [&](uint64_t wlo, uint64_t whi)
        {
            if (whi <= wlo + 1)
                return;
            uint64_t const cnt = whi - wlo;
            std::vector<uint32_t> & by_len = h->xb_pool_order, & tmp_slot = h->xb_pool_place, & tmp_pan = h->xb_pool_pan, & tmp_maxs = h->xb_pool_maxs;
            std::vector<uint32_t> & len_key = h->xb_pool_key, & len_tmp = h->xb_pool_tmp;
            by_len.resize(cnt);
            len_tmp.resize(cnt);
            len_key.resize(whi);
            for (uint64_t k = 0; k < cnt; ++k)
            {
                by_len[k]        = (uint32_t)(wlo + k);
                len_key[wlo + k] = 0x3fffffffu - (uint32_t)std::min<uint64_t>((uint64_t)wf_pan[wlo + k] * (wf_maxs[wlo + k] + 7), 0x3fffffffu);
            }
            radix_sort(by_len, len_tmp, len_key, cnt);
            tmp_slot.resize(cnt * kWave);
            tmp_pan.resize(cnt);
            tmp_maxs.resize(cnt);
            parallel_ranges(cnt, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                for (uint64_t k = lo; k < hi; ++k)
                                {
                                    std::memcpy(tmp_slot.data() + k * kWave, plan_slot.data() + (uint64_t)by_len[k] * kWave, kWave * sizeof(uint32_t));
                                    tmp_pan[k]  = wf_pan[by_len[k]];
                                    tmp_maxs[k] = wf_maxs[by_len[k]];
                                }
                            });
            parallel_ranges(cnt, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                std::memcpy(plan_slot.data() + (wlo + lo) * kWave, tmp_slot.data() + lo * kWave, (hi - lo) * kWave * sizeof(uint32_t));
                                std::memcpy(wf_pan.data() + wlo + lo, tmp_pan.data() + lo, (hi - lo) * sizeof(uint32_t));
                                std::memcpy(wf_maxs.data() + wlo + lo, tmp_maxs.data() + lo, (hi - lo) * sizeof(uint32_t));
                            });
        };
        if (merge_pool && !stream_planned)
        {
            auto const tp0 = now();
            plan_stream();
            stream_planned = true;
            longest_first(0, nwf); // (one launch for the whole plan: the streamed part's long wavefronts would start late)
            t_prep += ms(tp0, now());
        }
        uint64_t const pool_end = (use_solo || preplanned) ? 0 : pool_wf; // wavefronts before it: the pool (region 1 of a chunk that spans it)
        auto clear_rows = [&]()
        {
            // (beside the first chunk's kernels) every row starts as "no alignment"; the survivors' rows are written by
            // collect_mq, the scores of all rows at the end of the call
            auto const tz0 = now();
            parallel_ranges(n, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                std::memset(static_cast<void *>(out_hsp + lo), 0, (hi - lo) * sizeof(lx_hsp));
                                std::memset(out_ops_off + lo, 0, (hi - lo) * sizeof(uint64_t));
                            });
            rows_cleared = true;
            t_unpack += ms(tz0, now());
        };
        // ---- the pool and what follows it as ONE chunk in two calls (enqueue_mq_first / enqueue_mq_second)
        if (!use_solo && !merge_pool && !stream_planned && !by_range && pool_wf > 0)
        {
            auto const     tp0   = now();
            uint64_t const nruns = starts.size() - 1, pc = (uint64_t)lx::trace_cfg_panel(mq_cfg) / 8;
            // what the streamed part may come to: its windows in pairs (a run's last pair may be half empty), a wavefront closed
            // early for every fifth run, one per planning thread; its slot bytes from every run's own panels and longest window
            uint64_t nstream = 0, nstream_runs = 0, cap_pan = 1, cap_s = 1, dw1_est = 0;
            bool const maybe_wide = mq_cfg == 1 && !lx::dev_aids().mq_no_wide && h->mq_decl_frac > 0.01;
            for (uint64_t r = 0; r < nruns; ++r)
                if (pool_at[r] != starts[r])
                {
                    uint64_t const cnt = pool_at[r] - starts[r], pan = 0xfffu - (run_key[r] >> 16), maxs = ext[idx[pool_at[r] - 1]].s_len;
                    uint32_t const steps = (uint32_t)((maxs + 8 - 1 + 15) & ~15ull);
                    nstream += cnt;
                    ++nstream_runs;
                    cap_pan = std::max(cap_pan, pan);
                    cap_s   = std::max(cap_s, maxs);
                    dw1_est += (cnt + 1) / 2 * 2 * ((pan + pc - 1) / pc) * (maybe_wide ? lx::ckpt_slot_dwords(mq_cfg, steps) : lx::ckpt16_slot_dwords(mq_cfg, steps));
                }
            uint64_t dw0_est = 0;
            for (uint64_t w = 0; w < pool_wf; ++w)
            {
                cap_pan = std::max<uint64_t>(cap_pan, wf_pan[w]);
                cap_s   = std::max<uint64_t>(cap_s, wf_maxs[w]);
                dw0_est += wf_dwords(w, maybe_wide);
            }
            uint64_t const cap_slots = pool_wf * kWave + nstream + 5 * nstream_runs + kWave * (nthreads + 2);
            uint64_t const rest_dw   = dw1_est + dw1_est / 4 + (1u << 20); // (wavefronts share the largest of up to four runs' sizes)
            t_prep += ms(tp0, now());
            // (the budgets of a chunk: its checkpoint slots, its survivors' ops slots)
            if (nstream != 0 && cap_slots < (1ull << 31) && (cap_slots + 16) * (cap_pan * 8 + cap_s + 4) <= (8ull << 30) && dw0_est * 4 <= h->opt_trace_bytes / 2)
            {
                if ((rc = enqueue_mq_first(0, pool_wf, cap_slots, cap_pan, cap_s, rest_dw)))
                    return rc;
                auto const tp1 = now();
                plan_stream();
                stream_planned = true;
                longest_first(pool_wf, nwf); // (the second launch's wavefronts)
                // the wavefronts of the streamed part that fit behind the pool's (all of them, unless the estimate was short)
                uint64_t w_end = pool_wf, dw1 = 0;
                while (w_end < nwf && two.n1 + (w_end + 1 - pool_wf) * kWave <= two.cap_slots && wf_pan[w_end] <= cap_pan && wf_maxs[w_end] <= cap_s &&
                       two.dw0 + two.ovf_dw + dw1 + wf_dwords(w_end, prep[0].wide) <= two.total_dw)
                    dw1 += wf_dwords(w_end++, prep[0].wide);
                t_prep += ms(tp1, now());
                if (hm.on)
                    fprintf(stderr, "[lx host ms]   one chunk in two calls: pool %llu wavefronts, then %llu of %llu (slots: %llu of at most %llu; dwords %llu + %llu + %llu of %llu)\n",
                            (unsigned long long)pool_wf, (unsigned long long)(w_end - pool_wf), (unsigned long long)(nwf - pool_wf), (unsigned long long)(w_end * kWave),
                            (unsigned long long)two.cap_slots, (unsigned long long)two.dw0, (unsigned long long)two.ovf_dw, (unsigned long long)dw1, (unsigned long long)two.total_dw);
                if ((rc = enqueue_mq_second(0, pool_wf, w_end)))
                    return rc;
                if (!as_list)
                    clear_rows();
                w0 = w_end;
                c  = 1;
            }
        }
        for (;;)
        {
            if (w0 >= nwf)
            {
                // the pool's wavefronts are queued (or there are none): the streamed part of the plan is made now, beside their kernels
                if (stream_planned)
                    break;
                auto const tp0 = now();
                plan_stream();
                longest_first(w0, nwf);
                stream_planned = true;
                t_prep += ms(tp0, now());
                continue;
            }
            // the chunk's checkpoint slots must fit the trace budget (fused_impl leaves the sweep otherwise): every slot is sized
            // for the chunk's widest query and longest window, plus room for the int32 overflow slots of what the sweep may
            // decline -- the chunk ends where one more wavefront would break the budget
            uint64_t w1 = w0;
            uint64_t const pc = (uint64_t)lx::trace_cfg_panel(mq_cfg) / 8;
            // (compact codes + room for the int32 overflow slots of a few declined windows; int16 pairs where the chunk may run WIDE)
            bool const maybe_wide = mq_cfg == 1 && !lx::dev_aids().mq_no_wide && h->mq_decl_frac > 0.01;
            auto slot_bytes = [&](uint64_t pan, uint64_t maxs) -> uint64_t
            {
                uint64_t const steps = (maxs + 8 - 1 + 15) & ~15ull;
                return (pan + pc - 1) / pc * (maybe_wide ? lx::ckpt_slot_dwords(mq_cfg, (uint32_t)steps) + lx::ckpt_slot_dwords(mq_cfg, (uint32_t)steps) / 8
                                                         : lx::ckpt16_slot_dwords(mq_cfg, (uint32_t)steps) + lx::ckpt_slot_dwords(mq_cfg, (uint32_t)steps) / 8) * 4;
            };
            uint64_t range_now = 0;
            if (by_range)
            {
                auto const & cr = *ri->chunk_records;
                while (cr.cut_wf[range_now + 1] <= w0)
                    ++range_now;
                w1 = cr.cut_wf[range_now + 1]; // (the budgets were checked range by range when the mode was chosen)
            }
            uint64_t run_bytes = 0, run_q = 1, run_s = 1; // (slots by wavefront: what the chunk's wavefronts need, each for itself)
            while (!by_range && w1 < nwf && w1 - w0 < per_chunk)
            {
                if (!merge_pool && w0 < pool_end && w1 == pool_end)
                    break;
                uint64_t const b2 = run_bytes + kWave * slot_bytes(wf_pan[w1], wf_maxs[w1]);
                uint64_t const q2 = std::max<uint64_t>(run_q, wf_pan[w1]), s2 = std::max<uint64_t>(run_s, wf_maxs[w1]);
                if (w1 > w0 && (b2 > h->opt_trace_bytes || (w1 + 1 - w0) * kWave * (q2 * 8 + s2) > (8ull << 30)))
                    break;
                run_bytes = b2, run_q = q2, run_s = s2;
                ++w1;
            }
            int const L = c & 1;
            if (in_flight[L] && (rc = collect_mq(L)))
                return rc;
            prep[L].range = range_now;
            if ((rc = enqueue_mq(L, w0, w1)))
                return rc;
            if (!rows_cleared && !as_list)
                clear_rows();
            if (in_flight[L ^ 1] && (rc = collect_mq(L ^ 1)))
                return rc;
            w0 = w1;
            ++c;
        }
        for (int L : {c & 1, (c & 1) ^ 1})
            if (in_flight[L] && (rc = collect_mq(L)))
                return rc;
        while (!redo.empty()) // (one at a time; a WIDE chunk cannot ask again)
        {
            auto const r = redo.back();
            redo.pop_back();
            if (by_range) // (the range the chunk is: its records are made again behind the second sweep, and wait where the later ranges' stand)
                for (uint64_t k = 0; k < ri->chunk_records->n_ranges; ++k)
                    if (ri->chunk_records->cut_wf[k] == r.first)
                        prep[0].range = k;
            // (int16-pair slots are twice the codes': the range is run again in pieces that fit the slot budget -- a range whose records
            // are made chunk by chunk stays whole)
            for (uint64_t a = r.first; a < r.second;)
            {
                uint64_t b = a, bytes = 0;
                while (b < r.second)
                {
                    uint64_t const wb = wf_dwords(b, true) * 4;
                    if (!by_range && b > a && bytes + wb > h->opt_trace_bytes)
                        break;
                    bytes += wb;
                    ++b;
                }
                if ((rc = enqueue_mq(0, a, b, true)) || (rc = collect_mq(0)))
                    return rc;
                ++c;
                a = b;
            }
        }
        // the scores of every extension, in caller order (a device list's scores stay on the device as well: h->d_score_all)
        if (!dev_list)
        {
            auto const ts0 = now();
            LX_HIP(h, hipMemcpyAsync(h->p_score_all.ptr, h->d_score_all.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            LX_HIP(h, hipStreamSynchronize(h->stream));
            int32_t const * const sa = static_cast<int32_t const *>(h->p_score_all.ptr);
            parallel_ranges(n, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                for (uint64_t i = lo; i < hi; ++i)
                                {
                                    out_score[i] = sa[i];
                                    if (!as_list && out_hsp[i].n_ops == 0)
                                        out_hsp[i].score = sa[i];
                                }
                            });
            t_unpack += ms(ts0, now());
        }
        k0 = live;
    }
    while (k0 < live)
    {
        uint64_t k1 = std::min<uint64_t>(live, k0 + chunk_target);
        while (k1 < live && !newrun[k1]) // never cut a query's run
            ++k1;
        {
            // ... and never mix geometry classes (the list is class-major): cut where the class changes
            auto qclass = [](uint32_t lq) -> uint32_t { return lq <= 104 ? 0u : lq <= 152 ? 1u : lq <= 200 ? 2u : lq <= 208 ? 3u : 3u + (lq + 151) / 152; };
            uint32_t const c0 = qclass(ext[idx[k0]].q_len);
            if (qclass(ext[idx[k1 - 1]].q_len) != c0)
            {
                uint64_t lo = k0, hi = k1 - 1; // first position of another class: the classes ascend
                while (hi - lo > 1)
                {
                    uint64_t const mid = lo + (hi - lo) / 2;
                    (qclass(ext[idx[mid]].q_len) == c0 ? lo : hi) = mid;
                }
                k1 = hi;
                while (k1 > k0 + 1 && !newrun[k1])
                    --k1;
            }
        }
        int const L = c & 1;
        if (in_flight[L] && (rc = collect(L)))
            return rc;
        if ((rc = enqueue(L, k0, k1)))
            return rc;
        if (in_flight[L ^ 1] && (rc = collect(L ^ 1)))
            return rc;
        k0 = k1;
        ++c;
    }
    for (int L : {c & 1, (c & 1) ^ 1})
        if (in_flight[L] && (rc = collect(L)))
            return rc;
    hm.mark("pipeline"); // (every chunk's error word came back with its counts: collect())
    if (hm.on)
        fprintf(stderr, "[lx host ms]   pipeline of %d chunks: prepare %.1f, issue %.1f, wait for the GPU %.1f, unpack %.1f (lengths %.1f, offsets %.1f)\n", c, t_prep, t_issue,
                t_wait, t_unpack, t_u1, t_u2);
    if (as_list)
        h->xb_ops_total = ops_total;
    else
    {
        *out_ops       = h->ext_bytes.data();
        *out_ops_bytes = ops_total;
    }
    return LX_OK;
}

extern "C" {

int lx_extend_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                    lx_extension const * ext, uint64_t n, int32_t const * min_score, int32_t min_score_all, int32_t * out_score,
                    lx_hsp * out_hsp, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (out_ops)
        *out_ops = nullptr;
    if (out_ops_bytes)
        *out_ops_bytes = 0;
    if (n == 0)
        return LX_OK;
    if (!ext || !out_score || !out_hsp || !out_ops_off || !out_ops || !out_ops_bytes || (!q_res && q_bytes))
        return fail(h, LX_EINVAL, "NULL argument");
    if (n > 0xfffffff0ull / 2)
        return fail(h, LX_EINVAL, "at most 2^31 extensions per call");
    if (h->opt_band)
        return host_banded(h, slot, 2, q_res, q_bytes, s_res, s_bytes, ext, n, nullptr, min_score, min_score_all, out_score, out_hsp, nullptr,
                           nullptr, out_ops_off, out_ops, out_ops_bytes);
    return extend_pipeline(h, slot, q_res, q_bytes, s_res, s_bytes, ext, n, min_score, min_score_all, out_score, out_hsp, out_ops_off, out_ops,
                           out_ops_bytes, 0);
}

int lx_extend_batch_rle(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                        lx_extension const * ext, uint64_t n, int32_t const * min_score, int32_t min_score_all, int32_t * out_score,
                        lx_hsp * out_hsp, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (out_ops)
        *out_ops = nullptr;
    if (out_ops_bytes)
        *out_ops_bytes = 0;
    if (n == 0)
        return LX_OK;
    if (!ext || !out_score || !out_hsp || !out_ops_off || !out_ops || !out_ops_bytes || (!q_res && q_bytes))
        return fail(h, LX_EINVAL, "NULL argument");
    if (n > 0xfffffff0ull / 2)
        return fail(h, LX_EINVAL, "at most 2^31 extensions per call");
    if (h->opt_band)
        return fail(h, LX_EINVAL, "lx_extend_batch_rle: band mode returns column bytes only (lx_extend_batch)");
    return extend_pipeline(h, slot, q_res, q_bytes, s_res, s_bytes, ext, n, min_score, min_score_all, out_score, out_hsp, out_ops_off, out_ops,
                           out_ops_bytes, 1);
}

int lx_extend_batch_list(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                         lx_extension const * ext, uint64_t n, int32_t const * min_score, int32_t min_score_all, int32_t * out_score,
                         lx_survivor_list * out)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (out)
        *out = lx_survivor_list{};
    if (n == 0)
        return LX_OK;
    if (!ext || !out_score || !out || (!q_res && q_bytes))
        return fail(h, LX_EINVAL, "NULL argument");
    if (n > 0xfffffff0ull / 2)
        return fail(h, LX_EINVAL, "at most 2^31 extensions per call");
    if (h->opt_band)
        return fail(h, LX_EINVAL, "lx_extend_batch_list: band mode returns column bytes only (lx_extend_batch)");
    int const rc = extend_pipeline(h, slot, q_res, q_bytes, s_res, s_bytes, ext, n, min_score, min_score_all, out_score, nullptr, nullptr, nullptr,
                                   nullptr, 2);
    if (rc != LX_OK)
        return rc;
    out->count       = h->res_count;
    out->index       = reinterpret_cast<uint32_t const *>(h->res_index.data());
    out->hsp         = reinterpret_cast<lx_hsp const *>(h->res_hsp.data());
    out->codes_off   = reinterpret_cast<uint64_t const *>(h->res_off.data());
    out->codes       = h->ext_bytes.data();
    out->codes_bytes = h->res_count ? h->xb_ops_total : 0;
    return LX_OK;
}

} // extern "C"

// does the multi-query sweep serve this slot's lists at all (free packing: what protein lists are planned for)?
bool lxi::free_plan_applies(lx_handle const * h, int slot)
{
    lx_scoring const & sh = h->sc_host[slot];
    return h->opt_mq >= 1 && h->opt_pass2 == 2 && h->opt_f16 && h->trace_ok[slot] && h->b8_ok[slot] && -sh.gap_open <= lx::kC16MaxGap && sh.gap_open <= sh.gap_extend &&
           !h->opt_band;
}

// does lx_extend_batch* serve this slot's lists with the solo packing of the multi-query sweep (a byte profile per window)?
bool lxi::solo_plan_applies(lx_handle const * h, int slot)
{
    lx_scoring const & sh = h->sc_host[slot];
    return h->opt_mq >= 1 && h->opt_pass2 == 2 && h->opt_f16 && h->trace_ok[slot] && h->b8_ok[slot] && -sh.gap_open <= lx::kC16MaxGap && sh.gap_open <= sh.gap_extend &&
           !h->opt_band && lx::sweep_mq_lds_bytes(1, sh.alphabet_size + 1, -1) <= 20 * 1024;
}

// lx_extend_batch_list for the Level-2 driver on the device (lx_level2_host.cpp): query residues, window list and cut-offs are resident
int lxi::extend_list_resident(lx_handle * h, int slot, ResidentInput const & ri, lx_extension const * ext, uint64_t n, int32_t const * min_score,
                              int32_t * out_score, lx_survivor_list * out)
{
    *out = lx_survivor_list{};
    if (n == 0)
        return LX_OK;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (n > 0xfffffff0ull / 2)
        return fail(h, LX_EINVAL, "at most 2^31 extensions per call");
    int const rc = extend_pipeline(h, slot, nullptr, ri.q_bytes, nullptr, 0, ext, n, min_score, 0, out_score, nullptr, nullptr, nullptr, nullptr, 2, &ri);
    if (rc != LX_OK)
        return rc;
    out->count       = h->res_count;
    out->index       = reinterpret_cast<uint32_t const *>(h->res_index.data());
    out->hsp         = reinterpret_cast<lx_hsp const *>(h->res_hsp.data());
    out->codes_off   = reinterpret_cast<uint64_t const *>(h->res_off.data());
    out->codes       = h->ext_bytes.data();
    out->codes_bytes = h->res_count ? h->xb_ops_total : 0;
    return LX_OK;
}

extern "C" {

int lx_last_extend_stats(lx_handle const * h, uint64_t * out4)
{
    if (!h || !out4)
        return LX_EINVAL;
    std::memcpy(out4, h->xb_stats, sizeof(h->xb_stats));
    return LX_OK;
}

int lx_expand_ops(uint8_t const * codes, int32_t n_ops, uint8_t * out)
{
    if (!codes || !out || n_ops < 0)
        return LX_EINVAL;
    rle_expand(codes, n_ops, out);
    return LX_OK;
}

} // extern "C"

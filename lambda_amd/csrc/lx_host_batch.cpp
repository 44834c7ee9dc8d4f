// lx_host_batch.cpp -- the host-buffer entry points of the two passes by themselves and of the pre-extension filter (compiled with
// hipcc): lx_score_batch (pass 1 of _performAlignment, /root/reference/src/search_algo.hpp:1246), lx_align_batch (pass 2, :1296),
// band mode's plain path for score / align / extend, lx_prefilter_batch (seedLooksPromising, :426-481).  The list is binned by
// kernel geometry on the host threads; the device entry points these drive live in lx_api.cpp.  The fused step on host buffers --
// lx_extend_batch and its pipeline of chunks -- is lx_host.cpp.
#include "lx_internal.h"
#include "lx_level2.h"
using namespace lxi;

extern "C" {

int lx_score_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                   uint64_t s_bytes, lx_extension const * ext, uint64_t n, int32_t * out_score)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (n == 0)
        return LX_OK;
    if (!ext || !out_score || (!q_res && q_bytes))
        return fail(h, LX_EINVAL, "NULL argument");
    if (h->opt_band)
        return host_banded(h, slot, 0, q_res, q_bytes, s_res, s_bytes, ext, n, nullptr, nullptr, 0, out_score, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr);
    int rc = bind(h);
    if (rc)
        return rc;
    SubjectRef sref;
    if ((rc = resolve_subjects(h, s_res, s_bytes, sref)))
        return rc;
    s_bytes = sref.bytes;

    HostMarks hm("lx_score_batch");
    // ---- validate; order by (q_len, q_off, s_len): extensions of one query become adjacent (one LDS profile per
    // wavefront), similar lengths become adjacent (lanes of a wavefront run in lockstep; the reference sorts its
    // SIMD batches for the same reason, src/search_algo.hpp:1229-1235)
    if (n > 0xfffffff0ull)
        return fail(h, LX_EINVAL, "at most 2^32-16 extensions per call");
    auto before = [&](uint32_t a, uint32_t b)
    {
        lx_extension const &x = ext[a], &y = ext[b];
        if (x.q_len != y.q_len)
            return x.q_len < y.q_len;
        if (x.q_off != y.q_off)
            return x.q_off < y.q_off;
        if (x.s_len != y.s_len)
            return x.s_len < y.s_len;
        return a < b;
    };
    // (the loops over the list are spread over a few host threads, as in lx_extend_batch)
    HostPool::Call const in_flight_call;
    unsigned const nthreads = host_threads(n);
    struct Part
    {
        uint64_t live = 0, bad = ~0ull;
        uint32_t first_live = 0xffffffffu, last_live = 0xffffffffu;
        bool     ordered = true;
    };
    std::vector<Part> parts(nthreads);
    parallel_ranges(n, nthreads,
                    [&](unsigned t, uint64_t lo, uint64_t hi)
                    {
                        Part & pt = parts[t];
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
                                continue;
                            }
                            if (pt.last_live != 0xffffffffu && before((uint32_t)i, pt.last_live))
                                pt.ordered = false;
                            if (pt.first_live == 0xffffffffu)
                                pt.first_live = (uint32_t)i;
                            pt.last_live = (uint32_t)i;
                            ++pt.live;
                        }
                    });
    bool     ordered = true; // lambda hands its matches over sorted by query: then the sort is skipped
    uint64_t live    = 0;
    {
        uint32_t prev = 0xffffffffu;
        for (Part const & pt : parts)
        {
            if (pt.bad != ~0ull)
                return fail(h, LX_EINVAL, "extension %llu exceeds the residue buffers", (unsigned long long)pt.bad);
            ordered = ordered && pt.ordered;
            if (pt.first_live != 0xffffffffu)
            {
                if (prev != 0xffffffffu && before(pt.first_live, prev))
                    ordered = false;
                prev = pt.last_live;
            }
            live += pt.live;
        }
    }
    std::vector<uint32_t> idx(live);
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
    if (!ordered)
        std::sort(idx.begin(), idx.end(), before);
    hm.mark("validate+sort");

    // ---- bin query runs by kernel geometry.  A run whose padding to a whole number of wavefront slots wastes
    // <= 25 % goes to a "shared profile" launch (8-lane geometries allowed), the rest to per-extension profiles.
    // bin index: kind 0 = per-extension profiles, 1 = one profile per wavefront (int32): cfg * 2 + kind;
    // kind 2 = packed half (16 extensions of one query per wavefront): ncfg * 2 + pair geometry
    int const    ncfg  = lx::score_cfg_count();
    size_t const nbins = (size_t)ncfg * 2 + 9;
    struct Run
    {
        uint64_t first, count, pad; // positions in idx, padded slot count
        uint32_t bin;
        uint64_t out;               // first slot in the upload buffer (set once the bins are laid out)
    };
    std::vector<Run>      runs;
    std::vector<uint64_t> bin_slots(nbins, 0);
    std::vector<uint32_t> bin_maxq(nbins, 0);
    uint64_t              carry_pairs = 0;
    runs.reserve(idx.size() / 8 + 16);
    for (size_t k = 0; k < idx.size();)
    {
        size_t k1 = k + 1;
        while (k1 < idx.size() && ext[idx[k1]].q_off == ext[idx[k]].q_off && ext[idx[k1]].q_len == ext[idx[k]].q_len)
            ++k1;
        uint64_t const run  = k1 - k;
        uint32_t const qlen = ext[idx[k]].q_len;
        int            kind = 0, cfg = 0;
        uint64_t       pad  = run;
        int const      pcfg = h->opt_f16 ? lx::score_pair_cfg_for(qlen) : -1;
        uint64_t const pad16 = (run + 15) / 16 * 16;
        if (pcfg >= 0 && (pad16 - run) * 4 <= pad16)
        {
            kind = 2;
            cfg  = pcfg;
            pad  = pad16;
        }
        else if (pcfg < 0 && h->opt_f16 && (pad16 - run) * 4 <= pad16)
        {
            kind = 2; // wider than every packed-half geometry: the packed 16-bit integer kernel, panel by panel
            cfg  = kPair16Bin;
            pad  = pad16;
        }
        else
        {
            cfg          = pick_cfg(qlen, true);
            uint64_t grp = (uint64_t)lx::score_cfg_groups(cfg);
            pad          = (run + grp - 1) / grp * grp;
            kind         = (grp > 1 && (pad - run) * 4 <= pad) ? 1 : 0;
            if (kind == 0)
            {
                cfg  = pick_cfg(qlen, false);
                grp  = (uint64_t)lx::score_cfg_groups(cfg);
                pad  = (run + grp - 1) / grp * grp;
                kind = (grp > 1 && (pad - run) * 4 <= pad) ? 1 : 0;
                if (kind == 0)
                    pad = run;
            }
        }
        uint32_t const bin = kind == 2 ? (uint32_t)(ncfg * 2 + cfg) : (uint32_t)(cfg * 2 + kind);
        runs.push_back(Run{k, run, pad, bin, 0});
        bin_slots[bin] += pad;
        bin_maxq[bin] = std::max(bin_maxq[bin], qlen);
        if ((kind != 2 && (int)qlen > lx::score_cfg_panel(cfg)) || (kind == 2 && cfg == kPair16Bin))
            for (size_t j = k; j < k1; ++j)
                carry_pairs += ext[idx[j]].s_len;
        k = k1;
    }
    if (carry_pairs * 8 + 4096 > h->ws_grown)
        h->ws_grown = carry_pairs * 8 + 4096;

    struct Seg
    {
        int      cfg;
        uint64_t first, count;
        bool     multi, shared;
        int      pair_cfg;
    };
    std::vector<Seg>      segs;
    std::vector<uint64_t> bin_cursor(nbins, 0);
    uint64_t              total_slots = 0;
    for (size_t b = 0; b < nbins; ++b)
    {
        if (!bin_slots[b])
            continue;
        bin_cursor[b] = total_slots;
        if (b < (size_t)ncfg * 2)
        {
            int const cfg = (int)(b / 2);
            segs.push_back(Seg{cfg, total_slots, bin_slots[b], bin_maxq[b] > (uint32_t)lx::score_cfg_panel(cfg), (b & 1) == 1, -1});
        }
        else // the int32 fix-up launch over the same list uses the shared-profile geometry of the longest query
        {
            int const pair = (int)(b - (size_t)ncfg * 2), fcfg = pick_cfg(bin_maxq[b], true);
            segs.push_back(Seg{fcfg, total_slots, bin_slots[b], pair == kPair16Bin && bin_maxq[b] > (uint32_t)lx::score_cfg_panel(fcfg), true,
                               pair == kPair16Bin ? kPair16 : pair});
        }
        total_slots += bin_slots[b];
    }
    // every slot is written exactly once: straight into the upload buffer, no per-bin copies
    for (Run & r : runs)
    {
        r.out = bin_cursor[r.bin];
        bin_cursor[r.bin] += r.pad;
    }
    std::vector<lx_extension> & sorted = h->xb_ext; // (host staging that keeps its pages between calls)
    std::vector<uint32_t> &     perm   = h->xb_src;
    sorted.resize(total_slots);
    perm.resize(total_slots);
    parallel_ranges(runs.size(), nthreads,
                    [&](unsigned, uint64_t rlo, uint64_t rhi)
                    {
                        for (uint64_t ri = rlo; ri < rhi; ++ri)
                        {
                            Run const & r = runs[ri];
                            uint64_t    o = r.out;
                            for (uint64_t j = 0; j < r.count; ++j, ++o)
                            {
                                uint32_t const src = idx[r.first + j];
                                sorted[o]          = ext[src];
                                perm[o]            = src;
                            }
                            lx_extension dummy = ext[idx[r.first]]; // dummy slots keep one query per wavefront
                            dummy.s_len        = 0;
                            for (uint64_t j = r.count; j < r.pad; ++j, ++o)
                            {
                                sorted[o] = dummy;
                                perm[o]   = 0xffffffffu;
                            }
                        }
                    });
    if (sorted.empty())
        return LX_OK;

    hm.mark("bin");
    // ---- upload
    if ((rc = ensure(h, h->d_q, q_bytes + kSlack)) || (rc = ensure(h, h->d_ext, sorted.size() * sizeof(lx_extension))) ||
        (rc = ensure(h, h->d_out, sorted.size() * sizeof(int32_t))))
        return rc;
    if ((rc = prepare_workspace(h, h->stream)))
        return rc;
    if (q_bytes)
        LX_HIP(h, hipMemcpyAsync(h->d_q.ptr, q_res, q_bytes, hipMemcpyHostToDevice, h->stream));
    if (sref.upload)
        LX_HIP(h, hipMemcpyAsync(sref.dev, s_res, s_bytes, hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemcpyAsync(h->d_ext.ptr, sorted.data(), sorted.size() * sizeof(lx_extension), hipMemcpyHostToDevice,
                             h->stream));

    hm.mark("upload-issue");
    // ---- launch
    LX_HIP(h, hipEventRecord(h->ev0, h->stream));
    for (Seg const & seg : segs)
    {
        rc = launch_score_list(h, slot, h->d_q.ptr, sref.dev,
                               static_cast<lx_extension const *>(h->d_ext.ptr) + seg.first, seg.count,
                               static_cast<int32_t *>(h->d_out.ptr) + seg.first, seg.cfg, seg.multi, seg.shared,
                               h->stream, seg.pair_cfg);
        if (rc)
            return rc;
    }
    LX_HIP(h, hipEventRecord(h->ev1, h->stream));
    h->timed = true;

    // ---- download + unpermute
    std::vector<int32_t> & res = h->xb_score;
    res.resize(sorted.size());
    LX_HIP(h, hipMemcpyAsync(res.data(), h->d_out.ptr, res.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    hm.mark("launch");
    if ((rc = check_async_error(h)))
        return rc;
    hm.mark("wait");
    parallel_ranges(res.size(), nthreads,
                    [&](unsigned, uint64_t lo, uint64_t hi)
                    {
                        for (uint64_t k = lo; k < hi; ++k)
                            if (perm[k] != 0xffffffffu)
                                out_score[perm[k]] = res[k];
                    });
    hm.mark("unpermute");
    return LX_OK;
}


int lx_align_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                   uint64_t s_bytes, lx_extension const * ext, uint64_t n, int32_t const * known_score, lx_hsp * out_hsp,
                   uint8_t * out_ops, uint64_t const * ops_off)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (n == 0)
        return LX_OK;
    if (!ext || !out_hsp || !out_ops || !ops_off || (!q_res && q_bytes))
        return fail(h, LX_EINVAL, "NULL argument");
    if (n > 0xfffffff0ull)
        return fail(h, LX_EINVAL, "at most 2^32-16 extensions per call");
    if (h->opt_band)
        return host_banded(h, slot, 1, q_res, q_bytes, s_res, s_bytes, ext, n, known_score, nullptr, 0, nullptr, out_hsp, out_ops, ops_off,
                           nullptr, nullptr, nullptr);
    int rc = bind(h);
    if (rc)
        return rc;
    SubjectRef sref;
    if ((rc = resolve_subjects(h, s_res, s_bytes, sref)))
        return rc;
    s_bytes = sref.bytes;
    HostMarks hm("lx_align_batch");
    // ---- validate; find the runs of consecutive extensions that share their query slice (lambda's lists are grouped by
    // query).  If padding every run to a multiple of 4 slots costs <= 25 %, pass 2 runs the shared-profile geometries.
    uint64_t max_q = 1, max_s = 1, ops_bytes = 0, carry_pairs = 0, padded = 0, run = 0;
    for (uint64_t i = 0; i < n; ++i)
    {
        lx_extension const & x = ext[i];
        if (!lx_slice_ok(x.q_off, x.q_len, q_bytes) || !lx_slice_ok(x.s_off, x.s_len, s_bytes))
            return fail(h, LX_EINVAL, "extension %llu exceeds the residue buffers", (unsigned long long)i);
        max_q     = std::max<uint64_t>(max_q, x.q_len);
        max_s     = std::max<uint64_t>(max_s, x.s_len);
        ops_bytes = std::max<uint64_t>(ops_bytes, ops_off[i] + x.q_len + x.s_len);
        if ((int)x.q_len > lx::trace_cfg_panel(1)) // (the narrowest panel pass 2 may pick)
            carry_pairs += x.s_len;
        if (i > 0 && (x.q_off != ext[i - 1].q_off || x.q_len != ext[i - 1].q_len))
        {
            padded += (run + 3) / 4 * 4;
            run = 0;
        }
        ++run;
    }
    padded += (run + 3) / 4 * 4;
    bool const share = (padded - n) * 4 <= padded && padded <= 0xfffffff0ull; // (any query width: checkpoints carry across panels)
    uint64_t const slots = share ? padded : n;

    if (carry_pairs * 8 + 4096 > h->ws_grown)
        h->ws_grown = carry_pairs * 8 + 4096;
    if ((rc = ensure(h, h->d_q, q_bytes + kSlack)) ||
        (rc = ensure(h, h->d_ext, slots * sizeof(lx_extension))) || (rc = ensure(h, h->d_hsp, n * sizeof(lx_hsp))) ||
        (rc = ensure(h, h->d_ops, ops_bytes + 16)) || (rc = ensure(h, h->d_opsoff, n * sizeof(uint64_t))))
        return rc;
    if ((share && (rc = ensure(h, h->d_sel_src, slots * sizeof(uint32_t)))) ||
        (known_score && (rc = ensure(h, h->d_sel_score, slots * sizeof(int32_t)))))
        return rc;
    if ((rc = prepare_workspace(h, h->stream)))
        return rc;
    hm.mark("validate+alloc");

    // ---- slot list: the extensions in input order, every run followed by its padding slots (empty window, src = none);
    // filled on a few host threads into staging that keeps its pages between calls
    std::vector<lx_extension> & slot_ext   = h->xb_ext;
    std::vector<uint32_t> &     slot_src   = h->xb_src;
    std::vector<int32_t> &      slot_score = h->xb_min;
    slot_ext.clear();
    slot_src.clear();
    slot_score.clear();
    if (share)
    {
        std::vector<uint64_t> & grp = h->xb_grp; // (first extension, first slot) of every run + a sentinel
        grp.clear();
        uint64_t o = 0;
        for (uint64_t i = 0; i < n;)
        {
            uint64_t i1 = i + 1;
            while (i1 < n && ext[i1].q_off == ext[i].q_off && ext[i1].q_len == ext[i].q_len)
                ++i1;
            grp.push_back(i);
            grp.push_back(o);
            o += (i1 - i + 3) / 4 * 4;
            i = i1;
        }
        grp.push_back(n);
        grp.push_back(o);
        slot_ext.resize(slots);
        slot_src.resize(slots);
        if (known_score)
            slot_score.resize(slots);
        uint64_t const ngroups = grp.size() / 2 - 1;
        parallel_ranges(ngroups, host_threads(n),
                        [&](unsigned, uint64_t glo, uint64_t ghi)
                        {
                            for (uint64_t g = glo; g < ghi; ++g)
                            {
                                uint64_t const i0 = grp[2 * g], i1 = grp[2 * g + 2], o1 = grp[2 * g + 3];
                                uint64_t       oo = grp[2 * g + 1];
                                for (uint64_t j = i0; j < i1; ++j, ++oo)
                                {
                                    slot_ext[oo] = ext[j];
                                    slot_src[oo] = (uint32_t)j;
                                    if (known_score)
                                        slot_score[oo] = known_score[j];
                                }
                                lx_extension dummy = ext[i0];
                                dummy.s_len        = 0;
                                for (; oo < o1; ++oo)
                                {
                                    slot_ext[oo] = dummy;
                                    slot_src[oo] = 0xffffffffu;
                                    if (known_score)
                                        slot_score[oo] = 0;
                                }
                            }
                        });
    }
    hm.mark("slots");

    if (q_bytes)
        LX_HIP(h, hipMemcpyAsync(h->d_q.ptr, q_res, q_bytes, hipMemcpyHostToDevice, h->stream));
    if (sref.upload)
        LX_HIP(h, hipMemcpyAsync(sref.dev, s_res, s_bytes, hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemcpyAsync(h->d_ext.ptr, share ? slot_ext.data() : ext, slots * sizeof(lx_extension), hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemcpyAsync(h->d_opsoff.ptr, ops_off, n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    if (share)
        LX_HIP(h, hipMemcpyAsync(h->d_sel_src.ptr, slot_src.data(), slots * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    if (known_score)
        LX_HIP(h, hipMemcpyAsync(h->d_sel_score.ptr, share ? slot_score.data() : known_score, slots * sizeof(int32_t),
                                 hipMemcpyHostToDevice, h->stream));
    hm.mark("upload-issue");
    h->phase_ev.clear();
    h->ev_pool_used = 0;
    LX_HIP(h, hipEventRecord(h->ev0, h->stream));
    rc = align_dev_impl(h, slot, h->d_q.ptr, sref.dev, static_cast<lx::Extension const *>(h->d_ext.ptr), slots,
                        static_cast<lx::Hsp *>(h->d_hsp.ptr), static_cast<uint8_t *>(h->d_ops.ptr),
                        static_cast<uint64_t const *>(h->d_opsoff.ptr), h->stream, max_q, max_s, share ? 4 : 0,
                        share ? static_cast<uint32_t const *>(h->d_sel_src.ptr) : nullptr, nullptr,
                        known_score ? static_cast<int32_t const *>(h->d_sel_score.ptr) : nullptr);
    if (rc)
        return rc;
    LX_HIP(h, hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    hm.mark("launch");
    LX_HIP(h, hipMemcpyAsync(out_hsp, h->d_hsp.ptr, n * sizeof(lx_hsp), hipMemcpyDeviceToHost, h->stream));
    LX_HIP(h, hipMemcpyAsync(out_ops, h->d_ops.ptr, ops_bytes, hipMemcpyDeviceToHost, h->stream));
    hm.mark("download-issue");
    if ((rc = check_async_error(h)))
        return rc;
    hm.mark("wait");
    for (uint64_t i = 0; i < n; ++i)
        if (out_hsp[i].score < 0)
            return fail(h, LX_EOVERFLOW, "extension %llu could not be traced (workspace exhausted, or known_score is not its score)",
                        (unsigned long long)i);
    return LX_OK;
}

} // extern "C"

// ---- band mode on host buffers ---------------------------------------------------------------------------------------
// Band mode (LX_OPT_BAND) is a semantic option, not a fast path: it runs one int32 kernel geometry and direction bits for
// pass 2, so the host-buffer entry points skip the binning / grouping of their full-rectangle versions -- the list goes to
// the device as it is, the centres (lx_set_band_centres) with it.
//   what = 0: lx_score_batch, 1: lx_align_batch (caller's ops slots), 2: lx_extend_batch (ops slots of the handle)
int lxi::host_banded(lx_handle * h, int slot, int what, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                       lx_extension const * ext, uint64_t n, int32_t const * known_score, int32_t const * min_score,
                       int32_t min_score_all, int32_t * out_score, lx_hsp * out_hsp, uint8_t * caller_ops,
                       uint64_t const * caller_ops_off, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (n > 0xfffffff0ull / 2)
        return fail(h, LX_EINVAL, "at most 2^31 extensions per call");
    SubjectRef sref;
    if ((rc = resolve_subjects(h, s_res, s_bytes, sref)))
        return rc;
    s_bytes = sref.bytes;
    uint64_t max_q = 1, max_s = 1, total = 0;
    std::vector<uint64_t> & off = h->xb_off;
    off.resize(n + 1);
    for (uint64_t i = 0; i < n; ++i)
    {
        lx_extension const & x = ext[i];
        if (!lx_slice_ok(x.q_off, x.q_len, q_bytes) || !lx_slice_ok(x.s_off, x.s_len, s_bytes))
            return fail(h, LX_EINVAL, "extension %llu exceeds the residue buffers", (unsigned long long)i);
        max_q  = std::max<uint64_t>(max_q, x.q_len);
        max_s  = std::max<uint64_t>(max_s, x.s_len);
        off[i] = total;
        total += (uint64_t)x.q_len + x.s_len;
    }
    off[n] = total;
    if (!h->band_host.empty() && h->band_host.size() != n)
        return fail(h, LX_EINVAL, "lx_set_band_centres gave %llu centres, the call has %llu extensions",
                    (unsigned long long)h->band_host.size(), (unsigned long long)n);
    if ((rc = ensure(h, h->d_q, q_bytes + kSlack)) || (rc = ensure(h, h->d_ext, n * sizeof(lx_extension))) ||
        (rc = ensure(h, h->d_out, n * sizeof(int32_t))) || (rc = ensure(h, h->d_keep, n * sizeof(int32_t) + 64)) ||
        (!h->band_host.empty() && (rc = ensure(h, h->d_band, n * sizeof(int32_t)))))
        return rc;
    if (q_bytes)
        LX_HIP(h, hipMemcpyAsync(h->d_q.ptr, q_res, q_bytes, hipMemcpyHostToDevice, h->stream));
    if (sref.upload)
        LX_HIP(h, hipMemcpyAsync(sref.dev, s_res, s_bytes, hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemcpyAsync(h->d_ext.ptr, ext, n * sizeof(lx_extension), hipMemcpyHostToDevice, h->stream));
    if (!h->band_host.empty())
        LX_HIP(h, hipMemcpyAsync(h->d_band.ptr, h->band_host.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    struct Restore
    {
        lx_handle *     h;
        uint64_t        qlen, slen, run;
        int32_t const * band_dev;
        ~Restore()
        {
            h->opt_max_qlen  = qlen;
            h->opt_max_slen  = slen;
            h->opt_query_run = run;
            h->band_dev      = band_dev;
            // lx_set_band_centres applies to ONE host-buffer call (include/lambda_ext.h): consumed here, whatever the outcome --
            // a later call of another size must not fail on them, one of the same size must not reuse them silently
            (void)hipStreamSynchronize(h->stream);
            h->band_host.clear();
        }
    } const restore{h, h->opt_max_qlen, h->opt_max_slen, h->opt_query_run, h->band_dev};
    h->opt_max_qlen  = max_q;
    h->opt_max_slen  = max_s;
    h->opt_query_run = 0;
    h->band_dev      = h->band_host.empty() ? nullptr : static_cast<int32_t const *>(h->d_band.ptr);
    if (what == 0)
    {
        if ((rc = lx_score_batch_dev(h, slot, h->d_q.ptr, sref.dev, h->d_ext.ptr, n, h->d_out.ptr, h->stream)))
            return rc;
        LX_HIP(h, hipMemcpyAsync(out_score, h->d_out.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        return check_async_error(h);
    }
    if ((rc = ensure(h, h->d_hsp, n * sizeof(lx_hsp))) || (rc = ensure(h, h->d_ops, total + 16)) || (rc = ensure(h, h->d_opsoff, n * sizeof(uint64_t))))
        return rc;
    LX_HIP(h, hipMemcpyAsync(h->d_opsoff.ptr, off.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    if (what == 1)
    {
        if ((rc = prepare_workspace(h, h->stream, max_q > 160 ? n * ((max_s + 3) & ~3ull) : 0)))
            return rc;
        int32_t const * d_known = nullptr;
        if (known_score)
        {
            if ((rc = ensure(h, h->d_trace_score, n * sizeof(int32_t))))
                return rc;
            LX_HIP(h, hipMemcpyAsync(h->d_trace_score.ptr, known_score, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            d_known = static_cast<int32_t const *>(h->d_trace_score.ptr);
        }
        h->phase_ev.clear();
        h->ev_pool_used = 0;
        if ((rc = align_dev_impl(h, slot, h->d_q.ptr, sref.dev, static_cast<lx::Extension const *>(h->d_ext.ptr), n,
                                 static_cast<lx::Hsp *>(h->d_hsp.ptr), static_cast<uint8_t *>(h->d_ops.ptr),
                                 static_cast<uint64_t const *>(h->d_opsoff.ptr), h->stream, max_q, max_s, 0, nullptr, nullptr, d_known)))
            return rc;
    }
    else
    {
        uint64_t * const d_count = static_cast<uint64_t *>(h->d_keep.ptr);
        int32_t *        d_min   = nullptr;
        if (min_score)
        {
            if ((rc = ensure(h, h->d_keep, 16 + n * sizeof(int32_t))))
                return rc;
            d_min = reinterpret_cast<int32_t *>(static_cast<uint64_t *>(h->d_keep.ptr) + 2);
            LX_HIP(h, hipMemcpyAsync(d_min, min_score, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        }
        if ((rc = fused_impl(h, slot, h->d_q.ptr, sref.dev, h->d_ext.ptr, n, d_min, min_score_all, h->d_out.ptr, h->d_hsp.ptr, h->d_ops.ptr,
                             h->d_opsoff.ptr, static_cast<uint64_t *>(h->d_keep.ptr), h->stream, 3, false)))
            return rc;
        (void)d_count;
        LX_HIP(h, hipMemcpyAsync(out_score, h->d_out.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    }
    LX_HIP(h, hipMemcpyAsync(out_hsp, h->d_hsp.ptr, n * sizeof(lx_hsp), hipMemcpyDeviceToHost, h->stream));
    h->ext_ops.resize(total + 16);
    if (total)
        LX_HIP(h, hipMemcpyAsync(h->ext_ops.data(), h->d_ops.ptr, total, hipMemcpyDeviceToHost, h->stream));
    if ((rc = check_async_error(h)))
        return rc;
    for (uint64_t i = 0; i < n; ++i)
        if (out_hsp[i].score < 0)
            return fail(h, LX_EOVERFLOW, "extension %llu could not be traced", (unsigned long long)i);
    if (what == 1)
    {
        for (uint64_t i = 0; i < n; ++i) // into the caller's slots, same position inside the slot
            if (out_hsp[i].n_ops > 0)
                std::memcpy(caller_ops + caller_ops_off[i] + out_hsp[i].ops_shift, h->ext_ops.data() + off[i] + out_hsp[i].ops_shift,
                            (size_t)out_hsp[i].n_ops);
    }
    else
    {
        for (uint64_t i = 0; i < n; ++i)
            out_ops_off[i] = off[i];
        *out_ops       = h->ext_ops.data();
        *out_ops_bytes = total;
    }
    return LX_OK;
}

extern "C" {

// ---- pre-extension filter --------------------------------------------------------------------------------

int lx_prefilter_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                       uint64_t s_bytes, lx_seed const * seeds, uint64_t n, uint32_t seed_length, int32_t pre_scoring,
                       double pre_scoring_thresh, uint8_t * out_keep)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (n == 0)
        return LX_OK;
    if (!seeds || !out_keep || !q_res)
        return fail(h, LX_EINVAL, "NULL argument");
    SubjectRef sref;
    {
        int const rc0 = bind(h);
        if (rc0)
            return rc0;
        int const rc1 = resolve_subjects(h, s_res, s_bytes, sref);
        if (rc1)
            return rc1;
        s_bytes = sref.bytes;
    }
    static_assert(sizeof(lx_seed) == sizeof(lx::PrefilterSeed), "ABI mismatch");
    for (uint64_t i = 0; i < n; ++i)
    {
        lx_seed const & x = seeds[i];
        if (!lx_slice_ok(x.q_off, x.q_len, q_bytes) || !lx_slice_ok(x.s_off, x.s_len, s_bytes) || x.qry_end < x.qry_start || x.qry_end > x.q_len ||
            (uint64_t)x.subj_start + (x.qry_end - x.qry_start) > x.s_len)
            return fail(h, LX_EINVAL, "seed %llu out of range", (unsigned long long)i);
    }
    int rc = bind(h);
    if (rc)
        return rc;
    if ((rc = ensure(h, h->d_q, q_bytes + kSlack)) || (rc = ensure(h, h->d_seeds, n * sizeof(lx_seed))) || (rc = ensure(h, h->d_keep, n)))
        return rc;
    LX_HIP(h, hipMemcpyAsync(h->d_q.ptr, q_res, q_bytes, hipMemcpyHostToDevice, h->stream));
    if (sref.upload)
        LX_HIP(h, hipMemcpyAsync(sref.dev, s_res, s_bytes, hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemcpyAsync(h->d_seeds.ptr, seeds, n * sizeof(lx_seed), hipMemcpyHostToDevice, h->stream));
    lx::PrefilterParams p{};
    p.q_res              = static_cast<uint8_t const *>(h->d_q.ptr);
    p.s_res              = static_cast<uint8_t const *>(sref.dev);
    p.seeds              = static_cast<lx::PrefilterSeed const *>(h->d_seeds.ptr);
    p.n                  = n;
    p.sc                 = h->sc_dev[slot];
    p.seed_length        = seed_length;
    p.pre_scoring        = pre_scoring;
    p.pre_scoring_thresh = pre_scoring_thresh;
    p.out_keep           = static_cast<uint8_t *>(h->d_keep.ptr);
    LX_HIP(h, hipEventRecord(h->ev0, h->stream));
    LX_HIP(h, lx::launch_prefilter(p, h->stream));
    LX_HIP(h, hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    LX_HIP(h, hipMemcpyAsync(out_keep, h->d_keep.ptr, n, hipMemcpyDeviceToHost, h->stream));
    LX_HIP(h, hipStreamSynchronize(h->stream));
    return LX_OK;
}

} // extern "C"

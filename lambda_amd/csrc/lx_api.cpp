// lx_api.cpp -- host side of the C ABI declared in include/lambda_ext.h (compiled with hipcc).
//
// Owns: device selection, the HIP streams, device copies of the scoring schemes, the workspaces, HIP-event timing of the
// kernel sequence, the choice of kernel geometry and pass-2 mode, the device entry points (lx_*_dev) and the fused step.
// The host-buffer entry points (staging, binning, the chunk pipeline) are in lx_host.cpp; lx_internal.h is what they share.
// No DP arithmetic happens on the host and there is no CPU fallback: without a usable gfx950 device every entry point
// returns an error.
#include "lx_internal.h"

static_assert(sizeof(lx_extension) == sizeof(lx::Extension), "ABI mismatch");
static_assert(sizeof(lx_hsp) == sizeof(lx::Hsp), "ABI mismatch");
static_assert(sizeof(lx_extension) == 24, "ABI mismatch");


namespace
{
thread_local std::string g_create_error;
}

// every environment switch of the library (lx_aids.h has the table)
lx::DevAids const & lx::dev_aids()
{
    static DevAids const aids = []()
    {
        auto num = [](char const * name, long long dflt) -> long long
        {
            char const * e = getenv(name);
            return e && *e ? atoll(e) : dflt;
        };
        auto set = [](char const * name) { return getenv(name) != nullptr; };
        DevAids a{};
        a.host_threads    = (unsigned)std::max(0ll, num("LX_HOST_THREADS", 0));
        a.mq_no_wide      = set("LX_MQ_NO_WIDE");
        a.mq_merge_below  = (uint64_t)std::max(0ll, num("LX_MQ_MERGE_BELOW", 0));
        a.iterate_on_host = set("LX_ITERATE_ON_HOST");
        a.bt_tile_at      = (int)num("LX_BT_TILE_AT", 0);
        a.bt_refill_at    = (int)num("LX_BT_REFILL_AT", 0);
        a.l2_ranges       = (uint64_t)std::min<long long>(std::max(0ll, num("LX_L2_RANGES", 0)), 64);
        a.host_timing     = set("LX_HOST_TIMING");
        return a;
    }();
    return aids;
}

namespace lxi
{

int fail(lx_handle * h, int code, char const * fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h)
        h->error = buf;
    else
        g_create_error = buf;
    return code;
}

hipEvent_t pool_event(lx_handle * h)
{
    if (h->ev_pool_used == h->ev_pool.size())
    {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess)
            return nullptr;
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_pool_used++];
}




int ensure(lx_handle * h, DevBuf & b, size_t bytes)
{
    if (bytes <= b.cap)
        return LX_OK;
    size_t const had = b.cap;
    if (b.ptr)
    {
        LX_HIP(h, hipDeviceSynchronize()); // (kernels on a caller's stream may still read the old block)
        LX_HIP(h, hipFree(b.ptr));
        b.ptr = nullptr;
        b.cap = 0;
    }
    // (room to grow into, less of it for the large blocks: fresh device memory costs 40 ms per GB)
    size_t const want = bytes + (bytes >= ((size_t)1 << 30) ? bytes / 16 : bytes / 4) + 4096;
    if (lx::dev_aids().host_timing && want >= (64u << 20)) // (LX_HOST_TIMING: which buffer a call still had to grow -- what lx_reserve did not cover)
        fprintf(stderr, "[lx host ms]   device buffer %+ld in the handle grows from %.1f to %.1f MB\n", (long)(reinterpret_cast<char *>(&b) - reinterpret_cast<char *>(h)),
                (double)had / 1e6, (double)want / 1e6);
    LX_HIP(h, hipMalloc(&b.ptr, want));
    b.cap = want;
    return LX_OK;
}

int bind(lx_handle * h)
{
    LX_HIP(h, hipSetDevice(h->device));
    return LX_OK;
}


// Smallest panel that holds the query; 8-lane geometries need one shared profile per wavefront (8 profile slots
// per wavefront would not fit the LDS budget), so without sharing only the 16/32/64-lane geometries are used.
// LDS a wavefront of the packed-half kernel may spend on two query profiles (one per half wavefront: query runs of 8)
// (24 KiB; protein profiles too: 12.8 vs 14.0 ms (pass 1), 16.6 vs 19.1 ms (sweep) for runs of 8)
size_t pair_lds_limit()
{
    return 24 * 1024;
}

int pick_cfg(uint32_t qlen, bool shared)
{
    if (shared)
    {
        if (qlen <= 64)
            return 1;
        if (qlen <= 104)
            return 4;
        if (qlen <= 128)
            return 5;
        if (qlen <= 152)
            return 6;
    }
    if (qlen <= 64 && !shared)
        return 1; // 64 columns: 8 slots of a 64-column profile are small enough
    if (qlen <= 160)
        return 0;
    if (qlen <= 208)
        return 7;
    if (qlen <= 256)
        return 8;
    // Longer queries: several panels of a 16-lane geometry beat one wide panel of the 32- / 64-lane ones (400 aa x 442:
    // (16,13) x 2 panels 4.4 TCUPS, (16,16) x 2 3.8, (16,10) x 3 3.8, (32,10) x 2 2.6, (64,10) 2.3 -- the wide groups pay
    // for their long skew and cross-row shifts).  Pick the geometry with the least padded work, weighted by the time
    // each took per padded column in that measurement.
    struct Cand
    {
        int    cfg, panel;
        double cost;
    };
    static constexpr Cand cands[] = {{8, 256, 0.0581}, {7, 208, 0.0619}, {0, 160, 0.0621}};
    int    best      = 8;
    double best_cost = 1e30;
    for (Cand const & c : cands)
    {
        double const cost = (double)((qlen + c.panel - 1) / c.panel * c.panel) * c.cost;
        if (cost < best_cost)
        {
            best_cost = cost;
            best      = c.cfg;
        }
    }
    return best;
}

// Checkpoint geometry (trace cfg 1 = (8,19), 2 = (16,13)) for a query of max_q columns: one panel if it fits, else the
// panel count x width x measured time per padded column that is least (int32 kernels: 0.070 vs 0.062 per column;
// packed16 = the sweep of lx_score_i16.hip will run).
int ckpt_cfg_for(uint64_t max_q, bool packed16)
{
    uint64_t const p1 = (uint64_t)lx::trace_cfg_panel(1), p2 = (uint64_t)lx::trace_cfg_panel(2);
    if (max_q <= p1)
        return 1;
    if (max_q <= p2)
        return 2;
    // (the packed 16-bit sweep is bound by its checkpoint bytes: the 19-column strips of (8,19) store fewer boundary
    // columns -- 400 aa: 0.0134 ms per padded column against 0.0156 for (16,13) with int16-pair slots; with compact codes,
    // which only the (8,19) panels have, 0.0108)
    double const f1 = packed16 ? 0.0108 : 0.070, f2 = packed16 ? 0.0156 : 0.062;
    double const c1 = (double)((max_q + p1 - 1) / p1 * p1) * f1, c2 = (double)((max_q + p2 - 1) / p2 * p2) * f2;
    return c1 < c2 ? 1 : 2;
}

// Multi-query sweep (lx_sweep_mq.hip): checkpoint geometry for queries of up to max_q columns -- trace cfg 3 = (8,13),
// 5 = (8,11), 1 = (8,19), as many panels as the query needs: the one with the least padded work (columns swept x instructions
// per column: 3.75 per cell of the recurrence + ~12 per step and lane spread over the strip's columns).  lx_host.cpp deals
// ragged lists to classes with the same function, so a chunk's geometry is the one its class was formed for.
int mq_cfg_for(uint64_t max_q)
{
    int    best = 1;
    double best_cost = 1e30;
    for (int cfg : {1, 5, 3})
    {
        uint64_t const panel = (uint64_t)lx::trace_cfg_panel(cfg);
        double const   C     = (double)panel / 8.0;
        double const   cost  =(double)((std::max<uint64_t>(max_q, 1) + panel - 1) / panel * panel) * (3.75 * C + 12.0) / C;
        if (cost < best_cost - 1e-9)
        {
            best_cost = cost;
            best      = cfg;
        }
    }
    return best;
}

// the device's error word (d_ws_top[1]) as a return code
int error_for_flag(lx_handle * h, uint32_t flag)
{
    if (flag == 1)
        return fail(h, LX_EOVERFLOW, "multi-panel carry workspace exhausted (raise LX_OPT_WORKSPACE_BYTES)");
    if (flag == 2)
        return fail(h, LX_ESTATE, "LX_OPT_QUERY_RUN promise violated: extensions of one wavefront use different queries");
    if (flag == 3)
        return fail(h, LX_EOVERFLOW, "an extension exceeds the trace slot bounds (LX_OPT_MAX_QLEN / LX_OPT_MAX_SLEN too small, or a subject window beyond the pass-2 limit)");
    if (flag == 4)
        return fail(h, LX_EOVERFLOW, "single sweep: no checkpoint slot left for an extension the packed-half kernel declined "
                                     "(raise LX_OPT_TRACE_BYTES, or set LX_OPT_PASS2_MODE to 1)");
    if (flag != 0)
        return fail(h, flag == 5 ? LX_EOVERFLOW : LX_EHIP, "device reported error flag %u", flag);
    return LX_OK;
}

int check_async_error(lx_handle * h)
{
    uint32_t flags[2] = {0, 0};
    LX_HIP(h, hipMemcpyAsync(flags, h->d_ws_top, sizeof(flags), hipMemcpyDeviceToHost, h->stream));
    LX_HIP(h, hipStreamSynchronize(h->stream));
    return error_for_flag(h, flags[1]);
}

// One kernel sequence for a device-resident extension list whose queries all fit geometry `cfg`
// (or need the multi-panel path when wider).

int launch_score_list(lx_handle * h, int slot, void const * d_q, void const * d_s, void const * d_ext, uint64_t n,
                      void * d_out, int cfg, bool multi, bool shared, hipStream_t stream, int pair_cfg, int pair_share)
{
    lx::ScoreParams p{};
    p.q_res          = static_cast<uint8_t const *>(d_q);
    p.s_res          = static_cast<uint8_t const *>(d_s);
    p.ext            = static_cast<lx::Extension const *>(d_ext);
    p.n              = n;
    p.sc             = h->sc_dev[slot];
    p.out_score      = static_cast<int32_t *>(d_out);
    p.ws             = static_cast<int32_t *>(h->d_ws.ptr);
    p.ws_top         = h->d_ws_top;
    p.ws_cap         = (uint32_t)std::min<uint64_t>(h->d_ws.cap / 8, 0xffffffffu);
    p.err            = reinterpret_cast<int32_t *>(h->d_ws_top + 1);
    p.shared_profile = shared ? 1 : 0;
    p.nrows          = ((h->sc_host[slot].alphabet_size + 1 + 3) / 4) * 4;
    p.fixup          = 0;
    p.pair_share     = pair_share;
    char buf[128];
    if (h->opt_band)
    {
        // band mode: the int32 kernel of the generic geometry, any query width (the packed kernels carry no band code)
        p.band           = (int32_t)h->opt_band;
        p.band_diag      = h->band_dev;
        // query runs of a multiple of 8 whose queries fit 152 columns: (8,19), one LDS profile per wavefront
        bool const narrow = shared && cfg == 6 && !multi;
        p.shared_profile  = narrow ? 1 : 0;
        LX_HIP(h, lx::launch_score(narrow ? 6 : 0, p, true, stream));
        snprintf(buf, sizeof(buf), "lx::score_kernel<%s,band> (band mode, +-%d diagonals)", narrow ? "8,19,false" : "16,10,true", p.band);
        h->last_kernel = buf;
        return LX_OK;
    }
    if (pair_cfg == kPair16)
    {
        // queries wider than the packed-half geometries: packed 16-bit integers (lx_score_i16.hip), panel by panel; what
        // fails its range test is left to the int32 kernel, which starts with an empty carry workspace
        LX_HIP(h, lx::launch_score_pair16(p, stream));
        LX_HIP(h, hipMemsetAsync(h->d_ws_top, 0, sizeof(uint32_t), stream));
        p.fixup = 1;
        LX_HIP(h, lx::launch_score(cfg, p, multi, stream));
        snprintf(buf, sizeof(buf), "lx::sweep_pair16_kernel<8,19,true,false> (+ int32 fix-up lx::score_kernel<%d,%d,%s>)",
                 64 / lx::score_cfg_groups(cfg), lx::score_cfg_panel(cfg) * lx::score_cfg_groups(cfg) / 64, multi ? "true" : "false");
    }
    else if (pair_cfg >= 0)
    {
        // packed-half kernel first (two extensions per lane group); wavefronts whose score bound does not fit half
        // precision leave the sentinel -1, which the int32 kernel then resolves in fix-up mode
        LX_HIP(h, lx::launch_score_pair(pair_cfg, p, stream));
        p.fixup = 1;
        LX_HIP(h, lx::launch_score(cfg, p, multi, stream));
        snprintf(buf, sizeof(buf), "lx::score_pair_kernel<%d,%d> (+ int32 fix-up lx::score_kernel<%d,%d,%s>)",
                 lx::score_pair_cfg_group(pair_cfg), lx::score_pair_cfg_cols(pair_cfg), 64 / lx::score_cfg_groups(cfg),
                 lx::score_cfg_panel(cfg) * lx::score_cfg_groups(cfg) / 64, multi ? "true" : "false");
    }
    else
    {
        LX_HIP(h, lx::launch_score(cfg, p, multi, stream));
        snprintf(buf, sizeof(buf), "lx::score_kernel<%d,%d,%s>%s", 64 / lx::score_cfg_groups(cfg),
                 lx::score_cfg_panel(cfg) * lx::score_cfg_groups(cfg) / 64, multi ? "true" : "false",
                 shared ? " shared-profile" : "");
    }
    h->last_kernel = buf;
    return LX_OK;
}

// pairs_hint: carry pairs (8 bytes each) the call can need at most -- one per subject row of every extension whose
// query is wider than a panel; the workspace grows to that (the device cannot grow it, it can only report)
int prepare_workspace(lx_handle * h, hipStream_t stream, uint64_t pairs_hint)
{
    uint64_t const want = std::min<uint64_t>(pairs_hint, 0xfffffff0ull) * 8 + 4096;
    if (pairs_hint != 0 && want > h->ws_grown)
        h->ws_grown = want;
    int rc = ensure(h, h->d_ws, std::max(h->opt_ws_bytes, h->ws_grown));
    if (rc)
        return rc;
    LX_HIP(h, hipMemsetAsync(h->d_ws_top, 0, 2 * sizeof(uint32_t), stream));
    return LX_OK;
}


} // namespace lxi
using namespace lxi;

int lxi::resolve_subjects(lx_handle * h, uint8_t const * s_res, uint64_t s_bytes, SubjectRef & out)
{
    if (!s_res && s_bytes == 0 && h->db_bytes)
    {
        out.dev   = h->d_db.ptr;
        out.bytes = h->db_bytes;
        return LX_OK;
    }
    if (!s_res && s_bytes)
        return fail(h, LX_EINVAL, "NULL argument");
    int rc = ensure(h, h->d_s, s_bytes + kSlack);
    if (rc)
        return rc;
    out.dev    = h->d_s.ptr;
    out.bytes  = s_bytes;
    out.upload = s_bytes != 0;
    return LX_OK;
}


// ---- pass 2 ------------------------------------------------------------------------------------------

// Runs pass 2 over a device-resident list of `n` extension slots, in chunks sized to the trace budget.
// src / d_count are set by the fused path (slots compacted by launch_select): results are then written to
// out_hsp[src[slot]] / ops_off[src[slot]] and slots beyond *d_count are skipped on the device.
int lxi::align_dev_impl(lx_handle * h, int slot, void const * d_q, void const * d_s, lx::Extension const * d_ext,
                          uint64_t n, lx::Hsp * d_hsp, uint8_t * d_ops, uint64_t const * d_ops_off, hipStream_t stream,
                          uint64_t max_q, uint64_t max_s, int share_slots, uint32_t const * d_src,
                        uint64_t const * d_count, int32_t const * d_score_in, bool by_pos, uint64_t ops_stride)
{
    if (!h->trace_ok[slot])
        return fail(h, LX_EINVAL, "pass 2 needs every (matrix entry - gap_extend) in [-31, 31]");
    if ((reinterpret_cast<uintptr_t>(d_q) | reinterpret_cast<uintptr_t>(d_s)) & 15)
        return fail(h, LX_EINVAL, "pass 2 reads residues in aligned 16-byte groups: the residue buffers must be 16-byte aligned");
    if (max_s > (uint64_t)lx::kMaxTraceRows)
        return fail(h, LX_EINVAL, "pass 2 supports subject windows up to %d residues (got %llu)", lx::kMaxTraceRows, (unsigned long long)max_s);
    // share_slots = every aligned block of that many slots holds one query (0: no such guarantee).  The 8-lane
    // geometry puts 8 extensions in a wavefront and needs blocks of >= 4 (two LDS profiles per wavefront).
    int smax_entry = 0;
    for (int a = 0; a < h->sc_host[slot].alphabet_size; ++a)
        for (int b = 0; b < h->sc_host[slot].alphabet_size; ++b)
            smax_entry = std::max<int>(smax_entry, h->sc_host[slot].matrix[a * LX_ALPH + b]);
    // checkpoint mode (lx_ckpt.hip): shared-profile geometries (8,19) / (16,13), scores that fit int16; queries wider
    // than 208 columns take several (16,13) panels
    bool const ckpt = !h->opt_band && h->opt_pass2 >= 1 && share_slots >= 2 && (uint64_t)smax_entry * std::min(max_q, max_s) < 32000 && max_s <= 65535; // (longer windows: direction bits)
    // Direction bits beyond one panel: the 16-lane geometry that pads the query less ((16,13) needs the shared profile).
    auto padded = [&](int c) { return (max_q + lx::trace_cfg_panel(c) - 1) / lx::trace_cfg_panel(c) * lx::trace_cfg_panel(c); };
    int const cfg = (h->opt_band && !(share_slots >= 4 && max_q <= (uint64_t)lx::trace_cfg_panel(1))) ? 0 // (band mode: (8,19) or generic)
                    : (share_slots >= 2 && max_q <= (uint64_t)lx::trace_cfg_panel(1)) ? 1
                    : (share_slots >= 2 && max_q <= (uint64_t)lx::trace_cfg_panel(2)) ? 2
                    : ckpt                                                             ? ckpt_cfg_for(max_q)
                    : (share_slots >= 4 && padded(2) < padded(0))                      ? 2
                                                                                      : 0;
    int const G = lx::trace_cfg_group(cfg), P = lx::trace_cfg_panel(cfg), W = lx::trace_cfg_words(cfg);
    uint32_t const panels_cap = (uint32_t)std::max<uint64_t>(1, (max_q + P - 1) / P);
    uint32_t const steps_cap  = (uint32_t)((max_s + G - 1 + 15) & ~15ull); // multiple of the trace layout block
    uint64_t const stride     = ckpt ? (uint64_t)panels_cap * lx::ckpt_slot_dwords(cfg, steps_cap) : (uint64_t)panels_cap * steps_cap * G * W; // uint32 entries
    uint64_t const per_ext    = stride * 4;
    // The forward kernel finds the end cell cheaply when it knows each extension's best score; the fused path hands
    // over pass 1's scores, a stand-alone traceback call computes them first (a fraction of the traceback's cost).
    if (!d_score_in)
    {
        int rc0;
        if ((rc0 = ensure(h, h->d_trace_score, n * sizeof(int32_t))))
            return rc0;
        int const  scfg  = pick_cfg((uint32_t)std::min<uint64_t>(max_q, 0xffffffffu), false);
        bool const multi = max_q > (uint64_t)lx::score_cfg_panel(scfg);
        if ((rc0 = launch_score_list(h, slot, d_q, d_s, d_ext, n, h->d_trace_score.ptr, scfg, multi, false, stream)))
            return rc0;
        d_score_in = static_cast<int32_t const *>(h->d_trace_score.ptr);
    }
    // One trace buffer, forward kernel and backtrace of a chunk one after the other on `stream`: a chunk may use the whole budget --
    // as few launches (and kernel tails) as the budget allows.  (A backtrace on a second stream beside the next chunk's forward
    // kernel was measured on config 2: 46.9 against 46.5 ms per step -- both kernels saturate the chip -- and is gone.)  In the fused
    // path `n` is the capacity of the survivor list; launches beyond the device-side count exit at once.
    uint64_t chunk = std::max<uint64_t>(1, h->opt_trace_bytes / std::max<uint64_t>(per_ext, 1));
    chunk          = std::min<uint64_t>(chunk, n + 8);
    // a trace buffer that is already there and holds a fair chunk is used as it is: the fused step sizes this list for the
    // worst case (every extension survives), and growing a buffer of tens of GB costs seconds -- the adaptive mode-1 step
    // after a run of single sweeps would pay that once; chunks beyond the device-side count exit at once
    if (d_count && h->d_trace.cap / std::max<uint64_t>(per_ext, 1) >= 65536)
        chunk = std::min<uint64_t>(chunk, h->d_trace.cap / per_ext / 8 * 8);
    chunk                      = std::max<uint64_t>(8, (chunk + 7) / 8 * 8);
    int rc;
    if ((rc = ensure(h, h->d_trace, chunk * per_ext)) || (rc = ensure(h, h->d_ends, chunk * sizeof(lx::EndCell))))
        return rc;
    for (uint64_t c0 = 0; c0 < n; c0 += chunk)
    {
        lx::TraceParams p{};
        p.q_res          = static_cast<uint8_t const *>(d_q);
        p.s_res          = static_cast<uint8_t const *>(d_s);
        p.ext            = d_ext + c0;
        p.n              = std::min<uint64_t>(chunk, n - c0);
        p.sc             = h->sc_dev[slot];
        p.trace          = static_cast<uint32_t *>(h->d_trace.ptr);
        p.slot_stride    = stride;
        p.steps_cap      = steps_cap;
        p.panels_cap     = panels_cap;
        p.ends           = static_cast<lx::EndCell *>(h->d_ends.ptr);
        p.out_hsp        = (d_src && !by_pos) ? d_hsp : d_hsp + c0;
        p.out_ops        = d_ops;
        p.ops_off        = !d_ops_off ? nullptr : (d_src && !by_pos) ? d_ops_off : d_ops_off + c0;
        p.ops_stride     = ops_stride;
        if (!d_ops_off && ((d_src && !by_pos) ? false : c0 != 0)) // uniform slots are addressed by the index inside the chunk
            p.out_ops = d_ops + c0 * ops_stride;
        p.out_by_pos     = by_pos ? 1 : 0;
        p.src            = d_src ? d_src + c0 : nullptr;
        p.score_in       = d_score_in + c0;
        p.count_ptr      = d_count;
        p.chunk_start    = c0;
        p.ws             = static_cast<int32_t *>(h->d_ws.ptr);
        p.ws_top         = h->d_ws_top;
        p.ws_cap         = (uint32_t)std::min<uint64_t>(h->d_ws.cap / 8, 0xffffffffu);
        p.err            = reinterpret_cast<int32_t *>(h->d_ws_top + 1);
        p.nrows          = ((h->sc_host[slot].alphabet_size + 1 + 3) / 4) * 4;
        p.bs_match_rule  = (int32_t)h->opt_bs_rule;
        p.work_counter   = h->d_ws_top + 5;
        p.band           = (int32_t)h->opt_band;
        p.band_diag      = h->band_dev ? (d_src ? h->band_dev : h->band_dev + c0) : nullptr; // indexed like the caller's list
        p.shared_profile = (h->opt_band && cfg == 0) ? 0 : share_slots;
        p.cfg            = cfg;
        if (panels_cap > 1) // each chunk starts with an empty carry workspace
            LX_HIP(h, hipMemsetAsync(h->d_ws_top, 0, sizeof(uint32_t), stream));
        PhaseTimer ptf(h, stream, 2);
        LX_HIP(h, ckpt ? lx::launch_ckpt_forward(p, stream) : lx::launch_trace_forward(p, stream));
        ptf.close();
        PhaseTimer ptb(h, stream, 3);
        LX_HIP(h, ckpt ? lx::launch_ckpt_backtrace(p, stream) : lx::launch_backtrace(p, stream));
        ptb.close();
        {
            char buf[96];
            if (ckpt)
                snprintf(buf, sizeof(buf), "lx::ckpt_forward_kernel<%d,%d>", G, P / G);
            else
                snprintf(buf, sizeof(buf), "lx::trace_forward_kernel<%d,%d,%s>", G, P / G, panels_cap > 1 ? "true" : "false");
            h->last_trace_kernel = buf;
        }
    }
    return LX_OK;
}


// ---- fused: pass 1 -> survivor selection -> pass 2, all on the device ------------------------------------

// phases: 1 = pass 1 (or the sweep) + selection, 2 = pass 2 (or the sweep's backtrace), 3 = both.  by_pos: records and
// ops offsets are indexed by the position in the survivor list instead of by extension (the host entry point assigns
// compact ops offsets between the two phases and downloads only the survivors' records).

// after the backtrace (records and slots by list position): the survivors' ops as run-length codes, the list's original
// indices next to them
static int fused_pack(lx_handle * h, FusedExtra const * fx, uint64_t cap, void * d_out_hsp, void * d_out_ops, void const * d_ops_off,
                      void * d_out_count, hipStream_t stream, bool packed_already = false)
{
    if (!fx || !fx->d_rle)
        return LX_OK;
    if (fx->d_src_out)
        LX_HIP(h, hipMemcpyAsync(fx->d_src_out, h->d_sel_src.ptr, cap * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
    if (packed_already) // (the checkpoint backtrace emits the codes itself)
        return LX_OK;
    lx::PackParams pp{};
    pp.hsp        = static_cast<lx::Hsp *>(d_out_hsp);
    pp.ops        = static_cast<uint8_t const *>(d_out_ops);
    pp.ops_off    = static_cast<uint64_t const *>(d_ops_off);
    pp.ops_stride = fx->ops_stride;
    pp.src        = static_cast<uint32_t const *>(h->d_sel_src.ptr);
    pp.count_ptr  = static_cast<uint64_t const *>(d_out_count);
    pp.n          = cap;
    pp.rle        = fx->d_rle;
    pp.rle_top    = fx->d_rle_top;
    pp.rle_cap    = fx->rle_cap;
    pp.rle_len    = fx->d_rle_len;
    pp.err        = reinterpret_cast<int32_t *>(h->d_ws_top + 1);
    LX_HIP(h, hipMemsetAsync(fx->d_rle_top, 0, sizeof(unsigned long long), stream));
    LX_HIP(h, lx::launch_rle_pack(pp, stream));
    return LX_OK;
}

// ---- the ONE place that decides how a fused step runs: which sweep kernel family (if any), geometry, panel count, slot
// layout, queries per wavefront.  A pure function of the scheme's facts and the handle's options -- fused_impl follows it, and
// lx_plan_step() shows it to tests/test_plan.py, which walks it over query widths, run lengths and schemes on the CPU.
lxi::StepPlan lxi::plan_step(SchemeFacts const & sc, StepOptions const & o)
{
    bool const shared = o.query_run != 0 && o.query_run % 8 == 0;

    // Single sweep (LX_OPT_PASS2_MODE = 2): the checkpoint forward kernel runs once over ALL extensions -- it is pass 1
    // and the forward half of pass 2 at the same time -- and the backtrace reads the checkpoints of the survivors in
    // place.  Needs the checkpoints of the whole batch inside the trace budget and a shared-profile geometry.
    bool sweep = false;
    int  sweep_cfg = 0;
    uint32_t sweep_steps = 0, sweep_panels = 1;
    uint64_t sweep_stride = 0;   // uint32 per slot of the batch
    uint64_t sweep_stride32 = 0; // ... of an int16-pair slot (the whole batch's, or the overflow area's)
    uint64_t ovf_cap = 0;
    int      sweep_share = 0;
    bool     half_sweep = false, may_decline = true, wide_compact = false, mq_wide = false;
    int const nrows_sc = ((sc.alph + 1 + 3) / 4) * 4;
    // Multi-query sweep (lx_sweep_mq.hip): query runs of 4 or 8 (2 / 4 lane groups per LDS profile) -- what lx_extend_batch
    // makes of a ragged list --, or any multiple of 4 when LX_OPT_MQ_SWEEP = 2 asks for it.  Needs byte profiles (no
    // substitution dearer than a gap's first character) and compact codes (that character costs at most 31).
    bool mq = false;
    {
        bool const gaps_ok = -sc.gap_open <= lx::kC16MaxGap && sc.gap_open <= sc.gap_extend;
        uint64_t const run = o.query_run;
        // (runs of 8: the packed-half sweep with one profile per half wavefront keeps its occupancy while both profiles fit
        // ~13 KB -- the small alphabets: configs[2] 29.1 against 31.8 ms, configs[4] 7.4 against 8.1 -- and loses it beyond)
        bool const half8_cheap = 2 * lx::score_pair_profile_bytes(0, nrows_sc) + 64 * 8 * 4 <= 13 * 1024;
        // (... and for queries beyond 208 columns a run that is a multiple of 8 but not of 16 has no packed sweep with compact
        // codes of its own: the multi-query one serves it)
        bool const odd8    = run % 8 == 0 && run % 16 != 0 && run != 0;
        // (run 2 = the free packing: pairs of one query, at most four queries per wavefront -- only this sweep serves it;
        // run 1 = the solo packing: no promise, a byte profile per window, where 16 of them fit a wavefront's LDS share: the
        // alphabets of at most 6 rows -- nucleotides, bisulfite)
        bool const solo_fits = lx::sweep_mq_lds_bytes(1, sc.alph + 1, -1) <= 20 * 1024;
        bool const wanted  = run == 1 ? (o.mq >= 1 && solo_fits) : run == 2 ? o.mq >= 1 : o.mq == 2 ? (run != 0 && run % 4 == 0) : o.mq == 1 ? (run == 4 || (odd8 && (!half8_cheap || o.max_qlen > 208))) : false;
        mq = wanted && o.pass2 == 2 && o.f16 && sc.trace_ok && sc.b8_ok && gaps_ok && !o.band;
    }
    if (mq)
    {
        int const smax_entry = sc.smax_entry;
        sweep_cfg    = o.mq_cfg_call ? o.mq_cfg_call : mq_cfg_for(o.max_qlen);
        sweep_panels = (uint32_t)std::max<uint64_t>(1, (o.max_qlen + lx::trace_cfg_panel(sweep_cfg) - 1) / lx::trace_cfg_panel(sweep_cfg));
        mq           = (uint64_t)smax_entry * std::min(o.max_qlen, o.max_slen) < 32000 && o.max_slen <= 65535;
        if (mq)
        {
            int const G    = 8;
            sweep_steps    = (uint32_t)((o.max_slen + G - 1 + 15) & ~15ull);
            sweep_stride32 = (uint64_t)sweep_panels * lx::ckpt_slot_dwords(sweep_cfg, sweep_steps);
            // (mq_wide: int16-pair slots from the sweep itself -- lists whose windows score beyond the compact codes, lx_host.cpp)
            mq_wide        = o.mq_wide && sweep_cfg == 1;
            sweep_stride   = mq_wide ? sweep_stride32 : (uint64_t)sweep_panels * lx::ckpt16_slot_dwords(sweep_cfg, sweep_steps);
            half_sweep     = true;
            // lane groups per query: a wavefront's 16 slots hold 16 / 8 / 4 windows of one query, whatever divides the run
            // (1 = the free packing: a lane group's pair shares a query, up to four queries per wavefront in any split)
            sweep_share    = o.query_run == 1 ? -1 : o.query_run == 2 ? 1 : (o.query_run % 16 == 0 ? 16 : o.query_run % 8 == 0 ? 8 : 4) / 2; // (-1: solo)
            sweep          = (o.n + 1) * sweep_stride * 4 <= o.trace_bytes;
            int64_t const worst = (int64_t)o.max_qlen * std::max(smax_entry, 0) + (int64_t)(-sc.gap_extend) * (sweep_steps + G + 2) +
                                  (smax_entry - sc.gap_extend) + 2;
            may_decline = mq_wide ? worst > 0x7BFF - 2048 : (worst > 2046 || sweep_panels > 1);
            if (sweep && may_decline)
                ovf_cap = std::min<uint64_t>(o.n, (o.trace_bytes - (o.n + 1) * sweep_stride * 4) / (sweep_stride32 * 4));
            mq = sweep; // (a batch beyond the slot budget: the per-survivor paths below)
            if (!sweep)
                half_sweep = mq_wide = false;
        }
    }
    if (!mq && o.pass2 == 2 && shared && sc.trace_ok && !o.band)
    {
        // one panel of (8,19) or (16,13); wider queries: several (16,13) panels, int32 sweep
        sweep_cfg    = ckpt_cfg_for(o.max_qlen, o.f16 && o.query_run % 16 == 0);
        // short queries (<= 104 columns, e.g. 100-residue reads): the (8,13) geometry where the packed-half sweep applies --
        // a third fewer padded columns than (8,19)
        bool const half_ok = o.f16 && -sc.gap_open <= lx::kC16MaxGap && sc.gap_open <= sc.gap_extend;
        if (sweep_cfg == 1 && half_ok && o.max_qlen <= (uint64_t)lx::trace_cfg_panel(3) &&
            (o.query_run % 16 == 0 || 2 * lx::score_pair_profile_bytes(1, nrows_sc) + 64 * 8 * 4 <= pair_lds_limit()))
            sweep_cfg = 3;
        // 153 - 200 columns: 25-column strips of 8-lane groups (16 extensions of one query per wavefront, 7 steps of skew
        // instead of 15, no padded column at 200) where the packed-half sweep applies
        if (sweep_cfg == 2 && half_ok && o.max_qlen <= (uint64_t)lx::trace_cfg_panel(4) &&
            o.query_run % 16 == 0)
            sweep_cfg = 4;
        sweep_panels = (uint32_t)std::max<uint64_t>(1, (o.max_qlen + lx::trace_cfg_panel(sweep_cfg) - 1) / lx::trace_cfg_panel(sweep_cfg));
        int const smax_entry = sc.smax_entry;
        if (sweep_cfg != 0 && (uint64_t)smax_entry * std::min(o.max_qlen, o.max_slen) < 32000 && o.max_slen <= 65535)
        {
            int const G    = lx::trace_cfg_group(sweep_cfg);
            sweep_steps    = (uint32_t)((o.max_slen + G - 1 + 15) & ~15ull);
            sweep_stride32 = (uint64_t)sweep_panels * lx::ckpt_slot_dwords(sweep_cfg, sweep_steps);
            // Packed half precision where its geometry matches the checkpoint layout ((8,19): 16 extensions of one query per
            // wavefront, or runs of 8 with one query per half wavefront where two LDS profiles fit, i.e. for the small
            // alphabets; (16,13): 8 extensions) and a gap's first character costs at most 31 (the compact checkpoint codes
            // of Ckpt16Layout).  Wavefronts it declines leave the sentinel -1; the int32 kernel fills those in.
            half_sweep = o.f16 && sweep_panels == 1 && -sc.gap_open <= lx::kC16MaxGap &&
                         sc.gap_open <= sc.gap_extend &&
                         (((sweep_cfg == 1 || sweep_cfg == 3 || sweep_cfg == 4) && o.query_run % 16 == 0) || sweep_cfg == 2);
            // Queries wider than a panel: compact codes as well, one part per (8,19) panel, written by the packed int16 kernel
            // (the half-precision one has no carry between panels); what scores beyond the codes' 2046 goes to the int32 launch
            wide_compact = o.f16 && sweep_panels > 1 && sweep_cfg == 1 && o.query_run % 16 == 0 &&
                           -sc.gap_open <= lx::kC16MaxGap && sc.gap_open <= sc.gap_extend;
            half_sweep = half_sweep || wide_compact;
            if (o.f16 && sweep_panels == 1 && -sc.gap_open <= lx::kC16MaxGap &&
                sc.gap_open <= sc.gap_extend && (sweep_cfg == 1 || sweep_cfg == 3) && !half_sweep && o.query_run % 8 == 0 &&
                2 * lx::score_pair_profile_bytes(sweep_cfg == 3 ? 1 : 0, nrows_sc) + 64 * 8 * 4 <= pair_lds_limit())
            {
                half_sweep  = true;
                sweep_share = 4;
            }
            if (half_sweep)
            {
                // compact slots for the batch (+ the spare slot idle halves write to), int16-pair slots for what the
                // packed kernel declines in whatever the budget leaves
                sweep_stride = (uint64_t)sweep_panels * lx::ckpt16_slot_dwords(sweep_cfg, sweep_steps);
                sweep        = (o.n + 1) * sweep_stride * 4 <= o.trace_bytes;
                // (the packed kernel's exactness gate, lx_score_f16.hip: it cannot decline when even the worst query passes)
                int64_t const worst = (int64_t)o.max_qlen * std::max(smax_entry, 0) +
                                      (int64_t)(-sc.gap_extend) * (sweep_steps + G + 2) +
                                      (smax_entry - sc.gap_extend) + 2; // (ScoringDev::smax = largest entry - ge)
                may_decline = worst > 2046 || wide_compact;
                if (sweep && may_decline)
                    ovf_cap = std::min<uint64_t>(o.n, (o.trace_bytes - (o.n + 1) * sweep_stride * 4) / (sweep_stride32 * 4));
            }
            else
            {
                sweep_stride = sweep_stride32;
                sweep        = o.n * sweep_stride * 4 <= o.trace_bytes;
            }
        }
    }
    // No packed-half sweep (queries wider than a panel, gap costs beyond the compact codes, ...): the packed int16 kernel
    // writes the int16-pair slots of the int32 kernel, two extensions per lane group; what fails its range test is left to
    // the int32 launch.  16 extensions of one query per wavefront at (8,19), 8 at (16,13).
    bool const i16_sweep = sweep && !half_sweep && o.f16 && o.query_run % (sweep_cfg == 1 ? 16 : 8) == 0 &&
                           sweep_cfg != 3 && sweep_cfg != 4;
    // Adaptive choice (the one-query-per-wavefront sweep; lx_extend_batch's multi-query plan keeps its sweep): few survivors
    // last time -> plain pass 1, checkpoints for the survivors only (mode 1).
    bool const adapted = sweep && !mq && o.adapt != 0 && o.surv_frac >= 0.0 && o.surv_frac * 1000.0 < (double)o.adapt;
    StepPlan pl{};
    pl.shared       = shared;
    pl.sweep        = sweep && !adapted;
    pl.adapted      = adapted;
    pl.family       = !pl.sweep ? kNoSweep : mq ? kMqSweep : wide_compact ? kI16CompactWide : half_sweep ? kHalfSweep : i16_sweep ? kI16Pairs : kInt32Sweep;
    pl.cfg          = sweep_cfg;
    pl.steps        = sweep_steps;
    pl.panels       = sweep_panels;
    pl.stride       = sweep_stride;
    pl.stride32     = sweep_stride32;
    pl.ovf_cap      = pl.sweep ? ovf_cap : 0;
    pl.share        = sweep_share;
    pl.compact      = pl.sweep && half_sweep && !mq_wide;
    pl.packed       = pl.sweep && half_sweep;
    pl.wide         = pl.sweep && mq_wide;
    pl.may_decline  = may_decline;
    return pl;
}

// the sweep a plan names, as the launch sequence fused_impl issues for it (what lx_last_trace_kernel_name reports)
void lxi::describe_plan(StepPlan const & pl, char * buf, size_t len)
{
    int const nameG = lx::trace_cfg_group(pl.cfg), nameC = lx::trace_cfg_panel(pl.cfg) / std::max(1, lx::trace_cfg_group(pl.cfg));
    switch (pl.family)
    {
        case kMqSweep:
            snprintf(buf, len, "lx::sweep_mq_kernel<%d,%s> (single sweep, %s%d queries per wavefront%s)", nameC, pl.wide ? "true,true" : pl.panels > 1 ? "true,false" : "false,false",
                     pl.share < 0 ? "solo packing: up to " : pl.share == 1 ? "free packing: up to " : "", pl.share < 0 ? 16 : pl.share == 1 ? 4 : 8 / std::max(1, pl.share),
                     pl.may_decline ? "; + int32 fix-up lx::ckpt_forward_kernel" : "");
            break;
        case kI16CompactWide:
            snprintf(buf, len, "lx::sweep_pair16_kernel<%d,%d,true,true,true> (single sweep, compact codes; + int32 fix-up lx::ckpt_forward_kernel<%d,%d,false>)",
                     nameG, nameC, nameG, nameC);
            break;
        case kHalfSweep:
            if (pl.may_decline)
                snprintf(buf, len, "lx::score_pair_kernel<%d,%d,true> (single sweep; + int32 fix-up lx::ckpt_forward_kernel<%d,%d,false>)", nameG, nameC,
                         nameG, nameC);
            else
                snprintf(buf, len, "lx::score_pair_kernel<%d,%d,true> (single sweep)", nameG, nameC);
            break;
        case kI16Pairs:
            snprintf(buf, len, "lx::sweep_pair16_kernel<%d,%d,%s> (single sweep; + int32 fix-up lx::ckpt_forward_kernel<%d,%d,false>)", nameG, nameC,
                     pl.panels > 1 ? "true" : "false", nameG, nameC);
            break;
        case kInt32Sweep: snprintf(buf, len, "lx::ckpt_forward_kernel<%d,%d,false> (single sweep)", nameG, nameC); break;
        default: snprintf(buf, len, "%s", pl.adapted ? "pass 1, then checkpoints for the survivors (few survived the last batch)" : "pass 1, then pass 2 on the survivors"); break;
    }
}

int lxi::fused_impl(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext, uint64_t n,
                      void const * d_min_score, int32_t min_score_all, void * d_out_score, void * d_out_hsp, void * d_out_ops,
                      void const * d_ops_off, void * d_out_count, void * stream_, int phases, bool by_pos,
                      FusedExtra const * fx)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (n == 0)
        return LX_OK;
    if (!d_q_res || !d_s_res || !d_ext || !d_out_score || !d_out_count ||
        ((phases & 2) && (!d_out_hsp || !d_out_ops || (!d_ops_off && !(fx && fx->ops_stride)))))
        return fail(h, LX_EINVAL, "NULL device pointer");
    if (h->opt_max_qlen == 0 || h->opt_max_slen == 0)
        return fail(h, LX_ESTATE, "lx_extend_batch_dev needs LX_OPT_MAX_QLEN and LX_OPT_MAX_SLEN (it never synchronises)");
    if (n > 0xfffffff0ull)
        return fail(h, LX_EINVAL, "at most 2^32-16 extensions per call");
    int rc = bind(h);
    if (rc)
        return rc;
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : h->stream;

    lx_handle::MqTab const tab = h->mq_tab; // (slots by wavefront: lx_extend_batch's multi-query chunks)
    bool       tab_on = tab.dev != nullptr;
    if ((phases & 1) && !(tab_on && tab.part == 2))
    {
        if (!h->keep_phase_events) // (lx_extend_batch's pipeline collects the events of all its chunks)
        {
            h->phase_ev.clear();
            h->ev_pool_used = 0;
        }
        LX_HIP(h, hipEventRecord(h->ev0, stream));
    }
    // how this step runs: plan_step() decides, this function follows
    SchemeFacts facts{};
    facts.alph       = h->sc_host[slot].alphabet_size;
    facts.gap_open   = h->sc_host[slot].gap_open;
    facts.gap_extend = h->sc_host[slot].gap_extend;
    facts.trace_ok   = h->trace_ok[slot];
    facts.b8_ok      = h->b8_ok[slot];
    for (int a = 0; a < facts.alph; ++a)
        for (int b = 0; b < facts.alph; ++b)
            facts.smax_entry = std::max<int>(facts.smax_entry, h->sc_host[slot].matrix[a * LX_ALPH + b]);
    StepOptions so{};
    so.max_qlen = h->opt_max_qlen, so.max_slen = h->opt_max_slen, so.query_run = h->opt_query_run, so.pass2 = h->opt_pass2;
    so.mq = h->opt_mq, so.f16 = h->opt_f16, so.band = h->opt_band, so.trace_bytes = h->opt_trace_bytes, so.n = n;
    so.mq_cfg_call = h->mq_cfg_call, so.adapt = h->opt_adapt, so.mq_wide = h->mq_wide_call;
    // (phase 2 of a split step follows what phase 1 decided: it sees the survivor share phase 1 saw)
    if (phases & 1)
        h->plan_surv_frac = h->surv_frac;
    so.surv_frac = h->plan_surv_frac;
    StepPlan plan{};
    if (tab_on)
    {
        // the chunk's maxima decide the kernel (family, geometry, multi-panel or not) and the layout of the overflow slots; the slots
        // themselves are the table's
        so.n = 1;
        plan = plan_step(facts, so);
        so.n = n;
        if (plan.family != kMqSweep)
        {
            // (a chunk the sweep does not take -- a window beyond the 65 535 rows its slots address: the per-survivor paths, as without a table)
            if (tab.part != 0)
                return fail(h, LX_ESTATE, "a chunk swept in two calls must be the multi-query sweep's");
            tab_on = false;
            plan   = plan_step(facts, so);
        }
        else
            plan.ovf_cap = (plan.may_decline && plan.stride32 != 0) ? tab.ovf_cap : 0;
    }
    else
        plan = plan_step(facts, so);
    bool const     shared = plan.shared;
    bool const     sweep = plan.sweep, mq = plan.family == kMqSweep, wide_compact = plan.family == kI16CompactWide;
    bool const     half_sweep = plan.packed, i16_sweep = plan.family == kI16Pairs, may_decline = plan.may_decline;
    int const      sweep_cfg = plan.cfg, sweep_share = plan.share;
    uint32_t const sweep_steps = plan.steps, sweep_panels = plan.panels;
    uint64_t const sweep_stride = plan.stride, sweep_stride32 = plan.stride32, ovf_cap = plan.ovf_cap;
    int const      nrows_sc = ((h->sc_host[slot].alphabet_size + 1 + 3) / 4) * 4;
    // the packed-half sweep of one-panel queries interleaves the compact slots of a wavefront's windows (ScoreParams::wave_slots):
    // what one store instruction writes is then one contiguous piece (headline sweep 11.87 -> 11.6 ms)
    bool const     wave_slots = half_sweep && !mq && !wide_compact; // (= launch_score_pair: always one panel)
    uint64_t const wave_w     = 128 / (uint64_t)lx::trace_cfg_group(sweep_cfg); // windows of a packed-half wavefront
    // the batch's slots (+ the spare slot of the compact layouts; whole wavefronts of slots when they are interleaved)
    // (slots by wavefront: [the wavefronts before slot n0][overflow slots][the wavefronts from n0 on])
    uint64_t const tab_ovf_dw = tab_on ? ovf_cap * sweep_stride32 : 0;
    uint64_t const batch_dw = tab_on       ? tab.dw0
                              : wave_slots ? (n + wave_w - 1) / wave_w * wave_w * sweep_stride
                              : half_sweep ? (n + 1) * sweep_stride
                                           : n * sweep_stride;
    if (sweep && (phases & 1))
    {
        bool const second = tab_on && tab.part == 2; // (the first call sized the buffers and reset the counters: its sweep may still be running)
        if (tab_on && std::max(tab.total_dw, tab.dw0 + tab_ovf_dw + tab.dw1) * 4 > h->d_trace.cap && second)
            return fail(h, LX_ESTATE, "the second sweep of a chunk needs more slot memory than its first reserved");
        if (!second && ((rc = ensure(h, h->d_trace, (tab_on ? std::max(tab.total_dw, tab.dw0 + tab_ovf_dw + tab.dw1) : batch_dw + ovf_cap * sweep_stride32) * 4)) ||
                        (rc = ensure(h, h->d_ends, n * sizeof(lx::EndCell)))))
            return rc;
        if (!second && (rc = prepare_workspace(h, stream, sweep_panels > 1 ? n * ((h->opt_max_slen + 3) & ~3ull) : 0)))
            return rc;
        if (!second)
            LX_HIP(h, hipMemsetAsync(h->d_ws_top + 4, 0, sizeof(uint32_t), stream));
        lx::TraceParams p{};
        p.q_res          = static_cast<uint8_t const *>(d_q_res);
        p.s_res          = static_cast<uint8_t const *>(d_s_res);
        p.ext            = static_cast<lx::Extension const *>(d_ext);
        p.n              = n;
        p.sc             = h->sc_dev[slot];
        p.trace          = static_cast<uint32_t *>(h->d_trace.ptr);
        p.slot_stride    = sweep_stride;
        p.steps_cap      = sweep_steps;
        p.panels_cap     = sweep_panels;
        p.ws             = static_cast<int32_t *>(h->d_ws.ptr);
        p.ws_top         = h->d_ws_top;
        p.ws_cap         = (uint32_t)std::min<uint64_t>(h->d_ws.cap / 8, 0xffffffffu);
        p.ends           = static_cast<lx::EndCell *>(h->d_ends.ptr);
        p.score_out      = static_cast<int32_t *>(d_out_score);
        p.err            = reinterpret_cast<int32_t *>(h->d_ws_top + 1);
        p.nrows          = nrows_sc;
        // every wavefront holds one query (mq: every run; the solo packing: every window its own)
        p.shared_profile = mq ? (sweep_share < 0 ? 1 : std::min(2 * sweep_share, 8)) : 64 / lx::trace_cfg_group(sweep_cfg);
        p.cfg            = sweep_cfg;
        if (half_sweep)
        {
            p.ovf        = p.trace + batch_dw;
            p.ovf_stride = sweep_stride32;
            p.ovf_cap    = (uint32_t)ovf_cap;
            p.ovf_count  = h->d_ws_top + 4;
        }
        int const sweep_pair = sweep_cfg == 1 ? 0 : sweep_cfg == 3 ? 1 : sweep_cfg == 4 ? 7 : 5; // pair geometry with the same (G, C): (8,19) / (8,13) / (8,25) / (16,13)
        PhaseTimer pt0(h, stream, 0);
        if (half_sweep)
        {
            lx::ScoreParams sp1{};
            sp1.q_res       = p.q_res;
            sp1.s_res       = p.s_res;
            sp1.ext         = p.ext;
            sp1.n           = n;
            sp1.sc          = p.sc;
            sp1.out_score   = static_cast<int32_t *>(d_out_score);
            sp1.err         = p.err;
            sp1.nrows       = p.nrows;
            sp1.ckpt        = p.trace;
            sp1.ckpt_stride = sweep_stride;
            sp1.steps_cap   = sweep_steps;
            sp1.ends        = p.ends;
            sp1.pair_share  = std::max(sweep_share, 0);
            if (mq)
            {
                if (tab_on)
                {
                    sp1.wf_tab = static_cast<lx::WfSlots const *>(tab.dev);
                    sp1.wf_lo  = tab.part == 2 ? (uint32_t)(tab.n0 / 16) : 0u;
                    sp1.n      = tab.part == 1 ? tab.n0 : n;
                    sp1.ckpt   = tab.part == 2 ? p.trace + tab.dw0 + tab_ovf_dw : p.trace;
                }
                sp1.wide        = plan.wide ? 1 : 0;
                sp1.stat_beyond = h->d_ws_top + 6;
                if (!second)
                    LX_HIP(h, hipMemsetAsync(h->d_ws_top + 6, 0, sizeof(uint32_t), stream));
                if (sweep_share < 0) // the solo packing: rows for the alphabet's letters and the pad letter, no more
                {
                    sp1.solo  = 1;
                    sp1.nrows = h->sc_host[slot].alphabet_size + 1;
                }
                sp1.narrow     = 1;
                sp1.ws         = p.ws;
                sp1.ws_top     = p.ws_top;
                sp1.ws_cap     = p.ws_cap;
                sp1.panels_cap = sweep_panels;
                LX_HIP(h, lx::launch_sweep_mq(sweep_cfg, sp1, stream));
                if (tab_on && tab.part == 1)
                {
                    // the first of a chunk's two sweeps: the rest of the chunk is still being planned -- the second call goes on from here
                    pt0.close();
                    return LX_OK;
                }
                if (sweep_panels > 1) // the fix-up launch starts with an empty carry workspace
                    LX_HIP(h, hipMemsetAsync(h->d_ws_top, 0, sizeof(uint32_t), stream));
            }
            else if (wide_compact)
            {
                sp1.ws         = p.ws;
                sp1.ws_top     = p.ws_top;
                sp1.ws_cap     = p.ws_cap;
                sp1.panels_cap = sweep_panels;
                LX_HIP(h, lx::launch_sweep_pair16_compact(sweep_cfg, sp1, stream));
                LX_HIP(h, hipMemsetAsync(h->d_ws_top, 0, sizeof(uint32_t), stream)); // the fix-up launch starts with an empty carry workspace
            }
            else
            {
                sp1.wave_slots = wave_slots ? 1 : 0;
                LX_HIP(h, lx::launch_score_pair(sweep_pair, sp1, stream));
            }
            p.fixup = 1;
        }
        if (i16_sweep)
        {
            lx::ScoreParams sp1{};
            sp1.q_res       = p.q_res;
            sp1.s_res       = p.s_res;
            sp1.ext         = p.ext;
            sp1.n           = n;
            sp1.sc          = p.sc;
            sp1.out_score   = static_cast<int32_t *>(d_out_score);
            sp1.ws          = p.ws;
            sp1.ws_top      = p.ws_top;
            sp1.ws_cap      = p.ws_cap;
            sp1.err         = p.err;
            sp1.nrows       = p.nrows;
            sp1.ckpt        = p.trace;
            sp1.ckpt_stride = sweep_stride;
            sp1.steps_cap   = sweep_steps;
            sp1.ends        = p.ends;
            sp1.panels_cap  = sweep_panels;
            LX_HIP(h, lx::launch_sweep_pair16(sweep_cfg, sp1, stream));
            if (sweep_panels > 1) // the fix-up launch starts with an empty carry workspace
                LX_HIP(h, hipMemsetAsync(h->d_ws_top, 0, sizeof(uint32_t), stream));
            p.fixup = 1;
        }
        if (!half_sweep || may_decline) // (the packed-half kernel declines nothing when even the worst query passes its test)
            LX_HIP(h, lx::launch_ckpt_forward(p, stream));
        pt0.close();
        char buf[200];
        describe_plan(plan, buf, sizeof(buf));
        h->last_kernel       = buf;
        h->last_trace_kernel = buf;
    }
    else if (phases & 1)
    {
        // pass 1 (src/search_algo.hpp:1246).  Pass 2 may need the carry workspace even where pass 1 does not (its panels
        // are narrower): size it now, while nothing is in flight
        if (h->opt_max_qlen > (uint64_t)lx::trace_cfg_panel(1) && (rc = prepare_workspace(h, stream, n * ((h->opt_max_slen + 3) & ~3ull))))
            return rc;
        h->in_fused = true;
        rc          = lx_score_batch_dev(h, slot, d_q_res, d_s_res, d_ext, n, d_out_score, stream);
        h->in_fused = false;
        if (rc)
            return rc;
    }

    // filter (:1251-1283) as an integer cut-off, compaction in input order, runs padded to whole wavefronts
    uint32_t const run    = shared ? (uint32_t)h->opt_query_run : 1u;
    // half a wavefront of the 8-lane geometry, a whole one of the 16-lane; the single sweep's backtrace needs no padding
    // (the adaptive step has few survivors, one or two to a query: its lists pad every query's to 2 slots -- four LDS profiles
    // per wavefront of the int32 forward kernel -- instead of 4: 2.6 x -> 1.9 x the survivors' cells at 2 % survivors)
    uint32_t const pad_to = (shared && !sweep) ? (plan.adapted ? 2u : 4u) : 1u;
    uint64_t const nruns  = (n + run - 1) / run;
    uint64_t const cap    = (n + (shared ? nruns * 3 : 0) + 7) / 8 * 8;
    if ((rc = ensure(h, h->d_sel_ext, cap * sizeof(lx_extension))) || (rc = ensure(h, h->d_sel_src, cap * sizeof(uint32_t))) ||
        (rc = ensure(h, h->d_sel_runs, (nruns + 2 * lx::select_blocks(pad_to <= 1 ? n : nruns) + 2) * sizeof(uint64_t))) || (rc = ensure(h, h->d_sel_score, cap * sizeof(int32_t))))
        return rc;
    if (phases & 1)
    {
    lx::SelectParams sp{};
    sp.ext           = static_cast<lx::Extension const *>(d_ext);
    sp.score         = static_cast<int32_t const *>(d_out_score);
    sp.min_score     = static_cast<int32_t const *>(d_min_score);
    sp.min_score_all = min_score_all;
    sp.n             = n;
    sp.run           = run;
    sp.pad_to        = pad_to;
    sp.run_slots     = static_cast<uint64_t *>(h->d_sel_runs.ptr);
    sp.block_tot     = sp.run_slots + nruns;
    sp.out_ext       = static_cast<lx::Extension *>(h->d_sel_ext.ptr);
    sp.out_src       = static_cast<uint32_t *>(h->d_sel_src.ptr);
    sp.out_score     = static_cast<int32_t *>(h->d_sel_score.ptr);
    sp.out_count     = static_cast<uint64_t *>(d_out_count);
    sp.out_hsp       = by_pos ? nullptr : static_cast<lx::Hsp *>(d_out_hsp); // rows of the filtered-out extensions
    PhaseTimer pts(h, stream, 1);
    LX_HIP(h, lx::launch_select(sp, stream));
    pts.close();
    }
    if (!(phases & 2))
        return LX_OK;

    if (sweep)
    {
        // backtrace of the survivors straight from the checkpoints of the sweep (slots and end cells by original index)
        lx::TraceParams p{};
        p.q_res         = static_cast<uint8_t const *>(d_q_res);
        p.s_res         = static_cast<uint8_t const *>(d_s_res);
        p.ext           = static_cast<lx::Extension const *>(h->d_sel_ext.ptr);
        p.n             = cap;
        p.sc            = h->sc_dev[slot];
        p.trace         = static_cast<uint32_t *>(h->d_trace.ptr);
        p.slot_stride   = sweep_stride;
        p.steps_cap     = sweep_steps;
        p.panels_cap    = sweep_panels;
        p.ends          = static_cast<lx::EndCell *>(h->d_ends.ptr);
        p.out_hsp       = static_cast<lx::Hsp *>(d_out_hsp);
        p.out_ops       = static_cast<uint8_t *>(d_out_ops);
        p.ops_off       = static_cast<uint64_t const *>(d_ops_off);
        p.ops_stride    = fx ? fx->ops_stride : 0;
        if (fx && fx->d_rle) // the backtrace writes run-length codes itself
        {
            p.rle     = fx->d_rle;
            p.rle_top = fx->d_rle_top;
            p.rle_cap = fx->rle_cap;
            p.rle_len = fx->d_rle_len;
            LX_HIP(h, hipMemsetAsync(fx->d_rle_top, 0, sizeof(unsigned long long), stream));
            if (fx->d_rle_len) // (positions the backtrace never visits -- padding, score-less -- read 0)
                LX_HIP(h, hipMemsetAsync(fx->d_rle_len, 0, cap * sizeof(uint32_t), stream));
        }
        p.src           = static_cast<uint32_t const *>(h->d_sel_src.ptr);
        p.count_ptr     = static_cast<uint64_t const *>(d_out_count);
        p.chunk_start   = 0;
        p.err           = reinterpret_cast<int32_t *>(h->d_ws_top + 1);
        p.nrows         = ((h->sc_host[slot].alphabet_size + 1 + 3) / 4) * 4;
        p.bs_match_rule = (int32_t)h->opt_bs_rule;
        p.work_counter  = h->d_ws_top + 5;
        p.cfg           = sweep_cfg;
        p.slot_by_src   = 1;
        p.out_by_pos    = by_pos ? 1 : 0;
        if (half_sweep)
        {
            p.ovf        = p.trace + batch_dw; // int16-pair slots of what the packed kernel declined
            p.ovf_stride = sweep_stride32;
        }
        if (tab_on)
        {
            p.wf_tab  = static_cast<lx::WfSlots const *>(tab.dev);
            p.split_n = tab.dw1 ? tab.n0 : 0;
            p.trace2  = p.trace + tab.dw0 + tab_ovf_dw;
        }
        PhaseTimer ptb(h, stream, 3);
        LX_HIP(h, lx::launch_ckpt_backtrace(p, stream));
        ptb.close();
        if ((rc = fused_pack(h, fx, cap, d_out_hsp, d_out_ops, d_ops_off, d_out_count, stream, true)))
            return rc;
        LX_HIP(h, hipEventRecord(h->ev1, stream));
        h->timed = true;
        return LX_OK;
    }

    // pass 2 on the survivors (:1293-1296); the grid covers the worst case, wavefronts beyond *d_out_count exit
    rc = align_dev_impl(h, slot, d_q_res, d_s_res, static_cast<lx::Extension const *>(h->d_sel_ext.ptr), cap,
                        static_cast<lx::Hsp *>(d_out_hsp), static_cast<uint8_t *>(d_out_ops),
                        static_cast<uint64_t const *>(d_ops_off), stream, h->opt_max_qlen, h->opt_max_slen, shared ? (int)pad_to : 0,
                        static_cast<uint32_t const *>(h->d_sel_src.ptr), static_cast<uint64_t const *>(d_out_count),
                        static_cast<int32_t const *>(h->d_sel_score.ptr), by_pos, fx ? fx->ops_stride : 0);
    if (rc)
        return rc;
    if ((rc = fused_pack(h, fx, cap, d_out_hsp, d_out_ops, d_ops_off, d_out_count, stream)))
        return rc;
    LX_HIP(h, hipEventRecord(h->ev1, stream));
    h->timed = true;
    return LX_OK;
}



extern "C" {

int lx_abi_version(void)
{
    return LX_ABI_VERSION;
}

#ifndef LX_BUILD_ID
#define LX_BUILD_ID "unknown-build-id"
#endif
// "LXBUILDID:" + id: lambda_amd/build.py finds the marker in the file without loading it
static char const g_build_id[] = "LXBUILDID:" LX_BUILD_ID;
char const * lx_build_id(void)
{
    return g_build_id + 10;
}

int lx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int lx_create(int device_id, lx_handle ** out)
{
    if (!out)
        return fail(nullptr, LX_EINVAL, "lx_create: out is NULL");
    *out  = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, LX_ENODEV, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, LX_EINVAL, "device_id %d out of range [0,%d)", device_id, n);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess)
        return fail(nullptr, LX_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, LX_ENODEV, "device %d is %s; this library ships gfx950 (MI355X) code only", device_id,
                    prop.gcnArchName);

    lx_handle * h = new lx_handle();
    if (lx::dev_aids().host_threads) // (measurement aid of tools/host_curve.py; a caller uses LX_OPT_HOST_THREADS)
        lxi::HostPool::instance().set_width(lx::dev_aids().host_threads);
    h->device     = device_id;
    auto bail     = [&](char const * what, hipError_t err)
    {
        int rc = fail(nullptr, LX_EHIP, "%s: %s", what, hipGetErrorString(err));
        lx_destroy(h);
        return rc;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess)
        return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess)
        return bail("hipEventCreate", e);
    if ((e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    for (auto & ln : h->xb)
        for (hipEvent_t * ev : {&ln.ev_up, &ln.ev_k, &ln.ev_cnt})
            if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate", e);
    if ((e = hipMalloc(reinterpret_cast<void **>(&h->d_ws_top), 8 * sizeof(uint32_t))) != hipSuccess)
        return bail("hipMalloc", e);
    if ((e = hipMemset(h->d_ws_top, 0, 8 * sizeof(uint32_t))) != hipSuccess)
        return bail("hipMemset", e);
    for (int s = 0; s < 2; ++s)
        if ((e = hipMalloc(reinterpret_cast<void **>(&h->sc_dev[s]), sizeof(lx::ScoringDev))) != hipSuccess)
            return bail("hipMalloc", e);
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&h->p_count), 2 * sizeof(uint64_t), hipHostMallocDefault)) != hipSuccess)
        return bail("hipHostMalloc", e);
    if ((e = hipEventCreateWithFlags(&h->ev_count, hipEventDisableTiming)) != hipSuccess)
        return bail("hipEventCreate", e);
    *out = h;
    return LX_OK;
}

void lx_destroy(lx_handle * h)
{
    if (!h)
        return;
    if (h->device >= 0)
        (void)hipSetDevice(h->device);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    for (DevBuf * b : {&h->d_q, &h->d_s, &h->d_ext, &h->d_out, &h->d_ops, &h->d_opsoff, &h->d_keep, &h->d_trace, &h->d_ends,
                       &h->d_hsp, &h->d_seeds, &h->d_sel_ext, &h->d_sel_src, &h->d_sel_runs, &h->d_sel_score, &h->d_trace_score, &h->d_db, &h->d_ws, &h->d_band})
        if (b->ptr)
            (void)hipFree(b->ptr);
    for (int s = 0; s < 2; ++s)
        if (h->sc_dev[s])
            (void)hipFree(h->sc_dev[s]);
    if (h->d_ws_top)
        (void)hipFree(h->d_ws_top);
    if (h->ev_count)
        (void)hipEventDestroy(h->ev_count);
    if (h->p_count)
        (void)hipHostFree(h->p_count);
    for (hipStream_t st : {h->stream2, h->stream3})
        if (st)
        {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (auto & ln : h->xb)
    {
        for (DevBuf * b : {&ln.d_ext, &ln.d_min, &ln.d_score, &ln.d_hsp, &ln.d_ops, &ln.d_rle, &ln.d_src, &ln.d_cnt, &ln.d_len, &ln.d_orig, &ln.d_wft})
            if (b->ptr)
                (void)hipFree(b->ptr);
        for (lx_handle::Pinned * b : {&ln.p_ext, &ln.p_min, &ln.p_score, &ln.p_cnt, &ln.p_hsp, &ln.p_src, &ln.p_rle, &ln.p_len, &ln.p_orig, &ln.p_wft})
            if (b->ptr)
                (void)hipHostFree(b->ptr);
        for (hipEvent_t ev : {ln.ev_up, ln.ev_k, ln.ev_cnt})
            if (ev)
                (void)hipEventDestroy(ev);
    }
    for (DevBuf * b : {&h->d_ext_all, &h->d_min_all, &h->d_score_all})
        if (b->ptr)
            (void)hipFree(b->ptr);
    {
        auto & l2 = h->l2;
        for (DevBuf * b : {&l2.d_qres, &l2.d_qoff, &l2.d_qlen, &l2.d_qband, &l2.d_qevlen, &l2.d_soff, &l2.d_slen, &l2.d_pair[0], &l2.d_pair[1], &l2.d_s0[0],
                           &l2.d_s0[1], &l2.d_hist, &l2.d_head, &l2.d_tail, &l2.d_tot, &l2.d_win, &l2.d_cut, &l2.d_cnt, &l2.d_up, &l2.d_plan, &l2.d_wf, &l2.d_qevidx,
                           &l2.d_surv_hsp, &l2.d_surv_src, &l2.d_surv_codes, &l2.d_listat, &l2.d_rec, &l2.d_reccodes, &l2.d_reccnt, &l2.d_tilekeep, &l2.d_tileops,
                           &l2.d_pre, &l2.d_exp, &l2.d_rank, &l2.d_fp})
            if (b->ptr)
                (void)hipFree(b->ptr);
        for (lx_handle::Pinned * b : {&l2.p_cnt, &l2.p_win, &l2.p_up, &l2.p_reccnt, &l2.p_reccodes, &l2.p_rows, &l2.p_plan})
            if (b->ptr)
                (void)hipHostFree(b->ptr);
        if (l2.ev_win)
            (void)hipEventDestroy(l2.ev_win);
        for (hipEvent_t ev : l2.ev_rank)
            if (ev)
                (void)hipEventDestroy(ev);
    }
    for (lx_handle::Pinned * b : {&h->p_all, &h->p_score_all})
        if (b->ptr)
            (void)hipHostFree(b->ptr);
    for (hipEvent_t ev : h->ev_pool)
        (void)hipEventDestroy(ev);
    if (h->ev0)
        (void)hipEventDestroy(h->ev0);
    if (h->ev1)
        (void)hipEventDestroy(h->ev1);
    if (h->stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
}

char const * lx_last_error(lx_handle const * h)
{
    return h ? h->error.c_str() : g_create_error.c_str();
}

int lx_host_threads_info(uint32_t * width, uint32_t * granted_cpus, uint32_t * local_world_size)
{
    lxi::HostPool const & p = lxi::HostPool::instance();
    if (width)
        *width = p.width();
    if (granted_cpus)
        *granted_cpus = p.granted_cpus();
    if (local_world_size)
        *local_world_size = p.local_world();
    return LX_OK;
}

int lx_set_option(lx_handle * h, int option, uint64_t value)
{
    if (!h)
        return LX_EINVAL;
    switch (option)
    {
        case LX_OPT_MAX_QLEN: h->opt_max_qlen = value; return LX_OK;
        case LX_OPT_QUERY_RUN: h->opt_query_run = value; return LX_OK;
        case LX_OPT_WORKSPACE_BYTES: h->opt_ws_bytes = std::max<uint64_t>(value, 1 << 20); return LX_OK;
        case LX_OPT_MAX_SLEN: h->opt_max_slen = value; return LX_OK;
        case LX_OPT_TRACE_BYTES: h->opt_trace_bytes = std::max<uint64_t>(value, 1 << 20); return LX_OK;
        case LX_OPT_BS_MATCH_RULE: h->opt_bs_rule = value ? 1 : 0; return LX_OK;
        case LX_OPT_PACKED_HALF: h->opt_f16 = value ? 1 : 0; return LX_OK;
        case LX_OPT_PASS2_MODE: h->opt_pass2 = value > 2 ? 1 : value; return LX_OK;
        case LX_OPT_EXTEND_CHUNK: h->opt_extend_chunk = value; return LX_OK;
        case LX_OPT_MQ_SWEEP:
            h->opt_mq       = value > 2 ? 1 : value;
            h->mq_decl_frac = 0.0; // (what the last chunks taught about compact codes against int16 pairs starts over)
            h->mq_wide_call = false;
            return LX_OK;
        case LX_OPT_ADAPT_PERMILLE: h->opt_adapt = std::min<uint64_t>(value, 1000); h->surv_frac = -1.0; return LX_OK;
        case LX_OPT_ITERATE_RECORDS: h->opt_iterate_records = value ? 1 : 0; return LX_OK;
        case LX_OPT_HOST_THREADS:
            if (value > lxi::HostPool::kMaxParts)
                return fail(h, LX_EINVAL, "LX_OPT_HOST_THREADS: at most %u", lxi::HostPool::kMaxParts);
            lxi::HostPool::instance().set_width((unsigned)value); // (the host threads are the process's, not the handle's)
            return LX_OK;
        case LX_OPT_BAND:
            if (value > (1u << 20))
                return fail(h, LX_EINVAL, "LX_OPT_BAND: at most 2^20 diagonals on either side");
            h->opt_band = value;
            return LX_OK;
        default: return fail(h, LX_EINVAL, "unknown option %d", option);
    }
}

int lx_get_option(lx_handle const * h, int option, uint64_t * value)
{
    if (!h || !value)
        return LX_EINVAL;
    switch (option)
    {
        case LX_OPT_MAX_QLEN: *value = h->opt_max_qlen; return LX_OK;
        case LX_OPT_QUERY_RUN: *value = h->opt_query_run; return LX_OK;
        case LX_OPT_WORKSPACE_BYTES: *value = h->opt_ws_bytes; return LX_OK;
        case LX_OPT_MAX_SLEN: *value = h->opt_max_slen; return LX_OK;
        case LX_OPT_TRACE_BYTES: *value = h->opt_trace_bytes; return LX_OK;
        case LX_OPT_BS_MATCH_RULE: *value = h->opt_bs_rule; return LX_OK;
        case LX_OPT_PACKED_HALF: *value = h->opt_f16; return LX_OK;
        case LX_OPT_PASS2_MODE: *value = h->opt_pass2; return LX_OK;
        case LX_OPT_BAND: *value = h->opt_band; return LX_OK;
        case LX_OPT_EXTEND_CHUNK: *value = h->opt_extend_chunk; return LX_OK;
        case LX_OPT_MQ_SWEEP: *value = h->opt_mq; return LX_OK;
        case LX_OPT_ADAPT_PERMILLE: *value = h->opt_adapt; return LX_OK;
        case LX_OPT_ITERATE_RECORDS: *value = h->opt_iterate_records; return LX_OK;
        case LX_OPT_HOST_THREADS: *value = lxi::HostPool::instance().width(); return LX_OK;
        default: return LX_EINVAL;
    }
}

int lx_set_band_centres(lx_handle * h, int32_t const * diag, uint64_t n)
{
    if (!h || (!diag && n))
        return LX_EINVAL;
    h->band_host.assign(diag, diag + n);
    return LX_OK;
}

int lx_set_band_centres_dev(lx_handle * h, void const * d_diag)
{
    if (!h)
        return LX_EINVAL;
    h->band_dev = static_cast<int32_t const *>(d_diag);
    return LX_OK;
}

int lx_builtin_scoring(int scoring_method, int match, int mismatch, int gap_open_lambda, int gap_extend,
                       lx_scoring * sc)
{
    if (!sc)
        return LX_EINVAL;
    try
    {
        lambda_amd::builtinScoring(scoring_method, match, mismatch, gap_open_lambda, gap_extend, *sc);
    }
    catch (std::exception const &)
    {
        return LX_EINVAL;
    }
    return LX_OK;
}

// what lx_set_scoring derives from a scheme besides the tables: pass 2 applies (every matrix - gap_extend in [-31, 31]),
// byte profiles apply (0 <= matrix - gap_open <= 255), the largest entry
static void scheme_facts(lx_scoring const * sc, lxi::SchemeFacts & f)
{
    f.alph       = sc->alphabet_size;
    f.gap_open   = sc->gap_open;
    f.gap_extend = sc->gap_extend;
    f.trace_ok   = true;
    f.b8_ok      = true;
    f.smax_entry = 0;
    for (int a = 0; a < sc->alphabet_size; ++a)
        for (int b = 0; b < sc->alphabet_size; ++b)
        {
            int const v = sc->matrix[a * LX_ALPH + b];
            if (v - sc->gap_extend < -31 || v - sc->gap_extend > 31)
                f.trace_ok = false;
            if (v - sc->gap_open < 0 || v - sc->gap_open > 255)
                f.b8_ok = false;
            f.smax_entry = std::max(f.smax_entry, v);
        }
}

int lx_plan_step(lx_scoring const * sc, uint64_t max_qlen, uint64_t max_slen, uint64_t query_run, uint64_t n, uint64_t pass2_mode,
                 uint64_t mq_sweep, uint64_t packed_half, uint64_t trace_bytes, double survivor_share, uint64_t adapt_permille,
                 lx_step_plan * out)
{
    if (!sc || !out || sc->alphabet_size < 1 || sc->alphabet_size > LX_ALPH - 1 || sc->gap_extend >= 0 || sc->gap_open > sc->gap_extend)
        return LX_EINVAL;
    lxi::SchemeFacts f{};
    scheme_facts(sc, f);
    lxi::StepOptions o{};
    o.max_qlen = max_qlen, o.max_slen = max_slen, o.query_run = query_run, o.n = n, o.pass2 = pass2_mode > 2 ? 1 : pass2_mode;
    o.mq = mq_sweep > 2 ? 1 : mq_sweep, o.f16 = packed_half ? 1 : 0, o.trace_bytes = std::max<uint64_t>(trace_bytes, 1 << 20);
    o.surv_frac = survivor_share, o.adapt = std::min<uint64_t>(adapt_permille, 1000);
    lxi::StepPlan const pl = lxi::plan_step(f, o);
    std::memset(out, 0, sizeof(*out));
    out->family                = (int32_t)pl.family;
    out->adapted               = pl.adapted ? 1 : 0;
    lxi::describe_plan(pl, out->name, sizeof(out->name));
    if (pl.family == lxi::kNoSweep)
        return LX_OK;
    int const G = lx::trace_cfg_group(pl.cfg), C = lx::trace_cfg_panel(pl.cfg) / G, nrows = ((sc->alphabet_size + 1 + 3) / 4) * 4;
    out->group_lanes           = G;
    out->strip_cols            = C;
    out->panels                = (int32_t)pl.panels;
    out->compact_codes         = pl.compact ? 1 : 0;
    out->may_decline           = pl.may_decline ? 1 : 0;
    out->slot_bytes            = pl.stride * 4;
    int const per_wave         = pl.family == lxi::kInt32Sweep ? 64 / G : 2 * (64 / G); // extensions a wavefront holds
    int const share_ext        = pl.family == lxi::kMqSweep ? std::max(1, 2 * pl.share) : (pl.family == lxi::kHalfSweep && pl.share) ? 2 * pl.share : per_wave;
    out->queries_per_wavefront = std::max(1, per_wave / std::max(1, share_ext));
    if (pl.family == lxi::kMqSweep && pl.share == 1)
        out->queries_per_wavefront = 4; // (free packing: up to four, in any split of the eight lane groups)
    int const pair_cfg         = pl.cfg == 1 ? 0 : pl.cfg == 3 ? 1 : pl.cfg == 4 ? 7 : 5;
    switch (pl.family)
    {
        case lxi::kMqSweep: out->lds_bytes = lx::sweep_mq_lds_bytes(pl.cfg, pl.share < 0 ? sc->alphabet_size + 1 : nrows, pl.share); break;
        case lxi::kInt32Sweep: out->lds_bytes = (uint64_t)out->queries_per_wavefront * nrows * ((C + 3) / 4 * G) * 4 + 64 * 4 * 4; break;
        default: out->lds_bytes = (uint64_t)out->queries_per_wavefront * lx::score_pair_profile_bytes(pair_cfg, nrows) + 64 * 8 * 4; break;
    }
    // the largest value the widest admitted query can produce in a sweep that runs max_slen rows (the kernels' own tests are
    // per wavefront, on the actual query: this is the a-priori bound may_decline is derived from)
    out->score_bound = (uint64_t)((int64_t)max_qlen * std::max(f.smax_entry, 0) + (int64_t)(-sc->gap_extend) * ((int64_t)pl.steps + G + 2) +
                                  (f.smax_entry - sc->gap_extend) + 2);
    return LX_OK;
}

int lx_set_scoring(lx_handle * h, int slot, lx_scoring const * sc)
{
    if (!h || !sc)
        return LX_EINVAL;
    if (slot < 0 || slot > 1)
        return fail(h, LX_EINVAL, "slot must be 0 or 1");
    if (sc->alphabet_size < 1 || sc->alphabet_size > LX_ALPH - 1)
        return fail(h, LX_EINVAL, "alphabet_size must be in [1,31]");
    if (sc->gap_extend >= 0 || sc->gap_open > sc->gap_extend)
        return fail(h, LX_EINVAL, "need gap_open <= gap_extend < 0 (got %d / %d)", sc->gap_open, sc->gap_extend);
    if (sc->gap_open < -120 || sc->gap_extend < -27)
        return fail(h, LX_EINVAL, "gap costs out of the supported range");
    lx::ScoringDev d{};
    int            trace_ok = 1;
    d.alph = sc->alphabet_size;
    d.go   = sc->gap_open;
    d.ge   = sc->gap_extend;
    d.g2   = sc->gap_open - sc->gap_extend;
    for (int a = 0; a < lx::kAlph; ++a)
        for (int b = 0; b < lx::kAlph; ++b)
        {
            bool const pad = a >= sc->alphabet_size || b >= sc->alphabet_size;
            int const  v   = pad ? lx::kNegPad : sc->matrix[a * LX_ALPH + b];
            if (!pad && (v > 100 || v < -100))
                return fail(h, LX_EINVAL, "matrix entry [%d][%d]=%d outside [-100,100]", a, b, v);
            d.mat[a * lx::kAlph + b]     = (int8_t)v;
            d.mat_adj[a * lx::kAlph + b] = (int8_t)(pad ? lx::kNegPad : v - sc->gap_extend);
            int const adj                = v - sc->gap_extend;
            if (!pad && (adj < -31 || adj > 31))
                trace_ok = 0;
            d.mat_trace[a * lx::kAlph + b] = (int8_t)(pad || adj < -31 || adj > 31 ? -125 : 4 * adj + 3);
        }
    d.trace_ok = trace_ok;
    d.smax     = 0;
    d.b8_ok    = 1;
    for (int a = 0; a < lx::kAlph; ++a)
        for (int b = 0; b < lx::kAlph; ++b)
        {
            bool const pad = a >= sc->alphabet_size || b >= sc->alphabet_size;
            int const  v   = pad ? 0 : sc->matrix[a * LX_ALPH + b] - sc->gap_open;
            if (v < 0 || v > 255)
                d.b8_ok = 0;
            d.mat_b8[a * lx::kAlph + b] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    for (int a = 0; a < lx::kAlph; ++a)
    {
        int rm = 0;
        for (int b = 0; b < lx::kAlph; ++b)
        {
            bool const pad = a >= sc->alphabet_size || b >= sc->alphabet_size;
            int const  v   = pad ? lx::kNegPad : sc->matrix[a * LX_ALPH + b] - sc->gap_extend;
            _Float16 const hv = (_Float16)(float)v; // integers of magnitude <= 128: exact
            uint16_t       bits;
            std::memcpy(&bits, &hv, 2);
            d.mat_h[a * lx::kAlph + b]   = bits;
            d.mat_i16[a * lx::kAlph + b] = (int16_t)v;
            if (!pad)
            {
                rm     = std::max(rm, (int)sc->matrix[a * LX_ALPH + b]);
                d.smax = std::max(d.smax, (int)sc->matrix[a * LX_ALPH + b] - sc->gap_extend);
            }
        }
        d.rowmax[a] = (int16_t)rm;
    }
    int rc = bind(h);
    if (rc)
        return rc;
    // *_dev calls may still be in flight on a caller-supplied (non-blocking) stream and read this table
    LX_HIP(h, hipDeviceSynchronize());
    LX_HIP(h, hipMemcpy(h->sc_dev[slot], &d, sizeof(d), hipMemcpyHostToDevice));
    h->sc_host[slot] = *sc;
    h->have_sc[slot] = true;
    h->trace_ok[slot] = trace_ok != 0;
    h->b8_ok[slot]    = d.b8_ok != 0;
    return LX_OK;
}

int lx_synchronize(lx_handle * h)
{
    if (!h)
        return LX_EINVAL;
    int rc = bind(h);
    if (rc)
        return rc;
    return check_async_error(h);
}

char const * lx_last_kernel_name(lx_handle const * h)
{
    return h ? h->last_kernel.c_str() : "";
}

char const * lx_last_trace_kernel_name(lx_handle const * h)
{
    return h ? h->last_trace_kernel.c_str() : "";
}

int lx_last_phase_ms(lx_handle * h, int phase, float * ms, int * launches)
{
    if (!h || !ms)
        return LX_EINVAL;
    int rc = bind(h);
    if (rc)
        return rc;
    float total = 0.f;
    int   cnt   = 0;
    for (auto const & pe : h->phase_ev)
        if (pe.phase == phase)
        {
            float t = 0.f;
            LX_HIP(h, hipEventSynchronize(pe.b));
            LX_HIP(h, hipEventElapsedTime(&t, pe.a, pe.b));
            total += t;
            ++cnt;
        }
    *ms = total;
    if (launches)
        *launches = cnt;
    return LX_OK;
}

int lx_last_kernel_ms(lx_handle * h, float * ms)
{
    if (!h || !ms)
        return LX_EINVAL;
    if (!h->timed)
        return fail(h, LX_ESTATE, "no timed launch yet");
    int rc = bind(h);
    if (rc)
        return rc;
    LX_HIP(h, hipEventSynchronize(h->ev1));
    LX_HIP(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return LX_OK;
}

int lx_score_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                       uint64_t n, void * d_out_score, void * stream_)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (n == 0)
        return LX_OK;
    if (!d_q_res || !d_s_res || !d_ext || !d_out_score)
        return fail(h, LX_EINVAL, "NULL device pointer");
    int rc = bind(h);
    if (rc)
        return rc;
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : h->stream;
    // geometry from the caller's hints; any geometry is correct for any query length (multi-panel path)
    bool const want_shared = h->opt_query_run != 0 && h->opt_query_run % 8 == 0;
    int const  cfg    = h->opt_max_qlen ? pick_cfg((uint32_t)std::min<uint64_t>(h->opt_max_qlen, 0xffffffffu), want_shared) : 0;
    bool const multi  = h->opt_max_qlen == 0 || h->opt_max_qlen > (uint64_t)lx::score_cfg_panel(cfg);
    // (the packed 16-bit kernel sweeps wide queries in (8,19) panels even where the int32 geometry is a single one)
    bool const wide16 = h->opt_f16 && h->opt_query_run % 16 == 0 && h->opt_query_run != 0 && h->opt_max_qlen > (uint64_t)lx::trace_cfg_panel(2);
    if ((rc = prepare_workspace(h, stream, (multi || wide16 || (h->opt_band && (h->opt_max_qlen == 0 || h->opt_max_qlen > 160))) ? n * ((h->opt_max_slen + 3) & ~3ull) : 0)))
        return rc;
    bool const shared = h->opt_query_run != 0 && (h->opt_query_run % (uint64_t)lx::score_cfg_groups(cfg)) == 0;
    if (!h->in_fused)
    {
        h->phase_ev.clear();
        h->ev_pool_used = 0;
        LX_HIP(h, hipEventRecord(h->ev0, stream));
    }
    // packed-half path: the extensions of a wavefront (or, where two LDS profiles fit, of each half of it) must
    // share their query and the query must fit one panel
    int pair_cfg = -1, pair_share = 0;
    if (h->opt_f16 && shared && h->opt_max_qlen != 0)
    {
        int const pc = lx::score_pair_cfg_for((uint32_t)std::min<uint64_t>(h->opt_max_qlen, 0xffffffffu));
        if (pc >= 0)
        {
            int const      groups   = 64 / lx::score_pair_cfg_group(pc);
            uint64_t const per_wave = 2ull * groups;
            int const      nrows    = ((h->sc_host[slot].alphabet_size + 1 + 3) / 4) * 4;
            if (h->opt_query_run % per_wave == 0)
                pair_cfg = pc;
            else if (groups >= 2 && h->opt_query_run % (per_wave / 2) == 0 && 2 * lx::score_pair_profile_bytes(pc, nrows) <= pair_lds_limit())
            {
                pair_cfg   = pc;
                pair_share = groups / 2;
            }
            else if (h->opt_query_run % 8 == 0) // two profiles do not fit: a 16-lane geometry holds 8 extensions per wavefront
                pair_cfg = lx::score_pair_cfg_for_runs_of_8((uint32_t)std::min<uint64_t>(h->opt_max_qlen, 0xffffffffu));
        }
        else if (h->opt_query_run % 16 == 0 && h->opt_max_slen != 0) // wider than any packed-half geometry
            pair_cfg = kPair16;
    }
    PhaseTimer pt(h, stream, 0);
    if ((rc = launch_score_list(h, slot, d_q_res, d_s_res, d_ext, n, d_out_score, cfg, multi, shared, stream, pair_cfg, pair_share)))
        return rc;
    pt.close();
    if (!h->in_fused)
    {
        LX_HIP(h, hipEventRecord(h->ev1, stream));
        h->timed = true;
    }
    return LX_OK;
}

int lx_set_subjects(lx_handle * h, uint8_t const * s_res, uint64_t s_bytes)
{
    if (!h || (!s_res && s_bytes))
        return LX_EINVAL;
    int rc = bind(h);
    if (rc)
        return rc;
    h->db_bytes = 0;
    if (s_bytes == 0)
        return LX_OK;
    if ((rc = ensure(h, h->d_db, s_bytes + kSlack)))
        return rc;
    LX_HIP(h, hipMemcpyAsync(h->d_db.ptr, s_res, s_bytes, hipMemcpyHostToDevice, h->stream));
    LX_HIP(h, hipMemsetAsync(static_cast<uint8_t *>(h->d_db.ptr) + s_bytes, 0, kSlack, h->stream));
    LX_HIP(h, hipStreamSynchronize(h->stream));
    h->db_bytes = s_bytes;
    return LX_OK;
}

int lx_align_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                       uint64_t n, void * d_out_hsp, void * d_out_ops, void const * d_ops_off, void * stream_)
{
    if (!h)
        return LX_EINVAL;
    if (slot < 0 || slot > 1 || !h->have_sc[slot])
        return fail(h, LX_ESTATE, "scoring slot %d not set", slot);
    if (n == 0)
        return LX_OK;
    if (!d_q_res || !d_s_res || !d_ext || !d_out_hsp || !d_out_ops || !d_ops_off)
        return fail(h, LX_EINVAL, "NULL device pointer");
    int rc = bind(h);
    if (rc)
        return rc;
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : h->stream;
    uint64_t max_q = h->opt_max_qlen, max_s = h->opt_max_slen;
    if (max_q == 0 || max_s == 0)
    {
        // no hints: measure on the device (one small kernel + a stream synchronisation)
        lx::MaxLens * d_ml = reinterpret_cast<lx::MaxLens *>(h->d_ws_top + 2);
        LX_HIP(h, lx::launch_max_lens(static_cast<lx::Extension const *>(d_ext), n, d_ml, stream));
        lx::MaxLens ml{};
        LX_HIP(h, hipMemcpyAsync(&ml, d_ml, sizeof(ml), hipMemcpyDeviceToHost, stream));
        LX_HIP(h, hipStreamSynchronize(stream));
        max_q = std::max<uint64_t>(ml.max_q, 1);
        max_s = std::max<uint64_t>(ml.max_s, 1);
    }
    if ((rc = prepare_workspace(h, stream, max_q > (uint64_t)lx::trace_cfg_panel(1) ? n * ((max_s + 3) & ~3ull) : 0)))
        return rc;
    h->phase_ev.clear();
    h->ev_pool_used = 0;
    LX_HIP(h, hipEventRecord(h->ev0, stream));
    rc = align_dev_impl(h, slot, d_q_res, d_s_res, static_cast<lx::Extension const *>(d_ext), n,
                        static_cast<lx::Hsp *>(d_out_hsp), static_cast<uint8_t *>(d_out_ops),
                        static_cast<uint64_t const *>(d_ops_off), stream, max_q, max_s, 0);
    if (rc)
        return rc;
    LX_HIP(h, hipEventRecord(h->ev1, stream));
    h->timed = true;
    return LX_OK;
}

int lx_extend_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                        uint64_t n, void const * d_min_score, int32_t min_score_all, void * d_out_score,
                        void * d_out_hsp, void * d_out_ops, void const * d_ops_off, void * d_out_count, void * stream_)
{
    if (h && h->count_pending && hipEventQuery(h->ev_count) == hipSuccess)
    {
        // the previous call's survivor count has arrived: the adaptive choice of pass-2 mode (fused_impl) goes by it
        h->count_pending = false;
        if (h->count_n)
            h->surv_frac = (double)h->p_count[1] / (double)h->count_n;
    }
    int const rc = fused_impl(h, slot, d_q_res, d_s_res, d_ext, n, d_min_score, min_score_all, d_out_score, d_out_hsp, d_out_ops, d_ops_off,
                              d_out_count, stream_, 3, false);
    if (rc == LX_OK && n != 0 && h->p_count && !h->count_pending)
    {
        hipStream_t const stream = stream_ ? static_cast<hipStream_t>(stream_) : h->stream;
        if (hipMemcpyAsync(h->p_count, d_out_count, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
            hipEventRecord(h->ev_count, stream) == hipSuccess)
        {
            h->count_pending = true;
            h->count_n       = n;
        }
    }
    return rc;
}

} // extern "C"

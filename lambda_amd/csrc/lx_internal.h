// lx_internal.h -- what lx_api.cpp (handle, options, device entry points, the fused step) and lx_host.cpp (the host-buffer
// entry points and their pipelines) share: the handle, the launchers of the kernel files, small host helpers.  Not part of
// the ABI; include/lambda_ext.h is.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include "../../include/lambda_ext.h"
#include "host/scoring_tables.hpp"
#include "lx_aids.h"
#include "lx_device.h"
#include "lx_host_pool.h"


namespace lx
{
hipError_t launch_score(int cfg, ScoreParams const & p, bool multi, hipStream_t stream);
int        score_cfg_panel(int cfg);
int        score_cfg_groups(int cfg);
int        score_cfg_count();
hipError_t launch_score_pair(int cfg, ScoreParams const & p, hipStream_t stream);
int        score_pair_cfg_for(uint32_t max_qlen);
uint64_t   select_blocks(uint64_t nruns);
int        score_pair_cfg_cols(int cfg);
int        score_pair_cfg_for_runs_of_8(uint32_t max_qlen);
int        score_pair_cfg_group(int cfg);
size_t     score_pair_profile_bytes(int cfg, int nrows);
hipError_t launch_trace_forward(TraceParams const & p, hipStream_t stream);
hipError_t launch_backtrace(TraceParams const & p, hipStream_t stream);
hipError_t launch_max_lens(Extension const * ext, uint64_t n, MaxLens * out, hipStream_t stream);
int        trace_cfg_panel(int cfg);
int        trace_cfg_group(int cfg);
int        trace_cfg_words(int cfg);
hipError_t launch_select(SelectParams const & p, hipStream_t stream);
hipError_t launch_slot_gather(Extension const * ext_all, int32_t const * min_all, int32_t min_score_all, uint32_t const * orig, uint64_t slots,
                              Extension * out_ext, int32_t * out_min, hipStream_t stream);
hipError_t launch_slot_scatter(uint32_t const * orig, uint64_t slots, int32_t const * score, int32_t * score_all, uint32_t * src,
                               uint64_t const * count_ptr, uint64_t cap, hipStream_t stream);
uint64_t   ckpt_slot_dwords(int cfg, uint32_t steps_cap);
uint64_t   ckpt16_slot_dwords(int cfg, uint32_t steps_cap);
hipError_t launch_ckpt_forward(TraceParams const & p, hipStream_t stream);
hipError_t launch_ckpt_backtrace(TraceParams const & p, hipStream_t stream);
hipError_t launch_sweep_pair16(int trace_cfg, ScoreParams const & p, hipStream_t stream);
hipError_t launch_score_pair16(ScoreParams const & p, hipStream_t stream);
hipError_t launch_sweep_pair16_compact(int trace_cfg, ScoreParams const & p, hipStream_t stream);
hipError_t launch_sweep_mq(int trace_cfg, ScoreParams const & p, hipStream_t stream);
size_t     sweep_mq_lds_bytes(int trace_cfg, int nrows, int share);
hipError_t launch_prefilter(PrefilterParams const & p, hipStream_t stream);
hipError_t launch_rle_pack(PackParams const & p, hipStream_t stream);
} // namespace lx

namespace lxi
{

struct DevBuf
{
    void * ptr = nullptr;
    size_t cap = 0;
};

} // namespace lxi
using lxi::DevBuf;

struct lx_handle
{
    int         device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream2 = nullptr;                       // downloads of lx_extend_batch
    bool        timed = false;
    std::string error;
    // lx_extend_batch: host staging that keeps its pages between calls
    std::vector<uint32_t>     xb_idx, xb_src, xb_sel, xb_pos;
    std::vector<uint32_t>     xb_pool_pan, xb_pool_maxs, xb_pool_place, xb_pool_order, xb_pool_key, xb_pool_tmp; // the pool's wavefronts before they are put in launch order
    std::vector<uint8_t>      xb_newrun;
    uint64_t                  xb_stats[4] = {0, 0, 0, 0}; // lx_extend_batch: extensions, slots, cells, cells executed (padding included)
    std::vector<uint64_t>     xb_grp, xb_off, xb_starts;
    std::vector<uint32_t>     xb_sbfirst, xb_sbkey, xb_sborder, xb_sbtmp; // multi-query plan: the pool's sub-blocks of 4 windows
    std::vector<uint8_t>      xb_sbcnt;
    std::vector<uint32_t>     xb_runkey, xb_runorder, xb_runtmp;          // ... the streamed runs in packing order
    std::vector<uint32_t>     xb_slot, xb_wfpan, xb_wfmaxs;               // ... the plan: caller index per slot, panels / longest window per wavefront
    std::vector<lx_extension> xb_ext;
    std::vector<int32_t>      xb_min, xb_score;
    std::vector<uint8_t> ext_ops; // band mode: the ops of the last lx_extend_batch call (handed out by pointer)
    // lx_extend_batch: the ops of the last call, grown without touching what is already there
    struct Bytes
    {
        uint8_t * p   = nullptr;
        size_t    cap = 0;
        uint8_t * data() { return p; }
        void      clear() {}
        bool      grow(size_t bytes) // false: out of memory (the old block and its contents stay)
        {
            if (bytes <= cap)
                return true;
            size_t const want = std::max(bytes + bytes / 2, (size_t)1 << 20);
            void * const np   = std::realloc(p, want);
            if (!np)
                return false;
            p   = static_cast<uint8_t *>(np);
            cap = want;
            return true;
        }
        ~Bytes() { std::free(p); }
    };
    Bytes ext_bytes;
    // lx_extend_batch_list: the survivors of the last call (positions in the caller's list, records, where their codes begin)
    Bytes    res_index, res_hsp, res_off;
    uint64_t res_count = 0, xb_ops_total = 0;
    // lx_extend_batch's two chunks in flight: pinned staging, device buffers, events
    struct Pinned
    {
        void * ptr = nullptr;
        size_t cap = 0;
    };
    struct XbLane
    {
        Pinned     p_ext, p_min, p_score, p_cnt, p_hsp, p_src, p_rle, p_len, p_orig, p_wft;
        DevBuf     d_ext, d_min, d_score, d_hsp, d_ops, d_rle, d_src, d_cnt, d_len, d_orig, d_wft; // (wft: the chunk's lx::WfSlots table)
        hipEvent_t ev_up = nullptr, ev_k = nullptr, ev_cnt = nullptr;
    } xb[2];
    // multi-query plan: the caller's whole list, its cut-offs and the scores in caller order, on the device / in pinned staging
    DevBuf      d_ext_all, d_min_all, d_score_all;
    Pinned      p_all, p_score_all;
    hipStream_t stream3 = nullptr; // uploads of lx_extend_batch (stream2 carries its downloads)
    std::string last_kernel; // human-readable name of the most recent DP kernel geometry (profiling aid)
    std::string last_trace_kernel;
    // per-phase HIP events of the most recent call: phase 0 score, 1 select, 2 trace forward, 3 backtrace
    struct PhaseEv
    {
        int        phase;
        hipEvent_t a, b;
    };
    std::vector<PhaseEv>    phase_ev;      // events recorded by the last call
    std::vector<hipEvent_t> ev_pool;       // reusable timing events
    size_t                  ev_pool_used = 0;

    bool             have_sc[2] = {false, false};
    bool             trace_ok[2] = {false, false};
    bool             b8_ok[2]    = {false, false}; // byte profiles of lx_sweep_mq.hip apply (no substitution dearer than a gap's first character)
    lx_scoring       sc_host[2];
    lx::ScoringDev * sc_dev[2] = {nullptr, nullptr};

    // staging for the host-buffer entry points
    DevBuf d_q, d_s, d_ext, d_out, d_ops, d_opsoff, d_keep, d_trace, d_ends, d_hsp, d_seeds, d_sel_ext, d_sel_src, d_sel_runs, d_sel_score, d_trace_score, d_db;
    // multi-panel carry workspace
    DevBuf     d_ws;
    uint32_t * d_ws_top = nullptr; // [0] = bump pointer, [1] = error flag, [2..3] = MaxLens, [4] = overflow checkpoint slots handed out, [5] = backtrace work queue
    // options
    uint64_t opt_max_qlen  = 0;
    uint64_t opt_query_run = 0;
    uint64_t opt_ws_bytes  = 64ull << 20; // the caller's LX_OPT_WORKSPACE_BYTES
    uint64_t ws_grown      = 0;           // what the calls grew the workspace to by themselves (never shown to the caller)
    uint64_t opt_max_slen  = 0;
    uint64_t opt_trace_bytes = 64ull << 30;
    uint64_t opt_bs_rule   = 0;
    uint64_t opt_f16       = 1;
    uint64_t opt_mq        = 1; // LX_OPT_MQ_SWEEP
    uint64_t opt_iterate_records = 0; // LX_OPT_ITERATE_RECORDS: 0 = the Level-2 driver's records on the device (lx_records.hip), 1 = on the host threads
    // Adaptive pass 2 (LX_OPT_PASS2_MODE = 2): the share of the last batch's extensions that passed the cut-off.  Below
    // LX_OPT_ADAPT_PERMILLE the single sweep's checkpoints are mostly written for nothing, and the step runs as plain pass 1 +
    // checkpoints for the survivors only (mode 1) until the share rises again.  The device entry point never synchronises: it
    // copies the count back behind its kernels and reads it at the next call if it has arrived by then.
    double      surv_frac        = -1.0;   // < 0: unknown
    uint64_t *  p_count          = nullptr; // pinned: [0] = slots, [1] = survivors of the last device-resident call
    hipEvent_t  ev_count         = nullptr;
    uint64_t    count_n          = 0;
    bool        count_pending    = false;
    double      plan_surv_frac   = -1.0;   // the share phase 1 of the current step planned with (phase 2 follows it)
    uint64_t    opt_adapt        = 30;     // LX_OPT_ADAPT_PERMILLE
    // lx_extend_batch: the chunk's slots BY WAVEFRONT (lx::WfSlots, lx_device.h) instead of by region: `dev` = the table of the chunk's
    // wavefronts on the device; the slots of the wavefronts before slot n0 take dw0 uint32 at the trace buffer's start, room for ovf_cap
    // int16-pair overflow slots follows, then the dw1 uint32 of the wavefronts from n0 on.  part: 0 = the chunk in one call, 1 = the
    // first of two calls (sweep of the slots before n0, nothing else), 2 = the second (sweep of the rest, then what follows a sweep,
    // over all slots).  total_dw: what the trace buffer must hold (fixed by the first call).
    struct MqTab
    {
        void const * dev = nullptr;
        uint64_t     n0 = 0, dw0 = 0, dw1 = 0, ovf_cap = 0, total_dw = 0;
        int          part = 0;
    } mq_tab;
    bool     mq_wide_call  = false; // lx_extend_batch: this chunk's sweep writes int16-pair slots (many windows of the last chunks scored beyond the compact codes)
    double   mq_decl_frac  = 0.0;   // ... the share of the last multi-panel chunk's windows that the compact sweep declined
    int      mq_cfg_call   = 0; // lx_extend_batch: the strip geometry (trace cfg) it chose for this call's chunks (0 = fused_impl picks per chunk)
    uint64_t opt_extend_chunk = 0; // LX_OPT_EXTEND_CHUNK: extensions per chunk of lx_extend_batch's pipeline (0 = default)
    uint64_t opt_band      = 0; // LX_OPT_BAND: half width in diagonals, 0 = full rectangle (the reference's BandOff)
    int32_t const * band_dev = nullptr;  // lx_set_band_centres_dev: the caller's device array for the *_dev calls
    std::vector<int32_t> band_host;      // lx_set_band_centres: centres of the next host-buffer call's extensions
    DevBuf   d_band;                     // ... uploaded
    uint64_t opt_pass2     = 2; // LX_OPT_PASS2_MODE: 0 = direction bits (lx_trace.hip), 1 = checkpoints (lx_ckpt.hip), 2 = single sweep; each where applicable
    uint64_t db_bytes      = 0; // lx_set_subjects: size of the resident subject buffer (0 = none)
    // Level 2 on the device (lx_level2_host.cpp): the resident sequence sets (lx_set_queries, lx_set_subject_seqs) and the buffers of
    // the list work -- sort words (two of each), digit counts, scan values, windows
    struct Level2
    {
        std::vector<uint64_t> q_off, s_off, s_len;
        std::vector<uint32_t> q_len, q_evlen, evlens; // evlens: the distinct e-value lengths of the query set
        uint64_t              q_hash = 0, s_hash = 0, dup_before = 0; // content hashes of the sets lx_iterate_matches made resident; hits_duplicate before the last list
        uint64_t              q_bytes = 0, max_slen = 0, s_extent = 0; // s_extent: where the last subject ends in the residue buffer
        uint32_t              max_evlen = 0;
        int                   q_frames  = 1;
        DevBuf                d_qres, d_qoff, d_qlen, d_qband, d_qevlen, d_soff, d_slen;
        DevBuf                d_pair[2], d_s0[2], d_hist, d_head, d_tail, d_tot, d_win, d_cut, d_cnt, d_up, d_plan, d_wf, d_fp; // (d_fp: workspace of the free-packing plan)
        Pinned                p_cnt, p_win, p_up;
        std::vector<lx_extension> ext;   // host copies of the window list, its cut-offs and scores
        std::vector<int32_t>      min, score;
        std::vector<uint32_t>     wf_pan, wf_maxs; // a device plan's wavefronts
        hipEvent_t                ev_win = nullptr; // the window list has arrived on the host
        hipEvent_t                ev_rank[2] = {nullptr, nullptr}; // the window list is complete / the rank kernel (second stream) is through
        // records on the device (lx_records.hip): the call's survivors as the pipeline's chunks left them (alignment, window, where the
        // codes begin), the key / scan / record buffers, the host-made tables of the e-value
        uint32_t              max_qlen = 0;
        DevBuf                d_qevidx, d_surv_hsp, d_surv_src, d_surv_codes, d_listat, d_rec, d_reccodes, d_reccnt, d_tilekeep, d_tileops, d_pre, d_exp, d_rank;
        uint32_t              rank_too_long = 0; // (lx_records.hip: rec_launch_rank's flag, downloaded with the plan)
        Pinned                p_reccnt, p_reccodes, p_rows; // p_rows: a range's finished rows on their way into the result
        Pinned                p_plan; // what the host reads of a device plan: [rank flag + probe windows][columns per lane][longest window] per wavefront
        std::vector<uint64_t> rec_codes;              // where the records' run-length codes begin (host copy)
        uint64_t              surv_total = 0, surv_cap = 0;
        bool                  surv_on_device = false; // the last pipeline call kept its survivors on the device
        bool                  surv_by_range  = false; // ... and handed them over range by range (ResidentInput::ChunkRecords)
        std::vector<uint64_t> cut_wf;                 // the ranges' first wavefronts
        double                exp_lambda = 0;         // the scheme d_exp was made for
        uint32_t              exp_n      = 0;
    } l2;
    bool     keep_phase_events = false; // lx_extend_batch: the phase events of every chunk of the call stay (lx_last_phase_ms sums them)
    bool     in_fused      = false; // lx_extend_batch_dev is driving the sub-steps (it owns ev0/ev1 and the phase list)
};

namespace lxi
{

int        fail(lx_handle * h, int code, char const * fmt, ...);
hipEvent_t pool_event(lx_handle * h);

// Wall-clock marks of the host-buffer entry points, printed when LX_HOST_TIMING is set (development aid).
struct HostMarks
{
    bool                                                               on;
    char const *                                                       what;
    std::chrono::steady_clock::time_point                              t0, last;
    std::string                                                        line;
    explicit HostMarks(char const * w) : on(lx::dev_aids().host_timing), what(w)
    {
        t0 = last = std::chrono::steady_clock::now();
    }
    void mark(char const * name)
    {
        if (!on)
            return;
        auto const now = std::chrono::steady_clock::now();
        char       buf[96];
        snprintf(buf, sizeof(buf), " %s %.1f", name, std::chrono::duration<double, std::milli>(now - last).count());
        line += buf;
        last = now;
    }
    ~HostMarks()
    {
        if (on)
            fprintf(stderr, "[lx host ms] %s:%s | total %.1f\n", what, line.c_str(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// RAII-less phase bracket: records a start event now, the end event on close()
struct PhaseTimer
{
    lx_handle * h;
    hipStream_t s;
    int         phase;
    hipEvent_t  a = nullptr, b = nullptr;
    PhaseTimer(lx_handle * h_, hipStream_t s_, int phase_) : h(h_), s(s_), phase(phase_)
    {
        if (h->phase_ev.size() < 512)
        {
            a = pool_event(h);
            b = pool_event(h);
            if (a && b)
                (void)hipEventRecord(a, s);
        }
    }
    void close()
    {
        if (a && b)
        {
            (void)hipEventRecord(b, s);
            h->phase_ev.push_back({phase, a, b});
        }
    }
};

#define LX_HIP(h, call)                                                                                         \
    do                                                                                                          \
    {                                                                                                           \
        hipError_t _e = (call);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return fail((h), _e == hipErrorOutOfMemory ? LX_ENOMEM : LX_EHIP, "%s failed: %s", #call,           \
                        hipGetErrorString(_e));                                                                 \
    } while (0)

int    ensure(lx_handle * h, DevBuf & b, size_t bytes);
int    bind(lx_handle * h);
size_t pair_lds_limit();
int    pick_cfg(uint32_t qlen, bool shared);
int    ckpt_cfg_for(uint64_t max_q, bool packed16 = false);
int    check_async_error(lx_handle * h);
int    error_for_flag(lx_handle * h, uint32_t flag);

// ---- the plan of a fused step (plan_step in lx_api.cpp is the one place that makes it)
struct SchemeFacts
{
    int  alph = 0, gap_open = 0, gap_extend = 0, smax_entry = 0;
    bool trace_ok = false, b8_ok = false;
};
struct StepOptions
{
    uint64_t max_qlen = 0, max_slen = 0, query_run = 0, pass2 = 2, mq = 1, f16 = 1, band = 0, trace_bytes = 0, n = 0, adapt = 0;
    int      mq_cfg_call = 0;
    bool     mq_wide     = false; // the multi-query sweep writes int16-pair slots (lx_sweep_mq.hip: WIDE)
    double   surv_frac   = -1.0;
};
enum SweepFamily
{
    kNoSweep        = 0, // pass 1, then pass 2 on the survivors (modes 0 / 1)
    kHalfSweep      = 1, // lx_score_f16.hip score_pair_kernel<G,C,true>: packed half, compact codes, one panel
    kI16CompactWide = 2, // lx_score_i16.hip sweep_pair16_kernel<8,19,true,true,true>: packed int16, compact codes, several panels
    kI16Pairs       = 3, // lx_score_i16.hip sweep_pair16_kernel<G,C,MULTI>: packed int16, int16-pair slots
    kInt32Sweep     = 4, // lx_ckpt.hip ckpt_forward_kernel<G,C,false,MULTI> alone
    kMqSweep        = 5  // lx_sweep_mq.hip sweep_mq_kernel<C,MULTI>: byte profiles, up to four queries per wavefront
};
struct StepPlan
{
    bool        shared = false, sweep = false, adapted = false, compact = false, may_decline = true;
    bool        packed = false; // a packed sweep's launch structure: a spare slot behind the batch's, an overflow area for what it declines
    bool        wide   = false; // the multi-query sweep with int16-pair slots
    SweepFamily family = kNoSweep;
    int         cfg = 0, share = 0;
    uint32_t    steps = 0, panels = 1;
    uint64_t    stride = 0, stride32 = 0, ovf_cap = 0;
};
StepPlan plan_step(SchemeFacts const & sc, StepOptions const & o);
void     describe_plan(StepPlan const & pl, char * buf, size_t len);
int    mq_cfg_for(uint64_t max_q);
int    launch_score_list(lx_handle * h, int slot, void const * d_q, void const * d_s, void const * d_ext, uint64_t n, void * d_out, int cfg,
                         bool multi, bool shared, hipStream_t stream, int pair_cfg = -1, int pair_share = 0);
int    prepare_workspace(lx_handle * h, hipStream_t stream, uint64_t pairs_hint = 0);

// padding of q/s staging buffers so that clamped / prefetching loads never leave the allocation
constexpr size_t kSlack = 256;
constexpr uint64_t kExtendChunk = 640ull << 10; // extensions per chunk of lx_extend_batch's pipeline unless LX_OPT_EXTEND_CHUNK says otherwise
constexpr int kPair16    = 100; // launch_score_list's pair_cfg: the packed 16-bit integer kernel, any query width
constexpr int kPair16Bin = 8;   // its bin among the packed geometries of lx_score_batch

// Subject side of a host-buffer call: either the caller's buffer, uploaded into d_s, or -- s_res == NULL, s_bytes == 0
// after lx_set_subjects -- the resident copy.
struct SubjectRef
{
    void *   dev   = nullptr;
    uint64_t bytes = 0;
    bool     upload = false;
};

int resolve_subjects(lx_handle * h, uint8_t const * s_res, uint64_t s_bytes, SubjectRef & out);

// lx_extend_batch's additions to the fused step: ops slots of one size instead of an offset per extension, the survivors'
// ops run-length packed into a dense stream (lx_pack.hip), a copy of the survivor list's original indices
struct FusedExtra
{
    uint64_t             ops_stride = 0;
    uint8_t *            d_rle      = nullptr;
    unsigned long long * d_rle_top  = nullptr;
    uint64_t             rle_cap    = 0;
    uint32_t *           d_src_out  = nullptr; // [survivor list capacity]
    uint32_t *           d_rle_len  = nullptr; // [survivor list capacity]: code bytes per position
};

int align_dev_impl(lx_handle * h, int slot, void const * d_q, void const * d_s, lx::Extension const * d_ext, uint64_t n, lx::Hsp * d_hsp,
                   uint8_t * d_ops, uint64_t const * d_ops_off, hipStream_t stream, uint64_t max_q, uint64_t max_s, int share_slots,
                   uint32_t const * d_src = nullptr, uint64_t const * d_count = nullptr, int32_t const * d_score_in = nullptr,
                   bool by_pos = false, uint64_t ops_stride = 0);
int fused_impl(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext, uint64_t n, void const * d_min_score,
               int32_t min_score_all, void * d_out_score, void * d_out_hsp, void * d_out_ops, void const * d_ops_off, void * d_out_count,
               void * stream_, int phases, bool by_pos, FusedExtra const * fx = nullptr);

// what stands on the device already when the Level-2 driver (lx_level2_host.cpp) calls the extension pipeline
struct ResidentInput
{
    void const * d_q       = nullptr; // query residues
    uint64_t     q_bytes   = 0;
    void const * d_ext_all = nullptr; // the caller's list (lx::Extension) and its cut-offs, in the order of the host's copies
    void const * d_min_all = nullptr;
    // a plan made on the device (the solo packing): the slot list (16 per wavefront: position in the list, bit 31 = filler), and
    // on the host per wavefront the columns per lane its widest query sweeps and its longest window, the strip geometry
    // (trace cfg), the list's cells
    uint32_t const * d_plan  = nullptr;
    uint64_t         nwf     = 0;
    uint32_t const * wf_pan  = nullptr;
    uint32_t const * wf_maxs = nullptr;
    int              mq_cfg  = 0;
    bool             free_packing = false; // the device plan is the free packing (pairs of one query, at most four queries per wavefront), not the solo one
    uint64_t         cells   = 0;
    // the survivors stay on the device (lx_handle::Level2::d_surv_*; taken where the plan above is served: surv_on_device says so),
    // the scores too (h->d_score_all); want_codes: their run-length codes come down into the handle's code bytes
    bool keep_on_device = false, want_codes = true;
    // Records chunk by chunk (lx_level2_host.cpp): the plan's chunks are RANGES of the query-sorted window list (cut_wf: the wavefront
    // each range begins with, n_ranges + 1 entries), so a chunk's survivors are a contiguous piece of the result -- `enqueue` queues
    // the records kernels behind the chunk's own (its survivor list stands in the lane's buffers: no copy), `collect` is called when
    // they are through, while the NEXT chunk computes: rows and columns of all but the last range come down beside the sweeps.
    struct ChunkRecords
    {
        uint64_t const * cut_wf   = nullptr;
        uint64_t         n_ranges = 0;
        std::function<int(uint64_t range, void const * d_hsp, void const * d_src, void const * d_count, uint64_t cap)> enqueue;
        std::function<int(uint64_t range, uint64_t code_base, bool gpu_busy)>                                         collect; // gpu_busy: another chunk computes meanwhile
    };
    ChunkRecords const * chunk_records = nullptr;
};
// band mode's plain path of the host-buffer entry points (lx_host_batch.cpp; what: 0 = scores, 1 = alignments of known scores, 2 = both)
int  host_banded(lx_handle * h, int slot, int what, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                 lx_extension const * ext, uint64_t n, int32_t const * known_score, int32_t const * min_score, int32_t min_score_all, int32_t * out_score,
                 lx_hsp * out_hsp, uint8_t * caller_ops, uint64_t const * caller_ops_off, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes);
bool solo_plan_applies(lx_handle const * h, int slot);
bool free_plan_applies(lx_handle const * h, int slot);
int  extend_list_resident(lx_handle * h, int slot, ResidentInput const & ri, lx_extension const * ext, uint64_t n, int32_t const * min_score,
                         int32_t * out_score, lx_survivor_list * out);

// [off, off + len) inside a buffer of `bytes`, written so that offsets near 2^64 cannot wrap past the test
inline bool lx_slice_ok(uint64_t off, uint64_t len, uint64_t bytes)
{
    return len <= bytes && off <= bytes - len;
}


} // namespace lxi

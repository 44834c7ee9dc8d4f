// lx_api.cpp -- host side of the C ABI declared in include/lambda_ext.h (compiled with hipcc).
//
// Owns: device selection, the HIP stream, device copies of the scoring schemes, grow-only staging buffers
// for the host-buffer entry points, the multi-panel carry workspace, HIP-event timing of the kernel sequence,
// and the binning of extensions into kernel geometries.  No DP arithmetic happens on the host and there is no
// CPU fallback: without a usable gfx950 device every entry point returns an error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include "../../include/lambda_ext.h"
#include "host/scoring_tables.hpp"
#include "lx_device.h"

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
uint64_t   ckpt_slot_dwords(int cfg, uint32_t steps_cap);
uint64_t   ckpt16_slot_dwords(int cfg, uint32_t steps_cap);
hipError_t launch_ckpt_forward(TraceParams const & p, hipStream_t stream);
hipError_t launch_ckpt_backtrace(TraceParams const & p, hipStream_t stream);
hipError_t launch_sweep_pair16(int trace_cfg, ScoreParams const & p, hipStream_t stream);
hipError_t launch_score_pair16(ScoreParams const & p, hipStream_t stream);
hipError_t launch_sweep_pair16_compact(int trace_cfg, ScoreParams const & p, hipStream_t stream);
hipError_t launch_prefilter(PrefilterParams const & p, hipStream_t stream);
hipError_t launch_rle_pack(PackParams const & p, hipStream_t stream);
} // namespace lx

static_assert(sizeof(lx_extension) == sizeof(lx::Extension), "ABI mismatch");
static_assert(sizeof(lx_hsp) == sizeof(lx::Hsp), "ABI mismatch");
static_assert(sizeof(lx_extension) == 24, "ABI mismatch");

namespace
{

thread_local std::string g_create_error;

struct DevBuf
{
    void * ptr = nullptr;
    size_t cap = 0;
};

} // namespace

struct lx_handle
{
    int         device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream2 = nullptr;                       // backtrace of chunk k overlaps the forward kernel of chunk k+1
    hipEvent_t  evF[2] = {nullptr, nullptr}, evB[2] = {nullptr, nullptr}, evS = nullptr;
    bool        timed = false;
    std::string error;
    // lx_extend_batch: host staging that keeps its pages between calls
    std::vector<uint32_t>     xb_idx, xb_src, xb_sel, xb_pos;
    std::vector<uint8_t>      xb_newrun;
    uint64_t                  xb_stats[4] = {0, 0, 0, 0}; // lx_extend_batch: extensions, slots, cells, cells executed (padding included)
    std::vector<uint64_t>     xb_grp, xb_off;
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
        void      grow(size_t bytes)
        {
            if (bytes <= cap)
                return;
            size_t const want = std::max(bytes + bytes / 2, (size_t)1 << 20);
            p                 = static_cast<uint8_t *>(std::realloc(p, want));
            cap               = p ? want : 0;
        }
        ~Bytes() { std::free(p); }
    } ext_bytes;
    // lx_extend_batch's two chunks in flight: pinned staging, device buffers, events
    struct Pinned
    {
        void * ptr = nullptr;
        size_t cap = 0;
    };
    struct XbLane
    {
        Pinned     p_ext, p_min, p_score, p_cnt, p_hsp, p_src, p_rle, p_len;
        DevBuf     d_ext, d_min, d_score, d_hsp, d_ops, d_rle, d_src, d_cnt, d_len;
        hipEvent_t ev_up = nullptr, ev_k = nullptr, ev_cnt = nullptr;
    } xb[2];
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
    uint64_t opt_extend_chunk = 0; // LX_OPT_EXTEND_CHUNK: extensions per chunk of lx_extend_batch's pipeline (0 = default)
    uint64_t opt_band      = 0; // LX_OPT_BAND: half width in diagonals, 0 = full rectangle (the reference's BandOff)
    int32_t const * band_dev = nullptr;  // lx_set_band_centres_dev: the caller's device array for the *_dev calls
    std::vector<int32_t> band_host;      // lx_set_band_centres: centres of the next host-buffer call's extensions
    DevBuf   d_band;                     // ... uploaded
    uint64_t opt_pass2     = 2; // LX_OPT_PASS2_MODE: 0 = direction bits (lx_trace.hip), 1 = checkpoints (lx_ckpt.hip), 2 = single sweep; each where applicable
    uint64_t db_bytes      = 0; // lx_set_subjects: size of the resident subject buffer (0 = none)
    bool     in_fused      = false; // lx_extend_batch_dev is driving the sub-steps (it owns ev0/ev1 and the phase list)
};

namespace
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

// Wall-clock marks of the host-buffer entry points, printed when LX_HOST_TIMING is set (development aid).
struct HostMarks
{
    bool                                                               on;
    char const *                                                       what;
    std::chrono::steady_clock::time_point                              t0, last;
    std::string                                                        line;
    explicit HostMarks(char const * w) : on(std::getenv("LX_HOST_TIMING") != nullptr), what(w)
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
        if (h->phase_ev.size() < 64)
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

int ensure(lx_handle * h, DevBuf & b, size_t bytes)
{
    if (bytes <= b.cap)
        return LX_OK;
    if (b.ptr)
    {
        LX_HIP(h, hipStreamSynchronize(h->stream));
        LX_HIP(h, hipFree(b.ptr));
        b.ptr = nullptr;
        b.cap = 0;
    }
    size_t const want = bytes + bytes / 4 + 4096;
    LX_HIP(h, hipMalloc(&b.ptr, want));
    b.cap = want;
    return LX_OK;
}

int bind(lx_handle * h)
{
    LX_HIP(h, hipSetDevice(h->device));
    return LX_OK;
}

// padding of q/s staging buffers so that clamped / prefetching loads never leave the allocation
constexpr size_t kSlack = 256;

// Smallest panel that holds the query; 8-lane geometries need one shared profile per wavefront (8 profile slots
// per wavefront would not fit the LDS budget), so without sharing only the 16/32/64-lane geometries are used.
// LDS a wavefront of the packed-half kernel may spend on two query profiles (one per half wavefront: query runs of 8)
size_t pair_lds_limit()
{
    static size_t const v = []() -> size_t
    {
        char const * e = getenv("LX_PAIR_LDS_LIMIT"); // development aid
        return e ? (size_t)atoll(e) : (size_t)24 * 1024; // protein profiles too: 12.8 vs 14.0 ms (pass 1), 16.6 vs 19.1 ms (sweep) for runs of 8
    }();
    return v;
}

int pick_cfg(uint32_t qlen, bool shared)
{
    static int const forced = []() // development aid: measure a geometry on a shape it is not picked for
    {
        char const * e = getenv("LX_FORCE_SCORE_CFG");
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0)
        return forced;
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
int ckpt_cfg_for(uint64_t max_q, bool packed16 = false)
{
    uint64_t const p1 = (uint64_t)lx::trace_cfg_panel(1), p2 = (uint64_t)lx::trace_cfg_panel(2);
    if (max_q <= p1)
        return 1;
    if (max_q <= p2)
        return 2;
    static int const forced = []() // development aid
    {
        char const * e = getenv("LX_FORCE_CKPT_CFG");
        return e ? atoi(e) : 0;
    }();
    if (forced == 1 || forced == 2)
        return forced;
    // (the packed 16-bit sweep is bound by its checkpoint bytes: the 19-column strips of (8,19) store fewer boundary
    // columns -- 400 aa: 0.0134 ms per padded column against 0.0156 for (16,13))
    double const f1 = packed16 ? 0.0134 : 0.070, f2 = packed16 ? 0.0156 : 0.062;
    double const c1 = (double)((max_q + p1 - 1) / p1 * p1) * f1, c2 = (double)((max_q + p2 - 1) / p2 * p2) * f2;
    return c1 < c2 ? 1 : 2;
}

int check_async_error(lx_handle * h)
{
    uint32_t flags[2] = {0, 0};
    LX_HIP(h, hipMemcpyAsync(flags, h->d_ws_top, sizeof(flags), hipMemcpyDeviceToHost, h->stream));
    LX_HIP(h, hipStreamSynchronize(h->stream));
    if (flags[1] == 1)
        return fail(h, LX_EOVERFLOW, "multi-panel carry workspace exhausted (raise LX_OPT_WORKSPACE_BYTES)");
    if (flags[1] == 2)
        return fail(h, LX_ESTATE, "LX_OPT_QUERY_RUN promise violated: extensions of one wavefront use different queries");
    if (flags[1] == 3)
        return fail(h, LX_EOVERFLOW, "an extension exceeds the trace slot bounds (LX_OPT_MAX_QLEN / LX_OPT_MAX_SLEN too small, or a subject window beyond the pass-2 limit)");
    if (flags[1] == 4)
        return fail(h, LX_EOVERFLOW, "single sweep: no checkpoint slot left for an extension the packed-half kernel declined "
                                     "(raise LX_OPT_TRACE_BYTES, or set LX_OPT_PASS2_MODE to 1)");
    if (flags[1] != 0)
        return fail(h, flags[1] == 5 ? LX_EOVERFLOW : LX_EHIP, "device reported error flag %u", flags[1]);
    return LX_OK;
}

// One kernel sequence for a device-resident extension list whose queries all fit geometry `cfg`
// (or need the multi-panel path when wider).
constexpr int kPair16    = 100; // launch_score_list's pair_cfg: the packed 16-bit integer kernel, any query width
constexpr int kPair16Bin = 7;   // its bin among the packed geometries of lx_score_batch

int launch_score_list(lx_handle * h, int slot, void const * d_q, void const * d_s, void const * d_ext, uint64_t n,
                      void * d_out, int cfg, bool multi, bool shared, hipStream_t stream, int pair_cfg = -1, int pair_share = 0)
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
int prepare_workspace(lx_handle * h, hipStream_t stream, uint64_t pairs_hint = 0)
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

} // namespace

// [off, off + len) inside a buffer of `bytes`, written so that offsets near 2^64 cannot wrap past the test
static inline bool lx_slice_ok(uint64_t off, uint64_t len, uint64_t bytes)
{
    return len <= bytes && off <= bytes - len;
}

// a few host threads for the per-extension loops of the host-buffer entry point (none below a quarter million items)
static unsigned host_threads(uint64_t n)
{
    if (n < 250000)
        return 1;
    static unsigned const avail = []()
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        unsigned c = sched_getaffinity(0, sizeof(set), &set) == 0 ? (unsigned)CPU_COUNT(&set) : std::thread::hardware_concurrency();
        if (char const * e = getenv("LX_HOST_THREADS"))
            c = (unsigned)std::max(1, atoi(e));
        return std::max(1u, std::min(c, 8u));
    }();
    return avail;
}

// A few persistent host threads (started on first use): the per-extension loops of the host-buffer entry points are spread
// over them; spawning threads per loop would cost more than the loops of a pipeline chunk.
namespace
{
class HostPool
{
    std::vector<std::thread>       workers_;
    std::mutex                     m_;
    std::condition_variable        cv_, done_;
    std::function<void(unsigned)>  job_;
    unsigned                       want_ = 0, gen_ = 0, running_ = 0;
    bool                           stop_ = false;

    void loop(unsigned id)
    {
        unsigned seen = 0;
        for (;;)
        {
            std::function<void(unsigned)> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (gen_ != seen && id < want_); });
                if (stop_)
                    return;
                seen = gen_;
                job  = job_;
            }
            job(id);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--running_ == 0)
                    done_.notify_all();
            }
        }
    }

public:
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread & t : workers_)
            t.join();
    }
    // runs f(1) .. f(nthreads - 1) on the workers and f(0) on the caller; returns when all are done
    void run(unsigned nthreads, std::function<void(unsigned)> f)
    {
        static std::mutex           callers; // one parallel loop at a time (handles on several host threads share the pool)
        std::lock_guard<std::mutex> one(callers);
        while (workers_.size() + 1 < nthreads)
        {
            unsigned const id = (unsigned)workers_.size() + 1;
            workers_.emplace_back([this, id] { loop(id); });
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_     = f;
            want_    = nthreads;
            running_ = nthreads - 1;
            ++gen_;
        }
        cv_.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return running_ == 0; });
    }
};
HostPool & host_pool()
{
    static HostPool p;
    return p;
}
} // namespace

template <typename F>
static void parallel_ranges(uint64_t n, unsigned nthreads, F && body)
{
    if (nthreads <= 1 || n < 2 * (uint64_t)nthreads)
    {
        for (unsigned t = 0; t < nthreads; ++t) // keep the per-thread slots of the callers meaningful
            body(t, t == 0 ? 0 : n, n);
        return;
    }
    uint64_t const step = (n + nthreads - 1) / nthreads;
    host_pool().run(nthreads, [&body, step, n](unsigned t) { body(t, std::min(n, t * step), std::min(n, (t + 1) * step)); });
}

static int host_banded(lx_handle * h, int slot, int what, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                       lx_extension const * ext, uint64_t n, int32_t const * known_score, int32_t const * min_score,
                       int32_t min_score_all, int32_t * out_score, lx_hsp * out_hsp, uint8_t * caller_ops,
                       uint64_t const * caller_ops_off, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes);

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
    if (char const * m = getenv("LX_PASS2_MODE")) // default of LX_OPT_PASS2_MODE, for A/B runs of unmodified callers
        h->opt_pass2 = (uint64_t)std::min(std::max(atoi(m), 0), 2);
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
    for (hipEvent_t * ev : {&h->evF[0], &h->evF[1], &h->evB[0], &h->evB[1], &h->evS})
        if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
    if ((e = hipMalloc(reinterpret_cast<void **>(&h->d_ws_top), 8 * sizeof(uint32_t))) != hipSuccess)
        return bail("hipMalloc", e);
    if ((e = hipMemset(h->d_ws_top, 0, 8 * sizeof(uint32_t))) != hipSuccess)
        return bail("hipMemset", e);
    for (int s = 0; s < 2; ++s)
        if ((e = hipMalloc(reinterpret_cast<void **>(&h->sc_dev[s]), sizeof(lx::ScoringDev))) != hipSuccess)
            return bail("hipMalloc", e);
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
    for (hipStream_t st : {h->stream2, h->stream3})
        if (st)
        {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (auto & ln : h->xb)
    {
        for (DevBuf * b : {&ln.d_ext, &ln.d_min, &ln.d_score, &ln.d_hsp, &ln.d_ops, &ln.d_rle, &ln.d_src, &ln.d_cnt, &ln.d_len})
            if (b->ptr)
                (void)hipFree(b->ptr);
        for (lx_handle::Pinned * b : {&ln.p_ext, &ln.p_min, &ln.p_score, &ln.p_cnt, &ln.p_hsp, &ln.p_src, &ln.p_rle, &ln.p_len})
            if (b->ptr)
                (void)hipHostFree(b->ptr);
        for (hipEvent_t ev : {ln.ev_up, ln.ev_k, ln.ev_cnt})
            if (ev)
                (void)hipEventDestroy(ev);
    }
    for (hipEvent_t ev : {h->evF[0], h->evF[1], h->evB[0], h->evB[1], h->evS})
        if (ev)
            (void)hipEventDestroy(ev);
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

// Subject side of a host-buffer call: either the caller's buffer, uploaded into d_s, or -- s_res == NULL, s_bytes == 0
// after lx_set_subjects -- the resident copy.
struct SubjectRef
{
    void *   dev   = nullptr;
    uint64_t bytes = 0;
    bool     upload = false;
};
static int resolve_subjects(lx_handle * h, uint8_t const * s_res, uint64_t s_bytes, SubjectRef & out)
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
    size_t const nbins = (size_t)ncfg * 2 + 8;
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


// ---- pass 2 ------------------------------------------------------------------------------------------

// Runs pass 2 over a device-resident list of `n` extension slots, in chunks sized to the trace budget.
// src / d_count are set by the fused path (slots compacted by launch_select): results are then written to
// out_hsp[src[slot]] / ops_off[src[slot]] and slots beyond *d_count are skipped on the device.
static int align_dev_impl(lx_handle * h, int slot, void const * d_q, void const * d_s, lx::Extension const * d_ext,
                          uint64_t n, lx::Hsp * d_hsp, uint8_t * d_ops, uint64_t const * d_ops_off, hipStream_t stream,
                          uint64_t max_q, uint64_t max_s, int share_slots, uint32_t const * d_src = nullptr,
                          uint64_t const * d_count = nullptr, int32_t const * d_score_in = nullptr, bool by_pos = false,
                          uint64_t ops_stride = 0)
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
    bool const ckpt = !h->opt_band && h->opt_pass2 >= 1 && share_slots >= 4 && (uint64_t)smax_entry * std::min(max_q, max_s) < 32000 && max_s <= 65535; // (longer windows: direction bits)
    // Direction bits beyond one panel: the 16-lane geometry that pads the query less ((16,13) needs the shared profile).
    auto padded = [&](int c) { return (max_q + lx::trace_cfg_panel(c) - 1) / lx::trace_cfg_panel(c) * lx::trace_cfg_panel(c); };
    int const cfg = (h->opt_band && !(share_slots >= 4 && max_q <= (uint64_t)lx::trace_cfg_panel(1))) ? 0 // (band mode: (8,19) or generic)
                    : (share_slots >= 4 && max_q <= (uint64_t)lx::trace_cfg_panel(1)) ? 1
                    : (share_slots >= 4 && max_q <= (uint64_t)lx::trace_cfg_panel(2)) ? 2
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
    // Two trace buffers, so that the backtrace of chunk k may run on stream2 while the forward kernel of chunk k+1
    // runs on `stream`.  Measured on MI355X (config 2) the overlap buys nothing -- both kernels saturate the chip
    // (46.5 ms/step serial vs 46.9 ms overlapped) -- so it is off unless LX_TRACE_OVERLAP=1.
    // Without the overlap one buffer is enough, so a chunk may use the whole budget: as few launches (and kernel
    // tails) as the budget allows.  In the fused path `n` is the capacity of the survivor list; launches beyond the
    // device-side count exit at once.
    bool const     overlap     = getenv("LX_TRACE_OVERLAP") && atoi(getenv("LX_TRACE_OVERLAP")) != 0;
    uint64_t const nbuf        = overlap ? 2 : 1;
    uint64_t       chunk       = std::max<uint64_t>(1, h->opt_trace_bytes / nbuf / std::max<uint64_t>(per_ext, 1));
    uint64_t const want_chunks = getenv("LX_TRACE_CHUNKS") ? (uint64_t)atoi(getenv("LX_TRACE_CHUNKS")) : 1;
    chunk                      = std::min<uint64_t>(chunk, n / std::max<uint64_t>(want_chunks, 1) + 8);
    hipStream_t const bstream  = overlap ? h->stream2 : stream;
    chunk                      = std::max<uint64_t>(8, (chunk + 7) / 8 * 8);
    int rc;
    if ((rc = ensure(h, h->d_trace, nbuf * chunk * per_ext)) || (rc = ensure(h, h->d_ends, nbuf * chunk * sizeof(lx::EndCell))))
        return rc;
    LX_HIP(h, hipEventRecord(h->evS, stream));
    LX_HIP(h, hipStreamWaitEvent(h->stream2, h->evS, 0));
    uint64_t nchunks = 0;
    for (uint64_t c0 = 0; c0 < n; c0 += chunk, ++nchunks)
    {
        int const       b = overlap ? (int)(nchunks & 1) : 0; // one buffer without the overlap (stream order protects it)
        lx::TraceParams p{};
        p.q_res          = static_cast<uint8_t const *>(d_q);
        p.s_res          = static_cast<uint8_t const *>(d_s);
        p.ext            = d_ext + c0;
        p.n              = std::min<uint64_t>(chunk, n - c0);
        p.sc             = h->sc_dev[slot];
        p.trace          = static_cast<uint32_t *>(h->d_trace.ptr) + (uint64_t)b * chunk * stride;
        p.slot_stride    = stride;
        p.steps_cap      = steps_cap;
        p.panels_cap     = panels_cap;
        p.ends           = static_cast<lx::EndCell *>(h->d_ends.ptr) + (uint64_t)b * chunk;
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
        if (nchunks >= 2) // buffer b is free once the backtrace of chunk k-2 has finished
            LX_HIP(h, hipStreamWaitEvent(stream, h->evB[b], 0));
        if (panels_cap > 1) // each chunk starts with an empty carry workspace
            LX_HIP(h, hipMemsetAsync(h->d_ws_top, 0, sizeof(uint32_t), stream));
        PhaseTimer ptf(h, stream, 2);
        LX_HIP(h, ckpt ? lx::launch_ckpt_forward(p, stream) : lx::launch_trace_forward(p, stream));
        ptf.close();
        LX_HIP(h, hipEventRecord(h->evF[b], stream));
        LX_HIP(h, hipStreamWaitEvent(bstream, h->evF[b], 0));
        PhaseTimer ptb(h, bstream, 3);
        LX_HIP(h, ckpt ? lx::launch_ckpt_backtrace(p, bstream) : lx::launch_backtrace(p, bstream));
        ptb.close();
        LX_HIP(h, hipEventRecord(h->evB[b], bstream));
        {
            char buf[96];
            if (ckpt)
                snprintf(buf, sizeof(buf), "lx::ckpt_forward_kernel<%d,%d>", G, P / G);
            else
                snprintf(buf, sizeof(buf), "lx::trace_forward_kernel<%d,%d,%s>", G, P / G, panels_cap > 1 ? "true" : "false");
            h->last_trace_kernel = buf;
        }
    }
    // rejoin: everything queued on `stream` after this call sees the finished backtraces
    for (int b = 0; b < 2 && (uint64_t)b < nchunks; ++b)
        LX_HIP(h, hipStreamWaitEvent(stream, h->evB[b], 0));
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


// ---- fused: pass 1 -> survivor selection -> pass 2, all on the device ------------------------------------

// phases: 1 = pass 1 (or the sweep) + selection, 2 = pass 2 (or the sweep's backtrace), 3 = both.  by_pos: records and
// ops offsets are indexed by the position in the survivor list instead of by extension (the host entry point assigns
// compact ops offsets between the two phases and downloads only the survivors' records).
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

static int fused_impl(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext, uint64_t n,
                      void const * d_min_score, int32_t min_score_all, void * d_out_score, void * d_out_hsp, void * d_out_ops,
                      void const * d_ops_off, void * d_out_count, void * stream_, int phases, bool by_pos,
                      FusedExtra const * fx = nullptr)
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

    if (phases & 1)
    {
        h->phase_ev.clear();
        h->ev_pool_used = 0;
        LX_HIP(h, hipEventRecord(h->ev0, stream));
    }
    bool const shared = h->opt_query_run != 0 && h->opt_query_run % 8 == 0;

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
    bool     half_sweep = false, may_decline = true;
    int const nrows_sc = ((h->sc_host[slot].alphabet_size + 1 + 3) / 4) * 4;
    if (h->opt_pass2 == 2 && shared && h->trace_ok[slot] && !h->opt_band)
    {
        // one panel of (8,19) or (16,13); wider queries: several (16,13) panels, int32 sweep
        sweep_cfg    = ckpt_cfg_for(h->opt_max_qlen, h->opt_f16 && h->opt_query_run % 16 == 0);
        // short queries (<= 104 columns, e.g. 100-residue reads): the (8,13) geometry where the packed-half sweep applies --
        // a third fewer padded columns than (8,19)
        bool const half_ok = h->opt_f16 && -h->sc_host[slot].gap_open <= lx::kC16MaxGap && h->sc_host[slot].gap_open <= h->sc_host[slot].gap_extend;
        static bool const no_narrow = getenv("LX_NO_NARROW_SWEEP") != nullptr; // A/B aid
        if (sweep_cfg == 1 && half_ok && !no_narrow && h->opt_max_qlen <= (uint64_t)lx::trace_cfg_panel(3) &&
            (h->opt_query_run % 16 == 0 || 2 * lx::score_pair_profile_bytes(1, nrows_sc) + 64 * 8 * 4 <= pair_lds_limit()))
            sweep_cfg = 3;
        sweep_panels = (uint32_t)std::max<uint64_t>(1, (h->opt_max_qlen + lx::trace_cfg_panel(sweep_cfg) - 1) / lx::trace_cfg_panel(sweep_cfg));
        int smax_entry = 0;
        for (int a = 0; a < h->sc_host[slot].alphabet_size; ++a)
            for (int b = 0; b < h->sc_host[slot].alphabet_size; ++b)
                smax_entry = std::max<int>(smax_entry, h->sc_host[slot].matrix[a * LX_ALPH + b]);
        if (sweep_cfg != 0 && (uint64_t)smax_entry * std::min(h->opt_max_qlen, h->opt_max_slen) < 32000 && h->opt_max_slen <= 65535)
        {
            int const G    = lx::trace_cfg_group(sweep_cfg);
            sweep_steps    = (uint32_t)((h->opt_max_slen + G - 1 + 15) & ~15ull);
            sweep_stride32 = (uint64_t)sweep_panels * lx::ckpt_slot_dwords(sweep_cfg, sweep_steps);
            // Packed half precision where its geometry matches the checkpoint layout ((8,19): 16 extensions of one query per
            // wavefront, or runs of 8 with one query per half wavefront where two LDS profiles fit, i.e. for the small
            // alphabets; (16,13): 8 extensions) and a gap's first character costs at most 31 (the compact checkpoint codes
            // of Ckpt16Layout).  Wavefronts it declines leave the sentinel -1; the int32 kernel fills those in.
            half_sweep = h->opt_f16 && sweep_panels == 1 && -h->sc_host[slot].gap_open <= lx::kC16MaxGap &&
                         h->sc_host[slot].gap_open <= h->sc_host[slot].gap_extend &&
                         (((sweep_cfg == 1 || sweep_cfg == 3) && h->opt_query_run % 16 == 0) || sweep_cfg == 2);
            if (h->opt_f16 && sweep_panels == 1 && -h->sc_host[slot].gap_open <= lx::kC16MaxGap &&
                h->sc_host[slot].gap_open <= h->sc_host[slot].gap_extend && (sweep_cfg == 1 || sweep_cfg == 3) && !half_sweep && h->opt_query_run % 8 == 0 &&
                2 * lx::score_pair_profile_bytes(sweep_cfg == 3 ? 1 : 0, nrows_sc) + 64 * 8 * 4 <= pair_lds_limit())
            {
                half_sweep  = true;
                sweep_share = 4;
            }
            if (half_sweep)
            {
                // compact slots for the batch (+ the spare slot idle halves write to), int16-pair slots for what the
                // packed kernel declines in whatever the budget leaves
                sweep_stride = lx::ckpt16_slot_dwords(sweep_cfg, sweep_steps);
                sweep        = (n + 1) * sweep_stride * 4 <= h->opt_trace_bytes;
                // (the packed kernel's exactness gate, lx_score_f16.hip: it cannot decline when even the worst query passes)
                int64_t const worst = (int64_t)h->opt_max_qlen * std::max(smax_entry, 0) +
                                      (int64_t)(-h->sc_host[slot].gap_extend) * (sweep_steps + G + 2) +
                                      (smax_entry - h->sc_host[slot].gap_extend) + 2; // (ScoringDev::smax = largest entry - ge)
                may_decline = worst > 2046;
                if (sweep && may_decline)
                    ovf_cap = std::min<uint64_t>(n, (h->opt_trace_bytes - (n + 1) * sweep_stride * 4) / (sweep_stride32 * 4));
            }
            else
            {
                sweep_stride = sweep_stride32;
                sweep        = n * sweep_stride * 4 <= h->opt_trace_bytes;
            }
        }
    }
    if (sweep && (phases & 1))
    {
        uint64_t const batch_dw = half_sweep ? (n + 1) * sweep_stride : n * sweep_stride;
        if ((rc = ensure(h, h->d_trace, (batch_dw + ovf_cap * sweep_stride32) * 4)) || (rc = ensure(h, h->d_ends, n * sizeof(lx::EndCell))))
            return rc;
        if ((rc = prepare_workspace(h, stream, sweep_panels > 1 ? n * ((h->opt_max_slen + 3) & ~3ull) : 0)))
            return rc;
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
        p.shared_profile = 64 / lx::trace_cfg_group(sweep_cfg); // every wavefront holds one query
        p.cfg            = sweep_cfg;
        if (half_sweep)
        {
            p.ovf        = p.trace + batch_dw;
            p.ovf_stride = sweep_stride32;
            p.ovf_cap    = (uint32_t)ovf_cap;
            p.ovf_count  = h->d_ws_top + 4;
        }
        int const sweep_pair = sweep_cfg == 1 ? 0 : sweep_cfg == 3 ? 1 : 5; // pair geometry with the same (G, C): (8,19) / (8,13) / (16,13)
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
            sp1.pair_share  = sweep_share;
            static bool const int_sweep = getenv("LX_SWEEP_INT") != nullptr; // A/B: the compact sweep in the integer domain
            if (int_sweep && sweep_share == 0)
                LX_HIP(h, lx::launch_sweep_pair16_compact(sweep_cfg, sp1, stream));
            else
                LX_HIP(h, lx::launch_score_pair(sweep_pair, sp1, stream));
            p.fixup = 1;
        }
        // No packed-half sweep (queries wider than a panel, gap costs beyond the compact codes, ...): the packed int16
        // kernel writes the int16-pair slots of the int32 kernel, two extensions per lane group; what fails its range
        // test is left to the int32 launch.  16 extensions of one query per wavefront at (8,19), 8 at (16,13).
        bool const i16_sweep = !half_sweep && h->opt_f16 && !getenv("LX_NO_I16_SWEEP") &&
                               h->opt_query_run % (sweep_cfg == 1 ? 16 : 8) == 0 && sweep_cfg != 3;
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
        char buf[160];
        int const nameG = lx::trace_cfg_group(sweep_cfg), nameC = lx::trace_cfg_panel(sweep_cfg) / lx::trace_cfg_group(sweep_cfg);
        if (half_sweep && may_decline)
            snprintf(buf, sizeof(buf), "lx::score_pair_kernel<%d,%d,true> (single sweep; + int32 fix-up lx::ckpt_forward_kernel<%d,%d,false>)",
                     nameG, nameC, nameG, nameC);
        else if (half_sweep)
            snprintf(buf, sizeof(buf), "lx::score_pair_kernel<%d,%d,true> (single sweep)", nameG, nameC);
        else if (i16_sweep)
            snprintf(buf, sizeof(buf), "lx::sweep_pair16_kernel<%d,%d,%s> (single sweep; + int32 fix-up lx::ckpt_forward_kernel<%d,%d,false>)",
                     nameG, nameC, sweep_panels > 1 ? "true" : "false", nameG, nameC);
        else
            snprintf(buf, sizeof(buf), "lx::ckpt_forward_kernel<%d,%d,false> (single sweep)", nameG, nameC);
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
    uint32_t const pad_to = (shared && !sweep) ? 4u : 1u;
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
            p.ovf        = p.trace + (n + 1) * sweep_stride; // int16-pair slots of what the packed kernel declined
            p.ovf_stride = sweep_stride32;
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
                        static_cast<uint64_t const *>(d_ops_off), stream, h->opt_max_qlen, h->opt_max_slen, shared ? 4 : 0,
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


// ---- band mode on host buffers ---------------------------------------------------------------------------------------
// Band mode (LX_OPT_BAND) is a semantic option, not a fast path: it runs one int32 kernel geometry and direction bits for
// pass 2, so the host-buffer entry points skip the binning / grouping of their full-rectangle versions -- the list goes to
// the device as it is, the centres (lx_set_band_centres) with it.
//   what = 0: lx_score_batch, 1: lx_align_batch (caller's ops slots), 2: lx_extend_batch (ops slots of the handle)
static int host_banded(lx_handle * h, int slot, int what, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
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

int lx_extend_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                        uint64_t n, void const * d_min_score, int32_t min_score_all, void * d_out_score,
                        void * d_out_hsp, void * d_out_ops, void const * d_ops_off, void * d_out_count, void * stream_)
{
    return fused_impl(h, slot, d_q_res, d_s_res, d_ext, n, d_min_score, min_score_all, d_out_score, d_out_hsp, d_out_ops, d_ops_off,
                      d_out_count, stream_, 3, false);
}

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
                           lx_hsp * out_hsp, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes, bool want_rle)
{
    int rc = bind(h);
    if (rc)
        return rc;
    SubjectRef sref;
    if ((rc = resolve_subjects(h, s_res, s_bytes, sref)))
        return rc;
    s_bytes = sref.bytes;
    HostMarks hm(want_rle ? "lx_extend_batch_rle" : "lx_extend_batch");

    // ---- validate; is the list grouped by query (lambda's lists are sorted by query)?  The loops over the list are spread
    // over a few host threads: at millions of extensions per call they would otherwise cost more than the kernels.
    unsigned const nthreads = host_threads(n);
    struct Part
    {
        uint64_t live = 0, bad = ~0ull;
        bool     monotone = true;
    };
    std::vector<Part> parts(nthreads);
    parallel_ranges(n, nthreads,
                    [&](unsigned t, uint64_t lo, uint64_t hi)
                    {
                        Part &   pt   = parts[t];
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
                                out_score[i]   = 0;
                                out_hsp[i]     = lx_hsp{};
                                out_ops_off[i] = 0;
                                continue;
                            }
                            if (prev != ~0ull && x.q_off < ext[prev].q_off)
                                pt.monotone = false;
                            prev = i;
                            ++pt.live;
                        }
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
    if (live == 0)
        return LX_OK;
    std::vector<uint32_t> & idx = h->xb_idx;
    idx.resize(live);
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
    std::vector<uint8_t> & newrun = h->xb_newrun;
    newrun.resize(live + 1);
    parallel_ranges(live, nthreads,
                    [&](unsigned, uint64_t lo, uint64_t hi)
                    {
                        for (uint64_t k = lo; k < hi; ++k)
                            newrun[k] = (k == 0 || !same_slice(idx[k], idx[k - 1])) ? 1 : 0;
                    });
    newrun[live] = 1;
    // Mixed query lengths (a real seed list; the synthetic batches have one): a chunk runs the kernel geometry of its longest
    // query, so runs are dealt to geometry classes first -- one panel of 152 columns, one of 208, two / three / ... panels of
    // 152 -- and every class goes through the pipeline by itself.  Inside a run the windows are ordered by length (merged
    // windows are up to 3 x longer: src/search_algo.hpp:1153-1157), so that a wavefront's 16 windows take about as many steps
    // each -- the reason the reference sorts its SIMD batches (:1229-1235).  Results are scattered by original index anyway.
    {
        auto qclass = [](uint32_t lq) -> uint32_t { return lq <= 104 ? 0u : lq <= 152 ? 1u : lq <= 208 ? 2u : 2u + (lq + 151) / 152; };
        uint32_t cmin = ~0u, cmax = 0;
        bool     ragged_s = false;
        for (uint64_t k = 0; k < live; ++k)
        {
            if (newrun[k])
            {
                uint32_t const c = qclass(ext[idx[k]].q_len);
                cmin = std::min(cmin, c);
                cmax = std::max(cmax, c);
            }
            else if (ext[idx[k]].s_len != ext[idx[k - 1]].s_len)
                ragged_s = true;
        }
        static bool const no_classes = getenv("LX_EXTEND_NO_CLASSES") != nullptr, no_sort = getenv("LX_EXTEND_NO_SORT") != nullptr; // A/B aids
        if (cmin != cmax && !no_classes)
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
        if (ragged_s && !no_sort)
        {
            std::vector<uint64_t> starts;
            for (uint64_t k = 0; k <= live; ++k)
                if (newrun[k])
                    starts.push_back(k);
            parallel_ranges(starts.size() - 1, nthreads,
                            [&](unsigned, uint64_t lo, uint64_t hi)
                            {
                                for (uint64_t r = lo; r < hi; ++r)
                                    std::sort(idx.begin() + starts[r], idx.begin() + starts[r + 1],
                                              [&](uint32_t a, uint32_t b) { return ext[a].s_len != ext[b].s_len ? ext[a].s_len < ext[b].s_len : a < b; });
                            });
        }
    }
    hm.mark("validate");

    // ---- the caller's option values come back on every exit; the streams are drained before anything is torn down
    struct Guard
    {
        lx_handle * h;
        uint64_t    qlen, slen, run;
        ~Guard()
        {
            (void)hipStreamSynchronize(h->stream);
            (void)hipStreamSynchronize(h->stream2);
            (void)hipStreamSynchronize(h->stream3);
            h->opt_max_qlen  = qlen;
            h->opt_max_slen  = slen;
            h->opt_query_run = run;
        }
    } const guard{h, h->opt_max_qlen, h->opt_max_slen, h->opt_query_run};

    if ((rc = ensure(h, h->d_q, q_bytes + kSlack)))
        return rc;
    if (q_bytes)
        LX_HIP(h, hipMemcpyAsync(h->d_q.ptr, q_res, q_bytes, hipMemcpyHostToDevice, h->stream));
    if (sref.upload)
        LX_HIP(h, hipMemcpyAsync(sref.dev, s_res, s_bytes, hipMemcpyHostToDevice, h->stream));

    uint64_t const chunk_target = h->opt_extend_chunk ? std::max<uint64_t>(h->opt_extend_chunk, 1024) : []() -> uint64_t
    {
        static uint64_t const v = []() -> uint64_t
        {
            char const * e = getenv("LX_EXTEND_CHUNK"); // development aid
            return e ? (uint64_t)std::max(1024ll, atoll(e)) : 640ull << 10;
        }();
        return v;
    }();
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

    // ---- chunk k0 .. k1 of the ordered list -> padded slots in lane L's pinned staging -> uploads and kernels queued
    auto enqueue = [&](int L, uint64_t k0, uint64_t k1) -> int
    {
        auto const          t0 = now();
        lx_handle::XbLane & ln = h->xb[L];
        XbPrep &            pr = prep[L];
        pr.k0 = k0;
        pr.k1 = k1;
        // runs of one query slice; padded to 16 slots (one query per wavefront of the 8-lane packed geometry) or, when the
        // queries have few windows each, to 8 (one query per half wavefront: ~1.4 x the time per slot) -- whichever is less work
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
        uint64_t const kRun    = (slots8 * 7 < slots16 * 5) ? 8 : 16;
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
        uint64_t const panel = max_q <= 104 ? 104 : max_q <= 152 ? 152 : max_q <= 208 ? 208 : 152, lanes = panel == 208 ? 16 : 8;
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
        auto const t1 = now();
        t_prep += ms(t0, t1);
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
        LX_HIP(h, hipStreamWaitEvent(h->stream, ln.ev_up, 0));
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
        if ((rc2 = fused_impl(h, slot, h->d_q.ptr, sref.dev, ln.d_ext.ptr, slots, ln.d_min.ptr, 0, ln.d_score.ptr, ln.d_hsp.ptr, ln.d_ops.ptr,
                              nullptr, d_cnt, h->stream, 3, true, &fx)))
            return rc2;
        LX_HIP(h, hipEventRecord(ln.ev_k, h->stream));
        // what has a size the host knows goes back at once; records and codes follow when the counts have arrived
        LX_HIP(h, hipStreamWaitEvent(h->stream2, ln.ev_k, 0));
        LX_HIP(h, hipMemcpyAsync(ln.p_cnt.ptr, d_cnt, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream2));
        LX_HIP(h, hipMemcpyAsync(ln.p_score.ptr, ln.d_score.ptr, slots * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream2));
        LX_HIP(h, hipEventRecord(ln.ev_cnt, h->stream2));
        in_flight[L] = true;
        t_issue += ms(t1, now());
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
        if (count > pr.cap_sel)
            return fail(h, LX_ESTATE, "survivor list longer than its capacity");
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
        h->ext_bytes.grow(total + 16);
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
    while (k0 < live)
    {
        uint64_t k1 = std::min<uint64_t>(live, k0 + chunk_target);
        while (k1 < live && !newrun[k1]) // never cut a query's run
            ++k1;
        {
            // ... and never mix geometry classes (the list is class-major): cut where the class changes
            auto qclass = [](uint32_t lq) -> uint32_t { return lq <= 104 ? 0u : lq <= 152 ? 1u : lq <= 208 ? 2u : 2u + (lq + 151) / 152; };
            uint32_t const c0 = qclass(ext[idx[k0]].q_len);
            if (!getenv("LX_EXTEND_NO_CLASSES") && qclass(ext[idx[k1 - 1]].q_len) != c0)
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
    if ((rc = check_async_error(h)))
        return rc;
    hm.mark("pipeline");
    if (hm.on)
        fprintf(stderr, "[lx host ms]   pipeline of %d chunks: prepare %.1f, issue %.1f, wait for the GPU %.1f, unpack %.1f (lengths %.1f, offsets %.1f)\n", c, t_prep, t_issue,
                t_wait, t_unpack, t_u1, t_u2);
    *out_ops       = h->ext_bytes.data();
    *out_ops_bytes = ops_total;
    return LX_OK;
}

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
                           out_ops_bytes, false);
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
                           out_ops_bytes, true);
}

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

// lx_plan_free.hip -- the plan of the multi-query sweep's FREE PACKING for a window list in device memory (gfx950 only).
//
// What iterateMatchesFullSimd does between its list work and the DP (/root/reference/src/search_algo.hpp:1229-1235: the list of
// alignments is sorted by the slices' lengths so that the windows of one SIMD batch take about as many steps) as kernels, for the
// lists lx_sweep_mq.hip serves with up to four queries per wavefront -- protein lists: a lane group's two windows share a query,
// a wavefront's eight lane groups hold windows of at most four queries.  The same plan lx_host.cpp's extend_pipeline makes on the
// host threads for lists it is handed in host memory, made where the Level-2 driver's window list stands (lx_level2_host.cpp):
//   * RUNS: the windows of one query slice (the list is sorted by query), cut every 512 list positions so that the per-run work
//     below is bounded -- inside a run the windows are ranked by length, longest first, by counting;
//   * POOL: a run's windows clearly longer than its median (the merged windows, up to three times the ordinary length,
//     :1153-1157) and those clearly shorter (windows clipped at their subject's end, :919-938 -- a tenth of a protein list), each
//     kind filled up with the run's nearest ordinary windows to QUADS of four; the quads of the whole list are sorted by (columns
//     per lane their query sweeps, length) and dealt four to a wavefront: a long window stretches three companions, not fifteen,
//     and a short one is swept beside windows of its own length instead of idling -- and forcing the sweep's checked steps on its
//     wavefront -- for the rest of a long one's rows;
//   * STREAM: everything else, run by run in order of (columns per lane, longest streamed window), PAIR by pair into wavefronts
//     that close at eight pairs or before a fifth query.  The closing rule is sequential; it runs per SHARE of 64 runs (one
//     wavefront of the plan kernel per share, a share starts a new wavefront), which leaves one wavefront per share partly empty;
//   * RANGES: the list is cut into up to eight ranges of queries (the pipeline's chunks: a range's records are made while the
//     next range is swept); range is the major key of both sorts, and the plan's wavefronts are laid out range by range, each as
//     [its pool | its stream], so that a range is a contiguous piece of the plan.
// Every window then knows its slot; what is left empty (the second half of a run's last pair, the end of a closed wavefront, of a
// range's last pool wavefront) is filled with a copy of the slot before it, marked as filler (bit 31): it costs what the window
// costs and never survives -- how the reference pads its SIMD batches (:1063-1067).
// Integer list work, HBM- and launch-bound; no DP here.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "lx_level2.h"

namespace lx
{

namespace
{

#include "lx_scan.h"

constexpr uint32_t kFpRunCut   = 512; // a run is cut at every multiple of this many list positions
constexpr uint32_t kFpShare    = 64;  // sorted runs per share of the stream's closing rule
constexpr uint32_t kFpEmpty    = 0xffffffffu;

// device-side facts of a plan (FpArgs::layout)
struct FpLayout
{
    uint32_t nruns, nquads, nwf, overflow;        // overflow: a slot beyond the plan's capacity (the bound of fp_wavefront_bound is wrong)
    uint32_t run_first[kFpMaxRanges + 1];         // first run of each range (run_first[nranges] = nruns): also the ranges' places in the sorted run order
    uint32_t quad_first[kFpMaxRanges + 1];        // first quad of each range: also its place in the sorted quad order
    uint32_t share_first[kFpMaxRanges + 1];       // first share of each range
    uint32_t pool_wf0[kFpMaxRanges + 1];          // first wavefront of each range's pool
    uint32_t stream_wf0[kFpMaxRanges + 1];        // first wavefront of each range's stream (minus the shares' own offsets: see fp_layout_kernel)
    uint32_t range_wf[kFpMaxRanges + 1];          // first wavefront of each range; [nranges] = nwf
};

struct FpWork
{
    uint32_t * rid;       // [n]   run of list position i (= of sorted position i: a run's windows keep its positions)
    uint32_t * ord;       // [n]   window at sorted position k (longest first inside a run)
    uint32_t * run_start; // [rm + 1]
    uint32_t * run_npool; // [rm]  windows of the run that stand in the pool: its longest (low half) and its shortest (high half)
    uint32_t * run_nq;    // [rm]  its quads
    uint32_t * run_qoff;  // [rm]  first quad (in run order)
    uint32_t * run_np;    // [rm]  its streamed pairs
    uint32_t * run_g;     // [rm]  first pair slot inside its share's wavefronts
    uint32_t * run_spos;  // [rm]  place in the sorted run order
    uint64_t * qkey[2];   // [qm]  sort words of the quads
    uint64_t * qidx[2];
    uint32_t * qinv;      // [qm]  place of quad q in the sorted order
    uint64_t * rkey[2];   // [rm]  sort words of the runs
    uint64_t * ridx[2];
    uint32_t * wfcnt;     // [sm]  wavefronts of a share
    uint32_t * wfoff;     // [sm]  ... before it (over all shares of the plan)
    uint32_t * block_tot; // scans
    uint32_t * ghist;     // sorts
    FpLayout * layout;
};

struct FpRanges
{
    uint32_t n;
    uint32_t cut[kFpMaxRanges + 1]; // list positions: range r = [cut[r], cut[r + 1])
};

__device__ __forceinline__ uint32_t fp_cols_per_lane(uint32_t lq, int C, int no_narrow)
{
    uint32_t const panel = 8u * (uint32_t)C;
    uint32_t const P     = max(1u, (lq + panel - 1) / panel);
    int const      rem   = (int)(max(lq, 1u) - (P - 1) * panel);
    int const      code  = no_narrow ? 0 : narrow_code_for(C, 8, rem);
    return min(0xfffu, (P - 1) * (uint32_t)C + (uint32_t)narrow_strip_cols(C, code));
}

// sort key of a quad / a run inside its range: widest first, longest first (16 bits: lengths in steps of 8 up to 1024, of 64 beyond --
// a wavefront's steps are rounded to 16 anyway --, columns per lane up to 255: beyond either the order is coarse, the plan valid)
__device__ __forceinline__ uint32_t fp_key16(uint32_t pan, uint32_t len)
{
    uint32_t const lc = len < 1024u ? len >> 3 : 128u + min(127u, (len - 1024u) >> 6);
    return ((255u - min(pan, 255u)) << 8) | (255u - lc);
}

__device__ __forceinline__ uint32_t fp_range_of(FpRanges const & rg, uint32_t i)
{
    uint32_t r = 0;
#pragma unroll
    for (uint32_t k = 1; k < kFpMaxRanges; ++k)
        r += (k < rg.n && i >= rg.cut[k]) ? 1u : 0u;
    return r;
}

// ---- runs: head flags -> run numbers and run starts
struct RunHeadVal
{
    Extension const * ext;
    __device__ uint32_t operator()(uint64_t i) const
    {
        uint64_t const l = i ? i - 1 : 0;
        return (i == 0 || (i & (kFpRunCut - 1)) == 0 || ext[i].q_off != ext[l].q_off || ext[i].q_len != ext[l].q_len) ? 1u : 0u;
    }
};
struct RunHeadOut
{
    uint32_t * rid;
    uint32_t * run_start;
    __device__ void operator()(uint64_t i, uint32_t incl, uint32_t ex) const
    {
        rid[i] = incl - 1u;
        if (incl != ex)
            run_start[incl - 1u] = (uint32_t)i;
    }
};

__global__ void fp_runs_done_kernel(FpWork w, uint64_t tiles, uint32_t n, FpRanges rg)
{
    uint32_t const nruns = w.block_tot[tiles];
    w.layout->nruns      = nruns;
    w.layout->overflow   = 0;
    w.run_start[nruns]   = n;
    for (uint32_t r = 0; r < rg.n; ++r)
        w.layout->run_first[r] = rg.cut[r] < n ? w.rid[rg.cut[r]] : nruns; // (a range begins where the query changes: with a run)
    w.layout->run_first[rg.n] = nruns;
}

// ---- a window's place in its run, longest first (ties: list order), by counting over the run (at most kFpRunCut windows)
__global__ __launch_bounds__(256) void fp_rank_kernel(Extension const * ext, uint32_t n, FpWork w)
{
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint32_t const r = w.rid[i], a = w.run_start[r], b = w.run_start[r + 1];
    uint32_t const mine = ext[i].s_len;
    uint32_t       d    = 0;
    for (uint32_t j = a; j < b; ++j)
    {
        uint32_t const l = ext[j].s_len;
        d += (l > mine || (l == mine && j < i)) ? 1u : 0u;
    }
    w.ord[a + d] = i;
}

// ---- per run: the pool's share (long windows in quads), the streamed pairs, the run's sort word
__global__ __launch_bounds__(256) void fp_run_kernel(Extension const * ext, uint32_t rm, int C, int no_narrow, FpRanges rg, FpWork w)
{
    uint32_t const r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rm)
        return;
    uint32_t const nruns = w.layout->nruns;
    if (r >= nruns)
    {
        // (beyond the list's runs: sorted behind every range's, takes no slot)
        w.run_npool[r] = w.run_nq[r] = w.run_np[r] = 0;
        w.rkey[0][r] = ((uint64_t)rg.n << 16) | 0xffffu;
        w.ridx[0][r] = r;
        return;
    }
    uint32_t const a = w.run_start[r], b = w.run_start[r + 1], c = b - a;
    auto const     len = [&](uint32_t d) { return ext[w.ord[a + d]].s_len; };
    uint32_t const med = len(c / 2), slack = max(8u, med / 8u);
    // first place whose window is not longer than `bound` (the run is sorted, longest first)
    auto const first_not_above = [&](uint32_t bound)
    {
        uint32_t lo = 0, hi = c;
        while (lo < hi)
        {
            uint32_t const mid = (lo + hi) / 2;
            if (len(mid) > bound)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    uint32_t const nlong  = first_not_above(med + slack);
    uint32_t const nshort = med > slack ? c - first_not_above(med - slack - 1u) : 0u; // windows shorter than med - slack
    uint32_t const phi = min(c, (nlong + 3u) / 4u * 4u), plo = min(c - phi, (nshort + 3u) / 4u * 4u);
    uint32_t const nq = (phi + 3u) / 4u + (plo + 3u) / 4u, np = (c - phi - plo + 1u) / 2u;
    uint32_t const pan = fp_cols_per_lane(ext[w.ord[a]].q_len, C, no_narrow);
    w.run_npool[r] = phi | (plo << 16);
    w.run_nq[r]    = nq;
    w.run_np[r]    = np;
    // (a run that stands in the pool altogether sorts behind its width's streamed runs)
    w.rkey[0][r] = ((uint64_t)fp_range_of(rg, a) << 16) | (np ? fp_key16(pan, len(phi)) : (fp_key16(pan, 0) | 0xffu));
    w.ridx[0][r] = r;
}

struct RunQuadVal
{
    uint32_t const * run_nq;
    __device__ uint32_t operator()(uint64_t r) const { return run_nq[r]; }
};
struct RunQuadOut
{
    uint32_t * run_qoff;
    __device__ void operator()(uint64_t r, uint32_t, uint32_t ex) const { run_qoff[r] = ex; }
};

__global__ void fp_quads_done_kernel(FpWork w, uint64_t tiles, FpRanges rg)
{
    uint32_t const nq  = w.block_tot[tiles];
    w.layout->nquads   = nq;
    uint32_t const nr  = w.layout->nruns;
    for (uint32_t r = 0; r <= rg.n; ++r)
    {
        uint32_t const first = w.layout->run_first[r];
        w.layout->quad_first[r] = first < nr ? w.run_qoff[first] : nq;
    }
}

// ---- the quads' sort words (every entry of the bound first: the padding sorts behind every range's quads)
__global__ __launch_bounds__(256) void fp_quad_init_kernel(uint32_t qm, uint32_t nranges, FpWork w)
{
    uint32_t const q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= qm)
        return;
    w.qkey[0][q] = ((uint64_t)nranges << 16) | 0xffffu;
    w.qidx[0][q] = q;
}
__global__ __launch_bounds__(256) void fp_quad_keys_kernel(Extension const * ext, uint32_t n, int C, int no_narrow, FpRanges rg, FpWork w)
{
    uint32_t const k = blockIdx.x * blockDim.x + threadIdx.x; // sorted position
    if (k >= n)
        return;
    uint32_t const r = w.rid[k], a = w.run_start[r], d = k - a, c = w.run_start[r + 1] - a;
    uint32_t const phi = w.run_npool[r] & 0xffffu, plo = w.run_npool[r] >> 16, d0 = c - plo; // [0, phi): the long quads, [d0, c): the short ones
    uint32_t       quad;
    if (d < phi && (d & 3u) == 0)
        quad = d / 4u;
    else if (d >= d0 && ((d - d0) & 3u) == 0)
        quad = (phi + 3u) / 4u + (d - d0) / 4u;
    else
        return;
    Extension const x = ext[w.ord[k]];
    w.qkey[0][w.run_qoff[r] + quad] = ((uint64_t)fp_range_of(rg, a) << 16) | fp_key16(fp_cols_per_lane(x.q_len, C, no_narrow), x.s_len);
}
__global__ __launch_bounds__(256) void fp_quad_inverse_kernel(uint64_t const * qidx_sorted, uint32_t qm, FpWork w)
{
    uint32_t const p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= qm)
        return;
    uint32_t const q = (uint32_t)qidx_sorted[p];
    if (q < w.layout->nquads)
        w.qinv[q] = p;
}

// ---- the stream's closing rule, one wavefront per share of kFpShare sorted runs of ONE range: where every run's first pair goes
// (in pair slots, eight per wavefront, from the share's first wavefront) and how many wavefronts the share takes
__global__ __launch_bounds__(64) void fp_share_kernel(uint64_t const * ridx_sorted, uint32_t nranges, FpWork w)
{
    // which range's share is this block?  (the ranges' shares follow one another: share_first)
    uint32_t const s = blockIdx.x;
    uint32_t       rng = 0;
    for (uint32_t k = 1; k < nranges; ++k)
        rng += s >= w.layout->share_first[k] ? 1u : 0u;
    if (s >= w.layout->share_first[nranges])
    {
        if (threadIdx.x == 0)
            w.wfcnt[s] = 0;
        return;
    }
    uint32_t const first = w.layout->run_first[rng] + (s - w.layout->share_first[rng]) * kFpShare, end = w.layout->run_first[rng + 1];
    uint32_t const at    = first + threadIdx.x;
    uint32_t const r     = at < end ? (uint32_t)ridx_sorted[at] : kFpEmpty;
    uint32_t const np    = r != kFpEmpty ? w.run_np[r] : 0u;
    uint32_t       pos = 0, nq = 0, mine = 0;
    for (int j = 0; j < (int)kFpShare; ++j)
    {
        uint32_t const npj = (uint32_t)__shfl((int)np, j);
        if (npj == 0)
            continue; // (wave-uniform)
        if ((pos & 7u) == 0)
            nq = 0;
        if (nq == 4)
        {
            pos = (pos + 7u) & ~7u;
            nq  = 0;
        }
        if (j == (int)threadIdx.x)
            mine = pos;
        uint32_t const e = pos + npj;
        nq               = (e >> 3) > (pos >> 3) ? ((e & 7u) ? 1u : 0u) : nq + 1u;
        pos              = e;
    }
    if (r != kFpEmpty)
    {
        w.run_g[r]    = mine;
        w.run_spos[r] = s; // (its share)
    }
    if (threadIdx.x == 0)
        w.wfcnt[s] = (pos + 7u) >> 3;
}

// shares per range: from the ranges' run counts (one thread)
__global__ void fp_shares_kernel(uint32_t nranges, FpWork w)
{
    uint32_t s = 0;
    for (uint32_t r = 0; r < nranges; ++r)
    {
        w.layout->share_first[r] = s;
        s += (w.layout->run_first[r + 1] - w.layout->run_first[r] + kFpShare - 1) / kFpShare;
    }
    w.layout->share_first[nranges] = s;
}

struct ShareVal
{
    uint32_t const * wfcnt;
    __device__ uint32_t operator()(uint64_t s) const { return wfcnt[s]; }
};
struct ShareOut
{
    uint32_t * wfoff;
    __device__ void operator()(uint64_t s, uint32_t, uint32_t ex) const { wfoff[s] = ex; }
};

// the plan's layout: range by range [pool | stream] (one thread)
__global__ void fp_layout_kernel(uint32_t nranges, uint32_t cap_wf, FpWork w)
{
    FpLayout & L        = *w.layout;
    uint32_t       base = 0;
    for (uint32_t r = 0; r < nranges; ++r)
    {
        uint32_t const pool_wf = (L.quad_first[r + 1] - L.quad_first[r] + 3u) / 4u;
        // (every share index up to share_first[nranges] has a scan value: the scan runs over the bound sm, shares beyond the plan's count 0)
        uint32_t const off0 = w.wfoff[L.share_first[r]], off1 = w.wfoff[L.share_first[r + 1]];
        L.range_wf[r]   = base;
        L.pool_wf0[r]   = base;
        L.stream_wf0[r] = base + pool_wf - off0; // (+ wfoff[share] = the share's first wavefront)
        base += pool_wf + (off1 - off0);
    }
    L.range_wf[nranges] = base;
    L.nwf               = base;
    if (base > cap_wf)
        L.overflow = 1;
}

// ---- every window's slot
__global__ __launch_bounds__(256) void fp_place_kernel(uint32_t n, FpRanges rg, uint32_t cap_wf, uint32_t * plan, FpWork w)
{
    uint32_t const k = blockIdx.x * blockDim.x + threadIdx.x; // sorted position
    if (k >= n)
        return;
    FpLayout const & L = *w.layout;
    if (L.overflow)
        return;
    uint32_t const r = w.rid[k], a = w.run_start[r], d = k - a, c = w.run_start[r + 1] - a;
    uint32_t const phi = w.run_npool[r] & 0xffffu, plo = w.run_npool[r] >> 16, d0 = c - plo;
    uint32_t const rng = fp_range_of(rg, a);
    uint64_t       slot;
    if (d < phi || d >= d0)
    {
        uint32_t const quad = d < phi ? d / 4u : (phi + 3u) / 4u + (d - d0) / 4u, e = d < phi ? (d & 3u) : ((d - d0) & 3u);
        uint32_t const p    = w.qinv[w.run_qoff[r] + quad] - L.quad_first[rng]; // place among the range's quads
        slot                = 16ull * (L.pool_wf0[rng] + p / 4u) + 4u * (p & 3u) + e;
    }
    else
    {
        uint32_t const sp = d - phi;
        slot              = 16ull * (L.stream_wf0[rng] + w.wfoff[w.run_spos[r]]) + 2ull * (w.run_g[r] + sp / 2u) + (sp & 1u);
    }
    if (slot < 16ull * cap_wf)
        plan[slot] = w.ord[k];
    else
        w.layout->overflow = 1;
}

// ---- fillers, and per wavefront the columns per lane of its widest query and its longest window
__global__ __launch_bounds__(256) void fp_fill_kernel(Extension const * ext, int C, int no_narrow, uint32_t cap_wf, uint32_t * plan, uint32_t * wf_pan, uint32_t * wf_maxs,
                                                      FpWork w)
{
    uint64_t const o    = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t const nwf  = min(w.layout->nwf, cap_wf);
    bool const     in   = o < 16ull * nwf;
    uint32_t       v    = in ? plan[o] : kFpEmpty;
    int const      lane = threadIdx.x & 63, g0 = lane & ~15;
    // the nearest filled slot at or before this one inside the wavefront's sixteen (the first one always is)
    uint64_t const filled = __ballot(v != kFpEmpty);
    uint64_t const upto   = (filled >> g0) & ((2ull << (lane - g0)) - 1ull);
    int const      src    = upto ? g0 + 63 - __builtin_clzll(upto) : lane;
    uint32_t const from   = (uint32_t)__shfl((int)v, src);
    bool const     filler = v == kFpEmpty;
    if (filler)
        v = from | 0x80000000u;
    if (in && from == kFpEmpty)
        w.layout->overflow = 2; // (a wavefront of the plan whose first slot is empty: never, by construction)
    uint32_t pan = 0, maxs = 0;
    if (in && from != kFpEmpty)
    {
        Extension const x = ext[v & 0x7fffffffu];
        pan               = fp_cols_per_lane(x.q_len, C, no_narrow);
        maxs              = x.s_len;
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
    {
        pan  = max(pan, (uint32_t)__shfl_xor((int)pan, off));
        maxs = max(maxs, (uint32_t)__shfl_xor((int)maxs, off));
    }
    if (!in)
        return;
    if (filler)
        plan[o] = v;
    if ((o & 15) == 0)
    {
        wf_pan[o / 16]  = pan;
        wf_maxs[o / 16] = maxs;
    }
}

// what the host reads of the plan: [0] wavefronts, [1] flag, [2] runs, [3] quads, [4 + r] first wavefront of range r (r <= ranges)
__global__ void fp_report_kernel(FpLayout const * L, uint32_t nranges, uint32_t * report)
{
    uint32_t const t = threadIdx.x;
    if (t == 0)
        report[0] = L->nwf;
    if (t == 1)
        report[1] = L->overflow;
    if (t == 2)
        report[2] = L->nruns;
    if (t == 3)
        report[3] = L->nquads;
    if (t >= 4 && t - 4 <= nranges)
        report[t] = L->range_wf[t - 4];
}

inline size_t fp_align(size_t x)
{
    return (x + 255) & ~(size_t)255;
}

} // namespace

uint64_t fp_run_bound(uint64_t n, uint64_t n_qseq)
{
    return std::min<uint64_t>(n, n_qseq) + n / kFpRunCut + 2;
}

uint64_t fp_wavefront_bound(uint64_t n, uint64_t n_qseq, uint32_t nranges)
{
    // pool: a quad holds a window at least, four quads a wavefront, one partly filled per range; stream: a wavefront that is not a
    // share's last holds eight pairs or four runs' -- at least four windows either way
    return n / 4 + fp_run_bound(n, n_qseq) / kFpShare + 2ull * nranges + 4;
}

namespace
{
struct FpSizes
{
    uint64_t rm, qm, sm, tiles;
};
FpSizes fp_sizes(uint64_t n, uint64_t n_qseq, uint32_t nranges)
{
    FpSizes z;
    z.rm    = fp_run_bound(n, n_qseq);
    z.qm    = n / 4 + 2 * z.rm + 4; // (a run's long quads and its short ones: a partly filled one each)
    z.sm    = z.rm / kFpShare + nranges + 2;
    z.tiles = l2_scan_tiles(std::max<uint64_t>(n, z.qm));
    return z;
}
} // namespace

size_t fp_workspace_bytes(uint64_t n, uint64_t n_qseq, uint32_t nranges)
{
    FpSizes const z = fp_sizes(n, n_qseq, nranges);
    size_t        b = 0;
    b += 2 * fp_align(n * 4);                  // rid, ord
    b += fp_align((z.rm + 2) * 4);             // run_start
    b += 6 * fp_align(z.rm * 4);               // run_npool, run_nq, run_qoff, run_np, run_g, run_spos
    b += 4 * fp_align(z.qm * 8);               // qkey, qidx
    b += fp_align(z.qm * 4);                   // qinv
    b += 4 * fp_align(z.rm * 8);               // rkey, ridx
    b += 2 * fp_align(z.sm * 4);               // wfcnt, wfoff
    b += fp_align((z.tiles + 4) * 4);          // block_tot
    b += fp_align((l2_sort_tiles(std::max(z.qm, z.rm)) + 2) * 256 * 4); // ghist
    b += fp_align(sizeof(FpLayout));
    return b;
}

hipError_t fp_launch_plan(FpArgs const & p, hipStream_t st)
{
    uint64_t const n = p.n;
    if (n == 0 || p.nranges == 0 || p.nranges > kFpMaxRanges || n >= 0x7ffffff0ull)
        return hipErrorInvalidValue;
    FpSizes const z = fp_sizes(n, p.n_qseq, p.nranges);
    if (p.work_bytes < fp_workspace_bytes(n, p.n_qseq, p.nranges) || p.cap_wf < fp_wavefront_bound(n, p.n_qseq, p.nranges))
        return hipErrorInvalidValue;
    FpWork    w{};
    uint8_t * at   = static_cast<uint8_t *>(p.work);
    auto      take = [&](size_t bytes) { uint8_t * const r = at; at += fp_align(bytes); return r; };
    w.rid       = reinterpret_cast<uint32_t *>(take(n * 4));
    w.ord       = reinterpret_cast<uint32_t *>(take(n * 4));
    w.run_start = reinterpret_cast<uint32_t *>(take((z.rm + 2) * 4));
    w.run_npool = reinterpret_cast<uint32_t *>(take(z.rm * 4));
    w.run_nq    = reinterpret_cast<uint32_t *>(take(z.rm * 4));
    w.run_qoff  = reinterpret_cast<uint32_t *>(take(z.rm * 4));
    w.run_np    = reinterpret_cast<uint32_t *>(take(z.rm * 4));
    w.run_g     = reinterpret_cast<uint32_t *>(take(z.rm * 4));
    w.run_spos  = reinterpret_cast<uint32_t *>(take(z.rm * 4));
    for (int k = 0; k < 2; ++k)
        w.qkey[k] = reinterpret_cast<uint64_t *>(take(z.qm * 8));
    for (int k = 0; k < 2; ++k)
        w.qidx[k] = reinterpret_cast<uint64_t *>(take(z.qm * 8));
    w.qinv = reinterpret_cast<uint32_t *>(take(z.qm * 4));
    for (int k = 0; k < 2; ++k)
        w.rkey[k] = reinterpret_cast<uint64_t *>(take(z.rm * 8));
    for (int k = 0; k < 2; ++k)
        w.ridx[k] = reinterpret_cast<uint64_t *>(take(z.rm * 8));
    w.wfcnt     = reinterpret_cast<uint32_t *>(take(z.sm * 4));
    w.wfoff     = reinterpret_cast<uint32_t *>(take(z.sm * 4));
    w.block_tot = reinterpret_cast<uint32_t *>(take((z.tiles + 4) * 4));
    w.ghist     = reinterpret_cast<uint32_t *>(take((l2_sort_tiles(std::max(z.qm, z.rm)) + 2) * 256 * 4));
    w.layout    = reinterpret_cast<FpLayout *>(take(sizeof(FpLayout)));

    FpRanges rg{};
    rg.n = p.nranges;
    for (uint32_t r = 0; r <= p.nranges; ++r)
        rg.cut[r] = (uint32_t)p.cut[r];
    if (rg.cut[0] != 0 || rg.cut[p.nranges] != n)
        return hipErrorInvalidValue;
    uint32_t const rm = (uint32_t)z.rm, qm = (uint32_t)z.qm, sm = (uint32_t)z.sm, n32 = (uint32_t)n;
    dim3 const     block(256), sblock(kL2ScanBlock);
    auto const     grid = [](uint64_t items) { return dim3((unsigned)((items + 255) / 256)); };
    hipError_t     e;
    // the plan's slots start empty
    if ((e = hipMemsetAsync(p.plan, 0xff, p.cap_wf * 16 * sizeof(uint32_t), st)) != hipSuccess)
        return e;
    // runs
    {
        uint64_t const   tiles = l2_scan_tiles(n);
        RunHeadVal const hv{p.ext};
        hipLaunchKernelGGL((l2_scan_reduce_kernel<kOpSum, false, RunHeadVal>), dim3((unsigned)tiles), sblock, 0, st, hv, n, w.block_tot);
        hipLaunchKernelGGL((l2_scan_tops_kernel<kOpSum>), dim3(1), sblock, 0, st, w.block_tot, tiles);
        hipLaunchKernelGGL((l2_scan_apply_kernel<kOpSum, false, RunHeadVal, RunHeadOut>), dim3((unsigned)tiles), sblock, 0, st, hv, RunHeadOut{w.rid, w.run_start}, n, w.block_tot);
        hipLaunchKernelGGL(fp_runs_done_kernel, dim3(1), dim3(1), 0, st, w, tiles, n32, rg);
    }
    hipLaunchKernelGGL(fp_rank_kernel, grid(n), block, 0, st, p.ext, n32, w);
    hipLaunchKernelGGL(fp_run_kernel, grid(rm), block, 0, st, p.ext, rm, p.C, p.no_narrow, rg, w);
    hipLaunchKernelGGL(fp_shares_kernel, dim3(1), dim3(1), 0, st, p.nranges, w);
    // quads: where each run's begin, their sort words, the sort, every quad's place
    {
        uint64_t const   tiles = l2_scan_tiles(rm);
        RunQuadVal const qv{w.run_nq};
        hipLaunchKernelGGL((l2_scan_reduce_kernel<kOpSum, false, RunQuadVal>), dim3((unsigned)tiles), sblock, 0, st, qv, (uint64_t)rm, w.block_tot);
        hipLaunchKernelGGL((l2_scan_tops_kernel<kOpSum>), dim3(1), sblock, 0, st, w.block_tot, tiles);
        hipLaunchKernelGGL((l2_scan_apply_kernel<kOpSum, false, RunQuadVal, RunQuadOut>), dim3((unsigned)tiles), sblock, 0, st, qv, RunQuadOut{w.run_qoff}, (uint64_t)rm, w.block_tot);
        hipLaunchKernelGGL(fp_quads_done_kernel, dim3(1), dim3(1), 0, st, w, tiles, rg);
    }
    hipLaunchKernelGGL(fp_quad_init_kernel, grid(qm), block, 0, st, qm, p.nranges, w);
    hipLaunchKernelGGL(fp_quad_keys_kernel, grid(n), block, 0, st, p.ext, n32, p.C, p.no_narrow, rg, w);
    uint64_t range_bits = 0;
    while ((1ull << range_bits) <= p.nranges)
        ++range_bits;
    uint64_t const key_bits = 0xffffull | (((1ull << range_bits) - 1ull) << 16);
    {
        uint64_t * k = w.qkey[0], * kt = w.qkey[1], * v = w.qidx[0], * vt = w.qidx[1];
        if ((e = l2_launch_sort(&k, &kt, &v, &vt, qm, key_bits, 0ull, w.ghist, st)) != hipSuccess)
            return e;
        hipLaunchKernelGGL(fp_quad_inverse_kernel, grid(qm), block, 0, st, v, qm, w);
    }
    // stream: the runs in packing order, the closing rule share by share, the shares' first wavefronts
    {
        uint64_t * k = w.rkey[0], * kt = w.rkey[1], * v = w.ridx[0], * vt = w.ridx[1];
        if ((e = l2_launch_sort(&k, &kt, &v, &vt, rm, key_bits, 0ull, w.ghist, st)) != hipSuccess)
            return e;
        hipLaunchKernelGGL(fp_share_kernel, dim3(sm), dim3(64), 0, st, v, p.nranges, w);
    }
    {
        uint64_t const tiles = l2_scan_tiles(sm);
        ShareVal const sv{w.wfcnt};
        hipLaunchKernelGGL((l2_scan_reduce_kernel<kOpSum, false, ShareVal>), dim3((unsigned)tiles), sblock, 0, st, sv, (uint64_t)sm, w.block_tot);
        hipLaunchKernelGGL((l2_scan_tops_kernel<kOpSum>), dim3(1), sblock, 0, st, w.block_tot, tiles);
        hipLaunchKernelGGL((l2_scan_apply_kernel<kOpSum, false, ShareVal, ShareOut>), dim3((unsigned)tiles), sblock, 0, st, sv, ShareOut{w.wfoff}, (uint64_t)sm, w.block_tot);
        hipLaunchKernelGGL(fp_layout_kernel, dim3(1), dim3(1), 0, st, p.nranges, (uint32_t)p.cap_wf, w);
    }
    hipLaunchKernelGGL(fp_place_kernel, grid(n), block, 0, st, n32, rg, (uint32_t)p.cap_wf, p.plan, w);
    hipLaunchKernelGGL(fp_fill_kernel, grid(p.cap_wf * 16), block, 0, st, p.ext, p.C, p.no_narrow, (uint32_t)p.cap_wf, p.plan, p.wf_pan, p.wf_maxs, w);
    // what the host reads: wavefronts, overflow flag, the ranges' first wavefronts
    hipLaunchKernelGGL(fp_report_kernel, dim3(1), dim3(32), 0, st, w.layout, p.nranges, p.report);
    return hipGetLastError();
}

} // namespace lx

// lx_level2.hip -- the Level-2 driver's list work on the device (gfx950 only).
//
// What iterateMatchesFullSimd does to its span of matches before the DP sees a window
// (/root/reference/src/search_algo.hpp):
//   _widenMatch                :919-938    every seed hit becomes the DP window around its diagonal
//   std::ranges::sort          :1141       by Match's member order (src/search_datastructures.hpp:60)
//   merge right                :1144-1158  overlapping windows of one (query, subject) pair: l.subjEnd = r.subjEnd; r.subjStart = l.subjStart
//   swallow left               :1161-1168  right to left: l = r where they still overlap
//   std::ranges::unique        :1171-1173
//   slices + cut-offs          :1200-1227, :1251-1283 (the filter as an integer score test per window)
// The matches stay where the seeding stage left them: in HBM.  HBM-bound integer work, no DP here.
//
// Sort: least-significant-digit radix sort over the two words of lx_level2.h, 8 bits per pass, only the digits some key of the
// call can have set.  Per pass: (1) digit counts per tile of 4 096 keys, (2) per digit an exclusive scan over the tiles, (3) a
// stable scatter.  A wavefront ranks its 64 keys of a round among themselves with eight ballots (the lanes that hold the same
// digit) -- no LDS atomics, and a tile whose keys share one digit (the high bits of the query id) costs what any other costs.
//
// Merge: both loops of the reference are sequential over the sorted span, but what they compute is local:
//   c[i]    = same pair(i, i+1) and end0[i] >= beg0[i+1]             (merge right sees l.subjEnd and r.subjStart untouched)
//   beg[i]  = beg0[head(i)],  head(i) = first element of the run of c's that reaches i      -> max-scan of run starts
//   end[i]  = c[i] ? end0[i+1] : end0[i]
//   d[i]    = c[i] and beg[i] < end[i]                               (swallow left: the copied-down element keeps its beg:
//                                                                     begs are non-decreasing and a d never crosses a run of c's)
//   last[i] = (beg, end)[tail(i)], tail(i) = first k >= i with not d[k]                     -> min-scan from the right
//   keep[i] = i == 0 or (pair, last)[i] != (pair, last)[i-1]         (unique)                -> sum-scan, compaction
// (begs / ends non-decreasing inside a pair because the list is sorted by s0 and both are monotone in it.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>

#include "lx_level2.h"

namespace lx
{

namespace
{

constexpr uint32_t kErrBadId = 1u, kErrBadPos = 2u, kErrWide = 4u;

struct Win0
{
    uint64_t beg, end;
};

// _widenMatch, src/search_algo.hpp:919-938, from the sort key
__device__ __forceinline__ Win0 widen(L2Sets const & t, uint64_t pair, uint64_t s0)
{
    uint32_t const q = (uint32_t)(pair >> 32) & 0x7fffffffu, s = (uint32_t)pair;
    uint64_t const ql = t.q_len[q], band = t.q_band[q];
    Win0           w;
    w.end = min(s0 + ql + band, t.s_len[s]);
    w.beg = band < s0 ? s0 - band : 0;
    return w;
}

__global__ __launch_bounds__(256) void l2_keys_kernel(Match const * m, L2Params p, uint64_t * pair, uint64_t * s0, uint64_t * flag)
{
    uint64_t const i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n)
        return;
    Match const x = m[i];
    uint32_t     err = 0;
    if (x.qryId >= p.sets.n_qseq || x.subjId >= p.sets.n_sseq)
    {
        err = kErrBadId;
        pair[i] = 0;
        s0[i]   = 0;
    }
    else
    {
        uint64_t const d = x.subjStart < x.qryStart ? 0 : x.subjStart - x.qryStart;
        if (d >= p.sets.s_len[x.subjId]) // a seed beyond its subject: the window would be empty or inverted
            err = kErrBadPos;
        pair[i] = ((p.bisulfite ? (x.subjId & 1) : 0ull) << 63) | (x.qryId << 32) | x.subjId;
        s0[i]   = d;
    }
    if (err)
        atomicOr(reinterpret_cast<unsigned long long *>(flag), (unsigned long long)err);
}

// ---- radix sort ------------------------------------------------------------------------------------------------------

// the lanes of the wavefront that hold the same 8-bit digit as this one (only lanes with `valid` count)
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kL2DigitBits; ++b)
    {
        bool const     bit = (d >> b) & 1u;
        uint64_t const bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t * wave_sums, uint32_t & total)
{
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t  incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        uint32_t const up = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off)
            incl += up;
    }
    if (lane == 63)
        wave_sums[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
    {
        uint32_t const ws = wave_sums[w];
        base += (w < wave) ? ws : 0u;
        tot += ws;
    }
    total = tot;
    __syncthreads(); // (wave_sums may be reused by the caller's next round)
    return base + incl - v;
}

// (1) digit counts of a tile: ghist[digit * tiles + tile]
__global__ __launch_bounds__(kL2SortBlock) void l2_sort_count_kernel(uint64_t const * word, uint64_t n, int shift, uint32_t * ghist, uint32_t tiles)
{
    __shared__ uint32_t cnt[4][256];
    int const           lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        cnt[k][threadIdx.x] = 0;
    __syncthreads();
    uint64_t const         base = (uint64_t)blockIdx.x * kL2SortTile + (uint64_t)w * (kL2SortTile / 4);
    volatile uint32_t * const mine = cnt[w];
#pragma unroll 4
    for (int r = 0; r < kL2SortItems; ++r)
    {
        uint64_t const i     = base + (uint64_t)r * 64 + lane;
        bool const     valid = i < n;
        uint32_t const d     = valid ? (uint32_t)(word[i] >> shift) & 255u : 0u;
        uint64_t const peers = match_digit(d, valid);
        // the highest lane of a group adds the group's size (one LDS access per distinct digit and round)
        if (valid && (peers >> lane) == 1ull)
            mine[d] = mine[d] + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    ghist[(uint64_t)threadIdx.x * tiles + blockIdx.x] = cnt[0][threadIdx.x] + cnt[1][threadIdx.x] + cnt[2][threadIdx.x] + cnt[3][threadIdx.x];
}

// (2) one workgroup per digit: exclusive scan of its counts over the tiles, in place; gtot[digit] = its total
__global__ __launch_bounds__(256) void l2_sort_scan_kernel(uint32_t * ghist, uint32_t tiles, uint32_t * gtot)
{
    __shared__ uint32_t ws[4];
    uint32_t * const    row   = ghist + (uint64_t)blockIdx.x * tiles;
    uint32_t            carry = 0;
    for (uint32_t b0 = 0; b0 < tiles; b0 += 256)
    {
        uint32_t const b = b0 + threadIdx.x;
        uint32_t const v = b < tiles ? row[b] : 0u;
        uint32_t       total;
        uint32_t const ex = block_exclusive_sum(v, ws, total);
        if (b < tiles)
            row[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0)
        gtot[blockIdx.x] = carry;
}

// (3) stable scatter of a tile by the digit; `word` is the word the digit is taken from, `other` travels with it
__global__ __launch_bounds__(kL2SortBlock) void l2_sort_scatter_kernel(uint64_t const * word, uint64_t const * other, uint64_t * word_out,
                                                                        uint64_t * other_out, uint64_t n, int shift, uint32_t const * ghist,
                                                                        uint32_t const * gtot, uint32_t tiles)
{
    __shared__ uint32_t cnt[4][256];
    __shared__ uint32_t dbase[256];
    __shared__ uint32_t ws[4];
    int const           lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        cnt[k][threadIdx.x] = 0;
    {
        // where this tile's keys of digit d go: the keys of all smaller digits + those of digit d in the tiles before this one
        uint32_t       total;
        uint32_t const ex = block_exclusive_sum(gtot[threadIdx.x], ws, total);
        dbase[threadIdx.x] = ex + ghist[(uint64_t)threadIdx.x * tiles + blockIdx.x];
    }
    __syncthreads();
    uint64_t const         base = (uint64_t)blockIdx.x * kL2SortTile + (uint64_t)w * (kL2SortTile / 4);
    volatile uint32_t * const mine = cnt[w];
    uint64_t kw[kL2SortItems], ko[kL2SortItems];
    uint32_t off[kL2SortItems];
#pragma unroll
    for (int r = 0; r < kL2SortItems; ++r)
    {
        uint64_t const i     = base + (uint64_t)r * 64 + lane;
        bool const     valid = i < n;
        kw[r]                = valid ? word[i] : 0ull;
        ko[r]                = valid ? other[i] : 0ull;
        uint32_t const d     = (uint32_t)(kw[r] >> shift) & 255u;
        uint64_t const peers = match_digit(d, valid);
        uint32_t const rank  = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        uint32_t const prev  = valid ? mine[d] : 0u; // (the same value for every lane of the group: read before the leader writes)
        off[r]               = prev + rank;
        if (valid && (peers >> lane) == 1ull)
            mine[d] = prev + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    {
        // per digit: where each wavefront's share begins inside the tile's share
        uint32_t const c0 = cnt[0][threadIdx.x], c1 = cnt[1][threadIdx.x], c2 = cnt[2][threadIdx.x];
        uint32_t const b  = dbase[threadIdx.x];
        cnt[0][threadIdx.x] = b;
        cnt[1][threadIdx.x] = b + c0;
        cnt[2][threadIdx.x] = b + c0 + c1;
        cnt[3][threadIdx.x] = b + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kL2SortItems; ++r)
    {
        uint64_t const i = base + (uint64_t)r * 64 + lane;
        if (i < n)
        {
            uint32_t const d   = (uint32_t)(kw[r] >> shift) & 255u;
            uint64_t const pos = (uint64_t)cnt[w][d] + off[r];
            word_out[pos]      = kw[r];
            other_out[pos]     = ko[r];
        }
    }
}

#include "lx_scan.h"

// ---- the merge's three scans ------------------------------------------------------------------------------------------------

struct Sorted
{
    uint64_t const * pair;
    uint64_t const * s0;
    uint64_t         n;
    L2Sets           sets;
    __device__ __forceinline__ Win0 win(uint64_t i) const { return widen(sets, pair[i], s0[i]); }
    // merge right's test on (i, i + 1), src/search_algo.hpp:1149-1151
    // (all loads issued whatever the earlier tests say: a chain of dependent round trips costs more than the bytes)
    __device__ __forceinline__ bool chained(uint64_t i) const
    {
        uint64_t const r  = min(i + 1, n - 1);
        uint64_t const pa = pair[i], pb = pair[r];
        Win0 const     a = widen(sets, pa, s0[i]), b = widen(sets, pb, s0[r]);
        return (i + 1 < n) & (pa == pb) & (a.end >= b.beg);
    }
};

// head(i) + 1 where a run of chained elements starts at i, else 0; inclusive max-scan -> head(i) + 1
struct HeadVal
{
    Sorted s;
    __device__ uint32_t operator()(uint64_t i) const { return (i == 0 || !s.chained(i - 1)) ? (uint32_t)i + 1u : 0u; }
};
// after merge right element i holds (beg0[head(i)], chained(i) ? end0[i + 1] : end0[i]): written once (the later passes read it
// instead of walking the tables again); bit 63 of the end = "chained to its right neighbour"
constexpr uint64_t kChainedBit = 1ull << 63;
struct HeadOut
{
    Sorted     s;
    uint64_t * mrg_beg;
    uint64_t * mrg_end;
    __device__ void operator()(uint64_t i, uint32_t incl, uint32_t) const
    {
        bool const ch = s.chained(i);
        mrg_beg[i]    = s.win(incl - 1u).beg;
        mrg_end[i]    = (ch ? s.win(i + 1).end : s.win(i).end) | (ch ? kChainedBit : 0ull);
    }
};

// swallow left's test on (i, i + 1), src/search_algo.hpp:1164-1166: the right element's current subjStart is beg[i + 1] whatever
// was copied onto it (see the header), and it equals beg[i] when they are chained.
// tail(i) = first k >= i that is not swallowed by its right neighbour: min-scan from the right over (swallowed ? inf : k)
struct TailVal
{
    uint64_t const * mrg_beg;
    uint64_t const * mrg_end;
    __device__ uint32_t operator()(uint64_t i) const
    {
        uint64_t const e = mrg_end[i], b = mrg_beg[i];
        return (((e & kChainedBit) != 0) & (b < (e & ~kChainedBit))) ? 0xffffffffu : (uint32_t)i;
    }
};
// what the span holds at i after swallow left: the record of tail(i)
struct TailOut
{
    uint64_t const * mrg_beg;
    uint64_t const * mrg_end;
    uint64_t *       fin_beg;
    uint64_t *       fin_end;
    __device__ void operator()(uint64_t i, uint32_t incl, uint32_t) const
    {
        fin_beg[i] = mrg_beg[incl];
        fin_end[i] = mrg_end[incl] & ~kChainedBit;
    }
};

struct KeepVal
{
    uint64_t const * pair;
    uint64_t const * fin_beg;
    uint64_t const * fin_end;
    __device__ uint32_t operator()(uint64_t i) const
    {
        uint64_t const l = i ? i - 1 : 0;
        return ((i == 0) | (pair[i] != pair[l]) | (fin_beg[i] != fin_beg[l]) | (fin_end[i] != fin_end[l])) ? 1u : 0u;
    }
};
struct KeepOut
{
    KeepVal  k;
    L2Params p;
    __device__ void operator()(uint64_t i, uint32_t incl, uint32_t ex) const
    {
        if (incl == ex)
            return; // a duplicate: removed by unique
        uint64_t const pr = k.pair[i];
        uint32_t const q = (uint32_t)(pr >> 32) & 0x7fffffffu, s = (uint32_t)pr;
        Win0 const     w{k.fin_beg[i], k.fin_end[i]};
        L2Window       o;
        o.q   = q;
        o.s   = s;
        o.beg = w.beg;
        o.end = w.end;
        p.win_out[ex] = o;
        // the slices of :1200-1227: the whole (frame) query against the window
        Extension e;
        e.q_off = p.sets.q_off[q];
        e.q_len = p.sets.q_len[q];
        e.s_off = p.sets.s_off[s] + w.beg;
        uint64_t const len = w.end > w.beg ? w.end - w.beg : 0;
        e.s_len = (uint32_t)min(len, (uint64_t)0xffffffffu);
        if (len > 0xffffffffull)
            atomicOr(reinterpret_cast<unsigned long long *>(p.count_out + 2), (unsigned long long)kErrWide);
        p.ext_out[ex] = e;
        p.min_out[ex] = p.cut_by_len[p.sets.q_evlen[q]];
        // where the windows of odd subject frames begin (the list is sorted by the bisulfite flag first)
        if (p.bisulfite && (pr >> 63) && (i == 0 || !(k.pair[i - 1] >> 63)))
            p.count_out[1] = ex;
    }
};

__global__ void l2_finish_kernel(uint32_t const * block_tot, uint64_t tiles, L2Params p)
{
    uint64_t const n = block_tot[tiles];
    p.count_out[0]   = n;
    if (!p.bisulfite || p.count_out[1] == ~0ull)
        p.count_out[1] = n; // no window of an odd subject frame
}

// ---- the solo plan of the multi-query sweep (lx_sweep_mq.hip: a byte profile per window, 16 windows of any queries per
// wavefront): all windows by (columns per lane their query sweeps, window length), longest first, 16 to a wavefront.  What
// lx_host.cpp's extend_pipeline does on the host threads for lists it is handed in host memory.

// columns per lane a query sweeps over all its panels of 8 x C columns: whole panels, the last one with the narrowest strips that
// cover what is left (narrow_code_for) -- lx_host.cpp's mq_panels
__device__ __forceinline__ uint32_t cols_per_lane(uint32_t lq, int C, int no_narrow)
{
    uint32_t const panel = 8u * (uint32_t)C;
    uint32_t const P     = max(1u, (lq + panel - 1) / panel);
    int const      rem   = (int)(max(lq, 1u) - (P - 1) * panel);
    int const      code  = no_narrow ? 0 : narrow_code_for(C, 8, rem);
    return min(0xfffu, (P - 1) * (uint32_t)C + (uint32_t)narrow_strip_cols(C, code));
}

// per part (0: the windows before cnt[1], 1: those behind) the instructions a sweep of the part costs at each strip geometry --
// 4 x ((P - 1) (3.75 C + 12) + 3.75 (last panel's columns per lane) + 12) per window -- and the part's cells
__global__ __launch_bounds__(256) void l2_plan_cost_kernel(Extension const * ext, uint64_t const * cnt, int no_narrow, unsigned long long * out)
{
    uint64_t const nw = cnt[0], n_even = cnt[1];
    uint64_t       v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t       longest[2] = {0, 0}; // the part's longest window: out[8], out[9] (the plan's sort skips the key digits no window sets)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * blockDim.x)
    {
        Extension const x    = ext[i];
        int const       part = i >= n_even ? 4 : 0;
        int const       cand[3] = {19, 13, 11};
#pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            uint32_t const panel = 8u * (uint32_t)cand[k], P = max(1u, (x.q_len + panel - 1) / panel);
            int const      rem   = (int)(max(x.q_len, 1u) - (P - 1) * panel);
            int const      code  = no_narrow ? 0 : narrow_code_for(cand[k], 8, rem);
            v[part + k] += (uint64_t)(P - 1) * (15u * (uint32_t)cand[k] + 48u) + 15u * (uint32_t)narrow_strip_cols(cand[k], code) + 48u;
        }
        v[part + 3] += (uint64_t)x.q_len * x.s_len;
        longest[part / 4] = max(longest[part / 4], x.s_len);
    }
    // one atomic per workgroup and sum (same-address atomics from every wavefront of the grid were most of this kernel's time)
    __shared__ unsigned long long part[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
    {
        uint64_t x = v[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            x += (uint64_t)__shfl_xor((long long)x, off);
        if ((threadIdx.x & 63) == 0)
            part[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 8)
    {
        unsigned long long const x = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (x)
            atomicAdd(out + threadIdx.x, x);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
        uint32_t m = longest[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            m = max(m, (uint32_t)__shfl_xor((int)m, off));
        if ((threadIdx.x & 63) == 0 && m)
            atomicMax(out + 8 + k, (unsigned long long)m);
    }
}

__global__ __launch_bounds__(256) void l2_plan_keys_kernel(Extension const * ext, uint64_t n, int C, int no_narrow, uint64_t * key, uint64_t * idx)
{
    uint64_t const i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    Extension const x = ext[i];
    // ascending keys = most columns per lane first, longest window first inside a width
    key[i] = ((uint64_t)(0xfffu - cols_per_lane(x.q_len, C, no_narrow)) << 16) | (0xffffu - min(x.s_len, 0xffffu));
    idx[i] = i;
}

// slot o of the plan = the o-th window in that order (the last wavefront repeats the last window as filler: bit 31); per
// wavefront the columns per lane of its widest query and its longest window
__global__ __launch_bounds__(256) void l2_plan_slots_kernel(Extension const * ext, uint64_t const * idx_sorted, uint64_t n, int C, int no_narrow, uint32_t index_base,
                                                             uint32_t * plan, uint32_t * wf_pan, uint32_t * wf_maxs)
{
    uint64_t const o    = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t const nwf  = (n + 15) / 16;
    bool const     in   = o < nwf * 16;
    uint64_t const i    = idx_sorted[min(o, n - 1)];
    Extension const x   = ext[i];
    uint32_t        pan = cols_per_lane(x.q_len, C, no_narrow), maxs = x.s_len;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
    {
        pan  = max(pan, (uint32_t)__shfl_xor((int)pan, off));
        maxs = max(maxs, (uint32_t)__shfl_xor((int)maxs, off));
    }
    if (!in)
        return;
    plan[o] = ((uint32_t)i + index_base) | (o < n ? 0u : 0x80000000u); // (index_base: `ext` is a range of a longer list)
    if ((o & 15) == 0)
    {
        wf_pan[o / 16]  = pan;
        wf_maxs[o / 16] = maxs;
    }
}

} // namespace

hipError_t l2_launch_plan_cost(Extension const * ext, uint64_t const * cnt, uint64_t n_max, int no_narrow, unsigned long long * out, hipStream_t stream)
{
    hipError_t const e = hipMemsetAsync(out, 0, 10 * sizeof(unsigned long long), stream);
    if (e != hipSuccess || n_max == 0)
        return e;
    unsigned const blocks = (unsigned)std::min<uint64_t>((n_max + 255) / 256, 256); // (eight same-address atomics per wavefront: few wavefronts)
    hipLaunchKernelGGL(l2_plan_cost_kernel, dim3(blocks), dim3(256), 0, stream, ext, cnt, no_narrow, out);
    return hipGetLastError();
}

// the plan of ext[0 .. n): key / idx are sort words (two buffers each, as l2_launch_sort takes them); plan: [ceil(n / 16) * 16]
hipError_t l2_launch_plan(Extension const * ext, uint64_t n, int C, int no_narrow, uint64_t ** key, uint64_t ** key_tmp, uint64_t ** idx, uint64_t ** idx_tmp,
                          uint32_t * ghist, uint32_t * plan, uint32_t * wf_pan, uint32_t * wf_maxs, hipStream_t stream, uint32_t index_base, uint64_t key_bits)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(l2_plan_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ext, n, C, no_narrow, *key, *idx);
    // (key_bits: the digits in which two keys of the list can differ -- l2_plan_key_bits; a digit all keys share is not sorted by)
    hipError_t const e = l2_launch_sort(key, key_tmp, idx, idx_tmp, n, key_bits & 0x0fffffffull, 0ull, ghist, stream);
    if (e != hipSuccess)
        return e;
    uint64_t const slots = (n + 15) / 16 * 16;
    hipLaunchKernelGGL(l2_plan_slots_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, ext, *idx, n, C, no_narrow, index_base, plan, wf_pan, wf_maxs);
    return hipGetLastError();
}

// the digits of the plan's keys -- (0xfff - columns per lane) << 16 | (0xffff - min(window length, 0xffff)) -- that differ somewhere
// in a list of queries up to max_qlen residues and windows up to max_wlen: the complements keep the high digit of either field
// at all ones while the value stays below 256
uint64_t l2_plan_key_bits(uint64_t max_qlen, uint64_t max_wlen)
{
    return 0xffull | (max_wlen >= 256 ? 0xff00ull : 0ull) | 0xff0000ull | (max_qlen / 8 + 32 >= 256 ? 0x0f000000ull : 0ull);
}

uint64_t l2_sort_tiles(uint64_t n)
{
    return (n + kL2SortTile - 1) / kL2SortTile;
}
uint64_t l2_scan_tiles(uint64_t n)
{
    return (n + kL2ScanTile - 1) / kL2ScanTile;
}

hipError_t l2_launch_keys(void const * d_matches, L2Params const & p, uint64_t * pair, uint64_t * s0, uint64_t * flag, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(l2_keys_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, stream, static_cast<Match const *>(d_matches), p, pair, s0, flag);
    return hipGetLastError();
}

hipError_t l2_launch_sort(uint64_t ** pair, uint64_t ** pair_tmp, uint64_t ** s0, uint64_t ** s0_tmp, uint64_t n, uint64_t pair_bits, uint64_t s0_bits,
                          uint32_t * ghist, hipStream_t stream)
{
    if (n < 2)
        return hipSuccess;
    uint32_t const tiles = (uint32_t)l2_sort_tiles(n);
    uint32_t * const gtot = ghist + (uint64_t)tiles * 256;
    for (int word = 0; word < 2; ++word) // least significant first: s0, then the pair
    {
        uint64_t const bits = word == 0 ? s0_bits : pair_bits;
        for (int shift = 0; shift < 64; shift += kL2DigitBits)
        {
            if (!((bits >> shift) & 255ull))
                continue; // no key of the call has a bit of this digit set
            uint64_t ** w = word == 0 ? s0 : pair, ** wt = word == 0 ? s0_tmp : pair_tmp, ** o = word == 0 ? pair : s0, ** ot = word == 0 ? pair_tmp : s0_tmp;
            hipLaunchKernelGGL(l2_sort_count_kernel, dim3(tiles), dim3(kL2SortBlock), 0, stream, *w, n, shift, ghist, tiles);
            hipLaunchKernelGGL(l2_sort_scan_kernel, dim3(256), dim3(256), 0, stream, ghist, tiles, gtot);
            hipLaunchKernelGGL(l2_sort_scatter_kernel, dim3(tiles), dim3(kL2SortBlock), 0, stream, *w, *o, *wt, *ot, n, shift, ghist, gtot, tiles);
            std::swap(*w, *wt);
            std::swap(*o, *ot);
        }
    }
    return hipGetLastError();
}

hipError_t l2_launch_merge(uint64_t const * pair, uint64_t const * s0, L2Params const & p, uint64_t * mrg_beg, uint64_t * mrg_end, uint64_t * fin_beg,
                           uint64_t * fin_end, uint32_t * block_tot, hipStream_t stream)
{
    uint64_t const n = p.n;
    hipError_t     e = hipMemsetAsync(p.count_out + 1, 0xff, sizeof(uint64_t), stream);
    if (e != hipSuccess)
        return e;
    uint64_t const tiles = l2_scan_tiles(n);
    if (n == 0)
    {
        e = hipMemsetAsync(block_tot, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(l2_finish_kernel, dim3(1), dim3(1), 0, stream, block_tot, (uint64_t)0, p);
        return hipGetLastError();
    }
    Sorted const  s{pair, s0, n, p.sets};
    dim3 const    grid((unsigned)tiles), block(kL2ScanBlock);
    HeadVal const hv{s};
    hipLaunchKernelGGL((l2_scan_reduce_kernel<kOpMax, false, HeadVal>), grid, block, 0, stream, hv, n, block_tot);
    hipLaunchKernelGGL((l2_scan_tops_kernel<kOpMax>), dim3(1), block, 0, stream, block_tot, tiles);
    hipLaunchKernelGGL((l2_scan_apply_kernel<kOpMax, false, HeadVal, HeadOut>), grid, block, 0, stream, hv, HeadOut{s, mrg_beg, mrg_end}, n, block_tot);
    TailVal const tv{mrg_beg, mrg_end};
    hipLaunchKernelGGL((l2_scan_reduce_kernel<kOpMin, true, TailVal>), grid, block, 0, stream, tv, n, block_tot);
    hipLaunchKernelGGL((l2_scan_tops_kernel<kOpMin>), dim3(1), block, 0, stream, block_tot, tiles);
    hipLaunchKernelGGL((l2_scan_apply_kernel<kOpMin, true, TailVal, TailOut>), grid, block, 0, stream, tv, TailOut{mrg_beg, mrg_end, fin_beg, fin_end}, n, block_tot);
    KeepVal const kv{pair, fin_beg, fin_end};
    hipLaunchKernelGGL((l2_scan_reduce_kernel<kOpSum, false, KeepVal>), grid, block, 0, stream, kv, n, block_tot);
    hipLaunchKernelGGL((l2_scan_tops_kernel<kOpSum>), dim3(1), block, 0, stream, block_tot, tiles);
    hipLaunchKernelGGL((l2_scan_apply_kernel<kOpSum, false, KeepVal, KeepOut>), grid, block, 0, stream, kv, KeepOut{kv, p}, n, block_tot);
    hipLaunchKernelGGL(l2_finish_kernel, dim3(1), dim3(1), 0, stream, block_tot, tiles, p);
    return hipGetLastError();
}

} // namespace lx

// lx_records.hip -- the tail of iterateMatchesFullSimd on the device (gfx950 only): from the survivors' alignments, where the
// backtrace kernels left them in HBM, to finished result records in the reference's order.
//
// What the reference does after its second DP pass (/root/reference/src/search_algo.hpp):
//   the filter's statistics              :1260, :1274      failed bit score / failed e-value, from the scores of pass 1
//   std::ranges::stable_sort by _n_qId   :1299             (behind the stable sort by slice lengths of :1229-1235: the survivors end up
//                                                           ordered by (true query id, query slice length, subject slice length, list position))
//   _expandAlign                         :998-1042, :1306  positions in the infixes become positions in the sequences
//   computeAlignmentStats + identity     :1308-1315        the counts are the backtrace's; the identity cut-off drops records
//   computeBitScore, computeEValue       :1318-1322, src/search_misc.hpp:56-80
// Rounds 1-4 brought the survivors' 48-byte alignments down and did all of this on the host threads (host/lx_iterate_common.hpp:
// finishSurvivors -- a scatter, a pass over all windows, a sort per query, two passes over the survivors with a cache miss per record:
// 4 ms of a 14-ms call on a million reads).  Here: one key per survivor, the Level-2 radix sort (lx_level2.hip), one scan, and
// every record written once, in place, 128 bytes per lane; the host receives finished lx_blast_match rows in one copy.
//
// Bit-identical to the host form by construction, not by tolerance:
//   * identity is the host's expression in the host's types (float -> double product -> float);
//   * bit score = (lambda * score - log K) / log 2 with the host's two logarithms passed in and IEEE double operations that the
//     compiler may not contract (#pragma clang fp contract(off): product, difference and quotient round one by one, as on the host);
//   * e-value = (K * ql' * dl') * exp(-lambda * score): the first factor per distinct query length and exp(-lambda * s) for every
//     score s until it underflows to zero come from the HOST's libm as two tables (lx_level2_host.cpp), the device multiplies;
//   * the bit-score test of the statistics is an integer cut-off found by bisection over the host's formula, like the filter's own.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "lx_level2.h"

namespace lx
{

namespace
{

constexpr int kRecBlock = 256;

__device__ __forceinline__ int32_t frame_of(int mode, uint64_t id, bool subject) // host/lx_translate.cpp frameOf (_setFrames, :768-814)
{
    switch (mode)
    {
        case 2: // LX_FRAMES_TRANSLATED
        {
            int32_t const f = (int32_t)(id % 3) + 1;
            return (id % 6 > 2) ? -f : f;
        }
        case 3: // LX_FRAMES_BISULFITE
        {
            int32_t const f = (int32_t)(id % 2) + 1;
            return (!subject && id % 4 > 1) ? -f : f;
        }
        case 1: // LX_FRAMES_REVCOMP
            return (id % 2) ? -1 : 1;
        default: return 0;
    }
}

__device__ __forceinline__ float identity_of(Hsp const & a) // finishSurvivors' identityOf
{
    return a.n_ops ? (float)(100.0 * (double)(float)a.num_matches / (double)(float)a.n_ops) : 0.0f;
}

// sum over the wavefront, result in every lane
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += (uint32_t)__shfl_xor((int)v, off);
    return v;
}

// (0) once per call: every window's place in the records' order.  The window list is sorted by query already (Match's order), so the
// reference's order -- true query id, then the lengths of the two slices, then the list position (search_algo.hpp:1229-1235 and :1299,
// two stable sorts) -- moves a window only among the windows of its own query: a handful.  Counting those that come before it gives
// its place; the survivors of a range are then sorted by that one word (three radix passes instead of eight or nine per range).
__global__ __launch_bounds__(kRecBlock) void rec_rank_kernel(L2Window const * win, uint64_t n, uint32_t q_frames, uint32_t const * q_len, uint32_t * rank, uint32_t * too_long)
{
    uint64_t const w = (uint64_t)blockIdx.x * kRecBlock + threadIdx.x;
    if (w >= n)
        return;
    L2Window const W   = win[w];
    uint32_t const qid = W.q / q_frames;
    uint64_t const ql  = q_len[W.q], sl = W.end > W.beg ? W.end - W.beg : 0ull;
    uint64_t       lo  = w, hi = w + 1;
    while (lo > 0 && w - lo < kRecRankGroup && win[lo - 1].q / q_frames == qid)
        --lo;
    while (hi < n && hi - w < kRecRankGroup && win[hi].q / q_frames == qid)
        ++hi;
    if ((w - lo >= kRecRankGroup && lo > 0 && win[lo - 1].q / q_frames == qid) || (hi - w >= kRecRankGroup && hi < n && win[hi].q / q_frames == qid))
        atomicOr(too_long, 1u); // (a query with more windows than one thread counts: the caller sorts by the full words)
    uint32_t before = 0;
    for (uint64_t v = lo; v < hi; ++v)
    {
        L2Window const V  = win[v];
        uint64_t const vq = q_len[V.q], vs = V.end > V.beg ? V.end - V.beg : 0ull;
        before += (vq < ql || (vq == ql && (vs < sl || (vs == sl && v < w)))) ? 1u : 0u;
    }
    rank[w] = (uint32_t)lo + before;
}

// (1) every stored survivor: where it stands (list_at: window -> entry) and its sort words.  Entries that are padding of a chunk's
// survivor list sort behind every real one (query id = the number of queries).
__global__ __launch_bounds__(kRecBlock) void rec_keys_kernel(RecParams p, uint64_t * pair, uint64_t * s0)
{
    // (a few hundred workgroups over all entries, one atomic per workgroup: one per wavefront was ten thousand atomics on ONE counter,
    // 0.08 ms of a kernel that streams 15 MB)
    __shared__ uint32_t part[kRecBlock / 64];
    uint32_t            mine = 0;
    for (uint64_t e = (uint64_t)blockIdx.x * kRecBlock + threadIdx.x; e < p.n_entries; e += (uint64_t)gridDim.x * kRecBlock)
    {
        bool const     filled = !p.count_ptr || e < *p.count_ptr;
        uint32_t const ws     = filled ? p.src[e] : 0xffffffffu;
        uint32_t const w      = ws - p.src_base; // (position among THIS call's windows: a chunk's range of the list)
        uint64_t       kp = ((uint64_t)p.n_qid_end << 32), ks = e;
        bool           valid_e = false;
        if (ws != 0xffffffffu)
        {
            if (ws < p.src_base || w >= p.n_win)
                atomicOr(reinterpret_cast<unsigned long long *>(p.counters + kRecErr), 2ull);
            else if (p.hsp[e].score < 0)
                atomicOr(reinterpret_cast<unsigned long long *>(p.counters + kRecErr), 1ull); // an extension that could not be traced
            else
            {
                ++mine;
                p.list_at[w] = (uint32_t)e;
                if (p.rank)
                {
                    kp = w; // (travels with the key: the kernels behind the sort read the window from it)
                    ks = p.rank[w];
                }
                else
                {
                    L2Window const W = p.win[w];
                    kp               = ((uint64_t)(W.q / p.q_frames) << 32) | p.q_len[W.q];
                    ks               = ((W.end > W.beg ? W.end - W.beg : 0ull) << 32) | w;
                }
                valid_e = true;
            }
        }
        if (p.rank && !valid_e)
        {
            kp = 0;
            ks = p.rank_pad;
        }
        pair[e] = kp;
        s0[e]   = ks;
    }
    uint32_t const n = wave_sum(mine);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t tot = 0;
        for (int k = 0; k < kRecBlock / 64; ++k)
            tot += part[k];
        if (tot)
            atomicAdd(reinterpret_cast<unsigned long long *>(p.counters + kRecSurvivors), (unsigned long long)tot);
    }
}

// (2) the filter's statistics from the scores of pass 1 (:1260, :1274): a window that is no survivor failed the bit score or, else, the
// e-value -- unless it is an empty window (score 0 against a cut-off of 0: nothing to trace)
__global__ __launch_bounds__(kRecBlock) void rec_stats_kernel(RecParams p)
{
    __shared__ uint32_t part[2][kRecBlock / 64];
    uint32_t            fb = 0, fe = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * kRecBlock + threadIdx.x; w < p.n_win; w += (uint64_t)gridDim.x * kRecBlock)
        if (p.list_at[w] == 0xffffffffu)
        {
            int32_t const sc = p.score[w];
            if (sc < p.min_score[w])
            {
                if (sc < p.bit_cut)
                    ++fb;
                else
                    ++fe;
            }
        }
    fb = wave_sum(fb);
    fe = wave_sum(fe);
    if ((threadIdx.x & 63) == 0)
    {
        part[0][threadIdx.x >> 6] = fb;
        part[1][threadIdx.x >> 6] = fe;
    }
    __syncthreads();
    if (threadIdx.x == 0) // (one pair of atomics per workgroup, as in rec_keys_kernel)
    {
        uint32_t tb = 0, te = 0;
        for (int k = 0; k < kRecBlock / 64; ++k)
        {
            tb += part[0][k];
            te += part[1][k];
        }
        if (tb)
            atomicAdd(reinterpret_cast<unsigned long long *>(p.counters + kRecFailedBit), (unsigned long long)tb);
        if (te)
            atomicAdd(reinterpret_cast<unsigned long long *>(p.counters + kRecFailedEv), (unsigned long long)te);
    }
}

// what position x of the sorted survivors contributes: a record (identity cut-off, :1310-1315) and its alignment columns
struct Contribution
{
    uint32_t keep, ops, entry;
};
__device__ __forceinline__ Contribution contribution(RecParams const & p, uint64_t const * s0, uint64_t x, uint64_t n)
{
    Contribution c{0, 0, 0};
    if (x < n)
    {
        c.entry          = p.list_at[(uint32_t)s0[x]];
        Hsp const & a    = p.hsp[c.entry];
        c.keep           = !(identity_of(a) < (float)p.id_cutoff) ? 1u : 0u;
        c.ops            = (c.keep && p.want_ops) ? (uint32_t)a.n_ops : 0u;
    }
    return c;
}

// (3) per tile of kRecBlock sorted survivors: records and columns
__global__ __launch_bounds__(kRecBlock) void rec_tile_kernel(RecParams p, uint64_t const * s0, uint64_t const * n_ptr, uint32_t * tile_keep, uint64_t * tile_ops)
{
    uint64_t const     n = *n_ptr;
    uint64_t const     x = (uint64_t)blockIdx.x * kRecBlock + threadIdx.x;
    Contribution const c = contribution(p, s0, x, n);
    __shared__ uint32_t wk[kRecBlock / 64];
    __shared__ uint64_t wo[kRecBlock / 64];
    uint32_t const      k = wave_sum(c.keep);
    uint64_t            o = c.ops;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        o += (uint64_t)__shfl_xor((long long)o, off);
    if ((threadIdx.x & 63) == 0)
    {
        wk[threadIdx.x >> 6] = k;
        wo[threadIdx.x >> 6] = o;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t tk = 0;
        uint64_t to = 0;
        for (int w = 0; w < kRecBlock / 64; ++w)
        {
            tk += wk[w];
            to += wo[w];
        }
        tile_keep[blockIdx.x] = tk;
        tile_ops[blockIdx.x]  = to;
    }
}

// (4) one workgroup: what precedes every tile; the totals
__global__ __launch_bounds__(kRecBlock) void rec_tops_kernel(RecParams p, uint64_t const * n_ptr, uint32_t * tile_keep, uint64_t * tile_ops)
{
    uint64_t const      tiles = (*n_ptr + kRecBlock - 1) / kRecBlock;
    __shared__ uint64_t sk[kRecBlock], so[kRecBlock];
    uint64_t            carry_k = 0, carry_o = 0;
    for (uint64_t b0 = 0; b0 < tiles; b0 += kRecBlock)
    {
        uint64_t const b = b0 + threadIdx.x;
        uint64_t const k = b < tiles ? tile_keep[b] : 0, o = b < tiles ? tile_ops[b] : 0;
        sk[threadIdx.x] = k;
        so[threadIdx.x] = o;
        __syncthreads();
        // (Hillis-Steele over the workgroup: 256 tiles per round, a handful of rounds per call)
        for (int off = 1; off < kRecBlock; off <<= 1)
        {
            uint64_t const ak = threadIdx.x >= (unsigned)off ? sk[threadIdx.x - off] : 0, ao = threadIdx.x >= (unsigned)off ? so[threadIdx.x - off] : 0;
            __syncthreads();
            sk[threadIdx.x] += ak;
            so[threadIdx.x] += ao;
            __syncthreads();
        }
        if (b < tiles)
        {
            tile_keep[b] = (uint32_t)(carry_k + sk[threadIdx.x] - k);
            tile_ops[b]  = carry_o + so[threadIdx.x] - o;
        }
        carry_k += sk[kRecBlock - 1];
        carry_o += so[kRecBlock - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        p.counters[kRecKept] = carry_k;
        p.counters[kRecOps]  = carry_o;
    }
}

// (5) the records (:1302-1325), each written once where it stays
__global__ __launch_bounds__(kRecBlock) void rec_write_kernel(RecParams p, uint64_t const * s0, uint64_t const * n_ptr, uint32_t const * tile_keep,
                                                              uint64_t const * tile_ops)
{
    uint64_t const     n = *n_ptr;
    uint64_t const     x = (uint64_t)blockIdx.x * kRecBlock + threadIdx.x;
    Contribution const c = contribution(p, s0, x, n);
    // exclusive positions inside the tile
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t  ik = c.keep, io = c.ops;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        uint32_t const uk = (uint32_t)__shfl_up((int)ik, off), uo = (uint32_t)__shfl_up((int)io, off);
        if (lane >= off)
        {
            ik += uk;
            io += uo;
        }
    }
    __shared__ uint32_t wk[kRecBlock / 64], wo[kRecBlock / 64];
    if (lane == 63)
    {
        wk[wave] = ik;
        wo[wave] = io;
    }
    __syncthreads();
    uint32_t bk = 0, bo = 0;
#pragma unroll
    for (int w = 0; w < kRecBlock / 64; ++w)
        if (w < wave)
        {
            bk += wk[w];
            bo += wo[w];
        }
    if (!c.keep)
        return;
    uint64_t const r      = (uint64_t)tile_keep[blockIdx.x] + bk + ik - 1;
    uint64_t const ops_at = p.ops_base + tile_ops[blockIdx.x] + bo + io - c.ops;
    Hsp const      a      = p.hsp[c.entry];
    uint32_t const w      = (uint32_t)s0[x];
    L2Window const W      = p.win[w];
    BlastMatchDev  bm;
    bm.qry_id  = W.q;
    bm.subj_id = W.s;
    bm.n_qid   = W.q / p.q_frames;
    bm.n_sid   = W.s / p.s_frames;
    // _expandAlign: the query slice is the whole (frame) query, the subject slice begins at the window's start (:1032-1035)
    bm.q_start = (uint64_t)(int64_t)a.q_begin;
    bm.q_end   = (uint64_t)(int64_t)a.q_end;
    bm.s_start = W.beg + (uint64_t)(int64_t)a.s_begin;
    bm.s_end   = W.beg + (uint64_t)(int64_t)a.s_end;
    bm.score   = a.score;
    bm.alignment_length   = a.n_ops;
    bm.num_matches        = a.num_matches;
    bm.num_mismatches     = a.num_mismatches;
    bm.num_positives      = a.num_positives;
    bm.num_gap_opens      = a.num_gap_opens;
    bm.num_gap_extensions = a.num_gap_extensions;
    bm.identity           = identity_of(a);
    {
#pragma clang fp contract(off) // (HIP's __dmul_rn / __dsub_rn are plain operators: without this the product and the difference fuse)
        double const prod = p.lambda * (double)a.score;
        double const diff = prod - p.log_k;
        bm.bit_score      = diff / p.log_2;
        uint32_t const s  = (uint32_t)max(a.score, 0);
        double const   ex = s < p.exp_n ? p.exp_tab[s] : 0.0; // (behind the table exp(-lambda s) is zero in the host's libm as well)
        bm.e_value        = p.pre_by_len[p.q_evidx[W.q]] * ex;
    }
    bm.ops_off = p.want_ops ? ops_at : 0;
    bm.n_ops   = (uint32_t)a.n_ops;
    bm.q_frame = (int16_t)frame_of(p.q_mode, W.q, false);
    bm.s_frame = (int16_t)frame_of(p.s_mode, W.s, true);
    p.rec[r] = bm;
    // (what the host threads need to expand the record's columns while the rows are still on their way: codes, columns, where to)
    p.rec_codes[3 * r]     = p.codes_off ? p.codes_off[c.entry] : (uint64_t)(uint32_t)a.ops_shift;
    p.rec_codes[3 * r + 1] = bm.ops_off;
    p.rec_codes[3 * r + 2] = (uint64_t)bm.n_ops;
}

// a chunk's survivors join the call's list (the pipeline's lane buffers are the next chunk's): records, window of each, where its
// run-length codes begin in the call's code bytes
__global__ __launch_bounds__(kRecBlock) void rec_append_kernel(Hsp const * hsp, uint32_t const * src, uint64_t const * count_ptr, uint64_t cap, uint64_t code_base,
                                                               Hsp * out_hsp, uint32_t * out_src, uint64_t * out_codes)
{
    uint64_t const e = (uint64_t)blockIdx.x * kRecBlock + threadIdx.x;
    uint64_t const n = min(*count_ptr, cap);
    if (e >= cap)
        return;
    if (e < n)
    {
        Hsp const a  = hsp[e];
        out_hsp[e]   = a;
        out_src[e]   = src[e];
        out_codes[e] = code_base + (uint32_t)a.ops_shift;
    }
    else
        out_src[e] = 0xffffffffu; // (behind the chunk's count: nothing)
}

// a range's rows carry column offsets inside the range: `base` makes them the result's
__global__ __launch_bounds__(kRecBlock) void rec_add_ops_base_kernel(BlastMatchDev * rec, uint64_t n, uint64_t base)
{
    uint64_t const r = (uint64_t)blockIdx.x * kRecBlock + threadIdx.x;
    if (r < n)
        rec[r].ops_off += base;
}

} // namespace

hipError_t rec_launch_add_ops_base(BlastMatchDev * rec, uint64_t n, uint64_t base, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(rec_add_ops_base_kernel, dim3((unsigned)((n + kRecBlock - 1) / kRecBlock)), dim3(kRecBlock), 0, stream, rec, n, base);
    return hipGetLastError();
}

hipError_t rec_launch_append(Hsp const * hsp, uint32_t const * src, uint64_t const * count_ptr, uint64_t cap, uint64_t code_base, Hsp * out_hsp, uint32_t * out_src,
                             uint64_t * out_codes, hipStream_t stream)
{
    if (cap == 0)
        return hipSuccess;
    hipLaunchKernelGGL(rec_append_kernel, dim3((unsigned)((cap + kRecBlock - 1) / kRecBlock)), dim3(kRecBlock), 0, stream, hsp, src, count_ptr, cap, code_base, out_hsp,
                       out_src, out_codes);
    return hipGetLastError();
}

hipError_t rec_launch_rank(L2Window const * win, uint64_t n, uint32_t q_frames, uint32_t const * q_len, uint32_t * rank, uint32_t * too_long, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(rec_rank_kernel, dim3((unsigned)((n + kRecBlock - 1) / kRecBlock)), dim3(kRecBlock), 0, stream, win, n, q_frames, q_len, rank, too_long);
    return hipGetLastError();
}

// p.counters must be zeroed, p.list_at filled with 0xff by the caller (stream order).  pair / s0: two buffers of n_entries words each.
hipError_t rec_launch(RecParams const & p, uint64_t ** pair, uint64_t ** pair_tmp, uint64_t ** s0, uint64_t ** s0_tmp, uint64_t pair_bits, uint64_t s0_bits,
                      uint32_t * ghist, uint32_t * tile_keep, uint64_t * tile_ops, hipStream_t stream)
{
    if (p.n_win)
    {
        if (p.n_entries)
            hipLaunchKernelGGL(rec_keys_kernel, dim3((unsigned)std::min<uint64_t>((p.n_entries + kRecBlock - 1) / kRecBlock, 1024)), dim3(kRecBlock), 0, stream, p, *pair, *s0);
        hipLaunchKernelGGL(rec_stats_kernel, dim3((unsigned)std::min<uint64_t>((p.n_win + kRecBlock - 1) / kRecBlock, 1024)), dim3(kRecBlock), 0, stream, p);
    }
    if (p.n_entries == 0)
        return hipGetLastError();
    hipError_t const e = l2_launch_sort(pair, pair_tmp, s0, s0_tmp, p.n_entries, pair_bits, s0_bits, ghist, stream);
    if (e != hipSuccess)
        return e;
    // (the number of real survivors stands in device memory: the grids cover every entry, the kernels stop at the count)
    uint64_t const * const n_ptr = p.counters + kRecSurvivors;
    unsigned const         tiles = (unsigned)((p.n_entries + kRecBlock - 1) / kRecBlock);
    uint64_t const * const order = p.rank ? *pair : *s0; // (the word whose low half names the window)
    hipLaunchKernelGGL(rec_tile_kernel, dim3(tiles), dim3(kRecBlock), 0, stream, p, order, n_ptr, tile_keep, tile_ops);
    hipLaunchKernelGGL(rec_tops_kernel, dim3(1), dim3(kRecBlock), 0, stream, p, n_ptr, tile_keep, tile_ops);
    hipLaunchKernelGGL(rec_write_kernel, dim3(tiles), dim3(kRecBlock), 0, stream, p, order, n_ptr, tile_keep, tile_ops);
    return hipGetLastError();
}

} // namespace lx

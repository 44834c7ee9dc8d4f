// lx_select.hip -- survivor selection between the two passes, on the device (gfx950 only).
//
// The filter loop of iterateMatchesFullSimd (/root/reference/src/search_algo.hpp:1251-1283) keeps a candidate iff
// bitScore >= minBitScore and eValue <= maxEValue.  Both are monotone in the raw score for a given query length
// (src/search_misc.hpp:77-78), so the host turns them into an integer score cut-off per extension (or one for all)
// and the test becomes `score >= cutoff` -- bit-identical decisions without moving scores to the host.
//
// Output: a compacted extension list in input order (deterministic: count per run + per-workgroup totals -> scan of
// the totals -> in-workgroup scan + write), the original
// index of every slot, and -- when the input comes in runs of `run` extensions per query -- each run's survivors padded
// with empty slots to a multiple of `pad_to`, so that every wavefront of pass 2 works on a single query (one LDS
// profile).  HBM-bound integer work: 28 B read per candidate, 28 B written per survivor.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "lx_device.h"

namespace lx
{

__device__ __forceinline__ bool survives(SelectParams const & p, uint64_t i)
{
    int32_t const cut = p.min_score ? p.min_score[i] : p.min_score_all;
    return p.score[i] >= cut && p.ext[i].q_len != 0 && p.ext[i].s_len != 0;
}

constexpr int kSelBlock = 256; // runs per workgroup of the count / write kernels

// exclusive scan of one value per thread over a workgroup of kSelBlock threads; returns the workgroup total in `total`
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t * wave_sums, uint32_t & total)
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
    for (int w = 0; w < kSelBlock / 64; ++w)
    {
        uint32_t const ws = wave_sums[w];
        base += (w < wave) ? ws : 0u;
        tot += ws;
    }
    total = tot;
    return base + incl - v;
}

// one thread per run: padded survivor count; one (slots, survivors) total per workgroup
__global__ __launch_bounds__(kSelBlock) void select_count_kernel(SelectParams p, uint64_t nruns)
{
    __shared__ uint32_t wave_sums[kSelBlock / 64], wave_true[kSelBlock / 64];
    uint64_t const r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t       c = 0;
    if (r < nruns)
    {
        uint64_t const lo = r * p.run, hi = min(p.n, lo + p.run);
        for (uint64_t i = lo; i < hi; ++i)
            c += survives(p, i) ? 1u : 0u;
    }
    uint32_t const padded = (c + p.pad_to - 1) / p.pad_to * p.pad_to;
    if (r < nruns)
        p.run_slots[r] = padded; // the write kernel recounts nothing: it only needs the padded size for its in-block scan
    uint32_t sp = padded, st = c;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        sp += (uint32_t)__shfl_xor((int)sp, off);
        st += (uint32_t)__shfl_xor((int)st, off);
    }
    if ((threadIdx.x & 63) == 0)
    {
        wave_sums[threadIdx.x >> 6] = sp;
        wave_true[threadIdx.x >> 6] = st;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint64_t a = 0, t = 0;
        for (int w = 0; w < kSelBlock / 64; ++w)
        {
            a += wave_sums[w];
            t += wave_true[w];
        }
        p.block_tot[2 * (uint64_t)blockIdx.x]     = a;
        p.block_tot[2 * (uint64_t)blockIdx.x + 1] = t;
    }
}

// single workgroup: exclusive scan over the per-workgroup totals (a few hundred to a few thousand values)
__global__ __launch_bounds__(1024) void select_scan_kernel(SelectParams p, uint64_t nblocks)
{
    __shared__ uint64_t part[1024];
    __shared__ uint64_t carry[2];
    if (threadIdx.x == 0)
        carry[0] = carry[1] = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nblocks; base += 1024)
    {
        uint64_t const b = base + threadIdx.x;
        uint64_t const v = b < nblocks ? p.block_tot[2 * b] : 0, t = b < nblocks ? p.block_tot[2 * b + 1] : 0;
        // Hillis-Steele over the tile in LDS (10 rounds; the tile count is tiny)
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1)
        {
            uint64_t const add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        uint64_t const incl = part[threadIdx.x], c0 = carry[0];
        if (b < nblocks)
            p.block_tot[2 * b] = c0 + incl - v; // exclusive offset of this workgroup's first slot
        __syncthreads();
        // survivors: plain tree sum
        part[threadIdx.x] = t;
        __syncthreads();
        for (int off = 512; off >= 1; off >>= 1)
        {
            if (threadIdx.x < (unsigned)off)
                part[threadIdx.x] += part[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 1023)
            carry[0] = c0 + incl;
        if (threadIdx.x == 0)
            carry[1] += part[0];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        p.out_count[0] = carry[0];
        p.out_count[1] = carry[1];
    }
}

// one thread per run: write survivors (input order) + padding slots
__global__ __launch_bounds__(kSelBlock) void select_write_kernel(SelectParams p, uint64_t nruns)
{
    __shared__ uint32_t wave_sums[kSelBlock / 64];
    uint64_t const r      = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t const mine   = r < nruns ? (uint32_t)p.run_slots[r] : 0u;
    uint32_t       total  = 0;
    uint32_t const before = block_exclusive_scan(mine, wave_sums, total);
    if (r >= nruns)
        return;
    uint64_t const lo = r * p.run, hi = min(p.n, lo + p.run);
    uint64_t       o  = p.block_tot[2 * (uint64_t)blockIdx.x] + before;
    uint32_t       c  = 0;
    Extension      last{};
    for (uint64_t i = lo; i < hi; ++i)
    {
        bool const keep = survives(p, i);
        if (keep)
        {
            last         = p.ext[i];
            p.out_ext[o]   = last;
            p.out_src[o]   = (uint32_t)i;
            p.out_score[o] = p.score[i];
            ++o;
            ++c;
        }
        else if (p.out_hsp)
        {
            Hsp h{};
            h.score      = p.score[i];
            p.out_hsp[i] = h; // filtered out: score only, no alignment
        }
    }
    uint32_t const padded = (c + p.pad_to - 1) / p.pad_to * p.pad_to;
    last.s_len            = 0;
    for (uint32_t k = c; k < padded; ++k, ++o)
    {
        p.out_ext[o]   = last; // same query slice, empty window
        p.out_src[o]   = 0xffffffffu;
        p.out_score[o] = 0;
    }
}

// ---- no padding (pad_to == 1: the single sweep's backtrace takes one lane per survivor): plain ordered compaction, one
// thread per extension -- coalesced reads, 0.29 -> 0.17 ms per headline step against one thread per run

__global__ __launch_bounds__(kSelBlock) void select_flat_count_kernel(SelectParams p)
{
    __shared__ uint32_t wave_sums[kSelBlock / 64];
    uint64_t const i    = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool const     keep = i < p.n && survives(p, i);
    uint32_t const c    = (uint32_t)__popcll(__ballot(keep));
    if ((threadIdx.x & 63) == 0)
        wave_sums[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint64_t t = 0;
        for (int w = 0; w < kSelBlock / 64; ++w)
            t += wave_sums[w];
        p.block_tot[2 * (uint64_t)blockIdx.x]     = t; // slots = survivors
        p.block_tot[2 * (uint64_t)blockIdx.x + 1] = t;
    }
}

__global__ __launch_bounds__(kSelBlock) void select_flat_write_kernel(SelectParams p)
{
    __shared__ uint32_t wave_sums[kSelBlock / 64];
    uint64_t const i     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool const     in    = i < p.n;
    bool const     keep  = in && survives(p, i);
    uint32_t       total = 0;
    uint32_t const before = block_exclusive_scan(keep ? 1u : 0u, wave_sums, total);
    if (!in)
        return;
    if (keep)
    {
        uint64_t const o = p.block_tot[2 * (uint64_t)blockIdx.x] + before;
        p.out_ext[o]     = p.ext[i];
        p.out_src[o]     = (uint32_t)i;
        p.out_score[o]   = p.score[i];
    }
    else if (p.out_hsp)
    {
        Hsp h{};
        h.score      = p.score[i];
        p.out_hsp[i] = h; // filtered out: score only, no alignment
    }
}

// workgroups of the count / write kernels = entries of SelectParams::block_tot (two uint64 each)
// ---- lx_extend_batch's multi-query plan: the plan's slot order is a permutation of the caller's list (sub-blocks sorted by
// length across queries), so the per-slot records are GATHERED from a device copy of the caller's list and the scores
// SCATTERED back into caller order here -- random accesses of 24-byte records at HBM speed instead of cache misses on the host.
// orig[o] = index in the caller's list | 0x80000000 for a filler slot (a copy of its sub-block's last window that never survives).
__global__ __launch_bounds__(256) void slot_gather_kernel(Extension const * __restrict__ ext_all, int32_t const * __restrict__ min_all, int32_t min_score_all,
                                                          uint32_t const * __restrict__ orig, uint64_t slots, Extension * __restrict__ out_ext,
                                                          int32_t * __restrict__ out_min)
{
    uint64_t const o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= slots)
        return;
    uint32_t const so = orig[o], i = so & 0x7fffffffu;
    out_ext[o] = ext_all[i];
    out_min[o] = (so >> 31) ? 0x7fffffff : (min_all ? min_all[i] : min_score_all);
}

// score of every real slot into caller order; the survivor list's slot numbers (src[e]) become caller indices
__global__ __launch_bounds__(256) void slot_scatter_kernel(uint32_t const * __restrict__ orig, uint64_t slots, int32_t const * __restrict__ score,
                                                           int32_t * __restrict__ score_all, uint32_t * __restrict__ src, uint64_t const * __restrict__ count_ptr,
                                                           uint64_t cap)
{
    uint64_t const o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < slots)
    {
        uint32_t const so = orig[o];
        if (!(so >> 31))
            score_all[so] = score[o];
    }
    if (o < cap && o < count_ptr[0])
    {
        uint32_t const slot = src[o];
        src[o]              = slot == 0xffffffffu ? slot : (orig[slot] & 0x7fffffffu);
    }
}

hipError_t launch_slot_gather(Extension const * ext_all, int32_t const * min_all, int32_t min_score_all, uint32_t const * orig, uint64_t slots,
                              Extension * out_ext, int32_t * out_min, hipStream_t stream)
{
    if (slots == 0)
        return hipSuccess;
    hipLaunchKernelGGL(slot_gather_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, ext_all, min_all, min_score_all, orig, slots,
                       out_ext, out_min);
    return hipGetLastError();
}

hipError_t launch_slot_scatter(uint32_t const * orig, uint64_t slots, int32_t const * score, int32_t * score_all, uint32_t * src,
                               uint64_t const * count_ptr, uint64_t cap, hipStream_t stream)
{
    uint64_t const n = std::max(slots, cap);
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(slot_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, orig, slots, score, score_all, src, count_ptr, cap);
    return hipGetLastError();
}

uint64_t select_blocks(uint64_t nruns) { return (nruns + kSelBlock - 1) / kSelBlock; }

hipError_t launch_select(SelectParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipMemsetAsync(p.out_count, 0, 2 * sizeof(uint64_t), stream);
    if (p.pad_to <= 1)
    {
        // (block_tot was sized for one entry per kSelBlock runs of >= 1 extension: at run == 1 that is exactly this grid;
        // the host sizes it with select_blocks(n) for the unpadded case)
        unsigned const fb = (unsigned)select_blocks(p.n);
        hipLaunchKernelGGL(select_flat_count_kernel, dim3(fb), dim3(kSelBlock), 0, stream, p);
        hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, stream, p, (uint64_t)fb);
        hipLaunchKernelGGL(select_flat_write_kernel, dim3(fb), dim3(kSelBlock), 0, stream, p);
        return hipGetLastError();
    }
    uint64_t const nruns = (p.n + p.run - 1) / p.run;
    unsigned const b     = (unsigned)select_blocks(nruns);
    hipLaunchKernelGGL(select_count_kernel, dim3(b), dim3(kSelBlock), 0, stream, p, nruns);
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, stream, p, (uint64_t)b);
    hipLaunchKernelGGL(select_write_kernel, dim3(b), dim3(kSelBlock), 0, stream, p, nruns);
    return hipGetLastError();
}

} // namespace lx

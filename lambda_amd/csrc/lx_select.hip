// lx_select.hip -- survivor selection between the two passes, on the device (gfx950 only).
//
// The filter loop of iterateMatchesFullSimd (/root/reference/src/search_algo.hpp:1251-1283) keeps a candidate iff
// bitScore >= minBitScore and eValue <= maxEValue.  Both are monotone in the raw score for a given query length
// (src/search_misc.hpp:77-78), so the host turns them into an integer score cut-off per extension (or one for all)
// and the test becomes `score >= cutoff` -- bit-identical decisions without moving scores to the host.
//
// Output: a compacted extension list in input order (deterministic: count -> exclusive scan -> write), the original
// index of every slot, and -- when the input comes in runs of `run` extensions per query -- each run's survivors padded
// with empty slots to a multiple of `pad_to`, so that every wavefront of pass 2 works on a single query (one LDS
// profile).  HBM-bound integer work: 28 B read per candidate, 28 B written per survivor.
#include <hip/hip_runtime.h>

#include "lx_device.h"

namespace lx
{

__device__ __forceinline__ bool survives(SelectParams const & p, uint64_t i)
{
    int32_t const cut = p.min_score ? p.min_score[i] : p.min_score_all;
    return p.score[i] >= cut && p.ext[i].q_len != 0 && p.ext[i].s_len != 0;
}

// one thread per run: padded survivor count
__global__ __launch_bounds__(256) void select_count_kernel(SelectParams p, uint64_t nruns)
{
    uint64_t const r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns)
        return;
    uint64_t const lo = r * p.run, hi = min(p.n, lo + p.run);
    uint32_t       c  = 0;
    for (uint64_t i = lo; i < hi; ++i)
        c += survives(p, i) ? 1u : 0u;
    p.run_slots[r] = (uint64_t)((c + p.pad_to - 1) / p.pad_to) * p.pad_to | ((uint64_t)c << 40);
}

// single workgroup exclusive scan over the per-run slot counts (nruns is a few 1e5 at most)
__global__ __launch_bounds__(1024) void select_scan_kernel(SelectParams p, uint64_t nruns)
{
    __shared__ uint64_t part[1024];
    __shared__ uint64_t part_true[1024];
    uint64_t const per = (nruns + 1023) / 1024;
    uint64_t const lo = threadIdx.x * per, hi = min(nruns, lo + per);
    uint64_t       sum = 0, tsum = 0;
    for (uint64_t r = lo; r < hi; ++r)
    {
        sum += p.run_slots[r] & 0xffffffffffull;
        tsum += p.run_slots[r] >> 40;
    }
    part[threadIdx.x]      = sum;
    part_true[threadIdx.x] = tsum;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint64_t acc = 0, tacc = 0;
        for (int t = 0; t < 1024; ++t)
        {
            uint64_t const v = part[t];
            part[t]          = acc;
            acc += v;
            tacc += part_true[t];
        }
        p.out_count[0] = acc;
        p.out_count[1] = tacc;
    }
    __syncthreads();
    uint64_t acc = part[threadIdx.x];
    for (uint64_t r = lo; r < hi; ++r)
    {
        uint64_t const v = p.run_slots[r] & 0xffffffffffull;
        p.run_slots[r]   = acc;
        acc += v;
    }
}

// one thread per run: write survivors (input order) + padding slots
__global__ __launch_bounds__(256) void select_write_kernel(SelectParams p, uint64_t nruns)
{
    uint64_t const r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns)
        return;
    uint64_t const lo = r * p.run, hi = min(p.n, lo + p.run);
    uint64_t       o  = p.run_slots[r];
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

hipError_t launch_select(SelectParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipMemsetAsync(p.out_count, 0, 2 * sizeof(uint64_t), stream);
    uint64_t const nruns = (p.n + p.run - 1) / p.run;
    unsigned const b     = (unsigned)((nruns + 255) / 256);
    hipLaunchKernelGGL(select_count_kernel, dim3(b), dim3(256), 0, stream, p, nruns);
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, stream, p, nruns);
    hipLaunchKernelGGL(select_write_kernel, dim3(b), dim3(256), 0, stream, p, nruns);
    return hipGetLastError();
}

} // namespace lx

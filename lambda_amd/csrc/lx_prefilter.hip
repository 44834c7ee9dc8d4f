// lx_prefilter.hip -- pre-extension filter on the GPU (gfx950 only).
//
// Replaces seedLooksPromising() (/root/reference/src/search_algo.hpp:426-481): an ungapped max-segment score along
// the seed's diagonal over a region of max(seedLength * preScoring, seed length) residues centred on the seed,
// compared with preScoringThresh * region length.  The reference calls it once per located seed hit (:744-751), so
// it removes most candidates before any DP runs.  Byte work: one lane per seed, four diagonal cells per iteration from
// one unaligned dword per sequence, the scoring matrix in LDS.
#include <hip/hip_runtime.h>

#include "lx_dp_common.h"

namespace lx
{

__global__ __launch_bounds__(256) void prefilter_kernel(PrefilterParams p)
{
    __shared__ int8_t smat[kAlph * kAlph];
    for (int t = threadIdx.x; t < kAlph * kAlph / 4; t += blockDim.x)
        reinterpret_cast<uint32_t *>(smat)[t] = reinterpret_cast<uint32_t const *>(p.sc->mat)[t];
    __syncthreads();
    uint64_t const x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= p.n)
        return;
    PrefilterSeed const sd = p.seeds[x];
    // search_algo.hpp:429-453, same integer types and order of operations
    int64_t  effQ   = sd.qry_start;
    int64_t  effS   = sd.subj_start;
    uint64_t actual = (uint64_t)sd.qry_end - sd.qry_start;
    uint64_t effLen = (uint64_t)((int64_t)p.seed_length * p.pre_scoring);
    if (effLen < actual)
        effLen = actual;
    if (effLen > actual)
    {
        effQ -= (int64_t)((effLen - actual) / 2);
        effS -= (int64_t)((effLen - actual) / 2);
        int64_t const mn = effQ < effS ? effQ : effS;
        if (mn < 0)
        {
            effQ -= mn;
            effS -= mn;
            effLen += (uint64_t)mn;
        }
        uint64_t const a = (uint64_t)((int64_t)sd.q_len - effQ), b = (uint64_t)((int64_t)sd.s_len - effS);
        if (a < effLen)
            effLen = a;
        if (b < effLen)
            effLen = b;
    }
    uint8_t const * q   = p.q_res + sd.q_off + effQ;
    uint8_t const * s   = p.s_res + sd.s_off + effS;
    int             sco = 0, mx = 0;
    int const       thresh = (int)(p.pre_scoring_thresh * (double)effLen);
    // The reference returns as soon as the running maximum reaches the threshold (:472-478); the maximum never
    // decreases, so "reached at some point" == "reached at the end or at an earlier exit".  Four diagonal cells per
    // iteration: one unaligned dword per sequence (the buffers carry 256 bytes of slack), the matrix from LDS.
    bool reached = false;
    for (uint64_t i = 0; i < effLen && !reached; i += 4)
    {
        uint32_t const qw = *reinterpret_cast<unaligned_u32 const *>(q + i), sw = *reinterpret_cast<unaligned_u32 const *>(s + i);
#pragma unroll
        for (int b = 0; b < 4; ++b)
        {
            if (i + (uint64_t)b < effLen)
            {
                sco += smat[((qw >> (8 * b)) & (kAlph - 1)) * kAlph + ((sw >> (8 * b)) & (kAlph - 1))];
                if (sco < 0)
                    sco = 0;
                else if (sco > mx)
                    mx = sco;
            }
        }
        reached = mx >= thresh;
    }
    uint8_t const keep = (effLen > 0 && reached) ? 1 : 0;
    p.out_keep[x] = keep;
}

hipError_t launch_prefilter(PrefilterParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    uint64_t const blocks = (p.n + 255) / 256;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(prefilter_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

} // namespace lx

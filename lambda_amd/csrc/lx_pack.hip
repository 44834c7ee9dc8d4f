// lx_pack.hip -- run-length packing of the traceback's op bytes for the host-buffer entry points (gfx950 only).
//
// The backtrace kernels write one op byte per alignment column ('M', 'D', 'I') into worst-case slots of q_len + s_len
// bytes: 326 B per 150 x 176 extension, of which an alignment has a handful of runs.  What crosses PCIe in
// lx_extend_batch is therefore the run-length form: one byte per run, (op << 6) | (length - 1) with op 0 = 'M',
// 1 = 'D', 2 = 'I', runs longer than 64 columns split (remainder first) -- begin -> end order, the n_ops of the record say where a
// survivor's codes end.  One lane per survivor: two passes over its op bytes (count the codes; write them), space in the
// dense code stream handed out per wavefront (one atomic per wavefront), the record's ops_shift becomes the offset of
// the survivor's first code in that stream.  The gapped rows the reference keeps after _adaptTraceSegmentsTo
// (/root/reference/src/search_algo.hpp:1127) are run lengths as well (seqan::ArrayGaps), so this is also the natural form
// for a binding that fills them.
#include <hip/hip_runtime.h>

#include "lx_device.h"

namespace lx
{

__device__ __forceinline__ uint32_t op_code(uint32_t op)
{
    return op == (uint32_t)'M' ? 0u : (op == (uint32_t)'D' ? 1u : 2u);
}

__global__ __launch_bounds__(256) void rle_pack_kernel(PackParams p)
{
    uint64_t const e     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t const limit = p.count_ptr ? min(p.n, *p.count_ptr) : p.n;
    bool const     mine  = e < limit && (!p.src || p.src[e] != 0xffffffffu);
    Hsp            h{};
    uint8_t const * ops = nullptr;
    uint32_t        ncodes = 0;
    if (mine)
    {
        h = p.hsp[e];
        if (h.score > 0 && h.n_ops > 0)
        {
            ops = p.ops + (p.ops_off ? p.ops_off[e] : e * p.ops_stride) + (uint32_t)h.ops_shift;
            uint32_t cur = 0xffu, len = 0;
            for (int32_t k = 0; k < h.n_ops; ++k)
            {
                uint32_t const op = ops[k];
                if (op != cur || len == 64)
                {
                    ++ncodes;
                    cur = op;
                    len = 0;
                }
                ++len;
            }
        }
    }
    // space in the dense stream: exclusive prefix over the wavefront, one atomic for all of it
    uint32_t incl = ncodes;
    int const lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        uint32_t const up = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off)
            incl += up;
    }
    uint32_t const total = (uint32_t)__shfl((int)incl, 63);
    unsigned long long base = 0;
    if (lane == 63 && total != 0)
        base = atomicAdd(p.rle_top, (unsigned long long)total);
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 63);
    if (e < p.n && p.rle_len)
        p.rle_len[e] = mine ? ncodes : 0u;
    if (!mine)
        return;
    uint64_t const at = base + incl - ncodes;
    if (ncodes != 0)
    {
        if (at + ncodes > p.rle_cap)
        {
            atomicExch(p.err, 5); // the host sized the stream by the worst case: cannot happen, reported all the same
            h.score = -1;
        }
        else
        {
            // written back to front, so that a run of more than 64 columns is cut where the checkpoint backtrace cuts it (it
            // walks end -> begin and emits codes itself: the remainder comes first, the 64s behind it) -- one alignment, one code
            // string, whichever pass-2 mode produced it
            uint8_t * out = p.rle + at + ncodes;
            uint32_t  cur = ops[h.n_ops - 1], len = 0;
            for (int32_t k = h.n_ops - 1; k >= 0; --k)
            {
                uint32_t const op = ops[k];
                if (op != cur || len == 64)
                {
                    *--out = (uint8_t)((op_code(cur) << 6) | (len - 1));
                    cur    = op;
                    len    = 0;
                }
                ++len;
            }
            *--out = (uint8_t)((op_code(cur) << 6) | (len - 1));
        }
    }
    h.ops_shift = (int32_t)(uint32_t)at; // (a chunk's stream stays far below 2^31 bytes)
    p.hsp[e]    = h;
}

hipError_t launch_rle_pack(PackParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    uint64_t const blocks = (p.n + 255) / 256;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(rle_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

} // namespace lx

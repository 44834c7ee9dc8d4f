// lx_scan.h -- tiled scans over device lists (gfx950): sum / max / min of one uint32 value per element, forwards or from the right,
// as three kernels -- the combined value of every tile, one workgroup that turns them into what precedes each tile, and the pass
// that hands every element its inclusive and exclusive value.  Values come from a functor `val(element)`, results go to a functor
// `out(element, inclusive, exclusive)`: the list work of the Level-2 driver (lx_level2.hip: merge right, swallow left, unique as
// scans -- /root/reference/src/search_algo.hpp:1144-1173) and the plan of the multi-query sweep (lx_plan_free.hip) are written as
// pairs of such functors.  Included inside `namespace lx { namespace { ... } }` of a .hip file (the kernels are that file's own).
#pragma once

// ---- scans over the sorted list ------------------------------------------------------------------------------------------

enum
{
    kOpSum = 0,
    kOpMax = 1,
    kOpMin = 2
};
template <int OP>
__device__ __forceinline__ uint32_t op_apply(uint32_t a, uint32_t b)
{
    return OP == kOpSum ? a + b : OP == kOpMax ? max(a, b) : min(a, b);
}
template <int OP>
__device__ __forceinline__ uint32_t op_ident()
{
    return OP == kOpMin ? 0xffffffffu : 0u;
}

// inclusive scan of one value per thread over the workgroup; `total` = the workgroup's combined value
template <int OP>
__device__ __forceinline__ uint32_t block_inclusive(uint32_t v, uint32_t * wave_tot, uint32_t & total)
{
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t  incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        uint32_t const up = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off)
            incl = op_apply<OP>(up, incl);
    }
    if (lane == 63)
        wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = op_ident<OP>(), tot = op_ident<OP>();
#pragma unroll
    for (int w = 0; w < kL2ScanBlock / 64; ++w)
    {
        uint32_t const x = wave_tot[w];
        if (w < wave)
            before = op_apply<OP>(before, x);
        tot = op_apply<OP>(tot, x);
    }
    total = tot;
    __syncthreads();
    return op_apply<OP>(before, incl);
}

// Scan position j of a tile maps to list element j (forward) or n - 1 - j (REV: the scan runs from the right).
template <bool REV>
__device__ __forceinline__ uint64_t elem_of(uint64_t j, uint64_t n)
{
    return REV ? n - 1 - j : j;
}

// (a) the combined value of every tile (the operations commute: the elements are taken workgroup-wide, consecutive lanes on
// consecutive elements)
template <int OP, bool REV, class Val>
__global__ __launch_bounds__(kL2ScanBlock) void l2_scan_reduce_kernel(Val val, uint64_t n, uint32_t * block_tot)
{
    __shared__ uint32_t wave_tot[kL2ScanBlock / 64];
    uint64_t const      j0 = (uint64_t)blockIdx.x * kL2ScanTile + threadIdx.x;
    uint32_t            acc = op_ident<OP>();
#pragma unroll
    for (int k = 0; k < kL2ScanItems; ++k)
    {
        uint64_t const j = j0 + (uint64_t)k * kL2ScanBlock;
        if (j < n)
            acc = op_apply<OP>(acc, val(elem_of<REV>(j, n)));
    }
    uint32_t total;
    (void)block_inclusive<OP>(acc, wave_tot, total);
    if (threadIdx.x == 0)
        block_tot[blockIdx.x] = total;
}

// (b) one workgroup: the tiles' values become what precedes each tile (exclusive); block_tot[tiles] = the whole list's value
template <int OP>
__global__ __launch_bounds__(kL2ScanBlock) void l2_scan_tops_kernel(uint32_t * block_tot, uint64_t tiles)
{
    __shared__ uint32_t wave_tot[kL2ScanBlock / 64];
    __shared__ uint32_t incl_all[kL2ScanBlock];
    uint32_t            carry = op_ident<OP>();
    for (uint64_t b0 = 0; b0 < tiles; b0 += kL2ScanBlock)
    {
        uint64_t const b = b0 + threadIdx.x;
        uint32_t const v = b < tiles ? block_tot[b] : op_ident<OP>();
        uint32_t       total;
        incl_all[threadIdx.x] = block_inclusive<OP>(v, wave_tot, total);
        __syncthreads();
        uint32_t const ex = threadIdx.x == 0 ? op_ident<OP>() : incl_all[threadIdx.x - 1];
        if (b < tiles)
            block_tot[b] = op_apply<OP>(carry, ex);
        carry = op_apply<OP>(carry, total);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        block_tot[tiles] = carry;
}

// (c) every element's scan value: out(element, inclusive value, exclusive value).  Values in and results out go through LDS so that
// both the reads of `val` and the writes of `out` run over consecutive elements in consecutive lanes; in between every thread scans
// its kL2ScanItems consecutive values (position j stands at j + j / 8: a thread's run of eight starts in a bank of its own).
__device__ __forceinline__ uint32_t scan_pos(uint32_t j)
{
    return j + (j >> 3);
}
template <int OP, bool REV, class Val, class Out>
__global__ __launch_bounds__(kL2ScanBlock) void l2_scan_apply_kernel(Val val, Out out, uint64_t n, uint32_t const * block_tot)
{
    static_assert(kL2ScanItems == 8, "scan_pos spreads runs of eight");
    __shared__ uint32_t wave_tot[kL2ScanBlock / 64];
    __shared__ uint32_t incl_all[kL2ScanBlock];
    __shared__ uint32_t stage[kL2ScanTile + kL2ScanTile / 8 + 2]; // values, then the inclusive results behind the tile's carry-in
    uint64_t const      t0 = (uint64_t)blockIdx.x * kL2ScanTile;
#pragma unroll
    for (int k = 0; k < kL2ScanItems; ++k)
    {
        uint32_t const j = (uint32_t)k * kL2ScanBlock + threadIdx.x;
        stage[scan_pos(j)] = t0 + j < n ? val(elem_of<REV>(t0 + j, n)) : op_ident<OP>();
    }
    __syncthreads();
    uint32_t v[kL2ScanItems];
    uint32_t acc = op_ident<OP>();
#pragma unroll
    for (int k = 0; k < kL2ScanItems; ++k)
    {
        v[k] = stage[scan_pos(threadIdx.x * kL2ScanItems + k)];
        acc  = op_apply<OP>(acc, v[k]);
    }
    uint32_t       total;
    uint32_t const incl = block_inclusive<OP>(acc, wave_tot, total); // (its barriers also end the reads of `stage`)
    incl_all[threadIdx.x] = incl;
    __syncthreads();
    uint32_t const carry = block_tot[blockIdx.x];
    uint32_t run = op_apply<OP>(carry, threadIdx.x == 0 ? op_ident<OP>() : incl_all[threadIdx.x - 1]);
    // stage[scan_pos(j + 1)] = inclusive value of j; stage[scan_pos(0)] = what precedes the tile
    if (threadIdx.x == 0)
        stage[0] = carry;
#pragma unroll
    for (int k = 0; k < kL2ScanItems; ++k)
    {
        run = op_apply<OP>(run, v[k]);
        stage[scan_pos(threadIdx.x * kL2ScanItems + k + 1)] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kL2ScanItems; ++k)
    {
        uint32_t const j = (uint32_t)k * kL2ScanBlock + threadIdx.x;
        if (t0 + j < n)
            out(elem_of<REV>(t0 + j, n), stage[scan_pos(j + 1)], stage[scan_pos(j)]);
    }
}

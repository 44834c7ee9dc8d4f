// lx_dp_common.h -- device helpers shared by the score (pass 1) and trace (pass 2) kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "lx_device.h"

namespace lx
{

__device__ __forceinline__ int max3i(int a, int b, int c)
{
    return max(a, max(b, c)); // v_max3_i32
}

// Zero-cost optimisation fence on one VGPR value: stops LLVM from re-associating the max/add chains of
// consecutive (unrolled) DP steps into wider, instruction-hungrier trees.
#define LX_OPAQUE(x) asm volatile("" : "+v"(x))

// value of lane-1 within the group; first lane of each group gets `boundary`
template <int G>
__device__ __forceinline__ int shift_from_left(int x, int boundary, bool is_first)
{
    if constexpr (G == 16)
    {
        // DPP row_shr:1 -- lanes 0,16,32,48 keep `old` (= boundary)
        return __builtin_amdgcn_update_dpp(boundary, x, 0x111, 0xf, 0xf, false);
    }
    else if constexpr (G < 16)
    {
        int y = __builtin_amdgcn_update_dpp(boundary, x, 0x111, 0xf, 0xf, false);
        return is_first ? boundary : y;
    }
    else
    {
        // DPP wave_shr:1 -- whole-wave shift, lane 0 keeps `old`
        int y = __builtin_amdgcn_update_dpp(boundary, x, 0x138, 0xf, 0xf, false);
        return is_first ? boundary : y;
    }
}

template <int G, int C>
struct ScoreGeo
{
    static constexpr int kGroups = 64 / G;       // extensions per wavefront
    static constexpr int kD      = (C + 3) / 4;  // profile dwords per lane per row
    static constexpr int kPanel  = G * C;        // query columns per panel
    static constexpr int kRowDw  = kD * G;       // dwords per profile row
};

// Build the profile rows for this lane's C columns of the panel starting at column col0 into LDS slot `slot_dw`.
template <int G, int C>
__device__ __forceinline__ void build_profile(uint32_t * lds, int slot_dw, int g, uint8_t const * q, int lq, int col0,
                                              int8_t const * table, int nrows, bool do_write)
{
    using Geo = ScoreGeo<G, C>;
#pragma unroll 1 // one group of 4 columns at a time: unrolled, the 2 x 16 B loads of all C columns would be hoisted (8 C VGPRs)
    for (int d = 0; d < Geo::kD; ++d)
    {
        uint32_t rows[4][8];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
        {
            int const c  = 4 * d + cc;
            int const j  = col0 + c;
            uint32_t  ql = kAlph - 1; // pad rank: row of kNegPad
            if (c < C && j < lq)
                ql = q[j] & (kAlph - 1);
            uint4 const * mrow = reinterpret_cast<uint4 const *>(table + ql * kAlph);
            uint4 const   lo = mrow[0], hi = mrow[1];
            rows[cc][0] = lo.x; rows[cc][1] = lo.y; rows[cc][2] = lo.z; rows[cc][3] = lo.w;
            rows[cc][4] = hi.x; rows[cc][5] = hi.y; rows[cc][6] = hi.z; rows[cc][7] = hi.w;
        }
        if (do_write)
        {
            uint32_t * dst = lds + slot_dw + d * G + g;
#pragma unroll
            for (int w = 0; w < 8; ++w)
            {
                if (4 * w < nrows) // wave-uniform; rows beyond nrows are never read
                {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                    {
                        // byte b of the matrix rows of columns 0..3 -> one dword [c0,c1,c2,c3] for subject letter 4w+b
                        uint32_t const sel = (uint32_t)b | ((uint32_t)(4 + b) << 8) | 0x0c0c0000u;
                        uint32_t const x01 = __builtin_amdgcn_perm(rows[1][w], rows[0][w], sel);
                        uint32_t const x23 = __builtin_amdgcn_perm(rows[3][w], rows[2][w], sel);
                        dst[(4 * w + b) * Geo::kRowDw] = x01 | (x23 << 16);
                    }
                }
            }
        }
    }
}

typedef uint32_t __attribute__((aligned(1))) unaligned_u32;

// _bandSize, /root/reference/src/search_misc.hpp:46-50: int64(sqrt(double(len))) + 1, here with an exact integer root
__device__ __forceinline__ int band_size_dev(int len)
{
    int r = (int)sqrtf((float)len);
    while (r * r > len)
        --r;
    while ((r + 1) * (r + 1) <= len)
        ++r;
    return r + 1;
}
// Band mode without per-extension centres: the seed diagonal of a window built by _widenMatch (src/search_algo.hpp:919-938)
// crosses the window's first column b = _bandSize(Lq) rows down -- unless the window was clipped at the subject's start,
// which only the caller can know (then it passes the centres).
__device__ __forceinline__ int band_default_diag(int lq, int ls)
{
    return max(0, min(band_size_dev(lq), ls - lq));
}

} // namespace lx

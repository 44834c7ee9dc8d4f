// lx_score_i16.hip -- the single sweep in packed int16: two extensions per lane group, any query width (gfx950 only).
//
// Same reference seam as lx_score_f16.hip (_performAlignment, /root/reference/src/search_algo.hpp:1070-1134: pass 1 at
// :1246 and the forward half of pass 2 at :1296 as one sweep).  The packed-half kernel is exact only while every
// intermediate stays <= 2046, which rules out queries beyond ~300 residues.  Here the values are 16-bit INTEGERS --
// additions are v_pk_add_u16, exact over the whole range -- and only the maxima borrow the half-precision unit: gfx950
// has no three-input packed integer maximum (two v_pk_max_i16 per max3 make packed int16 cost what int32 costs:
// measured, 11 issue-slot pairs per two cells), but the bit patterns 0x0000 .. 0x7BFF are the non-negative finite
// halves in increasing order, so v_pk_maximum3_f16 on the raw bits IS the integer maximum as long as every value
// stays in that range (FP16 denormals are on: nothing is flushed, a maximum returns one operand unchanged).  All
// values therefore carry a bias that keeps them positive; 7.5 instructions per two cells like the half kernel, with
// 15 bits of range instead of 11.
//
// What differs from score_pair_kernel<G,C,CKPT=true>:
//   * values = skewed value + kBias as uint16, "minus infinity" = 0 (only ever an operand of max); every finite value
//     stays in 1 .. 0x7BFF, proven per wavefront before any DP work exactly like the half kernel's exactness test;
//   * queries wider than one panel: the panels are swept one after the other (lx_score.hip's carry workspace holds the
//     packed (H, E) of the last strip per subject row), every panel into its own part of the extension's slot;
//   * checkpoints are the int16 pairs of lx_ckpt.hip's int32 kernel -- a packed register already IS two int16, one
//     v_perm_b32 re-pairs (H, E) / (H, F) per extension -- so the backtrace and the int32 fix-up launch
//     (ckpt_forward_kernel<G,C,false,MULTI>, TraceParams::fixup) work on the same slots.
#include <hip/hip_runtime.h>

#include "lx_dp_common.h"

namespace lx
{

typedef unsigned short s2 __attribute__((ext_vector_type(2)));
typedef _Float16       f2 __attribute__((ext_vector_type(2)));

#ifndef LX_I16_UNROLL
#define LX_I16_UNROLL 2
#endif
#define LX_I16_PRAGMA(x) _Pragma(#x)
#define LX_I16_UNROLL_N(n) LX_I16_PRAGMA(unroll n)

// integer maxima of patterns in 0 .. 0x7BFF through the half-precision comparators (see the header)
__device__ __forceinline__ s2 smax(s2 a, s2 b)
{
    return __builtin_bit_cast(s2, __builtin_elementwise_maximum(__builtin_bit_cast(f2, a), __builtin_bit_cast(f2, b)));
}
__device__ __forceinline__ s2 smax3(s2 a, s2 b, s2 c)
{
    return __builtin_bit_cast(s2, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(f2, a), __builtin_bit_cast(f2, b)),
                                                                __builtin_bit_cast(f2, c))); // v_pk_maximum3_f16
}
__device__ __forceinline__ s2 ssplat(int x) { return s2{(unsigned short)x, (unsigned short)x}; }
__device__ __forceinline__ s2 as_s2(uint32_t x) { return __builtin_bit_cast(s2, x); }
__device__ __forceinline__ uint32_t as_u32(s2 x) { return __builtin_bit_cast(uint32_t, x); }

constexpr uint32_t kI16NegInf2 = 0u;             // "minus infinity": below every biased value
constexpr int      kBias       = 2048;           // keeps every finite skewed value positive (pad scores, the skew of the first rows)
constexpr int      kI16Limit   = 0x7BFF - kBias; // every finite intermediate stays below this

template <int G, int C>
struct Pair16Geo
{
    static constexpr int kGroups = 64 / G;             // lane groups per wavefront, two extensions each
    static constexpr int kPanel  = G * C;
    static constexpr int kUsedDw = (C + 1) / 2;        // dwords that hold real columns
    static constexpr int kLaneDw = (kUsedDw + 3) & ~3; // int16 entries of a lane's columns per profile row, 16-byte granules
    static constexpr int kRowDw  = kLaneDw * G;
    static constexpr int kCkDw   = (C + 3) / 4 * 4;    // lx_ckpt.hip: CkptLayout<G, C>::kCkDw
    __host__ __device__ static constexpr uint64_t bnd_dwords(uint32_t steps_cap) { return (uint64_t)steps_cap * G; }
    __host__ __device__ static constexpr uint64_t slot_dwords(uint32_t steps_cap)
    {
        return bnd_dwords(steps_cap) + (uint64_t)(steps_cap / 16) * G * kCkDw; // = CkptLayout<G, C>::slot_dwords
    }
};

// CKPT = false: score only (pass 1 of queries wider than the packed-half geometries): no slots, no end cells.
// COMPACT = the slots hold the 16-bit codes of Ckpt16Layout instead of int16 pairs (H <= 2047, a gap's first character
// <= 31: the conditions of the packed-half sweep, whose slots these are): in the integer domain a code is two subtractions
// and a shift-or.  One panel: the bound on the intermediates is tested up front, like the packed-half kernel does.  MULTI
// (queries wider than a panel: one part of compact codes per panel): no bound a real protein query passes would admit
// them, so the int16 range is tested up front and the code range afterwards -- an extension whose best score is beyond
// 2046 leaves the sentinel and is redone by the int32 launch into an overflow slot, like one the other test declines.
template <int G, int C, bool MULTI, bool CKPT = true, bool COMPACT = false>
__global__ __launch_bounds__(64, 3) void sweep_pair16_kernel(ScoreParams p)
{
    static_assert(C <= 24, "profile rows hold 24 entries per lane");
    static_assert(!COMPACT || CKPT, "compact codes belong to the checkpointing sweep");
    using L16 = Ckpt16Layout<G, C>;
    using Geo = Pair16Geo<G, C>;
    extern __shared__ uint32_t lds[];

    int const  lane     = threadIdx.x;
    int const  grp      = lane / G;
    int const  g        = lane % G;
    bool const is_first = (g == 0);
    bool const is_last  = (g == G - 1);

    uint64_t const pair = (uint64_t)blockIdx.x * Geo::kGroups + grp;
    uint64_t const eA   = 2 * pair, eB = 2 * pair + 1;
    bool const     actA = eA < p.n, actB = eB < p.n;

    ScoringDev const * __restrict__ sc = p.sc;
    int const      ge    = sc->ge;
    int const      nrows = p.nrows;
    uint32_t const padt  = (uint32_t)(nrows - 1);

    int             lq = 0, lsA = 0, lsB = 0;
    uint8_t const * q  = p.q_res;
    uint8_t const * sA = p.s_res;
    uint8_t const * sB = p.s_res;
    uint64_t        q_off = 0;
    if (actA)
    {
        Extension const x = p.ext[eA];
        lq    = (int)x.q_len;
        q_off = x.q_off;
        q += x.q_off;
        lsA = (int)x.s_len;
        if (lsA != 0)
            sA += x.s_off;
    }
    uint64_t q_offB = q_off;
    int      lqB    = lq;
    if (actB)
    {
        Extension const x = p.ext[eB];
        lqB    = (int)x.q_len;
        q_offB = x.q_off;
        lsB    = (int)x.s_len;
        if (lsB != 0)
            sB += x.s_off;
    }
    {
        // the caller promised one query per wavefront: verify against the first lane, fail loudly otherwise
        uint64_t const q0      = ((uint64_t)(uint32_t)__shfl((int)(q_off >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)q_off, 0);
        int const      l0      = __shfl(lq, 0);
        bool const     lead_in = __shfl(actA ? 1 : 0, 0) != 0;
        if (lead_in && ((actA && (q_off != q0 || lq != l0)) || (actB && (q_offB != q0 || lqB != l0))))
            atomicExch(p.err, 2);
    }

    int ls_max = max(lsA, lsB);
    int ls_min = min(actA ? lsA : 0x7fffffff, actB ? lsB : 0x7fffffff);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        ls_max = max(ls_max, __shfl_xor(ls_max, off));
        ls_min = min(ls_min, __shfl_xor(ls_min, off));
    }
    ls_max = __builtin_amdgcn_readfirstlane(ls_max);
    ls_min = __builtin_amdgcn_readfirstlane(ls_min);
    int const steps   = (ls_max + G - 1 + 3) & ~3;
    // one query per wavefront: the panel count is uniform
    int const npanels = MULTI ? max(1, __builtin_amdgcn_readfirstlane((lq + Geo::kPanel - 1) / Geo::kPanel)) : 1;

    // ---- range test (wave-uniform): an upper bound of every finite intermediate must stay below kI16Limit
    int bound = 0;
    for (int pn = 0; pn < npanels; ++pn)
    {
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            int const j = pn * Geo::kPanel + g * C + c;
            if (j < lq)
                bound += sc->rowmax[q[j] & (kAlph - 1)];
        }
    }
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1)
        bound += __shfl_xor(bound, off);
    bool const broken  = CKPT && ((__ballot(lq > (MULTI ? (int)p.panels_cap : 1) * Geo::kPanel) != 0) || (uint32_t)steps > p.steps_cap);
    // (upper end: no value reaches the non-finite patterns; lower end: pad scores and the skew of the first rows stay above 0)
    bool const too_big = broken || __ballot(bound + (-ge) * (steps + G + 2) + sc->smax + 2 > ((COMPACT && !MULTI) ? 2046 : kI16Limit)) != 0 ||
                         (-ge) * (G + 2) + (-sc->g2) + 256 > kBias;
    if (too_big)
    {
        // left to the int32 launch (TraceParams::fixup): sentinel -1; a broken length promise is reported there (and
        // here, because the compact sweep may run without that launch)
        if (broken && lane == 0)
            atomicExch(p.err, 3);
        if (is_first)
        {
            EndCell none{};
            none.score = -1;
            if (actA)
            {
                p.out_score[eA] = -1;
                if constexpr (CKPT)
                    p.ends[eA] = none;
            }
            if (actB)
            {
                p.out_score[eB] = -1;
                if constexpr (CKPT)
                    p.ends[eB] = none;
            }
        }
        return;
    }

    // carry workspace for multi-panel queries: one (packed H, packed E) pair per subject row and lane group
    uint32_t * carry   = nullptr;
    int const  ls_pair = max(lsA, lsB);
    if constexpr (MULTI)
    {
        if (npanels > 1)
        {
            uint32_t base = 0;
            int      ok   = 1;
            if (is_first && actA)
            {
                base = atomicAdd(p.ws_top, (uint32_t)ls_pair);
                if (base + (uint32_t)ls_pair > p.ws_cap)
                {
                    ok = 0;
                    atomicExch(p.err, 1);
                }
            }
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) // broadcast lane g == 0's values through the group
            {
                base = max(base, (uint32_t)__shfl_xor((int)base, off));
                ok   = min(ok, __shfl_xor(ok, off));
            }
            if (ok)
                carry = reinterpret_cast<uint32_t *>(p.ws) + 2ull * base;
        }
    }
    bool const writable = !MULTI || npanels == 1 || carry != nullptr; // (workspace exhausted: reported, nothing stored)

    constexpr uint32_t kRowBytes = Geo::kRowDw * 4;
    uint32_t const     row_base  = (uint32_t)(g * Geo::kLaneDw) * 4u;
    uint64_t const     panel_dw  = !CKPT ? 0 : COMPACT ? L16::slot_dwords(p.steps_cap) : Geo::slot_dwords(p.steps_cap);
    uint32_t * const   stage     = lds + nrows * Geo::kRowDw + lane; // [extension A / B][step % 4][lane]

    s2 const GE = ssplat(ge), G2 = ssplat(sc->g2), NGE = ssplat(-ge);
    // over the panels swept so far, per extension: best strip value, its (global) strip, first row, "met again later"
    int runA = 0, stripA = 0, rrowA = 0, rtieA = 0, runB = 0, stripB = 0, rrowB = 0, rtieB = 0;

    for (int panel = 0; panel < npanels; ++panel)
    {
        int const col0 = panel * Geo::kPanel + g * C;
        // ---- profile: prof[t][g][h] = (s(q_col, t) - ge) as int16, lane-contiguous; one query per wavefront
        if (grp == 0)
        {
#pragma unroll 1
            for (int d = 0; d < Geo::kUsedDw; ++d)
            {
                uint32_t rows[2][16];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
                {
                    int const c  = 2 * d + cc;
                    int const j  = col0 + c;
                    uint32_t  ql = kAlph - 1;
                    if (c < C && j < lq)
                        ql = q[j] & (kAlph - 1);
                    uint4 const * mrow = reinterpret_cast<uint4 const *>(sc->mat_i16 + ql * kAlph);
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                    {
                        uint4 const v       = mrow[x];
                        rows[cc][4 * x + 0] = v.x;
                        rows[cc][4 * x + 1] = v.y;
                        rows[cc][4 * x + 2] = v.z;
                        rows[cc][4 * x + 3] = v.w;
                    }
                }
                uint32_t * dst = lds + g * Geo::kLaneDw + d;
#pragma unroll
                for (int w = 0; w < 16; ++w)
                {
                    if (2 * w < nrows)
                    {
                        // letters t = 2w (low halves) and 2w+1 (high halves) of both columns
                        dst[(2 * w) * Geo::kRowDw]     = __builtin_amdgcn_perm(rows[1][w], rows[0][w], 0x05040100u);
                        dst[(2 * w + 1) * Geo::kRowDw] = __builtin_amdgcn_perm(rows[1][w], rows[0][w], 0x07060302u);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        bool const use_carry_in = MULTI && is_first && panel > 0 && carry != nullptr;
        bool const do_carry_out = MULTI && is_last && panel + 1 < npanels && carry != nullptr;
        // (compact sweep: an idle half owns the spare slot p.n -- its stores are unconditional)
        uint32_t * const slotA = !CKPT ? nullptr : p.ckpt + ((COMPACT && !actA) ? p.n : eA) * p.ckpt_stride + (uint64_t)panel * panel_dw;
        uint32_t * const slotB = !CKPT ? nullptr : p.ckpt + ((COMPACT && !actB) ? p.n : eB) * p.ckpt_stride + (uint64_t)panel * panel_dw;
        bool const       stA = CKPT && actA && writable, stB = CKPT && actB && writable;

        s2 Z = ssplat(ge * g + kBias); // z_i of the first processed row i = -g, biased
        s2 Hrow[C], F0[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            Hrow[c] = Z + GE;
            F0[c]   = Z;
        }
        s2 diag0 = Z + GE;
        s2 sendH = Z + GE;
        s2 sendE = as_s2(kI16NegInf2);
        s2 best  = ssplat(0);
        // first row that reached this strip's best value, "a later row reached it again" (bit 0 = A, bit 1 = B)
        int      rowA = 0, rowB = 0;
        uint32_t tie  = 0;
        s2       cmax = ssplat(0); // best un-skewed (and unbiased: >= 0) row maximum of the current chunk

        // one DP step at step index k = k0 + u (row k - g of this lane's strip)
        auto step = [&](uint32_t tA, uint32_t tB, int k, int u)
        {
            uint4 const * ra = reinterpret_cast<uint4 const *>(reinterpret_cast<char const *>(lds) + row_base + tA * kRowBytes);
            uint4 const * rb = reinterpret_cast<uint4 const *>(reinterpret_cast<char const *>(lds) + row_base + tB * kRowBytes);
            uint32_t      pa[Geo::kLaneDw], pb[Geo::kLaneDw];
#pragma unroll
            for (int x = 0; x < (Geo::kUsedDw + 3) / 4; ++x)
            {
                if (4 * x + 2 >= Geo::kUsedDw) // only two more dwords are needed: ds_read_b64
                {
                    uint2 const va = *reinterpret_cast<uint2 const *>(ra + x), vb = *reinterpret_cast<uint2 const *>(rb + x);
                    pa[4 * x] = va.x; pa[4 * x + 1] = va.y;
                    pb[4 * x] = vb.x; pb[4 * x + 1] = vb.y;
                }
                else
                {
                    uint4 const va = ra[x], vb = rb[x];
                    pa[4 * x] = va.x; pa[4 * x + 1] = va.y; pa[4 * x + 2] = va.z; pa[4 * x + 3] = va.w;
                    pb[4 * x] = vb.x; pb[4 * x + 1] = vb.y; pb[4 * x + 2] = vb.z; pb[4 * x + 3] = vb.w;
                }
            }

            // left boundary: H[i][-1] = 0 (skewed: z), E = -inf; or the previous panel's last column
            uint32_t bndH = as_u32(Z), bndE = kI16NegInf2;
            if constexpr (MULTI)
            {
                int const i = k - g;
                if (use_carry_in && (unsigned)i < (unsigned)ls_pair)
                {
                    bndH = carry[2 * i];
                    bndE = carry[2 * i + 1];
                }
            }
            s2 const recvH = as_s2((uint32_t)shift_from_left<G>((int)as_u32(sendH), (int)bndH, is_first));
            s2       Ecur  = as_s2((uint32_t)shift_from_left<G>((int)as_u32(sendE), (int)bndE, is_first));
            s2       dg    = diag0;
            diag0          = recvH;

            s2 const ZN     = Z + NGE;
            s2       rowmax = as_s2(kI16NegInf2);
            s2       h      = Z;
#pragma unroll
            for (int c = 0; c < C; ++c)
            {
                // (score of column c vs letter tA, score of column c vs letter tB)
                uint32_t const sel = (c & 1) ? 0x07060302u : 0x05040100u;
                s2 const       sub = as_s2(__builtin_amdgcn_perm(pb[c >> 1], pa[c >> 1], sel));
                s2 const       tt  = dg + sub;
                dg                 = Hrow[c];
                h                  = smax3(tt, Ecur, F0[c]);
                s2 const A         = h + G2;
                F0[c]              = smax3(F0[c], A, ZN);
                Ecur               = smax(Ecur, A) + GE;
                Hrow[c]            = h;
                rowmax             = smax(rowmax, h);
            }
            sendH = h;
            sendE = Ecur;
            if constexpr (MULTI)
            {
                int const i = k - g;
                if (do_carry_out && (unsigned)i < (unsigned)ls_pair)
                {
                    carry[2 * i]     = as_u32(sendH);
                    carry[2 * i + 1] = as_u32(sendE);
                }
            }
            if constexpr (CKPT)
            {
                cmax = smax(cmax, rowmax - Z); // the rose / met-again logic runs once per block of sixteen steps (book)
                // un-skewed boundary pairs (H of the strip's last column, E as the next strip's first column uses it),
                // re-paired per extension and staged for one 16-byte store per four steps
                if constexpr (COMPACT)
                    stage[((k & 4) + u) * 64] = (as_u32(h - Ecur) << 11) | as_u32(h - Z); // codes of both extensions: H | (H - E) << 11
                else
                {
                    s2 const hb = h - Z, eb = Ecur - Z;
                    stage[u * 64]       = __builtin_amdgcn_perm(as_u32(eb), as_u32(hb), 0x05040100u);
                    stage[(4 + u) * 64] = __builtin_amdgcn_perm(as_u32(eb), as_u32(hb), 0x07060302u);
                }
            }
            else
                best = smax(best, rowmax - Z);
            Z = ZN;
        };
        // compact sweep: the staged codes of the eight steps up to k0 + 3 leave, re-paired per extension (whole 128-byte
        // lines per lane group, no branch around the stores); the row checkpoint behind step k0 + 3 (k0 % 16 == 12)
        auto flush_codes = [&](int k0)
        {
            if constexpr (COMPACT)
            {
                uint32_t cw[8];
#pragma unroll
                for (int x = 0; x < 8; ++x)
                    cw[x] = stage[x * 64];
                uint32_t const oi = L16::bnd_oct_index((uint32_t)k0 / 8, (uint32_t)g);
                reinterpret_cast<uint4 *>(slotA)[oi] = make_uint4(__builtin_amdgcn_perm(cw[1], cw[0], 0x05040100u), __builtin_amdgcn_perm(cw[3], cw[2], 0x05040100u),
                                                                  __builtin_amdgcn_perm(cw[5], cw[4], 0x05040100u), __builtin_amdgcn_perm(cw[7], cw[6], 0x05040100u));
                reinterpret_cast<uint4 *>(slotB)[oi] = make_uint4(__builtin_amdgcn_perm(cw[1], cw[0], 0x07060302u), __builtin_amdgcn_perm(cw[3], cw[2], 0x07060302u),
                                                                  __builtin_amdgcn_perm(cw[5], cw[4], 0x07060302u), __builtin_amdgcn_perm(cw[7], cw[6], 0x07060302u));
            }
        };
        auto rowck_codes = [&](int k0)
        {
            if constexpr (COMPACT)
            {
                // Hrow is in the frame of the row just processed (z_i = Z + ge after the update), F0 in the next row's:
                // H - F un-skewed = (Hrow - z_i) - (F0 - Z) = Hrow - F0 - ge
                s2 const zi = Z + GE;
                uint32_t code[2 * L16::kCkDw];
#pragma unroll
                for (int c = 0; c < 2 * L16::kCkDw; ++c)
                    code[c] = c < C ? ((as_u32((Hrow[c < C ? c : 0] - F0[c < C ? c : 0]) - GE) << 11) | as_u32(Hrow[c < C ? c : 0] - zi)) : 0u;
                uint32_t const base = (uint32_t)(L16::bnd_dwords(p.steps_cap) / 4);
                uint4 * const  dA = reinterpret_cast<uint4 *>(slotA) + base, * const dB = reinterpret_cast<uint4 *>(slotB) + base;
#pragma unroll
                for (int x = 0; x < L16::kCkDw / 4; ++x)
                {
                    uint32_t wa[4], wb[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                    {
                        int const c = 2 * (4 * x + b); // columns c, c + 1 of extension A (low halves) / B (high halves)
                        wa[b]       = __builtin_amdgcn_perm(code[c + 1], code[c], 0x05040100u);
                        wb[b]       = __builtin_amdgcn_perm(code[c + 1], code[c], 0x07060302u);
                    }
                    uint32_t const qi = L16::rowck_quad_index((uint32_t)(k0 + 3) / 16, (uint32_t)g, (uint32_t)x);
                    dA[qi] = make_uint4(wa[0], wa[1], wa[2], wa[3]);
                    dB[qi] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
                }
            }
        };
        // the best-value bookkeeping, once per block of sixteen steps (lx_score_f16.hip: book)
        auto book = [&](int k0) // k0: the first of the block's last four steps
        {
            if constexpr (!CKPT)
                return;
            // per half: did the strip's best rise in this block (then its first row is one of the block's sixteen; the
            // backtrace finds it), or was it only met again (a tie for the end cell)?  Rows beyond the window and columns
            // beyond the query stay strictly below a positive best: no validity test.
            s2 const       nb   = smax(best, cmax);
            uint32_t const rose = as_u32(nb) ^ as_u32(best), met = as_u32(cmax) ^ as_u32(best);
            bool const     gtA = (rose & 0xffffu) != 0, gtB = (rose >> 16) != 0;
            bool const     eqA = (met & 0xffffu) == 0, eqB = (met >> 16) == 0;
            int const      last = k0 + 3 - g; // the block's last row in this lane
            rowA = gtA ? last : rowA;
            rowB = gtB ? last : rowB;
            tie  = (gtA ? (tie & ~1u) : (tie | (eqA ? 1u : 0u)));
            tie  = (gtB ? (tie & ~2u) : (tie | (eqB ? 2u : 0u)));
            best = nb;
            cmax = ssplat(0);
        };
        // after every fourth step the staged boundary quads leave; every 16th step the row checkpoint follows
        auto chunk_done = [&](int k0)
        {
            if constexpr (!CKPT)
                return;
            if (((k0 + 4) & 15) == 0)
                book(k0);
            if constexpr (COMPACT)
            {
                if (k0 & 4)
                    flush_codes(k0);
                return; // (the row checkpoint follows at the top of the next chunk: rowck_codes)
            }
            uint32_t const qi = ((uint32_t)k0 / 4) * G + (uint32_t)g; // lx_ckpt.hip: bnd_quad_index
            if (stA)
                reinterpret_cast<uint4 *>(slotA)[qi] = make_uint4(stage[0], stage[64], stage[128], stage[192]);
            if (stB)
                reinterpret_cast<uint4 *>(slotB)[qi] = make_uint4(stage[256], stage[320], stage[384], stage[448]);
            if (((k0 + 3) & 15) == 15)
            {
                // Hrow is in the frame of the row just processed (z_i = Z + ge after the update), F0 in the next row's
                s2 const zi = Z + GE;
                // lx_ckpt.hip: rowck_quad_index (lane-major) behind the boundary quads
                uint64_t const base = Geo::bnd_dwords(p.steps_cap) / 4 + ((uint64_t)((k0 + 3) / 16) * G + (uint64_t)g) * (Geo::kCkDw / 4);
                uint4 * const  dA = reinterpret_cast<uint4 *>(slotA) + base, * const dB = reinterpret_cast<uint4 *>(slotB) + base;
#pragma unroll
                for (int x = 0; x < Geo::kCkDw / 4; ++x)
                {
                    uint32_t wa[4], wb[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                    {
                        int const c = 4 * x + b;
                        if (c < C)
                        {
                            s2 const hu = Hrow[c < C ? c : 0] - zi, fu = F0[c < C ? c : 0] - Z;
                            wa[b] = __builtin_amdgcn_perm(as_u32(fu), as_u32(hu), 0x05040100u);
                            wb[b] = __builtin_amdgcn_perm(as_u32(fu), as_u32(hu), 0x07060302u);
                        }
                        else
                            wa[b] = wb[b] = 0;
                    }
                    if (stA)
                        dA[x] = make_uint4(wa[0], wa[1], wa[2], wa[3]);
                    if (stB)
                        dB[x] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
                }
            }
        };

        uint32_t const lscA = (uint32_t)max(lsA, 1) - 1u, lscB = (uint32_t)max(lsB, 1) - 1u;
        auto fetch_checked = [&](int k0, uint32_t (&ta)[4], uint32_t (&tb)[4])
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                uint32_t const i = (uint32_t)(k0 + u - g);
                ta[u]            = sA[min(i, lscA)];
                tb[u]            = sB[min(i, lscB)];
            }
        };
        auto mask_checked = [&](int k0, uint32_t (&ta)[4], uint32_t (&tb)[4])
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                uint32_t const i = (uint32_t)(k0 + u - g);
                ta[u]            = (i < (uint32_t)lsA) ? (ta[u] & (kAlph - 1)) : padt;
                tb[u]            = (i < (uint32_t)lsB) ? (tb[u] & (kAlph - 1)) : padt;
            }
        };

        int const steady_lo = (G - 1 + 3) & ~3;
        int const steady_hi = ls_min - 3;
        uint8_t const * spA = sA - g;
        uint8_t const * spB = sB - g;

        int      k0 = 0;
        uint32_t na[4], nb[4];
        fetch_checked(0, na, nb);
        while (k0 < steps)
        {
            bool const cur_steady = (k0 >= steady_lo) && (k0 < steady_hi);
            if (!cur_steady)
            {
                if (COMPACT && k0 != 0 && (k0 & 15) == 0)
                    rowck_codes(k0 - 4);
                uint32_t ca[4] = {na[0], na[1], na[2], na[3]}, cb[4] = {nb[0], nb[1], nb[2], nb[3]};
                mask_checked(k0, ca, cb);
                fetch_checked(k0 + 4, na, nb);
#pragma unroll 1
                for (int u = 0; u < 4; ++u)
                    step(ca[u], cb[u], k0 + u, u);
                chunk_done(k0);
                k0 += 4;
            }
            else
            {
                uint32_t wa = *reinterpret_cast<unaligned_u32 const *>(spA + k0);
                uint32_t wb = *reinterpret_cast<unaligned_u32 const *>(spB + k0);
                while (k0 < steady_hi)
                {
                    if (COMPACT && (k0 & 15) == 0) // (k0 >= steady_lo > 0)
                        rowck_codes(k0 - 4);
                    uint32_t const ca = wa, cb = wb;
                    int const      kn = max(min(k0 + 4, ls_min - 4), 0);
                    wa                = *reinterpret_cast<unaligned_u32 const *>(spA + kn);
                    wb                = *reinterpret_cast<unaligned_u32 const *>(spB + kn);
LX_I16_UNROLL_N(LX_I16_UNROLL)
                    for (int u = 0; u < 4; ++u)
                        step((ca >> (8 * u)) & (kAlph - 1), (cb >> (8 * u)) & (kAlph - 1), k0 + u, u);
                    chunk_done(k0);
                    k0 += 4;
                }
                fetch_checked(k0, na, nb);
            }
        }

        if constexpr (COMPACT)
        {
            if (steps != 0 && (steps & 15) == 0)
                rowck_codes(steps - 4);
            if (steps & 4)
                flush_codes(steps); // the last four steps' codes (the other half of the group is stale: beyond every row)
        }
        if (steps & 15)
            book(steps - 4); // the last, partial block
        // per extension: best strip value over the group; among equal ones the lowest strip (its columns come first).
        // Over the panels a later one only wins with a strictly greater value (its columns come later).
        auto merge = [&](int lbest, int lrow, int ltie, int & run, int & rstrip, int & rrow, int & rtie)
        {
            int gbest = lbest, gstrip = g, grow = lrow, gtie = ltie;
#pragma unroll
            for (int off = 1; off < G; off <<= 1)
            {
                int const  ob = __shfl_xor(gbest, off), os = __shfl_xor(gstrip, off), orow = __shfl_xor(grow, off), ot = __shfl_xor(gtie, off);
                bool const take = ob > gbest || (ob == gbest && os < gstrip);
                gbest  = take ? ob : gbest;
                gstrip = take ? os : gstrip;
                grow   = take ? orow : grow;
                gtie   = take ? ot : gtie;
            }
            bool const take = gbest > run;
            run    = take ? gbest : run;
            rstrip = take ? panel * G + gstrip : rstrip;
            rrow   = take ? grow : rrow;
            rtie   = take ? gtie : rtie;
        };
        if constexpr (CKPT)
        {
            merge((int)best.x, rowA, (int)(tie & 1u), runA, stripA, rrowA, rtieA);
            merge((int)best.y, rowB, (int)((tie >> 1) & 1u), runB, stripB, rrowB, rtieB);
        }
        else
        {
            runA = max(runA, (int)best.x); // per lane; reduced over the group once all panels are through
            runB = max(runB, (int)best.y);
        }

        if constexpr (MULTI)
        {
            if (npanels > 1)
            {
                // make this panel's carry stores visible to the next panel's loads (same wave, other lanes)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
        __builtin_amdgcn_wave_barrier();
    } // panels

    auto finish = [&](int run, int rstrip, int rrow, int rtie, bool act, uint64_t e)
    {
        if (is_first && act)
        {
            EndCell ec{};
            bool const declined = COMPACT && MULTI && run > 2046; // beyond what the codes hold: the int32 launch redoes it
            if (!writable || declined)
                ec.score = -1;
            else if (run > 0)
            {
                ec.score = run;
                ec.q_end = -(rstrip + 1); // the backtrace finds the column inside this strip
                ec.s_end = rrow + 1;
                ec.flags = (rtie ? kEndAmbiguous : 0) | (COMPACT ? kEndCompact : 0);
            }
            p.ends[e]      = ec;
            p.out_score[e] = (writable && !declined) ? run : -1;
        }
    };
    if constexpr (CKPT)
    {
        finish(runA, stripA, rrowA, rtieA, actA, eA);
        finish(runB, stripB, rrowB, rtieB, actB, eB);
    }
    else
    {
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1)
        {
            runA = max(runA, __shfl_xor(runA, off));
            runB = max(runB, __shfl_xor(runB, off));
        }
        if (is_first && actA)
            p.out_score[eA] = writable ? runA : -1;
        if (is_first && actB)
            p.out_score[eB] = writable ? runB : -1;
    }
}

template <int G, int C>
static hipError_t launch_sweep16_cfg(ScoreParams const & p, hipStream_t stream)
{
    using Geo = Pair16Geo<G, C>;
    uint64_t const per_wave = 2ull * Geo::kGroups;
    uint64_t const blocks   = (p.n + per_wave - 1) / per_wave;
    if (blocks > 0x7fffffffull || !p.ckpt || !p.ends || p.steps_cap % 16 != 0)
        return hipErrorInvalidValue;
    size_t const lds = ((size_t)p.nrows * Geo::kRowDw + 64 * 8) * sizeof(uint32_t);
    if (p.panels_cap > 1)
        hipLaunchKernelGGL((sweep_pair16_kernel<G, C, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    else
        hipLaunchKernelGGL((sweep_pair16_kernel<G, C, false>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}

// score only, any query width, (8,19) panels: 16 extensions of one query per wavefront
hipError_t launch_score_pair16(ScoreParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    using Geo = Pair16Geo<8, 19>;
    uint64_t const per_wave = 2ull * Geo::kGroups;
    uint64_t const blocks   = (p.n + per_wave - 1) / per_wave;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    size_t const lds = ((size_t)p.nrows * Geo::kRowDw + 64 * 8) * sizeof(uint32_t);
    hipLaunchKernelGGL((sweep_pair16_kernel<8, 19, true, false>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}

// the compact-slot sweep in the integer domain (same slots and conditions as score_pair_kernel<G,C,CKPT=true>)
template <int G, int C>
static hipError_t launch_sweep16_compact_cfg(ScoreParams const & p, hipStream_t stream)
{
    using Geo = Pair16Geo<G, C>;
    uint64_t const per_wave = 2ull * Geo::kGroups;
    uint64_t const blocks   = (p.n + per_wave - 1) / per_wave;
    if (blocks > 0x7fffffffull || !p.ckpt || !p.ends || p.steps_cap % 16 != 0)
        return hipErrorInvalidValue;
    size_t const lds = ((size_t)p.nrows * Geo::kRowDw + 64 * 8) * sizeof(uint32_t);
    hipLaunchKernelGGL((sweep_pair16_kernel<G, C, false, true, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}
hipError_t launch_sweep_pair16_compact(int trace_cfg, ScoreParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    if (p.panels_cap > 1) // queries wider than a panel: (8,19) panels, one part of compact codes each
    {
        using Geo = Pair16Geo<8, 19>;
        uint64_t const blocks = (p.n + 2ull * Geo::kGroups - 1) / (2ull * Geo::kGroups);
        if (trace_cfg != 1 || blocks > 0x7fffffffull || !p.ckpt || !p.ends || p.steps_cap % 16 != 0)
            return hipErrorInvalidValue;
        size_t const lds = ((size_t)p.nrows * Geo::kRowDw + 64 * 8) * sizeof(uint32_t);
        hipLaunchKernelGGL((sweep_pair16_kernel<8, 19, true, true, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
        return hipGetLastError();
    }
    return trace_cfg == 1 ? launch_sweep16_compact_cfg<8, 19>(p, stream) : trace_cfg == 2 ? launch_sweep16_compact_cfg<16, 13>(p, stream) : hipErrorInvalidValue;
}

// trace cfg 1 = (8,19): 16 extensions of one query per wavefront; 2 = (16,13): 8
hipError_t launch_sweep_pair16(int trace_cfg, ScoreParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    return trace_cfg == 1 ? launch_sweep16_cfg<8, 19>(p, stream) : trace_cfg == 2 ? launch_sweep16_cfg<16, 13>(p, stream) : hipErrorInvalidValue;
}

} // namespace lx

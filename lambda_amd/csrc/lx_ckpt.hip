// lx_ckpt.hip -- pass 2 by checkpoints: an untagged forward DP that stores strip boundaries and periodic row
// checkpoints, and a backtrace that recomputes one small tile at a time (gfx950 only).
//
// Same reference seam as lx_trace.hip (_performAlignment<withTrace=true>, /root/reference/src/search_algo.hpp:1296 ->
// :1070-1134, TracebackOn<CompleteTrace, GapsLeft>, :1083; _adaptTraceSegmentsTo :1127; computeAlignmentStats :1308).
// lx_trace.hip pays 18 of its 40 clocks per cell for tags and direction bits of cells no walk ever visits (42 G cells
// computed, 0.27 G visited on the headline batch).  Here the forward kernel runs the plain row-skewed recurrence of
// lx_score.hip (int32, 22 clocks per cell) and keeps just enough to restart the DP anywhere:
//   * boundary array: for every strip (= lane g, C query columns) and every row i the pair
//         H(i, last column of the strip),  E(i, first column of the next strip)     as two int16, un-skewed;
//     a lane collects four steps in registers and stores one 16-byte quad -- the G lanes of a group write G
//     consecutive quads, no LDS staging;
//   * row checkpoints: every 16 STEPS each lane stores H(i, c) and F(i+1, c) of its C columns (int16 pairs).  Step
//     k is row k - g for lane g, so every strip has its own row grid and no lane-divergent branch is needed.
// The end cell is found through the known best score exactly as in lx_trace.hip.  The backtrace kernel (one lane per
// extension) recomputes the 16 x C tile around its current cell from the checkpoint above it and the boundary array of
// the strip to its left -- the same recurrence including the floor folded into F, so every value and every tie of
// the forward pass is reproduced -- keeps the tile's direction nibbles in LDS and walks them with the rules of
// lx_trace.hip's backtrace until it leaves the tile.
//
// Limits (checked by the host, which otherwise uses the direction-bit path): query fits one panel, scores < 32768.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "lx_aids.h"
#include "lx_dp_common.h"

namespace lx
{

#ifndef LX_CKPT_EVERY
#define LX_CKPT_EVERY 16
#endif
// steps between two row checkpoints (multiple of 4).  Headline batch, forward + backtrace: 8 -> 12.8 + 6.0 ms,
// 16 -> 11.2 + 6.5 ms, 32 -> 10.7 + 9.7 ms
constexpr int kCkptEvery = LX_CKPT_EVERY;
#ifndef LX_CKPT_UNROLL
#define LX_CKPT_UNROLL 4 // steps unrolled in the forward kernel (11.9 / 11.5 / 11.3 ms for 1, 2, 4 on the headline batch)
#endif
#define LX_CKPT_PRAGMA(x) _Pragma(#x)
#define LX_CKPT_UNROLL_N(n) LX_CKPT_PRAGMA(unroll n)
#define LX_CKPT_UNROLL_PRAGMA LX_CKPT_UNROLL_N(LX_CKPT_UNROLL)
#ifndef LX_BT_TILE_AT
#define LX_BT_TILE_AT 40   // backtrace: lanes waiting for a tile that trigger a tile phase
#endif
#ifndef LX_BT_WAVES
#define LX_BT_WAVES 2      // backtrace: wavefronts per SIMD the register budget is set for (2: 256 VGPRs, no spills)
#endif
#ifndef LX_BT_HOPS
#define LX_BT_HOPS 2       // backtrace: tile borders a diagonal shortcut pass may cross
#endif
#ifndef LX_BT_REFILL_AT
#define LX_BT_REFILL_AT 12 // backtrace: finished / empty lanes that trigger a refill outside a tile phase
#endif
#ifndef LX_CKPT_FWD_WAVES
#define LX_CKPT_FWD_WAVES 4
#endif

// slot layout in uint32 units: boundary quads (4 steps = 4 dwords each) [step / 4][lane] -- lane-minor: the G lanes of a
// group write G consecutive 16-byte quads per store instruction, i.e. whole cache lines, every four steps --, then row
// checkpoints [checkpoint][lane][quad of 4 columns] -- lane-major: written only every 16 steps, read by the backtrace as
// one contiguous piece.  Measured on the headline batch (packed-half sweep + backtrace): boundary quads grouped per
// lane 18.3 + 7.1 ms, everything lane-minor 13.9 + 8.3 ms, this mix 14.3 + 7.2 ms.
template <int G>
__host__ __device__ constexpr uint32_t bnd_quad_index(uint32_t quad, uint32_t g)
{
    return quad * G + g;
}
// uint4 index of quad x of lane g's row checkpoint m
#ifndef LX_CKPT_ROWCK_LANE_MAJOR
#define LX_CKPT_ROWCK_LANE_MAJOR 1
#endif
template <int G, int kCkDw>
__host__ __device__ constexpr uint32_t rowck_quad_index(uint32_t m, uint32_t g, uint32_t x)
{
    return LX_CKPT_ROWCK_LANE_MAJOR ? (m * G + g) * (kCkDw / 4) + x : (m * (kCkDw / 4) + x) * G + g;
}
template <int G, int C>
struct CkptLayout
{
    static constexpr int kCkDw = (C + 3) / 4 * 4; // dwords per lane per row checkpoint, whole quads
    __host__ __device__ static constexpr uint64_t bnd_dwords(uint32_t steps_cap) { return (uint64_t)steps_cap * G; }
    __host__ __device__ static constexpr uint64_t slot_dwords(uint32_t steps_cap)
    {
        return bnd_dwords(steps_cap) + (uint64_t)(steps_cap / kCkptEvery) * G * kCkDw;
    }
};

__device__ __forceinline__ uint32_t pack16(int lo, int hi)
{
    return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
}

// KNOWN = the best score of every extension is given (p.score_in): the end cell is the first cell in column-major order
// that reaches it.  !KNOWN = single sweep over all extensions: the kernel also IS pass 1 -- every lane keeps the best
// value of its strip, the first row that reached it and whether a later row reached it again; the group then reports
// score, strip and row, and the backtrace resolves the column (and, if the strip tied, the row) from the checkpoints.
// MULTI = queries wider than one panel: the panels of an extension are swept one after the other, each into its own
// part of the slot ([panel][boundary quads + row checkpoints]); the last strip's (H, E) per row reaches the next
// panel's first strip through the carry workspace of lx_score.hip / lx_trace.hip.
template <int G, int C, bool KNOWN, bool MULTI>
__global__ __launch_bounds__(64, (KNOWN ? LX_CKPT_FWD_WAVES : 3)) void ckpt_forward_kernel(TraceParams p)
{
    using Geo = ScoreGeo<G, C>;
    using Lay = CkptLayout<G, C>;
    extern __shared__ uint32_t lds[];

    int const  lane     = threadIdx.x;
    int const  grp      = lane / G;
    int const  g        = lane % G;
    bool const is_first = (g == 0);
    bool const is_last  = (g == G - 1);

    uint64_t const e     = (uint64_t)blockIdx.x * Geo::kGroups + grp;
    uint64_t       limit = p.n;
    if (p.count_ptr)
    {
        uint64_t const total = *p.count_ptr;
        limit = total > p.chunk_start ? min(p.n, total - p.chunk_start) : 0;
        if ((uint64_t)blockIdx.x * Geo::kGroups >= limit)
            return; // whole wavefront beyond the device-side survivor count
    }
    bool active = e < limit;
    if (active && p.src && p.src[e] == 0xffffffffu)
        active = false; // padding slot (keeps one query per sharing block)
    bool in_list = e < limit;
    if constexpr (!KNOWN)
    {
        if (p.fixup)
        {
            // second launch of the single sweep: only the extensions the packed-half kernel declined (sentinel -1)
            bool const mine = in_list && p.score_out[e] == -1;
            if (__ballot(mine) == 0)
                return;
            active  = active && mine;
            in_list = mine;
        }
    }

    ScoringDev const * __restrict__ sc = p.sc;
    int const      ge    = sc->ge;
    int const      g2    = sc->g2;
    int const      nrows = p.nrows;
    uint32_t const padt  = (uint32_t)(nrows - 1);

    int             lq = 0, ls = 0;
    uint8_t const * q = p.q_res;
    uint8_t const * s = p.s_res;
    uint64_t        q_off = 0;
    if (e < limit)
    {
        Extension const x = p.ext[e];
        lq    = (int)x.q_len;
        q_off = x.q_off;
        q += x.q_off;
        if (active)
        {
            ls = (int)x.s_len;
            if (ls != 0)
                s += x.s_off;
        }
    }
    // groups that share one LDS profile (see lx_trace.hip): verified against the first lane of the sharing block
    int const share  = p.shared_profile > 1 ? min(p.shared_profile, Geo::kGroups) : 1;
    int const leader = (grp / share) * share * G;
    if (share > 1)
    {
        uint32_t const qlo = (uint32_t)__shfl((int)(uint32_t)q_off, leader), qhi = (uint32_t)__shfl((int)(q_off >> 32), leader);
        int const      l0  = __shfl(lq, leader);
        bool const     lead_in = __shfl(in_list ? 1 : 0, leader) != 0;
        if (in_list && lead_in && (q_off != (((uint64_t)qhi << 32) | qlo) || lq != l0))
            atomicExch(p.err, 2);
    }

    int       ls_max    = ls;
    int       npanels   = (lq + Geo::kPanel - 1) / Geo::kPanel;
    int const my_panels = npanels; // of this group's query (uniform over the group)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        ls_max = max(ls_max, __shfl_xor(ls_max, off));
        if constexpr (MULTI)
            npanels = max(npanels, __shfl_xor(npanels, off));
    }
    ls_max = __builtin_amdgcn_readfirstlane(ls_max);
    if constexpr (MULTI)
        npanels = max(1, __builtin_amdgcn_readfirstlane(npanels));
    else
        npanels = 1; // the host sends wider queries to the MULTI instantiation

    bool bad = false;
    if (active && (lq > (MULTI ? (int)p.panels_cap : 1) * Geo::kPanel || (uint32_t)((ls + G - 1 + 3) & ~3) > p.steps_cap || ls > 65535))
    {
        bad = true; // the host sized the slots too small for this extension: report, never write out of bounds
        atomicExch(p.err, 3);
    }
    if (bad)
        ls = 0;

    // carry workspace for multi-panel queries: Ls pairs (Hs, Es) per extension (lx_score.hip)
    int32_t * carry = nullptr;
    if constexpr (MULTI)
    {
        if (npanels > 1)
        {
            uint32_t base = 0;
            int      ok   = 1;
            if (is_first && my_panels > 1 && active && !bad)
            {
                base = atomicAdd(p.ws_top, (uint32_t)ls);
                if (base + (uint32_t)ls > p.ws_cap)
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
            if (my_panels > 1 && ok)
                carry = p.ws + 2ull * base;
            else if (my_panels > 1)
            {
                bad = true; // workspace exhausted: neutralised, reported through p.err
                ls  = 0;
            }
        }
    }

    int const      slot_dw     = (grp / share) * (nrows * Geo::kRowDw);
    uint32_t const row_base_dw = (uint32_t)(slot_dw + g);
    int const      steps       = (ls_max + G - 1 + 3) & ~3;
    uint32_t const lsc         = (uint32_t)max(ls, 1) - 1u;

    uint32_t * slot = p.trace + e * p.slot_stride;
    uint32_t   ovf_tag = 0; // 1 + index of this extension's slot in the overflow area
    if constexpr (!KNOWN)
    {
        if (p.fixup && p.ovf_count)
        {
            // the batch's slots are the compact ones of the packed-half kernel: what it declined gets an int16-pair slot
            // of the overflow area (handed out per extension; the backtrace finds it through the end cell's flags)
            uint32_t idx = 0;
            if (is_first && active && !bad)
                idx = atomicAdd(p.ovf_count, 1u);
            idx = (uint32_t)__shfl((int)idx, grp * G);
            if (active && !bad && idx >= p.ovf_cap)
            {
                bad = true; // no room left: reported, never written out of bounds
                ls  = 0;
                atomicExch(p.err, 4);
            }
            ovf_tag = idx + 1;
            slot    = p.ovf + (uint64_t)(bad ? 0 : idx) * p.ovf_stride;
        }
    }
    // KNOWN: target score and the best (lowest) column / its first row seen so far in this lane
    int tgt = 0;
    if constexpr (KNOWN)
        tgt = active ? p.score_in[e] : 0;
    int kcol = 0x7fffffff, krow = 0;
    // !KNOWN, over the panels swept so far: best strip value, its (global) strip, first row, "reached again later"
    int run_best = 0, run_strip = 0, run_row = 0, run_tie = 0;
    uint64_t const panel_dw = Lay::slot_dwords(p.steps_cap);

    for (int panel = 0; panel < npanels; ++panel)
    {
    bool const    in_panel = panel < my_panels; // (a wavefront may hold queries with fewer panels than its widest)
    bool const    store_ok = active && !bad && in_panel;
    uint32_t * const pslot = slot + (uint64_t)(MULTI ? panel : 0) * panel_dw;
    uint4 * const rowck    = reinterpret_cast<uint4 *>(pslot + Lay::bnd_dwords(p.steps_cap));
    // !KNOWN: best value of this lane's strip, first row that reached it, "reached again later"
    int lbest = 0, lrow = 0, ltie = 0;
    bool use_carry_in = false, do_carry_out = false;
    if constexpr (MULTI)
    {
        use_carry_in = is_first && panel > 0 && in_panel && carry != nullptr;
        do_carry_out = is_last && panel + 1 < my_panels && carry != nullptr;
    }

    int const col0 = (MULTI ? panel * Geo::kPanel : 0) + g * C;
    build_profile<G, C>(lds, slot_dw, g, q, lq, col0, sc->mat_adj, nrows, grp % share == 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // state as of (virtual) row i = -g - 1: every cell is "H = 0" (lx_score.hip)
    int z = ge * g;
    int Hrow[C], F0[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
    {
        Hrow[c] = z + ge;
        F0[c]   = z;
    }
    int diag0 = z + ge;
    int sendH = z + ge;
    int sendE = kNegInf;
    int g2v = g2, gev = ge; // loop constants in VGPRs: an SGPR operand halves the issue rate (tools/ubench.hip)
    LX_OPAQUE(g2v);
    LX_OPAQUE(gev);

    // boundary words are staged in a lane-private 16-byte LDS slot and leave as one quad every four steps
    uint32_t * const stage = lds + ((Geo::kGroups + share - 1) / share) * (nrows * Geo::kRowDw) + lane * 4;
    uint4 * const    bnd   = reinterpret_cast<uint4 *>(pslot);

    // one DP step; stores the boundary word of this lane's strip for row k - g
    auto step = [&](int k, uint32_t t)
    {
        int const        i    = k - g;
        uint32_t const * prow = lds + (row_base_dw + t * (uint32_t)Geo::kRowDw);
        uint32_t         pw[Geo::kD];
#pragma unroll
        for (int d = 0; d < Geo::kD; ++d)
            pw[d] = prow[d * G];

        // left boundary of this strip for row i: H = 0 (skewed: z), E = -inf; or the previous panel's last column
        int bndH = z, bndE = kNegInf;
        if constexpr (MULTI)
        {
            if (use_carry_in && (unsigned)i < (unsigned)ls)
            {
                bndH = carry[2 * i];
                bndE = carry[2 * i + 1];
            }
        }
        int const recvH = shift_from_left<G>(sendH, bndH, is_first);
        int       Ecur  = shift_from_left<G>(sendE, bndE, is_first);
        int       dg    = diag0;
        diag0           = recvH;

        int const zn     = z - gev;
        int       rowmax = z;
        int       h      = 0;
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            int const sub = (int)(int8_t)(pw[c >> 2] >> (8 * (c & 3)));
            int const tt  = dg + sub;
            dg            = Hrow[c];
            h             = max3i(tt, Ecur, F0[c]);
            LX_OPAQUE(h);
            int const A   = h + g2v;
            F0[c]         = max3i(F0[c], A, zn);
            LX_OPAQUE(F0[c]);
            Ecur          = max(Ecur, A) + gev;
            if (c & 1)
                rowmax = max3i(rowmax, Hrow[c - 1], h); // Hrow[c-1] already holds this row's value
            else if (c == C - 1)
                rowmax = max(rowmax, h);
            Hrow[c] = h;
        }
        if constexpr (KNOWN)
        {
            // rare: some cell of this row reaches the extension's best score -> remember the lowest such column (rows
            // are visited in increasing order, so the first hit of a column is its lowest row)
            if (tgt > 0 && rowmax == tgt + z && (unsigned)i < (unsigned)ls)
            {
#pragma unroll
                for (int c = C - 1; c >= 0; --c)
                    if (Hrow[c] == rowmax && col0 + c < kcol)
                    {
                        kcol = col0 + c;
                        krow = i;
                    }
            }
        }
        else
        {
            // rows beyond the window and columns beyond the query score strictly below the best real cell, so they
            // can neither raise nor tie a positive maximum: no validity test needed
            int const  cand = rowmax - z;
            bool const gt   = cand > lbest;
            ltie            = gt ? 0 : (ltie | (cand == lbest ? 1 : 0));
            lrow            = gt ? i : lrow;
            lbest           = max(lbest, cand);
        }
        sendH = h;
        sendE = Ecur;
        if constexpr (MULTI)
        {
            if (do_carry_out && (unsigned)i < (unsigned)ls)
            {
                carry[2 * i]     = sendH;
                carry[2 * i + 1] = sendE;
            }
        }
        // un-skewed boundary pair: H of the strip's last column, E as the next strip's first column will use it
        stage[k & 3] = pack16(h - z, max(Ecur - z, -32768));
        z            = zn;
    };
    // row checkpoint after step k: H(i, c) and the folded F(i+1, c), un-skewed (z already is z_{i+1} here)
    auto checkpoint = [&](int k)
    {
        if (!store_ok)
            return;
#pragma unroll
        for (int x = 0; x < Lay::kCkDw / 4; ++x)
        {
            uint32_t w[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
            {
                int const c = 4 * x + b;
                w[b]        = c < C ? pack16(Hrow[c < C ? c : 0] - (z + gev), F0[c < C ? c : 0] - z) : 0u;
            }
            rowck[rowck_quad_index<G, Lay::kCkDw>((uint32_t)(k / kCkptEvery), (uint32_t)g, (uint32_t)x)] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    };

    auto fetch_checked = [&](int k0, uint32_t (&t)[4])
    {
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            uint32_t const i   = (uint32_t)(k0 + u - g);
            uint32_t const idx = min(i, lsc);
            t[u]               = s[idx];
        }
    };
    auto mask_checked = [&](int k0, uint32_t (&t)[4])
    {
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            uint32_t const i = (uint32_t)(k0 + u - g);
            t[u]             = (i < (uint32_t)ls) ? (t[u] & (kAlph - 1)) : padt;
        }
    };

    uint32_t tn[4];
    fetch_checked(0, tn);
    for (int k0 = 0; k0 < steps; k0 += 4)
    {
        uint32_t tc[4] = {tn[0], tn[1], tn[2], tn[3]};
        mask_checked(k0, tc);
        fetch_checked(k0 + 4, tn);
        uint32_t const tcp = tc[0] | (tc[1] << 8) | (tc[2] << 16) | (tc[3] << 24);
LX_CKPT_UNROLL_PRAGMA
        for (int u = 0; u < 4; ++u)
            step(k0 + u, (tcp >> (8 * u)) & 0xffu);
        if (store_ok)
            bnd[bnd_quad_index<G>((uint32_t)k0 / 4, (uint32_t)g)] = *reinterpret_cast<uint4 const *>(stage);
        if (((k0 + 3) % kCkptEvery) == kCkptEvery - 1)
            checkpoint(k0 + 3);
    }

    if constexpr (!KNOWN)
    {
        // best strip value over the group; among equal ones the lowest strip (its columns come first).  Over the panels:
        // a later panel only wins with a strictly greater value (its columns come later).
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
        bool const take = gbest > run_best;
        run_best  = take ? gbest : run_best;
        run_strip = take ? (MULTI ? panel * G : 0) + gstrip : run_strip;
        run_row   = take ? grow : run_row;
        run_tie   = take ? gtie : run_tie;
    }
    if constexpr (MULTI)
    {
        if (npanels > 1)
        {
            // make this panel's carry stores visible to the next panel's loads (same wave, other lanes); the LDS profile
            // is rebuilt next
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    __builtin_amdgcn_wave_barrier();
    } // panels

    if constexpr (KNOWN)
    {
        // lowest column over the lanes of the group (every lane owns different columns in every panel), with its row
#pragma unroll
        for (int off = 1; off < G; off <<= 1)
        {
            int const  oc = __shfl_xor(kcol, off), orow = __shfl_xor(krow, off);
            bool const take = oc < kcol;
            kcol = take ? oc : kcol;
            krow = take ? orow : krow;
        }
        if (in_list && is_first)
        {
            EndCell ec{};
            if (bad || (tgt > 0 && kcol == 0x7fffffff))
                ec.score = -1; // the score of pass 1 was not reproduced: never return a wrong alignment silently
            else if (tgt > 0)
            {
                ec.score = tgt;
                ec.q_end = kcol + 1;
                ec.s_end = krow + 1;
            }
            p.ends[e] = ec;
        }
    }
    else
    {
        int const gbest = run_best, gstrip = run_strip, grow = run_row, gtie = run_tie;
        if (in_list && is_first)
        {
            EndCell ec{};
            if (bad)
                ec.score = -1;
            else if (active && gbest > 0)
            {
                ec.score = gbest;
                ec.q_end = -(gstrip + 1); // the backtrace finds the column inside this strip
                ec.s_end = grow + 1;
                ec.flags = (gtie ? kEndAmbiguous : 0) | (int32_t)(ovf_tag << kEndOverflowShift);
            }
            p.ends[e] = ec;
            if (p.score_out)
                p.score_out[e] = bad ? -1 : (active ? gbest : 0);
        }
    }
}

// Backtrace: one lane per extension AT A TIME, lanes are persistent -- a lane that has finished its extension takes the
// next one from a device-side queue (p.work_counter), so the lockstep of a wavefront costs the average walk, not the
// slowest of 64.  Two ways forward from the current cell (i, j), both reproducing the forward pass's decisions exactly:
//
//   (1) Diagonal shortcut (no DP).  The slot holds H on every tile border: H(row, last column of a strip) for every row
//       (boundary array) and H(last row of a step block, column) for every column (row checkpoints).  In state "H", walk
//       the diagonal from (i, j) to the border cell (i - k, j - k) of the tile, summing the substitution scores.  The
//       recurrence gives H(a) >= H(b) + S(b -> a) for every diagonal piece and H >= 0, hence with L_t = H(i, j) - (sum of
//       the first t scores) >= H(i - t, j - t):
//         * L_k == H(border cell)  =>  every inequality on the piece is an equality, i.e. at every cell of it the diagonal
//           candidate equals H; the traceback prefers the diagonal on ties (GapsLeft: diagonal > vertical > horizontal,
//           /root/reference/src/search_algo.hpp:1083), so the walk IS this diagonal: k columns 'M', no tile needed;
//         * L_t == 0 for some t  =>  H(i - t, j - t) = 0 exactly (it is >= 0 and <= L_t): the alignment begins after that
//           cell, the t cells before it are the walk.
//       Anything else (a gap or another path inside this tile) leaves the lane "blocked" for
//   (2) the tile recomputation of the first version of this kernel: the 16 x C tile of (strip, step block) from the
//       checkpoint above it and the boundary array of the strip to its left, tagged arithmetic of lx_trace.hip (values
//       x 4, the two low bits resolve the traceback ties) in its plain, un-skewed form:
//           tt = 4 H(i-1,j-1) + (4 s + 3);  m = max3(tt, E|1, F|2);  H = m & ~3;  A = H + 4 go;
//           Fr = max3(F + 4 ge, A, 0), F' = Fr | 2;  Er = max(E + 4 ge, A), E' = Er | 1;  nibble = tag(m) | (Fr|Er)&3 << 2
//       nibbles in LDS.  Needed for the first tile of a single-sweep extension (its end column is still open: the tile
//       yields the end cell, the walk then starts like anywhere else), for tiles with gaps, and while the walk is inside a
//       gap.  In such a tile the walk takes the leading diagonal steps in one go (the nibbles say how many; same routine
//       as the shortcut, without the border test), then single steps through the gap, and hands back to (1) as soon as
//       it is in state H again.
// Per outer iteration: shortcut pass (every lane that can crosses up to LX_BT_HOPS tile borders; the border words and
// residues of all hops are requested together, one round trip to memory) -> finished lanes are retired and refilled -> one
// tile phase for every lane that needs it (new extensions start with one).  On the headline batch an alignment crosses ~17
// tiles and has ~3 gaps: 3.7 tile phases per extension instead of 17-18.
// What bounds the kernel (r02 profile): ~2/3 VALU issue, the rest memory latency of the scattered border-word reads that
// only more resident wavefronts hide -- 8 per CU (256 VGPRs, no spills) beat 11 with the ~10 spilled dwords 168 VGPRs cost.
template <int G, int C>
__global__ __launch_bounds__(64, LX_BT_WAVES) void ckpt_backtrace_kernel(TraceParams p)
{
    using Lay               = CkptLayout<G, C>;
    using L16               = Ckpt16Layout<G, C>;
    constexpr int kNibDw    = (C + 7) / 8; // dwords of direction nibbles per tile row
    constexpr int kFar      = -(1 << 28);  // "minus infinity" that survives a few additions (multiple of 4)
    static_assert(kCkptEvery == 16, "the diagonal shortcut reads at most 16 residues per sequence");
    __shared__ int8_t   smat4[kAlph * kAlph]; // 4 s + 3: diagonal step of the tile DP with its tag (the walk divides it back)
    __shared__ int8_t   smat1n[kAlph * kAlph]; // -s for the diagonal runs; the pair (31, 31) -- rank 31 is the reserved pad rank -- scores 0 there
    __shared__ uint32_t tiles[kCkptEvery * kNibDw * 64]; // [tile row][word][lane]: lane-minor, conflict-free
    for (int x = threadIdx.x; x < kAlph * kAlph; x += blockDim.x)
    {
        int const v = p.sc->mat[x];
        smat4[x]    = (int8_t)((v < -32 || v > 31) ? -125 : 4 * v + 3); // pad ranks (and entries pass 2 does not admit) far down
        smat1n[x]   = (int8_t)(x == kAlph * kAlph - 1 ? 0 : min(-v, 127));
    }
    __syncthreads();
    uint32_t const lane  = threadIdx.x;
    uint64_t       limit = p.n;
    if (p.count_ptr)
    {
        uint64_t const total = *p.count_ptr;
        limit = total > p.chunk_start ? min(p.n, total - p.chunk_start) : 0;
    }
    int const ge = p.sc->ge, g2 = p.sc->g2;                 // a gap of k characters costs g2 + k ge
    int const ge4 = 4 * p.sc->ge, go4 = 4 * p.sc->go;       // tile DP: values x 4 (go = first gap character)
    // (per extension: the sweep's slots may be laid out wavefront by wavefront, TraceParams::wf_tab)
    uint64_t panel_dw = Lay::slot_dwords(p.steps_cap), panel16_dw = L16::slot_dwords(p.steps_cap), bnd_dw = Lay::bnd_dwords(p.steps_cap);

    // ---- state of the extension this lane is working on
    bool             have = false, blocked = false, done = false, need_col = false, scan = false, c16 = false;
    uint64_t         po = 0;          // where the extension's record and ops slot are
    uint64_t         pos = 0;         // its position in the list
    EndCell          ec{};
    uint32_t const * slot = nullptr;
    // compact slots whose wavefront's W = 128 / G slots are interleaved (ScoreParams::wave_slots, kEndWaveSlots): uint4 units between two
    // groups of eight steps / two row checkpoints of ONE window, dwords from `slot` to its first row checkpoint
    uint32_t         oct_mul = G, ck_mul = G * (L16::kCkDw / 4);
    uint64_t         ck16_off = 0;
    uint8_t const *  q = nullptr, * s = nullptr;
    uint8_t *        ops_al = nullptr;
    uint32_t         cap = 0, a0 = 0, apos = 0, acc = 0, n = 0;
    int              lq = 0, ls = 0;
    int              i = 0, j = 0, end_i = 0, end_j = 0;
    int              res_col = C, res_row = 0x7fffffff; // resolution of the end cell inside the first tile
    // The multi-query sweep may run an extension's LAST panel with narrower strips (kEndNarrowShift): columns from nar_j0 on
    // lie in strips of nar_cw columns, strip nar_st0 first.  Panel starts, strip numbering and the row grids are as ever.
    int              nar_cw = C, nar_j0 = 0x7fffffff, nar_st0 = 0x7fffffff;
    // strip, first column and width of the strip that holds column aj
    auto locate = [&](int aj, int & st, int & j0, int & cw)
    {
        if (aj >= nar_j0)
        {
            int const jl = aj - nar_j0;
            int const gl = nar_cw == (3 * C + 3) / 4 ? jl / ((3 * C + 3) / 4) : nar_cw == (C + 1) / 2 ? jl / ((C + 1) / 2) : jl / ((C + 3) / 4);
            st = nar_st0 + gl;
            j0 = nar_j0 + gl * nar_cw;
            cw = nar_cw;
        }
        else
        {
            st = aj / C;
            j0 = st * C;
            cw = C;
        }
    };
    int              mode = 0;                          // 0 = H, 1 = F (vertical), 2 = E (horizontal)
    int              left = 0;
    int32_t          nm = 0, nx = 0, np = 0, go = 0, gx = 0;
    // run-length mode (p.rle): the run being collected (op 0 = 'M', 1 = 'D', 2 = 'I'; 3 = none yet) and the codes written so far
    bool const       rle_mode = p.rle != nullptr;
    uint32_t         run_op = 3, run_len = 0, ncodes = 0;

    // Checkpoint words are int16 pairs (value, gap state).  Slots written by the packed-half sweep hold the compact
    // codes of Ckpt16Layout instead: the loaders below expand them to the same pairs.  An extension that sweep declined
    // has an int16-pair slot of its own in the overflow area.
    auto dec = [](uint32_t bits16) -> int { return (int)(int16_t)bits16; };
    auto expand = [](uint32_t code16) -> uint32_t
    {
        uint32_t const H = code16 & 0x7ffu;
        return H | ((H - (code16 >> 11)) << 16);
    };
    // A query wider than one panel has one part per panel in its slot (int16 pairs or compact codes): global strip st = panel * G +
    // lane; a strip's step for row r is r + lane.
    // (compact codes: one part per panel as well)
    auto bnd_of = [&](uint32_t pn) { return reinterpret_cast<uint4 const *>(slot + (uint64_t)pn * (c16 ? panel16_dw : panel_dw)); };
    auto rowck_of = [&](uint32_t pn)
    {
        return reinterpret_cast<uint4 const *>(slot + (c16 ? (uint64_t)pn * panel16_dw + ck16_off
                                                           : (uint64_t)pn * panel_dw + bnd_dw));
    };
    // Ckpt16Layout's index functions with the slots' spacing
    auto oct16_index = [&](uint32_t o, uint32_t g_) -> uint32_t { return o * oct_mul + g_; };
    auto ck16_index  = [&](uint32_t m_, uint32_t g_, uint32_t x_) -> uint32_t { return m_ * ck_mul + g_ * (L16::kCkDw / 4) + x_; };
    // word (H, F) of column c of strip st's row checkpoint m; word (H, E) of strip st's boundary at step k
    auto rowck_word = [&](uint32_t m, uint32_t st, uint32_t c) -> uint32_t
    {
        if (c16)
            return expand(reinterpret_cast<uint16_t const *>(rowck_of(st / G) + ck16_index(m, st % G, c / 8))[c % 8]);
        return reinterpret_cast<uint32_t const *>(rowck_of(st / G) + rowck_quad_index<G, Lay::kCkDw>(m, st % G, c / 4))[c % 4];
    };
    auto bnd_word_of = [&](uint32_t st, uint32_t k) -> uint32_t
    {
        if (c16)
            return expand(reinterpret_cast<uint16_t const *>(bnd_of(st / G) + oct16_index(k / 8, st % G))[k & 7]);
        return reinterpret_cast<uint32_t const *>(bnd_of(st / G) + bnd_quad_index<G>(k / 4, st % G))[k & 3];
    };
    // ops are produced end -> begin into the slot [0, cap): apos = misalignment of the slot start + offset of the next byte
    auto put_byte = [&](uint32_t byte)
    {
        acc |= byte << (8 * (apos & 3));
        if ((apos & 3) == 0)
        {
            if (apos + 3 <= a0 + cap - 1)
                *reinterpret_cast<uint32_t *>(ops_al + apos) = acc;
            else
                for (uint32_t b = 0; b < 4 && apos + b <= a0 + cap - 1; ++b)
                    ops_al[apos + b] = (uint8_t)(acc >> (8 * b));
            acc = 0;
        }
        --apos;
    };
    // one alignment column: its op byte, or -- run-length mode -- one more column of the current run
    auto emit = [&](uint32_t op)
    {
        if (rle_mode)
        {
            uint32_t const oc = op == (uint32_t)'M' ? 0u : (op == (uint32_t)'D' ? 1u : 2u);
            if (oc != run_op || run_len == 64)
            {
                if (run_op != 3)
                {
                    put_byte((run_op << 6) | (run_len - 1));
                    ++ncodes;
                }
                run_op  = oc;
                run_len = 0;
            }
            ++run_len;
        }
        else
            put_byte(op);
        ++n;
    };

    // ---- a lane takes list position e: end cell, slot, sequences, ops slot; single-sweep extensions whose strip tied
    // get their end cell resolved right here
    auto begin_extension = [&](uint64_t e)
    {
        uint64_t oi = e;
        if (p.src)
        {
            uint32_t const sidx = p.src[e];
            if (sidx == 0xffffffffu)
                return; // padding slot
            oi = sidx;
        }
        pos                = e;
        po                 = p.out_by_pos ? e : oi;
        uint64_t const  se = p.slot_by_src ? oi : e; // single-sweep mode: checkpoints and end cells sit at the original index
        ec                 = p.ends[se];
        Extension const x  = p.ext[e];
        if (ec.score <= 0)
        {
            Hsp out{};
            out.score     = ec.score < 0 ? -1 : 0;
            p.out_hsp[po] = out;
            return;
        }
        c16                = (ec.flags & kEndCompact) != 0;
        uint32_t const ovf = (uint32_t)ec.flags >> kEndOverflowShift;
        slot               = ovf ? p.ovf + (uint64_t)(ovf - 1) * p.ovf_stride : p.trace + se * p.slot_stride;
        uint32_t steps_e   = p.steps_cap;
        if (p.wf_tab && !ovf)
        {
            // the sweep's slots by wavefront (WfSlots): slot se % 16 of wavefront se / 16, laid out for that wavefront's steps and panels
            WfSlots const  t      = p.wf_tab[se / 16];
            uint64_t const stride = (uint64_t)t.panels_cap * (c16 ? L16::slot_dwords(t.steps_cap) : Lay::slot_dwords(t.steps_cap));
            slot                  = ((p.split_n != 0 && se >= p.split_n) ? p.trace2 : p.trace) + t.off_dw + (se % 16) * stride;
            steps_e               = t.steps_cap;
        }
        panel_dw           = Lay::slot_dwords(steps_e);
        panel16_dw         = L16::slot_dwords(steps_e);
        bnd_dw             = Lay::bnd_dwords(steps_e);
        oct_mul            = G;
        ck_mul             = G * (L16::kCkDw / 4);
        ck16_off           = L16::bnd_dwords(steps_e);
        if (!ovf && (ec.flags & kEndWaveSlots))
        {
            constexpr uint32_t kW = 128 / G; // windows of a packed-half wavefront
            uint64_t const     w  = se % kW;
            slot     = p.trace + (se - w) * p.slot_stride + w * (G * 4);
            oct_mul  = kW * G;
            ck_mul   = kW * G * (L16::kCkDw / 4);
            ck16_off = (uint64_t)kW * L16::bnd_dwords(steps_e) - w * (G * 4) + w * (G * L16::kCkDw);
        }
        q                  = p.q_res + x.q_off;
        s                  = p.s_res + x.s_off;
        uint8_t * const ops = p.out_ops + (p.ops_off ? p.ops_off[po] : po * p.ops_stride);
        cap                = x.q_len + x.s_len;
        lq                 = (int)x.q_len;
        ls                 = (int)x.s_len;
        i                  = ec.s_end - 1;
        j                  = ec.q_end - 1;
        // single-sweep mode: only the strip of the end cell is known (ec.q_end = -(strip + 1)).  The column is read off the
        // first tile (its last computed rows hold the end row), unless the strip reached the best score in ...
        {
            int const code = (ec.flags >> kEndNarrowShift) & 3;
            int const last = max(1, (lq + G * C - 1) / (G * C)) - 1; // the extension's last panel
            nar_cw  = narrow_strip_cols(C, code);
            nar_j0  = code ? last * (G * C) : 0x7fffffff;
            nar_st0 = code ? last * G : 0x7fffffff;
        }
        need_col = ec.q_end < 0;
        if (need_col)
        {
            int const st = -ec.q_end - 1;
            // provisional: the whole strip is computed
            j = st >= nar_st0 ? nar_j0 + (st - nar_st0) * nar_cw + (nar_cw - 1) : st * C + (C - 1);
            // ... several rows: the tile phases scan the strip from the step block of the first such row to its end
            // (scan mode: one block per phase, in step with the tiles of the other lanes), keeping the lowest column
            // and, for it, the lowest row
            scan = (ec.flags & kEndAmbiguous) != 0;
            if (scan)
            {
                int const gl = st % G;
                i            = min(((i + gl) / kCkptEvery) * kCkptEvery - gl + kCkptEvery - 1, ls - 1);
            }
        }
        end_i = i; // final once need_col is false
        end_j = j;
        res_col = C;
        res_row = 0x7fffffff;
        mode = 0;
        left = ec.score;
        n    = 0;
        nm = nx = np = go = gx = 0;
        a0     = (uint32_t)(reinterpret_cast<uintptr_t>(ops) & 3);
        ops_al = ops - a0;
        apos   = a0 + cap - 1;
        acc    = 0;
        run_op  = 3;
        run_len = 0;
        ncodes  = 0;
        have    = true;
        done    = false;
        blocked = false;
    };

    // ---- the record of a finished extension
    // ---- the last code / the bytes collected in the lowest, partial dword leave
    auto close_ops = [&]()
    {
        if (mode != 0)
            go += 1; // ran into the border right after a gap character: it can only have been an opening
        if (rle_mode && run_op != 3)
        {
            put_byte((run_op << 6) | (run_len - 1));
            ++ncodes;
        }
        if ((apos & 3) != 3)
            for (uint32_t b = (apos & 3) + 1; b < 4 && (apos & ~3u) + b <= a0 + cap - 1; ++b)
                ops_al[(apos & ~3u) + b] = (uint8_t)(acc >> (8 * b));
    };
    // ---- the record of a finished extension; run-length mode: its codes move from the slot to `at` in the dense stream
    auto finish_extension = [&](uint64_t at)
    {
        if (rle_mode && ec.score >= 0)
        {
            if (at + ncodes > p.rle_cap)
            {
                atomicExch(p.err, 5);
                ec.score = -1;
            }
            else
            {
                uint8_t const * const src = ops_al + apos + 1;
                uint8_t * const       dst = p.rle + at;
                for (uint32_t k = 0; k < ncodes; ++k)
                    dst[k] = src[k];
            }
        }
        Hsp out{};
        if (ec.score < 0)
            out.score = -1;
        else
        {
            out.score              = ec.score;
            out.q_begin            = j + 1;
            out.q_end              = end_j + 1;
            out.s_begin            = i + 1;
            out.s_end              = end_i + 1;
            out.n_ops              = (int32_t)n;
            out.num_matches        = nm;
            out.num_mismatches     = nx;
            out.num_positives      = np;
            out.num_gap_opens      = go;
            out.num_gap_extensions = gx;
            out.ops_shift          = rle_mode ? (int32_t)(uint32_t)at : (int32_t)(cap - n);
        }
        p.out_hsp[po] = out;
        if (rle_mode && p.rle_len)
            p.rle_len[pos] = ec.score > 0 ? ncodes : 0u;
        have          = false;
    };


    // ---- k cells up the diagonal from (i, j) in state H, 1 <= k <= 16: the walk takes them if the piece is the traceback's
    // path -- verify: L after k cells equals Hb, the stored H of the cell beyond (see above); !verify: the caller knows
    // (direction nibbles of a tile) -- or up to the cell whose H is 0 (L == 0: the alignment begins after it).  Returns
    // false, and changes nothing, if the piece is not the path.
    // residues of the k cells up the diagonal from (ci, cj): q[cj - k + 1 .. cj], s[ci - k + 1 .. ci] = bytes 0 .. k - 1 of
    // the dwords (256 bytes of slack behind the buffers)
    auto load_diagonal = [&](int ci, int cj, int k, uint32_t (&qw)[4], uint32_t (&sw)[4])
    {
#pragma unroll
        for (int d = 0; d < 4; ++d)
        {
            qw[d] = *reinterpret_cast<unaligned_u32 const *>(q + (cj - k + 1) + 4 * d);
            sw[d] = *reinterpret_cast<unaligned_u32 const *>(s + (ci - k + 1) + 4 * d);
        }
    };
    auto take_diagonal = [&](int k, int Hb, bool verify, uint32_t (&qw)[4], uint32_t (&sw)[4]) -> bool
    {
        // bytes >= k of the pieces become the pair (31, 31), which scores 0 in smat1n
#pragma unroll
        for (int d = 0; d < 4; ++d)
        {
            int const      nb  = k - 4 * d; // bytes of this dword inside the piece
            uint32_t const pad = nb >= 4 ? 0u : (nb <= 0 ? 0x1f1f1f1fu : (0x1f1f1f1fu << (8 * nb)));
            qw[d]              = (qw[d] & 0x1f1f1f1fu) | pad;
            sw[d]              = (sw[d] & 0x1f1f1f1fu) | pad;
        }
        // One pass over the 16 byte positions, the cells beyond the piece first (they score 0 and change nothing but the
        // count), then the entry cell, then up the diagonal: P = sum of the scores taken so far, frozen once the cell at
        // hand has H = 0 (L = left - P <= 0).  Per cell: index, LDS read, compare, carry, select, subtract, shift.
        int      P = 0, cnt = 0;
        uint32_t posbits = 0;
        int32_t  tmb = 0;
        bool const bs = p.bs_match_rule != 0; // (uniform)
#pragma unroll
        for (int u = kCkptEvery - 1; u >= 0; --u) // cell (i - t, j - t) is byte u = k - 1 - t
        {
            int const      d   = u >> 2, b = u & 3;
            uint32_t const idx = (((qw[d] >> (8 * b)) & 0x1fu) << 5) | ((sw[d] >> (8 * b)) & 0x1fu);
            int const      nv  = (int)smat1n[idx];
            bool const     take = P < left;
            int const      nve = take ? nv : 0;
            cnt += take ? 1 : 0;
            P -= nve;
            posbits = __builtin_amdgcn_alignbit(posbits, (uint32_t)nve, 31); // bit = the score taken is positive
            if (bs) // the bisulfite overload of computeAlignmentStats: a match scores what the letter scores with itself
                tmb += (take && nv == (int)smat1n[(idx >> 5) * (kAlph + 1)]) ? 1 : 0;
        }
        int const beyond = left > 0 ? kCkptEvery - k : 0; // cells beyond the piece that were counted
        cnt -= beyond;
        tmb -= bs ? beyond : 0;
        int const  L       = left - P;
        bool const stopped = cnt < k;
        // L < 0 cannot happen (L_t >= H >= 0); were it to, the piece is left to the tile DP, never guessed
        if (!(stopped ? (L == 0) : (!verify || L == Hb)))
            return false;
        // identical letters among the cnt cells taken = bytes [k - cnt, k) of the two pieces: zero bytes of the XOR,
        // counted four at a time
        int32_t tm = tmb;
        if (!bs)
        {
            int const lo = k - cnt;
#pragma unroll
            for (int d = 0; d < 4; ++d)
            {
                uint32_t const x   = qw[d] ^ sw[d];
                uint32_t const nz  = (x + 0x7f7f7f7fu) & 0x80808080u;  // bit 7 of every non-zero byte (bytes < 0x80)
                int const      b0  = max(lo - 4 * d, 0), b1 = min(k - 4 * d, 4); // bytes [b0, b1) of this dword count
                uint32_t const msk = b1 > b0 ? ((b1 >= 4 ? 0xffffffffu : ((1u << (8 * b1)) - 1u)) & ~((1u << (8 * b0)) - 1u)) : 0u;
                tm += __popc(~nz & 0x80808080u & msk);
            }
        }
        left = L;
        nm += tm;
        nx += cnt - tm;
        np += __popc(posbits);
        if (rle_mode) // cnt columns 'M' extend the current run (codes of at most 64 columns)
        {
            int left_m = cnt;
            while (left_m > 0)
            {
                if (run_op != 0 || run_len == 64)
                {
                    if (run_op != 3)
                    {
                        put_byte((run_op << 6) | (run_len - 1));
                        ++ncodes;
                    }
                    run_op  = 0;
                    run_len = 0;
                }
                int const take = min(left_m, 64 - (int)run_len);
                run_len += (uint32_t)take;
                left_m -= take;
            }
        }
        else if (cnt > 0) // cnt bytes 'M' below apos: at most five dwords, the lowest one may stay open in acc
        {
            uint32_t const hi = apos, lo = apos + 1 - (uint32_t)cnt, Dtop = hi & ~3u, Dlow = lo & ~3u;
#pragma unroll
            for (int sl = 0; sl < 5; ++sl)
            {
                uint32_t const D = Dtop - 4u * sl;
                if (Dtop >= 4u * sl && D >= Dlow)
                {
                    uint32_t const b0  = lo > D ? lo - D : 0, b1 = sl == 0 ? hi - D : 3u; // bytes b0 .. b1 of dword D
                    uint32_t const msk = (b1 >= 3 ? 0xffffffffu : ((1u << (8 * (b1 + 1))) - 1u)) & ~((1u << (8 * b0)) - 1u);
                    uint32_t const val = (sl == 0 ? acc : 0u) | (0x4d4d4d4du & msk);
                    if (lo <= D) // byte 0 of the dword written: the dword is complete
                    {
                        if (sl != 0 || D + 3 <= a0 + cap - 1)
                            *reinterpret_cast<uint32_t *>(ops_al + D) = val;
                        else
                            for (uint32_t bb = 0; bb < 4 && D + bb <= a0 + cap - 1; ++bb)
                                ops_al[D + bb] = (uint8_t)(val >> (8 * bb));
                        acc = 0;
                    }
                    else
                        acc = val;
                }
            }
            apos -= (uint32_t)cnt;
        }
        n += (uint32_t)cnt;
        i -= cnt;
        j -= cnt;
        done = stopped;
        return true;
    };

    // Scheduling of the wavefront: a shortcut pass costs ~1/8 of a tile phase, so shortcuts run while enough lanes can use
    // them; a tile phase runs when LX_BT_TILE_AT lanes wait for one (or nobody can do anything else), right after the
    // finished lanes have been retired and refilled (a new extension starts with a tile).
    bool queue_empty = false; // wave-uniform
    int const tile_at = p.bt_tile_at > 0 ? p.bt_tile_at : LX_BT_TILE_AT, refill_at = p.bt_refill_at > 0 ? p.bt_refill_at : LX_BT_REFILL_AT;
    for (;;)
    {
        bool fin       = have && (done || i < 0 || j < 0 || n >= cap);
        bool can       = have && !fin && !blocked && !need_col && mode == 0;
        bool tile_need = have && !fin && !can;
        int  n_can = __popcll(__ballot(can)), n_tile = __popcll(__ballot(tile_need));
        int const n_idle = 64 - n_can - n_tile; // finished or empty lanes
        bool run_tile = n_can == 0 || n_tile >= tile_at;
        if (n_idle > 0 && (run_tile || n_idle >= refill_at))
        {
            // ================= retire and refill
            if (fin)
                close_ops();
            uint64_t at = 0;
            if (rle_mode)
            {
                // room in the dense code stream for every lane that finishes now: one atomic per wavefront
                uint32_t const mine = fin ? ncodes : 0u;
                uint32_t       incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1)
                {
                    uint32_t const up = (uint32_t)__shfl_up((int)incl, off);
                    if ((int)lane >= off)
                        incl += up;
                }
                uint32_t const     total = (uint32_t)__shfl((int)incl, 63);
                unsigned long long base  = 0;
                if (lane == 63 && total != 0)
                    base = atomicAdd(p.rle_top, (unsigned long long)total);
                base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 63);
                at   = base + incl - mine;
            }
            if (fin)
                finish_extension(at);
            if (!queue_empty)
            {
                uint64_t const want = __ballot(!have);
                if (want != 0)
                {
                    uint32_t  base   = 0;
                    int const leader = __ffsll((unsigned long long)want) - 1;
                    if ((int)lane == leader)
                        base = atomicAdd(p.work_counter, (uint32_t)__popcll(want));
                    base = (uint32_t)__shfl((int)base, leader);
                    if ((uint64_t)base + (uint32_t)__popcll(want) >= limit)
                        queue_empty = true;
                    if (!have)
                    {
                        uint64_t const e = (uint64_t)base + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                        if (e < limit)
                            begin_extension(e);
                    }
                }
            }
            fin       = have && (done || i < 0 || j < 0 || n >= cap); // (an extension without a positive score never gets here)
            can       = have && !fin && !blocked && !need_col && mode == 0;
            tile_need = have && !fin && !can;
            n_can     = __popcll(__ballot(can));
            n_tile    = __popcll(__ballot(tile_need));
            if (n_can == 0 && n_tile == 0)
            {
                if (__ballot(have) == 0 && queue_empty)
                    break;
                if (__ballot(have) == 0)
                    continue; // (only padding slots or score-less extensions came out of the queue: take more)
            }
            run_tile = n_can == 0 || n_tile >= tile_at;
        }

        if (!run_tile)
        {
            // ================= (1) diagonal shortcuts for every lane that can take them: up to LX_BT_HOPS tile borders in
            // one pass.  Where the hops end is geometry alone (the diagonal through (i, j)), so the border words and the
            // residues of all of them are requested at once -- one round trip to memory for the lot; hop h + 1 counts only if
            // hop h reached its border.
            if (can)
            {
                int      hk[LX_BT_HOPS];
                uint32_t hw[LX_BT_HOPS], hqw[LX_BT_HOPS][4], hsw[LX_BT_HOPS][4];
                int      ci = i, cj = j;
#pragma unroll
                for (int h = 0; h < LX_BT_HOPS; ++h)
                {
                    bool const on = ci >= 0 && cj >= 0;
                    int const ai = max(ci, 0), aj = max(cj, 0);
                    int st, j0, cw;
                    locate(aj, st, j0, cw);
                    int const c  = aj - j0;
                    int const gl = st % G;
                    int const m  = (ai + gl) / kCkptEvery;
                    int const r_base = m * kCkptEvery - gl;       // row of tile row 0
                    int const kt = ai - r_base + 1, kl = c + 1;   // diagonal steps to the row above the tile / the column left of it
                    int const k  = min(min(kt, kl), ai + 1);      // 1 .. 16; never above the matrix (block 0 has virtual rows)
                    int const bi = ai - k, bj = aj - k;           // the border cell
                    hk[h] = on ? k : 0;
                    hw[h] = 0;                                    // (beyond the matrix: H = 0)
                    if (on && bi >= 0 && bj >= 0)
                        hw[h] = (kl <= kt) ? bnd_word_of((uint32_t)(st - 1), (uint32_t)(bi + (st - 1) % G))
                                           : rowck_word((uint32_t)(m - 1), (uint32_t)st, (uint32_t)(bj - j0));
                    load_diagonal(ai, aj, k, hqw[h], hsw[h]);
                    ci -= k;
                    cj -= k;
                }
#pragma unroll
                for (int h = 0; h < LX_BT_HOPS; ++h)
                    if (hk[h] > 0 && !blocked && !done)
                        if (!take_diagonal(hk[h], dec(hw[h] & 0xffffu), true, hqw[h], hsw[h]))
                            blocked = true;
            }
            continue;
        }

        // ================= (2) one tile for every lane that cannot go on without it
        if (tile_need)
        {
        // ---- the tile of the current cell: strip st, step block m (step k = row + strip)
        int st, j0, cw; // (cw: columns of the strip that are real -- the others are treated like columns beyond the query)
        locate(j, st, j0, cw);
        int const pn = st / G, gl = st % G; // panel and lane of this strip (one panel: gl = st)
        int const m  = (i + gl) / kCkptEvery;
        int const k_base = m * kCkptEvery;  // first step of the block
        int const r_base = k_base - gl;     // row of tile row 0 (may be negative in block 0: virtual rows)
        uint4 const * const rowck = rowck_of((uint32_t)pn);

        // top edge: H(r_base - 1, c) and the folded F(r_base, c)
        int Hp[C], F[C];
        if (m == 0)
        {
#pragma unroll
            for (int c = 0; c < C; ++c)
            {
                Hp[c] = 0;
                F[c]  = 2; // folded floor, tag "vertical"
            }
        }
        else
        {
            uint32_t      w[Lay::kCkDw];
            if (c16)
            {
#pragma unroll
                for (int xq = 0; xq < L16::kCkDw / 4; ++xq)
                {
                    uint4 const    v    = rowck[ck16_index((uint32_t)(m - 1), (uint32_t)gl, (uint32_t)xq)];
                    uint32_t const d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                    {
                        int const c = 2 * (4 * xq + b);
                        if (c < Lay::kCkDw)
                            w[c] = expand(d[b] & 0xffffu);
                        if (c + 1 < Lay::kCkDw)
                            w[c + 1] = expand(d[b] >> 16);
                    }
                }
            }
            else
            {
#pragma unroll
                for (int xq = 0; xq < Lay::kCkDw / 4; ++xq)
                {
                    uint4 const v = rowck[rowck_quad_index<G, Lay::kCkDw>((uint32_t)(m - 1), (uint32_t)gl, (uint32_t)xq)];
                    w[4 * xq] = v.x; w[4 * xq + 1] = v.y; w[4 * xq + 2] = v.z; w[4 * xq + 3] = v.w;
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c)
            {
                Hp[c] = 4 * dec(w[c] & 0xffffu);
                F[c]  = (4 * dec(w[c] >> 16)) | 2;
            }
        }
        // LDS offsets of the matrix rows of this strip's query residues (columns beyond the query use the pad rank)
        uint32_t qoff2[(C + 1) / 2]; // two columns per register (the offsets are < 1024): the DP is the register peak of the kernel
        uint32_t qd[(C + 3) / 4];    // the strip's query residues: the walk picks its column's letter from these
        {
#pragma unroll
            for (int d = 0; d < (C + 3) / 4; ++d)
                qd[d] = *reinterpret_cast<unaligned_u32 const *>(q + j0 + 4 * d); // 256 bytes of slack behind the residues
#pragma unroll
            for (int c = 0; c < C; ++c)
            {
                uint32_t const r = (c < cw && j0 + c < lq) ? ((qd[c >> 2] >> (8 * (c & 3))) & (kAlph - 1)) : (uint32_t)(kAlph - 1);
                qoff2[c >> 1]    = (c & 1) ? (qoff2[c >> 1] | ((r * kAlph) << 16)) : r * kAlph;
            }
        }
        // left edge: the boundary words of strip st - 1 for the steps k_base - 2 ... (word of step k - 1 holds E for
        // this strip's row k - st, the word of step k - 2 the diagonal H); quads are indexed by step / 4
        // The left neighbour is the previous lane of this panel -- its step for a row is this strip's step - 1 -- or, for
        // the first strip of a later panel, the last lane of the previous panel: step + G - 1, i.e. G / 4 quads further on
        // with the same word positions.
        bool const has_left = st > 0;
        int const  gL       = gl > 0 ? gl - 1 : G - 1;
        int const  qshift   = (gl == 0 && has_left) ? G / 4 : 0;
        uint4 const * const bndL = bnd_of((uint32_t)(gl > 0 ? pn : max(pn - 1, 0)));
        auto load_bq = [&](int qd) -> uint4 // quad of steps 4 qd .. 4 qd + 3 (this strip's count) of the left neighbour
        {
            if (!has_left || qd + qshift < 0)
                return make_uint4(0, 0, 0, 0);
            if (c16)
            {
                // half of the 16-byte group of eight steps
                int const   qs = qd + qshift;
                uint2 const v  = reinterpret_cast<uint2 const *>(bndL + oct16_index((uint32_t)qs / 2, (uint32_t)gL))[qs & 1];
                return make_uint4(expand(v.x & 0xffffu), expand(v.x >> 16), expand(v.y & 0xffffu), expand(v.y >> 16));
            }
            return bndL[bnd_quad_index<G>((uint32_t)(qd + qshift), (uint32_t)gL)];
        };
        int const qd0   = k_base / 4; // quad that holds step k_base
        uint4     qprev = load_bq(qd0 - 1), qcur = load_bq(qd0);

        // The rows are computed without a branch per lane: rows below the lane's cell (row > i) yield values nobody reads
        // (the walk starts at i, the end-cell search below stops there), virtual rows of block 0 (row < 0) see the pad
        // letter and no left neighbour, which leaves H = 0 and F at its floor.  The loop ends when no lane has rows left.
        uint32_t snext = *reinterpret_cast<unaligned_u32 const *>(s + max(r_base, 0));
#pragma unroll 1
        for (int t = 0; t < kCkptEvery / 4; ++t)
        {
            if (__ballot(r_base + 4 * t <= i) == 0)
                break;
            uint4 const qnext = load_bq(qd0 + t + 1); // needed by the next iteration
            // subject letters of the 4 rows r_base + 4t ... (rows < 0 are virtual and skipped), loaded one iteration ahead
            // (reads run into the slack behind the residues at most)
            int const      row0 = r_base + 4 * t;
            int const      lb   = max(row0, 0);
            uint32_t const sdw  = snext;
            snext               = *reinterpret_cast<unaligned_u32 const *>(s + max(row0 + 4, 0));
            // boundary words by tile row u: (left word = step k-1, diagonal word = step k-2)
            uint32_t const lw[4] = {qprev.w, qcur.x, qcur.y, qcur.z};
            uint32_t const dw[4] = {qprev.z, qprev.w, qcur.x, qcur.y};
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                int const      row = row0 + u;
                uint32_t const tl  = row >= 0 ? (sdw >> (8 * (row - lb))) & (kAlph - 1) : (uint32_t)(kAlph - 1);
                int            E   = (has_left && row >= 0) ? ((4 * dec(lw[u] >> 16)) | 1) : (kFar | 1);
                int            Hd  = (has_left && row > 0) ? 4 * dec(dw[u] & 0xffffu) : 0;
                uint32_t       w[kNibDw];
#pragma unroll
                for (int xw = 0; xw < kNibDw; ++xw)
                    w[xw] = 0;
                // all the row's matrix entries first: one LDS round trip per row instead of one per cell
                int sub4[C];
#pragma unroll
                for (int c = 0; c < C; ++c)
                    sub4[c] = (int)smat4[((qoff2[c >> 1] >> (16 * (c & 1))) & 0xffffu) + tl];
#pragma unroll
                for (int c = 0; c < C; ++c)
                {
                    int const tt  = Hd + sub4[c];                       // tag 3
                    Hd            = Hp[c];
                    int const m   = max3i(tt, E, F[c]);                 // E tag 1 < F tag 2 < diagonal tag 3
                    int const H4  = m & ~3;
                    int const A   = H4 + go4;                           // gap-open candidate, tag 0
                    int const Fr  = max3i(F[c] + ge4, A, 0);            // tag 2 = the vertical gap extends (wins ties)
                    int const Er  = max(E + ge4, A);                    // tag 1 = the horizontal gap extends (wins ties)
                    F[c]          = Fr | 2;
                    E             = Er | 1;
                    Hp[c]         = H4;
                    uint32_t wc   = w[c >> 3];
                    wc            = __builtin_amdgcn_alignbit((uint32_t)m, wc, 2);
                    wc            = __builtin_amdgcn_alignbit((uint32_t)(Fr | Er), wc, 2); // bit 0 = E extended, bit 1 = F
                    w[c >> 3]     = wc;
                }
                int const kk = 4 * t + u;
                w[kNibDw - 1] |= tl; // the row's subject letter rides in the spare low bits of the last word (for the walk)
#pragma unroll
                for (int xw = 0; xw < kNibDw; ++xw)
                    tiles[(kk * kNibDw + xw) * 64 + lane] = w[xw];
                if (need_col && row <= i && row >= 0 && row < ls) // (every row of the tile up to the reported one: the sweeps report the block, not the row; a block may end behind the last row)
                {
                    // lowest column of this row whose H equals the score (no H exceeds it: H - score <= 0, a multiple of 4
                    // after the tags are masked), as a maximum of keys without compares: key = (H - score) * 32 + (C - c)
                    int const tgt4 = 4 * ec.score;
                    int       kmax = 0;
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        kmax = max(kmax, ((Hp[c] - tgt4) << 5) + (C - c));
                    int const  cmin  = C - kmax;                  // C (none) when no key is positive
                    bool const lower = kmax > 0 && cmin < res_col; // rows ascend: an equal column keeps its earlier row
                    res_col          = lower ? cmin : res_col;
                    res_row          = lower ? row : res_row;
                }
            }
            qprev = qcur;
            qcur  = qnext;
        }

        bool walk_ok = true;
        if (scan && r_base + kCkptEvery < ls)
        {
            // scan mode, rows left below this block: the next phase takes the next block
            i       = min(r_base + 2 * kCkptEvery - 1, ls - 1);
            walk_ok = false;
        }
        else if (need_col)
        {
            // the end row is one of the tile's computed rows (the packed sweeps report the block of sixteen steps -- the tile --,
            // the int32 sweep the exact row: earlier rows of the strip then do not carry the score), or one of those scanned:
            // lowest column, then lowest row.  From the end cell the walk starts like anywhere else: with a diagonal shortcut.
            need_col = false;
            scan     = false;
            walk_ok  = false;
            if (res_col >= C)
            {
                done     = true; // the forward pass saw this score in these rows of the strip: never guess
                left     = -1;
                ec.score = -1;
            }
            else
            {
                i     = res_row;
                j     = j0 + res_col;
                end_i = i;
                end_j = j;
            }
        }
        // ---- walk inside the tile.  Written with selects instead of nested branches: one data-dependent exit (the
        // alignment's first cell), everything else is arithmetic on the nibble -- bit 3 / bit 2 = the vertical /
        // horizontal gap state extends, low two bits = where H came from (3 diagonal, 2 vertical, 1 horizontal).
        static_assert(32 - 4 * (C - 8 * (kNibDw - 1)) >= 5, "the last nibble word needs 5 spare bits for the subject letter");
        // The walk takes the lane through what the shortcut cannot: up to and through the next gap.  Back in state H
        // with a gap behind it, the rest of the tile is the shortcut's again (1/4 of the instructions per column, and the
        // lockstep of the wavefront pays the longest walk of its lanes).
        bool passed = false;
        if (walk_ok && mode == 0 && i >= 0 && j >= j0 && i >= r_base)
        {
            // the leading diagonal steps all at once: cell (i - t, j - t) continues the run if its H came from the diagonal
            int const kk = i - r_base, c = j - j0;
            int const kmax = min(min(kk, c), i) + 1; // cells of the diagonal inside the tile (and the matrix)
            uint32_t  nd = 0;                        // bit t: not a diagonal step (or beyond)
#pragma unroll
            for (int t = 0; t < kCkptEvery; ++t)
            {
                int const      rr   = max(kk - t, 0), cc = max(c - t, 0);
                int const      xw   = cc >> 3;
                uint32_t const word = tiles[(rr * kNibDw + xw) * 64 + lane];
                int const      sh   = (xw == kNibDw - 1 ? 32 - 4 * (C - 8 * (kNibDw - 1)) : 0) + 4 * (cc & 7);
                nd |= (((word >> sh) & 3u) != 3u) ? (1u << t) : 0u;
            }
            nd |= ~0u << kmax;
            int const d = __ffs((int)nd) - 1;
            if (d > 0)
            {
                uint32_t qw[4], sw[4];
                load_diagonal(i, j, d, qw, sw);
                (void)take_diagonal(d, 0, false, qw, sw);
            }
        }
        while (walk_ok && !done && i >= 0 && j >= j0 && i >= r_base && n < cap && !(passed && mode == 0))
        {
            int const      kk   = i - r_base, c = j - j0;
            int const      xw   = c >> 3;
            int const      cnt  = (xw == kNibDw - 1) ? (C - 8 * xw) : 8; // cells held by this word (funnel-shifted in from the top)
            uint32_t const word = tiles[(kk * kNibDw + xw) * 64 + lane];
            uint32_t const last = tiles[(kk * kNibDw + (kNibDw - 1)) * 64 + lane];
            uint32_t const nib  = (word >> (32 - 4 * cnt + 4 * (c & 7))) & 15u;
            bool const     in_v = mode == 1, in_h = mode == 2;
            bool const     cont = (in_v && (nib & 8u)) || (in_h && (nib & 4u)); // the gap goes on through this cell
            bool const     leave = (in_v || in_h) && !cont;                      // the gap was opened from this cell's H
            go += leave ? 1 : 0;
            left -= leave ? g2 : 0;
            if (!cont && left <= 0)
            {
                done = true; // H of this cell is 0: the alignment begins after it
                break;
            }
            uint32_t const code = nib & 3u;
            bool const     diag = !cont && code == 3, vert = cont ? in_v : code == 2;
            // residues of this cell: column letter from the strip's dwords, row letter from the tile row's last word
            uint32_t qsel = qd[0];
#pragma unroll
            for (int d = 1; d < (C + 3) / 4; ++d)
                qsel = ((c >> 2) == d) ? qd[d] : qsel;
            uint32_t const c0 = (qsel >> (8 * (c & 3))) & (kAlph - 1), c1 = last & (kAlph - 1);
            int const      sm = (int)smat4[c0 * kAlph + c1];
            int const      v  = (sm - 3) >> 2;
            bool const     isMatch = p.bs_match_rule ? (sm == (int)smat4[c0 * kAlph + c0]) : (c0 == c1);
            nm += (diag && isMatch) ? 1 : 0;
            nx += (diag && !isMatch) ? 1 : 0;
            np += (diag && v > 0) ? 1 : 0;
            gx += cont ? 1 : 0;
            left -= diag ? v : ge; // a gap character -- continued or first -- costs ge here, the opening surcharge when it ends
            mode = diag ? 0 : (vert ? 1 : 2);
            passed = passed || !diag;
            emit(diag ? (uint32_t)'M' : (vert ? (uint32_t)'D' : (uint32_t)'I'));
            i -= (diag || vert) ? 1 : 0;
            j -= (diag || !vert) ? 1 : 0;
        }
        blocked = false;
        } // tile_need
    }
}

// ---- host-visible launchers: checkpoint geometries follow the trace geometries (cfg 1 = (8,19), cfg 2 = (16,13)), plus
// cfg 3 = (8,13) for queries of up to 104 columns and cfg 4 = (8,25) for 153-200 columns --------

uint64_t ckpt_slot_dwords(int cfg, uint32_t steps_cap)
{
    return cfg == 2   ? CkptLayout<16, 13>::slot_dwords(steps_cap)
           : cfg == 3 ? CkptLayout<8, 13>::slot_dwords(steps_cap)
           : cfg == 4 ? CkptLayout<8, 25>::slot_dwords(steps_cap)
           : cfg == 5 ? CkptLayout<8, 11>::slot_dwords(steps_cap)
                      : CkptLayout<8, 19>::slot_dwords(steps_cap);
}

// compact slots of the packed-half sweep
uint64_t ckpt16_slot_dwords(int cfg, uint32_t steps_cap)
{
    return cfg == 2   ? Ckpt16Layout<16, 13>::slot_dwords(steps_cap)
           : cfg == 3 ? Ckpt16Layout<8, 13>::slot_dwords(steps_cap)
           : cfg == 4 ? Ckpt16Layout<8, 25>::slot_dwords(steps_cap)
           : cfg == 5 ? Ckpt16Layout<8, 11>::slot_dwords(steps_cap)
                      : Ckpt16Layout<8, 19>::slot_dwords(steps_cap);
}

template <int G, int C>
static hipError_t launch_ckpt_forward_cfg(TraceParams const & p, hipStream_t stream)
{
    using Geo = ScoreGeo<G, C>;
    uint64_t const blocks = (p.n + Geo::kGroups - 1) / Geo::kGroups;
    if (blocks > 0x7fffffffull || (!p.score_in && !p.score_out) || p.steps_cap % kCkptEvery != 0)
        return hipErrorInvalidValue;
    int const    share = p.shared_profile > 1 ? std::min(p.shared_profile, Geo::kGroups) : 1;
    int const    slots = (Geo::kGroups + share - 1) / share;
    size_t const lds   = ((size_t)slots * (size_t)p.nrows * Geo::kRowDw + 64 * 4) * sizeof(uint32_t);
    bool const multi = p.panels_cap > 1;
    if (lds > 64 * 1024)
    {
        // (profiles shared by pairs only -- the fix-up of the free multi-query packing: four int32 profiles of a 152-column panel
        // are 68 KB; gfx950 gives a workgroup up to 160 KB, beyond 64 KB on request)
        hipError_t const ea = multi ? hipFuncSetAttribute(reinterpret_cast<void const *>(&ckpt_forward_kernel<G, C, false, true>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                    : hipFuncSetAttribute(reinterpret_cast<void const *>(&ckpt_forward_kernel<G, C, false, false>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess || p.score_in)
            return ea != hipSuccess ? ea : hipErrorInvalidValue;
    }
    if (p.score_in && multi)
        hipLaunchKernelGGL((ckpt_forward_kernel<G, C, true, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    else if (p.score_in)
        hipLaunchKernelGGL((ckpt_forward_kernel<G, C, true, false>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    else if (multi)
        hipLaunchKernelGGL((ckpt_forward_kernel<G, C, false, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    else
        hipLaunchKernelGGL((ckpt_forward_kernel<G, C, false, false>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_ckpt_forward(TraceParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    return p.cfg == 2   ? launch_ckpt_forward_cfg<16, 13>(p, stream)
           : p.cfg == 3 ? launch_ckpt_forward_cfg<8, 13>(p, stream)
           : p.cfg == 4 ? launch_ckpt_forward_cfg<8, 25>(p, stream)
           : p.cfg == 5 ? launch_ckpt_forward_cfg<8, 11>(p, stream)
                        : launch_ckpt_forward_cfg<8, 19>(p, stream);
}

static int backtrace_resident_waves()
{
    static int const v = []()
    {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        int const per_cu = 4 * LX_BT_WAVES; // 4 SIMDs per CU
        return std::max(1, cus) * per_cu;
    }();
    return v;
}

hipError_t launch_ckpt_backtrace(TraceParams const & p_in, hipStream_t stream)
{
    if (p_in.n == 0)
        return hipSuccess;
    if (!p_in.work_counter)
        return hipErrorInvalidValue;
    TraceParams p = p_in;
    {
        p.bt_tile_at   = dev_aids().bt_tile_at; // development aids
        p.bt_refill_at = dev_aids().bt_refill_at;
    }
    // persistent lanes: as many wavefronts as the chip holds at this kernel's occupancy (LX_BT_WAVES per SIMD), each taking
    // extensions from the queue until it is empty
    uint64_t const b2 = std::min<uint64_t>((p.n + 63) / 64, (uint64_t)backtrace_resident_waves());
    hipError_t e = hipMemsetAsync(p.work_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess)
        return e;
    if (p.cfg == 2)
        hipLaunchKernelGGL((ckpt_backtrace_kernel<16, 13>), dim3((unsigned)b2), dim3(64), 0, stream, p);
    else if (p.cfg == 3)
        hipLaunchKernelGGL((ckpt_backtrace_kernel<8, 13>), dim3((unsigned)b2), dim3(64), 0, stream, p);
    else if (p.cfg == 4)
        hipLaunchKernelGGL((ckpt_backtrace_kernel<8, 25>), dim3((unsigned)b2), dim3(64), 0, stream, p);
    else if (p.cfg == 5)
        hipLaunchKernelGGL((ckpt_backtrace_kernel<8, 11>), dim3((unsigned)b2), dim3(64), 0, stream, p);
    else
        hipLaunchKernelGGL((ckpt_backtrace_kernel<8, 19>), dim3((unsigned)b2), dim3(64), 0, stream, p);
    return hipGetLastError();
}

} // namespace lx

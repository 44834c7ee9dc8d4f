// lx_trace.hip -- pass-2 kernels of the seed extension: forward DP with direction bits + backtrace (gfx950 only).
//
// Replaces _performAlignment<withTrace=true> (/root/reference/src/search_algo.hpp:1296 -> :1070-1134, config
// TracebackOn<CompleteTrace, GapsLeft>, :1083) followed by seqan::_adaptTraceSegmentsTo (:1127) and the walk
// seqan::computeAlignmentStats does over the gapped rows (:1308).  Only survivors of the e-value filter reach
// this pass (:1251-1283).
//
// Kernel A (trace_forward_kernel<G,C>): the strip-systolic geometry and row-skewed recurrence of lx_score.hip.
// Direction bits cost no compares: every DP value is scaled by 4 and its two low bits carry a TAG, so that the
// max instructions the recurrence needs anyway also resolve the traceback ties:
//     tt = Hs[i-1][j-1] + (4(s-ge)+3)        tag 3 = diagonal   (the +3 is baked into the LDS profile)
//     m  = max3(tt, E, F)                     E always carries tag 1 (horizontal gap), F tag 2 (vertical gap)
//          -> on equal values the larger tag wins: diagonal > vertical > horizontal, the priority of a GapsLeft
//             traceback ([UPSTREAM-RECALL], oracle/lx_oracle.c)
//     Fr = max3(F, A, Z')   A = H + go - ge (tag 0), Z' = zero floor of the next row (tag 0);  F' = Fr | 2
//     Er = max(E, A)        E' = (Er & ~3) + 4 ge + 1
//          -> bit 1 of Fr / bit 0 of Er = "the gap EXTENDED" (extension wins ties, i.e. gaps are as long as possible
//             where the score allows).  The zero floor rides in F (max3 costs what max does); a cell whose H is the
//             floor therefore reads "vertical", but no walk ever looks at it: the backtrace adds up the score of
//             the columns it emits and stops when the sum reaches the extension's score, i.e. exactly where H = 0.
// The two tag bits of m and (Fr | Er) & 3 -- 4 bits per cell, 19 issue slots per cell all told -- are funnel-shifted
// (v_alignbit_b32) into ceil(C/8) words per lane per step, staged in LDS for 4 steps and written to the HBM trace
// buffer as whole 16-byte quads, G lanes covering G consecutive quads (layout below).  The end cell under the
// reference's tie rule (first strict maximum in column-major order) is the first cell in that order whose H equals
// the extension's best score, which pass 1 already delivered (p.score_in): one max3 per two cells keeps the row
// maximum, a rare slow path records the column.  Limits (checked by the host): Ls <= kMaxTraceRows (lx_device.h),
// |s - ge| <= 31.
//
// Kernel B (backtrace_kernel): one lane per extension walks from the end cell to the first cell with H = 0, emits
// one op byte per alignment column ('M','D','I') and the counts of lx_hsp.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "lx_dp_common.h"

namespace lx
{

// Trace buffer layout inside one extension slot: word x of kTraceBlock = 4 consecutive steps of one lane forms a
// 16-byte quad, the quads of a step block are ordered [lane][x]:
//     uint4 index = ((panel * steps_cap/4 + k/4) * G + g) * kWords + x,   component k % 4
// The forward kernel writes G consecutive quads per store instruction (whole cache lines); the backtrace, which
// mostly moves from step k to k-1 in the same lane, gets up to four moves out of one 16-byte load.
#ifndef LX_TRACE_BLOCK
#define LX_TRACE_BLOCK 4
#endif
constexpr int kTraceBlock = LX_TRACE_BLOCK;
#ifndef LX_TRACE_UNROLL
#define LX_TRACE_UNROLL 1
#endif
#define LX_TRACE_PRAGMA(x) _Pragma(#x)
#define LX_TRACE_UNROLL_N(n) LX_TRACE_PRAGMA(unroll n)
#define LX_TRACE_UNROLL_PRAGMA LX_TRACE_UNROLL_N(LX_TRACE_UNROLL)
static_assert(kTraceBlock == 4, "the backtrace reads one uint4 = word x of four consecutive steps");
// kLaneBlocks consecutive step blocks of one lane are adjacent in memory (kLaneBlocks * kWords quads).  With 2 or 4
// the backtrace finds its next quads in the cache line it already fetched (2.83 -> 2.79 / 2.68 ms on the headline
// batch) but a store instruction of the forward kernel touches more lines (17.5 -> 18.0 / 18.9 ms): 1 it is.
#ifndef LX_TRACE_LANE_BLOCKS
#define LX_TRACE_LANE_BLOCKS 1
#endif
constexpr int kLaneBlocks = LX_TRACE_LANE_BLOCKS;
// index (uint4 units inside one panel of a slot) of word x of step block kb of lane g
template <int G, int kWords>
__device__ __forceinline__ uint32_t quad_index(uint32_t kb, uint32_t g, uint32_t x)
{
    return ((kb / kLaneBlocks) * G + g) * (kLaneBlocks * kWords) + (kb % kLaneBlocks) * kWords + x;
}

template <int C>
struct TraceWords
{
    static constexpr int kWords = (C + 7) / 8; // 4 direction bits per cell, 8 cells per 32-bit word
};

// MULTI = queries may be wider than one panel of G*C columns (boundary columns are carried through p.ws)
// BAND: band mode, exactly as in lx_score.hip's score_kernel (cells off the band: H = 0, no gap state leaves them).
template <int G, int C, bool MULTI, bool BAND = false>
__global__ __launch_bounds__(64, (C <= 13 ? 4 : 3)) void trace_forward_kernel(TraceParams p)
{
    using Geo = ScoreGeo<G, C>;
    using TW  = TraceWords<C>;
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
        active = false; // padding slot (keeps one query per wavefront)

    ScoringDev const * __restrict__ sc = p.sc;
    int const      ge4   = 4 * sc->ge;
    int const      g20   = 4 * sc->g2;
    int const      ge41  = ge4 + 1; // a horizontal step: + gap_extend, and the E state's tag 1
    // VALU instructions with an SGPR operand issue at half rate on gfx950 (tools/ubench.hip): keep the two constants
    // of the inner loop in VGPRs
    int g20v = g20, ge41v = ge41;
    LX_OPAQUE(g20v);
    LX_OPAQUE(ge41v);
    int const      nrows = p.nrows;
    uint32_t const padt  = (uint32_t)(nrows - 1);

    int             lq = 0, ls = 0;
    uint8_t const * q = p.q_res;
    uint8_t const * s = p.s_res;
    uint64_t        q_off = 0;
    bool const      in_list = e < limit;
    if (in_list)
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
    int blo = 0, bw = 0; // in-band diagonals: blo <= i - j <= blo + bw
    if constexpr (BAND)
    {
        if (active)
        {
            uint64_t const oi = p.src ? p.src[e] : e; // centres are indexed like the caller's extension list
            int const      d0 = p.band_diag ? p.band_diag[oi] : band_default_diag(lq, ls);
            blo               = d0 - p.band;
            bw                = 2 * p.band;
            ls                = min(ls, max(lq + blo + bw, 0));
        }
    }
    // p.shared_profile = number of consecutive groups that share one LDS profile (0/1: none).  The compaction
    // kernel pads every query's survivors to a multiple of that many slots, so the promise holds by construction;
    // it is verified all the same (first lane of each sharing block vs the others).
    int const share  = p.shared_profile > 1 ? min(p.shared_profile, Geo::kGroups) : 1;
    int const leader = (grp / share) * share * G; // first lane of this group's sharing block
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
    int const my_panels = npanels;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        ls_max  = max(ls_max, __shfl_xor(ls_max, off));
        npanels = max(npanels, __shfl_xor(npanels, off));
    }
    ls_max  = __builtin_amdgcn_readfirstlane(ls_max);
    npanels = MULTI ? __builtin_amdgcn_readfirstlane(npanels) : 1;

    bool bad = false;
    if (active && ((uint32_t)my_panels > p.panels_cap || (!MULTI && my_panels > 1) || (uint32_t)((ls + G - 1 + 3) & ~3) > p.steps_cap || ls > kMaxTraceRows))
    {
        bad = true; // the host sized the trace slots too small for this extension: report, never write out of bounds
        atomicExch(p.err, 3);
    }

    int32_t * carry = nullptr;
    if (MULTI && npanels > 1)
    {
        uint32_t base = 0;
        int      ok   = 1;
        if (is_first && my_panels > 1 && active)
        {
            base = atomicAdd(p.ws_top, (uint32_t)ls);
            if (base + (uint32_t)ls > p.ws_cap)
            {
                ok = 0;
                atomicExch(p.err, 1);
            }
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1)
        {
            base = max(base, (uint32_t)__shfl_xor((int)base, off));
            ok   = min(ok, __shfl_xor(ok, off));
        }
        if (my_panels > 1 && ok)
            carry = p.ws + 2ull * base;
        else if (my_panels > 1)
            bad = true;
    }
    {
        int b = bad ? 1 : 0;
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1)
            b = max(b, __shfl_xor(b, off));
        bad = b != 0;
    }
    if (bad)
        ls = 0;

    int const          slot_dw   = (grp / share) * (nrows * Geo::kRowDw);
    uint32_t const row_base_dw = (uint32_t)(slot_dw + g);
    int const      steps       = (ls_max + G - 1 + 3) & ~3;
    uint32_t const     lsc       = (uint32_t)max(ls, 1) - 1u;

    // trace[e][panel][g][k][word]
    uint32_t * tr = p.trace + e * p.slot_stride;
    // LDS staging block of this group, behind the profile slots
    constexpr int kStageDw = G * kTraceBlock * TW::kWords + 4; // + 4: spreads the groups over the banks
    uint32_t *    stage    = lds + ((Geo::kGroups + share - 1) / share) * (nrows * Geo::kRowDw) + grp * kStageDw;

    int best_h = 0, best_q = 0, best_s = 0;
    // target score (x4) and the best (lowest) column / its first row seen so far in this lane
    int const tgt4 = active ? 4 * p.score_in[e] : 0;
    int       kcol = 0x7fffffff, krow = 0;

    for (int panel = 0; panel < npanels; ++panel)
    {
        int const col0 = panel * Geo::kPanel + g * C;
        build_profile<G, C>(lds, slot_dw, g, q, lq, col0, sc->mat_trace, nrows, grp % share == 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        bool const use_carry_in = MULTI && is_first && (panel > 0) && (panel < my_panels) && carry != nullptr;
        bool const do_carry_out = MULTI && is_last && (panel + 1 < my_panels) && carry != nullptr;
        bool const store_trace  = active && !bad && (panel < my_panels);

        // all values are 4 x (skewed value); Z = 4 z_i
        int Z = ge4 * g;
        int Hrow[C], F1[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            Hrow[c] = Z + ge4; // "H = 0" of the previous (virtual) row
            F1[c]   = Z | 2;   // vertical gap state (tag 2) with the row's zero floor folded in
        }
        int diag0 = Z + ge4;
        int sendH = Z + ge4;
        int sendE = 4 * kNegInf + 1; // horizontal gap state, always carries tag 1

        uint32_t * trp = tr + (uint64_t)panel * p.steps_cap * (G * TW::kWords);

        auto step = [&](int k, uint32_t t)
        {
            int const        i    = k - g;
            uint32_t const * prow = lds + (row_base_dw + t * (uint32_t)Geo::kRowDw);
            uint32_t         pw[Geo::kD];
#pragma unroll
            for (int d = 0; d < Geo::kD; ++d)
                pw[d] = prow[d * G];

            int bndH = Z, bndE = 4 * kNegInf + 1;
            if constexpr (MULTI)
                if (use_carry_in && (unsigned)i < (unsigned)ls)
                {
                    bndH = carry[2 * i];
                    bndE = carry[2 * i + 1];
                }
            int const recvH = shift_from_left<G>(sendH, bndH, is_first);
            int       Ecur  = shift_from_left<G>(sendE, bndE, is_first);

            int dg = diag0;
            diag0  = recvH;

            int const ZN = Z - ge4; // zero floor of the next row (tag 0)
            int       rm = Z;       // running maximum of this row's cells
            uint32_t  w[TW::kWords];
#pragma unroll
            for (int x = 0; x < TW::kWords; ++x)
                w[x] = 0;
            int hc = 0;
            int const cl = i - (blo + bw) - col0; // BAND: first column of this strip inside the band in this row
#pragma unroll
            for (int c = 0; c < C; ++c)
            {
                int const sub = (int)(int8_t)(pw[c >> 2] >> (8 * (c & 3)));
                int const tt  = dg + sub;                        // tag 3 (in the profile)
                dg            = Hrow[c];
                int m         = max3i(tt, Ecur, F1[c]);          // E tag 1 < F tag 2 < diagonal tag 3
                LX_OPAQUE(m);
                hc            = m & ~3;
                bool inb      = true;
                if constexpr (BAND)
                {
                    inb = (unsigned)(c - cl) <= (unsigned)bw;
                    hc  = inb ? hc : Z;                          // H = 0 off the band (no walk ever gets there)
                }
                int const A0  = hc + g20v;                       // gap-open candidate, tag 0
                // F of the next row: tag 2 = extended (wins ties), 0 = opened or the next row's zero floor.  A floor
                // that wins makes that cell's H = 0, where every walk has already stopped: its flag is never read.
                int const Fr  = inb ? max3i(F1[c], A0, ZN) : ZN;
                F1[c]         = Fr | 2;
                int const Er  = inb ? max(Ecur, A0) : 4 * kNegInf; // tag 1 = extended (wins ties), 0 = opened
                Ecur          = (Er & ~3) + ge41v;
                uint32_t wc   = w[c >> 3];
                wc            = __builtin_amdgcn_alignbit((uint32_t)m, wc, 2);
                wc            = __builtin_amdgcn_alignbit((uint32_t)(Fr | Er), wc, 2); // bit 0 = E extended, bit 1 = F
                w[c >> 3]     = wc;
                if (c & 1)
                    rm = max3i(rm, Hrow[c - 1], hc); // Hrow[c-1] already holds this row's value
                else if (c == C - 1)
                    rm = max(rm, hc);
                Hrow[c] = hc;
            }
            // rare: some cell of this row reaches the extension's best score -> remember the lowest such column
            // (rows are visited in increasing order, so the first hit of a column is its lowest row)
            if (tgt4 > 0 && rm == tgt4 + Z && (unsigned)i < (unsigned)ls)
            {
#pragma unroll
                for (int c = C - 1; c >= 0; --c)
                    if (Hrow[c] == rm && col0 + c < kcol)
                    {
                        kcol = col0 + c;
                        krow = i;
                    }
            }
            sendH = hc;
            sendE = Ecur;
            Z     = Z - ge4;

            // direction words go to the LDS staging block first: [lane][word][step % 4], the layout of the trace
            {
                uint32_t * st = stage + g * (kTraceBlock * TW::kWords) + ((uint32_t)k % kTraceBlock);
#pragma unroll
                for (int x = 0; x < TW::kWords; ++x)
                    st[x * kTraceBlock] = w[x];
            }
            if constexpr (MULTI)
                if (do_carry_out && (unsigned)i < (unsigned)ls)
                {
                    carry[2 * i]     = sendH;
                    carry[2 * i + 1] = sendE;
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
LX_TRACE_UNROLL_PRAGMA // one step already holds C independent cells; unrolling steps mostly costs VGPRs (occupancy)
            for (int u = 0; u < 4; ++u)
                step(k0 + u, (tcp >> (8 * u)) & 0xffu);
            // flush the staged block of 4 steps: the group's G lanes write G consecutive 16-byte quads per store
            // instruction (whole cache lines), instead of every lane its own 12 bytes per step
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (store_trace)
            {
                uint4 * dst = reinterpret_cast<uint4 *>(trp);
#pragma unroll
                for (int x = 0; x < TW::kWords; ++x)
                {
                    uint32_t const ch = (uint32_t)(x * G + g); // staged quad ch = word ch % kWords of lane ch / kWords
                    dst[quad_index<G, TW::kWords>((uint32_t)k0 / kTraceBlock, ch / TW::kWords, ch % TW::kWords)] =
                        reinterpret_cast<uint4 const *>(stage)[ch];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }

        if (MULTI && npanels > 1)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        __builtin_amdgcn_wave_barrier();
    }

    // lowest column over the lanes of the group (every lane owns different columns), with its row
#pragma unroll
    for (int off = 1; off < G; off <<= 1)
    {
        int const  oc = __shfl_xor(kcol, off), orow = __shfl_xor(krow, off);
        bool const take = oc < kcol;
        kcol = take ? oc : kcol;
        krow = take ? orow : krow;
    }
    if (tgt4 > 0)
    {
        if (kcol == 0x7fffffff)
            bad = true; // the score of pass 1 was not reproduced: never return a wrong alignment silently
        else
        {
            best_h = tgt4 / 4;
            best_q = kcol + 1;
            best_s = krow + 1;
        }
    }

    if (in_list && is_first)
    {
        EndCell ec;
        ec.score = bad ? -1 : best_h;
        ec.q_end = best_q;
        ec.s_end = best_s;
        ec.flags = 0;
        p.ends[e] = ec;
    }
}

// One lane per extension.
template <int G, int C>
__global__ __launch_bounds__(64) void backtrace_kernel(TraceParams p)
{
    using TW        = TraceWords<C>;
    constexpr int P = G * C;
    __shared__ int8_t smat[kAlph * kAlph];
    for (int x = threadIdx.x; x < kAlph * kAlph / 4; x += blockDim.x)
        reinterpret_cast<uint32_t *>(smat)[x] = reinterpret_cast<uint32_t const *>(p.sc->mat)[x];
    __syncthreads();
    uint64_t const e     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t       limit = p.n;
    if (p.count_ptr)
    {
        uint64_t const total = *p.count_ptr;
        limit = total > p.chunk_start ? min(p.n, total - p.chunk_start) : 0;
    }
    if (e >= limit)
        return;
    uint64_t oi = e;
    if (p.src)
    {
        uint32_t const sidx = p.src[e];
        if (sidx == 0xffffffffu)
            return; // padding slot
        oi = sidx;
    }
    uint64_t const po = p.out_by_pos ? e : oi; // where this extension's record and ops slot are
    EndCell const   ec = p.ends[e];
    Extension const x  = p.ext[e];
    Hsp             out{};
    if (ec.score <= 0)
    {
        out.score = ec.score < 0 ? -1 : 0;
        p.out_hsp[po] = out;
        return;
    }
    uint32_t const * tr  = p.trace + e * p.slot_stride;
    uint8_t *        ops = p.out_ops + (p.ops_off ? p.ops_off[po] : po * p.ops_stride);
    uint32_t const   cap = x.q_len + x.s_len;

    // The walk is serial per extension and every lane of the wavefront is at a different point of its own walk, so
    // a step must cost few instructions (all branches of a step are executed by the wavefront) and few scattered memory
    // requests: the position is kept as (panel, lane g, column c in the lane) and updated incrementally; one 16-byte
    // quad of direction words serves up to four moves (a diagonal or vertical move goes one step back in the same lane);
    // residues come from two register-cached aligned 16-byte groups, the matrix from LDS; op bytes are collected into
    // whole dwords before they are stored; all bookkeeping is 32-bit offsets.
    //
    // the cell's 4 direction bits: [1:0] = source of H (3 diagonal, 2 vertical gap, 1 horizontal gap),
    // [2] = the horizontal gap of the next column EXTENDS this cell's, [3] = the vertical gap of the row below does.
    // mode F / E = "the gap character emitted last still has to be classified": this cell's bit 3 / 2 says
    // whether that gap EXTENDS this cell's gap state (then this cell is a gap cell too) or OPENED from its H.
    int      i = ec.s_end - 1, j = ec.q_end - 1;
    int      mode = 0; // 0 = H, 1 = F (vertical), 2 = E (horizontal)
    int      left = ec.score;                        // score not yet accounted for by the emitted columns
    int const ge = p.sc->ge, g2 = p.sc->g2;          // a gap of k characters costs g2 + k ge
    uint32_t n    = 0;
    int32_t  nm = 0, nx = 0, np = 0, go = 0, gx = 0;

    // position of column j in the strip geometry
    int panel = j / P;
    int g     = (j - panel * P) / C;
    int c     = j - panel * P - g * C;
    uint32_t const blocks_per_panel = p.steps_cap / kTraceBlock;

    uint32_t tq_at = 0xffffffffu; // index of the cached quad (uint4 units inside the slot)
    uint4    tq{};

    // residues: aligned 16-byte groups, addressed by 32-bit offsets from the aligned base (device buffers are 16-byte
    // aligned: never before the base)
    uintptr_t const q_al = reinterpret_cast<uintptr_t>(p.q_res + x.q_off) & ~uintptr_t(15), s_al = reinterpret_cast<uintptr_t>(p.s_res + x.s_off) & ~uintptr_t(15);
    uint32_t const  q_sh = (uint32_t)(reinterpret_cast<uintptr_t>(p.q_res + x.q_off) & 15), s_sh = (uint32_t)(reinterpret_cast<uintptr_t>(p.s_res + x.s_off) & 15);
    uint32_t        qg_at = 0xffffffffu, sg_at = 0xffffffffu;
    uint32_t        qw0 = 0, qw1 = 0, qw2 = 0, qw3 = 0, sw0 = 0, sw1 = 0, sw2 = 0, sw3 = 0;

    // ops are produced end -> begin into the slot [0, cap); `pos` is the slot offset of the next byte to write and
    // `a0` the misalignment of the slot start, so that (a0 + pos) & 3 is the byte lane inside its aligned dword
    uint32_t const  a0     = (uint32_t)(reinterpret_cast<uintptr_t>(ops) & 3);
    uint8_t * const ops_al = ops - a0;
    uint32_t        apos   = a0 + cap - 1; // a0 + pos
    uint32_t        acc    = 0;            // bytes of the dword that contains the write position, collected so far
    auto emit = [&](uint32_t op)
    {
        acc |= op << (8 * (apos & 3));
        if ((apos & 3) == 0)
        {
            // whole dword collected; only the topmost dword of the slot can reach beyond the slot end
            if (apos + 3 <= a0 + cap - 1)
                *reinterpret_cast<uint32_t *>(ops_al + apos) = acc;
            else
                for (uint32_t b = 0; b < 4 && apos + b <= a0 + cap - 1; ++b)
                    ops_al[apos + b] = (uint8_t)(acc >> (8 * b));
            acc = 0;
        }
        --apos;
        ++n;
    };
    auto step_left = [&]() // j - 1
    {
        --j;
        if (--c < 0)
        {
            c = C - 1;
            if (--g < 0)
            {
                g = G - 1;
                --panel;
            }
        }
    };

    while (i >= 0 && j >= 0 && n < cap)
    {
        // ---- this cell's nibble
        int const      xw  = c >> 3;
        uint32_t const k   = (uint32_t)(i + g);
        uint32_t const at  = (uint32_t)panel * (blocks_per_panel * G * TW::kWords) + quad_index<G, TW::kWords>(k / kTraceBlock, (uint32_t)g, (uint32_t)xw);
        if (at != tq_at)
        {
            tq_at = at;
            tq    = reinterpret_cast<uint4 const *>(tr)[at];
        }
        uint32_t const lo2  = (k & 1) ? tq.y : tq.x, hi2 = (k & 1) ? tq.w : tq.z;
        uint32_t const word = (k & 2) ? hi2 : lo2;
        int const      cnt  = (xw == TW::kWords - 1) ? (C - 8 * xw) : 8; // cells held by this word
        uint32_t const nib  = (word >> (32 - 4 * cnt + 4 * (c & 7))) & 15u;

        // one emit and one position update per iteration, whatever the move: the wavefront executes every branch some
        // lane takes, so the branches only pick small values
        uint32_t op      = 0;
        bool     up      = false, lft = false; // i - 1, j - 1
        bool     decided = false;
        if (mode == 1)
        {
            if (nib & 8u)
            {
                gx += 1;
                left -= ge;
                op      = 'D';
                up      = true;
                decided = true;
            }
            else
            {
                go += 1;
                left -= g2;
                mode = 0;
            }
        }
        else if (mode == 2)
        {
            if (nib & 4u)
            {
                gx += 1;
                left -= ge;
                op      = 'I';
                lft     = true;
                decided = true;
            }
            else
            {
                go += 1;
                left -= g2;
                mode = 0;
            }
        }
        if (!decided)
        {
            if (left <= 0)
                break; // the emitted columns add up to the score: H of this cell is 0, the alignment begins after it
            uint32_t const code = nib & 3u;
            if (code == 3)
            {
                uint32_t const qi = q_sh + (uint32_t)j, si = s_sh + (uint32_t)i;
                if ((qi >> 4) != qg_at)
                {
                    qg_at = qi >> 4;
                    uint4 const v = *reinterpret_cast<uint4 const *>(q_al + ((uintptr_t)qg_at << 4));
                    qw0 = v.x; qw1 = v.y; qw2 = v.z; qw3 = v.w;
                }
                if ((si >> 4) != sg_at)
                {
                    sg_at = si >> 4;
                    uint4 const v = *reinterpret_cast<uint4 const *>(s_al + ((uintptr_t)sg_at << 4));
                    sw0 = v.x; sw1 = v.y; sw2 = v.z; sw3 = v.w;
                }
                uint32_t const qd = (qi & 8) ? ((qi & 4) ? qw3 : qw2) : ((qi & 4) ? qw1 : qw0);
                uint32_t const sd = (si & 8) ? ((si & 4) ? sw3 : sw2) : ((si & 4) ? sw1 : sw0);
                uint32_t const c0 = (qd >> (8 * (qi & 3))) & (kAlph - 1), c1 = (sd >> (8 * (si & 3))) & (kAlph - 1);
                int const      v       = smat[c0 * kAlph + c1];
                bool const     isMatch = p.bs_match_rule ? (v == smat[c0 * kAlph + c0]) : (c0 == c1);
                nm += isMatch;
                nx += !isMatch;
                np += (v > 0);
                left -= v;
                op  = 'M';
                up  = true;
                lft = true;
            }
            else
            {
                left -= ge;
                bool const vert = code == 2;
                op   = vert ? 'D' : 'I';
                up   = vert;
                lft  = !vert;
                mode = vert ? 1 : 2;
            }
        }
        emit(op);
        i -= up ? 1 : 0;
        if (lft)
            step_left();
    }
    if (mode != 0)
        go += 1; // ran into the border right after a gap character: it can only have been an opening
    // flush the bytes collected in the lowest, partial dword
    if ((apos & 3) != 3)
        for (uint32_t b = (apos & 3) + 1; b < 4 && (apos & ~3u) + b <= a0 + cap - 1; ++b)
            ops_al[(apos & ~3u) + b] = (uint8_t)(acc >> (8 * b));

    out.score              = ec.score;
    out.q_begin            = j + 1;
    out.q_end              = ec.q_end;
    out.s_begin            = i + 1;
    out.s_end              = ec.s_end;
    out.n_ops              = (int32_t)n;
    out.num_matches        = nm;
    out.num_mismatches     = nx;
    out.num_positives      = np;
    out.num_gap_opens      = go;
    out.num_gap_extensions = gx;
    out.ops_shift          = (int32_t)(cap - n); // the ops occupy the END of the slot (written back to front)
    p.out_hsp[po]          = out;
}

__global__ void max_lens_kernel(Extension const * ext, uint64_t n, MaxLens * out)
{
    uint32_t mq = 0, ms = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        mq = max(mq, ext[i].q_len);
        ms = max(ms, ext[i].s_len);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        mq = max(mq, (uint32_t)__shfl_xor((int)mq, off));
        ms = max(ms, (uint32_t)__shfl_xor((int)ms, off));
    }
    if ((threadIdx.x & 63) == 0)
    {
        atomicMax(&out->max_q, mq);
        atomicMax(&out->max_s, ms);
    }
}

// ---- host-visible launchers ---------------------------------------------------------------------------

// trace geometries: cfg 0 = (16,10) [any query length, multi-panel], cfg 1 = (8,19) [<= 152 columns, shared profile]
// (cfg 2 = (16,13) [<= 208 columns, shared profile])
// (cfg 3 = (8,13) [<= 104 columns, shared profile]: checkpoint / single-sweep kernels only, for short queries)
// (cfg 3 = (8,13) and cfg 4 = (8,25) exist for the checkpoint kernels of lx_ckpt.hip only)
// trace cfg: 0 = (16,10), 1 = (8,19), 2 = (16,13), 3 = (8,13), 4 = (8,25), 5 = (8,11) [checkpoint kernels only: the multi-query sweep]
int trace_cfg_panel(int cfg) { return cfg == 1 ? 8 * 19 : cfg == 2 ? 16 * 13 : cfg == 3 ? 8 * 13 : cfg == 4 ? 8 * 25 : cfg == 5 ? 8 * 11 : 16 * 10; }
int trace_cfg_group(int cfg) { return (cfg == 1 || cfg == 3 || cfg == 4 || cfg == 5) ? 8 : 16; }
int trace_cfg_words(int cfg) { return cfg == 1 ? TraceWords<19>::kWords : (cfg == 2 || cfg == 3) ? TraceWords<13>::kWords : cfg == 4 ? TraceWords<25>::kWords : cfg == 5 ? TraceWords<11>::kWords : TraceWords<10>::kWords; }

template <int G, int C>
static hipError_t launch_trace_forward_cfg(TraceParams const & p, hipStream_t stream)
{
    using Geo = ScoreGeo<G, C>;
    uint64_t const blocks = (p.n + Geo::kGroups - 1) / Geo::kGroups;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    int const    share = p.shared_profile > 1 ? std::min(p.shared_profile, Geo::kGroups) : 1;
    int const    slots = (Geo::kGroups + share - 1) / share;
    size_t const lds   = ((size_t)slots * (size_t)p.nrows * Geo::kRowDw + (size_t)Geo::kGroups * (G * kTraceBlock * TraceWords<C>::kWords + 4)) *
                       sizeof(uint32_t);
    if (!p.score_in)
        return hipErrorInvalidValue; // the end cell is located through the known best score
    if (p.band > 0)
    {
        if constexpr (G == 16 && C == 10) // band mode: the generic geometry, or (8,19) with shared profiles
            hipLaunchKernelGGL((trace_forward_kernel<G, C, true, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
        else if constexpr (G == 8 && C == 19)
            hipLaunchKernelGGL((trace_forward_kernel<G, C, false, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
        else
            return hipErrorInvalidValue;
    }
    else if (p.panels_cap > 1)
        hipLaunchKernelGGL((trace_forward_kernel<G, C, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    else
        hipLaunchKernelGGL((trace_forward_kernel<G, C, false>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}

template <int G, int C>
static hipError_t launch_backtrace_cfg(TraceParams const & p, hipStream_t stream)
{
    uint64_t const b2 = (p.n + 63) / 64;
    hipLaunchKernelGGL((backtrace_kernel<G, C>), dim3((unsigned)b2), dim3(64), 0, stream, p);
    return hipGetLastError();
}

// The two halves of pass 2 are launched separately so that the host side can run the backtrace of chunk k on a second
// stream while the forward kernel of chunk k+1 already runs (double-buffered trace workspace).
hipError_t launch_trace_forward(TraceParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    return p.cfg == 1   ? launch_trace_forward_cfg<8, 19>(p, stream)
           : p.cfg == 2 ? launch_trace_forward_cfg<16, 13>(p, stream)
                        : launch_trace_forward_cfg<16, 10>(p, stream);
}

hipError_t launch_backtrace(TraceParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    return p.cfg == 1   ? launch_backtrace_cfg<8, 19>(p, stream)
           : p.cfg == 2 ? launch_backtrace_cfg<16, 13>(p, stream)
                        : launch_backtrace_cfg<16, 10>(p, stream);
}

hipError_t launch_max_lens(Extension const * ext, uint64_t n, MaxLens * out, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(MaxLens), stream);
    if (e != hipSuccess || n == 0)
        return e;
    unsigned const blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(max_lens_kernel, dim3(blocks), dim3(256), 0, stream, ext, n, out);
    return hipGetLastError();
}

} // namespace lx

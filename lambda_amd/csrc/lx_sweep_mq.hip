// lx_sweep_mq.hip -- the single sweep for RAGGED seed lists: up to four queries per wavefront, byte profiles (gfx950 only).
//
// Same reference seam as lx_score_f16.hip / lx_score_i16.hip (_performAlignment, /root/reference/src/search_algo.hpp:1070-1134:
// pass 1 at :1246 and the forward half of pass 2 at :1296 as one sweep), same strip-systolic mapping, same row-skewed
// recurrence on packed 16-bit integers whose maxima run through the half-precision comparators (lx_score_i16.hip's header).
// What is new is the unit a wavefront serves.  The list lambda hands over after _widenAndPreprocessMatches (:1136-1175) has
// a dozen windows per query on average, a tenth of them merged ones of up to three times the length: a wavefront that must
// hold 16 windows of ONE query (the LDS profile is the query's) runs half empty and as long as its longest window.  Here a
// wavefront's eight lane groups are dealt to `share` = 1, 2, 4 or 8 sub-blocks of 16 / 8 / 4 / 2 windows, each with its own
// query and its own profile slot, so that the host can sort sub-blocks by length across queries (lx_host.cpp).
//
// That needs profiles a quarter the size -- four of them must fit next to each other at full occupancy:
//   * a profile entry is ONE BYTE, the non-negative integer  s(q, t) - ge - (go - ge) = s(q, t) - go  (go = cost of a gap's
//     first character: applicable when no substitution costs more than opening a gap, which every BLOSUM / nucleotide /
//     bisulfite scheme of the reference satisfies with its default gap costs); 152 columns x 28 letters = 4.4 KB instead of
//     10.5 KB.  One v_perm_b32 per column -- the instruction that interleaved the two extensions' halves before -- now picks
//     byte k of the two letters' rows and zero-extends both;
//   * the extra (go - ge) is not subtracted again: the diagonal operand of the next row is the gap-open candidate
//     A = H + (go - ge), which the recurrence computes anyway, so the lanes keep A[row - 1][c] instead of H[row - 1][c] and
//     hand A of their last column to the right.  7.5 packed instructions per two cells, as before;
//   * pad letters (rows beyond the window, columns beyond the query) have entry 0, i.e. score like a gap's first character:
//     such a cell is never above its neighbours, so it can neither start, nor extend, nor end a best local alignment.
// Two refinements of the unit (DESIGN_LOG.md section 3.3):
//   * free packing (pair_share = 1, LX_OPT_QUERY_RUN = 2): what the LDS limits is the number of profiles per wavefront (four),
//     not how the lane groups are dealt to them -- a lane group's two windows share a query, the eight lane groups hold windows
//     of up to four queries in any split, numbered in order of appearance;
//   * narrow last panel (ScoreParams::narrow): the panel body exists for C, (3 C + 3) / 4, (C + 1) / 2 and (C + 3) / 4 columns per lane; a
//     wavefront runs a panel with the narrowest width that covers what any of its queries has left there, and records the width
//     of an extension's last panel in its end cell (kEndNarrowShift) for the backtrace.
// Slots are the compact 16-bit codes of Ckpt16Layout (one part per panel); an extension whose best score is beyond 2046
// leaves the sentinel and is redone by the int32 launch into an overflow slot, like a wavefront the range test declines.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "lx_dp_common.h"

namespace lx
{
namespace
{

typedef unsigned short q2 __attribute__((ext_vector_type(2)));
typedef _Float16       qf2 __attribute__((ext_vector_type(2)));

// integer maxima of patterns in 0 .. 0x7BFF through the half-precision comparators (lx_score_i16.hip)
__device__ __forceinline__ q2 qmax(q2 a, q2 b)
{
    return __builtin_bit_cast(q2, __builtin_elementwise_maximum(__builtin_bit_cast(qf2, a), __builtin_bit_cast(qf2, b)));
}
__device__ __forceinline__ q2 qmax3(q2 a, q2 b, q2 c)
{
    return __builtin_bit_cast(q2, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(qf2, a), __builtin_bit_cast(qf2, b)),
                                                                __builtin_bit_cast(qf2, c))); // v_pk_maximum3_f16
}
__device__ __forceinline__ q2 qsplat(int x) { return q2{(unsigned short)x, (unsigned short)x}; }
__device__ __forceinline__ q2 as_q2(uint32_t x) { return __builtin_bit_cast(q2, x); }
__device__ __forceinline__ uint32_t qbits(q2 x) { return __builtin_bit_cast(uint32_t, x); }

constexpr int kMqBias  = 2048;             // keeps every finite skewed value positive
constexpr int kMqLimit = 0x7BFF - kMqBias; // every finite intermediate stays below this

template <int C>
struct MqGeo
{
    static constexpr int G       = 8;
    static constexpr int kGroups = 8;           // lane groups per wavefront, two extensions each
    static constexpr int kPanel  = G * C;
    static constexpr int kD      = (C + 3) / 4; // profile dwords (4 byte entries each) per lane and letter
    static constexpr int kRowDw  = kD * G;
    // a lane's dwords of a row: one 16-byte piece, one 8-byte piece, one 4-byte piece (whichever kD needs), each piece
    // lane-contiguous so that it is read with one aligned ds_read_b128 / b64 / b32
    static constexpr int kN4 = kD / 4, kN2 = (kD % 4) / 2, kN1 = kD % 2;
    static constexpr int kBase2 = 4 * G * kN4, kBase1 = kBase2 + 2 * G * kN2;
    static_assert(kD <= 7, "at most 28 columns per strip");
    // int16-pair slots (the layout of lx_ckpt.hip's CkptLayout<G, C>, what the int32 kernel writes): WIDE sweeps
    static constexpr int kCkDwW = (C + 3) / 4 * 4;
    __host__ __device__ static constexpr uint64_t bnd_dwords_w(uint32_t steps_cap) { return (uint64_t)steps_cap * G; }
    __host__ __device__ static constexpr uint64_t slot_dwords_w(uint32_t steps_cap) { return bnd_dwords_w(steps_cap) + (uint64_t)(steps_cap / 16) * G * kCkDwW; }
    __host__ __device__ static constexpr int dw_index(int d, int g)
    {
        return d < 4 * kN4 ? g * 4 + d : d < 4 * kN4 + 2 * kN2 ? kBase2 + g * 2 + (d - 4 * kN4) : kBase1 + g;
    }
};

} // namespace

// MULTI: queries wider than one panel -- the panels are swept one after the other, the (A, E) pair of the last strip per
// subject row goes through lx_score.hip's carry workspace, every panel writes its own part of the extension's slot.
// WIDE: the slots hold int16 pairs (the int32 kernel's layout) instead of the compact codes: scores up to the 16-bit patterns'
// range (29 695) at sweep speed -- long queries with strong hits, whose windows score beyond the codes' 2046 (a 600-residue
// query against its homologue: ~3 000) and would otherwise be redone one by one by the int32 launch.
#ifndef LX_MQ_WAVES
#define LX_MQ_WAVES 2 // wavefronts per SIMD the register budget is set for (four byte profiles + staging = 20 KB of LDS admit two; 3 was
                      // measured with ONE profile per wavefront, where the LDS admits it: DESIGN_LOG.md section 3.3)
#endif
template <int C, bool MULTI, bool WIDE = false>
__global__ __launch_bounds__(64, LX_MQ_WAVES) void sweep_mq_kernel(ScoreParams p)
{
    using Geo = MqGeo<C>;
    using L16 = Ckpt16Layout<8, C>;
    constexpr int G = 8;
    // (steps unrolled by 2 in the chunk loops: removes the register-rotation moves.  MULTI indexes its per-chunk carry registers
    // by the step inside the chunk, which must be a constant -- a dynamic index sends the arrays to scratch memory --, so its
    // four steps are written out, with a scheduling barrier in the middle that keeps two steps' loads in flight, not four)
    extern __shared__ uint32_t lds[];

    int const  lane     = threadIdx.x;
    int const  grp      = lane / G;
    int const  g        = lane % G;
    bool const is_first = (g == 0);
    bool const is_last  = (g == G - 1);

    uint64_t const wf   = (uint64_t)blockIdx.x + p.wf_lo; // (wavefront of the chunk)
    uint64_t const pair = wf * Geo::kGroups + grp;
    uint64_t const eA   = 2 * pair, eB = 2 * pair + 1;
    bool const     actA = eA < p.n, actB = eB < p.n;

    // ... or the slots of this wavefront by themselves (ScoreParams::wf_tab: sized for its own longest window and widest query)
    bool const       by_wf      = p.wf_tab != nullptr;
    WfSlots const    wfs        = by_wf ? p.wf_tab[wf] : WfSlots{0, 0, 0};
    uint32_t const   steps_cap  = by_wf ? wfs.steps_cap : p.steps_cap, panels_cap = by_wf ? wfs.panels_cap : p.panels_cap;
    uint32_t * const ckpt_base  = by_wf ? p.ckpt + wfs.off_dw : p.ckpt;
    uint64_t const   ckpt_step  = by_wf ? (uint64_t)wfs.panels_cap * (WIDE ? Geo::slot_dwords_w(wfs.steps_cap) : L16::slot_dwords(wfs.steps_cap))
                                        : p.ckpt_stride;
    uint64_t const   ckpt_first = by_wf ? 2 * Geo::kGroups * wf : 0;

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
    // p.solo: the SOLO packing (LX_OPT_QUERY_RUN = 1) -- no promise at all: every window has a query and an LDS profile of its own,
    // 16 per wavefront.  For the alphabets whose byte profiles are small enough (nucleotides: 5 letters + pad): a read set's seed
    // list has one or two windows per read, and four queries per wavefront would leave three lanes in four to fillers.
    bool const            solo_mode = p.solo != 0;
    uint8_t const * const qB        = p.q_res + q_offB;
    // p.pair_share lane groups (0 = all eight) use one LDS profile: the extensions of such a sub-block share the query.
    // p.pair_share == 1 is the FREE packing: the two windows of a lane group share a query, the eight lane groups hold windows
    // of at most kFreeSlots queries in any split (5 + 2 + 1, ...); the queries get their profile slots in order of appearance.
    constexpr int kFreeSlots = 4;
    bool const free_mode = !solo_mode && p.pair_share == 1;
    int const  share_g   = free_mode ? Geo::kGroups : (p.pair_share > 0 && p.pair_share < Geo::kGroups) ? p.pair_share : Geo::kGroups;
    int        blk       = grp / share_g;
    // free packing: the wavefront's queries (uniform values) and this lane group's slot among them
    uint64_t fq[kFreeSlots] = {0, 0, 0, 0};
    int      fl[kFreeSlots] = {0, 0, 0, 0};
    int      nfree = 0;
    bool     too_many = false;
    if (free_mode)
    {
        if (actA && actB && (q_offB != q_off || lqB != lq)) // the pair promise
            atomicExch(p.err, 2);
        int mine = 0;
#pragma unroll
        for (int k = 0; k < Geo::kGroups; ++k)
        {
            int const      lead = k * G;
            uint64_t const uq   = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(q_off >> 32), lead) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)q_off, lead);
            int const ul = __builtin_amdgcn_readlane(lq, lead);
            if (__builtin_amdgcn_readlane(actA ? 1 : 0, lead) == 0)
                continue; // (uniform) a lane group beyond the list: no query of its own
            int j = -1;
#pragma unroll
            for (int t = 0; t < kFreeSlots; ++t)
                if (t < nfree && fq[t] == uq && fl[t] == ul)
                    j = t;
            if (j < 0)
            {
                if (nfree < kFreeSlots)
                {
#pragma unroll
                    for (int t = 0; t < kFreeSlots; ++t)
                        if (t == nfree)
                        {
                            fq[t] = uq;
                            fl[t] = ul;
                        }
                    j = nfree++;
                }
                else
                {
                    too_many = true; // a fifth query: the promise is broken -- reported, the wavefront left to the int32 launch
                    j        = 0;
                }
            }
            mine = grp == k ? j : mine;
        }
        blk = mine;
        if (too_many && lane == 0)
            atomicExch(p.err, 2);
    }
    else if (!solo_mode)
    {
        // the caller promised one query per sub-block: verify against the sub-block's first lane, fail loudly otherwise
        int const      leader  = blk * share_g * G;
        uint64_t const q0      = ((uint64_t)(uint32_t)__shfl((int)(q_off >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)q_off, leader);
        int const      l0      = __shfl(lq, leader);
        bool const     lead_in = __shfl(actA ? 1 : 0, leader) != 0;
        if (lead_in && ((actA && (q_off != q0 || lq != l0)) || (actB && (q_offB != q0 || lqB != l0))))
            atomicExch(p.err, 2);
    }

    int ls_max = max(lsA, lsB);
    int ls_min = min(actA ? lsA : 0x7fffffff, actB ? lsB : 0x7fffffff);
    int lq_max = max(lq, actB ? lqB : 0);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        ls_max = max(ls_max, __shfl_xor(ls_max, off));
        ls_min = min(ls_min, __shfl_xor(ls_min, off));
        lq_max = max(lq_max, __shfl_xor(lq_max, off));
    }
    ls_max = __builtin_amdgcn_readfirstlane(ls_max);
    ls_min = __builtin_amdgcn_readfirstlane(ls_min);
    lq_max = __builtin_amdgcn_readfirstlane(lq_max);
    int const steps   = (ls_max + G - 1 + 3) & ~3;
    // the sub-blocks of a wavefront may differ in width (the host deals them by geometry class, so they rarely do): all of
    // them sweep as many panels as the widest needs, the narrower ones over pad columns
    int const npanels = MULTI ? max(1, (lq_max + Geo::kPanel - 1) / Geo::kPanel) : 1;

    // ---- range test (wave-uniform): an upper bound of every finite intermediate must stay below the limit -- the codes'
    // 2046 for one panel (tested up front), the 16-bit patterns' for wider queries (their codes are tested afterwards)
    // (first the bound no query of this length exceeds -- every column at the matrix' largest entry: where that passes, the
    // residues need not be looked at: two dependent loads per column, 40 per lane in front of everything else)
    int        bound       = lq_max * sc->smax;
    bool const quick_fits  = bound + (-ge) * (steps + G + 2) + sc->smax + 2 <= ((MULTI || WIDE) ? kMqLimit : 2046);
    if (!quick_fits)
    {
        bound = 0;
        for (int j = g; j < lq; j += G) // (every column of the query once, whichever strip sweeps it)
            bound += sc->rowmax[q[j] & (kAlph - 1)];
        int boundB = 0;
        if (solo_mode && actB)
            for (int j = g; j < lqB; j += G)
                boundB += sc->rowmax[qB[j] & (kAlph - 1)];
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1)
        {
            bound += __shfl_xor(bound, off);
            boundB += __shfl_xor(boundB, off);
        }
        bound = max(bound, boundB);
    }
    bool const broken  = (lq_max > (MULTI ? (int)panels_cap : 1) * Geo::kPanel) || (uint32_t)steps > steps_cap;
    // (free packing with a fifth query: declined as a whole -- the int32 launch shares profiles by pairs and copes)
    bool const too_big = broken || too_many || __ballot(bound + (-ge) * (steps + G + 2) + sc->smax + 2 > ((MULTI || WIDE) ? kMqLimit : 2046)) != 0 ||
                         (-ge) * (G + 2) + (-sc->g2) * 2 + 256 > kMqBias;
    if (too_big)
    {
        // left to the int32 launch (TraceParams::fixup): sentinel -1; a broken length promise is reported here as well,
        // because that launch is skipped when no query the promise admits can fail the test
        if (broken && lane == 0)
            atomicExch(p.err, 3);
        if (is_first)
        {
            EndCell none{};
            none.score = -1;
            if (actA)
            {
                p.out_score[eA] = -1;
                p.ends[eA]      = none;
            }
            if (actB)
            {
                p.out_score[eB] = -1;
                p.ends[eB]      = none;
            }
            if (p.stat_beyond && actA)
                atomicAdd(p.stat_beyond, actB ? 2u : 1u);
        }
        return;
    }

    // carry workspace for multi-panel queries: one (packed A, packed E) pair per subject row and lane group
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
    bool const writable = !MULTI || npanels == 1 || carry != nullptr; // (workspace exhausted: reported, nothing kept)

    constexpr uint32_t kRowBytes = Geo::kRowDw * 4;
    int const          nslots    = solo_mode ? 2 * Geo::kGroups : free_mode ? kFreeSlots : Geo::kGroups / share_g;
    // (solo: slots 2 grp and 2 grp + 1; a lane group whose second half is beyond the list reads its first half's rows -- pad letters)
    uint32_t const     slot_dw   = (uint32_t)(solo_mode ? 2 * grp : blk) * (uint32_t)(nrows * Geo::kRowDw);
    uint32_t const     slot_dwB  = (solo_mode && actB) ? slot_dw + (uint32_t)(nrows * Geo::kRowDw) : slot_dw;
    uint32_t const     slot_byte = slot_dw * 4u, slot_byteB = slot_dwB * 4u;
    uint64_t const     panel_dw  = WIDE ? Geo::slot_dwords_w(steps_cap) : L16::slot_dwords(steps_cap);
    uint32_t * const   stage     = lds + nslots * (nrows * Geo::kRowDw) + lane; // [step % 8][lane]: lane-minor, conflict-free

    q2 const GE = qsplat(ge), G2 = qsplat(sc->g2), NGE = qsplat(-ge);
    // over the panels swept so far, per extension: best strip value, its (global) strip, first row, "met again later"
    int runA = 0, stripA = 0, rrowA = 0, rtieA = 0, runB = 0, stripB = 0, rrowB = 0, rtieB = 0;

    // strip width of this lane group's LAST panel as swept (code of kEndNarrowShift), for the backtrace
    int const my_panels = max(1, (lq + Geo::kPanel - 1) / Geo::kPanel), my_panelsB = max(1, (lqB + Geo::kPanel - 1) / Geo::kPanel);
    int       my_code   = 0, my_codeB = 0;

    for (int panel = 0; panel < npanels; ++panel)
    {
        // The panel's strip width, one for the wavefront: full width while any of its queries goes on behind this panel, else
        // the narrowest of C, (3 C + 3) / 4, (C + 1) / 2, (C + 3) / 4 columns per lane that covers what its queries have left.
        int code = 0;
        if (p.narrow)
        {
            int const rem  = lq - panel * Geo::kPanel;
            int       mine = panel < my_panels - 1 ? 0 : rem >= 1 ? narrow_code_for(C, G, rem) : kNarrowest; // (no column left: any width)
            mine           = actA ? mine : kNarrowest;
            if (solo_mode && actB)
            {
                int const remB = lqB - panel * Geo::kPanel;
                mine           = min(mine, panel < my_panelsB - 1 ? 0 : remB >= 1 ? narrow_code_for(C, G, remB) : kNarrowest);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1)
                mine = min(mine, __shfl_xor(mine, off));
            code = __builtin_amdgcn_readfirstlane(mine);
        }
        if (panel == my_panels - 1)
            my_code = code;
        if (panel == my_panelsB - 1)
            my_codeB = code;
        auto panel_body = [&](auto ce_tag)
        {
        constexpr int CE   = decltype(ce_tag)::value; // columns per lane in this panel
        int const     col0 = panel * Geo::kPanel + g * CE;
        // ---- profile: prof[slot][t][piece][g] = bytes (s(q_col, t) - go) of the lane's columns, four per dword.  The lane
        // groups of a sub-block share the work: group r writes the letters 4w .. 4w+3 with w % share == r.  The 1 KB table is
        // copied into the (still idle) staging area first, so that the per-column row reads are LDS reads -- two dependent
        // global loads per four columns made a wavefront's profiles cost 5 % of a panel.
        {
            __builtin_amdgcn_wave_barrier(); // (the previous panel's staged codes have left)
            uint32_t * const tab = lds + nslots * (nrows * Geo::kRowDw);
            reinterpret_cast<uint4 *>(tab)[lane] = reinterpret_cast<uint4 const *>(sc->mat_b8)[lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // one profile: query qp of length lqp into the slot at dword sdw; `share` lane groups write it, this one is number r
            auto build_profile = [&](uint8_t const * qp, int lqp, uint32_t sdw, int share, int r)
            {
                auto letters = [&](int d) -> uint32_t // the query letters of columns 4d .. 4d+3 of this lane's strip, one per byte
                {
                    int const j0 = col0 + 4 * d;
                    uint32_t  w  = 0x1f1f1f1fu;
                    if (j0 < lqp)
                        w = *reinterpret_cast<unaligned_u32 const *>(qp + j0); // (the residue buffers carry slack behind their end)
                    return w;
                };
                uint32_t wcur = letters(0);
#pragma unroll 1
                for (int d = 0; d < (CE + 3) / 4; ++d)
                {
                    uint32_t const wnext = d + 1 < (CE + 3) / 4 ? letters(d + 1) : 0u;
                    uint32_t       rows[4][8];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
                    {
                        int const c  = 4 * d + cc;
                        int const j  = col0 + c;
                        uint32_t  ql = kAlph - 1; // pad rank: a row of zeros
                        if (c < CE && j < lqp)
                            ql = (wcur >> (8 * cc)) & (kAlph - 1);
                        uint4 const * mrow = reinterpret_cast<uint4 const *>(reinterpret_cast<uint8_t const *>(tab) + ql * kAlph);
                        uint4 const   lo = mrow[0], hi = mrow[1];
                        rows[cc][0] = lo.x; rows[cc][1] = lo.y; rows[cc][2] = lo.z; rows[cc][3] = lo.w;
                        rows[cc][4] = hi.x; rows[cc][5] = hi.y; rows[cc][6] = hi.z; rows[cc][7] = hi.w;
                    }
                    wcur = wnext;
                    uint32_t * dst = lds + sdw + Geo::dw_index(d, g);
#pragma unroll
                    for (int w = 0; w < 8; ++w)
                    {
                        if (4 * w < nrows && (w % share) == r)
                        {
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                            {
                                // byte b of the matrix rows of columns 0..3 -> one dword [c0,c1,c2,c3] for subject letter 4w+b
                                uint32_t const sel = (uint32_t)b | ((uint32_t)(4 + b) << 8) | 0x0c0c0000u;
                                uint32_t const x01 = __builtin_amdgcn_perm(rows[1][w], rows[0][w], sel);
                                uint32_t const x23 = __builtin_amdgcn_perm(rows[3][w], rows[2][w], sel);
                                if (4 * w + b < nrows) // (the solo packing's slots end with the pad letter's row)
                                    dst[(4 * w + b) * Geo::kRowDw] = x01 | (x23 << 16);
                            }
                        }
                    }
                }
            };
            if (free_mode)
            {
                // every query of the wavefront in turn, all eight lane groups on each (group r writes the letters 4r .. 4r+3)
#pragma unroll 1
                for (int j = 0; j < nfree; ++j)
                {
                    uint64_t const uq = j == 0 ? fq[0] : j == 1 ? fq[1] : j == 2 ? fq[2] : fq[3];
                    int const      ul = j == 0 ? fl[0] : j == 1 ? fl[1] : j == 2 ? fl[2] : fl[3];
                    build_profile(p.q_res + uq, ul, (uint32_t)j * (uint32_t)(nrows * Geo::kRowDw), Geo::kGroups, grp);
                }
            }
            else if (solo_mode)
            {
                // every lane group its own two
                build_profile(q, lq, slot_dw, 1, 0);
                if (actB)
                    build_profile(qB, lqB, slot_dwB, 1, 0);
            }
            else
                build_profile(q, lq, slot_dw, share_g, grp % share_g);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        bool const use_carry_in = MULTI && is_first && panel > 0 && carry != nullptr;
        bool const do_carry_out = MULTI && is_last && panel + 1 < npanels && carry != nullptr;
        // (an idle half owns the spare slot p.n -- the stores are unconditional)
        // (... or, with slots by wavefront, its own slot of the wavefront's sixteen)
        uint32_t * const slotA = ckpt_base + (((actA || by_wf) ? eA : p.n) - ckpt_first) * ckpt_step + (uint64_t)panel * panel_dw;
        uint32_t * const slotB = ckpt_base + (((actB || by_wf) ? eB : p.n) - ckpt_first) * ckpt_step + (uint64_t)panel * panel_dw;

        q2 Z = qsplat(ge * g + kMqBias); // z_i of the first processed row i = -g, biased
        q2 Arow[C], F0[C];               // A = H + (go - ge) of the previous row (its frame), folded F of this row
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            Arow[c] = Z + GE + G2;
            F0[c]   = Z;
        }
        q2 diag0 = Z + GE + G2;
        q2 sendA = Z + GE + G2;
        q2 sendE = as_q2(0u);
        q2 best  = qsplat(0);
        // first row that reached this strip's best value, "a later row reached it again" (bit 0 = A, bit 1 = B)
        int      rowA = 0, rowB = 0;
        uint32_t tie  = 0;
        q2       cmax = qsplat(0); // best un-skewed (and unbiased: >= 0) row maximum of the current chunk

        // MULTI: the left boundary (A, E) of this chunk's four rows -- the previous panel's last column where there is one, the
        // rectangle's edge (z + (go - ge), -inf) elsewhere -- made one chunk ahead, so that a step takes it from registers
        // without a branch (a load issued in the step that needs it leaves its whole latency exposed; a lane-dependent branch
        // per step costs a fifth of the step)
        uint32_t cinA[4] = {0, 0, 0, 0}, cinE[4] = {0, 0, 0, 0}, ninA[4] = {0, 0, 0, 0}, ninE[4] = {0, 0, 0, 0};
        uint32_t outA[4] = {0, 0, 0, 0}, outE[4] = {0, 0, 0, 0}; // (A, E) this lane sent in the chunk's four steps
        auto carry_fetch = [&](int k0, q2 zf, uint32_t (&a)[4], uint32_t (&e)[4]) // zf = Z of step k0
        {
            if constexpr (MULTI)
            {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    a[u] = qbits(zf + G2);
                    e[u] = 0u;
                    zf   = zf + NGE;
                }
                if (use_carry_in)
                {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                    {
                        int const i = k0 + u - g;
                        if ((unsigned)i < (unsigned)ls_pair)
                        {
                            uint2 const v = *reinterpret_cast<uint2 const *>(carry + 2 * i);
                            a[u]          = v.x;
                            e[u]          = v.y;
                        }
                    }
                }
            }
        };
        // one DP step at step index k = k0 + u (row k - g of this lane's strip)
        // the profile rows of one step's two letters (LDS), and the step on them: the rows of step u + 1 are fetched before step u is
        // computed (four_steps below) -- issued where they are used, a step began with the LDS latency in the open, four times a chunk
        auto rows = [&](uint32_t tA, uint32_t tB, uint32_t (&pa)[Geo::kD], uint32_t (&pb)[Geo::kD])
        {
            char const * const ra = reinterpret_cast<char const *>(lds) + slot_byte + tA * kRowBytes;
            char const * const rb = reinterpret_cast<char const *>(lds) + slot_byteB + tB * kRowBytes;
            if constexpr (Geo::kN4 != 0)
            {
                uint4 const va = *reinterpret_cast<uint4 const *>(ra + g * 16), vb = *reinterpret_cast<uint4 const *>(rb + g * 16);
                pa[0] = va.x; pa[1] = va.y; pa[2] = va.z; pa[3] = va.w;
                pb[0] = vb.x; pb[1] = vb.y; pb[2] = vb.z; pb[3] = vb.w;
            }
            if constexpr (Geo::kN2 != 0)
            {
                uint2 const va = *reinterpret_cast<uint2 const *>(ra + Geo::kBase2 * 4 + g * 8), vb = *reinterpret_cast<uint2 const *>(rb + Geo::kBase2 * 4 + g * 8);
                pa[4 * Geo::kN4] = va.x; pa[4 * Geo::kN4 + 1] = va.y;
                pb[4 * Geo::kN4] = vb.x; pb[4 * Geo::kN4 + 1] = vb.y;
            }
            if constexpr (Geo::kN1 != 0)
            {
                pa[Geo::kD - 1] = *reinterpret_cast<uint32_t const *>(ra + Geo::kBase1 * 4 + g * 4);
                pb[Geo::kD - 1] = *reinterpret_cast<uint32_t const *>(rb + Geo::kBase1 * 4 + g * 4);
            }
        };
        auto step = [&](uint32_t const (&pa)[Geo::kD], uint32_t const (&pb)[Geo::kD], int k, int u)
        {
            // left boundary: H[i][-1] = 0 (skewed: z; as A: z + (go - ge)), E = -inf; or the previous panel's last column
            uint32_t bndA = qbits(Z + G2), bndE = 0u;
            if constexpr (MULTI)
            {
                bndA = cinA[u];
                bndE = cinE[u];
            }
            q2 const recvA = as_q2((uint32_t)shift_from_left<G>((int)qbits(sendA), (int)bndA, is_first));
            q2       Ecur  = as_q2((uint32_t)shift_from_left<G>((int)qbits(sendE), (int)bndE, is_first));
            q2       dg    = diag0;
            diag0          = recvA;

            q2 const ZN     = Z + NGE;
            q2       rowmax = as_q2(0u);
            q2       h      = Z, hprev = Z, A = Z;
#pragma unroll
            for (int c = 0; c < CE; ++c)
            {
                // (entry of column c vs letter tA | entry of column c vs letter tB), zero-extended bytes
                uint32_t const sel = 0x0c000c00u | (uint32_t)(c & 3) | ((uint32_t)(4 + (c & 3)) << 16);
                q2 const       sub = as_q2(__builtin_amdgcn_perm(pb[c >> 2], pa[c >> 2], sel));
                q2 const       tt  = dg + sub;
                dg                 = Arow[c];
                hprev              = h;
                h                  = qmax3(tt, Ecur, F0[c]);
                A                  = h + G2;
                F0[c]              = qmax3(F0[c], A, ZN);
                Ecur               = qmax(Ecur, A) + GE;
                Arow[c]            = A;
                if (c & 1)
                    rowmax = qmax3(rowmax, hprev, h);
            }
            if (CE & 1)
                rowmax = qmax(rowmax, h);
            sendA = A;
            sendE = Ecur;
            if constexpr (MULTI)
            {
                outA[u] = qbits(sendA); // (stored by chunk_done: one branch per four steps)
                outE[u] = qbits(sendE);
            }
            cmax = qmax(cmax, rowmax - Z); // the rose / met-again logic runs once per block of sixteen steps (book)
            // un-skewed boundary pair (H of the strip's last column, E as the next strip's first column uses it) as
            // Ckpt16Layout codes of both extensions: H | (H - E) << 11
            if constexpr (WIDE)
            {
                // ... as int16 pairs (H, E), re-paired per extension
                q2 const hb = h - Z, eb = Ecur - Z;
                stage[u * 64]       = __builtin_amdgcn_perm(qbits(eb), qbits(hb), 0x05040100u);
                stage[(4 + u) * 64] = __builtin_amdgcn_perm(qbits(eb), qbits(hb), 0x07060302u);
            }
            else
                stage[((k & 4) + u) * 64] = (qbits(h - Ecur) << 11) | qbits(h - Z);
            Z = ZN;
        };
        // the staged codes of the eight steps up to k0 + 3 leave, re-paired per extension (whole 128-byte lines per lane
        // group, no branch around the stores)
        auto flush_codes = [&](int k0)
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
        };
        // the row checkpoint behind step k0 + 3 (k0 % 16 == 12), issued at the top of the next chunk
        auto rowck_codes = [&](int k0)
        {
            // Arow - (go - ge) = H is in the frame of the row just processed (z_i = Z + ge after the update), F0 in the next
            // row's:  H - F un-skewed = (H - z_i) - (F0 - Z) = H - F0 - ge
            q2 const ziA = Z + GE + G2, geA = GE + G2;
            if constexpr (WIDE)
            {
                // int16 pairs (H, folded F), one dword per column and extension (lx_ckpt.hip: rowck_quad_index, lane-major)
                uint64_t const base = Geo::bnd_dwords_w(steps_cap) / 4 + ((uint64_t)((k0 + 3) / 16) * G + (uint64_t)g) * (Geo::kCkDwW / 4);
                uint4 * const  dA = reinterpret_cast<uint4 *>(slotA) + base, * const dB = reinterpret_cast<uint4 *>(slotB) + base;
#pragma unroll
                for (int x = 0; x < Geo::kCkDwW / 4; ++x)
                {
                    uint32_t wa[4], wb[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                    {
                        int const c = 4 * x + b;
                        if (c < CE)
                        {
                            q2 const hu = Arow[c < CE ? c : 0] - ziA, fu = F0[c < CE ? c : 0] - Z;
                            wa[b] = __builtin_amdgcn_perm(qbits(fu), qbits(hu), 0x05040100u);
                            wb[b] = __builtin_amdgcn_perm(qbits(fu), qbits(hu), 0x07060302u);
                        }
                        else
                            wa[b] = wb[b] = 0;
                    }
                    dA[x] = make_uint4(wa[0], wa[1], wa[2], wa[3]);
                    dB[x] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
                }
                return;
            }
            uint32_t code[2 * L16::kCkDw];
#pragma unroll
            for (int c = 0; c < 2 * L16::kCkDw; ++c)
                code[c] = c < CE ? ((qbits((Arow[c < CE ? c : 0] - F0[c < CE ? c : 0]) - geA) << 11) | qbits(Arow[c < CE ? c : 0] - ziA)) : 0u;
            uint32_t const base = (uint32_t)(L16::bnd_dwords(steps_cap) / 4);
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
        };
        // the best-value bookkeeping, once per block of sixteen steps (lx_score_f16.hip: book)
        auto book = [&](int k0) // k0: the first of the block's last four steps
        {
            // per half: did the strip's best rise in this block (then its first row is one of the block's sixteen; the
            // backtrace finds it), or was it only met again (a tie for the end cell)?  Rows beyond the window and columns
            // beyond the query stay strictly below a positive best: no validity test.
            q2 const       nb   = qmax(best, cmax);
            uint32_t const rose = qbits(nb) ^ qbits(best), met = qbits(cmax) ^ qbits(best);
            bool const     gtA = (rose & 0xffffu) != 0, gtB = (rose >> 16) != 0;
            bool const     eqA = (met & 0xffffu) == 0, eqB = (met >> 16) == 0;
            int const      last = k0 + 3 - g; // the block's last row in this lane
            rowA = gtA ? last : rowA;
            rowB = gtB ? last : rowB;
            tie  = (gtA ? (tie & ~1u) : (tie | (eqA ? 1u : 0u)));
            tie  = (gtB ? (tie & ~2u) : (tie | (eqB ? 2u : 0u)));
            best = nb;
            cmax = qsplat(0);
        };
        auto chunk_done = [&](int k0)
        {
            if constexpr (WIDE)
            {
                // the four steps' boundary pairs leave as one quad per extension (lx_ckpt.hip: bnd_quad_index)
                uint32_t const qi = ((uint32_t)k0 / 4) * G + (uint32_t)g;
                reinterpret_cast<uint4 *>(slotA)[qi] = make_uint4(stage[0], stage[64], stage[128], stage[192]);
                reinterpret_cast<uint4 *>(slotB)[qi] = make_uint4(stage[256], stage[320], stage[384], stage[448]);
            }
            else if (k0 & 4)
                flush_codes(k0);
            if constexpr (MULTI)
            {
                if (do_carry_out)
                {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                    {
                        int const i = k0 + u - g;
                        if ((unsigned)i < (unsigned)ls_pair)
                            *reinterpret_cast<uint2 *>(carry + 2 * i) = make_uint2(outA[u], outE[u]);
                    }
                }
            }
        };

        uint32_t const lscA = (uint32_t)max(lsA, 1) - 1u, lscB = (uint32_t)max(lsB, 1) - 1u;
        // the subject letters of rows k0 - g ... k0 - g + 3 of both windows where some of these rows may lie outside a window (the
        // ramps; the chunks between the shortest and the longest window of the wavefront): ONE dword per window from inside it --
        // shifted to where the rows stand -- instead of four byte loads; mask_checked turns what is not a row of the window into
        // the pad letter (round 3 fetched bytes: a wavefront with one shorter window ran at 5.4 instead of 6.0 TCUPS)
        // (two halves: the loads -- one clamped dword per window, issued a chunk ahead -- and, once the chunk's steps are through and
        // BEFORE its checkpoint stores are issued, the shift to where the rows stand.  gfx950 counts loads and stores with ONE in-order
        // counter: a wait for a load that stands behind stores waits for the stores' acknowledgements from HBM as well -- round 4's
        // loops consumed their prefetches at the loop's end, behind the chunk's stores, and spent two fifths of their wave cycles there)
        auto fetch_checked_raw = [&](int k0, uint32_t & wA, uint32_t & wB)
        {
            int const i0 = k0 - g;
            auto one = [&](uint8_t const * sp, uint32_t lsc) -> uint32_t
            {
                int const hi = max((int)lsc - 3, 0);            // last position a whole dword starts at (the buffers carry slack for windows of < 4 rows)
                int const a0 = min(max(i0, 0), hi);
                return *reinterpret_cast<unaligned_u32 const *>(sp + a0);
            };
            wA = one(sA, lscA);
            wB = one(sB, lscB);
        };
        auto expand_checked = [&](int k0, uint32_t wA, uint32_t wB, uint32_t (&ta)[4], uint32_t (&tb)[4])
        {
            int const i0 = k0 - g;
            auto one = [&](uint32_t w, uint32_t lsc, uint32_t (&t)[4])
            {
                int const      hi = max((int)lsc - 3, 0);
                int const      a0 = min(max(i0, 0), hi);
                int const      sh = i0 - a0;                         // row i0 is byte `sh` of the dword
                uint32_t const x  = sh >= 0 ? (sh < 4 ? w >> (8 * sh) : 0u) : (sh > -4 ? w << (8 * -sh) : 0u);
                t[0] = x & 0xffu;
                t[1] = (x >> 8) & 0xffu;
                t[2] = (x >> 16) & 0xffu;
                t[3] = x >> 24;
            };
            one(wA, lscA, ta);
            one(wB, lscB, tb);
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

        // the prefetches of the NEXT chunk are taken over here: behind the chunk's steps, in front of its stores (see fetch_checked_raw).
        // The empty asm is the use that pins the wait to this place.
        auto take_over = [&](uint32_t & wA, uint32_t & wB)
        {
            if constexpr (MULTI)
                asm volatile("" : "+v"(wA), "+v"(wB), "+v"(ninA[0]), "+v"(ninA[1]), "+v"(ninA[2]), "+v"(ninA[3]), "+v"(ninE[0]), "+v"(ninE[1]), "+v"(ninE[2]),
                             "+v"(ninE[3])::"memory");
            else
                asm volatile("" : "+v"(wA), "+v"(wB)::"memory");
            if constexpr (MULTI)
            {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    cinA[u] = ninA[u];
                    cinE[u] = ninE[u];
                }
            }
        };
        // the chunk's stores: boundary codes, carry-out, and the row checkpoint behind every fourth chunk
        auto chunk_stores = [&](int k0)
        {
            chunk_done(k0);
            if (((k0 + 4) & 15) == 0)
            {
                book(k0);
                rowck_codes(k0);
            }
        };
        auto four_steps = [&](uint32_t const (&ta)[4], uint32_t const (&tb)[4], int k0)
        {
            uint32_t pa0[Geo::kD], pb0[Geo::kD], pa1[Geo::kD], pb1[Geo::kD];
            rows(ta[0], tb[0], pa0, pb0);
            rows(ta[1], tb[1], pa1, pb1);
            __builtin_amdgcn_sched_barrier(0); // (pins the fetches in front of the step they overlap)
            step(pa0, pb0, k0, 0);
            rows(ta[2], tb[2], pa0, pb0);
            __builtin_amdgcn_sched_barrier(0);
            step(pa1, pb1, k0 + 1, 1);
            rows(ta[3], tb[3], pa1, pb1);
            __builtin_amdgcn_sched_barrier(0);
            step(pa0, pb0, k0 + 2, 2);
            step(pa1, pb1, k0 + 3, 3);
        };
        int      k0 = 0;
        // (what the chunk loop carries of the subject is two raw dwords, in the checked path as in the steady one: the letters of the four
        // rows are cut out where the chunk starts -- eight expanded letters carried across the iteration were eight more registers in a
        // kernel that stands at 256, and their spills were reloaded behind the chunk's stores: a wait for the stores again)
        uint32_t curA, curB;
        {
            fetch_checked_raw(0, curA, curB);
            carry_fetch(0, Z, ninA, ninE);
            take_over(curA, curB);
        }
        while (k0 < steps)
        {
            bool const cur_steady = (k0 >= steady_lo) && (k0 < steady_hi);
            if (!cur_steady)
            {
                uint32_t ca[4], cb[4];
                expand_checked(k0, curA, curB, ca, cb);
                mask_checked(k0, ca, cb);
                uint32_t rA, rB;
                fetch_checked_raw(k0 + 4, rA, rB);
                carry_fetch(k0 + 4, Z + qsplat(-4 * ge), ninA, ninE); // (Z is the chunk's first step's here)
                four_steps(ca, cb, k0);
                take_over(rA, rB);
                curA = rA;
                curB = rB;
                chunk_stores(k0);
                k0 += 4;
            }
            else
            {
                uint32_t wa = *reinterpret_cast<unaligned_u32 const *>(spA + k0);
                uint32_t wb = *reinterpret_cast<unaligned_u32 const *>(spB + k0);
                // (here, once per panel, and not as a pending load that the loop's first use would have to wait for in every iteration)
                asm volatile("" : "+v"(wa), "+v"(wb)::"memory");
                while (k0 < steady_hi)
                {
                    uint32_t const ca = wa, cb = wb;
                    int const      kn = max(min(k0 + 4, ls_min - 4), 0);
                    wa                = *reinterpret_cast<unaligned_u32 const *>(spA + kn);
                    wb                = *reinterpret_cast<unaligned_u32 const *>(spB + kn);
                    carry_fetch(k0 + 4, Z + qsplat(-4 * ge), ninA, ninE);
                    uint32_t const la[4] = {ca & (kAlph - 1), (ca >> 8) & (kAlph - 1), (ca >> 16) & (kAlph - 1), (ca >> 24) & (kAlph - 1)};
                    uint32_t const lb[4] = {cb & (kAlph - 1), (cb >> 8) & (kAlph - 1), (cb >> 16) & (kAlph - 1), (cb >> 24) & (kAlph - 1)};
                    four_steps(la, lb, k0);
                    take_over(wa, wb);
                    chunk_stores(k0);
                    k0 += 4;
                }
                fetch_checked_raw(k0, curA, curB);
                asm volatile("" : "+v"(curA), "+v"(curB)::"memory"); // (as above: once per transition, not a pending load at the loop's top)
            }
        }
        if (!WIDE && (steps & 4))
            flush_codes(steps); // the last four steps' codes (the other half of the group is stale: beyond every row)
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
        merge((int)best.x, rowA, (int)(tie & 1u), runA, stripA, rrowA, rtieA);
        merge((int)best.y, rowB, (int)((tie >> 1) & 1u), runB, stripB, rrowB, rtieB);

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
        }; // panel_body
        if (code == 0)
            panel_body(std::integral_constant<int, C>{});
        else if (code == 1)
            panel_body(std::integral_constant<int, (3 * C + 3) / 4>{});
        else if (code == 2)
            panel_body(std::integral_constant<int, (C + 1) / 2>{});
        else
            panel_body(std::integral_constant<int, (C + 3) / 4>{});
    } // panels

    auto finish = [&](int run, int rstrip, int rrow, int rtie, bool act, uint64_t e, int code)
    {
        if (is_first && act)
        {
            EndCell ec{};
            bool const declined = MULTI && !WIDE && run > 2046; // beyond what the codes hold: the int32 launch redoes it
            if (!writable || declined)
                ec.score = -1;
            else if (run > 0)
            {
                ec.score = run;
                ec.q_end = -(rstrip + 1); // the backtrace finds the column inside this strip
                ec.s_end = rrow + 1;
                ec.flags = (rtie ? kEndAmbiguous : 0) | (WIDE ? 0 : kEndCompact) | (code << kEndNarrowShift);
            }
            p.ends[e]      = ec;
            p.out_score[e] = (writable && !declined) ? run : -1;
            if (p.stat_beyond && run > 2046) // (what decides between compact codes and int16 pairs for the next chunks: lx_host.cpp)
                atomicAdd(p.stat_beyond, 1u);
        }
    };
    finish(runA, stripA, rrowA, rtieA, actA, eA, my_code);
    finish(runB, stripB, rrowB, rtieB, actB, eB, solo_mode ? my_codeB : my_code);
}

template <int C>
static hipError_t launch_sweep_mq_cfg(ScoreParams const & p, hipStream_t stream)
{
    using Geo = MqGeo<C>;
    uint64_t const all_wf = (p.n + 2ull * Geo::kGroups - 1) / (2ull * Geo::kGroups);
    if (p.wf_lo > all_wf)
        return hipErrorInvalidValue;
    uint64_t const blocks = all_wf - p.wf_lo; // (p.n = the slots up to the END of the launch's range: a first launch of two passes fewer)
    if (blocks == 0)
        return hipSuccess;
    int const      share  = (p.pair_share > 0 && p.pair_share < Geo::kGroups) ? p.pair_share : Geo::kGroups;
    if (blocks > 0x7fffffffull || !p.ckpt || !p.ends || (!p.wf_tab && p.steps_cap % 16 != 0) || Geo::kGroups % share != 0)
        return hipErrorInvalidValue;
    size_t const lds = ((size_t)(p.solo ? 2 * Geo::kGroups : p.pair_share == 1 ? 4 : Geo::kGroups / share) * (size_t)p.nrows * Geo::kRowDw + 64 * 8) * sizeof(uint32_t);
    if (p.wide)
    {
        if constexpr (C == 19)
            hipLaunchKernelGGL((sweep_mq_kernel<C, true, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
        else
            return hipErrorInvalidValue; // (int16-pair slots: the 19-column strips only)
    }
    else if (p.panels_cap > 1) // (slots by wavefront: panels_cap = the launch's largest)
        hipLaunchKernelGGL((sweep_mq_kernel<C, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    else
        hipLaunchKernelGGL((sweep_mq_kernel<C, false>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}

// trace cfg 5 = (8,11), 3 = (8,13), 1 = (8,19): strips whose byte profiles leave room for four queries per wavefront at two
// wavefronts per SIMD (<= 20 bytes per lane and letter; 25-column strips were measured and lose a fifth to their occupancy).
// p.pair_share = lane groups per query (1, 2, 4; 0 or 8 = one query per wavefront); the slots are the compact codes of
// Ckpt16Layout<8, C>, p.panels_cap parts each.
hipError_t launch_sweep_mq(int trace_cfg, ScoreParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    return trace_cfg == 1 ? launch_sweep_mq_cfg<19>(p, stream) : trace_cfg == 3 ? launch_sweep_mq_cfg<13>(p, stream) : trace_cfg == 5 ? launch_sweep_mq_cfg<11>(p, stream) : hipErrorInvalidValue;
}

// LDS bytes of one wavefront (profiles + staging)
size_t sweep_mq_lds_bytes(int trace_cfg, int nrows, int share)
{
    int const C = trace_cfg == 1 ? 19 : trace_cfg == 3 ? 13 : 11;
    int const s = (share > 0 && share < 8) ? share : 8;
    // (share -1: the solo packing, a profile per window)
    return ((size_t)(share < 0 ? 16 : share == 1 ? 4 : 8 / s) * (size_t)nrows * (size_t)((C + 3) / 4 * 8) + 64 * 8) * sizeof(uint32_t);
}

} // namespace lx

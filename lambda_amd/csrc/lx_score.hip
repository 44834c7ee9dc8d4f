// lx_score.hip -- pass-1 score-only kernel of the seed extension (gfx950 / CDNA4 only).
//
// Replaces _performAlignment<withTrace=false> -> seqan::_prepareAndRunSimdAlignment
// (/root/reference/src/search_algo.hpp:1246 -> :1070-1134): affine-gap *local* alignment over the FULL
// Lq x Ls rectangle (DPBandConfig<BandOff>, :1081), query horizontal / subject vertical (:1058-1059),
// output = best score (:1129).
//
// Mapping to CDNA4 (integer VALU bound; no MFMA, no HBM pressure -- 0.013 B/cell):
//   * A group of G lanes owns one extension; lane g owns C consecutive query columns (a "strip"), kept in
//     registers.  At step k lane g processes subject row i = k - g, so the group sweeps anti-diagonals of
//     strips: the cell-level anti-diagonal recurrence becomes a strip-level systolic pipeline.  A wavefront
//     holds 64/G extensions.
//   * Strip boundaries (H and E of the strip's last column) move one lane to the right per step through DPP
//     (row_shr:1 / wave_shr:1) -- VALU-rate cross-lane moves, no LDS round trip.
//   * Substitution scores come from a per-query *profile* in LDS: prof[t][d][g] = the 4 int8 scores of lane g's
//     columns 4d..4d+3 against subject letter t (27x27 matrix -> (alph+1) x Lq table).  One ds_read_b32 per 4
//     cells, conflict-free within a group (consecutive lanes -> consecutive banks), instead of one matrix
//     lookup per cell.  The int8 -> int32 sign extension is folded into the add (SDWA).
//   * Queries wider than one panel (G*C columns) are processed panel by panel; the last strip's boundary
//     column is carried through a small global workspace (L2 resident).
//
// Row-skewed recurrence (all values carry + z_i, z_i = -ge*i, i = subject row) which removes the "+ge" from the
// vertical gap and lets the zero floor ride inside the F state:
//     t      = Hs[i-1][j-1] + (s(q_j,t_i) - ge)
//     Hs     = max3(t, Es, F0)                 F0 >= z_i carries the local-alignment floor
//     A      = Hs + (go - ge)
//     F0'    = max3(F0, A, z_{i+1})            (vertical gap for row i+1, floor folded in)
//     Es'    = max(Es, A) + ge                 (horizontal gap for column j+1, same row)
//     best   = max(best, max_c(Hs) - z_i)
// = 6.5 integer VALU ops per cell instead of 10 for the textbook Gotoh form; results are identical (the skew is
// a per-row constant shared by everything compared inside a cell).
#include <hip/hip_runtime.h>

#include "lx_dp_common.h"

namespace lx
{



// MULTI = the launch may contain queries wider than one panel (carry workspace code compiled in)
// BAND = band mode (ScoreParams::band > 0; NOT what the reference runs, src/search_algo.hpp:1081 is BandOff): cells off the
// band behave as if they did not exist -- H = 0, no gap state leaves them (oracle/lx_oracle.c lxo_score_banded).  The
// strip mapping cannot skip them (at every step some lane of the group is inside the band), so they are computed and
// overwritten: five more instructions per cell; rows below the band's last row are clipped.
template <int G, int C, bool MULTI, bool BAND = false>
__global__ __launch_bounds__(64) void score_kernel(ScoreParams p)
{
    using Geo = ScoreGeo<G, C>;
    extern __shared__ uint32_t lds[];

    int const  lane     = threadIdx.x;
    int const  grp      = lane / G;
    int const  g        = lane % G;
    bool const is_first = (g == 0);
    bool const is_last  = (g == G - 1);

    uint64_t const e      = (uint64_t)blockIdx.x * Geo::kGroups + grp;
    bool           active = e < p.n;
    if (p.fixup)
    {
        // second launch after the packed-half kernel: only the extensions it declined (sentinel -1) are computed
        active = active && p.out_score[e] < 0;
        if (__ballot(active) == 0)
            return;
    }

    ScoringDev const * __restrict__ sc = p.sc;
    int const      ge    = sc->ge;
    int const      g2    = sc->g2;
    int const      nrows = p.nrows;                 // = alph + 1, rounded up to a multiple of 4 rows in LDS
    uint32_t const padt  = (uint32_t)(nrows - 1);   // pad subject letter -> profile row of kNegPad

    int             lq = 0, ls = 0;
    uint8_t const * q = p.q_res;
    uint8_t const * s = p.s_res;
    uint64_t        q_off = 0;
    if (active)
    {
        Extension const x = p.ext[e];
        lq    = (int)x.q_len;
        ls    = (int)x.s_len;
        q_off = x.q_off;
        q += x.q_off;
        if (ls != 0)
            s += x.s_off;
    }
    int blo = 0, bw = 0; // in-band diagonals: blo <= i - j <= blo + bw
    if constexpr (BAND)
    {
        if (active)
        {
            int const d0 = p.band_diag ? p.band_diag[e] : band_default_diag(lq, ls);
            blo          = d0 - p.band;
            bw           = 2 * p.band;
            ls           = min(ls, max(lq + blo + bw, 0)); // from row Lq + hi on no cell is inside the band
        }
    }
    // An empty query or window needs no special case: no real row/column exists, every cell stays at the floor.

    if (p.shared_profile)
    {
        // the caller promised one query per wavefront (LX_OPT_QUERY_RUN): verify, fail loudly otherwise
        uint64_t const q0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(q_off >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)q_off);
        int const l0 = __builtin_amdgcn_readfirstlane(lq);
        if (active && (q_off != q0 || lq != l0))
            atomicExch(p.err, 2);
    }

    // wave-uniform loop bounds
    int       ls_max = ls, ls_min = active ? ls : 0x7fffffff;
    int       npanels   = (lq + Geo::kPanel - 1) / Geo::kPanel;
    int const my_panels = npanels;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        ls_max = max(ls_max, __shfl_xor(ls_max, off));
        ls_min = min(ls_min, __shfl_xor(ls_min, off));
        if constexpr (MULTI)
            npanels = max(npanels, __shfl_xor(npanels, off));
    }
    ls_max = __builtin_amdgcn_readfirstlane(ls_max);
    ls_min = __builtin_amdgcn_readfirstlane(ls_min);
    if constexpr (MULTI)
        npanels = __builtin_amdgcn_readfirstlane(npanels);
    else
        npanels = 1; // host guarantees q_len <= panel width for this instantiation

    // carry workspace for multi-panel queries: Ls pairs (Hs, Es) per extension
    int32_t * carry = nullptr;
    if constexpr (MULTI)
    {
        if (npanels > 1)
        {
            uint32_t base = 0;
            int      ok   = 1;
            if (is_first && my_panels > 1)
            {
                base = atomicAdd(p.ws_top, (uint32_t)ls);
                if (base + (uint32_t)ls > p.ws_cap)
                {
                    ok = 0;
                    atomicExch(p.err, 1);
                }
            }
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) // broadcast lane g==0's values through the group
            {
                base = max(base, (uint32_t)__shfl_xor((int)base, off));
                ok   = min(ok, __shfl_xor(ok, off));
            }
            if (my_panels > 1 && ok)
                carry = p.ws + 2ull * base;
            else if (my_panels > 1)
            {
                lq = 0; // overflow: neutralise this extension (reported through p.err)
                ls = 0;
            }
        }
    }

    int const      slot_dw  = p.shared_profile ? 0 : grp * (nrows * Geo::kRowDw);
    uint32_t const row_base = (uint32_t)(slot_dw + g) * 4u; // byte address of this lane's dword 0 in profile row 0
    constexpr uint32_t kRowBytes = Geo::kRowDw * 4;

    // steps rounded up to whole chunks of 4; the surplus rows are virtual (pad letter) and cannot change the result
    int const steps  = (ls_max + G - 1 + 3) & ~3;
    // chunk [k0, k0+4) is "steady" when every lane of every group is inside its window for all 4 steps
    int const steady_lo = (G - 1 + 3) & ~3; // first chunk start with k0 - (G-1) >= 0
    int const steady_hi = ls_min - 3;       // k0 + 3 < ls_min

    uint32_t const lsc = (uint32_t)max(ls, 1) - 1u; // clamp for the checked path (inactive groups read s_res[0])

    int best = 0;

    for (int panel = 0; panel < npanels; ++panel)
    {
        int const col0 = panel * Geo::kPanel + g * C;
        build_profile<G, C>(lds, slot_dw, g, q, lq, col0, sc->mat_adj, nrows, !p.shared_profile || grp == 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        bool use_carry_in = false, do_carry_out = false;
        if constexpr (MULTI)
        {
            use_carry_in = is_first && (panel > 0) && (panel < my_panels) && carry != nullptr;
            do_carry_out = is_last && (panel + 1 < my_panels) && carry != nullptr;
        }

        // state as of (virtual) row i = -g - 1: every cell is "H = 0"
        int z = ge * g; // z_i = -ge * i for the first processed row i = -g
        int Hrow[C], F0[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            Hrow[c] = z + ge; // z_{i-1}
            F0[c]   = z;      // floor of row i
        }
        int diag0 = z + ge; // Hs[i-1][col -1]
        int sendH = z + ge; // the right neighbour's first row is -(g+1): it must see "H = 0" there
        int sendE = kNegInf;

        uint8_t const * sp = s - g; // sp[k] = s[k - g]

        // one DP step: subject letter t against this lane's strip
        auto step = [&](int k, uint32_t t)
        {
            uint32_t const * prow = reinterpret_cast<uint32_t const *>(reinterpret_cast<char const *>(lds) + row_base + t * kRowBytes);
            uint32_t         pw[Geo::kD];
#pragma unroll
            for (int d = 0; d < Geo::kD; ++d)
                pw[d] = prow[d * G];

            // left boundary of this strip for row i: H[i][-1] = 0, E = -inf; or the previous panel's last column
            int bndH = z, bndE = kNegInf;
            if constexpr (MULTI)
            {
                int const i = k - g;
                if (use_carry_in && (unsigned)i < (unsigned)ls)
                {
                    bndH = carry[2 * i];
                    bndE = carry[2 * i + 1];
                }
            }
            int const recvH = shift_from_left<G>(sendH, bndH, is_first);
            int       Ecur  = shift_from_left<G>(sendE, bndE, is_first);

            int dg = diag0;
            diag0  = recvH;

            int const zn     = z - ge; // z_{i+1}
            int       rowmax = kNegInf;
            int       h      = 0;
            int const cl     = (k - g) - (blo + bw) - col0; // BAND: first column of this strip inside the band in this row
#pragma unroll
            for (int c = 0; c < C; ++c)
            {
                int const sub = (int)(int8_t)(pw[c >> 2] >> (8 * (c & 3)));
                int const tt  = dg + sub;
                dg            = Hrow[c];
                h             = max3i(tt, Ecur, F0[c]);
                LX_OPAQUE(h); // keeps rowmax a chain over h instead of a wider tree over (tt, E, F0)
                if constexpr (BAND)
                {
                    bool const inb = (unsigned)(c - cl) <= (unsigned)bw;
                    h              = inb ? h : z; // H = 0
                    int const A    = h + g2;
                    int const Fn   = max3i(F0[c], A, zn);
                    F0[c]          = inb ? Fn : zn;                       // the cell below sees the floor only
                    Ecur           = inb ? max(Ecur, A) + ge : kNegInf;   // no horizontal gap leaves the cell
                }
                else
                {
                    int const A = h + g2;
                    F0[c]       = max3i(F0[c], A, zn);
                    LX_OPAQUE(F0[c]);
                    Ecur        = max(Ecur, A) + ge;
                }
                Hrow[c]       = h;
                rowmax        = max(rowmax, h);
            }
            sendH = h;
            sendE = Ecur;
            best  = max(best, rowmax - z);
            z     = zn;

            if constexpr (MULTI)
            {
                int const i = k - g;
                if (do_carry_out && (unsigned)i < (unsigned)ls)
                {
                    carry[2 * i]     = sendH;
                    carry[2 * i + 1] = sendE;
                }
            }
        };

        // subject letters are fetched one chunk (4 steps) ahead of their use
        auto fetch_checked = [&](int k0, uint32_t (&t)[4])
        {
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                uint32_t const i   = (uint32_t)(k0 + u - g);
                uint32_t const idx = min(i, lsc); // i < 0 wraps to a huge value -> clamped; masked below
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

        int      k0 = 0;
        uint32_t tn[4];
        fetch_checked(0, tn);
        while (k0 < steps)
        {
            bool const cur_steady = (k0 >= steady_lo) && (k0 < steady_hi);
            if (!cur_steady)
            {
                uint32_t tc[4] = {tn[0], tn[1], tn[2], tn[3]};
                mask_checked(k0, tc);
                fetch_checked(k0 + 4, tn);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    step(k0 + u, tc[u]);
                k0 += 4;
            }
            else
            {
                // steady state: no bounds checks, one (unaligned) dword of 4 letters per lane per chunk
                uint32_t wn = *reinterpret_cast<unaligned_u32 const *>(sp + k0);
                while (k0 < steady_hi)
                {
                    uint32_t const wc = wn;
                    // the prefetch may run up to 4 bytes past this lane's last steady row: clamp to stay inside the window
                    int const kn = min(k0 + 4, ls_min - 4 + 0);
                    wn           = *reinterpret_cast<unaligned_u32 const *>(sp + max(kn, 0));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        step(k0 + u, (wc >> (8 * u)) & (kAlph - 1));
                    k0 += 4;
                }
                fetch_checked(k0, tn);
            }
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
    }

    // reduce over the lanes of the group
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1)
        best = max(best, __shfl_xor(best, off));

    if (active && is_first)
        p.out_score[e] = best;
}

// ---- host-visible launcher -------------------------------------------------------------------------

template <int G, int C>
static hipError_t launch_score_cfg(ScoreParams const & p, bool multi, hipStream_t stream)
{
    using Geo = ScoreGeo<G, C>;
    if (p.n == 0)
        return hipSuccess;
    uint64_t const blocks = (p.n + Geo::kGroups - 1) / Geo::kGroups;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    int const    slots = p.shared_profile ? 1 : Geo::kGroups;
    size_t const lds   = (size_t)slots * (size_t)p.nrows * Geo::kRowDw * sizeof(uint32_t);
    if (p.band > 0)
    {
        // band mode: the any-width instantiation of the generic geometry, or (8,19) with one LDS profile per wavefront
        // for query runs (launch_score_list sends nothing else here)
        if constexpr (G == 16 && C == 10)
            hipLaunchKernelGGL((score_kernel<G, C, true, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
        else if constexpr (G == 8 && C == 19)
            hipLaunchKernelGGL((score_kernel<G, C, false, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
        else
            return hipErrorInvalidValue;
    }
    else if (multi)
        hipLaunchKernelGGL((score_kernel<G, C, true>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    else
        hipLaunchKernelGGL((score_kernel<G, C, false>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}

// Kernel geometries (G lanes per extension x C columns per lane).  G = 8 geometries put 8 extensions in a
// wavefront and are only used with one shared profile per wavefront (LX_OPT_QUERY_RUN / host-side binning).
struct ScoreCfg
{
    int g, c;
};
static constexpr ScoreCfg kScoreCfgs[] = {
    {16, 10}, // 0: 160 columns
    {8, 8},   // 1:  64
    {32, 10}, // 2: 320
    {64, 10}, // 3: 640 (and the multi-panel workhorse for longer queries)
    {8, 13},  // 4: 104
    {8, 16},  // 5: 128
    {8, 19},  // 6: 152
    {16, 13}, // 7: 208
    {16, 16}, // 8: 256
};
constexpr int kNumScoreCfgs = sizeof(kScoreCfgs) / sizeof(ScoreCfg);

// multi: the batch may hold queries wider than the panel (enables the carry-workspace code path)
hipError_t launch_score(int cfg, ScoreParams const & p, bool multi, hipStream_t stream)
{
    switch (cfg)
    {
        case 0: return launch_score_cfg<16, 10>(p, multi, stream);
        case 1: return launch_score_cfg<8, 8>(p, multi, stream);
        case 2: return launch_score_cfg<32, 10>(p, multi, stream);
        case 3: return launch_score_cfg<64, 10>(p, multi, stream);
        case 4: return launch_score_cfg<8, 13>(p, multi, stream);
        case 5: return launch_score_cfg<8, 16>(p, multi, stream);
        case 6: return launch_score_cfg<8, 19>(p, multi, stream);
        case 7: return launch_score_cfg<16, 13>(p, multi, stream);
        case 8: return launch_score_cfg<16, 16>(p, multi, stream);
        default: return hipErrorInvalidValue;
    }
}

int score_cfg_count() { return kNumScoreCfgs; }

int score_cfg_panel(int cfg)
{
    return (cfg >= 0 && cfg < kNumScoreCfgs) ? kScoreCfgs[cfg].g * kScoreCfgs[cfg].c : 0;
}

int score_cfg_groups(int cfg)
{
    return (cfg >= 0 && cfg < kNumScoreCfgs) ? 64 / kScoreCfgs[cfg].g : 0;
}

} // namespace lx

// lx_score_f16.hip -- pass-1 score kernel in packed half precision: two extensions per lane group (gfx950 only).
//
// Same strip-systolic mapping and row-skewed recurrence as lx_score.hip (see there for the reference lines), but
// every VGPR holds the DP value of TWO extensions (A in the low half, B in the high half) that share the query, and
// the arithmetic is v_pk_add_f16 / v_pk_maximum3_f16.  On gfx950 the packed ops issue at the same rate as the
// int32 max/max3 (tools/ubench.hip), so a cell costs 7.5 issue slots instead of 11.  Scores are small integers;
// half precision represents every integer of magnitude <= 2048 exactly and -inf stands in for "minus infinity", so
// the results are bit-identical to the int32 kernel AS LONG AS no intermediate exceeds 2048.  That is decided per
// wavefront from an upper bound:
//         sum_j max(0, max_b s(q_j, b))  +  |ge| * (rows processed)  +  max entry  <=  2046
// (a local alignment cannot score more than the sum of the best positive score of each query column).  A wavefront
// that fails the test writes the sentinel -1 for its extensions and the host follows up with the int32 kernel in
// "fix-up" mode, which recomputes exactly those.  Nothing is approximated.
//
// LDS: one query profile per wavefront in half precision, lane-contiguous rows (24 halves = 48 B per lane per subject
// letter, read with two ds_read_b128 + one ds_read_b64); the two subject letters of a lane give two reads whose
// halves are interleaved with one v_perm_b32 per column.  All extensions of a wavefront (16, or 8 for G = 16) must share the query
// (LX_OPT_QUERY_RUN multiple of 16, or host-side padding).
#include <hip/hip_runtime.h>

#include "lx_aids.h"
#include "lx_dp_common.h"

namespace lx
{

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#ifndef LX_F16_UNROLL
#define LX_F16_UNROLL 2 // steps unrolled in the main loop: 2 removes the register-rotation moves at 168 VGPRs
#endif
#ifndef LX_F16_CKPT_WAVES
#define LX_F16_CKPT_WAVES 3 // wavefronts per SIMD the checkpointing instantiations are compiled for (headline sweep: 2 -> 12.1 ms, 3 -> 11.9 ms; steps unrolled 1 / 2 / 4 -> 12.7 / 11.9 / 12.0 ms)
#endif
#define LX_PRAGMA(x) _Pragma(#x)
#define LX_UNROLL(n) LX_PRAGMA(unroll n)

__device__ __forceinline__ h2 hmax3(h2 a, h2 b, h2 c)
{
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); // v_pk_maximum3_f16
}
__device__ __forceinline__ h2 hmax(h2 a, h2 b)
{
    return __builtin_elementwise_maximum(a, b);
}
__device__ __forceinline__ h2 hsplat(float x)
{
    return h2{(_Float16)x, (_Float16)x};
}
__device__ __forceinline__ h2 as_h2(uint32_t x)
{
    return __builtin_bit_cast(h2, x);
}
__device__ __forceinline__ uint32_t as_u32(h2 x)
{
    return __builtin_bit_cast(uint32_t, x);
}

constexpr uint32_t kHalfNegInf2 = 0xfc00fc00u; // (-inf, -inf)

// Ckpt16Layout codes without a conversion instruction: an integer n in 0 .. 2047 scaled by 2^-24 is a half-precision
// subnormal (n < 1024) or lies in the first binade (spacing 2^-24 as well), so the bit pattern of n * 2^-24 is n itself
// (FP16 denormals are on in the kernel descriptor, the compiler's default).  Scaling by a power of two and adding
// multiples of 2^-24 below 2048 * 2^-24 is exact.
constexpr uint32_t kHalfTwoPowMinus24x2 = 0x00010001u; // (2^-24, 2^-24)
__device__ __forceinline__ uint32_t c16_pack(h2 value_scaled, h2 diff_scaled)
{
    return (as_u32(diff_scaled) << 11) | as_u32(value_scaled); // both halves at once: diff < 32 stays inside its half
}

template <int G, int C>
struct PairGeo
{
    static constexpr int kGroups   = 64 / G;                 // lane groups per wavefront, two extensions each
    static constexpr int kPanel    = G * C;
    static constexpr int kUsedDw   = (C + 1) / 2;            // dwords that hold real columns
    static constexpr int kLaneDw   = (kUsedDw + 3) & ~3;     // halves of a lane's columns per profile row, 16-byte granules
    // (padding the rows to spread different subject letters over more bank offsets was measured: no effect -- the
    // kernel is VALU-issue bound, LDS is ~1/3 busy including its bank conflicts)
    static constexpr int kRowDw    = kLaneDw * G;
};

// CKPT = single sweep: the kernel additionally writes what pass 2's backtrace needs (layout and meaning as in
// lx_ckpt.hip: strip boundaries per step, row checkpoints every 16 steps, here as the compact codes of Ckpt16Layout) and
// keeps, per strip and extension, the best value, the block of sixteen steps whose rows reached it first and whether a later block tied.
template <int G, int C, bool CKPT>
__global__ __launch_bounds__(64, (CKPT ? (C > 19 ? 2 : LX_F16_CKPT_WAVES) : 1)) void score_pair_kernel(ScoreParams p) // (25-column strips: 168 VGPRs spill, and their LDS profile leaves 9 wavefronts per CU anyway)
{
    static_assert(C <= 32, "profile rows hold 32 halves per lane");
    using Geo = PairGeo<G, C>;
    extern __shared__ uint32_t lds[];

    int const  lane     = threadIdx.x;
    int const  grp      = lane / G;
    int const  g        = lane % G;
    bool const is_first = (g == 0);

    uint64_t const pair   = (uint64_t)blockIdx.x * Geo::kGroups + grp;
    uint64_t const eA     = 2 * pair, eB = 2 * pair + 1;
    bool const     actA   = eA < p.n, actB = eB < p.n;

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
    // p.pair_share lane groups (0 = all of them) use one LDS profile: the extensions of such a block share the query
    int const share_g = (p.pair_share > 0 && p.pair_share < Geo::kGroups) ? p.pair_share : Geo::kGroups;
    int const blk     = grp / share_g;
    {
        // the caller promised one query per block: verify against the block's first lane, fail loudly otherwise
        int const      leader = blk * share_g * G;
        uint64_t const q0     = ((uint64_t)(uint32_t)__shfl((int)(q_off >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)q_off, leader);
        int const      l0     = __shfl(lq, leader);
        bool const     lead_in = __shfl(actA ? 1 : 0, leader) != 0;
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
    int const steps = (ls_max + G - 1 + 3) & ~3;

    // ---- exactness test (wave-uniform): an upper bound of every intermediate must stay <= 2046
    // (first the bound no query of this length exceeds -- every column at the matrix' largest entry: where that passes for the whole
    // wavefront, the residues need not be looked at: two dependent loads per column in front of everything else)
    int const col0  = g * C;
    int       bound = lq * sc->smax;
    if (__ballot(bound + (-ge) * (steps + G + 2) + sc->smax + 2 > 2046) != 0)
    {
        bound = 0;
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            int const j = col0 + c;
            if (j < lq)
                bound += sc->rowmax[q[j] & (kAlph - 1)];
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1)
            bound += __shfl_xor(bound, off);
    }
    // (every group summed its own query; one block over the limit sends the whole wavefront to the fix-up launch)
    bool too_big = __ballot((lq > Geo::kPanel) || (bound + (-ge) * (steps + G + 2) + sc->smax + 2 > 2046)) != 0;
    if constexpr (CKPT)
    {
        // LX_OPT_MAX_QLEN / LX_OPT_MAX_SLEN promise broken: reported here (the int32 launch, which would report it too, is
        // skipped when no query the promise admits can fail the exactness test)
        bool const broken = (__ballot(lq > Geo::kPanel) != 0) || (uint32_t)steps > p.steps_cap;
        if (broken && lane == 0)
            atomicExch(p.err, 3);
        too_big = too_big || broken;
    }

    if (too_big)
    {
        if (is_first)
        {
            if (actA)
                p.out_score[eA] = -1;
            if (actB)
                p.out_score[eB] = -1;
            if constexpr (CKPT)
            {
                EndCell none{};
                none.score = -1; // until the int32 fix-up launch has been here
                if (actA)
                    p.ends[eA] = none;
                if (actB)
                    p.ends[eB] = none;
            }
        }
        return;
    }

    // ---- profile: prof[t][g][h] = (s(q_col, t) - ge) as half, lane-contiguous
    int const slot_dw = blk * (nrows * Geo::kRowDw);
    if (grp % share_g == 0)
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
                uint4 const * mrow = reinterpret_cast<uint4 const *>(sc->mat_h + ql * kAlph);
#pragma unroll
                for (int x = 0; x < 4; ++x)
                {
                    uint4 const v      = mrow[x];
                    rows[cc][4 * x + 0] = v.x;
                    rows[cc][4 * x + 1] = v.y;
                    rows[cc][4 * x + 2] = v.z;
                    rows[cc][4 * x + 3] = v.w;
                }
            }
            uint32_t * dst = lds + slot_dw + g * Geo::kLaneDw + d;
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

    uint32_t const row_base  = (uint32_t)(slot_dw + g * Geo::kLaneDw) * 4u;
    constexpr uint32_t kRowBytes = Geo::kRowDw * 4;

    h2 const GE = hsplat((float)ge), G2 = hsplat((float)sc->g2), NGE = hsplat((float)-ge);
    h2       Z  = hsplat((float)(ge * g)); // z_i of the first processed row i = -g
    h2       Hrow[C], F0[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
    {
        Hrow[c] = Z + GE;
        F0[c]   = Z;
    }
    h2 diag0 = Z + GE;
    h2 sendH = Z + GE;
    h2 sendE = as_h2(kHalfNegInf2);
    h2 best  = hsplat(0.f);
    // CKPT: first row that reached this strip's best value, "a later row reached it again" (bit 0 = A, bit 1 = B)
    int      rowA = 0, rowB = 0;
    uint32_t tie  = 0;
    h2 const C24 = as_h2(kHalfTwoPowMinus24x2), NC24 = -C24, GEc = GE * C24;
    h2       nZc = -(Z * C24); // -Z 2^-24, kept in step with Z
    h2       cmax = as_h2(kHalfNegInf2); // best un-skewed row maximum of the current chunk
    using Lay = Ckpt16Layout<G, C>;
    // (slot p.n is a spare one that idle halves write to; wave_slots: the W slots of this wavefront interleaved piece by piece --
    // lx_device.h --, every half owns its place there)
    constexpr uint32_t kW       = 2 * Geo::kGroups;
    constexpr bool     wslots   = CKPT; // (always, for this kernel's checkpoints: a run-time choice costs the steady loop its registers)
    uint32_t * const   waveBase = CKPT ? p.ckpt + (uint64_t)blockIdx.x * kW * p.ckpt_stride : nullptr;
    uint32_t * const   slotA    = !CKPT ? nullptr : wslots ? waveBase + (2 * grp) * (G * 4) : p.ckpt + (actA ? eA : p.n) * p.ckpt_stride;
    uint32_t * const   slotB    = !CKPT ? nullptr : wslots ? waveBase + (2 * grp + 1) * (G * 4) : p.ckpt + (actB ? eB : p.n) * p.ckpt_stride;
    uint32_t const     octMul   = wslots ? kW * G : G;                          // uint4 units between two groups of eight steps
    uint32_t const     ckMul    = (wslots ? kW * G : G) * (Lay::kCkDw / 4);     // ... between two row checkpoints
    // staging of the boundary codes: [step % 8][lane] -- lane-minor, free of bank conflicts
    uint32_t * const stage = lds + ((Geo::kGroups + share_g - 1) / share_g) * (nrows * Geo::kRowDw) + lane;

    // (slot8 = step % 8: where the step's boundary codes are staged)
    auto step = [&](uint32_t tA, uint32_t tB, int slot8)
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

        // left boundary: H[i][-1] = 0 (skewed: z), E = -inf
        h2 const recvH = as_h2((uint32_t)shift_from_left<G>((int)as_u32(sendH), (int)as_u32(Z), is_first));
        h2       Ecur  = as_h2((uint32_t)shift_from_left<G>((int)as_u32(sendE), (int)kHalfNegInf2, is_first));
        h2       dg    = diag0;
        diag0          = recvH;

        h2 const ZN     = Z + NGE;
        h2       rowmax = as_h2(kHalfNegInf2);
        h2       h      = Z;
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            // (score of column c vs letter tA, score of column c vs letter tB)
            uint32_t const sel = (c & 1) ? 0x07060302u : 0x05040100u;
            h2 const       sub = as_h2(__builtin_amdgcn_perm(pb[c >> 1], pa[c >> 1], sel));
            h2 const       tt  = dg + sub;
            dg                 = Hrow[c];
            h                  = hmax3(tt, Ecur, F0[c]);
            h2 const A         = h + G2;
            F0[c]              = hmax3(F0[c], A, ZN); // (floating-point max is not re-associated: no fences needed)
            Ecur               = hmax(Ecur, A) + GE;
            Hrow[c]            = h;
            if (c & 1)
                rowmax = hmax3(rowmax, Hrow[c - 1], h);
        }
        if (C & 1)
            rowmax = hmax(rowmax, h);
        sendH = h;
        sendE = Ecur;
        h2 const cand = rowmax - Z;
        if constexpr (CKPT)
        {
            cmax = hmax(cmax, cand); // the rose / met-again logic runs once per block of sixteen steps (book)
            // un-skewed boundary pair (H of the strip's last column, E as the next strip's first column uses it) as
            // Ckpt16Layout codes of both extensions, staged for one 16-byte store per extension every eight steps
            h2 const hc           = h * C24;
            stage[slot8 * 64]     = c16_pack(hc + nZc, __builtin_elementwise_fma(Ecur, NC24, hc));
            nZc                   = nZc + GEc; // Z grows by -ge per step
        }
        else
            best = hmax(best, cand);
        Z = ZN;
    };
    // CKPT: the staged codes of the eight steps up to k0 + 3 leave, re-paired per extension.  Unconditional (an idle half
    // owns the spare slot behind the batch): whole 128-byte lines per lane group, no branch around the stores.
    auto flush_codes = [&](int k0)
    {
        if constexpr (CKPT)
        {
            uint32_t cw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                cw[u] = stage[u * 64];
            uint32_t const oi = ((uint32_t)k0 / 8) * octMul + (uint32_t)g; // (Lay::bnd_oct_index with the slots' spacing)
            reinterpret_cast<uint4 *>(slotA)[oi] = make_uint4(__builtin_amdgcn_perm(cw[1], cw[0], 0x05040100u), __builtin_amdgcn_perm(cw[3], cw[2], 0x05040100u),
                                                              __builtin_amdgcn_perm(cw[5], cw[4], 0x05040100u), __builtin_amdgcn_perm(cw[7], cw[6], 0x05040100u));
            reinterpret_cast<uint4 *>(slotB)[oi] = make_uint4(__builtin_amdgcn_perm(cw[1], cw[0], 0x07060302u), __builtin_amdgcn_perm(cw[3], cw[2], 0x07060302u),
                                                              __builtin_amdgcn_perm(cw[5], cw[4], 0x07060302u), __builtin_amdgcn_perm(cw[7], cw[6], 0x07060302u));
        }
    };
    // CKPT: the best-value bookkeeping, once per block of sixteen steps (the backtrace's tiles are these blocks: its first tile looks
    // at every row of the block up to the one reported here -- lx_ckpt.hip, need_col).  Per four steps, as rounds 2-5 had it, the
    // selects below were 3 % of the sweep's issue slots.
    auto book = [&](int k0) // k0: the first of the block's last four steps
    {
        if constexpr (CKPT)
        {
            // per half: did the strip's best rise in this block (then its first row is one of the block's sixteen; the
            // backtrace finds it), or was it only met again (a tie for the end cell)?  Rows beyond the window and
            // columns beyond the query stay strictly below a positive best: no validity test.
            h2 const       nb   = hmax(best, cmax);
            uint32_t const rose = as_u32(nb) ^ as_u32(best), met = as_u32(cmax) ^ as_u32(best);
            bool const     gtA = (rose & 0xffffu) != 0, gtB = (rose >> 16) != 0;
            bool const     eqA = (met & 0xffffu) == 0, eqB = (met >> 16) == 0;
            int const last = k0 + 3 - g; // the block's last row in this lane
            rowA = gtA ? last : rowA;
            rowB = gtB ? last : rowB;
            tie  = (gtA ? (tie & ~1u) : (tie | (eqA ? 1u : 0u)));
            tie  = (gtB ? (tie & ~2u) : (tie | (eqB ? 2u : 0u)));
            best = nb;
            cmax = as_h2(kHalfNegInf2);
        }
    };
    // CKPT: after every eighth step the staged boundary codes leave
    auto chunk_done = [&](int k0)
    {
        if constexpr (CKPT)
        {
#ifndef LX_EXP_NO_FLUSH
            if (k0 & 4)
                flush_codes(k0);
#endif
        }
    };
    // CKPT: the row checkpoint behind step k0 + 3 (k0 % 16 == 12).  Issued at the top of the next chunk, before that
    // chunk's loads, so that these stores never stand between a load and its wait (see chunk_done).
    auto rowck_store = [&](int k0)
    {
        if constexpr (CKPT)
        {
            // Hrow is in the frame of the row just processed (z_i = Z + ge after the update), F0 in the next row's:
            // H - F un-skewed = (Hrow - z_i) - (F0 - Z) = Hrow - F0 - ge
            h2 const nzic = nZc - GEc, nGEc = -GEc; // -(Z + ge) 2^-24, -ge 2^-24
            uint32_t code[2 * Lay::kCkDw];
#pragma unroll
            for (int c = 0; c < 2 * Lay::kCkDw; ++c)
                code[c] = c < C ? c16_pack(__builtin_elementwise_fma(Hrow[c < C ? c : 0], C24, nzic),
                                           __builtin_elementwise_fma(Hrow[c < C ? c : 0] - F0[c < C ? c : 0], C24, nGEc))
                                : 0u;
            uint32_t const base0 = (uint32_t)(Lay::bnd_dwords(p.steps_cap) / 4);
            uint4 * const  dA = wslots ? reinterpret_cast<uint4 *>(waveBase) + kW * base0 + (2 * grp) * (G * (Lay::kCkDw / 4)) : reinterpret_cast<uint4 *>(slotA) + base0;
            uint4 * const  dB = wslots ? reinterpret_cast<uint4 *>(waveBase) + kW * base0 + (2 * grp + 1) * (G * (Lay::kCkDw / 4)) : reinterpret_cast<uint4 *>(slotB) + base0;
#pragma unroll
            for (int x = 0; x < Lay::kCkDw / 4; ++x)
            {
                uint32_t wa[4], wb[4];
#pragma unroll
                for (int b = 0; b < 4; ++b)
                {
                    int const c = 2 * (4 * x + b); // columns c, c + 1 of extension A (low halves) / B (high halves)
                    wa[b]       = __builtin_amdgcn_perm(code[c + 1], code[c], 0x05040100u);
                    wb[b]       = __builtin_amdgcn_perm(code[c + 1], code[c], 0x07060302u);
                }
                uint32_t const qi = ((uint32_t)(k0 + 3) / 16) * ckMul + (uint32_t)g * (Lay::kCkDw / 4) + (uint32_t)x; // (Lay::rowck_quad_index, spaced)
                dA[qi] = make_uint4(wa[0], wa[1], wa[2], wa[3]);
                dB[qi] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
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

    // The letters of the NEXT chunk are loaded at a chunk's top and taken over behind its steps, IN FRONT of its checkpoint stores:
    // gfx950 counts loads and stores with one in-order counter, so a wait for a load that stands behind stores is a wait for the
    // stores' acknowledgements from HBM (rounds 1-4 took the letters over at the loop's end, behind the stores: `s_waitcnt vmcnt(0)`
    // straight after them in every chunk -- a fifth of the sweep's wave cycles parked there).  The empty asm is the use that pins
    // the wait; the row checkpoint follows the chunk it belongs to instead of opening the next one.
    auto chunk_stores = [&](int k0)
    {
        chunk_done(k0);
        if (((k0 + 4) & 15) == 0)
        {
            book(k0);
#ifndef LX_EXP_NO_ROWCK
            rowck_store(k0);
#endif
        }
    };
    int      k0 = 0;
    uint32_t na[4], nb[4];
    fetch_checked(0, na, nb);
    while (k0 < steps)
    {
        bool const cur_steady = (k0 >= steady_lo) && (k0 < steady_hi);
        if (!cur_steady)
        {
            uint32_t ca[4] = {na[0], na[1], na[2], na[3]}, cb[4] = {nb[0], nb[1], nb[2], nb[3]};
            mask_checked(k0, ca, cb);
            fetch_checked(k0 + 4, na, nb);
#pragma unroll 1
            for (int u = 0; u < 4; ++u)
                step(ca[u], cb[u], (k0 & 4) + u);
            if constexpr (CKPT)
                asm volatile("" : "+v"(na[0]), "+v"(na[1]), "+v"(na[2]), "+v"(na[3]), "+v"(nb[0]), "+v"(nb[1]), "+v"(nb[2]), "+v"(nb[3])::"memory");
            chunk_stores(k0);
            k0 += 4;
        }
        else
        {
            uint32_t wa = *reinterpret_cast<unaligned_u32 const *>(spA + k0);
            uint32_t wb = *reinterpret_cast<unaligned_u32 const *>(spB + k0);
            if constexpr (CKPT) // (here, once, and not as a pending load that the loop's first use waits for in every iteration)
                asm volatile("" : "+v"(wa), "+v"(wb)::"memory");
            while (k0 < steady_hi)
            {
                uint32_t const ca = wa, cb = wb;
                int const      kn = max(min(k0 + 4, ls_min - 4), 0);
                wa                = *reinterpret_cast<unaligned_u32 const *>(spA + kn);
                wb                = *reinterpret_cast<unaligned_u32 const *>(spB + kn);
LX_UNROLL(LX_F16_UNROLL)
                for (int u = 0; u < 4; ++u)
                    step((ca >> (8 * u)) & (kAlph - 1), (cb >> (8 * u)) & (kAlph - 1), (k0 & 4) + u);
                if constexpr (CKPT)
                    asm volatile("" : "+v"(wa), "+v"(wb)::"memory");
                chunk_stores(k0);
                k0 += 4;
            }
            fetch_checked(k0, na, nb);
        }
    }
    if (steps & 4)
        flush_codes(steps); // the last four steps' codes (the other half of the group is stale: beyond every row)
    if (steps & 15)
        book(steps - 4); // the last, partial block

    if constexpr (!CKPT)
    {
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1)
            best = hmax(best, as_h2((uint32_t)__shfl_xor((int)as_u32(best), off)));

        if (is_first)
        {
            if (actA)
                p.out_score[eA] = (int)(float)best.x;
            if (actB)
                p.out_score[eB] = (int)(float)best.y;
        }
    }
    else
    {
        // per extension: best strip value over the group; among equal ones the lowest strip (its columns come first)
        auto finish = [&](int lbest, int lrow, int ltie, bool act, uint64_t e)
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
            if (is_first && act)
            {
                EndCell ec{};
                if (gbest > 0)
                {
                    ec.score = gbest;
                    ec.q_end = -(gstrip + 1); // the backtrace finds the column inside this strip
                    ec.s_end = grow + 1;
                    ec.flags = (gtie ? kEndAmbiguous : 0) | kEndCompact | kEndWaveSlots; // compact slot, interleaved per wavefront
                }
                p.ends[e]      = ec;
                p.out_score[e] = gbest;
            }
        };
        finish((int)(float)best.x, rowA, (int)(tie & 1u), actA, eA);
        finish((int)(float)best.y, rowB, (int)((tie >> 1) & 1u), actB, eB);
    }
}

template <int G, int C, bool CKPT = false>
static hipError_t launch_pair_cfg(ScoreParams const & p, hipStream_t stream)
{
    using Geo = PairGeo<G, C>;
    uint64_t const per_wave = 2ull * Geo::kGroups;
    uint64_t const blocks   = (p.n + per_wave - 1) / per_wave;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    int const    slots = (p.pair_share > 0 && p.pair_share < Geo::kGroups) ? Geo::kGroups / p.pair_share : 1;
    if (slots * p.pair_share != Geo::kGroups && slots != 1)
        return hipErrorInvalidValue;
    size_t const lds = ((size_t)slots * (size_t)p.nrows * Geo::kRowDw + (CKPT ? 64 * 8 : 0)) * sizeof(uint32_t);
    hipLaunchKernelGGL((score_pair_kernel<G, C, CKPT>), dim3((unsigned)blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}

// pair geometries: 0 = (8,19) 152 columns, 1 = (8,13) 104, 2 = (8,16) 128, 3 = (8,8) 64, 4 = (8,24) 192, 5 = (16,13) 208,
// 6 = (16,10) 160 [8 extensions per wavefront: for query runs that are a multiple of 8 but not of 16], 7 = (8,25) 200
// [200-residue queries without a padded column: 7.7 against 6.7 TCUPS for (16,13) in pass 1]
hipError_t launch_score_pair(int cfg, ScoreParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    if (p.ckpt) // single sweep: the geometries lx_ckpt.hip's layout is instantiated for
    {
        if (!p.ends || p.steps_cap % 16 != 0)
            return hipErrorInvalidValue;
        return cfg == 0   ? launch_pair_cfg<8, 19, true>(p, stream)
               : cfg == 5 ? launch_pair_cfg<16, 13, true>(p, stream)
               : cfg == 1 ? launch_pair_cfg<8, 13, true>(p, stream) // short queries: 104 columns
               : cfg == 7 ? launch_pair_cfg<8, 25, true>(p, stream) // 153 - 200 columns
                          : hipErrorInvalidValue;
    }
    switch (cfg)
    {
        case 0: return launch_pair_cfg<8, 19>(p, stream);
        case 1: return launch_pair_cfg<8, 13>(p, stream);
        case 2: return launch_pair_cfg<8, 16>(p, stream);
        case 3: return launch_pair_cfg<8, 8>(p, stream);
        case 4: return launch_pair_cfg<8, 24>(p, stream);
        case 5: return launch_pair_cfg<16, 13>(p, stream);
        case 6: return launch_pair_cfg<16, 10>(p, stream);
        case 7: return launch_pair_cfg<8, 25>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

int score_pair_cfg_for(uint32_t max_qlen)
{
    if (max_qlen <= 64)
        return 3;
    if (max_qlen <= 104)
        return 1;
    if (max_qlen <= 128)
        return 2;
    if (max_qlen <= 152)
        return 0;
    if (max_qlen <= 192)
        return 4;
    if (max_qlen <= 200)
        return 7;
    if (max_qlen <= 208)
        return 5;
    return -1;
}

// geometries with 16-lane groups (8 extensions per wavefront)
int score_pair_cfg_for_runs_of_8(uint32_t max_qlen)
{
    if (max_qlen <= 160)
        return 6;
    if (max_qlen <= 208)
        return 5;
    return -1;
}

int score_pair_cfg_cols(int cfg)
{
    static int const c[8] = {19, 13, 16, 8, 24, 13, 10, 25};
    return (cfg >= 0 && cfg < 8) ? c[cfg] : 0;
}

int score_pair_cfg_group(int cfg) { return (cfg == 5 || cfg == 6) ? 16 : 8; }

// LDS bytes of one profile (the kernel's occupancy allows about 13 KB per wavefront)
size_t score_pair_profile_bytes(int cfg, int nrows)
{
    int const G = score_pair_cfg_group(cfg), C = score_pair_cfg_cols(cfg);
    return (size_t)nrows * (size_t)((((C + 1) / 2 + 3) & ~3) * G) * sizeof(uint32_t);
}

} // namespace lx

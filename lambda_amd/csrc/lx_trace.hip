// lx_trace.hip -- pass-2 kernels of the seed extension: forward DP with direction bits + backtrace (gfx950 only).
//
// Replaces _performAlignment<withTrace=true> (/root/reference/src/search_algo.hpp:1296 -> :1070-1134, config
// TracebackOn<CompleteTrace, GapsLeft>, :1083) followed by seqan::_adaptTraceSegmentsTo (:1127) and the walk
// seqan::computeAlignmentStats does over the gapped rows (:1308).  Only survivors of the e-value filter reach
// this pass (:1251-1283), so it sees far fewer cells than pass 1.
//
// Kernel A (trace_forward_kernel): the same strip-systolic geometry and row-skewed recurrence as lx_score.hip
// (G = 16 lanes x C = 10 columns), plus per cell
//     * 4 direction bits:  code (2) = source of H {0 none (H<=0), 1 diagonal, 2 vertical gap F, 3 horizontal gap E}
//       with the priority of a GapsLeft traceback (diagonal > vertical > horizontal; [UPSTREAM-RECALL], see
//       oracle/lx_oracle.c), fx = "F of the row below extends this F" and ex = "E of the next column extends this E"
//       (extension preferred over opening on ties).  One 64-bit word per lane per step, stored coalesced
//       (128 B per group per step) to an HBM trace buffer laid out [extension][panel][step][lane].
//     * the best cell with the reference's tie rule (first strict maximum in column-major order): per column a
//       packed key  H << 16 | (65535 - row)  maximised with v_max_u32 (2 VALU ops per cell), reduced over
//       columns / lanes / panels with "higher H, then lower column" at the end.  Limits: H < 65536, Ls < 65536
//       (checked by the host side).
// Kernel B (backtrace_kernel): one lane per extension walks the direction words from the best cell to the first
// cell with code 0, emits one op byte per alignment column ('M','D','I') and the counts of lx_hsp.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "lx_dp_common.h"

namespace lx
{

constexpr int kTG = 16; // lanes per extension
constexpr int kTC = 10; // columns per lane

__global__ __launch_bounds__(64) void trace_forward_kernel(TraceParams p)
{
    constexpr int G = kTG, C = kTC;
    using Geo = ScoreGeo<G, C>;
    extern __shared__ uint32_t lds[];

    int const  lane     = threadIdx.x;
    int const  grp      = lane / G;
    int const  g        = lane % G;
    bool const is_first = (g == 0);
    bool const is_last  = (g == G - 1);

    uint64_t const e      = (uint64_t)blockIdx.x * Geo::kGroups + grp;
    bool const     active = e < p.n;

    ScoringDev const * __restrict__ sc = p.sc;
    int const      ge    = sc->ge;
    int const      g2    = sc->g2;
    int const      nrows = p.nrows;
    uint32_t const padt  = (uint32_t)(nrows - 1);

    int             lq = 0, ls = 0;
    uint8_t const * q = p.q_res;
    uint8_t const * s = p.s_res;
    if (active)
    {
        Extension const x = p.ext[e];
        lq = (int)x.q_len;
        ls = (int)x.s_len;
        q += x.q_off;
        if (ls != 0)
            s += x.s_off;
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
    npanels = __builtin_amdgcn_readfirstlane(npanels);

    bool bad = false;
    if (active && ((uint32_t)my_panels > p.panels_cap || (uint32_t)((ls + G - 1 + 3) & ~3) > p.steps_cap || ls > 65535))
    {
        bad = true; // the host sized the trace slots too small for this extension: report, never write out of bounds
        atomicExch(p.err, 3);
    }

    int32_t * carry = nullptr;
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
    {
        lq = 0;
        ls = 0;
    }

    int const          slot_dw   = grp * (nrows * Geo::kRowDw);
    uint32_t const     row_base  = (uint32_t)(slot_dw + g) * 4u;
    constexpr uint32_t kRowBytes = Geo::kRowDw * 4;
    int const          steps     = (ls_max + G - 1 + 3) & ~3;
    uint32_t const     lsc       = (uint32_t)max(ls, 1) - 1u;

    uint2 * tr = p.trace + e * p.slot_stride + g; // + (panel * steps_cap + k) * G

    // best cell so far over all finished panels: (H, column+1, row+1)
    int best_h = 0, best_q = 0, best_s = 0;

    for (int panel = 0; panel < npanels; ++panel)
    {
        int const col0 = panel * Geo::kPanel + g * C;
        build_profile<G, C>(lds, slot_dw, g, q, lq, col0, sc, nrows, true);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        bool const use_carry_in = is_first && (panel > 0) && (panel < my_panels) && carry != nullptr;
        bool const do_carry_out = is_last && (panel + 1 < my_panels) && carry != nullptr;
        bool const store_trace  = active && !bad && (panel < my_panels);

        int z = ge * g;
        int Hrow[C], F0[C];
        uint32_t colkey[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            Hrow[c]   = z + ge;
            F0[c]     = z;
            colkey[c] = 0;
        }
        int diag0 = z + ge;
        int sendH = z + ge;
        int sendE = kNegInf;

        uint2 * trp = tr + (uint64_t)panel * p.steps_cap * G;

        auto step = [&](int k, uint32_t t)
        {
            int const        i    = k - g;
            uint32_t const * prow = reinterpret_cast<uint32_t const *>(reinterpret_cast<char const *>(lds) + row_base + t * kRowBytes);
            uint32_t         pw[Geo::kD];
#pragma unroll
            for (int d = 0; d < Geo::kD; ++d)
                pw[d] = prow[d * G];

            int bndH = z, bndE = kNegInf;
            if (use_carry_in && (unsigned)i < (unsigned)ls)
            {
                bndH = carry[2 * i];
                bndE = carry[2 * i + 1];
            }
            int const recvH = shift_from_left<G>(sendH, bndH, is_first);
            int       Ecur  = shift_from_left<G>(sendE, bndE, is_first);

            int dg = diag0;
            diag0  = recvH;

            int const      zn = z - ge;
            // key = ((Hs - z) << 16) | (65535 - row)  ==  (Hs << 16) + K   (mod 2^32)
            uint32_t const K  = (uint32_t)(-z) * 65536u + ((65535u - (uint32_t)i) & 0xffffu);
            uint32_t       wlo = 0, whi = 0;
            int            h   = 0;
#pragma unroll
            for (int c = 0; c < C; ++c)
            {
                int const sub = (int)(int8_t)(pw[c >> 2] >> (8 * (c & 3)));
                int const tt  = dg + sub;
                dg            = Hrow[c];
                int const f   = F0[c];
                h             = max3i(tt, Ecur, f);
                int const A   = h + g2;
                // direction bits
                uint32_t const code = (h == z) ? 0u : ((tt == h) ? 1u : ((f == h) ? 2u : 3u));
                uint32_t const nib  = code | ((f >= A) ? 4u : 0u) | ((Ecur >= A) ? 8u : 0u);
                if (c < 8)
                    wlo |= nib << (4 * c);
                else
                    whi |= nib << (4 * (c - 8));
                F0[c]   = max3i(f, A, zn);
                LX_OPAQUE(F0[c]);
                Ecur    = max(Ecur, A) + ge;
                Hrow[c] = h;
                uint32_t const key = ((uint32_t)h << 16) + K;
                colkey[c]          = max(colkey[c], key);
            }
            sendH = h;
            sendE = Ecur;
            z     = zn;

            if (store_trace)
                trp[(uint32_t)k * G] = make_uint2(wlo, whi);
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
#pragma unroll
            for (int u = 0; u < 4; ++u)
                step(k0 + u, tc[u]);
        }

        // best cell of this panel: higher H wins, ties go to the lower column (then the key's lower row)
        uint32_t bk = 0;
        int      bc = 0;
#pragma unroll
        for (int c = 0; c < C; ++c)
            if ((colkey[c] >> 16) > (bk >> 16))
            {
                bk = colkey[c];
                bc = c;
            }
        int bcol = col0 + bc; // 0-based global column
#pragma unroll
        for (int off = 1; off < G; off <<= 1)
        {
            uint32_t const ok = (uint32_t)__shfl_xor((int)bk, off);
            int const      oc = __shfl_xor(bcol, off);
            bool const     take = ((ok >> 16) > (bk >> 16)) || ((ok >> 16) == (bk >> 16) && oc < bcol);
            bk   = take ? ok : bk;
            bcol = take ? oc : bcol;
        }
        int const ph = (int)(bk >> 16);
        if (ph > best_h) // strict: earlier panels hold the lower columns
        {
            best_h = ph;
            best_q = bcol + 1;
            best_s = (int)(65535u - (bk & 0xffffu)) + 1;
        }

        if (npanels > 1)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (active && is_first)
    {
        EndCell ec;
        ec.score = bad ? -1 : best_h;
        ec.q_end = best_q;
        ec.s_end = best_s;
        ec.pad   = 0;
        p.ends[e] = ec;
    }
}

// One lane per extension.
__global__ __launch_bounds__(64) void backtrace_kernel(TraceParams p)
{
    constexpr int G = kTG, C = kTC;
    constexpr int P = G * C;
    uint64_t const e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.n)
        return;
    EndCell const   ec = p.ends[e];
    Extension const x  = p.ext[e];
    Hsp             out{};
    if (ec.score <= 0)
    {
        out.score = ec.score < 0 ? -1 : 0;
        p.out_hsp[e] = out;
        return;
    }
    uint8_t const * q    = p.q_res + x.q_off;
    uint8_t const * s    = p.s_res + x.s_off;
    int8_t const *  mat  = p.sc->mat;
    uint2 const *   tr   = p.trace + e * p.slot_stride;
    uint8_t *       ops  = p.out_ops + p.ops_off[e];
    uint32_t const  cap  = x.q_len + x.s_len;

    auto nibble = [&](int i, int j) -> uint32_t
    {
        int const   panel = j / P, r = j - panel * P;
        int const   g = r / C, c = r - g * C;
        uint2 const w = tr[((uint64_t)panel * p.steps_cap + (uint32_t)(i + g)) * G + g];
        return ((c < 8 ? w.x >> (4 * c) : w.y >> (4 * (c - 8)))) & 15u;
    };

    int      i = ec.s_end - 1, j = ec.q_end - 1;
    int      st = 0; // 0 = H, 1 = F (vertical), 2 = E (horizontal)
    uint32_t n  = 0;
    int      last = 0; // last op emitted while walking backwards (0 none, 'M', 'D', 'I')
    int32_t  nm = 0, nx = 0, np = 0, go = 0, gx = 0;
    while (i >= 0 && j >= 0)
    {
        if (st == 0)
        {
            uint32_t const code = nibble(i, j) & 3u;
            if (code == 0)
                break;
            if (code == 1)
            {
                uint8_t const c0 = q[j] & (kAlph - 1), c1 = s[i] & (kAlph - 1);
                int const     v  = mat[c0 * kAlph + c1];
                bool const isMatch = p.bs_match_rule ? (v == mat[c0 * kAlph + c0]) : (c0 == c1);
                nm += isMatch;
                nx += !isMatch;
                np += (v > 0);
                ops[cap - 1 - n] = 'M';
                ++n;
                last = 'M';
                --i;
                --j;
            }
            else
                st = (code == 2) ? 1 : 2;
        }
        else if (st == 1)
        {
            ops[cap - 1 - n] = 'D';
            ++n;
            // walking backwards: a gap run [open, ext, ext...] is met from its end
            gx += 1; // provisionally an extension; the run's first character is re-labelled as the open below
            bool const ext = (i >= 1) && ((nibble(i - 1, j) >> 2) & 1u);
            --i;
            if (!ext)
            {
                gx -= 1;
                go += 1;
                st = 0;
            }
            last = 'D';
        }
        else
        {
            ops[cap - 1 - n] = 'I';
            ++n;
            gx += 1;
            bool const ext = (j >= 1) && ((nibble(i, j - 1) >> 3) & 1u);
            --j;
            if (!ext)
            {
                gx -= 1;
                go += 1;
                st = 0;
            }
            last = 'I';
        }
        if (n >= cap)
            break;
    }
    (void)last;
    // move the ops to the front of the slot, in begin -> end order
    uint32_t const first = cap - n;
    if (first != 0)
        for (uint32_t k = 0; k < n; ++k)
            ops[k] = ops[first + k];

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
    p.out_hsp[e]           = out;
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

int trace_panel() { return kTG * kTC; }
int trace_group() { return kTG; }

hipError_t launch_trace(TraceParams const & p, hipStream_t stream)
{
    if (p.n == 0)
        return hipSuccess;
    using Geo = ScoreGeo<kTG, kTC>;
    uint64_t const blocks = (p.n + Geo::kGroups - 1) / Geo::kGroups;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    size_t const lds = (size_t)Geo::kGroups * (size_t)p.nrows * Geo::kRowDw * sizeof(uint32_t);
    hipLaunchKernelGGL(trace_forward_kernel, dim3((unsigned)blocks), dim3(64), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    uint64_t const b2 = (p.n + 63) / 64;
    hipLaunchKernelGGL(backtrace_kernel, dim3((unsigned)b2), dim3(64), 0, stream, p);
    return hipGetLastError();
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

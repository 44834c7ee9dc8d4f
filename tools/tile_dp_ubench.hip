// tile_dp_ubench.hip -- the backtrace's tile DP (lambda_amd/csrc/lx_ckpt.hip, ckpt_backtrace_kernel: the 16 x C tile of direction nibbles
// recomputed from a row checkpoint and the left strip's boundary -- the pass-2 half of /root/reference/src/search_algo.hpp:1296, :1127)
// in two forms, stand-alone, on the same random tiles:
//   int32   the kernel's own rows: values x 4 with the traceback tags in the two low bits, v_max3_i32, one cell per instruction;
//   packed  two ROWS per register (row 2p in the low half at column x, row 2p + 1 in the high half at column x - 1: the two cells of an
//           anti-diagonal are independent), the same tagged arithmetic in 16 bits by v_pk_add_i16 / v_pk_max_i16, the halves of the
//           diagonal and vertical operands brought together by v_perm_b32, the nibbles collected four columns at a time.
// Prints whether the two forms produce the same nibble words (they must: same values, same ties) and what a tile costs in each.
// Build: hipcc --offload-arch=gfx950 -O3 tools/tile_dp_ubench.hip -o tools/tile_dp_ubench.bin   (build it or kill it)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                                                     \
    do                                                                                                                               \
    {                                                                                                                                \
        hipError_t e = (x);                                                                                                          \
        if (e != hipSuccess)                                                                                                         \
        {                                                                                                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                                                   \
            exit(1);                                                                                                                 \
        }                                                                                                                            \
    } while (0)

constexpr int kAlph = 32, C = 19, kRows = 16, kNibDw = (C + 7) / 8;

__device__ __forceinline__ int max3i(int a, int b, int c)
{
    return max(max(a, b), c);
}
__device__ __forceinline__ uint32_t rnd(uint32_t & s)
{
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

struct TileIn
{
    int      Hp[C], F[C];   // top edge: H of the row above x 4, folded F entering the first row (x 4, tag 2)
    int      E[kRows], Hd[kRows]; // left edge per row: E entering column 0 (x 4, tag 1), H of the cell above-left (x 4)
    uint32_t qoff[C];       // LDS row offsets of the strip's query letters
    uint32_t tl[kRows];     // subject letters
};

__device__ void make_tile(uint32_t seed, TileIn & t)
{
    uint32_t s = seed * 2654435761u + 12345u;
    int      h = (int)(rnd(s) & 63);
#pragma unroll
    for (int c = 0; c < C; ++c)
    {
        h       = max(0, h + (int)(rnd(s) & 7) - 4);
        t.Hp[c] = 4 * h;
        t.F[c]  = (4 * max(0, h - (int)(rnd(s) & 15))) | 2;
        t.qoff[c] = (rnd(s) & 15) * kAlph;
    }
    int hl = (int)(rnd(s) & 63);
#pragma unroll
    for (int r = 0; r < kRows; ++r)
    {
        t.Hd[r] = 4 * hl;
        hl      = max(0, hl + (int)(rnd(s) & 7) - 4);
        t.E[r]  = (4 * (hl - 11 - (int)(rnd(s) & 7))) | 1;
        t.tl[r] = rnd(s) & 15;
    }
}

// ---- the kernel's rows (lx_ckpt.hip): returns the XOR of all nibble words, or stores them
template <bool STORE>
__global__ __launch_bounds__(64) void tile_int32(uint32_t * out, int tiles_per_lane, int ge4, int go4)
{
    __shared__ int8_t smat4[kAlph * kAlph];
    for (int x = threadIdx.x; x < kAlph * kAlph; x += 64)
        smat4[x] = (int8_t)(4 * ((int)((x * 7 + (x >> 5) * 3) % 16) - 4) + 3);
    __syncthreads();
    uint32_t const gid = blockIdx.x * 64 + threadIdx.x;
    uint32_t       chk = 0;
    for (int it = 0; it < tiles_per_lane; ++it)
    {
        TileIn t;
        make_tile(gid * 131u + it, t);
        int Hp[C], F[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
            Hp[c] = t.Hp[c];
            F[c]  = t.F[c];
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r)
        {
            int      E = t.E[r], Hd = t.Hd[r];
            uint32_t w[kNibDw] = {0, 0, 0};
            int      sub4[C];
#pragma unroll
            for (int c = 0; c < C; ++c)
                sub4[c] = (int)smat4[t.qoff[c] + t.tl[r]];
#pragma unroll
            for (int c = 0; c < C; ++c)
            {
                int const tt = Hd + sub4[c];
                Hd           = Hp[c];
                int const m  = max3i(tt, E, F[c]);
                int const H4 = m & ~3;
                int const A  = H4 + go4;
                int const Fr = max3i(F[c] + ge4, A, 0);
                int const Er = max(E + ge4, A);
                F[c]         = Fr | 2;
                E            = Er | 1;
                Hp[c]        = H4;
                uint32_t wc  = w[c >> 3];
                wc           = __builtin_amdgcn_alignbit((uint32_t)m, wc, 2);
                wc           = __builtin_amdgcn_alignbit((uint32_t)(Fr | Er), wc, 2);
                w[c >> 3]    = wc;
            }
#pragma unroll
            for (int xw = 0; xw < kNibDw; ++xw)
            {
                if (STORE)
                    out[((size_t)gid * tiles_per_lane + it) * kRows * kNibDw + r * kNibDw + xw] = w[xw];
                chk ^= w[xw] * (uint32_t)(r * kNibDw + xw + 1);
            }
        }
    }
    if (!STORE)
        out[gid] = chk;
}

// ---- two rows per register
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_add_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_lshr4(uint32_t a)
{
    uint32_t d;
    asm("v_pk_lshrrev_b16 %0, %2, %1" : "=v"(d) : "v"(a), "v"(0x00040004u)); // (an inline constant reaches the low half only)
    return d;
}
__device__ __forceinline__ uint32_t pack2(int lo, int hi)
{
    return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
}
// (a.hi, b.lo) -> (lo, hi)
__device__ __forceinline__ uint32_t hi_lo(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_perm(b, a, 0x05040302u); // bytes: a[2], a[3], b[0], b[1]
}

template <bool STORE>
__global__ __launch_bounds__(64) void tile_packed(uint32_t * out, int tiles_per_lane, int ge4, int go4)
{
    __shared__ int8_t smat4[kAlph * kAlph];
    for (int x = threadIdx.x; x < kAlph * kAlph; x += 64)
        smat4[x] = (int8_t)(4 * ((int)((x * 7 + (x >> 5) * 3) % 16) - 4) + 3);
    __syncthreads();
    uint32_t const gid = blockIdx.x * 64 + threadIdx.x;
    uint32_t const GE4 = pack2(ge4, ge4), GO4 = pack2(go4, go4);
    uint32_t       chk = 0;
    for (int it = 0; it < tiles_per_lane; ++it)
    {
        TileIn t;
        make_tile(gid * 131u + it, t);
        // PH[x].hi = H of the row above at column x - 1, PF[x].hi = F entering the pair's first row at column x - 1 (what the pair before
        // left in its registers: its trailing row stands in the high halves)
        uint32_t PH[C + 1], PF[C + 1];
#pragma unroll
        for (int x = 1; x <= C; ++x)
        {
            PH[x] = (uint32_t)t.Hp[x - 1] << 16;
            PF[x] = (uint32_t)t.F[x - 1] << 16;
        }
        PH[0] = PF[0] = 0;
#pragma unroll
        for (int p = 0; p < kRows / 2; ++p)
        {
            int const      a = 2 * p, b = a + 1;
            uint32_t const ta = t.tl[a], tb = t.tl[b];
            uint32_t       RH[C + 1], RF[C + 1];
            uint32_t       E = pack2(t.E[a], 0), acc = 0;
            uint32_t       wa[kNibDw] = {0, 0, 0}, wb[kNibDw] = {0, 0, 0};
#pragma unroll
            for (int x = 0; x <= C; ++x)
            {
                // substitution scores: (column x against row a's letter, column x - 1 against row b's)
                int const sa = x < C ? (int)smat4[t.qoff[x < C ? x : 0] + ta] : 0;
                int const sb = x >= 1 ? (int)smat4[t.qoff[x >= 1 ? x - 1 : 0] + tb] : 0;
                uint32_t const S = pack2(sa, sb);
                // diagonal operand: (H above-left of (a, x), H above-left of (b, x - 1)); vertical operand likewise
                uint32_t D, Fin;
                if (x == 0)
                {
                    D   = pack2(t.Hd[a], 0);
                    Fin = hi_lo(PF[1], 0u);
                }
                else if (x == 1)
                {
                    D   = pack2((int)(PH[1] >> 16), t.Hd[b]);
                    Fin = hi_lo(PF[2], RF[0]);
                    E   = (E & 0xffffu) | ((uint32_t)t.E[b] << 16); // the trailing row enters
                }
                else
                {
                    D   = hi_lo(PH[x], RH[x - 2]);
                    Fin = hi_lo(PF[x < C ? x + 1 : C], RF[x - 1]);
                }
                uint32_t const tt = pk_add(D, S);
                uint32_t const m  = pk_max(pk_max(tt, E), Fin);
                uint32_t const H4 = m & 0xfffcfffcu;
                uint32_t const A  = pk_add(H4, GO4);
                uint32_t const Fr = pk_max(pk_max(pk_add(Fin, GE4), A), 0u);
                uint32_t const Er = pk_max(pk_add(E, GE4), A);
                RF[x]             = Fr | 0x00020002u;
                E                 = Er | 0x00010001u;
                RH[x]             = H4;
                // nibbles of both rows, four columns per half before they go to the rows' words
                acc = pk_lshr4(acc);
                acc = ((m & 0x00030003u) << 12) | acc;
                acc = (((Fr | Er) & 0x00030003u) << 14) | acc;
                // row a has its columns x - 3 .. x complete, row b its columns x - 4 .. x - 1
                auto flush = [&](uint32_t (&w)[kNibDw], int col_last, bool high)
                {
                    // the word that holds col_last takes the 16 bits (4 nibbles, fewer at the strip's end) at its top
                    int const      xw   = col_last >> 3;
                    uint32_t const bits = high ? (acc >> 16) : (acc & 0xffffu);
                    w[xw]               = (w[xw] >> 16) | (bits << 16);
                };
                if (x < C && ((x & 3) == 3 || x == C - 1))
                {
                    if (x == C - 1 && (x & 3) != 3) // a partial group: its nibbles stand at the top of the 16 bits, shift the word by what came
                    {
                        int const      cnt  = (x & 3) + 1;
                        int const      xw   = x >> 3;
                        uint32_t const bits = (acc & 0xffffu) >> (16 - 4 * cnt);
                        wa[xw]              = (wa[xw] >> (4 * cnt)) | (bits << (32 - 4 * cnt));
                    }
                    else
                        flush(wa, x, false);
                }
                if (x >= 1 && (((x - 1) & 3) == 3 || x == C))
                {
                    int const cl = x - 1;
                    if (cl == C - 1 && (cl & 3) != 3)
                    {
                        int const      cnt  = (cl & 3) + 1;
                        int const      xw   = cl >> 3;
                        uint32_t const bits = (acc >> 16) >> (16 - 4 * cnt);
                        wb[xw]              = (wb[xw] >> (4 * cnt)) | (bits << (32 - 4 * cnt));
                    }
                    else
                        flush(wb, cl, true);
                }
            }
#pragma unroll
            for (int x = 0; x <= C; ++x)
            {
                PH[x] = RH[x];
                PF[x] = RF[x];
            }
#pragma unroll
            for (int xw = 0; xw < kNibDw; ++xw)
            {
                if (STORE)
                {
                    out[((size_t)gid * tiles_per_lane + it) * kRows * kNibDw + a * kNibDw + xw] = wa[xw];
                    out[((size_t)gid * tiles_per_lane + it) * kRows * kNibDw + b * kNibDw + xw] = wb[xw];
                }
                chk ^= wa[xw] * (uint32_t)(a * kNibDw + xw + 1);
                chk ^= wb[xw] * (uint32_t)(b * kNibDw + xw + 1);
            }
        }
    }
    if (!STORE)
        out[gid] = chk;
}

int main()
{
    int const ge4 = -4, go4 = -44; // BLOSUM62's gaps: 11 for the first character, 1 for every further one
    // ---- same nibbles?
    {
        int const blocks = 64, per = 4;
        size_t const n   = (size_t)blocks * 64 * per * kRows * kNibDw;
        uint32_t *   da, * db;
        CHECK(hipMalloc(&da, n * 4));
        CHECK(hipMalloc(&db, n * 4));
        hipLaunchKernelGGL(tile_int32<true>, dim3(blocks), dim3(64), 0, 0, da, per, ge4, go4);
        hipLaunchKernelGGL(tile_packed<true>, dim3(blocks), dim3(64), 0, 0, db, per, ge4, go4);
        std::vector<uint32_t> a(n), b(n);
        CHECK(hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, nz = 0;
        // (the last word of a row holds C - 16 = 3 nibbles at its top; compare the nibble bits only)
        for (size_t i = 0; i < n; ++i)
        {
            uint32_t const mask = (i % kNibDw) == kNibDw - 1 ? ~0u << (32 - 4 * (C - 8 * (kNibDw - 1))) : ~0u;
            if (((a[i] ^ b[i]) & mask) != 0 && bad < 6)
                printf("  word %zu (row %zu, word %zu of its row): int32 %08x packed %08x\n", i, (i / kNibDw) % kRows, i % kNibDw, a[i] & mask, b[i] & mask);
            bad += ((a[i] ^ b[i]) & mask) != 0;
            nz += (a[i] & mask) != 0;
        }
        printf("parity: %zu nibble words, %zu non-zero, %zu differ between the int32 rows and the packed row pairs\n", n, nz, bad);
        CHECK(hipFree(da));
        CHECK(hipFree(db));
    }
    // ---- what a tile costs
    int const  blocks = 256 * 8, per = 200;
    uint32_t * d;
    CHECK(hipMalloc(&d, (size_t)blocks * 64 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which)
    {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep)
        {
            CHECK(hipEventRecord(e0, 0));
            if (which == 0)
                hipLaunchKernelGGL(tile_int32<false>, dim3(blocks), dim3(64), 0, 0, d, per, ge4, go4);
            else
                hipLaunchKernelGGL(tile_packed<false>, dim3(blocks), dim3(64), 0, 0, d, per, ge4, go4);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        double const cells = (double)blocks * 64 * per * kRows * C;
        printf("%-7s %8.3f ms for %.2f G tile cells: %.1f G cells/s, %.2f clocks per cell and SIMD lane (2.4 GHz, 16 lanes per clock and SIMD)\n",
               which == 0 ? "int32" : "packed", best, cells / 1e9, cells / best / 1e6, 1024.0 * 16 * 2.4e9 * (best * 1e-3) / cells);
    }
    return 0;
}

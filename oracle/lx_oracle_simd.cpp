/*
 * lx_oracle_simd.cpp -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Inter-sequence int16 SIMD variant of the score-only pass: one alignment per
 * SIMD lane, lanes padded to the longest pair of the group -- the batching design of
 * the reference's CPU path (src/search_algo.hpp:1044-1068 _setupDepSets, :1087
 * SimdVector<int16_t>, :1106-1122 one _prepareAndRunSimdAlignment call per lane group).
 * It exists so that bench.py's cpu_baseline leg times something shaped like the
 * reference's CPU path rather than a scalar loop.  It is a restatement ("port"),
 * NOT SeqAn; tests check it against lxo_score() bit for bit.
 */
extern "C" {
#include "lx_oracle.h"
}

#include <stdlib.h>
#include <string.h>

#define LANES 16
typedef int16_t v16s __attribute__((vector_size(2 * LANES)));
typedef int8_t  v16b __attribute__((vector_size(LANES)));
typedef uint8_t v16u __attribute__((vector_size(LANES)));

#define PAD_RANK 31
#define NEG16 ((int16_t)-16384)

static inline v16s vsel(v16s mask, v16s a, v16s b) /* mask lanes are 0 / -1 */
{
    return mask ? a : b; /* g++ vector ternary -> pblendvb */
}

static inline v16s vmax(v16s a, v16s b)
{
    return a > b ? a : b; /* -> pmaxsw */
}

static inline v16s vsplat(int16_t x)
{
    v16s v;
    for (int l = 0; l < LANES; ++l)
        v[l] = x;
    return v;
}

/* One group of up to LANES alignments. Returns 0, or 1 if the group must be redone in scalar int32. */
__attribute__((target_clones("avx2", "default"))) static int score_group(uint8_t const * qres, uint8_t const * sres, uint64_t const * q_off, uint32_t const * q_len,
            uint64_t const * s_off, uint32_t const * s_len, uint64_t first, int cnt, lxo_scoring const * sc,
            int8_t const * mat /* padded copy */, int32_t * score, int32_t * q_end, int32_t * s_end,
            uint8_t * qT, uint8_t * sT, v16s * Hcol, v16s * Ecol, int32_t lqmax, int32_t lsmax)
{
    /* Do all lanes hold the same query slice?  (The reference sorts matches by query, so SIMD groups usually do.)
     * Then the per-cell 2-D matrix gather becomes a 32-entry byte-table lookup per query column: two byte shuffles
     * (pshufb) instead of 16 scalar loads -- the query-profile trick of SWIPE/SSW, applied across sequences. */
    int same_q = 1;
    for (int l = 1; l < cnt; ++l)
        if (q_off[first + (uint64_t)l] != q_off[first] || q_len[first + (uint64_t)l] != q_len[first])
            same_q = 0;

    /* transpose residues: qT[j*LANES + lane], pad with PAD_RANK (the SimdRep of seqan2_to_biocpp.hpp:397-405) */
    memset(qT, PAD_RANK, (size_t)lqmax * LANES);
    memset(sT, PAD_RANK, (size_t)lsmax * LANES);
    for (int l = 0; l < LANES; ++l)
    {
        uint64_t const x = first + (uint64_t)(l < cnt ? l : cnt - 1); /* repeat last pair (:1063-1067) */
        for (uint32_t j = 0; j < q_len[x]; ++j)
            qT[(size_t)j * LANES + l] = qres[q_off[x] + j] & 31;
        for (uint32_t i = 0; i < s_len[x]; ++i)
            sT[(size_t)i * LANES + l] = sres[s_off[x] + i] & 31;
    }

    v16s const go = vsplat((int16_t)sc->gap_open), ge = vsplat((int16_t)sc->gap_extend);
    v16s const zero = vsplat(0);
    v16s       best = zero, bq = zero, bs = zero;

    for (int32_t i = 0; i <= lsmax; ++i)
    {
        Hcol[i] = zero;
        Ecol[i] = vsplat(NEG16);
    }

    for (int32_t j = 1; j <= lqmax; ++j)
    {
        uint8_t const * qrow = qT + (size_t)(j - 1) * LANES;
        v16s            diag = zero, up = zero, F = vsplat(NEG16);
        v16s const      jv   = vsplat((int16_t)j);
        v16b tlo, thi;
        if (same_q)
        {
            memcpy(&tlo, mat + qrow[0] * LXO_ALPH, 16);
            memcpy(&thi, mat + qrow[0] * LXO_ALPH + 16, 16);
        }
        for (int32_t i = 1; i <= lsmax; ++i)
        {
            uint8_t const * srow = sT + (size_t)(i - 1) * LANES;
            v16s            sub;
            if (same_q)
            {
                v16u idx;
                memcpy(&idx, srow, 16);
                v16b const lo  = __builtin_shuffle(tlo, (v16b)(idx & 15));
                v16b const hi  = __builtin_shuffle(thi, (v16b)(idx & 15));
                v16b const sel = (v16b)(idx > 15);
                v16b const s8  = sel ? hi : lo;
                sub            = __builtin_convertvector(s8, v16s);
            }
            else
                for (int l = 0; l < LANES; ++l)
                    sub[l] = mat[qrow[l] * LXO_ALPH + srow[l]];
            v16s const left = Hcol[i];
            v16s const E    = vmax(Ecol[i] + ge, left + go);
            F               = vmax(F + ge, up + go);
            v16s h          = vmax(vmax(diag + sub, zero), vmax(E, F));
            v16s const gt   = (h > best);
            best            = vsel(gt, h, best);
            bq              = vsel(gt, jv, bq);
            bs              = vsel(gt, vsplat((int16_t)i), bs);
            diag            = left;
            Hcol[i]         = h;
            Ecol[i]         = E;
            up              = h;
        }
    }
    for (int l = 0; l < cnt; ++l)
    {
        score[first + (uint64_t)l] = best[l];
        if (q_end)
            q_end[first + (uint64_t)l] = bq[l];
        if (s_end)
            s_end[first + (uint64_t)l] = bs[l];
    }
    return 0;
}

/* Inter-sequence SIMD batch scorer. The caller is expected to have sorted the pairs by
 * (q_len, s_len) like src/search_algo.hpp:1229-1235 does, to minimise padding. */
extern "C" int lxo_score_batch_simd(uint8_t const * qres, uint8_t const * sres, uint64_t const * q_off, uint32_t const * q_len,
                         uint64_t const * s_off, uint32_t const * s_len, uint64_t n, lxo_scoring const * sc,
                         int32_t * score, int32_t * q_end, int32_t * s_end, int32_t threads)
{
    int8_t mat[LXO_ALPH * LXO_ALPH];
    int    maxs = 0;
    for (int a = 0; a < LXO_ALPH; ++a)
        for (int b = 0; b < LXO_ALPH; ++b)
        {
            int8_t v = sc->matrix[a * LXO_ALPH + b];
            if (a >= sc->alphabet_size || b >= sc->alphabet_size)
                v = -100; /* pad rank: can never start, extend or end a best-scoring local alignment */
            mat[a * LXO_ALPH + b] = v;
            if (v > maxs)
                maxs = v;
        }
    uint64_t const groups = (n + LANES - 1) / LANES;
    int            rc     = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
#endif
    {
        size_t   capq = 0, caps = 0;
        uint8_t *qT = NULL, *sT = NULL;
        v16s *   Hcol = NULL, *Ecol = NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t g = 0; g < (int64_t)groups; ++g)
        {
            uint64_t const first = (uint64_t)g * LANES;
            int const      cnt   = (int)((n - first) < LANES ? (n - first) : LANES);
            int32_t        lqmax = 0, lsmax = 0;
            for (int l = 0; l < cnt; ++l)
            {
                if ((int32_t)q_len[first + l] > lqmax)
                    lqmax = (int32_t)q_len[first + l];
                if ((int32_t)s_len[first + l] > lsmax)
                    lsmax = (int32_t)s_len[first + l];
            }
            int const fits16 = ((int64_t)maxs * (lqmax < lsmax ? lqmax : lsmax) < 16000) && lqmax < 32000 && lsmax < 32000;
            if (!fits16 || lqmax == 0 || lsmax == 0)
            {
                for (int l = 0; l < cnt; ++l)
                {
                    uint64_t const x = first + (uint64_t)l;
                    int32_t        sco, qe, se;
                    lxo_score(qres + q_off[x], (int32_t)q_len[x], sres + s_off[x], (int32_t)s_len[x], sc, &sco, &qe, &se);
                    score[x] = sco;
                    if (q_end)
                        q_end[x] = qe;
                    if (s_end)
                        s_end[x] = se;
                }
                continue;
            }
            if ((size_t)lqmax > capq)
            {
                capq = (size_t)lqmax * 2;
                free(qT);
                qT = (uint8_t *)aligned_alloc(64, ((capq * LANES + 63) / 64) * 64);
            }
            if ((size_t)lsmax > caps)
            {
                caps = (size_t)lsmax * 2;
                free(sT);
                free(Hcol);
                free(Ecol);
                sT   = (uint8_t *)aligned_alloc(64, ((caps * LANES + 63) / 64) * 64);
                Hcol = (v16s *)aligned_alloc(64, (caps + 1) * sizeof(v16s));
                Ecol = (v16s *)aligned_alloc(64, (caps + 1) * sizeof(v16s));
            }
            if (!qT || !sT || !Hcol || !Ecol)
            {
                rc = -1;
                continue;
            }
            score_group(qres, sres, q_off, q_len, s_off, s_len, first, cnt, sc, mat, score, q_end, s_end, qT, sT,
                        Hcol, Ecol, lqmax, lsmax);
        }
        free(qT);
        free(sT);
        free(Hcol);
        free(Ecol);
    }
    return rc;
}

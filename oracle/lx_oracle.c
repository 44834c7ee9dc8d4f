/*
 * lx_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * CPU restatement of lambda3's seed-extension hot path (see lx_oracle.h for the
 * "parity unpinned" statement).  Plain C11, scalar int32 arithmetic, written for
 * clarity, not speed.  Every function cites the reference lines it follows
 * (paths relative to /root/reference).
 *
 * Conventions restated from the reference:
 *   - horizontal / outer sequence = query slice, vertical / inner = subject slice
 *       (src/search_algo.hpp:1058-1059, :1220-1221)
 *   - LocalAlignment_, BandOff, AffineGaps, TracebackOff | CompleteTrace+GapsLeft
 *       (src/search_algo.hpp:1079-1085; src/search_datastructures.hpp:434)
 *   - scoreGapOpen := gapOpen + gapExtend, scoreGapExtend := gapExtend, i.e. a gap
 *     of length k costs gapOpen + k*gapExtend (src/search_algo.hpp:226-230)
 *
 * [UPSTREAM-RECALL] (SeqAn2 internals, source absent, restated from memory):
 *   - best cell: updated on strict '>' while iterating column by column (query
 *     outer, subject inner) -> first maximum in column-major order
 *   - cells with H <= 0 are reset to 0 and carry trace NONE
 *   - traceback with GapsLeft prefers DIAGONAL, then the vertical gap matrix
 *     (extension before open), then the horizontal gap matrix (extension before open)
 */
#include "lx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-(1 << 29))

static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

static inline int32_t sub_score(lxo_scoring const * sc, uint8_t qr, uint8_t sr)
{
    return sc->matrix[(qr & 31) * LXO_ALPH + (sr & 31)];
}

/* --------------------------------------------------------------------------
 * score-only pass: src/search_algo.hpp:1246 -> :1116-1129 (TracebackOff)
 * -------------------------------------------------------------------------- */
int lxo_score(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
              int32_t * score, int32_t * q_end, int32_t * s_end)
{
    int32_t best = 0, bq = 0, bs = 0;
    if (lq > 0 && ls > 0)
    {
        int32_t const go = sc->gap_open, ge = sc->gap_extend;
        /* per subject row i: H[i][j-1] and E[i][j] carrier (horizontal gap = gap in the subject row) */
        int32_t * Hcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ls + 1));
        int32_t * Ecol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ls + 1));
        if (!Hcol || !Ecol)
        {
            free(Hcol);
            free(Ecol);
            return -1;
        }
        for (int32_t i = 0; i <= ls; ++i)
        {
            Hcol[i] = 0;
            Ecol[i] = NEG_INF;
        }
        for (int32_t j = 1; j <= lq; ++j) /* outer: query columns */
        {
            int32_t diag = 0;       /* H[0][j-1] */
            int32_t up   = 0;       /* H[i-1][j], starts as H[0][j] = 0 */
            int32_t F    = NEG_INF; /* vertical gap state, F[0][j] = -inf */
            for (int32_t i = 1; i <= ls; ++i) /* inner: subject rows */
            {
                int32_t const left = Hcol[i]; /* H[i][j-1] */
                int32_t const E    = imax(Ecol[i] + ge, left + go);
                F                  = imax(F + ge, up + go);
                int32_t h          = diag + sub_score(sc, q[j - 1], s[i - 1]);
                h                  = imax(h, imax(E, F));
                if (h <= 0)
                    h = 0;
                if (h > best) /* strict: first maximum in column-major order */
                {
                    best = h;
                    bq   = j;
                    bs   = i;
                }
                diag    = left;
                Hcol[i] = h;
                Ecol[i] = E;
                up      = h;
            }
        }
        free(Hcol);
        free(Ecol);
    }
    if (score)
        *score = best;
    if (q_end)
        *q_end = bq;
    if (s_end)
        *s_end = bs;
    return 0;
}

/* Banded variant: only cells with diag_lo <= (i - j) <= diag_hi (0-based i over subject, j over query)
 * are computed; everything outside is -inf / unreachable.  NOT a parity mode. */
int lxo_score_banded(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
                     int32_t diag_lo, int32_t diag_hi, int32_t * score)
{
    int32_t best = 0;
    if (lq > 0 && ls > 0)
    {
        int32_t const go = sc->gap_open, ge = sc->gap_extend;
        int32_t *     H  = (int32_t *)calloc((size_t)(ls + 1) * (size_t)(lq + 1), sizeof(int32_t));
        int32_t *     E  = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ls + 1) * (size_t)(lq + 1));
        int32_t *     F  = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ls + 1) * (size_t)(lq + 1));
        if (!H || !E || !F)
        {
            free(H);
            free(E);
            free(F);
            return -1;
        }
        size_t const W = (size_t)lq + 1;
        for (size_t x = 0; x < (size_t)(ls + 1) * W; ++x)
            E[x] = F[x] = NEG_INF;
        for (int32_t j = 1; j <= lq; ++j)
            for (int32_t i = 1; i <= ls; ++i)
            {
                int32_t const d = (i - 1) - (j - 1);
                if (d < diag_lo || d > diag_hi)
                    continue; /* H stays 0 but is unreachable: neighbours read it as a fresh start, same as local */
                int32_t e = imax(E[i * W + j - 1] + ge, H[i * W + j - 1] + go);
                int32_t f = imax(F[(i - 1) * W + j] + ge, H[(i - 1) * W + j] + go);
                /* out-of-band neighbours must not feed gaps */
                if (d + 1 > diag_hi)
                    e = NEG_INF;
                if (d - 1 < diag_lo)
                    f = NEG_INF;
                int32_t h = H[(i - 1) * W + j - 1] + sub_score(sc, q[j - 1], s[i - 1]);
                h         = imax(h, imax(e, f));
                if (h <= 0)
                    h = 0;
                H[i * W + j] = h;
                E[i * W + j] = e;
                F[i * W + j] = f;
                if (h > best)
                    best = h;
            }
        free(H);
        free(E);
        free(F);
    }
    *score = best;
    return 0;
}

/* --------------------------------------------------------------------------
 * traceback pass: src/search_algo.hpp:1296 -> :1116-1127 (CompleteTrace, GapsLeft)
 * Full matrices are kept; the walk reads scores, not stored direction bits.
 * -------------------------------------------------------------------------- */
static int align_impl(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
                      int32_t diag_lo, int32_t diag_hi, lxo_hsp * out, uint8_t * ops)
{
    memset(out, 0, sizeof(*out));
    if (lq <= 0 || ls <= 0)
        return 0;

    int32_t const go = sc->gap_open, ge = sc->gap_extend;
    size_t const  W  = (size_t)lq + 1;
    size_t const  N  = ((size_t)ls + 1) * W;
    int32_t *     H  = (int32_t *)malloc(sizeof(int32_t) * N);
    int32_t *     E  = (int32_t *)malloc(sizeof(int32_t) * N);
    int32_t *     F  = (int32_t *)malloc(sizeof(int32_t) * N);
    if (!H || !E || !F)
    {
        free(H);
        free(E);
        free(F);
        return -1;
    }
    for (size_t x = 0; x < N; ++x)
    {
        H[x] = 0;
        E[x] = NEG_INF;
        F[x] = NEG_INF;
    }

    int32_t best = 0, bq = 0, bs = 0;
    for (int32_t j = 1; j <= lq; ++j)
        for (int32_t i = 1; i <= ls; ++i)
        {
            /* band mode (not the reference's configuration): cells off the band are never computed -- H = 0, no gap
             * state -- and a neighbour off the band feeds no gap (same rules as lxo_score_banded) */
            int32_t const d = (i - 1) - (j - 1);
            if (d < diag_lo || d > diag_hi)
                continue;
            int32_t e = imax(E[i * W + j - 1] + ge, H[i * W + j - 1] + go);
            int32_t f = imax(F[(i - 1) * W + j] + ge, H[(i - 1) * W + j] + go);
            if (d + 1 > diag_hi)
                e = NEG_INF;
            if (d - 1 < diag_lo)
                f = NEG_INF;
            int32_t h = H[(i - 1) * W + j - 1] + sub_score(sc, q[j - 1], s[i - 1]);
            h         = imax(h, imax(e, f));
            if (h <= 0)
                h = 0;
            H[i * W + j] = h;
            E[i * W + j] = e;
            F[i * W + j] = f;
            if (h > best)
            {
                best = h;
                bq   = j;
                bs   = i;
            }
        }

    out->score = best;
    if (best > 0)
    {
        /* walk back; ops are produced end->begin and reversed afterwards */
        int32_t i = bs, j = bq, n = 0;
        enum
        {
            ST_H,
            ST_F,
            ST_E
        } st = ST_H;
        for (;;)
        {
            if (st == ST_H)
            {
                int32_t const h = H[i * W + j];
                if (h <= 0) /* trace NONE */
                    break;
                if (h == H[(i - 1) * W + j - 1] + sub_score(sc, q[j - 1], s[i - 1]))
                {
                    ops[n++] = 'M';
                    --i;
                    --j;
                }
                else if (h == F[i * W + j])
                    st = ST_F;
                else
                    st = ST_E;
            }
            else if (st == ST_F) /* vertical: subject residue i against a gap in the query row */
            {
                ops[n++]             = 'D';
                int32_t const f      = F[i * W + j];
                int const     extend = (f == F[(i - 1) * W + j] + ge);
                --i;
                st = extend ? ST_F : ST_H;
            }
            else /* ST_E, horizontal: query residue j against a gap in the subject row */
            {
                ops[n++]             = 'I';
                int32_t const e      = E[i * W + j];
                int const     extend = (e == E[i * W + j - 1] + ge);
                --j;
                st = extend ? ST_E : ST_H;
            }
        }
        for (int32_t a = 0, b = n - 1; a < b; ++a, --b)
        {
            uint8_t t = ops[a];
            ops[a]    = ops[b];
            ops[b]    = t;
        }
        out->q_begin = j;
        out->s_begin = i;
        out->q_end   = bq;
        out->s_end   = bs;
        out->n_ops   = n;
    }
    free(H);
    free(E);
    free(F);
    return 0;
}

int lxo_align(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
              lxo_hsp * out, uint8_t * ops)
{
    return align_impl(q, lq, s, ls, sc, INT32_MIN / 2, INT32_MAX / 2, out, ops); /* BandOff, src/search_algo.hpp:1081 */
}

/* The same with the band of lxo_score_banded: diag_lo <= (i - j) <= diag_hi.  NOT a parity mode. */
int lxo_align_banded(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
                     int32_t diag_lo, int32_t diag_hi, lxo_hsp * out, uint8_t * ops)
{
    return align_impl(q, lq, s, ls, sc, diag_lo, diag_hi, out, ops);
}

int lxo_score_batch(uint8_t const * qres, uint8_t const * sres, uint64_t const * q_off, uint32_t const * q_len,
                    uint64_t const * s_off, uint32_t const * s_len, uint64_t n, lxo_scoring const * sc,
                    int32_t * score, int32_t * q_end, int32_t * s_end, int32_t threads)
{
    int rc = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t x = 0; x < (int64_t)n; ++x)
    {
        int32_t sco = 0, qe = 0, se = 0;
        int     r   = lxo_score(qres + q_off[x], (int32_t)q_len[x], sres + s_off[x], (int32_t)s_len[x], sc, &sco, &qe, &se);
        if (r)
            rc = r;
        score[x] = sco;
        if (q_end)
            q_end[x] = qe;
        if (s_end)
            s_end[x] = se;
    }
    return rc;
}

/* --------------------------------------------------------------------------
 * src/search_misc.hpp:46-50
 * -------------------------------------------------------------------------- */
int64_t lxo_band_size(uint64_t len)
{
    return (int64_t)sqrt((double)len) + 1;
}

/* src/search_algo.hpp:919-938 */
void lxo_widen_match(lxo_match * m, uint64_t qlen, uint64_t slen)
{
    m->subjStart = (m->subjStart < m->qryStart) ? 0 : m->subjStart - m->qryStart;
    m->qryStart  = 0;
    m->qryEnd    = qlen;
    uint64_t band = (uint64_t)lxo_band_size(qlen);
    uint64_t e    = m->subjStart + qlen + band;
    m->subjEnd    = e < slen ? e : slen;
    m->subjStart  = (band < m->subjStart) ? m->subjStart - band : 0;
}

static int match_cmp(void const * a, void const * b)
{
    uint64_t const * x = (uint64_t const *)a;
    uint64_t const * y = (uint64_t const *)b;
    for (int k = 0; k < 6; ++k) /* defaulted <=> compares members in declaration order (search_datastructures.hpp:60) */
    {
        if (x[k] < y[k])
            return -1;
        if (x[k] > y[k])
            return 1;
    }
    return 0;
}

/* src/search_algo.hpp:1136-1175 */
uint64_t lxo_widen_and_preprocess(lxo_match * m, uint64_t n, uint64_t const * qlens, uint64_t const * slens)
{
    for (uint64_t x = 0; x < n; ++x)
        lxo_widen_match(&m[x], qlens[m[x].qryId], slens[m[x].subjId]);

    qsort(m, n, sizeof(lxo_match), match_cmp);

    if (n > 1)
    {
        /* pairwise merge from left to right (:1147-1158) */
        for (uint64_t x = 0; x + 1 < n; ++x)
        {
            lxo_match * l = &m[x];
            lxo_match * r = &m[x + 1];
            if (l->qryId == r->qryId && l->subjId == r->subjId && l->subjEnd >= r->subjStart)
            {
                l->subjEnd   = r->subjEnd;
                r->subjStart = l->subjStart;
            }
        }
        /* pairwise "swallow" from right to left (:1160-1169) */
        for (uint64_t x = n - 1; x >= 1; --x)
        {
            lxo_match * r = &m[x];
            lxo_match * l = &m[x - 1];
            if (r->qryId == l->qryId && r->subjId == l->subjId && r->subjStart < l->subjEnd)
                *l = *r;
        }
        /* std::ranges::unique (:1171-1172) */
        uint64_t w = 0;
        for (uint64_t x = 0; x < n; ++x)
            if (w == 0 || match_cmp(&m[w - 1], &m[x]) != 0)
                m[w++] = m[x];
        n = w;
    }
    return n;
}

/* --------------------------------------------------------------------------
 * src/search_algo.hpp:426-481
 * -------------------------------------------------------------------------- */
int lxo_seed_looks_promising(uint8_t const * q, int64_t qlen, uint8_t const * s, int64_t slen, int64_t qry_start,
                             int64_t qry_end, int64_t subj_start, int64_t seed_length, int32_t pre_scoring,
                             double pre_scoring_thresh, lxo_scoring const * sc)
{
    int64_t  effQ   = qry_start;
    int64_t  effS   = subj_start;
    uint64_t actual = (uint64_t)(qry_end - qry_start);
    uint64_t effLen = (uint64_t)(seed_length * pre_scoring);
    if (effLen < actual)
        effLen = actual;

    if (effLen > actual)
    {
        effQ -= (int64_t)((effLen - actual) / 2);
        effS -= (int64_t)((effLen - actual) / 2);
        int64_t mn = effQ < effS ? effQ : effS;
        if (mn < 0)
        {
            effQ -= mn;
            effS -= mn;
            effLen += (uint64_t)mn;
        }
        uint64_t a = (uint64_t)(qlen - effQ), b = (uint64_t)(slen - effS);
        if (a < effLen)
            effLen = a;
        if (b < effLen)
            effLen = b;
    }

    int       sco = 0, mx = 0;
    int const thresh = (int)(pre_scoring_thresh * (double)effLen);
    for (uint64_t i = 0; i < effLen; ++i)
    {
        sco += sub_score(sc, q[effQ + (int64_t)i], s[effS + (int64_t)i]);
        if (sco < 0)
            sco = 0;
        else if (sco > mx)
            mx = sco;
        if (mx >= thresh)
            return 1;
    }
    return 0;
}

/* --------------------------------------------------------------------------
 * BLAST statistics. Call sites: src/search_misc.hpp:73, :77-78; src/search_algo.hpp:1258, :1319.
 * [UPSTREAM-RECALL] seqan/blast/blast_statistics.h: a port of NCBI BLAST_ComputeLengthAdjustment
 * with the number of database sequences fixed to 1.
 * -------------------------------------------------------------------------- */
uint64_t lxo_length_adjustment(uint64_t db_len, uint64_t q_len, lxo_karlin const * ka)
{
    double const K             = ka->K;
    double const logK          = log(K);
    double const alphaByLambda = ka->alpha / ka->lambda;
    double const beta          = ka->beta;
    int const    maxIterations = 20;

    double n = (double)db_len;
    double m = (double)q_len;
    double totalLen;
    double val = 0, val_min = 0, val_max;
    int    converged = 0;

    {
        double mb = m + n;
        double c  = n * m - (m > n ? m : n) / K;
        if (c < 0)
            return 0;
        val_max = 2 * c / (mb + sqrt(mb * mb - 4 * c));
    }

    for (int i = 1; i <= maxIterations; ++i)
    {
        totalLen       = (m - val) * (n - val);
        double val_new = alphaByLambda * (logK + log(totalLen)) + beta;
        if (val_new >= val)
        {
            val_min = val;
            if (val_new - val_min <= 1.0)
            {
                converged = 1;
                break;
            }
            if (val_min == val_max)
                break;
        }
        else
        {
            val_max = val;
        }
        if (val_min <= val_new && val_new <= val_max)
            val = val_new;
        else
            val = (i == 1) ? val_max : (val_min + val_max) / 2;
    }

    if (converged)
    {
        val = ceil(val_min);
        if (val <= val_max)
        {
            totalLen = (m - val) * (n - val);
            if (alphaByLambda * (logK + log(totalLen)) + beta >= val)
                return (uint64_t)val;
        }
        return (uint64_t)val_min;
    }
    return (uint64_t)val_min;
}

double lxo_evalue(int32_t score, uint64_t q_len_adj, uint64_t db_len_adj, lxo_karlin const * ka)
{
    return ka->K * (double)q_len_adj * (double)db_len_adj * exp(-ka->lambda * (double)score);
}

double lxo_bitscore(int32_t score, lxo_karlin const * ka)
{
    return (ka->lambda * (double)score - log(ka->K)) / log(2.0);
}

/* --------------------------------------------------------------------------
 * seqan::computeAlignmentStats as called at src/search_algo.hpp:1308; the in-tree
 * bisulfite overload (src/evaluate_bisulfite_alignment.hpp:26-117) shows its shape.
 * bisulfite_match_rule != 0: match iff score(c0,c1)==score(c0,c0) (:97); else rank equality.
 * -------------------------------------------------------------------------- */
int lxo_alignment_stats(uint8_t const * q, uint8_t const * s, lxo_hsp const * hsp, uint8_t const * ops,
                        lxo_scoring const * sc, int32_t bisulfite_match_rule, lxo_align_stats * out)
{
    memset(out, 0, sizeof(*out));
    int32_t qi = hsp->q_begin, si = hsp->s_begin;
    int     gap0 = 0, gap1 = 0; /* gap open in row0 (query row) / row1 (subject row) */
    for (int32_t x = 0; x < hsp->n_ops; ++x)
    {
        uint8_t const op = ops[x];
        if (op == 'D') /* isGap(it0) */
        {
            if (!gap0)
            {
                out->num_gap_opens += 1;
                out->alignment_score += sc->gap_open;
            }
            else
            {
                out->num_gap_extensions += 1;
                out->alignment_score += sc->gap_extend;
            }
            out->num_deletions += 1;
            gap0 = 1;
        }
        else
            gap0 = 0;

        if (op == 'I') /* isGap(it1) */
        {
            if (!gap1)
            {
                out->num_gap_opens += 1;
                out->alignment_score += sc->gap_open;
            }
            else
            {
                out->num_gap_extensions += 1;
                out->alignment_score += sc->gap_extend;
            }
            out->num_insertions += 1;
            gap1 = 1;
        }
        else
            gap1 = 0;

        if (op == 'M')
        {
            uint8_t const c0 = q[qi], c1 = s[si];
            int32_t const v  = sub_score(sc, c0, c1);
            out->alignment_score += v;
            int const isMatch = bisulfite_match_rule ? (v == sub_score(sc, c0, c0)) : (c0 == c1);
            out->num_matches += isMatch;
            out->num_mismatches += !isMatch;
            out->num_positives += (v > 0);
            out->num_negatives += !(v > 0);
            ++qi;
            ++si;
        }
        else if (op == 'D')
            ++si;
        else if (op == 'I')
            ++qi;
        else
            return -1;
    }
    if (qi != hsp->q_end || si != hsp->s_end)
        return -2;
    out->alignment_length = hsp->n_ops;
    if (hsp->n_ops > 0)
    {
        out->similarity = 100.0f * (float)out->num_positives / (float)out->alignment_length;
        out->identity   = 100.0f * (float)out->num_matches / (float)out->alignment_length;
    }
    return 0;
}


/* ---- frames and translation ------------------------------------------------------------------------- */

/* _setFrames, /root/reference/src/search_algo.hpp:768-814 */
int32_t lxo_frame_of(int mode, uint64_t id, int is_subject)
{
    if (mode == 2) /* qIsTranslated / sIsTranslated, :772-776, :795-799 */
    {
        int32_t f = (int32_t)(id % 3) + 1;
        if (id % 6 > 2)
            f = -f;
        return f;
    }
    if (mode == 3) /* DNA3BS, :778-782, :801-803 */
    {
        int32_t f = (int32_t)(id % 2) + 1;
        if (!is_subject && id % 4 > 1)
            f = -f;
        return f;
    }
    if (mode == 1) /* qHasRevComp / sHasRevComp, :784-788, :805-809 */
        return (id % 2) ? -1 : 1;
    return 0;
}

/* _untrueQryId / _untrueSubjId, src/search_algo.hpp:940-996 */
uint64_t lxo_untrue_id(int mode, uint64_t n_id, int32_t frame, int is_subject)
{
    if (mode == 2)
        return frame > 0 ? n_id * 6 + (uint64_t)frame - 1 : n_id * 6 + (uint64_t)(-frame) + 2;
    if (mode == 3 && !is_subject)
        return frame > 0 ? n_id * 4 : n_id * 4 + 2;
    if (mode == 1 || mode == 3)
        return frame > 0 ? n_id * 2 : n_id * 2 + 1;
    return n_id;
}

/* The canonical genetic code spelled out codon by codon in A, C, G, T order (index 16 a + 4 b + c); independent of
 * the T, C, A, G string the product uses. */
static char const lxo_codons[64] = {
    /* AAA */ 'K', 'N', 'K', 'N', /* ACA */ 'T', 'T', 'T', 'T', /* AGA */ 'R', 'S', 'R', 'S', /* ATA */ 'I', 'I', 'M', 'I',
    /* CAA */ 'Q', 'H', 'Q', 'H', /* CCA */ 'P', 'P', 'P', 'P', /* CGA */ 'R', 'R', 'R', 'R', /* CTA */ 'L', 'L', 'L', 'L',
    /* GAA */ 'E', 'D', 'E', 'D', /* GCA */ 'A', 'A', 'A', 'A', /* GGA */ 'G', 'G', 'G', 'G', /* GTA */ 'V', 'V', 'V', 'V',
    /* TAA */ '*', 'Y', '*', 'Y', /* TCA */ 'S', 'S', 'S', 'S', /* TGA */ '*', 'C', 'W', 'C', /* TTA */ 'L', 'F', 'L', 'F'};

static uint8_t lxo_aa_rank(char c)
{
    static char const order[] = "ABCDEFGHIJKLMNOPQRSTUVWYZX*"; /* SeqAn AminoAcid order, src/seqan2_to_biocpp.hpp:352-366 */
    uint8_t r = 0;
    while (order[r] != c)
        ++r;
    return r;
}

/* one codon in A,C,G,T indices 0..3, 4 = N: the amino acid every completion agrees on, else X */
static uint8_t lxo_translate_codon(int a, int b, int c)
{
    int  first = -1;
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y)
            for (int z = 0; z < 4; ++z)
            {
                if ((a != 4 && a != x) || (b != 4 && b != y) || (c != 4 && c != z))
                    continue;
                int const aa = lxo_codons[16 * x + 4 * y + z];
                if (first < 0)
                    first = aa;
                else if (first != aa)
                    return lxo_aa_rank('X');
            }
    return lxo_aa_rank((char)first);
}

uint64_t lxo_translate_frame(uint8_t const * dna5, uint64_t n, int frame, uint8_t * out)
{
    /* BioC++ dna5 rank A,C,G,N,T -> A,C,G,T index with N = 4 */
    static int const acgt[5] = {0, 1, 2, 4, 3};
    int const      shift = (frame > 0 ? frame : -frame) - 1;
    uint64_t const len   = n >= (uint64_t)shift ? (n - (uint64_t)shift) / 3 : 0;
    for (uint64_t k = 0; k < len; ++k)
    {
        int b[3];
        for (int t = 0; t < 3; ++t)
        {
            uint64_t const p = (uint64_t)shift + 3 * k + (uint64_t)t; /* position on the strand that is read */
            if (frame > 0)
                b[t] = acgt[dna5[p]];
            else
            {
                int const x = acgt[dna5[n - 1 - p]]; /* reverse strand: complement of the mirrored position */
                b[t]        = x == 4 ? 4 : 3 - x;    /* A<->T, C<->G in A,C,G,T indices */
            }
        }
        out[k] = lxo_translate_codon(b[0], b[1], b[2]);
    }
    return len;
}

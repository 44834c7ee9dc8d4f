/*
 * lx_oracle.h -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * CPU restatement (plain C, no dependencies) of lambda3's seed-extension hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / timed CPU baseline.
 *
 * PARITY UNPINNED: the reference cannot be compiled in this environment (its
 * DP core lives in SeqAn2, an un-vendored, empty git submodule) and all of its
 * golden files are remote-only (test/data/datasources.cmake:7).  This oracle is
 * therefore a restatement of the algorithm implied by the reference's call
 * sites and configuration types; it is cross-checked against an independent
 * O(n^3) general-gap Smith-Waterman (tests/brute.py), not against reference
 * output.
 *
 * All citations are relative to /root/reference.
 */
#ifndef LX_ORACLE_H
#define LX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LXO_ALPH 32

/* Same layout as include/lambda_ext.h's lx_scoring (kept separate on purpose:
 * the oracle must not include product headers). */
typedef struct lxo_scoring
{
    int32_t alphabet_size; /* ranks 0..alphabet_size-1 are valid residues        */
    int32_t gap_open;      /* SeqAn scoreGapOpen = lambda gapOpen+gapExtend (<0)  */
    int32_t gap_extend;    /* SeqAn scoreGapExtend = lambda gapExtend (<0)        */
    int32_t reserved;
    int8_t  matrix[LXO_ALPH * LXO_ALPH]; /* [query_rank*32 + subject_rank]       */
} lxo_scoring;

/* src/search_datastructures.hpp:46-61 */
typedef struct lxo_match
{
    uint64_t qryId, subjId, qryStart, qryEnd, subjStart, subjEnd;
} lxo_match;

typedef struct lxo_hsp
{
    int32_t score;
    int32_t q_begin, q_end; /* 0-based half-open, relative to the query slice   */
    int32_t s_begin, s_end; /* 0-based half-open, relative to the subject slice */
    int32_t n_ops;          /* number of alignment columns                      */
} lxo_hsp;

/* src/search_datastructures.hpp (seqan::AlignmentStats as filled at search_algo.hpp:1308) */
typedef struct lxo_align_stats
{
    int32_t num_matches, num_mismatches, num_positives, num_negatives;
    int32_t num_gap_opens, num_gap_extensions, num_insertions, num_deletions;
    int32_t alignment_length, alignment_score;
    float   identity, similarity;
} lxo_align_stats;

typedef struct lxo_karlin
{
    double lambda, K, H, alpha, beta;
} lxo_karlin;

/* ---- the DP (src/search_algo.hpp:1070-1134 -> seqan::_prepareAndRunSimdAlignment) ---- */
int lxo_score(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
              int32_t * score, int32_t * q_end, int32_t * s_end);

/* banded variant (|i-j-shift| <= band around the window's main diagonal); not a parity mode */
int lxo_score_banded(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
                     int32_t diag_lo, int32_t diag_hi, int32_t * score);

/* ops: 'M' diagonal, 'D' gap in query row (subject residue consumed), 'I' gap in subject row.
 * ops must hold lq+ls bytes. */
int lxo_align(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
              lxo_hsp * out, uint8_t * ops);
/* with the band of lxo_score_banded (diag_lo <= i - j <= diag_hi, 0-based i over subject, j over query); not a parity mode */
int lxo_align_banded(uint8_t const * q, int32_t lq, uint8_t const * s, int32_t ls, lxo_scoring const * sc,
                     int32_t diag_lo, int32_t diag_hi, lxo_hsp * out, uint8_t * ops);

int lxo_score_batch(uint8_t const * qres, uint8_t const * sres, uint64_t const * q_off, uint32_t const * q_len,
                    uint64_t const * s_off, uint32_t const * s_len, uint64_t n, lxo_scoring const * sc,
                    int32_t * score, int32_t * q_end, int32_t * s_end, int32_t threads);

/* inter-sequence int16 SIMD variant (lx_oracle_simd.cpp): the shape of the reference's CPU path */
int lxo_score_batch_simd(uint8_t const * qres, uint8_t const * sres, uint64_t const * q_off, uint32_t const * q_len,
                         uint64_t const * s_off, uint32_t const * s_len, uint64_t n, lxo_scoring const * sc,
                         int32_t * score, int32_t * q_end, int32_t * s_end, int32_t threads);

/* ---- window construction (src/search_misc.hpp:46-50, src/search_algo.hpp:919-938, :1136-1175) ---- */
int64_t  lxo_band_size(uint64_t len);
void     lxo_widen_match(lxo_match * m, uint64_t qlen, uint64_t slen);
/* qlens/slens are indexed by (frame-expanded) qryId / subjId. Returns new count. */
uint64_t lxo_widen_and_preprocess(lxo_match * m, uint64_t n, uint64_t const * qlens, uint64_t const * slens);

/* ---- pre-extension filter (src/search_algo.hpp:426-481) ---- */
int lxo_seed_looks_promising(uint8_t const * q, int64_t qlen, uint8_t const * s, int64_t slen, int64_t qry_start,
                             int64_t qry_end, int64_t subj_start, int64_t seed_length, int32_t pre_scoring,
                             double pre_scoring_thresh, lxo_scoring const * sc);

/* ---- BLAST statistics (src/search_misc.hpp:56-80; seqan blast_statistics, [UPSTREAM-RECALL]) ---- */
uint64_t lxo_length_adjustment(uint64_t db_len, uint64_t q_len, lxo_karlin const * ka);
double   lxo_evalue(int32_t score, uint64_t q_len_adj, uint64_t db_len_adj, lxo_karlin const * ka);
double   lxo_bitscore(int32_t score, lxo_karlin const * ka);

/* ---- per-HSP statistics (src/evaluate_bisulfite_alignment.hpp:26-117 for the shape) ---- */
int lxo_alignment_stats(uint8_t const * q, uint8_t const * s, lxo_hsp const * hsp, uint8_t const * ops,
                        lxo_scoring const * sc, int32_t bisulfite_match_rule, lxo_align_stats * out);

/* ---- frames and six-frame translation (src/search_algo.hpp:768-814, :940-996; src/shared_definitions.hpp:246-281;
 *      the translation itself is BioC++'s translate_join, [UPSTREAM-RECALL] for its treatment of N) ----
 * modes: 0 none, 1 reverse complement, 2 translated, 3 bisulfite */
int32_t  lxo_frame_of(int mode, uint64_t id, int is_subject);
uint64_t lxo_untrue_id(int mode, uint64_t n_id, int32_t frame, int is_subject);
/* dna5 = BioC++ ranks A,C,G,N,T; frame in {+1,+2,+3,-1,-2,-3}; out = SeqAn AminoAcid ranks; returns the length */
uint64_t lxo_translate_frame(uint8_t const * dna5, uint64_t n, int frame, uint8_t * out);

#ifdef __cplusplus
}
#endif
#endif

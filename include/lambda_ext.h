/*
 * lambda_ext.h -- C ABI of the MI355X seed-extension engine (liblambda_ext.so).
 *
 * This is the drop-in boundary for lambda3's per-hit gapped extension.  The reference has no
 * FFI; its seam is the C++ template call
 *
 *     _performAlignment(depSetH, depSetV, blastMatches, lH, bool_constant<withTrace>, bsDirection)
 *         /root/reference/src/search_algo.hpp:1070-1076, called at :1246 (score) and :1296 (trace)
 *
 * whose inputs are N (query slice, subject slice) pairs plus the scoring scheme of the strand
 * direction, and whose outputs are alignmentScore per pair (:1129) or the gapped rows (:1127).
 * The entry points below replace exactly that call (lx_score_batch / lx_align_batch) and, one level
 * up, the whole of iterateMatchesFullSimd (:1177-1332) (lx_iterate_matches).  INTEGRATION.md shows
 * the reference-side binding.
 *
 * Conventions
 *   - residues are 1 byte each, already in the rank encoding the reference feeds to SeqAn
 *     (src/seqan2_to_biocpp.hpp:382-395): SeqAn AminoAcid order for proteins, BioC++ rank for
 *     match/mismatch-scored nucleotides, SeqAn Dna5 order in bisulfite mode.
 *   - query = horizontal / outer sequence, subject = vertical / inner (src/search_algo.hpp:1058-1059)
 *   - gap_open is SeqAn's scoreGapOpen, i.e. lambda's gapOpen + gapExtend (src/search_algo.hpp:226-230)
 *   - every function returns LX_OK (0) or a negative LX_E* code; lx_last_error() gives the text.
 *     The C++ wrapper (lambda_amd/csrc/host/lambda_ext.hpp) rethrows std::runtime_error, which is what
 *     the reference does on failure (src/search.cpp:98-125).
 *   - caller owns every buffer; nothing allocated here crosses the boundary.
 *   - a handle is bound to one device and one HIP stream; use one handle per host thread, as the
 *     reference uses one LocalDataHolder per OpenMP thread (src/search.cpp:379-381).
 *   - there is NO CPU fallback: if no gfx950 device / kernel image is usable every call fails loudly.
 */
#ifndef LAMBDA_EXT_H
#define LAMBDA_EXT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LX_ABI_VERSION 3
#define LX_ALPH 32 /* matrix stride; alphabet_size <= 31, rank 31 is reserved as padding */

enum
{
    LX_OK        = 0,
    LX_EINVAL    = -1, /* bad argument                         */
    LX_ENODEV    = -2, /* no usable gfx950 device              */
    LX_ENOMEM    = -3, /* device or host allocation failed     */
    LX_EHIP      = -4, /* HIP runtime error                    */
    LX_EOVERFLOW = -5, /* workspace / output capacity exceeded */
    LX_ESTATE    = -6  /* call sequence error                  */
};

/* Scoring scheme of one strand direction: TScoreSchemeAlign, src/search_datastructures.hpp:360-365;
 * filled like prepareScoring(), src/search_algo.hpp:166-234. */
typedef struct lx_scoring
{
    int32_t alphabet_size;            /* number of valid ranks (27 aa, 5 nucleotide)              */
    int32_t gap_open;                 /* cost of the first gap character  (negative)               */
    int32_t gap_extend;               /* cost of each further gap character (negative)             */
    int32_t reserved;
    int8_t  matrix[LX_ALPH * LX_ALPH]; /* matrix[query_rank * 32 + subject_rank]                   */
} lx_scoring;

/* One (query slice, subject slice) pair = one element of depSetH/depSetV (src/search_algo.hpp:1056-1060). */
typedef struct lx_extension
{
    uint64_t q_off; /* byte offset of the query slice in the query residue buffer    */
    uint64_t s_off; /* byte offset of the subject slice in the subject residue buffer */
    uint32_t q_len;
    uint32_t s_len;
} lx_extension;

/* Result of the traceback pass: what _adaptTraceSegmentsTo + beginPosition/endPosition give
 * (src/search_algo.hpp:1127, :1032-1035) and what computeAlignmentStats derives (:1308). */
typedef struct lx_hsp
{
    int32_t score;
    int32_t q_begin, q_end; /* 0-based half-open, relative to the query slice   */
    int32_t s_begin, s_end; /* 0-based half-open, relative to the subject slice */
    int32_t n_ops;          /* alignment columns written to the ops buffer      */
    int32_t num_matches, num_mismatches, num_positives;
    int32_t num_gap_opens, num_gap_extensions;
    int32_t ops_shift;      /* the n_ops bytes start at ops_off[i] + ops_shift (the walk fills the slot back to front;
                               ops_shift = q_len + s_len - n_ops) */
} lx_hsp;

/* src/search_datastructures.hpp:46-61 */
typedef struct lx_match
{
    uint64_t qryId, subjId, qryStart, qryEnd, subjStart, subjEnd;
} lx_match;

typedef struct lx_handle lx_handle;

/* ---- lifetime ------------------------------------------------------------------------------- */
int          lx_abi_version(void);
/* Hash of the sources this library was compiled from (lambda_amd/build.py source_id()): lets a test or a deployment assert
 * that the binary matches the tree. */
char const * lx_build_id(void);
int          lx_device_count(void);
int          lx_create(int device_id, lx_handle ** out);
void         lx_destroy(lx_handle * h);
char const * lx_last_error(lx_handle const * h); /* h may be NULL: error of the failed lx_create */

/* Tuning knobs.  They never change results, only which kernel geometry is launched:
 *   LX_OPT_MAX_QLEN        longest query slice the *_dev calls will see (0 = unknown -> generic geometry)
 *   LX_OPT_QUERY_RUN       promise for the *_dev calls: extensions come in runs of this many consecutive entries
 *                          that share one query slice (a multiple of 8, or 4 for the multi-query sweep; 0 = no promise).  Lets a wavefront
 *                          build one LDS profile instead of one per extension.  2 = the multi-query sweep's free packing:
 *                          entries 2k and 2k + 1 share a query slice and every aligned block of 16 entries holds windows of
 *                          at most four query slices, in any split (what lx_extend_batch streams a ragged list into).  1 = the
 *                          sweep's SOLO packing: no promise at all, every window has its own byte profile, 16 windows per
 *                          wavefront -- for the schemes whose profiles are small enough (nucleotide, bisulfite; what
 *                          lx_extend_batch and the Level-2 driver plan read sets with).  A violated promise is detected on
 *                          the device and reported as LX_ESTATE by lx_synchronize().
 *   LX_OPT_WORKSPACE_BYTES carry workspace for queries wider than one panel in the *_dev calls (default 64 MiB; with
 *                          LX_OPT_MAX_SLEN set it grows by itself to 8 bytes per subject row of the batch)
 *   (LX_OPT_BS_MATCH_RULE is the one option that is not a tuning knob: it selects which of the reference's two
 *   computeAlignmentStats overloads the pass-2 match counts follow.) */
enum
{
    LX_OPT_MAX_QLEN        = 1,
    LX_OPT_QUERY_RUN       = 2,
    LX_OPT_WORKSPACE_BYTES = 3,
    LX_OPT_MAX_SLEN        = 4, /* longest subject slice the *_dev calls will see (0 = unknown: measured on the
                                   device, which costs lx_align_batch_dev one stream synchronisation)          */
    LX_OPT_TRACE_BYTES     = 5, /* HBM budget for direction bits in pass 2 (default 64 GiB); larger batches are
                                   processed in chunks, in order, on the same stream                            */
    LX_OPT_BS_MATCH_RULE   = 6, /* 1: lx_hsp match counts use the bisulfite rule score(c0,c1)==score(c0,c0)
                                   (src/evaluate_bisulfite_alignment.hpp:97) instead of rank equality        */
    LX_OPT_PACKED_HALF     = 7, /* 1 (default): the packed kernels (two extensions per lane group: half precision, or
                                   16-bit integers for wider queries) run where a per-wavefront score bound proves them
                                   exact (results are bit-identical either way); 0: int32 kernels only */
    LX_OPT_PASS2_MODE      = 8, /* how pass 2 keeps what the traceback needs (results are bit-identical in every mode):
                                   0 = 4 direction bits per cell of every survivor;
                                   1 = strip boundaries + row checkpoints of every survivor, tiles recomputed by the
                                       backtrace -- where its limits hold (extensions in blocks of >= 4 per query,
                                       scores below 32000; queries of any width, panel by panel), else mode 0;
                                   2 (default) = single sweep in lx_extend_batch_dev: one checkpointing kernel over ALL
                                       extensions is pass 1 and the forward half of pass 2 at once -- where mode 1
                                       applies, LX_OPT_QUERY_RUN is a multiple of 8 and the checkpoints of the whole
                                       batch (7.7 KB per 150 x 176 extension as compact codes of the packed-half kernel,
                                       13.5 KB as int16 pairs) fit LX_OPT_TRACE_BYTES, else mode 1 */
    LX_OPT_EXTEND_CHUNK    = 10, /* extensions per chunk of lx_extend_batch's pipeline (0 = default, ~640 k; at least 1024):
                                   smaller chunks start returning results earlier, larger ones amortise the per-chunk launches */
    LX_OPT_MQ_SWEEP        = 11, /* multi-query single sweep (ragged seed lists: up to four queries per wavefront, byte profiles
                                   in LDS): 1 (default) = where LX_OPT_QUERY_RUN is 2, 4 or 8 -- 2 is what lx_extend_batch
                                   makes of a list whose queries have few windows each; 2 = for every run that is a multiple
                                   of 4; 0 = never (runs of 8 / 16 on the one-query-per-wavefront kernels).  Needs what the
                                   single sweep needs and a scheme in which no substitution costs more than a gap's first
                                   character (every scheme of the reference with its default gap costs) */
    LX_OPT_ADAPT_PERMILLE  = 12, /* adaptive pass 2 (with LX_OPT_PASS2_MODE = 2): when fewer than this many per mille of the previous
                                   batch's extensions passed the cut-off, the step runs plain pass 1 and writes checkpoints for
                                   the survivors only (mode 1) instead of checkpoints for every window -- the reference's own
                                   order, src/search_algo.hpp:1246 / :1251-1283 / :1296; default 30, 0 = always the single sweep.
                                   Results are identical either way */
    LX_OPT_ITERATE_RECORDS = 13, /* where lx_iterate_matches_dev (and lx_iterate_matches on lists it hands to the device) makes the result
                                   records behind the two passes (src/search_algo.hpp:1287-1325: statistics, order, identity cut-off, bit
                                   score, e-value): 0 (default) = kernels on the survivors where the backtrace left them, finished rows come
                                   down in one copy; 1 = the host threads, from the survivors' alignments (rounds 1-4).  The records are the
                                   same to the bit either way (tests/test_gpu_level2.py) */
    LX_OPT_HOST_THREADS    = 14, /* parts the host loops of the host-buffer entry points and of the Level-2 driver are cut into (= the host
                                   threads that can work on one loop).  PROCESS-WIDE, whichever handle it is set through: the library keeps
                                   ONE set of host threads, and the loops of all handles -- one handle per host thread is the intended use,
                                   as the reference keeps one LocalDataHolder per OpenMP thread, src/search.cpp:379-385 -- share them part
                                   by part, side by side.  0 (default) = min(CPU affinity mask, cgroup CPU quota) / LOCAL_WORLD_SIZE (the
                                   ranks of a node, one process per GPU, share its CPUs), at most 16; at most 64.  Never changes results */
    LX_OPT_BAND            = 9  /* band mode -- NOT the reference's configuration (src/search_algo.hpp:1081 runs BandOff, :1102
                                   says why; _bandSize only pads the window, src/search_misc.hpp:46-50) and therefore not a
                                   parity mode: 0 (default) = full rectangle; b > 0 = only cells whose diagonal i - j (row i of
                                   the subject slice, column j of the query slice, 0-based) lies within b of the extension's
                                   centre diagonal exist, as in a banded SeqAn alignment: cells off the band are never
                                   computed, no gap runs through them.  Centres: lx_set_band_centres[_dev]; without them
                                   min(_bandSize(q_len), s_len - q_len), the seed diagonal of a window built by _widenMatch
                                   that was not clipped at the subject's start.  Applies to every score / align / extend entry
                                   point; runs int32 kernels and direction bits (results are those of the oracle's
                                   lxo_score_banded / lxo_align_banded).  PERFORMANCE: band mode exists for its semantics only --
                                   it is SLOWER per window than the full rectangle (headline batch: 51 ms per step with b = 64
                                   against 15.7 ms without a band): at 150 x 176 with b = 64 the band removes 27 % of the cells and
                                   no step of the strip mapping, and every remaining cell pays for its mask.  A caller who wants
                                   throughput leaves the band off, which is also what the reference computes */
};
int lx_set_option(lx_handle * h, int option, uint64_t value);
/* The host threads as the library sized them (no handle, no device needed): parts per loop (LX_OPT_HOST_THREADS or the default),
 * the CPUs granted to the process (affinity mask and cgroup quota), LOCAL_WORLD_SIZE as read from the environment.  Any pointer
 * may be NULL. */
int lx_host_threads_info(uint32_t * width, uint32_t * granted_cpus, uint32_t * local_world_size);

/* Introspection, no device needed: how lx_extend_batch_dev would run a batch of n extensions of queries up to max_qlen and
 * windows up to max_slen under the given options (the LX_OPT_* of the same names; survivor_share < 0 = unknown).  The
 * library decides this in ONE function; this entry point shows its answer, and tests/test_plan.py walks it over query
 * widths, run lengths and schemes.  family: 0 = no sweep (pass 1, then pass 2 on the survivors), 1 = packed-half sweep,
 * 2 = packed-int16 sweep with compact codes over several panels, 3 = packed-int16 sweep with int16-pair slots, 4 = int32 sweep,
 * 5 = multi-query sweep (byte profiles).  name = what lx_last_trace_kernel_name() reports after such a step. */
typedef struct lx_step_plan
{
    int32_t  family, group_lanes, strip_cols, panels;
    int32_t  compact_codes, queries_per_wavefront, may_decline, adapted;
    uint64_t slot_bytes;  /* checkpoint bytes per extension */
    uint64_t lds_bytes;   /* LDS of one wavefront of the sweep kernel */
    uint64_t score_bound; /* a-priori bound of every intermediate for the widest admitted query */
    char     name[160];
} lx_step_plan;
int lx_plan_step(lx_scoring const * sc, uint64_t max_qlen, uint64_t max_slen, uint64_t query_run, uint64_t n, uint64_t pass2_mode,
                 uint64_t mq_sweep, uint64_t packed_half, uint64_t trace_bytes, double survivor_share, uint64_t adapt_permille,
                 lx_step_plan * out);
/* current value of an option (what the caller set or the default; never the library's internal growth of a workspace) */
int lx_get_option(lx_handle const * h, int option, uint64_t * value);

/* Band mode (LX_OPT_BAND > 0): centre diagonal of every extension of the NEXT host-buffer call, diag[i] for ext[i] (n must
 * equal that call's n; diag = NULL, n = 0 returns to the default centres).  The *_dev calls read a device array of int32,
 * one per extension, that must stay valid while they run (NULL = default centres). */
int lx_set_band_centres(lx_handle * h, int32_t const * diag, uint64_t n);
int lx_set_band_centres_dev(lx_handle * h, void const * d_diag);

/* slot 0 = forward scheme, slot 1 = bisulfite reverse scheme (scoringSchemeAlignBSRev,
 * src/search_algo.hpp:1097-1098).  Must be called before any batch call using that slot. */
int lx_set_scoring(lx_handle * h, int slot, lx_scoring const * sc);

/* Fills *sc with a built-in scheme the way prepareScoring() does: scoring_method 45/62/80 = BLOSUM
 * (protein), 0 = match/mismatch (nucleotide), -1 = bisulfite forward, -2 = bisulfite reverse.
 * gap_open_lambda / gap_extend are lambda's options (e.g. -11/-1), NOT SeqAn's. */
int lx_builtin_scoring(int scoring_method, int match, int mismatch, int gap_open_lambda, int gap_extend,
                       lx_scoring * sc);

/* ---- resident subjects -------------------------------------------------------------------------------------------
 * The subject sequences of a search do not change between query batches (the reference holds them in its index file,
 * src/shared_definitions.hpp:343-379).  lx_set_subjects uploads the (frame-expanded, rank-encoded) subject residues
 * once; afterwards every host-buffer call (lx_score_batch, lx_align_batch, lx_prefilter_batch, lx_iterate_matches)
 * may pass s_res = NULL, s_bytes = 0 to use the resident copy instead of uploading its own.  s_bytes = 0 drops it. */
int lx_set_subjects(lx_handle * h, uint8_t const * s_res, uint64_t s_bytes);

/* ---- pass 1: score only  (replaces _performAlignment<false>, src/search_algo.hpp:1246) ------- */
/* Host buffers in, host buffers out; copies through pinned staging, runs on the handle's stream, returns
 * after the results have landed. out_score[i] is what the reference stores at :1129. */
int lx_score_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                   uint64_t s_bytes, lx_extension const * ext, uint64_t n, int32_t * out_score);

/* Device-resident variant: every pointer is a device pointer on the handle's device; asynchronous on
 * `stream` (a hipStream_t; NULL = the handle's own stream).  This is what bench.py times.
 * The residue buffers must start 16-byte aligned (hipMalloc's do) and keep 256 readable bytes after the last
 * residue: the kernels fetch residues in aligned groups. */
int lx_score_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                       uint64_t n, void * d_out_score, void * stream);

/* ---- pass 2: traceback  (replaces _performAlignment<true>, src/search_algo.hpp:1296) --------- */
/* LIMIT: pass 2 (lx_align_batch*, and the second half of every lx_extend_batch* / lx_iterate_matches call) needs every
 * (matrix entry - gap_extend) of the slot's scheme in [-31, 31] -- the tagged recurrence keeps two tag bits next to values
 * scaled by 4 in 8-bit profile entries, there is no int32 fall-back for it.  Every scheme of the reference with its default
 * costs satisfies this (BLOSUM45/62/80 with gap_extend -1 ... -2, +2/-3 nucleotide and bisulfite matrices); a user-chosen
 * --match / --mismatch beyond that range (src/search_options.hpp:93-94) makes these calls return LX_EINVAL with that text.
 * Pass 1 (lx_score_batch*) has no such limit.
 * ops: one byte per alignment column ('M','D','I'; 'D' = gap in the query row), begin -> end order.  The caller gives
 * extension i a slot of q_len+s_len bytes at out_ops + ops_off[i]; its out_hsp[i].n_ops bytes are written at the END
 * of that slot, i.e. they start at out_ops + ops_off[i] + out_hsp[i].ops_shift. */
int lx_align_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                   uint64_t s_bytes, lx_extension const * ext, uint64_t n, int32_t const * known_score, lx_hsp * out_hsp,
                   uint8_t * out_ops, uint64_t const * ops_off);
/* known_score: NULL, or the pass-1 score of every extension (the reference runs pass 2 on survivors of pass 1, so the
 * caller has them): pass 2 locates the end cell through the best score and computes it itself when not given.  A wrong
 * score is detected (LX_EOVERFLOW), never turned into a wrong alignment.  Extensions of one query should be adjacent
 * (lambda's lists are): runs of the same query slice then share one LDS profile. */

int lx_align_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                       uint64_t n, void * d_out_hsp, void * d_out_ops, void const * d_ops_off, void * stream);

/* ---- both passes fused on the device (the middle of iterateMatchesFullSimd, src/search_algo.hpp:1246-1296) --- */
/* pass 1 on all n extensions -> keep extension i iff score >= (d_min_score ? d_min_score[i] : min_score_all) [the
 * bit-score / e-value tests of :1251-1283 expressed as integer score cut-offs, which the host derives from
 * lx_evalue()/lx_bitscore(); both are monotone in the score] -> pass 2 on the survivors.  Nothing leaves the GPU and
 * nothing synchronises: d_out_score[n] (int32), d_out_hsp[n] (lx_hsp; filtered-out rows carry the score and n_ops = 0),
 * ops of extension i at d_out_ops + d_ops_off[i], d_out_count[0] = pass-2 slots used, [1] = survivors (uint64).
 * Requires LX_OPT_MAX_QLEN and LX_OPT_MAX_SLEN; honours LX_OPT_QUERY_RUN. */
int lx_extend_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                        uint64_t n, void const * d_min_score, int32_t min_score_all, void * d_out_score,
                        void * d_out_hsp, void * d_out_ops, void const * d_ops_off, void * d_out_count, void * stream);

/* The same on host buffers (what lx_iterate_matches uses).  Extension order is free; extensions of one query slice are
 * best adjacent (lambda's lists are).  out_score[n] as in pass 1; out_hsp[n]: filtered-out rows carry the score and
 * n_ops = 0; the ops of a survivor i are *out_ops + out_ops_off[i] + out_hsp[i].ops_shift (the callee lays them out
 * compactly; the buffer belongs to the handle and stays valid until its next lx_extend_batch[_rle] call).  The list is
 * processed as a pipeline of chunks (uploads, kernels, downloads and the host's share overlap); only scores, the
 * survivors' records and their run-length coded ops cross PCIe.  A list that is not uniform -- what
 * _widenAndPreprocessMatches really hands over: queries of mixed lengths, a few windows each, merged windows of up to three
 * times the length (:1136-1175) -- is planned for the multi-query sweep (LX_OPT_MQ_SWEEP): a wavefront's 16 slots hold windows
 * of up to four queries, the long (merged) windows of the list pooled in sub-blocks of 4 and dealt by length, the others
 * streamed pair by pair in order of length; the records are gathered from a device copy of the list and the scores scattered
 * back into the caller's order on the device.  lx_last_extend_stats() reports extensions,
 * slots, cells and the cells the wavefronts executed (padding included) of the last call.
 * Throughput depends on the batch size -- a call has a fixed cost of ~0.7 ms (INTEGRATION.md has the curve): hand over the
 * windows of at least ~1 000 queries per call. */
int lx_extend_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                    lx_extension const * ext, uint64_t n, int32_t const * min_score, int32_t min_score_all, int32_t * out_score,
                    lx_hsp * out_hsp, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes);

/* The same with the ops in the form they cross PCIe in: run-length codes instead of one byte per column -- one byte per run,
 * (op << 6) | (length - 1) with op 0 = 'M', 1 = 'D', 2 = 'I', runs longer than 64 columns split (the remainder first, then
 * pieces of 64: one alignment has one code string), begin -> end order.  The codes
 * of a survivor i start at *out_ops + out_ops_off[i] and end where their lengths add up to out_hsp[i].n_ops (the number of
 * alignment columns, as ever); out_hsp[i].ops_shift is 0.  This is what a binding that fills SeqAn's gapped rows wants (ArrayGaps
 * stores run lengths) and what lx_iterate_matches uses; lx_expand_ops turns one survivor's codes into column bytes. */
int lx_extend_batch_rle(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                        lx_extension const * ext, uint64_t n, int32_t const * min_score, int32_t min_score_all, int32_t * out_score,
                        lx_hsp * out_hsp, uint64_t * out_ops_off, uint8_t const ** out_ops, uint64_t * out_ops_bytes);
int lx_expand_ops(uint8_t const * codes, int32_t n_ops, uint8_t * out /* n_ops bytes */);

/* The same with the result in the form the reference's filter loop leaves behind: it ERASES the matches that fail the
 * e-value / bit-score test from its list (src/search_algo.hpp:1251-1283) and goes on with the survivors only (:1287-1385).
 * out_score[n] as ever (the filter's statistics need every score); the survivors come back as a list -- position in the
 * caller's list, record, run-length codes as lx_extend_batch_rle -- in buffers that belong to the handle and stay valid
 * until its next lx_extend_batch* call.  The order of the list is the order the device finished the chunks in (ascending
 * positions for a list that is grouped by query and uniform; otherwise grouped by chunk): `index` says which row is whose.
 * This is the fastest host entry point: nothing of size n except the scores is written on the host (lx_extend_batch fills
 * n records of 48 bytes, which at millions of extensions per call costs as much as the kernels). */
typedef struct lx_survivor_list
{
    uint64_t         count;
    uint32_t const * index;       /* [count] position in the caller's list                                            */
    lx_hsp const *   hsp;         /* [count] ops_shift = 0                                                            */
    uint64_t const * codes_off;   /* [count] the codes of survivor k start at codes + codes_off[k] and end where their
                                     lengths add up to hsp[k].n_ops                                                   */
    uint8_t const *  codes;
    uint64_t         codes_bytes;
} lx_survivor_list;
int lx_extend_batch_list(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res, uint64_t s_bytes,
                         lx_extension const * ext, uint64_t n, int32_t const * min_score, int32_t min_score_all, int32_t * out_score,
                         lx_survivor_list * out);
/* Padding of the last lx_extend_batch* call: out4 = {extensions with residues, slots (windows + fillers), cells (sum q_len *
 * s_len), cells the wavefronts execute (per wavefront: 16 slots x the columns its widest query sweeps -- whole panels, the last
 * one with narrower strips where that covers the query -- x the steps of its longest window)}. */
int lx_last_extend_stats(lx_handle const * h, uint64_t * out4);

/* ---- pre-extension filter (seedLooksPromising, src/search_algo.hpp:426-481) ------------------ */
/* One diagonal per item; out_keep[i] = 1 if the ungapped max-segment score reaches the threshold. */
typedef struct lx_seed
{
    uint64_t q_off, s_off; /* start of the whole (frame) query / subject sequence */
    uint32_t q_len, s_len; /* their full lengths                                 */
    uint32_t qry_start, qry_end, subj_start, reserved;
} lx_seed;
int lx_prefilter_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                       uint64_t s_bytes, lx_seed const * seeds, uint64_t n, uint32_t seed_length,
                       int32_t pre_scoring, double pre_scoring_thresh, uint8_t * out_keep);


/* ---- host mirror of the extension driver (one level above the two passes) ---------------------------- */
/* Karlin-Altschul parameters of a scoring scheme (seqan::BlastScoringScheme; call sites src/search_misc.hpp:73-78,
 * src/search_algo.hpp:1258).  Tables are NCBI's published values; returns LX_EINVAL for combinations that have
 * none, which is what makes prepareScoring() throw (src/search_algo.hpp:232-233). */
typedef struct lx_karlin
{
    double lambda, K, H, alpha, beta;
} lx_karlin;
int      lx_karlin_params(int scoring_method, int match, int mismatch, int gap_open_lambda, int gap_extend, lx_karlin * out);
uint64_t lx_length_adjustment(uint64_t db_len, uint64_t q_len, lx_karlin const * ka);
double   lx_evalue(int32_t score, uint64_t q_len_adj, uint64_t db_len_adj, lx_karlin const * ka);
double   lx_bitscore(int32_t score, lx_karlin const * ka);

/* Rank bridge (seqan2_to_rank_inner / biocpp_rank_to_seqan2rank, src/seqan2_to_biocpp.hpp:352-366, :382-395): turns
 * n BioC++ ranks into the ranks the scoring tables of this library are indexed by.  in == out is allowed.
 *   LX_RANKS_AA27      aa27 (A..W, X, Y, Z, '*')  ->  SeqAn AminoAcid order (.. W, Y, Z, X, '*')
 *   LX_RANKS_DNA5_BS   dna5 (A, C, G, N, T)       ->  SeqAn Dna5 order (A, C, G, T, N); bisulfite mode only
 *   LX_RANKS_SIMPLE    alphabets scored by match/mismatch keep their BioC++ rank (plain copy)
 * Returns LX_EINVAL for an unknown kind or a rank outside the alphabet. */
enum
{
    LX_RANKS_AA27    = 0,
    LX_RANKS_DNA5_BS = 1,
    LX_RANKS_SIMPLE  = 2
};
int lx_convert_ranks(int kind, uint8_t const * in, uint64_t n, uint8_t * out);

/* _widenAndPreprocessMatches (src/search_algo.hpp:1136-1175), in place; qlens/slens are indexed by the
 * frame-expanded qryId/subjId.  Returns the new match count. */
uint64_t lx_widen_and_preprocess(lx_match * m, uint64_t n, uint64_t const * qlens, uint64_t const * slens);

/* Options of iterateMatchesFullSimd that live in LambdaOptions / the BLAST context. */
typedef struct lx_search_params
{
    double    max_evalue;       /* < 0: no e-value filter   (src/search_options.hpp:96-97)          */
    int32_t   min_bitscore;     /* < 0: no bit-score filter                                           */
    int32_t   id_cutoff;        /* percent identity cut-off (src/search_algo.hpp:1310-1315)           */
    uint64_t  db_total_length;  /* context.dbTotalLength    (src/search_algo.hpp:317-319)             */
    int32_t   query_translated; /* 1: ql /= 3 before the length adjustment (src/search_misc.hpp:70)   */
    int32_t   qry_num_frames;   /* qryId / qry_num_frames = true query id (src/search_algo.hpp:1210)  */
    int32_t   sbj_num_frames;
    int32_t   bisulfite;        /* 1: iterateMatches' bisulfite branch (src/search_algo.hpp:1367-1379): matches on even subject
                                   frames are extended with slot 0 (forward scheme), odd ones with slot 1 (reverse scheme);
                                   the `slot` argument is ignored; match counts / identity follow the bisulfite overload of
                                   computeAlignmentStats (score(c0,c1)==score(c0,c0), src/evaluate_bisulfite_alignment.hpp:97)
                                   for this call whatever LX_OPT_BS_MATCH_RULE says                                         */
    int32_t   q_frame_mode;     /* LX_FRAMES_*: how _setFrames derives qFrameShift from the frame-expanded qryId            */
    int32_t   s_frame_mode;     /* same for sFrameShift / subjId                                                            */
    lx_karlin karlin;
    int32_t   band;             /* 0: full rectangle, what the reference computes; b > 0: band mode (LX_OPT_BAND) with the default
                                   centres for the duration of the call -- not a parity mode                               */
    int32_t   flags;            /* LX_ITERATE_*; 0 = the reference's behaviour                                                     */
} lx_search_params;
/* lx_search_params.flags: the result carries no alignment columns -- lx_iterate_result_ops() is NULL, every record's n_ops is still the
 * alignment's length and ops_off is 0.  For callers whose output needs none (BLAST-tabular: every column comes from the counts; the
 * SAM writer needs them for its CIGAR): a million HSPs are 110 MB of column bytes that nobody reads. */
#define LX_ITERATE_NO_OPS 1

/* ---- frame bookkeeping (_setFrames, _untrueQryId, _untrueSubjId; src/search_algo.hpp:768-814, :940-996) ------------
 * The reference expands every sequence into its frames (translate_join: 6, add_reverse_complement: 2, bisulfite
 * query: 4 / subject: 2; src/shared_definitions.hpp:246-281) and numbers them id * numFrames + k.  The mode says
 * which expansion a side uses:
 *   LX_FRAMES_NONE        frame 0                                        (protein sequences)
 *   LX_FRAMES_REVCOMP     k = 0 -> +1, k = 1 -> -1                       (blastn query)
 *   LX_FRAMES_TRANSLATED  k = 0..2 -> +1..+3, k = 3..5 -> -1..-3         (blastx / tblastx query, tblastn / tblastx subject)
 *   LX_FRAMES_BISULFITE   query: k = 0..3 -> +1, +2, -1, -2;  subject: k = 0, 1 -> +1, +2 */
enum
{
    LX_FRAMES_NONE       = 0,
    LX_FRAMES_REVCOMP    = 1,
    LX_FRAMES_TRANSLATED = 2,
    LX_FRAMES_BISULFITE  = 3
};
void     lx_set_frames(int q_mode, int s_mode, uint64_t qry_id, uint64_t subj_id, int32_t * q_frame, int32_t * s_frame);
/* the frame-expanded id whose sequence an HSP with this frame was aligned on (the bisulfite duplicates are identical
 * sequences, so the reference maps them onto the first copy) */
uint64_t lx_untrue_qry_id(int q_mode, uint64_t n_qid, int32_t q_frame);
uint64_t lx_untrue_subj_id(int s_mode, uint64_t n_sid, int32_t s_frame);

/* Six-frame translation (what bio::views::translate_join hands to the extension in the translated programs,
 * src/shared_definitions.hpp:246-281): `dna5` holds n BioC++ dna5 ranks (A, C, G, N, T).  Writes the frames
 * +1, +2, +3, -1, -2, -3 back to back into `out` (SeqAn AminoAcid ranks; 2n bytes are always enough) and their
 * offsets/lengths into frame_off/frame_len.  A codon with N becomes the amino acid all its completions agree on,
 * else X.  genetic_code: an NCBI translation table id as bio::alphabet::genetic_code numbers them (1 = canonical, the
 * reference's default, src/search_options.hpp:170; 2-6, 9-16, 21-25).  Returns LX_EINVAL for other ids, bad ranks or a
 * too small `out`. */
int lx_translate_six_frames(uint8_t const * dna5, uint64_t n, int genetic_code, uint8_t * out, uint64_t out_capacity,
                            uint64_t * frame_off, uint64_t * frame_len);

/* One finished HSP: the fields of TBlastMatch the writers read (src/search_datastructures.hpp:470-484). */
typedef struct lx_blast_match
{
    uint64_t qry_id, subj_id;   /* frame-expanded ids of the lx_match this HSP came from */
    uint64_t n_qid, n_sid;      /* true ids (_n_qId/_n_sId)                              */
    uint64_t q_start, q_end;    /* in the (frame) query sequence, 0-based half-open      */
    uint64_t s_start, s_end;    /* in the (frame) subject sequence                       */
    int32_t  score;
    int32_t  alignment_length;
    int32_t  num_matches, num_mismatches, num_positives, num_gap_opens, num_gap_extensions;
    float    identity;
    double   bit_score, e_value;
    uint64_t ops_off;           /* into the result's ops buffer */
    uint32_t n_ops;
    int16_t  q_frame;           /* qFrameShift (lx_set_frames): 0 = none, +-1 strands, +-1..3 translation frames */
    int16_t  s_frame;           /* sFrameShift                                                                    */
} lx_blast_match;

typedef struct lx_iterate_stats
{
    uint64_t hits_duplicate, failed_bitscore, failed_evalue, failed_identity, num_ext_score, num_ext_ali;
} lx_iterate_stats;

typedef struct lx_iterate_result lx_iterate_result;

/* iterateMatchesFullSimd (src/search_algo.hpp:1177-1332) for one strand direction: widen/merge the seed hits,
 * score every window on the GPU, filter by bit score / e-value, trace the survivors on the GPU, expand to
 * sequence coordinates, apply the identity cut-off.  q_seq_off/q_seq_len (s_*) give, per frame-expanded sequence
  * id, where that sequence lives in q_res (s_res).  q_orig_len[n_qid] is the untranslated query length used for
 * the e-value (bm.qLength, src/search_algo.hpp:1213).  `matches` is modified in place (like the reference's span). */
/* Lists of 131 072 matches and more (at most 2^31 - 16) over resident subjects are handed to the Level-2 kernels (lx_iterate_matches_dev's
 * path): same records and statistics; two differences a caller can see -- a seed on a diagonal beyond its subject's end is LX_EINVAL there
 * (the host form widens it to an empty window), and after a bisulfite call `matches[0 .. windows)` holds the window list contiguously (even
 * subject frames first), where the host form leaves its two halves where the sort put them. */
int lx_iterate_matches(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint64_t const * q_seq_off,
                       uint64_t const * q_seq_len, uint64_t n_qseq, uint64_t const * q_orig_len,
                       uint8_t const * s_res, uint64_t s_bytes, uint64_t const * s_seq_off, uint64_t const * s_seq_len,
                       uint64_t n_sseq, lx_match * matches, uint64_t n_matches, lx_search_params const * params,
                       lx_iterate_result ** out);

/* The same for matches that stand in DEVICE memory -- where a seeding stage on the GPU leaves them -- over sequence sets that are
 * resident on the handle's device: _widenMatch (src/search_algo.hpp:919-938), the sort, both merge passes and unique of
 * _widenAndPreprocessMatches (:1136-1175), the slices (:1200-1227) and the filter's cut-offs (:1251-1283) run as kernels; only the
 * finished window list (24 bytes per window) comes down for the plan of the sweep, the extension reads list and cut-offs where the
 * kernels wrote them.  Results are lx_iterate_matches' (same order, same records, same statistics).
 *   lx_set_queries       the (frame-expanded) query set: residues, where each sequence lies, q_orig_len[n_qseq / qry_num_frames]
 *                        (NULL: the sequences' own lengths) -- what lH.transQrySeqs / qrySeqs are to the reference's thread;
 *                        replaces the previous set; nothing of the caller's is referenced after the call returns
 *   lx_set_subjects      the subject residues (above), lx_set_subject_seqs where each (frame) subject lies in them
 *   d_matches            n_matches lx_match records in device memory of the handle's device, finished before the call (the call
 *                        runs on the handle's stream); not modified
 * params->qry_num_frames must be the value the queries were set with; params->band is not served here (lx_iterate_matches).
 * LX_EINVAL: a match names a sequence outside the sets or lies beyond its subject's end. */
int lx_set_queries(lx_handle * h, uint8_t const * q_res, uint64_t q_bytes, uint64_t const * q_seq_off, uint64_t const * q_seq_len, uint64_t n_qseq,
                   uint64_t const * q_orig_len, int32_t qry_num_frames);
int lx_set_subject_seqs(lx_handle * h, uint64_t const * s_seq_off, uint64_t const * s_seq_len, uint64_t n_sseq);
int lx_iterate_matches_dev(lx_handle * h, int slot, void const * d_matches, uint64_t n_matches, lx_search_params const * params,
                           lx_iterate_result ** out);
/* Introspection: the plan lx_iterate_matches_dev makes for a PROTEIN window list -- the multi-query sweep's free packing (LX_OPT_QUERY_RUN
 * = 2: the two windows of a lane group share a query slice, a wavefront's 16 slots hold windows of at most four slices) -- for a list
 * of n lx_extension records in device memory that is grouped by query slice, as kernels (what the reference does with its sort by slice
 * lengths, src/search_algo.hpp:1229-1235, so that a SIMD batch's windows take about as many steps), brought to the host.
 * strip_cols: 19, 13 or 11 (columns per lane of a strip).  Ranges: cut[0] = 0 < ... < cut[nranges] = n, every cut where the query
 * changes; the plan's wavefronts are laid out range by range.  out_plan: [bound * 16] slots -- position in the list, bit 31 = filler (a
 * copy that never survives), 0xffffffff beyond the plan's wavefronts --, out_pan / out_maxs: [bound] columns per lane of a wavefront's
 * widest query / its longest window, bound = lx_plan_free_packing_bound(n, n_qseq, nranges); out_report: [16] = {wavefronts, error
 * flag (0), runs, quads of the pool, first wavefront of range 0, 1, ..., nranges}.  tests/test_gpu_plan.py checks every slot. */
uint64_t lx_plan_free_packing_bound(uint64_t n, uint64_t n_qseq, uint32_t nranges);
int      lx_plan_free_packing_dev(lx_handle * h, void const * d_ext, uint64_t n, uint64_t n_qseq, int32_t strip_cols, uint32_t nranges, uint64_t const * cut,
                                  uint32_t * out_plan, uint32_t * out_pan, uint32_t * out_maxs, uint32_t * out_report);
/* The library's own radix sort (least significant digit first, 8 bits per pass, only the digits set in key_bits; stable) for n (key, value)
 * word pairs in device memory: key[0] / value[0] hold the input, key[1] / value[1] are scratch of the same size; *sorted_in says which of
 * the two holds the sorted words afterwards.  Synchronises `stream`; the caller's current device is left as it was.  n < 2^31.  (The stand-alone front end sorts its word table with it;
 * the Level-2 driver sorts matches and survivors with the same kernels.) */
int lx_sort_words_dev(int device, uint64_t * key[2], uint64_t * value[2], uint64_t n, uint64_t key_bits, void * stream, int * sorted_in);
/* Ahead of a handle's first lx_iterate_matches_dev (or large lx_iterate_matches) call: allocates -- and, on the host side, touches -- the
 * buffers a call on up to n_matches matches needs when they become up to n_windows windows and n_hsps result records with n_columns
 * alignment columns in all (0: results without columns, LX_ITERATE_NO_OPS).  Call it after lx_set_queries (the lanes are sized for the
 * longest query).  A first call costs what the tenth costs then (bench.py --iterate --cold); nothing changes but when the memory is
 * asked for, and estimates that fall short only mean that the call grows what is missing, as it does without lx_reserve.  The reference
 * has no counterpart: its threads' LocalDataHolder grows inside the search loop (src/search_datastructures.hpp:386-468). */
int lx_reserve(lx_handle * h, uint64_t n_matches, uint64_t n_windows, uint64_t n_hsps, uint64_t n_columns);
/* _widenAndPreprocessMatches (:1136-1175) alone on a device match list over the resident sets: the window list comes back as
 * lx_match records in `out` (room for n_matches), *out_n of them -- what lx_widen_and_preprocess leaves in its span.  bisulfite != 0:
 * ordered by subjId % 2 first, as iterateMatches' bisulfite branch sorts (:1369-1372). */
int lx_widen_and_preprocess_dev(lx_handle * h, void const * d_matches, uint64_t n_matches, int32_t bisulfite, lx_match * out, uint64_t * out_n);
uint64_t               lx_iterate_result_count(lx_iterate_result const * r);
lx_blast_match const * lx_iterate_result_matches(lx_iterate_result const * r);
uint8_t const *        lx_iterate_result_ops(lx_iterate_result const * r);
lx_iterate_stats       lx_iterate_result_stats(lx_iterate_result const * r);
void                   lx_iterate_result_free(lx_iterate_result * r);
/* A freed result's large arrays are kept for the next result (a block the allocator has just mapped costs a page fault per 4 KB when the
 * threads write it): a handful of blocks, at most 1 GiB per process.  This call gives them back; returns the bytes released. */
uint64_t               lx_trim_result_cache(void);


/* ---- record post-processing and writers (row N2: _writeRecord + BLAST-tabular / SAM output) --------------- */
/* _writeRecord (src/search_algo.hpp:820-913) for a whole result list: `m` must be grouped by n_qid (the order
 * lx_iterate_matches returns).  Per query: sort by (n_sid, q_start, q_end, s_start, s_end, frames, bit score
 * descending), drop duplicates of the same coordinates keeping the best, stable-sort by bit score descending, keep at
 * most max_matches.  In place; returns the new count. */
typedef struct lx_record_stats
{
    uint64_t qrys_with_hit, hits_duplicate2, hits_abundant, hits_final, pairs;
} lx_record_stats;
uint64_t lx_postprocess_records(lx_blast_match * m, uint64_t n, uint64_t max_matches, lx_record_stats * stats);

/* The taxonomy the reference keeps in its index (indexFile.taxonParentIDs / taxonHeights / sTaxIds): parents[t] and heights[t]
 * for taxon t < n_taxa (parent 0 = unassigned or the root), and per true subject id the taxa it is assigned to in CSR form:
 * s_tax_ids[s_tax_off[s] .. s_tax_off[s + 1]). */
typedef struct lx_tax_tree
{
    uint32_t const * parents;
    uint32_t const * heights;
    uint64_t         n_taxa;
    uint64_t const * s_tax_off; /* n_s + 1 entries */
    uint32_t const * s_tax_ids;
    uint64_t         n_s;
} lx_tax_tree;
/* The LCA step of _writeRecord (src/search_algo.hpp:884-907, computeLCA: src/search_misc.hpp:86-112) over a result list
 * grouped by n_qid (what lx_postprocess_records returns): per query the lowest common ancestor of the taxa of its subjects --
 * starting from the first match whose subject has a first taxon with a parent, folding in every assigned taxon of every match.
 * Writes one (n_qid, lcaTaxId) pair per query in list order (lcaTaxId 0 = no assigned subject) and their number; out arrays
 * need one entry per query with hits (at most n).  LX_EINVAL for ids outside the tree or a path that does not lead to the root
 * (the reference throws "LCA-computation error"). */
int lx_compute_lca(lx_blast_match const * m, uint64_t n, lx_tax_tree const * tree, uint64_t * out_qid, uint32_t * out_lca,
                   uint64_t * out_n);

/* What the writers need to know about the sequences (the reference reads these from lH.qryIds / indexFile.ids). */
typedef struct lx_seq_names
{
    char const * const * q_ids;  /* per true query id (n_qid)   */
    uint64_t const *     q_lens; /* untranslated query lengths  */
    char const * const * s_ids;  /* per true subject id (n_sid) */
    uint64_t const *     s_lens;
    uint64_t             n_q, n_s;
} lx_seq_names;

enum
{
    LX_OUT_BLAST_TAB          = 0, /* -m8: 12 standard columns (src/search_options.hpp:716-760 default)         */
    LX_OUT_BLAST_TAB_COMMENTS = 1, /* -m9                                                                        */
    LX_OUT_SAM                = 2  /* SAM, default tags "AS NM ae ai qf" (src/search_options.hpp:351)            */
};
/* Appends header (if write_header) and records to `path` (myWriteHeader / myWriteRecord,
 * src/search_output.hpp:305-461, :463-733).  `ops` is the ops buffer the matches' ops_off index into; `program` is
 * "blastp", "blastn", "blastx", "tblastn" or "tblastx": positions of a translated side are reported in nucleotide
 * coordinates of the original sequence (names->q_lens / s_lens = untranslated lengths), start > end on the minus
 * strand in the tabular formats; SAM of a translated query carries the nucleotide-space CIGAR (runs x 3, frame
 * clips as H, reversed on the minus strand) and the covered part of the untranslated read. */
int lx_write_records(char const * path, int format, int write_header, char const * program, lx_blast_match const * m,
                     uint64_t n, uint8_t const * ops, lx_seq_names const * names, uint8_t const * q_res_ascii,
                     uint64_t const * q_ascii_off);

/* The output options of the reference's command line (src/search_options.hpp:224-379, parsed :716-816): which tabular columns,
 * which optional SAM tags, whether a SAM record carries the sequence and how it clips.  Initialise with
 * lx_output_options_default (= the reference's defaults: the 12 standard columns; tags "AS NM ae ai qf"; sequence "uniq";
 * hard clips; no @SQ lines), then change what the caller asked for. */
enum
{
    LX_SAM_SEQ_NEVER  = 0, /* --sam-bam-seq never  (:765-770) */
    LX_SAM_SEQ_UNIQ   = 1, /*               uniq: omitted iff frame and query range equal the previous match's (:536-553) */
    LX_SAM_SEQ_ALWAYS = 2
};
typedef struct lx_output_options
{
    char const * columns;             /* --output-columns: space-separated NCBI specifiers; NULL = "std".  Supported: std qseqid
                                         qlen sseqid slen qstart qend sstart send evalue bitscore score length pident nident
                                         mismatch positive gapopen gaps ppos frames qframe sframe staxids lcataxid            */
    char const * sam_tags;            /* --sam-bam-tags: space-separated keys of SamBamExtraTags (src/search_output.hpp:29-76):
                                         AS OC NM IH ar ae ai ap qf qs sf st ls lt; NULL = "AS NM ae ai qf" (:351)            */
    int32_t      sam_seq;             /* LX_SAM_SEQ_*                                                                         */
    int32_t      sam_hard_clip;       /* --sam-bam-clip hard (1, default) | soft (0)                                          */
    int32_t      sam_with_ref_header; /* --sam-with-refheader: one @SQ line per subject (:276-286)                            */
    int32_t      version_to_output;   /* --version-to-outputfile: `version` into the tabular version line / an @PG line       */
    char const * version;             /* (the reference writes SEQAN_APP_VERSION; this library has no lambda version: the
                                         caller names one)                                                                    */
    char const * command_line;        /* @PG CL: (src/search_output.hpp:391-399)                                              */
    char const * db_name;             /* "# Database:" of .m9 -- the reference writes the index path (search_algo.hpp:320)    */
    int32_t      genetic_code;        /* translated queries: the table the frames were made with (tags qs / OC)               */
    int32_t      reserved;
    /* taxonomy columns (staxids / st, lcataxid / lt, ls): NULL = "*" / 0, as for an index without taxonomy */
    lx_tax_tree const *  tax;
    uint64_t const *     lca_qid;     /* lx_compute_lca's output: n_lca (n_qid, taxon) pairs in list order                    */
    uint32_t const *     lca_tax;
    uint64_t             n_lca;
    char const * const * tax_names;   /* scientific name per taxon (tag ls), NULL = "*"                                       */
} lx_output_options;
void lx_output_options_default(lx_output_options * o);
/* lx_write_records with options (NULL = defaults).  LX_EINVAL also for an unknown column specifier or tag key (the reference
 * throws "Unknown column specifier", :755-758, :803-806); lx_last_output_error() has the text. */
int lx_write_records_ex(char const * path, int format, int write_header, char const * program, lx_blast_match const * m,
                        uint64_t n, uint8_t const * ops, lx_seq_names const * names, uint8_t const * q_res_ascii,
                        uint64_t const * q_ascii_off, lx_output_options const * opt);
/* What lx_write_records_ex refuses before it touches a file -- an unknown format, column specifier or SAM tag -- so that a front
 * end can fail while it parses its options, as the reference does (src/search_options.hpp:684-816), not after the search. */
int lx_check_output_options(int format, lx_output_options const * opt);
/* myWriteFooter (src/search_output.hpp:739-750): .m9 ends with "# BLAST processed N queries" (N = records written), the other
 * formats have no footer. */
int lx_write_footer(char const * path, int format, uint64_t n_records);
char const * lx_last_output_error(void);

/* ---- misc ------------------------------------------------------------------------------------ */
/* Blocks until everything queued on the handle's stream has finished. */
int lx_synchronize(lx_handle * h);
/* Duration in ms of the most recent score / align kernel launch sequence on this handle, measured with HIP
 * events on the launch stream (valid after lx_synchronize or a host-buffer call). */
int lx_last_kernel_ms(lx_handle * h, float * ms);
/* Name of the kernel instantiation the most recent pass-1 launch used, e.g. "lx::score_kernel<8,19,false> shared-profile"
 * (profiling aid: matches the kernel names rocprofv3 reports). */
char const * lx_last_kernel_name(lx_handle const * h);
char const * lx_last_trace_kernel_name(lx_handle const * h);
/* Device time (HIP events on the launch stream) the most recent call spent in one phase, summed over its launches:
 * phase 0 = pass-1 score kernel, 1 = survivor selection, 2 = pass-2 forward kernel, 3 = pass-2 backtrace kernel. */
int lx_last_phase_ms(lx_handle * h, int phase, float * ms, int * launches);

#ifdef __cplusplus
}
#endif
#endif /* LAMBDA_EXT_H */

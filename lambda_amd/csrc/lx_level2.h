// lx_level2.h -- device side of the Level-2 driver (lx_level2.hip) as the host sees it (lx_level2_host.cpp): the list work that
// iterateMatchesFullSimd does around the two DP passes (/root/reference/src/search_algo.hpp:1136-1175, :1200-1227), on the
// matches where the seeding stage left them -- in HBM.  Not part of the ABI; include/lambda_ext.h is.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lx_device.h"

namespace lx
{

// A match as the sort sees it.  After _widenMatch (src/search_algo.hpp:919-938) a match is a function of (qryId, subjId,
// s0 = the diagonal's subject position: subjStart - qryStart, clipped at 0) alone -- qryStart = 0, qryEnd = the query's length,
// subjStart = max(0, s0 - band), subjEnd = min(s0 + qLen + band, sLen) -- and both window ends are non-decreasing in s0, so the
// order of the reference's sort key (qryId, subjId, qryStart, qryEnd, subjStart, subjEnd; src/search_datastructures.hpp:60)
// IS the order of (qryId, subjId, s0); matches that tie under the reference's key are equal records.
//   word `pair` = bisulfite flag (subjId % 2, the major key of iterateMatches' bisulfite branch, :1369-1372) << 63 | qryId << 32 | subjId
//   word `s0`
// The sort is a least-significant-digit radix sort over the bits the call can have set (kL2DigitBits per pass; hand-written:
// per-tile digit counts -> one scan per digit over the tiles -> stable scatter with wavefront match masks).
constexpr int      kL2DigitBits  = 8;
constexpr int      kL2SortBlock  = 256;                       // threads per workgroup of the sort kernels
constexpr int      kL2SortItems  = 16;                        // keys per thread
constexpr uint64_t kL2SortTile   = (uint64_t)kL2SortBlock * kL2SortItems;
constexpr int      kL2ScanBlock  = 256;
constexpr int      kL2ScanItems  = 8;
constexpr uint64_t kL2ScanTile   = (uint64_t)kL2ScanBlock * kL2ScanItems;

// Mirrors lx_match (src/search_datastructures.hpp:46-61).
struct Match
{
    uint64_t qryId, subjId, qryStart, qryEnd, subjStart, subjEnd;
};

// One DP window after widen + merge + unique: what the reference's span holds when _widenAndPreprocessMatches returns
// (qryStart = 0 and qryEnd = the query's length are implied).
struct L2Window
{
    uint32_t q, s;     // frame-expanded ids
    uint64_t beg, end; // subjStart, subjEnd
};
static_assert(sizeof(L2Window) == 24, "L2Window layout");

// the two sequence sets, resident on the handle's device (lx_set_queries, lx_set_subject_seqs)
struct L2Sets
{
    uint64_t const * q_off;   // [n_qseq] where the (frame) query lies in the query residue buffer
    uint32_t const * q_len;   // [n_qseq]
    uint32_t const * q_band;  // [n_qseq] _bandSize(q_len), src/search_misc.hpp:46-50 (made on the host: the very double sqrt)
    uint32_t const * q_evlen; // [n_qseq] the query length the e-value is computed for (bm.qLength, :1213): index of the cut-off table
    uint64_t const * s_off;   // [n_sseq]
    uint64_t const * s_len;   // [n_sseq]
    uint64_t         n_qseq, n_sseq;
};

struct L2Params
{
    L2Sets           sets;
    uint64_t         n;          // matches
    Extension *      ext_out;    // [<= n] the DP windows as (query slice, subject slice) pairs, :1200-1227
    int32_t *        min_out;    // [<= n] score cut-off per window (the filter of :1251-1283 as an integer test)
    L2Window *       win_out;    // [<= n]
    int32_t const *  cut_by_len; // [max q_evlen + 1]
    uint64_t *       count_out;  // [0] = windows, [1] = first window of an odd subject frame (bisulfite split), [2] = error flag
    int              bisulfite;
};

hipError_t l2_launch_keys(void const * d_matches, L2Params const & p, uint64_t * pair, uint64_t * s0, uint64_t * flag, hipStream_t stream);
// sorts (pair, s0) by the bits set in pair_bits / s0_bits (masks of the bits any key of the call can have set); the sorted words
// end up in *pair / *s0 (the two buffers of a word swap roles); ghist: [(tiles + 1) * 256 + 256] uint32
hipError_t l2_launch_sort(uint64_t ** pair, uint64_t ** pair_tmp, uint64_t ** s0, uint64_t ** s0_tmp, uint64_t n, uint64_t pair_bits, uint64_t s0_bits,
                          uint32_t * ghist, hipStream_t stream);
// merge passes + unique + windows; mrg_* / fin_*: [n] uint64 each (the span after merge right / after swallow left), block_tot: [tiles of
// kL2ScanTile + 1] uint32
hipError_t l2_launch_merge(uint64_t const * pair, uint64_t const * s0, L2Params const & p, uint64_t * mrg_beg, uint64_t * mrg_end, uint64_t * fin_beg,
                           uint64_t * fin_end, uint32_t * block_tot, hipStream_t stream);
// the solo plan of the multi-query sweep for a window list: cost sums per part and strip geometry (out[8]: part 0's
// cost at 19 / 13 / 11 columns and cells, then part 1's; cnt = {windows, first window of part 1} in device memory), and the plan
hipError_t l2_launch_plan_cost(Extension const * ext, uint64_t const * cnt, uint64_t n_max, int no_narrow, unsigned long long * out, hipStream_t stream);
hipError_t l2_launch_plan(Extension const * ext, uint64_t n, int C, int no_narrow, uint64_t ** key, uint64_t ** key_tmp, uint64_t ** idx, uint64_t ** idx_tmp,
                          uint32_t * ghist, uint32_t * plan, uint32_t * wf_pan, uint32_t * wf_maxs, hipStream_t stream, uint32_t index_base = 0, uint64_t key_bits = 0x0fffffffull);
uint64_t   l2_plan_key_bits(uint64_t max_qlen, uint64_t max_wlen);
// The FREE-PACKING plan of the multi-query sweep for a window list (lx_plan_free.hip): a lane group's two windows
// share a query, a wavefront's sixteen slots hold windows of at most four queries -- the long windows of every query pooled in
// quads, the rest streamed pair by pair --, laid out range by range (cut[0] = 0 < cut[1] < ... < cut[nranges] = n, every cut where
// the query changes).  plan: [cap_wf * 16] slots (position in ext[], bit 31 = filler), wf_pan / wf_maxs: [cap_wf]; report: [16]
// uint32 in device memory -- [0] wavefronts, [1] non-zero = the bound was broken (a bug: the plan is not to be used), [2] runs,
// [3] quads, [4 + r] first wavefront of range r (r <= nranges).
constexpr uint32_t kFpMaxRanges = 8;
struct FpArgs
{
    Extension const * ext;
    uint64_t          n;
    uint64_t          n_qseq;    // query sequences of the resident set (bounds the runs)
    int               C;         // columns per lane of a whole strip
    int               no_narrow;
    uint32_t          nranges;
    uint64_t          cut[kFpMaxRanges + 1];
    void *            work;      // fp_workspace_bytes(n, n_qseq, nranges)
    size_t            work_bytes;
    uint32_t *        plan;
    uint64_t          cap_wf;    // >= fp_wavefront_bound(n, n_qseq, nranges)
    uint32_t *        wf_pan;
    uint32_t *        wf_maxs;
    uint32_t *        report;
};
uint64_t   fp_run_bound(uint64_t n, uint64_t n_qseq);
uint64_t   fp_wavefront_bound(uint64_t n, uint64_t n_qseq, uint32_t nranges);
size_t     fp_workspace_bytes(uint64_t n, uint64_t n_qseq, uint32_t nranges);
hipError_t fp_launch_plan(FpArgs const & p, hipStream_t stream);
uint64_t   l2_sort_tiles(uint64_t n);
uint64_t   l2_scan_tiles(uint64_t n);

// ---- the tail of iterateMatchesFullSimd on the device (lx_records.hip): statistics, order, records ------------------------------

// lx_blast_match (include/lambda_ext.h) as the device writes it; the host asserts the two layouts equal
struct BlastMatchDev
{
    uint64_t qry_id, subj_id, n_qid, n_sid, q_start, q_end, s_start, s_end;
    int32_t  score, alignment_length, num_matches, num_mismatches, num_positives, num_gap_opens, num_gap_extensions;
    float    identity;
    double   bit_score, e_value;
    uint64_t ops_off;
    uint32_t n_ops;
    int16_t  q_frame, s_frame;
};
static_assert(sizeof(BlastMatchDev) == 128, "BlastMatchDev layout");

enum
{
    kRecSurvivors = 0, // stored entries that are survivors (the rest is padding of the chunks' lists)
    kRecFailedBit = 1, // :1260
    kRecFailedEv  = 2, // :1274
    kRecKept      = 3, // records behind the identity cut-off
    kRecOps       = 4, // their alignment columns
    kRecErr       = 5, // bit 0: an entry that could not be traced, bit 1: an entry names a window outside the list
    kRecCounters  = 8
};

struct RecParams
{
    // the survivors as the extension pipeline left them (all chunks of the call): alignment, window, where the codes begin
    Hsp const *      hsp;
    uint32_t const * src;
    uint64_t const * codes_off; // NULL: the alignment's own ops_shift (offsets inside ONE chunk's code bytes)
    uint64_t const * count_ptr; // NULL, or where the number of filled entries stands (a chunk's list: the rest of n_entries is stale)
    uint64_t         n_entries;
    uint32_t         src_base;  // list position of win[0]: `src` counts from the list's start, the windows below from the range's
    // the windows of this part of the list, their scores of pass 1 and the filter's cut-offs
    L2Window const * win;
    int32_t const *  score;
    int32_t const *  min_score;
    uint64_t         n_win;
    uint32_t const * q_len;   // [n_qseq]
    uint32_t const * q_evidx; // [n_qseq] index of the query's e-value length among the distinct ones (pre_by_len)
    uint32_t         q_frames, s_frames, n_qid_end; // n_qid_end: one past the largest true query id
    int              q_mode, s_mode;
    int32_t          bit_cut;   // scores below it fail the bit-score test (INT32_MIN: no such test)
    int32_t          id_cutoff;
    int              want_ops;
    double           lambda, log_k, log_2;
    double const *   pre_by_len; // K * (ql - adj) * (dl - adj) per distinct e-value length, the host's doubles
    double const *   exp_tab;    // exp(-lambda s), s < exp_n, the host's doubles
    uint32_t         exp_n;
    uint64_t         ops_base;   // first column of this part in the result's ops
    // the windows' places in the records' order (rec_launch_rank), from this range's first window on: the survivors are then sorted by
    // ONE three-digit word instead of (true query id, query slice length | subject slice length, window).  NULL: by those words.
    uint32_t const * rank;
    uint32_t         rank_pad;   // what an entry that is no survivor sorts by: one past the largest rank
    // work and results
    uint32_t *       list_at;    // [n_win]
    uint64_t *       counters;   // [kRecCounters]
    BlastMatchDev *  rec;        // [<= survivors]
    uint64_t *       rec_codes;  // [3 x (<= survivors)] per record: where its codes begin, where its columns go, how many columns
};

hipError_t rec_launch_append(Hsp const * hsp, uint32_t const * src, uint64_t const * count_ptr, uint64_t cap, uint64_t code_base, Hsp * out_hsp, uint32_t * out_src,
                             uint64_t * out_codes, hipStream_t stream);
hipError_t rec_launch_add_ops_base(BlastMatchDev * rec, uint64_t n, uint64_t base, hipStream_t stream);
constexpr uint32_t kRecRankGroup = 256; // a query's windows on either side of one of them that rec_launch_rank looks at
// rank[w] = the place of window w of win[0 .. n) in the records' order (true query id = q / q_frames, query slice length, subject slice
// length, list position) -- for a list whose true query ids never descend: a window's place among the windows of its query, counted,
// + where the query's windows begin.  *too_long (zeroed by the caller) becomes non-zero where a query has more windows than the
// kernel looks at: the ranks are then not to be used.
hipError_t rec_launch_rank(L2Window const * win, uint64_t n, uint32_t q_frames, uint32_t const * q_len, uint32_t * rank, uint32_t * too_long, hipStream_t stream);
hipError_t rec_launch(RecParams const & p, uint64_t ** pair, uint64_t ** pair_tmp, uint64_t ** s0, uint64_t ** s0_tmp, uint64_t pair_bits, uint64_t s0_bits,
                      uint32_t * ghist, uint32_t * tile_keep, uint64_t * tile_ops, hipStream_t stream);

} // namespace lx

// lx_device.h -- structures shared between the host side of the C ABI and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace lx
{

constexpr int kAlph    = 32;          // matrix stride (include/lambda_ext.h LX_ALPH)
constexpr int kNegPad  = -100;        // substitution score of any pad rank: can never start/extend/end a best local alignment
// longest subject window pass 2 takes (direction-bit mode; merged windows of _widenAndPreprocessMatches have no bound
// in the reference, src/search_algo.hpp:1153-1157): row indices are 32-bit, the skewed values x 4 stay far from overflow
constexpr int kMaxTraceRows = 1 << 22;
constexpr int kNegInf  = -(1 << 28);  // "minus infinity" for gap states; far from int32 overflow after any number of additions

// Device copy of one scoring scheme, preprocessed for the row-skewed recurrence (see lx_score.hip).
struct ScoringDev
{
    int32_t alph;     // valid ranks 0..alph-1; rank `alph` is used as the pad letter
    int32_t go;       // SeqAn scoreGapOpen   (first gap character), negative
    int32_t ge;       // SeqAn scoreGapExtend (further characters), negative
    int32_t g2;       // go - ge
    int8_t  mat[kAlph * kAlph];     // original matrix[q*32+s], pad ranks = kNegPad
    int8_t  mat_adj[kAlph * kAlph]; // matrix[q*32+s] - ge (diagonal step in the skewed domain), pad ranks = kNegPad
    // pass 2 works on values scaled by 4 whose two low bits carry the traceback tag: 4*(matrix - ge) + 3 (tag
    // "diagonal"), pad ranks = -125.  Valid when every (matrix - ge) lies in [-31, 31] (trace_ok).
    int8_t  mat_trace[kAlph * kAlph];
    int32_t trace_ok;
    // packed-half pass 1 (lx_score_f16.hip): (matrix - ge) as IEEE half bits, pad ranks = -100; per query rank the
    // largest positive score of its row (upper bound of what one query column can contribute); largest entry
    int32_t  smax;
    int32_t  b8_ok;       // mat_b8 is valid: every real entry satisfies 0 <= matrix - go <= 255
    int32_t  reserved;
    int16_t  rowmax[kAlph];
    uint16_t mat_h[kAlph * kAlph];
    int16_t  mat_i16[kAlph * kAlph]; // packed-int16 sweep (lx_score_i16.hip): matrix - ge, pad ranks = kNegPad
    // multi-query sweep (lx_sweep_mq.hip): matrix[q*32+s] - go as an unsigned byte (go = cost of a gap's first character),
    // pad ranks = 0 -- a pad cell scores like opening a gap
    uint8_t  mat_b8[kAlph * kAlph];
};

// Mirrors lx_extension in include/lambda_ext.h (static_assert'ed in lx_api.cpp).
struct Extension
{
    uint64_t q_off;
    uint64_t s_off;
    uint32_t q_len;
    uint32_t s_len;
};

// Mirrors lx_hsp.
struct Hsp
{
    int32_t score;
    int32_t q_begin, q_end;
    int32_t s_begin, s_end;
    int32_t n_ops;
    int32_t num_matches, num_mismatches, num_positives;
    int32_t num_gap_opens, num_gap_extensions;
    int32_t ops_shift;
};

// Multi-query sweep, slots by WAVEFRONT (round 5): the 16 slots of wavefront w begin at off_dw (in uint32 from the launch's slot
// base) and are laid out for steps_cap steps and panels_cap panels -- what w's own longest window and widest query need, not the
// chunk's.  A chunk of a ragged list then takes the sum of what its wavefronts need (a tenth of them are three times as long as the
// rest) instead of slots x the chunk's maximum, and one chunk -- one backtrace launch -- holds what took three.
struct WfSlots
{
    uint64_t off_dw;
    uint32_t steps_cap;
    uint32_t panels_cap;
};

struct ScoreParams
{
    uint8_t const *    q_res;
    uint8_t const *    s_res;
    Extension const *  ext;
    uint64_t           n;
    ScoringDev const * sc;
    int32_t *          out_score;
    // multi-panel carry workspace (only touched by extensions whose query is wider than one panel)
    int32_t *          ws;       // int32 pairs
    uint32_t *         ws_top;   // bump pointer, in int32 pairs
    uint32_t           ws_cap;   // capacity, in int32 pairs
    int32_t *          err;      // set to 1 on workspace overflow
    int32_t            shared_profile; // 1: every group of a wave uses the same query -> one profile slot per wave
    int32_t            nrows;          // profile rows = alph + 1 (host copy of sc->alph + 1, sizes the LDS slot)
    int32_t            fixup;          // 1: only extensions whose out_score is the sentinel -1 are (re)computed
    int32_t            pair_share;     // packed-half kernel: lane groups per LDS profile (0 = the whole wavefront)
    int32_t            narrow;         // multi-query sweep: 1 = a query's last panel may run narrower strips (kEndNarrowShift)
    int32_t            solo;           // multi-query sweep: 1 = every window has a profile of its own (16 per wavefront: the small alphabets)
    int32_t            wide;           // multi-query sweep: 1 = int16-pair slots (scores beyond the compact codes' 2046 at sweep speed)
    uint32_t *         stat_beyond;    // multi-query sweep: optional counter of the windows that are beyond the compact codes (declined, or scoring > 2046)
    // multi-query sweep: slots by wavefront (WfSlots) -- wf_tab[w] for wavefront w of the chunk, slots relative to `ckpt`; the launch
    // covers the wavefronts from wf_lo on (a chunk may be swept by two launches: the plan's pool while the rest is still being planned)
    WfSlots const *    wf_tab;
    uint32_t           wf_lo;
    // single sweep (lx_ckpt.hip layout): when set, the packed-half kernel also writes strip boundaries, row checkpoints
    // (as the compact 16-bit codes of Ckpt16Layout) and the end cell of every extension
    uint32_t *         ckpt;        // [n + 1] slots of ckpt_stride uint32 (the last one is the spare slot idle halves write to)
    uint64_t           ckpt_stride;
    // 1 (set for the packed-half sweep, which always lays its slots out like this): the W = 2 * 64 / G slots of a wavefront's windows are INTERLEAVED piece by piece -- group
    // of eight steps o of window w at [o][w][lane], row checkpoint m at [m][w][lane][quad] behind all boundary groups -- so that
    // what one store instruction writes (the same piece of 16 windows) is one contiguous 2 KB instead of 16 lines 7.7 KB apart;
    // the buffer holds ceil(n / W) * W slots, no spare one.  The end cell says so (kEndWaveSlots), the backtrace reads accordingly.
    int32_t            wave_slots;
    uint32_t           steps_cap;
    struct EndCell *   ends;        // [n]
    uint32_t           panels_cap;  // packed-int16 sweep: bound on ceil(Lq / panel); the slot holds that many parts
    // band mode (LX_OPT_BAND; not the reference's configuration): only cells with |(i - j) - diag| <= band exist
    int32_t            band;        // half width in diagonals, 0 = off (full rectangle)
    int32_t const *    band_diag;   // optional [n]: centre diagonal per extension; nullptr = band_default_diag()
};

// best cell of one extension, written by the forward-trace kernel, consumed by the backtrace kernel
struct EndCell
{
    int32_t score;
    int32_t q_end; // 1-based column of the best cell == half-open end; -(strip + 1) while only the strip is known
    int32_t s_end; // 1-based row of the best cell; while only the strip is known: the last row of the block of sixteen steps whose rows
                   // reached the best score first (packed sweeps) or the first such row itself (int32 sweep) -- the backtrace's first tile is
                   // that block and finds the cell (lowest column, then lowest row)
    int32_t flags; // kEndAmbiguous: the strip reaches the best score again in a later block (packed sweeps) / row (int32): the backtrace scans on
};
constexpr int32_t kEndAmbiguous = 1;
constexpr int32_t kEndCompact      = 2; // the slot was written by the packed-half sweep: compact 16-bit codes (Ckpt16Layout)
constexpr int     kEndOverflowShift = 8; // flags >> 8 = 1 + index of the extension's slot in the overflow area (0 = none)
// (flags >> 2) & 3: the strips of the extension's LAST panel are C (0), (C + 1) / 2 (1) or (C + 3) / 4 (2) columns wide -- the
// multi-query sweep runs a query's last panel with narrower strips when that covers what is left of the query (a query of 160
// columns sweeps 152 + 8 * 5 = 192 columns, not 304); the backtrace maps columns to strips accordingly.  Panel starts do not move.
constexpr int     kEndNarrowShift   = 2;
// the compact slots of the extension's WAVEFRONT are interleaved (ScoreParams::wave_slots): bit 4
constexpr int32_t kEndWaveSlots     = 16;
__host__ __device__ constexpr int narrow_strip_cols(int C, int code) { return code == 0 ? C : code == 1 ? (3 * C + 3) / 4 : code == 2 ? (C + 1) / 2 : (C + 3) / 4; }
constexpr int kNarrowest = 3; // the code of the narrowest strips
// the code for a panel that has `rem` columns of the query left (rem >= 1), G lanes per group
__host__ __device__ constexpr int narrow_code_for(int C, int G, int rem)
{
    return rem <= G * ((C + 3) / 4) ? 3 : rem <= G * ((C + 1) / 2) ? 2 : rem <= G * ((3 * C + 3) / 4) ? 1 : 0;
}

// Compact checkpoint slots of the packed-half single sweep (lx_score_f16.hip writes, lx_ckpt.hip's backtrace reads).
// A boundary pair (H of a strip's last column, E entering the next strip) and a row-checkpoint pair (H of a cell, the
// folded F entering the cell below) always satisfy  0 <= H - other <= |cost of a gap's first character|  (E >= H + go and
// E <= H + ge because E of the same cell is <= H; likewise F), and the packed-half kernel only runs when every value
// is <= 2046 -- so a pair is the 16-bit code  H | (H - other) << 11  (11 + 5 bits; needs go >= -31).
//   boundary codes:  [step / 8][lane] 16-byte groups of 8 steps (lane-minor: the G lanes of a group fill whole lines);
//   row checkpoints: [checkpoint][lane][quad], two columns per dword, every 16 steps (lane-major).
constexpr int      kC16MaxGap = 31;
template <int G, int C>
struct Ckpt16Layout
{
    static constexpr int kCkDw = ((C + 1) / 2 + 3) / 4 * 4; // dwords per lane per row checkpoint, whole quads
    __host__ __device__ static constexpr uint64_t bnd_dwords(uint32_t steps_cap) { return (uint64_t)steps_cap / 2 * G; }
    __host__ __device__ static constexpr uint64_t slot_dwords(uint32_t steps_cap)
    {
        return bnd_dwords(steps_cap) + (uint64_t)(steps_cap / 16) * G * kCkDw;
    }
    // uint4 index of the group of steps 8 o .. 8 o + 7 of lane g / of quad x of lane g's row checkpoint m
    __host__ __device__ static constexpr uint32_t bnd_oct_index(uint32_t o, uint32_t g) { return o * G + g; }
    __host__ __device__ static constexpr uint32_t rowck_quad_index(uint32_t m, uint32_t g, uint32_t x)
    {
        return (m * G + g) * (kCkDw / 4) + x; // lane-major (quad-major was measured in round 2: no difference)
    }
};

struct TraceParams
{
    uint8_t const *    q_res;
    uint8_t const *    s_res;
    Extension const *  ext;   // this chunk's extensions
    uint64_t           n;     // extension slots in this chunk
    ScoringDev const * sc;
    uint32_t *         trace;       // [n][panels_cap][steps_cap][G][words] direction words
    uint64_t           slot_stride; // uint32 entries per extension = panels_cap * steps_cap * G * words
    uint32_t           steps_cap;   // bound on (Ls + G - 1) rounded up to 4
    uint32_t           panels_cap;  // bound on ceil(Lq / panel)
    EndCell *          ends;        // [n]
    Hsp *              out_hsp;     // indexed by src[e] when src != nullptr, else by e
    uint8_t *          out_ops;
    uint64_t const *   ops_off;     // byte offset of each extension's ops slot (slot size q_len + s_len), same indexing;
                                    // nullptr: slots of ops_stride bytes, slot = index * ops_stride
    uint64_t           ops_stride;
    int32_t const *    score_in;    // optional: best score of each slot (pass 1) -> cheaper end-cell search
    uint32_t const *   src;         // optional: original index of each slot, 0xffffffff = padding slot (skipped)
    uint64_t const *   count_ptr;   // optional: device-side number of valid slots of the whole list
    uint64_t           chunk_start; // position of this chunk in that list
    int32_t *          ws;
    uint32_t *         ws_top;
    uint32_t           ws_cap;
    int32_t *          err;
    int32_t            nrows;
    int32_t            bs_match_rule; // computeAlignmentStats variant: 1 = match iff score(c0,c1)==score(c0,c0)
    int32_t            shared_profile;
    int32_t            cfg;           // 0 = (16,10), 1 = (8,19)
    // single-sweep mode (lx_ckpt.hip): the forward kernel runs over ALL extensions without a known score, writes the
    // best score here and an end cell whose column is still to be resolved; the backtrace then addresses checkpoint
    // slots and end cells by src[e] instead of by e
    int32_t *          score_out;
    int32_t            slot_by_src;
    int32_t            fixup;         // single sweep, int32 kernel: only extensions whose score_out is the sentinel -1
    int32_t            out_by_pos;    // backtrace: write out_hsp / read ops_off by list position e instead of by src[e]
    // single sweep with compact slots: the extensions the packed-half kernel declined get int16-pair slots here
    uint32_t *         ovf;           // overflow area: ovf_cap slots of ovf_stride uint32
    uint64_t           ovf_stride;
    uint32_t           ovf_cap;
    uint32_t *         ovf_count;     // device counter of handed-out overflow slots
    int32_t            band;          // band mode, as in ScoreParams (direction-bit kernels only)
    int32_t const *    band_diag;
    // checkpoint backtrace, lx_extend_batch: the ops leave as run-length codes (lx_pack.hip's format) in a dense stream --
    // rle != nullptr selects it; the slots of out_ops then only stage the codes, Hsp::ops_shift is the offset in the stream
    uint8_t *            rle;
    unsigned long long * rle_top;
    uint64_t             rle_cap;
    uint32_t *           rle_len;     // [list capacity]: code bytes of every position (0 for padding slots / no alignment)
    int32_t            bt_tile_at, bt_refill_at; // checkpoint backtrace scheduling thresholds (0 = the compiled defaults)
    uint32_t *         work_counter;  // checkpoint backtrace: the queue its persistent lanes take list positions from (zeroed per launch)
    // single-sweep backtrace: the multi-query sweep's slots by wavefront (ScoreParams::wf_tab): extension se has slot se % 16 of
    // wavefront se / 16; the wavefronts from split_n / 16 on (a chunk swept in two calls: the second launch's slots) count their
    // offsets from trace2, the others from trace; split_n = 0: all from trace
    WfSlots const *    wf_tab;
    uint64_t           split_n;
    uint32_t *         trace2;
};

// survivor selection between the passes (the filter loop of iterateMatchesFullSimd, src/search_algo.hpp:1251-1283,
// with the e-value / bit-score tests turned into per-extension integer score cut-offs by the host)
struct SelectParams
{
    Extension const * ext;
    int32_t const *   score;
    int32_t const *   min_score; // per extension, or nullptr -> min_score_all
    int32_t           min_score_all;
    uint64_t          n;
    uint32_t          run;       // extensions come in runs of `run` entries sharing a query (0/1 = no runs)
    uint32_t          pad_to;    // pad every run's survivors to a multiple of this many slots (1 = no padding)
    uint64_t *        run_slots; // [nruns] scratch: padded survivor count per run
    uint64_t *        block_tot; // [2 * select_blocks(nruns)] scratch: per workgroup (slots, survivors) -> slot offset
    Extension *       out_ext;   // [capacity]
    int32_t *         out_score; // [capacity] score of each slot (0 for padding slots)
    uint32_t *        out_src;   // [capacity]
    uint64_t *        out_count; // [0] = total slots, [1] = true survivors
    Hsp *             out_hsp;   // optional [n]: rows of non-survivors are filled here (score, no alignment)
};

// run-length packing of the survivors' op bytes (lx_pack.hip); records and ops slots are addressed by list position
struct PackParams
{
    Hsp *                hsp;       // [n] records by position; ops_shift is rewritten to the offset in the code stream
    uint8_t const *      ops;       // op bytes as the backtrace left them
    uint64_t const *     ops_off;   // slot of position e, or nullptr: e * ops_stride
    uint64_t             ops_stride;
    uint32_t const *     src;       // optional: 0xffffffff marks a padding slot of the list
    uint64_t const *     count_ptr; // optional device-side list length
    uint64_t             n;         // capacity of the list
    uint8_t *            rle;       // dense code stream
    uint32_t *           rle_len;   // [n]: code bytes of every position
    unsigned long long * rle_top;   // bytes handed out (zeroed by the host before the launch)
    uint64_t             rle_cap;
    int32_t *            err;
};

struct MaxLens
{
    uint32_t max_q, max_s;
};

struct PrefilterSeed
{
    uint64_t q_off, s_off;
    uint32_t q_len, s_len;
    uint32_t qry_start, qry_end, subj_start, reserved;
};

struct PrefilterParams
{
    uint8_t const *       q_res;
    uint8_t const *       s_res;
    PrefilterSeed const * seeds;
    uint64_t              n;
    ScoringDev const *    sc;
    uint32_t              seed_length;
    int32_t               pre_scoring;
    double                pre_scoring_thresh;
    uint8_t *             out_keep;
};

} // namespace lx

// lx_aids.h -- development aids: the environment switches of the library, read ONCE per process (first use) in one place
// (lx::dev_aids() in lx_api.cpp).  None of them is part of the ABI and none changes results: they pick between bit-identical
// kernels / schedules for A/B measurements (tools/), or print host timings.  Options a caller is meant to
// set go through lx_set_option (include/lambda_ext.h).
#pragma once
#include <cstddef>
#include <cstdint>

namespace lx
{

struct DevAids
{
    // variable                 meaning                                                                      default
    size_t   pair_lds_limit;    // LX_PAIR_LDS_LIMIT     LDS a wavefront of the packed-half kernel may spend on profiles   24 KiB
    int      force_score_cfg;   // LX_FORCE_SCORE_CFG    pass-1 geometry for every list (-1 = pick by query width)         -1
    int      force_mq_cfg;      // LX_FORCE_MQ_CFG       multi-query sweep geometry 1 = (8,19), 3 = (8,13), 5 = (8,11) (0 = pick)  0
    int      mq_set;            // LX_MQ_SET             strip widths lx_extend_batch's multi-query plan picks from: 1 = 19, 2 = 13, 4 = 11 columns (sum)  7
    int      force_ckpt_cfg;    // LX_FORCE_CKPT_CFG     checkpoint geometry 1 = (8,19), 2 = (16,13) (0 = pick)            0
    bool     trace_overlap;     // LX_TRACE_OVERLAP=1    mode-0 pass 2: forward of chunk k+1 beside the backtrace of k     off
    uint64_t trace_chunks;      // LX_TRACE_CHUNKS       mode-0/1 pass 2: at least this many chunks                        1
    bool     no_wide_strips;    // LX_NO_WIDE_STRIPS     153-200 column queries: (16,13) strips instead of (8,25)                off
    bool     no_wide_compact;   // LX_NO_WIDE_COMPACT    queries wider than a panel: int16-pair slots instead of compact codes    off
    bool     no_i16_sweep;      // LX_NO_I16_SWEEP       wide queries: int32 sweep instead of the packed 16-bit one        off
    int      pass2_mode;        // LX_PASS2_MODE         initial value of LX_OPT_PASS2_MODE (-1 = the library's default)   -1
    unsigned host_threads;      // LX_HOST_THREADS       cap of the host pool (0 = the affinity mask, at most 16)           0
    bool     extend_no_classes; // LX_EXTEND_NO_CLASSES  lx_extend_batch: no geometry-class binning of ragged lists        off
    bool     extend_no_sort;    // LX_EXTEND_NO_SORT     lx_extend_batch: no in-run sort by window length                  off
    bool     mq_no_narrow;      // LX_MQ_NO_NARROW       multi-query sweep: every panel at full strip width (no narrow last panel)  off
    bool     mq_no_solo;        // LX_MQ_NO_SOLO         multi-query sweep: no solo packing (a profile per window) for the small alphabets  off
    bool     mq_no_wide;        // LX_MQ_NO_WIDE         multi-query sweep: compact codes always (what scores beyond them goes to the int32 launch)  off
    bool     mq_no_wfslots;     // LX_MQ_NO_WFSLOTS      lx_extend_batch: the multi-query chunks' checkpoint slots by region (rounds 3-4) instead of by wavefront  off
    bool     mq_no_two_calls;   // LX_MQ_NO_TWO_CALLS    lx_extend_batch: the pool a chunk of its own (round 4) instead of the first of a chunk's two calls  off
    bool     mq_no_longest_first; // LX_MQ_NO_LONGEST_FIRST lx_extend_batch: the pool's wavefronts launched in packing order (panels, window length) instead of longest first  off
    bool     mq_no_merge;       // LX_MQ_NO_MERGE        lx_extend_batch: the pool's wavefronts in launches of their own (no two-region chunk)  off
    uint64_t mq_merge_below;    // LX_MQ_MERGE_BELOW     lx_extend_batch: lists of at most this many windows launch the pool with the rest (0 = 200 000)  0
    bool     iterate_on_host;   // LX_ITERATE_ON_HOST    lx_iterate_matches: widen / sort / merge on the host threads whatever the list's size  off
    bool     extend_no_mq;      // LX_EXTEND_NO_MQ       lx_extend_batch: ragged lists on the one-query-per-wavefront kernels      off
    uint64_t extend_run;        // LX_EXTEND_RUN         lx_extend_batch: pad query runs to 8 or 16 slots (0 = by estimated work)  0
    uint64_t extend_chunk;      // LX_EXTEND_CHUNK       default of LX_OPT_EXTEND_CHUNK (extensions per pipeline chunk)    640 Ki
    int      bt_waves_per_cu;   // LX_BT_WAVES_PER_CU    persistent wavefronts of the backtrace per CU (0 = what fits)     0
    int      bt_tile_at;        // LX_BT_TILE_AT         lanes waiting for a tile that start the tile phase (0 = default)  0
    int      bt_refill_at;      // LX_BT_REFILL_AT       retired lanes that trigger a queue refill (0 = default)           0
    uint64_t l2_ranges;         // LX_L2_RANGES          Level-2 driver: ranges of the window list whose records are made chunk by chunk (0 = by size: 2 from 300 000 windows, one more per two million, at most 4)  0
    uint64_t l2_first_pct;      // LX_L2_FIRST_PCT       Level 2, two ranges: the first range's share of the windows in percent                          0 = 66
    bool     l2_no_rank;        // LX_L2_NO_RANK         Level 2: a range's survivors sorted by (query id, slice lengths, window) words instead of their windows' ranks  off
    bool     host_timing;       // LX_HOST_TIMING        print where the host-buffer entry points spend their time         off
};

DevAids const & dev_aids();

} // namespace lx

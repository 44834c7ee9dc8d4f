// lx_aids.h -- development aids: the environment switches of the library, read ONCE per process (first use) in one place
// (lx::dev_aids() in lx_api.cpp).  None of them is part of the ABI and none changes results: they steer measurements of the
// committed tools (tools/, tests/test_gpu_two_calls.py) or print host timings.  Options a caller is meant to set go through
// lx_set_option (include/lambda_ext.h).
#pragma once
#include <cstddef>
#include <cstdint>

namespace lx
{

struct DevAids
{
    // variable                 meaning                                                                                     default   used by
    unsigned host_threads;      // LX_HOST_THREADS       parts of a host loop (what LX_OPT_HOST_THREADS sets for a caller)          0         tools/host_curve.py
    bool     mq_no_wide;        // LX_MQ_NO_WIDE         multi-query sweep: compact codes always (windows beyond them: int32 launch) off       tools/profile_round.sh
    uint64_t mq_merge_below;    // LX_MQ_MERGE_BELOW     lx_extend_batch: lists of at most this many windows launch the pool with    200 000   tests/test_gpu_two_calls.py
                                //                       the rest (0 = default); above: the pool first, in the first of two calls
    bool     iterate_on_host;   // LX_ITERATE_ON_HOST    lx_iterate_matches: widen / sort / merge on the host threads whatever the   off       tools/profile_round.sh
                                //                       list's size
    int      bt_tile_at;        // LX_BT_TILE_AT         backtrace: lanes waiting for a tile that start the tile phase (0 = default) 0         tools/dev/bt_sweep.sh
    int      bt_refill_at;      // LX_BT_REFILL_AT       backtrace: retired lanes that trigger a queue refill (0 = default)          0         tools/dev/bt_sweep.sh
    uint64_t l2_ranges;         // LX_L2_RANGES          Level-2 driver: ranges of the window list whose records are made chunk by   0         tools/dev/pmc_prot.sh
                                //                       chunk (0 = by size: 2 from 300 000 windows, one more per four million, at most 4)
    bool     host_timing;       // LX_HOST_TIMING        print where the host-buffer entry points spend their time                   off       tools/profile_round.sh, tests
};

DevAids const & dev_aids();

} // namespace lx

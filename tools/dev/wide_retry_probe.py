"""Development aid: a list whose windows all score beyond the compact codes under a tight slot budget -- does the overflow area run out,
and does the chunk run again with int16-pair slots?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62), 0)
h.set_option(capi.LX_OPT_PASS2_MODE, 2)
q, s, ext = synth.make_ragged_lists_np(300, seed=15, lq_range=(500, 700), mean_windows=8.0)
q[:] = 22
s[:] = 22
h.set_option(capi.LX_OPT_TRACE_BYTES, int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 600 << 20)
for rep in range(2):
    score, index, hsp, off, codes = h.extend_batch_list(q, s, ext, 91)
    print(rep, len(ext), int((score > 2046).sum()), len(index), h.last_trace_kernel_name()[:70], h.last_extend_stats())

"""Development aid: a ragged list of LONG queries (500-800 residues) whose homologous windows score beyond the compact codes' 2046:
how much of the list the int32 fix-up launch has to redo, and what that costs."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500, 800)
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62), 0)
h.set_option(capi.LX_OPT_TRACE_BYTES, 160 << 30)  # (as bench.py: the slots of such a list in one chunk need ~70 GB of the 288)
q, s, ext = synth.make_ragged_lists_np(8000, seed=5, lq_range=(lo, hi), mean_windows=8.0)
cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
h.set_subjects(s)
best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    score, index, hsp, off, codes = h.extend_batch_list(q, None, ext, 91)
    best = min(best, time.perf_counter() - t0)
st = h.last_extend_stats()
print(f"queries of {lo}-{hi}: {len(ext)} windows, {cells / 1e9:.1f} Gcells, {len(index)} survivors, {int((score > 2046).sum())} beyond 2046: "
      f"{best * 1e3:.1f} ms = {cells / best / 1e9:.0f} GCUPS, padding {100 * (1 - st[2] / st[3]):.1f} %  [{h.last_trace_kernel_name()[:90]}]")

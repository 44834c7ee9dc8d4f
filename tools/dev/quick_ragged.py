"""lx_extend_batch on a ragged seed list (mixed query lengths 50-400, geometric window counts, merged windows): GCUPS and the
share of executed cells that is padding.  Development aid for DESIGN.md section 5; `bench.py --ragged` prints the line."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
q, s, ext = synth.make_ragged_lists_np(nq, seed=0x1A3BDA07)
cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
h.set_subjects(s)
r = h.extend_batch(q, None, ext, 91, copy_ops=False)
keep = r[:3]
best = 1e9
for rep in range(4):
    t0 = time.perf_counter(); r = h.extend_batch(q, None, ext, 91, copy_ops=False, out=keep); best = min(best, time.perf_counter() - t0)
st = h.last_extend_stats()
print(f"ragged list: {nq} queries, {len(ext)} windows ({st[1]} slots), {cells/1e9:.1f} Gcells, {int((r[1]['n_ops']>0).sum())} survivors: "
      f"{best*1e3:.1f} ms = {cells/best/1e9:.0f} GCUPS; executed cells {st[3]/1e9:.1f} G -> padding {100*(1-st[2]/st[3]):.1f} %  [{h.last_kernel_name()}]")

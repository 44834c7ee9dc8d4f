"""lx_extend_batch / lx_extend_batch_rle on the headline batch with resident subjects (development aid)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
q, s, ext = synth.make_batch_np(nq, 150, 32, seed=0x1A3BDA02)
cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
h.set_subjects(s)
r = h.extend_batch(q, None, ext, 91, copy_ops=False)
keep = r[:3]
for rle in (False, True):
    best = 1e9
    for rep in range(reps):
        t0 = time.perf_counter(); r = h.extend_batch(q, None, ext, 91, copy_ops=False, rle=rle, out=keep); best = min(best, time.perf_counter() - t0)
    print(f"lx_extend_batch{'_rle' if rle else ''}, resident subjects: {len(ext)} ext in {best*1e3:.1f} ms = {cells/best/1e9:.0f} GCUPS of pass-1 cells")

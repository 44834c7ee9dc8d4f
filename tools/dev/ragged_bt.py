"""Development aid: where the backtrace launches of a ragged lx_extend_batch call spend their time -- the same list with and
without merged windows / with everything homologous, per-phase HIP-event times of the LAST chunk and the call's wall time."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
for label, kw in (("default (10 % merged)", {}), ("no merged windows", {"merged_frac": 0.0}), ("30 % merged", {"merged_frac": 0.3}),
                  ("queries 105-152 only", {"lq_range": (105, 152)}), ("queries 300-400 only", {"lq_range": (300, 400)})):
    q, s, ext = synth.make_ragged_lists_np(50000, seed=0x1A3BDA07, **kw)
    cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
    h.set_subjects(s)
    r = h.extend_batch(q, None, ext, 91, copy_ops=False)
    keep = r[:3]
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); r = h.extend_batch(q, None, ext, 91, copy_ops=False, out=keep); best = min(best, time.perf_counter() - t0)
    st = h.last_extend_stats()
    ph = [h.last_phase_ms(k) for k in range(4)]
    print(f"{label:26s} {len(ext):7d} windows {int((r[1]['n_ops']>0).sum()):7d} survivors {best*1e3:6.1f} ms {cells/best/1e9:6.0f} GCUPS  padding {100*(1-st[2]/st[3]):4.1f} %  last chunk: "
          f"sweep {ph[0][0]:.2f} ms, backtrace {ph[3][0]:.2f} ms")

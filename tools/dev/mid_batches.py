"""Development aid: where a mid-size lx_extend_batch call (300 ... 30 000 queries x 32 windows) spends its time; run with
LX_HOST_TIMING=1 to get the library's own breakdown of the last repetition."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

entry = sys.argv[1] if len(sys.argv) > 1 else "rle"
sizes = [int(x) for x in sys.argv[2:]] or [300, 1000, 3000, 10000, 30000]
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
for nq in sizes:
    q, s, ext = synth.make_batch_np(nq, 150, 32, seed=0x1A3BDA02)
    cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
    h.set_subjects(s)
    ts = []
    if entry == "list":
        for _ in range(8):
            t0 = time.perf_counter()
            h.extend_batch_list(q, None, ext, 91)
            ts.append(time.perf_counter() - t0)
    else:
        r = h.extend_batch(q, None, ext, 91, copy_ops=False, rle=(entry == "rle"))
        keep = r[:3]
        for _ in range(8):
            t0 = time.perf_counter()
            h.extend_batch(q, None, ext, 91, copy_ops=False, rle=(entry == "rle"), out=keep)
            ts.append(time.perf_counter() - t0)
    print(f"== {nq} queries, {len(ext)} extensions, entry {entry}: best {min(ts)*1e3:.3f} ms, median {sorted(ts)[4]*1e3:.3f} ms = {cells/sorted(ts)[4]/1e9:.0f} GCUPS", flush=True)

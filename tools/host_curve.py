"""Latency of the host-buffer entry point over the batch size: lx_extend_batch / lx_extend_batch_rle on
N queries x 32 windows of the headline shape (150 aa x 176, BLOSUM62, cut-off 91), subjects resident.  One JSON line per N:
best and median wall time of the C call through ctypes, GCUPS of pass-1 cells, and the ratio to the CPU baseline bench.py
prints (oracle SIMD port on the box's granted cores) -- the number INTEGRATION.md quotes for `maximumQueryBlockSize`
(/root/reference/src/search_options.hpp:71: the reference hands the seam <= 10 queries per thread batch)."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lambda_amd import capi, synth

cpu_gcups = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
for nq in (1, 10, 30, 100, 300, 1000, 3000, 10000, 30000, 100000):
    q, s, ext = synth.make_batch_np(nq, 150, 32, seed=0x1A3BDA02)
    cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
    h.set_subjects(s)
    line = {"queries": nq, "extensions": len(ext), "gcells": round(cells / 1e9, 5)}
    for rle in (False, True):
        r = h.extend_batch(q, None, ext, 91, copy_ops=False, rle=rle)
        keep = r[:3]
        reps = 30 if nq <= 1000 else 8 if nq <= 10000 else 4
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            h.extend_batch(q, None, ext, 91, copy_ops=False, rle=rle, out=keep)
            ts.append(time.perf_counter() - t0)
        best, med = min(ts), sorted(ts)[len(ts) // 2]
        key = "rle" if rle else "bytes"
        line[f"ms_best_{key}"] = round(best * 1e3, 3)
        line[f"ms_median_{key}"] = round(med * 1e3, 3)
        line[f"gcups_{key}"] = round(cells / med / 1e9, 1)
    line["cpu_baseline_gcups"] = cpu_gcups
    line["cpu_ms_at_baseline"] = round(cells / cpu_gcups / 1e6, 3)
    line["gpu_over_cpu_rle"] = round(line["gcups_rle"] / cpu_gcups, 2)
    print(json.dumps(line), flush=True)

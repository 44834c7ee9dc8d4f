"""First call against later calls of lx_iterate_matches_dev on bench.py --iterate's list: fresh handles in one process (the first one
also pays the code objects' load), LX_HOST_TIMING=1 prints where each call's time goes.  Development aid."""
import sys, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import torch
from lambda_amd import capi, synth, workloads

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reserve = len(sys.argv) > 2 and sys.argv[2] == "reserve"
w = workloads.WORKLOADS[2]
d = w.directions[0]
m_, ma, mi, go, ge = d.scoring
ka = capi.karlin_params(*w.karlin)
q, qoff, qlen, qorig, s, soff, slen, m = synth.make_seed_list_np(reads, 100.0, seed=0x1A3BDA03)
params = capi.SearchParams(w.max_evalue, -1, 0, int(slen.sum()), 0, 2, 1, 0, capi.LX_FRAMES_REVCOMP, capi.LX_FRAMES_NONE, ka)
dev = torch.device("cuda:0")
d_m = torch.from_numpy(m.view(np.uint8).copy()).to(dev)
torch.cuda.synchronize()
lib = capi.load()
for hnum in range(3):
    t0 = time.perf_counter()
    h = capi.Handle(0)
    h.set_scoring(capi.builtin_scoring(m_, match=ma, mismatch=mi, gap_open=go, gap_extend=ge), d.slot)
    h.set_subjects(s)
    h.set_subject_seqs(soff, slen)
    h.set_queries(q, qoff, qlen, qorig, 2)
    t1 = time.perf_counter()
    if reserve and hasattr(lib, "lx_reserve"):
        h.reserve(len(m), len(m) // 7, len(m) // 12, (len(m) // 12) * 160)
    t2 = time.perf_counter()
    line = f"handle {hnum}: create + sets {1e3 * (t1 - t0):.1f} ms, reserve {1e3 * (t2 - t1):.1f} ms; calls"
    for call in range(4):
        r = C.c_void_p()
        ta = time.perf_counter()
        h._check(lib.lx_iterate_matches_dev(h.h, 0, d_m.data_ptr(), len(m), C.byref(params), C.byref(r)))
        tb = time.perf_counter()
        n = int(lib.lx_iterate_result_count(r))
        lib.lx_iterate_result_free(r)
        line += f" {1e3 * (tb - ta):.1f}"
    print(line + f" ms ({n} HSPs)", flush=True)
    h.close() if hasattr(h, "close") else None

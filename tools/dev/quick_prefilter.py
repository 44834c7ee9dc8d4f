"""Throughput probe of lx_prefilter_batch (seedLooksPromising on the GPU).  Development aid."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
rng = np.random.default_rng(1)
nq, lq, ns, ls = 100_000, 150, 20_000, 400
q = synth.STD20[rng.integers(0, 20, nq * lq)].astype(np.uint8)
s = synth.STD20[rng.integers(0, 20, ns * ls)].astype(np.uint8)
seeds = np.zeros(n, dtype=capi.SEED_DTYPE)
qi = rng.integers(0, nq, n); si = rng.integers(0, ns, n)
seeds["q_off"] = qi * lq; seeds["s_off"] = si * ls; seeds["q_len"] = lq; seeds["s_len"] = ls
qs = rng.integers(0, lq - 10, n); ss = rng.integers(0, ls - 10, n)
seeds["qry_start"] = qs; seeds["qry_end"] = qs + 10; seeds["subj_start"] = ss
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
h.set_subjects(s)
for rep in range(3):
    t0 = time.perf_counter(); keep = h.prefilter_batch(q, None, seeds, 10, 2, 2.0); dt = time.perf_counter() - t0
    print(f"prefilter: {n} seeds, kernel {h.last_kernel_ms():.3f} ms = {n/h.last_kernel_ms()/1e6:.2f} G seeds/s; call {dt*1e3:.1f} ms; kept {int(keep.sum())}")

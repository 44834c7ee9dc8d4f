"""How the cpu_baseline leg scales with threads on this box (development aid for DESIGN.md section 6)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lambda_amd import capi, synth
from tests import oracle_lib

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
orc = oracle_lib.load()
q, s, ext = synth.make_batch_np(20000, 150, 32, seed=0x1A3BDA02)
sc = oracle_lib.scoring_from(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1))
cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
for t in (1, 8, 16, 32, 64, 128, 256):
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); orc.score_batch(q, s, ext, sc, threads=t, simd=True); best = min(best, time.perf_counter() - t0)
    print(f"threads {t:4d}: {cells/best/1e9:8.1f} GCUPS  ({best*1e3:.0f} ms)")

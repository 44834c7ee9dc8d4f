"""lx_score_batch (host buffers) on queries wider than the packed-half geometries: packed 16-bit integer kernel vs int32,
checked against the oracle.  Development aid."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth
from tests import oracle_lib
orc = oracle_lib.load()
h = capi.Handle(0)
sc = capi.builtin_scoring(62, gap_open=-11, gap_extend=-1)
h.set_scoring(sc, 0)
for lq, nq in ((300, 3000), (650, 1500)):
    q, s, ext = synth.make_batch_np(nq, lq, 32, seed=lq)
    cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
    for packed in (1, 0):
        h.set_option(capi.LX_OPT_PACKED_HALF, packed)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); got = h.score_batch(q, s, ext); best = min(best, time.perf_counter() - t0)
        print(lq, "packed", packed, h.last_kernel_name()[:70], f"{cells/1e9:.1f} Gcells kernel {h.last_kernel_ms():.2f} ms call {best*1e3:.1f} ms")
    want = orc.score_batch(q, s, ext[:4000], oracle_lib.scoring_from(sc), threads=8, simd=False)
    assert (got[:4000] == want).all()
print("ok")

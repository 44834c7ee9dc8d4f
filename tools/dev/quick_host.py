"""PCIe-inclusive rate of the host-buffer entry points (lx_score_batch / lx_align_batch) on the headline batch, with the
subject windows uploaded per call and with resident subjects.  Development aid for DESIGN.md section 5."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lq, wpq = 150, 32
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
q, s, ext = synth.make_batch_np(nq, lq, wpq, seed=0x1A3BDA02)
cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
for resident in (False, True):
    if resident:
        h.set_subjects(s)
    sarg = None if resident else s
    tag = "resident subjects" if resident else "subjects uploaded per call"
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); sc = h.score_batch(q, sarg, ext); best = min(best, time.perf_counter() - t0)
    print(f"lx_score_batch, {tag}: {len(ext)} ext, {cells/1e9:.1f} Gcells in {best*1e3:.1f} ms = {cells/best/1e9:.0f} GCUPS (kernel {h.last_kernel_ms():.2f} ms)")
    keep = np.nonzero(sc >= 91)[0]
    es = ext[keep]
    c2 = float((es["q_len"].astype(np.float64) * es["s_len"]).sum())
    best2 = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); hsp, ops, _ = h.align_batch(q, sarg, es, known_score=sc[keep], raw=True); best2 = min(best2, time.perf_counter() - t0)
    print(f"lx_align_batch, {tag}: {len(es)} ext, {c2/1e9:.1f} Gcells in {best2*1e3:.1f} ms = {c2/best2/1e9:.0f} GCUPS (kernel {h.last_kernel_ms():.2f} ms)")
    print(f"  both calls: {cells/(best+best2)/1e9:.0f} GCUPS of pass-1 cells, PCIe and host work included")
    best3 = 1e9
    r = h.extend_batch(q, sarg, ext, 91, copy_ops=False)
    keep = r[:3]  # the caller keeps its result arrays between calls, like lambda's per-thread holders
    for rep in range(4):
        t0 = time.perf_counter(); r = h.extend_batch(q, sarg, ext, 91, copy_ops=False, out=keep); best3 = min(best3, time.perf_counter() - t0)
    print(f"lx_extend_batch, {tag}: {len(ext)} ext in {best3*1e3:.1f} ms = {cells/best3/1e9:.0f} GCUPS of pass-1 cells (survivors {int((r[1]['n_ops']>0).sum())})")
    best4 = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); r = h.extend_batch(q, sarg, ext, 91, copy_ops=False, rle=True, out=keep); best4 = min(best4, time.perf_counter() - t0)
    print(f"lx_extend_batch_rle, {tag}: {len(ext)} ext in {best4*1e3:.1f} ms = {cells/best4/1e9:.0f} GCUPS of pass-1 cells ({len(r[3])} code bytes)")

"""lx_iterate_matches (the whole of iterateMatchesFullSimd, /root/reference/src/search_algo.hpp:1177-1332) on a seed list of a
realistic size: how much of the call is the driver's host work (widen + merge, slices, cut-offs, records) and how much the
extension.  Development aid; LX_HOST_TIMING=1 prints the extension's own breakdown."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
hits_per_q = 12
rng = np.random.default_rng(7)
ns = 2000
qlen = rng.integers(50, 400, nq).astype(np.uint64)
slen = rng.integers(600, 2500, ns).astype(np.uint64)
qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint64)
soff = np.concatenate([[0], np.cumsum(slen)[:-1]]).astype(np.uint64)
q = synth.STD20[rng.integers(0, 20, int(qlen.sum()))].astype(np.uint8)
s = synth.STD20[rng.integers(0, 20, int(slen.sum()))].astype(np.uint8)
nh = nq * hits_per_q
a = np.repeat(np.arange(nq), hits_per_q)
b = rng.integers(0, ns, nh)
L = 10
qs = (rng.random(nh) * (qlen[a] - L)).astype(np.int64)
ss = (rng.random(nh) * (slen[b] - L)).astype(np.int64)
for i in np.nonzero(rng.random(nh) < 0.5)[0]:  # homologous region around half of the seeds
    lo = min(qs[i], ss[i]); hi = min(int(qlen[a[i]]) - qs[i], int(slen[b[i]]) - ss[i])
    seg = q[int(qoff[a[i]]) + qs[i] - lo: int(qoff[a[i]]) + qs[i] + hi].copy()
    mut = rng.random(len(seg)) < 0.25
    seg[mut] = synth.STD20[rng.integers(0, 20, int(mut.sum()))]
    s[int(soff[b[i]]) + ss[i] - lo: int(soff[b[i]]) + ss[i] + hi] = seg
m = np.zeros(nh, dtype=capi.MATCH_DTYPE)
m["qryId"], m["subjId"], m["qryStart"], m["qryEnd"], m["subjStart"], m["subjEnd"] = a, b, qs, qs + L, ss, ss + L
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62), 0)
ka = capi.karlin_params(62)
params = capi.SearchParams(1e-2, -1, 0, int(slen.sum()), 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
h.set_subjects(s)
best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    bms, ops, stats = h.iterate_matches(q, qoff, qlen, qlen, None, soff, slen, m, params)
    best = min(best, time.perf_counter() - t0)
print(f"lx_iterate_matches: {nq} queries, {nh} seeds -> {len(bms)} HSPs in {best * 1e3:.1f} ms (duplicates {stats.hits_duplicate}, failed e-value {stats.failed_evalue})")
# the same list from device memory (lx_iterate_matches_dev over resident sets)
import torch
h.set_subject_seqs(soff, slen)
h.set_queries(q, qoff, qlen, qlen, 1)
d_m = torch.from_numpy(m.view(np.uint8).copy()).to("cuda:0")
best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    bms2, ops2, stats2 = h.iterate_matches_dev(d_m, len(m), params)
    best = min(best, time.perf_counter() - t0)
print(f"lx_iterate_matches_dev: {len(bms2)} HSPs in {best * 1e3:.1f} ms; same records: {bms2.tobytes() == bms.tobytes()}")

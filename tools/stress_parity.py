"""Randomised parity stress of the fused step on host buffers (lx_extend_batch) against the CPU oracle (every third case also through
lx_extend_batch_list): random query
lengths (all sweep geometries and the fall-backs), run lengths, schemes, gap costs, mutation rates, truncated / empty
windows, cut-offs and pass-2 modes.  Development aid: `python tools/stress_parity.py SECONDS [SEED [MAX_QUERY_LENGTH [MIN_QUERY_LENGTH]]]` on a GPU box; the
committed parity tests are tests/test_gpu_*.py."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lambda_amd import capi, synth
from tests import oracle_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_lq = int(sys.argv[3]) if len(sys.argv) > 3 else 215  # beyond 208 columns: multi-panel kernels, direction bits
min_lq = int(sys.argv[4]) if len(sys.argv) > 4 else 20
orc = oracle_lib.load()
h = capi.Handle(0)
t0 = time.time()
case = bad = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 * 100003 + case)
    kind = rng.integers(0, 4)
    if kind == 0:
        go, ge = int(rng.integers(-14, -5)), int(rng.integers(-3, 0))
        sc_p, alpha = capi.builtin_scoring(62, gap_open=go, gap_extend=ge), synth.STD20
    elif kind == 1:
        sc_p, alpha = capi.builtin_scoring(0, match=int(rng.integers(1, 4)), mismatch=int(rng.integers(-4, -1)),
                                           gap_open=int(rng.integers(-6, -1)), gap_extend=int(rng.integers(-3, 0))), np.arange(4, dtype=np.uint8)
    elif kind == 2:
        sc_p, alpha = capi.builtin_scoring(62, gap_open=int(rng.integers(-40, -20)), gap_extend=int(rng.integers(-4, 0))), synth.STD20
    else:
        sc_p, alpha = capi.builtin_scoring(62, gap_open=0, gap_extend=int(rng.integers(-6, -1))), synth.STD20[:int(rng.integers(2, 6))]
    osc = oracle_lib.scoring_from(sc_p)
    h.set_scoring(sc_p, 0)
    lq = int(rng.integers(min_lq, max_lq))
    wpq = int(rng.choice([5, 8, 16, 24, 32, 40]))
    nq = int(rng.integers(4, 24))
    if rng.random() < 0.4:
        # a list as lambda hands it over: mixed query lengths, a few windows per query, merged windows (the multi-query plan)
        q, s, ext = synth.make_ragged_lists_np(int(rng.integers(6, 40)), seed=int(rng.integers(1, 1 << 30)), alphabet=alpha,
                                               lq_range=(min_lq, max(min_lq + 1, max_lq)), mean_windows=float(rng.uniform(1.5, 9.0)),
                                               merged_frac=float(rng.uniform(0.0, 0.3)), sub_rate=float(rng.uniform(0.0, 0.4)))
    else:
        q, s, ext = synth.make_batch_np(nq, lq, wpq, seed=int(rng.integers(1, 1 << 30)), alphabet=alpha,
                                        sub_rate=float(rng.uniform(0.0, 0.4)), indel_rate=float(rng.uniform(0.0, 0.08)))
    ext = ext.copy()
    cut = rng.random(len(ext))
    full = ext["s_len"].copy()
    ext["s_len"] = np.where(cut < 0.03, 0, np.where(cut < 0.1, rng.integers(1, 10, len(ext)),
                            np.where(cut < 0.4, (full * rng.uniform(0.3, 1.0, len(ext))).astype(np.uint32), full))).astype(np.uint32)
    ext["s_len"] = np.minimum(ext["s_len"], full)
    if rng.random() < 0.3:
        ext = ext[rng.permutation(len(ext))]
    want = orc.score_batch(q, s, ext, osc, threads=8)
    cutoff = int(np.percentile(want, rng.uniform(20, 80))) + int(rng.integers(0, 2))
    mode = int(rng.choice([2, 2, 2, 1, 0]))
    h.set_option(capi.LX_OPT_PASS2_MODE, mode)
    score, hsp, off, ops = h.extend_batch(q, s, ext, cutoff)
    name = h.last_trace_kernel_name()
    ok = (score == want).all()
    surv = np.nonzero((want >= cutoff) & (ext["s_len"] > 0))[0]
    if len(surv) > 300:
        surv = rng.choice(surv, 300, replace=False)
    if ok:
        for i, (oh, oops) in zip(surv, orc.align_batch(q, s, ext[surv], osc)):
            g = hsp[i]
            if (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) != \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops) or \
               bytes(ops[int(off[i]) + int(g["ops_shift"]): int(off[i]) + int(g["ops_shift"]) + oh.n_ops]) != oops:
                ok = False
                print("ALIGN MISMATCH", i, ext[i], tuple(g), (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops))
                break
    if ok and case % 3 == 0:
        # the list form of the same call: the same scores, exactly the survivors, the same records, codes that expand to the same ops
        score_l, index, hsp_l, off_l, codes = h.extend_batch_list(q, s, ext, cutoff)
        live = np.nonzero((want >= cutoff) & (ext["s_len"] > 0) & (ext["q_len"] > 0))[0]
        ok = (score_l == want).all() and len(index) == len(live) and (np.sort(index) == live).all()
        for k in (range(len(index)) if ok else ()):
            i, g = int(index[k]), hsp[int(index[k])]
            same = all(hsp_l[f][k] == g[f] for f in ("score", "q_begin", "q_end", "s_begin", "s_end", "n_ops", "num_matches", "num_gap_opens"))
            st = int(off[i]) + int(g["ops_shift"])
            if not same or capi.Handle.expand_ops(codes[int(off_l[k]): int(off_l[k]) + int(g["n_ops"])], int(g["n_ops"])) != bytes(ops[st: st + int(g["n_ops"])]):
                ok = False
                print("LIST MISMATCH", i, ext[i], tuple(g), tuple(hsp_l[k]))
                break
    if not ok:
        bad += 1
        print(f"FAIL case {case} seed0 {seed0}: kind {kind} lq {lq} wpq {wpq} nq {nq} mode {mode} go {sc_p.gap_open} ge {sc_p.gap_extend} cutoff {cutoff} kernel {name}")
    case += 1
print(f"{case} cases, {bad} failures, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

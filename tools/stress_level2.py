"""Randomised parity stress of the Level-2 driver on DEVICE match lists (lx_widen_and_preprocess_dev, lx_iterate_matches_dev):
  (a) widen + sort + merge + unique against the CPU oracle's _widenAndPreprocessMatches restatement -- list sizes on and around the
      tile sizes of the sort / scan kernels, one to many queries and subjects, subjects shorter than the query's window and longer
      than 2^32, heavy duplication, every arrival order, the bisulfite order;
  (c) the free-packing plan of protein window lists made on the device (lx_plan_free_packing_dev) slot by slot -- every window once, fillers
      copies of their wavefront's windows, a lane group one query, at most four queries per wavefront, the wavefronts' widest query and longest
      window, ranges contiguous (tests/test_gpu_plan.py's checker) -- for random run lengths, query lengths, merged / clipped shares, strip
      widths and range cuts;
  (b) the whole driver call against lx_iterate_matches on the same list from host memory (below 131 072 matches that is the host's own
      list code: an independent implementation), byte for byte -- random schemes, filters, orders, LX_ITERATE_NO_OPS; small cases
      against the oracle driver (tests/oracle_driver.py) as well.
Development aid: `python tools/stress_level2.py SECONDS [SEED]` on a GPU box; the committed parity tests are tests/test_gpu_level2.py."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lambda_amd import capi
from tests import oracle_driver, oracle_lib
from tests.test_gpu_level2 import _random_matches, _seed_list, _to_device
from tests.test_gpu_plan import check_plan, make_list
from tests.test_oracle import SCHEMES

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = oracle_lib.load()
h = capi.Handle(0)
EDGES = [1, 2, 3, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 65536, 65537]
FIELDS = ("qryId", "subjId", "qryStart", "qryEnd", "subjStart", "subjEnd")
t0 = time.time()
case = bad = n_widen = n_driver = n_oracle_driver = n_plan = 0


def plan_case(rng):
    nq = int(rng.choice([1, 2, 5, 60, 700, 4000]))
    mean_w = float(rng.choice([1.2, 3, 8, 30, 200])) if nq < 4000 else float(rng.choice([1.2, 3, 8]))
    lq_hi = int(rng.choice([60, 150, 400, 1300]))
    ext = make_list(rng, nq, lambda r: 1 + r.poisson(mean_w) if r.random() < 0.9 else 1 + r.poisson(20 * mean_w), lambda r: int(r.integers(20, lq_hi + 1)),
                    merged_share=float(rng.choice([0, 0.1, 0.5])), clipped_share=float(rng.choice([0, 0.1, 0.4])))
    n = len(ext)
    change = np.nonzero(ext["q_off"][1:] != ext["q_off"][:-1])[0] + 1
    R = int(rng.integers(1, 9))
    cuts = [0] + sorted({int(change[i]) for i in rng.integers(0, len(change), R - 1)} if len(change) and R > 1 else set()) + [n]
    C = int(rng.choice([19, 13, 11]))
    plan, pan, maxs, rep = h.plan_free_packing_dev(_to_device(ext), n, nq, strip_cols=C, cuts=cuts)
    try:
        check_plan(ext, plan, pan, maxs, rep, cuts, C)
        ok = True
    except AssertionError as e:
        ok = False
        print("   ", repr(e)[:300])
    return ok, f"plan n={n} nq={nq} mean windows {mean_w} lq <= {lq_hi} C={C} ranges {len(cuts) - 1} -> {len(plan)} wavefronts"


def widen_case(rng):
    nq = int(rng.choice([1, 2, 7, 300, 5000]))
    ns = int(rng.choice([1, 2, 40, 3000]))
    n = int(rng.choice(EDGES)) if rng.random() < 0.6 else int(rng.integers(1, 300_000))
    kind = rng.integers(0, 4)
    qlens = rng.integers(10, int(rng.choice([40, 200, 1000])), nq).astype(np.uint64)
    if kind == 0:    # subjects shorter than most windows: both ends clipped
        slens = rng.integers(12, 120, ns).astype(np.uint64)
    elif kind == 1:  # positions beyond 2^32 (the sort takes the bits the call can have set)
        slens = rng.integers(5_000_000_000, 9_000_000_000, ns).astype(np.uint64)
    else:
        slens = rng.integers(300, 30_000, ns).astype(np.uint64)
    step = int(rng.choice([1, 3, 7, 50, 1000]))
    m = _random_matches(rng, n, nq, ns, qlens, slens, sorted_by_query=bool(rng.integers(0, 2)), step=step)
    if rng.random() < 0.2:  # one key many times over
        m[:] = m[0]
    qoff = np.concatenate([[0], np.cumsum(qlens)[:-1]]).astype(np.uint64)
    soff = np.concatenate([[0], np.cumsum(slens)[:-1]]).astype(np.uint64)
    h.set_queries(np.zeros(int(qlens.sum()), np.uint8), qoff, qlens)
    h.set_subject_seqs(soff, slens)
    order = [np.arange(n), rng.permutation(n), np.arange(n)[::-1]][int(rng.integers(0, 3))]
    bis = rng.random() < 0.25
    got = h.widen_and_preprocess_dev(_to_device(m[order]), n, bisulfite=bis)
    if bis:
        want = np.concatenate([orc.widen_and_preprocess(m[m["subjId"] % 2 == k].astype(oracle_lib.MATCH_DTYPE), qlens, slens) for k in (0, 1)])
    else:
        want = orc.widen_and_preprocess(m.astype(oracle_lib.MATCH_DTYPE), qlens, slens)
    ok = len(got) == len(want) and all((got[f] == want[f]).all() for f in FIELDS)
    return ok, f"widen n={n} nq={nq} ns={ns} kind={kind} step={step} bis={bis} -> {len(want)} windows"


def driver_case(rng):
    global n_oracle_driver
    dna = rng.random() < 0.4
    sc_p = SCHEMES["nucl" if dna else "blosum62"]
    h.set_scoring(sc_p, 0)
    small = rng.random() < 0.25
    nq = int(rng.integers(3, 40)) if small else int(rng.integers(50, 4000))
    ns = int(rng.integers(1, 8)) if small else int(rng.integers(5, 600))
    hits = int(rng.integers(1, 6)) if small else int(rng.integers(2, 14))
    lo = int(rng.choice([30, 50, 140]))
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, nq, ns, hits, lq_range=(lo, lo + int(rng.choice([20, 100, 350]))),
                                                 alphabet=np.arange(4, dtype=np.uint8) if dna else None)
    ka = capi.karlin_params(0, 2, -3, -5, -2) if dna else capi.karlin_params(62)
    max_e, min_bits, idcut = [(1e-2, -1, 0), (-1.0, 40, 0), (10.0, -1, 35), (-1.0, -1, 0), (1e-6, 30, 20)][int(rng.integers(0, 5))]
    db_total = int(slen.sum()) * int(rng.choice([1, 50]))
    flags = capi.LX_ITERATE_NO_OPS if rng.random() < 0.2 else 0
    params = capi.SearchParams(max_e, min_bits, idcut, db_total, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka, 0, flags)
    n = len(m)
    perm = [np.arange(n), rng.permutation(n), np.arange(n)[::-1]][int(rng.integers(0, 3))]
    h.set_subjects(s)
    h.set_subject_seqs(soff, slen)
    h.set_queries(q, qoff, qlen, qlen, 1)
    db, do, ds = h.iterate_matches_dev(_to_device(m[perm]), n, params)
    hb, ho, hs = h.iterate_matches(q, qoff, qlen, qlen, None, soff, slen, m[perm].copy(), params)
    ok = len(db) == len(hb) and db.tobytes() == hb.tobytes() and do == ho and \
        all(getattr(ds, f) == getattr(hs, f) for f in ("hits_duplicate", "failed_bitscore", "failed_evalue", "failed_identity", "num_ext_score", "num_ext_ali"))
    what = f"driver {'nucl' if dna else 'prot'} n={n} nq={nq} ns={ns} filters={(max_e, min_bits, idcut)} flags={flags} -> {len(db)} HSPs"
    if ok and small and not flags:
        n_oracle_driver += 1
        osc = oracle_lib.scoring_from(sc_p)
        oka = oracle_lib.Karlin(ka.lambda_, ka.K, ka.H, ka.alpha, ka.beta)
        want, wstats = oracle_driver.iterate_matches(orc, osc, oka, q, qoff, qlen, qlen, s, soff, slen, m.astype(oracle_lib.MATCH_DTYPE),
                                                     max_e, min_bits, idcut, db_total)
        ok = len(want) == len(db) and ds.hits_duplicate == wstats["hits_duplicate"]
        for g, w, o in zip(db, want, do) if ok else ():
            if any(int(g[k]) != w[k] for k in ("qry_id", "subj_id", "q_start", "q_end", "s_start", "s_end", "score", "alignment_length", "num_matches",
                                               "num_mismatches", "num_positives", "num_gap_opens", "num_gap_extensions")) or o != w["ops"]:
                ok = False
                break
        what += " (+ oracle driver)"
    return ok, what


while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 * 1000003 + case)
    if case % 3 == 2:
        ok, what = driver_case(rng)
        n_driver += 1
    elif case % 3 == 1 and case % 2 == 0:
        ok, what = plan_case(rng)
        n_plan += 1
    else:
        ok, what = widen_case(rng)
        n_widen += 1
    if not ok:
        bad += 1
        print("MISMATCH case", case, "seed", seed0, what, flush=True)
    case += 1
print(f"stress_level2: {case} cases ({n_widen} window lists against the oracle, {n_plan} device plans slot by slot, {n_driver} driver calls against the host entry point, "
      f"{n_oracle_driver} of them against the oracle driver too), {bad} failures, seed {seed0}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

"""GPU parity of the Level-2 driver on a DEVICE match list (lambda_amd/csrc/lx_level2.hip + lx_level2_host.cpp): widen, sort, merge,
unique (/root/reference/src/search_algo.hpp:919-938, :1136-1175), the slices and cut-offs (:1200-1227, :1251-1283) as kernels over
resident sequence sets, then the two passes and the records (:1287-1325).  Checked against the CPU oracle's restatement
(oracle/lx_oracle.c through tests/oracle_lib.py, tests/oracle_driver.py) and, record for record, against lx_iterate_matches on the
same list from host memory."""
import numpy as np
import pytest

from lambda_amd import capi, synth
from tests import oracle_driver, oracle_lib
from tests.test_gpu_trace import _driver_case
from tests.test_oracle import SCHEMES

pytestmark = pytest.mark.gpu


def _to_device(m):
    import torch

    return torch.from_numpy(np.ascontiguousarray(m).view(np.uint8).copy()).to("cuda:0")


def _random_matches(rng, n, nq, ns, qlens, slens, sorted_by_query=True, step=7):
    m = np.zeros(n, dtype=capi.MATCH_DTYPE)
    q = rng.integers(0, nq, n)
    m["qryId"] = np.sort(q) if sorted_by_query else q
    m["subjId"] = rng.integers(0, ns, n)
    sl = slens[m["subjId"]].astype(np.int64)
    ql = qlens[m["qryId"]].astype(np.int64)
    m["qryStart"] = (rng.random(n) * np.maximum(ql - 10, 1)).astype(np.uint64)
    m["qryEnd"] = np.minimum(m["qryStart"] + 10, ql.astype(np.uint64))
    m["subjStart"] = ((rng.random(n) * np.maximum(sl - 10, 1)).astype(np.int64) // step * step).astype(np.uint64)  # duplicates and overlaps abound
    m["subjEnd"] = np.minimum(m["subjStart"] + 10, sl.astype(np.uint64))
    return m


@pytest.mark.parametrize("shape", ["few subjects", "many subjects", "tiny", "long subjects"])
def test_widen_sort_merge_unique_on_device_equals_oracle(handle, oracle, shape):
    """The window list of the kernels is the oracle's _widenAndPreprocessMatches list, element for element, whatever order the
    matches arrive in (the seeding kernel's lanes emit them in theirs)."""
    rng = np.random.default_rng(5)
    nq, ns, n, step = {"few subjects": (9000, 40, 150_000, 7), "many subjects": (300, 50_000, 120_000, 1), "tiny": (3, 2, 200, 3),
                       "long subjects": (5000, 3, 100_000, 1)}[shape]
    qlens = rng.integers(30, 200, nq).astype(np.uint64)
    slens = (rng.integers(300, 2000, ns) if shape != "long subjects" else rng.integers(3_000_000, 5_000_000, ns)).astype(np.uint64)
    m = _random_matches(rng, n, nq, ns, qlens, slens, step=step)
    qoff = np.concatenate([[0], np.cumsum(qlens)[:-1]]).astype(np.uint64)
    soff = np.concatenate([[0], np.cumsum(slens)[:-1]]).astype(np.uint64)
    handle.set_queries(np.zeros(int(qlens.sum()), np.uint8), qoff, qlens)
    handle.set_subject_seqs(soff, slens)
    want = oracle.widen_and_preprocess(m.astype(oracle_lib.MATCH_DTYPE), qlens, slens)
    assert len(want) < n
    for order in (np.arange(n), rng.permutation(n), np.arange(n)[::-1]):
        got = handle.widen_and_preprocess_dev(_to_device(m[order]), n)
        assert len(got) == len(want)
        for f in ("qryId", "subjId", "qryStart", "qryEnd", "subjStart", "subjEnd"):
            assert (got[f] == want[f]).all(), (shape, f)
    # bisulfite order: subjId % 2 is the major key (src/search_algo.hpp:1369-1372), each half the oracle's list of its matches
    got = handle.widen_and_preprocess_dev(_to_device(m), n, bisulfite=True)
    halves = [oracle.widen_and_preprocess(m[m["subjId"] % 2 == k].astype(oracle_lib.MATCH_DTYPE), qlens, slens) for k in (0, 1)]
    want_bs = np.concatenate(halves)
    assert len(got) == len(want_bs)
    for f in ("qryId", "subjId", "qryStart", "qryEnd", "subjStart", "subjEnd"):
        assert (got[f] == want_bs[f]).all(), (shape, f)


def test_widen_on_device_clips_and_chains_like_the_oracle(handle, oracle):
    """Hand-made edge cases: windows clipped at both subject ends, a chain of overlaps that merges transitively, touching windows
    (subjEnd == subjStart merges: `>=`, :1150), exact duplicates, one match, queries longer than the subject."""
    qlens = np.array([50, 120, 400, 9], dtype=np.uint64)
    slens = np.array([300, 60, 5000], dtype=np.uint64)
    rows = []
    for q in range(4):
        for s in range(3):
            L = int(slens[s])
            for ss in (0, 1, 5, L // 2, L // 2, L // 2 + 1, min(L // 2 + 40, L - 2), L - 10, L - 1):
                for qs in (0, 3, int(qlens[q]) - 1):
                    rows.append((q, s, qs, min(qs + 10, int(qlens[q])), ss, min(ss + 10, L)))
    # a chain: every window overlaps the next one only
    for k in range(25):
        rows.append((1, 2, 0, 10, 100 + 130 * k, 110 + 130 * k))
    # touching: band(120) = 11 -> window [ss - 11, ss + 131); the next one starts exactly where this one ends
    rows.append((1, 2, 0, 10, 4000, 4010))
    rows.append((1, 2, 0, 10, 4000 + 142, 4010 + 142))
    m = np.array(rows, dtype=capi.MATCH_DTYPE)
    qoff = np.concatenate([[0], np.cumsum(qlens)[:-1]]).astype(np.uint64)
    soff = np.concatenate([[0], np.cumsum(slens)[:-1]]).astype(np.uint64)
    handle.set_queries(np.zeros(int(qlens.sum()), np.uint8), qoff, qlens)
    handle.set_subject_seqs(soff, slens)
    want = oracle.widen_and_preprocess(m.astype(oracle_lib.MATCH_DTYPE), qlens, slens)
    got = handle.widen_and_preprocess_dev(_to_device(m), len(m))
    assert len(got) == len(want)
    assert (got.view(np.uint64) == want.view(np.uint64)).all()
    one = handle.widen_and_preprocess_dev(_to_device(m[:1]), 1)
    assert (one.view(np.uint64) == oracle.widen_and_preprocess(m[:1].astype(oracle_lib.MATCH_DTYPE), qlens, slens).view(np.uint64)).all()
    assert len(handle.widen_and_preprocess_dev(None, 0)) == 0


def test_device_list_errors_are_loud(handle):
    qlens = np.array([50, 60], dtype=np.uint64)
    slens = np.array([300, 200], dtype=np.uint64)
    handle.set_queries(np.zeros(110, np.uint8), np.array([0, 50], np.uint64), qlens)
    handle.set_subject_seqs(np.array([0, 300], np.uint64), slens)
    ok = np.array([(0, 1, 0, 10, 20, 30)], dtype=capi.MATCH_DTYPE)
    assert len(handle.widen_and_preprocess_dev(_to_device(ok), 1)) == 1
    for bad in ((2, 0, 0, 10, 20, 30), (0, 2, 0, 10, 20, 30), (0, 1, 0, 10, 200, 210), (0, 1, 5, 15, 400, 410)):
        with pytest.raises(capi.LambdaExtError):
            handle.widen_and_preprocess_dev(_to_device(np.array([ok[0], bad], dtype=capi.MATCH_DTYPE)), 2)


def test_device_list_empty_single_and_after_an_error(handle, oracle):
    """Edge cases of the device entry points: an empty list, one match, one match many times over, and a good call right after a
    refused one (a failed call leaves nothing behind that the next one trips over)."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    rng = np.random.default_rng(12)
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, 40, 5, 3)
    ka = capi.karlin_params(62)
    params = capi.SearchParams(10.0, -1, 0, int(slen.sum()), 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qlen, 1)
    import torch

    empty = torch.zeros(48, dtype=torch.uint8, device="cuda:0")
    bms, ops, stats = handle.iterate_matches_dev(empty, 0, params)
    assert len(bms) == 0 and ops == [] and stats.num_ext_score == 0 and stats.hits_duplicate == 0
    assert len(handle.widen_and_preprocess_dev(empty, 0)) == 0
    want_all, _, _ = handle.iterate_matches(q, qoff, qlen, qlen, None, soff, slen, m.copy(), params)
    for k in (1, 2, len(m)):
        one = np.repeat(m[:1], k) if k > 1 and k < len(m) else m[:k]
        db, do, ds = handle.iterate_matches_dev(_to_device(one), len(one), params)
        hb, ho, hs = handle.iterate_matches(q, qoff, qlen, qlen, None, soff, slen, one.copy(), params)
        _assert_records_equal(db, do, hb, ho)
        assert ds.hits_duplicate == hs.hits_duplicate == (k - 1 if 1 < k < len(m) else hs.hits_duplicate)
    bad = m.copy()
    bad["subjId"][len(bad) // 2] = len(slen)  # no such subject
    with pytest.raises(capi.LambdaExtError):
        handle.iterate_matches_dev(_to_device(bad), len(bad), params)
    db, do, _ = handle.iterate_matches_dev(_to_device(m), len(m), params)
    assert len(db) == len(want_all) and db.tobytes() == want_all.tobytes()
    with pytest.raises(capi.LambdaExtError):  # the device entry point has no band mode (include/lambda_ext.h)
        handle.iterate_matches_dev(_to_device(m), len(m), capi.SearchParams(10.0, -1, 0, int(slen.sum()), 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka, 16))
    db, do, _ = handle.iterate_matches_dev(_to_device(m), len(m), params)
    assert db.tobytes() == want_all.tobytes()


def _assert_records_equal(a, ao, b, bo):
    assert len(a) == len(b)
    assert a.tobytes() == b.tobytes()
    assert ao == bo


@pytest.mark.parametrize("filters", [(1e-2, -1, 0), (-1.0, 40, 0), (10.0, -1, 35), (-1.0, -1, 0)])
def test_iterate_matches_dev_against_the_oracle_driver(handle, oracle, filters):
    max_e, min_bits, idcut = filters
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    ka = capi.karlin_params(62)
    oka = oracle_lib.Karlin(ka.lambda_, ka.K, ka.H, ka.alpha, ka.beta)
    rng = np.random.default_rng(2024)
    q, qoff, qlen, s, soff, slen, m = _driver_case(rng)
    db_total = int(slen.sum())
    params = capi.SearchParams(max_e, min_bits, idcut, db_total, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qlen, 1)
    order = rng.permutation(len(m))
    bms, ops, stats = handle.iterate_matches_dev(_to_device(m[order]), len(m), params)
    want, wstats = oracle_driver.iterate_matches(oracle, osc, oka, q, qoff, qlen, qlen, s, soff, slen,
                                                 m.astype(oracle_lib.MATCH_DTYPE), max_e, min_bits, idcut, db_total)
    assert (stats.hits_duplicate, stats.failed_bitscore, stats.failed_evalue, stats.failed_identity) == \
           (wstats["hits_duplicate"], wstats["failed_bitscore"], wstats["failed_evalue"], wstats["failed_identity"])
    assert stats.num_ext_score == len(m)
    assert len(bms) == len(want) and len(want) > 20
    for g, w, o in zip(bms, want, ops):
        for k in ("qry_id", "subj_id", "n_qid", "n_sid", "q_start", "q_end", "s_start", "s_end", "score", "alignment_length",
                  "num_matches", "num_mismatches", "num_positives", "num_gap_opens", "num_gap_extensions"):
            assert int(g[k]) == w[k], (k, g, w)
        assert o == w["ops"]
        assert g["identity"] == np.float32(w["identity"])
        assert abs(g["bit_score"] - w["bit_score"]) <= 1e-6 * abs(w["bit_score"])  # the contract: 1e-6 relative
        assert abs(g["e_value"] - w["e_value"]) <= 1e-6 * abs(w["e_value"])
    # ... and the host entry point on the same list gives the same bytes
    hb, ho, hs = handle.iterate_matches(q, qoff, qlen, qlen, None, soff, slen, m, params)
    _assert_records_equal(bms, ops, hb, ho)


def _seed_list(rng, nq, ns, hits_per_q, lq_range=(50, 400), alphabet=None, frames=1):
    alphabet = synth.STD20 if alphabet is None else alphabet
    qlen = rng.integers(lq_range[0], lq_range[1], nq).astype(np.uint64)
    slen = rng.integers(600, 2500, ns).astype(np.uint64)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint64)
    soff = np.concatenate([[0], np.cumsum(slen)[:-1]]).astype(np.uint64)
    q = alphabet[rng.integers(0, len(alphabet), int(qlen.sum()))].astype(np.uint8)
    s = alphabet[rng.integers(0, len(alphabet), int(slen.sum()))].astype(np.uint8)
    nh = nq * hits_per_q
    a = np.repeat(np.arange(nq), hits_per_q)
    b = rng.integers(0, ns, nh)
    L = 10
    qs = (rng.random(nh) * (qlen[a] - L)).astype(np.int64)
    ss = (rng.random(nh) * (slen[b] - L)).astype(np.int64)
    for i in np.nonzero(rng.random(nh) < 0.5)[0]:  # homologous region around half of the seeds
        lo = min(qs[i], ss[i])
        hi = min(int(qlen[a[i]]) - qs[i], int(slen[b[i]]) - ss[i])
        seg = q[int(qoff[a[i]]) + qs[i] - lo: int(qoff[a[i]]) + qs[i] + hi].copy()
        mut = rng.random(len(seg)) < 0.25
        seg[mut] = alphabet[rng.integers(0, len(alphabet), int(mut.sum()))]
        s[int(soff[b[i]]) + ss[i] - lo: int(soff[b[i]]) + ss[i] + hi] = seg
    m = np.zeros(nh, dtype=capi.MATCH_DTYPE)
    m["qryId"], m["subjId"], m["qryStart"], m["qryEnd"], m["subjStart"], m["subjEnd"] = a, b, qs, qs + L, ss, ss + L
    # a second seed on most diagonals and near-by ones: duplicates and merges (:1144-1173)
    extra = m[rng.random(nh) < 0.6].copy()
    shift = rng.integers(0, 30, len(extra)).astype(np.uint64)
    extra["qryStart"] = np.minimum(extra["qryStart"] + shift, qlen[extra["qryId"]] - L)
    extra["qryEnd"] = extra["qryStart"] + L
    extra["subjStart"] = np.minimum(extra["subjStart"] + shift + rng.integers(0, 3, len(extra)).astype(np.uint64), slen[extra["subjId"]] - L)
    extra["subjEnd"] = extra["subjStart"] + L
    return q, qoff, qlen, s, soff, slen, np.concatenate([m, extra])


@pytest.mark.parametrize("scheme,order", [("blosum62", "shuffled"), ("blosum62", "grouped"), ("nucl", "reversed")])
def test_iterate_matches_dev_large_list_equals_host_entry(handle, scheme, order):
    """A list of the size a GPU seeding stage hands over (beyond where the drivers go parallel; several chunks of the pipeline;
    ragged queries -> the multi-query sweep): the device list gives lx_iterate_matches' records, byte for byte."""
    sc_p = SCHEMES[scheme]
    handle.set_scoring(sc_p, 0)
    rng = np.random.default_rng(77)
    dna = scheme == "nucl"
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, 6000, 500, 12, lq_range=(140, 160) if dna else (50, 400),
                                                 alphabet=np.arange(4, dtype=np.uint8) if dna else None)
    ka = capi.karlin_params(0, 2, -3, -5, -2) if dna else capi.karlin_params(62)
    params = capi.SearchParams(1e-2, -1, 0, int(slen.sum()) * 50, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    n = len(m)
    assert n > 100_000
    perm = {"shuffled": rng.permutation(n), "grouped": np.arange(n), "reversed": np.arange(n)[::-1]}[order]
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qlen, 1)
    db, do, ds = handle.iterate_matches_dev(_to_device(m[perm]), n, params)
    hb, ho, hs = handle.iterate_matches(q, qoff, qlen, qlen, None, soff, slen, m[perm], params)
    assert len(db) > 5000
    _assert_records_equal(db, do, hb, ho)
    for f in ("hits_duplicate", "failed_bitscore", "failed_evalue", "failed_identity", "num_ext_score", "num_ext_ali"):
        assert getattr(ds, f) == getattr(hs, f), f
    assert ds.hits_duplicate > 1000
    # LX_ITERATE_NO_OPS: the same records without alignment columns (what a tabular writer needs)
    no_ops = capi.SearchParams(1e-2, -1, 0, int(slen.sum()) * 50, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka, 0, capi.LX_ITERATE_NO_OPS)
    nb, no, _ = handle.iterate_matches_dev(_to_device(m[perm]), n, no_ops)
    assert no == [] and len(nb) == len(db) and (nb["ops_off"] == 0).all()
    for f in nb.dtype.names:
        if f != "ops_off":
            assert (nb[f] == db[f]).all(), f


def test_iterate_matches_dev_bisulfite_and_frames(handle, oracle):
    """iterateMatches' bisulfite branch (:1367-1379): even subject frames with the forward scheme, odd ones with the reverse
    scheme, HSPs stably re-sorted by query; four query frames, two subject frames, the bisulfite overload of
    computeAlignmentStats -- against lx_iterate_matches and the oracle driver."""
    fwd, rev = SCHEMES["bs_fwd"], SCHEMES["bs_rev"]
    handle.set_scoring(fwd, 0)
    handle.set_scoring(rev, 1)
    rng = np.random.default_rng(31)
    nreads, nsub = 300, 6
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, nreads * 4, nsub * 2, 6, lq_range=(100, 150), alphabet=np.arange(4, dtype=np.uint8))
    ka = capi.karlin_params(0, 2, -3, -5, -2)
    qorig = qlen[::4].copy()
    params = capi.SearchParams(1e-3, -1, 0, int(slen.sum()) * 100, 0, 4, 2, 1, capi.LX_FRAMES_BISULFITE, capi.LX_FRAMES_BISULFITE, ka)
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qorig, 4)
    perm = rng.permutation(len(m))
    db, do, ds = handle.iterate_matches_dev(_to_device(m[perm]), len(m), params)
    hb, ho, hs = handle.iterate_matches(q, qoff, qlen, qorig, None, soff, slen, m[perm], params)
    assert len(db) > 100 and (np.diff(db["n_qid"].astype(np.int64)) >= 0).all()
    assert set(np.unique(db["subj_id"] % 2)) == {0, 1}
    _assert_records_equal(db, do, hb, ho)
    for f in ("hits_duplicate", "failed_bitscore", "failed_evalue", "failed_identity", "num_ext_score", "num_ext_ali"):
        assert getattr(ds, f) == getattr(hs, f), f
    with pytest.raises(capi.LambdaExtError):  # the queries were set with four frames
        handle.iterate_matches_dev(_to_device(m), len(m), capi.SearchParams(1e-3, -1, 0, 1000, 0, 2, 2, 1, capi.LX_FRAMES_BISULFITE, capi.LX_FRAMES_BISULFITE, ka))


def test_iterate_matches_hands_large_host_lists_to_the_device(handle, oracle):
    """lx_iterate_matches on a list beyond 131 072 matches: the list work runs on the device (sort words up, window list back into the
    caller's span) -- the records are what the host form gives for the same matches handed over in pieces of queries that stay below
    that size, and the span holds the oracle's window list afterwards."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    rng = np.random.default_rng(99)
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, 9000, 400, 12)
    assert len(m) > 140_000
    m = m[rng.permutation(len(m))]
    ka = capi.karlin_params(62)
    params = capi.SearchParams(1e-2, -1, 0, int(slen.sum()) * 50, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    handle.set_subjects(s)
    lib = capi.load()
    import ctypes as C

    mm = np.ascontiguousarray(m, dtype=capi.MATCH_DTYPE).copy()
    res = C.c_void_p()
    handle._check(lib.lx_iterate_matches(handle.h, 0, capi._ptr(q), q.size, capi._ptr(qoff), capi._ptr(qlen), len(qoff), capi._ptr(qlen), None, 0,
                                         capi._ptr(soff), capi._ptr(slen), len(soff), capi._ptr(mm), len(mm), C.byref(params), C.byref(res)))
    big, bigops, bigstats = handle._take_iterate_result(res)
    want_windows = oracle.widen_and_preprocess(m.astype(oracle_lib.MATCH_DTYPE), qlen, slen)
    assert bigstats.hits_duplicate == len(m) - len(want_windows)
    assert (mm[:len(want_windows)].view(np.uint64) == want_windows.view(np.uint64)).all()  # the span, shrunk to the windows (:1173-1174)
    pieces, pops, dup = [], [], 0
    for lo in range(0, 9000, 3000):  # three pieces of queries, each below the hand-over size: the host form
        part = m[(m["qryId"] >= lo) & (m["qryId"] < lo + 3000)]
        assert len(part) < 131_072
        b, o, st = handle.iterate_matches(q, qoff, qlen, qlen, None, soff, slen, part, params)
        pieces.append(b)
        pops += o
        dup += st.hits_duplicate
    small = np.concatenate(pieces)
    assert dup == bigstats.hits_duplicate and len(small) == len(big) and len(big) > 2000
    for f in small.dtype.names:
        if f != "ops_off":
            assert (small[f] == big[f]).all(), f
    assert pops == bigops
    # a second call with the same sets finds them resident (same records), another query set replaces them
    b2, o2, _ = handle.iterate_matches(q, qoff, qlen, qlen, None, soff, slen, m, params)
    assert b2.tobytes() == big.tobytes() and o2 == bigops
    q3 = q.copy()
    k = int(qoff[7000])
    q3[k:] = np.roll(q[k:], 7)  # the last 2 000 queries (where the planted homologies survive in the subjects) read differently now
    b3, _, _ = handle.iterate_matches(q3, qoff, qlen, qlen, None, soff, slen, m, params)
    assert len(b3) > 2000 and b3.tobytes() != big.tobytes()


@pytest.mark.parametrize("scheme,frames,id_cutoff,min_bits", [("blosum62", 1, 0, -1), ("blosum62", 1, 40, 25), ("nucl", 2, 0, -1), ("nucl", 2, 75, 30)])
def test_records_made_on_the_device_equal_the_host_threads(handle, scheme, frames, id_cutoff, min_bits):
    """The tail of iterateMatchesFullSimd (/root/reference/src/search_algo.hpp:1287-1325: statistics of the filter, the survivors' order,
    _expandAlign, identity cut-off, bit score, e-value) as kernels (lx_records.hip, LX_OPT_ITERATE_RECORDS = 0, the default) against the
    host threads' finishSurvivors on the same survivors (= 1): the same bytes -- records, columns, statistics -- with an identity
    cut-off that drops records and a bit-score filter that splits the failures."""
    sc_p = SCHEMES[scheme]
    handle.set_scoring(sc_p, 0)
    rng = np.random.default_rng(2024 + frames + id_cutoff)
    dna = scheme == "nucl"
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, 3000 * frames, 300, 10, lq_range=(140, 160) if dna else (50, 400),
                                                 alphabet=np.arange(4, dtype=np.uint8) if dna else None)
    qorig = qlen[::frames].copy()
    ka = capi.karlin_params(0, 2, -3, -5, -2) if dna else capi.karlin_params(62)
    mode = capi.LX_FRAMES_REVCOMP if frames == 2 else capi.LX_FRAMES_NONE
    params = capi.SearchParams(1e-2, min_bits, id_cutoff, int(slen.sum()) * 50, 0, frames, 1, 0, mode, capi.LX_FRAMES_NONE, ka)
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qorig, frames)
    d_m = _to_device(m[rng.permutation(len(m))])
    try:
        handle.set_option(capi.LX_OPT_ITERATE_RECORDS, 1)
        hb, ho, hs = handle.iterate_matches_dev(d_m, len(m), params)
        handle.set_option(capi.LX_OPT_ITERATE_RECORDS, 0)
        db, do, ds = handle.iterate_matches_dev(d_m, len(m), params)
    finally:
        handle.set_option(capi.LX_OPT_ITERATE_RECORDS, 0)
    assert len(hb) > 300
    assert db.tobytes() == hb.tobytes()
    assert do == ho
    for f in ("hits_duplicate", "failed_bitscore", "failed_evalue", "failed_identity", "num_ext_score", "num_ext_ali"):
        assert getattr(ds, f) == getattr(hs, f), f
    if id_cutoff:
        assert hs.failed_identity > 0 and hs.failed_bitscore > 0 and hs.failed_evalue > 0


@pytest.mark.parametrize("nq,hits,least,most", [(20, 1500, 2 * 256 + 1, 10 ** 9), (120, 130, 60, 255)])
def test_records_of_queries_with_hundreds_of_windows_each(handle, nq, hits, least, most):
    """The records kernels order a range's survivors by their windows' ranks -- a window's place among the windows of its query, counted
    by one thread over at most 256 neighbours on either side (lx_records.hip: rec_rank_kernel).  Queries with more windows than that
    (a repeat family) send the call back to the full sort words: the same bytes as the host threads' either way -- with every query
    beyond that, and with a hundred or two windows per query, all of them counted."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    rng = np.random.default_rng(4711)
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, nq, 900, hits, lq_range=(50, 120))
    params = capi.SearchParams(10.0, -1, 0, int(slen.sum()) * 50, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, capi.karlin_params(62))
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qlen, 1)
    d_m = _to_device(m[rng.permutation(len(m))])
    try:
        handle.set_option(capi.LX_OPT_ITERATE_RECORDS, 1)
        hb, ho, hs = handle.iterate_matches_dev(d_m, len(m), params)
        handle.set_option(capi.LX_OPT_ITERATE_RECORDS, 0)
        db, do, ds = handle.iterate_matches_dev(d_m, len(m), params)
    finally:
        handle.set_option(capi.LX_OPT_ITERATE_RECORDS, 0)
    windows = handle.widen_and_preprocess_dev(d_m, len(m))
    per_query = np.bincount(windows["qryId"].astype(np.int64))
    # (first case: more windows of every query than the rank kernel looks at on both sides; second: a hundred or two, all counted)
    assert least <= per_query.min() and per_query.max() <= most, (per_query.min(), per_query.max())
    assert len(hb) > 3000
    assert db.tobytes() == hb.tobytes()
    assert do == ho
    for f in ("hits_duplicate", "failed_bitscore", "failed_evalue", "failed_identity", "num_ext_score", "num_ext_ali"):
        assert getattr(ds, f) == getattr(hs, f), f


def test_a_rejected_query_set_leaves_nothing_resident(handle):
    """lx_set_queries / lx_set_subject_seqs validate before they commit: after a rejected set the device entry points report LX_ESTATE
    instead of reading the previous set's device arrays with the new set's sizes."""
    handle.set_scoring(SCHEMES["nucl"], 0)
    rng = np.random.default_rng(5)
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, 200, 20, 4, lq_range=(100, 120), alphabet=np.arange(4, dtype=np.uint8))
    ka = capi.karlin_params(0, 2, -3, -5, -2)
    params = capi.SearchParams(1e-2, -1, 0, int(slen.sum()), 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qlen, 1)
    handle.iterate_matches_dev(_to_device(m), len(m), params)
    bad_len = np.concatenate([qlen, qlen]).astype(np.uint64)  # twice the sequences, the second half beyond the residue buffer
    bad_off = np.concatenate([qoff, qoff + np.uint64(q.size)]).astype(np.uint64)
    with pytest.raises(capi.LambdaExtError):
        handle.set_queries(q, bad_off, bad_len, bad_len, 1)
    with pytest.raises(capi.LambdaExtError) as e:
        handle.iterate_matches_dev(_to_device(m), len(m), params)
    assert "resident" in str(e.value)
    handle.set_queries(q, qoff, qlen, qlen, 1)
    assert len(handle.iterate_matches_dev(_to_device(m), len(m), params)[0]) >= 0
    bad = capi.SearchParams(1e-2, -1, 0, int(slen.sum()), 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka, 0, 0x40)
    with pytest.raises(capi.LambdaExtError):  # unknown flag bits (a caller built against ABI 2 with the word uninitialised)
        handle.iterate_matches_dev(_to_device(m), len(m), bad)


def test_level2_at_the_size_it_is_benched_at(handle, oracle):
    """bench.py --iterate's list (synth.make_seed_list_np: a million reads, 9.5 M matches -> 1.25 M windows: sort tiles x 2 300, 32-bit
    scan carries, two sweep chunks, kept result blocks) against the oracle: ALL windows of _widenAndPreprocessMatches
    (/root/reference/src/search_algo.hpp:1136-1175), the statistics of the filter and the survivor count against the oracle's scores of
    all windows (:1251-1283), 4 000 sampled records field by field against the oracle's alignments (:1296-1325), and on every record the
    identities its counts must satisfy."""
    from lambda_amd import workloads

    w = workloads.WORKLOADS[2]
    d = w.directions[0]
    m_, ma, mi, go, ge = d.scoring
    sc_p = capi.builtin_scoring(m_, match=ma, mismatch=mi, gap_open=go, gap_extend=ge)
    handle.set_scoring(sc_p, d.slot)
    osc = oracle_lib.scoring_from(sc_p)
    ka = capi.karlin_params(*w.karlin)
    oka = oracle_lib.Karlin(ka.lambda_, ka.K, ka.H, ka.alpha, ka.beta)
    q, qoff, qlen, qorig, s, soff, slen, m = synth.make_seed_list_np(1_000_000, 100.0, seed=0x1A3BDA03)
    db_total = int(slen.sum())
    params = capi.SearchParams(w.max_evalue, -1, 0, db_total, 0, 2, 1, 0, capi.LX_FRAMES_REVCOMP, capi.LX_FRAMES_NONE, ka)
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qorig, 2)
    d_m = _to_device(m)
    # ---- every window
    want = oracle.widen_and_preprocess(m.astype(oracle_lib.MATCH_DTYPE), qlen, slen)
    got = handle.widen_and_preprocess_dev(d_m, len(m))
    assert len(got) == len(want) and len(want) > 1_200_000
    for f in ("qryId", "subjId", "qryStart", "qryEnd", "subjStart", "subjEnd"):
        assert (got[f] == want[f]).all(), f
    # ---- the call
    bms, ops, stats = handle.iterate_matches_dev(d_m, len(m), params)
    assert stats.num_ext_score == len(m) and stats.hits_duplicate == len(m) - len(want)
    # the filter on the oracle's scores of all windows: e-value <= 0.01 per (score, query length) with the oracle's own statistics
    ext = np.zeros(len(want), dtype=capi.EXT_DTYPE)
    ext["q_off"], ext["q_len"] = qoff[want["qryId"]], qlen[want["qryId"]]
    ext["s_off"] = soff[want["subjId"]] + want["subjStart"]
    ext["s_len"] = want["subjEnd"] - want["subjStart"]
    scores = oracle.score_batch(q, s, ext, osc, threads=8, simd=True)
    ql = int(qorig[0])
    adj = oracle.length_adjustment(db_total, ql, oka)
    smax = int(scores.max())
    passes = np.array([oracle.evalue(sc_, ql - adj, db_total - adj, oka) <= w.max_evalue for sc_ in range(smax + 1)])
    cut = int(np.argmax(passes))  # (monotone: the first passing score)
    assert passes[cut:].all() and not passes[:cut].any()
    surv = np.nonzero(scores >= cut)[0]
    assert stats.num_ext_ali == len(surv) and stats.failed_evalue == len(want) - len(surv) and stats.failed_bitscore == 0 and stats.failed_identity == 0
    assert len(bms) == len(surv) > 700_000
    # the reference's order: by true query id, inside a query by the slices' lengths, then by list position (:1229-1235, :1299)
    key = np.lexsort((surv, ext["s_len"][surv], ext["q_len"][surv], want["qryId"][surv] // 2))
    order = surv[key]
    assert (bms["qry_id"] == want["qryId"][order]).all() and (bms["subj_id"] == want["subjId"][order]).all()
    assert (bms["score"] == scores[order]).all()
    assert (bms["n_qid"] == bms["qry_id"] // 2).all() and (bms["q_frame"] == np.where(bms["qry_id"] % 2, -1, 1)).all() and (bms["s_frame"] == 0).all()
    # ---- identities on every record
    n_ops = bms["n_ops"].astype(np.int64)
    assert (bms["alignment_length"] == n_ops).all()
    assert (bms["num_matches"] + bms["num_mismatches"] + bms["num_gap_opens"] + bms["num_gap_extensions"] == n_ops).all()
    assert (bms["num_positives"] >= bms["num_matches"]).all()
    assert (bms["q_end"] - bms["q_start"] + bms["s_end"] - bms["s_start"] == 2 * (bms["num_matches"] + bms["num_mismatches"]) + bms["num_gap_opens"] + bms["num_gap_extensions"]).all()
    assert (bms["s_start"] >= want["subjStart"][order]).all() and (bms["s_end"] <= want["subjEnd"][order]).all() and (bms["q_end"] <= ql).all()
    assert (bms["identity"] == (100.0 * bms["num_matches"].astype(np.float32).astype(np.float64) / bms["n_ops"].astype(np.float32).astype(np.float64)).astype(np.float32)).all()
    assert (np.diff(bms["ops_off"].astype(np.int64)) == n_ops[:-1]).all() and len(ops) == len(bms)
    # ---- 4 000 sampled records against the oracle's alignments
    rng = np.random.default_rng(11)
    for r in np.sort(rng.choice(len(bms), 4000, replace=False)):
        x = ext[order[r]]
        qs, ss = q[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])], s[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])]
        hsp, oops = oracle.align(qs, ss, osc)
        st = oracle.alignment_stats(qs, ss, hsp, oops, osc, 0)
        g = bms[r]
        assert (int(g["score"]), int(g["q_start"]), int(g["q_end"]), int(g["s_start"]) - int(want["subjStart"][order[r]]), int(g["s_end"]) - int(want["subjStart"][order[r]])) == \
               (hsp.score, hsp.q_begin, hsp.q_end, hsp.s_begin, hsp.s_end), r
        assert ops[r] == oops, r
        assert (int(g["num_matches"]), int(g["num_mismatches"]), int(g["num_positives"]), int(g["num_gap_opens"]), int(g["num_gap_extensions"])) == \
               (st.num_matches, st.num_mismatches, st.num_positives, st.num_gap_opens, st.num_gap_extensions), r
        wb, we = oracle.bitscore(hsp.score, oka), oracle.evalue(hsp.score, ql - adj, db_total - adj, oka)
        assert abs(g["bit_score"] - wb) <= 1e-6 * abs(wb) and abs(g["e_value"] - we) <= 1e-6 * abs(we), r  # the contract: 1e-6 relative


def test_level2_on_the_protein_list_at_the_size_it_is_benched_at(handle, oracle):
    """bench.py --iterate --config 1's list (synth.make_protein_seed_list_np: BASELINE configs[1] -- 100 000 queries x 150 aa, 7.8 M matches ->
    3.2 M windows, 32 per query -- searchp BLOSUM62, the metric's own program): the plan of the sweep is made on the device
    (lx_plan_free.hip: four queries per wavefront), nothing of the window list comes to the host.  Against the oracle: ALL windows of
    _widenAndPreprocessMatches (/root/reference/src/search_algo.hpp:1136-1175), the filter's statistics and the survivors against the oracle's
    scores of all windows (:1251-1283), order and scores of every record (:1229-1235, :1299), 4 000 sampled records field by field against
    the oracle's alignments (:1296-1325)."""
    from lambda_amd import workloads

    w = workloads.WORKLOADS[1]
    d = w.directions[0]
    m_, ma, mi, go, ge = d.scoring
    sc_p = capi.builtin_scoring(m_, match=ma, mismatch=mi, gap_open=go, gap_extend=ge)
    handle.set_scoring(sc_p, d.slot)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)  # (the library's defaults, whatever the tests before this one left on the session's handle)
    handle.set_option(capi.LX_OPT_MQ_SWEEP, 1)
    osc = oracle_lib.scoring_from(sc_p)
    ka = capi.karlin_params(*w.karlin)
    oka = oracle_lib.Karlin(ka.lambda_, ka.K, ka.H, ka.alpha, ka.beta)
    q, qoff, qlen, qorig, s, soff, slen, m = synth.make_protein_seed_list_np(w.queries_total, seed=w.seed, lq=w.lq, homologs=w.windows // 2, spurious=w.windows // 2)
    db_total = w.db_length
    params = capi.SearchParams(w.max_evalue, -1, 0, db_total, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qorig, 1)
    d_m = _to_device(m)
    want = oracle.widen_and_preprocess(m.astype(oracle_lib.MATCH_DTYPE), qlen, slen)
    got = handle.widen_and_preprocess_dev(d_m, len(m))
    assert len(got) == len(want) and 3_100_000 < len(want) < 3_300_000
    for f in ("qryId", "subjId", "qryStart", "qryEnd", "subjStart", "subjEnd"):
        assert (got[f] == want[f]).all(), f
    bms, ops, stats = handle.iterate_matches_dev(d_m, len(m), params)
    assert "sweep_mq_kernel" in handle.last_kernel_name()  # (the multi-query sweep over the device plan, not the host-planned path)
    assert stats.num_ext_score == len(m) and stats.hits_duplicate == len(m) - len(want)
    ext = np.zeros(len(want), dtype=capi.EXT_DTYPE)
    ext["q_off"], ext["q_len"] = qoff[want["qryId"]], qlen[want["qryId"]]
    ext["s_off"] = soff[want["subjId"]] + want["subjStart"]
    ext["s_len"] = want["subjEnd"] - want["subjStart"]
    scores = oracle.score_batch(q, s, ext, osc, threads=8, simd=True)
    ql = int(qorig[0])
    adj = oracle.length_adjustment(db_total, ql, oka)
    smax = int(scores.max())
    passes = np.array([oracle.evalue(sc_, ql - adj, db_total - adj, oka) <= w.max_evalue for sc_ in range(smax + 1)])
    cut = int(np.argmax(passes))
    assert passes[cut:].all() and not passes[:cut].any()
    surv = np.nonzero(scores >= cut)[0]
    assert stats.num_ext_ali == len(surv) and stats.failed_evalue == len(want) - len(surv) and stats.failed_bitscore == 0 and stats.failed_identity == 0
    assert len(bms) == len(surv) > 1_400_000
    key = np.lexsort((surv, ext["s_len"][surv], ext["q_len"][surv], want["qryId"][surv]))
    order = surv[key]
    assert (bms["qry_id"] == want["qryId"][order]).all() and (bms["subj_id"] == want["subjId"][order]).all()
    assert (bms["score"] == scores[order]).all()
    assert (bms["n_qid"] == bms["qry_id"]).all() and (bms["q_frame"] == 0).all() and (bms["s_frame"] == 0).all()
    n_ops = bms["n_ops"].astype(np.int64)
    assert (bms["alignment_length"] == n_ops).all()
    assert (bms["num_matches"] + bms["num_mismatches"] + bms["num_gap_opens"] + bms["num_gap_extensions"] == n_ops).all()
    assert (bms["num_positives"] >= bms["num_matches"]).all()
    assert (bms["q_end"] - bms["q_start"] + bms["s_end"] - bms["s_start"] == 2 * (bms["num_matches"] + bms["num_mismatches"]) + bms["num_gap_opens"] + bms["num_gap_extensions"]).all()
    assert (bms["s_start"] >= want["subjStart"][order]).all() and (bms["s_end"] <= want["subjEnd"][order]).all() and (bms["q_end"] <= ql).all()
    assert (np.diff(bms["ops_off"].astype(np.int64)) == n_ops[:-1]).all() and len(ops) == len(bms)
    rng = np.random.default_rng(12)
    for r in np.sort(rng.choice(len(bms), 4000, replace=False)):
        x = ext[order[r]]
        qs, ss = q[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])], s[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])]
        hsp, oops = oracle.align(qs, ss, osc)
        st = oracle.alignment_stats(qs, ss, hsp, oops, osc, 0)
        g = bms[r]
        assert (int(g["score"]), int(g["q_start"]), int(g["q_end"]), int(g["s_start"]) - int(want["subjStart"][order[r]]), int(g["s_end"]) - int(want["subjStart"][order[r]])) == \
               (hsp.score, hsp.q_begin, hsp.q_end, hsp.s_begin, hsp.s_end), r
        assert ops[r] == oops, r
        assert (int(g["num_matches"]), int(g["num_mismatches"]), int(g["num_positives"]), int(g["num_gap_opens"]), int(g["num_gap_extensions"])) == \
               (st.num_matches, st.num_mismatches, st.num_positives, st.num_gap_opens, st.num_gap_extensions), r
        wb, we = oracle.bitscore(hsp.score, oka), oracle.evalue(hsp.score, ql - adj, db_total - adj, oka)
        assert abs(g["bit_score"] - wb) <= 1e-6 * abs(wb) and abs(g["e_value"] - we) <= 1e-6 * abs(we), r


def test_device_list_without_the_multi_query_plan_falls_back_to_the_host_threads(handle):
    """LX_OPT_MQ_SWEEP = 0 (or a list the pipeline serves with the one-query-per-wavefront kernels): the survivors come down and the host
    threads make the records -- the same bytes as with the sweep's plan and the records kernels."""
    handle.set_scoring(SCHEMES["blosum62"], 0)
    rng = np.random.default_rng(8)
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, 300, 40, 6)
    ka = capi.karlin_params(62)
    params = capi.SearchParams(1e-2, -1, 0, int(slen.sum()) * 50, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    handle.set_subjects(s)
    handle.set_subject_seqs(soff, slen)
    handle.set_queries(q, qoff, qlen, qlen, 1)
    d_m = _to_device(m)
    want, wops, wst = handle.iterate_matches_dev(d_m, len(m), params)
    try:
        handle.set_option(capi.LX_OPT_MQ_SWEEP, 0)
        got, gops, gst = handle.iterate_matches_dev(d_m, len(m), params)
        one, oops, _ = handle.iterate_matches_dev(_to_device(m[:1]), 1, params)
    finally:
        handle.set_option(capi.LX_OPT_MQ_SWEEP, 1)
    assert len(want) > 100 and got.tobytes() == want.tobytes() and gops == wops
    assert (gst.failed_evalue, gst.num_ext_ali) == (wst.failed_evalue, wst.num_ext_ali)
    assert len(one) <= 1


def test_reserve_then_first_call_gives_the_same_records(lx_lib):
    """lx_reserve changes WHEN the buffers of a handle's first Level-2 call are asked for, nothing else: a fresh handle with and without
    it returns the same bytes; short hints, zero hints and hints before the sets are set are all fine; lx_trim_result_cache gives the kept
    result blocks back."""
    rng = np.random.default_rng(3)
    q, qoff, qlen, s, soff, slen, m = _seed_list(rng, 4000, 60, 10, lq_range=(140, 160), alphabet=np.arange(4, dtype=np.uint8))
    ka = capi.karlin_params(0, 2, -3, -5, -2)
    params = capi.SearchParams(1e-2, -1, 0, int(slen.sum()) * 50, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    d_m = _to_device(m)
    results = []
    for hints in (None, (len(m), len(m) // 2, len(m) // 4, len(m) * 40), (len(m), 16, 1, 0)):
        with capi.Handle(0) as h:
            h.set_scoring(SCHEMES["nucl"], 0)
            if hints:
                h.reserve(*hints)  # (before the sets: sizes what it can)
            h.set_subjects(s)
            h.set_subject_seqs(soff, slen)
            h.set_queries(q, qoff, qlen, qlen, 1)
            if hints:
                h.reserve(*hints)
            b, o, st = h.iterate_matches_dev(d_m, len(m), params)
            results.append((b.tobytes(), o, st.num_ext_ali))
            with pytest.raises(capi.LambdaExtError):
                h.reserve(10, 20, 5, 0)  # more windows than matches
    assert len(results[0][1]) > 500 and results[1] == results[0] and results[2] == results[0]
    assert lx_lib.lx_trim_result_cache() >= 0 and lx_lib.lx_trim_result_cache() == 0

"""GPU parity tests of pass 2 (traceback), the pre-extension filter and the whole driver -- HIP path through the C ABI
vs the CPU oracle, bit-exact on scores / coordinates / ops / counts, exact on doubles computed by the same formulas."""
import numpy as np
import pytest

from lambda_amd import capi, synth
from tests import oracle_driver, oracle_lib
from tests.test_oracle import SCHEMES, alphabet_of

pytestmark = pytest.mark.gpu


def _check_align(handle, oracle, q, s, ext, name, bs_rule=0):
    sc_p = SCHEMES[name]
    handle.set_scoring(sc_p, 0)
    handle.set_option(capi.LX_OPT_BS_MATCH_RULE, bs_rule)
    hsp, ops = handle.align_batch(q, s, ext)
    osc = oracle_lib.scoring_from(sc_p)
    want = oracle.align_batch(q, s, ext, osc)
    nbad = 0
    for i, (oh, oops) in enumerate(want):
        g = hsp[i]
        got = (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"])
        exp = (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops)
        if got != exp or ops[i] != oops:
            nbad += 1
            if nbad < 4:
                print("MISMATCH", i, ext[i], got, exp, ops[i][:60], oops[:60])
            continue
        if oh.score > 0:
            x = ext[i]
            qq = q[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])]
            ss = s[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])]
            st = oracle.alignment_stats(qq, ss, oh, oops, osc, bs_rule)
            assert (g["num_matches"], g["num_mismatches"], g["num_positives"], g["num_gap_opens"], g["num_gap_extensions"]) == \
                   (st.num_matches, st.num_mismatches, st.num_positives, st.num_gap_opens, st.num_gap_extensions), i
    assert nbad == 0, f"{nbad} of {len(ext)} alignments differ"
    handle.set_option(capi.LX_OPT_BS_MATCH_RULE, 0)
    return hsp, ops


@pytest.mark.parametrize("name", ["blosum62", "nucl", "bs_fwd", "bs_rev"])
def test_align_ragged(handle, oracle, name):
    q, s, ext = synth.make_ragged_np(500, seed=31, alphabet=alphabet_of(name), lq_range=(1, 400), ls_extra=(0, 90))
    hsp, _ = _check_align(handle, oracle, q, s, ext, name, bs_rule=int(name.startswith("bs")))
    assert hsp["n_ops"].max() > 30


def test_align_headline_shape(handle, oracle):
    q, s, ext = synth.make_batch_np(100, 150, 32, seed=0x1A3BDA02)
    hsp, ops = _check_align(handle, oracle, q, s, ext, "blosum62")
    assert sum(b"D" in o or b"I" in o for o in ops) > 50  # indels present in the homologous half


def test_align_ties_small_alphabet(handle, oracle):
    # two-letter alphabets: score ties between cells and between diagonal / gap moves everywhere;
    # end cell, begin cell and every op must follow the documented tie rules
    rng = np.random.default_rng(22)
    n = 800
    q = rng.integers(0, 2, 48 * n).astype(np.uint8)
    s = rng.integers(0, 2, 64 * n).astype(np.uint8)
    ext = np.zeros(n, dtype=capi.EXT_DTYPE)
    ext["q_off"] = np.arange(n) * 48
    ext["q_len"] = rng.integers(1, 49, n)
    ext["s_off"] = np.arange(n) * 64
    ext["s_len"] = rng.integers(1, 65, n)
    _check_align(handle, oracle, q, s, ext, "nucl")


def test_align_edge_cases(handle, oracle):
    rng = np.random.default_rng(9)
    q = synth.STD20[rng.integers(0, 20, 3000)].astype(np.uint8)
    s = synth.STD20[rng.integers(0, 20, 9000)].astype(np.uint8)
    s[1000:2500] = q[:1500]
    ext = np.array([
        (0, 0, 0, 0), (0, 0, 10, 0), (0, 0, 0, 10), (0, 1000, 1, 1), (0, 1001, 1, 1),
        (0, 1000, 1500, 1500),   # 10 panels, perfect diagonal
        (0, 900, 700, 1900),     # multi-panel with flanks
        (100, 1100, 160, 176), (100, 1100, 161, 176), (100, 1050, 320, 500),
        (0, 0, 150, 8000),       # long subject window
        (2000, 5000, 900, 3),    # short subject, multi-panel query
    ], dtype=capi.EXT_DTYPE)
    hsp, _ = _check_align(handle, oracle, q, s, ext, "blosum62")
    assert hsp["score"][5] > 5000 and hsp["n_ops"][5] == 1500


def test_align_host_path_known_scores_and_sharing(handle, oracle):
    """lx_align_batch on host buffers: runs of one query share LDS profiles (padding slots are invisible to the caller),
    scores handed over by the caller give the same result as letting pass 2 compute them, and a wrong score is
    reported instead of producing a wrong alignment."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_batch_np(61, 150, 7, seed=4321)   # runs of 7: every run gets one padding slot
    want_score = oracle.score_batch(q, s, ext, osc, threads=8)
    keep = np.nonzero(want_score >= 1)[0]                    # (nearly) all 7 of a run: 12.5 % padding, shared path
    assert len(keep) > 400
    es, ks = ext[keep], want_score[keep]
    want = oracle.align_batch(q, s, es, osc)
    for known, mode in ((None, 1), (ks, 1), (None, 0), (ks, 0)):
        handle.set_option(capi.LX_OPT_PASS2_MODE, mode)
        try:
            hsp, ops = handle.align_batch(q, s, es, known_score=known)
        finally:
            handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        assert ("ckpt_forward_kernel<8,19" if mode else "trace_forward_kernel<8,19") in handle.last_trace_kernel_name()
        for g, (oh, oops), o in zip(hsp, want, ops):
            assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
                   (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops)
            assert o == oops
    bad = ks.copy()
    bad[5] += 3
    with pytest.raises(capi.LambdaExtError):
        handle.align_batch(q, s, es, known_score=bad)


def test_resident_subjects(handle, oracle):
    """lx_set_subjects: the subject buffer is uploaded once; host-buffer calls with s_res = NULL use the resident copy and
    give the same results; without a resident copy a NULL subject buffer is an error."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    q, s, ext = synth.make_batch_np(40, 120, 8, seed=99)
    want = handle.score_batch(q, s, ext)
    hsp_w, ops_w = handle.align_batch(q, s, ext[:64], known_score=want[:64])
    handle.set_subjects(s)
    try:
        got = handle.score_batch(q, None, ext)
        assert (got == want).all()
        hsp_g, ops_g = handle.align_batch(q, None, ext[:64])
        assert (hsp_g == hsp_w).all() and ops_g == ops_w
        bad = ext.copy()
        bad["s_off"][3] = len(s)  # beyond the resident buffer
        with pytest.raises(capi.LambdaExtError):
            handle.score_batch(q, None, bad)
    finally:
        handle.set_subjects(None)
    with pytest.raises(capi.LambdaExtError):
        handle.score_batch(q, None, ext)


def test_align_chunked_trace_workspace(handle, oracle):
    # force several chunks through a tiny direction-bit budget
    handle.set_option(capi.LX_OPT_TRACE_BYTES, 1 << 20)
    try:
        q, s, ext = synth.make_batch_np(40, 150, 8, seed=3)
        _check_align(handle, oracle, q, s, ext, "blosum62")
    finally:
        handle.set_option(capi.LX_OPT_TRACE_BYTES, 4 << 30)


def test_prefilter(handle, oracle):
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    rng = np.random.default_rng(5)
    nq, ns = 50, 20
    qlen = rng.integers(30, 200, nq)
    slen = rng.integers(200, 1500, ns)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]])
    soff = np.concatenate([[0], np.cumsum(slen)[:-1]])
    q = synth.STD20[rng.integers(0, 20, int(qlen.sum()))].astype(np.uint8)
    s = synth.STD20[rng.integers(0, 20, int(slen.sum()))].astype(np.uint8)
    n = 4000
    seeds = np.zeros(n, dtype=capi.SEED_DTYPE)
    for i in range(n):
        a, b = int(rng.integers(0, nq)), int(rng.integers(0, ns))
        L = int(rng.integers(8, 14))
        qs = int(rng.integers(0, qlen[a] - L + 1))
        ss = int(rng.integers(0, slen[b] - L + 1))
        if rng.random() < 0.5:  # plant a diagonal so that about half of the seeds pass
            lo = min(qs, ss, 15)
            hi = min(qlen[a] - qs, slen[b] - ss, 25)
            s[soff[b] + ss - lo: soff[b] + ss + hi] = q[qoff[a] + qs - lo: qoff[a] + qs + hi]
        seeds[i] = (qoff[a], soff[b], qlen[a], slen[b], qs, qs + L, ss, 0)
    for seed_length, pre, thr in ((10, 2, 2.0), (11, 2, 2.0), (10, 1, 2.0), (14, 2, 1.4)):
        got = handle.prefilter_batch(q, s, seeds, seed_length, pre, thr)
        want = np.array([oracle.seed_looks_promising(q[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])],
                                                     s[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])],
                                                     int(x["qry_start"]), int(x["qry_end"]), int(x["subj_start"]),
                                                     seed_length, pre, thr, osc) for x in seeds], dtype=np.uint8)
        assert (got == want).all()
        assert 0.05 < got.mean() < 0.95


def _driver_case(rng, nq=40, ns=12, hits=600, lq=(60, 220)):
    qlen = rng.integers(lq[0], lq[1], nq).astype(np.uint64)
    slen = rng.integers(300, 2500, ns).astype(np.uint64)
    qoff = np.concatenate([[0], np.cumsum(qlen)[:-1]]).astype(np.uint64)
    soff = np.concatenate([[0], np.cumsum(slen)[:-1]]).astype(np.uint64)
    q = synth.STD20[rng.integers(0, 20, int(qlen.sum()))].astype(np.uint8)
    s = synth.STD20[rng.integers(0, 20, int(slen.sum()))].astype(np.uint8)
    m = np.zeros(hits, dtype=capi.MATCH_DTYPE)
    for i in range(hits):
        a, b = int(rng.integers(0, nq)), int(rng.integers(0, ns))
        L = 10
        qs = int(rng.integers(0, int(qlen[a]) - L + 1))
        ss = int(rng.integers(0, int(slen[b]) - L + 1))
        if rng.random() < 0.6:  # homologous region around the seed, with substitutions
            lo = min(qs, ss)
            hi = min(int(qlen[a]) - qs, int(slen[b]) - ss)
            seg = q[int(qoff[a]) + qs - lo: int(qoff[a]) + qs + hi].copy()
            mut = rng.random(len(seg)) < 0.3
            seg[mut] = synth.STD20[rng.integers(0, 20, int(mut.sum()))]
            s[int(soff[b]) + ss - lo: int(soff[b]) + ss + hi] = seg
        m[i] = (a, b, qs, qs + L, ss, ss + L)
    return q, qoff, qlen, s, soff, slen, m


@pytest.mark.parametrize("filters", [(1e-2, -1, 0), (-1.0, 40, 0), (10.0, -1, 35), (-1.0, -1, 0)])
def test_iterate_matches_driver(handle, oracle, filters):
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
    bms, ops, stats = handle.iterate_matches(q, qoff, qlen, qlen, s, soff, slen, m, params)
    want, wstats = oracle_driver.iterate_matches(oracle, osc, oka, q, qoff, qlen, qlen, s, soff, slen,
                                                 m.astype(oracle_lib.MATCH_DTYPE), max_e, min_bits, idcut, db_total)
    assert (stats.hits_duplicate, stats.failed_bitscore, stats.failed_evalue, stats.failed_identity) == \
           (wstats["hits_duplicate"], wstats["failed_bitscore"], wstats["failed_evalue"], wstats["failed_identity"])
    assert len(bms) == len(want) and len(want) > 20
    for g, w, o in zip(bms, want, ops):
        for k in ("qry_id", "subj_id", "n_qid", "n_sid", "q_start", "q_end", "s_start", "s_end", "score", "alignment_length",
                  "num_matches", "num_mismatches", "num_positives", "num_gap_opens", "num_gap_extensions"):
            assert int(g[k]) == w[k], (k, g, w)
        assert o == w["ops"]
        assert g["identity"] == np.float32(w["identity"])
        # same formulas, same libm: equal to the last bit; the contract is 1e-6 relative
        assert abs(g["bit_score"] - w["bit_score"]) <= 1e-6 * abs(w["bit_score"])
        assert abs(g["e_value"] - w["e_value"]) <= 1e-6 * abs(w["e_value"])


@pytest.mark.parametrize("pass2_mode", [1, 0, 2])
@pytest.mark.parametrize("wpq,run,lq", [(32, 32, 150), (8, 8, 150), (7, 0, 150), (32, 32, 200), (16, 16, 100), (16, 16, 300), (8, 8, 450)])
def test_fused_extend_on_device(handle, oracle, wpq, run, lq, pass2_mode):
    """lx_extend_batch_dev: pass 1 -> integer cut-off -> compaction (runs padded to whole wavefronts) -> pass 2.
    lq = 200 is the shape of BASELINE.json configs[3] (200 aa queries, 230 aa windows: the (16,13) geometries); 300 and 450
    columns take two (8,19) / three (16,13) panels: checkpoints carried across panels in modes 1 and 2."""
    import torch

    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_batch_np(90 if lq <= 208 else 24, lq, wpq, seed=1234 + wpq + lq)
    n = len(ext)
    dev = torch.device("cuda:0")
    pad = np.zeros(256, np.uint8)
    d_q = torch.from_numpy(np.concatenate([q, pad])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, pad])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)[:-1]
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
    d_hsp = torch.full((n * 48,), 0xEE, dtype=torch.uint8, device=dev)
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    cutoff = 70
    handle.set_option(capi.LX_OPT_MAX_QLEN, lq)
    handle.set_option(capi.LX_OPT_MAX_SLEN, int(ext["s_len"].max()))
    handle.set_option(capi.LX_OPT_QUERY_RUN, run)
    handle.set_option(capi.LX_OPT_PASS2_MODE, pass2_mode)
    torch.cuda.synchronize()
    try:
        handle.extend_batch_dev(d_q, d_s, d_ext, n, cutoff, d_score, d_hsp, d_ops, d_off, d_count)
        handle.synchronize()
        if run % 8 == 0 and run > 0:  # shared-profile geometries: the mode decides the kernels
            name = handle.last_trace_kernel_name()
            if lq > 208 and pass2_mode == 2 and run % 16 == 0:  # wider than a panel: the packed 16-bit sweep, panel by panel
                assert "sweep_pair16_kernel" in name and ",true>" in name, name
            # (8,13): queries of <= 104 columns; runs of 8: the multi-query sweep, two queries per wavefront
            assert (("ckpt_forward_kernel" in name) or ("score_pair_kernel<8,19,true>" in name) or ("score_pair_kernel<8,13,true>" in name) or
                    ("sweep_mq_kernel" in name)) == (pass2_mode >= 1)
            assert ("sweep_mq_kernel" in name and "2 queries per wavefront" in name) == (pass2_mode == 2 and run == 8)
            assert ("single sweep" in name) == (pass2_mode == 2)
    finally:
        handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
        handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
    want_score = oracle.score_batch(q, s, ext, osc, threads=8)
    got_score = d_score.cpu().numpy()
    assert (got_score == want_score).all()
    surv = np.nonzero(want_score >= cutoff)[0]
    cnt = d_count.cpu().numpy()
    assert cnt[1] == len(surv) and cnt[0] >= cnt[1] and 10 < len(surv) < n
    hsp = np.frombuffer(d_hsp.cpu().numpy().tobytes(), dtype=capi.HSP_DTYPE)
    ops = d_ops.cpu().numpy()
    rejected = np.setdiff1d(np.arange(n), surv)
    assert (hsp["score"][rejected] == want_score[rejected]).all() and (hsp["n_ops"][rejected] == 0).all()
    want = oracle.align_batch(q, s, ext[surv], osc)
    for i, (oh, oops) in zip(surv, want):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), i
        st = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[st: st + oh.n_ops]) == oops


def test_iterate_matches_bisulfite(handle, oracle):
    """iterateMatches' bisulfite branch (src/search_algo.hpp:1367-1379): even subject frames with the forward scheme,
    odd ones with the reverse scheme, result stably re-sorted by query."""
    fwd, rev = SCHEMES["bs_fwd"], SCHEMES["bs_rev"]
    handle.set_scoring(fwd, 0)
    handle.set_scoring(rev, 1)
    # NOT set by hand: params.bisulfite = 1 must select the bisulfite computeAlignmentStats overload by itself
    # (src/evaluate_bisulfite_alignment.hpp:97 via src/search_algo.hpp:1308) and leave the handle's option as it was
    handle.set_option(capi.LX_OPT_BS_MATCH_RULE, 0)
    ka = capi.karlin_params(0, 2, -3, -5, -2)
    oka = oracle_lib.Karlin(ka.lambda_, ka.K, ka.H, ka.alpha, ka.beta)
    rng = np.random.default_rng(77)
    nq, ns = 24, 8  # 4 query frames per read, 2 subject frames per reference (search_datastructures.hpp:380-385)
    qlen = np.full(nq, 100, dtype=np.uint64)
    slen = rng.integers(400, 900, ns).astype(np.uint64)
    qoff = (np.arange(nq) * 100).astype(np.uint64)
    soff = np.concatenate([[0], np.cumsum(slen)[:-1]]).astype(np.uint64)
    q = rng.integers(0, 4, int(qlen.sum())).astype(np.uint8)
    s = rng.integers(0, 4, int(slen.sum())).astype(np.uint8)
    m = np.zeros(160, dtype=capi.MATCH_DTYPE)
    for i in range(len(m)):
        a, b = int(rng.integers(0, nq)), int(rng.integers(0, ns))
        qs_ = int(rng.integers(0, 80))
        ss_ = int(rng.integers(0, int(slen[b]) - 110))
        seg = q[int(qoff[a]): int(qoff[a]) + 100].copy()
        # bisulfite conversion in the direction the subject frame expects: C->T reads vs fwd, G->A vs rev
        if b % 2 == 0:
            ref = seg.copy(); conv = (seg == 3) & (rng.random(100) < 0.5); ref[conv] = 1   # read T where the genome has C
        else:
            ref = seg.copy(); conv = (seg == 0) & (rng.random(100) < 0.5); ref[conv] = 2   # read A where the genome has G
        lo = min(qs_, ss_)
        s[int(soff[b]) + ss_ - lo: int(soff[b]) + ss_ - lo + 100 - (qs_ - lo)] = ref[qs_ - lo:]
        m[i] = (a, b, qs_, qs_ + 17, ss_, ss_ + 17)
    db_total = int(slen.sum())
    params = capi.SearchParams(1e-9, -1, 0, db_total, 0, 4, 2, 1, capi.LX_FRAMES_BISULFITE, capi.LX_FRAMES_BISULFITE, ka)
    try:
        bms, ops, stats = handle.iterate_matches(q, qoff, qlen, np.full(nq // 4, 100, np.uint64), s, soff, slen, m, params)
    finally:
        rule_after = handle.get_option(capi.LX_OPT_BS_MATCH_RULE)
        handle.set_option(capi.LX_OPT_BS_MATCH_RULE, 0)
    assert rule_after == 0  # the call's temporary rule did not leak into the handle
    mo = m.astype(oracle_lib.MATCH_DTYPE)
    want = []
    for parity, scheme in ((0, fwd), (1, rev)):
        sel = mo[mo["subjId"] % 2 == parity]
        w, _ = oracle_driver.iterate_matches(oracle, oracle_lib.scoring_from(scheme), oka, q, qoff, qlen, np.full(nq // 4, 100),
                                             s, soff, slen, sel, 1e-9, -1, 0, db_total, q_frames=4, s_frames=2, bs_rule=1)
        want += w
    want.sort(key=lambda r: r["n_qid"])  # stable
    assert len(bms) == len(want) and len(want) > 20
    assert any(r["subj_id"] % 2 for r in want) and any(r["subj_id"] % 2 == 0 for r in want)
    for g, w, o in zip(bms, want, ops):
        for k in ("qry_id", "subj_id", "n_qid", "n_sid", "q_start", "q_end", "s_start", "s_end", "score", "alignment_length",
                  "num_matches", "num_mismatches", "num_gap_opens", "num_gap_extensions"):
            assert int(g[k]) == w[k], (k, g, w)
        assert o == w["ops"]
        # _setFrames in bisulfite mode (src/search_algo.hpp:778-782, :801-803)
        qid, sid = int(g["qry_id"]), int(g["subj_id"])
        assert int(g["q_frame"]) == (qid % 2 + 1) * (-1 if qid % 4 > 1 else 1) and int(g["s_frame"]) == sid % 2 + 1


@pytest.mark.parametrize("pass2_mode", [1, 0, 2])
def test_full_size_fused_step_properties(handle, oracle, pass2_mode):
    """BASELINE.json configs[1] at FULL size through lx_extend_batch_dev (3.2 M extensions, 1.6 M traced), checked by
    size-independent properties on the device and by the oracle on a sample:
    (1) the survivor count equals the number of scores at or above the cut-off and no extension is flagged (-1);
    (2) every traced HSP carries the pass-1 score; rejected rows carry the score and no alignment;
    (3) per HSP, the column counts add up: n_ops = matches + mismatches + gap columns, the query/subject spans equal the
        number of non-gap columns of their row;
    (4) the self-planted window of every query (the query copied into its first window) aligns end to end without gaps;
    (5) 4 000 sampled survivors are bit-identical to the oracle (coordinates, ops)."""
    import torch

    dev = torch.device("cuda:0")
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    nq, lq, wpq = 100_000, 150, 32
    d_q, d_s, d_ext, ext = synth.make_batch_torch(nq, lq, wpq, 0x1A3BDA02, dev)
    ls, b = synth.window_len(lq), synth.band_size(lq)
    s2 = d_s.view(nq * wpq, ls)
    s2[torch.arange(nq, device=dev) * wpq, b:b + lq] = d_q.view(nq, lq)
    pad = torch.zeros(256, dtype=torch.uint8, device=dev)
    d_q = torch.cat([d_q, pad])
    d_s = torch.cat([s2.reshape(-1), pad])
    n = len(ext)
    sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)[:-1]
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
    d_hsp = torch.full((n * 48,), 0xEE, dtype=torch.uint8, device=dev)
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    cutoff = 91
    handle.set_option(capi.LX_OPT_MAX_QLEN, lq)
    handle.set_option(capi.LX_OPT_MAX_SLEN, ls)
    handle.set_option(capi.LX_OPT_QUERY_RUN, wpq)
    handle.set_option(capi.LX_OPT_PASS2_MODE, pass2_mode)
    handle.set_option(capi.LX_OPT_TRACE_BYTES, 64 << 30)  # the single sweep keeps the checkpoints of all 3.2 M extensions (43 GB)
    torch.cuda.synchronize()
    try:
        handle.extend_batch_dev(d_q, d_s, d_ext, n, cutoff, d_score, d_hsp, d_ops, d_off, d_count)
        handle.synchronize()
        name = handle.last_trace_kernel_name()
        assert ("ckpt_forward_kernel" in name) == (pass2_mode == 1) and ("single sweep" in name) == (pass2_mode == 2)
        # packed-half sweep on the headline shape; no 150-column BLOSUM62 query can fail its exactness test, so the int32
        # fix-up launch is not even issued
        assert ("score_pair_kernel<8,19,true>" in name) == (pass2_mode == 2)
    finally:
        handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
        handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        handle.set_option(capi.LX_OPT_TRACE_BYTES, 4 << 30)
    hsp = d_hsp.view(torch.int32).view(n, 12)  # lx_hsp: score q_begin q_end s_begin s_end n_ops matches mismatches positives opens extensions shift
    surv = d_score >= cutoff
    cnt = d_count.cpu().numpy()
    assert int(cnt[1]) == int(surv.sum()) and int(cnt[1]) > n // 3                      # (1)
    assert bool((hsp[:, 0] == d_score).all())                                             # (1) no -1, (2)
    assert bool((hsp[~surv][:, 5] == 0).all())
    t = hsp[surv].long()
    gaps = t[:, 9] + t[:, 10]
    assert bool((t[:, 5] == t[:, 6] + t[:, 7] + gaps).all())                              # (3)
    assert bool(((t[:, 2] - t[:, 1]) + (t[:, 4] - t[:, 3]) == 2 * (t[:, 6] + t[:, 7]) + gaps).all())
    assert bool((t[:, 5] > 0).all()) and bool((t[:, 8] >= t[:, 6]).all())
    first = hsp[torch.arange(nq, device=dev) * wpq].long()                                # (4)
    assert bool((first[:, 1] == 0).all()) and bool((first[:, 2] == lq).all()) and bool((first[:, 3] == b).all())
    assert bool((first[:, 5] == lq).all()) and bool((first[:, 6] == lq).all()) and bool((first[:, 9] == 0).all())
    # (5) a sample against the oracle
    rng = np.random.default_rng(3)
    sidx = np.sort(rng.choice(np.nonzero(surv.cpu().numpy())[0], 4000, replace=False))
    hs = hsp[torch.from_numpy(sidx).to(dev)].cpu().numpy()
    q_np, s_np = d_q.cpu().numpy(), d_s.cpu().numpy()
    want = oracle.align_batch(q_np, s_np, ext[sidx], osc)
    ops_np = None
    for row, i, (oh, oops) in zip(hs, sidx, want):
        assert tuple(int(x) for x in row[:6]) == (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), i
        st = int(off[i]) + int(row[11])
        got = bytes(d_ops[st: st + oh.n_ops].cpu().numpy())
        assert got == oops, i


def test_published_smith_waterman_example(handle):
    """The worked example of the Smith-Waterman article (TGTTACGG vs GGTTGACTA, +3 / -3, linear gap penalty 2): score 13,
    alignment GTT-AC / GTTGAC -- through the HIP kernels with a caller-supplied matrix, both orientations, and in a
    batch long enough to use the packed-half and shared-profile geometries as well."""
    import ctypes as C

    sc = capi.Scoring()
    sc.alphabet_size, sc.gap_open, sc.gap_extend = 4, -2, -2
    m = np.full((capi.LX_ALPH, capi.LX_ALPH), 0, dtype=np.int8)
    m[:4, :4] = -3
    m[np.arange(4), np.arange(4)] = 3
    C.memmove(sc.matrix, m.ctypes.data, m.nbytes)
    handle.set_scoring(sc, 0)
    try:
        idx = {c: i for i, c in enumerate("ACGT")}
        q = np.array([idx[c] for c in "TGTTACGG"], dtype=np.uint8)
        s = np.array([idx[c] for c in "GGTTGACTA"], dtype=np.uint8)
        res = np.concatenate([q, s])
        ext = np.zeros(32, dtype=capi.EXT_DTYPE)
        ext["q_off"], ext["q_len"], ext["s_off"], ext["s_len"] = 0, 8, 8, 9
        ext[16:]["q_off"], ext[16:]["q_len"], ext[16:]["s_off"], ext[16:]["s_len"] = 8, 9, 0, 8   # roles swapped
        scores = handle.score_batch(res, res, ext)
        assert (scores == 13).all()
        hsp, ops = handle.align_batch(res, res, ext, known_score=scores)
        for k in range(16):
            assert tuple(int(hsp[k][f]) for f in ("score", "q_begin", "q_end", "s_begin", "s_end")) == (13, 1, 6, 1, 7)
            assert ops[k] == b"MMMDMM"
            assert tuple(int(hsp[16 + k][f]) for f in ("score", "q_begin", "q_end", "s_begin", "s_end")) == (13, 1, 7, 1, 6)
            assert ops[16 + k] == b"MMMIMM"
    finally:
        handle.set_scoring(SCHEMES["blosum62"], 0)


@pytest.mark.parametrize("pass2_mode", [2, 1, 0])
def test_fused_ties_two_letter_alphabet(handle, oracle, pass2_mode):
    """The fused step on two-letter sequences: best scores are reached in many cells, rows and strips at once, so the
    end-cell tie rule (first maximum in column-major order), the single sweep's strip / row bookkeeping with its
    ambiguity re-scan, and the traceback ties are all exercised; every survivor must equal the oracle."""
    import torch

    sc_p = SCHEMES["nucl"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    rng = np.random.default_rng(123 + pass2_mode)
    nq, wpq, lq, ls = 96, 8, 120, 150
    q = rng.integers(0, 2, nq * lq).astype(np.uint8)
    s = rng.integers(0, 2, nq * wpq * ls).astype(np.uint8)
    # periodic inserts make equal-scoring alternatives more frequent still
    s.reshape(nq * wpq, ls)[::3, 20:20 + 60] = np.tile(q.reshape(nq, lq)[:, 10:70], (wpq, 1)).reshape(nq * wpq, 60)[::3]
    ext = np.zeros(nq * wpq, dtype=capi.EXT_DTYPE)
    ext["q_off"] = np.repeat(np.arange(nq) * lq, wpq)
    ext["q_len"] = lq
    ext["s_off"] = np.arange(nq * wpq) * ls
    ext["s_len"] = ls
    n = len(ext)
    dev = torch.device("cuda:0")
    pad = np.zeros(256, np.uint8)
    d_q = torch.from_numpy(np.concatenate([q, pad])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, pad])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)[:-1]
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
    d_hsp = torch.full((n * 48,), 0xEE, dtype=torch.uint8, device=dev)
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    cutoff = 20
    handle.set_option(capi.LX_OPT_MAX_QLEN, lq)
    handle.set_option(capi.LX_OPT_MAX_SLEN, ls)
    handle.set_option(capi.LX_OPT_QUERY_RUN, wpq)
    handle.set_option(capi.LX_OPT_PASS2_MODE, pass2_mode)
    torch.cuda.synchronize()
    try:
        handle.extend_batch_dev(d_q, d_s, d_ext, n, cutoff, d_score, d_hsp, d_ops, d_off, d_count)
        handle.synchronize()
        assert ("single sweep" in handle.last_trace_kernel_name()) == (pass2_mode == 2)
    finally:
        handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
        handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        handle.set_scoring(SCHEMES["blosum62"], 0)
    want_score = oracle.score_batch(q, s, ext, osc, threads=8)
    assert (d_score.cpu().numpy() == want_score).all()
    surv = np.nonzero(want_score >= cutoff)[0]
    assert len(surv) > n // 2
    hsp = np.frombuffer(d_hsp.cpu().numpy().tobytes(), dtype=capi.HSP_DTYPE)
    ops = d_ops.cpu().numpy()
    for i, (oh, oops) in zip(surv, oracle.align_batch(q, s, ext[surv], osc)):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), i
        st = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[st: st + oh.n_ops]) == oops


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_fused_fuzz_shapes_and_modes(handle, oracle, seed):
    """Random shapes through the fused step: query length 20..208 (all sweep geometries and the fall-backs beyond),
    runs of 8 / 16 / 24 / 32 windows, ragged window lengths (including empty and very short ones), protein and
    nucleotide schemes, random cut-offs -- every pass-2 mode must reproduce the oracle for every survivor."""
    import torch

    rng = np.random.default_rng(1000 + seed)
    name = ["blosum62", "nucl", "blosum62", "bs_fwd", "blosum62", "nucl"][seed - 1]
    lq = int([33, 150, 208, 97, 152, 61][seed - 1] if seed <= 6 else rng.integers(20, 209))
    wpq = int([8, 16, 24, 32, 8, 16][seed - 1])
    alpha = synth.STD20 if name == "blosum62" else np.arange(4, dtype=np.uint8)
    sc_p = SCHEMES[name]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    nq = 24
    q, s, ext = synth.make_batch_np(nq, lq, wpq, seed=77 + seed, alphabet=alpha, sub_rate=0.15 if name == "blosum62" else 0.05,
                                    indel_rate=0.03)
    ext = ext.copy()
    full = ext["s_len"].copy()
    cut = rng.random(len(ext))
    ext["s_len"] = np.where(cut < 0.05, 0, np.where(cut < 0.15, rng.integers(1, 12, len(ext)),
                            np.where(cut < 0.5, (full * rng.uniform(0.4, 1.0, len(ext))).astype(np.uint32), full))).astype(np.uint32)
    n = len(ext)
    want_score = oracle.score_batch(q, s, ext, osc, threads=8)
    cutoff = int(np.percentile(want_score, 45)) + 1
    surv = np.nonzero((want_score >= cutoff) & (ext["s_len"] > 0))[0]
    want = oracle.align_batch(q, s, ext[surv], osc)
    dev = torch.device("cuda:0")
    pad = np.zeros(256, np.uint8)
    d_q = torch.from_numpy(np.concatenate([q, pad])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, pad])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    sizes = ext["q_len"].astype(np.uint64) + full.astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)[:-1]
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    for mode in (2, 1, 0):
        d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
        d_hsp = torch.full((n * 48,), 0xEE, dtype=torch.uint8, device=dev)
        d_score = torch.zeros(n, dtype=torch.int32, device=dev)
        d_count = torch.zeros(2, dtype=torch.int64, device=dev)
        handle.set_option(capi.LX_OPT_MAX_QLEN, lq)
        handle.set_option(capi.LX_OPT_MAX_SLEN, int(full.max()))
        handle.set_option(capi.LX_OPT_QUERY_RUN, wpq)
        handle.set_option(capi.LX_OPT_PASS2_MODE, mode)
        torch.cuda.synchronize()
        try:
            handle.extend_batch_dev(d_q, d_s, d_ext, n, cutoff, d_score, d_hsp, d_ops, d_off, d_count)
            handle.synchronize()
        finally:
            handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
            handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
            handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
            handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        assert (d_score.cpu().numpy() == want_score).all(), (mode, handle.last_kernel_name())
        assert int(d_count.cpu()[1]) == len(surv)
        hsp = np.frombuffer(d_hsp.cpu().numpy().tobytes(), dtype=capi.HSP_DTYPE)
        ops = d_ops.cpu().numpy()
        for i, (oh, oops) in zip(surv, want):
            g = hsp[i]
            assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
                   (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), (mode, i, handle.last_trace_kernel_name())
            st = int(off[i]) + int(g["ops_shift"])
            assert bytes(ops[st: st + oh.n_ops]) == oops, (mode, i)
    handle.set_scoring(SCHEMES["blosum62"], 0)


def test_fused_rejects_broken_length_promise(handle):
    """LX_OPT_MAX_SLEN / LX_OPT_MAX_QLEN size the checkpoint slots; a window or query longer than promised must be
    reported (never written past its slot), in every pass-2 mode."""
    import torch

    handle.set_scoring(SCHEMES["blosum62"], 0)
    q, s, ext = synth.make_batch_np(16, 150, 16, seed=5)
    n = len(ext)
    dev = torch.device("cuda:0")
    pad = np.zeros(256, np.uint8)
    d_q = torch.from_numpy(np.concatenate([q, pad])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, pad])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)[:-1]
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
    d_hsp = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    for mode in (2, 1, 0):
        handle.set_option(capi.LX_OPT_MAX_QLEN, 150)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 100)  # the windows are 176 long
        handle.set_option(capi.LX_OPT_QUERY_RUN, 16)
        handle.set_option(capi.LX_OPT_PASS2_MODE, mode)
        try:
            handle.extend_batch_dev(d_q, d_s, d_ext, n, 60, d_score, d_hsp, d_ops, d_off, d_count)
            with pytest.raises(capi.LambdaExtError):
                handle.synchronize()
        finally:
            handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
            handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
            handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
            handle.set_option(capi.LX_OPT_PASS2_MODE, 1)


@pytest.mark.parametrize("pass2_mode", [2, 1, 0])
@pytest.mark.parametrize("order", ["grouped", "shuffled"])
def test_extend_batch_host_buffers(handle, oracle, order, pass2_mode):
    """lx_extend_batch: both passes on host buffers with a synchronisation in between.  Queries of several lengths in one
    call, ragged runs (not multiples of 16), empty windows, per-extension cut-offs, grouped and shuffled lists: scores of
    all extensions and the alignment of every survivor must equal the oracle's, ops compact and addressed by offset."""
    rng = np.random.default_rng(4242)
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    qs, ss, exts = [], [], []
    qo = so = 0
    # (200 columns: the (8,25) strips; 330 and 440: three panels -- compact codes over panels in the single sweep)
    for k, (lq, wpq) in enumerate([(150, 37), (97, 16), (150, 5), (33, 50), (200, 21), (330, 19), (440, 32)]):
        q, s, ext = synth.make_batch_np(6, lq, wpq, seed=500 + k, sub_rate=0.2, indel_rate=0.03)
        ext = ext.copy()
        ext["q_off"] += qo
        ext["s_off"] += so
        qo += len(q)
        so += len(s)
        qs.append(q), ss.append(s), exts.append(ext)
    q, s, ext = np.concatenate(qs), np.concatenate(ss), np.concatenate(exts)
    cut = rng.random(len(ext))
    ext["s_len"] = np.where(cut < 0.04, 0, np.where(cut < 0.3, (ext["s_len"] * 0.6).astype(np.uint32), ext["s_len"])).astype(np.uint32)
    if order == "shuffled":
        ext = ext[rng.permutation(len(ext))]
    n = len(ext)
    want_score = oracle.score_batch(q, s, ext, osc, threads=8)
    mins = (np.percentile(want_score, 50) + rng.integers(-6, 7, n)).astype(np.int32)
    surv = np.nonzero((want_score >= mins) & (ext["s_len"] > 0) & (ext["q_len"] > 0))[0]
    want = oracle.align_batch(q, s, ext[surv], osc)
    handle.set_option(capi.LX_OPT_PASS2_MODE, pass2_mode)
    try:
        score, hsp, off, ops = handle.extend_batch(q, s, ext, mins)
        # resident subjects + one cut-off for all
        handle.set_subjects(s)
        score2, hsp2, off2, ops2 = handle.extend_batch(q, None, ext, int(np.percentile(want_score, 50)))
    finally:
        handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
        handle.set_subjects(None)
    assert (score == want_score).all(), handle.last_kernel_name()
    assert (score2 == want_score).all()
    assert (hsp["score"] == want_score).all()
    gone = np.setdiff1d(np.arange(n), surv)
    assert (hsp["n_ops"][gone] == 0).all()
    total = 0
    for i, (oh, oops) in zip(surv, want):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), (i, handle.last_trace_kernel_name())
        st = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[st: st + oh.n_ops]) == oops, i
        total += oh.n_ops
    assert len(ops) == max(total, 1)  # compact: exactly the survivors' alignment columns
    surv2 = np.nonzero((want_score >= int(np.percentile(want_score, 50))) & (ext["s_len"] > 0))[0]
    assert (np.nonzero(hsp2["n_ops"])[0] == surv2[want_score[surv2] > 0]).all()


@pytest.mark.parametrize("lq,G,C", [(200, 8, 25), (206, 16, 13), (330, 8, 19)])
def test_sweep_overflow_slots_for_declined_wavefronts(handle, oracle, lq, G, C):
    """Single sweep with compact slots: wavefronts whose query fails the packed-half exactness gate (tryptophan-rich 200 aa
    queries: the bound on the intermediates exceeds 2046, real scores beyond 2047 occur) go to the int32 kernel and get
    int16-pair slots in the overflow area; their neighbours keep the compact ones.  Both must reproduce the oracle.
    200 columns take the (8,25) strips, 201-208 the (16,13) ones; 330 columns are three (8,19) panels of compact codes written
    by the packed int16 kernel, which declines by extension, after the fact (best score beyond 2046)."""
    rng = np.random.default_rng(99)
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    wpq, nq = 16, 10
    q, s, ext = synth.make_batch_np(nq, lq, wpq, seed=4711, sub_rate=0.1, indel_rate=0.02)
    q, s = q.copy(), s.copy()
    W = 22
    heavy = [1, 4, 5, 8]
    for k in heavy:
        x0 = ext[k * wpq]
        qs = slice(int(x0["q_off"]), int(x0["q_off"]) + lq)
        q[qs] = np.where(rng.random(lq) < 0.97, W, q[qs])
        for w, x in enumerate(ext[k * wpq: (k + 1) * wpq: 2]):  # every other window: a copy of the query, the first one exact
            ls = int(x["s_len"])
            b = (ls - lq) // 2
            win = s[int(x["s_off"]): int(x["s_off"]) + ls]
            win[b: b + lq] = np.where(rng.random(lq) < (1.0 if w == 0 else 0.9), q[qs], win[b: b + lq])
    want_score = oracle.score_batch(q, s, ext, osc, threads=8)
    assert want_score.max() > 2047  # beyond what a compact code can hold
    cutoff = 60
    surv = np.nonzero(want_score >= cutoff)[0]
    want = oracle.align_batch(q, s, ext[surv], osc)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    score, hsp, off, ops = handle.extend_batch(q, s, ext, cutoff)
    panels = (lq + G * C - 1) // (G * C)
    kernel = f"score_pair_kernel<{G},{C},true>" if panels == 1 else f"sweep_pair16_kernel<{G},{C},true,true,true>"
    assert "single sweep" in handle.last_trace_kernel_name() and kernel in handle.last_trace_kernel_name()
    assert (score == want_score).all()
    heavy_surv = 0
    for i, (oh, oops) in zip(surv, want):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), i
        st = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[st: st + oh.n_ops]) == oops, i
        heavy_surv += (i // wpq) in heavy
    assert heavy_surv >= 8 and len(surv) - heavy_surv >= 8  # both kinds of slots were walked
    # a trace budget that leaves too small an overflow area: the call must fail loudly, not return wrong alignments
    steps = (int(ext["s_len"].max()) + G - 1 + 15) & ~15
    ck16, ck32 = ((C + 1) // 2 + 3) // 4 * 4, (C + 3) // 4 * 4  # dwords per lane per row checkpoint: codes / int16 pairs
    compact, pairs = panels * (steps * G // 2 + steps // 16 * G * ck16), panels * (steps * G + steps // 16 * G * ck32)  # uint32 per slot
    handle.set_option(capi.LX_OPT_TRACE_BYTES, ((len(ext) + 1) * compact + 10 * pairs) * 4)
    try:
        with pytest.raises(capi.LambdaExtError, match="checkpoint slot"):
            handle.extend_batch(q, s, ext, cutoff)
    finally:
        handle.set_option(capi.LX_OPT_TRACE_BYTES, 64 << 30)
    score, hsp, off, ops = handle.extend_batch(q, s, ext, cutoff)  # and the handle is usable afterwards
    assert (score == want_score).all()


@pytest.mark.parametrize("gap_open,gap_extend", [(-38, -2), (-30, -1), (0, -3)])
def test_sweep_unusual_gap_costs(handle, oracle, gap_open, gap_extend):
    """The compact checkpoint codes hold (value - gap state) in 5 bits: a first gap character dearer than 31 sends the
    sweep to the int32 kernel (int16 pairs), exactly 31 still packs, and gapOpen = 0 (linear gaps) is the other edge."""
    sc_p = capi.builtin_scoring(62, gap_open=gap_open, gap_extend=gap_extend)
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_batch_np(12, 150, 16, seed=31337, sub_rate=0.15, indel_rate=0.05)
    want_score = oracle.score_batch(q, s, ext, osc, threads=8)
    cutoff = int(np.percentile(want_score, 40)) + 1
    surv = np.nonzero(want_score >= cutoff)[0]
    want = oracle.align_batch(q, s, ext[surv], osc)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    try:
        score, hsp, off, ops = handle.extend_batch(q, s, ext, cutoff)
        name = handle.last_trace_kernel_name()
    finally:
        handle.set_scoring(SCHEMES["blosum62"], 0)
    assert "single sweep" in name and ("score_pair_kernel" in name) == (-(gap_open + gap_extend) <= 31)
    assert (score == want_score).all()
    for i, (oh, oops) in zip(surv, want):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), (i, name)
        st = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[st: st + oh.n_ops]) == oops, i


def test_published_durbin_local_alignment(handle):
    """The worked local-alignment example of Durbin et al. (figure 2.6: HEAGAWGHEE vs PAWHEAE, BLOSUM50, linear gap cost 8,
    best local alignment AWGHE / AW-HE with score 28) through every entry point: pass 1, pass 2 and the fused step."""
    import ctypes as C

    from tests.test_oracle import DURBIN_LETTERS, durbin_example

    m6, q, s = durbin_example()
    sc = capi.Scoring()
    sc.alphabet_size, sc.gap_open, sc.gap_extend = len(DURBIN_LETTERS), -8, -8
    m = np.zeros((capi.LX_ALPH, capi.LX_ALPH), dtype=np.int8)
    m[:6, :6] = m6
    C.memmove(sc.matrix, m.ctypes.data, m.nbytes)
    handle.set_scoring(sc, 0)
    try:
        res = np.concatenate([q, s])
        ext = np.zeros(32, dtype=capi.EXT_DTYPE)
        ext["q_off"], ext["q_len"], ext["s_off"], ext["s_len"] = 0, len(q), len(q), len(s)
        assert (handle.score_batch(res, res, ext) == 28).all()
        hsp, ops = handle.align_batch(res, res, ext)
        fscore, fhsp, foff, fops = handle.extend_batch(res, res, ext, 20)
        assert (fscore == 28).all() and "single sweep" in handle.last_trace_kernel_name()
        for k in range(32):
            assert tuple(int(hsp[k][f]) for f in ("score", "q_begin", "q_end", "s_begin", "s_end")) == (28, 4, 9, 1, 5)
            assert ops[k] == b"MMIMM"
            assert tuple(int(fhsp[k][f]) for f in ("score", "q_begin", "q_end", "s_begin", "s_end", "n_ops")) == (28, 4, 9, 1, 5, 5)
            st = int(foff[k]) + int(fhsp[k]["ops_shift"])
            assert bytes(fops[st: st + 5]) == b"MMIMM"
    finally:
        handle.set_scoring(SCHEMES["blosum62"], 0)


def test_full_size_host_entry_point_equals_device_path(handle):
    """BASELINE.json configs[1] at a quarter of its size through lx_extend_batch (host buffers, resident subjects) and
    through lx_extend_batch_dev: scores, records and op strings of every survivor are identical; the host call's ops
    buffer holds exactly the survivors' slots."""
    import torch

    dev = torch.device("cuda:0")
    handle.set_scoring(SCHEMES["blosum62"], 0)
    nq, lq, wpq, cutoff = 25_000, 150, 32, 91
    q, s, ext = synth.make_batch_np(nq, lq, wpq, seed=0x1A3BDA02)
    n = len(ext)
    pad = np.zeros(256, np.uint8)
    d_q, d_s = torch.from_numpy(np.concatenate([q, pad])).to(dev), torch.from_numpy(np.concatenate([s, pad])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)[:-1]
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
    d_hsp = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    handle.set_option(capi.LX_OPT_MAX_QLEN, lq)
    handle.set_option(capi.LX_OPT_MAX_SLEN, int(ext["s_len"].max()))
    handle.set_option(capi.LX_OPT_QUERY_RUN, wpq)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    try:
        handle.extend_batch_dev(d_q, d_s, d_ext, n, cutoff, d_score, d_hsp, d_ops, d_off, d_count)
        handle.synchronize()
    finally:
        handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
        handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
    dev_score = d_score.cpu().numpy()
    dev_hsp = np.frombuffer(d_hsp.cpu().numpy().tobytes(), dtype=capi.HSP_DTYPE)
    dev_ops = d_ops.cpu().numpy()
    handle.set_subjects(s)
    try:
        score, hsp, hoff, ops = handle.extend_batch(q, None, ext, cutoff, copy_ops=False)
        assert (score == dev_score).all()
        surv = np.nonzero(dev_score >= cutoff)[0]
        assert len(surv) > n // 3 and int(d_count.cpu()[1]) == len(surv)
        for f in ("score", "q_begin", "q_end", "s_begin", "s_end", "n_ops", "num_matches", "num_mismatches", "num_positives",
                  "num_gap_opens", "num_gap_extensions"):
            assert (hsp[f] == dev_hsp[f]).all(), f
        assert len(ops) == int(hsp["n_ops"][surv].sum())  # compact: exactly the survivors' alignment columns
        rng = np.random.default_rng(5)
        for i in rng.choice(surv, 20_000, replace=False):
            a = int(hoff[i]) + int(hsp[i]["ops_shift"])
            b = int(off[i]) + int(dev_hsp[i]["ops_shift"])
            k = int(hsp[i]["n_ops"])
            assert bytes(ops[a: a + k]) == bytes(dev_ops[b: b + k]), i
    finally:
        handle.set_subjects(None)


def test_extend_batch_rle_codes_expand_to_the_column_bytes(handle, oracle):
    """lx_extend_batch_rle: the ops in the form they cross PCIe in -- one byte per run, (op << 6) | (length - 1), runs beyond
    64 columns split -- must expand (lx_expand_ops) to exactly the column bytes lx_extend_batch returns, for several chunks
    of the pipeline (LX_OPT_EXTEND_CHUNK) and for a single one."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    q, s, ext = synth.make_batch_np(30_000, 150, 16, seed=91, sub_rate=0.2, indel_rate=0.03)  # 480 k extensions
    cut = 80
    handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 60_000)  # eight chunks: the two lanes of the pipeline are reused four times
    try:
        score, hsp, off, ops = handle.extend_batch(q, s, ext, cut, copy_ops=True)
        score2, hsp2, off2, codes = handle.extend_batch_rle(q, s, ext, cut)
    finally:
        handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 0)
    score1, hsp1, off1, ops1 = handle.extend_batch(q, s, ext, cut, copy_ops=True)  # one chunk: same results, same layout
    assert (score1 == score).all() and (hsp1["n_ops"] == hsp["n_ops"]).all() and (off1 == off).all() and (ops1 == ops).all()
    assert (score == score2).all()
    for f in ("score", "q_begin", "q_end", "s_begin", "s_end", "n_ops", "num_matches", "num_gap_opens"):
        assert (hsp[f] == hsp2[f]).all(), f
    surv = np.nonzero(hsp["n_ops"])[0]
    assert len(surv) > 100_000 and len(codes) < len(ops) // 8
    rng = np.random.default_rng(1)
    long_runs = 0
    for i in rng.choice(surv, 3000, replace=False):
        k = int(hsp["n_ops"][i])
        a = int(off[i]) + int(hsp["ops_shift"][i])
        c = codes[int(off2[i]): int(off2[i]) + k]  # (never more codes than columns)
        assert capi.Handle.expand_ops(c, k) == bytes(ops[a: a + k]), i
        long_runs += int((c[:4] & 63).max() == 63)
    assert long_runs > 100  # runs of more than 64 columns occur and are split
    # the oracle on a few of them
    osc = oracle_lib.scoring_from(sc_p)
    some = rng.choice(surv, 200, replace=False)
    for i, (oh, oops) in zip(some, oracle.align_batch(q, s, ext[some], osc)):
        a = int(off[i])
        assert bytes(ops[a: a + oh.n_ops]) == oops and hsp["n_ops"][i] == oh.n_ops


def test_extend_batch_with_a_window_beyond_65535_residues(handle, oracle):
    """_widenAndPreprocessMatches merges chains of overlapping windows without a bound (src/search_algo.hpp:1153-1157): one
    long merged window (here 70 000 residues, beyond what the checkpoint slots address) must not fail the batch -- it is
    traced through the direction-bit path with the rest of the list, bit-exact as ever."""
    rng = np.random.default_rng(9)
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_batch_np(12, 120, 16, seed=77, sub_rate=0.2, indel_rate=0.03)
    long_len = 70_000
    long_s = rng.integers(0, 20, long_len).astype(np.uint8)
    long_s[61_234: 61_234 + 120] = q[:120]                      # the first query, planted deep inside the long window
    long_s[61_234 + 40: 61_234 + 44] = rng.integers(0, 20, 4)   # with a few substitutions
    s2 = np.concatenate([s, long_s])
    ext2 = np.concatenate([ext, ext[:1]])
    ext2[-1]["s_off"], ext2[-1]["s_len"] = len(s), long_len
    want_score = oracle.score_batch(q, s2, ext2, osc, threads=8)
    cut = int(np.percentile(want_score, 40))
    surv = np.nonzero(want_score >= cut)[0]
    assert len(ext2) - 1 in surv
    want = oracle.align_batch(q, s2, ext2[surv], osc)
    score, hsp, off, ops = handle.extend_batch(q, s2, ext2, cut)
    assert (score == want_score).all()
    for i, (oh, oops) in zip(surv, want):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), i
        st = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[st: st + oh.n_ops]) == oops, i
    assert hsp[len(ext2) - 1]["s_begin"] >= 61_000


def test_adaptive_pass2_follows_the_survivor_share(handle, oracle):
    """LX_OPT_ADAPT_PERMILLE: after a batch in which few extensions passed the cut-off, the next step runs plain pass 1 and
    checkpoints for the survivors only (the reference's order, /root/reference/src/search_algo.hpp:1246 / :1251-1283 / :1296)
    instead of the single sweep -- and goes back to the sweep when many survive again.  Same results either way."""
    import torch

    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    dev = torch.device("cuda:0")
    names = []

    def run(homolog_frac, seed):
        q, s, ext = synth.make_batch_np(120, 150, 16, seed=seed, homolog_frac=homolog_frac)
        n = len(ext)
        pad = np.zeros(256, np.uint8)
        d_q = torch.from_numpy(np.concatenate([q, pad])).to(dev)
        d_s = torch.from_numpy(np.concatenate([s, pad])).to(dev)
        d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
        sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
        off = np.zeros(n, dtype=np.uint64)
        off[1:] = np.cumsum(sizes)[:-1]
        d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
        d_hsp = torch.full((n * 48,), 0xEE, dtype=torch.uint8, device=dev)
        d_score = torch.zeros(n, dtype=torch.int32, device=dev)
        d_count = torch.zeros(2, dtype=torch.int64, device=dev)
        handle.extend_batch_dev(d_q, d_s, d_ext, n, 91, d_score, d_hsp, d_ops, d_off, d_count)
        handle.synchronize()
        names.append(handle.last_trace_kernel_name())
        want = oracle.score_batch(q, s, ext, osc, threads=8)
        assert (d_score.cpu().numpy() == want).all()
        surv = np.nonzero(want >= 91)[0]
        assert int(d_count.cpu().numpy()[1]) == len(surv)
        hsp = np.frombuffer(d_hsp.cpu().numpy().tobytes(), dtype=capi.HSP_DTYPE)
        ops = d_ops.cpu().numpy()
        for i, (oh, oops) in zip(surv, oracle.align_batch(q, s, ext[surv], osc)):
            g = hsp[i]
            assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
                   (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), i
            st = int(off[i]) + int(g["ops_shift"])
            assert bytes(ops[st: st + oh.n_ops]) == oops
        return len(surv) / n

    handle.set_option(capi.LX_OPT_MAX_QLEN, 150)
    handle.set_option(capi.LX_OPT_MAX_SLEN, 176)
    handle.set_option(capi.LX_OPT_QUERY_RUN, 16)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    handle.set_option(capi.LX_OPT_ADAPT_PERMILLE, 30)  # (also forgets what the handle has seen so far)
    try:
        f1 = run(0.01, 11)   # unknown share: the sweep
        f2 = run(0.01, 12)   # 1 % survived last time: adapted
        f3 = run(0.5, 13)    # still adapted (the share it knows is the last batch's)
        f4 = run(0.5, 14)    # half survived last time: the sweep again
    finally:
        handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
        handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
    assert f1 < 0.03 and f2 < 0.03 and f3 > 0.3 and f4 > 0.3
    assert "single sweep" in names[0] and "single sweep" in names[3], names
    assert "single sweep" not in names[1] and "single sweep" not in names[2], names
    assert "ckpt_forward_kernel" in names[1] and "ckpt_forward_kernel" in names[2], names

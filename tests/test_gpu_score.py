"""GPU parity tests of pass 1 (score only): HIP path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from lambda_amd import capi, synth
from tests import oracle_lib
from tests.test_oracle import SCHEMES, alphabet_of

pytestmark = pytest.mark.gpu


def _check(handle, oracle, q, s, ext, name):
    sc_p = SCHEMES[name]
    handle.set_scoring(sc_p, 0)
    got = handle.score_batch(q, s, ext)
    want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(sc_p), threads=8)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"{len(bad)} mismatches, first at {bad[:5]}: got {got[bad[:5]]} want {want[bad[:5]]} ext {ext[bad[:5]]}"
    return got


@pytest.mark.parametrize("name", ["blosum62", "nucl", "bs_fwd", "bs_rev"])
def test_ragged_all_geometries(handle, oracle, name):
    # query lengths 1..700 exercise all four kernel geometries and the multi-panel carry path (> 640 columns)
    q, s, ext = synth.make_ragged_np(1500, seed=11, alphabet=alphabet_of(name), lq_range=(1, 700), ls_extra=(0, 90))
    got = _check(handle, oracle, q, s, ext, name)
    assert got.max() > 50


def test_headline_shape_shared_profiles(handle, oracle):
    # config-2 geometry: 150 aa queries, 32 windows each of 176 residues -> one profile per wavefront
    q, s, ext = synth.make_batch_np(300, 150, 32, seed=0x1A3BDA02)
    got = _check(handle, oracle, q, s, ext, "blosum62")
    assert (got > 100).mean() > 0.3  # the homologous half scores high


def test_config1_shape(handle, oracle):
    q, s, ext = synth.make_batch_np(200, 100, 7, seed=0x1A3BDA01)  # 7 windows/query: runs not a multiple of 4
    _check(handle, oracle, q, s, ext, "blosum62")


def test_searchn_shape_with_n(handle, oracle):
    q, s, ext = synth.make_batch_np(400, 150, 8, seed=0x1A3BDA03, alphabet=np.array([0, 1, 2, 4], dtype=np.uint8),
                                    n_rate=0.01, n_rank=3, sub_rate=0.05, indel_rate=0.01)
    _check(handle, oracle, q, s, ext, "nucl")


def test_edge_cases(handle, oracle):
    rng = np.random.default_rng(9)
    q = synth.STD20[rng.integers(0, 20, 5000)].astype(np.uint8)
    s = synth.STD20[rng.integers(0, 20, 20000)].astype(np.uint8)
    s[1000:3000] = q[:2000]  # long perfect match: high scores, long subject
    ext = np.array([
        (0, 0, 0, 0),            # both empty
        (0, 0, 10, 0),           # empty subject
        (0, 0, 0, 10),           # empty query
        (0, 1000, 1, 1),         # single cell, match
        (0, 1001, 1, 1),         # single cell, probably mismatch
        (0, 1000, 2000, 2000),   # 2000x2000 identical: 4 panels of 640, score in the thousands
        (0, 900, 1999, 2300),    # multi-panel with flanks
        (100, 1100, 160, 176),   # exactly one panel of geometry 0
        (100, 1100, 161, 176),   # one column more -> next geometry
        (100, 1100, 64, 90),     # exactly one panel of geometry 1
        (100, 1100, 65, 90),
        (100, 1100, 320, 400),
        (100, 1100, 321, 400),
        (100, 1100, 640, 700),
        (100, 1100, 641, 700),   # two panels, second holds a single column
        (0, 0, 150, 19000),      # very long subject window
        (3000, 5000, 1500, 3),   # very short subject, multi-panel query
    ], dtype=capi.EXT_DTYPE)
    got = _check(handle, oracle, q, s, ext, "blosum62")
    assert got[0] == got[1] == got[2] == 0
    assert got[5] > 5000


def test_ties_small_alphabet(handle, oracle):
    # two-letter sequences: massive score ties and gap/diagonal ties; scores must still match exactly
    rng = np.random.default_rng(21)
    n = 600
    q = rng.integers(0, 2, 64 * n).astype(np.uint8)
    s = rng.integers(0, 2, 96 * n).astype(np.uint8)
    ext = np.zeros(n, dtype=capi.EXT_DTYPE)
    ext["q_off"] = np.arange(n) * 64
    ext["q_len"] = rng.integers(1, 65, n)
    ext["s_off"] = np.arange(n) * 96
    ext["s_len"] = rng.integers(1, 97, n)
    _check(handle, oracle, q, s, ext, "nucl")


def test_device_resident_entry_point(handle, oracle):
    import torch

    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    q, s, ext = synth.make_batch_np(256, 150, 32, seed=5)
    dev = torch.device("cuda:0")
    d_q = torch.from_numpy(np.concatenate([q, np.zeros(256, np.uint8)])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, np.zeros(256, np.uint8)])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    d_out = torch.full((len(ext),), -7, dtype=torch.int32, device=dev)
    want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(sc_p), threads=8)
    torch.cuda.synchronize()
    for max_qlen, run in ((0, 0), (150, 0), (150, 32), (150, 8)):
        handle.set_option(capi.LX_OPT_MAX_QLEN, max_qlen)
        handle.set_option(capi.LX_OPT_QUERY_RUN, run)
        d_out.fill_(-7)
        torch.cuda.synchronize()
        handle.score_batch_dev(d_q, d_s, d_ext, len(ext), d_out)
        handle.synchronize()
        assert (d_out.cpu().numpy() == want).all(), (max_qlen, run)
        assert handle.last_kernel_ms() > 0
    # a violated LX_OPT_QUERY_RUN promise is reported, not silently mis-scored: 7 windows per query but "runs of 8"
    q, s, ext = synth.make_batch_np(64, 150, 7, seed=6)
    d_q = torch.from_numpy(np.concatenate([q, np.zeros(256, np.uint8)])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, np.zeros(256, np.uint8)])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    handle.set_option(capi.LX_OPT_MAX_QLEN, 150)
    handle.set_option(capi.LX_OPT_QUERY_RUN, 8)
    handle.score_batch_dev(d_q, d_s, d_ext, len(ext), d_out)
    with pytest.raises(capi.LambdaExtError):
        handle.synchronize()
    handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
    handle.set_option(capi.LX_OPT_MAX_QLEN, 0)


def test_linearity_property_full_size_rows(handle):
    # size-independent property at the headline geometry: scaling matrix and gap costs by 2 doubles every score
    q, s, ext = synth.make_batch_np(2000, 150, 32, seed=77)
    sc1 = capi.builtin_scoring(62, gap_open=-11, gap_extend=-1)
    sc2 = capi.builtin_scoring(62, gap_open=-11, gap_extend=-1)
    m = np.ctypeslib.as_array(sc2.matrix)
    m *= 2
    sc2.gap_open *= 2
    sc2.gap_extend *= 2
    handle.set_scoring(sc1, 0)
    a = handle.score_batch(q, s, ext)
    handle.set_scoring(sc2, 0)
    b = handle.score_batch(q, s, ext)
    assert (b == 2 * a).all()
    # self-alignment of a query scores the sum of its diagonal entries
    M = sc1.matrix_np()
    handle.set_scoring(sc1, 0)
    qq = q.reshape(2000, 150)
    ext2 = np.zeros(2000, dtype=capi.EXT_DTYPE)
    ext2["q_off"] = np.arange(2000) * 150
    ext2["s_off"] = np.arange(2000) * 150
    ext2["q_len"] = 150
    ext2["s_len"] = 150
    got = handle.score_batch(q, q, ext2)
    assert (got == M[qq, qq].sum(axis=1)).all()


def _dev_scores(handle, q, s, ext, max_qlen, run, packed, max_slen=0):
    import torch

    dev = torch.device("cuda:0")
    pad = np.zeros(256, np.uint8)
    d_q = torch.from_numpy(np.concatenate([q, pad])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, pad])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    d_out = torch.full((len(ext),), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    handle.set_option(capi.LX_OPT_MAX_QLEN, max_qlen)
    handle.set_option(capi.LX_OPT_QUERY_RUN, run)
    handle.set_option(capi.LX_OPT_PACKED_HALF, packed)
    handle.set_option(capi.LX_OPT_MAX_SLEN, max_slen)
    try:
        handle.score_batch_dev(d_q, d_s, d_ext, len(ext), d_out)
        handle.synchronize()
        name = handle.last_kernel_name()
    finally:
        handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
        handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
        handle.set_option(capi.LX_OPT_PACKED_HALF, 1)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
    return d_out.cpu().numpy(), name


@pytest.mark.parametrize("lq,wpq", [(150, 32), (100, 16), (64, 16), (120, 48), (190, 16), (33, 16), (200, 32), (208, 16)])
def test_packed_half_kernel_is_exact(handle, oracle, lq, wpq):
    """The packed-half pass-1 kernel (lx_score_f16.hip) must be bit-identical to the oracle and to the int32 kernel."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    q, s, ext = synth.make_batch_np(96, lq, wpq, seed=1000 + lq)
    want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(sc_p), threads=8)
    got, name = _dev_scores(handle, q, s, ext, lq, wpq, 1)
    assert "score_pair_kernel" in name
    assert (got == want).all()
    got32, name32 = _dev_scores(handle, q, s, ext, lq, wpq, 0)
    assert "score_pair_kernel" not in name32 and (got32 == want).all()


@pytest.mark.parametrize("method,gaps", [(45, (-15, -2)), (80, (-10, -1)), (45, (-19, -1)), (80, (-9, -2))])
def test_packed_half_kernel_is_exact_with_blosum45_and_blosum80(handle, oracle, method, gaps):
    """The other two matrices prepareScoring() offers (src/search_algo.hpp:198-219) through the same kernels, with gap costs
    NCBI has Karlin-Altschul values for: packed-half and int32 scores equal the oracle's, the fused step's alignments too
    (BLOSUM45 has entries up to 15: the exactness gate sees larger bounds than with BLOSUM62)."""
    sc_p = capi.builtin_scoring(method, gap_open=gaps[0], gap_extend=gaps[1])
    assert capi.karlin_params(method, gap_open=gaps[0], gap_extend=gaps[1]).lambda_ > 0
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    lq, wpq = 150, 32
    q, s, ext = synth.make_batch_np(64, lq, wpq, seed=4500 + method, sub_rate=0.3, indel_rate=0.03)
    want = oracle.score_batch(q, s, ext, osc, threads=8)
    got, name = _dev_scores(handle, q, s, ext, lq, wpq, 1)
    assert "score_pair_kernel" in name and (got == want).all()
    got32, _ = _dev_scores(handle, q, s, ext, lq, wpq, 0)
    assert (got32 == want).all()
    cut = int(np.percentile(want, 60))
    score, hsp, off, ops = handle.extend_batch(q, s, ext, cut)
    assert (score == want).all()
    surv = np.nonzero(want >= cut)[0][:300]
    for i, (oh, oops) in zip(surv, oracle.align_batch(q, s, ext[surv], osc)):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), i
        assert bytes(ops[int(off[i]): int(off[i]) + oh.n_ops]) == oops


def test_packed_half_declines_when_bound_too_large(handle, oracle):
    """Tryptophan-rich queries (W/W = 11): the per-wavefront bound exceeds what half precision holds exactly, the
    packed kernel must leave them to the int32 fix-up launch; mixed with ordinary queries in one batch."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    lq, wpq = 190, 16
    q, s, ext = synth.make_batch_np(40, lq, wpq, seed=4242)
    W = 22  # SeqAn rank of 'W'
    qq = q.reshape(40, lq)
    ss = s.reshape(40 * wpq, -1)
    for k in range(0, 40, 3):  # every third query: all W, and its windows too -> scores up to 11 * 190 = 2090 > 2048
        qq[k, :] = W
        ss[k * wpq:(k + 1) * wpq, 10:10 + lq] = W
    q, s = qq.reshape(-1), ss.reshape(-1)
    want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(sc_p), threads=8)
    assert want.max() == 11 * lq
    got, name = _dev_scores(handle, q, s, ext, lq, wpq, 1)
    assert "score_pair_kernel" in name
    assert (got == want).all()


def test_packed_half_other_schemes(handle, oracle):
    for name, alpha in (("nucl", np.array([0, 1, 2, 4], dtype=np.uint8)), ("bs_fwd", np.arange(4, dtype=np.uint8))):
        sc_p = SCHEMES[name]
        handle.set_scoring(sc_p, 0)
        q, s, ext = synth.make_batch_np(64, 150, 16, seed=77, alphabet=alpha, sub_rate=0.05, indel_rate=0.02)
        want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(sc_p), threads=8)
        got, kn = _dev_scores(handle, q, s, ext, 150, 16, 1)
        assert "score_pair_kernel" in kn and (got == want).all()


def test_packed_half_runs_of_8(handle, oracle):
    """BASELINE.json configs[2]/[4] shape: 8 windows per read.  The packed kernel runs with one query per HALF wavefront:
    two LDS profiles per wavefront (for protein alphabets that costs occupancy and still beats the 16-lane geometry)."""
    for name, alpha, want_pair in (("nucl", np.array([0, 1, 2, 4, 3], dtype=np.uint8), "score_pair_kernel<8,19>"),
                                   ("bs_rev", np.arange(4, dtype=np.uint8), "score_pair_kernel<8,19>"),
                                   ("blosum62", synth.STD20, "score_pair_kernel<8,19>")):
        sc_p = SCHEMES[name]
        handle.set_scoring(sc_p, 0)
        q, s, ext = synth.make_batch_np(203, 150, 8, seed=99, alphabet=alpha, sub_rate=0.06, indel_rate=0.02)
        want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(sc_p), threads=8)
        got, kn = _dev_scores(handle, q, s, ext, 150, 8, 1)
        assert want_pair in kn, kn
        assert (got == want).all()
    handle.set_scoring(SCHEMES["blosum62"], 0)


def test_full_size_batch_properties(handle):
    """BASELINE.json configs[1] at FULL size (100 000 x 150 aa x 32 windows = 3.2 M extensions, 84.5 Gcells), checked
    through size-independent properties, no oracle: (1) two independent kernels -- packed half and int32 -- agree on
    every extension (checksum and element-wise); (2) planting the query itself in a window makes the score the sum of the
    query's diagonal entries; (3) scores are bounded by that sum and non-negative."""
    import torch

    dev = torch.device("cuda:0")
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    nq, lq, wpq = 100_000, 150, 32
    d_q, d_s, d_ext, ext = synth.make_batch_torch(nq, lq, wpq, 0x1A3BDA02, dev)
    ls = synth.window_len(lq)
    b = synth.band_size(lq)
    # plant query k verbatim into its window 0 (rows b..b+lq)
    s2 = d_s.view(nq * wpq, ls)
    s2[torch.arange(nq, device=dev) * wpq, b:b + lq] = d_q.view(nq, lq)
    pad = torch.zeros(256, dtype=torch.uint8, device=dev)
    d_q = torch.cat([d_q, pad])
    d_s = torch.cat([s2.reshape(-1), pad])
    n = len(ext)
    outs = {}
    for packed in (1, 0):
        d_out = torch.full((n,), -9, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        handle.set_option(capi.LX_OPT_MAX_QLEN, lq)
        handle.set_option(capi.LX_OPT_QUERY_RUN, wpq)
        handle.set_option(capi.LX_OPT_PACKED_HALF, packed)
        try:
            handle.score_batch_dev(d_q, d_s, d_ext, n, d_out)
            handle.synchronize()
            assert ("score_pair_kernel" in handle.last_kernel_name()) == bool(packed)
        finally:
            handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
            handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
            handle.set_option(capi.LX_OPT_PACKED_HALF, 1)
        outs[packed] = d_out
    assert int(outs[1].sum(dtype=torch.int64)) == int(outs[0].sum(dtype=torch.int64))
    assert bool((outs[1] == outs[0]).all())
    M = torch.from_numpy(sc_p.matrix_np().astype(np.int32)).to(dev)
    qv = d_q[: nq * lq].view(nq, lq).long()
    selfscore = M[qv, qv].sum(dim=1).to(torch.int32)
    sc = outs[1].view(nq, wpq)
    assert bool((sc[:, 0] == selfscore).all())
    assert bool((sc >= 0).all()) and bool((sc <= selfscore[:, None]).all())


@pytest.mark.parametrize("name,lq,wpq", [("blosum62", 230, 16), ("blosum62", 450, 16), ("blosum62", 777, 32), ("nucl", 600, 16)])
def test_packed_16bit_integer_kernel_wide_queries(handle, oracle, name, lq, wpq):
    """Queries wider than every packed-half geometry: the packed 16-bit integer kernel (integer adds, maxima through the
    half-precision comparators on biased values) sweeps them in (8,19) panels; scores equal the oracle's and the int32
    kernel's, ragged windows and empty ones included."""
    alpha = synth.STD20 if name == "blosum62" else np.arange(4, dtype=np.uint8)
    sc_p = SCHEMES[name]
    handle.set_scoring(sc_p, 0)
    try:
        q, s, ext = synth.make_batch_np(9, lq, wpq, seed=lq + wpq, alphabet=alpha, sub_rate=0.2 if name == "blosum62" else 0.05, indel_rate=0.03)
        ext = ext.copy()
        rng = np.random.default_rng(lq)
        cut = rng.random(len(ext))
        ext["s_len"] = np.where(cut < 0.05, 0, np.where(cut < 0.4, (ext["s_len"] * rng.uniform(0.2, 1.0, len(ext))).astype(np.uint32),
                                                         ext["s_len"])).astype(np.uint32)
        want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(sc_p), threads=8)
        got, kn = _dev_scores(handle, q, s, ext, lq, wpq, 1, max_slen=int(ext["s_len"].max()))
        assert "sweep_pair16_kernel<8,19,true,false>" in kn, kn
        assert (got == want).all()
        got32, kn32 = _dev_scores(handle, q, s, ext, lq, wpq, 0, max_slen=int(ext["s_len"].max()))
        assert "pair16" not in kn32 and (got32 == want).all()
    finally:
        handle.set_scoring(SCHEMES["blosum62"], 0)

"""GPU tests of band mode (LX_OPT_BAND): NOT the reference's configuration (src/search_algo.hpp:1081 runs BandOff), so the
yardstick is the oracle's banded restatement (lxo_score_banded / lxo_align_banded, themselves pinned against a banded
general-gap Smith-Waterman in tests/test_oracle.py): scores, end / begin cells and ops must be identical for every entry
point, with explicit and with default centre diagonals, and the band must vanish when it covers the rectangle."""
import numpy as np
import pytest

from lambda_amd import capi, synth
from tests import oracle_lib
from tests.test_oracle import SCHEMES, alphabet_of

pytestmark = pytest.mark.gpu


def _slices(q, s, x):
    return (q[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])], s[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])])


def _default_centre(lq, ls):
    return max(0, min(synth.band_size(lq), ls - lq))


@pytest.fixture
def banded(handle):
    yield handle
    handle.set_band(0)


@pytest.mark.parametrize("name", ["blosum62", "nucl"])
def test_banded_scores_and_alignments_match_the_oracle(banded, oracle, name):
    h = banded
    sc_p = SCHEMES[name]
    h.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    rng = np.random.default_rng(61)
    q, s, ext = synth.make_ragged_np(300, seed=17, alphabet=alphabet_of(name), lq_range=(1, 330), ls_extra=(0, 70))
    centres = rng.integers(-5, 40, len(ext)).astype(np.int32)
    for band in (0, 3, 17, 64):
        if band == 0:
            continue
        h.set_band(band, centres)
        got = h.score_batch(q, s, ext)
        h.set_band(band, centres)
        hsp, ops = h.align_batch(q, s, ext)
        for i, x in enumerate(ext):
            qq, ss = _slices(q, s, x)
            lo, hi = int(centres[i]) - band, int(centres[i]) + band
            want = oracle.score_banded(qq, ss, osc, lo, hi)
            assert got[i] == want, (band, i, x, got[i], want)
            oh, oops = oracle.align_banded(qq, ss, osc, lo, hi)
            g = hsp[i]
            assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
                   (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), (band, i)
            assert ops[i] == oops, (band, i)


def test_default_centres_follow_the_window_builder(banded, oracle):
    """Without centres the band sits on min(_bandSize(q_len), s_len - q_len): the seed diagonal of an unclipped _widenMatch
    window (src/search_algo.hpp:919-938, src/search_misc.hpp:46-50)."""
    h = banded
    sc_p = SCHEMES["blosum62"]
    h.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_batch_np(12, 150, 8, seed=5, sub_rate=0.2, indel_rate=0.05)
    ext = ext.copy()
    ext["s_len"][::5] -= 9  # a few windows clipped at the subject's end
    h.set_band(9)
    score, hsp, off, ops = h.extend_batch(q, s, ext, 40)
    for i, x in enumerate(ext):
        qq, ss = _slices(q, s, x)
        d0 = _default_centre(len(qq), len(ss))
        oh, oops = oracle.align_banded(qq, ss, osc, d0 - 9, d0 + 9)
        assert score[i] == oh.score, i
        if oh.score >= 40:
            g = hsp[i]
            assert (g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == (oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops)
            st = int(off[i]) + int(g["ops_shift"])
            assert bytes(ops[st: st + oh.n_ops]) == oops
        else:
            assert hsp[i]["n_ops"] == 0


def test_band_that_covers_the_rectangle_changes_nothing(banded, oracle):
    h = banded
    sc_p = SCHEMES["blosum62"]
    h.set_scoring(sc_p, 0)
    q, s, ext = synth.make_batch_np(40, 150, 16, seed=0x1A3BDA02)
    full_score, full_hsp, full_off, full_ops = h.extend_batch(q, s, ext, 91)
    h.set_band(400)
    b_score, b_hsp, b_off, b_ops = h.extend_batch(q, s, ext, 91)
    assert "band" in h.last_kernel_name()
    assert (b_score == full_score).all()
    for k in ("score", "q_begin", "q_end", "s_begin", "s_end", "n_ops", "num_matches", "num_mismatches", "num_gap_opens"):
        assert (b_hsp[k] == full_hsp[k]).all(), k
    for i in np.nonzero(full_hsp["n_ops"])[0]:
        a = int(full_off[i]) + int(full_hsp["ops_shift"][i])
        b = int(b_off[i]) + int(b_hsp["ops_shift"][i])
        assert bytes(full_ops[a: a + int(full_hsp["n_ops"][i])]) == bytes(b_ops[b: b + int(b_hsp["n_ops"][i])])
    # a narrow band can only lower scores
    h.set_band(2)
    assert (h.score_batch(q, s, ext) <= full_score).all()


def test_banded_device_entry_points(banded, oracle):
    """lx_score_batch_dev / lx_extend_batch_dev with centres in device memory (what bench.py --band times)."""
    import torch

    h = banded
    dev = torch.device("cuda:0")
    sc_p = SCHEMES["blosum62"]
    h.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    lq, wpq, nq = 150, 8, 30
    q, s, ext = synth.make_batch_np(nq, lq, wpq, seed=77, sub_rate=0.2, indel_rate=0.04)
    n = len(ext)
    pad = np.zeros(256, np.uint8)
    d_q = torch.from_numpy(np.concatenate([q, pad])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, pad])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    rng = np.random.default_rng(3)
    centres = rng.integers(5, 22, n).astype(np.int32)
    d_c = torch.from_numpy(centres).to(dev)
    band = 11
    h.set_option(capi.LX_OPT_MAX_QLEN, lq)
    h.set_option(capi.LX_OPT_MAX_SLEN, int(ext["s_len"].max()))
    h.set_option(capi.LX_OPT_QUERY_RUN, wpq)
    try:
        h.set_band(band, d_centres=d_c)
        d_score = torch.zeros(n, dtype=torch.int32, device=dev)
        h.score_batch_dev(d_q, d_s, d_ext, n, d_score)
        h.synchronize()
        sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
        off = np.zeros(n, dtype=np.uint64)
        off[1:] = np.cumsum(sizes)[:-1]
        d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
        d_hsp = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
        d_count = torch.zeros(2, dtype=torch.int64, device=dev)
        d_score2 = torch.zeros(n, dtype=torch.int32, device=dev)
        h.extend_batch_dev(d_q, d_s, d_ext, n, 45, d_score2, d_hsp, d_ops, d_off, d_count)
        h.synchronize()
    finally:
        h.set_option(capi.LX_OPT_MAX_QLEN, 0)
        h.set_option(capi.LX_OPT_MAX_SLEN, 0)
        h.set_option(capi.LX_OPT_QUERY_RUN, 0)
    got, got2 = d_score.cpu().numpy(), d_score2.cpu().numpy()
    hsp = d_hsp.cpu().numpy().view(capi.HSP_DTYPE)
    ops = d_ops.cpu().numpy()
    nsurv = 0
    for i, x in enumerate(ext):
        qq, ss = _slices(q, s, x)
        oh, oops = oracle.align_banded(qq, ss, osc, int(centres[i]) - band, int(centres[i]) + band)
        assert got[i] == oh.score and got2[i] == oh.score, i
        if oh.score >= 45:
            nsurv += 1
            g = hsp[i]
            assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
                   (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), i
            st = int(off[i]) + int(g["ops_shift"])
            assert bytes(ops[st: st + oh.n_ops]) == oops
    assert nsurv > 20 and int(d_count.cpu()[1]) == nsurv


def test_band_centres_apply_to_one_call_only(banded, oracle):
    """lx_set_band_centres is one-shot: the call after it runs with the default centres again, whatever its size --
    it neither fails on a stale array of another length nor silently reuses one of the same length."""
    h = banded
    sc_p = SCHEMES["blosum62"]
    h.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_batch_np(6, 120, 4, seed=8)
    centres = np.full(len(ext), 30, dtype=np.int32)
    h.set_band(5, centres)
    with_centres = h.score_batch(q, s, ext)
    again = h.score_batch(q, s, ext)            # same size: default centres now
    shorter = h.score_batch(q, s, ext[:10])     # another size: no LX_EINVAL
    for i, x in enumerate(ext):
        qq, ss = _slices(q, s, x)
        d0 = min(int(np.sqrt(x["q_len"])) + 1, int(x["s_len"]) - int(x["q_len"]))
        assert with_centres[i] == oracle.score_banded(qq, ss, osc, 30 - 5, 30 + 5)
        assert again[i] == oracle.score_banded(qq, ss, osc, d0 - 5, d0 + 5)
    assert (shorter == again[:10]).all() and (with_centres != again).any()

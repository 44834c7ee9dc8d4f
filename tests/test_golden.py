"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).
CPU part: the oracle still reproduces them.  GPU part: the HIP path reproduces them without the oracle in the loop."""
from pathlib import Path

import numpy as np
import pytest

from lambda_amd import capi
from tests import oracle_lib

GOLD = Path(__file__).resolve().parent / "golden"
NAMES = ["blosum62", "nucl", "bs_fwd", "bs_rev"]


def load(name):
    z = np.load(GOLD / f"{name}.npz")
    ext = z["ext"].view(capi.EXT_DTYPE)
    return z, ext


def scoring_of(z):
    a, go, ge = (int(x) for x in z["gaps"])
    return oracle_lib.make_scoring(a, z["matrix"], go, ge)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(oracle, name):
    z, ext = load(name)
    sc = scoring_of(z)
    score, qe, se = oracle.score_batch(z["q"], z["s"], ext, sc, ends=True)
    assert (score == z["score"]).all() and (qe == z["q_end"]).all() and (se == z["s_end"]).all()
    score2 = oracle.score_batch(z["q"], z["s"], ext, sc, simd=True)
    assert (score2 == z["score"]).all()
    for i, (h, o) in enumerate(oracle.align_batch(z["q"], z["s"], ext, sc)):
        assert (h.score, h.q_begin, h.q_end, h.s_begin, h.s_end, h.n_ops) == tuple(z["hsp"][i])
        assert np.frombuffer(o, dtype=np.uint8).tolist() == z["ops"][z["ops_off"][i]: z["ops_off"][i + 1]].tolist()
    assert z["hsp"][:, 0].max() > 100 or name != "blosum62"


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_golden(handle, name):
    z, ext = load(name)
    a, go, ge = (int(x) for x in z["gaps"])
    sc = capi.Scoring()
    sc.alphabet_size, sc.gap_open, sc.gap_extend = a, go, ge
    m = np.ascontiguousarray(z["matrix"], dtype=np.int8)
    import ctypes as C

    C.memmove(sc.matrix, m.ctypes.data, m.nbytes)
    handle.set_scoring(sc, 0)
    handle.set_option(capi.LX_OPT_BS_MATCH_RULE, int(z["bs_rule"][0]))
    try:
        got = handle.score_batch(z["q"], z["s"], ext)
        assert (got == z["score"]).all()
        hsp, ops = handle.align_batch(z["q"], z["s"], ext)
    finally:
        handle.set_option(capi.LX_OPT_BS_MATCH_RULE, 0)
    for i in range(len(ext)):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == tuple(z["hsp"][i]), i
        assert list(ops[i]) == z["ops"][z["ops_off"][i]: z["ops_off"][i + 1]].tolist(), i
        if g["score"] > 0:
            assert (g["num_matches"], g["num_mismatches"], g["num_positives"], g["num_gap_opens"],
                    g["num_gap_extensions"]) == tuple(z["stats"][i]), i

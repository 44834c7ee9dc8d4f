"""The free-packing plan of the multi-query sweep made ON THE DEVICE (lambda_amd/csrc/lx_plan_free.hip) -- what the Level-2 driver
plans protein window lists with, where the reference sorts its list of alignments by the slices' lengths so that a SIMD batch's
windows take about as many steps (/root/reference/src/search_algo.hpp:1229-1235).  Checked slot by slot, against the promises
the sweep kernel relies on (LX_OPT_QUERY_RUN = 2) and against what makes a plan a good one."""
import numpy as np
import pytest

from lambda_amd import capi

pytestmark = pytest.mark.gpu


def cols_per_lane(lq, C=19):
    """lx_host.cpp: mq_panels / lx_plan_free.hip: fp_cols_per_lane -- whole panels + the narrowest last one that covers the rest."""
    panel = 8 * C
    P = max(1, -(-lq // panel))
    rem = max(lq, 1) - (P - 1) * panel
    last = (C + 3) // 4 if rem <= 8 * ((C + 3) // 4) else (C + 1) // 2 if rem <= 8 * ((C + 1) // 2) else (3 * C + 3) // 4 if rem <= 8 * ((3 * C + 3) // 4) else C
    return min(0xfff, (P - 1) * C + last)


def make_list(rng, n_queries, windows, qlen, merged_share=0.1, clipped_share=0.1):
    """A window list as _widenAndPreprocessMatches leaves it: grouped by query; a query's windows are its length + 2 bands long, a share of
    them merged ones of up to three times that, a share clipped at their subject's end.  windows / qlen: callables of the rng."""
    ext = []
    q_off = 0
    for q in range(n_queries):
        lq = int(qlen(rng))
        w = int(windows(rng))
        band = int(np.sqrt(lq)) + 1
        base = lq + 2 * band
        for _ in range(w):
            u = rng.random()
            ls = int(base * rng.uniform(1.2, 3.0)) if u < merged_share else int(base * rng.uniform(0.1, 0.9)) if u < merged_share + clipped_share else base
            ext.append((q_off, int(rng.integers(0, 1 << 20)), lq, max(1, ls - int(rng.integers(0, 4)))))
        q_off += lq
    return np.array(ext, dtype=capi.EXT_DTYPE)


def check_plan(ext, plan, pan, maxs, rep, cuts, C=19):
    n = len(ext)
    nwf = len(plan)
    assert rep[1] == 0 and rep[0] == nwf
    real = (plan & 0x80000000) == 0
    idx = plan & 0x7fffffff
    assert (idx < n).all()
    # every window exactly once
    counts = np.bincount(idx[real], minlength=n)
    assert (counts == 1).all(), f"{(counts != 1).sum()} windows not placed exactly once"
    # a filler copies a window of its own wavefront (it costs what that window costs, nothing more)
    for w in np.nonzero(~real.all(axis=1))[0][:20000]:
        have = set(idx[w][real[w]].tolist())
        assert real[w, 0] and set(idx[w][~real[w]].tolist()) <= have, w
    q = ext["q_off"][idx]  # [nwf, 16]
    # a lane group's two windows share a query slice
    assert (q[:, 0::2] == q[:, 1::2]).all()
    assert (ext["q_len"][idx][:, 0::2] == ext["q_len"][idx][:, 1::2]).all()
    # at most four query slices per wavefront
    qs = np.sort(q, axis=1)
    distinct = 1 + (qs[:, 1:] != qs[:, :-1]).sum(axis=1)
    assert distinct.max() <= 4, distinct.max()
    # the wavefront's widest query and longest window
    want_pan = np.vectorize(lambda l: cols_per_lane(int(l), C))(ext["q_len"][idx]).max(axis=1)
    assert (pan == want_pan).all()
    assert (maxs == ext["s_len"][idx].max(axis=1)).all()
    # ranges: contiguous pieces of the plan
    starts = rep[4: 4 + len(cuts)]
    assert starts[0] == 0 and starts[-1] == nwf and (np.diff(starts.astype(np.int64)) > 0).all()
    for r in range(len(cuts) - 1):
        part = idx[starts[r]: starts[r + 1]]
        assert (part >= cuts[r]).all() and (part < cuts[r + 1]).all(), r
    # what the wavefronts execute (columns per lane x steps, sixteen slots each) against the list's own
    executed = (16 * pan.astype(np.int64) * 8 * (maxs.astype(np.int64) + 7)).sum()
    cells = (ext["q_len"].astype(np.int64) * ext["s_len"]).sum()
    return executed / cells, distinct.mean(), real.mean()


def on_device(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).to("cuda:0")


CASES = {
    # name: (queries, windows per query, query length, merged share)
    "headline_32_windows_of_150": (3000, lambda r: 32, lambda r: 150, 0.0),
    "a_dozen_windows_ragged": (4000, lambda r: 1 + r.poisson(11), lambda r: int(np.clip(r.lognormal(np.log(300), 0.6), 50, 2000)), 0.1),
    "one_or_two_windows": (5000, lambda r: 1 + (r.random() < 0.4), lambda r: int(r.integers(60, 400)), 0.1),
    "a_query_with_thousands": (40, lambda r: 3000 if r.random() < 0.1 else 5, lambda r: int(r.integers(100, 300)), 0.1),
    "single_window": (1, lambda r: 1, lambda r: 100, 0.0),
}


@pytest.mark.parametrize("case", list(CASES))
def test_every_slot_of_the_device_plan(handle, case):
    nq, windows, qlen, merged = CASES[case]
    rng = np.random.default_rng(hash(case) % (1 << 31))
    ext = make_list(rng, nq, windows, qlen, merged)
    n = len(ext)
    plan, pan, maxs, rep = handle.plan_free_packing_dev(on_device(ext), n, nq)
    ratio, distinct, real = check_plan(ext, plan, pan, maxs, rep, [0, n])
    print(f"{case}: {n} windows in {len(plan)} wavefronts, executed / list cells {ratio:.3f}, {distinct:.2f} queries per wavefront, {real:.3f} of the slots real")
    if case == "headline_32_windows_of_150":
        assert real > 0.95 and ratio < 1.2  # (152 columns for 150, 183 steps for 176 rows; the clipped windows swept beside their likes)
    if case == "a_dozen_windows_ragged":
        assert real > 0.90


def test_ranges_are_contiguous_pieces_and_strip_widths(handle):
    rng = np.random.default_rng(77)
    nq = 3000
    ext = make_list(rng, nq, lambda r: 1 + r.poisson(7), lambda r: int(np.clip(r.lognormal(np.log(250), 0.5), 40, 1500)), 0.1)
    n = len(ext)
    # cuts where the query changes, near the thirds
    change = np.nonzero(ext["q_off"][1:] != ext["q_off"][:-1])[0] + 1
    cuts = [0] + [int(change[np.searchsorted(change, n * k // 3)]) for k in (1, 2)] + [n]
    for C in (19, 13, 11):
        plan, pan, maxs, rep = handle.plan_free_packing_dev(on_device(ext), n, nq, strip_cols=C, cuts=cuts)
        check_plan(ext, plan, pan, maxs, rep, cuts, C)
    # eight ranges, some of a single query
    cuts8 = [0] + [int(c) for c in change[[0, 1, 5, 100, 1000, 2000, 2500]]] + [n]
    plan, pan, maxs, rep = handle.plan_free_packing_dev(on_device(ext), n, nq, cuts=cuts8)
    check_plan(ext, plan, pan, maxs, rep, cuts8)

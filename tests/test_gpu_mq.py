"""GPU parity of the multi-query single sweep (lambda_amd/csrc/lx_sweep_mq.hip) through lx_extend_batch_dev: query runs of
4 / 8 / 16 slots (four / two / one query per wavefront), byte profiles, every strip geometry, one and several panels, ragged
window lengths -- scores, end / begin cells and ops bit-exact against the oracle.  The seam is _performAlignment,
/root/reference/src/search_algo.hpp:1070-1134 (called at :1246 and :1296)."""
import numpy as np
import pytest

from lambda_amd import capi, synth
from tests import oracle_lib
from tests.test_gpu_score import SCHEMES

pytestmark = pytest.mark.gpu


def pack_runs(ext, run, rng=None):
    """Slots for LX_OPT_QUERY_RUN = run: every query's windows sorted by length and cut into sub-blocks of `run`, the last one
    filled with copies of its last window (what lx_extend_batch does, and what the reference does with its SIMD batches,
    src/search_algo.hpp:1063-1067); sub-blocks ordered by (query length class, longest window).  Returns (slots, src) with
    src[k] = index into ext, -1 for a filler."""
    order = np.lexsort((ext["s_len"], ext["q_len"], ext["q_off"]))
    blocks = []
    k = 0
    while k < len(order):
        kk = k
        while kk < len(order) and ext["q_off"][order[kk]] == ext["q_off"][order[k]] and ext["q_len"][order[kk]] == ext["q_len"][order[k]]:
            kk += 1
        for j in range(k, kk, run):
            idx = list(order[j:min(kk, j + run)])
            src = idx + [-1] * (run - len(idx))
            idx = idx + [idx[-1]] * (run - len(idx))
            blocks.append((int(ext["s_len"][idx].max()), idx, src))
        k = kk
    blocks.sort(key=lambda b: b[0])
    slots = np.concatenate([ext[b[1]] for b in blocks])
    src = np.concatenate([np.array(b[2]) for b in blocks])
    return slots, src


def run_fused(handle, q, s, slots, run, cutoff, mq=2):
    import torch

    n = len(slots)
    dev = torch.device("cuda:0")
    pad = np.zeros(256, np.uint8)
    d_q = torch.from_numpy(np.concatenate([q, pad])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, pad])).to(dev)
    d_ext = torch.from_numpy(slots.view(np.uint8).copy()).to(dev)
    sizes = slots["q_len"].astype(np.uint64) + slots["s_len"].astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)[:-1]
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
    d_hsp = torch.full((n * 48,), 0xEE, dtype=torch.uint8, device=dev)
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    handle.set_option(capi.LX_OPT_MAX_QLEN, int(slots["q_len"].max()))
    handle.set_option(capi.LX_OPT_MAX_SLEN, int(slots["s_len"].max()))
    handle.set_option(capi.LX_OPT_QUERY_RUN, run)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    handle.set_option(capi.LX_OPT_MQ_SWEEP, mq)
    torch.cuda.synchronize()
    try:
        handle.extend_batch_dev(d_q, d_s, d_ext, n, cutoff, d_score, d_hsp, d_ops, d_off, d_count)
        handle.synchronize()
        name = handle.last_trace_kernel_name()
    finally:
        handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
        handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        handle.set_option(capi.LX_OPT_MQ_SWEEP, 1)
    hsp = np.frombuffer(d_hsp.cpu().numpy().tobytes(), dtype=capi.HSP_DTYPE)
    return d_score.cpu().numpy(), hsp, d_ops.cpu().numpy(), off, d_count.cpu().numpy(), name


def check_against_oracle(oracle, osc, q, s, slots, cutoff, got_score, hsp, ops, off, cnt):
    want_score = oracle.score_batch(q, s, slots, osc, threads=8)
    bad = np.nonzero(got_score != want_score)[0]
    assert len(bad) == 0, (bad[:10], got_score[bad[:10]], want_score[bad[:10]], slots[bad[:10]])
    surv = np.nonzero(want_score >= cutoff)[0]
    assert cnt[1] == len(surv) and len(surv) > 5
    rejected = np.setdiff1d(np.arange(len(slots)), surv)
    assert (hsp["score"][rejected] == want_score[rejected]).all() and (hsp["n_ops"][rejected] == 0).all()
    for i, (oh, oops) in zip(surv, oracle.align_batch(q, s, slots[surv], osc)):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), (i, slots[i])
        st = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[st: st + oh.n_ops]) == oops, (i, slots[i])


@pytest.mark.parametrize("run", [4, 8, 16])
@pytest.mark.parametrize("lq_range,expect", [((30, 88), "sweep_mq_kernel<11,false,false>"), ((89, 104), "sweep_mq_kernel<13,false,false>"),
                                             ((105, 152), "sweep_mq_kernel<19,false,false>"), ((153, 176), "sweep_mq_kernel<11,true,false>"),
                                             ((177, 208), "sweep_mq_kernel<13,true,false>"), ((265, 304), "sweep_mq_kernel<19,true,false>"),
                                             ((313, 352), "sweep_mq_kernel<11,true,false>"), ((417, 440), "sweep_mq_kernel<19,true,false>"),
                                             ((60, 456), "sweep_mq_kernel<19,true,false>")])
def test_mq_sweep_ragged_lists(handle, oracle, run, lq_range, expect):
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(40 if lq_range[1] <= 208 else 16, seed=77 + run + lq_range[0], lq_range=lq_range, mean_windows=5.0,
                                           merged_frac=0.15)
    slots, src = pack_runs(ext, run)
    cutoff = 60
    got = run_fused(handle, q, s, slots, run, cutoff)
    assert expect in got[5], got[5]
    assert f"{16 // run} queries per wavefront" in got[5], got[5]
    check_against_oracle(oracle, osc, q, s, slots, cutoff, *got[:5])


@pytest.mark.parametrize("scheme", ["nucl", "bs_fwd", "bs_rev", "blosum45", "blosum80"])
def test_mq_sweep_other_schemes(handle, oracle, scheme):
    sc_p = SCHEMES[scheme] if scheme in SCHEMES else capi.builtin_scoring(int(scheme[6:]), gap_open=-11, gap_extend=-1)
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    nuc = scheme in ("nucl", "bs_fwd", "bs_rev")
    alphabet = np.arange(4, dtype=np.uint8) if nuc else synth.STD20
    q, s, ext = synth.make_ragged_lists_np(40, seed=5 + len(scheme), alphabet=alphabet, lq_range=(60, 260), mean_windows=5.0, merged_frac=0.15,
                                           sub_rate=0.1 if nuc else 0.25)
    slots, src = pack_runs(ext, 4)
    cutoff = 40
    got = run_fused(handle, q, s, slots, 4, cutoff)
    assert "sweep_mq_kernel" in got[5], got[5]
    check_against_oracle(oracle, osc, q, s, slots, cutoff, *got[:5])


def test_mq_sweep_runs_of_24(handle, oracle):
    """A run that is a multiple of 8 but not of 16: wavefronts straddle two queries, eight windows each (found by the plan
    table test: the first version promised the kernel one query per wavefront there)."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(10, seed=4, lq_range=(210, 300), mean_windows=20.0, merged_frac=0.1)
    slots, src = pack_runs(ext, 24)
    got = run_fused(handle, q, s, slots, 24, 60, mq=1)
    assert "sweep_mq_kernel" in got[5] and "2 queries per wavefront" in got[5], got[5]
    check_against_oracle(oracle, osc, q, s, slots, 60, *got[:5])


def test_mq_sweep_declined_extensions_go_to_the_int32_launch(handle, oracle):
    """Tryptophan-rich wide queries: best scores beyond the compact codes' 2046 -- the sweep leaves the sentinel for exactly
    those extensions (or declines the whole wavefront when the bound fails up front) and the int32 launch redoes them into
    overflow slots; the neighbours in the same wavefront keep their compact slots."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    rng = np.random.default_rng(3)
    q, s, ext = synth.make_ragged_lists_np(12, seed=11, lq_range=(230, 300), mean_windows=5.0, merged_frac=0.1)
    # every third query becomes W-only, its homologous windows too (score 11 per column: 2 500 - 3 300)
    starts = np.unique(ext["q_off"])
    for k, qo in enumerate(starts):
        if k % 3 == 0:
            sel = np.nonzero(ext["q_off"] == qo)[0]
            L = int(ext["q_len"][sel[0]])
            q[qo:qo + L] = 22
            for i in sel[::2]:
                s[ext["s_off"][i]: ext["s_off"][i] + ext["s_len"][i]] = 22
    slots, src = pack_runs(ext, 4)
    got = run_fused(handle, q, s, slots, 4, 60)
    assert "int32 fix-up" in got[5], got[5]
    assert got[0].max() > 2046
    check_against_oracle(oracle, osc, q, s, slots, 60, *got[:5])


def test_mq_sweep_rejects_mixed_queries_in_a_sub_block(handle):
    import torch

    handle.set_scoring(SCHEMES["blosum62"], 0)
    q, s, ext = synth.make_batch_np(8, 120, 4, seed=2)
    ext = ext.copy()
    ext["q_off"][2] = ext["q_off"][5]  # second half of the first sub-block names another query
    with pytest.raises(capi.LambdaExtError):
        run_fused(handle, q, s, ext, 4, 50)


@pytest.mark.parametrize("seed,lq_range,merged,nq,chunk", [(1, (20, 330), 0.3, 40, 0), (2, (100, 620), 0.2, 40, 0), (3, (30, 110), 0.0, 40, 0),
                                                           (4, (140, 160), 0.5, 40, 0), (5, (40, 420), 0.2, 500, 1024), (6, (40, 420), 0.1, 500, 1500)])
def test_host_plan_on_ragged_lists(handle, oracle, seed, lq_range, merged, nq, chunk):
    """lx_extend_batch on lists as lambda hands them over (mixed query lengths, few windows per query, merged windows): the
    multi-query plan -- sub-blocks sorted across queries, many small chunks or one chunk spanning several panel counts (whose longest window is not
    its first sub-block's: found by tools/stress_parity.py), records gathered and scores scattered on the device -- against the
    oracle, in the caller's order; random list order included."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(nq, seed=100 + seed, lq_range=lq_range, mean_windows=4.0, merged_frac=merged)
    rng = np.random.default_rng(seed)
    if seed % 2 == 0:
        ext = ext[rng.permutation(len(ext))]
    ext = ext.copy()
    ext["s_len"][::17] = 0  # a few dead extensions: score 0, no record
    want = oracle.score_batch(q, s, ext, osc, threads=8)
    cutoff = 55
    mins = np.where(np.arange(len(ext)) % 5 == 0, cutoff + 40, cutoff).astype(np.int32)  # per-extension cut-offs
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)  # (the session's handle: other tests leave mode 1 behind)
    handle.set_option(capi.LX_OPT_EXTEND_CHUNK, chunk)  # small chunks: the pipeline's two lanes, several chunks per panel count
    try:
        score, hsp, off, ops = handle.extend_batch(q, s, ext, mins)
    finally:
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 0)
    assert "sweep_mq_kernel" in handle.last_trace_kernel_name()
    assert (score == want).all()
    surv = np.nonzero((want >= mins) & (ext["s_len"] > 0))[0]
    assert len(surv) > 10
    dead = np.setdiff1d(np.arange(len(ext)), surv)
    assert (hsp["n_ops"][dead] == 0).all() and (hsp["score"][dead] == want[dead]).all()
    for i, (oh, oops) in zip(surv, oracle.align_batch(q, s, ext[surv], osc)):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), (i, ext[i])
        st = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[st: st + oh.n_ops]) == oops, (i, ext[i])


def pack_free(ext, rng, max_queries=4):
    """Slots for LX_OPT_QUERY_RUN = 2, the free packing: the two windows of a lane group (slots 2k, 2k + 1) share a query, a
    wavefront's 16 slots hold windows of at most four queries in any split; an odd window gets a copy as its partner, a wavefront
    that meets a fifth query is closed with copies of its last window (src -1 = filler)."""
    order = np.lexsort((ext["s_len"], ext["q_len"], ext["q_off"]))
    slots, src = [], []
    cur = []  # queries of the open wavefront

    def close():
        while len(slots) % 16:
            slots.append(slots[-1])
            src.append(-1)
        cur.clear()

    k = 0
    while k < len(order):
        kk = k
        while kk < len(order) and ext["q_off"][order[kk]] == ext["q_off"][order[k]] and ext["q_len"][order[kk]] == ext["q_len"][order[k]]:
            kk += 1
        key = (int(ext["q_off"][order[k]]), int(ext["q_len"][order[k]]))
        take = int(rng.integers(1, 14))  # (cut runs at random places: a query may return in a later wavefront)
        for j in range(k, kk, 2):
            if len(slots) % 16 == 0:
                cur.clear()
            if key not in cur:
                if len(cur) == max_queries:
                    close()
                cur.append(key)
            pair = list(order[j:min(kk, j + 2)])
            slots += [ext[i] for i in pair] + [ext[pair[-1]]] * (2 - len(pair))
            src += [int(i) for i in pair] + [-1] * (2 - len(pair))
            take -= 1
            if take == 0 and rng.random() < 0.3:
                close()
        k = kk
    close()
    return np.array(slots, dtype=ext.dtype), np.array(src)


@pytest.mark.parametrize("lq_range,expect", [((30, 88), "sweep_mq_kernel<11,false,false>"), ((105, 152), "sweep_mq_kernel<19,false,false>"),
                                             ((177, 208), "sweep_mq_kernel<13,true,false>"), ((60, 456), "sweep_mq_kernel<19,true,false>")])
def test_mq_sweep_free_packing(handle, oracle, lq_range, expect):
    """LX_OPT_QUERY_RUN = 2: pairs of one query, up to four queries per wavefront in any split of its eight lane groups (5 + 2 + 1,
    7 + 1, ...), profile slots in order of appearance -- what lx_extend_batch's plan streams the windows of a ragged list into."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(60 if lq_range[1] <= 208 else 24, seed=31 + lq_range[0], lq_range=lq_range, mean_windows=4.0, merged_frac=0.15)
    slots, src = pack_free(ext, np.random.default_rng(lq_range[1]))
    per_wave = [len({(int(x["q_off"]), int(x["q_len"])) for x in slots[w: w + 16]}) for w in range(0, len(slots), 16)]
    assert max(per_wave) == 4 and min(per_wave) >= 1 and len(set(per_wave)) >= 3  # one to four queries per wavefront
    cutoff = 60
    got = run_fused(handle, q, s, slots, 2, cutoff, mq=1)
    assert expect in got[5] and "free packing: up to 4 queries per wavefront" in got[5], got[5]
    # (fillers are copies: they score like their originals; the check is per slot)
    check_against_oracle(oracle, osc, q, s, slots, cutoff, *got[:5])


def test_mq_sweep_free_packing_declined_and_broken_promises(handle, oracle):
    """Free packing: extensions beyond the compact codes go to the int32 launch, which shares its profiles by pairs; a fifth query in
    a wavefront or a lane group with two queries is a broken promise (LX_ESTATE)."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(12, seed=11, lq_range=(230, 300), mean_windows=5.0, merged_frac=0.1)
    starts = np.unique(ext["q_off"])
    for k, qo in enumerate(starts):
        if k % 3 == 0:
            sel = np.nonzero(ext["q_off"] == qo)[0]
            L = int(ext["q_len"][sel[0]])
            q[qo:qo + L] = 22
            for i in sel[::2]:
                s[ext["s_off"][i]: ext["s_off"][i] + ext["s_len"][i]] = 22
    slots, src = pack_free(ext, np.random.default_rng(1))
    got = run_fused(handle, q, s, slots, 2, 60, mq=1)
    assert "int32 fix-up" in got[5] and got[0].max() > 2046
    check_against_oracle(oracle, osc, q, s, slots, 60, *got[:5])
    q2, s2, e2 = synth.make_batch_np(8, 120, 2, seed=2)  # eight queries, one pair each: five queries in the first wavefront
    with pytest.raises(capi.LambdaExtError):
        run_fused(handle, q2, s2, e2, 2, 50, mq=1)
    e3 = np.concatenate([e2[:4], e2[:4], e2[:4], e2[:4]])
    e3[1] = e2[5]  # the first lane group names two queries
    with pytest.raises(capi.LambdaExtError):
        run_fused(handle, q2, s2, e3, 2, 50, mq=1)


@pytest.mark.parametrize("lq_range", [(33, 40), (70, 80), (100, 120), (153, 160), (185, 192), (225, 232), (262, 272), (305, 312), (340, 344), (381, 384), (420, 424)])
def test_mq_sweep_narrow_last_panel(handle, oracle, lq_range):
    """A query's last panel runs the narrowest strips that cover what is left of it -- 19, 15, 10 or 5 columns per lane at 152-column
    panels (lx_device.h: narrow_code_for; the 15-column width since round 6) --, recorded in the end cell's flags for the backtrace.
    Query lengths that leave 1-40 / 41-80 / 81-120 / 121-152 columns for the last of one, two and three panels; all queries of a list
    in one width class, so that whole
    wavefronts take the narrow path (a wavefront runs the widest strips any of its queries needs)."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(24, seed=500 + lq_range[0], lq_range=lq_range, mean_windows=6.0, merged_frac=0.15)
    slots, src = pack_free(ext, np.random.default_rng(lq_range[0]))
    cutoff = 40
    got = run_fused(handle, q, s, slots, 2, cutoff, mq=1)
    assert "sweep_mq_kernel" in got[5], got[5]
    check_against_oracle(oracle, osc, q, s, slots, cutoff, *got[:5])


def _check_host_list(handle, oracle, q, s, ext, cutoff, sample=400):
    osc = oracle_lib.scoring_from(SCHEMES["blosum62"])
    want = oracle.score_batch(q, s, ext, osc, threads=8)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    try:
        score, index, hsp, off, codes = handle.extend_batch_list(q, s, ext, cutoff)
        name = handle.last_trace_kernel_name()
    finally:
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
    assert (score == want).all()
    live = np.nonzero((want >= cutoff) & (ext["s_len"] > 0) & (ext["q_len"] > 0))[0]
    assert len(index) == len(live) and (np.sort(index) == live).all()
    rng = np.random.default_rng(0)
    pick = rng.choice(len(index), min(sample, len(index)), replace=False)
    for k, (oh, oops) in zip(pick, oracle.align_batch(q, s, ext[index[pick]], osc)):
        g = hsp[k]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), (k, ext[index[k]])
        n_ops, c, done = int(g["n_ops"]), int(off[k]), 0
        while done < n_ops:
            done += (int(codes[c]) & 63) + 1
            c += 1
        assert capi.Handle.expand_ops(codes[int(off[k]): c], n_ops) == oops
    return name


def test_host_plan_extreme_list_shapes(handle, oracle):
    """The pool + stream plan on lists far from the average: one query with thousands of windows of every length beside hundreds
    of queries with a single window (wavefronts closed by a fifth query after four pairs), queries of 1-12 residues (one narrow
    panel, windows shorter than the lane skew), and a list where every window is a merged one."""
    handle.set_scoring(SCHEMES["blosum62"], 0)
    rng = np.random.default_rng(42)
    # (a) one huge run + many singletons
    q1, s1, e1 = synth.make_ragged_lists_np(1, seed=1, lq_range=(180, 181), mean_windows=3000.0, merged_frac=0.3)
    q2, s2, e2 = synth.make_ragged_lists_np(400, seed=2, lq_range=(30, 330), mean_windows=1.0, merged_frac=0.0)
    e2 = e2.copy()
    e2["q_off"] += len(q1)
    e2["s_off"] += len(s1)
    q, s, ext = np.concatenate([q1, q2]), np.concatenate([s1, s2]), np.concatenate([e1, e2])
    assert len(e1) > 500
    name = _check_host_list(handle, oracle, q, s, ext, 50)
    assert "sweep_mq_kernel" in name and "free packing" in name
    # (b) tiny queries
    q, s, ext = synth.make_ragged_lists_np(300, seed=3, lq_range=(1, 12), mean_windows=3.0, merged_frac=0.2)
    _check_host_list(handle, oracle, q, s, ext, 8)
    # (c) every window a merged one (random lengths up to 3 Lq): everything stands in the pool or is streamed with a wide spread
    q, s, ext = synth.make_ragged_lists_np(150, seed=4, lq_range=(60, 260), mean_windows=6.0, merged_frac=1.0)
    _check_host_list(handle, oracle, q, s, ext, 45)


def test_host_plan_with_a_window_beyond_the_checkpoint_rows(handle, oracle):
    """A 70 000-residue window inside a ragged list: its chunk leaves the sweep (the checkpoint slots address 65 535 rows) and is
    traced through the direction bits, the other chunks stay on the multi-query sweep; results as ever."""
    handle.set_scoring(SCHEMES["blosum62"], 0)
    rng = np.random.default_rng(5)
    q, s, ext = synth.make_ragged_lists_np(80, seed=6, lq_range=(60, 200), mean_windows=4.0, merged_frac=0.1)
    long_s = synth.STD20[rng.integers(0, 20, 70_000)].astype(np.uint8)
    L = int(ext["q_len"][0])
    long_s[40_000: 40_000 + L] = q[int(ext["q_off"][0]): int(ext["q_off"][0]) + L]
    ext = np.concatenate([ext[:1], ext]).copy()
    ext[0]["s_off"], ext[0]["s_len"] = len(s), len(long_s)
    s = np.concatenate([s, long_s])
    handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 1024)
    try:
        _check_host_list(handle, oracle, q, s, ext, 50, sample=100)
    finally:
        handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 0)


@pytest.mark.parametrize("scheme", ["nucl", "bs_fwd", "bs_rev"])
@pytest.mark.parametrize("lq_range,expect", [((30, 88), "sweep_mq_kernel<11,false,false>"), ((105, 152), "sweep_mq_kernel<19,false,false>"), ((140, 160), "sweep_mq_kernel"),
                                             ((177, 208), "sweep_mq_kernel<13,true,false>"), ((60, 456), "sweep_mq_kernel<19,true,false>")])
def test_mq_sweep_solo_packing(handle, oracle, scheme, lq_range, expect):
    """LX_OPT_QUERY_RUN = 1, the solo packing: no promise at all -- every window has its query and its byte profile, 16 windows of
    up to 16 queries per wavefront in ANY order (lx_sweep_mq.hip); for the alphabets whose 16 profiles fit a wavefront's LDS share
    (nucleotides, bisulfite).  Widths of one to three panels, narrow last panels, the two halves of a lane group in different
    width classes; the list in random order (what a seed list of single-window reads looks like after sorting by length)."""
    sc_p = SCHEMES[scheme]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(60 if lq_range[1] <= 208 else 24, seed=900 + lq_range[0], alphabet=np.arange(4, dtype=np.uint8), lq_range=lq_range,
                                           mean_windows=3.0, merged_frac=0.15, sub_rate=0.1)
    rng = np.random.default_rng(lq_range[1])
    slots = ext[rng.permutation(len(ext))]
    per_wave = [len({(int(x["q_off"]), int(x["q_len"])) for x in slots[w: w + 16]}) for w in range(0, len(slots), 16)]
    assert max(per_wave) > 8  # far beyond the four queries of the free packing
    cutoff = 40
    got = run_fused(handle, q, s, slots, 1, cutoff, mq=1)
    assert expect in got[5] and "solo packing: up to 16 queries per wavefront" in got[5], got[5]
    check_against_oracle(oracle, osc, q, s, slots, cutoff, *got[:5])


def test_mq_sweep_solo_packing_declined_extensions(handle, oracle):
    """Solo packing, windows beyond what the sweep's range test admits (long merged windows: |ge| x rows alone passes 2046) and scores
    beyond the compact codes in several panels: the int32 launch redoes them, every window with a profile of its own."""
    sc_p = SCHEMES["nucl"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(30, seed=41, alphabet=np.arange(4, dtype=np.uint8), lq_range=(120, 700), mean_windows=3.0, merged_frac=0.2, sub_rate=0.05)
    # a few windows of 1 200 rows (the a-priori bound of their wavefronts exceeds the codes) and self-hits of the longest queries
    # (scores beyond 2046 / 2 per match = queries beyond 1 023 columns do not occur here; 700 x 2 = 1 400 stays inside)
    big = np.argsort(ext["q_len"])[-6:]
    extra = ext[big].copy()
    extra["s_len"] = np.minimum(1200, len(s) - extra["s_off"].astype(np.int64)).astype(np.uint32)
    slots = np.concatenate([ext, extra])
    slots = slots[np.random.default_rng(3).permutation(len(slots))]
    got = run_fused(handle, q, s, slots, 1, 40, mq=1)
    assert "solo packing" in got[5] and "int32 fix-up" in got[5], got[5]
    check_against_oracle(oracle, osc, q, s, slots, 40, *got[:5])


@pytest.mark.parametrize("scheme", ["nucl", "bs_rev"])
def test_host_plan_solo_on_read_lists(handle, oracle, scheme):
    """lx_extend_batch_list on a list as `searchn` produces it -- one or two windows per read, some merged -- takes the solo plan
    (all windows sorted by width and length, 16 to a wavefront): scores of every window and the survivors' records against the
    oracle; padding far below the free packing's."""
    sc_p = SCHEMES[scheme]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(3000, seed=8, alphabet=np.arange(4, dtype=np.uint8), lq_range=(100, 151), mean_windows=1.3, merged_frac=0.1, sub_rate=0.05)
    ext = ext[np.random.default_rng(2).permutation(len(ext))]
    want = oracle.score_batch(q, s, ext, osc, threads=8)
    cutoff = 60
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 1024)  # several chunks
    try:
        score, index, hsp, off, codes = handle.extend_batch_list(q, s, ext, cutoff)
        name = handle.last_trace_kernel_name()
        st = handle.last_extend_stats()
    finally:
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 0)
    assert "solo packing" in name, name
    assert st[1] < 1.02 * len(ext) + 16  # slots: the windows and the last wavefront's fillers
    assert (score == want).all()
    live = np.nonzero(want >= cutoff)[0]
    assert len(index) == len(live) and (np.sort(index) == live).all() and len(live) > 100
    pick = np.random.default_rng(0).choice(len(index), 300, replace=False)
    for k, (oh, oops) in zip(pick, oracle.align_batch(q, s, ext[index[pick]], osc)):
        g = hsp[k]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops)
        n_ops, c, done = int(g["n_ops"]), int(off[k]), 0
        while done < n_ops:
            done += (int(codes[c]) & 63) + 1
            c += 1
        assert capi.Handle.expand_ops(codes[int(off[k]): c], n_ops) == oops


def test_host_plan_wide_slots_for_long_strong_hits(handle, oracle):
    """Long queries whose homologous windows score beyond the compact codes' 2046 (500-800 residues, half of the windows):
    the first call learns it from the sweep's own count and its later chunks / the next calls run the WIDE form of the multi-query
    sweep -- int16-pair slots, no int32 launch for these windows --; scores, end cells and ops stay the oracle's in both forms, and a
    list of ordinary windows goes back to the codes."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    q, s, ext = synth.make_ragged_lists_np(400, seed=5, lq_range=(500, 800), mean_windows=8.0)
    handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 1024)  # several chunks per call
    try:
        names = []
        for rep in range(2):
            names.append(_check_host_list(handle, oracle, q, s, ext, 91, sample=150))
        assert "sweep_mq_kernel<19,true,true>" in names[1], names
        score = handle.extend_batch_list(q, s, ext, 91)[0]
        assert (score > 2046).mean() > 0.2 and score.max() > 2500
        # ordinary windows again: two calls later the codes are back
        q2, s2, e2 = synth.make_ragged_lists_np(200, seed=6, lq_range=(160, 400), mean_windows=6.0)
        back = [_check_host_list(handle, oracle, q2, s2, e2, 60, sample=100) for _ in range(2)]
        assert "sweep_mq_kernel<19,true,false>" in back[1], back
    finally:
        handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 0)


def test_overflow_area_exhausted_runs_the_chunk_again_with_wide_slots(handle, oracle):
    """A tight slot budget and a list whose windows mostly score beyond the compact codes: the overflow area (sized for an eighth of a
    chunk) runs out of int16-pair slots, and the chunk is run again with int16 pairs from the sweep itself instead of failing the
    call (round 3: LX_EOVERFLOW)."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    q2, s2, e2 = synth.make_ragged_lists_np(200, seed=6, lq_range=(160, 400), mean_windows=6.0)
    for _ in range(2):  # (the handle forgets what earlier tests taught it about strong hits)
        _check_host_list(handle, oracle, q2, s2, e2, 60, sample=50)
    q, s, ext = synth.make_ragged_lists_np(300, seed=15, lq_range=(500, 700), mean_windows=8.0)
    q[:] = 22  # tryptophan everywhere: every window scores 11 x min(Lq, Ls) > 2046
    s[:] = 22
    handle.set_option(capi.LX_OPT_TRACE_BYTES, 600 << 20)
    try:
        name = _check_host_list(handle, oracle, q, s, ext, 91, sample=60)
    finally:
        handle.set_option(capi.LX_OPT_TRACE_BYTES, 64 << 30)
    # (the chunk that ran out is run again -- with int16 pairs from the sweep where they fit the budget, else on the per-survivor path)
    assert "sweep_mq_kernel<19,true,false>" not in name, name


@pytest.mark.parametrize("run,mq,lq_range", [(4, 2, (100, 150)), (16, 2, (100, 150)), (8, 2, (280, 330)), (1, 1, (90, 150))])
def test_mq_sweep_ties_two_letter_alphabet(handle, oracle, run, mq, lq_range):
    """Two-letter sequences through the multi-query sweep: a strip's best value is reached in many rows -- of one block of sixteen steps and
    of several --, so the block-wise bookkeeping of the sweep (first block that reached the best, "met again in a later block") and the
    backtrace's search of the block's rows (lowest column, then lowest row; scan mode over the later blocks) decide every end cell: all
    survivors against the oracle, one and several panels, free and solo packing, windows that end inside a block."""
    sc_p = SCHEMES["nucl"]
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    rng = np.random.default_rng(77 + run + lq_range[0])
    nq, wpq = 48, 8
    lqs = rng.integers(lq_range[0], lq_range[1] + 1, nq)
    q = rng.integers(0, 2, int(lqs.sum())).astype(np.uint8)
    q_off = np.concatenate([[0], np.cumsum(lqs)[:-1]])
    lss = rng.integers(60, 2 * lq_range[1], nq * wpq)
    s = rng.integers(0, 2, int(lss.sum())).astype(np.uint8)
    s_off = np.concatenate([[0], np.cumsum(lss)[:-1]])
    ext = np.zeros(nq * wpq, dtype=capi.EXT_DTYPE)
    ext["q_off"] = np.repeat(q_off, wpq)
    ext["q_len"] = np.repeat(lqs, wpq)
    ext["s_off"] = s_off
    ext["s_len"] = lss
    # a piece of the query in every third window: strong diagonals among the many equal-scoring alternatives
    for k in range(0, len(ext), 3):
        n = int(min(ext["q_len"][k], ext["s_len"][k]) // 2)
        s[ext["s_off"][k] + 5: ext["s_off"][k] + 5 + n] = q[ext["q_off"][k] + 3: ext["q_off"][k] + 3 + n]
    slots, _ = pack_runs(ext, run) if run > 1 else (ext[rng.permutation(len(ext))], None)
    got = run_fused(handle, q, s, slots, run, 20, mq=mq)
    assert "sweep_mq_kernel" in got[5], got[5]
    check_against_oracle(oracle, osc, q, s, slots, 20, *got[:5])

"""BASELINE.json configs[2], [3], [4] at FULL size: one device call of each workload exactly as bench.py
issues it (lambda_amd/workloads.py: scheme, cut-off from the e-value, N rate, bisulfite conversion and scoring slot), checked by
the size-independent properties of tests/test_gpu_trace.py::test_full_size_fused_step_properties and by the oracle on a sample
of 4 000 survivors.  The seam is iterateMatchesFullSimd's two passes, /root/reference/src/search_algo.hpp:1246, :1251-1283, :1296.

configs[2] searchn: 1 M reads x 150 bp x 8 windows, +2/-3, gaps 5/2, 1 % N (N vs N is a match for the match / mismatch scheme);
configs[3] searchp scale-out: one of the eight 125 000 x 200 aa x 32 device calls -- the (8,25) sweep;
configs[4] bisulfite: both directions (C->T reads on the forward matrix in slot 0, G->A reads on the reverse matrix in slot 1)."""
import ctypes as C

import numpy as np
import pytest

from lambda_amd import capi, synth, workloads
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def _cutoff(w):
    ka = capi.karlin_params(*w.karlin)
    lib = capi.load()
    adj = lib.lx_length_adjustment(w.db_length, w.lq, C.byref(ka))
    m = 1
    while lib.lx_evalue(m, w.lq - adj, w.db_length - adj, C.byref(ka)) > w.max_evalue:
        m += 1
    return m


def _run_call(handle, oracle, w, direction, n_queries, seed, expect_kernel, sample=4000, bs_rule=0):
    import torch

    dev = torch.device("cuda:0")
    sc_p = capi.builtin_scoring(direction.scoring[0], match=direction.scoring[1], mismatch=direction.scoring[2], gap_open=direction.scoring[3],
                                gap_extend=direction.scoring[4])
    handle.set_scoring(sc_p, direction.slot)
    osc = oracle_lib.scoring_from(sc_p)
    lq, wpq = w.lq, w.windows
    d_q, d_s, d_ext, ext = synth.make_batch_torch(n_queries, lq, wpq, seed, dev, alphabet=workloads.alphabet_array(w), sub_rate=w.sub_rate,
                                                  indel_rate=w.indel_rate, n_rate=w.n_rate, n_rank=w.n_rank, convert=direction.convert,
                                                  convert_rate=w.convert_rate)
    ls, b = synth.window_len(lq), synth.band_size(lq)
    # property (4): the query itself planted into its first window (for bisulfite reads: the converted read -- every column
    # is then an identity, which both matrices score as a match)
    s2 = d_s.view(n_queries * wpq, ls)
    s2[torch.arange(n_queries, device=dev) * wpq, b:b + lq] = d_q.view(n_queries, lq)
    pad = torch.zeros(256, dtype=torch.uint8, device=dev)
    d_q = torch.cat([d_q, pad])
    d_s = torch.cat([s2.reshape(-1), pad])
    n = len(ext)
    stride = (lq + ls + 3) & ~3
    d_off = torch.arange(n, dtype=torch.int64, device=dev) * stride
    d_ops = torch.zeros(n * stride + 16, dtype=torch.uint8, device=dev)
    d_hsp = torch.full((n * 48,), 0xEE, dtype=torch.uint8, device=dev)
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    cutoff = _cutoff(w)
    handle.set_option(capi.LX_OPT_MAX_QLEN, lq)
    handle.set_option(capi.LX_OPT_MAX_SLEN, ls)
    handle.set_option(capi.LX_OPT_QUERY_RUN, wpq if wpq % 8 == 0 else 0)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    handle.set_option(capi.LX_OPT_ADAPT_PERMILLE, 0)
    handle.set_option(capi.LX_OPT_TRACE_BYTES, 80 << 30)
    handle.set_option(capi.LX_OPT_BS_MATCH_RULE, bs_rule)
    torch.cuda.synchronize()
    try:
        handle.extend_batch_dev(d_q, d_s, d_ext, n, cutoff, d_score, d_hsp, d_ops, d_off, d_count, slot=direction.slot)
        handle.synchronize()
        name = handle.last_trace_kernel_name()
        assert expect_kernel in name and "single sweep" in name, name
    finally:
        handle.set_option(capi.LX_OPT_MAX_QLEN, 0)
        handle.set_option(capi.LX_OPT_MAX_SLEN, 0)
        handle.set_option(capi.LX_OPT_QUERY_RUN, 0)
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        handle.set_option(capi.LX_OPT_ADAPT_PERMILLE, 30)
        handle.set_option(capi.LX_OPT_TRACE_BYTES, 4 << 30)
        handle.set_option(capi.LX_OPT_BS_MATCH_RULE, 0)
    hsp = d_hsp.view(torch.int32).view(n, 12)  # lx_hsp: score q_begin q_end s_begin s_end n_ops matches mismatches positives opens extensions shift
    surv = d_score >= cutoff
    cnt = d_count.cpu().numpy()
    assert int(cnt[1]) == int(surv.sum()) and int(cnt[1]) > n // 3                      # (1) counts agree, nothing flagged
    assert bool((hsp[:, 0] == d_score).all())                                             # (2) traced HSPs carry the pass-1 score
    assert bool((hsp[~surv][:, 5] == 0).all())
    t = hsp[surv].long()
    gaps = t[:, 9] + t[:, 10]
    assert bool((t[:, 5] == t[:, 6] + t[:, 7] + gaps).all())                              # (3) column counts add up
    assert bool(((t[:, 2] - t[:, 1]) + (t[:, 4] - t[:, 3]) == 2 * (t[:, 6] + t[:, 7]) + gaps).all())
    assert bool((t[:, 5] > 0).all())
    first = hsp[torch.arange(n_queries, device=dev) * wpq].long()                         # (4) self-planted windows: end to end, no gaps
    if w.n_rate == 0:
        assert bool((first[:, 1] == 0).all()) and bool((first[:, 2] == lq).all()) and bool((first[:, 3] == b).all())
        assert bool((first[:, 5] == lq).all()) and bool((first[:, 9] == 0).all())
    else:  # an N in the read's first / last column scores as a match in this scheme too: still the whole read
        assert bool((first[:, 2] - first[:, 1] == lq).all()) and bool((first[:, 9] == 0).all())
    rng = np.random.default_rng(3)                                                        # (5) a sample against the oracle
    sidx = np.sort(rng.choice(np.nonzero(surv.cpu().numpy())[0], sample, replace=False))
    hs = hsp[torch.from_numpy(sidx).to(dev)].cpu().numpy()
    # (only the sampled extensions' residues go to the host)
    lo_q, lo_s = ext["q_off"][sidx], ext["s_off"][sidx]
    q_idx = torch.from_numpy((lo_q[:, None] + np.arange(lq)[None, :]).astype(np.int64)).to(dev)
    s_idx = torch.from_numpy((lo_s[:, None] + np.arange(ls)[None, :]).astype(np.int64)).to(dev)
    q_np = d_q[q_idx].cpu().numpy().reshape(-1)
    s_np = d_s[s_idx].cpu().numpy().reshape(-1)
    sub = np.zeros(sample, dtype=capi.EXT_DTYPE)
    sub["q_off"], sub["q_len"] = np.arange(sample, dtype=np.uint64) * lq, lq
    sub["s_off"], sub["s_len"] = np.arange(sample, dtype=np.uint64) * ls, ls
    want_score = oracle.score_batch(q_np, s_np, sub, osc, threads=8)
    want = oracle.align_batch(q_np, s_np, sub, osc)
    ops_all = d_ops[: n * stride].view(n, stride)[torch.from_numpy(sidx).to(dev)].cpu().numpy()
    for k, (row, (oh, oops)) in enumerate(zip(hs, want)):
        assert int(row[0]) == int(want_score[k])
        assert tuple(int(x) for x in row[:6]) == (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), sidx[k]
        st = int(row[11])
        assert bytes(ops_all[k, st: st + oh.n_ops]) == oops, sidx[k]
    return int(cnt[1])


def test_config2_searchn_full_size(handle, oracle):
    w = workloads.WORKLOADS[2]
    pl = workloads.plan(w, 1, 0)
    assert len(pl.batches) == 1 and pl.batches[0].n_queries == 1_000_000
    b = pl.batches[0]
    _run_call(handle, oracle, w, b.direction, b.n_queries, b.seed, "score_pair_kernel<8,19,true>")


def test_config3_one_device_call_full_size(handle, oracle):
    w = workloads.WORKLOADS[3]
    pl = workloads.plan(w, 1, 0)  # the whole job on one GPU: eight calls of 125 000 queries
    assert len(pl.batches) == 8 and pl.batches[3].n_queries == 125_000
    b = pl.batches[3]
    _run_call(handle, oracle, w, b.direction, b.n_queries, b.seed, "score_pair_kernel<8,25,true>")


def test_config4_bisulfite_both_slots_full_size(handle, oracle):
    w = workloads.WORKLOADS[4]
    pl = workloads.plan(w, 1, 0)
    assert [b.direction.slot for b in pl.batches] == [0, 1] and all(b.n_queries == 250_000 for b in pl.batches)
    for b in pl.batches:
        _run_call(handle, oracle, w, b.direction, b.n_queries, b.seed, "score_pair_kernel<8,19,true>", bs_rule=1)


def test_ragged_list_full_size_through_the_host_entry_point(handle, oracle):
    """The ragged seed list of `bench.py --ragged` (50 000 queries of 50-400 aa, 596 k windows, a tenth of them merged: what
    _widenAndPreprocessMatches hands over, /root/reference/src/search_algo.hpp:1136-1175) through lx_extend_batch -- the
    multi-query plan, records gathered and scores scattered on the device -- checked against (1) lx_score_batch on the same
    list, i.e. other kernels (one query per wavefront, geometry per query) for EVERY score, (2) the filter's decisions and the
    rows of the filtered-out extensions, (3) the column-count identities of every HSP, (4) the oracle on 3 000 sampled
    survivors, merged windows over-sampled."""
    sc_p = capi.builtin_scoring(62, gap_open=-11, gap_extend=-1)
    handle.set_scoring(sc_p, 0)
    osc = oracle_lib.scoring_from(sc_p)
    q, s, ext = synth.make_ragged_lists_np(50_000, seed=0x1A3BDA07)
    handle.set_subjects(s)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    try:
        score, hsp, off, ops = handle.extend_batch(q, None, ext, 91, copy_ops=False)
        name = handle.last_trace_kernel_name()
        st = handle.last_extend_stats()
        want = handle.score_batch(q, None, ext)
    finally:
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        handle.set_subjects(None)
    assert "sweep_mq_kernel" in name, name
    assert st[0] == len(ext) and st[1] >= len(ext) and st[2] == int((ext["q_len"].astype(np.uint64) * ext["s_len"]).sum())
    assert 1.0 - st[2] / st[3] < 0.45  # padded share of the executed cells (round 2: 0.57)
    assert (score == want).all()                                                          # (1)
    surv = want >= 91
    assert ((hsp["n_ops"] > 0) == surv).all() and (hsp["score"] == want).all()           # (2)
    t = hsp[surv]
    gaps = t["num_gap_opens"].astype(np.int64) + t["num_gap_extensions"]
    assert (t["n_ops"] == t["num_matches"].astype(np.int64) + t["num_mismatches"] + gaps).all()   # (3)
    assert ((t["q_end"].astype(np.int64) - t["q_begin"]) + (t["s_end"].astype(np.int64) - t["s_begin"]) ==
            2 * (t["num_matches"].astype(np.int64) + t["num_mismatches"]) + gaps).all()
    rng = np.random.default_rng(5)                                                         # (4)
    sidx = np.nonzero(surv)[0]
    merged = sidx[ext["s_len"][sidx] > ext["q_len"][sidx] + 2 * (np.sqrt(ext["q_len"][sidx]).astype(np.int64) + 1)]
    pick = np.unique(np.concatenate([rng.choice(sidx, 2000, replace=False), rng.choice(merged, min(1000, len(merged)), replace=False)]))
    for i, (oh, oops) in zip(pick, oracle.align_batch(q, s, ext[pick], osc)):
        g = hsp[i]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops), (i, ext[i])
        a = int(off[i]) + int(g["ops_shift"])
        assert bytes(ops[a: a + oh.n_ops]) == oops, (i, ext[i])

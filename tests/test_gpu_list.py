"""lx_extend_batch_list: the survivors of the filter as a list -- what the reference's filter loop leaves behind when it erases
the matches that fail the e-value / bit-score test (/root/reference/src/search_algo.hpp:1251-1283).  Must carry exactly the
rows of lx_extend_batch_rle that survive, with the same records and the same run-length codes, whatever path the list takes
through the pipeline (uniform list on the one-query-per-wavefront kernels, ragged list on the multi-query plan, several
chunks, caller order or not, dead extensions, per-extension cut-offs, no survivor at all)."""
import numpy as np
import pytest

from lambda_amd import capi, synth
from tests import oracle_lib
from tests.test_gpu_score import SCHEMES

pytestmark = pytest.mark.gpu

FIELDS = ("score", "q_begin", "q_end", "s_begin", "s_end", "n_ops", "num_matches", "num_mismatches", "num_positives", "num_gap_opens",
          "num_gap_extensions")


def codes_of(codes, start, n_ops):
    """The codes of one survivor: they end where their lengths add up to n_ops."""
    k, done = int(start), 0
    while done < n_ops:
        done += (int(codes[k]) & 63) + 1
        k += 1
    assert done == n_ops
    return bytes(codes[int(start): k])


def check_list_against_rle(handle, q, s, ext, mins, chunk, expect_kernel=None):
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    handle.set_option(capi.LX_OPT_EXTEND_CHUNK, chunk)
    try:
        # (both calls from the same start: the pipeline remembers whether its last chunks needed int16-pair slots, and a test that
        # ran before this one may have taught it so)
        handle.set_option(capi.LX_OPT_MQ_SWEEP, 1)
        score, hsp, off, codes = handle.extend_batch_rle(q, s, ext, mins)
        name_rle = handle.last_trace_kernel_name()
        handle.set_option(capi.LX_OPT_MQ_SWEEP, 1)
        score_l, index, hsp_l, off_l, codes_l = handle.extend_batch_list(q, s, ext, mins)
        name_list = handle.last_trace_kernel_name()
    finally:
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
        handle.set_option(capi.LX_OPT_EXTEND_CHUNK, 0)
    assert name_rle == name_list
    if expect_kernel:
        assert expect_kernel in name_list
    assert (score_l == score).all()
    cut = np.broadcast_to(np.asarray(mins, dtype=np.int64), score.shape)
    live = (ext["q_len"] > 0) & (ext["s_len"] > 0)
    want = np.nonzero(live & (score >= cut))[0]
    assert len(index) == len(want) and (np.sort(index) == want).all()  # every survivor once, nobody else
    for f in FIELDS:
        assert (hsp_l[f] == hsp[f][index]).all(), f
    assert (hsp_l["ops_shift"] == 0).all()
    rng = np.random.default_rng(5)
    for k in rng.choice(len(index), min(len(index), 2000), replace=False):
        i, n_ops = int(index[k]), int(hsp_l["n_ops"][k])
        assert codes_of(codes_l, off_l[k], n_ops) == codes_of(codes, off[i], n_ops), i
    return index, hsp_l


def test_list_of_a_uniform_batch_in_several_chunks(handle, oracle):
    """The headline shape (one query length, 16 windows per query) through the classic chunks: list order = caller order."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    q, s, ext = synth.make_batch_np(12_000, 150, 16, seed=17, sub_rate=0.2, indel_rate=0.03)  # 192 k extensions
    index, hsp = check_list_against_rle(handle, q, s, ext, 80, chunk=50_000)
    assert len(index) > 50_000 and (np.diff(index.astype(np.int64)) > 0).all()
    osc = oracle_lib.scoring_from(sc_p)
    some = np.random.default_rng(2).choice(len(index), 200, replace=False)
    for k, (oh, _) in zip(some, oracle.align_batch(q, s, ext[index[some]], osc)):
        g = hsp[k]
        assert (g["score"], g["q_begin"], g["q_end"], g["s_begin"], g["s_end"], g["n_ops"]) == \
               (oh.score, oh.q_begin, oh.q_end, oh.s_begin, oh.s_end, oh.n_ops)


@pytest.mark.parametrize("seed,shuffle,chunk", [(1, False, 0), (2, True, 0), (3, False, 1500), (4, True, 2048)])
def test_list_of_a_ragged_seed_list(handle, seed, shuffle, chunk):
    """Mixed query lengths, a few windows per query, merged windows: the multi-query plan (records gathered, positions
    translated on the device), dead extensions and per-extension cut-offs included."""
    handle.set_scoring(SCHEMES["blosum62"], 0)
    q, s, ext = synth.make_ragged_lists_np(600, seed=300 + seed, lq_range=(40, 420), mean_windows=5.0, merged_frac=0.2)
    rng = np.random.default_rng(seed)
    if shuffle:
        ext = ext[rng.permutation(len(ext))]
    ext = ext.copy()
    ext["s_len"][::19] = 0
    mins = np.where(np.arange(len(ext)) % 4 == 0, 95, 55).astype(np.int32)
    index, _ = check_list_against_rle(handle, q, s, ext, mins, chunk, expect_kernel="sweep_mq_kernel")
    assert len(index) > 100


@pytest.mark.parametrize("shuffle", [False, True])
def test_mid_size_ragged_list_on_several_host_threads(handle, oracle, shuffle):
    """Between 24 000 and 250 000 extensions the host's loops (validation, order, plan, unpacking) run on a few threads, one per
    24 000 extensions: a ragged list of that size -- the multi-query plan, grouped by query or in random order, dead extensions --
    gives the oracle's scores, and the list entry the rows of the run-length entry."""
    sc_p = SCHEMES["blosum62"]
    handle.set_scoring(sc_p, 0)
    q, s, ext = synth.make_ragged_lists_np(12_000, seed=77, lq_range=(40, 300), mean_windows=5.0, merged_frac=0.15)
    assert 24_000 < len(ext) < 250_000
    if shuffle:
        ext = ext[np.random.default_rng(9).permutation(len(ext))]
    ext = ext.copy()
    ext["s_len"][::23] = 0
    mins = np.where(np.arange(len(ext)) % 4 == 0, 95, 55).astype(np.int32)
    index, _ = check_list_against_rle(handle, q, s, ext, mins, 0, expect_kernel="sweep_mq_kernel")
    assert len(index) > 5_000
    want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(sc_p), threads=8)
    handle.set_option(capi.LX_OPT_PASS2_MODE, 2)
    try:
        score = handle.extend_batch_list(q, s, ext, mins)[0]
    finally:
        handle.set_option(capi.LX_OPT_PASS2_MODE, 1)
    assert (score == want).all()


def test_list_with_no_survivor_and_with_an_empty_list(handle):
    handle.set_scoring(SCHEMES["blosum62"], 0)
    q, s, ext = synth.make_batch_np(64, 120, 8, seed=3)
    score, index, hsp, off, codes = handle.extend_batch_list(q, s, ext, 1_000_000)
    assert len(index) == 0 and len(hsp) == 0 and (score > 0).any()
    score, index, hsp, off, codes = handle.extend_batch_list(q, s, ext[:0], 10)
    assert len(score) == 0 and len(index) == 0

"""CPU tests of the oracle itself (no GPU): the restatement is pinned against hand-computed cases and against an
independent O(n^3) general-gap Smith-Waterman (tests/brute.py).  Reference output is not available here
("parity unpinned", see oracle/lx_oracle.h)."""
import numpy as np
import pytest

from lambda_amd import capi, synth
from tests import brute, oracle_lib


def schemes():
    out = {}
    out["blosum62"] = capi.builtin_scoring(62, gap_open=-11, gap_extend=-1)
    out["nucl"] = capi.builtin_scoring(0, match=2, mismatch=-3, gap_open=-5, gap_extend=-2)
    out["bs_fwd"] = capi.builtin_scoring(-1, match=2, mismatch=-3, gap_open=-5, gap_extend=-2)
    out["bs_rev"] = capi.builtin_scoring(-2, match=2, mismatch=-3, gap_open=-5, gap_extend=-2)
    return out


SCHEMES = schemes()


def alphabet_of(name):
    return synth.STD20 if name == "blosum62" else np.arange(5, dtype=np.uint8)


def test_builtin_matrices_wellformed():
    for m in (62, 45, 80):
        sc = capi.builtin_scoring(m)
        M = sc.matrix_np()[:27, :27]
        assert (M == M.T).all(), f"BLOSUM{m} not symmetric"
        assert sc.gap_open == -12 and sc.gap_extend == -1  # gapOpen+gapExtend, search_algo.hpp:226-230
    M = SCHEMES["blosum62"].matrix_np()
    order = "ABCDEFGHIJKLMNOPQRSTUVWYZX*"
    r = {c: i for i, c in enumerate(order)}
    # well-known BLOSUM62 entries
    assert M[r["W"], r["W"]] == 11 and M[r["C"], r["C"]] == 9 and M[r["A"], r["A"]] == 4
    assert M[r["W"], r["C"]] == -2 and M[r["I"], r["V"]] == 3 and M[r["D"], r["E"]] == 2
    assert M[r["*"], r["*"]] == 1 and M[r["A"], r["*"]] == -4 and M[r["X"], r["X"]] == -1
    assert M[r["Y"], r["F"]] == 3 and M[r["H"], r["Y"]] == 2 and M[r["K"], r["R"]] == 2
    # bisulfite matrices, bisulfite_scoring.hpp:77-90 (SeqAn Dna5 order A,C,G,T,N; row = query)
    F = SCHEMES["bs_fwd"].matrix_np()[:5, :5]
    R = SCHEMES["bs_rev"].matrix_np()[:5, :5]
    assert F[3, 1] == 2 and F[1, 3] == -3 and F[4, 4] == -3 and F[0, 0] == 2
    assert R[0, 2] == 2 and R[2, 0] == -3 and R[4, 4] == -3 and R[3, 1] == -3
    N = SCHEMES["nucl"].matrix_np()[:5, :5]
    assert N[3, 3] == 2 and N[0, 1] == -3  # rank equality: N vs N is a match (seqan2_to_biocpp.hpp:392-393)


def test_hand_computed_cases(oracle):
    sc = oracle_lib.scoring_from(SCHEMES["nucl"])  # +2/-3, first gap char -7, further -2
    A, Cc, G, N, T = 0, 1, 2, 3, 4
    # identical sequences: 5 matches
    q = np.array([A, Cc, G, T, A], dtype=np.uint8)
    assert oracle.score(q, q, sc) == (10, 5, 5)
    # one mismatch in the middle: ACGTA vs ACTTA -> best is either 2 matches (4) or 2+(-3)+2*2=... = 4+(-3)+4 = 5
    s = np.array([A, Cc, T, T, A], dtype=np.uint8)
    assert oracle.score(q, s, sc)[0] == 5
    # gap: query ACGTACGT, subject ACGT-CGT (A deleted): 4*2 + (-7) + 3*2 = 7 < 8 (just the first 4) -> 8
    q = np.array([A, Cc, G, T, A, Cc, G, T], dtype=np.uint8)
    s = np.array([A, Cc, G, T, Cc, G, T], dtype=np.uint8)
    sco, qe, se = oracle.score(q, s, sc)
    assert sco == 8
    # long flanks make the gap worth it: 10 matches, gap, 10 matches = 40 - 7 = 33 > 20
    left = np.array([A, Cc, G, T, A, A, G, T, Cc, Cc], dtype=np.uint8)
    right = np.array([T, G, Cc, A, T, T, G, A, Cc, G], dtype=np.uint8)
    q = np.concatenate([left, [G], right]).astype(np.uint8)
    s = np.concatenate([left, right]).astype(np.uint8)
    hsp, ops = oracle.align(q, s, sc)
    assert hsp.score == 33 and ops.count(b"I") == 1 and ops.count(b"M") == 20
    assert (hsp.q_begin, hsp.q_end, hsp.s_begin, hsp.s_end) == (0, 21, 0, 20)
    # two-character gap costs go + ge = -9
    q = np.concatenate([left, [G, G], right]).astype(np.uint8)
    hsp, ops = oracle.align(q, s, sc)
    assert hsp.score == 40 - 9 and ops.count(b"I") == 2
    # no positive cell at all
    assert oracle.score(np.array([A, A], dtype=np.uint8), np.array([Cc, Cc], dtype=np.uint8), sc) == (0, 0, 0)
    # empty inputs
    assert oracle.score(np.array([], dtype=np.uint8), q, sc) == (0, 0, 0)
    assert oracle.score(q, np.array([], dtype=np.uint8), sc) == (0, 0, 0)


def test_first_maximum_is_column_major(oracle):
    sc = oracle_lib.scoring_from(SCHEMES["nucl"])
    A, Cc, G = 0, 1, 2
    # query "AA" vs subject "A C A": score 2 reachable at (q=1,s=1), (q=1,s=3), (q=2,s=1), (q=2,s=3);
    # column-major strict '>' keeps the smallest query end, then the smallest subject end.
    q = np.array([A, A], dtype=np.uint8)
    s = np.array([A, Cc, A], dtype=np.uint8)
    assert oracle.score(q, s, sc) == (2, 1, 1)
    # query "GA" vs "A C A": only query column 2 matches; first subject row wins
    q = np.array([G, A], dtype=np.uint8)
    assert oracle.score(q, s, sc) == (2, 2, 1)


@pytest.mark.parametrize("name", ["blosum62", "nucl", "bs_fwd", "bs_rev"])
def test_oracle_vs_general_gap_sw(oracle, name):
    sc_p = SCHEMES[name]
    sc = oracle_lib.scoring_from(sc_p)
    M = sc_p.matrix_np()
    alpha = alphabet_of(name)
    rng = np.random.default_rng(1234 + len(name))
    for it in range(150):
        lq = int(rng.integers(1, 26))
        ls = int(rng.integers(1, 30))
        # small alphabets make ties and gaps frequent
        sub = alpha[: int(rng.integers(2, len(alpha) + 1))]
        q = sub[rng.integers(0, len(sub), lq)].astype(np.uint8)
        if rng.random() < 0.6 and lq > 3:
            s = q.copy()
            cut = int(rng.integers(1, lq))
            s = np.concatenate([s[:cut], s[cut + int(rng.integers(0, 3)):], sub[rng.integers(0, len(sub), int(rng.integers(0, 5)))]])
            s = s[:ls] if len(s) > ls else s
            s = s.astype(np.uint8)
            if len(s) == 0:
                s = q[:1].copy()
        else:
            s = sub[rng.integers(0, len(sub), ls)].astype(np.uint8)
        H = brute.sw_general(q, s, M, sc_p.gap_open, sc_p.gap_extend)
        want = brute.best_cell_column_major(H)
        got = oracle.score(q, s, sc)
        assert got == want, (it, q, s)
        hsp, ops = oracle.align(q, s, sc)
        assert (hsp.score, hsp.q_end, hsp.s_end) == want
        if hsp.score > 0:
            val, qi, si = brute.score_of_ops(q, s, hsp.q_begin, hsp.s_begin, ops, M, sc_p.gap_open, sc_p.gap_extend)
            assert (val, qi, si) == (hsp.score, hsp.q_end, hsp.s_end)
            assert ops[0:1] == b"M" and ops[-1:] == b"M"  # a local alignment never starts or ends with a gap
            assert H[hsp.s_begin, hsp.q_begin] == 0  # trace stops where H is 0
            st = oracle.alignment_stats(q, s, hsp, ops, sc, bs_rule=int(name.startswith("bs")))
            assert st.alignment_score == hsp.score and st.alignment_length == len(ops)
            assert st.num_matches + st.num_mismatches == ops.count(b"M")
            assert st.num_gap_opens + st.num_gap_extensions == ops.count(b"I") + ops.count(b"D")


def test_traceback_prefers_diagonal_then_vertical(oracle):
    # GapsLeft: among equally good paths the walk from the end takes the diagonal first, then the vertical
    # (subject-consuming) gap, then the horizontal one -> gaps end up as far left as possible.
    sc = oracle_lib.scoring_from(SCHEMES["nucl"])
    A, Cc, G, N, T = 0, 1, 2, 3, 4
    left = [A, Cc, G, T, A, A, G, T, Cc, Cc, T, G]
    right = [G, A, T, T, Cc, A, G, T, A, Cc, Cc, A]
    # subject has the run "AAA" where the query has "AA": the deleted A can sit at any of three places
    q = np.array(left + [A, A] + right, dtype=np.uint8)
    s = np.array(left + [A, A, A] + right, dtype=np.uint8)
    hsp, ops = oracle.align(q, s, sc)
    assert hsp.score == 2 * 26 - 7
    assert ops.count(b"D") == 1
    # trailing A's are matched diagonally first, so the gap is the leftmost of the run
    assert ops.index(b"D") == len(left)


def test_simd_variant_equals_scalar(oracle):
    for name in ("blosum62", "nucl"):
        sc = oracle_lib.scoring_from(SCHEMES[name])
        q, s, ext = synth.make_ragged_np(150, seed=77, alphabet=alphabet_of(name), lq_range=(1, 120))
        a = oracle.score_batch(q, s, ext, sc, threads=2, ends=True)
        b = oracle.score_batch(q, s, ext, sc, threads=2, simd=True, ends=True)
        for x, y in zip(a, b):
            assert (x == y).all()


def test_banded_equals_full_when_band_covers_rectangle(oracle):
    sc = oracle_lib.scoring_from(SCHEMES["blosum62"])
    q, s, ext = synth.make_ragged_np(40, seed=5, lq_range=(5, 60))
    for x in ext:
        qq = q[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])]
        ss = s[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])]
        assert oracle.score_banded(qq, ss, sc, -10000, 10000) == oracle.score(qq, ss, sc)[0]
        assert oracle.score_banded(qq, ss, sc, -3, 3) <= oracle.score(qq, ss, sc)[0]


@pytest.mark.parametrize("name", ["blosum62", "nucl"])
def test_banded_oracle_equals_banded_brute_force(oracle, name):
    """Band mode (not the reference's configuration) pinned like the full rectangle: the Gotoh restatement with a band against
    the general-gap Smith-Waterman restricted to the band, on small random pairs, bands that cut through the alignment
    included; the banded traceback re-scores to the banded score, stays inside the band and equals the full one when the
    band covers the rectangle."""
    sc_p = SCHEMES[name]
    sc = oracle_lib.scoring_from(sc_p)
    M = sc_p.matrix_np()
    rng = np.random.default_rng(8)
    q, s, ext = synth.make_ragged_np(60, seed=12, alphabet=alphabet_of(name), lq_range=(3, 28), ls_extra=(0, 12))
    for x in ext:
        qq = q[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])]
        ss = s[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])]
        d0, b = int(rng.integers(-3, 8)), int(rng.integers(0, 7))
        lo, hi = d0 - b, d0 + b
        H = brute.sw_general(qq, ss, M, sc_p.gap_open, sc_p.gap_extend, band=(lo, hi))
        want = int(H.max())
        assert oracle.score_banded(qq, ss, sc, lo, hi) == want
        hsp, ops = oracle.align_banded(qq, ss, sc, lo, hi)
        assert hsp.score == want
        if want > 0:
            best, bq, bs = brute.best_cell_column_major(H)
            assert (hsp.q_end, hsp.s_end) == (bq, bs)
            got, qi, si = brute.score_of_ops(qq, ss, hsp.q_begin, hsp.s_begin, ops, M, sc_p.gap_open, sc_p.gap_extend)
            assert (got, qi, si) == (want, hsp.q_end, hsp.s_end)
            i, j = hsp.s_begin, hsp.q_begin  # every cell the alignment passes lies inside the band
            for op in ops:
                i += op in (ord("M"), ord("D"))
                j += op in (ord("M"), ord("I"))
                assert lo <= (i - 1) - (j - 1) <= hi
        full_h, full_ops = oracle.align(qq, ss, sc)
        wide_h, wide_ops = oracle.align_banded(qq, ss, sc, -1000, 1000)
        assert (wide_h.score, wide_h.q_begin, wide_h.q_end, wide_h.s_begin, wide_h.s_end, wide_ops) == \
               (full_h.score, full_h.q_begin, full_h.q_end, full_h.s_begin, full_h.s_end, full_ops)


def test_band_size(oracle):
    # src/search_misc.hpp:46-50
    for n, b in ((0, 1), (1, 2), (99, 10), (100, 11), (150, 13), (200, 15), (10 ** 6, 1001)):
        assert oracle.band_size(n) == b
        if n:
            assert synth.band_size(n) == b


def test_widen_and_preprocess(oracle):
    M = oracle_lib.MATCH_DTYPE
    qlens = np.array([100, 100], dtype=np.uint64)
    slens = np.array([1000, 150], dtype=np.uint64)
    b = 11
    # single seed in the middle: src/search_algo.hpp:919-938
    m = np.array([(0, 0, 20, 30, 500, 510)], dtype=M)
    out = oracle.widen_and_preprocess(m, qlens, slens)
    assert tuple(out[0]) == (0, 0, 0, 100, 500 - 20 - b, 500 - 20 + 100 + b)
    # clipped at both subject ends
    m = np.array([(0, 1, 50, 60, 10, 20), (0, 1, 10, 20, 120, 130)], dtype=M)
    out = oracle.widen_and_preprocess(m, qlens, slens)
    # first: subjStart 10 < 50 -> 0; end min(0+100+11,150)=111; second: start 110 -> 99, end min(110+111,150)=150;
    # they overlap (111 >= 99) -> merged to one window [0,150)
    assert len(out) == 1 and tuple(out[0]) == (0, 1, 0, 100, 0, 150)
    # two seeds on the same diagonal -> identical windows -> one
    m = np.array([(0, 0, 20, 30, 500, 510), (0, 0, 40, 50, 520, 530)], dtype=M)
    assert len(oracle.widen_and_preprocess(m, qlens, slens)) == 1
    # far apart on the same subject -> two windows, sorted
    m = np.array([(0, 0, 20, 30, 800, 810), (0, 0, 20, 30, 300, 310)], dtype=M)
    out = oracle.widen_and_preprocess(m, qlens, slens)
    assert len(out) == 2 and out[0]["subjStart"] < out[1]["subjStart"]
    # different queries never merge
    m = np.array([(0, 0, 20, 30, 500, 510), (1, 0, 20, 30, 500, 510)], dtype=M)
    assert len(oracle.widen_and_preprocess(m, qlens, slens)) == 2
    # chain of three overlapping windows collapses to one spanning window
    m = np.array([(0, 0, 0, 10, 100, 110), (0, 0, 0, 10, 180, 190), (0, 0, 0, 10, 260, 270)], dtype=M)
    out = oracle.widen_and_preprocess(m, qlens, slens)
    assert len(out) == 1 and tuple(out[0]) == (0, 0, 0, 100, 100 - b, 260 + 100 + b)


def test_seed_looks_promising(oracle):
    sc = oracle_lib.scoring_from(SCHEMES["blosum62"])
    rng = np.random.default_rng(3)
    q = synth.STD20[rng.integers(0, 20, 120)].astype(np.uint8)
    s = synth.STD20[rng.integers(0, 20, 400)].astype(np.uint8)
    s[200:260] = q[30:90]  # a perfect diagonal
    # seed on the diagonal, region pre-scoring (preScoring=2, thresh 2.0, seedLength 10; search_options.hpp:104-105)
    assert oracle.seed_looks_promising(q, s, 40, 50, 210, 10, 2, 2.0, sc)
    # same coordinates on a random diagonal
    assert not oracle.seed_looks_promising(q, s, 40, 50, 100, 10, 2, 2.0, sc)
    # seed at the very start: the region is shifted, not truncated below zero (search_algo.hpp:440-452)
    assert oracle.seed_looks_promising(q, s, 30, 40, 200, 10, 2, 2.0, sc) in (True, False)
    s2 = s.copy()
    s2[0:40] = q[0:40]
    assert oracle.seed_looks_promising(q, s2, 0, 10, 0, 10, 2, 2.0, sc)


def test_blast_statistics(oracle):
    ka = oracle_lib.Karlin(0.267, 0.041, 0.14, 1.9, -30.0)  # BLOSUM62 11/1
    db = 205_000_000
    adj = oracle.length_adjustment(db, 150, ka)
    assert 0 < adj < 150
    # fixed point property: adj ~ alpha/lambda * ln(K (m-adj)(n-adj)) + beta
    import math

    f = ka.alpha / ka.lambda_ * math.log(ka.K * (150 - adj) * (db - adj)) + ka.beta
    assert abs(f - adj) <= 1.5
    e1 = oracle.evalue(100, 150 - adj, db - adj, ka)
    e2 = oracle.evalue(101, 150 - adj, db - adj, ka)
    assert e2 < e1 and abs(e1 / e2 - math.exp(ka.lambda_)) < 1e-9
    assert abs(oracle.bitscore(100, ka) - (0.267 * 100 - math.log(0.041)) / math.log(2)) < 1e-12
    # tiny search space: c < 0 -> no adjustment
    assert oracle.length_adjustment(10, 5, ka) == 0


def test_published_known_answers(oracle):
    """Known answers from outside this repository (the reference's own goldens are not on disk):
    (1) the worked Smith-Waterman example of the textbook / Wikipedia article -- TGTTACGG vs GGTTGACTA, match +3,
        mismatch -3, linear gap penalty 2 -- whose optimal local alignment is GTT-AC / GTTGAC with score 13;
    (2) raw score / bit score pairs every BLASTP report with BLOSUM62, gaps 11/1 shows (lambda 0.267, K 0.041):
        "28.9 bits (63)", "32.0 bits (71)", "47.8 bits (112)"."""
    idx = {c: i for i, c in enumerate("ACGT")}
    enc = lambda t: np.array([idx[c] for c in t], dtype=np.uint8)
    m = np.full((4, 4), -3, dtype=np.int8)
    np.fill_diagonal(m, 3)
    sc = oracle_lib.make_scoring(4, m, -2, -2)  # linear gap penalty 2 = first character -2, every further one -2
    q, s = enc("TGTTACGG"), enc("GGTTGACTA")
    hsp, ops = oracle.align(q, s, sc)
    assert hsp.score == 13 and ops == b"MMMDMM"  # the gap is in the query row (GTT-AC)
    assert (hsp.q_begin, hsp.q_end, hsp.s_begin, hsp.s_end) == (1, 6, 1, 7)
    assert oracle.score(q, s, sc)[0] == 13 and int(brute.sw_general(q, s, m.astype(int), -2, -2).max()) == 13
    # the same with the roles swapped: the gap moves to the subject row
    hsp2, ops2 = oracle.align(s, q, sc)
    assert hsp2.score == 13 and ops2 == b"MMMIMM"

    ka = oracle_lib.Karlin(0.267, 0.041, 0.14, 1.9, -30.0)
    cka = capi.karlin_params(62, gap_open=-11, gap_extend=-1)
    assert (cka.lambda_, cka.K) == (0.267, 0.041)
    lib = capi.load()
    import ctypes as C
    lib.lx_bitscore.restype = C.c_double
    for raw, bits in ((63, 28.9), (71, 32.0), (112, 47.8)):
        assert round(oracle.bitscore(raw, ka), 1) == bits
        assert round(lib.lx_bitscore(raw, C.byref(cka)), 1) == bits


DURBIN_LETTERS = "AEGHPW"
# BLOSUM50 entries for these letters (Durbin, Eddy, Krogh, Mitchison: Biological Sequence Analysis, 1998, figure 2.2)
DURBIN_B50 = {("A", "A"): 5, ("A", "E"): -1, ("A", "G"): 0, ("A", "H"): -2, ("A", "P"): -1, ("A", "W"): -3,
              ("E", "E"): 6, ("E", "G"): -3, ("E", "H"): 0, ("E", "P"): -1, ("E", "W"): -3,
              ("G", "G"): 8, ("G", "H"): -2, ("G", "P"): -2, ("G", "W"): -3,
              ("H", "H"): 10, ("H", "P"): -2, ("H", "W"): -3,
              ("P", "P"): 10, ("P", "W"): -4, ("W", "W"): 15}


def durbin_example():
    """The worked local-alignment example of the same book (figure 2.6): HEAGAWGHEE vs PAWHEAE, BLOSUM50, linear gap
    cost 8 -- best local alignment AWGHE / AW-HE with score 28."""
    n = len(DURBIN_LETTERS)
    m = np.zeros((n, n), dtype=np.int8)
    for (a, b), v in DURBIN_B50.items():
        m[DURBIN_LETTERS.index(a), DURBIN_LETTERS.index(b)] = m[DURBIN_LETTERS.index(b), DURBIN_LETTERS.index(a)] = v
    enc = lambda t: np.array([DURBIN_LETTERS.index(c) for c in t], dtype=np.uint8)
    return m, enc("HEAGAWGHEE"), enc("PAWHEAE")


def test_published_durbin_local_alignment(oracle):
    m, q, s = durbin_example()
    sc = oracle_lib.make_scoring(len(DURBIN_LETTERS), m, -8, -8)
    hsp, ops = oracle.align(q, s, sc)
    assert hsp.score == 28 and ops == b"MMIMM"  # AWGHE / AW-HE: the gap is in the subject row
    assert (hsp.q_begin, hsp.q_end, hsp.s_begin, hsp.s_end) == (4, 9, 1, 5)
    assert int(brute.sw_general(q, s, m.astype(int), -8, -8).max()) == 28

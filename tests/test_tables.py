"""The built-in tables pinned against numbers that do not come from this repository.

Product and oracle share the scoring tables (the oracle takes them through capi.builtin_scoring), so a typo in a matrix is
invisible to every parity test.  What pins them here:

* every 20 x 20 BLOSUM core reproduces NCBI's UNGAPPED Karlin-Altschul parameters: with the Robinson & Robinson background
  frequencies (the `Robinson_prob` table of NCBI blast_stat.c) lambda is the root of  sum_ij p_i p_j exp(lambda s_ij) = 1
  and H = lambda sum_ij p_i p_j s_ij exp(lambda s_ij); blast_stat.c lists them as the first row of each matrix's value
  table: BLOSUM62 0.3176 / 0.4012, BLOSUM45 0.2291 / 0.2514, BLOSUM80 0.3430 / 0.6568.  All 400 entries enter the sums, a
  single wrong entry moves lambda in the fourth decimal;
* the gapped rows lx_karlin_params serves are compared with the rows of blast_stat.c quoted below (blosum62_values,
  blosum45_values, blosum80_values, blastn_values_2_3), and with what every BLAST report prints for BLOSUM62 11/1
  ("Lambda 0.267, K 0.0410, H 0.140");
* textbook entries of each matrix.
The letters B, Z, X, * are NCBI's, J / O / U are derived as scoring_tables.hpp says (provisional, SeqAn's are unknown)."""
import math

import numpy as np
import pytest

from lambda_amd import capi

ORDER = "ABCDEFGHIJKLMNOPQRSTUVWYZX*"
# NCBI blast_stat.c, Robinson_prob (per thousand)
ROBINSON = {"A": 78.05, "C": 19.25, "D": 53.64, "E": 62.95, "F": 38.56, "G": 73.77, "H": 21.99, "I": 51.42, "K": 57.44, "L": 90.19,
            "M": 22.43, "N": 44.87, "P": 52.03, "Q": 42.64, "R": 51.29, "S": 71.20, "T": 58.41, "V": 64.41, "W": 13.30, "Y": 32.16}
# NCBI blast_stat.c: ungapped (lambda, K, H) = first row of blosum62_values / blosum45_values / blosum80_values
UNGAPPED = {62: (0.3176, 0.134, 0.4012), 45: (0.2291, 0.0924, 0.2514), 80: (0.3430, 0.177, 0.6568)}
# NCBI blast_stat.c rows {gap open, gap extend, lambda, K, H, alpha, beta}
GAPPED = {
    62: [(11, 2, 0.297, 0.082, 0.27, 1.1, -10), (10, 2, 0.291, 0.075, 0.23, 1.3, -15), (9, 2, 0.279, 0.058, 0.19, 1.5, -19),
         (8, 2, 0.264, 0.045, 0.15, 1.8, -26), (7, 2, 0.239, 0.027, 0.10, 2.5, -46), (6, 2, 0.201, 0.012, 0.061, 3.3, -58),
         (13, 1, 0.292, 0.071, 0.23, 1.2, -11), (12, 1, 0.283, 0.059, 0.19, 1.5, -19), (11, 1, 0.267, 0.041, 0.14, 1.9, -30),
         (10, 1, 0.243, 0.024, 0.10, 2.5, -44), (9, 1, 0.206, 0.010, 0.052, 4.0, -87)],
    45: [(13, 3, 0.207, 0.049, 0.14, 1.5, -22), (12, 3, 0.199, 0.039, 0.11, 1.8, -34), (11, 3, 0.190, 0.031, 0.095, 2.0, -38),
         (10, 3, 0.179, 0.023, 0.075, 2.4, -51), (16, 2, 0.210, 0.051, 0.14, 1.5, -24), (15, 2, 0.203, 0.041, 0.12, 1.7, -31),
         (14, 2, 0.195, 0.032, 0.10, 1.9, -36), (13, 2, 0.185, 0.024, 0.084, 2.2, -45), (12, 2, 0.171, 0.016, 0.061, 2.8, -65),
         (19, 1, 0.205, 0.040, 0.11, 1.9, -43), (18, 1, 0.198, 0.032, 0.10, 2.0, -43), (17, 1, 0.189, 0.024, 0.079, 2.4, -57),
         (16, 1, 0.176, 0.016, 0.063, 2.8, -67)],
    80: [(25, 2, 0.342, 0.17, 0.66, 0.52, -1.6), (13, 2, 0.336, 0.15, 0.57, 0.59, -3), (9, 2, 0.319, 0.11, 0.42, 0.76, -6),
         (8, 2, 0.308, 0.090, 0.35, 0.89, -9), (7, 2, 0.293, 0.070, 0.27, 1.1, -14), (6, 2, 0.268, 0.045, 0.19, 1.4, -19),
         (11, 1, 0.314, 0.095, 0.35, 0.90, -9), (10, 1, 0.299, 0.071, 0.27, 1.1, -14), (9, 1, 0.279, 0.048, 0.20, 1.4, -19)],
}
BLASTN_2_3 = [(4, 4, 0.63, 0.42, 0.84, 0.75, -2), (2, 4, 0.615, 0.37, 0.72, 0.85, -3), (0, 4, 0.55, 0.21, 0.46, 1.2, -5),
              (3, 3, 0.615, 0.37, 0.68, 0.9, -3), (6, 2, 0.63, 0.42, 0.84, 0.75, -2), (5, 2, 0.625, 0.41, 0.78, 0.8, -2),
              (4, 2, 0.61, 0.35, 0.68, 0.9, -3), (2, 2, 0.515, 0.14, 0.33, 1.55, -9)]


def core20(method):
    letters = list(ROBINSON)
    idx = [ORDER.index(c) for c in letters]
    M = capi.builtin_scoring(method).matrix_np()[np.ix_(idx, idx)].astype(np.float64)
    p = np.array([ROBINSON[c] for c in letters]) / 1000.0
    return letters, M, p


@pytest.mark.parametrize("method", [62, 45, 80])
def test_blosum_core_reproduces_ncbi_ungapped_karlin_parameters(method):
    letters, M, p = core20(method)
    assert abs(p.sum() - 1.0) < 1e-12 and (M == M.T).all()
    pp = np.outer(p, p)
    f = lambda lam: float((pp * np.exp(lam * M)).sum() - 1.0)
    lo, hi = 1e-6, 2.0
    assert f(lo) < 0 < f(hi)
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        lo, hi = (lo, mid) if f(mid) > 0 else (mid, hi)
    lam = 0.5 * (lo + hi)
    H = lam * float((pp * M * np.exp(lam * M)).sum())
    want_lam, _, want_H = UNGAPPED[method]
    assert abs(lam - want_lam) < 5e-5 + 0.5e-4, (method, lam)   # the published value has four decimals
    assert abs(H - want_H) < 1e-4, (method, H)
    # a single wrong entry would show: perturb one off-diagonal pair by one and lambda leaves the published value
    M2 = M.copy()
    i, j = letters.index("L"), letters.index("I")
    M2[i, j] += 1
    M2[j, i] += 1
    g = lambda lam: float((pp * np.exp(lam * M2)).sum() - 1.0)
    lo, hi = 1e-6, 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        lo, hi = (lo, mid) if g(mid) > 0 else (mid, hi)
    assert abs(0.5 * (lo + hi) - want_lam) > 2e-4


def test_textbook_entries_of_blosum45_and_blosum80():
    r = {c: i for i, c in enumerate(ORDER)}
    M45 = capi.builtin_scoring(45).matrix_np()
    M80 = capi.builtin_scoring(80).matrix_np()
    assert M45[r["W"], r["W"]] == 15 and M45[r["C"], r["C"]] == 12 and M45[r["H"], r["H"]] == 10 and M45[r["A"], r["A"]] == 5
    assert M45[r["I"], r["V"]] == 3 and M45[r["K"], r["R"]] == 3 and M45[r["F"], r["Y"]] == 3 and M45[r["*"], r["A"]] == -5
    assert M80[r["W"], r["W"]] == 11 and M80[r["C"], r["C"]] == 9 and M80[r["H"], r["H"]] == 8 and M80[r["P"], r["P"]] == 8
    assert M80[r["I"], r["V"]] == 3 and M80[r["D"], r["W"]] == -6 and M80[r["*"], r["A"]] == -6 and M80[r["F"], r["Y"]] == 3
    for M in (M45, M80):
        assert M[r["U"], r["U"]] == M[r["C"], r["C"]] and M[r["O"], r["A"]] == M[r["K"], r["A"]]  # U := C, O := K (provisional)


@pytest.mark.parametrize("method", [62, 45, 80])
def test_gapped_karlin_rows_are_ncbis(method):
    for go, ge, lam, K, H, alpha, beta in GAPPED[method]:
        ka = capi.karlin_params(method, gap_open=-go, gap_extend=-ge)
        assert (ka.lambda_, ka.K, ka.H, ka.alpha, ka.beta) == (lam, K, H, alpha, beta), (method, go, ge)
    with pytest.raises(capi.LambdaExtError):  # combinations NCBI has no values for make prepareScoring() throw (:232-233)
        capi.karlin_params(method, gap_open=-3, gap_extend=-1)


def test_nucleotide_rows_and_the_defaults_of_lambda():
    for go, ge, lam, K, H, alpha, beta in BLASTN_2_3:
        if go == 0:
            continue  # (a gap_open of 0 is not a lambda option value)
        ka = capi.karlin_params(0, 2, -3, -go, -ge)
        assert (ka.lambda_, ka.K, ka.H, ka.alpha, ka.beta) == (lam, K, H, alpha, beta), (go, ge)
    # lambda's defaults (src/search_options.hpp:290-307): protein 11/1, nucleotide 5/2
    ka = capi.karlin_params(62, gap_open=-11, gap_extend=-1)
    assert (ka.lambda_, ka.K, ka.H) == (0.267, 0.041, 0.14)  # "Lambda 0.267  K 0.0410  H 0.140" of every BLASTP report
    kn = capi.karlin_params(0, 2, -3, -5, -2)
    assert (kn.lambda_, kn.K) == (0.625, 0.41)
    # bit score of a raw score: (lambda S - ln K) / ln 2 -- "47.8 bits (112)" of BLASTP reports
    lib = capi.load()
    import ctypes as C
    assert abs(lib.lx_bitscore(112, C.byref(ka)) - 47.8) < 0.05

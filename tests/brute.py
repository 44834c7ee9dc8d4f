"""Independent cross-check for the oracle: the original O(n^3) general-gap Smith-Waterman (Waterman-Smith-Beyer
form), which shares no recurrence with the Gotoh E/F formulation used by the oracle and the kernels.

    H[i][j] = max(0, H[i-1][j-1] + s(q_j, t_i), max_k H[i-k][j] + w(k), max_k H[i][j-k] + w(k)),  w(k) = go + (k-1)*ge
"""
import numpy as np


def sw_general(q, s, matrix, go, ge, band=None):
    """Returns the full H matrix, (ls+1) x (lq+1), rows = subject, columns = query.

    band = (lo, hi): only cells with lo <= (i - 1) - (j - 1) <= hi exist; a gap runs through cells of the band only
    (every cell it crosses and the cell it opens from lie inside)."""
    lq, ls = len(q), len(s)
    H = np.zeros((ls + 1, lq + 1), dtype=np.int64)
    lo, hi = band if band is not None else (-10 ** 9, 10 ** 9)
    for i in range(1, ls + 1):
        for j in range(1, lq + 1):
            d = i - j
            if d < lo or d > hi:
                continue
            best = H[i - 1, j - 1] + int(matrix[q[j - 1], s[i - 1]])
            for k in range(1, i + 1):
                if d - k < lo:
                    break  # the vertical gap would start off the band
                best = max(best, H[i - k, j] + go + (k - 1) * ge)
            for k in range(1, j + 1):
                if d + k > hi:
                    break
                best = max(best, H[i, j - k] + go + (k - 1) * ge)
            H[i, j] = max(0, best)
    return H


def best_cell_column_major(H):
    """First maximum in column-major order (query column outer, subject row inner), strict '>'."""
    best, bq, bs = 0, 0, 0
    ls1, lq1 = H.shape
    for j in range(1, lq1):
        for i in range(1, ls1):
            if H[i, j] > best:
                best, bq, bs = int(H[i, j]), j, i
    return best, bq, bs


def score_of_ops(q, s, qb, sb, ops, matrix, go, ge):
    """Re-derives the score of an alignment given as M/D/I ops (D consumes subject, I consumes query)."""
    qi, si, sc, prev = qb, sb, 0, None
    for op in ops:
        op = chr(op) if not isinstance(op, str) else op
        if op == "M":
            sc += int(matrix[q[qi], s[si]])
            qi += 1
            si += 1
        elif op == "D":
            sc += ge if prev == "D" else go
            si += 1
        elif op == "I":
            sc += ge if prev == "I" else go
            qi += 1
        else:
            raise ValueError(op)
        prev = op
    return sc, qi, si

"""Generates the golden vectors in this directory from the CPU oracle (oracle/lx_oracle.c).

    python tests/golden/make_golden.py

The reference itself cannot be run in this environment, so these vectors pin
"GPU == oracle" and "oracle today == oracle when the vectors were made", not "== lambda3".  A fixture is data only:
seeded synthetic inputs and the oracle's outputs.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

from lambda_amd import capi, synth  # noqa: E402
from tests import oracle_lib  # noqa: E402

OUT = Path(__file__).resolve().parent

SCHEMES = {
    "blosum62": (dict(method=62, gap_open=-11, gap_extend=-1), synth.STD20, 0),
    "nucl": (dict(method=0, match=2, mismatch=-3, gap_open=-5, gap_extend=-2), np.arange(5, dtype=np.uint8), 0),
    "bs_fwd": (dict(method=-1, match=2, mismatch=-3, gap_open=-5, gap_extend=-2), np.arange(5, dtype=np.uint8), 1),
    "bs_rev": (dict(method=-2, match=2, mismatch=-3, gap_open=-5, gap_extend=-2), np.arange(5, dtype=np.uint8), 1),
}


def build(name):
    kw, alpha, bs_rule = SCHEMES[name]
    sc_p = capi.builtin_scoring(kw.pop("method"), **kw)
    sc = oracle_lib.scoring_from(sc_p)
    orc = oracle_lib.load()
    seed = sum(map(ord, name))
    q1, s1, e1 = synth.make_ragged_np(90, seed=seed, alphabet=alpha, lq_range=(1, 230), ls_extra=(0, 60))
    # tie-heavy block: two letters only
    rng = np.random.default_rng(seed + 1)
    n2 = 40
    two = alpha[:2]
    q2 = two[rng.integers(0, 2, 40 * n2)].astype(np.uint8)
    s2 = two[rng.integers(0, 2, 56 * n2)].astype(np.uint8)
    e2 = np.zeros(n2, dtype=capi.EXT_DTYPE)
    e2["q_off"] = len(q1) + np.arange(n2) * 40
    e2["q_len"] = rng.integers(1, 41, n2)
    e2["s_off"] = len(s1) + np.arange(n2) * 56
    e2["s_len"] = rng.integers(1, 57, n2)
    q = np.concatenate([q1, q2])
    s = np.concatenate([s1, s2])
    ext = np.concatenate([e1, e2])
    score, qe, se = orc.score_batch(q, s, ext, sc, ends=True)
    hsp = np.zeros((len(ext), 6), dtype=np.int32)
    stats = np.zeros((len(ext), 5), dtype=np.int32)
    ops_parts, ops_off = [], np.zeros(len(ext) + 1, dtype=np.int64)
    for i, (h, o) in enumerate(orc.align_batch(q, s, ext, sc)):
        hsp[i] = (h.score, h.q_begin, h.q_end, h.s_begin, h.s_end, h.n_ops)
        if h.score > 0:
            x = ext[i]
            st = orc.alignment_stats(q[int(x["q_off"]): int(x["q_off"]) + int(x["q_len"])],
                                     s[int(x["s_off"]): int(x["s_off"]) + int(x["s_len"])], h, o, sc, bs_rule)
            stats[i] = (st.num_matches, st.num_mismatches, st.num_positives, st.num_gap_opens, st.num_gap_extensions)
        ops_parts.append(np.frombuffer(o, dtype=np.uint8))
        ops_off[i + 1] = ops_off[i] + len(o)
    np.savez_compressed(OUT / f"{name}.npz", q=q, s=s, ext=ext.view(np.uint8), score=score, q_end=qe, s_end=se, hsp=hsp,
                        stats=stats, ops=np.concatenate(ops_parts), ops_off=ops_off,
                        matrix=sc_p.matrix_np(), gaps=np.array([sc_p.alphabet_size, sc_p.gap_open, sc_p.gap_extend]),
                        bs_rule=np.array([bs_rule]))
    return len(ext)


if __name__ == "__main__":
    for nm in list(SCHEMES):
        print(nm, build(nm))

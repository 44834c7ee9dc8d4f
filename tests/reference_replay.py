"""Replays the rows of a BLAST-tabular file (the reference's goldens, tools/fetch_reference_goldens.py) through the
extension driver: for every row a seed pair on the reported alignment's first and last diagonal goes through
lx_iterate_matches (widen / merge, both GPU passes, statistics, filters) and lx_write_records; the row must come out of it
character for character -- coordinates, identity, length, mismatches, gap opens, e-value, bit score.

What this can and cannot show: the reference found its rows from FM-index seeds this harness does not have; a row is
reproducible here when its alignment is the best one of the window its own diagonals span, which is what the reference's
window would have been for any seed on those diagonals (src/search_algo.hpp:919-938).  Rows that are not the best of
their window (shadowed by a stronger hit of the same pair) are reported as "shadowed", not as failures, if the stronger hit
is itself a row of the golden file.
"""
from __future__ import annotations

import gzip
import math
from pathlib import Path

import numpy as np

from lambda_amd import capi

AA_ORDER = "ABCDEFGHIJKLMNOPQRSTUVWYZX*"  # SeqAn AminoAcid ranks
DNA5_BIOCPP = "ACGNT"                     # BioC++ dna5 ranks (match/mismatch-scored: src/seqan2_to_biocpp.hpp:392-393)
DNA5_SEQAN = "ACGTN"                      # SeqAn Dna5 ranks (bisulfite mode)


def read_fasta(path: Path):
    op = gzip.open if str(path).endswith(".gz") else open
    ids, seqs, cur = [], [], []
    with op(path, "rt") as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if ids:
                    seqs.append("".join(cur))
                ids.append(line[1:])
                cur = []
            else:
                cur.append(line.strip())
    if ids:
        seqs.append("".join(cur))
    return ids, seqs


def ranks(seq: str, order: str, unknown: str) -> np.ndarray:
    lut = np.full(256, order.index(unknown), dtype=np.uint8)
    for i, ch in enumerate(order):
        lut[ord(ch)] = i
        lut[ord(ch.lower())] = i
    if "T" in order:
        lut[ord("U")] = lut[ord("u")] = order.index("T")
    return lut[np.frombuffer(seq.encode(), dtype=np.uint8)]


def revcomp(seq: str) -> str:
    return seq[::-1].translate(str.maketrans("ACGTUacgtuNn", "TGCAAtgcaaNn"))


def read_m8(path: Path):
    rows = []
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        f = line.rstrip("\n").split("\t")
        rows.append(dict(q=f[0], s=f[1], qs=int(f[6]), qe=int(f[7]), ss=int(f[8]), se=int(f[9]), line=line.rstrip("\n")))
    return rows


class Replay:
    """program = "blastp" | "blastn"; frames follow src/search_datastructures.hpp:380-385 (blastn query: 2 frames)."""

    def __init__(self, handle, program, queries, subjects, tmp_dir, max_evalue=1e-2):
        self.h, self.program, self.tmp = handle, program, Path(tmp_dir)
        self.q_ids, q_seqs = queries
        self.s_ids, s_seqs = subjects
        self.first = lambda name: name.split()[0] if name.split() else name
        self.qi = {self.first(n): i for i, n in enumerate(self.q_ids)}
        self.si = {self.first(n): i for i, n in enumerate(self.s_ids)}
        if program == "blastp":
            self.qf, order, unk = 1, AA_ORDER, "X"
            self.scoring = capi.builtin_scoring(62, gap_open=-11, gap_extend=-1)
            self.ka = capi.karlin_params(62, gap_open=-11, gap_extend=-1)
            frames_q = [q_seqs]
        else:
            self.qf, order, unk = 2, DNA5_BIOCPP, "N"
            self.scoring = capi.builtin_scoring(0, match=2, mismatch=-3, gap_open=-5, gap_extend=-2)
            self.ka = capi.karlin_params(0, 2, -3, -5, -2)
            frames_q = [q_seqs, [revcomp(x) for x in q_seqs]]
        self.q_orig_len = np.array([len(x) for x in q_seqs], dtype=np.uint64)
        qparts = [ranks(frames_q[k][i], order, unk) for i in range(len(q_seqs)) for k in range(self.qf)]
        sparts = [ranks(x, order, unk) for x in s_seqs]
        self.q_len = np.array([len(x) for x in qparts], dtype=np.uint64)
        self.s_len = np.array([len(x) for x in sparts], dtype=np.uint64)
        self.q_off = np.concatenate([[0], np.cumsum(self.q_len)[:-1]]).astype(np.uint64)
        self.s_off = np.concatenate([[0], np.cumsum(self.s_len)[:-1]]).astype(np.uint64)
        self.q_res = np.concatenate(qparts) if qparts else np.zeros(0, np.uint8)
        self.s_res = np.concatenate(sparts) if sparts else np.zeros(0, np.uint8)
        self.db_total = int(self.s_len.sum())
        self.max_evalue = max_evalue
        self.q_ascii = "".join(q_seqs).encode()
        self.q_ascii_off = np.concatenate([[0], np.cumsum([len(x) for x in q_seqs])[:-1]]).astype(np.uint64)
        handle.set_scoring(self.scoring, 0)

    def seeds_of(self, row):
        """Two 1-residue seeds: on the diagonal the alignment starts on and on the one it ends on (frame coordinates)."""
        nq, ns = self.qi[self.first(row["q"])], self.si[self.first(row["s"])]
        qs, qe, ss, se = row["qs"] - 1, row["qe"], row["ss"], row["se"]
        frame = 0
        if self.program == "blastn" and ss > se:  # minus strand: positions on the reverse-complemented query, subject swapped
            frame = 1
            ql = int(self.q_orig_len[nq])
            qs, qe = ql - qe, ql - qs
            ss, se = se, ss
        ss -= 1
        qid = nq * self.qf + frame
        m = np.zeros(2, dtype=capi.MATCH_DTYPE)
        m[0] = (qid, ns, qs, qs + 1, ss, ss + 1)
        m[1] = (qid, ns, qe - 1, qe, se - 1, se)
        return m

    def run(self, rows):
        """Returns (reproduced, shadowed, missing) lists of golden lines."""
        matches = np.concatenate([self.seeds_of(r) for r in rows])
        params = capi.SearchParams(self.max_evalue, -1, 0, self.db_total, 0, self.qf, 1, 0,
                                   capi.LX_FRAMES_REVCOMP if self.qf == 2 else capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, self.ka)
        bms, ops, _ = self.h.iterate_matches(self.q_res, self.q_off, self.q_len, self.q_orig_len, self.s_res, self.s_off, self.s_len,
                                             matches, params)
        out = self.tmp / "replay.m8"
        allops = b"".join(ops)
        bms = bms.copy()
        at = 0
        for i, o in enumerate(ops):
            bms["ops_off"][i] = at
            at += len(o)
        capi.write_records(out, capi.LX_OUT_BLAST_TAB, bms, allops, self.q_ids, self.q_orig_len, self.s_ids, self.s_len,
                           program=self.program, write_header=False, q_ascii=self.q_ascii, q_ascii_off=self.q_ascii_off)
        produced = set(open(out).read().splitlines())
        golden = {r["line"] for r in rows}
        reproduced = [r["line"] for r in rows if r["line"] in produced]
        rest = [r for r in rows if r["line"] not in produced]
        # a row is "shadowed" when everything this harness produced for its pair is itself a golden row (a stronger hit of the
        # same window won); "missing" when the harness produced a line for the pair that the reference does not have
        by_pair = {}
        for line in produced:
            f = line.split("\t")
            by_pair.setdefault((f[0], f[1]), []).append(line)
        shadowed, missing = [], []
        for r in rest:
            mine = by_pair.get((self.first(r["q"]), self.first(r["s"])), [])
            (shadowed if mine and all(x in golden for x in mine) else missing).append(r["line"])
        return reproduced, shadowed, missing

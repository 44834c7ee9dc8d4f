"""Replays the rows of a BLAST-tabular file (the reference's goldens, tools/fetch_reference_goldens.py) through the
extension driver: for every row a seed pair on the reported alignment's first and last diagonal goes through
lx_iterate_matches (widen / merge, both GPU passes, statistics, filters) and lx_write_records; the row must come out of it
character for character -- coordinates, identity, length, mismatches, gap opens, e-value, bit score.

The SAM goldens (output_blast{n,n_bs,p}_fm.sam) are replayed the same way from POS, CIGAR, FLAG and the `qf` tag; compared are the
fields that pin the traceback: strand, RNAME, POS, CIGAR (BLASTN / bisulfite; BLASTP writes "*", src/search_output.hpp:526-531),
`AS` and `NM` -- the only reference data that discriminates the GapsLeft tie rule.  program = "blastn_bs" is searchbs: four
query frames, two subject frames, both scoring schemes, E <= 1e-9 (src/search_options.hpp:261-264).

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


def read_sam(path: Path):
    rows = []
    for line in open(path):
        if line.startswith("@") or not line.strip():
            continue
        f = line.rstrip("\n").split("\t")
        tags = {t[:2]: t[5:] for t in f[11:]}
        rows.append(dict(q=f[0], flag=int(f[1]), s=f[2], pos=int(f[3]), cigar=f[5], tags=tags, line=line.rstrip("\n")))
    return rows


def cigar_elements(cigar: str):
    out, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num), ch))
            num = ""
    return out


def sam_key(row):
    """The fields of a SAM record the replay compares (see the module docstring)."""
    return (row["q"].split()[0], row["flag"] & 16, row["s"], row["pos"], row["cigar"], row["tags"].get("AS"), row["tags"].get("NM"),
            row["tags"].get("qf"))


class Replay:
    """program = "blastp" | "blastn"; frames follow src/search_datastructures.hpp:380-385 (blastn query: 2 frames)."""

    def __init__(self, handle, program, queries, subjects, tmp_dir, max_evalue=1e-2):
        self.h, self.program, self.tmp = handle, program, Path(tmp_dir)
        self.q_ids, q_seqs = queries
        self.s_ids, s_seqs = subjects
        self.first = lambda name: name.split()[0] if name.split() else name
        self.qi = {self.first(n): i for i, n in enumerate(self.q_ids)}
        self.si = {self.first(n): i for i, n in enumerate(self.s_ids)}
        self.sf, self.bs = 1, False
        if program == "blastp":
            self.qf, order, unk = 1, AA_ORDER, "X"
            self.scoring = capi.builtin_scoring(62, gap_open=-11, gap_extend=-1)
            self.ka = capi.karlin_params(62, gap_open=-11, gap_extend=-1)
            frames_q = [q_seqs]
        elif program == "blastn_bs":
            # searchbs: strand, strand, reverse complement, reverse complement / every subject twice
            # (src/shared_definitions.hpp:249-261), SeqAn Dna5 ranks, forward scheme in slot 0, reverse in slot 1
            self.qf, self.sf, self.bs, order, unk = 4, 2, True, DNA5_SEQAN, "N"
            self.scoring = capi.builtin_scoring(-1, match=2, mismatch=-3, gap_open=-5, gap_extend=-2)
            self.scoring_rev = capi.builtin_scoring(-2, match=2, mismatch=-3, gap_open=-5, gap_extend=-2)
            self.ka = capi.karlin_params(0, 2, -3, -5, -2)
            rc = [revcomp(x) for x in q_seqs]
            frames_q = [q_seqs, q_seqs, rc, rc]
            s_seqs = [x for x in s_seqs for _ in range(2)]
            if max_evalue == 1e-2:
                max_evalue = 1e-9
        else:
            self.qf, order, unk = 2, DNA5_BIOCPP, "N"
            self.scoring = capi.builtin_scoring(0, match=2, mismatch=-3, gap_open=-5, gap_extend=-2)
            self.ka = capi.karlin_params(0, 2, -3, -5, -2)
            frames_q = [q_seqs, [revcomp(x) for x in q_seqs]]
        self.writer_program = "blastn" if program == "blastn_bs" else program
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
        self.s_orig_len = self.s_len[:: self.sf].copy()
        handle.set_scoring(self.scoring, 0)
        if self.bs:
            handle.set_scoring(self.scoring_rev, 1)

    def seeds_of(self, row):
        """Two 1-residue seeds: on the diagonal the alignment starts on and on the one it ends on (frame coordinates)."""
        nq, ns = self.qi[self.first(row["q"])], self.si[self.first(row["s"])]
        qs, qe, ss, se = row["qs"] - 1, row["qe"], row["ss"], row["se"]
        frame = 0
        if self.program in ("blastn", "blastn_bs") and ss > se:  # minus strand: positions on the reverse-complemented query, subject swapped
            frame = 1
            ql = int(self.q_orig_len[nq])
            qs, qe = ql - qe, ql - qs
            ss, se = se, ss
        ss -= 1
        if self.bs:  # the table does not say which bisulfite duplicate found the row: both (the better one shadows the other)
            out = []
            for dup in (0, 1):
                qid, sid = nq * 4 + 2 * frame + dup, ns * 2 + dup
                out += [(qid, sid, qs, qs + 1, ss, ss + 1), (qid, sid, qe - 1, qe, se - 1, se)]
            m = np.zeros(len(out), dtype=capi.MATCH_DTYPE)
            for k, t in enumerate(out):
                m[k] = t
            return m
        qid = nq * self.qf + frame
        m = np.zeros(2, dtype=capi.MATCH_DTYPE)
        m[0] = (qid, ns, qs, qs + 1, ss, ss + 1)
        m[1] = (qid, ns, qe - 1, qe, se - 1, se)
        return m

    def seeds_of_sam(self, row):
        """Seeds from a SAM record: POS is the subject start, the CIGAR (reversed on the minus strand, hard clip of the second
        bisulfite duplicate in front: src/search_output.hpp:126-143, :192-193) gives the query start and both extents, the `qf`
        tag the frame."""
        nq, ns = self.qi[self.first(row["q"])], self.si[self.first(row["s"])]
        el = cigar_elements(row["cigar"]) if row["cigar"] != "*" else []
        qf = int(row["tags"].get("qf", "0"))
        if row["flag"] & 16:
            el = el[::-1]
        el = [e for e in el if e[1] != "H"]
        qs = el[0][0] if el and el[0][1] == "S" else 0
        qlen = sum(n for n, op in el if op in "MI")
        slen = sum(n for n, op in el if op in "MD")
        if not el:  # BLASTP: no CIGAR -- the whole query against a window around POS, extent from the alignment length
            return None
        ss = row["pos"] - 1
        if self.bs:
            frame = (0 if qf > 0 else 2) + (abs(qf) - 1)
            qid, sid = nq * 4 + frame, ns * 2 + (abs(qf) - 1)
        elif self.qf == 2:
            qid, sid = nq * 2 + (1 if qf < 0 else 0), ns
        else:
            qid, sid = nq, ns
        m = np.zeros(2, dtype=capi.MATCH_DTYPE)
        m[0] = (qid, sid, qs, qs + 1, ss, ss + 1)
        m[1] = (qid, sid, qs + qlen - 1, qs + qlen, ss + slen - 1, ss + slen)
        return m

    def _iterate(self, matches):
        params = capi.SearchParams(self.max_evalue, -1, 0, self.db_total, 0, self.qf, self.sf, 1 if self.bs else 0,
                                   capi.LX_FRAMES_BISULFITE if self.bs else capi.LX_FRAMES_REVCOMP if self.qf == 2 else capi.LX_FRAMES_NONE,
                                   capi.LX_FRAMES_BISULFITE if self.bs else capi.LX_FRAMES_NONE, self.ka)
        bms, ops, _ = self.h.iterate_matches(self.q_res, self.q_off, self.q_len, self.q_orig_len, self.s_res, self.s_off, self.s_len,
                                             matches, params)
        allops = b"".join(ops)
        bms = bms.copy()
        at = 0
        for i, o in enumerate(ops):
            bms["ops_off"][i] = at
            at += len(o)
        return bms, allops

    def run_sam(self, rows):
        """Returns (reproduced, shadowed, missing) lists of golden SAM lines, compared by sam_key."""
        seeds = [self.seeds_of_sam(r) for r in rows]
        usable = [r for r, sd in zip(rows, seeds) if sd is not None]
        bms, allops = self._iterate(np.concatenate([sd for sd in seeds if sd is not None]))
        out = self.tmp / "replay.sam"
        capi.write_records(out, capi.LX_OUT_SAM, bms, allops, self.q_ids, self.q_orig_len, self.s_ids, self.s_orig_len,
                           program=self.writer_program, write_header=False, q_ascii=self.q_ascii, q_ascii_off=self.q_ascii_off)
        produced = {sam_key(r) for r in read_sam(out)}
        golden = {sam_key(r) for r in usable}
        reproduced = [r["line"] for r in usable if sam_key(r) in produced]
        by_pair = {}
        for k in produced:
            by_pair.setdefault((k[0], k[2]), []).append(k)
        shadowed, missing = [], []
        for r in usable:
            if sam_key(r) in produced:
                continue
            mine = by_pair.get((self.first(r["q"]), self.first(r["s"])), [])
            (shadowed if mine and all(x in golden for x in mine) else missing).append(r["line"])
        return reproduced, shadowed, missing

    def run(self, rows):
        """Returns (reproduced, shadowed, missing) lists of golden lines."""
        bms, allops = self._iterate(np.concatenate([self.seeds_of(r) for r in rows]))
        out = self.tmp / "replay.m8"
        capi.write_records(out, capi.LX_OUT_BLAST_TAB, bms, allops, self.q_ids, self.q_orig_len, self.s_ids, self.s_orig_len,
                           program=self.writer_program, write_header=False, q_ascii=self.q_ascii, q_ascii_off=self.q_ascii_off)
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

    def run_lenient(self, rows):
        """run() for searchbs tables: a row is sought among what BOTH bisulfite duplicates of its strand produce (the table does
        not name the duplicate), lines of the other duplicate are not counted against it."""
        reproduced, shadowed, missing = self.run(rows)
        return reproduced, shadowed + missing, []

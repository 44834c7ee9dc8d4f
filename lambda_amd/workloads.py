"""The synthetic seed-batch workloads of BASELINE.json `configs[1..4]` and how a run of
them is split over ranks and sub-batches.  Pure host logic (no torch, no GPU): bench.py executes what `plan()` returns,
the CPU tests check the arithmetic.

Scheme parameters are the reference's defaults for each program (/root/reference/src/search_options.hpp:290-307:
protein gaps -11/-1, nucleotide and bisulfite -5/-2 with match/mismatch +2/-3, :93-94; maxEValue 1e-2, bisulfite 1e-9,
:96-97, :261-264).  The database only enters through dbTotalLength in the e-value (src/search_algo.hpp:317-319).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import shard


@dataclass(frozen=True)
class Direction:
    """One call of iterateMatchesFullSimd: the scoring slot it uses and how the reads of it were converted."""
    slot: int
    scoring: tuple  # (scoring_method, match, mismatch, gap_open_lambda, gap_extend) for lx_builtin_scoring
    convert: str | None = None  # bisulfite conversion of the query relative to the window: "CT" (forward) / "GA" (reverse)


@dataclass(frozen=True)
class Workload:
    key: int                 # index into BASELINE.json "configs"
    name: str
    program: str             # what the reference would call the run
    lq: int
    windows: int             # candidate windows per query
    queries_total: int       # queries of the whole job as BASELINE.json states it
    db_length: int           # dbTotalLength
    max_evalue: float
    karlin: tuple            # arguments of lx_karlin_params
    directions: tuple        # of Direction; queries are dealt round-robin to the directions (a read comes from one strand)
    alphabet: tuple          # residue ranks drawn uniformly
    n_rate: float = 0.0
    n_rank: int = 0
    sub_rate: float = 0.25
    indel_rate: float = 0.02
    convert_rate: float = 0.0
    seed: int = 0
    gpus: int = 1            # GPUs BASELINE.json quotes the config on
    scaling: str = "weak"    # default when --gpus N != gpus: "weak" = every rank owns queries_per_gpu, "strong" = queries_total split
    batch_queries: int = 0   # queries per device call (0 = all of a rank's queries of one direction at once)
    dtype_note: str = ""

    @property
    def queries_per_gpu(self) -> int:
        return self.queries_total // self.gpus


_STD20 = (0, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18, 19, 21, 22, 23)  # SeqAn AminoAcid ranks of ACDEFGHIKLMNPQRSTVWY
_PROT = Direction(0, (62, 0, 0, -11, -1))
_NUCL = Direction(0, (0, 2, -3, -5, -2))

WORKLOADS = {
    1: Workload(key=1, name="searchp headline", program="blastp", lq=150, windows=32, queries_total=100_000,
                db_length=205_000_000, max_evalue=1e-2, karlin=(62, 0, 0, -11, -1), directions=(_PROT,), alphabet=_STD20,
                seed=0x1A3BDA02, gpus=1, scaling="weak"),
    # BioC++ dna5 ranks A, C, G, N, T (match/mismatch-scored alphabets keep the BioC++ rank: N vs N is a match,
    # src/seqan2_to_biocpp.hpp:392-393)
    2: Workload(key=2, name="searchn", program="blastn", lq=150, windows=8, queries_total=1_000_000, db_length=100_000_000,
                max_evalue=1e-2, karlin=(0, 2, -3, -5, -2), directions=(_NUCL,), alphabet=(0, 1, 2, 4), n_rate=0.01, n_rank=3,
                sub_rate=0.05, indel_rate=0.01, seed=0x1A3BDA03, gpus=1, scaling="weak"),
    3: Workload(key=3, name="searchp scale-out", program="blastp", lq=200, windows=32, queries_total=1_000_000,
                db_length=200_000_000_000, max_evalue=1e-2, karlin=(62, 0, 0, -11, -1), directions=(_PROT,), alphabet=_STD20,
                seed=0x1A3BDA04, gpus=8, scaling="strong", batch_queries=125_000),
    # SeqAn Dna5 order A, C, G, T, N in bisulfite mode (src/seqan2_to_biocpp.hpp:382-395); reads of even index come from
    # the C->T strand (forward scheme, slot 0), odd ones from the G->A strand (reverse scheme, slot 1:
    # src/search_algo.hpp:1367-1379); 2 subject frames -> dbTotalLength = 2 x 200 Mbp
    4: Workload(key=4, name="searchn bisulfite", program="blastn (bisulfite)", lq=150, windows=8, queries_total=500_000,
                db_length=400_000_000, max_evalue=1e-9, karlin=(0, 2, -3, -5, -2),
                directions=(Direction(0, (-1, 2, -3, -5, -2), "CT"), Direction(1, (-2, 2, -3, -5, -2), "GA")),
                alphabet=(0, 1, 2, 3), n_rate=0.0, n_rank=4, sub_rate=0.05, indel_rate=0.01, convert_rate=0.99,
                seed=0x1A3BDA05, gpus=4, scaling="strong"),
}


@dataclass
class Batch:
    """One device call: queries [q_lo, q_hi) of the job (global numbering), all of one direction."""
    direction: Direction
    q_lo: int
    q_hi: int
    seed: int

    @property
    def n_queries(self) -> int:
        return self.q_hi - self.q_lo


@dataclass
class Plan:
    workload: Workload
    world: int
    rank: int
    scaling: str
    q_lo: int                # this rank's query range in the job
    q_hi: int
    job_queries: int         # queries of the whole job over all ranks
    batches: list = field(default_factory=list)

    @property
    def n_queries(self) -> int:
        return self.q_hi - self.q_lo


def plan(w: Workload, world: int, rank: int, total_queries: int | None = None, queries_per_rank: int | None = None,
         batch_queries: int | None = None) -> Plan:
    """Query range of `rank` and its split into device calls.

    strong scaling (--total-queries, or the workload's default): the job's queries are split contiguously over the ranks
    like the reference splits them over its threads (/root/reference/src/search.cpp:384-385); weak scaling: every rank
    owns `queries_per_rank` queries of its own (rank r's are numbered [r Q, (r + 1) Q))."""
    if total_queries is not None and queries_per_rank is not None:
        raise ValueError("give either the job's total or the per-rank query count")
    scaling = "strong" if total_queries is not None else "weak" if queries_per_rank is not None else w.scaling
    if scaling == "strong":
        job = total_queries if total_queries is not None else w.queries_total
        lo, hi = shard.shard_range(job, rank, world)
    else:
        per = queries_per_rank if queries_per_rank is not None else w.queries_per_gpu
        job = per * world
        lo, hi = rank * per, (rank + 1) * per
    p = Plan(w, world, rank, scaling, lo, hi, job)
    nd = len(w.directions)
    bq = batch_queries if batch_queries else (w.batch_queries or (hi - lo))
    # directions deal the queries round-robin (query i belongs to direction i % nd); each direction's share of the rank's
    # range is cut into calls of at most bq queries
    for d, direction in enumerate(w.directions):
        first = lo + ((d - lo) % nd)          # first query >= lo with index % nd == d
        count = 0 if first >= hi else (hi - first + nd - 1) // nd
        done = 0
        while done < count:
            take = min(max(bq, 1), count - done)
            # numbered in units of this direction's queries; the seed makes every call's data different and reproducible
            p.batches.append(Batch(direction, first // nd + done, first // nd + done + take,
                                   (w.seed * 1_000_003 + d * 7919 + (first // nd + done)) & 0x7FFFFFFFFFFF))
            done += take
    return p


def cells_of(w: Workload, n_queries: int) -> float:
    """Full-rectangle cells (sum Lq * Ls, the reference has no band) of `n_queries` queries of the workload."""
    from . import synth

    return float(n_queries) * w.windows * w.lq * synth.window_len(w.lq)


def describe(w: Workload) -> str:
    from . import synth

    unit = "aa" if w.program == "blastp" else "bp"
    scheme = "BLOSUM62 gap 11/1" if w.karlin[0] == 62 else "match/mismatch +2/-3 gap 5/2"
    if len(w.directions) == 2:
        scheme = "bisulfite matrices fwd+rev (bisulfite_scoring.hpp) +2/-3 gap 5/2, %.0f %% conversion" % (100 * w.convert_rate)
    extra = f", {100 * w.n_rate:.0f} % N" if w.n_rate else ""
    return (f"{w.name}: {w.program} {scheme}, {w.lq} {unit} queries x {w.windows} windows of {synth.window_len(w.lq)} {unit}{extra} "
            f"(BASELINE.json configs[{w.key}]: {w.queries_total} queries on {w.gpus} GPU{'s' if w.gpus > 1 else ''}), "
            f"cells = sum Lq*Ls (full rectangle, band off as in the reference)")


def alphabet_array(w: Workload) -> np.ndarray:
    return np.asarray(w.alphabet, dtype=np.uint8)

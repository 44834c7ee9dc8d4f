"""The pin to reference OUTPUT: replay of the reference's own goldens -- BLAST-tabular
(output_blastp_fm.m8, output_blastn_fm.m8, output_blastn_bs_fm.m8; /root/reference/test/data/datasources.cmake:179-181, :140-142,
:101-103) and SAM (output_blastn_fm.sam, output_blastn_bs_fm.sam; :146-148, :107-109: CIGAR, NM, AS, POS) -- through
lx_iterate_matches + lx_write_records.  The files are remote-only (no network here): `python tools/fetch_reference_goldens.py`
on a networked machine puts them under tests/golden/reference/; without them the replay tests SKIP, and
test_replay_harness_on_own_output keeps the harness itself honest on files made by this repo's own CLI."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from lambda_amd import capi
from tests import reference_replay as rr

pytestmark = pytest.mark.gpu
REF = Path(__file__).resolve().parent / "golden" / "reference"


def _have(*names):
    return all((REF / n).exists() for n in names)


@pytest.mark.skipif(not _have("db_prot.fasta.gz", "queries_prot.fasta.gz", "output_blastp_fm.m8"),
                    reason="reference goldens not fetched (tools/fetch_reference_goldens.py needs a network)")
def test_blastp_golden_rows_are_reproduced(handle, tmp_path):
    rep = rr.Replay(handle, "blastp", rr.read_fasta(REF / "queries_prot.fasta.gz"), rr.read_fasta(REF / "db_prot.fasta.gz"), tmp_path)
    rows = rr.read_m8(REF / "output_blastp_fm.m8")
    ok, shadowed, missing = rep.run(rows)
    print(f"blastp: {len(ok)} reproduced, {len(shadowed)} shadowed, {len(missing)} different of {len(rows)}")
    assert not missing, missing[:5]
    assert len(ok) >= 0.9 * len(rows)


@pytest.mark.skipif(not _have("db_nucl.fasta.gz", "queries_nucl.fasta.gz", "output_blastn_fm.m8"),
                    reason="reference goldens not fetched (tools/fetch_reference_goldens.py needs a network)")
def test_blastn_golden_rows_are_reproduced(handle, tmp_path):
    rep = rr.Replay(handle, "blastn", rr.read_fasta(REF / "queries_nucl.fasta.gz"), rr.read_fasta(REF / "db_nucl.fasta.gz"), tmp_path)
    rows = rr.read_m8(REF / "output_blastn_fm.m8")
    ok, shadowed, missing = rep.run(rows)
    print(f"blastn: {len(ok)} reproduced, {len(shadowed)} shadowed, {len(missing)} different of {len(rows)}")
    assert not missing, missing[:5]
    assert len(ok) >= 0.9 * len(rows)


@pytest.mark.skipif(not _have("db_nucl_bs.fasta.gz", "queries_nucl_bs.fasta.gz", "output_blastn_bs_fm.m8"),
                    reason="reference goldens not fetched (tools/fetch_reference_goldens.py needs a network)")
def test_bisulfite_golden_rows_are_reproduced(handle, tmp_path):
    """output_blastn_bs_fm.m8 (datasources.cmake:101-103): two scoring slots, E <= 1e-9 (src/search_options.hpp:261-264)."""
    rep = rr.Replay(handle, "blastn_bs", rr.read_fasta(REF / "queries_nucl_bs.fasta.gz"), rr.read_fasta(REF / "db_nucl_bs.fasta.gz"), tmp_path)
    rows = rr.read_m8(REF / "output_blastn_bs_fm.m8")
    ok, shadowed, missing = rep.run_lenient(rows)
    print(f"blastn-bisulfite: {len(ok)} reproduced, {len(shadowed)} not found among both duplicates' lines, of {len(rows)}")
    assert len(ok) >= 0.9 * len(rows)


@pytest.mark.parametrize("program,qry,db,sam", [("blastn", "queries_nucl.fasta.gz", "db_nucl.fasta.gz", "output_blastn_fm.sam"),
                                                ("blastn_bs", "queries_nucl_bs.fasta.gz", "db_nucl_bs.fasta.gz", "output_blastn_bs_fm.sam")])
def test_sam_golden_cigars_are_reproduced(handle, tmp_path, program, qry, db, sam):
    """The SAM goldens (datasources.cmake:146-148, :107-109): strand, POS, CIGAR, AS and NM of every record -- the data that
    pins the traceback's tie rule (GapsLeft, src/search_algo.hpp:1083)."""
    if not _have(qry, db, sam):
        pytest.skip("reference goldens not fetched (tools/fetch_reference_goldens.py needs a network)")
    rep = rr.Replay(handle, program, rr.read_fasta(REF / qry), rr.read_fasta(REF / db), tmp_path)
    rows = rr.read_sam(REF / sam)
    ok, shadowed, missing = rep.run_sam(rows)
    print(f"{sam}: {len(ok)} reproduced, {len(shadowed)} shadowed, {len(missing)} different of {len(rows)}")
    assert not missing, missing[:5]
    assert len(ok) >= 0.9 * len(rows)


def _own_nucleotide_run(tmp_path, cmd, ext):
    """A small searchn / searchbs run of this repo's own CLI (planted reads, both strands; bisulfite: both conversions)."""
    rng = np.random.default_rng(17)
    genome = ["".join("ACGT"[i] for i in rng.integers(0, 4, 20000)) for _ in range(2)]
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    for k in range(60):
        c, a = int(rng.integers(0, 2)), int(rng.integers(0, 19800))
        frag = genome[c][a:a + 150]
        if k % 2:
            frag = frag.translate(comp)[::-1]
        r = list(frag)
        if cmd == "searchbs":
            frm, to = ("C", "T") if k % 4 < 2 else ("G", "A")
            r = [to if ch == frm else ch for ch in r]
        for p in rng.integers(0, 150, 3):
            r[p] = "ACGT"[int(rng.integers(0, 4))]
        if k % 5 == 0:  # an indel now and then: gaps in the CIGAR
            r = r[:70] + r[72:]
        reads.append("".join(r))
    from tests.test_cli import _cli, _fasta

    _fasta(tmp_path / "g.fasta", [f"chr{i} genome" for i in range(2)], genome)
    _fasta(tmp_path / "r.fasta", [f"read{k}" for k in range(60)], reads)
    out = tmp_path / f"own.{ext}"
    r = subprocess.run([str(_cli()), cmd, "-q", str(tmp_path / "r.fasta"), "-d", str(tmp_path / "g.fasta"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return tmp_path / "r.fasta", tmp_path / "g.fasta", out


@pytest.mark.parametrize("cmd,program", [("searchn", "blastn"), ("searchbs", "blastn_bs")])
def test_sam_replay_on_own_output(handle, tmp_path, cmd, program):
    """The SAM replay end to end on a file made by this repo's own CLI: every record's strand, POS, CIGAR, AS, NM and qf must
    come back from POS + CIGAR + qf alone."""
    qry, db, out = _own_nucleotide_run(tmp_path, cmd, "sam")
    rows = rr.read_sam(out)
    assert len(rows) >= 50 and any("D" in r["cigar"] or "I" in r["cigar"] for r in rows)
    rep = rr.Replay(handle, program, rr.read_fasta(qry), rr.read_fasta(db), tmp_path)
    ok, shadowed, missing = rep.run_sam(rows)
    assert not missing, missing[:3]
    assert len(ok) + len(shadowed) == len(rows) and len(ok) >= 0.9 * len(rows)


def test_bisulfite_table_replay_on_own_output(handle, tmp_path):
    qry, db, out = _own_nucleotide_run(tmp_path, "searchbs", "m8")
    rows = rr.read_m8(out)
    assert len(rows) >= 50
    rep = rr.Replay(handle, "blastn_bs", rr.read_fasta(qry), rr.read_fasta(db), tmp_path)
    ok, rest, _ = rep.run_lenient(rows)
    assert len(ok) >= 0.9 * len(rows), (len(ok), len(rows), rest[:3])


def test_replay_harness_on_own_output(handle, tmp_path):
    """The harness end to end on a file pair this repo makes itself: planted homologs -> `lambda3 searchp` (own seeder, GPU
    extension, tabular writer) -> replay of its .m8 from the row coordinates alone.  Every row must come back."""
    from tests.test_cli import _cli, _make_config1  # the CLI tests' generator (scaled down)

    _make_config1(tmp_path, nq=150, ndb=500)
    db, qry, out = tmp_path / "db.fasta", tmp_path / "q.fasta", tmp_path / "own.m8"
    r = subprocess.run([str(_cli()), "searchp", "-q", str(qry), "-d", str(db), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = rr.read_m8(out)
    assert len(rows) >= 15
    rep = rr.Replay(handle, "blastp", rr.read_fasta(qry), rr.read_fasta(db), tmp_path)
    ok, shadowed, missing = rep.run(rows)
    assert not missing, missing[:3]
    assert len(ok) + len(shadowed) == len(rows) and len(ok) >= 0.9 * len(rows)

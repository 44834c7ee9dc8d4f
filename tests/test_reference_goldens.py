"""The pin to reference OUTPUT: replay of the reference's own BLAST-tabular goldens
(output_blastp_fm.m8, output_blastn_fm.m8; /root/reference/test/data/datasources.cmake:179-181, :140-142) through
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

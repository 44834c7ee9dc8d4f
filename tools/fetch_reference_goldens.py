#!/usr/bin/env python3
"""Fetches the reference's own whole-program goldens and their inputs (needs a network; this container has none).

    python tools/fetch_reference_goldens.py [--dest tests/golden/reference]

URLs and SHA-256 sums are the ones the reference's test suite declares (/root/reference/test/data/datasources.cmake: base URL
:7; db_prot :17-19, db_nucl :9-11, db_nucl_bs :13-15; queries_prot :65-67, queries_nucl :57-59, queries_nucl_bs :61-63;
output_blastp_fm.m8 :179-181, output_blastn_fm.m8 :140-142, output_blastn_bs_fm.m8 :101-103, the three .sam files
:185-187, :146-148, :107-109).  They are DATA (sequences and
BLAST-tabular result tables), the only reference outputs that exist for this path; with them in place
`pytest tests/test_reference_goldens.py -m gpu` replays every row through lx_iterate_matches and the tabular writer and
requires identical coordinates, counts, bit score and e-value -- the step that turns "parity unpinned" into a pin.
Every download is verified against its SHA-256; a mismatch aborts and leaves nothing behind.
"""
import argparse
import hashlib
import sys
import urllib.request
from pathlib import Path

BASEURL = "https://raw.githubusercontent.com/h-2/lambda-testdata/832453c8721094af511a2d7041dbb107a9ecc912"
FILES = {
    "input_files/db_prot.fasta.gz": "2c27f09f77e1f8ec0fea0aa1cae75f6634bab852a843a48033fc0f5251d57626",
    "input_files/db_nucl.fasta.gz": "614f8d7863c40facb7fffe666ce04341368a43ea90b8125cb675907f315bb0a2",
    "input_files/db_nucl_bs.fasta.gz": "160375ac5ff4426f1768981a0215495efd6de0b847e3226c5c7c112370a599a1",
    "input_files/queries_prot.fasta.gz": "e21411f422c1dd844696c8ca8a08b93e8243d7f77307731c5f2ee8dcc60d0703",
    "input_files/queries_nucl.fasta.gz": "7a8adcc7ee5d967992a0624b9fb704118223abd112534c2c6456104b30ddad88",
    "input_files/queries_nucl_bs.fasta.gz": "a358ace3e6f35bd379854fd8870a63a6da62b6bd3e72d95f11e5386398881f0a",
    "output_files/output_blastp_fm.m8": "99f520bb55f5c1b371ae5ba41b881c21360a84e7fcb8ff097e01036beb801d4c",
    "output_files/output_blastn_fm.m8": "18b7a0feb4b5e76a44be7ec4ae26ceb1be05fa257df0725a6d231f841de01d54",
    "output_files/output_blastn_bs_fm.m8": "732b439bded780e0b4b68478e6e89368934cab43d261c2f41e48d90ab505a01e",
    # the SAM goldens carry CIGAR / NM / AS / POS: the only reference data that discriminates the traceback tie rule
    # (datasources.cmake:185-187, :146-148, :107-109)
    "output_files/output_blastp_fm.sam": "d7267490e2c6c1ebf838fb7f08a5f370bf0f2b23024ca76db6bfed440851fc91",
    "output_files/output_blastn_fm.sam": "ca8202a68e22d59e731d023c021244e27e664077d8aa68662041346ac828c110",
    "output_files/output_blastn_bs_fm.sam": "2247d450da5ad0fa76f630e373bbf70d0fde79a736637d19f8f25d985178fcfa",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dest", default=str(Path(__file__).resolve().parent.parent / "tests" / "golden" / "reference"))
    ap.add_argument("--timeout", default="60")
    args = ap.parse_args()
    dest = Path(args.dest)
    dest.mkdir(parents=True, exist_ok=True)
    for rel, sha in FILES.items():
        out = dest / Path(rel).name
        if out.exists() and hashlib.sha256(out.read_bytes()).hexdigest() == sha:
            print(f"ok      {out.name}")
            continue
        url = f"{BASEURL}/{rel}"
        try:
            data = urllib.request.urlopen(url, timeout=float(args.timeout)).read()
        except Exception as e:  # no network here: say so, change nothing
            print(f"FAILED  {url}: {e}", file=sys.stderr)
            return 1
        got = hashlib.sha256(data).hexdigest()
        if got != sha:
            print(f"FAILED  {out.name}: SHA-256 {got} != {sha}", file=sys.stderr)
            return 1
        out.write_bytes(data)
        print(f"fetched {out.name} ({len(data)} bytes)")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""The minimal lambda3 front end (plumbing, BASELINE.json configs[0]): FASTA in, .m8 / .sam out, seeding -> prefilter ->
extension -> records, middle stages on the GPU.  CPU part: the binary exists and fails loudly without a device."""
import math
import subprocess
from pathlib import Path

import numpy as np
import pytest

from lambda_amd import build, capi, synth
from tests import oracle_lib

ORDER = "ABCDEFGHIJKLMNOPQRSTUVWYZX*"
STD = "ACDEFGHIKLMNPQRSTVWY"


def _cli():
    return build.build_cli()


def _fasta(path, ids, seqs):
    with open(path, "w") as f:
        for i, s in zip(ids, seqs):
            f.write(f">{i}\n")
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + "\n")


def _make_config1(tmp, nq=1000, ndb=10000, lq=100, seed=0x1A3BDA01):
    """DB lengths log-normal (mu = ln 300, sigma 0.6, clamp 50..2000), uniform over the 20
    standard residues; 30 % of the queries are mutated copies of DB regions (25 % substitutions, 2 % indels)."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.exp(rng.normal(math.log(300), 0.6, ndb)).astype(int), 50, 2000)
    db = ["".join(STD[i] for i in rng.integers(0, 20, L)) for L in lens]
    qs, truth = [], {}
    for k in range(nq):
        if rng.random() < 0.3:
            while True:
                j = int(rng.integers(0, ndb))
                if lens[j] >= lq + 10:
                    break
            a = int(rng.integers(0, lens[j] - lq - 5))
            src = db[j][a:a + lq + 5]
            out = []
            for ch in src:
                r = rng.random()
                if r < 0.01:
                    continue                      # deletion
                if r < 0.02:
                    out.append(STD[int(rng.integers(0, 20))])  # insertion
                out.append(STD[int(rng.integers(0, 20))] if rng.random() < 0.25 else ch)
            qs.append("".join(out)[:lq].ljust(lq, "A"))
            truth[k] = j
        else:
            qs.append("".join(STD[i] for i in rng.integers(0, 20, lq)))
    _fasta(tmp / "db.fasta", [f"sp{j} protein {j}" for j in range(ndb)], db)
    _fasta(tmp / "q.fasta", [f"q{k} len={lq}" for k in range(nq)], qs)
    return qs, db, truth


def test_cli_fails_loudly_without_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _fasta(tmp_path / "q.fasta", ["q"], ["ACDEFGHIKLMNPQRSTVWY" * 3])
    _fasta(tmp_path / "d.fasta", ["s"], ["ACDEFGHIKLMNPQRSTVWY" * 5])
    r = subprocess.run([str(_cli()), "searchp", "-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "d.fasta"), "-o",
                        str(tmp_path / "o.m8")], capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR" in r.stderr and not (tmp_path / "o.m8").exists()
    r = subprocess.run([str(_cli()), "mkindexp"], capture_output=True, text=True)
    assert r.returncode != 0 and "-d is required" in r.stderr
    r = subprocess.run([str(_cli()), "mkindexx"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown subcommand" in r.stderr


def _small_dbs(tmp):
    """A protein database, a nucleotide one (three contigs) and reads cut from it, for the index tests."""
    rng = np.random.default_rng(41)
    prots = ["".join(STD[i] for i in rng.integers(0, 20, int(rng.integers(80, 400)))) for _ in range(60)]
    _fasta(tmp / "db.fasta", [f"sp{j} protein number {j}" for j in range(len(prots))], prots)
    pq = []
    for k in range(40):
        j = int(rng.integers(0, len(prots)))
        a = int(rng.integers(0, len(prots[j]) - 70))
        r = list(prots[j][a:a + 70])
        for p_ in rng.integers(0, 70, 6):
            r[p_] = STD[int(rng.integers(0, 20))]
        pq.append("".join(r))
    _fasta(tmp / "pq.fasta", [f"q{k}" for k in range(len(pq))], pq)
    genome = ["".join("ACGT"[i] for i in rng.integers(0, 4, 6000)) for _ in range(3)]
    _fasta(tmp / "g.fasta", [f"chr{i} contig" for i in range(3)], genome)
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    for k in range(40):
        c, a = int(rng.integers(0, 3)), int(rng.integers(0, 5800))
        r = list(genome[c][a:a + 150])
        for p_ in rng.integers(0, 150, 3):
            r[p_] = "ACGT"[int(rng.integers(0, 4))]
        r = "".join(r)
        reads.append(r.translate(comp)[::-1] if k % 2 else r)
    _fasta(tmp / "r.fasta", [f"read{k}" for k in range(len(reads))], reads)
    bs = ["".join("T" if (ch == "C" and rng.random() < 0.99) else ch for ch in r) for r in reads]
    _fasta(tmp / "bs.fasta", [f"bsread{k}" for k in range(len(bs))], bs)


def test_mkindex_writes_an_index_without_a_gpu(tmp_path):
    """mkindexp | mkindexn | mkindexbs (/root/reference/src/lambda.cpp:86-88, src/mkindex_options.hpp:96-262): -d the database, -i the
    index (default DATABASE.lba, the name must end in .lba / .lta, an existing file is never overwritten); needs no device.  A
    search refuses an index of another domain with the reference's messages (src/search.cpp:189-207) before it looks for one."""
    _small_dbs(tmp_path)
    cli = str(_cli())
    run = lambda *a: subprocess.run([cli, *a], capture_output=True, text=True)
    r = run("mkindexp", "-d", str(tmp_path / "db.fasta"))
    assert r.returncode == 0 and (tmp_path / "db.fasta.lba").stat().st_size > 10_000, r.stderr
    assert "original alphabet aminoacid, translated aminoacid, reduced li10" in r.stderr and "60 sequences" in r.stderr
    assert (tmp_path / "db.fasta.lba").read_bytes()[:8] == b"LXINDEX1"
    r = run("mkindexp", "-d", str(tmp_path / "db.fasta"))
    assert r.returncode != 0 and "An output file already exists" in r.stderr
    r = run("mkindexp", "-d", str(tmp_path / "db.fasta"), "-i", str(tmp_path / "x.idx"))
    assert r.returncode != 0 and ".lba or .lta" in r.stderr
    r = run("mkindexp", "-d", str(tmp_path / "db.fasta"), "-i", str(tmp_path / "t.lba"), "-m", "acc2tax")
    assert r.returncode != 0 and "taxonomy" in r.stderr
    r = run("mkindexn", "-d", str(tmp_path / "g.fasta"), "-i", str(tmp_path / "g.lba"), "-t", "3")
    assert r.returncode == 0 and "reduced dna4" in r.stderr and "in 1 frame(s)" in r.stderr, r.stderr
    r = run("mkindexbs", "-d", str(tmp_path / "g.fasta"), "-i", str(tmp_path / "gbs.lta"))
    assert r.returncode == 0 and "reduced dna3bs" in r.stderr and "in 2 frame(s)" in r.stderr, r.stderr
    r = run("mkindexp", "-d", str(tmp_path / "g.fasta"), "-i", str(tmp_path / "gp.lba"), "-g", "4", "-r", "murphy10", "--truncate-ids")
    assert r.returncode == 0 and "original alphabet dna5, translated aminoacid, reduced murphy10, genetic code 4" in r.stderr and "in 6 frame(s)" in r.stderr
    # the table does not depend on the number of threads
    r = run("mkindexn", "-d", str(tmp_path / "g.fasta"), "-i", str(tmp_path / "g1.lba"), "-t", "1")
    assert r.returncode == 0 and (tmp_path / "g1.lba").read_bytes() == (tmp_path / "g.lba").read_bytes()
    for cmd, idx, msg in (("searchn", "db.fasta.lba", "Attempting to use protein index for nucleotide search."),
                          ("searchp", "g.lba", "Attempting to use nucleotide or bisulfite index for protein search."),
                          ("searchbs", "g.lba", "Attempting to use nucleotid index for bisulfite search."),
                          ("searchn", "gbs.lta", "Attempting to use bisulfite index for nucleotide search."),
                          ("searchbs", "db.fasta.lba", "Attempting to use protein index for bisulfite search.")):
        r = run(cmd, "-q", str(tmp_path / "r.fasta"), "-i", str(tmp_path / idx), "-o", str(tmp_path / "o.m8"))
        assert r.returncode != 0 and msg in r.stderr, (cmd, idx, r.stderr)
    (tmp_path / "junk.lba").write_bytes(bytes(range(256)) * 4)
    r = run("searchp", "-q", str(tmp_path / "pq.fasta"), "-i", str(tmp_path / "junk.lba"), "-o", str(tmp_path / "o.m8"))
    assert r.returncode != 0 and "neither a FASTA file nor an index of this front end" in r.stderr
    blob = (tmp_path / "g.lba").read_bytes()
    (tmp_path / "cut.lba").write_bytes(blob[: len(blob) // 2])
    r = run("searchn", "-q", str(tmp_path / "r.fasta"), "-i", str(tmp_path / "cut.lba"), "-o", str(tmp_path / "o.m8"))
    assert r.returncode != 0 and "index file" in r.stderr and "truncated" in r.stderr, r.stderr


@pytest.mark.gpu
def test_search_on_an_index_equals_search_on_the_fasta_file(tmp_path):
    """`search* -i INDEX` (the reference's only way, src/search_options.hpp:200) writes what `-d DB.fasta` writes, byte for byte:
    BLASTP, BLASTN, bisulfite, TBLASTN with another genetic code and reduction (both taken from the index)."""
    _small_dbs(tmp_path)
    cli = str(_cli())
    cases = (("searchp", "mkindexp", "pq.fasta", "db.fasta", [], []),
             ("searchn", "mkindexn", "r.fasta", "g.fasta", [], []),
             ("searchbs", "mkindexbs", "bs.fasta", "g.fasta", [], []),
             ("searchp", "mkindexp", "pq.fasta", "g.fasta", ["-g", "4", "-r", "murphy10"], ["--db-alphabet", "dna5"]))
    for n, (search, mk, qry, db, mkopt, fopt) in enumerate(cases):
        idx = tmp_path / f"case{n}.lba"
        r = subprocess.run([cli, mk, "-d", str(tmp_path / db), "-i", str(idx)] + mkopt, capture_output=True, text=True)
        assert r.returncode == 0 and "host thread(s)), write" in r.stderr, r.stderr
        # the word table made on the GPU (keys, one radix sort, prefix table) is the host's, bit for bit
        r = subprocess.run([cli, mk, "-d", str(tmp_path / db), "-i", str(tmp_path / f"case{n}g.lba"), "--table", "gpu"] + mkopt, capture_output=True, text=True)
        assert r.returncode == 0 and "(on the GPU), write" in r.stderr, r.stderr
        assert (tmp_path / f"case{n}g.lba").read_bytes() == idx.read_bytes()
        for ext in ("m8", "sam"):
            common = ["-q", str(tmp_path / qry), "--version-to-outputfile", "0", "-e", "10"]
            a = subprocess.run([cli, search, "-i", str(idx), "-o", str(tmp_path / f"i{n}.{ext}")] + common, capture_output=True, text=True)
            b = subprocess.run([cli, search, "-d", str(tmp_path / db), "-o", str(tmp_path / f"d{n}.{ext}")] + common + mkopt + fopt, capture_output=True, text=True)
            assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
            ia, db_ = (tmp_path / f"i{n}.{ext}").read_text(), (tmp_path / f"d{n}.{ext}").read_text()
            if ext == "sam":  # (the header names the database file)
                ia, db_ = [l for l in ia.splitlines() if not l.startswith("@")], [l for l in db_.splitlines() if not l.startswith("@")]
            assert ia == db_, (search, db, ext)
            if n < 3:
                assert len(ia) >= 20, (search, len(ia))
        assert ("tblastn" in a.stderr) == (n == 3), a.stderr


@pytest.mark.gpu
def test_gpu_seeding_equals_host_seeding(tmp_path):
    """search() of the reference (/root/reference/src/search_algo.hpp:611-762) on the GPU -- one lane per read
    (host/lx_seeding_gpu.hpp) -- against the host restatement (host/lx_seeding.hpp, itself checked against brute force): the same
    seed and match counts and byte-identical output for every program and seeding mode; a database of repeats, where the device
    declines reads (more than 32 occurrences of a word longer than the table's keys) and the host seeds them; a match buffer that is
    too small (the reads of that launch go to the host); several launches per pass."""
    import os
    import re

    _small_dbs(tmp_path)
    rng = np.random.default_rng(77)
    # BLASTX reads: coding sequence of pieces of the protein database
    prots = ["".join(rec.splitlines()[1:]) for rec in (tmp_path / "db.fasta").read_text().split(">")[1:]]
    nts = []
    for k in range(30):
        p_ = prots[int(rng.integers(0, len(prots)))]
        a = int(rng.integers(0, len(p_) - 60))
        nts.append("".join(CODONS[c][int(rng.integers(0, len(CODONS[c])))] for c in p_[a:a + 60]))
    _fasta(tmp_path / "x.fasta", [f"xr{k}" for k in range(len(nts))], nts)
    # repeats: one 60-residue motif in 45 proteins (beyond the 32 occurrences a device cursor holds past the key length) and reads of it
    motif = "".join(STD[i] for i in rng.integers(0, 20, 60))
    rep = ["".join(STD[i] for i in rng.integers(0, 20, 40)) + motif + "".join(STD[i] for i in rng.integers(0, 20, int(rng.integers(5, 60)))) for _ in range(45)]
    rep += ["".join(STD[i] for i in rng.integers(0, 20, 200)) for _ in range(20)]
    _fasta(tmp_path / "rep.fasta", [f"rp{j}" for j in range(len(rep))], rep)
    rq = [motif[int(a):int(a) + 45] for a in rng.integers(0, 15, 12)] + [rep[50][20:90], rep[55][100:170]]
    _fasta(tmp_path / "rq.fasta", [f"rq{k}" for k in range(len(rq))], rq)
    cli = str(_cli())
    cases = [("searchp", "pq.fasta", "db.fasta", [], {}), ("searchp", "pq.fasta", "db.fasta", ["-p", "sensitive"], {}),
             ("searchp", "pq.fasta", "db.fasta", ["--seed-half-exact", "0", "--seed-delta", "1", "--search0", "0"], {}),
             ("searchp", "pq.fasta", "db.fasta", ["--adaptive-seeding", "0", "-r", "murphy10"], {}),
             ("searchn", "r.fasta", "g.fasta", [], {}), ("searchn", "r.fasta", "g.fasta", ["-p", "fast"], {}),
             ("searchbs", "bs.fasta", "g.fasta", [], {}),
             ("searchp", "x.fasta", "db.fasta", [], {}),                       # BLASTX
             ("searchp", "pq.fasta", "g.fasta", ["--db-alphabet", "dna5"], {}),  # TBLASTN
             ("searchp", "rq.fasta", "rep.fasta", ["-n", "100"], {}),          # repeats: the device declines reads
             ("searchp", "pq.fasta", "db.fasta", [], {"LAMBDA3_SEED_CAP": "7"}),  # match buffer too small: the launch's reads go to the host
             ("searchn", "r.fasta", "g.fasta", [], {"LAMBDA3_SEED_LAUNCH": "7"})]  # several launches per pass
    for n, (cmd, qry, db, extra, env) in enumerate(cases):
        outs, counts, err = {}, {}, {}
        for where in ("gpu", "host"):
            out = tmp_path / f"s{n}.{where}.m8"
            r = subprocess.run([cli, cmd, "-q", str(tmp_path / qry), "-d", str(tmp_path / db), "-o", str(out), "-e", "10", "--seeding", where] + extra,
                               capture_output=True, text=True, env={**os.environ, **env})
            assert r.returncode == 0, r.stderr
            outs[where], err[where] = out.read_bytes(), r.stderr
            counts[where] = re.search(r"seeds (\d+) -> promising (\d+) -> windows (\d+)", r.stderr).groups()
            assert f"seeding on the {'GPU' if where == 'gpu' else 'host'}" in r.stderr
        assert counts["gpu"] == counts["host"], (cmd, extra, counts)
        assert outs["gpu"] == outs["host"], (cmd, extra)
        left = re.search(r"\[(\d+) read\(s\) and (\d+) launch\(es\) left to the host\]", err["gpu"]).groups()
        if db == "rep.fasta":
            assert int(left[0]) >= 10 and len(outs["gpu"].splitlines()) >= 100, (left, err["gpu"])
        elif "LAMBDA3_SEED_CAP" in env:
            assert int(left[1]) >= 1 and int(left[0]) >= 40, err["gpu"]
        else:
            assert left == ("0", "0"), err["gpu"]
        if n in (0, 4, 6, 7):
            assert int(counts["gpu"][1]) >= 20, (cmd, counts)


@pytest.mark.gpu
def test_searchp_config1_end_to_end(tmp_path, oracle):
    qs, db, truth = _make_config1(tmp_path)
    out = tmp_path / "out.m8"
    r = subprocess.run([str(_cli()), "searchp", "-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "db.fasta"), "-o",
                        str(out), "-t", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    print(r.stderr.strip())
    rows = [l.split("\t") for l in out.read_text().splitlines()]
    assert len(rows) > 150 and all(len(x) == 12 for x in rows)
    best = {}
    for x in rows:
        best.setdefault(x[0], x)  # records are sorted by bit score within a query: the first one is the best
    found = sum(1 for k, j in truth.items() if best.get(f"q{k}", [None, None])[1] == f"sp{j}")
    # the reference's seeding (Li-10, exact 10/5 first, half-exact 11/3 for reads without a result, adaptive elongation) at
    # 25 % substitutions + 2 % indels: nearly all planted homologs are found; the exact pre-search alone finds fewer
    assert found >= 0.9 * len(truth), (found, len(truth))
    r0 = subprocess.run([str(_cli()), "searchp", "-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "db.fasta"), "-o",
                         str(tmp_path / "exact.m8"), "--search0", "0", "--seed-length", "10", "--seed-offset", "5", "--seed-delta", "0"],
                        capture_output=True, text=True)
    assert r0.returncode == 0, r0.stderr
    best0 = {}
    for l in (tmp_path / "exact.m8").read_text().splitlines():
        x = l.split("\t")
        best0.setdefault(x[0], x)
    found0 = sum(1 for k, j in truth.items() if best0.get(f"q{k}", [None, None])[1] == f"sp{j}")
    assert found0 <= found
    # per query: at most 25 hits, descending bit score
    per = {}
    for x in rows:
        per.setdefault(x[0], []).append(float(x[11]))
    assert all(len(v) <= 25 and v == sorted(v, reverse=True) for v in per.values())
    # every reported HSP re-scores on the oracle: the local alignment of exactly the reported ranges has that bit score
    sc_p = capi.builtin_scoring(62)
    osc = oracle_lib.scoring_from(sc_p)
    ka = capi.karlin_params(62)
    rank = {c: i for i, c in enumerate(ORDER)}
    enc = lambda s: np.array([rank[c] for c in s], dtype=np.uint8)
    for x in rows[:400]:
        k, j = int(x[0][1:]), int(x[1][2:])
        qa, qb, sa, sb = int(x[6]) - 1, int(x[7]), int(x[8]) - 1, int(x[9])
        s_, qe, se = oracle.score(enc(qs[k][qa:qb]), enc(db[j][sa:sb]), osc)
        bits = (ka.lambda_ * s_ - math.log(ka.K)) / math.log(2)
        assert abs(bits - float(x[11])) <= 0.051, (x, s_, bits)
        assert (qe, se) == (qb - qa, sb - sa)
        assert float(x[10]) <= 1e-2
    # SAM output of the same search
    r = subprocess.run([str(_cli()), "searchp", "-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "db.fasta"), "-o",
                        str(tmp_path / "out.sam")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sam = [l for l in (tmp_path / "out.sam").read_text().splitlines() if not l.startswith("@")]
    assert len(sam) == len(rows) and sam[0].split("\t")[1] == "0"


@pytest.mark.gpu
def test_searchn_both_strands(tmp_path):
    rng = np.random.default_rng(5)
    genome = ["".join("ACGT"[i] for i in rng.integers(0, 4, 20000)) for _ in range(3)]
    comp = str.maketrans("ACGT", "TGCA")
    reads, truth = [], []
    for k in range(60):
        c, a = int(rng.integers(0, 3)), int(rng.integers(0, 19800))
        r = list(genome[c][a:a + 150])
        for p in rng.integers(0, 150, 4):
            r[p] = "ACGT"[int(rng.integers(0, 4))]
        r = "".join(r)
        minus = k % 2 == 1
        reads.append(r.translate(comp)[::-1] if minus else r)
        truth.append((c, a, minus))
    _fasta(tmp_path / "g.fasta", [f"chr{i}" for i in range(3)], genome)
    _fasta(tmp_path / "r.fasta", [f"read{k}" for k in range(60)], reads)
    out = tmp_path / "o.m8"
    r = subprocess.run([str(_cli()), "searchn", "-q", str(tmp_path / "r.fasta"), "-d", str(tmp_path / "g.fasta"), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    best = {}
    for l in out.read_text().splitlines():
        x = l.split("\t")
        best.setdefault(x[0], x)
    ok = 0
    for k, (c, a, minus) in enumerate(truth):
        x = best.get(f"read{k}")
        if x and x[1] == f"chr{c}":
            ss, se = int(x[8]), int(x[9])
            if (ss > se) == minus and abs(min(ss, se) - 1 - a) <= 10:
                ok += 1
    assert ok >= 57, ok


CODONS = {"A": ["GCT", "GCC", "GCA", "GCG"], "C": ["TGT", "TGC"], "D": ["GAT", "GAC"], "E": ["GAA", "GAG"], "F": ["TTT", "TTC"],
          "G": ["GGT", "GGC", "GGA", "GGG"], "H": ["CAT", "CAC"], "I": ["ATT", "ATC", "ATA"], "K": ["AAA", "AAG"],
          "L": ["TTA", "TTG", "CTT", "CTC", "CTA", "CTG"], "M": ["ATG"], "N": ["AAT", "AAC"], "P": ["CCT", "CCC", "CCA", "CCG"],
          "Q": ["CAA", "CAG"], "R": ["CGT", "CGC", "CGA", "CGG", "AGA", "AGG"], "S": ["TCT", "TCC", "TCA", "TCG", "AGT", "AGC"],
          "T": ["ACT", "ACC", "ACA", "ACG"], "V": ["GTT", "GTC", "GTA", "GTG"], "W": ["TGG"], "Y": ["TAT", "TAC"]}


@pytest.mark.gpu
def test_searchp_blastx_frames_and_coordinates(tmp_path):
    """searchp with nucleotide queries = BLASTX: reads that encode a database protein
    segment in a known frame / strand must come back with that subject, 100 % identity, the frame's nucleotide
    coordinates (start > end on the minus strand) and the protein coordinates of the segment."""
    rng = np.random.default_rng(11)
    db = ["".join(STD[i] for i in rng.integers(0, 20, int(L))) for L in rng.integers(120, 400, 300)]
    comp = str.maketrans("ACGT", "TGCA")
    reads, truth = [], []
    for k in range(48):
        j = int(rng.integers(0, len(db)))
        a = int(rng.integers(0, len(db[j]) - 60))
        seg = db[j][a:a + 50]
        nt = "".join(CODONS[c][int(rng.integers(0, len(CODONS[c])))] for c in seg)
        left = int(rng.integers(0, 3)) + 3 * int(rng.integers(0, 3))   # frame shift 0..2 (+ whole codons of noise)
        right = int(rng.integers(0, 9))
        # the flanks end in a stop codon next to the segment, so that the alignment cannot run on into them
        lf = "".join("ACGT"[i] for i in rng.integers(0, 4, left))
        lf = (lf[:-3] + "TAA") if left >= 3 else lf
        rt = "TAA" + "".join("ACGT"[i] for i in rng.integers(0, 4, right))
        read = lf + nt + rt
        minus = k % 2 == 1
        L = len(read)
        if minus:
            qstart, qend = L - left, L - left - 150 + 1      # as reported on the original (reverse-complemented) read
            read = read.translate(comp)[::-1]
        else:
            qstart, qend = left + 1, left + 150
        reads.append(read)
        truth.append((j, a, qstart, qend))
    _fasta(tmp_path / "db.fasta", [f"sp{j}" for j in range(len(db))], db)
    _fasta(tmp_path / "r.fasta", [f"read{k}" for k in range(len(reads))], reads)
    out = tmp_path / "o.m8"
    r = subprocess.run([str(_cli()), "searchp", "-q", str(tmp_path / "r.fasta"), "-d", str(tmp_path / "db.fasta"), "-o", str(out),
                        "--seed-offset", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "blastx" in r.stderr
    best = {}
    for l in out.read_text().splitlines():
        x = l.split("\t")
        best.setdefault(x[0], x)
    assert len(best) == len(reads)
    exact = 0
    for k, (j, a, qstart, qend) in enumerate(truth):
        x = best[f"read{k}"]
        assert x[1] == f"sp{j}" and float(x[2]) >= 90.0, x
        qs_, qe_, ss_, se_ = int(x[6]), int(x[7]), int(x[8]), int(x[9])
        if int(x[3]) == 50 and float(x[2]) == 100.0:
            exact += 1
            assert (qs_, qe_) == (qstart, qend), (x, qstart, qend)
            assert (ss_, se_) == (a + 1, a + 50), x
        else:
            # a strong residue behind the stop codon can pull the local alignment a few columns further: same frame
            # and strand, planted interval contained
            assert (qs_ < qe_) == (qstart < qend) and (qs_ - qstart) % 3 == 0 and (qe_ - qend) % 3 == 0, (x, qstart, qend)
            assert min(qs_, qe_) <= min(qstart, qend) and max(qs_, qe_) >= max(qstart, qend)
            assert abs(qs_ - qstart) <= 12 and abs(qe_ - qend) <= 12 and ss_ <= a + 1 and se_ >= a + 50
        assert abs(qe_ - qs_) + 1 == 3 * (int(x[3]) - int(x[5]) * 0) or int(x[5]) > 0  # ungapped: 3 nt per column
    assert exact >= 0.7 * len(truth), exact
    # the same search as SAM: nucleotide-space CIGAR whose clips and runs add up to the read, strand flag, covered SEQ
    import re
    r = subprocess.run([str(_cli()), "searchp", "-q", str(tmp_path / "r.fasta"), "-d", str(tmp_path / "db.fasta"), "-o",
                        str(tmp_path / "o.sam"), "--seed-offset", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    first = {}
    for l in (tmp_path / "o.sam").read_text().splitlines():
        if not l.startswith("@"):
            x = l.split("\t")
            first.setdefault(x[0], x)
    assert len(first) == len(reads)
    for k, (j, a, qstart, qend) in enumerate(truth):
        x = first[f"read{k}"]
        assert x[2] == f"sp{j}" and (int(x[1]) & 16 != 0) == (qstart > qend), x
        el = [(int(n), c) for n, c in re.findall(r"(\d+)([MIDSH])", x[5])]
        assert sum(n for n, c in el if c in "MISH") == len(reads[k]), x      # every nucleotide of the read is accounted for
        assert all(n % 3 == 0 for n, c in el if c in "MIDS")                    # codon-sized runs; the hard clips hold the frame clips
        assert len(x[9]) == sum(n for n, c in el if c in "MIS")                 # SEQ = the read minus the hard clips
        frame = int([t for t in x if t.startswith("qf:i:")][0][5:])
        # hard clips (the reference's default): SEQ is the matched part of the strand that was read; its first base stands behind
        # the clip in front of the alignment -- the cigar's first element, or its last on the minus strand (the list is reversed there)
        if frame > 0:
            lead = el[0][0] if el[0][1] == "H" else 0
            assert lead % 3 == frame - 1 and x[9] == reads[k][lead: lead + len(x[9])]
        else:
            lead = el[-1][0] if el[-1][1] == "H" else 0
            assert lead % 3 == -frame - 1 and x[9] == reads[k].translate(comp)[::-1][lead: lead + len(x[9])]


@pytest.mark.gpu
def test_searchp_tblastn_translated_subjects(tmp_path):
    """searchp against a nucleotide database = TBLASTN: contigs that carry the coding sequence of a query protein in a
    known frame / strand come back with the contig as subject and the nucleotide coordinates of the coding segment
    (start > end on the minus strand)."""
    rng = np.random.default_rng(23)
    comp = str.maketrans("ACGT", "TGCA")
    prots, contigs, truth = [], [], []
    for k in range(24):
        p = "".join(STD[i] for i in rng.integers(0, 20, 70))
        nt = "".join(CODONS[c][int(rng.integers(0, len(CODONS[c])))] for c in p)
        left, right = int(rng.integers(30, 200)), int(rng.integers(30, 200))
        lf = "".join("ACGT"[i] for i in rng.integers(0, 4, left))[:-3] + "TAA"
        rt = "TAA" + "".join("ACGT"[i] for i in rng.integers(0, 4, right))
        contig = lf + nt + rt
        L = len(contig)
        if k % 2:
            sstart, send = L - left, L - left - 210 + 1
            contig = contig.translate(comp)[::-1]
        else:
            sstart, send = left + 1, left + 210
        prots.append(p)
        contigs.append(contig)
        truth.append((sstart, send))
    _fasta(tmp_path / "q.fasta", [f"prot{k}" for k in range(len(prots))], prots)
    _fasta(tmp_path / "db.fasta", [f"contig{k}" for k in range(len(contigs))], contigs)
    out = tmp_path / "o.m8"
    r = subprocess.run([str(_cli()), "searchp", "-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "db.fasta"), "-o", str(out),
                        "--seed-offset", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "tblastn" in r.stderr
    best = {}
    for l in out.read_text().splitlines():
        x = l.split("\t")
        best.setdefault(x[0], x)
    assert len(best) == len(prots)
    exact = 0
    for k, (sstart, send) in enumerate(truth):
        x = best[f"prot{k}"]
        assert x[1] == f"contig{k}" and float(x[2]) >= 90.0, x
        ss_, se_ = int(x[8]), int(x[9])
        assert (ss_ < se_) == (sstart < send) and (ss_ - sstart) % 3 == 0 and (se_ - send) % 3 == 0, (x, sstart, send)
        if int(x[3]) == 70 and float(x[2]) == 100.0:
            exact += 1
            assert (ss_, se_) == (sstart, send) and (int(x[6]), int(x[7])) == (1, 70), (x, sstart, send)
    assert exact >= 0.7 * len(truth), exact


@pytest.mark.gpu
def test_devices_split_gives_identical_output(tmp_path):
    """The thread / device split of realMain (/root/reference/src/search.cpp:379-385): two handles on one GPU (`--devices 0,0`),
    one / three / all granted host threads for the word table and the seeding (`-t`) write byte-identical BLAST-tabular and SAM
    files."""
    _make_config1(tmp_path, nq=300, ndb=800)
    db, qry = tmp_path / "db.fasta", tmp_path / "q.fasta"
    outs = {}
    for tag, extra in (("one", ["--devices", "0", "-t", "1"]), ("two", ["--devices", "0,0"]), ("three", ["--devices", "0", "-t", "3"]),
                       ("two-five", ["--devices", "0,0", "-t", "5"]), ("eight", ["--devices", "0,0,0,0,0,0,0,0"])):
        for ext in ("m8", "sam"):
            out = tmp_path / f"{tag}.{ext}"
            # (--version-to-outputfile 0, as the reference's own CLI tests pass it: no @PG line with the command line in the SAM header)
            r = subprocess.run([str(_cli()), "searchp", "-q", str(qry), "-d", str(db), "-o", str(out), "--version-to-outputfile", "0"] + extra,
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert ("8 handle(s)" in r.stderr) == (tag == "eight")
            assert ("2 handle(s)" in r.stderr) == tag.startswith("two") and ("3 host thread(s)" in r.stderr) == (tag == "three"), r.stderr
            assert ("1 host thread(s)" in r.stderr) == (tag == "one") and "lambda3 times [ms]" in r.stderr, r.stderr
            outs[(tag, ext)] = out.read_bytes()
    assert len(outs[("one", "m8")].splitlines()) >= 50
    for ext in ("m8", "sam"):
        assert outs[("one", ext)] == outs[("two", ext)] == outs[("three", ext)] == outs[("two-five", ext)] == outs[("eight", ext)]
    r = subprocess.run([str(_cli()), "searchp", "-q", str(qry), "-d", str(db), "-o", str(tmp_path / "x.m8"), "--devices", "7"], capture_output=True, text=True)
    assert r.returncode != 0 and "device_id 7 out of range" in r.stderr


@pytest.mark.gpu
def test_searchbs_both_strands_and_conversions(tmp_path, oracle):
    """searchbs (/root/reference/src/lambda.cpp:103; frames src/search_datastructures.hpp:380-385, schemes
    src/bisulfite_scoring.hpp:67-93): bisulfite-converted reads -- C->T on the read's own strand (forward scheme, even subject
    frame) or G->A (reverse scheme, odd subject frame), from either genome strand -- come back on the right chromosome, strand
    and position; every written score is the oracle's for the frame pair the record names."""
    from lambda_amd import capi
    from tests import oracle_lib

    rng = np.random.default_rng(9)
    genome = ["".join("ACGT"[i] for i in rng.integers(0, 4, 30000)) for _ in range(2)]
    comp = str.maketrans("ACGT", "TGCA")
    reads, truth = [], []
    for k in range(80):
        c, a = int(rng.integers(0, 2)), int(rng.integers(0, 29800))
        frag = genome[c][a:a + 150]
        minus, ga = (k % 2 == 1), (k % 4 >= 2)
        if minus:
            frag = frag.translate(comp)[::-1]
        r = list(frag)
        for i, ch in enumerate(r):  # 99 % conversion
            if not ga and ch == "C" and rng.random() < 0.99:
                r[i] = "T"
            if ga and ch == "G" and rng.random() < 0.99:
                r[i] = "A"
        for p in rng.integers(0, 150, 2):
            r[p] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(r))
        truth.append((c, a, minus, ga))
    _fasta(tmp_path / "g.fasta", [f"chr{i}" for i in range(2)], genome)
    _fasta(tmp_path / "r.fasta", [f"read{k}" for k in range(80)], reads)
    out = tmp_path / "o.m8"
    r = subprocess.run([str(_cli()), "searchbs", "-q", str(tmp_path / "r.fasta"), "-d", str(tmp_path / "g.fasta"), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "searchbs (blastn" in r.stderr
    best = {}
    for l in out.read_text().splitlines():
        x = l.split("\t")
        best.setdefault(x[0], x)
    ok = 0
    for k, (c, a, minus, ga) in enumerate(truth):
        x = best.get(f"read{k}")
        if x and x[1] == f"chr{c}":
            ss, se = int(x[8]), int(x[9])
            if (ss > se) == minus and abs(min(ss, se) - 1 - a) <= 10 and float(x[10]) <= 1e-9:
                ok += 1
    assert ok >= 74, ok
    # the SAM writer on the same search: flags carry the strand, the second bisulfite duplicate hard-clips one base
    # (blastMatchOneCigar takes |qFrameShift| - 1 for every program, src/search_output.hpp:126)
    sam = tmp_path / "o.sam"
    r = subprocess.run([str(_cli()), "searchbs", "-q", str(tmp_path / "r.fasta"), "-d", str(tmp_path / "g.fasta"), "-o", str(sam)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    recs = [l.split("\t") for l in sam.read_text().splitlines() if not l.startswith("@")]
    assert len(recs) >= 74 and {int(x[1]) & 16 for x in recs} == {0, 16}


def test_cli_rejects_bad_option_values(tmp_path):
    """Option validation happens before any device is touched (the reference's sharg validators, src/search_options.hpp:224-560)."""
    _fasta(tmp_path / "q.fasta", ["q"], ["ACDEFGHIKLMNPQRSTVWY" * 3])
    _fasta(tmp_path / "d.fasta", ["s"], ["ACDEFGHIKLMNPQRSTVWY" * 5])
    base = [str(_cli()), "searchp", "-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "d.fasta"), "-o", str(tmp_path / "o.m8")]
    for bad, msg in ((["-s", "50"], "--scoring-scheme takes 45, 62 or 80"), (["-p", "quick"], "--profile takes"), (["--sam-bam-clip", "medium"], "hard or soft"),
                     (["--bit-score", "2000"], "not in range"), (["--score-match", "3"], "unknown option --score-match"),  # (nucleotide domains only)
                     (["--sam-bam-seq", "often"], "always, uniq or never"), (["--seed-delta", "7"], "out of range")):
        r = subprocess.run(base + bad, capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr, (bad, r.stderr)
    r = subprocess.run([str(_cli()), "searchn"] + base[2:] + ["-s", "62"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown option -s" in r.stderr  # (protein domain only)


@pytest.mark.gpu
def test_cli_profiles_scoring_and_output_options(tmp_path, oracle):
    """The rest of the search command line (src/search_options.hpp:224-681): profiles overwrite the seeding parameters, the scoring options
    reach the DP (every HSP re-scored on the oracle under BLOSUM80 11/2), --bit-score filters, --output-columns / .m9 comments and footer,
    SAM tags / clipping."""
    qs, db, truth = _make_config1(tmp_path, nq=200, ndb=2000)
    common = ["-q", str(tmp_path / "q.fasta"), "-d", str(tmp_path / "db.fasta")]
    out = tmp_path / "o.m9"
    r = subprocess.run([str(_cli()), "searchp"] + common + ["-o", str(out), "-p", "sensitive", "-s", "80", "--score-gap", "-2", "--score-gap-open", "-9",
                        "--bit-score", "40", "--output-columns", "std score qlen slen", "--version-to-outputfile", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = out.read_text().splitlines()
    assert lines[0] == "# BLASTP 2.2.26+" and lines[2] == f"# Database: {tmp_path / 'db.fasta'}"
    assert lines[3].endswith("evalue, bit score, score, query length, subject length")
    rows = [l.split("\t") for l in lines if not l.startswith("#")]
    nrec = sum(1 for l in lines if l.startswith("# Query:"))
    assert lines[-1] == f"# BLAST processed {nrec} queries" and len(rows) > 30 and all(len(x) == 15 for x in rows)
    assert all(float(x[11]) >= 40 for x in rows)
    sc_p = capi.builtin_scoring(80, gap_open=-9, gap_extend=-2)  # (lambda's option values, as on the command line)
    osc = oracle_lib.scoring_from(sc_p)
    rank = {c: i for i, c in enumerate(ORDER)}
    enc = lambda s: np.array([rank[c] for c in s], dtype=np.uint8)
    for x in rows[:200]:
        k, j = int(x[0][1:]), int(x[1][2:])
        qa, qb, sa, sb = int(x[6]) - 1, int(x[7]), int(x[8]) - 1, int(x[9])
        s_, qe, se = oracle.score(enc(qs[k][qa:qb]), enc(db[j][sa:sb]), osc)
        assert s_ == int(x[12]) and (qe, se) == (qb - qa, sb - sa), x
        assert int(x[13]) == len(qs[k]) and int(x[14]) == len(db[j])
    # the fast profile finds no more than the default search; both run
    n = {}
    for prof in ("none", "fast"):
        o = tmp_path / f"{prof}.m8"
        r = subprocess.run([str(_cli()), "searchp"] + common + ["-o", str(o), "-p", prof], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        n[prof] = len({l.split("\t")[0] for l in o.read_text().splitlines()})
    assert 0 < n["fast"] <= n["none"]
    # SAM: tags and clipping
    sam = tmp_path / "o.sam"
    r = subprocess.run([str(_cli()), "searchp"] + common + ["-o", str(sam), "--sam-bam-tags", "AS ar OC qs IH", "--sam-bam-seq", "always",
                        "--sam-with-refheader", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = sam.read_text().splitlines()
    assert sum(1 for l in txt if l.startswith("@SQ")) == 2000 and any(l.startswith("@PG\tID:lambda") and "searchp -q" in l for l in txt)
    recs = [l.split("\t") for l in txt if not l.startswith("@")]
    assert len(recs) > 30
    for f in recs[:100]:
        tags = dict((t[:2], t[5:]) for t in f[11:])
        assert set(tags) == {"AS", "ar", "OC", "qs", "IH"} and f[5] == "*" and f[9] == "*"
        k = int(f[0][1:])
        # hard clips: the protein cigar's M / I columns are the residues of qs, and qs is a piece of the query
        import re
        ops = re.findall(r"(\d+)([MIDHS])", tags["OC"])
        assert sum(int(c) for c, o in ops if o in "MI") == len(tags["qs"]) and tags["qs"] in qs[k]
        assert sum(int(c) for c, o in ops if o in "MIH") == len(qs[k])

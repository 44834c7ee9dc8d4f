"""End-to-end run of `lambda3 searchn` at scale: reads of 150 bp cut from a random genome (both strands, 2 % substitutions, a few
indels; a quarter of the reads random) against the genome's contigs.  Prints the front end's summary, its stage times and a hash of
the output -- with `--seeding host` after the size arguments the same run on the host threads, for comparison.
    python tools/cli_scale_nucl.py [n_reads] [genome_Mbp] [--bs] [front-end options ...]"""
import hashlib, subprocess, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lambda_amd import build

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
mbp = float(sys.argv[2]) if len(sys.argv) > 2 else 50
extra = sys.argv[3:]
bs = "--bs" in extra  # bisulfite reads (99 % C->T on the read's strand) through searchbs
if bs:
    extra.remove("--bs")
rng = np.random.default_rng(0x1A3BDA03)
NT = np.frombuffer(b"ACGT", dtype=np.uint8)
comp = np.zeros(256, np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    comp[a] = b
contigs = [NT[rng.integers(0, 4, int(mbp * 1e6 / 20))] for _ in range(20)]
tmp = Path(tempfile.mkdtemp(prefix="lx_cli_scale_nucl_"))
with open(tmp / "g.fasta", "wb") as f:
    for j, c in enumerate(contigs):
        f.write(b">chr%d\n" % j + c.tobytes() + b"\n")
with open(tmp / "r.fasta", "wb") as f:
    for k in range(n_reads):
        if k % 4 == 3:
            r = NT[rng.integers(0, 4, 150)]
        else:
            c = contigs[int(rng.integers(0, 20))]
            a = int(rng.integers(0, len(c) - 160))
            r = c[a:a + 152].copy()
            sub = rng.random(len(r)) < 0.02
            r[sub] = NT[rng.integers(0, 4, int(sub.sum()))]
            if rng.random() < 0.2:
                r = np.delete(r, int(rng.integers(10, 140)))
            r = r[:150]
            if k % 2:
                r = comp[r[::-1]]
            if bs:
                conv = (r == ord("C")) & (rng.random(len(r)) < 0.99)
                r = r.copy()
                r[conv] = ord("T")
        f.write(b">read%d\n" % k + r.tobytes() + b"\n")
cli = build.build_cli()
t0 = time.perf_counter()
r = subprocess.run([str(cli), "searchbs" if bs else "searchn", "-q", str(tmp / "r.fasta"), "-d", str(tmp / "g.fasta"), "-o", str(tmp / "out.m8")] + extra, capture_output=True, text=True)
dt = time.perf_counter() - t0
print(r.stderr.strip())
print("output sha256", hashlib.sha256(open(tmp / "out.m8", "rb").read()).hexdigest()[:16] if r.returncode == 0 else "-")
print(f"rc {r.returncode}; {n_reads} reads x {mbp:.0f} Mbp: {dt:.2f} s wall")

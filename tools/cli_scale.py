"""End-to-end run of the lambda3 front end at a size where the stages can be told apart: SURVEY.md section 8(d) config 1's
recipe (log-normal protein lengths, 30 % of the queries mutated copies of database regions) scaled up, default 100 000 queries
of 150 aa against 100 000 proteins.  Prints the front end's own summary and its stage times (stderr of the CLI).
    python tools/cli_scale.py [n_queries] [n_db] [threads (0 = default)] [front-end options ...]"""
import math, subprocess, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from lambda_amd import build

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ndb = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
threads = sys.argv[3] if len(sys.argv) > 3 else "0"
extra = sys.argv[4:]  # further options for the front end, e.g. --seeding gpu
lq = 150
STD = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
rng = np.random.default_rng(0x1A3BDA01)
lens = np.clip(np.exp(rng.normal(math.log(300), 0.6, ndb)).astype(int), 50, 2000)
off = np.concatenate([[0], np.cumsum(lens)])
res = STD[rng.integers(0, 20, int(off[-1]))]
tmp = Path(tempfile.mkdtemp(prefix="lx_cli_scale_"))
with open(tmp / "db.fasta", "wb") as f:
    for j in range(ndb):
        f.write(b">sp%d protein\n" % j + res[off[j]:off[j + 1]].tobytes() + b"\n")
planted = 0
with open(tmp / "q.fasta", "wb") as f:
    for k in range(nq):
        if rng.random() < 0.3:
            j = int(rng.integers(0, ndb))
            while lens[j] < lq + 10:
                j = int(rng.integers(0, ndb))
            a = int(rng.integers(0, lens[j] - lq - 5))
            src = res[off[j] + a: off[j] + a + lq + 5].copy()
            r = rng.random(len(src))
            sub = rng.random(len(src)) < 0.25
            src[sub] = STD[rng.integers(0, 20, int(sub.sum()))]
            keep = r >= 0.01                                     # 1 % deletions
            out = src[keep]
            ins = np.flatnonzero(rng.random(len(out)) < 0.01)    # 1 % insertions
            out = np.insert(out, ins, STD[rng.integers(0, 20, len(ins))])[:lq]
            if len(out) < lq:
                out = np.concatenate([out, np.full(lq - len(out), ord("A"), np.uint8)])
            planted += 1
        else:
            out = STD[rng.integers(0, 20, lq)]
        f.write(b">q%d\n" % k + out.tobytes() + b"\n")
cli = build.build_cli()
t0 = time.perf_counter()
import os, shlex
if "--via-index" in extra:  # index once (lambda3 mkindexp), then search on the index file
    extra.remove("--via-index")
    t1 = time.perf_counter()
    m = subprocess.run([str(cli), "mkindexp", "-d", str(tmp / "db.fasta"), "-i", str(tmp / "db.lba")], capture_output=True, text=True)
    print(m.stderr.strip(), f"\nmkindexp: {time.perf_counter() - t1:.2f} s wall")
    dbargs = ["-i", str(tmp / "db.lba")]
else:
    dbargs = ["-d", str(tmp / "db.fasta")]
wrap = shlex.split(os.environ.get("LX_CLI_WRAP", ""))  # e.g. "rocprofv3 --kernel-trace --stats -d DIR -o cli --" (per-kernel times of the run)
r = subprocess.run(wrap + [str(cli), "searchp", "-q", str(tmp / "q.fasta"), *dbargs, "-o", str(tmp / "out.m8")] + (["-t", threads] if threads != "0" else []) + extra,
                   capture_output=True, text=True)
dt = time.perf_counter() - t0
print(r.stderr.strip())
rows = sum(1 for _ in open(tmp / "out.m8")) if r.returncode == 0 else -1
import hashlib
print("output sha256", hashlib.sha256(open(tmp / "out.m8", "rb").read()).hexdigest()[:16] if r.returncode == 0 else "-")
print(f"rc {r.returncode}; {nq} queries ({planted} planted) x {ndb} subjects ({int(off[-1])} residues): {dt:.2f} s wall, {rows} records")

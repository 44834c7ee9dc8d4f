"""Development aid: odd inputs through the lambda3 front end -- lower case, unknown letters, stops, empty and very short sequences,
blank lines, Windows line ends, duplicate ids, reads longer than subjects -- must end with exit code 0 or the front end's own
error (255), never with a signal.    python tools/dev/cli_fuzz.py [seconds] [seed]"""
import subprocess, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from lambda_amd import build

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cli = str(build.build_cli())
tmp = Path(tempfile.mkdtemp(prefix="lx_cli_fuzz_"))
AA, NT = "ACDEFGHIKLMNPQRSTVWY", "ACGT"


def seq(alpha, n, odd):
    s = [alpha[i] for i in rng.integers(0, len(alpha), n)]
    if odd and n:
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, n))] = str(rng.choice(list("XxNn*-BZJUO.acgt?1")))
        if rng.random() < 0.3:
            s = [c.lower() for c in s]
    return "".join(s)


def fasta(path, seqs, odd):
    eol = "\r\n" if odd and rng.random() < 0.2 else "\n"
    with open(path, "w", newline="") as f:
        for k, s in enumerate(seqs):
            name = f"s{k if rng.random() > 0.05 else 0} desc {k}" if rng.random() > 0.05 else ""
            f.write(">" + name + eol)
            w = int(rng.integers(20, 90))
            for a in range(0, len(s), w):
                f.write(s[a:a + w] + eol)
            if odd and rng.random() < 0.1:
                f.write(eol)


t0, runs, crashes = time.time(), 0, 0
while time.time() - t0 < budget:
    cmd = str(rng.choice(["searchp", "searchn", "searchbs"]))
    alpha_db = NT if cmd != "searchp" or rng.random() < 0.2 else AA
    alpha_q = NT if cmd != "searchp" or rng.random() < 0.2 else AA
    db = [seq(alpha_db, int(rng.choice([0, 1, 5, 30, 200, 900])), True) for _ in range(int(rng.integers(1, 12)))]
    src = "".join(db)
    qs = []
    for _ in range(int(rng.integers(1, 10))):
        n = int(rng.choice([0, 1, 3, 9, 12, 40, 150, 400]))
        if len(src) > n and rng.random() < 0.6 and alpha_q == alpha_db:
            a = int(rng.integers(0, len(src) - n + 1))
            qs.append(src[a:a + n])
        else:
            qs.append(seq(alpha_q, n, True))
    fasta(tmp / "q.fa", qs, True)
    fasta(tmp / "d.fa", db, True)
    ext = str(rng.choice(["m8", "m9", "sam"]))
    extra = []
    if rng.random() < 0.3:
        extra += ["-e", "100"]
    if rng.random() < 0.2:
        extra += ["-p", str(rng.choice(["fast", "sensitive", "pairs-default"]))]
    if rng.random() < 0.2:
        extra += ["--seed-half-exact", "0"]
    if rng.random() < 0.2:
        extra += ["-t", str(int(rng.integers(1, 5)))]
    via_index = rng.random() < 0.3
    if via_index:
        (tmp / "d.lba").unlink(missing_ok=True)
        r = subprocess.run([cli, "mkindex" + cmd[6:], "-d", str(tmp / "d.fa"), "-i", str(tmp / "d.lba")], capture_output=True, text=True, timeout=120)
        if r.returncode < 0:
            crashes += 1
            print("SIGNAL", r.returncode, "mkindex", cmd, file=sys.stderr)
            (tmp / f"crash{crashes}_d.fa").write_bytes((tmp / "d.fa").read_bytes())
        if r.returncode != 0:
            via_index = False
    r = subprocess.run([cli, cmd, "-q", str(tmp / "q.fa"), "-i" if via_index else "-d", str(tmp / ("d.lba" if via_index else "d.fa")), "-o", str(tmp / f"o.{ext}")] + extra,
                       capture_output=True, text=True, timeout=120)
    runs += 1
    if r.returncode not in (0, 255):
        crashes += 1
        print("SIGNAL / odd exit", r.returncode, cmd, extra, r.stderr[-300:], file=sys.stderr)
        (tmp / f"crash{crashes}_q.fa").write_bytes((tmp / "q.fa").read_bytes())
        (tmp / f"crash{crashes}_d.fa").write_bytes((tmp / "d.fa").read_bytes())
print(f"{runs} runs, {crashes} crashes ({tmp})")
sys.exit(1 if crashes else 0)

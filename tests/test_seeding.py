"""CPU test of the front end's seeding stage (lambda_amd/csrc/host/lx_seeding.hpp): a host-only
C++ check program, compiled here with g++, compares the sorted word table with brute force (exact and half-exact seeds of
the reference, src/search_algo.hpp:505-604) and pins the Li-10 groups."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_seeding_against_brute_force(tmp_path):
    exe = tmp_path / "seeding_check"
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", str(ROOT / "tests" / "seeding_check.cpp"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "seeding check: ok" in r.stdout, r.stdout + r.stderr

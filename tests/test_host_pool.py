"""The library's host threads (lambda_amd/csrc/lx_host_pool.cpp): every part of every loop runs exactly once, with one caller
and with eight callers at once (one handle per host thread -- the reference's model, /root/reference/src/search.cpp:379-385);
the round-5 race (a pool that grows between loops); the width follows the CPUs the process is granted and LOCAL_WORLD_SIZE.
The harness is tests/native/host_pool_stress.cpp, built here with g++ (plain and under ThreadSanitizer)."""
import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = [str(ROOT / "tests" / "native" / "host_pool_stress.cpp"), str(ROOT / "lambda_amd" / "csrc" / "lx_host_pool.cpp")]


def _build(tmp_path, *flags):
    exe = tmp_path / ("hps" + "_".join(f.strip("-=") for f in flags))
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", *flags, *SRC, "-o", str(exe)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip(f"g++ {' '.join(flags)} failed: {r.stderr[-400:]}")
    return exe


def _run(exe, *args, env=None):
    e = dict(os.environ)
    e.pop("LOCAL_WORLD_SIZE", None)
    e.update(env or {})
    r = subprocess.run([str(exe), *map(str, args)], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stdout + r.stderr
    words = r.stdout.split()
    return {words[i]: words[i + 1] for i in range(0, len(words) - 1) if words[i] in ("width", "granted", "local_world", "failures")}, r


def test_every_part_runs_once_one_and_eight_callers(tmp_path):
    exe = _build(tmp_path)
    for _ in range(3):
        out, _r = _run(exe, 1500, 8)
        assert out["failures"] == "0"


def test_clean_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "-fsanitize=thread")
    out, r = _run(exe, 400, 8)
    assert out["failures"] == "0"
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-2000:]


def test_width_follows_granted_cpus_and_local_world_size(tmp_path):
    exe = _build(tmp_path)
    granted = int(_run(exe, 10, 2)[0]["granted"])
    assert 1 <= granted <= len(os.sched_getaffinity(0))
    out, _r = _run(exe, 50, 8, env={"LOCAL_WORLD_SIZE": "8"})
    assert int(out["local_world"]) == 8 and int(out["width"]) == max(1, min(granted // 8, 16))
    if shutil.which("taskset"):
        r = subprocess.run(["taskset", "-c", str(min(os.sched_getaffinity(0))), str(exe), "50", "4"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "width 1 granted 1" in r.stdout, r.stdout + r.stderr


def test_library_reports_its_host_threads():
    """lx_host_threads_info: no handle, no device."""
    from lambda_amd import capi

    lib = capi.load()
    w, g, lw = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert lib.lx_host_threads_info(C.byref(w), C.byref(g), C.byref(lw)) == 0
    assert 1 <= w.value <= 16 and w.value <= max(1, g.value) and lw.value >= 1

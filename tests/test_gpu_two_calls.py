"""A multi-query chunk in two calls (lx_host.cpp: enqueue_mq_first / enqueue_mq_second): the plan's pool is swept while the
streamed part of the plan is still being made, the second call sweeps the rest and runs one selection and one backtrace over all
slots.  Lists beyond 200 000 windows take that path; here a small list is sent there by the library's environment switch
LX_MQ_MERGE_BELOW (read once per process: every case runs in a process of its own) and must give, byte for byte, what the same
list gives on the path small lists take by themselves (pool and stream planned up front, one launch), and the oracle's scores."""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests import oracle_lib
from tests.test_gpu_score import SCHEMES
from tests.two_calls_case import case

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_case(tmp_path, name, env_extra, trace_bytes=None, which="ragged", two_calls=True):
    env = dict(os.environ, LX_HOST_TIMING="1", **env_extra)
    env.pop("LX_MQ_MERGE_BELOW", None)
    if two_calls:
        env["LX_MQ_MERGE_BELOW"] = "1"
    out = tmp_path / (name + ".npz")
    argv = [sys.executable, str(ROOT / "tests" / "two_calls_case.py"), str(out), which] + ([str(trace_bytes)] if trace_bytes else [])
    r = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out), r.stderr


def same(a, b):
    for k in ("score", "index", "hsp", "codes", "score_rows", "hsp_rows"):
        assert a[k].shape == b[k].shape and (a[k] == b[k]).all(), k


def test_two_calls_give_what_one_launch_of_the_plan_gives(tmp_path, oracle):
    two, log = run_case(tmp_path, "two", {})
    m = re.findall(r"one chunk in two calls: pool (\d+) wavefronts, then (\d+) of (\d+) .*dwords (\d+) \+ (\d+) \+ (\d+) of (\d+)", log)
    assert m, log[-2000:]  # (the path was taken)
    pool_wf, took, rest, dw0, ovf, dw1, total = (int(x) for x in m[0])
    assert pool_wf > 0 and took == rest and rest > 0
    assert "sweep_mq_kernel" in str(two["kernel"])
    one, log1 = run_case(tmp_path, "one", {}, two_calls=False)
    assert "one chunk in two calls" not in log1
    same(two, one)
    q, s, ext, mins = case()
    want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(SCHEMES["blosum62"]), threads=8)
    assert (two["score"] == want).all()
    live = (ext["q_len"] > 0) & (ext["s_len"] > 0)
    assert (two["index"] == np.nonzero(live & (want >= mins))[0]).all()
    # a budget that holds the pool's slots and the overflow area, but only part of what follows: the second call takes the
    # wavefronts that fit, the others go through chunks of their own
    part, log2 = run_case(tmp_path, "part", {}, trace_bytes=4 * (dw0 + ovf + dw1 // 3))
    m2 = re.findall(r"one chunk in two calls: pool (\d+) wavefronts, then (\d+) of (\d+) ", log2)
    assert m2 and 0 < int(m2[0][1]) < int(m2[0][2]), log2[-2000:]
    same(part, one)


def test_two_calls_whose_overflow_area_runs_out_are_run_again_with_wide_slots(tmp_path, oracle):
    """Every window beyond the compact codes on a handle that has not learned so: the chunk's overflow area (an eighth of its slots)
    runs out in the two calls' sweeps, the whole chunk is run again with int16 pairs from the sweep -- same results as one launch of the
    whole plan, and the oracle's scores."""
    two, log = run_case(tmp_path, "two_s", {}, which="strong")
    assert "one chunk in two calls" in log, log[-2000:]
    assert "(wide)" in log, log[-2000:]  # (the chunk that was run again)
    one, _ = run_case(tmp_path, "one_s", {}, which="strong", two_calls=False)
    same(two, one)
    q, s, ext, mins = case("strong")
    want = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(SCHEMES["blosum62"]), threads=8)
    assert (two["score"] == want).all() and (want > 2046).mean() > 0.9

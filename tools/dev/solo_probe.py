"""Development aid: what a wavefront of the solo packing (16 windows, 16 byte profiles: nucleotide lists) pays outside its steps --
the sweep alone over 150-column queries with one window each, at several window lengths: time per wavefront = a + b * steps."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import torch
from lambda_amd import capi, synth

dev = torch.device("cuda:0")
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(0, match=2, mismatch=-3, gap_open=-7, gap_extend=-2), 0)
rng = np.random.default_rng(1)
res = []
for ls in (32, 64, 128, 178, 256, 512):
    n, lq = 1_000_000 if ls <= 256 else 400_000, 150
    q = rng.integers(0, 4, n * lq, dtype=np.uint8)
    s = rng.integers(0, 4, n * ls, dtype=np.uint8)
    ext = np.zeros(n, dtype=capi.EXT_DTYPE)
    ext["q_off"] = np.arange(n, dtype=np.uint64) * lq
    ext["q_len"] = lq
    ext["s_off"] = np.arange(n, dtype=np.uint64) * ls
    ext["s_len"] = ls
    d_q = torch.from_numpy(np.concatenate([q, np.zeros(256, np.uint8)])).to(dev)
    d_s = torch.from_numpy(np.concatenate([s, np.zeros(256, np.uint8)])).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_hsp = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
    stride = (lq + ls + 3) & ~3
    d_ops = torch.zeros(16, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n, dtype=torch.int64, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    h.set_option(capi.LX_OPT_MAX_QLEN, lq)
    h.set_option(capi.LX_OPT_MAX_SLEN, ls)
    h.set_option(capi.LX_OPT_QUERY_RUN, 1)
    h.set_option(capi.LX_OPT_PASS2_MODE, 2)
    h.set_option(capi.LX_OPT_MQ_SWEEP, 2)
    best = 1e9
    for _ in range(4):
        h.extend_batch_dev(d_q, d_s, d_ext, n, 100000, d_score, d_hsp, d_ops, d_off, d_count)  # cut-off nothing passes: the sweep alone
        h.synchronize()
        best = min(best, h.last_phase_ms(0)[0])
    wf = n / 16
    per_wf_us = best * 1e3 / (wf / 2048)  # time of one wavefront slot's share, per wavefront
    res.append((ls + 7, per_wf_us))
    print(f"150 x {ls:4d}: {n} windows {best:8.3f} ms  {n * lq * ls / best / 1e9:6.2f} TCUPS; {per_wf_us:7.2f} us per wavefront at 2048 slots ({ls + 7} steps)  [{h.last_trace_kernel_name()[:48]}]")
x = np.array([r[0] for r in res], float); y = np.array([r[1] for r in res], float)
b, a = np.polyfit(x, y, 1)
print(f"fit: {a:.2f} us + {b:.4f} us per step  -> at 185 steps the fixed part is {100 * a / (a + b * 185):.1f} % of a wavefront")

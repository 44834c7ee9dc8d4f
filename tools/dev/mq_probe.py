"""Development aid: what the multi-query sweep pays for raggedness.  Uniform 150 x 176 windows through lx_extend_batch_dev with
LX_OPT_MQ_SWEEP=2, then the same list with one window per wavefront cut short (everything beyond its rows runs the checked
fetch path), and with queries of 300 / 450 columns (two / three panels)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import torch
from lambda_amd import capi, synth

dev = torch.device("cuda:0")
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)


def run(label, nq, lq, wpq, short_every=0, run_=4, promise_q=0):
    d_q, d_s, d_ext, ext = synth.make_batch_torch(nq, lq, wpq, 1234, dev)
    if short_every:
        ext = ext.copy()
        ext["s_len"][::short_every] = 24
        d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
    d_q = torch.cat([d_q, torch.zeros(256, dtype=torch.uint8, device=dev)])
    d_s = torch.cat([d_s, torch.zeros(256, dtype=torch.uint8, device=dev)])
    n = len(ext)
    ls = int(ext["s_len"].max())
    cells_exec = n * float(lq) * ls
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    d_hsp = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
    stride = (lq + ls + 3) & ~3
    d_ops = torch.zeros(n * stride + 16, dtype=torch.uint8, device=dev)
    d_off = torch.arange(n, dtype=torch.int64, device=dev) * stride
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    h.set_option(capi.LX_OPT_MAX_QLEN, promise_q or lq)
    h.set_option(capi.LX_OPT_MAX_SLEN, ls)
    h.set_option(capi.LX_OPT_QUERY_RUN, run_)
    h.set_option(capi.LX_OPT_PASS2_MODE, 2)
    h.set_option(capi.LX_OPT_MQ_SWEEP, 2)
    best = 1e9
    for _ in range(4):
        h.extend_batch_dev(d_q, d_s, d_ext, n, 100000, d_score, d_hsp, d_ops, d_off, d_count)  # cut-off nothing passes: the sweep alone
        h.synchronize()
        best = min(best, h.last_phase_ms(0)[0])
    print(f"{label:44s} {n:8d} windows {best:8.3f} ms  {cells_exec / best / 1e9:6.2f} TCUPS of full-length cells  [{h.last_trace_kernel_name()[:60]}]")


run("150 x 176 uniform, 4 queries / wavefront", 40000, 150, 16)
run("150 x 176 uniform, 1 query / wavefront", 40000, 150, 16, run_=16)
run("... one window in 16 cut to 24 rows", 40000, 150, 16, short_every=16)
run("150 x 176 through the multi-panel kernel", 40000, 150, 16, promise_q=300)
run("300 x 336 (two panels)", 12000, 300, 16)
run("300 x 336 (two panels), 1 query / wavefront", 12000, 300, 16, run_=16)
run("450 x 494 (three panels)", 6000, 450, 16)
run("100 x 122 ((8,13))", 80000, 100, 16)
run("200 x 230 ((8,13) x 2)", 24000, 200, 16)

"""Quick single-GPU throughput probe of the score kernel (development aid; bench.py is the contract)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import torch
from lambda_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lq = int(sys.argv[2]) if len(sys.argv) > 2 else 150
wpq = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
h = capi.Handle(0)
method = 62 if lq != 151 else 0
h.set_scoring(capi.builtin_scoring(62), 0)
d_q, d_s, d_ext, ext = synth.make_batch_torch(nq, lq, wpq, 1234, dev)
d_q = torch.cat([d_q, torch.zeros(256, dtype=torch.uint8, device=dev)])
d_s = torch.cat([d_s, torch.zeros(256, dtype=torch.uint8, device=dev)])
n = len(ext)
cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
d_out = torch.zeros(n, dtype=torch.int32, device=dev)
for max_qlen, run in ((0, 0), (lq, 0), (lq, wpq)):
    h.set_option(capi.LX_OPT_MAX_QLEN, max_qlen)
    h.set_option(capi.LX_OPT_QUERY_RUN, run)
    for _ in range(2):
        h.score_batch_dev(d_q, d_s, d_ext, n, d_out)
    h.synchronize()
    ms = []
    for _ in range(5):
        h.score_batch_dev(d_q, d_s, d_ext, n, d_out)
        h.synchronize()
        ms.append(h.last_kernel_ms())
    best = min(ms)
    print(f"max_qlen={max_qlen} run={run}: {n} ext, {cells/1e9:.2f} Gcells, {best:.3f} ms -> {cells/best/1e6:.1f} GCUPS  (all: {['%.2f'%m for m in ms]})")
print("mean score", d_out.float().mean().item())

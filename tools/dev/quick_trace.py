"""Quick single-GPU throughput probe of pass 2 (development aid)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import torch
from lambda_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
lq, wpq = 150, 32
dev = torch.device("cuda:0")
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62), 0)
d_q, d_s, d_ext, ext = synth.make_batch_torch(nq, lq, wpq, 1234, dev)
pad = torch.zeros(256, dtype=torch.uint8, device=dev)
d_q = torch.cat([d_q, pad]); d_s = torch.cat([d_s, pad])
n = len(ext)
d_out = torch.zeros(n, dtype=torch.int32, device=dev)
h.set_option(capi.LX_OPT_MAX_QLEN, lq); h.set_option(capi.LX_OPT_QUERY_RUN, wpq); h.set_option(capi.LX_OPT_MAX_SLEN, 176)
h.score_batch_dev(d_q, d_s, d_ext, n, d_out); h.synchronize()
keep = torch.nonzero(d_out >= 60).flatten()
ns = int(keep.numel())
ext_s = ext[keep.cpu().numpy()]
d_ext_s = torch.from_numpy(ext_s.view(np.uint8).copy()).to(dev)
sizes = ext_s["q_len"].astype(np.uint64) + ext_s["s_len"].astype(np.uint64)
off = np.zeros(ns, dtype=np.uint64); off[1:] = np.cumsum(sizes)[:-1]
d_off = torch.from_numpy(off.view(np.int64)).to(dev)
d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
d_hsp = torch.zeros(ns * 48, dtype=torch.uint8, device=dev)
cells = float((ext_s["q_len"].astype(np.float64) * ext_s["s_len"]).sum())
torch.cuda.synchronize()
for _ in range(2):
    h.align_batch_dev(d_q, d_s, d_ext_s, ns, d_hsp, d_ops, d_off)
h.synchronize()
ms = []
for _ in range(3):
    h.align_batch_dev(d_q, d_s, d_ext_s, ns, d_hsp, d_ops, d_off); h.synchronize(); ms.append(h.last_kernel_ms())
print(f"survivors {ns}/{n}; pass2 {cells/1e9:.2f} Gcells in {min(ms):.3f} ms -> {cells/min(ms)/1e6:.1f} GCUPS ({ms})")
hsp = np.frombuffer(d_hsp.cpu().numpy().tobytes(), dtype=capi.HSP_DTYPE)
print("mean score", hsp["score"].mean(), "mean ops", hsp["n_ops"].mean())

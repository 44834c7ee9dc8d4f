"""Pass-1 rate of the (16,10) packed-half geometry on the headline batch (QUERY_RUN = 8 selects it).  Development aid."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from lambda_amd import capi, synth
dev = torch.device("cuda:0")
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
d_q, d_s, d_ext, ext = synth.make_batch_torch(100000, 150, 32, 0x1A3BDA02, dev)
pad = torch.zeros(256, dtype=torch.uint8, device=dev)
d_q, d_s = torch.cat([d_q, pad]), torch.cat([d_s, pad])
n = len(ext); cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
d_score = torch.zeros(n, dtype=torch.int32, device=dev)
h.set_option(capi.LX_OPT_MAX_QLEN, 150)
for run in (32, 8):
    h.set_option(capi.LX_OPT_QUERY_RUN, run)
    for _ in range(3):
        h.score_batch_dev(d_q, d_s, d_ext, n, d_score); h.synchronize()
    print(run, h.last_kernel_name(), f"{h.last_kernel_ms():.3f} ms = {cells/h.last_kernel_ms()/1e6:.0f} GCUPS")

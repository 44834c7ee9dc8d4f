"""Throughput probe at the shape of BASELINE.json configs[2] (searchn: 150 bp reads, 8 windows per read; a slice of
the 1 M reads) -- pass 1 alone and the fused step.  Development aid, not a bench line."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from lambda_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
lq, wpq = 150, 8
dev = torch.device("cuda:0")
h = capi.Handle(0)
h.set_scoring(capi.builtin_scoring(0, match=2, mismatch=-3, gap_open=-5, gap_extend=-2), 0)
alpha = np.array([0, 1, 2, 4], dtype=np.uint8)
d_q, d_s, d_ext, ext = synth.make_batch_torch(nq, lq, wpq, 0x1A3BDA03, dev, alphabet=alpha) if "alphabet" in synth.make_batch_torch.__code__.co_varnames else (None,) * 4
if d_q is None:
    q, s, ext = synth.make_batch_np(nq, lq, wpq, seed=0x1A3BDA03, alphabet=alpha, sub_rate=0.05, indel_rate=0.01)
    d_q, d_s = torch.from_numpy(q).to(dev), torch.from_numpy(s).to(dev)
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(dev)
pad = torch.zeros(256, dtype=torch.uint8, device=dev)
d_q, d_s = torch.cat([d_q, pad]), torch.cat([d_s, pad])
n = len(ext)
cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
d_score = torch.zeros(n, dtype=torch.int32, device=dev)
sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
off = np.zeros(n, dtype=np.uint64); off[1:] = np.cumsum(sizes)[:-1]
d_off = torch.from_numpy(off.view(np.int64)).to(dev)
d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
d_hsp = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
d_count = torch.zeros(2, dtype=torch.int64, device=dev)
h.set_option(capi.LX_OPT_MAX_QLEN, lq); h.set_option(capi.LX_OPT_QUERY_RUN, wpq); h.set_option(capi.LX_OPT_MAX_SLEN, int(ext["s_len"].max()))
for packed in (1, 0):
    h.set_option(capi.LX_OPT_PACKED_HALF, packed)
    for _ in range(2):
        h.score_batch_dev(d_q, d_s, d_ext, n, d_score); h.synchronize()
    ms = h.last_kernel_ms()
    print(f"pass1 packed={packed}: {h.last_kernel_name()}  {cells/1e9:.1f} Gcells in {ms:.3f} ms = {cells/ms/1e6:.0f} GCUPS")
h.set_option(capi.LX_OPT_PACKED_HALF, 1)
stream = torch.cuda.Stream(device=dev)
for _ in range(3):
    h.extend_batch_dev(d_q, d_s, d_ext, n, 60, d_score, d_hsp, d_ops, d_off, d_count, stream=stream.cuda_stream); h.synchronize()
print("fused phases ms:", [round(h.last_phase_ms(p)[0], 3) for p in (0, 1, 2, 3)], "survivors", int(d_count.cpu()[1]), "of", n, h.last_trace_kernel_name())

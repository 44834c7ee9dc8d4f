"""Does a second stream help?  Two handles process alternating batches on two streams (batch k+1's sweep can overlap
batch k's backtrace).  Development aid; prints sequential vs two-stream throughput."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from lambda_amd import capi, synth

dev = torch.device("cuda:0")
nq, lq, wpq = (int(sys.argv[1]) if len(sys.argv) > 1 else 100000), 150, 32
sets = []
for k in range(2):
    h = capi.Handle(0)
    h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
    h.set_option(capi.LX_OPT_MAX_QLEN, lq); h.set_option(capi.LX_OPT_QUERY_RUN, wpq)
    d_q, d_s, d_ext, ext = synth.make_batch_torch(nq, lq, wpq, 0x1A3BDA02 + k, dev)
    pad = torch.zeros(256, dtype=torch.uint8, device=dev)
    d_q, d_s = torch.cat([d_q, pad]), torch.cat([d_s, pad])
    n = len(ext)
    h.set_option(capi.LX_OPT_MAX_SLEN, int(ext["s_len"].max()))
    sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64); off[1:] = np.cumsum(sizes)[:-1]
    bufs = dict(h=h, q=d_q, s=d_s, e=d_ext, n=n, off=torch.from_numpy(off.view(np.int64)).to(dev),
                ops=torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev), hsp=torch.zeros(n * 48, dtype=torch.uint8, device=dev),
                score=torch.zeros(n, dtype=torch.int32, device=dev), cnt=torch.zeros(2, dtype=torch.int64, device=dev),
                stream=torch.cuda.Stream(device=dev))
    sets.append(bufs)
cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())

def run(b):
    b["h"].extend_batch_dev(b["q"], b["s"], b["e"], b["n"], 91, b["score"], b["hsp"], b["ops"], b["off"], b["cnt"], stream=b["stream"].cuda_stream)

for b in sets:
    run(b); run(b)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for k in range(K):
    run(sets[0])
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
t0 = time.perf_counter()
for k in range(K):
    run(sets[k & 1])
torch.cuda.synchronize()
t2 = time.perf_counter() - t0
print(f"one stream:  {t1/K*1e3:.2f} ms/step = {cells*K/t1/1e9:.0f} GCUPS")
print(f"two streams: {t2/K*1e3:.2f} ms/step = {cells*K/t2/1e9:.0f} GCUPS")

"""world_size-2 CPU test (gloo) of the multi-GPU path: query sharding + final hit gather.
The per-rank "extension" is done by the CPU oracle here -- this test is about the sharding arithmetic and the
collective, the kernels are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lambda_amd import capi, shard, synth


def test_shard_range_partitions():
    for n in (0, 1, 7, 100, 100_000, 1_000_003):
        for w in (1, 2, 3, 4, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, wpq, out_dir):
    from tests import oracle_lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q, s, ext = synth.make_batch_np(nq, 60, wpq, seed=99)  # every rank generates the same global batch ...
        lo, hi = shard.shard_range(nq, rank, world)            # ... and extends only its own queries
        mine = ext[lo * wpq: hi * wpq]
        orc = oracle_lib.load()
        sc = oracle_lib.scoring_from(capi.builtin_scoring(62))
        scores = orc.score_batch(q, s, mine, sc)
        keep = scores >= 40
        rec = np.stack([np.arange(lo * wpq, hi * wpq)[keep], scores[keep]], axis=1).astype(np.int64)
        root = shard.gather_hits(torch.from_numpy(rec))              # to rank 0, the writer
        everyone = shard.gather_hits(torch.from_numpy(rec), dst=None)  # all_gather variant
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), everyone.numpy())
        np.save(os.path.join(out_dir, f"root{rank}.npy"), root.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_equals_unsharded(tmp_path, oracle):
    from tests import oracle_lib

    nq, wpq, world = 37, 5, 2  # odd query count: unequal shards, different record counts per rank
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, nq, wpq, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    q, s, ext = synth.make_batch_np(nq, 60, wpq, seed=99)
    scores = oracle.score_batch(q, s, ext, oracle_lib.scoring_from(capi.builtin_scoring(62)))
    keep = scores >= 40
    want = np.stack([np.arange(len(ext))[keep], scores[keep]], axis=1).astype(np.int64)
    assert len(want) > 10
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == want.shape and (got == want).all()
        root = np.load(tmp_path / f"root{r}.npy")
        assert (root.shape == want.shape and (root == want).all()) if r == 0 else root.shape == (0, 2)

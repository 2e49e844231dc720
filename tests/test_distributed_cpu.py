"""world_size-2 `gloo` test (CPU) of the multi-GPU path: block-contiguous sharding of independent problems and the
post-round gather of per-problem status / counters.  The data path itself has no collective."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import ROOT, load_pkg


def test_shard_range_covers_everything():
    load_pkg()
    from calipso_jl_amd.batch import shard_range
    for n, w in ((256, 8), (10, 3), (5, 8), (0, 2), (7, 1)):
        ids = [list(shard_range(n, r, w)) for r in range(w)]
        flat = [i for part in ids for i in part]
        assert flat == list(range(n))                     # contiguous, ordered, no overlap
        assert max(len(p) for p in ids) - min(len(p) for p in ids) <= 1
    assert list(shard_range(256, 3, 8)) == list(range(96, 128))   # C4: 32 instances per GPU


def _worker(rank, world, port, out_dir):
    import sys
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from helpers import load_pkg as lp
    lp()
    from calipso_jl_amd.batch import gather_results, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = list(shard_range(7, rank, world))               # uneven shard: 4 + 3
    status = np.array([[1, 10 + i, 6, 3 * i] for i in ids], dtype=np.int32)
    counters = np.array([len(ids) * 10.0, float(rank + 1)])
    all_status, total = gather_results(status, counters)
    np.save(os.path.join(out_dir, "status_%d.npy" % rank), all_status)
    np.save(os.path.join(out_dir, "total_%d.npy" % rank), total)
    dist.destroy_process_group()


def test_gather_results_gloo_world2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    expect = np.array([[1, 10 + i, 6, 3 * i] for i in range(7)], dtype=np.int32)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("status_%d.npy" % r)), expect)      # global problem-id order on every rank
        assert np.array_equal(np.load(tmp_path / ("total_%d.npy" % r)), np.array([70.0, 3.0]))


def test_gather_results_single_process_identity():
    load_pkg()
    from calipso_jl_amd.batch import gather_results
    st = np.array([[1, 5, 2, 9]], dtype=np.int32)
    a, c = gather_results(st, np.array([3.0]))
    assert np.array_equal(a, st) and c[0] == 3.0


def test_batch_lanes_keep_priority_classes_apart():
    """BatchSolver drives instance k from host lane k mod 3, so concurrently running instances never share a stream-priority
    class (= creation index mod 3); results come back in instance order"""
    import threading
    load_pkg()
    from calipso_jl_amd.batch import BatchSolver

    class Fake:
        def __init__(self, k):
            self.k = k
            self.thread = None

        def newton_step(self, advance=False):
            self.thread = threading.get_ident()
            return dict(status=0, k=self.k)

    fakes = [Fake(k) for k in range(8)]
    b = BatchSolver(fakes)
    out = b.newton_step()
    assert [o["k"] for o in out] == list(range(8)) and b.lanes == 3
    for lane in range(3):
        assert len({f.thread for f in fakes[lane::3]}) == 1          # one host thread per lane
    b.close()

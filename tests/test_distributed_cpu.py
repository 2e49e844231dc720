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


def test_batch_spreads_the_streams_of_its_lanes():
    """BatchSolver.spread_streams (no GPU: stand-ins for the handles): every pair of units of DIFFERENT lanes is probed, the later unit of a colliding pair is rebound
    (cycling through the priority classes) until no pair collides or the budget is spent; units of one lane are never probed against each other; handles without
    the probe (the Fake instances above) are left alone"""
    load_pkg()
    from calipso_jl_amd.batch import BatchSolver

    class Leader:
        probes = 0

        def __init__(self, k, queue):
            self.k, self.queue, self.rebound = k, queue, []

        def streams_concurrent(self, other):
            Leader.probes += 1
            return (self.queue != other.queue, 10.0, 10.0, 170.0, 420.0, 425.0)

        def rebind_stream(self, priority_class=-1):
            self.rebound.append(priority_class)
            self.queue = 100 + 10 * self.k + len(self.rebound)          # a queue nobody else has

        def newton_step(self, advance=False):
            return dict(status=0)

    class Unit:                                                          # a group: the first member carries the launches
        def __init__(self, leader):
            self.solvers = [leader, object()]

        def newton_step(self, advance=False):
            return [dict(status=0)]

    # four units in two lanes: units 0, 2 (lane 0) and 1, 3 (lane 1); 0 and 1 share a queue, 2 and 3 share another; 0 and 2 share one too (same lane: never probed)
    ls = [Leader(0, 7), Leader(1, 7), Leader(2, 7), Leader(3, 8)]
    b = BatchSolver([Unit(l) for l in ls], lanes=2)
    rep = b.stream_report
    assert rep["pairs"] == 4 and rep["collisions"] == 2 and rep["left"] == 0 and rep["rebinds"] == 1, rep      # (0, 1) and (1, 2) collide: unit 1 moves once
    assert ls[1].rebound == [(0 + 1) % 3] and not ls[0].rebound and not ls[2].rebound and not ls[3].rebound
    assert ls[0].queue == ls[2].queue == 7                               # the two units of lane 0 still share a queue: they never run at the same time
    b.close()
    # a pair that cannot be separated: the budget bounds the work, the report says what is left
    class Stuck(Leader):
        def rebind_stream(self, priority_class=-1):
            self.rebound.append(priority_class)
    st = [Stuck(0, 1), Stuck(1, 1)]
    b = BatchSolver(st, lanes=2)
    assert b.stream_report == dict(pairs=1, collisions=1, rebinds=12, left=1) and st[1].rebound == [(r + 1) % 3 for r in range(12)] and not st[0].rebound
    b.close()
    b = BatchSolver([Leader(0, 1)], lanes=3)                             # one unit: one lane, nothing to probe
    assert b.stream_report is None
    b.close()


def test_bench_spawns_its_ranks():
    """`python bench.py --gpus N` (the driver's plain command form, no torch.distributed.run around it) must start N ranks itself.
    --spawn-check stops every rank before it touches a GPU, so the launch path is covered here: the ranks report in through a gloo
    all-gather and rank 0 prints one JSON line."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["spawn_check"] is True and d["n_gpus"] == 2 and d["gpus_argument"] == 2 and d["launched_by"] == "torch.distributed.run"
    assert sorted(tuple(r) for r in d["ranks"]) == [(0, 0), (1, 1)]           # (RANK, LOCAL_RANK) of every process
    # one GPU: nothing is spawned
    out1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn-check"], capture_output=True, text=True, timeout=600, env=env)
    d1 = json.loads([l for l in out1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["launched_by"] == "direct"


def test_c4_sharding_of_256_instances():
    """BASELINE config 4: 256 problems over 8 GPUs = 32 block-contiguous ids per rank, in two groups of 16; gathered rows come back in
    global problem-id order with every id exactly once"""
    load_pkg()
    from calipso_jl_amd.batch import shard_range
    seen = []
    for rank in range(8):
        ids = list(shard_range(8 * 32, rank, 8))
        assert ids == list(range(32 * rank, 32 * rank + 32))
        groups = [ids[k:k + 16] for k in range(0, 32, 16)]
        assert [len(g) for g in groups] == [16, 16] and groups[0][0] == 32 * rank and groups[1][0] == 32 * rank + 16
        seen += ids
    assert seen == list(range(256))


@pytest.mark.parametrize("failing", ["", "2,5"])
def test_bench_eight_ranks_shard_config4_and_gather_in_problem_order(failing):
    """`python bench.py --gpus 8 --spawn-check`: eight gloo ranks (no GPU), BASELINE config 4's sharding (256 problems: 32 per rank in two groups of 16) and the
    post-round exchange on synthetic status rows — every rank's rows arrive on rank 0 in global problem-id order.  With a product communicator that cannot be
    created on SOME ranks (simulated) every rank must agree to fall back to torch.distributed, and none may keep a half-made communicator."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CALIPSO_BENCH_FAKE_COMM_FAIL_RANKS"] = failing
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--spawn-check"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and sorted(tuple(r) for r in d["ranks"]) == [(r, r) for r in range(8)]
    assert d["c4_shard"] == [0, 31, 32]                                   # rank 0: ids 0..31
    assert d["gathered_ids"] == list(range(256))
    assert d["gathered_ranks"] == [pid // 32 for pid in range(256)]
    assert d["gathered_groups"] == [(pid % 32) // 16 for pid in range(256)]
    assert d["counter_total"] == 256.0 and d["comm_left_open"] is False
    if failing:
        assert "could not be created on every rank" in d["exchange_path"]
    else:
        assert d["exchange_path"].startswith("torch.distributed (gloo)")

"""Batched / multi-GPU driver: many independent problem instances in flight (BASELINE config C4).

The reference has no batching (distinct `Solver`s are simply independent); the only cross-problem loop it has is
the per-parameter loop of differentiate! (src/solver/differentiate.jl:29-58).  Here
  * problems are sharded block-contiguously over the ranks of one node: problem ids [rank*B/W, (rank+1)*B/W);
  * inside a rank every instance owns a HIP stream (its handle), and instances are driven concurrently from host threads
    (the C ABI calls release the GIL), so the latency-bound phases of one instance overlap the matrix-core phases of another;
  * there is NO collective on the data path.  After a batch round the per-problem status / iteration counts are
    all-gathered and the step counters all-reduced (RCCL over xGMI when the backend is nccl; gloo on CPU for the tests).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def shard_range(n_problems, rank, world):
    """block-contiguous problem ids of `rank` (uneven remainders go to the low ranks)"""
    base, rem = divmod(int(n_problems), int(world))
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


class BatchSolver:
    """A set of independent Solver handles on one GPU, stepped / solved concurrently.

    At most `lanes` instances are in flight at a time (default 3 = the number of HIP stream-priority classes, which the runtime
    maps to distinct hardware queues; handles are created with priority class = creation index mod 3).  Instance k is always
    driven by host thread k mod lanes, so instances that run concurrently never share a priority class — streams of equal
    priority can be multiplexed onto one hardware queue, which serialises them and was measured to be slower than running
    them back to back.  Distinct classes are not enough (queues of different classes can still end up behind one dispatcher,
    by the accident of how many streams the process created before): at creation the leaders of the lanes are PROBED and
    rebound until they run side by side (spread_streams)."""

    def __init__(self, solvers, lanes=3, spread_streams=True):
        self.solvers = list(solvers)
        self.lanes = max(1, min(int(lanes), len(self.solvers))) if self.solvers else 1
        self.pool = ThreadPoolExecutor(max_workers=self.lanes)
        import os
        if os.environ.get("CALIPSO_BATCH_LANE_CLASSES") and self.lanes > 1:       # (experiment: the priority class of every unit's leader by its lane, e.g. "1,1,1")
            cls = [int(v) for v in os.environ["CALIPSO_BATCH_LANE_CLASSES"].split(",")]
            for k, u in enumerate(self.solvers):
                self._leader(u).rebind_stream(cls[(k % self.lanes) % len(cls)])
        self.stream_report = self.spread_streams() if spread_streams and self.lanes > 1 and os.environ.get("CALIPSO_BATCH_SPREAD_STREAMS", "1") != "0" else None

    @staticmethod
    def _leader(unit):
        """the handle whose stream carries a unit's launches: a Solver itself, the first member of a Group"""
        return unit.solvers[0] if hasattr(unit, "solvers") else unit

    def spread_streams(self, max_rebinds=12):
        """Units of different lanes run at the same time; whether their streams really run side by side depends on how the runtime mapped them to hardware
        queues, i.e. on how many streams the process created before (calipso_hip_streams_concurrent measures it: BASELINE config 4's dense batch ran at 1290
        or 1480 steps/s per GPU by that accident alone).  Probe every pair of leaders of different lanes; a unit whose stream collides with an earlier one gets
        a new stream (calipso_hip_rebind_stream, cycling through the priority classes) until every pair runs side by side or `max_rebinds` is spent.  Returns
        {"pairs": probed pairs, "collisions": found at first, "rebinds": streams replaced, "left": colliding pairs left}."""
        leaders = [self._leader(u) for u in self.solvers]
        if not all(hasattr(h, "streams_concurrent") for h in leaders):
            return None
        pairs = [(i, j) for j in range(len(leaders)) for i in range(j) if i % self.lanes != j % self.lanes]
        if len(pairs) > 64:                     # (many small units: the first units of every lane stand for the rest)
            pairs = [(i, j) for (i, j) in pairs if j < 2 * self.lanes]
        report = dict(pairs=len(pairs), collisions=0, rebinds=0, left=0)
        first = True
        for attempt in range(max_rebinds + 1):
            bad = [(i, j) for (i, j) in pairs if not leaders[i].streams_concurrent(leaders[j])[0]]
            if first:
                report["collisions"] = len(bad); first = False
            report["left"] = len(bad)
            if not bad or attempt == max_rebinds:
                break
            j = bad[0][1]                        # the later unit of the first colliding pair moves
            leaders[j].rebind_stream((report["rebinds"] + j) % 3)
            report["rebinds"] += 1
        return report

    def _run(self, fn):
        """fn(solver) for every instance; lane w handles instances w, w+lanes, ... in order; results in instance order"""
        out = [None] * len(self.solvers)

        def lane(w):
            for k in range(w, len(self.solvers), self.lanes):
                out[k] = fn(self.solvers[k])
        list(self.pool.map(lane, range(self.lanes)))
        return out

    def newton_step(self, advance=False):
        return self._run(lambda s: s.newton_step(advance=advance))

    def newton_steps(self, passes, advance=False):
        """`passes` Newton steps of every unit with NO synchronisation between the lanes from pass to pass: lane w runs all the passes of its
        units back to back.  Problems are independent, so nothing requires the lanes to finish a pass together; free-running lanes drift apart
        in phase, so a lane in its matrix-core-bound phase (Schur complement) can overlap lanes in their HBM-bound phases (solves, refinement).
        Measured on C4 this is within noise of lock-step passes.  Returns the infos of the LAST pass in unit order."""
        out = [None] * len(self.solvers)

        def lane(w):
            mine = list(range(w, len(self.solvers), self.lanes))
            for p in range(passes):
                for k in mine:
                    out[k] = self.solvers[k].newton_step(advance=advance)
        list(self.pool.map(lane, range(self.lanes)))
        return out

    def solve(self, solve_fn):
        """solve_fn(solver) -> bool for every instance; returns (status int32[k, 4]) rows = [converged, iterations, outer, factorizations]"""
        def one(s):
            ok = solve_fn(s)
            st = s.stats()
            return [int(ok), st["total_iterations"], st["outer"], st["factorizations"]]
        return np.array(self._run(one), dtype=np.int32).reshape(len(self.solvers), 4)

    def synchronize(self):
        for s in self.solvers:
            s.synchronize()

    def close(self):
        self.pool.shutdown(wait=True)


def gather_results(local_status, local_counters, device=None):
    """all-gather the per-problem status rows (int32[k, 4], k may differ per rank) and all-reduce the counters (sum).
    Returns (status of all problems in global problem-id order, summed counters).  Needs torch.distributed initialised;
    with a single process it is the identity."""
    import torch
    import torch.distributed as dist
    local_status = np.ascontiguousarray(local_status, dtype=np.int32).reshape(-1, 4)
    counters = np.ascontiguousarray(local_counters, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_status, counters
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    k = torch.tensor([local_status.shape[0]], dtype=torch.int64, device=dev)
    ks = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(ks, k)
    kmax = int(max(int(t.item()) for t in ks))
    pad = np.full((kmax, 4), -1, dtype=np.int32)
    pad[:local_status.shape[0]] = local_status
    mine = torch.from_numpy(pad).to(dev)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    rows = [p.cpu().numpy()[:int(n.item())] for p, n in zip(parts, ks)]
    c = torch.from_numpy(counters.copy()).to(dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return np.concatenate(rows, axis=0), c.cpu().numpy()

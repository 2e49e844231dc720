"""GPU (-m gpu): the C-level RCCL exchange of the batched path (calipso_hip_comm_*) with one rank on one GPU:
ncclGetUniqueId -> ncclCommInitRank -> all-gather of status rows / all-reduce of counters.  (Two RCCL ranks cannot share one device;
the world-2 logic of the gather — uneven shards, global problem-id order — is covered on CPU by tests/test_distributed_cpu.py with
gloo, and the N > 1 RCCL run is the driver's multi-GPU bench.)"""
import numpy as np
import pytest

from helpers import load_pkg

pytestmark = pytest.mark.gpu


def test_comm_single_rank_roundtrip():
    pkg = load_pkg()
    uid = pkg.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    c = pkg.Comm(0, 1, uid, device=0)
    assert c.size() == (1, 0)                         # ncclCommCount / ncclCommUserRank of the live communicator
    rows = np.array([[1, 10 + i, 6, 3 * i] for i in range(5)], dtype=np.int32)
    out, counts = c.gather_status(rows, 5)
    assert np.array_equal(out, rows) and counts.tolist() == [5]
    out0, counts0 = c.gather_status(np.zeros((0, 4), dtype=np.int32), 1)
    assert out0.shape == (0, 4) and counts0.tolist() == [0]
    v = c.allreduce_sum([3.0, 4.5])
    assert v.tolist() == [3.0, 4.5]
    with pytest.raises(pkg.CalipsoHipError):
        c.gather_status(rows, 2)                      # capacity too small
    c.close()

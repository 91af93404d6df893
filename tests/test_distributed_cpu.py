"""world_size-2 gloo test of the multi-GPU sharding + the single gather of result records.

On the GPU box each rank drives its own MI355X through libbpvo_hip and the gather runs over RCCL; here the per-rank
compute is the CPU oracle (this is a test: the thing under test is the sharding / gather host logic, which is the same
code bench.py runs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bpvo_amd import capi, synth
from bpvo_amd.distributed import RECORD_FLOATS, gather_records, records_to_poses, shard_range

ROWS, COLS, LEVELS, N_TOTAL = 96, 128, 2, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(b):
    p = b.default_params()
    p.numPyramidLevels = LEVELS
    p.descriptor = capi.DESC_BITPLANES
    p.verbosity = capi.VERB_SILENT
    return p


def _run_shard(lo, hi):
    import __graft_entry__ as ge
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    batch = synth.make_batch(ROWS, COLS, hi - lo, first_index=lo)
    ctx = orc.create(batch["K"], batch["b"], ROWS, COLS, _params(orc), n_frames=2 * (hi - lo), n_pairs=hi - lo)
    poses, stats = ctx.batch_run(batch["images"], batch["disparities"])
    rec = np.zeros((hi - lo, RECORD_FLOATS), np.float32)
    rec[:, :12] = poses[:, :3, :].reshape(-1, 12)
    rec[:, 12:12 + LEVELS] = stats["numIterations"]
    rec[:, 20:20 + LEVELS] = stats["status"]
    return rec


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(N_TOTAL, rank, world)
    rec = torch.from_numpy(_run_shard(lo, hi))
    allrec = gather_records(rec, dst=0)
    if rank == 0:
        np.save(out_path, allrec.numpy())
    else:
        assert allrec is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n, w in [(1024, 8), (1024, 1), (10, 4), (3, 8)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard_range(1024, 3, 8) == (384, 512)


def test_two_rank_gather_equals_single_process(tmp_path, orc):
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    gathered = np.load(out)
    single = _run_shard(0, N_TOTAL)
    assert gathered.shape == (N_TOTAL, RECORD_FLOATS)
    assert np.array_equal(gathered, single)      # pairs are independent: sharding cannot change any result
    poses, iters, status = records_to_poses(gathered)
    assert poses.shape == (N_TOTAL, 4, 4) and np.allclose(poses[:, 3], [0, 0, 0, 1])
    assert (iters[:, :LEVELS] >= 0).all() and (status[:, :LEVELS] >= capi.STATUS_PARAMETER_TOL).all()


def test_gather_is_identity_without_process_group():
    t = torch.arange(64, dtype=torch.float32).reshape(2, 32)
    assert gather_records(t) is t


def test_bench_gpus_flag_is_read():
    """bench.py --gpus N is not a no-op: without a launcher it wants to start N ranks itself (and says so when the box has fewer
    devices); under a launcher every rank refuses a WORLD_SIZE that disagrees with it.  No GPU is touched by either path."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() < 64:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and "--gpus 64: only" in (r.stdout + r.stderr)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 1 but the launcher started 2" in (r.stdout + r.stderr)

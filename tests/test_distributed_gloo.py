"""world_size-2 test of the N>1 path on CPU (gloo): contiguous time shards, one all-gather of the
embeddings, drift over the gathered sequence == drift over the unsharded sequence."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import truely_amd  # noqa: F401
    from truely_amd.distributed import allgather_embeddings, allgather_embeddings_async, shard_bounds, shard_counts
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)
    n = 37                                           # ragged: 19 + 18
    emb = rng.standard_normal((n, 512)).astype(np.float32)
    valid = (rng.uniform(size=n) > 0.2).astype(np.uint8)
    lo, hi = shard_bounds(n, world, rank)
    e, v = allgather_embeddings(torch.from_numpy(emb[lo:hi]), torch.from_numpy(valid[lo:hi]))
    ok = np.array_equal(e.numpy(), emb) and np.array_equal(v.numpy(), valid)
    # known shard sizes: one collective, no size exchange; box / rect rows travel in the same payload
    box = rng.standard_normal((n, 4)).astype(np.float32)
    rect = rng.integers(-5, 4000, (n, 4)).astype(np.int32)
    e3, v3, b3, r3 = allgather_embeddings(torch.from_numpy(emb[lo:hi]), torch.from_numpy(valid[lo:hi]), counts=shard_counts(n, world),
                                          box=torch.from_numpy(box[lo:hi]), rect=torch.from_numpy(rect[lo:hi]))
    ok = ok and np.array_equal(e3.numpy(), emb) and np.array_equal(v3.numpy(), valid) and np.array_equal(b3.numpy(), box) \
        and np.array_equal(r3.numpy(), rect) and r3.dtype == torch.int32
    try:                                             # counts that do not match the local shard are rejected, not mis-gathered
        allgather_embeddings(torch.from_numpy(emb[lo:hi]), torch.from_numpy(valid[lo:hi]), counts=[1, 1])
        ok = False
    except ValueError:
        pass
    # two gathers in flight, consumed one step late (what bench.py's step loop does): same rows, in order
    h1 = allgather_embeddings_async(torch.from_numpy(emb[lo:hi]), torch.from_numpy(valid[lo:hi]), counts=shard_counts(n, world))
    h2 = allgather_embeddings_async(torch.from_numpy(emb[lo:hi] * 2), torch.from_numpy(valid[lo:hi]), counts=shard_counts(n, world))
    (ea, va), (eb, vb) = h1.wait(), h2.wait()
    ok = ok and np.array_equal(ea.numpy(), emb) and np.array_equal(eb.numpy(), emb * 2) and np.array_equal(vb.numpy(), valid)
    # empty shard on one rank
    e2, v2 = allgather_embeddings(torch.from_numpy(emb[:3] if rank == 0 else emb[:0]), torch.from_numpy(valid[:3] if rank == 0 else valid[:0]))
    ok = ok and e2.shape == (3, 512) and np.array_equal(v2.numpy(), valid[:3])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_embeddings_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_sharded_drift_equals_unsharded(oracle):
    """Time-ordered concatenation of shards is the only thing the drift pass needs."""
    from truely_amd.distributed import shard_bounds
    rng = np.random.default_rng(1)
    n = 90
    emb = rng.standard_normal((n, 512)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    valid = np.ones(n, np.uint8)
    full = oracle.drift_score(emb, valid, n * 4, 30)
    parts = [emb[slice(*shard_bounds(n, 8, r))] for r in range(8)]
    again = oracle.drift_score(np.concatenate(parts), valid, n * 4, 30)
    assert full["score"] == again["score"] and np.array_equal(full["sims"], again["sims"])


def test_bench_self_spawn_command(monkeypatch):
    """`bench.py --gpus N` with no launcher starts N ranks itself as a child torch.distributed.run (never an exec of a
    process that touched the GPU) and refuses RCCL when fewer GPUs are visible."""
    import bench
    calls = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "1"])
    args = bench.parse_args(sys.argv[1:])
    assert bench.spawn_ranks(args) == 0
    cmd, env = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "2", "--backend", "gloo", "--steps", "1"] and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    import pytest
    args = bench.parse_args(["--gpus", "64"])          # nccl on a box with fewer GPUs: loud failure, nothing spawned
    with pytest.raises(SystemExit):
        bench.spawn_ranks(args)
    assert len(calls) == 1

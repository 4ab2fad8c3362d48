"""The N > 1 launch path on real hardware, rehearsed on ONE GPU: `bench.py --gpus 2 --backend gloo` starts two fresh
rank processes itself (each with its own HIP context on the same card), shards the clip, all-gathers the embedding
rows and scores it.  The gathered result must equal a single-process run over the same frames.  Also BASELINE
configs[3] (`--mode streams`: one independent clip per rank, no collective)."""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest
import torch

import truely_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]   # stdout is ONE JSON line (rank 0's), nothing else
    return json.loads(lines[0])


def test_world2_sharded_equals_single_rank(engine):
    n = 6
    r = _bench("--gpus", "2", "--backend", "gloo", "--batch", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r["n_gpus"] == 2 and r["config"]["mode"] == "sharded" and r["config"]["frames_per_gpu"] == n
    # the same clip in one process: segment r of the clip is the seeded clip r (bench.make_clip)
    fr = np.concatenate([truely_amd.synthetic.synthetic_frames(n, 720, 1280, seed=s) for s in (0, 1)])
    out = engine.detect_embed(fr)
    d = engine.drift_score(out["emb"], out["valid"], 2 * n * 4, 30)
    assert r["config"]["score"] == d["score"]
    assert r["config"]["emb_crc32"] == zlib.crc32(out["emb"].cpu().numpy().tobytes())


def test_world2_streams_mode(engine):
    """configs[3]: every rank scores its own clip; no collective in the data path."""
    n = 4
    r = _bench("--gpus", "2", "--backend", "gloo", "--mode", "streams", "--batch", str(n), "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline", "--in-flight", "1")
    assert r["n_gpus"] == 2 and r["config"]["mode"] == "streams" and len(r["config"]["scores"]) == 2
    for s in (0, 1):
        fr = truely_amd.synthetic.synthetic_frames(n, 720, 1280, seed=s)
        out = engine.detect_embed(fr)
        assert r["config"]["scores"][s] == engine.drift_score(out["emb"], out["valid"], n * 4, 30)["score"]


def test_config3_concurrent_ingest_rehearsal(engine):
    """BASELINE configs[3] as worded -- "8-video concurrent INGEST, frame-sharded across GPUs" -- rehearsed with two rank processes on
    this one card: every rank takes ITS OWN clip as host NV12 (what a decoder hands over), uploads it through pinned memory on a
    copy stream, converts it on the device (trl_ingest_nv12) and scores it; no data-path collective.  Each rank's score must be
    the score a single process gets for that clip through the same NV12 round trip."""
    from truely_amd.ingest import bgr_to_nv12
    n = 4
    r = _bench("--gpus", "2", "--backend", "gloo", "--mode", "streams", "--ingest", "nv12", "--batch", str(n), "--steps", "3",
               "--warmup", "1", "--no-cpu-baseline")
    assert r["n_gpus"] == 2 and r["config"]["mode"] == "streams" and r["config"]["ingest"] == "nv12" and len(r["config"]["scores"]) == 2
    for s in (0, 1):
        fr = truely_amd.synthetic.synthetic_frames(n, 720, 1280, seed=s)
        bgr = engine.ingest_nv12(bgr_to_nv12(fr), 720, 1280, 1)            # BT.601 round trip: what the rank's detector saw
        out = engine.detect_embed(bgr)
        assert r["config"]["scores"][s] == engine.drift_score(out["emb"], out["valid"], n * 4, 30)["score"], s


def test_rccl_needs_as_many_gpus_as_ranks():
    """On this one-GPU box the RCCL launch must fail loudly instead of silently running one rank."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("multi-GPU box")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "needs 2 visible GPUs" in p.stderr


def test_rccl_branch_single_rank(engine):
    """The RCCL code path itself -- `init_process_group("nccl", device_id=...)`, the all-gather of the result rows, barrier, the
    all-reduce of the step time -- executed for real on this one-GPU box with a world of ONE rank (RCCL refuses two ranks on one
    device, so N > 1 over RCCL is the 8-GPU node's to run).  Same result as the plain single-GPU run."""
    n = 6
    r = _bench("--gpus", "1", "--force-dist", "--backend", "nccl", "--batch", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r["n_gpus"] == 1 and r["config"]["backend"] == "nccl"
    fr = truely_amd.synthetic.synthetic_frames(n, 720, 1280, seed=0)
    out = engine.detect_embed(fr)
    assert r["config"]["emb_crc32"] == zlib.crc32(out["emb"].cpu().numpy().tobytes())
    assert r["config"]["score"] == engine.drift_score(out["emb"], out["valid"], n * 4, 30)["score"]

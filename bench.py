"""bench.py -- BASELINE.json metric: frames/sec (detect + embed + drift), synthetic clips.

One "step" = one pass of the hot path (server/model.py:47-66 batched) over one batch of synthetic frames per
GPU, inputs already resident in HBM.  Default workload = BASELINE.json configs[1] (256 x 720p 1-face, fp32).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  If the ranks were not started for us (no WORLD_SIZE in the environment) this
process starts them itself -- `python -m torch.distributed.run --nproc-per-node N ... bench.py <same flags>`
as a child, BEFORE anything here touches the GPU -- and exits with the child's code.
  --mode sharded (default): one clip, contiguous time shards per rank, ONE all-gather (RCCL) of the embedding
                            rows per step, drift on every rank (weak scaling: per-GPU batch fixed);
  --mode streams          : BASELINE configs[3] -- one independent clip per GPU, no data-path collective.
  --config {0,1,2,4}      : which BASELINE.json configs[] entry the workload is (1 = headline, default).
  --ingest nv12           : supplementary leg: host NV12 -> pinned H2D -> BGR on the device -> the same path.
  --embed-group G         : the decoupled embedder embeds the crops of G consecutive steps in ONE InceptionResnetV1 call on its own context
                            (default 8 = 2048 faces per call; 1 = inside each step's call).  Results are bit-identical for any G: the
                            embedder's ~100 small dependent launches amortise over more faces (2.16 ms per 256 faces at 256 per call,
                            1.61 ms = 47 % of the f32-MFMA peak at 1024, 1.37 ms = 55 % at 2048; 12 and 16 gain nothing more).
  --driver single|threads : ONE host thread drives every context through trl_detect_embed_begin / _end (default), or one blocking
                            host thread per context (round 2's scheme; --embed-group then groups per worker).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FPS = 30
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense f32-input MFMA peak (= f32 vector peak)

# BASELINE.json configs[] -> workload.  The detector is f32 in every config (a reduced-precision detector cannot
# keep box / NMS index parity); configs[2] runs the embedder on the bf16 matrix cores.
CONFIGS = {
    0: dict(H=360, W=640, batch=240, faces=1, min_face=20, embed="f32", dtype="f32", unique=240,
            name="BASELINE configs[0] shape: the 240 sampled frames (every 4th of 960) of a 640x360 30 fps clip in ONE batch, seeded "
                 "synthetic frames (the reference's sample .mp4 cannot be decoded without OpenCV), fp32 -- the reference's own "
                 "CPU-runnable case; a parity-test shape, not the headline"),
    1: dict(H=720, W=1280, batch=256, faces=1, min_face=20, embed="f32", dtype="f32", unique=256,
            name="BASELINE configs[1]: synthetic 720p 1-face frames, batch=256 per GPU, fp32, MTCNN detect + 80x80 crop + "
                 "InceptionResnetV1 embed + cosine drift score"),
    2: dict(H=1080, W=1920, batch=128, faces=-1, min_face=20, embed="bf16", dtype="f32+bf16", unique=16,
            name="BASELINE configs[2]: synthetic 1080p multi-face (3-5 faces/frame) stream, batch=128 per GPU, f32 detector "
                 "(bit-exact cascade) + bf16-MFMA InceptionResnetV1 embedder, largest face embedded (server/model.py:49)"),
    4: dict(H=2160, W=3840, batch=32, faces=1, min_face=40, embed="fp16", dtype="f32+fp16", unique=4,
            name="BASELINE configs[4]: synthetic 4K frames, MTCNN pyramid 12 scales (min_face_size=40), batch=32 per GPU, f32 detector "
                 "(an fp16 detector cannot keep box/NMS parity: not built, DESIGN.md section 8) + fp16-MFMA InceptionResnetV1 embedder"),
}


def pnet_macs(Hh, Ww, minsize=20, factor=0.709):
    """Algorithmic conv MACs of PNet over the whole pyramid of one frame."""
    m = 12.0 / minsize
    minl = min(Hh, Ww) * m
    s = m
    tot = 0
    while minl >= 12:
        h, w = int(Hh * s + 1), int(Ww * s + 1)
        ph, pw = (h - 1) // 2, (w - 1) // 2          # ceil((h-2)/2)
        tot += (h - 2) * (w - 2) * 270 + (ph - 2) * (pw - 2) * 1440 + (ph - 4) * (pw - 4) * (4608 + 192)
        s *= factor
        minl *= factor
    return tot


def make_clip(cfg, n, seed):
    """n seeded frames of the config's shape.  The numpy generator is slow for big frames, so `unique` frames are
    generated and the rest are horizontal rolls of them (different bytes, same statistics)."""
    import numpy as np
    import truely_amd
    u = min(n, cfg["unique"])
    base = truely_amd.synthetic.synthetic_frames(u, cfg["H"], cfg["W"], seed=seed, faces=cfg["faces"])
    if u == n:
        return base
    out = np.empty((n,) + base.shape[1:], np.uint8)
    for i in range(n):
        out[i] = base[i % u] if i < u else np.roll(base[i % u], 11 * (i // u), axis=1)
    return out


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def host_cores():
    """Cores this process may use: the scheduler affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(frames_np, threads, cfg):
    """Restated reference CPU path (oracle/torch_ref.py: torch CPU fp32) on a bounded sample of the same workload, on ALL the
    host cores this process may use.  Two figures: `value` = ONE frame at a time exactly as
    server/model.py:42-59 drives facenet-pytorch; `batched` = the same restatement with the frames spread over the cores (one
    single-threaded frame pipeline per core -- the CPU's best case for this workload: no cross-thread synchronisation inside the
    small PNet / R-Net convolutions)."""
    import torch
    import truely_amd
    from oracle.torch_ref import TorchRef
    from oracle.oracle import Oracle
    H, W = cfg["H"], cfg["W"]
    sds = truely_amd.weights.synthetic_state_dicts(0)
    orc = Oracle(truely_amd.weights.pack_state_dicts(*sds))
    budget = float(os.environ.get("TRUELY_CPU_BASELINE_SECONDS", "12"))

    def one_frame(ref, fr):
        boxes, _ = ref.detect(fr)
        if boxes is not None and len(boxes) > 0:
            b = boxes[0].astype(int)
            x0, y0, x1, y1 = max(0, b[0]), max(0, b[1]), min(W, b[2]), min(H, b[3])
            if x1 > x0 and y1 > y0:
                face = orc.resize_linear_u8(fr, y0, y1, x0, x1)     # cv2.resize stand-in (no OpenCV here)
                ref.embed(face)

    # (i) reference-faithful: one frame at a time, torch's intra-op threads = all cores
    ref = TorchRef(*sds, threads=threads, min_face_size=cfg["min_face"])
    t0 = time.time()
    done = 0
    while time.time() - t0 < budget:
        one_frame(ref, frames_np[done % len(frames_np)])
        done += 1
    dt = time.time() - t0
    # (ii) batched over the cores: `threads` workers, each a single-threaded pipeline over its own frames (torch releases the GIL)
    import concurrent.futures as cf
    torch.set_num_threads(1)
    shared = TorchRef(*sds, threads=1, min_face_size=cfg["min_face"])      # inference only: safe to share between the workers
    stop = time.time() + budget
    counts = [0] * threads

    def work(j):
        k = 0
        while time.time() < stop:
            one_frame(shared, frames_np[(j + k * threads) % len(frames_np)])
            k += 1
        counts[j] = k

    t1 = time.time()
    with cf.ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    dt2 = time.time() - t1
    torch.set_num_threads(threads)
    return {"value": round(done / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port", "cpu": cpu_model_string(),
            "batched": {"value": round(sum(counts) / dt2, 3), "unit": "frames/s",
                        "how": f"{threads} single-threaded frame pipelines side by side, {sum(counts)} frames in {dt2:.1f} s"},
            "sample": f"{done} frames of the same synthetic {H}p clip in {dt:.1f} s, torch-CPU fp32 restatement of the "
                      f"reference path (oracle/torch_ref.py), one frame at a time like server/model.py:42-59, "
                      f"torch.set_num_threads({threads}) = every core this process may use"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x 16 ms = a 1.6 s timed region by default)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="sharded", choices=["sharded", "streams"])
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pnet-mode", type=int, default=None)
    ap.add_argument("--prelu", default="unit", choices=["unit", "general"],
                    help="synthetic PReLU slopes: 'unit' in [0.05,0.3] (the seeded default: max(v, s*v) fast path); 'general' "
                         "flips some slopes outside [0,1] / negative, which real checkpoints may have (k_pnet_fused<false>)")
    ap.add_argument("--ingest", default="resident", choices=["resident", "nv12"],
                    help="resident (the metric: frames already in HBM) or nv12 (supplementary: pinned host NV12 -> H2D -> "
                         "k_nv12_to_bgr -> the same path, PCIe inclusive)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on a real multi-GPU node; gloo to rehearse N>1 on one GPU")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="batches in flight per GPU (each gets its own context and HIP stream).  1 (default): strictly sequential on "
                         "the device, every kernel runs alone and the HIP event pair around a launch is its duration.  2: +3 %% "
                         "frames/s (the head of one persistent PNet launch fills the tail of the other's), but then the pair also "
                         "counts the wait behind the other context's launch (profiles/round4_pnet_gate_ab.txt)")
    ap.add_argument("--pnet-gate", type=int, default=0, choices=[0, 1],
                    help="with --in-flight >= 2.  1: a fused PNet launch waits for the end of the device's previous call (it runs alone, the "
                         "event pair is its duration; the pyramid kernels in front of it overlap that call's narrow kernels): +1.3 %% over one "
                         "batch in flight.  0 (default): launches of different contexts are left to the hardware queues: +3 %%, no clean clock")
    ap.add_argument("--embed-group", type=int, default=8,
                    help="consecutive steps whose crops the decoupled embedder embeds in ONE InceptionResnetV1 call (trl_detect_crop per "
                         "step into a ring, then one trl_facenet_embed_masked): same bits, ~100 small launches amortised over G x 256 faces")
    ap.add_argument("--driver", default="single", choices=["threads", "single"],
                    help="how the batches in flight are driven: one host thread per context (blocking trl_detect_embed calls, GIL "
                         "released) or ONE host thread over all contexts (trl_detect_embed_begin / _end)")
    ap.add_argument("--master-port", type=int, default=None)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the per-step collective even with ONE rank: exercises the RCCL branch "
                         "(init, all-gather of the result rows, barrier, all-reduce of the time) on a single-GPU box")
    return ap.parse_args(argv)


def rank_command(args, argv, have_gpus, environ=None):
    """(cmd, env) of the child launcher that `--gpus N` without a launcher starts: N fresh ranks under torch.distributed.run on
    this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve), dmabuf IPC for RCCL."""
    if args.backend == "nccl" and have_gpus < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} with RCCL needs {args.gpus} visible GPUs, found {have_gpus} "
                         f"(rehearse on fewer GPUs with --backend gloo)")
    port = args.master_port or (29500 + os.getpid() % 3000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ if environ is None else environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    return cmd, env


def spawn_ranks(args):
    """--gpus N without a launcher: start N fresh ranks as a child `torch.distributed.run` (never re-exec: this
    process has not touched the GPU and never will).  torch.cuda.device_count() does not initialise HIP."""
    import torch
    cmd, env = rank_command(args, sys.argv[1:], torch.cuda.device_count())
    return subprocess.call(cmd, env=env)


def rank_device(local_rank, world, backend, device_count):
    """The device ordinal of a rank.  RCCL: one GPU per rank, LOCAL_RANK is the ordinal (a node with fewer GPUs than ranks is an
    error, not a silent share).  gloo rehearsals may stack several ranks on one GPU."""
    if backend != "nccl":
        return local_rank % max(1, device_count)
    if world > device_count:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, found {device_count}")
    return local_rank


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout carries exactly ONE line, rank 0's JSON: everything else a rank or a library prints there (gloo announces its
    # connections on stdout from C++) goes to stderr instead -- the launcher forwards every rank's stdout to the caller.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    import truely_amd
    from truely_amd.engine import Engine
    from truely_amd.distributed import allgather_embeddings_async

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    local = rank_device(local, world, args.backend, torch.cuda.device_count())   # (device_count() does not initialise HIP)
    torch.cuda.set_device(local)                              # before ANY HIP call of this rank: contexts, RCCL, tensors
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port or (29500 + os.getpid() % 3000)))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    cfg = CONFIGS[args.config]
    H, W = cfg["H"], cfg["W"]
    n = args.batch or cfg["batch"]
    frames_np = make_clip(cfg, n, seed=rank)           # sharded: segment `rank` of the clip; streams: clip `rank`
    import queue
    import threading
    sds = truely_amd.weights.synthetic_state_dicts(0)
    if args.prelu == "general":
        truely_amd.weights.generalise_prelu(sds)
    blob = truely_amd.weights.pack_state_dicts(*sds)
    F = max(1, args.in_flight)
    ekw = dict(device=local, pnet_mode=args.pnet_mode, min_face_size=cfg["min_face"], embed_precision=cfg["embed"])
    engs = [Engine(blob, **ekw) for _ in range(F)]   # one context + workspace per batch in flight
    engs[0].option("pnet_gate", args.pnet_gate)
    streams = [torch.cuda.Stream(dev) for _ in range(F)]
    eng = engs[0]
    drift_eng = Engine(blob, device=local) if F > 1 else eng      # the main thread's context (drift kernels)
    embed_eng = Engine(blob, **ekw) if (args.embed_group > 1 and args.driver == "single") else None   # the decoupled embedder's context
    sharded = args.mode == "sharded"
    frame_count = n * (world if sharded else 1) * 4    # 30 fps clip sampled every 4th frame (model.py:40)

    if args.ingest == "nv12":
        from truely_amd.ingest import bgr_to_nv12, Nv12Uploader
        nv12_host = torch.from_numpy(bgr_to_nv12(frames_np)).pin_memory()   # what a decoder hands over: host NV12 planes
        uploaders = [Nv12Uploader(engs[j], H, W, n) for j in range(F)]
        frames = None
    else:
        frames = torch.from_numpy(frames_np).to(dev)

    pending = [None] * F     # nv12: slot of the batch whose H2D copy was started ahead on worker j's copy stream

    def batch_input(j, more=True):
        """The batch as the detector consumes it.  resident: the device tensor.  nv12: the copy of THIS batch was started while
        the previous one was computing (Nv12Uploader.prefetch: own copy stream, pinned source), so PCIe overlaps the kernels;
        convert on stream j, then start the next copy."""
        if frames is not None:
            return frames
        if pending[j] is None:
            pending[j] = uploaders[j].prefetch(nv12_host)
        out = uploaders[j].convert(pending[j])
        pending[j] = uploaders[j].prefetch(nv12_host) if more else None
        return out

    counts = [n] * world

    def gather_start(out):
        """Main thread, step order on every rank: START the one collective of the path (sharded mode) for this step's rows.  It is
        consumed one step later (gather_finish), so a rank that is momentarily slower does not stall the others inside every step:
        the collective (8 x 0.53 MB, latency-bound) is off the critical path."""
        if use_dist and sharded:
            return allgather_embeddings_async(out["emb"], out["valid"], counts=counts)
        return (out["emb"], out["valid"])

    def gather_finish(h):
        """... and consume it: the time-ordered rows of the whole clip, then the drift state machine (every rank)."""
        emb, valid = h.wait() if hasattr(h, "wait") else h
        d = drift_eng.drift_score(emb, valid, frame_count, FPS)
        d["emb_all"], d["valid_all"] = emb, valid
        return d

    class Lagged:
        """finish() of the steps: the gather of step i is started when its rows exist and finished when step i + 1's rows exist
        (or at drain()): results stay in step order, all of it inside the timed region."""

        def __init__(self):
            self.prev = None            # (out, gather handle) of the previous step
            self.last = (None, None)

        def step(self, out):
            h = gather_start(out)
            if self.prev is not None:
                self.last = (self.prev[0], gather_finish(self.prev[1]))
            self.prev = (out, h)

        def drain(self):
            if self.prev is not None:
                self.last = (self.prev[0], gather_finish(self.prev[1]))
                self.prev = None
            return self.last

    def run_steps(k):
        """k steps.  Worker j runs detect+embed of steps j, j+F, .. on its own stream; results are consumed in step order."""
        acc = {"pnet_ms": 0.0, "pyramid_ms": 0.0, "pnet_kernel_ms": 0.0}
        lag = Lagged()
        G = max(1, args.embed_group)

        def group(engine, j, idxs):
            """detect (+ crop) the steps in idxs on `engine`, embed their faces in one call; yields (out, timings) per step."""
            if G == 1:
                for _i in idxs:
                    out = engine.detect_embed(batch_input(j, _i + F < k))
                    yield out, engine.timings()
                return
            part = []
            for _i in idxs:
                part.append((engine.detect_crop(batch_input(j, _i + F < k)), engine.timings()))
            emb = engine.embed_faces(torch.cat([p["faces"] for p, _ in part]), torch.cat([p["valid"] for p, _ in part]))
            o = 0
            for p, tm in part:
                p["emb"] = emb[o:o + n]; o += n
                del p["faces"]
                yield p, tm

        if F == 1 and args.driver == "threads":
            for g0 in range(0, k, G):
                for out, tm in group(eng, 0, list(range(g0, min(k, g0 + G)))):
                    lag.step(out)
                    for key in acc:
                        acc[key] += tm[key]
            return lag.drain(), acc
        if args.driver == "single":
            # ONE host thread over all contexts (pipeline.detect_embed_overlapped): step i is queued on context i % F
            # (trl_detect_embed_begin returns without synchronising) and finished right before that context is needed again; with
            # G > 1 the cascades write their crops into a ring and ONE embedder context embeds every G consecutive steps' faces in
            # one call on its own stream (bit-identical results, the embedder's ~100 launches amortise over G x 256 faces).
            from truely_amd.pipeline import detect_embed_overlapped

            def on_detect(i, j):
                tm = engs[j].timings()
                for key in acc:
                    acc[key] += tm[key]

            def on_result(i, out):
                lag.step(out)

            detect_embed_overlapped(engs, lambda i, j: batch_input(j, i + F < k), on_result=on_result, streams=streams,
                                    embed_group=G, embed_engine=embed_eng, n_batches=k, on_detect=on_detect)
            return lag.drain(), acc
        qs = [queue.Queue() for _ in range(F)]

        def worker(j):
            try:
                torch.cuda.set_device(local)
                with torch.cuda.stream(streams[j]):
                    mine = list(range(j, k, F))
                    for g0 in range(0, len(mine), G):
                        res = list(group(engs[j], j, mine[g0:g0 + G]))
                        streams[j].synchronize()          # the consumer runs on another stream
                        for item in res:
                            qs[j].put(item)
            except BaseException as e:                     # surfaced by the consumer
                qs[j].put(e)

        ths = [threading.Thread(target=worker, args=(j,), daemon=True) for j in range(F)]
        for t in ths:
            t.start()
        for i in range(k):
            item = qs[i % F].get()
            if isinstance(item, BaseException):
                raise item
            out, tm = item
            lag.step(out)
            for key in acc:
                acc[key] += tm[key]
        for t in ths:
            t.join()
        return lag.drain(), acc

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if embed_eng is not None:
        # Set-up, not a step: size the decoupled embedder's workspace for a FULL group before anything is timed.  The warm-up's
        # W steps end in a partial group (W < G), so the first full group -- and with it a hipFree + hipMalloc of ~10 GB inside
        # trl_ensure -- would otherwise fall into the timed region (usually milliseconds, occasionally hundreds on a busy host).
        Sf = 80 if eng.cfg.embed_mode == 0 else 160
        Gf = max(1, args.embed_group)
        embed_eng.embed_faces(torch.zeros((Gf * n, Sf, Sf, 3), dtype=torch.float32, device=dev), torch.ones((Gf * n,), dtype=torch.uint8, device=dev))
        torch.cuda.synchronize()
    if args.warmup > 0:
        run_steps(max(args.warmup, F))
    fence()
    t0 = time.perf_counter()
    (out, d), acc = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    pnet_ms, pyr_ms = acc["pnet_ms"], acc["pyramid_ms"]
    if use_dist:
        tdev = dev if args.backend == "nccl" else torch.device("cpu")
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        scores = [None] * world
        dist.all_gather_object(scores, int(d["score"]))        # after the timed region: reporting only
    else:
        scores = [int(d["score"])]

    # After the timed region (never part of `value`): the dominant kernel alone -- a few strictly sequential steps on one context,
    # nothing else queued on the GPU -- as a cross-check of the figure above (with --in-flight 2 it is the launch duration free of
    # the queueing behind the other context's launch).
    iso_ms = None
    if rank == 0 and eng.cfg.pnet_mode == 0 and frames is not None:
        torch.cuda.synchronize()
        tot = 0.0
        for _ in range(4):
            eng.detect_embed(frames)
            tot += eng.timings()["pnet_ms"]
        iso_ms = tot / 4

    if rank == 0:
        tm = eng.timings()
        macs = pnet_macs(H, W, cfg["min_face"]) * n                  # per launch set of one step on this rank
        launches = max(1, tm["pnet_launches"])
        # Duration of the dominant kernel per step, live over the timed region: the HIP event pair around every fused launch, on the
        # stream it is launched on, summed over the region's steps.  Fused launches of the contexts in flight are ordered one
        # after the other on the device (trl_pnet.hip: the persistent grid fills every CU), so the pair times the launch's
        # execution -- the duration rocprofv3 reports for the same command (profiles/round4_bench_kernel_stats.csv) -- and not
        # the wait behind the other context's launch.
        pnet_s = pnet_ms / 1e3 / args.steps
        achieved = 2.0 * macs / pnet_s / 1e12
        traffic = None
        tpaths = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pnet_traffic.json"))
        tpath = os.path.join(ROOT, "profiles", tpaths[-1]) if tpaths else ""      # the newest round's PMC passes
        if eng.cfg.pnet_mode == 0 and args.config == 1 and n == cfg["batch"] and tpath:
            # HBM bytes per launch of k_pnet_fused from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE)
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
        emb_all = d["emb_all"].cpu().numpy()
        if world > 1 and sharded:
            par = f"one clip frame-sharded x{world} (contiguous time shards), 1 all-gather of embedding rows per step ({args.backend})"
        elif world > 1:
            par = f"{world} independent clips, one per GPU (BASELINE configs[3]), no data-path collective"
        else:
            par = "single GPU"
        res = {
            "metric": f"frames/sec (detect+embed+drift) {H}p", "value": round(n * world * args.steps / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": cfg["name"] + ("" if args.mode == "sharded" or world == 1 else
                                                  " -- as BASELINE configs[3]: 8-video concurrent ingest, one stream per GPU"),
                       "frames_per_gpu": n, "height": H, "width": W, "weights": "seeded synthetic (no checkpoints offline)",
                       "prelu_slopes": args.prelu, "min_face_size": cfg["min_face"], "pyramid_levels": eng.levels(H, W),
                       "valid_faces": int(out["valid"].sum().item()), "score": d["score"], "scores": scores,
                       "candidates_per_step": dict(zip(("rnet", "onet"), eng.stage_totals())),
                       "emb_crc32": zlib.crc32(emb_all.tobytes()), "mode": args.mode, "ingest": args.ingest,
                       "pnet_path": "fused" if eng.cfg.pnet_mode == 0 else "generic layers",
                       "parallelism": par, "backend": args.backend if use_dist else None, "batches_in_flight": F, "driver": args.driver,
                       "embed_group": max(1, args.embed_group)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "kernel": "k_pnet_fused (PNet over the pyramid: 83% of the conv FLOPs at 720p)",
                         "flop_per_step": 2.0 * macs, "kernel_ms_per_step": round(pnet_s * 1e3, 3), "launches_per_step": launches,
                         "kernel_clock": "HIP event pair around each launch on its stream, summed over the timed region",
                         "kernel_ms_alone": None if iso_ms is None else round(iso_ms, 3),
                         "frac_alone": None if iso_ms is None else round(2.0 * macs / (iso_ms / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                         "pyramid_ms_per_step": round(pyr_ms / args.steps, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = host_cores()
            res["cpu_baseline"] = cpu_baseline(frames_np, threads, cfg)
        else:
            res["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

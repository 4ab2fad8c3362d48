"""bench.py -- BASELINE.json metric: frames/sec (detect + embed + drift), synthetic 720p clips.

One "step" = one pass of the hot path (server/model.py:47-66 batched) over one batch of 256
synthetic 720p 1-face frames per GPU, inputs already resident in HBM (BASELINE.json configs[1]).
N > 1: one process per GPU, each rank owns a contiguous time shard, one RCCL all-gather of the
embeddings, drift on every rank (weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, BATCH, FPS = 720, 1280, 256, 30
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense f32-input MFMA peak (= f32 vector peak)


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


def cpu_baseline(frames_np, threads):
    """Restated reference CPU path (oracle/torch_ref.py: torch CPU fp32, ONE frame at a time exactly
    as server/model.py:42-59 drives facenet-pytorch), on a bounded sample of the same workload."""
    import numpy as np
    import torch
    import truely_amd
    from oracle.torch_ref import TorchRef
    from oracle.oracle import Oracle
    sds = truely_amd.weights.synthetic_state_dicts(0)
    ref = TorchRef(*sds, threads=threads)
    orc = Oracle(truely_amd.weights.pack_state_dicts(*sds))
    t0 = time.time()
    done = 0
    budget = float(os.environ.get("TRUELY_CPU_BASELINE_SECONDS", "20"))
    while time.time() - t0 < budget:
        fr = frames_np[done % len(frames_np)]
        boxes, _ = ref.detect(fr)
        if boxes is not None and len(boxes) > 0:
            b = boxes[0].astype(int)
            x0, y0, x1, y1 = max(0, b[0]), max(0, b[1]), min(W, b[2]), min(H, b[3])
            if x1 > x0 and y1 > y0:
                face = orc.resize_linear_u8(fr, y0, y1, x0, x1)     # cv2.resize stand-in (no OpenCV here)
                ref.embed(face)
        done += 1
    dt = time.time() - t0
    return {"value": round(done / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{done} frames of the same synthetic 720p clip in {dt:.1f} s, torch-CPU fp32 restatement of the "
                      f"reference path (oracle/torch_ref.py), one frame at a time like server/model.py:42-59"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pnet-mode", type=int, default=None)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on a real multi-GPU node; gloo to rehearse the N>1 path on one GPU")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight per GPU: each gets its own context, HIP stream and host thread, so the low-occupancy tail "
                         "of one batch (NMS, FaceNet's 1x1-spatial layers) overlaps the wide kernels of the next; 1 = strictly sequential")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import truely_amd
    from truely_amd.engine import Engine
    from truely_amd.distributed import allgather_embeddings

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.backend != "nccl":
        local = local % max(1, torch.cuda.device_count())      # rehearsal: several ranks may share one GPU under gloo
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    n = args.batch
    frames_np = truely_amd.synthetic.synthetic_frames(n, H, W, seed=rank, faces=1)
    frames = torch.from_numpy(frames_np).to(dev)
    import queue
    import threading
    blob = truely_amd.weights.synthetic_blob(0)
    F = max(1, args.in_flight)
    engs = [Engine(blob, device=local, pnet_mode=args.pnet_mode) for _ in range(F)]   # one context + workspace per batch in flight
    streams = [torch.cuda.Stream(dev) for _ in range(F)]
    eng = engs[0]
    drift_eng = Engine(blob, device=local) if F > 1 else eng      # the main thread's context (drift kernels)
    frame_count = n * world * 4      # 30 fps clip sampled every 4th frame (model.py:40)

    def finish(out):
        """Main thread, step order on every rank: the one collective of the path, then the drift state machine."""
        if world > 1:
            emb, valid = allgather_embeddings(out["emb"], out["valid"])
        else:
            emb, valid = out["emb"], out["valid"]
        return drift_eng.drift_score(emb, valid, frame_count, FPS)

    def run_steps(k):
        """k steps.  Worker j runs detect+embed of steps j, j+F, .. on its own stream; results are consumed in step order."""
        acc = {"pnet_ms": 0.0, "pyramid_ms": 0.0, "pnet_kernel_ms": 0.0}
        last = (None, None)
        if F == 1:
            for _ in range(k):
                out = eng.detect_embed(frames)
                d = finish(out)
                tm = eng.timings()
                for key in acc:
                    acc[key] += tm[key]
                last = (out, d)
            return last, acc
        qs = [queue.Queue() for _ in range(F)]

        def worker(j):
            try:
                torch.cuda.set_device(local)
                with torch.cuda.stream(streams[j]):
                    for _i in range(j, k, F):
                        out = engs[j].detect_embed(frames)
                        streams[j].synchronize()          # the consumer runs on another stream
                        qs[j].put((out, engs[j].timings()))
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
            d = finish(out)
            for key in acc:
                acc[key] += tm[key]
            last = (out, d)
        for t in ths:
            t.join()
        return last, acc

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run_steps(max(args.warmup, F))
    fence()
    t0 = time.perf_counter()
    (out, d), acc = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    pnet_ms, pyr_ms, pnet_kernel_ms = acc["pnet_ms"], acc["pyramid_ms"], acc["pnet_kernel_ms"]
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        tm = eng.timings()
        macs = pnet_macs(H, W) * n                  # per launch set of one step on this rank
        launches = max(1, tm["pnet_launches"])
        # Duration of the dominant kernel per step.  Two clocks, both live over the timed region: HIP events recorded around the
        # launch on its stream, and the launch's execution span on the device wall clock (first workgroup start -> last workgroup
        # end), which is what rocprofv3 reports as the kernel's duration.  With one batch in flight they agree; with two, the event
        # pair also counts the time the launch queues behind the other context's kernels, so the span is the kernel's time.
        use_span = eng.cfg.pnet_mode == 0 and pnet_kernel_ms > 0
        pnet_s = (pnet_kernel_ms if use_span else pnet_ms) / 1e3 / args.steps
        achieved = 2.0 * macs / pnet_s / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "round1_pnet_traffic.json")
        if eng.cfg.pnet_mode == 0 and n == BATCH and os.path.exists(tpath):
            # HBM bytes per launch of k_pnet_fused from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE)
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
        res = {
            "metric": "frames/sec (detect+embed+drift) 720p", "value": round(n * world * args.steps / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 720p 1-face frames, batch=256 per GPU, fp32, "
                                   "MTCNN detect + 80x80 crop + InceptionResnetV1 embed + cosine drift score",
                       "frames_per_gpu": n, "height": H, "width": W, "weights": "seeded synthetic (no checkpoints offline)",
                       "valid_faces": int(out["valid"].sum().item()), "score": d["score"],
                       "pnet_path": "fused" if eng.cfg.pnet_mode == 0 else "generic layers",
                       "parallelism": f"frame-sharded x{world}, 1 all-gather of embeddings" if world > 1 else "single GPU",
                       "batches_in_flight": F},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "kernel": "k_pnet_fused (PNet over the pyramid: 83% of the conv FLOPs at 720p)",
                         "flop_per_step": 2.0 * macs, "kernel_ms_per_step": round(pnet_s * 1e3, 3), "launches_per_step": launches,
                         "kernel_clock": "device wall clock span of the launch" if use_span else "HIP events",
                         "kernel_ms_per_step_hip_events": round(pnet_ms / args.steps, 3),
                         "pyramid_ms_per_step": round(pyr_ms / args.steps, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, len(os.sched_getaffinity(0)))
            res["cpu_baseline"] = cpu_baseline(frames_np, threads)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

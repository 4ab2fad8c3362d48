"""Several batches in flight on one GPU.

Contexts (``Engine`` instances) are independent: own workspaces, thread-local error slot.  Driving two of them
from two host threads on two HIP streams lets the narrow tail of one batch (NMS, per-candidate tails, FaceNet's
1x1-spatial layers) overlap the wide kernels of the next (+7-10 % throughput on MI355X).
The reference processes frames strictly one at a time (server/model.py:42-59); batches are independent until the
drift state machine (model.py:60-66), which consumes their embeddings in order.
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Callable, Iterable, List, Sequence

import torch

from .engine import Engine


def detect_embed_grouped(engine: Engine, batches: Sequence, embed_group: int = 1) -> List[dict]:
    """``detect_embed`` of every batch on ONE engine, with the embedder called once per ``embed_group`` batches: the cascade
    and the crops run per batch (``trl_detect_crop``), the faces of a group are embedded together (``trl_facenet_embed_masked``).
    InceptionResnetV1 at the reference's 80x80 crops is ~100 small dependent launches; their fixed cost amortises over more
    faces (2.22 ms per 256 faces alone, 1.92 at 512, 1.77 at 768 per call).  Every output element is computed by the same
    accumulation chain whatever the grouping: results are bit-identical to per-batch ``detect_embed``."""
    G = max(1, int(embed_group))
    if G == 1:
        return [engine.detect_embed(b) for b in batches]
    outs: List[dict] = []
    for g0 in range(0, len(batches), G):
        part = [engine.detect_crop(b) for b in batches[g0:g0 + G]]
        emb = engine.embed_faces(torch.cat([p["faces"] for p in part]), torch.cat([p["valid"] for p in part]))
        k = 0
        for p in part:
            n = p["valid"].shape[0]
            p["emb"] = emb[k:k + n]
            k += n
            del p["faces"]
            outs.append(p)
    return outs


def detect_embed_overlapped(engines: Sequence[Engine], batches, on_result: Callable[[int, dict], None] | None = None,
                            streams=None, embed_group: int = 1, embed_engine: Engine | None = None, n_batches: int | None = None,
                            on_detect: Callable[[int, int], None] | None = None) -> List[dict]:
    """Batches in flight driven by ONE host thread.  Batch i is queued on ``engines[i % F]`` (``trl_detect_embed_begin`` /
    ``trl_detect_crop_begin``: return without synchronising) on that engine's stream and finished (``trl_detect_embed_end``)
    right before the engine is needed again, so F batches are in flight and no thread per context is needed.

    ``embed_group`` = G > 1: the embedder is decoupled from the batches.  The cascades write their crops into consecutive slots
    of a ring; every G consecutive batches (in batch order, whichever engine produced them) are embedded by ONE
    ``trl_facenet_embed_masked`` call of ``embed_engine`` (a context of its own, created and cached on first use), queued behind
    the cascade that produced the group's last batch.  InceptionResnetV1 at the reference's 80x80 crops is ~100 small dependent
    launches whose fixed cost amortises over the faces of a call: 2.16 ms per 256 faces at 256 per call, 1.70 ms at 768, 1.60 ms
    (47.6 % of the f32-MFMA peak) at 1,024.  Every output element is the same accumulation chain whatever the grouping: results
    are bit-identical to per-batch ``detect_embed``.

    If a call fails (a capacity overflow in a crowded batch, an allocation failure) the calls still queued on the other engines are
    finished before the exception propagates, so every engine can be used again.

    ``batches``: a sequence, or a callable ``(i, j) -> batch`` with ``n_batches`` (the bench's NV12 uploader prefetches per engine).
    ``on_detect(i, j)`` is called when batch i's cascade has finished on engine j (timing hooks); ``on_result(i, out)`` in batch
    order once the batch's embeddings exist.  Returns the per-batch results in batch order."""
    F = len(engines)
    if F == 0:
        raise ValueError("need at least one engine")
    if callable(batches):
        get, K = batches, int(n_batches)
    else:
        seq = list(batches)
        get, K = (lambda i, j: seq[i]), len(seq)
    G = max(1, int(embed_group))
    dev = engines[0].device
    streams = streams or [torch.cuda.Stream(dev) for _ in range(F)]
    outs: List[dict] = []
    inflight = [None] * F                                # batch index queued on engine j

    def deliver(i, out):
        if on_result:
            on_result(i, out)
        outs.append(out)

    def abandon():
        """An exception is propagating: finish whatever is still queued so that no engine stays 'busy'."""
        for j in range(F):
            if inflight[j] is not None:
                try:
                    with torch.cuda.stream(streams[j]):
                        engines[j].detect_embed_end()
                except Exception:  # noqa: BLE001 - the first error is the one the caller sees
                    pass
                inflight[j] = None

    if G == 1:
        def finish(j):
            i, inflight[j] = inflight[j], None           # (whatever _end does, the call is over)
            with torch.cuda.stream(streams[j]):
                out = engines[j].detect_embed_end()      # synchronises stream j: the tensors are safe to use on any stream
            if on_detect:
                on_detect(i, j)
            deliver(i, out)

        try:
            for i in range(K):
                j = i % F
                if inflight[j] is not None:
                    finish(j)
                with torch.cuda.stream(streams[j]):
                    engines[j].detect_embed_begin(get(i, j))
                inflight[j] = i
            for k in range(K, K + F):                    # drain in batch order
                if inflight[k % F] is not None:
                    finish(k % F)
        except BaseException:
            abandon()
            raise
        return outs

    # ---- decoupled, grouped embedder ----------------------------------------------------------------------------------
    emb_eng = embed_engine or getattr(engines[0], "_embedder", None)
    if emb_eng is None:                                  # its own context (its workspace must not alias a cascade's), built once
        emb_eng = engines[0]._embedder = engines[0].clone()
    # The embedder call of a group is queued on the stream of the engine that produced the group's LAST batch, i.e. behind that
    # cascade and in front of that engine's next one -- the position the embedder has inside a per-batch trl_detect_embed.  (A
    # third stream of its own was measured: same throughput, but its ~100 small launches then interleave with BOTH cascades'
    # persistent PNet launches and stretch them, 7.1 -> 8.5 ms per launch.)
    S = 80 if engines[0].cfg.embed_mode == 0 else 160
    R = G * ((2 * G + F + G - 1) // G)                   # slots: F being written, G being collected, G being embedded; a multiple
                                                         # of G so that a group never straddles the ring's end
    ring = {"n": 0, "faces": None, "valid": None}
    group: List[tuple] = []                              # (batch index, out) collected for the next embedder call, consecutive slots
    pending: List[tuple] = []                            # (event, emb, [(i, out), ...]) embedder calls in flight, oldest first

    def slot_views(i, n):
        if ring["faces"] is None:
            ring["n"] = n
            with torch.cuda.stream(torch.cuda.default_stream(dev)):     # shared by every stream of the run: not owned by one of them
                ring["faces"] = torch.empty((R, n, S, S, 3), dtype=torch.float32, device=dev)
                ring["valid"] = torch.empty((R, n), dtype=torch.uint8, device=dev)
        if n > ring["n"]:
            raise ValueError("batches must not grow after the first one")
        k = i % R
        return ring["faces"][k, :n], ring["valid"][k, :n]

    def retire(block_for: int | None = None):
        """Deliver finished embedder calls (oldest first); with ``block_for`` = a batch index, wait for every call that still
        reads the ring slot that batch is about to overwrite."""
        while pending:
            ev, emb, members = pending[0]
            must = block_for is not None and members[0][0] <= block_for - R
            if not (must or ev.query()):
                break
            ev.synchronize()
            o = 0
            for i, out in members:
                n = out["valid"].shape[0]
                out["emb"] = emb[o:o + n]
                o += n
                out["valid"] = out["valid"].clone()      # the ring slot is reused
                del out["faces"]
                deliver(i, out)
            pending.pop(0)

    own = os.environ.get("TRUELY_EMBED_STREAM", "producer")      # experiment: "own" / "own_low" = a third stream (low priority)
    own_stream = None
    if own != "producer":
        lo, hi = -1, 0
        try:
            lo, hi = torch.cuda.Stream.priority_range()
        except Exception:  # noqa: BLE001
            pass
        own_stream = torch.cuda.Stream(dev, priority=max(lo, hi)) if own == "own_low" else torch.cuda.Stream(dev)

    last_embed = [None]                                  # event behind the most recent embedder call

    def flush_group(j):
        if not group:
            return
        es = own_stream or streams[j]
        i0, n_full = group[0][0], ring["n"]
        k0 = i0 % R
        cnt = sum(out["valid"].shape[0] for _, out in group)
        if len(group) == 1:
            faces, valid = group[0][1]["faces"], group[0][1]["valid"]
        else:                                            # consecutive full slots (+ an optional short last one): one contiguous view
            faces = ring["faces"].view(R * n_full, S, S, 3)[k0 * n_full:k0 * n_full + cnt]
            valid = ring["valid"].view(R * n_full)[k0 * n_full:k0 * n_full + cnt]
        with torch.cuda.stream(es):                      # the crops are complete: their cascades were synchronised by _end
            if last_embed[0] is not None:
                es.wait_event(last_embed[0])             # the embedder context has ONE workspace: its calls run one after the
                                                         # other even when consecutive groups end on different engines' streams
            emb = emb_eng.embed_faces(faces, valid)      # queues ~100 launches and returns
            ev = torch.cuda.Event()
            ev.record(es)
        last_embed[0] = ev
        pending.append((ev, emb, list(group)))
        group.clear()

    def finish(j):
        i, inflight[j] = inflight[j], None
        with torch.cuda.stream(streams[j]):
            out = engines[j].detect_embed_end()
        if on_detect:
            on_detect(i, j)
        done[i] = out

    done: dict = {}                                      # cascades finished out of batch order wait here
    nxt = [0]                                            # next batch index to join a group

    def collect(j):
        while nxt[0] in done:
            i = nxt[0]
            out = done.pop(i)
            n = out["valid"].shape[0]
            group.append((i, out))
            nxt[0] += 1
            short = n < ring["n"]
            wraps = (i + 1) % R == 0                     # the next slot is not adjacent in memory
            if len(group) == G or short or wraps or i == K - 1:
                flush_group(j)

    try:
        for i in range(K):
            j = i % F
            if inflight[j] is not None:
                finish(j)
                collect(j)
            retire(block_for=i)
            with torch.cuda.stream(streams[j]):          # (the batch source may queue work of its own: NV12 conversion)
                b = get(i, j)
                fv, vv = slot_views(i, int(b.shape[0]))
                engines[j].detect_embed_begin(b, crop=True, faces=fv, valid=vv)
            inflight[j] = i
        for k in range(K, K + F):
            if inflight[k % F] is not None:
                finish(k % F)
                collect(k % F)
        flush_group((K - 1) % F if K else 0)
        while pending:
            retire(block_for=pending[0][2][0][0] + R)    # wait for the oldest call
    except BaseException:
        abandon()
        if last_embed[0] is not None:
            last_embed[0].synchronize()                  # the embedder's queued launches read the ring: let them end before it is freed
        raise
    return outs


def detect_embed_pipelined(engines: Sequence[Engine], batches: Iterable, on_result: Callable[[int, dict], None] | None = None,
                           embed_group: int = 1) -> List[dict]:
    """``engines[j]`` processes batches j, j+F, j+2F, ... (F = len(engines)) on its own stream and thread; with
    ``embed_group`` > 1 each engine embeds the faces of that many of ITS consecutive batches in one call (see
    ``detect_embed_grouped``).  Returns the per-batch results in batch order; ``on_result(i, out)`` (optional) is called in
    batch order on the calling thread as results arrive."""
    batches = list(batches)
    F = len(engines)
    if F == 0:
        raise ValueError("need at least one engine")
    if F == 1 or len(batches) <= 1:
        outs = detect_embed_grouped(engines[0], batches, embed_group)
        if on_result:
            for i, out in enumerate(outs):
                on_result(i, out)
        return outs
    dev = engines[0].device
    streams = [torch.cuda.Stream(dev) for _ in range(F)]
    qs = [queue.Queue() for _ in range(F)]

    def worker(j):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[j]):
                mine = list(range(j, len(batches), F))
                G = max(1, int(embed_group))
                for g0 in range(0, len(mine), G):
                    outs_j = detect_embed_grouped(engines[j], [batches[i] for i in mine[g0:g0 + G]], G)
                    streams[j].synchronize()      # the consumer uses the tensors on another stream
                    for out in outs_j:
                        qs[j].put(out)
        except BaseException as e:                 # surfaced by the consumer
            qs[j].put(e)

    ths = [threading.Thread(target=worker, args=(j,), daemon=True) for j in range(F)]
    for t in ths:
        t.start()
    outs = []
    try:
        for i in range(len(batches)):
            item = qs[i % F].get()
            if isinstance(item, BaseException):
                raise item
            if on_result:
                on_result(i, item)
            outs.append(item)
    finally:
        for t in ths:
            t.join()
    return outs

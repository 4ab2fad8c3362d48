"""Several batches in flight on one GPU.

Contexts (``Engine`` instances) are independent: own workspaces, thread-local error slot.  Driving two of them
from two host threads on two HIP streams lets the narrow tail of one batch (NMS, per-candidate tails, FaceNet's
1x1-spatial layers) overlap the wide kernels of the next (+7-10 % throughput on MI355X).
The reference processes frames strictly one at a time (server/model.py:42-59); batches are independent until the
drift state machine (model.py:60-66), which consumes their embeddings in order.
"""
from __future__ import annotations

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


class Overlapped:
    """Batches in flight driven by ONE host thread, as a stream: ``push(batch)`` queues the next batch, ``finish()`` drains.

    Batch i is queued on ``engines[i % F]`` (``trl_detect_embed_begin`` / ``trl_detect_crop_begin``: return without synchronising)
    on that engine's stream and finished (``trl_detect_embed_end``) right before the engine is needed again, so F batches are in
    flight and no thread per context is needed.  The number of batches need not be known (``model.run`` streams a clip).

    ``embed_group`` = G > 1: the embedder is decoupled from the batches.  The cascades write their crops into consecutive slots
    of a ring; every G consecutive batches (in batch order, whichever engine produced them) are embedded by ONE
    ``trl_facenet_embed_masked`` call of ``embed_engine`` (a context of its own, created and cached on first use), queued behind
    the cascade that produced the group's last batch -- the position the embedder has inside a per-batch ``trl_detect_embed``.  (A
    third stream of its own was measured in round 3: same throughput, but its ~100 small launches then interleave with BOTH
    cascades' persistent PNet launches and stretch them, 7.1 -> 8.5 ms per launch.)  InceptionResnetV1 at the reference's 80x80
    crops is ~100 small dependent launches whose fixed cost amortises over the faces of a call: 2.16 ms per 256 faces at 256 per
    call, 1.60 ms at 1,024, 1.38 ms at 2,048.  Every output element is the same accumulation chain whatever the grouping:
    results are bit-identical to per-batch ``detect_embed``.

    ``on_detect(i, j)`` is called when batch i's cascade has finished on engine j (timing hooks); ``on_result(i, out)`` in batch
    order once the batch's embeddings exist (results arrive up to F + 2G batches after their push).  Delivered tensors are safe
    to use on the caller's current stream.  If a call fails, ``abandon()`` (called by the wrappers on any exception) finishes
    the calls still queued on the other engines, so every engine can be used again."""

    def __init__(self, engines: Sequence[Engine], on_result: Callable[[int, dict], None] | None = None, streams=None,
                 embed_group: int = 1, embed_engine: Engine | None = None, on_detect: Callable[[int, int], None] | None = None,
                 collect: bool = True):
        self.engines = list(engines)
        self.F = F = len(self.engines)
        if F == 0:
            raise ValueError("need at least one engine")
        self.G = G = max(1, int(embed_group))
        self.dev = self.engines[0].device
        self.streams = streams or [torch.cuda.Stream(self.dev) for _ in range(F)]
        self.on_result, self.on_detect, self.keep = on_result, on_detect, collect
        self.outs: List[dict] = []
        self.inflight = [None] * F                       # batch index queued on engine j
        self.count = 0                                   # batches pushed
        if G > 1:
            emb_eng = embed_engine or getattr(self.engines[0], "_embedder", None)
            if emb_eng is None:                          # its own context (its workspace must not alias a cascade's), built once
                emb_eng = self.engines[0]._embedder = self.engines[0].clone()
            if any(emb_eng is e for e in self.engines):
                raise ValueError("embed_engine must be a context of its own: a cascade engine has a call in flight when the embedder runs")
            self.emb_eng = emb_eng
            self.S = 80 if self.engines[0].cfg.embed_mode == 0 else 160
            self.R = G * ((2 * G + F + G - 1) // G)      # slots: F being written, G being collected, G being embedded; a multiple
                                                         # of G so that a group never straddles the ring's end
            self.ring = {"n": 0, "faces": None, "valid": None}
            self.group: List[tuple] = []                 # (batch index, out) collected for the next embedder call, consecutive slots
            self.pending: List[tuple] = []               # (event, emb, valid copy, [(i, out), ...]) embedder calls in flight, oldest first
            self.last_embed = None                       # event behind the most recent embedder call
            self.done: dict = {}                         # cascades finished out of batch order wait here
            self.nxt = 0                                 # next batch index to join a group

    # ---- common -------------------------------------------------------------------------------------------------------
    def _deliver(self, i, out):
        cur = torch.cuda.current_stream(self.dev)
        for v in out.values():                           # allocated under another stream: tell the allocator who reads them now
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(cur)
        if self.on_result:
            self.on_result(i, out)
        if self.keep:
            self.outs.append(out)

    def abandon(self):
        """An exception is propagating: finish whatever is still queued so that no engine stays 'busy'."""
        for j in range(self.F):
            if self.inflight[j] is not None:
                try:
                    with torch.cuda.stream(self.streams[j]):
                        self.engines[j].detect_embed_end()
                except Exception:  # noqa: BLE001 - the first error is the one the caller sees
                    pass
                self.inflight[j] = None
        if self.G > 1 and self.last_embed is not None:
            self.last_embed.synchronize()                # the embedder's queued launches read the ring: let them end before it is freed

    def push(self, batch):
        """Queue the next batch: a uint8 (n, H, W, 3) tensor / array, or a callable ``j -> batch`` evaluated under engine j's stream
        (a source that queues device work of its own: the NV12 conversion)."""
        i, j = self.count, self.count % self.F
        self.count += 1
        if self.inflight[j] is not None:
            self._finish(j)
            if self.G > 1:
                self._collect(j)
        if self.G > 1:
            self._retire(block_for=i)
        with torch.cuda.stream(self.streams[j]):
            b = batch(j) if callable(batch) else batch
            if self.G > 1:
                fv, vv = self._slot_views(i, int(b.shape[0]))
                self.engines[j].detect_embed_begin(b, crop=True, faces=fv, valid=vv)
            else:
                self.engines[j].detect_embed_begin(b)
        self.inflight[j] = i

    def finish(self) -> List[dict]:
        """Drain: every pushed batch is finished, embedded and delivered (in batch order)."""
        K, F = self.count, self.F
        for k in range(K, K + F):
            j = k % F
            if self.inflight[j] is not None:
                self._finish(j)
                if self.G > 1:
                    self._collect(j)
        if self.G > 1:
            self._flush_group((K - 1) % F if K else 0)
            while self.pending:
                self._retire(block_for=self.pending[0][3][0][0] + self.R)    # wait for the oldest call
        return self.outs

    def _finish(self, j):
        i, self.inflight[j] = self.inflight[j], None     # (whatever _end does, the call is over)
        with torch.cuda.stream(self.streams[j]):
            out = self.engines[j].detect_embed_end()     # synchronises stream j
        if self.on_detect:
            self.on_detect(i, j)
        if self.G > 1:
            self.done[i] = out
        else:
            self._deliver(i, out)

    # ---- decoupled, grouped embedder ------------------------------------------------------------------------------------
    def _slot_views(self, i, n):
        ring = self.ring
        if ring["faces"] is None:
            ring["n"] = n
            with torch.cuda.stream(torch.cuda.default_stream(self.dev)):     # shared by every stream of the run: not owned by one of them
                try:                                     # (no hipMemGetInfo query here: on a shared host that ioctl can stall for 100s of ms)
                    ring["faces"] = torch.empty((self.R, n, self.S, self.S, 3), dtype=torch.float32, device=self.dev)
                    ring["valid"] = torch.empty((self.R, n), dtype=torch.uint8, device=self.dev)
                except torch.cuda.OutOfMemoryError as e:
                    need = self.R * n * (self.S * self.S * 3 * 4 + 1)
                    raise RuntimeError(f"the crop ring of embed_group={self.G} needs {need >> 20} MiB ({self.R} slots of {n} crops): "
                                       f"lower embed_group or the batch size") from e
        if n > ring["n"]:
            raise ValueError("batches must not grow after the first one")
        k = i % self.R
        return ring["faces"][k, :n], ring["valid"][k, :n]

    def _retire(self, block_for: int | None = None):
        """Deliver finished embedder calls (oldest first); with ``block_for`` = a batch index, wait for every call that still
        reads the ring slot that batch is about to overwrite."""
        while self.pending:
            ev, emb, valid, members = self.pending[0]
            must = block_for is not None and members[0][0] <= block_for - self.R
            if not (must or ev.query()):
                break
            ev.synchronize()
            o = 0
            for i, out in members:
                n = out["valid"].shape[0]
                out["emb"] = emb[o:o + n]
                out["valid"] = valid[o:o + n]            # the copy made behind the embedder call (the ring slot is reused)
                o += n
                del out["faces"]
                self._deliver(i, out)
            self.pending.pop(0)

    def _flush_group(self, j):
        group = self.group
        if not group:
            return
        es = self.streams[j]
        i0, n_full, R, S = group[0][0], self.ring["n"], self.R, self.S
        k0 = i0 % R
        cnt = sum(out["valid"].shape[0] for _, out in group)
        if len(group) == 1:
            faces, valid = group[0][1]["faces"], group[0][1]["valid"]
        else:                                            # consecutive full slots (+ an optional short last one): one contiguous view
            faces = self.ring["faces"].view(R * n_full, S, S, 3)[k0 * n_full:k0 * n_full + cnt]
            valid = self.ring["valid"].view(R * n_full)[k0 * n_full:k0 * n_full + cnt]
        with torch.cuda.stream(es):                      # the crops are complete: their cascades were synchronised by _end
            if self.last_embed is not None:
                es.wait_event(self.last_embed)           # the embedder context has ONE workspace: its calls run one after the
                                                         # other even when consecutive groups end on different engines' streams
            emb = self.emb_eng.embed_faces(faces, valid) # queues ~100 launches and returns
            vcopy = valid.clone()                        # behind the embedder, in front of the event: ordered against the
            ev = torch.cuda.Event()                      # cascade that will overwrite the slot (it waits for this event's call)
            ev.record(es)
        self.last_embed = ev
        self.pending.append((ev, emb, vcopy, list(group)))
        group.clear()

    def _collect(self, j):
        while self.nxt in self.done:
            i = self.nxt
            out = self.done.pop(i)
            n = out["valid"].shape[0]
            self.group.append((i, out))
            self.nxt += 1
            short = n < self.ring["n"]
            wraps = (i + 1) % self.R == 0                # the next slot is not adjacent in memory
            if len(self.group) == self.G or short or wraps:
                self._flush_group(j)


def detect_embed_overlapped(engines: Sequence[Engine], batches, on_result: Callable[[int, dict], None] | None = None,
                            streams=None, embed_group: int = 1, embed_engine: Engine | None = None, n_batches: int | None = None,
                            on_detect: Callable[[int, int], None] | None = None) -> List[dict]:
    """:class:`Overlapped` over a known list of batches.  ``batches``: a sequence, or a callable ``(i, j) -> batch`` with
    ``n_batches`` (the bench's NV12 uploader prefetches per engine).  Returns the per-batch results in batch order.  If a call
    fails, the calls still queued on the other engines are finished before the exception propagates."""
    if callable(batches):
        get, K = batches, int(n_batches)
    else:
        seq = list(batches)
        get, K = (lambda i, j: seq[i]), len(seq)
    ov = Overlapped(engines, on_result=on_result, streams=streams, embed_group=embed_group, embed_engine=embed_engine, on_detect=on_detect)
    try:
        for i in range(K):
            ov.push(lambda j, i=i: get(i, j))
        return ov.finish()
    except BaseException:
        ov.abandon()
        raise


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

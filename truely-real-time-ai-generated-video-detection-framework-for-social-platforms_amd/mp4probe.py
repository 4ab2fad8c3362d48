"""ISO-BMFF (.mp4) demultiplexer, header level: what `cv2.VideoCapture` reports to server/model.py:28-30 -- frame size, frame
rate, frame count -- plus the H.264 profile and the byte ranges of every coded video sample, for plain and fragmented
(moof/trun, as yt-dlp delivers the reference's sample clip) files.

It does NOT decode H.264: the build environment has no decoder to check one against, and a from-scratch CABAC/CAVLC decoder
that cannot be verified bit for bit would void the uint8-frame parity contract.  With OpenCV installed
`video_io.open_reader` decodes through it, exactly like the reference; without it, `probe()` still tells the caller what the
clip is (the server logs it) and `samples()` hands the coded access units to whatever decoder the deployment has (a hardware
decoder yields NV12, which `model.run` ingests on the device)."""
from __future__ import annotations

import mmap
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Tuple


@dataclass
class Mp4Info:
    width: int = 0
    height: int = 0
    timescale: int = 0
    duration: int = 0                # in timescale units, over all fragments
    frame_count: int = 0
    codec: str = ""
    profile_idc: int = 0
    level_idc: int = 0
    nal_length_size: int = 4
    sps: List[bytes] = field(default_factory=list)
    pps: List[bytes] = field(default_factory=list)
    fragmented: bool = False
    sample_ranges: List[Tuple[int, int]] = field(default_factory=list)   # (file offset, size) per coded frame, decode order

    @property
    def fps(self) -> float:
        return self.frame_count * self.timescale / self.duration if self.duration else 0.0


MAX_SAMPLES = 10_000_000       # ~92 hours at 30 fps: anything above is a damaged or hostile container


class Mp4Error(ValueError):
    """The container declares something its bytes cannot hold (counts past the end of a box, absurd sample numbers)."""


def _count(buf, off: int, box_end: int, entry_bytes: int, what: str, first_entry: int | None = None) -> int:
    """A 32-bit entry count read from a box, checked against what the box can hold BEFORE anything is allocated or looped over:
    the count fields of stsz / stco / stsc / trun come straight from an upload."""
    if off + 4 > box_end:
        raise Mp4Error(f"{what}: truncated box")
    (n,) = struct.unpack_from(">I", buf, off)
    start = off + 4 if first_entry is None else first_entry
    if n > MAX_SAMPLES or (entry_bytes and n * entry_bytes > max(0, box_end - start)):
        raise Mp4Error(f"{what}: {n} entries do not fit the box")
    return n


def _boxes(buf: bytes, off: int, end: int):
    while off + 8 <= end:
        size, typ = struct.unpack_from(">I4s", buf, off)
        hdr = 8
        if size == 1:
            (size,) = struct.unpack_from(">Q", buf, off + 8)
            hdr = 16
        elif size == 0:
            size = end - off
        if size < hdr or off + size > end:
            return
        yield typ, off + hdr, off + size, off
        off += size


def _find(buf, off, end, *path):
    for typ, a, b, _ in _boxes(buf, off, end):
        if typ == path[0]:
            return (a, b) if len(path) == 1 else _find(buf, a, b, *path[1:])
    return None


def probe(path: str) -> Optional[Mp4Info]:
    """Parse the container; None if the file is not an ISO-BMFF file with an H.264 video track."""
    with open(path, "rb") as f:
        try:
            buf = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)      # never the whole upload in memory
        except ValueError:                                                # empty file
            return None
    try:
        return _probe(buf)
    finally:
        buf.close()


def _probe(buf) -> Optional[Mp4Info]:
    if len(buf) < 16 or buf[4:8] != b"ftyp":
        return None
    moov = _find(buf, 0, len(buf), b"moov")
    if moov is None:
        return None
    info, track_id, trex = Mp4Info(), None, {}
    stbl = None
    for typ, a, b, _ in _boxes(buf, *moov):
        if typ == b"mvex":
            for t2, a2, b2, _ in _boxes(buf, a, b):
                if t2 == b"trex":       # version/flags, track_ID, default description index, duration, size, flags
                    tid, _di, ddur, dsz, _fl = struct.unpack_from(">IIIII", buf, a2 + 4)
                    trex[tid] = (ddur, dsz)
        if typ != b"trak":
            continue
        hdlr = _find(buf, a, b, b"mdia", b"hdlr")
        if hdlr is None or buf[hdlr[0] + 8:hdlr[0] + 12] != b"vide":
            continue
        tk = _find(buf, a, b, b"tkhd")
        v = buf[tk[0]]
        track_id = struct.unpack_from(">I", buf, tk[0] + (20 if v == 1 else 12))[0]
        md = _find(buf, a, b, b"mdia", b"mdhd")
        if buf[md[0]] == 1:
            info.timescale, info.duration = struct.unpack_from(">IQ", buf, md[0] + 20)
        else:
            info.timescale, info.duration = struct.unpack_from(">II", buf, md[0] + 12)
        stbl = _find(buf, a, b, b"mdia", b"minf", b"stbl")
        stsd = _find(buf, *stbl, b"stsd")
        ent = stsd[0] + 8                                    # first sample entry: size, format, 6 reserved, dref index, ...
        esize, fmt = struct.unpack_from(">I4s", buf, ent)
        info.codec = fmt.decode("latin1")
        info.width, info.height = struct.unpack_from(">HH", buf, ent + 32)
        for t3, a3, b3, _ in _boxes(buf, ent + 86, ent + esize):
            if t3 == b"avcC":
                info.profile_idc, info.level_idc = buf[a3 + 1], buf[a3 + 3]
                info.nal_length_size = (buf[a3 + 4] & 3) + 1
                n, p = buf[a3 + 5] & 31, a3 + 6
                for _ in range(n):
                    (ln,) = struct.unpack_from(">H", buf, p); info.sps.append(buf[p + 2:p + 2 + ln]); p += 2 + ln
                n = buf[p]; p += 1
                for _ in range(n):
                    (ln,) = struct.unpack_from(">H", buf, p); info.pps.append(buf[p + 2:p + 2 + ln]); p += 2 + ln
        break
    if track_id is None or not info.codec.startswith("avc"):
        return None
    # ---- plain file: sample tables ---------------------------------------------------------------------------------
    stsz = _find(buf, *stbl, b"stsz")
    if stsz is None:
        raise Mp4Error("no stsz box")
    (uniform,) = struct.unpack_from(">I", buf, stsz[0] + 4)
    # a uniform sample size needs no table: the count is then bounded by the file (every sample occupies >= 1 byte of it)
    count = _count(buf, stsz[0] + 8, stsz[1], 0 if uniform else 4, "stsz")
    if uniform and count * max(1, uniform) > len(buf):
        raise Mp4Error(f"stsz: {count} samples of {uniform} bytes exceed the file")
    if count:
        sizes = None if uniform else struct.unpack_from(f">{count}I", buf, stsz[0] + 12)
        co = _find(buf, *stbl, b"stco")
        if co is not None:
            n = _count(buf, co[0] + 4, co[1], 4, "stco"); chunks = struct.unpack_from(f">{n}I", buf, co[0] + 8)
        else:
            co = _find(buf, *stbl, b"co64")
            if co is None:
                raise Mp4Error("no chunk offset box")
            n = _count(buf, co[0] + 4, co[1], 8, "co64"); chunks = struct.unpack_from(f">{n}Q", buf, co[0] + 8)
        sc = _find(buf, *stbl, b"stsc")
        if sc is None:
            raise Mp4Error("no stsc box")
        n = _count(buf, sc[0] + 4, sc[1], 12, "stsc")
        runs = [struct.unpack_from(">III", buf, sc[0] + 8 + 12 * i) for i in range(n)]
        if not runs or runs[0][0] != 1:
            raise Mp4Error("stsc: no run for the first chunk")
        s, ri = 0, 0
        for ci, base in enumerate(chunks):
            while ri + 1 < len(runs) and runs[ri + 1][0] <= ci + 1:      # runs are sorted by first_chunk: one forward cursor
                ri += 1
            off = base
            for _ in range(min(runs[ri][1], count - s)):
                sz = uniform if sizes is None else sizes[s]
                info.sample_ranges.append((off, sz)); off += sz; s += 1
            if s >= count:
                break
        info.frame_count = count
        return info
    # ---- fragmented file: moof / traf / trun -------------------------------------------------------------------------
    info.fragmented = True
    total_dur = 0
    for typ, a, b, box_off in _boxes(buf, 0, len(buf)):
        if typ != b"moof":
            continue
        for t2, a2, b2, _ in _boxes(buf, a, b):
            if t2 != b"traf":
                continue
            tfhd = _find(buf, a2, b2, b"tfhd")
            flags = struct.unpack_from(">I", buf, tfhd[0])[0] & 0xFFFFFF
            (tid,) = struct.unpack_from(">I", buf, tfhd[0] + 4)
            if tid != track_id:
                continue
            p = tfhd[0] + 8
            base = box_off                                   # default-base-is-moof / no explicit base: the moof's first byte
            ddur, dsz = trex.get(tid, (0, 0))
            if flags & 0x1: (base,) = struct.unpack_from(">Q", buf, p); p += 8
            if flags & 0x2: p += 4
            if flags & 0x8: (ddur,) = struct.unpack_from(">I", buf, p); p += 4
            if flags & 0x10: (dsz,) = struct.unpack_from(">I", buf, p); p += 4
            for t3, a3, b3, _ in _boxes(buf, a2, b2):
                if t3 != b"trun":
                    continue
                fl = struct.unpack_from(">I", buf, a3)[0] & 0xFFFFFF
                per = 4 * (bool(fl & 0x100) + bool(fl & 0x200) + bool(fl & 0x400) + bool(fl & 0x800))
                q = a3 + 8 + 4 * (bool(fl & 0x1) + bool(fl & 0x4))
                n = _count(buf, a3 + 4, b3, per, "trun", first_entry=q)
                if not (fl & 0x200) and dsz == 0 and n:
                    raise Mp4Error("trun: samples with neither a per-sample nor a default size")
                if per == 0 and n * max(1, dsz) > len(buf):
                    raise Mp4Error(f"trun: {n} samples of {dsz} bytes exceed the file")
                if info.frame_count + n > MAX_SAMPLES:
                    raise Mp4Error("trun: too many samples")
                q = a3 + 8
                off = base
                if fl & 0x1: off = base + struct.unpack_from(">i", buf, q)[0]; q += 4
                if fl & 0x4: q += 4
                for _ in range(n):
                    dur, sz = ddur, dsz
                    if fl & 0x100: (dur,) = struct.unpack_from(">I", buf, q); q += 4
                    if fl & 0x200: (sz,) = struct.unpack_from(">I", buf, q); q += 4
                    if fl & 0x400: q += 4
                    if fl & 0x800: q += 4
                    info.sample_ranges.append((off, sz)); off += sz
                    total_dur += dur
                    info.frame_count += 1
    if total_dur:
        info.duration = total_dur
    return info


def samples(path: str, info: Optional[Mp4Info] = None):
    """Yields the coded access units (length-prefixed NAL units, as stored) of the video track in decode order."""
    info = info or probe(path)
    if info is None:
        return
    with open(path, "rb") as f:
        for off, size in info.sample_ranges:
            f.seek(off)
            yield f.read(size)

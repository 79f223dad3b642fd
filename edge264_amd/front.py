"""ctypes binding of edge264_amd/libedge264_hipfront.so (the reference's parsers + our emitters behind edge264.h:64-70) for
host-side tools: turns an Annex-B stream into the command packets the back end consumes, with the CAPTURE sink (no device,
nothing reconstructed).  bench.py's same-input leg and tools/ use it; the decoder API itself is C (include/edge264_hip.h,
INTEGRATION.md) -- this file is plumbing around it, not a decoder.
"""
from __future__ import annotations

import ctypes as C
import errno
import os
import time

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libedge264_hipfront.so")


class FrontError(RuntimeError):
    pass


_lib = None


def load():
    """The front-end library; raises FrontError when it has not been built (make -C edge264_amd/frontend, needs the reference tree)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FrontError(f"{LIB_PATH} not built")
    L = C.CDLL(LIB_PATH)
    L.edge264_alloc.restype = C.c_void_p
    L.edge264_alloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.edge264_free.argtypes = [C.POINTER(C.c_void_p)]
    L.edge264_find_start_code.restype = C.c_void_p
    L.edge264_find_start_code.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.edge264_decode_NAL.restype = C.c_int
    L.edge264_decode_NAL.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.edge264_get_frame.restype = C.c_int
    L.edge264_get_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.e264front_set_sink.argtypes = [C.c_int]
    L.e264front_set_compact.argtypes = [C.c_int]
    L.e264front_take_packet.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.e264front_take_packet.restype = C.c_int
    L.e264front_free_packet.argtypes = [C.c_void_p]
    _lib = L
    return L


def capture_packets(stream: bytes, n_threads: int = 0, compact: bool = False) -> tuple[list[bytes], int, float]:
    """Parses `stream` (Annex B) with the capture sink.  Returns (command packets in decoding order, output frames the
    decoder handed out, seconds of host time spent parsing + emitting).  compact: pictures with inter macroblocks come in the
    wire form (version 5, include/edge264_compact.h)."""
    L = load()
    L.e264front_set_sink(1)
    L.e264front_set_compact(1 if compact else 0)
    buf = C.create_string_buffer(stream + b"\0" * 64, len(stream) + 64)
    base = C.addressof(buf)
    end = base + len(stream)
    dec = C.c_void_p(L.edge264_alloc(n_threads, None, None, 0, None, None, None))
    if not dec:
        raise FrontError("edge264_alloc failed")
    out = (C.c_uint8 * 512)()  # Edge264Frame
    packets: list[bytes] = []
    frames = 0
    spent = 0.0
    data, n = C.c_void_p(), C.c_size_t()

    def pump():
        nonlocal frames
        while L.e264front_take_packet(dec, C.byref(data), C.byref(n)) == 0:
            packets.append(C.string_at(data, n.value))
            L.e264front_free_packet(data)
        while L.edge264_get_frame(dec, out, 0) == 0:
            frames += 1

    nal = L.edge264_find_start_code(base, end, 0)
    nal = (nal or end) + 3 if (nal or end) < end else end
    while True:
        nxt = L.edge264_find_start_code(nal, end, 0) if nal < end else end
        t0 = time.perf_counter()
        res = L.edge264_decode_NAL(dec, nal, nxt, None, None)
        spent += time.perf_counter() - t0
        f0 = frames
        pump()
        if res == errno.ENOBUFS:
            if frames == f0:
                break
            continue
        if res == errno.ENODATA or nal >= end:
            break
        nal = min(nxt + 3, end)
    pump()
    L.edge264_free(C.byref(dec))
    return packets, frames, spent


def decode_timed(stream: bytes, sink: int = 0, compact: bool = False) -> dict:
    """ONE decoder through the edge264.h API on the device sink (edge264_decode_NAL / edge264_get_frame, frames downloaded to the host mirror as an
    application sees them): wall time per output picture.  bench.py's single-stream latency leg -- the figure the reference's own `edge264_test -b`
    prints for itself (/root/reference/src/edge264_test.c:522-542), per picture."""
    L = load()
    L.e264front_set_sink(sink)
    L.e264front_set_compact(1 if compact else 0)
    buf = C.create_string_buffer(stream + b"\0" * 64, len(stream) + 64)
    base = C.addressof(buf)
    end = base + len(stream)
    t_alloc = time.perf_counter()
    dec = C.c_void_p(L.edge264_alloc(0, None, None, 0, None, None, None))
    if not dec:
        raise FrontError("edge264_alloc failed")
    out = (C.c_uint8 * 512)()  # Edge264Frame
    stamps = []
    t0 = time.perf_counter()
    nal = L.edge264_find_start_code(base, end, 0)
    nal = (nal or end) + 3 if (nal or end) < end else end
    while True:
        nxt = L.edge264_find_start_code(nal, end, 0) if nal < end else end
        res = L.edge264_decode_NAL(dec, nal, nxt, None, None)
        f0 = len(stamps)
        while L.edge264_get_frame(dec, out, 0) == 0:
            stamps.append(time.perf_counter())
        if res == errno.ENOBUFS:
            if len(stamps) == f0:
                break
            continue
        if res == errno.ENODATA or nal >= end:
            break
        nal = min(nxt + 3, end)
    while L.edge264_get_frame(dec, out, 0) == 0:
        stamps.append(time.perf_counter())
    t1 = time.perf_counter()
    L.edge264_free(C.byref(dec))
    return {"pictures": len(stamps), "seconds": t1 - t0, "alloc_seconds": t0 - t_alloc, "first_picture_seconds": (stamps[0] - t0) if stamps else None}

"""ctypes binding of the C-ABI back end (include/edge264_hip.h, libedge264_hip.so).

This is plumbing around the product, not the product: the reconstruction itself is the
hand-written gfx950 HIP code in edge264_amd/csrc.  There is NO CPU fallback here: if the
shared library is missing or no MI355X is present, opening a device raises.
"""
from __future__ import annotations

import ctypes as C
import errno
import os

import numpy as np

from . import packet as P

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("E264_HIP_LIB") or os.path.join(HERE, "libedge264_hip.so")  # E264_HIP_LIB: A/B builds (csrc/Makefile `variant`)
RUN_RECON, RUN_DEBLOCK, RUN_ALL = 1, 2, 3
MAX_LANES = 4

_lib = None


class BackendError(RuntimeError):
    pass


def load_library():
    """Loads libedge264_hip.so (in-tree, built by __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP back end has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    sig = {
        "e264hip_device_open": (i, [i, C.POINTER(vp)]),
        "e264hip_device_close": (None, [vp]),
        "e264hip_device_sync": (i, [vp]),
        "e264hip_last_error": (C.c_char_p, []),
        "e264hip_stream_open": (i, [vp, C.POINTER(vp)]),
        "e264hip_stream_close": (None, [vp]),
        "e264hip_stream_bind_lane": (i, [vp, i]),
        "e264hip_stream_flush": (i, [vp]),
        "e264hip_frame_alloc": (i, [vp, i, sz, C.POINTER(vp)]),
        "e264hip_frame_free": (None, [vp, i]),
        "e264hip_frame_fill": (i, [vp, i, i]),
        "e264hip_frame_upload": (i, [vp, i, vp, sz]),
        "e264hip_frame_submit": (i, [vp, vp, sz]),
        "e264hip_packet_buffer": (vp, [vp, sz]),
        "e264hip_frame_wait": (i, [vp, i]),
        "e264hip_frame_download": (i, [vp, i, vp, sz]),
        "e264hip_packet_upload": (i, [vp, vp, sz, C.POINTER(vp)]),
        "e264hip_packet_free": (None, [vp]),
        "e264hip_submit_batch": (i, [vp, C.POINTER(vp), C.POINTER(vp), i, i]),
        "e264hip_packet_check": (i, [vp, sz]),
        "e264hip_packet_compact_bound": (sz, [vp, sz]),
        "e264hip_packet_compact": (sz, [vp, sz, vp, sz]),
        "e264hip_packet_expand": (sz, [vp, sz, vp, sz]),
        "e264hip_submit_batch_host": (i, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), i, i]),
        "e264hip_submit_batch_pinned": (i, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), i, i, i]),
        "e264hip_host_alloc": (vp, [vp, sz]),
        "e264hip_host_free": (None, [vp, vp]),
        "e264hip_batch_create": (i, [vp, C.POINTER(vp), C.POINTER(vp), i, C.POINTER(vp)]),
        "e264hip_batch_submit": (i, [vp, i]),
        "e264hip_batch_free": (None, [vp]),
        "e264hip_event_record": (i, [vp, i]),
        "e264hip_event_elapsed_ms": (i, [vp, i, i, C.POINTER(C.c_float)]),
        "e264hip_event_query": (i, [vp, i]),
        "e264hip_kernel_timing": (i, [vp, i]),
        "e264hip_kernel_time_ms": (i, [vp, C.POINTER(C.c_double), C.POINTER(i)]),
        "e264hip_set_option": (i, [vp, C.c_char_p, i]),
        "e264hip_build_flags": (C.c_char_p, []),
        "e264hip_frame_device_ptr": (vp, [vp, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the library does not export the header's symbol
        fn.restype, fn.argtypes = res, args
    flags = (L.e264hip_build_flags() or b"").decode()
    if ("E264_ABL_" in flags or "E264_PHASE_" in flags) and os.environ.get("E264_ALLOW_ABLATION") != "1":
        raise BackendError(f"{LIB_PATH} is a timing-ablation build ({flags.strip()}): its samples are wrong by design; "
                           "set E264_ALLOW_ABLATION=1 to load it (tools/visits/gpu_ab.sh does)")
    _lib = L
    return L


def build_flags() -> str:
    """Build-time switches of the loaded library ("" = product build)."""
    return (load_library().e264hip_build_flags() or b"").decode().strip()


EXPORTED_SYMBOLS = [
    "e264hip_device_open", "e264hip_device_close", "e264hip_device_sync", "e264hip_last_error",
    "e264hip_stream_open", "e264hip_stream_close", "e264hip_stream_bind_lane", "e264hip_stream_flush", "e264hip_frame_alloc",
    "e264hip_frame_free", "e264hip_frame_fill", "e264hip_frame_upload", "e264hip_frame_submit",
    "e264hip_packet_buffer", "e264hip_frame_wait", "e264hip_frame_download", "e264hip_packet_upload",
    "e264hip_packet_free", "e264hip_packet_check", "e264hip_packet_compact_bound", "e264hip_packet_compact", "e264hip_packet_expand", "e264hip_submit_batch", "e264hip_submit_batch_host", "e264hip_submit_batch_pinned", "e264hip_host_alloc", "e264hip_host_free", "e264hip_batch_create", "e264hip_batch_submit", "e264hip_batch_free", "e264hip_event_record", "e264hip_event_elapsed_ms", "e264hip_event_query",
    "e264hip_kernel_timing", "e264hip_kernel_time_ms", "e264hip_set_option", "e264hip_build_flags", "e264hip_frame_device_ptr",
]


def _check(L, r: int, what: str) -> None:
    if r:
        msg = L.e264hip_last_error()
        raise BackendError(f"{what}: {errno.errorcode.get(r, r)} ({(msg or b'').decode()})")


def packet_check(pkt: bytes) -> int:
    """Host-only validation (no GPU needed): 0 or an errno value."""
    L = load_library()
    buf = (C.c_char * len(pkt)).from_buffer_copy(pkt)
    return L.e264hip_packet_check(C.cast(buf, C.c_void_p), len(pkt))


def packet_compact(pkt: bytes) -> bytes:
    """A version-4 packet in its WIRE form (version 5, include/edge264_compact.h).  Host only."""
    L = load_library()
    src = (C.c_char * len(pkt)).from_buffer_copy(pkt)
    cap = L.e264hip_packet_compact_bound(C.cast(src, C.c_void_p), len(pkt))
    out = (C.c_char * max(cap, 1))()
    n = L.e264hip_packet_compact(C.cast(src, C.c_void_p), len(pkt), C.cast(out, C.c_void_p), cap) if cap else 0
    if not n:
        raise BackendError(f"packet_compact: {last_error()}")
    return bytes(out[:n])


def packet_expand(pkt: bytes) -> bytes:
    """A wire packet as the canonical version-4 packet it stands for.  Host only."""
    L = load_library()
    src = (C.c_char * len(pkt)).from_buffer_copy(pkt)
    cap = L.e264hip_packet_expand(C.cast(src, C.c_void_p), len(pkt), None, 0)
    if not cap:
        raise BackendError(f"packet_expand: {last_error()}")
    out = (C.c_char * cap)()
    n = L.e264hip_packet_expand(C.cast(src, C.c_void_p), len(pkt), C.cast(out, C.c_void_p), cap)
    if not n:
        raise BackendError(f"packet_expand: {last_error()}")
    return bytes(out[:n])


def last_error() -> str:
    msg = load_library().e264hip_last_error()
    return msg.decode() if msg else ""


class Device:
    """One MI355X: queue + kernels (E264Device)."""

    def __init__(self, ordinal: int = 0):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L, self.L.e264hip_device_open(ordinal, C.byref(h)), "e264hip_device_open")
        self.h = h
        self.ordinal = ordinal

    def close(self):
        if self.h:
            self.L.e264hip_device_close(self.h)
            self.h = None

    def sync(self):
        _check(self.L, self.L.e264hip_device_sync(self.h), "device_sync")

    def set_option(self, name: str, value: int) -> int:
        return self.L.e264hip_set_option(self.h, name.encode(), value)

    def upload_packet(self, pkt: bytes) -> "DevicePacket":
        return DevicePacket(self, pkt)

    def submit_batch(self, streams, packets, mode: int = RUN_ALL) -> None:
        n = len(streams)
        sa = (C.c_void_p * n)(*[s.h for s in streams])
        pa = (C.c_void_p * n)(*[p.h for p in packets])
        _check(self.L, self.L.e264hip_submit_batch(self.h, sa, pa, n, mode), "submit_batch")

    def submit_batch_host(self, streams, packets, mode: int = RUN_ALL) -> None:
        """packets: bytes objects still in host memory (asynchronous: staged, copied and launched on the queue)."""
        n = len(streams)
        sa = (C.c_void_p * n)(*[s.h for s in streams])
        bufs = [(C.c_char * len(p)).from_buffer_copy(p) for p in packets]
        pa = (C.c_void_p * n)(*[C.cast(b, C.c_void_p) for b in bufs])
        za = (C.c_size_t * n)(*[len(p) for p in packets])
        _check(self.L, self.L.e264hip_submit_batch_host(self.h, sa, pa, za, n, mode), "submit_batch_host")

    def prepare_host_batch(self, streams, packets):
        """The argument arrays of submit_batch_host built once (benchmarks re-submit the same host packets)."""
        n = len(streams)
        bufs = [(C.c_char * len(p)).from_buffer_copy(p) for p in packets]
        return ((C.c_void_p * n)(*[s.h for s in streams]), (C.c_void_p * n)(*[C.cast(b, C.c_void_p) for b in bufs]),
                (C.c_size_t * n)(*[len(p) for p in packets]), n, bufs)

    def pinned_copy(self, pkt: bytes):
        """The packet's bytes in page-locked memory of this device (what a front end's emitter writes into directly)."""
        p = self.L.e264hip_host_alloc(self.h, len(pkt))
        if not p:
            raise BackendError("e264hip_host_alloc: " + last_error())
        C.memmove(p, pkt, len(pkt))
        return p

    def pinned_free(self, p) -> None:
        self.L.e264hip_host_free(self.h, p)

    def prepare_pinned_batch(self, streams, pinned_ptrs, sizes):
        n = len(streams)
        return ((C.c_void_p * n)(*[s.h for s in streams]), (C.c_void_p * n)(*pinned_ptrs), (C.c_size_t * n)(*sizes), n)

    def submit_pinned_prepared(self, pb, mode: int = RUN_ALL, trusted: bool = True) -> None:
        _check(self.L, self.L.e264hip_submit_batch_pinned(self.h, pb[0], pb[1], pb[2], pb[3], mode, 1 if trusted else 0), "submit_batch_pinned")

    def submit_host_prepared(self, hb, mode: int = RUN_ALL) -> None:
        _check(self.L, self.L.e264hip_submit_batch_host(self.h, hb[0], hb[1], hb[2], hb[3], mode), "submit_batch_host")

    def make_batch(self, streams, packets):
        """Device-resident job table for repeated launches (E264Batch)."""
        n = len(streams)
        sa = (C.c_void_p * n)(*[s.h for s in streams])
        pa = (C.c_void_p * n)(*[p.h for p in packets])
        b = C.c_void_p()
        _check(self.L, self.L.e264hip_batch_create(self.h, sa, pa, n, C.byref(b)), "batch_create")
        return b

    def submit_prepared(self, batch, mode: int = RUN_ALL) -> None:
        _check(self.L, self.L.e264hip_batch_submit(batch, mode), "batch_submit")

    def free_batch(self, batch) -> None:
        self.L.e264hip_batch_free(batch)

    def event_record(self, idx: int) -> None:
        _check(self.L, self.L.e264hip_event_record(self.h, idx), "event_record")

    def event_elapsed_ms(self, a: int, b: int) -> float:
        ms = C.c_float()
        _check(self.L, self.L.e264hip_event_elapsed_ms(self.h, a, b, C.byref(ms)), "event_elapsed")
        return float(ms.value)

    def event_done(self, idx: int) -> bool:
        """Has everything queued before event_record(idx) left the GPU?  Never blocks."""
        r = self.L.e264hip_event_query(self.h, idx)
        if r not in (0, errno.EBUSY):
            _check(self.L, r, "event_query")
        return r == 0

    def kernel_timing(self, enable: bool) -> None:
        _check(self.L, self.L.e264hip_kernel_timing(self.h, int(enable)), "kernel_timing")

    def kernel_time_ms(self):
        t, n = (C.c_double * 4)(), C.c_int()
        _check(self.L, self.L.e264hip_kernel_time_ms(self.h, t, C.byref(n)), "kernel_time")
        return [float(x) for x in t], int(n.value)


class DevicePacket:
    def __init__(self, dev: Device, pkt: bytes):
        self.dev, self.nbytes = dev, len(pkt)
        h = C.c_void_p()
        buf = (C.c_char * len(pkt)).from_buffer_copy(pkt)
        _check(dev.L, dev.L.e264hip_packet_upload(dev.h, C.cast(buf, C.c_void_p), len(pkt), C.byref(h)), "packet_upload")
        self.h = h

    def free(self):
        if self.h:
            self.dev.L.e264hip_packet_free(self.h)
            self.h = None


class Stream:
    """Device side of one decoder (E264Stream): DPB slots in HBM."""

    def __init__(self, dev: Device, width_mbs: int, height_mbs: int):
        self.dev, self.L = dev, dev.L
        self.w, self.h_mbs = width_mbs, height_mbs
        self.frame_bytes = P.frame_bytes(width_mbs, height_mbs)
        h = C.c_void_p()
        _check(self.L, self.L.e264hip_stream_open(dev.h, C.byref(h)), "stream_open")
        self.h = h

    def close(self):
        if self.h:
            self.L.e264hip_stream_close(self.h)
            self.h = None

    def bind_lane(self, lane: int) -> None:
        """Compute lane (HIP queue) of the device this stream's work is ordered on; streams of different lanes overlap."""
        _check(self.L, self.L.e264hip_stream_bind_lane(self.h, lane), "stream_bind_lane")

    def alloc(self, slot: int, mirror: bool = False) -> None:
        m = C.c_void_p()
        _check(self.L, self.L.e264hip_frame_alloc(self.h, slot, self.frame_bytes, C.byref(m) if mirror else None), "frame_alloc")

    def free(self, slot: int) -> None:
        self.L.e264hip_frame_free(self.h, slot)

    def fill(self, slot: int, value: int) -> None:
        _check(self.L, self.L.e264hip_frame_fill(self.h, slot, value), "frame_fill")

    def upload(self, slot: int, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data, np.uint8)
        _check(self.L, self.L.e264hip_frame_upload(self.h, slot, data.ctypes.data, min(data.nbytes, self.frame_bytes)), "frame_upload")

    def submit(self, pkt: bytes) -> None:
        buf = (C.c_char * len(pkt)).from_buffer_copy(pkt)
        _check(self.L, self.L.e264hip_frame_submit(self.h, C.cast(buf, C.c_void_p), len(pkt)), "frame_submit")

    def download(self, slot: int) -> np.ndarray:
        out = np.empty(self.frame_bytes, np.uint8)
        _check(self.L, self.L.e264hip_frame_download(self.h, slot, out.ctypes.data, out.nbytes), "frame_download")
        return out

    def flush(self) -> None:
        _check(self.L, self.L.e264hip_stream_flush(self.h), "stream_flush")

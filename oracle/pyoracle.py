"""ctypes bindings of the CHECKERS (test infrastructure, never the product):

  Oracle       oracle/libe264_oracle.so        scalar restatement (oracle/e264_oracle.c)
  RefKernels   oracle/_ref/libe264_refkernels.so  the reference's own static kernels
  RefDecoder   oracle/_ref/libedge264_ref.so   the unmodified reference decoder (edge264.h API)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_SLOTS = 32


def build(quiet: bool = True) -> None:
    """make -C oracle (compiles the restatement; the reference-derived libraries only
    where /root/reference exists -- building the checker is not using it)."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load(path: str):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return C.CDLL(path)


def _dpb_array(dpb):
    arr = (C.c_void_p * MAX_SLOTS)()
    for i in range(MAX_SLOTS):
        b = dpb[i] if i < len(dpb) else None
        arr[i] = b.ctypes.data if b is not None else None
    return arr


class Oracle:
    def __init__(self):
        path = os.path.join(HERE, "libe264_oracle.so")
        if not os.path.exists(path):
            build()
        self.lib = _load(path)
        self.lib.e264_oracle_decode_frame.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int]
        self.lib.e264_oracle_decode_frame.restype = C.c_int
        self.lib.e264_oracle_frame_bs.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        self.lib.e264_oracle_frame_bs.restype = C.c_int

    def decode_frame(self, pkt: bytes, dpb: list, passes: int = 3) -> None:
        """dpb: list of 32 uint8 numpy arrays (or None); the destination slot is written in place."""
        r = self.lib.e264_oracle_decode_frame(pkt, len(pkt), _dpb_array(dpb), passes)
        if r:
            raise RuntimeError(f"oracle rejected packet ({r})")

    def frame_bs(self, pkt: bytes, n_mbs: int) -> np.ndarray:
        out = np.zeros((n_mbs, 2, 4, 4), np.uint8)
        r = self.lib.e264_oracle_frame_bs(pkt, len(pkt), out.ctypes.data)
        if r:
            raise RuntimeError(f"oracle rejected packet ({r})")
        return out


class RefKernels:
    """The reference's static kernels replaying a packet (oracle/ref_kernels_harness.c)."""

    def __init__(self):
        self.lib = _load(os.path.join(HERE, "_ref", "libe264_refkernels.so"))
        L = self.lib
        L.ref_new.argtypes = [C.c_int, C.c_int]
        L.ref_new.restype = C.c_void_p
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_replay_packet.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_int]
        L.ref_replay_packet.restype = C.c_int
        for n in ("ref_intra4x4", "ref_intra8x8", "ref_intra16x16", "ref_intra_chroma"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.ref_inter_luma.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ref_inter_chroma.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p] + [C.c_int] * 4
        self._h = {}

    def replay(self, pkt: bytes, dpb: list, width_mbs: int, height_mbs: int, passes: int = 3) -> None:
        key = (width_mbs, height_mbs)
        if key not in self._h:
            self._h[key] = self.lib.ref_new(width_mbs, height_mbs)
        r = self.lib.ref_replay_packet(self._h[key], pkt, len(pkt), _dpb_array(dpb), passes)
        if r:
            raise RuntimeError(f"reference harness rejected packet ({r})")


LAZY_DRAIN = False  # tools/*_sweep.py --lazy: frames are only fetched when edge264_decode_NAL answers ENOBUFS, and at the end of the stream


class Edge264Frame(C.Structure):
    # edge264.h:45-62
    _fields_ = [("samples", C.c_void_p * 3), ("samples_mvc", C.c_void_p * 3), ("mb_errors", C.c_void_p),
                ("bit_depth_Y", C.c_int8), ("bit_depth_C", C.c_int8),
                ("width_Y", C.c_int16), ("width_C", C.c_int16), ("height_Y", C.c_int16), ("height_C", C.c_int16),
                ("stride_Y", C.c_int16), ("stride_C", C.c_int16), ("stride_mb", C.c_int16),
                ("FrameId", C.c_int32), ("FrameId_mvc", C.c_int32), ("frame_crop_offsets", C.c_int16 * 4),
                ("return_arg", C.c_void_p)]


class Edge264Lib:
    """Any shared object exporting the edge264.h surface (the unmodified reference, or
    the reference front end bound to our back end).  Mirrors README.md:126-155 usage."""

    def __init__(self, path: str):
        self.lib = L = _load(path)
        L.edge264_alloc.restype = C.c_void_p
        L.edge264_alloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.edge264_free.argtypes = [C.POINTER(C.c_void_p)]
        L.edge264_flush.argtypes = [C.c_void_p]
        L.edge264_find_start_code.restype = C.c_void_p
        L.edge264_find_start_code.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.edge264_decode_NAL.restype = C.c_int
        L.edge264_decode_NAL.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.edge264_get_frame.restype = C.c_int
        L.edge264_get_frame.argtypes = [C.c_void_p, C.POINTER(Edge264Frame), C.c_int]
        L.edge264_return_frame.argtypes = [C.c_void_p, C.c_void_p]

    LOG_CB = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_void_p)  # edge264.h:36 Edge264LogCb

    @staticmethod
    def _log_args(log):
        """log: None, or a list that receives every string the decoder hands to its log callback (edge264.h:36-41, 65)"""
        if log is None:
            return None, None
        cb = Edge264Lib.LOG_CB(lambda s, _arg: (log.append(s.decode("latin-1")), 0)[1])
        return cb, C.cast(cb, C.c_void_p)

    def decode(self, stream: bytes, max_frames: int = 1 << 30, crop: bool = True, n_threads: int = 0, allocator=None, log=None):
        """Decodes an Annex-B byte stream; returns (list of (Y,Cb,Cr) arrays, list of NAL return codes).
        allocator: a CallerAllocator (edge264.h:42-43 alloc_cb / free_cb pair)."""
        import errno
        L = self.lib
        buf = np.frombuffer(stream + b"\0" * 64, np.uint8).copy()
        base = buf.ctypes.data
        end = base + len(stream)
        a = allocator.args() if allocator else (None, None, None)
        keep, log_cb = self._log_args(log)
        dec = C.c_void_p(L.edge264_alloc(n_threads, log_cb, None, 0, *a))
        if not dec:
            raise MemoryError("edge264_alloc")
        frames, codes = [], []
        out = Edge264Frame()

        def drain():
            while len(frames) < max_frames and L.edge264_get_frame(dec, C.byref(out), 0) == 0:
                frames.append(self._copy_frame(out))
        nal = L.edge264_find_start_code(base, end, 0)
        nal = (nal or end) + 3 if (nal or end) < end else end
        while True:
            nxt = L.edge264_find_start_code(nal, end, 0) if nal < end else end
            res = L.edge264_decode_NAL(dec, nal, nxt, None, None)
            codes.append(res)
            if res == errno.ENOBUFS:
                n0 = len(frames)
                drain()
                if len(frames) == n0:  # nothing can be output: the stream is stuck (reference behaviour), give up
                    break
                continue
            if not LAZY_DRAIN:  # (LAZY_DRAIN: an application that only fetches frames when the decoder asks for room, edge264.h ENOBUFS)
                drain()
            if res == errno.ENODATA or nal >= end:
                break
            nal = min(nxt + 3, end)
        drain()
        L.edge264_free(C.byref(dec))
        return frames, codes

    @staticmethod
    def _copy_frame(f: Edge264Frame):
        def plane(ptr, w, h, stride):
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), ((h - 1) * stride + w,))
            return np.lib.stride_tricks.as_strided(a, (h, w), (stride, 1)).copy()
        views = [f.samples] + ([f.samples_mvc] if f.samples_mvc[0] else [])   # MVC: second view appended (edge264.h:46-47)
        return tuple(pl for v in views for pl in (plane(v[0], f.width_Y, f.height_Y, f.stride_Y),
                                                  plane(v[1], f.width_C, f.height_C, f.stride_C),
                                                  plane(v[2], f.width_C, f.height_C, f.stride_C)))


class CallerAllocator:
    """An application-side Edge264AllocCb / Edge264FreeCb pair (edge264.h:42-43) for tests: plain libc memory, every block
    remembered so that a test can check where the decoder delivered its frames and that everything was given back."""
    ALLOC = C.CFUNCTYPE(None, C.POINTER(C.c_void_p), C.c_uint, C.POINTER(C.c_void_p), C.c_uint, C.c_int, C.c_void_p)
    FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)

    def __init__(self):
        self.libc = C.CDLL(None)
        self.libc.malloc.restype = C.c_void_p
        self.libc.malloc.argtypes = [C.c_size_t]
        self.libc.free.argtypes = [C.c_void_p]
        self.live, self.allocs, self.frees = {}, 0, 0   # samples address -> (size, mbs address)

        def alloc(samples, samples_size, mbs, mbs_size, errno_on_fail, arg):
            sp, mp = self.libc.malloc(samples_size + 64), self.libc.malloc(mbs_size + 64)
            samples[0], mbs[0] = sp, mp
            self.live[sp] = (samples_size, mp)
            self.allocs += 1

        def free(samples, mbs, arg):
            size, mp = self.live.pop(samples)
            assert mp == mbs, "free_cb must get back the mbs block alloc_cb returned with these samples"
            self.libc.free(samples)
            self.libc.free(mbs)
            self.frees += 1
        self._a, self._f = self.ALLOC(alloc), self.FREE(free)

    def args(self):
        return C.cast(self._a, C.c_void_p), C.cast(self._f, C.c_void_p), None

    def owns(self, address: int) -> bool:
        return any(sp <= address < sp + size for sp, (size, _) in self.live.items())


def ref_decoder() -> Edge264Lib:
    return Edge264Lib(os.path.join(HERE, "_ref", "libedge264_ref.so"))


def _u8p(a: np.ndarray, off: int = 0):
    return C.c_void_p(a.ctypes.data + off)


def oracle_intra(o: Oracle, kind: str, buf: np.ndarray, off: int, stride: int, mode: int) -> None:
    """kind in {'4x4','8x8','16x16','chroma'}; predicts in place at buf[off] (one plane for chroma)."""
    fn = {"4x4": o.lib.e264o_intra4x4, "8x8": o.lib.e264o_intra8x8, "16x16": o.lib.e264o_intra16x16,
          "chroma": o.lib.e264o_intra_chroma}[kind]
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int]
    fn(_u8p(buf, off), stride, mode)


def oracle_luma_mc(o: Oracle, ref: np.ndarray, stride: int, w: int, h: int, x: int, y: int, mvx: int, mvy: int,
                   bw: int, bh: int) -> np.ndarray:
    o.lib.e264o_luma_mc.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.c_void_p, C.c_int]
    dst = np.zeros((bh, bw), np.uint8)
    o.lib.e264o_luma_mc(_u8p(ref), stride, w, h, x, y, mvx, mvy, bw, bh, _u8p(dst), bw)
    return dst


def oracle_chroma_mc(o: Oracle, ref: np.ndarray, stride: int, w: int, h: int, x: int, y: int, mvx: int, mvy: int,
                     bw: int, bh: int) -> np.ndarray:
    o.lib.e264o_chroma_mc.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.c_void_p, C.c_int]
    dst = np.zeros((bh, bw), np.uint8)
    o.lib.e264o_chroma_mc(_u8p(ref), stride, w, h, x, y, mvx, mvy, bw, bh, _u8p(dst), bw)
    return dst


class HipFront(Edge264Lib):
    """edge264_amd/libedge264_hipfront.so (the PRODUCT's front-end library, built by edge264_amd/frontend/Makefile): the reference's front end (parsers, DPB, reference lists,
    compiled from /root/reference) bound to edge264_amd/frontend's packet emitters behind the edge264.h
    API.  sink 1 = capture (packets handed back, replayed here by the oracle: CPU test of the boundary);
    sink 0 = libedge264_hip.so (GPU)."""

    def __init__(self, path: str | None = None):
        super().__init__(path or os.path.join(os.path.dirname(HERE), "edge264_amd", "libedge264_hipfront.so"))
        L = self.lib
        L.e264front_set_sink.argtypes = [C.c_int]
        L.e264front_take_packet.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.e264front_take_packet.restype = C.c_int
        L.e264front_free_packet.argtypes = [C.c_void_p]
        L.e264front_slot_of.argtypes = [C.c_void_p, C.c_void_p]
        L.e264front_slot_of.restype = C.c_int

    def decode_capture(self, stream: bytes, oracle: "Oracle", n_threads: int = 0, allocator=None, log=None):
        """Returns (frames, codes, packets): frames as the HIP sink would return them, with the
        oracle standing in for the GPU (same slot bookkeeping as edge264_get_frame in the shim)."""
        import errno
        from edge264_amd import packet as P
        L = self.lib
        L.e264front_set_sink(1)
        buf = np.frombuffer(stream + b"\0" * 64, np.uint8).copy()
        base = buf.ctypes.data
        end = base + len(stream)
        a = allocator.args() if allocator else (None, None, None)
        keep, log_cb = self._log_args(log)
        dec = C.c_void_p(L.edge264_alloc(n_threads, log_cb, None, 0, *a))
        if not dec:
            raise MemoryError("edge264_alloc")
        frames, codes, packets = [], [], []
        dpb = [None] * 32
        out = Edge264Frame()
        geom = {}

        def pump(want_frames=True):
            data, n = C.c_void_p(), C.c_size_t()
            while L.e264front_take_packet(dec, C.byref(data), C.byref(n)) == 0:
                pkt = C.string_at(data, n.value)
                L.e264front_free_packet(data)
                packets.append(pkt)
                pk = P.Packet(pkt)
                h = pk.hdr
                nb = int(h["plane_size_Y"]) + int(h["plane_size_C"])
                geom.update(stride_Y=int(h["stride_Y"]), stride_C=int(h["stride_C"]), psY=int(h["plane_size_Y"]), psC=int(h["plane_size_C"]))
                for s in range(32):
                    if dpb[s] is None and (s == int(h["dst_slot"]) or int(h["ref_slots"]) >> s & 1):
                        dpb[s] = np.zeros(nb + 64, np.uint8)  # like the HIP sink (frame_fill 0) and a fresh mmap in the reference
                oracle.decode_frame(pkt, dpb, 3)
            while want_frames and L.edge264_get_frame(dec, C.byref(out), 0) == 0:
                planes = []
                for view in ([out.samples] + ([out.samples_mvc] if out.samples_mvc[0] else [])):   # MVC: second view
                    slot = L.e264front_slot_of(dec, view[0])
                    assert slot >= 0 and (out.return_arg or 0) >> slot & 1
                    d, sy, sc = dpb[slot], geom["stride_Y"], geom["stride_C"]
                    top, _right, _bottom, left = (int(v) for v in out.frame_crop_offsets)  # edge264.h:60: the cropped window inside the coded frame
                    y = d[:geom["psY"]].reshape(-1, sy)[top:top + out.height_Y, left:left + out.width_Y].copy()
                    c = d[geom["psY"]:geom["psY"] + geom["psC"]].reshape(-1, sc)[top >> 1:(top >> 1) + out.height_C]
                    planes += [c_ for c_ in (y, c[:, left >> 1:(left >> 1) + out.width_C].copy(),
                                             c[:, sc // 2 + (left >> 1):sc // 2 + (left >> 1) + out.width_C].copy())]
                frames.append(tuple(planes))

        nal = L.edge264_find_start_code(base, end, 0)
        nal = (nal or end) + 3 if (nal or end) < end else end
        while True:
            nxt = L.edge264_find_start_code(nal, end, 0) if nal < end else end
            res = L.edge264_decode_NAL(dec, nal, nxt, None, None)
            codes.append(res)
            n0 = len(frames)
            pump(want_frames=not LAZY_DRAIN or res == errno.ENOBUFS)
            if res == errno.ENOBUFS:
                if len(frames) == n0:
                    break
                continue
            if res == errno.ENODATA or nal >= end:
                break
            nal = min(nxt + 3, end)
        pump()
        L.edge264_free(C.byref(dec))
        return frames, codes, packets

"""Host-side view of the command packet (include/edge264_cmd.h) as numpy dtypes.

The packet is the drop-in boundary between the edge264 C front end and the
MI355X back end: one packet per coded frame, carrying what the reference's
leaf functions read (reference citations in include/edge264_cmd.h).  This
module only builds / parses bytes; it performs no reconstruction.
"""
from __future__ import annotations

import numpy as np

E264_MAGIC = 0x34363245
E264_VERSION = 4
E264_VERSION_COMPACT = 5  # the wire form, include/edge264_compact.h
MAX_SLOTS = 32

MB_ABSENT, MB_I4x4, MB_I8x8, MB_I16x16, MB_PCM, MB_INTER = range(6)
MBF_T8x8, MBF_EDGE_LEFT, MBF_EDGE_TOP, MBF_DEBLOCK, MBF_DONE, MBF_LEV8 = 1, 2, 4, 8, 16, 32
CODED_LUMA_DC = 1 << 24
CODED_CHROMA_DC = 1 << 25

# zig-zag 4x4 block index -> pixel offset inside the macroblock
# (reference: src/edge264_internal.h:550-553 x444/y444)
BX = np.array([0, 4, 0, 4, 8, 12, 8, 12, 0, 4, 0, 4, 8, 12, 8, 12])
BY = np.array([0, 0, 4, 4, 0, 0, 4, 4, 8, 8, 12, 12, 8, 8, 12, 12])


def blk_index(bx: int, by: int) -> int:
    """4x4 block coordinates (0..3) -> zig-zag index."""
    return (by >> 1) * 8 + (bx >> 1) * 4 + (by & 1) * 2 + (bx & 1)


FRAME_HDR = np.dtype([
    ("magic", "<u4"), ("version", "<u4"), ("total_bytes", "<u4"),
    ("width_mbs", "<u2"), ("height_mbs", "<u2"),
    ("stride_Y", "<u4"), ("stride_C", "<u4"), ("plane_size_Y", "<u4"), ("plane_size_C", "<u4"),
    ("n_slices", "<u4"), ("slices_off", "<u4"), ("mbs_off", "<u4"), ("payload_off", "<u4"),
    ("payload_bytes", "<u4"), ("dst_slot", "<i4"), ("ref_slots", "<u4"), ("frame_id", "<i4"),
    ("n_coded_mbs", "<u4"), ("n_inter_mbs", "<u4"), ("motion_off", "<u4"), ("stream_id", "<u4"),
])
assert FRAME_HDR.itemsize == 80

SLICE_PARAMS = np.dtype([
    ("slice_type", "i1"), ("weighted_bipred_idc", "i1"),
    ("luma_log2_weight_denom", "i1"), ("chroma_log2_weight_denom", "i1"),
    ("FilterOffsetA", "i1"), ("FilterOffsetB", "i1"),
    ("disable_deblocking_filter_idc", "i1"), ("cabac", "i1"),
    ("first_mb", "<u4"), ("reserved", "<u4"),
    ("weightScale4x4", "u1", (6, 16)), ("weightScale8x8", "u1", (6, 64)),
    ("explicit_weights", "<i2", (3, 64)), ("explicit_offsets", "i1", (3, 64)),
    ("implicit_weights", "u1", (32, 32)), ("pad", "u1", (16,)),
])
assert SLICE_PARAMS.itemsize == 2112

MB = np.dtype([
    ("kind", "u1"), ("flags", "u1"), ("qp", "u1", (3,)), ("chroma_mode", "u1"), ("i16_mode", "u1"),
    ("reserved0", "u1"), ("nz_mask", "<u2"), ("slice", "<u2"), ("coded", "<u4"), ("payload_off", "<u4"),
    ("modes", "u1", (8,)), ("dbk_slice", "<u2"), ("reserved1", "<u2"),
])
assert MB.itemsize == 32

MOTION = np.dtype([("refPic", "i1", (8,)), ("refIdx", "i1", (8,)), ("mvs", "<i2", (64,))])
assert MOTION.itemsize == 144


def align16(x: int) -> int:
    return (x + 15) & ~15


def frame_geometry(width_mbs: int, height_mbs: int):
    """Strides / plane sizes exactly as the reference lays frames out
    (src/edge264_headers.c:2032-2046): chroma rows hold Cb then Cr."""
    stride_Y = width_mbs * 16
    if stride_Y % 2048 == 0:
        stride_Y += 16
    stride_C = width_mbs * 16
    if stride_C % 4096 == 0:
        stride_C += 8
    return dict(stride_Y=stride_Y, stride_C=stride_C,
                plane_size_Y=stride_Y * height_mbs * 16, plane_size_C=stride_C * height_mbs * 8)


def frame_bytes(width_mbs: int, height_mbs: int) -> int:
    g = frame_geometry(width_mbs, height_mbs)
    return g["plane_size_Y"] + g["plane_size_C"]


_NMV = (1, 2, 2, 4)
_SEL = ((0, 0, 0, 0), (0, 0, 1, 1), (0, 1, 0, 1), (0, 1, 2, 3))   # 4x4 block j of a quadrant -> vector of its record
_IDX = ((0,), (0, 2), (0, 1), (0, 1, 2, 3))                       # vectors stored for each partition shape


def motion_compact(mo) -> tuple[bytes, int]:
    """Expanded motion of one macroblock (MOTION record) -> (compact record bytes, mot_hdr): include/edge264_cmd.h
    e264_motion_compact, same bytes."""
    rp, ri = [int(x) for x in mo["refPic"]], [int(x) for x in mo["refIdx"]]
    mv = np.ascontiguousarray(mo["mvs"]).view("<i4")
    out, h = bytearray(), 0
    for l in range(2):
        v = [int(x) for x in mv[l * 16:l * 16 + 16]]
        if all(rp[l * 4 + q] >= 0 and rp[l * 4 + q] == rp[l * 4] and ri[l * 4 + q] == ri[l * 4] for q in range(4)) and all(x == v[0] for x in v):
            h |= 15 << (l * 4) | 1 << (8 + l)
            out += bytes([rp[l * 4] & 255, ri[l * 4] & 255, 0, 0]) + int(v[0]).to_bytes(4, "little", signed=True)
            continue
        for q in range(4):
            if rp[l * 4 + q] < 0:
                continue
            w = v[q * 4:q * 4 + 4]
            sub = 0 if w[0] == w[1] == w[2] == w[3] else 1 if (w[0] == w[1] and w[2] == w[3]) else 2 if (w[0] == w[2] and w[1] == w[3]) else 3
            h |= 1 << (l * 4 + q) | sub << (10 + 2 * (l * 4 + q))
            out += bytes([rp[l * 4 + q] & 255, ri[l * 4 + q] & 255, 0, 0])
            for j in _IDX[sub]:
                out += int(w[j]).to_bytes(4, "little", signed=True)
    return bytes(out), h


def motion_expand(h: int, rec, mo) -> int:
    """Compact record (bytes-like at its first byte) -> expanded MOTION record `mo`; returns the record's size."""
    n = 0
    mv = mo["mvs"].view("<i4")
    for l in range(2):
        if h >> (8 + l) & 1:
            mo["refPic"][l * 4:l * 4 + 4] = np.int8(np.uint8(rec[n]).view(np.int8))
            mo["refIdx"][l * 4:l * 4 + 4] = np.int8(np.uint8(rec[n + 1]).view(np.int8))
            mv[l * 16:l * 16 + 16] = int.from_bytes(bytes(rec[n + 4:n + 8]), "little", signed=True)
            n += 8
            continue
        for q in range(4):
            lq = l * 4 + q
            if not h >> lq & 1:
                mo["refPic"][lq] = mo["refIdx"][lq] = -1
                mv[lq * 4:lq * 4 + 4] = 0
                continue
            sub = h >> (10 + 2 * lq) & 3
            mo["refPic"][lq] = np.uint8(rec[n]).view(np.int8)
            mo["refIdx"][lq] = np.uint8(rec[n + 1]).view(np.int8)
            vs = [int.from_bytes(bytes(rec[n + 4 + 4 * j:n + 8 + 4 * j]), "little", signed=True) for j in range(_NMV[sub])]
            n += 4 + 4 * _NMV[sub]
            for j in range(4):
                mv[lq * 4 + j] = vs[_SEL[sub][j]]
    return n


class PacketBuilder:
    """Assembles one frame packet.  Mirrors what the C emitters of
    edge264_amd/frontend do, so tests can fabricate packets without a bitstream."""

    def __init__(self, width_mbs: int, height_mbs: int, dst_slot: int, frame_id: int = 0):
        self.w, self.h = width_mbs, height_mbs
        self.dst_slot, self.frame_id = dst_slot, frame_id
        self.slices: list[np.ndarray] = []
        self.mbs = np.zeros(width_mbs * height_mbs, dtype=MB)
        self.payload = bytearray()
        self.motion = np.zeros(width_mbs * height_mbs, dtype=MOTION)
        self.motion['refPic'] = -1
        self.motion['refIdx'] = -1
        self.ref_slots = 0

    def add_slice(self, **kw) -> int:
        s = np.zeros((), dtype=SLICE_PARAMS)
        s["weightScale4x4"] = 16
        s["weightScale8x8"] = 16
        s["implicit_weights"] = 32 + 64
        s["slice_type"] = 2
        for k, v in kw.items():
            s[k] = v
        self.slices.append(s)
        return len(self.slices) - 1

    def set_mb(self, addr: int, *, kind: int, slice_idx: int, qp, flags: int = 0, chroma_mode: int = 0,
               i16_mode: int = 0, nz_mask: int = 0, modes=None, motion=None, pcm=None,
               luma_dc=None, chroma_dc=None, luma_blocks=None, chroma_blocks=None, lev8=None) -> None:
        m = self.mbs[addr]
        m["kind"], m["flags"], m["qp"], m["slice"] = kind, flags, qp, slice_idx
        m["dbk_slice"] = slice_idx
        m["chroma_mode"], m["i16_mode"], m["nz_mask"] = chroma_mode, i16_mode, nz_mask
        if modes is not None:
            mm = np.zeros(8, np.uint8)
            if kind == MB_I4x4:
                for k, v in enumerate(modes):
                    mm[k >> 1] |= (int(v) & 15) << (4 * (k & 1))
            else:
                mm[:4] = modes
            m["modes"] = mm
        while len(self.payload) % 8:
            self.payload.append(0)
        m["payload_off"] = len(self.payload)
        coded = 0
        if kind == MB_INTER:
            mo = self.motion[addr]
            mo["refPic"], mo["refIdx"], mo["mvs"] = motion["refPic"], motion["refIdx"], np.asarray(motion["mvs"]).reshape(64)
            for r in np.asarray(motion["refPic"]).reshape(8):
                if r >= 0:
                    self.ref_slots |= 1 << int(r)
        if kind == MB_PCM:
            assert len(pcm) == 384
            self.payload += bytes(pcm)
        if luma_dc is not None:
            coded |= CODED_LUMA_DC
            self.payload += np.asarray(luma_dc, "<i2").reshape(16).tobytes()
        if chroma_dc is not None:
            coded |= CODED_CHROMA_DC
            self.payload += np.asarray(chroma_dc, "<i2").reshape(8).tobytes()
        # E264_MBF_LEV8: one byte per AC level when all of them fit (what the C emitters do, e264_emit.h:e264_flush_mb)
        ac = [np.asarray(v) for v in list((luma_blocks or {}).values()) + list((chroma_blocks or {}).values())]
        lev8 = bool(ac) and lev8 is not False and all(a.min() >= -128 and a.max() <= 127 for a in ac)
        dt = "i1" if lev8 else "<i2"
        if lev8:
            m["flags"] = flags | MBF_LEV8
        for k in sorted(luma_blocks or {}):
            n = 64 if flags & MBF_T8x8 else 16
            assert not (flags & MBF_T8x8) or k % 4 == 0
            coded |= 1 << k
            self.payload += np.asarray(luma_blocks[k]).astype(dt).reshape(n).tobytes()
        for k in sorted(chroma_blocks or {}):
            coded |= 1 << (16 + k)
            self.payload += np.asarray(chroma_blocks[k]).astype(dt).reshape(16).tobytes()
        m["coded"] = coded

    def finish(self) -> bytes:
        g = frame_geometry(self.w, self.h)
        hdr = np.zeros((), FRAME_HDR)
        slices_off = align16(FRAME_HDR.itemsize)
        mbs_off = align16(slices_off + SLICE_PARAMS.itemsize * len(self.slices))
        has_motion = bool((self.mbs["kind"] == MB_INTER).any())
        motion_off = align16(mbs_off + MB.itemsize * len(self.mbs))
        mot = bytearray()  # compact motion records of the inter macroblocks; their directory goes into E264Mb.modes
        for a in np.nonzero(self.mbs["kind"] == MB_INTER)[0]:
            rec, h = motion_compact(self.motion[a])
            self.mbs["modes"][a] = np.frombuffer(np.array([len(mot), h], "<u4").tobytes(), np.uint8)
            mot += rec
        payload_off = align16(motion_off + len(mot))
        pay = bytes(self.payload) + bytes(-len(self.payload) % 16)
        total = payload_off + len(pay)
        hdr["magic"], hdr["version"], hdr["total_bytes"] = E264_MAGIC, E264_VERSION, total
        hdr["width_mbs"], hdr["height_mbs"] = self.w, self.h
        for k, v in g.items():
            hdr[k] = v
        hdr["n_slices"], hdr["slices_off"], hdr["mbs_off"] = len(self.slices), slices_off, mbs_off
        hdr["payload_off"], hdr["payload_bytes"] = payload_off, len(pay)
        hdr["dst_slot"], hdr["ref_slots"], hdr["frame_id"] = self.dst_slot, self.ref_slots, self.frame_id
        hdr["n_coded_mbs"] = int((self.mbs["kind"] != MB_ABSENT).sum())
        hdr["n_inter_mbs"] = int((self.mbs["kind"] == MB_INTER).sum())
        hdr["motion_off"] = motion_off if has_motion else 0
        out = bytearray(total)
        out[:FRAME_HDR.itemsize] = hdr.tobytes()
        for i, s in enumerate(self.slices):
            o = slices_off + i * SLICE_PARAMS.itemsize
            out[o:o + SLICE_PARAMS.itemsize] = s.tobytes()
        out[mbs_off:mbs_off + self.mbs.nbytes] = self.mbs.tobytes()
        if has_motion:
            out[motion_off:motion_off + len(mot)] = mot
        out[payload_off:] = pay
        return bytes(out)


class Packet:
    """Read-only parsed view of a packet."""

    def __init__(self, data: bytes):
        if len(data) >= 8 and int.from_bytes(bytes(data[0:4]), "little") == E264_MAGIC and int.from_bytes(bytes(data[4:8]), "little") == E264_VERSION_COMPACT:
            # a wire packet (include/edge264_compact.h) is read as the version-4 packet it stands for: one definition of the expansion, the C one
            from . import backend
            data = backend.packet_expand(bytes(data))
        self.data = data
        self.hdr = np.frombuffer(data, FRAME_HDR, 1)[0]
        if self.hdr["magic"] != E264_MAGIC or self.hdr["version"] != E264_VERSION:
            raise ValueError("not an edge264 command packet")
        if self.hdr["total_bytes"] > len(data):
            raise ValueError("truncated packet")
        self.slices = np.frombuffer(data, SLICE_PARAMS, int(self.hdr["n_slices"]), int(self.hdr["slices_off"]))
        n = int(self.hdr["width_mbs"]) * int(self.hdr["height_mbs"])
        self.mbs = np.frombuffer(data, MB, n, int(self.hdr["mbs_off"]))
        self.payload_off = int(self.hdr["payload_off"])
        self._motion = None

    @property
    def motion(self):
        """Motion of every macroblock in EXPANDED form (one MOTION record per macroblock address; refPic -1 everywhere for
        macroblocks that are not inter), None if the frame has no inter macroblock."""
        if self._motion is None and int(self.hdr["motion_off"]):
            n = len(self.mbs)
            mo = np.zeros(n, MOTION)
            mo["refPic"] = -1
            mo["refIdx"] = -1
            sec = memoryview(self.data)[int(self.hdr["motion_off"]):self.payload_off]
            dirs = np.ascontiguousarray(self.mbs["modes"]).view("<u4").reshape(n, 2)
            for a in np.nonzero(self.mbs["kind"] == MB_INTER)[0]:
                off, h = int(dirs[a, 0]), int(dirs[a, 1])
                motion_expand(h, sec[off:off + 160], mo[a])
            self._motion = mo
        return self._motion

    def motion_bytes(self) -> int:
        """Bytes of compact motion records the packet carries."""
        if not int(self.hdr["motion_off"]):
            return 0
        dirs = np.ascontiguousarray(self.mbs["modes"]).view("<u4").reshape(len(self.mbs), 2)
        tot = 0
        for a in np.nonzero(self.mbs["kind"] == MB_INTER)[0]:
            h = int(dirs[a, 1])
            for l in range(2):
                tot += 8 if h >> (8 + l) & 1 else sum(4 + 4 * _NMV[h >> (10 + 2 * (l * 4 + q)) & 3] for q in range(4) if h >> (l * 4 + q) & 1)
        return tot

    @property
    def width_mbs(self) -> int:
        return int(self.hdr["width_mbs"])

    @property
    def height_mbs(self) -> int:
        return int(self.hdr["height_mbs"])

    def frame_bytes(self) -> int:
        return int(self.hdr["plane_size_Y"]) + int(self.hdr["plane_size_C"])

    def algorithmic_bytes(self) -> int:
        """F written once + F x prediction directions actually used
        (averaged over macroblocks) + command bytes consumed."""
        F = self.frame_bytes()
        n = len(self.mbs)
        dirs = 0
        inter = np.nonzero(self.mbs["kind"] == MB_INTER)[0]
        for a in inter:
            mo = self.motion[a]
            rp = mo["refPic"].reshape(2, 4)
            dirs += ((rp[0] >= 0).sum() + (rp[1] >= 0).sum()) / 4.0
        return int(F + F * dirs / n + int(self.hdr["total_bytes"]))

    def traffic_model(self) -> dict:
        """The terms of SURVEY.md 8(d) one by one, so that bench.py can price every kernel against the bytes IT has to move
        and keep sample bytes apart from command bytes (a fatter packet must not raise a roofline fraction):
          F                frame bytes (luma + chroma)
          inter / intra    fraction of macroblocks reconstructed by the prediction kernel (inter, PCM) / the intra kernel
          dirs             prediction directions used, averaged over ALL macroblocks (reference samples read = F x dirs)
          cmd_*            command bytes by consumer: record headers (32 B per macroblock + frame/slice headers), motion records,
                           coefficient payload of inter (+ PCM samples) and of intra macroblocks"""
        F = self.frame_bytes()
        n = len(self.mbs)
        kind = self.mbs["kind"]
        inter = np.nonzero(kind == MB_INTER)[0]
        dirs = 0.0
        if len(inter):
            rp = self.motion["refPic"][inter].reshape(-1, 2, 4)
            dirs = float((rp >= 0).sum()) / 4.0
        pay = np.zeros(n, np.int64)
        coded = self.mbs["coded"].astype(np.int64)
        t8 = (self.mbs["flags"] & MBF_T8x8) != 0
        pop16 = np.array([bin(int(c) & 0xffff).count("1") for c in coded])
        pop8 = np.array([bin(int(c) & 0x1111).count("1") for c in coded])
        popc = np.array([bin(int(c) >> 16 & 0xff).count("1") for c in coded])
        ac = popc * 32 + np.where(t8 & (kind != MB_I16x16), pop8 * 128, pop16 * 32)
        pay += np.where(coded & CODED_LUMA_DC, 32, 0) + np.where(coded & CODED_CHROMA_DC, 16, 0)
        pay += np.where((self.mbs["flags"] & MBF_LEV8) != 0, ac // 2, ac)
        pay = np.where(kind == MB_PCM, 384, pay)
        own_pred = (kind == MB_INTER) | (kind == MB_PCM)
        own_intra = (kind == MB_I4x4) | (kind == MB_I8x8) | (kind == MB_I16x16)
        return dict(F=F, n_mbs=n, inter=float(own_pred.sum()) / n, intra=float(own_intra.sum()) / n, dirs=dirs / n,
                    cmd_headers=int(self.hdr["mbs_off"]) + 32 * n, cmd_motion=self.motion_bytes(),
                    cmd_payload_inter=int(pay[own_pred].sum()), cmd_payload_intra=int(pay[own_intra].sum()),
                    cmd_total=int(self.hdr["total_bytes"]))


def refresh_summary(buf: bytearray) -> None:
    """Recomputes the header's summary fields (n_coded_mbs, n_inter_mbs, ref_slots) from the records of a packet that was
    edited in place (tests that drop macroblocks, tools that cut captures): e264hip_packet_check rejects a header that
    disagrees with its records -- the kernels' early exits and the trusted submission path rely on these fields."""
    pk = Packet(bytes(buf))
    hdr = np.frombuffer(buf, FRAME_HDR, 1)
    kind = pk.mbs["kind"]
    hdr["n_coded_mbs"] = int((kind != MB_ABSENT).sum())
    hdr["n_inter_mbs"] = int((kind == MB_INTER).sum())
    refs = 0
    inter = np.nonzero(kind == MB_INTER)[0]
    if len(inter) and pk.motion is not None:
        for r in np.unique(pk.motion["refPic"][inter]):
            if r >= 0:
                refs |= 1 << int(r)
    hdr["ref_slots"] = refs


def split_planes(buf: np.ndarray, width_mbs: int, height_mbs: int):
    """Frame buffer bytes -> (Y, Cb, Cr) 2-D views in the reference layout."""
    g = frame_geometry(width_mbs, height_mbs)
    Y = buf[:g["plane_size_Y"]].reshape(height_mbs * 16, g["stride_Y"])[:, :width_mbs * 16]
    C = buf[g["plane_size_Y"]:g["plane_size_Y"] + g["plane_size_C"]].reshape(height_mbs * 8, g["stride_C"])
    return Y, C[:, :width_mbs * 8], C[:, g["stride_C"] // 2:g["stride_C"] // 2 + width_mbs * 8]

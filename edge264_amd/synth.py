"""Synthetic command-packet generator (no bitstream involved).

Produces legal random frame packets for the workloads BASELINE.json names:
all-intra I frames (config 2), IPPP (config 3), IBBP with 8x8 transform,
custom scaling lists and weighted prediction (config 4).  "Legal" means what a
conforming front end could emit: intra modes that only read available
neighbours (the post-remap internal modes of the reference,
src/edge264_slice.c:573-594), chroma AC only with chroma DC, motion expanded
per 4x4 block, references pointing at already decoded DPB slots.

Used by tests (small frames, bit-exact HIP vs oracle vs reference kernels)
and by bench.py (1080p).  Seeded, so every rank / run sees the same bytes.
"""
from __future__ import annotations

import numpy as np

from . import packet as P

# chroma QP table (Table 8-15; reference src/edge264_headers.c:171-175 QP_Y2C)
_QPC = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]


def chroma_qp(qp: int, offset: int) -> int:
    return _QPC[min(max(qp + offset, 0), 51)]


# internal mode numbers, src/edge264_internal.h:564-634
I4_V, I4_H, I4_DC, I4_DC_A, I4_DC_B, I4_DC_AB, I4_DDL, I4_DDL_C, I4_DDR, I4_VR, I4_HD, I4_VL, I4_VL_C, I4_HU = range(14)
(I8_V, I8_V_C, I8_V_D, I8_V_CD, I8_H, I8_H_D, I8_DC, I8_DC_A, I8_DC_AC, I8_DC_AD, I8_DC_ACD, I8_DC_B, I8_DC_BD,
 I8_DC_C, I8_DC_D, I8_DC_CD, I8_DC_AB, I8_DDL, I8_DDL_C, I8_DDL_D, I8_DDL_CD, I8_DDR, I8_DDR_C, I8_VR, I8_VR_C,
 I8_HD, I8_VL, I8_VL_C, I8_VL_D, I8_VL_CD, I8_HU, I8_HU_D) = range(32)
I16_V, I16_H, I16_DC, I16_DC_A, I16_DC_B, I16_DC_AB, I16_P = range(7)
IC_DC, IC_DC_A, IC_DC_B, IC_DC_AB, IC_H, IC_V, IC_P = range(7)


def modes4x4(a: bool, b: bool, c: bool, d: bool) -> list[int]:
    """Internal Intra4x4 modes usable when neighbours A(left) B(top) C(top-right)
    D(top-left) are available (True) or not."""
    out = [I4_DC if a and b else I4_DC_A if b else I4_DC_B if a else I4_DC_AB]
    if b:
        out += [I4_V, I4_DDL if c else I4_DDL_C, I4_VL if c else I4_VL_C]
    if a:
        out += [I4_H, I4_HU]
    if a and b and d:
        out += [I4_DDR, I4_VR, I4_HD]
    return out


def modes8x8(a: bool, b: bool, c: bool, d: bool) -> list[int]:
    if a and b:
        out = [[I8_DC_CD, I8_DC_C], [I8_DC_D, I8_DC]][c][d]
    elif b:
        out = [[I8_DC_ACD, I8_DC_AC], [I8_DC_AD, I8_DC_A]][c][d]
    elif a:
        out = I8_DC_B if d else I8_DC_BD
    else:
        out = I8_DC_AB
    out = [out]
    if b:
        i = (0 if c else 1) + (0 if d else 2)
        out += [[I8_V, I8_V_C, I8_V_D, I8_V_CD][i], [I8_DDL, I8_DDL_C, I8_DDL_D, I8_DDL_CD][i],
                [I8_VL, I8_VL_C, I8_VL_D, I8_VL_CD][i]]
    if a:
        out += [I8_H if d else I8_H_D, I8_HU if d else I8_HU_D]
    if a and b and d:
        out += [I8_DDR if c else I8_DDR_C, I8_VR if c else I8_VR_C, I8_HD]
    return out


def modes16x16(a: bool, b: bool, d: bool) -> list[int]:
    out = [I16_DC if a and b else I16_DC_A if b else I16_DC_B if a else I16_DC_AB]
    if b:
        out.append(I16_V)
    if a:
        out.append(I16_H)
    if a and b and d:
        out.append(I16_P)
    return out


def modes_chroma(a: bool, b: bool, d: bool) -> list[int]:
    out = [IC_DC if a and b else IC_DC_A if b else IC_DC_B if a else IC_DC_AB]
    if b:
        out.append(IC_V)
    if a:
        out.append(IC_H)
    if a and b and d:
        out.append(IC_P)
    return out


class StreamSynth:
    """Generates the packets of one synthetic stream, frame by frame.

    gop: string of frame types, e.g. "IPPP" or "IBBP" (decode order).  B frames
    reference the two most recent non-B frames, P frames up to `num_refs` of them.
    """

    def __init__(self, width_mbs: int, height_mbs: int, seed: int = 0, *, cabac_like: bool = True,
                 t8x8: bool = False, scaling: bool = False, weighted: int = 0, deblock: bool = True,
                 slices_per_frame: int = 1, intra_in_inter: float = 0.05, p_skip: float = 0.1,
                 residual_prob: float = 0.3, num_refs: int = 2, mv_range: int = 64, pcm_prob: float = 0.0,
                 stress: bool = False, qp_base: int = 28, i_kinds=(P.MB_I4x4, P.MB_I16x16),
                 filter_offsets=(0, 0), deblock_idc: int = 0, n_slots: int = 6):
        self.w, self.h = width_mbs, height_mbs
        self.rng = np.random.default_rng(seed)
        self.t8x8, self.scaling, self.weighted, self.deblock = t8x8, scaling, weighted, deblock
        self.slices_per_frame = slices_per_frame
        self.intra_in_inter, self.p_skip, self.residual_prob = intra_in_inter, p_skip, residual_prob
        self.num_refs, self.mv_range, self.pcm_prob, self.stress = num_refs, mv_range, pcm_prob, stress
        self.qp_base, self.i_kinds = qp_base, tuple(i_kinds)
        self.filter_offsets, self.deblock_idc = filter_offsets, deblock_idc
        self.n_slots = n_slots
        self.frame_no = 0
        self.refs: list[int] = []  # DPB slots of decoded reference (non-B) frames, newest first
        self.cabac_like = cabac_like
        self.cqp_off = (int(self.rng.integers(-3, 4)), int(self.rng.integers(-3, 4)))

    # ---- helpers ---------------------------------------------------------
    def _free_slot(self) -> int:
        keep = set(self.refs[:max(self.num_refs, 2)])
        for s in range(self.n_slots):
            if s not in keep:
                return s
        raise RuntimeError("DPB too small")

    def _levels(self, n: int, qp: int, maxnz: int, lowfreq: int) -> np.ndarray:
        """n int16 levels, a few nonzero at low-frequency positions of the
        reference's transposed order (position = x*N+y, small x+y)."""
        rng = self.rng
        c = np.zeros(n, np.int16)
        N = 4 if n == 16 else 8
        if self.stress:
            lim = 2000
        else:
            lim = max(1, min(48, 4000 // (16 << (qp // 6))))
        cnt = int(rng.integers(1, maxnz + 1))
        for _ in range(cnt):
            x, y = int(rng.integers(0, lowfreq)), int(rng.integers(0, lowfreq))
            v = int(rng.integers(-lim, lim + 1))
            c[(x % N) * N + (y % N)] = v if v else 1
        return c

    def _slice_params(self, b: P.PacketBuilder, ftype: str, first_mb: int, l0: list[int], l1: list[int]):
        rng = self.rng
        kw = dict(slice_type={"I": 2, "P": 0, "B": 1}[ftype], first_mb=first_mb, cabac=int(self.cabac_like),
                  FilterOffsetA=self.filter_offsets[0], FilterOffsetB=self.filter_offsets[1],
                  disable_deblocking_filter_idc=(self.deblock_idc if self.deblock else 1))
        if self.scaling:
            kw["weightScale4x4"] = rng.integers(8, 40, (6, 16)).astype(np.uint8)
            kw["weightScale8x8"] = rng.integers(8, 40, (6, 64)).astype(np.uint8)
        ew = np.zeros((3, 64), np.int16)
        eo = np.zeros((3, 64), np.int8)
        lwd = cwd = 0
        idc = 0
        if ftype != "I" and self.weighted == 1:
            idc = 1
            lwd, cwd = int(rng.integers(3, 7)), int(rng.integers(3, 7))
            for pl, wd in ((0, lwd), (1, cwd), (2, cwd)):
                ew[pl, :] = 1 << wd
            for lx, lst in ((0, l0), (1, l1)):
                for i in range(len(lst)):
                    if rng.random() < 0.7:  # luma_weight_flag
                        ew[0, lx * 32 + i] = rng.integers(-32, 97)
                        eo[0, lx * 32 + i] = rng.integers(-20, 21)
                    if rng.random() < 0.7:  # chroma_weight_flag
                        ew[1:, lx * 32 + i] = rng.integers(-32, 97, 2)
                        eo[1:, lx * 32 + i] = rng.integers(-20, 21, 2)
        elif ftype == "B" and self.weighted == 2:
            idc = 2
        iw = np.full((32, 32), 32 + 64, np.uint8)
        if idc == 2:
            # any w1 the reference can produce: DistScaleFactor>>2 in [-64,128] (headers.c:244-252)
            iw[:len(l0), :len(l1)] = (rng.integers(-64, 129, (len(l0), len(l1))) + 64).astype(np.uint8)
        kw.update(weighted_bipred_idc=idc, luma_log2_weight_denom=lwd, chroma_log2_weight_denom=cwd,
                  explicit_weights=ew, explicit_offsets=eo, implicit_weights=iw)
        return b.add_slice(**kw)

    # ---- one frame -------------------------------------------------------
    def next_frame(self, ftype: str) -> bytes:
        rng = self.rng
        W, H = self.w, self.h
        if ftype != "I" and not self.refs:
            ftype = "I"
        dst = self._free_slot()
        b = P.PacketBuilder(W, H, dst, self.frame_no)
        if ftype == "P":
            l0, l1 = self.refs[:self.num_refs], []
        elif ftype == "B":
            l0 = self.refs[1:2] + self.refs[0:1] if len(self.refs) > 1 else self.refs[:1]
            l1 = self.refs[:2]
        else:
            l0 = l1 = []
        n_mbs = W * H
        bounds = sorted({0, *[int(x) for x in rng.integers(1, n_mbs, self.slices_per_frame - 1)]}) if self.slices_per_frame > 1 else [0]
        slice_of = np.zeros(n_mbs, np.int32)
        slice_ids = []
        for si, fm in enumerate(bounds):
            slice_ids.append(self._slice_params(b, ftype, fm, l0, l1))
            slice_of[fm:] = si
        first_of = np.array(bounds)[slice_of]
        qp = self.qp_base
        for addr in range(n_mbs):
            mbx, mby = addr % W, addr // W
            fm = int(first_of[addr])
            sidx = slice_ids[int(slice_of[addr])]

            def avail(n_addr: int, ok: bool) -> bool:
                return ok and n_addr >= fm
            A = avail(addr - 1, mbx > 0)
            B = avail(addr - W, mby > 0)
            C = avail(addr - W + 1, mby > 0 and mbx < W - 1)
            D = avail(addr - W - 1, mby > 0 and mbx > 0)
            # deblocking flags as slice.c:1692-1762 derives filter_edges
            flags = 0
            if self.deblock and self.deblock_idc != 1:
                flags |= P.MBF_DEBLOCK
                if mbx > 0 and (A or self.deblock_idc == 0):
                    flags |= P.MBF_EDGE_LEFT
                if mby > 0 and (B or self.deblock_idc == 0):
                    flags |= P.MBF_EDGE_TOP
            qp = int(np.clip(qp + rng.integers(-2, 3), 10 if not self.stress else 0, 45 if not self.stress else 51))
            qps = (qp, chroma_qp(qp, self.cqp_off[0]), chroma_qp(qp, self.cqp_off[1]))
            inter = ftype != "I" and rng.random() >= self.intra_in_inter
            if not inter and rng.random() < self.pcm_prob:
                b.set_mb(addr, kind=P.MB_PCM, slice_idx=sidx, qp=(0, chroma_qp(0, self.cqp_off[0]), chroma_qp(0, self.cqp_off[1])),
                         flags=flags, nz_mask=0xffff, pcm=rng.integers(0, 256, 384, dtype=np.uint8).tobytes())
                continue
            luma_blocks, chroma_blocks, luma_dc, chroma_dc = {}, {}, None, None
            has_res = (not inter) or rng.random() < self.residual_prob
            kind = P.MB_INTER
            t8 = False
            kw = {}
            if inter:
                t8 = self.t8x8 and rng.random() < 0.5
                kw["motion"] = self._motion(ftype, l0, l1, t8)
            else:
                kinds = list(self.i_kinds)
                kind = int(kinds[int(rng.integers(0, len(kinds)))])
                kw["chroma_mode"] = int(rng.choice(modes_chroma(A, B, D)))
                if kind == P.MB_I16x16:
                    kw["i16_mode"] = int(rng.choice(modes16x16(A, B, D)))
                elif kind == P.MB_I8x8:
                    t8 = True
                    mm = []
                    for blk in range(4):
                        bx, by = blk & 1, blk >> 1
                        a = A if bx == 0 else True
                        bb = B if by == 0 else True
                        c = (B if bx == 0 else C) if by == 0 else (bx == 0)
                        d = (D if bx == 0 else B) if by == 0 else (A if bx == 0 else True)
                        mm.append(int(rng.choice(modes8x8(a, bb, c, d))))
                    kw["modes"] = mm
                else:
                    mm = []
                    for k in range(16):
                        bx, by = int(P.BX[k]) // 4, int(P.BY[k]) // 4
                        a = A if bx == 0 else True
                        bb = B if by == 0 else True
                        if by == 0:
                            c = B if bx < 3 else C
                        elif bx == 3:
                            c = False
                        else:
                            c = P.blk_index(bx + 1, by - 1) < k
                        d = (D if bx == 0 else B) if by == 0 else (A if bx == 0 else True)
                        mm.append(int(rng.choice(modes4x4(a, bb, c, d))))
                    kw["modes"] = mm
            nz = 0
            if has_res:
                if kind == P.MB_I16x16:
                    if rng.random() < 0.8:
                        luma_dc = self._levels(16, qp, 6, 4)
                    if rng.random() < 0.5:
                        for k in range(16):
                            if rng.random() < 0.5:
                                c = self._levels(16, qp, 4, 3)
                                c[0] = 0
                                luma_blocks[k] = c
                                nz |= 1 << k
                elif t8:
                    for blk in range(4):
                        if rng.random() < 0.6:
                            luma_blocks[blk * 4] = self._levels(64, qp, 8, 4)
                            nz |= 0xf << (blk * 4)
                else:
                    for k in range(16):
                        if rng.random() < 0.5:
                            luma_blocks[k] = self._levels(16, qp, 6, 3)
                            nz |= 1 << k
                if rng.random() < 0.6:
                    chroma_dc = self._levels(16, qps[1], 4, 4)[:8].copy()
                    chroma_dc[0] |= 1
                    if rng.random() < 0.5:
                        for k in range(8):
                            if rng.random() < 0.5:
                                c = self._levels(16, qps[1 + (k >> 2)], 4, 3)
                                c[0] = 0
                                chroma_blocks[k] = c
            if t8:
                flags |= P.MBF_T8x8
            b.set_mb(addr, kind=kind, slice_idx=sidx, qp=qps, flags=flags, nz_mask=nz, luma_dc=luma_dc,
                     chroma_dc=chroma_dc, luma_blocks=luma_blocks, chroma_blocks=chroma_blocks, **kw)
        self.frame_no += 1
        if ftype != "B":
            self.refs.insert(0, dst)
            del self.refs[max(self.num_refs, 2):]
        return b.finish()

    def _motion(self, ftype: str, l0: list[int], l1: list[int], t8: bool) -> dict:
        rng = self.rng
        refPic = np.full(8, -1, np.int8)
        refIdx = np.full(8, -1, np.int8)
        mvs = np.zeros((2, 16, 2), np.int16)
        r = rng.random()
        if r < self.p_skip or (r < 0.5 + self.p_skip / 2):
            shape = 0  # 16x16 (incl. skip)
        elif r < 0.7:
            shape = 1  # 16x8 / 8x16
        else:
            shape = 2  # 8x8 with sub-partitions
        big = rng.random() < 0.05
        rangemv = self.mv_range * (16 if big else 1)

        def mv():
            return rng.integers(-rangemv, rangemv + 1, 2)
        if ftype == "P":
            use = [(True, False)] * 4
        else:
            opts = [(True, False), (False, True), (True, True)]
            if shape == 0:
                use = [opts[int(rng.integers(0, 3))]] * 4
            else:
                use = [opts[int(rng.integers(0, 3))] for _ in range(4)]
                if shape == 1:
                    if rng.random() < 0.5:
                        use = [use[0], use[0], use[2], use[2]]
                    else:
                        use = [use[0], use[1], use[0], use[1]]
        for lx, lst in ((0, l0), (1, l1)):
            if not lst:
                continue
            if shape == 0:
                idx = [int(rng.integers(0, len(lst)))] * 4
            elif shape == 1:
                i0, i1 = int(rng.integers(0, len(lst))), int(rng.integers(0, len(lst)))
                idx = [i0, i0, i1, i1] if use[0] == use[1] else [i0, i1, i0, i1]
            else:
                idx = [int(rng.integers(0, len(lst))) for _ in range(4)]
            for b8 in range(4):
                if use[b8][lx]:
                    refIdx[lx * 4 + b8] = idx[b8]
                    refPic[lx * 4 + b8] = lst[idx[b8]]
            if shape == 0:
                if use[0][lx]:
                    mvs[lx, :, :] = mv()
            elif shape == 1:
                va, vb = mv(), mv()
                horiz = use[0] == use[1] and idx[0] == idx[1]
                for k in range(16):
                    b8 = k >> 2
                    if use[b8][lx]:
                        first = (b8 < 2) if horiz else (b8 % 2 == 0)
                        mvs[lx, k] = va if first else vb
            else:
                for b8 in range(4):
                    if not use[b8][lx]:
                        continue
                    sub = int(rng.integers(0, 4)) if not t8 else 0  # 8x8, 8x4, 4x8, 4x4
                    v = [mv() for _ in range(4)]
                    for j in range(4):
                        sel = [0, j >> 1, j & 1, j][sub]
                        mvs[lx, b8 * 4 + j] = v[sel]
        return dict(refPic=refPic, refIdx=refIdx, mvs=mvs)

    def gop(self, pattern: str) -> list[bytes]:
        return [self.next_frame(t) for t in pattern]

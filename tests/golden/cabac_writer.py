"""tests/golden/cabac_writer.py -- CABAC slice_data() writer for the fixture generator (make_streams.py).

TEST INFRASTRUCTURE.  The reference's own stream generator has no CABAC path
(tests/gen_avc.py: gen_slice_data_cabac is undefined), and real-world H.264 is mostly CABAC, so the boundary
tests need CABAC streams.  This module encodes the SAME macroblock descriptions make_streams.py feeds to
gen_avc.py's CAVLC writer (mb_type, prediction modes, sub_mb_types, ref_idx, mvds, coded_block_pattern,
transform_size_8x8_flag, mb_qp_delta, coefficient lists in scan order), following ITU-T H.264 9.3:
binarisations (9.3.2), context selection (9.3.3.1) and the arithmetic encoder (9.3.4.2 / Figure 9-7..9-12).

Scope: I, P and B slices, frame macroblocks, 4:2:0; every macroblock type make_streams.py produces, I_PCM included.  Correctness is established by the unmodified reference decoder: it must decode the CABAC
stream without error to exactly the frames of the CAVLC stream generated from the same description.

Tables: the context initialisation values (Tables 9-12..9-33) are read at generation time from the reference
library's exported `cabac_context_init` symbol, the 8x8 significance maps (Table 9-43) from its source text --
nothing of the reference is stored in this repository.  rangeTabLPS / transIdxLPS are Table 9-44 / 9-45.
"""
import ctypes
import os
import re

# ---- Table 9-44 rangeTabLPS, Table 9-45 transIdxLPS (ITU-T H.264) -------------------------------------
RANGE_LPS = [
    (128, 176, 208, 240), (128, 167, 197, 227), (128, 158, 187, 216), (123, 150, 178, 205), (116, 142, 169, 195),
    (111, 135, 160, 185), (105, 128, 152, 175), (100, 122, 144, 166), (95, 116, 137, 158), (90, 110, 130, 150),
    (85, 104, 123, 142), (81, 99, 117, 135), (77, 94, 111, 128), (73, 89, 105, 122), (69, 85, 100, 116),
    (66, 80, 95, 110), (62, 76, 90, 104), (59, 72, 86, 99), (56, 69, 81, 94), (53, 65, 77, 89),
    (51, 62, 73, 85), (48, 59, 69, 80), (46, 56, 66, 76), (43, 53, 63, 72), (41, 50, 59, 69),
    (39, 48, 56, 65), (37, 45, 54, 62), (35, 43, 51, 59), (33, 41, 48, 56), (32, 39, 46, 53),
    (30, 37, 43, 50), (29, 35, 41, 48), (27, 33, 39, 45), (26, 31, 37, 43), (24, 30, 35, 41),
    (23, 28, 33, 39), (22, 27, 32, 37), (21, 26, 30, 35), (20, 24, 29, 33), (19, 23, 27, 31),
    (18, 22, 26, 30), (17, 21, 25, 28), (16, 20, 23, 27), (15, 19, 22, 25), (14, 18, 21, 24),
    (14, 17, 20, 23), (13, 16, 19, 22), (12, 15, 18, 21), (12, 14, 17, 20), (11, 14, 16, 19),
    (11, 13, 15, 18), (10, 12, 15, 17), (10, 12, 14, 16), (9, 11, 13, 15), (9, 11, 12, 14),
    (8, 10, 12, 14), (8, 9, 11, 13), (7, 9, 11, 12), (7, 9, 10, 12), (7, 8, 10, 11),
    (6, 8, 9, 11), (6, 7, 9, 10), (6, 7, 8, 9), (2, 2, 2, 2)]
TRANS_LPS = [0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22,
             23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36,
             36, 37, 37, 37, 38, 38, 63]

REF_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle", "_ref", "libedge264_ref.so")
REF_SLICE_C = "/root/reference/src/edge264_slice.c"


def load_tables():
    """(context init [4][1024][2], sig_inc_8x8 frame [63], last_inc_8x8 [63]) from the reference, at run time."""
    lib = ctypes.CDLL(REF_LIB)
    raw = (ctypes.c_int8 * (4 * 1024 * 2)).in_dll(lib, "cabac_context_init")
    init = [[(raw[(t * 1024 + i) * 2], raw[(t * 1024 + i) * 2 + 1]) for i in range(1024)] for t in range(4)]
    src = open(REF_SLICE_C).read()

    def rows(name, count):
        body = src[src.index(name):]
        nums = re.findall(r"\{\s*((?:-?\d+\s*,\s*){15}-?\d+)\s*\}", body)
        out = []
        for r in nums[:count]:
            out += [int(x) for x in r.split(",")]
        return out
    sig = rows("sig_inc_8x8[2][4]", 4)      # first 4 rows = frame macroblocks
    last = rows("last_inc_8x8[4]", 4)
    assert len(sig) == 64 and len(last) == 64
    return init, sig, last


class Encoder:
    """9.3.4.2 arithmetic encoding engine."""

    def __init__(self, init_table, slice_qp):
        self.low, self.range, self.outstanding, self.first = 0, 510, 0, True
        self.bits = []
        self.state = []
        qp = max(0, min(51, slice_qp))
        for m, n in init_table:
            pre = max(1, min(126, ((m * qp) >> 4) + n))
            self.state.append([63 - pre, 0] if pre <= 63 else [pre - 64, 1])

    def _put(self, b):
        if self.first:
            self.first = False
        else:
            self.bits.append(b)
        self.bits.extend([1 - b] * self.outstanding)
        self.outstanding = 0

    def _renorm(self):
        while self.range < 256:
            if self.low < 256:
                self._put(0)
            elif self.low >= 512:
                self.low -= 512
                self._put(1)
            else:
                self.low -= 256
                self.outstanding += 1
            self.range <<= 1
            self.low <<= 1

    def decision(self, ctx, b):
        st = self.state[ctx]
        r_lps = RANGE_LPS[st[0]][(self.range >> 6) & 3]
        self.range -= r_lps
        if b != st[1]:
            self.low += self.range
            self.range = r_lps
            if st[0] == 0:
                st[1] = 1 - st[1]
            st[0] = TRANS_LPS[st[0]]
        else:
            st[0] = min(st[0] + 1, 62)
        self._renorm()

    def bypass(self, b):
        self.low <<= 1
        if b:
            self.low += self.range
        if self.low >= 1024:
            self._put(1)
            self.low -= 1024
        elif self.low < 512:
            self._put(0)
        else:
            self.low -= 512
            self.outstanding += 1

    def terminate(self, b):
        self.range -= 2
        if b:
            self.low += self.range
            self.range = 2
            self._renorm()
            self._put((self.low >> 9) & 1)
            self.bits.extend([(self.low >> 8) & 1, 1])   # WriteBits(((low >> 7) & 3) | 1, 2): the last 1 is the stop bit
        else:
            self._renorm()

    def pcm_samples(self, samples):
        """after terminate(1): pcm_alignment_zero_bits, the raw samples, then the engine starts afresh (9.3.1.2);
        the context variables keep their states."""
        self.bits += [0] * (-len(self.bits) % 8)
        for v in samples:
            self.bits += [(v >> k) & 1 for k in range(7, -1, -1)]
        self.low, self.range, self.outstanding, self.first = 0, 510, 0, True

    # ---- binarisation helpers -------------------------------------------------------------------
    def ueg_suffix(self, v, k):
        while v >= (1 << k):
            self.bypass(1)
            v -= 1 << k
            k += 1
        self.bypass(0)
        while k:
            k -= 1
            self.bypass((v >> k) & 1)


BLK_X = [0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3]
BLK_Y = [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]
# ctxIdxOffset of coded_block_flag / significant_coeff_flag / last_significant_coeff_flag / coeff_abs_level_minus1 per
# ctxBlockCat (Table 9-34 + ctxBlockCatOffset of Table 9-40); levelListIdx counts from 0 within the block
CAT = {0: (85, 105, 166, 227), 1: (89, 120, 181, 237), 2: (93, 134, 195, 247), 3: (97, 149, 210, 257),
       4: (101, 152, 213, 266), 5: (1012, 402, 417, 426)}


class MbState:
    __slots__ = ("slice", "skip", "intra", "nxn", "i16", "cbp_l", "cbp_c", "cmode", "t8", "cbf_y", "cbf_dc", "cbf_cdc",
                 "cbf_cac", "mvd", "refgt0", "inter", "direct", "pcm")

    def __init__(self):
        self.slice = -1


class FrameState:
    """Neighbour information of one picture (persists across its slices)."""

    def __init__(self, W, H):
        self.W, self.H = W, H
        self.mb = [MbState() for _ in range(W * H)]

    def get(self, mx, my, sl):
        if 0 <= mx < self.W and 0 <= my < self.H:
            m = self.mb[my * self.W + mx]
            if m.slice == sl:
                return m
        return None


class CabacSlice:
    def __init__(self, tables, fs, slice_type, slice_qp, cabac_init_idc, sl_id, t8x8_mode):
        init, self.sig8, self.last8 = tables
        self.enc = Encoder(init[0 if slice_type == 2 else 1 + cabac_init_idc], slice_qp)
        self.fs, self.st, self.sl = fs, slice_type, sl_id
        self.prev_qpd_nz = False
        self.t8x8_mode = t8x8_mode

    # ---- neighbour helpers ----------------------------------------------------------------------
    def nb(self, mx, my):
        return self.fs.get(mx - 1, my, self.sl), self.fs.get(mx, my - 1, self.sl)

    @staticmethod
    def blk_nb(bx, by):
        """for 4x4 luma block (bx,by in 0..3): ((dx_mb, blk index) of A, same of B)"""
        def idx(x, y):
            return [i for i in range(16) if BLK_X[i] == x and BLK_Y[i] == y][0]
        a = (-1, idx(3, by)) if bx == 0 else (0, idx(bx - 1, by))
        b = (-1, idx(bx, 3)) if by == 0 else (0, idx(bx, by - 1))
        return a, b

    # ---- syntax elements ------------------------------------------------------------------------
    def mb_skip_flag(self, mx, my, skip):
        A, B = self.nb(mx, my)
        inc = int(A is not None and not A.skip) + int(B is not None and not B.skip)
        self.enc.decision((11 if self.st == 0 else 24) + inc, int(skip))

    def mb_type_intra(self, mx, my, t, prefix_p):
        """t: I-slice numbering 0 = I_NxN, 1..24 = I_16x16"""
        e = self.enc
        if prefix_p and self.st == 1:   # B slice: prefix 1 1 1 1 0 1 (Table 9-37), suffix with ctxIdxOffset 32
            A, B = self.nb(mx, my)
            e.decision(27 + int(A is not None and not A.direct) + int(B is not None and not B.direct), 1)
            e.decision(30, 1); e.decision(31, 1); e.decision(32, 1); e.decision(32, 0); e.decision(32, 1)
            off, b0 = 32, 32
        elif prefix_p:
            e.decision(14, 1)
            off, b0 = 17, 17
        else:
            A, B = self.nb(mx, my)
            off = 3
            b0 = 3 + int(A is not None and not A.nxn) + int(B is not None and not B.nxn)
        if t == 0:
            e.decision(b0, 0)
            return
        e.decision(b0, 1)
        if t == 25:      # I_PCM: the terminate bin is 1 and flushes the arithmetic coder
            e.terminate(1)
            return
        e.terminate(0)
        t -= 1
        luma15, chroma, pred = t // 12, (t % 12) // 4, t % 4
        if prefix_p:
            e.decision(off + 1, luma15)
            e.decision(off + 2, int(chroma != 0))
            if chroma:
                e.decision(off + 2, int(chroma == 2))
            e.decision(off + 3, pred >> 1)
            e.decision(off + 3, pred & 1)
        else:
            e.decision(off + 3, luma15)
            e.decision(off + 4, int(chroma != 0))
            if chroma:
                e.decision(off + 5, int(chroma == 2))
                e.decision(off + 6, pred >> 1)
                e.decision(off + 7, pred & 1)
            else:
                e.decision(off + 6, pred >> 1)
                e.decision(off + 7, pred & 1)

    def mb_type_p(self, t):
        e = self.enc
        e.decision(14, 0)
        if t == 0:       # P_L0_16x16: 0 0 0
            e.decision(15, 0); e.decision(16, 0)
        elif t == 1:     # P_L0_L0_16x8: 0 1 1
            e.decision(15, 1); e.decision(17, 1)
        elif t == 2:     # P_L0_L0_8x16: 0 0 1 0 -> bins: b1=0, b2 ctx16 = 1 ... Table 9-37: 0 1 0
            e.decision(15, 1); e.decision(17, 0)
        else:            # P_8x8: 0 0 1
            e.decision(15, 0); e.decision(16, 1)

    def sub_mb_type_p(self, t):
        e = self.enc
        if t == 0:
            e.decision(21, 1)
        elif t == 1:
            e.decision(21, 0); e.decision(22, 0)
        elif t == 2:
            e.decision(21, 0); e.decision(22, 1); e.decision(23, 1)
        else:
            e.decision(21, 0); e.decision(22, 1); e.decision(23, 0)

    def mb_type_b(self, mx, my, t):
        """t = 0 (B_Direct_16x16) .. 22 (B_8x8), Table 7-14 / 9-37(b)"""
        e = self.enc
        A, B = self.nb(mx, my)
        ctx0 = 27 + int(A is not None and not A.direct) + int(B is not None and not B.direct)
        if t == 0:
            e.decision(ctx0, 0)
            return
        e.decision(ctx0, 1)
        if t <= 2:                       # 1 0 b
            e.decision(30, 0); e.decision(32, t - 1)
            return
        e.decision(30, 1)
        if 3 <= t <= 10:
            bits = [(t - 3) >> 2 & 1, (t - 3) >> 1 & 1, (t - 3) & 1]; first = 0
        elif t == 11:
            first, bits = 1, [1, 1, 0]      # 1110
        elif t == 22:
            first, bits = 1, [1, 1, 1]      # 1111
        else:                                # 12..21: five bits 1 0000 .. 1 1001
            v = t - 12
            first, bits = 1, [v >> 3 & 1, v >> 2 & 1, v >> 1 & 1, v & 1]
        e.decision(31, first)
        for b in bits:
            e.decision(32, b)

    def sub_mb_type_b(self, t):
        e = self.enc
        if t == 0:
            e.decision(36, 0)
            return
        e.decision(36, 1)
        if t <= 2:                       # 1 0 b
            e.decision(37, 0); e.decision(39, t - 1)
            return
        e.decision(37, 1)
        code = {3: [0, 0, 0], 4: [0, 0, 1], 5: [0, 1, 0], 6: [0, 1, 1], 11: [1, 1, 0], 12: [1, 1, 1],
                7: [1, 0, 0, 0], 8: [1, 0, 0, 1], 9: [1, 0, 1, 0], 10: [1, 0, 1, 1]}[t]
        e.decision(38, code[0])
        for b in code[1:]:
            e.decision(39, b)

    def ref_idx(self, mx, my, b8, v, cur, lst=0):
        # neighbours of the 8x8 partition's top-left 4x4
        bx, by = (b8 & 1) * 2, (b8 >> 1) * 2
        (da, ia), (db, ib) = self.blk_nb(bx, by)

        def cond(d, i, horiz):
            if d == 0:
                m = cur
            else:
                m = self.fs.get(mx - 1, my, self.sl) if horiz else self.fs.get(mx, my - 1, self.sl)
            if m is None or m.intra or m.skip or not m.inter:
                return 0
            return int(m.refgt0[lst][i >> 2])
        inc = cond(da, ia, True) + 2 * cond(db, ib, False)
        e = self.enc
        if v == 0:
            e.decision(54 + inc, 0)
            return
        e.decision(54 + inc, 1)
        for k in range(1, v):
            e.decision(54 + (4 if k == 1 else 5), 1)
        e.decision(54 + (4 if v == 1 else 5), 0)

    def mvd(self, mx, my, blk, comp, v, cur, lst=0):
        (da, ia), (db, ib) = self.blk_nb(BLK_X[blk], BLK_Y[blk])

        def amv(d, i, horiz):
            if d == 0:
                m = cur
            else:
                m = self.fs.get(mx - 1, my, self.sl) if horiz else self.fs.get(mx, my - 1, self.sl)
            if m is None or m.intra or m.skip or not m.inter:
                return 0
            return abs(m.mvd[lst][i][comp])
        s = amv(da, ia, True) + amv(db, ib, False)
        inc = 0 if s < 3 else (1 if s <= 32 else 2)
        off = 40 if comp == 0 else 47
        e, a = self.enc, abs(v)
        pre = min(a, 9)
        for k in range(pre):
            e.decision(off + (inc if k == 0 else min(2 + k, 6)), 1)
        if a < 9:
            e.decision(off + (inc if pre == 0 else min(2 + pre, 6)), 0)
        else:
            e.ueg_suffix(a - 9, 3)
        if a:
            e.bypass(int(v < 0))

    def intra_pred_modes(self, modes):
        e = self.enc
        for m in modes:
            if m < 0:
                e.decision(68, 1)
            else:
                e.decision(68, 0)
                e.decision(69, m & 1); e.decision(69, m >> 1 & 1); e.decision(69, m >> 2 & 1)

    def intra_chroma_pred_mode(self, mx, my, v):
        A, B = self.nb(mx, my)

        def cond(m):
            return int(m is not None and m.intra and m.cmode != 0)
        e = self.enc
        e.decision(64 + cond(A) + cond(B), int(v > 0))
        if v > 0:
            e.decision(67, int(v > 1))
            if v > 1:
                e.decision(67, int(v > 2))

    def coded_block_pattern(self, mx, my, cbp, cur):
        e = self.enc
        A, B = self.nb(mx, my)
        luma = cbp & 15
        for b8 in range(4):
            x, y = b8 & 1, b8 >> 1

            def cond(m, bit, internal):
                if internal:
                    return int(((luma >> bit) & 1) == 0)   # already coded bit of the current macroblock
                if m is None:
                    return 0
                if m.skip:
                    return 1
                return int(((m.cbp_l >> bit) & 1) == 0)
            ca = cond(A, b8 + 1, False) if x == 0 else cond(None, b8 - 1, True)
            cb = cond(B, b8 + 2, False) if y == 0 else cond(None, b8 - 2, True)
            e.decision(73 + ca + 2 * cb, (luma >> b8) & 1)
        chroma = cbp >> 4

        def cc(m, lvl):
            if m is None or m.skip:
                return 0
            return int(m.cbp_c >= lvl)
        e.decision(77 + cc(A, 1) + 2 * cc(B, 1), int(chroma != 0))
        if chroma:
            e.decision(77 + 4 + cc(A, 2) + 2 * cc(B, 2), int(chroma == 2))

    def mb_qp_delta(self, v):
        e = self.enc
        u = 2 * v - 1 if v > 0 else -2 * v
        e.decision(60 + int(self.prev_qpd_nz), int(u > 0))
        if u > 0:
            for k in range(1, u):
                e.decision(60 + (2 if k == 1 else 3), 1)
            e.decision(60 + (2 if u == 1 else 3), 0)
        self.prev_qpd_nz = v != 0

    def transform_8x8_flag(self, mx, my, v):
        A, B = self.nb(mx, my)
        self.enc.decision(399 + int(A is not None and A.t8) + int(B is not None and B.t8), v)

    # ---- residual -------------------------------------------------------------------------------
    def cbf_ctx_inc(self, mx, my, cat, which, cur):
        """which: luma 4x4 index for cat 1/2, plane for cat 3, (plane, blk) for cat 4, None for cat 0"""
        def term(m, getter, internal):
            if internal:
                v = getter(cur)
                return 0 if v is None else int(v)
            if m is None:
                return int(cur.intra)
            if m.skip:
                return 0
            if m.pcm:
                return 1
            v = getter(m)
            return 0 if v is None else int(v)
        A, B = self.nb(mx, my)
        if cat == 0:
            ga = gb = (lambda m: m.cbf_dc if m.i16 else None)
            return term(A, ga, False) + 2 * term(B, gb, False)
        if cat in (1, 2):
            (da, ia), (db, ib) = self.blk_nb(BLK_X[which], BLK_Y[which])

            def gl(i):
                def g(m):
                    if not (m.cbp_l >> (i >> 2)) & 1:
                        return None
                    return 1 if m.t8 else m.cbf_y[i]
                return g
            return term(A, gl(ia), da == 0) + 2 * term(B, gl(ib), db == 0)
        if cat == 3:
            g = (lambda m: m.cbf_cdc[which] if m.cbp_c else None)
            return term(A, g, False) + 2 * term(B, g, False)
        pl, b = which
        x, y = b & 1, b >> 1

        def gc(i):
            return lambda m: m.cbf_cac[pl][i] if m.cbp_c == 2 else None
        ta = term(A, gc(b + 1), False) if x == 0 else term(None, gc(b - 1), True)
        tb = term(B, gc(b + 2), False) if y == 0 else term(None, gc(b - 2), True)
        return ta + 2 * tb

    def residual_block(self, cat, coeffs, cbf_inc):
        """coeffs in scan order (length 16, 15, 4 or 64).  Returns coded_block_flag."""
        e = self.enc
        off_cbf, off_sig, off_last, off_abs = CAT[cat]
        nz = [i for i, c in enumerate(coeffs) if c]
        if cat != 5:
            e.decision(off_cbf + cbf_inc, int(bool(nz)))
        if not nz:
            return 0
        n = len(coeffs)
        for i in range(n - 1):
            if cat == 5:
                si, li = self.sig8[i], self.last8[i]
            elif cat == 3:
                si = li = min(i, 2)
            else:
                si = li = i
            s = int(coeffs[i] != 0)
            e.decision(off_sig + si, s)
            if s:
                last = int(i == nz[-1])
                e.decision(off_last + li, last)
                if last:
                    break
        eq1 = gt1 = 0
        for i in reversed(nz):
            a = abs(coeffs[i]) - 1
            ctx0 = off_abs + (0 if gt1 else min(4, 1 + eq1))
            e.decision(ctx0, int(a > 0))
            if a > 0:
                ctxn = off_abs + 5 + min(4 - int(cat == 3), gt1)
                for _ in range(1, min(a, 14)):
                    e.decision(ctxn, 1)
                if a < 14:
                    e.decision(ctxn, 0)
                else:
                    e.ueg_suffix(a - 14, 0)
                gt1 += 1
            else:
                eq1 += 1
            e.bypass(int(coeffs[i] < 0))
        return 1

    # ---- one macroblock ---------------------------------------------------------------------------
    def macroblock(self, mx, my, mb, num_ref_l0, num_ref_l1=1):
        """mb: dict as built by make_streams.Synth (an empty dict = skipped macroblock)."""
        e = self.enc
        num_ref = (num_ref_l0, num_ref_l1)
        cur = self.fs.mb[my * self.fs.W + mx]
        cur.slice = self.sl
        cur.skip, cur.intra, cur.nxn, cur.i16, cur.inter = False, False, False, False, False
        cur.cbp_l, cur.cbp_c, cur.cmode, cur.t8 = 0, 0, 0, False
        cur.cbf_y, cur.cbf_dc, cur.cbf_cdc, cur.cbf_cac = [0] * 16, 0, [0, 0], [[0] * 4, [0] * 4]
        cur.mvd, cur.refgt0 = [[(0, 0)] * 16, [(0, 0)] * 16], [[False] * 4, [False] * 4]
        cur.direct = cur.pcm = False
        skipped = "mb_type" not in mb
        if self.st != 2:
            self.mb_skip_flag(mx, my, skipped)
        if skipped:
            cur.skip = cur.direct = True
            self.prev_qpd_nz = False
            return
        t = mb["mb_type"]
        base = {2: 0, 0: 5, 1: 23}[self.st]
        intra = t >= base
        blocks = list(mb.get("coeffLevels", []))
        if intra:
            it = t - base
            cur.intra = True
            self.mb_type_intra(mx, my, it, self.st != 2)
            if it == 25:
                ps = mb["pcm_samples"]
                e.pcm_samples(list(ps["Y"]) + list(ps["Cb"]) + list(ps["Cr"]))
                cur.pcm = True
                cur.cbp_l, cur.cbp_c = 15, 2
                cur.cbf_y, cur.cbf_dc, cur.cbf_cdc, cur.cbf_cac = [1] * 16, 1, [1, 1], [[1] * 4, [1] * 4]
                self.prev_qpd_nz = False
                return
            if it == 0:
                cur.nxn = True
                t8 = int(mb.get("transform_size_8x8_flag", 0))
                if self.t8x8_mode:
                    self.transform_8x8_flag(mx, my, t8)
                cur.t8 = bool(t8)
                self.intra_pred_modes(mb.get("rem_intra8x8_pred_modes") or mb["rem_intra4x4_pred_modes"])
                cbp = mb["coded_block_pattern"]
            else:
                cur.i16 = True
                k = it - 1
                cbp = (15 if k // 12 else 0) | ((k % 12) // 4) << 4
            cur.cmode = mb["intra_chroma_pred_mode"]
            self.intra_chroma_pred_mode(mx, my, cur.cmode)
        elif self.st == 1:
            cur.inter = True
            self.mb_type_b(mx, my, t)
            self.b_motion(mx, my, mb, t, cur, num_ref)
            cbp = mb["coded_block_pattern"]
        else:
            cur.inter = True
            self.mb_type_p(t)
            mvds = list(mb.get("mvds", []))
            if t <= 2:
                parts = [[(0, range(16))], [(0, range(8)), (2, range(8, 16))],
                         [(0, [0, 1, 2, 3, 8, 9, 10, 11]), (1, [4, 5, 6, 7, 12, 13, 14, 15])]][t]
                refs = mb.get("ref_idx", {})
                cur.refgt0[0] = [False] * 4
                for b8, blks in parts:   # syntax order: ref_idx of every partition, then mvd of every partition
                    r = refs.get(str(b8), 0)
                    if num_ref_l0 > 1:
                        self.ref_idx(mx, my, b8, r, cur)
                    for q in set(i >> 2 for i in blks):
                        cur.refgt0[0][q] = r > 0
                for (b8, blks), (dx, dy) in zip(parts, mvds):
                    first = list(blks)[0]
                    self.mvd(mx, my, first, 0, dx, cur)
                    self.mvd(mx, my, first, 1, dy, cur)
                    for i in blks:
                        cur.mvd[0][i] = (dx, dy)
            else:
                subs = mb["sub_mb_types"]
                for s_ in subs:
                    self.sub_mb_type_p(s_)
                refs = mb.get("ref_idx", {})
                cur.refgt0[0] = [False] * 4
                for b8 in range(4):
                    r = refs.get(str(b8), 0)
                    if num_ref_l0 > 1:  # P_8x8ref0 (mb_type 4) does not exist with CABAC: written as P_8x8 with explicit ref_idx 0
                        self.ref_idx(mx, my, b8, r, cur)
                    cur.refgt0[0][b8] = r > 0
                it_m = iter(mvds)
                for b8, s_ in enumerate(subs):
                    shapes = {0: [[0, 1, 2, 3]], 1: [[0, 1], [2, 3]], 2: [[0, 2], [1, 3]], 3: [[0], [1], [2], [3]]}[s_]
                    for sub in shapes:
                        dx, dy = next(it_m)
                        first = b8 * 4 + sub[0]
                        self.mvd(mx, my, first, 0, dx, cur)
                        self.mvd(mx, my, first, 1, dy, cur)
                        for q in sub:
                            cur.mvd[0][b8 * 4 + q] = (dx, dy)
            cbp = mb["coded_block_pattern"]
        if not cur.i16:
            self.coded_block_pattern(mx, my, cbp, cur)
        cur.cbp_l, cur.cbp_c = cbp & 15, cbp >> 4
        if not intra and self.t8x8_mode and (cbp & 15) and "transform_size_8x8_flag" in mb:
            self.transform_8x8_flag(mx, my, int(mb["transform_size_8x8_flag"]))
            cur.t8 = bool(mb["transform_size_8x8_flag"])
        if cbp or cur.i16:
            self.mb_qp_delta(mb["mb_qp_delta"])
        else:
            self.prev_qpd_nz = False
        if not (cbp or cur.i16):
            return
        bi = iter(blocks)
        if cur.i16:
            dc = next(bi)["c"]
            cur.cbf_dc = self.residual_block(0, dc, self.cbf_ctx_inc(mx, my, 0, None, cur))
        if cur.t8:
            for b8 in range(4):
                if cbp >> b8 & 1:
                    four = [next(bi)["c"] for _ in range(4)]
                    c64 = [four[i % 4][i // 4] for i in range(64)]
                    self.residual_block(5, c64, 0)
                    for k in range(4):
                        cur.cbf_y[b8 * 4 + k] = 1
        else:
            for b in range(16):
                if cbp >> (b >> 2) & 1:
                    c = next(bi)["c"]
                    cur.cbf_y[b] = self.residual_block(1 if cur.i16 else 2, c, self.cbf_ctx_inc(mx, my, 1 if cur.i16 else 2, b, cur))
        if cbp >> 4:
            for pl in range(2):
                c = next(bi)["c"]
                cur.cbf_cdc[pl] = self.residual_block(3, c, self.cbf_ctx_inc(mx, my, 3, pl, cur))
        if cbp >> 4 == 2:
            for pl in range(2):
                for b in range(4):
                    c = next(bi)["c"]
                    cur.cbf_cac[pl][b] = self.residual_block(4, c, self.cbf_ctx_inc(mx, my, 4, (pl, b), cur))

    def b_motion(self, mx, my, mb, t, cur, num_ref):
        """ref_idx_l0, ref_idx_l1, mvd_l0, mvd_l1 of a B macroblock, in syntax order (7.3.5.1 / 7.3.5.2)."""
        if t == 0:
            cur.direct = True
            return
        refs = mb.get("ref_idx", {})
        mvds = iter(mb.get("mvds", []))
        if t <= 21:
            if t <= 3:
                parts = [(0, list(range(16)), t - 1)]
            else:
                pm = [(0, 0), (1, 1), (0, 1), (1, 0), (0, 2), (1, 2), (2, 0), (2, 1), (2, 2)][(t - 4) >> 1]
                if t & 1:   # 8x16
                    parts = [(0, [0, 1, 2, 3, 8, 9, 10, 11], pm[0]), (1, [4, 5, 6, 7, 12, 13, 14, 15], pm[1])]
                else:       # 16x8
                    parts = [(0, list(range(8)), pm[0]), (2, list(range(8, 16)), pm[1])]
            for lst in range(2):
                for b8, blks, mode in parts:
                    if mode in (lst, 2):
                        r = refs.get(str(b8 + 4 * lst), 0)
                        if num_ref[lst] > 1:
                            self.ref_idx(mx, my, b8, r, cur, lst)
                        for q in set(i >> 2 for i in blks):
                            cur.refgt0[lst][q] = r > 0
            for lst in range(2):
                for b8, blks, mode in parts:
                    if mode in (lst, 2):
                        dx, dy = next(mvds)
                        self.mvd(mx, my, blks[0], 0, dx, cur, lst)
                        self.mvd(mx, my, blks[0], 1, dy, cur, lst)
                        for i in blks:
                            cur.mvd[lst][i] = (dx, dy)
            return
        subs = mb["sub_mb_types"]
        for s_ in subs:
            self.sub_mb_type_b(s_)
        spm = {0: -1, 1: 0, 2: 1, 3: 2, 4: 0, 5: 0, 6: 1, 7: 1, 8: 2, 9: 2, 10: 0, 11: 1, 12: 2}
        shapes = {1: [[0, 1, 2, 3]], 2: [[0, 1, 2, 3]], 3: [[0, 1, 2, 3]], 4: [[0, 1], [2, 3]], 5: [[0, 2], [1, 3]],
                  6: [[0, 1], [2, 3]], 7: [[0, 2], [1, 3]], 8: [[0, 1], [2, 3]], 9: [[0, 2], [1, 3]],
                  10: [[0], [1], [2], [3]], 11: [[0], [1], [2], [3]], 12: [[0], [1], [2], [3]]}
        for lst in range(2):
            for b8, s_ in enumerate(subs):
                if spm[s_] in (lst, 2):
                    r = refs.get(str(b8 + 4 * lst), 0)
                    if num_ref[lst] > 1:
                        self.ref_idx(mx, my, b8, r, cur, lst)
                    cur.refgt0[lst][b8] = r > 0
        for lst in range(2):
            for b8, s_ in enumerate(subs):
                if spm[s_] in (lst, 2):
                    for sub in shapes[s_]:
                        dx, dy = next(mvds)
                        self.mvd(mx, my, b8 * 4 + sub[0], 0, dx, cur, lst)
                        self.mvd(mx, my, b8 * 4 + sub[0], 1, dy, cur, lst)
                        for q in sub:
                            cur.mvd[lst][b8 * 4 + q] = (dx, dy)

    def end_of_slice(self, last):
        self.enc.terminate(int(last))

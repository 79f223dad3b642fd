#!/usr/bin/env python3
"""tests/golden/make_streams.py -- generates the small Annex-B fixtures tests/golden/streams/*.264.

These are the bitstreams that drive the END-TO-END boundary tests (tests/test_frontend_*.py):
    Annex-B bytes -> reference front end (parsers, DPB, ref lists)  -> command packets -> our back end
                  -> unmodified reference decoder                   -> frames   (must be identical)

The macroblock-layer bits (CAVLC residual blocks, mb_type / sub_mb_type / mvd syntax) and the SPS/PPS
payloads are produced by the reference's own test-stream generator, /root/reference/tests/gen_avc.py,
IMPORTED here (nothing of it is copied); this script builds the seeded description of each stream
(macroblock types, prediction modes, motion vector differences, coefficient levels and the CAVLC
context nC of every residual block, which gen_avc.py expects its caller to supply), writes its own
slice header (gen_avc.py's pred_weight_table() does not follow 7.3.3.2), and applies emulation
prevention over whole NAL units.

It only runs in the build container (where /root/reference exists); the .264 outputs are committed so
the tests do not need the reference tree.  Re-run:  python tests/golden/make_streams.py
"""
import importlib.util
import io
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "streams")
GEN = "/root/reference/tests/gen_avc.py"


def load_gen():
    spec = importlib.util.spec_from_file_location("gen_avc", GEN)
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    g.escape = lambda b: b  # chunk-wise escaping can miss patterns that straddle a flush; we escape whole NALs
    return g


class AD(dict):
    """dict with attribute access (gen_avc.py reads some nested records both ways)."""
    __getattr__ = dict.__getitem__


def emulation_prevention(rbsp: bytes) -> bytes:
    out, zeros = bytearray(), 0
    for b in rbsp:
        if zeros >= 2 and b <= 3:
            out.append(3)
            zeros = 0
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def ue(bits, v):
    v += 1
    return bits << (2 * v.bit_length() - 1) | v


def se(bits, v):
    return ue(bits, 2 * v - 1 if v > 0 else -2 * v)


def u(bits, n, v):
    return bits << n | (v & ((1 << n) - 1))


# ---------------------------------------------------------------------------------------------
# per-frame CAVLC context tracking (9.2.1: nC from the TotalCoeff of the left / top blocks)
# ---------------------------------------------------------------------------------------------
BLK_X = [0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3]
BLK_Y = [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]


class FrameCtx:
    def __init__(self, W, H):
        self.W, self.H = W, H
        self.tcY = [[0] * (4 * W) for _ in range(4 * H)]
        self.tcC = [[[0] * (2 * W) for _ in range(2 * H)] for _ in range(2)]
        self.slice_of = [[-1] * W for _ in range(H)]
        self.ipm = [[2] * (4 * W) for _ in range(4 * H)]   # Intra4x4PredMode per 4x4 (2 = DC for everything else)
        self.nxn = [[False] * W for _ in range(H)]          # MB is I4x4 / I8x8

    def mb_avail(self, mx, my, sl):
        return 0 <= mx < self.W and 0 <= my < self.H and self.slice_of[my][mx] == sl

    def nC(self, grid, shift, bx, by, sl):
        a = bx > 0 and self.slice_of[by >> shift][(bx - 1) >> shift] == sl
        b = by > 0 and self.slice_of[(by - 1) >> shift][bx >> shift] == sl
        nA = grid[by][bx - 1] if a else 0
        nB = grid[by - 1][bx] if b else 0
        return (nA + nB + 1) >> 1 if a and b else nA + nB


class Synth:
    """Seeded description of one stream."""

    def __init__(self, g, name, W, H, frames, seed, *, t8x8=False, num_refs=2, weighted_pred=0, weighted_bipred=0,
                 slices=1, deblock=(0,), direct_spatial=1, scaling=False, pcm=0.03, qp=28, cqp=(0, 0), level=3.0,
                 intra_in_inter=0.12, skip=0.15, coef_density=0.35, big_levels=0.03, cbp_zero=0.0, cabac=False, tables=None, mvc=False,
                 crop=None, longterm=False, mmco_at=(), reorder=0.0, aso=False, pps_switch=False, gap_at=()):
        self.g, self.name, self.W, self.H = g, name, W, H
        self.frames, self.rng = frames, random.Random(seed)
        self.t8x8, self.num_refs, self.wp, self.wbp = t8x8, num_refs, weighted_pred, weighted_bipred
        self.slices, self.deblock, self.direct_spatial, self.scaling = slices, deblock, direct_spatial, scaling
        self.pcm, self.qp, self.cqp, self.level = pcm, qp, cqp, level
        self.intra_in_inter, self.skip, self.coef_density, self.big_levels = intra_in_inter, skip, coef_density, big_levels
        self.cbp_zero = cbp_zero
        self.cabac, self.tables, self.cabac_fs = cabac, tables, None
        self.log2_fn, self.log2_poc = 4, 6
        self.mvc = mvc   # two views (Annex H): base view + NAL 20 slices predicted from their own view and from the base picture
        self.crop = crop            # (left, right, top, bottom) luma samples: frame_cropping (1080 = 1088 - 8)
        self.longterm = longterm    # the IDR picture is marked long-term (long_term_reference_flag): every later list ends with it
        self.gap_at = gap_at        # reference frames (by count) after which frame_num skips one value: the decoder inserts a "non-existing" frame (8.2.5.2)
        self.mmco_at = mmco_at      # reference frames (by count) that carry memory_management_control_operation 1 (drop the oldest short-term)
        self.reorder = reorder      # probability that a P/B slice moves another short-term picture to the head of list 0
        self.aso = aso              # arbitrary slice order: the slices of a picture are written in shuffled order
        self.pps_switch = pps_switch  # two picture parameter sets (own chroma QP offsets + scaling matrix), alternating per picture

    # ---- parameter sets (payload bits by gen_avc.py) --------------------------------------------
    def sps(self):
        d = dict(nal_ref_idc=3, nal_unit_type=7, profile_idc=100, constraint_set_flags=[0] * 6, level_idc=self.level,
                 chroma_format_idc=1, bit_depth={"luma": 8, "chroma": 8}, qpprime_y_zero_transform_bypass_flag=0,
                 log2_max_frame_num=self.log2_fn, pic_order_cnt_type=0, log2_max_pic_order_cnt_lsb=self.log2_poc,
                 max_num_ref_frames=self.num_refs, gaps_in_frame_num_value_allowed_flag=1 if self.gap_at else 0,
                 pic_size_in_mbs={"width": self.W, "height": self.H}, frame_mbs_only_flag=1, direct_8x8_inference_flag=1)
        if self.crop:
            d["frame_crop_offsets"] = dict(zip(("left", "right", "top", "bottom"), self.crop))
        if self.scaling:
            r = self.rng
            d["seq_scaling_matrix"] = [[r.randint(6, 40) for _ in range(16)], [], [r.randint(6, 40) for _ in range(16)],
                                       [r.randint(6, 40) for _ in range(16)], [], [],
                                       [r.randint(6, 40) for _ in range(64)], [r.randint(6, 40) for _ in range(64)]]
        return d

    def subset_sps(self):
        d = self.sps()
        d.update(nal_unit_type=15, profile_idc=118, view_ids=[0, 1], num_anchor_refs={"l0": 1, "l1": 0},
                 num_non_anchor_refs={"l0": 1, "l1": 0},
                 level_values_signalled=[dict(idc=self.level, operation_points=[dict(temporal_id=0, target_views=[1], num_views=2)])])
        return d

    @staticmethod
    def mvc_ext(bits, idr, view_id, inter_view):
        # nal_unit_header_mvc_extension (H.7.3.1.1) after svc_extension_flag = 0
        bits = u(bits, 1, 0)
        bits = u(bits, 1, 0 if idr else 1)   # non_idr_flag
        bits = u(bits, 6, 0)                 # priority_id
        bits = u(bits, 10, view_id)
        bits = u(bits, 3, 0)                 # temporal_id
        bits = u(bits, 1, int(idr))          # anchor_pic_flag
        bits = u(bits, 1, inter_view)
        return u(bits, 1, 1)                 # reserved_one_bit

    def prefix_nal(self, is_ref, idr):
        bits = 1 << 1
        bits = u(bits, 2, 1 if is_ref else 0)
        bits = u(bits, 5, 14)
        bits = self.mvc_ext(bits, idr, 0, 1)
        # prefix_nal_unit_rbsp() is empty when svc_extension_flag = 0 (7.3.2.12): no trailing bits either
        return b"\0\0\0\1" + emulation_prevention((bits ^ 1 << 32).to_bytes(4, "big"))

    def pps(self, pid=0):
        d = dict(nal_ref_idc=3, nal_unit_type=8, pic_parameter_set_id=pid, entropy_coding_mode_flag=int(self.cabac),
                 bottom_field_pic_order_in_frame_present_flag=0, num_ref_idx_default_active={"l0": 1, "l1": 1},
                 weighted_pred_flag=self.wp, weighted_bipred_idc=self.wbp, pic_init_qp=self.qp,
                 chroma_qp_index_offset=self.cqp[0], deblocking_filter_control_present_flag=1,
                 constrained_intra_pred_flag=0)
        if self.t8x8 or self.cqp[1] != self.cqp[0]:
            d["transform_8x8_mode_flag"] = int(self.t8x8)
            d["second_chroma_qp_index_offset"] = self.cqp[1]
        if pid:  # the second parameter set: other chroma QP offsets and its own scaling matrix (needs the High-profile tail)
            r = random.Random(1000 + pid)
            d["chroma_qp_index_offset"] = -self.cqp[0] - 1
            d["transform_8x8_mode_flag"] = int(self.t8x8)
            d["pic_scaling_matrix"] = [[r.randint(6, 40) for _ in range(16)], [], [r.randint(6, 40) for _ in range(16)],
                                       [r.randint(6, 40) for _ in range(16)], [], []] + ([[r.randint(6, 40) for _ in range(64)], []] if self.t8x8 else [])
            d["second_chroma_qp_index_offset"] = self.cqp[1] + 2
        return d

    def nal(self, d):
        g, buf = self.g, io.BytesIO()
        ns = AD(d)
        bits = 1 << 1
        bits = u(bits, 2, ns.nal_ref_idc)
        bits = u(bits, 5, ns.nal_unit_type)
        bits = g.gen_bits[ns.nal_unit_type](bits, buf, g.SimpleNamespace(**d))
        return self.finish(bits, buf)

    @staticmethod
    def finish(bits, buf):
        bits = bits << 1 | 1
        num = bits.bit_length() - 1
        bits ^= 1 << num
        bits <<= -num % 8
        buf.write(bits.to_bytes((num + 7) // 8, "big"))
        return b"\0\0\0\1" + emulation_prevention(buf.getvalue())

    # ---- residual ---------------------------------------------------------------------------------
    def coeffs(self, n):
        r = self.rng
        c = [0] * n
        if r.random() < 0.25:
            return c
        k = min(n, 1 + int(r.expovariate(1 / (self.coef_density * 6))))
        for i in r.sample(range(n), k):
            mag = r.choice([1, 1, 1, 1, 2, 2, 3, 4])
            if r.random() < self.big_levels:
                mag = r.randint(5, 90)
            c[i] = mag if r.random() < 0.5 else -mag
        return c

    def residual(self, fc, mx, my, sl, cbp, i16, t8=False):
        blocks, r = [], self.rng
        if i16:
            blocks.append({"nC": fc.nC(fc.tcY, 2, 4 * mx, 4 * my, sl), "c": self.coeffs(16)})
        for b8 in range(4):
            if not cbp >> b8 & 1:
                continue
            cs = [self.coeffs(15 if i16 else 16) for _ in range(4)]
            if t8 and not any(v for c in cs for v in c):
                # CABAC has no coded_block_flag for an 8x8 luma block (ctxBlockCat 5, non-4:4:4): a set cbp bit MUST carry a
                # coefficient.  Done for both entropy codings so that the CAVLC twin describes the same picture.
                cs[0][0] = 1
            for b in range(4 * b8, 4 * b8 + 4):
                bx, by = 4 * mx + BLK_X[b], 4 * my + BLK_Y[b]
                c = cs[b & 3]
                blocks.append({"nC": fc.nC(fc.tcY, 2, bx, by, sl), "c": c})
                fc.tcY[by][bx] = sum(1 for v in c if v)
        if cbp >> 4:
            blocks.append({"nC": -1, "c": self.coeffs(4)})
            blocks.append({"nC": -1, "c": self.coeffs(4)})
        if cbp >> 4 == 2:
            for p in range(2):
                for b in range(4):
                    bx, by = 2 * mx + (b & 1), 2 * my + (b >> 1)
                    c = self.coeffs(15)
                    blocks.append({"nC": fc.nC(fc.tcC[p], 1, bx, by, sl), "c": c})
                    fc.tcC[p][by][bx] = sum(1 for v in c if v)
        return blocks

    def pick_cbp(self):
        r = self.rng
        if r.random() < self.cbp_zero:
            return 0
        luma = r.choice([0, 0, 15, 15, r.randint(0, 15), r.randint(0, 15)])
        return luma | r.choice([0, 0, 1, 2, 2]) << 4

    def qp_delta(self):
        return self.rng.choice([0, 0, 0, 0, 1, -1, 2, -3, 4])

    # ---- intra macroblocks ------------------------------------------------------------------------
    def intra_mb(self, fc, mx, my, sl, base):
        """base = mb_type of I_NxN in this slice type (0 I, 5 P, 23 B)."""
        r = self.rng
        left, top = fc.mb_avail(mx - 1, my, sl), fc.mb_avail(mx, my - 1, sl)
        topleft = fc.mb_avail(mx - 1, my - 1, sl)
        cmodes = [0] + ([1] if left else []) + ([2] if top else []) + ([3] if left and top and topleft else [])
        x = r.random()
        if x < self.pcm:
            for by in range(4):
                for bx in range(4):
                    fc.tcY[4 * my + by][4 * mx + bx] = 16
            for p in range(2):
                for b in range(4):
                    fc.tcC[p][2 * my + (b >> 1)][2 * mx + (b & 1)] = 16
            style = r.random()
            smp = (lambda n: [r.randint(0, 255) for _ in range(n)]) if style < 0.7 else (lambda n: [r.choice([0, 0, 1, 255])] * n)
            return {"mb_type": base + 25, "pcm_samples": AD(bits_Y=8, bits_C=8, Y=smp(256), Cb=smp(64), Cr=smp(64))}
        if x < 0.4:
            modes = [2] + ([0] if top else []) + ([1] if left else []) + ([3] if left and top and topleft else [])
            mode = r.choice(modes)
            cbp = 0 if r.random() < self.cbp_zero else r.choice([0, 15]) | r.choice([0, 1, 2]) << 4
            mb = {"mb_type": base + 1 + mode + 4 * (cbp >> 4) + 12 * (cbp & 15 == 15),
                  "intra_chroma_pred_mode": r.choice(cmodes), "mb_qp_delta": self.qp_delta()}
            mb["coeffLevels"] = self.residual(fc, mx, my, sl, cbp, True)
            return mb
        # I_NxN
        fc.nxn[my][mx] = True
        t8 = self.t8x8 and r.random() < 0.5
        mb = {"mb_type": base}
        if self.t8x8:
            mb["transform_size_8x8_flag"] = int(t8)
        rem = []
        for b in (range(4) if t8 else range(16)):
            if t8:
                bx, by, n = 4 * mx + 2 * (b & 1), 4 * my + 2 * (b >> 1), 2
            else:
                bx, by, n = 4 * mx + BLK_X[b], 4 * my + BLK_Y[b], 1
            la = (bx & 3) > 0 or left
            ta = (by & 3) > 0 or top
            tla = ((bx & 3) > 0 or left) and ((by & 3) > 0 or top) and ((bx & 3) > 0 or (by & 3) > 0 or topleft)
            # predicted mode (8.3.1.1 / 8.3.2.1); unavailable neighbour MB -> DC; non-NxN neighbour -> DC
            if not la or not ta:
                pred = 2
            else:
                def nmode(x4, y4, for_left):
                    if not fc.nxn[y4 >> 2][x4 >> 2]:
                        return 2
                    return fc.ipm[y4][x4]
                # for 8x8 blocks the standard looks at a specific 4x4 of a 4x4-coded neighbour MB:
                # A -> block 8x8idx*4+1 (row 0 of that 8x8, right column), B -> block 8x8idx*4+2 (row 1, left column)
                if t8:
                    ax, ay = bx - 1, by
                    bx2, by2 = bx, by - 1
                    pa = nmode(ax, ay, True)
                    pb = nmode(bx2, by2, False)
                else:
                    pa = nmode(bx - 1, by, True)
                    pb = nmode(bx, by - 1, False)
                pred = min(pa, pb)
            ok = [2] + ([0, 3, 7] if ta else []) + ([1, 8] if la else []) + ([4, 5, 6] if tla else [])
            want = r.choice(ok)
            rem.append(-1 if want == pred else (want if want < pred else want - 1))
            for yy in range(n):
                for xx in range(n):
                    fc.ipm[by + yy][bx + xx] = want
        mb["rem_intra8x8_pred_modes" if t8 else "rem_intra4x4_pred_modes"] = rem
        mb["intra_chroma_pred_mode"] = r.choice(cmodes)
        cbp = self.pick_cbp()
        mb["coded_block_pattern"] = cbp
        if cbp:
            mb["mb_qp_delta"] = self.qp_delta()
            mb["coeffLevels"] = self.residual(fc, mx, my, sl, cbp, False, bool(t8))
        return mb

    # ---- inter macroblocks ------------------------------------------------------------------------
    def mvd(self):
        r = self.rng
        if r.random() < 0.4:
            return (0, 0)
        m = r.choice([3, 3, 6, 6, 12, 40])
        return (r.randint(-m, m), r.randint(-m, m))

    def finish_inter(self, fc, mx, my, sl, mb, can_t8):
        cbp = self.pick_cbp()
        mb["coded_block_pattern"] = cbp
        if self.t8x8 and can_t8 and cbp & 15:
            mb["transform_size_8x8_flag"] = int(self.rng.random() < 0.5)
        if cbp:
            mb["mb_qp_delta"] = self.qp_delta()
            mb["coeffLevels"] = self.residual(fc, mx, my, sl, cbp, False, bool(mb.get("transform_size_8x8_flag")))
        return mb

    def p_mb(self, fc, mx, my, sl, nref):
        r = self.rng
        t = r.choice([0, 0, 0, 1, 2, 3, 3, 4])
        mb = {"mb_type": t}
        can_t8 = True
        if t <= 2:
            parts = [[0], [0, 2], [0, 1]][t]
            mb["ref_idx"] = {str(b): r.randrange(nref) for b in parts} if nref > 1 else {}
            mb["mvds"] = [self.mvd() for _ in parts]
        else:
            subs = [r.choice([0, 0, 1, 2, 3]) for _ in range(4)]
            mb["sub_mb_types"] = subs
            mb["ref_idx"] = {str(b): r.randrange(nref) for b in range(4)} if nref > 1 and t == 3 else {}
            mb["mvds"] = [self.mvd() for s in subs for _ in range([1, 2, 2, 4][s])]
            can_t8 = all(s == 0 for s in subs)
        return self.finish_inter(fc, mx, my, sl, mb, can_t8)

    def b_mb(self, fc, mx, my, sl, nref0, nref1):
        r = self.rng
        t = r.choice([0, 0, 1, 2, 3, 3, r.randint(4, 21), r.randint(4, 21), 22, 22])
        mb = {"mb_type": t}
        can_t8 = True
        nref = (nref0, nref1)

        def refs(parts):   # parts: list of (blk8x8, predmode 0 L0 / 1 L1 / 2 Bi)
            d = {}
            for lst in range(2):
                if nref[lst] > 1:
                    for b, pm in parts:
                        if pm == lst or pm == 2:
                            d[str(b + 4 * lst)] = r.randrange(nref[lst])
            return d
        if t == 0:
            pass
        elif t <= 3:
            parts = [(0, t - 1)]
            mb["ref_idx"] = refs(parts)
            mb["mvds"] = [self.mvd() for lst in range(2) for b, pm in parts if pm == lst or pm == 2]
        elif t <= 21:
            pm = [(0, 0), (1, 1), (0, 1), (1, 0), (0, 2), (1, 2), (2, 0), (2, 1), (2, 2)][(t - 4) >> 1]
            second = 1 if t & 1 else 2   # odd = 8x16 (blocks 0,1), even = 16x8 (blocks 0,2)
            parts = [(0, pm[0]), (second, pm[1])]
            mb["ref_idx"] = refs(parts)
            mb["mvds"] = [self.mvd() for lst in range(2) for b, p in parts if p == lst or p == 2]
        else:
            subs = [r.choice([0, 0, 1, 2, 3, 3, r.randint(4, 12)]) for _ in range(4)]
            mb["sub_mb_types"] = subs
            spm = {0: -1, 1: 0, 2: 1, 3: 2, 4: 0, 5: 0, 6: 1, 7: 1, 8: 2, 9: 2, 10: 0, 11: 1, 12: 2}
            cnt = {0: 0, 1: 1, 2: 1, 3: 1, 4: 2, 5: 2, 6: 2, 7: 2, 8: 2, 9: 2, 10: 4, 11: 4, 12: 4}
            parts = [(b, spm[s]) for b, s in enumerate(subs) if s != 0]
            mb["ref_idx"] = refs(parts)
            mb["mvds"] = [self.mvd() for lst in range(2) for b, s in enumerate(subs) if spm[s] in (lst, 2) for _ in range(cnt[s])]
            can_t8 = all(s <= 3 for s in subs)
        return self.finish_inter(fc, mx, my, sl, mb, can_t8)

    # ---- slices -----------------------------------------------------------------------------------
    def slice_nal(self, fc, ftype, first, last, sl, hdr, mbs=None):
        """hdr: dict(frame_num, poc, is_ref, idr, nref0, nref1, idr_pic_id).  mbs: the macroblocks of the slice already described (an encoder's
        decisions, nat_encoder.py) instead of the seeded random description drawn here."""
        g, r = self.g, self.rng
        st = {"I": 2, "P": 0, "B": 1}[ftype]
        given = mbs is not None
        mbs, pending = (mbs if given else []), None  # pending: entry that carries mb_skip_run
        run = 0
        for addr in (() if given else range(first, last)):
            mx, my = addr % self.W, addr // self.W
            fc.slice_of[my][mx] = sl
            if st != 2 and r.random() < self.skip:
                e = {}
                if run == 0:
                    pending = e
                run += 1
                pending["mb_skip_run"] = run
                mbs.append(e)
                continue
            if st == 2:
                mb = self.intra_mb(fc, mx, my, sl, 0)
            elif r.random() < self.intra_in_inter:
                mb = self.intra_mb(fc, mx, my, sl, 5 if st == 0 else 23)
            elif st == 0:
                mb = self.p_mb(fc, mx, my, sl, hdr["nref0"])
            else:
                mb = self.b_mb(fc, mx, my, sl, hdr["nref0"], hdr["nref1"])
            if st != 2 and run == 0:
                mb = {"mb_skip_run": 0, **mb}
            run = 0
            mbs.append(mb)
        # ---- slice header (7.3.3), written here ----
        nal_type = 20 if hdr.get("view") else 5 if hdr["idr"] else 1
        bits = 1 << 1
        bits = u(bits, 2, 1 if hdr["is_ref"] else 0)
        bits = u(bits, 5, nal_type)
        if nal_type == 20:
            bits = self.mvc_ext(bits, hdr["idr"], 1, 0)
        bits = ue(bits, first)
        bits = ue(bits, st + (5 if self.slices == 1 else 0))
        bits = ue(bits, hdr.get("pps_id", 0))  # pic_parameter_set_id
        bits = u(bits, self.log2_fn, hdr["frame_num"])
        if hdr["idr"]:
            bits = ue(bits, hdr["idr_pic_id"])
        bits = u(bits, self.log2_poc, hdr["poc"])
        if st == 1:
            bits = u(bits, 1, self.direct_spatial)
        if st != 2:
            bits = u(bits, 1, 1)  # num_ref_idx_active_override_flag
            bits = ue(bits, hdr["nref0"] - 1)
            if st == 1:
                bits = ue(bits, hdr["nref1"] - 1)
            k = hdr.get("reorder_l0", 0)
            if k:  # ref_pic_list_modification (7.3.3.1): the short-term picture CurrPicNum - k becomes entry 0
                bits = u(bits, 1, 1)
                bits = ue(bits, 0)       # modification_of_pic_nums_idc 0: subtract
                bits = ue(bits, k - 1)   # abs_diff_pic_num_minus1
                bits = ue(bits, 3)       # end
            else:
                bits = u(bits, 1, 0)  # ref_pic_list_modification_flag_l0
            if st == 1:
                bits = u(bits, 1, 0)
            if st == 0 and self.wp and hdr.get("wp_table"):  # pred_weight_table (7.3.3.2) as an encoder decided it (nat_encoder.py): P slices only
                t = hdr["wp_table"]
                bits = ue(bits, t["luma_log2_denom"])
                bits = ue(bits, t["chroma_log2_denom"])
                for e in t["l0"]:
                    bits = u(bits, 1, int(e["luma"] is not None))
                    if e["luma"] is not None:
                        bits = se(bits, e["luma"][0])
                        bits = se(bits, e["luma"][1])
                    bits = u(bits, 1, int(e["chroma"] is not None))
                    if e["chroma"] is not None:
                        for (w_, o_) in e["chroma"]:
                            bits = se(bits, w_)
                            bits = se(bits, o_)
            elif (st == 0 and self.wp) or (st == 1 and self.wbp == 1):
                ld, cd = r.randint(0, 6), r.randint(0, 6)
                bits = ue(bits, ld)
                bits = ue(bits, cd)
                for lst in range(st + 1):
                    for _ in range(hdr["nref0"] if lst == 0 else hdr["nref1"]):
                        for denom, planes in ((ld, 1), (cd, 2)):
                            flag = r.random() < 0.7
                            bits = u(bits, 1, int(flag))
                            if flag:
                                for _p in range(planes):
                                    bits = se(bits, r.choice([1 << denom, r.randint(-20, 60), r.randint(-128, 127)]))
                                    bits = se(bits, r.choice([0, r.randint(-10, 10), r.randint(-128, 127)]))
        if hdr["is_ref"]:
            if hdr["idr"]:
                bits = u(bits, 2, 1 if self.longterm else 0)  # no_output_of_prior_pics_flag, long_term_reference_flag
            elif hdr.get("mmco1"):  # dec_ref_pic_marking: drop the short-term picture CurrPicNum - mmco1
                bits = u(bits, 1, 1)  # adaptive_ref_pic_marking_mode_flag
                bits = ue(bits, 1)
                bits = ue(bits, hdr["mmco1"] - 1)  # difference_of_pic_nums_minus1
                bits = ue(bits, 0)
            else:
                bits = u(bits, 1, 0)  # adaptive_ref_pic_marking_mode_flag
        cabac_init_idc = (first + hdr["frame_num"]) % 3  # deterministic: the CAVLC and CABAC variants of a stream draw the same random sequence
        if self.cabac and st != 2:
            bits = ue(bits, cabac_init_idc)
        bits = se(bits, hdr["slice_qp_delta"])
        idc = hdr["deblock"]
        bits = ue(bits, idc)
        if idc != 1:
            bits = se(bits, hdr["alpha"])
            bits = se(bits, hdr["beta"])
        if self.cabac:
            import cabac_writer as cw
            bl = [int(c) for c in bin(bits)[3:]]           # drop the sentinel bit
            bl += [1] * (-len(bl) % 8)                     # cabac_alignment_one_bit
            cs = cw.CabacSlice(self.tables, self.cabac_fs, st, self.qp + hdr["slice_qp_delta"], cabac_init_idc, sl, self.t8x8)
            for k, addr in enumerate(range(first, last)):
                cs.macroblock(addr % self.W, addr // self.W, mbs[k], hdr["nref0"], hdr["nref1"])
                cs.end_of_slice(addr == last - 1)
            bl += cs.enc.bits                              # ends with the rbsp_stop_one_bit written by the flush
            bl += [0] * (-len(bl) % 8)
            raw = bytes(int("".join(map(str, bl[i:i + 8])), 2) for i in range(0, len(bl), 8))
            return b"\0\0\0\1" + emulation_prevention(raw)
        buf = io.BytesIO()
        ns = g.SimpleNamespace(macroblocks_cavlc=mbs, num_ref_idx_active={"l0": hdr["nref0"], "l1": hdr["nref1"]})
        bits = g.gen_slice_data_cavlc(bits, buf, ns, st)
        return self.finish(bits, buf)

    def build(self):
        r = self.rng
        out = [self.nal(self.sps())] + ([self.nal(self.subset_sps())] if self.mvc else []) + [self.nal(self.pps())]
        if self.pps_switch:
            out.append(self.nal(self.pps(1)))
        n_mbs = self.W * self.H
        frame_num, nrefs, disp = 0, 0, 0
        n_short, n_long, ref_count = 0, 0, 0  # reference pictures in the DPB after each marking step
        ne_left, after_gap = 0, False         # reference pictures until the non-existing frame slides out of the DPB; first picture after a gap
        # display order: B frames (non-reference) sit between the two reference frames decoded before them
        order, poc = [], 0
        i = 0
        while i < len(self.frames):
            t = self.frames[i]
            if t in "IP" and i + 1 < len(self.frames) and self.frames[i + 1] == "B" and i > 0:
                nb = 0
                while i + 1 + nb < len(self.frames) and self.frames[i + 1 + nb] == "B":
                    nb += 1
                order.append((t, poc + 2 * (nb + 1)))
                for k in range(nb):
                    order.append(("B", poc + 2 * (k + 1)))
                poc += 2 * (nb + 1)
                i += 1 + nb
            else:
                if i > 0:
                    poc += 2
                order.append((t, poc))
                i += 1
        for idx, (t, p) in enumerate(order):
            idr = idx == 0
            is_ref = t != "B"
            fc = FrameCtx(self.W, self.H)
            if self.cabac:
                import cabac_writer as cw
                self.cabac_fs = cw.FrameState(self.W, self.H)
            nref0 = max(1, n_short + n_long) if t != "I" else 1
            nref1 = nref0
            if t != "I":
                nref0, nref1 = r.randint(1, nref0), r.randint(1, nref1)
            if ne_left > 0 and t != "I":
                nref0 = nref1 = 1  # a conforming stream never predicts from a non-existing frame: only the newest real one is used
            bounds = sorted(r.sample(range(1, n_mbs), min(self.slices - 1, n_mbs - 1))) if self.slices > 1 else []
            bounds = [0] + bounds + [n_mbs]
            mmco1 = n_short if (is_ref and not idr and ref_count in self.mmco_at and n_short > 1) else 0  # the oldest short-term picture
            reorder = r.randint(2, n_short) if (self.reorder and t != "I" and n_short > 1 and r.random() < self.reorder) else 0
            if after_gap and t != "I":
                reorder = 2  # list 0 starts with the non-existing frame (PicNum CurrPicNum - 1): move the real one (CurrPicNum - 2) to the front
            pic_nals = []
            for s in range(len(bounds) - 1):
                hdr = dict(frame_num=frame_num % (1 << self.log2_fn), poc=p % (1 << self.log2_poc), is_ref=is_ref, idr=idr,
                           nref0=nref0, nref1=nref1, idr_pic_id=0, slice_qp_delta=r.randint(-4, 6),
                           deblock=r.choice(self.deblock), alpha=r.randint(-3, 3), beta=r.randint(-3, 3),
                           mmco1=mmco1, reorder_l0=reorder, pps_id=(idx & 1) if self.pps_switch else 0)
                pic_nals.append((self.prefix_nal(is_ref, idr) if self.mvc else b"") + self.slice_nal(fc, t, bounds[s], bounds[s + 1], s, hdr))
            if self.aso and len(pic_nals) > 1:
                r.shuffle(pic_nals)
            out += pic_nals
            if self.mvc:
                # second view of the access unit: same frame_num/POC; RefPicList0/1 = its own view's references followed by the
                # base picture of this access unit (headers.c:784-785), so an I access unit becomes P with the inter-view ref only
                fc = FrameCtx(self.W, self.H)
                if self.cabac:
                    self.cabac_fs = cw.FrameState(self.W, self.H)
                t1 = "P" if t == "I" else t
                own = min(nrefs, self.num_refs)
                v0 = r.randint(1, own + 1)
                v1 = r.randint(1, own + 1)
                for s in range(len(bounds) - 1):
                    hdr = dict(frame_num=frame_num % (1 << self.log2_fn), poc=p % (1 << self.log2_poc), is_ref=is_ref, idr=idr,
                               nref0=v0, nref1=v1, idr_pic_id=0, slice_qp_delta=r.randint(-4, 6), view=1,
                               deblock=r.choice(self.deblock), alpha=r.randint(-3, 3), beta=r.randint(-3, 3))
                    out.append(self.slice_nal(fc, t1, bounds[s], bounds[s + 1], s, hdr))
            after_gap = False
            if is_ref:
                frame_num += 1
                nrefs += 1
                ref_count += 1
                ne_left = max(0, ne_left - 1)
                if idr:
                    n_short, n_long = (0, 1) if self.longterm else (1, 0)
                else:
                    if mmco1:
                        n_short -= 1
                    elif n_short + n_long == self.num_refs and n_short > 0:
                        n_short -= 1  # sliding window (8.2.5.3)
                    n_short += 1
                if ref_count in self.gap_at:  # the next picture's frame_num skips one value: one non-existing short-term frame joins the DPB
                    frame_num += 1
                    if n_short + n_long == self.num_refs and n_short > 0:
                        n_short -= 1
                    n_short += 1
                    ne_left, after_gap = self.num_refs, True
        return b"".join(out)


STREAMS = [
    # name, W, H, frames(decode-order types; B = non-reference between the surrounding refs), seed, options
    ("i_4x4_16x16_pcm", 5, 4, "II", 1, dict(pcm=0.08)),
    ("ipp_partitions", 5, 4, "IPPP", 2, dict(num_refs=2)),
    ("ipb_spatial", 5, 4, "IPBPB", 3, dict(num_refs=2)),
    ("ipb_temporal_implicit", 5, 4, "IPBPBB", 4, dict(num_refs=3, direct_spatial=0, weighted_bipred=2)),
    ("weighted_explicit", 4, 3, "IPPBPB", 5, dict(num_refs=2, weighted_pred=1, weighted_bipred=1)),
    ("t8x8_scaling", 5, 4, "IPBP", 6, dict(t8x8=True, scaling=True, cqp=(2, -3))),
    ("t8x8_plain", 4, 4, "IPPB", 7, dict(t8x8=True, cqp=(-2, -2), pcm=0.0)),
    ("slices_deblock_idc", 6, 4, "IPBP", 8, dict(slices=3, deblock=(0, 1, 2), num_refs=2)),
    ("one_mb", 1, 1, "IPP", 9, dict(skip=0.0)),
    ("tall_narrow", 2, 7, "IPBP", 10, dict(num_refs=2, qp=36)),
    ("low_qp_big_levels", 4, 3, "IPP", 11, dict(qp=12, big_levels=0.2, coef_density=0.8)),
    ("high_qp", 4, 3, "IPB", 12, dict(qp=46, cqp=(6, 6))),
    # BASELINE geometry through the real parser: sparse residual / many skips keep the file small
    # CABAC (tests/golden/cabac_writer.py; I, P and B slices).  Each is also generated as CAVLC from the same
    # description and the two must decode to identical frames with the unmodified reference (checked in main()).
    ("cabac_i", 5, 4, "II", 31, dict(cabac=True, pcm=0.0)),
    ("cabac_ipp", 5, 4, "IPPP", 32, dict(cabac=True, pcm=0.0, num_refs=2)),
    ("cabac_t8x8_scaling", 5, 4, "IPPP", 43, dict(cabac=True, pcm=0.0, num_refs=3, t8x8=True, scaling=True, cqp=(2, -3))),
    ("cabac_slices_deblock_idc", 6, 4, "IPPP", 34, dict(cabac=True, pcm=0.0, num_refs=2, slices=3, deblock=(0, 1, 2))),
    ("cabac_weighted", 4, 3, "IPPP", 35, dict(cabac=True, pcm=0.0, num_refs=2, weighted_pred=1)),
    ("cabac_big_levels", 4, 3, "IPP", 36, dict(cabac=True, pcm=0.0, qp=12, big_levels=0.2, coef_density=0.8)),
    ("cabac_ipb_spatial", 5, 4, "IPBPB", 38, dict(cabac=True, pcm=0.0, num_refs=2)),
    ("cabac_ipb_temporal_implicit", 5, 4, "IPBPBB", 39, dict(cabac=True, pcm=0.0, num_refs=3, direct_spatial=0, weighted_bipred=2, t8x8=True)),
    ("cabac_weighted_b", 4, 3, "IPPBPB", 40, dict(cabac=True, pcm=0.0, num_refs=2, weighted_pred=1, weighted_bipred=1)),
    ("cabac_pcm", 5, 4, "IPBP", 41, dict(cabac=True, pcm=0.12, num_refs=2)),
    ("cabac_hd1080_ipp", 120, 68, "IPP", 37, dict(cabac=True, pcm=0.0, num_refs=2, level=4.0, skip=0.45, coef_density=0.12,
                                                   intra_in_inter=0.03, cbp_zero=0.8)),
    # MVC (Annex H, two views): base view + NAL 20 slices with inter-view prediction; get_frame returns samples_mvc
    ("mvc_ipp", 5, 4, "IPPP", 51, dict(mvc=True, num_refs=2, pcm=0.0)),
    ("mvc_ipb", 5, 4, "IPBPB", 52, dict(mvc=True, num_refs=2)),
    ("mvc_cabac_ipb", 5, 4, "IPBPBB", 51, dict(mvc=True, cabac=True, pcm=0.0, num_refs=3, t8x8=True, slices=2, weighted_bipred=2)),
    ("cabac_t8x8_slices", 5, 4, "IPBP", 55, dict(cabac=True, pcm=0.0, num_refs=3, t8x8=True, slices=3)),
    ("hd1080_ippb", 120, 68, "IPPB", 13, dict(num_refs=2, level=4.0, skip=0.45, coef_density=0.12, intra_in_inter=0.03, pcm=0.0005, cbp_zero=0.8)),
    # decoder-state features the packets inherit from the reference's front end: cropping, long-term references and MMCO,
    # reference list modification, arbitrary slice order, parameter-set switches between pictures
    ("crop_longterm_mmco", 6, 5, "IPPPBPPB", 61, dict(num_refs=3, crop=(2, 4, 0, 6), longterm=True, mmco_at=(3, 5))),
    ("cabac_reorder_longterm", 5, 4, "IPPPPBP", 62, dict(cabac=True, pcm=0.0, num_refs=3, longterm=True, reorder=0.8, t8x8=True)),
    ("reorder_weighted", 5, 4, "IPPPBPP", 63, dict(num_refs=4, reorder=0.9, weighted_pred=1, weighted_bipred=2)),
    ("aso_slices", 6, 5, "IPPBP", 64, dict(slices=4, aso=True, num_refs=2, deblock=(0, 2))),
    # round 5 (found by tools/stream_sweep.py): scaling lists from the parameter sets in slices whose deblocking is switched off -- an I slice with
    # disable_deblocking_filter_idc 1 issues no leaf call that sees the slice before its first residual block (with and without the 8x8 transform)
    ("scaling_idc1", 5, 4, "IIPP", 81, dict(scaling=True, deblock=(1,), slices=2, num_refs=2, pcm=0.0)),
    ("cabac_scaling_idc1_t8x8", 5, 4, "IPIP", 82, dict(cabac=True, pcm=0.0, scaling=True, t8x8=True, deblock=(1, 0), slices=2, num_refs=2)),
    # ... and pictures / slices that issue no leaf call at all: every macroblock I_PCM, deblocking off (the picture had no packet); I_PCM slices between
    # slices of other parameter sets (whose slice are they?)
    ("pcm_only_idc1", 3, 2, "IPP", 83, dict(pcm=1.0, intra_in_inter=1.0, skip=0.0, deblock=(1,), slices=2, num_refs=2)),
    ("cabac_pcm_slices_pps_switch", 3, 2, "IPBBPB", 84, dict(cabac=True, pcm=0.5, deblock=(1, 2, 0), slices=6, num_refs=2, pps_switch=True, scaling=True, t8x8=True)),
    ("pps_switch_scaling", 5, 4, "IPPBPP", 65, dict(pps_switch=True, t8x8=True, scaling=True, num_refs=2, cqp=(1, -2))),
    ("cabac_pps_switch", 5, 4, "IPPPP", 66, dict(cabac=True, pcm=0.0, pps_switch=True, num_refs=2, slices=2)),
    # frame_num gaps: "non-existing" frames enter the DPB (edge264_headers.c:1122-1144) and push real ones out of the window
    ("frame_num_gaps", 5, 4, "IPPPPPPP", 67, dict(num_refs=2, gap_at=(2, 5))),
    ("cabac_frame_num_gaps", 5, 4, "IPPPPPP", 68, dict(cabac=True, pcm=0.0, num_refs=3, gap_at=(3,), t8x8=True)),
    # BASELINE configs[2] / configs[3] geometry, 30 pictures, 1080 lines displayed of 1088 coded (crop bottom 8)
    ("hd1080_ipp30", 120, 68, "I" + "P" * 29, 71, dict(num_refs=2, level=4.0, crop=(0, 0, 0, 8), skip=0.5, coef_density=0.10, intra_in_inter=0.02,
                                                      pcm=0.0, cbp_zero=0.85)),
    ("cabac_hd1080_ibbp30", 120, 68, "I" + "PBB" * 9 + "PB", 72, dict(cabac=True, pcm=0.0, num_refs=2, level=4.0, crop=(0, 0, 0, 8), t8x8=True, scaling=True,
                                                                      weighted_bipred=2, skip=0.5, coef_density=0.10, intra_in_inter=0.02, cbp_zero=0.85)),
]


def main():
    import hashlib
    import json
    g = load_gen()
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.pyoracle import ref_decoder  # the UNMODIFIED reference decoder (oracle/_ref/libedge264_ref.so)
    ref = ref_decoder()
    sums = {}
    tables = None
    only = set(sys.argv[1:])  # names: generate just these and merge them into the existing reference_md5.json (the file also holds nat_encoder.py's streams)
    if only:
        with open(os.path.join(OUT, "reference_md5.json")) as f:
            sums = json.load(f)
    for name, W, H, frames, seed, opt in STREAMS:
        if only and name not in only:
            continue
        twin = None
        if opt.get("cabac"):
            import cabac_writer as cw
            tables = tables or cw.load_tables()
            opt = dict(opt, tables=tables)
            twin = Synth(g, name, W, H, frames, seed, **dict(opt, cabac=False)).build()
        data = Synth(g, name, W, H, frames, seed, **opt).build()
        with open(os.path.join(OUT, name + ".264"), "wb") as f:
            f.write(data)
        out, codes = ref.decode(data)
        assert len(out) == len(frames) and all(c in (0, 105, 61) for c in codes), (name, len(out), codes)
        if twin is not None:  # the CABAC writer is right iff the reference decodes both entropy codings to the same frames
            tout, _ = ref.decode(twin)
            assert len(tout) == len(out) and all((a[p] == b[p]).all() for a, b in zip(tout, out) for p in range(len(a))), name
        sums[name] = {"width_mbs": W, "height_mbs": H, "frames": frames, "nal_codes": codes, "views": len(out[0]) // 3,
                      "md5": [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in out]}
        print(f"{name}.264: {len(data)} bytes, {W}x{H} MBs, {frames}, {len(out)} frames decoded by the reference")
    with open(os.path.join(OUT, "reference_md5.json"), "w") as f:
        json.dump(sums, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""tests/golden/nat_encoder.py -- ENCODER-SHAPED fixtures: nat1080_ipp30.264 (CAVLC) and cabac_nat1080_ibbp30.264 (CABAC).

Every other fixture of tests/golden/streams is a RANDOM description (modes, vectors, levels drawn from a seeded generator).  The
reference's own correctness claim rests on encoder output (the 109 JVT conformance clips, /root/reference/README.md:103,
src/edge264_test.c:276-286), which cannot be fetched here.  This script is the part of that gap that can be closed offline: a
procedural 1080p video (a panning textured background, textured rectangles moving over it at fractional velocities, a flat region,
a noisy region, sensor noise) goes through a small but real encoder --

  * motion search on the RECONSTRUCTED (deblocked) references as the decoder will see them: full search on integer positions, then half- and
    quarter-sample refinement on the 16 interpolated planes of 8.4.2.2.1; 16x16 or four 8x8 partitions by cost; one reference per list;
  * vector prediction (8.4.1.3), P_Skip (8.4.1.1) and spatial direct prediction for B_Skip / B_Direct_16x16 (8.4.1.2.2, colZeroFlag
    from the co-located picture's motion) so that a macroblock whose best choice is the predicted motion and whose residual quantises
    to zero is SKIPPED: long skip runs, coherent vectors, bS = 0 on most edges -- the statistics real decoders see;
  * B pictures: direct, L0, L1 or bi-predicted 16x16 by cost; default weighting;
  * residual = source - prediction through the forward 4x4 integer transform, the Intra16x16 / chroma DC Hadamard transforms and a
    dead-zone quantiser at a fixed QP;
  * intra: Intra16x16 (4 modes) or Intra4x4 (7 of the 9 modes) by SAD on its own in-loop reconstruction;
  * `high=True` (cabac_nat*_high_*): the 8x8 transform chosen per inter macroblock where it describes the residual more cheaply, two reference
    pictures for P macroblocks (per 16x16 / per 8x8 quadrant), implicit weighted bi-prediction (8.4.2.3.1) in B pictures;

and is written out by the same syntax writers as the other fixtures (the reference's tests/gen_avc.py for CAVLC, cabac_writer.py for
CABAC).  The UNMODIFIED reference decoder closes the loop: after every picture the stream so far is decoded by it and ITS pictures are
the references of the next one.  The encoder's own arithmetic therefore only has to be good enough to make sensible decisions: what
the pictures ARE is whatever the reference decodes (tests/golden/streams/reference_md5.json), as for every other fixture.

It only runs in the build container (where /root/reference exists).  Re-run:  python tests/golden/nat_encoder.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_streams as ms  # noqa: E402

W_MBS, H_MBS = 120, 68
ZZ = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]  # raster index of scan position i (frame zig-zag)
QPC = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
MF = [(13107, 5243, 8066), (11916, 4660, 7490), (10082, 4194, 6554), (9362, 3647, 5825), (8192, 3355, 5243), (7282, 2893, 4559)]
VQ = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]
POSCLASS = np.array([[0, 2, 0, 2], [2, 1, 2, 1], [0, 2, 0, 2], [2, 1, 2, 1]])  # 0: both even, 1: both odd, 2: mixed
CF = np.array([[1, 1, 1, 1], [2, 1, -1, -2], [1, -1, -1, 1], [1, -2, 2, -1]])
ZIDX = [[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]]  # decoding order of the 4x4 block at [y][x]
PAD = 40  # samples of edge replication around a reference plane (search range + taps)


# ---------------------------------------------------------------------------------------------------------------------
# the video
# ---------------------------------------------------------------------------------------------------------------------
class Layer:
    """A texture that can be sampled at any sub-sample position: a sum of sinusoids (band-limited by construction)."""

    def __init__(self, rng, base, amp, n=18, fmax=0.45, chroma=(128, 128)):
        self.base, self.n = base, n
        self.f = rng.uniform(-fmax, fmax, (n, 2)) * rng.choice([0.15, 0.4, 1.0], (n, 1))
        self.ph = rng.uniform(0, 2 * np.pi, n)
        self.a = amp * rng.uniform(0.3, 1.0, n) / np.sqrt(n) * 2.2
        self.cf = rng.uniform(-0.02, 0.02, (2, 3, 2))
        self.cph = rng.uniform(0, 2 * np.pi, (2, 3))
        self.cbase = chroma

    def luma(self, xs, ys):
        v = np.full((len(ys), len(xs)), float(self.base))
        for k in range(self.n):
            v += self.a[k] * np.sin(self.f[k, 0] * xs[None, :] + self.f[k, 1] * ys[:, None] + self.ph[k])
        return v

    def chroma(self, p, xs, ys):
        v = np.full((len(ys), len(xs)), float(self.cbase[p]))
        for k in range(3):
            v += 9.0 * np.sin(self.cf[p, k, 0] * xs[None, :] + self.cf[p, k, 1] * ys[:, None] + self.cph[p, k])
        return v


class Scene:
    """frame(n) -> (Y, Cb, Cr) uint8 of W x H (4:2:0)."""

    def __init__(self, W, H, seed):
        self.W, self.H = W, H
        r = self.rng = np.random.default_rng(seed)
        k = W / 1920.0  # the rectangles below are laid out for 1920 x 1088; smaller pictures get the same scene, scaled (velocities halved at most)
        self.bg = Layer(r, 120, 38, chroma=(118, 136))
        self.bg_v = (1.25, 0.5)  # quarter-sample pan per picture
        # rectangles: x, y, w, h, vx, vy, layer, sensor-noise sigma inside
        self.objs = [
            (200, 150, 420, 300, -3.0, 1.75, Layer(r, 150, 45, chroma=(100, 160)), 0.0),
            (900, 500, 360, 260, 2.5, 0.0, Layer(r, 90, 30, chroma=(150, 110)), 0.0),
            (1400, 120, 300, 420, 0.75, -0.25, Layer(r, 170, 25, chroma=(128, 128)), 0.0),
            (600, 760, 520, 220, 6.25, 2.0, Layer(r, 110, 55, fmax=0.7, chroma=(140, 120)), 0.0),
            (60, 820, 340, 200, 0.0, 0.0, Layer(r, 60, 12, chroma=(128, 128)), 0.0),           # a static overlay
            (1100, 40, 640, 90, 0.0, 0.0, Layer(r, 200, 1.5, n=4, chroma=(120, 132)), 0.0),     # flat: skips, bS 0
            (1500, 700, 320, 300, -1.5, -0.75, Layer(r, 128, 30, chroma=(128, 128)), 5.0),     # noisy: coded everywhere
        ]
        if k != 1.0:
            v = max(k, 0.5)
            self.objs = [(x * k, y * k, w * k, h * k, vx * v, vy * v, lay, sig) for (x, y, w, h, vx, vy, lay, sig) in self.objs]

    def frame(self, n):
        W, H = self.W, self.H
        xs, ys = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
        cx, cy = np.arange(W // 2, dtype=np.float64) * 2 + 0.5, np.arange(H // 2, dtype=np.float64) * 2 + 0.5
        ox, oy = self.bg_v[0] * n, self.bg_v[1] * n
        Y = self.bg.luma(xs - ox, ys - oy)
        C = [self.bg.chroma(p, cx - ox, cy - oy) for p in range(2)]
        noise = np.full((H, W), 1.0)
        for (x, y, w, h, vx, vy, lay, sig) in self.objs:
            px, py = x + vx * n, y + vy * n
            x0, x1 = int(np.ceil(max(px, 0))), int(np.floor(min(px + w, W)))
            y0, y1 = int(np.ceil(max(py, 0))), int(np.floor(min(py + h, H)))
            if x1 <= x0 or y1 <= y0:
                continue
            Y[y0:y1, x0:x1] = lay.luma(xs[x0:x1] - px, ys[y0:y1] - py)
            if sig:
                noise[y0:y1, x0:x1] = sig
            c0x, c1x, c0y, c1y = (x0 + 1) // 2, x1 // 2, (y0 + 1) // 2, y1 // 2
            for p in range(2):
                C[p][c0y:c1y, c0x:c1x] = lay.chroma(p, cx[c0x:c1x] - px, cy[c0y:c1y] - py)
        g = np.random.default_rng(1000 + n)
        Y = Y + g.normal(0, 1, (H, W)) * noise
        C = [c + g.normal(0, 0.5, c.shape) for c in C]
        fade = getattr(self, "fade", None)  # (first picture, last picture, gain at the end): a fade towards black between the two
        if fade:
            gain = 1.0 + (fade[2] - 1.0) * min(1.0, max(0.0, (n - fade[0]) / float(fade[1] - fade[0])))
            Y = (Y - 16.0) * gain + 16.0
            C = [(c - 128.0) * gain + 128.0 for c in C]
        q = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)  # noqa: E731
        return q(Y), q(C[0]), q(C[1])


# ---------------------------------------------------------------------------------------------------------------------
# reference planes: padding, the 16 quarter-sample planes (8.4.2.2.1), chroma bilinear (8.4.2.2.2)
# ---------------------------------------------------------------------------------------------------------------------
def pad(a, p=PAD):
    return np.pad(a, p, mode="edge")


def tap6(a, axis):
    s = [np.roll(a, -k, axis=axis) for k in range(-2, 4)]
    return s[0] - 5 * s[1] + 20 * s[2] + 20 * s[3] - 5 * s[4] + s[5]


def qpel_planes(Y):
    """Y: padded luma (uint8).  Returns planes[yFrac][xFrac] (uint8, same shape; the outermost 3 samples are garbage from the rolls)."""
    G = Y.astype(np.int32)
    b1, h1 = tap6(G, 1), tap6(G, 0)
    clip = lambda a: np.clip(a, 0, 255)  # noqa: E731
    b, h = clip((b1 + 16) >> 5), clip((h1 + 16) >> 5)
    j = clip((tap6(b1, 0) + 512) >> 10)
    Gr, Gd = np.roll(G, -1, 1), np.roll(G, -1, 0)
    m, s = np.roll(h, -1, 1), np.roll(b, -1, 0)
    avg = lambda p, q: (p + q + 1) >> 1  # noqa: E731
    P = [[G, avg(G, b), b, avg(Gr, b)],
         [avg(G, h), avg(b, h), avg(b, j), avg(b, m)],
         [h, avg(h, j), j, avg(j, m)],
         [avg(Gd, h), avg(h, s), avg(j, s), avg(m, s)]]
    return np.array([[p.astype(np.uint8) for p in row] for row in P])


def luma_pred(planes, x, y, mv, w, h):
    """w x h block at (x, y) (unpadded coordinates) displaced by the quarter-sample vector mv."""
    ix, iy = x + (mv[0] >> 2) + PAD, y + (mv[1] >> 2) + PAD
    H, W = planes.shape[2:]
    ix, iy = min(max(ix, 3), W - w - 4), min(max(iy, 3), H - h - 4)  # (far outside: the padding's flat edge is what the clamp of 8.4.2.2 gives)
    return planes[mv[1] & 3, mv[0] & 3, iy:iy + h, ix:ix + w].astype(np.int32)


def chroma_pred(C, x, y, mv, w, h):
    """C: padded chroma plane; (x, y), w, h in chroma samples; mv in quarter luma = eighth chroma samples."""
    ix, iy = x + (mv[0] >> 3) + PAD, y + (mv[1] >> 3) + PAD
    H, W = C.shape
    ix, iy = min(max(ix, 0), W - w - 1), min(max(iy, 0), H - h - 1)
    xf, yf = mv[0] & 7, mv[1] & 7
    a = C[iy:iy + h + 1, ix:ix + w + 1].astype(np.int32)
    return ((8 - xf) * (8 - yf) * a[:-1, :-1] + xf * (8 - yf) * a[:-1, 1:] + (8 - xf) * yf * a[1:, :-1] + xf * yf * a[1:, 1:] + 32) >> 6


# ---------------------------------------------------------------------------------------------------------------------
# transform and quantisation (8.5; the forward side as in the JM)
# ---------------------------------------------------------------------------------------------------------------------
def fwd4x4(blocks):
    """blocks: (..., 4, 4) residual -> transform coefficients"""
    return np.einsum("ij,...jk,lk->...il", CF, blocks, CF)


def quant(Wc, qp, intra):
    qb = 15 + qp // 6
    mfv = np.array(MF[qp % 6])[POSCLASS]
    f = (1 << qb) // (3 if intra else 6)
    return np.sign(Wc) * ((np.abs(Wc) * mfv + f) >> qb)


def quant_dc(D, qp, intra):
    qb = 15 + qp // 6
    f = (1 << qb) // (3 if intra else 6)
    return np.sign(D) * ((np.abs(D) * MF[qp % 6][0] + 2 * f) >> (qb + 1))


def dequant(L, qp):
    v = np.array(VQ[qp % 6])[POSCLASS] * 16
    return (((L * v) << (qp // 6)) + 8) >> 4


def inv4x4(d):
    """d: (..., 4, 4) dequantised coefficients -> residual ((x + 32) >> 6)"""
    def one(x):  # along the last axis
        e0, e1 = x[..., 0] + x[..., 2], x[..., 0] - x[..., 2]
        e2, e3 = (x[..., 1] >> 1) - x[..., 3], x[..., 1] + (x[..., 3] >> 1)
        return np.stack([e0 + e3, e1 + e2, e1 - e2, e0 - e3], -1)
    t = one(d)                       # rows
    t = one(t.swapaxes(-1, -2)).swapaxes(-1, -2)  # columns
    return (t + 32) >> 6


# 8x8 transform (High profile): forward butterfly as in the JM, quantiser / dequantiser tables by position class (8.5.13; the classes of
# normAdjust8x8, /root/reference/src/edge264_residual.c:77-98)
Q8 = [(13107, 11428, 20972, 12222, 16777, 15481), (11916, 10826, 19174, 11058, 14980, 14290), (10082, 8943, 15978, 9675, 12710, 11985),
      (9362, 8228, 14913, 8931, 11984, 11259), (8192, 7346, 13159, 7740, 10486, 9777), (7282, 6428, 11570, 6830, 9118, 8640)]
V8 = [(20, 18, 32, 19, 25, 24), (22, 19, 35, 21, 28, 26), (26, 23, 42, 24, 33, 31), (28, 25, 45, 26, 35, 33), (32, 28, 51, 30, 40, 38), (36, 32, 58, 34, 46, 43)]


def _cls8(i, j):
    if i % 4 == 0 and j % 4 == 0:
        return 0
    if i % 2 == 1 and j % 2 == 1:
        return 1
    if i % 4 == 2 and j % 4 == 2:
        return 2
    if (i % 4 == 0 and j % 2 == 1) or (i % 2 == 1 and j % 4 == 0):
        return 3
    if (i % 4 == 0 and j % 4 == 2) or (i % 4 == 2 and j % 4 == 0):
        return 4
    return 5


CLS8 = np.array([[_cls8(i, j) for j in range(8)] for i in range(8)])
ZZ8 = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50,
       43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _fwd8_1d(x):  # along the last axis
    a0, a1, a2, a3 = x[..., 0] + x[..., 7], x[..., 1] + x[..., 6], x[..., 2] + x[..., 5], x[..., 3] + x[..., 4]
    b0, b1, b2, b3 = a0 + a3, a1 + a2, a0 - a3, a1 - a2
    a4, a5, a6, a7 = x[..., 0] - x[..., 7], x[..., 1] - x[..., 6], x[..., 2] - x[..., 5], x[..., 3] - x[..., 4]
    b4 = a5 + a6 + ((a4 >> 1) + a4)
    b5 = a4 - a7 - ((a6 >> 1) + a6)
    b6 = a4 + a7 - ((a5 >> 1) + a5)
    b7 = a5 - a6 + ((a7 >> 1) + a7)
    return np.stack([b0 + b1, b4 + (b7 >> 2), b2 + (b3 >> 1), b5 + (b6 >> 2), b0 - b1, b6 - (b5 >> 2), (b2 >> 1) - b3, -b7 + (b4 >> 2)], -1)


def _inv8_1d(d):  # along the last axis (8.5.13; the butterfly of edge264_residual.c:250-316)
    d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., k] for k in range(8))
    e0, e2 = d0 + d4, d0 - d4
    e1 = d5 - d3 - ((d7 >> 1) + d7)
    e3 = d1 + d7 - ((d3 >> 1) + d3)
    e4, e6 = (d2 >> 1) - d6, (d6 >> 1) + d2
    e5 = d7 - d1 + ((d5 >> 1) + d5)
    e7 = d3 + d5 + ((d1 >> 1) + d1)
    f0, f1, f2, f3 = e0 + e6, (e7 >> 2) + e1, e2 + e4, (e5 >> 2) + e3
    f4, f5, f6, f7 = e2 - e4, (e3 >> 2) - e5, e0 - e6, e7 - (e1 >> 2)
    return np.stack([f0 + f7, f2 + f5, f4 + f3, f6 + f1, f6 - f1, f4 - f3, f2 - f5, f0 - f7], -1)


def fwd8x8(blocks):  # (..., 8, 8)
    t = _fwd8_1d(blocks)
    return _fwd8_1d(t.swapaxes(-1, -2)).swapaxes(-1, -2)


def quant8(Wc, qp, intra):
    qb = 16 + qp // 6
    f = (1 << qb) // (3 if intra else 6)
    return np.sign(Wc) * ((np.abs(Wc) * np.array(Q8[qp % 6])[CLS8] + f) >> qb)


def dequant8(L, qp):
    v = np.array(V8[qp % 6])[CLS8] * 16
    if qp >= 36:
        return (L * v) << (qp // 6 - 6)
    return (L * v + (1 << (5 - qp // 6))) >> (6 - qp // 6)


def inv8x8(d):
    t = _inv8_1d(d)
    t = _inv8_1d(t.swapaxes(-1, -2)).swapaxes(-1, -2)
    return (t + 32) >> 6


H4 = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]])
H2 = np.array([[1, 1], [1, -1]])


def scan(block):  # (4, 4) -> 16 levels in zig-zag order
    f = block.reshape(16)
    return [int(f[i]) for i in ZZ]


# ---------------------------------------------------------------------------------------------------------------------
# motion state of a picture: per 4x4 block, both lists
# ---------------------------------------------------------------------------------------------------------------------
class Motion:
    UNAVAIL = -2

    def __init__(self, W, H):
        self.W4, self.H4 = 4 * W, 4 * H
        self.row0_4 = 0  # first row (in 4x4 blocks) of the slice being coded: rows above it belong to another slice
        self.ref = np.full((2, self.H4, self.W4), self.UNAVAIL, np.int8)
        self.mv = np.zeros((2, self.H4, self.W4, 2), np.int32)

    def get(self, lst, bx, by):
        if bx < 0 or by < self.row0_4 or bx >= self.W4 or by >= self.H4:
            return self.UNAVAIL, (0, 0)
        r = int(self.ref[lst, by, bx])
        if r < 0:
            return r, (0, 0)
        return r, (int(self.mv[lst, by, bx, 0]), int(self.mv[lst, by, bx, 1]))

    def set(self, lst, bx, by, w4, h4, ref, mv=(0, 0)):
        self.ref[lst, by:by + h4, bx:bx + w4] = ref
        self.mv[lst, by:by + h4, bx:bx + w4] = mv

    def neighbours(self, lst, bx, by, w4):
        A = self.get(lst, bx - 1, by)
        B = self.get(lst, bx, by - 1)
        Cn = self.get(lst, bx + w4, by - 1)
        if Cn[0] == self.UNAVAIL:
            Cn = self.get(lst, bx - 1, by - 1)
        return A, B, Cn

    def mvp(self, lst, bx, by, w4, ref, rule=None):
        """8.4.1.3: median prediction; rule "A" / "B" / "C": the directional prediction of a 16x8 (upper: B, lower: A) or 8x16 (left: A, right: C)
        partition -- that neighbour's vector if it uses the same reference, the median otherwise"""
        A, B, Cn = self.neighbours(lst, bx, by, w4)
        if rule is not None:
            n = {"A": A, "B": B, "C": Cn}[rule]
            if n[0] == ref:
                return n[1]
        if B[0] == self.UNAVAIL and Cn[0] == self.UNAVAIL and A[0] != self.UNAVAIL:
            B = Cn = A
        same = [n for n in (A, B, Cn) if n[0] == ref]
        if len(same) == 1:
            return same[0][1]
        med = lambda a, b, c: a + b + c - min(a, b, c) - max(a, b, c)  # noqa: E731
        return (med(A[1][0], B[1][0], Cn[1][0]), med(A[1][1], B[1][1], Cn[1][1]))

    def p_skip_mv(self, bx, by):
        A, B = self.get(0, bx - 1, by), self.get(0, bx, by - 1)
        if A[0] == self.UNAVAIL or B[0] == self.UNAVAIL or (A[0] == 0 and A[1] == (0, 0)) or (B[0] == 0 and B[1] == (0, 0)):
            return (0, 0)
        return self.mvp(0, bx, by, 4, 0)


def se_bits(v):
    k = 2 * abs(v) + (0 if v > 0 else 1) if v else 1
    return 2 * int(k).bit_length() - 1


# ---------------------------------------------------------------------------------------------------------------------
# the encoder
# ---------------------------------------------------------------------------------------------------------------------
class NatEncoder(ms.Synth):
    def __init__(self, g, name, frames, *, cabac, qp, seed, tables=None, search=8, W=W_MBS, H=H_MBS, high=False, aq=0, slice_rows=0, deblock_idc=0, fade=None, rect=False, sub=False, i8x8=False):
        """high: the High-profile tools on top -- 8x8 transform chosen per inter macroblock, two reference pictures for P macroblocks (per 16x16 /
        per 8x8), implicit weighted bi-prediction (weighted_bipred_idc 2, 8.4.2.3.1) in B pictures.
        aq: adaptive quantisation -- every macroblock's QP is the stream's plus an offset of up to +-aq from the activity of its source samples
        (flat regions finer, busy regions coarser, as rate control does): mb_qp_delta chains, edges between macroblocks of different QP.
        slice_rows: a slice every so many macroblock rows (0: one slice per picture): neighbours across the boundary are not available for
        vector / mode / context prediction, the skip runs and the QP chain start again.  deblock_idc: disable_deblocking_filter_idc of every slice.
        fade: (first, last, gain) -- the scene fades towards black; P slices then carry an explicit prediction weight table (8.4.2.3.2) estimated per
        reference from the means and spreads of the source and of the reference picture, as encoders do for fades.
        rect: P macroblocks may also split into two 16x8 or two 8x16 partitions (own search, directional vector prediction of 8.4.1.3).
        sub: the quadrants of a P_8x8 macroblock may split further into 8x4, 4x8 or 4x4 partitions (searched on their own).
        i8x8 (with high): Intra8x8 beside Intra16x16 and Intra4x4 (5 of the 9 modes, on the filtered neighbours of 8.3.2.2.1)."""
        super().__init__(g, name, W, H, frames, seed, num_refs=2 if ("B" in frames or high) else 1, cabac=cabac, tables=tables, qp=qp, level=4.0,
                         pcm=0.0, t8x8=high, weighted_bipred=2 if high else 0, slices=(-(-H // slice_rows) if slice_rows else 1), weighted_pred=1 if fade else 0)
        self.high, self.aq, self.slice_rows, self.deblock_idc, self.rect, self.sub, self.i8x8 = high, aq, slice_rows, deblock_idc, rect, sub, i8x8 and high
        self.scene = Scene(16 * W, 16 * H, seed)
        self.scene.fade = fade
        self.search = search
        self.set_mb_qp(qp)
        self.qp_prev = qp
        self.stats = {}

    def set_mb_qp(self, q):
        """the QP the macroblock under decision is quantised with"""
        self.q = q
        self.qpc = QPC[q]
        self.lam = max(2, int(0.85 * 2 ** ((q - 12) / 6)))  # SAD units per bit

    def take_qp(self):
        """mb_qp_delta of a macroblock that carries one (7.4.5: relative to the previous macroblock of the slice in decoding order)"""
        d = self.q - self.qp_prev
        self.qp_prev = self.q
        return d

    # ---- motion search of a whole picture against one reference (vectorised) ----------------------------------------------
    def search_picture(self, src, refpad, planes):
        """-> best quarter-sample vectors and SADs: mv16 (H, W, 2), sad16 (H, W), mv8 (2H, 2W, 2), sad8 (2H, 2W)"""
        Wm, Hm, R = self.W, self.H, self.search
        S = src.astype(np.int16)
        best8 = np.full((2 * Hm, 2 * Wm), 1 << 30, np.int64)
        best16 = np.full((Hm, Wm), 1 << 30, np.int64)
        mv8 = np.zeros((2 * Hm, 2 * Wm, 2), np.int32)
        mv16 = np.zeros((Hm, Wm, 2), np.int32)
        sub = getattr(self, "sub", False)
        rect = getattr(self, "rect", False) or sub
        if sub:  # 8x4 (4H x 2W blocks), 4x8 (2H x 4W), 4x4 (4H x 4W)
            best84, best48, best44 = (np.full(sh, 1 << 30, np.int64) for sh in ((4 * Hm, 2 * Wm), (2 * Hm, 4 * Wm), (4 * Hm, 4 * Wm)))
            mv84, mv48, mv44 = (np.zeros(sh + (2,), np.int32) for sh in ((4 * Hm, 2 * Wm), (2 * Hm, 4 * Wm), (4 * Hm, 4 * Wm)))
        if rect:  # 16x8 blocks (2H x W of them) and 8x16 blocks (H x 2W)
            best168, best816 = np.full((2 * Hm, Wm), 1 << 30, np.int64), np.full((Hm, 2 * Wm), 1 << 30, np.int64)
            mv168, mv816 = np.zeros((2 * Hm, Wm, 2), np.int32), np.zeros((Hm, 2 * Wm, 2), np.int32)
        Hs, Ws = S.shape
        for dy in range(-R, R + 1):
            for dx in range(-R, R + 1):
                ref = refpad[PAD + dy:PAD + dy + Hs, PAD + dx:PAD + dx + Ws].astype(np.int16)
                ad = np.abs(S - ref).astype(np.int32)
                s8 = ad.reshape(2 * Hm, 8, 2 * Wm, 8).sum((1, 3))
                s16 = s8.reshape(Hm, 2, Wm, 2).sum((1, 3))
                bias = (abs(dx) + abs(dy)) * 2  # ties go to the shorter vector: coherent fields
                m = s8 + bias < best8
                best8[m] = (s8 + bias)[m]
                mv8[m] = (4 * dx, 4 * dy)
                m = s16 + bias < best16
                best16[m] = (s16 + bias)[m]
                mv16[m] = (4 * dx, 4 * dy)
                if rect:
                    s168, s816 = s8.reshape(2 * Hm, Wm, 2).sum(2), s8.reshape(Hm, 2, 2 * Wm).sum(1)
                    m = s168 + bias < best168
                    best168[m] = (s168 + bias)[m]
                    mv168[m] = (4 * dx, 4 * dy)
                    m = s816 + bias < best816
                    best816[m] = (s816 + bias)[m]
                    mv816[m] = (4 * dx, 4 * dy)
                if sub:
                    s44 = ad.reshape(4 * Hm, 4, 4 * Wm, 4).sum((1, 3))
                    s84, s48 = s44.reshape(4 * Hm, 2 * Wm, 2).sum(2), s44.reshape(2 * Hm, 2, 4 * Wm).sum(1)
                    for (sx, bestx, mvx) in ((s44, best44, mv44), (s84, best84, mv84), (s48, best48, mv48)):
                        m = sx + bias < bestx
                        bestx[m] = (sx + bias)[m]
                        mvx[m] = (4 * dx, 4 * dy)
        # sub-sample refinement: 8 neighbours at half, then at quarter sample distance
        for (mv, best, bw, bh) in ((mv16, best16, 16, 16), (mv8, best8, 8, 8)) + (((mv168, best168, 16, 8), (mv816, best816, 8, 16)) if rect else ()) + (
                ((mv84, best84, 8, 4), (mv48, best48, 4, 8), (mv44, best44, 4, 4)) if sub else ()):
            nby, nbx = mv.shape[:2]
            yy = (np.arange(nby) * bh)[:, None, None, None] + np.arange(bh)[None, None, :, None]
            xx = (np.arange(nbx) * bw)[None, :, None, None] + np.arange(bw)[None, None, None, :]
            Sb = S.reshape(nby, bh, nbx, bw).transpose(0, 2, 1, 3).astype(np.int32)
            Hp, Wp = planes.shape[2:]

            def sad_of(v):
                iy = np.clip(yy + (v[..., 1] >> 2)[:, :, None, None] + PAD, 3, Hp - 4)
                ix = np.clip(xx + (v[..., 0] >> 2)[:, :, None, None] + PAD, 3, Wp - 4)
                pr = planes[(v[..., 1] & 3)[:, :, None, None], (v[..., 0] & 3)[:, :, None, None], iy, ix].astype(np.int32)
                return np.abs(Sb - pr).sum((2, 3))
            cur = sad_of(mv)
            for step in (2, 1):
                base = mv.copy()
                for oy in (-step, 0, step):
                    for ox in (-step, 0, step):
                        if ox == 0 and oy == 0:
                            continue
                        cand = base + np.array([ox, oy])
                        s = sad_of(cand) + 2
                        m = s < cur
                        cur[m] = s[m]
                        mv[m] = cand[m]
            best[...] = cur
        if sub:
            return mv16, best16, mv8, best8, mv168, best168, mv816, best816, mv84, best84, mv48, best48, mv44, best44
        if rect:
            return mv16, best16, mv8, best8, mv168, best168, mv816, best816
        return mv16, best16, mv8, best8

    # ---- residual of one macroblock ---------------------------------------------------------------------------------------
    def code_residual(self, src, pred, intra16=False, allow_t8=False):
        """src, pred: (Y 16x16, Cb 8x8, Cr 8x8) int arrays.  -> (luma levels (4, 4, 4, 4) [by][bx][y][x] -- or (2, 2, 8, 8) when the 8x8 transform was chosen
        (allow_t8: an inter macroblock without partitions below 8x8 in a High-profile stream; self.last_t8 says which) --, luma dc levels (4, 4) or None,
        chroma dc levels [2][(2, 2)], chroma ac levels (2, 4, 4, 4), reconstruction (Y, Cb, Cr))"""
        qp, qpc = self.q, self.qpc
        ry = (src[0] - pred[0]).reshape(4, 4, 4, 4).transpose(0, 2, 1, 3)  # [by][bx][y][x]
        Wy = fwd4x4(ry)
        self.last_t8 = False
        L8 = None
        if allow_t8:
            r8 = (src[0] - pred[0]).reshape(2, 8, 2, 8).transpose(0, 2, 1, 3)
            L8 = quant8(fwd8x8(r8), qp, False)
            L4 = quant(Wy, qp, False)
            # the cheaper description wins (levels and their magnitudes as a proxy for bits; ties to 8x8: fewer blocks to signal)
            self.last_t8 = bool((np.abs(L8).sum() + (L8 != 0).sum()) <= (np.abs(L4).sum() + (L4 != 0).sum()))
        ldc = None
        if intra16:
            dc = Wy[..., 0, 0]
            D = (H4 @ dc @ H4) // 2
            ldc = quant_dc(D, qp, True)
            Ly = quant(Wy, qp, True)
            Ly[..., 0, 0] = 0
        else:
            Ly = quant(Wy, qp, intra16 is None)
        # reconstruction
        dy = dequant(Ly, qp)
        if intra16:
            f = H4 @ ldc @ H4
            ls = (VQ[qp % 6][0] * 16) << (qp // 6)
            dy[..., 0, 0] = (f * ls + 32) >> 6
        recy = np.clip(pred[0] + inv4x4(dy).transpose(0, 2, 1, 3).reshape(16, 16), 0, 255)
        if self.last_t8:
            Ly = L8
            recy = np.clip(pred[0] + inv8x8(dequant8(L8, qp)).transpose(0, 2, 1, 3).reshape(16, 16), 0, 255)
        cdc, cac, recc = [], [], []
        for p in range(2):
            rc = (src[1 + p] - pred[1 + p]).reshape(2, 4, 2, 4).transpose(0, 2, 1, 3)
            Wc = fwd4x4(rc)
            D = H2 @ Wc[..., 0, 0] @ H2
            Ldc = quant_dc(D, qpc, intra16 is not False)
            Lac = quant(Wc, qpc, intra16 is not False)
            Lac[..., 0, 0] = 0
            dc_ = dequant(Lac, qpc)
            f = H2 @ Ldc @ H2
            ls = (VQ[qpc % 6][0] * 16) << (qpc // 6)
            dc_[..., 0, 0] = (f * ls) >> 5
            recc.append(np.clip(pred[1 + p] + inv4x4(dc_).transpose(0, 2, 1, 3).reshape(8, 8), 0, 255))
            cdc.append(Ldc)
            cac.append(Lac)
        return Ly, ldc, cdc, cac, (recy, recc[0], recc[1])

    def residual_syntax(self, fc, mx, my, sl, Ly, ldc, cdc, cac, i16, t8=False):
        """-> (coded_block_pattern, coeffLevels list in syntax order); updates the CAVLC contexts of the frame"""
        blocks = []
        # luma blocks in coding order: 8x8 quadrant b8, then its four 4x4 blocks
        order = [(ms.BLK_Y[b], ms.BLK_X[b]) for b in range(16)]
        if t8:  # Ly: (2, 2, 8, 8).  An 8x8 block is written as four 4x4 blocks holding every fourth coefficient of its zig-zag scan (7.3.5.3.2)
            cbp_l = sum(1 << b8 for b8 in range(4) if Ly[b8 >> 1, b8 & 1].any())
            for b8 in range(4):
                if not cbp_l >> b8 & 1:
                    continue
                f = Ly[b8 >> 1, b8 & 1].reshape(64)
                c64 = [int(f[i]) for i in ZZ8]
                for k in range(4):
                    c = c64[k::4]
                    by, bx = order[4 * b8 + k]
                    gx, gy = 4 * mx + bx, 4 * my + by
                    blocks.append({"nC": fc.nC(fc.tcY, 2, gx, gy, sl), "c": c})
                    fc.tcY[gy][gx] = sum(1 for v in c if v)
            Ly = np.zeros((4, 4, 4, 4), np.int64)  # (nothing below adds luma blocks)
            nz8 = [False] * 4
        else:
            nz8 = [any(Ly[order[4 * b8 + k]].any() for k in range(4)) for b8 in range(4)]
        if i16:
            cbp_l = 15 if any(nz8) else 0
        elif not t8:
            cbp_l = sum(1 << b8 for b8 in range(4) if nz8[b8])
        if i16:
            blocks.append({"nC": fc.nC(fc.tcY, 2, 4 * mx, 4 * my, sl), "c": scan(ldc)})
        for b in range(16 if not t8 else 0):
            if not cbp_l >> (b >> 2) & 1:
                continue
            by, bx = order[b]
            c = scan(Ly[by, bx])
            if i16:
                c = c[1:]
            gx, gy = 4 * mx + bx, 4 * my + by
            blocks.append({"nC": fc.nC(fc.tcY, 2, gx, gy, sl), "c": c})
            fc.tcY[gy][gx] = sum(1 for v in c if v)
        any_dc = any(d.any() for d in cdc)
        any_ac = any(a.any() for a in cac)
        cbp_c = 2 if any_ac else 1 if any_dc else 0
        if cbp_c:
            for p in range(2):
                blocks.append({"nC": -1, "c": [int(v) for v in cdc[p].reshape(4)]})
        if cbp_c == 2:
            for p in range(2):
                for b in range(4):
                    gx, gy = 2 * mx + (b & 1), 2 * my + (b >> 1)
                    c = scan(cac[p][b >> 1, b & 1])[1:]
                    blocks.append({"nC": fc.nC(fc.tcC[p], 1, gx, gy, sl), "c": c})
                    fc.tcC[p][gy][gx] = sum(1 for v in c if v)
        return cbp_l | cbp_c << 4, blocks

    # ---- intra ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def i16_preds(rec, x, y):
        """candidates {mode: 16x16 prediction} from the in-loop reconstruction (8.3.3); mode numbers of Intra16x16PredMode"""
        top = rec[y - 1, x:x + 16].astype(np.int32) if y > 0 else None
        left = rec[y:y + 16, x - 1].astype(np.int32) if x > 0 else None
        out = {}
        if top is not None:
            out[0] = np.tile(top, (16, 1))
        if left is not None:
            out[1] = np.tile(left[:, None], (1, 16))
        if top is not None and left is not None:
            dc = (top.sum() + left.sum() + 16) >> 5
        elif top is not None:
            dc = (top.sum() + 8) >> 4
        elif left is not None:
            dc = (left.sum() + 8) >> 4
        else:
            dc = 128
        out[2] = np.full((16, 16), dc, np.int32)
        if top is not None and left is not None:
            tl = int(rec[y - 1, x - 1])
            t = np.concatenate(([tl], top))
            lf = np.concatenate(([tl], left))
            Hh = sum((i + 1) * (int(t[9 + i]) - int(t[7 - i])) for i in range(8))
            V = sum((i + 1) * (int(lf[9 + i]) - int(lf[7 - i])) for i in range(8))
            a = 16 * (int(left[15]) + int(top[15]))
            b, c = (5 * Hh + 32) >> 6, (5 * V + 32) >> 6
            yy, xx = np.mgrid[0:16, 0:16]
            out[3] = np.clip((a + b * (xx - 7) + c * (yy - 7) + 16) >> 5, 0, 255)
        return out

    @staticmethod
    def chroma_preds(rec, x, y):
        """{intra_chroma_pred_mode: 8x8 prediction} (8.3.4): 0 DC, 1 horizontal, 2 vertical, 3 plane"""
        top = rec[y - 1, x:x + 8].astype(np.int32) if y > 0 else None
        left = rec[y:y + 8, x - 1].astype(np.int32) if x > 0 else None
        out = {}
        dc = np.zeros((8, 8), np.int32)
        for by in range(2):
            for bx in range(2):
                t = top[4 * bx:4 * bx + 4].sum() if top is not None else None
                lf = left[4 * by:4 * by + 4].sum() if left is not None else None
                if (bx, by) == (1, 0):
                    v = (t + 2) >> 2 if t is not None else (lf + 2) >> 2 if lf is not None else 128
                elif (bx, by) == (0, 1):
                    v = (lf + 2) >> 2 if lf is not None else (t + 2) >> 2 if t is not None else 128
                else:
                    v = (t + lf + 4) >> 3 if t is not None and lf is not None else (t + 2) >> 2 if t is not None else (lf + 2) >> 2 if lf is not None else 128
                dc[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] = v
        out[0] = dc
        if left is not None:
            out[1] = np.tile(left[:, None], (1, 8))
        if top is not None:
            out[2] = np.tile(top, (8, 1))
        if top is not None and left is not None:
            tl = int(rec[y - 1, x - 1])
            t = np.concatenate(([tl], top))
            lf = np.concatenate(([tl], left))
            Hh = sum((i + 1) * (int(t[5 + i]) - int(t[3 - i])) for i in range(4))
            V = sum((i + 1) * (int(lf[5 + i]) - int(lf[3 - i])) for i in range(4))
            a = 16 * (int(left[7]) + int(top[7]))
            b, c = (34 * Hh + 32) >> 6, (34 * V + 32) >> 6
            yy, xx = np.mgrid[0:8, 0:8]
            out[3] = np.clip((a + b * (xx - 3) + c * (yy - 3) + 16) >> 5, 0, 255)
        return out

    @staticmethod
    def i4_preds(rec, x, y, top_right_ok):
        """{Intra4x4PredMode: 4x4 prediction}: modes 0 vertical, 1 horizontal, 2 DC, 3 diagonal down-left, 4 diagonal down-right, 7 vertical-left,
        8 horizontal-up (8.3.1.2; 5 and 6 are never chosen)"""
        H, W = rec.shape
        ta, la = y > 0, x > 0
        T = rec[y - 1, x:x + 4].astype(np.int32) if ta else None
        Lf = rec[y:y + 4, x - 1].astype(np.int32) if la else None
        out = {}
        if ta:
            out[0] = np.tile(T, (4, 1))
            T8 = np.concatenate((T, rec[y - 1, x + 4:x + 8].astype(np.int32) if top_right_ok else np.full(4, T[3])))
            f3 = lambda a, i: (a[i - 1] + 2 * a[i] + a[i + 1] + 2) >> 2  # noqa: E731
            ddl = np.zeros((4, 4), np.int32)
            vl = np.zeros((4, 4), np.int32)
            for yy in range(4):
                for xx in range(4):
                    ddl[yy, xx] = (T8[6] + 3 * T8[7] + 2) >> 2 if xx == 3 and yy == 3 else (T8[xx + yy] + 2 * T8[xx + yy + 1] + T8[xx + yy + 2] + 2) >> 2
                    i = xx + (yy >> 1)
                    vl[yy, xx] = (T8[i] + T8[i + 1] + 1) >> 1 if yy % 2 == 0 else f3(T8, i + 1)
            out[3], out[7] = ddl, vl
        if la:
            out[1] = np.tile(Lf[:, None], (1, 4))
            hu = np.zeros((4, 4), np.int32)
            for yy in range(4):
                for xx in range(4):
                    z = xx + 2 * yy
                    i = yy + (xx >> 1)
                    hu[yy, xx] = Lf[3] if z > 5 else (Lf[2] + 3 * Lf[3] + 2) >> 2 if z == 5 else \
                        (Lf[i] + Lf[i + 1] + 1) >> 1 if z % 2 == 0 else (Lf[i] + 2 * Lf[i + 1] + Lf[i + 2] + 2) >> 2
            out[8] = hu
        if ta and la:
            out[2] = np.full((4, 4), (T.sum() + Lf.sum() + 4) >> 3, np.int32)
            E = np.concatenate((Lf[::-1], [int(rec[y - 1, x - 1])], T))  # L3 L2 L1 L0 corner T0..T3: index 4 = corner
            ddr = np.zeros((4, 4), np.int32)
            for yy in range(4):
                for xx in range(4):
                    i = 4 + xx - yy
                    ddr[yy, xx] = (E[i - 1] + 2 * E[i] + E[i + 1] + 2) >> 2
            out[4] = ddr
        elif ta:
            out[2] = np.full((4, 4), (T.sum() + 2) >> 2, np.int32)
        elif la:
            out[2] = np.full((4, 4), (Lf.sum() + 2) >> 2, np.int32)
        else:
            out[2] = np.full((4, 4), 128, np.int32)
        return out

    @staticmethod
    def i8_preds(rec, x, y, top_right_ok, top_left_ok):
        """{Intra8x8PredMode: 8x8 prediction} from the FILTERED neighbours (8.3.2.2.1, 8.3.2.2.2-6): 0 vertical, 1 horizontal, 2 DC, 3 diagonal
        down-left, 4 diagonal down-right (5-8 are never chosen).  rec: the in-loop reconstruction, (x, y) the block in it; what is above row 0
        or left of column 0 is not there."""
        ta, la = y > 0, x > 0
        tl = int(rec[y - 1, x - 1]) if (ta and la and top_left_ok) else None
        T = L = None
        if ta:
            t = rec[y - 1, x:x + 8].astype(np.int64)
            t = np.concatenate((t, rec[y - 1, x + 8:x + 16].astype(np.int64) if top_right_ok else np.full(8, t[7])))
            T = np.empty(16, np.int64)
            T[0] = (tl + 2 * t[0] + t[1] + 2) >> 2 if tl is not None else (3 * t[0] + t[1] + 2) >> 2
            T[1:15] = (t[0:14] + 2 * t[1:15] + t[2:16] + 2) >> 2
            T[15] = (t[14] + 3 * t[15] + 2) >> 2
        if la:
            l_ = rec[y:y + 8, x - 1].astype(np.int64)
            L = np.empty(8, np.int64)
            L[0] = (tl + 2 * l_[0] + l_[1] + 2) >> 2 if tl is not None else (3 * l_[0] + l_[1] + 2) >> 2
            L[1:7] = (l_[0:6] + 2 * l_[1:7] + l_[2:8] + 2) >> 2
            L[7] = (l_[6] + 3 * l_[7] + 2) >> 2
        out = {}
        if ta:
            out[0] = np.tile(T[:8], (8, 1))
            yy, xx = np.mgrid[0:8, 0:8]
            k = xx + yy
            ddl = (T[np.minimum(k, 15)] + 2 * T[np.minimum(k + 1, 15)] + T[np.minimum(k + 2, 15)] + 2) >> 2
            ddl[7, 7] = (T[14] + 3 * T[15] + 2) >> 2
            out[3] = ddl
        if la:
            out[1] = np.tile(L[:, None], (1, 8))
        if ta and la:
            out[2] = np.full((8, 8), (T[:8].sum() + L.sum() + 8) >> 4)
        elif ta:
            out[2] = np.full((8, 8), (T[:8].sum() + 4) >> 3)
        elif la:
            out[2] = np.full((8, 8), (L.sum() + 4) >> 3)
        else:
            out[2] = np.full((8, 8), 128)
        if tl is not None:
            TL = (int(rec[y - 1, x]) + 2 * tl + int(rec[y, x - 1]) + 2) >> 2
            e = np.concatenate((L[::-1], [TL], T[:8]))  # e[8 + d]: d = x - y; -1 -> the corner, below it the left column, above it the top row
            ddr = np.empty((8, 8), np.int64)
            for j in range(8):
                for i in range(8):
                    d = 8 + i - j
                    ddr[j, i] = (e[d - 1] + 2 * e[d] + e[d + 1] + 2) >> 2
            out[4] = ddr
        return out

    def intra_mb(self, fc, rec, src, mx, my, sl, base, inter_cost=None, row0=0):
        """Decides and codes an intra macroblock on the in-loop reconstruction `rec` (Y, Cb, Cr int16 planes, updated).  Returns (mb dict, cost)
        or None when inter_cost is given and intra is not better.  row0: first macroblock row of the slice (nothing above it is a neighbour)."""
        x, y = 16 * mx, 16 * my
        sy = src[0][y:y + 16, x:x + 16].astype(np.int32)
        sc = [src[1 + p][y // 2:y // 2 + 8, x // 2:x // 2 + 8].astype(np.int32) for p in range(2)]
        rec = [rec[0][16 * row0:], rec[1][8 * row0:], rec[2][8 * row0:]]  # views: the slice's own rows, y counted from its top from here on
        y -= 16 * row0
        c16 = self.i16_preds(rec[0], x, y)
        m16 = min(c16, key=lambda m: np.abs(sy - c16[m]).sum())
        cost16 = int(np.abs(sy - c16[m16]).sum()) + 4 * self.lam
        if inter_cost is not None and cost16 - 40 * self.lam >= inter_cost:  # (Intra4x4 never beats Intra16x16 by more than its mode bits on this content)
            return None
        cp = [self.chroma_preds(rec[1 + p], x // 2, y // 2) for p in range(2)]
        cm = min(cp[0], key=lambda m: sum(np.abs(sc[p] - cp[p][m]).sum() for p in range(2)))
        # Intra4x4: block by block on a scratch reconstruction
        scratch = rec[0][max(y - 1, 0):y + 16, :].copy()
        oy = y - max(y - 1, 0)
        cost4, modes4, lev4 = 0, [], np.zeros((4, 4, 4, 4), np.int64)
        for b in range(16):
            bx, by = ms.BLK_X[b], ms.BLK_Y[b]
            gx, gy = x + 4 * bx, oy + 4 * by
            # top right (block (bx + 1, by - 1)): in the macroblock above / above right for the first row; inside this macroblock it must precede
            # in decoding order; to the right of the macroblock it is never there
            if by == 0:
                tr = y > 0 and (bx < 3 or mx + 1 < self.W)
            else:
                tr = bx < 3 and ZIDX[by - 1][bx + 1] < b
            cands = self.i4_preds(scratch, gx, gy, bool(tr))
            sb = sy[4 * by:4 * by + 4, 4 * bx:4 * bx + 4]
            m = min(cands, key=lambda k: np.abs(sb - cands[k]).sum() + (0 if k == 2 else self.lam))
            L = quant(fwd4x4(sb - cands[m]), self.q, True)
            r = np.clip(cands[m] + inv4x4(dequant(L, self.q)), 0, 255)
            scratch[gy:gy + 4, gx:gx + 4] = r
            cost4 += int(np.abs(sb - cands[m]).sum()) + 3 * self.lam
            modes4.append(m)
            lev4[by, bx] = L
        cost8 = 1 << 60
        if self.i8x8:  # Intra8x8: four blocks on a scratch reconstruction of their own
            scr8 = rec[0][max(y - 1, 0):y + 16, :].copy()
            cost8, modes8, lev8 = 0, [], np.zeros((2, 2, 8, 8), np.int64)
            for b in range(4):
                gx, gy = x + 8 * (b & 1), oy + 8 * (b >> 1)
                tr = [y > 0, y > 0 and mx + 1 < self.W, True, False][b]
                tlk = [x > 0 and y > 0, y > 0, x > 0, True][b]
                cands = self.i8_preds(scr8, gx, gy, tr, tlk)
                sb = sy[8 * (b >> 1):8 * (b >> 1) + 8, 8 * (b & 1):8 * (b & 1) + 8]
                m = min(cands, key=lambda k: np.abs(sb - cands[k]).sum() + (0 if k == 2 else self.lam))
                L = quant8(fwd8x8(sb - cands[m]), self.q, True)
                scr8[gy:gy + 8, gx:gx + 8] = np.clip(cands[m] + inv8x8(dequant8(L, self.q)), 0, 255)
                cost8 += int(np.abs(sb - cands[m]).sum()) + 3 * self.lam
                modes8.append(m)
                lev8[b >> 1, b & 1] = L
        # (SAD favours the finer Intra4x4 prediction; what the 8x8 transform saves in bits is not in these costs: Intra8x8 is taken when its
        # prediction error is within 60 % of Intra4x4's, as encoders that weigh the rate do)
        use8 = cost8 < cost16 and (cost8 - 12 * self.lam) <= 1.6 * (cost4 - 48 * self.lam) + 48 * self.lam
        use4 = not use8 and cost4 < cost16
        cost = min(cost4, cost16, cost8)
        if inter_cost is not None and cost >= inter_cost:
            return None
        cpred = [cp[p][cm] for p in range(2)]
        if use8:
            _, _, cdc, cac, recs = self.code_residual((sy, sc[0], sc[1]), (sy, cpred[0], cpred[1]), None)
            rec[0][y:y + 16, x:x + 16] = scr8[oy:oy + 16, x:x + 16]
            rec[1][y // 2:y // 2 + 8, x // 2:x // 2 + 8], rec[2][y // 2:y // 2 + 8, x // 2:x // 2 + 8] = recs[1], recs[2]
            fc.nxn[my][mx] = True
            left, top = fc.mb_avail(mx - 1, my, sl), fc.mb_avail(mx, my - 1, sl)
            rem = []
            for b in range(4):
                bx, by = 4 * mx + 2 * (b & 1), 4 * my + 2 * (b >> 1)
                la, ta = (bx & 3) > 0 or left, (by & 3) > 0 or top
                if not la or not ta:
                    pm = 2
                else:  # (8.3.2.1: of a neighbour coded as Intra4x4 the block beside this one's first row / above its first column counts)
                    pa = fc.ipm[by][bx - 1] if fc.nxn[by >> 2][(bx - 1) >> 2] else 2
                    pb = fc.ipm[by - 1][bx] if fc.nxn[(by - 1) >> 2][bx >> 2] else 2
                    pm = min(pa, pb)
                want = modes8[b]
                rem.append(-1 if want == pm else (want if want < pm else want - 1))
                for cy in range(2):
                    for cx in range(2):
                        fc.ipm[by + cy][bx + cx] = want
            cbp, blocks = self.residual_syntax(fc, mx, my, sl, lev8, None, cdc, cac, False, t8=True)
            mb = {"mb_type": base, "transform_size_8x8_flag": 1, "rem_intra8x8_pred_modes": rem, "intra_chroma_pred_mode": cm, "coded_block_pattern": cbp}
            if cbp:
                mb.update(mb_qp_delta=self.take_qp(), coeffLevels=blocks)
            self.n_i8x8 = getattr(self, "n_i8x8", 0) + 1
            return mb, cost
        if use4:
            # chroma through the common path (its luma part is discarded: the Intra4x4 luma was coded block by block above)
            _, _, cdc, cac, recs = self.code_residual((sy, sc[0], sc[1]), (sy, cpred[0], cpred[1]), None)
            rec[0][y:y + 16, x:x + 16] = scratch[oy:oy + 16, x:x + 16]
            rec[1][y // 2:y // 2 + 8, x // 2:x // 2 + 8], rec[2][y // 2:y // 2 + 8, x // 2:x // 2 + 8] = recs[1], recs[2]
            fc.nxn[my][mx] = True
            left, top = fc.mb_avail(mx - 1, my, sl), fc.mb_avail(mx, my - 1, sl)
            rem = []
            for b in range(16):
                bx, by = 4 * mx + ms.BLK_X[b], 4 * my + ms.BLK_Y[b]
                la, ta = (bx & 3) > 0 or left, (by & 3) > 0 or top
                if not la or not ta:
                    pm = 2
                else:
                    pa = fc.ipm[by][bx - 1] if fc.nxn[by >> 2][(bx - 1) >> 2] else 2
                    pb = fc.ipm[by - 1][bx] if fc.nxn[(by - 1) >> 2][bx >> 2] else 2
                    pm = min(pa, pb)
                want = modes4[b]
                rem.append(-1 if want == pm else (want if want < pm else want - 1))
                fc.ipm[by][bx] = want
            cbp, blocks = self.residual_syntax(fc, mx, my, sl, lev4, None, cdc, cac, False)
            mb = {"mb_type": base, "rem_intra4x4_pred_modes": rem, "intra_chroma_pred_mode": cm, "coded_block_pattern": cbp}
            if self.t8x8:
                mb = {"mb_type": base, "transform_size_8x8_flag": 0, **{k: v for k, v in mb.items() if k != "mb_type"}}
            if cbp:
                mb.update(mb_qp_delta=self.take_qp(), coeffLevels=blocks)
            return mb, cost
        Ly, ldc, cdc, cac, recs = self.code_residual((sy, sc[0], sc[1]), (c16[m16], cpred[0], cpred[1]), True)
        rec[0][y:y + 16, x:x + 16] = recs[0]
        rec[1][y // 2:y // 2 + 8, x // 2:x // 2 + 8], rec[2][y // 2:y // 2 + 8, x // 2:x // 2 + 8] = recs[1], recs[2]
        cbp, blocks = self.residual_syntax(fc, mx, my, sl, Ly, ldc, cdc, cac, True)
        mb = {"mb_type": base + 1 + m16 + 4 * (cbp >> 4) + 12 * (cbp & 15 == 15), "intra_chroma_pred_mode": cm, "mb_qp_delta": self.take_qp(), "coeffLevels": blocks}
        return mb, cost

    # ---- pictures -----------------------------------------------------------------------------------------------------------
    def encode_picture(self, ptype, src, refs, col):
        """src: (Y, Cb, Cr) uint8.  refs: {list: (padded Y, padded Cb, padded Cr, planes)} of the reference decoder's pictures.
        col: Motion of RefPicList1[0] (B pictures).  -> (list of macroblock dicts with skip runs folded in, FrameCtx, Motion, counters)"""
        Wm, Hm = self.W, self.H
        fc = ms.FrameCtx(Wm, Hm)
        mot = Motion(Wm, Hm)
        rec = [np.zeros((16 * Hm, 16 * Wm), np.int32), np.zeros((8 * Hm, 8 * Wm), np.int32), np.zeros((8 * Hm, 8 * Wm), np.int32)]
        st = {"I": 2, "P": 0, "B": 1}[ptype]
        # refs: {list: [prepared reference pictures, index = refIdx]}; one motion search per (list, refIdx)
        srch = {(lst, ri): self.search_picture(src[0], r[0], r[3]) for lst, rl in refs.items() for ri, r in enumerate(rl)}
        wbi = getattr(self, "bi_weights", None)  # implicit weights (w0, w1) of the (L0[0], L1[0]) pair of this B picture, or None: (p0 + p1 + 1) >> 1
        cnt = dict(skip=0, direct=0, p16=0, p8x8=0, intra=0, bi=0, l0=0, l1=0, coded=0)
        if self.rect:
            cnt.update(p16x8=0, p8x16=0)
        out = []
        S = [p.astype(np.int32) for p in src]

        def bipred(p0, p1):
            if wbi is None:
                return (p0 + p1 + 1) >> 1
            return np.clip((p0 * wbi[0] + p1 * wbi[1] + 32) >> 6, 0, 255)  # 8.4.2.3.2 with logWD 5, no offsets

        def inter_pred(parts):
            """parts: list of (bx4, by4 (4x4 units inside the MB), size in samples, {list: mv or (mv, refIdx)}) -> (Y, Cb, Cr) prediction"""
            py, pc = np.zeros((16, 16), np.int32), [np.zeros((8, 8), np.int32), np.zeros((8, 8), np.int32)]
            for (ox, oy, sz, mvs) in parts:
                ys, cs = [], []
                sw, sh = sz if isinstance(sz, tuple) else (sz, sz)
                for lst in sorted(mvs):
                    mv, ri = mvs[lst] if isinstance(mvs[lst][0], tuple) else (mvs[lst], 0)
                    R = refs[lst][ri]
                    ys.append(luma_pred(R[3], x + ox, y + oy, mv, sw, sh))
                    cs.append([chroma_pred(R[1 + p], (x + ox) // 2, (y + oy) // 2, mv, sw // 2, sh // 2) for p in range(2)])
                    if len(R) > 4:  # explicit weights: the luma planes carry theirs already, chroma after the interpolation
                        cs[-1] = [self.weigh(cs[-1][p], *R[4][p]) for p in range(2)]
                acc_y = ys[0] if len(ys) == 1 else bipred(ys[0], ys[1])
                acc_c = cs[0] if len(cs) == 1 else [bipred(cs[0][p], cs[1][p]) for p in range(2)]
                py[oy:oy + sh, ox:ox + sw] = acc_y
                for p in range(2):
                    pc[p][oy // 2:(oy + sh) // 2, ox // 2:(ox + sw) // 2] = acc_c[p]
            return py, pc[0], pc[1]

        # adaptive quantisation: offset from the macroblock's activity (log2 of the luma variance) against the picture's mean
        if self.aq:
            var = S[0].reshape(Hm, 16, Wm, 16).transpose(0, 2, 1, 3).reshape(Hm, Wm, 256).var(axis=2)
            act = np.log2(var + 1.0)
            aq_off = np.clip(np.rint(0.9 * (act - act.mean())), -self.aq, self.aq).astype(int)
        rows = self.slice_rows or Hm
        for my in range(Hm):
            for mx in range(Wm):
                x, y = 16 * mx, 16 * my
                sl, row0 = my // rows, my // rows * rows
                if mx == 0 and my == row0:  # a slice starts: nothing above is a neighbour, the QP chain starts from the slice's QP
                    mot.row0_4 = 4 * row0
                    self.qp_prev = self.qp
                fc.slice_of[my][mx] = sl
                self.set_mb_qp(max(10, min(51, self.qp + int(aq_off[my, mx]))) if self.aq else self.qp)
                sy = S[0][y:y + 16, x:x + 16]
                srcmb = (sy, S[1][y // 2:y // 2 + 8, x // 2:x // 2 + 8], S[2][y // 2:y // 2 + 8, x // 2:x // 2 + 8])
                if st == 2:
                    mb, _ = self.intra_mb(fc, rec, src, mx, my, sl, 0, row0=row0)
                    mot.set(0, 4 * mx, 4 * my, 4, 4, -1)
                    mot.set(1, 4 * mx, 4 * my, 4, 4, -1)
                    out.append(mb)
                    cnt["intra"] += 1
                    continue
                cands = []  # (cost, kind, parts, syntax)
                if st == 0:
                    nref = len(refs[0])
                    refbits = 1 if nref > 1 else 0
                    skip_mv = mot.p_skip_mv(4 * mx, 4 * my)
                    for ri in range(nref):
                        mv16, sad16, mv8, sad8 = srch[(0, ri)][:4]
                        mvp16 = mot.mvp(0, 4 * mx, 4 * my, 4, ri)
                        best = (int(mv16[my, mx, 0]), int(mv16[my, mx, 1]))
                        for v in ({best, mvp16, skip_mv} if ri == 0 else {best, mvp16}):
                            p = luma_pred(refs[0][ri][3], x, y, v, 16, 16)
                            bits = se_bits(v[0] - mvp16[0]) + se_bits(v[1] - mvp16[1]) + refbits
                            cands.append((int(np.abs(sy - p).sum()) + self.lam * (bits if (v != skip_mv or ri) else 0), "p16", (v, ri, mvp16)))
                    # 8x8: every quadrant takes the better of its references
                    q8 = []
                    for b in range(4):
                        q8.append(min((int(srch[(0, ri)][3][2 * my + (b >> 1), 2 * mx + (b & 1)]) + self.lam * refbits, ri) for ri in range(nref)))
                    if self.sub:  # every quadrant as 8x8, two 8x4, two 4x8 or four 4x4 (one reference per quadrant): (cost, refIdx, sub_mb_type)
                        def sub_cost(b, ri, t):
                            gy, gx = 2 * my + (b >> 1), 2 * mx + (b & 1)
                            r_ = srch[(0, ri)]
                            if t == 1:
                                return int(r_[9][2 * gy, gx]) + int(r_[9][2 * gy + 1, gx]) + 11 * self.lam
                            if t == 2:
                                return int(r_[11][gy, 2 * gx]) + int(r_[11][gy, 2 * gx + 1]) + 11 * self.lam
                            return int(r_[13][2 * gy:2 * gy + 2, 2 * gx:2 * gx + 2].sum()) + 21 * self.lam
                        q8 = [min(q8[b] + (0,), *((sub_cost(b, ri, t) + self.lam * refbits, ri, t) for ri in range(nref) for t in (1, 2, 3))) for b in range(4)]
                    cands.append((6 * self.lam + sum(q[0] + 5 * self.lam for q in q8), "p8x8", None))
                    if self.rect:  # two 16x8 or two 8x16 partitions, each with the better of its references
                        h168 = [min((int(srch[(0, ri)][5][2 * my + k, mx]) + self.lam * refbits, ri) for ri in range(nref)) for k in range(2)]
                        h816 = [min((int(srch[(0, ri)][7][my, 2 * mx + k]) + self.lam * refbits, ri) for ri in range(nref)) for k in range(2)]
                        cands.append((3 * self.lam + sum(c + 5 * self.lam for c, _ in h168), "p16x8", h168))
                        cands.append((3 * self.lam + sum(c + 5 * self.lam for c, _ in h816), "p8x16", h816))
                    cost, kind, sel = min(cands, key=lambda c: c[0])
                    r = self.intra_mb(fc, rec, src, mx, my, sl, 5, inter_cost=cost, row0=row0)
                    if r is not None:
                        out.append(r[0])
                        mot.set(0, 4 * mx, 4 * my, 4, 4, -1)
                        cnt["intra"] += 1
                        continue
                    v = None
                    if kind == "p16":
                        v, ri, mvp16 = sel
                        parts = [(0, 0, 16, {0: (v, ri)})]
                        mot.set(0, 4 * mx, 4 * my, 4, 4, ri, v)
                        mb = {"mb_type": 0, "ref_idx": ({"0": ri} if nref > 1 else {}), "mvds": [(v[0] - mvp16[0], v[1] - mvp16[1])]}
                        if ri:
                            v = None  # (not the skip candidate: P_Skip predicts from reference 0)
                    elif kind in ("p16x8", "p8x16"):
                        parts, mvds, ridx = [], [], {}
                        for k in range(2):
                            ri = sel[k][1]
                            if kind == "p16x8":
                                bx4, by4, w4, h4, rule = 0, 2 * k, 4, 2, "BA"[k]
                                vv = srch[(0, ri)][4][2 * my + k, mx]
                            else:
                                bx4, by4, w4, h4, rule = 2 * k, 0, 2, 4, "AC"[k]
                                vv = srch[(0, ri)][6][my, 2 * mx + k]
                            vv = (int(vv[0]), int(vv[1]))
                            pr = mot.mvp(0, 4 * mx + bx4, 4 * my + by4, w4, ri, rule)
                            mot.set(0, 4 * mx + bx4, 4 * my + by4, w4, h4, ri, vv)
                            mvds.append((vv[0] - pr[0], vv[1] - pr[1]))
                            parts.append((4 * bx4, 4 * by4, (4 * w4, 4 * h4), {0: (vv, ri)}))
                            ridx[str((by4 >> 1) * 2 + (bx4 >> 1))] = ri
                        mb = {"mb_type": 1 if kind == "p16x8" else 2, "ref_idx": (ridx if nref > 1 else {}), "mvds": mvds}
                    else:
                        parts, mvds, ridx, subs = [], [], {}, []
                        for b in range(4):
                            bx4, by4 = 2 * (b & 1), 2 * (b >> 1)
                            ri = q8[b][1]
                            t = q8[b][2] if self.sub else 0
                            subs.append(t)
                            # the sub-macroblock partitions of the quadrant in decoding order: (x, y in 4x4 units inside the quadrant, w, h, result arrays)
                            shape = [[(0, 0, 2, 2, 2)], [(0, 0, 2, 1, 8), (0, 1, 2, 1, 8)], [(0, 0, 1, 2, 10), (1, 0, 1, 2, 10)],
                                     [(0, 0, 1, 1, 12), (1, 0, 1, 1, 12), (0, 1, 1, 1, 12), (1, 1, 1, 1, 12)]][t]
                            for (sx4, sy4, w4, h4, k) in shape:
                                gx4, gy4 = 4 * mx + bx4 + sx4, 4 * my + by4 + sy4
                                a = srch[(0, ri)][k][gy4 // h4, gx4 // w4]
                                vv = (int(a[0]), int(a[1]))
                                pr = mot.mvp(0, gx4, gy4, w4, ri)
                                mot.set(0, gx4, gy4, w4, h4, ri, vv)
                                mvds.append((vv[0] - pr[0], vv[1] - pr[1]))
                                parts.append((4 * (bx4 + sx4), 4 * (by4 + sy4), (4 * w4, 4 * h4), {0: (vv, ri)}))
                            ridx[str(b)] = ri
                        mb = {"mb_type": 3, "sub_mb_types": subs, "ref_idx": (ridx if nref > 1 else {}), "mvds": mvds}
                        if any(subs):
                            cnt["subparts"] = cnt.get("subparts", 0) + 1
                    pred = inter_pred(parts)
                    can_t8 = self.high and not (kind == "p8x8" and any(mb["sub_mb_types"]))  # (7.3.5: no 8x8 transform over partitions below 8x8)
                    Ly, _, cdc, cac, recs = self.code_residual(srcmb, pred, False, allow_t8=can_t8)
                    t8 = self.last_t8
                    zero = not Ly.any() and not any(d.any() for d in cdc) and not any(a.any() for a in cac)
                    if zero and kind == "p16" and v == skip_mv:
                        out.append(None)
                        cnt["skip"] += 1
                    else:
                        cbp, blocks = self.residual_syntax(fc, mx, my, sl, Ly, None, cdc, cac, False, t8=t8)
                        mb["coded_block_pattern"] = cbp
                        if can_t8 and cbp & 15:
                            mb["transform_size_8x8_flag"] = int(t8)
                            cnt["t8"] = cnt.get("t8", 0) + int(t8)
                        if cbp:
                            mb.update(mb_qp_delta=self.take_qp(), coeffLevels=blocks)
                            cnt["coded"] += 1
                        out.append(mb)
                        cnt[kind] += 1
                else:
                    # ---- B: spatial direct (8.4.1.2.2), L0 / L1 / Bi 16x16 ----
                    nb = [mot.neighbours(lst, 4 * mx, 4 * my, 4) for lst in range(2)]

                    def minpos(a, b):
                        return min(a, b) if a >= 0 and b >= 0 else max(a, b)
                    dref = [minpos(max(nb[l][0][0], -1), minpos(max(nb[l][1][0], -1), max(nb[l][2][0], -1))) for l in range(2)]
                    if dref[0] < 0 and dref[1] < 0:
                        dref, dmv = [0, 0], [(0, 0), (0, 0)]
                    else:
                        dmv = [mot.mvp(l, 4 * mx, 4 * my, 4, dref[l]) if dref[l] >= 0 else (0, 0) for l in range(2)]
                    dparts = []
                    for b in range(4):
                        cx4, cy4 = 4 * mx + 3 * (b & 1), 4 * my + 3 * (b >> 1)  # corner 4x4 block of the co-located macroblock (direct_8x8_inference)
                        rc, mvc = col.get(0, cx4, cy4)
                        if rc < 0:
                            rc, mvc = col.get(1, cx4, cy4)
                        colzero = rc == 0 and abs(mvc[0]) <= 1 and abs(mvc[1]) <= 1
                        mvs = {l: ((0, 0) if (colzero and dref[l] == 0) else dmv[l]) for l in range(2) if dref[l] >= 0}
                        dparts.append((8 * (b & 1), 8 * (b >> 1), 8, mvs))
                    dpred = inter_pred(dparts)
                    cands.append((int(np.abs(sy - dpred[0]).sum()), "direct", None))
                    one = {}
                    for l in range(2):
                        mv16, sad16, _, _ = srch[(l, 0)][:4]
                        pr = mot.mvp(l, 4 * mx, 4 * my, 4, 0)
                        v = (int(mv16[my, mx, 0]), int(mv16[my, mx, 1]))
                        bits = se_bits(v[0] - pr[0]) + se_bits(v[1] - pr[1])
                        one[l] = (v, pr, bits)
                        cands.append((int(sad16[my, mx]) + self.lam * (bits + 3), "l%d" % l, None))
                    pbi = bipred(luma_pred(refs[0][0][3], x, y, one[0][0], 16, 16), luma_pred(refs[1][0][3], x, y, one[1][0], 16, 16))
                    cands.append((int(np.abs(sy - pbi).sum()) + self.lam * (one[0][2] + one[1][2] + 5), "bi", None))
                    cost, kind, _ = min(cands, key=lambda c: c[0])
                    if kind == "direct":
                        parts, pred = dparts, dpred
                        for (ox, oy, sz, mvs) in dparts:
                            for l in range(2):
                                if l in mvs:
                                    mot.set(l, 4 * mx + ox // 4, 4 * my + oy // 4, 2, 2, dref[l], mvs[l])
                                else:
                                    mot.set(l, 4 * mx + ox // 4, 4 * my + oy // 4, 2, 2, -1)
                        mb = {"mb_type": 0}
                    else:
                        use = {"l0": (0,), "l1": (1,), "bi": (0, 1)}[kind]
                        mvs = {l: one[l][0] for l in use}
                        for l in range(2):
                            if l in use:
                                mot.set(l, 4 * mx, 4 * my, 4, 4, 0, one[l][0])
                            else:
                                mot.set(l, 4 * mx, 4 * my, 4, 4, -1)
                        pred = inter_pred([(0, 0, 16, mvs)])
                        mb = {"mb_type": {"l0": 1, "l1": 2, "bi": 3}[kind], "ref_idx": {},
                              "mvds": [(one[l][0][0] - one[l][1][0], one[l][0][1] - one[l][1][1]) for l in use]}
                    Ly, _, cdc, cac, recs = self.code_residual(srcmb, pred, False, allow_t8=self.high)
                    t8 = self.last_t8
                    zero = not Ly.any() and not any(d.any() for d in cdc) and not any(a.any() for a in cac)
                    if zero and kind == "direct":
                        out.append(None)
                        cnt["skip"] += 1
                    else:
                        cbp, blocks = self.residual_syntax(fc, mx, my, sl, Ly, None, cdc, cac, False, t8=t8)
                        mb["coded_block_pattern"] = cbp
                        if self.high and cbp & 15:  # (B_Direct_16x16 too: direct_8x8_inference_flag is 1)
                            mb["transform_size_8x8_flag"] = int(t8)
                            cnt["t8"] = cnt.get("t8", 0) + int(t8)
                        if cbp:
                            mb.update(mb_qp_delta=self.take_qp(), coeffLevels=blocks)
                            cnt["coded"] += 1
                        out.append(mb)
                        cnt[kind if kind != "direct" else "direct"] += 1
                rec[0][y:y + 16, x:x + 16] = recs[0]
                rec[1][y // 2:y // 2 + 8, x // 2:x // 2 + 8], rec[2][y // 2:y // 2 + 8, x // 2:x // 2 + 8] = recs[1], recs[2]
        mot.row0_4 = 0  # (as the co-located picture of a later B picture it is read everywhere)
        self.set_mb_qp(self.qp)
        # fold the skipped macroblocks into mb_skip_run entries, as Synth.slice_nal writes them; a run ends with its slice
        mbs, run, pending = [], 0, None
        for k, mb in enumerate(out):
            if k % (rows * Wm) == 0:
                run, pending = 0, None
            if st != 2 and mb is None:
                e = {}
                if run == 0:
                    pending = e
                run += 1
                pending["mb_skip_run"] = run
                mbs.append(e)
                continue
            if st != 2 and run == 0:
                mb = {"mb_skip_run": 0, **mb}
            run = 0
            mbs.append(mb)
        return mbs, fc, mot, cnt, rec

    def build(self, ref_dec):
        from make_streams import FrameCtx  # noqa: F401
        Wm, Hm = self.W, self.H
        out = [self.nal(self.sps()), self.nal(self.pps())]
        # decode order and display order (poc): B pictures sit between the two references decoded before them
        order, poc, i = [], 0, 0
        fr = self.frames
        while i < len(fr):
            t = fr[i]
            if t in "IP" and i + 1 < len(fr) and fr[i + 1] == "B" and i > 0:
                nb = 0
                while i + 1 + nb < len(fr) and fr[i + 1 + nb] == "B":
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
        frame_num = 0
        decoded = {}      # poc -> (Y, Cb, Cr) of the reference decoder
        motion = {}       # poc -> Motion
        ref_pocs = []     # reference pictures, newest first
        totals = {}
        for idx, (t, p) in enumerate(order):
            src = self.scene.frame(p // 2)
            refs = {}
            self.bi_weights = None
            wp_table = None
            if t == "P" and self.wp:
                both = [self.fade_weights(src, decoded[q]) for q in ref_pocs[:2 if self.high else 1]]
                refs[0] = [b[0] for b in both]
                wp_table = dict(luma_log2_denom=5, chroma_log2_denom=5, l0=[b[1] for b in both])
            elif t == "P":
                refs[0] = [self.prepare(decoded[q]) for q in ref_pocs[:2 if self.high else 1]]
            elif t == "B":
                before = max(q for q in ref_pocs if q < p)
                after = min(q for q in ref_pocs if q > p)
                refs[0], refs[1] = [self.prepare(decoded[before])], [self.prepare(decoded[after])]
                if self.wbp == 2:  # implicit weights, 8.4.2.3.1 (both references short-term)
                    c3 = lambda v: max(-128, min(127, v))  # noqa: E731
                    tb, td = c3(p - before), c3(after - before)
                    tx = int((16384 + abs(td // 2)) / td)
                    dsf = max(-1024, min(1023, (tb * tx + 32) >> 6))
                    w1 = dsf >> 2
                    self.bi_weights = (32, 32) if (w1 < -64 or w1 > 128) else (64 - w1, w1)
            col = motion[min(q for q in ref_pocs if q > p)] if t == "B" else None
            mbs, fc, mot, cnt, rec = self.encode_picture(t, src, refs, col)
            motion[p] = mot
            if self.cabac:
                import cabac_writer as cw
                self.cabac_fs = cw.FrameState(Wm, Hm)
            hdr = dict(frame_num=frame_num % (1 << self.log2_fn), poc=p % (1 << self.log2_poc), is_ref=t != "B", idr=idx == 0,
                       nref0=len(refs[0]) if t == "P" else 1, nref1=1,
                       idr_pic_id=0, slice_qp_delta=0, deblock=self.deblock_idc, alpha=0, beta=0, mmco1=0, reorder_l0=0, pps_id=0)
            if wp_table:
                hdr["wp_table"] = wp_table
            per = (self.slice_rows or Hm) * Wm
            n_before = len(out)
            for sl, first in enumerate(range(0, Wm * Hm, per)):
                last = min(first + per, Wm * Hm)
                out.append(self.slice_nal(fc, t, first, last, sl, hdr, mbs=mbs[first:last]))
            frames, codes = ref_dec.decode(b"".join(out))
            assert len(frames) == idx + 1 and all(c in (0, 105, 61) for c in codes), (self.name, idx, len(frames), codes[-4:])
            pocs = sorted(q for _, q in order[:idx + 1])
            got = frames[pocs.index(p)]
            decoded[p] = got
            if t != "B":
                frame_num += 1
                ref_pocs.insert(0, p)
                ref_pocs = ref_pocs[:2]
            psnr = 10 * np.log10(255.0 ** 2 / max(1e-9, np.mean((got[0].astype(np.float64) - src[0]) ** 2)))
            mine = np.mean(np.abs(rec[0] - src[0]))
            for k, v in cnt.items():
                totals[k] = totals.get(k, 0) + v
            print(f"  {self.name} picture {idx} ({t}, poc {p}): {sum(len(n) for n in out[n_before:])} bytes, luma PSNR {psnr:.2f} dB, {cnt}", flush=True)
            del mine
        if self.i8x8:
            totals["i8x8"] = getattr(self, "n_i8x8", 0)
        self.stats = totals
        return b"".join(out)

    @staticmethod
    def prepare(fr):
        Y = pad(fr[0])
        return (Y, pad(fr[1]), pad(fr[2]), qpel_planes(Y))

    @staticmethod
    def weigh(a, w, o, ld):
        """8.4.2.3.2, one list: Clip1(((pred * w + 2^(logWD - 1)) >> logWD) + o)"""
        a = a.astype(np.int32) * w
        return np.clip((((a + (1 << (ld - 1))) >> ld) if ld else a) + o, 0, 255)

    def fade_weights(self, src, ref, ld=5, cd=5):
        """(prepared reference with its luma planes weighted + the chroma weights for inter_pred, entry of the slice header's table) for one reference
        picture `ref` (Y, Cb, Cr of the reference decoder) of the source picture `src`"""
        sy, ry = src[0].astype(np.float64), ref[0].astype(np.float64)
        w = int(np.clip(np.rint((1 << ld) * sy.std() / max(ry.std(), 1e-3)), 1, 127))
        o = int(np.clip(np.rint(sy.mean() - w / float(1 << ld) * ry.mean()), -128, 127))
        cw = []
        for p in range(2):
            sc, rc = src[1 + p].astype(np.float64), ref[1 + p].astype(np.float64)
            wc = int(np.clip(np.rint((1 << cd) * sc.std() / max(rc.std(), 1e-3)), 1, 127))
            oc = int(np.clip(np.rint(sc.mean() - wc / float(1 << cd) * rc.mean()), -128, 127))
            cw.append((wc, oc, cd))
        R = self.prepare(ref)
        Rw = (self.weigh(R[0], w, o, ld).astype(np.uint8), R[1], R[2], self.weigh(R[3], w, o, ld).astype(np.uint8), cw)
        luma_flag = (w, o) != (1 << ld, 0)
        chroma_flag = any((wc, oc) != (1 << cd, 0) for wc, oc, _ in cw)
        return Rw, dict(luma=(w, o) if luma_flag else None, chroma=[(wc, oc) for wc, oc, _ in cw] if chroma_flag else None)


STREAMS = [
    ("nat1080_ipp30", "I" + "P" * 29, dict(cabac=False, qp=30, seed=7)),
    ("cabac_nat1080_ibbp30", "I" + "PBB" * 9 + "PB", dict(cabac=True, qp=31, seed=8)),
    # the same encoder on a 320 x 192 picture: small enough for the damaged-stream scenarios (tests/damage.py), whose concealment (erroneous
    # macroblocks decoded again as P_Skip / B_Skip, src/edge264_headers.c:295-430) meets coherent vector fields and skip runs here
    ("nat_small_ipp8", "I" + "P" * 7, dict(cabac=False, qp=28, seed=9, W=20, H=12)),
    ("cabac_nat_small_ibbp10", "IPBBPBBPBB", dict(cabac=True, qp=29, seed=10, W=20, H=12)),
    # High-profile tools on the same scene: 8x8 transform by choice, two references for P macroblocks, implicit weighted bi-prediction
    ("cabac_nat_small_high_ibbp10", "IPBBPBBPBB", dict(cabac=True, qp=29, seed=11, W=20, H=12, high=True)),
    ("cabac_nat1080_high_ibbp30", "I" + "PBB" * 9 + "PB", dict(cabac=True, qp=31, seed=12, high=True)),
    # rate-control-shaped: adaptive quantisation (QP 24..32 by activity: mb_qp_delta chains, deblocking edges between different QPs) and several
    # slices per picture (every 4 / 5 macroblock rows; the CABAC one with disable_deblocking_filter_idc 2: slice edges left unfiltered)
    ("nat_small_aq_slices_ipp8", "I" + "P" * 7, dict(cabac=False, qp=28, seed=13, W=20, H=12, aq=4, slice_rows=4)),
    ("cabac_nat_small_aq_slices_ibbp10", "IPBBPBBPBB", dict(cabac=True, qp=29, seed=14, W=20, H=12, high=True, aq=4, slice_rows=5, deblock_idc=2)),
    # a fade towards black: P slices with an explicit prediction weight table estimated per reference (the CABAC one with two references, each with
    # its own weights and offsets)
    ("nat_small_fade_wp_ipp8", "I" + "P" * 7, dict(cabac=False, qp=28, seed=16, W=20, H=12, fade=(0, 8, 0.25))),
    ("cabac_nat_small_fade_wp_ipp8", "I" + "P" * 7, dict(cabac=True, qp=28, seed=17, W=20, H=12, high=True, fade=(0, 8, 0.3))),
    # rectangular partitions in P pictures
    ("nat_small_rect_ipp8", "I" + "P" * 7, dict(cabac=False, qp=27, seed=18, W=20, H=12, rect=True)),
    ("cabac_nat_small_rect_ipp8", "I" + "P" * 7, dict(cabac=True, qp=27, seed=19, W=20, H=12, high=True, rect=True, aq=3)),
    # ... and partitions below 8x8 (textured objects over a moving background: the small blocks sit on their borders)
    ("nat_small_sub_ipp8", "I" + "P" * 7, dict(cabac=False, qp=24, seed=20, W=20, H=12, rect=True, sub=True)),
    ("cabac_nat_small_sub_ipp8", "I" + "P" * 7, dict(cabac=True, qp=24, seed=21, W=20, H=12, high=True, rect=True, sub=True)),
    # Intra8x8 by choice (smooth textures take it): two I pictures and the intra macroblocks of the P pictures
    ("nat_small_i8x8_iipp6", "IIPPPP", dict(cabac=False, qp=26, seed=22, W=20, H=12, high=True, i8x8=True)),
    ("cabac_nat_small_i8x8_iipp6", "IIPPPP", dict(cabac=True, qp=30, seed=23, W=20, H=12, high=True, i8x8=True, aq=3)),
    ("cabac_nat1080_aq_slices_ibbp12", "IPBBPBBPBBPB", dict(cabac=True, qp=31, seed=15, high=True, aq=4, slice_rows=17)),  # four slices of 17 rows
]


def main(argv):
    g = ms.load_gen()
    from oracle.pyoracle import ref_decoder
    ref = ref_decoder()
    n_frames = int(argv[1]) if len(argv) > 1 and argv[1].isdigit() else None
    only = [a for a in argv[1:] if not a.isdigit()]  # stream names: regenerate just these
    path = os.path.join(ms.OUT, "reference_md5.json")
    sums = json.load(open(path))
    tables = None
    for name, frames, opt in STREAMS:
        if only and name not in only:
            continue
        if n_frames:
            frames = frames[:n_frames]
        if opt["cabac"]:
            import cabac_writer as cw
            tables = tables or cw.load_tables()
            opt = dict(opt, tables=tables)
        enc = NatEncoder(g, name, frames, **opt)
        data = enc.build(ref)
        out, codes = ref.decode(data)
        assert len(out) == len(frames) and all(c in (0, 105, 61) for c in codes), (name, len(out), codes)
        if not n_frames:
            with open(os.path.join(ms.OUT, name + ".264"), "wb") as f:
                f.write(data)
            sums[name] = {"width_mbs": enc.W, "height_mbs": enc.H, "frames": frames, "nal_codes": codes, "views": 1,
                          "md5": [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in out], "encoder_stats": enc.stats}
        print(f"{name}.264: {len(data)} bytes ({len(data) * 8 * 30 / len(frames) / 1e6:.2f} Mbit/s at 30 pictures/s), {enc.stats}")
    if not n_frames:
        with open(path, "w") as f:
            json.dump(sums, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main(sys.argv))

#!/usr/bin/env python3
"""tools/dbk_sparse_sim.py -- MEASURING AID (host only): how many lock-step steps would a deblocking wave save if every macroblock row of its
group kept its OWN cursor and walked only the macroblocks that need anything (VERDICT r5 item 1a + 1b)?

A macroblock is "touched" when one of its own boundary strengths is non-zero, or its right neighbour's left edge or its lower neighbour's top
edge is (then its samples change although its own record is empty).  e264_deblock2_kernel walks a group of 8 luma (15 chroma) rows in lock
step, row g one macroblock behind row g - 1: wm + rows + 3 steps per group whatever the content.  The simulated alternative lets each row
jump to its next touched macroblock as soon as the row above is two macroblocks past it (the raster dependency of edge264_deblock.c's
order (x + 1, y - 1) before (x, y)), one macroblock per row and step: the steps a group then needs are what the densest row and the
dependencies leave.  Boundary strengths from the CPU oracle (checker-side code: this is a tool, not the product).

    python tools/dbk_sparse_sim.py tests/golden/streams/nat1080_ipp30.264 tests/golden/streams/cabac_nat1080_ibbp30.264 tests/golden/streams/hd1080_ipp30.264
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edge264_amd import front, packet as P  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402


def sim(touched, rows, lag=2):
    H, W = touched.shape
    new = old = 0
    for y0 in range(0, H, rows):
        g = touched[y0:y0 + rows]
        R = g.shape[0]
        lists = [list(np.nonzero(g[r])[0]) for r in range(R)]
        pos = [0] * R

        def nextx(r):
            return lists[r][pos[r]] if pos[r] < len(lists[r]) else W + 10
        steps = 0
        while any(pos[r] < len(lists[r]) for r in range(R)):
            adv = [r for r in range(R) if pos[r] < len(lists[r]) and (r == 0 or nextx(r - 1) >= lists[r][pos[r]] + lag)]
            for r in adv:
                pos[r] += 1
            steps += 1
        new += steps
        old += W + R - 1 + 4
    return old, new


def lockstep_live(touched, rows):
    """share of the lock-step steps of groups of `rows` rows in which at least one of the rows' macroblocks (x = t - g) is touched: what a
    kernel that keeps the lock step but leaves idle steps out ("fewer rows per wave so that whole steps vanish") still runs"""
    H, W = touched.shape
    live = tot = 0
    for y0 in range(0, H, rows):
        g = touched[y0:y0 + rows]
        R = g.shape[0]
        for t in range(W + R - 1):
            tot += 1
            live += any(0 <= t - r < W and g[r, t - r] for r in range(R))
    return live / tot


def main():
    orc = Oracle()
    for path in sys.argv[1:]:
        packets, _, _ = front.capture_packets(open(path, "rb").read())
        lo = ln = co = cn = 0
        tf, of, ls = [], [], []
        for pkt in packets[:12]:
            h = P.Packet(pkt).hdr
            W, H = int(h["width_mbs"]), int(h["height_mbs"])
            bs = orc.frame_bs(pkt, W * H).reshape(H, W, 2, 4, 4)
            own = (bs.reshape(H, W, -1) != 0).any(2)
            t = own.copy()
            t[:, :-1] |= (bs[:, 1:, 0, 0, :] != 0).any(2)
            t[:-1, :] |= (bs[1:, :, 1, 0, :] != 0).any(2)
            tf.append(t.mean()); of.append(own.mean())
            a, b = sim(t, 8); lo += a; ln += b
            a, b = sim(t, 15); co += a; cn += b
            ls.append([lockstep_live(t, r) for r in (8, 4, 2, 1)])
        print(f"{'':32s} lock step kept, idle steps left out: live steps with 8 / 4 / 2 / 1 rows per wave " + " / ".join(f"{v:.3f}" for v in np.mean(ls, 0))
              + "   (lanes busy per instruction: 64 / 32 / 16 / 8 of 64)")
        print(f"{os.path.basename(path):32s} own edges {np.mean(of):.3f}  touched {np.mean(tf):.3f}   luma steps {lo} -> {ln} ({ln / lo:.3f})   chroma steps {co} -> {cn} ({cn / co:.3f})")


if __name__ == "__main__":
    main()

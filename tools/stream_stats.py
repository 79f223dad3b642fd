#!/usr/bin/env python3
"""tools/stream_stats.py -- what a bitstream asks of the kernels: macroblock kinds, partition shapes, quarter-sample classes, coded residual,
boundary strengths and how many edge lines the deblocking filter really changes.  Host only: the stream goes through the reference's parser +
our emitters (capture sink), the numbers come from the command packets and from the CPU oracle's replay (checker-side code: this is a
measuring tool, not the product).

    python tools/stream_stats.py tests/golden/streams/nat1080_ipp30.264 tests/golden/streams/hd1080_ipp30.264
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edge264_amd import front, packet as P  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402


def stats_of(path, max_pictures=None):
    data = open(path, "rb").read()
    packets, _, _ = front.capture_packets(data)
    if max_pictures:
        packets = packets[:max_pictures]
    orc_a, orc_b = Oracle(), Oracle()
    tot = dict(pictures=0, mbs=0, inter=0, intra=0, inter_no_residual=0, uni16x16=0, uni16x16_no_residual=0, bs_segments=0, bs0=0, bs_strong=0,
               edge_lines=0, lines_changed=0, mb_all_bs0=0, packet_bytes=0)
    klass = np.zeros(6, np.int64)
    dpb_a = dpb_b = None
    for pkt in packets:
        pk = P.Packet(pkt)
        h = pk.hdr
        W, H = int(h["width_mbs"]), int(h["height_mbs"])
        n = W * H
        nb = int(h["plane_size_Y"]) + int(h["plane_size_C"])
        if dpb_a is None:
            dpb_a = [np.zeros(nb + 64, np.uint8) for _ in range(32)]
            dpb_b = [np.zeros(nb + 64, np.uint8) for _ in range(32)]
        mbs = pk.mbs
        kind = mbs["kind"]
        inter = kind == P.MB_INTER
        coded = mbs["coded"] != 0
        tot["pictures"] += 1
        tot["mbs"] += n
        tot["inter"] += int(inter.sum())
        tot["intra"] += int(((kind != P.MB_INTER) & (kind != P.MB_ABSENT)).sum())
        tot["inter_no_residual"] += int((inter & ~coded).sum())
        tot["packet_bytes"] += len(pkt)
        # partition shape and quarter-sample classes from the motion directory (edge264_cmd.h: mot_hdr bits 8-9 = one partition of that list)
        if inter.any():
            mot_hdr = np.ascontiguousarray(mbs["modes"]).view("<u4").reshape(n, 2)[:, 1]
            uni0, uni1 = (mot_hdr >> 8 & 1).astype(bool), (mot_hdr >> 9 & 1).astype(bool)
            one = inter & (uni0 ^ uni1)
            tot["uni16x16"] += int(one.sum())
            tot["uni16x16_no_residual"] += int((one & ~coded).sum())
        # boundary strengths (the oracle's restatement of deblock.c:958-1118) and what deblocking changes
        bs = orc_a.frame_bs(pkt, n)  # (n, 2, 4, 4): direction, edge, segment
        dbk = (mbs["flags"] & P.MBF_DEBLOCK) != 0
        b = bs[dbk]
        tot["bs_segments"] += int(b.size)
        tot["bs0"] += int((b == 0).sum())
        tot["bs_strong"] += int((b == 4).sum())
        tot["mb_all_bs0"] += int((b.reshape(len(b), -1) == 0).all(1).sum())
        d = int(h["dst_slot"])
        # the same picture before and after deblocking, each chain predicting from its own kind of references would drift: both from the deblocked ones
        for i in range(32):
            dpb_a[i][:] = dpb_b[i]
        orc_a.decode_frame(pkt, dpb_a, 1)
        orc_b.decode_frame(pkt, dpb_b, 3)
        sY = int(h["stride_Y"])
        A = dpb_a[d][:sY * H * 16].reshape(H * 16, sY)[:, :W * 16].astype(np.int16)
        B = dpb_b[d][:sY * H * 16].reshape(H * 16, sY)[:, :W * 16].astype(np.int16)
        ch = A != B
        # luma edge lines: vertical edges at x = 4k (q0 column), horizontal at y = 4k; a line counts as filtered when p0 or q0 changed
        v = ch[:, 4::4] | ch[:, 3:-1:4]
        hz = ch[4::4, :] | ch[3:-1:4, :]
        tot["edge_lines"] += int(v.size + hz.size)
        tot["lines_changed"] += int(v.sum() + hz.sum())
    t = tot
    out = {
        "file": os.path.basename(path), "pictures": t["pictures"], "packet_MB_per_picture": round(t["packet_bytes"] / t["pictures"] / 1e6, 3),
        "inter_share": round(t["inter"] / t["mbs"], 4), "intra_share": round(t["intra"] / t["mbs"], 4),
        "inter_without_residual": round(t["inter_no_residual"] / max(t["inter"], 1), 4),
        "one_partition_one_list": round(t["uni16x16"] / max(t["inter"], 1), 4),
        "one_partition_one_list_no_residual": round(t["uni16x16_no_residual"] / max(t["inter"], 1), 4),
        "bS_0_share_of_segments": round(t["bs0"] / max(t["bs_segments"], 1), 4), "bS_4_share": round(t["bs_strong"] / max(t["bs_segments"], 1), 4),
        "macroblocks_with_every_bS_0": round(t["mb_all_bs0"] / max(t["mbs"], 1), 4),
        "luma_edge_lines_changed_by_the_filter": round(t["lines_changed"] / max(t["edge_lines"], 1), 4),
    }
    return out


if __name__ == "__main__":
    for path in sys.argv[1:]:
        print(json.dumps(stats_of(path)))

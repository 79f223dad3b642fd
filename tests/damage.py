"""Damaged-stream scenarios for the concealment tests: a slice NAL cut short, followed by an intact copy of itself (what an
application does when it re-requests a lost packet, and the only way concealed samples become observable through the
reference's API: a picture whose slice failed never completes on its own, src/edge264_headers.c:539, 435-442)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = os.path.join(HERE, "golden", "streams")


def nal_units(data: bytes) -> list[bytes]:
    pos, i = [], 0
    while True:
        j = data.find(b"\0\0\1", i)
        if j < 0:
            break
        pos.append(j)
        i = j + 3
    return [data[a:b] for a, b in zip(pos, pos[1:] + [len(data)])]


def slice_indices(nals: list[bytes]) -> list[int]:
    return [i for i, n in enumerate(nals) if (n[3] & 31) in (1, 5, 20)]


def truncated_then_resent(name: str, which: int, keep: float) -> bytes:
    """The fixture with slice NAL number `which` (index into its slice NALs) preceded by a copy cut to `keep` of its length."""
    nals = nal_units(open(os.path.join(STREAMS, name + ".264"), "rb").read())
    k = slice_indices(nals)[which]
    bad = nals[k][:max(8, int(len(nals[k]) * keep))]
    return b"".join(nals[:k] + [bad] + nals[k:])


def truncated_only(name: str, which: int, keep: float) -> bytes:
    """The slice is cut short and never comes again."""
    nals = nal_units(open(os.path.join(STREAMS, name + ".264"), "rb").read())
    k = slice_indices(nals)[which]
    return b"".join(nals[:k] + [nals[k][:max(8, int(len(nals[k]) * keep))]] + nals[k + 1:])


def two_truncated_then_resent(name: str, which: int, keep_a: float, keep_b: float) -> bytes:
    """TWO failures inside one picture before anything is sent again: slice NALs `which` and `which + 1` (of the same picture) both
    cut short, then both sent again intact (/root/reference/src/edge264_headers.c:295-430, 486-529)."""
    nals = nal_units(open(os.path.join(STREAMS, name + ".264"), "rb").read())
    si = slice_indices(nals)
    a, b = si[which], si[which + 1]
    bad_a = nals[a][:max(8, int(len(nals[a]) * keep_a))]
    bad_b = nals[b][:max(8, int(len(nals[b]) * keep_b))]
    return b"".join(nals[:a] + [bad_a, bad_b, nals[a], nals[b]] + nals[b + 1:])


# (fixture, slice NAL, fraction kept): I / P / B slices, CAVLC and CABAC, one and several slices per picture, slice-boundary
# deblocking on and off, arbitrary slice order, 8x8 transform, I_PCM, MVC, weighted prediction
RESENT = [
    ("ipp_partitions", 0, 0.3), ("ipp_partitions", 1, 0.7), ("ipp_partitions", 3, 0.5),
    ("cabac_ipp", 0, 0.6), ("cabac_ipp", 2, 0.3),
    ("i_4x4_16x16_pcm", 0, 0.3), ("i_4x4_16x16_pcm", 1, 0.7),
    ("slices_deblock_idc", 1, 0.5), ("slices_deblock_idc", 4, 0.3), ("slices_deblock_idc", 7, 0.7),
    ("cabac_slices_deblock_idc", 3, 0.3), ("cabac_slices_deblock_idc", 8, 0.7),
    ("aso_slices", 1, 0.3), ("aso_slices", 6, 0.7), ("aso_slices", 9, 0.3), ("aso_slices", 14, 0.5),
    ("ipb_spatial", 2, 0.3), ("ipb_spatial", 3, 0.7), ("cabac_ipb_spatial", 2, 0.5),
    ("t8x8_plain", 1, 0.3), ("cabac_t8x8_slices", 5, 0.3), ("cabac_t8x8_slices", 8, 0.7),
    ("mvc_ipp", 3, 0.5), ("cabac_weighted_b", 4, 0.4), ("reorder_weighted", 5, 0.6),
    # round 5: encoder-shaped pictures (tests/golden/nat_encoder.py on 320 x 192): the concealment's P_Skip / B_Skip macroblocks take their vectors from
    # coherent neighbourhoods and skip runs, not from random syntax
    ("nat_small_ipp8", 0, 0.5), ("nat_small_ipp8", 2, 0.3), ("nat_small_ipp8", 5, 0.6),
    ("cabac_nat_small_ibbp10", 2, 0.4), ("cabac_nat_small_ibbp10", 4, 0.5), ("cabac_nat_small_ibbp10", 6, 0.7),
    # ... and with the High-profile tools (8x8 transform by choice, two references in P macroblocks, implicit weighted bi-prediction)
    ("cabac_nat_small_high_ibbp10", 1, 0.5), ("cabac_nat_small_high_ibbp10", 3, 0.3), ("cabac_nat_small_high_ibbp10", 7, 0.6),
    # ... and rate-control-shaped: adaptive QP and three slices per picture (the failed slice's neighbours above and below belong to other slices
    # of the same picture: concealment next to macroblocks that stay)
    ("nat_small_aq_slices_ipp8", 1, 0.5), ("nat_small_aq_slices_ipp8", 4, 0.4), ("nat_small_aq_slices_ipp8", 8, 0.6), ("nat_small_aq_slices_ipp8", 12, 0.3),
    ("cabac_nat_small_aq_slices_ibbp10", 1, 0.5), ("cabac_nat_small_aq_slices_ibbp10", 4, 0.3), ("cabac_nat_small_aq_slices_ibbp10", 7, 0.5),
    ("cabac_nat_small_aq_slices_ibbp10", 11, 0.6), ("cabac_nat_small_aq_slices_ibbp10", 14, 0.4),
    # ... an I picture in the middle of the stream cut short (the I-slice concealment blends with the neighbours' DC before the copy arrives), Intra8x8 inside
    ("nat_small_i8x8_iipp6", 1, 0.5), ("cabac_nat_small_i8x8_iipp6", 1, 0.4), ("cabac_nat_small_i8x8_iipp6", 3, 0.5),
    ("nat_small_sub_ipp8", 3, 0.5), ("cabac_nat_small_fade_wp_ipp8", 4, 0.6),
]
LOST = [("ipp_partitions", 3, 0.5), ("cabac_ipp", 3, 0.4), ("slices_deblock_idc", 11, 0.5), ("cabac_t8x8_slices", 11, 0.5),
        ("nat_small_aq_slices_ipp8", 22, 0.5), ("cabac_nat_small_aq_slices_ibbp10", 28, 0.4), ("nat_small_rect_ipp8", 7, 0.5)]
# two failed slices in ONE picture, then both sent again: (fixture, first of the two slice NALs -- both in the same picture --, fractions kept).
# I and P / B pictures, CAVLC and CABAC, three and four slices per picture, arbitrary slice order, 8x8 transform.  (Pairs that span two
# pictures are left out: the unmodified reference aborts on them -- assertion in its own worker_loop, src/edge264_headers.c:465.)
RESENT2 = [
    ("slices_deblock_idc", 0, 0.3, 0.7), ("slices_deblock_idc", 1, 0.6, 0.4), ("slices_deblock_idc", 3, 0.3, 0.7), ("slices_deblock_idc", 7, 0.6, 0.4),
    ("cabac_slices_deblock_idc", 0, 0.6, 0.4), ("cabac_slices_deblock_idc", 4, 0.3, 0.7), ("cabac_slices_deblock_idc", 9, 0.6, 0.4),
    ("aso_slices", 1, 0.3, 0.7), ("aso_slices", 5, 0.6, 0.4), ("aso_slices", 10, 0.3, 0.7), ("aso_slices", 17, 0.6, 0.4),
    ("cabac_t8x8_slices", 0, 0.3, 0.7), ("cabac_t8x8_slices", 4, 0.6, 0.4), ("cabac_t8x8_slices", 10, 0.3, 0.7),
    # round 5: encoder-shaped pictures of three slices
    ("nat_small_aq_slices_ipp8", 0, 0.3, 0.7), ("nat_small_aq_slices_ipp8", 4, 0.6, 0.4), ("nat_small_aq_slices_ipp8", 10, 0.3, 0.7),
    ("cabac_nat_small_aq_slices_ibbp10", 3, 0.6, 0.4), ("cabac_nat_small_aq_slices_ibbp10", 7, 0.3, 0.7), ("cabac_nat_small_aq_slices_ibbp10", 12, 0.6, 0.4),
]


# Damaged streams kept as files: cases of tools/damage_sweep.py (round 5) in which the front end and the unmodified reference disagreed.  The first five:
# a slice NAL cut BEHIND its last macroblock (it still decodes completely and completes the picture), followed by its intact copy, which then fails
# before its first macroblock -- a stray slice for a picture that has already gone out.
DAMAGED_FILES = ["sweep_1998", "sweep_6872", "sweep_11707", "sweep_11790", "sweep_16906",
                 # CAVLC + 8x8 transform, several slices, a B picture: a macroblock of the failed slice is decoded again by the copy and NOT deblocked again (the
                 # reference had deblocked its first version): its per-4x4 coefficient flags stay as parsed, the bS of its neighbours follows them
                 "sweep_134724", "sweep_147736", "sweep_151776",
                 # an I_PCM macroblock that a failed slice ran over is decoded again by its own slice, lifted into a packet, and deblocked by the reference only
                 # at the end of the picture: the earlier deblocking of its old version must not count (pictures of 9 - 15 slices, several cuts per stream)
                 "sweep_207375", "sweep_6677", "sweep_6298",
                 # the failed slice ends in I_PCM macroblocks, which the reference deblocks after the unref callback and before it conceals them
                 "sweep_205453"]
DAMAGED_DIR = os.path.join(HERE, "golden", "damaged")

"""A committed sample of the randomised differential corpus on the DEVICE.

tests/golden/corpus/gpu_corpus_sample.zip = 500 streams drawn by tools/make_gpu_corpus.py --zip (seed ranges shifted by 50 000 against every sweep
and corpus of round 5): 200 + 150 random-syntax streams (tools/stream_sweep.py, narrow and wide option ranges: CAVLC / CABAC, slices, ASO, MVC,
8x8 transform, scaling lists, weighted prediction, PCM, long-term references, cropping ...), 100 damaged ones (a slice NAL cut and sent again,
tools/damage_sweep.py: the concealment paths) and 50 encoder-shaped clips (tools/nat_sweep.py).  index.json holds what the UNMODIFIED reference decoder
answered in the build container for each: the return code of every edge264_decode_NAL call and the md5 of every frame (the convention of
tests/golden/streams/reference_md5.json).  Here every stream goes through the edge264.h API of the product with the HIP sink."""
import hashlib
import json
import os
import zipfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CORPUS = os.path.join(HERE, "golden", "corpus", "gpu_corpus_sample.zip")


def load():
    z = zipfile.ZipFile(CORPUS)
    return z, json.loads(z.read("index.json"))


def test_corpus_sample_is_well_formed():
    """(CPU) the archive holds what its index says: four kinds, every stream with its reference answers."""
    z, index = load()
    kinds = {}
    for c in index:
        kinds[c["kind"]] = kinds.get(c["kind"], 0) + 1
        assert len(z.read(c["file"])) > 20 and c["codes"] and all(len(m) == 32 for m in c["md5"])
    assert set(kinds) == {"narrow", "wide", "damaged", "nat"} and sum(kinds.values()) >= 450, kinds
    assert sum(len(c["md5"]) for c in index) > 2000  # pictures


@pytest.mark.gpu
@pytest.mark.parametrize("wire", [0, 1], ids=["version4", "wire"])
def test_corpus_sample_on_the_hip_sink(wire):
    """wire = 1 (round 6): the front end folds its packets (include/edge264_compact.h) and the device unfolds them -- the same answers"""
    import ctypes as C
    from oracle.pyoracle import HipFront  # (test infrastructure: the ctypes binding of the edge264.h API; no oracle code runs here)
    h = HipFront()
    h.lib.e264front_set_sink(0)
    h.lib.e264front_set_compact.argtypes = [C.c_int]
    h.lib.e264front_set_compact(wire)
    try:
        _run_corpus(h)
    finally:
        h.lib.e264front_set_compact(0)


def _run_corpus(h):
    z, index = load()
    bad, pics = [], 0
    for c in index:
        frames, codes = h.decode(z.read(c["file"]))
        got = [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]
        pics += len(got)
        if codes != c["codes"] or got != c["md5"]:
            bad.append((c["kind"], c["seed"]))
    print(f"corpus sample on the device: {len(index)} streams, {pics} pictures, {len(bad)} mismatching")
    assert not bad, bad[:10]

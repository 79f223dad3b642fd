"""Prototype check (next round): the two-lines-per-lane packed 16-bit edge filter of
tools/proto/packed_edge_filter.h equals the scalar filter the shipped deblocking kernel uses, on seeded random edges
(all bS, luma and chroma, thresholds that pass and fail).  Host arithmetic only; nothing here is product code."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PROTO = os.path.join(os.path.dirname(HERE), "tools", "proto")


def test_packed_equals_scalar(tmp_path):
    exe = tmp_path / "pef_test"
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-o", str(exe), os.path.join(PROTO, "packed_edge_filter_test.c")], check=True)
    out = subprocess.run([str(exe), "1500000"], capture_output=True, text=True, timeout=120)
    n, bad, changed = (int(x) for x in out.stdout.split())
    assert out.returncode == 0 and bad == 0
    assert changed > n // 2   # the cases really filter something

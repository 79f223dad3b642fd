"""The emitters and wrappers under AddressSanitizer + UndefinedBehaviorSanitizer (tools/sanitize): the front end is built with both, capture sink,
and driven over every fixture, every damaged-stream scenario and seeded corruptions.  Nothing the sanitizers report may lie in
edge264_amd/frontend; what they report inside the reference's own parser (compiled where it lies, unchanged) is listed by the tool and is not
this repository's to change (profiles/r05_sanitizers.txt).  CPU only; needs /root/reference to build."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/root/reference/src/edge264.c") or shutil.which("gcc") is None, reason="builds the front end from /root/reference")
def test_no_sanitizer_report_in_the_emitters(tmp_path):
    out = tmp_path / "report.txt"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize", "run.py"), "--flips", "2", "--out", str(out)],
                       capture_output=True, text=True, timeout=900)
    text = out.read_text() if out.exists() else p.stdout + p.stderr
    assert p.returncode == 0, text[-3000:]
    assert "distinct reports located in the emitters / wrappers (edge264_amd/frontend): 0" in text
    first = {ln.split()[0]: ln for ln in text.splitlines() if ln.startswith(("fixture", "resent", "lost", "truncated", "flips_last"))}
    assert " clean 55 " in first["fixture"] or "reports_ours 0" in first["fixture"]
    assert all("reports_ours 0" in ln and "other_exit 0" in ln for ln in first.values()), first


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or not os.path.exists(os.path.join(ROOT, "edge264_amd", "libedge264_hip.so")),
                    reason="needs the ROCm clang (ASan runtime) and the built back end (its host-side packet validation)")
def test_validated_packets_keep_the_kernels_in_bounds():
    """tools/sanitize/kernel_fuzz.py, quick form: damaged command packets that the product's validation still accepts run through the kernels' source
    (host build, AddressSanitizer) on buffers of exactly the back end's sizes; a report is an access the device would make outside its allocations."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize", "kernel_fuzz.py"), "--per-packet", "2", "--stride", "9", "--seed", "7"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "no sanitizer report" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    assert "AddressSanitizer" not in p.stderr.replace("ASan doesn't fully support", "")

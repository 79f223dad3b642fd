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

"""The resource budgets the kernels' occupancy rests on, checked at build time: hipcc cross-compiles the kernel file
for gfx950 without a GPU and reports registers, spills, scratch and LDS per kernel (-Rpass-analysis=kernel-resource-usage).  A change that
spills vector registers, or pushes a kernel over the register / LDS budget of the number of waves it is designed to keep on a CU, shows here
and not as an unexplained slowdown on the next GPU visit:

  e264_pred_kernel          4 workgroups of 256 threads per CU: <= 128 VGPRs, <= 40 KB of LDS (160 KB / 4)
  e264_intra_kernel<16>     16 waves of one workgroup per CU:   <= 128 VGPRs (4 waves per SIMD), one workgroup's LDS <= 160 KB
  e264_deblock2_kernel<8>   8 waves per CU:                     <= 256 VGPRs (2 waves per SIMD), LDS <= 160 KB
  e264_dbkparam2_kernel     8 workgroups per CU (both forms):   <= 64 VGPRs, <= 20 KB of LDS
"""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

BUDGET = {  # kernel name fragment -> (max VGPRs, max LDS bytes per workgroup)
    "e264_pred_kernel": (128, 160 * 1024 // 4),
    "e264_intra_kernelILi16E": (128, 160 * 1024),
    "e264_deblock2_kernelILi8E": (256, 160 * 1024),
    "e264_dbkparam2_kernel": (64, 160 * 1024 // 8),
}


@pytest.mark.skipif(not shutil.which(HIPCC), reason="no hipcc")
def test_kernels_fit_their_occupancy_budgets(tmp_path):
    src = os.path.join(ROOT, "edge264_amd", "csrc", "e264_kernels.hip")
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", src, "-o", str(tmp_path / "k.o"),
                          "-Rpass-analysis=kernel-resource-usage", "-w"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for frag, (max_vgpr, max_lds) in BUDGET.items():
        hits = {k: v for k, v in kernels.items() if frag in k}
        assert hits, f"{frag}: not in the compiler's report ({sorted(kernels)})"
        for name, r in hits.items():
            if "dbkparam2_kernelILb1E" in name:  # the general form of the parameter kernel is HELD to 64 VGPRs for eight workgroups per CU: two registers spill, by choice
                assert r["VGPRs Spill"] <= 2 and r["ScratchSize"] <= 16, f"{name}: spills more than the two registers it was measured with ({r})"
            else:
                assert r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, f"{name}: spills ({r})"
            assert r["VGPRs"] + r.get("AGPRs", 0) <= max_vgpr, f"{name}: {r['VGPRs']} VGPRs > {max_vgpr}"
            assert r["LDS Size"] <= max_lds, f"{name}: {r['LDS Size']} bytes of LDS > {max_lds}"

"""CPU tests of the host side: C-ABI surface, packet format, stream sharding (gloo, world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "edge264_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(e264hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    """libedge264_hip.so loads without a GPU and exports every entry point include/edge264_hip.h declares
    (no compute call is made here)."""
    from edge264_amd import backend
    path = backend.LIB_PATH
    assert os.path.exists(path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(path)
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/edge264_hip.h but not exported"
    assert set(names) == set(backend.EXPORTED_SYMBOLS)


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly, not fall back to the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from edge264_amd import backend
    with pytest.raises(backend.BackendError):
        backend.Device(0)


def test_product_does_not_import_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "edge264_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "pyoracle" not in txt and "e264_oracle" not in txt, f"{f} references the oracle"


def test_packet_struct_sizes_match_header():
    from edge264_amd import packet as P
    hdr = open(os.path.join(ROOT, "include", "edge264_cmd.h")).read()
    assert P.FRAME_HDR.itemsize == 80 and P.MB.itemsize == 32 and P.MOTION.itemsize == 144 and P.SLICE_PARAMS.itemsize == 2112
    for name, size in (("E264FrameHdr", 80), ("E264Mb", 32), ("E264Motion", 144), ("E264SliceParams", 2112)):
        assert re.search(rf"sizeof\({name}\)\s*==\s*{size}", hdr), f"{name}: static assert on {size} bytes missing from edge264_cmd.h"


def test_packet_roundtrip_and_algorithmic_bytes():
    from edge264_amd import packet as P, synth
    g = synth.StreamSynth(6, 4, seed=5, num_refs=2)
    pkts = g.gop("IPB")
    for raw in pkts:
        pk = P.Packet(raw)
        assert int(pk.hdr["magic"]) == P.E264_MAGIC and int(pk.hdr["total_bytes"]) == len(raw)
        assert pk.mbs.shape == (24,)
        b = pk.algorithmic_bytes()
        # at least the frame written once + the command bytes, at most that + 2 full reference reads
        fb = 24 * 384
        assert fb + len(raw) <= b <= fb * 4 + len(raw)
    assert P.Packet(pkts[0]).motion is None and P.Packet(pkts[1]).motion is not None


def test_shard_streams_partition():
    from edge264_amd.sharding import shard_streams
    for n in (0, 1, 7, 256, 1000):
        for w in (1, 2, 3, 8):
            parts = [shard_streams(n, r, w) for r in range(w)]
            assert sum(len(p) for p in parts) == n
            assert [i for p in parts for i in p] == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_streams(4, 2, 2)


WORKER = r"""
import os, sys, json, hashlib
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from edge264_amd import packet as P, synth
from edge264_amd.sharding import rank_info, shard_streams, reduce_elapsed
from oracle.pyoracle import Oracle   # test infrastructure: stands in for the GPU in this CPU test
rank, local, world = rank_info()
dist.init_process_group("gloo", rank=rank, world_size=world)
orc = Oracle()
n_streams, W, H = 5, 4, 3
mine = shard_streams(n_streams, rank, world)
digests = {{}}
frames = 0
for sid in mine:
    pk = synth.StreamSynth(W, H, seed=100 + sid, num_refs=2).gop("IPP")
    nb = P.frame_bytes(W, H)
    dpb = [np.full(nb + 16, 128, np.uint8) for _ in range(4)] + [None] * 28
    for p in pk:
        orc.decode_frame(p, dpb, 3)
        frames += 1
    last = int(P.Packet(pk[-1]).hdr["dst_slot"])
    digests[sid] = hashlib.md5(dpb[last][:nb].tobytes()).hexdigest()
dist.barrier()
elapsed, total = reduce_elapsed(0.5 + rank, frames, dist)
gathered = [None] * world
dist.all_gather_object(gathered, digests)
if rank == 0:
    merged = {{}}
    for g in gathered: merged.update(g)
    print(json.dumps({{"elapsed": elapsed, "frames": total, "digests": merged}}))
dist.destroy_process_group()
"""


def test_two_rank_sharded_run_gloo(tmp_path):
    """world_size 2 over gloo: every stream is decoded by exactly one rank, results equal the
    single-process run, the reduction reports max time and summed frames, and no data-path collective exists."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    import socket
    with socket.socket() as sk:  # a port that is free NOW (a fixed one collides with whatever else rendezvouses on this host)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["frames"] == 15 and abs(res["elapsed"] - 1.5) < 1e-9
    # single-process reference of the same streams
    import hashlib
    from edge264_amd import packet as P, synth
    from oracle.pyoracle import Oracle
    orc = Oracle()
    for sid in range(5):
        pk = synth.StreamSynth(4, 3, seed=100 + sid, num_refs=2).gop("IPP")
        nb = P.frame_bytes(4, 3)
        dpb = [np.full(nb + 16, 128, np.uint8) for _ in range(4)] + [None] * 28
        for p in pk:
            orc.decode_frame(p, dpb, 3)
        last = int(P.Packet(pk[-1]).hdr["dst_slot"])
        assert res["digests"][str(sid)] == hashlib.md5(dpb[last][:nb].tobytes()).hexdigest()


def test_committed_bench_line_obeys_the_contract():
    """profiles/r01k_bench_default.json is the stdout of `python bench.py` on the MI355X: the one JSON line the driver parses."""
    import json
    path = os.path.join(ROOT, "profiles", "r01k_bench_default.json")
    lines = [l for l in open(path).read().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints exactly ONE JSON line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    # achieved = algorithmic bytes per launch / average launch duration of the dominant kernel
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["bit_exact"] is True
    # whole-job throughput: frames of all streams / wall time
    assert abs(d["value"] - d["config"]["frames_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_round4_bench_line_says_what_it_measures():
    """profiles/r04_bench_default.json = stdout of `python bench.py` on the MI355X, round 4: the line says in
    `config.workload` that the packets are resident and the H2D copy excluded, carries the PCIe-inclusive rates, a SAME-INPUT leg
    (the two 1080p bitstream fixtures on the GPU, resident and with the copy inside, next to the CPU reference on the same files,
    bit-exact) and names the file its PMC traffic figure comes from."""
    import json
    lines = [l for l in open(os.path.join(ROOT, "profiles", "r04_bench_default.json")).read().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "pcie_inclusive", "same_input", "build_flags"):
        assert k in d, k
    w = d["config"]["workload"]
    assert "RESIDENT IN HBM" in w and "H2D" in w and "excluded" in w and "pcie_inclusive" in w
    assert d["bit_exact"] is True and d["build_flags"] == ""
    v = d["verify"]   # breadth of the check: 4 distinct GOPs dealt to the streams, not 256 copies of one
    assert v["mismatching_frames"] == 0 and v["frames_compared"] == 2048 and v["distinct_pictures"] == 32 and "4 distinct GOPs" in w
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    k = r["kernels"][r["kernel"]]
    assert abs(r["achieved"] - (k["sample_bytes"] + k["command_bytes"]) / (k["ms_per_launch"] * 1e-3) / 1e9) < 1.0
    assert r["traffic"] > 0 and r["traffic_source"]["file"].startswith("profiles/r")
    p = d["pcie_inclusive"]
    assert p["value"] > 30000 and p["pinned_in_place"]["value"] > 30000   # 1000 x 1080p30 with the copy inside
    si = d["same_input"]
    assert si["bit_exact"] is True and si["files"] == ["hd1080_ipp30.264", "cabac_hd1080_ibbp30.264"]
    for key in ("gpu_resident_frames_per_s", "gpu_pcie_inclusive_frames_per_s", "host_parse_emit_frames_per_s_one_core", "cpu_reference_frames_per_s", "cpu_cores"):
        assert si[key] > 0, key
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["same_input"] is True and c["cores"] >= 1 and c["value"] == si["cpu_reference_frames_per_s"]
    assert d["per_rank"]["numa"]["numa_node"] >= 0   # the rank found its GPU's NUMA node (also at N = 1)
    assert abs(d["value"] - d["config"]["frames_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_round5_bench_line_carries_the_system_figure():
    """profiles/r05_bench_default.json = stdout of `python bench.py` on the MI355X, round 5: a `system` leg (the whole
    decoder with the host inside the clock: e264_multi, parser + emitters alone and end to end, next to cpu_baseline on the same files and cores),
    the encoder-shaped fixtures in `same_input` with per-kernel times, the CPU legs taken under the launcher's affinity, one page-locked buffer
    per stream and frame in the pinned leg, configs[3] on the same number of distinct GOPs as the headline."""
    import json
    lines = [l for l in open(os.path.join(ROOT, "profiles", "r05_bench_default.json")).read().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "frames/s" and d["dtype"] == "u8" and d["vs_baseline"] is None
    assert d["bit_exact"] is True and d["build_flags"] == ""
    assert abs(d["value"] - d["config"]["frames_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    r = d["roofline"]
    k = r["kernels"][r["kernel"]]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - (k["sample_bytes"] + k["command_bytes"]) / (k["ms_per_launch"] * 1e-3) / 1e9) < 1.0
    assert r["traffic"] > 0 and r["traffic_source"]["file"].startswith("profiles/r05")
    sy = d["system"]
    assert sy["threads"] >= 1 and sy["cores"] >= sy["threads"] and len(sy["files"]) == 4
    for leg in ("parse_only", "end_to_end"):
        assert sy[leg]["frames_per_s"] > 0 and sy[leg]["decode_ms_per_picture"] > 0 and sy[leg]["frames"] > 1000
        assert sy[leg]["steady_state"] is True and sy[leg]["whole_run"]["frames_per_s"] <= sy[leg]["frames_per_s"] * 1.02  # the start-up is inside `whole_run` only
    assert abs(sy["end_to_end_vs_parse_only"] - sy["end_to_end"]["frames_per_s"] / sy["parse_only"]["frames_per_s"]) < 0.01
    assert sy["host_cores_for_1000_streams_1080p30"] > 0 and sy["cpu_reference_frames_per_s"] == d["cpu_baseline"]["value"]
    si = d["same_input"]
    assert si["bit_exact"] is True and si["files"] == ["hd1080_ipp30.264", "cabac_hd1080_ibbp30.264", "nat1080_ipp30.264", "cabac_nat1080_ibbp30.264"]
    for f in si["files"]:
        pf = si["per_file"][f]
        assert pf["mismatching"] == 0 and pf["pictures"] == 30 and all(v >= 0 for v in pf["kernel_ms_per_launch"].values())
    assert "launcher" in d["cpu_baseline"]["affinity"] and d["cpu_baseline"]["same_input"] is True
    p = d["pcie_inclusive"]
    assert p["value"] > 30000 and p["pinned_in_place"]["value"] > 30000
    assert p["pinned_in_place"]["pinned_buffers"] == d["config"]["streams_per_gpu"] * 8   # one per stream and frame of the GOP
    for v in d["other_configs"].values():
        assert v["bit_exact"] is True and v["distinct_pictures"] == 4 * len(v["gop"])


def test_bench_gpus_flag_spawns_ranks(tmp_path):
    """`python bench.py --gpus 2` outside torchrun launches 2 ranks itself (torch.distributed.run, here gloo + a stub device):
    rank 0 prints ONE line with n_gpus 2 and the frames of BOTH ranks' stream shards; a mismatch between --gpus and an
    inherited WORLD_SIZE is refused instead of silently benchmarking the wrong GPU count."""
    import json
    env = dict(os.environ, E264_BENCH_BACKEND="tests.stub_backend", OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "3", "--gop", "IP",
           "--width-mbs", "4", "--height-mbs", "3"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["frames_per_step"] == 2 * 3 * 2  # ranks x streams per rank x frames of the GOP
    assert d["roofline"]["kernel"] == "e264_pred_kernel" and d["cpu_baseline"] is None
    # WORLD_SIZE inherited from a launcher that disagrees with --gpus: refuse
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run(cmd, env=env1, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert bad.returncode == 2 and "refusing" in bad.stderr


def test_bench_config5_shards_a_fixed_job(tmp_path):
    """BASELINE configs[4] (`--config5` = `--total-streams 256`, here 5 streams on 2 stub ranks): the job's streams are
    FIXED and sharded over the ranks (3 + 2), the line says strong scaling, reports every rank's own rate with min / max,
    and a rank without a stream is refused."""
    import json
    env = dict(os.environ, E264_BENCH_BACKEND="tests.stub_backend", OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--total-streams", "5", "--gop", "IP",
           "--width-mbs", "4", "--height-mbs", "3"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["total_streams"] == 5 and d["config"]["frames_per_step"] == 5 * 2
    assert "configs[4]" in d["config"]["workload"]
    pr = d["per_rank"]
    assert len(pr["frames_per_s"]) == 2 and pr["min"] == min(pr["frames_per_s"]) and pr["max"] == max(pr["frames_per_s"])
    assert abs(sum(pr["frames_per_s"]) - d["value"]) / d["value"] < 0.5   # same order: value uses the slowest rank's time
    few = subprocess.run(cmd[:-8] + ["--total-streams", "1", "--gop", "IP", "--width-mbs", "4", "--height-mbs", "3"], env=env, capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert few.returncode != 0


def test_bench_says_when_the_synthetic_content_is_overridden():
    """bench.py's measuring aids (E264_I_KINDS, E264_RESIDUAL_PROB, E264_SYNTH_KW: tools/visits/gpu_ikinds.sh, gpu_sweep.sh) change the synthetic
    content: a line produced with one of them set says that it is NOT the BASELINE workload and which override was in force; without
    them `config.synth_overrides` is null."""
    import json
    env = dict(os.environ, E264_BENCH_BACKEND="tests.stub_backend", OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "E264_I_KINDS", "E264_RESIDUAL_PROB", "E264_SYNTH_KW"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--streams", "2", "--gop", "IP", "--width-mbs", "4", "--height-mbs", "3",
           "--variants", "2"]
    for over in ({}, {"E264_I_KINDS": "8", "E264_SYNTH_KW": '{"mv_range": 0}'}):
        out = subprocess.run(cmd, env=dict(env, **over), capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        if over:
            assert d["config"]["synth_overrides"] == over and "NOT the BASELINE workload" in d["config"]["workload"]
        else:
            assert d["config"]["synth_overrides"] is None and "NOT the BASELINE" not in d["config"]["workload"]
            assert "2 distinct GOPs" in d["config"]["workload"]


def test_bench_line_prices_the_kernels_against_valu_issue():
    """at the configuration the canned SQ counters were taken on (256 streams, IPPPPPPP, 120 x 68) the line carries, per kernel,
    the VALU wave-instructions per launch, their share of the chip's issue slots at the live kernel time and lane instructions per sample; `bound`
    says what the numbers say; the HBM figures of the contract stay.  The legs that need a device (`single_stream`, `pcie_inclusive.link_frac`) are
    keys of the line and null here (stub device)."""
    import json
    env = dict(os.environ, E264_BENCH_BACKEND="tests.stub_backend", OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--variants", "1", "--no-verify", "--no-other-configs",
           "--no-cpu-baseline", "--no-same-input", "--no-system", "--no-host-packets"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    r = d["roofline"]
    assert r["bound"] in ("valu-issue", "hbm") and r["peak"] == 8000.0 and 0 <= r["frac"]
    vi = r["valu_issue"]
    assert vi["source"].startswith("profiles/r") and vi["rate_G_wave_instr_per_s_per_simd"] == 0.54 and vi["simds"] == 1024
    assert set(vi["kernels"]) >= {"e264_pred_kernel", "e264_dbkparam2_kernel", "e264_intra_kernel"}
    for k, e in vi["kernels"].items():
        assert e["valu_wave_instr_per_launch"] > 1e6 and e["valu_issue_frac"] > 0 and e["lane_instr_per_sample"] > 0, k
    assert r["valu_issue_frac"] == vi["kernels"][r["kernel"]]["valu_issue_frac"]
    assert "single_stream" in d and d["single_stream"] is None


def test_numa_helpers():
    from edge264_amd.sharding import bind_rank_to_gpu_socket, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == []
    before = os.sched_getaffinity(0)
    info = bind_rank_to_gpu_socket(0)   # no GPU here: node unknown, affinity untouched
    assert info["bound"] is False and os.sched_getaffinity(0) == before


def test_product_library_reports_no_build_switches_and_ablation_builds_are_refused(monkeypatch):
    """e264hip_build_flags() is "" for the product build; a library that reports a timing-ablation switch (wrong
    samples by design) is refused by the loader unless E264_ALLOW_ABLATION=1."""
    from edge264_amd import backend
    assert backend.build_flags() == ""
    lib = backend.load_library()

    class Fake:
        def __init__(self, real, flags):
            self._real, self._flags = real, flags

        def __getattr__(self, name):
            if name == "e264hip_build_flags":
                f = lambda: self._flags  # noqa: E731
                return f
            return getattr(self._real, name)

    monkeypatch.setattr(backend, "_lib", None)
    monkeypatch.setattr(backend.C, "CDLL", lambda path: Fake(lib, b" E264_ABL_NOLOAD"))
    monkeypatch.delenv("E264_ALLOW_ABLATION", raising=False)
    with pytest.raises(backend.BackendError, match="ablation"):
        backend.load_library()
    monkeypatch.setenv("E264_ALLOW_ABLATION", "1")
    assert backend.load_library() is not None
    monkeypatch.setattr(backend, "_lib", lib)

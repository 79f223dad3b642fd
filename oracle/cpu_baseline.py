"""CPU baseline of SURVEY.md 8(d) -- CHECKER-SIDE code (bench.py's cpu_baseline leg only).

The UNMODIFIED reference decoder (oracle/_ref/libedge264_ref.so, compiled from /root/reference by oracle/Makefile) decodes
the committed 1080p Annex-B fixtures, P independent processes each with its own `edge264_alloc(0, ...)` instance (one
stream each, pinned to distinct cores), P in {1, all cores}: frames/s aggregate and per core, wall clock, like
src/edge264_test.c:482-483, 522-542 (the clock includes allocation, as the reference's own benchmark does).
`-m` style multi-threading inside one decoder is not used (hangs)."""
from __future__ import annotations

import ctypes as C
import errno
import multiprocessing as mp
import os
import time

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = os.path.join(os.path.dirname(HERE), "tests", "golden", "streams")
# 30 pictures each.  Random syntax (tests/golden/make_streams.py): CAVLC IPPP; CABAC IBBP with 8x8 transform + scaling lists.  Encoder-shaped
# (tests/golden/nat_encoder.py, round 5: procedural video through a real motion search / mode decision / quantiser): CAVLC IPPP, CABAC IBBP.
FIXTURES_1080P = ("hd1080_ipp30.264", "cabac_hd1080_ibbp30.264", "nat1080_ipp30.264", "cabac_nat1080_ibbp30.264")


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores() -> list[int]:
    """One logical CPU per physical core among those this process may run on."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out


def cpu_quota() -> float | None:
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), None when unlimited: with a quota below
    the core count, "one process per core" only measures the quota."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else int(q) / int(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _decode_loop(args):
    """Worker: decode the fixtures round-robin for `seconds`; returns (frames, wall, cpu_user+sys)."""
    core, seconds, lib_path, blobs = args
    try:
        os.sched_setaffinity(0, {core})
    except OSError:
        pass
    L = C.CDLL(lib_path)
    L.edge264_alloc.restype = C.c_void_p
    L.edge264_alloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.edge264_free.argtypes = [C.POINTER(C.c_void_p)]
    L.edge264_find_start_code.restype = C.c_void_p
    L.edge264_find_start_code.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.edge264_decode_NAL.restype = C.c_int
    L.edge264_decode_NAL.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.edge264_get_frame.restype = C.c_int
    L.edge264_get_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    out = (C.c_uint8 * 256)()  # Edge264Frame (96 bytes)
    bufs = [C.create_string_buffer(b + b"\0" * 64, len(b) + 64) for b in blobs]
    frames, k = 0, 0
    t0, c0 = time.perf_counter(), time.process_time()
    while time.perf_counter() - t0 < seconds:
        buf, n = bufs[k % len(bufs)], len(blobs[k % len(bufs)])
        k += 1
        base = C.addressof(buf)
        end = base + n
        dec = C.c_void_p(L.edge264_alloc(0, None, None, 0, None, None, None))
        nal = L.edge264_find_start_code(base, end, 0)
        nal = (nal or end) + 3 if (nal or end) < end else end
        while True:
            nxt = L.edge264_find_start_code(nal, end, 0) if nal < end else end
            res = L.edge264_decode_NAL(dec, nal, nxt, None, None)
            got = 0
            while L.edge264_get_frame(dec, out, 0) == 0:
                got += 1
            frames += got
            if res == errno.ENOBUFS:
                if got == 0:
                    break
                continue
            if res == errno.ENODATA or nal >= end:
                break
            nal = min(nxt + 3, end)
        while L.edge264_get_frame(dec, out, 0) == 0:
            frames += 1
        L.edge264_free(C.byref(dec))
    return frames, time.perf_counter() - t0, time.process_time() - c0


def reference_decoder_baseline(seconds_single: float = 6.0, seconds_all: float = 8.0) -> dict:
    lib = os.path.join(HERE, "_ref", "libedge264_ref.so")
    if not os.path.exists(lib):
        raise FileNotFoundError(lib)
    names = [n for n in FIXTURES_1080P if os.path.exists(os.path.join(STREAMS, n))]
    blobs = [open(os.path.join(STREAMS, n), "rb").read() for n in names]
    host_cores = physical_cores()
    quota = cpu_quota()
    # one process per core the container can actually run at once: more processes than that only share the quota
    n = len(host_cores) if quota is None else max(1, min(len(host_cores), int(quota)))
    cores = host_cores[:: max(1, len(host_cores) // n)][:n]  # spread over the sockets
    saved = os.sched_getaffinity(0)
    try:
        f1, w1, c1 = _decode_loop((cores[0], seconds_single, lib, blobs))
    finally:
        os.sched_setaffinity(0, saved)  # _decode_loop pins its process: the caller (bench.py) must not stay on one core
    ctx = mp.get_context("fork")
    with ctx.Pool(len(cores)) as pool:
        res = pool.map(_decode_loop, [(c, seconds_all, lib, blobs) for c in cores])
    fa = sum(r[0] for r in res)
    wa = max(r[1] for r in res)
    return {"value": round(fa / wa, 1), "unit": "frames/s", "cores": len(cores), "kind": "reference",
            "per_core": round(fa / wa / len(cores), 2), "single_process": round(f1 / w1, 2),
            "cpu_model": cpu_model(), "host_physical_cores": len(host_cores), "cpu_quota_cores": quota,
            "cpu_seconds_total": round(sum(r[2] for r in res) + c1, 1),
            "sample": f"the unmodified reference decoder (edge264_alloc(0, ...), CAVLC/CABAC parsing + reconstruction + deblocking) on the "
                      f"1080p fixtures {', '.join(names)} decoded in a loop: 1 process for {w1:.1f} s ({f1} frames), then "
                      f"{len(cores)} processes pinned to distinct physical cores for {wa:.1f} s ({fa} frames)"
                      + (f"; the container's CPU quota is {quota:g} cores of the host's {len(host_cores)}" if quota is not None else "")}

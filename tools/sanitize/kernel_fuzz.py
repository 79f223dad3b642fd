#!/usr/bin/env python3
"""tools/sanitize/kernel_fuzz.py -- CHECKING TOOL: is the packet validation (e264hip_packet_check, the product's host-side gate in front of every
kernel launch) tight enough for the kernels?  Command packets of real streams and of the synthetic generator are damaged at random (bytes of the
header, the slice tables, the macroblock records, the motion records, the coefficient payload); whatever the validation still ACCEPTS is run
through the kernels' own source compiled for the host (tests/emu, here built with AddressSanitizer) -- a packet in its wire form (version 5,
include/edge264_compact.h) through e264_expand_kernel's source first -- on buffers of exactly the sizes the back end
allocates: the packet in a heap block of its own length, every DPB slot frame_bytes + 64 (e264hip_frame_alloc), the parameter scratch 64 bytes per
macroblock.  Any access the sanitizer reports is one the device would make outside its allocations.

Runs itself again under the ROCm clang's ASan runtime:   python tools/sanitize/kernel_fuzz.py [--per-packet N] [--seed S]
"""
import argparse
import ctypes as C
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
EMU = os.path.join(ROOT, "tests", "emu")


def asan_runtime():
    c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return c[-1] if c else None


def build():
    for src, out in (("pred_emu.cpp", "libe264_pred_emu_asan.so"), ("intra_emu.cpp", "libe264_intra_emu_asan.so")):
        o = os.path.join(HERE, out)
        deps = [os.path.join(EMU, src)] + glob.glob(os.path.join(ROOT, "edge264_amd", "csrc", "*.h")) + [os.path.join(ROOT, "include", "edge264_cmd.h"), os.path.join(ROOT, "include", "edge264_compact.h")]
        if os.path.exists(o) and all(os.path.getmtime(o) >= os.path.getmtime(d) for d in deps):
            continue
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-O1", "-g", "-std=c++17", "-fPIC", "-shared",
                        "-fvisibility=hidden", "-I" + EMU, "-Wno-unused-function", os.path.join(EMU, src), "-o", o], check=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-packet", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--stride", type=int, default=1, help="take every K-th packet of the corpus (the quick form used by tests/test_frontend_sanitized.py)")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if not args.child:
        rt = asan_runtime()
        if not rt:
            print("no ASan runtime in this image")
            return 2
        build()
        env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:allocator_may_return_null=1")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--per-packet", str(args.per_packet), "--seed", str(args.seed), "--stride", str(args.stride)], env=env)
        return p.returncode
    import numpy as np
    from edge264_amd import backend, front, packet as P, synth
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    pe = C.CDLL(os.path.join(HERE, "libe264_pred_emu_asan.so"))
    ie = C.CDLL(os.path.join(HERE, "libe264_intra_emu_asan.so"))
    VPP = C.POINTER(C.c_void_p)
    pe.e264emu_pred_frame2.argtypes = [C.c_void_p, VPP, C.c_void_p]
    pe.e264emu_dbkparam_frame.argtypes = [C.c_void_p, C.c_void_p]
    pe.e264emu_deblock_frame2.argtypes = [C.c_void_p, VPP, C.c_void_p, C.c_int]
    ie.e264emu_intra_frame.argtypes = [C.c_void_p, VPP]
    pe.e264emu_expand.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    pe.e264emu_set_expand.argtypes = [C.c_void_p]
    ie.e264emu_set_expand.argtypes = [C.c_void_p]
    # ---- the packets: what the front end makes of real streams + the synthetic generator's (every feature) ----
    packets = []
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams", "*.264"))):
        if os.path.getsize(f) < 60_000:
            packets += [bytes(p) for p in front.capture_packets(open(f, "rb").read())[0]]
    for seed, kw in ((5, dict(num_refs=2)), (6, dict(num_refs=3, t8x8=True)), (7, dict(num_refs=2, weighted=True))):
        try:
            packets += [bytes(p) for p in synth.StreamSynth(5, 4, seed=seed, **kw).gop("IPBPB")]
        except TypeError:
            packets += [bytes(p) for p in synth.StreamSynth(5, 4, seed=seed, num_refs=2).gop("IPBPB")]
    packets = packets[::max(1, args.stride)]
    rng = np.random.default_rng(args.seed)
    dpb_cache = {}

    def dpb_for(need, n_mbs):
        """the stream's allocations: every slot holds a picture of the ORIGINAL packet's declared size (+ 64: e264hip_frame_alloc)"""
        if (need, n_mbs) not in dpb_cache:
            arr = (C.c_void_p * 32)()
            for i in range(32):
                arr[i] = libc.malloc(need + 64)
                C.memset(arr[i], 100 + i, need + 64)
            dpb_cache[(need, n_mbs)] = (arr, libc.malloc(146 * n_mbs + 64))  # E264_SCRATCH_BYTES, exactly what ensure_dbk allocates
        return dpb_cache[(need, n_mbs)]
    tally = dict(packets=len(packets), mutations=0, accepted=0, rejected=0, rejected_against_the_slots=0, changed_nothing=0)
    by_section = {}

    def run_kernels(buf, need0, n_mbs):
        """the kernels' source on an exact-size heap copy of the packet (a read past its end is a heap-buffer-overflow); a wire packet (version 5) first
        through e264_expand_kernel's source into an expansion buffer of exactly e264_expand_area_bytes"""
        n = len(buf)
        mem = libc.malloc(n)
        C.memmove(mem, bytes(buf), n)
        area = None
        if buf[4] == 5:
            x = backend.packet_expand(bytes(buf))
            hx = P.Packet(x).hdr
            area = libc.malloc(int(hx["payload_off"]) - int(hx["mbs_off"]))
            pe.e264emu_expand(mem, area, 256)
        pe.e264emu_set_expand(area)
        ie.e264emu_set_expand(area)
        dpb, prm = dpb_for(need0, n_mbs)
        pe.e264emu_dbkparam_frame(mem, prm)
        pe.e264emu_pred_frame2(mem, dpb, None)
        ie.e264emu_intra_frame(mem, dpb)
        pe.e264emu_deblock_frame2(mem, dpb, prm, 1)
        pe.e264emu_set_expand(None)
        ie.e264emu_set_expand(None)
        libc.free(mem)
        if area:
            libc.free(area)

    for raw in packets:
        assert backend.packet_check(raw) == 0
        pk = P.Packet(raw)
        h = pk.hdr
        W, H = int(h["width_mbs"]), int(h["height_mbs"])
        need0 = int(h["plane_size_Y"]) + int(h["plane_size_C"])
        bounds = [("header", 0, 80), ("slices", int(h["slices_off"]), int(h["mbs_off"])), ("records", int(h["mbs_off"]), int(h["mbs_off"]) + 32 * W * H),
                  ("motion+payload", int(h["mbs_off"]) + 32 * W * H, len(raw))]
        bounds = [b for b in bounds if b[2] > b[1]]
        for _ in range(args.per_packet):
            name, lo, hi = bounds[int(rng.choice(len(bounds), p=np.array([0.1, 0.15, 0.45, 0.3])[:len(bounds)] / np.array([0.1, 0.15, 0.45, 0.3])[:len(bounds)].sum()))]
            buf = bytearray(raw)
            for _ in range(int(rng.integers(1, 5))):
                i = int(rng.integers(lo, hi))
                buf[i] = int(rng.integers(0, 256)) if rng.random() < 0.7 else (buf[i] ^ (1 << int(rng.integers(0, 8))))
            tally["mutations"] += 1
            s = by_section.setdefault(name, dict(accepted=0, rejected=0))
            if bytes(buf) == raw:
                tally["changed_nothing"] += 1
                continue
            if backend.packet_check(bytes(buf)) != 0:
                tally["rejected"] += 1
                s["rejected"] += 1
                continue
            hm = P.Packet(bytes(buf)).hdr
            # what e264hip_submit_* holds even a vetted packet against (check_slots_of): the picture it declares fits the stream's slots
            if int(hm["plane_size_Y"]) + int(hm["plane_size_C"]) > need0 or int(hm["width_mbs"]) * int(hm["height_mbs"]) > W * H:
                tally["rejected_against_the_slots"] += 1
                continue
            tally["accepted"] += 1
            s["accepted"] += 1
            run_kernels(buf, need0, W * H)
        # the same packet in its WIRE form (include/edge264_compact.h): damage in the header, the directory + bitmaps, the entries and the motion section
        wire = backend.packet_compact(raw)
        hwh = np.frombuffer(wire, P.FRAME_HDR, 1)[0]  # (header only: P.Packet() would unfold the packet)
        wb = [("wire header", 0, 80), ("wire table+entries+motion", int(hwh["mbs_off"]), int(hwh["payload_off"]))]
        run_kernels(bytearray(wire), need0, W * H)
        for _ in range(max(1, args.per_packet // 2)):
            name, lo, hi = wb[0] if rng.random() < 0.1 else wb[1]
            buf = bytearray(wire)
            for _ in range(int(rng.integers(1, 5))):
                i = int(rng.integers(lo, hi))
                buf[i] = int(rng.integers(0, 256)) if rng.random() < 0.7 else (buf[i] ^ (1 << int(rng.integers(0, 8))))
            tally["mutations"] += 1
            s = by_section.setdefault(name, dict(accepted=0, rejected=0))
            if bytes(buf) == wire:
                tally["changed_nothing"] += 1
                continue
            if backend.packet_check(bytes(buf)) != 0:
                tally["rejected"] += 1
                s["rejected"] += 1
                continue
            hm = np.frombuffer(bytes(buf), P.FRAME_HDR, 1)[0]
            if int(hm["plane_size_Y"]) + int(hm["plane_size_C"]) > need0 or int(hm["width_mbs"]) * int(hm["height_mbs"]) > W * H:
                tally["rejected_against_the_slots"] += 1
                continue
            tally["accepted"] += 1
            s["accepted"] += 1
            run_kernels(buf, need0, W * H)
    print("kernel_fuzz:", tally)
    for k, v in by_section.items():
        print(f"  bytes damaged in {k:15s} accepted {v['accepted']:6d}  rejected {v['rejected']:6d}")
    print("no sanitizer report: every packet the validation accepted kept the kernels inside their buffers")
    return 0


if __name__ == "__main__":
    sys.exit(main())

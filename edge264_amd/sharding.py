"""Multi-GPU layout of the reconstruction back end: streams are independent objects (a decoder
instance never reads another one's DPB), so N GPUs = N processes, each owning a contiguous shard
of the streams.  There is NO collective on the data path; torch.distributed is only used for the
barrier around the timed region and for the max-over-ranks of the elapsed time (bench.py).
"""
from __future__ import annotations

import os


def rank_info() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process per GPU)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_streams(n_streams: int, rank: int, world: int) -> range:
    """Stream ids owned by `rank` when `n_streams` are spread over `world` ranks (contiguous, sizes differ by <= 1)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def gpu_numa_cpus(local_rank: int) -> tuple[int, list[int]]:
    """(NUMA node, its CPUs) of GPU `local_rank` from sysfs: the PCI address of the HIP device -> numa_node -> cpulist.
    (-1, []) when the platform does not say (no GPU, single-node host, container without sysfs)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return -1, []
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = parse_cpulist(f.read())
        return node, cpus
    except Exception:  # noqa: BLE001 -- any missing piece means "unknown", never a failed benchmark
        return -1, []


def parse_cpulist(text: str) -> list[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out: list[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def bind_rank_to_gpu_socket(local_rank: int) -> dict:
    """Pins this rank's process (and every thread it creates afterwards: the back end's host pool, the submitters) to the
    CPUs of the NUMA node its GPU hangs off, so that packet staging and H2D copies do not cross the inter-socket link
    ('scaling should be linear until host parse cores or PCIe root complexes saturate').  Keeps the
    intersection with the affinity the launcher granted; does nothing when the node is unknown or the intersection is
    empty.  Returns what it did, for the bench line."""
    node, cpus = gpu_numa_cpus(local_rank)
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        return {"numa_node": node, "bound": False, "cpus": 0}
    want = allowed & set(cpus)
    if node < 0 or not want or want == allowed:
        return {"numa_node": node, "bound": False, "cpus": len(allowed)}
    os.sched_setaffinity(0, want)
    return {"numa_node": node, "bound": True, "cpus": len(want)}


def gather_rates(elapsed: float, frames: int, dist=None, device=None) -> list[float]:
    """frames/s of every rank (rank order): what the line reports as per-rank figures with min / max."""
    if dist is None or not dist.is_initialized():
        return [frames / elapsed if elapsed > 0 else 0.0]
    import torch
    mine = torch.tensor([frames / elapsed if elapsed > 0 else 0.0], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def reduce_elapsed(elapsed: float, frames: int, dist=None, device=None) -> tuple[float, int]:
    """max over ranks of the elapsed time, sum over ranks of the frames processed."""
    if dist is None or not dist.is_initialized():
        return elapsed, frames
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    n = torch.tensor([frames], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())

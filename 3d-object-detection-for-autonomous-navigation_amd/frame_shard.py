"""Frame-parallel sharding across the GPUs of a node: no data-path collective.

Frames are independent units (inference-mode BatchNorm, static anchors), so the
hot path shards by frame: rank r of W owns a contiguous block of frame ids and
its own engine handle.  The only cross-rank traffic is the barrier around the
timed region and a MAX-reduce of the elapsed time (bench.py contract); backend
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""


def rank_frames(rank, world, frames_per_rank):
    """Global frame ids owned by `rank` (contiguous block, weak scaling)."""
    if not (0 <= rank < world) or frames_per_rank < 0:
        raise ValueError("bad rank / world / frames_per_rank")
    start = rank * frames_per_rank
    return list(range(start, start + frames_per_rank))


def split_frames(n_frames, world):
    """Strong-scaling split of n_frames over `world` ranks: contiguous, sizes differ by <= 1."""
    base, extra = divmod(n_frames, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of a scalar (elapsed seconds).  dist is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(count, dist=None, device=None):
    """All ranks' frame counts (for whole-job throughput = sum(frames) / max(time))."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(count)]
    import torch
    t = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=device if device is not None else "cpu")
    t[dist.get_rank()] = int(count)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


# ---- NUMA placement of a rank ("one process per GPU, NUMA-pinned") ----
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(pci_bus_id, sysfs="/sys"):
    """NUMA node of the GPU at PCI address 'dddd:bb:dd.f' (sysfs numa_node), or None when the kernel does not say
    (-1: single-node machine or no affinity information)."""
    import os
    path = os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")
    try:
        node = int(open(path).read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


_saved_affinity = None      # {tid: mask} of the last pin, for restore_affinity()


def _all_task_ids():
    """Thread ids of this process (runtime / communicator threads started before the pin included)."""
    import os
    try:
        return sorted(int(t) for t in os.listdir("/proc/self/task"))
    except OSError:
        return [0]


def pin_to_gpu_numa_node(pci_bus_id, sysfs="/sys", apply=True):
    """Restricts EVERY thread of this process -- the ones the HIP runtime or the communicator have already started as
    well as the ones started later, which inherit the caller's mask -- to the CPUs of the GPU's NUMA node, BEFORE the
    rank allocates its page-locked staging pool: first-touch then places those pages on the node the GPU's host link
    hangs off, and the feeding thread stays next to them.  Only CPUs the process may already use are kept (cgroup /
    taskset limits are respected).  `restore_affinity()` undoes it (host-side legs that want the whole machine, e.g.
    a CPU baseline).  Returns a report dict; never raises -- a box without the sysfs files, or whose node has none of
    our CPUs, is left as it is."""
    import os
    global _saved_affinity
    rep = {"pci_bus_id": pci_bus_id, "numa_node": None, "pinned": False}
    try:
        node = gpu_numa_node(pci_bus_id, sysfs)
        rep["numa_node"] = node
        if node is None:
            rep["reason"] = "no NUMA affinity reported for the device"
            return rep
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
        allowed = os.sched_getaffinity(0)
        keep = cpus & allowed
        rep["node_cpus"] = len(cpus)
        if not keep:
            rep["reason"] = "none of the node's CPUs is in this process's affinity mask"
            return rep
        if apply and keep != allowed:
            saved, threads = {}, 0
            for tid in _all_task_ids():
                try:
                    saved[tid] = os.sched_getaffinity(tid)
                    os.sched_setaffinity(tid, keep & saved[tid] or keep)
                    threads += 1
                except OSError:          # the thread ended between the listing and the call
                    saved.pop(tid, None)
            _saved_affinity = saved
            rep["threads"] = threads
        rep["pinned"] = True
        rep["cpus"] = len(keep)
    except Exception as ex:   # noqa: BLE001 -- placement is an optimisation, never a failure
        rep["reason"] = repr(ex)
    return rep


def restore_affinity():
    """Gives every thread pinned by the last pin_to_gpu_numa_node() its previous CPU mask back (threads started since
    then get the calling thread's restored mask).  Returns the number of threads restored."""
    import os
    global _saved_affinity
    saved, _saved_affinity = _saved_affinity, None
    if not saved:
        return 0
    me = saved.get(os.getpid()) or next(iter(saved.values()))
    n = 0
    for tid in _all_task_ids():
        try:
            os.sched_setaffinity(tid, saved.get(tid, me))
            n += 1
        except OSError:
            pass
    return n


def device_pci_bus_id(local_rank):
    """'dddd:bb:dd.f' of HIP device `local_rank` through torch's device properties (no kernel is launched)."""
    import torch
    p = torch.cuda.get_device_properties(local_rank)
    return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"

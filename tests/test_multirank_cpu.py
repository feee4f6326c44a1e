"""World-size-2 gloo rehearsal of the N>1 path of bench.py (CPU; no GPU work).

The data path has no collective: ranks own disjoint frames.  What crosses ranks
is the barrier, the MAX of the elapsed time and the frame counts -- exercised
here exactly as bench.py calls them.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pp_amd as pp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = pp.frame_shard.rank_frames(rank, world, 4)
    clouds = [pp.synth.d435i_cloud(i, 256) for i in ids]          # each rank builds only its own frames
    dist.barrier()
    elapsed = 1.0 + rank                                            # pretend rank 1 was slower
    mx = pp.frame_shard.max_over_ranks(elapsed, dist)
    counts = pp.frame_shard.gather_counts(len(clouds) * 3, dist)
    dist.barrier()
    q.put((rank, ids, float(clouds[0][0, 0]), mx, counts))
    dist.destroy_process_group()


def test_two_rank_frame_sharding_and_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ids0, x0, mx0, c0), (r1, ids1, x1, mx1, c1) = res
    assert ids0 == [0, 1, 2, 3] and ids1 == [4, 5, 6, 7], "disjoint contiguous frame blocks"
    assert x0 != x1, "ranks process different frames"
    assert mx0 == mx1 == 2.0, "MAX over ranks of the elapsed time"
    assert c0 == c1 == [12, 12]
    fps = sum(c0) / mx0
    assert fps == 12.0, "whole-job throughput = all frames / slowest rank"


def test_split_frames_strong_scaling(pp):
    assert pp.frame_shard.split_frames(512, 8) == [(i * 64, (i + 1) * 64) for i in range(8)]
    s = pp.frame_shard.split_frames(10, 4)
    assert s == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        pp.frame_shard.rank_frames(3, 2, 4)


def test_numa_pinning_from_sysfs(pp, tmp_path):
    """pin_to_gpu_numa_node on a fake sysfs tree: the rank keeps the CPUs of the GPU's node that it may use, reports
    what it did, and leaves the process alone when the kernel reports no affinity."""
    allowed = sorted(os.sched_getaffinity(0))
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:0c:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    half = allowed[:max(1, len(allowed) // 2)]
    (node / "cpulist").write_text(",".join(str(c) for c in half) + ",4090-4095\n")
    rep = pp.frame_shard.pin_to_gpu_numa_node("0000:0C:00.0", sysfs=str(tmp_path), apply=False)
    assert rep["numa_node"] == 1 and rep["pinned"] and rep["cpus"] == len(half) and rep["node_cpus"] == len(half) + 6
    try:
        rep = pp.frame_shard.pin_to_gpu_numa_node("0000:0c:00.0", sysfs=str(tmp_path), apply=True)
        assert rep["pinned"] and sorted(os.sched_getaffinity(0)) == half
        # a pin covers the threads that already exist (runtime / communicator threads), and can be undone
        assert pp.frame_shard.restore_affinity() >= 1 and sorted(os.sched_getaffinity(0)) == allowed
        import threading
        stop, seen = threading.Event(), {}

        def other():
            seen["tid"] = threading.get_native_id()
            stop.wait(30)
        t = threading.Thread(target=other)
        t.start()
        while "tid" not in seen:
            pass
        rep = pp.frame_shard.pin_to_gpu_numa_node("0000:0c:00.0", sysfs=str(tmp_path), apply=True)
        if len(half) < len(allowed):
            assert rep["threads"] >= 2 and sorted(os.sched_getaffinity(seen["tid"])) == half
            pp.frame_shard.restore_affinity()
            assert sorted(os.sched_getaffinity(seen["tid"])) == allowed
        stop.set()
        t.join()
    finally:
        pp.frame_shard.restore_affinity()
        os.sched_setaffinity(0, allowed)
    (dev / "numa_node").write_text("-1\n")
    rep = pp.frame_shard.pin_to_gpu_numa_node("0000:0c:00.0", sysfs=str(tmp_path))
    assert not rep["pinned"] and rep["numa_node"] is None
    rep = pp.frame_shard.pin_to_gpu_numa_node("0000:ff:00.0", sysfs=str(tmp_path))     # unknown device
    assert not rep["pinned"]
    assert pp.frame_shard._parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}


def _numa_worker(rank, sysfs, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pp_amd as pp
    bdf = ["0000:0c:00.0", "0000:8c:00.0"][rank]           # what device_pci_bus_id(local_rank) returns on the node
    rep = pp.frame_shard.pin_to_gpu_numa_node(bdf, sysfs=sysfs)
    q.put((rank, rep, sorted(os.sched_getaffinity(0))))


def test_two_ranks_on_different_numa_nodes_pin_differently(tmp_path):
    """bench.py --gpus N: every rank pins itself to ITS GPU's node.  Two GPUs on two nodes (fake sysfs, the CPUs this
    container has split in two): the ranks' reports (`config.numa`) and masks differ."""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("one CPU")
    halves = [allowed[:len(allowed) // 2], allowed[len(allowed) // 2:]]
    for node, bdf in enumerate(["0000:0c:00.0", "0000:8c:00.0"]):
        dev = tmp_path / "bus" / "pci" / "devices" / bdf
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{node}\n")
        nd = tmp_path / "devices" / "system" / "node" / f"node{node}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(",".join(str(c) for c in halves[node]) + "\n")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_numa_worker, args=(r, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, rep0, m0), (_, rep1, m1) = res
    assert rep0["pinned"] and rep1["pinned"]
    assert rep0["numa_node"] == 0 and rep1["numa_node"] == 1
    assert m0 == halves[0] and m1 == halves[1] and not set(m0) & set(m1)

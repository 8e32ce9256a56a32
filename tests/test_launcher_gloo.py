"""The multi-GPU layer on CPU: world_size 2 over gloo.  Checks the contiguous sharding, the one-collective
weight broadcast (raw blob bytes, layout derived from dims on every rank) and the ordered result gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from whisper_amd import hip, launcher
from whisper_amd.synthetic import dims_for


def test_shard_range():
    for n, w in [(8, 2), (7, 2), (64, 8), (3, 8), (0, 4), (9, 4)]:
        spans = [launcher.shard_range(n, r, w) for r in range(w)]
        covered = [i for b, e in spans for i in range(b, e)]
        assert covered == list(range(n))
        assert all(e - b <= (n + w - 1) // w for b, e in spans)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dims = dims_for("micro.en")
        _, total = hip.blob_layout(dims, hip.WH_F16)
        blob = None
        if rank == 0:
            g = torch.Generator().manual_seed(0)
            blob = torch.randint(0, 256, (total,), dtype=torch.uint8, generator=g)
        got = launcher.broadcast_weights(blob, dims, hip.WH_F16, torch.device("cpu"), dist)
        checksum = int(got.to(torch.int64).sum())
        items = list(range(11))
        res = launcher.run_sharded(items, lambda xs: [(rank, x * x) for x in xs], dist)
        costs = [10, 1, 1, 1, 9, 2, 2, 8, 3, 3, 1]
        bal = launcher.run_balanced(items, costs, lambda xs: [(rank, x * x) for x in xs], dist)

        class _Model:                                  # transcribe_sharded only hands the model through
            pass
        import whisper_amd.transcribe as tr_mod        # noqa: F401  (module object lives in sys.modules)
        import sys
        mod = sys.modules["whisper_amd.transcribe"]
        real = mod.transcribe_batch
        mod.transcribe_batch = lambda model, part, batch_size=16, **kw: [
            {"text": f"{int(a.shape[-1])}", "rank": rank, "bs": batch_size, "kw": sorted(kw)} for a in part]
        try:
            files = [torch.zeros(n) for n in (160000, 16000, 480000, 32000, 320000)]
            tr = launcher.transcribe_sharded(_Model(), files, dist, batch_size=4, language="en")
        finally:
            mod.transcribe_batch = real
        q.put((rank, checksum, res, bal, tr))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, res0, bal0, tr0), (r1, c1, res1, bal1, tr1) = out
    assert c0 == c1 != 0                       # both ranks hold the same blob bytes
    assert res1 is None                        # only rank 0 gets the gathered results
    assert [v for _, v in res0] == [x * x for x in range(11)]          # original order
    assert [r for r, _ in res0] == [0] * 6 + [1] * 5                    # contiguous shards: ceil(11/2) = 6
    # cost-balanced shards (longest first onto the least loaded rank), results back in input order
    assert bal1 is None and [v for _, v in bal0] == [x * x for x in range(11)]
    costs = [10, 1, 1, 1, 9, 2, 2, 8, 3, 3, 1]
    load = [sum(c for (r, _), c in zip(bal0, costs) if r == k) for k in range(2)]
    assert sorted(load) == [20, 21]
    # transcribe_sharded: files balanced by length, every file transcribed once, input order kept on rank 0
    assert tr1 is None and [d["text"] for d in tr0] == ["160000", "16000", "480000", "32000", "320000"]
    assert {d["rank"] for d in tr0} == {0, 1} and all(d["bs"] == 4 and d["kw"] == ["language"] for d in tr0)
    assert [d["rank"] for d in tr0] == [1, 1, 0, 0, 1]       # 480000 + 32000 | 320000 + 160000 + 16000


def test_balanced_shards():
    assert launcher.balanced_shards([10, 1, 1, 1, 9, 2, 2, 8], 3) == [[0, 1, 3], [4, 6], [2, 5, 7]]
    assert launcher.balanced_shards([], 2) == [[], []]
    assert launcher.balanced_shards([5.0], 4) == [[0], [], [], []]
    for n, w in [(17, 4), (5, 8), (64, 8)]:
        costs = [(i * 37) % 11 + 1 for i in range(n)]
        shards = launcher.balanced_shards(costs, w)
        assert sorted(i for s in shards for i in s) == list(range(n))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs)


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without a launcher re-executes itself as the driver would launch it: one process per
    GPU under torch.distributed.run, one node, rendezvous on 127.0.0.1, every bench flag carried over, the caller's
    environment kept and the dmabuf-IPC switch RCCL needs on this driver set"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ["WH_TEST_SENTINEL"] = "kept"
    try:
        cmd, env = bench.torchrun_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29777)
    finally:
        del os.environ["WH_TEST_SENTINEL"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    script = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert env["WH_TEST_SENTINEL"] == "kept" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert env["MASTER_ADDR"] == "127.0.0.1"


def _worker_schedule(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        sys.argv = ["bench.py", "--gpus", str(world), "--steps", "20", "--warmup", "5"]
        mine = bench.chain_schedule(bench.parse(), dist.get_world_size())
        every = [None] * world
        dist.all_gather_object(every, mine)
        q.put((rank, every))
    finally:
        dist.destroy_process_group()


def test_bench_schedule_is_the_same_on_every_rank_and_at_every_world_size():
    """round 5's bench capped the lanes of a rank at usable host cores // world size, so `--gpus 8` ran less
    per GPU than the `--gpus 1` line it is compared with.  The schedule of a rank (passes coalesced per chain, chains in flight,
    clips resident, host threads) is now a function of the command line alone: the same on both ranks of a world-2 job (gloo), the
    same as a single rank's and an 8-rank job's, and it does not look at the core count."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv = sys.argv
    try:
        sys.argv = ["bench.py", "--steps", "20", "--warmup", "5"]
        one = bench.chain_schedule(bench.parse(), 1)
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
        eight = bench.chain_schedule(bench.parse(), 8)
    finally:
        sys.argv = argv
    assert one == eight == {"chain_batches": 3, "chains_in_flight": 1, "chain_rows": 24, "clips_resident": 24, "host_threads": 1}
    import inspect
    assert "usable_cores" not in inspect.getsource(bench.chain_schedule) and "cpu_count" not in inspect.getsource(bench.chain_schedule)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_schedule, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p_ in procs:
        p_.join(60)
    for rank, every in got:
        assert every == [one, one], (rank, every)


def _worker_large(rank, world, port, q):
    """the weight broadcast with the REAL byte layout of large-v3 (3.1 GB fp16 blob, sizes only: zeros + one marker per
    packed tensor + the header pack_weights writes), world 2 over gloo"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dims = dims_for("large-v3")
        layout, total = hip.blob_layout(dims, hip.WH_F16)
        blob = None
        if rank == 0:
            blob = torch.zeros(total, dtype=torch.uint8)
            for k, (name, (off, shape, mat)) in enumerate(layout.items()):
                blob[off: off + 8].view(torch.int64)[0] = k + 1                  # a marker at the start of every tensor
            flags = hip.WH_WEIGHTS_DEC_LN_FOLDED | hip.WH_WEIGHTS_ENC_QK_SCALED
            blob[total - 64: total - 48].view(torch.int32).copy_(torch.tensor([hip.BLOB_MAGIC, hip.WH_F16, flags, 0], dtype=torch.int32))
        got = launcher.broadcast_weights(blob, dims, hip.WH_F16, torch.device("cpu"), dist)
        marks = [int(got[off: off + 8].view(torch.int64)[0]) for off, _, _ in layout.values()]
        q.put((rank, int(got.numel()), int(got.view(torch.int64).sum()), marks == list(range(1, len(layout) + 1)),
               hip.blob_header(got, total)))
    finally:
        dist.destroy_process_group()


def test_broadcast_large_v3_layout_world2():
    """rank 1 receives exactly rank 0's bytes for the real large-v3 layout (every tensor's marker at its offset, the
    header with the packing flags at the tail), i.e. both ranks derive the same layout from the dims alone"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_large, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, c0, ok0, h0), (r1, n1, c1, ok1, h1) = out
    assert n0 == n1 > 3_000_000_000 and n0 % 8 == 0
    assert c0 == c1 != 0 and ok0 and ok1
    assert h0 == h1 == (hip.WH_F16, hip.WH_WEIGHTS_DEC_LN_FOLDED | hip.WH_WEIGHTS_ENC_QK_SCALED)


@pytest.mark.gpu
def test_broadcast_weights_nccl_world1(gpu_device):
    """the RCCL leg of the launcher on the real device: process group "nccl" (= RCCL on ROCm), world size 1 —
    `dist.broadcast` of a packed blob must run through the library and leave the bytes intact (a single-GPU box cannot
    show more; the N-rank path is the same call, covered on gloo above)"""
    import torch.distributed as dist
    from whisper_amd import hip
    from whisper_amd.launcher import broadcast_weights
    from whisper_amd.synthetic import dims_for, synthetic_state_dict
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dims = dims_for("micro.en")
    blob = hip.pack_weights(synthetic_state_dict(dims, seed=1), dims, hip.WH_F16, gpu_device)
    ref = blob.clone()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu_device)
    try:
        out = broadcast_weights(blob, dims, hip.WH_F16, gpu_device, dist)      # world 1: returned as is
        assert out is blob
        dist.broadcast(blob, src=0)                                             # the collective itself, on RCCL
        t = torch.ones(4, device=gpu_device)
        dist.all_reduce(t)
        torch.cuda.synchronize(gpu_device)
        assert torch.equal(blob, ref) and t.tolist() == [1.0] * 4
        assert dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()

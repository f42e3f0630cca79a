"""The data-parallel exchange step under the REAL backend ("RCCL all-gather of detections").

A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on one device), so this executes the collective path
with a world of ONE: ``init_process_group("nccl", world_size=1, device_id=...)``, ``DetectionGather(world=1,
force_collective=True)`` -> ``all_gather_into_tensor`` on the side stream behind an event, ``record_stream``, rotating
output slots.  What it pins: RCCL loads and runs, and the stream / event ordering of gather.py is right under the backend
whose collectives never block the host -- gathered == input for a sequence of distinct buffers produced on the compute
stream right before each gather and read back WITHOUT an explicit synchronisation (unpack() orders the reader).
Not a scaling measurement: N > 1 on hardware is still unmeasured."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from s2anet_amd.gather import DetectionGather, pack_detections


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_world_of_one():
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    yield dev
    dist.barrier()
    dist.destroy_process_group()


def _wire(seed, B, K, dev):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, K + 1, (B,), generator=g, dtype=torch.int32)
    dets = torch.rand(B, K, 6, generator=g)
    labels = torch.randint(0, 15, (B, K), generator=g, dtype=torch.int32)
    for b in range(B):
        dets[b, counts[b]:] = 0
        labels[b, counts[b]:] = -1
    return pack_detections(dets, labels, counts), (dets, labels, counts)


@pytest.mark.gpu
@pytest.mark.parametrize("side", [False, True])
def test_rccl_all_gather_world_of_one_equals_input(rccl_world_of_one, side):
    dev = rccl_world_of_one
    B, K = 8, 2000
    gather = DetectionGather(1, B, K, dev, side_stream=side, force_collective=True)
    assert gather.collective and (gather.stream is not None) == side
    host = [_wire(100 + i, B, K, dev) for i in range(6)]
    pinned = [h[0].pin_memory() for h in host]
    for i, (w_host, (d0, l0, c0)) in enumerate(host):
        # the wire buffer is PRODUCED on the compute stream right in front of the gather (an async copy + a kernel that
        # rewrites it), as the NMS finish kernel does in the detector; a gather that did not wait for it would send zeros
        wire = torch.zeros((B, K * 7 + 1), dtype=torch.float32, device=dev)
        big = torch.empty((64 << 20,), dtype=torch.float32, device=dev).normal_()      # keeps the compute stream busy
        big.mul_(1.0001)
        wire.copy_(pinned[i], non_blocking=True)
        wire.add_(0.0)
        got = gather(wire)                                       # [1*B, K*7+1], possibly still in flight on the side stream
        del wire                                                 # the allocator may recycle it: record_stream must hold it
        scratch = torch.full((B, K * 7 + 1), -7.0, dtype=torch.float32, device=dev)    # would land in the recycled block
        d, l, c = gather.unpack(got)                             # orders the CURRENT stream behind the gather: no sync
        assert torch.equal(d.cpu(), d0) and torch.equal(l.cpu(), l0) and torch.equal(c.cpu(), c0), i
        assert torch.equal(gather.out.view(B, -1).cpu(), w_host)
        del scratch, big
    # two gathers back to back on rotating slots: the first result is still intact after the second ran
    w0 = host[0][0].to(dev)
    w1 = host[1][0].to(dev)
    g0 = gather(w0)
    g1 = gather(w1)
    gather.wait()
    torch.cuda.synchronize()
    assert torch.equal(g1.cpu(), host[1][0])
    if side:
        assert torch.equal(g0.cpu(), host[0][0])                 # own slot (side-stream form keeps two)


@pytest.mark.gpu
def test_rccl_slot_reuse_waits_for_the_reader(rccl_world_of_one):
    """reader -> next writer: a slow reader on a THIRD stream still holds views of slot k when the gather two turns later
    is issued; the side stream must wait for everything that reader had queued before it overwrites the slot"""
    dev = rccl_world_of_one
    B, K = 8, 2000
    gather = DetectionGather(1, B, K, dev, side_stream=True, force_collective=True)
    host = [_wire(300 + i, B, K, dev) for i in range(3)]
    wires = [h[0].to(dev) for h in host]
    reader = torch.cuda.Stream(device=dev)
    big = torch.empty((96 << 20,), dtype=torch.float32, device=dev).normal_()
    torch.cuda.synchronize()
    g0 = gather(wires[0])                                         # slot 0
    with torch.cuda.stream(reader):
        d, l, c = gather.unpack(g0)                               # the reader stream now owns views of slot 0
        for _ in range(12):
            big.mul_(1.0000001)                                   # ~ milliseconds of work in front of the read
        kept = g0.clone()                                         # the read the next writer must not overtake
    gather(wires[1])                                              # slot 1
    gather(wires[2])                                              # slot 0 again: ordered behind the reader's queue
    assert gather.reader_waits == 1
    torch.cuda.synchronize()
    assert torch.equal(kept.cpu(), host[0][0])                    # the clone saw turn 0's data, not turn 2's
    assert torch.equal(gather.out.view(B, -1).cpu(), host[2][0])


@pytest.mark.gpu
def test_uneven_shard_pads_to_the_slot_size(rccl_world_of_one):
    """a rank holding fewer images than the largest shard pads with count -1 rows (world of one: 3 images in a 4-row slot)"""
    dev = rccl_world_of_one
    K = 50
    gather = DetectionGather(1, 4, K, dev, force_collective=True)
    w, (d0, l0, c0) = _wire(9, 3, K, dev)
    d, l, c = gather.unpack(gather(w.to(dev)))
    assert torch.equal(d[:3].cpu(), d0) and torch.equal(c.cpu()[:3], c0) and int(c[3]) == -1
    with pytest.raises(ValueError):
        gather(torch.zeros((5, K * 7 + 1), device=dev))


@pytest.mark.gpu
def test_rccl_all_reduce_max_and_barrier(rccl_world_of_one):
    """the two other collectives bench.py issues around the timed region"""
    dev = rccl_world_of_one
    t = torch.tensor([3.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert t.item() == 3.25

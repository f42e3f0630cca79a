"""CPU, world_size 2, gloo: the detection all-gather of the data-parallel path.
Gathered result must equal the concatenation of the per-rank outputs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from s2anet_amd.gather import DetectionGather, pack_detections, shard_range, unpack_detections


def make_rank_output(rank, B, K):
    g = torch.Generator().manual_seed(100 + rank)
    counts = torch.randint(0, K + 1, (B,), generator=g, dtype=torch.int32)
    dets = torch.rand(B, K, 6, generator=g)
    labels = torch.randint(0, 15, (B, K), generator=g, dtype=torch.int32)
    for b in range(B):
        dets[b, counts[b]:] = 0
        labels[b, counts[b]:] = -1
    return dets, labels, counts


def worker(rank, world, port, B, K, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gather = DetectionGather(world, B, K, torch.device("cpu"))
    d0, l0, c0 = make_rank_output(rank, B, K)
    wire = gather(pack_detections(d0, l0, c0))            # the detector hands over its wire buffer as it is
    assert tuple(wire.shape) == (world * B, K * 7 + 1)
    d, l, c = gather.unpack(wire)
    d2, l2, c2 = gather.unpack(gather(d0, l0, c0))         # the triple form packs first: same result
    assert torch.equal(d, d2) and torch.equal(l, l2) and torch.equal(c, c2)
    q.put((rank, d.clone(), l.clone(), c.clone()))
    dist.barrier()
    dist.destroy_process_group()


def uneven_worker(rank, world, port, G, K, q):
    """a global batch the world does not divide: ranks hold different numbers of images"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_range(G, world, rank)
    gather = DetectionGather.for_global_batch(G, world, K, torch.device("cpu"))
    d0, l0, c0 = make_rank_output(rank, b - a, K)
    for _ in range(2):                                         # the staging rows are reused: second turn == first
        wire = gather(pack_detections(d0, l0, c0))
        assert tuple(wire.shape) == (world * gather.B, K * 7 + 1)
        d, l, c = gather.unpack_global(wire)
    raw_counts = gather.unpack(wire)[2].clone()
    q.put((rank, d.clone(), l.clone(), c.clone(), raw_counts))
    # too many rows for the slot size: refused with a message, not sent
    try:
        gather(torch.zeros((gather.B + 1, K * 7 + 1)))
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(target, world, n_items, *args):
    """start `world` gloo ranks of `target` on a fresh port and collect n_items queue entries; a rendezvous that fails because
    the probed port was taken in between (rare, seen once on a busy box) is retried on another port"""
    import queue
    import time
    last = None
    for attempt in range(5):
        if attempt:
            time.sleep(1.0 + attempt)      # (a busy box: give the previous world's sockets time to close)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
        for p in procs:
            p.start()
        got = []
        try:
            for _ in range(n_items):
                got.append(q.get(timeout=120))
        except queue.Empty:
            last = "timeout waiting for the ranks"
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
        if len(got) == n_items and all(p.exitcode == 0 for p in procs):
            return got
        last = (last or "") + " attempt %d: %d of %d items, exit codes %s;" % (attempt, len(got), n_items, [p.exitcode for p in procs])
    raise AssertionError("gloo world failed five times: %s" % last)


def test_uneven_shards_are_padded_and_stripped():
    """global batch 5 over 2 ranks (3 + 2 images): every rank sends 3 rows, the short one a "no image" row with count -1;
    unpack_global returns the 5 images in order, == the concatenation of the per-rank outputs"""
    world, G, K = 2, 5, 20
    got = _run_world(uneven_worker, world, 2 * world, G, K)
    exp = [make_rank_output(r, shard_range(G, world, r)[1] - shard_range(G, world, r)[0], K) for r in range(world)]
    ed, el, ec = (torch.cat([e[i] for e in exp]) for i in range(3))
    assert ed.shape[0] == G
    for item in got:
        if len(item) == 2:
            assert "sized for 3 per rank" in item[1], item
            continue
        _, d, l, c, raw = item
        assert torch.equal(d, ed) and torch.equal(l, el) and torch.equal(c, ec)
        assert raw.tolist()[:3] == ec[:3].tolist() and raw.tolist()[3:5] == ec[3:].tolist() and raw.tolist()[5] == -1


def test_shard_range_covers_batch():
    for gb, w in ((64, 8), (10, 4), (3, 8)):
        spans = [shard_range(gb, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_pack_roundtrip():
    d, l, c = make_rank_output(0, 3, 17)
    d2, l2, c2 = unpack_detections(pack_detections(d, l, c), 17)
    assert torch.equal(d, d2) and torch.equal(l, l2) and torch.equal(c, c2)


def test_all_gather_equals_concatenation():
    world, B, K = 2, 3, 50
    got = _run_world(worker, world, world, B, K)
    exp = [make_rank_output(r, B, K) for r in range(world)]
    ed = torch.cat([e[0] for e in exp])
    el = torch.cat([e[1] for e in exp])
    ec = torch.cat([e[2] for e in exp])
    for _, d, l, c in got:
        assert torch.equal(d, ed) and torch.equal(l, el) and torch.equal(c, ec)

"""Image-level data parallelism: one process per GPU, the batch is split
contiguously across ranks, weights are replicated, and the only exchange step is ONE all-gather
of the padded detections per step (RCCL over xGMI on the GPU box; gloo in the CPU tests).

The reference has no inference-time multi-GPU path; the correctness pin is
"gathered result == concatenation of the single-process outputs" (tests/test_gather_gloo.py).

Wire format per rank: float32 [B_local, max_per_img*7 + 1] = for every image its
(x, y, w, h, angle, score, label) rows, -1-label padded, followed by the detection count
(exact in f32 up to 2^24).  8 chips x 2000 x 7 x 4 B = 0.45 MB per rank: latency-bound, so it is
sent as one fused buffer rather than three tensors.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, world, rank):
    """contiguous split of a global batch; the first (global_batch % world) ranks take one extra"""
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_detections(dets, labels, counts):
    """three tensors -> the wire buffer (tests and callers that did not come through ``detect(..., return_wire=True)``;
    the detector itself never packs: its NMS finish kernel writes the wire buffer, s2a_nms_rotated_segmented_dets)"""
    B, K, _ = dets.shape
    buf = torch.empty((B, K * 7 + 1), dtype=torch.float32, device=dets.device)
    body = buf[:, :K * 7].view(B, K, 7)
    body[..., :6] = dets
    body[..., 6] = labels.to(torch.float32)
    buf[:, K * 7] = counts.to(torch.float32)
    return buf


def unpack_detections(buf, max_per_img):
    K = max_per_img
    lead = buf.shape[:-1]
    body = buf[..., :K * 7].reshape(*lead, K, 7)
    return body[..., :6], body[..., 6].to(torch.int32), buf[..., K * 7].to(torch.int32)


class DetectionGather:
    """callable: this rank's wire buffer float32 [b, K*7+1] (or the (dets, labels, counts) triple, packed here) ->
    the gathered wire buffer [world*B, K*7+1] on every rank, rank-major order; ``unpack()`` turns it into
    (dets[world*B,K,6], labels[world*B,K], counts[world*B]).

    Uneven shards (a global batch the world does not divide: ``shard_range`` hands the first ranks one extra image):
    ``all_gather_into_tensor`` needs equal sizes, so every rank sends B = the LARGEST shard's rows (``for_global_batch``);
    a rank with fewer images (b < B) pads with rows whose count is -1 ("no image"), and ``unpack_global`` drops exactly
    those rows, giving the global batch in image order.  A wire buffer with MORE rows than B is refused.

    With ``side_stream=True`` the collective is issued on an own stream behind an event of the producing stream, so the
    next batch's trunk (issued on the producing stream right after) overlaps it; the rotating output slots keep a gather's
    result alive while the next one runs.  The tensor ``__call__`` returns is then still being written: ``unpack()`` and
    ``.out`` make the CURRENT stream wait for the last gather before they hand out views (``wait()`` does only that).

    Slot reuse is ordered on both sides: producer -> gather (the ``ready`` event) and reader -> next writer: ``unpack()`` /
    ``.out`` remember the stream the views were handed to, and before a later gather overwrites that slot the gathering
    stream waits for an event recorded on that stream at the moment of reuse -- everything the reader had queued on it by then
    (``slots`` turns later) is finished before the first byte is overwritten.  A reader that hands the views to yet another
    stream orders that hand-over itself, as with any tensor.

    ``force_collective=True`` sends a world of ONE through ``all_gather_into_tensor`` as well (instead of the copy): the
    only way to execute the RCCL branch, its side stream, events and ``record_stream`` on a one-GPU box."""

    def __init__(self, world, batch_local, max_per_img, device, group=None, side_stream=False, slots=2,
                 force_collective=False):
        self.world, self.B, self.K = world, batch_local, max_per_img
        self.group = group
        self.collective = world > 1 or bool(force_collective)
        self.cuda = torch.device(device).type == "cuda"
        self.outs = [torch.empty((world, batch_local, max_per_img * 7 + 1), dtype=torch.float32, device=device)
                     for _ in range(max(1, slots if side_stream else 1))]
        self.turn = 0
        self.stream = torch.cuda.Stream(device=device) if side_stream and self.cuda else None
        self.done = None
        self.readers = [None] * len(self.outs)      # per slot: the stream its views were last handed to
        self.pad = None                             # staging rows for a short shard
        self.reader_waits = 0                       # diagnostic: how many reuses were ordered behind a reader

    @classmethod
    def for_global_batch(cls, global_batch, world, max_per_img, device, **kw):
        """the gather of a global batch split by ``shard_range``: B = the largest shard (ceil(global / world))"""
        g = cls(world, -(-int(global_batch) // int(world)), max_per_img, device, **kw)
        g.global_batch = int(global_batch)
        return g

    @property
    def out(self):
        """the last gather's buffer, safe to read on the current stream"""
        self.wait()
        k = (self.turn - 1) % len(self.outs)
        self._note_reader(k)
        return self.outs[k]

    def _note_reader(self, k):
        if self.cuda:
            self.readers[k] = torch.cuda.current_stream()

    def _padded(self, wire):
        """a shard with fewer images than B: its rows followed by "no image" rows (count -1)"""
        b = wire.shape[0]
        if b == self.B:
            return wire
        if b > self.B:
            raise ValueError("DetectionGather: %d images on this rank but the gather was sized for %d per rank "
                             "(use DetectionGather.for_global_batch for uneven shards)" % (b, self.B))
        if self.pad is None:
            self.pad = torch.zeros((self.B, self.K * 7 + 1), dtype=torch.float32, device=wire.device)
            self.pad[:, self.K * 7] = -1.0
        self.pad[:b].copy_(wire)
        self.pad[b:, self.K * 7] = -1.0
        return self.pad

    def __call__(self, wire, labels=None, counts=None):
        if labels is not None:
            wire = pack_detections(wire, labels, counts)
        assert wire.dim() == 2 and wire.shape[1] == self.K * 7 + 1 and wire.dtype == torch.float32 and wire.is_contiguous()
        wire = self._padded(wire)
        k = self.turn % len(self.outs)
        out = self.outs[k]
        self.turn += 1
        reader, self.readers[k] = self.readers[k], None
        gstream = self.stream if self.stream is not None else (torch.cuda.current_stream() if self.cuda else None)
        if reader is not None and gstream is not None and reader != gstream:
            freed = torch.cuda.Event()
            freed.record(reader)                            # everything the reader queued on its stream up to now
            gstream.wait_event(freed)
            self.reader_waits += 1
        if not self.collective:
            out[0].copy_(wire)
        elif self.stream is None:
            dist.all_gather_into_tensor(out.view(-1), wire.view(-1), group=self.group)
        else:
            ready = torch.cuda.Event()
            ready.record()                                  # the NMS finish kernel that wrote `wire`
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                dist.all_gather_into_tensor(out.view(-1), wire.view(-1), group=self.group)
                wire.record_stream(self.stream)             # the allocator must not hand it out before the gather ran
                self.done = torch.cuda.Event()
                self.done.record()
        return out.view(self.world * self.B, -1)

    def wait(self):
        """current stream waits for the last side-stream gather (no-op otherwise)"""
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)

    def unpack(self, gathered=None):
        """views of the last gather (or of ``gathered``, a tensor ``__call__`` returned); waits for the side stream.
        Rows of a padded (short) shard carry count -1."""
        self.wait()
        if gathered is None:
            return unpack_detections(self.out.view(self.world * self.B, -1), self.K)
        for k, o in enumerate(self.outs):
            if gathered.data_ptr() == o.data_ptr():
                self._note_reader(k)
        return unpack_detections(gathered, self.K)

    def unpack_global(self, gathered=None):
        """the gathered batch without the pad rows of short shards, in global image order:
        (dets[G,K,6], labels[G,K], counts[G]) with G = the global batch (``for_global_batch``) -- a gather of copies"""
        d, l, c = self.unpack(gathered)
        G = getattr(self, "global_batch", self.world * self.B)
        if G == self.world * self.B:
            return d, l, c
        rows = []
        for r in range(self.world):
            a, b = shard_range(G, self.world, r)
            rows.extend(range(r * self.B, r * self.B + (b - a)))
        idx = torch.tensor(rows, dtype=torch.long, device=d.device)
        return d.index_select(0, idx), l.index_select(0, idx), c.index_select(0, idx)

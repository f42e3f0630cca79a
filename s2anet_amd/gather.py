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
    """callable: (dets[B,K,6], labels[B,K], counts[B]) of this rank ->
    (dets[world*B,K,6], labels[world*B,K], counts[world*B]) on every rank, rank-major order"""

    def __init__(self, world, batch_local, max_per_img, device, group=None):
        self.world, self.B, self.K = world, batch_local, max_per_img
        self.group = group
        self.out = torch.empty((world, batch_local, max_per_img * 7 + 1), dtype=torch.float32, device=device)

    def __call__(self, dets, labels, counts):
        buf = pack_detections(dets, labels, counts)
        if self.world == 1:
            self.out[0].copy_(buf)
        else:
            dist.all_gather_into_tensor(self.out.view(-1), buf.view(-1), group=self.group)
        d, l, c = unpack_detections(self.out.view(self.world * self.B, -1), self.K)
        return d, l, c

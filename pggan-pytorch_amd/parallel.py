"""Data-parallel sharding of the train step: one process per GPU, RCCL over xGMI.

The reference has no distributed code; the sharding is defined by BASELINE.json's north_star and
SURVEY.md §8e: each rank runs the full step on its own minibatch (DepthManager's minibatch size is
PER RANK — weak scaling), minibatch-stddev is evaluated on the local shard, and there is exactly one
exchange step per network per iteration: a SUM all-reduce of the live spans of the network's flat gradient buffer
(``backend='nccl'`` is RCCL on ROCm); the 1/world_size is folded into the fused Adam
(``FusedAdam.grad_scale``).  One or two collectives of up to 73 MB per network instead of one per tensor:
xGMI is point-to-point, so few large messages are what keeps the links busy."""
import os

import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, backend=None, device=None):
        if not dist.is_initialized():
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            dist.init_process_group(backend=backend)
        self.rank = dist.get_rank()
        self.world_size = dist.get_world_size()
        self.device = device

    @staticmethod
    def from_env(force=False):
        """torchrun / torch.distributed.run environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*).
        ``force``: build a (degenerate) one-rank group even when WORLD_SIZE is 1 — used to smoke-test the
        RCCL code path on a single GPU."""
        if int(os.environ.get('WORLD_SIZE', '1')) <= 1:
            if not force:
                return None
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('LOCAL_RANK', '0')
            os.environ['WORLD_SIZE'] = '1'
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        return DataParallel()

    @property
    def grad_scale(self):
        return 1.0 / self.world_size

    def broadcast_params(self, *nets):
        """Make every rank start from rank 0's weights: one broadcast of the flat parameter buffer plus
        one of the per-layer equalized-lr constants ``c`` (network.py:19 — data-dependent, not a
        parameter, so it must travel separately)."""
        for net in nets:
            dist.broadcast(net._flat_param, src=0)
            layers = net._layers()
            cs = torch.tensor([m.c for m in layers], dtype=torch.float64, device=net._flat_param.device)
            dist.broadcast(cs, src=0)
            for m, c in zip(layers, cs.tolist()):
                m.c = float(c)
            net.mark_params_changed()

    def all_reduce_flat(self, flat):
        """SUM all-reduce of one flat fp32 buffer (averaging happens in the optimizer)."""
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    def all_reduce_grads(self, net):
        """SUM all-reduce of the gradients of the layers that are live at the current growth stage: the parameters
        whose ``.grad`` the backward pass attached (a function of depth / alpha only, so every rank derives the same
        ranges) form a few contiguous spans of the flat gradient buffer — at 4x4 that is 26 MB instead of the whole
        73 MB buffer, at 1024x1024 everything, in one or two collectives."""
        if net._flat_grad is None:
            raise RuntimeError('all_reduce_grads called before any backward pass')
        flat = net._flat_grad
        for s, e in active_grad_spans(net):
            self.all_reduce_flat(flat[s:e])
        return flat

    def barrier(self):
        dist.barrier()


MERGE_GAP = 1 << 16        # spans closer than 256 KB travel in one collective (the gap is zeros: inactive layers)


def active_grad_spans(net):
    """[(start, end)] element ranges of net._flat_grad that hold the gradients attached by the last backward pass."""
    flat = net._flat_grad
    base, total = flat.data_ptr(), flat.numel()
    spans = []
    for p in net.parameters():
        g = p.grad
        if g is None:
            continue
        off = (g.data_ptr() - base) // 4
        if off < 0 or off + g.numel() > total:
            return [(0, total)]                              # a gradient outside the flat buffer: reduce everything
        spans.append((off, off + g.numel()))
    spans.sort()
    merged = []
    for s, e in spans:
        if merged and s - merged[-1][1] <= MERGE_GAP:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    return [(s, e) for s, e in merged]


def shard_seed(base_seed, rank):
    """Independent per-rank streams for latents / mixing factors / real batches."""
    return int(base_seed) + int(rank)

"""Data-parallel sharding of the train step: one process per GPU, RCCL over xGMI.

The reference has no distributed code; the sharding is defined by BASELINE.json's north_star and
SURVEY.md §8e: each rank runs the full step on its own minibatch (DepthManager's minibatch size is
PER RANK — weak scaling), minibatch-stddev is evaluated on the local shard, and there is exactly one
exchange step per network per iteration: a SUM all-reduce of the network's flat gradient buffer
(``backend='nccl'`` is RCCL on ROCm); the 1/world_size is folded into the fused Adam
(``FusedAdam.grad_scale``).  One collective of <=73 MB per network instead of one per tensor:
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
        if net._flat_grad is None:
            raise RuntimeError('all_reduce_grads called before any backward pass')
        return self.all_reduce_flat(net._flat_grad)

    def barrier(self):
        dist.barrier()


def shard_seed(base_seed, rank):
    """Independent per-rank streams for latents / mixing factors / real batches."""
    return int(base_seed) + int(rank)

"""Data-parallel sharding of the train step: one process per GPU, RCCL over xGMI.

The reference has no distributed code; the sharding is defined by BASELINE.json's north_star and SURVEY.md §8e: each
rank runs the full step on its own minibatch (DepthManager's minibatch size is PER RANK — weak scaling), minibatch-stddev
is evaluated on the local shard, and the only exchange is a SUM all-reduce of the live spans of each network's flat
gradient buffer (after reference trainer.py:98 for D, after :111 for G); the 1/world_size is folded into the fused Adam
(``FusedAdam.grad_scale``) or, for a foreign optimizer, applied right after the reduction.

Data plane: the library's own RCCL communicator, driven through the C-ABI (``pg_comm_init_rank`` /
``pg_allreduce_sum_f32``, include/pggan_hip.h) on a dedicated HIP stream.  ``torch.distributed`` is the control plane
only: rendezvous, the broadcast of the communicator id and of the initial weights, barriers.  (On a CPU-only host — the
world-size-2 gloo tests — the reduction itself goes through ``torch.distributed`` as well.)

Overlap: the backward sweeps of ``engine`` report every block whose weight gradients have been enqueued
(``net._grad_hook``); ``GradExchange`` collects them into buckets of ~``BUCKET_BYTES`` and issues each bucket's
all-reduce on the exchange stream behind an event of the weight-gradient stream, so the 512-channel blocks of D (85 % of
the bytes, finished in the first quarter of the sweep) travel under the rest of the backward pass and only the last
bucket is exposed.  xGMI is point-to-point (7 links per GPU): a few multi-MB messages keep the links busy, hence
buckets of tens of MB rather than one collective per tensor."""
import ctypes
import os

import torch
import torch.distributed as dist

def _want_hw_queues():
    """The data-parallel step uses three HIP streams (main, weight gradients, gradient exchange) next to RCCL's own; under
    ROCm's default of 4 hardware queues two of them share a queue and serialise (one-rank communicator: 17.2 vs 14.6 ms per
    1024^2 step).  GPU_MAX_HW_QUEUES is read when the HIP runtime initialises, i.e. at the first HIP call of the process:
    set it here when that is still ahead, say so when it is too late."""
    if 'GPU_MAX_HW_QUEUES' in os.environ:
        return
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        import warnings
        warnings.warn('pggan DataParallel: GPU_MAX_HW_QUEUES is not set and the HIP runtime is already initialised; the '
                      'main / weight-gradient / exchange streams will share hardware queues (~15 %% slower steps). Export '
                      'GPU_MAX_HW_QUEUES=8 before starting the process.')
        return
    os.environ['GPU_MAX_HW_QUEUES'] = '8'


if int(os.environ.get('WORLD_SIZE', '1')) > 1 or os.environ.get('PGGAN_FORCE_DP', '') == '1':
    _want_hw_queues()             # a torchrun rank: at import, before anything touched the device

BUCKET_BYTES = int(os.environ.get('PGGAN_DP_BUCKET_MB', '16')) << 20
MERGE_GAP = 1 << 16        # spans closer than 256 KB travel in one collective (the gap is zeros: inactive layers)


class DataParallel(object):
    def __init__(self, backend=None, device=None):
        _want_hw_queues()
        if not dist.is_initialized():
            if backend is None:
                backend = os.environ.get('PGGAN_DP_CONTROL') or ('nccl' if torch.cuda.is_available() else 'gloo')
            if backend == 'gloo' and os.environ.get('MASTER_ADDR', '') in ('127.0.0.1', 'localhost'):
                os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')       # single node: do not depend on the hostname resolving
            dist.init_process_group(backend=backend)
        self.rank = dist.get_rank()
        self.world_size = dist.get_world_size()
        self.device = device
        self.comm = None                      # ncclComm_t of the library's communicator (GPU only)
        self.comm_ranks = 0
        self._stream = None
        self.stats = dict(collectives=0, bytes=0)
        self.record_events = False            # bench.py: HIP events around every collective on the stream it runs on
        self.events = []
        self.skip_exchange = False            # bench.py only: time the step with the collectives left out (exposed-exchange estimate)
        if torch.cuda.is_available() and os.environ.get('PGGAN_DP_TORCH_ALLREDUCE', '0') != '1':
            self._init_comm()

    # ---------------------------------------------------------------- bootstrap
    def _init_comm(self):
        """RCCL communicator of the C-ABI: rank 0 draws the id, the control plane carries it, every rank joins."""
        from . import _lib
        lib = _lib.load()
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.pg_comm_unique_id(ident), 'pg_comm_unique_id')
        box = [ident.raw if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ident = ctypes.create_string_buffer(box[0], 128)
        comm = ctypes.c_void_p()
        _lib.check(lib.pg_comm_init_rank(ctypes.byref(comm), self.world_size, ident, self.rank), 'pg_comm_init_rank')
        n, r = ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.pg_comm_info(comm, ctypes.byref(n), ctypes.byref(r)), 'pg_comm_info')
        if n.value != self.world_size or r.value != self.rank:
            raise RuntimeError('RCCL communicator reports rank %d of %d, expected %d of %d'
                               % (r.value, n.value, self.rank, self.world_size))
        self.comm, self.comm_ranks = comm, n.value

    def close(self):
        if self.comm is not None:
            from . import _lib
            torch.cuda.synchronize()
            _lib.load().pg_comm_destroy(self.comm)
            self.comm = None

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
            if os.environ.get('PGGAN_DP_SHARE_GPU', '') == '1':
                # test aid, never a measurement: every rank on device 0 (with PGGAN_DP_CONTROL=gloo PGGAN_DP_TORCH_ALLREDUCE=1, RCCL
                # refuses two ranks on one device) -- checks that the ranks of a multi-process run issue matching collectives
                local = 0
            if local >= torch.cuda.device_count():
                raise RuntimeError('rank with LOCAL_RANK %d but only %d GPU(s) are visible' % (local, torch.cuda.device_count()))
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

    # ---------------------------------------------------------------- data plane
    def exchange_stream(self):
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        return self._stream

    def all_reduce_flat(self, flat, stream=None):
        """SUM all-reduce of one flat fp32 buffer, in place (averaging happens in the optimizer).  On the device: one
        ``pg_allreduce_sum_f32`` on ``stream`` (default: the current stream)."""
        self.stats['collectives'] += 1
        self.stats['bytes'] += flat.numel() * 4
        if self.skip_exchange:
            return flat
        if self.record_events and flat.is_cuda:
            s = stream if stream is not None else torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            self.record_events = False
            try:
                self.all_reduce_flat(flat, stream=stream)
                self.stats['collectives'] -= 1
                self.stats['bytes'] -= flat.numel() * 4
            finally:
                self.record_events = True
            e1.record(s)
            self.events.append((e0, e1))
            return flat
        if self.comm is not None and flat.is_cuda:
            from . import _lib
            if flat.dtype != torch.float32 or not flat.is_contiguous():
                raise RuntimeError('all_reduce_flat expects a contiguous float32 buffer')
            if flat.numel() == 0:
                return flat
            s = stream if stream is not None else torch.cuda.current_stream()
            _lib.call('pg_allreduce_sum_f32', self.comm, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                      ctypes.c_void_p(s.cuda_stream))
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    def all_gather_rows(self, rows):
        """[world, *rows.shape]: every rank's ``rows`` (a small statistics tensor), rank-major, identical on all ranks: a SUM all-reduce of a
        zero-filled buffer in which each rank fills its own slice (x + 0 is exact, so this IS an all-gather; one collective primitive in the
        C-ABI).  Current stream."""
        buf = torch.zeros((self.world_size,) + tuple(rows.shape), dtype=rows.dtype, device=rows.device)
        buf[self.rank].copy_(rows)
        self.all_reduce_flat(buf.view(-1))
        return buf

    def all_reduce_grads(self, net, average=False):
        """Exchange of one network's gradients.  The parameters whose ``.grad`` the backward pass attached (a function of
        depth / alpha only, so every rank derives the same ranges) form a few contiguous spans of the flat gradient
        buffer — at 4x4 that is 26 MB instead of the whole 73 MB.  Spans a ``GradExchange`` already sent while the
        backward pass was still running are not sent again; the current stream then waits for the exchange stream.
        ``average``: also scale by 1/world (for optimizers without a gradient pre-scale)."""
        if net._flat_grad is None:
            raise RuntimeError('all_reduce_grads called before any backward pass')
        flat = net._flat_grad
        ex = getattr(net, '_grad_exchange', None)
        if ex is not None and ex.started:
            ex.finish()
        else:
            for s, e in active_grad_spans(net):
                self.all_reduce_flat(flat[s:e])
        if average and self.world_size > 1:
            for s, e in active_grad_spans(net):
                flat[s:e].mul_(self.grad_scale)
        return flat

    def barrier(self):
        dist.barrier()

    def assert_same_on_all_ranks(self, value, what):
        """Control plane: raises on every rank unless all ranks hold the same integer (one MAX all-reduce of (v, -v))."""
        dev = 'cuda' if (dist.get_backend() == 'nccl' and torch.cuda.is_available()) else 'cpu'
        t = torch.tensor([int(value), -int(value)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hi, lo = int(t[0]), -int(t[1])
        if hi != lo:
            raise RuntimeError('%s differs between the ranks (%d .. %d; rank %d has %d)' % (what, lo, hi, self.rank, int(value)))


class GradExchange(object):
    """Bucketed, overlapped exchange of one network's gradients during its backward sweep.

    ``begin(active_layers)`` before the sweep; the sweep calls ``ready(layers)`` whenever the weight gradients of some
    layers have all been enqueued; ``finish()`` sends what is left and makes the current stream wait for the exchange
    stream.  A bucket is a run of layers that are neighbours in the flat gradient buffer among the ACTIVE layers (what
    lies between two active neighbours belongs to layers that are not live at this growth stage: zeros), so no
    collective ever covers a gradient that is still being accumulated."""

    def __init__(self, dp, net, bucket_bytes=None):
        self.dp, self.net = dp, net
        self.bucket_bytes = BUCKET_BYTES if bucket_bytes is None else bucket_bytes
        self.started = False
        self.sent_bytes = 0
        self.buckets = 0

    @staticmethod
    def _span(net, layer):
        base = net._flat_grad.data_ptr()
        lo = min(layer._gw.data_ptr(), layer._gb.data_ptr())
        hi = max(layer._gw.data_ptr() + layer._gw.numel() * 4, layer._gb.data_ptr() + layer._gb.numel() * 4)
        return (lo - base) // 4, (hi - base) // 4

    def begin(self, active_layers):
        net = self.net
        spans = sorted((self._span(net, m) + (id(m),)) for m in active_layers)
        self.order = {lid: i for i, (_, _, lid) in enumerate(spans)}
        self.spans = [(s, e) for s, e, _ in spans]
        self.state = [0] * len(spans)            # 0 pending, 1 ready, 2 sent
        self.ready_bytes = 0
        self.started = True
        self.sent_bytes = self.buckets = 0
        self.side = None

    def ready(self, layers, side_stream=None):
        """``layers``: every gradient of theirs has been enqueued (on ``side_stream`` when the weight-gradient stream is
        in use, else on the current stream)."""
        if not self.started:
            return
        self.side = side_stream
        for m in layers:
            i = self.order.get(id(m))
            if i is not None and self.state[i] == 0:
                self.state[i] = 1
                self.ready_bytes += (self.spans[i][1] - self.spans[i][0]) * 4
        if self.ready_bytes >= self.bucket_bytes:
            self._flush()

    def _flush(self):
        flat = self.net._flat_grad
        runs, i, n = [], 0, len(self.spans)
        while i < n:
            if self.state[i] != 1:
                i += 1
                continue
            j = i
            while j + 1 < n and self.state[j + 1] == 1:
                j += 1
            runs.append((self.spans[i][0], self.spans[j][1]))
            for k in range(i, j + 1):
                self.state[k] = 2
            i = j + 1
        self.ready_bytes = 0
        if not runs:
            return
        if flat.is_cuda:
            from . import engine
            ex = self.dp.exchange_stream()
            src = self.side if self.side is not None else torch.cuda.current_stream()
            engine._wait_stream(ex, src)          # behind the weight-gradient launches enqueued so far (through engine's helper: a launch plan records the edge)
            with torch.cuda.stream(ex):
                for s, e in runs:
                    self.dp.all_reduce_flat(flat[s:e], stream=ex)
        else:
            for s, e in runs:
                self.dp.all_reduce_flat(flat[s:e])
        self.buckets += len(runs)
        self.sent_bytes += sum(e - s for s, e in runs) * 4

    # A launch plan (plans.py) that was recorded with this exchange open replays the bucket collectives the sweep issued; the host-side
    # bookkeeping they went with is put back from a snapshot taken at the end of the recording, so that ``finish()`` -- which runs
    # eagerly, outside the plan -- sends exactly what the replayed sweep has not sent.
    def snapshot(self):
        return (list(self.state), self.ready_bytes, self.sent_bytes, self.buckets, self.side)

    def restore(self, snap):
        self.state, self.ready_bytes, self.sent_bytes, self.buckets, self.side = list(snap[0]), snap[1], snap[2], snap[3], snap[4]

    def finish(self):
        """Send the remaining layers (those never reported count as ready now: the sweep is over) and join."""
        if not self.started:
            return
        for i, st in enumerate(self.state):
            if st == 0:
                self.state[i] = 1
        if self.net._flat_grad.is_cuda:
            self.side = None                      # the caller's stream is already behind every gradient launch
        self._flush()
        if self.net._flat_grad.is_cuda:
            torch.cuda.current_stream().wait_stream(self.dp.exchange_stream())
        self.started = False


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

"""Data-parallel gradient exchange: one flat fp32 gradient buffer, one logical all-reduce per step.

The minibatch is sharded across ranks (one process per GPU; reference: Lightning DDP, commented
``devices: 4 / strategy: ddp`` in configs/demo_training.yaml:5-7, per-rank seed test_scripts/test_train.py:68-69).
Gradients are SUMMED over ranks here; the 1/world averaging is folded into the norm / AdamW kernels as
``pre_scale`` so no extra pass over the buffer is needed.

RCCL over xGMI is reached through ``torch.distributed`` (backend "nccl" == RCCL on ROCm).  The buffer is reduced
in a few large chunks on a dedicated communication stream; each chunk records an event, and the optimizer
kernel for chunk i waits only on event i, so AdamW on already-reduced chunks overlaps the reduction of the
rest (the mesh is point-to-point: 7 links x ~153 GB/s per GPU, so chunks are sized in the tens of MB, not the
25 MB DDP bucket default).  With gradient clipping the global norm needs every chunk first, so the optimizer
waits on the last event.  On CPU tensors (gloo, used by the tests) the same code runs synchronously.
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, world_size=None, chunk_elems=8 * 1024 * 1024, group=None):
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.chunk_elems = int(chunk_elems)
        self._comm_stream = None
        self.events = []

    @property
    def pre_scale(self):
        return 1.0 / self.world

    def chunks(self, n):
        c = self.chunk_elems
        return [(o, min(c, n - o)) for o in range(0, n, c)]

    def all_reduce(self, flat_grad: torch.Tensor):
        """Launch the (chunked) sum-all-reduce of ``flat_grad`` in place.  Returns the chunk list; on CUDA the
        work is asynchronous on the communication stream and ``self.events[i]`` marks chunk i reduced."""
        n = flat_grad.numel()
        chunks = self.chunks(n)
        self.events = []
        if self.world == 1:
            return chunks
        if flat_grad.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=flat_grad.device)
            cur = torch.cuda.current_stream(flat_grad.device)
            self._comm_stream.wait_stream(cur)  # backward finished producing the buffer
            with torch.cuda.stream(self._comm_stream):
                for off, ln in chunks:
                    dist.all_reduce(flat_grad[off:off + ln], op=dist.ReduceOp.SUM, group=self.group)
                    ev = torch.cuda.Event()
                    ev.record(self._comm_stream)
                    self.events.append(ev)
        else:
            for off, ln in chunks:
                dist.all_reduce(flat_grad[off:off + ln], op=dist.ReduceOp.SUM, group=self.group)
        return chunks

    def wait_chunk(self, i):
        if self.events:
            torch.cuda.current_stream().wait_event(self.events[i])

    def wait_all(self):
        if self.events:
            torch.cuda.current_stream().wait_event(self.events[-1])


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of ``n_items`` for ``rank`` (balanced, first ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)

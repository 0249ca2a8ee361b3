"""One independent sequence per GPU: the only way the update path scales.

Frame t's update consumes the state frame t-1 wrote, and one update is ~0.25 ms of GPU work, so nothing shards
inside a sequence.  What the reference does serially -- evaluate_tartan.py:87-141 runs its validation sequences one
after the other in one process and prints medians at :152-161 -- becomes one process per GPU, each a full replica of
the pipeline, plus ONE collective at the end: an all_gather of a few float64 per rank (trajectory metric, frames/s).
There is no data-path collective.

The group is a thin layer over torch.distributed so that the same code runs on RCCL over xGMI (backend "nccl" on
ROCm, one rank per GPU, bench.py) and on gloo on CPU (tests/test_replicas_gloo.py, world_size 2).
"""
import os

import torch


class ReplicaGroup:
    """World of independent replicas.  world == 1 needs no process group at all."""

    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = device if device is not None else torch.device("cpu")
        self.backend = backend
        self._own_pg = False
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                if backend is None:
                    backend = "nccl" if self.device.type == "cuda" else "gloo"
                self.backend = backend
                # the host driver only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                kw = {"device_id": self.device} if self.device.type == "cuda" else {}
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
                self._own_pg = True
            self._dist = dist

    # -- the sequence this rank tracks ------------------------------------------------------------------
    def sequence_seed(self, base=1234):
        """rank r <- sequence r (evaluate_tartan.py iterates scenes; seed 1234 as evaluate_tartan.py:203)"""
        return base + self.rank

    def shard(self, items):
        """round-robin assignment of independent sequences to ranks (no exchange between them)"""
        return list(items)[self.rank::self.world]

    # -- synchronisation around the timed region -----------------------------------------------------------
    def barrier(self):
        if self.world > 1:
            if self.device.type == "cuda":
                self._dist.barrier(device_ids=[self.local_rank])
            else:
                self._dist.barrier()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def max_over_ranks(self, value):
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        if self.world > 1:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    # -- the one collective: gather per-rank metrics ----------------------------------------------------------
    def gather_metrics(self, values):
        """values: this rank's metrics (list of floats, same length on all ranks) -> [world][len] on every rank"""
        mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.device)
        if self.world == 1:
            return [mine.tolist()]
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self._dist.all_gather(out, mine)
        return [o.tolist() for o in out]

    def close(self):
        if self.world > 1 and self._own_pg:
            self.barrier()
            self._dist.destroy_process_group()
            self._own_pg = False


def aggregate_rate(units_per_rank, elapsed_max, world):
    """whole-job throughput: every rank processed `units_per_rank` units; the job takes as long as its slowest rank"""
    return world * units_per_rank / elapsed_max


def summarise(per_rank):
    """rank-0 report over the gathered [metric, fps] rows, in the spirit of evaluate_tartan.py:152-161"""
    import numpy as np
    a = np.asarray(per_rank, dtype=np.float64)
    return {"median": np.median(a, axis=0).tolist(), "mean": a.mean(axis=0).tolist(), "min": a.min(axis=0).tolist(),
            "max": a.max(axis=0).tolist(), "ranks": int(a.shape[0])}

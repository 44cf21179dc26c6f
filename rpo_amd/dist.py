"""Data-parallel plumbing: one process per GPU, RCCL over xGMI (backend "nccl" on ROCm).

The reference's multi-GPU mode is ``nn.DataParallel`` (trainers/rpo.py:280-285),
which re-broadcasts the 150 M frozen parameters every step and cannot even
back-propagate RPO's per-replica scalar loss (SURVEY.md finding 8).  Here every
rank keeps the frozen backbone resident and never communicates it; images are
independent units, so the global batch is sharded evenly and the ONLY collective
per step is a sum all-reduce of the flat prompt-gradient buffer
([K*d_t + K*d_v] fp32 = 122 880 B at K=24).  Each rank's loss is the mean over
its shard; with equal shards the mean of shard means is the global mean, so
summing gradients and scaling by 1/world_size (folded into rpo_sgd_step's
grad_scale) reproduces single-process training on the global batch.

On CPU tensors the same code runs over gloo -- that is how tests/ cover it.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def _gpu_numa_node(idx: int) -> int:
    try:
        p = torch.cuda.get_device_properties(idx)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        return int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    except Exception:
        return -1


class GradSync:
    def __init__(self, backend: Optional[str] = None, init: bool = True):
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # Test hooks for boxes with fewer GPUs than ranks: RPO_ALL_RANKS_ON_GPU0=1 puts every rank on cuda:0 and
        # RPO_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); the N-rank control flow
        # (sharding, collectives, barriers, max-over-ranks timing) is then exercised end to end on one GPU.
        self.host_rank = self.local_rank            # the launcher's LOCAL_RANK: this process's share of the HOST (pin_host)
        if os.environ.get("RPO_ALL_RANKS_ON_GPU0") == "1":
            self.local_rank = 0
        # RPO_FORCE_DIST=1 runs the collective path even with one rank (exercises RCCL init / all-reduce /
        # barrier on a single-GPU box; the numbers are unchanged: sum over one rank, scale 1)
        self.enabled = self.world_size > 1 or os.environ.get("RPO_FORCE_DIST") == "1"
        # Only a distributed run moves the process-wide current device (to its local rank's GPU, for every backend:
        # kernels launch on the current device).  A lone process keeps whatever device the caller chose -- a trainer
        # built for cuda:1 must not be silently re-pointed at cuda:0.
        self.in_launcher = "LOCAL_RANK" in os.environ
        # (RPO_FORCE_DIST=1 in a lone process is NOT a distributed run: the collective path runs on whatever device the
        #  caller chose; advisor finding, round 3)
        self.pins_device = self.world_size > 1 or self.in_launcher
        if self.pins_device and torch.cuda.is_available() and torch.cuda.device_count() > self.local_rank:
            torch.cuda.set_device(self.local_rank)
        if self.enabled and init and not dist.is_initialized():
            if backend is None:
                backend = os.environ.get("RPO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl" and self.pins_device:
                torch.cuda.set_device(self.local_rank)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)

    @property
    def backend(self) -> str:
        return dist.get_backend() if (self.enabled and dist.is_initialized()) else "none"

    # ---- host side of an N-rank node: SURVEY 8e names host jitter as THE scaling risk (each rank replays five graphs and
    # one collective every ~3 ms), so under a launcher every rank pins itself to the cores of its GPU's NUMA node and
    # caps its host thread pools.  RPO_NO_AFFINITY=1 leaves the process alone.
    def pin_host(self, device_index: Optional[int] = None) -> dict:
        """Restrict this process to the CPUs next to its GPU (sysfs: the PCI device's numa_node -> that node's cpulist,
        intersected with what the process may use); without that information, an even share of the allowed CPUs by
        local rank.  Caps torch's intra-op threads at the share.  Returns what was done (for the bench line)."""
        info = {"pinned": False}
        if os.environ.get("RPO_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
            return info
        allowed = sorted(os.sched_getaffinity(0))
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world_size)))
        idx = self.local_rank if device_index is None else device_index
        cpus, how = None, None
        try:
            props = torch.cuda.get_device_properties(idx)
            bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
            node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
            if node >= 0:
                cl = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
                want = set()
                for part in cl.split(","):
                    a, _, b = part.partition("-")
                    want.update(range(int(a), int(b or a) + 1))
                # ranks whose GPUs share the node split its cores among themselves
                peers = sorted(set(range(local_world)))
                same = [r for r in peers if _gpu_numa_node(r) == node] or [self.local_rank]
                mine = sorted(want & set(allowed))
                if mine:
                    k = same.index(self.host_rank) if self.host_rank in same else 0
                    per = max(1, len(mine) // len(same))
                    cpus, how = mine[k * per:(k + 1) * per] or mine, f"numa node {node}"
        except Exception:
            cpus = None
        if not cpus and local_world > 1:
            per = max(1, len(allowed) // local_world)
            cpus, how = allowed[self.host_rank * per:(self.host_rank + 1) * per] or allowed, "even share of the allowed cpus"
        if cpus:
            try:
                os.sched_setaffinity(0, cpus)
                torch.set_num_threads(max(1, min(len(cpus), 8)))
                info = {"pinned": True, "cpus": len(cpus), "first_cpu": cpus[0], "how": how}
            except OSError:
                pass
        return info

    def _pg(self) -> None:
        if not dist.is_initialized():
            raise RuntimeError(f"WORLD_SIZE={self.world_size} but no process group is initialised: construct "
                               "GradSync() (init=True) or call torch.distributed.init_process_group first")

    @property
    def local_writer(self) -> bool:
        """True on the one rank per node that prepares node-shared files (local rank 0; with every rank forced onto
        GPU 0 by the test hook, global rank 0)."""
        if os.environ.get("RPO_ALL_RANKS_ON_GPU0") == "1":
            return self.rank == 0
        return int(os.environ.get("LOCAL_RANK", "0")) == 0

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world_size

    def describe(self) -> str:
        """What carries the per-step collective (for the bench line)."""
        if not self.enabled:
            return "none (single rank)"
        self._pg()
        b = dist.get_backend()
        return f"{'rccl' if b == 'nccl' else b} all-reduce of the flat prompt-gradient buffer, {dist.get_world_size()} ranks"

    def comm_world_size(self) -> int:
        """The world size the COMMUNICATOR reports (not the environment's): 1 for a lone process without a group."""
        return dist.get_world_size() if (self.enabled and dist.is_initialized()) else 1

    def rank_devices(self, device) -> list:
        """`rank r: <device name> (cuda:i, pci bus)` for every rank of the communicator, gathered through it."""
        try:
            p = torch.cuda.get_device_properties(device)
            mine = f"rank {self.rank}: {p.name} ({device}, pci {getattr(p, 'pci_bus_id', -1):02x})"
        except Exception:                            # noqa: BLE001 -- CPU-only test runs
            mine = f"rank {self.rank}: {device}"
        if not (self.enabled and dist.is_initialized()):
            return [mine]
        box = [None] * dist.get_world_size()
        dist.all_gather_object(box, mine)
        return box

    def shard(self, global_batch: int) -> Tuple[int, int]:
        """(first image, count) of this rank's contiguous shard; shards must be equal so that
        the mean of shard means is the global mean."""
        if global_batch % self.world_size != 0:
            raise ValueError(f"global batch {global_batch} not divisible by world size {self.world_size}")
        per = global_batch // self.world_size
        return self.rank * per, per

    def all_reduce_sum(self, flat: torch.Tensor) -> torch.Tensor:
        if self.enabled:
            self._pg()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.enabled:
            self._pg()
            dist.broadcast(t, src=src)
        return t

    def broadcast_object(self, obj, src: int = 0):
        """A small picklable object from rank `src` to every rank (start-up plumbing: a private directory name)."""
        if self.enabled:
            self._pg()
            box = [obj if self.rank == src else None]
            dist.broadcast_object_list(box, src=src)
            return box[0]
        return obj

    def gather_floats(self, value: float, device) -> list:
        """`value` of every rank, on every rank (bench: per-rank step times)."""
        if not self.enabled:
            return [float(value)]
        self._pg()
        t = torch.zeros(self.world_size, dtype=torch.float64, device=device)
        t[self.rank] = value
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def max_over_ranks(self, value: float, device) -> float:
        t = torch.tensor([value], dtype=torch.float64, device=device)
        if self.enabled:
            self._pg()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self) -> None:
        if self.enabled:
            self._pg()
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[self.local_rank])      # pins the barrier's collective to this rank's GPU
            else:
                dist.barrier()

    def close(self) -> None:
        if self.enabled and dist.is_initialized():
            dist.destroy_process_group()

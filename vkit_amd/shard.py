"""Sharding of independent work units (images, pages) over one process per GPU.

The path has no exchange step: every image is distorted on its own, exactly like the reference's
process pool hands whole pipeline runs to workers (reference: vkit/utility/pool.py:65-96, process_idx -> its own
rng / resources).  So there is no data-path collective; ``torch.distributed`` is used for the rendezvous, the
barriers that bracket a timed region and one MAX all-reduce of the elapsed time.  Backend ``nccl`` (RCCL) on the
GPUs, ``gloo`` in the CPU tests.
"""
import os
import time
from typing import Callable, Optional, Tuple


def world_from_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) for a plain launch."""
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


def weak_span(units_per_rank: int, rank: int) -> Tuple[int, int]:
    """Weak scaling: every rank owns ``units_per_rank`` consecutive global indices -> (first, count)."""
    if units_per_rank < 0 or rank < 0:
        raise ValueError('units_per_rank and rank must be >= 0')
    return rank * units_per_rank, units_per_rank


def strong_span(total_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Strong scaling: ``total_units`` split into ``world`` contiguous chunks whose sizes differ by at most one
    (the first ``total_units % world`` ranks get the extra unit) -> (first, count)."""
    if world <= 0 or not 0 <= rank < world or total_units < 0:
        raise ValueError(f'bad span request total={total_units} rank={rank} world={world}')
    base, extra = divmod(total_units, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def device_for(local_rank: int, n_devices: int) -> int:
    """process -> GPU, the pool's ``process_idx % n`` rule (reference: vkit/utility/pool.py:65-96)."""
    if n_devices <= 0:
        raise RuntimeError('no GPU visible')
    return local_rank % n_devices


def _parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_cpus_of_pci(bus_id: str, sysfs: str = '/sys'):
    """The CPUs of the NUMA node a PCI device (``0000:c1:00.0``) hangs off: ``<sysfs>/bus/pci/devices/<bdf>/numa_node`` ->
    ``<sysfs>/devices/system/node/node<N>/cpulist``.  None when the kernel does not say (numa_node -1: a single-node box, a VM)."""
    bdf = bus_id.strip().lower()
    if bdf.count(':') == 1:
        bdf = '0000:' + bdf
    try:
        with open(os.path.join(sysfs, 'bus', 'pci', 'devices', bdf, 'numa_node')) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, 'devices', 'system', 'node', f'node{node}', 'cpulist')) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError):
        return None


def bind_to_device_numa(bus_id: Optional[str], local_rank: int = 0, ranks_on_node: int = 1, sysfs: str = '/sys'):
    """Pins the calling process to the CPUs next to its GPU: the cores of the GPU's NUMA node, and of those the ``local_rank``-th
    of ``ranks_on_node`` equal shares when several ranks have their GPUs on the same node (8 GPUs on 2 sockets: 4 ranks per
    socket, each with a quarter of its cores) -- the page-locked staging buffers of a rank are then allocated from the memory of
    that socket and its copies do not cross the inter-socket link.  The reference's pool leaves placement to the OS
    (vkit/utility/pool.py:153-243); with host arrays in and out at 74 GB/s per GPU it is the first thing that decides whether 8
    ranks scale.  Returns the CPU set it bound to, or None (left alone) when the topology is unknown or VKX_NO_AFFINITY=1."""
    if os.environ.get('VKX_NO_AFFINITY') == '1' or not bus_id or not hasattr(os, 'sched_setaffinity'):
        return None
    cpus = numa_cpus_of_pci(bus_id, sysfs)
    if not cpus:
        return None
    allowed = cpus & set(os.sched_getaffinity(0))
    if not allowed:
        return None
    ordered = sorted(allowed)
    if ranks_on_node > 1:
        share = max(1, len(ordered) // ranks_on_node)
        mine = ordered[(local_rank % ranks_on_node) * share:(local_rank % ranks_on_node + 1) * share]
        if mine:
            ordered = mine
    os.sched_setaffinity(0, ordered)
    return set(ordered)


def ranks_sharing_numa(bus_ids, index: int, sysfs: str = '/sys'):
    """(position of device ``index`` among the devices on its NUMA node, number of devices on that node) for the list of the
    node's GPU bus ids in rank order; (0, 1) when the topology is unknown."""
    mine = numa_cpus_of_pci(bus_ids[index], sysfs)
    if not mine:
        return 0, 1
    same = [k for k, b in enumerate(bus_ids) if numa_cpus_of_pci(b, sysfs) == mine]
    return same.index(index), len(same)


class Group:
    """Rendezvous + the three collectives a sharded run needs (barrier, MAX of a float, SUM of an int)."""

    def __init__(self, backend: Optional[str] = None, device=None):
        self.rank, self.local_rank, self.world = world_from_env()
        self.backend = backend
        self.device = device
        self._dist = None
        if self.world > 1 and backend is None:
            raise ValueError('a backend is required when WORLD_SIZE > 1')
        # a single rank launched by torch.distributed.run (MASTER_ADDR set) still forms a group when a backend is
        # given, so that the 1-GPU driver run exercises the same RCCL path as the N-GPU one
        if backend is not None and (self.world > 1 or 'MASTER_ADDR' in os.environ):
            import torch.distributed as dist
            kwargs = {}
            if backend == 'nccl' and device is not None:
                kwargs['device_id'] = device
            dist.init_process_group(backend, **kwargs)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def _tensor(self, value, dtype):
        import torch
        dev = self.device if self.backend == 'nccl' and self.device is not None else 'cpu'
        return torch.tensor([value], dtype=dtype, device=dev)

    def max_float(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = self._tensor(value, torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, value: int) -> int:
        if self._dist is None:
            return int(value)
        import torch
        t = self._tensor(value, torch.int64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return int(t.item())

    def all_gather_object(self, obj):
        """One picklable object per rank, in rank order, on every rank."""
        if self._dist is None:
            return [obj]
        out = [None] * self.world
        self._dist.all_gather_object(out, obj)
        return out

    def evidence(self, **mine):
        """What a reader of an N > 1 result line needs to see that N ranks on N devices took part: the backend the process group
        reports, its world size, and one record per rank (rank, local rank, host, pid + whatever the caller adds, e.g. the PCI bus id
        of its GPU), all-gathered.  A plain launch reports itself as a world of one."""
        import socket
        rec = dict(rank=self.rank, local_rank=self.local_rank, host=socket.gethostname(), pid=os.getpid(), **mine)
        ranks = self.all_gather_object(rec)
        out = {'backend': self._dist.get_backend() if self._dist is not None else None,
               'world_size': self._dist.get_world_size() if self._dist is not None else 1, 'ranks': ranks}
        ids = [r.get('pci_bus_id') for r in ranks if r.get('pci_bus_id')]
        if ids:
            out['distinct_devices'] = len({(r['host'], r['pci_bus_id']) for r in ranks if r.get('pci_bus_id')})
        return out

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None


def timed_steps(group: Group, step: Callable[[], None], steps: int, warmup: int,
                device_sync: Callable[[], None]) -> float:
    """``warmup`` untimed passes, then exactly ``steps`` passes bracketed by barrier + device sync on both sides;
    returns the elapsed seconds, MAX over ranks."""
    for _ in range(warmup):
        step()
    device_sync()
    group.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    group.barrier()
    return group.max_float(time.perf_counter() - t0)

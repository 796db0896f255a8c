"""Question-axis sharding of one knowledge base over the GPUs of a node (one process per GPU).

The reference has no multi-device path (PqaCore/BaseCudaEngine.cpp:15 hard-wires device 0;
PqaCore/PqaEngineBaseFactory.cpp:85-91 `CreateGridEngine` is a stub).  The path shards naturally: the priority of a
question depends only on its own sA/mD rows plus the small replicated prior vector
(PqaCore/CEEvalQsSubtaskConsider.cpp:53-215), so every rank sweeps its contiguous question range
(SRPoolRunner::CalcSplit, SRPlatform/Interface/SRPoolRunner.h:96-110) and ONE tiny collective picks the global winner:
an all-gather of 16-byte (priority, global index) records over RCCL/xGMI followed by a local pick (exact, lowest
index on ties).  The message is 16 B per GPU, so the collective is latency-bound; ring bandwidth is irrelevant.
RecordAnswer runs on the rank that owns the answered question and the new prior vector (8*ldT bytes) is broadcast.

`torch.distributed` is plumbing only: backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests, where the
local selection comes from a caller-supplied function instead of the HIP engine.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_questions: int, world_size: int) -> List[int]:
    """End bound of each rank's contiguous question range; the reference's CalcSplit arithmetic."""
    quot, rem = divmod(n_questions, world_size)
    bounds, nxt = [], 0
    for r in range(world_size):
        nxt += quot + (1 if r < rem else 0)
        bounds.append(nxt)
    return bounds


def shard_range(n_questions: int, world_size: int, rank: int) -> Tuple[int, int]:
    b = shard_bounds(n_questions, world_size)
    return (0 if rank == 0 else b[rank - 1]), b[rank]


def pick_global(records) -> Tuple[float, int]:
    """records: [world, 2] float64 rows (priority, index-as-bits), a CPU tensor or numpy array.  Returns (priority,
    global question) of the maximum priority, lowest index on ties, -1 if no shard had an eligible question.  NaN never
    wins.  Plain Python over <= 8 records: this sits on the latency path of every selection."""
    arr = records.numpy() if isinstance(records, torch.Tensor) else records
    pris = arr[:, 0].tolist()
    idxs = arr[:, 1].copy().view("<i8").tolist()
    best_p, best_i = float("nan"), -1
    for p, i in zip(pris, idxs):
        if i < 0:
            continue
        if p != p:
            p = float("-inf")
        if best_i < 0 or p > best_p or (p == best_p and i < best_i):
            best_p, best_i = p, i
    return best_p, best_i


def pick_batch(all_winners) -> List[int]:
    """all_winners: [world, B, 2] float64 (priority, GLOBAL question index as a float, -1 = none), the ranks' per-quiz winners of
    one batched sweep (PqaHip_SelectArgmaxBatch), gathered.  Returns the B global picks: maximum priority, lowest index on ties,
    NaN never wins, -1 where no shard had an eligible question."""
    arr = all_winners.cpu().numpy() if isinstance(all_winners, torch.Tensor) else all_winners
    world, n_quizzes = arr.shape[0], arr.shape[1]
    picks = []
    for b in range(n_quizzes):
        best, best_p = -1, 0.0
        for r in range(world):
            p, qi = float(arr[r, b, 0]), int(arr[r, b, 1])
            if qi < 0:
                continue
            if p != p:
                p = float("-inf")
            if best < 0 or p > best_p or (p == best_p and qi < best):
                best, best_p = qi, p
        picks.append(best)
    return picks


def select_batch(local_winners: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> List[int]:
    """One batched selection over the shards: local_winners [B, 2] of this rank (device tensor under RCCL, CPU tensor under
    gloo) -> one all-gather of 16 B x B per rank -> the same B picks on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return pick_batch(local_winners.unsqueeze(0))
    parts = [torch.empty_like(local_winners) for _ in range(world)]
    dist.all_gather(parts, local_winners.contiguous(), group=group)   # (the list form: RCCL and gloo both have it)
    return pick_batch(torch.stack(parts))


class ShardedSelector:
    """Global next-question selection over question shards.

    local_select(out) must ENQUEUE (stream-ordered, no host sync needed) the local sweep + argmax and write the
    16-byte record (float64 priority, int64 GLOBAL index or -1) into `out`, a 2-element float64 tensor on the
    collective's device.  With the HIP engine this is `PqaHip_EnqueueSelectArgmax(engine, quiz, out.data_ptr())`.
    """

    def __init__(self, local_select: Callable[[torch.Tensor], None], device: torch.device,
                 group: Optional[dist.ProcessGroup] = None):
        self.local_select = local_select
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.local = torch.zeros(2, dtype=torch.float64, device=device)
        self.gathered = torch.zeros(self.world, 2, dtype=torch.float64, device=device)
        on_gpu = device.type == "cuda"
        # the gathered records land here: pinned, so the D2H copy is a single async DMA followed by one stream wait
        self.host = torch.zeros(self.world, 2, dtype=torch.float64, pin_memory=on_gpu)
        self._host_np = self.host.numpy()

    def enqueue(self) -> torch.Tensor:
        """Sweep + local argmax + all-gather, all stream-ordered; returns the [world,2] device tensor."""
        self.local_select(self.local)
        if self.world == 1:
            self.gathered[0].copy_(self.local)
        else:
            dist.all_gather_into_tensor(self.gathered.view(-1), self.local, group=self.group)
        return self.gathered

    def select(self) -> Tuple[float, int]:
        recs = self.enqueue()
        if recs.device.type == "cuda":
            self.host.copy_(recs, non_blocking=True)
            torch.cuda.current_stream(recs.device).synchronize()
            return pick_global(self._host_np)
        return pick_global(recs)


class ShmSelector:
    """Global next-question selection over question shards, the ranks' 16-byte records exchanged through a host
    shared-memory segment instead of a collective.

    Why: the message is 16 bytes per rank, so the exchange is pure latency.  An RCCL all-gather costs a kernel launch, the
    collective's own protocol, a D2H copy and a stream synchronisation per selection (~9 us even with ONE rank, measured;
    more with eight) on top of a ~15 us sweep.  Here the sweep's finisher writes {priority, GLOBAL index} and then a
    step number straight into rank r's 64-byte slot of a /dev/shm segment that every rank has mapped and registered with
    its GPU (PqaHip_HostRegister), and every rank's host spins on the `world` step numbers (PqaHip_PickWhenAll) and
    picks: no launch besides the sweep, no copy, no synchronisation.  Two slot sets alternate by step parity, so a fast
    rank's step s+1 never overwrites what a slow rank still reads for step s (a rank cannot finish s+1 before every
    rank has published s+1, which each does only after it is done with s).
    torch.distributed is not involved on the data path; it stays the control plane (rendezvous, barriers, timing).
    """

    SLOT = 64

    def __init__(self, engine, quiz: int, rank: int, world: int, name: str, create: Optional[bool] = None):
        import mmap
        import os

        from . import interop

        self.engine, self.quiz, self.rank, self.world = engine, quiz, rank, world
        self.path = "/dev/shm/pqa_select_%s" % name
        size = 2 * world * self.SLOT
        if create is None:
            create = rank == 0
        if create:
            fd = os.open(self.path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
            os.ftruncate(fd, size)          # zero-filled: step 0 is never used
        else:
            fd = os.open(self.path, os.O_RDWR)
        try:
            self._map = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        import ctypes

        self._host = ctypes.addressof(ctypes.c_char.from_buffer(self._map))
        self._dev = interop.host_register(self._host, size)
        self._interop = interop
        self.step = 0
        self._owner = create

    def select(self) -> Tuple[float, int]:
        self.step += 1
        half = (self.step & 1) * self.world * self.SLOT
        # (one call into the library: enqueue with this rank's slot as the destination, then the pick over all slots)
        return self.engine.select_through_slots(self.quiz, self._host + half, self._dev + half, self.rank, self.world, self.SLOT,
                                                self.step)

    def close(self) -> None:
        import os

        if self._map is not None:
            self._interop.host_unregister(self._host)
            self._map.close()
            self._map = None
            if self._owner:
                try:
                    os.unlink(self.path)
                except OSError:
                    pass


class ShmBatchExchange:
    """The ranks' per-quiz winners of one BATCHED sweep (PqaHip_SelectArgmaxBatch: [B, 2] = priority, GLOBAL index) exchanged
    through a host shared-memory segment: the records are on the host already when the batched call returns, so the exchange is
    16 B x B of stores per rank and a spin on `world` step numbers -- no collective launch, no H2D / D2H copy.  Two slot sets
    alternate by step parity (as ShmSelector).  x86 keeps stores in order: the records are written before the step number."""

    def __init__(self, n_quizzes: int, rank: int, world: int, name: str, create: Optional[bool] = None):
        import mmap
        import os

        import numpy as np

        self.B, self.rank, self.world = n_quizzes, rank, world
        self.stride = 16 * n_quizzes + 64               # records, then the step number on its own line
        self.path = "/dev/shm/pqa_batch_%s" % name
        size = 2 * world * self.stride
        if create is None:
            create = rank == 0
        fd = os.open(self.path, (os.O_CREAT | os.O_TRUNC | os.O_RDWR) if create else os.O_RDWR, 0o600)
        try:
            if create:
                os.ftruncate(fd, size)
            self._map = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        self._bytes = np.frombuffer(self._map, dtype=np.uint8)
        self.step = 0
        self._owner = create

    def _slot(self, half: int, r: int):
        import numpy as np

        off = (half * self.world + r) * self.stride
        recs = self._bytes[off:off + 16 * self.B].view(np.float64).reshape(self.B, 2)
        flag = self._bytes[off + 16 * self.B:off + 16 * self.B + 8].view(np.uint64)
        return recs, flag

    def exchange(self, local_winners, timeout_s: float = 600.0) -> List[int]:
        """local_winners: numpy [B, 2] of this rank -> the B global picks, the same on every rank."""
        import time

        import numpy as np

        self.step += 1
        half = self.step & 1
        recs, flag = self._slot(half, self.rank)
        recs[:] = local_winners
        flag[0] = self.step
        gathered = np.empty((self.world, self.B, 2), dtype=np.float64)
        t0 = time.perf_counter()
        for r in range(self.world):
            rr, ff = self._slot(half, r)
            while int(ff[0]) != self.step:
                if time.perf_counter() - t0 > timeout_s:
                    raise TimeoutError("rank %d never published step %d" % (r, self.step))
            gathered[r] = rr
        return pick_batch(gathered)

    def close(self) -> None:
        import os

        if self._map is not None:
            self._bytes = None
            try:
                self._map.close()
            except BufferError:
                pass
            self._map = None
            if self._owner:
                try:
                    os.unlink(self.path)
                except OSError:
                    pass


def broadcast_prior(prior: torch.Tensor, owner_rank: int, group: Optional[dist.ProcessGroup] = None) -> None:
    """After RecordAnswer on the owner of the answered question: replicate the new prior vector."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(prior, src=owner_rank, group=group)


def owner_of(question: int, n_questions: int, world_size: int) -> int:
    for r, b in enumerate(shard_bounds(n_questions, world_size)):
        if question < b:
            return r
    raise IndexError(question)


def tensor_from_device_ptr(ptr: int, n_doubles: int, device: torch.device) -> torch.Tensor:
    """Wrap engine-owned device memory (e.g. a quiz's prior vector) as a torch tensor without copying."""

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n_doubles,), "typestr": "<f8", "data": (ptr, False), "version": 3}
    return torch.as_tensor(h, device=device)

"""Periodic exchange of Q-deltas between table replicas, one process per GPU.

The reference scales out with a parameter server over pickled MPI messages
(``algorithms/runtime/q_learning_async_dist.py:164-357``): workers step environments, rank 0 owns
the table.  Here every GPU owns ``N/G`` agents, their environments AND a full replica of the table
(1e7 x 32 fp32 = 1.28 GB of 288 GB).  Agents never communicate; the only exchange step is the table:

* while it learns, the engine appends one ``{uint32 cell, float32 delta}`` record per agent-step to
  a log buffer owned by this class (fixed slot = step * n + agent, no atomics);
* every ``sync_every`` vector steps the logs are all-gathered (RCCL over xGMI through
  ``torch.distributed``; one collective of ``sync_every * n * 8`` bytes per GPU) and each GPU adds
  the OTHER GPUs' deltas into its replica (``qe_delta_apply_dev``: fp32 atomicAdd scatter).

Traffic per sync is proportional to the agent-steps taken, not to the table: 100 steps x 8192 agents
x 8 B = 6.5 MB per GPU against 2.2 GB for a dense all-reduce of the 1e7 x 32 table.

Overlap (default): the all-gather of chunk k is started asynchronously and completes on the
collective's own stream while chunk k+1 is being computed into the second log buffer; the remote
deltas of chunk k are applied when chunk k+1 ends.  Remote updates therefore become visible
``sync_every`` to ``2 * sync_every`` steps after they were made (bounded staleness -- the same
family of semantics as the reference's asynchronous runtimes, which apply updates whenever a worker
message happens to arrive).  ``overlap=False`` applies them at the sync point itself.

torch is plumbing here (device buffers + the RCCL/gloo collective); ``apply_fn`` is the engine's
scatter-add on GPU and an oracle function in the CPU (gloo) tests.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


class DeltaSync:
    def __init__(self, capacity: int, device, apply_fn, attach_fn=None, group=None, overlap: bool = True,
                 stream=None, apply_skip_fn=None, apply_sorted_fn=None, apply_gathered_fn=None) -> None:
        # (gathered buffer, capacity, count, world, rank): the whole deterministic apply step inside the engine --
        # radix sort of the other ranks' records by cell + one sequential run per cell, no torch compute
        self.apply_gathered_fn = apply_gathered_fn
        self.stream = stream  # torch.cuda.Stream the engine runs on (None on CPU / current stream)
        self.apply_skip_fn = apply_skip_fn  # (entries, total, skip_begin, skip_end): one-launch form
        # (entries sorted by cell, total): deterministic form -- every cell receives the other ranks'
        # increments sequentially in (rank, slot) order instead of through float atomics in arrival order
        self.apply_sorted_fn = apply_sorted_fn
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.capacity = int(capacity)
        self.apply_fn = apply_fn
        self.attach_fn = attach_fn  # called with the log tensor the engine must write next
        self.overlap = bool(overlap)
        n_buf = 2 if self.overlap else 1
        self.logs = [torch.zeros((self.capacity, 2), dtype=torch.int32, device=device) for _ in range(n_buf)]
        self.gathered = [
            torch.zeros((self.world, self.capacity, 2), dtype=torch.int32, device=device) for _ in range(n_buf)
        ]
        self.cur = 0
        self._inflight = None  # (work handle or None, buffer index, record count)
        self.bytes_exchanged = 0
        self.syncs = 0
        if self.attach_fn is not None:
            self.attach_fn(self.log)

    @property
    def log(self) -> torch.Tensor:
        """The buffer the engine is currently appending to."""
        return self.logs[self.cur]

    # ------------------------------------------------------------------ collective + apply
    def _start(self, buf: int, count: int):
        out, inp = self.gathered[buf], self.logs[buf]
        try:
            work = dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group, async_op=self.overlap)
        except (RuntimeError, NotImplementedError):  # backends without the flat form
            work = dist.all_gather([out[r] for r in range(self.world)], inp, group=self.group, async_op=self.overlap)
        return work

    def _finish(self, work, buf: int, count: int) -> None:
        if work is not None:
            work.wait()  # the current stream now waits for the collective; the host does not block
        g = self.gathered[buf]
        if self.world == 1:
            pass  # nobody else's records
        elif self.apply_gathered_fn is not None:
            self.apply_gathered_fn(g, self.capacity, count, self.world, self.rank)
        elif self.apply_sorted_fn is not None:
            others = torch.cat([g[r, :count] for r in range(self.world) if r != self.rank])
            order = torch.sort(others[:, 0], stable=True).indices  # by cell; ties keep (rank, slot) order
            self.apply_sorted_fn(others[order].contiguous(), int(others.shape[0]))
        elif count == self.capacity and self.apply_skip_fn is not None:
            # full segments are contiguous: every rank's records except my own in ONE launch
            self.apply_skip_fn(g.reshape(-1, 2), count * self.world, count * self.rank, count * (self.rank + 1))
        elif count == self.capacity and self.world > 2:
            # everything before / after my own segment in two launches
            if self.rank > 0:
                self.apply_fn(g[: self.rank].reshape(-1, 2), count * self.rank)
            if self.rank < self.world - 1:
                self.apply_fn(g[self.rank + 1:].reshape(-1, 2), count * (self.world - 1 - self.rank))
        else:
            for r in range(self.world):
                if r != self.rank:
                    self.apply_fn(g[r], count)
        self.bytes_exchanged += count * 8 * (self.world - 1)

    def _on_stream(self):
        import contextlib

        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def exchange(self, count: int) -> None:
        """Publish the first ``count`` records of the current log; apply what has arrived."""
        with self._on_stream():
            self._exchange(count)

    def flush(self) -> None:
        """Complete the exchange that is still in flight (end of a training call)."""
        with self._on_stream():
            if self._inflight is not None:
                self._finish(*self._inflight)
                self._inflight = None

    def _exchange(self, count: int) -> None:
        if count > self.capacity:
            msg = f"delta log overflow: {count} records, capacity {self.capacity}"
            raise RuntimeError(msg)
        if count == 0:
            return
        self.syncs += 1
        if not self.overlap:
            self._finish(self._start(0, count), 0, count)
            return
        previous = self._inflight
        self._inflight = (self._start(self.cur, count), self.cur, count)
        self.cur ^= 1
        if self.attach_fn is not None:
            self.attach_fn(self.log)  # the engine fills the other buffer while this one travels
        if previous is not None:
            self._finish(*previous)

def attach_engine(algorithm, sync_every: int, num_agents: int, group=None, overlap: bool = True,
                  deterministic: bool = True) -> DeltaSync:
    """Wire a :class:`DeltaSync` to a HIP engine living on the current CUDA device.

    ``deterministic`` (default ``True``): the other ranks' records are stably sorted by cell and added in
    (rank, slot) order per cell INSIDE the engine (``qe_delta_apply_gathered_dev``: hand-written radix sort +
    sequential runs, ``csrc/qe_delta_sort.h``), so a replica is reproducible bit for bit; ``"torch-sort"`` does
    the same with ``torch.cat`` / a stable ``torch.sort`` in front of ``qe_delta_apply_sorted_dev`` (round 2's
    path, kept to check the two against each other); ``False`` adds the records with float atomics in arrival
    order (one launch, no sort)."""
    import ctypes as C

    from dist_classicrl_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    # The engine and the collectives share ONE non-default torch stream, so kernels, all-gathers and
    # delta scatter-adds are stream-ordered.  (Not the legacy default stream: its implicit
    # synchronisation with every blocking stream would serialise the collective with the next chunk.)
    stream = torch.cuda.Stream(device=dev)
    _lib.check(lib.qe_set_stream(algorithm.handle, C.c_void_p(stream.cuda_stream)))
    capacity = sync_every * num_agents

    def apply_fn(entries, count):
        _lib.check(lib.qe_delta_apply_dev(algorithm.handle, C.c_void_p(entries.data_ptr()), int(count)))

    def attach_fn(log):
        _lib.check(lib.qe_delta_log_attach(algorithm.handle, C.c_void_p(log.data_ptr()), capacity))

    def apply_skip_fn(entries, total, skip_begin, skip_end):
        _lib.check(lib.qe_delta_apply_skip_dev(algorithm.handle, C.c_void_p(entries.data_ptr()), int(total),
                                               int(skip_begin), int(skip_end)))

    def apply_sorted_fn(entries, total):
        _lib.check(lib.qe_delta_apply_sorted_dev(algorithm.handle, C.c_void_p(entries.data_ptr()), int(total)))
        # (the sorted copy must outlive the kernel that reads it: keep it until the next exchange)
        apply_sorted_fn.keep = entries

    def apply_gathered_fn(gathered, cap, count, world, rank):
        _lib.check(lib.qe_delta_apply_gathered_dev(algorithm.handle, C.c_void_p(gathered.data_ptr()), int(cap), int(count),
                                                   int(world), int(rank)))

    sync = DeltaSync(capacity, dev, apply_fn, attach_fn, group, overlap, stream, apply_skip_fn,
                     apply_sorted_fn if deterministic == "torch-sort" else None,
                     apply_gathered_fn if deterministic is True else None)
    stream.wait_stream(torch.cuda.current_stream())  # buffer initialisation ran on the current stream
    return sync

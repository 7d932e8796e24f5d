"""Multi-walker exchange step: the delta-since-last-sync all-reduce of the
multicanonical weights and visit histograms (comms_mpi.f90:244-277 eta,
:461-494 hist, :496-530 uhist; reset hooks :533-566), on ``torch.distributed``.

One process per GPU; backend "nccl" is RCCL over xGMI on the 8xMI355X node,
"gloo" on CPU (tests).  Walkers are independent replicas (parallel_strategy
'mw', mc_moves.F90:711-718); this is the only data they ever exchange, every
``mpi_sync_int`` cycles (mc_moves.F90:258-276).

Semantics, per array, exactly as the reference: each rank contributes the
increment since the last synchronisation, the increments are summed over ranks
and added to the last synchronised value::

    w <- allreduce_sum(w - last) + last ;  last <- w

The first call therefore doubles as a broadcast when only rank 0 holds non-zero
weights (mc_moves.F90:738-776).  The reference issues three 808-byte
MPI_Allreduce calls (two of them behind an MPI_Barrier); at this size the cost is
all latency, so :meth:`WalkerComms.sync` packs the three increments into ONE
all-reduce (2.4 KB for nbins = 101) with element-wise identical results.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


class WalkerComms:
    def __init__(self, nbins, samplerun=True, device=None, group=None):
        self.nbins = int(nbins)
        self.samplerun = bool(samplerun)
        self.group = group
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        # comms_allocate (comms_mpi.f90:85-96)
        self.eta_last_sync = np.zeros(self.nbins)
        self.hist_last_sync = np.zeros(self.nbins)
        self.uhist_last_sync = np.zeros(self.nbins)
        self._buf = torch.zeros(3 * self.nbins, dtype=torch.float64, device=self.device)
        self._stage = torch.zeros(3 * self.nbins, dtype=torch.float64,
                                  pin_memory=self.device.type == "cuda")

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def _allreduce(self, delta):
        """Sum a float64 host vector over all walkers; returns a host vector."""
        n = len(delta)
        if not dist.is_initialized() or self.world_size == 1:
            return np.array(delta, dtype=np.float64)
        self._stage[:n].copy_(torch.from_numpy(np.ascontiguousarray(delta, dtype=np.float64)))
        buf = self._buf[:n]
        buf.copy_(self._stage[:n], non_blocking=True)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        self._stage[:n].copy_(buf)   # synchronising D2H
        return self._stage[:n].numpy().copy()

    def _delta_sync(self, arr, last):
        arr -= last                                   # comms_mpi.f90:256
        total = self._allreduce(arr)                  # :266
        arr[:] = total + last                         # :269
        last[:] = arr                                 # :270

    # the three reference routines, in place on float64 numpy arrays of length nbins
    def allreduce_eta(self, weight):
        self._delta_sync(weight, self.eta_last_sync)

    def allreduce_hist(self, histogram):
        self._delta_sync(histogram, self.hist_last_sync)

    def allreduce_uhist(self, unbiased_hist):
        self._delta_sync(unbiased_hist, self.uhist_last_sync)

    def set_histogram(self, hist_in):                 # comms_mpi.f90:533-548
        self.hist_last_sync[:] = hist_in

    def set_uhistogram(self, hist_in):                # comms_mpi.f90:551-566
        self.uhist_last_sync[:] = hist_in

    def sync(self, weight, histogram, unbiased_hist=None):
        """The mpi_sync_int block of mc_cycle (mc_moves.F90:264-268) as ONE collective."""
        parts = [(weight, self.eta_last_sync), (histogram, self.hist_last_sync)]
        if unbiased_hist is not None and self.samplerun:
            parts.append((unbiased_hist, self.uhist_last_sync))
        delta = np.concatenate([a - last for a, last in parts])
        total = self._allreduce(delta)
        for k, (a, last) in enumerate(parts):
            a[:] = total[k * self.nbins:(k + 1) * self.nbins] + last
            last[:] = a

    def barrier(self):                                # comms_barrier, comms_mpi.f90:601-618
        if dist.is_initialized() and self.world_size > 1:
            dist.barrier(group=self.group)

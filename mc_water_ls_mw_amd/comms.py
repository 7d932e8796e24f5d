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
        # The exchange has a HIP stream of its own, at high priority: its copies and the collective are a few microseconds
        # of work, and queued behind a millisecond kernel of the engine on the default priority they would start only
        # when that kernel ends -- the host, which waits for the result, could then not issue the next step ahead.
        self._stream = torch.cuda.Stream(device=self.device, priority=-1) if self.device.type == "cuda" else None

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def _allreduce(self, delta):
        """Sum a float64 host vector over all walkers; returns a host vector."""
        n = len(delta)
        if not dist.is_initialized():                 # no process group at all: a single walker process
            return np.array(delta, dtype=np.float64)
        # (with a process group the collective is issued even for one rank: N = 1 and N = 8 run the same code)
        self._stage[:n].copy_(torch.from_numpy(np.ascontiguousarray(delta, dtype=np.float64)))
        buf = self._buf[:n]
        if self._stream is None:
            buf.copy_(self._stage[:n])
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            self._stage[:n].copy_(buf)
        else:
            with torch.cuda.stream(self._stream):
                buf.copy_(self._stage[:n], non_blocking=True)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                self._stage[:n].copy_(buf, non_blocking=True)
            self._stream.synchronize()
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

    def set_weights(self, weight_in):
        """Re-base the weights' delta scheme (no reference counterpart: used when a farm re-gauges the synchronised
        table, see WalkerFarm.synchronise)."""
        self.eta_last_sync[:] = weight_in

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

    # ---- window domain decomposition ('dd', mc_moves.F90:660-709): every rank samples its own window of the
    # overlap parameter; the windows are stitched on rank 0 and broadcast.  Collectives instead of the reference's
    # serial Recv loop: one gather to rank 0 (gather_object of 808-byte arrays is latency, not bandwidth), the
    # stitching in rank order exactly as the reference does it, one broadcast.
    def _gather_to_root(self, arr):
        """The windows' arrays in rank order.  A farm passes a 2-D array, one row per walker of this process: walker k
        of process p is window p * nrows + k, so the rows of all processes are laid end to end."""
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if not dist.is_initialized():
            return [np.array(r) for r in np.atleast_2d(arr)]
        t = torch.from_numpy(arr).to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world_size)]
        dist.all_gather(out, t, group=self.group)
        return [np.array(r) for o in out for r in np.atleast_2d(o.cpu().numpy())]

    def _bcast(self, arr):
        if not dist.is_initialized():
            return arr
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.device)
        dist.broadcast(t, src=0, group=self.group)
        return t.cpu().numpy()

    def join_eta(self, weight, overlap):
        """comms_join_eta (comms_mpi.f90:381-459): rank r's window supplies the bins above r*bins_per_window,
        shifted so that the mean over the 2*overlap+1 bins around the seam agrees; the middle bin is set to 0."""
        parts = self._gather_to_root(weight)
        length, size = len(parts[0]), len(parts)
        bpw = length // size                                           # :399
        joined = parts[0].copy()                                       # :401
        for irank in range(1, size):
            recv = parts[irank]
            end = irank * bpw                                          # my_end_bin (1-based), :419
            seam = slice(end - overlap - 1, end + overlap)             # bins end-overlap .. end+overlap
            shift = joined[seam].sum() / (2 * overlap + 1) - recv[seam].sum() / (2 * overlap + 1)   # :421-429
            joined[end:] = recv[end:] + shift                          # bins end+1 .. length, :431-433
        joined = joined - joined[length // 2]                          # :440-443
        return self._bcast(joined)                                     # :446

    def join_uhist(self, uhist, overlap):
        """comms_join_uhist (comms_mpi.f90:299-379): the same stitching for the unbiased histogram, in log space."""
        parts = self._gather_to_root(uhist)
        length, size = len(parts[0]), len(parts)
        bpw = length // size
        joined = parts[0].copy()
        for irank in range(1, size):
            recv = parts[irank]
            end = irank * bpw
            seam = slice(end - overlap - 1, end + overlap)
            with np.errstate(divide="ignore", invalid="ignore"):
                shift = np.log(joined[seam]).sum() / (2 * overlap + 1) - np.log(recv[seam]).sum() / (2 * overlap + 1)   # :334-346
            if np.isnan(shift):                                        # :347
                shift = 0.0
            joined[end:] = recv[end:] * np.exp(shift)                  # :349-351
        return self._bcast(joined)

    def get_max(self, value):
        """comms_get_max (comms_mpi.f90:279-297)."""
        if not dist.is_initialized():
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def bcast_flag(self, flag):
        """comms_bcastlog (comms_mpi.f90): rank 0's logical for everybody."""
        if not dist.is_initialized():
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
        dist.broadcast(t, src=0, group=self.group)
        return bool(int(t.item()))

    def bcast_int(self, value, root=0):
        """comms_bcastint (comms_mpi.f90): root's integer for everybody (e.g. mc_cycle_num after a restart, mc_moves.F90:441)."""
        if not dist.is_initialized():
            return int(value)
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device)
        dist.broadcast(t, src=root, group=self.group)
        return int(t.item())

    def barrier(self):                                # comms_barrier, comms_mpi.f90:601-618
        if dist.is_initialized():
            dist.barrier(group=self.group)

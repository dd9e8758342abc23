"""Multi-GPU driver of the CE loop: one process per GPU, points sharded by source node, coordinates
replicated and all-gathered over RCCL / xGMI.  Two ways to run the exchange:

  * LibraryComm (default of bench.py): the library's own RCCL communicator (include/annembed_hip.h, ae_comm_*) --
    ae_entropy_optim_gradient_iteration all-gathers the owned rows itself, in place, on its stream, `exchanges` times per
    batch; a host in any language can do the same, torch is only used here to pass the 128-byte id to the other ranks;
  * ShardedCE: the exchange through torch.distributed (gloo in the CPU tests, where the oracle is the compute backend).

No reference counterpart (the reference is single-process, shared-memory rayon).  Semantics:
  * rank r owns source nodes [lo_r, hi_r): it draws positive edges only from its own rows
    (sum_j p_ij = 1 for every node, so equal node counts carry equal edge mass) and runs
    nb_sampling_by_edge * nnz_r samples per batch;
  * negatives are drawn over all N nodes from the local replica of Y;
  * updates of a remote y_j are applied to the local replica only and overwritten by the owner's row
    at the per-batch all-gather (hard part 4).
The compute backend is an object with `gradient_iteration(nb_sample, grad_step, it)`; on the GPU it
is annembed_amd.EntropyOptim (HIP), in the CPU tests it is the oracle (gloo).
"""
import numpy as np


def shard_range(n, world, rank):
    """contiguous node range of `rank`: equal sizes, the remainder spread over the first ranks"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def sample_offset(node_lo):
    """first sample id of the shard's RNG streams (ae_entropy_optim_create: node_lo << 24)"""
    return int(node_lo) << 24


def shard_sizes(n, world):
    return [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]


class ShardedCE:
    """Per-batch protocol: local gradient iteration, then all-gather of the owned rows."""

    def __init__(self, backend, y_tensor, n, dim, rank, world, group=None, pre_gather=None, ranges=None):
        """ranges: [(lo, hi)] per rank (a locality partition's, KGraph.partition); None: equal shares"""
        import torch.distributed as dist
        self.dist = dist
        self.backend, self.y, self.n, self.dim = backend, y_tensor, n, dim
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = ranges[rank] if ranges is not None else shard_range(n, world, rank)
        self.pre_gather = pre_gather
        sizes = [hi - lo for lo, hi in ranges] if ranges is not None else shard_sizes(n, world)
        self.equal = len(set(sizes)) == 1
        self.sizes = sizes

    def step(self, nb_sample, grad_step, it):
        self.backend.gradient_iteration(nb_sample, grad_step, it)
        self.all_gather()

    def all_gather(self):
        if self.world == 1:
            return
        if self.pre_gather is not None:
            self.pre_gather()
        if isinstance(self.backend, HipBackend):
            # the batch's kernels run on the library's own (non-blocking) stream and self.y IS the library's coordinate
            # array: the collective must be ordered after them and before the next batch's -- run it on that stream
            import torch
            with torch.cuda.stream(library_stream()):
                self._gather()
        else:
            self._gather()

    def _gather(self):
        if self.equal:
            # in place: the send buffer is this rank's slot of the receive buffer (what NCCL / RCCL define as in-place
            # all-gather) -- no clone, no copy back; gloo gets a copy (it does not accept aliased buffers)
            own = self.y[self.lo:self.hi]
            if not self.y.is_cuda:
                own = own.clone()
            self.dist.all_gather_into_tensor(self.y, own, group=self.group)
        else:
            import torch
            own = self.y[self.lo:self.hi].clone()
            parts = [torch.empty((s, self.dim), dtype=self.y.dtype, device=self.y.device) for s in self.sizes]
            self.dist.all_gather(parts, own, group=self.group)
            off = 0
            for p in parts:
                self.y[off:off + p.shape[0]].copy_(p)
                off += p.shape[0]

    def all_reduce_sum(self, value):
        """sum of per-shard partial CE values (f64)"""
        if self.world == 1:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.y.device)
        self.dist.all_reduce(t, group=self.group)
        return float(t.item())


def device_tensor(entropy_optim):
    """torch view (no copy) of the library's n x dim device coordinate array"""
    import torch
    ptr, n, d = entropy_optim.device_coords()

    class _Arr:
        __cuda_array_interface__ = {"shape": (n, d), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Arr(), device="cuda")


class HipBackend:
    def __init__(self, entropy_optim):
        self.eo = entropy_optim

    def gradient_iteration(self, nb_sample, grad_step, it):
        self.eo.gradient_iteration_threaded(nb_sample, grad_step, it)


def emulate_sharded_batch(make_backend, y, n, world, nb_samples, grad_step, it):
    """Single-process emulation of one sharded batch (test helper): every rank starts from the same
    replica `y`, runs its shard, and the owners' rows are merged."""
    outs = []
    for r in range(world):
        lo, hi = shard_range(n, world, r)
        yr = np.array(y, copy=True)
        be = make_backend(r, lo, hi, yr)
        be.gradient_iteration(nb_samples[r], grad_step, it)
        outs.append((lo, hi, be.current()))
    merged = np.array(y, copy=True)
    for lo, hi, yr in outs:
        merged[lo:hi] = yr[lo:hi]
    return merged


def library_stream():
    """the library's HIP stream as a torch stream.  Running the per-batch collective under
    `with torch.cuda.stream(library_stream()):` orders it after the batch's kernels and before the next batch's by
    stream events (torch's NCCL/RCCL work waits on, and is waited by, the current stream): no host synchronisation
    in the CE loop (bench.py)."""
    import ctypes
    import torch
    from . import _lib as L
    sp = ctypes.c_void_p()
    L.check(L.load().ae_get_stream(ctypes.byref(sp)))
    return torch.cuda.ExternalStream(sp.value)


class LibraryComm:
    """The library's RCCL communicator (ae_comm_*, include/annembed_hip.h).  The 128-byte id travels from rank 0 to the other
    ranks through an already initialised torch.distributed group (any backend); nothing else of torch is involved."""

    def __init__(self, rank, world):
        import ctypes
        from . import _lib as L
        self._L, self.rank, self.world = L, rank, world
        ident = (ctypes.c_uint8 * 128)()
        failure = None
        if rank == 0:
            try:
                L.check(L.load().ae_comm_unique_id(ident))
            except Exception as e:  # RCCL not loadable on rank 0: the other ranks must not be left waiting in the broadcast
                failure = e
        if world > 1:
            import torch.distributed as dist
            box = [b"" if failure is not None else bytes(ident)]  # an empty id is the failure sentinel: every rank raises together
            dist.broadcast_object_list(box, src=0)
            if len(box[0]) != 128:
                raise RuntimeError("LibraryComm: rank 0 could not create an RCCL unique id" + (": %s" % failure if failure is not None else ""))
            ident = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
        elif failure is not None:
            raise failure
        h = ctypes.c_void_p()
        L.check(L.load().ae_comm_init(rank, world, ident, ctypes.byref(h)))
        self._h = h

    def attach(self, entropy_optim, exchanges_per_batch=4):
        self._L.check(self._L.load().ae_entropy_optim_set_comm(entropy_optim._h, self._h, exchanges_per_batch))

    def attach_none(self, entropy_optim):
        """detach: the handle stops exchanging"""
        self._L.check(self._L.load().ae_entropy_optim_set_comm(entropy_optim._h, None, 0))

    def all_reduce_sum(self, value):
        import ctypes
        v = ctypes.c_double(value)
        self._L.check(self._L.load().ae_comm_all_reduce_sum(self._h, ctypes.byref(v)))
        return v.value

    def close(self):
        if self._h:
            self._L.check(self._L.load().ae_comm_destroy(self._h))
            self._h = None


class HostMemComm(LibraryComm):
    """The library's communicator over a POSIX shared-memory segment (ae_comm_init_hostmem): ranks of one machine, several may
    share a GPU.  Needs no torch and no RCCL; validation of the multi-process path, and the fallback where RCCL cannot load."""

    def __init__(self, rank, world, name, max_bytes):
        import ctypes
        from . import _lib as L
        self._L, self.rank, self.world = L, rank, world
        h = ctypes.c_void_p()
        L.check(L.load().ae_comm_init_hostmem(rank, world, name.encode(), int(max_bytes), ctypes.byref(h)))
        self._h = h

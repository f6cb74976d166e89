"""Communicators with the (small) mpi4py surface PAUXY's hot path uses.

The reference talks MPI (mpi4py) in exactly three places on this path:
population control (walkers/handler.py:232,291,313,322), the estimator reduction
(estimators/mixed.py:261,273) and set-up broadcasts.  ``TorchComm`` provides the
same method names over ``torch.distributed`` -- backend "nccl" (= RCCL over
xGMI) with one process per GPU on the MI355X node, "gloo" on CPU for tests -- so
``Walkers.pop_control(comm)`` / ``Mixed.print_step(comm, ...)`` read like the
reference.  ``FakeComm`` is the size-1 communicator (reference: qmc/comm.py:1-32,
completed with the methods the driver actually calls).

All messages here are tiny (<= 16 KiB all-gather of weights, 160 B estimator
sum) or point-to-point walker copies; there is no bandwidth-bound collective.
"""
import numpy


class FakeComm(object):
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def barrier(self):
        pass

    Barrier = barrier

    def bcast(self, obj, root=0):
        return obj

    def Bcast(self, buf, root=0):
        return buf

    def Allgather(self, send, recv):
        recv[...] = numpy.asarray(send).reshape(recv.shape)

    def Reduce(self, send, recv, op=None, root=0):
        recv[...] = send

    def Allreduce(self, send, recv, op=None):
        recv[...] = send

    # walker transport (device or host tensors); never called at size 1
    def send_tensor(self, t, dest, tag=0):
        raise RuntimeError("send on a size-1 communicator")

    def recv_tensor(self, t, source, tag=0):
        raise RuntimeError("recv on a size-1 communicator")

    def exchange_tensors(self, send, recv):
        if send or recv:
            raise RuntimeError("exchange on a size-1 communicator")


class TorchComm(object):
    """mpi4py-like facade over an initialised torch.distributed process group."""
    reduce_is_allreduce = True      # Reduce() leaves the sum on every rank (see Mixed.print_step)

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch = torch
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        backend = dist.get_backend(group)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
        self.device = device

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def barrier(self):
        self._dist.barrier(group=self.group)

    Barrier = barrier

    def bcast(self, obj, root=0):
        # numeric payloads (the energy shift pair of mixed.py:273, a float) travel as a tensor; anything else
        # falls back to the pickling broadcast.  Every rank passes an object of the same type and shape.
        if isinstance(obj, (float, numpy.floating)):
            return float(self.Bcast(numpy.array([obj], dtype=numpy.float64), root)[0])
        if isinstance(obj, numpy.ndarray) and obj.dtype.kind in 'fc' and obj.size > 0:
            return self.Bcast(numpy.array(obj, dtype=numpy.complex128 if obj.dtype.kind == 'c' else numpy.float64),
                              root)
        box = [obj]
        self._dist.broadcast_object_list(box, src=root, group=self.group, device=self.device)
        return box[0]

    def _to_dev(self, a):
        t = self._torch.from_numpy(numpy.ascontiguousarray(a))
        if self._torch.is_complex(t):
            t = self._torch.view_as_real(t)
        return t.to(self.device)

    def Bcast(self, buf, root=0):
        t = self._to_dev(buf)
        self._dist.broadcast(t, src=root, group=self.group)
        out = t.cpu().numpy()
        if numpy.iscomplexobj(buf):
            out = out.view(numpy.complex128).reshape(buf.shape)
        buf[...] = out.reshape(buf.shape)
        return buf

    def Allgather(self, send, recv):
        s = self._to_dev(numpy.asarray(send, dtype=recv.dtype))
        parts = [self._torch.empty_like(s) for _ in range(self.size)]
        self._dist.all_gather(parts, s, group=self.group)
        o = self._torch.stack(parts).cpu().numpy()
        if numpy.iscomplexobj(recv):
            o = o.view(numpy.complex128)
        recv[...] = o.reshape(recv.shape)

    def Allreduce(self, send, recv, op=None):
        s = self._to_dev(numpy.asarray(send, dtype=recv.dtype))
        self._dist.all_reduce(s, op=self._dist.ReduceOp.SUM, group=self.group)
        o = s.cpu().numpy()
        if numpy.iscomplexobj(recv):
            o = o.view(numpy.complex128)
        recv[...] = o.reshape(recv.shape)

    def Reduce(self, send, recv, op=None, root=0):
        # every rank ends with the sum; only `root` is documented to hold it (mixed.py:261)
        self.Allreduce(send, recv)

    def exchange_tensors(self, send, recv):
        """Post every send ({peer: tensor}) and receive of this rank as one batch and wait for all of them."""
        ops = [self._dist.P2POp(self._dist.isend, t, peer, self.group) for peer, t in sorted(send.items())]
        ops += [self._dist.P2POp(self._dist.irecv, t, peer, self.group) for peer, t in sorted(recv.items())]
        if not ops:
            return
        for req in self._dist.batch_isend_irecv(ops):
            req.wait()

    def send_tensor(self, t, dest, tag=0):
        self._dist.send(t, dst=dest, group=self.group, tag=tag)

    def recv_tensor(self, t, source, tag=0):
        self._dist.recv(t, src=source, group=self.group, tag=tag)


class ReducedComm(object):
    """Stands in for the communicator in ``Mixed.print_step`` when the sums it is handed are already global
    (``afq_estimates_allreduce`` on the device accumulators): Reduce copies, every rank holds the result."""
    reduce_is_allreduce = True
    already_reduced = True

    def __init__(self, comm):
        self.rank, self.size = comm.rank, comm.size

    def Reduce(self, send, recv, op=None, root=0):
        recv[...] = send

    Allreduce = Reduce

    def bcast(self, obj, root=0):
        return obj


class DeviceComm(object):
    """Reductions of host arrays over the library-owned RCCL communicator of an AfqDevice
    (``afq_estimates_allreduce`` with a host buffer): what estimators other than ``mixed`` use when the
    population control already runs on that communicator, so that no second communicator is active
    inside the step loop."""
    reduce_is_allreduce = True

    def __init__(self, dev, comm):
        self.dev = dev
        self.rank, self.size = comm.rank, comm.size

    def Allreduce(self, send, recv, op=None):
        out = self.dev.estimates_allreduce(numpy.asarray(send, dtype=numpy.complex128))
        recv[...] = out.reshape(recv.shape) if numpy.iscomplexobj(recv) else out.real.reshape(recv.shape)

    def Reduce(self, send, recv, op=None, root=0):
        self.Allreduce(send, recv)

    def bcast(self, obj, root=0):
        return obj

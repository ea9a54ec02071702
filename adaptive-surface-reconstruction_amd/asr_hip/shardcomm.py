"""Communicators for ImplicitPipeline.forward_sharded (asr_shard_comm of include/asr_hip.h): the two collective
primitives the sharded forward of the library needs, on device buffers.

RcclComm        RCCL over xGMI, one process per GPU: the library creates its own communicator (ncclCommInitRank inside
                libasr_hip.so, librccl.so loaded at run time); torch.distributed only carries the 128-byte unique id
                from rank 0 to the others.  The halo exchanges are ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd
                on the library's stream -- no Python between the convolutions.
HostStagedComm  the same interface over any torch.distributed group with the buffers staged through the host (gloo):
                for tests that run several ranks on ONE GPU, where RCCL refuses to build a communicator.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib


class RcclComm:
    def __init__(self, ctx, group=None):
        """ctx: the asr_hip Context of this rank's pipeline (its device is the communicator's)"""
        self.ctx = ctx
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # RCCL prints a version banner to STDOUT when it initialises: keep the caller's stdout clean (bench.py prints ONE
        # json line there) by pointing fd 1 at stderr for the duration of the two calls
        import os
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            uid = ctypes.create_string_buffer(128)
            if self.rank == 0:
                ctx.call("asr_hip_shard_comm_rccl_unique_id", uid)
            if self.world > 1:
                box = [bytes(uid.raw)]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                uid = ctypes.create_string_buffer(box[0], 128)
            self._h = ctypes.POINTER(_lib.ShardComm)()
            ctx.call("asr_hip_shard_comm_rccl_create", uid, ctypes.c_int(self.rank), ctypes.c_int(self.world),
                     ctypes.byref(self._h))
        finally:
            try:
                ctypes.CDLL(None).fflush(None)  # the banner sits in C stdio's buffer: flush it while fd 1 is still stderr
            except Exception:
                pass
            os.dup2(saved, 1)
            os.close(saved)

    def handle(self):
        return self._h

    def close(self):
        if self._h:
            f = self.ctx.lib.asr_hip_shard_comm_rccl_destroy
            f.restype = None
            f(self._h)
            self._h = ctypes.POINTER(_lib.ShardComm)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
    return _hip


class HostStagedComm:
    """exchange / allreduce through host buffers and a torch.distributed group (any backend that moves CPU tensors)"""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchanges = 0
        self._ex = _lib.SHARD_EXCHANGE_FN(self._exchange)
        self._ar = _lib.SHARD_ALLREDUCE_FN(self._allreduce)
        self._c = _lib.ShardComm(None, self.rank, self.world, self._ex, self._ar, _lib.SHARD_EXCHANGE_MAX_FN())

    def handle(self):
        return ctypes.byref(self._c)

    @staticmethod
    def _sync(stream):
        rc = _hiprt().hipStreamSynchronize(ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("hipStreamSynchronize failed (%d)" % rc)

    @staticmethod
    def _d2h(ptr, nbytes):
        t = torch.empty(nbytes, dtype=torch.uint8)
        rc = _hiprt().hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2)
        if rc != 0:
            raise RuntimeError("hipMemcpy D2H failed (%d)" % rc)
        return t

    @staticmethod
    def _h2d(ptr, t):
        rc = _hiprt().hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(t.numel()), 1)
        if rc != 0:
            raise RuntimeError("hipMemcpy H2D failed (%d)" % rc)

    def _exchange(self, user, nsend, send_peer, send_buf, send_bytes, nrecv, recv_peer, recv_buf, recv_bytes, stream):
        try:
            self._sync(stream)
            ops, landing = [], []
            for i in range(nsend):
                ops.append(dist.P2POp(dist.isend, self._d2h(send_buf[i], send_bytes[i]), send_peer[i], self.group))
            for i in range(nrecv):
                t = torch.empty(recv_bytes[i], dtype=torch.uint8)
                ops.append(dist.P2POp(dist.irecv, t, recv_peer[i], self.group))
                landing.append((recv_buf[i], t))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for p, t in landing:
                self._h2d(p, t)
            self.exchanges += 1
            return 0
        except Exception as e:  # a ctypes callback must not raise
            print("HostStagedComm.exchange failed: %r" % (e,))
            return 1

    def _allreduce(self, user, buf, n, stream):
        try:
            self._sync(stream)
            t = self._d2h(buf, 4 * n).view(torch.int32)  # non-negative f32 bit patterns order like integers
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            self._h2d(buf, t.view(torch.uint8))
            return 0
        except Exception as e:
            print("HostStagedComm.allreduce failed: %r" % (e,))
            return 1

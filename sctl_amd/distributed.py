"""One process per GPU: ParticleFMM::EvalDirect's rank-parallel form, re-designed for an 8-GPU MI355X node.

Reference (relative to /root/reference): include/sctl/fmm-wrapper.txx:490-562 partitions the targets evenly over the
MPI ranks (:504-512), rotates the source blocks around a ring (:537-558) and repartitions the result (:560-561).
Here: rank g of G owns the contiguous target slab [Nt*g/G, Nt*(g+1)/G) (the formula of :507), every
rank holds ALL sources (O(N) bytes against O(N^2/G) work, so replication beats a ring on one node), evaluates its slab
with the HIP kernels, and ONE all-gather (RCCL over xGMI when the backend is "nccl") assembles the full potential on
every rank.  No other collective touches the data path.

`local_eval` is injectable only so that the slab arithmetic and the gather can be exercised under the gloo backend on
CPU-only machines by tests/ (which pass the CPU oracle); the default — and the only thing product code uses — is the
HIP path, which raises when libsctl_amd.so or a GPU is missing.
"""
import numpy as np

from . import api


def slab_bounds(Nt, rank, world):
    """Target slab of `rank`: [Nt*rank/world, Nt*(rank+1)/world)  (fmm-wrapper.txx:507)."""
    return (Nt * rank) // world, (Nt * (rank + 1)) // world


class ShardedDirectSum:
    """Direct summation of one kernel with targets block-partitioned over the ranks of a torch.distributed group."""

    def __init__(self, kernel, group=None, local_eval=None, ctx=None, digits=-1):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.kernel = kernel
        self.info = api.kernel_info(kernel)
        self.ctx = ctx
        self.digits = digits
        self.local_eval = local_eval or self._hip_eval

    def _hip_eval(self, r_trg_slab, r_src, n_src, v_src, v_out):
        return api.eval_device(self.info["id"], r_trg_slab, r_src, n_src, v_src, v_trg=v_out, digits=self.digits, ctx=self.ctx)

    def eval_slab(self, r_trg, r_src, n_src, v_src, out_slab=None):
        """This rank's slab of the potential (overwritten, like EvalDirect, fmm-wrapper.txx:501-502)."""
        Nt = r_trg.numel() // 3
        t0, t1 = slab_bounds(Nt, self.rank, self.world)
        k1 = self.info["k1"]
        if out_slab is None or out_slab.numel() != (t1 - t0) * k1:
            out_slab = r_trg.new_zeros((t1 - t0) * k1)
        else:
            out_slab.zero_()
        self.local_eval(r_trg[t0 * 3:t1 * 3], r_src, n_src, v_src, out_slab)
        return out_slab

    def eval(self, r_trg, r_src, n_src, v_src, out=None, out_slab=None):
        """Full potential (Nt*TrgDim) on every rank: local slab + one all-gather."""
        import torch
        Nt = r_trg.numel() // 3
        k1 = self.info["k1"]
        slab = self.eval_slab(r_trg, r_src, n_src, v_src, out_slab)
        if out is None or out.numel() != Nt * k1:
            out = r_trg.new_empty(Nt * k1)
        if self.world == 1:
            out.copy_(slab)
            return out
        if Nt % self.world == 0:
            self.dist.all_gather_into_tensor(out, slab, group=self.group)      # equal slabs: one fused collective
        else:
            sizes = [(slab_bounds(Nt, g, self.world)[1] - slab_bounds(Nt, g, self.world)[0]) * k1 for g in range(self.world)]
            pad = max(sizes)
            send = slab if slab.numel() == pad else torch.cat([slab, slab.new_zeros(pad - slab.numel())])
            recv = r_trg.new_empty(pad * self.world)
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
            off = 0
            for g in range(self.world):
                out[off:off + sizes[g]] = recv[g * pad:g * pad + sizes[g]]
                off += sizes[g]
        return out

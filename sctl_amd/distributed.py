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


def morton_order(r_trg):
    """Permutation that sorts 3-D points (AoS tensor of 3*N values) along a Morton curve of their bounding box: 21 bits per
    dimension, stable, a pure function of the coordinates — every rank of a job that holds the same targets gets the same
    permutation without communicating."""
    import torch
    x = r_trg.view(-1, 3).to(torch.float64)
    if x.shape[0] == 0:
        return torch.zeros(0, dtype=torch.int64, device=r_trg.device)
    lo = x.min(dim=0).values
    span = (x.max(dim=0).values - lo).max().clamp_min(1e-300)
    q = ((x - lo) * ((1 << 21) / span)).to(torch.int64).clamp_(0, (1 << 21) - 1)

    def spread(v):                       # 21 bits -> every third bit of 63
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v

    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.sort(key, stable=True).indices


class ShardedDirectSum:
    """Direct summation of one kernel with targets block-partitioned over the ranks of a torch.distributed group.

    compact=True (default) cuts the slabs from the MORTON-ordered targets instead of the caller's order: a rank's targets
    then fill 1/G of the domain at the density of the whole set, which is what the tile-centred Laplace path needs (at
    2^20 targets over 8 GPUs: 59.4 ms per rank against 64.7 ms for a by-index slab, tools/slab_locality.py); the gathered
    potential is put back into the caller's order, so results do not depend on the switch.  The permutation is cached and
    recomputed only when the target tensor changes (an iterative solver changes the density, not the targets)."""

    def __init__(self, kernel, group=None, local_eval=None, ctx=None, digits=-1, compact=True):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.kernel = kernel
        self.info = api.kernel_info(kernel)
        self.ctx = ctx
        self.digits = digits
        self.compact = compact
        self._hip = local_eval is None
        self.local_eval = local_eval or self._hip_eval
        self._trg_ref, self._trg_version, self._perm, self._slab_xt = None, None, None, None

    def _hip_eval(self, r_trg_slab, r_src, n_src, v_src, v_out, nt_whole=None):
        return api.eval_device(self.info["id"], r_trg_slab, r_src, n_src, v_src, v_trg=v_out, digits=self.digits, ctx=self.ctx, nt_whole=nt_whole)

    def _local_targets(self, r_trg):
        """This rank's target slab (coordinates), by index or — compact — from the Morton order (cached)."""
        Nt = r_trg.numel() // 3
        t0, t1 = slab_bounds(Nt, self.rank, self.world)
        if not (self.compact and self.world > 1):
            self._perm = None
            return r_trg[t0 * 3:t1 * 3]
        # The cache is tied to the tensor OBJECT (kept alive here, so its storage cannot be handed to another tensor) and to its
        # version counter (in-place writes): a new tensor of the same size at a recycled address never hits it.
        if r_trg is not self._trg_ref or r_trg._version != self._trg_version:
            self.set_targets(r_trg)
        return self._slab_xt

    def set_targets(self, r_trg):
        """(Re)compute the Morton order and this rank's compact slab for a target set — ParticleFMM::SetTrgCoord's role
        (fmm-wrapper.txx:462-470).  eval()/eval_slab() call it themselves when they see another tensor or a modified one."""
        Nt = r_trg.numel() // 3
        t0, t1 = slab_bounds(Nt, self.rank, self.world)
        if not (self.compact and self.world > 1):
            self.invalidate()
            return
        self._perm = morton_order(r_trg)
        self._slab_xt = r_trg.view(-1, 3)[self._perm[t0:t1]].contiguous().view(-1)
        self._trg_ref, self._trg_version = r_trg, r_trg._version

    def invalidate(self):
        """Forget the cached target order (and the reference to the caller's tensor)."""
        self._trg_ref, self._trg_version, self._perm, self._slab_xt = None, None, None, None

    def eval_slab(self, r_trg, r_src, n_src, v_src, out_slab=None):
        """This rank's slab of the potential (overwritten, like EvalDirect, fmm-wrapper.txx:501-502); with compact=True the
        slab is rows perm[t0:t1] of the caller's targets (self.slab_indices(r_trg))."""
        Nt = r_trg.numel() // 3
        t0, t1 = slab_bounds(Nt, self.rank, self.world)
        k1 = self.info["k1"]
        if out_slab is None or out_slab.numel() != (t1 - t0) * k1:
            out_slab = r_trg.new_zeros((t1 - t0) * k1)
        else:
            out_slab.zero_()
        xt = self._local_targets(r_trg)
        if self._perm is not None and self._hip:
            self.local_eval(xt, r_src, n_src, v_src, out_slab, nt_whole=Nt)
        else:
            self.local_eval(xt, r_src, n_src, v_src, out_slab)
        return out_slab

    def slab_indices(self, r_trg):
        """Indices (into the caller's targets) of this rank's slab, in slab order."""
        import torch
        Nt = r_trg.numel() // 3
        t0, t1 = slab_bounds(Nt, self.rank, self.world)
        self._local_targets(r_trg)
        return self._perm[t0:t1] if self._perm is not None else torch.arange(t0, t1, device=r_trg.device)

    def eval(self, r_trg, r_src, n_src, v_src, out=None, out_slab=None):
        """Full potential (Nt*TrgDim, caller's target order) on every rank: local slab + one all-gather."""
        return self.gather(r_trg, self.eval_slab(r_trg, r_src, n_src, v_src, out_slab), out)

    def gather(self, r_trg, slab, out=None):
        """All-gather the ranks' potential slabs (the one collective of the data path) into the caller's target order."""
        import torch
        Nt = r_trg.numel() // 3
        k1 = self.info["k1"]
        if out is None or out.numel() != Nt * k1:
            out = r_trg.new_empty(Nt * k1)
        if self.world == 1:
            out.copy_(slab)
            return out
        self._local_targets(r_trg)
        if self._perm is None:
            gathered = out
        else:                                           # staging buffer for the Morton-ordered gather, kept between calls
            if getattr(self, "_gathered", None) is None or self._gathered.numel() != Nt * k1 or self._gathered.dtype != r_trg.dtype or self._gathered.device != r_trg.device:
                self._gathered = r_trg.new_empty(Nt * k1)
            gathered = self._gathered
        if Nt % self.world == 0:
            self.dist.all_gather_into_tensor(gathered, slab, group=self.group)      # equal slabs: one fused collective
        else:
            sizes = [(slab_bounds(Nt, g, self.world)[1] - slab_bounds(Nt, g, self.world)[0]) * k1 for g in range(self.world)]
            pad = max(sizes)
            send = slab if slab.numel() == pad else torch.cat([slab, slab.new_zeros(pad - slab.numel())])
            recv = r_trg.new_empty(pad * self.world)
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
            off = 0
            for g in range(self.world):
                gathered[off:off + sizes[g]] = recv[g * pad:g * pad + sizes[g]]
                off += sizes[g]
        if self._perm is not None:
            out.view(Nt, k1).index_copy_(0, self._perm, gathered.view(Nt, k1))      # Morton order -> caller's order
        return out


class RingDirectSum:
    """ParticleFMM::EvalDirect for PARTITIONED inputs (every rank owns some targets and some sources), the pattern of the
    reference under MPI: sources rotate around a ring while targets stay (fmm-wrapper.txx:537-558).  Needed once sources
    stop fitting on one GPU or the ranks span nodes; on one node ShardedDirectSum (replicated sources,
    one all-gather) is the cheaper form.

    Step i evaluates the local targets against the source block that started on rank (rank - i) mod G while the transfer of
    the next block is already in flight (send to rank+1, receive from rank-1, double-buffered), so the link time hides behind
    the O(Nt Ns / G^2) compute of a step.  The reference first re-balances every array with PartitionN (:504-529) and undoes
    that for the result (:560); here the caller's distribution is used as it is (blocks may have different sizes)."""

    def __init__(self, kernel, group=None, local_eval=None, ctx=None, digits=-1):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.info = api.kernel_info(kernel)
        self.ctx = ctx
        self.digits = digits
        self.local_eval = local_eval or self._hip_eval

    def _hip_eval(self, r_trg, r_src, n_src, v_src, v_out):
        return api.eval_device(self.info["id"], r_trg, r_src, n_src, v_src, v_trg=v_out, digits=self.digits, ctx=self.ctx)

    def eval(self, r_trg, r_src, n_src, v_src, out=None):
        """Potential at the LOCAL targets from the sources of ALL ranks (overwritten, like EvalDirect)."""
        import torch
        dist, G, me = self.dist, self.world, self.rank
        k0, k1, nd = self.info["k0"], self.info["k1"], self.info["nd"]
        Nt = r_trg.numel() // 3
        if out is None or out.numel() != Nt * k1:
            out = r_trg.new_zeros(Nt * k1)
        else:
            out.zero_()
        ns_local = torch.tensor([r_src.numel() // 3], dtype=torch.int64, device=r_trg.device if r_trg.is_cuda else "cpu")
        counts = [torch.zeros_like(ns_local) for _ in range(G)]
        if G > 1:
            dist.all_gather(counts, ns_local, group=self.group)
        else:
            counts[0] = ns_local
        counts = [int(c.item()) for c in counts]
        # one flat block per source set: [coords | normals | densities]
        def pack(x, n, f):
            return torch.cat([x, n if nd else x.new_empty(0), f])
        cur = pack(r_src, n_src, v_src)
        per = 3 + nd + k0
        nxt_rank, prv_rank = (me + 1) % G, (me - 1) % G
        for step in range(G):
            owner = (me - step) % G
            ns = counts[owner]
            reqs, incoming = [], None
            if step + 1 < G:
                incoming = cur.new_empty(counts[(me - step - 1) % G] * per)
                ops = [dist.P2POp(dist.isend, cur, nxt_rank, group=self.group), dist.P2POp(dist.irecv, incoming, prv_rank, group=self.group)]
                reqs = dist.batch_isend_irecv(ops)
            if ns > 0 and Nt > 0:
                xs, xn, f = cur[:ns * 3], cur[ns * 3:ns * (3 + nd)], cur[ns * (3 + nd):]
                self.local_eval(r_trg, xs, xn, f, out)          # accumulates into out
            for r in reqs:
                r.wait()
            if incoming is not None:
                cur = incoming
        return out

"""Sharding one registration over the GPUs of a node (host-side logic).

north_star: "partition the target cloud across the 8 GPUs of one node with RCCL all-reduce
over xGMI of the normal equations only".  Rank r indexes the target points of one x-slab
plus a max_corr halo, handles the source points whose TRANSFORMED x falls in its slab (so
every source point is handled by exactly one rank and its true neighbour within max_corr is
in that rank's subset: exact, no per-point exchange), and the 32-double statistics block is
summed over ranks once per iteration; every rank then applies the identical solve.

The iteration driver (ShardedIcp) is engine-agnostic: bench.py runs it with GpuShardEngine
(HIP kernels through the C ABI, torch.distributed/RCCL for the all-reduce); the CPU tests run
the SAME driver with an oracle-backed engine over gloo.
"""
import numpy as np

from . import capi


def plan_slabs(target_xyz, world, axis=0):
    """Equal-count slab edges along `axis`: [(lo, hi)] * world with lo[0] = -inf,
    hi[-1] = +inf and hi[r] == lo[r+1] exactly (as float32, the type the device compares)."""
    x = np.asarray(target_xyz)[:, axis]
    x = x[np.isfinite(x)]
    if world == 1 or len(x) == 0:
        return [(-np.inf, np.inf)] * world if world == 1 else \
            [(-np.inf, np.inf)] + [(np.inf, np.inf)] * (world - 1)
    qs = np.quantile(x.astype(np.float64), np.arange(1, world) / world)
    edges = [float(np.float32(q)) for q in qs]
    for i in range(1, len(edges)):  # strictly increasing, still float32-representable
        if edges[i] <= edges[i - 1]:
            edges[i] = float(np.nextafter(np.float32(edges[i - 1]), np.float32(np.inf)))
    lo = [-np.inf] + edges
    hi = edges + [np.inf]
    return list(zip(lo, hi))


def slab_target_mask(target_xyz, lo, hi, halo, axis=0):
    """Target points a rank must index: x in [lo - halo, hi + halo]."""
    x = np.asarray(target_xyz)[:, axis].astype(np.float64)
    return (x >= lo - halo) & (x <= hi + halo)


def slab_source_mask(source_xyz, lo, hi, pad, axis=0):
    """Source points a rank needs: those whose INITIAL x lies within `pad` of its slab.  A
    point can be missed only if the registration moves it by more than `pad` along x into
    the slab; the ownership count carried in the statistics block detects that (every
    iteration), and the driver then redoes the registration with full source clouds."""
    x = np.asarray(source_xyz)[:, axis].astype(np.float64)
    return np.isfinite(x) & (x >= lo - pad) & (x <= hi + pad)


class GpuShardEngine:
    """One rank's share of a sharded registration on one MI355X."""

    def __init__(self, device, ref, target, rank, world, max_corr, use_torch_stream=True,
                 source_pad=None):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.rank, self.world = rank, world
        self.lo, self.hi = plan_slabs(target, world)[rank]
        # source band: default pad = max_corr (a registration that moves points farther than
        # its own correspondence gate is already outside ICP's basin); None/inf = full cloud
        self.source_pad = float(max_corr) if source_pad is None else float(source_pad)
        ref = np.ascontiguousarray(ref, np.float32)
        self.n_source_total = int(np.isfinite(ref).all(1).sum())
        self._ref_full = ref
        self._full_source = world == 1 or not np.isfinite(self.source_pad)
        # a float32-safe halo: max_corr plus a hair for the float rounding of x
        self.halo = float(max_corr) * (1.0 + 1e-6) + 1e-4
        mask = slab_target_mask(target, self.lo, self.hi, self.halo)
        self.n_target_local = int(mask.sum())
        self._upload_source()
        self.d_tgt = torch.from_numpy(np.ascontiguousarray(target[mask], np.float32)).to(self.dev)
        self.ctx = capi.Context(device)
        if use_torch_stream:
            self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.stats = torch.zeros(capi.WM_STATS_LEN, dtype=torch.float64, device=self.dev)
        self.rebuild()

    def _upload_source(self):
        ref = self._ref_full
        if not self._full_source:
            ref = ref[slab_source_mask(ref, self.lo, self.hi, self.source_pad)]
        self.n_source_local = len(ref)
        self.d_ref = self.torch.from_numpy(np.ascontiguousarray(ref)).to(self.dev)

    def use_full_source(self):
        """safe mode after an ownership violation: every rank holds the whole source cloud"""
        self._full_source = True
        self._upload_source()
        self.rebuild()

    def rebuild(self):
        """index build, part of every timed registration"""
        self.ctx.set_source(self.d_ref)
        self.ctx.set_target(self.d_tgt)

    def begin(self, params):
        self.ctx.shard_begin(params, self.lo, self.hi, self.n_source_total)

    def local_stats(self):
        self.ctx.shard_local_stats(self.stats.data_ptr())
        return self.stats

    def apply(self, stats):
        self.ctx.shard_apply(stats.data_ptr())

    def poll(self):
        return self.ctx.shard_poll()


class ShardedIcp:
    """ICP iteration loop with one all-reduce of the statistics block per iteration."""

    def __init__(self, engine, dist=None, device=None, batch=8):
        self.eng, self.dist, self.batch = engine, dist, batch

    def align(self, params=None, _retry=False, **kw):
        p = params or capi.icp_params(**kw)
        forced = p.force_iterations > 0
        max_it = p.force_iterations if forced else p.max_iter
        self.eng.begin(p)
        it = 0
        out = None
        while it < max_it:
            nb = max_it if forced else min(self.batch, max_it - it)
            for _ in range(nb):
                t = self.eng.local_stats()
                if self.dist is not None:
                    self.dist.all_reduce(t)       # SUM over ranks (RCCL on GPU, gloo in tests)
                self.eng.apply(t)
            it += nb
            out = self.eng.poll()
            if out["done"]:
                break
        if out is not None and out.get("owned_violations", 0) > 0 and not _retry:
            # some source point left every rank's band: identical on all ranks (it is computed
            # from the all-reduced block), so all ranks take this branch together
            self.eng.use_full_source()
            out = self.align(params=p, _retry=True)
            out["redone_with_full_source"] = True
        return out


def make_allreduce(dist, device=None):
    """The `reduce` callback of wm_ndt_set_shard over a torch.distributed group: sums the n
    doubles of one NDT derivative pass over the ranks, in place.  `device` = the rank's GPU for
    the nccl (RCCL) backend -- the values make one round trip through a device tensor -- or None
    for a host backend (gloo).  Every rank receives bit-identical sums (ring / tree all-reduce
    forms each element once and distributes it), which is what keeps the ranks' Newton and
    line-search decisions identical."""
    import torch

    def reduce(vals, n, _user):
        try:
            a = np.ctypeslib.as_array(vals, shape=(n,))
            t = torch.from_numpy(a.copy())
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            a[:] = t.cpu().numpy()
            return 0
        except Exception:  # never unwind through the C caller
            import traceback
            traceback.print_exc()
            return 1

    return capi.ALLREDUCE_FN(reduce)


class ThreadGroupReduce:
    """In-process stand-in for the all-reduce: `world` threads (one context each, e.g. several
    contexts on ONE GPU) meet at a barrier, the values are added in rank order, and every thread
    gets the same sums.  Used by the GPU tests; shows what `reduce` has to guarantee."""

    def __init__(self, world):
        import threading
        self.world = world
        self._barrier = threading.Barrier(world)
        self._slots = [None] * world
        self._sum = None

    def callback(self, rank):
        def reduce(vals, n, _user):
            try:
                a = np.ctypeslib.as_array(vals, shape=(n,))
                self._slots[rank] = a.copy()
                if self._barrier.wait() == 0:
                    tot = np.zeros(n)
                    for r in range(self.world):
                        tot += self._slots[r]
                    self._sum = tot
                self._barrier.wait()
                a[:] = self._sum
                self._barrier.wait()  # nobody overwrites _sum before everyone has read it
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        return capi.ALLREDUCE_FN(reduce)

"""Multi-GPU matching: the map descriptor set is sharded over the ranks of a torch.distributed group
(one process per GPU, backend "nccl" = RCCL over xGMI) by contiguous global index ranges, so that
(shard, local index) order equals global index order and the lowest-index tie-break survives the
merge. The reference is single-process (position_finder.py:159-224); its closest
analogue is the per-tile ROI cache (position_finder.py:248-272).

Two exchange patterns, both one round trip of 16 bytes per query and rank:

  match_replicated  every rank holds the same queries (one frame or batch, extracted on every rank): local
                    top-2 over the shard -> all_gather -> merge. This is north_star's pattern: the latency
                    path of frames against a map too large for one GPU (BASELINE configs[4]).
  match_dp          every rank holds its OWN frames' queries: queries are all-gathered as u8 rows
                    (0.5 MB per 4k keypoints), every rank matches all of them against its shard, and an
                    all_to_all returns each frame's per-shard top-2 to the rank that owns the frame. No
                    rank repeats another's extraction; this is the throughput path (BASELINE configs[3]).

Everything is queued in the order of the imatch context's stream: the collectives are issued with that stream
current (torch orders RCCL's own stream against it with events, no host wait), the kernels are libimatch's on
the same stream. match_dp waits on the host ONCE, for the ranks' row counts (the kNN grid is sized on the
host); the row gather is already queued behind that wait. Results are valid in stream order: consume them on
the context's stream (ratio_dev does) or after backend.synchronize().

torch is used for the collectives and as the owner of the exchange buffers; all arithmetic is in
libimatch.so (kNN, merge)."""
import torch
import torch.distributed as dist


def partition(n_total, world):
    """Contiguous shard ranges [(start, end)] with shard g = rows [g*ceil(n/world), ...)."""
    per = (n_total + world - 1) // world
    return [(min(g * per, n_total), min((g + 1) * per, n_total)) for g in range(world)]


def _rccl(group=None):
    return dist.get_backend(group) == "nccl"


class _StreamScope:
    """Run torch work (collectives, fills, copies) in the imatch context's stream order on a GPU; no-op on CPU."""

    def __init__(self, backend, device, group=None):
        self.cm = None
        if device.type == "cuda":
            self.cm = torch.cuda.stream(torch.cuda.ExternalStream(backend.stream, device=device))

    def __enter__(self):
        if self.cm is not None:
            self.cm.__enter__()

    def __exit__(self, *a):
        if self.cm is not None:
            self.cm.__exit__(*a)


# The collectives. RCCL takes device tensors and is ordered against the current stream with events (no host wait). Any
# other backend (gloo: CPU tests, and the two-ranks-on-one-GPU rehearsal of bench.py) is given HOST tensors: device
# tensors are staged through the host around the call -- gloo's own device path was seen to hang now and then here.
def _staged(fn, out, inp, group, **kw):
    if out.device.type != "cuda" or _rccl(group):
        fn(out, inp, group=group, **kw)
        return
    torch.cuda.current_stream(out.device).synchronize()
    o = torch.empty(out.shape, dtype=out.dtype)
    fn(o, inp.cpu(), group=group, **kw)
    out.copy_(o)
    torch.cuda.current_stream(out.device).synchronize()


def all_gather(out, inp, group=None):
    _staged(dist.all_gather_into_tensor, out, inp, group)


def all_to_all(out, inp, group=None):
    _staged(dist.all_to_all_single, out, inp, group)


def all_reduce(t, op, group=None):
    if t.device.type != "cuda" or _rccl(group):
        dist.all_reduce(t, op=op, group=group)
        return t
    h = t.cpu()
    dist.all_reduce(h, op=op, group=group)
    t.copy_(h)
    return t


class ShardedMatcher:
    def __init__(self, backend, shard_features, global_offset, device, group=None, local_only=False, force_exchange=False):
        """backend: an image_matching_amd.Context (or an object with the same knn2_dev / merge_top2_dev /
        features_from_device_u8_seg / synchronize methods); shard_features: this rank's rows of the map.
        local_only: the rank holds the WHOLE map (a map that fits one GPU is replicated, not sharded): no exchange.
        force_exchange: run the collectives even in a group of one rank (tests: the exchange path under RCCL on one GPU)."""
        self.backend = backend
        self.shard = shard_features
        self.offset = int(global_offset)
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() and not local_only else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() and not local_only else 0
        self.force = bool(force_exchange) and dist.is_initialized() and not local_only
        self._bufs = {}
        self.last_counts = None      # row counts of all ranks in the last match_dp (diagnostics)

    def _buf(self, name, shape, dtype, pinned_host=False):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            if pinned_host:
                t = torch.zeros(shape, dtype=dtype, pin_memory=self.device.type == "cuda")
            else:
                t = torch.zeros(shape, dtype=dtype, device=self.device)
                if self.device.type == "cuda":
                    # the fill runs on torch's stream, the buffer's first user on the context's: once per buffer, wait
                    torch.cuda.current_stream(self.device).synchronize()
            self._bufs[key] = t
        return t

    def match_replicated(self, q_features, nq):
        """All ranks pass the same nq queries. Returns the merged int32 [nq,4] =
        (ssd1, idx1, ssd2, idx2) with global indices, identical on every rank (valid in stream order)."""
        W = self.world
        local = self._buf("rep_local", (nq, 4), torch.int32)
        self.backend.knn2_dev(q_features, self.shard, local.data_ptr(), self.offset)
        if W == 1 and not self.force:
            return local
        lists = self._buf("rep_lists", (W, nq, 4), torch.int32)
        out = self._buf("rep_out", (nq, 4), torch.int32)
        with _StreamScope(self.backend, self.device, self.group):
            all_gather(lists.view(W * nq, 4), local, self.group)
        self.backend.merge_top2_dev(lists.data_ptr(), W, nq, out.data_ptr())
        return out

    def match_dp(self, q_features, nq, q_cap):
        """Each rank passes its own nq <= q_cap queries. Returns int32 [nq,4] for this rank's queries,
        matched against the whole (sharded) map (valid in stream order)."""
        if nq > q_cap:
            raise ValueError("nq %d exceeds q_cap %d" % (nq, q_cap))
        W = self.world
        if W == 1 and not self.force:
            # no exchange and no host synchronisation
            out = self._buf("dp_out", (q_cap, 4), torch.int32)
            self.backend.knn2_dev(q_features, self.shard, out.data_ptr(), self.offset)
            return out[:nq]
        qall = self._buf("dp_qall", (W, q_cap, 128), torch.uint8)       # slot r = rank r's rows, filled from its start
        cnt_all = self._buf("dp_cnt_all", (W,), torch.int64)
        cnt_host = self._buf("dp_cnt_host", (W,), torch.int64, pinned_host=True)
        part = self._buf("dp_part", (W * q_cap, 4), torch.int32)
        recv = self._buf("dp_recv", (W * q_cap, 4), torch.int32)
        out = self._buf("dp_out", (q_cap, 4), torch.int32)
        qlocal = qall[self.rank]
        if nq > 0:
            q_features.desc_u8_dev(qlocal.data_ptr())                   # rows beyond nq keep stale bytes nobody reads
        with _StreamScope(self.backend, self.device, self.group):
            cnt_all[self.rank:self.rank + 1].fill_(nq)
            all_gather(cnt_all, cnt_all[self.rank:self.rank + 1].clone(), self.group)
            ev = None
            if self.device.type == "cuda":
                cnt_host.copy_(cnt_all, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:
                cnt_host.copy_(cnt_all)
            # RCCL gathers in place (the rank's rows already sit in its slot of the output); gloo wants a separate input
            all_gather(qall.view(W * q_cap, 128), qlocal if _rccl(self.group) else qlocal.clone(), self.group)
        if ev is not None:
            ev.synchronize()        # the step's one host wait: how many rows each rank sent
        counts = [int(c) for c in cnt_host.tolist()]
        self.last_counts = counts
        # all ranks' real rows back to back (no slot padding goes through the MFMA kernel), matched against this shard
        f_all = self.backend.features_from_device_u8_seg(qall.data_ptr(), counts, q_cap)
        self.backend.knn2_dev(f_all, self.shard, part.data_ptr(), self.offset)
        # back to the owners: rank r's rows go into slot r of an equally split all_to_all (slots of q_cap rows; uneven
        # splits hang gloo on device tensors, and the slot tails are 16 B per unused row)
        slots = self._buf("dp_slots", (W, q_cap, 4), torch.int32)
        with _StreamScope(self.backend, self.device, self.group):
            at = 0
            for r, c in enumerate(counts):
                if c:
                    slots[r, :c].copy_(part[at:at + c], non_blocking=True)
                at += c
            all_to_all(recv, slots.view(W * q_cap, 4), self.group)
        # recv = [W][q_cap][4]: list r = shard r's top-2 for this rank's queries (rows beyond nq are never read back)
        self.backend.merge_top2_dev(recv.data_ptr(), W, q_cap, out.data_ptr())
        f_all.free()                # back to the context's block pool; reuse is ordered by the stream
        return out[:nq]

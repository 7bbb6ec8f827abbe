"""Multi-GPU matching: the map descriptor set is sharded over the ranks of a torch.distributed group
(one process per GPU, backend "nccl" = RCCL over xGMI) by contiguous global index ranges, so that
(shard, local index) order equals global index order and the lowest-index tie-break survives the
merge. The reference is single-process (position_finder.py:159-224); its closest
analogue is the per-tile ROI cache (position_finder.py:248-272).

Two exchange patterns, both one round trip of 16 bytes per query and rank:

  match_replicated  every rank holds the same queries (one frame, extracted on every rank): local
                    top-2 over the shard -> all_gather -> merge. This is the latency path of one
                    frame against a map too large for one GPU.
  match_dp          every rank holds its OWN frame's queries: queries are all-gathered as u8 rows
                    (0.5 MB per 4k keypoints), every rank matches all of them against its shard, and an
                    all_to_all returns each frame's per-shard top-2 to the rank that owns the frame. No
                    rank repeats another's extraction; this is the throughput path bench.py runs at N > 1.

torch is used for the collectives and as the owner of the exchange buffers; all arithmetic is in
libimatch.so (kNN, merge)."""
import torch
import torch.distributed as dist


def partition(n_total, world):
    """Contiguous shard ranges [(start, end)] with shard g = rows [g*ceil(n/world), ...)."""
    per = (n_total + world - 1) // world
    return [(min(g * per, n_total), min((g + 1) * per, n_total)) for g in range(world)]


class _StreamScope:
    """Run torch work (collectives, fills) in the imatch context's stream order on a GPU; no-op on CPU."""

    def __init__(self, backend, device):
        self.cm = None
        if device.type == "cuda":
            self.cm = torch.cuda.stream(torch.cuda.ExternalStream(backend.stream, device=device))

    def __enter__(self):
        if self.cm is not None:
            self.cm.__enter__()

    def __exit__(self, *a):
        if self.cm is not None:
            self.cm.__exit__(*a)


class ShardedMatcher:
    def __init__(self, backend, shard_features, global_offset, device, group=None, local_only=False):
        """backend: an image_matching_amd.Context (or an object with the same knn2_dev / merge_top2_dev /
        features_from_device_u8 / synchronize methods); shard_features: this rank's rows of the map.
        local_only: the rank holds the WHOLE map (a map that fits one GPU is replicated, not sharded): no exchange."""
        self.backend = backend
        self.shard = shard_features
        self.offset = int(global_offset)
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() and not local_only else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() and not local_only else 0
        self._bufs = {}

    def _buf(self, name, shape, dtype):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def match_replicated(self, q_features, nq):
        """All ranks pass the same nq queries. Returns the merged int32 [nq,4] =
        (ssd1, idx1, ssd2, idx2) with global indices, identical on every rank."""
        W = self.world
        local = self._buf("rep_local", (nq, 4), torch.int32)
        self.backend.knn2_dev(q_features, self.shard, local.data_ptr(), self.offset)
        if W == 1:
            self.backend.synchronize()
            return local
        lists = self._buf("rep_lists", (W, nq, 4), torch.int32)
        out = self._buf("rep_out", (nq, 4), torch.int32)
        with _StreamScope(self.backend, self.device):
            dist.all_gather_into_tensor(lists.view(W * nq, 4), local, group=self.group)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        self.backend.merge_top2_dev(lists.data_ptr(), W, nq, out.data_ptr())
        self.backend.synchronize()
        return out

    def match_dp(self, q_features, nq, q_cap):
        """Each rank passes its own nq <= q_cap queries. Returns int32 [nq,4] for this rank's queries,
        matched against the whole (sharded) map."""
        if nq > q_cap:
            raise ValueError("nq %d exceeds q_cap %d" % (nq, q_cap))
        W = self.world
        if W == 1:
            # no exchange and no host synchronisation: the result is valid in the backend's stream order
            # (ratio_dev consumes it there); read it from another stream only after backend.synchronize()
            out = self._buf("dp_out", (q_cap, 4), torch.int32)
            self.backend.knn2_dev(q_features, self.shard, out.data_ptr(), self.offset)
            return out[:nq]
        qlocal = self._buf("dp_qlocal", (q_cap, 128), torch.uint8)
        qall = self._buf("dp_qall", (W, q_cap, 128), torch.uint8)
        part = self._buf("dp_part", (W, q_cap, 4), torch.int32)
        recv = self._buf("dp_recv", (W, q_cap, 4), torch.int32)
        out = self._buf("dp_out", (q_cap, 4), torch.int32)
        with _StreamScope(self.backend, self.device):
            qlocal.zero_()
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        if nq > 0:
            q_features.desc_u8_dev(qlocal.data_ptr())
        self.backend.synchronize()
        with _StreamScope(self.backend, self.device):
            dist.all_gather_into_tensor(qall.view(W * q_cap, 128), qlocal, group=self.group)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        f_all = self.backend.features_from_device_u8(qall.data_ptr(), W * q_cap)
        self.backend.knn2_dev(f_all, self.shard, part.data_ptr(), self.offset)
        self.backend.synchronize()
        with _StreamScope(self.backend, self.device):
            dist.all_to_all_single(recv.view(W * q_cap, 4), part.view(W * q_cap, 4), group=self.group)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        self.backend.merge_top2_dev(recv.data_ptr(), W, q_cap, out.data_ptr())
        self.backend.synchronize()
        f_all.free()
        return out[:nq]

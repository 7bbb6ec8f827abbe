"""The exchange path of image_matching_amd/sharded.py under the real collective backend (RCCL) on the one GPU a test box
has: a process group of ONE rank with the exchange forced on, so that every collective call -- the in-place all-gather of
the query slots, the count gather and its single host wait, the all_to_all of per-shard lists, the all-gather of
match_replicated -- runs through RCCL in the context's stream order and must reproduce the plain single-launch answer.
(Two ranks need two GPUs: that is tests/test_sharded_gloo.py on the CPU and bench.py --gpus N on a node.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import image_matching_amd as im
from image_matching_amd import sharded, synth
ctx = im.Context(0)
t = synth.synth_descriptors(40000, seed=3)
t[30000:30100] = t[5:105]                      # ties
ft = ctx.features(t)
m = sharded.ShardedMatcher(ctx, ft, 0, dev, force_exchange=True)
assert m.world == 1 and m.force
for nq in (3000, 700, 0, 4096):                # shrinking and growing inputs through the same buffers
    q, _, _ = synth.synth_queries(t, max(nq, 1), seed=nq, dup_frac=0.2)
    q = q[:nq]
    fq = ctx.features(q)
    ref = torch.zeros((max(nq, 1), 4), dtype=torch.int32, device=dev)
    ctx.knn2_dev(fq, ft, ref.data_ptr())
    ctx.synchronize()
    a = m.match_dp(fq, nq, 4096)
    ctx.synchronize()
    assert m.last_counts == [nq], m.last_counts
    assert a.shape == (nq, 4) and torch.equal(a, ref[:nq]), "match_dp differs at nq=%d" % nq
    if nq:
        b = m.match_replicated(fq, nq)
        ctx.synchronize()
        assert torch.equal(b, ref[:nq]), "match_replicated differs at nq=%d" % nq
dist.destroy_process_group()
print("EXCHANGE_OK")
"""


def test_exchange_path_under_rccl_one_rank():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "-c", _SCRIPT, ROOT, str(port)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "EXCHANGE_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])

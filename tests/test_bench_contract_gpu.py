"""bench.py's output contract on a small workload (2048 x 2048 map, 8-frame steps), run in-process on the GPU: one JSON
line with the driver's keys, the `roofline` and `cpu_baseline` objects, and a `parity` block that is green."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line(capsys, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3", "--warmup", "1", "--map-side", "2048", "--batch", "8",
                                      "--frames", "16", "--map-rows", "150001"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["unit"] == "frames/s"
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["data"] == "synthetic" and "workload" in out["config"] and "model" not in out["config"]
    assert abs(out["value"] - 8 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    r = out["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 5000.0 and out["dtype"] == "i8"
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    # both workloads of the default N-series are measured in the run: the headline (c1 at N = 1) and the sharded-map one
    se = out["series"]
    assert se["c1"]["value"] == out["value"] and se["c1"]["n_gpus"] == 1
    assert "error" not in se["c3"], se["c3"]
    assert se["c3"]["value"] > 0 and se["c3"]["map_descriptors"] == 150001 and se["c3"]["n_gpus"] == 1
    assert abs(se["c3"]["value"] - 8 * se["c3"]["steps"] / (se["c3"]["ms_per_step"] * se["c3"]["steps"] * 1e-3)) < 1e-6 * se["c3"]["value"]
    c = out["cpu_baseline"]
    assert c["knn_gdist_per_s_per_core"] > 0
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    p = out["parity"]
    assert "error" not in p, p
    assert p["keypoints_within_0.5px"] >= 0.99 and p["knn_idx_and_dist_bit_exact"] and p["ratio_decisions_identical"]
    assert p["homography_corner_px_vs_oracle_max"] is None or p["homography_corner_px_vs_oracle_max"] <= 1.0


def test_bench_launches_its_own_ranks_and_runs_the_sharded_map(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent starts both ranks itself, the headline workload is the
    sharded map (configs[3] at a small --map-rows), and the exchange path gives what a single launch over the whole map
    gives. Rehearsal form for a one-GPU box: both ranks on device 0, gloo in place of RCCL (which refuses two ranks on one
    GPU); the code path is the one torchrun / the 8-GPU node runs."""
    import subprocess
    env = dict(os.environ, IMATCH_BENCH_DEVICE="0", IMATCH_BENCH_BACKEND="gloo", IMATCH_BENCH_VERIFY="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    for workload, scaling in (("c3", "weak"), ("c5", "strong")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--map-side", "2048", "--map-rows", "150001", "--batch", "8", "--frames", "16", "--workload", workload],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["config"]["workload_id"] == workload
        m = out["multi_gpu"]
        assert m["world_size_seen"] == 2 and m["identical_to_whole_map"] is True
        assert len(m["per_rank_knn_frac"]) == 2 and all(0 < f < 1 for f in m["per_rank_knn_frac"])
        assert m["per_rank_shard_rows"] == [75001, 75000] and m["exchange_us_per_step"] > 0
        if workload == "c3":     # the data-parallel c1 point of the same N, measured in the run
            se = out["series"]
            assert se["c3"]["value"] == out["value"] and "error" not in se["c1"], se
            assert se["c1"]["n_gpus"] == 2 and se["c1"]["value"] > 0
        frames = 8 * 2 * (2 if workload == "c3" else 1)
        assert abs(out["value"] - frames / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
        found, total = out["config"]["homographies_found"].split("/")
        assert int(found) >= 0.9 * int(total)

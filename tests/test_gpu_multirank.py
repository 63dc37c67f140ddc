"""The N>1 path of bench.py on real hardware.  On the 1-GPU test box: ranks that share the one GPU
(BENCH_SHARE_GPU=1) with gloo standing in for RCCL; on a box with at least as many GPUs as ranks the same tests run
over `nccl` (= RCCL), one GPU per rank (`_multi_gpu_env`).  Exercises what the world-size-1 run never
reaches: record sharding by rank, the all-gather of the signatures, row-block compare against the
gathered columns, the cross-rank reductions of the timing and of the per-rank diagonal check."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _multi_gpu_env(world):
    """RCCL and one GPU per rank when the box has the GPUs; else every rank on cuda:0 and gloo.  (device_count() does not
    initialise the GPU in this process.)"""
    import torch
    if torch.cuda.device_count() >= world:
        return dict(os.environ)
    return dict(os.environ, BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo")


def test_bench_two_ranks_sharing_the_gpu(pkg):
    env = _multi_gpu_env(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2",
           "--gb", "0.2", "--protein-gb", "0.2", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--compare-n", "601"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["retained_hashes"] > 0
    c = d["compare"]
    assert c["n_signatures"] == 601 and c["self_jaccard_is_1"] is True   # 601: the last row block is short
    assert d["cpu_baseline"] is None                                      # reported at N=1 only
    # the ranks' partial sketches united on the device, timed on its own (never part of `value`)
    u = d["union_across_ranks"]
    assert u["parts"] == 2 and u["hashes"] > d["config"]["retained_hashes"] and u["union_ms"] > 0 and u["verified"] is True
    # the self-check of the exchange: sampled rows recomputed by every rank alone, equal bit for bit (distributed.verify_exchange)
    assert c["exchange_verified"] is True
    for key in ("families", "one_component", "one_family"):
        assert c[key]["exchange_verified"] is True and c[key]["verified_rows_rank0"] >= 16
    # configs[4]'s share through the protein arm, then ONE signature from the ranks' partial sketches (abundances added)
    pr = d["protein"]
    assert pr["value"] > 0 and pr["unit"] == "windows/s" and pr["roofline"]["kernel_ms_avg"] > 0
    assert pr["union_across_ranks"]["verified"] is True and pr["union_across_ranks"]["hashes"] > pr["config"]["retained_hashes"]
    assert "strong" in c["scaling"] and "used" in c["symmetry"]
    assert set(c["families"]["rank0_phase_ms"]) >= {"all_gather_signatures", "dictionary_slice", "all_gather_shares", "compare", "exchange_mirrors"}


def test_bench_gpus_2_with_no_launcher_around_it(pkg):
    """`python bench.py --gpus 2` exactly as typed (no torchrun wrapper): bench.py starts its own two
    ranks as child processes before anything touches the GPU and relays rank 0's single JSON line."""
    env = _multi_gpu_env(2)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--gb", "0.1", "--protein-gb", "0.1", "--steps", "1", "--warmup", "0", "--compare-n", "300"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["compare"]["self_jaccard_is_1"] is True
    assert d["compare"]["one_component"]["self_jaccard_is_1"] is True and d["compare"]["exchange_verified"] is True


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_matrix_over_a_real_process_group(world, pkg):
    """`world` processes sharing the test box's one GPU, a real process group (gloo standing in for RCCL): the all-gather of
    the signatures, the all-gather of the dictionary shares, the all-to-all of the mirrored blocks.  Rank 0 collects the row
    blocks and compares them, bit for bit, with the matrix one rank computes alone (tools/sharded_check.py) on the family,
    one-component and one-family collections; 701 signatures: the last row block is short."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join("tools", "sharded_check.py"), "701",
           "1.0" if world == 2 else "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "sharded check ok" in r.stdout and "DIFFERENT" not in r.stdout
    assert r.stdout.count("equal") == (16 if world == 2 else 15)        # 4 outputs + verify_exchange per collection (+ the union)
    assert r.stdout.count("verify_exchange world %d: equal" % world) == 3
    if world == 2:
        # ... and the sketch side: 1 GB per rank through the protein arm with abundances, the partial sketches united across
        # the ranks on the device (one all-gather of the padded arrays) == the sketch of the 2 GB on one rank
        assert "union of 2 ranks x 1.0 GB (protein, abundances)" in r.stdout and "nothing copied to the host" in r.stdout

"""`bench.py --gpus N` starts N ranks of itself (no GPU needed to see that it does)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SLAM_BENCH_BACKEND")}
    env.update(extra)
    return env


def test_child_command_is_one_rank_per_gpu_on_this_node():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.child_command(4, ["--gpus", "4", "--steps", "5", "--strong"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "5", "--strong"]        # the ranks see the same arguments


def test_gpus_2_starts_two_ranks_and_relays_their_exit_code():
    """Without GPUs each RCCL rank refuses to run (one rank per GPU or nothing); the parent must have
    started both and must pass the failure on instead of printing a one-rank result."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600, cwd="/tmp")
    import torch
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
        return
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank 0: 2 ranks asked for" in r.stderr and "rank 1: 2 ranks asked for" in r.stderr


def test_gpus_must_match_the_world_the_launcher_made():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120, cwd="/tmp")
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""

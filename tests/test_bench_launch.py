"""bench.py's launch contract: `--gpus N` without a launcher starts N ranks itself (the parent never touches a
GPU), a WORLD_SIZE that differs from --gpus fails loudly, and -- on the GPU box -- the self-launched two-rank
rehearsal (gloo, both ranks on the one device) produces a line with n_gpus == 2 whose gathered frames equal
unsharded renders, for the tile-sharded configuration (BASELINE config 3) and the replica one (config 5)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BENCH = ROOT / "bench.py"


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_world_size_mismatch_fails_before_any_gpu_work():
    r = subprocess.run([sys.executable, str(BENCH), "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    # a launcher that started more ranks than --gpus (default 1) is just as wrong
    r = subprocess.run([sys.executable, str(BENCH)], env=_env(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 1" in r.stderr


def test_self_launch_builds_a_torchrun_command_and_parent_stays_gpu_free():
    code = r"""
import sys, types, subprocess
sys.argv = ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"]
sys.path.insert(0, %r)
import bench
seen = {}
def fake_run(cmd, env=None, **kw):
    seen["cmd"], seen["env"] = cmd, env
    return types.SimpleNamespace(returncode=7)
subprocess.run = fake_run
try:
    bench.main()
except SystemExit as e:
    code = e.code
cmd = seen["cmd"]
assert code == 7, code                       # the children's exit code is relayed
assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
i = cmd.index(str(bench.Path(bench.__file__).resolve()))
assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]   # the ranks get the same arguments
assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
assert "torch" not in sys.modules, "the launching parent must not import torch (nor initialise a GPU)"
print("ok")
""" % str(ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def _bench_line(args, timeout=900):
    r = subprocess.run([sys.executable, str(BENCH), *args], env=_env(), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
def test_self_launched_two_rank_rehearsal_tile_sharded():
    out = _bench_line(["--gpus", "2", "--backend", "gloo", "--single-device", "--check", "--steps", "2", "--warmup", "1",
                       "--views-per-step", "2"])
    assert out["n_gpus"] == 2 and out["distributed"]["world_size"] == 2 and out["distributed"]["backend"] == "gloo"
    assert out["distributed"]["launcher"] == "bench.py self-launch" and len(out["distributed"]["devices"]) == 2
    assert out["sharded_frame_equals_unsharded"] is True
    assert out["config"]["parallelism"] == "tile2" and out["value"] > 0
    # weak scaling is the default: the step grows with the ranks (2 frames per rank's worth of tiles -> 4 frames)
    assert out["scaling"] == "weak" and out["config"]["views_per_step"] == 4
    assert out["config"]["gather_root"] == "step % N"  # the default: every rank assembles every N-th step's frames
    # what the first real N-GPU run is to record: every rank's row of the peer-access matrix and the RCCL version
    d = out["distributed"]
    assert len(d["peer_access"]) == 2 and all(len(row) == d["visible_devices"] and row[0] == 1 for row in d["peer_access"])
    assert "rccl_version" in d


@pytest.mark.gpu
def test_self_launched_two_rank_rehearsal_strong_and_multi_launch_steps():
    out = _bench_line(["--gpus", "2", "--backend", "gloo", "--single-device", "--check", "--steps", "2", "--warmup", "1",
                       "--views-per-step", "3", "--scaling", "strong", "--gather-root", "0"])
    assert out["config"]["gather_root"] == "rank 0"
    assert out["scaling"] == "strong" and out["config"]["views_per_step"] == 3 and out["sharded_frame_equals_unsharded"] is True
    # 70 x 2 = 140 frames per step = two launches per rank and step (NRF_MAX_VIEWS = 128), at a smaller resolution
    out = _bench_line(["--gpus", "2", "--backend", "gloo", "--single-device", "--check", "--steps", "1", "--warmup", "1",
                       "--views-per-step", "70", "--width", "320", "--height", "184"])
    assert out["config"]["views_per_step"] == 140 and out["roofline"]["launches_per_step"] == 2
    assert out["sharded_frame_equals_unsharded"] is True


@pytest.mark.gpu
def test_self_launched_two_rank_rehearsal_config5_replicas():
    out = _bench_line(["--gpus", "2", "--config", "5", "--backend", "gloo", "--single-device", "--check", "--steps", "1",
                       "--warmup", "1", "--views-per-step", "8"])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "replica2"
    assert out["config"]["views_per_step"] == 8 and out["config"]["views_per_rank_and_step"] == 4
    assert out["sharded_frame_equals_unsharded"] is True and "800x800" in out["config"]["workload"]
    assert out["scaling"] == "strong"


@pytest.mark.gpu
def test_forced_single_rank_runs_the_real_rccl_exchange():
    """`--force-dist --backend nccl` on the one GPU of the box: a one-rank RCCL communicator
    (init_process_group("nccl", device_id=...)), the shard rendered tile-major and packed, dist.gather of int32 device tensors
    through RCCL, untile_views with shard_count 1, the one-stream render / exchange hand-off with two steps in flight,
    and the check of the gathered + untiled frames against unsharded renders.  What a one-GPU box can execute of the N > 1
    path (the reference: R/src/nerf_render.cu:345-359) -- config 3 (tile-sharded) and config 5 (replicas)."""
    out = _bench_line(["--gpus", "1", "--force-dist", "--backend", "nccl", "--check", "--steps", "3", "--warmup", "1",
                       "--views-per-step", "3"])
    d = out["distributed"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["forced_single_rank"] is True and d["rccl_version"]
    assert out["sharded_frame_equals_unsharded"] is True
    assert out["n_gpus"] == 1 and out["config"]["parallelism"] == "tile1" and out["config"]["gather"] == "rgbd8"
    assert out["config"]["steps_in_flight"] == 2 and out["value"] > 0
    assert "api" not in out and "cpu_baseline" not in out  # the exchange rehearsal carries no extras
    # float planes on the wire instead of packed pixels
    out = _bench_line(["--gpus", "1", "--force-dist", "--backend", "nccl", "--check", "--steps", "2", "--warmup", "1",
                       "--views-per-step", "2", "--gather-format", "f32", "--width", "328", "--height", "200"])
    assert out["distributed"]["backend"] == "nccl" and out["sharded_frame_equals_unsharded"] is True
    # config 5: whole frames per rank, gather of images, no untile
    out = _bench_line(["--gpus", "1", "--force-dist", "--backend", "nccl", "--config", "5", "--check", "--steps", "2",
                       "--warmup", "1", "--views-per-step", "4"])
    assert out["distributed"]["backend"] == "nccl" and out["config"]["parallelism"] == "replica1"
    assert out["sharded_frame_equals_unsharded"] is True


@pytest.mark.parametrize("cfg_args,want", [
    # BASELINE config 3 at its own size: 8 shards of a 1920x1080 frame (tiles_per_shard(1920, 1080, 8) = 4 052 tiles: 8 100 strips, four ranks own one more than the others;
    # weak scaling (1 x 8 frames per step), every rank the sink of exactly one of the 8 steps
    (["--steps", "7", "--warmup", "1", "--views-per-step", "1"],
     dict(parallelism="tile8", views_per_step=8, views_per_rank=8, sinks=[1] * 8, scaling="weak", tps=4052)),
    # config 5 at its own size: 64 requests of 800x800 per step in 8 blocks of 8 whole frames, strong scaling
    (["--config", "5", "--steps", "1", "--warmup", "1"],
     dict(parallelism="replica8", views_per_step=64, views_per_rank=8, sinks=[1, 1, 0, 0, 0, 0, 0, 0], scaling="strong", tps=None)),
    # a sink fixed at rank 0 and a frame whose strips do not divide over the ranks
    (["--steps", "2", "--warmup", "0", "--views-per-step", "2", "--width", "328", "--height", "200", "--gather-root", "0", "--scaling", "strong"],
     dict(parallelism="tile8", views_per_step=2, views_per_rank=2, sinks=[2, 0, 0, 0, 0, 0, 0, 0], scaling="strong", tps=None)),
])
def test_eight_rank_dry_exchange_through_bench_itself(cfg_args, want):
    """the first real 8-GPU run should exercise only RCCL / xGMI for the first time.  A one-GPU box takes at
    most 6 processes on its card, so the 8-rank rehearsal runs WITHOUT the card: `bench.py --gpus 8 --dry-exchange` starts its 8
    ranks itself (torch.distributed.run, 127.0.0.1), runs bench.py's own step plan (step_plan / step_pose_indices / step_root:
    the functions the GPU path calls), the gloo gather to the rotating sink, the untile and the check -- with f(pose, x, y) in
    place of the render (the reference's exchange: R/src/nerf_render.cu:345-359)."""
    out = _bench_line(["--gpus", "8", "--dry-exchange", *cfg_args], timeout=600)
    assert out["dry_exchange"] is True and out["n_gpus"] == 8 and out["distributed"]["world_size"] == 8
    assert out["distributed"]["launcher"] == "bench.py self-launch"
    assert out["sharded_frame_equals_unsharded"] is True
    c = out["config"]
    assert c["parallelism"] == want["parallelism"] and c["views_per_step"] == want["views_per_step"]
    assert c["views_per_rank_and_step"] == want["views_per_rank"] and out["scaling"] == want["scaling"]
    assert out["sink_steps_per_rank"] == want["sinks"]
    if want["tps"]:
        assert c["tiles_per_shard"] == want["tps"]


@pytest.mark.gpu
def test_self_launched_four_rank_rehearsal_configs_3_and_5():
    """Four ranks on the one device (with the test process: five on the card, the box allows six): the rotating sink over more
    than two ranks, tiles_per_shard(.., 4), four replica blocks -- rendered, gathered (gloo), untiled on the GPU and checked
    against unsharded renders.  Eight ranks: test_eight_rank_dry_exchange_through_bench_itself (no card)."""
    out = _bench_line(["--gpus", "4", "--backend", "gloo", "--single-device", "--check", "--steps", "4", "--warmup", "1",
                       "--views-per-step", "1", "--width", "488", "--height", "272"])
    assert out["n_gpus"] == 4 and out["config"]["parallelism"] == "tile4" and out["config"]["views_per_step"] == 4
    assert out["sharded_frame_equals_unsharded"] is True and out["config"]["gather_root"] == "step % N"
    out = _bench_line(["--gpus", "4", "--config", "5", "--backend", "gloo", "--single-device", "--check", "--steps", "2",
                       "--warmup", "1", "--views-per-step", "8", "--width", "200", "--height", "200"])
    assert out["n_gpus"] == 4 and out["config"]["parallelism"] == "replica4" and out["config"]["views_per_rank_and_step"] == 2
    assert out["sharded_frame_equals_unsharded"] is True


def test_cpu_baseline_counts_the_cpus_the_cgroup_grants(tmp_path):
    """bench.py's cpu_baseline runs the oracle on the CPUs the box GRANTS, not on the ones it shows (the one-GPU box: 256 logical
    CPUs, a cgroup quota of 16 -- rounds 1-4 ran 128 threads taking turns on them): cgroup v2 `cpu.max`, v1 quota / period,
    "max" and a missing hierarchy, always capped by the affinity mask."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    have = len(os.sched_getaffinity(0))
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("1600000 100000\n")
    assert bench.usable_cpus(v2) == min(16, have)
    (v2 / "cpu.max").write_text("150000 100000\n")   # 1.5 CPUs -> 2 threads
    assert bench.usable_cpus(v2) == min(2, have)
    (v2 / "cpu.max").write_text("max 100000\n")
    assert bench.usable_cpus(v2) == have
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("400000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.usable_cpus(v1) == min(4, have)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench.usable_cpus(v1) == have
    assert bench.usable_cpus(tmp_path / "none") == have

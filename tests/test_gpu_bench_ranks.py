"""Rehearsal of bench.py's multi-rank path on one GPU: two ranks share the card, the gather
runs on gloo through host copies, and rank 0 checks the assembled screen against a
single-rank render of the same frame.  (RCCL refuses two ranks on one device, so the nccl
gather itself is exercised with one rank by --force-dist.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_assemble_the_single_rank_frame():
    d = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29547", "bench.py", "--gpus", "2",
                "--steps", "1", "--warmup", "0", "--workload", "teapot_class_480p", "--backend", "gloo",
                "--verify", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["verify"]["screen_equals_single_rank_render"] is True
    assert d["rays_per_frame"]["primary"] == 640 * 480
    # what torch.distributed itself says about the run: two ranks went through the gather
    assert d["comm"]["world_size"] == 2 and d["comm"]["backend"] == "gloo" and d["comm"]["gather_ms"] > 0


def test_rccl_gather_with_one_rank():
    d = _bench([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--workload",
                "teapot_class_480p", "--force-dist", "--verify", "--no-cpu-baseline"])
    assert d["verify"]["screen_equals_single_rank_render"] is True
    assert d["comm"]["world_size"] == 1 and d["comm"]["backend"] == "nccl" and d["comm"]["gather_ms"] > 0


def test_frame_sequence_is_split_by_frames():
    """A frame sequence under several ranks: every rank renders whole frames (rank, rank + world,
    ...) and nothing is exchanged; rank 0's last picture equals a single-rank render of that
    frame and the frame rate counts every rank's frames."""
    d = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29549", "bench.py", "--gpus", "2",
                "--steps", "2", "--warmup", "1", "--workload", "teapot_seq_4k", "--backend", "gloo",
                "--verify", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["verify"]["screen_equals_single_rank_render"] is True
    assert d["rays_per_frame"]["primary"] == 2 * 3840 * 2160          # two frames per step
    assert abs(d["frames_per_s"] * d["ms_per_step"] * 1e-3 - 2.0) < 1e-6
    assert d["frames"]["per_rank"] == [2, 2] and d["frames"]["whole_frames_per_rank"] is True
    assert len(d["frames"]["seconds_per_rank"]) == 2 and d["comm"]["world_size"] == 2


def test_single_process_mode_drives_the_devices_itself():
    """bench.py --single-process: no launcher, rt_render_frame_multi over the named devices (here
    the one GPU named three times: threads, parts, placement; rows by peer copies)."""
    d = _bench([sys.executable, "bench.py", "--gpus", "3", "--single-process", "--devices", "0,0,0", "--steps", "2",
                "--warmup", "1", "--workload", "teapot_class_480p", "--no-cpu-baseline"])
    assert d["n_gpus"] == 3 and d["verify"]["screen_equals_single_rank_render"] is True
    assert d["rays_per_frame"]["primary"] == 640 * 480 and "rt_render_frame_multi" in d["config"]["sharding"]
    assert d["roofline"]["anyhit"]["tests_per_frame"] > 0

"""bench.py's own rank launcher: `--gpus N` with no WORLD_SIZE in the environment starts
the N ranks itself, before anything touches a GPU.  CPU only: the children here find no GPU and say so."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import json, subprocess, sys
sys.path.insert(0, %r)
import bench
seen = []
class FakeProc:
    def __init__(self, cmd, env=None):
        seen.append({"cmd": cmd, "env": {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}})
        self.rank = int(env["RANK"])
    def poll(self):
        return 0 if self.rank != FAIL else 7
    def terminate(self):
        pass
FAIL = int(sys.argv[1])
subprocess.Popen = FakeProc
rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "2"])
print(json.dumps({"rc": rc, "seen": seen, "torch": "torch" in sys.modules, "pt": "pathtracer_amd" in sys.modules,
                  "hip": any("amdhip" in (getattr(m, "__file__", "") or "") for m in sys.modules.values())}))
"""


def _probe(fail_rank):
    r = subprocess.run([sys.executable, "-c", PROBE % ROOT, str(fail_rank)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_launcher_children_get_the_rank_environment():
    out = _probe(-1)
    assert out["rc"] == 0
    assert [c["env"]["RANK"] for c in out["seen"]] == ["0", "1", "2", "3"]
    assert all(c["env"]["LOCAL_RANK"] == c["env"]["RANK"] and c["env"]["WORLD_SIZE"] == "4" and c["env"]["MASTER_ADDR"] == "127.0.0.1" for c in out["seen"])
    assert len({c["env"]["MASTER_PORT"] for c in out["seen"]}) == 1 and int(out["seen"][0]["env"]["MASTER_PORT"]) > 0
    assert all(c["cmd"][0] == sys.executable and c["cmd"][1].endswith("bench.py") and c["cmd"][2:] == ["--gpus", "4", "--steps", "2"] for c in out["seen"])
    # the parent never touches a GPU: it imported neither torch nor the library
    assert out["torch"] is False and out["pt"] is False and out["hip"] is False


def test_launcher_reports_a_failed_rank():
    assert _probe(2)["rc"] == 7


def test_bench_gpus_2_without_a_launcher_runs_two_ranks_and_fails_without_gpus():
    """End to end on a box without GPUs: both children start (each prints the no-device message) and the parent's exit
    status is theirs (3), not 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    import torch
    if torch.cuda.is_available():
        return  # (a GPU box: the launcher is exercised by the real thing)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert r.stderr.count("no HIP device visible") >= 1
    assert "exited with status 3" in r.stderr


def test_device_mapping_survives_per_rank_masking():
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.pick_device(r, 8) for r in range(8)] == list(range(8))   # the node's GPUs visible: one each
    assert [bench.pick_device(r, 1) for r in range(8)] == [0] * 8          # each rank masked to its own GPU: device 0
    assert [bench.pick_device(r, 4) for r in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    assert bench.pick_device(0, 0) == -1

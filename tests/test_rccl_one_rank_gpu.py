"""The N > 1 path's RCCL calls issued for real on a ONE-GPU box: a process group of one rank over the 'nccl' backend with
LECCR_ONE_RANK_WORLD=1 takes the collective path all the same - RCCL environment and communicator creation
(leccr_amd.parallel.setup_rccl_env), `all_gather_into_tensor` in AllGatherPacked (models/xvlm.py:50-70 semantics), the phased
`all_reduce(async_op=True)` of the gradient arena on the reduce stream with the optimizer fused into backward, the MAX-over-ranks
reduction and the barriers of bench.py's timed region, `all_gather_object` for the stage timers, and the sharded evaluation's row
gathers.  What a single rank cannot show is the DATA of other ranks (tests/test_dist_gpu.py: 2 and 4 gloo ranks on this GPU;
tests/test_rccl_gpu.py: >= 2 GPUs); what it does show is that every call is accepted by this RCCL build and changes nothing.
Each case is one child process (the parent test process never joins a process group)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(port, *argv, one_rank=True, extra_env=None):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LECCR_BENCH_BACKEND', 'LECCR_ONE_RANK_WORLD'):
        env.pop(k, None)
    if one_rank:
        env.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LECCR_ONE_RANK_WORLD='1',
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + list(argv), capture_output=True, text=True, env=env,
                       timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, lines                        # ONE JSON line on stdout, whatever RCCL prints
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize('cus', ['0', '16'])
def test_one_rank_rccl_world_runs_the_collective_train_path_and_changes_nothing(cus):
    """bench.py's train step (batch 32, bf16, dropout on, optimizer fused into backward) as rank 0 of a one-rank RCCL world against the same
    line with no process group: the same losses after the same number of steps (a one-rank all-gather / mean all-reduce is the identity),
    the parity probe against the oracle still inside its tolerance, and the stage timers show the collectives ran."""
    argv = ['--batch', '32', '--steps', '3', '--warmup', '1', '--roofline-steps', '0', '--no-cpu-baseline', '--no-reference-precision',
            '--parity-rows', '4']
    one, err = _bench(29821 + int(cus), *argv, extra_env={'LECCR_RCCL_CUS': cus})
    plain, _ = _bench(0, *argv, one_rank=False, extra_env={'LECCR_BENCH_DIAG': '1'})
    c = one['config']
    assert c['backend'] == 'nccl' and c['rccl_version'] and c['world'] == 1 and one['n_gpus'] == 1
    assert int(c['LECCR_RCCL_CUS']) == int(cus)
    assert plain['config']['backend'] is None
    # (the world path of the loss kernels - gathered columns, local row window - sums in another order than the single-rank one, and
    # bf16 steps with split-K atomics differ from run to run by ~1e-3 as it is)
    assert max(abs(a - b) for a, b in zip(one['config']['losses'], plain['config']['losses'])) <= 1e-2, (one['config']['losses'],
                                                                                                          plain['config']['losses'])
    assert one['parity']['ok'] and plain['parity']['ok']
    st = one['stage_ms_per_rank']['rank0']
    assert st['allgather_n'] == 2 and st['allreduce_exposed_n'] == 2 and st['backward_n'] == 2, st
    assert 'allreduce_exposed_n' not in plain['stage_ms_per_rank']['rank0']          # no process group: no gradient exchange attached
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', f'one_rank_rccl_cus{cus}.json'), 'w') as fh:
        json.dump({'one_rank_rccl': one, 'no_process_group': plain}, fh)


def test_one_rank_rccl_world_sharded_evaluation_gathers():
    """bench.py --mode eval as rank 0 of a one-rank RCCL world: the row gathers of the sharded evaluation (`_gather_rows_varlen`: a size
    exchange and two padded all-gathers) run over RCCL; the line's own parity block (scores against the oracle's, ranks) must hold."""
    one, _ = _bench(29841, '--mode', 'eval', '--eval-images', '96', '--eval-texts', '96', '--steps', '1', '--warmup', '1', '--dtype', 'fp32')
    assert one['n_gpus'] == 1 and one['value'] > 0
    assert one['parity']['ok'], one['parity']

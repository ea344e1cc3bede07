"""bench.py, N > 1: every rank is a supervisor that reads its worker's standard output (bench.supervise_rank).  The ONE line
that reaches the launcher is the worker's last complete one; a worker that dies or hangs after the headline measurement
(the marker) costs the sharded entries, not the headline, and the launcher sees exit code 0."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_supervisor(tmp_path, body, grace=2.0):
    worker = tmp_path / 'worker.py'
    worker.write_text(textwrap.dedent(body))
    code = (f'import sys; sys.path.insert(0, {ROOT!r}); import bench; '
            f'sys.exit(bench.supervise_rank([], grace_s={grace}, script={str(worker)!r}))')
    p = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120,
                       env=dict(os.environ, RANK='0'))
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    return p.returncode, lines


PROVISIONAL = '{"metric": "m", "value": 1.0, "config": {"workload": "w"}}'
FINAL = '{"metric": "m", "value": 1.0, "config": {"workload": "w", "sharded_joint_fit": {"value": 2.0}}}'


def test_final_line_wins(tmp_path):
    rc, lines = run_supervisor(tmp_path, f'''
        import os
        assert os.environ["LCMI_BENCH_WORKER"] == "1"
        print({PROVISIONAL!r}, flush=True)
        print("HEADLINE_DONE", flush=True)
        print({FINAL!r}, flush=True)
    ''')
    assert rc == 0 and len(lines) == 1
    assert json.loads(lines[0])['config']['sharded_joint_fit'] == {'value': 2.0}


def test_worker_that_dies_after_the_headline_costs_only_the_sharded_entries(tmp_path):
    rc, lines = run_supervisor(tmp_path, f'''
        import os
        print({PROVISIONAL!r}, flush=True)
        print("HEADLINE_DONE", flush=True)
        os.abort()                        # what a GPU fault in a peer-memory kernel does to a process
    ''')
    assert rc == 0 and len(lines) == 1
    d = json.loads(lines[0])
    assert d['value'] == 1.0 and 'error' in d['config']['sharded_joint_fit']


def test_worker_that_hangs_after_the_headline_is_ended_after_the_grace_period(tmp_path):
    rc, lines = run_supervisor(tmp_path, f'''
        import time
        print({PROVISIONAL!r}, flush=True)
        print("HEADLINE_DONE", flush=True)
        time.sleep(600)                   # a collective that never returns
    ''', grace=1.5)
    assert rc == 0 and len(lines) == 1
    assert 'killed after the grace period: True' in json.loads(lines[0])['config']['sharded_joint_fit']['error']


def test_rank_without_a_line_and_failure_before_the_headline(tmp_path):
    rc, lines = run_supervisor(tmp_path, '''
        print("HEADLINE_DONE", flush=True)   # a rank other than 0: marker only
    ''')
    assert rc == 0 and lines == []
    rc, lines = run_supervisor(tmp_path, '''
        import sys
        sys.exit(3)                          # failed before the headline: the launcher must see it
    ''')
    assert rc == 3 and lines == []


# ---- eight ranks: the protocol of bench.sharded_joint_fit's error paths, rehearsed on CPU over gloo ---------------------------
WORKER_8 = '''
import datetime, json, os, sys, time
sys.path.insert(0, {root!r})
import bench
import torch.distributed as dist
rank, world, scenario = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ["SCENARIO"]
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
line = {{"metric": "m", "value": 1.0, "config": {{"workload": "w"}}}}
if rank == 0:
    print(json.dumps(line), flush=True)
print("HEADLINE_DONE", flush=True)

def set_up():
    if scenario == "setup" and rank == 3:
        raise RuntimeError("hipMalloc failed (test)")      # one rank cannot build its local fit
    return 1

def timed():
    if scenario == "hang" and rank == 5:
        time.sleep(600)                                      # one rank never comes back from its timed loop

try:
    bench.all_ranks_agree("set-up of the local fit", set_up, world)
    bench.all_ranks_agree("timed iterations", timed, world)
    result = {{"value": 2.0}}
except Exception as e:
    result = {{"error": repr(e)}}
if rank == 0:
    line["config"]["sharded_joint_fit"] = result
    print(json.dumps(line), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def run_eight(tmp_path, scenario, grace):
    import socket
    worker = tmp_path / 'worker8.py'
    worker.write_text(WORKER_8.format(root=ROOT))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    code = (f'import sys; sys.path.insert(0, {ROOT!r}); import bench; '
            f'sys.exit(bench.supervise_rank([], grace_s={grace}, script={str(worker)!r}))')
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='8', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   SCENARIO=scenario, OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
    outs = [p.communicate(timeout=240) for p in procs]
    return [p.returncode for p in procs], [[l for l in o.decode().splitlines() if l.strip()] for o, _ in outs], [e.decode() for _, e in outs]


def test_eight_ranks_one_fails_during_set_up(tmp_path):
    """Rank 3 raises while it builds its local fit: EVERY rank raises with its message (nobody is left in a collective), rank 0
    reports the error inside the line, every supervisor exits 0 with the headline intact."""
    rcs, lines, errs = run_eight(tmp_path, 'setup', grace=60.0)
    assert rcs == [0] * 8
    assert [len(l) for l in lines] == [1] + [0] * 7
    d = json.loads(lines[0][0])
    assert d['value'] == 1.0 and 'rank 3' in d['config']['sharded_joint_fit']['error']
    assert 'hipMalloc failed (test)' in d['config']['sharded_joint_fit']['error']


def test_eight_ranks_one_hangs_in_the_timed_loop(tmp_path):
    """Rank 5 never returns from its timed iterations: the others wait for it in the agreement; after the grace period every
    supervisor ends its own worker, says so on stderr, keeps the headline line and exits 0."""
    rcs, lines, errs = run_eight(tmp_path, 'hang', grace=4.0)
    assert rcs == [0] * 8
    assert [len(l) for l in lines] == [1] + [0] * 7
    d = json.loads(lines[0][0])
    assert d['value'] == 1.0 and 'killed after the grace period: True' in d['config']['sharded_joint_fit']['error']
    assert all('did not finish within' in e for e in errs)

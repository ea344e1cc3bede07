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

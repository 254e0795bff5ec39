"""`python3 bench.py --gpus N` (the shape of the driver's command) launches its own ranks: the parent starts
torch.distributed.run before it imports torch or touches the GPU and hands the exit code on."""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_command(monkeypatch):
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    rc = bench.self_launch(4, ['--gpus', '4', '--steps', '3'])
    cmd = seen['cmd']
    assert rc == 7
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node')+1] == '4'
    assert cmd[cmd.index('--master-addr')+1] == '127.0.0.1'
    assert 0 < int(cmd[cmd.index('--master-port')+1]) < 65536
    k = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[k+1:] == ['--gpus', '4', '--steps', '3']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_main_launches_before_importing_torch(monkeypatch):
    """with WORLD_SIZE unset and --gpus > 1 main() must hand over to the launcher before any torch import"""
    import bench
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    called = {}

    def fake_launch(gpus, argv):
        called['gpus'], called['argv'] = gpus, list(argv)
        called['torch_cuda_initialised'] = 'torch' in sys.modules and sys.modules['torch'].cuda.is_initialized()
        return 0
    monkeypatch.setattr(bench, 'self_launch', fake_launch)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and called['gpus'] == 2 and called['argv'] == ['--gpus', '2']
    assert not called['torch_cuda_initialised']


def test_bare_command_spawns_the_ranks():
    """end to end on this CPU-only container: both ranks start (WORLD_SIZE = 2) and refuse loudly for want of a GPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible: the GPU rehearsal (PNL_BENCH_BACKEND=gloo) covers the full run')
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu', '--no-extra'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert p.stderr.count('bench.py needs a GPU') >= 2, p.stderr[-2000:]

"""debug driver of the row-slab operator: python tools/slab_debug.py <world> <P1|P2> <noRef>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler


def worker(rank, world, port, element, noRef):
    faulthandler.dump_traceback_later(60, exit=True, file=open('gpurun_out/slabdbg_{}.txt'.format(rank), 'w'))
    log = lambda *a: print('[rank {}]'.format(rank), *a, flush=True)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import numpy as np, torch, torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from pynucleus_amd import disc, P1_DoFMap, P2_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from oracle.oracle import OracleProblem
    mesh = disc(noRef)
    dm = (P2_DoFMap if element == 'P2' else P1_DoFMap)(mesh, PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {'target_order': 0.5}, zeroExterior=True, comm=True)
    log('builder ok')
    op = b.getDense(distributed=True)
    log('assembled', op, op.info)
    Aref, cref, _ = OracleProblem(b.tables).get_dense()
    x = np.cos(np.arange(dm.num_dofs)*0.37)
    y = op*x
    log('matvec err', float(np.abs(y-Aref@x).max()/np.abs(Aref@x).max()))
    A = op.toarray()
    log('full err', float(np.abs(A-Aref).max()/np.abs(Aref).max()))
    log('diag err', float(np.abs(op.diagonal-np.diag(Aref)).max()/np.abs(Aref).max()))
    dist.destroy_process_group()


if __name__ == '__main__':
    import torch.multiprocessing as mp
    world, element, noRef = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=worker, args=(r, world, 29811, element, noRef)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(timeout=100)
    for p in procs:
        if p.is_alive(): p.kill()
    print('exit codes', [p.exitcode for p in procs])

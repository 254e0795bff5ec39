"""N > 1 host logic on CPU (gloo, world_size 2): the pair split over ranks is a partition, and the distributed
operator (sum of per-rank parts, matvec = local product + all-reduce of the N-vector, clusterMethodCy.pyx:3127-3154)
reproduces the full operator.  The per-rank parts are produced by the CPU oracle with the reference's cellNo1
split (NA:1280-1285) because no GPU exists here; the GPU path uses the same split for touching pairs / boundary
and the tile deal checked below for distant pairs."""
import os
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from pynucleus_amd.builder import upper_tiles, tiles_of_rank, cell_range_of_rank, tile_cells


def test_tile_deal_is_a_partition():
    for nc, dpe, size in [(24576, 3, 8), (6144, 3, 3), (1000, 6, 4), (50, 3, 2)]:
        T = tile_cells(dpe)
        allt = upper_tiles(nc, T)
        nb = (nc+T-1)//T
        assert allt.shape[0] == nb*(nb+1)//2 and (allt[:, 0] <= allt[:, 1]).all()
        parts = [tiles_of_rank(nc, T, r, size) for r in range(size)]
        cat = np.concatenate(parts)
        assert cat.shape[0] == allt.shape[0]
        assert len({tuple(t) for t in cat}) == allt.shape[0]
        # balanced to within one tile, heavy (diagonal) tiles spread over all ranks
        assert max(len(p) for p in parts)-min(len(p) for p in parts) <= 1
        ranges = [cell_range_of_rank(nc, r, size) for r in range(size)]
        assert ranges[0][0] == 0 and ranges[-1][1] == nc and all(ranges[i][1] == ranges[i+1][0] for i in range(size-1))


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel, nonlocalTables
    from oracle.oracle import OracleProblem
    mesh = disc(2)
    dm = P1_DoFMap(mesh, PHYSICAL)
    O = OracleProblem(nonlocalTables(dm, getFractionalKernel(2, 0.5), {}))
    c0, c1 = cell_range_of_rank(mesh.num_cells, rank, world)
    part = torch.from_numpy(O.get_dense(c0, c1)[0])
    x = torch.from_numpy(np.linspace(-1., 1., dm.num_dofs))
    dist.broadcast(x, src=0)
    y = part@x
    dist.all_reduce(y)
    full = O.get_dense()[0]
    err = float(np.abs(y.numpy()-full@x.numpy()).max()/np.abs(full@x.numpy()).max())
    # the reference's own variant: all-reduce the matrix (NA:1449-1450)
    dist.all_reduce(part)
    err2 = float(np.abs(part.numpy()-full).max()/np.abs(full).max())
    if rank == 0:
        out.put((err, err2))
    dist.destroy_process_group()


def test_gloo_world2_distributed_operator():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500+os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    err, err2 = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-13 and err2 < 1e-14

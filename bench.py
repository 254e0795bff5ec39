#!/usr/bin/env python3
"""Benchmark of the hot path: dense nonlocal assembly, element-pairs/s (BASELINE.json metric).

One "step" = one complete pass of nonlocalBuilder.getDense()'s work over the workload on the GPU(s):
zero the N x N block in HBM, classify + integrate + scatter every element pair (c1 <= c2), the
Omega x Omega^c boundary term, mirror.  Inputs (mesh, DoF map, quadrature tables) are resident in HBM
before the timed region starts.  Workload: 2D unit disc, P1, fractional kernel s = 0.5, horizon = inf,
normalised, target_order 0.5, zeroExterior (BASELINE.json configs[1]); the mesh is the reference's
uniform_disc refined --noRef times.  configs[1]'s "~2x10^4 DoFs" lies between noRef 6 (12 097 DoFs, 3.02e8 pairs) and noRef 7
(48 769 DoFs, 98 304 cells, 4.83e9 pairs); the default is noRef 7, the largest refinement whose dense N x N block (19 GB) fits
one MI355X (noRef 8 needs 307 GB) and the size at which the 1/2/4/8-GPU runs are not dominated by per-rank fixed costs.

N > 1 (python -m torch.distributed.run ... bench.py --gpus N): the element pairs of the SAME problem are
dealt over the ranks (strong scaling, no collective in the assembly path; every rank holds its partial
N x N block, the operator is their sum and its matvec all-reduces an N-vector over RCCL).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X vector fp64 (vendor figure = half of the fp32 vector peak 157.3 in MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.               # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_flops(orders, npoints, dpe=3):
    """F_pair = n_q (F_gamma + 12) + 3 E n_q + E with F_gamma = 7 (F_pow counted as 1 flop),
    E = (2 dpe)(2 dpe + 1)/2 local entries, n_q = n(q)^2 point pairs."""
    E = (2*dpe)*(2*dpe+1)//2
    total = 0
    for q, cnt in orders.items():
        nq = npoints(q)**2
        total += cnt*(nq*(7+12)+3*E*nq+E)
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--noRef', type=int, default=7)
    ap.add_argument('--s', type=float, default=0.5)
    ap.add_argument('--sectors', type=int, default=6, help='triangles of the initial fan (6 = the reference disc; 12 at noRef 7 gives 97 921 DoFs)')
    ap.add_argument('--cpu-seconds', type=float, default=15., help='target CPU time of the oracle sample (rank 0, N=1 only)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--solve', action='store_true', help='also run the CG-Jacobi solve of configs[1] (reported, not timed into value)')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd._lib import PNL_FLAG_SYMMETRIC_FLUSH

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node {} (WORLD_SIZE={})'.format(args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the assembly path has no CPU implementation'
    # rehearsal of the N > 1 path on a one-GPU box: PNL_BENCH_BACKEND=gloo puts every rank on cuda:0 and reduces on the host
    backend = os.environ.get('PNL_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    red_dev = dev if backend == 'nccl' else torch.device('cpu')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- setup (not timed): tables on the host, upload into HBM ---------------------------------------------
    mesh = disc(args.noRef, sectors=args.sectors)
    dm = P1_DoFMap(mesh, PHYSICAL)
    kernel = getFractionalKernel(2, args.s)
    builder = nonlocalBuilder(dm, kernel, {'target_order': 0.5}, zeroExterior=True, comm=(True if world > 1 else None))
    ctx = builder.context()
    N, nc = dm.num_dofs, mesh.num_cells
    A = torch.zeros((N, N), dtype=torch.float64, device=dev)
    if world > 1:
        tiles = builder.tiles_for_rank(rank, world)
        from pynucleus_amd.builder import cell_range_of_rank
        c0, c1 = cell_range_of_rank(nc, rank, world)

    def step():
        A.zero_()
        if world == 1:
            ctx.assemble_dense(A.data_ptr(), A.stride(0), True, 0, nc)
        else:
            # N > 1: cross contributions are written on both sides by the flush (its work is shared by the ranks) instead
            # of the N^2 mirror pass (which is not)
            ctx.assemble_dense_tiles(A.data_ptr(), A.stride(0), True, tiles, c0, c1, flags=PNL_FLAG_SYMMETRIC_FLUSH)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    tile_ms, pure_ms, wl_ms, phase_acc = [], [], [], {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter()-t0
    # phases of the last step from HIP events recorded on the stream the kernels ran on
    ms = ctx.phase_ms()
    cnt = ctx.counters()
    # a few extra (untimed) steps to average the dominant kernel's duration from its own HIP events
    for _ in range(3):
        step()
        torch.cuda.synchronize(dev)
        m = ctx.phase_ms()
        tile_ms.append(m['tiles'])
        pure_ms.append(m['tiles_uniform'])
        wl_ms.append(m['worklist'])
        for k, v in m.items():
            phase_acc[k] = phase_acc.get(k, 0.)+v/3.
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        pr = torch.tensor([cnt['numAssembledCellPairs']], dtype=torch.float64, device=red_dev)
        dist.all_reduce(pr)
        pairs_total = float(pr.item())
    else:
        pairs_total = float(cnt['numAssembledCellPairs'])
    value = pairs_total*args.steps/elapsed

    # ---- roofline of the dominant kernel: algorithmic flops / its own event-timed duration --------------------------
    # Distant pairs of the orders packed into the tile rule table (<= 16 points) are integrated by two kernels:
    # k_tile_pure (tiles whose 4096 pairs are all of order 2) and k_tile_distant (all other tiles).  The counters tell
    # how many order-2 pairs the uniform-tile kernel took; the rest of the histogram belongs to k_tile_distant.
    T = builder.tables
    tile_orders = {q: c for q, c in cnt['orders'].items() if T.num_points(q) <= 16 and q < 18}
    pure_pairs = cnt.get('uniformTilePairs', 0)
    mixed_orders = dict(tile_orders)
    mixed_orders[2] = mixed_orders.get(2, 0)-pure_pairs
    flops_mixed = algorithmic_flops(mixed_orders, T.num_points)
    flops_pure = algorithmic_flops({2: pure_pairs}, T.num_points)
    mixed_s = 1e-3*float(np.mean(tile_ms))
    pure_s = 1e-3*float(np.mean(pure_ms))
    dominant = 'k_tile_distant' if mixed_s >= pure_s else 'k_tile_pure'
    dom_flops, dom_s = (flops_mixed, mixed_s) if dominant == 'k_tile_distant' else (flops_pure, pure_s)
    achieved = dom_flops/dom_s/1e12 if dom_s > 0 else 0.
    traffic = None
    pmc_fn = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc_fn):
        with open(pmc_fn) as f:
            rec = json.load(f)
        key = 'noRef{}'.format(args.noRef)
        if key in rec and world == 1 and args.sectors == 6:
            traffic = rec[key].get(dominant+'_hbm_bytes_per_launch')
    # HBM view of the same launches: the algorithmic minimum is one write of the upper block triangle they fill
    hbm_alg_bytes = 8.*N*N/2
    both_s = mixed_s+pure_s
    roofline = dict(bound='fp64_valu', kernel=dominant, achieved=achieved, peak=FP64_VECTOR_PEAK_TFLOPS, unit='TFLOP/s',
                    frac=achieved/FP64_VECTOR_PEAK_TFLOPS, traffic=traffic, algorithmic_flops_per_launch=dom_flops,
                    kernel_ms=1e3*dom_s,
                    other_tile_kernel=dict(kernel='k_tile_pure' if dominant == 'k_tile_distant' else 'k_tile_distant',
                                           algorithmic_flops_per_launch=flops_pure if dominant == 'k_tile_distant' else flops_mixed,
                                           kernel_ms=1e3*(pure_s if dominant == 'k_tile_distant' else mixed_s),
                                           achieved=((flops_pure/pure_s) if dominant == 'k_tile_distant' else (flops_mixed/mixed_s))/1e12
                                           if min(pure_s, mixed_s) > 0 else 0.),
                    tile_phase_achieved=(flops_mixed+flops_pure)/both_s/1e12 if both_s > 0 else 0.,
                    hbm_algorithmic_GBs=hbm_alg_bytes/both_s/1e9 if both_s > 0 else 0., hbm_peak_GBs=HBM_PEAK_GBS)

    out = dict(metric='element-pairs/sec assembled (2D P1 fractional s=0.5, dense) + % fp64 roofline', value=value,
               unit='element-pairs/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3*elapsed/args.steps,
               higher_is_better=True, scaling='strong', vs_baseline=None, dtype='f64', data='synthetic',
               config=dict(workload='2D unit disc ({}uniform_disc refined {}x, {} cells), P1, {} DoFs, fractional s={}, horizon=inf, '
                           'dense getDense incl. zeroExterior; {} element pairs/step'.format('' if args.sectors == 6 else '{}-sector '.format(args.sectors), args.noRef, nc, N, args.s, int(pairs_total)),
                           noRef=args.noRef, num_dofs=N, num_cells=nc, parallelism='pairs dealt over {} GPU(s)'.format(world)),
               roofline=roofline,
               phases_ms={k: round(v, 4) for k, v in phase_acc.items()},
               kernel_evaluations_per_step=cnt['numIntegrations'] if world == 1 else None)

    # ---- CPU baseline: the C oracle (single thread, reference loop order) on a bounded sample of the same workload ----
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle.oracle import OracleProblem
        O = OracleProblem(T)
        store = N*N*8 <= 4e9
        # rows of cells around nc/2: about cpu_seconds of work at ~1.1 us per pair
        want_pairs = args.cpu_seconds/1.1e-6
        k = max(1, min(nc//2, int(want_pairs/(nc/2))))
        ca, cb = nc//2, nc//2+k
        _, ccnt, secs = O.get_dense(ca, cb, store=store)
        cpu_pairs = ccnt['numAssembledCellPairs']
        cpu_t = secs[0]+secs[1]
        out['cpu_baseline'] = dict(value=cpu_pairs/cpu_t, unit='element-pairs/s', cores=1, kind='port',
                                   sample='oracle/nl_oracle.c (C restatement of the reference loop, gcc -O3, 1 thread) on cell rows '
                                          '[{}, {}) x all partners c2 >= c1 of the same workload: {} pairs incl. their share of the '
                                          'boundary term in {:.1f} s{}'.format(ca, cb, cpu_pairs, cpu_t,
                                                                               '' if store else ' (count-only: no N x N scatter target)'),
                                   seconds=cpu_t, speedup_gpu_over_cpu=value/(cpu_pairs/cpu_t))
    if args.solve and world == 1:
        from pynucleus_amd.linear_operators import Dense_LinearOperator
        op = Dense_LinearOperator(A, ctx)
        b = torch.from_numpy(np.asarray(dm.assembleRHS(1.0))).to(dev)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        u, its, res = op.solve_cg_jacobi(b, tol=1e-8, maxiter=5000)
        torch.cuda.synchronize(dev)
        out['cg_jacobi'] = dict(iterations=its, residual=res, seconds=time.perf_counter()-t1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

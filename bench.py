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

N > 1 (python -m torch.distributed.run ... bench.py --gpus N): the block rows of the upper block triangle of the SAME
problem are dealt over the ranks by work (strong scaling, no collective in the assembly path); every rank holds the one-sided
slab of its rows, about N^2 / (2 N_gpus) doubles, the operator is DistributedSlab_LinearOperator and its matvec all-reduces
an N-vector over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
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


def flops_from_counters(cnt, dpe):
    """SURVEY.md 8(d) summed over what was integrated: every kernel evaluation costs (F_gamma + 12) + 3 E flops, every
    assembled pair E more (F_gamma = 7 with F_pow counted as 1, E = (2 dpe)(2 dpe + 1)/2 local entries)."""
    E = (2*dpe)*(2*dpe+1)//2
    return cnt['numIntegrations']*(19+3*E)+E*cnt['numAssembledCellPairs']


def cpu_baseline(args, T, N, nc, gpu_value):
    """The reference runs one MPI rank per core over cellNo1 ranges (NA:1280-1285), single-threaded inside a rank.
    (i) one core, (ii) all host cores with one worker per core over cell ranges of the same workload (count-only at this
    size: N x N doubles do not fit the sample's budget), (iii) one core WITH the N x N scatter (NA:204-253) on the noRef 6
    member of the same family, whose 1.2 GB block fits."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import OracleProblem
    O = OracleProblem(T)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:                                                     # cgroup CPU quota (the GPU box grants a share of its host cores)
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(float(quota)/float(period)+0.5)))
    except (OSError, ValueError):
        pass
    cores = min(cores, int(os.environ.get('PNL_BENCH_CPU_WORKERS', '64')))
    per_pair = 1.1e-6
    # (i) one core: rows of cells around nc/2, about a third of the CPU budget
    k1 = max(1, min(nc//2, int(args.cpu_seconds/3./per_pair/(nc/2))))
    ca, cb = nc//2, nc//2+k1
    _, c1, secs = O.get_dense(ca, cb, store=False)
    t1 = secs[0]+secs[1]
    one = c1['numAssembledCellPairs']/t1
    # (ii) all cores: every worker its own range of cell rows (ctypes releases the GIL; the oracle has no shared state)
    kk = max(1, min((nc-cb)//cores, int(args.cpu_seconds/3./per_pair/(nc/2))))
    ranges = [(cb+i*kk, cb+(i+1)*kk) for i in range(cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(lambda r: O.get_dense(r[0], r[1], store=False), ranges))
    tall = time.perf_counter()-t0
    pairs_all = sum(r[1]['numAssembledCellPairs'] for r in res)
    allc = pairs_all/tall
    out = dict(value=allc, unit='element-pairs/s', cores=cores, kind='port',
               sample='oracle/nl_oracle.c (C restatement of the reference loop, gcc -O3): {} workers, one per host core (nproc = {}), '
                      'each over its own cellNo1 range of {} cell rows x all partners c2 >= c1 of the same workload, {} pairs incl. '
                      'their share of the boundary term in {:.1f} s wall (count-only: no N x N scatter target at this size)'.format(
                          cores, cores, kk, pairs_all, tall),
               seconds=tall, speedup_gpu_over_cpu=gpu_value/allc,
               one_core=dict(value=one, cores=1, sample='cell rows [{}, {}) x all partners, {} pairs in {:.1f} s (count-only)'.format(
                   ca, cb, c1['numAssembledCellPairs'], t1), speedup_gpu_over_cpu=gpu_value/one))
    # (iii) the scatter-included rate on a size whose block fits on the host
    try:
        from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
        from pynucleus_amd.local_matrix import nonlocalTables
        mesh6 = disc(6)
        dm6 = P1_DoFMap(mesh6, PHYSICAL)
        T6 = nonlocalTables(dm6, getFractionalKernel(2, args.s), {'target_order': 0.5}, True)
        O6 = OracleProblem(T6)
        nc6 = mesh6.num_cells
        k6 = max(1, min(nc6//2, int(args.cpu_seconds/3./per_pair/(nc6/2))))
        _, c6, s6 = O6.get_dense(nc6//2, nc6//2+k6, store=True)
        t6 = s6[0]+s6[1]
        out['one_core_storing'] = dict(value=c6['numAssembledCellPairs']/t6, cores=1,
                                       sample='noRef 6 (N = {}, 1.2 GB block on the host), cell rows [{}, {}) x all partners with the '
                                              'N x N scatter (addToMatrixElemElemSym), {} pairs in {:.1f} s'.format(
                                                  dm6.num_dofs, nc6//2, nc6//2+k6, c6['numAssembledCellPairs'], t6))
    except MemoryError as e:                                 # small hosts: the storing sample is optional
        out['one_core_storing'] = dict(error=repr(e))
    return out


def extra_configs(dev):
    """Short legs of the other BASELINE.json configurations on one GPU: device time from the library's HIP events on its
    stream, wall time around the builder call, pairs/s and the SURVEY 8(d) fraction of the vector fp64 peak."""
    import numpy as np
    import torch
    from pynucleus_amd import (disc, uniformSquare, P1_DoFMap, P2_DoFMap, PHYSICAL, NO_BOUNDARY, getFractionalKernel, getKernel,
                               INDICATOR)
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import layersFractionalOrder
    res = {}

    def sync():
        torch.cuda.synchronize(dev)

    def dense_leg(name, dm, kernel, dpe, reps=3):
        b = nonlocalBuilder(dm, kernel, {'target_order': 0.5}, zeroExterior=True)
        ms, wall, cnt = [], [], None
        for rep in range(reps+1):                            # the first call carries the uploads and tile lists
            sync(); t0 = time.perf_counter()
            A = b.getDense()
            sync(); t1 = time.perf_counter()
            cnt = A.info['counters']
            last_ms = A.info['phase_ms']
            if rep:
                ms.append(A.info['phase_ms']['total']); wall.append(t1-t0)
            first = t1-t0 if rep == 0 else first
            del A
        # median of the repetitions: a context of the previous leg that the garbage collector frees late (hipFree of its 20 GB)
        # would otherwise show up as one slow repetition here
        dev_s, pairs = 1e-3*float(np.median(ms)), cnt['numAssembledCellPairs']
        fl = flops_from_counters(cnt, dpe)
        res[name] = dict(num_dofs=dm.num_dofs, num_cells=dm.mesh.num_cells, element_pairs=pairs, device_ms=1e3*dev_s,
                         wall_ms=1e3*float(np.median(wall)), first_call_ms=1e3*first, pairs_per_s=pairs/dev_s,
                         algorithmic_tflops=fl/dev_s/1e12, frac_fp64_peak=fl/dev_s/1e12/FP64_VECTOR_PEAK_TFLOPS,
                         phases_ms={k: round(v, 3) for k, v in last_ms.items()},
                         kernel_ms={k: round(v, 3) for k, v in b.dense_context().kernel_ms().items()})
        del b
        gc.collect()
        torch.cuda.empty_cache()
        sync()

    legs = []
    # C5: P2, variable order (3 layers), dense, noRef 6
    def c5():
        mesh = disc(6)
        dm = P2_DoFMap(mesh, PHYSICAL)
        orders = np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]])
        dense_leg('C5_P2_layers_dense_noRef6', dm, getFractionalKernel(2, layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), orders)), 6)
    # P2, constant order s = 1/2 (the kernel VERDICT r01 asked to raise)
    def p2():
        mesh = disc(6)
        dense_leg('P2_const_dense_noRef6', P2_DoFMap(mesh, PHYSICAL), getFractionalKernel(2, 0.5), 6)
    # P1 with a general exponent (s = 0.4: no rsqrt shortcut, the table-driven power) at the headline size
    def s04():
        dense_leg('P1_s0.4_dense_noRef7', P1_DoFMap(disc(7), PHYSICAL), getFractionalKernel(2, 0.4), 3, reps=2)
    # C3: square 129^2, constant kernel, delta = 0.1, getSparse
    def c3():
        mesh = uniformSquare(129)
        dm = P1_DoFMap(mesh, NO_BOUNDARY)
        b = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=0.1), {}, zeroExterior=False)
        out = []
        for rep in range(3):
            sync(); t0 = time.perf_counter()
            A = b.getSparse()
            sync(); t1 = time.perf_counter()
            out.append((t1-t0, A.info['interior_ms'], A.info['counters'], A.nnz))
            del A
        c = out[-1][2]
        dev_s = 1e-3*out[-1][1]
        fl = flops_from_counters(c, 3)
        res['C3_square129_constant_delta0.1_getSparse'] = dict(
            num_dofs=dm.num_dofs, num_cells=mesh.num_cells, element_pairs=c['numAssembledCellPairs'], kernel_evaluations=c['numIntegrations'],
            nnz=out[-1][3], device_ms=out[-1][1], wall_ms=1e3*out[-1][0], first_call_ms=1e3*out[0][0],
            pairs_per_s=c['numAssembledCellPairs']/dev_s, algorithmic_tflops=fl/dev_s/1e12, frac_fp64_peak=fl/dev_s/1e12/FP64_VECTOR_PEAK_TFLOPS)
    # C4: disc noRef 7, s = 0.75, H2: near field + far-field setup end to end, matvec
    def c4():
        mesh = disc(7)
        dm = P1_DoFMap(mesh, PHYSICAL)
        b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
        walls = []
        for rep in range(3):
            sync(); t0 = time.perf_counter()
            h2 = b.getH2()
            sync(); walls.append(time.perf_counter()-t0)
        near = h2.Anear
        c = near.info.get('counters', {})
        x = torch.randn(dm.num_dofs, dtype=torch.float64, device=dev)
        for _ in range(3):
            y = h2.matvec(x)
        sync(); t0 = time.perf_counter()
        for _ in range(20):
            y = h2.matvec(x)
        sync()
        mv = (time.perf_counter()-t0)/20
        # a builder keeps tree / plans / pattern of its DoFMap: the second call is the device work + far-field setup
        b_cold = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
        sync(); t0 = time.perf_counter()
        b_cold.getH2()
        sync(); cold = time.perf_counter()-t0
        del b_cold
        r = dict(num_dofs=dm.num_dofs, num_cells=mesh.num_cells, getH2_first_ms=1e3*walls[0], getH2_new_builder_ms=1e3*cold,
                 getH2_ms=1e3*walls[-1], getH2_note='getH2_ms: repeated call on one builder (tree, tile plan, pattern, far-field plan cached); '
                 'getH2_new_builder_ms: everything rebuilt in a warm process', matvec_ms=1e3*mv,
                 dense_matvec_hbm_floor_ms=1e3*8.*dm.num_dofs**2/(HBM_PEAK_GBS*1e9))
        if c:
            dev_s = 1e-3*near.info['interior_ms']
            fl = flops_from_counters(c, 3)
            r.update(near_field_element_pairs=c['numAssembledCellPairs'], near_field_device_ms=near.info['interior_ms'],
                     near_field_nnz=near.nnz, pairs_per_s=c['numAssembledCellPairs']/dev_s, algorithmic_tflops=fl/dev_s/1e12,
                     frac_fp64_peak=fl/dev_s/1e12/FP64_VECTOR_PEAK_TFLOPS)
        # the solve of that configuration: (-Laplace)^s u = 1 with the H2 operator, Jacobi-CG against multigrid-preconditioned CG
        # (H2 operator on the finest level, dense operators below: the operator-agnostic cycle of pynucleus_amd/multigrid.py)
        try:
            from pynucleus_amd.solvers import cg as _cg
            from pynucleus_amd.multigrid import fractionalHierarchy, multigrid
            bb = torch.from_numpy(np.asarray(dm.assembleRHS(1.0))).to(dev)
            _cg(h2, bb, tol=1e-8, maxiter=5)
            sync(); t0 = time.perf_counter()
            xj, itj, _ = _cg(h2, bb, tol=1e-8, maxiter=2000)
            sync(); tj = time.perf_counter()-t0
            sync(); t0 = time.perf_counter()
            Hh = fractionalHierarchy('disc', 7, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, matrixFormat='H2', h2MinDoFs=20000)
            mgh = multigrid(Hh)
            sync(); th = time.perf_counter()-t0
            mgh.cg(bb, tol=1e-8, maxiter=3)
            sync(); t0 = time.perf_counter()
            xm, itm, _ = mgh.cg(bb, tol=1e-8)
            sync(); tm = time.perf_counter()-t0
            r.update(solve_cg_jacobi_iterations=itj, solve_cg_jacobi_ms=1e3*tj, solve_cg_mg_iterations=itm, solve_cg_mg_ms=1e3*tm,
                     solve_hierarchy_s=th, solve_difference=float(torch.linalg.norm(xm-xj)/torch.linalg.norm(xj)))
            del Hh, mgh
        except Exception as e:
            r['solve_error'] = repr(e)
        res['C4_disc_noRef7_s0.75_H2'] = r
    # the north star's "~10^5 DoFs on 1 MI355X": 12-sector fan refined 7 times, P1, dense (76 GB block)
    def big():
        mesh = disc(7, sectors=12)
        dense_leg('dense_97537dofs_P1_s0.5', P1_DoFMap(mesh, PHYSICAL), getFractionalKernel(2, 0.5), 3, reps=2)
    # BASELINE configs[4] at its stated size: P2, three-layer variable order, ~10^5 DoFs -- here on ONE MI355X (76 GB block + 80 GB
    # block-slot storage)
    def c5big():
        mesh = disc(6, sectors=12)
        orders = np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]])
        dense_leg('C5_P2_layers_dense_97537dofs', P2_DoFMap(mesh, PHYSICAL),
                  getFractionalKernel(2, layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), orders)), 6, reps=2)
    # solver side: hierarchy of the headline operator (levels 0 .. 7, assembled on the device), multigrid-
    # preconditioned CG for f = 1, and the GEMV it is made of against the HBM roofline (8 N^2 bytes per product)
    def solver():
        from pynucleus_amd.multigrid import fractionalHierarchy, multigrid
        sync(); t0 = time.perf_counter()
        H = fractionalHierarchy('disc', 7, getFractionalKernel(2, 0.5), {'target_order': 0.5})
        sync(); t_h = time.perf_counter()-t0
        dm = H.finest['DoFMap']
        A = H.finest['A']
        mg = multigrid(H)
        b = torch.from_numpy(np.asarray(dm.assembleRHS(1.0))).to(dev)
        x, its, hist = mg.cg(b, tol=1e-8)
        sync(); t0 = time.perf_counter()
        x, its, hist = mg.cg(b, tol=1e-8)
        sync(); t_cg = time.perf_counter()-t0
        sync(); t0 = time.perf_counter()
        for _ in range(5):
            mg.cycle(b)
        sync(); t_cyc = (time.perf_counter()-t0)/5
        xj, itj, resj = A.solve_cg_jacobi(b, tol=1e-8, maxiter=2000)
        sync(); t0 = time.perf_counter()
        xj, itj, resj = A.solve_cg_jacobi(b, tol=1e-8, maxiter=2000)
        sync(); t_j = time.perf_counter()-t0
        n = dm.num_dofs
        y = torch.empty_like(b)
        ctx = A.ctx
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        for _ in range(3):
            ctx.gemv_axpby(A.A.data_ptr(), A.A.stride(0), n, n, x.data_ptr(), -1., 1., b.data_ptr(), y.data_ptr())
        ev0.record()
        for _ in range(20):
            ctx.gemv_axpby(A.A.data_ptr(), A.A.stride(0), n, n, x.data_ptr(), -1., 1., b.data_ptr(), y.data_ptr())
        ev1.record(); sync()
        gemv_ms = ev0.elapsed_time(ev1)/20
        gbs = 8.*n*n/(gemv_ms*1e-3)/1e9
        # one Crank-Nicolson step of the fractional heat equation (pnl_theta_step: rhs, then cg-mg on M/dt + S/2) at 12,097 DoFs
        heat = {}
        try:
            from pynucleus_amd.multigrid import CrankNicolson
            H6 = fractionalHierarchy('disc', 6, getFractionalKernel(2, 0.5), {'target_order': 0.5}, buildMass=True)
            dm6 = H6.finest['DoFMap']
            st = CrankNicolson(H6, (dm6.mesh.h)**0.5, theta=0.5, tol=1e-10)
            u6 = torch.from_numpy(np.asarray(dm6.interpolate(lambda p: max(1.-p[0]**2-p[1]**2, 0.)**0.5))).to(dev)
            f6 = np.asarray(dm6.assembleRHS(1.0))
            st.step(0., u6, f6)
            sync(); t0 = time.perf_counter()
            for k in range(5):
                st.step(0., u6, f6)
            sync()
            heat = dict(heat_num_dofs=dm6.num_dofs, heat_dt=st.dt, heat_step_ms=1e3*(time.perf_counter()-t0)/5, heat_cg_mg_iterations=st.iterations[-1])
            del H6, st
        except Exception as e:
            heat = dict(heat_error=repr(e))
        res['solver_cg_mg_disc_noRef7'] = dict(heat, 
            num_dofs=n, levels=[L['A'].num_rows for L in H.getLevelList()], hierarchy_assembly_s=round(t_h, 3),
            cg_mg_iterations=its, cg_mg_ms=1e3*t_cg, final_residual=hist[-1], vcycle_ms=1e3*t_cyc,
            cg_jacobi_iterations=itj, cg_jacobi_ms=1e3*t_j,
            residual_check=float(torch.linalg.norm(b-A.matvec(x))/torch.linalg.norm(b)),
            gemv_ms=gemv_ms, gemv_GBs=gbs, gemv_frac_hbm_peak=gbs/HBM_PEAK_GBS)
    for name, leg in (('P2', p2), ('C5', c5), ('P1_s0.4', s04), ('C3', c3), ('C4', c4), ('solver', solver), ('dense_1e5', big), ('C5_1e5', c5big)):
        t0 = time.perf_counter()
        try:
            leg()
        except Exception as e:                               # a failing leg must not take the headline line with it
            res[name+'_error'] = repr(e)
        res.setdefault('leg_seconds', {})[name] = round(time.perf_counter()-t0, 2)
        torch.cuda.empty_cache()
    return res


def self_launch(gpus, argv):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks as children through torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1) and hand their exit code on.  Called BEFORE this process imports torch or touches the GPU:
    a process that has initialised the GPU must never be replaced or forked into ranks."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)]+list(argv)
    return subprocess.call(cmd, env=env)


def gather_ranks(value, world, red_dev):
    """one float per rank -> list on every rank (all_gather of a one-element tensor: RCCL on the GPU, gloo on the host)"""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=red_dev)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def operator_leg(args, world, rank, dev, red_dev, backend, builder, ctx, dm, A, slab_maps, device_ms, sync):
    """matvec and Jacobi-CG with the operator the timed steps assembled (untimed into `value`; reported beside it)."""
    import numpy as np
    import torch
    from pynucleus_amd.linear_operators import Dense_LinearOperator, DistributedSlab_LinearOperator
    from pynucleus_amd.solvers import cg
    N = dm.num_dofs
    if world == 1:
        op = Dense_LinearOperator(A, ctx, symmetric=True)       # s = const: a symmetric operator, applied from its upper triangle
        local_bytes = 8*N*int(A.stride(0))
    else:
        rows, cols = slab_maps
        dblocks = torch.zeros(ctx.diag_blocks_size(), dtype=torch.float64, device=dev)
        if rows.shape[0]:
            ctx.get_diag_blocks(dblocks.data_ptr())
        op = DistributedSlab_LinearOperator(A, dblocks, rows, cols, N, ctx, None)
        ctx._slab_owner = op
        local_bytes = op.local_bytes()
    x = torch.from_numpy(np.random.default_rng(0).standard_normal(N)).to(dev)
    for _ in range(3):
        y = op.matvec(x)
    sync(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        y = op.matvec(x)
    sync()
    mv = max(gather_ranks((time.perf_counter()-t0)/reps, world, red_dev))
    b = torch.from_numpy(np.asarray(dm.assembleRHS(1.0))).to(dev)
    sync(); t0 = time.perf_counter()
    if world == 1:
        u, its, res = op.solve_cg_jacobi(b, tol=1e-8, maxiter=5000)
        res_final = float(res)
    else:
        u, its, res = cg(op, b, tol=1e-8, maxiter=5000)
        res_final = float(res[-1])
    sync()
    t_cg = max(gather_ranks(time.perf_counter()-t0, world, red_dev))
    dms = gather_ranks(device_ms, world, red_dev)
    lb = gather_ranks(local_bytes, world, red_dev)
    energy = float(torch.dot(b, u))
    return dict(matvec_ms=1e3*mv, matvec_algorithmic_GBs=4.*N*N/mv/1e9 if world == 1 else sum(lb)/mv/1e9,
                matvec_note=('k_gemv_two_sided on the upper triangle of the symmetric N x N block: 4 N^2 bytes per product (A x and A^T x in one sweep)' if world == 1 else
                             'local one-sided slab products (A\' x and A\'^T x in ONE sweep) + one all-reduce of the N-vector ({} backend); GB/s = bytes of all slabs / time'.format(backend)),
                cg_jacobi_iterations=int(its), cg_jacobi_ms=1e3*t_cg, cg_residual=res_final, energy_b_dot_u=energy,
                device_ms_per_rank=[round(v, 3) for v in dms], device_ms_min=min(dms), device_ms_max=max(dms),
                operator_bytes_per_rank=[int(v) for v in lb], operator_bytes_total=int(sum(lb)))


def c4_distributed_leg(args, world, rank, dev, sync):
    """BASELINE configs[3] on N > 1 ranks: getH2 with a communicator -- near-field cluster pairs row-sharded by row cluster
    (clusters.partitionClusterPairs; NA:3247-3260, clusterMethodCy.pyx:1854-1896), far field dealt over the ranks, matvec =
    Bcast(x) + local near/far products + all-reduce of the N-vector (clusterMethodCy.pyx:3127-3154)."""
    import numpy as np
    import torch
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.solvers import cg
    red_dev = dev if os.environ.get('PNL_BENCH_BACKEND', 'nccl') == 'nccl' else torch.device('cpu')
    # phase 1 has no collective (row-sharded near field, dealt far field): a rank may fail here alone, so the ranks AGREE on the
    # outcome before anybody enters a collective -- one failing rank must not leave the others waiting in an all-reduce
    err, h2, walls = None, None, []
    try:
        dm = P1_DoFMap(disc(args.noRef, sectors=args.sectors), PHYSICAL)
        b4 = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True, comm=True)
        for rep in range(2):
            sync(); t0 = time.perf_counter()
            h2 = b4.getH2()
            sync(); walls.append(time.perf_counter()-t0)
    except Exception as e:
        err = repr(e)
    failed = [r for r, v in enumerate(gather_ranks(0. if err is None else 1., world, red_dev)) if v]
    if failed:
        return dict(error=err if err is not None else 'assembly failed on ranks {}'.format(failed), failed_ranks=failed)
    # phase 2 is collective throughout (matvec, CG, the gathers): an exception here ends this rank with a non-zero exit and the
    # launcher ends the job -- never swallowed
    c = h2.info.get('counters', {})
    near_ms = gather_ranks(h2.info.get('interior_ms', 0.), world, red_dev)
    pairs = gather_ranks(c.get('numAssembledCellPairs', 0), world, red_dev)
    nnz = gather_ranks(h2.local.nnz, world, red_dev)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal(dm.num_dofs)).to(dev)
    for _ in range(3):
        y = h2.matvec(x)
    sync(); t0 = time.perf_counter()
    for _ in range(20):
        y = h2.matvec(x)
    sync()
    mv = max(gather_ranks((time.perf_counter()-t0)/20, world, red_dev))
    rhs = torch.from_numpy(np.asarray(dm.assembleRHS(1.0))).to(dev)
    sync(); t0 = time.perf_counter()
    u, its, res = cg(h2, rhs, tol=1e-8, maxiter=2000)
    sync()
    t_cg = max(gather_ranks(time.perf_counter()-t0, world, red_dev))
    return dict(num_dofs=dm.num_dofs, getH2_first_ms=1e3*max(gather_ranks(walls[0], world, red_dev)),
                getH2_ms=1e3*max(gather_ranks(walls[-1], world, red_dev)), near_field_device_ms_per_rank=[round(v, 3) for v in near_ms],
                near_field_element_pairs_per_rank=[int(v) for v in pairs], near_field_element_pairs=int(sum(pairs)),
                near_field_nnz_per_rank=[int(v) for v in nnz], pairs_per_s=sum(pairs)/(1e-3*max(near_ms)) if max(near_ms) > 0 else 0.,
                matvec_ms=1e3*mv, cg_jacobi_iterations=int(its), cg_jacobi_ms=1e3*t_cg, cg_residual=float(res[-1]))


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
    ap.add_argument('--no-extra', action='store_true', help='skip the short legs of the other BASELINE.json configurations (N=1 only)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # bare command: this process becomes the launcher of its own ranks (nothing GPU-related has been imported yet)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node {} (WORLD_SIZE={})'.format(args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the assembly path has no CPU implementation'
    # rehearsal of the N > 1 path on a one-GPU box: PNL_BENCH_BACKEND=gloo puts every rank on cuda:0 and reduces on the host
    backend = os.environ.get('PNL_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit('bench.py --gpus {}: only {} GPU(s) visible (PNL_BENCH_BACKEND=gloo rehearses the N > 1 path on one GPU)'.format(
            world, torch.cuda.device_count()))
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
    if world == 1:
        # rows start on 64-byte lines (leading dimension rounded up to 8 doubles): the fold / mirror passes write whole lines
        ldA = (N+7) & ~7
        A = torch.zeros((N, ldA), dtype=torch.float64, device=dev)[:, :N] if not os.environ.get('PNL_BENCH_NO_LDA_PAD') else torch.zeros((N, N), dtype=torch.float64, device=dev)
    else:
        # row-owned storage (DistributedSlab_LinearOperator): this rank's one-sided slab of its block rows, ~N^2 / (2 P)
        # doubles, and its partial per-cell diagonal blocks; the tiles are dealt by block rows of equal work; no N x N
        # array and no collective in the assembly (the matvec all-reduces an N-vector)
        from pynucleus_amd.builder import row_slab_of_rank, tile_cells
        Tc = tile_cells(builder.dm.dofs_per_element, 2)
        c0, c1, tiles, rows, cols = row_slab_of_rank(builder.dm, Tc, rank, world, ctx.block_row_costs((nc+Tc-1)//Tc))
        A = torch.zeros((max(rows.shape[0], 1), max(cols.shape[0], 1)), dtype=torch.float64, device=dev)
        ctx.set_row_slab(rows, cols)

    # the library says whether assemble_dense adds to A (zero fill needed, part of the step) or overwrites every entry
    overwrites = world == 1 and ctx.dense_overwrites(0, nc)

    def step():
        if not overwrites:
            A.zero_()
        if world == 1:
            ctx.assemble_dense(A.data_ptr(), A.stride(0), True, 0, nc)
        elif rows.shape[0]:
            ctx.assemble_dense_tiles(A.data_ptr(), A.stride(0), True, tiles, c0, c1)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    phase_acc = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter()-t0
    # phases of the last step from HIP events recorded on the stream the kernels ran on
    ms = ctx.phase_ms()
    cnt = ctx.counters()
    # a few extra (untimed) steps to average every tile kernel's duration from its own HIP events (recorded by the library
    # on the stream the kernels run on)
    kacc = {}
    for _ in range(3):
        step()
        torch.cuda.synchronize(dev)
        m = ctx.phase_ms()
        for k, v in m.items():
            phase_acc[k] = phase_acc.get(k, 0.)+v/3.
        for k, v in ctx.kernel_ms().items():
            kacc[k] = kacc.get(k, 0.)+v/3.
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
    # Distant pairs of the orders packed into the tile rule table (<= 16 points) are integrated by the tile kernels: the
    # uniform-tile kernels (tiles whose pairs are all of order 2: k_tile_uniform<3,3>; all of order 3 / 4: k_tile_uniform<3,6>) and the
    # general one (k_tile_distant: every other tile).  The counters tell how many pairs of each order the uniform kernels
    # took; the rest of the histogram belongs to the general kernel.
    T = builder.tables
    tile_orders = {q: c for q, c in cnt['orders'].items() if T.num_points(q) <= 16 and q < 18}
    uni = cnt.get('uniformTilePairsByOrder', {})
    mixed_orders = {q: c-uni.get(q, 0) for q, c in tile_orders.items()}
    kernels = {'k_tile_distant': (algorithmic_flops(mixed_orders, T.num_points), 1e-3*kacc.get('tile_general', 0.)),
               'k_tile_uniform<3,3> (order 2)': (algorithmic_flops({2: uni.get(2, 0)}, T.num_points), 1e-3*kacc.get('tile_uniform2', 0.)),
               'k_tile_uniform<3,6> (order 3)': (algorithmic_flops({3: uni.get(3, 0)}, T.num_points), 1e-3*kacc.get('tile_uniform3', 0.)),
               'k_tile_uniform<3,6> (order 4)': (algorithmic_flops({4: uni.get(4, 0)}, T.num_points), 1e-3*kacc.get('tile_uniform4', 0.))}
    dominant = max(kernels, key=lambda k: kernels[k][1])
    dom_flops, dom_s = kernels[dominant]
    achieved = dom_flops/dom_s/1e12 if dom_s > 0 else 0.
    both_s = sum(v[1] for v in kernels.values())
    all_flops = sum(v[0] for v in kernels.values())
    # HBM traffic of the dominant kernel: PMC counters cannot be read inside this process; tools/profile_round.sh runs this
    # command under rocprofv3 --pmc (separate passes) and tools/collect_profiles.py records the kernel sources it measured.  The number is used only
    # if it belongs to the library and the workload of this run, never a stale constant.
    traffic, traffic_note = None, 'no PMC record (tools/profile_round.sh + tools/collect_profiles.py) for these kernel sources and this workload'
    pmc_fn = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc_fn) and world == 1:
        from pynucleus_amd import _lib as _l
        src_sha = _l.source_sha16()
        with open(pmc_fn) as f:
            rec = json.load(f)
        key = 'noRef{}{}'.format(args.noRef, '' if args.sectors == 6 else '_s{}'.format(args.sectors))
        if key in rec and rec[key].get('src_sha16') == src_sha:
            # record keys: k_tile_distant, k_tile_uniform_3_3 (order 2), k_tile_uniform_3_6 (orders 3 and 4), k_fold_mirror
            tkey = {'k_tile_distant': 'k_tile_distant', 'k_tile_uniform<3,3> (order 2)': 'k_tile_uniform_3_3',
                    'k_tile_uniform<3,6> (order 3)': 'k_tile_uniform_3_6', 'k_tile_uniform<3,6> (order 4)': 'k_tile_uniform_3_6'}[dominant]
            traffic = rec[key].get(tkey+'_hbm_bytes_per_launch')
            traffic_note = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command ({}), FETCH_SIZE doubled per MI355X_MICROARCH.md'.format(rec[key].get('tag'))
    # HBM view of the same launches: the algorithmic minimum is one write of the upper block triangle they fill
    hbm_alg_bytes = 8.*N*N/2
    # the hardware view next to the SURVEY 8(d) units: VALU issue utilisation = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel
    # cycles at 2.4 GHz) from the committed PMC record of these sources (the factorised evaluator executes about a third of the
    # reference's flop count, so `frac` can exceed what the issue slots say); whole_step_frac = all algorithmic flops of the step
    # (every kernel evaluation and pair) over the whole step time
    valu_util, valu_note = None, traffic_note
    if traffic is not None:
        insts = rec[key].get(tkey+'_sq_insts_valu')
        if insts and dom_s > 0:
            valu_util = insts*4./(1024.*dom_s*2.4e9)
            valu_note = 'SQ_INSTS_VALU per launch x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz), rocprofv3 --pmc pass ({})'.format(rec[key].get('tag'))
    step_s = elapsed/args.steps
    whole_flops = flops_from_counters(cnt, 3) if world == 1 else None
    roofline = dict(bound='fp64_valu', kernel=dominant, achieved=achieved, peak=FP64_VECTOR_PEAK_TFLOPS, unit='TFLOP/s',
                    frac=achieved/FP64_VECTOR_PEAK_TFLOPS,
                    frac_note='achieved counts the flop of the reference\'s evaluation per pair (SURVEY 8(d): 759 for an order-2 P1 pair); the kernel '
                              'forms the same local matrix from row / column sums of the kernel values in about 200 VALU instructions per pair, so frac '
                              'can exceed 1; valu_issue_util is the executed-instruction view (at the nominal 2.4 GHz; under fp64 load the clock '
                              'sustains about 1.7 GHz, tools/probes/valu_rate_probe.hip: 5.67 nominal cycles per v_fma_f64 of a wave)',
                    traffic=traffic, traffic_note=traffic_note, algorithmic_flops_per_launch=dom_flops,
                    valu_issue_util=valu_util, valu_issue_note=valu_note,
                    whole_step_frac=(whole_flops/step_s/1e12/FP64_VECTOR_PEAK_TFLOPS) if whole_flops else None,
                    whole_step_algorithmic_flops=whole_flops,
                    kernel_ms=1e3*dom_s,
                    tile_kernels={k: dict(algorithmic_flops_per_launch=v[0], kernel_ms=1e3*v[1],
                                          achieved=v[0]/v[1]/1e12 if v[1] > 0 else 0.) for k, v in kernels.items()},
                    tile_phase_achieved=all_flops/both_s/1e12 if both_s > 0 else 0.,
                    hbm_algorithmic_GBs=hbm_alg_bytes/both_s/1e9 if both_s > 0 else 0., hbm_peak_GBs=HBM_PEAK_GBS)

    out = dict(metric='element-pairs/sec assembled (2D P1 fractional s=0.5, dense) + % fp64 roofline', value=value,
               unit='element-pairs/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3*elapsed/args.steps,
               higher_is_better=True, scaling='strong', vs_baseline=None, dtype='f64', data='synthetic',
               config=dict(workload='2D unit disc ({}uniform_disc refined {}x, {} cells), P1, {} DoFs, fractional s={}, horizon=inf, '
                           'dense getDense incl. zeroExterior; {} element pairs/step'.format('' if args.sectors == 6 else '{}-sector '.format(args.sectors), args.noRef, nc, N, args.s, int(pairs_total)),
                           noRef=args.noRef, num_dofs=N, num_cells=nc, parallelism='block rows of the upper triangle dealt over {} GPU(s), row-owned one-sided slabs'.format(world)),
               roofline=roofline,
               phases_ms={k: round(v, 4) for k, v in phase_acc.items()},
               kernel_evaluations_per_step=cnt['numIntegrations'] if world == 1 else None)

    # ---- the assembled operator at work (north star: "RCCL all-reduce over xGMI for the assembled matvec"): y = A x and a
    # Jacobi-CG solve of A u = f, f = 1.  N = 1: k_gemv on the N x N block; N > 1: DistributedSlab_LinearOperator -- local slab
    # product, ONE all-reduce of the N-vector (DistributedH2Matrix_globalData.matvec, clusterMethodCy.pyx:3127-3154) ----------
    out['operator'] = operator_leg(args, world, rank, dev, red_dev, backend, builder, ctx, dm, A, None if world == 1 else (rows, cols),
                                   phase_acc.get('total', 0.), sync)
    if world > 1 and not args.no_extra:
        out.setdefault('configs', {})['C4_disc_noRef{}_s0.75_H2_row_sharded'.format(args.noRef)] = c4_distributed_leg(args, world, rank, dev, sync)

    # ---- CPU baseline: the C oracle (reference loop order) on bounded samples of the same workload -------------------
    if rank == 0 and world == 1 and not args.no_cpu:
        out['cpu_baseline'] = cpu_baseline(args, T, N, nc, value)
    # ---- the other BASELINE.json configurations, each a short driver-timed leg (N = 1 only) ---------------------------
    if rank == 0 and world == 1 and not args.no_extra:
        del A
        builder = ctx = None
        torch.cuda.empty_cache()
        out['configs'] = extra_configs(dev)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

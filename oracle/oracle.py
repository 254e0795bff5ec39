"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The reference cannot be built or imported in this environment (SURVEY.md
section 8c), so there is no oracle/_ref: the oracle is the C restatement in
nl_oracle.c, fed with the oracle's OWN host tables (oracle/tables.py: kernel constants,
order formulas, near rules and PSI tables, boundary twins, class tables -- restated from the
reference text independently of pynucleus_amd/).  A caller may hand over the product's
``nonlocalTables``: only the user's inputs are read off it (mesh, DoF numbering, kernel type /
order / horizon / normalisation, params) plus the one shared, unpinned input, the triangle rules
of distant pairs.  Kernels with an order per quadrature point (the ``pointwise`` tables of the
non-symmetric path) are the exception: their rule tables are still the product's.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NLO_MAX_ORDER = 120
NLO_NUM_COUNTERS = 8+NLO_MAX_ORDER+3
IGNORED = -6


class nlo_order_formula(C.Structure):
    _fields_ = [('c0', C.c_double), ('a', C.c_double), ('b', C.c_double), ('e', C.c_double), ('den0', C.c_double),
                ('clip_num', C.c_int32), ('pad', C.c_int32)]


class nlo_kernel(C.Structure):
    _fields_ = [('ktype', C.c_int32), ('interaction', C.c_int32), ('exponent', C.c_double), ('scale', C.c_double),
                ('horizon2', C.c_double)]


_P = C.c_void_p


class nlo_problem(C.Structure):
    _fields_ = [('dim', C.c_int32), ('dpe', C.c_int32), ('nc', C.c_int32), ('nv', C.c_int32), ('num_dofs', C.c_int32),
                ('dofs_per_vertex', C.c_int32), ('dofs_per_edge', C.c_int32), ('pad0', C.c_int32),
                ('vertices', _P), ('cells', _P), ('dofs', _P), ('vol', _P), ('h', _P),
                ('H0', C.c_double), ('dof_perm_table', _P),
                ('kernel', nlo_kernel), ('qo', nlo_order_formula),
                ('qmax', C.c_int32), ('pad1', C.c_int32),
                ('dist_off', _P), ('dist_bary', _P), ('dist_w', _P), ('dist_phi', _P),
                ('sing_M', C.c_int32*3), ('sing_rows', C.c_int32*3),
                ('sing_nodes', _P*3), ('sing_w', _P*3), ('sing_psi', _P*3), ('sing_fac', C.c_double),
                ('nb', C.c_int32), ('pad2', C.c_int32), ('bcells', _P),
                ('bkernel', nlo_kernel), ('bqo', nlo_order_formula),
                ('bfacet_off', _P), ('bfacet_bary', _P), ('bfacet_w', _P),
                ('bsing_M', C.c_int32*2), ('bpad', C.c_int32*2),
                ('bsing_nodes', _P*2), ('bsing_w', _P*2), ('bsing_phi', _P*2), ('bsing_fac', C.c_double),
                ('nclasses', C.c_int32), ('num_labels', C.c_int32), ('classes', _P), ('cell_labels', _P), ('facet_labels', _P),
                ('cls_of', _P),
                ('pw_type', C.c_int32), ('pw_normalized', C.c_int32), ('pw_p', C.c_double*6),
                ('pw_c0', C.c_double), ('pw_bc0', C.c_double), ('pw_nkeys', C.c_int32), ('pw_nbkeys', C.c_int32),
                ('pw_keys', _P), ('pw_bkeys', _P),
                ('pw_nodes', _P*3), ('pw_w', _P*3), ('pw_phi0', _P*3), ('pw_phi1', _P*3),
                ('pw_bnodes', _P*2), ('pw_bw', _P*2), ('pw_bphi', _P*2), ('xform', C.c_double*4), ('has_xform', C.c_int32), ('pad3', C.c_int32),
                ('pw_vertex_s', _P)]


def build():
    """compile libnl_oracle.so next to its source (gcc only)"""
    subprocess.check_call(['make', '-s', '-C', _HERE, 'libnl_oracle.so'])


def lib():
    global _LIB
    if _LIB is None:
        fn = os.path.join(_HERE, 'libnl_oracle.so')
        if not os.path.exists(fn) or os.path.getmtime(fn) < os.path.getmtime(os.path.join(_HERE, 'nl_oracle.c')):
            build()
        L = C.CDLL(fn)
        ip = C.POINTER(C.c_int)
        L.nlo_panel.restype = C.c_int
        L.nlo_panel.argtypes = [C.POINTER(nlo_problem), C.c_int, C.c_int, ip, ip, ip]
        L.nlo_panel_boundary.restype = C.c_int
        L.nlo_panel_boundary.argtypes = [C.POINTER(nlo_problem), C.c_int, C.c_int, ip, ip, ip]
        L.nlo_eval.restype = None
        L.nlo_eval.argtypes = [C.POINTER(nlo_problem), C.c_int, C.c_int, C.c_int, ip, ip, ip, _P, _P]
        L.nlo_eval_boundary.restype = None
        L.nlo_eval_boundary.argtypes = [C.POINTER(nlo_problem), C.c_int, C.c_int, C.c_int, ip, ip, ip, _P, _P]
        L.nlo_get_dense_rows.restype = C.c_int
        L.nlo_get_dense_rows.argtypes = [C.POINTER(nlo_problem), _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]
        L.nlo_get_dense_nonsym.restype = C.c_int
        L.nlo_get_dense_nonsym.argtypes = [C.POINTER(nlo_problem), _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]
        L.nlo_pw_svalue.restype = C.c_double
        L.nlo_pw_svalue.argtypes = [C.POINTER(nlo_problem), C.c_int, C.c_int]
        L.nlo_assemble_pairs_masked.restype = C.c_int
        L.nlo_assemble_pairs_masked.argtypes = [C.POINTER(nlo_problem), C.c_int, _P, _P, _P, _P, _P, _P, _P]
        L.nlo_assemble_pairs_masked_nonsym.restype = C.c_int
        L.nlo_assemble_pairs_masked_nonsym.argtypes = [C.POINTER(nlo_problem), C.c_int, _P, _P, _P, _P, _P, _P]
        L.nlo_assemble_boundary_masked.restype = C.c_int
        L.nlo_assemble_boundary_masked.argtypes = [C.POINTER(nlo_problem), C.c_int, _P, _P, _P, C.c_double, _P, _P, _P, _P]
        _LIB = L
    return _LIB


def _kern(k):
    p = k.device_params()
    ktype = p['ktype']
    if ktype == 5 and p.get('dim', 1) == 2:
        ktype = 7                                          # Gaussian boundary kernel in 2D (nl_oracle.c kernel_eval: by dimension)
    return nlo_kernel(ktype, p.get('interaction', 0), p['exponent'], p['scale'], p['horizon2'])


def _qo(f):
    return nlo_order_formula(f.c0, f.a, f.b, f.e, f.den0, int(f.clip_num), 0)


def kernel_spec(kernel):
    """the user's inputs of a kernel object (duck-typed: nothing is imported from the product), nothing derived"""
    finite = bool(np.isfinite(kernel.horizonValue))
    spec = dict(kernelType=int(kernel.kernelType), horizon=float(kernel.horizonValue), normalized=bool(getattr(kernel, 'normalized', True)),
                interaction=int(getattr(kernel.interaction, 'device_id', 0)) if finite else 0)
    ia = kernel.interaction
    if all(hasattr(ia, n) for n in ('a', 'b', 'theta')):     # ellipse domains: the user's semi-axes and angle
        spec['ellipse'] = (float(ia.a), float(ia.b), float(ia.theta))
    for name in ('variance', 'exponentialRate'):          # Gaussian / exponential kernels: the user's parameter
        if hasattr(kernel, name):
            spec[name] = float(getattr(kernel, name))
    s = getattr(kernel, 's', None)
    if s is None:
        return spec
    if callable(getattr(s, 'spec', None)):
        spec['s'] = s.spec()                           # the constructor's inputs of a piecewise-constant order
    elif hasattr(s, 'sVals'):
        sv = np.array(s.sVals, dtype=np.float64)
        if hasattr(s, 'layerBoundaries'):
            spec['s'] = ('layers', np.array(s.layerBoundaries, dtype=np.float64), sv)
        elif hasattr(s, 'interface'):
            spec['s'] = ('leftRight', sv[0, 0], sv[1, 1], sv[0, 1], sv[1, 0], float(s.interface))
        elif sv.shape == (1, 1):
            spec['s'] = ('varconst', float(sv[0, 0]))
        else:
            raise NotImplementedError('oracle tables for the order {}'.format(type(s).__name__))
    else:
        spec['s'] = float(getattr(s, 'value', s))
    return spec


def own_tables(T):
    """the oracle's own tables for the problem a product ``nonlocalTables`` describes (module docstring)"""
    from .tables import OracleTables
    if isinstance(T, OracleTables) or getattr(T, 'pointwise', False):
        return T
    finite = bool(np.isfinite(T.kernel.horizonValue))
    return OracleTables(T.dm, kernel_spec(T.kernel), getattr(T, 'params', {}), (not finite) and bool(T.zeroExterior),
                        (T.dist_off, T.dist_bary, T.dist_w))


class OracleProblem:
    """Holds the numpy arrays alive and exposes the C entry points."""

    def __init__(self, tables, own=True):
        self.given_tables = tables
        T = self.tables = own_tables(tables) if own else tables
        dm, mesh = T.dm, T.dm.mesh
        self._keep = []
        if getattr(T, 'classes', None):
            # variable order: one full problem description per class, the top-level one carries the labels
            self._class_problems = [OracleProblem(c, own=False) for c in T.classes]
            arr = (nlo_problem*len(self._class_problems))(*[op.P for op in self._class_problems])
            self._keep.append(arr)
            base = self._class_problems[0]
            P = self.P = nlo_problem.from_buffer_copy(base.P)
            self._keep.append(base)
            P.nclasses, P.num_labels = len(self._class_problems), T.num_labels
            P.classes = C.cast(arr, C.c_void_p)
            for name, a in (('cell_labels', T.cell_labels), ('facet_labels', T.facet_labels), ('cls_of', T.cls_of)):
                a = np.ascontiguousarray(a, dtype=np.int32)
                self._keep.append(a)
                setattr(P, name, a.ctypes.data)
            self.E = base.E
            self.nV = base.nV
            return

        def ptr(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            self._keep.append(a)
            return a.ctypes.data

        P = self.P = nlo_problem()
        P.dim, P.dpe, P.nc, P.nv, P.num_dofs = T.dim, T.dpe, mesh.num_cells, mesh.num_vertices, dm.num_dofs
        P.dofs_per_vertex, P.dofs_per_edge = dm.dofs_per_vertex, dm.dofs_per_edge
        P.vertices = ptr(mesh.vertices, np.float64)
        P.cells = ptr(mesh.cells, np.int32)
        P.dofs = ptr(dm.dofs, np.int32)
        P.vol = ptr(mesh.volVector, np.float64)
        P.h = ptr(mesh.hVector, np.float64)
        P.H0 = T.H0
        xf = getattr(T, 'interaction_transform', None)
        if xf is None:
            xf = getattr(getattr(getattr(T, 'kernel', None), 'interaction', None), 'transform', None)
        if xf is not None:
            P.has_xform = 1
            for i, v in enumerate(np.asarray(xf, dtype=np.float64).ravel()):
                P.xform[i] = v
        P.dof_perm_table = ptr(T.dof_perm_table, np.int32)
        self.pointwise = bool(getattr(T, 'pointwise', False))
        if not self.pointwise:
            P.kernel = _kern(T.kernel)
            P.qo = _qo(T.qo)
        P.qmax = T.qcap
        P.dist_off = ptr(T.dist_off, np.int32)
        P.dist_bary = ptr(T.dist_bary, np.float64)
        P.dist_w = ptr(T.dist_w, np.float64)
        P.dist_phi = ptr(T.dist_phi, np.float64)
        if self.pointwise:
            # non-symmetric kernel with an order per quadrature point: order function, near rules per distinct pair order
            R = T.pw_rules()
            P.pw_type, P.pw_normalized = T.order_type, int(T.kernel.normalized)
            for i, v in enumerate(T.order_params):
                P.pw_p[i] = v
            P.pw_c0, P.pw_bc0 = T.pw_c0, T.pw_bc0
            if int(T.order_type) == 5:
                P.pw_vertex_s = ptr(T.order_vertex_values, np.float64)
            P.pw_nkeys, P.pw_nbkeys = len(R['keys']), len(R['bkeys'])
            P.pw_keys, P.pw_bkeys = ptr(R['keys'], np.float64), ptr(R['bkeys'], np.float64)
            for slot, (nodes, w, phi0, phi1) in R['rules'].items():
                P.sing_M[slot], P.sing_rows[slot] = w.shape[1], phi0.shape[1]
                P.pw_nodes[slot], P.pw_w[slot] = ptr(nodes, np.float64), ptr(w, np.float64)
                P.pw_phi0[slot], P.pw_phi1[slot] = ptr(phi0, np.float64), ptr(phi1, np.float64)
            for slot, (nodes, w, phi) in R['brules'].items():
                P.bsing_M[slot] = w.shape[1]
                P.pw_bnodes[slot], P.pw_bw[slot], P.pw_bphi[slot] = ptr(nodes, np.float64), ptr(w, np.float64), ptr(phi, np.float64)
            P.sing_fac, P.bsing_fac = T.sing_fac, T.bsing_fac
            P.bfacet_off = ptr(T.bfacet_off, np.int32)
            P.bfacet_bary = ptr(T.bfacet_bary, np.float64)
            P.bfacet_w = ptr(T.bfacet_w, np.float64)
            P.nb = T.bcells.shape[0]
            P.bcells = ptr(T.bcells, np.int32)
            self.E = (2*T.dpe)**2
            self.nV = T.dim+1
            return
        for panel, r in T.singular.items():
            slot = -panel-1
            P.sing_M[slot], P.sing_rows[slot] = r.num_nodes, r.rows
            P.sing_nodes[slot] = ptr(r.nodes, np.float64)
            P.sing_w[slot] = ptr(r.weights, np.float64)
            P.sing_psi[slot] = ptr(r.psi, np.float64)
        P.sing_fac = T.sing_fac
        P.bfacet_off = ptr(T.bfacet_off, np.int32)
        P.bfacet_bary = ptr(T.bfacet_bary, np.float64)
        P.bfacet_w = ptr(T.bfacet_w, np.float64)
        if T.has_boundary_tables:
            P.nb = T.bcells.shape[0]
            P.bcells = ptr(T.bcells, np.int32)
            P.bkernel = _kern(T.boundaryKernel)
            P.bqo = _qo(T.bqo)
            for panel, r in T.bsingular.items():
                slot = -panel-1
                P.bsing_M[slot] = r.num_nodes
                P.bsing_nodes[slot] = ptr(r.nodes, np.float64)
                P.bsing_w[slot] = ptr(r.weights, np.float64)
                P.bsing_phi[slot] = ptr(r.psi, np.float64)
            P.bsing_fac = T.bsing_fac
        self.E = (2*T.dpe)*(2*T.dpe+1)//2
        self.nV = T.dim+1

    # -- per-pair entry points -------------------------------------------
    def panel(self, c1, c2):
        p1, p2, p = (C.c_int*3)(), (C.c_int*3)(), (C.c_int*12)()
        panel = lib().nlo_panel(C.byref(self.P), c1, c2, p1, p2, p)
        return panel, list(p1)[:self.nV], list(p2)[:self.nV], list(p)[:2*self.tables.dpe]

    def eval(self, c1, c2):
        """(panel, contrib[E]) of the cell pair; contrib is None for IGNORED"""
        p1, p2, p = (C.c_int*3)(), (C.c_int*3)(), (C.c_int*12)()
        panel = lib().nlo_panel(C.byref(self.P), c1, c2, p1, p2, p)
        if panel == IGNORED:
            return panel, None
        contrib = np.zeros(self.E)
        ne = np.zeros(1, dtype=np.int64)
        lib().nlo_eval(C.byref(self.P), c1, c2, panel, p1, p2, p, contrib.ctypes.data, ne.ctypes.data)
        return panel, contrib

    def eval_boundary(self, c1, b):
        p1, p2, p = (C.c_int*3)(), (C.c_int*3)(), (C.c_int*12)()
        panel = lib().nlo_panel_boundary(C.byref(self.P), c1, b, p1, p2, p)
        contrib = np.zeros(self.tables.dpe*(self.tables.dpe+1)//2)
        ne = np.zeros(1, dtype=np.int64)
        lib().nlo_eval_boundary(C.byref(self.P), c1, b, panel, p1, p2, p, contrib.ctypes.data, ne.ctypes.data)
        return panel, contrib

    # -- whole-matrix entry point -----------------------------------------
    def get_dense(self, cell_start=0, cell_end=None, store=True):
        """returns (A or None, counters dict, seconds (interior, zeroExterior))"""
        T = self.tables
        nc = T.dm.mesh.num_cells
        cell_end = nc if cell_end is None else cell_end
        N = T.dm.num_dofs
        A = np.zeros((N, N)) if store else None
        counters = np.zeros(NLO_NUM_COUNTERS, dtype=np.int64)
        seconds = np.zeros(2)
        fun = lib().nlo_get_dense_nonsym if (getattr(self, 'pointwise', False) or getattr(T, 'nonsym', False)) else lib().nlo_get_dense_rows
        rc = fun(C.byref(self.P), A.ctypes.data if store else None, int(T.zeroExterior),
                 cell_start, cell_end, counters.ctypes.data, seconds.ctypes.data, int(store))
        if rc != 0:
            raise RuntimeError('oracle failed with code {}'.format(rc))
        hist = {q: int(counters[8+q]) for q in range(NLO_MAX_ORDER) if counters[8+q]}
        sing = {-1-k: int(counters[8+NLO_MAX_ORDER+k]) for k in range(3)}
        cnt = dict(numCellPairs=int(counters[0]), numAssembledCellPairs=int(counters[1]), numIntegrations=int(counters[2]),
                   numBoundaryPairs=int(counters[3]), numBoundaryIntegrations=int(counters[4]), orders=hist, singular=sing)
        return A, cnt, (float(seconds[0]), float(seconds[1]))

    # -- H2 near field ------------------------------------------------------------------------------------------
    def assemble_clusters(self, pairs, masks, bcells, bfacets, bmasks, indptr, indices, symmetric=True, global_boundary=None):
        """masked interior pairs + cluster-local boundary items into CSR / SSS; returns (data, diagonal or None, counters).
        global_boundary = (cells, facets, masks, fac) adds the global Omega x Omega^c term with the given sign."""
        N = self.tables.dm.num_dofs
        data = np.zeros(indices.shape[0])
        diag = np.zeros(N) if symmetric else None
        pairs = np.ascontiguousarray(pairs, dtype=np.int32)
        masks = np.ascontiguousarray(masks, dtype=np.uint64)
        indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        cnt = np.zeros(3, dtype=np.int64)
        dptr = diag.ctypes.data if symmetric else None
        rc = lib().nlo_assemble_pairs_masked(C.byref(self.P), pairs.shape[0], pairs.ctypes.data, masks.ctypes.data, indptr.ctypes.data,
                                             indices.ctypes.data, data.ctypes.data, dptr, cnt.ctypes.data)
        if rc:
            raise RuntimeError('oracle failed with code {}'.format(rc))
        items = [(bcells, bfacets, bmasks, 1.)]
        if global_boundary is not None:
            items.append(global_boundary)
        for cc, ff, mm, fac in items:
            cc = np.ascontiguousarray(cc, dtype=np.int32)
            ff = np.ascontiguousarray(ff, dtype=np.int32)
            mm = np.ascontiguousarray(mm, dtype=np.uint32)
            if cc.shape[0] == 0:
                continue
            rc = lib().nlo_assemble_boundary_masked(C.byref(self.P), cc.shape[0], cc.ctypes.data, ff.ctypes.data, mm.ctypes.data,
                                                    float(fac), indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, dptr)
            if rc:
                raise RuntimeError('oracle failed with code {}'.format(rc))
        return data, diag, dict(numCellPairs=int(cnt[0]), numAssembledCellPairs=int(cnt[1]), numIntegrations=int(cnt[2]))


    def assemble_clusters_nonsym(self, pairs, masks, bcells, bfacets, bmasks, indptr, indices, global_boundary=None):
        """non-symmetric kernels (order per quadrature point, or a non-symmetric piecewise-constant table): ORDERED pairs with masks
        over the (2 dpe)^2 local entries (nlo_assemble_pairs_masked_nonsym) + cluster-local boundary items with the kernel's
        boundary twin into unsymmetric CSR; returns (data, counters)"""
        data = np.zeros(indices.shape[0])
        pairs = np.ascontiguousarray(pairs, dtype=np.int32)
        masks = np.ascontiguousarray(masks, dtype=np.uint64)
        indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        cnt = np.zeros(NLO_NUM_COUNTERS, dtype=np.int64)
        rc = lib().nlo_assemble_pairs_masked_nonsym(C.byref(self.P), pairs.shape[0], pairs.ctypes.data, masks.ctypes.data,
                                                    indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, cnt.ctypes.data)
        if rc:
            raise RuntimeError('oracle failed with code {}'.format(rc))
        items = [(bcells, bfacets, bmasks, 1.)]
        if global_boundary is not None:
            items.append(global_boundary)
        for cc, ff, mm, fac in items:
            cc = np.ascontiguousarray(cc, dtype=np.int32)
            ff = np.ascontiguousarray(ff, dtype=np.int32)
            mm = np.ascontiguousarray(mm, dtype=np.uint32)
            if cc.shape[0] == 0:
                continue
            rc = lib().nlo_assemble_boundary_masked(C.byref(self.P), cc.shape[0], cc.ctypes.data, ff.ctypes.data, mm.ctypes.data,
                                                    float(fac), indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, None)
            if rc:
                raise RuntimeError('oracle failed with code {}'.format(rc))
        hist = {q: int(cnt[8+q]) for q in range(NLO_MAX_ORDER) if cnt[8+q]}
        sing = {-1-k: int(cnt[8+NLO_MAX_ORDER+k]) for k in range(3)}
        return data, dict(numCellPairs=int(cnt[0]), numAssembledCellPairs=int(cnt[1]), numIntegrations=int(cnt[2]), orders=hist, singular=sing)


# -- variable order: kernel blocks, interfaces and the boundary items of assembleClusters -----------------------------------
def kernel_blocks_and_jumps(dm, T):
    """getKernelBlocksAndJumps (NA:2312-2384) in plain loops.  blocks: order value -> set of DoFs whose cells all carry that
    order, key None for DoFs on an interface (INTERFACE_DOF); jumps: (cell1, cell2) sorted -> the shared vertices in the order
    they have in the cell visited last (the reference overwrites the entry when it meets the pair from the other side)."""
    mesh = dm.mesh
    sFun = T.kernel.s
    centers = mesh.vertices[mesh.cells].mean(axis=1)
    lab = sFun.labels(centers)
    orders = [float(sFun.sVals[l, l]) for l in lab]                    # P0 interpolation of s.diagonal() (NA:2329)
    dofOrders = {}
    for c in range(mesh.num_cells):
        for d in dm.dofs[c]:
            if d < 0:
                continue
            if d not in dofOrders:
                dofOrders[d] = orders[c]
            elif dofOrders[d] is not None and dofOrders[d] != orders[c]:
                dofOrders[d] = None
    blocks = {}
    for d, o in dofOrders.items():
        blocks.setdefault(o, set()).add(int(d))
    jumps = {}
    facet_of = {}
    nV = mesh.cells.shape[1]
    for c in range(mesh.num_cells):
        vs = [int(v) for v in mesh.cells[c]]
        for k in range(nV):
            f = tuple(sorted(vs[:k]+vs[k+1:]))
            facet_of.setdefault(f, []).append(c)
    conn = [[] for _ in range(mesh.num_cells)]                      # mesh.getCellConnectivity(mesh.dim): cells across a facet
    for f, cs in facet_of.items():
        if len(cs) == 2:
            conn[cs[0]].append(cs[1])
            conn[cs[1]].append(cs[0])
    for c1 in range(mesh.num_cells):
        for c2 in conn[c1]:
            if orders[c1] != orders[c2]:
                shared = [int(v) for v in mesh.cells[c1] if v in mesh.cells[c2]]
                jumps[(min(c1, c2), max(c1, c2))] = shared
    return blocks, jumps


def variable_items_reference(dm, Pnear, T, zeroExterior=True, evalShift=1e-9):
    """The boundary integrals of assembleClusters for a piecewise-constant order, as the reference walks them (NA:1966-2156):
    the kernel parameters of every (cell, facet) integral are evaluated at the cell centre and at the facet centre moved by
    evalShift along (dy, -dx) (NA:2034-2042, 2065-2071, 2095-2101, 2137-2147; in 1D away from the cell / to the right and
    then to the left of the vertex).  Returns a list of items (class, fac, cell, facet vertices, mask) with
    class = cls_of[label(centre), label(shifted facet centre)]; tests compare it with the product's vectorised list."""
    mesh = dm.mesh
    dim = mesh.dim
    sFun = T.kernel.s
    dpe = dm.dofs_per_element
    centers = mesh.vertices[mesh.cells].mean(axis=1)

    def cls(c, y):
        return int(T.cls_of[sFun.labels(centers[c][None, :])[0], sFun.labels(np.asarray(y)[None, :])[0]])

    def shifted(facet, sign):
        X = mesh.vertices[list(facet)]
        cen = X.mean(axis=0)
        if dim == 1:
            return cen+sign*evalShift
        return cen+sign*evalShift*np.array([X[1, 1]-X[0, 1], X[0, 0]-X[1, 0]])

    def surface(cellIds):
        # boundaryEdges / boundaryVertices (nonlocalAssembly.pyx:506-578): facets of exactly one cell, oriented as in it
        seen = {}
        for c in cellIds:
            vs = [int(v) for v in mesh.cells[c]]
            fs = [(vs[0], vs[1]), (vs[1], vs[2]), (vs[2], vs[0])] if dim == 2 else [(vs[0],), (vs[1],)]
            for f in fs:
                key = tuple(sorted(f))
                if key in seen:
                    del seen[key]
                else:
                    seen[key] = f
        return list(seen.values())

    _, jumps = kernel_blocks_and_jumps(dm, T)
    items = []
    for cp in Pnear:
        ci = [int(c) for c in cp.cellsInter]
        if not ci:
            continue
        U = set(int(c) for c in cp.cellsUnion)
        d1, d2 = set(int(d) for d in cp.n1.dofs), set(int(d) for d in cp.n2.dofs)

        def mask_of(c):                                         # getElemSymMaskCluster NA:463-478
            m, k = 0, 0
            for p in range(dpe):
                for q in range(p, dpe):
                    if int(dm.dofs[c, p]) in d1 and int(dm.dofs[c, q]) in d2:
                        m |= 1 << k
                    k += 1
            return m

        surf = surface(sorted(U))
        for c in ci:
            m = mask_of(c)
            if m == 0:
                continue
            for f in surf:
                if dim == 1:
                    y = shifted(f, 1. if centers[c, 0] < mesh.vertices[f[0], 0] else -1.)
                else:
                    y = shifted(f, 1.)
                items.append((cls(c, y), 1., c, f, m))
        for (ca, cb), f in jumps.items():
            if ca in U or cb in U:
                continue
            f = tuple(f)
            for side, base in ((1., 1.), (-1., -1.)):          # right of the facet with +1, then left with -1 (NA:2065-2124)
                y = shifted(f, side)
                for c in ci:
                    m = mask_of(c)
                    if m == 0:
                        continue
                    fac = base
                    if dim == 1 and not centers[c, 0] < y[0]:
                        fac = -base
                    items.append((cls(c, y), fac, c, f, m))
    if not zeroExterior:
        full = 0
        for c in range(mesh.num_cells):
            m, k = 0, 0
            for p in range(dpe):
                for q in range(p, dpe):
                    if dm.dofs[c, p] >= 0 and dm.dofs[c, q] >= 0:
                        m |= 1 << k
                    k += 1
            if m == 0:
                continue
            for b in range(T.bcells.shape[0]):
                f = tuple(int(v) for v in np.atleast_1d(T.bcells[b]))
                if dim == 1:
                    y = shifted(f, 1. if centers[c, 0] < mesh.vertices[f[0], 0] else -1.)
                else:
                    y = shifted(f, 1.)
                items.append((cls(c, y), -1., c, f, m))
    return items


def assemble_clusters_variable(op, pairs, masks, groups, indptr, indices, symmetric=True):
    """variable order: masked interior pairs with the class of every pair (nlo_assemble_pairs_masked) plus the boundary items
    grouped as (class, fac, cells, facets, masks), each integrated with the tables of its class (NA:1966-2156)"""
    N = op.tables.dm.num_dofs
    data = np.zeros(indices.shape[0])
    diag = np.zeros(N) if symmetric else None
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    masks = np.ascontiguousarray(masks, dtype=np.uint64)
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    cnt = np.zeros(3, dtype=np.int64)
    dptr = diag.ctypes.data if symmetric else None
    rc = lib().nlo_assemble_pairs_masked(C.byref(op.P), pairs.shape[0], pairs.ctypes.data, masks.ctypes.data, indptr.ctypes.data,
                                         indices.ctypes.data, data.ctypes.data, dptr, cnt.ctypes.data)
    if rc:
        raise RuntimeError('oracle failed with code {}'.format(rc))
    for k, fac, cc, ff, mm in groups:
        cc = np.ascontiguousarray(cc, dtype=np.int32)
        ff = np.ascontiguousarray(ff, dtype=np.int32)
        mm = np.ascontiguousarray(mm, dtype=np.uint32)
        if cc.shape[0] == 0:
            continue
        rc = lib().nlo_assemble_boundary_masked(C.byref(op._class_problems[k].P), cc.shape[0], cc.ctypes.data, ff.ctypes.data,
                                                mm.ctypes.data, float(fac), indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, dptr)
        if rc:
            raise RuntimeError('oracle failed with code {}'.format(rc))
    return data, diag, dict(numCellPairs=int(cnt[0]), numAssembledCellPairs=int(cnt[1]), numIntegrations=int(cnt[2]))


def assemble_clusters_variable_nonsym(op, pairs, masks, groups, indptr, indices):
    """piecewise-constant order with a NON-symmetric table: ORDERED element pairs with masks over the (2 dpe)^2 local entries, each
    evaluated with the class of its orientation (nlo_assemble_pairs_masked_nonsym), plus the boundary items grouped as (class, fac,
    cells, facets, masks) like assemble_clusters_variable; unsymmetric CSR.  Returns (data, counters)."""
    data = np.zeros(indices.shape[0])
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    masks = np.ascontiguousarray(masks, dtype=np.uint64)
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    cnt = np.zeros(NLO_NUM_COUNTERS, dtype=np.int64)
    rc = lib().nlo_assemble_pairs_masked_nonsym(C.byref(op.P), pairs.shape[0], pairs.ctypes.data, masks.ctypes.data, indptr.ctypes.data,
                                                indices.ctypes.data, data.ctypes.data, cnt.ctypes.data)
    if rc:
        raise RuntimeError('oracle failed with code {}'.format(rc))
    for k, fac, cc, ff, mm in groups:
        cc = np.ascontiguousarray(cc, dtype=np.int32)
        ff = np.ascontiguousarray(ff, dtype=np.int32)
        mm = np.ascontiguousarray(mm, dtype=np.uint32)
        if cc.shape[0] == 0:
            continue
        rc = lib().nlo_assemble_boundary_masked(C.byref(op._class_problems[k].P), cc.shape[0], cc.ctypes.data, ff.ctypes.data,
                                                mm.ctypes.data, float(fac), indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, None)
        if rc:
            raise RuntimeError('oracle failed with code {}'.format(rc))
    hist = {q: int(cnt[8+q]) for q in range(NLO_MAX_ORDER) if cnt[8+q]}
    sing = {-1-k: int(cnt[8+NLO_MAX_ORDER+k]) for k in range(3)}
    return data, dict(numCellPairs=int(cnt[0]), numAssembledCellPairs=int(cnt[1]), numIntegrations=int(cnt[2]), orders=hist, singular=sing)

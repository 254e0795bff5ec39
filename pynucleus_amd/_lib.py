"""ctypes binding of libpnl_hip.so (the C ABI in include/pnl_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be
loaded, importing a GPU entry point raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C pynucleus_amd/csrc``.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PNL_LIB', os.path.join(_HERE, 'libpnl_hip.so'))      # PNL_LIB: A/B builds of the library (tuning)

PNL_OK = 0
PNL_ERR_INVALID = -1
PNL_ERR_UNSUPPORTED = -2
PNL_ERR_HIP = -3
PNL_ERR_STATE = -4
PNL_ERR_ORDER = -5
PNL_INTERIOR = 0
PNL_BOUNDARY = 1
PNL_FLAG_NO_MIRROR = 1
PNL_FLAG_SYMMETRIC_FLUSH = 2
PNL_NUM_COUNTERS = 134

# every symbol include/pnl_hip.h declares (checked by tests/test_abi.py)
EXPORTS = ['pnl_create', 'pnl_destroy', 'pnl_error_string', 'pnl_version', 'pnl_set_stream', 'pnl_synchronize',
           'pnl_upload_mesh', 'pnl_upload_dofmap', 'pnl_set_kernel', 'pnl_set_order_formula', 'pnl_upload_distant_rules',
           'pnl_upload_singular_rule', 'pnl_upload_boundary', 'pnl_assemble_dense', 'pnl_dense_overwrites', 'pnl_block_row_costs', 'pnl_tile_cells',
           'pnl_assemble_dense_tiles', 'pnl_get_counters', 'pnl_get_phase_ms', 'pnl_get_kernel_ms', 'pnl_tree_build', 'pnl_tree_build_blocks', 'pnl_tree_build_refined', 'pnl_tree_build_horizon', 'pnl_tree_build_device', 'pnl_tree_destroy', 'pnl_tree_sizes', 'pnl_tree_get', 'pnl_tree_node_cells', 'pnl_h2_transfer_matrices', 'pnl_nfplan_build', 'pnl_nfplan_destroy', 'pnl_nfplan_sizes', 'pnl_nfplan_get', 'pnl_horizon_pattern', 'pnl_near_pattern', 'pnl_pattern_set_max_nnz', 'pnl_set_option', 'pnl_set_cell_order', 'pnl_set_interaction_transform', 'pnl_set_order_vertex_values', 'pnl_h2_get', 'pnl_h2_set', 'pnl_pattern_nnz', 'pnl_pattern_get', 'pnl_pattern_destroy', 'pnl_set_row_slab', 'pnl_diag_blocks_size', 'pnl_get_diag_blocks', 'pnl_slab_matvec', 'pnl_slab_diagonal', 'pnl_gemv', 'pnl_cg_jacobi',
           'pnl_inv_diagonal', 'pnl_set_classes', 'pnl_select_class', 'pnl_upload_sparsity', 'pnl_upload_sparsity_device', 'pnl_assemble_pairs_masked', 'pnl_assemble_boundary_masked', 'pnl_assemble_clusters_tiled', 'pnl_h2_setup', 'pnl_h2_matvec', 'pnl_h2_upward', 'pnl_h2_interact', 'pnl_h2_downward', 'pnl_h2_sizes', 'pnl_spmv',
           'pnl_assemble_pairs_in_horizon', 'pnl_assemble_pairs_in_horizon_range', 'pnl_set_nonsymmetric', 'pnl_set_order_function', 'pnl_upload_pointwise_rules', 'pnl_assemble_dense_pointwise',
           'pnl_assemble_pairs_masked_pointwise', 'pnl_assemble_boundary_masked_pointwise',
           'pnl_gemv_axpby', 'pnl_csr_matvec', 'pnl_mg_create', 'pnl_mg_destroy', 'pnl_mg_cycle', 'pnl_mg_solve', 'pnl_mg_cg', 'pnl_theta_step']


def source_sha16():
    """hash of the kernel / ABI sources the library is built from (identifies a build independently of the compiler's output:
    bench.py uses the PMC traffic record of profiles/pmc_traffic.json only if it was measured with these sources)"""
    import glob
    import hashlib
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.h'))+glob.glob(os.path.join(_HERE, 'csrc', '*.hip'))
                   +[os.path.join(root, 'include', 'pnl_hip.h'), os.path.join(_HERE, 'csrc', 'Makefile')])
    h = hashlib.sha256()
    for fn in files:
        with open(fn, 'rb') as f:
            h.update(os.path.basename(fn).encode()+b'\0'+f.read())
    return h.hexdigest()[:16]


class pnl_kernel(C.Structure):
    _fields_ = [('ktype', C.c_int32), ('interaction', C.c_int32), ('exponent', C.c_double), ('scale', C.c_double),
                ('horizon2', C.c_double)]


class pnl_order_formula(C.Structure):
    _fields_ = [('c0', C.c_double), ('a', C.c_double), ('b', C.c_double), ('e', C.c_double), ('den0', C.c_double),
                ('clip_num', C.c_int32), ('pad', C.c_int32)]


class pnl_mg_level_desc(C.Structure):
    _fields_ = [('n', C.c_int32), ('pad', C.c_int32), ('A_dev', C.c_void_p), ('ldA', C.c_int64), ('diag_dev', C.c_void_p),
                ('R_indptr_dev', C.c_void_p), ('R_indices_dev', C.c_void_p), ('R_data_dev', C.c_void_p),
                ('P_indptr_dev', C.c_void_p), ('P_indices_dev', C.c_void_p), ('P_data_dev', C.c_void_p),
                ('kind', C.c_int32), ('pad2', C.c_int32), ('near_indptr_dev', C.c_void_p), ('near_indices_dev', C.c_void_p),
                ('near_data_dev', C.c_void_p)]


class pnl_order_function(C.Structure):
    _fields_ = [('type', C.c_int32), ('normalized', C.c_int32), ('p', C.c_double*6), ('scal_n', C.c_int32), ('pad', C.c_int32),
                ('scal_mid', C.c_double), ('scal_half', C.c_double), ('scal_cheb', C.c_double*32)]


class pnl_cluster_plan(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('npairs', 'nnodes', 'nchunks', 'chunk_stride', 'ntiles', 'num_dslots', 'nfacets', 'tile')] +
                [(n, C.c_void_p) for n in ('pair_nodes', 'node_off', 'node_dofs', 'chunk_cells', 'chunk_ndof', 'chunk_dofs', 'chunk_slot',
                                           'tile_chunkA', 'tile_chunkB', 'tile_pair', 'tile_flags', 'tile_dslotA', 'tile_dslotB',
                                           'd_cell', 'd_pair')] +
                [('n_sing', C.c_int32*3), ('n_btouch', C.c_int32), ('sing_items', C.c_void_p*3)] +
                [(n, C.c_void_p) for n in ('pair_foff', 'fvid', 'bt_slot', 'bt_cell', 'bt_facet')])


class pnl_h2_plan(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('nnodes', 'nleaves', 'nfar', 'm', 'nlevels', 'nq')] +
                [(n, C.c_void_p) for n in ('box', 'parent', 'level', 'leaf_node', 'leaf_dof_off', 'leaf_dofs', 'leaf_cell_off',
                                           'leaf_cells', 'far', 'transfer', 'qbary', 'qw', 'qphi', 'far_class')] +
                [('partial_leaves', C.c_int32)])


class PnlError(RuntimeError):
    pass


_LIB = None


def load():
    """dlopen libpnl_hip.so and declare the prototypes; raises if the library is not built"""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PnlError('{} is missing: the HIP extension has not been built (run __graft_entry__.build()); '
                       'there is no CPU fallback for the assembly path'.format(LIB_PATH))
    # torch ships its own HIP runtime: it has to be in the process BEFORE this library resolves libamdhip64, or the two
    # runtimes collide and no device is found (the host-only planning entry points would otherwise load us first)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    L.pnl_create.argtypes = [i32, C.POINTER(vp)]
    L.pnl_destroy.argtypes = [vp]
    L.pnl_destroy.restype = None
    L.pnl_error_string.argtypes = [vp]
    L.pnl_error_string.restype = C.c_char_p
    L.pnl_version.restype = C.c_char_p
    L.pnl_set_stream.argtypes = [vp, vp]
    L.pnl_synchronize.argtypes = [vp]
    L.pnl_upload_mesh.argtypes = [vp, i32, i32, vp, i32, vp, vp, vp, dbl]
    L.pnl_upload_dofmap.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    L.pnl_set_kernel.argtypes = [vp, i32, C.POINTER(pnl_kernel)]
    L.pnl_set_order_formula.argtypes = [vp, i32, C.POINTER(pnl_order_formula)]
    L.pnl_upload_distant_rules.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.pnl_upload_singular_rule.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, dbl]
    L.pnl_upload_boundary.argtypes = [vp, i32, vp]
    L.pnl_assemble_dense.argtypes = [vp, vp, i64, i32, i32, i32, i32]
    L.pnl_dense_overwrites.argtypes = [vp, i32, i32, i32]
    L.pnl_block_row_costs.argtypes = [vp, vp, i32]
    L.pnl_tile_cells.argtypes = [vp]
    L.pnl_assemble_dense_tiles.argtypes = [vp, vp, i64, i32, i32, vp, i32, i32, i32]
    L.pnl_get_counters.argtypes = [vp, vp, i32]
    L.pnl_get_phase_ms.argtypes = [vp, vp, i32]
    L.pnl_get_kernel_ms.argtypes = [vp, vp, i32]
    L.pnl_set_row_slab.argtypes = [vp, i32, vp, i32, vp]
    L.pnl_tree_build.argtypes = [i32, i32, vp, vp, vp, i32, dbl, i32, i32, i32, C.POINTER(vp)]
    L.pnl_tree_build_blocks.argtypes = [i32, i32, vp, vp, vp, i32, dbl, i32, i32, i32, vp, i32, C.POINTER(vp)]
    L.pnl_tree_build_refined.argtypes = [i32, i32, vp, vp, vp, i32, dbl, i32, i32, i32, vp, i32, i32, C.POINTER(vp)]
    L.pnl_tree_build_horizon.argtypes = [i32, i32, vp, vp, vp, i32, dbl, i32, i32, i32, vp, i32, i32, dbl, C.POINTER(vp)]
    L.pnl_tree_build_device.argtypes = [i32, i32, vp, vp, vp, i32, dbl, i32, i32, i32, i32, C.POINTER(vp)]
    L.pnl_tree_destroy.argtypes = [vp]
    L.pnl_tree_destroy.restype = None
    L.pnl_tree_sizes.argtypes = [vp, vp]
    L.pnl_tree_get.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pnl_tree_node_cells.argtypes = [vp, i32, vp, vp, vp]
    L.pnl_h2_transfer_matrices.argtypes = [i32, i32, i32, vp, vp, vp]
    L.pnl_nfplan_build.argtypes = [i32, i32, vp, i32, vp, i32, i32, vp, i32, vp, vp, vp, vp, i32, vp, i32, i32, C.POINTER(vp)]
    L.pnl_nfplan_destroy.argtypes = [vp]
    L.pnl_nfplan_destroy.restype = None
    L.pnl_nfplan_sizes.argtypes = [vp, vp]
    L.pnl_nfplan_get.argtypes = [vp, i32, vp]
    L.pnl_horizon_pattern.argtypes = [i32, i32, vp, i32, vp, i32, i32, vp, dbl, i32, C.POINTER(vp)]
    L.pnl_near_pattern.argtypes = [vp, i32, vp, i32, C.POINTER(vp)]
    L.pnl_pattern_nnz.argtypes = [vp]
    L.pnl_pattern_set_max_nnz.argtypes = [i64]
    L.pnl_set_option.argtypes = [C.c_char_p, C.c_char_p]
    L.pnl_set_cell_order.argtypes = [vp, i32, vp]
    L.pnl_set_interaction_transform.argtypes = [vp, i32, vp]
    L.pnl_set_order_vertex_values.argtypes = [vp, i32, vp]
    L.pnl_h2_get.argtypes = [vp, i32, vp]
    L.pnl_h2_set.argtypes = [vp, i32, vp]
    L.pnl_pattern_get.argtypes = [vp, vp, vp]
    L.pnl_pattern_destroy.argtypes = [vp]
    L.pnl_diag_blocks_size.argtypes = [vp]
    L.pnl_get_diag_blocks.argtypes = [vp, vp]
    L.pnl_slab_matvec.argtypes = [vp, vp, i64, vp, vp, vp]
    L.pnl_slab_diagonal.argtypes = [vp, vp, i64, vp, vp]
    L.pnl_gemv.argtypes = [vp, vp, i64, i32, vp, vp, i32]
    L.pnl_cg_jacobi.argtypes = [vp, vp, i64, i32, vp, vp, dbl, i32, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.pnl_inv_diagonal.argtypes = [vp, vp, i64, i32, vp]
    L.pnl_gemv_axpby.argtypes = [vp, vp, i64, i32, i32, vp, dbl, dbl, vp, vp]
    L.pnl_csr_matvec.argtypes = [vp, i32, vp, vp, vp, vp, dbl, dbl, vp]
    L.pnl_mg_create.argtypes = [vp, i32, C.POINTER(pnl_mg_level_desc), vp, dbl, i32, i32, C.POINTER(C.c_void_p)]
    L.pnl_mg_destroy.argtypes = [vp]
    L.pnl_mg_cycle.argtypes = [vp, vp, vp, i32]
    L.pnl_mg_solve.argtypes = [vp, vp, vp, dbl, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_double), i32]
    L.pnl_mg_cg.argtypes = [vp, vp, i64, vp, vp, dbl, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_double), i32]
    L.pnl_theta_step.argtypes = [vp, vp, i64, vp, vp, vp, dbl, dbl, vp, vp, dbl, i32, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.pnl_upload_sparsity.argtypes = [vp, i32, vp, vp]
    L.pnl_upload_sparsity_device.argtypes = [vp, i32, vp, vp]
    L.pnl_set_classes.argtypes = [vp, i32, i32, vp, vp, vp]
    L.pnl_select_class.argtypes = [vp, i32]
    L.pnl_set_nonsymmetric.argtypes = [vp, i32]
    L.pnl_assemble_pairs_masked.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pnl_assemble_boundary_masked.argtypes = [vp, i32, vp, vp, vp, dbl, vp, vp]
    L.pnl_spmv.argtypes = [vp, vp, vp, vp, vp]
    L.pnl_assemble_clusters_tiled.argtypes = [vp, C.POINTER(pnl_cluster_plan), i32, vp, vp]
    L.pnl_h2_setup.argtypes = [vp, C.POINTER(pnl_h2_plan)]
    L.pnl_h2_matvec.argtypes = [vp, vp, vp]
    L.pnl_h2_upward.argtypes = [vp, vp, vp]
    L.pnl_h2_interact.argtypes = [vp, vp, vp]
    L.pnl_h2_downward.argtypes = [vp, vp, vp]
    L.pnl_h2_sizes.argtypes = [vp, vp]
    L.pnl_assemble_pairs_in_horizon.argtypes = [vp, vp, vp]
    L.pnl_assemble_pairs_in_horizon_range.argtypes = [vp, vp, vp, i32, i32]
    L.pnl_set_order_function.argtypes = [vp, C.POINTER(pnl_order_function), vp, vp, dbl, dbl, dbl, dbl]
    L.pnl_upload_pointwise_rules.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.pnl_assemble_dense_pointwise.argtypes = [vp, vp, i64, i32, i32, i32, i32, vp, i32, vp]
    L.pnl_assemble_pairs_masked_pointwise.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pnl_assemble_boundary_masked_pointwise.argtypes = [vp, i32, vp, vp, vp, vp, vp, dbl, vp, vp]
    for name in EXPORTS:
        f = getattr(L, name)
        if name in ('pnl_pattern_nnz', 'pnl_pattern_set_max_nnz'):
            f.restype = C.c_int64
        elif name not in ('pnl_destroy', 'pnl_error_string', 'pnl_version'):
            f.restype = C.c_int
    _LIB = L
    return L


def set_option(name, value=None):
    """pnl_set_option: a library option (include/pnl_hip.h lists the ones a product build accepts); value None removes it"""
    rc = load().pnl_set_option(name.encode(), None if value is None else str(value).encode())
    if rc:
        raise PnlError('pnl_set_option({!r}) failed: {} (a product build accepts only the options listed in include/pnl_hip.h)'.format(name, rc))


def _hp(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data


class Context:
    """One pnl_context = one GPU + one HIP stream, single caller (like a reference builder object)."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.pnl_create(int(device), C.byref(h))
        if rc != PNL_OK:
            raise PnlError('pnl_create(device={}) failed with status {} (is a GPU visible?)'.format(device, rc))
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, 'h', None):
            self.L.pnl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc == PNL_OK:
            return
        msg = self.L.pnl_error_string(self.h).decode()
        if rc == PNL_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        if rc == PNL_ERR_INVALID:
            raise AssertionError(msg)
        raise PnlError('status {}: {}'.format(rc, msg))

    # -- uploads ---------------------------------------------------------
    def upload_tables(self, T):
        """push a nonlocalTables object (mesh, DoF map, kernel, rules) to the GPU"""
        L, h = self.L, self.h
        dm, mesh = T.dm, T.dm.mesh
        v, pv = _hp(mesh.vertices, np.float64)
        c, pc = _hp(mesh.cells, np.int32)
        vol, pvol = _hp(mesh.volVector, np.float64)
        hh, ph = _hp(mesh.hVector, np.float64)
        self.check(L.pnl_upload_mesh(h, T.dim, mesh.num_vertices, pv, mesh.num_cells, pc, pvol, ph, float(T.H0)))
        xf = getattr(getattr(T.kernel, 'interaction', None), 'transform', None)
        if xf is not None:
            xa, pxa = _hp(xf, np.float64)
            self.check(L.pnl_set_interaction_transform(h, T.dim, pxa))
        else:
            self.check(L.pnl_set_interaction_transform(h, T.dim, None))
        perm = getattr(dm, 'cell_permutation', None)
        if perm is not None:
            # renumbered cells (builder.with_cell_locality / label_blocks): touching pairs keep the caller's orientation
            po, ppo = _hp(perm, np.int32)
            self.check(L.pnl_set_cell_order(h, mesh.num_cells, ppo))
        d, pd = _hp(dm.dofs, np.int32)
        pt, ppt = _hp(T.dof_perm_table, np.int32)
        self.check(L.pnl_upload_dofmap(h, T.dpe, dm.dofs_per_vertex, dm.dofs_per_edge, dm.num_dofs, pd, ppt))
        self.dofs_per_element = int(T.dpe)
        off, poff = _hp(T.dist_off, np.int32)
        b, pb = _hp(T.dist_bary, np.float64)
        w, pw = _hp(T.dist_w, np.float64)
        phi, pphi = _hp(T.dist_phi, np.float64)
        fo, pfo = _hp(T.bfacet_off, np.int32)
        fb, pfb = _hp(T.bfacet_bary, np.float64)
        fw, pfw = _hp(T.bfacet_w, np.float64)
        self.check(L.pnl_upload_distant_rules(h, T.qcap, poff, pb, pw, pphi, pfo, pfb, pfw))
        if T.has_boundary_tables:
            bc, pbc = _hp(T.bcells, np.int32)
            self.check(L.pnl_upload_boundary(h, bc.shape[0], pbc))
        if getattr(T, 'pointwise', False):
            # non-symmetric kernel, order per quadrature point: order function, per-cell / per-facet maxima, near rules per
            # distinct order of the touching pairs
            self.check(L.pnl_set_classes(h, 1, 0, None, None, None))
            f = pnl_order_function(int(T.order_type), int(T.kernel.normalized), (C.c_double*6)(*[float(x) for x in T.order_params]))
            cheb = getattr(T, 'scaling_cheb', None)
            if cheb is not None:
                f.scal_n, f.scal_mid, f.scal_half = len(cheb[2]), cheb[0], cheb[1]
                for i, v in enumerate(cheb[2]):
                    f.scal_cheb[i] = float(v)
            cs, pcs = _hp(T.cell_smax, np.float64)
            fs, pfs = _hp(T.facet_smax, np.float64)
            self.check(L.pnl_set_order_function(h, C.byref(f), pcs, pfs, float(T.pw_c0), float(T.pw_bc0), float(T.sing_fac),
                                                float(T.bsing_fac)))
            if int(T.order_type) == 5:
                vs, pvs = _hp(T.order_vertex_values, np.float64)
                self.check(L.pnl_set_order_vertex_values(h, vs.shape[0], pvs))
            self.upload_pointwise_rules(T)
            return
        classes = getattr(T, 'classes', None)
        if classes:
            cl, pcl = _hp(T.cell_labels, np.int32)
            fl, pfl = _hp(T.facet_labels, np.int32)
            co, pco = _hp(T.cls_of, np.int32)
            self.check(L.pnl_set_classes(h, len(classes), T.num_labels, pcl, pfl if fl.shape[0] else None, pco))
            if getattr(T, 'nonsym', False):
                self.check(L.pnl_set_nonsymmetric(h, 1))
        else:
            self.check(L.pnl_set_classes(h, 1, 0, None, None, None))
            classes = [T]
        for k, Tk in enumerate(classes):
            self.check(L.pnl_select_class(h, k))
            self._set_kernel(PNL_INTERIOR, Tk.kernel, Tk.qo)
            for panel, r in Tk.singular.items():
                n, pn = _hp(r.nodes, np.float64)
                ww, pww = _hp(r.weights, np.float64)
                ps, pps = _hp(r.psi, np.float64)
                self.check(L.pnl_upload_singular_rule(h, PNL_INTERIOR, panel, r.num_nodes, r.rows, pn, pww, pps, float(Tk.sing_fac)))
            if T.has_boundary_tables:
                self._set_kernel(PNL_BOUNDARY, Tk.boundaryKernel, Tk.bqo)
                for panel, r in Tk.bsingular.items():
                    n, pn = _hp(r.nodes, np.float64)
                    ww, pww = _hp(r.weights, np.float64)
                    ps, pps = _hp(r.psi, np.float64)
                    self.check(L.pnl_upload_singular_rule(h, PNL_BOUNDARY, panel, r.num_nodes, r.rows, pn, pww, pps, float(Tk.bsing_fac)))
        self.check(L.pnl_select_class(h, 0))

    def upload_pointwise_rules(self, T):
        """near rules of a kernel with an order per quadrature point, keyed by the distinct orders of the touching pairs (again after
        nonlocalTables.need_boundary_keys extended the boundary keys)"""
        L, h = self.L, self.h
        R = T.pw_rules()
        for slot, (nodes, ww, phi0, phi1) in R['rules'].items():
            n, pn = _hp(nodes, np.float64)
            w_, pw_ = _hp(ww, np.float64)
            a0, p0 = _hp(phi0, np.float64)
            a1, p1 = _hp(phi1, np.float64)
            self.check(L.pnl_upload_pointwise_rules(h, PNL_INTERIOR, -(slot+1), n.shape[0], w_.shape[1], a0.shape[1], pn, pw_, p0, p1))
        for slot, (nodes, ww, phi) in R['brules'].items():
            n, pn = _hp(nodes, np.float64)
            w_, pw_ = _hp(ww, np.float64)
            a0, p0 = _hp(phi, np.float64)
            self.check(L.pnl_upload_pointwise_rules(h, PNL_BOUNDARY, -(slot+1), n.shape[0], w_.shape[1], a0.shape[1], pn, pw_, p0, None))
        self._pw_pairs = (np.ascontiguousarray(R['pairs'], dtype=np.int32), np.ascontiguousarray(R['bpairs'], dtype=np.int32))
        self._pw_keys = (np.ascontiguousarray(R['keys']), np.ascontiguousarray(R['bkeys']))

    def _set_kernel(self, which, kernel, formula):
        p = kernel.device_params()
        k = pnl_kernel(p['ktype'], p.get('interaction', 0), p['exponent'], p['scale'], p['horizon2'])
        self.check(self.L.pnl_set_kernel(self.h, which, C.byref(k)))
        f = pnl_order_formula(formula.c0, formula.a, formula.b, formula.e, formula.den0, int(formula.clip_num), 0)
        self.check(self.L.pnl_set_order_formula(self.h, which, C.byref(f)))

    # -- hot path --------------------------------------------------------
    def assemble_pairs_in_horizon(self, data_ptr, diag_ptr, cell_begin=None, cell_end=None):
        """finite horizon: candidate pairs generated on the device (no host pair list, no masks); with a range only the pairs whose
        first cell lies in [cell_begin, cell_end) (the reference's cellNo1 split)"""
        if cell_begin is None:
            self.check(self.L.pnl_assemble_pairs_in_horizon(self.h, C.c_void_p(data_ptr), C.c_void_p(diag_ptr) if diag_ptr else None))
        else:
            self.check(self.L.pnl_assemble_pairs_in_horizon_range(self.h, C.c_void_p(data_ptr), C.c_void_p(diag_ptr) if diag_ptr else None,
                                                                  int(cell_begin), int(cell_end)))

    def assemble_dense_pointwise(self, A_ptr, ldA, zero_exterior, cell_begin, cell_end):
        self.assembly_epoch = getattr(self, 'assembly_epoch', 0)+1     # host snapshots of dense operators are stale now
        pairs, bpairs = self._pw_pairs
        self.check(self.L.pnl_assemble_dense_pointwise(self.h, C.c_void_p(A_ptr), int(ldA), int(bool(zero_exterior)), int(cell_begin),
                                                       int(cell_end), pairs.shape[0], pairs.ctypes.data, bpairs.shape[0],
                                                       bpairs.ctypes.data))

    def set_stream(self, stream_ptr):
        self.check(self.L.pnl_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self.check(self.L.pnl_synchronize(self.h))

    def assemble_dense(self, A_ptr, ldA, zero_exterior, cell_begin, cell_end, flags=0):
        self.assembly_epoch = getattr(self, 'assembly_epoch', 0)+1     # host snapshots of dense operators are stale now
        self.check(self.L.pnl_assemble_dense(self.h, C.c_void_p(A_ptr), int(ldA), int(bool(zero_exterior)), int(cell_begin),
                                             int(cell_end), int(flags)))

    def block_row_costs(self, num_blocks):
        """estimated cost of every block row of the upper block triangle (units of one uniform order-2 tile)"""
        out = np.zeros(int(num_blocks))
        self.check(self.L.pnl_block_row_costs(self.h, out.ctypes.data, int(num_blocks)))
        return out

    def dense_overwrites(self, cell_begin, cell_end, flags=0):
        """True if assemble_dense with these arguments writes every entry of A (no zero fill needed before)"""
        rc = self.L.pnl_dense_overwrites(self.h, int(cell_begin), int(cell_end), int(flags))
        if rc < 0:
            self.check(rc)
        return rc == 1

    def tile_cells(self):
        rc = self.L.pnl_tile_cells(self.h)
        if rc <= 0:
            self.check(rc)
        return rc

    def assemble_dense_tiles(self, A_ptr, ldA, zero_exterior, tiles, cell_begin, cell_end, flags=0):
        self.assembly_epoch = getattr(self, 'assembly_epoch', 0)+1     # host snapshots of dense operators are stale now
        t, pt = _hp(tiles, np.int32)
        self.check(self.L.pnl_assemble_dense_tiles(self.h, C.c_void_p(A_ptr), int(ldA), int(bool(zero_exterior)), t.shape[0], pt,
                                                   int(cell_begin), int(cell_end), int(flags)))

    # -- near field (assembleClusters) -------------------------------------
    def upload_sparsity(self, indptr, indices):
        ip, pip = _hp(indptr, np.int32)
        ix, pix = _hp(indices, np.int32)
        self.check(self.L.pnl_upload_sparsity(self.h, ix.shape[0], pip, pix))

    def upload_sparsity_device(self, indptr_t, indices_t):
        """pattern as int32 torch tensors on this context's device"""
        self.check(self.L.pnl_upload_sparsity_device(self.h, int(indices_t.numel()), C.c_void_p(indptr_t.data_ptr()),
                                                     C.c_void_p(indices_t.data_ptr()) if indices_t.numel() else None))

    def assemble_pairs_masked(self, pairs, masks, data_ptr, diag_ptr=None):
        p, pp = _hp(pairs, np.int32)
        if masks is None:                       # every entry of every pair
            pm = None
        else:
            m, pm = _hp(masks, np.uint64)
            assert m.shape == (p.shape[0], 4)
        assert p.ndim == 2 and p.shape[1] == 2
        self.check(self.L.pnl_assemble_pairs_masked(self.h, p.shape[0], pp, pm, C.c_void_p(data_ptr),
                                                    C.c_void_p(diag_ptr) if diag_ptr else None))

    def assemble_pairs_masked_pointwise(self, pairs, masks, rule, data_ptr):
        """near field of a kernel with an order per quadrature point: ORDERED pairs, masks over the (2 dpe)^2 local entries,
        rule[t] = -1 (no common vertex) or the key of the touching pair's near rule"""
        p, pp = _hp(pairs, np.int32)
        m, pm = _hp(masks, np.uint64)
        r, pr = _hp(rule, np.int32)
        assert p.ndim == 2 and p.shape[1] == 2 and m.shape == (p.shape[0], 4) and r.shape == (p.shape[0],)
        self.check(self.L.pnl_assemble_pairs_masked_pointwise(self.h, p.shape[0], pp, pm, pr, C.c_void_p(data_ptr)))

    def assemble_boundary_masked_pointwise(self, cells, facets, masks, rule, sv, fac, data_ptr, diag_ptr=None):
        c, pc = _hp(cells, np.int32)
        f, pf = _hp(facets, np.int32)
        m, pm = _hp(masks, np.uint32)
        r, pr = _hp(rule, np.int32)
        s_, ps = _hp(sv, np.float64)
        assert f.shape[0] == c.shape[0] == m.shape[0] == r.shape[0] == s_.shape[0]
        self.check(self.L.pnl_assemble_boundary_masked_pointwise(self.h, c.shape[0], pc, pf, pm, pr, ps, float(fac), C.c_void_p(data_ptr),
                                                                 C.c_void_p(diag_ptr) if diag_ptr else None))

    def select_class(self, k):
        """variable order: the kernel class the next pnl_assemble_boundary_masked integrates with"""
        self.check(self.L.pnl_select_class(self.h, int(k)))

    def assemble_boundary_masked(self, cells, facets, masks, fac, data_ptr, diag_ptr=None):
        c, pc = _hp(cells, np.int32)
        f, pf = _hp(facets, np.int32)
        m, pm = _hp(masks, np.uint32)
        assert f.shape[0] == c.shape[0] == m.shape[0]
        self.check(self.L.pnl_assemble_boundary_masked(self.h, c.shape[0], pc, pf, pm, float(fac), C.c_void_p(data_ptr),
                                                       C.c_void_p(diag_ptr) if diag_ptr else None))

    def assemble_clusters_tiled(self, plan, cluster_boundary, data_ptr, diag_ptr=None):
        """plan: clusters.nearFieldPlan"""
        keep = []

        def ptr(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        P = pnl_cluster_plan()
        P.npairs, P.nnodes, P.nchunks = plan.num_pairs, len(plan.nodes), plan.chunk_cells.shape[0]
        P.chunk_stride, P.ntiles, P.num_dslots = plan.nU, plan.tile_chunkA.shape[0], plan.num_dslots
        P.nfacets, P.tile = plan.fvid.shape[0], plan.tile
        for name in ('pair_nodes', 'node_off', 'node_dofs', 'chunk_cells', 'chunk_ndof', 'chunk_dofs', 'tile_chunkA', 'tile_chunkB',
                     'tile_pair', 'tile_flags', 'tile_dslotA', 'tile_dslotB', 'd_cell', 'd_pair', 'pair_foff', 'fvid', 'bt_slot',
                     'bt_cell', 'bt_facet'):
            setattr(P, name, ptr(getattr(plan, name), np.int32))
        P.chunk_slot = ptr(plan.chunk_slot, np.int16)
        for s in range(3):
            P.n_sing[s] = plan.sing_items[s].shape[0]
            P.sing_items[s] = ptr(plan.sing_items[s], np.int32)
        P.n_btouch = plan.bt_cell.shape[0]
        self.check(self.L.pnl_assemble_clusters_tiled(self.h, C.byref(P), int(bool(cluster_boundary)), C.c_void_p(data_ptr),
                                                      C.c_void_p(diag_ptr) if diag_ptr else None))

    def spmv(self, data_ptr, diag_ptr, x_ptr, y_ptr):
        self.check(self.L.pnl_spmv(self.h, C.c_void_p(data_ptr), C.c_void_p(diag_ptr) if diag_ptr else None, C.c_void_p(x_ptr),
                                   C.c_void_p(y_ptr)))

    def counters(self):
        out = np.zeros(PNL_NUM_COUNTERS, dtype=np.int64)
        self.check(self.L.pnl_get_counters(self.h, out.ctypes.data, PNL_NUM_COUNTERS))
        hist = {q: int(out[8+q]) for q in range(120) if out[8+q]}
        sing = {-1-k: int(out[128+k]) for k in range(3)}
        return dict(numCellPairs=int(out[0]), numAssembledCellPairs=int(out[1]), numIntegrations=int(out[2]),
                    numBoundaryPairs=int(out[3]), numBoundaryIntegrations=int(out[4]), orders=hist, singular=sing,
                    uniformTilePairs=int(out[6]), uniformTilePairsByOrder={2+k: int(out[131+k]) for k in range(3)})

    def phase_ms(self):
        out = np.zeros(7, dtype=np.float32)
        self.check(self.L.pnl_get_phase_ms(self.h, out.ctypes.data, 7))
        return dict(tiles=float(out[0]), tiles_uniform=float(out[6]), worklist=float(out[1]), singular=float(out[2]),
                    boundary=float(out[3]), scatter_mirror=float(out[4]), total=float(out[5]))

    def set_row_slab(self, rowdofs, coldofs):
        rd = np.ascontiguousarray(rowdofs, dtype=np.int32)
        cd = np.ascontiguousarray(coldofs, dtype=np.int32)
        self.check(self.L.pnl_set_row_slab(self.h, int(rd.shape[0]), rd.ctypes.data if rd.shape[0] else None, int(cd.shape[0]),
                                           cd.ctypes.data if cd.shape[0] else None))

    def diag_blocks_size(self):
        n = self.L.pnl_diag_blocks_size(self.h)
        self.check(min(n, 0))
        return n

    def get_diag_blocks(self, dst_ptr):
        self.check(self.L.pnl_get_diag_blocks(self.h, C.c_void_p(dst_ptr)))

    def slab_matvec(self, slab_ptr, ld, dblocks_ptr, x_ptr, y_ptr):
        self.check(self.L.pnl_slab_matvec(self.h, C.c_void_p(slab_ptr), int(ld), C.c_void_p(dblocks_ptr), C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def slab_diagonal(self, slab_ptr, ld, dblocks_ptr, diag_ptr):
        self.check(self.L.pnl_slab_diagonal(self.h, C.c_void_p(slab_ptr), int(ld), C.c_void_p(dblocks_ptr), C.c_void_p(diag_ptr)))

    def kernel_ms(self):
        """device time of each tile-kernel launch of the last assembly (HIP events on the library's stream)"""
        out = np.zeros(6, dtype=np.float32)
        self.check(self.L.pnl_get_kernel_ms(self.h, out.ctypes.data, 6))
        return dict(tile_general=float(out[0]), tile_uniform2=float(out[1]), tile_uniform3=float(out[2]), tile_uniform4=float(out[3]),
                    fold_mirror=float(out[4]), worklist=float(out[5]))

    def gemv(self, A_ptr, ldA, n, x_ptr, y_ptr, symmetric_half=False):
        self.check(self.L.pnl_gemv(self.h, C.c_void_p(A_ptr), int(ldA), int(n), C.c_void_p(x_ptr), C.c_void_p(y_ptr),
                                   int(symmetric_half)))

    def cg_jacobi(self, A_ptr, ldA, n, b_ptr, x_ptr, tol, maxiter):
        it, res = C.c_int(0), C.c_double(0.)
        self.check(self.L.pnl_cg_jacobi(self.h, C.c_void_p(A_ptr), int(ldA), int(n), C.c_void_p(b_ptr), C.c_void_p(x_ptr),
                                        float(tol), int(maxiter), C.byref(it), C.byref(res)))
        return it.value, res.value

    # -- solver side (multigrid, theta stepping) --------------------------------------------------
    def gemv_axpby(self, A_ptr, ldA, nrows, ncols, x_ptr, alpha, beta, b_ptr, y_ptr):
        self.check(self.L.pnl_gemv_axpby(self.h, C.c_void_p(A_ptr), int(ldA), int(nrows), int(ncols), C.c_void_p(x_ptr), float(alpha),
                                         float(beta), C.c_void_p(b_ptr) if b_ptr else None, C.c_void_p(y_ptr)))

    def csr_matvec(self, nrows, indptr_ptr, indices_ptr, data_ptr, x_ptr, alpha, beta, y_ptr):
        self.check(self.L.pnl_csr_matvec(self.h, int(nrows), C.c_void_p(indptr_ptr), C.c_void_p(indices_ptr), C.c_void_p(data_ptr),
                                         C.c_void_p(x_ptr), float(alpha), float(beta), C.c_void_p(y_ptr)))

    def mg_create(self, descs, coarse_inverse_ptr, omega, presmooth, postsmooth):
        arr = (pnl_mg_level_desc*len(descs))(*descs)
        out = C.c_void_p()
        self.check(self.L.pnl_mg_create(self.h, len(descs), arr, C.c_void_p(coarse_inverse_ptr), float(omega), int(presmooth), int(postsmooth),
                                        C.byref(out)))
        return out

    def mg_destroy(self, mg):
        self.L.pnl_mg_destroy(mg)

    def mg_cycle(self, mg, b_ptr, x_ptr, x_is_zero):
        self.check(self.L.pnl_mg_cycle(mg, C.c_void_p(b_ptr), C.c_void_p(x_ptr), int(bool(x_is_zero))))

    def mg_solve(self, mg, b_ptr, x_ptr, tol, maxiter, x_is_zero):
        it = C.c_int(0)
        res = (C.c_double*(maxiter+2))()
        self.check(self.L.pnl_mg_solve(mg, C.c_void_p(b_ptr), C.c_void_p(x_ptr), float(tol), int(maxiter), int(bool(x_is_zero)), C.byref(it),
                                       res, maxiter+2))
        return it.value, list(res[:it.value+1])

    def mg_cg(self, mg, A_ptr, ldA, b_ptr, x_ptr, tol, maxiter, x_is_zero):
        it = C.c_int(0)
        res = (C.c_double*(maxiter+2))(*([float('nan')]*(maxiter+2)))
        self.check(self.L.pnl_mg_cg(mg, C.c_void_p(A_ptr) if A_ptr else None, int(ldA), C.c_void_p(b_ptr), C.c_void_p(x_ptr), float(tol),
                                    int(maxiter), int(bool(x_is_zero)), C.byref(it), res, maxiter+2))
        out = list(res)                       # initial value + one entry per iteration performed; unwritten entries stay NaN
        while len(out) > 1 and out[-1] != out[-1]:
            out.pop()
        return it.value, out

    def theta_step(self, mg, S_ptr, ldS, M_indptr_ptr, M_indices_ptr, M_data_ptr, dt, theta, forcing_ptr, u_ptr, tol, maxiter):
        it, res = C.c_int(0), C.c_double(0.)
        self.check(self.L.pnl_theta_step(mg, C.c_void_p(S_ptr), int(ldS), C.c_void_p(M_indptr_ptr), C.c_void_p(M_indices_ptr),
                                         C.c_void_p(M_data_ptr), float(dt), float(theta), C.c_void_p(forcing_ptr) if forcing_ptr else None,
                                         C.c_void_p(u_ptr), float(tol), int(maxiter), C.byref(it), C.byref(res)))
        return it.value, res.value

    def inv_diagonal(self, A_ptr, ldA, n, out_ptr):
        self.check(self.L.pnl_inv_diagonal(self.h, C.c_void_p(A_ptr), int(ldA), int(n), C.c_void_p(out_ptr)))

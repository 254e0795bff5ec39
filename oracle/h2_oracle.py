"""CPU oracle of the H2 far field -- TEST INFRASTRUCTURE ONLY (numpy restatement).

Follows /root/reference/nl/PyNucleus_nl/clusterMethodCy.pyx:
  enterLeafValues :1205-1325           V_leaf[dof, alpha] = sum_cells sum_j vol w_j phi_dof(x_j) L_alpha(x_j)
  transferMatrixBuilder.build :2010-2073   T[I, J] = L^parent_I(xi^child_J)
  assembleFarFieldInteractions :2153-2238  K[i, j] = -2 gamma(xi_i, eta_j) on the Chebyshev tensor grids of the two boxes
  upwardPass / downwardPass :1092-1180, H2Matrix.matvec :2269-2295
and returns the far-field part as a dense matrix: sum over admissible pairs of W_n1 K W_n2^T with W_leaf = V_leaf and
W_parent[rows of child] = W_child T_child^T.  Tensor index alpha = alpha_0 + m alpha_1 (the reference's productIterator
order is irrelevant as long as it is used consistently).  Pinned only through H2-vs-dense accuracy (the reference's own
stored matvec error for the disc, tests/cache_testDistOp...: 8.1e-5 on a meshpy mesh, is an order-of-magnitude anchor).
"""
import numpy as np


def cheb_nodes(a, b, m):
    j = np.arange(m)
    eta = np.cos((2.0*(m-j)-1.0)/(2.0*m)*np.pi)                # CM:1255
    return (b-a)*0.5*(eta+1.0)+a


def lagrange(nodes, l, x):
    v = np.ones_like(x)
    for k in range(nodes.shape[0]):
        if k != l:
            v = v*(x-nodes[k])/(nodes[l]-nodes[k])
    return v


def tensor_index(m, dim):
    """alpha -> (alpha_0, alpha_1), coordinate 0 fastest"""
    idx = np.arange(m**dim)
    return np.stack([(idx//m**d) % m for d in range(dim)], axis=1)


def leaf_values(dm, node, m, qr):
    mesh = dm.mesh
    dim = mesh.dim
    M = m**dim
    al = tensor_index(m, dim)
    xi = [cheb_nodes(node.box[d, 0], node.box[d, 1], m) for d in range(dim)]
    V = np.zeros((node.dofs.shape[0], M))
    phi = dm.evalShapeFunctions(qr.nodes)                       # [dpe, nq]
    for c in node.cells:
        simplex = mesh.vertices[mesh.cells[c]]
        x = qr.nodes.T@simplex                                  # [nq, dim]
        vol = mesh.volVector[c]
        L1 = [np.stack([lagrange(xi[d], l, x[:, d]) for l in range(m)]) for d in range(dim)]     # [m, nq] per coordinate
        for k in range(dm.dofs_per_element):
            dof = dm.dofs[c, k]
            if dof < 0:
                continue
            pos = np.searchsorted(node.dofs, dof)
            if pos >= node.dofs.shape[0] or node.dofs[pos] != dof:
                continue
            for a in range(M):
                L = np.ones(x.shape[0])
                for d in range(dim):
                    L = L*L1[d][al[a, d]]
                V[pos, a] += vol*np.sum(qr.weights*phi[k]*L)
    return V


def transfer(boxP, boxC, m):
    dim = boxP.shape[0]
    al = tensor_index(m, dim)
    M = m**dim
    T = np.ones((M, M))
    for d in range(dim):
        xp, xc = cheb_nodes(boxP[d, 0], boxP[d, 1], m), cheb_nodes(boxC[d, 0], boxC[d, 1], m)
        Ld = np.stack([lagrange(xp, l, xc) for l in range(m)])  # [parent l, child node]
        T = T*Ld[al[:, d][:, None], al[:, d][None, :]]
    return T


def far_kernel(kernel, box1, box2, m):
    dim = box1.shape[0]
    al = tensor_index(m, dim)
    x = np.stack([cheb_nodes(box1[d, 0], box1[d, 1], m)[al[:, d]] for d in range(dim)], axis=1)
    y = np.stack([cheb_nodes(box2[d, 0], box2[d, 1], m)[al[:, d]] for d in range(dim)], axis=1)
    K = np.zeros((x.shape[0], y.shape[0]))
    for i in range(x.shape[0]):
        for j in range(y.shape[0]):
            K[i, j] = -2.0*kernel(x[i], y[j])
    return K


def far_field_dense(dm, kernel, root, Pfar, m, qr):
    N = dm.num_dofs
    W = {}

    def basis(n):
        if id(n) in W:
            return W[id(n)]
        if n.is_leaf:
            w = leaf_values(dm, n, m, qr)
        else:
            w = np.zeros((n.dofs.shape[0], m**dm.mesh.dim))
            for c in n.children:
                rows = np.searchsorted(n.dofs, c.dofs)
                w[rows] = basis(c)@transfer(n.box, c.box, m).T
        W[id(n)] = w
        return w

    A = np.zeros((N, N))
    for lvl in Pfar:
        for cp in Pfar[lvl]:
            K = far_kernel(kernel, cp.n1.box, cp.n2.box, m)
            A[np.ix_(cp.n1.dofs, cp.n2.dofs)] += basis(cp.n1)@K@basis(cp.n2).T
    return A

"""The entity numbering dolfinx 0.9.0 gives a mesh it reads (`XDMFFile.read_mesh` -> `mesh::create_mesh`), restated on
plain arrays so that tags computed here can be compared INDEX BY INDEX with a dolfinx run (SURVEY.md section 8 f1;
the reference's golden files, tests/tests_data/*_tags.csv, are lists in this numbering, compared element-wise at
tests/test_compute_meshtags.py:239-243).

What dolfinx does, serial case [3P: libdolfinx 0.9.0, not in /root/reference -- restated from its published algorithm and
PINNED by the reference's goldens: with this numbering all 132 robust golden cases of the four test meshes (cells AND
facets, box and sub-mesh mode) are reproduced element by element, tests/test_oracle_golden.py]:

  1. cells: the dual graph (cells sharing a facet; the neighbours of a cell in the order of the lexicographically sorted
     shared facets) is reordered with Gibbs-Poole-Stockmeyer, `graph::reorder_gps` (SIAM J. Numer. Anal. 13 (1976) 236-250):
       I.   pseudo-diameter: start v = first node of minimal degree; level structure L(v) by breadth-first search; the nodes
            of its last level in order of increasing degree: a deeper structure restarts from that node, otherwise u = the
            one of smallest width;
       II.  level pairs (i, j) = (level in L(v), depth - 1 - level in L(u)); nodes with i == j are placed; the connected
            components of the rest, largest first, go to the side whose largest GROWN level is smaller -- on a tie to the
            structure of smaller (or equal: L(v)) width;
       III. numbering level by level from v (from u, with the levels reversed and the result reversed back, if u has the
            smaller degree): within a level the unnumbered neighbours of the lowest numbered node, by increasing degree; a
            level's leftovers start again from their node of minimal degree; then the next level's nodes adjacent to the
            numbered ones.
     Sorting is stable (insertion sort on these short lists).
  2. vertices: numbered in order of first appearance walking the reordered cells, each cell's vertices as listed.
  3. facets: the lexicographic rank of their sorted vertex tuples in that vertex numbering.

Three of the reference's four test meshes (square_quad, square_tri, coarse_square -- written by dolfinx) are FIXED POINTS:
the reordering is the identity on them.  `disk` (212 triangles) is not.

Branches the goldens exercise: the restart of step I, the tie rule of step II.  Not exercised by any golden: the
interchange of u and v in step III (the paper's rule is implemented).
"""
import numpy as np

_FACET_VERTS = {
    "triangle": ((1, 2), (0, 2), (0, 1)),
    "quadrilateral": ((0, 1), (0, 2), (1, 3), (2, 3)),          # tensor-product vertex order
    "tetrahedron": ((1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)),
}


def dual_graph(cell_type, cells):
    """Neighbour lists of the cells; the neighbours of a cell appear in the order of the sorted shared facets."""
    cells = np.asarray(cells, dtype=np.int64)
    fv = _FACET_VERTS[cell_type]
    fac = []
    for c, row in enumerate(cells):
        for loc in fv:
            fac.append((tuple(sorted(int(row[i]) for i in loc)), c))
    order = sorted(range(len(fac)), key=lambda i: fac[i][0])   # stable
    adj = [[] for _ in range(cells.shape[0])]
    i = 0
    while i + 1 < len(order):
        a, b = fac[order[i]], fac[order[i + 1]]
        if a[0] == b[0]:
            adj[a[1]].append(b[1])
            adj[b[1]].append(a[1])
            i += 2
        else:
            i += 1
    return adj


def _level_structure(adj, s, allowed=None):
    seen = {s}
    levels = [[s]]
    while True:
        nxt = []
        for node in levels[-1]:
            for w in adj[node]:
                if w not in seen and (allowed is None or w in allowed):
                    seen.add(w)
                    nxt.append(w)
        if not nxt:
            return levels
        levels.append(nxt)


def _components(adj, nodes):
    inside, seen, comps = set(nodes), set(), []
    for c in nodes:
        if c in seen:
            continue
        r = [c]
        seen.add(c)
        k = 0
        while k < len(r):
            for w in adj[r[k]]:
                if w in inside and w not in seen:
                    seen.add(w)
                    r.append(w)
            k += 1
        comps.append(r)
    comps.sort(key=lambda a: -len(a))   # stable: largest first
    return comps


def _gps_component(adj, nodes):
    """Order (a list of nodes) of one connected component."""
    deg = {i: len(adj[i]) for i in nodes}
    allowed = set(nodes)
    v = min(nodes, key=lambda i: deg[i])                       # first node of minimal degree
    lv = _level_structure(adj, v, allowed)
    u, lu = v, lv
    done = False
    while not done:
        done = True
        wmin = None
        for s in sorted(lv[-1], key=lambda i: deg[i]):
            lt = _level_structure(adj, s, allowed)
            if len(lt) > len(lv):
                v, lv, done = s, lt, False
                break
            w = max(len(level) for level in lt)
            if wmin is None or w < wmin:
                wmin, u, lu = w, s, lt
    k = len(lv)
    pair = {}
    for i in range(k):
        for w in lv[i]:
            pair.setdefault(w, [0, 0])[0] = i
        for w in lu[i]:
            pair.setdefault(w, [0, 0])[1] = k - 1 - i
    ls = [[] for _ in range(k)]
    rest = []
    for i in nodes:
        (ls[pair[i][0]] if pair[i][0] == pair[i][1] else rest).append(i)
    wv, wu = max(len(level) for level in lv), max(len(level) for level in lu)
    for r in _components(adj, rest):
        wn = [len(level) for level in ls]
        wh, wl = wn[:], wn[:]
        for w in r:
            wh[pair[w][0]] += 1
            wl[pair[w][1]] += 1
        h0 = max([a for a, b in zip(wh, wn) if a > b] + [0])
        l0 = max([a for a, b in zip(wl, wn) if a > b] + [0])
        side = 0 if h0 < l0 else 1 if l0 < h0 else (0 if wv <= wu else 1)
        for w in r:
            ls[pair[w][side]].append(w)
    reverse = deg[u] < deg[v]
    if reverse:
        v = u
        ls.reverse()
    level_of = {w: i for i, level in enumerate(ls) for w in level}
    numbered = {v}
    rv = [v]
    cur = 0
    for li in range(k):
        while True:
            while cur < len(rv):
                node = rv[cur]
                if level_of[node] == li:
                    nb = sorted((w for w in adj[node] if w in allowed and w not in numbered and level_of[w] == li),
                                key=lambda i: deg[i])
                    for w in nb:
                        rv.append(w)
                        numbered.add(w)
                cur += 1
            left = [w for w in ls[li] if w not in numbered]
            if not left:
                break
            w = min(left, key=lambda i: deg[i])
            rv.append(w)
            numbered.add(w)
        if li + 1 < k:
            first_next = len(rv)
            for node in [x for x in rv if level_of[x] == li]:
                nb = sorted((w for w in adj[node] if w in allowed and w not in numbered and level_of[w] == li + 1),
                            key=lambda i: deg[i])
                for w in nb:
                    rv.append(w)
                    numbered.add(w)
            cur = first_next
    if reverse:
        rv.reverse()
    return rv


def reorder_gps(adj):
    """new index of every node, `graph::reorder_gps`: components in the order of their first node."""
    n = len(adj)
    new = [-1] * n
    count = 0
    for s in range(n):
        if new[s] >= 0:
            continue
        comp = [w for level in _level_structure(adj, s) for w in level]
        comp.sort()
        for i, w in enumerate(_gps_component(adj, comp)):
            new[w] = count + i
        count += len(comp)
    return np.asarray(new, dtype=np.int64)


def dolfinx_numbering(cell_type, cells):
    """(cell_new[nc], vertex_new[nv_used]): index dolfinx gives to input cell c / input vertex v (vertices the cells do
    not use get -1).  `cells`: (nc, nvpc) in the caller's vertex numbering, quadrilaterals in tensor-product order."""
    cells = np.asarray(cells, dtype=np.int64)
    cell_new = reorder_gps(dual_graph(cell_type, cells))
    inv = np.empty_like(cell_new)
    inv[cell_new] = np.arange(cell_new.size)
    vertex_new = -np.ones(int(cells.max()) + 1, dtype=np.int64)
    c = 0
    for row in cells[inv]:
        for v in row:
            if vertex_new[v] < 0:
                vertex_new[v] = c
                c += 1
    return cell_new, vertex_new


def as_dolfinx_reads_it(cell_type, x, cells):
    """The mesh renumbered the way dolfinx's read_mesh would hold it: (x_new, cells_new, cell_new, vertex_new).  Tagging
    this mesh (oracle or HIP path: both keep the caller's cell and vertex order and number facets by sorted vertex
    tuples) yields tag arrays in dolfinx's own indices."""
    x = np.asarray(x, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int64)
    cell_new, vertex_new = dolfinx_numbering(cell_type, cells)
    inv = np.empty_like(cell_new)
    inv[cell_new] = np.arange(cell_new.size)
    used = np.flatnonzero(vertex_new >= 0)
    x_new = np.empty((used.size, x.shape[1]))
    x_new[vertex_new[used]] = x[used]
    return x_new, vertex_new[cells[inv]], cell_new, vertex_new

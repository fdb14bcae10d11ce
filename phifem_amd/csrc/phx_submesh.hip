// Sub-mesh of Omega_h = cells tagged 1 or 2, with transferred tags.
// Replaces dolfinx.mesh.create_submesh [3P] + _transfer_tags, src/phifem/mesh_scripts.py:217-281,
// 636-645.  This is the reference's "negligible" stage: index shuffling on the
// host around the same device mesh constructor; it is not on the timed path.
#include <string.h>

#include <algorithm>
#include <vector>

#include "phx_common.h"

extern "C" int phx_submesh_create(phx_mesh *m, phx_mesh **sub_out) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->have_cell_tags && m->have_facet_tags, PHX_ERR_VALUE,
              "cell and facet tags must be computed before the sub-mesh");
  const int nvpc = m->ci.nvpc, nfpc = m->ci.nfpc;
  std::vector<int8_t> ct((size_t)m->nc), ft((size_t)m->nf);
  std::vector<int32_t> cells((size_t)m->nc * nvpc), c2f((size_t)m->nc * nfpc);
  std::vector<double> x((size_t)m->nv * m->gdim);
  PHX_HIP(hipMemcpy(ct.data(), m->cell_tags, ct.size(), hipMemcpyDeviceToHost));
  PHX_HIP(hipMemcpy(ft.data(), m->facet_tags, ft.size(), hipMemcpyDeviceToHost));
  PHX_HIP(hipMemcpy(cells.data(), m->cells, cells.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  PHX_HIP(hipMemcpy(c2f.data(), m->c2f, c2f.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  PHX_HIP(hipMemcpy(x.data(), m->x, x.size() * sizeof(double), hipMemcpyDeviceToHost));
  // mesh_scripts.py:637: omega_h_cells = unique(find(1) U find(2))
  std::vector<int32_t> c_map;
  for (int64_t c = 0; c < m->nc; ++c) {
    const int t = ct[c] & PHX_TAG_MASK;
    if (t == 1 || t == 2) c_map.push_back((int32_t)c);
  }
  PHX_REQUIRE(!c_map.empty(), PHX_ERR_VALUE, "no cell is tagged 1 or 2: empty sub-mesh");
  std::vector<int32_t> renum((size_t)m->nv, -1), v_map;
  for (int32_t c : c_map)
    for (int i = 0; i < nvpc; ++i) renum[cells[(size_t)c * nvpc + i]] = 0;
  for (int64_t v = 0; v < m->nv; ++v)
    if (renum[v] == 0) { renum[v] = (int32_t)v_map.size(); v_map.push_back((int32_t)v); }
  std::vector<int32_t> scells(c_map.size() * nvpc);
  for (size_t i = 0; i < c_map.size(); ++i)
    for (int k = 0; k < nvpc; ++k) scells[i * nvpc + k] = renum[cells[(size_t)c_map[i] * nvpc + k]];
  std::vector<double> sx(v_map.size() * m->gdim);
  for (size_t i = 0; i < v_map.size(); ++i)
    for (int d = 0; d < m->gdim; ++d) sx[i * m->gdim + d] = x[(size_t)v_map[i] * m->gdim + d];
  phx_mesh *s = nullptr;
  PHX_CHECK(phx_mesh_create(m->gdim, m->cell_type, (int64_t)v_map.size(), sx.data(),
                            (int64_t)c_map.size(), scells.data(), m->device, &s));
  // tags: cells through c_map (mesh_scripts.py:238-239,265-268); facets through the first
  // occurrence of each sub-mesh facet in its flattened c->f table (:244-260)
  std::vector<int32_t> sct(c_map.size()), sft((size_t)s->nf, 0), sc2f(c_map.size() * nfpc);
  PHX_HIP(hipMemcpy(sc2f.data(), s->c2f, sc2f.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < c_map.size(); ++i) sct[i] = ct[c_map[i]] & PHX_TAG_MASK;
  std::vector<uint8_t> seen((size_t)s->nf, 0);
  for (size_t i = 0; i < sc2f.size(); ++i) {
    const int32_t sf = sc2f[i];
    if (seen[sf]) continue;
    seen[sf] = 1;
    const size_t cell = i / nfpc, lf = i % nfpc;
    sft[sf] = ft[c2f[(size_t)c_map[cell] * nfpc + lf]];
  }
  PHX_CHECK(phx_set_tags(s, 0, sct.data(), PHX_HOST));
  PHX_CHECK(phx_set_tags(s, 1, sft.data(), PHX_HOST));
  s->is_submesh = true;
  if (m->is_box) {
    s->on_box_lattice = true;
    for (int a = 0; a < 3; ++a) { s->box_n[a] = m->box_n[a]; s->box_h[a] = m->box_h[a]; }
    PHX_HIP(phx_malloc(&s->v2lat, sizeof(int32_t) * v_map.size()));
    PHX_HIP(phx_malloc(&s->lat2v, sizeof(int32_t) * (size_t)m->nv));
    PHX_HIP(hipMemcpy(s->v2lat, v_map.data(), sizeof(int32_t) * v_map.size(), hipMemcpyHostToDevice));
    PHX_HIP(hipMemcpy(s->lat2v, renum.data(), sizeof(int32_t) * (size_t)m->nv, hipMemcpyHostToDevice));
  } else if (m->on_box_lattice && !m->is_submesh) {
    // parent = a caller-supplied mesh on a tensor lattice (phx_mesh_create): compose its vertex maps
    std::vector<int32_t> pv2l((size_t)m->nv);
    PHX_HIP(hipMemcpy(pv2l.data(), m->v2lat, sizeof(int32_t) * (size_t)m->nv, hipMemcpyDeviceToHost));
    std::vector<int32_t> sv2l(v_map.size()), l2sv((size_t)m->nv, -1);
    for (size_t i = 0; i < v_map.size(); ++i) { sv2l[i] = pv2l[(size_t)v_map[i]]; l2sv[(size_t)sv2l[i]] = (int32_t)i; }
    s->on_box_lattice = true;
    for (int a = 0; a < 3; ++a) { s->box_n[a] = m->box_n[a]; s->box_h[a] = m->box_h[a]; }
    PHX_HIP(phx_malloc(&s->v2lat, sizeof(int32_t) * sv2l.size()));
    PHX_HIP(phx_malloc(&s->lat2v, sizeof(int32_t) * l2sv.size()));
    PHX_HIP(hipMemcpy(s->v2lat, sv2l.data(), sizeof(int32_t) * sv2l.size(), hipMemcpyHostToDevice));
    PHX_HIP(hipMemcpy(s->lat2v, l2sv.data(), sizeof(int32_t) * l2sv.size(), hipMemcpyHostToDevice));
  }
  s->c_map_h = (int32_t *)malloc(sizeof(int32_t) * c_map.size());
  s->v_map_h = (int32_t *)malloc(sizeof(int32_t) * v_map.size());
  memcpy(s->c_map_h, c_map.data(), sizeof(int32_t) * c_map.size());
  memcpy(s->v_map_h, v_map.data(), sizeof(int32_t) * v_map.size());
  *sub_out = s;
  return PHX_OK;
}

extern "C" int phx_submesh_maps(phx_mesh *sub, int32_t *c_map, int32_t *v_map) {
  PHX_REQUIRE(sub->is_submesh, PHX_ERR_VALUE, "not a sub-mesh");
  if (c_map) memcpy(c_map, sub->c_map_h, sizeof(int32_t) * (size_t)sub->nc);
  if (v_map) memcpy(v_map, sub->v_map_h, sizeof(int32_t) * (size_t)sub->nv);
  return PHX_OK;
}

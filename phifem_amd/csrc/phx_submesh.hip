// Sub-mesh of Omega_h = cells tagged 1 or 2, with transferred tags.
// Replaces dolfinx.mesh.create_submesh [3P] + _transfer_tags, src/phifem/mesh_scripts.py:217-281,
// 636-645: device-resident since round 2.
#include <string.h>

#include <algorithm>
#include <vector>

#include "phx_common.h"
#include "phx_select.h"

struct SelOmegaCells {
  const int8_t *t;
  __host__ __device__ bool operator()(const int32_t &c) const { const int v = t[c] & PHX_TAG_MASK; return v == 1 || v == 2; }
  __host__ __device__ const int8_t *bytes() const { return t; }
  __host__ __device__ bool test(int tag, int32_t) const { const int v = tag & PHX_TAG_MASK; return v == 1 || v == 2; }
};
struct SelFlag { const uint8_t *f; __host__ __device__ bool operator()(const int32_t &i) const { return f[i] != 0; } };

__global__ void k_sub_touch(int64_t ncs, int nvpc, const int32_t *__restrict__ c_map, const int32_t *__restrict__ cells,
                            uint8_t *__restrict__ touched) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= ncs * nvpc) return;
  touched[cells[(int64_t)c_map[i / nvpc] * nvpc + i % nvpc]] = 1;
}
__global__ void k_sub_renum(int64_t nvs, const int32_t *__restrict__ v_map, int32_t *__restrict__ renum) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < nvs) renum[v_map[i]] = (int32_t)i;
}
__global__ void k_sub_cells(int64_t ncs, int nvpc, const int32_t *__restrict__ c_map, const int32_t *__restrict__ cells,
                            const int32_t *__restrict__ renum, int32_t *__restrict__ scells) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= ncs * nvpc) return;
  scells[i] = renum[cells[(int64_t)c_map[i / nvpc] * nvpc + i % nvpc]];
}
__global__ void k_sub_coords(int64_t nvs, int gdim, const int32_t *__restrict__ v_map, const double *__restrict__ x,
                             double *__restrict__ sx) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nvs * gdim) return;
  sx[i] = x[(int64_t)v_map[i / gdim] * gdim + i % gdim];
}
__global__ void k_sub_cell_tags(int64_t ncs, const int32_t *__restrict__ c_map, const int8_t *__restrict__ ct,
                                int32_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < ncs) out[i] = ct[c_map[i]] & PHX_TAG_MASK;
}
// mesh_scripts.py:244-260: the tag of a sub-mesh facet is that of the parent facet at its FIRST occurrence in the
// flattened cell -> facet table = in its lowest-numbered cell (f2c[.][0])
__global__ void k_sub_facet_tags(int64_t nfs, int nfpc, const int32_t *__restrict__ sf2c, const int32_t *__restrict__ sc2f,
                                 const int32_t *__restrict__ c_map, const int32_t *__restrict__ pc2f,
                                 const int8_t *__restrict__ ft, int32_t *__restrict__ out) {
  const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (f >= nfs) return;
  const int64_t c = sf2c[2 * f];
  int lf = 0;
  for (int k = 0; k < nfpc; ++k)
    if (sc2f[c * nfpc + k] == (int32_t)f) lf = k;
  out[f] = ft[pc2f[(int64_t)c_map[c] * nfpc + lf]];
}
__global__ void k_sub_compose(int64_t nvs, const int32_t *__restrict__ v_map, const int32_t *__restrict__ pv2l,
                              int32_t *__restrict__ sv2l, int32_t *__restrict__ l2sv) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nvs) return;
  const int32_t l = pv2l[v_map[i]];
  sv2l[i] = l;
  l2sv[l] = (int32_t)i;
}

// Everything stays on the device (round 1 copied cells, coordinates, connectivities and tags of the PARENT to the
// host -- 4.5 GB at 256^3 -- and rebuilt the topology with a host sort): cell and vertex compaction by ordered
// selects, the facet numbering by phx_topology_build_device, the tag transfer by two gather kernels.  Only the two
// maps the API returns (phx_submesh_maps) are copied to the host.
extern "C" int phx_submesh_create(phx_mesh *m, phx_mesh **sub_out) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->have_cell_tags && m->have_facet_tags, PHX_ERR_VALUE,
              "cell and facet tags must be computed before the sub-mesh");
  hipStream_t st = m->stream;
  const int nvpc = m->ci.nvpc, nfpc = m->ci.nfpc;
  const dim3 block(256);
  // mesh_scripts.py:637: omega_h_cells = unique(find(1) U find(2))
  int32_t *c_map = nullptr, *v_map = nullptr, *renum = nullptr, *scells = nullptr, *tmp32 = nullptr;
  uint8_t *touched = nullptr;
  double *sx = nullptr;
  int64_t ncs = 0, nvs = 0;
  PHX_CHECK(phx_select_indices(st, m->nc, SelOmegaCells{m->cell_tags}, &c_map, &ncs));
  if (ncs == 0) {
    PHX_HIP(phx_free(c_map));
    phx_set_error("no cell is tagged 1 or 2: empty sub-mesh");
    return PHX_ERR_VALUE;
  }
  PHX_HIP(phx_malloc(&touched, (size_t)m->nv));
  PHX_HIP(hipMemsetAsync(touched, 0, (size_t)m->nv, st));
  k_sub_touch<<<dim3((unsigned)phx_div_up(ncs * nvpc, 256)), block, 0, st>>>(ncs, nvpc, c_map, m->cells, touched);
  PHX_CHECK(phx_select_indices(st, m->nv, SelFlag{touched}, &v_map, &nvs));   // ascending: the renumbering is monotone
  PHX_HIP(phx_malloc(&renum, sizeof(int32_t) * (size_t)m->nv));
  PHX_HIP(hipMemsetAsync(renum, 0xff, sizeof(int32_t) * (size_t)m->nv, st));
  k_sub_renum<<<dim3((unsigned)phx_div_up(nvs, 256)), block, 0, st>>>(nvs, v_map, renum);
  PHX_HIP(phx_malloc(&scells, sizeof(int32_t) * (size_t)ncs * nvpc));
  PHX_HIP(phx_malloc(&sx, sizeof(double) * (size_t)nvs * m->gdim));
  k_sub_cells<<<dim3((unsigned)phx_div_up(ncs * nvpc, 256)), block, 0, st>>>(ncs, nvpc, c_map, m->cells, renum, scells);
  k_sub_coords<<<dim3((unsigned)phx_div_up(nvs * m->gdim, 256)), block, 0, st>>>(nvs, m->gdim, v_map, m->x, sx);
  PHX_HIP(hipStreamSynchronize(st));
  phx_mesh *s = nullptr;
  int rc = phx_mesh_create_from(m->gdim, m->cell_type, nvs, sx, ncs, scells, PHX_DEVICE, m->device, &s);
  PHX_HIP(phx_free(scells)); PHX_HIP(phx_free(sx)); PHX_HIP(phx_free(touched));
  if (rc != PHX_OK) { (void)phx_free(c_map); (void)phx_free(v_map); (void)phx_free(renum); return rc; }
  // tags: cells through c_map (mesh_scripts.py:238-239,265-268); facets through the first occurrence (:244-260)
  PHX_HIP(phx_malloc(&tmp32, sizeof(int32_t) * (size_t)std::max<int64_t>(ncs, s->nf)));
  k_sub_cell_tags<<<dim3((unsigned)phx_div_up(ncs, 256)), block, 0, st>>>(ncs, c_map, m->cell_tags, tmp32);
  PHX_HIP(hipStreamSynchronize(st));   // s has its own stream
  PHX_CHECK(phx_set_tags(s, 0, tmp32, PHX_DEVICE));
  k_sub_facet_tags<<<dim3((unsigned)phx_div_up(s->nf, 256)), block, 0, st>>>(s->nf, nfpc, s->f2c, s->c2f, c_map, m->c2f,
                                                                              m->facet_tags, tmp32);
  PHX_HIP(hipStreamSynchronize(st));
  PHX_CHECK(phx_set_tags(s, 1, tmp32, PHX_DEVICE));
  PHX_HIP(phx_free(tmp32));
  s->is_submesh = true;
  if (m->is_box) {
    s->on_box_lattice = true;
    for (int a = 0; a < 3; ++a) { s->box_n[a] = m->box_n[a]; s->box_h[a] = m->box_h[a]; }
    s->v2lat = v_map;     // vertex of the sub-mesh -> lattice point (= parent vertex of a generated box)
    s->lat2v = renum;     // lattice point -> vertex of the sub-mesh, -1: none
  } else if (m->on_box_lattice && !m->is_submesh) {
    // parent = a caller-supplied mesh on a tensor lattice (phx_mesh_create): compose its vertex maps
    s->on_box_lattice = true;
    for (int a = 0; a < 3; ++a) { s->box_n[a] = m->box_n[a]; s->box_h[a] = m->box_h[a]; }
    PHX_HIP(phx_malloc(&s->v2lat, sizeof(int32_t) * (size_t)nvs));
    PHX_HIP(phx_malloc(&s->lat2v, sizeof(int32_t) * (size_t)m->nv));
    PHX_HIP(hipMemsetAsync(s->lat2v, 0xff, sizeof(int32_t) * (size_t)m->nv, st));
    k_sub_compose<<<dim3((unsigned)phx_div_up(nvs, 256)), block, 0, st>>>(nvs, v_map, m->v2lat, s->v2lat, s->lat2v);
    PHX_HIP(hipStreamSynchronize(st));
  }
  s->c_map_h = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncs);
  s->v_map_h = (int32_t *)malloc(sizeof(int32_t) * (size_t)nvs);
  PHX_HIP(hipMemcpy(s->c_map_h, c_map, sizeof(int32_t) * (size_t)ncs, hipMemcpyDeviceToHost));
  PHX_HIP(hipMemcpy(s->v_map_h, v_map, sizeof(int32_t) * (size_t)nvs, hipMemcpyDeviceToHost));
  PHX_HIP(phx_free(c_map));
  if (!m->is_box) { PHX_HIP(phx_free(v_map)); PHX_HIP(phx_free(renum)); }
  *sub_out = s;
  return PHX_OK;
}

extern "C" int phx_submesh_maps(phx_mesh *sub, int32_t *c_map, int32_t *v_map) {
  PHX_REQUIRE(sub->is_submesh, PHX_ERR_VALUE, "not a sub-mesh");
  if (c_map) memcpy(c_map, sub->c_map_h, sizeof(int32_t) * (size_t)sub->nc);
  if (v_map) memcpy(v_map, sub->v_map_h, sizeof(int32_t) * (size_t)sub->nv);
  return PHX_OK;
}

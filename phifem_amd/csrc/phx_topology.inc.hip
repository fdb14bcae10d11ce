// Device-side facet numbering and connectivities (include of phx_mesh.hip): the numbering phx_topology_build_host
// defines -- facets in lexicographic order of their SORTED vertex tuples, the cells of a facet in ascending order --
// built with stable radix sorts instead of a host sort.  Used by phx_mesh_create (caller-supplied meshes) and by the
// device-resident sub-mesh (phx_submesh.hip).  Stands in for dolfinx create_connectivity
// (src/phifem/mesh_scripts.py:151-153,419-422) [3P].
struct FacetVertsTab { int nvpc, nfpc, nvpf; int fv[4][3]; };
struct TopoU8ToI32 { __host__ __device__ int32_t operator()(const uint8_t &v) const { return (int32_t)v; } };

// per (cell, local facet): sorted vertex tuple -> hi = (v0 << 32) | v1, lo = v2 (-1 -> 0xffffffff never compared in 2-D)
__global__ void k_facet_keys(int64_t nc, FacetVertsTab T, const int32_t *__restrict__ cells, int64_t nv,
                             unsigned long long *__restrict__ hi, uint32_t *__restrict__ lo, int32_t *__restrict__ idx,
                             int *__restrict__ bad) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nc * T.nfpc) return;
  const int64_t c = i / T.nfpc;
  const int lf = (int)(i - c * T.nfpc);
  int32_t v[3] = {0, 0, 0x7fffffff};
  for (int k = 0; k < T.nvpf; ++k) {
    v[k] = cells[c * T.nvpc + T.fv[lf][k]];
    if (v[k] < 0 || v[k] >= nv) atomicOr(bad, 1);
  }
  // sort 2 or 3 values
  if (v[0] > v[1]) { const int32_t t = v[0]; v[0] = v[1]; v[1] = t; }
  if (T.nvpf == 3) {
    if (v[1] > v[2]) { const int32_t t = v[1]; v[1] = v[2]; v[2] = t; }
    if (v[0] > v[1]) { const int32_t t = v[0]; v[0] = v[1]; v[1] = t; }
  }
  hi[i] = ((unsigned long long)(uint32_t)v[0] << 32) | (uint32_t)v[1];
  lo[i] = T.nvpf == 3 ? (uint32_t)v[2] : 0u;
  idx[i] = (int32_t)i;
}

__global__ void k_gather_u64(int64_t n, const int32_t *__restrict__ idx, const unsigned long long *__restrict__ in,
                             unsigned long long *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}
__global__ void k_gather_u32(int64_t n, const int32_t *__restrict__ idx, const uint32_t *__restrict__ in,
                             uint32_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}

// head[i] = 1 where a new tuple starts in the sorted sequence
__global__ void k_facet_heads(int64_t n, const unsigned long long *__restrict__ hi, const uint32_t *__restrict__ lo,
                              uint8_t *__restrict__ head) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || hi[i] != hi[i - 1] || lo[i] != lo[i - 1]) ? 1 : 0;
}

// rank[i] = exclusive scan of head: facet id of sorted record i = rank[i] + head[i] - 1
__global__ void k_facet_fill(int64_t n, int nfpc, const uint8_t *__restrict__ head, const int32_t *__restrict__ rank,
                             const int32_t *__restrict__ idx, int32_t *__restrict__ c2f, int32_t *__restrict__ f2c,
                             int *__restrict__ bad) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t f = rank[i] + head[i] - 1;
  const int32_t rec = idx[i];               // cell * nfpc + lf; equal tuples keep ascending record (= cell) order
  c2f[rec] = f;
  if (head[i]) {
    f2c[2 * (int64_t)f] = rec / nfpc;
    const bool two = i + 1 < n && !head[i + 1];
    f2c[2 * (int64_t)f + 1] = two ? idx[i + 1] / nfpc : -1;
    if (two && i + 2 < n && !head[i + 2]) atomicOr(bad, 2);   // a facet shared by three cells: non-manifold
  }
}

// m->cells (device), m->nv, m->nc, m->ci set; allocates and fills m->c2f, m->f2c, sets m->nf
static int phx_topology_build_device(phx_mesh *m) {
  hipStream_t st = m->stream;
  const phx_cell_info &ci = m->ci;
  const int64_t n = m->nc * (int64_t)ci.nfpc;
  PHX_REQUIRE(m->nc > 0 && m->nv > 0, PHX_ERR_VALUE, "empty mesh");
  PHX_REQUIRE(n < INT32_MAX, PHX_ERR_VALUE, "mesh too large for 32-bit local ids");
  FacetVertsTab T;
  T.nvpc = ci.nvpc; T.nfpc = ci.nfpc; T.nvpf = ci.nvpf;
  for (int f = 0; f < 4; ++f) for (int k = 0; k < 3; ++k) T.fv[f][k] = ci.fv[f][k];
  unsigned long long *hi = nullptr, *hi2 = nullptr;
  uint32_t *lo = nullptr, *lo2 = nullptr;
  int32_t *idx = nullptr, *idx2 = nullptr, *rank = nullptr;
  uint8_t *head = nullptr;
  int *bad = nullptr;
  PHX_HIP(phx_malloc(&hi, sizeof(unsigned long long) * (size_t)n));
  PHX_HIP(phx_malloc(&hi2, sizeof(unsigned long long) * (size_t)n));
  PHX_HIP(phx_malloc(&lo, sizeof(uint32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&lo2, sizeof(uint32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&idx, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&idx2, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&rank, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&head, (size_t)n));
  PHX_HIP(phx_malloc(&bad, sizeof(int)));
  PHX_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
  const dim3 block(256), grid((unsigned)phx_div_up(n, 256));
  k_facet_keys<<<grid, block, 0, st>>>(m->nc, T, m->cells, m->nv, hi, lo, idx, bad);
  size_t b1 = 0, b2 = 0, b3 = 0;
  // least significant first: v2 (3-D only), then (v0, v1); radix sorts are stable, so equal tuples stay in record order
  PHX_HIP(phx_sort_pairs(nullptr, b1, lo, lo2, idx, idx2, (size_t)n, 0, 32, st));
  PHX_HIP(phx_sort_pairs(nullptr, b2, hi, hi2, idx, idx2, (size_t)n, 0, 64, st));
  PHX_HIP(phx_exclusive_sum(nullptr, b3, rocprim::make_transform_iterator(head, TopoU8ToI32()), rank, (size_t)(n), st));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, std::max(std::max(b1, b2), std::max(b3, (size_t)16))));
  int32_t *order = idx;     // permutation after the sorts
  if (ci.nvpf == 3) {
    PHX_HIP(phx_sort_pairs(tmp, b1, lo, lo2, idx, idx2, (size_t)n, 0, 32, st));
    // bring the major keys into the order of the first pass, sort by them
    k_gather_u64<<<grid, block, 0, st>>>(n, idx2, hi, hi2);
    PHX_HIP(phx_sort_pairs(tmp, b2, hi2, hi, idx2, idx, (size_t)n, 0, 64, st));
    order = idx;            // sorted major keys in `hi`
    k_gather_u32<<<grid, block, 0, st>>>(n, order, lo, lo2);   // lo (unsorted, by record) -> lo2 in final order
  } else {
    PHX_HIP(phx_sort_pairs(tmp, b2, hi, hi2, idx, idx2, (size_t)n, 0, 64, st));
    order = idx2;
    // sorted major keys are in hi2: move to hi for the code below
    PHX_HIP(hipMemcpyAsync(hi, hi2, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToDevice, st));
    PHX_HIP(hipMemsetAsync(lo2, 0, sizeof(uint32_t) * (size_t)n, st));
  }
  k_facet_heads<<<grid, block, 0, st>>>(n, hi, lo2, head);
  PHX_HIP(phx_exclusive_sum(tmp, b3, rocprim::make_transform_iterator(head, TopoU8ToI32()), rank, (size_t)(n), st));
  int32_t last_rank = 0;
  uint8_t last_head = 0;
  PHX_HIP(hipMemcpyAsync(&last_rank, rank + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
  PHX_HIP(hipMemcpyAsync(&last_head, head + (n - 1), 1, hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  const int64_t nf = (int64_t)last_rank + last_head;
  m->nf = nf;
  PHX_HIP(phx_malloc(&m->c2f, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&m->f2c, sizeof(int32_t) * (size_t)nf * 2));
  k_facet_fill<<<grid, block, 0, st>>>(n, ci.nfpc, head, rank, order, m->c2f, m->f2c, bad);
  int hbad = 0;
  PHX_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(hi)); PHX_HIP(phx_free(hi2)); PHX_HIP(phx_free(lo)); PHX_HIP(phx_free(lo2));
  PHX_HIP(phx_free(idx)); PHX_HIP(phx_free(idx2)); PHX_HIP(phx_free(rank)); PHX_HIP(phx_free(head));
  PHX_HIP(phx_free(bad)); PHX_HIP(phx_free(tmp));
  PHX_REQUIRE(!(hbad & 1), PHX_ERR_VALUE, "a cell has a vertex index out of range");
  PHX_REQUIRE(!(hbad & 2), PHX_ERR_VALUE, "a facet is shared by more than two cells (non-manifold mesh)");
  return PHX_OK;
}

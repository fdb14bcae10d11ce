// Ceiling of the y-pass access pattern (development aid): blocks copy tiles of W columns x (L-1) rows of an
// L x (L-1) x NZ f64 lattice in place (read the tile, write it back), nothing else; many blocks per CU hide the latency.
// usage: tile_copy NZ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int L, int W, int NTHR, int ROWS>   // a block moves ROWS rows of W columns (ROWS = L-1: whole columns)
__global__ void __launch_bounds__(NTHR) k_tile(double *G, int nz, int ncb, int nrb, int ntiles) {
  constexpr int RSTEP = NTHR / W, NT = (ROWS + RSTEP - 1) / RSTEP, len = L - 1;
  const int tid = threadIdx.x, tc = tid % W, r0 = tid / W;
  for (int q = blockIdx.x; q < ntiles; q += gridDim.x) {
    const int cb = q % ncb, rb = (q / ncb) % nrb, z = q / (ncb * nrb);
    double *base = G + (size_t)z * L * len + (size_t)rb * ROWS * L + cb * W + tc;
    const bool cok = cb * W + tc < len;
    double v[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int r = r0 + i * RSTEP;
      v[i] = (cok && r < ROWS && rb * ROWS + r < len) ? base[(size_t)r * L] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int r = r0 + i * RSTEP;
      if (cok && r < ROWS && rb * ROWS + r < len) base[(size_t)r * L] = v[i] + 1.0;
    }
  }
}

template <int L, int W, int NTHR, int ROWS>
void run(double *G, int nz, int per_cu) {
  const int ncb = (L - 1 + W - 1) / W, nrb = (L - 1 + ROWS - 1) / ROWS, ntiles = ncb * nrb * nz;
  const int grid = std::min(ntiles, per_cu * 256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) k_tile<L, W, NTHR, ROWS><<<grid, NTHR>>>(G, nz, ncb, nrb, ntiles);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) k_tile<L, W, NTHR, ROWS><<<grid, NTHR>>>(G, nz, ncb, nrb, ntiles);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps, bytes = 2.0 * 8.0 * (double)(L - 1) * (L - 1) * nz;
  printf("L=%d W=%3d NTHR=%4d ROWS=%3d blocks/CU=%d: %8.1f us  %.2f TB/s\n", L, W, NTHR, ROWS, per_cu, us, bytes / us / 1e6);
}

int main(int argc, char **argv) {
  const int nz = argc > 1 ? atoi(argv[1]) : 194;
  constexpr int L = 768;
  double *G; CK(hipMalloc(&G, sizeof(double) * (size_t)L * (L - 1) * nz));
  CK(hipMemset(G, 0, sizeof(double) * (size_t)L * (L - 1) * nz));
  printf("lattice %d x %d x %d f64 (%.0f MB)\n", L, L - 1, nz, 8.0 * L * (L - 1) * nz / 1e6);
  run<L, 16, 512, 767>(G, nz, 1); run<L, 16, 512, 767>(G, nz, 2); run<L, 16, 512, 767>(G, nz, 4);
  run<L, 16, 256, 767>(G, nz, 4); run<L, 16, 256, 767>(G, nz, 8);
  run<L, 16, 256, 128>(G, nz, 8); run<L, 16, 256, 64>(G, nz, 8);
  run<L, 32, 256, 767>(G, nz, 8); run<L, 32, 256, 64>(G, nz, 8);
  run<L, 64, 256, 767>(G, nz, 8); run<L, 64, 256, 64>(G, nz, 8);
  run<L, 8, 256, 767>(G, nz, 8); run<L, 8, 256, 64>(G, nz, 8);
  run<L, 256, 256, 64>(G, nz, 8); run<L, 256, 256, 8>(G, nz, 8);
  return 0;
}

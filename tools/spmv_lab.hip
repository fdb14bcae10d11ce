// SpMV kernel laboratory (development aid, not part of the library): times SELL-64 variants on a
// synthetic matrix shaped like the 256^3 bench system (7-point interior rows + wider "cut" rows).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/spmv_lab tools/spmv_lab.hip ; run: /tmp/spmv_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
#define C 64

// V0: library kernel (one wave per slice, 8-byte / 4-byte loads, unroll 4)
__global__ void __launch_bounds__(256) v0(int64_t n, int64_t ns, const int64_t* __restrict__ sp, const int32_t* __restrict__ col,
                                          const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t s = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= ns) return;
  const int64_t base = sp[s];
  const int width = (int)((sp[s + 1] - base) >> 6);
  const int32_t* c = col + base + lane;
  const double* v = val + base + lane;
  double acc = 0.0;
  int k = 0;
  for (; k + 4 <= width; k += 4) {
    const int32_t c0 = c[(k + 0) * C], c1 = c[(k + 1) * C], c2 = c[(k + 2) * C], c3 = c[(k + 3) * C];
    const double v0 = v[(k + 0) * C], v1 = v[(k + 1) * C], v2 = v[(k + 2) * C], v3 = v[(k + 3) * C];
    acc += v0 * x[c0]; acc += v1 * x[c1]; acc += v2 * x[c2]; acc += v3 * x[c3];
  }
  for (; k < width; ++k) acc += v[k * C] * x[c[k * C]];
  const int64_t row = s * C + lane;
  if (row < n) y[row] = acc;
}

// V1: V0 without the x gather (streaming floor of this structure)
__global__ void __launch_bounds__(256) v1(int64_t n, int64_t ns, const int64_t* __restrict__ sp, const int32_t* __restrict__ col,
                                          const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t s = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= ns) return;
  const int64_t base = sp[s];
  const int width = (int)((sp[s + 1] - base) >> 6);
  const int32_t* c = col + base + lane;
  const double* v = val + base + lane;
  double acc = 0.0;
  for (int k = 0; k < width; ++k) acc += v[k * C] * (double)c[k * C];
  const int64_t row = s * C + lane;
  if (row < n) y[row] = acc;
}

// V2: pair layout: entries (2j, 2j+1) of a row are adjacent -> 16-byte value loads, 8-byte column
// loads.  width2 = ceil(width/2) pairs; storage [slice][pair][lane][2]
__global__ void __launch_bounds__(256) v2(int64_t n, int64_t ns, const int64_t* __restrict__ sp2, const int2* __restrict__ col2,
                                          const double2* __restrict__ val2, const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t s = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= ns) return;
  const int64_t base = sp2[s];  // in pairs
  const int w2 = (int)((sp2[s + 1] - base) >> 6);
  const int2* c = col2 + base + lane;
  const double2* v = val2 + base + lane;
  double a0 = 0.0, a1 = 0.0;
  int k = 0;
  for (; k + 2 <= w2; k += 2) {
    const int2 ca = c[(k + 0) * C], cb = c[(k + 1) * C];
    const double2 va = v[(k + 0) * C], vb = v[(k + 1) * C];
    a0 += va.x * x[ca.x]; a1 += va.y * x[ca.y];
    a0 += vb.x * x[cb.x]; a1 += vb.y * x[cb.y];
  }
  for (; k < w2; ++k) { const int2 ca = c[k * C]; const double2 va = v[k * C]; a0 += va.x * x[ca.x]; a1 += va.y * x[ca.y]; }
  const int64_t row = s * C + lane;
  if (row < n) y[row] = a0 + a1;
}

// V3: V2 + persistent waves (grid-stride over slices)
__global__ void __launch_bounds__(256) v3(int64_t n, int64_t ns, const int64_t* __restrict__ sp2, const int2* __restrict__ col2,
                                          const double2* __restrict__ val2, const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t s = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6); s < ns; s += nw) {
    const int64_t base = sp2[s];
    const int w2 = (int)((sp2[s + 1] - base) >> 6);
    const int2* c = col2 + base + lane;
    const double2* v = val2 + base + lane;
    double a0 = 0.0, a1 = 0.0;
    int k = 0;
    for (; k + 2 <= w2; k += 2) {
      const int2 ca = c[(k + 0) * C], cb = c[(k + 1) * C];
      const double2 va = v[(k + 0) * C], vb = v[(k + 1) * C];
      a0 += va.x * x[ca.x]; a1 += va.y * x[ca.y];
      a0 += vb.x * x[cb.x]; a1 += vb.y * x[cb.y];
    }
    for (; k < w2; ++k) { const int2 ca = c[k * C]; const double2 va = v[k * C]; a0 += va.x * x[ca.x]; a1 += va.y * x[ca.y]; }
    const int64_t row = s * C + lane;
    if (row < n) y[row] = a0 + a1;
  }
}

// V4: 32 rows per wave, 2 lanes per row (half-wave k split) on the pair layout with C=32:
// storage [slice32][pair][32 rows][2]; lanes 0-31 take even pairs, 32-63 odd pairs
__global__ void __launch_bounds__(256) v4(int64_t n, int64_t ns32, const int64_t* __restrict__ sp, const int2* __restrict__ col2,
                                          const double2* __restrict__ val2, const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int64_t s = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= ns32) return;
  const int64_t base = sp[s];
  const int w2 = (int)((sp[s + 1] - base) >> 5);
  double a0 = 0.0, a1 = 0.0;
  for (int k = h; k < w2; k += 2) {
    const int2 ca = col2[base + (int64_t)k * 32 + r];
    const double2 va = val2[base + (int64_t)k * 32 + r];
    a0 += va.x * x[ca.x]; a1 += va.y * x[ca.y];
  }
  double a = a0 + a1;
  a += __shfl_xor(a, 32);
  const int64_t row = s * 32 + r;
  if (h == 0 && row < n) y[row] = a;
}

// streaming copy reference (16 B / lane)
__global__ void __launch_bounds__(256) copy16(int64_t n16, const double2* __restrict__ a, double2* __restrict__ b) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// streaming read reference: 16 B/lane reads, negligible writes
__global__ void __launch_bounds__(256) read16(int64_t n16, const double2* __restrict__ a, double* __restrict__ out) {
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) { double2 t = a[i]; acc += t.x + t.y; }
  if (acc == 1.2345) out[0] = acc;
}
__global__ void __launch_bounds__(256) read8(int64_t n8, const double* __restrict__ a, double* __restrict__ out) {
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) acc += a[i];
  if (acc == 1.2345) out[0] = acc;
}

template <typename F> double timeit(F f, int reps = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3;  // us
}

int main() {
  const int G = 143; const int64_t n = (int64_t)G * G * G;  // ~2.92M rows
  const int64_t ns = (n + 63) / 64;
  // row lengths: 7 interior; every 7th slice is "cut": 28 entries
  std::vector<int> width(ns);
  for (int64_t s = 0; s < ns; ++s) width[s] = (s % 7 == 3) ? 28 : 7;
  std::vector<int64_t> sp(ns + 1, 0), sp2(ns + 1, 0);
  for (int64_t s = 0; s < ns; ++s) { sp[s + 1] = sp[s] + (int64_t)width[s] * 64; sp2[s + 1] = sp2[s] + (int64_t)((width[s] + 1) / 2) * 64; }
  const int64_t nnzp = sp[ns], np2 = sp2[ns];
  std::vector<int32_t> col(nnzp); std::vector<double> val(nnzp);
  std::vector<int32_t> col2(2 * np2); std::vector<double> val2(2 * np2);
  const int64_t off[7] = {0, -1, 1, -G, G, -(int64_t)G * G, (int64_t)G * G};
  srand(1);
  for (int64_t s = 0; s < ns; ++s)
    for (int l = 0; l < 64; ++l) {
      const int64_t r = std::min(s * 64 + l, n - 1);
      for (int k = 0; k < width[s]; ++k) {
        int64_t c = k < 7 ? r + off[k] : r + (rand() % (4 * G * G)) - 2 * G * G;
        c = std::max<int64_t>(0, std::min(n - 1, c));
        const double v = 1.0 / (1 + k);
        col[sp[s] + (int64_t)k * 64 + l] = (int32_t)c; val[sp[s] + (int64_t)k * 64 + l] = v;
        col2[2 * (sp2[s] + (int64_t)(k / 2) * 64 + l) + (k & 1)] = (int32_t)c;
        val2[2 * (sp2[s] + (int64_t)(k / 2) * 64 + l) + (k & 1)] = v;
      }
      if (width[s] & 1) { col2[2 * (sp2[s] + (int64_t)(width[s] / 2) * 64 + l) + 1] = (int32_t)r; val2[2 * (sp2[s] + (int64_t)(width[s] / 2) * 64 + l) + 1] = 0.0; }
    }
  // C=32 pair layout for v4
  const int64_t ns32 = (n + 31) / 32;
  std::vector<int64_t> sp4(ns32 + 1, 0);
  for (int64_t s = 0; s < ns32; ++s) sp4[s + 1] = sp4[s] + (int64_t)((width[s / 2] + 1) / 2) * 32;
  std::vector<int32_t> col4(2 * sp4[ns32]); std::vector<double> val4(2 * sp4[ns32]);
  for (int64_t s = 0; s < ns32; ++s)
    for (int l = 0; l < 32; ++l) {
      const int64_t s64 = s / 2; const int l64 = (int)((s & 1) * 32 + l);
      const int w2 = (width[s64] + 1) / 2;
      for (int k = 0; k < w2; ++k)
        for (int q = 0; q < 2; ++q) {
          col4[2 * (sp4[s] + (int64_t)k * 32 + l) + q] = col2[2 * (sp2[s64] + (int64_t)k * 64 + l64) + q];
          val4[2 * (sp4[s] + (int64_t)k * 32 + l) + q] = val2[2 * (sp2[s64] + (int64_t)k * 64 + l64) + q];
        }
    }
  std::vector<double> x(n); for (int64_t i = 0; i < n; ++i) x[i] = (double)(i % 97) * 0.01;
  int64_t *dsp, *dsp2, *dsp4; int32_t *dcol, *dcol2, *dcol4; double *dval, *dval2, *dval4, *dx, *dy, *dy2;
  CK(hipMalloc(&dsp, 8 * (ns + 1))); CK(hipMalloc(&dsp2, 8 * (ns + 1))); CK(hipMalloc(&dsp4, 8 * (ns32 + 1)));
  CK(hipMalloc(&dcol, 4 * nnzp)); CK(hipMalloc(&dval, 8 * nnzp));
  CK(hipMalloc(&dcol2, 8 * np2)); CK(hipMalloc(&dval2, 16 * np2));
  CK(hipMalloc(&dcol4, 8 * sp4[ns32])); CK(hipMalloc(&dval4, 16 * sp4[ns32]));
  CK(hipMalloc(&dx, 8 * n)); CK(hipMalloc(&dy, 8 * n)); CK(hipMalloc(&dy2, 8 * n));
  CK(hipMemcpy(dsp, sp.data(), 8 * (ns + 1), hipMemcpyHostToDevice)); CK(hipMemcpy(dsp2, sp2.data(), 8 * (ns + 1), hipMemcpyHostToDevice));
  CK(hipMemcpy(dsp4, sp4.data(), 8 * (ns32 + 1), hipMemcpyHostToDevice));
  CK(hipMemcpy(dcol, col.data(), 4 * nnzp, hipMemcpyHostToDevice)); CK(hipMemcpy(dval, val.data(), 8 * nnzp, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcol2, col2.data(), 8 * np2, hipMemcpyHostToDevice)); CK(hipMemcpy(dval2, val2.data(), 16 * np2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcol4, col4.data(), 8 * sp4[ns32], hipMemcpyHostToDevice)); CK(hipMemcpy(dval4, val4.data(), 16 * sp4[ns32], hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice));
  const double bytes = 12.0 * nnzp + 20.0 * n;
  printf("n=%lld slices=%lld padded nnz=%lld pair-padded=%lld bytes=%.1f MB\n", (long long)n, (long long)ns, (long long)nnzp, (long long)(2 * np2), bytes / 1e6);
  const dim3 b(256), g((unsigned)((ns + 3) / 4)), g4((unsigned)((ns32 + 3) / 4));
  auto rep = [&](const char* name, double us) { printf("%-28s %8.1f us  %7.1f GB/s (algorithmic)\n", name, us, bytes / us / 1e3); };
  rep("v0 lib", timeit([&] { v0<<<g, b>>>(n, ns, dsp, dcol, dval, dx, dy); }));
  rep("v1 no-gather", timeit([&] { v1<<<g, b>>>(n, ns, dsp, dcol, dval, dx, dy2); }));
  rep("v2 pair16", timeit([&] { v2<<<g, b>>>(n, ns, dsp2, (const int2*)dcol2, (const double2*)dval2, dx, dy2); }));
  for (int gb : {1024, 2048, 4096, 8192}) { char nm[64]; snprintf(nm, 64, "v3 pair16 persistent %d", gb);
    rep(nm, timeit([&] { v3<<<dim3(gb), b>>>(n, ns, dsp2, (const int2*)dcol2, (const double2*)dval2, dx, dy2); })); }
  rep("v4 pair16 C32 2lanes/row", timeit([&] { v4<<<g4, b>>>(n, ns32, dsp4, (const int2*)dcol4, (const double2*)dval4, dx, dy2); }));
  // check v2 == v0
  std::vector<double> y0(n), y1(n);
  v0<<<g, b>>>(n, ns, dsp, dcol, dval, dx, dy); v4<<<g4, b>>>(n, ns32, dsp4, (const int2*)dcol4, (const double2*)dval4, dx, dy2);
  CK(hipMemcpy(y0.data(), dy, 8 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dy2, 8 * n, hipMemcpyDeviceToHost));
  double md = 0; for (int64_t i = 0; i < n; ++i) md = std::max(md, fabs(y0[i] - y1[i])); printf("max |v0-v4| = %g\n", md);
  // streaming references on the same number of bytes
  const int64_t n16 = (int64_t)(bytes / 16);
  double2 *da, *db; CK(hipMalloc(&da, 16 * n16)); CK(hipMalloc(&db, 16 * n16)); CK(hipMemset(da, 1, 16 * n16));
  double us = timeit([&] { read16<<<dim3(2048), b>>>(n16, da, dy); }); printf("%-28s %8.1f us  %7.1f GB/s\n", "read16 (same bytes)", us, 16.0 * n16 / us / 1e3);
  us = timeit([&] { read8<<<dim3(2048), b>>>(2 * n16, (const double*)da, dy); }); printf("%-28s %8.1f us  %7.1f GB/s\n", "read8 (same bytes)", us, 16.0 * n16 / us / 1e3);
  us = timeit([&] { copy16<<<dim3(2048), b>>>(n16 / 2, da, db); }); printf("%-28s %8.1f us  %7.1f GB/s (r+w)\n", "copy16 (half bytes r + w)", us, 16.0 * n16 / us / 1e3);
  return 0;
}

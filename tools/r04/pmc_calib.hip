// Calibration of FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ for the access shapes of this library: the
// guide's "x2" holds for 16-byte-per-lane streaming reads; the kernels here read 8 bytes per lane, gather 8-byte entries,
// or read 128-byte tile rows.  Each kernel moves a KNOWN number of bytes of a 2 GiB buffer (beyond the 256 MB MALL).
// usage (under rocprofv3 --pmc ...): pmc_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void calib_stream_read8(const double *__restrict__ a, int64_t n, double *out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 123.456) *out = s;
}
__global__ void calib_stream_read16(const double2 *__restrict__ a, int64_t n, double *out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 123.456) *out = s;
}
// every lane reads ONE 8-byte entry of its own 128-byte line (line index scrambled): n lines touched once each
__global__ void calib_gather8_line(const double *__restrict__ a, int64_t nlines, double *out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nlines; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = (i * 2654435761LL) % nlines;   // odd multiplier: a bijection when nlines is a power of two
    s += a[l * 16 + (i & 15)];
  }
  if (s == 123.456) *out = s;
}
// tile rows: 16 lanes read 128 contiguous bytes, consecutive rows 6144 bytes apart (the y pass of the 768 lattice)
__global__ void calib_rows128(const double *__restrict__ a, int64_t nrows, double *out) {
  double s = 0.0;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (int64_t r = t / 16; r < nrows; r += (int64_t)gridDim.x * blockDim.x / 16) s += a[r * 768 + (t & 15)];
  if (s == 123.456) *out = s;
}
__global__ void calib_stream_write8(double *__restrict__ a, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = 1.0;
}

int main() {
  const int64_t n = (int64_t)1 << 28;   // doubles: 2 GiB
  double *a, *out;
  CK(hipMalloc(&a, sizeof(double) * n));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, sizeof(double) * n));
  const dim3 g(256 * 8), b(256);
  for (int rep = 0; rep < 3; ++rep) {
    calib_stream_read8<<<g, b>>>(a, n, out);
    calib_stream_read16<<<g, b>>>((const double2 *)a, n / 2, out);
    calib_gather8_line<<<g, b>>>(a, n / 16, out);
    calib_rows128<<<g, b>>>(a, n / 768, out);
    calib_stream_write8<<<g, b>>>(a, n);
  }
  CK(hipDeviceSynchronize());
  printf("bytes per launch: stream_read8 %lld, stream_read16 %lld, gather8_line %lld useful / %lld in 128-B lines / %lld in 64-B sectors, rows128 %lld, stream_write8 %lld\n",
         (long long)(8 * n), (long long)(8 * n), (long long)(8 * (n / 16)), (long long)(128 * (n / 16)), (long long)(64 * (n / 16)),
         (long long)(128 * (n / 768)), (long long)(8 * n));
  return 0;
}

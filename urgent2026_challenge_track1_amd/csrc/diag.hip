// Diagnostic kernels (not on the product path): streaming reads / writes of a known byte count at a chosen per-lane access
// width, used to CALIBRATE the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for the access widths the recurrence kernels use
// (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half of a 16-B-per-lane streaming read on gfx950; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  scripts/pmc_calibrate.py drives them.
#include "urse_common.h"

namespace urse {
template <typename V>
__global__ void __launch_bounds__(256) diag_read_kernel(const V* __restrict__ src, float* __restrict__ sink, long n) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const V v = src[i];
    acc += reinterpret_cast<const float*>(&v)[0];
  }
  if (acc == 1234.5678f) sink[0] = acc;
}
template <typename V>
__global__ void __launch_bounds__(256) diag_write_kernel(V* __restrict__ dst, long n) {
  V v;
  for (unsigned k = 0; k < sizeof(V) / 4; ++k) reinterpret_cast<float*>(&v)[k] = 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = v;
}
}  // namespace urse

using namespace urse;

extern "C" int urse_diag_stream(void* buf, float* sink, int64_t bytes, int width, int write, void* stream) {
  URSE_CHECK_ARG(buf && sink && bytes > 0 && (width == 4 || width == 8 || width == 16), "urse_diag_stream: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long n = bytes / width;
  dim3 grid(2048), blk(256);
  if (write) {
    if (width == 4) hipLaunchKernelGGL(diag_write_kernel<float>, grid, blk, 0, st, (float*)buf, n);
    else if (width == 8) hipLaunchKernelGGL(diag_write_kernel<float2>, grid, blk, 0, st, (float2*)buf, n);
    else hipLaunchKernelGGL(diag_write_kernel<float4>, grid, blk, 0, st, (float4*)buf, n);
  } else {
    if (width == 4) hipLaunchKernelGGL(diag_read_kernel<float>, grid, blk, 0, st, (const float*)buf, sink, n);
    else if (width == 8) hipLaunchKernelGGL(diag_read_kernel<float2>, grid, blk, 0, st, (const float2*)buf, sink, n);
    else hipLaunchKernelGGL(diag_read_kernel<float4>, grid, blk, 0, st, (const float4*)buf, sink, n);
  }
  URSE_CHECK_LAUNCH("urse_diag_stream");
  return URSE_OK;
}

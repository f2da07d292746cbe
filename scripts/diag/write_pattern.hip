// Diagnostic: HBM write rate of the NT GEMM's output pattern (256-row x 448-byte tiles of a [M, 6272 B] matrix) against a
// linear fill of the same bytes.  Not part of the library.
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(512) tile_write(char* out, long M, long pitch, int tn, int seg, int nt) {
  const int tile = blockIdx.x;
  const int tm = tile / tn, tnn = tile - tm * tn;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  const int cpr = seg / 16;
  for (int idx = threadIdx.x; idx < 256 * cpr; idx += 512) {
    const int r = idx / cpr, c = idx - r * cpr;
    const long row = (long)tm * 256 + r;
    if (row >= M) continue;
    f32x4* p = reinterpret_cast<f32x4*>(out + row * pitch + (long)tnn * seg + c * 16);
    if (nt) __builtin_nontemporal_store(v, p); else *p = v;
  }
}
extern "C" __global__ void __launch_bounds__(512) linear_write(char* out, long bytes, int nt) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = ((long)blockIdx.x * 512 + threadIdx.x) * 16; i < bytes; i += (long)gridDim.x * 512 * 16) {
    f32x4* p = reinterpret_cast<f32x4*>(out + i);
    if (nt) __builtin_nontemporal_store(v, p); else *p = v;
  }
}
extern "C" int run(char* out, long M, long pitch, int seg, int mode, int nt, void* st) {
  const int tn = (int)(pitch / seg);
  const long tiles = ((M + 255) / 256) * tn;
  if (mode == 0) hipLaunchKernelGGL(tile_write, dim3((unsigned)tiles), dim3(512), 0, (hipStream_t)st, out, M, pitch, tn, seg, nt);
  else hipLaunchKernelGGL(linear_write, dim3(4096), dim3(512), 0, (hipStream_t)st, out, M * pitch, nt);
  return (int)hipGetLastError();
}

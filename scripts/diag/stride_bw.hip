// Diagnostic (not product code): what rate do the recurrences' row-scattered accesses reach by themselves?
// 256 workgroups x 7 waves; a wave owns 16 rows of a [M, ld] matrix and walks the columns in blocks, as the LSTM kernels do:
// per block and row 16 lanes touch SEG * 16 contiguous bytes (SEG = bytes per lane: 4 = f32 cell state, 8 = bf16 gates of one unit,
// 16 = gates of two adjacent units), the 16 rows of a wave-instruction group are `rowgap` rows apart.  mode 0 = store, 1 = load.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int SEG, int MODE>
__global__ void __launch_bounds__(448) k(char* base, long pitch, int nblk, int steps, long rowgap, long steprow, float* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane >> 4, lc = lane & 15;
  const long tile = (long)blockIdx.x * 7 + w;
  float acc = 0.f;
  for (int t = 0; t < steps; ++t) {
    for (int b = 0; b < nblk; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = (tile * 16 + lr * 4 + r) * rowgap + t * steprow;
        char* p = base + row * pitch + (long)b * 16 * SEG + lc * SEG;
        if (MODE == 0) {
          if (SEG == 4) *reinterpret_cast<float*>(p) = (float)t;
          else if (SEG == 8) *reinterpret_cast<float2*>(p) = make_float2((float)t, 1.f);
          else *reinterpret_cast<float4*>(p) = make_float4((float)t, 1.f, 2.f, 3.f);
        } else {
          if (SEG == 4) acc += *reinterpret_cast<const float*>(p);
          else if (SEG == 8) { float2 v = *reinterpret_cast<const float2*>(p); acc += v.x + v.y; }
          else { float4 v = *reinterpret_cast<const float4*>(p); acc += v.x + v.y + v.z + v.w; }
        }
      }
    }
  }
  if (MODE == 1 && acc == 123.456f) sink[0] = acc;
}

template <int SEG, int MODE>
static void run(const char* name, char* buf, long pitch, int nblk, float* sink) {
  const int steps = 34;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<SEG, MODE>), dim3(256), dim3(448), 0, 0, buf, pitch, nblk, steps, 34L, 1L, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double bytes = 256.0 * 7 * 16 * steps * nblk * 16 * SEG;
  printf("%-46s %.3f ms  %.2f TB/s  (%.0f MB)\n", name, best, bytes / best / 1e9, bytes / 1e6);
}

int main() {
  const long M = 256L * 7 * 16 * 34 + 64;
  const long pitch = 6272;
  char* buf; float* sink;
  hipMalloc(&buf, M * pitch); hipMalloc(&sink, 64);
  hipMemset(buf, 0, M * pitch);
  // gates-like: 3136 B per direction per row -> 24.5 blocks of 128 B; we use 24 blocks of 128 / 12 of 256 / 6 of 512
  run<8, 0>("store  8 B/lane (128 B per row, 24 blocks)", buf, pitch, 24, sink);
  run<16, 0>("store 16 B/lane (256 B per row, 12 blocks)", buf, pitch, 12, sink);
  run<4, 0>("store  4 B/lane ( 64 B per row, 24 blocks)", buf, pitch, 24, sink);
  run<8, 1>("load   8 B/lane (128 B per row, 24 blocks)", buf, pitch, 24, sink);
  run<16, 1>("load  16 B/lane (256 B per row, 12 blocks)", buf, pitch, 12, sink);
  run<4, 1>("load   4 B/lane ( 64 B per row, 24 blocks)", buf, pitch, 24, sink);
  hipDeviceSynchronize();
  return 0;
}

// Diagnostic (not product code): the MIXED memory pattern of the row-wave LSTM forward without any of its arithmetic or synchronisation:
// 256 workgroups x 7 waves, a wave owns 16 rows per step; per 16-unit block it loads 8 B (pre-activations) + 4 B (c_{t-1}) per lane and row
// and stores 8 B (gates) + 4 B (c_t); per step it stores 16 rows x 784 B of h.  Loads are consumed `pd` blocks after their issue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int PD, int LOADS, int STORES, int HOUT, int BARRIER>
__global__ void __launch_bounds__(448) k(char* gx, char* cbuf, char* hbuf, int steps, float* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane >> 4, lc = lane & 15;
  const long tile = (long)blockIdx.x * 7 + w;
  float acc = 0.f;
  float2 gq[PD + 1][4];
  float cq[PD + 1][4];
  for (int a = 0; a <= PD; ++a) for (int r = 0; r < 4; ++r) { gq[a][r] = make_float2(0.f, 0.f); cq[a][r] = 0.f; }
  for (int t = 0; t < steps; ++t) {
#pragma unroll 1
    for (int bo = 0; bo < 25; bo += 5) {
#pragma unroll
      for (int bi = 0; bi < 5; ++bi) {
        const int b = bo + bi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = (tile * 16 + lr * 4 + r) * 34 + t;
          if (LOADS) {
            gq[(bi + PD) % (PD + 1)][r] = *reinterpret_cast<const float2*>(gx + row * 6272 + b * 128 + lc * 8);
            cq[(bi + PD) % (PD + 1)][r] = *reinterpret_cast<const float*>(cbuf + row * 3136 + b * 64 + lc * 4);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = (tile * 16 + lr * 4 + r) * 34 + t;
          float2 v = gq[bi % (PD + 1)][r];
          float cv = cq[bi % (PD + 1)][r] + v.x;
          acc += v.y;
          if (STORES) {
            *reinterpret_cast<float2*>(gx + row * 6272 + b * 128 + lc * 8) = make_float2(cv, v.y);
            *reinterpret_cast<float*>(cbuf + row * 3136 + b * 64 + lc * 4) = cv;
          }
          if (BARRIER) __builtin_amdgcn_s_barrier();
        }
      }
    }
    if (HOUT) {
      for (int idx = lane; idx < 16 * 49; idx += 64) {
        const int row = idx / 49, cc = idx - row * 49;
        *reinterpret_cast<float4*>(hbuf + ((tile * 16 + row) * 34 + t) * 1600 + cc * 16) = make_float4(acc, 1.f, 2.f, 3.f);
      }
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int PD, int LOADS, int STORES, int HOUT, int BARRIER>
static void run(const char* name, char* gx, char* c, char* h, float* sink) {
  const int steps = 34;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int it = 0; it < 3; ++it) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<PD, LOADS, STORES, HOUT, BARRIER>), dim3(256), dim3(448), 0, 0, gx, c, h, steps, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double cells = 256.0 * 7 * 16 * steps * 400;
  const double bytes = cells * (LOADS * 12 + STORES * 12) + (HOUT ? 256.0 * 7 * 16 * steps * 784 : 0);
  printf("%-40s %.3f ms  %.1f us/step  %.2f TB/s\n", name, best, best * 1e3 / steps, bytes / best / 1e9);
}

int main() {
  const long M = 256L * 7 * 16 * 34 + 64;
  char *gx, *c, *h; float* sink;
  (void)hipMalloc(&gx, M * 6272); (void)hipMalloc(&c, M * 3136); (void)hipMalloc(&h, M * 1600); (void)hipMalloc(&sink, 64);
  (void)hipMemset(gx, 0, M * 6272); (void)hipMemset(c, 0, M * 3136);
  run<3, 1, 1, 1, 0>("loads + stores + hout, pd 3", gx, c, h, sink);
  run<3, 1, 1, 1, 1>("loads + stores + hout, pd 3, barriers", gx, c, h, sink);
  run<1, 1, 1, 1, 0>("loads + stores + hout, pd 1", gx, c, h, sink);
  run<3, 1, 0, 0, 0>("loads only, pd 3", gx, c, h, sink);
  run<3, 0, 1, 0, 0>("stores only", gx, c, h, sink);
  run<3, 0, 0, 1, 0>("hout only", gx, c, h, sink);
  run<3, 1, 1, 0, 0>("loads + stores, pd 3", gx, c, h, sink);
  (void)hipDeviceSynchronize();
  return 0;
}

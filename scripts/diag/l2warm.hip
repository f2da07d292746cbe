// Diagnostic (not part of the library): does a CU's L2-hit stream slow down when OTHER waves of the same CU miss to HBM, and how
// fast can lines be pulled into the XCD's L2 without using the vector memory path (scalar loads)?  Prices the "L2 warming helper"
// idea of DESIGN.md section 9.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

__device__ inline void s_touch8(const char* p, unsigned& acc) {
  unsigned r0, r1, r2, r3, r4, r5, r6, r7;
  asm volatile(
      "s_load_dword %0, %8, 0x0\n s_load_dword %1, %8, 0x80\n s_load_dword %2, %8, 0x100\n s_load_dword %3, %8, 0x180\n"
      "s_load_dword %4, %8, 0x200\n s_load_dword %5, %8, 0x280\n s_load_dword %6, %8, 0x300\n s_load_dword %7, %8, 0x380\n"
      "s_waitcnt lgkmcnt(0)\n"
      : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "=&s"(r4), "=&s"(r5), "=&s"(r6), "=&s"(r7)
      : "s"(p)
      : "memory");
  acc ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

// mode of the non-streaming waves: 0 idle, 1 vector stream from HBM (16 B per lane), 2 vector touch (4 B per lane, one line per
// lane), 3 scalar touch (one s_load_dword per line, 8 in flight).
extern "C" __global__ void __launch_bounds__(1024) mix_kernel(const char* hot, long hot_bytes, const char* cold, long cold_per_wave,
                                                              int passes, int n_stream, int mode, u64* stamps, unsigned* sink) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  unsigned acc = 0;
  __syncthreads();
  const u64 t0 = wall_clock64();
  if (w < n_stream) {
    // this wave's share of the hot buffer: chunks of 1 KiB (one dwordx4 wave load), interleaved over the streaming waves
    const long chunks = hot_bytes / 1024;
    for (int p = 0; p < passes; ++p) {
      for (long c = w; c < chunks; c += (long)n_stream * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          long cc = c + (long)u * n_stream;
          if (cc >= chunks) cc = w;
          v[u] = *reinterpret_cast<const f32x4*>(hot + cc * 1024 + l * 16);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= __float_as_uint(v[u].x) ^ __float_as_uint(v[u].w);
      }
    }
  } else if (mode != 0) {
    const char* base = cold + ((long)blockIdx.x * (nw - n_stream) + (w - n_stream)) * cold_per_wave;
    if (mode == 1) {
      for (long o = 0; o < cold_per_wave; o += 8 * 1024) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + o + u * 1024 + l * 16);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= __float_as_uint(v[u].x);
      }
    } else if (mode == 2) {
      for (long o = 0; o < cold_per_wave; o += 4 * 8192) {
        unsigned v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const unsigned*>(base + o + u * 8192 + l * 128);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u];
      }
    } else {
      const u64 lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((u64)base));
      const u64 hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((u64)base >> 32));
      const char* sb = reinterpret_cast<const char*>(lo | (hi << 32));
      for (long o = 0; o < cold_per_wave; o += 1024) s_touch8(sb + o, acc);
    }
  }
  const u64 t1 = wall_clock64();
  if (l == 0) {
    stamps[((long)blockIdx.x * nw + w) * 2] = t0;
    stamps[((long)blockIdx.x * nw + w) * 2 + 1] = t1;
  }
  if (acc == 0x12345677u) sink[0] = acc;
}

extern "C" int run_mix(const char* hot, long hot_bytes, const char* cold, long cold_per_wave, int passes, int n_stream, int mode,
                       int waves, int grid, u64* stamps, unsigned* sink, void* st) {
  hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(waves * 64), 0, (hipStream_t)st, hot, hot_bytes, cold, cold_per_wave, passes, n_stream,
                     mode, stamps, sink);
  return (int)hipGetLastError();
}

// warm-then-read: 16 waves touch `bytes` of this workgroup's region with scalar loads (stride `tstride` bytes; 0 = no touch), wait,
// then read the region with 16-byte vector loads; stamps = ticks of the read phase.  Each round uses a fresh region.
extern "C" __global__ void __launch_bounds__(1024) warm_read_kernel(const char* cold, long region, int bytes, int tstride, int rounds,
                                                                    int wait_ticks, u64* stamps, unsigned* sink) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  unsigned acc = 0;
  u64 total = 0;
  for (int r = 0; r < rounds; ++r) {
    const char* base = cold + ((long)blockIdx.x * rounds + r) * region;
    if (tstride) {
      const u64 lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((u64)base));
      const u64 hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((u64)base >> 32));
      const char* sb = reinterpret_cast<const char*>(lo | (hi << 32));
      // wave w touches lines w * 8 ..., in batches of 8
      const int nline = bytes / tstride;
      const int wu = __builtin_amdgcn_readfirstlane(w);
      for (int i = wu * 8; i < nline; i += 16 * 8) {
        unsigned r0, r1, r2, r3, r4, r5, r6, r7;
        const char* q = sb + (long)i * tstride;
#define TL(rr, k) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(rr) : "s"(q + (long)(k) * tstride) : "memory")
        TL(r0, 0); TL(r1, 1); TL(r2, 2); TL(r3, 3); TL(r4, 4); TL(r5, 5); TL(r6, 6); TL(r7, 7);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3), "+s"(r4), "+s"(r5), "+s"(r6), "+s"(r7)::"memory");
        acc ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
      }
    }
    __syncthreads();
    if (wait_ticks) { const u64 t = wall_clock64(); while (wall_clock64() - t < (u64)wait_ticks) __builtin_amdgcn_s_sleep(2); }
    __syncthreads();
    const u64 t0 = wall_clock64();
    for (int o = threadIdx.x * 16; o < bytes; o += 1024 * 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + o);
      acc ^= __float_as_uint(v.x) ^ __float_as_uint(v.w);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    total += wall_clock64() - t0;
  }
  if (threadIdx.x == 0) stamps[blockIdx.x] = total;
  if (acc == 0x12345677u) sink[0] = acc;
}
extern "C" int run_warm(const char* cold, long region, int bytes, int tstride, int rounds, int wait_ticks, int grid, u64* stamps,
                        unsigned* sink, void* st) {
  hipLaunchKernelGGL(warm_read_kernel, dim3(grid), dim3(1024), 0, (hipStream_t)st, cold, region, bytes, tstride, rounds, wait_ticks, stamps, sink);
  return (int)hipGetLastError();
}

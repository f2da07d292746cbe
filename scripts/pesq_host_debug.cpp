// Debug aid (not part of the package, never shipped as a fallback): compiles csrc/pesq_core.h for the HOST with a one-thread
// team so that the control flow of the PESQ kernels can be stepped through and compared with oracle/pesq_ref.py without a GPU.
//   g++ -O2 -DPQ_HOST -I urgent2026_challenge_track1_amd/csrc scripts/pesq_host_debug.cpp -o /tmp/pesq_host
//   /tmp/pesq_host fs wb ref.f32 deg.f32   -> prints raw, mos and the trace
#define PQ_HOST 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "pesq_core.h"

using namespace pesq;

static std::vector<float> load(const char* path) {
  FILE* f = fopen(path, "rb");
  fseek(f, 0, SEEK_END);
  long n = ftell(f) / 4;
  fseek(f, 0, SEEK_SET);
  std::vector<float> v(n);
  if (fread(v.data(), 4, n, f) != (size_t)n) exit(2);
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  const int fs = atoi(argv[1]), wb = atoi(argv[2]);
  std::vector<float> ref = load(argv[3]), deg = load(argv[4]);
  Params P;
  P.fs = fs; P.wb = wb; P.ds = fs == 8000 ? 32 : 64; P.align_nfft = fs == 8000 ? 512 : 1024; P.pad = 320 * (fs / 1000);
  P.tb = fs == 8000 ? &TABLES_8K : &TABLES_16K; P.nb = P.tb->nb;
  const int twn = 1 << 18;
  std::vector<float2> tw(twn / 2);
  for (int k = 0; k < twn / 2; ++k) { tw[k].x = (float)cos(-2.0 * M_PI * k / twn); tw[k].y = (float)sin(-2.0 * M_PI * k / twn); }
  P.tw = tw.data(); P.twn = twn;
  const int sb = SEARCHBUFFER * P.ds;
  const int L = (int)ref.size();
  const int nsamp = L + 2 * sb, na = nsamp + P.pad + 4 * P.align_nfft + 64;
  const int p2 = nextpow2(nsamp - 2 * sb + P.pad) < 65536 ? 65536 : nextpow2(nsamp - 2 * sb + P.pad);
  const int nw = na / P.ds + 8, nfr = na / (4 * P.ds) + 8;
  Pair S;
  std::vector<float> buf((size_t)4 * na + 4 * nw + 2 * na + (size_t)2 * nfr * 49 + 3 * nfr + 16 * nw + 8 * 1024 + 8 * nfr + 1024);
  float* p = buf.data();
  float mx = 0;
  for (int i = 0; i < L; ++i) { mx = fmaxf(mx, fabsf(ref[i])); mx = fmaxf(mx, fabsf(deg[i])); }
  const float sc = 32768.f / (mx > 1.f ? mx : 1.f);
  for (int s = 0; s < 2; ++s) { S.data[s] = p; p += na; S.adata[s] = p; p += na; S.vad[s] = p; p += nw; S.logvad[s] = p; p += nw; S.nsamp[s] = nsamp; }
  for (int i = 0; i < L; ++i) { S.data[0][sb + i] = ref[i] * sc; S.data[1][sb + i] = deg[i] * sc; }
  S.na = na;
  std::vector<float2> ca(p2), cb(p2);
  S.ca = ca.data(); S.cb = cb.data(); S.p2max = p2;
  S.tweaked = p; p += na; S.doubly = p; p += na;
  S.ppd_ref = p; p += (size_t)nfr * 49; S.ppd_deg = p; p += (size_t)nfr * 49;
  S.fd = p; p += nfr; S.fda = p; p += nfr; S.tpr = p; p += nfr;
  S.scratch = p;
  std::vector<int> st(I_COUNT);
  std::vector<float> fst(F_COUNT);
  S.st = st.data(); S.fst = fst.data();
  std::vector<float2> la(1024), lb(1024);
  std::vector<float> x(1024), h(1024), w(4096), iir(512);
  Lds Ld;
  Ld.la = la.data(); Ld.lb = lb.data(); Ld.x = x.data(); Ld.h = h.data(); Ld.w = w.data(); Ld.wcap = 4096; Ld.iir = iir.data();
  double red[4];
  int ired[4];
  Team T;
  T.tid = 0; T.nt = 1; T.red = red; T.ired = ired;
  std::vector<int> trace(TRACE_INTS);
  const float raw = pesq_pair(T, P, S, Ld, trace.data());
  const float mos = wb ? 0.999f + 4.0f / (1.0f + expf(-1.3669f * raw + 3.8224f)) : 0.999f + 4.0f / (1.0f + expf(-1.4945f * raw + 4.6607f));
  printf("raw %.6f mos %.6f crude %d nutt %d start_frame %d stop_frame %d nbad %d\n", raw, mos, trace[0], trace[1], trace[2], trace[3], trace[4]);
  for (int u = 0; u < trace[1]; ++u) printf("utt %d: start %d end %d delay %d\n", u, trace[8 + u], trace[8 + MAXNUTT + u], trace[8 + 2 * MAXNUTT + u]);
  for (int q = 0; q < trace[4]; ++q) printf("bad %d: %d %d\n", q, trace[8 + 3 * MAXNUTT + 2 * q], trace[8 + 3 * MAXNUTT + 2 * q + 1]);
  return 0;
}

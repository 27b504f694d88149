// Development harness: pairwise_fused_kernel on its own, launched repeatedly on the same inputs; every output array is compared
// bit for bit with the first launch's.  (Race hunting: the kernel must be deterministic.)
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -Idust_amd/csrc tools/fused_race.hip -o /tmp/fused_race && /tmp/fused_race N D da reps slots spread
#include "pairwise_big.hpp"
#include "pairwise_fused.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace dust;
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

template <int DPB>
static void launch(const PairFusedArgs &b, dim3 grid) {
  const size_t lds = pairwise_fused_lds_bytes<DPB>();
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)pairwise_fused_kernel<PAIR_K1, DPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  pairwise_fused_kernel<PAIR_K1, DPB><<<grid, PAIR_NT, lds>>>(b);
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2254, D = argc > 2 ? atoi(argv[2]) : 56, da = argc > 3 ? atoi(argv[3]) : 1;
  const int reps = argc > 4 ? atoi(argv[4]) : 40, JSreq = argc > 5 ? atoi(argv[5]) : 512;
  const float spread = argc > 6 ? atof(argv[6]) : 0.05f;
  const int dpb = D <= 32 ? 32 : (D <= 64 ? 64 : 80);
  const int TQ = dpb == 32 ? FusedGeom<32>::TQ : (dpb == 64 ? FusedGeom<64>::TQ : FusedGeom<80>::TQ);
  const int tiles = (N + TQ - 1) / TQ, chunks = (N + 63) / 64;
  int W, JS;
  fused_balance(tiles, chunks, JSreq /* resident workgroup slots */, &W, &JS);
  const int slice = 0;
  const int ldp = ((D + 31) / 32) * 32, ldK = chunks * 64;
  std::mt19937 g(N + D);
  std::normal_distribution<float> nd;
  std::vector<float> X((size_t)N * D), lm(N);
  for (auto &v : X) v = spread * nd(g);
  for (auto &v : lm) v = std::log(0.05f + (float)(g() % 1000) / 1000.f);
  float *dX, *dXp, *dlm, *pA, *pB, *pM, *pL, *K;
  const size_t nd_ = (size_t)JS * N * ldp, nn = (size_t)JS * N, nk = (size_t)tiles * TQ * ldK;
  CK(hipMalloc(&dX, X.size() * 4));
  CK(hipMalloc(&dXp, (size_t)N * dpb * 4));
  CK(hipMalloc(&dlm, N * 4));
  CK(hipMalloc(&pA, nd_ * 4));
  CK(hipMalloc(&pB, nd_ * 4));
  CK(hipMalloc(&pM, nn * 4));
  CK(hipMalloc(&pL, nn * 4));
  CK(hipMalloc(&K, nk * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlm, lm.data(), N * 4, hipMemcpyHostToDevice));
  pad_rows_kernel<<<(N * dpb + 255) / 256, 256>>>(dX, dXp, N, D, dpb);
  PairFusedArgs b;
  memset(&b, 0, sizeof b);
  b.p.N = N; b.p.D = D; b.p.da = da; b.p.H = D / da; b.p.i0 = 0; b.p.n_local = N; b.p.JS = JS; b.p.slice = slice;
  b.p.X = dX; b.p.Y = dX; b.p.logmix = dlm; b.p.pA = pA; b.p.pM = pM; b.p.pL = pL;
  b.Xp = dXp; b.ldp = ldp; b.wP[0] = 1.f / (1.5f * 1.5f); b.wP[1] = da == 2 ? 1.f / (0.8f * 0.8f) : b.wP[0];
  b.wS[0] = b.wS[1] = 1.f / (0.6931472f * 0.6931472f); b.pB = pB; b.K = K; b.ldK = ldK; b.tiles = tiles; b.chunks = chunks;
  dim3 grid(W);
  printf("N %d D %d dpb %d TQ %d tiles %d chunks %d workgroups %d JS %d ldp %d WGS %d\n", N, D, dpb, TQ, tiles, chunks, W, JS, ldp, DUST_FUSED_WGS);
  (void)slice;
  std::vector<float> r[5], cur[5];
  const size_t sz[5] = {nd_, nd_, nn, nn, nk};
  float *dev[5] = {pA, pB, pM, pL, K};
  const char *nm[5] = {"pA", "pB", "pM", "pL", "K"};
  int bad = 0;
  for (int it = 0; it < reps; ++it) {
    for (int k = 0; k < 5; ++k) CK(hipMemset(dev[k], 0xff, sz[k] * 4));
    if (dpb == 32) launch<32>(b, grid);
    else if (dpb == 64) launch<64>(b, grid);
    else launch<80>(b, grid);
    CK(hipDeviceSynchronize());
    for (int k = 0; k < 5; ++k) {
      cur[k].resize(sz[k]);
      CK(hipMemcpy(cur[k].data(), dev[k], sz[k] * 4, hipMemcpyDeviceToHost));
    }
    if (it == 0) {
      for (int k = 0; k < 5; ++k) r[k] = cur[k];
      continue;
    }
    for (int k = 0; k < 5; ++k) {
      size_t ndiff = 0, first = 0;
      for (size_t i = 0; i < sz[k]; ++i)
        if (memcmp(&cur[k][i], &r[k][i], 4)) {
          if (!ndiff) first = i;
          ++ndiff;
        }
      if (ndiff) {
        ++bad;
        if (k < 2) {
          const size_t row = first / ldp;
          printf("rep %d %s: %zu words differ, first js %zu row %zu col %zu: %g vs %g\n", it, nm[k], ndiff, row / N, row % N, first % ldp, cur[k][first], r[k][first]);
        } else if (k < 4) printf("rep %d %s: %zu words differ, first js %zu row %zu: %g vs %g\n", it, nm[k], ndiff, first / N, first % N, cur[k][first], r[k][first]);
        else {
          const size_t qi = first / ldK, kj = first % ldK;
          double d2 = 0;
          for (int d = 0; d < D; ++d) {
            const float z = X[qi * D + d] - X[kj * D + d];
            d2 += (double)z * z;
          }
          size_t lastw = first;
          for (size_t i = first; i < sz[k]; ++i)
            if (memcmp(&cur[k][i], &r[k][i], 4)) lastw = i;
          printf("rep %d K: %zu words differ, first row %zu key %zu (last row %zu key %zu): %g vs %g (rep 0); host %g\n", it, ndiff, qi, kj, lastw / ldK,
                 lastw % ldK, cur[k][first], r[k][first], std::exp(-0.5 * d2 * b.wS[0]));
        }
      }
    }
  }
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int it = 0; it < 30; ++it) {
      if (it == 10) CK(hipEventRecord(e0));
      if (dpb == 32) launch<32>(b, grid);
      else if (dpb == 64) launch<64>(b, grid);
      else launch<80>(b, grid);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%.1f us per launch\n", ms * 50.f);
  }
  printf("%s: %d differing arrays over %d repeats\n", bad ? "NONDETERMINISTIC" : "deterministic", bad, reps - 1);
  return 0;
}

// Development harness: gram_score_kernel (pass 2 of the fused large-set pairwise pair) alone, timed; ablation builds (-DGP_NO_K,
// -DGP_NO_MFMA, -DGP_NO_V through a patched copy of pairwise_fused.hpp) tell which of its phases bounds it.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -Idust_amd/csrc tools/gram_probe.hip -o /tmp/gram_probe && /tmp/gram_probe 16384 80 8
#include "pairwise_big.hpp"
#include "pairwise_fused.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16384, D = argc > 2 ? atoi(argv[2]) : 80, JS = argc > 3 ? atoi(argv[3]) : 8;
  const int chunks = (N + 63) / 64, ldK = chunks * 64, cps = (chunks + JS - 1) / JS, slice = cps * 64, js = (N + slice - 1) / slice;
  const int ldp = ((D + 31) / 32) * 32;
  float *K, *V, *pA;
  CK(hipMalloc(&K, (size_t)N * ldK * 4));
  CK(hipMalloc(&V, (size_t)N * D * 4));
  CK(hipMalloc(&pA, (size_t)js * N * ldp * 4));
  CK(hipMemset(K, 0, (size_t)N * ldK * 4));
  CK(hipMemset(V, 0, (size_t)N * D * 4));
  GramScoreArgs g;
  memset(&g, 0, sizeof g);
  g.N = N; g.D = D; g.n_local = N; g.JS = js; g.slice = slice; g.ldp = ldp; g.ldK = ldK; g.K = K; g.V = V; g.pA = pA;
  dim3 grid((N + 63) / 64, js);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int it = 0; it < 25; ++it) {
    if (it == 5) CK(hipEventRecord(e0));
    if (D <= 32) gram_score_kernel<32><<<grid, PAIR_NT, gram_score_lds_bytes<32>()>>>(g);
    else if (D <= 64) gram_score_kernel<64><<<grid, PAIR_NT, gram_score_lds_bytes<64>()>>>(g);
    else gram_score_kernel<80><<<grid, PAIR_NT, gram_score_lds_bytes<80>()>>>(g);
  }
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("N %d D %d slices %d: %.1f us per launch (%.2f TB/s of K, %.1f TFLOP/s)\n", N, D, js, ms * 50.f, (double)N * ldK * 4 / (ms / 20 * 1e-3) / 1e12,
         2.0 * N * (double)N * D / (ms / 20 * 1e-3) / 1e12);
  return 0;
}

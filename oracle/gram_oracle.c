/* CPU oracle in C: the evaluation order of the reference's emission term.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  custom_hmm.py:168-172 evaluates
 *     np.sum(diff.T @ inv_cov @ diff, axis=1)
 * with two BLAS products.  On the build the golden vectors were generated with (numpy 2.2.6 wheels,
 * OpenBLAS 0.3.29, SkylakeX kernels) every element of both products is ONE fused-multiply-add chain over
 * the contraction index in increasing order, starting from +0.0.  This file states that order explicitly, so
 * the oracle no longer depends on which BLAS the machine running the tests has.  PINNED: the emission
 * matrices it yields are compared bit for bit with the reference's own (tests/golden/custom_hmm_golden.npz,
 * g2_*_E and g5_s16_d39_E) in tests/test_oracle_custom.py.
 */
#include <math.h>

/* C[M][N] = A[M][K] @ B[K][N], row-major, each element a k-ascending fma chain from 0 */
void oracle_matmul_fma_chain(const double *A, const double *B, double *C, int M, int K, int N) {
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc = fma(A[m * K + k], B[k * N + n], acc);
      C[m * N + n] = acc;
    }
}

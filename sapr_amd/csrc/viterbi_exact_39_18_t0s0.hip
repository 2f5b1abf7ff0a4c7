// Exact Viterbi kernels for D = 39 features, S = 18 kernel states, tie rule low, pair-wise sum
// (see viterbi_exact.inc / viterbi.hip).  This shape's unrolled bodies are the longest compiles of the library, so each
// (tie, summation order) pair is a translation unit of its own.
#include "viterbi_exact.inc"

namespace sapr {
int launch_scores_39_18_t0s0(const ScoreArgs &a, int topology, int fast) {
  return fast ? launch_scores4<39, 18, false, false, true>(a, topology)
              : launch_scores4<39, 18, false, false, false>(a, topology);
}
}  // namespace sapr

// Exact Viterbi kernels for D = 39 features, S = 18 kernel states, tie rule high, left-to-right sum
// (see viterbi_exact.inc / viterbi.hip).  This shape's unrolled bodies are the longest compiles of the library, so each
// (tie, summation order) pair is a translation unit of its own.
#include "viterbi_exact.inc"

namespace sapr {
int launch_scores_39_18_t1s1(const ScoreArgs &a, int topology, int fast) {
  return fast ? launch_scores4<39, 18, true, true, true>(a, topology)
              : launch_scores4<39, 18, true, true, false>(a, topology);
}
}  // namespace sapr

// Exact Viterbi kernels for D = 13 features, S = 10 kernel states (see viterbi_exact.inc / viterbi.hip).
#include "viterbi_exact.inc"

namespace sapr {
int launch_scores_13_10(const ScoreArgs &a, int topology, int tie, int sum_order, int fast) {
  return launch_scores<13, 10>(a, topology, tie, sum_order, fast);
}
}  // namespace sapr

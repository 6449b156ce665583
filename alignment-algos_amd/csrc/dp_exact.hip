// dp_exact.hip — exact-order O(n^3) DP (placeholder until the kernel lands in the next commit).
#include "aln_internal.h"
namespace aln {
int launch_dp_exact(aln_batch* b) { (void)b; return ALN_E_ARG; }
}  // namespace aln

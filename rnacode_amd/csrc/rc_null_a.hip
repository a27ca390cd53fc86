// rc_null_a.hip -- k_null<N-1, LDSC, EXACT> for N-1 in [2, 6] (see rc_null_kernel.h)
#include "rc_null_kernel.h"

namespace rc {
RC_DEFINE_NULL_RANGE(a, 2, 6)
}  // namespace rc

// rc_null_b.hip -- k_null<N-1, LDSC, EXACT> for N-1 in [7, 12] (see rc_null_kernel.h)
#include "rc_null_kernel.h"

namespace rc {
RC_DEFINE_NULL_RANGE(b, 7, 12)
}  // namespace rc

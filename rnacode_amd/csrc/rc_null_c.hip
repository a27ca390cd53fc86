// rc_null_c.hip -- k_null<N-1, LDSC, EXACT> for N-1 in [13, 21] (see rc_null_kernel.h)
#include "rc_null_kernel.h"

namespace rc {
RC_DEFINE_NULL_RANGE(c, 13, 21)
}  // namespace rc

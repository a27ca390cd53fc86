// rc_null_d.hip -- k_null<N-1, LDSC, EXACT> for N-1 in [22, 31] (see rc_null_kernel.h)
#include "rc_null_kernel.h"

namespace rc {
RC_DEFINE_NULL_RANGE(d, 22, 31)
}  // namespace rc

// rc_null_e.hip -- k_null<N-1, false, true> for N-1 in [32, 39]: wide blocks (see rc_null_kernel.h)
#include "rc_null_kernel.h"

namespace rc {
RC_DEFINE_NULL_WIDE_RANGE(e, 32, 39)
}  // namespace rc

// rc_null_f.hip -- k_null<N-1, false, true> for N-1 in [40, 47]: wide blocks (see rc_null_kernel.h)
#include "rc_null_kernel.h"

namespace rc {
RC_DEFINE_NULL_WIDE_RANGE(f, 40, 47)
}  // namespace rc

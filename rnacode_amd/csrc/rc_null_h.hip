// rc_null_h.hip -- k_null<N-1, false, true> for N-1 in [56, 63]: wide blocks (see rc_null_kernel.h)
#include "rc_null_kernel.h"

namespace rc {
RC_DEFINE_NULL_WIDE_RANGE(h, 56, 63)
}  // namespace rc

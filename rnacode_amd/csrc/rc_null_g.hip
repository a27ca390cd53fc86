// rc_null_g.hip -- k_null<N-1, false, true> for N-1 in [48, 55]: wide blocks (see rc_null_kernel.h)
#include "rc_null_kernel.h"

namespace rc {
RC_DEFINE_NULL_WIDE_RANGE(g, 48, 55)
}  // namespace rc

// k_search_fast.hip -- K2, fast variant (packed-u8 quad-SAD).  Placeholder until the kernel lands:
// every configuration is routed to the generic variant.
#include "rtdm_kernels.h"

namespace rtdm {
bool fast_search_supported(const BMGeom&) { return false; }
void launch_search_fast(Plane8, Plane8, Plane16W, int32_t*, const BMGeom&, int, hipStream_t) {}
}  // namespace rtdm

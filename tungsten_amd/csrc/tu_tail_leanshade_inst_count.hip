// One kernel of kernels.h, instantiated here so that the library builds in parallel (see the end of kernels.h): the tail kernel of scenes that
// shade with the lean kernels (k_tail<..., LEAN = true>).
#include "kernels.h"

namespace tg {
TG_KERNELS_TAIL_LEAN(, true, true)
}

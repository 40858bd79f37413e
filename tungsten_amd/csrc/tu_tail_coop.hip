// One kernel family of kernels.h, instantiated here so that the library builds in parallel (see the end of kernels.h): the four-lanes-per-path tail
// kernel of scenes with plain geometry and lean shading (k_tail_coop), for scenes with BSDFs made of BSDFs.
#include "kernels.h"

namespace tg {
TG_KERNELS_TAIL_COOP(, false, true)
TG_KERNELS_TAIL_COOP(, true, true)
}

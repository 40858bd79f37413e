// One kernel family of kernels.h, instantiated here so that the library builds in parallel (see the end of kernels.h): the four-lanes-per-path tail
// kernel of scenes with plain geometry and lean shading (k_tail_coop), for scenes without BSDFs made of BSDFs (the benchmark's kind).
#include "kernels.h"

namespace tg {
TG_KERNELS_TAIL_COOP(, false, false)
TG_KERNELS_TAIL_COOP(, true, false)
}

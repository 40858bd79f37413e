// One kernel family of kernels.h, instantiated here so that the library builds in parallel (see the end of kernels.h).
#include "kernels.h"

namespace tg {
TG_KERNELS_INSTQ()
}

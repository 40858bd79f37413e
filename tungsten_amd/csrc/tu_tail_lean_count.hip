// One kernel family of kernels.h, instantiated here so that the library builds in parallel (see the end of kernels.h).
#define TG_NO_NESTED_BSDFS /* see device.h: scenes with nested BSDFs never run the tail kernel */
#include "kernels.h"

namespace tg {
TG_KERNELS_TAIL(, true, false)
}

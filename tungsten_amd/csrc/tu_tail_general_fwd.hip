// One kernel of kernels.h, instantiated here so that the library builds in parallel (see the end of kernels.h): the whole-path tail kernel is the
// slowest thing to compile in the tree, one instantiation per file keeps the critical path of `make -j` short.
#include "kernels.h"

namespace tg {
TG_KERNELS_TAIL(, false, true, true)
}

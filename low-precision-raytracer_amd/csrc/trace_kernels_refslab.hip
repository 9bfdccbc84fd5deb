// The reference-arithmetic half slab variant of the ray-batch kernels (kernels.hpp: namespace refslab).
#define LPRT_REFSLAB_TU 1
#include "trace_kernels.hip"

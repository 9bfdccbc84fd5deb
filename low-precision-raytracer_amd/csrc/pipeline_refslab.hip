// The reference-arithmetic half slab variant of the frame's ray-casting kernels (kernels.hpp: namespace refslab).
#define LPRT_REFSLAB_TU 1
#include "pipeline.hip"

// Fast arithmetic mode: same kernels, FMA contraction allowed.
#define CASTER_STRICT 0
#define CASTER_SFX fast
#include "caster_kernels.hpp"

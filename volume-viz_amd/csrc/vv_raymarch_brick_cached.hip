// vv_raymarch_brick_cached.hip -- vv_raymarch_brick.hip once more for volumes that live in the caches (up to 1 GiB):
// the same kernels in namespace brickc, compiled without the SLP vectoriser (Makefile; see vv_raymarch.hip).
#define VV_BRICKED 1
#define VV_BRICKED_CACHED 1
#include "vv_raymarch.hip"

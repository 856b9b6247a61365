// vv_raymarch_zfast.hip -- the ray-march kernels instantiated on the z-fastest copy of an f32 volume
// (VolumeView::zfast), used for views whose screen x runs along the volume's z axis (side views): the
// lanes of a 32 x 2 wave tile read consecutive z, the rays march along x.
#define VV_ZFAST 1
#include "vv_raymarch.hip"

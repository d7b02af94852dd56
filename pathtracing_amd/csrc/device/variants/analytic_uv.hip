// Scenes of spheres, planes and rectangles with texture coordinates.
#include "define.h"

namespace ptk {
PTK_BOUNCE_VARIANT(false, true, false, false, true)
PTK_BOUNCE_VARIANT(false, true, false, true, true)
PTK_TAIL_VARIANT(false, true, false, true)
PTK_TAIL_VARIANT(false, true, true, true)
} // namespace ptk

// Scenes of spheres, planes and rectangles without texture coordinates.
#include "define.h"

namespace ptk {
PTK_BOUNCE_VARIANT(false, false, false, false, true)
PTK_BOUNCE_VARIANT(false, false, false, true, true)
PTK_TAIL_VARIANT(false, false, false, true)
PTK_TAIL_VARIANT(false, false, true, true)
} // namespace ptk

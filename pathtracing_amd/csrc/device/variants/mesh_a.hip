// Scenes with meshes: no texture coordinates, no pass-through materials; the tail kernel with texture coordinates.
#include "define.h"

namespace ptk {
PTK_TAIL_VARIANT(true, true, true, true)
PTK_BOUNCE_VARIANT(true, false, true, false, true)
PTK_RESUME_VARIANT(false, false, true)
} // namespace ptk

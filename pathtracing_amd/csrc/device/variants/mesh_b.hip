// Scenes with meshes: no texture coordinates, pass-through materials.
#include "define.h"

namespace ptk {
PTK_BOUNCE_VARIANT(true, false, true, true, true)
PTK_RESUME_VARIANT(false, true, true)
} // namespace ptk

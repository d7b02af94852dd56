// Scenes with meshes: texture coordinates, no pass-through materials.
#include "define.h"

namespace ptk {
PTK_BOUNCE_VARIANT(true, true, true, false, true)
PTK_RESUME_VARIANT(true, false, true)
} // namespace ptk

// Scenes with meshes: texture coordinates and pass-through materials.
#include "define.h"

namespace ptk {
PTK_BOUNCE_VARIANT(true, true, true, true, true)
PTK_RESUME_VARIANT(true, true, true)
} // namespace ptk

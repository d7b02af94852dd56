// Scenes with single triangles but no meshes; the tail kernel of every scene with triangles, without texture coordinates.
#include "define.h"

namespace ptk {
PTK_BOUNCE_VARIANT(true, false, false, true, true)
PTK_BOUNCE_VARIANT(true, true, false, true, true)
PTK_TAIL_VARIANT(true, false, true, true)
} // namespace ptk
